// pg_comm.hip -- the one exchange step of the multi-GPU path: an in-place
// sum of sdTree_current's int64 accumulators over the ranks with RCCL (ncclAllReduce, ncclInt64,
// ncclSum) before pg_refine_and_swap.  The reference has no multi-GPU code; this is new design around
// refineAndPrepareSDTreeForNextIteration (src/path_guiding_integrator.py:566-586).
//
// RCCL is bound at run time (dlopen): libpgsd.so has no link-time dependency on it, a single-GPU
// host never loads it, and a process that already holds an RCCL (PyTorch ships its own copy) keeps
// exactly one -- the copy already loaded is the one used.  PGSD_RCCL_LIBRARY names a specific file.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "pg_context.hpp"

namespace {

typedef struct { char internal[128]; } rccl_unique_id; // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *rccl_comm;                               // ncclComm_t
constexpr int kNcclSuccess = 0, kNcclInt64 = 4, kNcclSum = 0;

struct Rccl {
	void *handle = nullptr;
	int (*GetUniqueId)(rccl_unique_id *) = nullptr;
	int (*CommInitRank)(rccl_comm *, int, rccl_unique_id, int) = nullptr;
	int (*CommDestroy)(rccl_comm) = nullptr;
	int (*AllReduce)(const void *, void *, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
	int (*CommCount)(const rccl_comm, int *) = nullptr;    // optional: the witness of pg_comm_info
	int (*CommUserRank)(const rccl_comm, int *) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	std::string why; // why loading failed
};

Rccl &rccl()
{
	static Rccl r;
	if (r.handle || !r.why.empty()) return r;
	const char *env = getenv("PGSD_RCCL_LIBRARY");
	const char *names[] = {"librccl.so.1", "librccl.so"};
	if (env && *env) r.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
	for (int pass = 0; pass < 2 && !r.handle; ++pass) // first a copy the process already holds, then the search path
		for (const char *n : names) {
			r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
			if (r.handle) break;
		}
	if (!r.handle) {
		const char *e = dlerror();
		r.why = std::string("cannot load RCCL (librccl.so.1): ") + (e ? e : "not found");
		return r;
	}
	r.GetUniqueId = (int (*)(rccl_unique_id *))dlsym(r.handle, "ncclGetUniqueId");
	r.CommInitRank = (int (*)(rccl_comm *, int, rccl_unique_id, int))dlsym(r.handle, "ncclCommInitRank");
	r.CommDestroy = (int (*)(rccl_comm))dlsym(r.handle, "ncclCommDestroy");
	r.AllReduce = (int (*)(const void *, void *, size_t, int, int, rccl_comm, hipStream_t))dlsym(r.handle, "ncclAllReduce");
	r.GetErrorString = (const char *(*)(int))dlsym(r.handle, "ncclGetErrorString");
	r.CommCount = (int (*)(const rccl_comm, int *))dlsym(r.handle, "ncclCommCount");
	r.CommUserRank = (int (*)(const rccl_comm, int *))dlsym(r.handle, "ncclCommUserRank");
	if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
		r.why = "the RCCL library found lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
		r.handle = nullptr;
	}
	return r;
}

int rccl_fail(pg_context *ctx, const char *what, int rc)
{
	Rccl &r = rccl();
	const char *msg = r.GetErrorString ? r.GetErrorString(rc) : "?";
	return pg::fail(ctx, PG_ERR_HIP, std::string(what) + ": " + (msg ? msg : "?"));
}

// ---- the 24-byte exchange format (pg_math.hpp: xchg_pack / xchg_unpack) ----
constexpr int kXBlk = 256;

__global__ __launch_bounds__(kXBlk) void k_xchg_pack(const long long *__restrict__ acc, uint64_t n_acc, uint64_t n_tail,
                                                     long long *__restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * kXBlk + threadIdx.x;
	if (i < n_acc) {
		const uint4 lo = pg::gather16(acc + i * pg::kAccWords);       // words 0, 1 (two 16-byte loads of one 32-byte sector)
		const uint4 hi = pg::gather16(acc + i * pg::kAccWords + 2);   // words 2, 3
		const long long a[4] = {(long long)(((uint64_t)lo.y << 32) | lo.x), (long long)(((uint64_t)lo.w << 32) | lo.z),
		                        (long long)(((uint64_t)hi.y << 32) | hi.x), (long long)(((uint64_t)hi.w << 32) | hi.z)};
		long long p[3];
		pg::xchg_pack(a, p);
		long long *o = out + i * pg::kXchgWords;
		o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
	} else if (i < n_acc + n_tail) {
		out[n_acc * pg::kXchgWords + (i - n_acc)] = acc[n_acc * pg::kAccWords + (i - n_acc)]; // the fallback counters as they are
	}
}

__global__ __launch_bounds__(kXBlk) void k_xchg_unpack(const long long *__restrict__ in, uint64_t n_acc, uint64_t n_tail,
                                                       long long *__restrict__ acc)
{
	const uint64_t i = (uint64_t)blockIdx.x * kXBlk + threadIdx.x;
	if (i < n_acc) {
		const long long *q = in + i * pg::kXchgWords;
		const long long p[3] = {q[0], q[1], q[2]};
		long long a[4];
		pg::xchg_unpack(p, a);
		long long *o = acc + i * pg::kAccWords;
		o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
	} else if (i < n_acc + n_tail) {
		acc[n_acc * pg::kAccWords + (i - n_acc)] = in[n_acc * pg::kXchgWords + (i - n_acc)];
	}
}

int xchg_pack_launch(pg_context *ctx, hipStream_t s)
{
	pg::Forest &f = ctx->f;
	const uint64_t n = f.n_acc(), tail = f.n_trees;
	if (f.xchg.ensure((size_t)f.xchg_count()) != hipSuccess) {
		(void)hipGetLastError();
		return pg::fail(ctx, PG_ERR_NOMEM, "pg_exchange_pack: cannot allocate the exchange buffer");
	}
	hipLaunchKernelGGL(k_xchg_pack, dim3((unsigned)((n + tail + kXBlk - 1) / kXBlk)), dim3(kXBlk), 0, s, f.acc.p, n, tail, f.xchg.p);
	PG_HIP(ctx, hipGetLastError());
	return PG_OK;
}

int xchg_unpack_launch(pg_context *ctx, hipStream_t s)
{
	pg::Forest &f = ctx->f;
	const uint64_t n = f.n_acc(), tail = f.n_trees;
	if (!f.xchg.p || f.xchg.cap < f.xchg_count()) return pg::fail(ctx, PG_ERR_INVALID, "pg_exchange_unpack: call pg_exchange_pack first");
	hipLaunchKernelGGL(k_xchg_unpack, dim3((unsigned)((n + tail + kXBlk - 1) / kXBlk)), dim3(kXBlk), 0, s, f.xchg.p, n, tail, f.acc.p);
	PG_HIP(ctx, hipGetLastError());
	return PG_OK;
}

} // namespace

void pg::destroy_comm(pg_context *ctx)
{
	if (ctx->comm && ctx->comm_owned) {
		Rccl &r = rccl();
		if (r.CommDestroy) (void)r.CommDestroy((rccl_comm)ctx->comm);
	}
	ctx->comm = nullptr;
	ctx->comm_owned = false;
}

extern "C" {

int pg_comm_unique_id(pg_context *ctx, uint8_t *h_id_out)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!h_id_out) return pg::fail(ctx, PG_ERR_INVALID, "pg_comm_unique_id: NULL pointer");
	Rccl &r = rccl();
	if (!r.handle) return pg::fail(ctx, PG_ERR_INVALID, r.why);
	rccl_unique_id id;
	const int rc = r.GetUniqueId(&id);
	if (rc != kNcclSuccess) return rccl_fail(ctx, "ncclGetUniqueId", rc);
	memcpy(h_id_out, id.internal, PG_COMM_ID_BYTES);
	return PG_OK;
}

int pg_comm_init(pg_context *ctx, int32_t n_ranks, int32_t rank, const uint8_t *h_id)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!h_id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return pg::fail(ctx, PG_ERR_INVALID, "pg_comm_init: bad arguments");
	Rccl &r = rccl();
	if (!r.handle) return pg::fail(ctx, PG_ERR_INVALID, r.why);
	PG_HIP(ctx, hipSetDevice(ctx->device));
	pg::destroy_comm(ctx);
	rccl_unique_id id;
	memcpy(id.internal, h_id, PG_COMM_ID_BYTES);
	rccl_comm comm = nullptr;
	const int rc = r.CommInitRank(&comm, n_ranks, id, rank);
	if (rc != kNcclSuccess) return rccl_fail(ctx, "ncclCommInitRank", rc);
	ctx->comm = comm;
	ctx->comm_owned = true;
	ctx->comm_ranks = n_ranks;
	return PG_OK;
}

int pg_comm_attach(pg_context *ctx, void *nccl_comm, int32_t n_ranks)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!nccl_comm || n_ranks < 1) return pg::fail(ctx, PG_ERR_INVALID, "pg_comm_attach: bad arguments");
	Rccl &r = rccl();
	if (!r.handle) return pg::fail(ctx, PG_ERR_INVALID, r.why);
	pg::destroy_comm(ctx);
	ctx->comm = nccl_comm;
	ctx->comm_owned = false;
	ctx->comm_ranks = n_ranks;
	return PG_OK;
}

int pg_comm_info(pg_context *ctx, int32_t *n_ranks_out, int32_t *rank_out)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!n_ranks_out || !rank_out) return pg::fail(ctx, PG_ERR_INVALID, "pg_comm_info: NULL pointer");
	if (!ctx->comm) return pg::fail(ctx, PG_ERR_INVALID, "pg_comm_info: call pg_comm_init or pg_comm_attach first");
	Rccl &r = rccl();
	if (!r.handle) return pg::fail(ctx, PG_ERR_INVALID, r.why);
	if (!r.CommCount || !r.CommUserRank) return pg::fail(ctx, PG_ERR_INVALID, "the RCCL library found lacks ncclCommCount / ncclCommUserRank");
	int n = -1, me = -1;
	int rc = r.CommCount((rccl_comm)ctx->comm, &n);
	if (rc != kNcclSuccess) return rccl_fail(ctx, "ncclCommCount", rc);
	rc = r.CommUserRank((rccl_comm)ctx->comm, &me);
	if (rc != kNcclSuccess) return rccl_fail(ctx, "ncclCommUserRank", rc);
	*n_ranks_out = n;
	*rank_out = me;
	return PG_OK;
}

int pg_exchange_pack(pg_context *ctx, int64_t **d_buffer, uint64_t *count, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!ctx->configured) return pg::fail(ctx, PG_ERR_INVALID, "call pg_setup or pg_import first");
	if (!d_buffer || !count) return pg::fail(ctx, PG_ERR_INVALID, "pg_exchange_pack: NULL pointer");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	const int rc = xchg_pack_launch(ctx, (hipStream_t)stream);
	if (rc != PG_OK) return rc;
	*d_buffer = reinterpret_cast<int64_t *>(ctx->f.xchg.p);
	*count = ctx->f.xchg_count();
	return PG_OK;
}

int pg_exchange_unpack(pg_context *ctx, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!ctx->configured) return pg::fail(ctx, PG_ERR_INVALID, "call pg_setup or pg_import first");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	return xchg_unpack_launch(ctx, (hipStream_t)stream);
}

/* the same arithmetic on host arrays (no device, no context): what a host-side collective sums, and how the format is tested
 * with eight ranks on a machine without a GPU */
int pg_exchange_pack_words(const int64_t *h_acc, uint64_t n_acc, int64_t *h_out)
{
	if (!h_acc || !h_out) return PG_ERR_INVALID;
	for (uint64_t i = 0; i < n_acc; ++i) {
		const long long a[4] = {h_acc[4 * i], h_acc[4 * i + 1], h_acc[4 * i + 2], h_acc[4 * i + 3]};
		long long p[3];
		pg::xchg_pack(a, p);
		h_out[3 * i] = p[0]; h_out[3 * i + 1] = p[1]; h_out[3 * i + 2] = p[2];
	}
	return PG_OK;
}

int pg_exchange_unpack_words(const int64_t *h_in, uint64_t n_acc, int64_t *h_acc_out)
{
	if (!h_in || !h_acc_out) return PG_ERR_INVALID;
	for (uint64_t i = 0; i < n_acc; ++i) {
		const long long p[3] = {h_in[3 * i], h_in[3 * i + 1], h_in[3 * i + 2]};
		long long a[4];
		pg::xchg_unpack(p, a);
		h_acc_out[4 * i] = a[0]; h_acc_out[4 * i + 1] = a[1]; h_acc_out[4 * i + 2] = a[2]; h_acc_out[4 * i + 3] = a[3];
	}
	return PG_OK;
}

int pg_comm_destroy(pg_context *ctx)
{
	if (!ctx) return PG_ERR_INVALID;
	pg::destroy_comm(ctx);
	return PG_OK;
}

int pg_allreduce(pg_context *ctx, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!ctx->configured) return pg::fail(ctx, PG_ERR_INVALID, "call pg_setup or pg_import first");
	if (!ctx->comm) return pg::fail(ctx, PG_ERR_INVALID, "pg_allreduce: call pg_comm_init or pg_comm_attach first");
	Rccl &r = rccl();
	if (!r.handle) return pg::fail(ctx, PG_ERR_INVALID, r.why);
	PG_HIP(ctx, hipSetDevice(ctx->device));
	if (ctx->f.acc_count() == 0) return PG_OK;
	// topology is frozen during an iteration, so the buffers of all ranks are index-aligned.  What travels is the 24-byte
	// exchange format (pg_math.hpp): pack -> ncclAllReduce of the packed words -> unpack, all on `stream`; element-wise int64
	// sums, no carry handling, any order, exact.  $PGSD_EXCHANGE_RAW=1 sums the 32-byte accumulators in place instead (the
	// form of rounds 1-4: 32 payload bits per limb and counts far below 2^63) -- the same sums, a third more bytes.
	static const bool raw = [] { const char *e = getenv("PGSD_EXCHANGE_RAW"); return e && *e == '1'; }();
	if (raw) {
		const int rc = r.AllReduce(ctx->f.acc.p, ctx->f.acc.p, (size_t)ctx->f.acc_count(), kNcclInt64, kNcclSum, (rccl_comm)ctx->comm,
		                           (hipStream_t)stream);
		if (rc != kNcclSuccess) return rccl_fail(ctx, "ncclAllReduce", rc);
		return PG_OK;
	}
	int rc = xchg_pack_launch(ctx, (hipStream_t)stream);
	if (rc != PG_OK) return rc;
	rc = r.AllReduce(ctx->f.xchg.p, ctx->f.xchg.p, (size_t)ctx->f.xchg_count(), kNcclInt64, kNcclSum, (rccl_comm)ctx->comm,
	                 (hipStream_t)stream);
	if (rc != kNcclSuccess) return rccl_fail(ctx, "ncclAllReduce", rc);
	return xchg_unpack_launch(ctx, (hipStream_t)stream);
}

} // extern "C"
