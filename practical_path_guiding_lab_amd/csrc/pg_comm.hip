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
	const uint64_t count = ctx->f.acc_count();
	if (count == 0) return PG_OK;
	// topology is frozen during an iteration, so the buffers of all ranks are index-aligned; limbs carry
	// 32 payload bits in 64 and counts are far below 2^63: no carry handling, any order, exact
	const int rc = r.AllReduce(ctx->f.acc.p, ctx->f.acc.p, (size_t)count, kNcclInt64, kNcclSum, (rccl_comm)ctx->comm,
	                           (hipStream_t)stream);
	if (rc != kNcclSuccess) return rccl_fail(ctx, "ncclAllReduce", rc);
	return PG_OK;
}

} // extern "C"
