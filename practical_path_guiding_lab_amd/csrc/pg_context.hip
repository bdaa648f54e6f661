// pg_context.hip -- C ABI entry points (include/pgsd.h): context, setup, queries, recording,
// import/export in the reference's npz column schema.  Host code only; kernels live in
// pg_kernels_*.hip and pg_refine.hip.
#include "pg_context.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <limits>

#include "pg_math.hpp"

namespace pg {

static thread_local std::string g_create_error;

int fail(pg_context *ctx, int code, const std::string &msg)
{
	if (ctx) ctx->err = msg;
	else g_create_error = msg;
	return code;
}

int hip_fail(pg_context *ctx, hipError_t e, const char *what)
{
	(void)hipGetLastError();
	return fail(ctx, e == hipErrorOutOfMemory ? PG_ERR_NOMEM : PG_ERR_HIP,
	            std::string(what) + ": " + hipGetErrorString(e));
}

template <class T> static hipError_t upload(DevBuf<T> &d, const std::vector<T> &h, double slack = 1.0)
{
	hipError_t e = d.ensure(h.size(), slack);
	if (e != hipSuccess) return e;
	if (h.empty()) return hipSuccess;
	return hipMemcpy(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

template <class T> static hipError_t download(std::vector<T> &h, const DevBuf<T> &d, size_t n)
{
	h.resize(n);
	if (n == 0) return hipSuccess;
	return hipMemcpy(h.data(), d.p, n * sizeof(T), hipMemcpyDeviceToHost);
}

static int zero_accumulators(pg_context *ctx, hipStream_t s)
{
	Forest &f = ctx->f;
	PG_HIP(ctx, f.acc.ensure(f.acc_count(), 1.25));
	PG_HIP(ctx, hipMemsetAsync(f.acc.p, 0, f.acc_count() * sizeof(long long), s));
	return PG_OK;
}

// Host description of a forest in the device layout; built by import, consumed by install().
struct HostForest {
	std::vector<KdNode> kd;
	std::vector<float> kd_bmin, kd_bmax, kd_vcount;
	std::vector<QuadRec> rec;
	std::vector<uint32_t> level_off;
	std::vector<TreeHead> head;
	std::vector<float> thr;
};

static int install(pg_context *ctx, const HostForest &h)
{
	Forest &f = ctx->f;
	PG_HIP(ctx, upload(f.kd, h.kd, 1.5));
	PG_HIP(ctx, upload(f.kd_bmin, h.kd_bmin, 1.5));
	PG_HIP(ctx, upload(f.kd_bmax, h.kd_bmax, 1.5));
	PG_HIP(ctx, upload(f.kd_vcount, h.kd_vcount, 1.5));
	PG_HIP(ctx, upload(f.rec, h.rec, 1.25));
	PG_HIP(ctx, upload(f.head, h.head, 1.5));
	PG_HIP(ctx, upload(f.tree_thr, h.thr, 1.5));
	f.n_kd = (uint32_t)h.kd.size();
	f.n_rec = (uint32_t)h.rec.size();
	f.n_trees = (uint32_t)h.head.size();
	f.level_off = h.level_off;
	int rc = zero_accumulators(ctx, nullptr);
	if (rc != PG_OK) return rc;
	rc = rebuild_jump(ctx, nullptr);
	if (rc != PG_OK) return rc;
	PG_HIP(ctx, hipDeviceSynchronize());
	return PG_OK;
}

} // namespace pg

using namespace pg;

extern "C" {

int pg_abi_version(void) { return PGSD_ABI_VERSION; }

const char *pg_last_error(const pg_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int pg_create(pg_context **out, int device_ordinal)
{
	if (!out) return fail(nullptr, PG_ERR_INVALID, "pg_create: out is NULL");
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		(void)hipGetLastError();
		return fail(nullptr, PG_ERR_NO_DEVICE,
		            std::string("pg_create: no HIP device visible (this library has no CPU fallback): ") +
		                (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
	}
	if (device_ordinal < 0 || device_ordinal >= count)
		return fail(nullptr, PG_ERR_INVALID, "pg_create: device ordinal out of range");
	e = hipSetDevice(device_ordinal);
	if (e != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device_ordinal);
	if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(nullptr, PG_ERR_NO_DEVICE,
		            std::string("pg_create: device is ") + prop.gcnArchName + ", this build targets gfx950 only");
	pg_context *ctx = new (std::nothrow) pg_context();
	if (!ctx) return fail(nullptr, PG_ERR_NOMEM, "pg_create: out of host memory");
	ctx->device = device_ordinal;
	if (prop.multiProcessorCount > 0) ctx->n_cus = prop.multiProcessorCount;
	if (const char *b = getenv("PGSD_JUMP_TABLE_MAX_BYTES")) { // memory budget of the quadtree jump tables (pg_refine.hip: rebuild_jump)
		char *end = nullptr;
		const unsigned long long v = strtoull(b, &end, 10);
		if (end != b) ctx->jump_budget = v;
	}
	e = hipMalloc((void **)&ctx->dc, sizeof(DepthCounters));
	if (e != hipSuccess) { delete ctx; return hip_fail(nullptr, e, "hipMalloc(depth counters)"); }
	(void)hipMemset(ctx->dc, 0, sizeof(DepthCounters));
	*out = ctx;
	return PG_OK;
}

int pg_destroy(pg_context *ctx)
{
	if (!ctx) return PG_OK;
	(void)hipSetDevice(ctx->device);
	if (ctx->dc) (void)hipFree(ctx->dc);
	destroy_render_state(ctx);
	destroy_comm(ctx);
	delete ctx;
	return PG_OK;
}

int pg_setup(pg_context *ctx, const float bbox_min[3], const float bbox_max[3], uint64_t num_rays,
             int32_t max_depth, int32_t kd_max_depth, int32_t quad_max_depth, int32_t store_nee,
             float bsdf_fraction)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!bbox_min || !bbox_max) return fail(ctx, PG_ERR_INVALID, "pg_setup: NULL bbox");
	if (kd_max_depth < 0 || kd_max_depth > kMaxLevels - 2 || quad_max_depth < 0 || quad_max_depth > kMaxLevels - 2)
		return fail(ctx, PG_ERR_INVALID, "pg_setup: tree depth limits must be in [0,30]");
	for (int a = 0; a < 3; ++a)
		if (!(bbox_min[a] <= bbox_max[a])) return fail(ctx, PG_ERR_INVALID, "pg_setup: bbox_min must be <= bbox_max (and not NaN)");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	for (int a = 0; a < 3; ++a) { ctx->bmin[a] = bbox_min[a]; ctx->bmax[a] = bbox_max[a]; }
	ctx->num_rays = num_rays;
	ctx->max_depth = max_depth;
	ctx->kd_max_depth = kd_max_depth;
	ctx->quad_max_depth = quad_max_depth;
	ctx->store_nee = store_nee ? 1 : 0;
	ctx->bsdf_fraction = bsdf_fraction;
	ctx->iteration = 0;
	ctx->is_final = 0;
	ctx->kd_max_leaf_size = 1.0; // KDTree.__init__ default (kdtree.py:118)
	// kdtree.py:122-124, quadtree.py:355-359: one leaf KD node owning one leaf quadtree
	HostForest h;
	KdNode n0 = {0u, 0.0f, 0u, 0u};
	h.kd.push_back(n0);
	for (int a = 0; a < 3; ++a) { h.kd_bmin.push_back(bbox_min[a]); h.kd_bmax.push_back(bbox_max[a]); }
	h.kd_vcount.push_back(0.0f);
	TreeHead th = {kNoRecord, 0.0f};
	h.head.push_back(th);
	h.thr.push_back(std::numeric_limits<float>::infinity());
	h.level_off.assign(1, 0u);
	int rc = install(ctx, h);
	if (rc == PG_OK) ctx->configured = true;
	return rc;
}

int pg_set_iteration(pg_context *ctx, int32_t iteration, int32_t is_final)
{
	if (!ctx) return PG_ERR_INVALID;
	ctx->iteration = iteration;
	ctx->is_final = is_final ? 1 : 0;
	return PG_OK;
}

#define PG_READY(ctx)                                                                  \
	do {                                                                               \
		if (!(ctx)) return PG_ERR_INVALID;                                             \
		if (!(ctx)->configured) return fail((ctx), PG_ERR_INVALID, "call pg_setup or pg_import first"); \
		PG_HIP((ctx), hipSetDevice((ctx)->device));                                    \
	} while (0)

#define PG_LAUNCHED(ctx)                                                               \
	do {                                                                               \
		hipError_t e__ = hipGetLastError();                                            \
		if (e__ != hipSuccess) return hip_fail((ctx), e__, "kernel launch");          \
	} while (0)

int pg_get_leaf_node_index(pg_context *ctx, uint64_t n, const float *p, const uint8_t *active,
                           uint32_t *node_out, void *stream)
{
	PG_READY(ctx);
	if (n && (!p || !node_out)) return fail(ctx, PG_ERR_INVALID, "pg_get_leaf_node_index: NULL pointer");
	launch_leaf_index(ctx->view(), n, p, active, node_out, (hipStream_t)stream);
	PG_LAUNCHED(ctx);
	return PG_OK;
}

int pg_sample(pg_context *ctx, uint64_t n, const float *p, uint64_t *rng_state, const uint64_t *rng_inc,
              const uint8_t *active, float *dir_out, float *pdf_out, void *stream)
{
	PG_READY(ctx);
	if (n && (!p || !rng_state || !rng_inc || !dir_out || !pdf_out))
		return fail(ctx, PG_ERR_INVALID, "pg_sample: NULL pointer");
	launch_sample(ctx->view(), n, p, rng_state, rng_inc, active, dir_out, pdf_out,
	              ctx->dc_on ? ctx->dc : nullptr, (hipStream_t)stream);
	PG_LAUNCHED(ctx);
	return PG_OK;
}

int pg_pdf(pg_context *ctx, uint64_t n, const float *p, const float *dir, const uint8_t *active,
           float *pdf_out, void *stream)
{
	PG_READY(ctx);
	if (n && (!p || !dir || !pdf_out)) return fail(ctx, PG_ERR_INVALID, "pg_pdf: NULL pointer");
	launch_pdf(ctx->view(), n, p, dir, active, pdf_out, ctx->dc_on ? ctx->dc : nullptr, (hipStream_t)stream);
	PG_LAUNCHED(ctx);
	return PG_OK;
}

int pg_guide_bounce(pg_context *ctx, uint64_t n, const float *p, const float *dir_nee,
                    const uint8_t *nee_active, const uint8_t *select, float *dir_io, uint64_t *rng_state,
                    const uint64_t *rng_inc, float *pdf_nee_out, float *pdf_out, const uint32_t *lane_index,
                    const uint32_t *d_lane_count, void *stream)
{
	PG_READY(ctx);
	if (n && (!p || !dir_nee || !dir_io || !rng_state || !rng_inc || !pdf_nee_out || !pdf_out))
		return fail(ctx, PG_ERR_INVALID, "pg_guide_bounce: NULL pointer");
	if ((lane_index == nullptr) != (d_lane_count == nullptr))
		return fail(ctx, PG_ERR_INVALID, "pg_guide_bounce: lane_index and d_lane_count go together");
	if (n > 0xffffffffull) return fail(ctx, PG_ERR_INVALID, "pg_guide_bounce: more than 2^32 lanes");
	launch_guide_bounce(ctx->view(), n, p, dir_nee, nee_active, select, dir_io, rng_state, rng_inc,
	                    pdf_nee_out, pdf_out, lane_index, d_lane_count, ctx->dc_on ? ctx->dc : nullptr,
	                    (hipStream_t)stream);
	PG_LAUNCHED(ctx);
	return PG_OK;
}

int pg_compact_lanes(pg_context *ctx, uint64_t n, const uint8_t *select, const uint8_t *nee_active,
                     uint32_t *idx_out, uint32_t *d_count, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	PG_HIP(ctx, hipSetDevice(ctx->device));
	if (!d_count || (n && (!select || !idx_out))) return fail(ctx, PG_ERR_INVALID, "pg_compact_lanes: NULL pointer");
	if (n > 0xffffffffull) return fail(ctx, PG_ERR_INVALID, "pg_compact_lanes: more than 2^32 lanes");
	if (((uintptr_t)select & 15u) || ((uintptr_t)nee_active & 15u))
		return fail(ctx, PG_ERR_INVALID, "pg_compact_lanes: masks must be 16-byte aligned");
	launch_compact_lanes(n, select, nee_active, idx_out, d_count, (hipStream_t)stream);
	PG_LAUNCHED(ctx);
	return PG_OK;
}

int pg_rng_seed(pg_context *ctx, uint64_t n, uint32_t seed, uint32_t lane0, uint64_t *rng_state,
                uint64_t *rng_inc, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	PG_HIP(ctx, hipSetDevice(ctx->device));
	if (n && (!rng_state || !rng_inc)) return fail(ctx, PG_ERR_INVALID, "pg_rng_seed: NULL pointer");
	launch_rng_seed(n, seed, lane0, rng_state, rng_inc, (hipStream_t)stream);
	PG_LAUNCHED(ctx);
	return PG_OK;
}

int pg_splat(pg_context *ctx, uint64_t m, const pg_records *rec, const uint32_t *d_count, void *stream)
{
	PG_READY(ctx);
	if (!rec) return fail(ctx, PG_ERR_INVALID, "pg_splat: NULL records");
	if (m && (!rec->position || !rec->direction || !rec->radiance || !rec->wo_pdf ||
	          (ctx->store_nee && (!rec->direction_nee || !rec->radiance_nee_lum))))
		return fail(ctx, PG_ERR_INVALID, "pg_splat: NULL record column");
	if (m > 0xffffffffull) return fail(ctx, PG_ERR_INVALID, "pg_splat: more than 2^32 records in one call");
	launch_splat(ctx->view(), ctx->f.accum_view(), ctx->store_nee, m, *rec, d_count,
	             ctx->dc_on ? ctx->dc : nullptr, (hipStream_t)stream);
	PG_LAUNCHED(ctx);
	return PG_OK;
}

static bool dense_ok(const pg_dense_records *r)
{
	return r && r->active && r->position && r->direction && r->bsdf && r->throughput_bsdf &&
	       r->throughput_radiance && r->radiance_nee && r->direction_nee && r->wo_pdf;
}

int pg_process_records(pg_context *ctx, uint64_t num_rays, int32_t max_depth, const float *l_final,
                       const pg_dense_records *rec, const pg_records_out *out, uint32_t *d_count,
                       void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	PG_HIP(ctx, hipSetDevice(ctx->device));
	if (max_depth <= 0) return fail(ctx, PG_ERR_INVALID, "pg_process_records: max_depth must be > 0");
	if (!dense_ok(rec) || !l_final || !out || !d_count || !out->position || !out->direction ||
	    !out->radiance || !out->wo_pdf || !out->direction_nee || !out->radiance_nee_lum)
		return fail(ctx, PG_ERR_INVALID, "pg_process_records: NULL pointer");
	if (num_rays * (uint64_t)max_depth > 0xffffffffull)
		return fail(ctx, PG_ERR_INVALID, "pg_process_records: more than 2^32 slots");
	launch_process_records(num_rays, max_depth, l_final, *rec, *out, d_count, (hipStream_t)stream);
	PG_LAUNCHED(ctx);
	return PG_OK;
}

int pg_process_and_splat(pg_context *ctx, uint64_t num_rays, int32_t max_depth, const float *l_final,
                         const pg_dense_records *rec, void *stream)
{
	PG_READY(ctx);
	if (max_depth <= 0) return fail(ctx, PG_ERR_INVALID, "pg_process_and_splat: max_depth must be > 0");
	if (!dense_ok(rec) || !l_final) return fail(ctx, PG_ERR_INVALID, "pg_process_and_splat: NULL pointer");
	launch_process_and_splat(ctx->view(), ctx->f.accum_view(), ctx->store_nee, num_rays, max_depth, l_final,
	                         *rec, ctx->dc_on ? ctx->dc : nullptr, (hipStream_t)stream);
	PG_LAUNCHED(ctx);
	return PG_OK;
}

int pg_refine_and_swap(pg_context *ctx, void *stream)
{
	PG_READY(ctx);
	// (a transaction, pg_refine.hip: on any error sdTree_prev and sdTree_current are exactly what they were)
	return refine_and_swap(ctx, (hipStream_t)stream);
}

int pg_debug_fail_alloc(int64_t successes_before_failure)
{
	g_alloc_fail_countdown = successes_before_failure < 0 ? -1 : (long long)successes_before_failure;
	return PG_OK;
}

int pg_debug_fail_alloc_pending(void) { return g_alloc_fail_countdown >= 0 ? 1 : 0; }

int pg_accumulators(pg_context *ctx, int64_t **d_buffer, uint64_t *count)
{
	PG_READY(ctx);
	if (!d_buffer || !count) return fail(ctx, PG_ERR_INVALID, "pg_accumulators: NULL pointer");
	*d_buffer = reinterpret_cast<int64_t *>(ctx->f.acc.p);
	*count = ctx->f.acc_count();
	return PG_OK;
}

int pg_sort_places(pg_context *ctx, uint64_t n, const uint16_t *d_keys, const uint32_t *d_live, uint32_t *d_places_out, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	if (n == 0) return PG_OK;
	if (!d_keys || !d_places_out) return fail(ctx, PG_ERR_INVALID, "pg_sort_places: NULL pointer");
	if (n >= (1ull << 28)) return fail(ctx, PG_ERR_INVALID, "pg_sort_places: at most 2^28 - 1 places");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t s = (hipStream_t)stream;
	DevBuf<char> tmp;
	DevBuf<uint16_t> keys_mid;
	const size_t bytes = sort_pairs_temp_bytes((uint32_t)n);
	PG_HIP(ctx, tmp.ensure(bytes));
	PG_HIP(ctx, keys_mid.ensure((size_t)n));
	PG_HIP(ctx, sort_places16(tmp.p, bytes, d_keys, keys_mid.p, d_places_out, (uint32_t)n, d_live, s));
	PG_HIP(ctx, hipStreamSynchronize(s)); // (the scratch buffers go out of scope)
	return PG_OK;
}

int pg_enable_depth_counters(pg_context *ctx, int32_t on)
{
	if (!ctx) return PG_ERR_INVALID;
	ctx->dc_on = on == 1;
	ctx->ph_on = false;
	if ((on == 1 || on == 2) && shade_phases_compiled_in()) { // (2: a probe build's phase stamps alone, without the counters' atomics)
		PG_HIP(ctx, hipSetDevice(ctx->device));
		if (!ctx->ph_buf.p) {
			PG_HIP(ctx, ctx->ph_buf.ensure((size_t)kPhaseStripes * kPhaseWords));
			PG_HIP(ctx, hipMemset(ctx->ph_buf.p, 0, (size_t)kPhaseStripes * kPhaseWords * sizeof(unsigned long long)));
		}
		ctx->ph_on = true;
	}
	return PG_OK;
}

int pg_read_depth_counters(pg_context *ctx, pg_depth_counters *out, int32_t reset)
{
	if (!ctx || !out) return PG_ERR_INVALID;
	PG_HIP(ctx, hipSetDevice(ctx->device));
	PG_HIP(ctx, hipDeviceSynchronize());
	DepthCounters h;
	PG_HIP(ctx, hipMemcpy(&h, ctx->dc, sizeof(h), hipMemcpyDeviceToHost));
	out->kd_levels = h.kd_levels;
	out->kd_queries = h.kd_queries;
	out->quad_levels = h.quad_levels;
	out->quad_queries = h.quad_queries;
	out->layout_bytes = h.layout_bytes;
	if (getenv("PGSD_TRACE_SHADOW") && h.body_waves)
		fprintf(stderr, "[pgsd] k_wave_shade: %llu waves ran the body, %llu of them walked shadow rays with %llu lanes (%.1f of 64 per walking wave)\n",
		        h.body_waves, h.shadow_waves, h.shadow_lanes, h.shadow_waves ? (double)h.shadow_lanes / (double)h.shadow_waves : 0.0);
	if (reset) PG_HIP(ctx, hipMemset(ctx->dc, 0, sizeof(DepthCounters)));
	return PG_OK;
}

int pg_read_shade_phases(pg_context *ctx, uint64_t *h_out, int32_t reset)
{
	if (!ctx || !h_out) return PG_ERR_INVALID;
	for (int i = 0; i < 10; ++i) h_out[i] = 0;
	h_out[0] = shade_phases_compiled_in() ? 1u : 0u;
	if (!ctx->ph_buf.p) return PG_OK; // (the product build, or never enabled)
	PG_HIP(ctx, hipSetDevice(ctx->device));
	PG_HIP(ctx, hipDeviceSynchronize());
	std::vector<unsigned long long> h((size_t)kPhaseStripes * kPhaseWords);
	PG_HIP(ctx, hipMemcpy(h.data(), ctx->ph_buf.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	for (int st = 0; st < kPhaseStripes; ++st) {
		h_out[1] += h[(size_t)st * kPhaseWords + 7];
		for (int i = 0; i < 7; ++i) h_out[2 + i] += h[(size_t)st * kPhaseWords + i];
	}
	if (getenv("PGSD_TRACE_SHADOW") && h_out[1]) {
		unsigned long long tot = 0;
		for (int i = 0; i < 7; ++i) tot += h_out[2 + i];
		static const char *names[7] = {"records + staging", "stage_a1", "shadow walk", "stage_a2", "SD-tree calls", "stage_b", "append"};
		fprintf(stderr, "[pgsd] k_wave_shade: a wave's %.0f cycles on average, by phase:", (double)tot / (double)h_out[1]);
		for (int i = 0; i < 7; ++i) fprintf(stderr, " %s %.1f %%%s", names[i], 100.0 * (double)h_out[2 + i] / (double)(tot ? tot : 1), i < 6 ? "," : "\n");
	}
	if (reset) PG_HIP(ctx, hipMemset(ctx->ph_buf.p, 0, h.size() * sizeof(unsigned long long)));
	return PG_OK;
}

// ---------------------------------------------------------------------------------------------
// import / export (reference column schema, canonical layout of SURVEY Appendix A8)
// ---------------------------------------------------------------------------------------------

// canonical node counts per level implied by the record levels
static void canonical_bases(const Forest &f, std::vector<uint64_t> &node_base, uint64_t &n_nodes)
{
	const size_t L = f.level_off.size() - 1; // number of record levels
	node_base.assign(L + 2, 0);
	node_base[0] = 0;
	node_base[1] = f.n_trees;
	for (size_t l = 0; l < L; ++l)
		node_base[l + 2] = node_base[l + 1] + 4ull * (f.level_off[l + 1] - f.level_off[l]);
	n_nodes = node_base[L + 1];
}

int pg_export_sizes(pg_context *ctx, pg_tree_sizes *sizes)
{
	PG_READY(ctx);
	if (!sizes) return fail(ctx, PG_ERR_INVALID, "pg_export_sizes: NULL");
	std::vector<uint64_t> nb;
	uint64_t n_nodes;
	canonical_bases(ctx->f, nb, n_nodes);
	sizes->n_kd = ctx->f.n_kd;
	sizes->n_quad = n_nodes;
	sizes->n_roots = ctx->f.n_trees;
	return PG_OK;
}

int pg_export(pg_context *ctx, const pg_tree_sizes *sizes, pg_tree_columns *o)
{
	PG_READY(ctx);
	if (!sizes || !o) return fail(ctx, PG_ERR_INVALID, "pg_export: NULL");
	Forest &f = ctx->f;
	std::vector<uint64_t> nb;
	uint64_t n_nodes;
	canonical_bases(f, nb, n_nodes);
	if (sizes->n_kd != f.n_kd || sizes->n_quad != n_nodes || sizes->n_roots != f.n_trees)
		return fail(ctx, PG_ERR_INVALID, "pg_export: sizes do not match pg_export_sizes");
	PG_HIP(ctx, hipDeviceSynchronize());
	std::vector<KdNode> kd;
	std::vector<float> vc;
	PG_HIP(ctx, download(kd, f.kd, f.n_kd));
	PG_HIP(ctx, download(vc, f.kd_vcount, f.n_kd));
	PG_HIP(ctx, hipMemcpy(o->kd_bbox_min, f.kd_bmin.p, (size_t)f.n_kd * 3 * sizeof(float), hipMemcpyDeviceToHost));
	PG_HIP(ctx, hipMemcpy(o->kd_bbox_max, f.kd_bmax.p, (size_t)f.n_kd * 3 * sizeof(float), hipMemcpyDeviceToHost));
	o->kd_max_leaf_size = ctx->kd_max_leaf_size;
	o->kd_max_depth = ctx->kd_max_depth;
	o->quad_max_depth = ctx->quad_max_depth;
	o->quad_store_nee = ctx->store_nee;
	for (uint32_t i = 0; i < f.n_kd; ++i) {
		o->kd_depth[i] = kd[i].axis_depth >> 2;
		o->kd_vert_count[i] = vc[i];
		o->kd_is_leaf[i] = kd[i].child == 0;
		o->kd_quad_root_index[i] = kd[i].tree;
		o->kd_child_left[i] = kd[i].child;                      // new nodes default to 0 (kdtree.py:89-90)
		o->kd_child_right[i] = kd[i].child ? kd[i].child + 1 : 0;
	}
	std::vector<QuadRec> rec;
	std::vector<TreeHead> head;
	std::vector<float> thr;
	PG_HIP(ctx, download(rec, f.rec, f.n_rec));
	PG_HIP(ctx, download(head, f.head, f.n_trees));
	PG_HIP(ctx, download(thr, f.tree_thr, f.n_trees));
	// roots
	std::vector<uint64_t> node_of_rec(f.n_rec);
	std::vector<uint32_t> tree_of_rec(f.n_rec);
	for (uint32_t t = 0; t < f.n_trees; ++t) {
		o->quad_root_node_index[t] = t;
		o->quad_bbox_min[2 * t] = o->quad_bbox_min[2 * t + 1] = 0.0f;
		o->quad_bbox_max[2 * t] = o->quad_bbox_max[2 * t + 1] = 1.0f;
		o->quad_depth[t] = 0;
		o->quad_irradiance[t] = head[t].root_irr;
		o->quad_threshold[t] = thr[t];
		const bool leaf = head[t].root_rec == kNoRecord;
		o->quad_is_leaf[t] = leaf;
		for (int j = 0; j < 4; ++j) o->quad_child[j][t] = 0;
		if (!leaf) { node_of_rec[head[t].root_rec] = t; tree_of_rec[head[t].root_rec] = t; }
	}
	const size_t L = f.level_off.size() - 1;
	for (size_t l = 0; l < L; ++l) {
		for (uint32_t r = f.level_off[l]; r < f.level_off[l + 1]; ++r) {
			const uint64_t pn = node_of_rec[r];
			const uint32_t tree = tree_of_rec[r];
			const float mnx = o->quad_bbox_min[2 * pn], mny = o->quad_bbox_min[2 * pn + 1];
			const float mxx = o->quad_bbox_max[2 * pn], mxy = o->quad_bbox_max[2 * pn + 1];
			const float mdx = (mnx + mxx) / 2.0f, mdy = (mny + mxy) / 2.0f;
			const float cmin[4][2] = {{mdx, mdy}, {mnx, mdy}, {mnx, mny}, {mdx, mny}};
			const float cmax[4][2] = {{mxx, mxy}, {mdx, mxy}, {mdx, mdy}, {mxx, mdy}};
			for (int j = 0; j < 4; ++j) {
				const uint64_t cn = nb[l + 1] + 4ull * (r - f.level_off[l]) + j;
				o->quad_child[j][pn] = (uint32_t)cn;
				o->quad_bbox_min[2 * cn] = cmin[j][0]; o->quad_bbox_min[2 * cn + 1] = cmin[j][1];
				o->quad_bbox_max[2 * cn] = cmax[j][0]; o->quad_bbox_max[2 * cn + 1] = cmax[j][1];
				o->quad_depth[cn] = (uint32_t)l + 1;
				o->quad_irradiance[cn] = rec[r].irr[j];
				o->quad_threshold[cn] = thr[tree];
				o->quad_is_leaf[cn] = rec[r].child[j] == 0;
				for (int k = 0; k < 4; ++k) o->quad_child[k][cn] = 0;
				if (rec[r].child[j]) { node_of_rec[rec[r].child[j]] = cn; tree_of_rec[rec[r].child[j]] = tree; }
			}
		}
	}
	return PG_OK;
}

int pg_export_accumulators(pg_context *ctx, const pg_tree_sizes *sizes, uint64_t *kd_count,
                           uint64_t *acc_lo, int64_t *acc_hi)
{
	PG_READY(ctx);
	if (!sizes || !kd_count || !acc_lo || !acc_hi) return fail(ctx, PG_ERR_INVALID, "pg_export_accumulators: NULL");
	Forest &f = ctx->f;
	std::vector<uint64_t> nb;
	uint64_t n_nodes;
	canonical_bases(f, nb, n_nodes);
	if (sizes->n_kd != f.n_kd || sizes->n_quad != n_nodes || sizes->n_roots != f.n_trees)
		return fail(ctx, PG_ERR_INVALID, "pg_export_accumulators: sizes do not match");
	PG_HIP(ctx, hipDeviceSynchronize());
	std::vector<long long> acc;
	PG_HIP(ctx, download(acc, f.acc, f.acc_count()));
	std::vector<KdNode> kd;
	std::vector<QuadRec> rec;
	std::vector<TreeHead> head;
	PG_HIP(ctx, download(kd, f.kd, f.n_kd));
	PG_HIP(ctx, download(rec, f.rec, f.n_rec));
	PG_HIP(ctx, download(head, f.head, f.n_trees));
	const long long *rec_acc = acc.data();
	const long long *root_acc = acc.data() + (size_t)f.n_rec * 4 * kAccWords;
	const unsigned long long *leaf_count =
	    reinterpret_cast<const unsigned long long *>(root_acc + (size_t)f.n_trees * kAccWords);
	// quadtree: per-record totals bottom-up (child records have larger indices than the parent);
	// word 3 of every accumulator counts the records whose path direction ended there
	std::vector<I128> tot(f.n_rec);
	std::vector<uint64_t> cnt_tot(f.n_rec);
	std::vector<I128> slot((size_t)f.n_rec * 4);
	for (int64_t r = (int64_t)f.n_rec - 1; r >= 0; --r) {
		I128 s = {0, 0};
		uint64_t c = 0;
		for (int j = 0; j < 4; ++j) {
			I128 v;
			if (rec[r].child[j]) { v = tot[rec[r].child[j]]; c += cnt_tot[rec[r].child[j]]; }
			else {
				const long long *l = rec_acc + ((size_t)r * 4 + j) * kAccWords;
				v = limbs_resolve(l[0], l[1], l[2]);
				c += (uint64_t)l[3];
			}
			slot[(size_t)r * 4 + j] = v;
			s = i128_add(s, v);
		}
		tot[r] = s;
		cnt_tot[r] = c;
	}
	std::vector<uint64_t> tree_count(f.n_trees);
	for (uint32_t t = 0; t < f.n_trees; ++t) {
		I128 v;
		const long long *ra = root_acc + (size_t)t * kAccWords;
		if (head[t].root_rec == kNoRecord) { v = limbs_resolve(ra[0], ra[1], ra[2]); tree_count[t] = (uint64_t)ra[3]; }
		else { v = tot[head[t].root_rec]; tree_count[t] = cnt_tot[head[t].root_rec]; }
		tree_count[t] += leaf_count[t];
		acc_lo[t] = v.lo;
		acc_hi[t] = v.hi;
	}
	// KD: leaves take their tree's count, inner nodes the sum of their children (children have
	// larger indices than their parent: kdtree.py:243-245)
	for (int64_t i = (int64_t)f.n_kd - 1; i >= 0; --i)
		kd_count[i] = kd[i].child == 0 ? tree_count[kd[i].tree] : kd_count[kd[i].child] + kd_count[kd[i].child + 1];
	const size_t L = f.level_off.size() - 1;
	for (size_t l = 0; l < L; ++l)
		for (uint32_t r = f.level_off[l]; r < f.level_off[l + 1]; ++r)
			for (int j = 0; j < 4; ++j) {
				const uint64_t cn = nb[l + 1] + 4ull * (r - f.level_off[l]) + j;
				acc_lo[cn] = slot[(size_t)r * 4 + j].lo;
				acc_hi[cn] = slot[(size_t)r * 4 + j].hi;
			}
	return PG_OK;
}

int pg_import(pg_context *ctx, const pg_tree_sizes *sizes, const pg_tree_columns *c)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!sizes || !c) return fail(ctx, PG_ERR_INVALID, "pg_import: NULL");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	const uint64_t nk = sizes->n_kd, nq = sizes->n_quad, nr = sizes->n_roots;
	if (nk == 0 || nq == 0 || nr == 0) return fail(ctx, PG_ERR_FORMAT, "pg_import: empty tree");
	if (nk > 0x7fffffffull || nq > 0xffffffffull) return fail(ctx, PG_ERR_FORMAT, "pg_import: tree too large");
	if (c->kd_max_depth < 0 || c->kd_max_depth > kMaxLevels - 2 || c->quad_max_depth < 0 ||
	    c->quad_max_depth > kMaxLevels - 2)
		return fail(ctx, PG_ERR_FORMAT, "pg_import: depth limits must be in [0,30]");
	HostForest h;
	h.kd.resize(nk);
	h.kd_bmin.assign(c->kd_bbox_min, c->kd_bbox_min + nk * 3);
	h.kd_bmax.assign(c->kd_bbox_max, c->kd_bbox_max + nk * 3);
	h.kd_vcount.assign(c->kd_vert_count, c->kd_vert_count + nk);
	for (uint64_t i = 0; i < nk; ++i) {
		KdNode n;
		const uint32_t depth = c->kd_depth[i];
		n.axis_depth = (depth % 3u) | (depth << 2);
		n.tree = c->kd_quad_root_index[i];
		if (n.tree >= nr) return fail(ctx, PG_ERR_FORMAT, "pg_import: kd quadTreeRootIndex out of range");
		if (c->kd_is_leaf[i]) { n.child = 0; n.split = 0.0f; }
		else {
			const uint32_t l = c->kd_child_left[i], r = c->kd_child_right[i];
			if (r != l + 1 || l <= i || r >= nk)
				return fail(ctx, PG_ERR_FORMAT, "pg_import: kd children must be adjacent and follow their parent");
			if (depth >= (uint32_t)kMaxLevels - 1) return fail(ctx, PG_ERR_FORMAT, "pg_import: kd tree too deep");
			const uint32_t axis = depth % 3u;
			n.child = l;
			n.split = c->kd_bbox_max[3 * (uint64_t)l + axis];
			if (!(n.split == c->kd_bbox_min[3 * (uint64_t)r + axis]))
				return fail(ctx, PG_ERR_FORMAT, "pg_import: kd children do not meet at a split plane");
		}
		h.kd[i] = n;
	}
	// quadtree forest: BFS in canonical order assigns record ids level by level
	h.head.resize(nr);
	h.thr.resize(nr);
	std::vector<uint32_t> cur, nxt; // source node index of each record of the current level
	for (uint64_t t = 0; t < nr; ++t) {
		const uint32_t node = c->quad_root_node_index[t];
		if (node >= nq) return fail(ctx, PG_ERR_FORMAT, "pg_import: rootNodeIndex out of range");
		if (!(c->quad_bbox_min[2 * (uint64_t)node] == 0.0f && c->quad_bbox_min[2 * (uint64_t)node + 1] == 0.0f &&
		      c->quad_bbox_max[2 * (uint64_t)node] == 1.0f && c->quad_bbox_max[2 * (uint64_t)node + 1] == 1.0f))
			return fail(ctx, PG_ERR_FORMAT, "pg_import: quadtree root is not the unit square");
		h.thr[t] = c->quad_threshold[node];
		h.head[t].root_irr = c->quad_irradiance[node];
		if (c->quad_is_leaf[node]) h.head[t].root_rec = kNoRecord;
		else { h.head[t].root_rec = (uint32_t)cur.size(); cur.push_back(node); }
	}
	h.level_off.assign(1, 0u);
	uint64_t next_id = cur.size();
	while (!cur.empty()) {
		if (h.level_off.size() >= (size_t)kMaxLevels - 1) return fail(ctx, PG_ERR_FORMAT, "pg_import: quadtree too deep");
		nxt.clear();
		for (uint32_t node : cur) {
			QuadRec q;
			const float mnx = c->quad_bbox_min[2 * (uint64_t)node], mny = c->quad_bbox_min[2 * (uint64_t)node + 1];
			const float mxx = c->quad_bbox_max[2 * (uint64_t)node], mxy = c->quad_bbox_max[2 * (uint64_t)node + 1];
			const float mdx = (mnx + mxx) / 2.0f, mdy = (mny + mxy) / 2.0f;
			const float cmin[4][2] = {{mdx, mdy}, {mnx, mdy}, {mnx, mny}, {mdx, mny}};
			const float cmax[4][2] = {{mxx, mxy}, {mdx, mxy}, {mdx, mdy}, {mxx, mdy}};
			for (int j = 0; j < 4; ++j) {
				const uint32_t ch = c->quad_child[j][node];
				if (ch >= nq) return fail(ctx, PG_ERR_FORMAT, "pg_import: quadtree child index out of range");
				if (!(c->quad_bbox_min[2 * (uint64_t)ch] == cmin[j][0] && c->quad_bbox_min[2 * (uint64_t)ch + 1] == cmin[j][1] &&
				      c->quad_bbox_max[2 * (uint64_t)ch] == cmax[j][0] && c->quad_bbox_max[2 * (uint64_t)ch + 1] == cmax[j][1]))
					return fail(ctx, PG_ERR_FORMAT, "pg_import: quadtree child is not the midpoint quadrant of its parent");
				q.irr[j] = c->quad_irradiance[ch];
				if (c->quad_is_leaf[ch]) q.child[j] = 0;
				else {
					if (next_id >= 0xfffffff0ull) return fail(ctx, PG_ERR_FORMAT, "pg_import: too many quadtree nodes");
					q.child[j] = (uint32_t)next_id++;
					nxt.push_back(ch);
				}
			}
			h.rec.push_back(q);
		}
		h.level_off.push_back((uint32_t)h.rec.size());
		cur.swap(nxt);
	}
	for (int a = 0; a < 3; ++a) { ctx->bmin[a] = c->kd_bbox_min[a]; ctx->bmax[a] = c->kd_bbox_max[a]; }
	ctx->kd_max_leaf_size = c->kd_max_leaf_size;
	ctx->kd_max_depth = c->kd_max_depth;
	ctx->quad_max_depth = c->quad_max_depth;
	ctx->store_nee = c->quad_store_nee ? 1 : 0;
	int rc = install(ctx, h);
	if (rc == PG_OK) ctx->configured = true;
	return rc;
}

int pg_get_stats(pg_context *ctx, pg_stats *out)
{
	PG_READY(ctx);
	if (!out) return fail(ctx, PG_ERR_INVALID, "pg_get_stats: NULL");
	Forest &f = ctx->f;
	PG_HIP(ctx, hipDeviceSynchronize());
	std::vector<KdNode> kd;
	std::vector<QuadRec> rec;
	std::vector<TreeHead> head;
	PG_HIP(ctx, download(kd, f.kd, f.n_kd));
	PG_HIP(ctx, download(rec, f.rec, f.n_rec));
	PG_HIP(ctx, download(head, f.head, f.n_trees));
	memset(out, 0, sizeof(*out));
	out->n_kd_nodes = f.n_kd;
	out->n_quad_records = f.n_rec;
	out->n_trees = f.n_trees;
	double kd_sum = 0;
	for (uint32_t i = 0; i < f.n_kd; ++i)
		if (kd[i].child == 0) {
			++out->n_kd_leaves;
			const uint32_t d = kd[i].axis_depth >> 2;
			kd_sum += d;
			if (d > out->max_kd_depth) out->max_kd_depth = d;
		}
	out->mean_kd_leaf_depth = out->n_kd_leaves ? kd_sum / (double)out->n_kd_leaves : 0.0;
	double q_sum = 0;
	uint64_t q_leaves = 0;
	for (uint32_t t = 0; t < f.n_trees; ++t)
		if (head[t].root_rec == kNoRecord) ++q_leaves;
	const size_t L = f.level_off.size() - 1;
	for (size_t l = 0; l < L; ++l)
		for (uint32_t r = f.level_off[l]; r < f.level_off[l + 1]; ++r)
			for (int j = 0; j < 4; ++j)
				if (rec[r].child[j] == 0) {
					++q_leaves;
					q_sum += (double)(l + 1);
					if (l + 1 > out->max_quad_depth) out->max_quad_depth = (uint32_t)l + 1;
				}
	out->mean_quad_leaf_depth = q_leaves ? q_sum / (double)q_leaves : 0.0;
	out->n_quad_nodes = (uint64_t)f.n_trees + 4ull * f.n_rec;
	out->bytes_kd = (uint64_t)f.n_kd * sizeof(KdNode);
	out->bytes_quad_records = (uint64_t)f.n_rec * sizeof(QuadRec) + (uint64_t)f.n_trees * sizeof(TreeHead);
	out->bytes_accumulators = f.acc_count() * sizeof(long long);
	out->jump_bits = f.jump_valid ? (uint32_t)f.jump_bits : 0u;
	out->bytes_jump_tables = f.jump_valid ? (((uint64_t)f.n_trees * sizeof(QuadJump)) << (2 * f.jump_bits)) : 0ull;
	out->kd_grid_bits = f.kd_grid_valid ? (uint32_t)f.kd_grid_bits : 0u;
	return PG_OK;
}

} // extern "C"
