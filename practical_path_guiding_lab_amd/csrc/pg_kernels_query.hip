// pg_kernels_query.hip -- sdTree_prev queries: one lane per ray, wave64, 256-thread groups.
//
// Bound: HBM/L2 latency of dependent 16-B (KD) and 32-B (quadtree) gathers; no reuse inside a
// lane, so occupancy (registers <= 64 -> 8 waves/SIMD) is what hides the latency.
#include "pg_descent.hpp"
#include "pg_kernels.hpp"

namespace pg {

constexpr int kBlock = 256;

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}

__device__ __forceinline__ void count_depths(DepthCounters *dc, unsigned kd_lv, unsigned kd_q,
                                             unsigned q_lv, unsigned q_q)
{
	if (dc == nullptr) return; // wave-uniform
	const unsigned long long a = wave_sum(kd_lv), b = wave_sum(kd_q), c = wave_sum(q_lv), d = wave_sum(q_q);
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&dc->kd_levels, a);
		atomicAdd(&dc->kd_queries, b);
		atomicAdd(&dc->quad_levels, c);
		atomicAdd(&dc->quad_queries, d);
	}
}

__device__ __forceinline__ TreeHead load_head(const TreeHead *h, uint32_t t)
{
	const uint2 v = *reinterpret_cast<const uint2 *>(h + t);
	TreeHead r;
	r.root_rec = v.x;
	r.root_irr = __uint_as_float(v.y);
	return r;
}

__global__ __launch_bounds__(kBlock) void k_leaf_index(TreeView t, uint64_t n, const float *__restrict__ p,
                                                       const uint8_t *__restrict__ active,
                                                       uint32_t *__restrict__ node_out)
{
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const float x = p[i], y = p[n + i], z = p[2 * n + i];
	const bool act = active ? active[i] != 0 : true;
	KdNode leaf;
	uint32_t lv;
	node_out[i] = kd_descend(t.kd, x, y, z, act && inside_root(t, x, y, z), leaf, lv);
}

__global__ __launch_bounds__(kBlock) void k_sample(TreeView t, uint64_t n, const float *__restrict__ p,
                                                   uint64_t *__restrict__ rng_state,
                                                   const uint64_t *__restrict__ rng_inc,
                                                   const uint8_t *__restrict__ active,
                                                   float *__restrict__ dir_out, float *__restrict__ pdf_out,
                                                   DepthCounters *dc)
{
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	unsigned kd_lv = 0, q_lv = 0, did = 0;
	if (i < n) {
		const bool act = active ? active[i] != 0 : true;
		float dx = 0.0f, dy = 0.0f, dz = -1.0f, pdf = 1.0f; // inactive lanes (quadtree.py:940, 1011)
		if (act) {
			const float x = p[i], y = p[n + i], z = p[2 * n + i];
			KdNode leaf;
			kd_descend(t.kd, x, y, z, inside_root(t, x, y, z), leaf, kd_lv);
			Pcg32 rng = {rng_state[i], rng_inc[i]};
			quad_sample(t.rec, load_head(t.head, leaf.tree), rng, dx, dy, dz, pdf, q_lv);
			rng_state[i] = rng.state;
			did = 1;
		}
		dir_out[i] = dx;
		dir_out[n + i] = dy;
		dir_out[2 * n + i] = dz;
		pdf_out[i] = pdf;
	}
	count_depths(dc, kd_lv, did, q_lv, did);
}

__global__ __launch_bounds__(kBlock) void k_pdf(TreeView t, uint64_t n, const float *__restrict__ p,
                                                const float *__restrict__ dir,
                                                const uint8_t *__restrict__ active,
                                                float *__restrict__ pdf_out, DepthCounters *dc)
{
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	unsigned kd_lv = 0, q_lv = 0, did = 0;
	if (i < n) {
		const bool act = active ? active[i] != 0 : true;
		float pdf = 1.0f;
		if (act) {
			const float x = p[i], y = p[n + i], z = p[2 * n + i];
			KdNode leaf;
			kd_descend(t.kd, x, y, z, inside_root(t, x, y, z), leaf, kd_lv);
			float cx, cy;
			dir_to_canonical(dir[i], dir[n + i], dir[2 * n + i], cx, cy);
			pdf = quad_pdf(t.rec, load_head(t.head, leaf.tree), cx, cy, q_lv);
			did = 1;
		}
		pdf_out[i] = pdf;
	}
	count_depths(dc, kd_lv, did, q_lv, did);
}

// One KD descent shared by the NEE pdf and the sample-or-pdf of the continuation direction
// (path_guiding_integrator.py:244, 301, 307 all pass the same si.p).
__global__ __launch_bounds__(kBlock) void k_guide_bounce(TreeView t, uint64_t n, const float *__restrict__ p,
                                                         const float *__restrict__ dir_nee,
                                                         const uint8_t *__restrict__ nee_active,
                                                         const uint8_t *__restrict__ select,
                                                         float *__restrict__ dir_io,
                                                         uint64_t *__restrict__ rng_state,
                                                         const uint64_t *__restrict__ rng_inc,
                                                         float *__restrict__ pdf_nee_out,
                                                         float *__restrict__ pdf_out, DepthCounters *dc)
{
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	unsigned kd_lv = 0, kd_q = 0, q_lv = 0, q_q = 0;
	if (i < n) {
		const bool nee = nee_active ? nee_active[i] != 0 : true;
		const int sel = select ? (int)select[i] : 2;
		float pdf_nee = 1.0f, pdf = 1.0f;
		if (nee || sel != 0) {
			const float x = p[i], y = p[n + i], z = p[2 * n + i];
			KdNode leaf;
			kd_descend(t.kd, x, y, z, inside_root(t, x, y, z), leaf, kd_lv);
			kd_q = 1;
			const TreeHead head = load_head(t.head, leaf.tree);
			if (nee) {
				float cx, cy;
				uint32_t lv;
				dir_to_canonical(dir_nee[i], dir_nee[n + i], dir_nee[2 * n + i], cx, cy);
				pdf_nee = quad_pdf(t.rec, head, cx, cy, lv);
				q_lv += lv;
				++q_q;
			}
			if (sel == 2) {
				Pcg32 rng = {rng_state[i], rng_inc[i]};
				float dx, dy, dz;
				uint32_t lv;
				quad_sample(t.rec, head, rng, dx, dy, dz, pdf, lv);
				rng_state[i] = rng.state;
				dir_io[i] = dx;
				dir_io[n + i] = dy;
				dir_io[2 * n + i] = dz;
				q_lv += lv;
				++q_q;
			} else if (sel == 1) {
				float cx, cy;
				uint32_t lv;
				dir_to_canonical(dir_io[i], dir_io[n + i], dir_io[2 * n + i], cx, cy);
				pdf = quad_pdf(t.rec, head, cx, cy, lv);
				q_lv += lv;
				++q_q;
			}
		}
		pdf_nee_out[i] = pdf_nee;
		pdf_out[i] = pdf;
	}
	count_depths(dc, kd_lv, kd_q, q_lv, q_q);
}

__global__ __launch_bounds__(kBlock) void k_rng_seed(uint64_t n, uint32_t seed, uint32_t lane0,
                                                     uint64_t *__restrict__ state, uint64_t *__restrict__ inc)
{
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const Pcg32 r = pcg32_seed(seed, lane0 + (uint32_t)i);
	state[i] = r.state;
	inc[i] = r.inc;
}

static inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

void launch_leaf_index(const TreeView &t, uint64_t n, const float *p, const uint8_t *active,
                       uint32_t *node_out, hipStream_t s)
{
	if (n == 0) return;
	hipLaunchKernelGGL(k_leaf_index, grid_for(n), dim3(kBlock), 0, s, t, n, p, active, node_out);
}

void launch_sample(const TreeView &t, uint64_t n, const float *p, uint64_t *rng_state,
                   const uint64_t *rng_inc, const uint8_t *active, float *dir_out, float *pdf_out,
                   DepthCounters *dc, hipStream_t s)
{
	if (n == 0) return;
	hipLaunchKernelGGL(k_sample, grid_for(n), dim3(kBlock), 0, s, t, n, p, rng_state, rng_inc, active,
	                   dir_out, pdf_out, dc);
}

void launch_pdf(const TreeView &t, uint64_t n, const float *p, const float *dir, const uint8_t *active,
                float *pdf_out, DepthCounters *dc, hipStream_t s)
{
	if (n == 0) return;
	hipLaunchKernelGGL(k_pdf, grid_for(n), dim3(kBlock), 0, s, t, n, p, dir, active, pdf_out, dc);
}

void launch_guide_bounce(const TreeView &t, uint64_t n, const float *p, const float *dir_nee,
                         const uint8_t *nee_active, const uint8_t *select, float *dir_io,
                         uint64_t *rng_state, const uint64_t *rng_inc, float *pdf_nee_out,
                         float *pdf_out, DepthCounters *dc, hipStream_t s)
{
	if (n == 0) return;
	hipLaunchKernelGGL(k_guide_bounce, grid_for(n), dim3(kBlock), 0, s, t, n, p, dir_nee, nee_active,
	                   select, dir_io, rng_state, rng_inc, pdf_nee_out, pdf_out, dc);
}

void launch_rng_seed(uint64_t n, uint32_t seed, uint32_t lane0, uint64_t *state, uint64_t *inc,
                     hipStream_t s)
{
	if (n == 0) return;
	hipLaunchKernelGGL(k_rng_seed, grid_for(n), dim3(kBlock), 0, s, n, seed, lane0, state, inc);
}

} // namespace pg
