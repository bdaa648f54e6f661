// pg_render.hip -- wavefront renderer substrate around the SD-tree: what the reference's
// PathGuidingIntegrator.sample() (src/path_guiding_integrator.py:126-431) does per pass, with the
// Mitsuba calls it makes (scene.ray_intersect :185, emitter eval/pdf :189-198,
// sample_emitter_direction :213, bsdf.eval_pdf/sample :220, 272, 304, si.to_local/to_world/spawn_ray
// :219, 277, 352, sampler.next_1d/2d) implemented for the scene subset of scenes/cornell-box:
// quads, twosided diffuse BSDFs, one-sided area emitters, perspective camera.
//
// One kernel per bounce over all ray slots (lanes whose path has ended retire at once); the
// SD-tree queries are the same device functions the stand-alone query kernels use, so a bounce
// costs one KD descent and at most two quadtree descents per live lane and no intermediate
// wavefront buffers.  Live rays are re-compacted after every bounce (k_compact_lanes).  Path-vertex
// records go to a dense slot buffer (the reference's :318, stored depth-major here so that a
// wavefront's stores coalesce) that pg_process_and_splat consumes after the last bounce (:388-395).
//
// Arithmetic mirrors oracle/pg_oracle_render.c operation by operation (fp32, no contraction), so
// radiance, records and therefore the refined trees are bit-identical to the CPU restatement.
#include "pg_context.hpp"
#include "pg_descent.hpp"
#include "pg_kernels.hpp"

namespace pg {

constexpr int kRBlock = 256;
constexpr float kInvPiF = 0.31830988618379067154f;
constexpr float kRayEps = 1e-4f;
constexpr float kShadowEps = 1e-3f;
constexpr int kQuadStride = 24;

struct v3 {
	float x, y, z;
};
__device__ __forceinline__ v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
__device__ __forceinline__ v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
__device__ __forceinline__ float max3(v3 a) { const float m = a.x > a.y ? a.x : a.y; return m > a.z ? m : a.z; }
__device__ __forceinline__ float fabs_(float v) { return __builtin_fabsf(v); }

// path_guiding_integrator.py:16-24
__device__ __forceinline__ float mis_weight(float a, float b)
{
	const float a2 = a * a;
	float r = a > 0.0f ? a2 / (b * b + a2) : 0.0f;
	if (r != r) r = 0.0f;
	return r;
}

struct Frame {
	v3 s, t, n;
};
// Mitsuba coordinate_system(n) (Duff et al. 2017)
__device__ __forceinline__ Frame make_frame(v3 n)
{
	const float sign = (__float_as_uint(n.z) >> 31) ? -1.0f : 1.0f;
	const float a = -1.0f / (sign + n.z);
	const float b = (n.x * n.y) * a;
	Frame f;
	f.n = n;
	f.s = V(1.0f + (sign * (n.x * n.x)) * a, sign * b, -sign * n.x);
	f.t = V(b, sign + (n.y * n.y) * a, -n.y);
	return f;
}
__device__ __forceinline__ v3 to_local(const Frame &f, v3 v) { return V(dot3(v, f.s), dot3(v, f.t), dot3(v, f.n)); }
__device__ __forceinline__ v3 to_world(const Frame &f, v3 v)
{
	return vadd(vadd(vscale(f.s, v.x), vscale(f.t, v.y)), vscale(f.n, v.z));
}

// closest hit over all quads, 0 < t < tmax (scene.ray_intersect / ray_test)
__device__ __forceinline__ int intersect(int nq, const float *__restrict__ quads, v3 o, v3 d, float tmax, float &t_out)
{
	int best = -1;
	float bt = tmax;
	for (int q = 0; q < nq; ++q) {
		const float *Q = quads + q * kQuadStride;
		const v3 n = ld3(Q + 9);
		const float denom = dot3(n, d);
		if (denom == 0.0f) continue;
		const float t = dot3(n, vsub(ld3(Q), o)) / denom;
		if (!(t > 0.0f && t < bt)) continue;
		const v3 w = vsub(vadd(o, vscale(d, t)), ld3(Q));
		const float u = dot3(w, ld3(Q + 3)) * Q[12];
		const float v = dot3(w, ld3(Q + 6)) * Q[13];
		if (u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f) { bt = t; best = q; }
	}
	t_out = bt;
	return best;
}

// Mitsuba warp::square_to_cosine_hemisphere (concentric disk)
__device__ __forceinline__ v3 square_to_cosine_hemisphere(float u, float v)
{
	const float x = 2.0f * u - 1.0f, y = 2.0f * v - 1.0f;
	const bool is_zero = (x == 0.0f) && (y == 0.0f);
	const bool q13 = fabs_(x) < fabs_(y);
	const float r = q13 ? y : x, rp = q13 ? x : y;
	float phi = (0.25f * 3.14159265358979323846f) * (rp / r);
	if (q13) phi = (0.5f * 3.14159265358979323846f) - phi;
	if (is_zero) phi = 0.0f;
	float s, c;
	sincos_f32(phi, s, c);
	const float px = r * c, py = r * s;
	const float zz = 1.0f - (px * px + py * py);
	float z = zz > 0.0f ? __builtin_sqrtf(zz) : 0.0f;
	if (z == 0.0f) z = 1e-10f;
	return V(px, py, z);
}

__device__ __forceinline__ void bsdf_eval_pdf(v3 refl, v3 wi, v3 wo, bool active, v3 &value, float &pdf)
{
	value = V(0, 0, 0);
	pdf = 0.0f;
	if (!active) return;
	if (wi.z < 0.0f) { wi.z = -wi.z; wo.z = -wo.z; }
	if (!(wi.z > 0.0f && wo.z > 0.0f)) return;
	value = vscale(vscale(refl, kInvPiF), wo.z);
	pdf = kInvPiF * wo.z;
}

__device__ __forceinline__ void bsdf_sample(v3 refl, v3 wi, float u, float v, bool active, v3 &wo, float &pdf,
                                            v3 &weight, float &eta)
{
	wo = V(0, 0, 0); pdf = 0.0f; weight = V(0, 0, 0); eta = 0.0f;
	if (!active) return;
	const bool flip = wi.z < 0.0f;
	const float cos_i = flip ? -wi.z : wi.z;
	if (!(cos_i > 0.0f)) return;
	v3 w = square_to_cosine_hemisphere(u, v);
	const float p = kInvPiF * w.z;
	eta = 1.0f;
	pdf = p;
	if (p > 0.0f) weight = refl;
	if (flip) w.z = -w.z;
	wo = w;
}

struct RenderArgs {
	TreeView tree;
	const float *quads;
	int n_quads, emitter_quad;
	pg_camera cam;
	uint64_t n_lanes, n_pixels;      // of this pass (tile)
	uint64_t pixel_begin, film_pixels; // first pixel of the tile, pixels of the whole film
	int spp, max_depth, rr_depth, guided, record, store_nee;
	float frac;
	uint32_t seed;
	DepthCounters *dc;
	const uint32_t *order;       // compacted live-ray list (NULL for the first bounce)
	const uint32_t *order_count; // [2], as written by k_compact_lanes
	// per-lane state (planar)
	float *ray_o, *ray_d, *thr, *L, *prev_p, *prev_pdf, *ior;
	uint32_t *depth;
	uint8_t *active, *prev_delta;
	uint64_t *rng_state, *rng_inc;
	// dense records
	uint8_t *r_act;
	float *r_pos, *r_dir, *r_bsdf, *r_tb, *r_tr, *r_nee, *r_dnee, *r_wp;
};

__global__ __launch_bounds__(kRBlock) void k_generate(RenderArgs a)
{
	const uint64_t lane = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	if (lane >= a.n_lanes) return;
	const uint64_t N = a.n_lanes;
	// streams are keyed by the GLOBAL lane id (pixel*spp + s): a tile renders exactly the samples
	// the full-frame pass would, whatever the number of ranks
	Pcg32 rng = pcg32_seed(a.seed, (uint32_t)(a.pixel_begin * (uint64_t)a.spp + lane));
	const uint64_t pixel = a.pixel_begin + lane / (uint64_t)a.spp;
	const int W = a.cam.width, H = a.cam.height;
	const float px = (float)(pixel % (uint64_t)W), py = (float)(pixel / (uint64_t)W);
	const float jx = rng.next_f32(), jy = rng.next_f32();
	const float tan_y = a.cam.tan_half_fov_x / ((float)W / (float)H);
	const float cx = (1.0f - 2.0f * ((px + jx) / (float)W)) * a.cam.tan_half_fov_x;
	const float cy = (1.0f - 2.0f * ((py + jy) / (float)H)) * tan_y;
	const float len = __builtin_sqrtf((cx * cx + cy * cy) + 1.0f);
	const v3 dc = V(cx / len, cy / len, 1.0f / len);
	const v3 d = vadd(vadd(vscale(ld3(a.cam.axis_x), dc.x), vscale(ld3(a.cam.axis_y), dc.y)), vscale(ld3(a.cam.axis_z), dc.z));
	a.ray_o[lane] = a.cam.origin[0]; a.ray_o[N + lane] = a.cam.origin[1]; a.ray_o[2 * N + lane] = a.cam.origin[2];
	a.ray_d[lane] = d.x; a.ray_d[N + lane] = d.y; a.ray_d[2 * N + lane] = d.z;
	for (int c = 0; c < 3; ++c) { a.thr[c * N + lane] = 1.0f; a.L[c * N + lane] = 0.0f; a.prev_p[c * N + lane] = 0.0f; }
	a.prev_pdf[lane] = 1.0f;
	a.ior[lane] = 1.0f;
	a.depth[lane] = 0;
	a.active[lane] = 1;
	a.prev_delta[lane] = 1;
	a.rng_state[lane] = rng.state;
	a.rng_inc[lane] = rng.inc;
}

// one loop iteration of :179-381 for every live lane
__global__ __launch_bounds__(kRBlock) void k_bounce(RenderArgs a)
{
	__shared__ uint4 s_kd[kLdsKdNodes];
	const uint64_t tid = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	const uint64_t N = a.n_lanes;
	// live-ray list from the stream compaction that ran after the previous bounce (none before the
	// first): thread t serves ray slot order[N-1-t]; workgroups past the live count retire at once
	const uint64_t live = a.order ? (uint64_t)a.order_count[0] + (uint64_t)a.order_count[1] : N;
	if ((uint64_t)blockIdx.x * kRBlock >= live) return;
	if (a.guided) stage_kd_top(s_kd, a.tree.kd, a.tree.n_kd);
	if (tid >= live || tid >= N) return;
	uint64_t lane = tid;
	if (a.order) {
		const uint64_t front = a.order_count[0];
		lane = a.order[tid < front ? tid : N - 1 - (tid - front)];
	}
	if (!a.active[lane]) return;
	const int D = a.max_depth;
	const float f = a.frac;
	const float *quads = a.quads;
	Pcg32 rng = {a.rng_state[lane], a.rng_inc[lane]};
	v3 ray_o = V(a.ray_o[lane], a.ray_o[N + lane], a.ray_o[2 * N + lane]);
	v3 ray_d = V(a.ray_d[lane], a.ray_d[N + lane], a.ray_d[2 * N + lane]);
	v3 thr = V(a.thr[lane], a.thr[N + lane], a.thr[2 * N + lane]);
	v3 L = V(a.L[lane], a.L[N + lane], a.L[2 * N + lane]);
	const v3 prev_p = V(a.prev_p[lane], a.prev_p[N + lane], a.prev_p[2 * N + lane]);
	const float prev_bsdf_pdf = a.prev_pdf[lane];
	const bool prev_delta = a.prev_delta[lane] != 0;
	uint32_t depth = a.depth[lane];
	float ior = a.ior[lane];

	// ---- :185 ray_intersect ----
	float t_hit;
	const int q = intersect(a.n_quads, quads, ray_o, ray_d, __builtin_huge_valf(), t_hit);
	const bool valid = q >= 0;
	const float *Q = valid ? quads + q * kQuadStride : quads;
	const v3 p = valid ? vadd(ray_o, vscale(ray_d, t_hit)) : V(0, 0, 0);
	const v3 n = valid ? ld3(Q + 9) : V(0, 0, 1);
	const Frame fr = make_frame(n);
	const v3 wi = to_local(fr, V(-ray_d.x, -ray_d.y, -ray_d.z));
	const v3 refl = valid ? ld3(Q + 16) : V(0, 0, 0);
	const bool is_em = valid && Q[15] != 0.0f;
	// ---- :189-200 direct emission ----
	const v3 em_radiance = (is_em && wi.z > 0.0f) ? ld3(Q + 19) : V(0, 0, 0);
	float emitter_pdf = 0.0f;
	if (is_em && !prev_delta) {
		const v3 dd = vsub(p, prev_p);
		const float d2 = dot3(dd, dd), dist = __builtin_sqrtf(d2);
		const v3 dn = vdivs(dd, dist);
		const float dp = dot3(dn, n);
		if (dp < 0.0f) emitter_pdf = d2 / (fabs_(dp) * Q[14]);
	}
	const float mis = mis_weight(prev_bsdf_pdf, emitter_pdf);
	const v3 Le = vmul(vscale(thr, mis), em_radiance);
	// ---- :207-220 emitter sampling ----
	bool active_next = (depth + 1 < (uint32_t)D) && valid;
	bool active_em = active_next;
	const float e1 = rng.next_f32(), e2 = rng.next_f32(); // :214, unmasked
	v3 ds_d = V(0, 0, 0), em_weight = V(0, 0, 0);
	float ds_pdf = 0.0f;
	if (active_em && a.emitter_quad >= 0) {
		const float *E = quads + a.emitter_quad * kQuadStride;
		const v3 pl = vadd(vadd(ld3(E), vscale(ld3(E + 3), e1)), vscale(ld3(E + 6), e2));
		const v3 dir0 = vsub(pl, p);
		float mag = (1.0f + max3(V(fabs_(p.x), fabs_(p.y), fabs_(p.z)))) * kRayEps;
		if (dot3(n, dir0) < 0.0f) mag = -mag;
		const v3 so = vadd(p, vscale(n, mag));
		const v3 dd = vsub(pl, p);
		const float d2 = dot3(dd, dd), dist = __builtin_sqrtf(d2);
		ds_d = vdivs(dd, dist);
		const float dp = dot3(ds_d, ld3(E + 9));
		float pdf = dp < 0.0f ? d2 / (fabs_(dp) * E[14]) : 0.0f;
		if (!(pdf == pdf) || pdf == __builtin_huge_valf()) pdf = 0.0f;
		ds_pdf = pdf;
		if (pdf > 0.0f) {
			const v3 sd = vsub(pl, so);
			const float sdist = __builtin_sqrtf(dot3(sd, sd));
			const v3 sdn = vdivs(sd, sdist);
			float th;
			const bool occ = intersect(a.n_quads, quads, so, sdn, sdist * (1.0f - kShadowEps), th) >= 0;
			if (!occ) em_weight = vdivs(ld3(E + 19), pdf);
		}
	}
	active_em = active_em && (ds_pdf != 0.0f); // :216
	const v3 wo_em = to_local(fr, ds_d);
	v3 bsdf_value_em;
	float bsdf_pdf_em;
	bsdf_eval_pdf(refl, wi, wo_em, active_em, bsdf_value_em, bsdf_pdf_em);
	// ---- :223-256 NEE MIS against the mixture pdf ----
	const bool active_sd_em = active_em && a.guided;
	const float pdf_diffuse = 1.0f; // :222-241
	TreeHead head = {kNoRecord, 0.0f};
	bool tree_known = false;
	float sdtree_pdf_em = 1.0f;
	uint32_t lv;
	unsigned c_kd = 0, c_kdq = 0, c_q = 0, c_qq = 0; // descent statistics for the byte model
	if (active_sd_em) {
		KdNode leaf;
		kd_descend_lds(a.tree.kd, s_kd, p.x, p.y, p.z, inside_root(a.tree, p.x, p.y, p.z), leaf, lv);
		c_kd += lv; ++c_kdq;
		const uint2 hv = *reinterpret_cast<const uint2 *>(a.tree.head + leaf.tree);
		head.root_rec = hv.x;
		head.root_irr = __uint_as_float(hv.y);
		tree_known = true;
		float cx, cy;
		dir_to_canonical(ds_d.x, ds_d.y, ds_d.z, cx, cy);
		sdtree_pdf_em = quad_pdf(a.tree.rec, head, cx, cy, lv);
		c_q += lv; ++c_qq;
	}
	float surface_pdf_em = f * bsdf_pdf_em + ((1.0f - f) * sdtree_pdf_em) * pdf_diffuse;
	if (!a.guided) surface_pdf_em = bsdf_pdf_em;
	const float mis_em = mis_weight(ds_pdf, surface_pdf_em);
	const v3 Lr_dir = vmul(vmul(vscale(thr, mis_em), bsdf_value_em), em_weight);
	L = vadd(L, vadd(Le, Lr_dir)); // :261
	// ---- :272-311 next direction ----
	float s2x = 0.0f, s2y = 0.0f;
	if (active_next) { rng.skip(); s2x = rng.next_f32(); s2y = rng.next_f32(); } // next_1d (unused by diffuse), next_2d
	v3 wo_local, bsdf_weight;
	float bsdf_pdf, eta;
	bsdf_sample(refl, wi, s2x, s2y, active_next, wo_local, bsdf_pdf, bsdf_weight, eta);
	v3 bsdf_value = vscale(bsdf_weight, bsdf_pdf);
	float woPdf = bsdf_pdf;
	v3 wo_world = to_world(fr, wo_local);
	const bool do_mis = active_next && a.guided; // no delta lobes in this substrate
	bool pick_tree = false;
	if (active_next) pick_tree = rng.next_f32() > f; // :286
	const bool smp_tree = pick_tree && do_mis;
	const bool bsdf_mis = do_mis && !smp_tree;
	float sdtree_pdf = 1.0f;
	if ((smp_tree || bsdf_mis) && !tree_known) {
		KdNode leaf;
		kd_descend_lds(a.tree.kd, s_kd, p.x, p.y, p.z, inside_root(a.tree, p.x, p.y, p.z), leaf, lv);
		c_kd += lv; ++c_kdq;
		const uint2 hv = *reinterpret_cast<const uint2 *>(a.tree.head + leaf.tree);
		head.root_rec = hv.x;
		head.root_irr = __uint_as_float(hv.y);
	}
	if (smp_tree) { // :301-304
		float dx, dy, dz;
		quad_sample(a.tree.rec, head, rng, dx, dy, dz, sdtree_pdf, lv);
		c_q += lv; ++c_qq;
		wo_world = V(dx, dy, dz);
		wo_local = to_local(fr, wo_world);
		bsdf_eval_pdf(refl, wi, wo_local, true, bsdf_value, bsdf_pdf);
	}
	if (bsdf_mis) { // :307
		float cx, cy;
		dir_to_canonical(wo_world.x, wo_world.y, wo_world.z, cx, cy);
		sdtree_pdf = quad_pdf(a.tree.rec, head, cx, cy, lv);
		c_q += lv; ++c_qq;
	}
	if (a.dc && c_kdq) { // instrumented passes only (pg_enable_depth_counters)
		atomicAdd(&a.dc->kd_levels, (unsigned long long)c_kd);
		atomicAdd(&a.dc->kd_queries, (unsigned long long)c_kdq);
		atomicAdd(&a.dc->quad_levels, (unsigned long long)c_q);
		atomicAdd(&a.dc->quad_queries, (unsigned long long)c_qq);
	}
	if (do_mis) { // :310-311
		woPdf = f * bsdf_pdf + (1.0f - f) * sdtree_pdf;
		bsdf_weight = vdivs(bsdf_value, woPdf);
		// deliberate deviation: 0/0 when a zero-energy tree proposes a direction below
		// the surface; the reference's throughput turns NaN there, here the path simply ends
		if (!(woPdf > 0.0f)) bsdf_weight = V(0, 0, 0);
	}
	// ---- :318-346 record ----
	if (a.record && valid) {
		const uint64_t S = N * (uint64_t)D;
		// the reference's slot is ray*max_depth + depth (:318), a stride-max_depth scatter; the library's
		// own buffer is depth-major so that a wavefront's stores coalesce (pg_process_and_splat is told)
		const uint64_t g = (uint64_t)depth * N + lane;
		float c0, c1;
		a.r_act[g] = 1;
		a.r_pos[g] = p.x; a.r_pos[S + g] = p.y; a.r_pos[2 * S + g] = p.z;
		dir_to_canonical(wo_world.x, wo_world.y, wo_world.z, c0, c1);
		a.r_dir[g] = c0; a.r_dir[S + g] = c1;
		a.r_bsdf[g] = bsdf_weight.x; a.r_bsdf[S + g] = bsdf_weight.y; a.r_bsdf[2 * S + g] = bsdf_weight.z;
		a.r_tb[g] = thr.x; a.r_tb[S + g] = thr.y; a.r_tb[2 * S + g] = thr.z;
		a.r_tr[g] = L.x; a.r_tr[S + g] = L.y; a.r_tr[2 * S + g] = L.z;
		if (a.store_nee) {
			const v3 rn = vdiv(Lr_dir, thr);
			a.r_nee[g] = rn.x; a.r_nee[S + g] = rn.y; a.r_nee[2 * S + g] = rn.z;
			dir_to_canonical(ds_d.x, ds_d.y, ds_d.z, c0, c1);
			a.r_dnee[g] = c0; a.r_dnee[S + g] = c1;
		} else {
			a.r_nee[g] = 0.0f; a.r_nee[S + g] = 0.0f; a.r_nee[2 * S + g] = 0.0f;
			a.r_dnee[g] = 0.0f; a.r_dnee[S + g] = 0.0f;
		}
		a.r_wp[g] = woPdf;
	}
	// ---- :352-381 advance ----
	{
		float mag = (1.0f + max3(V(fabs_(p.x), fabs_(p.y), fabs_(p.z)))) * kRayEps;
		if (dot3(n, wo_world) < 0.0f) mag = -mag;
		ray_o = vadd(p, vscale(n, mag));
		ray_d = wo_world;
	}
	ior = ior * eta;
	thr = vmul(thr, bsdf_weight);
	const float tmax = max3(thr);
	active_next = active_next && (tmax != 0.0f);
	float rr_prob = tmax * (ior * ior);
	if (!(rr_prob < 0.95f)) rr_prob = 0.95f;
	const bool rr_active = depth >= (uint32_t)a.rr_depth;
	const float rr = rng.next_f32(); // :377, unmasked
	const bool rr_continue = rr < rr_prob;
	active_next = active_next && (!rr_active || rr_continue);
	if (valid) depth += 1;
	// store state
	a.rng_state[lane] = rng.state;
	a.ray_o[lane] = ray_o.x; a.ray_o[N + lane] = ray_o.y; a.ray_o[2 * N + lane] = ray_o.z;
	a.ray_d[lane] = ray_d.x; a.ray_d[N + lane] = ray_d.y; a.ray_d[2 * N + lane] = ray_d.z;
	a.thr[lane] = thr.x; a.thr[N + lane] = thr.y; a.thr[2 * N + lane] = thr.z;
	a.L[lane] = L.x; a.L[N + lane] = L.y; a.L[2 * N + lane] = L.z;
	a.prev_p[lane] = p.x; a.prev_p[N + lane] = p.y; a.prev_p[2 * N + lane] = p.z;
	a.prev_pdf[lane] = woPdf;
	a.prev_delta[lane] = 0;
	a.ior[lane] = ior;
	a.depth[lane] = depth;
	a.active[lane] = active_next ? 1 : 0;
}

// :400-431: valid flag and per-pixel sums, samples of a pixel added in lane order
__global__ __launch_bounds__(kRBlock) void k_finish(RenderArgs a, uint8_t *__restrict__ valid_out,
                                                    float *__restrict__ sumL, float *__restrict__ sumL2)
{
	const uint64_t pix = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	if (pix >= a.n_pixels) return;
	const uint64_t N = a.n_lanes, P = a.film_pixels;
	const uint64_t gpix = a.pixel_begin + pix; // the sums are full-film arrays
	for (int s = 0; s < a.spp; ++s) {
		const uint64_t lane = pix * (uint64_t)a.spp + (uint64_t)s;
		if (valid_out) valid_out[lane] = a.depth[lane] != 0;
		if (sumL && sumL2)
			for (int c = 0; c < 3; ++c) {
				const float v = a.L[c * N + lane];
				sumL[c * P + gpix] = sumL[c * P + gpix] + v;
				sumL2[c * P + gpix] = sumL2[c * P + gpix] + v * v;
			}
	}
}

} // namespace pg

using namespace pg;

// library-owned renderer state
struct pg_render_state {
	DevBuf<float> quads;
	int n_quads = 0, emitter_quad = -1;
	pg_camera cam;
	bool have_scene = false;
	DevBuf<float> ray_o, ray_d, thr, prev_p, prev_pdf, ior;
	DevBuf<uint32_t> depth;
	DevBuf<uint8_t> active, prev_delta;
	DevBuf<uint64_t> rng_state, rng_inc;
	DevBuf<uint32_t> order, order_count;
	DevBuf<uint8_t> r_act;
	DevBuf<float> r_pos, r_dir, r_bsdf, r_tb, r_tr, r_nee, r_dnee, r_wp;
	// optional per-kernel timing: (kind, start, stop) event triples still to be read
	bool timing_on = false;
	struct Ev { int kind; hipEvent_t a, b; };
	std::vector<Ev> events;
	pg_kernel_timing acc = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {
struct Timed { // records an event pair around a launch when timing is enabled
	pg_render_state *r;
	hipStream_t s;
	int kind;
	hipEvent_t a = nullptr, b = nullptr;
	Timed(pg_render_state *r_, hipStream_t s_, int kind_) : r(r_), s(s_), kind(kind_)
	{
		if (!r->timing_on) return;
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
		(void)hipEventRecord(a, s);
	}
	~Timed()
	{
		if (!a) return;
		(void)hipEventRecord(b, s);
		r->events.push_back({kind, a, b});
	}
};
} // namespace

static pg_render_state *rstate(pg_context *ctx)
{
	if (!ctx->render) ctx->render = new pg_render_state();
	return ctx->render;
}

void pg::destroy_render_state(pg_context *ctx)
{
	delete ctx->render;
	ctx->render = nullptr;
}

extern "C" {

int pg_scene_set(pg_context *ctx, uint64_t n_quads, const float *h_quads, const pg_camera *cam)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!h_quads || !cam || n_quads == 0 || n_quads > 4096)
		return fail(ctx, PG_ERR_INVALID, "pg_scene_set: need 1..4096 quads and a camera");
	if (cam->width <= 0 || cam->height <= 0) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: bad film size");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	pg_render_state *r = rstate(ctx);
	PG_HIP(ctx, r->quads.ensure(n_quads * kQuadStride));
	PG_HIP(ctx, hipMemcpy(r->quads.p, h_quads, n_quads * kQuadStride * sizeof(float), hipMemcpyHostToDevice));
	r->n_quads = (int)n_quads;
	r->emitter_quad = -1;
	for (uint64_t q = 0; q < n_quads; ++q)
		if (h_quads[q * kQuadStride + 15] != 0.0f) { r->emitter_quad = (int)q; break; }
	r->cam = *cam;
	r->have_scene = true;
	return PG_OK;
}

int pg_render_pass(pg_context *ctx, const pg_pass_params *prm, float *L_out, uint8_t *valid_out, float *sumL,
                   float *sumL2, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!ctx->configured) return fail(ctx, PG_ERR_INVALID, "call pg_setup or pg_import first");
	if (!ctx->render || !ctx->render->have_scene) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: call pg_scene_set first");
	if (!prm || !L_out) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: NULL pointer");
	if (prm->spp <= 0 || ctx->max_depth <= 0) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: spp and max_depth must be > 0");
	if ((sumL == nullptr) != (sumL2 == nullptr)) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: sumL and sumL2 go together");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t s = (hipStream_t)stream;
	pg_render_state *r = ctx->render;
	const uint64_t film = (uint64_t)r->cam.width * (uint64_t)r->cam.height;
	if (prm->pixel_begin > film || prm->pixel_count > film - prm->pixel_begin)
		return fail(ctx, PG_ERR_INVALID, "pg_render_pass: pixel range outside the film");
	const uint64_t P = prm->pixel_count ? prm->pixel_count : film - prm->pixel_begin; // 0 = to the end
	if (P == 0) return PG_OK;
	const uint64_t N = P * (uint64_t)prm->spp;
	const int D = ctx->max_depth;
	const uint64_t S = N * (uint64_t)D;
	if (S > 0xffffffffull) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: more than 2^32 record slots in one pass");
	const bool record = !ctx->is_final;
	PG_HIP(ctx, r->ray_o.ensure(3 * N)); PG_HIP(ctx, r->ray_d.ensure(3 * N)); PG_HIP(ctx, r->thr.ensure(3 * N));
	PG_HIP(ctx, r->prev_p.ensure(3 * N)); PG_HIP(ctx, r->prev_pdf.ensure(N)); PG_HIP(ctx, r->ior.ensure(N));
	PG_HIP(ctx, r->depth.ensure(N)); PG_HIP(ctx, r->active.ensure(N)); PG_HIP(ctx, r->prev_delta.ensure(N));
	PG_HIP(ctx, r->rng_state.ensure(N)); PG_HIP(ctx, r->rng_inc.ensure(N));
	PG_HIP(ctx, r->order.ensure(N)); PG_HIP(ctx, r->order_count.ensure(2));
	if (record) {
		PG_HIP(ctx, r->r_act.ensure(S)); PG_HIP(ctx, r->r_pos.ensure(3 * S)); PG_HIP(ctx, r->r_dir.ensure(2 * S));
		PG_HIP(ctx, r->r_bsdf.ensure(3 * S)); PG_HIP(ctx, r->r_tb.ensure(3 * S)); PG_HIP(ctx, r->r_tr.ensure(3 * S));
		PG_HIP(ctx, r->r_nee.ensure(3 * S)); PG_HIP(ctx, r->r_dnee.ensure(2 * S)); PG_HIP(ctx, r->r_wp.ensure(S));
		// only the `active` column has to be cleared: inactive slots are never read for anything else
		PG_HIP(ctx, hipMemsetAsync(r->r_act.p, 0, S, s));
	}
	RenderArgs a;
	a.tree = ctx->view();
	a.quads = r->quads.p;
	a.n_quads = r->n_quads;
	a.emitter_quad = r->emitter_quad;
	a.cam = r->cam;
	a.n_lanes = N;
	a.n_pixels = P;
	a.pixel_begin = prm->pixel_begin;
	a.film_pixels = film;
	a.spp = prm->spp;
	a.max_depth = D;
	a.rr_depth = prm->rr_depth;
	a.guided = ctx->iteration > 1 ? 1 : 0; // :223, 250, 283
	a.record = record ? 1 : 0;
	a.store_nee = ctx->store_nee;
	a.frac = ctx->bsdf_fraction;
	a.seed = prm->seed;
	a.dc = ctx->dc_on ? ctx->dc : nullptr;
	a.ray_o = r->ray_o.p; a.ray_d = r->ray_d.p; a.thr = r->thr.p; a.L = L_out; a.prev_p = r->prev_p.p;
	a.prev_pdf = r->prev_pdf.p; a.ior = r->ior.p; a.depth = r->depth.p; a.active = r->active.p;
	a.prev_delta = r->prev_delta.p; a.rng_state = r->rng_state.p; a.rng_inc = r->rng_inc.p;
	a.r_act = r->r_act.p; a.r_pos = r->r_pos.p; a.r_dir = r->r_dir.p; a.r_bsdf = r->r_bsdf.p; a.r_tb = r->r_tb.p;
	a.r_tr = r->r_tr.p; a.r_nee = r->r_nee.p; a.r_dnee = r->r_dnee.p; a.r_wp = r->r_wp.p;
	const dim3 grid((unsigned)((N + kRBlock - 1) / kRBlock));
	{
		Timed t(r, s, 0);
		hipLaunchKernelGGL(k_generate, grid, dim3(kRBlock), 0, s, a);
	}
	a.order = nullptr;
	a.order_count = r->order_count.p;
	for (int it = 0; it < D; ++it) {
		{
			Timed t(r, s, 1);
			hipLaunchKernelGGL(k_bounce, grid, dim3(kRBlock), 0, s, a);
		}
		if (it + 1 < D) { // active-ray stream compaction for the next bounce
			Timed t(r, s, 4);
			launch_compact_lanes(N, r->active.p, nullptr, r->order.p, r->order_count.p, s);
			a.order = r->order.p;
		}
	}
	PG_HIP(ctx, hipGetLastError());
	if (record) {
		pg_dense_records d;
		d.active = r->r_act.p; d.position = r->r_pos.p; d.direction = r->r_dir.p; d.bsdf = r->r_bsdf.p;
		d.throughput_bsdf = r->r_tb.p; d.throughput_radiance = r->r_tr.p; d.radiance_nee = r->r_nee.p;
		d.direction_nee = r->r_dnee.p; d.wo_pdf = r->r_wp.p;
		Timed t(r, s, 2);
		// the depth counters of an instrumented pass describe the bounce kernels only
		launch_process_and_splat(ctx->view(), ctx->f.accum_view(), ctx->store_nee, N, D, L_out, d, nullptr, s,
		                         /*depth_major=*/1);
		PG_HIP(ctx, hipGetLastError());
	}
	if (valid_out || sumL) {
		Timed t(r, s, 3);
		hipLaunchKernelGGL(k_finish, dim3((unsigned)((P + kRBlock - 1) / kRBlock)), dim3(kRBlock), 0, s, a, valid_out,
		                   sumL, sumL2);
		PG_HIP(ctx, hipGetLastError());
	}
	if (r->timing_on) ++r->acc.passes;
	return PG_OK;
}

int pg_enable_kernel_timing(pg_context *ctx, int32_t on)
{
	if (!ctx) return PG_ERR_INVALID;
	rstate(ctx)->timing_on = on != 0;
	return PG_OK;
}

int pg_read_kernel_timing(pg_context *ctx, pg_kernel_timing *out, int32_t reset)
{
	if (!ctx || !out) return PG_ERR_INVALID;
	PG_HIP(ctx, hipSetDevice(ctx->device));
	pg_render_state *r = rstate(ctx);
	for (auto &e : r->events) {
		float ms = 0.0f;
		PG_HIP(ctx, hipEventSynchronize(e.b));
		PG_HIP(ctx, hipEventElapsedTime(&ms, e.a, e.b));
		switch (e.kind) {
		case 0: r->acc.generate_ms += ms; break;
		case 1: r->acc.bounce_ms += ms; ++r->acc.bounce_launches; break;
		case 2: r->acc.splat_ms += ms; ++r->acc.splat_launches; break;
		case 4: r->acc.compact_ms += ms; break;
		default: r->acc.finish_ms += ms; break;
		}
		(void)hipEventDestroy(e.a);
		(void)hipEventDestroy(e.b);
	}
	r->events.clear();
	*out = r->acc;
	if (reset) r->acc = pg_kernel_timing{0, 0, 0, 0, 0, 0, 0, 0};
	return PG_OK;
}

} // extern "C"
