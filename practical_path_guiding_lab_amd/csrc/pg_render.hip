// pg_render.hip -- wavefront renderer substrate around the SD-tree: what the reference's
// PathGuidingIntegrator.sample() (src/path_guiding_integrator.py:126-431) does per pass, with the
// Mitsuba calls it makes (scene.ray_intersect :185, emitter eval/pdf :189-198,
// sample_emitter_direction :213, bsdf.eval_pdf/sample :220, 272, 304, si.to_local/to_world/spawn_ray
// :219, 277, 352, sampler.next_1d/2d) implemented for the scene subset of scenes/cornell-box
// (quads, twosided diffuse BSDFs, a one-sided area emitter, perspective camera) and, in the
// kGeneral instantiations, of scenes/veach-mis (spheres, several emitters, Beckmann rough conductors).
//
// One kernel per bounce over the live rays; the SD-tree queries are the same device functions the
// stand-alone query kernels use, so a bounce costs one KD descent and at most two quadtree descents
// per live lane and no intermediate wavefront buffers.  The first bounce generates its camera rays
// itself, every bounce appends its survivors to the live-ray list of the next one (one atomic per
// workgroup), and a path carries 57 B of state between bounces.  Path-vertex records go to a dense
// slot buffer (the reference's :318, stored depth-major here so that a wavefront's stores coalesce)
// that pg_process_and_splat consumes after the last bounce (:388-395).
//
// Arithmetic mirrors oracle/pg_oracle_render.c operation by operation (fp32, no contraction), so
// radiance, records and therefore the refined trees are bit-identical to the CPU restatement.
// (The microfacet helpers are inlined here too since round 5: kept out of line, every call cost the level-1 kernels a 32-byte
// stack frame in scratch memory; inlined, k_bounce<*, 1> is 87 / 89 registers at five waves per SIMD with no scratch at
// all -- veach-mis 2.33 -> 2.32 ms per pass, profiles/r05/ab_fused_kernel_helpers_inlined.txt.)
#include "pg_render_dev.hpp"

namespace pg {


// One loop iteration of :179-381 for one live lane; returns whether the path continues.
// kFirst: the camera ray is generated here (mi.render's sensor.sample_ray_differential: one 2-D
// jitter draw per sample, box reconstruction) instead of being read back from a generate kernel.
// kGeneral: the scene has spheres or rough conductors; false compiles the all-diffuse quad scene only.
// stash (k_bounce; nullptr in the tail kernel): column threadIdx.x of a [kBounceStash][kRBlock] array in LDS, where values
// that only the code BEHIND an SD-tree walk reads wait while the walk runs -- this kernel leaves the LDS almost empty, and
// the walks are where its registers peak (tools/vgpr_liveness.py).
constexpr int kBounceStash = 17;
typedef __attribute__((address_space(3))) float LdsFloat;
template <bool kFirst, int kGeneral, bool kStash = false>
__device__ __forceinline__ bool bounce_lane(const RenderArgs &a, const uint4 *s_kd, const uint64_t lane,
                                            const uint64_t rec_slot, const uint32_t depth, LdsFloat *stash = nullptr)
{
	const uint64_t N = a.n_lanes;
	const int D = a.max_depth;
	const float f = a.frac;
	const Shapes &sh = a.shapes;
	Pcg32 rng;
	v3 ray_o, ray_d, thr, L, prev_p;
	float prev_bsdf_pdf;
	const bool prev_delta = kFirst; // no delta lobes in these scenes (feature levels 0 and 1): only the camera "vertex" counts as one
	const float ior = 1.0f;         // every BSDF of these scenes has eta 1: the running product stays exactly 1
	static_assert(kGeneral < 2, "mesh scenes run the split pipeline of pg_render_wave.hip");
	if (kFirst) {
		const uint64_t pixel = global_pixel(a, lane / (uint64_t)a.spp);
		rng = lane_stream(a.seed, a.spp, a.batched, pixel, (uint32_t)(lane % (uint64_t)a.spp));
		const int W = a.cam.width, H = a.cam.height;
		const float px = (float)(pixel % (uint64_t)W), py = (float)(pixel / (uint64_t)W);
		const float jx = rng.next_f32(), jy = rng.next_f32();
		const float tan_y = a.cam.tan_half_fov_x / ((float)W / (float)H);
		const float cx = (1.0f - 2.0f * ((px + jx) / (float)W)) * a.cam.tan_half_fov_x;
		const float cy = (1.0f - 2.0f * ((py + jy) / (float)H)) * tan_y;
		const float len = __builtin_sqrtf((cx * cx + cy * cy) + 1.0f);
		const v3 dc = V(cx / len, cy / len, 1.0f / len);
		ray_d = vadd(vadd(vscale(ld3(a.cam.axis_x), dc.x), vscale(ld3(a.cam.axis_y), dc.y)), vscale(ld3(a.cam.axis_z), dc.z));
		ray_o = ld3(a.cam.origin);
		thr = V(1, 1, 1);
		L = V(0, 0, 0);
		prev_p = V(0, 0, 0);
		prev_bsdf_pdf = 1.0f;
	} else {
		rng.state = a.rng_state[lane];
		rng.inc = a.rng_inc[lane];
		ray_d = V(a.ray_d[lane], a.ray_d[N + lane], a.ray_d[2 * N + lane]);
		thr = V(a.thr[lane], a.thr[N + lane], a.thr[2 * N + lane]);
		L = V(a.L[lane], a.L[N + lane], a.L[2 * N + lane]);
		prev_p = V(a.prev_p[lane], a.prev_p[N + lane], a.prev_p[2 * N + lane]);
		prev_bsdf_pdf = a.prev_pdf[lane];
		// :352 spawn_ray of the previous vertex: the same three operations that produced the origin
		const uint32_t pq = a.prev_quad[lane];
		const v3 pn = normal_at<kGeneral>(sh, (int)pq, prev_p);
		float mag = (1.0f + max3(V(fabs_(prev_p.x), fabs_(prev_p.y), fabs_(prev_p.z)))) * kRayEps;
		if (dot3(pn, ray_d) < 0.0f) mag = -mag;
		ray_o = vadd(prev_p, vscale(pn, mag));
	}

	// ---- :185 ray_intersect ----
	float t_hit;
	const int q = intersect<kGeneral>(sh, ray_o, ray_d, __builtin_huge_valf(), t_hit);
	const bool valid = q >= 0;
	Surface sf;
	sf.p = V(0, 0, 0); sf.n = V(0, 0, 1); sf.ng = V(0, 0, 1); sf.radiance = V(0, 0, 0); sf.is_em = false;
	sf.m.type = 0; sf.m.refl = V(0, 0, 0); sf.m.M = nullptr; sf.m.one_sided = false;
	if (valid) sf = surface_at<kGeneral>(sh, a.mats, q, ray_o, ray_d, t_hit);
	const v3 p = sf.p, n = sf.n;
	const Material &mt = sf.m;
	const Frame fr = make_frame(n);
	const v3 wi = to_local(fr, V(-ray_d.x, -ray_d.y, -ray_d.z));
	const bool is_em = valid && sf.is_em;
	const float inv_em_count = 1.0f / (float)a.n_emitters; // only used when an emitter was hit
	// ---- :189-200 direct emission ----
	const v3 em_radiance = (is_em && wi.z > 0.0f) ? sf.radiance : V(0, 0, 0);
	float emitter_pdf = 0.0f;
	if (is_em && !prev_delta) emitter_pdf = emitter_hit_pdf<kGeneral>(sh, q, prev_p, p, n, inv_em_count);
	const float mis = mis_weight(prev_bsdf_pdf, emitter_pdf);
	const v3 Le = vmul(vscale(thr, mis), em_radiance);
	// ---- :207-220 emitter sampling ----
	bool active_next = (depth + 1 < (uint32_t)D) && valid;
	bool active_em = active_next; // :210 BSDFFlags.Smooth: every BSDF of these scenes has a non-delta lobe
	const float e1 = rng.next_f32(), e2 = rng.next_f32(); // :214, unmasked
	v3 ds_d = V(0, 0, 0), em_weight = V(0, 0, 0);
	float ds_pdf = 0.0f;
	bool ds_delta = false;
	if (active_em)
		sample_emitter<kGeneral>(sh, a.dir_lights, a.emitters, a.n_emitters, p, sf.ng, e1, e2, ds_d, ds_pdf, em_weight, ds_delta);
	active_em = active_em && (ds_pdf != 0.0f); // :216
	const v3 wo_em = to_local(fr, ds_d);
	v3 bsdf_value_em;
	float bsdf_pdf_em;
	bsdf_eval_pdf<kGeneral>(mt, wi, wo_em, active_em, bsdf_value_em, bsdf_pdf_em);
	// ---- :223-256 NEE MIS against the mixture pdf ----
	const bool active_sd_em = active_em && a.guided;
	const float pdf_diffuse = 1.0f; // :222-241
	TreeHead head = {kNoRecord, 0.0f};
	uint32_t tree_id = 0;
	bool tree_known = false;
	float sdtree_pdf_em = 1.0f;
	uint32_t lv;
	unsigned c_kd = 0, c_kdq = 0, c_q = 0, c_qq = 0; // descent statistics for the byte model
	const bool do_record = a.record && valid;
	// dirToCanonical of the emitter direction feeds the NEE pdf query (:244) and the record (:338):
	// one evaluation serves both
	float nee_cx = 0.0f, nee_cy = 0.0f;
	// a recorded vertex names its accumulators in sdTree_current (pg_list_records): the leaves the walks of sdTree_prev
	// below end in (same topology, :582), or -- unguided iterations, the last vertex of a path -- walks made for them
	uint32_t slot_path = kSlotNone, slot_nee = kSlotNone, tree_flags = 0u;
	const bool nee_slot_wanted = do_record && a.store_nee && active_em;
	if (active_sd_em || (do_record && a.store_nee)) dir_to_canonical(ds_d.x, ds_d.y, ds_d.z, nee_cx, nee_cy);
	if (kStash) { // (what only the NEE term and the record read, behind the first walks)
		stash[0 * kRBlock] = thr.x; stash[1 * kRBlock] = thr.y; stash[2 * kRBlock] = thr.z;
		stash[3 * kRBlock] = L.x; stash[4 * kRBlock] = L.y; stash[5 * kRBlock] = L.z;
		stash[6 * kRBlock] = Le.x; stash[7 * kRBlock] = Le.y; stash[8 * kRBlock] = Le.z;
		stash[9 * kRBlock] = bsdf_value_em.x; stash[10 * kRBlock] = bsdf_value_em.y; stash[11 * kRBlock] = bsdf_value_em.z;
		stash[12 * kRBlock] = em_weight.x; stash[13 * kRBlock] = em_weight.y; stash[14 * kRBlock] = em_weight.z;
		stash[15 * kRBlock] = bsdf_pdf_em; stash[16 * kRBlock] = ds_pdf;
	}
	if (active_sd_em || do_record) {
		KdNode leaf;
		const bool inside = inside_root(a.tree, p.x, p.y, p.z);
		kd_descend_grid(a.tree, reinterpret_cast<const float *>(s_kd), p.x, p.y, p.z, inside, leaf, lv);
		c_kd += lv; ++c_kdq;
		const uint2 hv = gather8(a.tree.head + leaf.tree);
		head.root_rec = hv.x;
		head.root_irr = __uint_as_float(hv.y);
		tree_known = true;
		tree_id = leaf.tree;
		tree_flags = tree_id | (inside ? 0x80000000u : 0u);
	}
	if (active_sd_em) {
		sdtree_pdf_em = quad_pdf_t<true>(a.tree.rec, a.tree.jump, tree_id, head, nee_cx, nee_cy, lv, slot_nee);
		c_q += lv; ++c_qq;
	}
	v3 Le_ = Le, bsdf_value_em_ = bsdf_value_em, em_weight_ = em_weight;
	float bsdf_pdf_em_ = bsdf_pdf_em, ds_pdf_ = ds_pdf;
	if (kStash) {
		thr = V(stash[0 * kRBlock], stash[1 * kRBlock], stash[2 * kRBlock]);
		L = V(stash[3 * kRBlock], stash[4 * kRBlock], stash[5 * kRBlock]);
		Le_ = V(stash[6 * kRBlock], stash[7 * kRBlock], stash[8 * kRBlock]);
		bsdf_value_em_ = V(stash[9 * kRBlock], stash[10 * kRBlock], stash[11 * kRBlock]);
		em_weight_ = V(stash[12 * kRBlock], stash[13 * kRBlock], stash[14 * kRBlock]);
		bsdf_pdf_em_ = stash[15 * kRBlock]; ds_pdf_ = stash[16 * kRBlock];
	}
	float surface_pdf_em = f * bsdf_pdf_em_ + ((1.0f - f) * sdtree_pdf_em) * pdf_diffuse;
	if (!a.guided) surface_pdf_em = bsdf_pdf_em_;
	const float mis_em = mis_weight(ds_pdf_, surface_pdf_em); // :253 (no delta emitters in these scenes)
	v3 Lr_dir = vmul(vmul(vscale(thr, mis_em), bsdf_value_em_), em_weight_);
	L = vadd(L, vadd(Le_, Lr_dir)); // :261
	if (kStash) { // (what only the record and the next bounce's state read, behind the second walks)
		stash[0 * kRBlock] = thr.x; stash[1 * kRBlock] = thr.y; stash[2 * kRBlock] = thr.z;
		stash[3 * kRBlock] = L.x; stash[4 * kRBlock] = L.y; stash[5 * kRBlock] = L.z;
		stash[6 * kRBlock] = Lr_dir.x; stash[7 * kRBlock] = Lr_dir.y; stash[8 * kRBlock] = Lr_dir.z;
	}
	// ---- :272-311 next direction ----
	float s1 = 0.0f, s2x = 0.0f, s2y = 0.0f;
	if (active_next) { // next_1d (lobe choice: only the dielectric reads it), next_2d
		rng.skip(); // (the lobe choice: only dielectrics read it)
		s2x = rng.next_f32();
		s2y = rng.next_f32();
	}
	v3 wo_local, bsdf_weight;
	float bsdf_pdf, eta;
	bool delta;
	bsdf_sample<kGeneral>(mt, wi, s1, s2x, s2y, active_next, wo_local, bsdf_pdf, bsdf_weight, eta, delta);
	v3 bsdf_value = vscale(bsdf_weight, bsdf_pdf);
	float woPdf = bsdf_pdf;
	v3 wo_world = to_world(fr, wo_local);
	const bool do_mis = active_next && !delta && a.guided; // :283
	bool pick_tree = false;
	if (active_next) pick_tree = rng.next_f32() > f; // :286
	const bool smp_tree = pick_tree && do_mis;
	const bool bsdf_mis = do_mis && !smp_tree;
	float sdtree_pdf = 1.0f;
	if ((smp_tree || bsdf_mis) && !tree_known) {
		KdNode leaf;
		kd_descend_grid(a.tree, reinterpret_cast<const float *>(s_kd), p.x, p.y, p.z, inside_root(a.tree, p.x, p.y, p.z), leaf, lv);
		c_kd += lv; ++c_kdq;
		const uint2 hv = gather8(a.tree.head + leaf.tree);
		head.root_rec = hv.x;
		head.root_irr = __uint_as_float(hv.y);
		tree_id = leaf.tree;
	}
	if (smp_tree) { // :301-304
		float dx, dy, dz;
		quad_sample_t<true>(a.tree.rec, a.tree.jump, tree_id, head, rng, dx, dy, dz, sdtree_pdf, lv, slot_path);
		c_q += lv; ++c_qq;
		wo_world = V(dx, dy, dz);
		wo_local = to_local(fr, wo_world);
		bsdf_eval_pdf<kGeneral>(mt, wi, wo_local, true, bsdf_value, bsdf_pdf);
	}
	// dirToCanonical of the continuation direction feeds the pdf query (:307) and the record (:327)
	float wo_cx = 0.0f, wo_cy = 0.0f;
	if (bsdf_mis || do_record) dir_to_canonical(wo_world.x, wo_world.y, wo_world.z, wo_cx, wo_cy);
	if (bsdf_mis) { // :307
		sdtree_pdf = quad_pdf_t<true>(a.tree.rec, a.tree.jump, tree_id, head, wo_cx, wo_cy, lv, slot_path);
		c_q += lv; ++c_qq;
	}
	{ // the leaves no query has walked to: the two walks of QuadTree.addDataPropagate (quadtree.py:443-464), in lock step
		const bool walk_path = do_record && !smp_tree && !bsdf_mis, walk_nee = nee_slot_wanted && !active_sd_em;
		if (walk_path || walk_nee) {
			LeafCursor cp = leaf_cursor(a.tree.jump, tree_id, head, wo_cx, wo_cy, walk_path);
			LeafCursor cn = leaf_cursor(a.tree.jump, tree_id, head, nee_cx, nee_cy, walk_nee);
			quad_find_leaf_slots2(a.tree.rec, cp, cn);
			if (walk_path) { slot_path = cursor_slot(cp); c_q += cp.levels; ++c_qq; }
			if (walk_nee) { slot_nee = cursor_slot(cn); c_q += cn.levels; ++c_qq; }
		}
	}
	if (a.dc && c_kdq) { // instrumented passes only (pg_enable_depth_counters)
		atomicAdd(&a.dc->kd_levels, (unsigned long long)stat_levels(c_kd)); // (c_kd, c_q: sums of statistics words, pg_descent.hpp)
		atomicAdd(&a.dc->kd_queries, (unsigned long long)c_kdq);
		atomicAdd(&a.dc->quad_levels, (unsigned long long)stat_levels(c_q));
		atomicAdd(&a.dc->quad_queries, (unsigned long long)c_qq);
		atomicAdd(&a.dc->layout_bytes, (unsigned long long)(stat_bytes(c_kd) + stat_bytes(c_q)));
	}
	if (kStash) {
		thr = V(stash[0 * kRBlock], stash[1 * kRBlock], stash[2 * kRBlock]);
		L = V(stash[3 * kRBlock], stash[4 * kRBlock], stash[5 * kRBlock]);
		Lr_dir = V(stash[6 * kRBlock], stash[7 * kRBlock], stash[8 * kRBlock]);
	}
	if (do_mis) { // :310-311
		woPdf = f * bsdf_pdf + (1.0f - f) * sdtree_pdf;
		bsdf_weight = vdivs(bsdf_value, woPdf);
		// deliberate deviation: 0/0 when a zero-energy tree proposes a direction below
		// the surface; the reference's throughput turns NaN there, here the path simply ends
		if (!(woPdf > 0.0f)) bsdf_weight = V(0, 0, 0);
	}
	// ---- :318-346 record ----
	// The reference's slot is ray*max_depth + depth (:318), a stride-max_depth scatter into a buffer
	// that is mostly empty at the deeper bounces.  The library's own buffer is a list instead: this
	// launch's thread t owns entry rec_slot = (records of the earlier bounces) + t, so a wavefront's
	// stores coalesce and the splat visits no empty tail; ray_of names the path (k_splat_list looks its
	// final radiance up, :440), kNoRay marks a path that left the scene here.  An entry holds what
	// processPathData (:434-453) needs and the accumulators found above (pg_list_records).
	if (a.record) a.ray_of[rec_slot] = valid ? (uint32_t)lane : 0xffffffffu;
	if (do_record) {
		const uint64_t S = N * (uint64_t)D;
		const uint64_t g = rec_slot;
		a.r_bsdf[g] = bsdf_weight.x; a.r_bsdf[S + g] = bsdf_weight.y; a.r_bsdf[2 * S + g] = bsdf_weight.z;
		a.r_tb[g] = thr.x; a.r_tb[S + g] = thr.y; a.r_tb[2 * S + g] = thr.z;
		a.r_tr[g] = L.x; a.r_tr[S + g] = L.y; a.r_tr[2 * S + g] = L.z;
		float nee_lum = 0.0f;
		if (a.store_nee) { // :336, and the NaN scrub + luminance of :467, 471 (the only use of the three channels)
			v3 rn = vdiv(Lr_dir, thr);
			if (rn.x != rn.x) rn.x = 0.0f;
			if (rn.y != rn.y) rn.y = 0.0f;
			if (rn.z != rn.z) rn.z = 0.0f;
			nee_lum = luminance(rn.x, rn.y, rn.z);
		}
		a.r_nee[g] = nee_lum;
		a.r_wp[g] = woPdf;
		a.r_slot[g] = make_uint2(slot_path, slot_nee);
		a.r_tree[g] = tree_flags;
	}
	// ---- :352-381 advance ----
	// ior (:357): without a dielectric every sampled direction has eta = 1, the running product stays
	// exactly 1 and is not carried; general scenes carry it
	thr = vmul(thr, bsdf_weight);
	const float tmax = max3(thr);
	active_next = active_next && (tmax != 0.0f);
	float rr_prob = tmax * (ior * ior);
	if (!(rr_prob < 0.95f)) rr_prob = 0.95f;
	const bool rr_active = depth >= (uint32_t)a.rr_depth;
	const float rr = rng.next_f32(); // :377, unmasked
	const bool rr_continue = rr < rr_prob;
	active_next = active_next && (!rr_active || rr_continue);
	// ---- state for the next bounce; a path that ends here leaves only its radiance ----
	// (The addresses are made HERE, from the lane's number taken as a new value: left alone the compiler keeps the sixteen
	// 64-bit addresses it formed for the loads at the top -- the same planes -- alive through the whole bounce, 32 of this
	// kernel's 122 vector registers: tools/vgpr_liveness.py)
	uint32_t lane32 = (uint32_t)lane;
	asm volatile("" : "+v"(lane32));
	const uint64_t ln = lane32;
	a.L[ln] = L.x; a.L[N + ln] = L.y; a.L[2 * N + ln] = L.z;
	if (kFirst) a.hit0[ln] = valid ? 1 : 0;
	if (active_next) {
		a.rng_state[ln] = rng.state;
		if (kFirst) a.rng_inc[ln] = rng.inc;
		a.ray_d[ln] = wo_world.x; a.ray_d[N + ln] = wo_world.y; a.ray_d[2 * N + ln] = wo_world.z;
		a.thr[ln] = thr.x; a.thr[N + ln] = thr.y; a.thr[2 * N + ln] = thr.z;
		a.prev_p[ln] = p.x; a.prev_p[N + ln] = p.y; a.prev_p[2 * N + ln] = p.z;
		a.prev_pdf[ln] = woPdf;
		a.prev_quad[ln] = (uint32_t)q;
	}
	return active_next;
}

// One bounce of the wavefront: thread t serves the t-th entry of the live-ray list the previous
// bounce wrote, runs the loop body, and the survivors of a workgroup append themselves to the
// next list with one atomic per workgroup (order inside a workgroup is kept, so neighbouring
// pixels stay neighbours; the order of workgroups is free -- every lane's result depends on its
// own state only).
// (Scenes with triangle meshes do not come here: pg_render_wave.hip splits their bounce into a
// ray-casting, a shading and an SD-tree kernel.)
template <bool kFirst, int kGeneral>
__global__ __launch_bounds__(kRBlock) void k_bounce(RenderArgs a)
{
	__shared__ float s_planes[3 * kKdGridPlanes];
	const uint4 *s_kd = reinterpret_cast<const uint4 *>(s_planes); // (bounce_lane's parameter: the staged table of whichever kind)
	__shared__ uint32_t s_wave[kRBlock / 64];
	__shared__ uint32_t s_base;
	__shared__ float s_stash[kBounceStash][kRBlock];
	const uint64_t tid = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	const uint64_t live = kFirst ? a.n_lanes : (uint64_t)live_final(a, a.bounce - 1);
	if ((uint64_t)blockIdx.x * kRBlock >= live) return; // whole workgroup past the list
	if (!kFirst && tail_took_over(a, a.bounce)) return;  // a tail launch is finishing these paths
	if (a.guided || a.record) stage_kd_planes(s_planes, a.tree); // (a recorded vertex descends the KD tree too: its accumulators)
	const bool alive = tid < live;
	const uint64_t lane = alive ? (kFirst ? tid : (uint64_t)a.order_in[tid]) : 0;
	// records of the earlier bounces: all paths for the first, the survivors of bounce j for bounce j+1
	uint64_t rec_base = 0;
	if (!kFirst) {
		rec_base = a.n_lanes;
		for (int j = 0; j + 1 < a.bounce; ++j) rec_base += live_final(a, j);
	}
	bool cont = false;
	if (alive) cont = bounce_lane<kFirst, kGeneral, true>(a, s_kd, lane, rec_base + tid, (uint32_t)a.bounce, (LdsFloat *)&s_stash[0][threadIdx.x]);
	if (a.last) return; // nothing survives the last bounce
	const unsigned long long ballot = __ballot(cont);
	const unsigned wl = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	if (wl == 0) s_wave[wv] = (uint32_t)__popcll(ballot);
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tot = 0;
		for (int w = 0; w < kRBlock / 64; ++w) tot += s_wave[w];
		s_base = tot ? atomicAdd(&a.live_count[a.bounce], tot) : 0u;
	}
	__syncthreads();
	if (cont) {
		uint32_t off = s_base + (uint32_t)__popcll(ballot & ((1ull << wl) - 1ull));
		for (unsigned w = 0; w < wv; ++w) off += s_wave[w];
		a.order_out[off] = (uint32_t)lane;
	}
}

// See tail_checkpoint: launched before the per-bounce launch of bounce a.bounce with a grid for
// kTailPaths lanes.  The first bounce done here reads the live list like k_bounce would; after it a
// lane keeps its path (the state still goes through its global slots: the same thread reads back
// what it wrote).  Record entries of the later bounces are handed out wave by wave behind the
// entries of bounce a.bounce (a.live_count[max_depth] counts them), and the survivors of every
// bounce are added to live_count[] as the per-bounce launches would have, so the splat finds
// N + sum(live_count) entries and pg_render_live_counts reports the same numbers either way.
template <int kGeneral>
__global__ __launch_bounds__(kRBlock) void k_bounce_tail(RenderArgs a)
{
	__shared__ float s_planes[3 * kKdGridPlanes];
	const uint4 *s_kd = reinterpret_cast<const uint4 *>(s_planes); // (bounce_lane's parameter: the staged table of whichever kind)
	const uint64_t tid = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	const uint64_t live = (uint64_t)live_final(a, a.bounce - 1);
	if (live > kTailPaths || (uint64_t)blockIdx.x * kRBlock >= live) return;
	if (tail_took_over(a, a.bounce - 1)) return; // an earlier checkpoint already did
	if (a.guided || a.record) stage_kd_planes(s_planes, a.tree); // (a recorded vertex descends the KD tree too: its accumulators)
	bool alive = tid < live;
	const uint64_t lane = alive ? (uint64_t)a.order_in[tid] : 0;
	uint64_t rec_base = a.n_lanes; // entries of the bounces before a.bounce
	for (int j = 0; j + 1 < a.bounce; ++j) rec_base += live_final(a, j);
	const uint64_t tail_base = rec_base + live; // behind the entries of bounce a.bounce
	uint64_t slot = rec_base + tid;
	const unsigned wl = threadIdx.x & 63u;
	for (int depth = a.bounce; depth < a.max_depth; ++depth) {
		if (alive) alive = bounce_lane<false, kGeneral>(a, s_kd, lane, slot, (uint32_t)depth);
		const unsigned long long ballot = __ballot(alive);
		if (ballot == 0ull) break; // (nothing survives the last bounce)
		const uint32_t n = (uint32_t)__popcll(ballot);
		uint32_t off = 0;
		if (wl == (unsigned)__builtin_ctzll(ballot)) {
			atomicAdd(&a.live_count[depth], n);
			off = atomicAdd(&a.live_count[a.max_depth], n);
		}
		off = __shfl(off, __builtin_ctzll(ballot), 64);
		slot = tail_base + off + (uint32_t)__popcll(ballot & ((1ull << wl) - 1ull));
	}
}

// the split pipeline left the radiance of lane l in Lq[l] (one 16-byte entry, written where the path ended): the output
// column L (3, N), one thread per lane
__global__ __launch_bounds__(kRBlock) void k_layout_L(const uint4 *__restrict__ Lq, float *__restrict__ L, uint64_t N)
{
	const uint64_t i = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	if (i >= N) return;
	const uint4 q = Lq[i];
	L[i] = __uint_as_float(q.x); L[N + i] = __uint_as_float(q.y); L[2 * N + i] = __uint_as_float(q.z);
}

// :400-431: valid flag and per-pixel sums, samples of a pixel added in lane order
__global__ __launch_bounds__(kRBlock) void k_finish(RenderArgs a, uint8_t *__restrict__ valid_out,
                                                    float *__restrict__ sumL, float *__restrict__ sumL2)
{
	const uint64_t pix = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	if (pix >= a.n_pixels) return;
	const uint64_t N = a.n_lanes, P = a.film_pixels;
	const uint64_t gpix = global_pixel(a, pix); // the sums are full-film arrays
	const uint64_t first = pix * (uint64_t)a.spp;
	if (valid_out)
		for (int s = 0; s < a.spp; ++s) valid_out[first + s] = a.hit0[first + s];
	if (sumL && sumL2) {
		// the running sums stay in registers; the additions and their order are those of the reference
		float s1[3], s2[3];
		for (int c = 0; c < 3; ++c) { s1[c] = sumL[c * P + gpix]; s2[c] = sumL2[c * P + gpix]; }
		for (int s = 0; s < a.spp; ++s)
			for (int c = 0; c < 3; ++c) {
				const float v = a.L[c * N + first + s];
				s1[c] = s1[c] + v;
				s2[c] = s2[c] + v * v;
			}
		for (int c = 0; c < 3; ++c) { sumL[c * P + gpix] = s1[c]; sumL2[c * P + gpix] = s2[c]; }
	}
}

// element-wise evaluation of the library's deterministic fp32 functions (pg_math_eval)
__global__ __launch_bounds__(kRBlock) void k_math_eval(int which, uint64_t n, const float *__restrict__ x,
                                                       float *__restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	if (i >= n) return;
	const float v = x[i];
	float r, c;
	switch (which) {
	case 0: r = exp_f32(v); break;
	case 1: r = log_f32(v); break;
	case 2: r = erf_f32(v); break;
	case 3: r = erfinv_f32(v); break;
	case 4: sincos_f32(v, r, c); break;
	default: sincos_f32(v, c, r); break;
	}
	out[i] = r;
}

__device__ __forceinline__ float tent1(float d)
{
	const float a = 1.0f - fabs_(d);
	return a > 0.0f ? a : 0.0f;
}

// hdrfilm with <rfilter type="tent"/>: gather form of ImageBlock::put -- pixel (x,y) collects the
// samples of its 3x3 neighbourhood (rows, then columns, then samples, ascending), so the sums have
// one order and need no atomics.  The film position of a sample is recomputed from its stream.
// Mitsuba's gaussian rfilter (hdrfilm's default, scenes/torus/scene.xml:46): stddev 0.5, radius 2
__device__ __forceinline__ float gauss1(float d)
{
	const float a = exp_f32(-2.0f * (d * d)) - exp_f32(-8.0f);
	return a > 0.0f ? a : 0.0f;
}

// kBatched (pg_film_batched): L holds spp one-sample passes traced together (pg_pass_params.batched); the film is
// developed for each of them by itself -- image s from the samples s of the neighbourhood, exactly what pg_film gives
// for the pass seed + s alone -- into out[s][3][W * H].
// With `acc` (pg_film_batched's accumulating form) the images are not written: image s is scaled and added to acc in pass
// order, acc = acc + image_s * scale in fp32 (the first one assigned when acc_set is 0) -- the running mean main.py keeps
// of an iteration's passes (:218-239), the very operations the host would make on the separate images.
template <int kFilter, bool kBatched> // kFilter: 0 tent (3x3 neighbourhood), 1 gaussian (5x5)
__global__ __launch_bounds__(kRBlock) void k_film(uint32_t seed, int spp, int W, int H,
                                                  const float *__restrict__ L, float *__restrict__ out,
                                                  uint32_t stripe_rows, uint32_t stripe_index, uint32_t stripe_count,
                                                  float *__restrict__ acc, float scale, int acc_set)
{
	constexpr int R = kFilter == 1 ? 2 : 1;
	const uint64_t npix = (uint64_t)W * (uint64_t)H, N = npix * (uint64_t)spp;
	const uint64_t o = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	if (o >= npix) return;
	const int x = (int)(o % (uint64_t)W), y = (int)(o / (uint64_t)W);
	// (pg_film_stripes: the pixels of this rank's bands only; the others keep what `out` held)
	if (stripe_count > 1u && ((uint32_t)y / stripe_rows) % stripe_count != stripe_index) return;
	const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
	for (int img = 0; img < (kBatched ? spp : 1); ++img) {
		float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, wsum = 0.0f;
		for (int ny = y - R; ny <= y + R; ++ny)
			for (int nx = x - R; nx <= x + R; ++nx) {
				if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
				const uint64_t pix = (uint64_t)ny * (uint64_t)W + (uint64_t)nx;
				for (int s = kBatched ? img : 0; s < (kBatched ? img + 1 : spp); ++s) {
					const uint64_t lane = pix * (uint64_t)spp + (uint64_t)s;
					Pcg32 rng = lane_stream(seed, spp, kBatched ? 1 : 0, pix, (uint32_t)s);
					const float jx = rng.next_f32(), jy = rng.next_f32();
					const float ddx = cx - ((float)nx + jx), ddy = cy - ((float)ny + jy);
					const float w = kFilter == 1 ? gauss1(ddx) * gauss1(ddy) : tent1(ddx) * tent1(ddy);
					a0 = a0 + w * L[lane];
					a1 = a1 + w * L[N + lane];
					a2 = a2 + w * L[2 * N + lane];
					wsum = wsum + w;
				}
			}
		const bool ok = wsum > 0.0f;
		const float r0 = ok ? a0 / wsum : 0.0f, r1 = ok ? a1 / wsum : 0.0f, r2 = ok ? a2 / wsum : 0.0f;
		if (kBatched && acc) {
			const float v0 = r0 * scale, v1 = r1 * scale, v2 = r2 * scale;
			const bool first = img == 0 && !acc_set;
			acc[o] = first ? v0 : acc[o] + v0;
			acc[npix + o] = first ? v1 : acc[npix + o] + v1;
			acc[2 * npix + o] = first ? v2 : acc[2 * npix + o] + v2;
		} else {
			float *dst = out + (uint64_t)img * 3u * npix;
			dst[o] = r0; dst[npix + o] = r1; dst[2 * npix + o] = r2;
		}
	}
}

// The same film by 16 x 16 pixel tiles, the samples' film positions made ONCE per tile.  k_film recomputes a sample's position
// (its stream's seeding and first two draws) in every pixel that looks at it -- nine times with the tent filter, 25 with the
// gaussian -- and that, not the 400 MB of radiance it reads, is where its 2.8 ms per 16-spp pass of a 1920 x 1080 film went.
// Here a workgroup first writes the positions of every sample of its tile and the filter's rim into LDS ([sample][pixel of
// the region], so that neighbouring threads read neighbouring words), then every pixel makes the very sums of k_film in their
// very order from those: the images are the same bit for bit.  Launched when the region's positions fit 64 KB of LDS
// (16 spp with either filter); k_film serves the rest.
constexpr int kFilmTile = 16;
template <int kFilter, bool kBatched>
__global__ __launch_bounds__(kFilmTile * kFilmTile) void k_film_tiled(uint32_t seed, int spp, int W, int H,
                                                                       const float *__restrict__ L, float *__restrict__ out,
                                                                       uint32_t stripe_rows, uint32_t stripe_index, uint32_t stripe_count,
                                                                       float *__restrict__ acc, float scale, int acc_set)
{
	constexpr int R = kFilter == 1 ? 2 : 1;
	constexpr int kSide = kFilmTile + 2 * R, kRegion = kSide * kSide;
	extern __shared__ float2 s_jit[]; // [spp][kRegion]
	const int tiles_x = (W + kFilmTile - 1) / kFilmTile;
	const int tx0 = (int)(blockIdx.x % (unsigned)tiles_x) * kFilmTile, ty0 = (int)(blockIdx.x / (unsigned)tiles_x) * kFilmTile;
	for (int i = (int)threadIdx.x; i < kRegion * spp; i += kFilmTile * kFilmTile) {
		const int sidx = i / kRegion, p = i % kRegion;
		const int px = tx0 - R + p % kSide, py = ty0 - R + p / kSide;
		float2 j = make_float2(0.0f, 0.0f);
		if (px >= 0 && py >= 0 && px < W && py < H) {
			Pcg32 rng = lane_stream(seed, spp, kBatched ? 1 : 0, (uint64_t)py * (uint64_t)W + (uint64_t)px, (uint32_t)sidx);
			j.x = rng.next_f32();
			j.y = rng.next_f32();
		}
		s_jit[i] = j;
	}
	__syncthreads();
	const int x = tx0 + (int)(threadIdx.x % kFilmTile), y = ty0 + (int)(threadIdx.x / kFilmTile);
	if (x >= W || y >= H) return;
	if (stripe_count > 1u && ((uint32_t)y / stripe_rows) % stripe_count != stripe_index) return;
	const uint64_t npix = (uint64_t)W * (uint64_t)H, N = npix * (uint64_t)spp;
	const uint64_t o = (uint64_t)y * (uint64_t)W + (uint64_t)x;
	const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
	for (int img = 0; img < (kBatched ? spp : 1); ++img) {
		float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, wsum = 0.0f;
		for (int ny = y - R; ny <= y + R; ++ny)
			for (int nx = x - R; nx <= x + R; ++nx) {
				if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
				const uint64_t pix = (uint64_t)ny * (uint64_t)W + (uint64_t)nx;
				const int p = (ny - (ty0 - R)) * kSide + (nx - (tx0 - R));
				for (int sidx = kBatched ? img : 0; sidx < (kBatched ? img + 1 : spp); ++sidx) {
					const uint64_t lane = pix * (uint64_t)spp + (uint64_t)sidx;
					const float2 j = s_jit[sidx * kRegion + p];
					const float ddx = cx - ((float)nx + j.x), ddy = cy - ((float)ny + j.y);
					const float w = kFilter == 1 ? gauss1(ddx) * gauss1(ddy) : tent1(ddx) * tent1(ddy);
					a0 = a0 + w * L[lane];
					a1 = a1 + w * L[N + lane];
					a2 = a2 + w * L[2 * N + lane];
					wsum = wsum + w;
				}
			}
		const bool ok = wsum > 0.0f;
		const float r0 = ok ? a0 / wsum : 0.0f, r1 = ok ? a1 / wsum : 0.0f, r2 = ok ? a2 / wsum : 0.0f;
		if (kBatched && acc) {
			const float v0 = r0 * scale, v1 = r1 * scale, v2 = r2 * scale;
			const bool first = img == 0 && !acc_set;
			acc[o] = first ? v0 : acc[o] + v0;
			acc[npix + o] = first ? v1 : acc[npix + o] + v1;
			acc[2 * npix + o] = first ? v2 : acc[2 * npix + o] + v2;
		} else {
			float *dst = out + (uint64_t)img * 3u * npix;
			dst[o] = r0; dst[npix + o] = r1; dst[2 * npix + o] = r2;
		}
	}
}

} // namespace pg

using namespace pg;

// The buffers of one pass in flight.  pg_pass_params.slot picks one of two sets, so that two passes -- issued by the
// caller on two streams -- can be on the device at once: a pass is a chain of kernels bound by different units (BVH
// walks and SD-tree queries by the vector-memory path, shading by the vector ALUs and HBM, the splat by L2 atomics),
// and two chains out of step fill each other's gaps.  Everything a pass writes is in its set, except sdTree_current
// (integer atomics: any order) and the per-pixel sums (ordered between the sets by events, pg_render_pass).
struct PassBuf {
	// mesh scenes: ray origins, the per-bounce workspace and the BVH stacks' overflow strips (pg_render_wave.hip)
	DevBuf<uint4> st[2];       // two sets of five entries per path, swapped per bounce (pg_render_wave.hip, st_load)
	DevBuf<uint64_t> inc[2];   // and of the sampler increments
	DevBuf<uint32_t> ws;
	DevBuf<uint2> bvh_ovf;
	DevBuf<uint32_t> shadow_list;
	DevBuf<float> ray_d, thr, prev_p, prev_pdf;
	DevBuf<uint32_t> prev_quad;
	DevBuf<uint8_t> hit0;
	DevBuf<uint64_t> rng_state, rng_inc;
	DevBuf<uint32_t> order[2], live_count;
	DevBuf<uint32_t> ray_of;
	DevBuf<float> r_bsdf, r_tb, r_tr, r_nee, r_wp;
	// pg_render_sort: keys of the places (written by k_wave_trace), the sorted keys, the identity, the places in sorted
	// order, the sort's temporary storage (pg_sort.hip)
	DevBuf<uint16_t> sort_key, sort_key_out;
	DevBuf<uint32_t> sort_perm;
	DevBuf<uint4> carry[2]; // the paths' 128-byte records of a sorted bounce (RenderArgs::carry_in), two sets swapped per bounce:
	                        // k_wave_shade reads the records of its bounce and writes the next bounce's in one launch
	// which bounces are worth a sort is read off the PREVIOUS pass of this set: its live counts come back to the host
	// behind the pass (pinned memory, an event), and a bounce is sorted when at least kSortMinLive of the lanes were alive
	// going into it -- a sort costs what N pairs cost however few are left (scenes/torus: 24 % after the second bounce)
	uint32_t *h_live = nullptr;       // pinned, max_depth entries
	int h_live_n = 0;
	hipEvent_t ev_live = nullptr;
	bool live_pending = false, live_known = false;
	std::vector<uint32_t> live_prev;
	uint64_t live_prev_lanes = 0;
	DevBuf<uint4> Lq;    // the split pipeline's radiance by lane (RenderArgs::Lq)
	DevBuf<char> sort_tmp;
	size_t sort_tmp_bytes = 0;
	DevBuf<uint2> r_slot;   // the list names accumulators instead of positions and directions (pg_list_records)
	DevBuf<uint32_t> r_tree;
	// pg_render_overlap: k_wave_guide beside k_wave_cast on a library-owned stream
	hipStream_t side = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	hipEvent_t ev_finish = nullptr; // recorded behind this set's k_finish: the other set's next k_finish waits for it
	bool finish_recorded = false;
	~PassBuf()
	{
		if (ev_fork) (void)hipEventDestroy(ev_fork);
		if (ev_join) (void)hipEventDestroy(ev_join);
		if (ev_finish) (void)hipEventDestroy(ev_finish);
		if (ev_live) (void)hipEventDestroy(ev_live);
		if (h_live) (void)hipHostFree(h_live);
		if (side) (void)hipStreamDestroy(side);
	}
};

// pg_render_sort: a bounce is sorted when at least 3/10 of the pass's lanes were alive going into it (last pass's counts)
constexpr uint64_t kSortMinLiveNum = 3, kSortMinLiveDen = 10;

// library-owned renderer state
struct pg_render_state {
	DevBuf<float> quads, spheres, mats, boxes, tris, dir_lights, ior, tri_normals, tri_uvs, srgb_lut;
	DevBuf<uint32_t> textures, texels;
	bool have_tri_normals = false, have_tri_uvs = false;
	float bsphere[4] = {0, 0, 0, 0};
	DevBuf<uint32_t> bvh;
	DevBuf<int32_t> emitters;
	int n_quads = 0, n_spheres = 0, n_emitters = 0, n_boxes = 0, n_bvh_nodes = 0;
	int general = 0; // feature level of the kernels to launch (0 cornell-box class, 1 veach-mis class, 2 everything)
	pg_camera cam;
	bool have_scene = false;
	bool split_always = false; // pg_render_split_pipeline: quad scenes run the split pipeline too
	int overlap = 0;           // pg_render_overlap
	int stages = 0;            // pg_render_stages
	int sort = 0;              // pg_render_sort
	PassBuf pb[2];
	int last_slot = 0;         // of the most recent pass (pg_render_live_counts)
	// optional per-kernel timing: (kind, start, stop) event triples still to be read
	bool timing_on = false;
	struct Ev { int kind; hipEvent_t a, b; };
	std::vector<Ev> events;
	pg_kernel_timing acc = {};
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

// The buffers a pass over n_lanes lanes needs (the reference allocates its numRays x max_depth record
// arrays in setup(), path_guiding_integrator.py:93): per-lane path state, the live lists, the record
// list, and for mesh scenes the ray origins, the per-bounce workspace and the BVH stacks' overflow strips.
static int ensure_pass_buffers(pg_context *ctx, int slot, uint64_t N, bool record)
{
	pg_render_state *r = ctx->render;
	PassBuf &b = r->pb[slot];
	const int D = ctx->max_depth;
	const uint64_t S = N * (uint64_t)D;
	PG_HIP(ctx, b.hit0.ensure(N));
	if (r->general < 2) {
		PG_HIP(ctx, b.ray_d.ensure(3 * N)); PG_HIP(ctx, b.thr.ensure(3 * N));
		PG_HIP(ctx, b.prev_p.ensure(3 * N)); PG_HIP(ctx, b.prev_pdf.ensure(N));
		PG_HIP(ctx, b.prev_quad.ensure(N)); PG_HIP(ctx, b.rng_state.ensure(N));
	} else {
		for (int k = 0; k < 2; ++k) { PG_HIP(ctx, b.st[k].ensure(5 * N)); PG_HIP(ctx, b.inc[k].ensure(N)); }
		PG_HIP(ctx, b.Lq.ensure(N));
		if (r->sort) { // pg_render_sort: keys, the sorted places, the paths' 128-byte records, the sort's temporary storage
			if (N > 0xfffffff0ull) return fail(ctx, PG_ERR_INVALID, "pg_render_sort: more than 2^32 lanes in one pass");
			PG_HIP(ctx, b.sort_key.ensure(N)); PG_HIP(ctx, b.sort_key_out.ensure(N)); PG_HIP(ctx, b.sort_perm.ensure(N));
			PG_HIP(ctx, b.carry[0].ensure(8 * N)); PG_HIP(ctx, b.carry[1].ensure(8 * N));
			const size_t need = sort_pairs_temp_bytes((uint32_t)N);
			if (need > b.sort_tmp_bytes) { PG_HIP(ctx, b.sort_tmp.ensure(need)); b.sort_tmp_bytes = need; }
		}
		PG_HIP(ctx, b.ws.ensure((size_t)wave_workspace_planes() * N));
		PG_HIP(ctx, b.shadow_list.ensure(N));
		// one overflow strip of the BVH stack per list position (closest-hit launches) or walking thread; a lane names its strip
		// by a 32-bit entry index (BvhStack::ovf_first)
		if ((N > kTailPaths ? N : (uint64_t)kTailPaths) * (uint64_t)kOvfStack > 0xffffffffull)
			return fail(ctx, PG_ERR_INVALID, "pg_render_pass: more lanes than the BVH stacks' overflow strips can be indexed for (2^32 / 28)");
		PG_HIP(ctx, b.bvh_ovf.ensure((size_t)kOvfStack * (N > kTailPaths ? N : kTailPaths)));
	}
	PG_HIP(ctx, b.rng_inc.ensure(N));
	PG_HIP(ctx, b.order[0].ensure(N)); PG_HIP(ctx, b.order[1].ensure(N));
	PG_HIP(ctx, b.live_count.ensure((size_t)D + 1 + 3 * (size_t)D));
	if (record) {
		PG_HIP(ctx, b.ray_of.ensure(S));
		PG_HIP(ctx, b.r_bsdf.ensure(3 * S)); PG_HIP(ctx, b.r_tb.ensure(3 * S)); PG_HIP(ctx, b.r_tr.ensure(3 * S));
		PG_HIP(ctx, b.r_wp.ensure(S));
		// (pg_list_records: 60 B per entry -- no position, no directions, one plane of radiance_nee)
		PG_HIP(ctx, b.r_nee.ensure(S));
		PG_HIP(ctx, b.r_slot.ensure(S)); PG_HIP(ctx, b.r_tree.ensure(S));
	}
	return PG_OK;
}

extern "C" {

int pg_scene_set(pg_context *ctx, uint64_t n_quads, const float *h_quads, const pg_camera *cam)
{
	pg_scene_desc d;
	d.n_quads = n_quads; d.quads = h_quads;
	d.n_spheres = 0; d.spheres = nullptr;
	d.n_materials = 0; d.materials = nullptr;
	d.n_boxes = 0; d.boxes = nullptr;
	d.n_tris = 0; d.tris = nullptr;
	d.n_bvh_nodes = 0; d.bvh = nullptr;
	d.n_dir_lights = 0; d.dir_lights = nullptr;
	d.bsphere[0] = d.bsphere[1] = d.bsphere[2] = d.bsphere[3] = 0.0f;
	d.tri_normals = nullptr;
	d.tri_uvs = nullptr;
	d.n_textures = 0; d.textures = nullptr; d.texels = nullptr; d.n_texels = 0; d.srgb_lut = nullptr;
	return pg_scene_set_ex(ctx, &d, cam);
}

int pg_scene_set_ex(pg_context *ctx, const pg_scene_desc *sc, const pg_camera *cam)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!sc || !cam) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: NULL pointer");
	const uint64_t nq = sc->n_quads, ns = sc->n_spheres, nm = sc->n_materials, nb = sc->n_boxes;
	if (nq + ns + nb + sc->n_tris == 0 || nq > 4096 || ns > 4096 || nb > 4096 || (nq && !sc->quads) || (ns && !sc->spheres) || (nb && !sc->boxes))
		return fail(ctx, PG_ERR_INVALID, "pg_scene_set: need 1..4096 quads, spheres and/or boxes");
	if ((nm && !sc->materials) || (!sc->materials && (ns || nb)))
		return fail(ctx, PG_ERR_INVALID, "pg_scene_set: spheres and boxes need a material table");
	if (cam->width <= 0 || cam->height <= 0) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: bad film size");
	// host copies: material indices checked, a material table made up for scenes that come without one,
	// the diffuse reflectance mirrored into the quads (the quad-only kernels read it there)
	std::vector<float> quads(sc->quads, sc->quads + nq * kQuadStride);
	std::vector<float> mats;
	if (sc->materials) {
		mats.assign(sc->materials, sc->materials + nm * kMaterialStride);
	} else {
		mats.assign(nq * kMaterialStride, 0.0f);
		for (uint64_t q = 0; q < nq; ++q) {
			for (int c = 0; c < 3; ++c) mats[q * kMaterialStride + 1 + c] = quads[q * kQuadStride + 16 + c];
			quads[q * kQuadStride + 22] = (float)q;
		}
	}
	const uint64_t n_mats = mats.size() / kMaterialStride;
	const uint64_t n_tex = sc->n_textures;
	if (n_tex > 65536 || (n_tex && (!sc->textures || !sc->srgb_lut))) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: textures need their table and the sRGB lookup table");
	for (uint64_t t = 0; t < n_tex; ++t) { // a texture's texels must lie inside the texel array
		const uint32_t *T = sc->textures + t * kTextureStride;
		if (T[0] == 1u) {
			if (T[1] == 0u || T[2] == 0u || T[1] > 65536u || T[2] > 65536u || !sc->texels || (uint64_t)T[3] + (uint64_t)T[1] * T[2] > sc->n_texels)
				return fail(ctx, PG_ERR_INVALID, "pg_scene_set: bitmap texture outside the texel array");
		} else if (T[0] != 2u) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: unknown texture kind");
	}
	for (uint64_t m = 0; m < n_mats; ++m) {
		const float type = mats[m * kMaterialStride];
		if (type != 0.0f && type != 1.0f && type != 2.0f && type != 3.0f && type != 4.0f)
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: unknown material type");
		const float alpha = mats[m * kMaterialStride + 4]; // > 0: Beckmann, < 0: GGX of roughness -alpha
		if ((type == 1.0f || type == 4.0f) && !(fabsf(alpha) > 0.0f && fabsf(alpha) < 3.0e38f))
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: microfacet alpha must be finite and not 0");
		if ((type == 3.0f || type == 4.0f) && !(mats[m * kMaterialStride + 5] > 0.0f))
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: dielectric index ratio must be > 0");
		const float tex = mats[m * kMaterialStride + 12];
		if (!(tex >= 0.0f && tex <= (float)n_tex) || tex != (float)(uint64_t)tex)
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: material texture index out of range");
	}
	// feature level of the kernels (see intersect()): decided by the materials the shapes USE
	int general = ns > 0 ? 1 : 0;
	auto use_material = [&](uint64_t m) {
		const float type = mats[m * kMaterialStride];
		if (type == 1.0f && general < 1) general = 1;                              // rough conductor
		if (type >= 2.0f || mats[m * kMaterialStride + 11] != 0.0f) general = 3;   // transmission, delta lobes, one-sided BSDFs
	};
	const uint64_t nd = sc->n_dir_lights;
	if (nd > 64 || (nd && !sc->dir_lights)) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: at most 64 directional lights");
	if (nd && !(sc->bsphere[3] > 0.0f)) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: directional lights need the scene's bounding sphere");
	if (nd) general = 3;
	for (uint64_t q = 0; q < nq; ++q) {
		const float mi = quads[q * kQuadStride + 22];
		if (!(mi >= 0.0f && mi < (float)n_mats) || mi != (float)(uint64_t)mi)
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: quad material index out of range");
		use_material((uint64_t)mi);
		const float *M = &mats[(uint64_t)mi * kMaterialStride];
		if (M[0] == 0.0f)
			for (int c = 0; c < 3; ++c) quads[q * kQuadStride + 16 + c] = M[1 + c];
	}
	for (uint64_t s = 0; s < ns; ++s) {
		const float mi = sc->spheres[s * kSphereStride + 4];
		if (!(mi >= 0.0f && mi < (float)n_mats) || mi != (float)(uint64_t)mi)
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: sphere material index out of range");
		use_material((uint64_t)mi);
		if (!(sc->spheres[s * kSphereStride + 3] > 0.0f)) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: sphere radius must be > 0");
	}
	for (uint64_t b = 0; b < nb; ++b) {
		const float mi = sc->boxes[b * kBoxStride + 21];
		if (!(mi >= 0.0f && mi < (float)n_mats) || mi != (float)(uint64_t)mi)
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: box material index out of range");
		use_material((uint64_t)mi);
		for (int k = 0; k < 21; ++k)
			if (!(sc->boxes[b * kBoxStride + k] == sc->boxes[b * kBoxStride + k]) || fabsf(sc->boxes[b * kBoxStride + k]) > 3.0e38f)
				return fail(ctx, PG_ERR_INVALID, "pg_scene_set: box transform is not finite");
	}
	// triangle meshes: the kernels walk the BVH with a fixed-size stack and trust it, so check it here:
	// children follow their parent (no cycles, one parent each), leaves stay inside the triangle
	// array, and no walk can have more than kLdsStack + kOvfStack siblings waiting on its stack
	const uint64_t nt = sc->n_tris, nn = sc->n_bvh_nodes;
	// (the walk addresses a node by a 32-bit byte offset: 2^25 nodes of 128 bytes)
	if ((nt == 0) != (nn == 0) || (nt && (!sc->tris || !sc->bvh)) || nt > 0x0fffffffull || nn > 0x02000000ull)
		return fail(ctx, PG_ERR_INVALID, "pg_scene_set: triangles and BVH nodes go together");
	if (nt && !sc->materials) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: meshes need a material table");
	if (nn) {
		std::vector<uint8_t> waiting(nn, 0); // siblings on the stack when the walk opens node i, at most
		std::vector<uint8_t> seen(nn, 0);
		seen[0] = 1;
		for (uint64_t i = 0; i < nn; ++i) {
			if (!seen[i]) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH node without a parent");
			const uint32_t *N = sc->bvh + i * kBvhStride;
			int kids = 0;
			for (int c = 0; c < 4; ++c) kids += N[24 + c] != 0xffffffffu;
			if (kids == 0) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH node without children");
			const int below = (int)waiting[i] + kids - 1; // its other children wait while the walk is in one of them
			if (below > kMinLdsStack + kOvfStack) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH too deep for the walk's stack (32 waiting siblings)");
			for (int c = 0; c < 4; ++c) {
				const uint32_t ref = N[24 + c];
				if (ref == 0xffffffffu) continue;
				if (ref & 0x80000000u) {
					const uint64_t first = ref & 0x0fffffffu, count = ((ref >> 28) & 7u) + 1u;
					if (first + count > nt) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH leaf outside the triangle array");
				} else {
					if (ref <= i || ref >= nn) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH children must follow their parent");
					if (seen[ref]) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH node with two parents");
					seen[ref] = 1;
					waiting[ref] = (uint8_t)below;
				}
			}
		}
		for (uint64_t t = 0; t < nt; ++t) {
			const float mi = sc->tris[t * kTriStride + 12];
			if (!(mi >= 0.0f && mi < (float)n_mats) || mi != (float)(uint64_t)mi)
				return fail(ctx, PG_ERR_INVALID, "pg_scene_set: triangle material index out of range");
			use_material((uint64_t)mi);
		}
		if (general < 2) general = 2;
	}
	// (a textured material on anything but a triangle with texture coordinates keeps its plain colour)
	std::vector<int32_t> em;
	for (uint64_t q = 0; q < nq; ++q)
		if (quads[q * kQuadStride + 15] != 0.0f) em.push_back((int32_t)q);
	for (uint64_t s = 0; s < ns; ++s)
		if (sc->spheres[s * kSphereStride + 5] != 0.0f) em.push_back((int32_t)(nq + s));
	for (uint64_t k = 0; k < nd; ++k) em.push_back(-1 - (int32_t)k);
	PG_HIP(ctx, hipSetDevice(ctx->device));
	pg_render_state *r = rstate(ctx);
	PG_HIP(ctx, r->dir_lights.ensure(nd * 8));
	if (nd) PG_HIP(ctx, hipMemcpy(r->dir_lights.p, sc->dir_lights, nd * 8 * sizeof(float), hipMemcpyHostToDevice));
	for (int c = 0; c < 4; ++c) r->bsphere[c] = sc->bsphere[c];
	PG_HIP(ctx, r->quads.ensure(nq * kQuadStride)); PG_HIP(ctx, r->spheres.ensure(ns * kSphereStride));
	PG_HIP(ctx, r->mats.ensure(mats.size())); PG_HIP(ctx, r->emitters.ensure(em.size()));
	PG_HIP(ctx, r->boxes.ensure(nb * kBoxStride));
	if (nb) PG_HIP(ctx, hipMemcpy(r->boxes.p, sc->boxes, nb * kBoxStride * sizeof(float), hipMemcpyHostToDevice));
	r->n_boxes = (int)nb;
	PG_HIP(ctx, r->tris.ensure(nt * kTriStride)); PG_HIP(ctx, r->bvh.ensure(nn * kBvhStride));
	if (nt) PG_HIP(ctx, hipMemcpy(r->tris.p, sc->tris, nt * kTriStride * sizeof(float), hipMemcpyHostToDevice));
	if (nn) {
		// an absent child gets a box no ray reaches, (+inf, -inf) on every axis, whatever the caller left there: the walk
		// then needs no test of the reference (bvh_node_step)
		std::vector<uint32_t> nodes(sc->bvh, sc->bvh + nn * kBvhStride);
		for (uint64_t i = 0; i < nn; ++i)
			for (int c = 0; c < 4; ++c)
				if (nodes[i * kBvhStride + 24 + c] == 0xffffffffu)
					for (int row = 0; row < 6; ++row) nodes[i * kBvhStride + row * 4 + c] = row < 3 ? 0x7f800000u : 0xff800000u;
		PG_HIP(ctx, hipMemcpy(r->bvh.p, nodes.data(), nn * kBvhStride * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	r->n_bvh_nodes = (int)nn;
	r->have_tri_normals = nt && sc->tri_normals;
	if (r->have_tri_normals) {
		PG_HIP(ctx, r->tri_normals.ensure(nt * 9));
		PG_HIP(ctx, hipMemcpy(r->tri_normals.p, sc->tri_normals, nt * 9 * sizeof(float), hipMemcpyHostToDevice));
	}
	r->have_tri_uvs = nt && sc->tri_uvs && n_tex;
	if (r->have_tri_uvs) {
		PG_HIP(ctx, r->tri_uvs.ensure(nt * 6));
		PG_HIP(ctx, hipMemcpy(r->tri_uvs.p, sc->tri_uvs, nt * 6 * sizeof(float), hipMemcpyHostToDevice));
		PG_HIP(ctx, r->textures.ensure(n_tex * kTextureStride));
		PG_HIP(ctx, hipMemcpy(r->textures.p, sc->textures, n_tex * kTextureStride * sizeof(uint32_t), hipMemcpyHostToDevice));
		PG_HIP(ctx, r->texels.ensure(sc->n_texels ? sc->n_texels : 1));
		if (sc->n_texels) PG_HIP(ctx, hipMemcpy(r->texels.p, sc->texels, sc->n_texels * sizeof(uint32_t), hipMemcpyHostToDevice));
		PG_HIP(ctx, r->srgb_lut.ensure(256));
		PG_HIP(ctx, hipMemcpy(r->srgb_lut.p, sc->srgb_lut, 256 * sizeof(float), hipMemcpyHostToDevice));
	}
	if (nq) PG_HIP(ctx, hipMemcpy(r->quads.p, quads.data(), quads.size() * sizeof(float), hipMemcpyHostToDevice));
	if (ns) PG_HIP(ctx, hipMemcpy(r->spheres.p, sc->spheres, ns * kSphereStride * sizeof(float), hipMemcpyHostToDevice));
	PG_HIP(ctx, hipMemcpy(r->mats.p, mats.data(), mats.size() * sizeof(float), hipMemcpyHostToDevice));
	if (!em.empty()) PG_HIP(ctx, hipMemcpy(r->emitters.p, em.data(), em.size() * sizeof(int32_t), hipMemcpyHostToDevice));
	r->n_quads = (int)nq;
	r->n_spheres = (int)ns;
	r->n_emitters = (int)em.size();
	if (r->split_always && general < 2) general = 2; // pg_render_split_pipeline
	r->general = general;
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
	uint64_t P = prm->pixel_count ? prm->pixel_count : film - prm->pixel_begin; // 0 = to the end
	if (prm->stripe_count > 1) { // bands of stripe_rows rows dealt round-robin: this rank's rows
		if (prm->stripe_rows == 0 || prm->stripe_index >= prm->stripe_count || prm->pixel_begin != 0)
			return fail(ctx, PG_ERR_INVALID, "pg_render_pass: bad stripe parameters");
		uint64_t rows = 0;
		for (uint64_t row = 0; row < (uint64_t)r->cam.height; ++row)
			rows += (row / prm->stripe_rows) % prm->stripe_count == prm->stripe_index;
		if (prm->pixel_count && prm->pixel_count != rows * (uint64_t)r->cam.width)
			return fail(ctx, PG_ERR_INVALID, "pg_render_pass: pixel_count does not match the stripes of this rank");
		P = rows * (uint64_t)r->cam.width;
	}
	if (P == 0) return PG_OK;
	const uint64_t N = P * (uint64_t)prm->spp;
	const int D = ctx->max_depth;
	const uint64_t S = N * (uint64_t)D;
	if (S > 0xffffffffull) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: more than 2^32 record slots in one pass");
	const bool record = !ctx->is_final;
	const bool wave = r->general >= 2; // mesh scenes: the split pipeline
	if (prm->slot < 0 || prm->slot > 1) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: slot must be 0 or 1");
	const int slot = prm->slot;
	PassBuf &b = r->pb[slot];
	r->last_slot = slot;
	if (sumL && film > ctx->num_rays)
		return fail(ctx, PG_ERR_INVALID, "pg_render_pass: sumL/sumL2 are sized by pg_setup's num_rays, which is smaller than the film");
	{
		const int rc = ensure_pass_buffers(ctx, slot, N, record);
		if (rc != PG_OK) return rc;
	}
	// counters of a pass, zeroed together: live_count[D + 1] ([D]: entries handed out by the tail launch), then for
	// mesh scenes cast_count[2 D] and shadow_count[D] of the persistent ray-casting kernels
	const size_t n_counters = (size_t)D + 1 + 3 * (size_t)D;
	PG_HIP(ctx, hipMemsetAsync(b.live_count.p, 0, n_counters * sizeof(uint32_t), s));
	RenderArgs a;
	a.tree = ctx->view();
	a.shapes.quads = r->quads.p;
	a.shapes.spheres = r->spheres.p;
	a.shapes.boxes = r->boxes.p;
	a.shapes.tris = r->tris.p;
	a.shapes.tri_normals = r->have_tri_normals ? r->tri_normals.p : nullptr;
	a.shapes.tri_uvs = r->have_tri_uvs ? r->tri_uvs.p : nullptr;
	a.shapes.textures = r->textures.p;
	a.shapes.texels = r->texels.p;
	a.shapes.srgb_lut = r->srgb_lut.p;
	a.st_in = nullptr; a.st_out = nullptr; a.inc_in = nullptr; a.inc_out = nullptr; // (set per bounce below)
	a.sort_key = nullptr; a.perm = nullptr; a.carry_in = nullptr; a.carry_out = nullptr; a.n_sort = 0;
	a.Lq = wave ? b.Lq.p : nullptr;
	// sorted bounces (pg_render_sort): from the second bounce (camera rays find neighbouring vertices by themselves) to
	// the depth at which Russian roulette thins the list out (:375: a sort costs what 33 M pairs cost however few are alive)
	const int sort_until = wave && r->sort ? (prm->rr_depth < D ? prm->rr_depth : D) : 0;
	if (sort_until > 1) {
		if (b.live_pending && hipEventQuery(b.ev_live) == hipSuccess) { // the previous pass's live counts have arrived
			b.live_prev.assign(b.h_live, b.h_live + b.h_live_n);
			b.live_pending = false;
			b.live_known = true;
		}
		if (b.live_known && (b.live_prev_lanes != N || (int)b.live_prev.size() != D)) b.live_known = false; // (another pass size)
	}
	a.ws = b.ws.p;
	a.bvh_ovf = b.bvh_ovf.p;
	a.cast_count = b.live_count.p + (D + 1);
	a.shadow_count = b.live_count.p + (D + 1) + 2 * D;
	a.shadow_list = b.shadow_list.p;
	a.shapes.bvh = r->bvh.p;
	a.shapes.n_bvh_nodes = r->n_bvh_nodes;
	a.shapes.n_quads = r->n_quads;
	a.shapes.n_spheres = r->n_spheres;
	a.shapes.n_boxes = r->n_boxes;
	a.mats = r->mats.p;
	a.emitters = r->emitters.p;
	a.n_emitters = r->n_emitters;
	a.dir_lights.lights = r->dir_lights.p;
	for (int c = 0; c < 4; ++c) a.dir_lights.bsphere[c] = r->bsphere[c];
	a.ior = r->ior.p;
	a.cam = r->cam;
	a.n_lanes = N;
	a.n_pixels = P;
	a.pixel_begin = prm->pixel_begin;
	a.stripe_rows = prm->stripe_rows; a.stripe_index = prm->stripe_index; a.stripe_count = prm->stripe_count;
	a.film_pixels = film;
	a.spp = prm->spp;
	a.max_depth = D;
	a.rr_depth = prm->rr_depth;
	a.guided = ctx->iteration > 1 ? 1 : 0; // :223, 250, 283
	a.record = record ? 1 : 0;
	a.store_nee = ctx->store_nee;
	a.frac = ctx->bsdf_fraction;
	a.seed = prm->seed;
	a.batched = prm->batched ? 1 : 0;
	a.dc = ctx->dc_on ? ctx->dc : nullptr;
	a.ph = ctx->ph_on ? ctx->ph_buf.p : nullptr;
	a.ray_d = b.ray_d.p; a.thr = b.thr.p; a.L = L_out; a.prev_p = b.prev_p.p;
	a.prev_pdf = b.prev_pdf.p; a.prev_quad = b.prev_quad.p; a.hit0 = b.hit0.p;
	a.rng_state = b.rng_state.p; a.rng_inc = b.rng_inc.p; a.live_count = b.live_count.p;
	a.ray_of = b.ray_of.p; a.r_bsdf = b.r_bsdf.p; a.r_tb = b.r_tb.p;
	a.r_tr = b.r_tr.p; a.r_nee = b.r_nee.p; a.r_wp = b.r_wp.p;
	a.r_slot = b.r_slot.p; a.r_tree = b.r_tree.p;
	const dim3 grid((unsigned)((N + kRBlock - 1) / kRBlock));
	// pg_render_stages: one shading kernel per bounce (k_wave_shade), or k_wave_shade_a with the SD-tree calls in it |
	// k_wave_cast | k_wave_shade_b, or those with k_wave_guide on its own -- which pg_render_overlap needs, to run it beside
	// the shadow rays
	const int stages = (r->overlap & 1) ? 2 : r->stages;
	a.fuse_guide = wave && stages < 2 ? 1 : 0;
	const bool joint = wave && stages == 0;
	for (int it = 0; it < D; ++it) {
		a.bounce = it;
		a.last = it + 1 == D ? 1 : 0;
		a.order_in = b.order[it & 1].p;
		a.order_out = b.order[(it + 1) & 1].p;
		if (wave) { // pg_render_wave.hip: the kernels of a bounce (pg_render_stages), each timed on its own (kinds 5-9; 10 = tail; k_wave_shade: kind 6)
			// the state set this bounce reads and the one its survivors are written to; the camera rays of the first
			// launch go to the set the first bounce reads
			a.st_in = b.st[it & 1].p; a.inc_in = b.inc[it & 1].p;
			a.st_out = b.st[(it + 1) & 1].p; a.inc_out = b.inc[(it + 1) & 1].p;
			// (the choice changes no result: without counts of a previous pass every bounce below rr_depth is sorted)
			auto worth_sorting = [&](int bounce) {
				if (bounce < 1 || bounce >= sort_until) return false;
				return !b.live_known || (uint64_t)b.live_prev[bounce - 1] * kSortMinLiveDen >= N * kSortMinLiveNum;
			};
			if (tail_checkpoint(it, D)) {
				Timed t(r, s, 10);
				// (the state of a sorted bounce is in the paths' records: the tail launch reads it there)
				a.sort_key = nullptr; a.perm = nullptr; a.carry_out = nullptr;
				a.carry_in = worth_sorting(it) ? b.carry[it & 1].p : nullptr;
				launch_wave_stage(5, r->general, false, a, (unsigned)((kTailPaths + kRBlock - 1) / kRBlock), (unsigned)ctx->n_cus, s);
			}
			const bool sorted = worth_sorting(it), next_sorted = worth_sorting(it + 1);
			a.sort_key = sorted ? b.sort_key.p : nullptr;
			a.carry_in = sorted ? b.carry[it & 1].p : nullptr;
			a.carry_out = next_sorted ? b.carry[(it + 1) & 1].p : nullptr;
			a.perm = nullptr;
			if (sorted) { // closest hits in list order (they write every live place's key), the sort, then everything else at k
				// the sort covers the first n_sort places: all N without counts of a previous pass, else a little more than
				// were alive then (a radix sort costs what its n costs).  Places beyond n_sort -- none, unless this pass keeps
				// more paths alive than the last -- are served in list order (k_wave_shade_a): correct either way.
				uint64_t n_sort = N;
				if (b.live_known) {
					n_sort = (uint64_t)b.live_prev[it - 1] + (uint64_t)b.live_prev[it - 1] / 32 + 65536;
					if (n_sort > N) n_sort = N;
				}
				a.n_sort = (uint32_t)n_sort;
				PG_HIP(ctx, hipMemsetAsync(b.sort_key.p, 0xff, n_sort * sizeof(uint16_t), s)); // (0xffff: a place without a path)
				{
					Timed t(r, s, 5);
					launch_wave_stage(0, r->general, false, a, grid.x, (unsigned)ctx->n_cus, s);
				}
				{
					Timed t(r, s, 11);
					PG_HIP(ctx, sort_places16(b.sort_tmp.p, b.sort_tmp_bytes, b.sort_key.p, b.sort_key_out.p, b.sort_perm.p, (uint32_t)n_sort,
					                            a.live_count + (it - 1), s));
				}
				a.perm = b.sort_perm.p;
				if (joint) {
					Timed t(r, s, 6);
					launch_wave_stage(6, r->general, false, a, grid.x, (unsigned)ctx->n_cus, s);
					continue;
				}
				for (int stage = 1; stage < 5; ++stage) {
					if (stage == 3 && a.fuse_guide) continue;
					Timed t(r, s, 5 + stage);
					launch_wave_stage(stage, r->general, false, a, grid.x, (unsigned)ctx->n_cus, s);
				}
				continue;
			}
			if (r->overlap & 1) {
				if (!b.side) {
					PG_HIP(ctx, hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking));
					PG_HIP(ctx, hipEventCreateWithFlags(&b.ev_fork, hipEventDisableTiming));
					PG_HIP(ctx, hipEventCreateWithFlags(&b.ev_join, hipEventDisableTiming));
				}
				// the SD-tree queries (stage 3) and the shadow rays (stage 2) both read what k_wave_shade_a left and
				// write planes of their own: one is bound by divergent gathers into the tree, the other by the BVH
				// walk's dependent loads -- side by side they fill each other's stalls
				for (int stage = 0; stage < 2; ++stage) {
					Timed t(r, s, 5 + stage);
					launch_wave_stage(stage, r->general, it == 0, a, grid.x, (unsigned)ctx->n_cus, s);
				}
				PG_HIP(ctx, hipEventRecord(b.ev_fork, s));
				PG_HIP(ctx, hipStreamWaitEvent(b.side, b.ev_fork, 0));
				{
					Timed t(r, b.side, 8);
					launch_wave_stage(3, r->general, it == 0, a, grid.x, (unsigned)ctx->n_cus, b.side);
				}
				PG_HIP(ctx, hipEventRecord(b.ev_join, b.side));
				{
					Timed t(r, s, 7);
					launch_wave_stage(2, r->general, it == 0, a, grid.x, (unsigned)ctx->n_cus, s);
				}
				PG_HIP(ctx, hipStreamWaitEvent(s, b.ev_join, 0));
				Timed t(r, s, 9);
				launch_wave_stage(4, r->general, it == 0, a, grid.x, (unsigned)ctx->n_cus, s);
				continue;
			}
			if (joint) {
				{
					Timed t(r, s, 5);
					launch_wave_stage(0, r->general, it == 0, a, grid.x, (unsigned)ctx->n_cus, s);
				}
				Timed t(r, s, 6);
				launch_wave_stage(6, r->general, it == 0, a, grid.x, (unsigned)ctx->n_cus, s);
				continue;
			}
			for (int stage = 0; stage < 5; ++stage) {
				if (stage == 3 && a.fuse_guide) continue;
				Timed t(r, s, 5 + stage);
				launch_wave_stage(stage, r->general, it == 0, a, grid.x, (unsigned)ctx->n_cus, s);
			}
			continue;
		}
		Timed t(r, s, 1);
		if (tail_checkpoint(it, D)) { // finishes every path in one launch once few are left (see k_bounce_tail)
			const dim3 tgrid((unsigned)((kTailPaths + kRBlock - 1) / kRBlock));
			if (r->general == 1) hipLaunchKernelGGL((k_bounce_tail<1>), tgrid, dim3(kRBlock), 0, s, a);
			else hipLaunchKernelGGL((k_bounce_tail<0>), tgrid, dim3(kRBlock), 0, s, a);
		}
		// every launch is sized for the whole wavefront: the live count is only known on the device,
		// and workgroups past it retire on their first instruction
		if (r->general == 1) {
			if (it == 0) hipLaunchKernelGGL((k_bounce<true, 1>), grid, dim3(kRBlock), 0, s, a);
			else hipLaunchKernelGGL((k_bounce<false, 1>), grid, dim3(kRBlock), 0, s, a);
		} else {
			if (it == 0) hipLaunchKernelGGL((k_bounce<true, 0>), grid, dim3(kRBlock), 0, s, a);
			else hipLaunchKernelGGL((k_bounce<false, 0>), grid, dim3(kRBlock), 0, s, a);
		}
	}
	PG_HIP(ctx, hipGetLastError());
	if (record) {
		Timed t(r, s, 2);
		// the depth counters of an instrumented pass describe the bounce kernels only
		pg_list_records lr;
		lr.ray_of = b.ray_of.p; lr.bsdf = b.r_bsdf.p; lr.throughput_bsdf = b.r_tb.p; lr.throughput_radiance = b.r_tr.p;
		lr.nee_lum = b.r_nee.p; lr.wo_pdf = b.r_wp.p; lr.slot = b.r_slot.p; lr.tree = b.r_tree.p;
		launch_splat_list(ctx->view(), ctx->f.accum_view(), ctx->store_nee, N, D, L_out, a.Lq, lr, b.live_count.p, s);
		PG_HIP(ctx, hipGetLastError());
	}
	if (wave) {
		Timed t(r, s, 3);
		hipLaunchKernelGGL(k_layout_L, grid, dim3(kRBlock), 0, s, b.Lq.p, L_out, N);
	}
	if (valid_out || sumL) {
		// the per-pixel sums are read, added to and written back (fp32: the order of the passes is part of the result,
		// :400-429): a pass of the other buffer set that was issued before this one finishes its sums first
		PassBuf &other = r->pb[1 - slot];
		if (sumL && other.finish_recorded) PG_HIP(ctx, hipStreamWaitEvent(s, other.ev_finish, 0));
		Timed t(r, s, 3);
		hipLaunchKernelGGL(k_finish, dim3((unsigned)((P + kRBlock - 1) / kRBlock)), dim3(kRBlock), 0, s, a, valid_out,
		                   sumL, sumL2);
		PG_HIP(ctx, hipGetLastError());
		if (sumL) {
			if (!b.ev_finish) PG_HIP(ctx, hipEventCreateWithFlags(&b.ev_finish, hipEventDisableTiming));
			PG_HIP(ctx, hipEventRecord(b.ev_finish, s));
			b.finish_recorded = true;
		}
	}
	if (sort_until > 1) { // this pass's live counts, for the next pass of this set to decide by
		if (!b.h_live || b.h_live_n != D) {
			if (b.h_live) (void)hipHostFree(b.h_live);
			b.h_live = nullptr;
			PG_HIP(ctx, hipHostMalloc((void **)&b.h_live, (size_t)D * sizeof(uint32_t)));
			b.h_live_n = D;
		}
		if (!b.ev_live) PG_HIP(ctx, hipEventCreateWithFlags(&b.ev_live, hipEventDisableTiming));
		if (!b.live_pending) {
			PG_HIP(ctx, hipMemcpyAsync(b.h_live, b.live_count.p, (size_t)D * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
			PG_HIP(ctx, hipEventRecord(b.ev_live, s));
			b.live_pending = true;
			b.live_prev_lanes = N;
		}
	}
	if (r->timing_on) ++r->acc.passes;
	return PG_OK;
}

int pg_render_split_pipeline(pg_context *ctx, int32_t on)
{
	if (!ctx) return PG_ERR_INVALID;
	rstate(ctx)->split_always = on != 0;
	return PG_OK;
}

int pg_render_overlap(pg_context *ctx, int32_t mode)
{
	if (!ctx) return PG_ERR_INVALID;
	if (mode < 0 || mode > 1) return fail(ctx, PG_ERR_INVALID, "pg_render_overlap: mode must be 0 or 1");
	rstate(ctx)->overlap = mode; // (the side stream of a buffer set is made by the first pass that needs it)
	return PG_OK;
}

int pg_render_stages(pg_context *ctx, int32_t mode)
{
	if (!ctx) return PG_ERR_INVALID;
	if (mode < 0 || mode > 2) return fail(ctx, PG_ERR_INVALID, "pg_render_stages: mode must be 0, 1 or 2");
	rstate(ctx)->stages = mode;
	return PG_OK;
}

int pg_render_sort(pg_context *ctx, int32_t on)
{
	if (!ctx) return PG_ERR_INVALID;
	rstate(ctx)->sort = on != 0;
	return PG_OK;
}

int pg_render_reserve(pg_context *ctx, uint64_t n_lanes)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!ctx->configured) return fail(ctx, PG_ERR_INVALID, "call pg_setup or pg_import first");
	if (!ctx->render || !ctx->render->have_scene) return fail(ctx, PG_ERR_INVALID, "pg_render_reserve: call pg_scene_set first");
	if (ctx->max_depth <= 0 || n_lanes == 0) return fail(ctx, PG_ERR_INVALID, "pg_render_reserve: max_depth and n_lanes must be > 0");
	if (n_lanes * (uint64_t)ctx->max_depth > 0xffffffffull) return fail(ctx, PG_ERR_INVALID, "pg_render_reserve: more than 2^32 record slots in one pass");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	// The pass buffers first: they are what this call is for, and what a pass cannot run without.
	const int rc = ensure_pass_buffers(ctx, 0, n_lanes, true); // (a second set, pg_pass_params.slot 1, is allocated by its first pass)
	if (rc != PG_OK) return rc;
	// Then, where memory is plentiful, the quadtree jump tables' buffer at its budget: while the SD-tree of a real render is
	// trained the tables grow towards it (64 KB per quadtree, 1.4 GB at the veach-ajar bench tree), and a fresh allocation of
	// hundreds of megabytes in the middle of a refine is the one thing that makes a 3 ms refine take 25 or 84 ms on some boxes
	// (bench.py's exchange_refine_ms on a slow-to-allocate box, round 5).  It is an OPTIMISATION of allocation time only -- the
	// resolution a forest gets is a function of the forest and the budget alone (pg_refine.hip), with or without it -- so it is
	// taken only
	//   * for a pass of a real render (>= 2^20 lanes) and unless $PGSD_JUMP_TABLE_RESERVE is 0,
	//   * AFTER the pass buffers, and only while the device still has four times the budget free: a context on a crowded
	//     device (ranks of a rehearsal sharing one GPU, a small-memory configuration) leaves the memory alone and its refines
	//     allocate what their forests need, as before round 5,
	//   * without ever failing the call: a refused allocation changes nothing.
	const char *rsv = getenv("PGSD_JUMP_TABLE_RESERVE");
	const bool want_reserve = !(rsv && rsv[0] == '0');
	if (want_reserve && n_lanes >= (1ull << 20) && ctx->f.jump.cap * sizeof(QuadJump) < ctx->jump_budget) {
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= 4ull * ctx->jump_budget) {
			DevBuf<QuadJump> big;
			if (big.ensure((size_t)(ctx->jump_budget / sizeof(QuadJump))) == hipSuccess) {
				const size_t in_use = ctx->f.jump_valid ? ((size_t)ctx->f.n_trees << (2 * ctx->f.jump_bits)) : 0;
				// (synchronous copy + device synchronisation: nothing in flight reads the old table when it is freed with `big`)
				if ((in_use == 0 || hipMemcpy(big.p, ctx->f.jump.p, in_use * sizeof(QuadJump), hipMemcpyDeviceToDevice) == hipSuccess) &&
				    hipDeviceSynchronize() == hipSuccess)
					ctx->f.jump.swap(big);
				else (void)hipGetLastError();
			} else (void)hipGetLastError();
		} else (void)hipGetLastError();
	}
	return PG_OK;
}

int pg_film_tent(pg_context *ctx, uint32_t seed, int32_t spp, const float *L, float *image_out, void *stream)
{
	return pg_film(ctx, PG_FILTER_TENT, seed, spp, L, image_out, stream);
}

int pg_film(pg_context *ctx, int32_t filter, uint32_t seed, int32_t spp, const float *L, float *image_out, void *stream)
{
	return pg_film_stripes(ctx, filter, seed, spp, L, image_out, 0, 0, 0, stream);
}

static int film_launch(pg_context *ctx, int32_t filter, bool batched, uint32_t seed, int32_t spp, const float *L, float *image_out,
                       uint32_t stripe_rows, uint32_t stripe_index, uint32_t stripe_count, void *stream, float *acc = nullptr,
                       float scale = 1.0f, int acc_set = 0)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!ctx->render || !ctx->render->have_scene) return fail(ctx, PG_ERR_INVALID, "pg_film: call pg_scene_set first");
	if (!L || (!image_out && !acc) || spp <= 0) return fail(ctx, PG_ERR_INVALID, "pg_film: NULL pointer or spp <= 0");
	if (stripe_count > 1 && (stripe_rows == 0 || stripe_index >= stripe_count)) return fail(ctx, PG_ERR_INVALID, "pg_film_stripes: bad stripe parameters");
	if (filter != PG_FILTER_TENT && filter != PG_FILTER_GAUSSIAN) return fail(ctx, PG_ERR_INVALID, "pg_film: unknown filter");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	const pg_camera &cam = ctx->render->cam;
	const uint64_t npix = (uint64_t)cam.width * (uint64_t)cam.height;
	const dim3 grid((unsigned)((npix + kRBlock - 1) / kRBlock)), block(kRBlock);
	hipStream_t st = (hipStream_t)stream;
	// by tiles with the samples' film positions staged in LDS where those fit (k_film_tiled), else pixel by pixel
	const int side = kFilmTile + 2 * (filter == PG_FILTER_GAUSSIAN ? 2 : 1);
	const size_t lds = (size_t)side * side * (size_t)spp * sizeof(float2);
	static const bool no_tiles = getenv("PGSD_FILM_TILES") && atoi(getenv("PGSD_FILM_TILES")) == 0; // (A/B switch)
	if (lds <= 64 * 1024 && !no_tiles) {
		const dim3 tgrid((unsigned)(((cam.width + kFilmTile - 1) / kFilmTile) * ((cam.height + kFilmTile - 1) / kFilmTile))), tblock(kFilmTile * kFilmTile);
#define PG_FILM_T(F, B)                                                                                                              \
	do {                                                                                                                             \
		if (lds > 48 * 1024)                                                                                                         \
			PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_film_tiled<F, B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
		hipLaunchKernelGGL((k_film_tiled<F, B>), tgrid, tblock, lds, st, seed, spp, cam.width, cam.height, L, image_out, stripe_rows, stripe_index, \
		                   stripe_count, acc, scale, acc_set);                                                                        \
	} while (0)
		if (filter == PG_FILTER_GAUSSIAN) { if (batched) PG_FILM_T(1, true); else PG_FILM_T(1, false); }
		else { if (batched) PG_FILM_T(0, true); else PG_FILM_T(0, false); }
#undef PG_FILM_T
		PG_HIP(ctx, hipGetLastError());
		return PG_OK;
	}
#define PG_FILM(F, B) hipLaunchKernelGGL((k_film<F, B>), grid, block, 0, st, seed, spp, cam.width, cam.height, L, image_out, stripe_rows, stripe_index, stripe_count, acc, scale, acc_set)
	if (filter == PG_FILTER_GAUSSIAN) { if (batched) PG_FILM(1, true); else PG_FILM(1, false); }
	else { if (batched) PG_FILM(0, true); else PG_FILM(0, false); }
#undef PG_FILM
	PG_HIP(ctx, hipGetLastError());
	return PG_OK;
}

int pg_film_stripes(pg_context *ctx, int32_t filter, uint32_t seed, int32_t spp, const float *L, float *image_out,
                    uint32_t stripe_rows, uint32_t stripe_index, uint32_t stripe_count, void *stream)
{
	return film_launch(ctx, filter, false, seed, spp, L, image_out, stripe_rows, stripe_index, stripe_count, stream);
}

int pg_film_batched(pg_context *ctx, int32_t filter, uint32_t seed, int32_t n_passes, const float *L, float *images_out,
                    uint32_t stripe_rows, uint32_t stripe_index, uint32_t stripe_count, void *stream)
{
	return film_launch(ctx, filter, true, seed, n_passes, L, images_out, stripe_rows, stripe_index, stripe_count, stream);
}

int pg_film_batched_accumulate(pg_context *ctx, int32_t filter, uint32_t seed, int32_t n_passes, const float *L, float *acc_io,
                               float scale, int32_t acc_has_value, uint32_t stripe_rows, uint32_t stripe_index,
                               uint32_t stripe_count, void *stream)
{
	if (!acc_io) return fail(ctx, PG_ERR_INVALID, "pg_film_batched_accumulate: NULL accumulator");
	return film_launch(ctx, filter, true, seed, n_passes, L, nullptr, stripe_rows, stripe_index, stripe_count, stream, acc_io, scale,
	                   acc_has_value ? 1 : 0);
}

int pg_math_eval(pg_context *ctx, int32_t which, uint64_t n, const float *x, float *out, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	if (which < 0 || which > 5 || (n && (!x || !out))) return fail(ctx, PG_ERR_INVALID, "pg_math_eval: bad arguments");
	if (n == 0) return PG_OK;
	PG_HIP(ctx, hipSetDevice(ctx->device));
	hipLaunchKernelGGL(k_math_eval, dim3((unsigned)((n + kRBlock - 1) / kRBlock)), dim3(kRBlock), 0, (hipStream_t)stream,
	                   which, n, x, out);
	PG_HIP(ctx, hipGetLastError());
	return PG_OK;
}

int pg_render_live_counts(pg_context *ctx, uint32_t *out, int32_t n)
{
	if (!ctx || !out || n < 0) return PG_ERR_INVALID;
	if (!ctx->render || !ctx->render->pb[ctx->render->last_slot].live_count.p) return fail(ctx, PG_ERR_INVALID, "pg_render_live_counts: no pass rendered yet");
	PassBuf &b = ctx->render->pb[ctx->render->last_slot];
	PG_HIP(ctx, hipSetDevice(ctx->device));
	PG_HIP(ctx, hipDeviceSynchronize());
	int m = n < ctx->max_depth ? n : ctx->max_depth;
	if ((size_t)m > b.live_count.cap) m = (int)b.live_count.cap;
	PG_HIP(ctx, hipMemcpy(out, b.live_count.p, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToHost));
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
		case 5: r->acc.trace_ms += ms; r->acc.bounce_ms += ms; ++r->acc.trace_launches; ++r->acc.bounce_launches; break;
		case 6: r->acc.shade_ms += ms; r->acc.shade_a_ms += ms; r->acc.bounce_ms += ms; break;
		case 7: r->acc.shadow_ms += ms; r->acc.bounce_ms += ms; break;
		case 8: r->acc.guide_ms += ms; r->acc.bounce_ms += ms; ++r->acc.guide_launches; break;
		case 9: r->acc.shade_ms += ms; r->acc.shade_b_ms += ms; r->acc.bounce_ms += ms; break;
		case 10: r->acc.tail_ms += ms; r->acc.bounce_ms += ms; break;
		case 11: r->acc.sort_ms += ms; r->acc.bounce_ms += ms; break;
		case 2: r->acc.splat_ms += ms; ++r->acc.splat_launches; break;
		case 4: r->acc.compact_ms += ms; break;
		default: r->acc.finish_ms += ms; break;
		}
		(void)hipEventDestroy(e.a);
		(void)hipEventDestroy(e.b);
	}
	r->events.clear();
	*out = r->acc;
	if (reset) r->acc = pg_kernel_timing{};
	return PG_OK;
}

} // extern "C"
