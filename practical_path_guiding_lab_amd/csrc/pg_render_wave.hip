// pg_render_wave.hip -- the bounce of PathGuidingIntegrator.sample() (src/path_guiding_integrator.py:
// 179-381) for scenes with triangle meshes (scenes/veach-ajar, scenes/torus), as a wavefront pipeline.
// The stages of a bounce are device functions -- stage_a, stage_guide, stage_b, the BVH walks of
// pg_render_dev.hpp -- and pg_render_stages chooses how many kernels they are cut into behind the closest hits:
//
//   k_wave_trace    :185   scene.ray_intersect: closest hit of every live ray (BVH walk; stacks and the hottest
//                          nodes in LDS); the first launch also makes the camera rays.  Between it and what
//                          follows the live list of a sorted bounce is put in spatial order (pg_sort.hip).
//   k_wave_shade    :189-381   (the default) everything else of the bounce on one lane's registers: stage_a,
//                          stage_guide, the shadow ray as an inline any-hit walk, stage_b, the survivors' append
// or, handing their results on through the planes of `ws` (indexed by the lane's place in the live list):
//   k_wave_shade_a  :189-220, 272-297   stage_a: surface and textures at the hit, emitted radiance and its MIS
//                          weight, one emitter sample (a shadow ray to trace), the BSDF towards it, the
//                          BSDF sample, the lane's class (delta | bsdf | bsdf-mis | sdtree-mis) -- and stage_guide
//                          unless that runs as
//   k_wave_guide    :244, 301, 307   stage_guide alone: the three SD-tree calls of the bounce and nothing else --
//                          one KD descent, the pdf of the emitter direction, sample-or-pdf of the continuation
//                          direction, the canonical coordinates and accumulators the record needs -- the kernel
//                          the HBM roofline of the hot path is measured on
//   k_wave_cast     :213   test_visibility: any-hit walk of the shadow rays, a persistent kernel (a lane whose
//                          ray is done takes the next one: shadow rays end after very different numbers of steps)
//   k_wave_shade_b  :247-261, 302-381   stage_b: mixture pdfs and MIS weights, radiance, the path-vertex record,
//                          throughput, Russian roulette, the next ray; survivors are appended to the
//                          next live list (one atomic per workgroup)
//   k_wave_tail     all stages in sequence for the last few thousand paths of a long pass (see tail_checkpoint)
//
// History in one paragraph (DESIGN.md 5.2 has the measurements): round 1's single kernel per bounce needed 153
// vector registers and kept its BVH stack in scratch memory; rounds 2 and 3 split it into the five kernels above,
// which is how each stage's bound was found -- address path for the walks, HBM for the planes between the shading
// kernels -- and round 3 put the stages that only handed each other numbers back into one kernel, with the closest
// hits and the sort still on their own.  What a path carries from bounce to bounce travels with its place in the
// live list (st_load below).  Arithmetic and sampler draw order are those of oracle/pg_oracle_render.c, operation
// by operation, in every form.
#include <stdio.h>
#include <stdlib.h>

#include "pg_render_dev.hpp"

namespace pg {

// ---- the workspace: planes of n_lanes 32-bit words ----
enum : int {
	WS_HIT_PRIM = 0, WS_HIT_T, WS_HIT_U, WS_HIT_V,
	WS_P, WS_NG = WS_P + 3,
	WS_FLAGS = WS_NG + 3,
	// ten planes that hold one of two things, by the lane's class: a lane that takes its next direction from
	// the tree (F_SMP_TREE) evaluates its BSDF again for that direction in k_wave_shade_b and needs the
	// shading normal (0-2), wi (3-5), the material (6) and the reflectance (7-9); every other lane keeps its
	// BSDF sample: wo (0-2), the weight (3-5), the pdf (6).  Neither reads the other's set, so they share the
	// planes -- 42 bytes less through HBM per lane and bounce than planes of their own.
	WS_U, WS_LE = WS_U + 10,
	WS_DS_D = WS_LE + 3, WS_DS_PDF = WS_DS_D + 3, WS_EM_W, WS_BV_EM = WS_EM_W + 3, WS_BP_EM = WS_BV_EM + 3,
	WS_SH_O, WS_SH_D = WS_SH_O + 3, WS_SH_T = WS_SH_D + 3,
	WS_WO_T, WS_ETA = WS_WO_T + 3, // the direction k_wave_guide sampled from the tree (F_SMP_TREE lanes); the BSDF sample's eta
	WS_RNG_LO, WS_RNG_HI,
	WS_OCC,
	WS_PDF_NEE, WS_PDF_TREE,
	// sorted bounces (pg_render_sort): k_wave_shade_a is the one kernel that reads the state through the permutation; what
	// k_wave_guide and k_wave_shade_b need of it travels on in ten planes -- throughput (3), ior word, radiance (3), lane,
	// sampler increment (2)
	WS_FWD,
	WS_COUNT = WS_FWD + 10
};

// lane classes and switches a bounce decides in stage_a
enum : uint32_t {
	F_VALID = 1u,        // the ray hit something (:185)
	F_ACTIVE_NEXT = 2u,  // depth + 1 < max_depth and valid (:208)
	F_ACTIVE_EM = 4u,    // emitter sampling happened and ds.pdf != 0 (:210, 216)
	F_NEED_SHADOW = 8u,  // em_weight holds the unoccluded value: trace the shadow ray
	F_DS_DELTA = 16u,    // the emitter sample came from a delta light
	F_DELTA = 32u,       // the BSDF sample is a delta lobe (:282)
	F_DO_MIS = 64u,      // bsdf-mis or sdtree-mis lane (:283)
	F_SMP_TREE = 128u,   // sdtree-mis: the direction comes from the SD-tree (:297)
	F_BSDF_MIS = 256u,   // bsdf-mis: BSDF direction, SD-tree pdf (:293)
	F_NEE_LIVE = 512u,   // the emitter sample can contribute: its BSDF value is not zero (see stage_a)
	F_HAS_LE = 1024u,    // emitted radiance reached the path here (Le has a non-zero bit)
};

struct HitRec {
	int prim;
	float t, u, v;
};

struct StageA {
	v3 p, n, ng, wi, refl, Le, ds_d, em_w, bv_em, sh_o, sh_d, wo, bsdf_w;
	float ds_pdf, bp_em, sh_tmax, bsdf_pdf, eta;
	int mat;
	uint32_t flags;
};

struct GuideOut {
	float nee_cx, nee_cy, wo_cx, wo_cy, pdf_nee, pdf_tree;
	v3 wo;
	// the record's accumulators in sdTree_current (kSlotNone / kSlotRoot / rec * 4 + child, pg_descent.hpp), found by
	// the walks of sdTree_prev this stage makes anyway; tree_flags = the KD leaf's quadtree, bit 31 = the vertex
	// lies inside the root box and is counted (kdtree.py:193)
	uint32_t slot_path, slot_nee, tree_flags;
};

__device__ __forceinline__ Material material_of(const RenderArgs &a, int mat, v3 refl, int level)
{
	Material m;
	const float *M = a.mats + (size_t)mat * kMaterialStride;
	m.type = (int)M[0];
	m.refl = refl;
	m.M = M;
	m.one_sided = level >= 3 && M[11] != 0.0f;
	return m;
}

// ---- :185-220 ----
// stage_a in two halves, so that a kernel can put something between them (k_wave_shade walks the shadow ray there, with
// the BSDF sample not made yet and so not alive): stage_a1 -- the surface, emitted radiance and its MIS weight, the
// emitter sample, the BSDF towards it, the shadow ray; stage_a2 -- the BSDF sample and the lane's class.  The sampler
// draws keep their order (:214 before :272, 286); stage_a = the two in sequence.
template <int kLevel>
__device__ __forceinline__ void stage_a1(const RenderArgs &a, Pcg32 &rng, v3 ray_o, v3 ray_d, v3 thr, v3 prev_p,
                                         float prev_bsdf_pdf, bool prev_delta, const HitRec &h, uint32_t depth, StageA &o)
{
	const Shapes &sh = a.shapes;
	const int D = a.max_depth;
	const bool valid = h.prim >= 0;
	Surface sf;
	sf.p = V(0, 0, 0); sf.n = V(0, 0, 1); sf.ng = V(0, 0, 1); sf.radiance = V(0, 0, 0); sf.is_em = false;
	sf.m.type = 0; sf.m.refl = V(0, 0, 0); sf.m.M = a.mats; sf.m.one_sided = false;
	if (valid) sf = surface_at<kLevel>(sh, a.mats, h.prim, ray_o, ray_d, h.t, h.u, h.v);
	const v3 p = sf.p, n = sf.n;
	const Material &mt = sf.m;
	const Frame fr = make_frame(n);
	const v3 wi = to_local(fr, V(-ray_d.x, -ray_d.y, -ray_d.z));
	const bool is_em = valid && sf.is_em;
	const float inv_em_count = 1.0f / (float)a.n_emitters; // only used when an emitter was hit
	// ---- :189-200 direct emission ----
	const v3 em_radiance = (is_em && wi.z > 0.0f) ? sf.radiance : V(0, 0, 0);
	float emitter_pdf = 0.0f;
	if (is_em && !prev_delta) emitter_pdf = emitter_hit_pdf<kLevel>(sh, h.prim, prev_p, p, n, inv_em_count);
	const float mis = mis_weight(prev_bsdf_pdf, emitter_pdf);
	o.Le = vmul(vscale(thr, mis), em_radiance);
	// ---- :207-220 emitter sampling ----
	const bool active_next = (depth + 1 < (uint32_t)D) && valid;
	bool active_em = active_next && (kLevel < 3 || material_is_smooth(mt)); // :210 BSDFFlags.Smooth
	const float e1 = rng.next_f32(), e2 = rng.next_f32(); // :214, unmasked
	bool ds_delta = false, need_shadow = false;
	o.ds_d = V(0, 0, 0); o.em_w = V(0, 0, 0); o.ds_pdf = 0.0f;
	o.sh_o = V(0, 0, 0); o.sh_d = V(0, 0, 1); o.sh_tmax = 0.0f;
	if (active_em)
		sample_emitter_ray<kLevel>(sh, a.dir_lights, a.emitters, a.n_emitters, p, sf.ng, e1, e2, o.ds_d, o.ds_pdf, o.em_w,
		                           ds_delta, need_shadow, o.sh_o, o.sh_d, o.sh_tmax);
	active_em = active_em && (o.ds_pdf != 0.0f); // :216
	const v3 wo_em = to_local(fr, o.ds_d);
	bsdf_eval_pdf<kLevel>(mt, wi, wo_em, active_em, o.bv_em, o.bp_em);
	// An emitter sample whose BSDF value is zero (the light is behind the surface) contributes
	// Lr_dir = ((thr mis_em) 0) em_weight = +0 whatever the visibility test and the SD-tree pdf of its
	// direction say -- as long as em_weight is finite, and it is unless the light point all but touches
	// the surface.  Such a lane needs neither the shadow ray nor the tree query: result-neutral, bit for
	// bit (the oracle performs both and multiplies by zero).
	const bool nee_live = active_em && !(o.bv_em.x == 0.0f && o.bv_em.y == 0.0f && o.bv_em.z == 0.0f && finite_f32(o.em_w.x) &&
	                                     finite_f32(o.em_w.y) && finite_f32(o.em_w.z));
	if (!nee_live) need_shadow = false;
	o.p = p; o.n = n; o.ng = sf.ng; o.wi = wi; o.refl = mt.refl;
	o.mat = valid ? (int)((mt.M - a.mats) / kMaterialStride) : 0;
	asm volatile("" : "+v"(o.mat)); // (the row's NUMBER from here on, one register: not the 64-bit pointer it was made of, kept for later)
	o.flags = (valid ? F_VALID : 0u) | (active_next ? F_ACTIVE_NEXT : 0u) | (active_em ? F_ACTIVE_EM : 0u) |
	          (need_shadow ? F_NEED_SHADOW : 0u) | (ds_delta ? F_DS_DELTA : 0u) | (nee_live ? F_NEE_LIVE : 0u) |
	          ((__float_as_uint(o.Le.x) | __float_as_uint(o.Le.y) | __float_as_uint(o.Le.z)) != 0u ? F_HAS_LE : 0u);
}

// ---- :272-297 next direction (the surface comes back from what stage_a1 left in `o`: the Duff frame of the shading normal and
// the material row are functions of o.n and o.mat) ----
template <int kLevel>
__device__ __forceinline__ void stage_a2(const RenderArgs &a, Pcg32 &rng, StageA &o)
{
	const float f = a.frac;
	const bool active_next = (o.flags & F_ACTIVE_NEXT) != 0u;
	Material mt = material_of(a, o.mat, o.refl, kLevel);
	if (!(o.flags & F_VALID)) { mt.type = 0; mt.one_sided = false; } // (the ray left the scene: stage_a1's placeholder surface)
	const Frame fr = make_frame(o.n);
	float s1 = 0.0f, s2x = 0.0f, s2y = 0.0f;
	if (active_next) { // next_1d (lobe choice: only the dielectrics read it), next_2d
		if (kLevel >= 3) s1 = rng.next_f32();
		else rng.skip();
		s2x = rng.next_f32();
		s2y = rng.next_f32();
	}
	v3 wo_local;
	bool delta;
	bsdf_sample<kLevel>(mt, o.wi, s1, s2x, s2y, active_next, wo_local, o.bsdf_pdf, o.bsdf_w, o.eta, delta);
	o.wo = to_world(fr, wo_local);
	const bool do_mis = active_next && !delta && a.guided; // :283
	bool pick_tree = false;
	if (active_next) pick_tree = rng.next_f32() > f; // :286
	const bool smp_tree = pick_tree && do_mis;
	const bool bsdf_mis = do_mis && !smp_tree;
	o.flags |= (delta ? F_DELTA : 0u) | (do_mis ? F_DO_MIS : 0u) | (smp_tree ? F_SMP_TREE : 0u) | (bsdf_mis ? F_BSDF_MIS : 0u);
}

template <int kLevel>
__device__ __forceinline__ void stage_a(const RenderArgs &a, Pcg32 &rng, v3 ray_o, v3 ray_d, v3 thr, v3 prev_p,
                                        float prev_bsdf_pdf, bool prev_delta, const HitRec &h, uint32_t depth, StageA &o)
{
	stage_a1<kLevel>(a, rng, ray_o, ray_d, thr, prev_p, prev_bsdf_pdf, prev_delta, h, depth, o);
	stage_a2<kLevel>(a, rng, o);
}

// ---- :244, 301, 307: the SD-tree calls of a bounce (one KD descent) and the canonical coordinates of
// the two directions (dirToCanonical feeds the pdf queries and the record, :327, 338) ----
__device__ __forceinline__ bool guide_has_work(const RenderArgs &a, uint32_t flags)
{
	const bool do_record = a.record && (flags & F_VALID);
	return do_record || ((flags & F_NEE_LIVE) && a.guided) || (flags & (F_SMP_TREE | F_BSDF_MIS));
}

__device__ __forceinline__ void stage_guide(const RenderArgs &a, const float *s_planes, Pcg32 &rng, v3 p, v3 ds_d, v3 wo_in,
                                            uint32_t flags, GuideOut &g)
{
	const bool active_sd_em = (flags & F_NEE_LIVE) && a.guided; // (a dead emitter sample's pdf would multiply zero: stage_a)
	const bool do_record = a.record && (flags & F_VALID);
	const bool smp_tree = (flags & F_SMP_TREE) != 0u, bsdf_mis = (flags & F_BSDF_MIS) != 0u;
	// a recorded vertex names its accumulators (KDTree.addDataPropagate, kdtree.py:180-225): the leaf of its path
	// direction and, when the emitter sample can carry energy, the leaf of the emitter direction
	const bool nee_slot_wanted = do_record && a.store_nee && (flags & F_NEE_LIVE);
	TreeHead head = {kNoRecord, 0.0f};
	uint32_t tree_id = 0;
	uint32_t lv;
	unsigned c_kd = 0, c_kdq = 0, c_q = 0, c_qq = 0; // descent statistics for the byte model
	g.nee_cx = 0.0f; g.nee_cy = 0.0f; g.wo_cx = 0.0f; g.wo_cy = 0.0f;
	g.pdf_nee = 1.0f; g.pdf_tree = 1.0f;
	g.wo = wo_in;
	g.slot_path = kSlotNone; g.slot_nee = kSlotNone; g.tree_flags = 0u;
	if (active_sd_em || (do_record && a.store_nee)) dir_to_canonical(ds_d.x, ds_d.y, ds_d.z, g.nee_cx, g.nee_cy);
	if (active_sd_em || smp_tree || bsdf_mis || do_record) { // one KD descent serves every query of the vertex
		KdNode leaf;
		const bool inside = inside_root(a.tree, p.x, p.y, p.z);
		kd_descend_grid(a.tree, s_planes, p.x, p.y, p.z, inside, leaf, lv);
		c_kd += lv; ++c_kdq;
		const uint2 hv = gather8(a.tree.head + leaf.tree);
		head.root_rec = hv.x;
		head.root_irr = __uint_as_float(hv.y);
		tree_id = leaf.tree; // (outside the box: node 0's stale tree, kdtree.py:224)
		g.tree_flags = tree_id | (inside ? 0x80000000u : 0u);
	}
	if (active_sd_em) { // :244
		g.pdf_nee = quad_pdf_t<true>(a.tree.rec, a.tree.jump, tree_id, head, g.nee_cx, g.nee_cy, lv, g.slot_nee);
		c_q += lv; ++c_qq;
	}
	if (smp_tree) { // :301
		float dx, dy, dz;
		quad_sample_t<true>(a.tree.rec, a.tree.jump, tree_id, head, rng, dx, dy, dz, g.pdf_tree, lv, g.slot_path);
		c_q += lv; ++c_qq;
		g.wo = V(dx, dy, dz);
	}
	if (bsdf_mis || do_record) dir_to_canonical(g.wo.x, g.wo.y, g.wo.z, g.wo_cx, g.wo_cy);
	if (bsdf_mis) { // :307
		g.pdf_tree = quad_pdf_t<true>(a.tree.rec, a.tree.jump, tree_id, head, g.wo_cx, g.wo_cy, lv, g.slot_path);
		c_q += lv; ++c_qq;
	}
	// the leaves no query has walked to (unguided iterations, delta lobes, the last vertex of a path): the two
	// walks of QuadTree.addDataPropagate (quadtree.py:443-464), in lock step
	const bool walk_path = do_record && !smp_tree && !bsdf_mis, walk_nee = nee_slot_wanted && !active_sd_em;
	if (walk_path || walk_nee) {
		LeafCursor cp = leaf_cursor(a.tree.jump, tree_id, head, g.wo_cx, g.wo_cy, walk_path);
		LeafCursor cn = leaf_cursor(a.tree.jump, tree_id, head, g.nee_cx, g.nee_cy, walk_nee);
		quad_find_leaf_slots2(a.tree.rec, cp, cn);
		if (walk_path) { g.slot_path = cursor_slot(cp); c_q += cp.levels; ++c_qq; }
		if (walk_nee) { g.slot_nee = cursor_slot(cn); c_q += cn.levels; ++c_qq; }
	}
	if (a.dc && c_kdq) { // instrumented passes only (pg_enable_depth_counters)
		atomicAdd(&a.dc->kd_levels, (unsigned long long)stat_levels(c_kd)); // (c_kd, c_q: sums of statistics words, pg_descent.hpp)
		atomicAdd(&a.dc->kd_queries, (unsigned long long)c_kdq);
		atomicAdd(&a.dc->quad_levels, (unsigned long long)stat_levels(c_q));
		atomicAdd(&a.dc->quad_queries, (unsigned long long)c_qq);
		atomicAdd(&a.dc->layout_bytes, (unsigned long long)(stat_bytes(c_kd) + stat_bytes(c_q)));
	}
}

// Streaming stores (the "nt" bit): what a kernel writes once for a LATER kernel to read -- the record list, the paths'
// records -- should not push the trees, the BVH nodes and the textures this kernel gathers from out of L2.  Measured:
// k_wave_shade 35.1 -> 34.3 ms per step.  (The same bit on the record's seven LOADS made them miss seven times:
// 35.1 -> 37.7.)
#define PG_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
// the accumulators of a recorded vertex go straight into the record list (pg_list_records, pg_kernels.hpp)
__device__ __forceinline__ void store_slots(const RenderArgs &a, uint64_t rec_slot, const GuideOut &g)
{
	PG_ST(reinterpret_cast<unsigned long long *>(a.r_slot + rec_slot), (unsigned long long)g.slot_path | ((unsigned long long)g.slot_nee << 32));
	PG_ST(a.r_tree + rec_slot, g.tree_flags);
}

// ---- :247-261, 302-381; returns whether the path continues, with its state for the next bounce in
// thr, L, ior, ray_o, ray_d, prev_pdf, delta_out ----
template <int kLevel>
__device__ __forceinline__ bool stage_b(const RenderArgs &a, Pcg32 &rng, v3 &thr, v3 &L, float &ior, const StageA &A,
                                        const GuideOut &g, bool occluded, uint64_t lane, uint64_t rec_slot, uint32_t depth,
                                        v3 &ray_o, v3 &ray_d, float &prev_pdf, bool &delta_out)
{
	const uint64_t N = a.n_lanes;
	const int D = a.max_depth;
	const float f = a.frac;
	const bool valid = (A.flags & F_VALID) != 0u;
	bool active_next = (A.flags & F_ACTIVE_NEXT) != 0u;
	const bool ds_delta = (A.flags & F_DS_DELTA) != 0u, delta = (A.flags & F_DELTA) != 0u;
	const bool do_mis = (A.flags & F_DO_MIS) != 0u, smp_tree = (A.flags & F_SMP_TREE) != 0u;
	const bool nee_live = (A.flags & F_NEE_LIVE) != 0u;
	const v3 em_weight = (occluded || !nee_live) ? V(0, 0, 0) : A.em_w;
	// ---- :223-256 NEE MIS against the mixture pdf ----
	const float pdf_diffuse = 1.0f; // :222-241
	const float sdtree_pdf_em = g.pdf_nee;
	// a lane without a live emitter sample: every factor stage_a computed for it is zero or multiplies
	// zero -- the same formula on zeros gives the same +0
	const float bp_em = nee_live ? A.bp_em : 0.0f, ds_pdf = nee_live ? A.ds_pdf : 0.0f;
	const v3 bv_em = nee_live ? A.bv_em : V(0, 0, 0);
	float surface_pdf_em = f * bp_em + ((1.0f - f) * sdtree_pdf_em) * pdf_diffuse;
	if (!a.guided) surface_pdf_em = bp_em;
	const float mis_em = (kLevel >= 3 && ds_delta && nee_live) ? 1.0f : mis_weight(ds_pdf, surface_pdf_em); // :253
	const v3 Lr_dir = vmul(vmul(vscale(thr, mis_em), bv_em), em_weight);
	const v3 Le = (A.flags & F_HAS_LE) ? A.Le : V(0, 0, 0);
	L = vadd(L, vadd(Le, Lr_dir)); // :261
	// ---- :302-311 ----
	v3 bsdf_weight = A.bsdf_w;
	float bsdf_pdf = A.bsdf_pdf;
	v3 bsdf_value = vscale(bsdf_weight, bsdf_pdf);
	float woPdf = bsdf_pdf;
	const v3 wo_world = g.wo;
	if (smp_tree) { // :302-304
		const Frame fr = make_frame(A.n);
		const v3 wo_local = to_local(fr, wo_world);
		const Material mt = material_of(a, A.mat, A.refl, kLevel);
		bsdf_eval_pdf<kLevel>(mt, A.wi, wo_local, true, bsdf_value, bsdf_pdf);
	}
	if (do_mis) { // :310-311
		woPdf = f * bsdf_pdf + (1.0f - f) * g.pdf_tree;
		bsdf_weight = vdivs(bsdf_value, woPdf);
		// deliberate deviation: 0/0 when a zero-energy tree proposes a direction below
		// the surface; the reference's throughput turns NaN there, here the path simply ends
		if (!(woPdf > 0.0f)) bsdf_weight = V(0, 0, 0);
	}
	// ---- :318-346 record (a list in visiting order, see pg_render.hip).  The list of the split pipeline holds what
	// processPathData (:434-453) needs of a vertex -- the throughputs, the BSDF weight, woPdf, the luminance of the
	// emitter sample's share -- and, instead of position and directions, the accumulators they lead to (store_slots) ----
	const bool do_record = a.record && valid;
	if (a.record) PG_ST(a.ray_of + rec_slot, valid ? (uint32_t)lane : 0xffffffffu);
	if (do_record) {
		const uint64_t S = N * (uint64_t)D;
		const uint64_t s = rec_slot;
		PG_ST(a.r_bsdf + s, bsdf_weight.x); PG_ST(a.r_bsdf + S + s, bsdf_weight.y); PG_ST(a.r_bsdf + 2 * S + s, bsdf_weight.z);
		PG_ST(a.r_tb + s, thr.x); PG_ST(a.r_tb + S + s, thr.y); PG_ST(a.r_tb + 2 * S + s, thr.z);
		PG_ST(a.r_tr + s, L.x); PG_ST(a.r_tr + S + s, L.y); PG_ST(a.r_tr + 2 * S + s, L.z);
		float nee_lum = 0.0f;
		if (a.store_nee) { // :336, and the NaN scrub + luminance of :467, 471 (the only use of the three channels)
			v3 rn = vdiv(Lr_dir, thr);
			if (rn.x != rn.x) rn.x = 0.0f;
			if (rn.y != rn.y) rn.y = 0.0f;
			if (rn.z != rn.z) rn.z = 0.0f;
			nee_lum = luminance(rn.x, rn.y, rn.z);
		}
		PG_ST(a.r_nee + s, nee_lum);
		PG_ST(a.r_wp + s, woPdf);
	}
	// ---- :352-381 advance ----
	if (kLevel >= 3) ior = ior * A.eta; // :357 (the BSDF sample's eta also when the direction came from the tree)
	thr = vmul(thr, bsdf_weight);
	const float tmax = max3(thr);
	active_next = active_next && (tmax != 0.0f);
	float rr_prob = tmax * (ior * ior);
	if (!(rr_prob < 0.95f)) rr_prob = 0.95f;
	const bool rr_active = depth >= (uint32_t)a.rr_depth;
	const float rr = rng.next_f32(); // :377, unmasked
	const bool rr_continue = rr < rr_prob;
	active_next = active_next && (!rr_active || rr_continue);
	// :352 spawn_ray: the vertex pushed off the surface along the geometric normal, towards wo
	float mag = (1.0f + max3(V(fabs_(A.p.x), fabs_(A.p.y), fabs_(A.p.z)))) * kRayEps;
	if (dot3(A.ng, wo_world) < 0.0f) mag = -mag;
	ray_o = vadd(A.p, vscale(A.ng, mag));
	ray_d = wo_world;
	prev_pdf = woPdf;
	delta_out = delta;
	return active_next;
}

// ---- lane state and workspace access ----
__device__ __forceinline__ v3 ldp(const float *b, uint64_t N, uint64_t i) { return V(b[i], b[N + i], b[2 * N + i]); }
__device__ __forceinline__ void stp(float *b, uint64_t N, uint64_t i, v3 v) { b[i] = v.x; b[N + i] = v.y; b[2 * N + i] = v.z; }
__device__ __forceinline__ float wsf(const RenderArgs &a, int plane, uint64_t i) { return __uint_as_float(a.ws[(uint64_t)plane * a.n_lanes + i]); }
__device__ __forceinline__ uint32_t wsu(const RenderArgs &a, int plane, uint64_t i) { return a.ws[(uint64_t)plane * a.n_lanes + i]; }
__device__ __forceinline__ v3 ws3(const RenderArgs &a, int plane, uint64_t i) { return V(wsf(a, plane, i), wsf(a, plane + 1, i), wsf(a, plane + 2, i)); }
__device__ __forceinline__ void wsput(const RenderArgs &a, int plane, uint64_t i, float v) { a.ws[(uint64_t)plane * a.n_lanes + i] = __float_as_uint(v); }
__device__ __forceinline__ void wsputu(const RenderArgs &a, int plane, uint64_t i, uint32_t v) { a.ws[(uint64_t)plane * a.n_lanes + i] = v; }
__device__ __forceinline__ void wsput3(const RenderArgs &a, int plane, uint64_t i, v3 v)
{
	wsput(a, plane, i, v.x); wsput(a, plane + 1, i, v.y); wsput(a, plane + 2, i, v.z);
}

// the camera ray of a lane (mi.render's sensor.sample_ray_differential: one 2-D jitter draw per sample)
__device__ __forceinline__ void camera_ray(const RenderArgs &a, uint64_t lane, Pcg32 &rng, v3 &ray_o, v3 &ray_d)
{
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
}

// which live-list entry does this thread serve, if any: false for a whole workgroup past the list or
// when a tail launch is finishing the paths (both uniform over the workgroup).  Tile = workgroup: the hardware deals workgroups
// b, b + 8, b + 16, ... to ONE of the chip's eight XCDs (MI355X_MICROARCH.md "Workgroup dispatch"), so every XCD's L2 sees every
// eighth tile of the (sorted) list.  XCD-aware assignments were measured in round 5 and are not here any more: contiguous eighths
// of the list per XCD lose 5 ms per step (the regions of a sorted list cost unequal amounts and a workgroup can only go to its own
// XCD: seven wait for the eighth), runs of 4 / 16 / 160 tiles per XCD change nothing -- these kernels' time is not in L2 capacity
// (profiles/r05/ab_xcd_tile_mapping_rejected.txt).
template <bool kFirst>
__device__ __forceinline__ bool wave_entry(const RenderArgs &a, uint64_t &tid, bool &alive)
{
	const uint64_t live = kFirst ? a.n_lanes : (uint64_t)live_final(a, a.bounce - 1);
	const uint32_t tile = blockIdx.x;
	tid = (uint64_t)tile * kRBlock + threadIdx.x;
	if ((uint64_t)tile * kRBlock >= live) return false;
	if (!kFirst && tail_took_over(a, a.bounce)) return false;
	alive = tid < live;
	return true;
}

// The state of a path TRAVELS WITH ITS PLACE IN THE LIVE LIST: the survivor that k_wave_shade_b appends at
// place `off` of the next bounce's list gets its state written to place `off` of the other state set (two sets,
// swapped per bounce), so that every kernel of the next bounce reads state[place] -- whole cache lines, no
// gather through a list of lane numbers, no dependent load before the first useful one.  Six entries per path:
//   q 0 {ray origin, sampler state low word}   q 1 {ray direction, sampler state high word}
//   q 2 {throughput, ior with its sign bit = "the previous lobe was a delta"}
//   q 3 {previous vertex, previous bsdf pdf}   q 4 {radiance so far, the path's lane (pixel * spp + s)}
//   and the sampler's increment (8 bytes)
// The radiance is written to L_out[lane] once, where the path ends.  (Round 2 kept the state in place, indexed by
// lane: scattered 16-byte accesses whose share of each cache line shrank with every bounce's survivors.)
__device__ __forceinline__ uint4 st_load(const uint4 *st, const RenderArgs &a, int q, uint64_t i) { return st[(uint64_t)q * a.n_lanes + i]; }
__device__ __forceinline__ void st_store(uint4 *st, const RenderArgs &a, int q, uint64_t i, v3 v, uint32_t w)
{
	st[(uint64_t)q * a.n_lanes + i] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), w);
}
__device__ __forceinline__ v3 st_v3(uint4 q) { return V(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z)); }
// q 2's fourth word: the ior (positive) with the delta bit in its sign
__device__ __forceinline__ uint32_t st_pack_ior(float ior, bool delta) { return __float_as_uint(ior) | (delta ? 0x80000000u : 0u); }

// The sort key of a vertex (pg_sort.hip), 16 bits: the Morton code of its cell in a 32^3 grid over the SD-tree's root box
// (15 bits) and, below it, which half of that cell along the box's longest axis (measured: 12 / 15 / 16 key bits ->
// 35.5 / 34.9 / 34.8 ms of shading per step; 18 bits cost a third radix pass and gain nothing more); 0xfffe for a ray that
// left the scene (it still has to be shaded: :189-200 does not apply, the path ends); places of the list that hold no
// path keep the 0xffff the buffer was filled with and sort behind everything.
__device__ __forceinline__ uint32_t spread5(uint32_t v) // bits 0-4 -> bits 0, 3, 6, 9, 12
{
	return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}
// The key's LOWEST bit on a guided pass: the lane's CLASS at this bounce -- whether it will take its direction from the SD-tree
// (:286 `next_1d() > bsdfSamplingFraction`) or from its BSDF.  The draw is the sixth of the bounce (:214 two, :272 three, :286),
// so the closest-hit kernel, which holds the path's sampler state, can look it up ahead of time.  Inside a cell of the sort the
// lanes then stand class by class and most waves hold ONE class: the sampling walk and the second BSDF evaluation run on full
// waves of "tree" lanes, the pdf walk on full waves of "BSDF" lanes, instead of every wave making all three for half its lanes
// each.  A hint, not a contract: where the guess is wrong (a delta lobe, :283) only the order differs, and the order is free.
__device__ __forceinline__ uint32_t lane_class_ahead(uint64_t state, uint64_t inc, float frac)
{
	Pcg32 r;
	r.state = state; r.inc = inc;
	r.skip(); r.skip(); r.skip(); r.skip(); r.skip();
	return r.next_f32() > frac ? 1u : 0u;
}

__device__ __forceinline__ uint32_t vertex_sort_key(const RenderArgs &a, v3 p, int cls = -1)
{
	uint32_t c[3];
	const float q[3] = {p.x, p.y, p.z};
	const float ext[3] = {a.tree.bmax[0] - a.tree.bmin[0], a.tree.bmax[1] - a.tree.bmin[1], a.tree.bmax[2] - a.tree.bmin[2]};
	const int longest = ext[0] >= ext[1] ? (ext[0] >= ext[2] ? 0 : 2) : (ext[1] >= ext[2] ? 1 : 2); // (uniform)
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		float f = (q[k] - a.tree.bmin[k]) / ext[k] * 64.0f;
		if (!(f > 0.0f)) f = 0.0f; // (NaN too)
		if (f > 63.0f) f = 63.0f;
		c[k] = (uint32_t)f;
	}
	const uint32_t half = cls >= 0 ? (uint32_t)cls : ((longest == 0 ? c[0] : (longest == 1 ? c[1] : c[2])) & 1u);
	// (measured: the class as the key's highest bit instead -- the list in two halves, each in cell order -- shades in the same time)
	const uint32_t m = ((spread5(c[0] >> 1) | (spread5(c[1] >> 1) << 1) | (spread5(c[2] >> 1) << 2)) << 1) | half;
	return m < 0xfffdu ? m : 0xfffdu; // (0xfffe and 0xffff mean something else)
}

// The first nodes of the BVH in LDS (bvh_node_step): every thread of the workgroup calls this.  How many: what the LDS
// leaves beside the walks' stacks at seven workgroups per compute unit (16 KB of stack + 6 KB of nodes each).  Measured on
// veach-ajar, ms per step: closest hits 12.7 / 11.8 / 11.6 with 16 / 32 / 48 nodes (15.1 with none); shadow rays 6.9 /
// 6.6 / 6.6 with 32 / 64 / 80 at six waves per SIMD (7.9 with none), 6.5 / 6.4 with 32 / 48 at seven.
constexpr int kBvhTopNodes = 48;
template <int kNodes>
__device__ __forceinline__ void stage_bvh_top(u32x4_t *s_top, const RenderArgs &a, BvhStack &stk)
{
	const uint32_t n = a.shapes.n_bvh_nodes < kNodes ? (uint32_t)a.shapes.n_bvh_nodes : (uint32_t)kNodes;
	// (every load of a thread is asked for before the first is stored: as a loop -- load, wait, store, again -- the second
	// 16 bytes of a thread began their round trip when the first had ended)
	constexpr int kLoads = (kNodes * 8 + kRBlock - 1) / kRBlock;
	if (n) { // (uniform; a scene without meshes has no table: nothing is read)
		u32x4_t v[kLoads];
#pragma unroll
		for (int k = 0; k < kLoads; ++k) { // (unconditional, so that nothing ties a load to its store: a lane past the nodes reads entry 0)
			const uint32_t i = threadIdx.x + (uint32_t)k * kRBlock;
			v[k] = reinterpret_cast<const u32x4_t *>(a.shapes.bvh)[i < n * 8u ? i : 0u];
		}
#pragma unroll
		for (int k = 0; k < kLoads; ++k) {
			const uint32_t i = threadIdx.x + (uint32_t)k * kRBlock;
			if (i < n * 8u) s_top[i] = v[k];
		}
	}
	__syncthreads();
	stk.top = (const LdsQuad *)s_top;
	stk.n_top = n;
}

// ---- :185 scene.ray_intersect: one ray per lane.  (The persistent form below was measured too: the walks of
// closest-hit rays are all about equally long, handing idle lanes new rays gains nothing after the second
// bounce and loses a factor of two on the coherent camera rays.) ----
template <int kLevel, bool kFirst>
__global__ __launch_bounds__(kRBlock) void k_wave_trace(RenderArgs a)
{
	__shared__ uint2 s_stack[kLdsStack][kRBlock];
	uint64_t tid;
	bool alive;
	if (!wave_entry<kFirst>(a, tid, alive)) return;
	// The ray is ASKED FOR ahead of the staging of the BVH's top (round 6): the staging ends in a workgroup barrier no load
	// moves across, and the ray's round trip used to begin behind it.
	const bool want_cls = !kFirst && a.carry_in && a.guided && a.bounce + 1 < a.max_depth; // (uniform)
	uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0, q5 = q0;
	if (!kFirst) {
		// a sorted bounce: the state is in the paths' 128-byte records only; else in the planes of the state set.  ONE
		// unconditional load site per entry: a load under a per-lane condition is merged with its zero behind the branch, and
		// that merge waits for the load -- so a lane past the list reads place 0's entries (the list is not empty: this
		// workgroup runs) and drops them, and the entry of the sort key's class bit is read whether or not it is used.
		// (Measured and removed, round 5: these three entries read by four lanes per record and handed over through the wave's
		// stack columns, as k_wave_shade reads its PERMUTED records -- here the wave's records lie side by side, consecutive
		// lanes ask for consecutive lines, and the detour through LDS cost 11.5 -> 12.8 ms per step:
		// profiles/r05/ab_trace_coop_record_load_rejected.txt.)
		const uint64_t it = alive ? tid : 0;
		const uint4 *p0 = a.carry_in ? a.carry_in + it * 8 : a.st_in + it;
		const uint4 *p1 = a.carry_in ? a.carry_in + it * 8 + 1 : a.st_in + a.n_lanes + it;
		const uint4 *p5 = a.carry_in ? a.carry_in + it * 8 + 5 : p0;
		q0 = *p0; q1 = *p1; q5 = *p5;
	}
	__shared__ u32x4_t s_top[kBvhTopNodes * 8];
	BvhStack stk = bvh_stack(&s_stack[0][threadIdx.x], a.bvh_ovf, (uint32_t)tid * (uint32_t)kOvfStack);
	stage_bvh_top<kBvhTopNodes>(s_top, a, stk);
	if (!alive) return;
	v3 ray_o, ray_d;
	int cls_ahead = -1; // (a sorted, guided bounce: the lane's class, for the sort key)
	if (kFirst) { // (place = lane in the first list)
		Pcg32 rng;
		camera_ray(a, tid, rng, ray_o, ray_d);
		// (this launch FILLS the set the first bounce reads: the only writer of an "in" set)
		const_cast<uint64_t *>(a.inc_in)[tid] = rng.inc;
		st_store(const_cast<uint4 *>(a.st_in), a, 0, tid, ray_o, (uint32_t)rng.state);
		st_store(const_cast<uint4 *>(a.st_in), a, 1, tid, ray_d, (uint32_t)(rng.state >> 32));
	} else {
		ray_o = st_v3(q0);
		ray_d = st_v3(q1);
		// (the class bit of the sort key is made HERE, ahead of the walk, from the sampler state in the two entries just read: one
		// register through the walk instead of the state's four)
		if (want_cls && tid < (uint64_t)a.n_sort)
			cls_ahead = (int)lane_class_ahead((uint64_t)q0.w | ((uint64_t)q1.w << 32), (uint64_t)q5.x | ((uint64_t)q5.y << 32), a.frac);
	}
	HitRec h;
	h.u = 0.0f; h.v = 0.0f;
	h.prim = intersect<kLevel, false>(a.shapes, ray_o, ray_d, __builtin_huge_valf(), h.t, stk, h.u, h.v);
	if (a.carry_in) { // a sorted bounce: the hit joins the path's 128-byte record, which k_wave_shade_a reads through the permutation
		a.carry_in[tid * 8 + 6] = make_uint4((uint32_t)h.prim, __float_as_uint(h.t), __float_as_uint(h.u), __float_as_uint(h.v));
		if (tid < (uint64_t)a.n_sort) a.sort_key[tid] = (uint16_t)(h.prim >= 0 ? vertex_sort_key(a, vadd(ray_o, vscale(ray_d, h.t)), cls_ahead) : 0xfffeu);
		return;
	}
	wsputu(a, WS_HIT_PRIM, tid, (uint32_t)h.prim);
	wsput(a, WS_HIT_T, tid, h.t);
	wsput(a, WS_HIT_U, tid, h.u);
	wsput(a, WS_HIT_V, tid, h.v);
}

// ---- :213 test_visibility ----
// Persistent: the grid is as many workgroups as the device holds at once, and a lane whose ray is done
// does not wait for the slowest ray of its wave -- as soon as kRefillIdle lanes of a wave are idle they
// take the next rays of the launch's list.  The list is dealt out in chunks of kCastChunk entries, chunk
// c to wave c mod (number of waves): no atomic counter (one counter word serialises at about 11 ns per
// atomic -- a counter bumped per refill cost more than the walks, and big chunks per atomic left the
// waves unevenly loaded at the end), and with some fifty chunks per wave the rays' costs even out.
// Shadow rays end after very different numbers of steps (an occluder may be the first thing met):
// measured on veach-ajar, the one-ray-per-lane form kept 14 % of the lanes of a wave busy.  Every ray
// still takes exactly the steps intersect() takes for it, in the same order.  The rays are the entries
// of the live list flagged F_NEED_SHADOW; WS_OCC receives 0 / 1.
constexpr int kRefillIdle = 16;
constexpr uint32_t kCastChunk = 128;
// (compiled for seven waves per SIMD -- 72 vector registers, none spilled; left alone the compiler takes 77 and six fit)
template <int kLevel, bool kFirst>
__global__ __launch_bounds__(kRBlock) __attribute__((amdgpu_waves_per_eu(7))) void k_wave_cast(RenderArgs a)
{
	__shared__ uint2 s_stack[kLdsStack][kRBlock];
	if (!kFirst && tail_took_over(a, a.bounce)) return; // a tail launch is finishing these paths
	const uint32_t total = kFirst ? (uint32_t)a.n_lanes : live_final(a, a.bounce - 1);
	const uint32_t gtid = blockIdx.x * kRBlock + threadIdx.x, n_static = gridDim.x * kRBlock;
	if (blockIdx.x * kRBlock >= total) return; // (uniform) not even a first entry for this workgroup
	const unsigned wl = threadIdx.x & 63u;
	const Shapes &sh = a.shapes;
	__shared__ u32x4_t s_top[kBvhTopNodes * 8];
	BvhStack stk = bvh_stack(&s_stack[0][threadIdx.x], a.bvh_ovf, gtid * (uint32_t)kOvfStack);
	stage_bvh_top<kBvhTopNodes>(s_top, a, stk);
	const int tri_base = sh.n_quads + sh.n_spheres + 6 * sh.n_boxes;
	BvhWalk w;
	w.next = kBvhNone; w.sp = 0; w.budget = 0; w.best = -1; w.bt = 0.0f; w.bu = 0.0f; w.bv = 0.0f;
	bool has = false, first_round = true;
	bool exhausted = n_static >= total; // no entries beyond the first round
	uint32_t pool_next = 0, pool_end = 0; // (wave-uniform) the entries of this wave's current chunk not dealt out yet
	const uint32_t n_waves = n_static / 64u;
	uint64_t chunk = gtid / 64u; // this wave's next chunk of the list beyond the first round
	uint32_t my = 0; // the entry this lane is walking the shadow ray of
	for (;;) {
		// ---- idle lanes take new rays ----
		const unsigned long long idle = __ballot(!has);
		const uint32_t n_idle = (uint32_t)__popcll(idle);
		if (first_round || (!exhausted && n_idle >= (uint32_t)kRefillIdle)) {
			uint32_t idx = 0xffffffffu;
			if (first_round) idx = gtid;
			else {
				if (pool_next == pool_end) { // this wave's next chunk of the list
					const uint64_t begin = (uint64_t)n_static + chunk * kCastChunk;
					chunk += n_waves;
					if (begin >= (uint64_t)total) exhausted = true;
					else {
						pool_next = (uint32_t)begin;
						pool_end = begin + kCastChunk < (uint64_t)total ? (uint32_t)(begin + kCastChunk) : total;
					}
				}
				const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << wl) - 1ull));
				const uint32_t take = n_idle < pool_end - pool_next ? n_idle : pool_end - pool_next;
				if (rank < take) idx = pool_next + rank;
				pool_next += take;
			}
			first_round = false;
			if (!has && idx < total && (wsu(a, WS_FLAGS, idx) & F_NEED_SHADOW)) {
				my = idx;
				const v3 ray_o = ws3(a, WS_SH_O, my), ray_d = ws3(a, WS_SH_D, my);
				int best = -1;
				float bt = wsf(a, WS_SH_T, my);
				intersect_linear<kLevel>(sh, ray_o, ray_d, bt, best);
				bvh_begin(w, sh, ray_o, ray_d, bt, best);
				has = true;
				if (!sh.n_bvh_nodes || best >= 0) { // nothing to walk: the round below finds the stack empty
					w.budget = 0;
					w.next = kBvhNone;
				}
			}
		}
		if (__ballot(has) == 0ull) {
			if (exhausted) break;
			continue; // (every lane idle and entries left: the refill above runs again)
		}
		// ---- one round of every lane's walk, the loop body of intersect(): down through nodes to a leaf
		// (the wave stays in the node loop until all its lanes have left it), the leaf's triangles, the
		// next candidate from the stack ----
		if (has) {
			bool done = false;
			while (!(w.next & 0x80000000u) && w.budget > 0) { // (see intersect)
				bvh_node_step(w, sh, stk);
				if (w.next == kBvhNone) bvh_pop(w, stk);
			}
			if (w.next != kBvhNone && (w.next & 0x80000000u)) {
				bvh_leaf_step(w, sh, tri_base);
				if (w.best >= 0) done = true; // a shadow ray needs one occluder, not the nearest
			}
			if (!done) {
				bvh_pop(w, stk);
				if (w.next == kBvhNone) done = true;
			}
			if (done) {
				wsputu(a, WS_OCC, my, w.best >= 0 ? 1u : 0u);
				has = false;
			}
		}
	}
}


// ---- :189-220, 272-297 ----
// kGuide: the SD-tree calls of the bounce (stage_guide, otherwise k_wave_guide's) follow in the same kernel
template <int kLevel, bool kFirst, bool kGuide>
__global__ __launch_bounds__(kRBlock) void k_wave_shade_a(RenderArgs a)
{
	__shared__ float s_planes[kGuide ? 3 * kKdGridPlanes : 1];
	uint64_t tid;
	bool alive;
	if (!wave_entry<kFirst>(a, tid, alive)) return;
	if (kGuide) stage_kd_planes(s_planes, a.tree);
	if (!alive) return;
	Pcg32 rng;
	v3 ray_o, ray_d, thr = V(1, 1, 1), prev_p = V(0, 0, 0);
	float prev_pdf = 1.0f;
	bool prev_delta = true;
	HitRec h;
	if (!kFirst && a.perm) {
		// a sorted bounce: this thread serves the place the sort put k-th.  The one random pass over the state: the
		// path's 128-byte record (k_wave_shade_b and k_wave_trace filled it) is one cache line; what k_wave_guide and
		// k_wave_shade_b need of it travels on at k (WS_FWD), so that nothing after this kernel looks through the permutation
		const uint4 *rec = a.carry_in + (tid < (uint64_t)a.n_sort ? (uint64_t)a.perm[tid] : tid) * 8;
		const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6];
		rng.state = (uint64_t)q0.w | ((uint64_t)q1.w << 32);
		rng.inc = (uint64_t)q5.x | ((uint64_t)q5.y << 32);
		ray_o = st_v3(q0); ray_d = st_v3(q1);
		thr = st_v3(q2);
		prev_delta = (q2.w >> 31) != 0u;
		prev_p = st_v3(q3);
		prev_pdf = __uint_as_float(q3.w);
		h.prim = (int)q6.x; h.t = __uint_as_float(q6.y); h.u = __uint_as_float(q6.z); h.v = __uint_as_float(q6.w);
		wsputu(a, WS_FWD + 0, tid, q2.x); wsputu(a, WS_FWD + 1, tid, q2.y); wsputu(a, WS_FWD + 2, tid, q2.z); wsputu(a, WS_FWD + 3, tid, q2.w);
		wsputu(a, WS_FWD + 4, tid, q4.x); wsputu(a, WS_FWD + 5, tid, q4.y); wsputu(a, WS_FWD + 6, tid, q4.z); wsputu(a, WS_FWD + 7, tid, q4.w);
		wsputu(a, WS_FWD + 8, tid, q5.x); wsputu(a, WS_FWD + 9, tid, q5.y);
	} else {
		const uint4 q0 = st_load(a.st_in, a, 0, tid), q1 = st_load(a.st_in, a, 1, tid);
		rng.state = (uint64_t)q0.w | ((uint64_t)q1.w << 32);
		rng.inc = a.inc_in[tid];
		ray_o = st_v3(q0); ray_d = st_v3(q1);
		if (!kFirst) {
			const uint4 q2 = st_load(a.st_in, a, 2, tid), q3 = st_load(a.st_in, a, 3, tid);
			thr = st_v3(q2);
			prev_delta = (q2.w >> 31) != 0u;
			prev_p = st_v3(q3);
			prev_pdf = __uint_as_float(q3.w);
		}
		h.prim = (int)wsu(a, WS_HIT_PRIM, tid);
		h.t = wsf(a, WS_HIT_T, tid); h.u = wsf(a, WS_HIT_U, tid); h.v = wsf(a, WS_HIT_V, tid);
	}
	StageA A;
	stage_a<kLevel>(a, rng, ray_o, ray_d, thr, prev_p, prev_pdf, prev_delta, h, (uint32_t)a.bounce, A);
	// (storing a value for some lanes of a wave only saves nothing by itself -- the partial store touches the same
	// sectors -- which is why the two classes of lanes share planes instead: WS_U; making every lane store so that no
	// line is written in part was measured too: no different, 84.0 vs 84.8 ms per step)
	wsput3(a, WS_P, tid, A.p); wsput3(a, WS_NG, tid, A.ng);
	wsputu(a, WS_FLAGS, tid, A.flags);
	if (A.flags & F_SMP_TREE) { // (see WS_U)
		wsput3(a, WS_U, tid, A.n); wsput3(a, WS_U + 3, tid, A.wi); wsputu(a, WS_U + 6, tid, (uint32_t)A.mat);
		wsput3(a, WS_U + 7, tid, A.refl);
	} else {
		wsput3(a, WS_U, tid, A.wo); wsput3(a, WS_U + 3, tid, A.bsdf_w); wsput(a, WS_U + 6, tid, A.bsdf_pdf);
	}
	if (A.flags & F_HAS_LE) wsput3(a, WS_LE, tid, A.Le);
	if (!kGuide) wsput3(a, WS_DS_D, tid, A.ds_d); // (only k_wave_guide reads it)
	if (A.flags & F_NEE_LIVE) { // (what k_wave_shade_b reads of an emitter sample only when it can contribute)
		wsput(a, WS_DS_PDF, tid, A.ds_pdf); wsput3(a, WS_EM_W, tid, A.em_w);
		wsput3(a, WS_BV_EM, tid, A.bv_em); wsput(a, WS_BP_EM, tid, A.bp_em);
	}
	if (A.flags & F_NEED_SHADOW) {
		wsput3(a, WS_SH_O, tid, A.sh_o); wsput3(a, WS_SH_D, tid, A.sh_d); wsput(a, WS_SH_T, tid, A.sh_tmax);
	}
	if (kLevel >= 3) wsput(a, WS_ETA, tid, A.eta); // (:357: only the dielectrics of level 3 have an eta other than 1)
	if (kGuide && guide_has_work(a, A.flags)) { // what k_wave_guide does, with its inputs still in registers
		GuideOut g;
		stage_guide(a, s_planes, rng, A.p, A.ds_d, (A.flags & F_SMP_TREE) ? V(0, 0, 0) : A.wo, A.flags, g);
		if (a.record && (A.flags & F_VALID)) {
			uint64_t rec_base = 0;
			if (!kFirst) {
				rec_base = a.n_lanes;
				for (int j = 0; j + 1 < a.bounce; ++j) rec_base += live_final(a, j);
			}
			store_slots(a, rec_base + tid, g);
		}
		wsput(a, WS_PDF_NEE, tid, g.pdf_nee); wsput(a, WS_PDF_TREE, tid, g.pdf_tree);
		if (A.flags & F_SMP_TREE) wsput3(a, WS_WO_T, tid, g.wo);
	}
	wsputu(a, WS_RNG_LO, tid, (uint32_t)rng.state); wsputu(a, WS_RNG_HI, tid, (uint32_t)(rng.state >> 32));
}

// ---- :244, 301, 307 ----
__global__ __launch_bounds__(kRBlock) void k_wave_guide(RenderArgs a)
{
	__shared__ float s_planes[3 * kKdGridPlanes];
	uint64_t tid;
	bool alive;
	if (a.bounce == 0) { if (!wave_entry<true>(a, tid, alive)) return; }
	else if (!wave_entry<false>(a, tid, alive)) return;
	// The lane's inputs are ASKED FOR first -- together with its flags and ahead of the staging of the KD planes (round 6: the
	// listing showed staging -> barrier -> flags -> wait -> branch -> inputs -> wait, three round trips at the head of every
	// wave where one does): the planes hold a value for every live place, a lane without work throws its copies away.  (The
	// empty asm behind the barrier keeps the loads where they are written; the compiler would sink them into the branch.)
	// (unconditional: a load under a per-lane condition is merged with its default behind the branch, and that merge waits for
	// the load -- a lane past the list reads place 0's values, the list is not empty, and leaves below)
	const uint64_t it = alive ? tid : 0;
	const uint32_t flags = wsu(a, WS_FLAGS, it);
	v3 p_in = ws3(a, WS_P, it), ds_in = ws3(a, WS_DS_D, it), u_in = ws3(a, WS_U, it);
	uint32_t rng_lo = wsu(a, WS_RNG_LO, it), rng_hi = wsu(a, WS_RNG_HI, it);
	stage_kd_planes(s_planes, a.tree);
	if (!alive) return;
	asm volatile("" : "+v"(p_in.x), "+v"(p_in.y), "+v"(p_in.z), "+v"(ds_in.x), "+v"(ds_in.y), "+v"(ds_in.z), "+v"(u_in.x), "+v"(u_in.y),
	                  "+v"(u_in.z), "+v"(rng_lo), "+v"(rng_hi));
	if (!guide_has_work(a, flags)) return;
	Pcg32 rng;
	rng.state = (uint64_t)rng_lo | ((uint64_t)rng_hi << 32);
	rng.inc = a.perm ? ((uint64_t)wsu(a, WS_FWD + 8, tid) | ((uint64_t)wsu(a, WS_FWD + 9, tid) << 32)) : a.inc_in[tid];
	GuideOut g;
	// (the BSDF-sampled direction of a lane that keeps it; a lane that samples the tree has none to evaluate)
	const v3 wo_in = (flags & F_SMP_TREE) ? V(0, 0, 0) : u_in;
	stage_guide(a, s_planes, rng, p_in, ds_in, wo_in, flags, g);
	if (a.record && (flags & F_VALID)) { // the entry k_wave_shade_b fills for this vertex
		uint64_t rec_base = 0;
		if (a.bounce > 0) {
			rec_base = a.n_lanes;
			for (int j = 0; j + 1 < a.bounce; ++j) rec_base += live_final(a, j);
		}
		store_slots(a, rec_base + tid, g);
	}
	wsput(a, WS_PDF_NEE, tid, g.pdf_nee); wsput(a, WS_PDF_TREE, tid, g.pdf_tree);
	if (flags & F_SMP_TREE) {
		wsput3(a, WS_WO_T, tid, g.wo);
		wsputu(a, WS_RNG_LO, tid, (uint32_t)rng.state); wsputu(a, WS_RNG_HI, tid, (uint32_t)(rng.state >> 32));
	}
}

// The survivors of a workgroup go to the next live list with their state (k_wave_shade_b, k_wave_shade): every thread of
// the workgroup calls this.  s_rec: kStage * 8 entries of LDS when the next bounce is sorted (a.carry_out), else unused.
template <int kStage>
__device__ __forceinline__ void append_survivors(const RenderArgs &a, bool cont, v3 ray_o, v3 ray_d, v3 thr, float ior, bool delta, v3 p_here,
                                                 float prev_pdf, v3 L, uint64_t lane, const Pcg32 &rng, uint32_t *s_wave, uint32_t &s_base,
                                                 uint32_t (*s_oct)[8], uint4 *s_rec)
{
	if (a.last) return; // nothing survives the last bounce
	const unsigned long long ballot = __ballot(cont);
	const unsigned wl = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	if (wl == 0) s_wave[wv] = (uint32_t)__popcll(ballot);
	// The survivors of a workgroup are appended grouped by the octant of their next direction (the order inside a
	// workgroup is free): on a sorted bounce the workgroup holds neighbours in space, and a wave of the next closest-hit
	// launch then walks the BVH from one corner in one direction -- k_wave_trace 17.5 -> 16.4 ms per step (two of the three
	// signs, so that a bin fills a wave: 16.7).  In list order the grouping changes nothing (round 2 measured that).
	const unsigned oct = (__float_as_uint(ray_d.x) >> 31) | ((__float_as_uint(ray_d.y) >> 31) << 1) | ((__float_as_uint(ray_d.z) >> 31) << 2);
	uint32_t rank_in_bin = 0;
	for (unsigned k = 0; k < 8; ++k) {
		const unsigned long long m = __ballot(cont && oct == k);
		if (wl == 0) s_oct[wv][k] = (uint32_t)__popcll(m);
		if (oct == k) rank_in_bin = (uint32_t)__popcll(m & ((1ull << wl) - 1ull));
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tot = 0;
		for (int w = 0; w < kRBlock / 64; ++w) tot += s_wave[w];
		s_base = tot ? atomicAdd(&a.live_count[a.bounce], tot) : 0u;
	}
	__syncthreads();
	uint32_t off = s_base + rank_in_bin; // the survivor's place in the next list (whole lines: a wave's survivors are neighbours)
	if (cont) {
		for (unsigned k = 0; k < oct; ++k)
			for (unsigned w = 0; w < kRBlock / 64; ++w) off += s_oct[w][k]; // (every earlier bin of the workgroup)
		for (unsigned w = 0; w < wv; ++w) off += s_oct[w][oct];           // (this bin, the earlier waves)
	}
	if (a.carry_out) {
		// the next bounce is sorted: the state goes into ONE 128-byte record per path and nowhere else -- the kernel that
		// looks through the permutation reads it as one cache line; k_wave_trace takes the ray from it and adds the hit.
		// The records are written as WHOLE lines, all 128 bytes of each, by consecutive threads, through LDS, kStage
		// records at a time: 96 of the 128 bytes -- by the lanes themselves or through LDS alike -- cost k_wave_shade_b
		// 13.6 -> 18.4 ms per step.  A line written in part is read, merged and written back.
		uint32_t tot = 0;
		for (int w = 0; w < kRBlock / 64; ++w) tot += s_wave[w];
		const uint32_t idx = cont ? off - s_base : 0xffffffffu;
		uint4 *dst = a.carry_out + (uint64_t)s_base * 8;
		for (uint32_t h = 0; h < tot; h += (uint32_t)kStage) { // (uniform over the workgroup)
			if (h) __syncthreads(); // (the round before has been copied out)
			if (idx >= h && idx < h + (uint32_t)kStage) {
				uint4 *rec = s_rec + (idx - h) * 8u;
				rec[0] = make_uint4(__float_as_uint(ray_o.x), __float_as_uint(ray_o.y), __float_as_uint(ray_o.z), (uint32_t)rng.state);
				rec[1] = make_uint4(__float_as_uint(ray_d.x), __float_as_uint(ray_d.y), __float_as_uint(ray_d.z), (uint32_t)(rng.state >> 32));
				rec[2] = make_uint4(__float_as_uint(thr.x), __float_as_uint(thr.y), __float_as_uint(thr.z), st_pack_ior(ior, delta));
				rec[3] = make_uint4(__float_as_uint(p_here.x), __float_as_uint(p_here.y), __float_as_uint(p_here.z), __float_as_uint(prev_pdf));
				rec[4] = make_uint4(__float_as_uint(L.x), __float_as_uint(L.y), __float_as_uint(L.z), (uint32_t)lane);
				rec[5] = make_uint4((uint32_t)rng.inc, (uint32_t)(rng.inc >> 32), 0u, 0u);
				rec[6] = make_uint4(0u, 0u, 0u, 0u); rec[7] = make_uint4(0u, 0u, 0u, 0u); // (the hit: k_wave_trace; spare)
			}
			__syncthreads();
			const uint32_t n = tot - h < (uint32_t)kStage ? tot - h : (uint32_t)kStage;
			for (uint32_t j = threadIdx.x; j < n * 8u; j += kRBlock) { // (streaming stores: see PG_ST)
				const uint4 q = s_rec[j];
				const u32x4_t v = {q.x, q.y, q.z, q.w};
				__builtin_nontemporal_store(v, reinterpret_cast<u32x4_t *>(dst + (uint64_t)h * 8 + j));
			}
		}
	} else if (cont) {
		st_store(a.st_out, a, 0, off, ray_o, (uint32_t)rng.state);
		st_store(a.st_out, a, 1, off, ray_d, (uint32_t)(rng.state >> 32));
		st_store(a.st_out, a, 2, off, thr, st_pack_ior(ior, delta));
		st_store(a.st_out, a, 3, off, p_here, __float_as_uint(prev_pdf));
		st_store(a.st_out, a, 4, off, L, (uint32_t)lane);
		a.inc_out[off] = rng.inc;
	}
}

// ---- :247-261, 302-381 ----
template <int kLevel, bool kFirst>
__global__ __launch_bounds__(kRBlock) void k_wave_shade_b(RenderArgs a)
{
	__shared__ uint32_t s_wave[kRBlock / 64];
	__shared__ uint32_t s_base;
	__shared__ uint32_t s_oct[kRBlock / 64][8];
	extern __shared__ uint4 s_rec[]; // kRBlock * 8 entries when the next bounce is sorted (a.carry_out), else none
	uint64_t tid;
	bool alive;
	if (!wave_entry<kFirst>(a, tid, alive)) return;
	// records of the earlier bounces: all paths for the first, the survivors of bounce j for bounce j+1
	uint64_t rec_base = 0;
	if (!kFirst) {
		rec_base = a.n_lanes;
		for (int j = 0; j + 1 < a.bounce; ++j) rec_base += live_final(a, j);
	}
	bool cont = false;
	// what a survivor takes along to its place in the next list
	v3 ray_o = V(0, 0, 0), ray_d = V(0, 0, 1), thr = V(1, 1, 1), L = V(0, 0, 0), p_here = V(0, 0, 0);
	float ior = 1.0f, prev_pdf = 1.0f;
	bool delta = false;
	uint64_t lane = tid; // (the first list is in lane order)
	Pcg32 rng;
	rng.state = 0; rng.inc = 1;
	if (alive) {
		StageA A;
		A.flags = wsu(a, WS_FLAGS, tid);
		A.p = ws3(a, WS_P, tid); A.ng = ws3(a, WS_NG, tid);
		A.n = V(0, 0, 1); A.wi = V(0, 0, 1); A.mat = 0; A.refl = V(0, 0, 0);
		if (A.flags & F_SMP_TREE) { // (only these lanes evaluate their BSDF again)
			A.n = ws3(a, WS_U, tid); A.wi = ws3(a, WS_U + 3, tid);
			A.mat = (int)wsu(a, WS_U + 6, tid); A.refl = ws3(a, WS_U + 7, tid);
		}
		A.Le = (A.flags & F_HAS_LE) ? ws3(a, WS_LE, tid) : V(0, 0, 0);
		A.ds_d = V(0, 0, 0); // (stage_b does not read it: the record takes the canonical form k_wave_guide made)
		A.ds_pdf = 0.0f; A.em_w = V(0, 0, 0); A.bv_em = V(0, 0, 0); A.bp_em = 0.0f;
		if (A.flags & F_NEE_LIVE) {
			A.ds_pdf = wsf(a, WS_DS_PDF, tid); A.em_w = ws3(a, WS_EM_W, tid);
			A.bv_em = ws3(a, WS_BV_EM, tid); A.bp_em = wsf(a, WS_BP_EM, tid);
		}
		// (a tree-sampling lane's own BSDF sample is not used again: its direction comes from k_wave_guide, its
		// value and pdf from the evaluation in stage_b -- only the sample's eta is, :357)
		A.wo = V(0, 0, 0); A.bsdf_pdf = 0.0f; A.bsdf_w = V(0, 0, 0);
		if (A.flags & F_SMP_TREE) A.wo = ws3(a, WS_WO_T, tid);
		else { A.wo = ws3(a, WS_U, tid); A.bsdf_w = ws3(a, WS_U + 3, tid); A.bsdf_pdf = wsf(a, WS_U + 6, tid); }
		A.eta = kLevel >= 3 ? wsf(a, WS_ETA, tid) : 1.0f;
		GuideOut g;
		g.nee_cx = 0.0f; g.nee_cy = 0.0f; g.wo_cx = 0.0f; g.wo_cy = 0.0f; g.pdf_nee = 1.0f; g.pdf_tree = 1.0f;
		g.slot_path = kSlotNone; g.slot_nee = kSlotNone; g.tree_flags = 0u; // (k_wave_guide has put them into the record list)
		g.wo = A.wo; // (k_wave_guide's direction for the lanes that sample the tree)
		if (guide_has_work(a, A.flags)) { g.pdf_nee = wsf(a, WS_PDF_NEE, tid); g.pdf_tree = wsf(a, WS_PDF_TREE, tid); }
		const bool occluded = (A.flags & F_NEED_SHADOW) && wsu(a, WS_OCC, tid) != 0u;
		rng.state = (uint64_t)wsu(a, WS_RNG_LO, tid) | ((uint64_t)wsu(a, WS_RNG_HI, tid) << 32);
		rng.inc = a.perm ? ((uint64_t)wsu(a, WS_FWD + 8, tid) | ((uint64_t)wsu(a, WS_FWD + 9, tid) << 32)) : a.inc_in[tid];
		if (!kFirst) {
			uint4 q2, q4;
			if (a.perm) { // (a sorted bounce: k_wave_shade_a has brought them along)
				q2 = make_uint4(wsu(a, WS_FWD + 0, tid), wsu(a, WS_FWD + 1, tid), wsu(a, WS_FWD + 2, tid), wsu(a, WS_FWD + 3, tid));
				q4 = make_uint4(wsu(a, WS_FWD + 4, tid), wsu(a, WS_FWD + 5, tid), wsu(a, WS_FWD + 6, tid), wsu(a, WS_FWD + 7, tid));
			} else {
				q2 = st_load(a.st_in, a, 2, tid);
				q4 = st_load(a.st_in, a, 4, tid);
			}
			thr = st_v3(q2);
			ior = __uint_as_float(q2.w & 0x7fffffffu);
			L = st_v3(q4);
			lane = q4.w;
		}
		cont = stage_b<kLevel>(a, rng, thr, L, ior, A, g, occluded, lane, rec_base + tid, (uint32_t)a.bounce, ray_o, ray_d,
		                       prev_pdf, delta);
		p_here = A.p;
		if (kFirst) a.hit0[lane] = (A.flags & F_VALID) ? 1 : 0;
		if (!cont) a.Lq[lane] = make_uint4(__float_as_uint(L.x), __float_as_uint(L.y), __float_as_uint(L.z), 0u); // the path ends here: its radiance (:431), written once -- one 16-byte store (k_finish lays the output column out)
	}
	append_survivors<kRBlock>(a, cont, ray_o, ray_d, thr, ior, delta, p_here, prev_pdf, L, lane, rng, s_wave, s_base, s_oct, s_rec);
}

// ---- :189-381 in one kernel: everything of a bounce but the closest hit ----
// stage_a, the SD-tree calls, the shadow ray (an any-hit walk inline) and stage_b on one lane's registers: none of the
// workspace planes the split kernels hand each other is written or read -- about 360 of the 900 bytes a lane and bounce move
// through HBM.  The price: the shadow ray is walked by the lane that made it (no handing of finished lanes' slots to other
// rays as in k_wave_cast), and the tree walks run at the occupancy the shading's registers leave.  LDS: the walk's stacks
// and the hottest BVH nodes, and -- over the same bytes, once every walk of the workgroup is done -- the survivors'
// records of a sorted next bounce.
constexpr int kShadeStage = 128; // survivors' records staged in LDS at a time (224 at a time: no gain, profiles/r05/ab_shade_stage224_rejected.txt)
constexpr int kShadeTopNodes = 40; // BVH nodes kept in LDS beside the stash (the ray-casting kernels keep kBvhTopNodes): what five workgroups per compute unit leave
// The walk's stack keeps kShadeStack entries per lane in LDS here (the ray-casting kernels keep kLdsStack = 8; deeper ones go
// to the lane's overflow strip either way), which leaves room for the stash: kShadeStash values per lane that only stage_b
// reads wait in LDS while the two walks run.
constexpr int kShadeStack = 6;
static_assert(kShadeStack >= kMinLdsStack && kShadeStack <= kLdsStack, "k_wave_shade: LDS stack depth");
// (the stash: the nine values of stage_a1 that only stage_b reads and the path's throughput; the radiance so far, the index of
// refraction and the lane stay in registers -- rounds 4-5 measured the other splits)
constexpr int kShadeStash = 12;
constexpr int kShadeWalkQuads = kShadeStack * kRBlock / 2 + kShadeTopNodes * 8 + kShadeStash * kRBlock / 4;
constexpr int kShadeLdsQuads = kShadeStage * 8 > kShadeWalkQuads ? kShadeStage * 8 : kShadeWalkQuads;
// (122 vector registers, four waves per SIMD, which is also what 33 KB of LDS per workgroup allow.  Measured: staging the
// records 128 at a time -- 23 KB -- changes nothing by itself, and compiled for five waves on top of that the kernel spills
// 31 registers: 35.2 -> 37.6 ms per step.)
// The paths' 128-byte records of a sorted bounce are read THROUGH LDS by the wave as a whole: a lane that
// gathers the seven 16-byte entries of its own record makes seven vector loads that each touch 64 different cache lines --
// the compute unit's L1 looks up about one line per clock, so such a load holds the vector-memory path for 64 clocks however
// few bytes it wants -- where eight lanes that read one record's eight entries side by side touch 8 lines per load: the same
// 64 records in eight loads of 8 lines instead of seven of 64.  The entries cross to the lanes that own them through LDS
// ([entry][thread] planes over the bytes the walks' stacks and the stash use later: one more workgroup barrier, ahead of the
// staging of the BVH's top).
static_assert(7 * kRBlock <= kShadeLdsQuads, "k_wave_shade: the records' seven entries per thread must fit the dynamic LDS");
template <int kLevel, bool kFirst>
__device__ __forceinline__ void shade_body(const RenderArgs &a)
{
	__shared__ float s_planes[3 * kKdGridPlanes];
	__shared__ uint32_t s_wave[kRBlock / 64];
	__shared__ uint32_t s_base;
	__shared__ uint32_t s_oct[kRBlock / 64][8];
	extern __shared__ uint4 s_dyn[]; // kShadeLdsQuads entries: [stacks kLdsStack * kRBlock * 8 B][BVH top][...], later the records
	uint64_t tid;
	bool alive;
	if (!wave_entry<kFirst>(a, tid, alive)) return;
	// Probe builds only (-DPG_SHADE_PHASES=1; the instrumented pass of such a build prints the shares under $PGSD_TRACE_SHADOW):
	// where a wave spends its life, DepthCounters::phase.  NOT in the product: the seven s_memtime stamps, behind a uniform
	// branch that is never taken in a timed pass, took k_wave_shade from 28.6 to 45-47 ms per step (profiles/r05/
	// ab_shade_early_staging.txt: every variant of that run carries them).
#ifndef PG_SHADE_PHASES
#define PG_SHADE_PHASES 0
#endif
#if PG_SHADE_PHASES
	// A wave keeps its stamps in its own row of LDS (one lane, plain stores: LDS operations of a wave complete in order) and
	// flushes the row once at the end, by one eight-lane atomic into its stripe.  (Round 5's probe added every stamp to ONE line
	// of device memory: its k_wave_shade ran eleven times slower than the product's.  And what made a probe build 1.7 times
	// slower even with its stamps idle was not the stamps at all: see live_final(), pg_render_dev.hpp.  With both repaired this
	// build's k_wave_shade takes 27.7 ms per step against the product's 27.5, stamping or not: profiles/r06/phase_probe.txt.)
	// Seven stamps behind the start: 0 the records + staging, 1 stage_a1, 2 the shadow walk, 3 stage_a2, 4 the SD-tree calls, 5 stage_b,
	// 6 the survivors' append (pg_kernels.hpp).
	__shared__ unsigned long long s_ph[kRBlock / 64][10]; // [0..6] the phases, [7] stamps taken, [8] the last stamp
	if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(__ballot(1))) {
		s_ph[threadIdx.x >> 6][8] = (unsigned long long)clock64();
		s_ph[threadIdx.x >> 6][7] = 0ull;
	}
#define PG_PHASE(i)                                                                                                          \
	if (a.ph && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(__ballot(1))) {                                            \
		const unsigned long long t_now = (unsigned long long)clock64();                                                      \
		unsigned long long *row = s_ph[threadIdx.x >> 6];                                                                    \
		row[i] = t_now - row[8];                                                                                             \
		row[8] = t_now;                                                                                                      \
		row[7] |= 1ull << (i);                                                                                               \
	}
#else
#define PG_PHASE(i)
#endif
	// What the workgroup stages in LDS -- the KD grid's planes and the BVH's top -- is ASKED FOR first and written last: the loads
	// are in flight while the records make their own two round trips (the permutation, then the entries), instead of being
	// two more round trips behind them, each with a barrier of its own (an instrumented pass showed a quarter of a wave's life
	// gone before stage_a1 began: profiles/r05/shade_phases.txt).
	constexpr int kPlaneLoads = 1, kTopLoads = (kShadeTopNodes * 8 + kRBlock - 1) / kRBlock;
	float pl_pre[kPlaneLoads];
	u32x4_t top_pre[kTopLoads];
#pragma unroll
	for (int k = 0; k < kPlaneLoads; ++k) pl_pre[k] = 0.0f;
#pragma unroll
	for (int k = 0; k < kTopLoads; ++k) top_pre[k] = u32x4_t{0u, 0u, 0u, 0u};
	const uint32_t n_top = a.shapes.n_bvh_nodes < kShadeTopNodes ? (uint32_t)a.shapes.n_bvh_nodes : (uint32_t)kShadeTopNodes;
#pragma unroll
	for (int k = 0; k < kPlaneLoads; ++k)
		if (a.tree.kd_grid != nullptr && threadIdx.x + k * kRBlock < 3u * kKdGridPlanes) pl_pre[k] = a.tree.kd_planes[threadIdx.x + k * kRBlock];
#pragma unroll
	for (int k = 0; k < kTopLoads; ++k)
		if (threadIdx.x + k * kRBlock < n_top * 8u) top_pre[k] = reinterpret_cast<const u32x4_t *>(a.shapes.bvh)[threadIdx.x + k * kRBlock];
	uint4 cq0 = make_uint4(0u, 0u, 0u, 0u), cq1 = cq0, cq2 = cq0, cq3 = cq0, cq4 = cq0, cq5 = cq0, cq6 = cq0;
	uint32_t coop_place = (uint32_t)tid;
	const bool coop = !kFirst && a.perm != nullptr; // (uniform)
	if (coop) {
		if (alive && tid < (uint64_t)a.n_sort) coop_place = a.perm[tid];
		if (!alive) coop_place = 0u; // (a lane past the list still serves its wave's loads: any record that exists)
		const uint32_t l = threadIdx.x & 63u, wbase = threadIdx.x & ~63u, q = l & 7u;
		// ALL EIGHT gathers are issued before the first of them is waited for (round 6).  Written as one loop -- gather, then store
		// under `q < 7` -- every round was a basic block of its own: permute, wait, gather, WAIT, store, eight round trips one
		// after the other at the head of every wave (the ISA showed it: eight `s_waitcnt vmcnt(0)` in a row).  The gather is
		// unconditional now (entry 7 of a record is its unused tail, in the same 128-byte line as the rest: loaded, not kept),
		// so nothing splits the loop, and the stores follow in a loop of their own.
		uint4 g[8];
#pragma unroll
		for (uint32_t k = 0; k < 8u; ++k) {
			const uint32_t j = k * 8u + (l >> 3);
			const uint32_t pj = (uint32_t)__shfl((int)coop_place, (int)j, 64);
			g[k] = gather16(a.carry_in + (uint64_t)pj * 8 + q);
		}
#pragma unroll
		for (uint32_t k = 0; k < 8u; ++k)
			if (q < 7u) s_dyn[q * kRBlock + wbase + k * 8u + (l >> 3)] = g[k];
		// (written and read by the same wave: its LDS operations complete in order, so no workgroup barrier -- but the lanes read
		// entries OTHER lanes wrote, which the memory model only orders through a release / acquire pair at wavefront scope; the
		// fences and the wave barrier emit no instruction, they pin the order for the compiler: ADVICE r5)
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		cq0 = s_dyn[0 * kRBlock + threadIdx.x]; cq1 = s_dyn[1 * kRBlock + threadIdx.x]; cq2 = s_dyn[2 * kRBlock + threadIdx.x];
		cq3 = s_dyn[3 * kRBlock + threadIdx.x]; cq4 = s_dyn[4 * kRBlock + threadIdx.x]; cq5 = s_dyn[5 * kRBlock + threadIdx.x];
		cq6 = s_dyn[6 * kRBlock + threadIdx.x];
		__syncthreads(); // every wave has its entries: the bytes become the BVH's top, the stacks and the stash
	}
	uint2 *s_stack = reinterpret_cast<uint2 *>(s_dyn);
	u32x4_t *s_top = reinterpret_cast<u32x4_t *>(s_dyn) + kShadeStack * kRBlock / 2;
	BvhStack stk = bvh_stack(s_stack + threadIdx.x, a.bvh_ovf, (uint32_t)tid * (uint32_t)kOvfStack, kShadeStack);
#pragma unroll
	for (int k = 0; k < kPlaneLoads; ++k)
		if (a.tree.kd_grid != nullptr && threadIdx.x + k * kRBlock < 3u * kKdGridPlanes) s_planes[threadIdx.x + k * kRBlock] = pl_pre[k];
	if (a.tree.kd_grid != nullptr) // (a workgroup smaller than the planes' count -- variant builds -- takes the rest here)
		for (uint32_t i = threadIdx.x + kPlaneLoads * kRBlock; i < 3u * kKdGridPlanes; i += kRBlock) s_planes[i] = a.tree.kd_planes[i];
#pragma unroll
	for (int k = 0; k < kTopLoads; ++k)
		if (threadIdx.x + k * kRBlock < n_top * 8u) s_top[threadIdx.x + k * kRBlock] = top_pre[k];
	__syncthreads();
	stk.top = (const LdsQuad *)s_top;
	stk.n_top = n_top;
	bool cont = false;
	v3 ray_o = V(0, 0, 0), ray_d = V(0, 0, 1), thr = V(1, 1, 1), L = V(0, 0, 0), p_here = V(0, 0, 0), prev_p = V(0, 0, 0);
	float ior = 1.0f, prev_pdf = 1.0f;
	bool delta = false, prev_delta = true;
	uint64_t lane = tid; // (the first list is in lane order)
	Pcg32 rng;
	rng.state = 0; rng.inc = 1;
	uint64_t rec_base = 0;
	if (!kFirst) {
		rec_base = a.n_lanes;
		for (int j = 0; j + 1 < a.bounce; ++j) rec_base += live_final(a, j);
		// (the same number in every lane, but read by vector loads -- live_count is written by this very grid -- so the
		// compiler keeps it in two vector registers through the whole kernel unless it is told that it is uniform)
		rec_base = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)rec_base) |
		           ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(rec_base >> 32)) << 32);
	}
	if (alive) {
		// What a path carries from bounce to bounce is read ONCE, here: the ray, the sampler, the hit, the throughput, the index
		// of refraction, the radiance so far and the path's lane.  (Reading the last four a second time behind the walks, to keep
		// them out of the walks' registers, cost 1.25 ms per step -- two more gathers per lane; they wait in registers and, the
		// throughput, in the LDS stash: DESIGN 5.7.)
		HitRec h;
		const bool from_rec = !kFirst && a.perm != nullptr; // a sorted bounce: the path's 128-byte record, through the permutation
		uint32_t place = (uint32_t)tid; // (32-bit: a place in the live list; one register through the walks, not two)
		if (!coop && from_rec && tid < (uint64_t)a.n_sort) place = a.perm[tid];
		if (from_rec) {
			const uint4 *rec = a.carry_in + (uint64_t)place * 8;
			uint4 q0, q1, q2, q3, q5, q6;
			if (coop) { q0 = cq0; q1 = cq1; q2 = cq2; q3 = cq3; q5 = cq5; q6 = cq6; }
			else { q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; q3 = rec[3]; q5 = rec[5]; q6 = rec[6]; }
			rng.state = (uint64_t)q0.w | ((uint64_t)q1.w << 32);
			rng.inc = (uint64_t)q5.x | ((uint64_t)q5.y << 32);
			ray_o = st_v3(q0); ray_d = st_v3(q1);
			thr = st_v3(q2);
			{ const uint4 q4 = coop ? cq4 : rec[4]; ior = __uint_as_float(q2.w & 0x7fffffffu); L = st_v3(q4); lane = q4.w; }
			prev_delta = (q2.w >> 31) != 0u;
			prev_p = st_v3(q3);
			prev_pdf = __uint_as_float(q3.w);
			h.prim = (int)q6.x; h.t = __uint_as_float(q6.y); h.u = __uint_as_float(q6.z); h.v = __uint_as_float(q6.w);
		} else {
			const uint4 q0 = st_load(a.st_in, a, 0, tid), q1 = st_load(a.st_in, a, 1, tid);
			rng.state = (uint64_t)q0.w | ((uint64_t)q1.w << 32);
			rng.inc = a.inc_in[tid];
			ray_o = st_v3(q0); ray_d = st_v3(q1);
			if (!kFirst) {
				const uint4 q2 = st_load(a.st_in, a, 2, tid), q3 = st_load(a.st_in, a, 3, tid);
				thr = st_v3(q2);
				{ const uint4 q4 = st_load(a.st_in, a, 4, tid); ior = __uint_as_float(q2.w & 0x7fffffffu); L = st_v3(q4); lane = q4.w; }
				prev_delta = (q2.w >> 31) != 0u;
				prev_p = st_v3(q3);
				prev_pdf = __uint_as_float(q3.w);
			}
			h.prim = (int)wsu(a, WS_HIT_PRIM, tid);
			h.t = wsf(a, WS_HIT_T, tid); h.u = wsf(a, WS_HIT_U, tid); h.v = wsf(a, WS_HIT_V, tid);
		}
		StageA A;
		PG_PHASE(0)
		stage_a1<kLevel>(a, rng, ray_o, ray_d, thr, prev_p, prev_pdf, prev_delta, h, (uint32_t)a.bounce, A);
		// What stage_a1 leaves is PINNED here: every output is made now, so that its inputs die.  (Left alone the compiler
		// sinks the last operations of a value that is only read behind the walk -- the emitted radiance's three products, the
		// flag word's bits -- below the walk, and carries their more numerous inputs through it instead.)
		asm volatile("" : "+v"(A.flags), "+v"(A.wi.x), "+v"(A.wi.y), "+v"(A.wi.z), "+v"(A.p.x), "+v"(A.p.y), "+v"(A.p.z),
		                  "+v"(A.ng.x), "+v"(A.ng.y), "+v"(A.ng.z), "+v"(A.refl.x), "+v"(A.refl.y), "+v"(A.refl.z));
		asm volatile("" : "+v"(A.ds_d.x), "+v"(A.ds_d.y), "+v"(A.ds_d.z), "+v"(A.ds_pdf), "+v"(A.bp_em), "+v"(A.n.x), "+v"(A.n.y), "+v"(A.n.z));
		// nine values only stage_b reads wait in LDS while the two walks run (column threadIdx.x of a [9][kRBlock] array)
		float *stash = reinterpret_cast<float *>(s_dyn) + (kShadeStack * kRBlock * 2 + kShadeTopNodes * 32) + threadIdx.x;
		stash[0 * kRBlock] = A.Le.x; stash[1 * kRBlock] = A.Le.y; stash[2 * kRBlock] = A.Le.z;
		stash[3 * kRBlock] = A.bv_em.x; stash[4 * kRBlock] = A.bv_em.y; stash[5 * kRBlock] = A.bv_em.z;
		stash[6 * kRBlock] = A.em_w.x; stash[7 * kRBlock] = A.em_w.y; stash[8 * kRBlock] = A.em_w.z;
		// ... and with them what the path carries that only stage_b needs: throughput, radiance so far, index of refraction, lane
		// (round-4 measurement: reading these a second time from the path's record cost 1.25 ms per step -- two more 16-byte
		// gathers per lane in a kernel whose time follows its gathers -- where eight LDS words cost nothing)
		stash[9 * kRBlock] = thr.x; stash[10 * kRBlock] = thr.y; stash[11 * kRBlock] = thr.z;
		// The shadow ray is walked HERE, between the two halves of stage_a: the BSDF sample is not made yet and the SD-tree
		// calls have no results yet, so neither is alive across the walk -- the walk's 50 registers on top of everything a
		// bounce keeps were this kernel's register peak (123 with the walk at the end).  The walk draws no
		// sample, so the sampler's order is untouched.
		bool occluded = false;
		PG_PHASE(1)
		if (a.dc) { // (instrumented passes only: how full are the waves that walk?)
			const unsigned long long m = __ballot((A.flags & F_NEED_SHADOW) != 0u);
			if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(__ballot(1))) {
				atomicAdd(&a.dc->body_waves, 1ull);
				if (m) { atomicAdd(&a.dc->shadow_waves, 1ull); atomicAdd(&a.dc->shadow_lanes, (unsigned long long)__popcll(m)); }
			}
		}
		if (A.flags & F_NEED_SHADOW) { // :213 test_visibility
			float th, bu, bv;
			occluded = intersect<kLevel, true, true>(a.shapes, A.sh_o, A.sh_d, A.sh_tmax, th, stk, bu, bv) >= 0;
		}
		PG_PHASE(2)
		// (the shading frame and the material row are functions of the normal and the material's number: made again from
		// them behind the walk -- a dozen operations and two loads -- instead of being carried through it, which is what
		// the compiler does unless it is told that these ARE new values)
		asm volatile("" : "+v"(A.n.x), "+v"(A.n.y), "+v"(A.n.z), "+v"(A.mat));
		stage_a2<kLevel>(a, rng, A);
		// (stage_a2's outputs pinned like stage_a1's: the nine products of to_world() were carried through the SD-tree walks
		// instead of the three sums)
		asm volatile("" : "+v"(A.wo.x), "+v"(A.wo.y), "+v"(A.wo.z), "+v"(A.bsdf_w.x), "+v"(A.bsdf_w.y), "+v"(A.bsdf_w.z),
		                  "+v"(A.bsdf_pdf), "+v"(A.flags));
		if (kLevel >= 3) asm volatile("" : "+v"(A.eta));
		GuideOut g;
		g.nee_cx = 0.0f; g.nee_cy = 0.0f; g.wo_cx = 0.0f; g.wo_cy = 0.0f; g.pdf_nee = 1.0f; g.pdf_tree = 1.0f;
		g.slot_path = kSlotNone; g.slot_nee = kSlotNone; g.tree_flags = 0u;
		g.wo = A.wo;
		PG_PHASE(3)
		if (guide_has_work(a, A.flags)) stage_guide(a, s_planes, rng, A.p, A.ds_d, (A.flags & F_SMP_TREE) ? V(0, 0, 0) : A.wo, A.flags, g);
		if (a.record && (A.flags & F_VALID)) store_slots(a, rec_base + tid, g);
		PG_PHASE(4)
		asm volatile("" : "+v"(A.n.x), "+v"(A.n.y), "+v"(A.n.z), "+v"(A.mat)); // (the same behind the SD-tree walks, for stage_b's second BSDF evaluation)
		A.Le = V(stash[0 * kRBlock], stash[1 * kRBlock], stash[2 * kRBlock]);
		A.bv_em = V(stash[3 * kRBlock], stash[4 * kRBlock], stash[5 * kRBlock]);
		A.em_w = V(stash[6 * kRBlock], stash[7 * kRBlock], stash[8 * kRBlock]);
		thr = V(stash[9 * kRBlock], stash[10 * kRBlock], stash[11 * kRBlock]);
		cont = stage_b<kLevel>(a, rng, thr, L, ior, A, g, occluded, lane, rec_base + tid, (uint32_t)a.bounce, ray_o, ray_d, prev_pdf, delta);
		p_here = A.p;
		if (kFirst) a.hit0[lane] = (A.flags & F_VALID) ? 1 : 0;
		if (!cont) a.Lq[lane] = make_uint4(__float_as_uint(L.x), __float_as_uint(L.y), __float_as_uint(L.z), 0u); // the path ends here
		PG_PHASE(5)
	}
	// (append_survivors' first barrier comes after every walk of the workgroup: from there on the stacks' bytes hold records)
	append_survivors<kShadeStage>(a, cont, ray_o, ray_d, thr, ior, delta, p_here, prev_pdf, L, lane, rng, s_wave, s_base, s_oct, s_dyn);
	PG_PHASE(6)
#undef PG_PHASE
#if PG_SHADE_PHASES
	if (a.ph && (threadIdx.x & 63u) < 8u) { // the wave's row, once: eight lanes, one line of the wave's stripe
		const unsigned long long *row = s_ph[threadIdx.x >> 6];
		const unsigned l = threadIdx.x & 63u;
		// (a slot whose stamp this wave never passed -- a wave without a live lane skips the body -- holds nothing of this launch)
		const unsigned long long v = l == 7u ? 1ull : ((row[7] >> l) & 1ull) ? row[l] : 0ull;
		atomicAdd(&a.ph[(size_t)((blockIdx.x * (kRBlock / 64) + (threadIdx.x >> 6)) & (unsigned)(kPhaseStripes - 1)) * kPhaseWords + l], v);
	}
#endif
}

// (The probe build's stamps cost the level-2 kernel one register more than five waves per SIMD leave -- 97: four waves, and the
// kernel 15 % -- so it is told the occupancy, as the level-3 kernel below is: 96 registers and a few spilled bytes.  A probe
// must run at the product's occupancy to say anything about the product.)
#if PG_SHADE_PHASES
#define PG_PROBE_WAVES __attribute__((amdgpu_waves_per_eu(5)))
#else
#define PG_PROBE_WAVES
#endif
template <int kLevel, bool kFirst>
__global__ __launch_bounds__(kRBlock) PG_PROBE_WAVES void k_wave_shade(RenderArgs a)
{
	shade_body<kLevel, kFirst>(a);
}
// Feature level 3 (torus: transmission, delta lobes, one-sided BSDFs) needs three registers more than five waves per SIMD leave
// (99 of 96); told the occupancy it is wanted at, the compiler finds an allocation of 96 without a byte of scratch.  (The
// same hint makes the level-2 kernel, which fits by itself at 96, spill 12 bytes per lane: it is left alone.)
template <bool kFirst>
__global__ __launch_bounds__(kRBlock) __attribute__((amdgpu_waves_per_eu(5))) void k_wave_shade_l3(RenderArgs a)
{
	shade_body<3, kFirst>(a);
}

// See tail_checkpoint (pg_render_dev.hpp): launched before the launches of bounce a.bounce with a grid
// for kTailPaths lanes.  When no more paths than that are alive every lane follows its own path to
// its end here, all stages in sequence with their intermediate results in registers; the state still
// goes through the per-lane arrays only at the start.  Record entries of the later bounces are handed
// out wave by wave behind the entries of bounce a.bounce (a.live_count[max_depth] counts them), and
// the survivors of every bounce are added to live_count[] as the per-bounce launches would have.
template <int kLevel>
__global__ __launch_bounds__(kRBlock) void k_wave_tail(RenderArgs a)
{
	__shared__ uint2 s_stack[kLdsStack][kRBlock];
	__shared__ float s_planes[3 * kKdGridPlanes];
	const uint64_t tid = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	const uint64_t live = (uint64_t)live_final(a, a.bounce - 1);
	if (live > kTailPaths || (uint64_t)blockIdx.x * kRBlock >= live) return;
	if (tail_took_over(a, a.bounce - 1)) return; // an earlier checkpoint already did
	stage_kd_planes(s_planes, a.tree);
	bool alive = tid < live;
	uint64_t lane = 0;
	uint64_t rec_base = a.n_lanes; // entries of the bounces before a.bounce
	for (int j = 0; j + 1 < a.bounce; ++j) rec_base += live_final(a, j);
	const uint64_t tail_base = rec_base + live; // behind the entries of bounce a.bounce
	uint64_t slot = rec_base + tid;
	const unsigned wl = threadIdx.x & 63u;
	const BvhStack stk = bvh_stack(&s_stack[0][threadIdx.x], a.bvh_ovf, (uint32_t)tid * (uint32_t)kOvfStack);
	Pcg32 rng;
	v3 ray_o = V(0, 0, 0), ray_d = V(0, 0, 1), thr = V(0, 0, 0), L = V(0, 0, 0), prev_p = V(0, 0, 0);
	float prev_pdf = 1.0f, ior = 1.0f;
	bool prev_delta = false;
	rng.state = 0; rng.inc = 1;
	if (alive) {
		uint4 q0, q1, q2, q3, q4;
		if (a.carry_in) { // the bounce taken over would have been a sorted one: its state is in the paths' records
			const uint4 *rec = a.carry_in + tid * 8;
			q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; q3 = rec[3]; q4 = rec[4];
			const uint4 q5 = rec[5];
			rng.inc = (uint64_t)q5.x | ((uint64_t)q5.y << 32);
		} else {
			q0 = st_load(a.st_in, a, 0, tid); q1 = st_load(a.st_in, a, 1, tid); q2 = st_load(a.st_in, a, 2, tid);
			q3 = st_load(a.st_in, a, 3, tid); q4 = st_load(a.st_in, a, 4, tid);
			rng.inc = a.inc_in[tid];
		}
		rng.state = (uint64_t)q0.w | ((uint64_t)q1.w << 32);
		ray_o = st_v3(q0); ray_d = st_v3(q1); thr = st_v3(q2); prev_p = st_v3(q3);
		ior = __uint_as_float(q2.w & 0x7fffffffu);
		prev_delta = (q2.w >> 31) != 0u;
		prev_pdf = __uint_as_float(q3.w);
		L = st_v3(q4);
		lane = q4.w;
	}
	for (int depth = a.bounce; depth < a.max_depth; ++depth) {
		if (alive) {
			HitRec h;
			h.u = 0.0f; h.v = 0.0f;
			h.prim = intersect<kLevel, false>(a.shapes, ray_o, ray_d, __builtin_huge_valf(), h.t, stk, h.u, h.v);
			StageA A;
			stage_a<kLevel>(a, rng, ray_o, ray_d, thr, prev_p, prev_pdf, prev_delta, h, (uint32_t)depth, A);
			bool occluded = false;
			if (A.flags & F_NEED_SHADOW) {
				float th, bu, bv;
				occluded = intersect<kLevel, true>(a.shapes, A.sh_o, A.sh_d, A.sh_tmax, th, stk, bu, bv) >= 0;
			}
			GuideOut g;
			g.nee_cx = 0.0f; g.nee_cy = 0.0f; g.wo_cx = 0.0f; g.wo_cy = 0.0f; g.pdf_nee = 1.0f; g.pdf_tree = 1.0f;
			g.slot_path = kSlotNone; g.slot_nee = kSlotNone; g.tree_flags = 0u;
			g.wo = A.wo;
			if (guide_has_work(a, A.flags)) stage_guide(a, s_planes, rng, A.p, A.ds_d, A.wo, A.flags, g);
			if (a.record && (A.flags & F_VALID)) store_slots(a, slot, g);
			bool delta;
			alive = stage_b<kLevel>(a, rng, thr, L, ior, A, g, occluded, lane, slot, (uint32_t)depth, ray_o, ray_d, prev_pdf, delta);
			prev_p = A.p;
			prev_delta = delta;
			if (!alive) a.Lq[lane] = make_uint4(__float_as_uint(L.x), __float_as_uint(L.y), __float_as_uint(L.z), 0u); // the path ends here
		}
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

// The grid of a persistent kernel: as many workgroups as the device holds at once (the occupancy the
// register count of this instantiation allows, asked of the runtime once), never more than the rays need.
template <class K>
static dim3 persistent_grid(K kernel, unsigned &cached_per_cu, unsigned n_cus, uint64_t n_rays)
{
	if (cached_per_cu == 0) {
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kRBlock, 0) != hipSuccess || nb < 1) nb = 4;
		cached_per_cu = (unsigned)nb;
	}
	uint64_t blocks = (uint64_t)cached_per_cu * n_cus;
	const uint64_t need = (n_rays + kRBlock - 1) / kRBlock;
	if (blocks > need) blocks = need;
	return dim3((unsigned)(blocks ? blocks : 1));
}

// ---- launcher: one stage of one bounce (pg_render_pass wraps each in its timing events) ----
template <int kLevel>
static void launch_stage_level(int stage, bool first, const RenderArgs &a, dim3 grid, unsigned n_cus, hipStream_t s)
{
	const dim3 block(kRBlock);
	static unsigned occ[2] = {0, 0}; // resident workgroups per CU of the two shadow-ray instantiations of this level
	switch (stage) {
	case 0:
		if (first) hipLaunchKernelGGL((k_wave_trace<kLevel, true>), grid, block, 0, s, a);
		else hipLaunchKernelGGL((k_wave_trace<kLevel, false>), grid, block, 0, s, a);
		break;
	case 1:
		if (a.fuse_guide) {
			if (first) hipLaunchKernelGGL((k_wave_shade_a<kLevel, true, true>), grid, block, 0, s, a);
			else hipLaunchKernelGGL((k_wave_shade_a<kLevel, false, true>), grid, block, 0, s, a);
		} else if (first) hipLaunchKernelGGL((k_wave_shade_a<kLevel, true, false>), grid, block, 0, s, a);
		else hipLaunchKernelGGL((k_wave_shade_a<kLevel, false, false>), grid, block, 0, s, a);
		break;
	case 2:
		if (first) hipLaunchKernelGGL((k_wave_cast<kLevel, true>), persistent_grid(k_wave_cast<kLevel, true>, occ[0], n_cus, a.n_lanes), block, 0, s, a);
		else hipLaunchKernelGGL((k_wave_cast<kLevel, false>), persistent_grid(k_wave_cast<kLevel, false>, occ[1], n_cus, a.n_lanes), block, 0, s, a);
		break;
	case 3: hipLaunchKernelGGL(k_wave_guide, grid, block, 0, s, a); break;
	case 4: {
		const size_t lds = a.carry_out ? (size_t)kRBlock * 8 * sizeof(uint4) : 0; // (the survivors' records of a sorted next bounce)
		if (first) hipLaunchKernelGGL((k_wave_shade_b<kLevel, true>), grid, block, lds, s, a);
		else hipLaunchKernelGGL((k_wave_shade_b<kLevel, false>), grid, block, lds, s, a);
		break;
	}
	case 6: {
		// (dev switch: $PGSD_SHADE_LDS_PAD bytes of extra dynamic LDS per workgroup, to see what one resident wave fewer costs)
		static const size_t pad = getenv("PGSD_SHADE_LDS_PAD") ? (size_t)atol(getenv("PGSD_SHADE_LDS_PAD")) : 0;
		const size_t lds = (size_t)kShadeLdsQuads * sizeof(uint4) + pad;
		static bool told = false;
		if (!told && getenv("PGSD_TRACE_OCC")) { // (dev switch: how many workgroups of this kernel a compute unit holds)
			told = true;
			int nb = 0;
			if constexpr (kLevel == 3) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_wave_shade_l3<false>, kRBlock, lds);
			else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_wave_shade<kLevel, false>, kRBlock, lds);
			fprintf(stderr, "[pgsd] k_wave_shade<%d>: %d workgroups of %d threads per compute unit with %zu bytes of dynamic LDS\n", kLevel, nb, kRBlock, lds);
		}
		if constexpr (kLevel == 3) {
			if (first) hipLaunchKernelGGL((k_wave_shade_l3<true>), grid, block, lds, s, a);
			else hipLaunchKernelGGL((k_wave_shade_l3<false>), grid, block, lds, s, a);
		} else {
			if (first) hipLaunchKernelGGL((k_wave_shade<kLevel, true>), grid, block, lds, s, a);
			else hipLaunchKernelGGL((k_wave_shade<kLevel, false>), grid, block, lds, s, a);
		}
		break;
	}
	default: hipLaunchKernelGGL((k_wave_tail<kLevel>), grid, block, 0, s, a); break;
	}
}

void launch_wave_stage(int stage, int level, bool first, const RenderArgs &a, unsigned grid_blocks, unsigned n_cus, hipStream_t s)
{
	if (level >= 3) launch_stage_level<3>(stage, first, a, dim3(grid_blocks), n_cus, s);
	else launch_stage_level<2>(stage, first, a, dim3(grid_blocks), n_cus, s);
}

int wave_workspace_planes() { return WS_COUNT; }

bool shade_phases_compiled_in() { return PG_SHADE_PHASES != 0; }

} // namespace pg
