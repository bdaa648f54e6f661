// pg_kernels_splat.hip -- recording into sdTree_current.
//
// The reference adds fp32 values with one float atomic per node on the root->leaf path
// (kdtree.py:199, quadtree.py:93): D_kd + 2*D_quad atomics per record, all records hitting the
// same root words.  Here a record touches only the accumulators of the two quadtree leaves its
// directions fall into: integer sums are order independent, so inner-node totals (and the KD
// counts, carried in word 3 of the path direction's accumulator) are formed once at refine time
// by a bottom-up pass, and the multi-GPU exchange is an exact int64 all-reduce.
//
// Scattered global atomics are the bound of this kernel (about 24 G sector-updates/s on MI355X,
// tools/atomic_probe.hip), and the hardware charges one update per 32-byte sector touched by a
// wave-instruction, not per lane.  A record's four words therefore go out from four ADJACENT lanes
// of one instruction (transposed through LDS): one update per direction instead of up to four.
#include "pg_descent.hpp"
#include "pg_kernels.hpp"

namespace pg {

constexpr int kBlock = 256;
static_assert(kBlock == kStageThreads, "stage_kd_planes copies one plane per thread");

__device__ __forceinline__ TreeHead load_head_s(const TreeHead *h, uint32_t t)
{
	const uint2 v = gather8(h + t);
	TreeHead r;
	r.root_rec = v.x;
	r.root_irr = __uint_as_float(v.y);
	return r;
}

struct SlotAdd { // what one (direction, weight) pair adds, and where
	long long *ptr; // accumulator base (kAccWords words), nullptr = nothing to add
	long long w0, w1, w2, w3;
};

// quadtree.py:398-441 for one (direction, weight) pair whose leaf has been found; `count` goes to word 3
__device__ __forceinline__ SlotAdd plan_dir(const AccumView &a, uint32_t tree, const LeafCursor &c, float w,
                                            long long count)
{
	SlotAdd s = {nullptr, 0, 0, 0, 0};
	if (!c.found) return s;
	const Limbs q = quantize_weight(w);
	s.ptr = c.is_root ? a.root_acc + (size_t)kAccWords * tree : a.rec_acc + (size_t)kAccWords * c.slot;
	s.w0 = q.l0; s.w1 = q.l1; s.w2 = q.l2; s.w3 = count;
	return s;
}

// KDTree.addDataPropagate (kdtree.py:180-225) + QuadTree.addDataPropagate (quadtree.py:389-464):
// finds where the record's two contributions go; the adds themselves are issued by coop_add
__device__ __forceinline__ void plan_record(const TreeView &t, const AccumView &a, const uint4 *s_kd, int store_nee,
                                            float x, float y, float z, float dx, float dy, float radiance,
                                            float wo_pdf, float nx, float ny, float nee_lum, SlotAdd &path,
                                            SlotAdd &nee, unsigned &kd_lv, unsigned &q_lv, unsigned &q_q, unsigned &bytes)
{
	const bool inside = inside_root(t, x, y, z);
	KdNode leaf;
	uint32_t lv;
	kd_descend_grid(t, reinterpret_cast<const float *>(s_kd), x, y, z, inside, leaf, lv);
	kd_lv += stat_levels(lv); // (statistics words, pg_descent.hpp: a thread may plan many records, so they are unpacked here)
	bytes += stat_bytes(lv);
	const uint32_t tree = leaf.tree; // outside the bbox: node 0's (stale) tree (kdtree.py:224)
	// (the tree's head and the jump-table entries of the record's two directions need the tree's number only: three
	// gathers in flight at once instead of the entries behind the head)
	const TreeHead head = load_head_s(t.head, tree);
	const JumpPre pre_p = jump_prefetch(t.jump, tree, dx, dy, in_unit_square(dx, dy));
	const JumpPre pre_n = jump_prefetch(t.jump, tree, nx, ny, store_nee != 0 && in_unit_square(nx, ny));
	const float w = wo_pdf > 0.0f ? radiance / wo_pdf : 0.0f;   // quadtree.py:451
	const float wn = wo_pdf > 0.0f ? nee_lum / wo_pdf : 0.0f;   // quadtree.py:462
	LeafCursor cp = leaf_cursor_pre(head, dx, dy, true, pre_p), cn = leaf_cursor_pre(head, nx, ny, store_nee != 0, pre_n);
	quad_find_leaf_slots2(t.rec, cp, cn);
	path = plan_dir(a, tree, cp, w, inside ? 1 : 0);
	q_lv += stat_levels(cp.levels);
	bytes += stat_bytes(cp.levels);
	++q_q;
	// a counted record whose direction reaches no leaf (outside the unit square): fallback counter
	if (inside && path.ptr == nullptr) atomicAdd(a.leaf_count + tree, 1ull);
	nee = plan_dir(a, tree, cn, wn, 0);
	if (store_nee) {
		q_lv += stat_levels(cn.levels);
		bytes += stat_bytes(cn.levels);
		++q_q;
	}
}

// The exchange below stays inside one wave (each wave owns its 64 entries of s_val / s_ptr), and a
// wave's LDS operations execute in order: ordering the compiler's view is all that is needed, no
// workgroup barrier (which would make every wave wait for the slowest descent of the workgroup).
__device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// Every thread of the wave calls this (convergent).  Lane L of a wave issues word (L & 3) of
// the record held by lane r*16 + (L >> 2) in round r: the four words of one accumulator leave in
// one wave-instruction from four adjacent lanes.
__device__ __forceinline__ void coop_add(const SlotAdd &s, long long *s_val, unsigned long long *s_ptr)
{
	const unsigned t = threadIdx.x;
	wave_lds_sync(); // the previous call's reads are done
	s_val[4 * t + 0] = s.w0;
	s_val[4 * t + 1] = s.w1;
	s_val[4 * t + 2] = s.w2;
	s_val[4 * t + 3] = s.w3;
	const unsigned long long mine = reinterpret_cast<unsigned long long>(s.ptr);
	s_ptr[t] = mine;
	wave_lds_sync();
	const unsigned lane = t & 63u, wbase = t & ~63u, word = lane & 3u;
	// Lanes of a wave often add to the SAME accumulator (the NEE direction of a small light is the same for a whole surface;
	// coarse quadtree cells) -- and not only neighbouring lanes: a sorted bounce puts the vertices of one spatial cell side by
	// side, in no order inside the cell, so the same target comes back every few lanes.  Every lane finds the LOWEST lane of
	// its wave with its target (eight ballots on a hash of the address pick the candidates, the candidate's address is
	// compared: a collision of the hash only loses a merge), adds its four words to that lane's in LDS (64-bit integer adds:
	// exact, any order) and drops out -- one update per distinct target of the wave instead of one per run of neighbours
	// (rounds 1-5), which the memory side serialises.
	const unsigned hkey = (unsigned)((mine >> 5) ^ (mine >> 13) ^ (mine >> 21)) & 255u;
	unsigned long long peers = __ballot(mine != 0);
#pragma unroll
	for (int b = 0; b < 8; ++b) {
		const unsigned long long m = __ballot(mine != 0 && ((hkey >> b) & 1u));
		peers &= ((hkey >> b) & 1u) ? m : ~m;
	}
	const unsigned leader = mine != 0 ? (unsigned)__builtin_ctzll(peers) : lane; // (a lane with a target is its own peer)
	const bool follower = mine != 0 && leader != lane && s_ptr[wbase + leader] == mine;
	if (follower) {
		unsigned long long *dst = reinterpret_cast<unsigned long long *>(s_val + 4 * (wbase + leader));
		if (s.w0) atomicAdd(dst + 0, (unsigned long long)s.w0);
		if (s.w1) atomicAdd(dst + 1, (unsigned long long)s.w1);
		if (s.w2) atomicAdd(dst + 2, (unsigned long long)s.w2);
		if (s.w3) atomicAdd(dst + 3, (unsigned long long)s.w3);
	}
	wave_lds_sync(); // every follower has added
	if (follower) s_ptr[t] = 0;
	wave_lds_sync();
#pragma unroll
	for (unsigned r = 0; r < 4; ++r) {
		const unsigned src = wbase + r * 16u + (lane >> 2);
		const long long v = s_val[4 * src + word];
		long long *p = reinterpret_cast<long long *>(s_ptr[src]);
		if (p != nullptr && v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(p + word), (unsigned long long)v);
	}
}

__device__ __forceinline__ unsigned long long wave_sum_s(unsigned long long v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}

__device__ __forceinline__ void count_depths_s(DepthCounters *dc, unsigned kd_lv, unsigned kd_q,
                                               unsigned q_lv, unsigned q_q, unsigned bytes)
{
	if (dc == nullptr) return;
	const unsigned long long a = wave_sum_s(kd_lv), b = wave_sum_s(kd_q), c = wave_sum_s(q_lv), d = wave_sum_s(q_q);
	const unsigned long long e = wave_sum_s(bytes);
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&dc->kd_levels, a);
		atomicAdd(&dc->kd_queries, b);
		atomicAdd(&dc->quad_levels, c);
		atomicAdd(&dc->quad_queries, d);
		atomicAdd(&dc->layout_bytes, e);
	}
}

__global__ __launch_bounds__(kBlock) void k_splat(TreeView t, AccumView a, int store_nee, uint64_t m,
                                                  const float *__restrict__ pos, const float *__restrict__ dir,
                                                  const float *__restrict__ radiance,
                                                  const float *__restrict__ wo_pdf,
                                                  const float *__restrict__ dir_nee,
                                                  const float *__restrict__ nee_lum,
                                                  const uint32_t *__restrict__ d_count, DepthCounters *dc)
{
	__shared__ float s_planes[3 * kKdGridPlanes];
	const uint4 *s_kd = reinterpret_cast<const uint4 *>(s_planes); // (plan_record's parameter: the staged table of whichever kind)
	__shared__ long long s_val[kBlock * 4];
	__shared__ unsigned long long s_ptr[kBlock];
	stage_kd_planes(s_planes, t);
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	const uint64_t valid = d_count ? (uint64_t)*d_count : m; // plane stride stays m
	unsigned kd_lv = 0, q_lv = 0, q_q = 0, did = 0, st_bytes = 0;
	SlotAdd path = {nullptr, 0, 0, 0, 0}, nee = {nullptr, 0, 0, 0, 0};
	if (i < valid && i < m) {
		const float nx = store_nee ? dir_nee[i] : 0.0f, ny = store_nee ? dir_nee[m + i] : 0.0f;
		const float nl = store_nee ? nee_lum[i] : 0.0f;
		plan_record(t, a, s_kd, store_nee, pos[i], pos[m + i], pos[2 * m + i], dir[i], dir[m + i], radiance[i],
		            wo_pdf[i], nx, ny, nl, path, nee, kd_lv, q_lv, q_q, st_bytes);
		did = 1;
	}
	coop_add(path, s_val, s_ptr);
	if (store_nee) coop_add(nee, s_val, s_ptr);
	count_depths_s(dc, kd_lv, did, q_lv, q_q, st_bytes);
}

// processPathData + scatterDataIntoSDTree's filter for dense slot g
// (path_guiding_integrator.py:434-478).  Returns keep; outputs the tree's inputs.
__device__ __forceinline__ bool process_slot(uint64_t g, uint64_t S, uint64_t num_rays, uint64_t ray, bool active,
                                             const float *__restrict__ l_final, const pg_dense_records &r,
                                             float &radiance, float &nee_lum, float &wp)
{
	radiance = 0.0f; nee_lum = 0.0f; wp = 0.0f;
	if (!active) return false; // (an unused slot, or a path that left the scene: nothing to read)
	float in[3], nee[3];
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		float out = (l_final[ch * num_rays + ray] - r.throughput_radiance[ch * S + g]) / r.throughput_bsdf[ch * S + g];
		if (out != out) out = 0.0f;                         // :444
		float v = out / r.bsdf[ch * S + g];
		if (v != v) v = 0.0f;                               // :449
		in[ch] = v;
		float e = r.radiance_nee[ch * S + g];
		if (e != e) e = 0.0f;                               // :467
		nee[ch] = e;
	}
	radiance = luminance(in[0], in[1], in[2]);            // :452
	if (radiance != radiance) radiance = 0.0f;            // :466
	nee_lum = luminance(nee[0], nee[1], nee[2]);
	wp = r.wo_pdf[g];
	const bool both_zero = (radiance == 0.0f) && (nee_lum == 0.0f); // :470-472
	return active && !both_zero && !(wp == 0.0f) && !(wp != wp); // :475-478
}

// Stream compaction: thread-local keep flag -> workgroup prefix -> ONE atomic per workgroup
// (a single counter word serialises at ~11 ns per atomic: one per wave would dominate the kernel).
__global__ __launch_bounds__(kBlock) void k_process_records(uint64_t num_rays, int32_t max_depth,
                                                            const float *__restrict__ l_final,
                                                            pg_dense_records r, pg_records_out o,
                                                            uint32_t *__restrict__ d_count)
{
	__shared__ uint32_t s_wave[kBlock / 64];
	__shared__ uint32_t s_base;
	const uint64_t S = num_rays * (uint64_t)max_depth;
	const uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	float radiance = 0.0f, nee_lum = 0.0f, wp = 0.0f;
	// slot = ray*max_depth + depth, as the reference lays the buffer out (:318)
	const bool keep = g < S && process_slot(g, S, num_rays, g / (uint64_t)max_depth, r.active[g] != 0, l_final, r,
	                                        radiance, nee_lum, wp);
	const unsigned long long mask = __ballot(keep);
	const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	if (lane == 0) s_wave[wid] = (uint32_t)__popcll(mask);
	__syncthreads();
	uint32_t off = 0, tot = 0;
#pragma unroll
	for (unsigned w = 0; w < kBlock / 64; ++w) {
		if (w < wid) off += s_wave[w];
		tot += s_wave[w];
	}
	if (threadIdx.x == 0) s_base = tot ? atomicAdd(d_count, tot) : 0;
	__syncthreads();
	if (keep) {
		const uint64_t k = (uint64_t)s_base + off +
		                   __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
		o.position[k] = r.position[g];
		o.position[S + k] = r.position[S + g];
		o.position[2 * S + k] = r.position[2 * S + g];
		o.direction[k] = r.direction[g];
		o.direction[S + k] = r.direction[S + g];
		o.direction_nee[k] = r.direction_nee[g];
		o.direction_nee[S + k] = r.direction_nee[S + g];
		o.radiance[k] = radiance;
		o.wo_pdf[k] = wp;
		o.radiance_nee_lum[k] = nee_lum;
	}
}

// The reference's dense record buffer behind pg_process_and_splat: slot = ray*max_depth + depth (:318), every slot
// visited.  (The library's own renderer does not come here: its list names the accumulators, k_splat_list below.)
constexpr uint32_t kNoRay = 0xffffffffu; // k_splat_list: the path left the scene at this entry

// A fixed grid (kSplatGroupsPerCu workgroups per CU, enough of them to even out tiles of unequal cost) striding
// over the 256-entry tiles: the jump grid's planes are staged once per workgroup instead of once per tile, and the
// first iteration's single accumulator gets four atomics per workgroup.
__global__ __launch_bounds__(kBlock) void k_process_and_splat(TreeView t, AccumView a, int store_nee,
                                                              uint64_t num_rays, int32_t max_depth,
                                                              const float *__restrict__ l_final,
                                                              pg_dense_records r, DepthCounters *dc)
{
	__shared__ float s_planes[3 * kKdGridPlanes];
	const uint4 *s_kd = reinterpret_cast<const uint4 *>(s_planes); // (plan_record's parameter: the staged table of whichever kind)
	__shared__ long long s_val[kBlock * 4];
	__shared__ unsigned long long s_ptr[kBlock];
	const uint64_t S = num_rays * (uint64_t)max_depth;
	const uint64_t total = S;
	stage_kd_planes(s_planes, t);
	// First iteration: one KD leaf owning a single-leaf quadtree (kdtree.py:122, quadtree.py:355), so
	// every record of the pass lands in the same accumulator.  One word takes ~11 ns per atomic;
	// sum inside the workgroup and send four atomics per workgroup instead of two per record.
	const bool single = t.n_rec == 0 && t.n_trees == 1;
	long long v[4] = {0, 0, 0, 0};
	unsigned kd_lv = 0, q_lv = 0, q_q = 0, did = 0, st_bytes = 0;
	for (uint64_t base = (uint64_t)blockIdx.x * kBlock; base < total; base += (uint64_t)gridDim.x * kBlock) {
		const uint64_t g = base + threadIdx.x;
		float radiance = 0.0f, nee_lum = 0.0f, wp = 0.0f;
		SlotAdd path = {nullptr, 0, 0, 0, 0}, nee = {nullptr, 0, 0, 0, 0};
		bool keep = false;
		if (g < total) keep = process_slot(g, S, num_rays, g / (uint64_t)max_depth, r.active[g] != 0, l_final, r, radiance, nee_lum, wp);
		if (keep) {
			plan_record(t, a, s_kd, store_nee, r.position[g], r.position[S + g], r.position[2 * S + g], r.direction[g],
			            r.direction[S + g], radiance, wp, r.direction_nee[g], r.direction_nee[S + g], nee_lum, path, nee,
			            kd_lv, q_lv, q_q, st_bytes);
			++did;
		}
		if (single) {
			if (path.ptr) { v[0] += path.w0; v[1] += path.w1; v[2] += path.w2; v[3] += path.w3; }
			if (nee.ptr) { v[0] += nee.w0; v[1] += nee.w1; v[2] += nee.w2; v[3] += nee.w3; }
		} else {
			coop_add(path, s_val, s_ptr);
			if (store_nee) coop_add(nee, s_val, s_ptr);
		}
	}
	if (single) {
		__syncthreads();
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const unsigned long long sw = wave_sum_s((unsigned long long)v[k]);
			if ((threadIdx.x & 63) == 0) s_val[(threadIdx.x >> 6) * 4 + k] = (long long)sw;
		}
		__syncthreads();
		if (threadIdx.x < 4) {
			unsigned long long tot = 0;
			for (int w = 0; w < kBlock / 64; ++w) tot += (unsigned long long)s_val[w * 4 + threadIdx.x];
			if (tot) atomicAdd(reinterpret_cast<unsigned long long *>(a.root_acc + threadIdx.x), tot);
		}
	}
	count_depths_s(dc, kd_lv, did, q_lv, q_q, st_bytes);
}

// The split render pipeline's list (pg_list_records): processPathData + the filter of scatterDataIntoSDTree
// (path_guiding_integrator.py:434-478) + the adds of KDTree / QuadTree.addDataPropagate (kdtree.py:180-225,
// quadtree.py:389-464) at accumulators the entry names -- no tree is walked here: the bounce that made the vertex
// walked sdTree_prev to these very leaves for its pdfs (stage_guide).  What is left is one gather (the path's
// final radiance), the division chain of :434-453, and the atomics.
__global__ __launch_bounds__(kBlock) void k_splat_list(TreeView t, AccumView a, int store_nee, uint64_t num_rays,
                                                       int32_t max_depth, const float *__restrict__ l_final,
                                                       const uint4 *__restrict__ l_final_q, pg_list_records r,
                                                       const uint32_t *__restrict__ live_count)
{
	__shared__ long long s_val[kBlock * 4];
	__shared__ unsigned long long s_ptr[kBlock];
	const uint64_t S = num_rays * (uint64_t)max_depth;
	uint64_t total = num_rays; // the first bounce visits every path, bounce b+1 the survivors of bounce b
	for (int b = 0; b + 1 < max_depth; ++b) total += live_count[b];
	if ((uint64_t)blockIdx.x * kBlock >= total) return;
	const bool single = t.n_rec == 0 && t.n_trees == 1; // (first iteration: see k_process_and_splat)
	long long v[4] = {0, 0, 0, 0};
	for (uint64_t base = (uint64_t)blockIdx.x * kBlock; base < total; base += (uint64_t)gridDim.x * kBlock) {
		const uint64_t g = base + threadIdx.x;
		SlotAdd path = {nullptr, 0, 0, 0, 0}, nee = {nullptr, 0, 0, 0, 0};
		if (g < total) {
			// (the list is read once, front to back: streaming loads keep it from pushing the accumulators out of L2 --
			// 8.0 -> 7.65 ms per step)
#define PG_LD(p) __builtin_nontemporal_load(p)
			const uint32_t ray = PG_LD(r.ray_of + g);
			if (ray != kNoRay) {
				float in[3], lf[3];
				if (l_final_q) { // (the split pipeline keeps a path's final radiance as one 16-byte entry: one gather)
					const uint4 q = l_final_q[ray];
					lf[0] = __uint_as_float(q.x); lf[1] = __uint_as_float(q.y); lf[2] = __uint_as_float(q.z);
				} else {
					lf[0] = l_final[ray]; lf[1] = l_final[num_rays + ray]; lf[2] = l_final[2 * num_rays + ray];
				}
#pragma unroll
				for (int ch = 0; ch < 3; ++ch) {
					float out = (lf[ch] - PG_LD(r.throughput_radiance + ch * S + g)) / PG_LD(r.throughput_bsdf + ch * S + g);
					if (out != out) out = 0.0f;                         // :444
					float q = out / PG_LD(r.bsdf + ch * S + g);
					if (q != q) q = 0.0f;                               // :449
					in[ch] = q;
				}
				float radiance = luminance(in[0], in[1], in[2]);       // :452
				if (radiance != radiance) radiance = 0.0f;             // :466
				const float nee_lum = PG_LD(r.nee_lum + g);
				const float wp = PG_LD(r.wo_pdf + g);
				const bool both_zero = (radiance == 0.0f) && (nee_lum == 0.0f); // :470-472
				if (!both_zero && !(wp == 0.0f) && !(wp != wp)) {      // :475-478
					const uint2 sl = r.slot[g];
					const uint32_t tf = PG_LD(r.tree + g), tree = tf & 0x7fffffffu;
#undef PG_LD
					const bool inside = (tf >> 31) != 0u;
					const float w = wp > 0.0f ? radiance / wp : 0.0f;  // quadtree.py:451
					const float wn = wp > 0.0f ? nee_lum / wp : 0.0f;  // quadtree.py:462
					if (sl.x != kSlotNone) {
						const Limbs q = quantize_weight(w);
						path.ptr = sl.x == kSlotRoot ? a.root_acc + (size_t)kAccWords * tree : a.rec_acc + (size_t)kAccWords * sl.x;
						path.w0 = q.l0; path.w1 = q.l1; path.w2 = q.l2; path.w3 = inside ? 1 : 0;
					} else if (inside) {
						atomicAdd(a.leaf_count + tree, 1ull); // counted, but its direction reaches no leaf
					}
					if (store_nee && sl.y != kSlotNone) {
						const Limbs q = quantize_weight(wn);
						nee.ptr = sl.y == kSlotRoot ? a.root_acc + (size_t)kAccWords * tree : a.rec_acc + (size_t)kAccWords * sl.y;
						nee.w0 = q.l0; nee.w1 = q.l1; nee.w2 = q.l2; nee.w3 = 0;
					}
				}
			}
		}
		if (single) {
			if (path.ptr) { v[0] += path.w0; v[1] += path.w1; v[2] += path.w2; v[3] += path.w3; }
			if (nee.ptr) { v[0] += nee.w0; v[1] += nee.w1; v[2] += nee.w2; v[3] += nee.w3; }
		} else {
			coop_add(path, s_val, s_ptr);
			if (store_nee) coop_add(nee, s_val, s_ptr);
		}
	}
	if (single) {
		__syncthreads();
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const unsigned long long sw = wave_sum_s((unsigned long long)v[k]);
			if ((threadIdx.x & 63) == 0) s_val[(threadIdx.x >> 6) * 4 + k] = (long long)sw;
		}
		__syncthreads();
		if (threadIdx.x < 4) {
			unsigned long long tot = 0;
			for (int w = 0; w < kBlock / 64; ++w) tot += (unsigned long long)s_val[w * 4 + threadIdx.x];
			if (tot) atomicAdd(reinterpret_cast<unsigned long long *>(a.root_acc + threadIdx.x), tot);
		}
	}
}

constexpr unsigned kSplatGroupsPerCu = 64; // measured (round 2): 8 -> 847, 16 -> 798, 32 -> 746, 64 -> 726, 128 -> 755 us on cornell-box

static inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

void launch_splat(const TreeView &t, const AccumView &a, int store_nee, uint64_t m, const pg_records &rec,
                  const uint32_t *d_count, DepthCounters *dc, hipStream_t s)
{
	if (m == 0) return;
	hipLaunchKernelGGL(k_splat, grid_for(m), dim3(kBlock), 0, s, t, a, store_nee, m, rec.position,
	                   rec.direction, rec.radiance, rec.wo_pdf, rec.direction_nee, rec.radiance_nee_lum,
	                   d_count, dc);
}

void launch_process_records(uint64_t num_rays, int32_t max_depth, const float *l_final,
                            const pg_dense_records &rec, const pg_records_out &out, uint32_t *d_count,
                            hipStream_t s)
{
	const uint64_t S = num_rays * (uint64_t)max_depth;
	(void)hipMemsetAsync(d_count, 0, sizeof(uint32_t), s);
	if (S == 0) return;
	hipLaunchKernelGGL(k_process_records, grid_for(S), dim3(kBlock), 0, s, num_rays, max_depth, l_final, rec,
	                   out, d_count);
}

// a fixed grid striding over the tiles (the list's length is known only on the device)
static dim3 strided_grid(uint64_t S)
{
	static int cus[64] = {0};
	int dev = 0;
	(void)hipGetDevice(&dev);
	int n_cu = (dev >= 0 && dev < 64) ? cus[dev] : 0;
	if (n_cu <= 0) {
		if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
		if (dev >= 0 && dev < 64) cus[dev] = n_cu;
	}
	const uint64_t tiles = (S + kBlock - 1) / kBlock, cap = (uint64_t)n_cu * kSplatGroupsPerCu;
	return dim3((unsigned)(tiles < cap ? tiles : cap));
}

void launch_splat_list(const TreeView &t, const AccumView &a, int store_nee, uint64_t num_rays, int32_t max_depth,
                       const float *l_final, const uint4 *l_final_q, const pg_list_records &rec, const uint32_t *live_count,
                       hipStream_t s)
{
	const uint64_t S = num_rays * (uint64_t)max_depth;
	if (S == 0) return;
	hipLaunchKernelGGL(k_splat_list, strided_grid(S), dim3(kBlock), 0, s, t, a, store_nee, num_rays, max_depth, l_final, l_final_q,
	                   rec, live_count);
}

void launch_process_and_splat(const TreeView &t, const AccumView &a, int store_nee, uint64_t num_rays,
                              int32_t max_depth, const float *l_final, const pg_dense_records &rec,
                              DepthCounters *dc, hipStream_t s)
{
	const uint64_t S = num_rays * (uint64_t)max_depth;
	if (S == 0) return;
	hipLaunchKernelGGL(k_process_and_splat, strided_grid(S), dim3(kBlock), 0, s, t, a, store_nee, num_rays, max_depth, l_final,
	                   rec, dc);
}

} // namespace pg
