// pg_kernels_query.hip -- sdTree_prev queries: one lane per ray, wave64, 256-thread groups.
//
// Bound: HBM/L2 latency of dependent 16-B (KD) and 32-B (quadtree) gathers; no reuse inside a
// lane, so occupancy (registers <= 64 -> 8 waves/SIMD) is what hides the latency.
#include "pg_descent.hpp"
#include "pg_kernels.hpp"

namespace pg {

constexpr int kBlock = 256;
static_assert(kBlock == kStageThreads, "stage_kd_planes copies one plane per thread");

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
	// (kd_lv, q_lv: sums of statistics words, pg_descent.hpp)
	const unsigned long long a = wave_sum(stat_levels(kd_lv)), b = wave_sum(kd_q), c = wave_sum(stat_levels(q_lv)), d = wave_sum(q_q);
	const unsigned long long e = wave_sum(stat_bytes(kd_lv) + stat_bytes(q_lv));
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&dc->kd_levels, a);
		atomicAdd(&dc->kd_queries, b);
		atomicAdd(&dc->quad_levels, c);
		atomicAdd(&dc->quad_queries, d);
		atomicAdd(&dc->layout_bytes, e);
	}
}

__device__ __forceinline__ TreeHead load_head(const TreeHead *h, uint32_t t)
{
	const uint2 v = gather8(h + t);
	TreeHead r;
	r.root_rec = v.x;
	r.root_irr = __uint_as_float(v.y);
	return r;
}

__global__ __launch_bounds__(kBlock) void k_leaf_index(TreeView t, uint64_t n, const float *__restrict__ p,
                                                       const uint8_t *__restrict__ active,
                                                       uint32_t *__restrict__ node_out)
{
	__shared__ float s_planes[3 * kKdGridPlanes];
	stage_kd_planes(s_planes, t);
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const float x = p[i], y = p[n + i], z = p[2 * n + i];
	const bool act = active ? active[i] != 0 : true;
	KdNode leaf;
	uint32_t lv;
	node_out[i] = kd_descend_grid(t, s_planes, x, y, z, act && inside_root(t, x, y, z), leaf, lv);
}

__global__ __launch_bounds__(kBlock) void k_sample(TreeView t, uint64_t n, const float *__restrict__ p,
                                                   uint64_t *__restrict__ rng_state,
                                                   const uint64_t *__restrict__ rng_inc,
                                                   const uint8_t *__restrict__ active,
                                                   float *__restrict__ dir_out, float *__restrict__ pdf_out,
                                                   DepthCounters *dc)
{
	__shared__ float s_planes[3 * kKdGridPlanes];
	stage_kd_planes(s_planes, t);
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	unsigned kd_lv = 0, q_lv = 0, did = 0;
	if (i < n) {
		const bool act = active ? active[i] != 0 : true;
		float dx = 0.0f, dy = 0.0f, dz = -1.0f, pdf = 1.0f; // inactive lanes (quadtree.py:940, 1011)
		if (act) {
			const float x = p[i], y = p[n + i], z = p[2 * n + i];
			KdNode leaf;
			kd_descend_grid(t, s_planes, x, y, z, inside_root(t, x, y, z), leaf, kd_lv);
			Pcg32 rng = {rng_state[i], rng_inc[i]};
			quad_sample(t.rec, t.jump, leaf.tree, load_head(t.head, leaf.tree), rng, dx, dy, dz, pdf, q_lv);
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
	__shared__ float s_planes[3 * kKdGridPlanes];
	stage_kd_planes(s_planes, t);
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	unsigned kd_lv = 0, q_lv = 0, did = 0;
	if (i < n) {
		const bool act = active ? active[i] != 0 : true;
		float pdf = 1.0f;
		if (act) {
			const float x = p[i], y = p[n + i], z = p[2 * n + i];
			float cx, cy;
			dir_to_canonical(dir[i], dir[n + i], dir[2 * n + i], cx, cy);
			KdNode leaf;
			kd_descend_grid(t, s_planes, x, y, z, inside_root(t, x, y, z), leaf, kd_lv);
			// (the tree's head and the walk's jump-table entry need the leaf's tree number only: two gathers in flight at once)
			const TreeHead head = load_head(t.head, leaf.tree);
			const JumpPre pre = jump_prefetch(t.jump, leaf.tree, cx, cy, true);
			uint32_t slot;
			pdf = quad_pdf_pre<false>(t.rec, head, cx, cy, pre, q_lv, slot);
			did = 1;
		}
		pdf_out[i] = pdf;
	}
	count_depths(dc, kd_lv, did, q_lv, did);
}

// One KD descent shared by the NEE pdf and the sample-or-pdf of the continuation direction
// (path_guiding_integrator.py:244, 301, 307 all pass the same si.p).
//
// lane_index/d_lane_count (optional): a compacted list of live ray slots produced by
// k_compact_lanes.  Thread i then serves ray lane_index[i]; threads past *d_lane_count retire at
// once, so a late bounce with few live rays costs few waves without a host round trip.
__global__ __launch_bounds__(kBlock) void k_guide_bounce(TreeView t, uint64_t n, const float *__restrict__ p,
                                                         const float *__restrict__ dir_nee,
                                                         const uint8_t *__restrict__ nee_active,
                                                         const uint8_t *__restrict__ select,
                                                         float *__restrict__ dir_io,
                                                         uint64_t *__restrict__ rng_state,
                                                         const uint64_t *__restrict__ rng_inc,
                                                         float *__restrict__ pdf_nee_out,
                                                         float *__restrict__ pdf_out,
                                                         const uint32_t *__restrict__ lane_index,
                                                         const uint32_t *__restrict__ d_lane_count,
                                                         DepthCounters *dc)
{
	__shared__ float s_planes[3 * kKdGridPlanes];
	const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	// compacted list: [0, front) from the start of lane_index (sample lanes), the next `back`
	// entries from its end, walking down (pdf-only lanes): waves are homogeneous in `select`
	const uint64_t front = lane_index ? (uint64_t)d_lane_count[0] : n;
	const uint64_t live = lane_index ? front + (uint64_t)d_lane_count[1] : n;
	if ((uint64_t)blockIdx.x * kBlock >= live) return; // whole workgroup idle (uniform)
	stage_kd_planes(s_planes, t);
	unsigned kd_lv = 0, kd_q = 0, q_lv = 0, q_q = 0;
	if (tid < live && tid < n) {
		const uint64_t i = lane_index ? (uint64_t)lane_index[tid < front ? tid : n - 1 - (tid - front)] : tid;
		const bool nee = nee_active ? nee_active[i] != 0 : true;
		const int sel = select ? (int)select[i] : 2;
		float pdf_nee = 1.0f, pdf = 1.0f;
		if (nee || sel != 0) {
			const float x = p[i], y = p[n + i], z = p[2 * n + i];
			// (the canonical forms first: the tree's head and the jump-table entries of the two pdf walks then need the leaf's
			// tree number only and leave together, three gathers in flight at once)
			float ncx = 0.0f, ncy = 0.0f, wcx = 0.0f, wcy = 0.0f;
			if (nee) dir_to_canonical(dir_nee[i], dir_nee[n + i], dir_nee[2 * n + i], ncx, ncy);
			if (sel == 1) dir_to_canonical(dir_io[i], dir_io[n + i], dir_io[2 * n + i], wcx, wcy);
			KdNode leaf;
			kd_descend_grid(t, s_planes, x, y, z, inside_root(t, x, y, z), leaf, kd_lv);
			kd_q = 1;
			const TreeHead head = load_head(t.head, leaf.tree);
			const JumpPre pre_nee = jump_prefetch(t.jump, leaf.tree, ncx, ncy, nee);
			const JumpPre pre_wo = jump_prefetch(t.jump, leaf.tree, wcx, wcy, sel == 1);
			if (nee) {
				uint32_t lv, slot;
				pdf_nee = quad_pdf_pre<false>(t.rec, head, ncx, ncy, pre_nee, lv, slot);
				q_lv += lv;
				++q_q;
			}
			if (sel == 2) {
				Pcg32 rng = {rng_state[i], rng_inc[i]};
				float dx, dy, dz;
				uint32_t lv;
				quad_sample(t.rec, t.jump, leaf.tree, head, rng, dx, dy, dz, pdf, lv);
				rng_state[i] = rng.state;
				dir_io[i] = dx;
				dir_io[n + i] = dy;
				dir_io[2 * n + i] = dz;
				q_lv += lv;
				++q_q;
			} else if (sel == 1) {
				uint32_t lv, slot;
				pdf = quad_pdf_pre<false>(t.rec, head, wcx, wcy, pre_wo, lv, slot);
				q_lv += lv;
				++q_q;
			}
		}
		pdf_nee_out[i] = pdf_nee;
		pdf_out[i] = pdf;
	}
	count_depths(dc, kd_lv, kd_q, q_lv, q_q);
}

// Active-ray stream compaction, partitioned by what the lane will do: lanes that sample the tree
// (select == 2) are packed from the front of idx_out, the other live lanes (select == 1, or only
// an NEE pdf) from its back, so the bounce kernel's waves do not mix the two code paths.
// One workgroup owns kCompactTile consecutive lanes (16 per thread, one 16-B load of each mask):
// thread-local counts -> wave prefix (shuffles) -> workgroup prefix (LDS) -> ONE atomic pair per
// workgroup.  A single counter word takes ~11 ns per atomic (MI355X_MICROARCH.md, "dequeue"), so
// one atomic per wave would cost 45 us at 2^18 lanes; per 4096 lanes it is < 1 us.
constexpr int kCompactItems = 16;
constexpr int kCompactTile = kBlock * kCompactItems;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
	const unsigned lane = threadIdx.x & 63;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(v, o, 64);
		if ((int)lane >= o) v += t;
	}
	return v;
}

__global__ __launch_bounds__(kBlock) void k_compact_lanes(uint64_t n, const uint8_t *__restrict__ select,
                                                          const uint8_t *__restrict__ nee_active,
                                                          uint32_t *__restrict__ idx_out,
                                                          uint32_t *__restrict__ d_count /*[2]*/)
{
	__shared__ uint32_t s_wave[2][kBlock / 64];
	__shared__ uint32_t s_base[2];
	const uint64_t base = (uint64_t)blockIdx.x * kCompactTile + (uint64_t)threadIdx.x * kCompactItems;
	alignas(16) uint8_t sel[kCompactItems];
	alignas(16) uint8_t nee[kCompactItems];
	if (base + kCompactItems <= n) {
		*reinterpret_cast<uint4 *>(sel) = *reinterpret_cast<const uint4 *>(select + base);
		if (nee_active) *reinterpret_cast<uint4 *>(nee) = *reinterpret_cast<const uint4 *>(nee_active + base);
	} else {
#pragma unroll
		for (int k = 0; k < kCompactItems; ++k) {
			sel[k] = base + k < n ? select[base + k] : 0;
			nee[k] = (nee_active && base + k < n) ? nee_active[base + k] : 0;
		}
	}
	uint32_t front_bits = 0, back_bits = 0;
#pragma unroll
	for (int k = 0; k < kCompactItems; ++k) {
		const bool f = sel[k] == 2;
		const bool b = !f && (sel[k] != 0 || (nee_active && nee[k] != 0));
		front_bits |= (uint32_t)f << k;
		back_bits |= (uint32_t)b << k;
	}
	const uint32_t nf = __popc(front_bits), nb = __popc(back_bits);
	const uint32_t inf = wave_inclusive_scan(nf), inb = wave_inclusive_scan(nb);
	const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	if (lane == 63) { s_wave[0][wid] = inf; s_wave[1][wid] = inb; }
	__syncthreads();
	uint32_t off_f = inf - nf, off_b = inb - nb, tot_f = 0, tot_b = 0;
#pragma unroll
	for (unsigned w = 0; w < kBlock / 64; ++w) {
		if (w < wid) { off_f += s_wave[0][w]; off_b += s_wave[1][w]; }
		tot_f += s_wave[0][w];
		tot_b += s_wave[1][w];
	}
	if (threadIdx.x == 0) {
		s_base[0] = tot_f ? atomicAdd(&d_count[0], tot_f) : 0;
		s_base[1] = tot_b ? atomicAdd(&d_count[1], tot_b) : 0;
	}
	__syncthreads();
	uint32_t wf = s_base[0] + off_f;
	uint32_t wb = (uint32_t)(n - 1) - (s_base[1] + off_b);
#pragma unroll
	for (int k = 0; k < kCompactItems; ++k) {
		if ((front_bits >> k) & 1u) idx_out[wf++] = (uint32_t)(base + k);
		if ((back_bits >> k) & 1u) idx_out[wb--] = (uint32_t)(base + k);
	}
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

// The jump table of one quadtree per workgroup, built TOP-DOWN in LDS: the entry of a cell of the 2^l x 2^l grid is the
// state of pdfQuadTree's loop (quadtree.py:1020-1098) and of addIrradiancePropagate's walk (quadtree.py:398-441) after the l
// levels a point strictly inside the cell passes through; the four cells it splits into at level l + 1 follow from it and ONE
// read of the node's record, by the very operations of that loop in its order (so a table built this way equals, bit for
// bit, the one round 3 built by running the loop from the root for every cell's centre -- 4096 six-level descents per tree,
// 87 M dependent gather chains for the veach-ajar bench forest; here a tree's records are read once each, at most 1365 of
// them, and the 64 KB leave as whole lines).  A level-l entry lives at the table position of its cell's lowest corner, so
// the expansion is in place: a thread reads its parent, then overwrites it with child 3 (the quadrant at the cell's own
// corner) and writes the three others to positions no level-l entry occupies.
// Quadrants (quadtree.py:153-175; `quadrant` in pg_descent.hpp): child 0 = (x >= mid, y >= mid), 1 = (x <= mid, y >= mid),
// 2 = (x <= mid, y <= mid), 3 = (x >= mid, y <= mid).
__global__ __launch_bounds__(kBlock) void k_build_jump(TreeView t, QuadJump *__restrict__ out, int bits)
{
	extern __shared__ u32x4_t s_tab[]; // 4^bits entries
	const uint32_t tree = blockIdx.x;
	const uint32_t cells = 1u << (2 * bits);
	if (threadIdx.x == 0) {
		const TreeHead head = load_head(t.head, tree);
		u32x4_t e;
		if (head.root_rec == kNoRecord) {
			// the root is a leaf: the walk ends in the table like at any other leaf within it, with the ROOT's accumulator
			// for a slot (bit 31) -- a pdf walk that hits the table never needs the tree's head
			e.x = kNoRecord; e.y = __float_as_uint(kInvFourPiF); e.z = __float_as_uint(head.root_irr); e.w = 0x80000000u;
		} else {
			e.x = head.root_rec; e.y = __float_as_uint(1.0f); e.z = __float_as_uint(head.root_irr); e.w = 0u;
		}
		s_tab[0] = e;
	}
	__syncthreads();
	for (int l = 0; l < bits; ++l) {
		const uint32_t n_par = 1u << (2 * l);
		const int shift = bits - l;            // a level-l cell spans 2^shift table cells per axis
		const uint32_t half = 1u << (shift - 1);
		for (uint32_t p = threadIdx.x; p < n_par; p += kBlock) {
			const uint32_t iy = p >> l, ix = p & ((1u << l) - 1u);
			const uint32_t pos = ((iy << shift) << bits) | (ix << shift);
			const u32x4_t e = s_tab[pos];
			u32x4_t c[4] = {e, e, e, e}; // (a walk that has ended: its entry covers the whole cell)
			if (e.x != kNoRecord) {
				const QuadLoad q = load_rec(t.rec, e.x);
				const float pdf = __uint_as_float(e.y), node_irr = __uint_as_float(e.z);
				const uint32_t levels = ((e.w >> 26) & 15u) + 1u;
				const bool was_dead = ((e.w >> 30) & 1u) != 0u;
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const float child_irr = sel4f(k, q.i0, q.i1, q.i2, q.i3);
					float pk = pdf * ((4.0f * child_irr) / node_irr);
					const bool dead = was_dead || pk != pk; // quadtree.py:1090-1092 ends the pdf loop here; the splat's walk goes on
					const uint32_t ch = sel4u(k, q.c0, q.c1, q.c2, q.c3);
					uint32_t slot = 0;
					if (ch == 0) { // the child is a leaf
						slot = (e.x * 4u + (uint32_t)k) & kJumpSlotMask;
						pk = pk * kInvFourPiF;
					}
					c[k].x = ch == 0 ? kNoRecord : ch;
					c[k].y = __float_as_uint(pk);
					c[k].z = __float_as_uint(child_irr);
					c[k].w = slot | (levels << 26) | (dead ? (1u << 30) : 0u);
				}
			}
			// (dx, dy) of child k: 0 -> (1, 1), 1 -> (0, 1), 2 -> (0, 0), 3 -> (1, 0)
			s_tab[pos + (half << bits) + half] = c[0];
			s_tab[pos + (half << bits)] = c[1];
			s_tab[pos + half] = c[3];
			s_tab[pos] = c[2];
		}
		__syncthreads();
	}
	u32x4_t *dst = reinterpret_cast<u32x4_t *>(out + ((size_t)tree << (2 * bits)));
	for (uint32_t i = threadIdx.x; i < cells; i += kBlock) dst[i] = s_tab[i];
}

// One thread per cell of the KD jump grid: descend with the cell's interval for as long as every point
// strictly inside the cell takes the same branch.  Two more threads write the ROOT entries behind the
// cells (kd_descend_grid): the root as a searching lane that is in no cell starts from it, and the root
// as a lane that does not search returns it.
__global__ __launch_bounds__(kBlock) void k_build_kd_grid(TreeView t, KdGridEntry *__restrict__ out)
{
	const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t n_cells = 1u << (3 * t.grid_bits);
	if (c >= n_cells + kKdGridRootEntries) return;
	uint32_t node = 0, levels = 0;
	KdNode nd = load_kd(t.kd, 0);
	if (c < n_cells) {
		const uint32_t M = (1u << t.grid_bits) - 1u;
		const int idx[3] = {(int)(c & M), (int)((c >> t.grid_bits) & M), (int)(c >> (2 * t.grid_bits))};
		float lo[3], hi[3];
		for (int a = 0; a < 3; ++a) {
			lo[a] = t.kd_planes[a * kKdGridPlanes + idx[a]];
			hi[a] = t.kd_planes[a * kKdGridPlanes + idx[a] + 1];
		}
		for (int it = 0; it < kMaxLevels && nd.child != 0; ++it) {
			const uint32_t axis = nd.axis_depth & 3u;
			const float l = axis == 0 ? lo[0] : (axis == 1 ? lo[1] : lo[2]), h = axis == 0 ? hi[0] : (axis == 1 ? hi[1] : hi[2]);
			uint32_t next;
			if (l >= nd.split) next = nd.child + 1u;      // every v > l is >= split: right (kdtree.py:462-468)
			else if (h <= nd.split) next = nd.child;       // every v < h is < split: left
			else break;                                    // the plane cuts through the cell: queries go on from here
			node = next;
			nd = load_kd(t.kd, node);
			++levels;
		}
	}
	KdGridEntry e;
	e.node = node;
	e.meta = (levels << 16) | (nd.axis_depth & 0xffffu);
	e.child = nd.child;
	e.value = (nd.child == 0 || c == n_cells + 1u) ? nd.tree : __float_as_uint(nd.split);
	out[c] = e;
}

static inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

void launch_build_kd_grid(const TreeView &t, KdGridEntry *out, hipStream_t s)
{
	hipLaunchKernelGGL(k_build_kd_grid, grid_for((1ull << (3 * t.grid_bits)) + kKdGridRootEntries), dim3(kBlock), 0, s, t, out);
}

// false: the device refuses the dynamic LDS a table of this resolution is built in (nothing was launched; the caller
// takes a coarser table)
bool launch_build_jump(const TreeView &t, QuadJump *out, int bits, hipStream_t s)
{
	if (t.n_trees == 0) return true;
	const size_t lds = sizeof(QuadJump) << (2 * bits); // 64 KB for the finest table
	// (more than the default 48 KB of dynamic LDS has to be asked for; per device, so it is asked every time -- once per refine)
	if (lds > 48 * 1024 &&
	    hipFuncSetAttribute(reinterpret_cast<const void *>(k_build_jump), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	hipLaunchKernelGGL(k_build_jump, dim3(t.n_trees), dim3(kBlock), lds, s, t, out, bits);
	return true;
}

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
                         float *pdf_out, const uint32_t *lane_index, const uint32_t *d_lane_count,
                         DepthCounters *dc, hipStream_t s)
{
	if (n == 0) return;
	hipLaunchKernelGGL(k_guide_bounce, grid_for(n), dim3(kBlock), 0, s, t, n, p, dir_nee, nee_active,
	                   select, dir_io, rng_state, rng_inc, pdf_nee_out, pdf_out, lane_index, d_lane_count, dc);
}

void launch_compact_lanes(uint64_t n, const uint8_t *select, const uint8_t *nee_active, uint32_t *idx_out,
                          uint32_t *d_count, hipStream_t s)
{
	(void)hipMemsetAsync(d_count, 0, 2 * sizeof(uint32_t), s);
	if (n == 0) return;
	hipLaunchKernelGGL(k_compact_lanes, dim3((unsigned)((n + kCompactTile - 1) / kCompactTile)), dim3(kBlock), 0, s, n,
	                   select, nee_active, idx_out, d_count);
}

void launch_rng_seed(uint64_t n, uint32_t seed, uint32_t lane0, uint64_t *state, uint64_t *inc,
                     hipStream_t s)
{
	if (n == 0) return;
	hipLaunchKernelGGL(k_rng_seed, grid_for(n), dim3(kBlock), 0, s, n, seed, lane0, state, inc);
}

} // namespace pg
