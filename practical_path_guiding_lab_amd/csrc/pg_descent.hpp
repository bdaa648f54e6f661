// pg_descent.hpp -- per-lane SD-tree descents (device functions shared by the query and
// splat kernels).  Each function states the reference lines whose result it reproduces.
#pragma once

#include "pg_math.hpp"
#include "pg_tree.hpp"

namespace pg {

// ---- walk statistics ----
// Every walk reports ONE word, `levels`: bits 0-7 the levels of the reference's descent it stands for (what SURVEY 8d's
// algorithmic model prices), bits 8.. the BYTES this lane gathered from the tables of the built layout to get there (16 per
// KD grid entry or node, 16 per jump-table entry, 32 per quadtree record of a pdf or sampling walk, 16 per record of a
// leaf walk) -- nothing credited for lanes that share a line.  Callers sum the words of a lane's walks (at most three
// quadtree walks of at most 32 levels: the low byte cannot carry) and unpack them where the counters are added, in
// instrumented launches only (stat_levels / stat_bytes).
__device__ __forceinline__ uint32_t stat_word(uint32_t levels, uint32_t bytes) { return levels | (bytes << 8); }
__device__ __forceinline__ uint32_t stat_levels(uint32_t w) { return w & 0xffu; }
__device__ __forceinline__ uint32_t stat_bytes(uint32_t w) { return w >> 8; }

__device__ __forceinline__ KdNode load_kd(const KdNode *kd, uint32_t i)
{
	const uint4 v = gather16(kd + i);
	KdNode n;
	n.child = v.x;
	n.split = __uint_as_float(v.y);
	n.axis_depth = v.z;
	n.tree = v.w;
	return n;
}

__device__ __forceinline__ bool inside_root(const TreeView &t, float x, float y, float z)
{
	// mi.BoundingBox3f.contains, inclusive; NaN fails (kdtree.py:446-447)
	return x >= t.bmin[0] && x <= t.bmax[0] && y >= t.bmin[1] && y <= t.bmax[1] &&
	       z >= t.bmin[2] && z <= t.bmax[2];
}

// KDTree.getLeafNodeIndex (kdtree.py:435-470).  Inside a node, "left.contains then
// right.contains, right overwrites" is p[axis] >= mid ? right : left because the children
// share the parent's other four planes and meet at mid (kdtree.py:282-297).
// Returns the reference node index; `leaf` receives the node's packed record.
__device__ __forceinline__ uint32_t kd_descend(const KdNode *kd, float x, float y, float z,
                                               bool search, KdNode &leaf, uint32_t &levels)
{
	uint32_t node = 0;
	KdNode nd = load_kd(kd, 0);
	levels = 0;
	if (search) {
		for (int it = 0; it < kMaxLevels && nd.child != 0; ++it) {
			const uint32_t axis = nd.axis_depth & 3u;
			const float v = axis == 0 ? x : (axis == 1 ? y : z);
			node = nd.child + (v >= nd.split ? 1u : 0u);
			nd = load_kd(kd, node);
			++levels;
		}
	}
	levels = stat_word(levels, 16u * (levels + 1u)); // the root and one node per level
	leaf = nd;
	return node;
}

// ---- the KD jump grid (pg_tree.hpp) ----
// The planes of the grid staged in LDS once per workgroup (3 * kKdGridPlanes floats): every query reads six.
// Every kernel that stages them runs workgroups of kStageThreads threads (kBlock / kRBlock: static_asserts beside the callers), so a
// thread copies ONE plane (195 of 256 threads) -- written with `i += blockDim.x` the copy compiled to three loops (a vectorised
// one, its two remainders), each load -> wait -> store, at the head of every kernel that walks the KD tree (round 6, from the listings).
constexpr int kStageThreads = 256;
__device__ __forceinline__ void stage_kd_planes(float *s_planes, const TreeView &t)
{
	static_assert(3 * kKdGridPlanes <= kStageThreads, "one plane per thread");
	if (t.kd_grid != nullptr && threadIdx.x < 3u * kKdGridPlanes) s_planes[threadIdx.x] = t.kd_planes[threadIdx.x];
	__syncthreads();
}

// the cell a point lies strictly inside of, or false; `planes` = the staged copy
__device__ __forceinline__ bool kd_grid_cell(const TreeView &t, const float *planes, float x, float y, float z, uint32_t &cell)
{
	const int G = 1 << t.grid_bits;
	const float p[3] = {x, y, z};
	int idx[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		const float f = (p[a] - t.bmin[a]) * t.grid_inv[a];
		if (!(f >= 0.0f && f < (float)G)) return false;
		int i = (int)f;
		const float *P = planes + a * kKdGridPlanes;
		// the guess is rounded and the planes are bisection points, not multiples of a width: look next door too
		if (p[a] <= P[i]) --i;
		else if (p[a] >= P[i + 1]) ++i;
		if (i < 0 || i >= G) return false;
		if (!(p[a] > P[i] && p[a] < P[i + 1])) return false;
		idx[a] = i;
	}
	cell = ((uint32_t)idx[2] << (2 * t.grid_bits)) | ((uint32_t)idx[1] << t.grid_bits) | (uint32_t)idx[0];
	return true;
}

// KDTree.getLeafNodeIndex (kdtree.py:435-470) through the jump grid: the same node, leaf record and
// level count as kd_descend.
//
// EVERY lane makes exactly one gather from the table, at one load site: a lane strictly inside a cell
// reads the cell's entry; a lane on a cell face, outside the box or NaN reads one of the two ROOT
// entries the build appends behind the cells (k_build_kd_grid) -- [cells] the root as a searching lane
// starts from it, [cells + 1] the root as a non-searching lane returns it (node 0 with its stale
// quadtree, kdtree.py:446-447, 224).  There is no "grid entry or kd[0]" branch any more.  Round 2 had
// one, and ROCm 7.2's clang merged the two loads of the node's child word into a single load through
// a pointer phi (&entry.child | &kd[0].child) placed after the join (profiles/r03/kd_descend_isa/); a
// build of that form faulted in the fused splat kernel (a memory-aperture violation for lanes on a
// cell face) and was held together by an empty asm barrier.  One load site leaves nothing to merge.
__device__ __forceinline__ uint32_t kd_descend_grid(const TreeView &t, const float *planes, float x, float y, float z, bool search,
                                                    KdNode &leaf, uint32_t &levels)
{
	if (t.kd_grid == nullptr) return kd_descend(t.kd, x, y, z, search, leaf, levels); // (uniform: a degenerate root box has no grid)
	uint32_t cell = 0;
	const bool in_cell = search && kd_grid_cell(t, planes, x, y, z, cell);
	const uint32_t root_entry = (1u << (3 * t.grid_bits)) + (search ? 0u : 1u);
	const uint4 e = gather16(t.kd_grid + (in_cell ? cell : root_entry));
	uint32_t node = e.x;
	levels = e.y >> 16;
	const uint32_t from_grid = levels;
	KdNode nd;
	nd.child = e.z;
	nd.axis_depth = e.y & 0xffffu;
	nd.split = __uint_as_float(e.w); // (meaningful for an inner node)
	nd.tree = e.w;                   // (meaningful for a leaf and for the non-searching root entry)
	if (search) {
		for (int it = (int)levels; it < kMaxLevels && nd.child != 0; ++it) {
			const uint32_t axis = nd.axis_depth & 3u;
			const float v = axis == 0 ? x : (axis == 1 ? y : z);
			node = nd.child + (v >= nd.split ? 1u : 0u);
			nd = load_kd(t.kd, node);
			++levels;
		}
		if (nd.child != 0) nd = load_kd(t.kd, node); // (kMaxLevels deep without a leaf: not a tree this library holds)
	}
	levels = stat_word(levels, 16u + 16u * (levels - from_grid)); // the grid entry and one node per level below it
	leaf = nd;
	return node;
}

struct QuadLoad {
	float i0, i1, i2, i3;
	uint32_t c0, c1, c2, c3;
};

__device__ __forceinline__ QuadLoad load_rec(const QuadRec *rec, uint32_t r)
{
	const uint4 a = gather16(rec + r);
	const uint4 b = gather16(reinterpret_cast<const uint4 *>(rec + r) + 1);
	QuadLoad q;
	q.i0 = __uint_as_float(a.x); q.i1 = __uint_as_float(a.y);
	q.i2 = __uint_as_float(a.z); q.i3 = __uint_as_float(a.w);
	q.c0 = b.x; q.c1 = b.y; q.c2 = b.z; q.c3 = b.w;
	return q;
}

__device__ __forceinline__ float sel4f(int k, float a, float b, float c, float d)
{
	return k == 0 ? a : (k == 1 ? b : (k == 2 ? c : d));
}
__device__ __forceinline__ uint32_t sel4u(int k, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
	return k == 0 ? a : (k == 1 ? b : (k == 2 ? c : d));
}

// Which child quadrants of the cell [lo, lo+2h]^2 contain (cx,cy), inclusive on every edge
// (quadtree.py:153-175, 1043-1053).  `first` = lowest-numbered containing child (the energy
// pick, quadtree.py:1063-1075), `last` = highest-numbered (the descent, quadtree.py:424-438,
// 1095-1098).  Both are -1 when the point is outside the cell.
__device__ __forceinline__ void quadrant(float cx, float cy, float mx, float my, int &first, int &last)
{
	const bool xge = cx >= mx, xle = cx <= mx, yge = cy >= my, yle = cy <= my;
	const bool t0 = xge && yge, t1 = xle && yge, t2 = xle && yle, t3 = xge && yle;
	first = t0 ? 0 : (t1 ? 1 : (t2 ? 2 : (t3 ? 3 : -1)));
	last = t3 ? 3 : (t2 ? 2 : (t1 ? 1 : (t0 ? 0 : -1)));
}

// The cell of a 2^bits x 2^bits jump table a canonical position falls strictly inside of, or false (on a cell
// boundary, outside the unit square, NaN): 2^bits * c is exact in fp32, and so is every product with 2^-bits.
__device__ __forceinline__ bool jump_cell(int bits, float cx, float cy, uint32_t &cell, float &lox, float &loy)
{
	const float S = (float)(1 << bits), inv = 1.0f / S; // (uniform; a power of two: the reciprocal is exact)
	const float fx = cx * S, fy = cy * S;
	if (!(fx > 0.0f && fx < S && fy > 0.0f && fy < S)) return false;
	const float ix = __builtin_floorf(fx), iy = __builtin_floorf(fy);
	if (fx == ix || fy == iy) return false;
	cell = ((uint32_t)iy << bits) | (uint32_t)ix;
	lox = ix * inv;
	loy = iy * inv;
	return true;
}

// ---- accumulator slots ----
// sdTree_prev and sdTree_current share their topology (path_guiding_integrator.py:582), so the leaf a pdf or
// sampling walk of sdTree_prev ends in IS the leaf whose accumulator a record with that direction adds to in
// sdTree_current (quadtree.py:398-441 descends by the same highest-numbered containing child, quadtree.py:
// 424-438 / 1095-1098).  A walk can therefore hand the accumulator over as a by-product:
constexpr uint32_t kSlotNone = 0xffffffffu; // the direction reaches no leaf (outside the unit square, quadtree.py:404-405)
constexpr uint32_t kSlotRoot = 0x80000000u; // the tree's root is the leaf: its root accumulator (the tree is named beside the slot)
//                                             anything else: rec * 4 + child, the accumulator of a leaf below record `rec`
__device__ __forceinline__ bool in_unit_square(float cx, float cy) { return cx >= 0.0f && cx <= 1.0f && cy >= 0.0f && cy <= 1.0f; }

// The jump-table entry of a pdf walk, fetched AHEAD of the walk: the gather needs only the tree's number and the
// direction, so a caller that knows both issues it together with the gather of the tree's head -- one round trip to
// memory for the two -- instead of behind it (quad_pdf_pre looks at the head first, as the reference does).
struct JumpPre {
	uint4 e;
	float jx, jy;
	int bits; // levels the table covers (uniform)
	bool hit; // the direction lies strictly inside a cell of the table and the entry was fetched
};
__device__ __forceinline__ JumpPre jump_prefetch(JumpRef jump, uint32_t tree, float cx, float cy, bool wanted)
{
	JumpPre p;
	p.e = make_uint4(0u, 0u, 0u, 0u);
	p.jx = 0.0f; p.jy = 0.0f;
	p.bits = jump.bits;
	uint32_t cell = 0;
	p.hit = wanted && jump.p != nullptr && jump_cell(jump.bits, cx, cy, cell, p.jx, p.jy);
	if (p.hit) p.e = gather16(jump.p + (((size_t)tree << (2 * jump.bits)) + cell));
	return p;
}

// QuadTree.pdfQuadTree (quadtree.py:1001-1101) for canonical position (cx,cy) in [0,1]^2.
// pre: the entry of the quadtree's jump table for (cx,cy), if there is one (jump_prefetch; else every level is walked).
// kSlot: also the accumulator slot of the leaf that holds (cx,cy) -- the walk then goes on to the leaf where the
// pdf alone would stop (a 0/0 on the way, quadtree.py:1090-1092); the pdf is the same value either way.
template <bool kSlot>
__device__ __forceinline__ float quad_pdf_pre(const QuadRec *rec, TreeHead head, float cx, float cy, const JumpPre &pre,
                                              uint32_t &levels, uint32_t &slot)
{
	float pdf = 1.0f;
	levels = 0;
	slot = kSlotNone;
	const uint32_t tab_bytes = pre.hit ? 16u : 0u; // (the entry was gathered whether or not the walk can use it)
	const bool inside = in_unit_square(cx, cy);
	// The table first: its entries say everything about the levels they cover -- also that a tree's root is a leaf (bit 31,
	// k_build_jump) -- so a walk that hits the table reads the tree's head only if it has to go on below it, and the
	// entry's gather does not wait for the head's.
	uint32_t r = kNoRecord;
	float node_irr = 0.0f;
	float lox = 0.0f, loy = 0.0f, h = 0.5f;
	int it0 = 0;
	bool dead = false; // (kSlot) the product met a 0/0: the value is 0, the walk goes on for the slot
	bool from_table = false;
	if (pre.hit) {
		const uint4 e = pre.e;
		const float jx = pre.jx, jy = pre.jy;
		const bool undefined = ((e.w >> 30) & 1u) != 0u; // the product is not defined along this path: the loop finds out where
		if (!undefined || kSlot) {
			levels = (e.w >> 26) & 15u;
			if (e.x == kNoRecord) { // a leaf within the table (the root itself: bit 31): the final value
				if (kSlot) slot = (e.w >> 31) ? kSlotRoot : (e.w & kJumpSlotMask);
				levels = stat_word(levels, tab_bytes);
				return undefined ? 0.0f : __uint_as_float(e.y);
			}
			r = e.x;
			pdf = __uint_as_float(e.y);
			node_irr = __uint_as_float(e.z);
			dead = undefined;
			lox = jx; loy = jy;
			h = 0.5f / (float)(1 << pre.bits);
			it0 = pre.bits;
			from_table = true;
		}
	}
	if (!from_table) { // from the root (quadtree.py:1011-1019)
		if (head.root_rec == kNoRecord) {
			if (kSlot && inside) slot = kSlotRoot;
			levels = stat_word(0u, tab_bytes);
			return pdf * kInvFourPiF;
		}
		r = head.root_rec;
		node_irr = head.root_irr;
	}
	// (bytes: 32 per record loaded = per pass of the loop, it - it0 + 1 when leaving from inside it)
	for (int it = it0; it < kMaxLevels; ++it) {
		const QuadLoad q = load_rec(rec, r);
		const float mx = lox + h, my = loy + h;
		int first, last;
		quadrant(cx, cy, mx, my, first, last);
		if (!dead) {
			const float child_irr = first < 0 ? 0.0f : sel4f(first, q.i0, q.i1, q.i2, q.i3);
			pdf = pdf * ((4.0f * child_irr) / node_irr);
			if (pdf != pdf) {                     // quadtree.py:1090-1092
				if (!kSlot) { levels = stat_word(levels, tab_bytes + 32u * (uint32_t)(it - it0 + 1)); return 0.0f; }
				dead = true;
			}
		}
		if (last < 0) { // outside every child: the reference would spin
			levels = stat_word(levels, tab_bytes + 32u * (uint32_t)(it - it0 + 1));
			return dead ? 0.0f : pdf;
		}
		++levels;
		node_irr = sel4f(last, q.i0, q.i1, q.i2, q.i3);
		const uint32_t c = sel4u(last, q.c0, q.c1, q.c2, q.c3);
		if (last == 0 || last == 3) lox = mx;
		if (last == 0 || last == 1) loy = my;
		h *= 0.5f;
		if (c == 0) { // child is a leaf (quadtree.py:1025-1030)
			if (kSlot && inside) slot = r * 4u + (uint32_t)last;
			levels = stat_word(levels, tab_bytes + 32u * (uint32_t)(it - it0 + 1));
			return dead ? 0.0f : pdf * kInvFourPiF;
		}
		r = c;
	}
	levels = stat_word(levels, tab_bytes + 32u * (uint32_t)(kMaxLevels - it0));
	return dead ? 0.0f : pdf;
}

// the same with the table's entry fetched here (it does not wait for the head)
template <bool kSlot>
__device__ __forceinline__ float quad_pdf_t(const QuadRec *rec, JumpRef jump, uint32_t tree, TreeHead head,
                                            float cx, float cy, uint32_t &levels, uint32_t &slot)
{
	return quad_pdf_pre<kSlot>(rec, head, cx, cy, jump_prefetch(jump, tree, cx, cy, true), levels, slot);
}

__device__ __forceinline__ float quad_pdf(const QuadRec *rec, JumpRef jump, uint32_t tree, TreeHead head,
                                          float cx, float cy, uint32_t &levels)
{
	uint32_t slot;
	return quad_pdf_t<false>(rec, jump, tree, head, cx, cy, levels, slot);
}

// QuadTree.sampleQuadTree + the pdfQuadTree call of KDTree.sample (kdtree.py:483-484,
// quadtree.py:931-998).  Three uniforms are drawn per visited node, the leaf included
// (quadtree.py:956, 980).  The pdf along the sampled path is accumulated on the way down;
// it equals pdfQuadTree(dir) whenever the round trip dir -> canonical lands strictly inside
// the sampled leaf cell, otherwise the literal second descent is taken.
// kSlot: also the accumulator slot of the leaf that holds the canonical form of the sampled direction.
template <bool kSlot>
__device__ __forceinline__ void quad_sample_t(const QuadRec *rec, JumpRef jump, uint32_t tree, TreeHead head,
                                              Pcg32 &rng, float &dx, float &dy, float &dz, float &pdf_out,
                                              uint32_t &levels, uint32_t &slot)
{
	float px = 0.0f, py = 0.0f;
	float lox = 0.0f, loy = 0.0f, size = 1.0f;
	float pdf = 1.0f, node_irr = head.root_irr;
	bool dead = false, reached_leaf = false;
	uint32_t r = head.root_rec;
	uint32_t leaf_slot = kSlotRoot; // of the node the walk stands on, should it be a leaf
	levels = 0;
	for (int it = 0; it < kMaxLevels + 1; ++it) {
		// draw order per visited node: next_2d (x, y) then next_1d; the leaf uses the first two,
		// an inner node only the third -- unused draws still advance the stream
		if (r == kNoRecord) { // leaf: uniform point in the cell (quadtree.py:956)
			const float u = rng.next_f32();
			const float v = rng.next_f32();
			rng.skip();
			const float ux = u * size, uy = v * size;
			px = lox + ux;
			py = loy + uy;
			reached_leaf = true;
			break;
		}
		rng.skip();
		rng.skip();
		const float xi = rng.next_f32();
		const QuadLoad q = load_rec(rec, r);
		const float c1 = q.i0;
		const float c2 = q.i1 + c1;
		const float c3 = q.i2 + c2;
		const float c4 = q.i3 + c3;
		const float s = xi * c4;
		int k = -1;
		if (s < c1) k = 0;
		if (c1 <= s && s < c2) k = 1;
		if (c2 <= s && s < c3) k = 2;
		if (c3 <= s) k = 3;
		if (k < 0) break; // NaN energies: the reference would never terminate
		++levels;
		const float child_irr = sel4f(k, q.i0, q.i1, q.i2, q.i3);
		if (!dead) {
			pdf = pdf * ((4.0f * child_irr) / node_irr);
			if (pdf != pdf) { pdf = 0.0f; dead = true; }
		}
		node_irr = child_irr;
		const float half = size * 0.5f;
		if (k == 0 || k == 3) lox = lox + half;
		if (k == 0 || k == 1) loy = loy + half;
		size = half;
		const uint32_t c = sel4u(k, q.c0, q.c1, q.c2, q.c3);
		leaf_slot = r * 4u + (uint32_t)k;
		r = c == 0 ? kNoRecord : c;
	}
	canonical_to_dir(px, py, dx, dy, dz);
	float qx, qy;
	dir_to_canonical(dx, dy, dz, qx, qy);
	const bool strictly_inside = reached_leaf && qx > lox && qx < lox + size && qy > loy && qy < loy + size;
	levels = stat_word(levels, 32u * levels); // one record per level sampled
	if (strictly_inside) {
		pdf_out = dead ? 0.0f : pdf * kInvFourPiF;
		slot = leaf_slot;
	} else {
		uint32_t lv;
		pdf_out = quad_pdf_t<kSlot>(rec, jump, tree, head, qx, qy, lv, slot);
		levels += stat_word(0u, stat_bytes(lv)); // (the literal second descent: its bytes, not its levels)
	}
}

__device__ __forceinline__ void quad_sample(const QuadRec *rec, JumpRef jump, uint32_t tree, TreeHead head,
                                            Pcg32 &rng, float &dx, float &dy, float &dz, float &pdf_out,
                                            uint32_t &levels)
{
	uint32_t slot;
	quad_sample_t<false>(rec, jump, tree, head, rng, dx, dy, dz, pdf_out, levels, slot);
}

// addIrradiancePropagate (quadtree.py:398-441): the accumulator slot of the leaf that (cx,cy) falls
// into -- rec*4+child for a leaf below a record, the tree's root accumulator when the root itself
// is the leaf -- or nothing when the root cell does not contain the point (quadtree.py:404-405).
// The walk is done for the two directions of one record together (path direction and emitter
// direction fall into the same quadtree, quadtree.py:443-464), advanced in lock step so that the two
// dependent gather chains overlap instead of running one after the other.
struct LeafCursor {
	uint32_t r, slot, levels; // (levels: a statistics word, see stat_word)
	float lox, loy, h, cx, cy;
	bool walking, found, is_root;
};

// pre: the jump-table entry for (cx,cy) fetched ahead (jump_prefetch, with `wanted` at least where this cursor walks)
__device__ __forceinline__ LeafCursor leaf_cursor_pre(TreeHead head, float cx, float cy, bool enable, const JumpPre &pre)
{
	LeafCursor c;
	c.r = head.root_rec;
	c.slot = 0;
	c.levels = 0;
	c.lox = 0.0f; c.loy = 0.0f; c.h = 0.5f;
	c.cx = cx; c.cy = cy;
	const bool inside = enable && cx >= 0.0f && cx <= 1.0f && cy >= 0.0f && cy <= 1.0f; // quadtree.py:404-405
	c.is_root = head.root_rec == kNoRecord;
	c.found = inside && c.is_root;
	c.walking = inside && !c.is_root;
	if (pre.hit) c.levels = stat_word(0u, 16u); // (gathered, used or not)
	if (c.walking && pre.hit) { // skip the levels the table covers
		const uint4 e = pre.e;
		c.levels += (e.w >> 26) & 15u;
		if (e.x == kNoRecord) {
			c.slot = e.w & kJumpSlotMask;
			c.found = true;
			c.walking = false;
		} else {
			c.r = e.x;
			c.lox = pre.jx; c.loy = pre.jy;
			c.h = 0.5f / (float)(1 << pre.bits);
		}
	}
	return c;
}

// the same with the table's entry fetched here, behind the head (and only where the cursor walks)
__device__ __forceinline__ LeafCursor leaf_cursor(JumpRef jump, uint32_t tree, TreeHead head, float cx, float cy,
                                                  bool enable)
{
	const bool walks = enable && cx >= 0.0f && cx <= 1.0f && cy >= 0.0f && cy <= 1.0f; // (not a question to the head: the two gathers overlap)
	return leaf_cursor_pre(head, cx, cy, enable, jump_prefetch(jump, tree, cx, cy, walks));
}

__device__ __forceinline__ void leaf_step(LeafCursor &c, uint4 ch)
{
	const float mx = c.lox + c.h, my = c.loy + c.h;
	int first, last;
	quadrant(c.cx, c.cy, mx, my, first, last);
	c.levels += stat_word(1u, 16u); // (the child words of one record)
	const uint32_t child = sel4u(last, ch.x, ch.y, ch.z, ch.w);
	if (child == 0) {
		c.slot = c.r * 4u + (uint32_t)last;
		c.found = true;
		c.walking = false;
		return;
	}
	if (last == 0 || last == 3) c.lox = mx;
	if (last == 0 || last == 1) c.loy = my;
	c.h *= 0.5f;
	c.r = child;
}

__device__ __forceinline__ void quad_find_leaf_slots2(const QuadRec *rec, LeafCursor &a, LeafCursor &b)
{
	for (int it = 0; it < kMaxLevels && (a.walking || b.walking); ++it) {
		// a cursor that has stopped re-reads record 0 (it exists: the other one is inside a record)
		const uint4 cha = gather16(reinterpret_cast<const uint4 *>(rec + (a.walking ? a.r : 0u)) + 1);
		const uint4 chb = gather16(reinterpret_cast<const uint4 *>(rec + (b.walking ? b.r : 0u)) + 1);
		if (a.walking) leaf_step(a, cha);
		if (b.walking) leaf_step(b, chb);
	}
	// kMaxLevels records deep without reaching a leaf: not a tree this library builds
	a.walking = false;
	b.walking = false;
}

// the slot word of a finished cursor (see kSlotNone / kSlotRoot)
__device__ __forceinline__ uint32_t cursor_slot(const LeafCursor &c)
{
	return !c.found ? kSlotNone : (c.is_root ? kSlotRoot : c.slot);
}

} // namespace pg
