// pg_tree.hpp -- device-resident layout of the SD-tree.
//
// The reference keeps one Dr.Jit column per field (45 B per KD node, 45 B per quadtree node,
// explicit bounding boxes everywhere: kdtree.py:16-35, quadtree.py:12-37).  The descent
// kernels need far less, so the hot data is packed for one aligned vector load per level:
//
//   KdNode   16 B  one per KD node, indexed by the REFERENCE'S node numbering
//                  (children of a split are adjacent: kdtree.py:243-245)
//   QuadRec  32 B  one per NON-LEAF quadtree node; it carries the four children's energies
//                  and, per child, the index of the child's own record (0 = child is a leaf).
//                  Leaves own no record: bounding boxes are implicit (midpoint splits,
//                  quadtree.py:151) and a leaf is known from its parent.
//   TreeHead  8 B  one per quadtree (= per KD leaf): root record + root energy.
//
// Records are stored level-major over the whole forest, each level in the order induced by
// the previous one (SURVEY Appendix A8 restricted to non-leaf nodes), so export to the
// reference's canonical arena is a pure expansion and a bottom-up pass is one contiguous
// range per level.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

// A tree gather is ONE vector load.  uint4 / uint2 are structs to the compiler: a load through them is four
// (two) scalar loads that it may or may not put together again, and where the words are used in different
// blocks it does not -- the 16-byte KD grid entry came out as four dword gathers, a KD node as a dwordx3 in
// the descent loop plus a dword behind it.  The kernels that walk the trees are bound by the number of
// divergent gathers their lanes make, so every table access goes through these.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 gather16(const void *p)
{
	const u32x4_t v = *reinterpret_cast<const u32x4_t *>(p);
	return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 gather8(const void *p)
{
	const u32x2_t v = *reinterpret_cast<const u32x2_t *>(p);
	return make_uint2(v.x, v.y);
}

constexpr uint32_t kNoRecord = 0xffffffffu; // TreeHead.root_rec: the root itself is a leaf
constexpr int kMaxLevels = 32;              // hard bound on any descent loop

struct alignas(16) KdNode {
	uint32_t child; // index of the left child (right = child+1); 0 = leaf
	float split;    // fp32 midpoint of the node's bbox on `axis` (kdtree.py:270)
	uint32_t axis_depth; // axis (= depth % 3) in bits 0-1, depth in bits 2..
	uint32_t tree;  // quadTreeRootIndex (kdtree.py:23); for inner nodes the stale parent value
};

struct alignas(16) QuadRec {
	float irr[4];      // irradiance of child 1..4 (quadrants of quadtree.py:153-175)
	uint32_t child[4]; // record index of child 1..4, 0 = leaf
};

struct alignas(8) TreeHead {
	uint32_t root_rec; // kNoRecord when the root is a leaf
	float root_irr;
};

// Jump table over the top kJumpBits levels of every quadtree: entry (tree, iy, ix) describes where
// a descent for a point strictly inside cell (ix, iy) of the 2^kJumpBits grid stands after those
// levels -- one 16-byte gather instead of two per level.  The path to such a cell is unique (no
// tie rule applies off the cell boundaries), and the pdf product is formed at build time by the
// same operations in the same order as the level-by-level loop, so results are bit-identical;
// points on a cell boundary or outside the unit square take the loop from the root.
// (measured on the veach-ajar bench, ms per step: 3 bits 57.5, 4 bits 56.9, 5 bits 56.4, 6 bits 56.3 -- the quadtrees' leaves
// lie 4.6 levels deep on average, so a 64 x 64 table ends most walks in its one gather; it costs 64 KB per quadtree)
constexpr int kJumpBits = 6; // the finest table.  A forest takes fewer bits when its tables would not fit their
                                        // memory budget (pg_refine.hip: rebuild_jump): 4^bits entries per tree, the same for
                                        // every tree of a forest, a number the kernels read from the view (uniform)
constexpr uint32_t kJumpCells = 1u << (2 * kJumpBits); // entries per tree of the finest table
struct alignas(16) QuadJump {
	uint32_t next;  // record to continue from; kNoRecord: the walk ended in a leaf within the table
	float pdf;      // product of 4*child/node over the levels taken (final value incl. 1/(4 pi) when ended)
	float irr;      // energy of the node reached (the next level's denominator)
	uint32_t info;  // bits 0-25 accumulator slot of the leaf (when ended), 26-29 levels taken, 30 pdf undefined
	                // (a 0/0 on the way: take the loop), 31 the tree's ROOT is the leaf (every entry of such a tree: the
	                // slot is the root's accumulator; a walk that hits the table never needs the tree's head)
};
constexpr uint32_t kJumpSlotMask = (1u << 26) - 1u;
// what a kernel holds of a forest's tables: 4^bits entries per tree, tree-major; p == nullptr: none (every level is walked)
struct JumpRef {
	const QuadJump *p;
	int bits;
};

// Jump grid over the top of the KD tree: the root box cut into 2^kKdGridBits cells per axis; entry
// (iz, iy, ix) holds the node every point strictly inside that cell reaches from the root before a
// split plane cuts through the cell, the number of levels taken, and that node's own record -- one
// 16-byte gather instead of a chain of up to 3 * kKdGridBits + 1 dependent ones (the splits are
// midpoints, axis = depth % 3, so a cell of the grid usually lies inside one node 3 * kKdGridBits levels
// down, or inside a leaf above that: then the gather is the whole descent).  The kernels that walk
// the tree are bound by the number of divergent gathers their lanes make (profiles/r02/pmc_guide.txt),
// not by bytes.  The table is built by descending with the cell's interval ([lo, hi) decides like every
// point in it as long as lo >= split or hi <= split); a query verifies with the build's own cell bounds
// that its point lies strictly inside the cell it computed, and otherwise (a point on a cell face,
// outside the box, NaN) descends from the root: results are those of KDTree.getLeafNodeIndex
// (kdtree.py:435-470) either way.
// The resolution follows the tree: 2^bits cells per axis with bits = ceil(log2(leaves) / 3) + 1, at most
// kKdGridBits (a table much finer than the tree only spreads the queries over more cache lines: uniform
// random queries into a depth-12 tree ran 15 % slower on a 64^3 table than on a 32^3 one).
constexpr int kKdGridBits = 6; // the finest grid
constexpr uint32_t kKdGridCells = 1u << (3 * kKdGridBits);
// Behind the 8^grid_bits cells the table holds two entries for the ROOT: [cells] = node 0 as a searching
// lane that lies in no cell (on a cell face) starts from it, [cells + 1] = node 0 as a lane that does not
// search (outside the root box, NaN) returns it -- so that every query is one gather at one load site.
constexpr uint32_t kKdGridRootEntries = 2;
// The cell boundaries are not bmin + i * width: they are made by the KD tree's own arithmetic, bisecting
// [bmin, bmax] recursively with mid = (lo + hi) / 2 in fp32 (kdtree.py:270), so that they coincide with
// the split planes of the tree bit for bit and a cell is never cut by a plane that is "its own face but
// one ulp off".
constexpr int kKdGridPlanes = (1 << kKdGridBits) + 1;
struct alignas(16) KdGridEntry {
	uint32_t node;  // reference node index reached
	uint32_t meta;  // bits 0-15 the node's axis_depth word, bits 16.. levels descended to get there
	uint32_t child; // the node's child word (0 = leaf)
	uint32_t value; // a leaf's quadtree (KdNode::tree), an inner node's split plane (fp32 bits)
};

// Read-only view handed to the query kernels (sdTree_prev).
struct TreeView {
	const KdNode *kd;
	const QuadRec *rec;
	const TreeHead *head;
	JumpRef jump;         // n_trees << (2 * jump.bits) entries, or {nullptr, 0}
	const KdGridEntry *kd_grid; // 8^grid_bits cell entries + kKdGridRootEntries, or nullptr
	const float *kd_planes;     // 3 * kKdGridPlanes cell boundaries of the grid (x planes, y planes, z planes), ascending
	float bmin[3], bmax[3]; // root bounding box (kdtree.py:138)
	float grid_inv[3];      // 2^grid_bits / (bmax - bmin): a first guess of the cell index
	int grid_bits;          // cells per axis = 2^grid_bits (<= kKdGridBits); plane k of an axis at [axis * kKdGridPlanes + k]
	uint32_t n_kd, n_rec, n_trees;
};

// Accumulation view (sdTree_current: same topology, integer accumulators).
// One accumulator = kAccWords int64 words in one 32-byte sector: three limbs of the irradiance sum
//  and the number of in-bbox records whose path direction ended in this leaf.
// Keeping all four words in one sector matters: gfx950 services the atomics of one wave-instruction
// per touched sector, not per lane (tools/atomic_probe.hip: two lanes adding to the two halves of a
// 16-byte slot cost one atomic), so the splat kernels hand a record's words to adjacent lanes.
// acc layout: one contiguous int64 buffer [ rec_acc | root_acc | leaf_count ]
//   rec_acc   : n_rec*4 slots * 4 words   (slot = rec*4 + child)
//   root_acc  : n_trees * 4 words         (used directly when the root is a leaf)
//   leaf_count: n_trees                   (fallback counter: in-bbox records whose direction lies
//                                          outside the unit square, which reach no quadtree leaf)
// The KD vertCount of a leaf (kdtree.py:199) is the sum of word 3 over its tree plus the fallback.
constexpr int kAccWords = 4;
struct AccumView {
	long long *rec_acc;
	long long *root_acc;
	unsigned long long *leaf_count;
};

} // namespace pg
