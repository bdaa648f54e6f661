// pg_context.hpp -- host-side state of one pg_context (library-owned device memory).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pgsd.h"
#include "pg_kernels.hpp"
#include "pg_math.hpp"
#include "pg_tree.hpp"

namespace pg {

// Growable device array.  ensure() discards contents when it has to reallocate.  Growth is geometric where the caller
// asks for slack: a hipFree synchronises the whole device and a fresh hipMalloc of a gigabyte maps pages for tens of
// milliseconds, so a buffer that follows a growing tree must not be reallocated every time the tree grows a little
// (round 3's refine spent 20-190 ms per iteration there).
// Fault injection for the tests (pg_debug_fail_alloc, include/pgsd.h): when >= 0, the number of device allocations that still
// succeed -- the one after them reports hipErrorOutOfMemory exactly as a refused hipMalloc does (the buffer is left empty), and
// the hook disarms itself.  One variable for the whole library (C++17 inline variable); -1 = off.
inline long long g_alloc_fail_countdown = -1;

template <class T> struct DevBuf {
	T *p = nullptr;
	size_t cap = 0;
	~DevBuf() { release(); }
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	void release()
	{
		if (p) (void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
	hipError_t ensure(size_t n, double slack = 1.0)
	{
		if (n <= cap) return hipSuccess;
		release();
		size_t want = (size_t)((double)n * slack);
		if (want < n) want = n;
		if (want < 16) want = 16;
		if (g_alloc_fail_countdown >= 0 && g_alloc_fail_countdown-- == 0) return hipErrorOutOfMemory; // (tests only; p is empty)
		hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
		if (e != hipSuccess) { p = nullptr; return e; }
		cap = want;
		return hipSuccess;
	}
	void swap(DevBuf &o)
	{
		T *tp = p; p = o.p; o.p = tp;
		size_t tc = cap; cap = o.cap; o.cap = tc;
	}
};

// One SD-tree topology + sampling values (sdTree_prev) + integer accumulators (sdTree_current).
// The two reference objects always share their topology (prev <- copy of current after every
// refine, path_guiding_integrator.py:582), so it is stored once.
struct Forest {
	// KD tree, reference node numbering
	DevBuf<KdNode> kd;
	DevBuf<float> kd_bmin, kd_bmax;  // [n_kd][3], kept for refine/export only
	DevBuf<float> kd_vcount;         // fp32 vertCount column of sdTree_prev (export only)
	uint32_t n_kd = 0;
	// quadtree forest
	DevBuf<QuadRec> rec;
	uint32_t n_rec = 0;
	std::vector<uint32_t> level_off; // level l records are [level_off[l], level_off[l+1])
	DevBuf<TreeHead> head;
	DevBuf<float> tree_thr;          // refinementThreshold of each tree (export only)
	uint32_t n_trees = 0;
	DevBuf<QuadJump> jump;           // n_trees << (2 * jump_bits) entries, rebuilt whenever rec/head change
	bool jump_valid = false;
	int jump_bits = 0;               // levels the tables of this forest cover (<= kJumpBits; what fits the memory budget)
	DevBuf<KdGridEntry> kd_grid;     // 8^kd_grid_bits cells + kKdGridRootEntries, rebuilt whenever the KD tree changes
	DevBuf<float> kd_planes;         // 3 * kKdGridPlanes cell boundaries (they follow the root box)
	bool kd_grid_valid = false;
	int kd_grid_bits = 0;            // cells per axis = 2^kd_grid_bits
	// accumulators of the running iteration: [rec_acc | root_acc | leaf_count]
	DevBuf<long long> acc;
	DevBuf<long long> xchg;          // the accumulators in their 24-byte exchange format (pg_exchange_pack): what travels
	uint64_t n_acc() const { return (uint64_t)n_rec * 4 + n_trees; }             // four-word accumulators
	uint64_t xchg_count() const { return n_acc() * kXchgWords + n_trees; }       // [n_acc x 3 | fallback counters]
	uint64_t acc_count() const
	{
		return (uint64_t)n_rec * 4 * kAccWords + (uint64_t)n_trees * kAccWords + n_trees;
	}
	// what a refine needs besides the forest itself: kept between refines (pg_refine.hip) so that a refine of a tree that
	// did not outgrow them allocates nothing
	struct RefineScratch {
		DevBuf<I128> tot;
		DevBuf<float> slot_irr, root_irr, new_thr;
		DevBuf<unsigned long long> kd_cnt, cnt_tot, tree_count, plan;
		DevBuf<uint32_t> tree_src, counts, cnt, pos, scan_sums, scan_total;
		DevBuf<TreeHead> new_head;
		DevBuf<QuadRec> new_rec;
		// the refined KD tree and the next iteration's accumulators are built HERE and become the forest's by a pointer swap
		// (pg_refine.hip: a refine is a transaction); what they replace becomes the next refine's scratch
		DevBuf<KdNode> new_kd;
		DevBuf<float> new_bmin, new_bmax, new_vc;
		DevBuf<long long> new_acc;
		DevBuf<unsigned char> pend_a, pend_b; // (Pending entries: the type lives in pg_refine.hip)
		void *pinned = nullptr;             // 64 bytes of page-locked host memory for the counters a refine reads back
		~RefineScratch() { if (pinned) (void)hipHostFree(pinned); }
	} scratch;
	AccumView accum_view()
	{
		AccumView a;
		a.rec_acc = acc.p;
		a.root_acc = acc.p + (uint64_t)n_rec * 4 * kAccWords;
		a.leaf_count = reinterpret_cast<unsigned long long *>(a.root_acc + (uint64_t)n_trees * kAccWords);
		return a;
	}
};

} // namespace pg

struct pg_render_state; // pg_render.hip

struct pg_context {
	pg_render_state *render = nullptr;
	int device = 0;
	int n_cus = 256; // compute units of the device (sizes the persistent grids)
	std::string err;
	bool configured = false;
	float bmin[3] = {0, 0, 0}, bmax[3] = {1, 1, 1};
	uint64_t num_rays = 0;
	int32_t max_depth = 0, kd_max_depth = 10, quad_max_depth = 30, store_nee = 1;
	float bsdf_fraction = 0.5f;
	int32_t iteration = 0, is_final = 0;
	double kd_max_leaf_size = 1.0;
	pg::Forest f;
	pg::DepthCounters *dc = nullptr; // device
	bool dc_on = false;
	bool ph_on = false;              // a probe build's phase stamps are on (pg_enable_depth_counters 1 or 2; never in the product build)
	pg::DevBuf<unsigned long long> ph_buf; // their striped counters (kPhaseStripes x kPhaseWords), allocated by the first enable of a probe build
	// memory budget of the quadtree jump tables (bytes): $PGSD_JUMP_TABLE_MAX_BYTES at pg_create, default 2 GiB.  The resolution a
	// forest gets follows from the forest and this budget alone (pg_refine.hip: rebuild_jump)
	uint64_t jump_budget = 2ull << 30;
	void *comm = nullptr;            // ncclComm_t of the multi-GPU exchange (pg_comm.hip)
	bool comm_owned = false;         // made by pg_comm_init (destroyed with the context) or attached by the caller
	int comm_ranks = 0;

	pg::TreeView view() const
	{
		pg::TreeView t;
		t.kd = f.kd.p;
		t.rec = f.rec.p;
		t.head = f.head.p;
		t.jump.p = f.jump_valid ? f.jump.p : nullptr;
		t.jump.bits = f.jump_valid ? f.jump_bits : 0;
		t.kd_grid = f.kd_grid_valid ? f.kd_grid.p : nullptr;
		t.kd_planes = f.kd_planes.p;
		for (int a = 0; a < 3; ++a) {
			t.bmin[a] = bmin[a]; t.bmax[a] = bmax[a];
			t.grid_inv[a] = (float)(1 << f.kd_grid_bits) / (bmax[a] - bmin[a]);
		}
		t.grid_bits = f.kd_grid_bits;
		t.n_kd = f.n_kd;
		t.n_rec = f.n_rec;
		t.n_trees = f.n_trees;
		return t;
	}
};

namespace pg {
// pg_refine.hip
int refine_and_swap(pg_context *ctx, hipStream_t s);
int rebuild_jump(pg_context *ctx, hipStream_t s); // after every change of the quadtree records or heads
// pg_render.hip
void destroy_render_state(pg_context *ctx);
// pg_render_wave.hip
bool shade_phases_compiled_in();
// pg_comm.hip
void destroy_comm(pg_context *ctx);
// pg_sort.hip
size_t sort_pairs_temp_bytes(uint32_t n);
hipError_t sort_places16(void *temp, size_t temp_bytes, const uint16_t *keys_in, uint16_t *keys_out, uint32_t *places_out, uint32_t n,
                         const uint32_t *d_live, hipStream_t s);
// helpers shared by pg_context.hip / pg_refine.hip
int fail(pg_context *ctx, int code, const std::string &msg);
int hip_fail(pg_context *ctx, hipError_t e, const char *what);
} // namespace pg

#define PG_HIP(ctx, call)                                                      \
	do {                                                                       \
		hipError_t e__ = (call);                                               \
		if (e__ != hipSuccess) return pg::hip_fail((ctx), e__, #call);        \
	} while (0)
