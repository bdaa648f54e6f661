// pg_refine.hip -- refineAndPrepareSDTreeForNextIteration on the device
// (path_guiding_integrator.py:566-586).
//
// The reference drives this step from Python with one device sync per BFS level and a full
// reallocation of every column per split round (kdtree.py:229-358, quadtree.py:512-637,
// 695-851).  Here it is a fixed sequence of level-synchronous kernels:
//
//   1. resolve   bottom-up, one kernel per record level: exact 128-bit totals of every node
//                from the leaf accumulators, each rounded once to fp32
//   2. kd        single-workgroup kernel replaying the reference's split rounds (node and tree
//                numbering of kdtree.py:243-245, 316-323 is reproduced exactly)
//   3. rebuild   top-down, two kernels + a scan per level: merge (irr < thr), keep, or split
//                (irr > thr, depth < max) every node of every (possibly cloned) tree and emit the
//                new records directly in canonical order
//   4. swap      the new forest becomes both sdTree_prev (values) and sdTree_current (zeroed
//                accumulators)
//
// A refine is a TRANSACTION (round 6; the reference grows by allocate-new + copy, common.py:161-189, so its trees survive a
// failed split): steps 1-3 read the forest and write only scratch buffers -- the split KD tree is made in a COPY of the node
// arrays, the new records, heads and thresholds in their own buffers, the next iteration's accumulators in a spare one -- and
// step 4 commits by swapping pointers and counts, which cannot fail.  Any error before the commit (an allocation, a limit)
// returns with sdTree_prev and sdTree_current exactly as they were: every query, pass and export answers as before, and
// the refine can be called again.  After the commit only the two accelerators are rebuilt (KD jump grid, quadtree jump
// tables); if that fails the forest simply has none and every consumer walks from the roots (results are the same).
//
// Cloning a quadtree for the right child of a KD split (kdtree.py:316-323) commutes with the
// per-tree refinement that follows (both copies start identical and see the same threshold), so
// a clone is just a second level-0 entry pointing at the same source tree.
#include "pg_context.hpp"

#include <math.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>

#include "pg_math.hpp"

namespace pg {

constexpr int kBlk = 256;
constexpr int kScanItems = 4;                 // elements per thread in the scan kernels
constexpr int kScanTile = kBlk * kScanItems;  // 1024

// ---------------------------------------------------------------------------------------------
// exclusive prefix sum over uint32 (three small kernels; n <= 2^32)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &block_total)
{
	const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	uint32_t incl = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(incl, o, 64);
		if ((int)lane >= o) incl += t;
	}
	if (lane == 63) lds[wid] = incl;
	__syncthreads();
	uint32_t wave_off = 0, tot = 0;
	const unsigned nw = blockDim.x >> 6;
	for (unsigned w = 0; w < nw; ++w) {
		const uint32_t s = lds[w];
		if (w < wid) wave_off += s;
		tot += s;
	}
	__syncthreads();
	block_total = tot;
	return wave_off + incl - v;
}

__global__ __launch_bounds__(kBlk) void k_scan_reduce(const uint32_t *__restrict__ in, uint32_t n,
                                                      uint32_t *__restrict__ sums)
{
	__shared__ uint32_t lds[kBlk / 64];
	const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
	uint32_t s = 0;
#pragma unroll
	for (int k = 0; k < kScanItems; ++k)
		if (base + k < n) s += in[base + k];
	uint32_t tot;
	block_exclusive_scan(s, lds, tot);
	if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of the block sums in place, grand total to *total
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t *__restrict__ sums, uint32_t nb,
                                                    uint32_t *__restrict__ total)
{
	__shared__ uint32_t lds[1024 / 64];
	uint32_t carry = 0;
	for (uint32_t base = 0; base < nb; base += 1024) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < nb ? sums[i] : 0;
		uint32_t tot;
		const uint32_t ex = block_exclusive_scan(v, lds, tot);
		if (i < nb) sums[i] = carry + ex;
		carry += tot;
	}
	if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(kBlk) void k_scan_apply(const uint32_t *__restrict__ in, uint32_t n,
                                                     const uint32_t *__restrict__ sums,
                                                     uint32_t *__restrict__ out)
{
	__shared__ uint32_t lds[kBlk / 64];
	const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
	uint32_t v[kScanItems], s = 0;
#pragma unroll
	for (int k = 0; k < kScanItems; ++k) {
		v[k] = base + k < n ? in[base + k] : 0;
		s += v[k];
	}
	uint32_t tot;
	uint32_t ex = block_exclusive_scan(s, lds, tot) + sums[blockIdx.x];
#pragma unroll
	for (int k = 0; k < kScanItems; ++k) {
		if (base + k < n) out[base + k] = ex;
		ex += v[k];
	}
}

// ---------------------------------------------------------------------------------------------
// 1. resolve accumulators
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlk) void k_resolve_level(const QuadRec *__restrict__ rec,
                                                        const long long *__restrict__ rec_acc,
                                                        I128 *__restrict__ tot,
                                                        unsigned long long *__restrict__ cnt_tot,
                                                        float *__restrict__ slot_irr, uint32_t begin, uint32_t end)
{
	const uint32_t r = begin + blockIdx.x * kBlk + threadIdx.x;
	if (r >= end) return;
	const uint4 ch = reinterpret_cast<const uint4 *>(rec + r)[1];
	const uint32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
	I128 s = {0, 0};
	unsigned long long n = 0;
	float irr[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		I128 v;
		if (c[j]) { // child records live on the next level: already resolved
			v = tot[c[j]];
			n += cnt_tot[c[j]];
		} else {
			const long long *l = rec_acc + ((size_t)r * 4 + j) * kAccWords;
			v = limbs_resolve(l[0], l[1], l[2]);
			n += (unsigned long long)l[3];
		}
		irr[j] = i128_to_f32(v);
		s = i128_add(s, v);
	}
	tot[r] = s;
	cnt_tot[r] = n;
	reinterpret_cast<float4 *>(slot_irr)[r] = make_float4(irr[0], irr[1], irr[2], irr[3]);
}

// per tree: root irradiance and the number of records its KD leaf received (= vertCount, kdtree.py:199)
__global__ __launch_bounds__(kBlk) void k_resolve_roots(const TreeHead *__restrict__ head,
                                                        const long long *__restrict__ root_acc,
                                                        const I128 *__restrict__ tot,
                                                        const unsigned long long *__restrict__ cnt_tot,
                                                        const unsigned long long *__restrict__ fallback_count,
                                                        float *__restrict__ root_irr,
                                                        unsigned long long *__restrict__ tree_count, uint32_t n_trees)
{
	const uint32_t t = blockIdx.x * kBlk + threadIdx.x;
	if (t >= n_trees) return;
	const uint32_t rr = head[t].root_rec;
	const long long *ra = root_acc + (size_t)t * kAccWords;
	const I128 v = rr == kNoRecord ? limbs_resolve(ra[0], ra[1], ra[2]) : tot[rr];
	root_irr[t] = i128_to_f32(v);
	tree_count[t] = (rr == kNoRecord ? (unsigned long long)ra[3] : cnt_tot[rr]) + fallback_count[t];
}

// exact per-node record counts (what kdtree.py:199 accumulates on every visited node)
__global__ __launch_bounds__(kBlk) void k_kd_counts(const KdNode *__restrict__ kd, uint32_t n_kd,
                                                    const unsigned long long *__restrict__ leaf_count,
                                                    unsigned long long *__restrict__ cnt, int depth, int leaves)
{
	const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
	if (i >= n_kd) return;
	const KdNode n = kd[i];
	if (leaves) {
		if (n.child == 0) cnt[i] = leaf_count[n.tree];
	} else if (n.child != 0 && (int)(n.axis_depth >> 2) == depth) {
		cnt[i] = cnt[n.child] + cnt[n.child + 1];
	}
}

__global__ __launch_bounds__(kBlk) void k_kd_vcount(const unsigned long long *__restrict__ cnt,
                                                    float *__restrict__ vc, uint32_t n_kd)
{
	const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
	if (i < n_kd) vc[i] = count_to_f32(cnt[i]);
}

// ---------------------------------------------------------------------------------------------
// 2. KD refine (kdtree.py:333-358 -> 229-323)
// ---------------------------------------------------------------------------------------------
// how many nodes / trees the split rounds will add (every over-full leaf grows a complete subtree
// because children inherit vertCount/2: kdtree.py:261-264)
__global__ __launch_bounds__(kBlk) void k_kd_plan(const KdNode *__restrict__ kd, uint32_t n_kd,
                                                  const float *__restrict__ vc, float thr, int max_depth,
                                                  unsigned long long *__restrict__ out /*[2]*/)
{
	const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
	if (i >= n_kd) return;
	const KdNode n = kd[i];
	if (n.child != 0) return;
	float v = vc[i];
	int d = (int)(n.axis_depth >> 2), j = 0;
	while (v > thr && d < max_depth) {
		if (v > 0.0f) v = v / 2.0f;
		++d;
		++j;
	}
	if (j) {
		atomicAdd(&out[0], (2ull << j) - 2ull); // new nodes
		atomicAdd(&out[1], (1ull << j) - 1ull); // new trees
	}
}

struct KdRefineArgs {
	KdNode *kd;
	float *bmin, *bmax, *vc;
	uint32_t *tree_src; // for every tree id (old and new): the pre-refine tree it is a copy of
	uint32_t n_kd, n_trees;
	float thr;
	int max_depth;
	uint32_t *out_counts; // [0] = final n_kd, [1] = final n_trees
};

__global__ __launch_bounds__(1024) void k_kd_refine(KdRefineArgs a)
{
	__shared__ uint32_t lds[1024 / 64];
	__shared__ uint32_t s_n, s_trees, s_split;
	if (threadIdx.x == 0) { s_n = a.n_kd; s_trees = a.n_trees; }
	__syncthreads();
	for (int round = 0; round <= a.max_depth; ++round) {
		const uint32_t n = s_n, trees = s_trees;
		__syncthreads();
		uint32_t carry = 0;
		for (uint32_t base = 0; base < n; base += 1024) {
			const uint32_t i = base + threadIdx.x;
			bool split = false;
			KdNode nd = {0, 0.0f, 0, 0};
			if (i < n) {
				nd = a.kd[i];
				split = nd.child == 0 && a.vc[i] > a.thr && (int)(nd.axis_depth >> 2) < a.max_depth; // kdtree.py:348
			}
			uint32_t tot;
			const uint32_t pos = carry + block_exclusive_scan(split ? 1u : 0u, lds, tot);
			carry += tot;
			if (split) { // kdtree.py:243-323, i-th split node of the round (ascending index)
				const uint32_t l = n + 2 * pos, r = l + 1;
				const uint32_t depth = nd.axis_depth >> 2, axis = depth % 3u;
				float v = a.vc[i];
				if (v > 0.0f) v = v / 2.0f;
				a.vc[l] = v;
				a.vc[r] = v;
				float mn[3], mx[3];
				for (int k = 0; k < 3; ++k) { mn[k] = a.bmin[3 * (size_t)i + k]; mx[k] = a.bmax[3 * (size_t)i + k]; }
				const float mid = (mn[axis] + mx[axis]) / 2.0f;
				for (int k = 0; k < 3; ++k) {
					a.bmin[3 * (size_t)l + k] = mn[k];
					a.bmax[3 * (size_t)l + k] = (uint32_t)k == axis ? mid : mx[k];
					a.bmin[3 * (size_t)r + k] = (uint32_t)k == axis ? mid : mn[k];
					a.bmax[3 * (size_t)r + k] = mx[k];
				}
				const uint32_t cd = depth + 1;
				const uint32_t new_tree = trees + pos;
				KdNode ln = {0u, 0.0f, (cd % 3u) | (cd << 2), nd.tree};
				KdNode rn = {0u, 0.0f, (cd % 3u) | (cd << 2), new_tree};
				a.kd[l] = ln;
				a.kd[r] = rn;
				a.tree_src[new_tree] = a.tree_src[nd.tree];
				nd.child = l;
				nd.split = mid;
				a.kd[i] = nd;
			}
		}
		if (threadIdx.x == 0) { s_split = carry; s_n = n + 2 * carry; s_trees = trees + carry; }
		__threadfence_block();
		__syncthreads();
		if (s_split == 0) break;
	}
	if (threadIdx.x == 0) { a.out_counts[0] = s_n; a.out_counts[1] = s_trees; }
}

// ---------------------------------------------------------------------------------------------
// 3. quadtree rebuild
// ---------------------------------------------------------------------------------------------
struct alignas(16) Pending { // a non-leaf node of the NEW forest waiting to be emitted
	uint32_t old_rec; // record of the old forest it continues, or kNoRecord for a freshly split node
	float value;      // its own irradiance (only used by fresh nodes: children get value/4)
	uint32_t tree;    // new tree id (threshold lookup)
	uint32_t pad;
};

struct ChildPlan {
	float irr[4];
	uint32_t old_child[4]; // old record to continue (0 = none)
	bool nonleaf[4];
};

// quadtree.py:563-637 applied to the four children of one kept node
__device__ __forceinline__ ChildPlan plan_children(const Pending &p, const QuadRec *__restrict__ rec,
                                                   const float *__restrict__ slot_irr, float thr,
                                                   int child_depth, int max_depth)
{
	ChildPlan c;
	if (p.old_rec != kNoRecord) {
		const float4 ir = reinterpret_cast<const float4 *>(slot_irr)[p.old_rec];
		const uint4 ch = reinterpret_cast<const uint4 *>(rec + p.old_rec)[1];
		c.irr[0] = ir.x; c.irr[1] = ir.y; c.irr[2] = ir.z; c.irr[3] = ir.w;
		const uint32_t oc[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (oc[j]) { // existing inner node: merged away when below the threshold (quadtree.py:586, 595)
				const bool keep = !(c.irr[j] < thr);
				c.nonleaf[j] = keep && (c.irr[j] >= thr);
				c.old_child[j] = c.nonleaf[j] ? oc[j] : 0;
			} else {     // existing leaf: split while above the threshold (quadtree.py:626)
				c.nonleaf[j] = c.irr[j] > thr && child_depth < max_depth;
				c.old_child[j] = 0;
			}
		}
	} else {
		const float q = p.value / 4.0f; // quadtree.py:133-134
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			c.irr[j] = q;
			c.nonleaf[j] = q > thr && child_depth < max_depth;
			c.old_child[j] = 0;
		}
	}
	return c;
}

// level 0: one entry per NEW tree
__global__ __launch_bounds__(kBlk) void k_roots_plan(const uint32_t *__restrict__ tree_src,
                                                     const TreeHead *__restrict__ old_head,
                                                     const float *__restrict__ root_irr, uint32_t n_new,
                                                     int max_depth, TreeHead *__restrict__ new_head,
                                                     float *__restrict__ new_thr, uint32_t *__restrict__ cnt)
{
	const uint32_t t = blockIdx.x * kBlk + threadIdx.x;
	if (t >= n_new) return;
	const uint32_t src = tree_src[t];
	const float irr = root_irr[src];
	const float thr = irr / 100.0f; // quadtree.py:519
	const uint32_t rr = old_head[src].root_rec;
	bool nonleaf;
	if (rr != kNoRecord) nonleaf = !(irr < thr) && (irr >= thr); // merge pass on the root
	else nonleaf = irr > thr && 0 < max_depth;                  // split pass on a leaf root
	TreeHead h;
	h.root_rec = nonleaf ? 0u : kNoRecord; // record index patched by k_roots_emit
	h.root_irr = irr;
	new_head[t] = h;
	new_thr[t] = thr;
	cnt[t] = nonleaf ? 1u : 0u;
}

__global__ __launch_bounds__(kBlk) void k_roots_emit(const uint32_t *__restrict__ tree_src,
                                                     const TreeHead *__restrict__ old_head, uint32_t n_new,
                                                     const uint32_t *__restrict__ cnt,
                                                     const uint32_t *__restrict__ pos,
                                                     TreeHead *__restrict__ new_head, Pending *__restrict__ out)
{
	const uint32_t t = blockIdx.x * kBlk + threadIdx.x;
	if (t >= n_new || cnt[t] == 0) return;
	const uint32_t src = tree_src[t];
	Pending p;
	p.old_rec = old_head[src].root_rec; // kNoRecord when the old root was a leaf that now splits
	p.value = new_head[t].root_irr;
	p.tree = t;
	p.pad = 0;
	out[pos[t]] = p;
	new_head[t].root_rec = pos[t]; // level 0 starts at record 0
}

__global__ __launch_bounds__(kBlk) void k_level_plan(const Pending *__restrict__ pend, uint32_t n,
                                                     const QuadRec *__restrict__ rec,
                                                     const float *__restrict__ slot_irr,
                                                     const float *__restrict__ new_thr, int child_depth,
                                                     int max_depth, uint32_t *__restrict__ cnt)
{
	const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
	if (i >= n) return;
	const Pending p = pend[i];
	const ChildPlan c = plan_children(p, rec, slot_irr, new_thr[p.tree], child_depth, max_depth);
	cnt[i] = (uint32_t)c.nonleaf[0] + (uint32_t)c.nonleaf[1] + (uint32_t)c.nonleaf[2] + (uint32_t)c.nonleaf[3];
}

__global__ __launch_bounds__(kBlk) void k_level_emit(const Pending *__restrict__ pend, uint32_t n,
                                                     const QuadRec *__restrict__ rec,
                                                     const float *__restrict__ slot_irr,
                                                     const float *__restrict__ new_thr, int child_depth,
                                                     int max_depth, const uint32_t *__restrict__ pos,
                                                     uint32_t level_off, uint32_t next_off,
                                                     QuadRec *__restrict__ new_rec, Pending *__restrict__ next)
{
	const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
	if (i >= n) return;
	const Pending p = pend[i];
	const ChildPlan c = plan_children(p, rec, slot_irr, new_thr[p.tree], child_depth, max_depth);
	uint32_t k = pos[i];
	uint32_t child[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		child[j] = 0;
		if (c.nonleaf[j]) {
			Pending q;
			q.old_rec = c.old_child[j] ? c.old_child[j] : kNoRecord;
			q.value = c.irr[j];
			q.tree = p.tree;
			q.pad = 0;
			next[k] = q;
			child[j] = next_off + k;
			++k;
		}
	}
	float4 *dst = reinterpret_cast<float4 *>(new_rec + level_off + i);
	dst[0] = make_float4(c.irr[0], c.irr[1], c.irr[2], c.irr[3]);
	reinterpret_cast<uint4 *>(dst)[1] = make_uint4(child[0], child[1], child[2], child[3]);
}

__global__ __launch_bounds__(kBlk) void k_iota(uint32_t *p, uint32_t n)
{
	const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
	if (i < n) p[i] = i;
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + kBlk - 1) / kBlk)); }

// Growth policy of everything that follows the tree: twice what is needed now.  The tree roughly doubles per training
// iteration (2^(k+2) spp), so doubling leaves room for the next refine; what matters more is that NOTHING is allocated or
// freed by a refine whose tree fits what the last one left (hipFree synchronises the device, a hipMalloc of a gigabyte
// takes tens of milliseconds: round 3's refine spent 20-190 ms there).
constexpr double kGrow = 2.0;

struct Scanner {
	DevBuf<uint32_t> &sums, &total;
	uint32_t *h_total; // page-locked
	// out[i] = sum(in[0..i)), returns grand total (synchronises the stream)
	int run(pg_context *ctx, const uint32_t *in, uint32_t n, uint32_t *out, uint32_t &tot, hipStream_t s)
	{
		tot = 0;
		if (n == 0) return PG_OK;
		const uint32_t nb = (n + kScanTile - 1) / kScanTile;
		PG_HIP(ctx, sums.ensure(nb, kGrow));
		PG_HIP(ctx, total.ensure(1));
		hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(kBlk), 0, s, in, n, sums.p);
		hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, sums.p, nb, total.p);
		hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kBlk), 0, s, in, n, sums.p, out);
		PG_HIP(ctx, hipGetLastError());
		PG_HIP(ctx, hipMemcpyAsync(h_total, total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		PG_HIP(ctx, hipStreamSynchronize(s));
		tot = *h_total;
		return PG_OK;
	}
};

template <class T> static hipError_t grow_preserve(DevBuf<T> &b, size_t keep, size_t need, hipStream_t s)
{
	if (need <= b.cap) return hipSuccess;
	DevBuf<T> n;
	hipError_t e = n.ensure(need, kGrow);
	if (e != hipSuccess) return e;
	if (keep) {
		e = hipMemcpyAsync(n.p, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, s);
		if (e != hipSuccess) return e;
		e = hipStreamSynchronize(s);
		if (e != hipSuccess) return e;
	}
	b.swap(n);
	return hipSuccess;
}

// PGSD_TRACE_REFINE=1: wall time of every phase of a refine on stderr (each mark synchronises the stream: diagnosis only)
struct PhaseTrace {
	bool on;
	hipStream_t s;
	std::chrono::steady_clock::time_point t0;
	PhaseTrace(hipStream_t s_) : on(getenv("PGSD_TRACE_REFINE") != nullptr), s(s_), t0(std::chrono::steady_clock::now()) {}
	void mark(const char *what)
	{
		if (!on) return;
		(void)hipStreamSynchronize(s);
		const auto t1 = std::chrono::steady_clock::now();
		fprintf(stderr, "[pgsd refine] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
		t0 = t1;
	}
};

int refine_and_swap(pg_context *ctx, hipStream_t s)
{
	PhaseTrace tr(s);
	Forest &f = ctx->f;
	Forest::RefineScratch &w = f.scratch;
	const AccumView av = f.accum_view();
	const uint32_t L = (uint32_t)f.level_off.size() - 1;
	if (!w.pinned) PG_HIP(ctx, hipHostMalloc(&w.pinned, 64, hipHostMallocDefault));
	unsigned long long *h_plan = static_cast<unsigned long long *>(w.pinned);       // [2]
	uint32_t *h_counts = reinterpret_cast<uint32_t *>(h_plan + 2);                   // [2]
	uint32_t *h_total = h_counts + 2;                                                // [1]

	// ======== A. everything that can fail: reads the forest, writes scratch only ========
	// ---- 1. resolve --------------------------------------------------------------------------
	PG_HIP(ctx, w.tot.ensure(f.n_rec, kGrow));
	PG_HIP(ctx, w.cnt_tot.ensure(f.n_rec, kGrow));
	PG_HIP(ctx, w.slot_irr.ensure((size_t)f.n_rec * 4, kGrow));
	PG_HIP(ctx, w.root_irr.ensure(f.n_trees, kGrow));
	PG_HIP(ctx, w.tree_count.ensure(f.n_trees, kGrow));
	for (int l = (int)L - 1; l >= 0; --l) {
		const uint32_t b = f.level_off[l], e = f.level_off[l + 1];
		if (e > b)
			hipLaunchKernelGGL(k_resolve_level, grid_for(e - b), dim3(kBlk), 0, s, f.rec.p, av.rec_acc, w.tot.p,
			                   w.cnt_tot.p, w.slot_irr.p, b, e);
	}
	hipLaunchKernelGGL(k_resolve_roots, grid_for(f.n_trees), dim3(kBlk), 0, s, f.head.p, av.root_acc, w.tot.p,
	                   w.cnt_tot.p, av.leaf_count, w.root_irr.p, w.tree_count.p, f.n_trees);
	PG_HIP(ctx, w.kd_cnt.ensure(f.n_kd, kGrow));
	hipLaunchKernelGGL(k_kd_counts, grid_for(f.n_kd), dim3(kBlk), 0, s, f.kd.p, f.n_kd, w.tree_count.p, w.kd_cnt.p, 0, 1);
	for (int d = ctx->kd_max_depth - 1; d >= 0; --d)
		hipLaunchKernelGGL(k_kd_counts, grid_for(f.n_kd), dim3(kBlk), 0, s, f.kd.p, f.n_kd, w.tree_count.p, w.kd_cnt.p, d, 0);
	// (the vertCount column of the NEW sdTree_prev: the old one's stays in f.kd_vcount until the commit)
	PG_HIP(ctx, w.new_vc.ensure(f.n_kd, kGrow));
	hipLaunchKernelGGL(k_kd_vcount, grid_for(f.n_kd), dim3(kBlk), 0, s, w.kd_cnt.p, w.new_vc.p, f.n_kd);
	PG_HIP(ctx, hipGetLastError());

	tr.mark("resolve");
	// ---- 2. KD refine, in a copy of the node arrays ------------------------------------------
	const double kd_max_leaf_size = 12000.0 * sqrt(pow(2.0, (double)ctx->iteration)); // kdtree.py:327-330
	const float kd_thr = (float)kd_max_leaf_size;
	PG_HIP(ctx, w.plan.ensure(2));
	PG_HIP(ctx, hipMemsetAsync(w.plan.p, 0, 2 * sizeof(unsigned long long), s));
	hipLaunchKernelGGL(k_kd_plan, grid_for(f.n_kd), dim3(kBlk), 0, s, f.kd.p, f.n_kd, w.new_vc.p, kd_thr,
	                   ctx->kd_max_depth, w.plan.p);
	PG_HIP(ctx, hipGetLastError());
	PG_HIP(ctx, hipMemcpyAsync(h_plan, w.plan.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
	PG_HIP(ctx, hipStreamSynchronize(s));
	const uint64_t want_kd = (uint64_t)f.n_kd + h_plan[0], want_trees = (uint64_t)f.n_trees + h_plan[1];
	if (want_kd > 0x7fffffffull) return fail(ctx, PG_ERR_NOMEM, "refine: KD tree would exceed 2^31 nodes");
	PG_HIP(ctx, w.new_kd.ensure(want_kd, kGrow));
	PG_HIP(ctx, w.new_bmin.ensure(want_kd * 3, kGrow));
	PG_HIP(ctx, w.new_bmax.ensure(want_kd * 3, kGrow));
	PG_HIP(ctx, grow_preserve(w.new_vc, f.n_kd, want_kd, s));
	PG_HIP(ctx, hipMemcpyAsync(w.new_kd.p, f.kd.p, (size_t)f.n_kd * sizeof(KdNode), hipMemcpyDeviceToDevice, s));
	PG_HIP(ctx, hipMemcpyAsync(w.new_bmin.p, f.kd_bmin.p, (size_t)f.n_kd * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
	PG_HIP(ctx, hipMemcpyAsync(w.new_bmax.p, f.kd_bmax.p, (size_t)f.n_kd * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
	PG_HIP(ctx, w.tree_src.ensure(want_trees, kGrow));
	PG_HIP(ctx, w.counts.ensure(2));
	hipLaunchKernelGGL(k_iota, grid_for(f.n_trees), dim3(kBlk), 0, s, w.tree_src.p, f.n_trees);
	uint32_t n_kd_new = f.n_kd, n_trees_new = f.n_trees;
	if (h_plan[0]) {
		KdRefineArgs a;
		a.kd = w.new_kd.p; a.bmin = w.new_bmin.p; a.bmax = w.new_bmax.p; a.vc = w.new_vc.p;
		a.tree_src = w.tree_src.p; a.n_kd = f.n_kd; a.n_trees = f.n_trees; a.thr = kd_thr;
		a.max_depth = ctx->kd_max_depth; a.out_counts = w.counts.p;
		hipLaunchKernelGGL(k_kd_refine, dim3(1), dim3(1024), 0, s, a);
		PG_HIP(ctx, hipGetLastError());
		PG_HIP(ctx, hipMemcpyAsync(h_counts, w.counts.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		PG_HIP(ctx, hipStreamSynchronize(s));
		n_kd_new = h_counts[0];
		n_trees_new = h_counts[1];
		if (n_kd_new != want_kd || n_trees_new != want_trees)
			return fail(ctx, PG_ERR_INVALID, "refine: KD plan and KD split rounds disagree (internal error)");
	}

	tr.mark("kd refine");
	// ---- 3. quadtree rebuild -----------------------------------------------------------------
	const int qmax = ctx->quad_max_depth;
	Scanner scan{w.scan_sums, w.scan_total, h_total};
	PG_HIP(ctx, w.new_head.ensure(n_trees_new, kGrow));
	PG_HIP(ctx, w.new_thr.ensure(n_trees_new, kGrow));
	PG_HIP(ctx, w.cnt.ensure(n_trees_new, kGrow));
	PG_HIP(ctx, w.pos.ensure(n_trees_new, kGrow));
	hipLaunchKernelGGL(k_roots_plan, grid_for(n_trees_new), dim3(kBlk), 0, s, w.tree_src.p, f.head.p, w.root_irr.p,
	                   n_trees_new, qmax, w.new_head.p, w.new_thr.p, w.cnt.p);
	PG_HIP(ctx, hipGetLastError());
	uint32_t n_level = 0;
	int rc = scan.run(ctx, w.cnt.p, n_trees_new, w.pos.p, n_level, s);
	if (rc != PG_OK) return rc;
	PG_HIP(ctx, w.pend_a.ensure((size_t)n_level * sizeof(Pending), kGrow));
	if (n_level)
		hipLaunchKernelGGL(k_roots_emit, grid_for(n_trees_new), dim3(kBlk), 0, s, w.tree_src.p, f.head.p, n_trees_new,
		                   w.cnt.p, w.pos.p, w.new_head.p, reinterpret_cast<Pending *>(w.pend_a.p));
	PG_HIP(ctx, hipGetLastError());
	std::vector<uint32_t> new_level_off(1, 0u);
	uint64_t off = 0;
	int level = 0;
	// (the new forest is rarely smaller than the old one: start the record buffer there, so that it grows at most once or twice)
	PG_HIP(ctx, w.new_rec.ensure(f.n_rec, kGrow));
	while (n_level) {
		if (level >= kMaxLevels - 2) return fail(ctx, PG_ERR_INVALID, "refine: quadtree deeper than supported");
		if (off + n_level > 0xfffffff0ull) return fail(ctx, PG_ERR_NOMEM, "refine: more than 2^32 quadtree records");
		PG_HIP(ctx, grow_preserve(w.new_rec, (size_t)off, (size_t)off + n_level, s));
		PG_HIP(ctx, w.cnt.ensure(n_level, kGrow));
		PG_HIP(ctx, w.pos.ensure(n_level, kGrow));
		Pending *pa = reinterpret_cast<Pending *>(w.pend_a.p);
		hipLaunchKernelGGL(k_level_plan, grid_for(n_level), dim3(kBlk), 0, s, pa, n_level, f.rec.p, w.slot_irr.p,
		                   w.new_thr.p, level + 1, qmax, w.cnt.p);
		PG_HIP(ctx, hipGetLastError());
		uint32_t n_next = 0;
		rc = scan.run(ctx, w.cnt.p, n_level, w.pos.p, n_next, s);
		if (rc != PG_OK) return rc;
		PG_HIP(ctx, w.pend_b.ensure((size_t)n_next * sizeof(Pending), kGrow));
		hipLaunchKernelGGL(k_level_emit, grid_for(n_level), dim3(kBlk), 0, s, pa, n_level, f.rec.p, w.slot_irr.p,
		                   w.new_thr.p, level + 1, qmax, w.pos.p, (uint32_t)off, (uint32_t)(off + n_level), w.new_rec.p,
		                   reinterpret_cast<Pending *>(w.pend_b.p));
		PG_HIP(ctx, hipGetLastError());
		off += n_level;
		new_level_off.push_back((uint32_t)off);
		w.pend_a.swap(w.pend_b);
		n_level = n_next;
		++level;
	}

	tr.mark("quadtree rebuild");
	// ---- the next iteration's accumulators (sdTree_current after the reset, path_guiding_integrator.py:583-586): zeroed in
	// the spare buffer; the running iteration's stay in f.acc until the commit
	const uint64_t new_acc_count = (uint64_t)off * 4 * kAccWords + (uint64_t)n_trees_new * kAccWords + n_trees_new;
	PG_HIP(ctx, w.new_acc.ensure(new_acc_count, kGrow));
	PG_HIP(ctx, hipMemsetAsync(w.new_acc.p, 0, new_acc_count * sizeof(long long), s));
	PG_HIP(ctx, hipStreamSynchronize(s)); // (the last call of the transaction that can fail)

	// ======== B. commit: pointer swaps and counts, nothing here can fail ========
	f.kd.swap(w.new_kd);       // (what the forest held becomes the next refine's scratch)
	f.kd_bmin.swap(w.new_bmin);
	f.kd_bmax.swap(w.new_bmax);
	f.kd_vcount.swap(w.new_vc);
	f.rec.swap(w.new_rec);
	f.head.swap(w.new_head);
	f.tree_thr.swap(w.new_thr);
	f.acc.swap(w.new_acc);
	f.n_rec = (uint32_t)off;
	f.n_trees = n_trees_new;
	f.n_kd = n_kd_new;
	f.level_off.swap(new_level_off);
	ctx->kd_max_leaf_size = kd_max_leaf_size;
	tr.mark("swap + accumulators");

	// ======== C. the accelerators of the new forest.  They are optional: a forest without them is walked from its roots by
	// every consumer with the same results, so a failure here (it would be a device fault: their buffers were sized by
	// setup / an earlier refine, or fall back by themselves) leaves a valid, slower forest and is not an error of the refine ========
	const std::string err_keep = ctx->err;
	if (rebuild_jump(ctx, s) != PG_OK || hipStreamSynchronize(s) != hipSuccess) {
		(void)hipGetLastError();
		f.jump_valid = false;
		f.kd_grid_valid = false;
		ctx->err = err_keep;
	}
	tr.mark("jump tables");
	return PG_OK;
}

// The jump table follows the quadtree records: built after setup, import and every refine.
int rebuild_jump(pg_context *ctx, hipStream_t s)
{
	Forest &f = ctx->f;
	f.jump_valid = false;
	// the KD jump grid follows the KD tree (a degenerate root box gets none: every query then descends from the root)
	f.kd_grid_valid = false;
	if (f.n_kd > 0 && ctx->bmax[0] > ctx->bmin[0] && ctx->bmax[1] > ctx->bmin[1] && ctx->bmax[2] > ctx->bmin[2]) {
		// the cell boundaries: [bmin, bmax] bisected like the tree bisects its boxes, mid = (lo + hi) / 2 in fp32
		// (kdtree.py:270; k_kd_refine above) -- they must ascend strictly for the grid to be usable
		// resolution: one level more than a balanced tree of this many leaves has (pg_tree.hpp)
		int bits = 1;
		while (bits < kKdGridBits && (1ull << (3 * (bits - 1))) < ((uint64_t)f.n_kd + 1) / 2) ++bits;
		f.kd_grid_bits = bits;
		const int G = 1 << bits;
		float planes[3 * kKdGridPlanes] = {};
		bool ok = true;
		for (int a = 0; a < 3; ++a) {
			float *P = planes + a * kKdGridPlanes;
			P[0] = ctx->bmin[a];
			P[G] = ctx->bmax[a];
			for (int step = G / 2; step >= 1; step /= 2)
				for (int i = step; i < G; i += 2 * step) P[i] = (P[i - step] + P[i + step]) / 2.0f;
			for (int i = 0; i < G; ++i) ok = ok && P[i] < P[i + 1];
		}
		TreeView tg = ctx->view();
		ok = ok && tg.grid_inv[0] < 3.0e38f && tg.grid_inv[1] < 3.0e38f && tg.grid_inv[2] < 3.0e38f;
		if (ok) {
			PG_HIP(ctx, f.kd_grid.ensure((size_t)kKdGridCells + kKdGridRootEntries)); // (the finest grid, 4 MB, once: the resolution follows the tree)
			PG_HIP(ctx, f.kd_planes.ensure(3 * kKdGridPlanes));
			PG_HIP(ctx, hipMemcpyAsync(f.kd_planes.p, planes, sizeof planes, hipMemcpyHostToDevice, s));
			PG_HIP(ctx, hipStreamSynchronize(s)); // (planes[] is on this stack frame)
			tg = ctx->view();
			launch_build_kd_grid(tg, f.kd_grid.p, s);
			PG_HIP(ctx, hipGetLastError());
			f.kd_grid_valid = true;
		}
	}
	// The quadtree jump tables.  Accumulator slots are packed into 26 bits of an entry: forests beyond that walk every
	// level.  Memory: 16 B << 2 * bits per quadtree (64 KB with six bits: 1.4 GB for the 21 000 trees of the veach-ajar
	// bench) against a budget -- $PGSD_JUMP_TABLE_MAX_BYTES, default 2 GiB: a forest too big for six bits gets five (a
	// quarter of the memory; measured on the veach-ajar bench, ms per step: 6 bits 56.3, 5 bits 56.4, 4 bits 56.9, 3 bits
	// 57.5), then four, ...  The resolution is a function of the FOREST and the budget alone, so that every rank of a
	// sharded run, and every run of a bench, builds the same tables (round 4 also looked at the memory that happened to be
	// free: results were the same bit for bit, timings and pg_stats were not -- ADVICE r4).  Only when the table cannot be
	// ALLOCATED, or the device refuses the LDS its builder needs, does a forest get a coarser table than that, or none:
	// it then walks every level from the root, which every consumer handles (a failed allocation never fails a refine).
	if (f.n_trees == 0 || (uint64_t)f.n_rec * 4ull > (uint64_t)kJumpSlotMask) return PG_OK;
	const uint64_t budget = ctx->jump_budget;
	int bits = kJumpBits;
	while (bits >= 2 && (((uint64_t)f.n_trees * sizeof(QuadJump)) << (2 * bits)) > budget) --bits;
	for (; bits >= 2; --bits) { // (a 2 x 2 table saves one level: below that nothing)
		const size_t need = (size_t)f.n_trees << (2 * bits);
		if (need > f.jump.cap) {
			// room for the forest to double, as long as that stays within the budget
			size_t want = need * 2;
			if ((uint64_t)want * sizeof(QuadJump) > budget) want = need;
			if (f.jump.ensure(want) != hipSuccess) {
				(void)hipGetLastError(); // (not sticky)
				if (f.jump.ensure(need) != hipSuccess) {
					(void)hipGetLastError();
					continue; // a quarter of the memory
				}
			}
		}
		TreeView t = ctx->view();
		if (!launch_build_jump(t, f.jump.p, bits, s)) continue;
		PG_HIP(ctx, hipGetLastError());
		f.jump_bits = bits;
		f.jump_valid = true;
		break;
	}
	return PG_OK;
}

} // namespace pg
