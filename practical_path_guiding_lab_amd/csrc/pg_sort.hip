// pg_sort.hip -- a global order for the live list of a bounce (pg_render_sort, pg_render_wave.hip): the places of the
// list grouped by a 16-bit spatial key (the Morton cell of the vertex the ray has just found, the lane's class in its
// lowest bit), so that the lanes of a wave stand near each other in the scene -- in one KD leaf, under one quadtree, in
// front of the same BVH nodes -- for every kernel that follows.  Not in the reference (Dr.Jit's wavefront keeps pixel
// order); a lane's result depends on its own state only, so the order is free.
//
// Rounds 3 and 4 called rocPRIM's device radix sort (two onesweep passes of eight bits; its kernel for (u16, u32) pairs
// holds 96 registers and spills 80 bytes per lane).  Round 5 writes the sort for what it is needed for: two counting passes
// of eight bits -- low digit, then high digit -- over tiles of 4096 pairs:
//   k_sort_hist     one workgroup per tile: the tile's 256 digit counts (LDS atomics), stored digit-major
//   k_sort_scan_*   one exclusive scan over the 256 x tiles counts: where a (digit, tile) run starts in the output
//   k_sort_scatter  the tile again: every pair gets its rank inside its digit of the tile from an LDS atomic (the order
//                   INSIDE a (digit, tile) run is whatever the atomics give), the tile is laid out digit by digit in
//                   LDS and leaves as runs of consecutive addresses
// Neither pass is stable, and the result is not a sort: it is a permutation in which the places are in the order of the
// key's HIGH byte, and inside one value of it in the order of the low byte up to the pairs of ONE tile of the second pass --
// 4096 consecutive pairs of a sequence sorted by the low byte.  How much low-byte order that is DEPENDS ON n:
// a tile of the second pass spans 4096 * 256 / n values of the low byte on average -- equal or adjacent low bytes from about
// 2^21 places up (the 33 M lanes of the bench's step: an eighth of one value), 4 values at 2^18, and NO low-byte order at all
// below 2^12 * 2^8 = 2^20 places, where one tile holds every low byte: the one or two million lanes of a cornell-box pass or
// of a late bounce are ordered by the 8 high bits of the Morton key (and the class bit, bit 0, is not grouped there).
// The permutation also differs from run to run (LDS atomics; and the list it is applied to was appended by atomics).
// That is all the order the shading needs (cells of the spatial sort are 2^-15 of the box; the results do not depend on the
// order at all), and it is what makes the passes cheap: no decoupled look-back, no ranking by match-any, 27 KB of LDS.
// Measured against the stable form (ballot-match ranking, the true LSD sort: the same answer as numpy's stable argsort): the
// full 16 bits are worth 2-3 % of the two SD-tree-walking kernels on lists of a million lanes and nothing on long ones, and
// cost the sort 25 % -- a net loss of 1-2 % of the step at every size but the torus' (profiles/r06/ab_sort_stable_rejected.txt).
// The first pass reads no places (a pair's place is its index) and the second writes no keys.
// Because the order inside a tile of the second pass is free, a place WITHOUT a path (key 0xffff) could come to stand in front
// of a live one with key 0xfffe or 0xfffd.  The live list is dense -- places 0 .. live-1 hold paths, the rest of the n places
// the host sized the sort for hold none -- so the kernels read the live count from device memory (`d_live`) and never look at
// the places beyond it: places_out[0 .. live) is a permutation of 0 .. live-1, what lies behind it is not written.
#include <stdint.h>


#include "pg_context.hpp"

namespace pg {


namespace {

constexpr int kSortBlock = 256;
constexpr int kSortItems = 16;                          // pairs per thread
constexpr uint32_t kSortTile = kSortBlock * kSortItems; // 4096 pairs per workgroup
constexpr uint32_t kSortChunk = 4096;                   // counts per workgroup of the scan

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
		if ((int)(threadIdx.x & 63u) >= d) v += o;
	}
	return v;
}

// exclusive scan of one value per thread over the workgroup (256 threads); s_wave: four words of LDS; total: the sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave, uint32_t &total)
{
	const uint32_t inc = wave_inclusive_scan(v);
	const uint32_t w = threadIdx.x >> 6;
	if ((threadIdx.x & 63u) == 63u) s_wave[w] = inc;
	__syncthreads();
	uint32_t base = 0;
	for (uint32_t k = 0; k < w; ++k) base += s_wave[k];
	total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
	__syncthreads();
	return base + inc - v;
}

template <int kShift>
__global__ __launch_bounds__(kSortBlock) void k_sort_hist(const uint16_t *__restrict__ keys, uint32_t n, const uint32_t *__restrict__ d_live,
                                                          uint32_t n_tiles, uint32_t *__restrict__ counts)
{
	__shared__ uint32_t s_hist[256];
	if (d_live != nullptr && *d_live < n) n = *d_live; // (uniform)
	s_hist[threadIdx.x] = 0u;
	__syncthreads();
	const uint32_t first = blockIdx.x * kSortTile + threadIdx.x * kSortItems;
	if (first + kSortItems <= n) { // sixteen keys of this thread: two 16-byte loads
		const uint4 a = reinterpret_cast<const uint4 *>(keys + first)[0], b = reinterpret_cast<const uint4 *>(keys + first)[1];
		const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			atomicAdd(&s_hist[(w[k] >> kShift) & 255u], 1u);
			atomicAdd(&s_hist[(w[k] >> (16 + kShift)) & 255u], 1u);
		}
	} else {
		for (uint32_t i = first; i < n && i < first + kSortItems; ++i) atomicAdd(&s_hist[((uint32_t)keys[i] >> kShift) & 255u], 1u);
	}
	__syncthreads();
	counts[(uint64_t)threadIdx.x * n_tiles + blockIdx.x] = s_hist[threadIdx.x]; // digit-major: a digit's tiles side by side
}

// exclusive scan of `m` counts in place, chunk by chunk; the chunks' sums go to chunk_sum
__global__ __launch_bounds__(kSortBlock) void k_sort_scan_chunks(uint32_t *__restrict__ counts, uint32_t m, uint32_t *__restrict__ chunk_sum)
{
	__shared__ uint32_t s_wave[4];
	const uint32_t first = blockIdx.x * kSortChunk + threadIdx.x * 16u;
	uint32_t v[16], sum = 0;
#pragma unroll
	for (int k = 0; k < 16; ++k) {
		v[k] = first + k < m ? counts[first + k] : 0u;
		sum += v[k];
	}
	uint32_t total;
	uint32_t run = block_exclusive_scan(sum, s_wave, total);
#pragma unroll
	for (int k = 0; k < 16; ++k) {
		if (first + k < m) counts[first + k] = run;
		run += v[k];
	}
	if (threadIdx.x == 0) chunk_sum[blockIdx.x] = total;
}

// exclusive scan of the chunks' sums (one workgroup; up to 256 x 16 chunks = 16 M counts = 256 digits x 65536 tiles)
__global__ __launch_bounds__(kSortBlock) void k_sort_scan_sums(uint32_t *__restrict__ chunk_sum, uint32_t n_chunks)
{
	__shared__ uint32_t s_wave[4];
	const uint32_t first = threadIdx.x * 16u;
	uint32_t v[16], sum = 0;
#pragma unroll
	for (int k = 0; k < 16; ++k) {
		v[k] = first + k < n_chunks ? chunk_sum[first + k] : 0u;
		sum += v[k];
	}
	uint32_t total;
	uint32_t run = block_exclusive_scan(sum, s_wave, total);
#pragma unroll
	for (int k = 0; k < 16; ++k) {
		if (first + k < n_chunks) chunk_sum[first + k] = run;
		run += v[k];
	}
}

// kFirst: the places are the pairs' indices (not read); kLast: the keys are not written
template <int kShift, bool kFirst, bool kLast>
__global__ __launch_bounds__(kSortBlock) void k_sort_scatter(const uint16_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                             uint32_t n, const uint32_t *__restrict__ d_live, uint32_t n_tiles,
                                                             const uint32_t *__restrict__ counts, const uint32_t *__restrict__ chunk_base,
                                                             uint16_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out)
{
	if (d_live != nullptr && *d_live < n) n = *d_live; // (uniform)
	if (blockIdx.x * kSortTile >= n) return;           // a tile past the live list: nothing to place
	__shared__ uint32_t s_cnt[256];   // the tile's count per digit, then where the digit's run starts in the tile
	__shared__ uint32_t s_gbase[256]; // where the digit's run of this tile starts in the output
	__shared__ uint32_t s_wave[4];
	__shared__ uint32_t s_val[kSortTile];
	__shared__ uint16_t s_key[kSortTile];
	s_cnt[threadIdx.x] = 0u;
	{
		const uint64_t at = (uint64_t)threadIdx.x * n_tiles + blockIdx.x;
		s_gbase[threadIdx.x] = counts[at] + chunk_base[at / kSortChunk];
	}
	__syncthreads();
	const uint32_t tile0 = blockIdx.x * kSortTile;
	const uint32_t first = tile0 + threadIdx.x * kSortItems;
	uint32_t key[kSortItems], rank[kSortItems];
	if (first + kSortItems <= n) {
		const uint4 a = reinterpret_cast<const uint4 *>(keys_in + first)[0], b = reinterpret_cast<const uint4 *>(keys_in + first)[1];
		const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			key[2 * k] = w[k] & 0xffffu;
			key[2 * k + 1] = w[k] >> 16;
		}
	} else {
#pragma unroll
		for (int k = 0; k < kSortItems; ++k) key[k] = first + k < n ? (uint32_t)keys_in[first + k] : 0x10000u; // (bit 16: no pair)
	}
#pragma unroll
	for (int k = 0; k < kSortItems; ++k) rank[k] = key[k] < 0x10000u ? atomicAdd(&s_cnt[(key[k] >> kShift) & 255u], 1u) : 0u;
	__syncthreads();
	{
		uint32_t total;
		const uint32_t c = s_cnt[threadIdx.x];
		const uint32_t ex = block_exclusive_scan(c, s_wave, total);
		s_cnt[threadIdx.x] = ex;
	}
	__syncthreads();
	if (kFirst) {
#pragma unroll
		for (int k = 0; k < kSortItems; ++k)
			if (key[k] < 0x10000u) {
				const uint32_t at = s_cnt[(key[k] >> kShift) & 255u] + rank[k];
				s_key[at] = (uint16_t)key[k];
				s_val[at] = first + k;
			}
	} else {
		uint32_t val[kSortItems];
		if (first + kSortItems <= n) {
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const uint4 v = reinterpret_cast<const uint4 *>(vals_in + first)[q];
				val[4 * q] = v.x; val[4 * q + 1] = v.y; val[4 * q + 2] = v.z; val[4 * q + 3] = v.w;
			}
		} else {
#pragma unroll
			for (int k = 0; k < kSortItems; ++k) val[k] = first + k < n ? vals_in[first + k] : 0u;
		}
#pragma unroll
		for (int k = 0; k < kSortItems; ++k)
			if (key[k] < 0x10000u) {
				const uint32_t at = s_cnt[(key[k] >> kShift) & 255u] + rank[k];
				s_key[at] = (uint16_t)key[k];
				s_val[at] = val[k];
			}
	}
	__syncthreads();
	// the tile, digit by digit, leaves as runs of consecutive addresses (consecutive threads, consecutive pairs of a run)
	const uint32_t in_tile = n - tile0 < kSortTile ? n - tile0 : kSortTile;
#pragma unroll
	for (int k = 0; k < kSortItems; ++k) {
		const uint32_t j = (uint32_t)k * kSortBlock + threadIdx.x;
		if (j < in_tile) {
			const uint32_t kk = s_key[j], d = (kk >> kShift) & 255u;
			const uint32_t to = s_gbase[d] + (j - s_cnt[d]);
			if (!kLast) keys_out[to] = (uint16_t)kk;
			vals_out[to] = s_val[j];
		}
	}
}

struct SortPlan {
	uint32_t n_tiles, m, n_chunks;
	size_t off_counts, off_sums, off_vals, bytes;
};
SortPlan sort_plan(uint32_t n)
{
	SortPlan p;
	p.n_tiles = (n + kSortTile - 1) / kSortTile;
	p.m = 256u * p.n_tiles;
	p.n_chunks = (p.m + kSortChunk - 1) / kSortChunk;
	auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
	p.off_counts = 0;
	p.off_sums = up((size_t)p.m * 4);
	p.off_vals = p.off_sums + up((size_t)p.n_chunks * 4);
	p.bytes = p.off_vals + up((size_t)n * 4);
	return p;
}

} // namespace

// temporary storage for n pairs: the 256 x tiles counts, their chunks' sums, and the places between the two passes
size_t sort_pairs_temp_bytes(uint32_t n)
{
	return sort_plan(n ? n : 1u).bytes;
}

// places_out[0 .. live) = a permutation of 0 .. live-1 in the order described at the top of this file, live = min(n, *d_live)
// (d_live: the number of places that hold a path, in device memory; nullptr: all n do); keys_out: the keys between the two
// passes -- scratch of n entries; asynchronous on `s`.  n < 2^28 (4096 x 65536 tiles): pg_render_pass keeps the lanes of a
// pass below that.
hipError_t sort_places16(void *temp, size_t temp_bytes, const uint16_t *keys_in, uint16_t *keys_out, uint32_t *places_out, uint32_t n,
                         const uint32_t *d_live, hipStream_t s)
{
	if (n == 0) return hipSuccess;
	const SortPlan p = sort_plan(n);
	if (!temp || temp_bytes < p.bytes || p.n_chunks > 4096u) return hipErrorInvalidValue;
	uint32_t *counts = reinterpret_cast<uint32_t *>((char *)temp + p.off_counts);
	uint32_t *sums = reinterpret_cast<uint32_t *>((char *)temp + p.off_sums);
	uint32_t *vals = reinterpret_cast<uint32_t *>((char *)temp + p.off_vals);
	// the low byte
	hipLaunchKernelGGL(k_sort_hist<0>, dim3(p.n_tiles), dim3(kSortBlock), 0, s, keys_in, n, d_live, p.n_tiles, counts);
	hipLaunchKernelGGL(k_sort_scan_chunks, dim3(p.n_chunks), dim3(kSortBlock), 0, s, counts, p.m, sums);
	hipLaunchKernelGGL(k_sort_scan_sums, dim3(1), dim3(kSortBlock), 0, s, sums, p.n_chunks);
	hipLaunchKernelGGL((k_sort_scatter<0, true, false>), dim3(p.n_tiles), dim3(kSortBlock), 0, s, keys_in, (const uint32_t *)nullptr, n, d_live,
	                   p.n_tiles, counts, sums, keys_out, vals);
	// the high byte
	hipLaunchKernelGGL(k_sort_hist<8>, dim3(p.n_tiles), dim3(kSortBlock), 0, s, (const uint16_t *)keys_out, n, d_live, p.n_tiles, counts);
	hipLaunchKernelGGL(k_sort_scan_chunks, dim3(p.n_chunks), dim3(kSortBlock), 0, s, counts, p.m, sums);
	hipLaunchKernelGGL(k_sort_scan_sums, dim3(1), dim3(kSortBlock), 0, s, sums, p.n_chunks);
	hipLaunchKernelGGL((k_sort_scatter<8, false, true>), dim3(p.n_tiles), dim3(kSortBlock), 0, s, (const uint16_t *)keys_out, (const uint32_t *)vals, n,
	                   d_live, p.n_tiles, counts, sums, (uint16_t *)nullptr, places_out);
	return hipGetLastError();
}

} // namespace pg
