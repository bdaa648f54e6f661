// Dev probe: what would a quadtree record that holds TWO levels in one cache line buy a sampling walk?
// S1's forest (4096 complete quadtrees of depth 5: records at levels 0..4, 341 per tree, level-major over the forest)
// walked by 2^22 lanes with uniformly random trees and draws, two ways:
//   A  today's layout: one 32-byte record per level (two 16-byte gathers), five dependent lines per walk
//   B  128-byte blocks rooted at the records of levels 0, 2 and 4: a node's irradiances, its four children's irradiances
//      and the children's first-block words in ONE line -- three dependent lines per walk, the gathers between them hit it
// Both pick children by the CDF arithmetic of quad_sample_t (pg_descent.hpp).  Prints ms per launch and the sum (equal).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/quad_block_probe.hip -o /tmp/quad_block_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 gather16(const void *p)
{
	const u32x4_t v = *reinterpret_cast<const u32x4_t *>(p);
	return make_uint4(v.x, v.y, v.z, v.w);
}
__host__ __device__ __forceinline__ uint32_t hash32(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}
__host__ __device__ __forceinline__ float irr_of(uint32_t level, uint32_t tree, uint32_t j, uint32_t k)
{
	return (float)((hash32(((level * 4099u + tree) * 65537u + j) * 4u + k) >> 8) + 1u) * (1.0f / 16777216.0f);
}

constexpr uint32_t T = 4096, L = 5;
__host__ __device__ constexpr uint32_t pow4(uint32_t l) { return 1u << (2 * l); }
__host__ __device__ constexpr uint32_t base_of(uint32_t l) { return T * ((pow4(l) - 1u) / 3u); } // records of the levels above, forest-wide
// blocks exist for levels 0, 2, 4
__host__ __device__ constexpr uint32_t bbase_of(uint32_t l) { return l == 0 ? 0u : (l == 2 ? T : T + T * 16u); }

__global__ void k_init_a(uint4 *rec)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= base_of(L)) return;
	uint32_t l = 0;
	while (l + 1 < L && i >= base_of(l + 1)) ++l;
	const uint32_t o = i - base_of(l), t = o / pow4(l), j = o % pow4(l);
	uint4 a, b;
	a.x = __float_as_uint(irr_of(l, t, j, 0)); a.y = __float_as_uint(irr_of(l, t, j, 1));
	a.z = __float_as_uint(irr_of(l, t, j, 2)); a.w = __float_as_uint(irr_of(l, t, j, 3));
	const uint32_t c = l + 1 < L ? base_of(l + 1) + t * pow4(l + 1) + 4u * j : 0u;
	b = l + 1 < L ? make_uint4(c, c + 1, c + 2, c + 3) : make_uint4(0, 0, 0, 0);
	rec[2 * (size_t)i] = a;
	rec[2 * (size_t)i + 1] = b;
}

__global__ void k_init_b(uint4 *blk)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t n = T + T * 16u + T * 256u;
	if (i >= n) return;
	const uint32_t l = i < T ? 0u : (i < T + T * 16u ? 2u : 4u);
	const uint32_t o = i - bbase_of(l), t = o / pow4(l), j = o % pow4(l);
	uint4 *B = blk + 8 * (size_t)i;
	B[0] = make_uint4(__float_as_uint(irr_of(l, t, j, 0)), __float_as_uint(irr_of(l, t, j, 1)), __float_as_uint(irr_of(l, t, j, 2)),
	                  __float_as_uint(irr_of(l, t, j, 3)));
	uint32_t f[4] = {0, 0, 0, 0};
	for (uint32_t k = 0; k < 4; ++k) {
		if (l + 2 < L) f[k] = bbase_of(l + 2) + t * pow4(l + 2) + 16u * j + 4u * k;
		const uint32_t jc = 4u * j + k;
		B[2 + k] = l + 1 < L ? make_uint4(__float_as_uint(irr_of(l + 1, t, jc, 0)), __float_as_uint(irr_of(l + 1, t, jc, 1)),
		                                  __float_as_uint(irr_of(l + 1, t, jc, 2)), __float_as_uint(irr_of(l + 1, t, jc, 3)))
		                     : make_uint4(0, 0, 0, 0);
	}
	B[1] = make_uint4(f[0], f[1], f[2], f[3]);
	B[6] = make_uint4(0, 0, 0, 0); B[7] = make_uint4(0, 0, 0, 0);
}

__device__ __forceinline__ int pick(uint4 a, float xi, float &child_irr)
{
	const float i0 = __uint_as_float(a.x), i1 = __uint_as_float(a.y), i2 = __uint_as_float(a.z), i3 = __uint_as_float(a.w);
	const float c1 = i0, c2 = i1 + c1, c3 = i2 + c2, c4 = i3 + c3;
	const float s = xi * c4;
	int k = 3;
	if (s < c1) k = 0;
	else if (s < c2) k = 1;
	else if (s < c3) k = 2;
	child_irr = k == 0 ? i0 : (k == 1 ? i1 : (k == 2 ? i2 : i3));
	return k;
}
__device__ __forceinline__ float draw(uint32_t lane, uint32_t l) { return (float)(hash32(lane * 8u + l + 0x9e3779b9u) >> 8) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(256) void k_walk_a(const uint4 *__restrict__ rec, float *out, uint32_t n)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint32_t r = hash32(i) % T;
	float pdf = 1.0f, node = 1.0f;
	for (uint32_t l = 0; l < L; ++l) {
		const uint4 a = gather16(rec + 2 * (size_t)r), b = gather16(rec + 2 * (size_t)r + 1);
		float ci;
		const int k = pick(a, draw(i, l), ci);
		pdf = pdf * ((4.0f * ci) / node);
		node = ci;
		r = k == 0 ? b.x : (k == 1 ? b.y : (k == 2 ? b.z : b.w));
	}
	out[i] = pdf;
}

__global__ __launch_bounds__(256) void k_walk_b(const uint4 *__restrict__ blk, float *out, uint32_t n)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint32_t b = hash32(i) % T;
	float pdf = 1.0f, node = 1.0f;
	for (uint32_t l = 0; l < L; l += 2) {
		const uint4 *B = blk + 8 * (size_t)b;
		const uint4 a = gather16(B);
		float ci;
		const int k = pick(a, draw(i, l), ci);
		pdf = pdf * ((4.0f * ci) / node);
		node = ci;
		if (l + 1 >= L) break;
		const uint4 a2 = gather16(B + 2 + k), f = gather16(B + 1);
		const int k2 = pick(a2, draw(i, l + 1), ci);
		pdf = pdf * ((4.0f * ci) / node);
		node = ci;
		b = (k == 0 ? f.x : (k == 1 ? f.y : (k == 2 ? f.z : f.w))) + (uint32_t)k2;
	}
	out[i] = pdf;
}

int main()
{
	const uint32_t n = 1u << 22;
	const uint32_t n_rec = base_of(L), n_blk = T + T * 16u + T * 256u;
	uint4 *rec, *blk;
	float *out;
	hipMalloc(&rec, (size_t)n_rec * 32);
	hipMalloc(&blk, (size_t)n_blk * 128);
	hipMalloc(&out, (size_t)n * 4);
	hipLaunchKernelGGL(k_init_a, dim3((n_rec + 255) / 256), dim3(256), 0, 0, rec);
	hipLaunchKernelGGL(k_init_b, dim3((n_blk + 255) / 256), dim3(256), 0, 0, blk);
	hipDeviceSynchronize();
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	float *h = (float *)malloc((size_t)n * 4);
	for (int mode = 0; mode < 2; ++mode) {
		float best = 1e30f;
		for (int rep = 0; rep < 8; ++rep) {
			hipEventRecord(e0);
			if (mode == 0) hipLaunchKernelGGL(k_walk_a, dim3(n / 256), dim3(256), 0, 0, rec, out, n);
			else hipLaunchKernelGGL(k_walk_b, dim3(n / 256), dim3(256), 0, 0, blk, out, n);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			if (ms < best) best = ms;
		}
		hipMemcpy(h, out, (size_t)n * 4, hipMemcpyDeviceToHost);
		double sum = 0;
		for (uint32_t i = 0; i < n; ++i) sum += h[i];
		printf("%s: %.4f ms per launch of 2^22 walks (%u %s, %.1f MB), sum %.9g\n", mode == 0 ? "A records" : "B blocks ", best,
		       mode == 0 ? n_rec : n_blk, mode == 0 ? "records of 32 B" : "blocks of 128 B", mode == 0 ? n_rec * 32e-6 : n_blk * 128e-6, sum);
	}
	return 0;
}
