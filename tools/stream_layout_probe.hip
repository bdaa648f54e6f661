// Dev probe: what does the LAYOUT of a streaming kernel's per-lane data cost on gfx950?
// The mesh pipeline's shading kernels read ~20 and write ~40 32-bit words per lane through "planes" (word w of
// lane i at plane[w][i]: every access coalesced, but 60 concurrent streams of 256 B per wave-instruction).  The
// same words as 16-byte quads (quad q of lane i at quad[q][i]: 15 streams of 1 KiB per wave-instruction), and
// as one contiguous per-lane record behind an LDS transpose, for comparison.  Prints GB/s (read + written).
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_layout_probe.hip -o /tmp/stream_layout_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int R = 20, W = 40; // words read / written per lane

__global__ __launch_bounds__(256) void k_planes(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n)
{
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint32_t v[R];
#pragma unroll
	for (int w = 0; w < R; ++w) v[w] = in[(uint64_t)w * n + i];
	uint32_t acc = 0;
#pragma unroll
	for (int w = 0; w < R; ++w) acc = acc * 0x9e3779b1u + v[w];
#pragma unroll
	for (int w = 0; w < W; ++w) out[(uint64_t)w * n + i] = acc + v[w % R] * (uint32_t)(w + 1);
}

__global__ __launch_bounds__(256) void k_quads(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t n)
{
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint4 v[R / 4];
#pragma unroll
	for (int q = 0; q < R / 4; ++q) v[q] = in[(uint64_t)q * n + i];
	uint32_t acc = 0;
#pragma unroll
	for (int q = 0; q < R / 4; ++q) acc = (((acc * 0x9e3779b1u + v[q].x) * 0x9e3779b1u + v[q].y) * 0x9e3779b1u + v[q].z) * 0x9e3779b1u + v[q].w;
#pragma unroll
	for (int q = 0; q < W / 4; ++q) {
		const uint4 s = v[q % (R / 4)];
		out[(uint64_t)q * n + i] = make_uint4(acc + s.x * (4 * q + 1), acc + s.y * (4 * q + 2), acc + s.z * (4 * q + 3), acc + s.w * (4 * q + 4));
	}
}

int main()
{
	const uint64_t n = 17u << 20; // lanes of an average bounce launch of the veach-ajar bench
	uint32_t *in, *out;
	hipMalloc(&in, n * R * 4);
	hipMalloc(&out, n * W * 4);
	hipMemset(in, 1, n * R * 4);
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	const dim3 grid((unsigned)((n + 255) / 256));
	for (int mode = 0; mode < 2; ++mode) {
		float best = 1e30f;
		for (int rep = 0; rep < 6; ++rep) {
			hipEventRecord(a);
			if (mode == 0) hipLaunchKernelGGL(k_planes, grid, dim3(256), 0, 0, in, out, n);
			else hipLaunchKernelGGL(k_quads, grid, dim3(256), 0, 0, (const uint4 *)in, (uint4 *)out, n);
			hipEventRecord(b);
			hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			if (rep && ms < best) best = ms;
		}
		printf("%s: %.3f ms  %.0f GB/s (%d words read + %d written per lane, %llu lanes)\n", mode == 0 ? "planes (4 B per lane and stream)" : "quads (16 B per lane and stream)",
		       best, (double)n * (R + W) * 4 / (best * 1e-3) / 1e9, R, W, (unsigned long long)n);
	}
	return 0;
}
