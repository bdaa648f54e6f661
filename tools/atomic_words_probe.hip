// Dev probe: rate of scattered 64-bit atomic adds on gfx950 as a function of how many adjacent words
// of one 32-byte accumulator a record touches (1, 2 or 4, issued from adjacent lanes of one
// wave-instruction), and of the returning form.  n records, random slots of 32 bytes.
// build: hipcc --offload-arch=gfx950 -O3 tools/atomic_words_probe.hip -o tools/atomic_words_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}

// W words per record from W adjacent lanes; 64/W records per wave-instruction, W rounds per wave
template <int W, bool kReturn>
__global__ void k_add(unsigned long long *buf, uint32_t nslots, uint32_t n, unsigned long long *sink)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t lane = threadIdx.x & 63, base = i - lane;
	unsigned long long acc = 0;
	for (int round = 0; round < W; ++round) {
		const uint32_t rec = base + round * (64 / W) + lane / W;
		const uint32_t s = hash32(rec) % nslots;
		unsigned long long *p = buf + 4ull * s + (lane % W);
		if (kReturn) acc += atomicAdd(p, 3ull);
		else atomicAdd(p, 3ull);
	}
	if (kReturn && acc == 0xdeadbeefdeadbeefull) *sink = acc;
}

template <int W, bool kReturn>
static float run(unsigned long long *buf, uint32_t nslots, uint32_t n, unsigned long long *sink)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	float ms = 0;
	for (int rep = 0; rep < 2; ++rep) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL((k_add<W, kReturn>), dim3(n / 256), dim3(256), 0, 0, buf, nslots, n, sink);
		(void)hipEventRecord(e1);
		(void)hipEventSynchronize(e1);
		(void)hipEventElapsedTime(&ms, e0, e1);
	}
	return ms;
}

int main()
{
	const uint32_t n = 1u << 24;
	for (uint32_t nslots : {1u << 20, 1u << 22}) {
		unsigned long long *buf, *sink;
		(void)hipMalloc(&buf, 32ull * nslots);
		(void)hipMalloc(&sink, 8);
		(void)hipMemset(buf, 0, 32ull * nslots);
		const float a1 = run<1, false>(buf, nslots, n, sink), a2 = run<2, false>(buf, nslots, n, sink);
		const float a4 = run<4, false>(buf, nslots, n, sink), r1 = run<1, true>(buf, nslots, n, sink);
		const float r2 = run<2, true>(buf, nslots, n, sink);
		printf("slots %u (%.0f MB): records/s  1 word %.1f G  2 words %.1f G  4 words %.1f G | returning: 1 word %.1f G  2 words %.1f G\n",
		       nslots, 32.0 * nslots / 1e6, n / a1 / 1e6, n / a2 / 1e6, n / a4 / 1e6, n / r1 / 1e6, n / r2 / 1e6);
		(void)hipFree(buf);
		(void)hipFree(sink);
	}
	return 0;
}
