// Dev probe: does gfx950 merge same-line 64-bit atomics issued by different lanes of ONE instruction?
//   A: every lane adds to word 0 and word 1 of its own random 16-B slot (two instructions)
//   B: lane pairs (2j, 2j+1) add to word 0 / word 1 of record j's slot in one instruction, two rounds
//   C: one returning atomic on word 0 only (cost of a returning form)
//   D: one non-returning atomic on word 0 only
// build: hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip -o /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}

__global__ void kA(unsigned long long *buf, uint32_t nslots, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t s = hash32(i) % nslots;
	atomicAdd(buf + 2ull * s, 3ull);
	atomicAdd(buf + 2ull * s + 1, 5ull);
}

__global__ void kB(unsigned long long *buf, uint32_t nslots, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t lane = threadIdx.x & 63, base = i - lane;
	for (int round = 0; round < 2; ++round) {
		const uint32_t rec = base + round * 32 + (lane >> 1);
		const uint32_t s = hash32(rec) % nslots;
		atomicAdd(buf + 2ull * s + (lane & 1), (lane & 1) ? 5ull : 3ull);
	}
}

__global__ void kC(unsigned long long *buf, uint32_t nslots, uint32_t n, unsigned long long *sink)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t s = hash32(i) % nslots;
	const unsigned long long old = atomicAdd(buf + 2ull * s, 3ull);
	if (old == 0xdeadbeefdeadbeefull) *sink = old;
}

__global__ void kD(unsigned long long *buf, uint32_t nslots, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t s = hash32(i) % nslots;
	atomicAdd(buf + 2ull * s, 3ull);
}

int main()
{
	const uint32_t n = 1u << 24;
	for (uint32_t nslots : {1u << 12, 1u << 18, 1u << 22}) {
		unsigned long long *buf, *sink;
		hipMalloc(&buf, 16ull * nslots);
		hipMalloc(&sink, 8);
		hipMemset(buf, 0, 16ull * nslots);
		hipEvent_t e0, e1;
		hipEventCreate(&e0);
		hipEventCreate(&e1);
		const dim3 g(n / 256), b(256);
		float ms[4];
		for (int k = 0; k < 4; ++k) {
			for (int rep = 0; rep < 2; ++rep) { // first rep warms up
				hipEventRecord(e0);
				if (k == 0) hipLaunchKernelGGL(kA, g, b, 0, 0, buf, nslots, n);
				if (k == 1) hipLaunchKernelGGL(kB, g, b, 0, 0, buf, nslots, n);
				if (k == 2) hipLaunchKernelGGL(kC, g, b, 0, 0, buf, nslots, n, sink);
				if (k == 3) hipLaunchKernelGGL(kD, g, b, 0, 0, buf, nslots, n);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				hipEventElapsedTime(&ms[k], e0, e1);
			}
		}
		printf("slots %8u (%.1f MB): A two-instr %.3f ms (%.1f Grec/s)  B paired %.3f ms (%.1f Grec/s)  "
		       "C returning x1 %.3f ms (%.1f G/s)  D plain x1 %.3f ms (%.1f G/s)\n",
		       nslots, 16.0 * nslots / 1e6, ms[0], n / ms[0] / 1e6, ms[1], n / ms[1] / 1e6, ms[2], n / ms[2] / 1e6, ms[3],
		       n / ms[3] / 1e6);
		hipFree(buf);
		hipFree(sink);
	}
	return 0;
}
