// Dev probe: where do 64-bit integer atomics execute on gfx950, and at what rate, as a function of
// the memory scope?  Agent-scope atomics run memory-side (MI355X_MICROARCH.md, Global float atomics).
// If workgroup-scope ones run in the XCD's L2, a per-XCD private accumulator copy (indexed by
// HW_REG_XCC_ID) could take the splat's scattered adds at L2 speed, merged once afterwards.
//   A: agent scope, one shared buffer            (what the library does today)
//   W: workgroup scope, buffer copy of this XCD  (candidate)
//   S: workgroup scope, one shared buffer        (rate only; results may be wrong across XCDs)
// Every kernel adds 4 adjacent words (one 32-B sector) from 4 adjacent lanes, as coop_add does.
// Correctness: the sum of all words must be n * (1+2+3+4).
// build: hipcc --offload-arch=gfx950 -O3 tools/atomic_scope_probe.hip -o /tmp/atomic_scope_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}

__device__ __forceinline__ uint32_t xcc_id()
{
	// s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
	return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
}

template <int kMode>
__global__ void k_add(unsigned long long *buf, uint32_t nslots, uint32_t n, uint32_t *xcc_hist)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t lane = threadIdx.x & 63, base = i - lane;
	const uint32_t x = xcc_id();
	if (threadIdx.x == 0 && xcc_hist) atomicAdd(xcc_hist + x, 1u);
	unsigned long long *b = kMode == 1 ? buf + (size_t)x * 4ull * nslots : buf;
	for (int round = 0; round < 4; ++round) {
		const uint32_t rec = base + round * 16 + (lane >> 2);
		const uint32_t s = hash32(rec) % nslots;
		unsigned long long *p = b + 4ull * s + (lane & 3);
		const unsigned long long v = (lane & 3) + 1;
		if (kMode == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}

int main()
{
	const uint32_t n = 1u << 24;
	for (uint32_t nslots : {1u << 14, 1u << 20, 1u << 22}) {
		const size_t words = 4ull * nslots * 8; // room for 8 XCD copies
		unsigned long long *buf;
		uint32_t *hist;
		hipMalloc(&buf, words * 8);
		hipMalloc(&hist, 8 * 4);
		hipEvent_t e0, e1;
		hipEventCreate(&e0);
		hipEventCreate(&e1);
		const dim3 g(n / 256), b(256);
		const char *name[3] = {"A agent/shared", "W workgroup/per-XCD", "S workgroup/shared"};
		printf("slots %u (%.1f MB per copy)\n", nslots, 32.0 * nslots / 1e6);
		for (int k = 0; k < 3; ++k) {
			float ms = 0;
			unsigned long long total = 0;
			for (int rep = 0; rep < 2; ++rep) {
				hipMemset(buf, 0, words * 8);
				hipMemset(hist, 0, 32);
				hipDeviceSynchronize();
				hipEventRecord(e0);
				if (k == 0) hipLaunchKernelGGL(k_add<0>, g, b, 0, 0, buf, nslots, n, hist);
				if (k == 1) hipLaunchKernelGGL(k_add<1>, g, b, 0, 0, buf, nslots, n, hist);
				if (k == 2) hipLaunchKernelGGL(k_add<2>, g, b, 0, 0, buf, nslots, n, hist);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				hipEventElapsedTime(&ms, e0, e1);
			}
			std::vector<unsigned long long> h(words);
			hipMemcpy(h.data(), buf, words * 8, hipMemcpyDeviceToHost);
			for (size_t w = 0; w < words; ++w) total += h[w];
			uint32_t hh[8];
			hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost);
			printf("  %-22s %.3f ms  %.1f G sector-updates/s  sum %s (%llu vs %llu)  blocks per XCC: %u %u %u %u %u %u %u %u\n",
			       name[k], ms, n / ms / 1e6, total == 10ull * n ? "OK" : "WRONG", total, 10ull * n, hh[0], hh[1], hh[2], hh[3],
			       hh[4], hh[5], hh[6], hh[7]);
		}
		hipFree(buf);
		hipFree(hist);
	}
	return 0;
}
