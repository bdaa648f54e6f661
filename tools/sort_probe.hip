// Dev probe: what does a global sort of a bounce's live list cost?  rocPRIM radix sort of (key, index) pairs,
// n = 17 M and 33 M, keys of 12 / 16 / 20 significant bits.
// build: hipcc --offload-arch=gfx950 -O3 tools/sort_probe.hip -o tools/sort_probe.bin
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>
#include <stdio.h>

__global__ void fill(uint32_t *k, uint32_t *v, uint32_t n, uint32_t mask)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t x = i * 2654435761u; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
	k[i] = x & mask; v[i] = i;
}

int main()
{
	const uint32_t nmax = 33u << 20;
	uint32_t *k0, *k1, *v0, *v1;
	(void)hipMalloc(&k0, nmax * 4); (void)hipMalloc(&k1, nmax * 4); (void)hipMalloc(&v0, nmax * 4); (void)hipMalloc(&v1, nmax * 4);
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	for (uint32_t n : {17u << 20, 33u << 20})
		for (int bits : {6, 8, 10, 12, 16}) {
			size_t tmp_bytes = 0;
			(void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
			void *tmp; (void)hipMalloc(&tmp, tmp_bytes);
			float best = 1e30f;
			for (int rep = 0; rep < 5; ++rep) {
				hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, k0, v0, n, (1u << bits) - 1u);
				(void)hipEventRecord(a);
				(void)rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
				(void)hipEventRecord(b); (void)hipEventSynchronize(b);
				float ms; (void)hipEventElapsedTime(&ms, a, b);
				if (ms < best) best = ms;
			}
			printf("n %u bits %d: %.3f ms  (%.1f G pairs/s)  temp %zu MB\n", n, bits, best, n / best / 1e6, tmp_bytes >> 20);
			(void)hipFree(tmp);
		}
	return 0;
}
