// Dev probe (GPU box): the library's own sort of (16-bit key, place) pairs (csrc/pg_sort.hip, included as it is) checked and
// timed beside rocPRIM's radix_sort_pairs on the same keys.
//   checks: places_out[0 .. live) is a permutation of 0 .. live-1 (live = 8/9 n, read from device memory); the keys along it ascend in their HIGH byte; inside one high byte the low
//           bytes descend nowhere by more than one tile of the second pass can mix (reported: pairs out of low-byte order)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I practical_path_guiding_lab_amd/csrc tools/sort_check.hip -o tools/sort_check.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "pg_sort.hip"

// mode 0: uniform keys; 1: every live key the same; 2: two cells and the "left the scene" key; 3: keys ascending with the place
__global__ void fill(uint16_t *k, uint32_t n, uint32_t live, int mode)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t x = i * 2654435761u; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15;
	uint32_t key = x & 0xffffu;
	if (mode == 1) key = 0x1234u;
	if (mode == 2) key = (x & 1u) ? 0x0101u : ((x & 2u) ? 0x0100u : 0xfffeu);
	if (mode == 3) key = (uint32_t)(((uint64_t)i << 16) / n);
	if (key >= 0xfffdu) key = 0xfffdu;
	if ((x >> 20) % 97u == 0u) key = 0xfffeu; // a ray that left the scene
	k[i] = i < live ? (uint16_t)key : (uint16_t)0xffffu; // places without a path
}

int main()
{
	const uint32_t nmax = 33u << 20;
	uint16_t *k0, *k1;
	uint32_t *perm, *perm2, *d_live;
	(void)hipMalloc(&d_live, 4);
	(void)hipMalloc(&k0, nmax * 2 + 64); (void)hipMalloc(&k1, nmax * 2 + 64); (void)hipMalloc(&perm, nmax * 4); (void)hipMalloc(&perm2, nmax * 4);
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	int bad = 0;
	for (int mode = 0; mode < 4; ++mode)
	for (uint32_t n : {1u, 255u, 4096u, 4097u, 1000003u, 8294400u + 324731u, 17u << 20, 33u << 20}) {
		if (mode && n < 4096u) continue;
		const uint32_t live = n - n / 9;
		(void)hipMemcpy(d_live, &live, 4, hipMemcpyHostToDevice);
		const size_t tb = pg::sort_pairs_temp_bytes(n);
		void *tmp; (void)hipMalloc(&tmp, tb);
		size_t rb = 0;
		(void)rocprim::radix_sort_pairs(nullptr, rb, (const uint16_t *)k0, k1, rocprim::counting_iterator<uint32_t>(0), perm2, n, 0, 16, 0);
		void *rtmp; (void)hipMalloc(&rtmp, rb);
		float best = 1e30f, best_r = 1e30f;
		for (int rep = 0; rep < 5; ++rep) {
			hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, k0, n, live, mode);
			(void)hipMemsetAsync(perm, 0xff, (size_t)n * 4, 0);
			(void)hipEventRecord(a);
			if (pg::sort_places16(tmp, tb, k0, k1, perm, n, d_live, 0) != hipSuccess) { printf("sort_places16 failed\n"); return 1; }
			(void)hipEventRecord(b); (void)hipEventSynchronize(b);
			float ms; (void)hipEventElapsedTime(&ms, a, b);
			if (ms < best) best = ms;
			(void)hipEventRecord(a);
			(void)rocprim::radix_sort_pairs(rtmp, rb, (const uint16_t *)k0, k1, rocprim::counting_iterator<uint32_t>(0), perm2, n, 0, 16, 0);
			(void)hipEventRecord(b); (void)hipEventSynchronize(b);
			(void)hipEventElapsedTime(&ms, a, b);
			if (ms < best_r) best_r = ms;
		}
		// check the library's result (the keys are still in k0; k1 was scratch of both)
		hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, k0, n, live, mode);
		(void)pg::sort_places16(tmp, tb, k0, k1, perm, n, d_live, 0);
		std::vector<uint16_t> hk(n);
		std::vector<uint32_t> hp(n);
		(void)hipMemcpy(hk.data(), k0, (size_t)n * 2, hipMemcpyDeviceToHost);
		(void)hipMemcpy(hp.data(), perm, (size_t)n * 4, hipMemcpyDeviceToHost);
		std::vector<uint8_t> seen(n, 0);
		uint64_t dup = 0, high_desc = 0, low_desc = 0, low_desc_far = 0;
		for (uint32_t i = 0; i < live; ++i) { // (the live prefix: a permutation of the live places)
			if (hp[i] >= live || seen[hp[i]]) { ++dup; continue; }
			seen[hp[i]] = 1;
			if (i) {
				if (hp[i - 1] >= live) continue;
				const uint32_t ka = hk[hp[i - 1]], kb = hk[hp[i]];
				if ((kb >> 8) < (ka >> 8)) ++high_desc;
				else if ((kb >> 8) == (ka >> 8) && (kb & 255u) < (ka & 255u)) { ++low_desc; if ((ka & 255u) - (kb & 255u) > 1u) ++low_desc_far; }
			}
		}
		const bool ok = dup == 0 && high_desc == 0 && (low_desc_far == 0 || n < (1u << 21) || mode != 0); // (a short list: one tile of the second pass spans many low bytes)
		if (!ok) ++bad;
		printf("keys %d n %9u: pg %.3f ms (%.1f G pairs/s)  rocPRIM %.3f ms (%.1f)  | permutation %s, high byte descends %llu x, low byte descends %llu x (by more than one step: %llu)%s\n",
		       mode, n, best, n / best / 1e6, best_r, n / best_r / 1e6, dup ? "BROKEN" : "ok", (unsigned long long)high_desc, (unsigned long long)low_desc,
		       (unsigned long long)low_desc_far, ok ? "" : "  <-- FAIL");
		(void)hipFree(tmp); (void)hipFree(rtmp);
	}
	printf(bad ? "sort_check FAILED\n" : "sort_check ok\n");
	return bad ? 1 : 0;
}
