// Dev probe (GPU box): what does the chip do when every lane asks for its own 16 bytes?  n lanes, lane i gathers entry
// hash(i) mod rows of a table of `rows` 16-byte entries (one global_load_dwordx4 per lane, nothing else but a sum that keeps
// it alive) -- for tables that fit L2 (4 MB), the Infinity Cache (128 MB) and neither (268 MB, 1 GB), and with the lanes of a
// wave spread over the whole table or confined to one 64 KB window of it (a wave that stands in one quadtree's jump table).
// The calibration for the gathered-bytes fraction of pg_pdf / pg_sample on S1: a 16-byte gather costs the memory a whole
// sector, so such a kernel meets THIS rate long before 8 TB/s of useful bytes.
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}

template <int kWindow>
__global__ __launch_bounds__(256) void k_gather(const u32x4 *__restrict__ tbl, uint32_t rows_mask, uint32_t n, uint32_t *__restrict__ out)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint32_t at;
	if (kWindow) { // the wave's lanes inside one window of 4096 entries (64 KB)
		const uint32_t w = mix(i >> 6) & (rows_mask >> 12);
		at = (w << 12) | (mix(i * 2654435761u) & 4095u);
	} else at = mix(i * 2654435761u) & rows_mask;
	const u32x4 v = tbl[at];
	if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) out[i & 255] = v.x; // (never: the table holds zeros)
}

int main()
{
	const uint32_t n = 1u << 24;
	uint32_t *out;
	(void)hipMalloc(&out, 1024);
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	for (uint32_t bits : {18u, 23u, 24u, 26u}) { // 4 MB, 128 MB, 268 MB, 1 GB of 16-byte entries
		u32x4 *tbl;
		const size_t bytes = (size_t)16 << bits;
		(void)hipMalloc(&tbl, bytes);
		(void)hipMemset(tbl, 0, bytes);
		for (int window = 0; window < 2; ++window) {
			float best = 1e30f;
			for (int rep = 0; rep < 6; ++rep) {
				(void)hipEventRecord(a);
				if (window) hipLaunchKernelGGL(k_gather<1>, dim3(n / 256), dim3(256), 0, 0, tbl, (1u << bits) - 1u, n, out);
				else hipLaunchKernelGGL(k_gather<0>, dim3(n / 256), dim3(256), 0, 0, tbl, (1u << bits) - 1u, n, out);
				(void)hipEventRecord(b); (void)hipEventSynchronize(b);
				float ms; (void)hipEventElapsedTime(&ms, a, b);
				if (ms < best) best = ms;
			}
			printf("table %5zu MB, a wave's lanes %s: %.3f ms for 2^24 gathers of 16 B = %.1f G gathers/s = %.2f TB/s of useful bytes\n", bytes >> 20,
			       window ? "inside one 64 KB window" : "over the whole table  ", best, n / best / 1e6, 16.0 * n / best / 1e9);
		}
		(void)hipFree(tbl);
	}
	return 0;
}
