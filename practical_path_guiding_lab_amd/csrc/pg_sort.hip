// pg_sort.hip -- a global order for the live list of a bounce (pg_render_sort, pg_render_wave.hip): the places of the
// list sorted by a 16-bit spatial key (the Morton cell of the vertex the ray has just found), so that the lanes of a
// wave stand near each other in the scene -- in one KD leaf, under one quadtree, in front of the same BVH nodes --
// for every kernel that follows.  Not in the reference (Dr.Jit's wavefront keeps pixel order); a lane's result
// depends on its own state only, so the order is free.
//
// The sort itself is rocPRIM's device radix sort of (key, place) pairs (header-only, part of ROCm): 0.59 ms for
// 33 M pairs of 16-bit keys on MI355X (tools/sort_probe.hip).
#include <cstring>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "pg_context.hpp"

namespace pg {

// temporary storage rocPRIM asks for when sorting n pairs
size_t sort_pairs_temp_bytes(uint32_t n)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
	                                (uint32_t *)nullptr, n, 0, 16, (hipStream_t) nullptr);
	return bytes;
}

// values_out[k] = the place with the k-th smallest key (bits 0-15); stable; asynchronous on `s`
hipError_t sort_pairs16(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *values_in,
                        uint32_t *values_out, uint32_t n, hipStream_t s)
{
	return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, values_in, values_out, n, 0, 16, s);
}

} // namespace pg
