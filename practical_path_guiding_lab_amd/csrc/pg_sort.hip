// pg_sort.hip -- a global order for the live list of a bounce (pg_render_sort, pg_render_wave.hip): the places of the
// list sorted by a 16-bit spatial key (the Morton cell of the vertex the ray has just found), so that the lanes of a
// wave stand near each other in the scene -- in one KD leaf, under one quadtree, in front of the same BVH nodes --
// for every kernel that follows.  Not in the reference (Dr.Jit's wavefront keeps pixel order); a lane's result
// depends on its own state only, so the order is free.
//
// The sort itself is rocPRIM's device radix sort of (key, place) pairs (header-only, part of ROCm): two onesweep passes of
// eight bits.  Round 4 feeds it what it needs and no more -- the keys as 16-bit words (they were 32-bit words with 16 empty
// bits: 4 bytes read and written per pair and pass for nothing) and the places as a counting iterator instead of an iota buffer
// read from memory: 36 -> 22 bytes per pair over the two passes.
#include <cstring>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "pg_context.hpp"

namespace pg {

// temporary storage rocPRIM asks for when sorting n pairs
size_t sort_pairs_temp_bytes(uint32_t n)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint16_t *)nullptr, (uint16_t *)nullptr, rocprim::counting_iterator<uint32_t>(0),
	                                (uint32_t *)nullptr, n, 0, 16, (hipStream_t) nullptr);
	return bytes;
}

// places_out[k] = the place (0 .. n-1) with the k-th smallest key; stable; asynchronous on `s`
hipError_t sort_places16(void *temp, size_t temp_bytes, const uint16_t *keys_in, uint16_t *keys_out, uint32_t *places_out, uint32_t n,
                         hipStream_t s)
{
	return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rocprim::counting_iterator<uint32_t>(0), places_out, n, 0, 16, s);
}

} // namespace pg
