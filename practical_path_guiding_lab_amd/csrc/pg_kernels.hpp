// pg_kernels.hpp -- host-callable launchers of the HIP kernels (definitions in pg_kernels_*.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pgsd.h"
#include "pg_tree.hpp"

namespace pg {

// Probe builds (-DPG_SHADE_PHASES=1): where does a wave of k_wave_shade spend its life?  Wave-clock cycles (s_memtime), phase by phase
// -- 0 the records + staging, 1 stage_a1, 2 the shadow walk, 3 stage_a2, 4 the SD-tree calls, 5 stage_b, 6 the survivors' append; word 7
// counts the waves -- accumulated by each wave in LDS and flushed ONCE, by one eight-lane atomic, into one of kPhaseStripes lines of
// kPhaseWords words (pg_read_shade_phases sums the stripes).  Round 5's probe added every stamp to ONE line of device memory: its
// k_wave_shade ran eleven times slower than the product's, and the shares it reported were those of a kernel waiting for that line.
constexpr int kPhaseStripes = 1024, kPhaseWords = 16;

struct DepthCounters { // device-resident, optional
	unsigned long long kd_levels, kd_queries, quad_levels, quad_queries;
	unsigned long long layout_bytes; // bytes the lanes gathered from the built tables (stat_word, pg_descent.hpp); tree heads not included
	// (diagnostics of instrumented passes, printed by pg_read_depth_counters under $PGSD_TRACE_SHADOW: how full are the waves
	// that walk shadow rays in k_wave_shade?)  waves that walked, their lanes with a ray, waves that ran the kernel's body
	unsigned long long shadow_waves, shadow_lanes, body_waves;
};

// ---- queries (pg_kernels_query.hip) ----
void launch_leaf_index(const TreeView &t, uint64_t n, const float *p, const uint8_t *active,
                       uint32_t *node_out, hipStream_t s);
void launch_sample(const TreeView &t, uint64_t n, const float *p, uint64_t *rng_state,
                   const uint64_t *rng_inc, const uint8_t *active, float *dir_out, float *pdf_out,
                   DepthCounters *dc, hipStream_t s);
void launch_pdf(const TreeView &t, uint64_t n, const float *p, const float *dir,
                const uint8_t *active, float *pdf_out, DepthCounters *dc, hipStream_t s);
void launch_guide_bounce(const TreeView &t, uint64_t n, const float *p, const float *dir_nee,
                         const uint8_t *nee_active, const uint8_t *select, float *dir_io,
                         uint64_t *rng_state, const uint64_t *rng_inc, float *pdf_nee_out,
                         float *pdf_out, const uint32_t *lane_index, const uint32_t *d_lane_count,
                         DepthCounters *dc, hipStream_t s);
void launch_compact_lanes(uint64_t n, const uint8_t *select, const uint8_t *nee_active, uint32_t *idx_out,
                          uint32_t *d_count, hipStream_t s);
void launch_rng_seed(uint64_t n, uint32_t seed, uint32_t lane0, uint64_t *state, uint64_t *inc,
                     hipStream_t s);

// (re)builds the quadtree jump tables of `t`, 4^bits entries per tree, into out[t.n_trees << (2 * bits)] (t.jump is not read)
bool launch_build_jump(const TreeView &t, QuadJump *out, int bits, hipStream_t s);
// (re)builds the KD jump grid of `t` into out[8^t.grid_bits + kKdGridRootEntries] (t.kd_grid is not read; t.grid_bits, kd_planes and grid_inv are)
void launch_build_kd_grid(const TreeView &t, KdGridEntry *out, hipStream_t s);

// ---- recording (pg_kernels_splat.hip) ----
void launch_splat(const TreeView &t, const AccumView &a, int store_nee, uint64_t m,
                  const pg_records &rec, const uint32_t *d_count, DepthCounters *dc, hipStream_t s);
void launch_process_records(uint64_t num_rays, int32_t max_depth, const float *l_final,
                            const pg_dense_records &rec, const pg_records_out &out,
                            uint32_t *d_count, hipStream_t s);
// The record list of the split render pipeline (pg_render_wave.hip): one entry per live path and bounce in visiting
// order, planes of stride num_rays * max_depth.  An entry holds what processPathData (path_guiding_integrator.py:
// 434-453) needs of the vertex and -- instead of its position and directions -- the accumulators of sdTree_current
// they lead to, found by the walks of sdTree_prev the bounce made anyway (same topology, :582).
struct pg_list_records {
	const uint32_t *ray_of;    // the path (lane) of the entry, 0xffffffff: the path left the scene there
	const float *bsdf, *throughput_bsdf, *throughput_radiance; // 3 planes each (:331-341)
	const float *nee_lum;      // luminance of the NaN-scrubbed radiance_nee (:336, 467, 471)
	const float *wo_pdf;
	const uint2 *slot;         // {path direction's, emitter direction's} accumulator: kSlotNone / kSlotRoot / rec * 4 + child
	const uint32_t *tree;      // quadtree of the vertex's KD leaf; bit 31: inside the root box (counted, kdtree.py:193)
};
// l_final (3, num_rays) planar, or -- when l_final_q is given -- one 16-byte entry per path
void launch_splat_list(const TreeView &t, const AccumView &a, int store_nee, uint64_t num_rays, int32_t max_depth,
                       const float *l_final, const uint4 *l_final_q, const pg_list_records &rec, const uint32_t *live_count,
                       hipStream_t s);

void launch_process_and_splat(const TreeView &t, const AccumView &a, int store_nee,
                              uint64_t num_rays, int32_t max_depth, const float *l_final,
                              const pg_dense_records &rec, DepthCounters *dc, hipStream_t s);

} // namespace pg
