/*
 * pg_oracle.h -- TEST INFRASTRUCTURE. CPU restatement ("oracle") of the reference's
 * SD-tree hot path (takkasila/practical_path_guiding_lab: src/kdtree.py,
 * src/quadtree.py, src/common.py, src/path_guiding_integrator.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (practical_path_guiding_lab_amd) never links or calls it.
 *
 * PARITY UNPINNED: the reference holds no golden vectors for this path and cannot be
 * imported here (mitsuba/drjit absent: SURVEY.md 8c), so this oracle is pinned only by
 * the reference-derived invariants and hand-derivable cases in tests/test_oracle_*.py.
 *
 * All arrays are planar SoA: a Vector3f[n] is 3*n floats, plane-major (x[n] y[n] z[n]).
 */
#ifndef PG_ORACLE_H
#define PG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgo_tree pgo_tree; /* one SD-tree = KDTree + its QuadTree forest */

/* Fixed-point contract for irradiance accumulation */
#define PGO_FRAC_BITS 40
#define PGO_W_CLAMP_LOG2 48

pgo_tree *pgo_tree_new(void);                         /* kdtree.py:117-130, quadtree.py:350-362 */
void pgo_tree_free(pgo_tree *t);
/* path_guiding_integrator.py:77-105 (the SD-tree part of setup) */
void pgo_tree_setup(pgo_tree *t, const float bbox_min[3], const float bbox_max[3],
                    int kd_max_depth, int quad_max_depth, int store_nee);
void pgo_tree_copy_from(pgo_tree *dst, const pgo_tree *src); /* kdtree.py:141-153 */

/* kdtree.py:435-470 */
void pgo_get_leaf_node_index(const pgo_tree *t, size_t n, const float *p,
                             const uint8_t *active, uint32_t *node_out);
/* kdtree.py:473-486 (-> quadtree.py:931-998, 1001-1101) */
void pgo_sample(const pgo_tree *t, size_t n, const float *p, uint64_t *rng_state,
                uint64_t *rng_inc, const uint8_t *active, float *dir_out, float *pdf_out);
/* kdtree.py:489-496 */
void pgo_pdf(const pgo_tree *t, size_t n, const float *p, const float *dir,
             const uint8_t *active, float *pdf_out);
/* quadtree.py:931-998 / 1001-1101 on explicit tree ids (rootIndex) */
void pgo_sample_quadtree(const pgo_tree *t, size_t n, const uint32_t *root_index,
                         uint64_t *rng_state, uint64_t *rng_inc, const uint8_t *active,
                         float *dir_out);
void pgo_pdf_quadtree(const pgo_tree *t, size_t n, const uint32_t *root_index,
                      const float *dir, const uint8_t *active, float *pdf_out);

/* kdtree.py:180-225 -> quadtree.py:389-464.  radiance_nee_lum = mi.luminance(radiance_nee). */
void pgo_add_data_propagate(pgo_tree *t, size_t m, const float *position,
                            const float *direction, const float *radiance,
                            const float *wo_pdf, const float *direction_nee,
                            const float *radiance_nee_lum);

/* path_guiding_integrator.py:434-500: processPathData + scatterDataIntoSDTree filter.
 * Inputs are the dense numRays*max_depth record columns; output is the compacted
 * (position, direction, radiance, woPdf, direction_nee, radiance_nee_lum) stream.
 * Returns the number of records kept. */
size_t pgo_process_records(size_t num_rays, size_t max_depth, const float *Lfinal,
                           const uint8_t *rec_active, const float *rec_position,
                           const float *rec_direction, const float *rec_bsdf,
                           const float *rec_throughput_bsdf,
                           const float *rec_throughput_radiance,
                           const float *rec_radiance_nee, const float *rec_direction_nee,
                           const float *rec_wo_pdf, float *out_position, float *out_direction,
                           float *out_radiance, float *out_wo_pdf, float *out_direction_nee,
                           float *out_radiance_nee_lum);

/* individual refine steps (kdtree.py:327-358, 503-532; quadtree.py:512-637, 844-851) */
void pgo_finalize_accumulators(pgo_tree *t); /* integer accumulators -> fp32 columns */
void pgo_set_refinement_threshold(pgo_tree *t, int iteration);
void pgo_kd_refine(pgo_tree *t);
void pgo_set_quadtree_refinement_threshold(pgo_tree *t);
void pgo_refine_all_quadtree(pgo_tree *t);
void pgo_clean_unused_quadtree(pgo_tree *t);
void pgo_reset(pgo_tree *t); /* resetTreeVertCount + resetAllQuadTreeIrradiance */
/* path_guiding_integrator.py:566-586 */
void pgo_refine_and_prepare(pgo_tree *current, pgo_tree *prev, int iteration);

/* forced splits, as the reference self-tests do (kdtree.py:708-712, quadtree.py:1143-1152) */
void pgo_kd_split(pgo_tree *t, size_t n, const uint32_t *node_idx);      /* kdtree.py:229-323 */
void pgo_quad_split(pgo_tree *t, size_t n, const uint32_t *node_idx);    /* quadtree.py:96-191 */
void pgo_kd_all_leaves(const pgo_tree *t, uint32_t *out, size_t *n_out); /* kdtree.py:173-177 */
void pgo_quad_all_leaves(const pgo_tree *t, uint32_t *out, size_t *n_out); /* quadtree.py:288-345, all roots */

/* --- column access for export / comparison (SURVEY Appendix B order) --- */
size_t pgo_kd_size(const pgo_tree *t);
size_t pgo_quad_size(const pgo_tree *t);
size_t pgo_quad_roots(const pgo_tree *t);
double pgo_kd_max_leaf_size(const pgo_tree *t);
int pgo_kd_max_depth(const pgo_tree *t);
int pgo_quad_max_depth(const pgo_tree *t);
int pgo_quad_store_nee(const pgo_tree *t);
/* name in {bbox_min(3n f32), bbox_max, depth(u32), vertCount(f32), isLeaf(u8),
 *          quadTreeRootIndex, child_left_index, child_right_index, count(u64)} */
const void *pgo_kd_column(const pgo_tree *t, const char *name);
/* name in {rootNodeIndex, bbox_min(2n f32), bbox_max, depth, irradiance, isLeaf,
 *          refinementThreshold, child_1_index..child_4_index, acc_lo(u64), acc_hi(i64)} */
const void *pgo_quad_column(const pgo_tree *t, const char *name);

/* load a tree from reference-schema columns (kdtree.py:53-63,156-170; quadtree.py:58-71) */
void pgo_tree_load(pgo_tree *t, size_t n_kd, const float *kd_bbox_min, const float *kd_bbox_max,
                   const uint32_t *kd_depth, const float *kd_vert_count, const uint8_t *kd_is_leaf,
                   const uint32_t *kd_qroot, const uint32_t *kd_left, const uint32_t *kd_right,
                   double kd_max_leaf_size, int kd_max_depth, size_t n_roots,
                   const uint32_t *q_root_node_index, size_t n_q, const float *q_bbox_min,
                   const float *q_bbox_max, const uint32_t *q_depth, const float *q_irradiance,
                   const uint8_t *q_is_leaf, const float *q_threshold, const uint32_t *q_c1,
                   const uint32_t *q_c2, const uint32_t *q_c3, const uint32_t *q_c4,
                   int q_max_depth, int q_store_nee);

/* --- scalar helpers exported for known-answer tests --- */
void pgo_canonical_to_dir_v(size_t n, const float *p2, float *d3);  /* common.py:100-129 */
void pgo_dir_to_canonical_v(size_t n, const float *d3, float *p2);  /* common.py:132-158 */
void pgo_rng_seed(size_t n, uint32_t seed, uint32_t lane0, uint64_t *state, uint64_t *inc);
void pgo_rng_next_f32(size_t n, uint64_t *state, const uint64_t *inc, float *out);
void pgo_quantize_v(size_t n, const float *w, uint64_t *lo, int64_t *hi);
void pgo_acc_to_float_v(size_t n, const uint64_t *lo, const int64_t *hi, float *out);
void pgo_sincos_v(size_t n, const float *phi, float *s, float *c);
void pgo_atan2_v(size_t n, const float *y, const float *x, float *out);
void pgo_math1_v(size_t n, int which, const float *x, float *out); /* 0 exp, 1 log, 2 erf, 3 erfinv (pgo_math.h) */

#ifdef __cplusplus
}
#endif
#endif
