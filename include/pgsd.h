/*
 * pgsd.h -- C ABI of libpgsd.so, the MI355X-native SD-tree ("spatial-directional tree")
 * path-guiding library.  This is the drop-in boundary for the hot path of
 * takkasila/practical_path_guiding_lab: everything the reference's Python integrator does
 * with its two KDTree objects (src/path_guiding_integrator.py -> src/kdtree.py ->
 * src/quadtree.py) is reachable through these entry points.  Citations below are
 * file:line in that repository.
 *
 * Conventions
 *  - Plain C, no exceptions cross the boundary.  Every call returns PG_OK (0) or a negative
 *    pg_status; pg_last_error() returns a human-readable message for the last failure.
 *  - All bulk pointers are DEVICE pointers owned by the caller unless a parameter is named
 *    `h_*` (host).  The library owns the trees.  `stream` is a hipStream_t (NULL = default
 *    stream); calls are asynchronous on it unless stated otherwise.
 *  - Vector arrays are planar SoA: a Vector3f[n] is 3*n floats, plane-major
 *    (x[0..n) y[0..n) z[0..n)); a Vector2f[n] likewise with 2 planes.  This is how Dr.Jit
 *    stores mi.Vector3f and gives coalesced per-lane loads.
 *  - Masks are one byte per lane (0 = inactive); a NULL mask means "all active".
 *  - Arithmetic contract: DESIGN.md section 4 (fp32, no contraction; fixed-point accumulation).
 */
#ifndef PGSD_H
#define PGSD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 3): pg_pass_params gained `slot`, pg_kernel_timing gained shade_a_ms / shade_b_ms / sort_ms, pg_scene_desc's
 * BVH is limited to 2^25 nodes; new entry points pg_film_stripes, pg_render_overlap, pg_render_sort, pg_render_stages.
 * A caller compiled against version 2 must be rebuilt (the two structs changed size): check pg_abi_version(). */
/* 4 (round 4): pg_depth_counters gained layout_bytes, pg_stats gained bytes_jump_tables / jump_bits / kd_grid_bits,
 * pg_pass_params.reserved2 became `batched` (same size); new entry points pg_film_batched, pg_film_batched_accumulate. */
/* 5 (round 5): new entry points pg_comm_info, pg_exchange_pack, pg_exchange_unpack, pg_exchange_pack_words,
 * pg_exchange_unpack_words; pg_allreduce moves the accumulators in the 24-byte exchange format (same sums).  No struct changed. */
/* 6 (round 6): pg_refine_and_swap is a transaction (on any error both trees are exactly what they were; the context stays usable);
 * new entry points pg_debug_fail_alloc, pg_debug_fail_alloc_pending (fault injection for the tests), pg_read_shade_phases (probe
 * builds); pg_enable_depth_counters takes mode 2.  No struct changed. */
#define PGSD_ABI_VERSION 6

typedef struct pg_context pg_context;

typedef enum pg_status {
	PG_OK = 0,
	PG_ERR_INVALID = -1,    /* bad argument / state                                   */
	PG_ERR_HIP = -2,        /* a HIP runtime call failed (message has the HIP error)   */
	PG_ERR_NOMEM = -3,      /* allocation failed                                       */
	PG_ERR_FORMAT = -4,     /* imported tree violates the layout the kernels rely on   */
	PG_ERR_NO_DEVICE = -5   /* no usable gfx950 device                                 */
} pg_status;

/* Fixed-point accumulation contract: weights are truncated to multiples of
 * 2^-PG_FRAC_BITS after clamping |w| to 2^PG_W_CLAMP_LOG2; each accumulator is three signed
 * 64-bit limbs of 32 payload bits (value = l0 + l1*2^32 + l2*2^64). */
#define PG_FRAC_BITS 40
#define PG_W_CLAMP_LOG2 48
#define PG_ACC_LIMBS 3
/* An accumulator occupies PG_ACC_WORDS int64 in one 32-byte sector: the three limbs and a record
 * counter (records inside the KD bbox whose path direction ended in this quadtree leaf). */
#define PG_ACC_WORDS 4

/* ---- lifetime ------------------------------------------------------------------------- */

/* Creates a context bound to HIP device `device_ordinal`.  Fails with PG_ERR_NO_DEVICE when
 * no GPU is visible: there is no CPU fallback. */
int pg_create(pg_context **out, int device_ordinal);
int pg_destroy(pg_context *ctx);
/* ctx may be NULL: returns the message of the last failed pg_create on this thread. */
const char *pg_last_error(const pg_context *ctx);
int pg_abi_version(void);

/* PathGuidingIntegrator.setup (path_guiding_integrator.py:77-105): both SD-trees become a
 * single-leaf KD tree over [bbox_min,bbox_max] owning a single-leaf quadtree
 * (kdtree.py:117-138, quadtree.py:350-362).  num_rays / max_depth size the dense record
 * buffer of pg_process_records (path_guiding_integrator.py:93). */
int pg_setup(pg_context *ctx, const float h_bbox_min[3], const float h_bbox_max[3],
             uint64_t num_rays, int32_t max_depth, int32_t kd_max_depth, int32_t quad_max_depth,
             int32_t store_nee, float bsdf_sampling_fraction);

/* PathGuidingIntegrator.setIteration (path_guiding_integrator.py:121-123). */
int pg_set_iteration(pg_context *ctx, int32_t iteration, int32_t is_final);

/* ---- queries on sdTree_prev ----------------------------------------------------------- */

/* KDTree.getLeafNodeIndex (kdtree.py:435-470): node index in the reference's numbering. */
int pg_get_leaf_node_index(pg_context *ctx, uint64_t n, const float *p, const uint8_t *active,
                           uint32_t *node_out, void *stream);

/* KDTree.sample (kdtree.py:473-486): direction sampled from the quadtree of the leaf that
 * contains p, and its pdf.  rng_state/rng_inc are per-lane PCG32 streams, advanced in place
 * by 3 draws per visited quadtree node (quadtree.py:956, 980).  Inactive lanes return
 * dir (0,0,-1), pdf 1 and leave their stream untouched. */
int pg_sample(pg_context *ctx, uint64_t n, const float *p, uint64_t *rng_state,
              const uint64_t *rng_inc, const uint8_t *active, float *dir_out, float *pdf_out,
              void *stream);

/* KDTree.pdf (kdtree.py:489-496). */
int pg_pdf(pg_context *ctx, uint64_t n, const float *p, const float *dir, const uint8_t *active,
           float *pdf_out, void *stream);

/* The three SD-tree calls of one bounce (path_guiding_integrator.py:244, 301, 307) with a
 * single KD descent:
 *   pdf_nee_out[i] = sdTree_prev.pdf(p, dir_nee)             if nee_active[i]   (else 1)
 *   select[i] == 2 : (dir_io, pdf_out) = sdTree_prev.sample(p, rng)   ("sdtree-mis" lanes)
 *   select[i] == 1 : pdf_out = sdTree_prev.pdf(p, dir_io)             ("bsdf-mis" lanes)
 *   select[i] == 0 : pdf_out = 1, dir_io untouched, stream untouched
 * dir_io holds the BSDF-sampled world direction on entry.
 * lane_index / d_lane_count (both NULL, or both set): the compacted list of live ray slots and
 * its two counters as written by pg_compact_lanes; only those slots are processed (and written),
 * and the launch costs waves in proportion to the counters, which are read on the device (no
 * host round trip between bounces). */
int pg_guide_bounce(pg_context *ctx, uint64_t n, const float *p, const float *dir_nee,
                    const uint8_t *nee_active, const uint8_t *select, float *dir_io,
                    uint64_t *rng_state, const uint64_t *rng_inc, float *pdf_nee_out,
                    float *pdf_out, const uint32_t *lane_index, const uint32_t *d_lane_count,
                    void *stream);

/* Active-ray stream compaction between bounces of the wavefront loop (the reference masks dead
 * lanes instead, path_guiding_integrator.py:179, 380).  A lane is live when select[i] != 0 or
 * nee_active[i] != 0 (nee_active may be NULL).  idx_out (uint32[n]) receives the lanes with
 * select == 2 packed from the front, idx_out[0 .. d_count[0]), and the other live lanes packed
 * from the back, idx_out[n-1], idx_out[n-2], ... (d_count[1] of them), so that waves of the
 * bounce kernel do not mix the sampling and the pdf-only code paths.  Order inside each class
 * is unspecified.  d_count is a device uint32[2], zeroed by the call; masks 16-byte aligned. */
int pg_compact_lanes(pg_context *ctx, uint64_t n, const uint8_t *select, const uint8_t *nee_active,
                     uint32_t *idx_out, uint32_t *d_count, void *stream);

/* Mitsuba `independent` sampler seeding for lanes lane0..lane0+n (PCG32 + TEA). */
int pg_rng_seed(pg_context *ctx, uint64_t n, uint32_t seed, uint32_t lane0, uint64_t *rng_state,
                uint64_t *rng_inc, void *stream);

/* ---- recording into sdTree_current ---------------------------------------------------- */

/* Compact record stream = what KDTree.addDataPropagate consumes (kdtree.py:180-225,
 * quadtree.py:389-464) after scatterDataIntoSDTree's gathers
 * (path_guiding_integrator.py:485-497).  radiance_nee_lum = mi.luminance(radiance_nee). */
typedef struct pg_records {
	const float *position;         /* Vector3f[m] planar */
	const float *direction;        /* Vector2f[m] canonical, planar */
	const float *radiance;         /* Float[m] */
	const float *wo_pdf;           /* Float[m] */
	const float *direction_nee;    /* Vector2f[m] canonical, planar (ignored if !store_nee) */
	const float *radiance_nee_lum; /* Float[m]                       (ignored if !store_nee) */
} pg_records;

/* KDTree.addDataPropagate: adds the m records to sdTree_current.  If d_count is non-NULL it
 * points to a device uint32 holding the number of valid records (<= m), e.g. the counter
 * written by pg_process_records; otherwise all m are used. */
int pg_splat(pg_context *ctx, uint64_t m, const pg_records *rec, const uint32_t *d_count,
             void *stream);

/* Dense per-pass record buffer, slot = ray*max_depth + depth
 * (path_guiding_integrator.py:318-346). */
typedef struct pg_dense_records {
	const uint8_t *active;
	const float *position;            /* Vector3f[S] */
	const float *direction;           /* Vector2f[S] canonical */
	const float *bsdf;                /* Color3f[S]  bsdf_weight */
	const float *throughput_bsdf;     /* Color3f[S]  */
	const float *throughput_radiance; /* Color3f[S]  */
	const float *radiance_nee;        /* Color3f[S]  */
	const float *direction_nee;       /* Vector2f[S] canonical */
	const float *wo_pdf;              /* Float[S] */
} pg_dense_records;

typedef struct pg_records_out {
	float *position, *direction, *radiance, *wo_pdf, *direction_nee, *radiance_nee_lum;
} pg_records_out;

/* processPathData + scatterDataIntoSDTree's filter (path_guiding_integrator.py:434-497):
 * writes the surviving records (any order) with plane stride S = num_rays*max_depth and their
 * number to *d_count (device uint32, zeroed by the call). */
int pg_process_records(pg_context *ctx, uint64_t num_rays, int32_t max_depth, const float *l_final,
                       const pg_dense_records *rec, const pg_records_out *out, uint32_t *d_count,
                       void *stream);

/* Same, fused with pg_splat: no intermediate stream (path_guiding_integrator.py:388-395). */
int pg_process_and_splat(pg_context *ctx, uint64_t num_rays, int32_t max_depth,
                         const float *l_final, const pg_dense_records *rec, void *stream);

/* ---- per-iteration refinement --------------------------------------------------------- */

/* refineAndPrepareSDTreeForNextIteration (path_guiding_integrator.py:566-586) with the
 * iteration number given to pg_set_iteration: KD split, quadtree threshold/merge/split,
 * canonical re-layout, prev <- current, reset.  Synchronises `stream`.
 * A TRANSACTION, as the reference's allocate-new + copy growth is (common.py:161-189: its arrays are replaced, never edited, so
 * sdTree_prev survives a failed split): the refined KD tree, the new quadtree forest and the next iteration's accumulators are
 * built in spare buffers and committed by pointer swap.  On ANY error (PG_ERR_NOMEM from an allocation, a limit) sdTree_prev and
 * sdTree_current -- topology, sampling values and the running iteration's accumulators -- are exactly what they were before the
 * call: every query, pass and export answers bit for bit as before, and the call may simply be repeated. */
int pg_refine_and_swap(pg_context *ctx, void *stream);

/* Fault injection for the tests of that guarantee (tests/test_gpu_errors.py); process-wide, not for production use:
 * after `successes_before_failure` further device allocations of the library have succeeded, the next one reports out of memory
 * exactly as a refused hipMalloc does, and the hook disarms itself (-1 disarms it at once).  pg_debug_fail_alloc_pending
 * returns 1 while a failure is armed and has not happened yet, 0 otherwise (the hook fired or was never armed). */
int pg_debug_fail_alloc(int64_t successes_before_failure);
int pg_debug_fail_alloc_pending(void);

/* Views of sdTree_current's integer accumulators for the multi-GPU exchange: one contiguous
 * device buffer of `count` int64 that is element-wise summable across ranks (topology is
 * frozen during an iteration, so the buffers are index-aligned).  Sum it with an RCCL
 * all-reduce (ncclInt64, ncclSum) before pg_refine_and_swap. */
int pg_accumulators(pg_context *ctx, int64_t **d_buffer, uint64_t *count);

/* The exchange itself for hosts that do not bring their own collective library: one RCCL
 * all-reduce (ncclAllReduce, ncclInt64, ncclSum, in place) of that buffer over the ranks, asynchronous
 * on `stream`; call it on every rank after the iteration's last pass and before pg_refine_and_swap --
 * exact integer sums, so every rank then refines the same tree.  RCCL is loaded at run
 * time (the copy the process already holds, else librccl.so.1 from the library path, else
 * $PGSD_RCCL_LIBRARY); nothing of it is touched before the first pg_comm_* call.
 *   pg_comm_unique_id : ncclGetUniqueId into h_id_out[PG_COMM_ID_BYTES] (rank 0 calls it and hands the
 *                       bytes to the other ranks by whatever channel the host has)
 *   pg_comm_init      : ncclCommInitRank on the context's device; collective over the n_ranks callers
 *   pg_comm_attach    : use a communicator the host already owns (an ncclComm_t of the same RCCL
 *                       library; it stays the host's: pg_comm_destroy / pg_destroy do not destroy it)
 * There is no reference counterpart (single process, single GPU: SURVEY 8e). */
#define PG_COMM_ID_BYTES 128
int pg_comm_unique_id(pg_context *ctx, uint8_t *h_id_out);
int pg_comm_init(pg_context *ctx, int32_t n_ranks, int32_t rank, const uint8_t *h_id);
int pg_comm_attach(pg_context *ctx, void *nccl_comm, int32_t n_ranks);
int pg_comm_destroy(pg_context *ctx);
int pg_allreduce(pg_context *ctx, void *stream);
/* What RCCL itself says about the context's communicator -- ncclCommCount and ncclCommUserRank read back, not the numbers
 * the caller passed to pg_comm_init: a witness that the library's exchange really spans n ranks. */
int pg_comm_info(pg_context *ctx, int32_t *n_ranks_out, int32_t *rank_out);

/* The exchange format.  An accumulator is four int64 words on the device (three limbs of 32 payload bits + a record
 * count: 32 bytes, cheap to add to with 64-bit atomics); what has to travel between GPUs is its VALUE, 24 bytes:
 *     T = l0 + l1 2^32 + l2 2^64      p0 = T mod 2^52    p1 = (T >> 52) mod 2^52    p2 = (count << 24) + (T >> 104)
 * (|T| < 2^119 and sum of counts < 2^31 per accumulator and iteration over all ranks -- the bounds the limbs rely on
 * anyway).  Element-wise int64 sums of packed buffers over up to 2^11 ranks cannot overflow and decode to the exact sums.
 *   pg_exchange_pack   : packs sdTree_current's accumulators into a library-owned device buffer of *count int64
 *                        ([accumulators x 3 | per-tree fallback counters]) on `stream`; the accumulators stay as they are
 *   pg_exchange_unpack : writes the (summed) packed buffer back into the accumulators on `stream`: same values, same counts
 * A host with its own collective: pack, all-reduce *d_buffer (int64, sum), unpack, pg_refine_and_swap.  pg_allreduce does
 * exactly that with RCCL.  Summing the raw pg_accumulators buffer stays valid (a third more bytes).
 *   pg_exchange_pack_words / _unpack_words : the same arithmetic on HOST arrays of n_acc accumulators (4 words in, 3 out and
 *                        back); no context, no device. */
int pg_exchange_pack(pg_context *ctx, int64_t **d_buffer, uint64_t *count, void *stream);
int pg_exchange_unpack(pg_context *ctx, void *stream);
int pg_exchange_pack_words(const int64_t *h_acc, uint64_t n_acc, int64_t *h_out);
int pg_exchange_unpack_words(const int64_t *h_in, uint64_t n_acc, int64_t *h_acc_out);

/* ---- import / export in the reference's schema (kdtree.py:539-602) -------------------- */

typedef struct pg_tree_sizes {
	uint64_t n_kd, n_quad, n_roots;
} pg_tree_sizes;

/* HOST arrays in the reference's column layout: bbox columns are row-major [n][3] / [n][2]
 * exactly as `.numpy()` yields them. */
typedef struct pg_tree_columns {
	double kd_max_leaf_size;
	int32_t kd_max_depth, quad_max_depth, quad_store_nee;
	float *kd_bbox_min, *kd_bbox_max;
	uint32_t *kd_depth;
	float *kd_vert_count;
	uint8_t *kd_is_leaf;
	uint32_t *kd_quad_root_index, *kd_child_left, *kd_child_right;
	uint32_t *quad_root_node_index;
	float *quad_bbox_min, *quad_bbox_max;
	uint32_t *quad_depth;
	float *quad_irradiance;
	uint8_t *quad_is_leaf;
	float *quad_threshold;
	uint32_t *quad_child[4];
} pg_tree_columns;

/* Sizes of sdTree_prev in canonical layout. Synchronous. */
int pg_export_sizes(pg_context *ctx, pg_tree_sizes *sizes);
/* KDTree.saveToFile's columns for sdTree_prev (path_guiding_integrator.py:589-594). */
int pg_export(pg_context *ctx, const pg_tree_sizes *sizes, pg_tree_columns *h_out);
/* loadSDTreeFromFile (path_guiding_integrator.py:597-608): prev <- file, current <- reset copy. */
int pg_import(pg_context *ctx, const pg_tree_sizes *sizes, const pg_tree_columns *h_in);
/* Exact accumulators of sdTree_current resolved per canonical node (tests, diagnostics):
 * kd_count[n_kd] (records through each KD node), quad_acc_lo/hi[n_quad] (128-bit sums). */
int pg_export_accumulators(pg_context *ctx, const pg_tree_sizes *sizes, uint64_t *h_kd_count,
                           uint64_t *h_quad_acc_lo, int64_t *h_quad_acc_hi);

/* ---- renderer substrate: PathGuidingIntegrator.sample() on device ----- */

/* Scene subset of the reference's scenes/cornell-box/scene.xml: parallelograms ("rectangle" and
 * the six faces of "cube" shapes) with twosided diffuse BSDFs, one one-sided area emitter, a
 * perspective sensor.  h_quads is a HOST array of n_quads*24 floats:
 *   0-2 origin  3-5 edge1  6-8 edge2  9-11 unit normal (= normalised e1 x e2)
 *   12 1/|e1|^2  13 1/|e2|^2  14 area  15 emitter flag  16-18 diffuse reflectance
 *   19-21 emitted radiance  22 material index (pg_scene_set_ex with a material table)  23 unused
 * (practical_path_guiding_lab_amd/scene.py builds it from Mitsuba XML). */
#define PG_QUAD_STRIDE 24
/* sphere: 0-2 centre  3 radius  4 material index  5 emitter flag  6-8 emitted radiance  9-11 unused */
#define PG_SPHERE_STRIDE 12
/* material: 0 type (0 diffuse; 1 roughconductor, beckmann, sample_visible; 2 smooth conductor;
 *   3 smooth dielectric; 4 roughdielectric, beckmann, sample_visible)  1-3 reflectance |
 *   specular_reflectance  4 alpha (> 0: beckmann; < 0: ggx of roughness -alpha)
 *   5-7 eta (dielectrics: 5 = int_ior / ext_ior)  8-10 k
 *   11 one-sided flag (0 = wrapped in `twosided`)
 *   12 texture index + 1 (0: none): on triangles with texture coordinates the texture replaces
 *      words 1-3 (scenes/veach-ajar/scene.xml:29-80: `bitmap` reflectances, a `checkerboard`
 *      specular_reflectance)  13-15 unused */
#define PG_MATERIAL_STRIDE 16
/* texture, 16 32-bit words: 0 kind (1 `bitmap`, bilinear, repeat; 2 `checkerboard`)  1 width  2 height
 *   3 index of its first texel in `texels`  4-6 color0  7-9 color1 (checkerboard; f32 bit patterns)
 *   10-11 to_uv scale  12-13 to_uv offset (f32 bit patterns)  14-15 unused */
#define PG_TEXTURE_STRIDE 16
/* box (Mitsuba's `cube`: [-1,1]^3 under an affine to_world), intersected as three slabs in its local
 * frame instead of six quads: 0-8 rows of A = (linear part of to_world)^-1, 9-11 centre c
 * (local = A (p - c)), 12-20 outward unit normals of the +x, +y, +z faces (the normalised rows
 * of A), 21 material index, 22-31 unused by the library */
#define PG_BOX_STRIDE 32
typedef struct pg_camera {
	float origin[3];
	float axis_x[3], axis_y[3], axis_z[3]; /* columns of the sensor's to_world rotation */
	float tan_half_fov_x;
	int32_t width, height;
} pg_camera;
int pg_scene_set(pg_context *ctx, uint64_t n_quads, const float *h_quads, const pg_camera *cam);

/* The scene subset of scenes/veach-mis/scene.xml on top of that: "sphere" shapes (also as area
 * emitters, sampled over the cone they subtend), several emitters (one is chosen uniformly per
 * sample, scene.sample_emitter_direction), and a material table with twosided rough conductors
 * (Beckmann, visible-normal sampling).  Shapes are numbered quads first, then spheres.  All HOST
 * arrays; materials == NULL means "quad i is diffuse with quads[i][16..18]" (no spheres then). */
typedef struct pg_scene_desc {
	uint64_t n_quads;
	const float *quads;     /* PG_QUAD_STRIDE floats each */
	uint64_t n_spheres;
	const float *spheres;   /* PG_SPHERE_STRIDE floats each */
	uint64_t n_materials;
	const float *materials; /* PG_MATERIAL_STRIDE floats each */
	uint64_t n_boxes;
	const float *boxes;     /* PG_BOX_STRIDE floats each; need a material table */
	/* triangle meshes (`obj` / `serialized` shapes of the reference's scenes) behind one four-wide
	 * BVH, built by the caller (practical_path_guiding_lab_amd/mesh.py):
	 *   triangle, PG_TRI_STRIDE floats: 0-2 v0, 3-5 v1 - v0, 6-8 v2 - v0, 9-11 unit geometric
	 *     normal, 12 material index; stored in BVH leaf order
	 *   node, PG_BVH_STRIDE 32-bit words (128 bytes): 0-23 (f32) the boxes of its up to four
	 *     children, lo_x[4] lo_y[4] lo_z[4] hi_x[4] hi_y[4] hi_z[4]; 24-27 (u32) the children: a node
	 *     index, or 0x80000000 | (count-1) << 28 | first triangle for a leaf of 1..8 triangles, or
	 *     0xffffffff for none; 28-31 unused by the library.  Node 0 is the root (also when it holds
	 *     a single leaf), children have larger indices than their parent and one parent each, leaves
	 *     stay inside the triangle array, no root-to-node path leaves more than 32 siblings waiting,
	 *     at most 2^25 nodes (all checked).  The boxes of absent children are ignored.  The ray-casting
	 *     kernels read the first 48 nodes of the table from LDS: a builder that gives the lowest numbers
	 *     to the nodes most rays open (the biggest boxes; mesh.build_bvh does) makes them faster, any
	 *     numbering that keeps children behind their parent is walked correctly.  Shape numbers of
	 *     triangles follow the box faces. */
	uint64_t n_tris;
	const float *tris;
	uint64_t n_bvh_nodes;
	const uint32_t *bvh;
	/* `directional` emitters (scenes/torus/scene.xml), PG_DIRLIGHT_STRIDE floats each: 0-2 unit
	 * direction the light travels in, 3-5 irradiance.  They follow the area emitters in the uniform
	 * emitter choice.  bsphere: centre and radius of the scene's bounding sphere (Mitsuba places a
	 * directional sample two radii up the light's direction). */
	uint64_t n_dir_lights;
	const float *dir_lights;
	float bsphere[4];
	/* optional unit vertex normals, 9 floats per triangle in the order of `tris` (NULL: face
	 * normals).  The shading frame follows the interpolated normal, ray offsets the geometric one
	 * (Mitsuba's si.sh_frame.n and si.n). */
	const float *tri_normals;
	/* optional texture coordinates, 6 floats per triangle (uv0 uv1 uv2) in the order of `tris`, with v
	 * already flipped as Mitsuba's obj loader does (row 0 of a bitmap is v = 0); the texture table; the
	 * texels of all bitmaps (8-bit sRGB packed R | G << 8 | B << 16) and the 256 linear values they are
	 * looked up in (8-bit sRGB -> linear, as Mitsuba converts a JPG texel).  All HOST arrays. */
	const float *tri_uvs;
	uint64_t n_textures;
	const uint32_t *textures; /* PG_TEXTURE_STRIDE words each */
	uint64_t n_texels;
	const uint32_t *texels;
	const float *srgb_lut;    /* 256 floats */
} pg_scene_desc;
#define PG_DIRLIGHT_STRIDE 8
#define PG_TRI_STRIDE 16
#define PG_BVH_STRIDE 32
int pg_scene_set_ex(pg_context *ctx, const pg_scene_desc *scene, const pg_camera *cam);

typedef struct pg_pass_params {
	uint32_t seed;    /* sampler seed of the pass (main.py:218: initial_seed + cumm_spp) */
	int32_t spp;      /* samples per pixel traced by this pass; lane = pixel*spp + s (:414-417); see `batched` */
	int32_t rr_depth; /* Russian roulette from this depth on (:39, 375) */
	/* 0 or 1: the set of pass buffers this pass uses.  The passes of an iteration are independent (main.py:208-218 seeds
	 * each with initial_seed + cumm_spp), so two may be on the device at once: issue them alternately with slot 0 on one
	 * stream and slot 1 on another.  The per-pixel sums are still added in the order the passes were issued (the library
	 * orders the two streams there); sdTree_current receives integer adds, whose order is free.  Passes of one slot must
	 * be issued on one stream.  0 everywhere = the reference's one-pass-at-a-time behaviour. */
	int32_t slot;
	/* image tile traced by this call (multi-GPU sharding): pixels [pixel_begin, pixel_begin +
	 * pixel_count) in row-major film order; pixel_count 0 = up to the end of the film.  Sampler
	 * streams are keyed by the global lane id, so the union of tiles equals the full-frame pass.
	 * L_out / valid_out hold only the tile's lanes; sumL / sumL2 are full-film arrays. */
	uint64_t pixel_begin, pixel_count;
	/* Interleaved sharding (load balance across GPUs): with stripe_count > 1 the call traces the rows r
	 * of the film with (r / stripe_rows) % stripe_count == stripe_index -- bands of stripe_rows rows dealt
	 * round-robin -- in ascending order; pixel_begin must be 0 and pixel_count the number of pixels of
	 * those rows (or 0).  The tile-local pixel i of L_out / valid_out is column i % width of the
	 * (i / width)-th owned row.  stripe_count 0 or 1: the contiguous range above. */
	uint32_t stripe_rows, stripe_index, stripe_count;
	/* 0: one pass of `spp` samples per pixel, lane = pixel*spp + s, sampler stream = lane of `seed` (mi.render(spp=spp, seed)).
	 * 1: the call stands for `spp` consecutive ONE-sample passes with the seeds seed, seed + 1, ... seed + spp - 1 -- the
	 *    reference's training passes (main.py:192 renders them with spp 1, :218 seeds them initial_seed + cumm_spp) --
	 *    traced together in one wavefront: sample s of a pixel is the sample pass seed + s gives that pixel (stream =
	 *    pixel of seed + s).  Radiance per sample, the per-pixel sums (added in pass order), sdTree_current and -- with
	 *    pg_film_batched -- the developed images are those of the spp separate calls, bit for bit; only the number of
	 *    kernel launches differs.  L_out keeps the lane order pixel*spp + s. */
	uint32_t batched;
} pg_pass_params;

/* One call of PathGuidingIntegrator.sample() for all width*height*spp lanes
 * (path_guiding_integrator.py:126-431): camera rays, max_depth bounces with NEE + one-sample MIS
 * between BSDF and sdTree_prev sampling (guiding active when iteration > 1), record store, then
 * processPathData + scatterDataIntoSDTree into sdTree_current unless the iteration is final.
 * max_depth, iteration, is_final, store_nee and the sampling fraction come from pg_setup /
 * pg_set_iteration.  L_out: Color3f[lanes] planar (device); valid_out: uint8[lanes] or NULL;
 * sumL/sumL2: Color3f[width*height] planar accumulators (:400-429) or both NULL. */
int pg_render_pass(pg_context *ctx, const pg_pass_params *prm, float *L_out, uint8_t *valid_out,
                   float *sumL, float *sumL2, void *stream);

/* Not in the reference (Dr.Jit orders its kernels by data dependence on one stream): mode 1 lets the two kernels
 * of a mesh scene's bounce that do not depend on each other -- the SD-tree queries (k_wave_guide,
 * src/path_guiding_integrator.py:244, 301, 307) and the shadow rays (k_wave_cast, :213 test_visibility) -- run side by
 * side, the queries on a library-owned stream that forks from and joins the caller's stream inside pg_render_pass.
 * Results are identical (the two write disjoint buffers).  Default 0: every kernel on the caller's stream. */
int pg_render_overlap(pg_context *ctx, int32_t mode);

/* How the pipeline of mesh scenes cuts a bounce into kernels behind the closest hits (k_wave_trace) -- the same device
 * functions on the same numbers either way, results identical:
 *   0 (default)  one kernel, k_wave_shade: surface, emitter sample, BSDF sample, the SD-tree calls
 *                (src/path_guiding_integrator.py:244, 301, 307 and the leaves a record adds to), the shadow ray (:213), mixture
 *                pdfs, record, throughput, roulette, the next ray -- on one lane's registers, nothing of it through memory;
 *   1            three: k_wave_shade_a (:189-220, 272-297 and the SD-tree calls), k_wave_cast (:213, a persistent any-hit
 *                kernel), k_wave_shade_b (:247-261, 302-381), handing their results on through workspace planes;
 *   2            four: the SD-tree calls as a kernel of their own, k_wave_guide, which reads the vertex back from the
 *                workspace -- the hot path of SURVEY 8 by itself, for timing and counters (pg_kernel_timing.guide_ms;
 *                bench.py measures its roofline this way).  pg_render_overlap(1) implies it. */
int pg_render_stages(pg_context *ctx, int32_t mode);

/* Not in the reference (Dr.Jit's wavefront keeps pixel order): with `on`, the bounces of a mesh scene from the second one
 * up to rr_depth process the live list in a global spatial order -- the places sorted by the Morton cell of the vertex
 * the ray has just found (a device radix sort of 16-bit keys per bounce) -- so that the lanes of a wave stand next to
 * each other in the scene for the shading, the shadow rays, the SD-tree queries and, after the survivors are appended
 * in that order, the next bounce's closest hits.  A lane's result depends on its own state only: radiance, sums,
 * accumulators and trees are those of the unsorted run, bit for bit.  Default 0. */
int pg_render_sort(pg_context *ctx, int32_t on);

/* Allocates what pg_render_pass needs for passes of up to n_lanes lanes (pixels of the tile x spp) ahead
 * of time -- the reference allocates its numRays x max_depth record arrays in setup()
 * (path_guiding_integrator.py:93, 116); without this call the first pass of a size allocates them.
 * After the pass buffers, for passes of >= 2^20 lanes and only while the device still has four times the jump tables' budget
 * ($PGSD_JUMP_TABLE_MAX_BYTES, default 2 GiB) free, the tables' buffer is also taken at that budget, so that no refine of the
 * training allocates it; $PGSD_JUMP_TABLE_RESERVE=0 switches that off.  It changes allocation times only, never a result, and
 * a refused allocation of it is not an error. */
int pg_render_reserve(pg_context *ctx, uint64_t n_lanes);

/* Which kernels run a bounce.  Scenes of quads, boxes, spheres with diffuse / rough-conductor BSDFs run
 * one fused kernel per bounce; scenes with meshes or the other BSDFs run the split pipeline (ray casting,
 * shading, the SD-tree queries and the rest of the bounce as kernels of their own).  on != 0: every scene
 * set from now on runs the split pipeline.  Results are the same bit for bit either way (the two are
 * independent implementations of path_guiding_integrator.py:126-431: a cross-check, and a way to time the
 * SD-tree queries of any scene as a kernel of their own).  Takes effect at the next pg_scene_set[_ex]. */
int pg_render_split_pipeline(pg_context *ctx, int32_t on);

/* Film reconstruction of one full-frame pass with Mitsuba's `tent` reconstruction filter of radius
 * one pixel -- the <rfilter type="tent"/> of the reference's scenes (scenes/cornell-box/scene.xml:27),
 * i.e. the image mi.render returns at main.py:218.  `seed` and `spp` are those of the pass that
 * produced L (Color3f[W*H*spp] planar, as pg_render_pass wrote it): a sample's film position is
 * pixel + its first two sampler draws, which the kernel recomputes.  image_out: Color3f[W*H] planar,
 * sum(w L) / sum(w) over the samples of the pixel's 3x3 neighbourhood, summed in a fixed order. */
int pg_film_tent(pg_context *ctx, uint32_t seed, int32_t spp, const float *L, float *image_out, void *stream);
/* The same with a choice of filter: PG_FILTER_TENT, or PG_FILTER_GAUSSIAN = Mitsuba's default
 * gaussian (stddev 0.5, radius 2, w(d) = max(0, exp(-2 d^2) - exp(-8)) per axis; 5x5 neighbourhood),
 * the film of scenes/torus/scene.xml:46. */
#define PG_FILTER_TENT 0
#define PG_FILTER_GAUSSIAN 1
int pg_film(pg_context *ctx, int32_t filter, uint32_t seed, int32_t spp, const float *L, float *image_out, void *stream);

/* The same for a film sharded by interleaved bands of rows (pg_pass_params.stripe_*; not in the reference, which has
 * one device): develops only the pixels of the rows r with (r / stripe_rows) % stripe_count == stripe_index and leaves
 * the rest of image_out untouched.  L is still indexed by the full frame's lanes, but only this rank's rows and the
 * filter's reach beyond them -- one row for the tent filter, two for the gaussian -- are read: what a rank has to
 * fetch from its neighbours is that halo, not the film.  stripe_count 0 or 1: the whole film (pg_film). */
int pg_film_stripes(pg_context *ctx, int32_t filter, uint32_t seed, int32_t spp, const float *L, float *image_out,
                    uint32_t stripe_rows, uint32_t stripe_index, uint32_t stripe_count, void *stream);

/* The film of a batched call (pg_pass_params.batched = 1): n_passes images, image s = what pg_film_stripes develops for the
 * one-sample pass seed + s alone, from L as the batched pg_render_pass wrote it (lane = pixel*n_passes + s).
 * images_out: n_passes x Color3f[W*H] planar, image-major. */
int pg_film_batched(pg_context *ctx, int32_t filter, uint32_t seed, int32_t n_passes, const float *L, float *images_out,
                    uint32_t stripe_rows, uint32_t stripe_index, uint32_t stripe_count, void *stream);

/* The same without the n_passes images: the running mean main.py keeps of an iteration's passes (:218-239: every pass's image
 * times spp_of_pass / spp_of_iteration, summed in pass order).  acc_io (Color3f[W*H] planar) becomes
 * ((acc + image_0 * scale) + image_1 * scale) + ... in fp32, each product and each sum rounded on its own -- what a host
 * makes of the separate images; with acc_has_value 0 the first image is assigned, not added (acc_io may then be
 * uninitialised on the developed pixels).  A striped call touches its own rows only. */
int pg_film_batched_accumulate(pg_context *ctx, int32_t filter, uint32_t seed, int32_t n_passes, const float *L, float *acc_io,
                               float scale, int32_t acc_has_value, uint32_t stripe_rows, uint32_t stripe_index,
                               uint32_t stripe_count, void *stream);

/* Element-wise evaluation of the library's own fp32 transcendental functions (fixed
 * sequences of double operations, no vendor math library), so that a caller -- the parity tests --
 * can compare them with another implementation of the same contract.
 * which: 0 exp, 1 log, 2 erf, 3 erfinv, 4 sin, 5 cos.  x, out: float[n] device pointers. */
int pg_math_eval(pg_context *ctx, int32_t which, uint64_t n, const float *x, float *out, void *stream);

/* The ordering step of a sorted bounce (pg_render_sort) by itself, for tests: d_places_out[0 .. live) becomes a permutation of
 * the places 0 .. live-1 in the order of their 16-bit keys' HIGH byte -- for every n -- and, inside one high byte, in the order of
 * the low byte only as far as one 4096-pair tile of the second pass allows: equal or adjacent low bytes for n >= 2^21, coarser
 * below, none for n < 2^20 (csrc/pg_sort.hip: two unstable counting passes; the permutation differs from run to run; the
 * renderer's results do not depend on the order at all).  live = min(n, *d_live) -- the places that hold a path, read from device memory; d_live NULL: all n.  Entries of
 * d_places_out behind `live` are not written.  d_keys: uint16[n], d_places_out: uint32[n], device pointers.  Synchronises. */
int pg_sort_places(pg_context *ctx, uint64_t n, const uint16_t *d_keys, const uint32_t *d_live, uint32_t *d_places_out, void *stream);

/* Per-kernel device time of pg_render_pass, measured with HIP events recorded on the launch
 * stream around each kernel (off by default).  pg_read_kernel_timing synchronises. */
typedef struct pg_kernel_timing {
	double bounce_ms, splat_ms, generate_ms, finish_ms, compact_ms;
	uint64_t bounce_launches, splat_launches, passes;
	/* mesh scenes run a bounce as two to five kernels (pg_render_stages: closest hits; then one shading kernel,
	 * or shading before and after the shadow rays with the SD-tree calls in the first, or those calls as a kernel
	 * of their own): their shares of bounce_ms; bounce_launches then counts bounces.  shade_ms = the shading
	 * kernels (pg_render_stages(0): the one, with shadow_ms = guide_ms = shade_b_ms = 0); tail_ms = the launch
	 * that finishes the last paths. */
	double trace_ms, shade_ms, shadow_ms, guide_ms, tail_ms;
	uint64_t trace_launches, guide_launches;
	double shade_a_ms, shade_b_ms; /* the two shading kernels of shade_ms, each on its own */
	double sort_ms;                /* pg_render_sort: the radix sorts of the sorted bounces */
} pg_kernel_timing;
int pg_enable_kernel_timing(pg_context *ctx, int32_t on);
int pg_read_kernel_timing(pg_context *ctx, pg_kernel_timing *out, int32_t reset);

/* Paths still alive after each bounce of the most recent pg_render_pass: out[b] for
 * b < min(n, max_depth) (the loop condition `active` of path_guiding_integrator.py:179 counted
 * per iteration).  Synchronises the device. */
int pg_render_live_counts(pg_context *ctx, uint32_t *out, int32_t n);

/* ---- statistics for the roofline model ------------------------------------ */
typedef struct pg_stats {
	uint64_t n_kd_nodes, n_kd_leaves, n_quad_records, n_quad_nodes, n_trees;
	double mean_kd_leaf_depth;   /* over KD leaves                           */
	double mean_quad_leaf_depth; /* over quadtree leaves of all trees        */
	uint32_t max_kd_depth, max_quad_depth;
	uint64_t bytes_kd, bytes_quad_records, bytes_accumulators;
	/* the tables that stand in for the top levels of the descents (not in the reference): the quadtree jump tables
	 * (4^jump_bits entries of 16 bytes per quadtree; 0 bits: none -- the forest is too big for its memory budget,
	 * $PGSD_JUMP_TABLE_MAX_BYTES, default 2 GiB: a function of the forest and the budget alone, so that every rank and
	 * every run builds the same tables; coarser only when the table cannot be allocated) and the KD jump grid
	 * (8^kd_grid_bits cells of 16 bytes) */
	uint64_t bytes_jump_tables;
	uint32_t jump_bits, kd_grid_bits;
} pg_stats;
int pg_get_stats(pg_context *ctx, pg_stats *out);

/* Measured mean descent depths of the last pg_guide_bounce / pg_splat launch (device counters;
 * enabled by pg_enable_depth_counters(ctx, 1); off by default because they add atomics). */
typedef struct pg_depth_counters {
	uint64_t kd_levels, kd_queries;     /* sum of KD leaf depths, number of descents  */
	uint64_t quad_levels, quad_queries; /* sum of quadtree leaf depths, descents      */
	/* bytes the lanes of those descents gathered from the tables of the BUILT layout, nothing credited for lanes
	 * that share a cache line: 16 per KD grid entry and per KD node below it, 16 per jump-table entry, 32 per quadtree
	 * record of a pdf or sampling walk, 16 per record of a leaf walk (the 8-byte tree heads not included: add 8 per
	 * KD descent that is followed by a quadtree walk) */
	uint64_t layout_bytes;
} pg_depth_counters;
int pg_enable_depth_counters(pg_context *ctx, int32_t on);
int pg_read_depth_counters(pg_context *ctx, pg_depth_counters *out, int32_t reset);

/* Where a wave of k_wave_shade spends its life -- a PROBE BUILD's instrument (the library compiled with -DPG_SHADE_PHASES=1,
 * csrc/Makefile target `probe`: libpgsd_phases.so; the product build has no stamps, h_out[0] = 0 and zeros): the kernel that makes
 * the three SD-tree calls of a bounce (path_guiding_integrator.py:244, 301, 307) in the default pipeline also shades, walks the
 * shadow ray and appends the survivors, so the calls cannot be timed apart by events; the probe stamps the wave clock at the
 * kernel's start and at its seven phase boundaries (a wave's stamps wait in LDS and leave in one striped atomic: the probe's
 * kernel takes the product's time, profiles/r06/phase_probe.txt).  pg_enable_depth_counters(ctx, 2) switches the stamps on WITHOUT the depth counters (no atomics in the
 * walks themselves); pg_enable_depth_counters(ctx, 1) switches both on.  h_out[10]:
 *   [0] 1 when the stamps are compiled in, [1] waves that ran the kernel's body,
 *   [2..8] wave-clock cycles summed over those waves: records + staging, stage_a1, shadow walk, stage_a2, SD-tree calls,
 *          stage_b, survivors' append;  [9] 0. */
int pg_read_shade_phases(pg_context *ctx, uint64_t *h_out, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* PGSD_H */
