/*
 * pg_oracle_render.h -- TEST INFRASTRUCTURE.  CPU restatement of PathGuidingIntegrator.sample()
 * (src/path_guiding_integrator.py:126-431) over a minimal renderer substrate (quads, twosided
 * diffuse BSDFs, one-sided area emitters, perspective camera) that stands in for the Mitsuba 3
 * calls the reference makes (scene.ray_intersect, sample_emitter_direction, bsdf.sample, ...).
 *
 * PARITY UNPINNED for the substrate: Mitsuba is third-party and absent; the
 * formulas below follow Mitsuba 3's documented behaviour (concentric-disk cosine sampling,
 * area-light solid-angle pdf, Duff et al. frame) but cannot be checked against it here.
 * The integrator loop itself follows the reference line by line.
 */
#ifndef PG_ORACLE_RENDER_H
#define PG_ORACLE_RENDER_H

#include <stddef.h>
#include <stdint.h>

#include "pg_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one parallelogram: 24 floats
 *  0-2 origin  3-5 edge1  6-8 edge2  9-11 unit normal  12 1/|e1|^2  13 1/|e2|^2  14 area
 *  15 emitter flag  16-18 diffuse reflectance  19-21 emitted radiance  22-23 pad            */
#define PGO_QUAD_STRIDE 24

typedef struct pgo_camera {
	float origin[3];
	float axis_x[3], axis_y[3], axis_z[3]; /* world-space camera axes (columns of to_world) */
	float tan_half_fov_x;
	int32_t width, height;
} pgo_camera;

typedef struct pgo_render_params {
	int32_t max_depth, rr_depth;
	int32_t iteration, is_final;
	int32_t store_nee;
	float bsdf_sampling_fraction;
	uint32_t seed;
	int32_t spp; /* samples per pixel in this pass; lane = pixel*spp + s */
} pgo_render_params;

/* Traces one pass.  L_out: Color3f[n_lanes] planar; valid_out: u8[n_lanes] (depth != 0).
 * Records of the pass are post-processed and splatted into `current` unless is_final.
 * sumL / sumL2 (Color3f[width*height] planar) are accumulated as :400-429. */
void pgo_render_pass(const pgo_tree *prev, pgo_tree *current, size_t n_quads, const float *quads,
                     const pgo_camera *cam, const pgo_render_params *prm, float *L_out,
                     uint8_t *valid_out, float *sumL, float *sumL2);

/* Film reconstruction of one full-frame pass with Mitsuba's `tent` reconstruction filter, radius one
 * pixel (the <rfilter type="tent"/> of scenes/cornell-box/scene.xml:27): what mi.render returns at
 * main.py:218.  Sample s of pixel (px,py) sits at (px + jx, py + jy), its first two sampler draws;
 * pixel (x,y) receives weight tent(x + 0.5 - sx) * tent(y + 0.5 - sy), tent(d) = max(0, 1 - |d|),
 * and the image is sum(w L) / sum(w).  L: Color3f[W*H*spp] planar; image_out: Color3f[W*H] planar.
 * Third-party behaviour (hdrfilm + ImageBlock::put), unpinned like the rest of the substrate. */
void pgo_film_tent(uint32_t seed, int32_t spp, int32_t width, int32_t height, const float *L, float *image_out);

#ifdef __cplusplus
}
#endif
#endif
