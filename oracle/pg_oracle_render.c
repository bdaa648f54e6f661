/*
 * pg_oracle_render.c -- TEST INFRASTRUCTURE.  Scalar restatement of the reference's integrator
 * loop, PathGuidingIntegrator.sample() (src/path_guiding_integrator.py:126-431), one lane at a
 * time, on top of the oracle's SD-tree (pg_oracle.c) and a minimal quad/diffuse/area-light
 * substrate that replaces the Mitsuba calls (see pg_oracle_render.h for the unpinned-parity note).
 */
#include "pg_oracle_render.h"

#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "pgo_math.h"

static int g_threads = 1; /* lanes are traced by this many threads (pgo_set_threads) */

int pgo_set_threads(int n)
{
#ifdef _OPENMP
	if (n <= 0) n = omp_get_num_procs();
	g_threads = n;
#else
	(void)n;
	g_threads = 1;
#endif
	return g_threads;
}

int pgo_threads_in_effect(void) { return g_threads; }

/* scalar hooks exported by pg_oracle.c */
uint32_t pgo_i_quadtree_of(const pgo_tree *t, const float p[3], int active);
void pgo_i_sample(const pgo_tree *t, uint32_t root, uint64_t *state, uint64_t inc, int active, float dir[3]);
float pgo_i_pdf(const pgo_tree *t, uint32_t root, const float dir[3], int active);

typedef struct { float x, y, z; } v3;

#define INV_PI_F 0.31830988618379067154f
#define RAY_EPS_F 1e-4f
#define SHADOW_EPS_F 1e-3f

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline float max3(v3 a) { float m = a.x > a.y ? a.x : a.y; return m > a.z ? m : a.z; }

/* path_guiding_integrator.py:16-24 */
static inline float mis_weight(float a, float b)
{
	float a2 = a * a;
	float r = a > 0.0f ? a2 / (b * b + a2) : 0.0f;
	if (r != r) r = 0.0f;
	return r;
}

/* Mitsuba coordinate_system(n) (Duff et al. 2017) */
static inline void frame_from_normal(v3 n, v3 *s, v3 *t)
{
	float sign = n.z >= 0.0f ? 1.0f : -1.0f; /* copysign(1, n.z) up to -0 */
	if (pgo_f2u(n.z) >> 31) sign = -1.0f;
	float a = -1.0f / (sign + n.z);
	float b = (n.x * n.y) * a;
	*s = V(1.0f + (sign * (n.x * n.x)) * a, sign * b, -sign * n.x);
	*t = V(b, sign + (n.y * n.y) * a, -n.y);
}

typedef struct { v3 s, t, n; } frame;
static inline v3 to_local(const frame *f, v3 v) { return V(dot3(v, f->s), dot3(v, f->t), dot3(v, f->n)); }
static inline v3 to_world(const frame *f, v3 v)
{
	return vadd(vadd(vscale(f->s, v.x), vscale(f->t, v.y)), vscale(f->n, v.z));
}

#define PI_F 3.14159265358979323846f
#define INV_TWO_PI_F 0.15915494309189533577f
#define INV_SQRT_PI_F 0.56418958354775628695f
#define SPHERE_EPS_F 8.94069671630859375e-05f /* Mitsuba's math::RayEpsilon<float> = 1500 * 2^-24 */

static inline float safe_sqrtf(float v) { return sqrtf(v > 0.0f ? v : 0.0f); }
static inline v3 normalize3(v3 v) { return vdivs(v, sqrtf(dot3(v, v))); }

/* does the ray reach the box of child c of BVH node N before bt?  *tmin = where it enters (>= 0) */
static int bvh_box_hit(const uint32_t *N, int c, const float oo[3], const float inv[3], float bt, float *tmin_out)
{
	/* the plane met first on an axis follows from the sign of the direction (of its reciprocal);
	 * fmaxf / fminf ignore a NaN operand (0 * inf: the ray lies in a face's plane), which keeps the
	 * test conservative */
	float tn[3], tf[3];
	for (int k = 0; k < 3; ++k) {
		const float lo = pgo_u2f(N[4 * k + c]), hi = pgo_u2f(N[12 + 4 * k + c]);
		const int neg = signbit(inv[k]) != 0;
		tn[k] = ((neg ? hi : lo) - oo[k]) * inv[k];
		tf[k] = ((neg ? lo : hi) - oo[k]) * inv[k];
	}
	const float tmin = fmaxf(fmaxf(fmaxf(tn[0], tn[1]), tn[2]), 0.0f);
	const float tmax = fminf(fminf(fminf(tf[0], tf[1]), tf[2]), bt);
	*tmin_out = tmin;
	return tmin <= tmax * 1.0000004f;
}

/* closest hit over all shapes: 0 < t < tmax; returns the shape number (quads, spheres, box faces,
 * triangles) */
static int intersect(const pgo_scene *sc, v3 o, v3 d, float tmax, float *t_out)
{
	const size_t nq = sc->n_quads;
	const float *quads = sc->quads;
	int best = -1;
	float bt = tmax;
	for (size_t q = 0; q < nq; ++q) {
		const float *Q = quads + q * PGO_QUAD_STRIDE;
		v3 n = ld3(Q + 9);
		float denom = dot3(n, d);
		if (denom == 0.0f) continue;
		float t = dot3(n, vsub(ld3(Q), o)) / denom;
		if (!(t > 0.0f && t < bt)) continue;
		v3 w = vsub(vadd(o, vscale(d, t)), ld3(Q));
		float u = dot3(w, ld3(Q + 3)) * Q[12];
		float v = dot3(w, ld3(Q + 6)) * Q[13];
		if (u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f) { bt = t; best = (int)q; }
	}
	/* spheres: the quadratic in double precision, as Mitsuba's Sphere::ray_intersect_preliminary */
	for (size_t s = 0; s < sc->n_spheres; ++s) {
		const float *S = sc->spheres + s * PGO_SPHERE_STRIDE;
		const double ox = (double)o.x - (double)S[0], oy = (double)o.y - (double)S[1], oz = (double)o.z - (double)S[2];
		const double dx = (double)d.x, dy = (double)d.y, dz = (double)d.z, r = (double)S[3];
		const double A = (dx * dx + dy * dy) + dz * dz;
		const double B = 2.0 * ((ox * dx + oy * dy) + oz * dz);
		const double C = ((ox * ox + oy * oy) + oz * oz) - r * r;
		const double disc = B * B - (4.0 * A) * C;
		if (!(disc >= 0.0) || A == 0.0) continue;
		const double root = sqrt(disc);
		const double temp = -0.5 * (B + (B < 0.0 ? -root : root)); /* the cancellation-free root first */
		double x0 = temp / A, x1 = temp != 0.0 ? C / temp : x0;
		if (x0 > x1) { double tt = x0; x0 = x1; x1 = tt; }
		const float t = (float)(x0 > 0.0 ? x0 : x1);
		if (t > 0.0f && t < bt) { bt = t; best = (int)(nq + s); }
	}
	/* boxes: three slabs in the box's local frame (t is the same in both frames: A is linear) */
	for (size_t b = 0; b < sc->n_boxes; ++b) {
		const float *B = sc->boxes + b * PGO_BOX_STRIDE;
		const v3 oc = vsub(o, ld3(B + 9));
		const float ol[3] = { dot3(ld3(B), oc), dot3(ld3(B + 3), oc), dot3(ld3(B + 6), oc) };
		const float dl[3] = { dot3(ld3(B), d), dot3(ld3(B + 3), d), dot3(ld3(B + 6), d) };
		float tn = -INFINITY, tf = INFINITY;
		int an = 0, af = 0, miss = 0;
		for (int k = 0; k < 3; ++k) {
			if (dl[k] == 0.0f) { /* parallel to this slab: inside it or never */
				if (!(ol[k] >= -1.0f && ol[k] <= 1.0f)) miss = 1;
				continue;
			}
			const float inv = 1.0f / dl[k];
			const float t1 = (-1.0f - ol[k]) * inv, t2 = (1.0f - ol[k]) * inv;
			const float lo = dl[k] > 0.0f ? t1 : t2, hi = dl[k] > 0.0f ? t2 : t1;
			if (lo > tn) { tn = lo; an = k; }
			if (hi < tf) { tf = hi; af = k; }
		}
		if (miss || !(tn <= tf)) continue;
		const int entering = tn > 0.0f;
		const float t = entering ? tn : tf;
		if (!(t > 0.0f && t < bt)) continue;
		/* the face crossed: entering through the side that faces the ray, leaving through the one it points to */
		const int axis = entering ? an : af;
		const int negative = entering ? dl[axis] > 0.0f : dl[axis] < 0.0f;
		bt = t;
		best = (int)(nq + sc->n_spheres + 6 * b) + 2 * axis + negative;
	}
	/* triangle meshes: the four-wide BVH.  A node's (up to four) children are tested at once and
	 * ordered by where the ray enters them (a fixed five-comparator network, so that product and
	 * oracle agree on ties); the walk goes on in the nearest, the others wait on the stack with their
	 * entry distance, farthest at the bottom, and are dropped when popped if the ray has become
	 * shorter than that.  Slab test padded as Ize 2013, Moeller-Trumbore triangles.  (The order
	 * matters only for which of several equally near triangles is reported: the first one met.) */
	if (sc->n_bvh_nodes) {
		const size_t tri_base = nq + sc->n_spheres + 6 * sc->n_boxes;
		const float inv[3] = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
		const float oo[3] = { o.x, o.y, o.z };
		const uint32_t NONE = 0xffffffffu;
		uint32_t st_ref[64];
		float st_t[64];
		int sp = 0;
		uint32_t next = 0; /* the root node */
		for (;;) {
			if (next == NONE) { /* next candidate from the stack */
				if (!sp) break;
				--sp;
				if (!(st_t[sp] <= bt * 1.0000004f)) continue;
				next = st_ref[sp];
			}
			if (!(next & 0x80000000u)) { /* a node: test its children, go on in the nearest */
				const uint32_t *N = sc->bvh + (size_t)next * PGO_BVH_STRIDE;
				uint32_t r[4];
				float t[4];
				for (int c = 0; c < 4; ++c) {
					r[c] = N[24 + c];
					t[c] = INFINITY;
					if (r[c] != NONE && !bvh_box_hit(N, c, oo, inv, bt, &t[c])) { r[c] = NONE; t[c] = INFINITY; }
				}
#define PGO_CSWAP(a, b) if (t[a] > t[b]) { const float tt = t[a]; t[a] = t[b]; t[b] = tt; const uint32_t rr = r[a]; r[a] = r[b]; r[b] = rr; }
				PGO_CSWAP(0, 1) PGO_CSWAP(2, 3) PGO_CSWAP(0, 2) PGO_CSWAP(1, 3) PGO_CSWAP(1, 2)
#undef PGO_CSWAP
				for (int c = 3; c >= 1; --c)
					if (r[c] != NONE) { st_ref[sp] = r[c]; st_t[sp] = t[c]; ++sp; }
				next = r[0];
				continue;
			}
			{ /* a leaf */
				const uint32_t first = next & 0x0fffffffu, count = ((next >> 28) & 7u) + 1u;
				next = NONE;
				for (uint32_t i = first; i < first + count; ++i) {
					const float *T = sc->tris + (size_t)i * PGO_TRI_STRIDE;
					const v3 e1 = ld3(T + 3), e2 = ld3(T + 6);
					const v3 p = V(d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x);
					const float det = dot3(e1, p);
					if (det == 0.0f) continue;
					const float inv_det = 1.0f / det;
					const v3 s = vsub(o, ld3(T));
					const float u = dot3(s, p) * inv_det;
					if (!(u >= 0.0f && u <= 1.0f)) continue;
					const v3 q = V(s.y * e1.z - s.z * e1.y, s.z * e1.x - s.x * e1.z, s.x * e1.y - s.y * e1.x);
					const float v = dot3(d, q) * inv_det;
					if (!(v >= 0.0f && u + v <= 1.0f)) continue;
					const float t = dot3(e2, q) * inv_det;
					if (t > 0.0f && t < bt) { bt = t; best = (int)(tri_base + i); }
				}
			}
		}
	}
	*t_out = bt;
	return best;
}

/* outward unit normal of box face `face` (2 axis + negative) */
static v3 box_face_normal(const float *B, int face)
{
	const v3 n = ld3(B + 12 + 3 * (face >> 1));
	return (face & 1) ? V(-n.x, -n.y, -n.z) : n;
}

/* ---- surface description at a hit ---- */
typedef struct { int type; v3 refl; float alpha; v3 eta, k; int one_sided; int tex; } material;
typedef struct { v3 p, n, ng; int is_em; v3 radiance; material m; } surface; /* n: shading normal, ng: geometric */

static material load_material(const float *M)
{
	material m;
	m.type = (int)M[0];
	m.refl = ld3(M + 1);
	m.alpha = M[4];
	m.eta = ld3(M + 5);
	m.k = ld3(M + 8);
	m.one_sided = M[11] != 0.0f;
	m.tex = (int)M[12];
	return m;
}

/* `bitmap` (bilinear, repeat wrap) and `checkerboard` textures after Mitsuba 3's bitmap.cpp /
 * checkerboard.cpp: uv' = to_uv(uv) (scale and offset); checkerboard: color0 where frac(u') > .5 and
 * frac(v') > .5 agree, else color1; bitmap: texel centres at (i + .5) / size, the four neighbours
 * looked up in the sRGB table and blended in linear light, rows first */
static v3 texture_eval(const pgo_scene *sc, int index, float u, float v)
{
	const uint32_t *T = sc->textures + (size_t)index * PGO_TEXTURE_STRIDE;
	const float uu = pgo_u2f(T[10]) * u + pgo_u2f(T[12]);
	const float vv = pgo_u2f(T[11]) * v + pgo_u2f(T[13]);
	if (T[0] == 2u) {
		const float fu = uu - floorf(uu), fv = vv - floorf(vv);
		const int mx = fu > 0.5f, my = fv > 0.5f;
		return mx == my ? V(pgo_u2f(T[4]), pgo_u2f(T[5]), pgo_u2f(T[6])) : V(pgo_u2f(T[7]), pgo_u2f(T[8]), pgo_u2f(T[9]));
	}
	const int32_t W = (int32_t)T[1], H = (int32_t)T[2];
	float x = uu * (float)W - 0.5f, y = vv * (float)H - 0.5f;
	if (!(fabsf(x) < 1e9f)) x = 0.0f; /* NaN or far outside what an int holds */
	if (!(fabsf(y) < 1e9f)) y = 0.0f;
	const float fx = floorf(x), fy = floorf(y);
	const float wx1 = x - fx, wy1 = y - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
	const int32_t ix = (int32_t)fx, iy = (int32_t)fy;
	const int32_t ix0 = ((ix % W) + W) % W, ix1 = (((ix + 1) % W) + W) % W;
	const int32_t iy0 = ((iy % H) + H) % H, iy1 = (((iy + 1) % H) + H) % H;
	const uint32_t *tx = sc->texels + T[3];
	const uint32_t t00 = tx[(size_t)iy0 * W + ix0], t10 = tx[(size_t)iy0 * W + ix1];
	const uint32_t t01 = tx[(size_t)iy1 * W + ix0], t11 = tx[(size_t)iy1 * W + ix1];
	const float *lut = sc->srgb_lut;
	float out[3];
	for (int c = 0; c < 3; ++c) {
		const float c00 = lut[(t00 >> (8 * c)) & 255u], c10 = lut[(t10 >> (8 * c)) & 255u];
		const float c01 = lut[(t01 >> (8 * c)) & 255u], c11 = lut[(t11 >> (8 * c)) & 255u];
		out[c] = (c00 * wx0 + c10 * wx1) * wy0 + (c01 * wx0 + c11 * wx1) * wy1;
	}
	return V(out[0], out[1], out[2]);
}

void pgo_texture_eval(const pgo_scene *sc, int index, float u, float v, float rgb[3])
{
	const v3 r = texture_eval(sc, index, u, v);
	rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}

static surface surface_at(const pgo_scene *sc, int prim, v3 o, v3 d, float t)
{
	surface s;
	memset(&s, 0, sizeof s);
	int mi;
	if ((size_t)prim < sc->n_quads) {
		const float *Q = sc->quads + (size_t)prim * PGO_QUAD_STRIDE;
		s.p = vadd(o, vscale(d, t));
		s.n = ld3(Q + 9);
		s.is_em = Q[15] != 0.0f;
		s.radiance = ld3(Q + 19);
		if (!sc->materials) { s.m.type = 0; s.m.refl = ld3(Q + 16); return s; }
		mi = (int)Q[22];
	} else if ((size_t)prim >= sc->n_quads + sc->n_spheres + 6 * sc->n_boxes) {
		const size_t ti = (size_t)prim - sc->n_quads - sc->n_spheres - 6 * sc->n_boxes;
		const float *T = sc->tris + ti * PGO_TRI_STRIDE;
		s.p = vadd(o, vscale(d, t));
		s.n = ld3(T + 9); /* face normal */
		mi = (int)T[12];
		s.m = load_material(sc->materials + (size_t)mi * PGO_MATERIAL_STRIDE);
		const int textured = s.m.tex > 0 && sc->tri_uvs;
		if (sc->tri_normals || textured) { /* the barycentrics of the hit, by the intersection's own formulas */
			const v3 e1 = ld3(T + 3), e2 = ld3(T + 6);
			const v3 pp = V(d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x);
			const float inv_det = 1.0f / dot3(e1, pp);
			const v3 sv = vsub(o, ld3(T));
			const float u = dot3(sv, pp) * inv_det;
			const v3 qq = V(sv.y * e1.z - sv.z * e1.y, sv.z * e1.x - sv.x * e1.z, sv.x * e1.y - sv.y * e1.x);
			const float v = dot3(d, qq) * inv_det;
			const float b0 = (1.0f - u) - v;
			if (sc->tri_normals) { /* interpolated vertex normals */
				const float *Nn = sc->tri_normals + ti * 9;
				const v3 ns = vadd(vadd(vscale(ld3(Nn), b0), vscale(ld3(Nn + 3), u)), vscale(ld3(Nn + 6), v));
				const float l2 = dot3(ns, ns);
				if (l2 > 0.0f) {
					s.ng = s.n;
					s.n = vdivs(ns, sqrtf(l2));
				}
			}
			if (textured) { /* interpolated texture coordinates, then the texture in place of the reflectance */
				const float *U = sc->tri_uvs + ti * 6;
				const float tu = (U[0] * b0 + U[2] * u) + U[4] * v;
				const float tv = (U[1] * b0 + U[3] * u) + U[5] * v;
				s.m.refl = texture_eval(sc, s.m.tex - 1, tu, tv);
			}
		}
		return s;
	} else if ((size_t)prim >= sc->n_quads + sc->n_spheres) {
		const size_t f = (size_t)prim - sc->n_quads - sc->n_spheres;
		const float *B = sc->boxes + (f / 6) * PGO_BOX_STRIDE;
		s.p = vadd(o, vscale(d, t));
		s.n = box_face_normal(B, (int)(f % 6));
		mi = (int)B[21];
	} else {
		const float *S = sc->spheres + ((size_t)prim - sc->n_quads) * PGO_SPHERE_STRIDE;
		const v3 c = ld3(S);
		/* sphere.h: n = normalize(ray(t) - c), p = c + n r.  The normal is then taken again from the
		 * re-projected point, so that it is a function of p alone (the device recomputes the ray
		 * origin of the next bounce from the stored vertex) */
		const v3 n0 = normalize3(vsub(vadd(o, vscale(d, t)), c));
		s.p = vadd(c, vscale(n0, S[3]));
		s.n = normalize3(vsub(s.p, c));
		s.is_em = S[5] != 0.0f;
		s.radiance = ld3(S + 6);
		mi = (int)S[4];
	}
	s.m = load_material(sc->materials + (size_t)mi * PGO_MATERIAL_STRIDE);
	return s;
}

/* scene.pdf_emitter_direction(prev, ds) for a hit on emitter shape `prim` at p (normal n) seen from
 * ref: the solid-angle density emitter sampling has for that direction, times the 1/count of the
 * uniform emitter choice */
static float emitter_hit_pdf(const pgo_scene *sc, int prim, v3 ref, v3 p, v3 n, float inv_count)
{
	const v3 dd = vsub(p, ref);
	const float d2 = dot3(dd, dd), dist = sqrtf(d2);
	const v3 dn = vdivs(dd, dist);
	const float dp = dot3(dn, n);
	if (!(dp < 0.0f)) return 0.0f;
	float pdf;
	if ((size_t)prim < sc->n_quads) {
		const float *Q = sc->quads + (size_t)prim * PGO_QUAD_STRIDE;
		pdf = d2 / (fabsf(dp) * Q[14]);
	} else { /* Sphere::pdf_direction */
		const float *S = sc->spheres + ((size_t)prim - sc->n_quads) * PGO_SPHERE_STRIDE;
		const v3 cv = vsub(ld3(S), ref);
		const float sin_alpha = S[3] / sqrtf(dot3(cv, cv));
		const float cos_alpha = safe_sqrtf(1.0f - sin_alpha * sin_alpha);
		if (sin_alpha < 0.99999994f) pdf = INV_TWO_PI_F / (1.0f - cos_alpha);
		else pdf = (d2 / fabsf(dp)) / ((4.0f * PI_F) * (S[3] * S[3]));
	}
	return pdf * inv_count;
}

/* scene.sample_emitter_direction(si, (e1, e2), test_visibility=True): uniform choice of one
 * emitter (e1 is reused after the choice), then a point on it; returns ds.d, ds.pdf and
 * radiance / pdf (zero when occluded, facing away, or from inside a sphere) */
static void sample_emitter(const pgo_scene *sc, const int *em, int n_em, v3 p, v3 n, float e1, float e2, v3 *ds_d,
                           float *ds_pdf, v3 *em_weight, int *ds_delta)
{
	*ds_d = V(0, 0, 0);
	*ds_pdf = 0.0f;
	*em_weight = V(0, 0, 0);
	*ds_delta = 0;
	if (n_em <= 0) return;
	const float count = (float)n_em, inv_count = 1.0f / count;
	uint32_t idx = (uint32_t)(e1 * count);
	if (idx > (uint32_t)(n_em - 1)) idx = (uint32_t)(n_em - 1);
	e1 = e1 * count - (float)idx;
	const int prim = em[idx];
	if (prim < 0) { /* directional.cpp sample_direction: a point 2 radii up the light's direction, pdf 1, delta */
		const float *Dl = sc->dir_lights + (size_t)(-1 - prim) * PGO_DIRLIGHT_STRIDE;
		const v3 dl = ld3(Dl);
		const v3 cd = vsub(p, V(sc->bsphere[0], sc->bsphere[1], sc->bsphere[2]));
		const float dc = sqrtf(dot3(cd, cd));
		const float dist = 2.0f * (sc->bsphere[3] > dc ? sc->bsphere[3] : dc);
		const v3 pl = vsub(p, vscale(dl, dist));
		*ds_d = V(-dl.x, -dl.y, -dl.z);
		*ds_delta = 1;
		*ds_pdf = 1.0f * inv_count;
		float mag = (1.0f + max3(V(fabsf(p.x), fabsf(p.y), fabsf(p.z)))) * RAY_EPS_F;
		if (dot3(n, *ds_d) < 0.0f) mag = -mag;
		const v3 so = vadd(p, vscale(n, mag));
		const v3 sd = vsub(pl, so);
		const float sdist = sqrtf(dot3(sd, sd));
		float th;
		const int occ = intersect(sc, so, vdivs(sd, sdist), sdist * (1.0f - SHADOW_EPS_F), &th) >= 0;
		if (!occ) *em_weight = vscale(ld3(Dl + 3), count);
		return;
	}
	v3 pl, ln, radiance;
	float pdf_cone = 0.0f; /* spheres: density over directions, known before the geometry term */
	int is_sphere = (size_t)prim >= sc->n_quads;
	if (!is_sphere) {
		const float *E = sc->quads + (size_t)prim * PGO_QUAD_STRIDE;
		pl = vadd(vadd(ld3(E), vscale(ld3(E + 3), e1)), vscale(ld3(E + 6), e2));
		ln = ld3(E + 9);
		radiance = ld3(E + 19);
	} else { /* Sphere::sample_direction, reference point outside */
		const float *S = sc->spheres + ((size_t)prim - sc->n_quads) * PGO_SPHERE_STRIDE;
		const v3 c = ld3(S);
		const float r = S[3];
		const v3 dc_v = vsub(c, p);
		const float dc_2 = dot3(dc_v, dc_v);
		const float radius_adj = r * (1.0f - SPHERE_EPS_F);
		if (!(dc_2 > radius_adj * radius_adj)) return;
		const float inv_dc = 1.0f / sqrtf(dc_2);
		const float sin_max = r * inv_dc, sin_max2 = sin_max * sin_max, inv_sin_max = 1.0f / sin_max;
		const float cos_max = safe_sqrtf(1.0f - sin_max2);
		float sin_theta_2;
		if (sin_max2 > 0.00068523f) { /* sin^2(1.5 deg) */
			const float tt = 1.0f + (cos_max - 1.0f) * e1;
			sin_theta_2 = 1.0f - tt * tt;
		} else sin_theta_2 = sin_max2 * e1; /* small-angle Taylor expansion */
		const float cos_theta = safe_sqrtf(1.0f - sin_theta_2);
		const float cos_alpha = sin_theta_2 * inv_sin_max +
		                        cos_theta * safe_sqrtf(1.0f - sin_theta_2 * (inv_sin_max * inv_sin_max));
		const float sin_alpha = safe_sqrtf(1.0f - cos_alpha * cos_alpha);
		float sin_phi, cos_phi;
		pgo_sincos(e2 * (2.0f * PI_F), &sin_phi, &cos_phi);
		frame fr;
		fr.n = vscale(dc_v, -inv_dc);
		frame_from_normal(fr.n, &fr.s, &fr.t);
		const v3 dl = to_world(&fr, V(cos_phi * sin_alpha, sin_phi * sin_alpha, cos_alpha));
		pl = vadd(c, vscale(dl, r));
		ln = dl;
		radiance = ld3(S + 6);
		pdf_cone = INV_TWO_PI_F / (1.0f - cos_max);
	}
	const v3 dir0 = vsub(pl, p);
	/* si.spawn_ray_to(pl): offset origin, then aim at the light point */
	float mag = (1.0f + max3(V(fabsf(p.x), fabsf(p.y), fabsf(p.z)))) * RAY_EPS_F;
	if (dot3(n, dir0) < 0.0f) mag = -mag;
	const v3 so = vadd(p, vscale(n, mag));
	const float d2 = dot3(dir0, dir0), dist = sqrtf(d2);
	*ds_d = vdivs(dir0, dist);
	const float dp = dot3(*ds_d, ln);
	float pdf = 0.0f;
	if (dp < 0.0f) pdf = is_sphere ? (dist == 0.0f ? 0.0f : pdf_cone) : d2 / (fabsf(dp) * sc->quads[(size_t)prim * PGO_QUAD_STRIDE + 14]);
	if (!(pdf == pdf) || pdf == INFINITY) pdf = 0.0f;
	*ds_pdf = pdf * inv_count;
	if (pdf > 0.0f) {
		const v3 sd = vsub(pl, so);
		const float sdist = sqrtf(dot3(sd, sd));
		const v3 sdn = vdivs(sd, sdist);
		float th;
		const int occ = intersect(sc, so, sdn, sdist * (1.0f - SHADOW_EPS_F), &th) >= 0;
		if (!occ) *em_weight = vscale(vdivs(radiance, pdf), count);
	}
}

/* Mitsuba warp::square_to_uniform_disk_concentric */
static inline void square_to_disk(float u, float v, float *px, float *py)
{
	float x = 2.0f * u - 1.0f, y = 2.0f * v - 1.0f;
	int is_zero = (x == 0.0f) && (y == 0.0f);
	int q13 = fabsf(x) < fabsf(y);
	float r = q13 ? y : x, rp = q13 ? x : y;
	float phi = (0.25f * 3.14159265358979323846f) * (rp / r);
	if (q13) phi = (0.5f * 3.14159265358979323846f) - phi;
	if (is_zero) phi = 0.0f;
	float s, c;
	pgo_sincos(phi, &s, &c);
	*px = r * c;
	*py = r * s;
}

/* Mitsuba warp::square_to_cosine_hemisphere (concentric disk + z = safe_sqrt(1 - r^2)) */
static inline v3 square_to_cosine_hemisphere(float u, float v)
{
	float px, py;
	square_to_disk(u, v, &px, &py);
	float zz = 1.0f - (px * px + py * py);
	float z = zz > 0.0f ? sqrtf(zz) : 0.0f;
	if (z == 0.0f) z = 1e-10f;
	return V(px, py, z);
}

/* ---- roughconductor (isotropic, sample_visible) after Mitsuba 3's microfacet.h /
 * roughconductor.cpp; every function below works in the local frame with cos(theta_i) > 0.
 * The sign of `alpha` names the distribution: > 0 Beckmann, < 0 GGX of roughness -alpha ---- */
static float rc_D(v3 m, float alpha) /* MicrofacetDistribution::eval */
{
	const float ct = m.z, ct2 = ct * ct;
	const float a = fabsf(alpha);
	const float ax = m.x / a, ay = m.y / a;
	float result;
	if (alpha < 0.0f) {
		const float t = (ax * ax + ay * ay) + ct2;
		result = 1.0f / (((PI_F * a) * a) * (t * t));
	} else {
		result = pgo_exp(-((ax * ax + ay * ay) / ct2)) / (((PI_F * a) * a) * (ct2 * ct2));
	}
	return result * ct > 1e-20f ? result : 0.0f; /* "prevent potential numerical issues in other stages" */
}

static float rc_G1(v3 v, v3 m, float alpha) /* smith_g1: exact for GGX, the rational approximation for Beckmann */
{
	const float ax = alpha * v.x, ay = alpha * v.y;
	const float xy = ax * ax + ay * ay;
	const float a = 1.0f / sqrtf(xy / (v.z * v.z));
	const float a2 = a * a;
	float result = a >= 1.6f ? 1.0f : (3.535f * a + 2.181f * a2) / ((1.0f + 2.276f * a) + 2.577f * a2);
	if (alpha < 0.0f) result = 2.0f / (1.0f + sqrtf(1.0f + xy / (v.z * v.z)));
	if (xy == 0.0f) result = 1.0f;                 /* perpendicular incidence */
	if (dot3(v, m) * v.z <= 0.0f) result = 0.0f;    /* the back of a microfacet is not seen from the front */
	return result;
}

static float fresnel_conductor(float cos_i, float eta_r, float eta_i)
{
	const float c2 = cos_i * cos_i, s2 = 1.0f - c2, s4 = s2 * s2;
	const float temp_1 = (eta_r * eta_r - eta_i * eta_i) - s2;
	const float a2pb2 = safe_sqrtf(temp_1 * temp_1 + ((4.0f * eta_i) * eta_i) * (eta_r * eta_r));
	const float a = safe_sqrtf(0.5f * (a2pb2 + temp_1));
	const float term_1 = a2pb2 + c2, term_2 = (2.0f * cos_i) * a;
	const float r_s = (term_1 - term_2) / (term_1 + term_2);
	const float term_3 = a2pb2 * c2 + s4, term_4 = term_2 * s2;
	const float r_p = r_s * ((term_3 - term_4) / (term_3 + term_4));
	return 0.5f * (r_s + r_p);
}

static v3 rc_fresnel(const material *m, float cos_i)
{
	return V(fresnel_conductor(cos_i, m->eta.x, m->k.x), fresnel_conductor(cos_i, m->eta.y, m->k.y),
	         fresnel_conductor(cos_i, m->eta.z, m->k.z));
}

/* sample_visible_11: slopes of the visible Beckmann normals for alpha = 1 (numerical inversion,
 * three Newton steps in the erf domain) */
static void rc_sample_visible_11(float cos_i, float u1, float u2, float *sx, float *sy)
{
	const float tan_i = safe_sqrtf(1.0f - cos_i * cos_i) / cos_i;
	const float cot_i = 1.0f / tan_i;
	const float maxval = pgo_erf(cot_i);
	u1 = u1 < 1.0f - 1e-6f ? u1 : 1.0f - 1e-6f; u1 = u1 > 1e-6f ? u1 : 1e-6f;
	u2 = u2 < 1.0f - 1e-6f ? u2 : 1.0f - 1e-6f; u2 = u2 > 1e-6f ? u2 : 1e-6f;
	float x = maxval - (maxval + 1.0f) * pgo_erf(sqrtf(-pgo_log(u1)));
	/* tan(theta) exp(-cot^2): 0 at normal incidence (inf * 0 otherwise) */
	const float tail = tan_i == 0.0f ? 0.0f : (INV_SQRT_PI_F * tan_i) * pgo_exp(-(cot_i * cot_i));
	u1 = u1 * ((1.0f + maxval) + tail);
	for (int i = 0; i < 3; ++i) {
		const float slope = pgo_erfinv(x);
		const float value = ((1.0f + x) + (INV_SQRT_PI_F * tan_i) * pgo_exp(-(slope * slope))) - u1;
		const float derivative = 1.0f - slope * tan_i;
		x = x - value / derivative;
	}
	*sx = pgo_erfinv(x);
	*sy = pgo_erfinv(2.0f * u2 - 1.0f);
}

/* sample_visible_11 for GGX: a point of the unit disk, its half towards the viewer compressed by
 * (1 + cos theta_i)/2, projected onto the hemisphere around the viewing direction, as slopes */
static void ggx_sample_visible_11(float cos_i, float u1, float u2, float *sx, float *sy)
{
	float px, py;
	square_to_disk(u1, u2, &px, &py);
	const float s = 0.5f * (1.0f + cos_i);
	const float h = safe_sqrtf(1.0f - px * px);
	py = h * (1.0f - s) + py * s;
	const float z = safe_sqrtf(1.0f - (px * px + py * py));
	const float sin_i = safe_sqrtf(1.0f - cos_i * cos_i);
	const float norm = 1.0f / (sin_i * py + cos_i * z);
	*sx = (cos_i * py - sin_i * z) * norm;
	*sy = px * norm;
}

/* MicrofacetDistribution::sample (visible normals): microfacet normal and its density */
static v3 rc_sample_m(v3 wi, float signed_alpha, float u1, float u2, float *pdf)
{
	const float alpha = fabsf(signed_alpha);
	const v3 wip = normalize3(V(alpha * wi.x, alpha * wi.y, wi.z));
	const float s2 = wip.x * wip.x + wip.y * wip.y; /* Frame::sincos_phi */
	float cos_phi = 1.0f, sin_phi = 0.0f;
	if (fabsf(s2) > 4.0f * 5.9604644775390625e-08f) {
		const float inv = 1.0f / sqrtf(s2);
		cos_phi = wip.x * inv; sin_phi = wip.y * inv;
		cos_phi = cos_phi < -1.0f ? -1.0f : (cos_phi > 1.0f ? 1.0f : cos_phi);
		sin_phi = sin_phi < -1.0f ? -1.0f : (sin_phi > 1.0f ? 1.0f : sin_phi);
	}
	float sx, sy;
	if (signed_alpha < 0.0f) ggx_sample_visible_11(wip.z, u1, u2, &sx, &sy);
	else rc_sample_visible_11(wip.z, u1, u2, &sx, &sy);
	const float rx = (cos_phi * sx - sin_phi * sy) * alpha;
	const float ry = (sin_phi * sx + cos_phi * sy) * alpha;
	const v3 m = normalize3(V(-rx, -ry, 1.0f));
	*pdf = ((rc_D(m, signed_alpha) * rc_G1(wi, m, signed_alpha)) * fabsf(dot3(wi, m))) / wi.z;
	return m;
}

static void rc_eval_pdf(const material *mt, v3 wi, v3 wo, v3 *value, float *pdf) /* wi.z > 0 */
{
	*value = V(0, 0, 0);
	*pdf = 0.0f;
	if (!(wi.z > 0.0f && wo.z > 0.0f)) return;
	const v3 H = normalize3(vadd(wo, wi));
	const float D = rc_D(H, mt->alpha);
	if (D == 0.0f) return;
	const float g_i = rc_G1(wi, H, mt->alpha);
	const float res = (D * (g_i * rc_G1(wo, H, mt->alpha))) / (4.0f * wi.z);
	const v3 F = rc_fresnel(mt, dot3(wi, H));
	*value = vmul(F, vscale(mt->refl, res));
	if (dot3(wi, H) > 0.0f && dot3(wo, H) > 0.0f) *pdf = (D * g_i) / (4.0f * wi.z);
}

static void rc_sample(const material *mt, v3 wi, float u1, float u2, v3 *wo, float *pdf, v3 *weight) /* wi.z > 0 */
{
	*wo = V(0, 0, 0); *pdf = 0.0f; *weight = V(0, 0, 0);
	float pdf_m;
	const v3 m = rc_sample_m(wi, mt->alpha, u1, u2, &pdf_m);
	const float wim = dot3(wi, m);
	const v3 o = vsub(vscale(m, 2.0f * wim), wi); /* reflect(wi, m) */
	if (!(pdf_m != 0.0f && o.z > 0.0f)) return;
	const float p = pdf_m / (4.0f * dot3(o, m));
	const v3 F = rc_fresnel(mt, wim);
	*wo = o;
	*pdf = p;
	*weight = vmul(F, vscale(mt->refl, rc_G1(o, m, mt->alpha)));
}

/* Mitsuba fresnel(cos_theta_i, eta): unpolarised reflectance of a dielectric interface, the signed
 * cosine of the transmitted direction, and the relative index along / against the ray */
static float fresnel_dielectric(float cos_i, float eta, float *cos_t, float *eta_it, float *eta_ti)
{
	const int outside = cos_i >= 0.0f;
	const float rcp_eta = 1.0f / eta;
	*eta_it = outside ? eta : rcp_eta;
	*eta_ti = outside ? rcp_eta : eta;
	const float cos_t_sqr = 1.0f - ((1.0f - cos_i * cos_i) * (*eta_ti * *eta_ti));
	const float ci = fabsf(cos_i), ct = safe_sqrtf(cos_t_sqr);
	const float a_s = (*eta_it * ct - ci) / (*eta_it * ct + ci);
	const float a_p = (*eta_it * ci - ct) / (*eta_it * ci + ct);
	float r = 0.5f * (a_s * a_s + a_p * a_p);
	if (eta == 1.0f) r = 0.0f;
	else if (ci == 0.0f) r = 1.0f;
	*cos_t = cos_i >= 0.0f ? -ct : ct; /* on the other side of the interface */
	return r;
}

/* ---- roughdielectric (Beckmann, isotropic, sample_visible) after Mitsuba 3's roughdielectric.cpp:
 * reflection and transmission through a rough interface, radiance transport.  wi may be on either
 * side; eta = int_ior / ext_ior ---- */
static inline v3 vflip_if(v3 v, int c) { return c ? V(-v.x, -v.y, -v.z) : v; }

static v3 rd_half_vector(v3 wi, v3 wo, float eta_side, int reflect)
{
	v3 m = normalize3(vadd(wi, vscale(wo, reflect ? 1.0f : eta_side)));
	return vflip_if(m, m.z < 0.0f); /* into the hemisphere of the macro-surface normal */
}

static void rd_eval_pdf(const material *mt, v3 wi, v3 wo, v3 *value, float *pdf)
{
	*value = V(0, 0, 0);
	*pdf = 0.0f;
	const float alpha = mt->alpha, eta_m = mt->eta.x;
	const float ci = wi.z, co = wo.z;
	if (ci == 0.0f) return;
	const int reflect = ci * co > 0.0f;
	const float eta = ci > 0.0f ? eta_m : 1.0f / eta_m, inv_eta = ci > 0.0f ? 1.0f / eta_m : eta_m;
	const v3 m = rd_half_vector(wi, wo, eta, reflect);
	const float D = rc_D(m, alpha);
	float cos_t, eta_it, eta_ti;
	const float wim = dot3(wi, m), wom = dot3(wo, m);
	const float F = fresnel_dielectric(wim, eta_m, &cos_t, &eta_it, &eta_ti);
	const float G = rc_G1(wi, m, alpha) * rc_G1(wo, m, alpha);
	float val;
	if (reflect) val = ((F * D) * G) / (4.0f * fabsf(ci));
	else {
		const float denom = wim + eta * wom;
		val = fabsf(((((((inv_eta * inv_eta) * (1.0f - F)) * D) * G) * (eta * eta)) * (wim * wom)) / (ci * (denom * denom)));
	}
	if (!(val == val)) val = 0.0f;
	*value = V(val, val, val);
	/* pdf: the micro- and macro-surface must agree on the sides */
	if (!(wim * ci > 0.0f && wom * co > 0.0f)) return;
	const float denom = wim + eta * wom;
	const float dwh_dwo = reflect ? 1.0f / (4.0f * wom) : ((eta * eta) * wom) / (denom * denom);
	const v3 wiu = vflip_if(wi, ci < 0.0f);
	float prob = ((D * rc_G1(wiu, m, alpha)) * fabsf(dot3(wiu, m))) / wiu.z;
	prob = prob * (reflect ? F : 1.0f - F);
	float p = prob * fabsf(dwh_dwo);
	if (!(p == p)) p = 0.0f;
	*pdf = p;
}

static void rd_sample(const material *mt, v3 wi, float u1, float u, float v, v3 *wo, float *pdf, v3 *weight, float *eta_out)
{
	*wo = V(0, 0, 0); *pdf = 0.0f; *weight = V(0, 0, 0); *eta_out = 0.0f;
	const float alpha = mt->alpha, eta_m = mt->eta.x;
	const float ci = wi.z;
	if (ci == 0.0f) return;
	float pdf_m;
	const v3 m = rc_sample_m(vflip_if(wi, ci < 0.0f), alpha, u, v, &pdf_m);
	if (!(pdf_m != 0.0f)) return;
	float cos_t, eta_it, eta_ti;
	const float wim = dot3(wi, m);
	const float F = fresnel_dielectric(wim, eta_m, &cos_t, &eta_it, &eta_ti);
	const int reflect = u1 <= F;
	float p = pdf_m * (reflect ? F : 1.0f - F);
	v3 o;
	float w = 1.0f, dwh_dwo;
	if (reflect) {
		o = vsub(vscale(m, 2.0f * wim), wi);
		dwh_dwo = 1.0f / (4.0f * dot3(o, m));
		*eta_out = 1.0f;
	} else {
		o = vsub(vscale(m, wim * eta_ti + cos_t), vscale(wi, eta_ti)); /* refract(wi, m, cos_theta_t, eta_ti) */
		w = eta_ti * eta_ti;
		const float om = dot3(o, m), denom = wim + eta_it * om;
		dwh_dwo = ((eta_it * eta_it) * om) / (denom * denom);
		*eta_out = eta_it;
	}
	w = w * rc_G1(o, m, alpha);
	p = p * fabsf(dwh_dwo);
	if (!(p == p) || !(w == w)) return;
	*wo = o;
	*pdf = p;
	*weight = V(w, w, w);
}

/* BSDF flags: does the material have a non-delta lobe (BSDFFlags.Smooth, :210)? */
static inline int material_is_smooth(const material *mt) { return mt->type != 2 && mt->type != 3; }

/* bsdf.eval_pdf (twosided unless the material says otherwise): value includes cos(theta_o) */
static inline void bsdf_eval_pdf(const material *mt, v3 wi, v3 wo, int active, v3 *value, float *pdf)
{
	*value = V(0, 0, 0);
	*pdf = 0.0f;
	if (!active) return;
	if (mt->type == 2 || mt->type == 3) return; /* smooth conductor / dielectric: delta lobes only */
	if (mt->type == 4) {
		rd_eval_pdf(mt, wi, wo, value, pdf);
		return;
	}
	if (wi.z < 0.0f && !mt->one_sided) { wi.z = -wi.z; wo.z = -wo.z; }
	if (mt->type == 1) {
		rc_eval_pdf(mt, wi, wo, value, pdf);
		return;
	}
	const v3 refl = mt->refl;
	if (!(wi.z > 0.0f && wo.z > 0.0f)) return;
	*value = vscale(vscale(refl, INV_PI_F), wo.z);
	*pdf = INV_PI_F * wo.z;
}

/* bsdf.sample(ctx, si, u1, (u, v)): wo (local), pdf, weight = value/pdf, relative index along the
 * sampled direction, and whether a delta lobe was sampled (BSDFFlags.Delta, :282) */
static inline void bsdf_sample(const material *mt, v3 wi, float u1, float u, float v, int active, v3 *wo, float *pdf,
                               v3 *weight, float *eta, int *delta)
{
	*wo = V(0, 0, 0); *pdf = 0.0f; *weight = V(0, 0, 0); *eta = 0.0f; *delta = 0;
	if (!active) return;
	if (mt->type == 3) { /* smooth dielectric (dielectric.cpp), radiance transport */
		float cos_t, eta_it, eta_ti;
		const float r_i = fresnel_dielectric(wi.z, mt->eta.x, &cos_t, &eta_it, &eta_ti);
		const int reflect = u1 <= r_i;
		*delta = 1;
		*pdf = reflect ? r_i : 1.0f - r_i;
		*wo = reflect ? V(-wi.x, -wi.y, wi.z) : V(-eta_ti * wi.x, -eta_ti * wi.y, cos_t);
		*eta = reflect ? 1.0f : eta_it;
		*weight = reflect ? V(1, 1, 1) : V(eta_ti * eta_ti, eta_ti * eta_ti, eta_ti * eta_ti);
		return;
	}
	if (mt->type == 4) {
		rd_sample(mt, wi, u1, u, v, wo, pdf, weight, eta);
		return;
	}
	int flip = wi.z < 0.0f && !mt->one_sided;
	float cos_i = flip ? -wi.z : wi.z;
	if (!(cos_i > 0.0f)) return;
	if (mt->type == 2) { /* smooth conductor (conductor.cpp): the mirror direction, weighted by Fresnel */
		*delta = 1;
		*pdf = 1.0f;
		*eta = 1.0f;
		*wo = V(-wi.x, -wi.y, wi.z);
		*weight = vmul(rc_fresnel(mt, cos_i), mt->refl);
		return;
	}
	if (mt->type == 1) {
		v3 o;
		rc_sample(mt, V(wi.x, wi.y, cos_i), u, v, &o, pdf, weight);
		*eta = 1.0f;
		if (flip) o.z = -o.z;
		*wo = o;
		return;
	}
	const v3 refl = mt->refl;
	v3 w = square_to_cosine_hemisphere(u, v);
	float p = INV_PI_F * w.z;
	*eta = 1.0f;
	*pdf = p;
	if (p > 0.0f) *weight = refl;
	if (flip) w.z = -w.z;
	*wo = w;
}

void pgo_bsdf_eval_pdf(const float *m, const float wi[3], const float wo[3], float value[3], float *pdf)
{
	const material mt = load_material(m);
	v3 val;
	bsdf_eval_pdf(&mt, ld3(wi), ld3(wo), 1, &val, pdf);
	value[0] = val.x; value[1] = val.y; value[2] = val.z;
}

void pgo_bsdf_sample(const float *m, const float wi[3], float u1, float u2, float wo[3], float *pdf, float weight[3])
{
	float eta;
	int delta;
	pgo_bsdf_sample_full(m, wi, 0.5f, u1, u2, wo, pdf, weight, &eta, &delta);
}

void pgo_bsdf_sample_full(const float *m, const float wi[3], float lobe, float u1, float u2, float wo[3], float *pdf,
                          float weight[3], float *eta_out, int *delta_out)
{
	const material mt = load_material(m);
	v3 o, w;
	float eta;
	int delta;
	bsdf_sample(&mt, ld3(wi), lobe, u1, u2, 1, &o, pdf, &w, &eta, &delta);
	*eta_out = eta;
	*delta_out = delta;
	wo[0] = o.x; wo[1] = o.y; wo[2] = o.z;
	weight[0] = w.x; weight[1] = w.y; weight[2] = w.z;
}

void pgo_render_pass(const pgo_tree *prev, pgo_tree *current, size_t nq, const float *quads,
                     const pgo_camera *cam, const pgo_render_params *prm, float *L_out, uint8_t *valid_out,
                     float *sumL, float *sumL2)
{
	pgo_scene sc;
	memset(&sc, 0, sizeof sc);
	sc.n_quads = nq;
	sc.quads = quads;
	pgo_render_pass_scene(prev, current, &sc, cam, prm, L_out, valid_out, sumL, sumL2);
}

void pgo_render_pass_scene(const pgo_tree *prev, pgo_tree *current, const pgo_scene *sc,
                           const pgo_camera *cam, const pgo_render_params *prm, float *L_out, uint8_t *valid_out,
                           float *sumL, float *sumL2)
{
	/* emitters: flagged quads, then flagged spheres */
	int *em = malloc((sc->n_quads + sc->n_spheres + sc->n_dir_lights + 1) * sizeof(int));
	int n_em = 0;
	for (size_t q = 0; q < sc->n_quads; ++q)
		if (sc->quads[q * PGO_QUAD_STRIDE + 15] != 0.0f) em[n_em++] = (int)q;
	for (size_t s = 0; s < sc->n_spheres; ++s)
		if (sc->spheres[s * PGO_SPHERE_STRIDE + 5] != 0.0f) em[n_em++] = (int)(sc->n_quads + s);
	for (size_t k = 0; k < sc->n_dir_lights; ++k) em[n_em++] = -1 - (int)k; /* directional lights: negative codes */
	const float inv_em_count = n_em > 0 ? 1.0f / (float)n_em : 0.0f;
	const int W = cam->width, H = cam->height, spp = prm->spp, D = prm->max_depth;
	const size_t npix = (size_t)W * H, N = npix * (size_t)spp, S = N * (size_t)(D > 0 ? D : 1);
	const float f = prm->bsdf_sampling_fraction;
	const int guided = prm->iteration > 1; /* :223, 250, 283 */
	const int record = !prm->is_final;
	/* dense record buffer, slot = ray*max_depth + depth (:318), zero-filled every pass (:155) */
	uint8_t *r_act = NULL;
	float *r_pos = NULL, *r_dir = NULL, *r_bsdf = NULL, *r_tb = NULL, *r_tr = NULL, *r_nee = NULL, *r_dnee = NULL, *r_wp = NULL;
	if (record) {
		r_act = calloc(S, 1);
		r_pos = calloc(3 * S, 4); r_dir = calloc(2 * S, 4); r_bsdf = calloc(3 * S, 4); r_tb = calloc(3 * S, 4);
		r_tr = calloc(3 * S, 4); r_nee = calloc(3 * S, 4); r_dnee = calloc(2 * S, 4); r_wp = calloc(S, 4);
	}
	const float aspect_tan_y = cam->tan_half_fov_x / ((float)W / (float)H);
	/* lanes are independent (every lane has its own sampler stream, record slots and outputs): the
	 * loop may run on several threads without changing a bit of the result */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
#endif
	for (size_t lane = 0; lane < N; ++lane) {
		pgo_pcg32 rng;
		pgo_pcg32_seed(&rng, prm->seed, (uint32_t)lane);
		const size_t pixel = lane / (size_t)spp;
		const float px = (float)(pixel % (size_t)W), py = (float)(pixel / (size_t)W);
		/* camera ray: position sample next_2d */
		float jx = pgo_pcg32_next_f32(&rng), jy = pgo_pcg32_next_f32(&rng);
		float cx = (1.0f - 2.0f * ((px + jx) / (float)W)) * cam->tan_half_fov_x;
		float cy = (1.0f - 2.0f * ((py + jy) / (float)H)) * aspect_tan_y;
		float len = sqrtf((cx * cx + cy * cy) + 1.0f);
		v3 dc = V(cx / len, cy / len, 1.0f / len);
		v3 ray_o = ld3(cam->origin);
		v3 ray_d = vadd(vadd(vscale(ld3(cam->axis_x), dc.x), vscale(ld3(cam->axis_y), dc.y)), vscale(ld3(cam->axis_z), dc.z));
		v3 thr = V(1, 1, 1), L = V(0, 0, 0);
		uint32_t depth = 0;
		float ior = 1.0f;
		int active = 1;
		v3 prev_p = V(0, 0, 0);
		float prev_bsdf_pdf = 1.0f;
		int prev_delta = 1;
		for (int it = 0; it < D && active; ++it) {
			/* ---- :185 ray_intersect ---- */
			float t_hit;
			int q = intersect(sc, ray_o, ray_d, INFINITY, &t_hit);
			int valid = q >= 0;
			surface sf;
			memset(&sf, 0, sizeof sf);
			sf.n = V(0, 0, 1);
			if (valid) sf = surface_at(sc, q, ray_o, ray_d, t_hit);
			if (sf.ng.x == 0.0f && sf.ng.y == 0.0f && sf.ng.z == 0.0f) sf.ng = sf.n; /* every shape but a smooth-shaded triangle */
			const v3 p = sf.p, n = sf.n, ng = sf.ng;
			const material *mt = &sf.m;
			frame fr;
			fr.n = n;
			frame_from_normal(n, &fr.s, &fr.t);
			v3 wi = to_local(&fr, V(-ray_d.x, -ray_d.y, -ray_d.z));
			int is_em = valid && sf.is_em;
			/* ---- :189-200 direct emission ---- */
			v3 em_radiance = (is_em && wi.z > 0.0f) ? sf.radiance : V(0, 0, 0);
			float emitter_pdf = 0.0f;
			if (is_em && !prev_delta) emitter_pdf = emitter_hit_pdf(sc, q, prev_p, p, n, inv_em_count);
			float mis = mis_weight(prev_bsdf_pdf, emitter_pdf);
			v3 Le = vmul(vscale(thr, mis), em_radiance);
			/* ---- :207-220 emitter sampling ---- */
			int active_next = (depth + 1 < (uint32_t)D) && valid;
			int active_em = active_next && material_is_smooth(mt); /* :210 BSDFFlags.Smooth */
			float e1 = pgo_pcg32_next_f32(&rng), e2 = pgo_pcg32_next_f32(&rng); /* :214 next_2d, unmasked */
			v3 ds_d = V(0, 0, 0), em_weight = V(0, 0, 0);
			float ds_pdf = 0.0f;
			int ds_delta = 0;
			if (active_em) sample_emitter(sc, em, n_em, p, ng, e1, e2, &ds_d, &ds_pdf, &em_weight, &ds_delta);
			active_em = active_em && (ds_pdf != 0.0f); /* :216 */
			v3 wo_em = to_local(&fr, ds_d);
			v3 bsdf_value_em;
			float bsdf_pdf_em;
			bsdf_eval_pdf(mt, wi, wo_em, active_em, &bsdf_value_em, &bsdf_pdf_em);
			/* ---- :223-256 NEE MIS against the mixture pdf ---- */
			int active_sd_em = active_em && guided;
			float pdf_diffuse = 1.0f; /* :222-241, SURVEY A12 */
			float pp[3] = { p.x, p.y, p.z };
			uint32_t tree = 0;
			int tree_known = 0;
			float sdtree_pdf_em = 1.0f;
			if (active_sd_em) {
				tree = pgo_i_quadtree_of(prev, pp, 1);
				tree_known = 1;
				float dv[3] = { ds_d.x, ds_d.y, ds_d.z };
				sdtree_pdf_em = pgo_i_pdf(prev, tree, dv, 1);
			}
			float surface_pdf_em = f * bsdf_pdf_em + ((1.0f - f) * sdtree_pdf_em) * pdf_diffuse;
			if (!guided) surface_pdf_em = bsdf_pdf_em;
			float mis_em = ds_delta ? 1.0f : mis_weight(ds_pdf, surface_pdf_em); /* :253 */
			v3 Lr_dir = vmul(vmul(vscale(thr, mis_em), bsdf_value_em), em_weight);
			L = vadd(L, vadd(Le, Lr_dir)); /* :261 */
			/* ---- :272-311 next direction ---- */
			float s1 = 0.0f, s2x = 0.0f, s2y = 0.0f;
			if (active_next) { s1 = pgo_pcg32_next_f32(&rng); s2x = pgo_pcg32_next_f32(&rng); s2y = pgo_pcg32_next_f32(&rng); }
			v3 wo_local, bsdf_weight;
			float bsdf_pdf, eta;
			int delta;
			bsdf_sample(mt, wi, s1, s2x, s2y, active_next, &wo_local, &bsdf_pdf, &bsdf_weight, &eta, &delta);
			v3 bsdf_value = vscale(bsdf_weight, bsdf_pdf);
			float woPdf = bsdf_pdf;
			v3 wo_world = to_world(&fr, wo_local);
			int do_mis = active_next && !delta && guided;
			int pick_tree = 0;
			if (active_next) pick_tree = pgo_pcg32_next_f32(&rng) > f; /* :286 */
			int smp_tree = pick_tree && do_mis;
			int bsdf_mis = do_mis && !smp_tree;
			float sdtree_pdf = 1.0f;
			if (smp_tree) { /* :301-304 */
				if (!tree_known) { tree = pgo_i_quadtree_of(prev, pp, 1); tree_known = 1; }
				float dv[3];
				pgo_i_sample(prev, tree, &rng.state, rng.inc, 1, dv);
				sdtree_pdf = pgo_i_pdf(prev, tree, dv, 1);
				wo_world = V(dv[0], dv[1], dv[2]);
				wo_local = to_local(&fr, wo_world);
				bsdf_eval_pdf(mt, wi, wo_local, 1, &bsdf_value, &bsdf_pdf);
			}
			if (bsdf_mis) { /* :307 */
				if (!tree_known) { tree = pgo_i_quadtree_of(prev, pp, 1); tree_known = 1; }
				float dv[3] = { wo_world.x, wo_world.y, wo_world.z };
				sdtree_pdf = pgo_i_pdf(prev, tree, dv, 1);
			}
			if (do_mis) { /* :310-311 */
				woPdf = f * bsdf_pdf + (1.0f - f) * sdtree_pdf;
				bsdf_weight = vdivs(bsdf_value, woPdf);
				/* DELIBERATE DEVIATION: a direction sampled from a zero-energy quadtree
				 * (pdf 0, quadtree.py:1086-1092) below the surface has bsdf_pdf = 0 too, and the
				 * reference then divides 0/0 -- its throughput turns NaN and poisons the pixel
				 * (about 1e-7 of the paths on cornell-box).  Such a path carries no energy: stop it. */
				if (!(woPdf > 0.0f)) bsdf_weight = V(0, 0, 0);
			}
			/* ---- :318-346 record ---- */
			if (record && active && valid) {
				size_t g = lane * (size_t)D + depth;
				float c[2];
				r_act[g] = 1;
				r_pos[g] = p.x; r_pos[S + g] = p.y; r_pos[2 * S + g] = p.z;
				pgo_dir_to_canonical(wo_world.x, wo_world.y, wo_world.z, c);
				r_dir[g] = c[0]; r_dir[S + g] = c[1];
				r_bsdf[g] = bsdf_weight.x; r_bsdf[S + g] = bsdf_weight.y; r_bsdf[2 * S + g] = bsdf_weight.z;
				r_tb[g] = thr.x; r_tb[S + g] = thr.y; r_tb[2 * S + g] = thr.z;
				r_tr[g] = L.x; r_tr[S + g] = L.y; r_tr[2 * S + g] = L.z;
				if (prm->store_nee) {
					v3 rn = vdiv(Lr_dir, thr);
					r_nee[g] = rn.x; r_nee[S + g] = rn.y; r_nee[2 * S + g] = rn.z;
					pgo_dir_to_canonical(ds_d.x, ds_d.y, ds_d.z, c);
					r_dnee[g] = c[0]; r_dnee[S + g] = c[1];
				}
				r_wp[g] = woPdf;
			}
			/* ---- :352-381 advance ---- */
			{
				float mag = (1.0f + max3(V(fabsf(p.x), fabsf(p.y), fabsf(p.z)))) * RAY_EPS_F;
				if (dot3(ng, wo_world) < 0.0f) mag = -mag;
				ray_o = vadd(p, vscale(ng, mag));
				ray_d = wo_world;
			}
			ior = ior * eta;
			thr = vmul(thr, bsdf_weight);
			prev_p = p;
			prev_bsdf_pdf = woPdf;
			prev_delta = delta;
			float tmax = max3(thr);
			active_next = active_next && (tmax != 0.0f);
			float rr_prob = tmax * (ior * ior);
			if (!(rr_prob < 0.95f)) rr_prob = 0.95f; /* dr.minimum(x, 0.95): NaN -> 0.95 */
			int rr_active = depth >= (uint32_t)prm->rr_depth;
			float rr = pgo_pcg32_next_f32(&rng); /* :377, unmasked */
			int rr_continue = rr < rr_prob;
			active_next = active_next && (!rr_active || rr_continue);
			active = active_next;
			if (valid) depth += 1;
		}
		L_out[lane] = L.x; L_out[N + lane] = L.y; L_out[2 * N + lane] = L.z;
		valid_out[lane] = depth != 0;
	}
	/* ---- :388-395 ---- */
	if (record) {
		float *o_pos = malloc(3 * S * 4), *o_dir = malloc(2 * S * 4), *o_rad = malloc(S * 4), *o_wp = malloc(S * 4);
		float *o_dnee = malloc(2 * S * 4), *o_nl = malloc(S * 4);
		size_t kept = pgo_process_records(N, (size_t)D, L_out, r_act, r_pos, r_dir, r_bsdf, r_tb, r_tr, r_nee, r_dnee,
		                                  r_wp, o_pos, o_dir, o_rad, o_wp, o_dnee, o_nl);
		if (kept) {
			/* planes have stride S: repack to stride `kept` */
			float *c_pos = malloc(3 * kept * 4), *c_dir = malloc(2 * kept * 4), *c_dnee = malloc(2 * kept * 4);
			for (int a = 0; a < 3; ++a) memcpy(c_pos + a * kept, o_pos + a * S, kept * 4);
			for (int a = 0; a < 2; ++a) { memcpy(c_dir + a * kept, o_dir + a * S, kept * 4); memcpy(c_dnee + a * kept, o_dnee + a * S, kept * 4); }
			pgo_add_data_propagate(current, kept, c_pos, c_dir, o_rad, o_wp, c_dnee, o_nl);
			free(c_pos); free(c_dir); free(c_dnee);
		}
		free(o_pos); free(o_dir); free(o_rad); free(o_wp); free(o_dnee); free(o_nl);
		free(r_act); free(r_pos); free(r_dir); free(r_bsdf); free(r_tb); free(r_tr); free(r_nee); free(r_dnee); free(r_wp);
	}
	free(em);
	/* ---- :400-429 per-pixel sums, samples of a pixel in lane order ---- */
	if (sumL && sumL2) {
		for (size_t pix = 0; pix < npix; ++pix)
			for (int s = 0; s < spp; ++s) {
				size_t lane = pix * (size_t)spp + (size_t)s;
				for (int ch = 0; ch < 3; ++ch) {
					float v = L_out[(size_t)ch * N + lane];
					sumL[(size_t)ch * npix + pix] += v;
					sumL2[(size_t)ch * npix + pix] += v * v;
				}
			}
	}
}

static float tent1(float d)
{
	float a = 1.0f - fabsf(d);
	return a > 0.0f ? a : 0.0f;
}

/* hdrfilm with <rfilter type="tent"/>: gather form of ImageBlock::put -- pixel (x,y) collects the
 * samples of its 3x3 neighbourhood (rows, then columns, then samples, in ascending order) */
void pgo_film_tent(uint32_t seed, int32_t spp, int32_t width, int32_t height, const float *L, float *image_out)
{
	pgo_film(0, seed, spp, width, height, L, image_out);
}

/* Mitsuba's gaussian rfilter (hdrfilm's default; scenes/torus/scene.xml:46): stddev 0.5, radius
 * 4 stddev = 2, w(d) = max(0, exp(-d^2 / (2 stddev^2)) - exp(-radius^2 / (2 stddev^2))) */
static float gauss1(float d)
{
	const float a = pgo_exp(-2.0f * (d * d)) - pgo_exp(-8.0f);
	return a > 0.0f ? a : 0.0f;
}

void pgo_film(int32_t filter, uint32_t seed, int32_t spp, int32_t width, int32_t height, const float *L, float *image_out)
{
	const size_t npix = (size_t)width * (size_t)height, N = npix * (size_t)spp;
	const int32_t R = filter == 1 ? 2 : 1; /* pixels a sample can reach on either side */
	for (int32_t y = 0; y < height; ++y)
		for (int32_t x = 0; x < width; ++x) {
			float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
			const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
			for (int32_t ny = y - R; ny <= y + R; ++ny)
				for (int32_t nx = x - R; nx <= x + R; ++nx) {
					if (nx < 0 || ny < 0 || nx >= width || ny >= height) continue;
					const size_t pix = (size_t)ny * (size_t)width + (size_t)nx;
					for (int32_t s = 0; s < spp; ++s) {
						const size_t lane = pix * (size_t)spp + (size_t)s;
						pgo_pcg32 rng;
						pgo_pcg32_seed(&rng, seed, (uint32_t)lane);
						const float jx = pgo_pcg32_next_f32(&rng), jy = pgo_pcg32_next_f32(&rng);
						const float ddx = cx - ((float)nx + jx), ddy = cy - ((float)ny + jy);
						const float w = filter == 1 ? gauss1(ddx) * gauss1(ddy) : tent1(ddx) * tent1(ddy);
						acc[0] = acc[0] + w * L[lane];
						acc[1] = acc[1] + w * L[N + lane];
						acc[2] = acc[2] + w * L[2 * N + lane];
						wsum = wsum + w;
					}
				}
			const size_t o = (size_t)y * (size_t)width + (size_t)x;
			for (int ch = 0; ch < 3; ++ch) image_out[(size_t)ch * npix + o] = wsum > 0.0f ? acc[ch] / wsum : 0.0f;
		}
}
