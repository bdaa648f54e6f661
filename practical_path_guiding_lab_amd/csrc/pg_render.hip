// pg_render.hip -- wavefront renderer substrate around the SD-tree: what the reference's
// PathGuidingIntegrator.sample() (src/path_guiding_integrator.py:126-431) does per pass, with the
// Mitsuba calls it makes (scene.ray_intersect :185, emitter eval/pdf :189-198,
// sample_emitter_direction :213, bsdf.eval_pdf/sample :220, 272, 304, si.to_local/to_world/spawn_ray
// :219, 277, 352, sampler.next_1d/2d) implemented for the scene subset of scenes/cornell-box
// (quads, twosided diffuse BSDFs, a one-sided area emitter, perspective camera) and, in the
// kGeneral instantiations, of scenes/veach-mis (spheres, several emitters, Beckmann rough conductors).
//
// One kernel per bounce over the live rays; the SD-tree queries are the same device functions the
// stand-alone query kernels use, so a bounce costs one KD descent and at most two quadtree descents
// per live lane and no intermediate wavefront buffers.  The first bounce generates its camera rays
// itself, every bounce appends its survivors to the live-ray list of the next one (one atomic per
// workgroup), and a path carries 57 B of state between bounces.  Path-vertex records go to a dense
// slot buffer (the reference's :318, stored depth-major here so that a wavefront's stores coalesce)
// that pg_process_and_splat consumes after the last bounce (:388-395).
//
// Arithmetic mirrors oracle/pg_oracle_render.c operation by operation (fp32, no contraction), so
// radiance, records and therefore the refined trees are bit-identical to the CPU restatement.
#include "pg_context.hpp"
#include "pg_descent.hpp"
#include "pg_kernels.hpp"

namespace pg {

constexpr int kRBlock = 256;
constexpr float kInvPiF = 0.31830988618379067154f;
constexpr float kRayEps = 1e-4f;
constexpr float kShadowEps = 1e-3f;
constexpr int kQuadStride = 24;

struct v3 {
	float x, y, z;
};
__device__ __forceinline__ v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
__device__ __forceinline__ v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
__device__ __forceinline__ float max3(v3 a) { const float m = a.x > a.y ? a.x : a.y; return m > a.z ? m : a.z; }
__device__ __forceinline__ float fabs_(float v) { return __builtin_fabsf(v); }

// path_guiding_integrator.py:16-24
__device__ __forceinline__ float mis_weight(float a, float b)
{
	const float a2 = a * a;
	float r = a > 0.0f ? a2 / (b * b + a2) : 0.0f;
	if (r != r) r = 0.0f;
	return r;
}

struct Frame {
	v3 s, t, n;
};
// Mitsuba coordinate_system(n) (Duff et al. 2017)
__device__ __forceinline__ Frame make_frame(v3 n)
{
	const float sign = (__float_as_uint(n.z) >> 31) ? -1.0f : 1.0f;
	const float a = -1.0f / (sign + n.z);
	const float b = (n.x * n.y) * a;
	Frame f;
	f.n = n;
	f.s = V(1.0f + (sign * (n.x * n.x)) * a, sign * b, -sign * n.x);
	f.t = V(b, sign + (n.y * n.y) * a, -n.y);
	return f;
}
__device__ __forceinline__ v3 to_local(const Frame &f, v3 v) { return V(dot3(v, f.s), dot3(v, f.t), dot3(v, f.n)); }
__device__ __forceinline__ v3 to_world(const Frame &f, v3 v)
{
	return vadd(vadd(vscale(f.s, v.x), vscale(f.t, v.y)), vscale(f.n, v.z));
}

constexpr float kPiF = 3.14159265358979323846f;
constexpr float kInvTwoPiF = 0.15915494309189533577f;
constexpr float kInvSqrtPiF = 0.56418958354775628695f;
constexpr float kSphereEps = 8.94069671630859375e-05f; // Mitsuba's math::RayEpsilon<float> = 1500 * 2^-24
constexpr int kSphereStride = 12;                       // PG_SPHERE_STRIDE
constexpr int kMaterialStride = 12;                     // PG_MATERIAL_STRIDE

__device__ __forceinline__ float safe_sqrtf(float v) { return __builtin_sqrtf(v > 0.0f ? v : 0.0f); }
__device__ __forceinline__ v3 normalize3(v3 v) { return vdivs(v, __builtin_sqrtf(dot3(v, v))); }

constexpr int kBoxStride = 32; // PG_BOX_STRIDE

// The shapes of a scene: quads, then spheres, then box faces (shape number = quad index,
// n_quads + sphere index, or n_quads + n_spheres + 6 box + 2 axis + (outward normal negative))
// ... then the triangles of the meshes, in BVH leaf order (general scenes only)
constexpr int kTriStride = 16; // PG_TRI_STRIDE
constexpr int kBvhStride = 32; // PG_BVH_STRIDE
struct Shapes {
	const float *quads, *spheres, *boxes, *tris;
	const float *tri_normals; // 9 per triangle, or nullptr (face normals)
	const uint32_t *bvh;
	int n_quads, n_spheres, n_boxes, n_bvh_nodes;
};

__device__ __forceinline__ v3 box_face_normal(const float *B, int face)
{
	const v3 n = ld3(B + 12 + 3 * (face >> 1));
	return (face & 1) ? V(-n.x, -n.y, -n.z) : n;
}

// closest hit over all shapes, 0 < t < tmax (scene.ray_intersect / ray_test).  kGeneral is the
// feature level the kernel is compiled for: 0 = quads and boxes with twosided diffuse BSDFs
// (cornell-box), 1 = + spheres and rough conductors (veach-mis), 2 = + triangle meshes, delta
// lobes, one-sided BSDFs, directional lights and the running index of refraction (torus-class
// scenes).  What a level does not need is compiled out.
// Does the ray reach the box [lo, hi] before bt?  tmin = where it enters (>= 0).  Slab test padded
// as Ize 2013.  The plane a ray meets first on an axis is known from the sign of its direction
// (neg: sign bits of d, once per ray), and fmaxf/fminf (v_max3/v_min3: a NaN operand -- 0 * inf, the
// ray lies in a face's plane -- is ignored, which keeps the test conservative) fold the three axes.
__device__ __forceinline__ bool bvh_box_hit(float lox, float loy, float loz, float hix, float hiy, float hiz, v3 o,
                                            v3 inv, bool negx, bool negy, bool negz, float bt, float &tmin_out)
{
	const float nx = ((negx ? hix : lox) - o.x) * inv.x, fx = ((negx ? lox : hix) - o.x) * inv.x;
	const float ny = ((negy ? hiy : loy) - o.y) * inv.y, fy = ((negy ? loy : hiy) - o.y) * inv.y;
	const float nz = ((negz ? hiz : loz) - o.z) * inv.z, fz = ((negz ? loz : hiz) - o.z) * inv.z;
	const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx, ny), nz), 0.0f);
	const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fminf(fx, fy), fz), bt);
	tmin_out = tmin;
	return tmin <= tmax * 1.0000004f;
}

__device__ __forceinline__ void bvh_cswap(float &ta, uint32_t &ra, float &tb, uint32_t &rb)
{
	if (ta > tb) {
		const float t = ta; ta = tb; tb = t;
		const uint32_t r = ra; ra = rb; rb = r;
	}
}

// kAny: the caller asks whether anything is hit (shadow rays): the BVH walk stops at its first
// triangle.  The answer is that of the closest-hit walk, which visits the same nodes until then.
template <int kGeneral, bool kAny = false>
__device__ __forceinline__ int intersect(const Shapes &sh, v3 o, v3 d, float tmax, float &t_out)
{
	const int nq = sh.n_quads;
	const float *__restrict__ quads = sh.quads;
	int best = -1;
	float bt = tmax;
	for (int q = 0; q < nq; ++q) {
		const float *Q = quads + q * kQuadStride;
		const v3 n = ld3(Q + 9);
		const float denom = dot3(n, d);
		if (denom == 0.0f) continue;
		const float num = dot3(n, vsub(ld3(Q), o));
		// IEEE division keeps the sign: when the signs differ t is not > 0 and the (correctly rounded,
		// hence long) division can be skipped without changing any result
		if ((__float_as_uint(num) ^ __float_as_uint(denom)) >> 31) continue;
		const float t = num / denom;
		if (!(t > 0.0f && t < bt)) continue;
		const v3 w = vsub(vadd(o, vscale(d, t)), ld3(Q));
		const float u = dot3(w, ld3(Q + 3)) * Q[12];
		const float v = dot3(w, ld3(Q + 6)) * Q[13];
		if (u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f) { bt = t; best = q; }
	}
	if (kGeneral) { // spheres: the quadratic in double precision, as Mitsuba's Sphere::ray_intersect_preliminary
		for (int s = 0; s < sh.n_spheres; ++s) {
			const float *S = sh.spheres + s * kSphereStride;
			const double ox = (double)o.x - (double)S[0], oy = (double)o.y - (double)S[1], oz = (double)o.z - (double)S[2];
			const double dx = (double)d.x, dy = (double)d.y, dz = (double)d.z, r = (double)S[3];
			const double A = (dx * dx + dy * dy) + dz * dz;
			const double B = 2.0 * ((ox * dx + oy * dy) + oz * dz);
			const double C = ((ox * ox + oy * oy) + oz * oz) - r * r;
			const double disc = B * B - (4.0 * A) * C;
			if (!(disc >= 0.0) || A == 0.0) continue;
			const double root = __builtin_sqrt(disc);
			const double temp = -0.5 * (B + (B < 0.0 ? -root : root)); // the cancellation-free root first
			double x0 = temp / A, x1 = temp != 0.0 ? C / temp : x0;
			if (x0 > x1) { const double tt = x0; x0 = x1; x1 = tt; }
			const float t = (float)(x0 > 0.0 ? x0 : x1);
			if (t > 0.0f && t < bt) { bt = t; best = nq + s; }
		}
	}
	// boxes (Mitsuba `cube` shapes): three slabs in the box's local frame, one reciprocal per axis,
	// instead of six quad tests (t is the same in both frames: the map is linear)
	for (int b = 0; b < sh.n_boxes; ++b) {
		const float *B = sh.boxes + b * kBoxStride;
		const v3 oc = vsub(o, ld3(B + 9));
		const float ol[3] = {dot3(ld3(B), oc), dot3(ld3(B + 3), oc), dot3(ld3(B + 6), oc)};
		const float dl[3] = {dot3(ld3(B), d), dot3(ld3(B + 3), d), dot3(ld3(B + 6), d)};
		float tn = -__builtin_huge_valf(), tf = __builtin_huge_valf();
		int an = 0, af = 0;
		bool miss = false;
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			if (dl[k] == 0.0f) { // parallel to this slab: inside it or never
				if (!(ol[k] >= -1.0f && ol[k] <= 1.0f)) miss = true;
				continue;
			}
			const float inv = 1.0f / dl[k];
			const float t1 = (-1.0f - ol[k]) * inv, t2 = (1.0f - ol[k]) * inv;
			const float lo = dl[k] > 0.0f ? t1 : t2, hi = dl[k] > 0.0f ? t2 : t1;
			if (lo > tn) { tn = lo; an = k; }
			if (hi < tf) { tf = hi; af = k; }
		}
		if (miss || !(tn <= tf)) continue;
		const bool entering = tn > 0.0f;
		const float t = entering ? tn : tf;
		if (!(t > 0.0f && t < bt)) continue;
		const int axis = entering ? an : af;
		const float da = axis == 0 ? dl[0] : (axis == 1 ? dl[1] : dl[2]);
		const int negative = entering ? (da > 0.0f) : (da < 0.0f);
		bt = t;
		best = nq + sh.n_spheres + 6 * b + 2 * axis + negative;
	}
	// triangle meshes: the four-wide BVH.  One 128-byte node holds the boxes of its (up to four)
	// children: they are tested together and ordered by where the ray enters them (a fixed
	// five-comparator network), the walk goes on in the nearest -- a leaf's triangles are named by the
	// reference itself, no node is read for it -- and the others wait on the stack with their entry
	// distance, farthest at the bottom, to be dropped when popped if the ray has become shorter than
	// that.  Half the dependent round trips of a binary tree: the walk is latency-bound.  The oracle
	// visits the same nodes in the same order, so the first of several equally near triangles is the
	// same one in both.
	if (kGeneral >= 2 && sh.n_bvh_nodes && !(kAny && best >= 0)) {
		const int tri_base = nq + sh.n_spheres + 6 * sh.n_boxes;
		const v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
		const bool ngx = (__float_as_uint(d.x) >> 31) != 0u, ngy = (__float_as_uint(d.y) >> 31) != 0u, ngz = (__float_as_uint(d.z) >> 31) != 0u;
		const uint4 *__restrict__ nodes = reinterpret_cast<const uint4 *>(sh.bvh);
		// pg_scene_set_ex has checked the tree: children follow their parent, and no root-to-node path
		// can leave more than 64 siblings waiting, so the walk opens every node at most once and the
		// stack cannot overflow; the budget is a second fence
		constexpr uint32_t kNone = 0xffffffffu;
		const float kInf = __builtin_huge_valf();
		uint32_t st_ref[64];
		float st_t[64];
		int sp = 0;
		int budget = 8 * sh.n_bvh_nodes + 8;
		uint32_t next = 0; // the root
		while (true) {
			while (!(next & 0x80000000u) && budget > 0) { // a node: test its children, go on in the nearest
				const uint4 *N = nodes + 8 * (size_t)next;
				const uint4 lx = N[0], ly = N[1], lz = N[2], hx = N[3], hy = N[4], hz = N[5], rf = N[6];
				uint32_t r0 = rf.x, r1 = rf.y, r2 = rf.z, r3 = rf.w;
				float t0, t1, t2, t3;
#define PG_F(v) __uint_as_float(v)
				if (!(r0 != kNone && bvh_box_hit(PG_F(lx.x), PG_F(ly.x), PG_F(lz.x), PG_F(hx.x), PG_F(hy.x), PG_F(hz.x), o, inv, ngx, ngy, ngz, bt, t0))) { r0 = kNone; t0 = kInf; }
				if (!(r1 != kNone && bvh_box_hit(PG_F(lx.y), PG_F(ly.y), PG_F(lz.y), PG_F(hx.y), PG_F(hy.y), PG_F(hz.y), o, inv, ngx, ngy, ngz, bt, t1))) { r1 = kNone; t1 = kInf; }
				if (!(r2 != kNone && bvh_box_hit(PG_F(lx.z), PG_F(ly.z), PG_F(lz.z), PG_F(hx.z), PG_F(hy.z), PG_F(hz.z), o, inv, ngx, ngy, ngz, bt, t2))) { r2 = kNone; t2 = kInf; }
				if (!(r3 != kNone && bvh_box_hit(PG_F(lx.w), PG_F(ly.w), PG_F(lz.w), PG_F(hx.w), PG_F(hy.w), PG_F(hz.w), o, inv, ngx, ngy, ngz, bt, t3))) { r3 = kNone; t3 = kInf; }
#undef PG_F
				bvh_cswap(t0, r0, t1, r1);
				bvh_cswap(t2, r2, t3, r3);
				bvh_cswap(t0, r0, t2, r2);
				bvh_cswap(t1, r1, t3, r3);
				bvh_cswap(t1, r1, t2, r2);
				if (r3 != kNone) { st_ref[sp] = r3; st_t[sp] = t3; ++sp; }
				if (r2 != kNone) { st_ref[sp] = r2; st_t[sp] = t2; ++sp; }
				if (r1 != kNone) { st_ref[sp] = r1; st_t[sp] = t1; ++sp; }
				next = r0;
				--budget;
			}
			if (next != kNone && (next & 0x80000000u)) { // a leaf
				const uint32_t first = next & 0x0fffffffu, count = ((next >> 28) & 7u) + 1u;
				for (uint32_t i = first; i < first + count; ++i) {
					const float *T = sh.tris + (size_t)i * kTriStride;
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
					if (t > 0.0f && t < bt) { bt = t; best = tri_base + (int)i; }
				}
				if (kAny && best >= 0) break; // a shadow ray needs one occluder, not the nearest
			}
			next = kNone;
			while (sp && next == kNone && budget > 0) { // the nearest waiting child the (now shorter) ray still reaches
				--sp;
				if (st_t[sp] <= bt * 1.0000004f) next = st_ref[sp];
				--budget;
			}
			if (next == kNone) break;
		}
	}
	t_out = bt;
	return best;
}

// ---- surface description at a hit ----
struct Material {
	int type;        // 0 diffuse, 1 roughconductor (Beckmann, visible normals), 2 smooth conductor, 3 smooth dielectric, 4 roughdielectric
	v3 refl;         // reflectance | specular_reflectance
	const float *M;  // the material row: alpha, eta, k are read where the BSDF needs them
	bool one_sided;  // not wrapped in `twosided` (row word 11)
};
struct Surface {
	v3 p, n, radiance; // n: the normal of the shading frame
	v3 ng;             // geometric normal (ray offsets); differs from n on smooth-shaded triangles only
	bool is_em;
	Material m;
};

template <int kGeneral>
__device__ __forceinline__ Surface surface_at(const Shapes &sh, const float *mats, int prim, v3 o, v3 d, float t)
{
	Surface s;
	const float *M;
	if (kGeneral >= 2 && prim >= sh.n_quads + sh.n_spheres + 6 * sh.n_boxes) { // a mesh triangle (face normals)
		const size_t ti = (size_t)(prim - sh.n_quads - sh.n_spheres - 6 * sh.n_boxes);
		const float *T = sh.tris + ti * kTriStride;
		s.p = vadd(o, vscale(d, t));
		s.n = ld3(T + 9);
		s.ng = s.n;
		if (sh.tri_normals) { // interpolated vertex normals: the barycentrics of the hit, by the intersection's own formulas
			const v3 e1 = ld3(T + 3), e2 = ld3(T + 6);
			const v3 pp = V(d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x);
			const float inv_det = 1.0f / dot3(e1, pp);
			const v3 sv = vsub(o, ld3(T));
			const float u = dot3(sv, pp) * inv_det;
			const v3 qq = V(sv.y * e1.z - sv.z * e1.y, sv.z * e1.x - sv.x * e1.z, sv.x * e1.y - sv.y * e1.x);
			const float v = dot3(d, qq) * inv_det;
			const float *Nn = sh.tri_normals + ti * 9;
			const v3 ns = vadd(vadd(vscale(ld3(Nn), (1.0f - u) - v), vscale(ld3(Nn + 3), u)), vscale(ld3(Nn + 6), v));
			const float l2 = dot3(ns, ns);
			if (l2 > 0.0f) s.n = vdivs(ns, __builtin_sqrtf(l2));
		}
		s.is_em = false;
		s.radiance = V(0, 0, 0);
		M = mats + (int)T[12] * kMaterialStride;
	} else if (prim >= sh.n_quads + sh.n_spheres) { // a box face
		const int f = prim - sh.n_quads - sh.n_spheres;
		const float *B = sh.boxes + (f / 6) * kBoxStride;
		s.p = vadd(o, vscale(d, t));
		s.n = box_face_normal(B, f % 6);
		s.ng = s.n;
		s.is_em = false;
		s.radiance = V(0, 0, 0);
		M = mats + (int)B[21] * kMaterialStride;
	} else if (!kGeneral || prim < sh.n_quads) {
		const float *Q = sh.quads + prim * kQuadStride;
		s.p = vadd(o, vscale(d, t));
		s.n = ld3(Q + 9);
		s.ng = s.n;
		s.is_em = Q[15] != 0.0f;
		s.radiance = ld3(Q + 19);
		if (!kGeneral) { // all-diffuse quad scene: the reflectance sits in the quad itself (pg_scene_set keeps it there)
			s.m.type = 0;
			s.m.refl = ld3(Q + 16);
			s.m.M = nullptr;
			s.m.one_sided = false;
			return s;
		}
		M = mats + (int)Q[22] * kMaterialStride;
	} else {
		const float *S = sh.spheres + (prim - sh.n_quads) * kSphereStride;
		const v3 c = ld3(S);
		// sphere.h: n = normalize(ray(t) - c), p = c + n r; the normal is then taken again from the
		// re-projected point so that it is a function of p alone (the next bounce recomputes it)
		const v3 n0 = normalize3(vsub(vadd(o, vscale(d, t)), c));
		s.p = vadd(c, vscale(n0, S[3]));
		s.n = normalize3(vsub(s.p, c));
		s.ng = s.n;
		s.is_em = S[5] != 0.0f;
		s.radiance = ld3(S + 6);
		M = mats + (int)S[4] * kMaterialStride;
	}
	s.m.type = kGeneral ? (int)M[0] : 0; // a scene with anything but twosided diffuse runs the general kernels
	s.m.one_sided = kGeneral >= 2 && M[11] != 0.0f;
	s.m.refl = ld3(M + 1);
	s.m.M = M;
	return s;
}

// normal of shape `prim` at the surface point p (quads: constant; spheres: as surface_at defines it)
template <int kGeneral>
__device__ __forceinline__ v3 normal_at(const Shapes &sh, int prim, v3 p)
{
	if (kGeneral >= 2 && prim >= sh.n_quads + sh.n_spheres + 6 * sh.n_boxes)
		return ld3(sh.tris + (size_t)(prim - sh.n_quads - sh.n_spheres - 6 * sh.n_boxes) * kTriStride + 9);
	if (prim >= sh.n_quads + sh.n_spheres) {
		const int f = prim - sh.n_quads - sh.n_spheres;
		return box_face_normal(sh.boxes + (f / 6) * kBoxStride, f % 6);
	}
	if (!kGeneral || prim < sh.n_quads) return ld3(sh.quads + prim * kQuadStride + 9);
	return normalize3(vsub(p, ld3(sh.spheres + (prim - sh.n_quads) * kSphereStride)));
}

// scene.pdf_emitter_direction(prev, ds) for a hit on emitter shape `prim` at p (normal n) seen from
// `ref`, times the 1/count of the uniform emitter choice
template <int kGeneral>
__device__ __forceinline__ float emitter_hit_pdf(const Shapes &sh, int prim, v3 ref, v3 p, v3 n, float inv_count)
{
	const v3 dd = vsub(p, ref);
	const float d2 = dot3(dd, dd), dist = __builtin_sqrtf(d2);
	const v3 dn = vdivs(dd, dist);
	const float dp = dot3(dn, n);
	if (!(dp < 0.0f)) return 0.0f;
	float pdf;
	if (!kGeneral || prim < sh.n_quads) {
		pdf = d2 / (fabs_(dp) * sh.quads[prim * kQuadStride + 14]);
	} else { // Sphere::pdf_direction
		const float *S = sh.spheres + (prim - sh.n_quads) * kSphereStride;
		const v3 cv = vsub(ld3(S), ref);
		const float sin_alpha = S[3] / __builtin_sqrtf(dot3(cv, cv));
		const float cos_alpha = safe_sqrtf(1.0f - sin_alpha * sin_alpha);
		if (sin_alpha < 0.99999994f) pdf = kInvTwoPiF / (1.0f - cos_alpha);
		else pdf = (d2 / fabs_(dp)) / ((4.0f * kPiF) * (S[3] * S[3]));
	}
	return pdf * inv_count;
}

// Mitsuba warp::square_to_uniform_disk_concentric
__device__ __forceinline__ void square_to_disk(float u, float v, float &px, float &py)
{
	const float x = 2.0f * u - 1.0f, y = 2.0f * v - 1.0f;
	const bool is_zero = (x == 0.0f) && (y == 0.0f);
	const bool q13 = fabs_(x) < fabs_(y);
	const float r = q13 ? y : x, rp = q13 ? x : y;
	float phi = (0.25f * 3.14159265358979323846f) * (rp / r);
	if (q13) phi = (0.5f * 3.14159265358979323846f) - phi;
	if (is_zero) phi = 0.0f;
	float s, c;
	sincos_f32(phi, s, c);
	px = r * c;
	py = r * s;
}

// Mitsuba warp::square_to_cosine_hemisphere (concentric disk)
__device__ __forceinline__ v3 square_to_cosine_hemisphere(float u, float v)
{
	float px, py;
	square_to_disk(u, v, px, py);
	const float zz = 1.0f - (px * px + py * py);
	float z = zz > 0.0f ? __builtin_sqrtf(zz) : 0.0f;
	if (z == 0.0f) z = 1e-10f;
	return V(px, py, z);
}

// ---- roughconductor (Beckmann, isotropic, sample_visible) after Mitsuba 3's microfacet.h /
// roughconductor.cpp; local frame, cos(theta_i) > 0.  Not inlined: the diffuse-only kernel never
// references them, and the general kernel calls them from two places each.
// The sign of `alpha` names the distribution: > 0 Beckmann, < 0 GGX of roughness -alpha.
__device__ __noinline__ float rc_D(v3 m, float alpha) // MicrofacetDistribution::eval
{
	const float ct = m.z, ct2 = ct * ct;
	const float a = fabs_(alpha);
	const float ax = m.x / a, ay = m.y / a;
	float result;
	if (alpha < 0.0f) {
		const float t = (ax * ax + ay * ay) + ct2;
		result = 1.0f / (((kPiF * a) * a) * (t * t));
	} else {
		result = exp_f32(-((ax * ax + ay * ay) / ct2)) / (((kPiF * a) * a) * (ct2 * ct2));
	}
	return result * ct > 1e-20f ? result : 0.0f;
}

__device__ __forceinline__ float rc_G1(v3 v, v3 m, float alpha) // smith_g1: exact for GGX, rational approximation for Beckmann
{
	const float ax = alpha * v.x, ay = alpha * v.y;
	const float xy = ax * ax + ay * ay;
	const float a = 1.0f / __builtin_sqrtf(xy / (v.z * v.z));
	const float a2 = a * a;
	float result = a >= 1.6f ? 1.0f : (3.535f * a + 2.181f * a2) / ((1.0f + 2.276f * a) + 2.577f * a2);
	if (alpha < 0.0f) result = 2.0f / (1.0f + __builtin_sqrtf(1.0f + xy / (v.z * v.z)));
	if (xy == 0.0f) result = 1.0f;
	if (dot3(v, m) * v.z <= 0.0f) result = 0.0f;
	return result;
}

__device__ __forceinline__ float fresnel_conductor(float cos_i, float eta_r, float eta_i)
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

__device__ __forceinline__ v3 rc_fresnel(const float *M, float cos_i)
{
	return V(fresnel_conductor(cos_i, M[5], M[8]), fresnel_conductor(cos_i, M[6], M[9]), fresnel_conductor(cos_i, M[7], M[10]));
}

__device__ __noinline__ float erfinv_call(float x) { return erfinv_f32(x); }

// sample_visible_11: slopes of the visible Beckmann normals for alpha = 1
__device__ __noinline__ void rc_sample_visible_11(float cos_i, float u1, float u2, float &sx, float &sy)
{
	const float tan_i = safe_sqrtf(1.0f - cos_i * cos_i) / cos_i;
	const float cot_i = 1.0f / tan_i;
	const float maxval = erf_f32(cot_i);
	u1 = u1 < 1.0f - 1e-6f ? u1 : 1.0f - 1e-6f; u1 = u1 > 1e-6f ? u1 : 1e-6f;
	u2 = u2 < 1.0f - 1e-6f ? u2 : 1.0f - 1e-6f; u2 = u2 > 1e-6f ? u2 : 1e-6f;
	float x = maxval - (maxval + 1.0f) * erf_f32(__builtin_sqrtf(-log_f32(u1)));
	const float tail = tan_i == 0.0f ? 0.0f : (kInvSqrtPiF * tan_i) * exp_f32(-(cot_i * cot_i));
	u1 = u1 * ((1.0f + maxval) + tail);
	for (int i = 0; i < 3; ++i) {
		const float slope = erfinv_call(x);
		const float value = ((1.0f + x) + (kInvSqrtPiF * tan_i) * exp_f32(-(slope * slope))) - u1;
		const float derivative = 1.0f - slope * tan_i;
		x = x - value / derivative;
	}
	sx = erfinv_call(x);
	sy = erfinv_call(2.0f * u2 - 1.0f);
}

// sample_visible_11 for GGX: a point of the unit disk, its half towards the viewer compressed by
// (1 + cos theta_i)/2, projected onto the hemisphere around the viewing direction, as slopes
__device__ __noinline__ void ggx_sample_visible_11(float cos_i, float u1, float u2, float &sx, float &sy)
{
	float px, py;
	square_to_disk(u1, u2, px, py);
	const float s = 0.5f * (1.0f + cos_i);
	const float h = safe_sqrtf(1.0f - px * px);
	py = h * (1.0f - s) + py * s;
	const float z = safe_sqrtf(1.0f - (px * px + py * py));
	const float sin_i = safe_sqrtf(1.0f - cos_i * cos_i);
	const float norm = 1.0f / (sin_i * py + cos_i * z);
	sx = (cos_i * py - sin_i * z) * norm;
	sy = px * norm;
}

// MicrofacetDistribution::sample (visible normals): microfacet normal and its density
__device__ __forceinline__ v3 rc_sample_m(v3 wi, float signed_alpha, float u1, float u2, float &pdf)
{
	const float alpha = fabs_(signed_alpha);
	const v3 wip = normalize3(V(alpha * wi.x, alpha * wi.y, wi.z));
	const float s2 = wip.x * wip.x + wip.y * wip.y; // Frame::sincos_phi
	float cos_phi = 1.0f, sin_phi = 0.0f;
	if (fabs_(s2) > 4.0f * 5.9604644775390625e-08f) {
		const float inv = 1.0f / __builtin_sqrtf(s2);
		cos_phi = wip.x * inv; sin_phi = wip.y * inv;
		cos_phi = cos_phi < -1.0f ? -1.0f : (cos_phi > 1.0f ? 1.0f : cos_phi);
		sin_phi = sin_phi < -1.0f ? -1.0f : (sin_phi > 1.0f ? 1.0f : sin_phi);
	}
	float sx, sy;
	if (signed_alpha < 0.0f) ggx_sample_visible_11(wip.z, u1, u2, sx, sy);
	else rc_sample_visible_11(wip.z, u1, u2, sx, sy);
	const float rx = (cos_phi * sx - sin_phi * sy) * alpha;
	const float ry = (sin_phi * sx + cos_phi * sy) * alpha;
	const v3 m = normalize3(V(-rx, -ry, 1.0f));
	pdf = ((rc_D(m, signed_alpha) * rc_G1(wi, m, signed_alpha)) * fabs_(dot3(wi, m))) / wi.z;
	return m;
}

__device__ __forceinline__ void rc_eval_pdf(const Material &mt, v3 wi, v3 wo, v3 &value, float &pdf) // wi.z > 0
{
	value = V(0, 0, 0);
	pdf = 0.0f;
	if (!(wi.z > 0.0f && wo.z > 0.0f)) return;
	const float alpha = mt.M[4];
	const v3 H = normalize3(vadd(wo, wi));
	const float D = rc_D(H, alpha);
	if (D == 0.0f) return;
	const float g_i = rc_G1(wi, H, alpha);
	const float res = (D * (g_i * rc_G1(wo, H, alpha))) / (4.0f * wi.z);
	const v3 F = rc_fresnel(mt.M, dot3(wi, H));
	value = vmul(F, vscale(mt.refl, res));
	if (dot3(wi, H) > 0.0f && dot3(wo, H) > 0.0f) pdf = (D * g_i) / (4.0f * wi.z);
}

__device__ __forceinline__ void rc_sample(const Material &mt, v3 wi, float u1, float u2, v3 &wo, float &pdf, v3 &weight) // wi.z > 0
{
	wo = V(0, 0, 0); pdf = 0.0f; weight = V(0, 0, 0);
	const float alpha = mt.M[4];
	float pdf_m;
	const v3 m = rc_sample_m(wi, alpha, u1, u2, pdf_m);
	const float wim = dot3(wi, m);
	const v3 o = vsub(vscale(m, 2.0f * wim), wi); // reflect(wi, m)
	if (!(pdf_m != 0.0f && o.z > 0.0f)) return;
	const float p = pdf_m / (4.0f * dot3(o, m));
	const v3 F = rc_fresnel(mt.M, wim);
	wo = o;
	pdf = p;
	weight = vmul(F, vscale(mt.refl, rc_G1(o, m, alpha)));
}

// Mitsuba fresnel(cos_theta_i, eta): unpolarised reflectance of a dielectric interface, the signed
// cosine of the transmitted direction, the relative index along / against the ray
__device__ __forceinline__ float fresnel_dielectric(float cos_i, float eta, float &cos_t, float &eta_it, float &eta_ti)
{
	const bool outside = cos_i >= 0.0f;
	const float rcp_eta = 1.0f / eta;
	eta_it = outside ? eta : rcp_eta;
	eta_ti = outside ? rcp_eta : eta;
	const float cos_t_sqr = 1.0f - ((1.0f - cos_i * cos_i) * (eta_ti * eta_ti));
	const float ci = fabs_(cos_i), ct = safe_sqrtf(cos_t_sqr);
	const float a_s = (eta_it * ct - ci) / (eta_it * ct + ci);
	const float a_p = (eta_it * ci - ct) / (eta_it * ci + ct);
	float r = 0.5f * (a_s * a_s + a_p * a_p);
	if (eta == 1.0f) r = 0.0f;
	else if (ci == 0.0f) r = 1.0f;
	cos_t = cos_i >= 0.0f ? -ct : ct;
	return r;
}

// ---- roughdielectric (Beckmann, isotropic, sample_visible) after Mitsuba 3's roughdielectric.cpp:
// reflection and transmission through a rough interface, radiance transport; wi on either side,
// M[4] = alpha, M[5] = int_ior / ext_ior.  Out of line: only scenes with such a material get here.
__device__ __forceinline__ v3 vflip_if(v3 v, bool c) { return c ? V(-v.x, -v.y, -v.z) : v; }

__device__ __noinline__ void rd_eval_pdf(const float *M, v3 wi, v3 wo, v3 &value, float &pdf)
{
	value = V(0, 0, 0);
	pdf = 0.0f;
	const float alpha = M[4], eta_m = M[5];
	const float ci = wi.z, co = wo.z;
	if (ci == 0.0f) return;
	const bool reflect = ci * co > 0.0f;
	const float eta = ci > 0.0f ? eta_m : 1.0f / eta_m, inv_eta = ci > 0.0f ? 1.0f / eta_m : eta_m;
	v3 m = normalize3(vadd(wi, vscale(wo, reflect ? 1.0f : eta)));
	m = vflip_if(m, m.z < 0.0f); // into the hemisphere of the macro-surface normal
	const float D = rc_D(m, alpha);
	float cos_t, eta_it, eta_ti;
	const float wim = dot3(wi, m), wom = dot3(wo, m);
	const float F = fresnel_dielectric(wim, eta_m, cos_t, eta_it, eta_ti);
	const float G = rc_G1(wi, m, alpha) * rc_G1(wo, m, alpha);
	const float denom = wim + eta * wom;
	float val;
	if (reflect) val = ((F * D) * G) / (4.0f * fabs_(ci));
	else val = fabs_(((((((inv_eta * inv_eta) * (1.0f - F)) * D) * G) * (eta * eta)) * (wim * wom)) / (ci * (denom * denom)));
	if (!(val == val)) val = 0.0f;
	value = V(val, val, val);
	if (!(wim * ci > 0.0f && wom * co > 0.0f)) return; // the micro- and macro-surface must agree on the sides
	const float dwh_dwo = reflect ? 1.0f / (4.0f * wom) : ((eta * eta) * wom) / (denom * denom);
	const v3 wiu = vflip_if(wi, ci < 0.0f);
	float prob = ((D * rc_G1(wiu, m, alpha)) * fabs_(dot3(wiu, m))) / wiu.z;
	prob = prob * (reflect ? F : 1.0f - F);
	float p = prob * fabs_(dwh_dwo);
	if (!(p == p)) p = 0.0f;
	pdf = p;
}

__device__ __noinline__ void rd_sample(const float *M, v3 wi, float u1, float u, float v, v3 &wo, float &pdf, v3 &weight, float &eta_out)
{
	wo = V(0, 0, 0); pdf = 0.0f; weight = V(0, 0, 0); eta_out = 0.0f;
	const float alpha = M[4], eta_m = M[5];
	const float ci = wi.z;
	if (ci == 0.0f) return;
	float pdf_m;
	const v3 m = rc_sample_m(vflip_if(wi, ci < 0.0f), alpha, u, v, pdf_m);
	if (!(pdf_m != 0.0f)) return;
	float cos_t, eta_it, eta_ti;
	const float wim = dot3(wi, m);
	const float F = fresnel_dielectric(wim, eta_m, cos_t, eta_it, eta_ti);
	const bool reflect = u1 <= F;
	float p = pdf_m * (reflect ? F : 1.0f - F);
	v3 o;
	float w = 1.0f, dwh_dwo, e;
	if (reflect) {
		o = vsub(vscale(m, 2.0f * wim), wi);
		dwh_dwo = 1.0f / (4.0f * dot3(o, m));
		e = 1.0f;
	} else {
		o = vsub(vscale(m, wim * eta_ti + cos_t), vscale(wi, eta_ti)); // refract(wi, m, cos_theta_t, eta_ti)
		w = eta_ti * eta_ti;
		const float om = dot3(o, m), denom = wim + eta_it * om;
		dwh_dwo = ((eta_it * eta_it) * om) / (denom * denom);
		e = eta_it;
	}
	eta_out = e;
	w = w * rc_G1(o, m, alpha);
	p = p * fabs_(dwh_dwo);
	if (!(p == p) || !(w == w)) return;
	wo = o;
	pdf = p;
	weight = V(w, w, w);
}

// BSDFFlags.Smooth (:210): does the material have a non-delta lobe?
__device__ __forceinline__ bool material_is_smooth(const Material &mt) { return mt.type != 2 && mt.type != 3; }

// bsdf.eval_pdf (twosided unless the material says otherwise): value includes cos(theta_o)
template <int kGeneral>
__device__ __forceinline__ void bsdf_eval_pdf(const Material &mt, v3 wi, v3 wo, bool active, v3 &value, float &pdf)
{
	value = V(0, 0, 0);
	pdf = 0.0f;
	if (!active) return;
	if (kGeneral >= 2 && (mt.type == 2 || mt.type == 3)) return; // smooth conductor / dielectric: delta lobes only
	if (kGeneral >= 2 && mt.type == 4) {
		rd_eval_pdf(mt.M, wi, wo, value, pdf);
		return;
	}
	if (wi.z < 0.0f && !(kGeneral >= 2 && mt.one_sided)) { wi.z = -wi.z; wo.z = -wo.z; }
	if (kGeneral && mt.type == 1) {
		rc_eval_pdf(mt, wi, wo, value, pdf);
		return;
	}
	const v3 refl = mt.refl;
	if (!(wi.z > 0.0f && wo.z > 0.0f)) return;
	value = vscale(vscale(refl, kInvPiF), wo.z);
	pdf = kInvPiF * wo.z;
}

// bsdf.sample(ctx, si, u1, (u, v)): wo (local), pdf, weight = value / pdf, the relative index along
// wo, and whether a delta lobe was sampled (BSDFFlags.Delta, :282)
template <int kGeneral>
__device__ __forceinline__ void bsdf_sample(const Material &mt, v3 wi, float u1, float u, float v, bool active, v3 &wo,
                                            float &pdf, v3 &weight, float &eta, bool &delta)
{
	wo = V(0, 0, 0); pdf = 0.0f; weight = V(0, 0, 0); eta = 0.0f; delta = false;
	if (!active) return;
	if (kGeneral >= 2 && mt.type == 3) { // smooth dielectric (dielectric.cpp), radiance transport
		float cos_t, eta_it, eta_ti;
		const float r_i = fresnel_dielectric(wi.z, mt.M[5], cos_t, eta_it, eta_ti);
		const bool reflect = u1 <= r_i;
		const float sc = eta_ti * eta_ti;
		delta = true;
		pdf = reflect ? r_i : 1.0f - r_i;
		wo = reflect ? V(-wi.x, -wi.y, wi.z) : V(-eta_ti * wi.x, -eta_ti * wi.y, cos_t);
		eta = reflect ? 1.0f : eta_it;
		weight = reflect ? V(1, 1, 1) : V(sc, sc, sc);
		return;
	}
	if (kGeneral >= 2 && mt.type == 4) {
		rd_sample(mt.M, wi, u1, u, v, wo, pdf, weight, eta);
		return;
	}
	const bool flip = wi.z < 0.0f && !(kGeneral >= 2 && mt.one_sided);
	const float cos_i = flip ? -wi.z : wi.z;
	if (!(cos_i > 0.0f)) return;
	if (kGeneral >= 2 && mt.type == 2) { // smooth conductor (conductor.cpp): the mirror direction, weighted by Fresnel
		delta = true;
		pdf = 1.0f;
		eta = 1.0f;
		wo = V(-wi.x, -wi.y, wi.z);
		weight = vmul(rc_fresnel(mt.M, cos_i), mt.refl);
		return;
	}
	if (kGeneral && mt.type == 1) {
		v3 o;
		rc_sample(mt, V(wi.x, wi.y, cos_i), u, v, o, pdf, weight);
		eta = 1.0f;
		if (flip) o.z = -o.z;
		wo = o;
		return;
	}
	const v3 refl = mt.refl;
	v3 w = square_to_cosine_hemisphere(u, v);
	const float p = kInvPiF * w.z;
	eta = 1.0f;
	pdf = p;
	if (p > 0.0f) weight = refl;
	if (flip) w.z = -w.z;
	wo = w;
}

// scene.sample_emitter_direction(si, (e1, e2), test_visibility=True): uniform choice of one emitter
// (e1 is reused after the choice), then a point on it; returns ds.d, ds.pdf and radiance / pdf
// (zero when occluded, facing away, or from inside a sphere)
// Directional emitters of a scene (scenes/torus/scene.xml) and the bounding sphere their samples sit on
struct DirLights {
	const float *lights; // 8 floats each: 0-2 unit direction the light travels in, 3-5 irradiance
	float bsphere[4];    // centre, radius
};

template <int kGeneral>
__device__ __forceinline__ void sample_emitter(const Shapes &sh, const DirLights &dls, const int32_t *__restrict__ emitters,
                                               int n_em, v3 p, v3 n, float e1, float e2, v3 &ds_d, float &ds_pdf,
                                               v3 &em_weight, bool &ds_delta)
{
	ds_d = V(0, 0, 0);
	ds_pdf = 0.0f;
	em_weight = V(0, 0, 0);
	ds_delta = false;
	if (n_em <= 0) return;
	const float count = (float)n_em, inv_count = 1.0f / count;
	uint32_t idx = (uint32_t)(e1 * count);
	if (idx > (uint32_t)(n_em - 1)) idx = (uint32_t)(n_em - 1);
	e1 = e1 * count - (float)idx;
	const int prim = emitters[idx];
	if (kGeneral >= 2 && prim < 0) { // directional.cpp sample_direction: a point two radii up the light's direction, pdf 1, delta
		const float *Dl = dls.lights + (size_t)(-1 - prim) * 8;
		const v3 dl = ld3(Dl);
		const v3 cd = vsub(p, V(dls.bsphere[0], dls.bsphere[1], dls.bsphere[2]));
		const float dc = __builtin_sqrtf(dot3(cd, cd));
		const float dist = 2.0f * (dls.bsphere[3] > dc ? dls.bsphere[3] : dc);
		const v3 pl = vsub(p, vscale(dl, dist));
		ds_d = V(-dl.x, -dl.y, -dl.z);
		ds_delta = true;
		ds_pdf = 1.0f * inv_count;
		float mag = (1.0f + max3(V(fabs_(p.x), fabs_(p.y), fabs_(p.z)))) * kRayEps;
		if (dot3(n, ds_d) < 0.0f) mag = -mag;
		const v3 so = vadd(p, vscale(n, mag));
		const v3 sd = vsub(pl, so);
		const float sdist = __builtin_sqrtf(dot3(sd, sd));
		float th;
		const bool occ = intersect<kGeneral, true>(sh, so, vdivs(sd, sdist), sdist * (1.0f - kShadowEps), th) >= 0;
		if (!occ) em_weight = vscale(ld3(Dl + 3), count);
		return;
	}
	v3 pl, ln, radiance;
	float pdf_cone = 0.0f, area = 1.0f;
	const bool is_sphere = kGeneral && prim >= sh.n_quads;
	if (!is_sphere) {
		const float *E = sh.quads + prim * kQuadStride;
		pl = vadd(vadd(ld3(E), vscale(ld3(E + 3), e1)), vscale(ld3(E + 6), e2));
		ln = ld3(E + 9);
		radiance = ld3(E + 19);
		area = E[14];
	} else { // Sphere::sample_direction, reference point outside
		const float *S = sh.spheres + (prim - sh.n_quads) * kSphereStride;
		const v3 c = ld3(S);
		const float r = S[3];
		const v3 dc_v = vsub(c, p);
		const float dc_2 = dot3(dc_v, dc_v);
		const float radius_adj = r * (1.0f - kSphereEps);
		if (!(dc_2 > radius_adj * radius_adj)) return;
		const float inv_dc = 1.0f / __builtin_sqrtf(dc_2);
		const float sin_max = r * inv_dc, sin_max2 = sin_max * sin_max, inv_sin_max = 1.0f / sin_max;
		const float cos_max = safe_sqrtf(1.0f - sin_max2);
		float sin_theta_2;
		if (sin_max2 > 0.00068523f) { // sin^2(1.5 deg)
			const float tt = 1.0f + (cos_max - 1.0f) * e1;
			sin_theta_2 = 1.0f - tt * tt;
		} else sin_theta_2 = sin_max2 * e1; // small-angle Taylor expansion
		const float cos_theta = safe_sqrtf(1.0f - sin_theta_2);
		const float cos_alpha = sin_theta_2 * inv_sin_max +
		                        cos_theta * safe_sqrtf(1.0f - sin_theta_2 * (inv_sin_max * inv_sin_max));
		const float sin_alpha = safe_sqrtf(1.0f - cos_alpha * cos_alpha);
		float sin_phi, cos_phi;
		sincos_f32(e2 * (2.0f * kPiF), sin_phi, cos_phi);
		const Frame fr = make_frame(vscale(dc_v, -inv_dc));
		const v3 dl = to_world(fr, V(cos_phi * sin_alpha, sin_phi * sin_alpha, cos_alpha));
		pl = vadd(c, vscale(dl, r));
		ln = dl;
		radiance = ld3(S + 6);
		pdf_cone = kInvTwoPiF / (1.0f - cos_max);
	}
	const v3 dir0 = vsub(pl, p);
	// si.spawn_ray_to(pl): offset origin, then aim at the light point
	float mag = (1.0f + max3(V(fabs_(p.x), fabs_(p.y), fabs_(p.z)))) * kRayEps;
	if (dot3(n, dir0) < 0.0f) mag = -mag;
	const v3 so = vadd(p, vscale(n, mag));
	const float d2 = dot3(dir0, dir0), dist = __builtin_sqrtf(d2);
	ds_d = vdivs(dir0, dist);
	const float dp = dot3(ds_d, ln);
	float pdf = 0.0f;
	if (dp < 0.0f) pdf = is_sphere ? (dist == 0.0f ? 0.0f : pdf_cone) : d2 / (fabs_(dp) * area);
	if (!(pdf == pdf) || pdf == __builtin_huge_valf()) pdf = 0.0f;
	ds_pdf = pdf * inv_count;
	if (pdf > 0.0f) {
		const v3 sd = vsub(pl, so);
		const float sdist = __builtin_sqrtf(dot3(sd, sd));
		const v3 sdn = vdivs(sd, sdist);
		float th;
#ifdef PG_ABLATE_SHADOW // timing experiment only: no shadow rays
		const bool occ = false; (void)th; (void)sdn;
#else
		const bool occ = intersect<kGeneral, true>(sh, so, sdn, sdist * (1.0f - kShadowEps), th) >= 0;
#endif
		if (!occ) em_weight = vscale(vdivs(radiance, pdf), count);
	}
}

struct RenderArgs {
	TreeView tree;
	Shapes shapes;
	const float *mats;          // material table (general scenes)
	const int32_t *emitters;    // the emitters: shape numbers of flagged quads, then flagged spheres, then -1-k for directional light k
	int n_emitters;
	DirLights dir_lights;
	float *ior;                 // general scenes: running product of the relative indices along the path (:357)
	pg_camera cam;
	uint64_t n_lanes, n_pixels;      // of this pass (tile)
	uint64_t pixel_begin, film_pixels; // first pixel of the tile, pixels of the whole film
	int spp, max_depth, rr_depth, guided, record, store_nee;
	float frac;
	uint32_t seed;
	DepthCounters *dc;
	int bounce, last;          // index of this launch = path depth of every live lane; last launch of the pass
	const uint32_t *order_in;  // live-ray list written by the previous bounce (unused by the first)
	uint32_t *order_out;       // live-ray list for the next bounce
	uint32_t *live_count;      // [max_depth + 1]: live_count[b] = lanes alive after bounce b; [max_depth]: see k_bounce_tail (zeroed per pass)
	// per-lane state (planar), 57 B in and out per live lane and bounce.  The ray origin is not
	// state: it is the previous vertex pushed off its quad (:352 spawn_ray), recomputed from prev_p
	// and the quad id; depth is the launch index; ior stays 1 (every BSDF of the substrate has eta 1)
	float *ray_d, *thr, *L, *prev_p, *prev_pdf;
	uint32_t *prev_quad; // shape number of the previous vertex
	uint8_t *hit0; // the first bounce hit something (the `valid` flag, :400)
	uint64_t *rng_state, *rng_inc;
	// path-vertex records: a list in visiting order, planes of stride n_lanes*max_depth
	uint32_t *ray_of;
	float *r_pos, *r_dir, *r_bsdf, *r_tb, *r_tr, *r_nee, *r_dnee, *r_wp;
};

// One loop iteration of :179-381 for one live lane; returns whether the path continues.
// kFirst: the camera ray is generated here (mi.render's sensor.sample_ray_differential: one 2-D
// jitter draw per sample, box reconstruction) instead of being read back from a generate kernel.
// kGeneral: the scene has spheres or rough conductors; false compiles the all-diffuse quad scene only.
template <bool kFirst, int kGeneral>
__device__ __forceinline__ bool bounce_lane(const RenderArgs &a, const uint4 *s_kd, const uint64_t lane,
                                            const uint64_t rec_slot, const uint32_t depth)
{
	const uint64_t N = a.n_lanes;
	const int D = a.max_depth;
	const float f = a.frac;
	const Shapes &sh = a.shapes;
	Pcg32 rng;
	v3 ray_o, ray_d, thr, L, prev_p;
	float prev_bsdf_pdf;
	bool prev_delta = kFirst; // general scenes: bit 31 of the stored shape number (a delta lobe was sampled)
	float ior = 1.0f;         // general scenes: a.ior
	if (kFirst) {
		// streams are keyed by the GLOBAL lane id (pixel*spp + s): a tile renders exactly the samples
		// the full-frame pass would, whatever the number of ranks
		rng = pcg32_seed(a.seed, (uint32_t)(a.pixel_begin * (uint64_t)a.spp + lane));
		const uint64_t pixel = a.pixel_begin + lane / (uint64_t)a.spp;
		const int W = a.cam.width, H = a.cam.height;
		const float px = (float)(pixel % (uint64_t)W), py = (float)(pixel / (uint64_t)W);
		const float jx = rng.next_f32(), jy = rng.next_f32();
		const float tan_y = a.cam.tan_half_fov_x / ((float)W / (float)H);
		const float cx = (1.0f - 2.0f * ((px + jx) / (float)W)) * a.cam.tan_half_fov_x;
		const float cy = (1.0f - 2.0f * ((py + jy) / (float)H)) * tan_y;
		const float len = __builtin_sqrtf((cx * cx + cy * cy) + 1.0f);
		const v3 dc = V(cx / len, cy / len, 1.0f / len);
		ray_d = vadd(vadd(vscale(ld3(a.cam.axis_x), dc.x), vscale(ld3(a.cam.axis_y), dc.y)), vscale(ld3(a.cam.axis_z), dc.z));
		ray_o = ld3(a.cam.origin);
		thr = V(1, 1, 1);
		L = V(0, 0, 0);
		prev_p = V(0, 0, 0);
		prev_bsdf_pdf = 1.0f;
	} else {
		rng.state = a.rng_state[lane];
		rng.inc = a.rng_inc[lane];
		ray_d = V(a.ray_d[lane], a.ray_d[N + lane], a.ray_d[2 * N + lane]);
		thr = V(a.thr[lane], a.thr[N + lane], a.thr[2 * N + lane]);
		L = V(a.L[lane], a.L[N + lane], a.L[2 * N + lane]);
		prev_p = V(a.prev_p[lane], a.prev_p[N + lane], a.prev_p[2 * N + lane]);
		prev_bsdf_pdf = a.prev_pdf[lane];
		// :352 spawn_ray of the previous vertex: the same three operations that produced the origin
		const uint32_t pq = a.prev_quad[lane];
		if (kGeneral >= 2) {
			prev_delta = (pq >> 31) != 0u;
			ior = a.ior[lane];
		}
		const v3 pn = normal_at<kGeneral>(sh, (int)(pq & 0x7fffffffu), prev_p);
		float mag = (1.0f + max3(V(fabs_(prev_p.x), fabs_(prev_p.y), fabs_(prev_p.z)))) * kRayEps;
		if (dot3(pn, ray_d) < 0.0f) mag = -mag;
		ray_o = vadd(prev_p, vscale(pn, mag));
	}

	// ---- :185 ray_intersect ----
	float t_hit;
	const int q = intersect<kGeneral>(sh, ray_o, ray_d, __builtin_huge_valf(), t_hit);
	const bool valid = q >= 0;
	Surface sf;
	sf.p = V(0, 0, 0); sf.n = V(0, 0, 1); sf.ng = V(0, 0, 1); sf.radiance = V(0, 0, 0); sf.is_em = false;
	sf.m.type = 0; sf.m.refl = V(0, 0, 0); sf.m.M = nullptr; sf.m.one_sided = false;
	if (valid) sf = surface_at<kGeneral>(sh, a.mats, q, ray_o, ray_d, t_hit);
	const v3 p = sf.p, n = sf.n;
	const Material &mt = sf.m;
	const Frame fr = make_frame(n);
	const v3 wi = to_local(fr, V(-ray_d.x, -ray_d.y, -ray_d.z));
	const bool is_em = valid && sf.is_em;
	const float inv_em_count = 1.0f / (float)a.n_emitters; // only used when an emitter was hit
	// ---- :189-200 direct emission ----
	const v3 em_radiance = (is_em && wi.z > 0.0f) ? sf.radiance : V(0, 0, 0);
	float emitter_pdf = 0.0f;
	if (is_em && !prev_delta) emitter_pdf = emitter_hit_pdf<kGeneral>(sh, q, prev_p, p, n, inv_em_count);
	const float mis = mis_weight(prev_bsdf_pdf, emitter_pdf);
	const v3 Le = vmul(vscale(thr, mis), em_radiance);
	// ---- :207-220 emitter sampling ----
	bool active_next = (depth + 1 < (uint32_t)D) && valid;
	bool active_em = active_next && (kGeneral < 2 || material_is_smooth(mt)); // :210 BSDFFlags.Smooth
	const float e1 = rng.next_f32(), e2 = rng.next_f32(); // :214, unmasked
	v3 ds_d = V(0, 0, 0), em_weight = V(0, 0, 0);
	float ds_pdf = 0.0f;
	bool ds_delta = false;
	if (active_em)
		sample_emitter<kGeneral>(sh, a.dir_lights, a.emitters, a.n_emitters, p, sf.ng, e1, e2, ds_d, ds_pdf, em_weight, ds_delta);
	active_em = active_em && (ds_pdf != 0.0f); // :216
	const v3 wo_em = to_local(fr, ds_d);
	v3 bsdf_value_em;
	float bsdf_pdf_em;
	bsdf_eval_pdf<kGeneral>(mt, wi, wo_em, active_em, bsdf_value_em, bsdf_pdf_em);
	// ---- :223-256 NEE MIS against the mixture pdf ----
	const bool active_sd_em = active_em && a.guided;
	const float pdf_diffuse = 1.0f; // :222-241
	TreeHead head = {kNoRecord, 0.0f};
	uint32_t tree_id = 0;
	bool tree_known = false;
	float sdtree_pdf_em = 1.0f;
	uint32_t lv;
	unsigned c_kd = 0, c_kdq = 0, c_q = 0, c_qq = 0; // descent statistics for the byte model
	const bool do_record = a.record && valid;
	// dirToCanonical of the emitter direction feeds the NEE pdf query (:244) and the record (:338):
	// one evaluation serves both
	float nee_cx = 0.0f, nee_cy = 0.0f;
	if (active_sd_em || (do_record && a.store_nee)) dir_to_canonical(ds_d.x, ds_d.y, ds_d.z, nee_cx, nee_cy);
	if (active_sd_em) {
		KdNode leaf;
		kd_descend_lds(a.tree.kd, s_kd, p.x, p.y, p.z, inside_root(a.tree, p.x, p.y, p.z), leaf, lv);
		c_kd += lv; ++c_kdq;
		const uint2 hv = *reinterpret_cast<const uint2 *>(a.tree.head + leaf.tree);
		head.root_rec = hv.x;
		head.root_irr = __uint_as_float(hv.y);
		tree_known = true;
		tree_id = leaf.tree;
		sdtree_pdf_em = quad_pdf(a.tree.rec, a.tree.jump, tree_id, head, nee_cx, nee_cy, lv);
		c_q += lv; ++c_qq;
	}
	float surface_pdf_em = f * bsdf_pdf_em + ((1.0f - f) * sdtree_pdf_em) * pdf_diffuse;
	if (!a.guided) surface_pdf_em = bsdf_pdf_em;
	const float mis_em = (kGeneral >= 2 && ds_delta) ? 1.0f : mis_weight(ds_pdf, surface_pdf_em); // :253
	const v3 Lr_dir = vmul(vmul(vscale(thr, mis_em), bsdf_value_em), em_weight);
	L = vadd(L, vadd(Le, Lr_dir)); // :261
	// ---- :272-311 next direction ----
	float s1 = 0.0f, s2x = 0.0f, s2y = 0.0f;
	if (active_next) { // next_1d (lobe choice: only the dielectric reads it), next_2d
		if (kGeneral >= 2) s1 = rng.next_f32();
		else rng.skip();
		s2x = rng.next_f32();
		s2y = rng.next_f32();
	}
	v3 wo_local, bsdf_weight;
	float bsdf_pdf, eta;
	bool delta;
	bsdf_sample<kGeneral>(mt, wi, s1, s2x, s2y, active_next, wo_local, bsdf_pdf, bsdf_weight, eta, delta);
	v3 bsdf_value = vscale(bsdf_weight, bsdf_pdf);
	float woPdf = bsdf_pdf;
	v3 wo_world = to_world(fr, wo_local);
	const bool do_mis = active_next && !delta && a.guided; // :283
	bool pick_tree = false;
	if (active_next) pick_tree = rng.next_f32() > f; // :286
	const bool smp_tree = pick_tree && do_mis;
	const bool bsdf_mis = do_mis && !smp_tree;
	float sdtree_pdf = 1.0f;
	if ((smp_tree || bsdf_mis) && !tree_known) {
		KdNode leaf;
		kd_descend_lds(a.tree.kd, s_kd, p.x, p.y, p.z, inside_root(a.tree, p.x, p.y, p.z), leaf, lv);
		c_kd += lv; ++c_kdq;
		const uint2 hv = *reinterpret_cast<const uint2 *>(a.tree.head + leaf.tree);
		head.root_rec = hv.x;
		head.root_irr = __uint_as_float(hv.y);
		tree_id = leaf.tree;
	}
	if (smp_tree) { // :301-304
		float dx, dy, dz;
		quad_sample(a.tree.rec, a.tree.jump, tree_id, head, rng, dx, dy, dz, sdtree_pdf, lv);
		c_q += lv; ++c_qq;
		wo_world = V(dx, dy, dz);
		wo_local = to_local(fr, wo_world);
		bsdf_eval_pdf<kGeneral>(mt, wi, wo_local, true, bsdf_value, bsdf_pdf);
	}
	// dirToCanonical of the continuation direction feeds the pdf query (:307) and the record (:327)
	float wo_cx = 0.0f, wo_cy = 0.0f;
	if (bsdf_mis || do_record) dir_to_canonical(wo_world.x, wo_world.y, wo_world.z, wo_cx, wo_cy);
	if (bsdf_mis) { // :307
		sdtree_pdf = quad_pdf(a.tree.rec, a.tree.jump, tree_id, head, wo_cx, wo_cy, lv);
		c_q += lv; ++c_qq;
	}
	if (a.dc && c_kdq) { // instrumented passes only (pg_enable_depth_counters)
		atomicAdd(&a.dc->kd_levels, (unsigned long long)c_kd);
		atomicAdd(&a.dc->kd_queries, (unsigned long long)c_kdq);
		atomicAdd(&a.dc->quad_levels, (unsigned long long)c_q);
		atomicAdd(&a.dc->quad_queries, (unsigned long long)c_qq);
	}
	if (do_mis) { // :310-311
		woPdf = f * bsdf_pdf + (1.0f - f) * sdtree_pdf;
		bsdf_weight = vdivs(bsdf_value, woPdf);
		// deliberate deviation: 0/0 when a zero-energy tree proposes a direction below
		// the surface; the reference's throughput turns NaN there, here the path simply ends
		if (!(woPdf > 0.0f)) bsdf_weight = V(0, 0, 0);
	}
	// ---- :318-346 record ----
	// The reference's slot is ray*max_depth + depth (:318), a stride-max_depth scatter into a buffer
	// that is mostly empty at the deeper bounces.  The library's own buffer is a list instead: this
	// launch's thread t owns entry rec_slot = (records of the earlier bounces) + t, so a wavefront's
	// stores coalesce and the splat visits no empty tail; ray_of names the path (pg_process_and_splat
	// looks its final radiance up, :440), kNoRay marks a path that left the scene here.
	if (a.record) a.ray_of[rec_slot] = valid ? (uint32_t)lane : 0xffffffffu;
	if (do_record) {
		const uint64_t S = N * (uint64_t)D;
		const uint64_t g = rec_slot;
		a.r_pos[g] = p.x; a.r_pos[S + g] = p.y; a.r_pos[2 * S + g] = p.z;
		a.r_dir[g] = wo_cx; a.r_dir[S + g] = wo_cy;
		a.r_bsdf[g] = bsdf_weight.x; a.r_bsdf[S + g] = bsdf_weight.y; a.r_bsdf[2 * S + g] = bsdf_weight.z;
		a.r_tb[g] = thr.x; a.r_tb[S + g] = thr.y; a.r_tb[2 * S + g] = thr.z;
		a.r_tr[g] = L.x; a.r_tr[S + g] = L.y; a.r_tr[2 * S + g] = L.z;
		if (a.store_nee) {
			const v3 rn = vdiv(Lr_dir, thr);
			a.r_nee[g] = rn.x; a.r_nee[S + g] = rn.y; a.r_nee[2 * S + g] = rn.z;
			a.r_dnee[g] = nee_cx; a.r_dnee[S + g] = nee_cy;
		} else {
			a.r_nee[g] = 0.0f; a.r_nee[S + g] = 0.0f; a.r_nee[2 * S + g] = 0.0f;
			a.r_dnee[g] = 0.0f; a.r_dnee[S + g] = 0.0f;
		}
		a.r_wp[g] = woPdf;
	}
	// ---- :352-381 advance ----
	// ior (:357): without a dielectric every sampled direction has eta = 1, the running product stays
	// exactly 1 and is not carried; general scenes carry it
	if (kGeneral >= 2) ior = ior * eta;
	thr = vmul(thr, bsdf_weight);
	const float tmax = max3(thr);
	active_next = active_next && (tmax != 0.0f);
	float rr_prob = tmax * (ior * ior);
	if (!(rr_prob < 0.95f)) rr_prob = 0.95f;
	const bool rr_active = depth >= (uint32_t)a.rr_depth;
	const float rr = rng.next_f32(); // :377, unmasked
	const bool rr_continue = rr < rr_prob;
	active_next = active_next && (!rr_active || rr_continue);
	// ---- state for the next bounce; a path that ends here leaves only its radiance ----
	a.L[lane] = L.x; a.L[N + lane] = L.y; a.L[2 * N + lane] = L.z;
	if (kFirst) a.hit0[lane] = valid ? 1 : 0;
	if (active_next) {
		a.rng_state[lane] = rng.state;
		if (kFirst) a.rng_inc[lane] = rng.inc;
		a.ray_d[lane] = wo_world.x; a.ray_d[N + lane] = wo_world.y; a.ray_d[2 * N + lane] = wo_world.z;
		a.thr[lane] = thr.x; a.thr[N + lane] = thr.y; a.thr[2 * N + lane] = thr.z;
		a.prev_p[lane] = p.x; a.prev_p[N + lane] = p.y; a.prev_p[2 * N + lane] = p.z;
		a.prev_pdf[lane] = woPdf;
		a.prev_quad[lane] = (uint32_t)q | ((kGeneral >= 2 && delta) ? 0x80000000u : 0u);
		if (kGeneral >= 2) a.ior[lane] = ior;
	}
	return active_next;
}

// One bounce of the wavefront: thread t serves the t-th entry of the live-ray list the previous
// bounce wrote, runs the loop body, and the survivors of a workgroup append themselves to the
// next list with one atomic per workgroup (order inside a workgroup is kept, so neighbouring
// pixels stay neighbours; the order of workgroups is free -- every lane's result depends on its
// own state only).
// Waves per SIMD the register allocator aims at.  Levels 0 and 1 are left to the compiler (4 waves
// at 112-124 VGPRs; 5 or 6 measured no faster).  Level 2 would get 3 waves at 153 VGPRs; held to
// 5 (about 100 registers, the rest spilled) it is 10 % faster on the torus scene: 2 -> 15.1, 3 ->
// 11.5, 4 -> 10.9, 5 -> 10.4 ms per pass (make EXTRA=-DPG_BOUNCE_WAVES_L2=n to try others).
#ifndef PG_BOUNCE_WAVES_L2
#define PG_BOUNCE_WAVES_L2 5
#endif
#define PG_BOUNCE_ATTR(level) \
	__attribute__((amdgpu_waves_per_eu((level) >= 2 ? PG_BOUNCE_WAVES_L2 : 1, (level) >= 2 ? PG_BOUNCE_WAVES_L2 : 8)))

// The tail of a long path (max_depth 30 in scenes/torus): a launch cannot be shorter than the slowest
// single path's bounce (0.1-0.3 ms when that is two BVH walks inside a glass case), so a few
// thousand survivors would cost that floor once per bounce.  At fixed checkpoints the host also
// launches k_bounce_tail: when no more than kTailPaths paths are alive it takes all of them over
// and every lane follows its own path to its end in this one launch; the per-bounce launches after
// it find that out from the same counts and retire.
#ifndef PG_TAIL_PATHS
#define PG_TAIL_PATHS (128u * 1024u) // torus (tools/exp_tail.sh): 32 Ki -> 12.9, 128 Ki -> 12.4, 512 Ki -> 13.2, 2 Mi -> 13.2 ms per pass
#endif
constexpr uint32_t kTailPaths = PG_TAIL_PATHS;
__host__ __device__ constexpr bool tail_checkpoint(int bounce, int max_depth)
{
	return max_depth > 8 && bounce >= 4 && bounce + 1 < max_depth &&
	       (bounce < 8 || (bounce < 16 && bounce % 2 == 0) || bounce % 4 == 0);
}
// Did a tail launch at a checkpoint <= bounce take the paths over?  live_count[c-1] is final when
// checkpoint c is launched, and the first checkpoint that fires decides: the entries a tail launch
// adds to afterwards are never looked at before one that already said yes.
__device__ __forceinline__ bool tail_took_over(const RenderArgs &a, int bounce)
{
	if (a.max_depth <= 8) return false;
	for (int c = 4; c <= bounce; ++c)
		if (tail_checkpoint(c, a.max_depth) && a.live_count[c - 1] <= kTailPaths) return true;
	return false;
}

template <bool kFirst, int kGeneral>
__global__ __launch_bounds__(kRBlock) PG_BOUNCE_ATTR(kGeneral) void k_bounce(RenderArgs a)
{
	__shared__ uint4 s_kd[kLdsKdNodes];
	__shared__ uint32_t s_wave[kRBlock / 64];
	__shared__ uint32_t s_base;
	const uint64_t tid = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	const uint64_t live = kFirst ? a.n_lanes : (uint64_t)a.live_count[a.bounce - 1];
	if ((uint64_t)blockIdx.x * kRBlock >= live) return; // whole workgroup past the list
	if (!kFirst && tail_took_over(a, a.bounce)) return;  // a tail launch is finishing these paths
	if (a.guided) stage_kd_top(s_kd, a.tree.kd, a.tree.n_kd);
	const bool alive = tid < live;
	const uint64_t lane = alive ? (kFirst ? tid : (uint64_t)a.order_in[tid]) : 0;
	// records of the earlier bounces: all paths for the first, the survivors of bounce j for bounce j+1
	uint64_t rec_base = 0;
	if (!kFirst) {
		rec_base = a.n_lanes;
		for (int j = 0; j + 1 < a.bounce; ++j) rec_base += a.live_count[j];
	}
	bool cont = false;
	if (alive) cont = bounce_lane<kFirst, kGeneral>(a, s_kd, lane, rec_base + tid, (uint32_t)a.bounce);
	if (a.last) return; // nothing survives the last bounce
	const unsigned long long ballot = __ballot(cont);
	const unsigned wl = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	if (wl == 0) s_wave[wv] = (uint32_t)__popcll(ballot);
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tot = 0;
		for (int w = 0; w < kRBlock / 64; ++w) tot += s_wave[w];
		s_base = tot ? atomicAdd(&a.live_count[a.bounce], tot) : 0u;
	}
	__syncthreads();
	if (cont) {
		uint32_t off = s_base + (uint32_t)__popcll(ballot & ((1ull << wl) - 1ull));
		for (unsigned w = 0; w < wv; ++w) off += s_wave[w];
		a.order_out[off] = (uint32_t)lane;
	}
}

// See tail_checkpoint: launched before the per-bounce launch of bounce a.bounce with a grid for
// kTailPaths lanes.  The first bounce done here reads the live list like k_bounce would; after it a
// lane keeps its path (the state still goes through its global slots: the same thread reads back
// what it wrote).  Record entries of the later bounces are handed out wave by wave behind the
// entries of bounce a.bounce (a.live_count[max_depth] counts them), and the survivors of every
// bounce are added to live_count[] as the per-bounce launches would have, so the splat finds
// N + sum(live_count) entries and pg_render_live_counts reports the same numbers either way.
template <int kGeneral>
__global__ __launch_bounds__(kRBlock) PG_BOUNCE_ATTR(kGeneral) void k_bounce_tail(RenderArgs a)
{
	__shared__ uint4 s_kd[kLdsKdNodes];
	const uint64_t tid = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	const uint64_t live = (uint64_t)a.live_count[a.bounce - 1];
	if (live > kTailPaths || (uint64_t)blockIdx.x * kRBlock >= live) return;
	if (tail_took_over(a, a.bounce - 1)) return; // an earlier checkpoint already did
	if (a.guided) stage_kd_top(s_kd, a.tree.kd, a.tree.n_kd);
	bool alive = tid < live;
	const uint64_t lane = alive ? (uint64_t)a.order_in[tid] : 0;
	uint64_t rec_base = a.n_lanes; // entries of the bounces before a.bounce
	for (int j = 0; j + 1 < a.bounce; ++j) rec_base += a.live_count[j];
	const uint64_t tail_base = rec_base + live; // behind the entries of bounce a.bounce
	uint64_t slot = rec_base + tid;
	const unsigned wl = threadIdx.x & 63u;
	for (int depth = a.bounce; depth < a.max_depth; ++depth) {
		if (alive) alive = bounce_lane<false, kGeneral>(a, s_kd, lane, slot, (uint32_t)depth);
		const unsigned long long ballot = __ballot(alive);
		if (ballot == 0ull) break; // (nothing survives the last bounce)
		const uint32_t n = (uint32_t)__popcll(ballot);
		uint32_t off = 0;
		if (wl == (unsigned)__builtin_ctzll(ballot)) {
			atomicAdd(&a.live_count[depth], n);
			off = atomicAdd(&a.live_count[a.max_depth], n);
		}
		off = __shfl(off, __builtin_ctzll(ballot), 64);
		slot = tail_base + off + (uint32_t)__popcll(ballot & ((1ull << wl) - 1ull));
	}
}

// :400-431: valid flag and per-pixel sums, samples of a pixel added in lane order
__global__ __launch_bounds__(kRBlock) void k_finish(RenderArgs a, uint8_t *__restrict__ valid_out,
                                                    float *__restrict__ sumL, float *__restrict__ sumL2)
{
	const uint64_t pix = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	if (pix >= a.n_pixels) return;
	const uint64_t N = a.n_lanes, P = a.film_pixels;
	const uint64_t gpix = a.pixel_begin + pix; // the sums are full-film arrays
	const uint64_t first = pix * (uint64_t)a.spp;
	if (valid_out)
		for (int s = 0; s < a.spp; ++s) valid_out[first + s] = a.hit0[first + s];
	if (sumL && sumL2) {
		// the running sums stay in registers; the additions and their order are those of the reference
		float s1[3], s2[3];
		for (int c = 0; c < 3; ++c) { s1[c] = sumL[c * P + gpix]; s2[c] = sumL2[c * P + gpix]; }
		for (int s = 0; s < a.spp; ++s)
			for (int c = 0; c < 3; ++c) {
				const float v = a.L[c * N + first + s];
				s1[c] = s1[c] + v;
				s2[c] = s2[c] + v * v;
			}
		for (int c = 0; c < 3; ++c) { sumL[c * P + gpix] = s1[c]; sumL2[c * P + gpix] = s2[c]; }
	}
}

// element-wise evaluation of the library's deterministic fp32 functions (pg_math_eval)
__global__ __launch_bounds__(kRBlock) void k_math_eval(int which, uint64_t n, const float *__restrict__ x,
                                                       float *__restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	if (i >= n) return;
	const float v = x[i];
	float r, c;
	switch (which) {
	case 0: r = exp_f32(v); break;
	case 1: r = log_f32(v); break;
	case 2: r = erf_f32(v); break;
	case 3: r = erfinv_f32(v); break;
	case 4: sincos_f32(v, r, c); break;
	default: sincos_f32(v, c, r); break;
	}
	out[i] = r;
}

__device__ __forceinline__ float tent1(float d)
{
	const float a = 1.0f - fabs_(d);
	return a > 0.0f ? a : 0.0f;
}

// hdrfilm with <rfilter type="tent"/>: gather form of ImageBlock::put -- pixel (x,y) collects the
// samples of its 3x3 neighbourhood (rows, then columns, then samples, ascending), so the sums have
// one order and need no atomics.  The film position of a sample is recomputed from its stream.
// Mitsuba's gaussian rfilter (hdrfilm's default, scenes/torus/scene.xml:46): stddev 0.5, radius 2
__device__ __forceinline__ float gauss1(float d)
{
	const float a = exp_f32(-2.0f * (d * d)) - exp_f32(-8.0f);
	return a > 0.0f ? a : 0.0f;
}

template <int kFilter> // 0 tent (3x3 neighbourhood), 1 gaussian (5x5)
__global__ __launch_bounds__(kRBlock) void k_film(uint32_t seed, int spp, int W, int H,
                                                  const float *__restrict__ L, float *__restrict__ out)
{
	constexpr int R = kFilter == 1 ? 2 : 1;
	const uint64_t npix = (uint64_t)W * (uint64_t)H, N = npix * (uint64_t)spp;
	const uint64_t o = (uint64_t)blockIdx.x * kRBlock + threadIdx.x;
	if (o >= npix) return;
	const int x = (int)(o % (uint64_t)W), y = (int)(o / (uint64_t)W);
	const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
	float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, wsum = 0.0f;
	for (int ny = y - R; ny <= y + R; ++ny)
		for (int nx = x - R; nx <= x + R; ++nx) {
			if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
			const uint64_t pix = (uint64_t)ny * (uint64_t)W + (uint64_t)nx;
			for (int s = 0; s < spp; ++s) {
				const uint64_t lane = pix * (uint64_t)spp + (uint64_t)s;
				Pcg32 rng = pcg32_seed(seed, (uint32_t)lane);
				const float jx = rng.next_f32(), jy = rng.next_f32();
				const float ddx = cx - ((float)nx + jx), ddy = cy - ((float)ny + jy);
				const float w = kFilter == 1 ? gauss1(ddx) * gauss1(ddy) : tent1(ddx) * tent1(ddy);
				a0 = a0 + w * L[lane];
				a1 = a1 + w * L[N + lane];
				a2 = a2 + w * L[2 * N + lane];
				wsum = wsum + w;
			}
		}
	const bool ok = wsum > 0.0f;
	out[o] = ok ? a0 / wsum : 0.0f;
	out[npix + o] = ok ? a1 / wsum : 0.0f;
	out[2 * npix + o] = ok ? a2 / wsum : 0.0f;
}

} // namespace pg

using namespace pg;

// library-owned renderer state
struct pg_render_state {
	DevBuf<float> quads, spheres, mats, boxes, tris, dir_lights, ior, tri_normals;
	bool have_tri_normals = false;
	float bsphere[4] = {0, 0, 0, 0};
	DevBuf<uint32_t> bvh;
	DevBuf<int32_t> emitters;
	int n_quads = 0, n_spheres = 0, n_emitters = 0, n_boxes = 0, n_bvh_nodes = 0;
	int general = 0; // feature level of the kernels to launch (0 cornell-box class, 1 veach-mis class, 2 everything)
	pg_camera cam;
	bool have_scene = false;
	DevBuf<float> ray_d, thr, prev_p, prev_pdf;
	DevBuf<uint32_t> prev_quad;
	DevBuf<uint8_t> hit0;
	DevBuf<uint64_t> rng_state, rng_inc;
	DevBuf<uint32_t> order[2], live_count;
	DevBuf<uint32_t> ray_of;
	DevBuf<float> r_pos, r_dir, r_bsdf, r_tb, r_tr, r_nee, r_dnee, r_wp;
	// optional per-kernel timing: (kind, start, stop) event triples still to be read
	bool timing_on = false;
	struct Ev { int kind; hipEvent_t a, b; };
	std::vector<Ev> events;
	pg_kernel_timing acc = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {
struct Timed { // records an event pair around a launch when timing is enabled
	pg_render_state *r;
	hipStream_t s;
	int kind;
	hipEvent_t a = nullptr, b = nullptr;
	Timed(pg_render_state *r_, hipStream_t s_, int kind_) : r(r_), s(s_), kind(kind_)
	{
		if (!r->timing_on) return;
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
		(void)hipEventRecord(a, s);
	}
	~Timed()
	{
		if (!a) return;
		(void)hipEventRecord(b, s);
		r->events.push_back({kind, a, b});
	}
};
} // namespace

static pg_render_state *rstate(pg_context *ctx)
{
	if (!ctx->render) ctx->render = new pg_render_state();
	return ctx->render;
}

void pg::destroy_render_state(pg_context *ctx)
{
	delete ctx->render;
	ctx->render = nullptr;
}

extern "C" {

int pg_scene_set(pg_context *ctx, uint64_t n_quads, const float *h_quads, const pg_camera *cam)
{
	pg_scene_desc d;
	d.n_quads = n_quads; d.quads = h_quads;
	d.n_spheres = 0; d.spheres = nullptr;
	d.n_materials = 0; d.materials = nullptr;
	d.n_boxes = 0; d.boxes = nullptr;
	d.n_tris = 0; d.tris = nullptr;
	d.n_bvh_nodes = 0; d.bvh = nullptr;
	d.n_dir_lights = 0; d.dir_lights = nullptr;
	d.bsphere[0] = d.bsphere[1] = d.bsphere[2] = d.bsphere[3] = 0.0f;
	d.tri_normals = nullptr;
	return pg_scene_set_ex(ctx, &d, cam);
}

int pg_scene_set_ex(pg_context *ctx, const pg_scene_desc *sc, const pg_camera *cam)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!sc || !cam) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: NULL pointer");
	const uint64_t nq = sc->n_quads, ns = sc->n_spheres, nm = sc->n_materials, nb = sc->n_boxes;
	if (nq + ns + nb + sc->n_tris == 0 || nq > 4096 || ns > 4096 || nb > 4096 || (nq && !sc->quads) || (ns && !sc->spheres) || (nb && !sc->boxes))
		return fail(ctx, PG_ERR_INVALID, "pg_scene_set: need 1..4096 quads, spheres and/or boxes");
	if ((nm && !sc->materials) || (!sc->materials && (ns || nb)))
		return fail(ctx, PG_ERR_INVALID, "pg_scene_set: spheres and boxes need a material table");
	if (cam->width <= 0 || cam->height <= 0) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: bad film size");
	// host copies: material indices checked, a material table made up for scenes that come without one,
	// the diffuse reflectance mirrored into the quads (the quad-only kernels read it there)
	std::vector<float> quads(sc->quads, sc->quads + nq * kQuadStride);
	std::vector<float> mats;
	if (sc->materials) {
		mats.assign(sc->materials, sc->materials + nm * kMaterialStride);
	} else {
		mats.assign(nq * kMaterialStride, 0.0f);
		for (uint64_t q = 0; q < nq; ++q) {
			for (int c = 0; c < 3; ++c) mats[q * kMaterialStride + 1 + c] = quads[q * kQuadStride + 16 + c];
			quads[q * kQuadStride + 22] = (float)q;
		}
	}
	const uint64_t n_mats = mats.size() / kMaterialStride;
	int general = ns > 0 ? 1 : 0; // feature level, see intersect()
	for (uint64_t m = 0; m < n_mats; ++m) {
		const float type = mats[m * kMaterialStride];
		if (type != 0.0f && type != 1.0f && type != 2.0f && type != 3.0f && type != 4.0f)
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: unknown material type");
		if (type == 1.0f && general < 1) general = 1;                                  // rough conductor
		if (type >= 2.0f || mats[m * kMaterialStride + 11] != 0.0f) general = 2;       // transmission, delta lobes, one-sided BSDFs
		const float alpha = mats[m * kMaterialStride + 4]; // > 0: Beckmann, < 0: GGX of roughness -alpha
		if ((type == 1.0f || type == 4.0f) && !(fabsf(alpha) > 0.0f && fabsf(alpha) < 3.0e38f))
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: microfacet alpha must be finite and not 0");
		if ((type == 3.0f || type == 4.0f) && !(mats[m * kMaterialStride + 5] > 0.0f))
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: dielectric index ratio must be > 0");
	}
	const uint64_t nd = sc->n_dir_lights;
	if (nd > 64 || (nd && !sc->dir_lights)) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: at most 64 directional lights");
	if (nd && !(sc->bsphere[3] > 0.0f)) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: directional lights need the scene's bounding sphere");
	if (nd) general = 2;
	for (uint64_t q = 0; q < nq; ++q) {
		const float mi = quads[q * kQuadStride + 22];
		if (!(mi >= 0.0f && mi < (float)n_mats) || mi != (float)(uint64_t)mi)
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: quad material index out of range");
		const float *M = &mats[(uint64_t)mi * kMaterialStride];
		if (M[0] == 0.0f)
			for (int c = 0; c < 3; ++c) quads[q * kQuadStride + 16 + c] = M[1 + c];
	}
	for (uint64_t s = 0; s < ns; ++s) {
		const float mi = sc->spheres[s * kSphereStride + 4];
		if (!(mi >= 0.0f && mi < (float)n_mats) || mi != (float)(uint64_t)mi)
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: sphere material index out of range");
		if (!(sc->spheres[s * kSphereStride + 3] > 0.0f)) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: sphere radius must be > 0");
	}
	for (uint64_t b = 0; b < nb; ++b) {
		const float mi = sc->boxes[b * kBoxStride + 21];
		if (!(mi >= 0.0f && mi < (float)n_mats) || mi != (float)(uint64_t)mi)
			return fail(ctx, PG_ERR_INVALID, "pg_scene_set: box material index out of range");
		for (int k = 0; k < 21; ++k)
			if (!(sc->boxes[b * kBoxStride + k] == sc->boxes[b * kBoxStride + k]) || fabsf(sc->boxes[b * kBoxStride + k]) > 3.0e38f)
				return fail(ctx, PG_ERR_INVALID, "pg_scene_set: box transform is not finite");
	}
	// triangle meshes: the kernels walk the BVH with a fixed-size stack and trust it, so check it here:
	// children follow their parent (no cycles, one parent each), leaves stay inside the triangle
	// array, and no walk can have more than 64 siblings waiting on its stack
	const uint64_t nt = sc->n_tris, nn = sc->n_bvh_nodes;
	if ((nt == 0) != (nn == 0) || (nt && (!sc->tris || !sc->bvh)) || nt > 0x0fffffffull || nn > 0x7fffffffull)
		return fail(ctx, PG_ERR_INVALID, "pg_scene_set: triangles and BVH nodes go together");
	if (nt && !sc->materials) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: meshes need a material table");
	if (nn) {
		std::vector<uint8_t> waiting(nn, 0); // siblings on the stack when the walk opens node i, at most
		std::vector<uint8_t> seen(nn, 0);
		seen[0] = 1;
		for (uint64_t i = 0; i < nn; ++i) {
			if (!seen[i]) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH node without a parent");
			const uint32_t *N = sc->bvh + i * kBvhStride;
			int kids = 0;
			for (int c = 0; c < 4; ++c) kids += N[24 + c] != 0xffffffffu;
			if (kids == 0) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH node without children");
			const int below = (int)waiting[i] + kids - 1; // its other children wait while the walk is in one of them
			if (below > 64) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH too deep for the walk's stack");
			for (int c = 0; c < 4; ++c) {
				const uint32_t ref = N[24 + c];
				if (ref == 0xffffffffu) continue;
				if (ref & 0x80000000u) {
					const uint64_t first = ref & 0x0fffffffu, count = ((ref >> 28) & 7u) + 1u;
					if (first + count > nt) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH leaf outside the triangle array");
				} else {
					if (ref <= i || ref >= nn) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH children must follow their parent");
					if (seen[ref]) return fail(ctx, PG_ERR_INVALID, "pg_scene_set: BVH node with two parents");
					seen[ref] = 1;
					waiting[ref] = (uint8_t)below;
				}
			}
		}
		for (uint64_t t = 0; t < nt; ++t) {
			const float mi = sc->tris[t * kTriStride + 12];
			if (!(mi >= 0.0f && mi < (float)n_mats) || mi != (float)(uint64_t)mi)
				return fail(ctx, PG_ERR_INVALID, "pg_scene_set: triangle material index out of range");
		}
		general = 2;
	}
	std::vector<int32_t> em;
	for (uint64_t q = 0; q < nq; ++q)
		if (quads[q * kQuadStride + 15] != 0.0f) em.push_back((int32_t)q);
	for (uint64_t s = 0; s < ns; ++s)
		if (sc->spheres[s * kSphereStride + 5] != 0.0f) em.push_back((int32_t)(nq + s));
	for (uint64_t k = 0; k < nd; ++k) em.push_back(-1 - (int32_t)k);
	PG_HIP(ctx, hipSetDevice(ctx->device));
	pg_render_state *r = rstate(ctx);
	PG_HIP(ctx, r->dir_lights.ensure(nd * 8));
	if (nd) PG_HIP(ctx, hipMemcpy(r->dir_lights.p, sc->dir_lights, nd * 8 * sizeof(float), hipMemcpyHostToDevice));
	for (int c = 0; c < 4; ++c) r->bsphere[c] = sc->bsphere[c];
	PG_HIP(ctx, r->quads.ensure(nq * kQuadStride)); PG_HIP(ctx, r->spheres.ensure(ns * kSphereStride));
	PG_HIP(ctx, r->mats.ensure(mats.size())); PG_HIP(ctx, r->emitters.ensure(em.size()));
	PG_HIP(ctx, r->boxes.ensure(nb * kBoxStride));
	if (nb) PG_HIP(ctx, hipMemcpy(r->boxes.p, sc->boxes, nb * kBoxStride * sizeof(float), hipMemcpyHostToDevice));
	r->n_boxes = (int)nb;
	PG_HIP(ctx, r->tris.ensure(nt * kTriStride)); PG_HIP(ctx, r->bvh.ensure(nn * kBvhStride));
	if (nt) PG_HIP(ctx, hipMemcpy(r->tris.p, sc->tris, nt * kTriStride * sizeof(float), hipMemcpyHostToDevice));
	if (nn) PG_HIP(ctx, hipMemcpy(r->bvh.p, sc->bvh, nn * kBvhStride * sizeof(uint32_t), hipMemcpyHostToDevice));
	r->n_bvh_nodes = (int)nn;
	r->have_tri_normals = nt && sc->tri_normals;
	if (r->have_tri_normals) {
		PG_HIP(ctx, r->tri_normals.ensure(nt * 9));
		PG_HIP(ctx, hipMemcpy(r->tri_normals.p, sc->tri_normals, nt * 9 * sizeof(float), hipMemcpyHostToDevice));
	}
	if (nq) PG_HIP(ctx, hipMemcpy(r->quads.p, quads.data(), quads.size() * sizeof(float), hipMemcpyHostToDevice));
	if (ns) PG_HIP(ctx, hipMemcpy(r->spheres.p, sc->spheres, ns * kSphereStride * sizeof(float), hipMemcpyHostToDevice));
	PG_HIP(ctx, hipMemcpy(r->mats.p, mats.data(), mats.size() * sizeof(float), hipMemcpyHostToDevice));
	if (!em.empty()) PG_HIP(ctx, hipMemcpy(r->emitters.p, em.data(), em.size() * sizeof(int32_t), hipMemcpyHostToDevice));
	r->n_quads = (int)nq;
	r->n_spheres = (int)ns;
	r->n_emitters = (int)em.size();
	r->general = general;
	r->cam = *cam;
	r->have_scene = true;
	return PG_OK;
}

int pg_render_pass(pg_context *ctx, const pg_pass_params *prm, float *L_out, uint8_t *valid_out, float *sumL,
                   float *sumL2, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!ctx->configured) return fail(ctx, PG_ERR_INVALID, "call pg_setup or pg_import first");
	if (!ctx->render || !ctx->render->have_scene) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: call pg_scene_set first");
	if (!prm || !L_out) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: NULL pointer");
	if (prm->spp <= 0 || ctx->max_depth <= 0) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: spp and max_depth must be > 0");
	if ((sumL == nullptr) != (sumL2 == nullptr)) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: sumL and sumL2 go together");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t s = (hipStream_t)stream;
	pg_render_state *r = ctx->render;
	const uint64_t film = (uint64_t)r->cam.width * (uint64_t)r->cam.height;
	if (prm->pixel_begin > film || prm->pixel_count > film - prm->pixel_begin)
		return fail(ctx, PG_ERR_INVALID, "pg_render_pass: pixel range outside the film");
	const uint64_t P = prm->pixel_count ? prm->pixel_count : film - prm->pixel_begin; // 0 = to the end
	if (P == 0) return PG_OK;
	const uint64_t N = P * (uint64_t)prm->spp;
	const int D = ctx->max_depth;
	const uint64_t S = N * (uint64_t)D;
	if (S > 0xffffffffull) return fail(ctx, PG_ERR_INVALID, "pg_render_pass: more than 2^32 record slots in one pass");
	const bool record = !ctx->is_final;
	PG_HIP(ctx, r->ray_d.ensure(3 * N)); PG_HIP(ctx, r->thr.ensure(3 * N));
	PG_HIP(ctx, r->prev_p.ensure(3 * N)); PG_HIP(ctx, r->prev_pdf.ensure(N));
	PG_HIP(ctx, r->prev_quad.ensure(N)); PG_HIP(ctx, r->hit0.ensure(N));
	if (r->general >= 2) PG_HIP(ctx, r->ior.ensure(N));
	PG_HIP(ctx, r->rng_state.ensure(N)); PG_HIP(ctx, r->rng_inc.ensure(N));
	PG_HIP(ctx, r->order[0].ensure(N)); PG_HIP(ctx, r->order[1].ensure(N)); PG_HIP(ctx, r->live_count.ensure((uint64_t)D + 1));
	PG_HIP(ctx, hipMemsetAsync(r->live_count.p, 0, ((size_t)D + 1) * sizeof(uint32_t), s)); // [D]: entries handed out by k_bounce_tail
	if (record) {
		PG_HIP(ctx, r->ray_of.ensure(S)); PG_HIP(ctx, r->r_pos.ensure(3 * S)); PG_HIP(ctx, r->r_dir.ensure(2 * S));
		PG_HIP(ctx, r->r_bsdf.ensure(3 * S)); PG_HIP(ctx, r->r_tb.ensure(3 * S)); PG_HIP(ctx, r->r_tr.ensure(3 * S));
		PG_HIP(ctx, r->r_nee.ensure(3 * S)); PG_HIP(ctx, r->r_dnee.ensure(2 * S)); PG_HIP(ctx, r->r_wp.ensure(S));
	}
	RenderArgs a;
	a.tree = ctx->view();
	a.shapes.quads = r->quads.p;
	a.shapes.spheres = r->spheres.p;
	a.shapes.boxes = r->boxes.p;
	a.shapes.tris = r->tris.p;
	a.shapes.tri_normals = r->have_tri_normals ? r->tri_normals.p : nullptr;
	a.shapes.bvh = r->bvh.p;
	a.shapes.n_bvh_nodes = r->n_bvh_nodes;
	a.shapes.n_quads = r->n_quads;
	a.shapes.n_spheres = r->n_spheres;
	a.shapes.n_boxes = r->n_boxes;
	a.mats = r->mats.p;
	a.emitters = r->emitters.p;
	a.n_emitters = r->n_emitters;
	a.dir_lights.lights = r->dir_lights.p;
	for (int c = 0; c < 4; ++c) a.dir_lights.bsphere[c] = r->bsphere[c];
	a.ior = r->ior.p;
	a.cam = r->cam;
	a.n_lanes = N;
	a.n_pixels = P;
	a.pixel_begin = prm->pixel_begin;
	a.film_pixels = film;
	a.spp = prm->spp;
	a.max_depth = D;
	a.rr_depth = prm->rr_depth;
	a.guided = ctx->iteration > 1 ? 1 : 0; // :223, 250, 283
	a.record = record ? 1 : 0;
	a.store_nee = ctx->store_nee;
	a.frac = ctx->bsdf_fraction;
	a.seed = prm->seed;
	a.dc = ctx->dc_on ? ctx->dc : nullptr;
	a.ray_d = r->ray_d.p; a.thr = r->thr.p; a.L = L_out; a.prev_p = r->prev_p.p;
	a.prev_pdf = r->prev_pdf.p; a.prev_quad = r->prev_quad.p; a.hit0 = r->hit0.p;
	a.rng_state = r->rng_state.p; a.rng_inc = r->rng_inc.p; a.live_count = r->live_count.p;
	a.ray_of = r->ray_of.p; a.r_pos = r->r_pos.p; a.r_dir = r->r_dir.p; a.r_bsdf = r->r_bsdf.p; a.r_tb = r->r_tb.p;
	a.r_tr = r->r_tr.p; a.r_nee = r->r_nee.p; a.r_dnee = r->r_dnee.p; a.r_wp = r->r_wp.p;
	const dim3 grid((unsigned)((N + kRBlock - 1) / kRBlock));
	for (int it = 0; it < D; ++it) {
		a.bounce = it;
		a.last = it + 1 == D ? 1 : 0;
		a.order_in = r->order[it & 1].p;
		a.order_out = r->order[(it + 1) & 1].p;
		Timed t(r, s, 1);
		if (tail_checkpoint(it, D)) { // finishes every path in one launch once few are left (see k_bounce_tail)
			const dim3 tgrid((unsigned)((kTailPaths + kRBlock - 1) / kRBlock));
			if (r->general >= 2) hipLaunchKernelGGL((k_bounce_tail<2>), tgrid, dim3(kRBlock), 0, s, a);
			else if (r->general == 1) hipLaunchKernelGGL((k_bounce_tail<1>), tgrid, dim3(kRBlock), 0, s, a);
			else hipLaunchKernelGGL((k_bounce_tail<0>), tgrid, dim3(kRBlock), 0, s, a);
		}
		// every launch is sized for the whole wavefront: the live count is only known on the device,
		// and workgroups past it retire on their first instruction
		if (r->general >= 2) {
			if (it == 0) hipLaunchKernelGGL((k_bounce<true, 2>), grid, dim3(kRBlock), 0, s, a);
			else hipLaunchKernelGGL((k_bounce<false, 2>), grid, dim3(kRBlock), 0, s, a);
		} else if (r->general == 1) {
			if (it == 0) hipLaunchKernelGGL((k_bounce<true, 1>), grid, dim3(kRBlock), 0, s, a);
			else hipLaunchKernelGGL((k_bounce<false, 1>), grid, dim3(kRBlock), 0, s, a);
		} else {
			if (it == 0) hipLaunchKernelGGL((k_bounce<true, 0>), grid, dim3(kRBlock), 0, s, a);
			else hipLaunchKernelGGL((k_bounce<false, 0>), grid, dim3(kRBlock), 0, s, a);
		}
	}
	PG_HIP(ctx, hipGetLastError());
	if (record) {
		pg_dense_records d;
		d.active = nullptr; d.position = r->r_pos.p; d.direction = r->r_dir.p; d.bsdf = r->r_bsdf.p;
		d.throughput_bsdf = r->r_tb.p; d.throughput_radiance = r->r_tr.p; d.radiance_nee = r->r_nee.p;
		d.direction_nee = r->r_dnee.p; d.wo_pdf = r->r_wp.p;
		Timed t(r, s, 2);
		// the depth counters of an instrumented pass describe the bounce kernels only
		launch_process_and_splat(ctx->view(), ctx->f.accum_view(), ctx->store_nee, N, D, L_out, d, nullptr, s,
		                         r->ray_of.p, r->live_count.p);
		PG_HIP(ctx, hipGetLastError());
	}
	if (valid_out || sumL) {
		Timed t(r, s, 3);
		hipLaunchKernelGGL(k_finish, dim3((unsigned)((P + kRBlock - 1) / kRBlock)), dim3(kRBlock), 0, s, a, valid_out,
		                   sumL, sumL2);
		PG_HIP(ctx, hipGetLastError());
	}
	if (r->timing_on) ++r->acc.passes;
	return PG_OK;
}

int pg_film_tent(pg_context *ctx, uint32_t seed, int32_t spp, const float *L, float *image_out, void *stream)
{
	return pg_film(ctx, PG_FILTER_TENT, seed, spp, L, image_out, stream);
}

int pg_film(pg_context *ctx, int32_t filter, uint32_t seed, int32_t spp, const float *L, float *image_out, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	if (!ctx->render || !ctx->render->have_scene) return fail(ctx, PG_ERR_INVALID, "pg_film: call pg_scene_set first");
	if (!L || !image_out || spp <= 0) return fail(ctx, PG_ERR_INVALID, "pg_film: NULL pointer or spp <= 0");
	if (filter != PG_FILTER_TENT && filter != PG_FILTER_GAUSSIAN) return fail(ctx, PG_ERR_INVALID, "pg_film: unknown filter");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	const pg_camera &cam = ctx->render->cam;
	const uint64_t npix = (uint64_t)cam.width * (uint64_t)cam.height;
	const dim3 grid((unsigned)((npix + kRBlock - 1) / kRBlock));
	if (filter == PG_FILTER_GAUSSIAN)
		hipLaunchKernelGGL(k_film<1>, grid, dim3(kRBlock), 0, (hipStream_t)stream, seed, spp, cam.width, cam.height, L, image_out);
	else
		hipLaunchKernelGGL(k_film<0>, grid, dim3(kRBlock), 0, (hipStream_t)stream, seed, spp, cam.width, cam.height, L, image_out);
	PG_HIP(ctx, hipGetLastError());
	return PG_OK;
}

int pg_math_eval(pg_context *ctx, int32_t which, uint64_t n, const float *x, float *out, void *stream)
{
	if (!ctx) return PG_ERR_INVALID;
	if (which < 0 || which > 5 || (n && (!x || !out))) return fail(ctx, PG_ERR_INVALID, "pg_math_eval: bad arguments");
	if (n == 0) return PG_OK;
	PG_HIP(ctx, hipSetDevice(ctx->device));
	hipLaunchKernelGGL(k_math_eval, dim3((unsigned)((n + kRBlock - 1) / kRBlock)), dim3(kRBlock), 0, (hipStream_t)stream,
	                   which, n, x, out);
	PG_HIP(ctx, hipGetLastError());
	return PG_OK;
}

int pg_render_live_counts(pg_context *ctx, uint32_t *out, int32_t n)
{
	if (!ctx || !out || n < 0) return PG_ERR_INVALID;
	if (!ctx->render || !ctx->render->live_count.p) return fail(ctx, PG_ERR_INVALID, "pg_render_live_counts: no pass rendered yet");
	PG_HIP(ctx, hipSetDevice(ctx->device));
	PG_HIP(ctx, hipDeviceSynchronize());
	int m = n < ctx->max_depth ? n : ctx->max_depth;
	if ((size_t)m > ctx->render->live_count.cap) m = (int)ctx->render->live_count.cap;
	PG_HIP(ctx, hipMemcpy(out, ctx->render->live_count.p, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToHost));
	return PG_OK;
}

int pg_enable_kernel_timing(pg_context *ctx, int32_t on)
{
	if (!ctx) return PG_ERR_INVALID;
	rstate(ctx)->timing_on = on != 0;
	return PG_OK;
}

int pg_read_kernel_timing(pg_context *ctx, pg_kernel_timing *out, int32_t reset)
{
	if (!ctx || !out) return PG_ERR_INVALID;
	PG_HIP(ctx, hipSetDevice(ctx->device));
	pg_render_state *r = rstate(ctx);
	for (auto &e : r->events) {
		float ms = 0.0f;
		PG_HIP(ctx, hipEventSynchronize(e.b));
		PG_HIP(ctx, hipEventElapsedTime(&ms, e.a, e.b));
		switch (e.kind) {
		case 0: r->acc.generate_ms += ms; break;
		case 1: r->acc.bounce_ms += ms; ++r->acc.bounce_launches; break;
		case 2: r->acc.splat_ms += ms; ++r->acc.splat_launches; break;
		case 4: r->acc.compact_ms += ms; break;
		default: r->acc.finish_ms += ms; break;
		}
		(void)hipEventDestroy(e.a);
		(void)hipEventDestroy(e.b);
	}
	r->events.clear();
	*out = r->acc;
	if (reset) r->acc = pg_kernel_timing{0, 0, 0, 0, 0, 0, 0, 0};
	return PG_OK;
}

} // extern "C"
