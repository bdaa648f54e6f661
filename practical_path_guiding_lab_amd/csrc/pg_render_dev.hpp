// pg_render_dev.hpp -- device functions of the renderer substrate shared by the bounce kernels
// (pg_render.hip: the fused per-bounce kernel of quad/sphere/box scenes; pg_render_wave.hip: the
// split wavefront pipeline of mesh scenes): vectors, frames, ray casting, surfaces, textures, BSDFs,
// emitter sampling, and the argument block of a pass.  Arithmetic mirrors oracle/pg_oracle_render.c
// operation by operation (fp32, no contraction).
#pragma once

#include "pg_context.hpp"
#include "pg_descent.hpp"
#include "pg_kernels.hpp"

// The microfacet helpers are called from several places and inlined at every one of them: a call needs a stack frame in
// scratch memory, and no kernel of the library is to use any.  (Rounds 1-4 kept them out of line in the fused kernel of
// pg_render.hip: 32 bytes of scratch per lane in its level-1 instantiations.)
#define PG_OUTLINE __forceinline__

namespace pg {

constexpr int kRBlock = 256; // threads of a workgroup in every render kernel
static_assert(kRBlock == kStageThreads, "stage_kd_planes copies one plane per thread");
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
constexpr int kMaterialStride = 16;                     // PG_MATERIAL_STRIDE
constexpr int kTextureStride = 16;                      // PG_TEXTURE_STRIDE (32-bit words)

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
	const float *tri_uvs;     // 6 per triangle (uv0 uv1 uv2), or nullptr
	const uint32_t *bvh;
	const uint32_t *textures; // kTextureStride words each
	const uint32_t *texels;   // RGBA8 sRGB texels of all bitmaps
	const float *srgb_lut;    // 256 floats: 8-bit sRGB -> linear
	int n_quads, n_spheres, n_boxes, n_bvh_nodes;
};

// The stack of the BVH walk: no per-lane array (that would live in scratch memory).  The first
// kLdsStack entries of a lane sit in LDS, column threadIdx.x of a [kLdsStack][kRBlock] array (a
// wave's accesses to one level hit 64 different banks); the rare deeper ones go to a per-lane strip
// of a global workspace.  An entry is (reference, entry distance).  pg_scene_set_ex checks that no
// walk can need more than kLdsStack + kOvfStack entries.
constexpr int kLdsStack = 8;  // entries of a lane in LDS in the ray-casting kernels (k_wave_shade keeps fewer: BvhStack::n_lds)
constexpr int kOvfStack = 28; // entries of a lane's overflow strip: 32 (what pg_scene_set_ex admits) minus the fewest any kernel keeps in LDS
constexpr int kMinLdsStack = 4;
typedef __attribute__((address_space(3))) uint32_t LdsWord; // an LDS pointer stays one: ds_read / ds_write, never flat
typedef __attribute__((address_space(3))) u32x4_t LdsQuad;
struct BvhStack {
	LdsWord *lds; // the two words of &s_stack[0][threadIdx.x]; level sp is 2 * kRBlock words further
	uint2 *ovf;   // the workspace of overflow strips (uniform) ...
	uint32_t ovf_first; // ... and this lane's first entry in it, kOvfStack of them: an index, not a pointer -- one register
	                    // instead of two alive through the whole walk (pg_render_pass keeps n_lanes * kOvfStack below 2^32)
	const LdsQuad *top; // the first n_top nodes of the BVH, which the kernel has copied into LDS (stage_bvh_top); 0: none
	uint32_t n_top;
	int n_lds;          // entries of this lane that sit in LDS (a constant of the kernel: kLdsStack, or fewer where LDS is short)
	__device__ __forceinline__ void push(int sp, uint32_t ref, float t) const
	{
		if (sp < n_lds) {
			lds[sp * (2 * kRBlock)] = ref;
			lds[sp * (2 * kRBlock) + 1] = __float_as_uint(t);
		} else ovf[ovf_first + (uint32_t)(sp - n_lds)] = make_uint2(ref, __float_as_uint(t));
	}
	__device__ __forceinline__ uint2 at(int sp) const
	{
		// the LDS read is unconditional (of a clamped level) and volatile so that it stays a ds_read: two
		// loads in two branches get folded into one flat load of a selected address
		const volatile LdsWord *w = lds + (sp < n_lds ? sp : n_lds - 1) * (2 * kRBlock);
		uint2 e = make_uint2(w[0], w[1]);
		if (sp >= n_lds) e = ovf[ovf_first + (uint32_t)(sp - n_lds)];
		return e;
	}
};
__device__ __forceinline__ BvhStack bvh_stack(uint2 *lds_column, uint2 *ovf, uint32_t ovf_first, int n_lds = kLdsStack)
{
	BvhStack s;
	s.n_lds = n_lds;
	s.lds = (LdsWord *)lds_column;
	s.ovf = ovf;
	s.ovf_first = ovf_first;
	s.top = nullptr; s.n_top = 0;
	return s;
}

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

// The same test with the near and far planes already picked (the walk loads the rows of a node in
// the order the ray's signs dictate): identical arithmetic, no selects.
__device__ __forceinline__ bool bvh_box_hit_nf(float nxp, float nyp, float nzp, float fxp, float fyp, float fzp, v3 o, v3 inv,
                                               float bt, float &tmin_out)
{
	const float nx = (nxp - o.x) * inv.x, fx = (fxp - o.x) * inv.x;
	const float ny = (nyp - o.y) * inv.y, fy = (fyp - o.y) * inv.y;
	const float nz = (nzp - o.z) * inv.z, fz = (fzp - o.z) * inv.z;
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

// ---- ray casting ----
// kAny: the caller asks whether anything is hit (shadow rays): the BVH walk stops at its first
// triangle.  The answer is that of the closest-hit walk, which visits the same nodes until then.

// The shapes outside the BVH (quads, spheres, boxes), tested one after the other: updates bt / best.
template <int kGeneral>
__device__ __forceinline__ void intersect_linear(const Shapes &sh, v3 o, v3 d, float &bt, int &best)
{
	const int nq = sh.n_quads;
	const float *__restrict__ quads = sh.quads;
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
}

// The walk through the four-wide BVH as three steps on a small state, so that a ray can be walked by
// a loop of its own (intersect) or a few steps at a time (the persistent kernels of pg_render_wave.hip,
// which hand a lane a new ray as soon as its old one is done).  One 128-byte node holds the boxes of
// its (up to four) children: they are tested together and ordered by where the ray enters them (a
// fixed five-comparator network), the walk goes on in the nearest -- a leaf's triangles are named by
// the reference itself, no node is read for it -- and the others wait on the stack with their entry
// distance, farthest at the bottom, to be dropped when popped if the ray has become shorter than that.
// The oracle visits the same nodes in the same order, so the first of several equally near triangles
// is the same one in both.  pg_scene_set_ex has checked the tree: children follow their parent, and no
// root-to-node path can leave more than kLdsStack + kOvfStack siblings waiting, so the walk opens every
// node at most once and the stack cannot overflow; the budget is a second fence.
constexpr uint32_t kBvhNone = 0xffffffffu;
struct BvhWalk {
	v3 o, d, inv;
	int row_nx, row_fx, row_ny, row_fy, row_nz, row_fz; // byte offsets of the rows of a node that hold the near / far planes for this ray
	                                                    // (a kSlim walk does not keep them: bvh_rows)
	float bt, bu, bv;
	int best, sp, budget;
	uint32_t next; // a node, a leaf (bit 31), or kBvhNone: take the next candidate from the stack
};
// the six offsets from the signs of the ray's direction (bvh_begin keeps them in six registers; a walk inside a kernel that is
// short of registers -- k_wave_shade -- makes them again at every node: six selects against seven gathers)
__device__ __forceinline__ void bvh_rows(v3 d, int &nx, int &fx, int &ny, int &fy, int &nz, int &fz)
{
	const bool ngx = (__float_as_uint(d.x) >> 31) != 0u, ngy = (__float_as_uint(d.y) >> 31) != 0u, ngz = (__float_as_uint(d.z) >> 31) != 0u;
	nx = ngx ? 48 : 0; fx = ngx ? 0 : 48; ny = ngy ? 64 : 16; fy = ngy ? 16 : 64; nz = ngz ? 80 : 32; fz = ngz ? 32 : 80;
}

__device__ __forceinline__ void bvh_begin(BvhWalk &w, const Shapes &sh, v3 o, v3 d, float bt, int best)
{
	w.o = o; w.d = d;
	w.inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	bvh_rows(d, w.row_nx, w.row_fx, w.row_ny, w.row_fy, w.row_nz, w.row_fz);
	w.bt = bt; w.bu = 0.0f; w.bv = 0.0f;
	w.best = best; w.sp = 0;
	w.budget = 8 * sh.n_bvh_nodes + 8;
	w.next = 0; // the root
}

// w.next is a node: test its children, push the farther ones, go on in the nearest
template <bool kSlim = false>
__device__ __forceinline__ void bvh_node_step(BvhWalk &w, const Shapes &sh, const BvhStack &stk)
{
	const float kInf = __builtin_huge_valf();
	int row_nx = w.row_nx, row_fx = w.row_fx, row_ny = w.row_ny, row_fy = w.row_fy, row_nz = w.row_nz, row_fz = w.row_fz;
	if (kSlim) {
		v3 d = w.d;
		asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z)); // (made here, every time: not six values hoisted out of the walk's loop)
		bvh_rows(d, row_nx, row_fx, row_ny, row_fy, row_nz, row_fz);
	}
	// a node's rows are lo_x lo_y lo_z hi_x hi_y hi_z (four children each): the row holding the planes
	// the ray meets first on an axis is known from the sign of its direction, so the rows are
	// loaded as (near, far) per axis -- per-ray offsets, no per-child selects
	// 32-bit byte offsets from the uniform table pointer: the seven loads take the scalar base + vector offset form, no 64-bit
	// address per row -- ten vector registers less, seven waves per SIMD instead of six for the closest hits (15.6 -> 15.0 ms
	// per step, shadow rays 8.5 -> 7.8); pg_scene_set_ex keeps the table under 4 GiB
	const char *B = reinterpret_cast<const char *>(sh.bvh);
	const uint32_t nb = w.next * (uint32_t)(kBvhStride * 4);
	uint4 nx4, ny4, nz4, fx4, fy4, fz4, rf;
	// The walks are bound by the texture-address path -- seven 16-byte gathers per lane and node, 16 cycles of the unit
	// each, whether they hit the L1 or not.  The nodes most rays open (mesh.build_bvh numbers them first) are read from a
	// copy in LDS instead: closest hits 15.1 -> 11.9 ms per step of veach-ajar, shadow rays 7.9 -> 7.0 with 32 nodes.
	if (w.next < stk.n_top) {
		const LdsQuad *T = stk.top + w.next * 8u;
#define PG_Q(r) ({ const u32x4_t q_ = T[r]; make_uint4(q_.x, q_.y, q_.z, q_.w); })
		nx4 = PG_Q(row_nx >> 4); ny4 = PG_Q(row_ny >> 4); nz4 = PG_Q(row_nz >> 4);
		fx4 = PG_Q(row_fx >> 4); fy4 = PG_Q(row_fy >> 4); fz4 = PG_Q(row_fz >> 4); rf = PG_Q(6);
#undef PG_Q
	} else {
		nx4 = gather16(B + (nb + (uint32_t)row_nx)); ny4 = gather16(B + (nb + (uint32_t)row_ny)); nz4 = gather16(B + (nb + (uint32_t)row_nz));
		fx4 = gather16(B + (nb + (uint32_t)row_fx)); fy4 = gather16(B + (nb + (uint32_t)row_fy)); fz4 = gather16(B + (nb + (uint32_t)row_fz));
		rf = gather16(B + (nb + 96u));
	}
	uint32_t r0 = rf.x, r1 = rf.y, r2 = rf.z, r3 = rf.w;
	float t0, t1, t2, t3;
#define PG_F(v) __uint_as_float(v)
	// (all four tests unconditionally: a test skipped for an absent child would split the row loads)
	const bool h0 = bvh_box_hit_nf(PG_F(nx4.x), PG_F(ny4.x), PG_F(nz4.x), PG_F(fx4.x), PG_F(fy4.x), PG_F(fz4.x), w.o, w.inv, w.bt, t0);
	const bool h1 = bvh_box_hit_nf(PG_F(nx4.y), PG_F(ny4.y), PG_F(nz4.y), PG_F(fx4.y), PG_F(fy4.y), PG_F(fz4.y), w.o, w.inv, w.bt, t1);
	const bool h2 = bvh_box_hit_nf(PG_F(nx4.z), PG_F(ny4.z), PG_F(nz4.z), PG_F(fx4.z), PG_F(fy4.z), PG_F(fz4.z), w.o, w.inv, w.bt, t2);
	const bool h3 = bvh_box_hit_nf(PG_F(nx4.w), PG_F(ny4.w), PG_F(nz4.w), PG_F(fx4.w), PG_F(fy4.w), PG_F(fz4.w), w.o, w.inv, w.bt, t3);
#undef PG_F
	if (!(r0 != kBvhNone && h0)) { r0 = kBvhNone; t0 = kInf; }
	if (!(r1 != kBvhNone && h1)) { r1 = kBvhNone; t1 = kInf; }
	if (!(r2 != kBvhNone && h2)) { r2 = kBvhNone; t2 = kInf; }
	if (!(r3 != kBvhNone && h3)) { r3 = kBvhNone; t3 = kInf; }
	bvh_cswap(t0, r0, t1, r1);
	bvh_cswap(t2, r2, t3, r3);
	bvh_cswap(t0, r0, t2, r2);
	bvh_cswap(t1, r1, t3, r3);
	bvh_cswap(t1, r1, t2, r2);
	if (r3 != kBvhNone) { stk.push(w.sp, r3, t3); ++w.sp; }
	if (r2 != kBvhNone) { stk.push(w.sp, r2, t2); ++w.sp; }
	if (r1 != kBvhNone) { stk.push(w.sp, r1, t1); ++w.sp; }
	w.next = r0;
	--w.budget;
}

// w.next is a leaf: Moeller-Trumbore on its 1..8 triangles
__device__ __forceinline__ void bvh_tri_test(BvhWalk &w, v3 v0, v3 e1, v3 e2, int prim)
{
	const v3 o = w.o, d = w.d;
	const v3 p = V(d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x);
	const float det = dot3(e1, p);
	if (det == 0.0f) return;
	const float inv_det = 1.0f / det;
	const v3 s = vsub(o, v0);
	const float u = dot3(s, p) * inv_det;
	if (!(u >= 0.0f && u <= 1.0f)) return;
	const v3 q = V(s.y * e1.z - s.z * e1.y, s.z * e1.x - s.x * e1.z, s.x * e1.y - s.y * e1.x);
	const float v = dot3(d, q) * inv_det;
	if (!(v >= 0.0f && u + v <= 1.0f)) return;
	const float t = dot3(e2, q) * inv_det;
	if (t > 0.0f && t < w.bt) { w.bt = t; w.best = prim; w.bu = u; w.bv = v; }
}

__device__ __forceinline__ void bvh_leaf_step(BvhWalk &w, const Shapes &sh, int tri_base)
{
	const uint32_t first = w.next & 0x0fffffffu, count = ((w.next >> 28) & 7u) + 1u;
	for (uint32_t i = first; i < first + count; ++i) {
		const float *T = sh.tris + (size_t)i * kTriStride;
		v3 v0 = ld3(T);
		const v3 e1 = ld3(T + 3), e2 = ld3(T + 6);
		// (the triangle's nine floats are asked for TOGETHER: left alone the compiler sinks the load of v0 behind the test of the
		// determinant, which only e1 and e2 enter -- two round trips per triangle where one does; round 6, from the listing)
		asm volatile("" : "+v"(v0.x), "+v"(v0.y), "+v"(v0.z));
		bvh_tri_test(w, v0, e1, e2, tri_base + (int)i);
	}
}

// the nearest waiting child the (now shorter) ray still reaches, or kBvhNone: the walk is over
__device__ __forceinline__ void bvh_pop(BvhWalk &w, const BvhStack &stk)
{
	w.next = kBvhNone;
	while (w.sp && w.next == kBvhNone && w.budget > 0) {
		--w.sp;
		const uint2 e = stk.at(w.sp);
		if (__uint_as_float(e.y) <= w.bt * 1.0000004f) w.next = e.x;
		--w.budget;
	}
}

// stk: the walk's stack (mesh scenes only); bu, bv: barycentrics of the triangle hit (closest-hit walks).
template <int kGeneral, bool kAny = false, bool kSlim = false>
__device__ __forceinline__ int intersect(const Shapes &sh, v3 o, v3 d, float tmax, float &t_out, const BvhStack &stk,
                                         float &bu, float &bv)
{
	int best = -1;
	float bt = tmax;
	intersect_linear<kGeneral>(sh, o, d, bt, best);
	if (kGeneral >= 2 && sh.n_bvh_nodes && !(kAny && best >= 0)) {
		const int tri_base = sh.n_quads + sh.n_spheres + 6 * sh.n_boxes;
		BvhWalk w;
		bvh_begin(w, sh, o, d, bt, best);
		while (true) {
			// (a lane none of whose children the ray reaches takes its next candidate from the stack at once instead of
			// idling until the other lanes of its wave have found their leaves: closest hits 16.4 -> 15.6 ms per step,
			// shadow rays 9.5 -> 8.6; the ray's own sequence of steps is the same)
			while (!(w.next & 0x80000000u) && w.budget > 0) {
				bvh_node_step<kSlim>(w, sh, stk);
				if (w.next == kBvhNone) bvh_pop(w, stk);
			}
			if (w.next != kBvhNone && (w.next & 0x80000000u)) {
				bvh_leaf_step(w, sh, tri_base);
				if (kAny && w.best >= 0) break; // a shadow ray needs one occluder, not the nearest
			}
			bvh_pop(w, stk);
			if (w.next == kBvhNone) break;
		}
		bt = w.bt; best = w.best;
		if (w.best >= tri_base) { bu = w.bu; bv = w.bv; }
	}
	t_out = bt;
	return best;
}

// (Measured and removed, round 5: TWO closest-hit walks per lane -- a lane works on whichever of its two rays stands at a node
// and sits out a node round only when both stand at a leaf; the walks change places in the registers, one v_swap each.  By the
// counters 31 of 64 lanes are busy per vector cycle of k_wave_trace, and this fills the rounds -- but two walks are 118
// registers, four waves per SIMD instead of seven, and the kernel went from 11.5 to 15.9 ms per step (torus 3.8 -> 5.1),
// bit-exact: profiles/r05/ab_trace_two_rays_per_lane_rejected.txt.  What the walk lacks is not lanes but waves.)

// scenes without meshes (feature levels 0 and 1): no stack
template <int kGeneral, bool kAny = false>
__device__ __forceinline__ int intersect(const Shapes &sh, v3 o, v3 d, float tmax, float &t_out)
{
	static_assert(kGeneral < 2, "mesh scenes pass their BVH stack");
	const BvhStack none = bvh_stack(nullptr, nullptr, 0u);
	float bu, bv;
	return intersect<kGeneral, kAny>(sh, o, d, tmax, t_out, none, bu, bv);
}

// `bitmap` (bilinear, repeat) and `checkerboard` textures after Mitsuba 3's bitmap.cpp /
// checkerboard.cpp, the arithmetic of oracle/pg_oracle_render.c texture_eval
__device__ __forceinline__ v3 texture_eval(const Shapes &sh, int index, float u, float v)
{
	const uint32_t *T = sh.textures + (size_t)index * kTextureStride;
	const uint4 t0 = *reinterpret_cast<const uint4 *>(T), t2 = *reinterpret_cast<const uint4 *>(T + 8),
	            t3 = *reinterpret_cast<const uint4 *>(T + 12);
	const float uu = __uint_as_float(t2.z) * u + __uint_as_float(t3.x);
	const float vv = __uint_as_float(t2.w) * v + __uint_as_float(t3.y);
	if (t0.x == 2u) {
		const uint4 t1 = *reinterpret_cast<const uint4 *>(T + 4);
		const float fu = uu - __builtin_floorf(uu), fv = vv - __builtin_floorf(vv);
		const bool mx = fu > 0.5f, my = fv > 0.5f;
		return mx == my ? V(__uint_as_float(t1.x), __uint_as_float(t1.y), __uint_as_float(t1.z))
		                : V(__uint_as_float(t1.w), __uint_as_float(t2.x), __uint_as_float(t2.y));
	}
	const int W = (int)t0.y, H = (int)t0.z;
	float x = uu * (float)W - 0.5f, y = vv * (float)H - 0.5f;
	if (!(fabs_(x) < 1e9f)) x = 0.0f;
	if (!(fabs_(y) < 1e9f)) y = 0.0f;
	const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
	const float wx1 = x - fx, wy1 = y - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
	const int ix = (int)fx, iy = (int)fy;
	const int ix0 = ((ix % W) + W) % W, ix1 = (((ix + 1) % W) + W) % W;
	const int iy0 = ((iy % H) + H) % H, iy1 = (((iy + 1) % H) + H) % H;
	const uint32_t *tx = sh.texels + t0.w;
	const uint32_t t00 = tx[(size_t)iy0 * W + ix0], t10 = tx[(size_t)iy0 * W + ix1];
	const uint32_t t01 = tx[(size_t)iy1 * W + ix0], t11 = tx[(size_t)iy1 * W + ix1];
	const float *lut = sh.srgb_lut;
	float out[3];
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const float c00 = lut[(t00 >> (8 * c)) & 255u], c10 = lut[(t10 >> (8 * c)) & 255u];
		const float c01 = lut[(t01 >> (8 * c)) & 255u], c11 = lut[(t11 >> (8 * c)) & 255u];
		out[c] = (c00 * wx0 + c10 * wx1) * wy0 + (c01 * wx0 + c11 * wx1) * wy1;
	}
	return V(out[0], out[1], out[2]);
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

// bu, bv: the barycentrics the closest-hit walk found for a triangle (the oracle recomputes them by
// the intersection's own formulas: the same numbers)
template <int kGeneral>
__device__ __forceinline__ Surface surface_at(const Shapes &sh, const float *mats, int prim, v3 o, v3 d, float t,
                                              float bu = 0.0f, float bv = 0.0f)
{
	Surface s;
	const float *M;
	if (kGeneral >= 2 && prim >= sh.n_quads + sh.n_spheres + 6 * sh.n_boxes) { // a mesh triangle
		// (Measured and removed, round 5: ONE 128-byte line per triangle with everything this branch reads -- face normal +
		// material, nine vertex normals, six texture coordinates: five 16-byte gathers of one line instead of seven or eight of
		// three arrays.  k_wave_shade 28.0 -> 28.3 ms per step: the tightly packed arrays put two to five neighbouring
		// triangles into a cache line, and the lanes of a spatially sorted wave hit neighbouring triangles.
		// profiles/r05/ab_triangle_shading_line_rejected.txt)
		const size_t ti = (size_t)(prim - sh.n_quads - sh.n_spheres - 6 * sh.n_boxes);
		const float *T = sh.tris + ti * kTriStride;
		s.p = vadd(o, vscale(d, t));
		s.n = ld3(T + 9);
		s.ng = s.n;
		s.is_em = false;
		s.radiance = V(0, 0, 0);
		M = mats + (int)T[12] * kMaterialStride;
		s.m.type = (int)M[0];
		s.m.one_sided = kGeneral >= 3 && M[11] != 0.0f;
		s.m.refl = ld3(M + 1);
		s.m.M = M;
		const int tex = (int)M[12];
		const bool textured = tex > 0 && sh.tri_uvs;
		const float b0 = (1.0f - bu) - bv;
		if (sh.tri_normals) { // interpolated vertex normals
			const float *Nn = sh.tri_normals + ti * 9;
			const v3 ns = vadd(vadd(vscale(ld3(Nn), b0), vscale(ld3(Nn + 3), bu)), vscale(ld3(Nn + 6), bv));
			const float l2 = dot3(ns, ns);
			if (l2 > 0.0f) s.n = vdivs(ns, __builtin_sqrtf(l2));
		}
		if (textured) { // interpolated texture coordinates, then the texture in place of the reflectance
			const float *U = sh.tri_uvs + ti * 6;
			const float tu = (U[0] * b0 + U[2] * bu) + U[4] * bv;
			const float tv = (U[1] * b0 + U[3] * bu) + U[5] * bv;
			s.m.refl = texture_eval(sh, tex - 1, tu, tv);
		}
		return s;
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
	s.m.one_sided = kGeneral >= 3 && M[11] != 0.0f;
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
__device__ PG_OUTLINE float rc_D(v3 m, float alpha) // MicrofacetDistribution::eval
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

__device__ PG_OUTLINE float erfinv_call(float x) { return erfinv_f32(x); }

// sample_visible_11: slopes of the visible Beckmann normals for alpha = 1
__device__ PG_OUTLINE void rc_sample_visible_11(float cos_i, float u1, float u2, float &sx, float &sy)
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
__device__ PG_OUTLINE void ggx_sample_visible_11(float cos_i, float u1, float u2, float &sx, float &sy)
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

__device__ PG_OUTLINE void rd_eval_pdf(const float *M, v3 wi, v3 wo, v3 &value, float &pdf)
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

__device__ PG_OUTLINE void rd_sample(const float *M, v3 wi, float u1, float u, float v, v3 &wo, float &pdf, v3 &weight, float &eta_out)
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
	if (kGeneral >= 3 && (mt.type == 2 || mt.type == 3)) return; // smooth conductor / dielectric: delta lobes only
	if (kGeneral >= 3 && mt.type == 4) {
		rd_eval_pdf(mt.M, wi, wo, value, pdf);
		return;
	}
	if (wi.z < 0.0f && !(kGeneral >= 3 && mt.one_sided)) { wi.z = -wi.z; wo.z = -wo.z; }
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
	if (kGeneral >= 3 && mt.type == 3) { // smooth dielectric (dielectric.cpp), radiance transport
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
	if (kGeneral >= 3 && mt.type == 4) {
		rd_sample(mt.M, wi, u1, u, v, wo, pdf, weight, eta);
		return;
	}
	const bool flip = wi.z < 0.0f && !(kGeneral >= 3 && mt.one_sided);
	const float cos_i = flip ? -wi.z : wi.z;
	if (!(cos_i > 0.0f)) return;
	if (kGeneral >= 3 && mt.type == 2) { // smooth conductor (conductor.cpp): the mirror direction, weighted by Fresnel
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
		const bool occ = intersect<kGeneral, true>(sh, so, sdn, sdist * (1.0f - kShadowEps), th) >= 0;
		if (!occ) em_weight = vscale(vdivs(radiance, pdf), count);
	}
}

// sample_emitter without the visibility test, for the split pipeline: returns the shadow ray
// (origin, unit direction, length to test) and em_weight as if unoccluded; `need_shadow` says whether
// the caller has to trace it (and zero em_weight when it is occluded).  Same arithmetic otherwise.
template <int kGeneral>
__device__ __forceinline__ void sample_emitter_ray(const Shapes &sh, const DirLights &dls, const int32_t *__restrict__ emitters,
                                                   int n_em, v3 p, v3 n, float e1, float e2, v3 &ds_d, float &ds_pdf,
                                                   v3 &em_weight, bool &ds_delta, bool &need_shadow, v3 &sh_o, v3 &sh_d,
                                                   float &sh_tmax)
{
	ds_d = V(0, 0, 0);
	ds_pdf = 0.0f;
	em_weight = V(0, 0, 0);
	ds_delta = false;
	need_shadow = false;
	sh_o = V(0, 0, 0); sh_d = V(0, 0, 1); sh_tmax = 0.0f;
	if (n_em <= 0) return;
	const float count = (float)n_em, inv_count = 1.0f / count;
	uint32_t idx = (uint32_t)(e1 * count);
	if (idx > (uint32_t)(n_em - 1)) idx = (uint32_t)(n_em - 1);
	e1 = e1 * count - (float)idx;
	const int prim = emitters[idx];
	if (kGeneral >= 3 && prim < 0) { // directional.cpp sample_direction: a point two radii up the light's direction, pdf 1, delta
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
		need_shadow = true;
		sh_o = so; sh_d = vdivs(sd, sdist); sh_tmax = sdist * (1.0f - kShadowEps);
		em_weight = vscale(ld3(Dl + 3), count);
		return;
	}
	v3 pl, ln, radiance;
	float pdf_cone = 0.0f, area = 1.0f;
	const bool is_sphere = prim >= sh.n_quads;
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
		need_shadow = true;
		sh_o = so; sh_d = vdivs(sd, sdist); sh_tmax = sdist * (1.0f - kShadowEps);
		em_weight = vscale(vdivs(radiance, pdf), count);
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
	uint32_t stripe_rows, stripe_index, stripe_count; // interleaved sharding (pg_pass_params); count <= 1: the contiguous tile
	int spp, max_depth, rr_depth, guided, record, store_nee;
	float frac;
	uint32_t seed;
	int batched;               // pg_pass_params.batched: the spp samples of a pixel are spp one-sample passes seed, seed + 1, ...
	DepthCounters *dc;
	unsigned long long *ph; // probe builds (-DPG_SHADE_PHASES=1) only: the striped phase counters of k_wave_shade (pg_kernels.hpp: kPhaseStripes lines), on in instrumented passes and -- without the depth counters' atomics -- with pg_enable_depth_counters(2); never read by the product build
	int bounce, last;          // index of this launch = path depth of every live lane; last launch of the pass
	int fuse_guide;            // split pipeline: k_wave_shade_a also makes the SD-tree calls, no k_wave_guide launch
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
	// path-vertex records: a list in visiting order, planes of stride n_lanes*max_depth (pg_list_records): what
	// processPathData needs of a vertex -- bsdf weight, the two throughputs (3 planes each), the luminance of the emitter
	// sample's share, woPdf -- and, instead of its position and directions, the accumulators they lead to: r_slot
	// {path direction's, emitter direction's} and r_tree (the KD leaf's quadtree, bit 31: counted)
	uint32_t *ray_of;
	float *r_bsdf, *r_tb, *r_tr, *r_nee, *r_wp;
	uint2 *r_slot;
	uint32_t *r_tree;
	// mesh scenes (the split pipeline of pg_render_wave.hip): the state of a path between two bounces travels with
	// its place in the live list (pg_render_wave.hip, st_load): five 16-byte entries st[q * n_lanes + place] and the
	// sampler increment, read from st_in / inc_in and written -- survivors only, to their places in the next list --
	// to st_out / inc_out; the two sets swap per bounce.  L is the OUTPUT column (written once per path, where it
	// ends); ray_d, thr, prev_p, prev_pdf, prev_quad, rng_state, rng_inc, ior and the order lists are not used.
	// `ws` is the per-bounce workspace -- planes of n_lanes 32-bit words indexed by the place in the live list --
	// and bvh_ovf the overflow strips of the BVH stacks (kOvfStack entries per list position)
	const uint4 *st_in;
	uint4 *st_out;
	const uint64_t *inc_in;
	uint64_t *inc_out;
	// pg_render_sort: sort_key (k_wave_trace writes the key of every place it serves; nullptr = this bounce is not sorted)
	// and perm (the places in sorted order, read by k_wave_shade_a; nullptr = list order)
	uint16_t *sort_key;
	const uint32_t *perm;
	uint32_t n_sort; // perm covers the places [0, n_sort); a live place beyond it is served in list order
	// ... and the 128-byte records of the paths (8 entries of 16 bytes per place: the five state entries, the sampler
	// increment, the hit): carry_in = this bounce is sorted (k_wave_trace adds the hit, k_wave_shade_a reads the record
	// through perm), carry_out = the next one is (k_wave_shade_b writes the survivors' records)
	uint4 *carry_in, *carry_out;
	// the split pipeline writes a path's radiance where the path ends, by lane -- scattered -- as ONE 16-byte entry of
	// Lq; k_finish lays the output column L out from it (nullptr: the fused kernels keep L itself up to date)
	uint4 *Lq;
	uint32_t *ws;
	uint2 *bvh_ovf;
	// the ray-casting kernels are persistent: a lane whose ray is done takes the next one of the launch's
	// list.  cast_count[2 b] / [2 b + 1]: rays of bounce b's closest-hit / shadow launch handed out beyond
	// the first one per thread; shadow_count[b] and shadow_list: the shadow rays of bounce b (list
	// positions, appended by k_wave_shade_a).  All zeroed per pass.
	uint32_t *cast_count, *shadow_count, *shadow_list;
};


// The tail of a long path (max_depth 30 in scenes/torus): a launch cannot be shorter than the slowest
// single path's bounce (0.1-0.3 ms when that is two BVH walks inside a glass case), so a few
// thousand survivors would cost that floor once per bounce.  At fixed checkpoints the host also
// launches k_bounce_tail: when no more than kTailPaths paths are alive it takes all of them over
// and every lane follows its own path to its end in this one launch; the per-bounce launches after
// it find that out from the same counts and retire.
constexpr uint32_t kTailPaths = 128u * 1024u; // torus (tools/exp_tail.sh): 32 Ki -> 12.9, 128 Ki -> 12.4, 512 Ki -> 13.2, 2 Mi -> 13.2 ms per pass
__host__ __device__ constexpr bool tail_checkpoint(int bounce, int max_depth)
{
	return max_depth > 8 && bounce >= 4 && bounce + 1 < max_depth &&
	       (bounce < 8 || (bounce < 16 && bounce % 2 == 0) || bounce % 4 == 0);
}
// The counters of EARLIER bounces are final when a launch of bounce b reads them (the launches of a pass are stream-ordered) and the
// same for every lane: they are read as CONSTANTS -- address space 4, scalar loads (s_load, counted by lgkmcnt) -- whatever stands
// around the read.  Left to the compiler this was a scalar load only by luck: with ANY instruction it cannot see through behind the
// kernel's entry (one `s_nop`, a probe's stamp) the read of live_count[] became a vector load followed by `s_waitcnt vmcnt(0)` --
// a wait for EVERY vector load in flight, among them the records' gathers and the staging loads k_wave_shade issues first so
// that they overlap -- and k_wave_shade took 46 ms per step instead of 27.5 at identical registers, LDS and occupancy
// (profiles/r06/phase_probe.txt: this, not the stamps, was round 5's "a never-taken branch costs 60 %").
typedef const __attribute__((address_space(4))) uint32_t *ConstU32Ptr;
__device__ __forceinline__ uint32_t live_final(const RenderArgs &a, int j)
{
	return reinterpret_cast<ConstU32Ptr>(reinterpret_cast<uintptr_t>(a.live_count))[j];
}
// entries of the record list ahead of bounce a.bounce's: the camera rays' and the earlier bounces' survivors
__device__ __forceinline__ uint64_t records_before(const RenderArgs &a)
{
	uint64_t r = a.n_lanes;
	for (int j = 0; j + 1 < a.bounce; ++j) r += live_final(a, j);
	return r;
}

// Did a tail launch at a checkpoint <= bounce take the paths over?  live_count[c-1] is final when
// checkpoint c is launched, and the first checkpoint that fires decides: the entries a tail launch
// adds to afterwards are never looked at before one that already said yes.
__device__ __forceinline__ bool tail_took_over(const RenderArgs &a, int bounce)
{
	if (a.max_depth <= 8) return false;
	for (int c = 4; c <= bounce; ++c)
		if (tail_checkpoint(c, a.max_depth) && live_final(a, c - 1) <= kTailPaths) return true;
	return false;
}


// The sampler stream of sample s of film pixel `pixel` (Mitsuba's `independent` sampler: one PCG32 stream per lane of the
// wavefront, lane = pixel * spp + s, path_guiding_integrator.py:414-417; keyed by the GLOBAL pixel, so a tile renders
// exactly the samples the full-frame pass would).  A batched call stands for spp consecutive ONE-sample passes with the
// seeds seed, seed + 1, ... (main.py:192, 218: a training pass is one sample per pixel, seeded initial_seed + cumm_spp):
// its sample s IS the sample pass seed + s gives the pixel -- lane `pixel` of that pass's wavefront.
__device__ __forceinline__ Pcg32 lane_stream(uint32_t seed, int spp, int batched, uint64_t pixel, uint32_t s)
{
	return batched ? pcg32_seed(seed + s, (uint32_t)pixel) : pcg32_seed(seed, (uint32_t)(pixel * (uint64_t)spp + s));
}

// film pixel of the tile-local pixel i (pg_pass_params: a contiguous range, or bands of rows dealt round-robin)
__device__ __forceinline__ uint64_t global_pixel(const RenderArgs &a, uint64_t i)
{
	if (a.stripe_count <= 1u) return a.pixel_begin + i;
	const uint64_t W = (uint64_t)a.cam.width, row = i / W, col = i % W;
	const uint64_t grow = (row / a.stripe_rows) * ((uint64_t)a.stripe_rows * a.stripe_count) +
	                      (uint64_t)a.stripe_index * a.stripe_rows + row % a.stripe_rows;
	return grow * W + col;
}

// pg_render_wave.hip: one stage (0 trace, 1 shade_a, 2 shadow, 3 guide, 4 shade_b, 5 tail) of one bounce of
// the split pipeline, feature level 2 or 3; the number of 32-bit planes of its workspace
// n_cus: compute units of the device (the two ray-casting stages are persistent kernels sized by it)
void launch_wave_stage(int stage, int level, bool first, const RenderArgs &a, unsigned grid_blocks, unsigned n_cus, hipStream_t s);
constexpr int kCastBlocksPerCU = 8; // upper bound of the resident 256-thread workgroups of a ray-casting kernel per compute unit
int wave_workspace_planes();

} // namespace pg
