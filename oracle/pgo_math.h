/*
 * pgo_math.h -- TEST INFRASTRUCTURE (CPU oracle). Not part of the product.
 *
 * Deterministic scalar arithmetic used by the CPU restatement of the reference's
 * SD-tree path.  Everything here is specified in DESIGN.md section 4 ("Arithmetic
 * contract"); the HIP product implements the same contract independently in
 * practical_path_guiding_lab_amd/csrc/pg_math.hpp.
 *
 * Reference sites restated:
 *   src/common.py:100-129  canonicalToDir
 *   src/common.py:132-158  dirToCanonical
 *   Mitsuba 3 `independent` sampler (PCG32 + TEA seeding; third-party, version
 *   unpinned -- see DESIGN.md, "parity unpinned" note)
 *
 * The reference evaluates sincos/atan2 with Dr.Jit's fp32 CUDA intrinsics whose
 * last-bit behaviour cannot be reproduced here.  The contract instead fixes a
 * transparent double-precision Taylor evaluation rounded once to fp32, so that CPU
 * and GPU agree bit-for-bit.  Only +,-,*,/ and rint on IEEE doubles are used and the
 * translation unit is compiled with -ffp-contract=off.
 */
#ifndef PGO_MATH_H
#define PGO_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define PGO_TWO_PI_F 6.28318530717958647692f /* fl32(2*pi)  = 0x40C90FDB */
#define PGO_INV_FOUR_PI_F 0.07957747154594766788f /* fl32(1/(4*pi)) */

static inline uint32_t pgo_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float pgo_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---- sin/cos of an fp32 angle, evaluated in double, rounded once to fp32 ---- */
static inline void pgo_sincos(float phi, float *s_out, float *c_out)
{
	const double two_over_pi = 0.63661977236758134308;
	/* pi/2 split: HI has 33 significant bits so k*HI is exact for |k| < 2^20 */
	const double pio2_hi = 1.57079632673412561417;
	const double pio2_lo = 6.07710050650619224932e-11;
	double x = (double)phi;
	double k = rint(x * two_over_pi);
	double r = (x - k * pio2_hi) - k * pio2_lo;
	double z = r * r;
	/* Taylor series, |r| <= pi/4 (+ tiny): remainder < 1e-19 */
	double ps = 1.0 / 355687428096000.0;            /* 1/17! */
	ps = -1.0 / 1307674368000.0 + z * ps;         /* 1/15! */
	ps = 1.0 / 6227020800.0 + z * ps;             /* 1/13! */
	ps = -1.0 / 39916800.0 + z * ps;              /* 1/11! */
	ps = 1.0 / 362880.0 + z * ps;                 /* 1/9!  */
	ps = -1.0 / 5040.0 + z * ps;                  /* 1/7!  */
	ps = 1.0 / 120.0 + z * ps;                    /* 1/5!  */
	ps = -1.0 / 6.0 + z * ps;                     /* 1/3!  */
	double sr = r + r * (z * ps);
	double pc = -1.0 / 6402373705728000.0;         /* 1/18! */
	pc = 1.0 / 20922789888000.0 + z * pc;         /* 1/16! */
	pc = -1.0 / 87178291200.0 + z * pc;           /* 1/14! */
	pc = 1.0 / 479001600.0 + z * pc;              /* 1/12! */
	pc = -1.0 / 3628800.0 + z * pc;               /* 1/10! */
	pc = 1.0 / 40320.0 + z * pc;                  /* 1/8!  */
	pc = -1.0 / 720.0 + z * pc;                   /* 1/6!  */
	pc = 1.0 / 24.0 + z * pc;                     /* 1/4!  */
	pc = -0.5 + z * pc;                           /* 1/2!  */
	double cr = 1.0 + z * pc;
	long kk = (long)k;
	double s, c;
	switch (kk & 3) {
	case 0: s = sr; c = cr; break;
	case 1: s = cr; c = -sr; break;
	case 2: s = -sr; c = -cr; break;
	default: s = -cr; c = sr; break;
	}
	*s_out = (float)s;
	*c_out = (float)c;
}

/* ---- atan2 of fp32 operands, evaluated in double, rounded once to fp32 ---- */
static inline double pgo_atan_unit(double t) /* t in [0,1] */
{
	const double pi_4 = 0.78539816339744830962;
	double u = t, base = 0.0;
	if (t > 0.41421356237309504880) { u = (t - 1.0) / (t + 1.0); base = pi_4; }
	double z = u * u;
	/* sum_{n=0}^{19} (-1)^n z^n/(2n+1), Horner from the top; z <= 0.1716 */
	double p = -1.0 / 39.0;
	p = 1.0 / 37.0 + z * p;
	p = -1.0 / 35.0 + z * p;
	p = 1.0 / 33.0 + z * p;
	p = -1.0 / 31.0 + z * p;
	p = 1.0 / 29.0 + z * p;
	p = -1.0 / 27.0 + z * p;
	p = 1.0 / 25.0 + z * p;
	p = -1.0 / 23.0 + z * p;
	p = 1.0 / 21.0 + z * p;
	p = -1.0 / 19.0 + z * p;
	p = 1.0 / 17.0 + z * p;
	p = -1.0 / 15.0 + z * p;
	p = 1.0 / 13.0 + z * p;
	p = -1.0 / 11.0 + z * p;
	p = 1.0 / 9.0 + z * p;
	p = -1.0 / 7.0 + z * p;
	p = 1.0 / 5.0 + z * p;
	p = -1.0 / 3.0 + z * p;
	p = 1.0 + z * p;
	return base + u * p;
}

static inline float pgo_atan2(float yf, float xf)
{
	const double pi = 3.14159265358979323846;
	const double pi_2 = 1.57079632679489661923;
	if (yf != yf || xf != xf) return yf + xf; /* NaN */
	double x = (double)xf, y = (double)yf;
	double ax = fabs(x), ay = fabs(y);
	double hi = ax > ay ? ax : ay, lo = ax > ay ? ay : ax;
	double t;
	if (hi == 0.0) t = 0.0;
	else if (hi == (double)INFINITY) t = (lo == (double)INFINITY) ? 1.0 : 0.0;
	else t = lo / hi;
	double a = pgo_atan_unit(t);
	if (ay > ax) a = pi_2 - a;
	if (signbit(xf)) a = pi - a;
	if (signbit(yf)) a = -a;
	return (float)a;
}

/* src/common.py:100-129 */
static inline void pgo_canonical_to_dir(float px, float py, float d[3])
{
	float cosTheta = 2.0f * py - 1.0f;
	float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
	float phi = PGO_TWO_PI_F * px;
	float sinPhi, cosPhi;
	pgo_sincos(phi, &sinPhi, &cosPhi);
	d[0] = sinTheta * cosPhi;
	d[1] = sinTheta * sinPhi;
	d[2] = cosTheta;
}

/* src/common.py:132-158 */
static inline void pgo_dir_to_canonical(float dx, float dy, float dz, float p[2])
{
	float cosTheta = dz < -1.0f ? -1.0f : (dz > 1.0f ? 1.0f : dz);
	float phi = pgo_atan2(dy, dx);
	while (phi < 0.0f) phi += PGO_TWO_PI_F; /* common.py:148-150, 2.0*pi as fp32 */
	p[0] = phi / PGO_TWO_PI_F;
	p[1] = (cosTheta + 1.0f) / 2.0f;
	if (!(isfinite(dx) && isfinite(dy) && isfinite(dz))) { p[0] = 0.0f; p[1] = 0.0f; }
}

/* Rec.709 luminance as mi.luminance(Color3f) (third-party; weights assumed) */
static inline float pgo_luminance(float r, float g, float b)
{
	return r * 0.212671f + g * 0.715160f + b * 0.072169f;
}

/* ---- exp / log / erf / erfinv of fp32 arguments for the rough-conductor BSDF ----
 * Same contract as sincos/atan2 above: a fixed sequence of IEEE double operations (no fma, no
 * libm), rounded once to fp32, written out identically in csrc/pg_math.hpp.  exp and log are
 * accurate to the last fp32 bit or so; erf (Abramowitz & Stegun 7.1.26, |err| < 1.5e-7) and erfinv
 * (M. Giles' single-precision polynomial, rel. err < 1.3e-7) are as good as the approximations
 * Dr.Jit ships for the same functions. */
static inline double pgo_u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t pgo_d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

static inline double pgo_exp_d(double x) /* finite x in [-750, 700] */
{
	const double inv_ln2 = 1.44269504088896338700;
	const double ln2_hi = 6.93147180369123816490e-01; /* low 20 bits zero: k*ln2_hi is exact */
	const double ln2_lo = 1.90821492927058770002e-10;
	double k = rint(x * inv_ln2);
	double r = (x - k * ln2_hi) - k * ln2_lo; /* |r| <= 0.3466 */
	double p = 1.0 / 6227020800.0;  /* 1/13! */
	p = 1.0 / 479001600.0 + r * p;
	p = 1.0 / 39916800.0 + r * p;
	p = 1.0 / 3628800.0 + r * p;
	p = 1.0 / 362880.0 + r * p;
	p = 1.0 / 40320.0 + r * p;
	p = 1.0 / 5040.0 + r * p;
	p = 1.0 / 720.0 + r * p;
	p = 1.0 / 120.0 + r * p;
	p = 1.0 / 24.0 + r * p;
	p = 1.0 / 6.0 + r * p;
	p = 0.5 + r * p;
	p = 1.0 + r * p;
	p = 1.0 + r * p;
	long kk = (long)k;
	return p * pgo_u2d((uint64_t)(kk + 1023) << 52);
}

static inline float pgo_exp(float xf)
{
	if (xf != xf) return xf;
	if (xf > 88.8f) return INFINITY;
	if (xf < -87.4f) return 0.0f; /* results below the smallest normal fp32 are flushed: no denormal dependence */
	float r = (float)pgo_exp_d((double)xf);
	return r < 1.17549435e-38f ? 0.0f : r;
}

static inline double pgo_log_d(double x) /* finite x > 0, normal double */
{
	const double ln2_hi = 6.93147180369123816490e-01;
	const double ln2_lo = 1.90821492927058770002e-10;
	uint64_t u = pgo_d2u(x);
	long e = (long)((u >> 52) & 0x7ff) - 1023;
	double m = pgo_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1, 2) */
	if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
	double s = (m - 1.0) / (m + 1.0); /* |s| <= 0.1716 */
	double z = s * s;
	double p = 1.0 / 27.0;
	p = 1.0 / 25.0 + z * p;
	p = 1.0 / 23.0 + z * p;
	p = 1.0 / 21.0 + z * p;
	p = 1.0 / 19.0 + z * p;
	p = 1.0 / 17.0 + z * p;
	p = 1.0 / 15.0 + z * p;
	p = 1.0 / 13.0 + z * p;
	p = 1.0 / 11.0 + z * p;
	p = 1.0 / 9.0 + z * p;
	p = 1.0 / 7.0 + z * p;
	p = 1.0 / 5.0 + z * p;
	p = 1.0 / 3.0 + z * p;
	p = 1.0 + z * p;
	double ed = (double)e;
	return ed * ln2_hi + (ed * ln2_lo + (2.0 * s) * p);
}

static inline float pgo_log(float xf)
{
	if (xf != xf || xf < 0.0f) return NAN;
	if (xf == 0.0f) return -INFINITY;
	if (xf == INFINITY) return INFINITY;
	return (float)pgo_log_d((double)xf); /* fp32 denormals are normal doubles */
}

static inline float pgo_erf(float xf)
{
	if (xf != xf) return xf;
	double a = fabs((double)xf);
	double r;
	if (a >= 4.0) r = 1.0;
	else {
		double t = 1.0 / (1.0 + 0.3275911 * a);
		double poly = 1.061405429;
		poly = -1.453152027 + t * poly;
		poly = 1.421413741 + t * poly;
		poly = -0.284496736 + t * poly;
		poly = 0.254829592 + t * poly;
		poly = t * poly;
		r = 1.0 - poly * pgo_exp_d(-(a * a));
	}
	return (float)(signbit(xf) ? -r : r);
}

static inline float pgo_erfinv(float xf)
{
	if (xf != xf) return xf;
	double x = (double)xf;
	double q = (1.0 - x) * (1.0 + x);
	if (!(q > 0.0)) return q == 0.0 ? (signbit(xf) ? -INFINITY : INFINITY) : NAN;
	double w = -pgo_log_d(q);
	double p;
	if (w < 5.0) {
		w = w - 2.5;
		p = 2.81022636e-08;
		p = 3.43273939e-07 + p * w;
		p = -3.5233877e-06 + p * w;
		p = -4.39150654e-06 + p * w;
		p = 0.00021858087 + p * w;
		p = -0.00125372503 + p * w;
		p = -0.00417768164 + p * w;
		p = 0.246640727 + p * w;
		p = 1.50140941 + p * w;
	} else {
		w = sqrt(w) - 3.0;
		p = -0.000200214257;
		p = 0.000100950558 + p * w;
		p = 0.00134934322 + p * w;
		p = -0.00367342844 + p * w;
		p = 0.00573950773 + p * w;
		p = -0.0076224613 + p * w;
		p = 0.00943887047 + p * w;
		p = 1.00167406 + p * w;
		p = 2.83297682 + p * w;
	}
	return (float)(p * x);
}

/* ---- PCG32 (O'Neill) + TEA seeding, as Mitsuba's `independent` sampler uses them ---- */
typedef struct { uint64_t state, inc; } pgo_pcg32;

static inline uint32_t pgo_pcg32_next_u32(pgo_pcg32 *r)
{
	uint64_t old = r->state;
	r->state = old * 0x5851f42d4c957f2dULL + r->inc;
	uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

static inline float pgo_pcg32_next_f32(pgo_pcg32 *r)
{
	return pgo_u2f((pgo_pcg32_next_u32(r) >> 9) | 0x3f800000u) - 1.0f;
}

static inline uint64_t pgo_tea64(uint32_t v0, uint32_t v1)
{
	uint32_t sum = 0;
	for (int i = 0; i < 4; ++i) {
		sum += 0x9e3779b9u;
		v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
		v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
	}
	return ((uint64_t)v0 << 32) | v1;
}

static inline void pgo_pcg32_seed(pgo_pcg32 *r, uint32_t seed, uint32_t lane)
{
	uint64_t initstate = pgo_tea64(seed, lane);
	uint64_t initseq = pgo_tea64(lane, seed);
	r->state = 0;
	r->inc = (initseq << 1) | 1u;
	pgo_pcg32_next_u32(r);
	r->state += initstate;
	pgo_pcg32_next_u32(r);
}

#endif /* PGO_MATH_H */
