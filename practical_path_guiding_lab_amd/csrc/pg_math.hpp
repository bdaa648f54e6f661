// pg_math.hpp -- device arithmetic of the SD-tree library.
//
// fp32 everywhere the reference computes in fp32 (Dr.Jit Float), with every operation
// individually rounded (the build passes -ffp-contract=off); the two transcendental pairs
// the path needs (sincos of 2*pi*x in common.py:113-115, atan2 in common.py:142) are evaluated
// as plain double-precision Taylor polynomials and rounded once, so results do not depend on
// a vendor math library.  Radiance weights are accumulated as exact fixed-point integers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

constexpr float kTwoPiF = 6.28318530717958647692f;
constexpr float kInvFourPiF = 0.07957747154594766788f;
constexpr int kFracBits = 40;      // PG_FRAC_BITS
constexpr int kClampLog2 = 48;     // PG_W_CLAMP_LOG2

// ---------------------------------------------------------------------------------------
// sin/cos of an fp32 angle: Cody-Waite reduction by pi/2 in double, Taylor to r^17 / r^18
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_f32(float phi, float &s_out, float &c_out)
{
	const double x = (double)phi;
	const double k = __builtin_rint(x * 0.63661977236758134308);
	// pi/2 = hi + lo, hi has 33 significant bits: k*hi is exact for the k that occur
	const double r = (x - k * 1.57079632673412561417) - k * 6.07710050650619224932e-11;
	const double z = r * r;
	double ps = 1.0 / 355687428096000.0;
	ps = -1.0 / 1307674368000.0 + z * ps;
	ps = 1.0 / 6227020800.0 + z * ps;
	ps = -1.0 / 39916800.0 + z * ps;
	ps = 1.0 / 362880.0 + z * ps;
	ps = -1.0 / 5040.0 + z * ps;
	ps = 1.0 / 120.0 + z * ps;
	ps = -1.0 / 6.0 + z * ps;
	const double sr = r + r * (z * ps);
	double pc = -1.0 / 6402373705728000.0;
	pc = 1.0 / 20922789888000.0 + z * pc;
	pc = -1.0 / 87178291200.0 + z * pc;
	pc = 1.0 / 479001600.0 + z * pc;
	pc = -1.0 / 3628800.0 + z * pc;
	pc = 1.0 / 40320.0 + z * pc;
	pc = -1.0 / 720.0 + z * pc;
	pc = 1.0 / 24.0 + z * pc;
	pc = -0.5 + z * pc;
	const double cr = 1.0 + z * pc;
	const int q = (int)(long long)k & 3;
	const double s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
	const double c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
	s_out = (float)s;
	c_out = (float)c;
}

// atan on [0,1]: reduce to |u| <= sqrt(2)-1 by atan t = pi/4 + atan((t-1)/(t+1)), 20-term series
__device__ __forceinline__ double atan_unit(double t)
{
	const bool big = t > 0.41421356237309504880;
	const double u = big ? (t - 1.0) / (t + 1.0) : t;
	const double base = big ? 0.78539816339744830962 : 0.0;
	const double z = u * u;
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

__device__ __forceinline__ float atan2_f32(float yf, float xf)
{
	if (yf != yf || xf != xf) return yf + xf;
	const double x = (double)xf, y = (double)yf;
	const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
	const double hi = ax > ay ? ax : ay, lo = ax > ay ? ay : ax;
	const double inf = __builtin_huge_val();
	double t;
	if (hi == 0.0) t = 0.0;
	else if (hi == inf) t = (lo == inf) ? 1.0 : 0.0;
	else t = lo / hi;
	double a = atan_unit(t);
	if (ay > ax) a = 1.57079632679489661923 - a;
	if (__float_as_uint(xf) >> 31) a = 3.14159265358979323846 - a;
	if (__float_as_uint(yf) >> 31) a = -a;
	return (float)a;
}

__device__ __forceinline__ bool finite_f32(float v)
{
	return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u;
}

// ---------------------------------------------------------------------------------------
// exp / log / erf / erfinv of fp32 arguments for the rough-conductor BSDF (pg_render.hip): the
// same fixed sequences of double operations as oracle/pgo_math.h, rounded once to fp32.
// erf: Abramowitz & Stegun 7.1.26; erfinv: M. Giles' single-precision polynomial.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double exp_d(double x) // finite x in [-750, 700]
{
	const double k = __builtin_rint(x * 1.44269504088896338700);
	const double r = (x - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
	double p = 1.0 / 6227020800.0;
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
	const long long kk = (long long)k;
	return p * __longlong_as_double((kk + 1023) << 52);
}

__device__ __forceinline__ float exp_f32(float xf)
{
	if (xf != xf) return xf;
	if (xf > 88.8f) return __builtin_huge_valf();
	if (xf < -87.4f) return 0.0f;
	const float r = (float)exp_d((double)xf);
	return r < 1.17549435e-38f ? 0.0f : r;
}

__device__ __forceinline__ double log_d(double x) // finite normal x > 0
{
	const unsigned long long u = (unsigned long long)__double_as_longlong(x);
	long long e = (long long)((u >> 52) & 0x7ffull) - 1023;
	double m = __longlong_as_double((long long)((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
	if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
	const double s = (m - 1.0) / (m + 1.0);
	const double z = s * s;
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
	const double ed = (double)e;
	return ed * 6.93147180369123816490e-01 + (ed * 1.90821492927058770002e-10 + (2.0 * s) * p);
}

__device__ __forceinline__ float log_f32(float xf)
{
	if (xf != xf || xf < 0.0f) return __builtin_nanf("");
	if (xf == 0.0f) return -__builtin_huge_valf();
	if (xf == __builtin_huge_valf()) return xf;
	return (float)log_d((double)xf);
}

__device__ __forceinline__ float erf_f32(float xf)
{
	if (xf != xf) return xf;
	const double a = __builtin_fabs((double)xf);
	double r;
	if (a >= 4.0) r = 1.0;
	else {
		const double t = 1.0 / (1.0 + 0.3275911 * a);
		double poly = 1.061405429;
		poly = -1.453152027 + t * poly;
		poly = 1.421413741 + t * poly;
		poly = -0.284496736 + t * poly;
		poly = 0.254829592 + t * poly;
		poly = t * poly;
		r = 1.0 - poly * exp_d(-(a * a));
	}
	return (float)((__float_as_uint(xf) >> 31) ? -r : r);
}

__device__ __forceinline__ float erfinv_f32(float xf)
{
	if (xf != xf) return xf;
	const double x = (double)xf;
	const double q = (1.0 - x) * (1.0 + x);
	if (!(q > 0.0)) {
		if (q == 0.0) return (__float_as_uint(xf) >> 31) ? -__builtin_huge_valf() : __builtin_huge_valf();
		return __builtin_nanf("");
	}
	double w = -log_d(q);
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
		w = __builtin_sqrt(w) - 3.0;
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

// common.py:100-129
__device__ __forceinline__ void canonical_to_dir(float px, float py, float &dx, float &dy, float &dz)
{
	const float cosTheta = 2.0f * py - 1.0f;
	const float sinTheta = __builtin_sqrtf(1.0f - cosTheta * cosTheta);
	const float phi = kTwoPiF * px;
	float sinPhi, cosPhi;
	sincos_f32(phi, sinPhi, cosPhi);
	dx = sinTheta * cosPhi;
	dy = sinTheta * sinPhi;
	dz = cosTheta;
}

// common.py:132-158
__device__ __forceinline__ void dir_to_canonical(float dx, float dy, float dz, float &px, float &py)
{
	const float cosTheta = dz < -1.0f ? -1.0f : (dz > 1.0f ? 1.0f : dz);
	float phi = atan2_f32(dy, dx);
	// common.py:148-150; atan2 >= -pi so one add suffices, the bounded loop keeps it literal
	for (int i = 0; i < 4 && phi < 0.0f; ++i) phi += kTwoPiF;
	px = phi / kTwoPiF;
	py = (cosTheta + 1.0f) / 2.0f;
	if (!(finite_f32(dx) && finite_f32(dy) && finite_f32(dz))) { px = 0.0f; py = 0.0f; }
}

// mi.luminance(Color3f), Rec.709 weights (third-party; SURVEY 8c assumption)
__device__ __forceinline__ float luminance(float r, float g, float b)
{
	return r * 0.212671f + g * 0.715160f + b * 0.072169f;
}

// ---------------------------------------------------------------------------------------
// PCG32 stream as used by Mitsuba's `independent` sampler
// ---------------------------------------------------------------------------------------
struct Pcg32 {
	uint64_t state, inc;
	__device__ __forceinline__ uint32_t next_u32()
	{
		const uint64_t old = state;
		state = old * 0x5851f42d4c957f2dULL + inc;
		const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		const uint32_t rot = (uint32_t)(old >> 59u);
		return (xs >> rot) | (xs << ((0u - rot) & 31u));
	}
	__device__ __forceinline__ float next_f32()
	{
		return __uint_as_float((next_u32() >> 9) | 0x3f800000u) - 1.0f;
	}
	// a draw whose value is not needed: only the LCG step, no output permutation
	__device__ __forceinline__ void skip() { state = state * 0x5851f42d4c957f2dULL + inc; }
};

__device__ __forceinline__ uint64_t tea64(uint32_t v0, uint32_t v1)
{
	uint32_t sum = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		sum += 0x9e3779b9u;
		v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
		v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
	}
	return ((uint64_t)v0 << 32) | v1;
}

__device__ __forceinline__ Pcg32 pcg32_seed(uint32_t seed, uint32_t lane)
{
	Pcg32 r;
	r.state = 0;
	r.inc = (tea64(lane, seed) << 1) | 1u;
	r.next_u32();
	r.state += tea64(seed, lane);
	r.next_u32();
	return r;
}

// ---------------------------------------------------------------------------------------
// Fixed-point weight quantisation into three 32-bit-payload limbs
// ---------------------------------------------------------------------------------------
struct Limbs {
	int64_t l0, l1, l2;
	__device__ __forceinline__ bool zero() const { return (l0 | l1 | l2) == 0; }
};

// trunc(clamp(w, +-2^48) * 2^40): 24-bit significand shifted into a 96-bit magnitude
__device__ __forceinline__ Limbs quantize_weight(float w)
{
	const uint32_t u = __float_as_uint(w);
	const bool neg = (u >> 31) != 0;
	int e = (int)((u >> 23) & 0xffu);
	uint32_t m = u & 0x7fffffu;
	Limbs out = {0, 0, 0};
	if (e == 255 && m != 0) return out; // NaN adds nothing
	int exp2;
	if (e == 0) exp2 = -149;
	else { m |= 0x800000u; exp2 = e - 150; }
	if (e == 255 || e - 127 >= kClampLog2) { m = 0x800000u; exp2 = kClampLog2 - 23; }
	const int shift = exp2 + kFracBits; // in [-109, 65]
	uint64_t lo = 0, hi = 0;            // 128-bit magnitude
	if (shift >= 64) { hi = (uint64_t)m << (shift - 64); }
	else if (shift > 0) { lo = (uint64_t)m << shift; hi = (shift > 40) ? ((uint64_t)m >> (64 - shift)) : 0; }
	else if (shift > -32) { lo = (uint64_t)(m >> (-shift)); }
	int64_t a = (int64_t)(lo & 0xffffffffu), b = (int64_t)(lo >> 32), c = (int64_t)hi;
	out.l0 = neg ? -a : a;
	out.l1 = neg ? -b : b;
	out.l2 = neg ? -c : c;
	return out;
}

// 128-bit two's complement value of three limbs: l0 + l1*2^32 + l2*2^64
struct I128 {
	uint64_t lo;
	int64_t hi;
};

__device__ __host__ __forceinline__ I128 limbs_resolve(int64_t l0, int64_t l1, int64_t l2)
{
	// start with l0 sign-extended, add l1 << 32 and l2 << 64 with carries
	uint64_t lo = (uint64_t)l0;
	int64_t hi = l0 < 0 ? -1 : 0;
	const uint64_t add_lo = (uint64_t)l1 << 32;
	const int64_t add_hi = l1 >> 32; // arithmetic shift keeps the sign
	const uint64_t nlo = lo + add_lo;
	hi += add_hi + (nlo < lo ? 1 : 0);
	lo = nlo;
	hi += l2;
	I128 r = {lo, hi};
	return r;
}

__device__ __host__ __forceinline__ I128 i128_add(I128 a, I128 b)
{
	I128 r;
	r.lo = a.lo + b.lo;
	r.hi = a.hi + b.hi + (r.lo < a.lo ? 1 : 0);
	return r;
}

// ---- the exchange format of an accumulator: 24 bytes instead of 32 ------------------------------------
// On the device an accumulator is four int64 words -- three limbs of 32 payload bits (cheap to add to with 64-bit
// atomics) and a record count.  What travels between GPUs is its VALUE: T = l0 + l1 2^32 + l2 2^64 (|T| < 2^119: at
// most 2^31 records of |q| <= 2^88 per accumulator and iteration over ALL ranks -- the bound the limbs already rely on)
// cut into two unsigned pieces of 52 bits and a signed top, the count folded into the top word's upper bits:
//     p0 = T mod 2^52          p1 = (T >> 52) mod 2^52          p2 = (count << 24) + (T >> 104)
// Element-wise int64 sums of these words over R ranks cannot overflow (R 2^52 < 2^63 for R < 2^11; sum of counts < 2^31, so
// p2 < 2^55 + |sum of tops|, the tops within +-(2^15 + R) of each other's sum: 24 bits with room) and decode to the exact
// sums of T and of the counts, whatever R and whatever the order: unpack gives the limbs of the sum, same value,
// same count -- the refine that follows cannot tell how the sum travelled.
constexpr int kXchgWords = 3;
constexpr int kXchgPieceBits = 52;
constexpr int kXchgCountShift = 24;

__device__ __host__ __forceinline__ void xchg_pack(const long long a[4], long long out[3])
{
	const I128 T = limbs_resolve(a[0], a[1], a[2]);
	const uint64_t mask = (1ull << kXchgPieceBits) - 1ull;
	out[0] = (long long)(T.lo & mask);
	out[1] = (long long)(((T.lo >> kXchgPieceBits) | ((uint64_t)T.hi << (64 - kXchgPieceBits))) & mask);
	const long long top = (long long)(T.hi >> (2 * kXchgPieceBits - 64)); // arithmetic: T >> 104, signed
	out[2] = (long long)((unsigned long long)a[3] << kXchgCountShift) + top;
}

__device__ __host__ __forceinline__ void xchg_unpack(const long long p[3], long long a[4])
{
	// the top: the low 24 bits of p2 as a signed number; what is left above them is the count
	const long long top = (long long)((unsigned long long)p[2] << (64 - kXchgCountShift)) >> (64 - kXchgCountShift);
	const long long cnt = (p[2] - top) >> kXchgCountShift;
	I128 T = {(uint64_t)p[0], 0};
	const I128 mid = {(uint64_t)p[1] << kXchgPieceBits, (int64_t)((uint64_t)p[1] >> (64 - kXchgPieceBits))};
	T = i128_add(T, mid);
	T.hi += top * (1ll << (2 * kXchgPieceBits - 64));
	a[0] = (long long)(T.lo & 0xffffffffull);
	a[1] = (long long)(T.lo >> 32);
	a[2] = (long long)T.hi;
	a[3] = cnt;
}

// exact integer -> fp32 with one round-to-nearest-even, then the exact 2^-kFracBits scaling
__device__ __host__ __forceinline__ float i128_to_f32(I128 v)
{
	const bool neg = v.hi < 0;
	uint64_t lo = v.lo, hi = (uint64_t)v.hi;
	if (neg) { lo = ~lo + 1; hi = ~hi + (lo == 0 ? 1 : 0); }
	if ((lo | hi) == 0) return 0.0f;
	int msb;
	if (hi) msb = 127 - __builtin_clzll(hi);
	else msb = 63 - __builtin_clzll(lo);
	uint32_t mant;
	int sh = 0;
	if (msb <= 23) mant = (uint32_t)lo;
	else {
		sh = msb - 23;
		// mant = v >> sh ; rem = v & (2^sh - 1) compared with half = 2^(sh-1)
		uint64_t top, rem_hi, rem_lo;
		if (sh >= 64) {
			top = hi >> (sh - 64);
			rem_hi = (sh == 64) ? 0 : (hi & ((1ull << (sh - 64)) - 1));
			rem_lo = lo;
		} else {
			top = (lo >> sh) | (hi << (64 - sh)); // sh in [1,63]
			rem_hi = 0;
			rem_lo = lo & ((1ull << sh) - 1);
		}
		mant = (uint32_t)top;
		// half = 2^(sh-1)
		const uint64_t half_hi = (sh - 1 >= 64) ? (1ull << (sh - 65)) : 0;
		const uint64_t half_lo = (sh - 1 >= 64) ? 0 : (1ull << (sh - 1));
		const bool gt = (rem_hi > half_hi) || (rem_hi == half_hi && rem_lo > half_lo);
		const bool eq = (rem_hi == half_hi) && (rem_lo == half_lo);
		if (gt || (eq && (mant & 1u))) ++mant;
		if (mant == 0x1000000u) { mant >>= 1; ++sh; }
	}
	// mant < 2^24 is exact in fp32; scale by 2^(sh - kFracBits) via the exponent field
	const int e2 = sh - kFracBits; // in [-40, 104-40]
	float f = (float)mant;
	union { uint32_t u; float f; } sc;
	sc.u = (uint32_t)(e2 + 127) << 23;
	f = f * sc.f;
	return neg ? -f : f;
}

// what repeated fp32 "+1" atomics (kdtree.py:199) yield: exact below 2^24, stuck at 2^24 after
__device__ __host__ __forceinline__ float count_to_f32(uint64_t c)
{
	return c >= 16777216ull ? 16777216.0f : (float)c;
}

} // namespace pg
