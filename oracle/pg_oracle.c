/*
 * pg_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's SD-tree path.
 *
 * This file follows the reference's data structures literally (one SoA column per
 * field, explicit per-node bounding boxes, BFS host loops) so that each function can
 * be read side by side with the Python it restates; the file:line it follows is cited
 * at each function.  It is deliberately NOT how the HIP product is organised.
 *
 * The only intentional departure from the reference's arithmetic is the accumulation
 * contract of DESIGN.md 4.1: the reference adds fp32 values with float atomics in
 * arbitrary order (quadtree.py:93, kdtree.py:199), which is not reproducible even
 * against itself.  Here, and in the product, every node accumulates the exact integer
 * sum of fixed-point-quantised weights (order independent); the fp32 `irradiance` /
 * `vertCount` columns are produced from those sums once, when an iteration's splatting
 * is finalised.  KD counts saturate at 2^24 exactly as repeated fp32 "+1" does.
 *
 * PARITY UNPINNED (see pg_oracle.h).
 */
#include "pg_oracle.h"
#include "pgo_math.h"

#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef __int128 i128;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------- */
/* column containers                                                           */
/* ------------------------------------------------------------------------- */

typedef struct {
	size_t n, cap;
	float *bmin, *bmax;      /* planar would be awkward to grow: stored AoS [n][3] here */
	uint32_t *depth;
	float *vertCount;
	uint8_t *isLeaf;
	uint32_t *qroot;
	uint32_t *left, *right;
	uint64_t *count;         /* exact number of records that passed through the node */
	double maxLeafSize;
	int maxDepth;
} KD;

typedef struct {
	size_t n, cap;
	float *bmin, *bmax;      /* AoS [n][2] */
	uint32_t *depth;
	float *irr;
	uint8_t *isLeaf;
	float *thr;
	uint32_t *c[4];
	i128 *acc;               /* exact fixed-point sum of weights through the node */
	uint32_t *root;          /* rootNodeIndex */
	size_t nroots, rootcap;
	int maxDepth;
	int storeNEE;
} QT;

struct pgo_tree {
	KD kd;
	QT qt;
	/* scratch for column export */
	float *exp_f;
	uint64_t *exp_u64;
	int64_t *exp_i64;
};

static void *xrealloc(void *p, size_t bytes)
{
	void *q = realloc(p, bytes ? bytes : 1);
	if (!q) { fprintf(stderr, "pg_oracle: out of memory (%zu bytes)\n", bytes); abort(); }
	return q;
}

/* quadtree.py:216-247 (resize: zeros, isLeaf defaults to True) */
static void qt_resize(QT *q, size_t newSize)
{
	if (newSize > q->cap) {
		size_t cap = q->cap ? q->cap : 16;
		while (cap < newSize) cap *= 2;
		q->bmin = xrealloc(q->bmin, cap * 2 * sizeof(float));
		q->bmax = xrealloc(q->bmax, cap * 2 * sizeof(float));
		q->depth = xrealloc(q->depth, cap * sizeof(uint32_t));
		q->irr = xrealloc(q->irr, cap * sizeof(float));
		q->isLeaf = xrealloc(q->isLeaf, cap);
		q->thr = xrealloc(q->thr, cap * sizeof(float));
		for (int k = 0; k < 4; ++k) q->c[k] = xrealloc(q->c[k], cap * sizeof(uint32_t));
		q->acc = xrealloc(q->acc, cap * sizeof(i128));
		q->cap = cap;
	}
	for (size_t i = q->n; i < newSize; ++i) {
		q->bmin[2 * i] = q->bmin[2 * i + 1] = 0.0f;
		q->bmax[2 * i] = q->bmax[2 * i + 1] = 0.0f;
		q->depth[i] = 0;
		q->irr[i] = 0.0f;
		q->isLeaf[i] = 1;
		q->thr[i] = 0.0f;
		for (int k = 0; k < 4; ++k) q->c[k][i] = 0;
		q->acc[i] = 0;
	}
	q->n = newSize;
}

static void qt_root_resize(QT *q, size_t newRoots)
{
	if (newRoots > q->rootcap) {
		size_t cap = q->rootcap ? q->rootcap : 16;
		while (cap < newRoots) cap *= 2;
		q->root = xrealloc(q->root, cap * sizeof(uint32_t));
		q->rootcap = cap;
	}
	for (size_t i = q->nroots; i < newRoots; ++i) q->root[i] = 0;
	q->nroots = newRoots;
}

static void qt_free(QT *q)
{
	free(q->bmin); free(q->bmax); free(q->depth); free(q->irr); free(q->isLeaf);
	free(q->thr); for (int k = 0; k < 4; ++k) free(q->c[k]); free(q->acc); free(q->root);
	memset(q, 0, sizeof(*q));
}

/* quadtree.py:350-362 */
static void qt_init(QT *q, int maxDepth, int storeNEE)
{
	memset(q, 0, sizeof(*q));
	qt_resize(q, 1);
	qt_root_resize(q, 1);
	q->root[0] = 0;
	q->thr[0] = INFINITY;
	q->isLeaf[0] = 1;
	q->bmin[0] = q->bmin[1] = 0.0f;
	q->bmax[0] = q->bmax[1] = 1.0f;
	q->maxDepth = maxDepth;
	q->storeNEE = storeNEE;
}

static void qt_copy_all(QT *dst, const QT *src) /* quadtree.py:831-841 + 40-55 */
{
	dst->n = 0;
	qt_resize(dst, src->n);
	memcpy(dst->bmin, src->bmin, src->n * 2 * sizeof(float));
	memcpy(dst->bmax, src->bmax, src->n * 2 * sizeof(float));
	memcpy(dst->depth, src->depth, src->n * sizeof(uint32_t));
	memcpy(dst->irr, src->irr, src->n * sizeof(float));
	memcpy(dst->isLeaf, src->isLeaf, src->n);
	memcpy(dst->thr, src->thr, src->n * sizeof(float));
	for (int k = 0; k < 4; ++k) memcpy(dst->c[k], src->c[k], src->n * sizeof(uint32_t));
	memcpy(dst->acc, src->acc, src->n * sizeof(i128));
	dst->nroots = 0;
	qt_root_resize(dst, src->nroots);
	memcpy(dst->root, src->root, src->nroots * sizeof(uint32_t));
	dst->maxDepth = src->maxDepth;
	dst->storeNEE = src->storeNEE;
}

/* kdtree.py:79-105 */
static void kd_resize(KD *k, size_t newSize)
{
	if (newSize > k->cap) {
		size_t cap = k->cap ? k->cap : 16;
		while (cap < newSize) cap *= 2;
		k->bmin = xrealloc(k->bmin, cap * 3 * sizeof(float));
		k->bmax = xrealloc(k->bmax, cap * 3 * sizeof(float));
		k->depth = xrealloc(k->depth, cap * sizeof(uint32_t));
		k->vertCount = xrealloc(k->vertCount, cap * sizeof(float));
		k->isLeaf = xrealloc(k->isLeaf, cap);
		k->qroot = xrealloc(k->qroot, cap * sizeof(uint32_t));
		k->left = xrealloc(k->left, cap * sizeof(uint32_t));
		k->right = xrealloc(k->right, cap * sizeof(uint32_t));
		k->count = xrealloc(k->count, cap * sizeof(uint64_t));
		k->cap = cap;
	}
	for (size_t i = k->n; i < newSize; ++i) {
		for (int a = 0; a < 3; ++a) k->bmin[3 * i + a] = k->bmax[3 * i + a] = 0.0f;
		k->depth[i] = 0;
		k->vertCount[i] = 0.0f;
		k->isLeaf[i] = 1;
		k->qroot[i] = 0;
		k->left[i] = k->right[i] = 0;
		k->count[i] = 0;
	}
	k->n = newSize;
}

static void kd_free(KD *k)
{
	free(k->bmin); free(k->bmax); free(k->depth); free(k->vertCount); free(k->isLeaf);
	free(k->qroot); free(k->left); free(k->right); free(k->count);
	memset(k, 0, sizeof(*k));
}

/* ------------------------------------------------------------------------- */
/* fixed-point accumulation contract                           */
/* ------------------------------------------------------------------------- */

/* w (fp32) -> trunc(clamp(w) * 2^PGO_FRAC_BITS) as a signed 128-bit integer.
 * NaN -> 0; |w| >= 2^PGO_W_CLAMP_LOG2 (incl. inf) -> +-2^PGO_W_CLAMP_LOG2. */
static i128 quantize(float w)
{
	uint32_t u = pgo_f2u(w);
	uint32_t sign = u >> 31;
	int e = (int)((u >> 23) & 0xffu);
	uint32_t m = u & 0x7fffffu;
	int exp2;
	if (e == 255 && m != 0) return 0; /* NaN */
	if (e == 0) exp2 = -149;
	else { m |= 0x800000u; exp2 = e - 150; }
	if (e == 255 || e - 127 >= PGO_W_CLAMP_LOG2) { m = 0x800000u; exp2 = PGO_W_CLAMP_LOG2 - 23; }
	int shift = exp2 + PGO_FRAC_BITS;
	i128 q;
	if (shift >= 0) q = (i128)m << shift;
	else if (shift > -32) q = (i128)(m >> (-shift));
	else q = 0;
	return sign ? -q : q;
}

/* exact sum -> fp32, round-to-nearest-even once, then exact scaling by 2^-FRAC */
static float acc_to_float(i128 a)
{
	int neg = a < 0;
	u128 mag = neg ? (u128)(-(a + 1)) + 1u : (u128)a;
	if (mag == 0) return 0.0f;
	int msb = 127;
	while (!((mag >> msb) & 1u)) --msb;
	uint32_t mant;
	int exp2 = 0; /* value = mant * 2^exp2 */
	if (msb <= 23) mant = (uint32_t)mag;
	else {
		int sh = msb - 23;
		mant = (uint32_t)(mag >> sh);
		u128 rem = mag & (((u128)1 << sh) - 1u);
		u128 half = (u128)1 << (sh - 1);
		if (rem > half || (rem == half && (mant & 1u))) ++mant;
		if (mant == 0x1000000u) { mant >>= 1; ++sh; }
		exp2 = sh;
	}
	float f = ldexpf((float)mant, exp2 - PGO_FRAC_BITS);
	return neg ? -f : f;
}

/* what repeated fp32 "+= 1" atomics give (kdtree.py:199): exact up to 2^24, then stuck */
static float count_to_float(uint64_t c)
{
	return c >= 16777216ull ? 16777216.0f : (float)c;
}

/* ------------------------------------------------------------------------- */
/* QuadTree                                                                    */
/* ------------------------------------------------------------------------- */

static inline int q_contains(const QT *q, uint32_t node, float x, float y)
{
	/* mi.BoundingBox2f.contains: inclusive on both ends (SURVEY 8c assumption) */
	return x >= q->bmin[2 * node] && x <= q->bmax[2 * node] &&
	       y >= q->bmin[2 * node + 1] && y <= q->bmax[2 * node + 1];
}

/* quadtree.py:96-191 */
static void qt_split(QT *q, const uint32_t *idx, size_t k)
{
	size_t oldSize = q->n;
	qt_resize(q, oldSize + 4 * k);
	for (size_t i = 0; i < k; ++i) {
		uint32_t p = idx[i];
		uint32_t ch[4];
		for (int j = 0; j < 4; ++j) { ch[j] = (uint32_t)(oldSize + 4 * i + j); q->c[j][p] = ch[j]; }
		q->isLeaf[p] = 0;
		uint32_t d = q->depth[p] + 1;
		float irr = q->irr[p] / 4.0f;
		float thr = q->thr[p];
		float mnx = q->bmin[2 * p], mny = q->bmin[2 * p + 1];
		float mxx = q->bmax[2 * p], mxy = q->bmax[2 * p + 1];
		float mdx = (mnx + mxx) / 2.0f, mdy = (mny + mxy) / 2.0f;
		for (int j = 0; j < 4; ++j) { q->depth[ch[j]] = d; q->irr[ch[j]] = irr; q->thr[ch[j]] = thr; }
		/* quadrants: quadtree.py:153-175 */
		q->bmin[2 * ch[0]] = mdx; q->bmin[2 * ch[0] + 1] = mdy; q->bmax[2 * ch[0]] = mxx; q->bmax[2 * ch[0] + 1] = mxy;
		q->bmin[2 * ch[1]] = mnx; q->bmin[2 * ch[1] + 1] = mdy; q->bmax[2 * ch[1]] = mdx; q->bmax[2 * ch[1] + 1] = mxy;
		q->bmin[2 * ch[2]] = mnx; q->bmin[2 * ch[2] + 1] = mny; q->bmax[2 * ch[2]] = mdx; q->bmax[2 * ch[2] + 1] = mdy;
		q->bmin[2 * ch[3]] = mdx; q->bmin[2 * ch[3] + 1] = mny; q->bmax[2 * ch[3]] = mxx; q->bmax[2 * ch[3] + 1] = mdy;
	}
}

/* quadtree.py:194-213 */
static void qt_merge(QT *q, const uint32_t *idx, size_t k)
{
	for (size_t i = 0; i < k; ++i) {
		for (int j = 0; j < 4; ++j) q->c[j][idx[i]] = 0;
		q->isLeaf[idx[i]] = 1;
	}
}

typedef struct { uint32_t *v; size_t n, cap; } U32Vec;
static void vec_push(U32Vec *v, uint32_t x)
{
	if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->v = xrealloc(v->v, v->cap * sizeof(uint32_t)); }
	v->v[v->n++] = x;
}
static void vec_free(U32Vec *v) { free(v->v); v->v = NULL; v->n = v->cap = 0; }

/* The reference's BFS frontier after one level is
 *   concat(concat(child1s, child2s), concat(child3s, child4s))   (quadtree.py:336-339, 551-554, 605-608)
 * i.e. grouped by child slot, not by parent.  next_frontier reproduces that order. */
static void next_frontier(const QT *q, const U32Vec *parents, U32Vec *out)
{
	out->n = 0;
	for (int j = 0; j < 4; ++j)
		for (size_t i = 0; i < parents->n; ++i) vec_push(out, q->c[j][parents->v[i]]);
}

/* quadtree.py:288-345 (rootIndex given) */
static void qt_get_all_leaf(const QT *q, const uint32_t *rootIndex, size_t k, U32Vec *leaves)
{
	U32Vec cur = {0}, nonleaf = {0}, nxt = {0};
	leaves->n = 0;
	for (size_t i = 0; i < k; ++i) vec_push(&cur, q->root[rootIndex[i]]);
	while (cur.n > 0) {
		nonleaf.n = 0;
		for (size_t i = 0; i < cur.n; ++i) {
			if (q->isLeaf[cur.v[i]]) vec_push(leaves, cur.v[i]);
			else vec_push(&nonleaf, cur.v[i]);
		}
		next_frontier(q, &nonleaf, &nxt);
		U32Vec t = cur; cur = nxt; nxt = t;
	}
	vec_free(&cur); vec_free(&nonleaf); vec_free(&nxt);
}

/* quadtree.py:512-560 */
static void qt_set_refinement_threshold(QT *q, const uint32_t *rootIndex, const float *flux, size_t k)
{
	U32Vec cur = {0}, nonleaf = {0}, nxt = {0};
	float *thr = xrealloc(NULL, (k ? k : 1) * sizeof(float));
	for (size_t i = 0; i < k; ++i) { vec_push(&cur, q->root[rootIndex[i]]); thr[i] = flux[i] / 100.0f; }
	int active = cur.n > 0;
	while (active) {
		for (size_t i = 0; i < cur.n; ++i) q->thr[cur.v[i]] = thr[i];
		nonleaf.n = 0;
		size_t m = 0;
		for (size_t i = 0; i < cur.n; ++i) if (!q->isLeaf[cur.v[i]]) ++m;
		active = m > 0;
		if (active) {
			/* children inherit the parent's threshold; frontier order is slot-major */
			float *nthr = xrealloc(NULL, 4 * m * sizeof(float));
			size_t w = 0;
			for (size_t i = 0; i < cur.n; ++i)
				if (!q->isLeaf[cur.v[i]]) { vec_push(&nonleaf, cur.v[i]); nthr[w++] = thr[i]; }
			for (int j = 1; j < 4; ++j) memcpy(nthr + j * m, nthr, m * sizeof(float));
			next_frontier(q, &nonleaf, &nxt);
			U32Vec t = cur; cur = nxt; nxt = t;
			free(thr);
			thr = nthr;
		}
	}
	free(thr);
	vec_free(&cur); vec_free(&nonleaf); vec_free(&nxt);
}

/* quadtree.py:563-637 */
static void qt_refine(QT *q, const uint32_t *rootIndex, size_t k)
{
	U32Vec parent = {0}, small = {0}, valid = {0}, nxt = {0}, leaves = {0}, split = {0};
	/* merge pass */
	for (size_t i = 0; i < k; ++i) vec_push(&parent, q->root[rootIndex[i]]);
	while (parent.n > 0) {
		small.n = valid.n = 0;
		for (size_t i = 0; i < parent.n; ++i) {
			uint32_t p = parent.v[i];
			int notLeaf = !q->isLeaf[p];
			float irr = q->irr[p], thr = q->thr[p];
			if (notLeaf && irr < thr) vec_push(&small, p);
			if (notLeaf && irr >= thr) vec_push(&valid, p);
		}
		qt_merge(q, small.v, small.n);
		next_frontier(q, &valid, &nxt);
		U32Vec t = parent; parent = nxt; nxt = t;
	}
	/* split pass */
	for (;;) {
		qt_get_all_leaf(q, rootIndex, k, &leaves);
		split.n = 0;
		for (size_t i = 0; i < leaves.n; ++i) {
			uint32_t l = leaves.v[i];
			if (q->irr[l] > q->thr[l] && q->depth[l] < (uint32_t)q->maxDepth) vec_push(&split, l);
		}
		if (split.n == 0) break;
		qt_split(q, split.v, split.n);
	}
	vec_free(&parent); vec_free(&small); vec_free(&valid); vec_free(&nxt); vec_free(&leaves); vec_free(&split);
}

/* quadtree.py:695-828.  Result: roots [0..k), then level by level, 4 consecutive
 * children per non-leaf parent in parent order; leaf child indices stay 0. */
static void qt_copy_tree(const QT *src, const uint32_t *rootIndex, size_t k, QT *out)
{
	memset(out, 0, sizeof(*out));
	out->maxDepth = src->maxDepth;
	out->storeNEE = src->storeNEE;
	qt_root_resize(out, k);
	for (size_t i = 0; i < k; ++i) out->root[i] = (uint32_t)i;
	U32Vec node = {0}, parentIdx = {0}, parentChild = {0}, nnode = {0}, nparent = {0}, nchild = {0};
	for (size_t i = 0; i < k; ++i) vec_push(&node, src->root[rootIndex[i]]);
	int active = node.n > 0;
	while (active) {
		size_t numNodes = node.n, oldSize = out->n;
		qt_resize(out, oldSize + numNodes);
		for (size_t i = 0; i < parentIdx.n; ++i)
			out->c[parentChild.v[i] - 1][parentIdx.v[i]] = (uint32_t)(oldSize + i);
		nnode.n = nparent.n = nchild.n = 0;
		for (size_t i = 0; i < numNodes; ++i) {
			uint32_t s = node.v[i];
			size_t d = oldSize + i;
			out->bmin[2 * d] = src->bmin[2 * s]; out->bmin[2 * d + 1] = src->bmin[2 * s + 1];
			out->bmax[2 * d] = src->bmax[2 * s]; out->bmax[2 * d + 1] = src->bmax[2 * s + 1];
			out->depth[d] = src->depth[s];
			out->irr[d] = src->irr[s];
			out->isLeaf[d] = src->isLeaf[s];
			out->thr[d] = src->thr[s];
			out->acc[d] = src->acc[s];
			if (!src->isLeaf[s])
				for (int j = 0; j < 4; ++j) {
					vec_push(&nnode, src->c[j][s]);
					vec_push(&nparent, (uint32_t)d);
					vec_push(&nchild, (uint32_t)(j + 1));
				}
		}
		active = nnode.n > 0;
		U32Vec t;
		t = node; node = nnode; nnode = t;
		t = parentIdx; parentIdx = nparent; nparent = t;
		t = parentChild; parentChild = nchild; nchild = t;
	}
	vec_free(&node); vec_free(&parentIdx); vec_free(&parentChild);
	vec_free(&nnode); vec_free(&nparent); vec_free(&nchild);
}

/* quadtree.py:854-928: returns the first new root id (new ids are consecutive) */
static uint32_t qt_append(QT *dst, const QT *in)
{
	size_t oldRoot = dst->nroots, oldSize = dst->n;
	qt_root_resize(dst, oldRoot + in->nroots);
	qt_resize(dst, oldSize + in->n);
	uint32_t off = (uint32_t)oldSize;
	for (size_t i = 0; i < in->nroots; ++i) dst->root[oldRoot + i] = in->root[i] + off;
	for (size_t i = 0; i < in->n; ++i) {
		size_t d = oldSize + i;
		dst->bmin[2 * d] = in->bmin[2 * i]; dst->bmin[2 * d + 1] = in->bmin[2 * i + 1];
		dst->bmax[2 * d] = in->bmax[2 * i]; dst->bmax[2 * d + 1] = in->bmax[2 * i + 1];
		dst->depth[d] = in->depth[i];
		dst->irr[d] = in->irr[i];
		dst->isLeaf[d] = in->isLeaf[i];
		dst->thr[d] = in->thr[i];
		dst->acc[d] = in->acc[i];
		for (int j = 0; j < 4; ++j) dst->c[j][d] = in->isLeaf[i] ? in->c[j][i] : in->c[j][i] + off;
	}
	return (uint32_t)oldRoot;
}

/* quadtree.py:844-851 */
static void qt_clear_unused(QT *q)
{
	size_t k = q->nroots;
	uint32_t *all = xrealloc(NULL, (k ? k : 1) * sizeof(uint32_t));
	for (size_t i = 0; i < k; ++i) all[i] = (uint32_t)i;
	QT fresh;
	qt_copy_tree(q, all, k, &fresh);
	free(all);
	qt_free(q);
	*q = fresh;
}

/* An exact 128-bit add that several threads may make to one accumulator at once: the low half by a 64-bit
 * fetch-and-add whose returned old value tells THIS add whether it wrapped, the wrap carried into the high half by a
 * second one.  The sum of integers does not depend on the order of the adds, so the threaded stand-alone
 * entry points below (bench.py's CPU columns for S1-S3) give the single-threaded result bit for bit. */
typedef uint64_t __attribute__((may_alias)) u64_alias;
typedef int64_t __attribute__((may_alias)) i64_alias;
static inline void acc_add(i128 *a, i128 w, int mt)
{
	if (!mt) { *a += w; return; }
	u64_alias *lo = (u64_alias *)a;            /* (little-endian: low half first) */
	i64_alias *hi = (i64_alias *)a + 1;
	uint64_t wlo = (uint64_t)(u128)w;
	int64_t whi = (int64_t)(w >> 64);
	uint64_t old = __atomic_fetch_add(lo, wlo, __ATOMIC_RELAXED);
	int64_t up = whi + (int64_t)((uint64_t)(old + wlo) < old);
	if (up) __atomic_fetch_add(hi, up, __ATOMIC_RELAXED);
}

/* quadtree.py:398-441 (one addIrradiancePropagate call, scalar lane) */
static void qt_add_one_mt(QT *q, uint32_t rootIndex, float x, float y, float w, int mt)
{
	uint32_t node = q->root[rootIndex];
	if (!q_contains(q, node, x, y)) return;
	i128 wq = quantize(w);
	for (int guard = 0; guard < 64; ++guard) {
		acc_add(&q->acc[node], wq, mt);
		if (q->isLeaf[node]) return;
		uint32_t next = node;
		for (int j = 0; j < 4; ++j) { /* sequential overwrite: highest containing child wins */
			uint32_t ch = q->c[j][node];
			if (q_contains(q, ch, x, y)) next = ch;
		}
		if (next == node) return; /* unreachable for p inside the node (children tile it) */
		node = next;
	}
}

/* quadtree.py:931-998, scalar lane.  Draw order per visited node: next_2d (x then y), next_1d. */
static void qt_sample_one(const QT *q, uint32_t rootIndex, pgo_pcg32 *rng, int active, float dir[3])
{
	uint32_t node = q->root[rootIndex];
	float px = 0.0f, py = 0.0f;
	for (int guard = 0; active && guard < 64; ++guard) {
		int isLeaf = q->isLeaf[node];
		float mnx = q->bmin[2 * node], mny = q->bmin[2 * node + 1];
		float mxx = q->bmax[2 * node], mxy = q->bmax[2 * node + 1];
		float u = pgo_pcg32_next_f32(rng);
		float v = pgo_pcg32_next_f32(rng);
		if (isLeaf) {
			float tx = mxx - mnx, ty = mxy - mny;
			float ux = u * tx, uy = v * ty;
			px = mnx + ux; py = mny + uy;
		}
		active = active && !isLeaf;
		float xi = pgo_pcg32_next_f32(rng); /* quadtree.py:980 draws for every lane in the iteration */
		if (!active) break;
		uint32_t ch[4];
		float c[4];
		for (int j = 0; j < 4; ++j) { ch[j] = q->c[j][node]; c[j] = q->irr[ch[j]]; }
		c[1] += c[0]; c[2] += c[1]; c[3] += c[2];
		float s = xi * c[3];
		uint32_t next = node;
		if (s < c[0]) next = ch[0];
		if (c[0] <= s && s < c[1]) next = ch[1];
		if (c[1] <= s && s < c[2]) next = ch[2];
		if (c[2] <= s) next = ch[3];
		if (next == node) break; /* NaN energies: the reference would spin forever */
		node = next;
	}
	pgo_canonical_to_dir(px, py, dir);
}

/* quadtree.py:1001-1101, scalar lane */
static float qt_pdf_one(const QT *q, uint32_t rootIndex, const float d[3], int active)
{
	uint32_t node = q->root[rootIndex];
	float pdf = 1.0f;
	float pos[2];
	pgo_dir_to_canonical(d[0], d[1], d[2], pos);
	for (int guard = 0; active && guard < 64; ++guard) {
		if (q->isLeaf[node]) { pdf *= PGO_INV_FOUR_PI_F; break; }
		uint32_t ch[4];
		int t[4];
		for (int j = 0; j < 4; ++j) { ch[j] = q->c[j][node]; t[j] = q_contains(q, ch[j], pos[0], pos[1]); }
		float nodeIrr = q->irr[node];
		float childIrr = t[0] ? q->irr[ch[0]] : t[1] ? q->irr[ch[1]] : t[2] ? q->irr[ch[2]] : t[3] ? q->irr[ch[3]] : 0.0f;
		float ratio = (4.0f * childIrr) / nodeIrr;
		pdf = pdf * ratio;
		if (pdf != pdf) { pdf = 0.0f; break; }
		uint32_t next = node;
		for (int j = 0; j < 4; ++j) if (t[j]) next = ch[j];
		if (next == node) break; /* position outside every child: reference would spin */
		node = next;
	}
	return pdf;
}

/* quadtree.py:640-683 via 679-683 */
static void qt_reset_all(QT *q)
{
	U32Vec cur = {0}, nonleaf = {0}, nxt = {0};
	for (size_t i = 0; i < q->nroots; ++i) vec_push(&cur, q->root[i]);
	while (cur.n > 0) {
		nonleaf.n = 0;
		for (size_t i = 0; i < cur.n; ++i) {
			q->irr[cur.v[i]] = 0.0f;
			q->acc[cur.v[i]] = 0;
			if (!q->isLeaf[cur.v[i]]) vec_push(&nonleaf, cur.v[i]);
		}
		if (nonleaf.n == 0) break;
		next_frontier(q, &nonleaf, &nxt);
		U32Vec t = cur; cur = nxt; nxt = t;
	}
	vec_free(&cur); vec_free(&nonleaf); vec_free(&nxt);
}

/* ------------------------------------------------------------------------- */
/* KDTree                                                                      */
/* ------------------------------------------------------------------------- */

static inline int kd_contains(const KD *k, uint32_t node, const float p[3])
{
	for (int a = 0; a < 3; ++a)
		if (!(p[a] >= k->bmin[3 * node + a] && p[a] <= k->bmax[3 * node + a])) return 0;
	return 1;
}

/* kdtree.py:435-470, scalar lane */
static uint32_t kd_leaf_index(const KD *k, const float p[3], int active)
{
	uint32_t node = 0;
	int search = kd_contains(k, 0, p) && active;
	for (int guard = 0; search && guard < 64; ++guard) {
		if (k->isLeaf[node]) break;
		uint32_t l = k->left[node], r = k->right[node];
		uint32_t next = node;
		if (kd_contains(k, l, p)) next = l;
		if (kd_contains(k, r, p)) next = r;
		if (next == node) break;
		node = next;
	}
	return node;
}

/* kdtree.py:229-323 */
static void kd_split(pgo_tree *t, const uint32_t *idx, size_t n)
{
	KD *k = &t->kd;
	size_t oldSize = k->n;
	kd_resize(k, oldSize + 2 * n);
	uint32_t *parentRoots = xrealloc(NULL, (n ? n : 1) * sizeof(uint32_t));
	for (size_t i = 0; i < n; ++i) {
		uint32_t p = idx[i];
		uint32_t l = (uint32_t)(oldSize + 2 * i), r = l + 1;
		k->left[p] = l; k->right[p] = r;
		k->isLeaf[p] = 0;
		uint32_t depth = k->depth[p];
		k->depth[l] = k->depth[r] = depth + 1;
		float vc = k->vertCount[p];
		if (vc > 0.0f) vc = vc / 2.0f;
		k->vertCount[l] = k->vertCount[r] = vc;
		uint32_t axis = depth % 3;
		for (int a = 0; a < 3; ++a) {
			float mn = k->bmin[3 * p + a], mx = k->bmax[3 * p + a];
			float mid = (mn + mx) / 2.0f;
			k->bmin[3 * l + a] = mn; k->bmax[3 * l + a] = ((uint32_t)a == axis) ? mid : mx;
			k->bmin[3 * r + a] = ((uint32_t)a == axis) ? mid : mn; k->bmax[3 * r + a] = mx;
		}
		k->qroot[l] = k->qroot[p];
		parentRoots[i] = k->qroot[p];
	}
	QT clone;
	qt_copy_tree(&t->qt, parentRoots, n, &clone);
	uint32_t firstNewRoot = qt_append(&t->qt, &clone);
	qt_free(&clone);
	for (size_t i = 0; i < n; ++i) k->qroot[oldSize + 2 * i + 1] = firstNewRoot + (uint32_t)i;
	free(parentRoots);
}

/* ------------------------------------------------------------------------- */
/* public API                                                                  */
/* ------------------------------------------------------------------------- */

pgo_tree *pgo_tree_new(void)
{
	pgo_tree *t = calloc(1, sizeof(*t));
	kd_resize(&t->kd, 1);
	t->kd.isLeaf[0] = 1;
	for (int a = 0; a < 3; ++a) { t->kd.bmin[a] = 0.0f; t->kd.bmax[a] = 1.0f; }
	t->kd.maxLeafSize = 1.0;
	t->kd.maxDepth = 10;
	qt_init(&t->qt, 20, 0);
	return t;
}

void pgo_tree_free(pgo_tree *t)
{
	if (!t) return;
	kd_free(&t->kd); qt_free(&t->qt);
	free(t->exp_f); free(t->exp_u64); free(t->exp_i64);
	free(t);
}

void pgo_tree_setup(pgo_tree *t, const float bmin[3], const float bmax[3], int kdMaxDepth,
                    int quadMaxDepth, int storeNEE)
{
	for (int a = 0; a < 3; ++a) { t->kd.bmin[a] = bmin[a]; t->kd.bmax[a] = bmax[a]; }
	t->kd.maxDepth = kdMaxDepth;
	t->qt.maxDepth = quadMaxDepth;
	t->qt.storeNEE = storeNEE;
}

void pgo_tree_copy_from(pgo_tree *dst, const pgo_tree *src)
{
	const KD *s = &src->kd;
	KD *d = &dst->kd;
	d->n = 0;
	kd_resize(d, s->n);
	memcpy(d->bmin, s->bmin, s->n * 3 * sizeof(float));
	memcpy(d->bmax, s->bmax, s->n * 3 * sizeof(float));
	memcpy(d->depth, s->depth, s->n * sizeof(uint32_t));
	memcpy(d->vertCount, s->vertCount, s->n * sizeof(float));
	memcpy(d->isLeaf, s->isLeaf, s->n);
	memcpy(d->qroot, s->qroot, s->n * sizeof(uint32_t));
	memcpy(d->left, s->left, s->n * sizeof(uint32_t));
	memcpy(d->right, s->right, s->n * sizeof(uint32_t));
	memcpy(d->count, s->count, s->n * sizeof(uint64_t));
	d->maxLeafSize = s->maxLeafSize;
	d->maxDepth = s->maxDepth;
	qt_copy_all(&dst->qt, &src->qt);
}

/* The stand-alone entry points below loop over independent lanes (every lane has its own sampler stream and outputs):
 * they run on pgo_set_threads() threads -- 1 unless a caller asked for more -- without changing a bit of the result. */
int pgo_threads_in_effect(void); /* pg_oracle_render.c */
#ifdef _OPENMP
#define PGO_LANES _Pragma("omp parallel for schedule(static) num_threads(pgo_threads_in_effect()) if (pgo_threads_in_effect() > 1)")
#else
#define PGO_LANES
#endif

void pgo_get_leaf_node_index(const pgo_tree *t, size_t n, const float *p, const uint8_t *active,
                             uint32_t *out)
{
	PGO_LANES
	for (size_t i = 0; i < n; ++i) {
		float q[3] = { p[i], p[n + i], p[2 * n + i] };
		out[i] = kd_leaf_index(&t->kd, q, active ? active[i] : 1);
	}
}

void pgo_sample_quadtree(const pgo_tree *t, size_t n, const uint32_t *rootIndex, uint64_t *st,
                         uint64_t *inc, const uint8_t *active, float *dir)
{
	for (size_t i = 0; i < n; ++i) {
		pgo_pcg32 r = { st[i], inc[i] };
		float d[3];
		qt_sample_one(&t->qt, rootIndex[i], &r, active ? active[i] : 1, d);
		st[i] = r.state;
		dir[i] = d[0]; dir[n + i] = d[1]; dir[2 * n + i] = d[2];
	}
}

void pgo_pdf_quadtree(const pgo_tree *t, size_t n, const uint32_t *rootIndex, const float *dir,
                      const uint8_t *active, float *pdf)
{
	for (size_t i = 0; i < n; ++i) {
		float d[3] = { dir[i], dir[n + i], dir[2 * n + i] };
		pdf[i] = qt_pdf_one(&t->qt, rootIndex[i], d, active ? active[i] : 1);
	}
}

void pgo_sample(const pgo_tree *t, size_t n, const float *p, uint64_t *st, uint64_t *inc,
                const uint8_t *active, float *dir, float *pdf)
{
	PGO_LANES
	for (size_t i = 0; i < n; ++i) {
		int act = active ? active[i] : 1;
		float q[3] = { p[i], p[n + i], p[2 * n + i] };
		uint32_t leaf = kd_leaf_index(&t->kd, q, act);
		uint32_t root = act ? t->kd.qroot[leaf] : 0; /* masked gather yields 0 */
		pgo_pcg32 r = { st[i], inc[i] };
		float d[3];
		qt_sample_one(&t->qt, root, &r, act, d);
		st[i] = r.state;
		dir[i] = d[0]; dir[n + i] = d[1]; dir[2 * n + i] = d[2];
		pdf[i] = qt_pdf_one(&t->qt, root, d, act);
	}
}

void pgo_pdf(const pgo_tree *t, size_t n, const float *p, const float *dir, const uint8_t *active,
             float *pdf)
{
	PGO_LANES
	for (size_t i = 0; i < n; ++i) {
		int act = active ? active[i] : 1;
		float q[3] = { p[i], p[n + i], p[2 * n + i] };
		float d[3] = { dir[i], dir[n + i], dir[2 * n + i] };
		uint32_t leaf = kd_leaf_index(&t->kd, q, act);
		uint32_t root = act ? t->kd.qroot[leaf] : 0;
		pdf[i] = qt_pdf_one(&t->qt, root, d, act);
	}
}

void pgo_add_data_propagate(pgo_tree *t, size_t m, const float *pos, const float *dir,
                            const float *radiance, const float *woPdf, const float *dirNee,
                            const float *radNeeLum)
{
	KD *k = &t->kd;
	/* several threads: the counts are exact integers and the irradiance sums exact 128-bit integers (acc_add), so the
	 * records may be added in any order by any number of threads.  The KD counts go to a private array per thread first
	 * (every record bumps the root and the nodes under it: one shared counter would be all the threads do), summed at the end;
	 * the quadtree sums are added in place with carry-correct atomic adds. */
	const int nth = pgo_threads_in_effect();
	const int mt = nth > 1 && m > 1024;
	uint64_t *part = NULL; /* [thread][node] */
	if (mt && (size_t)nth * k->n * sizeof(uint64_t) <= ((size_t)256 << 20)) part = calloc((size_t)nth * k->n, sizeof(uint64_t));
#ifdef _OPENMP
#pragma omp parallel num_threads(nth) if (mt)
#endif
	{
#ifdef _OPENMP
		uint64_t *mine = part ? part + (size_t)omp_get_thread_num() * k->n : NULL;
#pragma omp for schedule(static)
#else
		uint64_t *mine = NULL;
#endif
		for (size_t i = 0; i < m; ++i) {
			float p[3] = { pos[i], pos[m + i], pos[2 * m + i] };
			/* kdtree.py:185-217: vertCount += 1 on every visited node */
			uint32_t node = 0;
			int active = kd_contains(k, 0, p);
			for (int guard = 0; active && guard < 64; ++guard) {
				if (mine) mine[node] += 1;
				else if (mt) __atomic_fetch_add(&k->count[node], 1, __ATOMIC_RELAXED);
				else k->count[node] += 1;
				if (k->isLeaf[node]) break;
				uint32_t l = k->left[node], r = k->right[node], next = node;
				if (kd_contains(k, l, p)) next = l;
				if (kd_contains(k, r, p)) next = r;
				if (next == node) break;
				node = next;
			}
			/* kdtree.py:224-225: unmasked gather -> out-of-bbox records use node 0's tree */
			uint32_t root = k->qroot[node];
			float wp = woPdf[i];
			float w = wp > 0.0f ? radiance[i] / wp : 0.0f;            /* quadtree.py:451 */
			qt_add_one_mt(&t->qt, root, dir[i], dir[m + i], w, mt);
			if (t->qt.storeNEE) {
				float wn = wp > 0.0f ? radNeeLum[i] / wp : 0.0f;      /* quadtree.py:461-462 */
				qt_add_one_mt(&t->qt, root, dirNee[i], dirNee[m + i], wn, mt);
			}
		}
	}
	if (part) {
		for (int th = 0; th < nth; ++th)
			for (size_t j = 0; j < k->n; ++j) k->count[j] += part[(size_t)th * k->n + j];
		free(part);
	}
}

size_t pgo_process_records(size_t numRays, size_t maxDepth, const float *Lfinal,
                           const uint8_t *recActive, const float *recPos, const float *recDir,
                           const float *recBsdf, const float *recThrBsdf, const float *recThrRad,
                           const float *recRadNee, const float *recDirNee, const float *recWoPdf,
                           float *oPos, float *oDir, float *oRad, float *oWoPdf, float *oDirNee,
                           float *oRadNeeLum)
{
	size_t S = numRays * maxDepth, kept = 0;
	/* pass 1: count so the planar outputs can be laid out with the final stride */
	/* (outputs are written with stride = S; the caller slices [0,kept) of each plane) */
	for (size_t g = 0; g < S; ++g) {
		size_t ray = g / maxDepth;
		float in[3];
		for (int ch = 0; ch < 3; ++ch) {
			/* path_guiding_integrator.py:443-449 */
			float out = (Lfinal[ch * numRays + ray] - recThrRad[ch * S + g]) / recThrBsdf[ch * S + g];
			if (out != out) out = 0.0f;
			float v = out / recBsdf[ch * S + g];
			if (v != v) v = 0.0f;
			in[ch] = v;
		}
		float radiance = pgo_luminance(in[0], in[1], in[2]);        /* :452 */
		if (radiance != radiance) radiance = 0.0f;                    /* :466 */
		float nee[3];
		for (int ch = 0; ch < 3; ++ch) {
			nee[ch] = recRadNee[ch * S + g];
			if (nee[ch] != nee[ch]) nee[ch] = 0.0f;                   /* :467 */
		}
		float neeLum = pgo_luminance(nee[0], nee[1], nee[2]);
		int bothZero = (radiance == 0.0f) && (neeLum == 0.0f);        /* :470-472 */
		float wp = recWoPdf[g];
		int keep = recActive[g] && !bothZero && !(wp == 0.0f) && !(wp != wp); /* :475-478 */
		if (!keep) continue;
		for (int a = 0; a < 3; ++a) oPos[a * S + kept] = recPos[a * S + g];
		for (int a = 0; a < 2; ++a) oDir[a * S + kept] = recDir[a * S + g];
		for (int a = 0; a < 2; ++a) oDirNee[a * S + kept] = recDirNee[a * S + g];
		oRad[kept] = radiance;
		oWoPdf[kept] = wp;
		oRadNeeLum[kept] = neeLum;
		++kept;
	}
	return kept;
}

void pgo_finalize_accumulators(pgo_tree *t)
{
	for (size_t i = 0; i < t->kd.n; ++i) t->kd.vertCount[i] = count_to_float(t->kd.count[i]);
	for (size_t i = 0; i < t->qt.n; ++i) t->qt.irr[i] = acc_to_float(t->qt.acc[i]);
}

/* kdtree.py:327-330 */
void pgo_set_refinement_threshold(pgo_tree *t, int iteration)
{
	t->kd.maxLeafSize = 12000.0 * sqrt(pow(2.0, (double)iteration));
}

/* kdtree.py:333-358 */
void pgo_kd_refine(pgo_tree *t)
{
	KD *k = &t->kd;
	float thr = (float)k->maxLeafSize; /* Dr.Jit casts the python scalar to Float */
	U32Vec split = {0};
	for (;;) {
		split.n = 0;
		size_t n = k->n;
		for (size_t i = 0; i < n; ++i)
			if (k->isLeaf[i] && k->vertCount[i] > thr && k->depth[i] < (uint32_t)k->maxDepth)
				vec_push(&split, (uint32_t)i);
		if (split.n == 0) break;
		kd_split(t, split.v, split.n);
	}
	vec_free(&split);
}

static size_t kd_leaf_roots(const pgo_tree *t, uint32_t **rootsOut)
{
	const KD *k = &t->kd;
	size_t m = 0;
	for (size_t i = 0; i < k->n; ++i) m += k->isLeaf[i];
	uint32_t *roots = xrealloc(NULL, (m ? m : 1) * sizeof(uint32_t));
	size_t w = 0;
	for (size_t i = 0; i < k->n; ++i) if (k->isLeaf[i]) roots[w++] = k->qroot[i];
	*rootsOut = roots;
	return m;
}

/* kdtree.py:503-514 */
void pgo_set_quadtree_refinement_threshold(pgo_tree *t)
{
	uint32_t *roots;
	size_t m = kd_leaf_roots(t, &roots);
	float *flux = xrealloc(NULL, (m ? m : 1) * sizeof(float));
	for (size_t i = 0; i < m; ++i) flux[i] = t->qt.irr[t->qt.root[roots[i]]];
	qt_set_refinement_threshold(&t->qt, roots, flux, m);
	free(flux); free(roots);
}

/* kdtree.py:517-524 */
void pgo_refine_all_quadtree(pgo_tree *t)
{
	uint32_t *roots;
	size_t m = kd_leaf_roots(t, &roots);
	qt_refine(&t->qt, roots, m);
	free(roots);
}

void pgo_clean_unused_quadtree(pgo_tree *t) { qt_clear_unused(&t->qt); }

/* kdtree.py:401-432 + quadtree.py:679-683 */
void pgo_reset(pgo_tree *t)
{
	KD *k = &t->kd;
	/* BFS from the root; after a refine every node is reachable (Appendix A13) but stay literal */
	U32Vec cur = {0}, nxt = {0};
	vec_push(&cur, 0);
	while (cur.n > 0) {
		nxt.n = 0;
		for (size_t i = 0; i < cur.n; ++i) {
			k->vertCount[cur.v[i]] = 0.0f;
			k->count[cur.v[i]] = 0;
		}
		/* children of non-leaf: concat(lefts, rights) */
		for (size_t i = 0; i < cur.n; ++i) if (!k->isLeaf[cur.v[i]]) vec_push(&nxt, k->left[cur.v[i]]);
		for (size_t i = 0; i < cur.n; ++i) if (!k->isLeaf[cur.v[i]]) vec_push(&nxt, k->right[cur.v[i]]);
		U32Vec tmp = cur; cur = nxt; nxt = tmp;
	}
	vec_free(&cur); vec_free(&nxt);
	qt_reset_all(&t->qt);
}

/* path_guiding_integrator.py:566-586 (+553-563) */
void pgo_refine_and_prepare(pgo_tree *current, pgo_tree *prev, int iteration)
{
	pgo_finalize_accumulators(current);
	pgo_set_refinement_threshold(current, iteration);
	pgo_kd_refine(current);
	pgo_set_quadtree_refinement_threshold(current);
	pgo_refine_all_quadtree(current);
	pgo_clean_unused_quadtree(current);
	pgo_tree_copy_from(prev, current);
	pgo_reset(current);
}

void pgo_kd_split(pgo_tree *t, size_t n, const uint32_t *idx) { kd_split(t, idx, n); }
void pgo_quad_split(pgo_tree *t, size_t n, const uint32_t *idx) { qt_split(&t->qt, idx, n); }
void pgo_kd_all_leaves(const pgo_tree *t, uint32_t *out, size_t *n_out)
{
	size_t w = 0;
	for (size_t i = 0; i < t->kd.n; ++i) if (t->kd.isLeaf[i]) out[w++] = (uint32_t)i;
	*n_out = w;
}
void pgo_quad_all_leaves(const pgo_tree *t, uint32_t *out, size_t *n_out)
{
	U32Vec leaves = {0};
	size_t k = t->qt.nroots;
	uint32_t *all = xrealloc(NULL, (k ? k : 1) * sizeof(uint32_t));
	for (size_t i = 0; i < k; ++i) all[i] = (uint32_t)i;
	qt_get_all_leaf(&t->qt, all, k, &leaves);
	memcpy(out, leaves.v, leaves.n * sizeof(uint32_t));
	*n_out = leaves.n;
	free(all); vec_free(&leaves);
}

size_t pgo_kd_size(const pgo_tree *t) { return t->kd.n; }
size_t pgo_quad_size(const pgo_tree *t) { return t->qt.n; }
size_t pgo_quad_roots(const pgo_tree *t) { return t->qt.nroots; }
double pgo_kd_max_leaf_size(const pgo_tree *t) { return t->kd.maxLeafSize; }
int pgo_kd_max_depth(const pgo_tree *t) { return t->kd.maxDepth; }
int pgo_quad_max_depth(const pgo_tree *t) { return t->qt.maxDepth; }
int pgo_quad_store_nee(const pgo_tree *t) { return t->qt.storeNEE; }

const void *pgo_kd_column(const pgo_tree *t, const char *name)
{
	const KD *k = &t->kd;
	if (!strcmp(name, "bbox_min")) return k->bmin;   /* AoS [n][3] */
	if (!strcmp(name, "bbox_max")) return k->bmax;
	if (!strcmp(name, "depth")) return k->depth;
	if (!strcmp(name, "vertCount")) return k->vertCount;
	if (!strcmp(name, "isLeaf")) return k->isLeaf;
	if (!strcmp(name, "quadTreeRootIndex")) return k->qroot;
	if (!strcmp(name, "child_left_index")) return k->left;
	if (!strcmp(name, "child_right_index")) return k->right;
	if (!strcmp(name, "count")) return k->count;
	return NULL;
}

const void *pgo_quad_column(const pgo_tree *tc, const char *name)
{
	pgo_tree *t = (pgo_tree *)tc;
	const QT *q = &t->qt;
	if (!strcmp(name, "rootNodeIndex")) return q->root;
	if (!strcmp(name, "bbox_min")) return q->bmin;   /* AoS [n][2] */
	if (!strcmp(name, "bbox_max")) return q->bmax;
	if (!strcmp(name, "depth")) return q->depth;
	if (!strcmp(name, "irradiance")) return q->irr;
	if (!strcmp(name, "isLeaf")) return q->isLeaf;
	if (!strcmp(name, "refinementThreshold")) return q->thr;
	if (!strcmp(name, "child_1_index")) return q->c[0];
	if (!strcmp(name, "child_2_index")) return q->c[1];
	if (!strcmp(name, "child_3_index")) return q->c[2];
	if (!strcmp(name, "child_4_index")) return q->c[3];
	if (!strcmp(name, "acc_lo")) {
		t->exp_u64 = xrealloc(t->exp_u64, (q->n ? q->n : 1) * sizeof(uint64_t));
		for (size_t i = 0; i < q->n; ++i) t->exp_u64[i] = (uint64_t)(u128)q->acc[i];
		return t->exp_u64;
	}
	if (!strcmp(name, "acc_hi")) {
		t->exp_i64 = xrealloc(t->exp_i64, (q->n ? q->n : 1) * sizeof(int64_t));
		for (size_t i = 0; i < q->n; ++i) t->exp_i64[i] = (int64_t)(q->acc[i] >> 64);
		return t->exp_i64;
	}
	return NULL;
}

void pgo_tree_load(pgo_tree *t, size_t nkd, const float *kbmin, const float *kbmax,
                   const uint32_t *kdepth, const float *kvc, const uint8_t *kleaf,
                   const uint32_t *kqroot, const uint32_t *kleft, const uint32_t *kright,
                   double maxLeafSize, int kdMaxDepth, size_t nroots, const uint32_t *qroot,
                   size_t nq, const float *qbmin, const float *qbmax, const uint32_t *qdepth,
                   const float *qirr, const uint8_t *qleaf, const float *qthr, const uint32_t *c1,
                   const uint32_t *c2, const uint32_t *c3, const uint32_t *c4, int qMaxDepth,
                   int storeNEE)
{
	KD *k = &t->kd;
	k->n = 0;
	kd_resize(k, nkd);
	memcpy(k->bmin, kbmin, nkd * 3 * sizeof(float));
	memcpy(k->bmax, kbmax, nkd * 3 * sizeof(float));
	memcpy(k->depth, kdepth, nkd * sizeof(uint32_t));
	memcpy(k->vertCount, kvc, nkd * sizeof(float));
	memcpy(k->isLeaf, kleaf, nkd);
	memcpy(k->qroot, kqroot, nkd * sizeof(uint32_t));
	memcpy(k->left, kleft, nkd * sizeof(uint32_t));
	memcpy(k->right, kright, nkd * sizeof(uint32_t));
	k->maxLeafSize = maxLeafSize;
	k->maxDepth = kdMaxDepth;
	QT *q = &t->qt;
	q->n = 0;
	qt_resize(q, nq);
	memcpy(q->bmin, qbmin, nq * 2 * sizeof(float));
	memcpy(q->bmax, qbmax, nq * 2 * sizeof(float));
	memcpy(q->depth, qdepth, nq * sizeof(uint32_t));
	memcpy(q->irr, qirr, nq * sizeof(float));
	memcpy(q->isLeaf, qleaf, nq);
	memcpy(q->thr, qthr, nq * sizeof(float));
	memcpy(q->c[0], c1, nq * sizeof(uint32_t));
	memcpy(q->c[1], c2, nq * sizeof(uint32_t));
	memcpy(q->c[2], c3, nq * sizeof(uint32_t));
	memcpy(q->c[3], c4, nq * sizeof(uint32_t));
	q->nroots = 0;
	qt_root_resize(q, nroots);
	memcpy(q->root, qroot, nroots * sizeof(uint32_t));
	q->maxDepth = qMaxDepth;
	q->storeNEE = storeNEE;
}

/* ---- vector helpers for known-answer tests ---- */
void pgo_canonical_to_dir_v(size_t n, const float *p2, float *d3)
{
	for (size_t i = 0; i < n; ++i) {
		float d[3];
		pgo_canonical_to_dir(p2[i], p2[n + i], d);
		d3[i] = d[0]; d3[n + i] = d[1]; d3[2 * n + i] = d[2];
	}
}
void pgo_dir_to_canonical_v(size_t n, const float *d3, float *p2)
{
	for (size_t i = 0; i < n; ++i) {
		float p[2];
		pgo_dir_to_canonical(d3[i], d3[n + i], d3[2 * n + i], p);
		p2[i] = p[0]; p2[n + i] = p[1];
	}
}
void pgo_rng_seed(size_t n, uint32_t seed, uint32_t lane0, uint64_t *state, uint64_t *inc)
{
	for (size_t i = 0; i < n; ++i) {
		pgo_pcg32 r;
		pgo_pcg32_seed(&r, seed, lane0 + (uint32_t)i);
		state[i] = r.state; inc[i] = r.inc;
	}
}
void pgo_rng_next_f32(size_t n, uint64_t *state, const uint64_t *inc, float *out)
{
	for (size_t i = 0; i < n; ++i) {
		pgo_pcg32 r = { state[i], inc[i] };
		out[i] = pgo_pcg32_next_f32(&r);
		state[i] = r.state;
	}
}
void pgo_quantize_v(size_t n, const float *w, uint64_t *lo, int64_t *hi)
{
	for (size_t i = 0; i < n; ++i) {
		i128 q = quantize(w[i]);
		lo[i] = (uint64_t)(u128)q; hi[i] = (int64_t)(q >> 64);
	}
}
void pgo_acc_to_float_v(size_t n, const uint64_t *lo, const int64_t *hi, float *out)
{
	for (size_t i = 0; i < n; ++i) {
		i128 a = (i128)(((u128)(uint64_t)hi[i] << 64) | (u128)lo[i]); /* (shifting a negative value is undefined) */
		out[i] = acc_to_float(a);
	}
}
void pgo_sincos_v(size_t n, const float *phi, float *s, float *c)
{
	for (size_t i = 0; i < n; ++i) pgo_sincos(phi[i], &s[i], &c[i]);
}
void pgo_atan2_v(size_t n, const float *y, const float *x, float *out)
{
	for (size_t i = 0; i < n; ++i) out[i] = pgo_atan2(y[i], x[i]);
}
void pgo_math1_v(size_t n, int which, const float *x, float *out)
{
	for (size_t i = 0; i < n; ++i)
		out[i] = which == 0 ? pgo_exp(x[i]) : (which == 1 ? pgo_log(x[i]) : (which == 2 ? pgo_erf(x[i]) : pgo_erfinv(x[i])));
}

/* ---- scalar entry points for the oracle's integrator loop (pg_oracle_render.c) ---- */
uint32_t pgo_i_quadtree_of(const pgo_tree *t, const float p[3], int active)
{
	/* kdtree.py:481-482: masked gather of quadTreeRootIndex at the leaf */
	uint32_t leaf = kd_leaf_index(&t->kd, p, active);
	return active ? t->kd.qroot[leaf] : 0;
}
void pgo_i_sample(const pgo_tree *t, uint32_t root, uint64_t *state, uint64_t inc, int active, float dir[3])
{
	pgo_pcg32 r = { *state, inc };
	qt_sample_one(&t->qt, root, &r, active, dir);
	*state = r.state;
}
float pgo_i_pdf(const pgo_tree *t, uint32_t root, const float dir[3], int active)
{
	return qt_pdf_one(&t->qt, root, dir, active);
}
