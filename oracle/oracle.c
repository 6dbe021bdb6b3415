/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h for the parity statement).
 *
 * Plain-C restatement of the reference's Marin IBDWT path.  Registers are kept exactly as the
 * reference keeps them on the device: n weighted digits x[k] = digit_k * w_k mod P in natural digit
 * order, weakly carried (engine_gpu.h:1476-1484, kernels/marin.cl:1696-1728,2198-2216).  The
 * transform is the reference's pair formulation (marin.cl:379-392): two
 * length-n/2 transforms over (x[2i], x[2i+1]) closed by a multiply mod (t^2 - rho).  The butterfly
 * schedule itself is free (only the cyclic convolution matters), so a straightforward
 * radix-5 x radix-2 DIF/DIT is used here instead of the reference's per-size kernel chain.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* small transforms run single-threaded: OpenMP fork/join would dominate */
#define OMP_BIG (c->m >= 32768)

#define MOD_P 0xffffffff00000001ull /* arith.h:24 */
#define MOD_MP64 0xffffffffu        /* arith.h:25 */

/* ---- field: arith.h:27-72 ---- */
static inline uint64_t mod_add(uint64_t lhs, uint64_t rhs) { return lhs + rhs + ((lhs >= MOD_P - rhs) ? MOD_MP64 : 0); }
static inline uint64_t mod_sub(uint64_t lhs, uint64_t rhs) { return lhs - rhs - ((lhs < rhs) ? MOD_MP64 : 0); }
static inline uint64_t reduce(uint64_t lo, uint64_t hi) {
  /* arith.h:31-36: hih*2^96 + hil*2^64 + lo = lo + hil*2^32 - (hih + hil) */
  const uint64_t r = (lo >= MOD_P) ? lo - MOD_P : lo;
  return mod_sub(mod_add(r, (hi << 32) - (uint32_t)hi), hi >> 32);
}
static inline uint64_t mod_mul(uint64_t lhs, uint64_t rhs) {
  const u128 t = (u128)lhs * rhs;
  return reduce((uint64_t)t, (uint64_t)(t >> 64));
}
static uint64_t mod_pow(uint64_t lhs, uint64_t e) { /* arith.h:56-68 */
  if (e == 0) return 1;
  uint64_t r = 1, y = lhs;
  for (uint64_t i = e; i != 1; i /= 2) {
    if (i % 2 != 0) r = mod_mul(r, y);
    y = mod_mul(y, y);
  }
  return mod_mul(r, y);
}
static uint64_t mod_invert(uint64_t lhs) { return mod_pow(lhs, MOD_P - 2); }               /* arith.h:70 */
static uint64_t mod_root_nth(uint64_t n) { return mod_pow(7, (MOD_P - 1) / n); }            /* arith.h:72 */

uint64_t orc_mod_add(uint64_t a, uint64_t b) { return mod_add(a, b); }
uint64_t orc_mod_sub(uint64_t a, uint64_t b) { return mod_sub(a, b); }
uint64_t orc_mod_mul(uint64_t a, uint64_t b) { return mod_mul(a, b); }
uint64_t orc_mod_pow(uint64_t a, uint64_t e) { return mod_pow(a, e); }
uint64_t orc_mod_invert(uint64_t a) { return mod_invert(a); }

/* ---- digit helpers: arith.h:75-99, marin.cl:185-300 ---- */
static inline uint32_t adc(uint64_t lhs, uint8_t width, uint64_t* carry) {
  const uint64_t s = lhs + *carry;
  const uint64_t c = (s < lhs) ? 1 : 0;
  *carry = (s >> width) + (c << (64 - width));
  return (uint32_t)s & ((1u << width) - 1);
}
static inline uint32_t adc_mul(uint64_t lhs, uint32_t a, uint8_t width, uint64_t* carry) {
  uint64_t c = 0;
  const uint32_t d = adc(lhs, width, &c);
  const uint32_t r = adc((uint64_t)d * a, width, carry);
  *carry += a * c;
  return r;
}
static inline uint64_t sbc(uint64_t lhs, uint8_t width, uint32_t* carry) { /* arith.h:94-99 */
  const int borrow = (lhs < *carry);
  const uint64_t r = lhs - *carry + (borrow ? ((uint64_t)1 << width) : 0);
  *carry = borrow ? 1 : 0;
  return r;
}

/* ---- ibdwt.h:17-43 ---- */
size_t orc_transform_size(uint32_t exponent) {
  uint32_t w = 0, log2_n = 1, log2_n5 = 2;
  do {
    ++log2_n;
    w = exponent >> log2_n;
  } while ((w + 1) * 2 + log2_n >= 64);
  do {
    ++log2_n5;
    w = exponent / (5u << log2_n5);
  } while ((w + 1) * 2 + (log2_n5 + 2.4) >= 64);
  const size_t invalid = (size_t)-1;
  const size_t n2 = (log2_n <= 26) ? ((size_t)1 << log2_n) : invalid;
  const size_t n5 = (log2_n5 <= 26) ? ((size_t)5 << log2_n5) : invalid;
  return n2 < n5 ? n2 : n5;
}

struct orc_ctx {
  uint32_t p;
  size_t n, m, regs;
  size_t cwm;       /* carry work-group size in 4-digit groups (engine_gpu.h:110) */
  uint64_t inv_m;   /* INV_N_2, engine_gpu.h:1275 */
  uint8_t* width;   /* n */
  uint64_t* w;      /* n, natural order */
  uint64_t* wi;     /* n */
  uint64_t* wm;     /* m: wm[e] = omega_m^e */
  uint64_t* reg;    /* regs * n */
  uint64_t* carry;  /* n/4/cwm */
  uint64_t *t0, *t1; /* m each: even / odd planes */
};

static int ilog2z(size_t v) { int r = -1; while (v) { v >>= 1; ++r; } return r; }

orc_ctx* orc_create(uint32_t p, size_t reg_count) {
  if (p < 3 || reg_count == 0) return NULL;
  orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  const size_t n = orc_transform_size(p);
  c->p = p; c->n = n; c->m = n / 2; c->regs = reg_count;
  const size_t n5 = (n % 5 == 0) ? n / 5 : n;
  size_t g = n5 / 4; if (g > 256) g = 256; if (g < 1) g = 1;
  c->cwm = (size_t)1 << ilog2z(g);
  c->inv_m = MOD_P - (MOD_P - 1) / c->m;
  c->width = (uint8_t*)malloc(n);
  c->w = (uint64_t*)malloc(n * 8);
  c->wi = (uint64_t*)malloc(n * 8);
  c->wm = (uint64_t*)malloc(c->m * 8);
  c->reg = (uint64_t*)calloc(reg_count * n, 8);
  c->carry = (uint64_t*)calloc(n / 4 / c->cwm + 1, 8);
  c->t0 = (uint64_t*)malloc(c->m * 8);
  c->t1 = (uint64_t*)malloc(c->m * 8);

  /* ibdwt.h:111-147 (weights kept in natural order; the reference permutes storage only) */
  const uint64_t nr2 = mod_pow(554, (MOD_P - 1) / 192 / n);
  c->w[0] = 1; c->wi[0] = 1;
  uint32_t ceil_qjm1_n = 0;
  /* powers nr2^e: the reference calls mod_pow per digit; identical values, computed here through
     a two-level table to keep start-up short */
  const size_t sq = (size_t)1 << ((ilog2z(n) + 2) / 2);
  uint64_t* lo = (uint64_t*)malloc(sq * 8);
  uint64_t* hi = (uint64_t*)malloc((n / sq + 2) * 8);
  lo[0] = 1; for (size_t i = 1; i < sq; ++i) lo[i] = mod_mul(lo[i - 1], nr2);
  const uint64_t step = mod_mul(lo[sq - 1], nr2);
  hi[0] = 1; for (size_t i = 1; i < n / sq + 2; ++i) hi[i] = mod_mul(hi[i - 1], step);
  for (size_t j = 1; j <= n; ++j) {
    const uint64_t qj = (uint64_t)p * j;
    const uint32_t ceil_qj_n = (uint32_t)((qj - 1) / n + 1);
    c->width[j - 1] = (uint8_t)(ceil_qj_n - ceil_qjm1_n);
    if (j == n) break;
    const uint32_t r = (uint32_t)(qj % n);
    const size_t e = n - r;
    const uint64_t nr2r = (r != 0) ? mod_mul(hi[e / sq], lo[e % sq]) : 1;
    c->w[j] = nr2r;
    ceil_qjm1_n = ceil_qj_n;
  }
  free(lo); free(hi);
  /* batch inversion of the weights (same values as mod_invert per digit) */
  {
    uint64_t* pref = (uint64_t*)malloc(n * 8);
    uint64_t acc = 1;
    for (size_t j = 0; j < n; ++j) { pref[j] = acc; acc = mod_mul(acc, c->w[j]); }
    uint64_t inv = mod_invert(acc);
    for (size_t j = n; j-- > 0;) { c->wi[j] = mod_mul(inv, pref[j]); inv = mod_mul(inv, c->w[j]); }
    free(pref);
  }
  /* roots: omega_m = 7^((P-1)/m) (arith.h:72, ibdwt.h:86) */
  const uint64_t om = mod_root_nth(c->m);
  c->wm[0] = 1; for (size_t e = 1; e < c->m; ++e) c->wm[e] = mod_mul(c->wm[e - 1], om);
  return c;
}

void orc_destroy(orc_ctx* c) {
  if (!c) return;
  free(c->width); free(c->w); free(c->wi); free(c->wm); free(c->reg); free(c->carry); free(c->t0); free(c->t1);
  free(c);
}
size_t orc_size(const orc_ctx* c) { return c->n; }
uint32_t orc_exponent(const orc_ctx* c) { return c->p; }
void orc_widths(const orc_ctx* c, uint8_t* out) { memcpy(out, c->width, c->n); }
void orc_weights(const orc_ctx* c, uint64_t* w, uint64_t* winv) { memcpy(w, c->w, c->n * 8); memcpy(winv, c->wi, c->n * 8); }
int orc_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_threads(int t) {
#ifdef _OPENMP
  if (t > 0) omp_set_num_threads(t);
#else
  (void)t;
#endif
}

/* ---- length-m transform of one plane (m = {1,5} * 2^k).  Output order: block k1 (0..r5-1),
   position pos holds frequency k1 + r5 * bitrev(pos). ---- */
static size_t bitrev(size_t i, size_t len) {
  size_t r = 0;
  for (size_t k = len; k > 1; k >>= 1, i >>= 1) r = (r << 1) | (i & 1);
  return r;
}

static void fwd_plane(const orc_ctx* c, uint64_t* a) {
  const size_t m = c->m, r5 = (m % 5 == 0) ? 5 : 1, L = m / r5;
  const uint64_t* wm = c->wm;
  if (r5 == 5) {
    const size_t s5 = m / 5;
#pragma omp parallel for schedule(static) if (OMP_BIG)
    for (size_t i2 = 0; i2 < L; ++i2) {
      uint64_t x[5], y[5];
      for (int i1 = 0; i1 < 5; ++i1) x[i1] = a[L * i1 + i2];
      for (int k1 = 0; k1 < 5; ++k1) {
        uint64_t s = 0;
        for (int i1 = 0; i1 < 5; ++i1) s = mod_add(s, mod_mul(x[i1], wm[(s5 * ((size_t)i1 * k1 % 5)) % m]));
        y[k1] = mod_mul(s, wm[(i2 * (size_t)k1) % m]);
      }
      for (int k1 = 0; k1 < 5; ++k1) a[L * k1 + i2] = y[k1];
    }
  }
  /* radix-2 DIF on each block of length L with omega_L = omega_m^r5 */
  for (size_t len = L; len >= 2; len >>= 1) {
    const size_t half = len / 2, tstep = m / len; /* omega_len^t = wm[t * m/len] */
#pragma omp parallel for schedule(static) if (OMP_BIG)
    for (size_t b = 0; b < m / 2; ++b) {
      const size_t blk = b / half, t = b % half, i = blk * len + t;
      const uint64_t u = a[i], v = a[i + half];
      a[i] = mod_add(u, v);
      a[i + half] = mod_mul(mod_sub(u, v), wm[t * tstep]);
    }
  }
}

static void bck_plane(const orc_ctx* c, uint64_t* a) {
  const size_t m = c->m, r5 = (m % 5 == 0) ? 5 : 1, L = m / r5;
  const uint64_t* wm = c->wm;
  for (size_t len = 2; len <= L; len <<= 1) {
    const size_t half = len / 2, tstep = m / len;
#pragma omp parallel for schedule(static) if (OMP_BIG)
    for (size_t b = 0; b < m / 2; ++b) {
      const size_t blk = b / half, t = b % half, i = blk * len + t;
      const uint64_t u = a[i], v = mod_mul(a[i + half], wm[(m - t * tstep) % m]);
      a[i] = mod_add(u, v);
      a[i + half] = mod_sub(u, v);
    }
  }
  if (r5 == 5) {
    const size_t s5 = m / 5;
#pragma omp parallel for schedule(static) if (OMP_BIG)
    for (size_t i2 = 0; i2 < L; ++i2) {
      uint64_t x[5], y[5];
      for (int k1 = 0; k1 < 5; ++k1) x[k1] = mod_mul(a[L * k1 + i2], wm[(m - (i2 * (size_t)k1) % m) % m]);
      for (int i1 = 0; i1 < 5; ++i1) {
        uint64_t s = 0;
        for (int k1 = 0; k1 < 5; ++k1) s = mod_add(s, mod_mul(x[k1], wm[(m - (s5 * ((size_t)i1 * k1 % 5)) % m) % m]));
        y[i1] = s;
      }
      for (int i1 = 0; i1 < 5; ++i1) a[L * i1 + i2] = y[i1];
    }
  }
}

static inline size_t freq_of(const orc_ctx* c, size_t idx) {
  const size_t m = c->m, r5 = (m % 5 == 0) ? 5 : 1, L = m / r5;
  return idx / L + r5 * bitrev(idx % L, L);
}

static void split_planes(const orc_ctx* c, const uint64_t* x, uint64_t* e, uint64_t* o) {
#pragma omp parallel for schedule(static) if (OMP_BIG)
  for (size_t i = 0; i < c->m; ++i) { e[i] = x[2 * i]; o[i] = x[2 * i + 1]; }
}
static void join_planes(const orc_ctx* c, uint64_t* x, const uint64_t* e, const uint64_t* o) {
#pragma omp parallel for schedule(static) if (OMP_BIG)
  for (size_t i = 0; i < c->m; ++i) { x[2 * i] = e[i]; x[2 * i + 1] = o[i]; }
}

/* marin.cl:1696-1728 + 2198-2216: unweight, x a, carry (weak), re-weight */
static void carry_weight_mul(orc_ctx* c, uint64_t* x, uint32_t a) {
  const size_t n = c->n, groups = n / 4, cwm = c->cwm, wgs = groups / cwm;
  uint64_t* cl = (uint64_t*)malloc(groups * 8);
  uint64_t* u = (uint64_t*)malloc(n * 8);
  /* pass 1a: per 4-digit group, carry chain starting from 0 */
#pragma omp parallel for schedule(static) if (OMP_BIG)
  for (size_t g = 0; g < groups; ++g) {
    uint64_t cc = 0;
    for (int l = 0; l < 4; ++l) {
      const size_t k = 4 * g + l;
      const uint64_t v = mod_mul(mod_mul(x[k], c->inv_m), c->wi[k]);
      u[k] = adc_mul(v, a, c->width[k], &cc);
    }
    cl[g] = cc;
  }
  /* pass 1b: add the left neighbour's carry inside a work-group (adc4, marin.cl:203-212) */
#pragma omp parallel for schedule(static) if (OMP_BIG)
  for (size_t g = 0; g < groups; ++g) {
    const size_t lid = g % cwm;
    uint64_t cc = (lid == 0) ? 0 : cl[g - 1];
    for (int l = 0; l < 3; ++l) { const size_t k = 4 * g + l; u[k] = adc(u[k], c->width[k], &cc); }
    u[4 * g + 3] += cc;
    if (lid == cwm - 1) c->carry[(g != groups - 1) ? g / cwm + 1 : 0] = cl[g];
  }
  /* pass 2: work-group carry into the first group of each block */
  for (size_t wg = 0; wg < wgs; ++wg) {
    const size_t g = wg * cwm;
    uint64_t cc = c->carry[wg];
    for (int l = 0; l < 3; ++l) { const size_t k = 4 * g + l; u[k] = adc(u[k], c->width[k], &cc); }
    u[4 * g + 3] += cc;
  }
#pragma omp parallel for schedule(static) if (OMP_BIG)
  for (size_t k = 0; k < n; ++k) x[k] = mod_mul(u[k], c->w[k]);
  free(cl); free(u);
}

void orc_set_u32(orc_ctx* c, size_t dst, uint32_t a) { /* engine_gpu.h:1432-1450 */
  uint64_t* x = c->reg + dst * c->n;
  memset(x, 0, c->n * 8);
  x[0] = a;
}
void orc_copy(orc_ctx* c, size_t dst, size_t src) { if (dst != src) memcpy(c->reg + dst * c->n, c->reg + src * c->n, c->n * 8); }

void orc_square_mul(orc_ctx* c, size_t r, uint32_t a) { /* engine_gpu.h:1568-1630 */
  uint64_t* x = c->reg + r * c->n;
  uint64_t *e = c->t0, *o = c->t1;
  split_planes(c, x, e, o);
  fwd_plane(c, e); fwd_plane(c, o);
#pragma omp parallel for schedule(static) if (OMP_BIG)
  for (size_t i = 0; i < c->m; ++i) { /* sqr22, marin.cl:379-384 with rho = omega_m^k */
    const uint64_t rho = c->wm[freq_of(c, i)];
    const uint64_t u0 = e[i], u1 = o[i];
    e[i] = mod_add(mod_mul(u0, u0), mod_mul(mod_mul(u1, u1), rho));
    o[i] = mod_mul(u1, mod_add(u0, u0));
  }
  bck_plane(c, e); bck_plane(c, o);
  join_planes(c, x, e, o);
  carry_weight_mul(c, x, a);
}

void orc_set_multiplicand(orc_ctx* c, size_t dst, size_t src) { /* engine_gpu.h:1695-1756 */
  orc_copy(c, dst, src);
  uint64_t* x = c->reg + dst * c->n;
  uint64_t *e = c->t0, *o = c->t1;
  split_planes(c, x, e, o);
  fwd_plane(c, e); fwd_plane(c, o);
  join_planes(c, x, e, o);
}

void orc_mul(orc_ctx* c, size_t dst, size_t src, uint32_t a) { /* engine_gpu.h:1823-1884, mul22 marin.cl:387-392 */
  uint64_t* x = c->reg + dst * c->n;
  const uint64_t* y = c->reg + src * c->n;
  uint64_t *e = c->t0, *o = c->t1;
  split_planes(c, x, e, o);
  fwd_plane(c, e); fwd_plane(c, o);
#pragma omp parallel for schedule(static) if (OMP_BIG)
  for (size_t i = 0; i < c->m; ++i) {
    const uint64_t rho = c->wm[freq_of(c, i)];
    const uint64_t x0 = e[i], x1 = o[i], y0 = y[2 * i], y1 = y[2 * i + 1];
    e[i] = mod_add(mod_mul(x0, y0), mod_mul(mod_mul(x1, y1), rho));
    o[i] = mod_add(mod_mul(x0, y1), mod_mul(x1, y0));
  }
  bck_plane(c, e); bck_plane(c, o);
  join_planes(c, x, e, o);
  carry_weight_mul(c, x, a);
}

void orc_sub_u32(orc_ctx* c, size_t r, uint32_t a) { /* marin.cl:2376-2393 */
  uint64_t* x = c->reg + r * c->n;
  uint32_t cc = a;
  while (cc != 0) {
    for (size_t k = 0; k < c->n; ++k) {
      x[k] = mod_mul(sbc(mod_mul(x[k], c->wi[k]), c->width[k], &cc), c->w[k]);
      if (cc == 0) return;
    }
  }
}

/* shared tail of carry_weight_add_p1 / add_neg_p1 (marin.cl:2160-2193, 1856-1888) + p2 */
static void carry_weight_addlike(orc_ctx* c, uint64_t* y, const uint64_t* x, int negate) {
  const size_t n = c->n, groups = n / 4, cwm = c->cwm, wgs = groups / cwm;
  uint64_t* cl = (uint64_t*)malloc(groups * 8);
  uint64_t* u = (uint64_t*)malloc(n * 8);
  for (size_t g = 0; g < groups; ++g) {
    uint64_t cc = 0;
    for (int l = 0; l < 4; ++l) {
      const size_t k = 4 * g + l;
      const uint64_t uy = mod_mul(y[k], c->wi[k]);
      uint64_t vx = mod_mul(x[k], c->wi[k]);
      if (negate) vx = (((uint64_t)1 << c->width[k]) * 2 - 2) - vx; /* neg2_mp4, marin.cl:246-256 */
      cc += vx;                                                     /* addc4, marin.cl:271-281 */
      u[k] = adc(uy, c->width[k], &cc);
    }
    cl[g] = cc;
  }
  for (size_t g = 0; g < groups; ++g) {
    const size_t lid = g % cwm;
    uint64_t cc = (lid == 0) ? 0 : cl[g - 1];
    for (int l = 0; l < 3; ++l) { const size_t k = 4 * g + l; u[k] = adc(u[k], c->width[k], &cc); }
    u[4 * g + 3] += cc;
    if (lid == cwm - 1) c->carry[(g != groups - 1) ? g / cwm + 1 : 0] = cl[g];
  }
  for (size_t wg = 0; wg < wgs; ++wg) {
    const size_t g = wg * cwm;
    uint64_t cc = c->carry[wg];
    for (int l = 0; l < 3; ++l) { const size_t k = 4 * g + l; u[k] = adc(u[k], c->width[k], &cc); }
    u[4 * g + 3] += cc;
  }
  for (size_t k = 0; k < n; ++k) y[k] = mod_mul(u[k], c->w[k]);
  free(cl); free(u);
}
void orc_add(orc_ctx* c, size_t dst, size_t src) { carry_weight_addlike(c, c->reg + dst * c->n, c->reg + src * c->n, 0); }
void orc_sub_reg(orc_ctx* c, size_t dst, size_t src) { carry_weight_addlike(c, c->reg + dst * c->n, c->reg + src * c->n, 1); }

void orc_get_digits(const orc_ctx* c, size_t src, uint64_t* d) { /* engine_gpu.h:1534-1561 */
  const size_t n = c->n;
  const uint64_t* x = c->reg + src * n;
  uint64_t cc = 0;
  for (size_t k = 0; k < n; ++k) d[k] = adc(mod_mul(x[k], c->wi[k]), c->width[k], &cc);
  while (cc != 0) {
    for (size_t k = 0; k < n; ++k) {
      d[k] = adc(d[k], c->width[k], &cc);
      if (cc == 0) break;
    }
  }
  for (size_t k = 0; k < n; ++k) d[k] = (uint32_t)d[k] | ((uint64_t)c->width[k] << 32);
}
void orc_set_digits(orc_ctx* c, size_t dst, const uint64_t* d) { /* engine_gpu.h:1452-1485 */
  uint64_t* x = c->reg + dst * c->n;
  for (size_t k = 0; k < c->n; ++k) x[k] = mod_mul((uint32_t)d[k], c->w[k]);
}
void orc_get_raw(const orc_ctx* c, size_t src, uint64_t* x) { memcpy(x, c->reg + src * c->n, c->n * 8); }
void orc_set_raw(orc_ctx* c, size_t dst, const uint64_t* x) { memcpy(c->reg + dst * c->n, x, c->n * 8); }

/* ---- engine::digit, engine.h:257-295 ---- */
uint64_t orc_digits_res64(const uint64_t* d, size_t n) {
  uint64_t r64 = 0; uint8_t s = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint64_t u = (uint32_t)d[i];
    const uint8_t width = (uint8_t)(d[i] >> 32);
    r64 += u << s;
    s += width;
    if (s >= 64) break;
  }
  return r64;
}
int orc_digits_equal_to(const uint64_t* d, size_t n, uint64_t a) {
  uint64_t r = a;
  for (size_t i = 0; i < n; ++i) {
    const uint64_t u = (uint32_t)d[i];
    const uint8_t width = (uint8_t)(d[i] >> 32);
    if ((r & (((uint64_t)1 << width) - 1)) != u) return 0;
    r >>= width;
  }
  return 1;
}
int orc_digits_equal_to_Mp(const uint64_t* d, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const uint64_t u = (uint32_t)d[i];
    const uint8_t width = (uint8_t)(d[i] >> 32);
    if (u != ((uint64_t)1 << width) - 1) return 0;
  }
  return 1;
}

/* ---- AlgoUtils.hpp:165-223 ---- */
size_t orc_word_count(uint32_t p) { return ((size_t)p + 31) / 32; }
void orc_pack_words(const uint64_t* d, size_t n, uint32_t p, uint32_t* out) {
  const size_t total = orc_word_count(p);
  memset(out, 0, total * 4);
  uint64_t acc = 0; int acc_bits = 0; size_t o = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t w = (uint8_t)(d[i] >> 32);
    uint32_t v = (uint32_t)d[i];
    if (w < 32) v &= (uint32_t)(((uint64_t)1 << w) - 1);
    acc |= (uint64_t)v << acc_bits;
    acc_bits += (int)w;
    while (acc_bits >= 32 && o < total) { out[o++] = (uint32_t)acc; acc >>= 32; acc_bits -= 32; }
  }
  if (o < total) out[o++] = (uint32_t)acc;
}
static uint32_t mod3_words(const uint32_t* W, size_t count) {
  uint32_t r = 0; for (size_t i = 0; i < count; ++i) r = (r + (W[i] % 3)) % 3; return r;
}
static void div3_words(uint32_t E, uint32_t* W, size_t count) {
  uint32_t r = (3 - mod3_words(W, count)) % 3;
  const int topBits = (int)(E % 32);
  { uint64_t t = ((uint64_t)r << topBits) + W[count - 1]; W[count - 1] = (uint32_t)(t / 3); r = (uint32_t)(t % 3); }
  for (size_t i = count - 1; i-- > 0;) { uint64_t t = ((uint64_t)r << 32) + W[i]; W[i] = (uint32_t)(t / 3); r = (uint32_t)(t % 3); }
}
void orc_prp3_div9(uint32_t p, uint32_t* words, size_t count) { div3_words(p, words, count); div3_words(p, words, count); }
void orc_format_res64(const uint32_t* W, size_t count, char out[17]) {
  static const char hx[] = "0123456789ABCDEF";
  const uint64_t r64 = ((uint64_t)(count > 1 ? W[1] : 0) << 32) | (count ? W[0] : 0u);
  for (int i = 0; i < 16; ++i) out[i] = hx[(r64 >> (60 - 4 * i)) & 15];
  out[16] = 0;
}
void orc_format_res2048(const uint32_t* W, size_t count, char out[513]) {
  static const char hx[] = "0123456789abcdef";
  char* q = out;
  for (int i = 63; i >= 0; --i) {
    const uint32_t w = ((size_t)i < count) ? W[i] : 0u;
    for (int k = 0; k < 8; ++k) *q++ = hx[(w >> (28 - 4 * k)) & 15];
  }
  *q = 0;
}

void orc_get_words(const orc_ctx* c, size_t src, uint32_t* words, size_t count) { /* engine.h:173-203 */
  uint64_t* d = (uint64_t*)malloc(c->n * 8);
  orc_get_digits(c, src, d);
  memset(words, 0, count * 4);
  if (!orc_digits_equal_to_Mp(d, c->n)) {
    uint32_t* tmp = (uint32_t*)calloc(orc_word_count(c->p) + 1, 4);
    orc_pack_words(d, c->n, c->p, tmp);
    const size_t wc = orc_word_count(c->p);
    memcpy(words, tmp, (count < wc ? count : wc) * 4);
    free(tmp);
  }
  free(d);
}
void orc_set_words(orc_ctx* c, size_t dst, const uint32_t* words, size_t count) { /* engine.h:206-232 */
  const size_t n = c->n;
  uint64_t* d = (uint64_t*)malloc(n * 8);
  uint32_t* v = (uint32_t*)calloc(orc_word_count(c->p) + 2, 4);
  const size_t wc = orc_word_count(c->p);
  memcpy(v, words, (count < wc ? count : wc) * 4);
  size_t bit = 0;
  for (size_t k = 0; k < n; ++k) {
    const uint8_t width = c->width[k];
    const size_t i = bit / 32, s = bit % 32;
    uint32_t u = v[i] >> s;
    if (s != 0) u |= v[i + 1] << (32 - s);
    d[k] = (u & ((1u << width) - 1)) | ((uint64_t)width << 32);
    bit += width;
  }
  orc_set_digits(c, dst, d);
  free(v); free(d);
}
