/* ORACLE (second field family) -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's paired-NTT squaring x <- x^2 mod 2^p-1 over GF(M61^2) x GF(M31^2) with a prime-factor
 * (Good-Thomas) odd axis of radix 3 or 9 -- SURVEY.md 8f row N1: the "Aevum" backend, third_party/aevum/src/cl/fft-middle.cl:663-720
 * (pfaDft3 / radix-9 = 3 x 3 with scalar roots), pfaunpack.cl:12-56 (index map), Gpu.cpp:423-453 (host defines), and its
 * single-file CPU illustration docs/mersenne2_mixed_crt_2d_half_fast/mersenne2_mixed_crt_2d_half_fast.cpp (cited as "m2:" below).
 * What is restated, step by step (m2:1077-1085 square()):
 *   weight            digit j times 2^((n - qj mod n)/n): n-th roots of two are powers of two in Z/M61 and Z/M31
 *                     (2 has odd order 61 / 31), so a weight is a bit rotation (m2:194,347,1040-1046)
 *   odd axis          logical digit j <-> (a, b) = (j mod odd, j mod m), no twiddles between the axes (m2:733-758); DFT of length
 *                     3 or 9 along a with roots that are REAL scalars of Z/M61 and Z/M31 (9 | M31 - 1, 9 | M61 - 1; m2:95-134)
 *   power-of-two axis every row stays a real sequence (scalar roots), so a row of m reals is transformed as m/2 complex values
 *                     of GF(p^2) = Z/p[i] and untangled with the conjugate symmetry before the point-wise squaring
 *                     (m2:829-915; here in the textbook split-real form instead of the fused sqr_row butterflies)
 *   inverse, 1/odd, 1/m, unweight, then Garner: the value below M61*M31 ~ 2^92 with the two residues (m2:429-441), carry in
 *                     base 2^width with wrap-around (m2:954-1001)
 * Only tests/ may load this; the product library never links or falls back to it.
 *
 * Parity status: PINNED (tests/test_oracle_crt.py) by (a) Python big integers on every iteration of small exponents at radix
 * 1 / 3 / 9, (b) the reference's own prototype compiled where it lies (oracle/_ref/ref_mixed_crt: prime / composite verdicts of
 * complete LL tests at the same forced radices), (c) the libgmp pins of tests/golden/big_p_pins.json at p = 205271257 with
 * n = 9*2^20 and 3*2^21 words (BASELINE configs[3]; README.md:907-926) -- canonical residues do not depend on the transform.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef unsigned __int128 u128;

#define M61 ((uint64_t)0x1fffffffffffffffull)
#define M31 ((uint32_t)0x7fffffffu)

/* ---- Z/M61 ---- */
static inline uint64_t a61(uint64_t a, uint64_t b) { uint64_t s = a + b; s = (s & M61) + (s >> 61); return s >= M61 ? s - M61 : s; }
static inline uint64_t s61(uint64_t a, uint64_t b) { return a >= b ? a - b : a + M61 - b; }
static inline uint64_t m61(uint64_t a, uint64_t b) { u128 t = (u128)a * b; uint64_t lo = (uint64_t)t & M61, hi = (uint64_t)(t >> 61); return a61(lo, hi >= M61 ? hi - M61 : hi); }
static inline uint64_t rot61(uint64_t a, unsigned s) { s %= 61; if (!s) return a; uint64_t r = ((a << s) & M61) | (a >> (61 - s)); return r >= M61 ? r - M61 : r; }
static uint64_t pow61(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = m61(r, a); a = m61(a, a); e >>= 1; } return r; }
/* ---- Z/M31 ---- */
static inline uint32_t a31(uint32_t a, uint32_t b) { uint32_t s = a + b; s = (s & M31) + (s >> 31); return s >= M31 ? s - M31 : s; }
static inline uint32_t s31(uint32_t a, uint32_t b) { return a >= b ? a - b : a + M31 - b; }
static inline uint32_t m31(uint32_t a, uint32_t b) { uint64_t t = (uint64_t)a * b; uint32_t lo = (uint32_t)t & M31, hi = (uint32_t)(t >> 31); return a31(lo, hi >= M31 ? hi - M31 : hi); }
static inline uint32_t rot31(uint32_t a, unsigned s) { s %= 31; if (!s) return a; uint32_t r = ((a << s) & M31) | (a >> (31 - s)); return r >= M31 ? r - M31 : r; }
static uint32_t pow31(uint32_t a, uint64_t e) { uint32_t r = 1; while (e) { if (e & 1) r = m31(r, a); a = m31(a, a); e >>= 1; } return r; }

/* ---- GF(p^2) = Z/p[i], i^2 = -1 (p = 3 mod 4) ---- */
typedef struct { uint64_t re, im; } c61;
typedef struct { uint32_t re, im; } c31;
static inline c61 cadd61(c61 a, c61 b) { c61 r = {a61(a.re, b.re), a61(a.im, b.im)}; return r; }
static inline c61 csub61(c61 a, c61 b) { c61 r = {s61(a.re, b.re), s61(a.im, b.im)}; return r; }
static inline c61 cmul61(c61 a, c61 b) { c61 r = {s61(m61(a.re, b.re), m61(a.im, b.im)), a61(m61(a.re, b.im), m61(a.im, b.re))}; return r; }
static inline c61 conj61(c61 a) { c61 r = {a.re, a.im ? M61 - a.im : 0}; return r; }
static inline c31 cadd31(c31 a, c31 b) { c31 r = {a31(a.re, b.re), a31(a.im, b.im)}; return r; }
static inline c31 csub31(c31 a, c31 b) { c31 r = {s31(a.re, b.re), s31(a.im, b.im)}; return r; }
static inline c31 cmul31(c31 a, c31 b) { c31 r = {s31(m31(a.re, b.re), m31(a.im, b.im)), a31(m31(a.re, b.im), m31(a.im, b.re))}; return r; }
static inline c31 conj31(c31 a) { c31 r = {a.re, a.im ? M31 - a.im : 0}; return r; }
static c61 cpow61(c61 a, u128 e) { c61 r = {1, 0}; while (e) { if (e & 1) r = cmul61(r, a); a = cmul61(a, a); e >>= 1; } return r; }
static c31 cpow31(c31 a, uint64_t e) { c31 r = {1, 0}; while (e) { if (e & 1) r = cmul31(r, a); a = cmul31(a, a); e >>= 1; } return r; }

/* a primitive 2^k-th root of unity of norm 1 (k <= 60 resp. 30): the group GF(p^2)* has order (p - 1)(p + 1) with p + 1 = 2^61
   resp. 2^31; g^((p-1) * 2^(61-k)) for the first g whose power has exact order 2^k (found by search, not a copied constant) */
static c61 root2k_61(unsigned k) {
  for (uint64_t t = 2;; ++t) {
    c61 g = {t, 1};
    c61 h = cpow61(g, (u128)(M61 - 1) << (61 - k));
    c61 z = h;
    for (unsigned i = 1; i < k; ++i) z = cmul61(z, z);   /* h^(2^(k-1)) must be -1 */
    if (z.re == M61 - 1 && z.im == 0) return h;
  }
}
static c31 root2k_31(unsigned k) {
  for (uint32_t t = 2;; ++t) {
    c31 g = {t, 1};
    c31 h = cpow31(g, (uint64_t)(M31 - 1) << (31 - k));
    c31 z = h;
    for (unsigned i = 1; i < k; ++i) z = cmul31(z, z);
    if (z.re == M31 - 1 && z.im == 0) return h;
  }
}
/* a primitive odd-th root of unity that is a scalar (odd | p - 1) */
static uint64_t oddroot61(unsigned odd) {
  for (uint64_t g = 2;; ++g) {
    uint64_t r = pow61(g, (M61 - 1) / odd);
    int ok = r != 1;
    for (unsigned d = 2; ok && d < odd; ++d) if (odd % d == 0 && pow61(r, odd / d) == 1) ok = 0;
    if (ok && pow61(r, odd) == 1) return r;
  }
}
static uint32_t oddroot31(unsigned odd) {
  for (uint32_t g = 2;; ++g) {
    uint32_t r = pow31(g, (M31 - 1) / odd);
    int ok = r != 1;
    for (unsigned d = 2; ok && d < odd; ++d) if (odd % d == 0 && pow31(r, odd / d) == 1) ok = 0;
    if (ok && pow31(r, odd) == 1) return r;
  }
}
static uint64_t inv_small(uint64_t a, uint64_t m) { for (uint64_t x = 1; x < m; ++x) if (a * x % m == 1) return x; return 0; }

typedef struct orcc_ctx {
  uint32_t p;
  size_t n, odd, m, h;          /* n = odd * m words, m = 2^ln, h = m / 2 complex values per row */
  unsigned ln;
  uint8_t* width;               /* [n] */
  uint8_t *w61, *w31;           /* [n] weight exponents: digit j times 2^w (m2:1040-1046) */
  size_t* j_of;                 /* [odd * m]: logical digit of coordinate (a, b) (m2:733-758) */
  c61 *t61, *t61i; c31 *t31, *t31i;   /* omega_h^k (k < h) and omega_m^k (k <= h) tables: [h] and [h + 1] */
  c61 *u61; c31 *u31;           /* omega_m^k, k <= h */
  uint64_t r61[9], r61i[9]; uint32_t r31[9], r31i[9];   /* odd root powers */
  uint64_t* x;                  /* [n] the residue: unweighted digits in logical order */
  uint64_t *pre61; uint32_t* pre31;   /* [n] last squaring: residues of the unweighted coefficients before Garner / carry, logical order */
  uint64_t *wtd61; uint32_t* wtd31;   /* [n] the same still weighted (what the fused unweight + Garner + carry kernel of the GPU path reads) */
} orcc_ctx;

/* m2:479-503 size rule for a forced odd radix: smallest 2^ln with log2(n) + 2 (q/n + 1) < 92 */
size_t orcc_transform_size(uint32_t p, unsigned odd) {
  for (unsigned ln = 2; ln <= 30; ++ln) {
    const size_t n = (size_t)odd << ln;
    if (n > p) break;
    const long double log2n = (long double)ln + log2l((long double)odd);
    if (log2n + 2.0L * ((long double)p / (long double)n + 1.0L) < 92.0L) return n;
  }
  return 0;
}

static size_t bitrev_z(size_t i, unsigned bits) { size_t r = 0; for (unsigned k = 0; k < bits; ++k) { r = (r << 1) | (i & 1); i >>= 1; } return r; }

orcc_ctx* orcc_create(uint32_t p, unsigned odd, size_t n_forced) {
  if (odd != 1 && odd != 3 && odd != 9) return NULL;
  const size_t n = n_forced ? n_forced : orcc_transform_size(p, odd);
  if (!n || n % odd) return NULL;
  size_t m = n / odd; unsigned ln = 0;
  while (((size_t)1 << ln) < m) ++ln;
  if (((size_t)1 << ln) != m || ln < 2) return NULL;
  orcc_ctx* c = (orcc_ctx*)calloc(1, sizeof *c);
  c->p = p; c->n = n; c->odd = odd; c->m = m; c->h = m / 2; c->ln = ln;
  c->width = (uint8_t*)malloc(n); c->w61 = (uint8_t*)malloc(n); c->w31 = (uint8_t*)malloc(n);
  c->j_of = (size_t*)malloc(n * sizeof(size_t));
  c->x = (uint64_t*)calloc(n, 8); c->pre61 = (uint64_t*)calloc(n, 8); c->pre31 = (uint32_t*)calloc(n, 4);
  c->wtd61 = (uint64_t*)calloc(n, 8); c->wtd31 = (uint32_t*)calloc(n, 4);
  /* widths ceil(p(j+1)/n) - ceil(pj/n) and weight exponents: 2^(1/n) = 2^(n^-1 mod 61) in Z/M61 (2 has order 61), likewise mod 31 */
  const uint64_t l61 = inv_small(n % 61, 61), l31 = inv_small(n % 31, 31);
  uint64_t prev = 0;
  for (size_t j = 0; j < n; ++j) {
    const uint64_t next = ((uint64_t)p * (j + 1) + n - 1) / n;
    c->width[j] = (uint8_t)(next - prev); prev = next;
    const uint64_t r = (uint64_t)p * j % n, e = r ? n - r : 0;
    c->w61[j] = (uint8_t)(l61 * (e % 61) % 61); c->w31[j] = (uint8_t)(l31 * (e % 31) % 31);
  }
  /* Good-Thomas: coordinate (a, b) holds the digit j with j = a (mod odd), j = b (mod m) */
  const size_t minv = odd > 1 ? inv_small(m % odd, odd) : 0;
  for (size_t a = 0; a < odd; ++a)
    for (size_t b = 0; b < m; ++b) {
      const size_t t = odd > 1 ? ((a + odd - b % odd) % odd) * minv % odd : 0;
      c->j_of[a * m + b] = b + m * t;
    }
  /* tables: omega_h^k for the half-length complex transform, omega_m^k for the split-real step */
  const size_t h = c->h;
  c->t61 = (c61*)malloc(h * sizeof(c61)); c->t61i = (c61*)malloc(h * sizeof(c61));
  c->t31 = (c31*)malloc(h * sizeof(c31)); c->t31i = (c31*)malloc(h * sizeof(c31));
  c->u61 = (c61*)malloc((h + 1) * sizeof(c61)); c->u31 = (c31*)malloc((h + 1) * sizeof(c31));
  const c61 wm61 = root2k_61(ln); const c31 wm31 = root2k_31(ln);
  c61 z61 = {1, 0}; c31 z31 = {1, 0};
  for (size_t k = 0; k <= h; ++k) { c->u61[k] = z61; c->u31[k] = z31; z61 = cmul61(z61, wm61); z31 = cmul31(z31, wm31); }
  for (size_t k = 0; k < h; ++k) {   /* omega_h^k = omega_m^(2k); beyond k = h / 2 through omega_m^h = -1 */
    if (2 * k <= h) { c->t61[k] = c->u61[2 * k]; c->t31[k] = c->u31[2 * k]; }
    else { c->t61[k] = cmul61(c->u61[h], c->u61[2 * k - h]); c->t31[k] = cmul31(c->u31[h], c->u31[2 * k - h]); }
    c->t61i[k] = conj61(c->t61[k]); c->t31i[k] = conj31(c->t31[k]);   /* norm 1: the inverse is the conjugate */
  }
  if (odd > 1) {
    const uint64_t r = oddroot61(odd); const uint32_t s = oddroot31(odd);
    for (unsigned k = 0; k < odd; ++k) {
      c->r61[k] = pow61(r, k); c->r61i[k] = pow61(r, (odd - k) % odd);
      c->r31[k] = pow31(s, k); c->r31i[k] = pow31(s, (odd - k) % odd);
    }
  }
  return c;
}

void orcc_destroy(orcc_ctx* c) {
  if (!c) return;
  free(c->width); free(c->w61); free(c->w31); free(c->j_of); free(c->t61); free(c->t61i); free(c->t31); free(c->t31i);
  free(c->u61); free(c->u31); free(c->x); free(c->pre61); free(c->pre31); free(c->wtd61); free(c->wtd31); free(c);
}
size_t orcc_size(const orcc_ctx* c) { return c->n; }
void orcc_widths(const orcc_ctx* c, uint8_t* out) { memcpy(out, c->width, c->n); }

/* in-place radix-2 transform of h complex values, natural order in and out (bit reversal first) */
#define DEF_FFT(NAME, T, MUL, ADD, SUB)                                                                   \
  static void NAME(T* z, size_t h, unsigned bits, const T* tw) {                                         \
    for (size_t i = 0; i < h; ++i) { const size_t r = bitrev_z(i, bits); if (r > i) { T t = z[i]; z[i] = z[r]; z[r] = t; } } \
    for (size_t len = 2; len <= h; len <<= 1) {                                                           \
      const size_t half = len / 2, step = h / len;                                                        \
      for (size_t s = 0; s < h; s += len)                                                                 \
        for (size_t k = 0; k < half; ++k) { const T u = z[s + k], v = MUL(z[s + k + half], tw[k * step]); z[s + k] = ADD(u, v); z[s + k + half] = SUB(u, v); } \
    }                                                                                                     \
  }
DEF_FFT(fft61, c61, cmul61, cadd61, csub61)
DEF_FFT(fft31, c31, cmul31, cadd31, csub31)

/* one row: m reals (re / im of h complex values) -> cyclic self-convolution of length m, unnormalised by h */
static void row_square61(const orcc_ctx* c, c61* z) {
  const size_t h = c->h; const unsigned bits = c->ln - 1;
  fft61(z, h, bits, c->t61);
  const uint64_t half = (M61 + 1) / 2;
  c61* X = (c61*)malloc((h + 1) * sizeof(c61));
  for (size_t k = 0; k <= h; ++k) {   /* split-real: X_k = E_k + omega_m^k O_k, E = (Z_k + conj Z_-k)/2, O = (Z_k - conj Z_-k)/(2i) */
    const c61 zk = z[k % h], zc = conj61(z[(h - k) % h]);
    const c61 e = cadd61(zk, zc), d = csub61(zk, zc);
    const c61 o = {d.im, d.re ? M61 - d.re : 0};   /* d / i */
    c61 x = cadd61(e, cmul61(c->u61[k], o));
    x.re = m61(x.re, half); x.im = m61(x.im, half);
    X[k] = cmul61(x, x);
  }
  for (size_t k = 0; k < h; ++k) {    /* back to the packed form: Z'_k = E'_k + i O'_k with E' = (X_k + conj X_{h-k})/2, O' = (X_k - conj X_{h-k}) conj(omega_m^k)/2 */
    const c61 xk = X[k], xc = conj61(X[h - k]);
    const c61 e = cadd61(xk, xc), d = cmul61(csub61(xk, xc), conj61(c->u61[k]));
    const c61 io = {d.im ? M61 - d.im : 0, d.re};   /* i * d */
    c61 r = cadd61(e, io);
    r.re = m61(r.re, half); r.im = m61(r.im, half);
    z[k] = r;
  }
  free(X);
  fft61(z, h, bits, c->t61i);
}
static void row_square31(const orcc_ctx* c, c31* z) {
  const size_t h = c->h; const unsigned bits = c->ln - 1;
  fft31(z, h, bits, c->t31);
  const uint32_t half = (M31 + 1) / 2;
  c31* X = (c31*)malloc((h + 1) * sizeof(c31));
  for (size_t k = 0; k <= h; ++k) {
    const c31 zk = z[k % h], zc = conj31(z[(h - k) % h]);
    const c31 e = cadd31(zk, zc), d = csub31(zk, zc);
    const c31 o = {d.im, d.re ? M31 - d.re : 0};
    c31 x = cadd31(e, cmul31(c->u31[k], o));
    x.re = m31(x.re, half); x.im = m31(x.im, half);
    X[k] = cmul31(x, x);
  }
  for (size_t k = 0; k < h; ++k) {
    const c31 xk = X[k], xc = conj31(X[h - k]);
    const c31 e = cadd31(xk, xc), d = cmul31(csub31(xk, xc), conj31(c->u31[k]));
    const c31 io = {d.im ? M31 - d.im : 0, d.re};
    c31 r = cadd31(e, io);
    r.re = m31(r.re, half); r.im = m31(r.im, half);
    z[k] = r;
  }
  free(X);
  fft31(z, h, bits, c->t31i);
}

/* x <- x^2 * a mod 2^p - 1 */
void orcc_square_mul(orcc_ctx* c, uint32_t a) {
  const size_t n = c->n, odd = c->odd, m = c->m, h = c->h;
  c61* z61 = (c61*)malloc(odd * h * sizeof(c61));
  c31* z31 = (c31*)malloc(odd * h * sizeof(c31));
  /* weight into the (a, b) grid: value b of row a is the real or imaginary part of complex slot b / 2 */
#pragma omp parallel for schedule(static)
  for (size_t cb = 0; cb < odd * m; ++cb) {
    const size_t ra = cb / m, b = cb % m, j = c->j_of[cb];
    const uint64_t v61 = rot61(c->x[j] % M61, c->w61[j]); const uint32_t v31 = rot31((uint32_t)(c->x[j] % M31), c->w31[j]);
    if (b & 1) { z61[ra * h + b / 2].im = v61; z31[ra * h + b / 2].im = v31; } else { z61[ra * h + b / 2].re = v61; z31[ra * h + b / 2].re = v31; }
  }
  /* odd axis: DFT along a with scalar roots (direct sums; radix 9 = the same sums, the reference factors them 3 x 3) */
  if (odd > 1) {
#pragma omp parallel for schedule(static)
    for (size_t k = 0; k < h; ++k) {
      c61 in61[9], o61[9]; c31 in31[9], o31[9];
      for (size_t ra = 0; ra < odd; ++ra) { in61[ra] = z61[ra * h + k]; in31[ra] = z31[ra * h + k]; }
      for (size_t ka = 0; ka < odd; ++ka) {
        c61 s = {0, 0}; c31 t = {0, 0};
        for (size_t ra = 0; ra < odd; ++ra) {
          const uint64_t r = c->r61[ra * ka % odd]; const uint32_t q = c->r31[ra * ka % odd];
          s.re = a61(s.re, m61(in61[ra].re, r)); s.im = a61(s.im, m61(in61[ra].im, r));
          t.re = a31(t.re, m31(in31[ra].re, q)); t.im = a31(t.im, m31(in31[ra].im, q));
        }
        o61[ka] = s; o31[ka] = t;
      }
      for (size_t ra = 0; ra < odd; ++ra) { z61[ra * h + k] = o61[ra]; z31[ra * h + k] = o31[ra]; }
    }
  }
#pragma omp parallel for schedule(dynamic)
  for (size_t r = 0; r < 2 * odd; ++r) { if (r < odd) row_square61(c, z61 + r * h); else row_square31(c, z31 + (r - odd) * h); }
  if (odd > 1) {
#pragma omp parallel for schedule(static)
    for (size_t k = 0; k < h; ++k) {
      c61 in61[9], o61[9]; c31 in31[9], o31[9];
      for (size_t ra = 0; ra < odd; ++ra) { in61[ra] = z61[ra * h + k]; in31[ra] = z31[ra * h + k]; }
      for (size_t ka = 0; ka < odd; ++ka) {
        c61 s = {0, 0}; c31 t = {0, 0};
        for (size_t ra = 0; ra < odd; ++ra) {
          const uint64_t r = c->r61i[ra * ka % odd]; const uint32_t q = c->r31i[ra * ka % odd];
          s.re = a61(s.re, m61(in61[ra].re, r)); s.im = a61(s.im, m61(in61[ra].im, r));
          t.re = a31(t.re, m31(in31[ra].re, q)); t.im = a31(t.im, m31(in31[ra].im, q));
        }
        o61[ka] = s; o31[ka] = t;
      }
      for (size_t ra = 0; ra < odd; ++ra) { z61[ra * h + k] = o61[ra]; z31[ra * h + k] = o31[ra]; }
    }
  }
  /* 1 / (odd * h) [the half-length transforms are unnormalised by h], unweight: residues of every coefficient, logical order */
  const uint64_t s61v = pow61((uint64_t)(odd * h) % M61, M61 - 2); const uint32_t s31v = pow31((uint32_t)((odd * h) % M31), M31 - 2);
#pragma omp parallel for schedule(static)
  for (size_t cb = 0; cb < odd * m; ++cb) {
    const size_t ra = cb / m, b = cb % m, j = c->j_of[cb];
    const uint64_t v61 = (b & 1) ? z61[ra * h + b / 2].im : z61[ra * h + b / 2].re;
    const uint32_t v31 = (b & 1) ? z31[ra * h + b / 2].im : z31[ra * h + b / 2].re;
    c->wtd61[j] = m61(v61, s61v); c->wtd31[j] = m31(v31, s31v);
    c->pre61[j] = rot61(c->wtd61[j], 61 - c->w61[j] % 61);
    c->pre31[j] = rot31(c->wtd31[j], 31 - c->w31[j] % 31);
  }
  free(z61); free(z31);
  /* Garner: v = r31 + M31 * ((r61 - r31) / M31 mod M61) < M61 * M31 (m2:429-441); then * a and the carry with wrap-around */
  const uint64_t inv31 = pow61(M31, M61 - 2);
  u128 carry = 0;
  for (int lap = 0; lap < 4; ++lap) {
    for (size_t j = 0; j < n; ++j) {
      u128 v;
      if (lap == 0) {
        const uint64_t r31 = c->pre31[j], t = m61(s61(c->pre61[j], r31 % M61), inv31);
        v = ((u128)t * M31 + r31) * a + carry;
      } else v = (u128)c->x[j] + carry;
      c->x[j] = (uint64_t)(v & (((uint64_t)1 << c->width[j]) - 1));
      carry = v >> c->width[j];
      if (lap && !carry) break;
    }
    if (!carry) break;
  }
}

void orcc_set_u32(orcc_ctx* c, uint32_t a) {
  memset(c->x, 0, c->n * 8);
  uint64_t v = a;
  for (size_t j = 0; j < c->n && v; ++j) { c->x[j] = v & (((uint64_t)1 << c->width[j]) - 1); v >>= c->width[j]; }
}
void orcc_sub_u32(orcc_ctx* c, uint32_t a) {   /* m2:1095-1111 */
  uint64_t borrow = a;
  for (int lap = 0; lap < 3 && borrow; ++lap)
    for (size_t j = 0; j < c->n && borrow; ++j) {
      const uint64_t base = (uint64_t)1 << c->width[j];
      if (c->x[j] >= borrow) { c->x[j] -= borrow; borrow = 0; }
      else { const uint64_t need = borrow - c->x[j], k = (need + base - 1) >> c->width[j]; c->x[j] = c->x[j] + k * base - borrow; borrow = k; }
    }
}
/* digits: plain values in logical order (widths reach 39 bits here, so they do not fit the value | width << 32 encoding of
   engine::get); canonical, the value 2^p - 1 stays all ones */
void orcc_get_digits(const orcc_ctx* c, uint64_t* d) { memcpy(d, c->x, c->n * 8); }
void orcc_set_digits(orcc_ctx* c, const uint64_t* d) { memcpy(c->x, d, c->n * 8); }
/* canonical little-endian 32-bit words of the residue, 2^p - 1 -> 0 (what the plugin ABI exchanges: EngineApi.cpp:210-218) */
void orcc_get_words(const orcc_ctx* c, uint32_t* w, size_t count) {
  memset(w, 0, count * 4);
  int all_ones = 1;
  for (size_t j = 0; j < c->n && all_ones; ++j) all_ones = c->x[j] == (((uint64_t)1 << c->width[j]) - 1);
  if (all_ones) return;
  size_t bit = 0;
  for (size_t j = 0; j < c->n; ++j) {
    const size_t i = bit / 32, sh = bit % 32;
    const u128 v = (u128)c->x[j] << sh;
    for (unsigned k = 0; k < 3 && i + k < count; ++k) w[i + k] |= (uint32_t)(v >> (32 * k));
    bit += c->width[j];
  }
}
/* the two residues of every convolution coefficient of the last squaring, before Garner and the carry (logical order) */
void orcc_get_precarry(const orcc_ctx* c, uint64_t* r61, uint32_t* r31) { memcpy(r61, c->pre61, c->n * 8); memcpy(r31, c->pre31, c->n * 4); }
/* ... and before the unweighting (scaled by 1 / (odd * h) already) */
void orcc_get_weighted(const orcc_ctx* c, uint64_t* r61, uint32_t* r31) { memcpy(r61, c->wtd61, c->n * 8); memcpy(r31, c->wtd31, c->n * 4); }
