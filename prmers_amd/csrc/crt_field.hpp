// Z/M61, Z/M31 and their quadratic extensions Z/p[i] (i^2 = -1) for the paired-NTT squaring over
// GF(M61^2) x GF(M31^2) -- SURVEY.md 8f row N1 (reference: third_party/aevum/src/cl/math.cl, fft-middle.cl:663-720; CPU form
// docs/mersenne2_mixed_crt_2d_half_fast/mersenne2_mixed_crt_2d_half_fast.cpp:95-441).  Host + gfx950 device.
// All values canonical ([0, p)).  2 has order 61 resp. 31, so n-th roots of two are powers of two and an IBDWT weight is a
// bit rotation; the 2-power roots of unity live in the norm-1 subgroup of Z/p[i] (order p + 1 = 2^61 resp. 2^31).
#pragma once
#include <stdint.h>

#include "gf.hpp"

namespace mi355 {
namespace crt {

constexpr uint64_t M61 = 0x1fffffffffffffffull;
constexpr uint32_t M31 = 0x7fffffffu;

GF_HD uint64_t red61(uint64_t x) { x = (x & M61) + (x >> 61); return x >= M61 ? x - M61 : x; }
GF_HD uint32_t red31(uint64_t x) {   // any 64-bit value
  x = (x & M31) + (x >> 31);         // < 2^34
  uint32_t y = uint32_t(x & M31) + uint32_t(x >> 31);
  return y >= M31 ? y - M31 : y;
}
GF_HD uint64_t rot61(uint64_t a, uint32_t s) {   // a * 2^s mod M61, a < M61, s < 61
  if (s == 0) return a;
  const uint64_t r = ((a << s) & M61) | (a >> (61 - s));
  return r >= M61 ? r - M61 : r;
}
GF_HD uint32_t rot31(uint32_t a, uint32_t s) {
  if (s == 0) return a;
  const uint32_t r = ((a << s) & M31) | (a >> (31 - s));
  return r >= M31 ? r - M31 : r;
}
GF_HD uint64_t mul61(uint64_t a, uint64_t b) {
  uint64_t lo, hi;
  gf::mul64x64(a, b, lo, hi);                                    // < 2^122
  return red61((lo & M61) + ((lo >> 61) | (hi << 3)));           // hi 2^64 + lo = (hi 2^3 + lo >> 61) 2^61 + (lo mod 2^61)
}
GF_HD uint32_t mul31(uint32_t a, uint32_t b) {
  const uint64_t t = uint64_t(a) * b;                            // < 2^62
  const uint32_t s = uint32_t(t & M31) + uint32_t(t >> 31);      // < 2^32
  const uint32_t r = (s & M31) + (s >> 31);
  return r >= M31 ? r - M31 : r;
}

struct F61 {
  using S = uint64_t;
  struct alignas(16) C { uint64_t re, im; };
  static constexpr uint64_t M = M61;
  static GF_HD S add(S a, S b) { const S s = a + b; return s >= M ? s - M : s; }
  static GF_HD S sub(S a, S b) { return a >= b ? a - b : a + M - b; }
  static GF_HD S neg(S a) { return a ? M - a : 0; }
  static GF_HD S mul(S a, S b) { return mul61(a, b); }
  static GF_HD S half(S a) { return (a & 1) ? (a + M) >> 1 : a >> 1; }
};
struct F31 {
  using S = uint32_t;
  struct alignas(8) C { uint32_t re, im; };
  static constexpr uint32_t M = M31;
  static GF_HD S add(S a, S b) { const S s = a + b; return s >= M ? s - M : s; }
  static GF_HD S sub(S a, S b) { return a >= b ? a - b : a + M - b; }
  static GF_HD S neg(S a) { return a ? M - a : 0; }
  static GF_HD S mul(S a, S b) { return mul31(a, b); }
  static GF_HD S half(S a) { return (a & 1) ? (a + M) >> 1 : a >> 1; }
};

template <class F> GF_HD typename F::C cadd(typename F::C a, typename F::C b) { return {F::add(a.re, b.re), F::add(a.im, b.im)}; }
template <class F> GF_HD typename F::C csub(typename F::C a, typename F::C b) { return {F::sub(a.re, b.re), F::sub(a.im, b.im)}; }
template <class F> GF_HD typename F::C cneg(typename F::C a) { return {F::neg(a.re), F::neg(a.im)}; }
template <class F> GF_HD typename F::C cconj(typename F::C a) { return {a.re, F::neg(a.im)}; }
template <class F> GF_HD typename F::C cmul(typename F::C a, typename F::C b) {
  return {F::sub(F::mul(a.re, b.re), F::mul(a.im, b.im)), F::add(F::mul(a.re, b.im), F::mul(a.im, b.re))};
}
template <class F> GF_HD typename F::C csqr(typename F::C a) {   // (re + im)(re - im), 2 re im
  const typename F::S t = F::mul(a.re, a.im);
  return {F::mul(F::add(a.re, a.im), F::sub(a.re, a.im)), F::add(t, t)};
}
template <class F> GF_HD typename F::C chalf(typename F::C a) { return {F::half(a.re), F::half(a.im)}; }
template <class F> GF_HD typename F::C cscale(typename F::C a, typename F::S s) { return {F::mul(a.re, s), F::mul(a.im, s)}; }
template <class F> GF_HD typename F::C cmul_i(typename F::C a) { return {F::neg(a.im), a.re}; }      // a * i
template <class F> GF_HD typename F::C cdiv_i(typename F::C a) { return {a.im, F::neg(a.re)}; }      // a / i

// n = odd << ln; p = q n + t; l61 = n^-1 mod 61, l31 = n^-1 mod 31 (2^(1/n) = 2^l61 in Z/M61); lt61 = l61 t mod 61, lt31 likewise
struct Geom { uint32_t p, n, odd, ln, l61, l31, q, t, lt61, lt31; uint64_t inv31; uint32_t a; };

constexpr int kRun = 8;   // digits per thread of the carry sweep

// Per digit j the kernels need s_j = p j mod n: the width is q + [s + t > 0] + [s + t > n] - [s > 0] (the difference of two
// ceilings, plan.hpp width_of_s) and the weight exponent is l (n - s) mod 61 = 1 - l s mod 61 (l n = 1), both kept incrementally:
// s advances by t, l s by l t, and a wrap of s takes n resp. 1 off -- no division after the run's first digit.
struct DigitWalk {
  uint32_t s, A61, A31;   // p j mod n, l61 s mod 61, l31 s mod 31
  GF_HD void start(const Geom& g, uint32_t j) {
    s = uint32_t((uint64_t(g.p) * j) % g.n);
    A61 = uint32_t((uint64_t(g.l61) * (s % 61)) % 61); A31 = uint32_t((uint64_t(g.l31) * (s % 31)) % 31);
  }
  GF_HD uint32_t width(const Geom& g) const {
    const uint64_t st = uint64_t(s) + g.t;
    return g.q + (st > 0 ? 1u : 0u) + (st > g.n ? 1u : 0u) - (s > 0 ? 1u : 0u);
  }
  GF_HD uint32_t weight61() const { return s ? (62 - A61) % 61 : 0; }     // l (n - s) = 1 - A (mod 61)
  GF_HD uint32_t weight31() const { return s ? (32 - A31) % 31 : 0; }
  GF_HD uint32_t unweight61() const { return s ? (A61 + 60) % 61 : 0; }   // 61 - (1 - A) mod 61 = (A - 1) mod 61
  GF_HD uint32_t unweight31() const { return s ? (A31 + 30) % 31 : 0; }
  GF_HD void next(const Geom& g) {
    uint64_t sn = uint64_t(s) + g.t;
    A61 += g.lt61; A31 += g.lt31;
    if (sn >= g.n) { sn -= g.n; A61 += 60; A31 += 30; }   // l n = 1 (mod 61 / 31)
    s = uint32_t(sn);
    A61 = A61 >= 122 ? A61 - 122 : (A61 >= 61 ? A61 - 61 : A61);
    A31 = A31 >= 62 ? A31 - 62 : (A31 >= 31 ? A31 - 31 : A31);
  }
};

Geom make_geom(uint32_t p, size_t n, uint32_t odd, uint32_t a);   // crt_carry.hip (host)
// the fused unweight + Garner + carry sweep on device buffers (crt_carry.hip): digits[n], carry[2 * runs], residual[runs]
void crt_carry_launch(const Geom& g, const uint64_t* in61, const uint32_t* in31, uint64_t* digits, uint64_t* carry, uint64_t* residual,
                      bool fold_residual, hipStream_t s);

void crt_carry_launch_linked(const Geom& g, const uint64_t* in61, const uint32_t* in31, uint64_t* digits, uint64_t* edge, hipStream_t s);

}  // namespace crt
}  // namespace mi355
