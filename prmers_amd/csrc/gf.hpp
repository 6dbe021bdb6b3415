// GF(P), P = 2^64 - 2^32 + 1 (Goldilocks) -- host + gfx950 device arithmetic.
//
// Written from scratch for CDNA4: there is no 64x64->128 multiplier on the VALU, so mul() is
// four 32x32+64 multiply-adds (v_mad_u64_u32) followed by the 2^64 = 2^32-1, 2^96 = -1 folding.
// Device forms (round 2, measured in profiles/r02_microbench_gf.txt): the VALU issues VOP1/VOP2 adds, logic
// and selects at one wave-instruction per ~2.7 cycles and everything else (carry forms, compares, VOP3
// encodings, 64-bit shifts) per ~4.5-5.2, so the reductions below (a) let v_mad_u64_u32 do the "x + h (2^32-1)"
// addition and hand its carry-out to the correction as a lane mask, (b) keep the select masks in VCC with the
// selected constant in a VGPR (VOP2 v_cndmask_b32), (c) apply "- (2^32-1)" as one signed multiply-add
// (65535 * -65537), and (d) send the two corrections of a product that almost never occur (probability
// ~2^-32 per lane) through a wave-uniform branch instead of computing their masks every time.
// Field definition follows the reference (include/marin/arith.h:24-72, kernels/marin.cl:112-148):
// same prime, same generator 7, sqrt(-1) = 2^48.  Everything here is canonical (inputs and
// outputs in [0, P)) unless a function says "lazy".
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GF_HD __host__ __device__ __forceinline__
#else
#define GF_HD static inline
#endif

namespace gf {

constexpr uint64_t P = 0xffffffff00000001ull;
constexpr uint64_t EPS = 0xffffffffull;  // 2^64 mod P = 2^32 - 1

#if defined(__HIP_DEVICE_COMPILE__) && !defined(GF_PORTABLE_DEVICE)
#define GF_ASM 1
namespace dev {
// Per-thread constants behind an opaque asm (no inputs: the copies within a kernel are merged and hoisted), so
// that "mask ? c : 0" stays a VOP2 v_cndmask_b32 with a VGPR operand instead of a VOP3 one with literals.
__device__ __forceinline__ uint32_t k_ones() { uint32_t r; asm("v_mov_b32 %0, -1" : "=v"(r)); return r; }
__device__ __forceinline__ uint32_t k_ffff() { uint32_t r; asm("v_mov_b32 %0, 0xffff" : "=v"(r)); return r; }
constexpr uint64_t PM1 = 0xffffffff00000000ull;   // P - 1

// d - (v ? 2^32-1 : 0) for v in {0, 65535}:  d + v * -65537 (one v_mad_i64_i32)
// The unused carry-out goes to a fixed scalar pair that nothing ever reads: an allocated one could be rewritten by a
// scalar instruction and read by a vector one right after this statement, inside the two wait states the compiler
// keeps after VALU writes of an SGPR that it can see (it cannot see into asm).
__device__ __forceinline__ uint64_t sub_eps_if(uint64_t d, uint32_t v) {
  uint64_t r;
  asm("v_mad_i64_i32 %0, s[94:95], %1, %2, %3" : "=v"(r) : "v"(v), "s"(int32_t(-65537)), "v"(d) : "s94", "s95");
  return r;
}
// x*y + z, carry out as a lane mask
__device__ __forceinline__ uint64_t mad_c(uint32_t x, uint32_t y, uint64_t z, uint64_t& cy) {
  uint64_t r;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(cy) : "v"(x), "v"(y), "v"(z));
  return r;
}
// x + h (2^32-1) for x + h (2^32-1) < 2^64 + (2^32-1)^2:  canonical.  A carry means "+ 2^64 == + EPS" and leaves
// r < h EPS, so the addition of EPS cannot carry again; without a carry r >= P is folded the same way.
__device__ __forceinline__ uint64_t mad_eps_fold(uint32_t h, uint64_t x) {
  uint64_t r, t;
  uint32_t m;
  asm("v_mad_u64_u32 %0, vcc, %3, -1, %4\n\t"
      "v_cmp_lt_u64 %1, %5, %0\n\t"
      "s_or_b64 vcc, vcc, %1\n\t"
      "v_cndmask_b32_e32 %2, 0, %6, vcc"
      : "=&v"(r), "=&s"(t), "=&v"(m)
      : "v"(h), "v"(x), "s"(PM1), "v"(k_ones())
      : "vcc", "scc");   // s_or_b64 writes SCC
  return r + uint64_t(m);
}
// lo - hh - cin (cin: lane mask), borrow out as a lane mask
__device__ __forceinline__ uint64_t sub32_c(uint64_t lo, uint32_t hh, uint64_t cin, uint64_t& bw) {
  uint32_t r0, r1;
  asm("v_subb_co_u32 %0, vcc, %3, %4, %5\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, %2, 0, %6, vcc"
      : "=&v"(r0), "=v"(r1), "=s"(bw)
      : "v"(uint32_t(lo)), "v"(hh), "s"(cin), "v"(uint32_t(lo >> 32))
      : "vcc");
  return (uint64_t(r1) << 32) | r0;
}
// x + hl (2^32-1), canonical, where x = lo - hh - cin was formed with borrow mask bw (sub32_c):
//   bw = 0: x exact;           carry -> + EPS (every other product);  r >= P without carry -> + EPS   (rare)
//   bw = 1: x = true + 2^64;   carry -> exact;                        no carry -> - EPS               (rare)
// bw itself needs lo < 2^32 and so is rare: the fast path is "carry -> + EPS" with one test that no lane has
// bw or r >= P; the slow path applies +-EPS = 65535 * +-65537 to the lanes that need it.
__device__ __forceinline__ uint64_t reduce_tail(uint32_t hl, uint64_t x, uint64_t bw) {
  uint64_t r, t;
  uint32_t m, k;
  asm("v_mad_u64_u32 %0, vcc, %4, -1, %5\n\t"
      "v_cmp_lt_u64 %1, %7, %0\n\t"
      "s_or_b64 %1, %1, %6\n\t"            // r >= P, or borrowed
      "s_and_b64 %1, %1, exec\n\t"
      "s_cbranch_scc0 .Lgf_fast%=\n\t"
      "s_andn2_b64 %1, %1, vcc\n\t"        // ... and no carry: the lanes that change (with bw: -EPS, without: +EPS)
      "v_mov_b32 %2, 0x10001\n\t"
      "v_mov_b32 %3, 0xfffeffff\n\t"
      "v_cndmask_b32 %2, %2, %3, %6\n\t"   // 65537 or -65537
      "v_mov_b32 %3, 0xffff\n\t"
      "v_cndmask_b32 %3, 0, %3, %1\n\t"
      "v_mad_i64_i32 %0, %1, %3, %2, %0\n\t"
      "s_andn2_b64 vcc, vcc, %6\n"          // a borrowed lane that carried is exact
      ".Lgf_fast%=:\n\t"
      "v_cndmask_b32_e32 %2, 0, %8, vcc"
      : "=&v"(r), "=&s"(t), "=&v"(m), "=&v"(k)
      : "v"(hl), "v"(x), "s"(bw), "s"(PM1), "v"(k_ones())
      : "vcc", "scc");
  return r + uint64_t(m);
}
}  // namespace dev
#endif

GF_HD uint64_t add(uint64_t a, uint64_t b) {
#if defined(GF_ASM)
  unsigned c0, c1;
  const uint32_t lo = __builtin_addc((uint32_t)a, (uint32_t)b, 0u, &c0);
  const uint32_t hi = __builtin_addc((uint32_t)(a >> 32), (uint32_t)(b >> 32), c0, &c1);
  const uint64_t s = ((uint64_t)hi << 32) | lo;
  const uint32_t ones = dev::k_ones();
  const uint32_t m = ((c1 != 0) | (s >= P)) ? ones : 0u;
  return s + m;
#else
  uint64_t s = a + b;
  // true sum < 2P.  Subtract P (== add EPS mod 2^64) when the 65-bit sum is >= P.
  bool ge = (s < a) | (s >= P);
  return s + (ge ? EPS : 0ull);
#endif
}

// a + b without the final ">= P" fold: the result is congruent to a + b but may lie anywhere in [0, 2^64).
// For operands <= P the single carry fold cannot overflow again (a + b - 2^64 + EPS < 2^64).  Only for
// values whose next use is a multiplication (mul, mul_u32, mul_pow2 accept any 64-bit operand).
GF_HD uint64_t add_lazy(uint64_t a, uint64_t b) {
#if defined(GF_ASM)
  unsigned c0, c1;
  const uint32_t lo = __builtin_addc((uint32_t)a, (uint32_t)b, 0u, &c0);
  const uint32_t hi = __builtin_addc((uint32_t)(a >> 32), (uint32_t)(b >> 32), c0, &c1);
  const uint32_t ones = dev::k_ones();
  const uint32_t m = c1 ? ones : 0u;
  return (((uint64_t)hi << 32) | lo) + m;
#else
  uint64_t s = a + b;
  return s + ((s < a) ? EPS : 0ull);
#endif
}

// any 64-bit representative -> [0, P)
GF_HD uint64_t fold(uint64_t a) {
#if defined(GF_ASM)
  const uint32_t ones = dev::k_ones();
  const uint32_t m = (a >= P) ? ones : 0u;
  return a + m;
#else
  return a + ((a >= P) ? EPS : 0ull);
#endif
}

GF_HD uint64_t sub(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // the borrow comes out of the 32-bit subtract-with-borrow pair itself (one 64-bit compare less than
  // the portable form); d - EPS == d + P (mod 2^64)
  unsigned c0, c1;
  uint32_t lo = __builtin_subc((uint32_t)a, (uint32_t)b, 0u, &c0);
  uint32_t hi = __builtin_subc((uint32_t)(a >> 32), (uint32_t)(b >> 32), c0, &c1);
  uint64_t d = ((uint64_t)hi << 32) | lo;
#if defined(GF_ASM)
  const uint32_t kf = dev::k_ffff();
  return dev::sub_eps_if(d, c1 ? kf : 0u);   // d + P == d - EPS (mod 2^64)
#else
  return d + (c1 ? P : 0ull);
#endif
#else
  uint64_t d = a - b;
  return d - ((a < b) ? EPS : 0ull);
#endif
}

GF_HD uint64_t neg(uint64_t a) { return a ? P - a : 0ull; }

GF_HD uint64_t dbl(uint64_t a) { return add(a, a); }

// (hi:lo) mod P for hi:lo < P^2  (hi = hh*2^32 + hl):  lo + hl*(2^32-1) - hh.
GF_HD uint64_t reduce128(uint64_t lo, uint64_t hi) {
  uint32_t hh = (uint32_t)(hi >> 32), hl = (uint32_t)hi;
  // s = hl*(2^32-1) - hh + (2^32-1)  in [0, P): never wraps.
  uint64_t s = (((uint64_t)hl << 32) | (uint32_t)(~hh)) - hl;
  uint64_t r = s + lo;
  // carry: r + 2^64 == X + EPS  =>  r == X (mod P) because 2^64 == EPS;  r < s < P.
  // no carry: r == X + EPS: remove EPS, adding P back if that borrows.
  if (r >= s) {
    uint64_t t = r - EPS;
    r = (r < EPS) ? t - EPS : t;
  }
  return r;
}

GF_HD void mul64x64(uint64_t a, uint64_t b, uint64_t& lo, uint64_t& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
  uint32_t b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  uint64_t t0 = (uint64_t)a0 * b0;
  uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32);
  uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t1;
  uint64_t t3 = (uint64_t)a1 * b1 + (t1 >> 32) + (t2 >> 32);
  lo = (t2 << 32) | (uint32_t)t0;
  hi = t3;
#else
  unsigned __int128 t = (unsigned __int128)a * b;
  lo = (uint64_t)t;
  hi = (uint64_t)(t >> 64);
#endif
}

GF_HD uint64_t mul(uint64_t a, uint64_t b) {
#if defined(GF_ASM)
  const uint32_t a0 = uint32_t(a), a1 = uint32_t(a >> 32), b0 = uint32_t(b), b1 = uint32_t(b >> 32);
  const uint64_t t0 = uint64_t(a0) * b0;
  const uint64_t t1 = uint64_t(a0) * b1 + (t0 >> 32);
  uint64_t c, bw;
  const uint64_t t2 = dev::mad_c(a1, b0, t1, c);            // the carry c has weight 2^96: the product's high half is t3 + c 2^32
  const uint64_t t3 = uint64_t(a1) * b1 + (t2 >> 32);
  const uint64_t lo = (t2 << 32) | uint32_t(t0);
  // lo + hl EPS - hh - c.  The two instructions that form t3 separate the write of c from its read below
  // (gfx950 needs two wait states between a VALU write of an SGPR and a VALU read; tools/check_isa_hazards.py).
  const uint64_t x = dev::sub32_c(lo, uint32_t(t3 >> 32), c, bw);
  return dev::reduce_tail(uint32_t(t3), x, bw);
#else
  uint64_t lo, hi;
  mul64x64(a, b, lo, hi);
  return reduce128(lo, hi);
#endif
}

GF_HD uint64_t sqr(uint64_t a) { return mul(a, a); }

// a * b for a 32-bit b (digit * weight, small constants): two multiply-adds.
GF_HD uint64_t mul_u32(uint64_t a, uint32_t b) {
  uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
  uint64_t t0 = (uint64_t)a0 * b;
  uint64_t t1 = (uint64_t)a1 * b + (t0 >> 32);
  uint64_t lo = (t1 << 32) | (uint32_t)t0;
  uint64_t hi = t1 >> 32;  // < 2^32: hh = 0
  // lo + hi*(2^32-1)
#if defined(GF_ASM)
  return dev::mad_eps_fold(uint32_t(hi), lo);
#else
  uint64_t s = (hi << 32) - hi;  // < P
  return add(lo, s);  // lo may be >= P: see mul_pow2's note
#endif
}

// a * 2^s for 0 <= s < 192 (2 is a primitive 192nd root of unity, 2^96 = -1); a may be any 64-bit
// representative (s = 0 returns it unchanged, every other case returns a canonical value, or P for a negated zero
// on the device).
// With s a compile-time constant after inlining, or wave-uniform (scalar branches), this is a few
// shifts and one or two modular add/sub -- a third to a half of a general mul().
// add() below is called with a possibly non-canonical first operand (< 2^64) and a second operand
// <= (2^32-1)^2; it still returns the canonical sum ("lazy operands of add").
GF_HD uint64_t mul_pow2(uint64_t a, unsigned s) {
  bool negate = false;
  if (s >= 96) { s -= 96; negate = true; }
  uint64_t r;
  if (s == 0) {
    r = negate ? fold(a) : a;   // the only case that passes the operand through: fold it before a negation
  } else if (s < 32) {
    // a*2^s = hi*2^64 + lo, hi < 2^32:  lo + hi*(2^32-1)
    const uint64_t lo = a << s, hi = a >> (64 - s);
#if defined(GF_ASM)
    r = dev::mad_eps_fold(uint32_t(hi), lo);
#else
    r = add(lo, (hi << 32) - hi);
#endif
  } else if (s < 64) {
    // a*2^s = top*2^96 + mid*2^64 + lh*2^32 (the low word of a << s is zero)
    //       = (lh + mid)*2^32 - (mid + top).  The 33-bit sum lh + mid = c*2^32 + xhi folds its carry as
    // c*2^64 = c*(2^32-1) into the empty low word, which leaves a canonical value; one sub() finishes.
    const uint64_t h = a >> (64 - s);
    const uint32_t lh = (uint32_t)a << (s - 32), mid = (uint32_t)h, top = (uint32_t)(h >> 32);
    const uint32_t xhi = lh + mid;
    const uint64_t x = ((uint64_t)xhi << 32) | ((xhi < lh) ? 0xffffffffu : 0u);
    r = sub(x, (uint64_t)mid + top);
  } else if (s == 64) {
    const uint64_t ah = a >> 32, al = a & 0xffffffffull;
    r = sub((al << 32) - al, ah);
  } else {
    // 64 < s < 96: (a*2^k)*2^64, k = s-64: a*2^k = hi2*2^64 + l1*2^32 + l0
    //   l0*2^64 = l0*(2^32-1);  l1*2^96 = -l1;  hi2*2^128 = -hi2*2^32
    const unsigned k = s - 64;
    const uint64_t lo2 = a << k, hi2 = a >> (64 - k);
    const uint64_t l1 = lo2 >> 32, l0 = lo2 & 0xffffffffull;
    r = sub((l0 << 32) - l0, l1 + (hi2 << 32));
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // On the device a negated zero is left as P (two instructions instead of five).  P behaves as zero in
  // add / sub / mul / mul_pow2 (their bounds hold for any operand <= P) and every kernel ends its chain
  // with a multiplication, whose result is canonical again, before anything is compared or split into digits.
  return negate ? P - r : r;
#else
  return negate ? neg(r) : r;
#endif
}

// sqrt(-1) = 2^48
GF_HD uint64_t muli(uint64_t a) { return mul_pow2(a, 48); }

GF_HD uint64_t half(uint64_t a) { return (a & 1) ? (a >> 1) + ((P + 1) >> 1) : (a >> 1); }

GF_HD uint64_t pow(uint64_t a, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = mul(r, a);
    a = mul(a, a);
    e >>= 1;
  }
  return r;
}

GF_HD uint64_t inv(uint64_t a) { return pow(a, P - 2); }

// primitive n-th root of unity, n | P-1 (generator 7: arith.h:72)
GF_HD uint64_t root_of_unity(uint64_t n) { return pow(7, (P - 1) / n); }

// n-th root of two, n | (P-1)/192 (ibdwt.h:116: 554^((P-1)/192) = 2)
GF_HD uint64_t root_of_two(uint64_t n) { return pow(554, (P - 1) / 192 / n); }

}  // namespace gf
