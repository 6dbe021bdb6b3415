// Multiplication-free small DFTs over GF(P): every root of unity of order <= 64 is a power of two
// (omega_64 = 7^((P-1)/64) = 2^39), so radix-8 butterflies need only add/sub and constant shifts.
// The reference gets the same effect for radix-4 only (sqrt(-1) = 2^48, kernels/marin.cl:148,304-318).
#pragma once
#include "gf.hpp"

namespace gf {

constexpr unsigned LOG2_W64 = 39;   // omega_64 = 2^39; omega_32 = 2^78, omega_16 = 2^156, omega_8 = 2^120 = -2^24, omega_4 = 2^48
constexpr unsigned shift64(unsigned e) { return (LOG2_W64 * e) % 192; }   // omega_64^e = 2^shift64(e)

// out[k] = sum_j in[j] * W^(jk), W = omega_8 (forward) or omega_8^-1 (INV, unnormalised), in place.
// Signs of W^1 = -2^24, W^3 = -2^72 (forward) and W^-2 = -2^48 (inverse) are folded into the order
// of the subtractions.
// LAZY: outputs that may be left un-folded because the caller shifts or multiplies them next (mul, mul_u32, mul_pow2 with a non-zero
// shift accept any 64-bit representative):
//   0 none;
//   1 outputs 1..3 are lazy sums and output 5 may be un-folded too (round 4: e0 is a lazy sum -- add_lazy needs ONE canonical operand and
//     sub a canonical subtrahend, and e0 only ever meets the canonical e1); outputs 0, 2's partner 6, 4 and 7 stay canonical;
//   2 every output but 7 may be un-folded (round 4: a0, c0 and e0 are lazy sums as well; each of them meets a canonical partner in the
//     next level: a0 with a2, c0 with c1, e0 with e1).
// Three instructions fewer per transform at LAZY = 2, one at LAZY = 1 (GF_R3_FORMS restores the round-3 network for A/B builds).
template <bool INV, int LAZY = 0>
GF_HD void dft8(uint64_t (&x)[8]) {
#if defined(GF_R3_FORMS)
  constexpr bool LZ_A0 = false, LZ_E0 = false;
#else
  constexpr bool LZ_A0 = (LAZY >= 2), LZ_E0 = (LAZY >= 1);
#endif
  const uint64_t a0 = LZ_A0 ? add_lazy(x[0], x[4]) : add(x[0], x[4]);
  const uint64_t a1 = add(x[1], x[5]), a2 = add(x[2], x[6]), a3 = add(x[3], x[7]);
  uint64_t b0 = sub(x[0], x[4]), b1, b2, b3;
  if (!INV) {
    b1 = mul_pow2(sub(x[5], x[1]), 24);
    b2 = mul_pow2(sub(x[2], x[6]), 48);
    b3 = mul_pow2(sub(x[7], x[3]), 72);
  } else {
    b1 = mul_pow2(sub(x[1], x[5]), 72);
    b2 = mul_pow2(sub(x[6], x[2]), 48);
    b3 = mul_pow2(sub(x[3], x[7]), 24);
  }
  const uint64_t c0 = LZ_A0 ? add_lazy(a0, a2) : add(a0, a2);   // a2 canonical
  const uint64_t c1 = add(a1, a3), d0 = sub(a0, a2);            // d0: a lazy minuend gives an un-folded difference (LAZY = 2: outputs 2, 6)
  const uint64_t d1 = mul_pow2(INV ? sub(a3, a1) : sub(a1, a3), 48);
  const uint64_t e0 = LZ_E0 ? add_lazy(b0, b2) : add(b0, b2);   // b0, b2 canonical
  const uint64_t e1 = add(b1, b3), f0 = sub(b0, b2);
  const uint64_t f1 = mul_pow2(INV ? sub(b3, b1) : sub(b1, b3), 48);
  x[0] = (LAZY >= 2) ? add_lazy(c0, c1) : add(c0, c1); x[4] = sub(c0, c1);
  x[2] = (LAZY >= 1) ? add_lazy(d0, d1) : add(d0, d1); x[6] = sub(d0, d1);
  x[1] = (LAZY >= 1) ? add_lazy(e0, e1) : add(e0, e1); x[5] = sub(e0, e1);
  x[3] = (LAZY >= 1) ? add_lazy(f0, f1) : add(f0, f1); x[7] = sub(f0, f1);
}

}  // namespace gf
