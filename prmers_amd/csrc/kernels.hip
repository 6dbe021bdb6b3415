// gfx950 kernels for the Goldilocks IBDWT squaring (generic, size-agnostic set).
//
// One squaring x <- x^2 * a mod 2^p-1 is three sweeps over m = n/2 pairs (u64,u64), m = M1 x M2:
//   k_front : digits(u32, tile-major) -> weight -> length-M1 column DFT (LDS) -> twiddle -> work buffer
//   k_middle: length-M2 row DFT (LDS) -> pointwise square / multiply mod (t^2 - rho) -> inverse row DFT
//   k_back  : twiddle^-1 -> inverse column DFT (LDS) -> unweight -> base-2^width carry over runs of 2C
//             digits -> digits(u32) + one carry word per run
//   k_carry_fix: adds each run's carry word into the first digits of the next run (weak carry)
// This replaces the reference's kernel chain forward64_0 / forward256 / sqr512 / backward256 /
// backward64_0 / carry_weight_mul_p1 / carry_weight_p2 (include/marin/engine_gpu.h:1568-1630,
// kernels/marin.cl:989-1075,1517-1528,1696-1728,2198-2216).  Butterfly algebra follows
// marin.cl:304-392 (radix-4 with sqrt(-1), pair squaring mod t^2 - rho), restated for a plain
// DIF/DIT ordering with universal root tables.
//
// All kernels take the geometry in a DevPlan by value; nothing is compiled per exponent.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "gf.hpp"
#include "kernels.hpp"

namespace mi355 {

struct alignas(16) P2 { uint64_t a, b; };

__device__ __forceinline__ P2 p2_add(P2 x, P2 y) { return {gf::add(x.a, y.a), gf::add(x.b, y.b)}; }
__device__ __forceinline__ P2 p2_sub(P2 x, P2 y) { return {gf::sub(x.a, y.a), gf::sub(x.b, y.b)}; }
__device__ __forceinline__ P2 p2_mul(P2 x, uint64_t w) { return {gf::mul(x.a, w), gf::mul(x.b, w)}; }
__device__ __forceinline__ P2 p2_shift48(P2 x) { return {gf::mul_pow2(x.a, 48), gf::mul_pow2(x.b, 48)}; }

// Accesses to the data the sweeps hand to each other (work buffer, digits, run carries).  COH = false: plain loads and stores (the
// three-launch chain: a kernel boundary makes them visible).  COH = true (k_coop: producer and consumer run in the same launch on
// different XCDs, whose L2s are not coherent with each other): agent-scope relaxed atomics, i.e. loads and stores with the sc1 bit that
// go to the memory side instead of the XCD's L2 -- the grid barrier then needs no L2 write-back / invalidate, which costs ~18 us each
// (measured: 0.073 ms per squaring at C2 with agent-scope fences against 0.032 for three launches).
template <bool COH> __device__ __forceinline__ uint64_t ld64(const uint64_t* p) {
  if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool COH> __device__ __forceinline__ void st64(uint64_t* p, uint64_t v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH> __device__ __forceinline__ P2 ldp2(const P2* p) {
  if (COH) return {ld64<true>(&p->a), ld64<true>(&p->b)};
  return *p;
}
template <bool COH> __device__ __forceinline__ void stp2(P2* p, P2 v) {
  if (COH) { st64<true>(&p->a, v.a); st64<true>(&p->b, v.b); }
  else *p = v;
}
template <bool COH> __device__ __forceinline__ uint2 ldu2(const uint2* p) {
  if (COH) { const uint64_t v = ld64<true>(reinterpret_cast<const uint64_t*>(p)); return make_uint2(uint32_t(v), uint32_t(v >> 32)); }
  return *p;
}
template <bool COH> __device__ __forceinline__ void stu2(uint2* p, uint2 v) {
  if (COH) st64<true>(reinterpret_cast<uint64_t*>(p), uint64_t(v.x) | (uint64_t(v.y) << 32));
  else *p = v;
}

// omega_m^e from the two-level table (e < m)
__device__ __forceinline__ uint64_t tw_lookup(const DevPlan& pl, uint64_t e) {
  const uint64_t lo = pl.TWlo[e & ((1u << pl.twh) - 1)], hi = pl.TWhi[e >> pl.twh];
  return gf::mul(lo, hi);
}

// column-DFT output slot -> frequency k1 (radix-5 block major, bit reversed inside the block)
__device__ __forceinline__ uint32_t freq1(const DevPlan& pl, uint32_t pos) {
  const uint32_t blk = pos >> pl.logL1, q = pos & (pl.L1 - 1);   // L1 is a power of two
  const uint32_t rq = pl.logL1 ? (__brev(q) >> (32 - pl.logL1)) : 0u;
  return col_label(pl, blk, rq);
}

// p*j mod n for digit (i1, x = 2*i2+b), its width, and whether the factored weight wrapped
__device__ __forceinline__ void digit_info(const DevPlan& pl, uint32_t sa, uint32_t sb, uint32_t& width, bool& wrap) {
  uint64_t s = uint64_t(sa) + sb;
  wrap = (sa > 0) && (sb > 0) && (s <= pl.n);
  if (s >= pl.n) s -= pl.n;
  width = pl.q + ((s + pl.t > 0) ? 1u : 0u) + ((s + pl.t > pl.n) ? 1u : 0u) - ((s > 0) ? 1u : 0u);
}

// ---------------------------------------------------------------------------------------------
// radix-4 / radix-2 passes over an LDS array of pairs.
// Elements of one transform sit at X[(base + i) * stride + col]; `groups` independent transforms of
// length L (blocks x columns) are processed by the whole work-group.  root[e * rstep] = omega_L^e.
// ---------------------------------------------------------------------------------------------
// The same passes with one PLANE of a butterfly per thread (the two words of a pair go through identical, independent arithmetic):
// twice the threads, half the instruction stream per wave.  Small tiles are latency-bound -- one or two waves per SIMD, so a launch
// lasts as long as the dependent instruction stream of one wave (C2: 0.041 ms for three launches whose VALU work is 3 us) -- and the
// way to shorten that stream is more threads with less work each, not wider butterflies (a radix-8 form made C2 35 % slower).
template <bool INVERSE>
__device__ __forceinline__ void lds_pow2_dft_planes(P2* X, uint32_t L, uint32_t logL, uint32_t nblocks, uint32_t ncols, uint32_t logcols,
                                                    const uint64_t* __restrict__ root, uint32_t rootN, uint32_t rstep, uint32_t tid, uint32_t nthr) {
  uint64_t* Xw = reinterpret_cast<uint64_t*>(X);   // element e, plane b at Xw[2 e + b]
  const uint32_t n4 = logL / 2, has2 = logL & 1;
  const uint32_t npass = n4 + has2;
  for (uint32_t ps = 0; ps < npass; ++ps) {
    const uint32_t pf = INVERSE ? (npass - 1 - ps) : ps;
    if (pf < n4) {
      const uint32_t loglen = logL - 2 * pf, len = 1u << loglen, q = len >> 2;
      const uint32_t per = L >> 2;
      const uint32_t total = per * nblocks * ncols * 2;
      const uint32_t tstep = (L >> loglen) * rstep;
      for (uint32_t idx2 = tid; idx2 < total; idx2 += nthr) {
        const uint32_t plane = idx2 & 1, idx = idx2 >> 1;
        const uint32_t col = idx & (ncols - 1), bi = idx >> logcols;
        const uint32_t blk = bi >> (logL - 2), bj = bi & (per - 1);
        const uint32_t sub = bj >> (loglen - 2), t = bj & (q - 1);
        const uint32_t e0 = 2 * ((blk * L + sub * len + t) * ncols + col) + plane, es = 2 * q * ncols;
        const uint32_t r1 = t * tstep;
        const uint64_t x0 = Xw[e0], x1 = Xw[e0 + es], x2 = Xw[e0 + 2 * es], x3 = Xw[e0 + 3 * es];
        if (!INVERSE) {
          const uint64_t w1 = root[r1], w2 = root[2 * r1], w3 = root[3 * r1];
          const uint64_t a = gf::add(x0, x2), b = gf::add(x1, x3), c = gf::sub(x0, x2), d = gf::mul_pow2(gf::sub(x1, x3), 48);
          Xw[e0] = gf::add(a, b);
          Xw[e0 + es] = gf::mul(gf::sub(a, b), w2);
          Xw[e0 + 2 * es] = gf::mul(gf::add(c, d), w1);
          Xw[e0 + 3 * es] = gf::mul(gf::sub(c, d), w3);
        } else {
          const uint64_t w1 = root[r1 ? rootN - r1 : 0], w2 = root[r1 ? rootN - 2 * r1 : 0], w3 = root[r1 ? rootN - 3 * r1 : 0];
          const uint64_t y1 = gf::mul(x1, w2);
          const uint64_t A = gf::add(x0, y1), B = gf::sub(x0, y1);
          const uint64_t y2 = gf::mul(x2, w1), y3 = gf::mul(x3, w3);
          const uint64_t Cc = gf::add(y2, y3), D = gf::mul_pow2(gf::sub(y3, y2), 48);
          Xw[e0] = gf::add(A, Cc);
          Xw[e0 + 2 * es] = gf::sub(A, Cc);
          Xw[e0 + es] = gf::add(B, D);
          Xw[e0 + 3 * es] = gf::sub(B, D);
        }
      }
    } else {
      const uint32_t per = L >> 1, total = per * nblocks * ncols * 2;
      for (uint32_t idx2 = tid; idx2 < total; idx2 += nthr) {
        const uint32_t plane = idx2 & 1, idx = idx2 >> 1;
        const uint32_t col = idx & (ncols - 1), bi = idx >> logcols;
        const uint32_t blk = bi >> (logL - 1), bj = bi & (per - 1);
        const uint32_t e0 = 2 * ((blk * L + 2 * bj) * ncols + col) + plane;
        const uint64_t u = Xw[e0], v = Xw[e0 + 2 * ncols];
        Xw[e0] = gf::add(u, v);
        Xw[e0 + 2 * ncols] = gf::sub(u, v);
      }
    }
    __syncthreads();
  }
}

template <bool INVERSE>
__device__ __forceinline__ void lds_pow2_dft(P2* X, uint32_t L, uint32_t logL, uint32_t nblocks, uint32_t ncols, uint32_t logcols,
                                             const uint64_t* __restrict__ root, uint32_t rootN, uint32_t rstep,
                                             uint64_t I4, uint32_t tid, uint32_t nthr) {
  // small tiles: one plane per thread (see above); the work-group was sized for that by the launcher
  if (nthr * 2 >= (L >> 2) * nblocks * ncols * 2 && nthr > (L >> 2) * nblocks * ncols) {
    lds_pow2_dft_planes<INVERSE>(X, L, logL, nblocks, ncols, logcols, root, rootN, rstep, tid, nthr);
    return;
  }
  // forward: len = L, L/4, ... (radix-4) then a final radix-2 when log2 L is odd; inverse mirrors
  const uint32_t n4 = logL / 2, has2 = logL & 1;
  const uint32_t npass = n4 + has2;
  for (uint32_t ps = 0; ps < npass; ++ps) {
    const uint32_t pf = INVERSE ? (npass - 1 - ps) : ps;  // pass index in forward order
    if (pf < n4) {
      const uint32_t loglen = logL - 2 * pf, len = 1u << loglen, q = len >> 2;
      const uint32_t per = L >> 2;  // butterflies per transform
      const uint32_t total = per * nblocks * ncols;
      const uint32_t tstep = (L >> loglen) * rstep;  // omega_len^t = root[t * tstep]
      for (uint32_t idx = tid; idx < total; idx += nthr) {
        // ncols, per and q are powers of two: shifts and masks, no integer division in the loop
        const uint32_t col = idx & (ncols - 1), bi = idx >> logcols;
        const uint32_t blk = bi >> (logL - 2), bj = bi & (per - 1);
        const uint32_t sub = bj >> (loglen - 2), t = bj & (q - 1);
        const uint32_t e0 = (blk * L + sub * len + t) * ncols + col, es = q * ncols;
        const uint32_t r1 = t * tstep;
        P2 x0 = X[e0], x1 = X[e0 + es], x2 = X[e0 + 2 * es], x3 = X[e0 + 3 * es];
        if (!INVERSE) {
          const uint64_t w1 = root[r1], w2 = root[2 * r1], w3 = root[3 * r1];
          const P2 a = p2_add(x0, x2), b = p2_add(x1, x3), c = p2_sub(x0, x2), d = p2_shift48(p2_sub(x1, x3));   // omega_4 = 2^48 (checked in make_plan)
          X[e0] = p2_add(a, b);
          X[e0 + es] = p2_mul(p2_sub(a, b), w2);
          X[e0 + 2 * es] = p2_mul(p2_add(c, d), w1);
          X[e0 + 3 * es] = p2_mul(p2_sub(c, d), w3);
        } else {
          // inverse roots: omega^-e = root[(N - e) % N]
          const uint64_t w1 = root[r1 ? rootN - r1 : 0], w2 = root[r1 ? rootN - 2 * r1 : 0], w3 = root[r1 ? rootN - 3 * r1 : 0];
          const P2 y1 = p2_mul(x1, w2);
          const P2 A = p2_add(x0, y1), B = p2_sub(x0, y1);
          const P2 y2 = p2_mul(x2, w1), y3 = p2_mul(x3, w3);
          const P2 Cc = p2_add(y2, y3), D = p2_shift48(p2_sub(y3, y2));   // omega_4^-1 = -2^48
          X[e0] = p2_add(A, Cc);
          X[e0 + 2 * es] = p2_sub(A, Cc);
          X[e0 + es] = p2_add(B, D);
          X[e0 + 3 * es] = p2_sub(B, D);
        }
      }
    } else {
      // radix-2, len = 2: no twiddle
      const uint32_t per = L >> 1, total = per * nblocks * ncols;
      for (uint32_t idx = tid; idx < total; idx += nthr) {
        const uint32_t col = idx & (ncols - 1), bi = idx >> logcols;
        const uint32_t blk = bi >> (logL - 1), bj = bi & (per - 1);
        const uint32_t e0 = (blk * L + 2 * bj) * ncols + col;
        const P2 u = X[e0], v = X[e0 + ncols];
        X[e0] = p2_add(u, v);
        X[e0 + ncols] = p2_sub(u, v);
      }
    }
    __syncthreads();
  }
}

// 5-point DFT, X[k] = sum_r x[r] w^(rk) with w = omega_5 (INVERSE: w^-1), 4 general multiplications:
//   t1 = x1+x4, t2 = x2+x3, t3 = x1-x4, t4 = x2-x3, t5 = t1+t2;  X0 = x0 + t5
//   A = x0 - t5/4 (1 + (w+w^4) + (w^2+w^3) = 0; -1/4 = 2^94);  B1,2 = A +- beta (t1 - t2), beta = ((w+w^4) - (w^2+w^3))/4
//   P = k1 t3 + k2 t4, Q = k2 t3 - k1 t4 with k1 = (w-w^4)/2, k2 = (w^2-w^3)/2, from three products
//   k1 (t3+t4), (k2-k1) t4, (k1+k2) t3;  X1,4 = B1 +- P, X2,3 = B2 +- Q (signs swapped for the inverse).
// c5 = {beta, k1, k2-k1, k1+k2} (plan.hpp).
template <bool INVERSE>
__device__ __forceinline__ void dft5(P2 (&x)[5], const uint64_t (&c5)[4]) {
  const P2 t1 = p2_add(x[1], x[4]), t2 = p2_add(x[2], x[3]), t3 = p2_sub(x[1], x[4]), t4 = p2_sub(x[2], x[3]);
  const P2 t5 = p2_add(t1, t2);
  const P2 A = p2_add(x[0], P2{gf::mul_pow2(t5.a, 94), gf::mul_pow2(t5.b, 94)});
  const P2 m2 = p2_mul(p2_sub(t1, t2), c5[0]);
  const P2 B1 = p2_add(A, m2), B2 = p2_sub(A, m2);
  const P2 m3 = p2_mul(p2_add(t3, t4), c5[1]), m4 = p2_mul(t4, c5[2]), m5 = p2_mul(t3, c5[3]);
  const P2 Pp = p2_add(m3, m4), Q = p2_sub(m5, m3);
  x[0] = p2_add(x[0], t5);
  if (!INVERSE) { x[1] = p2_add(B1, Pp); x[4] = p2_sub(B1, Pp); x[2] = p2_add(B2, Q); x[3] = p2_sub(B2, Q); }
  else          { x[1] = p2_sub(B1, Pp); x[4] = p2_add(B1, Pp); x[2] = p2_sub(B2, Q); x[3] = p2_add(B2, Q); }
}

// radix-5 stage of the column DFT (first forward / last inverse): 5 blocks of L1.
// Mixed-radix form (lab_u == 1: the split sweeps, and every plan with L1 = 1): inputs (L1 r + t), twiddle omega_M1^(t k) on output k.
// Prime-factor form (round 4; 5 and L1 are coprime): inputs i1 = (L1 r + 5 t) mod M1, no twiddle, output k0 in place at L1 k0 + (5 t mod L1) --
// the same five slots -- so block k0 holds its sequence permuted by t -> 5 t mod L1, the power-of-two transform of that block puts the
// frequency kr = 5 k mod L1 at output k, and slot (k0, k) holds the column frequency (lab_u k0 + 5 k) mod M1 with lab_u = L1 (L1^-1 mod 5):
// the frequency label the engine hands to every kernel (kernels.hpp col_label).  Four table products per butterfly and direction fewer.
template <bool INVERSE>
__device__ __forceinline__ void lds_radix5(const DevPlan& pl, P2* X, uint32_t ncols, uint32_t tid, uint32_t nthr) {
  const uint32_t L1 = pl.L1, total = L1 * ncols;
  const bool pfa = pl.lab_u != 1;
  for (uint32_t idx = tid; idx < total; idx += nthr) {
    const uint32_t col = idx & (ncols - 1), t = idx >> pl.logC;   // ncols = C
    const uint32_t t5 = 5 * t, rho = pfa ? (t5 & (L1 - 1)) : t, w0 = pfa ? (t5 >> pl.logL1) : 0u;   // 5 t = rho + L1 w0: input r sits in block (r + w0) mod 5
    P2 x[5];
    if (!INVERSE) {
#pragma unroll
      for (uint32_t r = 0; r < 5; ++r) { const uint32_t b = r + w0; x[r] = X[(L1 * (b >= 5 ? b - 5 : b) + rho) * ncols + col]; }
      dft5<false>(x, pl.W5c);
      if (!pfa) {
#pragma unroll
        for (int k = 1; k < 5; ++k) x[k] = p2_mul(x[k], pl.UT1[t * k]);  // t*k < M1
      }
#pragma unroll
      for (int r = 0; r < 5; ++r) X[(L1 * r + rho) * ncols + col] = x[r];
    } else {
#pragma unroll
      for (int r = 0; r < 5; ++r) x[r] = X[(L1 * r + rho) * ncols + col];
      if (!pfa) {
#pragma unroll
        for (int k = 1; k < 5; ++k) x[k] = p2_mul(x[k], pl.UT1[(t * k) ? pl.M1 - t * k : 0]);
      }
      dft5<true>(x, pl.W5c);
#pragma unroll
      for (uint32_t r = 0; r < 5; ++r) { const uint32_t b = r + w0; X[(L1 * (b >= 5 ? b - 5 : b) + rho) * ncols + col] = x[r]; }
    }
  }
  __syncthreads();
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

// previous run (in digit order) of run (T, i1): same row, previous tile; first tile wraps to the
// last tile of the previous row (cyclic: 2^p = 1)
template <bool COH = false>
__device__ __forceinline__ uint64_t carry_in_of(const DevPlan& pl, const uint64_t* cbuf, uint32_t T, uint32_t i1) {
  const uint32_t NT = pl.M2 / pl.C;
  if (T > 0) return ld64<COH>(cbuf + size_t(T - 1) * pl.M1 + i1);
  return ld64<COH>(cbuf + size_t(NT - 1) * pl.M1 + (i1 ? i1 - 1 : pl.M1 - 1));
}

// ---------------------------------------------------------------------------------------------
// front: one work-group per tile T (C adjacent columns, all M1 rows)
// ---------------------------------------------------------------------------------------------
// body of the front sweep for tile T (shared by k_front and the one-launch kernel k_coop; no __restrict__ here: k_coop reads and writes
// the same buffers in one launch).  sub: the small subtraction of a Lucas-Lehmer step, applied in the field to digit 0 (weight 1).
template <bool COH>
__device__ __forceinline__ void front_body(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint64_t* Wout, uint32_t T, uint32_t sub,
                                           P2* X, uint32_t tid, uint32_t nthr) {
  const uint32_t C = pl.C, M1 = pl.M1, tile = M1 * C;
  const uint2* dg = reinterpret_cast<const uint2*>(digits) + size_t(T) * tile;

  for (uint32_t e = tid; e < tile; e += nthr) {
    const uint32_t i1 = e >> pl.logC, c = e & (C - 1), i2 = T * C + c;
    uint2 d = ldu2<COH>(dg + e);
    const uint32_t sa = pl.SA[i1];
    if (cbuf_in && c < 2) {
      // deferred run carries (C >= 2): the carry word left by the previous run goes into the first digits
      // of this one (three masked digits, the remainder onto the fourth: adc4, marin.cl:203-212).  The two
      // threads that own the run's first two pairs each redo the four-digit chain and keep their pair.
      uint64_t cin = carry_in_of<COH>(pl, cbuf_in, T, i1);
      if (cin) {
        const uint2 p0 = ldu2<COH>(dg + i1 * C), p1 = ldu2<COH>(dg + i1 * C + 1);
        uint32_t dd[4] = {p0.x, p0.y, p1.x, p1.y};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          uint32_t width; bool wr;
          digit_info(pl, sa, pl.SB[2 * (T * C) + k], width, wr);
          const uint64_t v = uint64_t(dd[k]) + cin;
          dd[k] = uint32_t(v & ((uint64_t(1) << width) - 1));
          cin = v >> width;
        }
        dd[3] += uint32_t(cin);
        d = c ? make_uint2(dd[2], dd[3]) : make_uint2(dd[0], dd[1]);
      }
    }
    // weight = TA * TB / (wrap ? 2 : 1).  The odd digit takes its exponent split from the second half of
    // SA / TA (plan.hpp) so that both digits of the pair share the column factor TB[2 i2]; the halving sits
    // on TA and the digits that do not wrap are doubled instead (digits are < 2^21: still a mul_u32).
    uint32_t w0, w1; bool wr0, wr1;
    const uint32_t sb = pl.SB[2 * i2];
    digit_info(pl, sa, sb, w0, wr0);
    digit_info(pl, pl.SA[M1 + i1], sb, w1, wr1);
    uint64_t a0 = gf::mul_u32(pl.TAh[i1], d.x << (wr0 ? 0 : 1));
    const uint64_t a1 = gf::mul_u32(pl.TAh[M1 + i1], d.y << (wr1 ? 0 : 1));
    if (sub && (T | e) == 0) a0 = gf::sub(a0, uint64_t(sub));
    X[e] = {a0, a1};
  }
  __syncthreads();

  // small tiles (one output element per thread): the three table words of its four-step twiddle are requested before the transform,
  // whose passes hide their latency (latency-bound launches: see lds_pow2_dft_planes)
  const bool one = tile <= nthr;
  uint64_t pre_lo = 0, pre_hi = 0, pre_tb = 0;
  if (one && tid < tile) {
    const uint32_t pos = tid >> pl.logC, i2 = T * C + (tid & (C - 1));
    const uint64_t ex = uint64_t(i2) * freq1(pl, pos);
    pre_lo = pl.TWlo[ex & ((1u << pl.twh) - 1)]; pre_hi = pl.TWhi[ex >> pl.twh]; pre_tb = pl.TB[2 * i2];
  }
  if (pl.r5 == 5) lds_radix5<false>(pl, X, C, tid, nthr);
  if (pl.logL1) lds_pow2_dft<false>(X, pl.L1, pl.logL1, pl.r5, C, pl.logC, pl.UT1, M1, pl.r5, pl.I4, tid, nthr);

  P2* W = reinterpret_cast<P2*>(Wout);
  for (uint32_t e = tid; e < tile; e += nthr) {
    const uint32_t pos = e >> pl.logC, c = e & (C - 1), i2 = T * C + c;
    const uint32_t k1 = freq1(pl, pos);
    const uint32_t ex = i2 * k1;   // i2 < M2, k1 < M1: below m, no reduction needed
    const uint64_t tw = one ? gf::mul(pre_lo, pre_hi) : tw_lookup(pl, ex);
    const P2 x = X[e];
    const uint64_t twb = gf::mul(tw, one ? pre_tb : pl.TB[2 * i2]);
    stp2<COH>(W + size_t(pos) * pl.M2 + i2, P2{gf::mul(x.a, twb), gf::mul(x.b, twb)});
  }
}
__global__ void __launch_bounds__(1024) k_front(DevPlan pl, const uint32_t* __restrict__ digits, const uint64_t* __restrict__ cbuf_in,
                                                uint64_t* __restrict__ Wout) {
  if (blockIdx.x >= pl.boost_tiles) __builtin_amdgcn_s_setprio(3);   // last half round: see kernels_v2.hip, boost_if_late (C4: -2.7 %)
  front_body<false>(pl, digits, cbuf_in, Wout, blockIdx.x, 0, reinterpret_cast<P2*>(smem_raw), threadIdx.x, blockDim.x);
}

// ---------------------------------------------------------------------------------------------
// middle: one work-group per row.  mode 0: square, 1: multiply by image Y, 2: forward only.
// ---------------------------------------------------------------------------------------------
template <bool COH>
__device__ __forceinline__ void middle_body(const DevPlan& pl, const uint64_t* Win, const uint64_t* Yimg, uint64_t* Wout, int mode, uint32_t sub, uint32_t row,
                                            P2* X, uint32_t tid, uint32_t nthr) {
  const uint32_t M2 = pl.M2;
  const P2* in = reinterpret_cast<const P2*>(Win) + size_t(row) * M2;
  P2* out = reinterpret_cast<P2*>(Wout) + size_t(row) * M2;

  for (uint32_t e = tid; e < M2; e += nthr) X[e] = ldp2<COH>(in + e);
  __syncthreads();
  // deferred small subtraction on a front image (digit 0 -> column 0, plane a of every row, weight 1)
  if (sub != 0 && tid == 0) X[0].a = gf::sub(X[0].a, uint64_t(sub));
  __syncthreads();
  const uint32_t k1 = freq1(pl, row);
  const bool one = M2 <= nthr;   // one element per thread: its twiddle words are requested before the transform (see k_front)
  uint64_t pre_lo = 0, pre_hi = 0;
  if (one && tid < M2 && mode != 2) {
    const uint64_t ex = rho_exponent(pl, k1, __brev(tid) >> (32 - pl.logM2));
    pre_lo = pl.TWlo[ex & ((1u << pl.twh) - 1)]; pre_hi = pl.TWhi[ex >> pl.twh];
  }
  lds_pow2_dft<false>(X, M2, pl.logM2, 1, 1, 0, pl.UT2, M2, 1, pl.I4, tid, nthr);
  if (mode == 2) {
    for (uint32_t e = tid; e < M2; e += nthr) stp2<COH>(out + e, X[e]);
    return;
  }
  const P2* Y = reinterpret_cast<const P2*>(Yimg) + size_t(row) * M2;
  for (uint32_t e = tid; e < M2; e += nthr) {
    const uint32_t k2 = __brev(e) >> (32 - pl.logM2);
    const uint64_t rho = one ? gf::mul(pre_lo, pre_hi) : tw_lookup(pl, rho_exponent(pl, k1, k2));
    const P2 u = X[e];
    P2 r;
    if (mode == 0) {  // (u0 + u1 t)^2 mod (t^2 - rho), marin.cl:379-384
      r.a = gf::add(gf::sqr(u.a), gf::mul(gf::sqr(u.b), rho));
      r.b = gf::mul(u.b, gf::dbl(u.a));
    } else {          // marin.cl:387-392
      const P2 y = Y[e];
      r.a = gf::add(gf::mul(u.a, y.a), gf::mul(gf::mul(u.b, y.b), rho));
      r.b = gf::add(gf::mul(u.a, y.b), gf::mul(u.b, y.a));
    }
    X[e] = r;
  }
  __syncthreads();
  lds_pow2_dft<true>(X, M2, pl.logM2, 1, 1, 0, pl.UT2, M2, 1, pl.I4inv, tid, nthr);
  for (uint32_t e = tid; e < M2; e += nthr) stp2<COH>(out + e, X[e]);
}
__global__ void __launch_bounds__(1024) k_middle(DevPlan pl, const uint64_t* __restrict__ Win, const uint64_t* __restrict__ Yimg,
                                                uint64_t* __restrict__ Wout, int mode, uint32_t sub) {
  middle_body<false>(pl, Win, Yimg, Wout, mode, sub, blockIdx.x, reinterpret_cast<P2*>(smem_raw), threadIdx.x, blockDim.x);
}

// ---------------------------------------------------------------------------------------------
// back: inverse of front + unweight + carry over the tile's M1 runs of 2C digits
// ---------------------------------------------------------------------------------------------
// weak carry of a run's incoming carry word into its first digits (adc4, marin.cl:203-212); dd: the run's first four digits
__device__ __forceinline__ void run_carry_in(const DevPlan& pl, uint32_t sa, uint32_t T, uint64_t cin, uint32_t (&dd)[4]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uint32_t width; bool wr;
    digit_info(pl, sa, pl.SB[2 * (T * pl.C) + k], width, wr);
    const uint64_t v = uint64_t(dd[k]) + cin;
    dd[k] = uint32_t(v & ((uint64_t(1) << width) - 1));
    cin = v >> width;
  }
  dd[3] += uint32_t(cin);
}

// blocks that share an XCD (b, b + 8, ...) take neighbouring tiles: the 64-byte pieces of a work-buffer line meet in one L2
// (same order as the register-resident back sweep; C4: 93.3 -> 90.4 us).  MI355_TUNE bit 0 switches it off.
__device__ __forceinline__ uint32_t xcd_tile(const DevPlan& pl, uint32_t b, uint32_t groups) {
  return (!(pl.tune & 1) && groups % 8 == 0) ? (b & 7) * (groups >> 3) + (b >> 3) : b;
}
template <bool EXT, bool COH>
__device__ __forceinline__ void back_body(const DevPlan& pl, const uint64_t* Win, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& ext, uint32_t T,
                                          P2* X, uint32_t tid, uint32_t nthr) {
  const uint32_t C = pl.C, M1 = pl.M1, tile = M1 * C;
  const P2* W = reinterpret_cast<const P2*>(Win);

  for (uint32_t e = tid; e < tile; e += nthr) {
    const uint32_t pos = e >> pl.logC, c = e & (C - 1), i2 = T * C + c;
    const uint32_t k1 = freq1(pl, pos);
    const uint32_t ex = i2 * k1;   // i2 < M2, k1 < M1: below m, no reduction needed
    const uint64_t tw = tw_lookup(pl, ex ? pl.m - ex : 0);
    const P2 x = ldp2<COH>(W + size_t(pos) * pl.M2 + i2);
    const uint64_t twb = gf::mul(tw, pl.TBi[2 * i2]);   // one column factor per pair (see k_front)
    X[e] = {gf::mul(x.a, twb), gf::mul(x.b, twb)};
  }
  __syncthreads();

  if (pl.logL1) lds_pow2_dft<true>(X, pl.L1, pl.logL1, pl.r5, C, pl.logC, pl.UT1, M1, pl.r5, pl.I4inv, tid, nthr);
  if (pl.r5 == 5) lds_radix5<true>(pl, X, C, tid, nthr);

  uint2* dg = reinterpret_cast<uint2*>(digits) + size_t(T) * tile;
  for (uint32_t i1 = tid; i1 < M1; i1 += nthr) {
    const uint32_t sa[2] = {pl.SA[i1], pl.SA[M1 + i1]};          // even / odd digit (second half: see k_front)
    const uint64_t tai[2] = {pl.TAi[i1], pl.TAi[M1 + i1]};
    const uint64_t tai2[2] = {pl.TAi2[i1], pl.TAi2[M1 + i1]};  // wrapped exponents: the weight was halved
    uint64_t carry = 0;
    uint32_t addh[4] = {0, 0, 0, 0};   // first digits of the addend's run with its pending carry folded in (C >= 2)
    const uint2* ad = nullptr;
    if (EXT && ext.add_digits) {
      ad = reinterpret_cast<const uint2*>(ext.add_digits) + size_t(T) * tile + size_t(i1) * C;
      if (ext.add_cbuf) {
        const uint2 p0 = ad[0], p1 = ad[1];
        addh[0] = p0.x; addh[1] = p0.y; addh[2] = p1.x; addh[3] = p1.y;
        run_carry_in(pl, pl.SA[i1], T, carry_in_of(pl, ext.add_cbuf, T, i1), addh);
      }
    }
    for (uint32_t c = 0; c < C; ++c) {
      const uint32_t i2 = T * C + c;
      const P2 x = X[i1 * C + c];
      const uint32_t sb = pl.SB[2 * i2];
      uint32_t out[2];
      uint32_t av[2] = {0, 0};
      if (EXT && ad) {
        if (ext.add_cbuf && c < 2) { av[0] = addh[2 * c]; av[1] = addh[2 * c + 1]; }
        else { const uint2 q = ad[c]; av[0] = q.x; av[1] = q.y; }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        uint32_t width; bool wrap;
        digit_info(pl, sa[b], sb, width, wrap);
        const uint64_t u = gf::mul(b ? x.b : x.a, wrap ? tai2[b] : tai[b]);
        const uint64_t mask = (uint64_t(1) << width) - 1;
        if (a == 1) {               // the common case (uniform): no 64-bit multiplies
          const uint64_t r = u + carry + (EXT ? av[b] : 0u);   // u < 2^63 by the size rule (ibdwt.h:28-30), carry < 2^48, addend < 2^32
          out[b] = uint32_t(r & mask);
          carry = r >> width;
        } else {                    // adc_mul (marin.cl:194-201): digit first, so that everything stays in 64 bits
          const uint64_t dlo = u & mask, chi = u >> width;
          const uint64_t r = dlo * a + carry + (EXT ? av[b] : 0u);
          out[b] = uint32_t(r & mask);
          carry = (r >> width) + chi * a;
        }
      }
      stu2<COH>(dg + i1 * C + c, make_uint2(out[0], out[1]));
      if (EXT && ext.digits2) reinterpret_cast<uint2*>(ext.digits2)[size_t(T) * tile + size_t(i1) * C + c] = make_uint2(out[0], out[1]);
    }
    st64<COH>(cbuf + size_t(T) * M1 + i1, carry);
    if (EXT && ext.digits2) ext.cbuf2[size_t(T) * M1 + i1] = carry;
  }
}
template <bool EXT>
__global__ void __launch_bounds__(1024) k_back(DevPlan pl, const uint64_t* __restrict__ Win, uint32_t* __restrict__ digits,
                                              uint64_t* __restrict__ cbuf, uint32_t a, BackExt ext) {
  if (blockIdx.x >= pl.boost_tiles) __builtin_amdgcn_s_setprio(3);   // last half round: see kernels_v2.hip, boost_if_late (C4: -2.7 %)
  back_body<EXT, false>(pl, Win, digits, cbuf, a, ext, xcd_tile(pl, blockIdx.x, gridDim.x), reinterpret_cast<P2*>(smem_raw), threadIdx.x, blockDim.x);
}

#if defined(MI355_EXPERIMENTAL)   // measured slower than three launches: only in libmi355_engine_exp.so (make exp), not in the product library
// ---------------------------------------------------------------------------------------------
// One launch per squaring (or per run of squarings) for transforms whose tiles all fit on the chip at once: n <= 2^20 words has at
// most 256 column tiles and 512 rows, a launch of the three kernels above lasts as long as ONE tile's dependent stream (5-6 us) plus
// ~2 us to the next launch, and the chip's work is ~1 us per sweep.  k_coop runs front | rows | back of `count` squarings in one
// cooperative launch (hipLaunchCooperativeKernel: every work-group is resident, so the barrier below cannot wait for a group that has
// not started).  Reference: the same chain as three launches, forward1024_0 / sqr512 / backward1024_0, kernels/marin.cl:1190,1517,
// engine_gpu.h:1591.
//
// Grid barrier: one flag word per work-group, written with its epoch once the group's hand-over stores are acknowledged; wave 0 of every
// group polls all flags (one 4-byte agent-scope load per lane and 64 groups).  No read-modify-write on a shared word and no L2
// write-back / invalidate: everything that crosses the barrier is accessed with sc1 (ld64 / st64 above).  A group that waits longer than kCoopTimeoutTicks (100 MHz
// wall clock: 0.2 s) or sees the error word set raises it and leaves; every later barrier of every group then returns at once, so the
// grid drains and the host reports the error (Engine::sync / the next call).
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kCoopTimeoutTicks = 20000000ull;

__device__ __forceinline__ bool grid_sync(uint32_t* flags, uint32_t* err, uint32_t ngroups, uint32_t epoch) {
  __shared__ uint32_t ok_sh;
  // the sweeps' hand-over data travels in sc1 stores (st64<true>): once they are acknowledged they are at the memory side, where the
  // sc1 loads of the next sweep find them -- no L2 maintenance here.  Every wave drains its OWN stores (s_waitcnt vmcnt(0)) before the
  // barrier that precedes the flag store: a workgroup-scope release fence does not wait for vmcnt on gfx950 without tgsplit (ADVICE r03:
  // the ISA had only lgkmcnt(0) on this path, so another XCD could see the epoch before the data had landed); tests/test_isa_hazards.py
  // checks that the wait is in the generated code.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t t0 = wall_clock64();
    bool ok = true;
    uint32_t spins = 0;
    for (;;) {
      bool here = true;
      for (uint32_t g = threadIdx.x; g < ngroups; g += 64)
        here = here && int32_t(__hip_atomic_load(&flags[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
      if (__all(here)) break;
      const bool bad = (++spins & 63u) == 0 && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > kCoopTimeoutTicks);
      if (__any(bad)) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (threadIdx.x == 0) ok_sh = ok ? 1u : 0u;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return ok_sh != 0;
}

struct CoopArgs {
  uint32_t* digits; uint64_t* cbuf; uint64_t* W;     // register (tile-major digits), its run carries, the work buffer
  uint32_t* flags; uint32_t* err;                     // barrier words (one per work-group), error word
  uint32_t a, sub, count, epoch0, carry_in;           // x <- (x^2 - sub?) ... : every squaring is square_mul(a) preceded by the pending "- sub"
  uint32_t sub_next;                                  // subtraction folded into the front sweeps of squarings 2 .. count (Lucas-Lehmer runs)
  uint32_t fault;                                     // test hook (MI355_COOP_FAULT=1): work-group 0 leaves before its first barrier
};

__global__ void __launch_bounds__(1024) k_coop(DevPlan pl, CoopArgs ca) {
  P2* X = reinterpret_cast<P2*>(smem_raw);
  const uint32_t tid = threadIdx.x, nthr = blockDim.x, b = blockIdx.x, G = gridDim.x;
  const uint32_t NT = pl.M2 / pl.C, M1 = pl.M1;
  uint32_t epoch = ca.epoch0;
  if (ca.fault && b == 0) return;   // the others must notice at their barrier, raise the error word and drain (tests/test_gpu_coop.py)
  for (uint32_t it = 0; it < ca.count; ++it) {
    const uint64_t* cin = (it || ca.carry_in) ? ca.cbuf : nullptr;
    const uint32_t sub = it ? ca.sub_next : ca.sub;
    for (uint32_t t = b; t < NT; t += G) { front_body<true>(pl, ca.digits, cin, ca.W, xcd_tile(pl, t, NT), sub, X, tid, nthr); __syncthreads(); }
    if (!grid_sync(ca.flags, ca.err, G, ++epoch)) return;
    for (uint32_t r = b; r < M1; r += G) { middle_body<true>(pl, ca.W, nullptr, ca.W, 0, 0, r, X, tid, nthr); __syncthreads(); }
    if (!grid_sync(ca.flags, ca.err, G, ++epoch)) return;
    for (uint32_t t = b; t < NT; t += G) { back_body<false, true>(pl, ca.W, ca.digits, ca.cbuf, ca.a, BackExt(), xcd_tile(pl, t, NT), X, tid, nthr); __syncthreads(); }
    if (it + 1 < ca.count && !grid_sync(ca.flags, ca.err, G, ++epoch)) return;
  }
}

#endif   // MI355_EXPERIMENTAL

// ---------------------------------------------------------------------------------------------
// Columns too long for LDS: M1 = 5 L1 with 16 M1 bytes above a CU's 160 KiB -- n = 5 * 2^26, the largest entry of the reference's
// schedule (include/marin/engine_gpu.h:1624; forward80_0 ... there).  The radix-5 stage of the column transform runs on its own through a
// second work buffer U ([k0][i2][t]: 5 blocks x M2 columns x L1 pairs), one thread per (column, t) and C = 1; the power-of-two part of
// every block stays LDS-resident (L1 pairs x Cb columns).  Four sweeps instead of two around the row kernel: correct first, this size
// exists for exponents above 3.0e9.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_front_split_a(DevPlan pl, const uint32_t* __restrict__ digits, uint64_t* __restrict__ Uout) {
  const size_t e = size_t(blockIdx.x) * 256 + threadIdx.x;
  const uint32_t t = uint32_t(e & (pl.L1 - 1)), i2 = uint32_t(e >> pl.logL1), M1 = pl.M1;
  if (i2 >= pl.M2) return;
  const uint2* dg = reinterpret_cast<const uint2*>(digits) + size_t(i2) * M1;   // C = 1: tile i2 holds the column's M1 pairs
  const uint32_t sb = pl.SB[2 * i2];
  P2 x[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const uint32_t i1 = pl.L1 * r + t;
    const uint2 d = dg[i1];
    uint32_t w0, w1; bool wr0, wr1;
    digit_info(pl, pl.SA[i1], sb, w0, wr0);
    digit_info(pl, pl.SA[M1 + i1], sb, w1, wr1);
    x[r] = {gf::mul_u32(pl.TAh[i1], d.x << (wr0 ? 0 : 1)), gf::mul_u32(pl.TAh[M1 + i1], d.y << (wr1 ? 0 : 1))};   // see k_front
  }
  dft5<false>(x, pl.W5c);
#pragma unroll
  for (int k = 1; k < 5; ++k) x[k] = p2_mul(x[k], pl.UT1[t * k]);
  P2* U = reinterpret_cast<P2*>(Uout);
#pragma unroll
  for (int k0 = 0; k0 < 5; ++k0) U[(size_t(k0) * pl.M2 + i2) * pl.L1 + t] = x[k0];
}
// block k0, columns Cb T .. Cb T + Cb - 1: L1-point transforms in LDS, four-step twiddle, rows k0 L1 + p of the work buffer
__global__ void __launch_bounds__(1024) k_front_split_b(DevPlan pl, const uint64_t* __restrict__ Uin, uint64_t* __restrict__ Wout, uint32_t logCb) {
  P2* X = reinterpret_cast<P2*>(smem_raw);
  const uint32_t tid = threadIdx.x, nthr = blockDim.x, Cb = 1u << logCb, per = pl.M2 >> logCb;
  const uint32_t k0 = blockIdx.x / per, T = blockIdx.x - k0 * per, tile = pl.L1 << logCb;
  const P2* U = reinterpret_cast<const P2*>(Uin);
  for (uint32_t e = tid; e < tile; e += nthr) {
    const uint32_t c = e >> pl.logL1, pp = e & (pl.L1 - 1);   // reads run along t: contiguous
    X[(pp << logCb) + c] = U[(size_t(k0) * pl.M2 + (T << logCb) + c) * pl.L1 + pp];
  }
  __syncthreads();
  if (pl.logL1) lds_pow2_dft<false>(X, pl.L1, pl.logL1, 1, Cb, logCb, pl.UT1, pl.M1, pl.r5, pl.I4, tid, nthr);
  P2* W = reinterpret_cast<P2*>(Wout);
  for (uint32_t e = tid; e < tile; e += nthr) {
    const uint32_t pp = e >> logCb, c = e & (Cb - 1), i2 = (T << logCb) + c, pos = k0 * pl.L1 + pp;
    const uint64_t ex = uint64_t(i2) * freq1(pl, pos);
    const uint64_t twb = gf::mul(tw_lookup(pl, ex), pl.TB[2 * i2]);
    const P2 x = X[e];
    W[size_t(pos) * pl.M2 + i2] = {gf::mul(x.a, twb), gf::mul(x.b, twb)};
  }
}
__global__ void __launch_bounds__(1024) k_back_split_b(DevPlan pl, const uint64_t* __restrict__ Win, uint64_t* __restrict__ Uout, uint32_t logCb) {
  P2* X = reinterpret_cast<P2*>(smem_raw);
  const uint32_t tid = threadIdx.x, nthr = blockDim.x, Cb = 1u << logCb, per = pl.M2 >> logCb;
  const uint32_t k0 = blockIdx.x / per, T = blockIdx.x - k0 * per, tile = pl.L1 << logCb;
  const P2* W = reinterpret_cast<const P2*>(Win);
  for (uint32_t e = tid; e < tile; e += nthr) {
    const uint32_t pp = e >> logCb, c = e & (Cb - 1), i2 = (T << logCb) + c, pos = k0 * pl.L1 + pp;
    const uint64_t ex = uint64_t(i2) * freq1(pl, pos);
    const uint64_t twb = gf::mul(tw_lookup(pl, ex ? pl.m - ex : 0), pl.TBi[2 * i2]);
    const P2 x = W[size_t(pos) * pl.M2 + i2];
    X[e] = {gf::mul(x.a, twb), gf::mul(x.b, twb)};
  }
  __syncthreads();
  if (pl.logL1) lds_pow2_dft<true>(X, pl.L1, pl.logL1, 1, Cb, logCb, pl.UT1, pl.M1, pl.r5, pl.I4inv, tid, nthr);
  P2* U = reinterpret_cast<P2*>(Uout);
  for (uint32_t e = tid; e < tile; e += nthr) {
    const uint32_t c = e >> pl.logL1, pp = e & (pl.L1 - 1);
    U[(size_t(k0) * pl.M2 + (T << logCb) + c) * pl.L1 + pp] = X[(pp << logCb) + c];
  }
}
// inverse radix-5 stage, unweighting, x a and the carry inside every pair (runs of two digits: the engine follows with k_carry_fix and
// the local carry passes, as for every C = 1 plan)
__global__ void __launch_bounds__(256) k_back_split_a(DevPlan pl, const uint64_t* __restrict__ Uin, uint32_t* __restrict__ digits, uint64_t* __restrict__ cbuf,
                                                     uint32_t a) {
  const size_t e = size_t(blockIdx.x) * 256 + threadIdx.x;
  const uint32_t t = uint32_t(e & (pl.L1 - 1)), i2 = uint32_t(e >> pl.logL1), M1 = pl.M1;
  if (i2 >= pl.M2) return;
  const P2* U = reinterpret_cast<const P2*>(Uin);
  P2 x[5];
#pragma unroll
  for (int k0 = 0; k0 < 5; ++k0) x[k0] = U[(size_t(k0) * pl.M2 + i2) * pl.L1 + t];
#pragma unroll
  for (int k = 1; k < 5; ++k) x[k] = p2_mul(x[k], pl.UT1[(t * k) ? M1 - t * k : 0]);
  dft5<true>(x, pl.W5c);
  uint2* dg = reinterpret_cast<uint2*>(digits) + size_t(i2) * M1;
  const uint32_t sb = pl.SB[2 * i2];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const uint32_t i1 = pl.L1 * r + t;
    const uint32_t sa[2] = {pl.SA[i1], pl.SA[M1 + i1]};
    const uint64_t tai[2] = {pl.TAi[i1], pl.TAi[M1 + i1]};
    uint64_t carry = 0;
    uint32_t out[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      uint32_t width; bool wrap;
      digit_info(pl, sa[b], sb, width, wrap);
      const uint64_t u = gf::mul(b ? x[r].b : x[r].a, wrap ? pl.TAi2[b ? M1 + i1 : i1] : tai[b]);
      const uint64_t mask = (uint64_t(1) << width) - 1;
      const uint64_t dlo = u & mask, chi = u >> width;      // adc_mul (marin.cl:194-201)
      const uint64_t rr = dlo * a + carry;
      out[b] = uint32_t(rr & mask);
      carry = (rr >> width) + chi * a;
    }
    dg[i1] = make_uint2(out[0], out[1]);
    cbuf[size_t(i2) * M1 + i1] = carry;
  }
}

// One thread per run: a, b (+ their pending carry words) -> sum and / or difference, each written to up to two
// registers, with the run's carry-out word left pending (kernels.hpp LinArgs).  The difference is a - b + 2 Mp
// digit by digit (neg2_mp4, marin.cl:246-256), so nothing goes negative: digits are below 2^width + a small excess.
__global__ void __launch_bounds__(256) k_linear(DevPlan pl, LinArgs la) {
  const uint32_t run = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t M1 = pl.M1, C = pl.C, NT = pl.M2 / C;
  if (run >= M1 * NT) return;
  const uint32_t T = run / M1, i1 = run - T * M1;
  const size_t off = (size_t(T) * M1 + i1) * C * 2;
  const uint32_t sa = pl.SA[i1];
  uint64_t ca = la.ca ? carry_in_of(pl, la.ca, T, i1) : 0, cb = la.cb ? carry_in_of(pl, la.cb, T, i1) : 0;
  uint64_t cs = 0, cd = 0;
  const bool want_s = la.s1 != nullptr, want_d = la.d1 != nullptr;
  for (uint32_t k = 0; k < 2 * C; ++k) {
    uint32_t width; bool wrap;
    digit_info(pl, sa, pl.SB[2 * (T * C) + k], width, wrap);
    const uint64_t mask = (uint64_t(1) << width) - 1;
    // inputs with their own (weak) carry-in: three masked digits, the rest stays on the fourth (adc4); runs of
    // two digits (C = 1) take all of it on the second
    uint64_t av = la.a[off + k], bv = la.b[off + k];
    if (k < 3 && k + 1 < 2 * C) { av += ca; ca = av >> width; av &= mask; bv += cb; cb = bv >> width; bv &= mask; }
    else if (k == 3 || k + 1 == 2 * C) { av += ca; ca = 0; bv += cb; cb = 0; }
    if (want_s) {
      const uint64_t v = av + bv + cs;
      const uint32_t o = uint32_t(v & mask);
      cs = v >> width;
      la.s1[off + k] = o;
      if (la.s2) la.s2[off + k] = o;
    }
    if (want_d) {
      const uint64_t v = av + (4 * mask - bv) + cd;   // b's digit may exceed its width by a carry remainder: 4 Mp keeps it positive
      const uint32_t o = uint32_t(v & mask);
      cd = v >> width;
      la.d1[off + k] = o;
      if (la.d2) la.d2[off + k] = o;
    }
  }
  const size_t ci = size_t(T) * M1 + i1;
  if (want_s) { la.cs1[ci] = cs; if (la.s2) la.cs2[ci] = cs; }
  if (want_d) { la.cd1[ci] = cd; if (la.d2) la.cd2[ci] = cd; }
}


// carry fix: one thread per run; weak carry (the remainder, if any, stays on the run's last digit:
// same contract as adc4, marin.cl:203-212)
__global__ void __launch_bounds__(256) k_carry_fix(DevPlan pl, uint32_t* __restrict__ digits, const uint64_t* __restrict__ cbuf) {
  const uint32_t run = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t M1 = pl.M1, C = pl.C, NT = pl.M2 / C;
  if (run >= M1 * NT) return;
  const uint32_t T = run / M1, i1 = run - T * M1;
  uint64_t cin = carry_in_of(pl, cbuf, T, i1);
  if (cin == 0) return;
  uint32_t* d = digits + (size_t(T) * M1 + i1) * C * 2;
  const uint32_t sa = pl.SA[i1];
  for (uint32_t k = 0; k < 2 * C; ++k) {
    if (k == 2 * C - 1) { d[k] += uint32_t(cin); break; }
    uint32_t width; bool wrap;
    digit_info(pl, sa, pl.SB[2 * (T * C) + k], width, wrap);
    const uint64_t v = uint64_t(d[k]) + cin;
    d[k] = uint32_t(v & ((uint64_t(1) << width) - 1));
    cin = v >> width;
    if (cin == 0) break;
  }
}

// dst <- dst + src   (negate = 0)   or   dst <- dst - src + 2*Mp  (negate = 1; neg2_mp4, marin.cl:246-256)
__global__ void __launch_bounds__(256) k_addsub(DevPlan pl, uint32_t* __restrict__ dst, const uint32_t* __restrict__ src,
                                                uint64_t* __restrict__ cbuf, int negate) {
  const uint32_t run = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t M1 = pl.M1, C = pl.C, NT = pl.M2 / C;
  if (run >= M1 * NT) return;
  const uint32_t T = run / M1, i1 = run - T * M1;
  uint32_t* d = dst + (size_t(T) * M1 + i1) * C * 2;
  const uint32_t* s = src + (size_t(T) * M1 + i1) * C * 2;
  const uint32_t sa = pl.SA[i1];
  uint64_t carry = 0;
  for (uint32_t k = 0; k < 2 * C; ++k) {
    uint32_t width; bool wrap;
    digit_info(pl, sa, pl.SB[2 * (T * C) + k], width, wrap);
    const uint64_t mask = (uint64_t(1) << width) - 1;
    uint64_t sv = s[k];
    if (negate) sv = 2 * mask - sv;   // src digits are < 2^width + small: keep it non-negative
    const uint64_t v = uint64_t(d[k]) + sv + carry;
    d[k] = uint32_t(v & mask);
    carry = v >> width;
  }
  cbuf[size_t(T) * M1 + i1] = carry;
}

// digit j -> memory slot (Plan::pos)
__device__ __forceinline__ size_t slot_of(const DevPlan& pl, uint32_t j) {
  const uint32_t i = j >> 1, b = j & 1;
  const uint32_t i1 = i / pl.M2, i2 = i - i1 * pl.M2;
  const uint32_t T = i2 / pl.C, c = i2 - T * pl.C;
  return ((size_t(T) * pl.M1 + i1) * pl.C + c) * 2 + b;
}

// x <- x - a with borrow, one thread (marin.cl:2376-2393)
__global__ void k_sub_small(DevPlan pl, uint32_t* __restrict__ digits, uint32_t a) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  uint64_t borrow = a;
  for (int lap = 0; lap < 3 && borrow; ++lap) {
    for (uint32_t j = 0; j < pl.n && borrow; ++j) {
      const uint32_t i = j >> 1, i1 = i / pl.M2, i2 = i - i1 * pl.M2;
      uint32_t width; bool wrap;
      digit_info(pl, pl.SA[i1], pl.SB[2 * i2 + (j & 1)], width, wrap);
      const size_t s = slot_of(pl, j);
      const uint64_t dv = digits[s];
      if (dv >= borrow) { digits[s] = uint32_t(dv - borrow); borrow = 0; }
      else {
        // borrow k units of 2^width: smallest k with dv + k*2^width >= borrow
        const uint64_t need = borrow - dv;
        const uint64_t k = (need + (uint64_t(1) << width) - 1) >> width;
        digits[s] = uint32_t(dv + (k << width) - borrow);
        borrow = k;
      }
    }
  }
}

// ------------------------------- launch wrappers ---------------------------------------------

static inline uint32_t block_for(size_t work) {
  static const size_t cap = [] { const char* e = getenv("MI355_THREADS"); size_t v = e ? size_t(atoi(e)) : 512; return v < 64 ? 64 : (v > 1024 ? 1024 : v); }();
  size_t b = 64;
  while (b < cap && b < work) b <<= 1;
  return uint32_t(b);
}

// work-group size for a tile of `pairs` pairs: a quarter of it (one radix-4 butterfly per thread and pass), or half of it when the whole
// launch has no more work-groups than the chip has CUs (latency-bound: one plane per thread, lds_pow2_dft_planes)
static inline uint32_t block_for_small(const DevPlan& pl, size_t pairs) {
  const size_t groups = size_t(pl.M1) * pl.M2 / (pairs ? pairs : 1);
  const bool small = groups <= 256 && !(pl.tune & 8);   // at most one work-group per CU (with two per CU the classic form wins: n = 2^21, 0.066 vs 0.073 ms)
  if (!small) return block_for(pairs / 4 ? pairs / 4 : 1);
  // one pair per thread in the element-wise loops where the tile allows (C2: 0.0331 -> 0.0318 ms), else one plane per thread
  const size_t want = (pl.tune & 16) ? pairs / 2 : pairs;
  size_t b = 64;
  while (b < 1024 && b < want) b <<= 1;
  return uint32_t(b);
}

hipError_t launch_front(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint64_t* W, hipStream_t s) {
  const size_t tile = size_t(pl.M1) * pl.C;
  hipLaunchKernelGGL(k_front, dim3(pl.M2 / pl.C), dim3(block_for_small(pl, tile)), tile * 16, s, pl, digits, cbuf_in, W);
  return hipGetLastError();
}
hipError_t launch_middle(const DevPlan& pl, const uint64_t* Win, const uint64_t* Y, uint64_t* Wout, int mode, uint32_t sub, hipStream_t s) {
  hipLaunchKernelGGL(k_middle, dim3(pl.M1), dim3(block_for_small(pl, pl.M2)), size_t(pl.M2) * 16, s, pl, Win, Y, Wout, mode, sub);
  return hipGetLastError();
}
hipError_t launch_back(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, hipStream_t s) {
  const size_t tile = size_t(pl.M1) * pl.C;
  hipLaunchKernelGGL(k_back<false>, dim3(pl.M2 / pl.C), dim3(block_for_small(pl, tile)), tile * 16, s, pl, W, digits, cbuf, a, BackExt());
  return hipGetLastError();
}
hipError_t launch_back_ext(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& x, hipStream_t s) {
  const size_t tile = size_t(pl.M1) * pl.C;
  hipLaunchKernelGGL(k_back<true>, dim3(pl.M2 / pl.C), dim3(block_for_small(pl, tile)), tile * 16, s, pl, W, digits, cbuf, a, x);
  return hipGetLastError();
}
// the split column sweeps (M1 = 5 L1 beyond LDS): U is a second work buffer of 8 n bytes
static uint32_t split_log_cb(const DevPlan& pl) { uint32_t l = 0; while ((size_t(pl.L1) << (l + 1)) * 16 <= size_t(128) * 1024 && (2u << l) <= pl.M2 && l < 2) ++l; return l; }
hipError_t launch_front_split(const DevPlan& pl, const uint32_t* digits, uint64_t* U, uint64_t* W, hipStream_t s) {
  const size_t threads = size_t(pl.M2) * pl.L1;
  hipLaunchKernelGGL(k_front_split_a, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, s, pl, digits, U);
  const uint32_t lcb = split_log_cb(pl);
  const size_t tile = size_t(pl.L1) << lcb;
  hipLaunchKernelGGL(k_front_split_b, dim3(5 * (pl.M2 >> lcb)), dim3(block_for(tile / 4 ? tile / 4 : 1)), tile * 16, s, pl, U, W, lcb);
  return hipGetLastError();
}
hipError_t launch_back_split(const DevPlan& pl, const uint64_t* W, uint64_t* U, uint32_t* digits, uint64_t* cbuf, uint32_t a, hipStream_t s) {
  const uint32_t lcb = split_log_cb(pl);
  const size_t tile = size_t(pl.L1) << lcb;
  hipLaunchKernelGGL(k_back_split_b, dim3(5 * (pl.M2 >> lcb)), dim3(block_for(tile / 4 ? tile / 4 : 1)), tile * 16, s, pl, W, U, lcb);
  const size_t threads = size_t(pl.M2) * pl.L1;
  hipLaunchKernelGGL(k_back_split_a, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, s, pl, U, digits, cbuf, a);
  return hipGetLastError();
}
hipError_t configure_split(const DevPlan& pl) {
  const size_t bytes = (size_t(pl.L1) << split_log_cb(pl)) * 16;
  if (bytes <= 48 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_split_b), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_back_split_b), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
}
#if defined(MI355_EXPERIMENTAL)
// one cooperative launch for `count` squarings: supported when the generic kernels serve the plan with run carries folded into the front
// sweep (C >= 2, no split columns) and a grid of max(tiles, rows) work-groups is resident at once
static inline uint32_t coop_threads(const DevPlan& pl) { return std::max(block_for_small(pl, size_t(pl.M1) * pl.C), block_for_small(pl, pl.M2)); }
static inline size_t coop_lds(const DevPlan& pl) { return std::max(size_t(pl.M1) * pl.C, size_t(pl.M2)) * 16 + 16; }
uint32_t coop_groups(const DevPlan& pl, int device) {
  if (pl.C < 2) return 0;
  const size_t lds = coop_lds(pl);
  if (lds > 160 * 1024) return 0;
  if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_coop), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) return 0;
  int coop = 0, per_cu = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) != hipSuccess || !coop) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_coop, int(coop_threads(pl)), lds) != hipSuccess || per_cu < 1) return 0;
  const uint32_t want = std::max(pl.M2 / pl.C, pl.M1);
  const uint32_t room = uint32_t(per_cu) * uint32_t(prop.multiProcessorCount);
  return want <= room ? want : 0;   // a grid smaller than the tile count would serialise tiles inside a group: not the case this is for
}
hipError_t launch_coop(const DevPlan& pl, uint32_t groups, uint32_t* digits, uint64_t* cbuf, bool carry_in, uint64_t* W, uint32_t a, uint32_t sub, uint32_t sub_next,
                       uint32_t count, uint32_t* flags, uint32_t* err, uint32_t epoch0, uint32_t fault, hipStream_t s) {
  DevPlan plc = pl;
  CoopArgs ca{digits, cbuf, W, flags, err, a, sub, count, epoch0, carry_in ? 1u : 0u, sub_next, fault};
  void* args[2] = {&plc, &ca};
  return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_coop), dim3(groups), dim3(coop_threads(pl)), args, unsigned(coop_lds(pl)), s);
}
#endif   // MI355_EXPERIMENTAL
hipError_t launch_linear(const DevPlan& pl, const LinArgs& la, hipStream_t s) {
  const size_t runs = size_t(pl.M1) * (pl.M2 / pl.C);
  hipLaunchKernelGGL(k_linear, dim3((runs + 255) / 256), dim3(256), 0, s, pl, la);
  return hipGetLastError();
}
hipError_t launch_carry_fix(const DevPlan& pl, uint32_t* digits, const uint64_t* cbuf, hipStream_t s) {
  const size_t runs = size_t(pl.M1) * (pl.M2 / pl.C);
  hipLaunchKernelGGL(k_carry_fix, dim3((runs + 255) / 256), dim3(256), 0, s, pl, digits, cbuf);
  return hipGetLastError();
}
hipError_t launch_addsub(const DevPlan& pl, uint32_t* dst, const uint32_t* src, uint64_t* cbuf, int negate, hipStream_t s) {
  const size_t runs = size_t(pl.M1) * (pl.M2 / pl.C);
  hipLaunchKernelGGL(k_addsub, dim3((runs + 255) / 256), dim3(256), 0, s, pl, dst, src, cbuf, negate);
  return hipGetLastError();
}
hipError_t launch_sub_small(const DevPlan& pl, uint32_t* digits, uint32_t a, hipStream_t s) {
  hipLaunchKernelGGL(k_sub_small, dim3(1), dim3(64), 0, s, pl, digits, a);
  return hipGetLastError();
}
hipError_t configure_kernels(size_t lds_front, size_t lds_mid) {
  hipError_t e = hipSuccess;
  if (lds_front > 48 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_front), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_front));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_back<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_front));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_back<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_front));
    if (e != hipSuccess) return e;
  }
  if (lds_mid > 48 * 1024)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_middle), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_mid));
  return e;
}

}  // namespace mi355
