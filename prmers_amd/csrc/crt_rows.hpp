// Row transforms of the GF(M61^2) x GF(M31^2) squaring, second kernel set: 256 threads hold 2048 complex slots of BOTH fields
// (8 per thread) in three 8-byte LDS planes (re61, im61, (re31, im31)) and run a mixed-radix 8.8.8.{1,2,4} decimation-in-frequency
// in place -- 4 LDS round trips for 1024 points instead of 10, general twiddles on 7/8 of the slots once per radix-8 step instead
// of on half of them every level, and the 8th roots inside a step are (1 +- i) 2^30 resp. (1 +- i) 2^15: an add, a sub and two bit
// rotations (the host rotates its generator so that omega_m^(m/8) is exactly that root).
//   k_cols_fast   columns of the H1 x H2 view of a row (CA = 2048 / H1 adjacent columns per work-group) + four-step twiddle
//   k_mid_fast    rows k1 and H1 - k1 (H2 = 1024 slots each): forward, conjugate-symmetric untangle + square + re-tangle against the
//                 partner row in LDS, inverse -- one launch and one pass over the data for what were five launches
// Frequencies are left in the digit-reversed order of the in-place transform inside a kernel and in natural order in HBM.
// Reference for the algebra: third_party/aevum/src/cl/fft-middle.cl, fftp.cl (the tail square), docs/mersenne2_mixed_crt_2d_half_fast/
// mersenne2_mixed_crt_2d_half_fast.cpp:829-915.  Included by crt_engine.hip only.
#pragma once

namespace mi355 {
namespace crt {

constexpr uint32_t kFastSlots = 2048;
constexpr uint32_t kFastPlane = kFastSlots + kFastSlots / 16;   // skewed by a + a / 16
constexpr uint32_t kFastLdsBytes = kFastPlane * 8 * 3;

struct FastTables {   // per field: omega_L^x for the two pass lengths (x < L), omega_m^(H1 k2) (k2 < H2), omega_m^k (k <= h),
                      // and the two-level table of the four-step twiddles: omega_m^e = lo[e & 1023] * hi[e >> 10] (e < m)
  const F61::C *w1_61, *w2_61, *v61, *u61, *lo61, *hi61;
  const F31::C *w1_31, *w2_31, *v31, *u31, *lo31, *hi31;
};

struct Planes { uint64_t* re; uint64_t* im; uint2* c3; };

__device__ __forceinline__ uint32_t skw(uint32_t a) { return a + (a >> 4); }

template <class F> struct Slot;
template <> struct Slot<F61> {
  static __device__ __forceinline__ F61::C get(const Planes& P, uint32_t a) { const uint32_t s = skw(a); return {P.re[s], P.im[s]}; }
  static __device__ __forceinline__ void put(const Planes& P, uint32_t a, F61::C v) { const uint32_t s = skw(a); P.re[s] = v.re; P.im[s] = v.im; }
};
template <> struct Slot<F31> {
  static __device__ __forceinline__ F31::C get(const Planes& P, uint32_t a) { const uint2 v = P.c3[skw(a)]; return {v.x, v.y}; }
  static __device__ __forceinline__ void put(const Planes& P, uint32_t a, F31::C v) { P.c3[skw(a)] = make_uint2(v.re, v.im); }
};

// multiplication by the 8th root (1 + i) / sqrt 2 and by its conjugate: 1 / sqrt 2 = 2^30 in Z/M61, 2^15 in Z/M31
template <class F> struct Rot8;
template <> struct Rot8<F61> { static __device__ __forceinline__ uint64_t r(uint64_t a) { return rot61(a, 30); } };
template <> struct Rot8<F31> { static __device__ __forceinline__ uint32_t r(uint32_t a) { return rot31(a, 15); } };
template <class F, bool INV>
__device__ __forceinline__ typename F::C mul_w8(typename F::C a) {
  if (!INV) return {Rot8<F>::r(F::sub(a.re, a.im)), Rot8<F>::r(F::add(a.re, a.im))};
  return {Rot8<F>::r(F::add(a.re, a.im)), Rot8<F>::r(F::sub(a.im, a.re))};
}
template <class F, bool INV>
__device__ __forceinline__ typename F::C mul_w4(typename F::C a) { return INV ? cdiv_i<F>(a) : cmul_i<F>(a); }   // omega_4 = i

// ---- Z/M61 without a reduction per operation -------------------------------------------------------------------------------------
// M61 leaves three spare bits in a 64-bit register, exactly what the three levels of a radix-8 butterfly need: with canonical inputs
// (<= M61) the sums are plain 64-bit additions (one v_lshl_add_u64) and a difference is a + (K - b) with K = M61, 2 M61, 4 M61 at the
// three levels, so every intermediate stays <= 8 M61 = 2^64 - 8.  The products fold their operands once ((v & M61) + (v >> 61) <= M61 + 7),
// split them into 31-bit limbs and accumulate both products of a complex component limb-wise in 64-bit multiply-adds
// (re = a c + b (M61 - d): M61 - d is a bit complement of d's limbs), with ONE reduction per component: 16 multiply-adds and two
// reductions per complex product instead of four full multiplications with a reduction each.
constexpr uint64_t K1 = M61, K2 = 2 * M61, K4 = 4 * M61;
__device__ __forceinline__ uint64_t fold61(uint64_t v) { return (v & M61) + (v >> 61); }                    // any v -> <= M61 + 7
__device__ __forceinline__ uint64_t canon61(uint64_t v) { v = fold61(v); return v >= M61 ? v - M61 : v; }  // any v -> [0, M61)
__device__ __forceinline__ uint64_t shl30_61(uint64_t v) { return ((v & 0x7fffffffull) << 30) + (v >> 31); } // v 2^30, any v -> < 2^61 + 2^33
struct Lz61 { uint64_t re, im; };   // lazy complex value; bounds are tracked in the comments of the callers

template <bool INV, uint64_t K>   // x * omega_8 (or its conjugate); components <= K on entry, <= 2 M61 on exit (K <= 2 M61)
__device__ __forceinline__ Lz61 lz_w8(Lz61 a) {
  if (!INV) return {shl30_61(a.re + (K - a.im)), shl30_61(a.re + a.im)};
  return {shl30_61(a.re + a.im), shl30_61(a.im + (K - a.re))};
}
template <bool INV, uint64_t K>   // x * i (or / i); components <= K stay <= K
__device__ __forceinline__ Lz61 lz_w4(Lz61 a) { return INV ? Lz61{a.im, K - a.re} : Lz61{K - a.im, a.re}; }
template <uint64_t K> __device__ __forceinline__ Lz61 lz_add(Lz61 a, Lz61 b) { return {a.re + b.re, a.im + b.im}; }
template <uint64_t K> __device__ __forceinline__ Lz61 lz_sub(Lz61 a, Lz61 b) { return {a.re + (K - b.re), a.im + (K - b.im)}; }   // b <= K

// canonical in (<= M61), lazy out (<= 8 M61 for R = 8, 4 M61 for R = 4, 2 M61 for R = 2)
template <int R, bool INV>
__device__ __forceinline__ void bfly61(Lz61 (&x)[R]) {
  if constexpr (R == 2) {
    const Lz61 a = lz_add<K1>(x[0], x[1]), b = lz_sub<K1>(x[0], x[1]);
    x[0] = a; x[1] = b;
  } else if constexpr (R == 4) {
    const Lz61 a0 = lz_add<K1>(x[0], x[2]), a1 = lz_add<K1>(x[1], x[3]), b0 = lz_sub<K1>(x[0], x[2]), b1 = lz_w4<INV, K2>(lz_sub<K1>(x[1], x[3]));
    x[0] = lz_add<K2>(a0, a1); x[2] = lz_sub<K2>(a0, a1); x[1] = lz_add<K2>(b0, b1); x[3] = lz_sub<K2>(b0, b1);
  } else {
    const Lz61 a0 = lz_add<K1>(x[0], x[4]), a1 = lz_add<K1>(x[1], x[5]), a2 = lz_add<K1>(x[2], x[6]), a3 = lz_add<K1>(x[3], x[7]);           // <= 2 M61
    const Lz61 b0 = lz_sub<K1>(x[0], x[4]), b1 = lz_w8<INV, K2>(lz_sub<K1>(x[1], x[5])), b2 = lz_w4<INV, K2>(lz_sub<K1>(x[2], x[6])),
              b3 = lz_w4<INV, K2>(lz_w8<INV, K2>(lz_sub<K1>(x[3], x[7])));                                                                  // <= 2 M61
    const Lz61 c0 = lz_add<K2>(a0, a2), c1 = lz_add<K2>(a1, a3), d0 = lz_sub<K2>(a0, a2), d1 = lz_w4<INV, K4>(lz_sub<K2>(a1, a3));            // <= 4 M61
    const Lz61 e0 = lz_add<K2>(b0, b2), e1 = lz_add<K2>(b1, b3), f0 = lz_sub<K2>(b0, b2), f1 = lz_w4<INV, K4>(lz_sub<K2>(b1, b3));
    x[0] = lz_add<K4>(c0, c1); x[4] = lz_sub<K4>(c0, c1); x[2] = lz_add<K4>(d0, d1); x[6] = lz_sub<K4>(d0, d1);                               // <= 8 M61
    x[1] = lz_add<K4>(e0, e1); x[5] = lz_sub<K4>(e0, e1); x[3] = lz_add<K4>(f0, f1); x[7] = lz_sub<K4>(f0, f1);
  }
}

// (a + i b)(c + i d) or, CONJ, (a + i b)(c - i d): a, b <= M61 + 7 (folded), c, d canonical; canonical result
template <bool CONJ>
__device__ __forceinline__ F61::C cmul61(Lz61 x, F61::C w) {
  const uint32_t a0 = uint32_t(x.re) & 0x7fffffffu, a1 = uint32_t(x.re >> 31), b0 = uint32_t(x.im) & 0x7fffffffu, b1 = uint32_t(x.im >> 31);
  const uint32_t c0 = uint32_t(w.re) & 0x7fffffffu, c1 = uint32_t(w.re >> 31);
  uint32_t d0 = uint32_t(w.im) & 0x7fffffffu, d1 = uint32_t(w.im >> 31);
  uint32_t n0 = d0 ^ 0x7fffffffu, n1 = d1 ^ 0x3fffffffu;                        // limbs of M61 - d
  if (CONJ) { uint32_t t = d0; d0 = n0; n0 = t; t = d1; d1 = n1; n1 = t; }
  const uint64_t P0 = uint64_t(a0) * c0 + uint64_t(b0) * n0;                      // < 2^63
  const uint64_t P1 = uint64_t(a0) * c1 + uint64_t(a1) * c0 + uint64_t(b0) * n1 + uint64_t(b1) * n0;   // < 2^63
  const uint64_t P2 = uint64_t(a1) * c1 + uint64_t(b1) * n1;                      // <= 2^61
  const uint64_t Q0 = uint64_t(a0) * d0 + uint64_t(b0) * c0;
  const uint64_t Q1 = uint64_t(a0) * d1 + uint64_t(a1) * d0 + uint64_t(b0) * c1 + uint64_t(b1) * c0;
  const uint64_t Q2 = uint64_t(a1) * d1 + uint64_t(b1) * c1;
  // P0 + P1 2^31 + P2 2^62 with 2^61 = 1: P1 = l + h 2^30 -> l 2^31 + h; P2 2^62 -> 2 P2; the sum stays below 2^64
  const uint64_t S = P0 + (P2 << 1) + (P1 >> 30) + (uint64_t(uint32_t(P1) & 0x3fffffffu) << 31);
  const uint64_t T = Q0 + (Q2 << 1) + (Q1 >> 30) + (uint64_t(uint32_t(Q1) & 0x3fffffffu) << 31);
  return {canon61(S), canon61(T)};
}

// Z/M31[i] product with both partial products of a component accumulated in one 64-bit multiply-add chain (a c + b (M31 - d) < 2^63)
// and one reduction per component (canonical operands and result)
__device__ __forceinline__ uint32_t red31_63(uint64_t x) {                       // x < 2^63
  const uint32_t y = (uint32_t(x) & M31) + (uint32_t(x >> 31) & M31) + uint32_t(x >> 62);   // <= 2^32 - 1
  const uint32_t z = (y & M31) + (y >> 31);                                       // <= M31 + 1
  return min(z, z - M31);                                                         // unsigned wrap: z < M31 keeps z
}
template <bool CONJ>
__device__ __forceinline__ F31::C cmul31(F31::C x, F31::C w) {
  const uint32_t d = CONJ ? (w.im ^ M31) : w.im, n = CONJ ? w.im : (w.im ^ M31);  // M31 - d is the bit complement
  return {red31_63(uint64_t(x.re) * w.re + uint64_t(x.im) * n), red31_63(uint64_t(x.re) * d + uint64_t(x.im) * w.re)};
}

// out[k] = sum_q in[q] w^(qk), w = omega_R (forward) or its conjugate (INV, unnormalised); natural order in and out
template <class F, int R, bool INV>
__device__ __forceinline__ void bfly(typename F::C (&x)[R]) {
  using C = typename F::C;
  if constexpr (R == 2) {
    const C a = cadd<F>(x[0], x[1]), b = csub<F>(x[0], x[1]);
    x[0] = a; x[1] = b;
  } else if constexpr (R == 4) {
    const C a0 = cadd<F>(x[0], x[2]), a1 = cadd<F>(x[1], x[3]), b0 = csub<F>(x[0], x[2]), b1 = mul_w4<F, INV>(csub<F>(x[1], x[3]));
    x[0] = cadd<F>(a0, a1); x[2] = csub<F>(a0, a1); x[1] = cadd<F>(b0, b1); x[3] = csub<F>(b0, b1);
  } else {
    const C a0 = cadd<F>(x[0], x[4]), a1 = cadd<F>(x[1], x[5]), a2 = cadd<F>(x[2], x[6]), a3 = cadd<F>(x[3], x[7]);
    const C b0 = csub<F>(x[0], x[4]), b1 = mul_w8<F, INV>(csub<F>(x[1], x[5])), b2 = mul_w4<F, INV>(csub<F>(x[2], x[6])),
            b3 = mul_w4<F, INV>(mul_w8<F, INV>(csub<F>(x[3], x[7])));
    const C c0 = cadd<F>(a0, a2), c1 = cadd<F>(a1, a3), d0 = csub<F>(a0, a2), d1 = mul_w4<F, INV>(csub<F>(a1, a3));
    const C e0 = cadd<F>(b0, b2), e1 = cadd<F>(b1, b3), f0 = csub<F>(b0, b2), f1 = mul_w4<F, INV>(csub<F>(b1, b3));
    x[0] = cadd<F>(c0, c1); x[4] = csub<F>(c0, c1); x[2] = cadd<F>(d0, d1); x[6] = csub<F>(d0, d1);
    x[1] = cadd<F>(e0, e1); x[5] = csub<F>(e0, e1); x[3] = cadd<F>(f0, f1); x[7] = csub<F>(f0, f1);
  }
}

// one in-place step of radix R = 2^LR on sub-transforms of size 2^logS inside transforms of size 2^logL (2048 slots per work-group):
// forward: butterfly, then y_k *= omega_S^(jk); inverse: x_k *= conj(omega_S^(jk)), then the conjugate butterfly.  WL[x] = omega_L^x.
template <class F, int LR, bool INV>
__device__ __forceinline__ void step(const Planes& P, uint32_t tid, uint32_t logL, uint32_t logS, const typename F::C* __restrict__ WL) {
  using C = typename F::C;
  constexpr int R = 1 << LR, G = 8 / R;
  const uint32_t logSr = logS - LR;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const uint32_t bf = tid * G + g;
    const uint32_t c = bf >> (logL - LR), bfl = bf & ((1u << (logL - LR)) - 1);
    const uint32_t j = bfl & ((1u << logSr) - 1), blk = bfl >> logSr;
    const uint32_t base = (c << logL) + (blk << logS) + j;
    C x[R];
#pragma unroll
    for (int q = 0; q < R; ++q) x[q] = Slot<F>::get(P, base + (uint32_t(q) << logSr));
    if (INV && logSr) {
#pragma unroll
      for (int k = 1; k < R; ++k) x[k] = cmul31<true>(x[k], WL[(j * uint32_t(k)) << (logL - logS)]);
    }
    bfly<F, R, INV>(x);
    if (!INV && logSr) {
#pragma unroll
      for (int k = 1; k < R; ++k) x[k] = cmul31<false>(x[k], WL[(j * uint32_t(k)) << (logL - logS)]);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) Slot<F>::put(P, base + (uint32_t(q) << logSr), x[q]);
  }
}

// the same step for Z/M61[i] on the lazy forms above: canonical in LDS, one canonicalisation per value and step
template <int LR, bool INV>
__device__ __forceinline__ void step61(const Planes& P, uint32_t tid, uint32_t logL, uint32_t logS, const F61::C* __restrict__ WL) {
  constexpr int R = 1 << LR, G = 8 / R;
  const uint32_t logSr = logS - LR;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const uint32_t bf = tid * G + g;
    const uint32_t c = bf >> (logL - LR), bfl = bf & ((1u << (logL - LR)) - 1);
    const uint32_t j = bfl & ((1u << logSr) - 1), blk = bfl >> logSr;
    const uint32_t base = (c << logL) + (blk << logS) + j;
    Lz61 x[R];
#pragma unroll
    for (int q = 0; q < R; ++q) { const F61::C v = Slot<F61>::get(P, base + (uint32_t(q) << logSr)); x[q] = {v.re, v.im}; }
    if (INV && logSr) {
#pragma unroll
      for (int k = 1; k < R; ++k) { const F61::C v = cmul61<true>(x[k], WL[(j * uint32_t(k)) << (logL - logS)]); x[k] = {v.re, v.im}; }
    }
    bfly61<R, INV>(x);
#pragma unroll
    for (int q = 0; q < R; ++q) {
      F61::C v;
      if (!INV && logSr && q) v = cmul61<false>(Lz61{fold61(x[q].re), fold61(x[q].im)}, WL[(j * uint32_t(q)) << (logL - logS)]);
      else v = {canon61(x[q].re), canon61(x[q].im)};
      Slot<F61>::put(P, base + (uint32_t(q) << logSr), v);
    }
  }
}

// radices of a transform of length 2^logL: the odd bits first (radix 2 or 4), then radix 8 -- the last step needs no twiddles, so it
// should be a wide one (1024 points: 2.8.8.8 multiplies 2.25 slots in 8 by a table twiddle per direction, 8.8.8.2 would 2.6)
__device__ __forceinline__ uint32_t step_bits(uint32_t logS) { return (logS % 3u) ? (logS % 3u) : 3u; }

template <bool INV>
__device__ __forceinline__ void transform_both(const Planes& P, uint32_t tid, uint32_t logL, const F61::C* __restrict__ W61, const F31::C* __restrict__ W31) {
  // forward: sizes logL, logL - 3, ...; inverse: the same steps in reverse order
  uint32_t sizes[4]; int ns = 0;
  for (uint32_t logS = logL; logS; logS -= step_bits(logS)) sizes[ns++] = logS;
  for (int t = 0; t < ns; ++t) {
    const uint32_t logS = sizes[INV ? ns - 1 - t : t];
    const uint32_t lr = step_bits(logS);
    __syncthreads();
    if (lr == 3) { step61<3, INV>(P, tid, logL, logS, W61); step<F31, 3, INV>(P, tid, logL, logS, W31); }
    else if (lr == 2) { step61<2, INV>(P, tid, logL, logS, W61); step<F31, 2, INV>(P, tid, logL, logS, W31); }
    else { step61<1, INV>(P, tid, logL, logS, W61); step<F31, 1, INV>(P, tid, logL, logS, W31); }
  }
  __syncthreads();
}

// position of frequency k after the in-place forward transform (mixed-radix digit reversal), and its inverse
__device__ __forceinline__ uint32_t pos_of_freq(uint32_t k, uint32_t logL) {
  uint32_t p = 0;
  for (uint32_t logS = logL; logS;) { const uint32_t lr = step_bits(logS); logS -= lr; p += (k & ((1u << lr) - 1)) << logS; k >>= lr; }
  return p;
}
__device__ __forceinline__ uint32_t freq_of_pos(uint32_t p, uint32_t logL) {
  uint32_t k = 0, sh = 0;
  for (uint32_t logS = logL; logS;) { const uint32_t lr = step_bits(logS); logS -= lr; k += ((p >> logS) & ((1u << lr) - 1)) << sh; sh += lr; }
  return k;
}

__device__ __forceinline__ Planes planes_of(unsigned char* smem) {
  Planes P;
  P.re = reinterpret_cast<uint64_t*>(smem); P.im = P.re + kFastPlane; P.c3 = reinterpret_cast<uint2*>(P.im + kFastPlane);
  return P;
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_crt[];

// Work-group b runs on XCD b mod 8 (round-robin dispatch), each XCD with its own L2.  Adjacent column tiles share 128-byte lines on the
// strided side of the four-step (CA slots of 16 or 8 bytes per row of the view), so XCD x takes the x-th contiguous eighth of the tiles:
// the groups b, b + 8, b + 16, ... of one XCD are neighbours in memory and in time, and both halves (all quarters) of a line meet in one L2.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t groups, uint32_t tune) {   // MI355_CRT_TUNE bit 0: plain order (A/B runs)
  return ((groups & 7u) || (tune & 1u)) ? b : (b & 7u) * (groups >> 3) + (b >> 3);
}

// ---- columns ----
template <bool INV>
__global__ void __launch_bounds__(256) k_cols_fast(Grid gr, FastTables T, F61::C* __restrict__ Z61, F31::C* __restrict__ Z31) {
  const Planes P = planes_of(smem_crt);
  const uint32_t tid = threadIdx.x, logL = gr.logH1, H2 = 1u << gr.logH2;
  const uint32_t CA = kFastSlots >> logL, per_row = H2 / CA;
  const uint32_t bid = xcd_tile(blockIdx.x, gridDim.x, gr.tune);
  const uint32_t row = bid / per_row, col0 = (bid - row * per_row) * CA;
  F61::C* z61 = Z61 + size_t(row) * gr.h; F31::C* z31 = Z31 + size_t(row) * gr.h;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t e = tid + 256u * it, c = e % CA, i = e / CA;        // i: row index i1 (forward) or frequency k1 (inverse)
    const size_t addr = size_t(i) * H2 + col0 + c;
    F61::C a = z61[addr]; F31::C b = z31[addr];
    uint32_t slot = (c << logL) + i;
    if (INV) {   // conj(omega_h^(k1 i2)) from the two small tables (a gather from the full table costs two 128-byte lines a slot)
      const uint32_t tw = 2u * i * (col0 + c);
      a = cmul61<true>(Lz61{a.re, a.im}, T.lo61[tw & 1023]); a = cmul61<true>(Lz61{a.re, a.im}, T.hi61[tw >> 10]);
      b = cmul31<true>(cmul31<true>(b, T.lo31[tw & 1023]), T.hi31[tw >> 10]);
      slot = (c << logL) + pos_of_freq(i, logL);
    }
    Slot<F61>::put(P, slot, a); Slot<F31>::put(P, slot, b);
  }
  transform_both<INV>(P, tid, logL, T.w1_61, T.w1_31);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t e = tid + 256u * it, c = e % CA, i = e / CA;        // i: frequency k1 (forward) or row index i1 (inverse)
    const uint32_t slot = (c << logL) + (INV ? i : pos_of_freq(i, logL));
    F61::C a = Slot<F61>::get(P, slot); F31::C b = Slot<F31>::get(P, slot);
    if (!INV) {
      const uint32_t tw = 2u * i * (col0 + c);
      a = cmul61<false>(Lz61{a.re, a.im}, T.lo61[tw & 1023]); a = cmul61<false>(Lz61{a.re, a.im}, T.hi61[tw >> 10]);
      b = cmul31<false>(cmul31<false>(b, T.lo31[tw & 1023]), T.hi31[tw >> 10]);
    }
    const size_t addr = size_t(i) * H2 + col0 + c;
    z61[addr] = a; z31[addr] = b;
  }
}

// ---- columns, one field per launch: twice the columns per work-group for Z/M61 (4096 slots, 512 threads: 128-byte row segments at
// H1 = 512) and four times for Z/M31 (8192 slots, 1024 threads), i.e. no half-used cache lines on the strided side of the four-step
template <class F> struct FieldOps;
template <> struct FieldOps<F61> {
  static __device__ __forceinline__ F61::C tw(F61::C a, F61::C lo, F61::C hi, bool conj) {
    if (conj) { a = cmul61<true>(Lz61{a.re, a.im}, lo); return cmul61<true>(Lz61{a.re, a.im}, hi); }
    a = cmul61<false>(Lz61{a.re, a.im}, lo); return cmul61<false>(Lz61{a.re, a.im}, hi);
  }
  template <int LR, bool INV> static __device__ __forceinline__ void step(const Planes& P, uint32_t tid, uint32_t logL, uint32_t logS, const F61::C* W) {
    step61<LR, INV>(P, tid, logL, logS, W);
  }
};
template <> struct FieldOps<F31> {
  static __device__ __forceinline__ F31::C tw(F31::C a, F31::C lo, F31::C hi, bool conj) {
    return conj ? cmul31<true>(cmul31<true>(a, lo), hi) : cmul31<false>(cmul31<false>(a, lo), hi);
  }
  template <int LR, bool INV> static __device__ __forceinline__ void step(const Planes& P, uint32_t tid, uint32_t logL, uint32_t logS, const F31::C* W) {
    crt::step<F31, LR, INV>(P, tid, logL, logS, W);
  }
};
template <class F, bool INV>
__device__ __forceinline__ void transform_one(const Planes& P, uint32_t tid, uint32_t logL, const typename F::C* __restrict__ W) {
  uint32_t sizes[4]; int ns = 0;
  for (uint32_t logS = logL; logS; logS -= step_bits(logS)) sizes[ns++] = logS;
  for (int t = 0; t < ns; ++t) {
    const uint32_t logS = sizes[INV ? ns - 1 - t : t];
    const uint32_t lr = step_bits(logS);
    __syncthreads();
    if (lr == 3) FieldOps<F>::template step<3, INV>(P, tid, logL, logS, W);
    else if (lr == 2) FieldOps<F>::template step<2, INV>(P, tid, logL, logS, W);
    else FieldOps<F>::template step<1, INV>(P, tid, logL, logS, W);
  }
  __syncthreads();
}

template <class F, bool INV, uint32_t SLOTS>
__global__ void __launch_bounds__(SLOTS / 8) k_cols_one(Grid gr, const typename F::C* __restrict__ W1, const typename F::C* __restrict__ LO,
                                                        const typename F::C* __restrict__ HI, typename F::C* __restrict__ Z) {
  using C = typename F::C;
  constexpr uint32_t NT = SLOTS / 8, PLANE = SLOTS + SLOTS / 16;
  Planes P;
  P.re = reinterpret_cast<uint64_t*>(smem_crt); P.im = P.re + PLANE; P.c3 = reinterpret_cast<uint2*>(smem_crt);   // one field: planes may overlap in name only
  const uint32_t tid = threadIdx.x, logL = gr.logH1, H2 = 1u << gr.logH2;
  const uint32_t CA = SLOTS >> logL, per_row = H2 / CA;
  const uint32_t bid = xcd_tile(blockIdx.x, gridDim.x, gr.tune);
  const uint32_t row = bid / per_row, col0 = (bid - row * per_row) * CA;
  C* z = Z + size_t(row) * gr.h;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t e = tid + NT * it, c = e % CA, i = e / CA;
    C a = z[size_t(i) * H2 + col0 + c];
    uint32_t slot = (c << logL) + i;
    if (INV) {
      const uint32_t tw = 2u * i * (col0 + c);
      a = FieldOps<F>::tw(a, LO[tw & 1023], HI[tw >> 10], true);
      slot = (c << logL) + pos_of_freq(i, logL);
    }
    Slot<F>::put(P, slot, a);
  }
  transform_one<F, INV>(P, tid, logL, W1);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t e = tid + NT * it, c = e % CA, i = e / CA;
    C a = Slot<F>::get(P, (c << logL) + (INV ? i : pos_of_freq(i, logL)));
    if (!INV) {
      const uint32_t tw = 2u * i * (col0 + c);
      a = FieldOps<F>::tw(a, LO[tw & 1023], HI[tw >> 10], false);
    }
    z[size_t(i) * H2 + col0 + c] = a;
  }
}

// ---- middle: rows k1 and H1 - k1 (work-group 0 of a grid row: rows 0 and H1 / 2, each its own partner) ----
// The conjugate-symmetric untangle / square / re-tangle (crt_engine.hip: spectrum_sq, repack) per field: Z/M31[i] on the generic forms
// with the fused product, Z/M61[i] on the lazy forms (sums of up to three canonical values, one fold before a product).
struct Pw31 {
  using C = F31::C;
  static __device__ __forceinline__ C mul(C a, C w) { return cmul31<false>(a, w); }
  static __device__ __forceinline__ C mulc(C a, C w) { return cmul31<true>(a, w); }
  static __device__ __forceinline__ C lin(C zk, C zmk, C w) {          // X_k of the real sequence from the packed spectrum
    const C zc = cconj<F31>(zmk);
    const C e = cadd<F31>(zk, zc), o = cdiv_i<F31>(csub<F31>(zk, zc));
    return chalf<F31>(cadd<F31>(e, mul(o, w)));
  }
  static __device__ __forceinline__ C spectrum(C zk, C zmk, C w) { const C x = lin(zk, zmk, w); return mul(x, x); }
  static __device__ __forceinline__ C repack(C yk, C yhk, C w) {
    const C yc = cconj<F31>(yhk);
    const C e = cadd<F31>(yk, yc), d = mulc(csub<F31>(yk, yc), w);
    return chalf<F31>(cadd<F31>(e, cmul_i<F31>(d)));
  }
};
__device__ __forceinline__ uint64_t half61(uint64_t v) { return (v & 1) ? (v + M61) >> 1 : v >> 1; }   // v <= 3 M61 -> <= 2 M61
struct Pw61 {
  using C = F61::C;
  static __device__ __forceinline__ C mul(C a, C w) { return cmul61<false>(Lz61{a.re, a.im}, w); }
  static __device__ __forceinline__ C lin(C zk, C zmk, C w) {
    // e = zk + conj(zmk), o = (zk - conj(zmk)) / i = (d.im, -d.re); all <= 2 M61
    const Lz61 e{zk.re + zmk.re, zk.im + (K1 - zmk.im)};
    const Lz61 o{zk.im + zmk.im, K2 - (zk.re + (K1 - zmk.re))};
    const C t = cmul61<false>(Lz61{fold61(o.re), fold61(o.im)}, w);
    return {canon61(half61(e.re + t.re)), canon61(half61(e.im + t.im))};
  }
  static __device__ __forceinline__ C spectrum(C zk, C zmk, C w) { const C x = lin(zk, zmk, w); return mul(x, x); }
  static __device__ __forceinline__ C repack(C yk, C yhk, C w) {
    const Lz61 e{yk.re + yhk.re, yk.im + (K1 - yhk.im)};
    const Lz61 df{fold61(yk.re + (K1 - yhk.re)), fold61(yk.im + yhk.im)};
    const C d = cmul61<true>(df, w);
    return {canon61(half61(e.re + (K1 - d.im))), canon61(half61(e.im + d.re))};   // (e + i d) / 2
  }
};
template <class F> struct PwOf;
template <> struct PwOf<F61> { using T = Pw61; };
template <> struct PwOf<F31> { using T = Pw31; };

// the stored spectrum of a multiplicand for the two rows of a work-group (a == nullptr: square instead): slot x < L is row a, L + x row b
template <class F> struct Img {
  const typename F::C* a; const typename F::C* b; uint32_t L;
  __device__ __forceinline__ explicit operator bool() const { return a != nullptr; }
  __device__ __forceinline__ typename F::C operator[](uint32_t slot) const { return slot < L ? a[slot] : b[slot - L]; }
};
template <class F>
__device__ __forceinline__ void pointwise_pair(const Planes& P, uint32_t sa, uint32_t sb, typename F::C wa, Img<F> img) {
  using C = typename F::C;
  using PW = typename PwOf<F>::T;
  const C za = Slot<F>::get(P, sa), zb = Slot<F>::get(P, sb);
  const C wb = cneg<F>(cconj<F>(wa));                                   // omega_m^(h - k) = -conj(omega_m^k)
  C ya, yb;
  if (img) {
    const C ia = img[sa], ib = img[sb];
    ya = PW::mul(PW::lin(za, zb, wa), PW::lin(ia, ib, wa)); yb = PW::mul(PW::lin(zb, za, wb), PW::lin(ib, ia, wb));
  } else { ya = PW::spectrum(za, zb, wa); yb = PW::spectrum(zb, za, wb); }
  Slot<F>::put(P, sa, PW::repack(ya, yb, wa));
  Slot<F>::put(P, sb, PW::repack(yb, ya, wb));
}
template <class F>
__device__ __forceinline__ void pointwise_self(const Planes& P, uint32_t s, typename F::C w, Img<F> img) {   // k = h / 2
  using PW = typename PwOf<F>::T;
  const typename F::C z = Slot<F>::get(P, s);
  typename F::C y;
  if (img) { const typename F::C i = img[s]; y = PW::mul(PW::lin(z, z, w), PW::lin(i, i, w)); } else y = PW::spectrum(z, z, w);
  Slot<F>::put(P, s, PW::repack(y, y, w));
}
template <class F>
__device__ __forceinline__ void pointwise_zero(const Planes& P, uint32_t s, Img<F> img) {   // k = 0 with k = h folded in
  using C = typename F::C;
  using PW = typename PwOf<F>::T;
  const C z = Slot<F>::get(P, s);
  const C one{1, 0}, mone{F::M - 1, 0};
  C y0, yh;
  if (img) { const C i = img[s]; y0 = PW::mul(PW::lin(z, z, one), PW::lin(i, i, one)); yh = PW::mul(PW::lin(z, z, mone), PW::lin(i, i, mone)); }
  else { y0 = PW::spectrum(z, z, one); yh = PW::spectrum(z, z, mone); }
  Slot<F>::put(P, s, PW::repack(y0, yh, one));
}

// MODE 0: square in place; 1: forward only, the packed spectrum (in the in-place order of the transform) goes to I61 / I31 (set_multiplicand);
// 2: multiply by the spectrum stored in I61 / I31
template <int MODE>
__global__ void __launch_bounds__(256) k_mid_fast(Grid gr, FastTables T, F61::C* __restrict__ Z61, F31::C* __restrict__ Z31, F61::C* __restrict__ I61,
                                                  F31::C* __restrict__ I31) {
  const Planes P = planes_of(smem_crt);
  const uint32_t tid = threadIdx.x, logL = gr.logH2, L = 1u << logL, H1 = 1u << gr.logH1;   // L = 1024: two rows per work-group
  const uint32_t per_row = H1 >> 1;
  const uint32_t row = blockIdx.x / per_row, b = blockIdx.x - row * per_row;
  const uint32_t k1a = b, k1b = b ? H1 - b : (H1 >> 1);
  F61::C* z61 = Z61 + size_t(row) * gr.h; F31::C* z31 = Z31 + size_t(row) * gr.h;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t e = tid + 256u * it, r = e >> logL, i = e & (L - 1);
    const size_t addr = size_t(r ? k1b : k1a) * L + i;
    Slot<F61>::put(P, e, z61[addr]); Slot<F31>::put(P, e, z31[addr]);
  }
  transform_both<false>(P, tid, logL, T.w2_61, T.w2_31);
  if (MODE == 1) {
    F61::C* i61 = I61 + size_t(row) * gr.h; F31::C* i31 = I31 + size_t(row) * gr.h;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const uint32_t e = tid + 256u * it, r = e >> logL, i = e & (L - 1);
      const size_t addr = size_t(r ? k1b : k1a) * L + i;
      i61[addr] = Slot<F61>::get(P, e); i31[addr] = Slot<F31>::get(P, e);
    }
    return;
  }
  Img<F61> g61{nullptr, nullptr, L}; Img<F31> g31{nullptr, nullptr, L};
  if (MODE == 2) {
    g61.a = I61 + size_t(row) * gr.h + size_t(k1a) * L; g61.b = I61 + size_t(row) * gr.h + size_t(k1b) * L;
    g31.a = I31 + size_t(row) * gr.h + size_t(k1a) * L; g31.b = I31 + size_t(row) * gr.h + size_t(k1b) * L;
  }
  // pointwise: 1024 pairs (4 per thread)
  const F61::C ua61 = T.u61[k1a], ub61 = T.u61[k1b]; const F31::C ua31 = T.u31[k1a], ub31 = T.u31[k1b];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const uint32_t q = tid + 256u * it;                                 // 0 .. 1023
    if (b) {                                                            // (row a, k2 = q) <-> (row b, k2 = L - 1 - q)
      const uint32_t sa = pos_of_freq(q, logL), sb = L + pos_of_freq(L - 1 - q, logL);
      pointwise_pair<F61>(P, sa, sb, cmul61<false>(Lz61{ua61.re, ua61.im}, T.v61[q]), g61);
      pointwise_pair<F31>(P, sa, sb, cmul31<false>(ua31, T.v31[q]), g31);
    } else if (q < (L >> 1)) {                                          // row H1/2: k2 = q <-> L - 1 - q
      const uint32_t sa = L + pos_of_freq(q, logL), sb = L + pos_of_freq(L - 1 - q, logL);
      pointwise_pair<F61>(P, sa, sb, cmul61<false>(Lz61{ub61.re, ub61.im}, T.v61[q]), g61);
      pointwise_pair<F31>(P, sa, sb, cmul31<false>(ub31, T.v31[q]), g31);
    } else {                                                            // row 0: k2 = q' <-> L - q'
      const uint32_t qq = q - (L >> 1);
      if (qq == 0) {
        pointwise_zero<F61>(P, 0, g61); pointwise_zero<F31>(P, 0, g31);
        const uint32_t sm = pos_of_freq(L >> 1, logL);
        pointwise_self<F61>(P, sm, T.v61[L >> 1], g61); pointwise_self<F31>(P, sm, T.v31[L >> 1], g31);
      } else {
        const uint32_t sa = pos_of_freq(qq, logL), sb = pos_of_freq(L - qq, logL);
        pointwise_pair<F61>(P, sa, sb, T.v61[qq], g61);
        pointwise_pair<F31>(P, sa, sb, T.v31[qq], g31);
      }
    }
  }
  transform_both<true>(P, tid, logL, T.w2_61, T.w2_31);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t e = tid + 256u * it, r = e >> logL, i = e & (L - 1);
    const size_t addr = size_t(r ? k1b : k1a) * L + i;
    z61[addr] = Slot<F61>::get(P, e); z31[addr] = Slot<F31>::get(P, e);
  }
}

}  // namespace crt
}  // namespace mi355
