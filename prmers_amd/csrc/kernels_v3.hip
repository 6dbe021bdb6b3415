// gfx950 kernels, register-resident radix-4 set ("v3") for the small transforms: tiles of 1024 pairs, 256 threads, 4 pairs per thread.
//
// BASELINE configs[1] (p = 9815459, n = 2^19: columns of 256 with runs of four pairs, rows of 1024) is the shape the reference serves with
// forward1024_0 / sqr512 / backward1024_0 (kernels/marin.cl:1190,1517, schedule include/marin/engine_gpu.h:1591).  A 4096-pair tile of the
// radix-8 set (kernels_v2.hip) would give it 64 work-groups for 256 CUs, and that set's wave-specialised shift seams need eight waves per
// tile; the generic set (kernels.hip) keeps the tile in LDS and walks it with run-time loops: 10 passes per row transform, each with its
// table words on the critical path, about twice the instructions per word.  Here a tile is held in registers by ONE wave per SIMD:
//   * rows of 1024 = 4.4.4.4.4 and columns of 256 = 4.4.4.4 as decimation-in-frequency radix-4 steps in registers (omega_4 = 2^48: add, sub
//     and one shift), LDS only for the digit-permuting exchanges between the steps (four per direction in a row, 3 + 1 in a column);
//   * the twiddle after a step is one table word per register from the universal omega_M1 / omega_M2 tables (2 KiB / 8 KiB, cache resident),
//     ALL requested at kernel entry, so that no exchange waits for memory: with one wave per SIMD nothing else would hide that latency;
//   * every exchange has its own LDS slot map, chosen conflict-free for the lane groups gfx950 serves 128-bit accesses in (stores: eight
//     groups of 8 lanes on 32 banks, loads: four non-contiguous groups of 16 lanes on 64 banks; census in tools/lds_census.py);
//   * digits, run carries, the deferred LL subtraction, weights (TA / TB split, halved-weight bits in the DI table), the four-step twiddle
//     chain and the work-buffer row order are those of the other two sets, so the sets interoperate kernel by kernel.
// Shapes served: rows M2 = 1024 (any M1, also the 5 2^k sizes whose columns run on kernels_v5.hip); columns M1 = 256 with C = 4.
// Value ranges as in kernels_v2.hip: canonical values everywhere (P for a negated zero).
#include "kernels_v2_common.hpp"

namespace mi355 {
namespace v3 {
using v2::P2;
using v2::p2_mul;
using v2::lds_barrier;

constexpr uint32_t kThreads = 256;
constexpr uint32_t kLdsBytes = 1024 * 16;

// slot maps of the exchanges (16-byte slots of the 1024-pair tile)
__device__ __forceinline__ uint32_t m2(uint32_t i) { return i ^ ((i >> 2) & 15u); }
__device__ __forceinline__ uint32_t m3(uint32_t i) { return i ^ ((i >> 3) & 15u); }

template <bool INV>
__device__ __forceinline__ void dft4p(P2 (&x)[4]) {
  v2::dft4<INV>(x[0].a, x[1].a, x[2].a, x[3].a);
  v2::dft4<INV>(x[0].b, x[1].b, x[2].b, x[3].b);
}
__device__ __forceinline__ void twiddle3(P2 (&x)[4], const uint64_t (&w)[3]) {
#pragma unroll
  for (int k = 1; k < 4; ++k) x[k] = p2_mul(x[k], w[k - 1]);
}
// the digit fields of a thread index as the stages see it
__device__ __forceinline__ uint32_t hi2(uint32_t t) { return t >> 6; }          // top digit
__device__ __forceinline__ uint32_t d2nd(uint32_t t) { return (t >> 4) & 3u; }
__device__ __forceinline__ uint32_t d3rd(uint32_t t) { return (t >> 2) & 3u; }
// tile index of (top | second | third | fourth = j | low two bits of t) and friends: the five base-4 digits of a tile element
__device__ __forceinline__ uint32_t idx_a(uint32_t t, uint32_t j) { return 256u * j + t; }                                         // (j | t)
__device__ __forceinline__ uint32_t idx_b(uint32_t t, uint32_t j) { return 256u * hi2(t) + 64u * j + (t & 63u); }                  // (t7..6 | j | t5..0)
__device__ __forceinline__ uint32_t idx_c(uint32_t t, uint32_t j) { return 256u * hi2(t) + 64u * d2nd(t) + 16u * j + (t & 15u); }  // (t7..4 | j | t3..0)
__device__ __forceinline__ uint32_t idx_d(uint32_t t, uint32_t j) { return (t & ~3u) * 4u + 4u * j + (t & 3u); }                   // (t7..2 | j | t1..0)
__device__ __forceinline__ uint32_t idx_e(uint32_t t, uint32_t j) { return 4u * t + j; }                                           // (t | j)

#define V3_EXCH(X, x, WIDX, RIDX)                                       \
  lds_barrier();                                                        \
  _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) X[WIDX(k_)] = x[k_]; \
  lds_barrier();                                                        \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) x[j_] = X[RIDX(j_)];

// ---------------------------------------------------------------------------------------------
// middle, M2 = 1024 = 4.4.4.4.4.  Element e = 256 d1 + 64 d2 + 16 d3 + 4 d4 + d5; thread t keeps its index, the registers hold:
//   S1 d1 (elements 256 j + t)            -> k1 ; x omega_1024^(k1 t)
//   S2 d2 (k1 = t7..6, rest t5..0)         -> k2 ; x omega_256^(k2 (t & 63))
//   S3 d3 (k1, k2 = t5..4, rest t3..0)     -> k3 ; x omega_64^(k3 (t & 15))
//   S4 d4 (k1, k2, k3 = t3..2, d5 = t1..0) -> k4 ; x omega_16^(k4 (t & 3))
//   S5 d5 (k1, k2, k3, k4 = t1..0)         -> k5 ; X[k], k = k1 + 4 k2 + 16 k3 + 64 k4 + 256 k5
// pointwise in registers (rho = rho0 omega_4^k5), then the mirror image back to natural order.
// mode 0: square, 1: multiply by image Y, 2: forward only (writes the image: register j of thread t at 256 j + t).
// ---------------------------------------------------------------------------------------------
template <int mode>
__global__ void __launch_bounds__(kThreads) k2_rows1024(DevPlan pl, const uint64_t* __restrict__ Win, const uint64_t* __restrict__ Yimg,
                                                        uint64_t* __restrict__ Wout, uint32_t sub) {
  P2* X = reinterpret_cast<P2*>(v2::smem_v2);
  const uint32_t t = threadIdx.x, row = blockIdx.x;
  const P2* in = reinterpret_cast<const P2*>(Win) + size_t(row) * 1024;
  P2* out = reinterpret_cast<P2*>(Wout) + size_t(row) * 1024;
  const uint64_t* __restrict__ UT = pl.UT2;   // omega_1024^e, e < 1024
  P2 x[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = in[256 * j + t];
  // every table word of the kernel is requested now: nothing below waits for memory again
  uint64_t w1[3], w2[3], w3[3], w4[3], v1[3], v2w[3], v3w[3], v4[3];
#pragma unroll
  for (uint32_t k = 1; k < 4; ++k) {
    const uint32_t e1 = k * t, e2 = 4 * k * (t & 63u), e3 = 16 * k * (t & 15u), e4 = 64 * k * (t & 3u);
    w1[k - 1] = UT[e1]; w2[k - 1] = UT[e2]; w3[k - 1] = UT[e3]; w4[k - 1] = UT[e4];
    if (mode != 2) {
      v1[k - 1] = UT[(1024 - e1) & 1023]; v2w[k - 1] = UT[(1024 - e2) & 1023]; v3w[k - 1] = UT[(1024 - e3) & 1023]; v4[k - 1] = UT[(1024 - e4) & 1023];
    }
  }
  // rho0 = omega_m^(k1row + M1 kb): row frequency of the thread's register 0 after S5 (kernels.hip freq1 for the row's own frequency)
  const uint32_t kb = hi2(t) + 4 * d2nd(t) + 16 * d3rd(t) + 64 * (t & 3u);
  const uint32_t blk = row / pl.L1, qq = row - blk * pl.L1;
  const uint32_t k1row = col_label(pl, blk, pl.logL1 ? (__brev(qq) >> (32 - pl.logL1)) : 0u);
  const uint64_t erho = rho_exponent(pl, k1row, kb);
  uint64_t rho_lo = 0, rho_hi = 0;
  if (mode != 2) { rho_lo = pl.TWlo[erho & ((1u << pl.twh) - 1)]; rho_hi = pl.TWhi[erho >> pl.twh]; }

  // deferred small subtraction (LL's -2) on a front image: digit 0 has weight 1 and reaches column 0, plane a of every row unchanged
  if (sub != 0 && t == 0) x[0].a = gf::sub(x[0].a, uint64_t(sub));

  // ---- forward ----
  dft4p<false>(x); twiddle3(x, w1);
#define WI(k) idx_a(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  dft4p<false>(x); twiddle3(x, w2);
#define WI(k) idx_b(t, k)
#define RI(j) idx_c(t, j)
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  dft4p<false>(x); twiddle3(x, w3);
#define WI(k) m2(idx_c(t, k))
#define RI(j) m2(idx_d(t, j))
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  dft4p<false>(x); twiddle3(x, w4);
#define WI(k) m2(idx_d(t, k))
#define RI(j) m2(idx_e(t, j))
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  dft4p<false>(x);

  if (mode == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[256 * j + t] = x[j];
    return;
  }

  // ---- pointwise: register k5 holds X[kb + 256 k5]; rho = rho0 omega_4^k5 = rho0 {1, 2^48, -1, -2^48} ----
  {
    const uint64_t rho0 = gf::mul(rho_lo, rho_hi);
    const P2* Y = reinterpret_cast<const P2*>(Yimg) + size_t(row) * 1024;
#pragma unroll
    for (int k5 = 0; k5 < 4; ++k5) {
      const P2 u = x[k5];
      P2 r;
      uint64_t q, s0;
      if (mode == 0) {   // (u0 + u1 t)^2 mod (t^2 - rho), marin.cl:379-384
        q = gf::mul(gf::sqr(u.b), rho0);
        s0 = gf::sqr(u.a);
        r.b = gf::dbl(gf::mul(u.b, u.a));
      } else {           // marin.cl:387-392
        const P2 y = Y[256 * k5 + t];
        q = gf::mul(gf::mul(u.b, y.b), rho0);
        s0 = gf::mul(u.a, y.a);
        r.b = gf::add(gf::mul(u.a, y.b), gf::mul(u.b, y.a));
      }
      if (k5 & 1) q = gf::mul_pow2(q, 48);
      r.a = (k5 & 2) ? gf::sub(s0, q) : gf::add(s0, q);
      x[k5] = r;
    }
  }

  // ---- inverse (mirror) ----
  dft4p<true>(x);
#define WI(k) m2(idx_e(t, k))
#define RI(j) m2(idx_d(t, j))
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  twiddle3(x, v4); dft4p<true>(x);
#define WI(k) m2(idx_d(t, k))
#define RI(j) m2(idx_c(t, j))
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  twiddle3(x, v3w); dft4p<true>(x);
#define WI(k) idx_c(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  twiddle3(x, v2w); dft4p<true>(x);
#define WI(k) idx_b(t, k)
#define RI(j) idx_a(t, j)
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  twiddle3(x, v1); dft4p<true>(x);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[256 * j + t] = x[j];
}

// ---------------------------------------------------------------------------------------------
// The same rows with ONE PLANE per thread (512 threads: lane 2 t + plane, the two words of a pair go through identical, independent
// arithmetic): where a CU gets a single row (M1 < 512: C2, n = 2^18) the launch lasts as long as one wave's dependent stream, and two waves
// per SIMD with half the stream each are shorter than one.  Same stages, maps and table words as k2_rows1024; LDS slots are 8 bytes
// (2 map(i) + plane); the pointwise stage meets its partner plane through a DPP lane swap (quad_perm 1,0,3,2): every lane computes its
// own square / product, the product by rho travels from the b lane to the a lane.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t swap_planes(uint64_t v) {
  const uint32_t lo = uint32_t(__builtin_amdgcn_update_dpp(0, int(uint32_t(v)), 0xB1, 0xF, 0xF, false));
  const uint32_t hi = uint32_t(__builtin_amdgcn_update_dpp(0, int(uint32_t(v >> 32)), 0xB1, 0xF, 0xF, false));
  return (uint64_t(hi) << 32) | lo;
}
template <bool INV>
__device__ __forceinline__ void dft4w(uint64_t (&x)[4]) { v2::dft4<INV>(x[0], x[1], x[2], x[3]); }
__device__ __forceinline__ void twiddle3w(uint64_t (&x)[4], const uint64_t (&w)[3]) {
#pragma unroll
  for (int k = 1; k < 4; ++k) x[k] = gf::mul(x[k], w[k - 1]);
}
#define V3_EXCHW(X, x, pln, WIDX, RIDX)                                              \
  lds_barrier();                                                                     \
  _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) X[2 * (WIDX(k_)) + pln] = x[k_];  \
  lds_barrier();                                                                     \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) x[j_] = X[2 * (RIDX(j_)) + pln];

template <int mode>
__global__ void __launch_bounds__(2 * kThreads) k2_rows1024_planes(DevPlan pl, const uint64_t* __restrict__ Win, const uint64_t* __restrict__ Yimg,
                                                                   uint64_t* __restrict__ Wout, uint32_t sub) {
  uint64_t* X = reinterpret_cast<uint64_t*>(v2::smem_v2);
  const uint32_t t = threadIdx.x >> 1, pln = threadIdx.x & 1u, row = blockIdx.x;
  const uint64_t* in = Win + size_t(row) * 2048;
  uint64_t* out = Wout + size_t(row) * 2048;
  const uint64_t* __restrict__ UT = pl.UT2;
  uint64_t x[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = in[2 * (256 * j + t) + pln];
  uint64_t w1[3], w2[3], w3[3], w4[3], v1[3], v2w[3], v3w[3], v4[3];
#pragma unroll
  for (uint32_t k = 1; k < 4; ++k) {
    const uint32_t e1 = k * t, e2 = 4 * k * (t & 63u), e3 = 16 * k * (t & 15u), e4 = 64 * k * (t & 3u);
    w1[k - 1] = UT[e1]; w2[k - 1] = UT[e2]; w3[k - 1] = UT[e3]; w4[k - 1] = UT[e4];
    if (mode != 2) {
      v1[k - 1] = UT[(1024 - e1) & 1023]; v2w[k - 1] = UT[(1024 - e2) & 1023]; v3w[k - 1] = UT[(1024 - e3) & 1023]; v4[k - 1] = UT[(1024 - e4) & 1023];
    }
  }
  const uint32_t kb = hi2(t) + 4 * d2nd(t) + 16 * d3rd(t) + 64 * (t & 3u);
  const uint32_t blk = row / pl.L1, qq = row - blk * pl.L1;
  const uint32_t k1row = col_label(pl, blk, pl.logL1 ? (__brev(qq) >> (32 - pl.logL1)) : 0u);
  const uint64_t erho = rho_exponent(pl, k1row, kb);
  uint64_t rho_lo = 0, rho_hi = 0;
  if (mode != 2) { rho_lo = pl.TWlo[erho & ((1u << pl.twh) - 1)]; rho_hi = pl.TWhi[erho >> pl.twh]; }
  if (sub != 0 && threadIdx.x == 0) x[0] = gf::sub(x[0], uint64_t(sub));   // element 0, plane a

  // ---- forward ----
  dft4w<false>(x); twiddle3w(x, w1);
#define WI(k) idx_a(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x); twiddle3w(x, w2);
#define WI(k) idx_b(t, k)
#define RI(j) idx_c(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x); twiddle3w(x, w3);
#define WI(k) m2(idx_c(t, k))
#define RI(j) m2(idx_d(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x); twiddle3w(x, w4);
#define WI(k) m2(idx_d(t, k))
#define RI(j) m2(idx_e(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x);

  if (mode == 2) {   // the image: the layout of k2_rows1024 (register j of thread t at pair 256 j + t)
#pragma unroll
    for (int j = 0; j < 4; ++j) out[2 * (256 * j + t) + pln] = x[j];
    return;
  }

  // ---- pointwise: lane a holds u.a, lane b holds u.b of X[kb + 256 k5]; rho = rho0 {1, 2^48, -1, -2^48} ----
  {
    const uint64_t rho0 = gf::mul(rho_lo, rho_hi);
    const uint64_t* Y = Yimg + size_t(row) * 2048;
#pragma unroll
    for (int k5 = 0; k5 < 4; ++k5) {
      const uint64_t u = x[k5], uo = swap_planes(u);
      uint64_t m1, cross;
      if (mode == 0) {   // (u0 + u1 t)^2 mod (t^2 - rho), marin.cl:379-384
        m1 = gf::sqr(u);                      // a: u.a^2, b: u.b^2
        cross = gf::dbl(gf::mul(u, uo));      // 2 u.a u.b (both lanes)
      } else {           // marin.cl:387-392
        const uint64_t y = Y[2 * (256 * k5 + t) + pln], yo = swap_planes(y);
        m1 = gf::mul(u, y);                   // a: u.a y.a, b: u.b y.b
        const uint64_t m3 = gf::mul(u, yo);   // a: u.a y.b, b: u.b y.a
        cross = gf::add(m3, swap_planes(m3));
      }
      uint64_t q = swap_planes(gf::mul(m1, rho0));   // lane a receives u.b^2 rho0 (u.b y.b rho0)
      if (k5 & 1) q = gf::mul_pow2(q, 48);
      const uint64_t ra = (k5 & 2) ? gf::sub(m1, q) : gf::add(m1, q);
      x[k5] = pln ? cross : ra;
    }
  }

  // ---- inverse (mirror) ----
  dft4w<true>(x);
#define WI(k) m2(idx_e(t, k))
#define RI(j) m2(idx_d(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v4); dft4w<true>(x);
#define WI(k) m2(idx_d(t, k))
#define RI(j) m2(idx_c(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v3w); dft4w<true>(x);
#define WI(k) idx_c(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v2w); dft4w<true>(x);
#define WI(k) idx_b(t, k)
#define RI(j) idx_a(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v1); dft4w<true>(x);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[2 * (256 * j + t) + pln] = x[j];
}

// ---------------------------------------------------------------------------------------------
// Column tiles, M1 = 256 = 4.4.4.4 with C = 4 pairs per run (tile = 1024 pairs).  Tile element (i1, c), i1 = 64 d1 + 16 d2 + 4 d3 + d4,
// tile index 4 i1 + c (five base-4 digits d1 | d2 | d3 | d4 | c).
// front (digits -> work buffer):
//   S0 thread i1 = t, registers c: one run of 8 digits -> weight                    ; exchange (t | c) -> (d1 | t)
//   S1 thread (d2 d3 d4 | c) regs d1 -> k1 ; x omega_256^(k1 (t >> 2))
//   S2 thread (k1 | d3 d4 | c) regs d2 -> k2 ; x omega_64^(k2 ((t >> 2) & 15))
//   S3 thread (k1 k2 | d4 | c) regs d3 -> k3 ; x omega_16^(k3 ((t >> 2) & 3))
//   S4 thread (k1 k2 k3 | c)   regs d4 -> k4 ; k1col = k1 + 4 k2 + 16 k3 + 64 k4
//   then the four-step twiddle omega_m^(i2 k1col) * TB (geometric in k4: one chain multiply per pair) and the store to work-buffer row
//   bitrev(k1col), column i2 = 4 T + c.
// back is the mirror image, followed by unweight and the sequential carry of the thread's run.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rev2(uint32_t k) { return ((k & 1u) << 1) | (k >> 1); }

__global__ void __launch_bounds__(kThreads) k1_cols256(DevPlan pl, const uint32_t* __restrict__ digits, const uint64_t* __restrict__ cbuf_in, uint32_t sub,
                                                       uint64_t* __restrict__ Wout) {
  P2* X = reinterpret_cast<P2*>(v2::smem_v2);
  const uint32_t t = threadIdx.x, T = blockIdx.x;
  const uint64_t* __restrict__ UT = pl.UT1;   // omega_256^e
  // table words first
  uint64_t w1[3], w2[3], w3[3];
#pragma unroll
  for (uint32_t k = 1; k < 4; ++k) { w1[k - 1] = UT[k * (t >> 2)]; w2[k - 1] = UT[4 * k * ((t >> 2) & 15u)]; w3[k - 1] = UT[16 * k * ((t >> 2) & 3u)]; }
  const uint32_t c4 = t & 3u, kb = hi2(t) + 4 * d2nd(t) + 16 * d3rd(t), i2 = 4 * T + c4;
  const uint64_t fca0 = pl.F0f[size_t(T) * kThreads + t], fB = pl.FBf[i2];   // chain start omega_m^(i2 kb) TB[2 i2], ratio omega_m^(64 i2)
  const uint32_t di = pl.DI[size_t(T) * kThreads + t];
  const uint64_t tah = pl.TAh[t], tah1 = pl.TAh[256 + t];   // odd digits: exponent split SA[M1 + i1] + SB[2 i2] (plan.hpp)
  uint32_t dg[8];
  {
    const uint4* src = reinterpret_cast<const uint4*>(digits) + (size_t(T) * 256 + t) * 2;
    const uint4 a = src[0], b = src[1];
    dg[0] = a.x; dg[1] = a.y; dg[2] = a.z; dg[3] = a.w; dg[4] = b.x; dg[5] = b.y; dg[6] = b.z; dg[7] = b.w;
  }
  if (cbuf_in) v2::apply_carry_in<8>(pl, di, 0, v2::carry_in_of(pl, cbuf_in, T, t), dg);
  P2 x[4];
  const uint32_t nowrap = ~di;   // bit 2 idx + 1 of di: the weight exponents of digit idx wrapped
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    // weight TA*TB, halved when the exponents wrap: the halving sits on TA and the un-wrapped digits are doubled instead
    x[c] = {gf::mul_u32(tah, dg[2 * c] << ((nowrap >> (4 * c + 1)) & 1u)), gf::mul_u32(tah1, dg[2 * c + 1] << ((nowrap >> (4 * c + 3)) & 1u))};
  }
  if (sub != 0 && T == 0 && t == 0) x[0].a = gf::sub(x[0].a, uint64_t(sub));   // digit 0 has weight 1
#define WI(k) m3(idx_e(t, k))
#define RI(j) m3(idx_a(t, j))
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  dft4p<false>(x); twiddle3(x, w1);
#define WI(k) idx_a(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  dft4p<false>(x); twiddle3(x, w2);
#define WI(k) idx_b(t, k)
#define RI(j) idx_c(t, j)
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  dft4p<false>(x); twiddle3(x, w3);
#define WI(k) m2(idx_c(t, k))
#define RI(j) m2(idx_d(t, j))
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  dft4p<false>(x);
  {
    uint64_t ca = fca0;
    const uint32_t row0 = __brev(kb) >> 24;   // bitrev8(kb): its low 2 bits are zero
    P2* W = reinterpret_cast<P2*>(Wout);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      W[size_t(row0 + rev2(j)) * pl.M2 + i2] = {gf::mul(x[j].a, ca), gf::mul(x[j].b, ca)};
      if (j < 3) ca = gf::mul(ca, fB);
    }
  }
}

template <bool EXT>
__global__ void __launch_bounds__(kThreads) k3_cols256(DevPlan pl, const uint64_t* __restrict__ Win, uint32_t* __restrict__ digits, uint64_t* __restrict__ cbuf,
                                                       uint32_t a, uint64_t scale, BackExt ext) {
  P2* X = reinterpret_cast<P2*>(v2::smem_v2);
  const uint32_t t = threadIdx.x, T = v2::tile_of_block(pl, blockIdx.x, gridDim.x);
  const uint64_t* __restrict__ UT = pl.UT1;
  uint64_t v1[3], v2w[3], v3w[3];
#pragma unroll
  for (uint32_t k = 1; k < 4; ++k) {
    v1[k - 1] = UT[(256 - k * (t >> 2)) & 255]; v2w[k - 1] = UT[(256 - 4 * k * ((t >> 2) & 15u)) & 255]; v3w[k - 1] = UT[(256 - 16 * k * ((t >> 2) & 3u)) & 255];
  }
  const uint32_t c4 = t & 3u, kb = hi2(t) + 4 * d2nd(t) + 16 * d3rd(t), i2 = 4 * T + c4;
  const uint32_t di = pl.DI[size_t(T) * kThreads + t];
  const uint64_t tai_e = pl.TAi[t], tai_o = pl.TAi[256 + t];
  uint32_t ad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (EXT && ext.add_digits) {
    const uint4* src = reinterpret_cast<const uint4*>(ext.add_digits) + (size_t(T) * 256 + t) * 2;
    const uint4 p = src[0], q = src[1];
    ad[0] = p.x; ad[1] = p.y; ad[2] = p.z; ad[3] = p.w; ad[4] = q.x; ad[5] = q.y; ad[6] = q.z; ad[7] = q.w;
    if (ext.add_cbuf) v2::apply_carry_in<8>(pl, di, 0, v2::carry_in_of(pl, ext.add_cbuf, T, t), ad);
  }
  P2 x[4];
  {
    uint64_t ca = pl.F0i[size_t(T) * kThreads + t];   // chain start omega_m^-(i2 kb) TBi[2 i2], ratio omega_m^-(64 i2)
    const uint64_t B = pl.FBi[i2];
    if (scale != 1) ca = gf::mul(ca, scale);
    const uint32_t row0 = __brev(kb) >> 24;
    const P2* W = reinterpret_cast<const P2*>(Win);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = W[size_t(row0 + rev2(j)) * pl.M2 + i2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = {gf::mul(x[j].a, ca), gf::mul(x[j].b, ca)};
      if (j < 3) ca = gf::mul(ca, B);
    }
  }
  dft4p<true>(x);
#define WI(k) m2(idx_d(t, k))
#define RI(j) m2(idx_c(t, j))
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  twiddle3(x, v3w); dft4p<true>(x);
#define WI(k) idx_c(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  twiddle3(x, v2w); dft4p<true>(x);
#define WI(k) idx_b(t, k)
#define RI(j) idx_a(t, j)
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  twiddle3(x, v1); dft4p<true>(x);
#define WI(k) m3(idx_a(t, k))
#define RI(j) m3(idx_e(t, j))
  V3_EXCH(X, x, WI, RI)
#undef WI
#undef RI
  // unweight, x a, carry along the thread's run (i1 = t)
  const uint64_t tai2_e = pl.TAi2[t], tai2_o = pl.TAi2[256 + t];
  uint64_t carry = 0;
  uint32_t dg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t bits = di >> (2 * k);   // digit-info table: width - q, wrap
    const uint32_t width = pl.q + (bits & 1u);
    const bool wrap = (bits & 2u) != 0;
    const P2 v = x[k >> 1];
    const uint64_t u = (k & 1) ? gf::mul(v.b, wrap ? tai2_o : tai_o) : gf::mul(v.a, wrap ? tai2_e : tai_e);   // wrapped exponents: weight was halved
    const uint64_t mask = (uint64_t(1) << width) - 1;   // adc_mul, marin.cl:194-201
    if (a == 1) {
      const uint64_t r = u + carry + (EXT ? ad[k] : 0u);
      dg[k] = __builtin_amdgcn_ubfe(uint32_t(r), 0u, width);
      carry = r >> width;
    } else {
      const uint64_t dlo = u & mask, chi = u >> width;
      const uint64_t r = dlo * a + carry + (EXT ? ad[k] : 0u);
      dg[k] = uint32_t(r & mask);
      carry = (r >> width) + chi * a;
    }
  }
  uint4* dst = reinterpret_cast<uint4*>(digits) + (size_t(T) * 256 + t) * 2;
  dst[0] = make_uint4(dg[0], dg[1], dg[2], dg[3]); dst[1] = make_uint4(dg[4], dg[5], dg[6], dg[7]);
  cbuf[size_t(T) * 256 + t] = carry;
  if (EXT && ext.digits2) {
    uint4* d2 = reinterpret_cast<uint4*>(ext.digits2) + (size_t(T) * 256 + t) * 2;
    d2[0] = make_uint4(dg[0], dg[1], dg[2], dg[3]); d2[1] = make_uint4(dg[4], dg[5], dg[6], dg[7]);
    ext.cbuf2[size_t(T) * 256 + t] = carry;
  }
}

// ---------------------------------------------------------------------------------------------
// The column kernels with one plane per thread (512 threads, lane 2 t + plane): plane a = the even digits of the run, plane b the odd ones.
// Same stages, maps and tables as k1_cols256 / k3_cols256; both lanes of a pair load the run and compute its carry-in / carry chain (a few
// integer instructions), each weights, transforms and twiddles its own plane; the back sweep meets the partner plane through a DPP lane swap
// before the carry.  Chosen where the launch is a single round of tiles (one wave per SIMD otherwise): shorter dependent stream per wave.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(2 * kThreads) k1_cols256_planes(DevPlan pl, const uint32_t* __restrict__ digits, const uint64_t* __restrict__ cbuf_in, uint32_t sub,
                                                                  uint64_t* __restrict__ Wout) {
  uint64_t* X = reinterpret_cast<uint64_t*>(v2::smem_v2);
  const uint32_t t = threadIdx.x >> 1, pln = threadIdx.x & 1u, T = blockIdx.x;
  const uint64_t* __restrict__ UT = pl.UT1;
  uint64_t w1[3], w2[3], w3[3];
#pragma unroll
  for (uint32_t k = 1; k < 4; ++k) { w1[k - 1] = UT[k * (t >> 2)]; w2[k - 1] = UT[4 * k * ((t >> 2) & 15u)]; w3[k - 1] = UT[16 * k * ((t >> 2) & 3u)]; }
  const uint32_t c4 = t & 3u, kb = hi2(t) + 4 * d2nd(t) + 16 * d3rd(t), i2 = 4 * T + c4;
  const uint64_t fca0 = pl.F0f[size_t(T) * kThreads + t], fB = pl.FBf[i2];
  const uint32_t di = pl.DI[size_t(T) * kThreads + t];
  const uint64_t tah = pl.TAh[256 * pln + t];   // odd digits: the second half of TA (plan.hpp)
  uint32_t dg[8];
  {
    const uint4* src = reinterpret_cast<const uint4*>(digits) + (size_t(T) * 256 + t) * 2;
    const uint4 a = src[0], b = src[1];
    dg[0] = a.x; dg[1] = a.y; dg[2] = a.z; dg[3] = a.w; dg[4] = b.x; dg[5] = b.y; dg[6] = b.z; dg[7] = b.w;
  }
  if (cbuf_in) v2::apply_carry_in<8>(pl, di, 0, v2::carry_in_of(pl, cbuf_in, T, t), dg);
  uint64_t x[4];
  const uint32_t nowrap = ~di;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    // (a select between two elements of dg would be turned into a run-time index and put the array into scratch: mask arithmetic instead)
    const uint32_t d = dg[2 * c] ^ ((dg[2 * c] ^ dg[2 * c + 1]) & (0u - pln));
    const uint32_t sh = nowrap >> (4 * c + 1 + 2 * pln);
    x[c] = gf::mul_u32(tah, d << (sh & 1u));
  }
  if (sub != 0 && T == 0 && threadIdx.x == 0) x[0] = gf::sub(x[0], uint64_t(sub));   // digit 0 has weight 1
#define WI(k) m2(idx_e(t, k))
#define RI(j) m2(idx_a(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x); twiddle3w(x, w1);
#define WI(k) idx_a(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x); twiddle3w(x, w2);
#define WI(k) idx_b(t, k)
#define RI(j) idx_c(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x); twiddle3w(x, w3);
#define WI(k) m2(idx_c(t, k))
#define RI(j) m2(idx_d(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x);
  {
    uint64_t ca = fca0;
    const uint32_t row0 = __brev(kb) >> 24;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Wout[2 * (size_t(row0 + rev2(j)) * pl.M2 + i2) + pln] = gf::mul(x[j], ca);
      if (j < 3) ca = gf::mul(ca, fB);
    }
  }
}

template <bool EXT>
__global__ void __launch_bounds__(2 * kThreads) k3_cols256_planes(DevPlan pl, const uint64_t* __restrict__ Win, uint32_t* __restrict__ digits,
                                                                  uint64_t* __restrict__ cbuf, uint32_t a, uint64_t scale, BackExt ext) {
  uint64_t* X = reinterpret_cast<uint64_t*>(v2::smem_v2);
  const uint32_t t = threadIdx.x >> 1, pln = threadIdx.x & 1u, T = v2::tile_of_block(pl, blockIdx.x, gridDim.x);
  const uint64_t* __restrict__ UT = pl.UT1;
  uint64_t v1[3], v2w[3], v3w[3];
#pragma unroll
  for (uint32_t k = 1; k < 4; ++k) {
    v1[k - 1] = UT[(256 - k * (t >> 2)) & 255]; v2w[k - 1] = UT[(256 - 4 * k * ((t >> 2) & 15u)) & 255]; v3w[k - 1] = UT[(256 - 16 * k * ((t >> 2) & 3u)) & 255];
  }
  const uint32_t c4 = t & 3u, kb = hi2(t) + 4 * d2nd(t) + 16 * d3rd(t), i2 = 4 * T + c4;
  const uint32_t di = pl.DI[size_t(T) * kThreads + t];
  const uint64_t tai = pl.TAi[256 * pln + t];
  uint32_t ad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (EXT && ext.add_digits) {
    const uint4* src = reinterpret_cast<const uint4*>(ext.add_digits) + (size_t(T) * 256 + t) * 2;
    const uint4 p = src[0], q = src[1];
    ad[0] = p.x; ad[1] = p.y; ad[2] = p.z; ad[3] = p.w; ad[4] = q.x; ad[5] = q.y; ad[6] = q.z; ad[7] = q.w;
    if (ext.add_cbuf) v2::apply_carry_in<8>(pl, di, 0, v2::carry_in_of(pl, ext.add_cbuf, T, t), ad);
  }
  uint64_t x[4];
  {
    uint64_t ca = pl.F0i[size_t(T) * kThreads + t];
    const uint64_t B = pl.FBi[i2];
    if (scale != 1) ca = gf::mul(ca, scale);
    const uint32_t row0 = __brev(kb) >> 24;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = Win[2 * (size_t(row0 + rev2(j)) * pl.M2 + i2) + pln];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = gf::mul(x[j], ca);
      if (j < 3) ca = gf::mul(ca, B);
    }
  }
  dft4w<true>(x);
#define WI(k) m2(idx_d(t, k))
#define RI(j) m2(idx_c(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v3w); dft4w<true>(x);
#define WI(k) idx_c(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v2w); dft4w<true>(x);
#define WI(k) idx_b(t, k)
#define RI(j) idx_a(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v1); dft4w<true>(x);
#define WI(k) m2(idx_a(t, k))
#define RI(j) m2(idx_e(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  // unweight this lane's plane (digit 2 c + plane of the run i1 = t), then both lanes take the partner's four values and run the carry
  const uint64_t tai2 = pl.TAi2[256 * pln + t];
  uint64_t own[4], oth[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool wrap = ((di >> (2 * (2 * c + int(pln)))) & 2u) != 0;   // digit-info table: width - q, wrap
    own[c] = gf::mul(x[c], wrap ? tai2 : tai);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) oth[c] = swap_planes(own[c]);
  uint64_t carry = 0;
  uint32_t dg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t bits = di >> (2 * k);
    const uint32_t width = pl.q + (bits & 1u);
    const uint64_t u = ((k & 1) == int(pln)) ? own[k >> 1] : oth[k >> 1];
    const uint64_t mask = (uint64_t(1) << width) - 1;   // adc_mul, marin.cl:194-201
    if (a == 1) {
      const uint64_t r = u + carry + (EXT ? ad[k] : 0u);
      dg[k] = __builtin_amdgcn_ubfe(uint32_t(r), 0u, width);
      carry = r >> width;
    } else {
      const uint64_t dlo = u & mask, chi = u >> width;
      const uint64_t r = dlo * a + carry + (EXT ? ad[k] : 0u);
      dg[k] = uint32_t(r & mask);
      carry = (r >> width) + chi * a;
    }
  }
  // lane a stores the first four digits (and the carry word), lane b the last four
  uint4* dst = reinterpret_cast<uint4*>(digits) + (size_t(T) * 256 + t) * 2;
  dst[pln] = pln ? make_uint4(dg[4], dg[5], dg[6], dg[7]) : make_uint4(dg[0], dg[1], dg[2], dg[3]);
  if (!pln) cbuf[size_t(T) * 256 + t] = carry;
  if (EXT && ext.digits2) {
    uint4* d2 = reinterpret_cast<uint4*>(ext.digits2) + (size_t(T) * 256 + t) * 2;
    d2[pln] = pln ? make_uint4(dg[4], dg[5], dg[6], dg[7]) : make_uint4(dg[0], dg[1], dg[2], dg[3]);
    if (!pln) ext.cbuf2[size_t(T) * 256 + t] = carry;
  }
}

#if defined(MI355_EXPERIMENTAL)
// ---------------------------------------------------------------------------------------------
// Back sweep of squaring i and front sweep of squaring i + 1 in ONE launch (plane-per-thread form), for the shapes whose tiles are all
// resident at once (C2: 256 tiles, n = 2^20: 512; Engine::square_mul_n runs front | rows | [this | rows] x (count - 1) | back).
// Everything between the two sweeps is local to a tile -- the thread that ends the back sweep with the eight digits of run i1 = t is the
// thread that starts the front sweep with them, so the digits never leave their registers -- except the carry word of the PREVIOUS run in
// digit order, which the neighbouring tile produces (reference: carry_weight_mul_p1 / p2 hand it through a carry array and a second
// kernel, kernels/marin.cl:1696-1728,2198-2216).  Hand-over inside the launch: every group stores its 256 carry words with agent-scope
// stores (past the XCD's non-coherent L2), drains them (s_waitcnt vmcnt(0)), publishes an epoch in its flag word, then waits for the flag
// of the one tile it depends on and reads that tile's words with agent-scope loads.  A carry-out depends only on the tile's own data, so
// there is no cycle; tiles are taken in block order, so the tile waited for is resident or done (see the kernel's first lines).
// A wait that outlasts kChainTimeoutTicks (100 MHz clock: 50 ms) raises the error word -- the engine refuses every later read-out -- and
// goes on, so that the grid always drains.  MEASURED: C2 0.0265 -> 0.0259 ms (-2 %), n = 2^20 no change (profiles/r04_ab_chain_backfront.txt):
// the hand-over costs what the kernel boundary it replaces costs.  Kept out of the product library for that (inter-group waits for 2 %):
// built only with -DMI355_EXPERIMENTAL (make exp), parity-tested there (tools/exp_coop_check.py).
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kChainTimeoutTicks = 5000000ull;

__global__ void __launch_bounds__(2 * kThreads) k31_cols256_planes(DevPlan pl, uint64_t* __restrict__ Wbuf, uint32_t a, uint32_t sub, uint64_t* __restrict__ xbuf,
                                                                   uint32_t* __restrict__ flags, uint32_t* __restrict__ err, uint32_t epoch) {
  uint64_t* X = reinterpret_cast<uint64_t*>(v2::smem_v2);
  __shared__ uint32_t wait_ok;
  // Tile = block index, in plain order: a group waits for its predecessor tile only, work-groups are dispatched in ascending order, so the
  // group waited for is resident or done whatever else shares the GPU (only tile 0 waits for the last tile, which every other group's
  // progress brings in): no dead-lock even when the launch is not resident as a whole.
  const uint32_t t = threadIdx.x >> 1, pln = threadIdx.x & 1u, T = blockIdx.x, NT = gridDim.x;
  const uint64_t* __restrict__ UT = pl.UT1;
  uint64_t v1[3], v2w[3], v3w[3], w1[3], w2[3], w3[3];
#pragma unroll
  for (uint32_t k = 1; k < 4; ++k) {
    const uint32_t e1 = k * (t >> 2), e2 = 4 * k * ((t >> 2) & 15u), e3 = 16 * k * ((t >> 2) & 3u);
    v1[k - 1] = UT[(256 - e1) & 255]; v2w[k - 1] = UT[(256 - e2) & 255]; v3w[k - 1] = UT[(256 - e3) & 255];
    w1[k - 1] = UT[e1]; w2[k - 1] = UT[e2]; w3[k - 1] = UT[e3];
  }
  const uint32_t c4 = t & 3u, kb = hi2(t) + 4 * d2nd(t) + 16 * d3rd(t), i2 = 4 * T + c4;
  const uint32_t di = pl.DI[size_t(T) * kThreads + t];
  const uint64_t tai = pl.TAi[256 * pln + t], tah = pl.TAh[256 * pln + t];
  const uint64_t fca0 = pl.F0f[size_t(T) * kThreads + t], fB = pl.FBf[i2];
  const uint32_t row0 = __brev(kb) >> 24;
  // ---- back sweep of the tile (k3_cols256_planes) ----
  uint64_t x[4];
  {
    uint64_t ca = pl.F0i[size_t(T) * kThreads + t];
    const uint64_t B = pl.FBi[i2];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = Wbuf[2 * (size_t(row0 + rev2(j)) * pl.M2 + i2) + pln];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = gf::mul(x[j], ca);
      if (j < 3) ca = gf::mul(ca, B);
    }
  }
  dft4w<true>(x);
#define WI(k) m2(idx_d(t, k))
#define RI(j) m2(idx_c(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v3w); dft4w<true>(x);
#define WI(k) idx_c(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v2w); dft4w<true>(x);
#define WI(k) idx_b(t, k)
#define RI(j) idx_a(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  twiddle3w(x, v1); dft4w<true>(x);
#define WI(k) m2(idx_a(t, k))
#define RI(j) m2(idx_e(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  const uint64_t tai2 = pl.TAi2[256 * pln + t];
  uint64_t own[4], oth[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool wrap = ((di >> (2 * (2 * c + int(pln)))) & 2u) != 0;
    own[c] = gf::mul(x[c], wrap ? tai2 : tai);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) oth[c] = swap_planes(own[c]);
  uint64_t carry = 0;
  uint32_t dg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t bits = di >> (2 * k);
    const uint32_t width = pl.q + (bits & 1u);
    const uint64_t u = ((k & 1) == int(pln)) ? own[k >> 1] : oth[k >> 1];
    const uint64_t mask = (uint64_t(1) << width) - 1;   // adc_mul, marin.cl:194-201
    if (a == 1) {
      const uint64_t r = u + carry;
      dg[k] = __builtin_amdgcn_ubfe(uint32_t(r), 0u, width);
      carry = r >> width;
    } else {
      const uint64_t dlo = u & mask, chi = u >> width;
      const uint64_t r = dlo * a + carry;
      dg[k] = uint32_t(r & mask);
      carry = (r >> width) + chi * a;
    }
  }
  // ---- hand the run carries to the next tile in digit order, take those of the previous one ----
  if (!pln) __hip_atomic_store(&xbuf[size_t(T) * 256 + t], carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are at the memory side before the flag can be seen
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&flags[T], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t Tdep = T ? T - 1 : NT - 1;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t ok = 1, spins = 0;
    while (int32_t(__hip_atomic_load(&flags[Tdep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
      if ((++spins & 63u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > kChainTimeoutTicks) { ok = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_ok = ok;
  }
  __syncthreads();
  {
    // previous run in digit order (v2::carry_in_of): same row of the previous tile; the first tile wraps to the last tile of the previous row
    const size_t src = T ? size_t(T - 1) * 256 + t : size_t(NT - 1) * 256 + (t ? t - 1 : 255u);
    const uint64_t cin = __hip_atomic_load(&xbuf[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v2::apply_carry_in<8>(pl, di, 0, wait_ok ? cin : 0, dg);
  }
  // ---- front sweep of the same tile for the next squaring (k1_cols256_planes) ----
  const uint32_t nowrap = ~di;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t d = dg[2 * c] ^ ((dg[2 * c] ^ dg[2 * c + 1]) & (0u - pln));
    const uint32_t sh = nowrap >> (4 * c + 1 + 2 * pln);
    x[c] = gf::mul_u32(tah, d << (sh & 1u));
  }
  if (sub != 0 && T == 0 && threadIdx.x == 0) x[0] = gf::sub(x[0], uint64_t(sub));
#define WI(k) m2(idx_e(t, k))
#define RI(j) m2(idx_a(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x); twiddle3w(x, w1);
#define WI(k) idx_a(t, k)
#define RI(j) idx_b(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x); twiddle3w(x, w2);
#define WI(k) idx_b(t, k)
#define RI(j) idx_c(t, j)
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x); twiddle3w(x, w3);
#define WI(k) m2(idx_c(t, k))
#define RI(j) m2(idx_d(t, j))
  V3_EXCHW(X, x, pln, WI, RI)
#undef WI
#undef RI
  dft4w<false>(x);
  {
    uint64_t ca = fca0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Wbuf[2 * (size_t(row0 + rev2(j)) * pl.M2 + i2) + pln] = gf::mul(x[j], ca);
      if (j < 3) ca = gf::mul(ca, fB);
    }
  }
}

#endif   // MI355_EXPERIMENTAL

// chain starts and ratios of the four-step twiddle chains (same thread map as the last stage of k1_cols256 / first stage of k3_cols256):
// F0f[T][t] = omega_m^(i2 kb) TB[2 i2], F0i the inverse with TBi, FBf[i2] = omega_m^(64 i2), FBi its inverse
__global__ void __launch_bounds__(kThreads) k_build_f0(DevPlan pl, uint64_t* __restrict__ f0f, uint64_t* __restrict__ f0i, uint64_t* __restrict__ fbf,
                                                       uint64_t* __restrict__ fbi) {
  const uint32_t t = threadIdx.x, T = blockIdx.x;
  const uint32_t c4 = t & 3u, kb = hi2(t) + 4 * d2nd(t) + 16 * d3rd(t), i2 = 4 * T + c4;
  const uint32_t ea = i2 * kb;
  f0f[size_t(T) * kThreads + t] = gf::mul(v2::tw_lookup(pl, ea), pl.TB[2 * i2]);
  f0i[size_t(T) * kThreads + t] = gf::mul(v2::tw_lookup(pl, ea ? pl.m - ea : 0), pl.TBi[2 * i2]);
  if (t < 4) {
    const uint32_t eb = i2 * 64;
    fbf[i2] = v2::tw_lookup(pl, eb);
    fbi[i2] = v2::tw_lookup(pl, eb ? pl.m - eb : 0);
  }
}

}  // namespace v3

// ------------------------------- shapes and launch wrappers -----------------------------------
// MI355_TUNE bit 7 switches the set off (A/B runs against the generic kernels).
// Rows: with one row per CU the launch lasts as long as one wave's dependent stream: the pair-per-thread kernel (one wave per SIMD) takes
// 12.8 us at C2 against 10.6 us for the generic rows (eight waves per row, one plane of a butterfly per thread); the plane-per-thread form
// below serves those sizes; with five rows per CU (n = 5 2^19) the pair form's lower instruction count wins, 26.0 against 30.1 us
// (same-box A/B, profiles/r04_ab_radix4_set.txt).
bool v3_rows_shape(const DevPlan& pl) { return pl.M2 == 1024 && !(pl.tune & 128); }
bool v3_cols_shape(const DevPlan& pl) { return pl.r5 == 1 && pl.M1 == 256 && pl.C == 4 && pl.M2 >= 8 && pl.DI != nullptr && !(pl.tune & 128); }
size_t v3_threads_per_tile() { return v3::kThreads; }

hipError_t v3_build_fourstep(const DevPlan& pl, uint64_t* f0f, uint64_t* f0i, uint64_t* fbf, uint64_t* fbi, hipStream_t s) {
  hipLaunchKernelGGL(v3::k_build_f0, dim3(pl.M2 / 4), dim3(v3::kThreads), 0, s, pl, f0f, f0i, fbf, fbi);
  return hipGetLastError();
}
hipError_t v3_launch_middle(const DevPlan& pl, const uint64_t* Win, const uint64_t* Y, uint64_t* Wout, int mode, uint32_t sub, hipStream_t s) {
  const dim3 grid(pl.M1), block(v3::kThreads);
  // one row per CU or fewer: one plane per thread (twice the waves, half the stream each); more rows: a pair per thread (fewer
  // instructions per word).  MI355_TUNE bit 8 forces the pair form, bit 9 the plane form (A/B runs).
  const bool planes = ((pl.M1 < 512) && !(pl.tune & 256)) || (pl.tune & 512);
  if (planes) {
    const dim3 block2(2 * v3::kThreads);
    switch (mode) {
      case 0: hipLaunchKernelGGL(v3::k2_rows1024_planes<0>, grid, block2, v3::kLdsBytes, s, pl, Win, Y, Wout, sub); break;
      case 1: hipLaunchKernelGGL(v3::k2_rows1024_planes<1>, grid, block2, v3::kLdsBytes, s, pl, Win, Y, Wout, sub); break;
      default: hipLaunchKernelGGL(v3::k2_rows1024_planes<2>, grid, block2, v3::kLdsBytes, s, pl, Win, Y, Wout, sub); break;
    }
    return hipGetLastError();
  }
  switch (mode) {
    case 0: hipLaunchKernelGGL(v3::k2_rows1024<0>, grid, block, v3::kLdsBytes, s, pl, Win, Y, Wout, sub); break;
    case 1: hipLaunchKernelGGL(v3::k2_rows1024<1>, grid, block, v3::kLdsBytes, s, pl, Win, Y, Wout, sub); break;
    default: hipLaunchKernelGGL(v3::k2_rows1024<2>, grid, block, v3::kLdsBytes, s, pl, Win, Y, Wout, sub); break;
  }
  return hipGetLastError();
}
// columns: one plane per thread where the tiles make a single round (at most two per CU; MI355_TUNE bit 10 forces the pair form, bit 11 the plane form)
static bool cols_planes(const DevPlan& pl) { return ((pl.M2 / 4 <= 512) && !(pl.tune & 1024)) || (pl.tune & 2048); }
#if defined(MI355_EXPERIMENTAL)
// back of one squaring + front of the next in one launch: only where every tile of the launch is resident at once (the groups wait for
// each other's carry words) and the plane form is the one in use; 0: not served.  MI355_TUNE bit 12 switches it off (A/B runs).
uint32_t v3_chain_tiles(const DevPlan& pl, int device) {
  if (!v3_cols_shape(pl) || !cols_planes(pl) || (pl.tune & 4096)) return 0;
  int per_cu = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v3::k31_cols256_planes, int(2 * v3::kThreads), v3::kLdsBytes + 16) != hipSuccess || per_cu < 1) return 0;
  const uint32_t tiles = pl.M2 / 4, room = uint32_t(per_cu) * uint32_t(prop.multiProcessorCount);
  return tiles <= room ? tiles : 0;
}
// xbuf: tiles x 256 carry words; flags: tiles words (+ the error word at flags[tiles]); epoch: > every epoch used before on these flags
hipError_t v3_launch_backfront(const DevPlan& pl, uint64_t* W, uint32_t a, uint32_t sub, uint64_t* xbuf, uint32_t* flags, uint32_t epoch, hipStream_t s) {
  const uint32_t tiles = pl.M2 / 4;
  hipLaunchKernelGGL(v3::k31_cols256_planes, dim3(tiles), dim3(2 * v3::kThreads), v3::kLdsBytes, s, pl, W, a, sub, xbuf, flags, flags + tiles, epoch);
  return hipGetLastError();
}
#endif
hipError_t v3_launch_front(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint32_t sub, uint64_t* W, hipStream_t s) {
  if (cols_planes(pl)) hipLaunchKernelGGL(v3::k1_cols256_planes, dim3(pl.M2 / 4), dim3(2 * v3::kThreads), v3::kLdsBytes, s, pl, digits, cbuf_in, sub, W);
  else hipLaunchKernelGGL(v3::k1_cols256, dim3(pl.M2 / 4), dim3(v3::kThreads), v3::kLdsBytes, s, pl, digits, cbuf_in, sub, W);
  return hipGetLastError();
}
hipError_t v3_launch_back(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, uint64_t scale, hipStream_t s) {
  if (cols_planes(pl)) hipLaunchKernelGGL(v3::k3_cols256_planes<false>, dim3(pl.M2 / 4), dim3(2 * v3::kThreads), v3::kLdsBytes, s, pl, W, digits, cbuf, a, scale, BackExt());
  else hipLaunchKernelGGL(v3::k3_cols256<false>, dim3(pl.M2 / 4), dim3(v3::kThreads), v3::kLdsBytes, s, pl, W, digits, cbuf, a, scale, BackExt());
  return hipGetLastError();
}
hipError_t v3_launch_back_ext(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& x, hipStream_t s) {
  if (cols_planes(pl)) hipLaunchKernelGGL(v3::k3_cols256_planes<true>, dim3(pl.M2 / 4), dim3(2 * v3::kThreads), v3::kLdsBytes, s, pl, W, digits, cbuf, a, uint64_t(1), x);
  else hipLaunchKernelGGL(v3::k3_cols256<true>, dim3(pl.M2 / 4), dim3(v3::kThreads), v3::kLdsBytes, s, pl, W, digits, cbuf, a, uint64_t(1), x);
  return hipGetLastError();
}

}  // namespace mi355
