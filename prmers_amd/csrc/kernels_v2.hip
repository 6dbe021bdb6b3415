// gfx950 kernels, register-resident radix-8 set ("v2") for the large power-of-two shapes.
//
// Same three sweeps as kernels.hip, but each work-group of 512 threads keeps its 4096-pair tile in
// registers (8 pairs per thread) and uses LDS only to exchange between radix-8 stages, and the stages
// are grouped into multiplication-free 64-point blocks:
//   * inside a block every root of unity is a power of two (omega_64 = 2^39): butterflies are add/sub
//     plus constant shifts (gfdft.hpp), the twiddle between the two radix-8 halves of a block is a
//     shift whose amount is uniform per wavefront (the thread->element maps below put the digit that
//     selects it in the wave index): the seam is compiled once per wave index with constant shifts and
//     selected by one scalar switch;
//   * only the seam between two blocks is a general GF(P) multiplication (one per element and
//     direction, from a universal omega_M table), against three per radix-4 level pair in the
//     reference's schedule (marin.cl:304-318: fwd4/bck4 with r1, r23.s0, r23.s1).
// Shapes served: rows M2 = 4096 (8.8.8.8), 8192 (2 x 4096 under one radix-2 level) and 2048 (4.8.8.8: two rows to a tile, or one row with
// one plane per thread -- k2_rows2048_planes below); columns M1 = 512 R (R.8.8.8, R = 1, 2, 4) with C = 8/R pairs per run.  The small
// transforms run on the radix-4 set (kernels_v3.hip), columns of 1280 on kernels_v5.hip, everything else on the generic set.
// Row order of the work buffer and digit layout are those of kernels.hip, so the two sets interoperate
// kernel by kernel (the multiplicand image layout differs: an engine uses one middle kernel for both
// set_multiplicand and mul).
// Value ranges: field values are canonical ([0, P)) except that (a) a negated zero may be P (gf.hpp,
// mul_pow2) and (b) sums marked LAZY (gfdft.hpp) may be any 64-bit representative; (b) only ever feeds a
// multiplication, a shift or the minuend / lazy-sum slot that gfdft.hpp names, and every kernel's last arithmetic step before a store is a
// multiplication or a canonical add/sub -- except the row kernel's last inverse stage, whose un-folded outputs the back sweep multiplies first --
// so nothing non-canonical other than P reaches a canonical-only operand or a digit.
//
// LDS exchanges (P2 = 16 B slots, slot map phys() of kernels_v2_common.hpp against bank conflicts):
//   writer "thread-major": slot t*8 + r          reader: slot j*512 + t'
//   writer "wave-major":   slot w*512 + r*64 + f(lane)   reader: slot j*512 + t'
// which is the digit permutation that hands each thread the 8 elements of its next radix-8.
#include "kernels_v2_common.hpp"

namespace mi355 {
namespace v2 {
// ---------------------------------------------------------------------------------------------
// middle, M2 = 4096 = 8.8.8.8.  Element index e = 512 d1 + 64 d2 + 8 d3 + d4.
//   S1 thread (d2|d3|d4) regs d1 -> k1 ; shift omega_64^(k1 d2)         [d2 = wave]
//   S2 thread (d3|d4|k1) regs d2 -> k2 ; general omega_4096^((k1+8k2)(8d3+d4))
//   S3 thread (d4|k1|k2) regs d3 -> k3 ; shift omega_64^(k3 d4)         [d4 = wave]
//   S4 thread (k3|k1|k2) regs d4 -> k4 ; X[k], k = k1 + 8 k2 + 64 k3 + 512 k4
// pointwise in registers, then the mirror image back to natural order.
// mode 0: square, 1: multiply by image Y, 2: forward only (writes the image).
// ---------------------------------------------------------------------------------------------
// H = 2 serves rows of 8192 with 1024 threads: one radix-2 level on top (element i and i + 4096; thread group
// h = 0 forms the sums, h = 1 the differences times omega_8192^i, each group loads both halves), then each
// group runs the 4096-point machinery on its own LDS half (2 x 72 KiB); output index k of group h is row
// frequency 2k + h.  The inverse ends with the mirror butterfly through LDS.
// RL = 1 serves rows of 2048 (round 3; the reference's sqr512 / forward1024 shapes, kernels/marin.cl:1190,1517): a tile is two adjacent rows,
// the first-stage registers 4 r .. 4 r + 3 belong to row r of the tile, the first stage is a radix-4 per row (32-point blocks: seam
// omega_32^(k1' d2), still uniform per wave), the table seam reads omega_2048^((k1' + 4 k2)(8 d3 + d4)) (plan.hpp builds S2r for this shape)
// and everything after it is the 4096-point code with the row carried in the top bit of the k1 field: X[k], k = k1' + 4 k2 + 32 k3 + 256 k4
// of row 2 tile + (k1 >> 2).  n = 2^21: 0.0645 -> 0.0577 ms, n = 5 2^20 (p ~ 100 M): 0.130 -> 0.120 ms (same-box A/B against the generic rows,
// MI355_TUNE bit 6).  (Four rows of 1024 to a tile -- RL = 2, a radix-2 first stage -- were built and measured too: n = 5 2^19 has only
// 320 such tiles for 256 CUs and lost 2.4 % against the generic rows; not kept.)
template <int RL, int W, bool INV>
__device__ __forceinline__ void seam_rows_const(P2 (&x)[8]) {   // x[(8 >> RL) r + k1'] *= omega_(64 >> RL)^(+-k1' W) = 2^(+-39 2^RL k1' W)
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    constexpr int R1 = 8 >> RL;
    if ((k & (R1 - 1)) == 0) continue;
    const unsigned f = ((1u << RL) * gf::LOG2_W64 * unsigned(k & (R1 - 1)) * unsigned(W)) % 192u;
    const unsigned sh = INV ? (192u - f) % 192u : f;
    x[k] = {gf::mul_pow2(x[k].a, sh), gf::mul_pow2(x[k].b, sh)};
  }
}
template <int RL, bool INV>
__device__ __forceinline__ void seam_rows(P2 (&x)[8], uint32_t wave) {
  switch (wave) {
    case 0: break;
    case 1: seam_rows_const<RL, 1, INV>(x); break;
    case 2: seam_rows_const<RL, 2, INV>(x); break;
    case 3: seam_rows_const<RL, 3, INV>(x); break;
    case 4: seam_rows_const<RL, 4, INV>(x); break;
    case 5: seam_rows_const<RL, 5, INV>(x); break;
    case 6: seam_rows_const<RL, 6, INV>(x); break;
    default: seam_rows_const<RL, 7, INV>(x); break;
  }
}
template <int RL, bool INV>
__device__ __forceinline__ void dft_rows_first(P2 (&x)[8]) {   // the first-stage transform of every row of the tile, both planes
  static_assert(RL == 1, "two rows to a tile");   // a radix-4 on registers 0 .. 3 and 4 .. 7
  dft4<INV>(x[0].a, x[1].a, x[2].a, x[3].a); dft4<INV>(x[0].b, x[1].b, x[2].b, x[3].b);
  dft4<INV>(x[4].a, x[5].a, x[6].a, x[7].a); dft4<INV>(x[4].b, x[5].b, x[6].b, x[7].b);
}

template <int mode, int H, int RL = 0>
__global__ void __launch_bounds__(512 * H, 4) k2_rows4096(DevPlan pl, const uint64_t* __restrict__ Win, const uint64_t* __restrict__ Yimg,
                                                          uint64_t* __restrict__ Wout, uint32_t sub) {
  static_assert(RL == 0 || (RL == 1 && H == 1), "two rows to a tile only with one tile per work-group");
  constexpr bool HALF = (RL > 0);
  constexpr int HH = H;                 // 4096-pair tiles per work-group
  constexpr int R1 = 8 >> RL;           // first-stage radix
  const uint32_t h = (H == 2) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 9) : 0u;
  if (H == 1) boost_if_late(pl.boost_rows);
  P2* X = reinterpret_cast<P2*>(smem_v2) + h * kLdsSlots;
  const uint32_t t = threadIdx.x & 511, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 7);
  const uint32_t tile = PROBE_BLOCK(pl);
  const uint32_t row = HALF ? (tile << RL) + (lane >> (6 - RL)) : tile;    // several rows: the row this thread's S4 outputs (and pointwise words) belong to
  PROBE_BEGIN(pl)
  const P2* in = reinterpret_cast<const P2*>(Win) + size_t(tile) * (4096 * HH);
  P2* out = reinterpret_cast<P2*>(Wout) + size_t(tile) * (4096 * HH) + h * 4096;
  P2 x[8];
  // table words of the pointwise stage, requested first: their latency hides behind the forward transform
  // S4 thread (k3|k1|k2): frequency base k1 + 8 k2 + 64 k3 (several rows: k1' + R1 k2 + 8 R1 k3 with k1' = k1 & (R1 - 1))
  const uint32_t kb = ((lane >> 3) & uint32_t(R1 - 1)) + R1 * (lane & 7) + 8 * R1 * wave;
  const uint32_t blk = row / pl.L1, qq = row - blk * pl.L1;       // column-DFT slot -> frequency (kernels.hip freq1)
  const uint32_t k1row = col_label(pl, blk, pl.logL1 ? (__brev(qq) >> (32 - pl.logL1)) : 0u);
  const uint64_t erho = rho_exponent(pl, k1row, HH * kb + h);   // row frequency of X[kb]: H kb + h
  const uint64_t rho_lo = pl.TWlo[erho & ((1u << pl.twh) - 1)], rho_hi = pl.TWhi[erho >> pl.twh];

  // ---- forward ----
  // deferred small subtraction (LL's -2) on a front image: digit 0 has weight 1 and reaches column 0,
  // plane a of every row unchanged
  if (H == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = w_load(&in[512 * j + t]);
    if (sub != 0 && t == 0) {
#pragma unroll
      for (int r = 0; r < (1 << RL); ++r) x[R1 * r].a = gf::sub(x[R1 * r].a, uint64_t(sub));   // element 0 of every row of the tile
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      P2 lo = in[512 * j + t];
      const P2 hi = in[4096 + 512 * j + t];
      if (j == 0 && sub != 0 && t == 0) lo.a = gf::sub(lo.a, uint64_t(sub));
      if (h == 0) x[j] = {gf::add(lo.a, hi.a), gf::add(lo.b, hi.b)};
      else x[j] = p2_mul(P2{gf::sub(lo.a, hi.a), gf::sub(lo.b, hi.b)}, pl.UT2[512 * j + t]);
    }
  }
  if constexpr (HALF) { dft_rows_first<RL, false>(x); seam_rows<RL, false>(x, wave); }
  else {
    dft8p<false, 1>(x);   // outputs 1..7 are shifted next, output 0 is not
    seam64<false, true>(x, wave);
  }
  uint64_t sw[8];   // seam twiddles: loaded before the exchange so that their latency hides behind it
  {
    const uint32_t k1 = t & 7, b = t >> 3;
    const uint64_t* __restrict__ tw = pl.S2r + b * 64 + k1 * 8;   // [b][k1][k2]: 64 contiguous bytes per thread
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) sw[k2] = tw[k2];
  }
  EXCH_THREAD_MAJOR_TO_STRIDED(X, x, t)
  dft8p<false, 2>(x);   // all outputs are multiplied next
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) x[k2] = p2_mul(x[k2], sw[k2]);
  EXCH_THREAD_MAJOR_TO_STRIDED(X, x, t)
  dft8p<false, 1>(x);
  seam64<false, true>(x, wave);
  lds_barrier();
#pragma unroll
  for (int k = 0; k < 8; ++k) X[phys(wave * 512 + k * 64 + lane)] = x[k];
  lds_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = X[phys(j * 512 + t)];
  dft8p<false, 2>(x);   // mode 2 stores them for a later multiplication, the pointwise stage multiplies

  if (mode == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) out[512 * j + t] = x[j];
    return;
  }

  PROBE_MID(pl)
  // ---- pointwise: reg k4 holds X[kb + 512 k4]; rho = omega_m^(k1row + M1 k) = rho0 * omega_8^k4 ----
  {
    const uint64_t rho0 = gf::mul(rho_lo, rho_hi);
    const P2* Y = reinterpret_cast<const P2*>(Yimg) + size_t(tile) * (4096 * HH) + h * 4096;
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      // omega_8^k4 = 2^(120 k4): +1, -2^24, +2^48, -2^72, -1, +2^24, -2^48, +2^72
      const unsigned sh = 24u * (k4 & 3);
      const bool neg = (k4 == 1) || (k4 == 3) || (k4 == 4) || (k4 == 6);
      const P2 u = x[k4];
      P2 r;
      if (mode == 0) {   // (u0 + u1 t)^2 mod (t^2 - rho), marin.cl:379-384
        const uint64_t q = gf::mul_pow2(gf::mul(gf::sqr(u.b), rho0), sh);
        const uint64_t s0 = gf::sqr(u.a);
        r.a = neg ? gf::sub(s0, q) : gf::add(s0, q);
        r.b = gf::dbl(gf::mul(u.b, u.a));   // (u.a may be an un-folded sum: double the product, not the operand)
      } else {           // marin.cl:387-392
        const P2 y = Y[512 * k4 + t];
        const uint64_t q = gf::mul_pow2(gf::mul(gf::mul(u.b, y.b), rho0), sh);
        const uint64_t s0 = gf::mul(u.a, y.a);
        r.a = neg ? gf::sub(s0, q) : gf::add(s0, q);
        r.b = gf::add(gf::mul(u.a, y.b), gf::mul(u.b, y.a));
      }
      x[k4] = r;
    }
  }

  // ---- inverse (mirror) ----
  dft8p<true>(x);
  lds_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) X[phys(j * 512 + t)] = x[j];
  lds_barrier();
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = X[phys(wave * 512 + k * 64 + lane)];
  seam64<true>(x, wave);
  dft8p<true, 2>(x);   // exchanged, then multiplied by the seam twiddles
  {
    const uint32_t k1 = t & 7, b = t >> 3;
    const uint64_t* __restrict__ tw = pl.S2ri + b * 64 + k1 * 8;   // [b][k1][k2]: 64 contiguous bytes per thread
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) sw[k2] = tw[k2];
  }
  EXCH_STRIDED_TO_THREAD_MAJOR(X, x, t)
  if (H == 1) __builtin_amdgcn_s_setprio(0);   // boosted groups: back to normal for the last stages (measured best drop point)
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) x[k2] = p2_mul(x[k2], sw[k2]);
  dft8p<true>(x);
  EXCH_STRIDED_TO_THREAD_MAJOR(X, x, t)
  if constexpr (HALF) { seam_rows<RL, true>(x, wave); dft_rows_first<RL, true>(x); }
  else {
    seam64<true>(x, wave);
    if (H == 1) dft8p<true, 2>(x);   // stored for a back sweep, which multiplies by its twiddle first
    else dft8p<true>(x);
  }
  if (H == 2) {   // mirror of the top radix-2 level: lo = A + B w^-i, hi = A - B w^-i (A from group 0, B from group 1)
    if (h == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const uint32_t i = 512 * j + t; x[j] = p2_mul(x[j], pl.UT2[i ? 8192 - i : 0]); }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) X[phys(j * 512 + t)] = x[j];
    __syncthreads();
    const P2* Xo = reinterpret_cast<const P2*>(smem_v2) + (1 - h) * kLdsSlots;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const P2 o = Xo[phys(j * 512 + t)];
      x[j] = (h == 0) ? P2{gf::add(x[j].a, o.a), gf::add(x[j].b, o.b)} : P2{gf::sub(o.a, x[j].a), gf::sub(o.b, x[j].b)};
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) w_store(&out[512 * j + t], x[j]);
  PROBE_END(pl)
}

// ---------------------------------------------------------------------------------------------
// Rows of 2048, one row to a tile, ONE PLANE PER THREAD (round 4).  The two planes of a pair go through the same GF(P)-linear transform and
// meet only in the pointwise step, so the "two rows to a tile" kernel above runs unchanged on 64-bit words with the row bit read as the plane
// bit: a tile is one row of 2048 pairs = 4096 words, 512 threads x 8 words, the first-stage registers 4 pl .. 4 pl + 3 are plane pl of the
// four pairs a thread loads (16-byte loads and stores as before), after the last forward exchange plane = lane >> 5, and the pointwise step
// trades words with lane ^ 32 (ds_bpermute: crossbar only, no LDS memory).  Half the registers and half the serial work per thread of the
// pair form, twice the tiles: where two rows to a tile leave CUs without a tile (n = 2^20: 128 tiles) or with one group of eight waves
// (n = 2^21) this fills the chip.  LDS: 32 KiB of 8-byte slots, map i ^ ((i >> 3) & 31) (conflict-free for the 16-lane / 32-lane groups
// 64-bit accesses are served in: thread-major, strided and wave-major orders alike).
// Multiplicand image (mode 2 -> mode 1): word 512 k4 + t of the row, i.e. plane-major inside 64-word runs -- private to this kernel.
__device__ __forceinline__ uint32_t phys8(uint32_t i) { return i ^ ((i >> 3) & 31u); }
__device__ __forceinline__ uint64_t swap32(uint64_t v, uint32_t partner_byte) {   // the word lane ^ 32 holds
  const uint32_t lo = uint32_t(__builtin_amdgcn_ds_bpermute(int(partner_byte), int(uint32_t(v))));
  const uint32_t hi = uint32_t(__builtin_amdgcn_ds_bpermute(int(partner_byte), int(uint32_t(v >> 32))));
  return (uint64_t(hi) << 32) | lo;
}
template <int W, bool INV>
__device__ __forceinline__ void seam64w_const(uint64_t (&x)[8]) {
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const unsigned f = (gf::LOG2_W64 * unsigned(k) * unsigned(W)) % 192u;
    x[k] = gf::mul_pow2(x[k], INV ? (192u - f) % 192u : f);
  }
}
template <bool INV, bool FOLD0 = false>
__device__ __forceinline__ void seam64w(uint64_t (&x)[8], uint32_t wave) {
  switch (wave) {
    case 0:
      if (FOLD0) { x[1] = gf::fold(x[1]); x[2] = gf::fold(x[2]); x[3] = gf::fold(x[3]); x[5] = gf::fold(x[5]); }
      break;
    case 1: seam64w_const<1, INV>(x); break;
    case 2: seam64w_const<2, INV>(x); break;
    case 3: seam64w_const<3, INV>(x); break;
    case 4: seam64w_const<4, INV>(x); break;
    case 5: seam64w_const<5, INV>(x); break;
    case 6: seam64w_const<6, INV>(x); break;
    default: seam64w_const<7, INV>(x); break;
  }
}
template <int W, bool INV>
__device__ __forceinline__ void seam32w_const(uint64_t (&x)[8]) {   // x[4 pl + k1'] *= omega_32^(+-k1' W)
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    if ((k & 3) == 0) continue;
    const unsigned f = (2u * gf::LOG2_W64 * unsigned(k & 3) * unsigned(W)) % 192u;
    x[k] = gf::mul_pow2(x[k], INV ? (192u - f) % 192u : f);
  }
}
template <bool INV>
__device__ __forceinline__ void seam32w(uint64_t (&x)[8], uint32_t wave) {
  switch (wave) {
    case 0: break;
    case 1: seam32w_const<1, INV>(x); break;
    case 2: seam32w_const<2, INV>(x); break;
    case 3: seam32w_const<3, INV>(x); break;
    case 4: seam32w_const<4, INV>(x); break;
    case 5: seam32w_const<5, INV>(x); break;
    case 6: seam32w_const<6, INV>(x); break;
    default: seam32w_const<7, INV>(x); break;
  }
}
#define EXCHW_THREAD_MAJOR_TO_STRIDED(X, x, t)                      \
  lds_barrier();                                                    \
  _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) X[phys8((t) * 8 + r_)] = x[r_]; \
  lds_barrier();                                                    \
  _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) x[j_] = X[phys8(j_ * 512 + (t))];
#define EXCHW_STRIDED_TO_THREAD_MAJOR(X, x, t)                      \
  lds_barrier();                                                    \
  _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) X[phys8(j_ * 512 + (t))] = x[j_]; \
  lds_barrier();                                                    \
  _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) x[r_] = X[phys8((t) * 8 + r_)];

constexpr uint32_t kLdsBytesPlanes = 4096 * 8;

template <int mode>
__global__ void __launch_bounds__(512, 4) k2_rows2048_planes(DevPlan pl, const uint64_t* __restrict__ Win, const uint64_t* __restrict__ Yimg,
                                                             uint64_t* __restrict__ Wout, uint32_t sub) {
  uint64_t* X = reinterpret_cast<uint64_t*>(smem_v2);
  const uint32_t t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t row = blockIdx.x;
  const P2* in = reinterpret_cast<const P2*>(Win) + size_t(row) * 2048;
  P2* out = reinterpret_cast<P2*>(Wout) + size_t(row) * 2048;
  uint64_t x[8];
  // pointwise stage: thread (k3|pl|k1'|k2) holds plane pl = lane >> 5 of X[kb + 256 k4], kb = k1' + 4 k2 + 32 k3
  const uint32_t pln = lane >> 5, partner = ((lane ^ 32u) << 2);
  const uint32_t kb = ((lane >> 3) & 3u) + 4 * (lane & 7) + 32 * wave;
  const uint32_t blk = row / pl.L1, qq = row - blk * pl.L1;       // column-DFT slot -> frequency (kernels.hip freq1)
  const uint32_t k1row = col_label(pl, blk, pl.logL1 ? (__brev(qq) >> (32 - pl.logL1)) : 0u);
  const uint64_t erho = rho_exponent(pl, k1row, kb);
  const uint64_t rho_lo = pl.TWlo[erho & ((1u << pl.twh) - 1)], rho_hi = pl.TWhi[erho >> pl.twh];

  // ---- forward ----
#pragma unroll
  for (int d = 0; d < 4; ++d) { const P2 v = in[512 * d + t]; x[d] = v.a; x[4 + d] = v.b; }
  if (sub != 0 && t == 0) x[0] = gf::sub(x[0], uint64_t(sub));   // LL's -2: element 0, plane a
  dft4<false>(x[0], x[1], x[2], x[3]); dft4<false>(x[4], x[5], x[6], x[7]);
  seam32w<false>(x, wave);
  uint64_t sw[8];
  {
    const uint32_t k1 = t & 7, b = t >> 3;
    const uint64_t* __restrict__ tw = pl.S2r + b * 64 + k1 * 8;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) sw[k2] = tw[k2];
  }
  EXCHW_THREAD_MAJOR_TO_STRIDED(X, x, t)
  gf::dft8<false, 2>(x);
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) x[k2] = gf::mul(x[k2], sw[k2]);
  EXCHW_THREAD_MAJOR_TO_STRIDED(X, x, t)
  gf::dft8<false, 1>(x);
  seam64w<false, true>(x, wave);
  lds_barrier();
#pragma unroll
  for (int k = 0; k < 8; ++k) X[phys8(wave * 512 + k * 64 + lane)] = x[k];
  lds_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = X[phys8(j * 512 + t)];
  gf::dft8<false, 2>(x);

  if (mode == 2) {
    uint64_t* img = Wout + size_t(row) * 4096;
#pragma unroll
    for (int j = 0; j < 8; ++j) img[512 * j + t] = x[j];   // (un-folded sums: a multiplication reads them)
    return;
  }

  // ---- pointwise: (a + b t)^2 or (a + b t)(ya + yb t) mod (t^2 - rho), rho = rho0 omega_8^k4; plane a computes and keeps the t^0 word ----
  {
    const uint64_t rho0 = gf::mul(rho_lo, rho_hi);
    const uint64_t* Y = Yimg + size_t(row) * 4096;
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      const unsigned sh = 24u * (k4 & 3);
      const bool neg = (k4 == 1) || (k4 == 3) || (k4 == 4) || (k4 == 6);
      const uint64_t w = x[k4];
      uint64_t keep, send;
      if (mode == 0) {
        const uint64_t o = swap32(w, partner);
        keep = gf::sqr(w);                                            // a^2 | b^2
        send = gf::mul(pln ? keep : w, pln ? rho0 : o);               // a b | b^2 rho0
        if (pln) send = gf::mul_pow2(send, sh);
      } else {
        const uint64_t y = Y[512 * k4 + t], yo = Y[512 * k4 + (t ^ 32u)];
        keep = gf::mul(w, y);                                         // a ya | b yb
        send = gf::mul(w, yo);                                        // a yb | b ya
        if (pln) { const uint64_t q = gf::mul_pow2(gf::mul(keep, rho0), sh); keep = send; send = q; }   // plane b keeps b ya, sends b yb rho
      }
      const uint64_t recv = swap32(send, partner);
      if (mode == 0) x[k4] = pln ? gf::dbl(recv) : (neg ? gf::sub(keep, recv) : gf::add(keep, recv));
      else x[k4] = pln ? gf::add(keep, recv) : (neg ? gf::sub(keep, recv) : gf::add(keep, recv));
    }
  }

  // ---- inverse (mirror) ----
  gf::dft8<true>(x);
  lds_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) X[phys8(j * 512 + t)] = x[j];
  lds_barrier();
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = X[phys8(wave * 512 + k * 64 + lane)];
  seam64w<true>(x, wave);
  gf::dft8<true, 2>(x);
  {
    const uint32_t k1 = t & 7, b = t >> 3;
    const uint64_t* __restrict__ tw = pl.S2ri + b * 64 + k1 * 8;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) sw[k2] = tw[k2];
  }
  EXCHW_STRIDED_TO_THREAD_MAJOR(X, x, t)
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) x[k2] = gf::mul(x[k2], sw[k2]);
  gf::dft8<true>(x);
  EXCHW_STRIDED_TO_THREAD_MAJOR(X, x, t)
  seam32w<true>(x, wave);
  dft4<true>(x[0], x[1], x[2], x[3]); dft4<true>(x[4], x[5], x[6], x[7]);
#pragma unroll
  for (int d = 0; d < 4; ++d) out[512 * d + t] = P2{x[d], x[4 + d]};
}

// ---------------------------------------------------------------------------------------------
// Column tiles, M1 = 512 R = R.8.8.8 with R in {1, 2, 4} and C = 8 / R pairs per run (tile = 4096 pairs).
// Tile element (i1, c), i1 = 512 d1 + 64 d2 + 8 d3 + d4 (d1 < R).  kc = k1 C + c is a 3-bit register /
// thread field throughout.
// front_tile (digits -> work buffer):
//   S1 thread (d2|d3|d4) regs (d1,c): R whole runs of 2C digits -> weight -> DFT_R -> k1 ; shift omega_8R^(k1 d2)
//   S2 thread (d3|d4|k1|c) regs d2 -> k2 ; general omega_M1^((k1 + R k2)(8d3+d4))
//   S3 thread (d4|k1|c|k2) regs d3 -> k3 ; shift omega_64^(k3 d4)
//   S4 thread (k3|k1|k2|c) regs d4 -> k4 ; k1col = k1 + R k2 + 8R k3 + 64R k4
//   then the four-step twiddle omega_m^(i2 k1col) * TB (geometric in k4: one chain multiply per pair)
//   and the store to work-buffer row bitrev(k1col), column i2 = C T + c.
// back_tile is the mirror image, followed by unweight and the sequential carry of the thread's R runs.
// ---------------------------------------------------------------------------------------------

template <int R> struct ColShape {
  static constexpr int C = 8 / R;                      // pairs per run
  static constexpr int LC = (C == 8) ? 3 : (C == 4) ? 2 : 1;
  static constexpr int LR = 3 - LC;
  static constexpr int M1 = 512 * R, LM = 9 + LR, ND = 2 * C;   // digits per run
};

// S1 of the front: DFT_R over d1 (register slots d1 C + c -> k1 C + c), then x[k1 C + c] *= omega_8R^(k1 W)
template <int R, int W>
__device__ __forceinline__ void stage_r_fwd_const(P2 (&x)[8]) {
  constexpr int C = 8 / R;
  if (R == 2) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const P2 u = x[c], v = x[C + c];
      x[c] = {gf::add(u.a, v.a), gf::add(u.b, v.b)};
      x[C + c] = {gf::sub(u.a, v.a), gf::sub(u.b, v.b)};
    }
  } else if (R == 4) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      dft4<false>(x[c].a, x[C + c].a, x[2 * C + c].a, x[3 * C + c].a);
      dft4<false>(x[c].b, x[C + c].b, x[2 * C + c].b, x[3 * C + c].b);
    }
  }
#pragma unroll
  for (int k1 = 1; k1 < R; ++k1) {
    const unsigned s = ((8u / unsigned(R)) * gf::LOG2_W64 * unsigned(k1) * unsigned(W)) % 192u;   // omega_8R = omega_64^(8/R)
#pragma unroll
    for (int c = 0; c < C; ++c) x[k1 * C + c] = {gf::mul_pow2(x[k1 * C + c].a, s), gf::mul_pow2(x[k1 * C + c].b, s)};
  }
}
// last stage of the back: the inverse seam, then the inverse DFT_R.  For R = 2 a negative sign (shift >= 96)
// is absorbed by swapping the sum and the difference.
template <int R, int W>
__device__ __forceinline__ void stage_r_inv_const(P2 (&x)[8]) {
  constexpr int C = 8 / R;
  if (R == 2) {
    const unsigned f = (192u - (4u * gf::LOG2_W64 * unsigned(W)) % 192u) % 192u;
    const bool neg = f >= 96u;
    const unsigned s = neg ? f - 96u : f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const P2 u = x[c];
      const P2 v = {gf::mul_pow2(x[C + c].a, s), gf::mul_pow2(x[C + c].b, s)};
      const P2 sum = {gf::add(u.a, v.a), gf::add(u.b, v.b)}, dif = {gf::sub(u.a, v.a), gf::sub(u.b, v.b)};
      x[c] = neg ? dif : sum;
      x[C + c] = neg ? sum : dif;
    }
  } else if (R == 4) {
#pragma unroll
    for (int k1 = 1; k1 < R; ++k1) {
      const unsigned s = (192u - ((8u / unsigned(R)) * gf::LOG2_W64 * unsigned(k1) * unsigned(W)) % 192u) % 192u;
#pragma unroll
      for (int c = 0; c < C; ++c) x[k1 * C + c] = {gf::mul_pow2(x[k1 * C + c].a, s), gf::mul_pow2(x[k1 * C + c].b, s)};
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      dft4<true>(x[c].a, x[C + c].a, x[2 * C + c].a, x[3 * C + c].a);
      dft4<true>(x[c].b, x[C + c].b, x[2 * C + c].b, x[3 * C + c].b);
    }
  }
}

// digits of the thread's R runs (i1 = 512 d1 + t) of tile T  ->  work buffer (forward columns).
// sub: small constant to subtract at digit 0 of the whole number, in the field (LL's x^2 - 2).
template <int R>
__device__ __forceinline__ void front_tile(const DevPlan& pl, P2* X, uint32_t T, uint32_t t, uint32_t lane, uint32_t wave,
                                           const uint32_t (&dg)[R][16 / R], uint32_t di, uint32_t sub, uint64_t* __restrict__ Wout) {
  using S = ColShape<R>;
  constexpr int C = S::C, LC = S::LC;
  P2 x[8];
  const uint32_t nowrap = ~di;   // bit 2 idx + 1 of di: the weight exponents of digit idx wrapped
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) {
    const uint32_t i1 = 512 * d1 + t;
    // odd digits: exponent split SA[M1 + i1] + SB[2 i2] (plan.hpp), hence their own TA entry
    const uint64_t tah = pl.TAh[i1], tah1 = pl.TAh[S::M1 + i1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      // weight TA*TB, halved when the exponents wrap: the halving is moved onto TA (once per run) and
      // the un-wrapped digits are doubled instead (digits are < 2^21, the product stays a mul_u32)
      const int idx = d1 * S::ND + 2 * c;
      const uint64_t a0 = gf::mul_u32(tah, dg[d1][2 * c] << ((nowrap >> (2 * idx + 1)) & 1u));
      const uint64_t a1 = gf::mul_u32(tah1, dg[d1][2 * c + 1] << ((nowrap >> (2 * idx + 3)) & 1u));
      x[C * d1 + c] = {a0, a1};
    }
    if (sub != 0 && T == 0 && i1 == 0) x[0].a = gf::sub(x[0].a, uint64_t(sub));   // digit 0 has weight 1
  }
#define MI355_CALL(W) stage_r_fwd_const<R, W>(x)
  MI355_SWITCH8(wave, MI355_CALL)
#undef MI355_CALL
  uint64_t sw[8];   // seam twiddles, requested before the exchange that hides their latency
  {
    const uint32_t k1 = (t & 7) >> LC, b = t >> 3;
    const uint64_t* __restrict__ tw = pl.S1r + b * (8 * R) + k1 * 8;   // [b][k1][k2]
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) sw[k2] = tw[k2];
  }
  EXCH_THREAD_MAJOR_TO_STRIDED(X, x, t)
  dft8p<false, 2>(x);
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) x[k2] = p2_mul(x[k2], sw[k2]);
  // four-step twiddle ingredients for the last stage (thread (k3|k1|k2|c)), requested two exchanges early
  const uint32_t fc = t & (C - 1), fkb = ((t >> (3 + LC)) & (R - 1)) + R * ((t >> LC) & 7) + 8 * R * (t >> 6), fi2 = C * T + fc;
  // chain start omega_m^(i2 kb) TB[2 i2] and ratio omega_m^(64R i2), ready-made per (tile, thread) / per column
  const uint64_t fca0 = pl.F0f[size_t(T) * 512 + t], fB = pl.FBf[fi2];
  EXCH_THREAD_MAJOR_TO_STRIDED(X, x, t)
  dft8p<false, 1>(x);
  seam64<false, true>(x, wave);
  {
    // lane = (k1 C + c) 8 + k2  ->  slot offset (k1 | k2 | c)
    const uint32_t off = ((lane >> (3 + LC)) << (3 + LC)) | ((lane & 7) << LC) | ((lane >> 3) & (C - 1));
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 8; ++k) X[phys(wave * 512 + k * 64 + off)] = x[k];
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = X[phys(j * 512 + t)];
  }
  dft8p<false, 2>(x);   // four-step twiddle next
  {
    const uint32_t kb = fkb, i2 = fi2;
    const uint64_t B = fB;
    uint64_t ca = fca0;   // one chain for both digits of a pair (plan.hpp: SA/TA second half)
    const uint32_t row0 = __brev(kb) >> (32 - S::LM);   // bitrev(kb): its low 3 bits are zero
    P2* W = reinterpret_cast<P2*>(Wout);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t rj = ((j & 1) << 2) | (j & 2) | (j >> 2);   // bitrev3(j)
      w_store(&W[size_t(row0 + rj) * pl.M2 + i2], P2{gf::mul(x[j].a, ca), gf::mul(x[j].b, ca)});
      if (j < 7) ca = gf::mul(ca, B);
    }
  }
}

// work buffer -> digits of the thread's R runs of tile T (inverse columns, unweight, x a, carry).
// scale: extra field factor (1, or M2 when the input is a front image rather than a middle output).
// carry0[d1]: carry entering run d1 (strong: propagated through the whole run); cout[d1]: carry leaving it.
// ADD: ad[d1][k] (the digits of another residue's runs, pending carries already folded in) join the carry chain (mul_add)
template <int R, bool ADD = false>
__device__ __forceinline__ void back_tile(const DevPlan& pl, P2* X, uint32_t T, uint32_t t, uint32_t lane, uint32_t wave,
                                          const uint64_t* __restrict__ Win, uint32_t a, uint64_t scale,
                                          const uint64_t (&carry0)[R], uint32_t di, uint32_t (&dg)[R][16 / R], uint64_t (&cout)[R],
                                          const uint32_t (*ad)[16 / R] = nullptr) {
  using S = ColShape<R>;
  constexpr int C = S::C, LC = S::LC;
  P2 x[8];
  {
    const uint32_t c = t & (C - 1), k2 = (t >> LC) & 7, k1 = (t >> (3 + LC)) & (R - 1), k3 = t >> 6;
    const uint32_t kb = k1 + R * k2 + 8 * R * k3;
    const uint32_t i2 = C * T + c;
    // chain start omega_m^-(i2 kb) TBi[2 i2] and ratio omega_m^-(64R i2), ready-made (k_build_f0)
    uint64_t ca = pl.F0i[size_t(T) * 512 + t];   // one chain for both digits of a pair (plan.hpp: SA/TA second half)
    const uint64_t B = pl.FBi[i2];
    if (scale != 1) ca = gf::mul(ca, scale);
    const uint32_t row0 = __brev(kb) >> (32 - S::LM);
    const P2* W = reinterpret_cast<const P2*>(Win);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t rj = ((j & 1) << 2) | (j & 2) | (j >> 2);
      x[j] = w_load(&W[size_t(row0 + rj) * pl.M2 + i2]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x[j] = {gf::mul(x[j].a, ca), gf::mul(x[j].b, ca)};
      if (j < 7) ca = gf::mul(ca, B);
    }
  }
  dft8p<true>(x);
  {
    const uint32_t off = ((lane >> (3 + LC)) << (3 + LC)) | ((lane & 7) << LC) | ((lane >> 3) & (C - 1));
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) X[phys(j * 512 + t)] = x[j];
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = X[phys(wave * 512 + k * 64 + off)];
  }
  seam64<true>(x, wave);
  dft8p<true, 2>(x);
  uint64_t sw[8];
  {
    const uint32_t k1 = (t & 7) >> LC, b = t >> 3;
    const uint64_t* __restrict__ tw = pl.S1ri + b * (8 * R) + k1 * 8;   // [b][k1][k2]
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) sw[k2] = tw[k2];
  }
  EXCH_STRIDED_TO_THREAD_MAJOR(X, x, t)
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) x[k2] = p2_mul(x[k2], sw[k2]);
  dft8p<true>(x);
  // unweighting tables of the carry phase (thread (d2|d3|d4): runs i1 = 512 d1 + t), requested one exchange early;
  // odd digits take theirs from the second half of SA / TAi
  // (the doubled entries too, from their own table -- except with four runs a thread, where 16 more registers across the exchange would spill)
  constexpr bool PRE2 = (R <= 2);
  uint64_t btai_e[R], btai_o[R], btai2_e[R], btai2_o[R];
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) {
    btai_e[d1] = pl.TAi[512 * d1 + t]; btai_o[d1] = pl.TAi[S::M1 + 512 * d1 + t];
    if (PRE2) { btai2_e[d1] = pl.TAi2[512 * d1 + t]; btai2_o[d1] = pl.TAi2[S::M1 + 512 * d1 + t]; }
  }
  EXCH_STRIDED_TO_THREAD_MAJOR(X, x, t)
#define MI355_CALL(W) stage_r_inv_const<R, W>(x)
  MI355_SWITCH8(wave, MI355_CALL)
#undef MI355_CALL
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) {
    const uint64_t tai_e = btai_e[d1], tai_o = btai_o[d1];
    const uint64_t tai2_e = PRE2 ? btai2_e[d1] : gf::dbl(tai_e), tai2_o = PRE2 ? btai2_o[d1] : gf::dbl(tai_o);
    uint64_t carry = carry0[d1];
#pragma unroll
    for (int k = 0; k < S::ND; ++k) {
      const uint32_t bits = di >> (2 * (d1 * S::ND + k));   // digit-info table: width - q, wrap
      const uint32_t width = pl.q + (bits & 1u);
      const bool wrap = (bits & 2u) != 0;
      const P2 v = x[C * d1 + (k >> 1)];
      const uint64_t u = (k & 1) ? gf::mul(v.b, wrap ? tai2_o : tai_o) : gf::mul(v.a, wrap ? tai2_e : tai_e);   // wrapped exponents: weight was halved
      const uint64_t mask = (uint64_t(1) << width) - 1;   // adc_mul, marin.cl:194-201
      if (a == 1) {               // the common case (uniform): no 64-bit multiplies
        const uint64_t r = u + carry + (ADD ? ad[d1][k] : 0u);   // u < 2^63 by the size rule (ibdwt.h:28-30), carry < 2^48
        dg[d1][k] = __builtin_amdgcn_ubfe(uint32_t(r), 0u, width);   // width < 32: one bit-field extract
        carry = r >> width;
      } else {
        const uint64_t dlo = u & mask, chi = u >> width;
        const uint64_t r = dlo * a + carry + (ADD ? ad[d1][k] : 0u);
        dg[d1][k] = uint32_t(r & mask);
        carry = (r >> width) + chi * a;
      }
    }
    cout[d1] = carry;
  }
}

// digit runs <-> registers: run (T, i1) is ND = 2C consecutive u32
template <int R>
__device__ __forceinline__ void load_run(const uint32_t* __restrict__ digits, uint32_t T, uint32_t i1, uint32_t (&d)[16 / R]) {
  constexpr int Q = 4 / R;   // uint4 per run
  const uint4* src = reinterpret_cast<const uint4*>(digits) + (size_t(T) * (512 * R) + i1) * Q;
#pragma unroll
  for (int q = 0; q < Q; ++q) { const uint4 v = src[q]; d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w; }
}
template <int R>
__device__ __forceinline__ void store_run(uint32_t* __restrict__ digits, uint32_t T, uint32_t i1, const uint32_t (&d)[16 / R]) {
  constexpr int Q = 4 / R;
  uint4* dst = reinterpret_cast<uint4*>(digits) + (size_t(T) * (512 * R) + i1) * Q;
#pragma unroll
  for (int q = 0; q < Q; ++q) dst[q] = make_uint4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
}

// front sweep: digits (+ deferred run carries cbuf_in, nullable; + deferred subtraction) -> work buffer
template <int R>
__global__ void __launch_bounds__(512, 4) k1_cols(DevPlan pl, const uint32_t* __restrict__ digits, const uint64_t* __restrict__ cbuf_in,
                                                  uint32_t sub, uint64_t* __restrict__ Wout) {
  P2* X = reinterpret_cast<P2*>(smem_v2);
  const uint32_t t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // tile order: with two pairs per run (R = 4) four tiles share each 128-byte line of the work buffer, and keeping
  // them on one XCD lets its L2 merge the 32-byte pieces (n = 2^24: front sweep 97 -> 89 us); with runs of eight pairs (R = 1) the
  // XCD-contiguous order wins too (n = 2^22: 20.3 -> 19.0 us), with four (R = 2) the plain order does (C3: 34.5 vs 35.4 us; round 3,
  // same-box A/B with MI355_TUNE bit 5, which forces the XCD-contiguous order; profiles/r03_microbench_pitch.txt has the bare patterns)
  const uint32_t T = (R != 2 || (pl.tune & 32)) ? tile_of_block(pl, PROBE_BLOCK(pl), PROBE_GRID(pl)) : PROBE_BLOCK(pl);
  boost_if_late(pl.boost_tiles);
  PROBE_BEGIN(pl)
  uint32_t dg[R][16 / R];
  const uint32_t di = pl.DI[size_t(T) * 512 + t];
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) {
    const uint32_t i1 = 512 * d1 + t;
    load_run<R>(digits, T, i1, dg[d1]);
    if (cbuf_in) apply_carry_in<16 / R>(pl, di, d1, carry_in_of(pl, cbuf_in, T, i1), dg[d1]);
  }
  front_tile<R>(pl, X, T, t, lane, wave, dg, di, sub, Wout);
  PROBE_END(pl)
}

// back sweep: work buffer -> digits + one carry word per run
template <int R>
__global__ void __launch_bounds__(512, 4) k3_cols(DevPlan pl, const uint64_t* __restrict__ Win, uint32_t* __restrict__ digits,
                                                  uint64_t* __restrict__ cbuf, uint32_t a, uint64_t scale) {
  P2* X = reinterpret_cast<P2*>(smem_v2);
  const uint32_t t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), T = tile_of_block(pl, PROBE_BLOCK(pl), PROBE_GRID(pl));
  boost_if_late(pl.boost_tiles);
  PROBE_BEGIN(pl)
  uint32_t dg[R][16 / R];
  uint64_t cout[R], zero[R];
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) zero[d1] = 0;
  back_tile<R>(pl, X, T, t, lane, wave, Win, a, scale, zero, pl.DI[size_t(T) * 512 + t], dg, cout);
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) {
    const uint32_t i1 = 512 * d1 + t;
    store_run<R>(digits, T, i1, dg[d1]);
    cbuf[size_t(T) * (512 * R) + i1] = cout[d1];
  }
  PROBE_END(pl)
}

// back sweep with extras (kernels.hpp BackExt): a second destination register and / or an addend in the carry chain
template <int R>
__global__ void __launch_bounds__(512, 4) k3_cols_ext(DevPlan pl, const uint64_t* __restrict__ Win, uint32_t* __restrict__ digits,
                                                      uint64_t* __restrict__ cbuf, uint32_t a, BackExt ext) {
  P2* X = reinterpret_cast<P2*>(smem_v2);
  const uint32_t t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), T = tile_of_block(pl, blockIdx.x, gridDim.x);
  boost_if_late(pl.boost_tiles);
  uint32_t dg[R][16 / R], ad[R][16 / R];
  uint64_t cout[R], zero[R];
  const uint32_t di = pl.DI[size_t(T) * 512 + t];
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) {
    zero[d1] = 0;
#pragma unroll
    for (int k = 0; k < 16 / R; ++k) ad[d1][k] = 0;
    if (ext.add_digits) {
      const uint32_t i1 = 512 * d1 + t;
      load_run<R>(ext.add_digits, T, i1, ad[d1]);
      if (ext.add_cbuf) apply_carry_in<16 / R>(pl, di, d1, carry_in_of(pl, ext.add_cbuf, T, i1), ad[d1]);
    }
  }
  back_tile<R, true>(pl, X, T, t, lane, wave, Win, a, 1, zero, di, dg, cout, ad);
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) {
    const uint32_t i1 = 512 * d1 + t;
    store_run<R>(digits, T, i1, dg[d1]);
    cbuf[size_t(T) * (512 * R) + i1] = cout[d1];
    if (ext.digits2) { store_run<R>(ext.digits2, T, i1, dg[d1]); ext.cbuf2[size_t(T) * (512 * R) + i1] = cout[d1]; }
  }
}

#if defined(MI355_EXPERIMENTAL)
// ---------------------------------------------------------------------------------------------
// Back sweep of squaring i and front sweep of squaring i + 1 in ONE launch (round 4, second session; Engine::square_mul_n with MI355_CHAIN=1,
// experimental build only).
// Between the two sweeps everything is local to a column tile -- the thread that ends the back sweep with the digits of its R runs is the
// thread that starts the front sweep with them -- except the carry word of the PREVIOUS run in digit order, which the neighbouring tile
// T - 1 produces (reference: carry_weight_mul_p1 / p2 hand it through a carry array and a second kernel, kernels/marin.cl:1696-1728,2198-2216).
// Hand-over inside the launch, one 64-bit word per run: {tag of the launch (12 bits) | carry (52 bits)}, stored and polled with agent-scope
// accesses (past the XCD's non-coherent L2) by the two threads concerned: the word is its own flag, nothing else has to be ordered.
// No cycle and no wait on a group that is not yet dispatched:
//   * tiles are taken in the XCD-contiguous order of the back sweep (block b -> XCD b mod 8, tile (b & 7) NT / 8 + (b >> 3)), so the tile a
//     group waits for belongs to the block dispatched just before it on the SAME XCD;
//   * the first block of every XCD (b < 8), whose predecessor tile is the last one of another XCD's range (or, for tile 0, the last tile of
//     all: 2^p = 1), only runs the back sweep and publishes its carries; eight extra blocks at the end of the grid (b = NT + x, again on
//     XCD x) run that tile in full: by then the tile they wait for is resident or done.  Cost: eight half tiles in 1024.
// The digits themselves are not stored inside a run of squarings (the last squaring of a run goes through the plain back sweep).
// MEASURED (profiles/r04_ab_chain_backfront.txt, same box): C3 0.1426 -> 0.1445 ms (+1.4 %), n = 2^22 +4 %, 2^21 +5 %, 2^24 +13 %: a group
// now lives twice as long, so the end of the launch (CUs left with one late group) costs twice as much, and the waits come on top; the kernel
// boundary it replaces costs less.  Not in the product library: built only with -DMI355_EXPERIMENTAL (make exp), parity-checked there
// (tools/exp_coop_check.py; 6 GPU cases incl. C3 passed on the product build before it was moved).
// A wait that outlasts kChainTimeoutTicks (100 MHz: 50 ms) raises the error word -- the engine refuses every later read-out -- and goes on
// with a zero carry, so that the grid always drains.
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kChainTimeoutTicks = 5000000ull;
constexpr uint32_t kChainTagShift = 52;

template <int R>
__global__ void __launch_bounds__(512, 4) k31_cols(DevPlan pl, uint64_t* __restrict__ Wbuf, uint32_t sub, uint64_t* __restrict__ xbuf, uint32_t* __restrict__ err, uint32_t tag) {
  using S = ColShape<R>;
  P2* X = reinterpret_cast<P2*>(smem_v2);
  const uint32_t t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint32_t NT = pl.M2 / pl.C, per = NT >> 3, b = blockIdx.x;
  const bool extra = b >= NT;                                  // second visit of the first tile of an XCD's range: the full job
  const bool publish_only = b < 8;                             // first visit: back sweep + publish
  const uint32_t T = extra ? (b - NT) * per : (b & 7) * per + (b >> 3);
  if (b >= pl.boost_tiles) __builtin_amdgcn_s_setprio(3);
  uint32_t dg[R][16 / R];
  uint64_t cout[R], zero[R];
  const uint32_t di = pl.DI[size_t(T) * 512 + t];
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) zero[d1] = 0;
  back_tile<R>(pl, X, T, t, lane, wave, Wbuf, 1u, 1, zero, di, dg, cout);
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1)
    __hip_atomic_store(&xbuf[size_t(T) * S::M1 + 512 * d1 + t], (uint64_t(tag) << kChainTagShift) | cout[d1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (publish_only) return;
  // the carries of the previous runs in digit order (carry_in_of): same row of tile T - 1; tile 0 wraps to the last tile of the previous row
  {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t cin[R];
    bool ok = true;
#pragma unroll
    for (int d1 = 0; d1 < R; ++d1) {
      const uint32_t i1 = 512 * d1 + t;
      const uint64_t* src = &xbuf[T ? size_t(T - 1) * S::M1 + i1 : size_t(NT - 1) * S::M1 + (i1 ? i1 - 1 : S::M1 - 1)];
      uint64_t v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t spins = 0;
      while (uint32_t(v >> kChainTagShift) != tag) {
        if ((++spins & 15u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > kChainTimeoutTicks) { ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
        v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      cin[d1] = ok ? (v & ((uint64_t(1) << kChainTagShift) - 1)) : 0;
    }
    if (!ok) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int d1 = 0; d1 < R; ++d1) apply_carry_in<16 / R>(pl, di, d1, cin[d1], dg[d1]);
  }
  asm volatile("" ::: "memory");   // the front sweep's table loads stay behind the wait (hoisted into the back sweep they cost registers: spills)
  front_tile<R>(pl, X, T, t, lane, wave, dg, di, sub, Wbuf);
}

#endif   // MI355_EXPERIMENTAL

}  // namespace v2

namespace v2 {
// chain starts and ratios of the four-step twiddle chains of the column kernels (same thread map as the last
// stage of front_tile / first stage of back_tile): F0f[T][t] = omega_m^(i2 kb) TB[2 i2], F0i the inverse with
// TBi, FBf[i2] = omega_m^(64R i2), FBi its inverse.
template <int R>
__global__ void __launch_bounds__(512) k_build_f0(DevPlan pl, uint64_t* __restrict__ f0f, uint64_t* __restrict__ f0i,
                                                  uint64_t* __restrict__ fbf, uint64_t* __restrict__ fbi) {
  using S = ColShape<R>;
  constexpr int C = S::C, LC = S::LC;
  const uint32_t t = threadIdx.x, T = blockIdx.x;
  const uint32_t c = t & (C - 1), kb = ((t >> (3 + LC)) & (R - 1)) + R * ((t >> LC) & 7) + 8 * R * (t >> 6), i2 = C * T + c;
  const uint32_t ea = i2 * kb;
  f0f[size_t(T) * 512 + t] = gf::mul(tw_lookup(pl, ea), pl.TB[2 * i2]);
  f0i[size_t(T) * 512 + t] = gf::mul(tw_lookup(pl, ea ? pl.m - ea : 0), pl.TBi[2 * i2]);
  if (t < C) {
    const uint32_t eb = i2 * (64 * R);
    fbf[i2] = tw_lookup(pl, eb);
    fbi[i2] = tw_lookup(pl, eb ? pl.m - eb : 0);
  }
}
}  // namespace v2

size_t v2_threads_per_tile(const DevPlan& pl) { return v5_cols_shape(pl) ? v5_threads_per_tile() : v3_cols_shape(pl) ? v3_threads_per_tile() : 512; }

hipError_t v2_build_fourstep(const DevPlan& pl, uint64_t* f0f, uint64_t* f0i, uint64_t* fbf, uint64_t* fbi, hipStream_t s) {
  const dim3 grid(pl.M2 / pl.C), block(512);
  if (v5_cols_shape(pl)) return v5_build_fourstep(pl, f0f, f0i, fbf, fbi, s);
  if (v3_cols_shape(pl)) return v3_build_fourstep(pl, f0f, f0i, fbf, fbi, s);
  switch (pl.M1) {
    case 512: hipLaunchKernelGGL(v2::k_build_f0<1>, grid, block, 0, s, pl, f0f, f0i, fbf, fbi); break;
    case 1024: hipLaunchKernelGGL(v2::k_build_f0<2>, grid, block, 0, s, pl, f0f, f0i, fbf, fbi); break;
    default: hipLaunchKernelGGL(v2::k_build_f0<4>, grid, block, 0, s, pl, f0f, f0i, fbf, fbi); break;
  }
  return hipGetLastError();
}

// ------------------------------- launch wrappers ---------------------------------------------

// rows of 2048 go two to a tile: only where that still gives at least one work-group per CU (n = 2^21, 5 2^20); below that the generic
// rows win.  MI355_TUNE bit 6 switches them off (A/B runs)
// Below 512 rows (n = 2^20) and at 1280 rows (n = 5 2^20: 640 two-row tiles are 1.25 rounds of the chip) a row is a tile of its own with
// one plane per thread (k2_rows2048_planes): same-box A/B 0.0428 -> 0.0378 ms at n = 2^20, 0.1182 -> 0.1137 at 5 2^20, but 0.0571 -> 0.0586
// at n = 2^21 (512 rows: two to a tile stay), profiles/r04_ab_rows2048_planes.txt.  MI355_TUNE bit 13 switches the plane form off, bit 14
// forces it for every row count (A/B runs)
static bool rows2048_planes(const DevPlan& pl) {
  return pl.M2 == 2048 && (((pl.M1 < 512 || pl.M1 == 1280) && !(pl.tune & 8192)) || (pl.tune & 16384));
}
bool v2_rows_supported(const DevPlan& pl) {
  if (v3_rows_shape(pl)) return true;   // rows of 1024: the radix-4 set (kernels_v3.hip)
  if (pl.S2r == nullptr) return false;
  if (pl.M2 == 4096 || pl.M2 == 8192) return true;
  if (rows2048_planes(pl)) return true;
  return pl.M2 == 2048 && pl.M1 % 2 == 0 && pl.M1 >= 512 && !(pl.tune & 64);
}
// columns: M1 = 512 R, R in {1, 2, 4}, with C = 8 / R pairs per run (one 4096-pair tile per work-group)
bool v2_cols_supported(const DevPlan& pl) {
  if (v5_cols_shape(pl)) return pl.DI != nullptr;
  if (v3_cols_shape(pl)) return true;   // columns of 256 x 4: the radix-4 set
  return pl.r5 == 1 && (pl.M1 == 512 || pl.M1 == 1024 || pl.M1 == 2048) && pl.M1 * pl.C == 4096 && pl.M2 >= pl.C * 2 && pl.S1r != nullptr &&
         pl.DI != nullptr;
}

#define MI355_SET_LDS(KERNEL, BYTES)                                                                                          \
  { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, int(BYTES)); \
    if (e_ != hipSuccess) return e_; }
hipError_t v2_configure() {
  MI355_SET_LDS((v2::k2_rows4096<0, 1>), v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<1, 1>), v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<2, 1>), v2::kLdsBytes)
  MI355_SET_LDS((v2::k2_rows4096<0, 2>), 2 * v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<1, 2>), 2 * v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<2, 2>), 2 * v2::kLdsBytes)
  MI355_SET_LDS((v2::k2_rows4096<0, 1, 1>), v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<1, 1, 1>), v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<2, 1, 1>), v2::kLdsBytes)
  MI355_SET_LDS(v2::k1_cols<1>, v2::kLdsBytes) MI355_SET_LDS(v2::k1_cols<2>, v2::kLdsBytes) MI355_SET_LDS(v2::k1_cols<4>, v2::kLdsBytes)
  MI355_SET_LDS(v2::k3_cols<1>, v2::kLdsBytes) MI355_SET_LDS(v2::k3_cols<2>, v2::kLdsBytes) MI355_SET_LDS(v2::k3_cols<4>, v2::kLdsBytes)
  { hipError_t e5 = v5_configure(); if (e5 != hipSuccess) return e5; }
  MI355_SET_LDS(v2::k3_cols_ext<1>, v2::kLdsBytes) MI355_SET_LDS(v2::k3_cols_ext<2>, v2::kLdsBytes) MI355_SET_LDS(v2::k3_cols_ext<4>, v2::kLdsBytes)
#if defined(MI355_EXPERIMENTAL)
  MI355_SET_LDS(v2::k31_cols<1>, v2::kLdsBytes) MI355_SET_LDS(v2::k31_cols<2>, v2::kLdsBytes) MI355_SET_LDS(v2::k31_cols<4>, v2::kLdsBytes)
#endif
  return hipSuccess;
}
#undef MI355_SET_LDS
hipError_t v2_launch_middle(const DevPlan& pl, const uint64_t* Win, const uint64_t* Y, uint64_t* Wout, int mode, uint32_t sub, hipStream_t s) {
  if (v3_rows_shape(pl)) return v3_launch_middle(pl, Win, Y, Wout, mode, sub, s);
#define MI355_ROWS(MODE, HH) hipLaunchKernelGGL((v2::k2_rows4096<MODE, HH>), dim3(pl.M1), dim3(512 * HH), HH * v2::kLdsBytes, s, pl, Win, Y, Wout, sub)
  if (rows2048_planes(pl)) {
#define MI355_ROWS_PL(MODE) hipLaunchKernelGGL((v2::k2_rows2048_planes<MODE>), dim3(pl.M1), dim3(512), v2::kLdsBytesPlanes, s, pl, Win, Y, Wout, sub)
    switch (mode) { case 0: MI355_ROWS_PL(0); break; case 1: MI355_ROWS_PL(1); break; default: MI355_ROWS_PL(2); break; }
#undef MI355_ROWS_PL
  } else if (pl.M2 == 2048) {   // two rows to a tile
#define MI355_ROWS_TWO(MODE) hipLaunchKernelGGL((v2::k2_rows4096<MODE, 1, 1>), dim3(pl.M1 / 2), dim3(512), v2::kLdsBytes, s, pl, Win, Y, Wout, sub)
    switch (mode) { case 0: MI355_ROWS_TWO(0); break; case 1: MI355_ROWS_TWO(1); break; default: MI355_ROWS_TWO(2); break; }
#undef MI355_ROWS_TWO
  } else if (pl.M2 == 4096) {   // one instantiation per mode: the squaring kernel carries no multiply / image code
    switch (mode) { case 0: MI355_ROWS(0, 1); break; case 1: MI355_ROWS(1, 1); break; default: MI355_ROWS(2, 1); break; }
  } else {
    switch (mode) { case 0: MI355_ROWS(0, 2); break; case 1: MI355_ROWS(1, 2); break; default: MI355_ROWS(2, 2); break; }
  }
#undef MI355_ROWS
  return hipGetLastError();
}
hipError_t v2_launch_front(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint32_t sub, uint64_t* W, hipStream_t s) {
  const dim3 grid(pl.M2 / pl.C), block(512);
  if (v5_cols_shape(pl)) return v5_launch_front(pl, digits, cbuf_in, sub, W, s);
  if (v3_cols_shape(pl)) return v3_launch_front(pl, digits, cbuf_in, sub, W, s);
  switch (pl.M1) {
    case 512: hipLaunchKernelGGL(v2::k1_cols<1>, grid, block, v2::kLdsBytes, s, pl, digits, cbuf_in, sub, W); break;
    case 1024: hipLaunchKernelGGL(v2::k1_cols<2>, grid, block, v2::kLdsBytes, s, pl, digits, cbuf_in, sub, W); break;
    default: hipLaunchKernelGGL(v2::k1_cols<4>, grid, block, v2::kLdsBytes, s, pl, digits, cbuf_in, sub, W); break;
  }
  return hipGetLastError();
}
hipError_t v2_launch_back(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, uint64_t scale, hipStream_t s) {
  const dim3 grid(pl.M2 / pl.C), block(512);
  if (v5_cols_shape(pl)) return v5_launch_back(pl, W, digits, cbuf, a, scale, s);
  if (v3_cols_shape(pl)) return v3_launch_back(pl, W, digits, cbuf, a, scale, s);
  switch (pl.M1) {
    case 512: hipLaunchKernelGGL(v2::k3_cols<1>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, scale); break;
    case 1024: hipLaunchKernelGGL(v2::k3_cols<2>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, scale); break;
    default: hipLaunchKernelGGL(v2::k3_cols<4>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, scale); break;
  }
  return hipGetLastError();
}
hipError_t v2_launch_back_ext(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& x, hipStream_t s) {
  const dim3 grid(pl.M2 / pl.C), block(512);
  if (v5_cols_shape(pl)) return v5_launch_back_ext(pl, W, digits, cbuf, a, x, s);
  if (v3_cols_shape(pl)) return v3_launch_back_ext(pl, W, digits, cbuf, a, x, s);
  switch (pl.M1) {
    case 512: hipLaunchKernelGGL(v2::k3_cols_ext<1>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, x); break;
    case 1024: hipLaunchKernelGGL(v2::k3_cols_ext<2>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, x); break;
    default: hipLaunchKernelGGL(v2::k3_cols_ext<4>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, x); break;
  }
  return hipGetLastError();
}

#if defined(MI355_EXPERIMENTAL)
// back + front in one launch (k31_cols): served where the power-of-two column kernels run on at least 16 tiles, a multiple of 8, and a run
// carry fits the 52 bits of a hand-over word (a = 1: carry < 2^(64 - q))
bool v2_chain_supported(const DevPlan& pl) {
  const uint32_t NT = pl.M2 / pl.C;
  return v2_cols_supported(pl) && !v5_cols_shape(pl) && !v3_cols_shape(pl) && NT % 8 == 0 && NT >= 16 && pl.q >= 13;
}
hipError_t v2_launch_backfront(const DevPlan& pl, uint64_t* W, uint32_t sub, uint64_t* xbuf, uint32_t* err, uint32_t tag, hipStream_t s) {
  const dim3 grid(pl.M2 / pl.C + 8), block(512);
  switch (pl.M1) {
    case 512: hipLaunchKernelGGL(v2::k31_cols<1>, grid, block, v2::kLdsBytes, s, pl, W, sub, xbuf, err, tag); break;
    case 1024: hipLaunchKernelGGL(v2::k31_cols<2>, grid, block, v2::kLdsBytes, s, pl, W, sub, xbuf, err, tag); break;
    default: hipLaunchKernelGGL(v2::k31_cols<4>, grid, block, v2::kLdsBytes, s, pl, W, sub, xbuf, err, tag); break;
  }
  return hipGetLastError();
}
#endif

#if defined(MI355_PROBE)
size_t v2_lds_bytes() { return v2::kLdsBytes; }
hipError_t v2_probe_launch(const DevPlan& pl, int kind, int grid_mult, int extra_lds, const uint32_t* digits, uint64_t* cbuf, uint64_t* W, uint32_t* dout, hipStream_t s) {
  const size_t lds = v2::kLdsBytes + size_t(extra_lds);
  if (kind == 1) {
    if (pl.M2 != 4096) return hipErrorNotSupported;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(v2::k2_rows4096<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((v2::k2_rows4096<0, 1>), dim3(pl.M1 * grid_mult), dim3(512), lds, s, pl, W, nullptr, W, 0u);
    return hipGetLastError();
  }
  if (v5_cols_shape(pl)) return v5_probe_launch(pl, kind, grid_mult, extra_lds, digits, cbuf, W, dout, s);
  if (pl.M1 != 1024) return hipErrorNotSupported;
  const dim3 grid((pl.M2 / pl.C) * grid_mult), block(512);
  if (kind == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(v2::k1_cols<2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(v2::k1_cols<2>, grid, block, lds, s, pl, digits, cbuf, 0u, W);
  } else {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(v2::k3_cols<2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(v2::k3_cols<2>, grid, block, lds, s, pl, W, dout, cbuf, 1u, uint64_t(1));
  }
  return hipGetLastError();
}
#endif

}  // namespace mi355
