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
// Shapes served: rows M2 = 4096 (8.8.8.8) and 8192 (2 x 4096 under one radix-2 level); columns
// M1 = 512 R (R.8.8.8, R = 1, 2, 4) with C = 8/R pairs per run.  Everything else runs on the generic set.
// Row order of the work buffer and digit layout are those of kernels.hip, so the two sets interoperate
// kernel by kernel (the multiplicand image layout differs: an engine uses one middle kernel for both
// set_multiplicand and mul).
// Value ranges: field values are canonical ([0, P)) except that (a) a negated zero may be P (gf.hpp,
// mul_pow2) and (b) sums marked LAZY (gfdft.hpp) may be any 64-bit representative; (b) only ever feeds a
// multiplication, and every kernel's last arithmetic step before a store is a multiplication or a canonical
// add/sub, so nothing non-canonical other than P reaches an add/sub operand or a digit.
//
// LDS exchanges (P2 = 16 B slots, index skewed by i + i/8 against bank conflicts):
//   writer "thread-major": slot t*8 + r          reader: slot j*512 + t'
//   writer "wave-major":   slot w*512 + r*64 + f(lane)   reader: slot j*512 + t'
// which is the digit permutation that hands each thread the 8 elements of its next radix-8.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gfdft.hpp"
#include "kernels.hpp"

namespace mi355 {
namespace v2 {

struct alignas(16) P2 { uint64_t a, b; };

__device__ __forceinline__ uint32_t phys(uint32_t i) { return i + (i >> 3); }
constexpr uint32_t kLdsSlots = 4096 + 512;
constexpr uint32_t kLdsBytes = kLdsSlots * 16;

__device__ __forceinline__ P2 p2_mul(P2 x, uint64_t w) { return {gf::mul(x.a, w), gf::mul(x.b, w)}; }

template <bool INV, int LAZY = 0>
__device__ __forceinline__ void dft8p(P2 (&x)[8]) {
  uint64_t u[8], v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { u[j] = x[j].a; v[j] = x[j].b; }
  gf::dft8<INV, LAZY>(u);
  gf::dft8<INV, LAZY>(v);
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = {u[j], v[j]};
}

// Seams inside a 64-point block: x[k] *= omega_64^(k w) = 2^(39 k w) with w uniform over the wavefront.
// The wave index is made a template parameter: every shift amount is then a compile-time constant and
// mul_pow2 collapses to its 8-11 instruction cases (a general multiply by a scalar-built power of two,
// the previous form, is 24).  The caller switches on the scalar wave index once per seam and each wave
// runs only its own copy (code grows by ~1 KB per copy; measured -5 % on the row kernel, -4 % on the
// column kernels).
template <int W, bool INV>
__device__ __forceinline__ void seam64_const(P2 (&x)[8]) {
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const unsigned f = (gf::LOG2_W64 * unsigned(k) * unsigned(W)) % 192u;
    const unsigned s = INV ? (192u - f) % 192u : f;
    x[k] = {gf::mul_pow2(x[k].a, s), gf::mul_pow2(x[k].b, s)};
  }
}
// FOLD0: the caller left x[1..3] un-folded (dft8 LAZY = 1) because the shifts below accept any operand; wave 0
// shifts by nothing, so it folds them here instead.
template <bool INV, bool FOLD0 = false>
__device__ __forceinline__ void seam64(P2 (&x)[8], uint32_t wave) {
  switch (wave) {
    case 0:
      if (FOLD0) {
#pragma unroll
        for (int k = 1; k < 4; ++k) x[k] = {gf::fold(x[k].a), gf::fold(x[k].b)};
      }
      break;
    case 1: seam64_const<1, INV>(x); break;
    case 2: seam64_const<2, INV>(x); break;
    case 3: seam64_const<3, INV>(x); break;
    case 4: seam64_const<4, INV>(x); break;
    case 5: seam64_const<5, INV>(x); break;
    case 6: seam64_const<6, INV>(x); break;
    default: seam64_const<7, INV>(x); break;
  }
}

#define MI355_SWITCH8(w, CALL) \
  switch (w) {                 \
    case 0: CALL(0); break;    \
    case 1: CALL(1); break;    \
    case 2: CALL(2); break;    \
    case 3: CALL(3); break;    \
    case 4: CALL(4); break;    \
    case 5: CALL(5); break;    \
    case 6: CALL(6); break;    \
    default: CALL(7); break;   \
  }

// omega_m^e from the two-level table (e < m)
__device__ __forceinline__ uint64_t tw_lookup(const DevPlan& pl, uint64_t e) {
  const uint64_t lo = pl.TWlo[e & ((1u << pl.twh) - 1)], hi = pl.TWhi[e >> pl.twh];
  return gf::mul(lo, hi);
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_v2[];

// block -> tile for the back sweep: blocks that share an XCD (b, b+8, ... under round-robin dispatch)
// get neighbouring tiles, so the two 64-byte halves of a 128-byte work-buffer line meet in one L2
// (speed only; measured 73 -> 66 us at C3).  MI355_TUNE bit 0 switches it off.
__device__ __forceinline__ uint32_t tile_of_block(const DevPlan& pl, uint32_t b, uint32_t nblocks) {
  if (!(pl.tune & 1) && (nblocks % 8 == 0)) return (b & 7) * (nblocks >> 3) + (b >> 3);
  return b;
}
// Issue-priority boost for the work-groups of the last half round.  Two work-groups share a CU and the older
// one wins the issue arbitration, so at the end of a launch every CU is left with one late-started group
// running alone at ~3/4 of the pair rate.  Raising the priority of exactly those late
// groups (block index >= boost_from, set by the engine from the grid size and the CU count) lets them overtake
// their older neighbour, and the two finish closer together: -3 % per squaring at C3 (same-box A/B).
__device__ __forceinline__ void boost_if_late(uint32_t boost_from) {
  if (blockIdx.x >= boost_from) __builtin_amdgcn_s_setprio(3);
}

// Barrier for LDS exchanges: waits for this wave's LDS operations only.  __syncthreads() also drains the vector-memory
// counter (it carries a work-group fence), which makes every wave wait at the barrier for table words that were
// requested on purpose before the exchange; global data is not exchanged between waves inside these kernels.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#if defined(MI355_PROBE)
// timeline probe: thread 0 of every work-group records {realtime0, realtime1, shader clock0, clock1, HW_ID, XCC_ID, mid realtime}
#define PROBE_BEGIN(pl)                                                                             \
  uint64_t pb_r0_ = 0, pb_c0_ = 0, pb_rm_ = 0;                                                      \
  if (pl.probe) { pb_r0_ = __builtin_amdgcn_s_memrealtime(); pb_c0_ = __builtin_amdgcn_s_memtime(); }
#define PROBE_MID(pl) if (pl.probe) pb_rm_ = __builtin_amdgcn_s_memrealtime();
#define PROBE_END(pl)                                                                               \
  if (pl.probe && threadIdx.x == 0) {                                                               \
    uint64_t* o_ = pl.probe + size_t(blockIdx.x) * 8;                                               \
    o_[0] = pb_r0_; o_[1] = __builtin_amdgcn_s_memrealtime(); o_[2] = pb_c0_; o_[3] = __builtin_amdgcn_s_memtime(); \
    o_[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o_[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20); o_[6] = pb_rm_; \
  }
#define PROBE_BLOCK(pl) (pl.probe_mod ? blockIdx.x % pl.probe_mod : blockIdx.x)
#define PROBE_GRID(pl) (pl.probe_mod ? pl.probe_mod : gridDim.x)
#else
#define PROBE_BEGIN(pl)
#define PROBE_MID(pl)
#define PROBE_END(pl)
#define PROBE_BLOCK(pl) blockIdx.x
#define PROBE_GRID(pl) gridDim.x
#endif

// exchange helpers: barrier, write 8, barrier, read 8
#define EXCH_THREAD_MAJOR_TO_STRIDED(X, x, t)                      \
  lds_barrier();                                                   \
  _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) X[phys((t) * 8 + r_)] = x[r_]; \
  lds_barrier();                                                   \
  _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) x[j_] = X[phys(j_ * 512 + (t))];

#define EXCH_STRIDED_TO_THREAD_MAJOR(X, x, t)                      \
  lds_barrier();                                                   \
  _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) X[phys(j_ * 512 + (t))] = x[j_]; \
  lds_barrier();                                                   \
  _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) x[r_] = X[phys((t) * 8 + r_)];

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
template <int mode, int H>
__global__ void __launch_bounds__(512 * H, 4) k2_rows4096(DevPlan pl, const uint64_t* __restrict__ Win, const uint64_t* __restrict__ Yimg,
                                                          uint64_t* __restrict__ Wout, uint32_t sub) {
  const uint32_t h = (H == 2) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 9) : 0u;
  if (H == 1) boost_if_late(pl.boost_rows);
  P2* X = reinterpret_cast<P2*>(smem_v2) + h * kLdsSlots;
  const uint32_t t = threadIdx.x & 511, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 7), row = PROBE_BLOCK(pl);
  PROBE_BEGIN(pl)
  const P2* in = reinterpret_cast<const P2*>(Win) + size_t(row) * (4096 * H);
  P2* out = reinterpret_cast<P2*>(Wout) + size_t(row) * (4096 * H) + h * 4096;
  P2 x[8];
  // table words of the pointwise stage, requested first: their latency hides behind the forward transform
  const uint32_t kb = (lane >> 3) + 8 * (lane & 7) + 64 * wave;   // S4 thread (k3|k1|k2): frequency base k1 + 8 k2 + 64 k3
  const uint32_t blk = row / pl.L1, qq = row - blk * pl.L1;       // column-DFT slot -> frequency (kernels.hip freq1)
  const uint32_t k1row = blk + pl.r5 * (pl.logL1 ? (__brev(qq) >> (32 - pl.logL1)) : 0u);
  const uint64_t erho = uint64_t(k1row) + uint64_t(pl.M1) * (H * kb + h);   // row frequency of X[kb]: H kb + h
  const uint64_t rho_lo = pl.TWlo[erho & ((1u << pl.twh) - 1)], rho_hi = pl.TWhi[erho >> pl.twh];

  // ---- forward ----
  // deferred small subtraction (LL's -2) on a front image: digit 0 has weight 1 and reaches column 0,
  // plane a of every row unchanged
  if (H == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = in[512 * j + t];
    if (sub != 0 && t == 0) x[0].a = gf::sub(x[0].a, uint64_t(sub));
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
  dft8p<false, 1>(x);   // outputs 1..7 are shifted next, output 0 is not
  seam64<false, true>(x, wave);
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
    const P2* Y = reinterpret_cast<const P2*>(Yimg) + size_t(row) * (4096 * H) + h * 4096;
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
  seam64<true>(x, wave);
  if (H == 1) dft8p<true, 2>(x);   // stored for a back sweep, which multiplies by its twiddle first
  else dft8p<true>(x);
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
  for (int j = 0; j < 8; ++j) out[512 * j + t] = x[j];
  PROBE_END(pl)
}

// previous run (in digit order) of run (T, i1); see kernels.hip
__device__ __forceinline__ uint64_t carry_in_of(const DevPlan& pl, const uint64_t* cbuf, uint32_t T, uint32_t i1) {
  const uint32_t NT = pl.M2 / pl.C;
  if (T > 0) return cbuf[size_t(T - 1) * pl.M1 + i1];
  return cbuf[size_t(NT - 1) * pl.M1 + (i1 ? i1 - 1 : pl.M1 - 1)];
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

// out[k] = sum_j in[j] omega_4^(jk), omega_4 = 2^48 (forward; the inverse uses omega_4^-1 = -2^48), in place
template <bool INV>
__device__ __forceinline__ void dft4(uint64_t& x0, uint64_t& x1, uint64_t& x2, uint64_t& x3) {
  const uint64_t a0 = gf::add(x0, x2), a1 = gf::add(x1, x3), b0 = gf::sub(x0, x2);
  const uint64_t b1 = gf::mul_pow2(INV ? gf::sub(x3, x1) : gf::sub(x1, x3), 48);
  x0 = gf::add(a0, a1); x2 = gf::sub(a0, a1); x1 = gf::add(b0, b1); x3 = gf::sub(b0, b1);
}

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

// weak carry of a run's incoming carry word into its first digits (adc4, marin.cl:203-212)
// di: the thread's word of the digit-info table (2 bits per digit: width - q, wrap), run = index of the run
// among the thread's R runs
template <int ND>
__device__ __forceinline__ void apply_carry_in(const DevPlan& pl, uint32_t di, int run, uint64_t cin, uint32_t (&d)[ND]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t width = pl.q + ((di >> (2 * (run * ND + k))) & 1u);
    const uint64_t v = uint64_t(d[k]) + cin;
    d[k] = __builtin_amdgcn_ubfe(uint32_t(v), 0u, width);
    cin = v >> width;
  }
  d[3] += uint32_t(cin);
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
    const uint64_t tah = gf::half(pl.TA[i1]), tah1 = gf::half(pl.TA[S::M1 + i1]);
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
      W[size_t(row0 + rj) * pl.M2 + i2] = {gf::mul(x[j].a, ca), gf::mul(x[j].b, ca)};
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
      x[j] = W[size_t(row0 + rj) * pl.M2 + i2];
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
  uint64_t btai_e[R], btai_o[R];
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) { btai_e[d1] = pl.TAi[512 * d1 + t]; btai_o[d1] = pl.TAi[S::M1 + 512 * d1 + t]; }
  EXCH_STRIDED_TO_THREAD_MAJOR(X, x, t)
#define MI355_CALL(W) stage_r_inv_const<R, W>(x)
  MI355_SWITCH8(wave, MI355_CALL)
#undef MI355_CALL
#pragma unroll
  for (int d1 = 0; d1 < R; ++d1) {
    const uint64_t tai_e = btai_e[d1], tai_o = btai_o[d1];
    const uint64_t tai2_e = gf::dbl(tai_e), tai2_o = gf::dbl(tai_o);
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
  // them on one XCD lets its L2 merge the 32-byte pieces (n = 2^24: front sweep 97 -> 89 us); with wider runs the
  // plain order is as good or better (C3: 43.2 vs 44.9 us)
  const uint32_t T = (R == 4) ? tile_of_block(pl, PROBE_BLOCK(pl), PROBE_GRID(pl)) : PROBE_BLOCK(pl);
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

}  // namespace v2

// ---------------------------------------------------------------------------------------------
// Columns of M1 = 1280 = 5 x 256 (n = 5 * 2^21: BASELINE configs[3] on the Goldilocks path), C = 4 pairs per run: one tile
// of 5120 pairs per work-group of 640 threads, 8 pairs per thread, the radix-5 stage on 512 threads with 10.
// The reference serves this size with forward80_0 / backward80_0 (kernels/marin.cl:1019-1040, engine_gpu.h:1619).
//   i1 = 256 d0 + r,  r = 64 e1 + 8 e2 + e3;  column frequency k = k0 + 5 kr,  kr = k1 + 4 k2 + 32 k3
//   L   thread t: runs i1 = t, t + 640 (digits -> carry-in -> weight)
//   A   thread u < 512: groups g = u, u + 512 (r = g / 4, c = g % 4): DFT5 over d0, twiddle omega_1280^(r k0)
//   B1  thread (k0 | e2 | e3 | c/2): DFT4 over e1, twiddle omega_256^(k1 (8 e2 + e3))
//   B2  thread (k0 | k1 | e3 | c):   DFT8 over e2, twiddle omega_64^(k2 e3)
//   B3  thread (k0 | k1 | k2 | c):   DFT8 over e3, then the four-step twiddle chain (ratio omega_m^(160 i2)) and the store to
//       row k0 256 + bitrev8(kr), the row order of kernels.hip freq1
// The factor 5 leaves no digit that is uniform over a wavefront, so the seams inside the power-of-two part are table
// multiplications (omega_1280 powers from UT1) instead of the compile-time shifts of the 512 R shapes.  LDS carries one
// plane (8 bytes per slot) at a time: 46 KiB per work-group, so that two of them share a CU.
// The back sweep is the mirror image, ending with the unweighting and the carry along the thread's two runs.
// ---------------------------------------------------------------------------------------------
namespace v5 {
constexpr uint32_t kThreads = 640, kTile = 5120, kM1 = 1280;
// Launched with 768 threads: twelve waves spread evenly over the four SIMDs of a CU, the last two leave at once.  A work-group of ten waves
// (3 + 3 + 2 + 2) is not placed next to a resident one for up to 17 us after a slot has become free (profiles/r03_probe_c4.md: 44 % of
// the dispatches of a launch, a CU then runs one group for half of its time); twelve are placed within 2 us like the 512-thread groups.
constexpr uint32_t kLaunchThreads = 768;
constexpr uint32_t kPlaneWords = kTile + kTile / 32 + 8;
// the exchange plane, then a copy of the omega_1280 table (10 KiB): the seams of this shape are table multiplications and their roots
// come out of LDS (~100 cycles) instead of L2 (several hundred, exposed at every stage)
constexpr uint32_t kLdsBytes = (kPlaneWords + kM1) * 8;
__device__ __forceinline__ uint32_t ph(uint32_t i) { return i + (i >> 5); }
__device__ __forceinline__ const uint64_t* stage_roots(const DevPlan& pl, uint64_t* X) {
  uint64_t* R = X + kPlaneWords;
  for (uint32_t i = threadIdx.x; i < kM1; i += kThreads) R[i] = pl.UT1[i];
  return R;   // visible after the first barrier of the first exchange
}

// write 8 (or 10) values to element ids wi[], read ids ri[]; one plane after the other
template <int NW, int NR, class WI, class RI>
__device__ __forceinline__ void exchange(uint64_t* X, const v2::P2* in, v2::P2* out, bool writer, bool reader, WI wi, RI ri) {
  v2::lds_barrier();
  if (writer) { _Pragma("unroll") for (int k = 0; k < NW; ++k) X[ph(wi(k))] = in[k].a; }
  v2::lds_barrier();
  uint64_t ta[NR];
  if (reader) { _Pragma("unroll") for (int k = 0; k < NR; ++k) ta[k] = X[ph(ri(k))]; }
  v2::lds_barrier();
  if (writer) { _Pragma("unroll") for (int k = 0; k < NW; ++k) X[ph(wi(k))] = in[k].b; }
  v2::lds_barrier();
  if (reader) { _Pragma("unroll") for (int k = 0; k < NR; ++k) out[k] = {ta[k], X[ph(ri(k))]}; }
}

__device__ __forceinline__ void dft5p(v2::P2 (&x)[5], const uint64_t (&c5)[4], bool inverse) {
  // kernels.hip dft5, on the two planes (Winograd: 4 table multiplications + one shift per plane)
  using v2::P2;
  auto add = [](P2 p, P2 q) { return P2{gf::add(p.a, q.a), gf::add(p.b, q.b)}; };
  auto sub = [](P2 p, P2 q) { return P2{gf::sub(p.a, q.a), gf::sub(p.b, q.b)}; };
  auto mul = [](P2 p, uint64_t w) { return P2{gf::mul(p.a, w), gf::mul(p.b, w)}; };
  const P2 t1 = add(x[1], x[4]), t2 = add(x[2], x[3]), t3 = sub(x[1], x[4]), t4 = sub(x[2], x[3]);
  const P2 t5 = add(t1, t2);
  const P2 A = add(x[0], P2{gf::mul_pow2(t5.a, 94), gf::mul_pow2(t5.b, 94)});
  const P2 m2 = mul(sub(t1, t2), c5[0]);
  const P2 B1 = add(A, m2), B2 = sub(A, m2);
  const P2 m3 = mul(add(t3, t4), c5[1]), m4 = mul(t4, c5[2]), m5 = mul(t3, c5[3]);
  const P2 Pp = add(m3, m4), Q = sub(m5, m3);
  x[0] = add(x[0], t5);
  if (!inverse) { x[1] = add(B1, Pp); x[4] = sub(B1, Pp); x[2] = add(B2, Q); x[3] = sub(B2, Q); }
  else          { x[1] = sub(B1, Pp); x[4] = add(B1, Pp); x[2] = sub(B2, Q); x[3] = add(B2, Q); }
}

__device__ __forceinline__ uint32_t brev8(uint32_t k) { return __brev(k) >> 24; }

__global__ void __launch_bounds__(kLaunchThreads, 2) k1_cols5(DevPlan pl, const uint32_t* __restrict__ digits, const uint64_t* __restrict__ cbuf_in, uint32_t sub,
                                                   uint64_t* __restrict__ Wout) {
  using v2::P2;
  if (threadIdx.x >= kThreads) return;   // the two padding waves (kLaunchThreads): ended waves do not count at a barrier
  uint64_t* X = reinterpret_cast<uint64_t*>(v2::smem_v2);
  const uint32_t t = threadIdx.x, T = PROBE_BLOCK(pl);
  v2::boost_if_late(pl.boost_tiles);
  PROBE_BEGIN(pl)
  const uint64_t* __restrict__ UT = stage_roots(pl, X);
  // ---- L: the thread's two runs ----
  P2 x[8];
  {
    const uint32_t di = pl.DI[size_t(T) * kThreads + t];
    const uint32_t nowrap = ~di;
#pragma unroll
    for (int d1 = 0; d1 < 2; ++d1) {
      const uint32_t i1 = t + kThreads * d1;
      uint32_t dg[8];
      {
        const uint4* src = reinterpret_cast<const uint4*>(digits) + (size_t(T) * kM1 + i1) * 2;
        const uint4 v0 = src[0], v1 = src[1];
        dg[0] = v0.x; dg[1] = v0.y; dg[2] = v0.z; dg[3] = v0.w; dg[4] = v1.x; dg[5] = v1.y; dg[6] = v1.z; dg[7] = v1.w;
      }
      if (cbuf_in) v2::apply_carry_in<8>(pl, di, d1, v2::carry_in_of(pl, cbuf_in, T, i1), dg);
      const uint64_t tah = gf::half(pl.TA[i1]), tah1 = gf::half(pl.TA[kM1 + i1]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int idx = d1 * 8 + 2 * c;
        x[d1 * 4 + c] = {gf::mul_u32(tah, dg[2 * c] << ((nowrap >> (2 * idx + 1)) & 1u)), gf::mul_u32(tah1, dg[2 * c + 1] << ((nowrap >> (2 * idx + 3)) & 1u))};
      }
      if (sub != 0 && T == 0 && i1 == 0) x[0].a = gf::sub(x[0].a, uint64_t(sub));
    }
  }
  // ---- A: DFT5 over d0 ----
  P2 y[10];
  exchange<8, 10>(X, x, y, true, t < 512,
                  [&](int k) { return (t + kThreads * (k >> 2)) * 4 + (k & 3); },
                  [&](int k) { const uint32_t g = t + 512 * (k / 5); return (256 * (k % 5) + (g >> 2)) * 4 + (g & 3); });
  if (t < 512) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = (t + 512 * q) >> 2;
      P2 z[5] = {y[5 * q], y[5 * q + 1], y[5 * q + 2], y[5 * q + 3], y[5 * q + 4]};
      dft5p(z, pl.W5c, false);
#pragma unroll
      for (int k0 = 1; k0 < 5; ++k0) z[k0] = v2::p2_mul(z[k0], UT[r * k0]);
#pragma unroll
      for (int k0 = 0; k0 < 5; ++k0) y[5 * q + k0] = z[k0];
    }
  }
  // ---- B1: DFT4 over e1 ----
  {
    const uint32_t chi = t & 1, e3 = (t >> 1) & 7, e2 = (t >> 4) & 7, k0 = t >> 7;
    exchange<10, 8>(X, y, x, t < 512, true,
                    [&](int k) { const uint32_t g = t + 512 * (k / 5); return ((k % 5) * 256 + (g >> 2)) * 4 + (g & 3); },
                    [&](int k) { return (k0 * 256 + 64 * (k >> 1) + 8 * e2 + e3) * 4 + 2 * chi + (k & 1); });
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      v2::dft4<false>(x[cl].a, x[2 + cl].a, x[4 + cl].a, x[6 + cl].a);
      v2::dft4<false>(x[cl].b, x[2 + cl].b, x[4 + cl].b, x[6 + cl].b);
    }
    const uint32_t rr = 8 * e2 + e3;
#pragma unroll
    for (int k1 = 1; k1 < 4; ++k1) {
      const uint64_t w = UT[5 * k1 * rr];   // omega_256 = omega_1280^5
      x[2 * k1] = v2::p2_mul(x[2 * k1], w); x[2 * k1 + 1] = v2::p2_mul(x[2 * k1 + 1], w);
    }
    // ---- B2: DFT8 over e2 ----
    P2 z[8];
    const uint32_t c = t & 3, f3 = (t >> 2) & 7, f1 = (t >> 5) & 3, f0 = t >> 7;   // reader (k0 | k1 | e3 | c)
    exchange<8, 8>(X, x, z, true, true,
                   [&](int k) { return (((k0 * 4 + (k >> 1)) * 8 + e2) * 8 + e3) * 4 + 2 * chi + (k & 1); },
                   [&](int k) { return (((f0 * 4 + f1) * 8 + k) * 8 + f3) * 4 + c; });
    v2::dft8p<false, 1>(z);
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) z[k2] = v2::p2_mul(z[k2], UT[20 * k2 * f3]);   // omega_64 = omega_1280^20
    z[0] = {gf::fold(z[0].a), gf::fold(z[0].b)};
    // ---- B3: DFT8 over e3 ----
    const uint32_t g2 = (t >> 2) & 7;   // reader (k0 | k1 | k2 | c): same decode, e3's place holds k2
    exchange<8, 8>(X, z, x, true, true,
                   [&](int k) { return (((f0 * 4 + f1) * 8 + k) * 8 + f3) * 4 + c; },
                   [&](int k) { return (((f0 * 4 + f1) * 8 + g2) * 8 + k) * 4 + c; });
    v2::dft8p<false, 2>(x);
    const uint32_t i2 = 4 * T + c;
    uint64_t ca = pl.F0f[size_t(T) * kThreads + t];
    const uint64_t B = pl.FBf[i2];
    P2* W = reinterpret_cast<P2*>(Wout);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t row = f0 * 256 + brev8(f1 + 4 * g2 + 32 * j);
      W[size_t(row) * pl.M2 + i2] = {gf::mul(x[j].a, ca), gf::mul(x[j].b, ca)};
      if (j < 7) ca = gf::mul(ca, B);
    }
  }
  PROBE_END(pl)
}

template <bool EXT>
__global__ void __launch_bounds__(kLaunchThreads, 2) k3_cols5(DevPlan pl, const uint64_t* __restrict__ Win, uint32_t* __restrict__ digits, uint64_t* __restrict__ cbuf,
                                                   uint32_t a, uint64_t scale, BackExt ext) {
  using v2::P2;
  if (threadIdx.x >= kThreads) return;   // the two padding waves (kLaunchThreads)
  uint64_t* X = reinterpret_cast<uint64_t*>(v2::smem_v2);
  const uint32_t t = threadIdx.x, T = v2::tile_of_block(pl, PROBE_BLOCK(pl), PROBE_GRID(pl));
  v2::boost_if_late(pl.boost_tiles);
  PROBE_BEGIN(pl)
  const uint64_t* __restrict__ UT = stage_roots(pl, X);
  const uint32_t c = t & 3, g2 = (t >> 2) & 7, f1 = (t >> 5) & 3, f0 = t >> 7, f3 = g2;
  P2 x[8], z[8];
  {
    const uint32_t i2 = 4 * T + c;
    uint64_t ca = pl.F0i[size_t(T) * kThreads + t];
    const uint64_t B = pl.FBi[i2];
    if (scale != 1) ca = gf::mul(ca, scale);
    const P2* W = reinterpret_cast<const P2*>(Win);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = W[size_t(f0 * 256 + brev8(f1 + 4 * g2 + 32 * j)) * pl.M2 + i2];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = {gf::mul(x[j].a, ca), gf::mul(x[j].b, ca)}; if (j < 7) ca = gf::mul(ca, B); }
  }
  v2::dft8p<true, 2>(x);   // k3 -> e3; multiplied by the seam next
  exchange<8, 8>(X, x, z, true, true,
                 [&](int k) { return (((f0 * 4 + f1) * 8 + g2) * 8 + k) * 4 + c; },
                 [&](int k) { return (((f0 * 4 + f1) * 8 + k) * 8 + f3) * 4 + c; });
#pragma unroll
  for (int k2 = 1; k2 < 8; ++k2) { const uint32_t e = 20 * k2 * f3; z[k2] = v2::p2_mul(z[k2], UT[e ? kM1 - e : 0]); }
  z[0] = {gf::fold(z[0].a), gf::fold(z[0].b)};
  v2::dft8p<true, 2>(z);   // k2 -> e2
  const uint32_t chi = t & 1, e3 = (t >> 1) & 7, e2 = (t >> 4) & 7, k0 = t >> 7;
  exchange<8, 8>(X, z, x, true, true,
                 [&](int k) { return (((f0 * 4 + f1) * 8 + k) * 8 + f3) * 4 + c; },
                 [&](int k) { return (((k0 * 4 + (k >> 1)) * 8 + e2) * 8 + e3) * 4 + 2 * chi + (k & 1); });
  {
    const uint32_t rr = 8 * e2 + e3;
    x[0] = {gf::fold(x[0].a), gf::fold(x[0].b)}; x[1] = {gf::fold(x[1].a), gf::fold(x[1].b)};
#pragma unroll
    for (int k1 = 1; k1 < 4; ++k1) {
      const uint32_t e = 5 * k1 * rr;
      const uint64_t w = UT[e ? kM1 - e : 0];
      x[2 * k1] = v2::p2_mul(x[2 * k1], w); x[2 * k1 + 1] = v2::p2_mul(x[2 * k1 + 1], w);
    }
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      v2::dft4<true>(x[cl].a, x[2 + cl].a, x[4 + cl].a, x[6 + cl].a);
      v2::dft4<true>(x[cl].b, x[2 + cl].b, x[4 + cl].b, x[6 + cl].b);
    }
  }
  P2 y[10];
  exchange<8, 10>(X, x, y, true, t < 512,
                  [&](int k) { return (k0 * 256 + 64 * (k >> 1) + 8 * e2 + e3) * 4 + 2 * chi + (k & 1); },
                  [&](int k) { const uint32_t g = t + 512 * (k / 5); return ((k % 5) * 256 + (g >> 2)) * 4 + (g & 3); });
  if (t < 512) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = (t + 512 * q) >> 2;
      P2 w5[5] = {y[5 * q], y[5 * q + 1], y[5 * q + 2], y[5 * q + 3], y[5 * q + 4]};
#pragma unroll
      for (int k0i = 1; k0i < 5; ++k0i) { const uint32_t e = r * k0i; w5[k0i] = v2::p2_mul(w5[k0i], UT[e ? kM1 - e : 0]); }
      dft5p(w5, pl.W5c, true);
#pragma unroll
      for (int d0 = 0; d0 < 5; ++d0) y[5 * q + d0] = w5[d0];
    }
  }
  exchange<10, 8>(X, y, x, t < 512, true,
                  [&](int k) { const uint32_t g = t + 512 * (k / 5); return (256 * (k % 5) + (g >> 2)) * 4 + (g & 3); },
                  [&](int k) { return (t + kThreads * (k >> 2)) * 4 + (k & 3); });
  // ---- unweight, x a, carry along the thread's two runs ----
  const uint32_t di = pl.DI[size_t(T) * kThreads + t];
#pragma unroll
  for (int d1 = 0; d1 < 2; ++d1) {
    const uint32_t i1 = t + kThreads * d1;
    const uint64_t tai_e = pl.TAi[i1], tai_o = pl.TAi[kM1 + i1];
    const uint64_t tai2_e = gf::dbl(tai_e), tai2_o = gf::dbl(tai_o);
    uint32_t ad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (EXT && ext.add_digits) {
      const uint4* src = reinterpret_cast<const uint4*>(ext.add_digits) + (size_t(T) * kM1 + i1) * 2;
      const uint4 v0 = src[0], v1 = src[1];
      ad[0] = v0.x; ad[1] = v0.y; ad[2] = v0.z; ad[3] = v0.w; ad[4] = v1.x; ad[5] = v1.y; ad[6] = v1.z; ad[7] = v1.w;
      if (ext.add_cbuf) v2::apply_carry_in<8>(pl, di, d1, v2::carry_in_of(pl, ext.add_cbuf, T, i1), ad);
    }
    uint64_t carry = 0;
    uint32_t dg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t bits = di >> (2 * (d1 * 8 + k));
      const uint32_t width = pl.q + (bits & 1u);
      const bool wrap = (bits & 2u) != 0;
      const P2 v = x[4 * d1 + (k >> 1)];
      const uint64_t u = (k & 1) ? gf::mul(v.b, wrap ? tai2_o : tai_o) : gf::mul(v.a, wrap ? tai2_e : tai_e);
      const uint64_t mask = (uint64_t(1) << width) - 1;
      if (a == 1) {
        const uint64_t r = u + carry + (EXT ? ad[k] : 0u);
        dg[k] = __builtin_amdgcn_ubfe(uint32_t(r), 0u, width);
        carry = r >> width;
      } else {
        const uint64_t dlo = u & mask, chi2 = u >> width;
        const uint64_t r = dlo * a + carry + (EXT ? ad[k] : 0u);
        dg[k] = uint32_t(r & mask);
        carry = (r >> width) + chi2 * a;
      }
    }
    uint4* dst = reinterpret_cast<uint4*>(digits) + (size_t(T) * kM1 + i1) * 2;
    dst[0] = make_uint4(dg[0], dg[1], dg[2], dg[3]); dst[1] = make_uint4(dg[4], dg[5], dg[6], dg[7]);
    cbuf[size_t(T) * kM1 + i1] = carry;
    if (EXT && ext.digits2) {
      uint4* d2 = reinterpret_cast<uint4*>(ext.digits2) + (size_t(T) * kM1 + i1) * 2;
      d2[0] = make_uint4(dg[0], dg[1], dg[2], dg[3]); d2[1] = make_uint4(dg[4], dg[5], dg[6], dg[7]);
      ext.cbuf2[size_t(T) * kM1 + i1] = carry;
    }
  }
  PROBE_END(pl)
}

// chain starts omega_m^(i2 (k0 + 5 k1 + 20 k2)) TB[2 i2] and ratios omega_m^(160 i2) of the B3 thread map (and inverses with TBi)
__global__ void __launch_bounds__(640) k_build_f0(DevPlan pl, uint64_t* __restrict__ f0f, uint64_t* __restrict__ f0i, uint64_t* __restrict__ fbf, uint64_t* __restrict__ fbi) {
  const uint32_t t = threadIdx.x, T = blockIdx.x;
  const uint32_t c = t & 3, g2 = (t >> 2) & 7, f1 = (t >> 5) & 3, f0 = t >> 7, i2 = 4 * T + c;
  const uint64_t ea = uint64_t(i2) * (f0 + 5 * f1 + 20 * g2);
  f0f[size_t(T) * kThreads + t] = gf::mul(v2::tw_lookup(pl, ea), pl.TB[2 * i2]);
  f0i[size_t(T) * kThreads + t] = gf::mul(v2::tw_lookup(pl, ea ? pl.m - ea : 0), pl.TBi[2 * i2]);
  if (t < 4) {
    const uint64_t eb = uint64_t(i2) * 160;
    fbf[i2] = v2::tw_lookup(pl, eb);
    fbi[i2] = v2::tw_lookup(pl, eb ? pl.m - eb : 0);
  }
}
}  // namespace v5

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

bool v5_cols_shape(const DevPlan& pl) { return pl.r5 == 5 && pl.M1 == 1280 && pl.C == 4 && pl.M2 >= 8; }
size_t v2_threads_per_tile(const DevPlan& pl) { return v5_cols_shape(pl) ? v5::kThreads : 512; }

hipError_t v2_build_fourstep(const DevPlan& pl, uint64_t* f0f, uint64_t* f0i, uint64_t* fbf, uint64_t* fbi, hipStream_t s) {
  const dim3 grid(pl.M2 / pl.C), block(512);
  if (v5_cols_shape(pl)) {
    hipLaunchKernelGGL(v5::k_build_f0, grid, dim3(v5::kThreads), 0, s, pl, f0f, f0i, fbf, fbi);
    return hipGetLastError();
  }
  switch (pl.M1) {
    case 512: hipLaunchKernelGGL(v2::k_build_f0<1>, grid, block, 0, s, pl, f0f, f0i, fbf, fbi); break;
    case 1024: hipLaunchKernelGGL(v2::k_build_f0<2>, grid, block, 0, s, pl, f0f, f0i, fbf, fbi); break;
    default: hipLaunchKernelGGL(v2::k_build_f0<4>, grid, block, 0, s, pl, f0f, f0i, fbf, fbi); break;
  }
  return hipGetLastError();
}

// ------------------------------- launch wrappers ---------------------------------------------

bool v2_rows_supported(const DevPlan& pl) { return (pl.M2 == 4096 || pl.M2 == 8192) && pl.S2r != nullptr; }
// columns: M1 = 512 R, R in {1, 2, 4}, with C = 8 / R pairs per run (one 4096-pair tile per work-group)
bool v2_cols_supported(const DevPlan& pl) {
  if (v5_cols_shape(pl)) return pl.DI != nullptr;
  return pl.r5 == 1 && (pl.M1 == 512 || pl.M1 == 1024 || pl.M1 == 2048) && pl.M1 * pl.C == 4096 && pl.M2 >= pl.C * 2 && pl.S1r != nullptr &&
         pl.DI != nullptr;
}

#define MI355_SET_LDS(KERNEL, BYTES)                                                                                          \
  { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, int(BYTES)); \
    if (e_ != hipSuccess) return e_; }
hipError_t v2_configure() {
  MI355_SET_LDS((v2::k2_rows4096<0, 1>), v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<1, 1>), v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<2, 1>), v2::kLdsBytes)
  MI355_SET_LDS((v2::k2_rows4096<0, 2>), 2 * v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<1, 2>), 2 * v2::kLdsBytes) MI355_SET_LDS((v2::k2_rows4096<2, 2>), 2 * v2::kLdsBytes)
  MI355_SET_LDS(v2::k1_cols<1>, v2::kLdsBytes) MI355_SET_LDS(v2::k1_cols<2>, v2::kLdsBytes) MI355_SET_LDS(v2::k1_cols<4>, v2::kLdsBytes)
  MI355_SET_LDS(v2::k3_cols<1>, v2::kLdsBytes) MI355_SET_LDS(v2::k3_cols<2>, v2::kLdsBytes) MI355_SET_LDS(v2::k3_cols<4>, v2::kLdsBytes)
  MI355_SET_LDS(v5::k1_cols5, v5::kLdsBytes) MI355_SET_LDS(v5::k3_cols5<false>, v5::kLdsBytes) MI355_SET_LDS(v5::k3_cols5<true>, v5::kLdsBytes)
  MI355_SET_LDS(v2::k3_cols_ext<1>, v2::kLdsBytes) MI355_SET_LDS(v2::k3_cols_ext<2>, v2::kLdsBytes) MI355_SET_LDS(v2::k3_cols_ext<4>, v2::kLdsBytes)
  return hipSuccess;
}
#undef MI355_SET_LDS
hipError_t v2_launch_middle(const DevPlan& pl, const uint64_t* Win, const uint64_t* Y, uint64_t* Wout, int mode, uint32_t sub, hipStream_t s) {
#define MI355_ROWS(MODE, HH) hipLaunchKernelGGL((v2::k2_rows4096<MODE, HH>), dim3(pl.M1), dim3(512 * HH), HH * v2::kLdsBytes, s, pl, Win, Y, Wout, sub)
  if (pl.M2 == 4096) {   // one instantiation per mode: the squaring kernel carries no multiply / image code
    switch (mode) { case 0: MI355_ROWS(0, 1); break; case 1: MI355_ROWS(1, 1); break; default: MI355_ROWS(2, 1); break; }
  } else {
    switch (mode) { case 0: MI355_ROWS(0, 2); break; case 1: MI355_ROWS(1, 2); break; default: MI355_ROWS(2, 2); break; }
  }
#undef MI355_ROWS
  return hipGetLastError();
}
hipError_t v2_launch_front(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint32_t sub, uint64_t* W, hipStream_t s) {
  const dim3 grid(pl.M2 / pl.C), block(512);
  if (v5_cols_shape(pl)) {
    hipLaunchKernelGGL(v5::k1_cols5, grid, dim3(v5::kLaunchThreads), v5::kLdsBytes, s, pl, digits, cbuf_in, sub, W);
    return hipGetLastError();
  }
  switch (pl.M1) {
    case 512: hipLaunchKernelGGL(v2::k1_cols<1>, grid, block, v2::kLdsBytes, s, pl, digits, cbuf_in, sub, W); break;
    case 1024: hipLaunchKernelGGL(v2::k1_cols<2>, grid, block, v2::kLdsBytes, s, pl, digits, cbuf_in, sub, W); break;
    default: hipLaunchKernelGGL(v2::k1_cols<4>, grid, block, v2::kLdsBytes, s, pl, digits, cbuf_in, sub, W); break;
  }
  return hipGetLastError();
}
hipError_t v2_launch_back(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, uint64_t scale, hipStream_t s) {
  const dim3 grid(pl.M2 / pl.C), block(512);
  if (v5_cols_shape(pl)) {
    hipLaunchKernelGGL(v5::k3_cols5<false>, grid, dim3(v5::kLaunchThreads), v5::kLdsBytes, s, pl, W, digits, cbuf, a, scale, BackExt());
    return hipGetLastError();
  }
  switch (pl.M1) {
    case 512: hipLaunchKernelGGL(v2::k3_cols<1>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, scale); break;
    case 1024: hipLaunchKernelGGL(v2::k3_cols<2>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, scale); break;
    default: hipLaunchKernelGGL(v2::k3_cols<4>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, scale); break;
  }
  return hipGetLastError();
}
hipError_t v2_launch_back_ext(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& x, hipStream_t s) {
  const dim3 grid(pl.M2 / pl.C), block(512);
  if (v5_cols_shape(pl)) {
    hipLaunchKernelGGL(v5::k3_cols5<true>, grid, dim3(v5::kLaunchThreads), v5::kLdsBytes, s, pl, W, digits, cbuf, a, uint64_t(1), x);
    return hipGetLastError();
  }
  switch (pl.M1) {
    case 512: hipLaunchKernelGGL(v2::k3_cols_ext<1>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, x); break;
    case 1024: hipLaunchKernelGGL(v2::k3_cols_ext<2>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, x); break;
    default: hipLaunchKernelGGL(v2::k3_cols_ext<4>, grid, block, v2::kLdsBytes, s, pl, W, digits, cbuf, a, x); break;
  }
  return hipGetLastError();
}

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
  if (v5_cols_shape(pl)) {
    const dim3 g5((pl.M2 / pl.C) * grid_mult), b5(v5::kLaunchThreads);
    const size_t l5 = v5::kLdsBytes + size_t(extra_lds);
    const void* f = kind == 0 ? reinterpret_cast<const void*>(v5::k1_cols5) : reinterpret_cast<const void*>(v5::k3_cols5<false>);
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(l5));
    if (e != hipSuccess) return e;
    if (kind == 0) hipLaunchKernelGGL(v5::k1_cols5, g5, b5, l5, s, pl, digits, cbuf, 0u, W);
    else hipLaunchKernelGGL(v5::k3_cols5<false>, g5, b5, l5, s, pl, W, dout, cbuf, 1u, uint64_t(1), BackExt());
    return hipGetLastError();
  }
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
