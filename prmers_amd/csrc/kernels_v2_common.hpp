// Shared pieces of the register-resident kernel sets (kernels_v2.hip: radix-8 rows and columns; kernels_v5.hip: radix-5 columns): pair type,
// LDS exchange macros, the wave-index-specialised shift seams, table lookups, barrier and probe helpers.  Two translation units because the
// two sets want different instruction schedulers (Makefile: kernels_v2.hip is built with -amdgpu-sched-strategy=iterative-maxocc, -1 % at
// C3 and -1.7 % at n = 2^24 in a same-box A/B; the radix-5 kernels lose 0.8 % with it and 40 % with max-ilp: profiles/r03_ab_sched_strategies.txt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gfdft.hpp"
#include "kernels.hpp"

namespace mi355 {
namespace v2 {

struct alignas(16) P2 { uint64_t a, b; };

// LDS slot of tile element i (16-byte slots).  The exchanges write and read the tile in three lane orders -- thread-major (slot 8 t + r),
// strided / wave-major (consecutive lanes on consecutive slots) and the column kernels' (k1 | k2 | c) order -- and a 128-bit access is served
// in fixed lane groups: stores in eight groups of 8 contiguous lanes on 32 banks, loads in four NON-contiguous groups of 16 lanes
// ({0-3, 12-15, 20-27}, ...) on 64 banks (MI355X_MICROARCH.md, LDS).  Round 4: i ^ ((i >> 3) & 15) is conflict-free for every one of the
// row kernel's six orders (census: tools/lds_census.py; the round-1 skew i + i / 8 left the consecutive-lane loads 2-way conflicted -- the 18 %
// of SQ_LDS_BANK_CONFLICT over SQ_LDS_IDX_ACTIVE that VERDICT r03 lists) and needs no padding: 64 KiB instead of 72 per group.
// -DMI355_LDS_ADD3 restores the old skew for A/B builds.
#if defined(MI355_LDS_ADD3)
__device__ __forceinline__ uint32_t phys(uint32_t i) { return i + (i >> 3); }
constexpr uint32_t kLdsSlots = 4096 + 512;
#else
__device__ __forceinline__ uint32_t phys(uint32_t i) { return i ^ ((i >> 3) & 15u); }
constexpr uint32_t kLdsSlots = 4096;
#endif
constexpr uint32_t kLdsBytes = kLdsSlots * 16;

// Work-buffer accesses of the sweeps.  -DMI355_NT=<bits> (A/B builds only): bit 0 stores, bit 1 loads with the non-temporal hint (streamed
// through L2: a sweep never re-reads what it wrote and the next launch reads it from the memory side anyway).
typedef uint64_t u64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void w_store(P2* p, P2 v) {
#if defined(MI355_NT) && (MI355_NT & 1)
  u64x2_t q; q.x = v.a; q.y = v.b;
  __builtin_nontemporal_store(q, reinterpret_cast<u64x2_t*>(p));
#else
  *p = v;
#endif
}
__device__ __forceinline__ P2 w_load(const P2* p) {
#if defined(MI355_NT) && (MI355_NT & 2)
  const u64x2_t q = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(p));
  return {q.x, q.y};
#else
  return *p;
#endif
}

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
// FOLD0: the caller left x[1..3] and x[5] un-folded (dft8 LAZY = 1) because the shifts below accept any operand; wave 0
// shifts by nothing, so it folds them here instead.
template <bool INV, bool FOLD0 = false>
__device__ __forceinline__ void seam64(P2 (&x)[8], uint32_t wave) {
  switch (wave) {
    case 0:
      if (FOLD0) {
#pragma unroll
        for (int k = 1; k < 4; ++k) x[k] = {gf::fold(x[k].a), gf::fold(x[k].b)};
        x[5] = {gf::fold(x[5].a), gf::fold(x[5].b)};
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

// previous run (in digit order) of run (T, i1); see kernels.hip
__device__ __forceinline__ uint64_t carry_in_of(const DevPlan& pl, const uint64_t* cbuf, uint32_t T, uint32_t i1) {
  const uint32_t NT = pl.M2 / pl.C;
  if (T > 0) return cbuf[size_t(T - 1) * pl.M1 + i1];
  return cbuf[size_t(NT - 1) * pl.M1 + (i1 ? i1 - 1 : pl.M1 - 1)];
}

// out[k] = sum_j in[j] omega_4^(jk), omega_4 = 2^48 (forward; the inverse uses omega_4^-1 = -2^48), in place
template <bool INV>
__device__ __forceinline__ void dft4(uint64_t& x0, uint64_t& x1, uint64_t& x2, uint64_t& x3) {
  const uint64_t a0 = gf::add(x0, x2), a1 = gf::add(x1, x3), b0 = gf::sub(x0, x2);
  const uint64_t b1 = gf::mul_pow2(INV ? gf::sub(x3, x1) : gf::sub(x1, x3), 48);
  x0 = gf::add(a0, a1); x2 = gf::sub(a0, a1); x1 = gf::add(b0, b1); x3 = gf::sub(b0, b1);
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

}  // namespace v2

// launch entry points of the radix-5 column shapes (kernels_v5.hip; v5_cols_shape, v5_pfa: kernels.hpp)
size_t v5_threads_per_tile();
hipError_t v5_configure();
hipError_t v5_build_fourstep(const DevPlan& pl, uint64_t* f0f, uint64_t* f0i, uint64_t* fbf, uint64_t* fbi, hipStream_t s);
hipError_t v5_launch_front(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint32_t sub, uint64_t* W, hipStream_t s);
hipError_t v5_launch_back(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, uint64_t scale, hipStream_t s);
hipError_t v5_launch_back_ext(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& x, hipStream_t s);
#if defined(MI355_PROBE)
hipError_t v5_probe_launch(const DevPlan& pl, int kind, int grid_mult, int extra_lds, const uint32_t* digits, uint64_t* cbuf, uint64_t* W, uint32_t* dout, hipStream_t s);
#endif

// radix-4 set for the small transforms (kernels_v3.hip): rows of 1024, columns of 256 with runs of four pairs (1024-pair tiles, 256 threads)
bool v3_rows_shape(const DevPlan& pl);
bool v3_cols_shape(const DevPlan& pl);
size_t v3_threads_per_tile();
hipError_t v3_build_fourstep(const DevPlan& pl, uint64_t* f0f, uint64_t* f0i, uint64_t* fbf, uint64_t* fbi, hipStream_t s);
hipError_t v3_launch_middle(const DevPlan& pl, const uint64_t* Win, const uint64_t* Y, uint64_t* Wout, int mode, uint32_t sub, hipStream_t s);
hipError_t v3_launch_front(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint32_t sub, uint64_t* W, hipStream_t s);
hipError_t v3_launch_back(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, uint64_t scale, hipStream_t s);
hipError_t v3_launch_back_ext(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& x, hipStream_t s);
#if defined(MI355_EXPERIMENTAL)
// back sweep of one squaring + front sweep of the next in one launch (kernels_v3.hip k31_cols256_planes): tiles served (0: not), launch
uint32_t v3_chain_tiles(const DevPlan& pl, int device);
hipError_t v3_launch_backfront(const DevPlan& pl, uint64_t* W, uint32_t a, uint32_t sub, uint64_t* xbuf, uint32_t* flags, uint32_t epoch, hipStream_t s);
#endif

}  // namespace mi355
