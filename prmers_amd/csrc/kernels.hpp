// Device-side view of a Plan and the kernel launch entry points (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

struct DevPlan {
  uint32_t n, m, M1, M2, L1, logL1, logM2, r5, C, logC, q, t, twh;
  const uint32_t *SA, *SB;
  const uint64_t *TA, *TAi, *TB, *TBi;
  const uint64_t *TAh, *TAi2;   // TA / 2 and 2 TAi, entry by entry: what the sweeps multiply with when the exponent split wraps (no per-thread half / double)
  const uint64_t *TWlo, *TWhi, *UT1, *UT2;
  const uint64_t *S2r, *S2ri, *S1r, *S1ri;   // seam tables of the radix-8 kernels (null when the shape is not served)
  uint64_t I4, I4inv;
  uint64_t W5c[4];   // 5-point DFT constants {beta, k1, k2-k1, k1+k2} (kernels.hip dft5)
  const uint64_t *F0f, *F0i, *FBf, *FBi;   // four-step chain starts [tile][thread] and ratios [column] of the v2 column kernels
  const uint32_t* DI;   // digit-info words of the v2 column kernels: [tile][thread] 16 x (width - q, wrap), or null
  uint32_t boost_rows, boost_tiles;   // first block index of the last half round of the row / column launches (or ~0u)
  // Frequency label of the column-transform output slot (blk, rq): lab_u blk + lab_v rq.  (1, r5) for the mixed-radix columns of every kernel
  // set but one; the radix-5 columns in prime-factor form (kernels_v5.hip) hold the frequency (PU k0 + PV kr) mod M1 in slot (k0, kr) and the
  // engine sets (PU, PV) when they run both column sweeps.  See col_label() below.
  uint32_t lab_u, lab_v;
  uint32_t lab_red;   // the generic prime-factor radix-5 stage (kernels.hip lds_radix5): labels are taken mod M1 (lab_red = M1; they are below 5 M1); 0: as they are
#if defined(MI355_PROBE)
  uint64_t* probe;        // timeline probe (tools/probe.py, libmi355_engine_probe.so only): 8 words per work-group, or null
  uint32_t probe_mod;     // blockIdx.x is taken modulo this (launches of several rounds over the same tiles)
#endif
  uint32_t tune;   // MI355_TUNE bit 0: plain (not XCD-contiguous) tile order in the back sweep, for A/B runs; bit 2: no issue-priority boost of the last half round
};

#if defined(__HIPCC__)
// Frequency label of the column-transform output slot (blk, rq) -- rq the bit-reversed position inside the radix-5 block (kernels.hip freq1).
// Every use of a column frequency k1 (the four-step twiddle omega_m^(i2 k1), the point rho = omega_m^(k1 + M1 k) of the pointwise stage) only
// needs SOME integer congruent to it mod M1, the same one in the front sweep, the row sweep and the back sweep: a representative k1 + M1 s
// shifts the row transform's outputs by s places and the total frequency k1 + M1 k2 stays what the label says.  The prime-factor columns use
// PU blk + PV rq unreduced (< 2^20; their twiddle chains step through rq by constant ratios); the two-level root table reaches m + 2^20.
__device__ __forceinline__ uint32_t col_label(const DevPlan& pl, uint32_t blk, uint32_t rq) {
  uint32_t l = pl.lab_u * blk + pl.lab_v * rq;
  if (pl.lab_red) {   // the generic kernels multiply a label by a column index and need the product below m: the representative below M1
#pragma unroll
    for (int i = 0; i < 4; ++i) l = (l >= pl.lab_red) ? l - pl.lab_red : l;
  }
  return l;
}
// exponent of rho = omega_m^(label + M1 k): below m + 2^20 (plan.hpp TWhi)
__device__ __forceinline__ uint64_t rho_exponent(const DevPlan& pl, uint32_t label, uint64_t k) { return uint64_t(label) + uint64_t(pl.M1) * k; }
#endif

// Extras of a back sweep (the reference's fused carry variants, kernels/marin.cl:2160-2365):
//   digits2 / cbuf2: a second register that receives the same result (square_mul_copy, mul_copy)
//   add_digits (+ add_cbuf: its run carries when they are still pending): a residue added inside the carry chain (mul_add)
struct BackExt {
  uint32_t* digits2 = nullptr; uint64_t* cbuf2 = nullptr;
  const uint32_t* add_digits = nullptr; const uint64_t* add_cbuf = nullptr;
  bool any() const { return digits2 || add_digits; }
};
// Linear combinations of digit registers in one sweep (add / sub_reg / addsub / addsub_copy, marin.cl:1856-1947):
// a, b with their pending run carries (null when none); outputs sum (s1, s2) and difference (d1, d2), each with the carry
// words it leaves pending.  Digit outputs may alias the inputs; carry outputs must not alias ca / cb.
struct LinArgs {
  const uint32_t* a = nullptr; const uint64_t* ca = nullptr; const uint32_t* b = nullptr; const uint64_t* cb = nullptr;
  uint32_t* s1 = nullptr; uint64_t* cs1 = nullptr; uint32_t* s2 = nullptr; uint64_t* cs2 = nullptr;
  uint32_t* d1 = nullptr; uint64_t* cd1 = nullptr; uint32_t* d2 = nullptr; uint64_t* cd2 = nullptr;
};
hipError_t launch_linear(const DevPlan& pl, const LinArgs& la, hipStream_t s);
hipError_t launch_back_ext(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& x, hipStream_t s);
hipError_t v2_launch_back_ext(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& x, hipStream_t s);

hipError_t configure_kernels(size_t lds_front, size_t lds_mid);
// cbuf_in (nullable, needs C >= 2): run carries left by the last back sweep, folded into the load
hipError_t launch_front(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint64_t* W, hipStream_t s);
hipError_t launch_middle(const DevPlan& pl, const uint64_t* Win, const uint64_t* Y, uint64_t* Wout, int mode, uint32_t sub, hipStream_t s);
hipError_t launch_back(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, hipStream_t s);
hipError_t launch_carry_fix(const DevPlan& pl, uint32_t* digits, const uint64_t* cbuf, hipStream_t s);
#if defined(MI355_EXPERIMENTAL)   // libmi355_engine_exp.so only (make exp): measured slower than three launches, DESIGN.md 5.2c
// `count` squarings in ONE cooperative launch (kernels.hip k_coop): grid size for this plan, or 0 when the plan is not served
uint32_t coop_groups(const DevPlan& pl, int device);
// flags: `groups` barrier words, err: error word (both device-visible, zero at first use); epoch0: barriers passed so far on these flags
// (each squaring passes 3, the last one of a launch 2); sub: subtracted before the first squaring, sub_next before each later one
hipError_t launch_coop(const DevPlan& pl, uint32_t groups, uint32_t* digits, uint64_t* cbuf, bool carry_in, uint64_t* W, uint32_t a, uint32_t sub, uint32_t sub_next,
                       uint32_t count, uint32_t* flags, uint32_t* err, uint32_t epoch0, uint32_t fault, hipStream_t s);   // fault: test hook, see CoopArgs
#endif
// columns of 5 L1 pairs that do not fit LDS (n = 5 * 2^26): the radix-5 stage through a second work buffer U (8 n bytes), C = 1
hipError_t configure_split(const DevPlan& pl);
hipError_t launch_front_split(const DevPlan& pl, const uint32_t* digits, uint64_t* U, uint64_t* W, hipStream_t s);
hipError_t launch_back_split(const DevPlan& pl, const uint64_t* W, uint64_t* U, uint32_t* digits, uint64_t* cbuf, uint32_t a, hipStream_t s);
hipError_t launch_addsub(const DevPlan& pl, uint32_t* dst, const uint32_t* src, uint64_t* cbuf, int negate, hipStream_t s);
hipError_t launch_sub_small(const DevPlan& pl, uint32_t* digits, uint32_t a, hipStream_t s);


// radix-5 column shapes of kernels_v5.hip (M1 = 1280 = 5 x 256 with C = 4, M1 = 2560 = 5 x 512 with C = 2) and the frequency map of their
// prime-factor form (DevPlan.lab_u / lab_v)
inline bool v5_cols_shape(const DevPlan& pl) { return pl.r5 == 5 && ((pl.M1 == 1280 && pl.C == 4) || (pl.M1 == 2560 && pl.C == 2)) && pl.M2 >= 8; }
void v5_pfa(const DevPlan& pl, uint32_t* u, uint32_t* v);

// register-resident radix-8 set (kernels_v2.hip); shapes: rows M2 = 4096, columns M1 = 1024 x C = 4
bool v2_rows_supported(const DevPlan& pl);
bool v2_cols_supported(const DevPlan& pl);
size_t v2_threads_per_tile(const DevPlan& pl);   // 512 (columns of 512 R) or 640 (columns of 1280 = 5 x 256)
hipError_t v2_configure();
// fills the chain-start / ratio tables of the column kernels' four-step twiddles (tiles*512, tiles*512, M2, M2 words)
hipError_t v2_build_fourstep(const DevPlan& pl, uint64_t* f0f, uint64_t* f0i, uint64_t* fbf, uint64_t* fbi, hipStream_t s);
hipError_t v2_launch_middle(const DevPlan& pl, const uint64_t* Win, const uint64_t* Y, uint64_t* Wout, int mode, uint32_t sub, hipStream_t s);
hipError_t v2_launch_front(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint32_t sub, uint64_t* W, hipStream_t s);
hipError_t v2_launch_back(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, uint64_t scale, hipStream_t s);
#if defined(MI355_EXPERIMENTAL)
// runs of squarings on the radix-8 column shapes (kernels_v2.hip k31_cols): the back sweep of one squaring and the front sweep of the next in one
// launch (a = 1).  xbuf: one hand-over word per run (tiles x M1), err: the error word a timed-out wait raises, tag: 1 .. 4095, different from
// the previous launch's on the same xbuf
bool v2_chain_supported(const DevPlan& pl);
hipError_t v2_launch_backfront(const DevPlan& pl, uint64_t* W, uint32_t sub, uint64_t* xbuf, uint32_t* err, uint32_t tag, hipStream_t s);
// runs of squarings on the small shapes (kernels_v3.hip): the back sweep of one squaring and the front sweep of the next in one launch.
// v3_chain_tiles: tiles of such a launch (all resident at once on `device`), 0 where the plan is not served.  xbuf: tiles x 256 carry
// words, flags: tiles + 1 words (the last one is the error word a timed-out wait raises), epoch: larger than any used before on these flags
uint32_t v3_chain_tiles(const DevPlan& pl, int device);
hipError_t v3_launch_backfront(const DevPlan& pl, uint64_t* W, uint32_t a, uint32_t sub, uint64_t* xbuf, uint32_t* flags, uint32_t epoch, hipStream_t s);
#endif
#if defined(MI355_PROBE)
size_t v2_lds_bytes();
// one launch of sweep `kind` (0 front, 1 rows, 2 back) over grid_mult x the normal grid with extra_lds bytes of padding LDS
hipError_t v2_probe_launch(const DevPlan& pl, int kind, int grid_mult, int extra_lds, const uint32_t* digits, uint64_t* cbuf, uint64_t* W, uint32_t* dout, hipStream_t s);
#endif

// device-side canonical form (canon.hip): strong carry with wrap-around into natural order, compare, scatter
size_t canon_scratch_words(const DevPlan& pl);
hipError_t canon_launch(const DevPlan& pl, uint32_t p, const uint32_t* digits, uint32_t* out, uint32_t* scratch, hipStream_t s);
uint32_t* canon_flags(const DevPlan& pl, uint32_t* scratch);
hipError_t canon_compare(const uint32_t* a, const uint32_t* b, uint32_t n, uint32_t* diff_flag, hipStream_t s);
hipError_t canon_relax(const DevPlan& pl, uint32_t p, const uint32_t* in, uint32_t* out, hipStream_t s);   // one local carry pass, tile-major both sides
hipError_t canon_scatter(const DevPlan& pl, uint32_t p, const uint32_t* nat, uint32_t* digits, hipStream_t s);
hipError_t canon_set_small(const DevPlan& pl, uint32_t p, uint32_t* digits, uint32_t value, hipStream_t s);

}  // namespace mi355
