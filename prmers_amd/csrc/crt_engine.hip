// Squaring x <- x^2 a mod 2^p - 1 over GF(M61^2) x GF(M31^2) with a prime-factor (Good-Thomas) axis of radix 1, 3 or 9:
// the second field family of the reference.  Reference: the Aevum backend, third_party/aevum/src/cl/
// fft-middle.cl:663-720 (pfaDft3, radix 9 = 3 x 3 with scalar roots), pfaunpack.cl:12-56 (index map), carry.cl:506-588, policy
// README.md:907-926; CPU illustration docs/mersenne2_mixed_crt_2d_half_fast/mersenne2_mixed_crt_2d_half_fast.cpp ("m2:").
//
// n = odd * m words of up to 39 bits, m = 2^ln.  Logical digit j sits at grid coordinate (a, b) = (j mod odd, j mod m); there are no
// twiddles between the two axes (m2:733-758).  Row a of the grid is a real sequence of length m, held as h = m / 2 values of
// Z/p[i] (slot s = (b = 2s) + i (b = 2s + 1)), once for p = M61 (16 bytes a slot) and once for p = M31 (8 bytes): 12 bytes a word.
//   front      digits -> weight (bit rotations) -> DFT of length odd along a with scalar roots -> Z[a][s]
//   rows       half-length complex DFT of every row, h = H1 x H2 four-step (columns of H1 through LDS, twiddle omega_h^(k1 i2),
//              rows of H2 through LDS); frequency k = k1 + H1 k2 ends at slot k1 H2 + k2
//   pointwise  conjugate-symmetric untangling of the packed real rows, square, re-tangle (m2:829-915 in its textbook split form)
//   rows^-1, back: inverse odd DFT, 1 / (odd h), scatter to logical order; then the fused unweight + Garner + carry sweep of
//              crt_carry.hip.
// This file is the straightforward kernel set (radix-2 butterflies in LDS, one launch per stage): parity first, see DESIGN.md for
// the measured cost and what the register-resident version has to beat.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "crt_engine.hpp"
#include "crt_field.hpp"

namespace mi355 {
namespace crt {

constexpr uint32_t kPassElems = 2048;   // complex values of one work-group of a row pass (32 KiB of LDS for M61)

struct Grid {
  uint32_t odd, ln, m, h, logh, logH1, logH2, minv;   // h = 2^logh = H1 H2; minv = m^-1 mod odd
  uint64_t r61[9], r61i[9], s61, c3_61;               // odd-root powers r^e, their inverses, 1 / (odd h), (w3 - w3^2) / 2
  uint32_t r31[9], r31i[9], s31, c3_31;
  uint32_t mm, pm, lpm61, lpm31;                      // m mod odd; p m mod n and its images l61 (p m) mod 61, l31 (p m) mod 31
  uint32_t tune;                                      // MI355_CRT_TUNE (A/B runs): bit 0 plain tile order in the column kernels, bit 1 back and carry as two kernels
};

template <class F>
__device__ __forceinline__ typename F::C tw_m(const typename F::C* __restrict__ U, uint32_t e, uint32_t h) {   // omega_m^e, e < m = 2h
  return e < h ? U[e] : cneg<F>(U[e - h]);
}

__device__ __forceinline__ uint32_t brev(uint32_t i, uint32_t bits) { return bits ? (__brev(i) >> (32 - bits)) : 0u; }

// ---- odd axis: DFT of length 1, 3 or 9 with scalar roots --------------------------------------------------------------------
// DFT-3 with w + w^2 = -1: y0 = x0 + (x1 + x2), y1,2 = x0 - (x1 + x2) / 2 +- c (x1 - x2), c = (w - w^2) / 2: one scalar product.
// DFT-9 = 3 x 3 (Cooley-Tukey): three DFT-3 over a1 (a = 3 a1 + a0), twiddles r^(a0 k0) (four non-trivial), three DFT-3 over a0
// -> X[k0 + 3 k1]: 10 scalar-times-complex products instead of the 64 of the direct sums.  (Reference: fft-middle.cl:663-720.)
template <class F>
__device__ __forceinline__ void dft3(typename F::C& x0, typename F::C& x1, typename F::C& x2, typename F::S c) {
  using C = typename F::C;
  const C t1 = cadd<F>(x1, x2), t2 = cscale<F>(csub<F>(x1, x2), c);
  const C u = csub<F>(x0, chalf<F>(t1));
  x0 = cadd<F>(x0, t1); x1 = cadd<F>(u, t2); x2 = csub<F>(u, t2);
}
template <class F, int ODD>
__device__ __forceinline__ void dft_odd(typename F::C (&x)[ODD], const typename F::S* __restrict__ r /* r^e, e < 9 */, typename F::S c3) {
  using C = typename F::C;
  if (ODD == 3) {
    dft3<F>(x[0], x[1], x[2], c3);
  } else if (ODD == 9) {
#pragma unroll
    for (int a0 = 0; a0 < 3; ++a0) dft3<F>(x[a0], x[a0 + 3], x[a0 + 6], c3);     // x[a0 + 3 k0] <- Y[a0][k0]
    x[1 + 3] = cscale<F>(x[1 + 3], r[1]); x[1 + 6] = cscale<F>(x[1 + 6], r[2]);
    x[2 + 3] = cscale<F>(x[2 + 3], r[2]); x[2 + 6] = cscale<F>(x[2 + 6], r[4]);
    C y[9];
#pragma unroll
    for (int k0 = 0; k0 < 3; ++k0) {
      C z0 = x[3 * k0], z1 = x[3 * k0 + 1], z2 = x[3 * k0 + 2];
      dft3<F>(z0, z1, z2, c3);
      y[k0] = z0; y[k0 + 3] = z1; y[k0 + 6] = z2;                                // X[k0 + 3 k1]
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) x[k] = y[k];
  }
}

// weight exponents of the digits b + m t, t = 0 .. odd-1, kept incrementally: s = p j mod n advances by p m mod n
struct ColumnWalk {
  uint32_t s, A61, A31;
  __device__ __forceinline__ void step_t(const Geom& g, const Grid& gr) {
    uint32_t sn = s + gr.pm;
    A61 += gr.lpm61; A31 += gr.lpm31;
    if (sn >= g.n) { sn -= g.n; A61 += 60; A31 += 30; }
    s = sn;
    A61 = A61 >= 122 ? A61 - 122 : (A61 >= 61 ? A61 - 61 : A61);
    A31 = A31 >= 62 ? A31 - 62 : (A31 >= 31 ? A31 - 31 : A31);
  }
};

// ---- front: weight + odd axis --------------------------------------------------------------------------------------------
// thread = slot s (b = 2s, 2s + 1) of every row.  For t = 0 .. odd-1 the digit pair (b + m t) is one 16-byte load; digit j belongs to
// row a = j mod odd, which changes with t and b: the weighted values go through a private LDS column ([a][thread], no barrier) to
// reach the registers of the odd-axis DFT in row order.
template <int ODD>
__global__ void __launch_bounds__(256) k_front(Geom g, Grid gr, const uint64_t* __restrict__ x, F61::C* __restrict__ Z61, F31::C* __restrict__ Z31) {
  __shared__ uint64_t S61[2 * ODD][256];
  __shared__ uint32_t S31[2 * ODD][256];
  const uint32_t tid = threadIdx.x, s = blockIdx.x * 256 + tid;
  if (s >= gr.h) return;
  const uint32_t b = 2 * s;
  DigitWalk d0; d0.start(g, b);
  ColumnWalk w{d0.s, d0.A61, d0.A31};
  uint32_t a = b % ODD;                                   // row of digit b + m t
#pragma unroll
  for (int t = 0; t < ODD; ++t) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(x + b + size_t(gr.m) * t);
    DigitWalk d; d.s = w.s; d.A61 = w.A61; d.A31 = w.A31;
    const uint32_t a1 = (a + 1 == ODD) ? 0 : a + 1;       // digit b + 1 + m t sits one row further
    S61[2 * a][tid] = rot61(red61(v.x), d.weight61()); S31[2 * a][tid] = rot31(red31(v.x), d.weight31());
    d.next(g);
    S61[2 * a1 + 1][tid] = rot61(red61(v.y), d.weight61()); S31[2 * a1 + 1][tid] = rot31(red31(v.y), d.weight31());
    w.step_t(g, gr);
    a += gr.mm; if (a >= ODD) a -= ODD;
  }
  F61::C in61[ODD]; F31::C in31[ODD];
#pragma unroll
  for (int k = 0; k < ODD; ++k) { in61[k] = {S61[2 * k][tid], S61[2 * k + 1][tid]}; in31[k] = {S31[2 * k][tid], S31[2 * k + 1][tid]}; }
  dft_odd<F61, ODD>(in61, gr.r61, gr.c3_61);
  dft_odd<F31, ODD>(in31, gr.r31, gr.c3_31);
#pragma unroll
  for (int ka = 0; ka < ODD; ++ka) { Z61[size_t(ka) * gr.h + s] = in61[ka]; Z31[size_t(ka) * gr.h + s] = in31[ka]; }
}

// ---- back: inverse odd axis, 1 / (odd h), to logical order (16-byte and 8-byte stores of digit pairs) ----------------------
template <int ODD>
__global__ void __launch_bounds__(256) k_back(Grid gr, const F61::C* __restrict__ Z61, const F31::C* __restrict__ Z31, uint64_t* __restrict__ out61,
                                              uint32_t* __restrict__ out31) {
  __shared__ uint64_t S61[2 * ODD][256];
  __shared__ uint32_t S31[2 * ODD][256];
  const uint32_t tid = threadIdx.x, s = blockIdx.x * 256 + tid;
  if (s >= gr.h) return;
  F61::C in61[ODD]; F31::C in31[ODD];
#pragma unroll
  for (int k = 0; k < ODD; ++k) { in61[k] = Z61[size_t(k) * gr.h + s]; in31[k] = Z31[size_t(k) * gr.h + s]; }
  dft_odd<F61, ODD>(in61, gr.r61i, F61::neg(gr.c3_61));
  dft_odd<F31, ODD>(in31, gr.r31i, F31::neg(gr.c3_31));
#pragma unroll
  for (int k = 0; k < ODD; ++k) {
    const F61::C o61 = cscale<F61>(in61[k], gr.s61); const F31::C o31 = cscale<F31>(in31[k], gr.s31);
    S61[2 * k][tid] = o61.re; S61[2 * k + 1][tid] = o61.im; S31[2 * k][tid] = o31.re; S31[2 * k + 1][tid] = o31.im;
  }
  const uint32_t b = 2 * s;
  uint32_t a = b % ODD;
#pragma unroll
  for (int t = 0; t < ODD; ++t) {
    const uint32_t a1 = (a + 1 == ODD) ? 0 : a + 1;
    const size_t j = b + size_t(gr.m) * t;
    *reinterpret_cast<ulonglong2*>(out61 + j) = make_ulonglong2(S61[2 * a][tid], S61[2 * a1 + 1][tid]);
    *reinterpret_cast<uint2*>(out31 + j) = make_uint2(S31[2 * a][tid], S31[2 * a1 + 1][tid]);
    a += gr.mm; if (a >= ODD) a -= ODD;
  }
}

// ---- back + carry in one kernel (round 3): the inverse odd axis of 256 slots, then the unweight + Garner + carry sweep of crt_carry.hip on
// the 2 x 256 x ODD digits they hold, straight out of LDS -- the unweighted residues (12 bytes a word) no longer make a round trip
// through HBM between k_back and k_crt_runs_linked, and one launch goes.  The slots s0 .. s0 + 255 of a work-group hold ODD ranges of
// 512 consecutive digits, [2 s0 + m t, 2 s0 + 512 + m t): one thread per run of kRun digits (64 runs a range, ODD x 64 virtual threads on
// 256 real ones), sequential carry inside a run, run-to-run hand-over through LDS inside a range exactly as in k_crt_runs_linked;
// edge_out[3 (ODD g + t) ..] = 128-bit carry and leftover of the last run of range t, folded into the following range by
// k_crt_range_edges (n / 512 threads).  Reference: third_party/aevum/src/cl/carry.cl:506-588 (carry), fft-middle.cl:663-720 (pfaDft).
template <int ODD>
__global__ void __launch_bounds__(256) k_back_carry(Geom g, Grid gr, const F61::C* __restrict__ Z61, const F31::C* __restrict__ Z31, uint64_t* __restrict__ digits,
                                                    uint64_t* __restrict__ edge_out) {
  __shared__ uint64_t S61[2 * ODD][256];
  __shared__ uint32_t S31[2 * ODD][256];
  __shared__ uint64_t Clo[256], Chi[256], Rs[256];
  const uint32_t tid = threadIdx.x, s0 = blockIdx.x * 256, s = s0 + tid;   // h is a multiple of 256 on this path (checked by the launcher)
  {
    F61::C in61[ODD]; F31::C in31[ODD];
#pragma unroll
    for (int k = 0; k < ODD; ++k) { in61[k] = Z61[size_t(k) * gr.h + s]; in31[k] = Z31[size_t(k) * gr.h + s]; }
    dft_odd<F61, ODD>(in61, gr.r61i, F61::neg(gr.c3_61));
    dft_odd<F31, ODD>(in31, gr.r31i, F31::neg(gr.c3_31));
#pragma unroll
    for (int k = 0; k < ODD; ++k) {   // row k at this slot: re = the row's even position 2 s, im = the odd one
      const F61::C o61 = cscale<F61>(in61[k], gr.s61); const F31::C o31 = cscale<F31>(in31[k], gr.s31);
      S61[2 * k][tid] = o61.re; S61[2 * k + 1][tid] = o61.im; S31[2 * k][tid] = o31.re; S31[2 * k + 1][tid] = o31.im;
    }
  }
  __syncthreads();
  for (uint32_t base = 0; base < uint32_t(ODD) * 64u; base += 256u) {
    const uint32_t vt = base + tid, t = vt >> 6, r = tid & 63u;   // range, run inside the range (base is a multiple of 256)
    const bool live = vt < uint32_t(ODD) * 64u;
    uint64_t out[kRun]; uint32_t wd[kRun];
    unsigned __int128 carry = 0;
    uint32_t j0 = 0;
    if (live) {
      const uint32_t sl0 = 4u * r;
      j0 = 2u * (s0 + sl0) + gr.m * t;
      DigitWalk dw; dw.start(g, j0);
      const uint32_t a0 = (2u * (s0 + sl0) + gr.mm * t) % uint32_t(ODD);   // row of digit j0 (j mod ODD with m = mm mod ODD)
#pragma unroll
      for (int k = 0; k < kRun; ++k) {
        const uint32_t sl = sl0 + uint32_t(k >> 1);
        const uint32_t a = (a0 + uint32_t(k)) % uint32_t(ODD);            // digit j0 + k sits in row (j0 + k) mod ODD
        const uint32_t plane = 2u * a + uint32_t(k & 1);                   // even position: re, odd position: im
        const uint64_t x61 = rot61(S61[plane][sl], dw.unweight61());
        const uint32_t x31 = rot31(S31[plane][sl], dw.unweight31());
        const uint64_t d = x61 >= x31 ? x61 - x31 : x61 + M61 - x31;       // Garner, as in k_crt_runs
        const uint64_t tt = mul61(d, g.inv31);
        const unsigned __int128 v = ((unsigned __int128)tt << 31) - tt + x31;
        const unsigned __int128 sum = v * g.a + carry;
        wd[k] = dw.width(g);
        out[k] = uint64_t(sum) & ((uint64_t(1) << wd[k]) - 1);
        carry = sum >> wd[k];
        dw.next(g);
      }
    }
    Clo[tid] = uint64_t(carry); Chi[tid] = uint64_t(carry >> 64);
    __syncthreads();
    unsigned __int128 in = r ? (((unsigned __int128)Chi[tid - 1] << 64) | Clo[tid - 1]) : 0;
    if (live) {
#pragma unroll
      for (int k = 0; k < kRun; ++k) {
        const unsigned __int128 sum = (unsigned __int128)out[k] + in;
        out[k] = uint64_t(sum) & ((uint64_t(1) << wd[k]) - 1);
        in = sum >> wd[k];
      }
    }
    Rs[tid] = uint64_t(in);
    __syncthreads();
    if (live) {
      if (r) out[0] += Rs[tid - 1];
      ulonglong2* po = reinterpret_cast<ulonglong2*>(digits + j0);
#pragma unroll
      for (int k = 0; k < kRun / 2; ++k) po[k] = make_ulonglong2(out[2 * k], out[2 * k + 1]);
      if (r == 63u) {
        uint64_t* eo = edge_out + 3 * (size_t(blockIdx.x) * ODD + t);
        eo[0] = uint64_t(carry); eo[1] = uint64_t(carry >> 64); eo[2] = uint64_t(in);
      }
    }
    __syncthreads();   // Clo / Chi / Rs are reused by the next ranges
  }
}
// first run of every range: the carry of the range before it in digit order (same t of the previous work-group; the last group's range t - 1
// for the first group; cyclically, 2^p = 1) runs through its digits, the leftovers go in front of this run and of the next one without
// further propagation (weak carry) -- k_crt_edges for the ranges of k_back_carry
__global__ void __launch_bounds__(256) k_crt_range_edges(Geom g, Grid gr, uint64_t* __restrict__ digits, const uint64_t* __restrict__ edge) {
  const uint32_t idx = blockIdx.x * 256 + threadIdx.x, G = gr.h >> 8, odd = gr.odd;
  if (idx >= G * odd) return;
  const uint32_t grp = idx / odd, t = idx - grp * odd;
  const uint32_t prev = grp ? (grp - 1) * odd + t : (G - 1) * odd + (t ? t - 1 : odd - 1);
  unsigned __int128 carry = ((unsigned __int128)edge[3 * size_t(prev) + 1] << 64) | edge[3 * size_t(prev)];
  const uint32_t j0 = 512u * grp + gr.m * t;
  DigitWalk dw; dw.start(g, j0);
  for (int k = 0; k < kRun; ++k) {
    const uint32_t width = dw.width(g);
    const unsigned __int128 sum = (unsigned __int128)digits[j0 + k] + carry;
    digits[j0 + k] = uint64_t(sum) & ((uint64_t(1) << width) - 1);
    carry = sum >> width;
    if (carry == 0) break;
    dw.next(g);
  }
  digits[j0] += edge[3 * size_t(prev) + 2];
  if (carry) digits[j0 + kRun] += uint64_t(carry);   // (a range has 512 digits: still inside it)
}

// ---- one pass of the row transforms --------------------------------------------------------------------------------------
// A work-group holds CA transforms of length L = 2^logL in LDS.  cols != 0: the transforms are the columns i2 .. i2 + CA - 1 of
// the H1 x H2 view of one row (stride H2), followed (forward) or preceded (inverse) by the four-step twiddle omega_h^(+-k1 i2);
// cols == 0: they are CA consecutive contiguous segments (the rows of that view, or whole grid rows when H1 = 1).
// Radix-2 decimation in frequency; the bit-reversed result is read back in natural order.
template <class F>
__global__ void __launch_bounds__(256) k_pass(Grid gr, typename F::C* __restrict__ Z, const typename F::C* __restrict__ U, uint32_t logL, uint32_t CA,
                                              int cols, int inverse) {
  using C = typename F::C;
  __shared__ C X[kPassElems];
  const uint32_t L = 1u << logL, tid = threadIdx.x, nt = blockDim.x;
  const uint32_t per_row = gr.h >> logL;               // transforms per grid row
  const uint32_t d0 = blockIdx.x * CA;                 // first transform of this group (CA divides per_row)
  const uint32_t row = d0 / per_row, r0 = d0 - row * per_row;
  C* base = Z + size_t(row) * gr.h;
  const uint32_t H2 = 1u << gr.logH2;
  // load
  for (uint32_t e = tid; e < CA * L; e += nt) {
    uint32_t c, i; size_t addr;
    if (cols) { c = e % CA; i = e / CA; addr = size_t(i) * H2 + (r0 + c); }
    else { i = e & (L - 1); c = e >> logL; addr = size_t(r0 + c) * L + i; }
    C v = base[addr];
    if (cols && inverse) {   // conj(omega_h^(k1 i2)) = conj(omega_m^(2 k1 i2)): here i is k1
      v = cmul<F>(v, cconj<F>(tw_m<F>(U, 2u * i * (r0 + c), gr.h)));
    }
    X[c * L + i] = v;
  }
  // butterflies
  const uint32_t ushift = gr.ln - logL;                // omega_L^j = omega_m^(j m / L)
  for (uint32_t half = L >> 1, sh = 0; half >= 1; half >>= 1, ++sh) {
    __syncthreads();
    for (uint32_t bfy = tid; bfy < CA * (L >> 1); bfy += nt) {
      const uint32_t c = bfy / (L >> 1), q = bfy - c * (L >> 1);
      const uint32_t j = q & (half - 1), i = ((q - j) << 1) + j;
      const C u = X[c * L + i], v = X[c * L + i + half];
      C w = U[size_t(j << sh) << ushift];
      if (inverse) w = cconj<F>(w);
      X[c * L + i] = cadd<F>(u, v);
      X[c * L + i + half] = (j == 0) ? csub<F>(u, v) : cmul<F>(csub<F>(u, v), w);
    }
  }
  __syncthreads();
  // store, natural order
  for (uint32_t e = tid; e < CA * L; e += nt) {
    uint32_t c, k; size_t addr;
    if (cols) { c = e % CA; k = e / CA; addr = size_t(k) * H2 + (r0 + c); }
    else { k = e & (L - 1); c = e >> logL; addr = size_t(r0 + c) * L + k; }
    C v = X[c * L + brev(k, logL)];
    if (cols && !inverse) v = cmul<F>(v, tw_m<F>(U, 2u * k * (r0 + c), gr.h));
    base[addr] = v;
  }
}

// ---- pointwise -----------------------------------------------------------------------------------------------------------
// Row of m reals packed as h complex values z; Z = DFT_h(z).  With W = omega_m:
//   X_k = (Z_k + conj Z_{-k}) / 2 + W^k (Z_k - conj Z_{-k}) / (2i)          k = 0 .. h      (the real sequence's spectrum)
//   Y_k = X_k^2
//   Z'_k = (Y_k + conj Y_{h-k}) / 2 + i conj(W^k) (Y_k - conj Y_{h-k}) / 2  k = 0 .. h - 1  (packed spectrum of the square)
// One thread owns the pair (k, h - k), k <= h / 2; frequency k = k1 + H1 k2 sits at slot k1 H2 + k2.
__device__ __forceinline__ uint32_t slot_of(const Grid& gr, uint32_t k) { return ((k & ((1u << gr.logH1) - 1)) << gr.logH2) + (k >> gr.logH1); }

template <class F>
__device__ __forceinline__ typename F::C spectrum_sq(typename F::C zk, typename F::C zmk, typename F::C w) {
  const typename F::C zc = cconj<F>(zmk);
  const typename F::C e = cadd<F>(zk, zc), o = cdiv_i<F>(csub<F>(zk, zc));
  return csqr<F>(chalf<F>(cadd<F>(e, cmul<F>(w, o))));
}
template <class F>
__device__ __forceinline__ typename F::C repack(typename F::C yk, typename F::C yhk, typename F::C w) {
  const typename F::C yc = cconj<F>(yhk);
  const typename F::C e = cadd<F>(yk, yc), d = cmul<F>(csub<F>(yk, yc), cconj<F>(w));
  return chalf<F>(cadd<F>(e, cmul_i<F>(d)));
}

template <class F>
__device__ __forceinline__ typename F::C spectrum_lin(typename F::C zk, typename F::C zmk, typename F::C w) {
  const typename F::C zc = cconj<F>(zmk);
  const typename F::C e = cadd<F>(zk, zc), o = cdiv_i<F>(csub<F>(zk, zc));
  return chalf<F>(cadd<F>(e, cmul<F>(w, o)));
}
// I == nullptr: square; otherwise multiply by the packed spectrum I (same slot order)
template <class F>
__global__ void __launch_bounds__(256) k_pointwise(Grid gr, typename F::C* __restrict__ Z, const typename F::C* __restrict__ U, const typename F::C* __restrict__ I) {
  using C = typename F::C;
  const uint32_t per_row = (gr.h >> 1) + 1;
  const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= per_row * gr.odd) return;
  const uint32_t row = idx / per_row, k = idx - row * per_row;
  C* z = Z + size_t(row) * gr.h;
  const C* im = I ? I + size_t(row) * gr.h : nullptr;
  const uint32_t h = gr.h;
  auto prod = [&](C zk, C zmk, C ik, C imk, C w) {
    const C x = spectrum_lin<F>(zk, zmk, w);
    return im ? cmul<F>(x, spectrum_lin<F>(ik, imk, w)) : csqr<F>(x);
  };
  if (k == 0) {
    const C z0 = z[0], i0 = im ? im[0] : z0;
    const C y0 = prod(z0, z0, i0, i0, U[0]);              // X_0
    const C yh = prod(z0, z0, i0, i0, U[h]);              // X_h (W^h = -1)
    z[0] = repack<F>(y0, yh, U[0]);
    if (h >= 2) {   // the self-paired middle slot
      const uint32_t sm = slot_of(gr, h >> 1);
      const C zm = z[sm], imm = im ? im[sm] : zm;
      const C ym = prod(zm, zm, imm, imm, U[h >> 1]);
      z[sm] = repack<F>(ym, ym, U[h >> 1]);
    }
    return;
  }
  if (2 * k >= h) return;   // k = h / 2 was handled with k = 0
  const uint32_t sa = slot_of(gr, k), sb = slot_of(gr, h - k);
  const C za = z[sa], zb = z[sb];
  const C ia = im ? im[sa] : za, ib = im ? im[sb] : zb;
  const C wa = U[k], wb = U[h - k];
  const C ya = prod(za, zb, ia, ib, wa), yb = prod(zb, za, ib, ia, wb);
  z[sa] = repack<F>(ya, yb, wa);
  z[sb] = repack<F>(yb, ya, wb);
}

// dst[j] += src[j]: digit-wise sum of two weakly carried residues (one more bit per digit; the next squaring's carry sweep absorbs it)
__global__ void __launch_bounds__(256) k_add_digits(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, uint32_t n) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < n) dst[j] += src[j];
}

}  // namespace crt
}  // namespace mi355
#include "crt_rows.hpp"
namespace mi355 {
namespace crt {

// ---- small helpers -------------------------------------------------------------------------------------------------------
__global__ void k_set_small(Geom g, uint64_t* __restrict__ x, uint32_t a) {   // x = a (one thread: a touches at most a few digits)
  if (blockIdx.x || threadIdx.x) return;
  uint64_t v = a;
  DigitWalk dw; dw.start(g, 0);
  for (uint32_t j = 0; j < g.n && v; ++j) { const uint32_t w = dw.width(g); x[j] = v & ((uint64_t(1) << w) - 1); v >>= w; dw.next(g); }
}
__global__ void k_sub_small(Geom g, uint64_t* __restrict__ x, uint32_t a) {   // x -= a with borrow and wrap-around (m2:1095-1111)
  if (blockIdx.x || threadIdx.x) return;
  uint64_t borrow = a;
  for (int lap = 0; lap < 3 && borrow; ++lap) {
    DigitWalk dw; dw.start(g, 0);
    for (uint32_t j = 0; j < g.n && borrow; ++j) {
      const uint32_t w = dw.width(g);
      const uint64_t v = x[j];
      if (v >= borrow) { x[j] = v - borrow; borrow = 0; }
      else { const uint64_t need = borrow - v, k = (need + (uint64_t(1) << w) - 1) >> w; x[j] = v + (k << w) - borrow; borrow = k; }
      dw.next(g);
    }
  }
}

}  // namespace crt

// ==========================================================================================================================
// host
// ==========================================================================================================================
namespace {

using crt::F31; using crt::F61; using crt::M31; using crt::M61;

void chk(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string("crt engine: ") + what + ": " + hipGetErrorString(e)); }

uint64_t pow61(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = crt::mul61(r, a); a = crt::mul61(a, a); e >>= 1; } return r; }
uint32_t pow31(uint32_t a, uint64_t e) { uint32_t r = 1; while (e) { if (e & 1) r = crt::mul31(r, a); a = crt::mul31(a, a); e >>= 1; } return r; }

// an element of exact order 2^k in the norm-1 subgroup of Z/p[i] (order p + 1 = 2^61 resp. 2^31): (t + i)^((p - 1) 2^(bits - k)) for the first t
// that gives exact order 2^k; the exponent is applied as (p - 1) first (z^(p-1) = conj(z) / z has norm 1), then by squaring
template <class F>
typename F::C root_2k(unsigned k, unsigned bits) {
  using C = typename F::C;
  for (typename F::S t = 2;; ++t) {
    const C g{t, 1};
    // g^(p-1): p - 1 = 2^bits - 2
    C r{1, 0}, b = g;
    for (unsigned i = 0; i < bits; ++i) { if (i >= 1) r = crt::cmul<F>(r, b); b = crt::cmul<F>(b, b); }   // sum of 2^i, i = 1 .. bits-1 = 2^bits - 2
    for (unsigned i = k; i < bits; ++i) r = crt::cmul<F>(r, r);                                          // ^ 2^(bits - k)
    C z = r;
    for (unsigned i = 1; i < k; ++i) z = crt::cmul<F>(z, z);
    if (z.re == F::M - 1 && z.im == 0) return r;   // r^(2^(k-1)) = -1: exact order 2^k
  }
}

// r^t for the odd t < 8 that makes r^(2^(ln-3)) the wanted 8th root (the four primitive 8th roots are the odd powers of any one)
template <class F>
typename F::C normalise_root(typename F::C r, unsigned ln, typename F::C want) {
  using C = typename F::C;
  C w8 = r;
  for (unsigned i = 3; i < ln; ++i) w8 = crt::cmul<F>(w8, w8);
  C p = w8;
  const C w8sq = crt::cmul<F>(w8, w8);
  for (unsigned t = 1; t < 8; t += 2) {
    if (p.re == want.re && p.im == want.im) {
      C out{1, 0};
      for (unsigned i = 0; i < t; ++i) out = crt::cmul<F>(out, r);
      return out;
    }
    p = crt::cmul<F>(p, w8sq);
  }
  throw std::runtime_error("crt engine: no 8th root of the expected form");
}

template <class S, class POW>
S odd_root(unsigned odd, S modulus, POW pw) {   // a primitive odd-th root of unity among the scalars (odd | p - 1)
  for (S g = 2;; ++g) {
    const S r = pw(g, (uint64_t(modulus) - 1) / odd);
    bool ok = r != 1;
    for (unsigned d = 2; ok && d < odd; ++d) if (odd % d == 0 && pw(r, odd / d) == 1) ok = false;
    if (ok && pw(r, odd) == 1) return r;
  }
}

}  // namespace

size_t crt_transform_size(uint32_t p, uint32_t odd) {   // m2:479-503: smallest odd 2^ln with log2(n) + 2 (p / n + 1) < 92
  for (unsigned ln = 3; ln <= 28; ++ln) {
    const size_t n = size_t(odd) << ln;
    if (n > p) break;
    if (std::log2(double(n)) + 2.0 * (double(p) / double(n) + 1.0) < 92.0) return n;
  }
  return 0;
}

// Automatic choice between the stock power-of-two size and the prime-factor sizes, the reference's policy (README.md:888-926,
// third_party/aevum/src/FFTConfig.cpp:425-520): radix 9 when the stock / PFA size ratio reaches 1.60, else radix 3 when it reaches 1.30,
// else the stock plan.  With the size rule above the candidates below the stock 2^k are 9 2^(k-4) (ratio 1.778) and 3 2^(k-2) (1.333).
// (The reference's exponent boundaries come from its measured bits-per-word tables, fftbpw.h, which admit ~39 bits per word where the
// worst-case rule used here admits ~34: same policy, boundaries of this engine's own capacity rule; tests/test_host_logic.py.)
uint32_t crt_auto_radix(uint32_t p, size_t* words) {
  const size_t stock = crt_transform_size(p, 1), n3 = crt_transform_size(p, 3), n9 = crt_transform_size(p, 9);
  uint32_t odd = 1;
  size_t n = stock;
  if (n9 && stock && double(stock) / double(n9) >= 1.60) { odd = 9; n = n9; }
  else if (n3 && stock && double(stock) / double(n3) >= 1.30) { odd = 3; n = n3; }
  else if (!stock) { if (n9) { odd = 9; n = n9; } else if (n3) { odd = 3; n = n3; } }
  if (words) *words = n;
  return n ? odd : 0;
}

// device-side canonical form for u64 digits in natural order (canon.hip)
size_t canon64_scratch_bytes(uint32_t n);
uint32_t* canon64_flags(uint32_t n, void* scratch);
hipError_t canon64_launch(uint32_t p, uint32_t n, uint32_t odd, const uint64_t* digits, uint64_t* out, void* scratch, hipStream_t s);
hipError_t canon64_compare(const uint64_t* a, const uint64_t* b, uint32_t n, uint32_t* diff_flag, hipStream_t s);
hipError_t canon64_relax(uint32_t p, uint32_t n, uint32_t odd, const uint64_t* in, uint64_t* out, hipStream_t s);
hipError_t canon64_add_complement(uint32_t p, uint32_t n, uint32_t odd, uint64_t* dst, const uint64_t* canon, hipStream_t s);

struct CrtEngine::Impl {
  crt::Geom g;
  crt::Grid gr;
  int device = 0;
  hipStream_t stream = nullptr;
  struct Register {
    uint64_t* x = nullptr;        // [n] digits, logical order, weakly carried
    F61::C* i61 = nullptr;        // packed spectrum of a multiplicand (set_multiplicand), allocated on first use
    F31::C* i31 = nullptr;
    bool image = false;           // holds a multiplicand image instead of digits
  };
  std::vector<Register> regs;
  uint64_t* scratch = nullptr;    // [n] digits: temporary of addsub
  F61::C *Z61 = nullptr, *U61 = nullptr;
  F31::C *Z31 = nullptr, *U31 = nullptr;
  uint64_t *w61 = nullptr, *carry = nullptr, *residual = nullptr;
  uint32_t* w31 = nullptr;
  bool fast = false;              // crt_rows.hpp kernels (rows of 1024, columns of 2 .. 2048)
  bool cols_joint = false;        // MI355_CRT_KERNELS=joint / split: force both fields in one column launch, or one field per launch (A/B runs)
  bool cols_split = false;
  F61::C *W1_61 = nullptr, *W2_61 = nullptr, *V61 = nullptr, *LO61 = nullptr, *HI61 = nullptr;
  F31::C *W1_31 = nullptr, *W2_31 = nullptr, *V31 = nullptr, *LO31 = nullptr, *HI31 = nullptr;
  hipEvent_t ev[kKernels + 1] = {};
  std::vector<uint8_t> width;
  // device-side canonical form (canon.hip, SURVEY.md 8f N4 extended to this family): scratch + two outputs of n digits, allocated on first use
  void* canon = nullptr;
  uint64_t* canon_out[2] = {nullptr, nullptr};
  bool host_carry = false;       // MI355_HOST_CARRY=1: the round-2 host paths (A/B tests)
  // bits by which a register's digits may exceed their widths: 0 after a carry sweep, +1 per digit-wise addition; a transform needs
  // log2(n) + 2 (w + excess) < 92 and relaxes the register first (one local carry pass) when that fails
  std::vector<int> excess;
};

const char* CrtEngine::kernel_name(size_t k) {
  static const char* names[kKernels] = {"k_front", "k_rows_fwd", "k_pointwise", "k_rows_inv", "k_back", "k_crt_carry"};
  return k < kKernels ? names[k] : "";
}

CrtEngine::CrtEngine(uint32_t p, size_t reg_count, uint32_t odd, size_t n_forced, int device, const char* spec) : im_(new Impl) {
  Impl& im = *im_;
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device available: the MI355X engine has no CPU fallback");
    if (odd != 1 && odd != 3 && odd != 9) throw std::runtime_error("crt engine: odd radix must be 1, 3 or 9");
    if (reg_count == 0 || reg_count > 64) throw std::runtime_error("crt engine: register count must be 1 .. 64");
    const size_t n = n_forced ? n_forced : crt_transform_size(p, odd);
    if (!n) throw std::runtime_error("crt engine: no admissible transform size for this exponent");
    if (std::log2(double(n)) + 2.0 * (double(p) / double(n) + 1.0) >= 92.0) throw std::runtime_error("crt engine: transform too small for this exponent");
    im.g = crt::make_geom(p, n, odd, 1);
    // the carry sweep hands a run's carry to the next run and lets it die inside that run's kRun digits: (kRun - 1) words must hold a
    // coefficient of up to 92 bits (every size the reference's rule picks has more than 20 bits per word)
    if (uint64_t(im.g.q) * (crt::kRun - 1) < 100) throw std::runtime_error("crt engine: fewer than 15 bits per word at this transform size");
    crt::Grid& gr = im.gr;
    gr.odd = odd; gr.ln = im.g.ln; gr.m = 1u << gr.ln; gr.h = gr.m >> 1; gr.logh = gr.ln - 1;
    if (gr.ln < 3) throw std::runtime_error("crt engine: power-of-two axis too short");
    uint32_t logH2 = std::min<uint32_t>(10, gr.logh);
    if (spec && std::strncmp(spec, "h2=", 3) == 0) logH2 = uint32_t(std::atoi(spec + 3));
    if (logH2 < 1 || logH2 > std::min<uint32_t>(10, gr.logh) || gr.logh - logH2 > 11) throw std::runtime_error("crt engine: bad row split");
    gr.logH2 = logH2; gr.logH1 = gr.logh - logH2;
    gr.minv = 0;
    if (odd > 1) for (uint32_t y = 1; y < odd; ++y) if ((uint64_t(gr.m % odd) * y) % odd == 1) gr.minv = y;
    const uint64_t r61 = odd > 1 ? odd_root<uint64_t>(odd, M61, pow61) : 1;
    const uint32_t r31 = odd > 1 ? odd_root<uint32_t>(odd, M31, pow31) : 1;
    for (unsigned k = 0; k < 9; ++k) {
      gr.r61[k] = pow61(r61, k % odd); gr.r61i[k] = pow61(r61, (odd - k % odd) % odd);
      gr.r31[k] = pow31(r31, k % odd); gr.r31i[k] = pow31(r31, (odd - k % odd) % odd);
    }
    {
      const unsigned cube = odd == 9 ? 3 : 1;            // w3 = r^3 for radix 9, r itself for radix 3
      const uint64_t w61 = gr.r61[cube % 9], w61sq = crt::mul61(w61, w61);
      const uint32_t w31 = gr.r31[cube % 9], w31sq = crt::mul31(w31, w31);
      gr.c3_61 = odd > 1 ? F61::half(F61::sub(w61, w61sq)) : 0; gr.c3_31 = odd > 1 ? F31::half(F31::sub(w31, w31sq)) : 0;
      gr.mm = gr.m % odd;
      gr.pm = uint32_t((uint64_t(p) * gr.m) % n);
      gr.lpm61 = uint32_t(uint64_t(im.g.l61) * (gr.pm % 61) % 61); gr.lpm31 = uint32_t(uint64_t(im.g.l31) * (gr.pm % 31) % 31);
    }
    { const char* tn = std::getenv("MI355_CRT_TUNE"); gr.tune = tn ? uint32_t(std::atoi(tn)) : 0u; }
    gr.s61 = pow61((uint64_t(odd) * gr.h) % M61, M61 - 2); gr.s31 = pow31(uint32_t((uint64_t(odd) * gr.h) % M31), M31 - 2);

    im.device = device;
    chk(hipSetDevice(device), "hipSetDevice");
    chk(hipStreamCreateWithFlags(&im.stream, hipStreamNonBlocking), "stream");
    for (auto& e : im.ev) chk(hipEventCreate(&e), "event");
    const size_t h = gr.h, nruns = (n + crt::kRun - 1) / crt::kRun;
    im.regs.resize(reg_count);
    chk(hipMalloc(reinterpret_cast<void**>(&im.scratch), n * 8), "hipMalloc");   // addsub's temporary
    for (auto& r : im.regs) { chk(hipMalloc(reinterpret_cast<void**>(&r.x), n * 8), "hipMalloc"); chk(hipMemset(r.x, 0, n * 8), "memset"); }
    chk(hipMalloc(reinterpret_cast<void**>(&im.Z61), size_t(odd) * h * 16), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&im.Z31), size_t(odd) * h * 8), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&im.U61), (h + 1) * 16), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&im.U31), (h + 1) * 8), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&im.w61), n * 8), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&im.w31), n * 4), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&im.carry), nruns * 16), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&im.residual), nruns * 8), "hipMalloc");
    // omega_m^k, k <= h
    {
      std::vector<F61::C> u61(h + 1); std::vector<F31::C> u31(h + 1);
      // the generator is rotated (an odd power keeps its order) so that omega_m^(m/8) is (1 + i) / sqrt 2 = (1 + i) 2^30 resp. (1 + i) 2^15:
      // the radix-8 steps of crt_rows.hpp multiply by that root with an add, a sub and two bit rotations
      const F61::C w61 = normalise_root<F61>(root_2k<F61>(gr.ln, 61), gr.ln, F61::C{uint64_t(1) << 30, uint64_t(1) << 30});
      const F31::C w31 = normalise_root<F31>(root_2k<F31>(gr.ln, 31), gr.ln, F31::C{1u << 15, 1u << 15});
      F61::C a{1, 0}; F31::C b{1, 0};
      for (size_t k = 0; k <= h; ++k) { u61[k] = a; u31[k] = b; a = crt::cmul<F61>(a, w61); b = crt::cmul<F31>(b, w31); }
      chk(hipMemcpy(im.U61, u61.data(), (h + 1) * 16, hipMemcpyHostToDevice), "copy"); chk(hipMemcpy(im.U31, u31.data(), (h + 1) * 8, hipMemcpyHostToDevice), "copy");
      const char* ks = std::getenv("MI355_CRT_KERNELS");
      im.fast = gr.logH2 == 10 && gr.logH1 >= 1 && gr.logH1 <= 11 && !(ks && std::strcmp(ks, "generic") == 0);
      im.cols_joint = ks && std::strcmp(ks, "joint") == 0;
      im.cols_split = ks && std::strcmp(ks, "split") == 0;
      if (im.fast) {   // the one-field column kernels use 68 KiB of LDS
        const int l61 = int((4096 + 4096 / 16) * 16), l31 = int((8192 + 8192 / 16) * 8);
        chk(hipFuncSetAttribute(reinterpret_cast<const void*>(&crt::k_cols_one<F61, false, 4096>), hipFuncAttributeMaxDynamicSharedMemorySize, l61), "lds attribute");
        chk(hipFuncSetAttribute(reinterpret_cast<const void*>(&crt::k_cols_one<F61, true, 4096>), hipFuncAttributeMaxDynamicSharedMemorySize, l61), "lds attribute");
        chk(hipFuncSetAttribute(reinterpret_cast<const void*>(&crt::k_cols_one<F31, false, 8192>), hipFuncAttributeMaxDynamicSharedMemorySize, l31), "lds attribute");
        chk(hipFuncSetAttribute(reinterpret_cast<const void*>(&crt::k_cols_one<F31, true, 8192>), hipFuncAttributeMaxDynamicSharedMemorySize, l31), "lds attribute");
      }
      if (im.fast) {
        // omega_L^x = omega_m^(x m / L) for the two pass lengths (x < L; beyond h through omega_m^h = -1), omega_m^(H1 k2)
        auto pick61 = [&](size_t e) { return e <= h ? u61[e] : crt::cneg<F61>(u61[e - h]); };
        auto pick31 = [&](size_t e) { return e <= h ? u31[e] : crt::cneg<F31>(u31[e - h]); };
        const size_t H1 = size_t(1) << gr.logH1, H2 = size_t(1) << gr.logH2, m = size_t(gr.m);
        std::vector<F61::C> t61; std::vector<F31::C> t31;
        auto upload = [&](size_t count, size_t stride, F61::C*& d61, F31::C*& d31) {
          t61.resize(count); t31.resize(count);
          for (size_t x = 0; x < count; ++x) { t61[x] = pick61(x * stride); t31[x] = pick31(x * stride); }
          chk(hipMalloc(reinterpret_cast<void**>(&d61), count * 16), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&d31), count * 8), "hipMalloc");
          chk(hipMemcpy(d61, t61.data(), count * 16, hipMemcpyHostToDevice), "copy"); chk(hipMemcpy(d31, t31.data(), count * 8, hipMemcpyHostToDevice), "copy");
        };
        upload(H1, m / H1, im.W1_61, im.W1_31);
        upload(H2, m / H2, im.W2_61, im.W2_31);
        upload(H2, H1, im.V61, im.V31);
        upload(1024, 1, im.LO61, im.LO31);                 // m >= 2^12 on this path
        upload(m >> 10, 1024, im.HI61, im.HI31);
      }
    }
    im.width.resize(n);
    uint64_t prev = 0;
    for (size_t j = 0; j < n; ++j) { const uint64_t next = (uint64_t(p) * (j + 1) + n - 1) / n; im.width[j] = uint8_t(next - prev); prev = next; }
    im.excess.assign(reg_count, 0);
    { const char* hc = std::getenv("MI355_HOST_CARRY"); im.host_carry = hc && hc[0] == '1'; }
  } catch (...) {
    release();
    throw;
  }
}

void CrtEngine::release() {
  if (!im_) return;
  Impl& im = *im_;
  (void)hipSetDevice(im.device);
  if (im.stream) (void)hipStreamSynchronize(im.stream);
  for (auto& r : im.regs) for (void* q : {static_cast<void*>(r.x), static_cast<void*>(r.i61), static_cast<void*>(r.i31)}) if (q) (void)hipFree(q);
  if (im.scratch) (void)hipFree(im.scratch);
  if (im.canon) (void)hipFree(im.canon);
  for (void* q : {static_cast<void*>(im.Z61), static_cast<void*>(im.Z31), static_cast<void*>(im.U61), static_cast<void*>(im.U31),
                  static_cast<void*>(im.w61), static_cast<void*>(im.w31), static_cast<void*>(im.carry), static_cast<void*>(im.residual),
                  static_cast<void*>(im.W1_61), static_cast<void*>(im.W2_61), static_cast<void*>(im.V61), static_cast<void*>(im.W1_31),
                  static_cast<void*>(im.W2_31), static_cast<void*>(im.V31), static_cast<void*>(im.LO61), static_cast<void*>(im.HI61),
                  static_cast<void*>(im.LO31), static_cast<void*>(im.HI31)})
    if (q) (void)hipFree(q);
  for (auto& e : im.ev) if (e) (void)hipEventDestroy(e);
  if (im.stream) (void)hipStreamDestroy(im.stream);
  delete im_;
  im_ = nullptr;
}
CrtEngine::~CrtEngine() { release(); }

size_t CrtEngine::size() const { return im_->g.n; }
uint32_t CrtEngine::odd() const { return im_->g.odd; }
uint32_t CrtEngine::exponent() const { return im_->g.p; }
std::string CrtEngine::describe() const {
  const crt::Grid& gr = im_->gr;
  return "crt-hip:n=" + std::to_string(im_->g.n) + ":odd=" + std::to_string(gr.odd) + ":m=" + std::to_string(gr.m) + ":h1=" + std::to_string(1u << gr.logH1) + ":h2=" +
         std::to_string(1u << gr.logH2) + (im_->fast ? ":radix8" : ":generic");
}
size_t CrtEngine::algorithmic_bytes() const {   // digits r + w, 4 row passes r + w, and (two-kernel form only) the carry sweep's input w + r
  const crt::Grid& gr = im_->gr;
  const bool fused = gr.odd > 1 && (gr.h & 255u) == 0 && !(gr.tune & 2u);   // k_back_carry (launch_transform)
  return size_t(im_->g.n) * (8 + 8 + 8 * 12 + (fused ? 0 : 2 * 12));
}

void CrtEngine::sync() {
  chk(hipSetDevice(im_->device), "hipSetDevice");
  chk(hipStreamSynchronize(im_->stream), "sync");
  chk(hipGetLastError(), "kernel");
}

template <class F>
static void launch_rows(const crt::Grid& gr, typename F::C* Z, const typename F::C* U, bool inverse, hipStream_t s) {
  auto pass = [&](uint32_t logL, int cols) {
    const uint32_t L = 1u << logL;
    uint32_t CA = std::max<uint32_t>(1, crt::kPassElems / L);
    const uint32_t per_row = gr.h >> logL;
    CA = std::min(CA, per_row);                       // powers of two: CA divides per_row
    const uint32_t groups = gr.odd * per_row / CA;
    hipLaunchKernelGGL((crt::k_pass<F>), dim3(groups), dim3(256), 0, s, gr, Z, U, logL, CA, cols, inverse ? 1 : 0);
  };
  if (!inverse) {
    if (gr.logH1) pass(gr.logH1, 1);
    pass(gr.logH2, 0);
  } else {
    pass(gr.logH2, 0);
    if (gr.logH1) pass(gr.logH1, 1);
  }
}

// forward transform of register `reg` into the work arrays Z (front + forward columns; the rows are part of the next stage), then
//   mode 0: square, inverse, carry sweep back into `reg` (x a)
//   mode 1: rows forward only -> the packed spectrum becomes the image of register `dst` (set_multiplicand)
//   mode 2: multiply by the image of register `src`, inverse, carry sweep back into `reg` (x a)
void CrtEngine::launch_transform(size_t reg, int mode, size_t other, uint32_t a, bool timed) {
  Impl& im = *im_;
  const crt::Grid& gr = im.gr;
  crt::Geom g = im.g; g.a = a;
  hipStream_t s = im.stream;
  const dim3 b256(256), gslots((gr.h + 255) / 256);
  uint64_t* x = im.regs[reg].x;
  F61::C* i61 = mode ? im.regs[other].i61 : nullptr; F31::C* i31 = mode ? im.regs[other].i31 : nullptr;
  int e = 0;
  auto mark = [&] { if (timed) chk(hipEventRecord(im.ev[e++], s), "event"); };
  mark();
  switch (gr.odd) {
    case 1: hipLaunchKernelGGL((crt::k_front<1>), gslots, b256, 0, s, g, gr, x, im.Z61, im.Z31); break;
    case 3: hipLaunchKernelGGL((crt::k_front<3>), gslots, b256, 0, s, g, gr, x, im.Z61, im.Z31); break;
    default: hipLaunchKernelGGL((crt::k_front<9>), gslots, b256, 0, s, g, gr, x, im.Z61, im.Z31); break;
  }
  mark();
  if (im.fast) {
    const crt::FastTables T{im.W1_61, im.W2_61, im.V61, im.U61, im.LO61, im.HI61, im.W1_31, im.W2_31, im.V31, im.U31, im.LO31, im.HI31};
    const uint32_t CA = crt::kFastSlots >> gr.logH1, gcols = gr.odd * ((1u << gr.logH2) / CA), gmid = gr.odd * (1u << gr.logH1) / 2;
    // columns: one field per launch where that gives wider row segments (H2 columns must hold at least one group of each kind)
    constexpr uint32_t S61 = 4096, S31 = 8192;
    // measured: at H1 = 512 the joint kernel is faster (0.121 / 0.114 ms against 0.150 / 0.132), from H1 = 1024 on the split ones are
    const bool split = !im.cols_joint && (gr.logH1 >= 10 || im.cols_split) && (S31 >> gr.logH1) >= 1 && (S31 >> gr.logH1) <= (1u << gr.logH2);
    const uint32_t g61 = gr.odd * (1u << gr.logH2) / std::max(1u, S61 >> gr.logH1), g31 = gr.odd * (1u << gr.logH2) / std::max(1u, S31 >> gr.logH1);
    const size_t lds61 = size_t(S61 + S61 / 16) * 16, lds31 = size_t(S31 + S31 / 16) * 8;
    if (split) {
      hipLaunchKernelGGL((crt::k_cols_one<F61, false, S61>), dim3(g61), dim3(S61 / 8), lds61, s, gr, im.W1_61, im.LO61, im.HI61, im.Z61);
      hipLaunchKernelGGL((crt::k_cols_one<F31, false, S31>), dim3(g31), dim3(S31 / 8), lds31, s, gr, im.W1_31, im.LO31, im.HI31, im.Z31);
    } else {
      hipLaunchKernelGGL((crt::k_cols_fast<false>), dim3(gcols), b256, crt::kFastLdsBytes, s, gr, T, im.Z61, im.Z31);
    }
    mark();
    if (mode == 0) hipLaunchKernelGGL((crt::k_mid_fast<0>), dim3(gmid), b256, crt::kFastLdsBytes, s, gr, T, im.Z61, im.Z31, i61, i31);
    else if (mode == 1) { hipLaunchKernelGGL((crt::k_mid_fast<1>), dim3(gmid), b256, crt::kFastLdsBytes, s, gr, T, im.Z61, im.Z31, i61, i31); return; }
    else hipLaunchKernelGGL((crt::k_mid_fast<2>), dim3(gmid), b256, crt::kFastLdsBytes, s, gr, T, im.Z61, im.Z31, i61, i31);
    mark();
    if (split) {
      hipLaunchKernelGGL((crt::k_cols_one<F61, true, S61>), dim3(g61), dim3(S61 / 8), lds61, s, gr, im.W1_61, im.LO61, im.HI61, im.Z61);
      hipLaunchKernelGGL((crt::k_cols_one<F31, true, S31>), dim3(g31), dim3(S31 / 8), lds31, s, gr, im.W1_31, im.LO31, im.HI31, im.Z31);
    } else {
      hipLaunchKernelGGL((crt::k_cols_fast<true>), dim3(gcols), b256, crt::kFastLdsBytes, s, gr, T, im.Z61, im.Z31);
    }
    mark();   // slots: k_rows_fwd = forward columns, k_pointwise = the fused row kernel, k_rows_inv = inverse columns
  } else {
    launch_rows<F61>(gr, im.Z61, im.U61, false, s);
    launch_rows<F31>(gr, im.Z31, im.U31, false, s);
    mark();
    if (mode == 1) {
      chk(hipMemcpyAsync(i61, im.Z61, size_t(gr.odd) * gr.h * 16, hipMemcpyDeviceToDevice, s), "copy");
      chk(hipMemcpyAsync(i31, im.Z31, size_t(gr.odd) * gr.h * 8, hipMemcpyDeviceToDevice, s), "copy");
      return;
    }
    const uint32_t pw = ((gr.h >> 1) + 1) * gr.odd;
    hipLaunchKernelGGL((crt::k_pointwise<F61>), dim3((pw + 255) / 256), b256, 0, s, gr, im.Z61, im.U61, static_cast<const F61::C*>(i61));
    hipLaunchKernelGGL((crt::k_pointwise<F31>), dim3((pw + 255) / 256), b256, 0, s, gr, im.Z31, im.U31, static_cast<const F31::C*>(i31));
    mark();
    launch_rows<F61>(gr, im.Z61, im.U61, true, s);
    launch_rows<F31>(gr, im.Z31, im.U31, true, s);
    mark();
  }
  // back + carry fused (k_back_carry) wherever a work-group has whole ranges of 512 digits; MI355_CRT_TUNE bit 1: the two-kernel form
  if (gr.odd > 1 && (gr.h & 255u) == 0 && !(gr.tune & 2u)) {
    const dim3 ggroups(gr.h >> 8), gedges((uint32_t(gr.h >> 8) * gr.odd + 255) / 256);
    if (gr.odd == 3) hipLaunchKernelGGL((crt::k_back_carry<3>), ggroups, b256, 0, s, g, gr, im.Z61, im.Z31, x, im.carry);
    else hipLaunchKernelGGL((crt::k_back_carry<9>), ggroups, b256, 0, s, g, gr, im.Z61, im.Z31, x, im.carry);
    mark();   // slot k_back: the fused kernel; slot k_crt_carry: the range edges
    hipLaunchKernelGGL(crt::k_crt_range_edges, gedges, b256, 0, s, g, gr, x, im.carry);
    mark();
    return;
  }
  switch (gr.odd) {
    case 1: hipLaunchKernelGGL((crt::k_back<1>), gslots, b256, 0, s, gr, im.Z61, im.Z31, im.w61, im.w31); break;
    case 3: hipLaunchKernelGGL((crt::k_back<3>), gslots, b256, 0, s, gr, im.Z61, im.Z31, im.w61, im.w31); break;
    default: hipLaunchKernelGGL((crt::k_back<9>), gslots, b256, 0, s, gr, im.Z61, im.Z31, im.w61, im.w31); break;
  }
  mark();
  crt::crt_carry_launch_linked(g, im.w61, im.w31, x, im.carry, s);   // (im.carry: 16 bytes per run allocated, 3 words per 256 runs used)
  mark();
}

size_t CrtEngine::reg_count() const { return im_->regs.size(); }

void CrtEngine::check_digits(size_t reg, const char* what) const {
  if (reg >= im_->regs.size()) throw std::runtime_error(std::string(what) + ": register index out of range");
  if (im_->regs[reg].image) throw std::runtime_error(std::string(what) + ": register holds a multiplicand image, not a residue");
}

// Headroom of the transform: the convolution sums stay below M61 M31 ~ 2^92 while log2(n) + 2 (w + excess) < 92 (the size rule is that
// bound at excess 0).  A register that has been through digit-wise additions is relaxed first when it would not fit: one local carry
// pass (canon.hip k_local) brings digits of w + e bits below 2^w + 2^e.
void CrtEngine::ensure_headroom(size_t reg) {
  Impl& im = *im_;
  const int e = im.excess[reg];
  if (e == 0) return;
  const double bits = std::log2(double(im.g.n)) + 2.0 * (double(im.g.q) + 1.0 + double(e));
  if (bits < 92.0 && e < 8) return;
  chk(canon64_relax(im.g.p, im.g.n, im.gr.odd, im.regs[reg].x, im.scratch, im.stream), "relax");
  std::swap(im.regs[reg].x, im.scratch);
  im.excess[reg] = 0;
}

// canonical digits of `reg` on the device (strong carry with wrap-around, 2^p - 1 -> 0 and flag [0]); slot 0 / 1
uint64_t* CrtEngine::canon_digits(size_t reg, int slot) {
  Impl& im = *im_;
  chk(hipSetDevice(im.device), "hipSetDevice");
  if (!im.canon) {
    const size_t sb = (canon64_scratch_bytes(im.g.n) + 255) & ~size_t(255);
    chk(hipMalloc(&im.canon, sb + 2 * size_t(im.g.n) * 8), "hipMalloc");
    im.canon_out[0] = reinterpret_cast<uint64_t*>(static_cast<unsigned char*>(im.canon) + sb);
    im.canon_out[1] = im.canon_out[0] + im.g.n;
    chk(hipMemsetAsync(canon64_flags(im.g.n, im.canon), 0, 16 * 4, im.stream), "memset");
  }
  chk(canon64_launch(im.g.p, im.g.n, im.gr.odd, im.regs[reg].x, im.canon_out[slot], im.canon, im.stream), "canon");
  return im.canon_out[slot];
}
// flags: [0] all ones, [1] a digit too wide for the 0/1 chain (fall back to the host carry), [2] compare differs; cleared for the next use
bool CrtEngine::canon_flags_ok(uint32_t (&flags)[4]) {
  Impl& im = *im_;
  uint32_t* df = canon64_flags(im.g.n, im.canon);
  chk(hipMemcpyAsync(flags, df, 16, hipMemcpyDeviceToHost, im.stream), "copy");
  chk(hipMemsetAsync(df, 0, 16 * 4, im.stream), "memset");
  chk(hipStreamSynchronize(im.stream), "sync");
  return flags[1] == 0;
}

void CrtEngine::square_mul(size_t reg, uint32_t a) {
  check_digits(reg, "square_mul");
  if (a == 0) throw std::runtime_error("square_mul: factor must be >= 1");
  chk(hipSetDevice(im_->device), "hipSetDevice");
  ensure_headroom(reg);
  launch_transform(reg, 0, 0, a, false);
  im_->excess[reg] = 0;
}

// dst <- the transformed image of src (engine::set_multiplicand, engine.h:53); dst may be src
void CrtEngine::set_multiplicand(size_t dst, size_t src) {
  Impl& im = *im_;
  check_digits(src, "set_multiplicand");
  if (dst >= im.regs.size()) throw std::runtime_error("set_multiplicand: register index out of range");
  chk(hipSetDevice(im.device), "hipSetDevice");
  Impl::Register& d = im.regs[dst];
  if (!d.i61) {
    chk(hipMalloc(reinterpret_cast<void**>(&d.i61), size_t(im.gr.odd) * im.gr.h * 16), "hipMalloc");
    chk(hipMalloc(reinterpret_cast<void**>(&d.i31), size_t(im.gr.odd) * im.gr.h * 8), "hipMalloc");
  }
  ensure_headroom(src);
  launch_transform(src, 1, dst, 1, false);
  d.image = true;
}

// dst <- dst * src * a with src a multiplicand image (engine::mul, engine.h:60)
void CrtEngine::mul(size_t dst, size_t src, uint32_t a) {
  Impl& im = *im_;
  check_digits(dst, "mul");
  if (src >= im.regs.size() || !im.regs[src].image) throw std::runtime_error("mul: the source register is not a multiplicand (call set_multiplicand first)");
  if (a == 0) throw std::runtime_error("mul: factor must be >= 1");
  chk(hipSetDevice(im.device), "hipSetDevice");
  ensure_headroom(dst);
  launch_transform(dst, 2, src, a, false);
  im.excess[dst] = 0;
}

void CrtEngine::copy(size_t dst, size_t src) {
  Impl& im = *im_;
  if (dst >= im.regs.size() || src >= im.regs.size()) throw std::runtime_error("copy: register index out of range");
  if (dst == src) return;
  chk(hipSetDevice(im.device), "hipSetDevice");
  Impl::Register& d = im.regs[dst]; const Impl::Register& r = im.regs[src];
  if (r.image) {
    if (!d.i61) {
      chk(hipMalloc(reinterpret_cast<void**>(&d.i61), size_t(im.gr.odd) * im.gr.h * 16), "hipMalloc");
      chk(hipMalloc(reinterpret_cast<void**>(&d.i31), size_t(im.gr.odd) * im.gr.h * 8), "hipMalloc");
    }
    chk(hipMemcpyAsync(d.i61, r.i61, size_t(im.gr.odd) * im.gr.h * 16, hipMemcpyDeviceToDevice, im.stream), "copy");
    chk(hipMemcpyAsync(d.i31, r.i31, size_t(im.gr.odd) * im.gr.h * 8, hipMemcpyDeviceToDevice, im.stream), "copy");
  } else {
    chk(hipMemcpyAsync(d.x, r.x, size_t(im.g.n) * 8, hipMemcpyDeviceToDevice, im.stream), "copy");
    im.excess[dst] = im.excess[src];
  }
  d.image = r.image;
}

// dst <- dst + src, digit-wise on weakly carried digits (engine::add, engine.h:64)
void CrtEngine::add(size_t dst, size_t src) {
  Impl& im = *im_;
  check_digits(dst, "add"); check_digits(src, "add");
  chk(hipSetDevice(im.device), "hipSetDevice");
  hipLaunchKernelGGL(crt::k_add_digits, dim3((im.g.n + 255) / 256), dim3(256), 0, im.stream, im.regs[dst].x, im.regs[src].x, im.g.n);
  im.excess[dst] = std::max(im.excess[dst], im.excess[src]) + 1;
  if (im.excess[dst] >= 8) ensure_headroom(dst);   // repeated additions without a transform in between
}

// the canonical digits of `src` on the device (slot 1), through the host when the device chain reports a digit too wide for it
const uint64_t* CrtEngine::canonical_on_device(size_t src) {
  Impl& im = *im_;
  const uint64_t* c = canon_digits(src, 1);
  uint32_t flags[4];
  if (canon_flags_ok(flags) && !im.host_carry) return c;
  std::vector<uint64_t> d(im.g.n);
  get_digits_host(src, d.data(), im.g.n);
  chk(hipMemcpy(im.canon_out[1], d.data(), size_t(im.g.n) * 8, hipMemcpyHostToDevice), "copy");
  return im.canon_out[1];
}

// dst <- dst - src = dst + (2^p - 1 - src): the digit-wise complement of the canonical form of src, taken on the device
void CrtEngine::sub_reg(size_t dst, size_t src) {
  Impl& im = *im_;
  check_digits(dst, "sub_reg"); check_digits(src, "sub_reg");
  const uint64_t* c = canonical_on_device(src);
  chk(canon64_add_complement(im.g.p, im.g.n, im.gr.odd, im.regs[dst].x, c, im.stream), "sub_reg");
  im.excess[dst] = im.excess[dst] + 1;
  if (im.excess[dst] >= 8) ensure_headroom(dst);
}

// sum -> sum_out (and sum_copy), difference -> diff_out (and diff_copy); -1: not wanted.  a and b may be among the outputs.
void CrtEngine::addsub(long sum_out, long sum_copy, long diff_out, long diff_copy, size_t a, size_t b) {
  Impl& im = *im_;
  check_digits(a, "addsub"); check_digits(b, "addsub");
  const long outs[4] = {sum_out, sum_copy, diff_out, diff_copy};
  for (int i = 0; i < 4; ++i) {
    if (outs[i] >= long(im.regs.size())) throw std::runtime_error("addsub: register index out of range");
    for (int j = 0; j < i; ++j) if (outs[i] >= 0 && outs[i] == outs[j]) throw std::runtime_error("addsub: output registers must differ");
  }
  if ((sum_copy >= 0 && sum_out < 0) || (diff_copy >= 0 && diff_out < 0)) throw std::runtime_error("addsub: copy output without a primary output");
  chk(hipSetDevice(im.device), "hipSetDevice");
  const size_t n = im.g.n, bytes = n * 8;
  const dim3 grid((im.g.n + 255) / 256), block(256);
  // scratch = a + (2^p - 1 - b): the complement of the canonical digits of b (taken on the device, as in sub_reg), before anything is overwritten
  const int ea = im.excess[a], eb = im.excess[b];
  if (diff_out >= 0) {
    const uint64_t* c = canonical_on_device(b);
    chk(hipMemcpyAsync(im.scratch, im.regs[a].x, bytes, hipMemcpyDeviceToDevice, im.stream), "copy");
    chk(canon64_add_complement(im.g.p, im.g.n, im.gr.odd, im.scratch, c, im.stream), "addsub");
  }
  if (sum_out >= 0) {
    if (size_t(sum_out) == b) {   // b + a
      hipLaunchKernelGGL(crt::k_add_digits, grid, block, 0, im.stream, im.regs[b].x, im.regs[a].x, im.g.n);
    } else {
      if (size_t(sum_out) != a) chk(hipMemcpyAsync(im.regs[sum_out].x, im.regs[a].x, bytes, hipMemcpyDeviceToDevice, im.stream), "copy");
      hipLaunchKernelGGL(crt::k_add_digits, grid, block, 0, im.stream, im.regs[sum_out].x, im.regs[b].x, im.g.n);
    }
    im.regs[sum_out].image = false; im.excess[sum_out] = std::max(ea, eb) + 1;
    if (sum_copy >= 0) { chk(hipMemcpyAsync(im.regs[sum_copy].x, im.regs[sum_out].x, bytes, hipMemcpyDeviceToDevice, im.stream), "copy"); im.regs[sum_copy].image = false; im.excess[sum_copy] = im.excess[sum_out]; }
  }
  if (diff_out >= 0) {
    chk(hipMemcpyAsync(im.regs[diff_out].x, im.scratch, bytes, hipMemcpyDeviceToDevice, im.stream), "copy");
    im.regs[diff_out].image = false; im.excess[diff_out] = ea + 1;
    if (diff_copy >= 0) { chk(hipMemcpyAsync(im.regs[diff_copy].x, im.scratch, bytes, hipMemcpyDeviceToDevice, im.stream), "copy"); im.regs[diff_copy].image = false; im.excess[diff_copy] = ea + 1; }
  }
}
void CrtEngine::mul_add(size_t dst, size_t mul_src, size_t add_src, uint32_t f) { mul(dst, mul_src, f); add(dst, add_src); }
void CrtEngine::square_mul_copy(size_t src, size_t dst_copy, uint32_t f) { square_mul(src, f); copy(dst_copy, src); }
void CrtEngine::mul_copy(size_t dst, size_t src, size_t dst_copy, uint32_t f) { mul(dst, src, f); copy(dst_copy, dst); }

void CrtEngine::set_u32(size_t reg, uint32_t a) {
  Impl& im = *im_;
  if (reg >= im.regs.size()) throw std::runtime_error("set: register index out of range");
  chk(hipSetDevice(im.device), "hipSetDevice");
  chk(hipMemsetAsync(im.regs[reg].x, 0, size_t(im.g.n) * 8, im.stream), "memset");
  if (a) hipLaunchKernelGGL(crt::k_set_small, dim3(1), dim3(1), 0, im.stream, im.g, im.regs[reg].x, a);
  im.regs[reg].image = false; im.excess[reg] = 0;
}
void CrtEngine::sub_u32(size_t reg, uint32_t a) {
  Impl& im = *im_;
  check_digits(reg, "sub");
  chk(hipSetDevice(im.device), "hipSetDevice");
  if (a) hipLaunchKernelGGL(crt::k_sub_small, dim3(1), dim3(1), 0, im.stream, im.g, im.regs[reg].x, a);
}

void CrtEngine::set_digits(size_t reg, const uint64_t* d, size_t count) {
  Impl& im = *im_;
  if (reg >= im.regs.size()) throw std::runtime_error("set_digits: register index out of range");
  if (count != im.g.n) throw std::runtime_error("set_digits: wrong digit count");
  for (size_t j = 0; j < count; ++j) if (d[j] >> 62) throw std::runtime_error("set_digits: digit out of range");
  chk(hipSetDevice(im.device), "hipSetDevice");
  chk(hipStreamSynchronize(im.stream), "sync");
  chk(hipMemcpy(im.regs[reg].x, d, count * 8, hipMemcpyHostToDevice), "copy");
  im.regs[reg].image = false;
  int e = 0;   // callers may hand over digits wider than their slots (weakly carried vectors)
  for (size_t j = 0; j < count; ++j) { const int bits = d[j] ? 64 - __builtin_clzll(d[j]) : 0; e = std::max(e, bits - int(im.width[j])); }
  im.excess[reg] = e;
}

// digits as they are on the device (weakly carried) or canonical: strong carry with wrap-around, 2^p - 1 stays all ones.
// The canonical form is made on the device (canon.hip); MI355_HOST_CARRY=1 or a device chain that reports an over-wide digit use the
// host loop below (the reference's way: engine_gpu.h:1534-1561).
void CrtEngine::get_digits(size_t reg, uint64_t* d, size_t count, bool canonical) {
  Impl& im = *im_;
  check_digits(reg, "get_digits");
  if (count != im.g.n) throw std::runtime_error("get_digits: wrong digit count");
  if (!canonical) {
    sync();
    chk(hipMemcpy(d, im.regs[reg].x, count * 8, hipMemcpyDeviceToHost), "copy");
    return;
  }
  if (!im.host_carry) {
    const uint64_t* c = canon_digits(reg, 0);
    chk(hipMemcpyAsync(d, c, count * 8, hipMemcpyDeviceToHost, im.stream), "copy");
    uint32_t flags[4];
    if (canon_flags_ok(flags)) {
      if (flags[0]) for (size_t j = 0; j < count; ++j) d[j] = (uint64_t(1) << im.width[j]) - 1;
      return;
    }
  }
  get_digits_host(reg, d, count);
}
void CrtEngine::get_digits_host(size_t reg, uint64_t* d, size_t count) {
  Impl& im = *im_;
  sync();
  chk(hipMemcpy(d, im.regs[reg].x, count * 8, hipMemcpyDeviceToHost), "copy");
  uint64_t carry = 0;
  for (int lap = 0; lap < 4; ++lap) {
    for (size_t j = 0; j < count; ++j) {
      const uint64_t v = d[j] + carry;
      d[j] = v & ((uint64_t(1) << im.width[j]) - 1);
      carry = v >> im.width[j];
      if (lap && !carry) break;
    }
    if (!carry) break;
  }
}

void CrtEngine::get_digits_encoded(size_t reg, uint64_t* d, size_t count) {
  Impl& im = *im_;
  get_digits(reg, d, count, true);
  for (size_t j = 0; j < count; ++j) {
    if (im.width[j] > 32) throw std::runtime_error("get_digits: this transform size has words of more than 32 bits, which the value | width << 32 encoding cannot hold (use get_words)");
    d[j] |= uint64_t(im.width[j]) << 32;
  }
}
void CrtEngine::set_digits_encoded(size_t reg, const uint64_t* d, size_t count) {
  Impl& im = *im_;
  if (count != im.g.n) throw std::runtime_error("set_digits: wrong digit count");
  std::vector<uint64_t> v(count);
  for (size_t j = 0; j < count; ++j) {
    if ((d[j] >> 32) != im.width[j]) throw std::runtime_error("set_digits: digit width mismatch");
    v[j] = d[j] & 0xffffffffull;
  }
  set_digits(reg, v.data(), count);
}

// canonical little-endian 32-bit words of the residue, 2^p - 1 -> 0 (what the plugin ABI exchanges: EngineApi.cpp:210-218)
void CrtEngine::get_words(size_t reg, uint32_t* w, size_t count) {
  Impl& im = *im_;
  const size_t n = im.g.n, need = (size_t(im.g.p) + 31) / 32;
  if (count < need) throw std::runtime_error("get_words: buffer too small");
  std::vector<uint64_t> d(n);
  get_digits(reg, d.data(), n, true);
  bool ones = true;
  for (size_t j = 0; j < n && ones; ++j) ones = d[j] == ((uint64_t(1) << im.width[j]) - 1);
  std::memset(w, 0, count * 4);
  if (ones) return;
  size_t bit = 0;
  for (size_t j = 0; j < n; ++j) {
    const size_t wi = bit >> 5, sh = bit & 31;
    const unsigned __int128 v = (unsigned __int128)d[j] << sh;
    w[wi] |= uint32_t(v);
    if (wi + 1 < count) w[wi + 1] |= uint32_t(v >> 32);
    if (wi + 2 < count) w[wi + 2] |= uint32_t(v >> 64);
    bit += im.width[j];
  }
}
// reg <- the value of `count` little-endian 32-bit words (< 2^p; bits beyond p must be zero): cut into digits on the host
void CrtEngine::set_words(size_t reg, const uint32_t* w, size_t count) {
  Impl& im = *im_;
  const size_t n = im.g.n, need = (size_t(im.g.p) + 31) / 32;
  if (count > need) for (size_t k = need; k < count; ++k) if (w[k]) throw std::runtime_error("set_words: value does not fit 2^p");
  if (count >= need && (im.g.p & 31) && (w[need - 1] >> (im.g.p & 31))) throw std::runtime_error("set_words: value does not fit 2^p");
  std::vector<uint64_t> d(n, 0);
  size_t bit = 0;
  auto word = [&](size_t k) -> uint64_t { return k < count ? w[k] : 0; };
  for (size_t j = 0; j < n; ++j) {
    const size_t wi = bit >> 5, sh = bit & 31;
    const unsigned __int128 v = ((unsigned __int128)word(wi) | ((unsigned __int128)word(wi + 1) << 32) | ((unsigned __int128)word(wi + 2) << 64)) >> sh;
    d[j] = uint64_t(v) & ((uint64_t(1) << im.width[j]) - 1);
    bit += im.width[j];
  }
  set_digits(reg, d.data(), n);
}
// the low 64 bits of the canonical residue: canonical form on the device, the first digits cross PCIe
uint64_t CrtEngine::res64(size_t reg) {
  Impl& im = *im_;
  check_digits(reg, "res64");
  if (!im.host_carry) {
    const uint64_t* c = canon_digits(reg, 0);
    const size_t have = std::min<size_t>(im.g.n, 8);     // widths are at least 15 bits here (constructor): 8 digits hold more than 64 bits
    uint64_t head[8];
    chk(hipMemcpyAsync(head, c, have * 8, hipMemcpyDeviceToHost, im.stream), "copy");
    uint32_t flags[4];
    if (canon_flags_ok(flags)) {
      if (flags[0]) return 0;                             // 2^p - 1 = 0
      unsigned __int128 r = 0; unsigned sh = 0;
      for (size_t k = 0; k < have && sh < 64; ++k) { r |= (unsigned __int128)head[k] << sh; sh += im.width[k]; }
      if (size_t(im.g.p) < 64) r &= (((unsigned __int128)1) << im.g.p) - 1;
      return uint64_t(r);
    }
  }
  std::vector<uint32_t> w((size_t(im.g.p) + 31) / 32 + 2, 0);
  get_words(reg, w.data(), w.size());
  return uint64_t(w[0]) | (uint64_t(w[1]) << 32);
}
// same value mod 2^p - 1 (engine::is_equal, engine.h:148): both canonical forms and the comparison on the device, 16 bytes cross PCIe
// (the reference reads both registers back: engine.h:148-157)
bool CrtEngine::equal(size_t a, size_t b) {
  Impl& im = *im_;
  check_digits(a, "is_equal"); check_digits(b, "is_equal");
  if (!im.host_carry) {
    const uint64_t* ca = canon_digits(a, 0);
    const uint64_t* cb = canon_digits(b, 1);
    chk(canon64_compare(ca, cb, im.g.n, canon64_flags(im.g.n, im.canon) + 2, im.stream), "compare");
    uint32_t flags[4];
    if (canon_flags_ok(flags)) return flags[2] == 0;      // (2^p - 1 is written as 0 by both, so 0 == 2^p - 1 holds)
  }
  const size_t need = (size_t(im.g.p) + 31) / 32;
  std::vector<uint32_t> wa(need), wb(need);
  get_words(a, wa.data(), need); get_words(b, wb.data(), need);
  return wa == wb;
}

size_t CrtEngine::register_data_size() const { return size_t(im_->g.n) * 12 + 8; }
void CrtEngine::get_data(size_t src, void* data, size_t size) {
  Impl& im = *im_;
  if (src >= im.regs.size()) throw std::runtime_error("get_data: register index out of range");
  if (size != register_data_size()) throw std::runtime_error("get_data: size mismatch");
  sync();
  unsigned char* out = static_cast<unsigned char*>(data);
  const size_t n = im.g.n, slots = size_t(im.gr.odd) * im.gr.h;   // 2 slots' worth of words per slot: 16 + 8 bytes = 12 bytes a word
  std::memset(out, 0, size);
  const Impl::Register& r = im.regs[src];
  if (r.image) { chk(hipMemcpy(out, r.i61, slots * 16, hipMemcpyDeviceToHost), "copy"); chk(hipMemcpy(out + slots * 16, r.i31, slots * 8, hipMemcpyDeviceToHost), "copy"); }
  else chk(hipMemcpy(out, r.x, n * 8, hipMemcpyDeviceToHost), "copy");
  const uint64_t tag = r.image ? 1 : 0;
  std::memcpy(out + n * 12, &tag, 8);
}
void CrtEngine::set_data(size_t dst, const void* data, size_t size) {
  Impl& im = *im_;
  if (dst >= im.regs.size()) throw std::runtime_error("set_data: register index out of range");
  if (size != register_data_size()) throw std::runtime_error("set_data: size mismatch");
  const unsigned char* in = static_cast<const unsigned char*>(data);
  const size_t n = im.g.n, slots = size_t(im.gr.odd) * im.gr.h;
  uint64_t tag = 0;
  std::memcpy(&tag, in + n * 12, 8);
  if (tag > 1) throw std::runtime_error("set_data: not an image written by this engine");
  sync();
  Impl::Register& r = im.regs[dst];
  if (tag == 1) {
    if (!r.i61) { chk(hipMalloc(reinterpret_cast<void**>(&r.i61), slots * 16), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&r.i31), slots * 8), "hipMalloc"); }
    chk(hipMemcpy(r.i61, in, slots * 16, hipMemcpyHostToDevice), "copy"); chk(hipMemcpy(r.i31, in + slots * 16, slots * 8, hipMemcpyHostToDevice), "copy");
  } else {
    chk(hipMemcpy(r.x, in, n * 8, hipMemcpyHostToDevice), "copy");
    int e = 0;   // the image holds weakly carried digits, possibly after additions: measure what it needs
    for (size_t j = 0; j < n; ++j) { uint64_t v; std::memcpy(&v, in + j * 8, 8); const int bits = v ? 64 - __builtin_clzll(v) : 0; e = std::max(e, bits - int(im.width[j])); }
    im.excess[dst] = e;
  }
  r.image = tag == 1;
}

void CrtEngine::time_square_mul(size_t reg, uint32_t a, size_t iters, double* total_ms, double* kernel_ms, size_t kernel_count) {
  Impl& im = *im_;
  check_digits(reg, "time_square_mul");
  chk(hipSetDevice(im.device), "hipSetDevice");
  std::vector<double> acc(kKernels, 0.0);
  double total = 0;
  for (size_t it = 0; it < iters; ++it) {
    launch_transform(reg, 0, 0, a, true);
    chk(hipEventSynchronize(im.ev[kKernels]), "sync");
    for (int k = 0; k < kKernels; ++k) { float ms = 0; chk(hipEventElapsedTime(&ms, im.ev[k], im.ev[k + 1]), "elapsed"); acc[k] += ms; }
    float ms = 0; chk(hipEventElapsedTime(&ms, im.ev[0], im.ev[kKernels]), "elapsed"); total += ms;
  }
  chk(hipGetLastError(), "kernel");
  if (total_ms) *total_ms = total;   // sum over the iterations, like Engine::time_square_mul
  for (size_t k = 0; k < kernel_count && k < size_t(kKernels); ++k) if (kernel_ms) kernel_ms[k] = iters ? acc[k] / double(iters) : 0;
}

}  // namespace mi355
