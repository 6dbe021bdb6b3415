// Device-side canonical form of a residue register: strong carry with wrap-around, equality and res64 without
// reading the register back.  The reference does this on the host: D2H of the whole register and a
// sequential O(n) carry loop (include/marin/engine_gpu.h:1534-1561, include/marin/engine.h:148-157,257-295).
//
// Input: a normalised digit register (run carries and small subtraction applied; u32 digits in tile-major order,
// the last digit of a run may exceed its width).  Output: the n digits in NATURAL order, each < 2^width, the value
// 2^p - 1 mapped to 0 (engine.h:188-196), in a u32 array.
//   k_gather      tile-major -> natural order
//   k_local x 3   d'[j] = (d[j] mod 2^w_j) + (d[j-1] >> w_{j-1}), cyclic: carries shrink by a factor 2^w per pass, so
//                 after them every digit is <= 2^w_j (checked: flag [1]) and what is left is a 0/1 carry chain
//   k_scan_blocks (generate, propagate) of each block of 4096 digits;  k_scan_top: the carry into every block,
//                 closed cyclically (2^p = 1: the carry out of the last digit enters digit 0)
//   k_apply       resolves the chain inside each block
// Widths are recomputed from ceil(p j / n) (ibdwt.h:127-132), no table.  HBM-bound, ~6 sweeps of 4n bytes: a Gerbicz
// check moves a few words over PCIe instead of two registers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace mi355 {
namespace {

struct CanonGeom { uint32_t p, n, logn2, r5, M1, M2, C; };   // n = r5 * 2^logn2 (r5: the odd factor 1, 3, 5 or 9); M1 = 0: natural digit order

__device__ __forceinline__ uint64_t ceil_pj_n(const CanonGeom& g, uint64_t j) {
  const uint64_t x = uint64_t(g.p) * j + (g.n - 1);
  const uint64_t y = x >> g.logn2;
  return g.r5 == 1 ? y : y / g.r5;
}
__device__ __forceinline__ uint32_t width_of(const CanonGeom& g, uint64_t j) { return uint32_t(ceil_pj_n(g, j + 1) - ceil_pj_n(g, j)); }

// memory position of natural digit j (plan.hpp Plan::pos)
__device__ __forceinline__ size_t pos_of(const CanonGeom& g, uint32_t j) {
  if (g.M1 == 0) return j;
  const uint32_t i = j >> 1, b = j & 1;
  const uint32_t i1 = i / g.M2, i2 = i - i1 * g.M2;
  const uint32_t T = i2 / g.C, c = i2 - T * g.C;
  return ((size_t(T) * g.M1 + i1) * g.C + c) * 2 + b;
}

// T: digit type, uint32_t (Goldilocks engine, widths < 32) or uint64_t (the GF(M61^2) x GF(M31^2) engine: widths up to 39 bits)
template <class T>
__global__ void __launch_bounds__(256) k_gather(CanonGeom g, const T* __restrict__ digits, T* __restrict__ nat) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < g.n) nat[j] = digits[pos_of(g, j)];
}

template <class T>
__global__ void __launch_bounds__(256) k_local(CanonGeom g, const T* __restrict__ in, T* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= g.n) return;
  const uint32_t jp = j ? j - 1 : g.n - 1;
  const uint64_t o0 = ceil_pj_n(g, jp), o1 = ceil_pj_n(g, uint64_t(jp) + 1), o2 = ceil_pj_n(g, uint64_t(j) + 1);
  const uint32_t wp = uint32_t(o1 - o0);
  const uint32_t w = uint32_t(o2 - (j ? o1 : 0));   // for j = 0 the previous digit is n - 1: o1 = p, offset of digit 0 is 0
  out[j] = (in[j] & ((T(1) << w) - T(1))) + (in[jp] >> wp);
}

constexpr int kPerThread = 16, kBlockDigits = 256 * kPerThread;

// (G, P) of the thread's digits / of the block.  Digit value v <= 2^w: generates iff v == 2^w, propagates iff v == 2^w - 1.
template <class T>
__global__ void __launch_bounds__(256) k_scan_blocks(CanonGeom g, const T* __restrict__ nat, uint32_t* __restrict__ agg, uint32_t* __restrict__ err) {
  __shared__ uint32_t sg[256], sp[256];
  const uint32_t t = threadIdx.x, j0 = blockIdx.x * kBlockDigits + t * kPerThread;
  uint32_t G = 0, Pm = 1;
  uint64_t o = (j0 < g.n) ? ceil_pj_n(g, j0) : 0;
  for (int k = 0; k < kPerThread; ++k) {
    const uint32_t j = j0 + k;
    if (j >= g.n) break;
    const uint64_t on = ceil_pj_n(g, uint64_t(j) + 1);
    const uint32_t w = uint32_t(on - o); o = on;
    const T v = nat[j];
    if (v > (T(1) << w)) atomicOr(err, 1u);   // cannot happen after the local passes (the caller falls back to the host carry)
    const uint32_t gj = uint32_t(v >> w), pj = (v == (T(1) << w) - T(1)) ? 1u : 0u;
    G = gj | (pj & G);
    Pm &= pj;
  }
  sg[t] = G; sp[t] = Pm;
  __syncthreads();
  if (t == 0) {
    uint32_t bg = 0, bp = 1;
    for (int i = 0; i < 256; ++i) { bg = sg[i] | (sp[i] & bg); bp &= sp[i]; }
    agg[blockIdx.x] = bg | (bp << 1);
  }
}

// carry into every block (cin[b]); flags[0] = 1 when the value is 2^p - 1 (every digit propagates, nothing generates).
// One work-group of 256 threads: each composes the (generate, propagate) pairs of its stretch of blocks, the 256 stretch summaries are
// scanned by thread 0 (256 steps in LDS instead of 2 x nblocks dependent loads from memory: 185 -> ~10 us at n = 2^23), then every
// thread walks its stretch again with its carry-in.
__global__ void __launch_bounds__(256) k_scan_top(const uint32_t* __restrict__ agg, uint32_t nblocks, uint32_t* __restrict__ cin, uint32_t* __restrict__ flags) {
  __shared__ uint32_t sg[256], sp[256], sc[256];
  if (blockIdx.x != 0) return;
  const uint32_t t = threadIdx.x, per = (nblocks + 255) / 256;
  const uint32_t b0 = t * per, b1 = min(b0 + per, nblocks);
  uint32_t G = 0, Pm = 1;
  for (uint32_t b = b0; b < b1; ++b) { const uint32_t a = agg[b]; G = (a & 1u) | ((a >> 1) & G); Pm &= (a >> 1); }
  sg[t] = G; sp[t] = Pm;
  __syncthreads();
  if (t == 0) {
    uint32_t TG = 0, TP = 1;
    for (uint32_t k = 0; k < 256; ++k) { TG = sg[k] | (sp[k] & TG); TP &= sp[k]; }
    uint32_t c = TP ? 0u : TG;   // the carry out of the last digit re-enters digit 0; with P_total it could be anything: take 0
    flags[0] = (TP && !TG) ? 1u : 0u;
    for (uint32_t k = 0; k < 256; ++k) { sc[k] = c; c = sg[k] | (sp[k] & c); }
  }
  __syncthreads();
  uint32_t c = sc[t];
  for (uint32_t b = b0; b < b1; ++b) { cin[b] = c; const uint32_t a = agg[b]; c = (a & 1u) | ((a >> 1) & c); }
}

template <class T>
__global__ void __launch_bounds__(256) k_apply(CanonGeom g, const T* __restrict__ nat, const uint32_t* __restrict__ cin, const uint32_t* __restrict__ flags,
                                               T* __restrict__ out) {
  __shared__ uint32_t sg[256], sp[256], sc[256];
  const uint32_t t = threadIdx.x, j0 = blockIdx.x * kBlockDigits + t * kPerThread;
  T v[kPerThread];
  uint32_t w[kPerThread];
  uint32_t G = 0, Pm = 1;
  uint64_t o = (j0 < g.n) ? ceil_pj_n(g, j0) : 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const uint32_t j = j0 + k;
    v[k] = 0; w[k] = 1;
    if (j < g.n) {
      const uint64_t on = ceil_pj_n(g, uint64_t(j) + 1);
      w[k] = uint32_t(on - o); o = on;
      v[k] = nat[j];
      const uint32_t gj = uint32_t(v[k] >> w[k]), pj = (v[k] == (T(1) << w[k]) - T(1)) ? 1u : 0u;
      G = gj | (pj & G);
      Pm &= pj;
    }
  }
  sg[t] = G; sp[t] = Pm;
  __syncthreads();
  if (t == 0) {
    uint32_t c = cin[blockIdx.x];
    for (int i = 0; i < 256; ++i) { sc[i] = c; c = sg[i] | (sp[i] & c); }
  }
  __syncthreads();
  uint32_t c = sc[t];
  const bool zero = flags[0] != 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const uint32_t j = j0 + k;
    if (j < g.n) {
      const T s = v[k] + c;
      out[j] = zero ? T(0) : (s & ((T(1) << w[k]) - T(1)));
      c = uint32_t(s >> w[k]);
    }
  }
}

// One local carry pass in place of layout (tile-major in, tile-major out): d'[j] = (d[j] mod 2^w_j) + (d[j-1] >> w_{j-1}).
// For plans whose runs are only two digits long (C = 1): the weak carry of a back sweep then leaves up to log2(n) - 2
// excess bits on a run's second digit, too much for the next squaring once n >= 2^19; one pass brings it down to
// log2(n) - 2 - w bits (the reference spreads a work-group's last carry over four digits, adc4 marin.cl:203-212).
// One thread per PAIR in memory order (tile T, row i1, column c): its two digits are one 8-byte word, the digit before the pair is the
// odd digit of the pair one column to the left -- the previous pair of the run, or the last pair of the same row in tile T - 1 -- so every
// access of a wavefront runs along i1 and is coalesced (the round-2 form walked the digits in natural order: 4-byte gathers M1 pairs
// apart, 4.3 ms of a 12.1 ms squaring at n = 5 2^25).
__global__ void __launch_bounds__(256) k_relax(CanonGeom g, const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
  const size_t e = size_t(blockIdx.x) * 256 + threadIdx.x;   // pair index in memory: (T M1 + i1) C + c
  if (e >= g.n / 2) return;
  const uint32_t c = uint32_t(e % g.C), i1 = uint32_t((e / g.C) % g.M1), T = uint32_t(e / (size_t(g.C) * g.M1));
  const uint32_t i2 = T * g.C + c;
  const uint64_t j = 2 * (uint64_t(i1) * g.M2 + i2);        // natural index of the pair's even digit
  // the pair to the left in digit order: column i2 - 1 of the same row, or the last column of the row above (cyclic)
  const uint32_t pi1 = i2 ? i1 : (i1 ? i1 - 1 : g.M1 - 1), pi2 = i2 ? i2 - 1 : g.M2 - 1;
  const size_t pe = (size_t(pi2 / g.C) * g.M1 + pi1) * g.C + (pi2 % g.C);
  const uint64_t jp = j ? j - 1 : uint64_t(g.n) - 1;
  const uint64_t o0 = ceil_pj_n(g, jp), o1 = j ? ceil_pj_n(g, j) : 0, o2 = ceil_pj_n(g, j + 1), o3 = ceil_pj_n(g, j + 2);
  const uint32_t wp = uint32_t((j ? o1 : uint64_t(g.p)) - o0), w0 = uint32_t(o2 - o1), w1 = uint32_t(o3 - o2);
  const uint2 d = reinterpret_cast<const uint2*>(in)[e];
  const uint32_t prev_odd = in[2 * pe + 1];
  reinterpret_cast<uint2*>(out)[e] = make_uint2((d.x & ((1u << w0) - 1u)) + (prev_odd >> wp), (d.y & ((1u << w1) - 1u)) + (d.x >> w0));
}

// natural order -> tile-major (set_digits / set_words without the host-side re-tiling)
__global__ void __launch_bounds__(256) k_scatter(CanonGeom g, const uint32_t* __restrict__ nat, uint32_t* __restrict__ digits) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < g.n) digits[pos_of(g, j)] = nat[j];
}
// a small constant spread over the first digits (the register is zero otherwise)
__global__ void k_set_small(CanonGeom g, uint32_t* __restrict__ digits, uint32_t value) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t v = value;
  for (uint32_t j = 0; j < g.n && v; ++j) {
    const uint32_t w = width_of(g, j);
    digits[pos_of(g, j)] = uint32_t(v & ((uint64_t(1) << w) - 1));
    v >>= w;
  }
}

template <class T>
__global__ void __launch_bounds__(256) k_compare(const T* __restrict__ a, const T* __restrict__ b, uint32_t n, uint32_t* __restrict__ diff) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  const bool ne = (j < n) && (a[j] != b[j]);
  if (__any(ne) && (threadIdx.x & 63) == 0) atomicOr(diff, 1u);
}

// dst[j] += (2^w_j - 1) - canon[j]: the digit-wise complement of a canonical residue, i.e. dst - src mod 2^p - 1 (engine::sub_reg)
template <class T>
__global__ void __launch_bounds__(256) k_add_complement(CanonGeom g, T* __restrict__ dst, const T* __restrict__ canon) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= g.n) return;
  const uint32_t w = width_of(g, j);
  dst[pos_of(g, j)] += ((T(1) << w) - T(1)) - canon[j];
}

// the whole pipeline on an array in natural order or tile-major order (g.M1)
template <class T>
void launch_pipeline(const CanonGeom& g, const T* digits, T* out, T* A, T* B, uint32_t* agg, uint32_t* cin, uint32_t* flags, hipStream_t s) {
  const uint32_t n = g.n, nb = (n + kBlockDigits - 1) / kBlockDigits, ge = (n + 255) / 256;
  const T* first = digits;
  if (g.M1 != 0) { hipLaunchKernelGGL(k_gather<T>, dim3(ge), dim3(256), 0, s, g, digits, A); first = A; }
  CanonGeom gn = g; gn.M1 = 0;   // from here on everything is in natural order
  hipLaunchKernelGGL(k_local<T>, dim3(ge), dim3(256), 0, s, gn, first, B);
  hipLaunchKernelGGL(k_local<T>, dim3(ge), dim3(256), 0, s, gn, B, A);
  hipLaunchKernelGGL(k_local<T>, dim3(ge), dim3(256), 0, s, gn, A, B);
  hipLaunchKernelGGL(k_scan_blocks<T>, dim3(nb), dim3(256), 0, s, gn, B, agg, flags + 1);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, s, agg, nb, cin, flags);
  hipLaunchKernelGGL(k_apply<T>, dim3(nb), dim3(256), 0, s, gn, B, cin, flags, out);
}

CanonGeom geom_of(const DevPlan& pl, uint32_t p) {
  CanonGeom g;
  g.p = p; g.n = pl.n; g.r5 = pl.r5; g.M1 = pl.M1; g.M2 = pl.M2; g.C = pl.C;
  g.logn2 = 0;
  while ((uint64_t(pl.r5) << g.logn2) < pl.n) ++g.logn2;
  return g;
}

}  // namespace

size_t canon_scratch_words(const DevPlan& pl) {
  const size_t nb = (size_t(pl.n) + kBlockDigits - 1) / kBlockDigits;
  return 2 * size_t(pl.n) + 2 * nb + 16;   // two digit arrays, block aggregates, block carries, flags
}

// digits: normalised register (tile-major).  out: n canonical digits, natural order.  scratch: canon_scratch_words() u32.
// flags (device, inside scratch; the caller clears them): [0] value was 2^p - 1 (written as 0), [1] a digit was still
// too wide for the 0/1 carry chain (sticky; the caller then uses the host carry), [2] compare result (sticky).
hipError_t canon_launch(const DevPlan& pl, uint32_t p, const uint32_t* digits, uint32_t* out, uint32_t* scratch, hipStream_t s) {
  const CanonGeom g = geom_of(pl, p);
  const uint32_t n = pl.n, nb = (n + kBlockDigits - 1) / kBlockDigits;
  uint32_t* agg = scratch + 2 * size_t(n);
  launch_pipeline<uint32_t>(g, digits, out, scratch, scratch + n, agg, agg + nb, agg + 2 * nb, s);
  return hipGetLastError();
}
uint32_t* canon_flags(const DevPlan& pl, uint32_t* scratch) {
  const size_t nb = (size_t(pl.n) + kBlockDigits - 1) / kBlockDigits;
  return scratch + 2 * size_t(pl.n) + 2 * nb;
}
hipError_t canon_relax(const DevPlan& pl, uint32_t p, const uint32_t* in, uint32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_relax, dim3(uint32_t((size_t(pl.n) / 2 + 255) / 256)), dim3(256), 0, s, geom_of(pl, p), in, out);
  return hipGetLastError();
}
hipError_t canon_scatter(const DevPlan& pl, uint32_t p, const uint32_t* nat, uint32_t* digits, hipStream_t s) {
  hipLaunchKernelGGL(k_scatter, dim3((pl.n + 255) / 256), dim3(256), 0, s, geom_of(pl, p), nat, digits);
  return hipGetLastError();
}
hipError_t canon_set_small(const DevPlan& pl, uint32_t p, uint32_t* digits, uint32_t value, hipStream_t s) {
  hipLaunchKernelGGL(k_set_small, dim3(1), dim3(64), 0, s, geom_of(pl, p), digits, value);
  return hipGetLastError();
}
hipError_t canon_compare(const uint32_t* a, const uint32_t* b, uint32_t n, uint32_t* diff_flag, hipStream_t s) {
  hipLaunchKernelGGL(k_compare<uint32_t>, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, diff_flag);
  return hipGetLastError();
}

// ---- the same for the second field family: u64 digits in natural order, n = odd * 2^ln (crt_engine.hip) ----
static CanonGeom geom64(uint32_t p, uint32_t n, uint32_t odd) {
  CanonGeom g;
  g.p = p; g.n = n; g.r5 = odd; g.M1 = 0; g.M2 = 0; g.C = 0;
  g.logn2 = 0;
  while ((uint64_t(odd) << g.logn2) < n) ++g.logn2;
  return g;
}
size_t canon64_scratch_bytes(uint32_t n) {
  const size_t nb = (size_t(n) + kBlockDigits - 1) / kBlockDigits;
  return 2 * size_t(n) * 8 + (2 * nb + 16) * 4;
}
uint32_t* canon64_flags(uint32_t n, void* scratch) {
  const size_t nb = (size_t(n) + kBlockDigits - 1) / kBlockDigits;
  return reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(scratch) + 2 * size_t(n) * 8) + 2 * nb;
}
// digits: weakly carried u64 digits (any excess the three local passes remove: up to ~3 w bits); out: canonical digits, 2^p - 1 -> 0
hipError_t canon64_launch(uint32_t p, uint32_t n, uint32_t odd, const uint64_t* digits, uint64_t* out, void* scratch, hipStream_t s) {
  const CanonGeom g = geom64(p, n, odd);
  const size_t nb = (size_t(n) + kBlockDigits - 1) / kBlockDigits;
  uint64_t* A = static_cast<uint64_t*>(scratch);
  uint32_t* agg = reinterpret_cast<uint32_t*>(A + 2 * size_t(n));
  launch_pipeline<uint64_t>(g, digits, out, A, A + n, agg, agg + nb, agg + 2 * nb, s);
  return hipGetLastError();
}
hipError_t canon64_compare(const uint64_t* a, const uint64_t* b, uint32_t n, uint32_t* diff_flag, hipStream_t s) {
  hipLaunchKernelGGL(k_compare<uint64_t>, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, diff_flag);
  return hipGetLastError();
}
// one local carry pass (in -> out): digits of up to w + e bits come out below 2^w + 2^e
hipError_t canon64_relax(uint32_t p, uint32_t n, uint32_t odd, const uint64_t* in, uint64_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_local<uint64_t>, dim3((n + 255) / 256), dim3(256), 0, s, geom64(p, n, odd), in, out);
  return hipGetLastError();
}
hipError_t canon64_add_complement(uint32_t p, uint32_t n, uint32_t odd, uint64_t* dst, const uint64_t* canon, hipStream_t s) {
  hipLaunchKernelGGL(k_add_complement<uint64_t>, dim3((n + 255) / 256), dim3(256), 0, s, geom64(p, n, odd), dst, canon);
  return hipGetLastError();
}

}  // namespace mi355
