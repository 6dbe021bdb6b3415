// First kernel of the second field family: the fused unweight + Garner + carry sweep of the paired-NTT
// squaring over GF(M61^2) x GF(M31^2).  Reference: third_party/aevum/src/cl/carry.cl:506-588 (FFT3161 `carry`) with
// weightAndCarryPair, carryutil.cl:440-470 ("n3161 = n61 * M31 + n31"); CPU form docs/mersenne2_mixed_crt_2d_half_fast/
// mersenne2_mixed_crt_2d_half_fast.cpp:429-441,931-1001.  Input: the two residues of every (still weighted, already scaled)
// convolution coefficient in logical digit order; output: digits in base 2^width (widths up to 39 bits: u64) and one carry
// word per run.  One thread owns a run of kRun consecutive digits (sequential carry inside the run, as the Goldilocks back
// sweep does); k_crt_runs_fix then folds each run's carry word into the following run.  HBM-bound by design: 12 bytes in and
// 8 bytes out per word, ~60 VALU instructions per word (two rotations, one M61 multiply, 128-bit carry arithmetic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>

#include "crt_field.hpp"

namespace mi355 {
namespace crt {

// digits[j]: value mod 2^width_j; carry_out[2 run .. 2 run + 1]: the 128-bit carry leaving the run
__global__ void __launch_bounds__(256) k_crt_runs(Geom g, const uint64_t* __restrict__ in61, const uint32_t* __restrict__ in31,
                                                  uint64_t* __restrict__ digits, uint64_t* __restrict__ carry_out) {
  const uint32_t run = blockIdx.x * 256 + threadIdx.x;
  const uint32_t j0 = run * kRun;
  if (j0 >= g.n) return;
  DigitWalk dw; dw.start(g, j0);
  unsigned __int128 carry = 0;
  // the run's inputs first (16-byte loads, all in flight before the carry chain starts); n is a multiple of kRun (checked on the host)
  uint64_t v61[kRun]; uint32_t v31[kRun];
  {
    const ulonglong2* p61 = reinterpret_cast<const ulonglong2*>(in61 + j0);
    const uint4* p31 = reinterpret_cast<const uint4*>(in31 + j0);
#pragma unroll
    for (int k = 0; k < kRun / 2; ++k) { const ulonglong2 q = p61[k]; v61[2 * k] = q.x; v61[2 * k + 1] = q.y; }
#pragma unroll
    for (int k = 0; k < kRun / 4; ++k) { const uint4 q = p31[k]; v31[4 * k] = q.x; v31[4 * k + 1] = q.y; v31[4 * k + 2] = q.z; v31[4 * k + 3] = q.w; }
  }
  uint64_t out[kRun];
#pragma unroll
  for (int k = 0; k < kRun; ++k) {
    const uint64_t x61 = rot61(v61[k], dw.unweight61());
    const uint32_t x31 = rot31(v31[k], dw.unweight31());
    // Garner: v = x31 + M31 * ((x61 - x31) / M31 mod M61)  <  M61 * M31
    const uint64_t d = x61 >= x31 ? x61 - x31 : x61 + M61 - x31;
    const uint64_t t = mul61(d, g.inv31);
    const unsigned __int128 v = ((unsigned __int128)t << 31) - t + x31;
    const unsigned __int128 s = v * g.a + carry;
    const uint32_t width = dw.width(g);
    out[k] = uint64_t(s) & ((uint64_t(1) << width) - 1);
    carry = s >> width;
    dw.next(g);
  }
  {
    ulonglong2* po = reinterpret_cast<ulonglong2*>(digits + j0);
#pragma unroll
    for (int k = 0; k < kRun / 2; ++k) po[k] = make_ulonglong2(out[2 * k], out[2 * k + 1]);
  }
  carry_out[2 * size_t(run)] = uint64_t(carry);
  carry_out[2 * size_t(run) + 1] = uint64_t(carry >> 64);
}

// the carry word of the previous run (cyclically: 2^p = 1) goes through this run; what is left after its last digit (at most
// a few units) is returned in residual[run] for the final strong carry
__global__ void __launch_bounds__(256) k_crt_runs_fix(Geom g, uint64_t* __restrict__ digits, const uint64_t* __restrict__ carry_in, uint64_t* __restrict__ residual) {
  const uint32_t run = blockIdx.x * 256 + threadIdx.x;
  const uint32_t nruns = (g.n + kRun - 1) / kRun;
  if (run >= nruns) return;
  const uint32_t prev = run ? run - 1 : nruns - 1;
  unsigned __int128 carry = ((unsigned __int128)carry_in[2 * size_t(prev) + 1] << 64) | carry_in[2 * size_t(prev)];
  const uint32_t j0 = run * kRun;
  DigitWalk dw; dw.start(g, j0);
  for (int k = 0; k < kRun && j0 + k < g.n; ++k) {
    const uint32_t j = j0 + k;
    const uint32_t width = dw.width(g);
    const unsigned __int128 s = (unsigned __int128)digits[j] + carry;
    digits[j] = uint64_t(s) & ((uint64_t(1) << width) - 1);
    carry = s >> width;
    if (carry == 0) break;   // a 92-bit carry is gone after three digits
    dw.next(g);
  }
  residual[run] = uint64_t(carry);
}

// The same sweep with the run-to-run hand-over inside the work-group (the engine's path): after the private pass every thread takes
// the carry of the run before it from LDS, lets it run through its own eight digits (still in registers) and passes what is left (0 or
// a unit) on to the first digit of the following run, again through LDS.  Only the first run of a work-group depends on another
// work-group: edge_out[3 g .. 3 g + 2] = carry (128 bits) and leftover of the last run of group g, folded in by k_crt_edges
// (n / 2048 threads) -- instead of a second sweep over all digits (k_crt_runs_fix, 42 us at 9.4 M words).
__global__ void __launch_bounds__(256) k_crt_runs_linked(Geom g, const uint64_t* __restrict__ in61, const uint32_t* __restrict__ in31,
                                                         uint64_t* __restrict__ digits, uint64_t* __restrict__ edge_out) {
  __shared__ uint64_t Clo[256], Chi[256], Rs[256];
  const uint32_t tid = threadIdx.x, run = blockIdx.x * 256 + tid;
  const uint32_t j0 = run * kRun;
  const bool live = j0 < g.n;
  uint64_t out[kRun];
  uint32_t wd[kRun];
  unsigned __int128 carry = 0;
  if (live) {
    DigitWalk dw; dw.start(g, j0);
    uint64_t v61[kRun]; uint32_t v31[kRun];
    const ulonglong2* p61 = reinterpret_cast<const ulonglong2*>(in61 + j0);
    const uint4* p31 = reinterpret_cast<const uint4*>(in31 + j0);
#pragma unroll
    for (int k = 0; k < kRun / 2; ++k) { const ulonglong2 q = p61[k]; v61[2 * k] = q.x; v61[2 * k + 1] = q.y; }
#pragma unroll
    for (int k = 0; k < kRun / 4; ++k) { const uint4 q = p31[k]; v31[4 * k] = q.x; v31[4 * k + 1] = q.y; v31[4 * k + 2] = q.z; v31[4 * k + 3] = q.w; }
#pragma unroll
    for (int k = 0; k < kRun; ++k) {
      const uint64_t x61 = rot61(v61[k], dw.unweight61());
      const uint32_t x31 = rot31(v31[k], dw.unweight31());
      const uint64_t d = x61 >= x31 ? x61 - x31 : x61 + M61 - x31;
      const uint64_t t = mul61(d, g.inv31);
      const unsigned __int128 v = ((unsigned __int128)t << 31) - t + x31;
      const unsigned __int128 s = v * g.a + carry;
      wd[k] = dw.width(g);
      out[k] = uint64_t(s) & ((uint64_t(1) << wd[k]) - 1);
      carry = s >> wd[k];
      dw.next(g);
    }
  }
  Clo[tid] = uint64_t(carry); Chi[tid] = uint64_t(carry >> 64);
  __syncthreads();
  unsigned __int128 in = tid ? (((unsigned __int128)Chi[tid - 1] << 64) | Clo[tid - 1]) : 0;
  if (live) {
#pragma unroll
    for (int k = 0; k < kRun; ++k) {
      const unsigned __int128 s = (unsigned __int128)out[k] + in;
      out[k] = uint64_t(s) & ((uint64_t(1) << wd[k]) - 1);
      in = s >> wd[k];
    }
  }
  Rs[tid] = uint64_t(in);
  __syncthreads();
  if (live) {
    if (tid) out[0] += Rs[tid - 1];
    ulonglong2* po = reinterpret_cast<ulonglong2*>(digits + j0);
#pragma unroll
    for (int k = 0; k < kRun / 2; ++k) po[k] = make_ulonglong2(out[2 * k], out[2 * k + 1]);
  }
  // the last live run of the group hands over to the next group
  const uint32_t nruns = g.n / kRun, last = min(blockIdx.x * 256u + 255u, nruns - 1);
  if (run == last) { edge_out[3 * size_t(blockIdx.x)] = uint64_t(carry); edge_out[3 * size_t(blockIdx.x) + 1] = uint64_t(carry >> 64); edge_out[3 * size_t(blockIdx.x) + 2] = uint64_t(in); }
}
// first run of every work-group: the carry of the previous group's last run (cyclically: 2^p = 1) runs through its digits, the leftovers
// go in front of this run and of the next one without further propagation (weak carry)
__global__ void __launch_bounds__(256) k_crt_edges(Geom g, uint64_t* __restrict__ digits, const uint64_t* __restrict__ edge) {
  const uint32_t grp = blockIdx.x * 256 + threadIdx.x;
  const uint32_t nruns = g.n / kRun, ngroups = (nruns + 255) / 256;
  if (grp >= ngroups) return;
  const uint32_t prev = grp ? grp - 1 : ngroups - 1;
  unsigned __int128 carry = ((unsigned __int128)edge[3 * size_t(prev) + 1] << 64) | edge[3 * size_t(prev)];
  const uint32_t j0 = grp * 256u * kRun;
  DigitWalk dw; dw.start(g, j0);
  for (int k = 0; k < kRun; ++k) {
    const uint32_t width = dw.width(g);
    const unsigned __int128 s = (unsigned __int128)digits[j0 + k] + carry;
    digits[j0 + k] = uint64_t(s) & ((uint64_t(1) << width) - 1);
    carry = s >> width;
    if (carry == 0) break;
    dw.next(g);
  }
  digits[j0] += edge[3 * size_t(prev) + 2];                       // leftover of the previous group's last run
  if (carry) digits[(j0 + kRun) % g.n] += uint64_t(carry);        // this run's own leftover: in front of the following run
}

// residual[run] (a unit here and there, left by k_crt_runs_fix) goes in front of the following run, without propagation: the digit
// vector stays weakly carried (a digit may exceed its width by that unit), which the next transform takes as it is
__global__ void __launch_bounds__(256) k_crt_residual(Geom g, uint64_t* __restrict__ digits, const uint64_t* __restrict__ residual) {
  const uint32_t run = blockIdx.x * 256 + threadIdx.x;
  const uint32_t nruns = (g.n + kRun - 1) / kRun;
  if (run >= nruns) return;
  const uint64_t c = residual[run ? run - 1 : nruns - 1];
  if (c) digits[size_t(run) * kRun] += c;
}

static uint64_t host_pow61(uint64_t a, uint64_t e) {
  uint64_t r = 1;
  while (e) { if (e & 1) r = mul61(r, a); a = mul61(a, a); e >>= 1; }
  return r;
}

Geom make_geom(uint32_t p, size_t n, uint32_t odd, uint32_t a) {
  if (odd != 1 && odd != 3 && odd != 9) throw std::runtime_error("crt: odd radix must be 1, 3 or 9");
  if (n == 0 || n % odd || n % kRun || n > 0xfffffff0ull) throw std::runtime_error("crt: bad transform size");
  uint32_t ln = 0;
  while ((size_t(odd) << ln) < n) ++ln;
  if ((size_t(odd) << ln) != n) throw std::runtime_error("crt: transform size must be odd * 2^k");
  if (a == 0) throw std::runtime_error("crt: factor must be >= 1");
  Geom g;
  g.p = p; g.n = uint32_t(n); g.odd = odd; g.ln = ln; g.a = a;
  auto inv_small = [](uint64_t x, uint64_t m) { for (uint64_t y = 1; y < m; ++y) if (x * y % m == 1) return y; return uint64_t(0); };
  g.l61 = uint32_t(inv_small(n % 61, 61)); g.l31 = uint32_t(inv_small(n % 31, 31));
  g.inv31 = host_pow61(M31, M61 - 2);
  g.q = uint32_t(p / n); g.t = uint32_t(p % n);
  g.lt61 = uint32_t(uint64_t(g.l61) * (g.t % 61) % 61); g.lt31 = uint32_t(uint64_t(g.l31) * (g.t % 31) % 31);
  return g;
}

void crt_carry_launch(const Geom& g, const uint64_t* in61, const uint32_t* in31, uint64_t* digits, uint64_t* carry, uint64_t* residual,
                      bool fold_residual, hipStream_t s) {
  const size_t nruns = (size_t(g.n) + kRun - 1) / kRun;
  const dim3 grid(uint32_t((nruns + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_crt_runs, grid, block, 0, s, g, in61, in31, digits, carry);
  hipLaunchKernelGGL(k_crt_runs_fix, grid, block, 0, s, g, digits, carry, residual);
  if (fold_residual) hipLaunchKernelGGL(k_crt_residual, grid, block, 0, s, g, digits, residual);
}

// the engine's form: edge = 3 words per work-group of 256 runs
void crt_carry_launch_linked(const Geom& g, const uint64_t* in61, const uint32_t* in31, uint64_t* digits, uint64_t* edge, hipStream_t s) {
  const uint32_t nruns = g.n / kRun, groups = (nruns + 255) / 256;
  hipLaunchKernelGGL(k_crt_runs_linked, dim3(groups), dim3(256), 0, s, g, in61, in31, digits, edge);
  hipLaunchKernelGGL(k_crt_edges, dim3((groups + 255) / 256), dim3(256), 0, s, g, digits, edge);
}

}  // namespace crt

// host buffers in, host buffers out (a parity / timing entry point of the sweep alone; the resident engine is crt_engine.hip);
// returns the time of the two kernels in ms through *kernel_ms when it is non-null
void crt_carry_host(uint32_t p, size_t n, uint32_t odd, uint32_t a, const uint64_t* in61, const uint32_t* in31, uint64_t* digits_out,
                    uint64_t* residual_out, int device, double* kernel_ms) {
  using namespace crt;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device available: the MI355X engine has no CPU fallback");
  auto chk = [](hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string("crt_carry: ") + what + ": " + hipGetErrorString(e)); };
  const Geom g = make_geom(p, n, odd, a);
  chk(hipSetDevice(device), "hipSetDevice");
  const size_t nruns = (n + kRun - 1) / kRun;
  uint64_t *d61 = nullptr, *dd = nullptr, *dc = nullptr, *dr = nullptr; uint32_t* d31 = nullptr;
  chk(hipMalloc(reinterpret_cast<void**>(&d61), n * 8), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&d31), n * 4), "hipMalloc");
  chk(hipMalloc(reinterpret_cast<void**>(&dd), n * 8), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&dc), nruns * 16), "hipMalloc");
  chk(hipMalloc(reinterpret_cast<void**>(&dr), nruns * 8), "hipMalloc");
  chk(hipMemcpy(d61, in61, n * 8, hipMemcpyHostToDevice), "copy"); chk(hipMemcpy(d31, in31, n * 4, hipMemcpyHostToDevice), "copy");
  hipEvent_t e0, e1;
  chk(hipEventCreate(&e0), "event"); chk(hipEventCreate(&e1), "event");
  for (int rep = 0; rep < (kernel_ms ? 5 : 1); ++rep) {   // timed runs repeat the sweep (same inputs, same outputs)
    chk(hipEventRecord(e0), "event");
    crt_carry_launch(g, d61, d31, dd, dc, dr, false, nullptr);
    chk(hipEventRecord(e1), "event");
    chk(hipEventSynchronize(e1), "sync");
  }
  chk(hipGetLastError(), "launch");
  if (kernel_ms) { float ms = 0; chk(hipEventElapsedTime(&ms, e0, e1), "elapsed"); *kernel_ms = ms; }
  chk(hipMemcpy(digits_out, dd, n * 8, hipMemcpyDeviceToHost), "copy"); chk(hipMemcpy(residual_out, dr, nruns * 8, hipMemcpyDeviceToHost), "copy");
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(d61); (void)hipFree(d31); (void)hipFree(dd); (void)hipFree(dc); (void)hipFree(dr);
}

}  // namespace mi355
