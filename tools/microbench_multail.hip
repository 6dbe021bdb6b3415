// What the rare-case test at the end of gf::mul costs (round 4).  gf::mul ends with "v_cmp -> s_or -> s_cbranch": the scalar unit has to
// wait for a VALU result, and the wave issues nothing else meanwhile (in-order issue).  Variants, 8 independent products per thread and
// iteration, 256-thread blocks, 4 waves per SIMD, cycles per wave-operation per SIMD at 2.4 GHz:
//   shipped      gf::mul as it is (one test per product)
//   unchecked    the same arithmetic without the test (WRONG in ~2^-32 of the lanes: timing only -- the upper bound of what can be won)
//   pair         two products, one test (the two planes of a pair share their twiddle: p2_mul)
//   batch8       eight products, one test; the masks of the eight borrows are kept for the fix-up
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I prmers_amd/csrc -o tools/microbench_multail tools/microbench_multail.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gf.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

namespace mt {
using namespace gf;
struct Part { uint64_t r; uint64_t ge; uint64_t bw; uint32_t m; };   // r + m EPS is the product unless a lane of ge | bw is set
// the product up to the test: r, the lanes with r >= P (ge), the lanes whose lo - hh - c borrowed (bw), and the carry fold m
__device__ __forceinline__ Part mul_part(uint64_t a, uint64_t b) {
#if !defined(GF_ASM)
  return Part{gf::mul(a, b), 0, 0, 0};   // (host pass of the compiler: never called)
#else
  const uint32_t a0 = uint32_t(a), a1 = uint32_t(a >> 32), b0 = uint32_t(b), b1 = uint32_t(b >> 32);
  const uint64_t t0 = uint64_t(a0) * b0;
  const uint64_t t1 = uint64_t(a0) * b1 + (t0 >> 32);
  uint64_t c;
  Part p;
  const uint64_t t2 = dev::mad_c(a1, b0, t1, c);
  const uint64_t t3 = uint64_t(a1) * b1 + (t2 >> 32);
  const uint64_t lo = (t2 << 32) | uint32_t(t0);
  const uint64_t x = dev::sub32_c(lo, uint32_t(t3 >> 32), c, p.bw);
  asm("v_mad_u64_u32 %0, vcc, %3, -1, %4\n\t"
      "v_cmp_lt_u64 %1, %5, %0\n\t"
      "v_cndmask_b32_e32 %2, 0, %6, vcc"
      : "=&v"(p.r), "=&s"(p.ge), "=&v"(p.m)
      : "v"(uint32_t(t3)), "v"(x), "s"(dev::PM1), "v"(dev::k_ones())
      : "vcc");
  return p;
#endif
}
// the rare lanes of one product, after the fact: borrowed lanes carry 2^64 = EPS too much whether or not they carried; the others fold
__device__ __forceinline__ uint64_t fix(uint64_t v, uint64_t bw) {
#if defined(GF_ASM)
  const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#else
  const uint32_t lane = 0;
#endif
  if ((bw >> lane) & 1) return v - EPS;
  return v >= P ? v - P : v;
}
__device__ __forceinline__ uint64_t mul_unchecked(uint64_t a, uint64_t b) { const Part p = mul_part(a, b); return p.r + uint64_t(p.m); }
__device__ __forceinline__ void mul_pair(uint64_t& a0, uint64_t& a1, uint64_t w) {
  const Part p = mul_part(a0, w), q = mul_part(a1, w);
  a0 = p.r + uint64_t(p.m); a1 = q.r + uint64_t(q.m);
  if (__builtin_expect((p.ge | p.bw | q.ge | q.bw) != 0, 0)) { a0 = fix(a0, p.bw); a1 = fix(a1, q.bw); }
}
__device__ __forceinline__ void mul_batch8(uint64_t (&x)[8], uint64_t w) {
  uint64_t bw[8], any = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const Part p = mul_part(x[i], w); x[i] = p.r + uint64_t(p.m); bw[i] = p.bw; any |= p.ge | p.bw; }
  if (__builtin_expect(any != 0, 0)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = fix(x[i], bw[i]);
  }
}
// the same with the sixteen masks combined by ONE scalar block after the last product (the compiler otherwise puts every s_or right
// behind its v_cmp, where the scalar unit waits for it)
__device__ __forceinline__ void mul_batch8_late(uint64_t (&x)[8], uint64_t w) {
  uint64_t bw[8], ge[8], any;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const Part p = mul_part(x[i], w); x[i] = p.r + uint64_t(p.m); bw[i] = p.bw; ge[i] = p.ge; }
#if defined(GF_ASM)
  asm volatile("s_or_b64 %0, %1, %2\n\ts_or_b64 %0, %0, %3\n\ts_or_b64 %0, %0, %4\n\ts_or_b64 %0, %0, %5\n\ts_or_b64 %0, %0, %6\n\ts_or_b64 %0, %0, %7\n\t"
               "s_or_b64 %0, %0, %8\n\ts_or_b64 %0, %0, %9\n\ts_or_b64 %0, %0, %10\n\ts_or_b64 %0, %0, %11\n\ts_or_b64 %0, %0, %12\n\ts_or_b64 %0, %0, %13\n\t"
               "s_or_b64 %0, %0, %14\n\ts_or_b64 %0, %0, %15\n\ts_or_b64 %0, %0, %16"
               : "=&s"(any)
               : "s"(ge[0]), "s"(ge[1]), "s"(ge[2]), "s"(ge[3]), "s"(ge[4]), "s"(ge[5]), "s"(ge[6]), "s"(ge[7]),
                 "s"(bw[0]), "s"(bw[1]), "s"(bw[2]), "s"(bw[3]), "s"(bw[4]), "s"(bw[5]), "s"(bw[6]), "s"(bw[7])
               : "scc");
#else
  any = 0;
#endif
  if (__builtin_expect(any != 0, 0)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = fix(x[i], bw[i]);
  }
}
}  // namespace mt

enum { V_SHIPPED, V_UNCHECKED, V_PAIR, V_BATCH8, V_BATCH8_LATE, V_COUNT };
static const char* kNames[V_COUNT] = {"mul, shipped (one test per product)", "mul without the test (timing only)", "two products, one test", "eight products, one test", "eight products, one test, masks combined at the end"};

template <int V>
__global__ void __launch_bounds__(256) k_time(uint64_t* out, int iters) {
  uint64_t x[8], y = (uint64_t(threadIdx.x) * 0x9e3779b97f4a7c15ull + blockIdx.x) % gf::P;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (y * (2 * i + 3) + i) % gf::P;
  for (int it = 0; it < iters; ++it) {
    if (V == V_SHIPPED) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = gf::mul(x[i], y);
    } else if (V == V_UNCHECKED) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = mt::mul_unchecked(x[i], y);
    } else if (V == V_PAIR) {
#pragma unroll
      for (int i = 0; i < 8; i += 2) mt::mul_pair(x[i], x[i + 1], y);
    }
    else if (V == V_BATCH8) mt::mul_batch8(x, y);
    else mt::mul_batch8_late(x, y);
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_check(const uint64_t* a, const uint64_t* b, uint64_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x[8];
  for (int j = 0; j < 8; ++j) x[j] = a[(i + j) % n];
  uint64_t p0 = a[i], p1 = a[(i + 1) % n];
  mt::mul_pair(p0, p1, b[i]);
  mt::mul_batch8(x, b[i]);
  uint64_t z[8];
  for (int j = 0; j < 8; ++j) z[j] = a[(i + j) % n];
  mt::mul_batch8_late(z, b[i]);
  for (int j = 0; j < 8; ++j) if (z[j] != x[j]) x[j] = ~0ull;   // (any difference shows up as a mismatch below)
  out[size_t(i) * 10] = p0; out[size_t(i) * 10 + 1] = p1;
  for (int j = 0; j < 8; ++j) out[size_t(i) * 10 + 2 + j] = x[j];
}

typedef unsigned __int128 u128;
static uint64_t mulmod(uint64_t a, uint64_t b) { return uint64_t((u128(a) * b) % gf::P); }

template <int V> void time_v(uint64_t* out, int blocks, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    k_time<V><<<blocks, 256>>>(out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t;
  }
  printf("  %-48s %7.3f ms  %7.2f cycles per product per wave and SIMD\n", kNames[V], best, best * 1e-3 * 2.4e9 * 1024.0 / (double(blocks) * 4 * iters * 8));
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const uint64_t P = gf::P;
  // operands that take the rare paths (products with an empty low word, all-ones high words, 2^64 - 1 = (2^32 + 1)(2^32 - 1), (P - 1)^2 ...) + random
  std::vector<uint64_t> edge = {0, 1, 2, P - 1, P - 2, P, 0xffffffffull, 0x100000000ull, 0xffffffff00000000ull, 0x8000000000000000ull, 0xfffffffeffffffffull,
                                0x00000000fffffffeull, 0x100000001ull, 0x00000001ffffffffull, 0xfffffffe00000001ull, 0x00000001fffffffeull, 0xfffffffeffffffffull,
                                0x00000000ffff0001ull, 0xffff0000ffff0001ull, 0xffffffffffffffffull};
  std::vector<uint64_t> a, b;
  for (uint64_t x : edge) for (uint64_t y : edge) { a.push_back(x); b.push_back(y); }
  uint64_t s = 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < 100000; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; a.push_back(s); s = s * 6364136223846793005ull + 1442695040888963407ull; b.push_back(s); }
  const int n = int(a.size());
  uint64_t *da, *db, *dout;
  CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dout, size_t(n) * 10 * 8));
  CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
  k_check<<<(n + 255) / 256, 256>>>(da, db, dout, n);
  std::vector<uint64_t> got(size_t(n) * 10);
  CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const uint64_t w = b[i] % P;
    const uint64_t* g = &got[size_t(i) * 10];
    if (g[0] != mulmod(a[i] % P, w) || g[1] != mulmod(a[(i + 1) % n] % P, w)) { if (bad < 5) printf("  MISMATCH pair at %d\n", i); ++bad; }
    for (int j = 0; j < 8; ++j) if (g[2 + j] != mulmod(a[(i + j) % n] % P, w)) { if (bad < 5) printf("  MISMATCH batch at %d/%d: %016llx\n", i, j, (unsigned long long)g[2 + j]); ++bad; }
  }
  printf("correctness of the pair / batch forms: %s (%d mismatches over %d operand pairs, canonical results required)\n", bad ? "FAILED" : "ok", bad, n);
  const int blocks = 1024, iters = 2000;
  uint64_t* tout; CK(hipMalloc(&tout, size_t(blocks) * 256 * 8));
  for (int w = 0; w < 600; ++w) k_time<V_SHIPPED><<<blocks, 256>>>(tout, iters);   // ~1.5 s: the clock has ramped before anything is timed
  CK(hipDeviceSynchronize());
  for (int round = 0; round < 2; ++round) {   // two rounds: the second one shows whether the order mattered
  time_v<V_SHIPPED>(tout, blocks, iters); time_v<V_UNCHECKED>(tout, blocks, iters); time_v<V_PAIR>(tout, blocks, iters); time_v<V_BATCH8>(tout, blocks, iters); time_v<V_BATCH8_LATE>(tout, blocks, iters);
  }
  return bad ? 1 : 0;
}
