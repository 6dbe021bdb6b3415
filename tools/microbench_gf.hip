// Throughput and bit-exactness of the GF(P) primitives (prmers_amd/csrc/gf.hpp, gfdft.hpp) on gfx950.
// Each timing kernel runs 8 independent dependency chains per thread, 256-thread blocks, 4 waves per SIMD; the
// report is cycles per wave-operation per SIMD at 2.4 GHz.  Every op is first checked against 128-bit host
// arithmetic on edge + random operands.  Build twice to compare the device forms:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I prmers_amd/csrc -o tools/microbench_gf tools/microbench_gf.hip
//   hipcc ... -DGF_PORTABLE_DEVICE -o tools/microbench_gf_portable tools/microbench_gf.hip     (round-1 forms)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gfdft.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

enum { OP_MUL, OP_MULU32, OP_ADD, OP_SUB, OP_ADD_LAZY, OP_P2_24, OP_P2_48, OP_P2_72, OP_P2_39, OP_P2_57, OP_P2_177, OP_DFT8, OP_DFT8I, OP_COUNT };
static const char* kNames[OP_COUNT] = {"mul", "mul_u32", "add", "sub", "add_lazy", "mul_pow2 24", "mul_pow2 48", "mul_pow2 72", "mul_pow2 39", "mul_pow2 57",
                                       "mul_pow2 177", "dft8 forward, per point", "dft8 inverse, per point"};

template <int OP>
__device__ __forceinline__ uint64_t apply(uint64_t x, uint64_t y) {
  switch (OP) {
    case OP_MUL: return gf::mul(x, y);
    case OP_MULU32: return gf::mul_u32(x, uint32_t(y));
    case OP_ADD: return gf::add(x, y);
    case OP_SUB: return gf::sub(x, y);
    case OP_ADD_LAZY: return gf::add_lazy(x, y);
    case OP_P2_24: return gf::mul_pow2(x, 24);
    case OP_P2_48: return gf::mul_pow2(x, 48);
    case OP_P2_72: return gf::mul_pow2(x, 72);
    case OP_P2_39: return gf::mul_pow2(x, 39);
    case OP_P2_57: return gf::mul_pow2(x, 57);
    case OP_P2_177: return gf::mul_pow2(x, 177);
    default: return x;
  }
}

template <int OP>
__global__ void __launch_bounds__(256) k_time(uint64_t* out, int iters) {
  uint64_t x[8], y = (uint64_t(threadIdx.x) * 0x9e3779b97f4a7c15ull + blockIdx.x) % gf::P;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (y * (2 * i + 3) + i) % gf::P;
  if (OP == OP_DFT8) {
    for (int it = 0; it < iters; ++it) gf::dft8<false, 0>(x);
  } else if (OP == OP_DFT8I) {
    for (int it = 0; it < iters; ++it) gf::dft8<true, 0>(x);
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = apply<OP>(x[i], y);
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_check(const uint64_t* a, const uint64_t* b, uint64_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t x = a[i], y = b[i];
  uint64_t* o = out + size_t(i) * 8;
  o[0] = gf::mul(x, y); o[1] = gf::mul_u32(x, uint32_t(y)); o[2] = gf::add(x, y); o[3] = gf::sub(x, y); o[4] = gf::add_lazy(x, y); o[5] = gf::fold(x);
  o[6] = gf::sqr(x); o[7] = gf::dbl(x);
}
__global__ void k_check_shift(const uint64_t* a, uint64_t* out, int n) {   // every shift, runtime s
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (unsigned s = 0; s < 192; ++s) out[size_t(i) * 192 + s] = gf::mul_pow2(a[i], s);
}

typedef unsigned __int128 u128;
static uint64_t mulmod(uint64_t a, uint64_t b) { return uint64_t((u128(a) * b) % gf::P); }
static uint64_t pow2mod(unsigned s) { uint64_t r = 1; for (unsigned i = 0; i < s; ++i) r = mulmod(r, 2); return r; }

template <int OP> void time_op(uint64_t* out, int blocks, int iters, float* ms) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    k_time<OP><<<blocks, 256>>>(out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t;
  }
  *ms = best;
  printf("  %-28s %7.3f ms  %6.2f cycles per wave-op per SIMD\n", kNames[OP], best, best * 1e-3 * 2.4e9 * 1024.0 / (double(blocks) * 4 * iters * 8));
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const uint64_t P = gf::P;
  std::vector<uint64_t> edge = {0, 1, 2, P - 1, P - 2, P, P + 1, 0xffffffffull, 0x100000000ull, 0x100000001ull, 0xffffffff00000000ull, 0x8000000000000000ull,
                                0xfffffffeffffffffull, 0x00000000fffffffeull, 0xffffffffffffffffull, 0xfffffffffffffffeull, 0x0000ffff00000000ull, 0xffff000000000000ull,
                                0x00000001ffffffffull, 0x123456789abcdef0ull % P};
  std::vector<uint64_t> a, b;
  for (uint64_t x : edge) for (uint64_t y : edge) { a.push_back(x); b.push_back(y); }
  uint64_t s = 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < 200000; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull; uint64_t x = s;
    s = s * 6364136223846793005ull + 1442695040888963407ull; uint64_t y = s;
    if (i % 4 == 1) x >>= 32;
    if (i % 4 == 2) y >>= 32;
    if (i % 8 == 7) x |= 0xffffffff00000000ull;
    a.push_back(x); b.push_back(y);
  }
  const int n = int(a.size());
  uint64_t *da, *db, *dout;
  CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dout, size_t(20000) * 192 * 8 + size_t(n) * 8 * 8));
  CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
  int bad = 0;
  {
    k_check<<<(n + 255) / 256, 256>>>(da, db, dout, n);
    std::vector<uint64_t> got(size_t(n) * 8);
    CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
    auto report = [&](const char* what, uint64_t x, uint64_t y, uint64_t g, uint64_t want) {
      if (bad < 10) printf("  MISMATCH %s: x=%016llx y=%016llx got %016llx want %016llx\n", what, (unsigned long long)x, (unsigned long long)y, (unsigned long long)g, (unsigned long long)want);
      ++bad;
    };
    for (int i = 0; i < n; ++i) {
      const uint64_t x = a[i], y = b[i], xr = x % P, yr = y % P;
      const uint64_t* g = &got[size_t(i) * 8];
      // multiplications accept any 64-bit representative and return a canonical value
      if (g[0] >= P || g[0] != mulmod(xr, yr)) report("mul", x, y, g[0], mulmod(xr, yr));
      if (g[1] >= P || g[1] != mulmod(xr, uint32_t(y))) report("mul_u32", x, y, g[1], mulmod(xr, uint32_t(y)));
      if (g[6] >= P || g[6] != mulmod(xr, xr)) report("sqr", x, x, g[6], mulmod(xr, xr));
      if (x <= P && y <= P) {   // add / sub / add_lazy / dbl: operands <= P (P stands for a negated zero)
        const uint64_t sum = uint64_t((u128(xr) + yr) % P), dif = uint64_t((u128(xr) + P - yr) % P);
        if (g[2] > P || g[2] % P != sum || (x < P && y < P && g[2] >= P)) report("add", x, y, g[2], sum);   // P + P stays P (a negated zero)
        if (g[3] > P || g[3] % P != dif || (x < P && g[3] >= P)) report("sub", x, y, g[3], dif);
        if (g[4] % P != sum) report("add_lazy", x, y, g[4], sum);
        if (g[7] > P || g[7] % P != uint64_t((u128(xr) * 2) % P) || (x < P && g[7] >= P)) report("dbl", x, x, g[7], 0);
      }
      if (g[5] >= P || g[5] != xr) report("fold", x, 0, g[5], xr);
    }
  }
  {   // every shift of every operand (any 64-bit value)
    const int ns = 20000;
    k_check_shift<<<(ns + 255) / 256, 256>>>(da, dout, ns);
    std::vector<uint64_t> got(size_t(ns) * 192);
    CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> p2(192); for (unsigned sft = 0; sft < 192; ++sft) p2[sft] = pow2mod(sft);
    for (int i = 0; i < ns; ++i) for (unsigned sft = 0; sft < 192; ++sft) {
      const uint64_t g = got[size_t(i) * 192 + sft], want = mulmod(a[i] % P, p2[sft]);
      const bool ok = (sft == 0) ? (g == a[i]) : (g <= P && g % P == want);
      if (!ok) { if (bad < 10) printf("  MISMATCH mul_pow2: x=%016llx s=%u got %016llx want %016llx\n", (unsigned long long)a[i], sft, (unsigned long long)g, (unsigned long long)want); ++bad; }
    }
  }
  printf("correctness: %s (%d mismatches over %d operand pairs)\n", bad ? "FAILED" : "ok", bad, n);

  const int blocks = 1024, iters = 2000;   // 4 waves per SIMD
  uint64_t* tout; CK(hipMalloc(&tout, size_t(blocks) * 256 * 8));
  float ms;
  time_op<OP_MUL>(tout, blocks, iters, &ms); time_op<OP_MULU32>(tout, blocks, iters, &ms); time_op<OP_ADD>(tout, blocks, iters, &ms); time_op<OP_SUB>(tout, blocks, iters, &ms);
  time_op<OP_ADD_LAZY>(tout, blocks, iters, &ms); time_op<OP_P2_24>(tout, blocks, iters, &ms); time_op<OP_P2_48>(tout, blocks, iters, &ms); time_op<OP_P2_72>(tout, blocks, iters, &ms);
  time_op<OP_P2_39>(tout, blocks, iters, &ms); time_op<OP_P2_57>(tout, blocks, iters, &ms); time_op<OP_P2_177>(tout, blocks, iters, &ms);
  time_op<OP_DFT8>(tout, blocks, iters, &ms); time_op<OP_DFT8I>(tout, blocks, iters, &ms);
  return bad ? 1 : 0;
}
