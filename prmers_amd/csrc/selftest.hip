// Device self-test of the GF(P) primitives (gf.hpp, gfdft.hpp): the device code paths differ from the host
// ones (borrow-reusing sub, P left for a negated zero), so they are checked on the GPU itself against 128-bit
// host arithmetic, modulo P, on edge values and random operands.  Reached through mi355_engine_selftest().
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "gfdft.hpp"

namespace mi355 {

namespace {
constexpr int kOps = 6 + 192;   // add, sub, mul, add_lazy, fold, mul_u32, then mul_pow2 for every shift

__global__ void k_selftest(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t x = a[i], y = b[i];
  uint64_t* o = out + size_t(i) * kOps;
  o[0] = gf::add(x, y);
  o[1] = gf::sub(x, y);
  o[2] = gf::mul(x, y);
  o[3] = gf::add_lazy(x, y);
  o[4] = gf::fold(x + y);                 // any 64-bit value
  o[5] = gf::mul_u32(x, uint32_t(y));
  for (unsigned s = 0; s < 192; ++s) o[6 + s] = gf::mul_pow2(x, s);   // runtime s: every branch of mul_pow2
}

__global__ void k_selftest_dft8(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t f[8], g[8], h[8];
  for (int j = 0; j < 8; ++j) f[j] = g[j] = h[j] = in[size_t(i) * 8 + j];
  gf::dft8<false, 0>(f); gf::dft8<true, 1>(g); gf::dft8<false, 2>(h);
  for (int j = 0; j < 8; ++j) { out[size_t(i) * 24 + j] = f[j]; out[size_t(i) * 24 + 8 + j] = g[j]; out[size_t(i) * 24 + 16 + j] = h[j]; }
}

typedef unsigned __int128 u128;
uint64_t mulmod(uint64_t a, uint64_t b) { return uint64_t((u128(a) * b) % gf::P); }
void chk(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string("selftest: ") + what + ": " + hipGetErrorString(e)); }
}  // namespace

// throws std::runtime_error with the first mismatch
void selftest_primitives(int device) {
  chk(hipSetDevice(device), "hipSetDevice");
  const uint64_t P = gf::P;
  std::vector<uint64_t> edge = {0, 1, 2, P - 1, P - 2, P, 0xffffffffull, 0x100000000ull, 0xffffffff00000000ull, 0x8000000000000000ull,
                                0xfffffffeffffffffull, 0x00000000fffffffeull, 0x123456789abcdef0ull % P,
                                // operands whose products take the rare paths of gf::mul's tail: 2^64 - 1 = (2^32 + 1)(2^32 - 1) (low half >= P, no
                                // carry), (P - 1)^2 (borrow out of lo - hh - c), products with an empty low word or an all-ones high word
                                0x100000001ull, 0x00000001ffffffffull, 0xfffffffe00000001ull, 0x0000000100000000ull + 0xfffffffeull, 0xffffffff00000000ull - 1,
                                0x00000000ffff0001ull, 0xffff0000ffff0001ull};
  std::vector<uint64_t> a, b;
  for (uint64_t x : edge) for (uint64_t y : edge) { a.push_back(x); b.push_back(y); }
  uint64_t s = 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < 4096; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; a.push_back(s % P); s = s * 6364136223846793005ull + 1442695040888963407ull; b.push_back(s % P); }
  const int n = int(a.size());
  uint64_t *da, *db, *dout;
  chk(hipMalloc(reinterpret_cast<void**>(&da), n * 8), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&db), n * 8), "hipMalloc");
  chk(hipMalloc(reinterpret_cast<void**>(&dout), size_t(n) * kOps * 8), "hipMalloc");
  chk(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice), "copy"); chk(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice), "copy");
  hipLaunchKernelGGL(k_selftest, dim3((n + 63) / 64), dim3(64), 0, 0, da, db, dout, n);
  std::vector<uint64_t> out(size_t(n) * kOps);
  chk(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost), "kernel / copy back");
  std::string err;
  auto fail = [&](const char* op, int i, unsigned sft) {
    if (err.empty()) { char buf[160]; std::snprintf(buf, sizeof buf, "%s mismatch: a=%016llx b=%016llx s=%u", op, (unsigned long long)a[i], (unsigned long long)b[i], sft); err = buf; }
  };
  for (int i = 0; i < n; ++i) {
    const uint64_t x = a[i], y = b[i], xm = x % P, ym = y % P;   // operands may be P itself (the lazy zero)
    const uint64_t* o = &out[size_t(i) * kOps];
    if (o[0] % P != uint64_t((u128(xm) + ym) % P) || o[0] > P) fail("add", i, 0);
    if (o[1] % P != uint64_t((u128(xm) + P - ym) % P) || o[1] > P) fail("sub", i, 0);
    if (o[2] != mulmod(xm, ym)) fail("mul", i, 0);
    if (o[3] % P != uint64_t((u128(xm) + ym) % P)) fail("add_lazy", i, 0);
    if (o[4] != (x + y) % P) fail("fold", i, 0);
    if (o[5] != mulmod(xm, uint32_t(y))) fail("mul_u32", i, 0);
    for (unsigned sft = 0; sft < 192; ++sft) {
      const uint64_t r = o[6 + sft], want = mulmod(xm, gf::pow(2, sft));
      if (r % P != want || (sft != 0 && r > P)) fail("mul_pow2", i, sft);
    }
  }
  // butterflies: canonical and LAZY variants against the host's canonical dft8 (same source, host path)
  const int nd = 512;
  std::vector<uint64_t> din(size_t(nd) * 8), dres(size_t(nd) * 24);
  for (size_t i = 0; i < din.size(); ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; din[i] = (i < 64) ? edge[i % edge.size()] % P : s % P; }
  uint64_t *dd, *dr;
  chk(hipMalloc(reinterpret_cast<void**>(&dd), din.size() * 8), "hipMalloc"); chk(hipMalloc(reinterpret_cast<void**>(&dr), dres.size() * 8), "hipMalloc");
  chk(hipMemcpy(dd, din.data(), din.size() * 8, hipMemcpyHostToDevice), "copy");
  hipLaunchKernelGGL(k_selftest_dft8, dim3((nd + 63) / 64), dim3(64), 0, 0, dd, dr, nd);
  chk(hipMemcpy(dres.data(), dr, dres.size() * 8, hipMemcpyDeviceToHost), "dft8 kernel / copy back");
  for (int i = 0; i < nd && err.empty(); ++i) {
    uint64_t f[8], g[8];
    for (int j = 0; j < 8; ++j) f[j] = g[j] = din[size_t(i) * 8 + j];
    gf::dft8<false, 0>(f); gf::dft8<true, 0>(g);
    for (int j = 0; j < 8; ++j)
      if (dres[size_t(i) * 24 + j] % P != f[j] || dres[size_t(i) * 24 + 8 + j] % P != g[j] || dres[size_t(i) * 24 + 16 + j] % P != f[j]) err = "dft8 mismatch";
  }
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout); (void)hipFree(dd); (void)hipFree(dr);
  if (!err.empty()) throw std::runtime_error("selftest: " + err);
}

}  // namespace mi355
