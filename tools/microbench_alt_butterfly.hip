// Numbers behind "measured or costed and dropped" (DESIGN.md 5.1): alternative forms of the radix-8 butterfly of gfdft.hpp, each checked
// against gf::dft8 on random operands and timed like tools/microbench_gf.hip (8 points per thread, 256-thread blocks, 4 waves per SIMD;
// cycles per wave-operation per SIMD and POINT at 2.4 GHz, next to the shipped butterfly measured in the same process).
//
//   A  shipped: gf::dft8<false, 0> (canonical 64-bit values, 12 add + 12 sub + 5 constant shifts)
//   B  redundant limbs: a value is four signed 24-bit limbs in 32-bit registers, value = sum l_i 2^(24 i) mod 2^96 + 1 (P divides
//      2^96 + 1).  Sums and differences are four carry-less VOP2 adds, omega_8 = -2^24 is a rotation of the limbs (free, the sign goes
//      into the order of a subtraction), eight bits of head room take the three levels.  What it costs is getting in and out:
//      64-bit -> limbs (4 instructions) and limbs -> a canonical 64-bit value (~20), because the LDS exchanges, the table twiddles and
//      the general multiplier want 64-bit words.  Timed whole (in + 3 levels + out) and as the bare three levels (B-core: limbs in,
//      limbs out, re-masked to 24 bits so that the loop can run).
//   C  byte limbs through the matrix cores: the DFT-8 of eight 64-bit values is a 0 / +-1 matrix on their 64 bytes (omega_8 shifts by
//      three bytes), i.e. v_mfma_i32_*_i8 work with twelve 32-bit accumulators per output value.  Timed: ONLY the VALU re-assembly of
//      the twelve accumulators into a canonical 64-bit value (the MFMA issue, the operand layout and the bias of the signed bytes are
//      not counted) -- a lower bound of that form's VALU cost per point.
//   D  candidates for gf.hpp that keep the canonical form: mul_pow2(x, 48) through the 128-bit reduction (one instruction shorter),
//      and dft8 with the three sums that may stay un-folded (a0, c0, e0: "lazy sums where one operand is canonical").
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I prmers_amd/csrc -o tools/microbench_alt_butterfly tools/microbench_alt_butterfly.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gfdft.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// ---------------------------------------------------------------- B: four signed 24-bit limbs, mod 2^96 + 1
struct L4 { int32_t l[4]; };
__device__ __forceinline__ L4 to_limbs(uint64_t x) {
  L4 r;
  r.l[0] = int32_t(uint32_t(x) & 0xffffffu);
  r.l[1] = int32_t(uint32_t(x >> 24) & 0xffffffu);
  r.l[2] = int32_t(uint32_t(x >> 48));
  r.l[3] = 0;
  return r;
}
__device__ __forceinline__ L4 ladd(const L4& a, const L4& b) { L4 r; for (int j = 0; j < 4; ++j) r.l[j] = a.l[j] + b.l[j]; return r; }
__device__ __forceinline__ L4 lsub(const L4& a, const L4& b) { L4 r; for (int j = 0; j < 4; ++j) r.l[j] = a.l[j] - b.l[j]; return r; }
// (a - b) 2^(24 K): limb j takes the difference of limb j - K, negated when it wrapped past 2^96 = -1
template <int K> __device__ __forceinline__ L4 lsub_rot(const L4& a, const L4& b) {
  L4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = (j - K + 4) & 3;
    r.l[j] = (j < K) ? b.l[s] - a.l[s] : a.l[s] - b.l[s];
  }
  return r;
}
// limbs (|l_i| < 2^27) -> canonical value
__device__ __forceinline__ uint64_t from_limbs(const L4& v) {
  const int64_t B = int64_t(v.l[2]) + (int64_t(v.l[3]) << 24);                 // |B| < 2^52
  const uint32_t b0 = uint32_t(B) & 0xffffu, b1 = uint32_t(uint64_t(B) >> 16);
  const int32_t b2 = int32_t(B >> 48);                                         // B 2^48 = b0 2^48 + b1 2^64 + b2 2^96 = b0 2^48 + b1 EPS - b2
  const int64_t A = int64_t(v.l[0] - b2) + (int64_t(v.l[1]) << 24);            // |A| < 2^52
  uint64_t t = uint64_t(A);
#if defined(GF_ASM)
  t = gf::dev::sub_eps_if(t, uint32_t(int32_t(A >> 32) >> 31) & 0xffffu);      // negative: + P
#else
  if (A < 0) t += gf::P;
#endif
  unsigned c;
  const uint32_t hi = __builtin_addc(uint32_t(t >> 32), b0 << 16, 0u, &c);     // + b0 2^48, a carry is 2^64 = EPS (cannot carry twice: t < P)
  uint64_t u = (uint64_t(hi) << 32) | uint32_t(t);
#if defined(GF_ASM)
  u += uint64_t(c ? gf::dev::k_ones() : 0u);
  return gf::dev::mad_eps_fold(b1, u);
#else
  u += c ? gf::EPS : 0;
  return gf::add(gf::fold(u), (uint64_t(b1) << 32) - b1);
#endif
}
__device__ __forceinline__ void dft8_limbs_core(L4 (&x)[8]) {   // the network of gf::dft8<false>
  const L4 a0 = ladd(x[0], x[4]), a1 = ladd(x[1], x[5]), a2 = ladd(x[2], x[6]), a3 = ladd(x[3], x[7]);
  const L4 b0 = lsub(x[0], x[4]), b1 = lsub_rot<1>(x[5], x[1]), b2 = lsub_rot<2>(x[2], x[6]), b3 = lsub_rot<3>(x[7], x[3]);
  const L4 c0 = ladd(a0, a2), c1 = ladd(a1, a3), d0 = lsub(a0, a2), d1 = lsub_rot<2>(a1, a3);
  const L4 e0 = ladd(b0, b2), e1 = ladd(b1, b3), f0 = lsub(b0, b2), f1 = lsub_rot<2>(b1, b3);
  x[0] = ladd(c0, c1); x[4] = lsub(c0, c1); x[2] = ladd(d0, d1); x[6] = lsub(d0, d1);
  x[1] = ladd(e0, e1); x[5] = lsub(e0, e1); x[3] = ladd(f0, f1); x[7] = lsub(f0, f1);
}
__device__ __forceinline__ void dft8_limbs(uint64_t (&x)[8]) {
  L4 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = to_limbs(x[j]);
  dft8_limbs_core(v);
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = from_limbs(v[j]);
}

// ---------------------------------------------------------------- C: twelve byte-position accumulators -> canonical value
// value = sum_q acc[q] 2^(8 q) mod 2^96 + 1, |acc[q]| <= 2^11 (eight signed bytes summed)
__device__ __forceinline__ uint64_t from_byte_accumulators(const int32_t (&acc)[12]) {
  int64_t W[3];
#pragma unroll
  for (int g = 0; g < 3; ++g)
    W[g] = int64_t(acc[4 * g]) + (int64_t(acc[4 * g + 1]) << 8) + (int64_t(acc[4 * g + 2]) << 16) + (int64_t(acc[4 * g + 3]) << 24);   // |W| < 2^36
  // W0 + W1 2^32 + W2 2^64 = (W0 - W2) + (W1 + W2) 2^32
  const int64_t X = W[0] - W[2], Y = W[1] + W[2];
  const uint32_t y0 = uint32_t(Y);
  const int32_t y1 = int32_t(Y >> 32);                 // Y 2^32 = y0 2^32 + y1 2^64 = y0 2^32 + y1 EPS
  const int64_t A = X - y1;                             // + y1 2^32 below
  uint64_t t = uint64_t(A);
  if (A < 0) t += gf::P;
  t = gf::fold(t);
  const uint64_t hi = gf::fold((uint64_t(y0) << 32));
  uint64_t r = gf::add(t, hi);
  const uint64_t y1m = (y1 < 0) ? gf::P - (uint64_t(uint32_t(-y1)) << 32) : (uint64_t(uint32_t(y1)) << 32);
  return gf::add(r, y1m);
}

// ---------------------------------------------------------------- D: candidates that keep the canonical form
// x 2^48 through the 128-bit reduction: x 2^48 = top 2^96 + mid 2^64 + (h0 2^16) 2^32  ->  lo + mid EPS - top with lo = h0 2^48
__device__ __forceinline__ uint64_t mul_pow2_48_v2(uint64_t a) {
#if defined(GF_ASM)
  const uint32_t lh = uint32_t(a) << 16, mid = uint32_t(a >> 16), top = uint32_t(a >> 48);
  uint64_t bw;
  const uint64_t x = gf::dev::sub32_c(uint64_t(lh) << 32, top, 0, bw);
  return gf::dev::reduce_tail(mid, x, bw);
#else
  return gf::mul_pow2(a, 48);
#endif
}
// dft8 with a0, c0, e0 left un-folded (every one of them meets a canonical partner: add_lazy needs one canonical operand, sub a canonical
// subtrahend).  Outputs 0 .. 6 may then be any 64-bit representative: only for consumers that shift or multiply all of them.
__device__ __forceinline__ void dft8_lazier(uint64_t (&x)[8]) {
  using namespace gf;
  const uint64_t a0 = add_lazy(x[0], x[4]), a1 = add(x[1], x[5]), a2 = add(x[2], x[6]), a3 = add(x[3], x[7]);
  const uint64_t b0 = sub(x[0], x[4]);
  const uint64_t b1 = mul_pow2(sub(x[5], x[1]), 24), b2 = mul_pow2(sub(x[2], x[6]), 48), b3 = mul_pow2(sub(x[7], x[3]), 72);
  const uint64_t c0 = add_lazy(a0, a2), c1 = add(a1, a3), d0 = sub(a0, a2);
  const uint64_t d1 = mul_pow2(sub(a1, a3), 48);
  const uint64_t e0 = add_lazy(b0, b2), e1 = add(b1, b3), f0 = sub(b0, b2);
  const uint64_t f1 = mul_pow2(sub(b1, b3), 48);
  x[0] = add_lazy(c0, c1); x[4] = sub(c0, c1);
  x[2] = add_lazy(d0, d1); x[6] = sub(d0, d1);
  x[1] = add_lazy(e0, e1); x[5] = sub(e0, e1);
  x[3] = add_lazy(f0, f1); x[7] = sub(f0, f1);
}
__device__ __forceinline__ void dft8_sh48v2(uint64_t (&x)[8]) {   // the shipped network with the shorter x 2^48
  using namespace gf;
  const uint64_t a0 = add(x[0], x[4]), a1 = add(x[1], x[5]), a2 = add(x[2], x[6]), a3 = add(x[3], x[7]);
  const uint64_t b0 = sub(x[0], x[4]);
  const uint64_t b1 = mul_pow2(sub(x[5], x[1]), 24), b2 = mul_pow2_48_v2(sub(x[2], x[6])), b3 = mul_pow2(sub(x[7], x[3]), 72);
  const uint64_t c0 = add(a0, a2), c1 = add(a1, a3), d0 = sub(a0, a2);
  const uint64_t d1 = mul_pow2_48_v2(sub(a1, a3));
  const uint64_t e0 = add(b0, b2), e1 = add(b1, b3), f0 = sub(b0, b2);
  const uint64_t f1 = mul_pow2_48_v2(sub(b1, b3));
  x[0] = add(c0, c1); x[4] = sub(c0, c1); x[2] = add(d0, d1); x[6] = sub(d0, d1);
  x[1] = add(e0, e1); x[5] = sub(e0, e1); x[3] = add(f0, f1); x[7] = sub(f0, f1);
}

enum { V_SHIPPED, V_SHIPPED_LAZY2, V_LIMBS, V_LIMBS_CORE, V_BYTES, V_LAZIER, V_SH48V2, V_SH48, V_SH48_NEW, V_COUNT };
static const char* kNames[V_COUNT] = {"A  shipped dft8, per point", "A  shipped dft8<LAZY = 2> (four lazy output sums), per point", "B  limb ring: in + 3 levels + out, per point", "B  limb ring: 3 levels + re-mask only, per point",
                                      "C  byte accumulators -> value (VALU only), per point", "D  dft8, three more lazy sums, per point",
                                      "D  dft8 with the shorter x 2^48, per point", "   mul_pow2(x, 48) shipped", "   mul_pow2(x, 48) through reduce_tail"};

template <int V>
__global__ void __launch_bounds__(256) k_time(uint64_t* out, int iters) {
  uint64_t x[8], y = (uint64_t(threadIdx.x) * 0x9e3779b97f4a7c15ull + blockIdx.x) % gf::P;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (y * (2 * i + 3) + i) % gf::P;
  if (V == V_SHIPPED) { for (int it = 0; it < iters; ++it) gf::dft8<false, 0>(x); }
  else if (V == V_LIMBS) { for (int it = 0; it < iters; ++it) dft8_limbs(x); }
  else if (V == V_SHIPPED_LAZY2) { for (int it = 0; it < iters; ++it) gf::dft8<false, 2>(x); }   // (timing only: un-folded sums re-enter the loop)
  else if (V == V_LAZIER) { for (int it = 0; it < iters; ++it) dft8_lazier(x); }                  // (the same)
  else if (V == V_SH48V2) { for (int it = 0; it < iters; ++it) dft8_sh48v2(x); }
  else if (V == V_SH48) { for (int it = 0; it < iters; ++it) { for (int i = 0; i < 8; ++i) x[i] = gf::mul_pow2(x[i], 48); } }
  else if (V == V_SH48_NEW) { for (int it = 0; it < iters; ++it) { for (int i = 0; i < 8; ++i) x[i] = mul_pow2_48_v2(x[i]); } }
  else if (V == V_LIMBS_CORE) {
    L4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = to_limbs(x[j]);
    for (int it = 0; it < iters; ++it) {
      dft8_limbs_core(v);
#pragma unroll
      for (int j = 0; j < 8; ++j) for (int q = 0; q < 4; ++q) v[j].l[q] &= 0xffffff;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = uint64_t(uint32_t(v[j].l[0] ^ v[j].l[1])) | (uint64_t(uint32_t(v[j].l[2] ^ v[j].l[3])) << 32);
  } else if (V == V_BYTES) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int32_t acc[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q] = int32_t(uint32_t(x[(i + q) & 7]) >> (20 + q)) - 1024;   // (stand-in for MFMA outputs: two cheap VOP2 each, subtracted below)
        x[i] = from_byte_accumulators(acc);
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_check(const uint64_t* a, uint64_t* out, int n) {   // n groups of 8 canonical values
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t r[8], l[8], z[8], s[8];
  for (int j = 0; j < 8; ++j) r[j] = l[j] = z[j] = s[j] = a[size_t(i) * 8 + j];
  gf::dft8<false, 0>(r); dft8_limbs(l); dft8_lazier(z); dft8_sh48v2(s);
  for (int j = 0; j < 8; ++j) { out[size_t(i) * 40 + j] = r[j]; out[size_t(i) * 40 + 8 + j] = l[j]; out[size_t(i) * 40 + 16 + j] = z[j]; out[size_t(i) * 40 + 24 + j] = s[j]; }
  // byte accumulators: the digits of a[8i] in base 2^8 with signed perturbations that cancel
  int32_t acc[12];
  const uint64_t v = a[size_t(i) * 8];
  for (int q = 0; q < 12; ++q) acc[q] = (q < 8) ? int32_t((v >> (8 * q)) & 0xff) : 0;
  acc[3] += 1024; acc[4] -= 4;           // 1024 2^24 = 4 2^32
  acc[11] += 7; acc[10] -= 7 * 256;      // 7 2^88 = 7 256 2^80
  acc[11] += 512; acc[0] += 2;           // 512 2^88 = 2 2^96 = -2 (the wrap of 2^96 + 1)
  acc[6] -= 300; acc[5] += 300 * 256 - 65536; acc[7] += 1; acc[5] -= 0;   // -300 2^48 + (76800 - 65536) 2^40 + 2^56 ... = 0: 76800 2^40 = 300 2^48, 65536 2^40 = 2^56
  out[size_t(i) * 40 + 32] = from_byte_accumulators(acc);
  out[size_t(i) * 40 + 33] = mul_pow2_48_v2(a[size_t(i) * 8 + 1]);
  out[size_t(i) * 40 + 34] = mul_pow2_48_v2(~a[size_t(i) * 8 + 2]);   // any 64-bit operand
}

typedef unsigned __int128 u128;
static uint64_t mulmod(uint64_t a, uint64_t b) { return uint64_t((u128(a) * b) % gf::P); }
static uint64_t pow2mod(unsigned s) { uint64_t r = 1; for (unsigned i = 0; i < s; ++i) r = mulmod(r, 2); return r; }

template <int V> void time_v(uint64_t* out, int blocks, int iters, double per_iter_ops, double subtract_cycles = 0) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    k_time<V><<<blocks, 256>>>(out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t;
  }
  const double cyc = best * 1e-3 * 2.4e9 * 1024.0 / (double(blocks) * 4 * iters * per_iter_ops) - subtract_cycles;
  printf("  %-58s %7.3f ms  %7.2f cycles per wave-op per SIMD\n", kNames[V], best, cyc);
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const uint64_t P = gf::P;
  const int n = 40000;
  std::vector<uint64_t> a(size_t(n) * 8);
  uint64_t s = 0x9e3779b97f4a7c15ull;
  const uint64_t edge[8] = {0, 1, P - 1, P - 2, 0xffffffffull, 0xffffffff00000000ull, 0x0000ffffffffffffull, 0x8000000000000000ull};
  for (size_t i = 0; i < a.size(); ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    a[i] = (i < 512) ? edge[(i * 7 + i / 8) & 7] : s % P;
  }
  uint64_t *da, *dout;
  CK(hipMalloc(&da, a.size() * 8)); CK(hipMalloc(&dout, size_t(n) * 40 * 8));
  CK(hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice));
  k_check<<<(n + 255) / 256, 256>>>(da, dout, n);
  std::vector<uint64_t> got(size_t(n) * 40);
  CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  const uint64_t p48 = pow2mod(48);
  for (int i = 0; i < n; ++i) {
    const uint64_t* g = &got[size_t(i) * 40];
    for (int j = 0; j < 8; ++j) {
      if (g[j] % P != g[8 + j] || g[8 + j] >= P) { if (bad < 5) printf("  MISMATCH limb ring group %d out %d: %016llx vs %016llx\n", i, j, (unsigned long long)g[8 + j], (unsigned long long)g[j]); ++bad; }
      if (g[j] % P != g[16 + j] % P) { if (bad < 5) printf("  MISMATCH lazier group %d out %d\n", i, j); ++bad; }
      if (g[j] % P != g[24 + j] % P || g[24 + j] > P) { if (bad < 5) printf("  MISMATCH sh48v2 group %d out %d\n", i, j); ++bad; }
    }
    if (g[32] != a[size_t(i) * 8] % P) { if (bad < 5) printf("  MISMATCH byte accumulators group %d: %016llx vs %016llx\n", i, (unsigned long long)g[32], (unsigned long long)a[size_t(i) * 8]); ++bad; }
    if (g[33] % P != mulmod(a[size_t(i) * 8 + 1], p48) || g[33] > P) { if (bad < 5) printf("  MISMATCH mul_pow2_48_v2 %d\n", i); ++bad; }
    if (g[34] % P != mulmod((~a[size_t(i) * 8 + 2]) % P, p48) || g[34] > P) { if (bad < 5) printf("  MISMATCH mul_pow2_48_v2 (any operand) %d\n", i); ++bad; }
  }
  printf("correctness: %s (%d mismatches over %d groups of eight)\n", bad ? "FAILED" : "ok", bad, n);

  const int blocks = 1024, iters = 2000;
  uint64_t* tout; CK(hipMalloc(&tout, size_t(blocks) * 256 * 8));
  time_v<V_SHIPPED>(tout, blocks, iters, 8);
  time_v<V_SHIPPED_LAZY2>(tout, blocks, iters, 8);
  time_v<V_LIMBS>(tout, blocks, iters, 8);
  time_v<V_LIMBS_CORE>(tout, blocks, iters, 8);
  time_v<V_BYTES>(tout, blocks, iters, 8, 12 * 2 * 2.7);   // minus the stand-in's 24 VOP2 instructions per value (2.7 cycles each, profiles/r02_microbench_isa2.txt)
  time_v<V_LAZIER>(tout, blocks, iters, 8);
  time_v<V_SH48V2>(tout, blocks, iters, 8);
  time_v<V_SH48>(tout, blocks, iters, 8);
  time_v<V_SH48_NEW>(tout, blocks, iters, 8);
  return bad ? 1 : 0;
}
