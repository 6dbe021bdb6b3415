// Micro-benchmarks that size the design (run on the GPU box, results quoted in DESIGN.md):
//   1. GF(P) multiply / add / shift-multiply throughput per chip (is the squaring ALU- or HBM-bound?)
//   2. raw v_mad_u64_u32 issue rate
//   3. streaming copy bandwidth for a cache-resident (64 MiB) and an HBM-resident (1 GiB) buffer
//   4. "column tile" access: runs of R bytes at a large power-of-two stride, read-modify-write in place
// Build: hipcc -O3 --offload-arch=gfx950 -o microbench microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../prmers_amd/csrc/gf.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_alu(uint64_t* out, int iters, uint64_t seed) {
  uint64_t x[8], c[8];
  const uint64_t tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    x[i] = (seed * (tid + 1) * (2 * i + 3)) % gf::P;
    c[i] = (seed * 0x9E3779B97F4A7C15ull * (tid + 7) * (i + 1)) % gf::P;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = gf::mul(x[i], c[i]);
      if (MODE == 1) { x[i] = gf::add(x[i], c[i]); c[i] = gf::sub(c[i], x[i]); }
      if (MODE == 2) { x[i] = gf::mul_pow2(x[i], 24); }
      if (MODE == 3) { x[i] = gf::mul_pow2(x[i], 48); }
      if (MODE == 4) { x[i] = gf::mul_pow2(x[i], 72); }
      if (MODE == 5) {  // raw mad_u64_u32 chain
        x[i] = (uint64_t)(uint32_t)c[i] * (uint32_t)(x[i] >> 7) + x[i];
      }
      if (MODE == 6) {  // mul_lo + mul_hi
        uint32_t a = (uint32_t)x[i], b = (uint32_t)c[i];
        x[i] = (uint64_t)(a * b) ^ ((uint64_t)__umulhi(a + 1, b) << 20);
      }
      if (MODE == 7) x[i] = gf::sqr(x[i]);
      if (MODE == 8) x[i] = gf::mul_u32(x[i], (uint32_t)c[i]);
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= x[i] ^ c[i];
  out[tid] = s;
}

__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) {
    uint4 v = src[i];
    v.x += 1;
    dst[i] = v;
  }
}

// Column-tile pattern: buffer = rows x rowlen16 (uint4 units).  A workgroup owns a column group of
// RUN16 adjacent uint4 (RUN16*16 bytes) and touches it in all `rows` rows (stride rowlen16), in place.
// threads: 256; each wave-load covers (64/RUN16) rows x RUN16 columns.
template <int RUN16>
__global__ void __launch_bounds__(256) k_coltile(uint4* __restrict__ buf, int rows, size_t rowlen16) {
  const size_t col0 = (size_t)blockIdx.x * RUN16;
  const int c = threadIdx.x % RUN16;
  const int r0 = threadIdx.x / RUN16;
  constexpr int RSTEP = 256 / RUN16;
  for (int r = r0; r < rows; r += RSTEP * 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int rr = r + u * RSTEP;
      if (rr < rows) v[u] = buf[(size_t)rr * rowlen16 + col0 + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int rr = r + u * RSTEP;
      if (rr < rows) { v[u].x += 1; buf[(size_t)rr * rowlen16 + col0 + c] = v[u]; }
    }
  }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  // ---- ALU ----
  {
    const int blocks = getenv("MB_BLOCKS") ? atoi(getenv("MB_BLOCKS")) : 256 * 8, threads = 256, iters = 2000;
    printf("ALU section: blocks=%d (%.1f waves/SIMD)\n", blocks, blocks * 4.0 / 1024.0);
    uint64_t* out; CK(hipMalloc(&out, (size_t)blocks * threads * 8));
    const char* names[] = {"gf::mul", "gf::add+sub", "mul_pow2(24)", "mul_pow2(48)", "mul_pow2(72)", "mad_u64_u32", "mul_lo+mul_hi", "gf::sqr", "gf::mul_u32"};
    for (int mode = 0; mode < 9; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        switch (mode) {
          case 0: k_alu<0><<<blocks, threads>>>(out, iters, 12345); break;
          case 1: k_alu<1><<<blocks, threads>>>(out, iters, 12345); break;
          case 2: k_alu<2><<<blocks, threads>>>(out, iters, 12345); break;
          case 3: k_alu<3><<<blocks, threads>>>(out, iters, 12345); break;
          case 4: k_alu<4><<<blocks, threads>>>(out, iters, 12345); break;
          case 5: k_alu<5><<<blocks, threads>>>(out, iters, 12345); break;
          case 6: k_alu<6><<<blocks, threads>>>(out, iters, 12345); break;
          case 7: k_alu<7><<<blocks, threads>>>(out, iters, 12345); break;
          case 8: k_alu<8><<<blocks, threads>>>(out, iters, 12345); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = time_ms(e0, e1); if (ms < best) best = ms;
      }
      double ops = (double)blocks * threads * iters * 8 * (mode == 1 ? 2 : 1);
      printf("ALU %-14s: %8.3f ms  %8.2f Gop/s  (%.2f cycles/wave-op/SIMD @2.4GHz)\n", names[mode], best,
             ops / best * 1e-6, 1024.0 * 2.4e9 / (ops / 64.0 / (best * 1e-3)));
    }
    CK(hipFree(out));
  }

  if (getenv("MB_ALU_ONLY")) return 0;
  // ---- copy ----
  for (size_t mib : {64, 1024}) {
    size_t bytes = mib << 20, n16 = bytes / 16;
    uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    for (int blocks : {2048, 8192}) {
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        k_copy<<<blocks, 256>>>(a, b, n16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = time_ms(e0, e1); if (ms < best) best = ms;
      }
      printf("COPY %4zu MiB blocks=%5d: %8.3f ms  %8.1f GB/s (R+W)\n", mib, blocks, best, 2.0 * bytes / best * 1e-6);
    }
    // in-place RMW
    {
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        k_copy<<<8192, 256>>>(a, a, n16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = time_ms(e0, e1); if (ms < best) best = ms;
      }
      printf("RMW  %4zu MiB in place   : %8.3f ms  %8.1f GB/s (R+W)\n", mib, best, 2.0 * bytes / best * 1e-6);
    }
    CK(hipFree(a)); CK(hipFree(b));
  }

  // ---- column tiles on a 64 MiB buffer (2^22 uint4) and a 1 GiB buffer ----
  for (size_t mib : {64, 1024}) {
    size_t bytes = mib << 20, n16 = bytes / 16;
    uint4* a; CK(hipMalloc(&a, bytes)); CK(hipMemset(a, 1, bytes));
    for (int rows : {256, 1024, 2048}) {
      size_t rowlen16 = n16 / rows;
#define RUNCASE(R)                                                                                   \
      {                                                                                              \
        int blocks = (int)(rowlen16 / R);                                                            \
        float best = 1e30f;                                                                          \
        for (int rep = 0; rep < 5; ++rep) {                                                          \
          CK(hipEventRecord(e0));                                                                    \
          k_coltile<R><<<blocks, 256>>>(a, rows, rowlen16);                                          \
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                       \
          float ms = time_ms(e0, e1); if (ms < best) best = ms;                                      \
        }                                                                                            \
        printf("COLTILE %4zu MiB rows=%4d run=%4d B blocks=%7d: %8.3f ms %8.1f GB/s (R+W)\n", mib, rows, R * 16, blocks, best, 2.0 * bytes / best * 1e-6); \
      }
      RUNCASE(4) RUNCASE(8) RUNCASE(16) RUNCASE(32) RUNCASE(64)
    }
    CK(hipFree(a));
  }
  return 0;
}
