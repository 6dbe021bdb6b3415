// Work-buffer layout for transforms beyond the Infinity Cache (n >= 5 2^22: W = 160 MiB and more).  The column sweeps touch, for every row of a
// tile, a run of 16 C bytes (C = 2: 32 bytes) at a stride of one row (64 KiB); the row sweep streams whole rows.  This measures both access
// shapes on a row-major buffer and on a blocked one, W[i2 / B][row][i2 % B] (B pairs of 16 bytes), with the caches flushed by a 1 GiB stream
// between launches so that the data really comes from / goes to HBM.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_wlayout.hip -o tools/microbench_wlayout && tools/microbench_wlayout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ size_t widx(uint32_t row, uint32_t i2, uint32_t M1, uint32_t M2, uint32_t lb) {   // lb = log2 B; 31: row-major
  if (lb == 31) return size_t(row) * M2 + i2;
  return ((size_t(i2 >> lb) * M1 + row) << lb) + (i2 & ((1u << lb) - 1));
}
// column tile: C adjacent columns x all rows, 512 threads, 8 elements per thread in flight; MODE 0 read, 1 write
template <int C, int MODE>
__global__ void __launch_bounds__(512) k_cols(uint4* __restrict__ W, uint32_t M1, uint32_t M2, uint32_t lb, uint4* __restrict__ sink) {
  const uint32_t nb = gridDim.x, b = blockIdx.x;
  const uint32_t T = (nb % 8 == 0) ? (b & 7) * (nb >> 3) + (b >> 3) : b;   // XCD-contiguous tile order
  const uint32_t c = threadIdx.x % C, r0 = threadIdx.x / C;
  constexpr uint32_t RS = 512 / C;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t r = r0; r < M1; r += RS * 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t rr = r + u * RS;
      if (MODE == 0) { if (rr < M1) v[u] = W[widx(rr, T * C + c, M1, M2, lb)]; } else v[u] = make_uint4(rr, c, T, 1);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t rr = r + u * RS;
      if (rr >= M1) continue;
      if (MODE == 0) { acc.x += v[u].x; acc.y ^= v[u].y; } else W[widx(rr, T * C + c, M1, M2, lb)] = v[u];
    }
  }
  if (MODE == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}
// row tile: one row of M2 = 4096 pairs, 512 threads, element 512 j + t; MODE 0 read, 1 write, 2 read + write in place
template <int MODE>
__global__ void __launch_bounds__(512) k_rows(uint4* __restrict__ W, uint32_t M1, uint32_t M2, uint32_t lb, uint4* __restrict__ sink) {
  const uint32_t row = blockIdx.x, t = threadIdx.x;
  uint4 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { if (MODE != 1) v[j] = W[widx(row, 512 * j + t, M1, M2, lb)]; else v[j] = make_uint4(row, t, j, 1); }
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (MODE == 0) { acc.x += v[j].x; acc.y ^= v[j].y; }
    else { v[j].x += 1; W[widx(row, 512 * j + t, M1, M2, lb)] = v[j]; }
  }
  if (MODE == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}
__global__ void k_flush(uint4* __restrict__ f, size_t n16) {
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x) { uint4 v = f[i]; v.x += 1; f[i] = v; }
}

int main() {
  const uint32_t M1 = 2560, M2 = 4096;
  const size_t bytes = size_t(M1) * M2 * 16, fbytes = size_t(1) << 30;
  uint4 *W, *F, *sink;
  CK(hipMalloc(&W, bytes)); CK(hipMalloc(&F, fbytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(W, 1, bytes)); CK(hipMemset(F, 1, fbytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto launch, bool flush) -> float {
    float best = 1e30f, sum = 0; int n = 0;
    for (int rep = 0; rep < 5; ++rep) {
      if (flush) k_flush<<<2048, 256>>>(F, fbytes / 16);
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep) { sum += ms; ++n; if (ms < best) best = ms; }
    }
    return sum / n;
  };
  std::printf("W = %u x %u pairs (%.0f MiB); times in us, average of 4; GB/s of useful bytes\n", M1, M2, bytes / 1048576.0);
  for (int flush = 0; flush < 2; ++flush) {
    for (uint32_t lb : {31u, 1u, 2u, 3u, 4u, 5u, 6u}) {
      const float r2 = timeit([&] { k_cols<2, 0><<<M2 / 2, 512>>>(W, M1, M2, lb, sink); }, flush);
      const float w2 = timeit([&] { k_cols<2, 1><<<M2 / 2, 512>>>(W, M1, M2, lb, sink); }, flush);
      const float r4 = timeit([&] { k_cols<4, 0><<<M2 / 4, 512>>>(W, M1, M2, lb, sink); }, flush);
      const float w4 = timeit([&] { k_cols<4, 1><<<M2 / 4, 512>>>(W, M1, M2, lb, sink); }, flush);
      const float rr = timeit([&] { k_rows<0><<<M1, 512>>>(W, M1, M2, lb, sink); }, flush);
      const float rw = timeit([&] { k_rows<1><<<M1, 512>>>(W, M1, M2, lb, sink); }, flush);
      const float rm = timeit([&] { k_rows<2><<<M1, 512>>>(W, M1, M2, lb, sink); }, flush);
      char name[32];
      if (lb == 31) std::snprintf(name, sizeof name, "row-major"); else std::snprintf(name, sizeof name, "blocked B=%u (%u B)", 1u << lb, 16u << lb);
      std::printf("%s %-20s cols C=2: read %6.1f (%5.0f) write %6.1f (%5.0f) | cols C=4: read %6.1f write %6.1f | rows: read %6.1f (%5.0f) write %6.1f rmw %6.1f (%5.0f)\n",
                  flush ? "flushed " : "resident", name, r2 * 1e3, bytes / r2 * 1e-6, w2 * 1e3, bytes / w2 * 1e-6, r4 * 1e3, w4 * 1e3, rr * 1e3, bytes / rr * 1e-6,
                  rw * 1e3, rm * 1e3, 2.0 * bytes / rm * 1e-6);
    }
  }
  return 0;
}
