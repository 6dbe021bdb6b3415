// Does the row pitch of the work buffer matter?  The column sweeps touch W[row][i2 .. i2 + C) for every row of a tile: runs of 16 C bytes at a
// stride of one row, a power of two (64 KiB at C3, 128 KiB with rows of 8192).  This measures read-only, write-only and read-modify-write
// column tiles with the row pitch padded by 0 .. 1 KiB, plain and XCD-contiguous tile order.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_pitch.hip -o tools/microbench_pitch && tools/microbench_pitch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int RUN16, int MODE>   // MODE 0 read, 1 write, 2 read-modify-write
__global__ void __launch_bounds__(512) k_tile(uint4* __restrict__ buf, int rows, size_t pitch16, int xcd, uint4* __restrict__ sink) {
  const uint32_t nb = gridDim.x;
  const uint32_t T = (xcd && nb % 8 == 0) ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const size_t col0 = size_t(T) * RUN16;
  const int c = threadIdx.x % RUN16, r0 = threadIdx.x / RUN16;
  constexpr int RSTEP = 512 / RUN16;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int r = r0; r < rows; r += RSTEP * 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + u * RSTEP;
      if (MODE != 1) { if (rr < rows) v[u] = buf[size_t(rr) * pitch16 + col0 + c]; }
      else v[u] = make_uint4(rr, c, T, 1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + u * RSTEP;
      if (rr >= rows) continue;
      if (MODE == 0) { acc.x += v[u].x; acc.y ^= v[u].y; }
      else { v[u].x += 1; buf[size_t(rr) * pitch16 + col0 + c] = v[u]; }
    }
  }
  if (MODE == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

template <int RUN16>
int run_shape(const char* name, int rows, size_t rowlen16) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint4* sink; CK(hipMalloc(&sink, 64));
  for (int pad16 : {0, 4, 8, 16, 17, 32, 64}) {
    const size_t pitch16 = rowlen16 + pad16, bytes = size_t(rows) * pitch16 * 16, useful = size_t(rows) * rowlen16 * 16;
    uint4* a; CK(hipMalloc(&a, bytes)); CK(hipMemset(a, 1, bytes));
    const int blocks = int(rowlen16 / RUN16);
    for (int xcd = 0; xcd < 2; ++xcd) {
      float best[3] = {1e30f, 1e30f, 1e30f};
      for (int rep = 0; rep < 6; ++rep) {
        for (int mode = 0; mode < 3; ++mode) {
          CK(hipEventRecord(e0));
          if (mode == 0) k_tile<RUN16, 0><<<blocks, 512>>>(a, rows, pitch16, xcd, sink);
          else if (mode == 1) k_tile<RUN16, 1><<<blocks, 512>>>(a, rows, pitch16, xcd, sink);
          else k_tile<RUN16, 2><<<blocks, 512>>>(a, rows, pitch16, xcd, sink);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best[mode]) best[mode] = ms;
        }
      }
      std::printf("%-22s run %3d B pad %4d B %s: read %7.1f GB/s (%.1f us)  write %7.1f GB/s (%.1f us)  rmw %7.1f GB/s\n", name, RUN16 * 16, pad16 * 16,
                  xcd ? "xcd-order" : "plain    ", useful / best[0] * 1e-6, best[0] * 1e3, useful / best[1] * 1e-6, best[1] * 1e3, 2.0 * useful / best[2] * 1e-6);
    }
    CK(hipFree(a));
  }
  CK(hipFree(sink));
  return 0;
}

int main() {
  if (run_shape<4>("C3 1024 x 4096", 1024, 4096)) return 1;            // 64 MiB, runs of 64 bytes, stride 64 KiB
  if (run_shape<4>("5*2^22 1280 x 8192", 1280, 8192)) return 1;        // 160 MiB, stride 128 KiB
  if (run_shape<2>("2^25 2048 x 8192", 2048, 8192)) return 1;          // 256 MiB, runs of 32 bytes
  if (run_shape<2>("2^24 2048 x 4096", 2048, 4096)) return 1;
  if (run_shape<8>("2^22 512 x 4096", 512, 4096)) return 1;
  return 0;
}
