// Cost of a grid barrier on MI355X (8 XCDs, non-coherent L2s) against a kernel boundary: what a one-launch squaring of the small
// transforms (kernels.hip k_coop) has to beat.  Variants: (0) flag array, sc1 polling, no L2 maintenance; (1) the same with agent-scope
// release / acquire fences (buffer_wbl2 / buffer_inv); (2) one atomic counter; (3) empty kernels back to back (the boundary itself).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_gridsync.hip -o tools/microbench_gridsync && tools/microbench_gridsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) k_bar(uint32_t* flags, uint32_t* counter, uint32_t epoch0, uint32_t rounds, uint64_t* sink) {
  const uint32_t G = gridDim.x;
  uint32_t epoch = epoch0;
  for (uint32_t r = 0; r < rounds; ++r) {
    ++epoch;
    __syncthreads();
    if (threadIdx.x < 64) {
      if (MODE == 2) {
        if (threadIdx.x == 0) {
          __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while (int32_t(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch * G) < 0) __builtin_amdgcn_s_sleep(1);
        }
      } else {
        if (threadIdx.x == 0) {
          if (MODE == 1) __threadfence();
          __hip_atomic_store(&flags[blockIdx.x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (;;) {
          uint32_t f[8]; bool here = true;
#pragma unroll
          for (int k = 0; k < 8; ++k) { const uint32_t g = threadIdx.x + 64u * k; f[k] = g < G ? __hip_atomic_load(&flags[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch; }
#pragma unroll
          for (int k = 0; k < 8; ++k) here = here && int32_t(f[k] - epoch) >= 0;
          if (__all(here)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (MODE == 1) __threadfence();
      }
    }
    __syncthreads();
  }
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = epoch;
}
__global__ void k_empty(uint64_t* sink) { if (sink && threadIdx.x == 5000) *sink = 1; }

int main() {
  uint32_t *flags, *counter; uint64_t* sink;
  CK(hipMalloc(&flags, 4096 * 4)); CK(hipMalloc(&counter, 256)); CK(hipMalloc(&sink, 8));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t rounds = 2000;
  for (uint32_t G : {64u, 128u, 256u, 512u}) {
    for (uint32_t T : {256u, 1024u}) {
      for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemsetAsync(flags, 0, 4096 * 4, s)); CK(hipMemsetAsync(counter, 0, 256, s));
        uint32_t epoch0 = 0; uint32_t rr = rounds; uint64_t* sk = sink;
        void* args[5] = {&flags, &counter, &epoch0, &rr, &sk};
        const void* fn = mode == 0 ? (const void*)k_bar<0> : mode == 1 ? (const void*)k_bar<1> : (const void*)k_bar<2>;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0, s));
          CK(hipLaunchCooperativeKernel(fn, dim3(G), dim3(T), args, 0, s));
          CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
          epoch0 += rounds;
        }
        std::printf("groups %4u threads %4u mode %d (%s): %.3f us per barrier\n", G, T, mode, mode == 0 ? "flags, sc1 only" : mode == 1 ? "flags + agent fences" : "atomic counter", ms * 1000.0 / rounds);
      }
    }
  }
  // kernel boundary: back-to-back launches of an empty kernel, plain and cooperative
  for (int coop = 0; coop < 2; ++coop) {
    const int N = 2000; float ms = 0;
    uint64_t* sk = nullptr; void* args[1] = {&sk};
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) {
        if (coop) CK(hipLaunchCooperativeKernel((const void*)k_empty, dim3(256), dim3(1024), args, 0, s));
        else hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, s, sk);
      }
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::printf("%s launches of an empty 256 x 1024 kernel back to back: %.3f us each\n", coop ? "cooperative" : "plain", ms * 1000.0 / N);
  }
  return 0;
}
