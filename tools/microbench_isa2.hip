// Issue cost of further gfx950 VALU opcodes (round 2 catalogue; complements microbench_isa.hip): which integer
// forms run at the 2-cycle rate of v_add_u32 and which at the 4-cycle rate of the carry / shift / multiply forms.
// 16 independent instances per loop iteration (8 registers x 2), 256-thread blocks, 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/microbench_isa2 tools/microbench_isa2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#define IND16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define OPS(X) \
  X(0, "v_add_u32", asm volatile("v_add_u32 %0, %0, %1" : "+v"(ra[i]) : "v"(b));) \
  X(1, "v_sub_u32", asm volatile("v_sub_u32 %0, %0, %1" : "+v"(ra[i]) : "v"(b));) \
  X(2, "v_and_b32", asm volatile("v_and_b32 %0, %0, %1" : "+v"(ra[i]) : "v"(b));) \
  X(3, "v_or_b32", asm volatile("v_or_b32 %0, %0, %1" : "+v"(ra[i]) : "v"(b));) \
  X(4, "v_not_b32", asm volatile("v_not_b32 %0, %0" : "+v"(ra[i]));) \
  X(5, "v_mov_b32", asm volatile("v_mov_b32 %0, %1" : "+v"(ra[i]) : "v"(b));) \
  X(6, "v_min_u32", asm volatile("v_min_u32 %0, %0, %1" : "+v"(ra[i]) : "v"(b));) \
  X(7, "v_max_u32", asm volatile("v_max_u32 %0, %0, %1" : "+v"(ra[i]) : "v"(b));) \
  X(8, "v_cndmask_b32 (sgpr mask)", asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(ra[i]) : "v"(b), "s"(mask));) \
  X(9, "v_cndmask_b32 0,-1 (sgpr mask)", asm volatile("v_cndmask_b32 %0, 0, -1, %1" : "=v"(ra[i]) : "s"(mask));) \
  X(10, "v_bfe_u32", asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(ra[i]));) \
  X(11, "v_bfi_b32", asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(ra[i]) : "v"(b), "v"(c));) \
  X(12, "v_perm_b32", asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(ra[i]) : "v"(b), "v"(sel));) \
  X(13, "v_lshl_or_b32", asm volatile("v_lshl_or_b32 %0, %0, 5, %1" : "+v"(ra[i]) : "v"(b));) \
  X(14, "v_and_or_b32", asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(ra[i]) : "v"(b), "v"(c));) \
  X(15, "v_or3_b32", asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(ra[i]) : "v"(b), "v"(c));) \
  X(16, "v_add_lshl_u32", asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(ra[i]) : "v"(b));) \
  X(17, "v_lshl_add_u32", asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(ra[i]) : "v"(b));) \
  X(18, "v_lshrrev_b32", asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(ra[i]));) \
  X(19, "v_ashrrev_i32", asm volatile("v_ashrrev_i32 %0, 5, %0" : "+v"(ra[i]));) \
  X(20, "v_alignbyte_b32", asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(ra[i]) : "v"(b));) \
  X(21, "v_bitop3_b32", asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ra[i]) : "v"(b), "v"(c));) \
  X(22, "v_xad_u32", asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(ra[i]) : "v"(b), "v"(c));) \
  X(23, "v_mad_i64_i32", asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(rp[i]) : "v"(a), "v"(b) : "vcc");) \
  X(24, "v_mad_u64_u32 x,-1", asm volatile("v_mad_u64_u32 %0, vcc, %1, -1, %0" : "+v"(rp[i]) : "v"(a) : "vcc");) \
  X(25, "v_mov_b64", asm volatile("v_mov_b64 %0, %1" : "+v"(rp[i]) : "v"(q));) \
  X(26, "v_pk_add_u16", asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(ra[i]) : "v"(b));) \
  X(27, "v_sub_co_u32", asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(ra[i]) : "v"(b) : "vcc");) \
  X(28, "v_addc_co_u32 (vcc)", asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(ra[i]) : "v"(b) : "vcc");) \
  X(29, "v_addc_co_u32 (sgpr in/out)", asm volatile("v_addc_co_u32 %0, %1, %0, %2, %1" : "+v"(ra[i]), "+s"(mask2) : "v"(b));) \
  X(30, "v_cmp_eq_u32 -> sgpr", asm volatile("v_cmp_eq_u32 %0, %1, %2" : "=s"(mask2) : "v"(ra[i]), "v"(b));) \
  X(31, "v_cmp_lt_u64 -> sgpr", asm volatile("v_cmp_lt_u64 %0, %1, %2" : "=s"(mask2) : "v"(rp[i]), "v"(q));) \
  X(32, "v_lshl_add_u64", asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(rp[i]) : "v"(q));) \
  X(33, "v_add_co_u32 + s_nop 1 + v_addc_co_u32", asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(ra[i]), "+v"(rb[i]) : "v"(b), "v"(c) : "vcc");) \
  X(34, "v_add_co_u32 + v_addc_co_u32 (no nop; timing only)", asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(ra[i]), "+v"(rb[i]) : "v"(b), "v"(c) : "vcc");) \
  X(35, "v_add_f64", asm volatile("v_add_f64 %0, %0, %1" : "+v"(rd[i]) : "v"(dq));) \
  X(36, "v_fma_f64", asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(rd[i]) : "v"(dq));) \
  X(37, "v_mul_hi_u32", asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(ra[i]) : "v"(b));) \
  X(38, "v_lshlrev_b32", asm volatile("v_lshlrev_b32 %0, 7, %0" : "+v"(ra[i]));) \
  X(39, "v_alignbit_b32", asm volatile("v_alignbit_b32 %0, %0, %1, 9" : "+v"(ra[i]) : "v"(b));) \
  X(40, "v_add_u32 + v_add_co_u32 alternating", asm volatile("v_add_u32 %0, %0, %2\n\tv_add_co_u32 %1, vcc, %1, %2" : "+v"(ra[i]), "+v"(rb[i]) : "v"(b) : "vcc");) \
  X(41, "v_add_u32 + v_mad_u64_u32 alternating", asm volatile("v_add_u32 %0, %0, %2\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(ra[i]), "+v"(rp[i]) : "v"(b), "v"(c) : "vcc");) \
  X(42, "v_add3_u32", asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(ra[i]) : "v"(b), "v"(c));) \
  X(43, "v_sub_u32 sdwa-free e64 (sgpr src)", asm volatile("v_sub_u32 %0, %0, %1" : "+v"(ra[i]) : "s"(sb));) \
  X(44, "v_cmp_eq_u32_e32 -> vcc (VOPC)", asm volatile("v_cmp_eq_u32_e32 vcc, %0, %1" : : "v"(ra[i]), "v"(b) : "vcc");) \
  X(45, "v_cmp_lt_u64_e32 -> vcc (VOPC)", asm volatile("v_cmp_lt_u64_e32 vcc, %0, %1" : : "v"(rp[i]), "v"(q) : "vcc");) \
  X(46, "v_cmp_lt_u64_e32 vcc + v_cndmask_e32 (pair)", asm volatile("v_cmp_lt_u64_e32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(ra[i]) : "v"(rp[i]), "v"(q), "v"(b) : "vcc");) \
  X(47, "v_cmp_lt_u64_e64 sgpr + v_cndmask_e64 (pair)", asm volatile("v_cmp_lt_u64 %1, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %4, %1" : "+v"(ra[i]), "=&s"(mask2) : "v"(rp[i]), "v"(q), "v"(b));)

static const int kPer[] = {1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,2,2,1,1,1,1,1,2,2,1,1,1,1,2,2};

template <int MODE>
__global__ void __launch_bounds__(256) k_isa(uint32_t* out, int iters, uint32_t sb) {
  uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7, c = a ^ b, sel = 0x02050104;
  uint64_t q = (uint64_t(c) << 32) | (a + b);
  double dq = 1.000001;
  uint64_t mask = 0x5555aaaa3333ccccull, mask2 = 0;
  uint32_t ra[8], rb[8]; uint64_t rp[8]; double rd[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ra[i] = a + i * 77u; rb[i] = b + i; rp[i] = q + i * 1234567ull; rd[i] = 1.0 + i; }
  asm volatile("" : "+s"(mask));
  for (int it = 0; it < iters; ++it) {
#define X(M, NAME, CODE) if (MODE == M) { 
#define ENDX }
#define OPI(i) 
    ;
#undef X
#define X(M, NAME, CODE) if (MODE == M) { _Pragma("unroll") for (int rep = 0; rep < 2; ++rep) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { CODE } } }
    OPS(X)
#undef X
  }
  uint32_t acc = uint32_t(mask2);
  for (int i = 0; i < 8; ++i) { acc ^= ra[i] ^ rb[i] ^ uint32_t(rp[i]) ^ uint32_t(rp[i] >> 32) ^ uint32_t(rd[i]); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE> float run(uint32_t* out, int blocks, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    k_isa<MODE><<<blocks, 256>>>(out, iters, 12345u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int iters = 4000;
  for (int blocks : {512, 1024}) {
    uint32_t* out; CK(hipMalloc(&out, size_t(blocks) * 256 * 4));
    printf("blocks=%d (%.0f waves/SIMD)\n", blocks, blocks * 4.0 / 1024.0);
#define X(M, NAME, CODE) { float ms = run<M>(out, blocks, iters); const double w = double(blocks) * 4 * iters * 16 * kPer[M]; \
      printf("  %-52s %7.3f ms  %6.2f cycles per wave-instruction per SIMD\n", NAME, ms, ms * 1e-3 * 2.4e9 * 1024.0 / w); }
    OPS(X)
#undef X
    CK(hipFree(out));
  }
  return 0;
}
