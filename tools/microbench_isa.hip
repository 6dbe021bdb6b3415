// Per-instruction issue cost on gfx950 for the integer VALU opcodes the GF(P) arithmetic is made of.
// Each test runs a loop of 16 instances of one instruction on independent registers (wave64, all CUs,
// 4 or 8 waves per SIMD) and reports cycles per wave-instruction per SIMD at 2.4 GHz.
// Build: hipcc -O3 --offload-arch=gfx950 -o microbench_isa microbench_isa.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

#define REP16(X) X X X X X X X X X X X X X X X X
#define IND8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int MODE>
__global__ void __launch_bounds__(256) k_isa(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7, c = a ^ b, d = a + b;
  uint64_t p = (uint64_t(a) << 32) | b, q = (uint64_t(c) << 32) | d, r = p + q;
  uint32_t ra[8]; uint64_t rp[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ra[i] = a + i * 77u; rp[i] = p + i * 1234567ull; }
  for (int it = 0; it < iters; ++it) {
#define OP_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ra[i]) : "v"(b));
#define OP_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(ra[i]) : "v"(b) : "vcc");
#define OP_ADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(rp[i]) : "v"(q));
#define OP_CMP64(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(rp[i]), "v"(q) : "vcc");
#define OP_SHL64(i) asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(rp[i]));
#define OP_SHL32(i) asm volatile("v_lshlrev_b32 %0, 7, %0" : "+v"(ra[i]));
#define OP_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(rp[i]) : "v"(a), "v"(b) : "vcc");
#define OP_CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ra[i]) : "v"(b) : "vcc");
#define OP_PAIR(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(ra[i]), "+v"(ra[(i + 4) & 7]) : "v"(b), "v"(d) : "vcc");
    if (MODE == 20) { IND8(OP_ADD) }
    if (MODE == 21) { IND8(OP_ADDCO) }
    if (MODE == 22) { IND8(OP_ADD64) }
    if (MODE == 23) { IND8(OP_CMP64) }
    if (MODE == 24) { IND8(OP_SHL64) }
    if (MODE == 25) { IND8(OP_SHL32) }
    if (MODE == 26) { IND8(OP_MAD64) }
    if (MODE == 27) { IND8(OP_CNDM) }
    if (MODE == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (MODE == 1) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a) : "v"(b) : "vcc");) }
    if (MODE == 2) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a), "+v"(c) : "v"(b), "v"(d) : "vcc");) }
    if (MODE == 3) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p) : "v"(q));) }
    if (MODE == 4) { REP16(asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(p), "v"(q) : "vcc");) }
    if (MODE == 5) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");) }
    if (MODE == 6) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }
    if (MODE == 7) { REP16(asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(p));) }
    if (MODE == 8) { REP16(asm volatile("v_lshlrev_b32 %0, 7, %0" : "+v"(a));) }
    if (MODE == 9) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, 9" : "+v"(a) : "v"(b));) }
    if (MODE == 10) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p) : "v"(a), "v"(b) : "vcc");) }
    if (MODE == 11) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (MODE == 12) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (MODE == 13) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }
    if (MODE == 14) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (MODE == 15) { uint32_t tmp; REP16(asm volatile("v_sub_co_u32 %0, vcc, %0, %3\n\tv_subb_co_u32 %1, vcc, %1, %4, vcc\n\tv_cndmask_b32_e64 %2, 0, -1, vcc\n\tv_sub_co_u32 %0, vcc, %0, %2\n\tv_subbrev_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a), "+v"(c), "=&v"(tmp) : "v"(b), "v"(d) : "vcc");) }
    if (MODE == 16) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (MODE == 17) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (MODE == 18) { REP16(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (MODE == 19) { REP16(asm volatile("v_lshrrev_b64 %0, 9, %0" : "+v"(p));) }
  }
  for (int i = 0; i < 8; ++i) { a ^= ra[i]; p ^= rp[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ uint32_t(p) ^ uint32_t(q >> 32) ^ uint32_t(r);
}

int main() {
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"v_add_u32", "v_add_co_u32", "add_co+addc_co (pair)", "v_lshl_add_u64", "v_cmp_lt_u64", "v_cmp_lt_u32", "v_cndmask_b32",
                         "v_lshlrev_b64", "v_lshlrev_b32", "v_alignbit_b32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "cmp+cndmask (pair)",
                         "v_add3_u32", "modsub 5-instr chain", "v_mul_u32_u24", "v_mad_u32_u24", "v_xor_b32", "v_lshrrev_b64",
                         "IND v_add_u32", "IND v_add_co_u32", "IND v_lshl_add_u64", "IND v_cmp_lt_u64", "IND v_lshlrev_b64", "IND v_lshlrev_b32", "IND v_mad_u64_u32", "IND v_cndmask_b32"};
  const int per[] = {1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 5, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
  for (int blocks : {1024, 2048}) {
    uint32_t* out; CK(hipMalloc(&out, size_t(blocks) * 256 * 4));
    printf("blocks=%d (%.0f waves/SIMD)\n", blocks, blocks * 4.0 / 1024.0);
    for (int mode = 0; mode < 28; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        switch (mode) {
#define C(M) case M: k_isa<M><<<blocks, 256>>>(out, iters); break;
          C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27)
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      const double winstr = double(blocks) * 4 * iters * 16 * per[mode];   // wave-instructions
      printf("  %-24s %7.3f ms  %6.2f cycles per wave-instruction per SIMD\n", names[mode], best, best * 1e-3 * 2.4e9 * 1024.0 / winstr);
    }
    CK(hipFree(out));
  }
  return 0;
}
