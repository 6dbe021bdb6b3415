#include <cstdio>
#include <cstdlib>
#include "gfdft.hpp"
typedef unsigned __int128 u128;
static uint64_t ref_mul(uint64_t a, uint64_t b){ return (uint64_t)(((u128)a*b) % gf::P); }
int main(){
  uint64_t vals[] = {0,1,2,gf::P-1,gf::P-2,0xffffffffull,0x100000000ull,0xfffffffeffffffffull,0x8000000000000000ull,0xffffffff00000000ull,12345678901234567ull};
  srand(1);
  int bad=0;
  for (int it=0; it<20000; ++it){
    uint64_t a = it < 11 ? vals[it] : ((((uint64_t)rand()<<42) ^ ((uint64_t)rand()<<21) ^ rand()) % gf::P);
    for (unsigned s=0; s<192; ++s){
      uint64_t p2 = gf::pow(2, s);
      if (gf::mul_pow2(a, s) != ref_mul(a, p2)) { if (bad++<5) printf("mul_pow2 bad a=%llx s=%u\n",(unsigned long long)a,s); }
    }
    uint32_t b = (uint32_t)rand()*2654435761u;
    if (gf::mul_u32(a,b) != ref_mul(a,b)) { if (bad++<5) printf("mul_u32 bad\n"); }
    uint64_t c = ((((uint64_t)rand()<<42) ^ ((uint64_t)rand()<<21) ^ rand()) % gf::P);
    if (gf::mul(a,c) != ref_mul(a,c)) { if (bad++<5) printf("mul bad\n"); }
  }
  // dft8
  const uint64_t w8 = gf::root_of_unity(8), w8i = gf::inv(w8);
  if (gf::pow(2, gf::LOG2_W64) != gf::root_of_unity(64)) { printf("LOG2_W64 wrong\n"); bad++; }
  for (int it=0; it<2000; ++it){
    uint64_t x[8], y[8], z[8];
    for (int j=0;j<8;++j) x[j] = ((((uint64_t)rand()<<42) ^ ((uint64_t)rand()<<21) ^ rand()) % gf::P);
    if (it==0) for (int j=0;j<8;++j) x[j]=gf::P-1;
    for (int k=0;k<8;++k){ uint64_t s=0, si=0; for(int j=0;j<8;++j){ s=gf::add(s, ref_mul(x[j], gf::pow(w8,(uint64_t)(j*k)%8))); si=gf::add(si, ref_mul(x[j], gf::pow(w8i,(uint64_t)(j*k)%8))); } y[k]=s; z[k]=si; }
    uint64_t f[8], g[8]; for(int j=0;j<8;++j){f[j]=x[j]; g[j]=x[j];}
    gf::dft8<false>(f); gf::dft8<true>(g);
    for(int k=0;k<8;++k){ if (f[k]!=y[k]) { if(bad++<5) printf("dft8 fwd bad k=%d\n",k);} if (g[k]!=z[k]) { if(bad++<5) printf("dft8 inv bad k=%d\n",k);} }
  }
  // lazy forms: add_lazy is congruent for every operand pair <= P (incl. the tail cases), fold brings any
  // 64-bit representative back to [0, P), and the LAZY butterflies agree with the canonical ones after a fold;
  // mul / mul_u32 / mul_pow2 accept any 64-bit representative (values in [P, 2^64) included)
  {
    const uint64_t P = gf::P;
    const uint64_t edge[] = {0, 1, P - 1, P, P - 2, 0xffffffffull, 0xffffffff00000000ull, 0x8000000000000000ull, 0xfffffffefffffffeull};
    for (uint64_t a : edge) for (uint64_t b : edge) {
      const uint64_t l = gf::add_lazy(a, b);
      if ((u128)l % P != ((u128)a + b) % P) { if (bad++<5) printf("add_lazy bad %llx %llx\n",(unsigned long long)a,(unsigned long long)b); }
      if (gf::fold(l) != (uint64_t)(((u128)a + b) % P)) { if (bad++<5) printf("fold bad\n"); }
      if (a <= P && b < P && gf::add(a, b) % P != (uint64_t)(((u128)a + b) % P)) { if (bad++<5) printf("add with P operand bad\n"); }
      if (a <= P && b <= P && gf::sub(a, b) % P != (uint64_t)(((u128)a + P + P - b) % P)) { if (bad++<5) printf("sub with P operand bad\n"); }
    }
    const uint64_t tail[] = {P, P + 1, 0xffffffffffffffffull, 0xffffffff80000000ull};
    for (uint64_t a : tail) {
      for (unsigned s = 0; s < 192; ++s)
        if (gf::mul_pow2(a, s) % P != ref_mul(a % P, gf::pow(2, s))) { if (bad++<5) printf("mul_pow2 non-canonical operand bad s=%u\n", s); }
      if (gf::mul(a, 0x123456789abcdefull) != ref_mul(a % P, 0x123456789abcdefull)) { if (bad++<5) printf("mul non-canonical operand bad\n"); }
    }
    for (int it = 0; it < 500; ++it) {
      uint64_t x[8], f0[8], f1[8], f2[8];
      for (int j = 0; j < 8; ++j) { x[j] = ((((uint64_t)rand()<<42) ^ ((uint64_t)rand()<<21) ^ rand()) % P); if (it < 8 && j == it) x[j] = P - 1; f0[j] = f1[j] = f2[j] = x[j]; }
      gf::dft8<false, 0>(f0); gf::dft8<false, 1>(f1); gf::dft8<true, 2>(f2);
      uint64_t g0[8]; for (int j = 0; j < 8; ++j) g0[j] = x[j];
      gf::dft8<true, 0>(g0);
      for (int k = 0; k < 8; ++k) {
        if (gf::fold(f1[k]) != f0[k]) { if (bad++<5) printf("dft8 LAZY=1 bad\n"); }
        if (gf::fold(f2[k]) != g0[k]) { if (bad++<5) printf("dft8 LAZY=2 bad\n"); }
      }
    }
  }
  printf(bad? "FAIL %d\n":"OK %d\n", bad); return bad!=0;
}
