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
  printf(bad? "FAIL %d\n":"OK %d\n", bad); return bad!=0;
}
