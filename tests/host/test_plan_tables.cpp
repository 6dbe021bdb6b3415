// Host check of the plan's weight factorisation (plan.hpp): for every digit j of a few exponents / forced
// shapes, the digit width equals ceil(p(j+1)/n) - ceil(pj/n), the factored weight TA*TB/(wrap?2:1) equals
// 2^(((n - pj mod n) mod n)/n) for both splits (odd digits also through the second half of SA/TA with
// TB[2 i2]), TAi carries 1/(m TA), and the digit-info words (DI) of the register-resident column kernels hold
// exactly those widths and wrap flags.  Prints OK or the first mismatches.
#include <cstdio>
#include "plan.hpp"
using namespace mi355;
static bool wrap_of(uint64_t sa, uint64_t sb, uint64_t n) { return sa > 0 && sb > 0 && sa + sb <= n; }
int main() {
  struct Case { uint32_t p; const char* spec; };
  const Case cases[] = {{127, nullptr}, {9941, nullptr}, {3997, nullptr}, {300007, "m2=8,c=4"}, {300007, "m2=16,c=8"},
                        {300007, "m2=4,c=2"}, {600011, "m2=32,c=8"}, {102701, nullptr},
                        {132049, "m2=16,c=4"},                              // columns of 256 x 4 (kernels_v3.hip: 256 threads, one run each)
                        {400063, "m2=8,c=4"}, {800283, "m2=8,c=2"}};        // radix-5 columns 1280 x 4 and 2560 x 2 (kernels_v5.hip: 640 threads, runs i1 = t + 640 d1)
  int bad = 0;
  for (const Case& cs : cases) {
    const Plan pl = make_plan(cs.p, cs.spec, true);
    const uint64_t n = pl.n, p = cs.p, r = gf::root_of_two(n), inv_m = gf::inv(uint64_t(pl.m) % gf::P);
    size_t di_checked = 0;
    for (uint64_t j = 0; j < n; ++j) {
      const uint64_t s = (p * j) % n, ceil0 = (p * j + n - 1) / n, ceil1 = (p * (j + 1) + n - 1) / n;
      const uint32_t width = uint32_t(ceil1 - ceil0);
      const uint64_t w = gf::pow(r, (n - s) % n);
      const uint64_t i = j >> 1, b = j & 1, i1 = i / pl.M2, i2 = i % pl.M2;
      if (pl.width(j) != width) { if (bad++ < 5) printf("width p=%u j=%llu\n", cs.p, (unsigned long long)j); }
      // split 1: SA[i1] + SB[2 i2 + b]
      uint64_t f = gf::mul(pl.TA[i1], pl.TB[2 * i2 + b]);
      if (wrap_of(pl.SA[i1], pl.SB[2 * i2 + b], n)) f = gf::half(f);
      if (f != w) { if (bad++ < 5) printf("weight p=%u j=%llu\n", cs.p, (unsigned long long)j); }
      // split 2 (odd digits): SA[M1 + i1] + SB[2 i2]
      const uint64_t sa2 = b ? pl.SA[pl.M1 + i1] : pl.SA[i1], ta2 = b ? pl.TA[pl.M1 + i1] : pl.TA[i1];
      const uint64_t tai2 = b ? pl.TAi[pl.M1 + i1] : pl.TAi[i1];
      const bool wr2 = wrap_of(sa2, pl.SB[2 * i2], n);
      uint64_t f2 = gf::mul(ta2, pl.TB[2 * i2]);
      if (wr2) f2 = gf::half(f2);
      if (f2 != w || (sa2 + pl.SB[2 * i2]) % n != s) { if (bad++ < 5) printf("weight2 p=%u j=%llu\n", cs.p, (unsigned long long)j); }
      if (gf::mul(gf::mul(tai2, ta2), uint64_t(pl.m) % gf::P) != 1 || gf::mul(pl.TBi[2 * i2], pl.TB[2 * i2]) != 1 || inv_m == 0) {
        if (bad++ < 5) printf("inverse p=%u j=%llu\n", cs.p, (unsigned long long)j);
      }
      if (!pl.DI.empty()) {
        const uint32_t TP = (pl.r5 == 5) ? 640u : (pl.M1 == 256 ? 256u : 512u);   // threads per tile of the kernel set that reads the words
        const uint32_t C = pl.C, ND = 2 * C, T = uint32_t(i2 / C), c = uint32_t(i2 % C), t = uint32_t(i1 % TP), d1 = uint32_t(i1 / TP);
        const uint32_t bits = (pl.DI[size_t(T) * TP + t] >> (2 * (d1 * ND + 2 * c + b))) & 3u;
        if ((bits & 1u) != width - pl.q || ((bits >> 1) != 0) != wr2) { if (bad++ < 5) printf("DI p=%u j=%llu\n", cs.p, (unsigned long long)j); }
        ++di_checked;
      }
    }
    printf("p=%u %s digits=%llu di_checked=%zu\n", cs.p, pl.describe().c_str(), (unsigned long long)n, di_checked);
  }
  printf(bad ? "FAILED %d\n" : "OK\n", bad);
  return bad != 0;
}
