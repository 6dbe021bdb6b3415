// Transform plan for one exponent: sizes, tile geometry and the small host-built tables.
//
// Host-only (no HIP), so the size policy and the index maps are testable without a GPU.
//
// Reference behaviour kept:  n = ibdwt::transform_size(p) (include/marin/ibdwt.h:17-43), digit
// widths ceil(p(j+1)/n) - ceil(pj/n) (ibdwt.h:127-132), weights 2^frac(-pj/n) built from
// nr2 = 554^((P-1)/192/n) (ibdwt.h:116-143), roots from generator 7 (arith.h:72).
// What is different by design (MI355X-first):
//   * the length-m (m = n/2 pairs) transform is a two-level decomposition m = M1 x M2 ("columns" of
//     length M1 at stride M2, then contiguous rows of length M2) so one squaring is three sweeps
//     (front, middle, back) instead of the reference's five transform kernels + two carry kernels;
//   * the weights are never stored per digit: w_j = TA[i1] * TB[2*i2+b] / (wrap ? 2 : 1) with
//     M1 + 2*M2 table entries (the reference streams a 2n-word table, 128 MiB at n = 2^23);
//   * registers hold UNWEIGHTED digits as u32 in a tile-major order chosen so that both the front
//     and the back sweep touch them fully coalesced.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "gf.hpp"

namespace mi355 {

inline size_t transform_size(uint32_t exponent) {
  // ibdwt.h:17-43: smallest n in {2^k, 5*2^k}, k <= 26, with 2*(floor(p/n)+1) + log2(n) < 64
  uint32_t w = 0, log2_n = 1, log2_n5 = 2;
  do { ++log2_n; w = exponent >> log2_n; } while ((w + 1) * 2 + log2_n >= 64);
  do { ++log2_n5; w = exponent / (5u << log2_n5); } while ((w + 1) * 2 + (log2_n5 + 2.4) >= 64);
  const size_t invalid = size_t(-1);
  const size_t n2 = (log2_n <= 26) ? (size_t(1) << log2_n) : invalid;
  const size_t n5 = (log2_n5 <= 26) ? (size_t(5) << log2_n5) : invalid;
  return n2 < n5 ? n2 : n5;
}

inline int ilog2(size_t v) { int r = -1; while (v) { v >>= 1; ++r; } return r; }

inline uint32_t bitrev(uint32_t i, int bits) {
  uint32_t r = 0;
  for (int k = 0; k < bits; ++k) { r = (r << 1) | (i & 1); i >>= 1; }
  return r;
}

struct Plan {
  uint32_t p = 0;
  size_t n = 0, m = 0;      // digits, pairs
  uint32_t r5 = 1;          // 1 or 5
  uint32_t M1 = 1, M2 = 1;  // m = M1 * M2, M2 = 2^logM2, M1 = r5 * L1, L1 = 2^logL1
  uint32_t L1 = 1, logL1 = 0, logM2 = 0;
  uint32_t C = 1;           // adjacent columns per front/back tile (runs of 2C digits)
  bool split5 = false;      // columns of 5 L1 pairs beyond LDS: the radix-5 stage runs through memory (kernels.hip k_front_split_*), C = 1
  uint32_t q = 0, t = 0;    // p = q*n + t
  uint32_t twh = 0;         // omega_m^e = TWlo[e & (2^twh-1)] * TWhi[e >> twh]
  size_t lds_front = 0, lds_mid = 0;

  // host tables (uploaded as-is)
  std::vector<uint32_t> SA, SB;          // p*j mod n split: SA[i1] + SB[2*i2+b]
  std::vector<uint64_t> TA, TAi, TB, TBi;  // 2^(1/n) powers; TAi carries the 1/m factor
  std::vector<uint64_t> TWlo, TWhi;      // two-level omega_m table
  std::vector<uint64_t> UT1, UT2;        // omega_M1^e (e < M1), omega_M2^e (e < M2)
  // seam twiddles of the radix-8 kernels, laid out [b][k1][k2] so that a thread's 8 words are contiguous (64 B):
  // S2r[b*64+ka] = omega_4096^(ka*b) (rows of 4096 = 64 x 64), S1r[b*(M1/64)+ka] = omega_M1^(ka*b) (columns M1 = 512/1024/2048 = (M1/64) x 64); *i = inverses
  std::vector<uint64_t> S2r, S2ri, S1r, S1ri;
  // register-resident column kernels: one word per (tile, thread) with 2 bits per digit of the thread's 16
  // digits (run d1 = 0..R-1 of i1 = 512 d1 + t, digit k of the run at bits 2 (d1 2C + k)): width - q, wrap
  std::vector<uint32_t> DI;
  uint64_t I4 = 0, I4inv = 0;            // omega_4, omega_4^-1 (forward root convention)
  uint64_t W5[5] = {1, 0, 0, 0, 0}, W5i[5] = {1, 0, 0, 0, 0};
  uint64_t W5c[4] = {0, 0, 0, 0};        // 5-point DFT constants {beta, k1, k2-k1, k1+k2} (kernels.hip dft5)

  size_t tiles() const { return M2 / C; }
  size_t runs() const { return tiles() * M1; }

  // width of digit with s = p*j mod n   (derivation: DESIGN.md "digit widths on the fly")
  uint32_t width_of_s(uint64_t s) const {
    return q + ((s + t > 0) ? 1u : 0u) + ((s + t > n) ? 1u : 0u) - ((s > 0) ? 1u : 0u);
  }
  uint32_t width(size_t j) const { return width_of_s((uint64_t(p) * j) % n); }

  // memory position (u32 index) of natural digit j inside a register (tile-major)
  size_t pos(size_t j) const {
    const size_t i = j >> 1, b = j & 1;
    const size_t i1 = i / M2, i2 = i % M2;
    const size_t T = i2 / C, c = i2 % C;
    return ((T * M1 + i1) * C + c) * 2 + b;
  }

  std::string describe() const {
    char buf[160];
    std::snprintf(buf, sizeof buf, "marin-hip:n=%zu:m1=%u:m2=%u:c=%u%s", n, M1, M2, C, split5 ? ":split5" : "");
    return buf;
  }
};

// spec: "" (auto) or comma/colon separated "m2=<pow2>", "c=<pow2>"
inline Plan make_plan(uint32_t p, const char* spec = nullptr, bool build_tables = true) {
  if (p < 3) throw std::runtime_error("exponent must be >= 3");
  Plan pl;
  pl.p = p;
  pl.n = transform_size(p);
  if (pl.n == size_t(-1)) throw std::runtime_error("exponent too large for the Goldilocks IBDWT");
  pl.m = pl.n / 2;
  pl.r5 = (pl.m % 5 == 0) ? 5 : 1;
  const int k = ilog2(pl.m / pl.r5);

  long want_m2 = -1, want_c = -1;
  bool want_split = false;
  if (spec && *spec) {
    std::string s(spec);
    for (size_t i = 0; i < s.size();) {
      size_t e = s.find_first_of(",:; ", i);
      if (e == std::string::npos) e = s.size();
      const std::string tok = s.substr(i, e - i);
      if (tok.rfind("m2=", 0) == 0) want_m2 = std::atol(tok.c_str() + 3);
      else if (tok.rfind("c=", 0) == 0) want_c = std::atol(tok.c_str() + 2);
      else if (tok == "split5") want_split = true;   // tests: the split column sweeps at a small 5 2^k size
      else if (!tok.empty() && tok != "marin-hip" && tok.rfind("n=", 0) != 0 && tok.rfind("m1=", 0) != 0)
        throw std::runtime_error("unknown plan token '" + tok + "'");
      i = e + 1;
    }
  }

  // rows vs columns: balance the work-group counts of the row sweep (M1 groups) and the column
  // sweeps (M2/C groups), keep a row <= 4096 pairs (64 KiB of LDS) and a column <= 1024 (x r5)
  int b = (k + 3) / 2;
  const int bmin = k - (pl.r5 == 5 ? 8 : 10);
  if (b < bmin) b = bmin;
  if (b > 12) b = 12;
  if (b > k) b = k;
  if (b < 1) b = 1;
  if (want_m2 > 0) {
    b = ilog2(size_t(want_m2));
    if ((long(1) << b) != want_m2 || b > k || b < 1 || b > 13) throw std::runtime_error("bad m2 in plan spec");
  } else {
    // very large transforms: 8192-pair rows (128 KiB) so that M1 stays <= 2048 -- or 2560 = 5 x 512, which the register-resident radix-5
    // columns serve with runs of two pairs (kernels_v5.hip, J = 1): n = 5 2^22 runs as 2560 x 4096 instead of 1280 x 8192
    while ((pl.m >> b) > (pl.r5 == 5 ? 2560u : 2048u) && b < 13 && b < k) ++b;
    if (pl.r5 == 5 && (pl.m >> b) > 10240 && b < k) b = std::min(13, k);   // n = 5 2^26: rows of 8192, columns of 5 x 4096 (split sweeps)
  }
  pl.logM2 = uint32_t(b);
  pl.M2 = 1u << b;
  pl.M1 = uint32_t(pl.m / pl.M2);
  pl.L1 = pl.M1 / pl.r5;
  pl.logL1 = uint32_t(ilog2(pl.L1));
  // columns live in LDS on the generic kernel set: M1 x C pairs of 16 bytes within the 160 KiB of a CU; beyond that (n = 5 2^26:
  // 5 x 4096 pairs) the radix-5 stage goes through memory and only the L1 = M1 / 5 part needs LDS
  if (want_split && pl.r5 != 5) throw std::runtime_error("split5 needs a 5 * 2^k transform");
  pl.split5 = want_split || size_t(pl.M1) * 16 > 160 * 1024;
  if (pl.split5 && (pl.r5 != 5 || size_t(pl.L1) * 16 > 128 * 1024)) throw std::runtime_error("transform size not supported (columns beyond the split sweeps)");

  // tile: up to 4096 pairs (64 KiB), runs of at most 16 pairs (256 B of the work buffer);
  // grow to 8192 pairs (128 KiB) when that is what C >= 4 needs
  uint32_t C = 1;
  while (C * 2 <= pl.M2 && size_t(pl.M1) * (C * 2) <= 4096 && C < 16) C *= 2;
  // (not for columns of 2048: the register-resident column kernels take them as 4096-pair tiles with C = 2)
  while (C < 4 && C * 2 <= pl.M2 && size_t(pl.M1) * (C * 2) <= 8192 && !(pl.r5 == 1 && pl.M1 == 2048)) C *= 2;
  while (C > 4 && pl.M2 / C < 256) C /= 2;   // mid-size transforms: enough tiles to fill 256 CUs
  // runs of two digits cannot absorb the run carries of a large transform (they leave log2(n) - 2 excess bits on a digit):
  // take C = 2 wherever the tile still fits the 160 KiB of LDS; beyond that the engine adds a local carry pass
  if (C == 1 && pl.M2 >= 2 && size_t(pl.M1) * 2 * 16 <= 160 * 1024 && pl.n >= (size_t(1) << 19)) C = 2;
  if (pl.split5) C = 1;
  // columns of 256 run on the register-resident radix-4 kernels with runs of four pairs (kernels_v3.hip): n = 2^20 as 256 x 2048 with
  // C = 4 takes 0.043 ms per squaring against 0.051 with C = 8 on the generic columns (round 4, profiles/r04_ab_radix4_set.txt)
  if (pl.r5 == 1 && pl.M1 == 256 && pl.M2 >= 8 && C > 4 && !pl.split5) C = 4;
  if (want_c > 0 && !pl.split5) {
    C = uint32_t(want_c);
    if ((C & (C - 1)) != 0 || C > pl.M2 || size_t(pl.M1) * C > 10240) throw std::runtime_error("bad c in plan spec");
  }
  pl.C = C;
  pl.lds_front = pl.split5 ? 0 : size_t(pl.M1) * pl.C * 16;
  pl.lds_mid = size_t(pl.M2) * 16;
  pl.q = uint32_t(p / pl.n);
  pl.t = uint32_t(p % pl.n);
  if (!build_tables) return pl;

  const size_t n = pl.n, m = pl.m;
  const uint64_t r = gf::root_of_two(n);   // 2^(1/n)
  const uint64_t om = gf::root_of_unity(m);
  const uint64_t inv_m = gf::inv(uint64_t(m) % gf::P);

  // entries [0, M1): digit 2*(M2*i1 + i2) + b split as SA[i1] + SB[2*i2 + b];  entries [M1, 2*M1): the odd
  // digit's "+p" moved to the i1 side, SA[M1 + i1] + SB[2*i2], so that both digits of a pair share the
  // column factor TB[2*i2] (the register-resident column kernels fold it into one four-step twiddle chain)
  pl.SA.resize(2 * size_t(pl.M1)); pl.TA.resize(2 * size_t(pl.M1)); pl.TAi.resize(2 * size_t(pl.M1));
  for (uint32_t i1 = 0; i1 < 2 * pl.M1; ++i1) {
    const uint64_t s0 = (uint64_t(2) * pl.M2 % n * (uint64_t(p) % n) % n * (i1 % pl.M1)) % n;
    const uint64_t s = (i1 < pl.M1) ? s0 : (s0 + pl.t) % n;
    pl.SA[i1] = uint32_t(s);
    pl.TA[i1] = gf::pow(r, (n - s) % n);
    pl.TAi[i1] = gf::mul(gf::inv(pl.TA[i1]), inv_m);
  }
  pl.SB.resize(2 * size_t(pl.M2)); pl.TB.resize(2 * size_t(pl.M2)); pl.TBi.resize(2 * size_t(pl.M2));
  for (uint32_t x = 0; x < 2 * pl.M2; ++x) {
    const uint64_t s = (uint64_t(p) * x) % n;
    pl.SB[x] = uint32_t(s);
    pl.TB[x] = gf::pow(r, (n - s) % n);
    pl.TBi[x] = gf::inv(pl.TB[x]);
  }
  pl.twh = uint32_t((ilog2(m) + 2) / 2);
  // (TWhi covers exponents below m + 2^20: the row kernels index it with label + M1 k, k < M2, and a frequency label of the prime-factor
  // columns -- kernels.hpp col_label -- is an unreduced 1536 k0 + 1025 kr < 2^20)
  const size_t lo_n = size_t(1) << pl.twh, hi_n = ((m + (size_t(1) << 20)) >> pl.twh) + 2;
  pl.TWlo.resize(lo_n); pl.TWhi.resize(hi_n);
  pl.TWlo[0] = 1; for (size_t i = 1; i < lo_n; ++i) pl.TWlo[i] = gf::mul(pl.TWlo[i - 1], om);
  const uint64_t step = gf::mul(pl.TWlo[lo_n - 1], om);
  pl.TWhi[0] = 1; for (size_t i = 1; i < hi_n; ++i) pl.TWhi[i] = gf::mul(pl.TWhi[i - 1], step);
  const uint64_t om1 = gf::pow(om, pl.M2), om2 = gf::pow(om, pl.M1);
  pl.UT1.resize(pl.M1); pl.UT1[0] = 1; for (uint32_t i = 1; i < pl.M1; ++i) pl.UT1[i] = gf::mul(pl.UT1[i - 1], om1);
  pl.UT2.resize(pl.M2); pl.UT2[0] = 1; for (uint32_t i = 1; i < pl.M2; ++i) pl.UT2[i] = gf::mul(pl.UT2[i - 1], om2);
  if (pl.M2 == 4096 || pl.M2 == 8192) {   // rows of 8192 = one radix-2 level over two 4096-point transforms
    const uint32_t st = pl.M2 / 4096;       // omega_4096 = omega_M2^st
    pl.S2r.resize(4096); pl.S2ri.resize(4096);
    for (uint32_t b = 0; b < 64; ++b) for (uint32_t ka = 0; ka < 64; ++ka) {
      const uint32_t e = ka * b;
      const uint32_t idx = b * 64 + (ka & 7) * 8 + (ka >> 3);   // [b][k1][k2], ka = k1 + 8 k2: a thread's 8 words are contiguous
      pl.S2r[idx] = pl.UT2[st * e]; pl.S2ri[idx] = pl.UT2[st * ((4096 - e) & 4095)];
    }
  }
  if (pl.M2 == 2048) {   // rows of 2048 two to a tile (kernels_v2.hip, RL = 1): [b][k1][k2] with k1 = 4 row + k1', omega_2048^((k1' + 4 k2) b)
    pl.S2r.resize(4096); pl.S2ri.resize(4096);
    for (uint32_t b = 0; b < 64; ++b) for (uint32_t k1 = 0; k1 < 8; ++k1) for (uint32_t k2 = 0; k2 < 8; ++k2) {
      const uint32_t e = (((k1 & 3) + 4 * k2) * b) & 2047;
      pl.S2r[b * 64 + k1 * 8 + k2] = pl.UT2[e]; pl.S2ri[b * 64 + k1 * 8 + k2] = pl.UT2[(2048 - e) & 2047];
    }
  }
  if (pl.r5 == 1 && (pl.M1 == 512 || pl.M1 == 1024 || pl.M1 == 2048)) {   // M1 = 512 R = (8R) x 64
    const uint32_t ka_n = pl.M1 / 64;
    pl.S1r.resize(pl.M1); pl.S1ri.resize(pl.M1);
    for (uint32_t b = 0; b < 64; ++b) for (uint32_t ka = 0; ka < ka_n; ++ka) {
      const uint32_t e = ka * b;
      const uint32_t Rr = pl.M1 / 512, idx = b * ka_n + (ka % Rr) * 8 + (ka / Rr);   // [b][k1][k2], ka = k1 + R k2
      pl.S1r[idx] = pl.UT1[e]; pl.S1ri[idx] = pl.UT1[(pl.M1 - e) & (pl.M1 - 1)];
    }
  }
  if (m % 4 == 0) { pl.I4 = gf::pow(om, m / 4); pl.I4inv = gf::inv(pl.I4); }
  else { pl.I4 = gf::root_of_unity(4); pl.I4inv = gf::inv(pl.I4); }
  if (!pl.S1r.empty() && size_t(pl.M1) * pl.C == 4096) {
    const uint32_t R = pl.M1 / 512, ND = 2 * pl.C;
    pl.DI.assign(pl.tiles() * 512, 0u);
    for (size_t T = 0; T < pl.tiles(); ++T)
      for (uint32_t t = 0; t < 512; ++t) {
        uint32_t w = 0;
        for (uint32_t d1 = 0; d1 < R; ++d1)
          for (uint32_t k = 0; k < ND; ++k) {
            const uint32_t i1 = 512 * d1 + t, i2 = uint32_t(T) * pl.C + (k >> 1);
            const uint64_t sb = pl.SB[2 * i2 + (k & 1)], s = (uint64_t(pl.SA[i1]) + sb) % n;
            const uint64_t wa = (k & 1) ? pl.SA[pl.M1 + i1] : pl.SA[i1], wb = pl.SB[2 * i2];   // the split the kernels weight with
            const uint32_t wbit = pl.width_of_s(s) - pl.q, wrap = (wa > 0 && wb > 0 && wa + wb <= n) ? 1u : 0u;
            w |= (wbit | (wrap << 1)) << (2 * (d1 * ND + k));
          }
        pl.DI[T * 512 + t] = w;
      }
  }
  if (pl.r5 == 1 && pl.M1 == 256 && pl.C == 4 && pl.M2 >= 8) {
    // columns of 256 (kernels_v3.hip): 256 threads per tile, thread t owns the run i1 = t (8 digits, 2 bits each)
    pl.DI.assign(pl.tiles() * 256, 0u);
    for (size_t T = 0; T < pl.tiles(); ++T)
      for (uint32_t t = 0; t < 256; ++t) {
        uint32_t w = 0;
        for (uint32_t k = 0; k < 8; ++k) {
          const uint32_t i1 = t, i2 = uint32_t(T) * 4 + (k >> 1);
          const uint64_t sb = pl.SB[2 * i2 + (k & 1)], s = (uint64_t(pl.SA[i1]) + sb) % n;
          const uint64_t wa = (k & 1) ? pl.SA[pl.M1 + i1] : pl.SA[i1], wb = pl.SB[2 * i2];
          const uint32_t wbit = pl.width_of_s(s) - pl.q, wrap = (wa > 0 && wb > 0 && wa + wb <= n) ? 1u : 0u;
          w |= (wbit | (wrap << 1)) << (2 * k);
        }
        pl.DI[T * 256 + t] = w;
      }
  }
  if (pl.r5 == 5 && ((pl.M1 == 1280 && pl.C == 4) || (pl.M1 == 2560 && pl.C == 2)) && pl.M2 >= 8) {
    // columns of 1280 = 5 x 256 and 2560 = 5 x 512 (kernels_v5.hip): 640 threads per tile, thread t owns the runs i1 = t + 640 d1
    // (two runs of eight digits, or four of four)
    const uint32_t NR = pl.M1 / 640, ND = 2 * pl.C;
    pl.DI.assign(pl.tiles() * 640, 0u);
    for (size_t T = 0; T < pl.tiles(); ++T)
      for (uint32_t t = 0; t < 640; ++t) {
        uint32_t w = 0;
        for (uint32_t d1 = 0; d1 < NR; ++d1)
          for (uint32_t k = 0; k < ND; ++k) {
            const uint32_t i1 = 640 * d1 + t, i2 = uint32_t(T) * pl.C + (k >> 1);
            const uint64_t sb = pl.SB[2 * i2 + (k & 1)], s = (uint64_t(pl.SA[i1]) + sb) % n;
            const uint64_t wa = (k & 1) ? pl.SA[pl.M1 + i1] : pl.SA[i1], wb = pl.SB[2 * i2];
            const uint32_t wbit = pl.width_of_s(s) - pl.q, wrap = (wa > 0 && wb > 0 && wa + wb <= n) ? 1u : 0u;
            w |= (wbit | (wrap << 1)) << (2 * (d1 * ND + k));
          }
        pl.DI[T * 640 + t] = w;
      }
  }
  if (pl.I4 != (uint64_t(1) << 48)) throw std::runtime_error("internal: omega_4 is expected to be 2^48");   // the kernels shift instead of multiplying
  if (pl.r5 == 5) {
    const uint64_t w5 = gf::pow(om, m / 5);
    for (int i = 0; i < 5; ++i) { pl.W5[i] = gf::pow(w5, uint64_t(i)); pl.W5i[i] = gf::inv(pl.W5[i]); }
    const uint64_t a = gf::add(pl.W5[1], pl.W5[4]), b = gf::add(pl.W5[2], pl.W5[3]);
    const uint64_t k1 = gf::half(gf::sub(pl.W5[1], pl.W5[4])), k2 = gf::half(gf::sub(pl.W5[2], pl.W5[3]));
    pl.W5c[0] = gf::half(gf::half(gf::sub(a, b))); pl.W5c[1] = k1; pl.W5c[2] = gf::sub(k2, k1); pl.W5c[3] = gf::add(k1, k2);
  }
  return pl;
}

}  // namespace mi355
