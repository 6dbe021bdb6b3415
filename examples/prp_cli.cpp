// mi355_prp: PRP (base 3, Gerbicz-Li checked) or LL test of 2^p-1 through the reference's `engine`
// interface served by engine_hip (include/mi355/engine_hip.h) -- the C++ twin of prmers_amd/prp.py.
// The loop restates the reference's Marin driver (src/modes/RunPrpOrLlMarin.cpp:212-462): registers
// R0 = x, R1 = Gerbicz accumulator d, R2 = multiplicand of x, R3 = check register, R4/R5 = last good
// (x, d), RBASE/RTMP = 3 and its multiplicand; check every `checklevel` blocks of B = floor(sqrt(p)).
//
//   g++ -std=c++17 -O2 -Iinclude examples/prp_cli.cpp -ldl -lgmp -o mi355_prp
//   ./mi355_prp <p> [-ll] [-erroriter N] [-checklevel L] [-maxiters K] [-lib path/to/libmi355_engine.so]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "mi355/engine_hip.h"

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <p> [-ll] [-erroriter N] [-checklevel L] [-maxiters K] [-lib so]\n", argv[0]); return 2; }
  const uint32_t p = uint32_t(std::strtoul(argv[1], nullptr, 10));
  bool ll = false; uint64_t erroriter = 0, checklevel = 0, maxiters = 0; std::string lib;
  for (int i = 2; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-ll")) ll = true;
    else if (!std::strcmp(argv[i], "-erroriter") && i + 1 < argc) erroriter = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-checklevel") && i + 1 < argc) checklevel = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-maxiters") && i + 1 < argc) maxiters = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-lib") && i + 1 < argc) lib = argv[++i];
  }
  try {
    std::unique_ptr<engine> eng(new engine_hip(p, 8, 0, false, lib));
    const size_t R0 = 0, R1 = 1, R2 = 2, R3 = 3, R4 = 4, R5 = 5, RBASE = 6, RTMP = 7;
    eng->set(R1, 1);
    eng->set(R0, ll ? 4 : 3);
    eng->copy(R4, R0); eng->copy(R5, R1);
    eng->set(RBASE, 3); eng->set_multiplicand(RTMP, RBASE);
    const uint64_t total = ll ? p - 2 : p;
    const uint64_t B = uint64_t(std::sqrt(double(p)));
    uint64_t level = checklevel ? checklevel : uint64_t(600000.0 / double(B));
    if (level == 0) level = 1;
    uint64_t itersave = 0, jsave = total - 1, checkpass = 0, errors = 0, done = 0;
    bool errordone = false, complete = true;
    mpz_t z0, z1; mpz_inits(z0, z1, nullptr);
    for (uint64_t iter = 0, j = total - 1; iter < total; ++iter, --j) {
      if (maxiters && done >= maxiters) { complete = false; break; }
      eng->square_mul(R0);
      if (ll) eng->sub(R0, 2);
      ++done;
      if (erroriter && iter + 1 == erroriter && !errordone) { errordone = true; eng->sub(R0, 2); std::printf("Injected error at iteration %llu\n", (unsigned long long)(iter + 1)); }
      if (!ll && ((j != 0 && j % B == 0) || iter == total - 1)) {
        ++checkpass;
        eng->copy(R3, R1); eng->set_multiplicand(R2, R0); eng->mul(R1, R2);
        if (!(checkpass != level && iter != total - 1)) {
          checkpass = 0;
          const uint64_t modB = (p % B == 0) ? B : p % B;
          for (uint64_t z = 0; z < (B > modB ? B - modB - 1 : 0); ++z) eng->square_mul(R3);
          if (p % B == 0) eng->mul(R3, RTMP); else eng->square_mul(R3, 3);
          for (uint64_t z = 0; z < modB; ++z) eng->square_mul(R3);
          eng->get_mpz(z0, R3); eng->get_mpz(z1, R1);
          if (mpz_cmp(z0, z1) != 0) {
            std::printf("[Gerbicz Li] Mismatch \n[Gerbicz Li] Check FAILED! iter=%llu\n[Gerbicz Li] Restore iter=%llu (j=%llu)\n",
                        (unsigned long long)(iter + 1), (unsigned long long)itersave, (unsigned long long)jsave);
            j = jsave; iter = itersave;
            if (iter == 0) { iter = iter - 1; j = j + 1; }
            ++errors;
            eng->copy(R0, R4); eng->copy(R1, R5);
          } else {
            std::printf("[Gerbicz Li] Check passed! iter=%llu\n", (unsigned long long)(iter + 1));
            eng->copy(R4, R0); eng->copy(R5, R1);
            itersave = iter; jsave = j;
          }
        }
      }
    }
    mpz_clears(z0, z1, nullptr);
    engine::digit d(eng.get(), R0);
    const bool prime = ll ? (d.equal_to(0) || d.equal_to_Mp()) : d.equal_to(9);
    std::printf("M%u %s: %s  res64(raw)=%016llX  gerbicz_errors=%llu  n=%zu\n", p, ll ? "LL" : "PRP-3",
                complete ? (prime ? "probably prime" : "composite") : "partial run",
                (unsigned long long)d.res64(), (unsigned long long)errors, eng->get_size());
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "Error: %s\n", e.what());
    return 2;   // src/main.cpp:159-164
  }
}
