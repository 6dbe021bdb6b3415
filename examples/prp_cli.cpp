// mi355_prp: PRP (base 3, Gerbicz-Li checked) or LL test of 2^p-1 through the reference's `engine`
// interface served by engine_hip (include/mi355/engine_hip.h) -- the C++ twin of prmers_amd/prp.py.
// The loop restates the reference's Marin driver (src/modes/RunPrpOrLlMarin.cpp:212-462): registers
// R0 = x, R1 = Gerbicz accumulator d, R2 = multiplicand of x, R3 = check register, R4/R5 = last good
// (x, d), RBASE/RTMP = 3 and its multiplicand; check every `checklevel` blocks of B = floor(sqrt(p)).
//
// Caller-side files as the reference leaves them (include/mi355/caller_formats.h): -worktodo takes the first PRP= / Test= entry
// and rotates the file when the test is complete; -ckpt DIR resumes from / saves version-2 checkpoints together with the
// Gerbicz-Li rollback point (every -backup N iterations, at the end of a partial run, and on SIGINT / SIGTERM, which end the
// run with exit code 0 like the reference's interrupt path, RunPrpOrLlMarin.cpp:296-309); -proof POWER writes the residues a proof of that power needs under
// <p>/proof/; -json FILE appends the result line.
//
//   g++ -std=c++17 -O2 -Iinclude examples/prp_cli.cpp -ldl -lgmp -o mi355_prp
//   ./mi355_prp <p> | -worktodo FILE  [-d DEVICE] [-ll | -llsafe [-llsafe_block B]] [-erroriter N] [-checklevel L] [-maxiters K] [-ckpt DIR]
//               [-backup N] [-proof POWER] [-json FILE] [-lib path/to/libmi355_engine.so]
// -d DEVICE is the reference's device selector (src/io/CliParser.cpp:198): BASELINE configs[4] is eight of these processes, `-d i
// -worktodo file_i`, one per GPU.  -llsafe is the Lucas-Lehmer test with error detection by block re-computation
// (src/modes/RunLlSafeMarin.cpp:95-392, run_ll_safe below).
#include <cmath>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "mi355/caller_formats.h"
#include "mi355/engine_hip.h"

namespace fmt = mi355::formats;

static volatile std::sig_atomic_t g_interrupted = 0;
static void on_signal(int) { g_interrupted = 1; }

static std::vector<uint32_t> residue_words(engine* eng, size_t reg, uint32_t p) {
  engine::digit d(eng, reg);
  std::vector<uint64_t> v(d.get_size());
  for (size_t i = 0; i < v.size(); ++i) v[i] = (uint64_t(d.width(i)) << 32) | uint64_t(d.val(i));
  return fmt::pack_words(v, p);
}

// Lucas-Lehmer with error detection by block re-computation (src/modes/RunLlSafeMarin.cpp:95-392; prmers_amd/prp.py run_ll_safe is the
// Python twin).  V follows x -> x^2 - 2 from 4, U accumulates the product of the V's (set_multiplicand + mul per iteration, :257-260);
// every B = p / sqrt(p) iterations (or -llsafe_block) the block is recomputed from the last good (V, U) and both pairs must agree
// (:268-296), otherwise the state rolls back to the block start (:297-318).  -erroriter injects V -= 2 once (:245-255).
static int run_ll_safe(engine* eng, uint32_t p, uint64_t block, uint64_t erroriter, uint64_t maxiters, const std::string& json_file) {
  const size_t RV = 0, RU = 1, RVC = 2, RUC = 3, RVCHK = 4, RUCHK = 5, RTMP = 6;   // RunLlSafeMarin.cpp:20-28
  const uint64_t total = p >= 2 ? uint64_t(p) - 2 : 0;
  eng->set(RV, 4); eng->set(RU, 2);
  eng->copy(RVC, RV); eng->copy(RUC, RU); eng->copy(RVCHK, RVC); eng->copy(RUCHK, RUC);
  uint64_t B = block ? block : uint64_t(double(p) / std::sqrt(double(p)));
  if (B < 1) B = 1;
  if (B > total && total) B = total;
  auto step = [&](size_t rv, size_t ru) { eng->set_multiplicand(RTMP, rv); eng->mul(ru, RTMP); eng->square_mul(rv); eng->sub(rv, 2); };
  bool errordone = false, complete = true;
  uint64_t itersave = 0, errors = 0, checks = 0, done = 0;
  for (uint64_t iter = 0; iter < total; ++iter) {
    if (maxiters && done >= maxiters) { complete = false; break; }
    if (g_interrupted) { std::printf("\nInterrupted by user at iteration %llu\n", (unsigned long long)iter); return 0; }
    if (erroriter && iter + 1 == erroriter && !errordone) { errordone = true; eng->sub(RV, 2); std::printf("Injected error at iteration %llu\n", (unsigned long long)(iter + 1)); }
    step(RV, RU);
    ++done;
    if ((iter + 1) % B == 0 || iter + 1 == total) {
      const uint64_t blk = ((iter + 1) % B == 0) ? B : (iter + 1) - itersave;
      eng->copy(RVCHK, RVC); eng->copy(RUCHK, RUC);
      for (uint64_t z = 0; z < blk; ++z) step(RVCHK, RUCHK);
      ++checks;
      // the reference compares mpz read-backs (:281-293); is_equal compares the canonical forms on the device
      if (!(eng->is_equal(RVCHK, RV) && eng->is_equal(RUCHK, RU))) {
        std::printf("[Error check] Mismatch \n[Error check] Check FAILED! iter=%llu\n[Error check] Restore iter=%llu\n", (unsigned long long)iter, (unsigned long long)itersave);
        if (++errors > 64) { std::fprintf(stderr, "Error: %llu failed block checks, giving up\n", (unsigned long long)errors); return 1; }
        eng->copy(RV, RVC); eng->copy(RU, RUC);
        iter = itersave - 1;   // (wraps to -1 for itersave = 0: the loop increment brings it back to 0, as :305-311)
      } else {
        std::printf("[Error check] Check passed! iter=%llu\n", (unsigned long long)iter);
        eng->copy(RVC, RV); eng->copy(RUC, RU);
        itersave = iter + 1;
      }
    }
  }
  engine::digit d(eng, RV);
  const bool is_mp = d.equal_to_Mp();
  const bool prime = complete && (d.equal_to(0) || is_mp);
  std::vector<uint32_t> W = residue_words(eng, RV, p);
  if (prime && is_mp) std::fill(W.begin(), W.end(), 0u);   // the all-ones vector stands for 0 (:335-338)
  std::printf("M%u LL-safe: %s  res64=%s  checks=%llu errors=%llu  n=%zu\n", p, complete ? (prime ? "prime" : "composite") : "partial run",
              fmt::res64_hex(W).c_str(), (unsigned long long)checks, (unsigned long long)errors, eng->get_size());
  if (complete) {
    fmt::ResultInfo r; r.exponent = p; r.ll = true; r.is_prime = prime; r.res64 = fmt::res64_hex(W); r.res2048 = fmt::res2048_hex(W);
    r.gerbicz_errors = unsigned(errors); r.fft_length = unsigned(eng->get_size());
    const std::string line = fmt::result_json(r);
    std::printf("%s\n", line.c_str());
    if (!json_file.empty()) { std::ofstream f(json_file, std::ios::app); f << line << "\n"; }
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <p> | -worktodo FILE [-d DEVICE] [-ll | -llsafe [-llsafe_block B]] [-erroriter N] [-checklevel L] [-maxiters K] [-ckpt DIR] [-backup N] [-proof POWER] [-json FILE] [-fft SPEC] [-lib so]\n", argv[0]); return 2; }
  uint32_t p = uint32_t(std::strtoul(argv[1], nullptr, 10));
  bool ll = false, llsafe = false; uint64_t erroriter = 0, checklevel = 0, maxiters = 0, backup = 0, llsafe_block = 0; std::string lib, worktodo, ckpt_dir, json_file, fft; uint32_t proof_power = 0;
  size_t device = 0;
  fmt::WorkEntry entry;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-ll") || !std::strcmp(argv[i], "-llunsafe")) ll = true;
    else if (!std::strcmp(argv[i], "-llsafe")) { ll = true; llsafe = true; }
    else if (!std::strcmp(argv[i], "-llsafe_block") && i + 1 < argc) llsafe_block = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-d") && i + 1 < argc) device = size_t(std::strtoul(argv[++i], nullptr, 10));   // src/io/CliParser.cpp:198
    else if (!std::strcmp(argv[i], "-worktodo") && i + 1 < argc) worktodo = argv[++i];
    else if (!std::strcmp(argv[i], "-fft") && i + 1 < argc) fft = argv[++i];   // e.g. crt:9 (the reference's -fft, README.md:907-926)
    else if (!std::strcmp(argv[i], "-ckpt") && i + 1 < argc) ckpt_dir = argv[++i];
    else if (!std::strcmp(argv[i], "-backup") && i + 1 < argc) backup = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-proof") && i + 1 < argc) proof_power = uint32_t(std::strtoul(argv[++i], nullptr, 10));
    else if (!std::strcmp(argv[i], "-json") && i + 1 < argc) json_file = argv[++i];
    else if (!std::strcmp(argv[i], "-erroriter") && i + 1 < argc) erroriter = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-checklevel") && i + 1 < argc) checklevel = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-maxiters") && i + 1 < argc) maxiters = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-lib") && i + 1 < argc) lib = argv[++i];
  }
  if (!worktodo.empty()) {
    entry = fmt::first_worktodo_entry(worktodo);
    if (!entry.valid) { std::fprintf(stderr, "Error: no PRP= / Test= entry in %s\n", worktodo.c_str()); return 2; }
    p = entry.exponent; ll = entry.ll;
    std::printf("worktodo: %s\n", entry.raw.c_str());
  }
  std::signal(SIGINT, on_signal);
  std::signal(SIGTERM, on_signal);
  try {
    std::unique_ptr<engine> eng(new engine_hip(p, 8, device, false, lib, fft));
    if (llsafe) return run_ll_safe(eng.get(), p, llsafe_block, erroriter, maxiters, json_file);
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
    // resume (RunPrpOrLlMarin.cpp:212-236): the checkpoint holds every register, the callers rebuild their multiplicands
    uint32_t ri = 0; double elapsed0 = 0;
    const std::string ckpt = ckpt_dir.empty() ? std::string() : fmt::checkpoint_name(p, ll, ckpt_dir);
    if (!ckpt.empty()) {
      int rc = fmt::load_checkpoint(ckpt, *eng, p, ll, ri, elapsed0);
      if (rc != 0) rc = fmt::load_checkpoint(ckpt + ".old", *eng, p, ll, ri, elapsed0);
      if (rc == 0) {
        std::printf("Resuming from a checkpoint at iteration %u\n", ri);
        eng->set(RBASE, 3); eng->set_multiplicand(RTMP, RBASE);
        fmt::GerbiczState g;
        if (!ll && fmt::load_gerbicz_state(ckpt, ri, g)) {   // R4 / R5 of the checkpoint are the state these counters name
          itersave = g.itersave; jsave = g.jsave; checkpass = g.checkpass;
        } else {                                              // no rollback point on file: roll back to the resumed state itself
          eng->copy(R4, R0); eng->copy(R5, R1);
          itersave = ri ? ri - 1 : 0; jsave = ri ? total - ri : total - 1;
        }
      } else ri = 0;
    }
    auto checkpoint = [&](uint32_t at) {
      fmt::save_checkpoint(ckpt, *eng, p, ll, at, elapsed0);
      if (!ll) fmt::save_gerbicz_state(ckpt, at, fmt::GerbiczState{itersave, jsave, checkpass});
    };
    bool interrupted = false;
    std::unique_ptr<fmt::ProofPoints> proof;
    if (proof_power && !ll) proof.reset(new fmt::ProofPoints(p, proof_power));
    uint64_t last_iter = ri;
    for (uint64_t iter = ri, j = total - ri - 1; iter < total; ++iter, --j) {
      last_iter = iter;   // iterations completed so far (a rollback moves it back too)
      if (maxiters && done >= maxiters) { complete = false; break; }
      if (g_interrupted) {
        complete = false; interrupted = true;
        std::printf("\nInterrupted by user, state saved at iteration %llu j=%llu\n", (unsigned long long)iter, (unsigned long long)j);
        break;
      }
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
          // the reference compares two mpz read-backs (RunPrpOrLlMarin.cpp:363-366); is_equal does it on the device
          if (!eng->is_equal(R3, R1)) {
            std::printf("[Gerbicz Li] Mismatch \n[Gerbicz Li] Check FAILED! iter=%llu\n[Gerbicz Li] Restore iter=%llu (j=%llu)\n",
                        (unsigned long long)(iter + 1), (unsigned long long)itersave, (unsigned long long)jsave);
            j = jsave; iter = itersave;
            if (iter == 0) { iter = iter - 1; j = j + 1; }
            checkpass = 0;   // RunPrpOrLlMarin.cpp:391
            ++errors;
            if (errors > 64) { std::fprintf(stderr, "Error: %llu failed Gerbicz-Li checks, giving up (hardware or checkpoint problem)\n", (unsigned long long)errors); return 1; }
            eng->copy(R0, R4); eng->copy(R1, R5);
          } else {
            std::printf("[Gerbicz Li] Check passed! iter=%llu\n", (unsigned long long)(iter + 1));
            eng->copy(R4, R0); eng->copy(R5, R1);
            itersave = iter; jsave = j;
          }
        }
      }
      // proof residues (ProofManagerMarin::checkpointMarin, ProofManagerMarin.cpp:84-120) and periodic backups (:430-447)
      if (proof && proof->should_checkpoint(uint32_t(iter + 1))) proof->save(uint32_t(iter + 1), residue_words(eng.get(), R0, p));
      if (!ckpt.empty() && backup && done % backup == 0 && iter + 1 < total) checkpoint(uint32_t(iter + 1));
    }
    if (!ckpt.empty() && !complete) checkpoint(uint32_t(last_iter));   // iterations completed = the next iteration index
    if (interrupted) return 0;
    engine::digit d(eng.get(), R0);
    const bool prime = ll ? (d.equal_to(0) || d.equal_to_Mp()) : d.equal_to(9);
    std::printf("M%u %s: %s  res64(raw)=%016llX  gerbicz_errors=%llu  n=%zu\n", p, ll ? "LL" : "PRP-3",
                complete ? (prime ? "probably prime" : "composite") : "partial run",
                (unsigned long long)d.res64(), (unsigned long long)errors, eng->get_size());
    if (complete) {
      std::vector<uint32_t> W = residue_words(eng.get(), R0, p);
      if (!ll && p % 6 != 0) { uint32_t two_p_mod9 = 1; for (uint32_t i = 0; i < p % 6; ++i) two_p_mod9 = two_p_mod9 * 2 % 9; if (two_p_mod9 != 1) fmt::prp3_div9(p, W); }
      fmt::ResultInfo r; r.exponent = p; r.ll = ll; r.is_prime = prime; r.res64 = fmt::res64_hex(W); r.res2048 = fmt::res2048_hex(W);
      r.gerbicz_errors = unsigned(errors); r.fft_length = unsigned(eng->get_size()); r.aid = entry.aid;
      const std::string line = fmt::result_json(r);
      std::printf("%s\n", line.c_str());
      if (!json_file.empty()) { std::ofstream f(json_file, std::ios::app); f << line << "\n"; }
      if (!worktodo.empty()) {
        const bool more = fmt::rotate_worktodo(worktodo, "worktodo_save.txt");
        std::printf("Entry removed from %s and saved to worktodo_save.txt%s\n", worktodo.c_str(), more ? "; more entries remain" : "; no more entries");
      }
    }
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "Error: %s\n", e.what());
    return 2;   // src/main.cpp:159-164
  }
}
