// ORACLE SUPPORT -- TEST INFRASTRUCTURE ONLY.
//
// Driver (our code) around the reference's OWN host headers, compiled where they lie under
// /root/reference (never copied): include/marin/ibdwt.h, include/marin/arith.h,
// include/marin/engine.h, include/core/AlgoUtils.hpp.  Built by oracle/Makefile into
// oracle/_ref/ref_host (git-ignored).  It pins the oracle's restatement of
//   - transform size / digit widths / IBDWT weights   (ibdwt.h:17-147)
//   - engine::digit::res64 / equal_to / equal_to_Mp, engine::get_mpz / set_mpz (engine.h:173-296)
//   - pack_words_from_eng_digits / prp3_div9 / format_res64_hex / format_res2048_hex
//     (AlgoUtils.hpp:165-223)
// The reference's device kernels need an OpenCL device and cannot run in the build container, so
// nothing here executes the transform itself.
//
// usage:
//   ref_host tables <p> <out.bin>     -> u64 n, u8 width[n], u64 w[n], u64 winv[n] (natural order)
//   ref_host size <p>                 -> prints n
//   ref_host digits <p> <in.bin>      -> in: u64 d[n] encoded (value | width<<32); prints
//                                        res64 equal9 equalMp mpz_hex res64hex_div9 res2048hex_div9 res64hex_raw
//   ref_host setmpz <p> <hex> <out.bin> -> engine::set_mpz of the hex value, writes u64 d[n]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "core/AlgoUtils.hpp"  // reference header (pulls marin/engine.h, gmp)
#include "marin/ibdwt.h"       // reference header

using namespace core::algo;

namespace {

// Minimal engine whose registers are plain digit vectors: lets the reference's non-virtual /
// default-virtual host code (digit, get_mpz, set_mpz) run on data we hand it.
class digit_engine final : public engine {
  size_t n_;
  mutable std::vector<std::vector<uint64>> regs_;
  std::vector<uint8> width_;

 protected:
  void get(uint64* const d, const size_t src) const override {
    for (size_t k = 0; k < n_; ++k) d[k] = uint32(regs_[src][k]) | (uint64(width_[k]) << 32);
  }
  void set(const size_t dst, uint64* const d) const override {
    for (size_t k = 0; k < n_; ++k) regs_[dst][k] = uint32(d[k]);
  }

 public:
  digit_engine(uint32_t p, size_t regs) : n_(ibdwt::transform_size(p)), regs_(regs, std::vector<uint64>(n_, 0)), width_(n_) {
    std::vector<uint64> w(2 * n_);
    ibdwt::weights_widths(n_, p, w.data(), width_.data());
  }
  void load(size_t r, const std::vector<uint64>& d) { for (size_t k = 0; k < n_; ++k) regs_[r][k] = uint32(d[k]); }
  std::vector<uint64> raw(size_t r) const { std::vector<uint64> d(n_); get(d.data(), r); return d; }
  size_t get_size() const override { return n_; }
  void set(const Reg, const uint32) const override {}
  void copy(const Reg, const Reg) const override {}
  void square_mul(const Reg, const uint32) const override {}
  void set_multiplicand(const Reg, const Reg) const override {}
  void mul(const Reg, const Reg, const uint32) const override {}
  void sub(const Reg, const uint32) const override {}
  void add(const Reg, const Reg) const override {}
  void sub_reg(const Reg, const Reg) const override {}
  size_t get_register_data_size() const override { return n_ * 8; }
  bool get_data(std::vector<char>&, const Reg) const override { return false; }
  bool set_data(const Reg, const std::vector<char>&) const override { return false; }
  size_t get_checkpoint_size() const override { return 0; }
  bool get_checkpoint(std::vector<char>&) const override { return false; }
  bool set_checkpoint(const std::vector<char>&) const override { return false; }
};

int die(const char* m) { std::fprintf(stderr, "ref_host: %s\n", m); return 2; }

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) return die("usage");
  const std::string cmd = argv[1];
  const uint32_t p = uint32_t(std::strtoul(argv[2], nullptr, 10));
  const size_t n = ibdwt::transform_size(p);

  if (cmd == "size") { std::printf("%zu\n", n); return 0; }

  if (cmd == "tables") {
    if (argc < 4) return die("tables <p> <out.bin>");
    std::vector<uint64> w(2 * n); std::vector<uint8> width(n);
    ibdwt::weights_widths(n, p, w.data(), width.data());
    std::vector<uint64> wn(n), win(n);
    for (size_t k = 0; k < n; ++k) {  // undo the storage permutation (engine_gpu.h:1476)
      const size_t i = k / 4 + (k % 4) * (n / 4);
      wn[k] = w[2 * i + 0]; win[k] = w[2 * i + 1];
    }
    FILE* f = std::fopen(argv[3], "wb"); if (!f) return die("open out");
    const uint64 n64 = n;
    std::fwrite(&n64, 8, 1, f); std::fwrite(width.data(), 1, n, f);
    std::fwrite(wn.data(), 8, n, f); std::fwrite(win.data(), 8, n, f);
    std::fclose(f);
    return 0;
  }

  if (cmd == "digits") {
    if (argc < 4) return die("digits <p> <in.bin>");
    std::vector<uint64> d(n);
    FILE* f = std::fopen(argv[3], "rb"); if (!f) return die("open in");
    if (std::fread(d.data(), 8, n, f) != n) return die("short read");
    std::fclose(f);
    digit_engine eng(p, 1);
    eng.load(0, d);
    engine::digit dg(&eng, 0);
    mpz_t z; mpz_init(z); eng.get_mpz(z, 0);
    char* zs = mpz_get_str(nullptr, 16, z);
    std::vector<uint32_t> words = pack_words_from_eng_digits(dg, p);
    const std::string raw64 = format_res64_hex(words);
    std::vector<uint32_t> w9 = words;
    prp3_div9(p, w9);
    std::printf("%016llx %d %d %s %s %s %s\n", (unsigned long long)dg.res64(), int(dg.equal_to(9)), int(dg.equal_to_Mp()), zs,
                format_res64_hex(w9).c_str(), format_res2048_hex(w9).c_str(), raw64.c_str());
    mpz_clear(z);
    return 0;
  }

  if (cmd == "setmpz") {
    if (argc < 5) return die("setmpz <p> <hex> <out.bin>");
    digit_engine eng(p, 1);
    mpz_t z; mpz_init(z);
    if (mpz_set_str(z, argv[3], 16) != 0) return die("bad hex");
    eng.set_mpz(0, z);
    mpz_clear(z);
    const std::vector<uint64> d = eng.raw(0);
    FILE* f = std::fopen(argv[4], "wb"); if (!f) return die("open out");
    std::fwrite(d.data(), 8, n, f); std::fclose(f);
    return 0;
  }
  return die("unknown command");
}
