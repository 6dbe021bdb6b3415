// Engine: one exponent, one HIP device, one stream, reg_count registers.
// The register machine of the reference's `engine` (include/marin/engine.h:16-303) for the Marin
// path, re-designed for MI355X (see plan.hpp / kernels.hip).  Exposed through capi.cpp only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "plan.hpp"

namespace mi355 {

class Engine {
 public:
  Engine(uint32_t p, size_t reg_count, int device, bool verbose, const char* spec);
  ~Engine();
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  const Plan& plan() const { return pl_; }
#if defined(MI355_EXPERIMENTAL)
  std::string describe() const { return pl_.describe() + (coop_groups_ ? ":coop=" + std::to_string(coop_groups_) : std::string()); }
#else
  std::string describe() const { return pl_.describe(); }
#endif
  size_t n() const { return pl_.n; }
  size_t word_count() const { return (size_t(pl_.p) + 31) / 32; }
  size_t reg_count() const { return nregs_; }

  void sync();
  void set_u32(size_t dst, uint32_t v);
  void set_digits(size_t dst, const uint64_t* d, size_t count);
  void get_digits(size_t src, uint64_t* d, size_t count);
  void set_words(size_t dst, const uint32_t* w, size_t count);
  void get_words(size_t src, uint32_t* w, size_t count);
  uint64_t res64(size_t src);
  void copy(size_t dst, size_t src);
  void prepare(size_t dst, size_t src);
  void square_mul(size_t r, uint32_t a);
  // count x { square_mul(r, a); sub(r, sub) } -- the inner loop of a PRP (sub = 0) or Lucas-Lehmer (sub = 2) run between two checks.
  // One call for the whole run: the loop of launches it stands for (the one-cooperative-launch form of the small transforms was measured
  // slower and lives only in the experimental build: make exp, MI355_COOP=1).
  void square_mul_n(size_t r, uint32_t a, size_t count, uint32_t sub);
  void mul(size_t dst, size_t src, uint32_t a);
  void add(size_t dst, size_t src);
  void sub_reg(size_t dst, size_t src);
  void sub_u32(size_t r, uint32_t v);
  bool equal(size_t lhs, size_t rhs);
  // fused variants of the reference's engine (include/marin/engine.h:65-131; kernels/marin.cl:1856-2365): one sweep each
  void addsub(size_t sum_out, size_t diff_out, size_t a, size_t b);
  void addsub_copy(size_t sum, size_t diff, size_t sum_copy, size_t diff_copy, size_t a, size_t b);
  void mul_add(size_t dst, size_t mul_src, size_t add_src, uint32_t a);
  void square_mul_copy(size_t src, size_t dst_copy, uint32_t a);
  void mul_copy(size_t dst, size_t src, size_t dst_copy, uint32_t a);

  size_t register_data_size() const { return reg_bytes_ + 8; }
  void get_data(size_t src, void* data, size_t size);
  void set_data(size_t dst, const void* data, size_t size);
  size_t checkpoint_size() const { return nregs_ * register_data_size(); }
  void get_checkpoint(void* data, size_t size);
  void set_checkpoint(const void* data, size_t size);

  // measurement
  static constexpr size_t kKernels = 6;   // five kernel slots of a squaring + the measured cost of an event record
  static const char* kernel_name(size_t k);
  void time_square_mul(size_t r, uint32_t a, uint32_t sub, size_t iters, double* total_ms, double* kernel_ms, size_t kcount);
  size_t algorithmic_bytes() const { return 48 * pl_.n; }
#if defined(MI355_PROBE)
  // diagnostics build only: `iters` timed launches of one sweep (kind 0 front, 1 rows, 2 back) over grid_mult x its grid with
  // extra_lds bytes of padding LDS (forces fewer groups per CU), then one launch with the timeline probe on (8 words per group -> tl)
  void probe(int kind, int grid_mult, int extra_lds, int boost_pct, size_t iters, double* avg_ms, uint64_t* tl, size_t tl_words);
#endif

 private:
  // kDigits: unweighted u32 digits (+ deferred run carries / subtraction); kImage: multiplicand
  enum Kind : uint8_t { kDigits = 0, kImage = 1 };
  void check_reg(size_t r) const;
  void need_digits(size_t r, const char* op) const;
  uint32_t* digits(size_t r) { return reinterpret_cast<uint32_t*>(slot_[r]); }
  uint64_t* image(size_t r) { return reinterpret_cast<uint64_t*>(slot_[r]); }
  uint64_t* work() { return reinterpret_cast<uint64_t*>(slot_[nregs_]); }
  void swap_with_work(size_t r) { std::swap(slot_[r], slot_[nregs_]); }
  void read_values(size_t src, std::vector<uint64_t>& v);   // natural order, strongly carried digits
  void read_values_host(size_t src, std::vector<uint64_t>& v);   // the same through D2H + host carry (reference's way)
  uint32_t* canon_digits(size_t r, int slot);   // device: canonical digits of r in natural order (canon.hip), slot 0 / 1
  bool canon_flags_ok(uint32_t (&flags)[4]);    // reads the flag words; false: fall back to the host carry
  void write_values(size_t dst, const std::vector<uint32_t>& natural);
  void square_chain(size_t r, uint32_t a, hipEvent_t* ev);
#if defined(MI355_EXPERIMENTAL)
  void coop_launch(size_t r, uint32_t a, size_t count, uint32_t sub_next);   // count squarings in one cooperative launch (coop_groups_ != 0)
  void coop_check();                                                         // throws when a grid barrier of an earlier launch timed out
  bool coop_on() const { return coop_groups_ != 0; }
#else
  void coop_check() {}
  static constexpr bool coop_on() { return false; }
#endif
  uint64_t* cbuf(size_t r) { return cb_[r]; }
  uint64_t* take_spare_cbuf();                      // carry-word buffers are handed around like the register slots
  void adopt_cbuf(size_t r, uint64_t* fresh);       // r's pending carries are now in `fresh`; its old buffer becomes spare
  void digits_ready(size_t r);                      // r as a digit register for a run-wise kernel (small subtraction applied)
  void linear(long s1, long s2, long d1, long d2, size_t a, size_t b);
  void back_ext(size_t dst, uint32_t a, long copy_to, long add_src);
  void normalize(size_t r);          // apply deferred run carries / small subtraction
  void carry_fix_now(size_t r);      // run carries into the digits right away (plans with runs of two digits cannot defer them)
  void run_front(size_t r);          // digits(r) -> work_, consuming pending state when the kernel can
  void run_middle(const uint64_t* in, const uint64_t* y, uint64_t* out, int mode, uint32_t sub);
  void run_back(size_t r, uint32_t a);

  Plan pl_;
  DevPlan dp_{};
  int device_ = 0;
  bool verbose_ = false;
  hipStream_t stream_ = nullptr;
  size_t nregs_ = 0, reg_bytes_ = 0;
  unsigned char* regs_ = nullptr;            // (reg_count + 1) slots of 8n bytes; the extra one is the work buffer
  std::vector<unsigned char*> slot_;          // slot_[r]: storage of register r; slot_[reg_count]: work buffer (swappable)
  uint64_t* cbuf_ = nullptr;                  // reg_count + 4 buffers of runs() carry words
  std::vector<uint64_t*> cb_, cb_spare_;      // cb_[r]: the buffer register r uses now
  void* tables_ = nullptr;
  uint64_t* split_ = nullptr;   // second work buffer of the split column sweeps (plan.split5: n = 5 2^26)
  uint64_t* f0_ = nullptr;   // four-step chain starts / ratios of the register-resident column kernels
  uint32_t* di_ = nullptr;   // digit-info words of the register-resident column kernels (plan.hpp DI)
  std::vector<uint8_t> kind_;
  std::vector<uint8_t> pending_carry_;   // cbuf(r) not yet folded into the digits
  std::vector<uint32_t> pending_sub_;    // small constant still to subtract (LL's -2)
  bool v2rows_ = false, v2cols_ = false;
#if defined(MI355_EXPERIMENTAL)
  // runs of squarings with back + front in one launch (kernels_v3.hip k31_cols256_planes; square_mul_n): tiles of such a launch (0: not
  // served), carry hand-over words, flag words (+ the error word), epochs used so far, whether a launch is still unchecked / has failed
  uint32_t chain_tiles_ = 0, chain_epoch_ = 0;
  uint64_t* chain_x_ = nullptr;
  uint32_t* chain_flags_ = nullptr;
  bool chain_used_ = false, chain_failed_ = false;
  // runs of squarings with back + front in one launch on the radix-8 column shapes (kernels_v2.hip k31_cols; square_mul_n, MI355_CHAIN=1):
  // hand-over words (one per run; null: not served / off) + the error word behind them, tag of the next launch, whether a launch is still
  // unchecked / has failed
  uint64_t* xchain_ = nullptr;
  uint32_t xchain_tag_ = 1;
  bool xchain_used_ = false, xchain_failed_ = false;
  uint32_t* xchain_err() const { return reinterpret_cast<uint32_t*>(xchain_ + pl_.runs()); }
  void chain_check();   // throws when a hand-over wait of an earlier launch timed out (the registers are not valid then)
#else
  void chain_check() {}
#endif
  std::vector<uint8_t> width_;   // natural order
  std::vector<uint32_t> stage_;  // host staging (one register of digits)
  uint32_t* canon_ = nullptr;    // device scratch of the canonicalisation: work arrays + two outputs of n digits (lazy)
  bool host_carry_ = false;      // MI355_HOST_CARRY=1: compare / res64 / read-back through the host (A/B tests)
#if defined(MI355_EXPERIMENTAL)
  // one-launch squarings of the small transforms (kernels.hip k_coop): grid size (0: not served / MI355_COOP=0), barrier words
  // [groups] + error word, barriers passed so far, squarings per launch in time_square_mul (MI355_COOP_BATCH, A/B and bench)
  uint32_t coop_groups_ = 0, coop_epoch_ = 0, coop_fault_ = 0;
  uint32_t* coop_flags_ = nullptr;
  size_t coop_batch_ = 1;
  bool coop_used_ = false, coop_failed_ = false;
#endif
};

}  // namespace mi355
