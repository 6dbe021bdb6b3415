// Resident engine of the GF(M61^2) x GF(M31^2) squaring (crt_engine.hip): one residue on the device, digits of up to 39 bits in
// logical order.  The subset of the reference's plugin ABI (third_party/aevum/src/EngineApi.h:28-59) that an LL test or a plain
// PRP needs: set, square_mul, sub, read-back as canonical words.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace mi355 {

size_t crt_transform_size(uint32_t p, uint32_t odd);

class CrtEngine {
 public:
  static constexpr int kKernels = 6;
  static const char* kernel_name(size_t k);

  // odd in {1, 3, 9}; n_words = 0: the smallest admissible odd * 2^k; spec: "h2=K" forces rows of 2^K complex values (tests)
  CrtEngine(uint32_t p, uint32_t odd, size_t n_words, int device, const char* spec);
  ~CrtEngine();
  CrtEngine(const CrtEngine&) = delete;
  CrtEngine& operator=(const CrtEngine&) = delete;

  size_t size() const;
  uint32_t odd() const;
  uint32_t exponent() const;
  std::string describe() const;
  size_t algorithmic_bytes() const;

  void set_u32(uint32_t a);
  void square_mul(uint32_t a);
  void sub_u32(uint32_t a);
  void set_digits(const uint64_t* d, size_t count);
  void get_digits(uint64_t* d, size_t count, bool canonical);
  void get_words(uint32_t* w, size_t count);
  uint64_t res64();
  void sync();
  void time_square_mul(uint32_t a, size_t iters, double* total_ms, double* kernel_ms, size_t kernel_count);

 private:
  struct Impl;
  Impl* im_;
  void release();
  void launch_square(uint32_t a, bool timed);
};

}  // namespace mi355
