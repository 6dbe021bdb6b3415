// Resident engine of the GF(M61^2) x GF(M31^2) squaring (crt_engine.hip): a register file of residues (digits of up to 39 bits in
// logical order) with the operations of the reference's plugin ABI (third_party/aevum/src/EngineApi.h:28-59): set, copy, square_mul,
// set_multiplicand / mul, add, sub, compare, read-back as canonical words.  Selected behind mi355_engine_create with
// fft_spec = "crt[:odd][:words=N][:h2=K]" (capi.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace mi355 {

size_t crt_transform_size(uint32_t p, uint32_t odd);
uint32_t crt_auto_radix(uint32_t p, size_t* words);   // 1, 3 or 9 by the reference's stock / PFA size-ratio gates (0: no admissible size)

class CrtEngine {
 public:
  static constexpr int kKernels = 6;
  static const char* kernel_name(size_t k);

  // odd in {1, 3, 9}; n_words = 0: the smallest admissible odd * 2^k; spec: "h2=K" forces rows of 2^K complex values (tests)
  CrtEngine(uint32_t p, size_t reg_count, uint32_t odd, size_t n_words, int device, const char* spec);
  ~CrtEngine();
  CrtEngine(const CrtEngine&) = delete;
  CrtEngine& operator=(const CrtEngine&) = delete;

  size_t size() const;
  size_t reg_count() const;
  uint32_t odd() const;
  uint32_t exponent() const;
  std::string describe() const;
  size_t algorithmic_bytes() const;

  void set_u32(size_t reg, uint32_t a);
  void copy(size_t dst, size_t src);
  void square_mul(size_t reg, uint32_t a);
  void set_multiplicand(size_t dst, size_t src);
  void mul(size_t dst, size_t src, uint32_t a);
  void add(size_t dst, size_t src);
  void sub_reg(size_t dst, size_t src);
  void sub_u32(size_t reg, uint32_t a);
  // the fused variants of engine.h:65-131 as the compositions its base class defines (one hidden scratch register)
  void addsub(long sum_out, long sum_copy, long diff_out, long diff_copy, size_t a, size_t b);
  void mul_add(size_t dst, size_t mul_src, size_t add_src, uint32_t f);
  void square_mul_copy(size_t src, size_t dst_copy, uint32_t f);
  void mul_copy(size_t dst, size_t src, size_t dst_copy, uint32_t f);
  bool equal(size_t a, size_t b);
  void set_digits(size_t reg, const uint64_t* d, size_t count);
  void get_digits(size_t reg, uint64_t* d, size_t count, bool canonical);
  // engine::get / engine::set form (engine.h:24-25): canonical value | width << 32 -- only for sizes whose words have at most 32 bits
  void get_digits_encoded(size_t reg, uint64_t* d, size_t count);
  void set_digits_encoded(size_t reg, const uint64_t* d, size_t count);
  void set_words(size_t reg, const uint32_t* w, size_t count);
  void get_words(size_t reg, uint32_t* w, size_t count);
  uint64_t res64(size_t reg);
  // raw register images (engine.h:134-146): 12 bytes per word + an 8-byte kind tag; a residue uses the first 8 n bytes (digits), a
  // multiplicand all 12 n (its packed spectrum).  Implementation-defined, as the reference's images are.
  size_t register_data_size() const;
  void get_data(size_t src, void* data, size_t size);
  void set_data(size_t dst, const void* data, size_t size);
  void sync();
  void time_square_mul(size_t reg, uint32_t a, size_t iters, double* total_ms, double* kernel_ms, size_t kernel_count);

 private:
  struct Impl;
  Impl* im_;
  void release();
  void check_digits(size_t reg, const char* what) const;
  void ensure_headroom(size_t reg);                       // relax a register whose additions would overflow the next transform
  uint64_t* canon_digits(size_t reg, int slot);           // device-side canonical form (canon.hip), slot 0 / 1
  bool canon_flags_ok(uint32_t (&flags)[4]);
  const uint64_t* canonical_on_device(size_t src);        // canonical digits of src in device memory (slot 1)
  void get_digits_host(size_t reg, uint64_t* d, size_t count);   // read-back + host carry (fallback, MI355_HOST_CARRY=1)
  void launch_transform(size_t reg, int mode, size_t other, uint32_t a, bool timed);
};

}  // namespace mi355
