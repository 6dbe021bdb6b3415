// Stand-alone statement of the reference's `engine` register-machine interface
// (cherubrock-seb/PrMers include/marin/engine.h:16-303), for building and testing the MI355X adapter
// in this repository without the reference tree.  Inside the reference, include its own
// "marin/engine.h" instead (define MI355_USE_REFERENCE_ENGINE_H before including engine_hip.h):
// the adapter only relies on the virtual interface below, whose names, argument order and meaning
// are the reference's.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#if __has_include(<gmp.h>)
#include <gmp.h>
#define MI355_HAVE_GMP 1
#endif

class engine {
 public:
  typedef size_t Reg;                       // engine.h:37
  enum class gpu_backend { marin, aevum, auto_select };
  enum class gpu_workload { generic, prp, ll, pm1, pm1_lowmem, pm1_ultralowmem, ecm };

 protected:
  // n digits, low 32 bits = value, high 32 bits = width of the digit's base (engine.h:23-25)
  virtual void get(uint64_t* const d, const size_t src) const = 0;
  virtual void set(const size_t dst, uint64_t* const d) const = 0;

 public:
  engine() {}
  virtual ~engine() {}
  virtual bool is_aevum_backend() const { return false; }
  virtual void release_gpu_resources_for_lowmem_handoff() {}
  virtual size_t get_size() const = 0;                                     // :40
  virtual void sync() const {}                                             // :45
  virtual void set(const Reg dst, const uint32_t a) const = 0;             // :47  dst = a
  virtual void copy(const Reg dst, const Reg src) const = 0;               // :49
  virtual void square_mul(const Reg src, const uint32_t a = 1) const = 0;  // :51  src = src^2 * a
  virtual void set_multiplicand(const Reg dst, const Reg src) const = 0;   // :53
  virtual void set_multiplicand2(const Reg dst, const Reg src) const { set_multiplicand(dst, src); }
  virtual void mul(const Reg dst, const Reg src, const uint32_t a = 1) const = 0;  // :60 dst = dst * src * a
  virtual void sub(const Reg src, const uint32_t a) const = 0;             // :62
  virtual void add(const Reg dst, const Reg src) const = 0;                // :64
  virtual void sub_reg(const Reg dst, const Reg src) const = 0;            // :71
  // compositions with the reference's default behaviour (engine.h:65-131)
  virtual void mul_add(const Reg dst, const Reg mul_src, const Reg add_src, const uint32_t a = 1) const { mul(dst, mul_src, a); add(dst, add_src); }
  virtual void addsub(const Reg s, const Reg d, const Reg a, const Reg b) const { copy(s, a); copy(d, a); add(s, b); sub_reg(d, b); }
  virtual void square_mul_copy(const Reg src, const Reg cp, const uint32_t a = 1) const { square_mul(src, a); copy(cp, src); }
  virtual void mul_new(const Reg dst, const Reg src, const uint32_t a = 1) const { mul(dst, src, a); }
  virtual void mul_copy(const Reg dst, const Reg src, const Reg cp, const uint32_t a = 1) const { mul(dst, src, a); copy(cp, dst); }
  virtual void mul_pair_unit(const Reg d0, const Reg s0, const Reg d1, const Reg s1) const {
    set_multiplicand(s0, s0); mul(d0, s0); set_multiplicand(s1, s1); mul(d1, s1);
  }
  virtual void mul_pair_prepared(const Reg d0, const Reg m0, const Reg d1, const Reg m1, const uint32_t a0 = 1, const uint32_t a1 = 1) const {
    mul(d0, m0, a0); mul(d1, m1, a1);
  }
  virtual void addsub_copy(const Reg s, const Reg d, const Reg sc, const Reg dc, const Reg a, const Reg b) const {
    addsub(s, d, a, b); copy(sc, s); copy(dc, d);
  }
  virtual size_t get_register_data_size() const = 0;                       // :134
  virtual bool get_data(std::vector<char>& data, const Reg src) const = 0;  // :137
  virtual bool set_data(const Reg dst, const std::vector<char>& data) const = 0;
  virtual size_t get_checkpoint_size() const = 0;                          // :142
  virtual bool get_checkpoint(std::vector<char>& data) const = 0;
  virtual bool set_checkpoint(const std::vector<char>& data) const = 0;

  // dst = src^e, src is erased (engine.h:160-170)
  void pow(const Reg dst, const Reg src, const uint64_t e) const {
    set_multiplicand(src, src);
    set(dst, 1);
    for (int i = 63; i >= 0; --i) {
      if ((e >> i) == 0) continue;
      square_mul(dst);
      if ((e >> i) & 1) mul(dst, src);
    }
  }

#ifdef MI355_HAVE_GMP
  virtual bool is_equal(const Reg lhs, const Reg rhs) const {
    mpz_t a, b; mpz_inits(a, b, nullptr);
    get_mpz(a, lhs); get_mpz(b, rhs);
    const bool eq = mpz_cmp(a, b) == 0;
    mpz_clears(a, b, nullptr);
    return eq;
  }
  // canonical value in [0, 2^p-1): the all-ones digit vector is zero (engine.h:173-203)
  virtual void get_mpz(mpz_t& z, const Reg src) const {
    const size_t n = get_size();
    std::vector<uint64_t> d(n);
    get(d.data(), src);
    bool ones = true;
    for (uint64_t v : d) ones = ones && (uint32_t(v) == (uint64_t(1) << (v >> 32)) - 1);
    mpz_set_ui(z, 0);
    if (ones) return;
    mpz_t t; mpz_init(t);
    size_t bit = 0;
    for (uint64_t v : d) { mpz_set_ui(t, uint32_t(v)); mpz_mul_2exp(t, t, bit); mpz_add(z, z, t); bit += size_t(v >> 32); }
    mpz_clear(t);
  }
  // digits of the low p bits of z (engine.h:206-232)
  virtual void set_mpz(const Reg dst, const mpz_t& z) const {
    const size_t n = get_size();
    std::vector<uint64_t> d(n);
    get(d.data(), dst);   // widths
    mpz_t t, q; mpz_init_set(q, z); mpz_init(t);
    for (uint64_t& v : d) {
      const unsigned w = unsigned(v >> 32);
      mpz_tdiv_r_2exp(t, q, w);
      mpz_tdiv_q_2exp(q, q, w);
      v = uint64_t(mpz_get_ui(t)) | (uint64_t(w) << 32);
    }
    mpz_clears(t, q, nullptr);
    set(dst, d.data());
  }
#endif

  // unsigned digit view of a register (engine.h:234-296)
  class digit {
    std::vector<uint64_t> _data;
   public:
    digit(engine* const eng, const Reg src) { _data.resize(eng->get_size()); eng->get(_data.data(), src); }
    virtual ~digit() {}
    size_t get_size() const { return _data.size(); }
    uint32_t val(const size_t i) const { return uint32_t(_data[i]); }
    uint8_t width(const size_t i) const { return uint8_t(_data[i] >> 32); }
    uint64_t res64() const {
      uint64_t r = 0; unsigned s = 0;
      for (uint64_t v : _data) { r += uint64_t(uint32_t(v)) << s; s += unsigned(v >> 32); if (s >= 64) break; }
      return r;
    }
    bool equal_to(const uint64_t a) const {
      uint64_t r = a;
      for (uint64_t v : _data) { const unsigned w = unsigned(v >> 32); if ((r & ((uint64_t(1) << w) - 1)) != uint32_t(v)) return false; r >>= w; }
      return true;
    }
    bool equal_to_Mp() const {
      for (uint64_t v : _data) if (uint32_t(v) != (uint64_t(1) << (v >> 32)) - 1) return false;
      return true;
    }
  };
};
