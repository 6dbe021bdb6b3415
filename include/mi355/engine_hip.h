// engine_hip: the reference's `engine` interface (include/marin/engine.h) served by
// libmi355_engine.so through its C ABI (include/mi355_engine.h).  Header-only, dlopen-based, written
// after the reference's own plugin adapter (src/aevum/EngineAevum.cpp:75-82,225-243,252-493):
// same load order (environment variable, then next to the executable / cwd), same error policy
// (0 from the ABI -> std::runtime_error("MI355 <op> failed: <last_error>"), EngineAevum.cpp:477-485).
//
// Drop-in hook: in src/marin/gpu.cpp:149 replace
//     return new engine_gpu(q, reg_count, device, verbose);
// by
//     return new engine_hip(q, reg_count, device, verbose);
// so that `-engine-marin` selects this backend.
#pragma once
#ifdef MI355_USE_REFERENCE_ENGINE_H
#include "marin/engine.h"
#else
#include "engine_iface.h"
#endif

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

class engine_hip final : public engine {
  struct Api {
    void* so = nullptr;
    const char* (*version)() = nullptr;
    const char* (*last_error)() = nullptr;
    void* (*create)(uint32_t, size_t, uint32_t, int, const char*, const char*) = nullptr;
    void (*destroy)(void*) = nullptr;
    size_t (*transform_size)(void*) = nullptr;
    size_t (*word_count)(void*) = nullptr;
    int (*sync)(void*) = nullptr;
    int (*set_u32)(void*, size_t, uint32_t) = nullptr;
    int (*copy)(void*, size_t, size_t) = nullptr;
    int (*prepare)(void*, size_t, size_t) = nullptr;
    int (*square_mul)(void*, size_t, uint32_t) = nullptr;
    int (*mul)(void*, size_t, size_t, uint32_t) = nullptr;
    int (*add)(void*, size_t, size_t) = nullptr;
    int (*sub_reg)(void*, size_t, size_t) = nullptr;
    int (*sub_u32)(void*, size_t, uint32_t) = nullptr;
    int (*equal)(void*, size_t, size_t, int*) = nullptr;
    int (*get_digits)(void*, size_t, uint64_t*, size_t) = nullptr;
    int (*set_digits)(void*, size_t, const uint64_t*, size_t) = nullptr;
    size_t (*register_data_size)(void*) = nullptr;
    int (*get_data)(void*, size_t, void*, size_t) = nullptr;
    int (*set_data)(void*, size_t, const void*, size_t) = nullptr;
    size_t (*checkpoint_size)(void*) = nullptr;
    int (*get_checkpoint)(void*, void*, size_t) = nullptr;
    int (*set_checkpoint)(void*, const void*, size_t) = nullptr;
    int (*addsub)(void*, size_t, size_t, size_t, size_t) = nullptr;
    int (*addsub_copy)(void*, size_t, size_t, size_t, size_t, size_t, size_t) = nullptr;
    int (*mul_add)(void*, size_t, size_t, size_t, uint32_t) = nullptr;
    int (*square_mul_copy)(void*, size_t, size_t, uint32_t) = nullptr;
    int (*mul_copy)(void*, size_t, size_t, size_t, uint32_t) = nullptr;
    int (*square_mul_n)(void*, size_t, uint32_t, size_t, uint32_t) = nullptr;   // optional (libraries before round 3 lack it)

    template <class F> void bind(F& f, const char* name) {
      f = reinterpret_cast<F>(dlsym(so, name));
      if (!f) throw std::runtime_error(std::string("libmi355_engine.so lacks symbol ") + name);
    }
    explicit Api(const std::string& hint) {
      std::vector<std::string> cand;
      if (const char* e = std::getenv("MI355_ENGINE_LIB")) cand.push_back(e);
      if (!hint.empty()) cand.push_back(hint);
      cand.push_back("./libmi355_engine.so");
      cand.push_back("libmi355_engine.so");
      cand.push_back("/usr/local/lib/prmers/libmi355_engine.so");
      std::string tried;
      for (const auto& c : cand) {
        so = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (so) break;
        tried += " " + c;
      }
      if (!so) throw std::runtime_error("cannot load libmi355_engine.so (tried:" + tried + ")");
      bind(version, "mi355_engine_version"); bind(last_error, "mi355_engine_last_error");
      bind(create, "mi355_engine_create"); bind(destroy, "mi355_engine_destroy");
      bind(transform_size, "mi355_engine_transform_size"); bind(word_count, "mi355_engine_word_count");
      bind(sync, "mi355_engine_sync"); bind(set_u32, "mi355_engine_set_u32"); bind(copy, "mi355_engine_copy");
      bind(prepare, "mi355_engine_prepare"); bind(square_mul, "mi355_engine_square_mul"); bind(mul, "mi355_engine_mul");
      bind(add, "mi355_engine_add"); bind(sub_reg, "mi355_engine_sub_reg"); bind(sub_u32, "mi355_engine_sub_u32");
      bind(equal, "mi355_engine_equal"); bind(get_digits, "mi355_engine_get_digits"); bind(set_digits, "mi355_engine_set_digits");
      bind(register_data_size, "mi355_engine_register_data_size"); bind(get_data, "mi355_engine_get_data");
      bind(set_data, "mi355_engine_set_data"); bind(checkpoint_size, "mi355_engine_checkpoint_size");
      bind(get_checkpoint, "mi355_engine_get_checkpoint"); bind(set_checkpoint, "mi355_engine_set_checkpoint");
      bind(addsub, "mi355_engine_addsub"); bind(addsub_copy, "mi355_engine_addsub_copy"); bind(mul_add, "mi355_engine_mul_add");
      bind(square_mul_copy, "mi355_engine_square_mul_copy"); bind(mul_copy, "mi355_engine_mul_copy");
      square_mul_n = reinterpret_cast<decltype(square_mul_n)>(dlsym(so, "mi355_engine_square_mul_n"));
    }
    ~Api() { if (so) dlclose(so); }
  };

  Api _api;
  void* _h = nullptr;
  size_t _n = 0;

  void ok(int rc, const char* op) const {
    if (!rc) throw std::runtime_error(std::string("MI355 ") + op + " failed: " + _api.last_error());
  }

 protected:
  void get(uint64_t* const d, const size_t src) const override { ok(_api.get_digits(_h, src, d, _n), "get"); }
  void set(const size_t dst, uint64_t* const d) const override { ok(_api.set_digits(_h, dst, d, _n), "set"); }

 public:
  // fft_spec: "" = the Goldilocks path with its automatic plan, "m2=..,c=.." a forced plan, "crt[:odd][:words=N]" the
  // GF(M61^2) x GF(M31^2) family with a prime-factor axis (the reference hands its -fft string to its plugin the same way,
  // src/aevum/EngineAevum.cpp:252-300)
  engine_hip(const uint32_t q, const size_t reg_count, const size_t device, const bool verbose, const std::string& lib_hint = "",
             const std::string& fft_spec = "")
      : _api(lib_hint) {
    _h = _api.create(q, reg_count, uint32_t(device), verbose ? 1 : 0, fft_spec.empty() ? nullptr : fft_spec.c_str(), nullptr);
    if (!_h) throw std::runtime_error(std::string("MI355 create failed: ") + _api.last_error());
    _n = _api.transform_size(_h);
  }
  ~engine_hip() override { if (_h) _api.destroy(_h); }
  engine_hip(const engine_hip&) = delete;
  engine_hip& operator=(const engine_hip&) = delete;

  size_t get_size() const override { return _n; }
  void sync() const override { ok(_api.sync(_h), "sync"); }
  void set(const Reg dst, const uint32_t a) const override { ok(_api.set_u32(_h, dst, a), "set"); }
  void copy(const Reg dst, const Reg src) const override { ok(_api.copy(_h, dst, src), "copy"); }
  void square_mul(const Reg src, const uint32_t a = 1) const override { ok(_api.square_mul(_h, src, a), "square_mul"); }
  void set_multiplicand(const Reg dst, const Reg src) const override { ok(_api.prepare(_h, dst, src), "set_multiplicand"); }
  void mul(const Reg dst, const Reg src, const uint32_t a = 1) const override { ok(_api.mul(_h, dst, src, a), "mul"); }
  void sub(const Reg src, const uint32_t a) const override { ok(_api.sub_u32(_h, src, a), "sub"); }
  void add(const Reg dst, const Reg src) const override { ok(_api.add(_h, dst, src), "add"); }
  void sub_reg(const Reg dst, const Reg src) const override { ok(_api.sub_reg(_h, dst, src), "sub_reg"); }
  // count x { src = src^2 * a; src -= sub }: the run of plain iterations of the callers' loops (RunPrpOrLlMarin.cpp:338-409) as one call;
  // not part of the reference's engine (a non-virtual extra of this adapter), the loop of calls when the library does not export it
  void square_mul_n(const Reg src, const size_t count, const uint32_t a = 1, const uint32_t sub_after = 0) const {
    if (_api.square_mul_n) { ok(_api.square_mul_n(_h, src, a, count, sub_after), "square_mul_n"); return; }
    for (size_t i = 0; i < count; ++i) { square_mul(src, a); if (sub_after) sub(src, sub_after); }
  }
  // fused variants: one sweep each in the library instead of the base-class compositions (engine.h:65-131)
  void mul_add(const Reg dst, const Reg mul_src, const Reg add_src, const uint32_t a = 1) const override { ok(_api.mul_add(_h, dst, mul_src, add_src, a), "mul_add"); }
  void addsub(const Reg sum_out, const Reg diff_out, const Reg a, const Reg b) const override { ok(_api.addsub(_h, sum_out, diff_out, a, b), "addsub"); }
  void square_mul_copy(const Reg src, const Reg dst_copy, const uint32_t a = 1) const override { ok(_api.square_mul_copy(_h, src, dst_copy, a), "square_mul_copy"); }
  void mul_copy(const Reg dst, const Reg src, const Reg dst_copy, const uint32_t a = 1) const override { ok(_api.mul_copy(_h, dst, src, dst_copy, a), "mul_copy"); }
  void addsub_copy(const Reg sum, const Reg diff, const Reg sum_copy, const Reg diff_copy, const Reg a, const Reg b) const override {
    ok(_api.addsub_copy(_h, sum, diff, sum_copy, diff_copy, a, b), "addsub_copy");
  }
  bool is_equal(const Reg lhs, const Reg rhs) const
#if defined(MI355_HAVE_GMP) || defined(MI355_USE_REFERENCE_ENGINE_H)
      override
#endif
  { int eq = 0; ok(_api.equal(_h, lhs, rhs, &eq), "is_equal"); return eq != 0; }

  size_t get_register_data_size() const override { return _api.register_data_size(_h); }
  bool get_data(std::vector<char>& data, const Reg src) const override {   // false on size mismatch (engine_gpu.h:2138)
    return data.size() == get_register_data_size() && _api.get_data(_h, src, data.data(), data.size()) != 0;
  }
  bool set_data(const Reg dst, const std::vector<char>& data) const override {
    return data.size() == get_register_data_size() && _api.set_data(_h, dst, data.data(), data.size()) != 0;
  }
  size_t get_checkpoint_size() const override { return _api.checkpoint_size(_h); }
  bool get_checkpoint(std::vector<char>& data) const override {
    return data.size() == get_checkpoint_size() && _api.get_checkpoint(_h, data.data(), data.size()) != 0;
  }
  bool set_checkpoint(const std::vector<char>& data) const override {
    return data.size() == get_checkpoint_size() && _api.set_checkpoint(_h, data.data(), data.size()) != 0;
  }
};
