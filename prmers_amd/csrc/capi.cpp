// C ABI of libmi355_engine.so (include/mi355_engine.h).  Every entry point catches everything and
// reports through last_error, as the reference's plugin does (third_party/aevum/src/EngineApi.cpp:447-517).
#include "../../include/mi355_engine.h"

#include <cstring>
#include <exception>
#include <string>

#include "crt_engine.hpp"
#include "engine.hpp"
namespace mi355 {
void selftest_primitives(int device);
void crt_carry_host(uint32_t p, size_t n, uint32_t odd, uint32_t a, const uint64_t* in61, const uint32_t* in31, uint64_t* digits_out,
                    uint64_t* residual_out, int device, double* kernel_ms);
}

namespace {

thread_local std::string g_last_error;

template <class F>
int guarded(F&& f) {
  try {
    f();
    return 1;
  } catch (const std::exception& e) {
    g_last_error = e.what();
  } catch (...) {
    g_last_error = "unknown error";
  }
  return 0;
}

mi355::CrtEngine* crt(mi355_crt_handle h) {
  if (!h) throw std::runtime_error("null engine handle");
  return static_cast<mi355::CrtEngine*>(h);
}

mi355::Engine* eng(mi355_engine_handle h) {
  if (!h) throw std::runtime_error("null engine handle");
  return static_cast<mi355::Engine*>(h);
}

}  // namespace

extern "C" {

const char* mi355_engine_version(void) { return "mi355-marin-hip 0.1 (gfx950)"; }
const char* mi355_engine_last_error(void) { return g_last_error.c_str(); }

int mi355_engine_resolve_fft(uint32_t exponent, const char* fft_spec, char* output, size_t output_size) {
  return guarded([&] {
    if (!output || output_size == 0) throw std::runtime_error("resolve_fft: no output buffer");
    const mi355::Plan pl = mi355::make_plan(exponent, fft_spec, false);
    const std::string s = pl.describe();
    if (s.size() + 1 > output_size) throw std::runtime_error("resolve_fft: output buffer too small");
    std::memcpy(output, s.c_str(), s.size() + 1);
  });
}

mi355_engine_handle mi355_engine_create(uint32_t exponent, size_t register_count, uint32_t device, int verbose,
                                        const char* fft_spec, const char* /*tune_dir*/) {
  mi355::Engine* e = nullptr;
  if (!guarded([&] { e = new mi355::Engine(exponent, register_count, int(device), verbose != 0, fft_spec); })) return nullptr;
  return e;
}

void mi355_engine_destroy(mi355_engine_handle h) {
  guarded([&] { delete static_cast<mi355::Engine*>(h); });
}

size_t mi355_engine_transform_size(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = eng(h)->n(); }); return r; }
size_t mi355_engine_word_count(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = eng(h)->word_count(); }); return r; }
int mi355_engine_sync(mi355_engine_handle h) { return guarded([&] { eng(h)->sync(); }); }

int mi355_engine_set_u32(mi355_engine_handle h, size_t dst, uint32_t v) { return guarded([&] { eng(h)->set_u32(dst, v); }); }
int mi355_engine_set_words(mi355_engine_handle h, size_t dst, const uint32_t* w, size_t count) {
  return guarded([&] { if (!w) throw std::runtime_error("set_words: null buffer"); eng(h)->set_words(dst, w, count); });
}
int mi355_engine_get_words(mi355_engine_handle h, size_t src, uint32_t* w, size_t count) {
  return guarded([&] { if (!w) throw std::runtime_error("get_words: null buffer"); eng(h)->get_words(src, w, count); });
}
int mi355_engine_copy(mi355_engine_handle h, size_t dst, size_t src) { return guarded([&] { eng(h)->copy(dst, src); }); }
int mi355_engine_prepare(mi355_engine_handle h, size_t dst, size_t src) { return guarded([&] { eng(h)->prepare(dst, src); }); }
int mi355_engine_square_mul(mi355_engine_handle h, size_t r, uint32_t f) { return guarded([&] { eng(h)->square_mul(r, f); }); }
int mi355_engine_mul(mi355_engine_handle h, size_t dst, size_t src, uint32_t f) { return guarded([&] { eng(h)->mul(dst, src, f); }); }
int mi355_engine_add(mi355_engine_handle h, size_t dst, size_t src) { return guarded([&] { eng(h)->add(dst, src); }); }
int mi355_engine_sub_reg(mi355_engine_handle h, size_t dst, size_t src) { return guarded([&] { eng(h)->sub_reg(dst, src); }); }
int mi355_engine_sub_u32(mi355_engine_handle h, size_t dst, uint32_t v) { return guarded([&] { eng(h)->sub_u32(dst, v); }); }
int mi355_engine_equal(mi355_engine_handle h, size_t lhs, size_t rhs, int* out) {
  return guarded([&] { if (!out) throw std::runtime_error("equal: null output"); *out = eng(h)->equal(lhs, rhs) ? 1 : 0; });
}

int mi355_engine_addsub(mi355_engine_handle h, size_t so, size_t dout, size_t a, size_t b) { return guarded([&] { eng(h)->addsub(so, dout, a, b); }); }
int mi355_engine_addsub_copy(mi355_engine_handle h, size_t s1, size_t d1, size_t s2, size_t d2, size_t a, size_t b) {
  return guarded([&] { eng(h)->addsub_copy(s1, d1, s2, d2, a, b); });
}
int mi355_engine_mul_add(mi355_engine_handle h, size_t dst, size_t ms, size_t as, uint32_t f) { return guarded([&] { eng(h)->mul_add(dst, ms, as, f); }); }
int mi355_engine_square_mul_copy(mi355_engine_handle h, size_t src, size_t cp, uint32_t f) { return guarded([&] { eng(h)->square_mul_copy(src, cp, f); }); }
int mi355_engine_mul_copy(mi355_engine_handle h, size_t dst, size_t src, size_t cp, uint32_t f) { return guarded([&] { eng(h)->mul_copy(dst, src, cp, f); }); }

int mi355_engine_get_digits(mi355_engine_handle h, size_t src, uint64_t* d, size_t count) {
  return guarded([&] { if (!d) throw std::runtime_error("get_digits: null buffer"); eng(h)->get_digits(src, d, count); });
}
int mi355_engine_set_digits(mi355_engine_handle h, size_t dst, const uint64_t* d, size_t count) {
  return guarded([&] { if (!d) throw std::runtime_error("set_digits: null buffer"); eng(h)->set_digits(dst, d, count); });
}
int mi355_engine_res64(mi355_engine_handle h, size_t src, uint64_t* out) {
  return guarded([&] { if (!out) throw std::runtime_error("res64: null output"); *out = eng(h)->res64(src); });
}
size_t mi355_engine_register_data_size(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = eng(h)->register_data_size(); }); return r; }
int mi355_engine_get_data(mi355_engine_handle h, size_t src, void* data, size_t size) {
  return guarded([&] { if (!data) throw std::runtime_error("get_data: null buffer"); eng(h)->get_data(src, data, size); });
}
int mi355_engine_set_data(mi355_engine_handle h, size_t dst, const void* data, size_t size) {
  return guarded([&] { if (!data) throw std::runtime_error("set_data: null buffer"); eng(h)->set_data(dst, data, size); });
}
size_t mi355_engine_checkpoint_size(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = eng(h)->checkpoint_size(); }); return r; }
int mi355_engine_get_checkpoint(mi355_engine_handle h, void* data, size_t size) {
  return guarded([&] { if (!data) throw std::runtime_error("get_checkpoint: null buffer"); eng(h)->get_checkpoint(data, size); });
}
int mi355_engine_set_checkpoint(mi355_engine_handle h, const void* data, size_t size) {
  return guarded([&] { if (!data) throw std::runtime_error("set_checkpoint: null buffer"); eng(h)->set_checkpoint(data, size); });
}

int mi355_engine_time_square_mul(mi355_engine_handle h, size_t reg, uint32_t factor, uint32_t sub, size_t iters,
                                 double* total_ms, double* kernel_ms, size_t kernel_count) {
  return guarded([&] { eng(h)->time_square_mul(reg, factor, sub, iters, total_ms, kernel_ms, kernel_count); });
}
size_t mi355_engine_kernel_count(mi355_engine_handle) { return mi355::Engine::kKernels; }
const char* mi355_engine_kernel_name(mi355_engine_handle, size_t k) { return mi355::Engine::kernel_name(k); }
int mi355_engine_selftest(size_t device) { return guarded([&] { mi355::selftest_primitives(int(device)); }); }
int mi355_crt_carry(uint32_t exponent, size_t words, uint32_t odd, uint32_t factor, const uint64_t* in61, const uint32_t* in31, uint64_t* digits_out,
                    uint64_t* residual_out, size_t device, double* kernel_ms) {
  return guarded([&] {
    if (!in61 || !in31 || !digits_out || !residual_out) throw std::runtime_error("crt_carry: null buffer");
    mi355::crt_carry_host(exponent, words, odd, factor, in61, in31, digits_out, residual_out, int(device), kernel_ms);
  });
}
size_t mi355_crt_transform_size(uint32_t exponent, uint32_t odd) { size_t r = 0; guarded([&] { r = mi355::crt_transform_size(exponent, odd); }); return r; }
mi355_crt_handle mi355_crt_create(uint32_t exponent, uint32_t odd, size_t words, uint32_t device, const char* spec) {
  mi355::CrtEngine* e = nullptr;
  if (!guarded([&] { e = new mi355::CrtEngine(exponent, odd, words, int(device), spec); })) return nullptr;
  return e;
}
void mi355_crt_destroy(mi355_crt_handle h) { guarded([&] { delete static_cast<mi355::CrtEngine*>(h); }); }
size_t mi355_crt_size(mi355_crt_handle h) { size_t r = 0; guarded([&] { r = crt(h)->size(); }); return r; }
int mi355_crt_describe(mi355_crt_handle h, char* output, size_t output_size) {
  return guarded([&] {
    const std::string s = crt(h)->describe();
    if (!output || s.size() + 1 > output_size) throw std::runtime_error("describe: output buffer too small");
    std::memcpy(output, s.c_str(), s.size() + 1);
  });
}
int mi355_crt_sync(mi355_crt_handle h) { return guarded([&] { crt(h)->sync(); }); }
int mi355_crt_set_u32(mi355_crt_handle h, uint32_t v) { return guarded([&] { crt(h)->set_u32(v); }); }
int mi355_crt_square_mul(mi355_crt_handle h, uint32_t a) { return guarded([&] { crt(h)->square_mul(a); }); }
int mi355_crt_sub_u32(mi355_crt_handle h, uint32_t v) { return guarded([&] { crt(h)->sub_u32(v); }); }
int mi355_crt_get_digits(mi355_crt_handle h, uint64_t* d, size_t count, int canonical) {
  return guarded([&] { if (!d) throw std::runtime_error("get_digits: null buffer"); crt(h)->get_digits(d, count, canonical != 0); });
}
int mi355_crt_set_digits(mi355_crt_handle h, const uint64_t* d, size_t count) {
  return guarded([&] { if (!d) throw std::runtime_error("set_digits: null buffer"); crt(h)->set_digits(d, count); });
}
int mi355_crt_get_words(mi355_crt_handle h, uint32_t* w, size_t count) {
  return guarded([&] { if (!w) throw std::runtime_error("get_words: null buffer"); crt(h)->get_words(w, count); });
}
int mi355_crt_res64(mi355_crt_handle h, uint64_t* out) { return guarded([&] { if (!out) throw std::runtime_error("res64: null output"); *out = crt(h)->res64(); }); }
int mi355_crt_time_square_mul(mi355_crt_handle h, uint32_t a, size_t iters, double* total_ms, double* kernel_ms, size_t kernel_count) {
  return guarded([&] { crt(h)->time_square_mul(a, iters, total_ms, kernel_ms, kernel_count); });
}
size_t mi355_crt_kernel_count(void) { return mi355::CrtEngine::kKernels; }
const char* mi355_crt_kernel_name(size_t k) { return mi355::CrtEngine::kernel_name(k); }
size_t mi355_crt_algorithmic_bytes(mi355_crt_handle h) { size_t r = 0; guarded([&] { r = crt(h)->algorithmic_bytes(); }); return r; }
size_t mi355_engine_algorithmic_bytes(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = eng(h)->algorithmic_bytes(); }); return r; }

}  // extern "C"
