// C ABI of libmi355_engine.so (include/mi355_engine.h).  Every entry point catches everything and
// reports through last_error, as the reference's plugin does (third_party/aevum/src/EngineApi.cpp:447-517).
#include "../../include/mi355_engine.h"

#include <cctype>
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

// One handle type for both field families: the Goldilocks engine (engine.hip) or, when the fft_spec starts with "crt", the
// GF(M61^2) x GF(M31^2) engine with the prime-factor axis (crt_engine.hip) -- the reference selects its backends the same way, by the
// spec string handed to its plugin (third_party/aevum/src/EngineApi.cpp:75-120, `-fft`).
struct Handle {
  mi355::Engine* g = nullptr;
  mi355::CrtEngine* c = nullptr;
  ~Handle() { delete g; delete c; }
};
Handle* hnd(mi355_engine_handle h) {
  if (!h) throw std::runtime_error("null engine handle");
  return static_cast<Handle*>(h);
}
mi355::Engine* eng(mi355_engine_handle h) {
  Handle* H = hnd(h);
  if (!H->g) throw std::runtime_error("this operation is not implemented for the crt field family");
  return H->g;
}
mi355::CrtEngine* crt(mi355_engine_handle h) { return hnd(h)->c; }   // nullptr for a Goldilocks handle

// "crt" / "crt:auto" (radix by the reference's policy, crt_auto_radix), "crt:9", "crt:3:words=6291456", "crt:9:h2=5" ...
struct CrtSpec { uint32_t odd = 0; size_t words = 0; std::string rest; };   // odd 0: automatic
bool parse_crt_spec(const char* spec, CrtSpec& out) {
  if (!spec || std::strncmp(spec, "crt", 3) != 0 || (spec[3] != 0 && spec[3] != ':')) return false;
  std::string s = spec[3] ? spec + 4 : "";
  size_t pos = 0;
  bool first = true;
  while (pos <= s.size() && !s.empty()) {
    const size_t e = s.find(':', pos);
    const std::string tok = s.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
    if (first && !tok.empty() && std::isdigit(static_cast<unsigned char>(tok[0]))) out.odd = uint32_t(std::stoul(tok));
    else if (first && tok == "auto") out.odd = 0;
    else if (tok.rfind("words=", 0) == 0) out.words = std::stoull(tok.substr(6));
    else if (!tok.empty()) out.rest = tok;
    first = false;
    if (e == std::string::npos) break;
    pos = e + 1;
  }
  return true;
}
// the radix of an automatic spec ("crt", "crt:auto", "crt:words=N": N decides when it is given)
void settle_crt_radix(uint32_t exponent, CrtSpec& cs) {
  if (cs.odd != 0) return;
  if (cs.words) cs.odd = (cs.words % 9 == 0) ? 9u : (cs.words % 3 == 0) ? 3u : 1u;
  else cs.odd = mi355::crt_auto_radix(exponent, nullptr);
  if (cs.odd == 0) throw std::runtime_error("crt: no admissible transform size for this exponent");
}

}  // namespace

extern "C" {

const char* mi355_engine_version(void) { return "mi355-marin-hip " MI355_ENGINE_VERSION " (gfx950)"; }
const char* mi355_engine_last_error(void) { return g_last_error.c_str(); }

int mi355_engine_resolve_fft(uint32_t exponent, const char* fft_spec, char* output, size_t output_size) {
  return guarded([&] {
    if (!output || output_size == 0) throw std::runtime_error("resolve_fft: no output buffer");
    std::string s;
    CrtSpec cs;
    if (parse_crt_spec(fft_spec, cs)) {
      settle_crt_radix(exponent, cs);
      const size_t n = cs.words ? cs.words : mi355::crt_transform_size(exponent, cs.odd);
      if (!n) throw std::runtime_error("resolve_fft: no admissible crt transform size");
      s = "crt-hip:n=" + std::to_string(n) + ":odd=" + std::to_string(cs.odd);
    } else {
      s = mi355::make_plan(exponent, fft_spec, false).describe();
    }
    if (s.size() + 1 > output_size) throw std::runtime_error("resolve_fft: output buffer too small");
    std::memcpy(output, s.c_str(), s.size() + 1);
  });
}

mi355_engine_handle mi355_engine_create(uint32_t exponent, size_t register_count, uint32_t device, int verbose,
                                        const char* fft_spec, const char* /*tune_dir*/) {
  Handle* H = nullptr;
  if (!guarded([&] {
        H = new Handle;
        CrtSpec cs;
        if (parse_crt_spec(fft_spec, cs)) { settle_crt_radix(exponent, cs); H->c = new mi355::CrtEngine(exponent, register_count, cs.odd, cs.words, int(device), cs.rest.empty() ? nullptr : cs.rest.c_str()); }
        else H->g = new mi355::Engine(exponent, register_count, int(device), verbose != 0, fft_spec);
      })) {
    delete H;
    return nullptr;
  }
  return H;
}

void mi355_engine_destroy(mi355_engine_handle h) {
  guarded([&] { delete static_cast<Handle*>(h); });
}

size_t mi355_engine_transform_size(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = (crt(h) ? crt(h)->size() : eng(h)->n()); }); return r; }
size_t mi355_engine_word_count(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = (crt(h) ? (size_t(crt(h)->exponent()) + 31) / 32 : eng(h)->word_count()); }); return r; }
int mi355_engine_sync(mi355_engine_handle h) { return guarded([&] { if (crt(h)) crt(h)->sync(); else eng(h)->sync(); }); }

int mi355_engine_set_u32(mi355_engine_handle h, size_t dst, uint32_t v) { return guarded([&] { if (crt(h)) crt(h)->set_u32(dst, v); else eng(h)->set_u32(dst, v); }); }
int mi355_engine_set_words(mi355_engine_handle h, size_t dst, const uint32_t* w, size_t count) {
  return guarded([&] { if (!w) throw std::runtime_error("set_words: null buffer"); if (crt(h)) crt(h)->set_words(dst, w, count); else eng(h)->set_words(dst, w, count); });
}
int mi355_engine_get_words(mi355_engine_handle h, size_t src, uint32_t* w, size_t count) {
  return guarded([&] { if (!w) throw std::runtime_error("get_words: null buffer"); if (crt(h)) crt(h)->get_words(src, w, count); else eng(h)->get_words(src, w, count); });
}
int mi355_engine_copy(mi355_engine_handle h, size_t dst, size_t src) { return guarded([&] { if (crt(h)) crt(h)->copy(dst, src); else eng(h)->copy(dst, src); }); }
int mi355_engine_prepare(mi355_engine_handle h, size_t dst, size_t src) { return guarded([&] { if (crt(h)) crt(h)->set_multiplicand(dst, src); else eng(h)->prepare(dst, src); }); }
int mi355_engine_square_mul(mi355_engine_handle h, size_t r, uint32_t f) { return guarded([&] { if (crt(h)) crt(h)->square_mul(r, f); else eng(h)->square_mul(r, f); }); }
int mi355_engine_mul(mi355_engine_handle h, size_t dst, size_t src, uint32_t f) { return guarded([&] { if (crt(h)) crt(h)->mul(dst, src, f); else eng(h)->mul(dst, src, f); }); }
int mi355_engine_add(mi355_engine_handle h, size_t dst, size_t src) { return guarded([&] { if (crt(h)) crt(h)->add(dst, src); else eng(h)->add(dst, src); }); }
int mi355_engine_sub_reg(mi355_engine_handle h, size_t dst, size_t src) { return guarded([&] { if (crt(h)) crt(h)->sub_reg(dst, src); else eng(h)->sub_reg(dst, src); }); }
int mi355_engine_sub_u32(mi355_engine_handle h, size_t dst, uint32_t v) { return guarded([&] { if (crt(h)) crt(h)->sub_u32(dst, v); else eng(h)->sub_u32(dst, v); }); }
int mi355_engine_equal(mi355_engine_handle h, size_t lhs, size_t rhs, int* out) {
  return guarded([&] { if (!out) throw std::runtime_error("equal: null output"); *out = (crt(h) ? crt(h)->equal(lhs, rhs) : eng(h)->equal(lhs, rhs)) ? 1 : 0; });
}

int mi355_engine_addsub(mi355_engine_handle h, size_t so, size_t dout, size_t a, size_t b) { return guarded([&] { if (crt(h)) crt(h)->addsub(long(so), -1, long(dout), -1, a, b); else eng(h)->addsub(so, dout, a, b); }); }
int mi355_engine_addsub_copy(mi355_engine_handle h, size_t s1, size_t d1, size_t s2, size_t d2, size_t a, size_t b) {
  return guarded([&] { if (crt(h)) crt(h)->addsub(long(s1), long(s2), long(d1), long(d2), a, b); else eng(h)->addsub_copy(s1, d1, s2, d2, a, b); });
}
int mi355_engine_mul_add(mi355_engine_handle h, size_t dst, size_t ms, size_t as, uint32_t f) { return guarded([&] { if (crt(h)) crt(h)->mul_add(dst, ms, as, f); else eng(h)->mul_add(dst, ms, as, f); }); }
int mi355_engine_square_mul_copy(mi355_engine_handle h, size_t src, size_t cp, uint32_t f) { return guarded([&] { if (crt(h)) crt(h)->square_mul_copy(src, cp, f); else eng(h)->square_mul_copy(src, cp, f); }); }
int mi355_engine_square_mul_n(mi355_engine_handle h, size_t r, uint32_t f, size_t count, uint32_t sub) {
  return guarded([&] {
    if (crt(h)) { for (size_t i = 0; i < count; ++i) { crt(h)->square_mul(r, f); if (sub) crt(h)->sub_u32(r, sub); } }
    else eng(h)->square_mul_n(r, f, count, sub);
  });
}
int mi355_engine_mul_copy(mi355_engine_handle h, size_t dst, size_t src, size_t cp, uint32_t f) { return guarded([&] { if (crt(h)) crt(h)->mul_copy(dst, src, cp, f); else eng(h)->mul_copy(dst, src, cp, f); }); }

int mi355_engine_get_digits(mi355_engine_handle h, size_t src, uint64_t* d, size_t count) {
  return guarded([&] { if (!d) throw std::runtime_error("get_digits: null buffer"); if (crt(h)) crt(h)->get_digits_encoded(src, d, count); else eng(h)->get_digits(src, d, count); });
}
int mi355_engine_set_digits(mi355_engine_handle h, size_t dst, const uint64_t* d, size_t count) {
  return guarded([&] { if (!d) throw std::runtime_error("set_digits: null buffer"); if (crt(h)) crt(h)->set_digits_encoded(dst, d, count); else eng(h)->set_digits(dst, d, count); });
}
int mi355_engine_res64(mi355_engine_handle h, size_t src, uint64_t* out) {
  return guarded([&] { if (!out) throw std::runtime_error("res64: null output"); *out = crt(h) ? crt(h)->res64(src) : eng(h)->res64(src); });
}
size_t mi355_engine_register_data_size(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = crt(h) ? crt(h)->register_data_size() : eng(h)->register_data_size(); }); return r; }
int mi355_engine_get_data(mi355_engine_handle h, size_t src, void* data, size_t size) {
  return guarded([&] { if (!data) throw std::runtime_error("get_data: null buffer"); if (crt(h)) crt(h)->get_data(src, data, size); else eng(h)->get_data(src, data, size); });
}
int mi355_engine_set_data(mi355_engine_handle h, size_t dst, const void* data, size_t size) {
  return guarded([&] { if (!data) throw std::runtime_error("set_data: null buffer"); if (crt(h)) crt(h)->set_data(dst, data, size); else eng(h)->set_data(dst, data, size); });
}
size_t mi355_engine_checkpoint_size(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = crt(h) ? crt(h)->register_data_size() * crt(h)->reg_count() : eng(h)->checkpoint_size(); }); return r; }
int mi355_engine_get_checkpoint(mi355_engine_handle h, void* data, size_t size) {
  return guarded([&] { if (!data) throw std::runtime_error("get_checkpoint: null buffer"); if (crt(h)) {
      const size_t rs = crt(h)->register_data_size(), nr = crt(h)->reg_count();
      if (size != rs * nr) throw std::runtime_error("get_checkpoint: size mismatch");
      for (size_t r = 0; r < nr; ++r) crt(h)->get_data(r, static_cast<unsigned char*>(data) + r * rs, rs);
    } else eng(h)->get_checkpoint(data, size); });
}
int mi355_engine_set_checkpoint(mi355_engine_handle h, const void* data, size_t size) {
  return guarded([&] { if (!data) throw std::runtime_error("set_checkpoint: null buffer"); if (crt(h)) {
      const size_t rs = crt(h)->register_data_size(), nr = crt(h)->reg_count();
      if (size != rs * nr) throw std::runtime_error("set_checkpoint: size mismatch");
      for (size_t r = 0; r < nr; ++r) crt(h)->set_data(r, static_cast<const unsigned char*>(data) + r * rs, rs);
    } else eng(h)->set_checkpoint(data, size); });
}

int mi355_engine_time_square_mul(mi355_engine_handle h, size_t reg, uint32_t factor, uint32_t sub, size_t iters,
                                 double* total_ms, double* kernel_ms, size_t kernel_count) {
  return guarded([&] { if (crt(h)) { if (sub) throw std::runtime_error("time_square_mul: no deferred subtraction on the crt family"); crt(h)->time_square_mul(reg, factor, iters, total_ms, kernel_ms, kernel_count); } else eng(h)->time_square_mul(reg, factor, sub, iters, total_ms, kernel_ms, kernel_count); });
}
size_t mi355_engine_kernel_count(mi355_engine_handle h) { return (h && static_cast<Handle*>(h)->c) ? size_t(mi355::CrtEngine::kKernels) : size_t(mi355::Engine::kKernels); }
const char* mi355_engine_kernel_name(mi355_engine_handle h, size_t k) {
  return (h && static_cast<Handle*>(h)->c) ? mi355::CrtEngine::kernel_name(k) : mi355::Engine::kernel_name(k);
}
int mi355_engine_describe(mi355_engine_handle h, char* output, size_t output_size) {
  return guarded([&] {
    const std::string s = crt(h) ? crt(h)->describe() : eng(h)->describe();
    if (!output || s.size() + 1 > output_size) throw std::runtime_error("describe: output buffer too small");
    std::memcpy(output, s.c_str(), s.size() + 1);
  });
}
int mi355_engine_selftest(size_t device) { return guarded([&] { mi355::selftest_primitives(int(device)); }); }
int mi355_crt_carry(uint32_t exponent, size_t words, uint32_t odd, uint32_t factor, const uint64_t* in61, const uint32_t* in31, uint64_t* digits_out,
                    uint64_t* residual_out, size_t device, double* kernel_ms) {
  return guarded([&] {
    if (!in61 || !in31 || !digits_out || !residual_out) throw std::runtime_error("crt_carry: null buffer");
    mi355::crt_carry_host(exponent, words, odd, factor, in61, in31, digits_out, residual_out, int(device), kernel_ms);
  });
}
int mi355_crt_get_raw_digits(mi355_engine_handle h, size_t src, uint64_t* d, size_t count, int canonical) {
  return guarded([&] { if (!d || !crt(h)) throw std::runtime_error("get_raw_digits: needs a crt engine and a buffer"); crt(h)->get_digits(src, d, count, canonical != 0); });
}
int mi355_crt_set_raw_digits(mi355_engine_handle h, size_t dst, const uint64_t* d, size_t count) {
  return guarded([&] { if (!d || !crt(h)) throw std::runtime_error("set_raw_digits: needs a crt engine and a buffer"); crt(h)->set_digits(dst, d, count); });
}
size_t mi355_crt_transform_size(uint32_t exponent, uint32_t odd) { size_t r = 0; guarded([&] { r = mi355::crt_transform_size(exponent, odd); }); return r; }
#if defined(MI355_PROBE)
__attribute__((visibility("default"))) int mi355_probe(mi355_engine_handle h, int kind, int grid_mult, int extra_lds, int boost_pct, size_t iters, double* avg_ms, uint64_t* tl, size_t tl_words) {
  return guarded([&] { eng(h)->probe(kind, grid_mult, extra_lds, boost_pct, iters, avg_ms, tl, tl_words); });
}
#endif
size_t mi355_engine_algorithmic_bytes(mi355_engine_handle h) { size_t r = 0; guarded([&] { r = crt(h) ? crt(h)->algorithmic_bytes() : eng(h)->algorithmic_bytes(); }); return r; }

}  // extern "C"
