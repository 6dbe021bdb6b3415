/* Alias shim: exports the reference's plugin ABI (third_party/aevum/src/EngineApi.h:28-59, the 19 aevum_engine_*
 * symbols its adapter binds with dlsym, src/aevum/EngineAevum.cpp:225-243) on top of libmi355_engine.so, so that
 * the reference's own `-aevum` adapter can load the MI355X engine for an in-tree A/B run:
 *   cc -shared -fPIC -Iinclude examples/mi355_as_aevum.c -o libaevum_engine.so -Lprmers_amd -lmi355_engine -Wl,-rpath,'$ORIGIN'
 *   AEVUM_ENGINE_LIB=./libaevum_engine.so ./prmers 136279841 -prp -aevum
 * Every signature is identical; only the prefix differs. */
#include "mi355_engine.h"

#define EXPORT __attribute__((visibility("default")))

EXPORT const char* aevum_engine_version(void) { return mi355_engine_version(); }
EXPORT const char* aevum_engine_last_error(void) { return mi355_engine_last_error(); }
EXPORT int aevum_engine_resolve_fft(uint32_t exponent, const char* fft_spec, char* output, size_t output_size) {
  return mi355_engine_resolve_fft(exponent, fft_spec, output, output_size);
}
EXPORT void* aevum_engine_create(uint32_t exponent, size_t register_count, uint32_t device, int verbose, const char* fft_spec, const char* tune_dir) {
  return mi355_engine_create(exponent, register_count, device, verbose, fft_spec, tune_dir);
}
EXPORT void aevum_engine_destroy(void* h) { mi355_engine_destroy(h); }
EXPORT size_t aevum_engine_transform_size(void* h) { return mi355_engine_transform_size(h); }
EXPORT size_t aevum_engine_word_count(void* h) { return mi355_engine_word_count(h); }
EXPORT int aevum_engine_sync(void* h) { return mi355_engine_sync(h); }
EXPORT int aevum_engine_set_u32(void* h, size_t dst, uint32_t value) { return mi355_engine_set_u32(h, dst, value); }
EXPORT int aevum_engine_set_words(void* h, size_t dst, const uint32_t* words, size_t count) { return mi355_engine_set_words(h, dst, words, count); }
EXPORT int aevum_engine_get_words(void* h, size_t src, uint32_t* words, size_t count) { return mi355_engine_get_words(h, src, words, count); }
EXPORT int aevum_engine_copy(void* h, size_t dst, size_t src) { return mi355_engine_copy(h, dst, src); }
EXPORT int aevum_engine_prepare(void* h, size_t dst, size_t src) { return mi355_engine_prepare(h, dst, src); }
EXPORT int aevum_engine_square_mul(void* h, size_t reg, uint32_t factor) { return mi355_engine_square_mul(h, reg, factor); }
EXPORT int aevum_engine_mul(void* h, size_t dst, size_t src, uint32_t factor) { return mi355_engine_mul(h, dst, src, factor); }
EXPORT int aevum_engine_add(void* h, size_t dst, size_t src) { return mi355_engine_add(h, dst, src); }
EXPORT int aevum_engine_sub_reg(void* h, size_t dst, size_t src) { return mi355_engine_sub_reg(h, dst, src); }
EXPORT int aevum_engine_sub_u32(void* h, size_t dst, uint32_t value) { return mi355_engine_sub_u32(h, dst, value); }
EXPORT int aevum_engine_equal(void* h, size_t lhs, size_t rhs, int* out) { return mi355_engine_equal(h, lhs, rhs, out); }
