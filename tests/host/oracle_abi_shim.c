/* TEST INFRASTRUCTURE ONLY: the C ABI of include/mi355_engine.h served by the CPU oracle (oracle/oracle.c), so that the C++
 * callers (examples/prp_cli.cpp over include/mi355/engine_hip.h) can be driven in the CPU test suite -- checkpoint resume,
 * Gerbicz-Li rollback after a resume, SIGINT.  Built and loaded by tests/test_host_logic.py only (-lib <this>.so); the
 * product library never sees it.  Only the entry points engine_hip binds are provided. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/oracle.h"

#define API __attribute__((visibility("default")))

typedef struct { orc_ctx* c; size_t regs, n; uint32_t p; } shim;
static const char* g_err = "";

API const char* mi355_engine_version(void) { return "oracle-shim (tests only)"; }
API const char* mi355_engine_last_error(void) { return g_err; }
API void* mi355_engine_create(uint32_t p, size_t regs, uint32_t device, int verbose, const char* spec, const char* tune) {
  (void)device; (void)verbose; (void)tune;
  if (spec && spec[0]) { g_err = "oracle shim: no transform specs"; return NULL; }
  shim* s = (shim*)calloc(1, sizeof(shim));
  s->c = orc_create(p, regs + 1);   /* one scratch register for the compositions */
  if (!s->c) { free(s); g_err = "oracle shim: orc_create failed"; return NULL; }
  s->regs = regs; s->n = orc_size(s->c); s->p = p;
  return s;
}
API void mi355_engine_destroy(void* h) { if (h) { orc_destroy(((shim*)h)->c); free(h); } }
API size_t mi355_engine_transform_size(void* h) { return ((shim*)h)->n; }
API size_t mi355_engine_word_count(void* h) { return orc_word_count(((shim*)h)->p); }
API int mi355_engine_sync(void* h) { (void)h; return 1; }
static int bad(shim* s, size_t r) { if (r >= s->regs) { g_err = "register index out of range"; return 1; } return 0; }
API int mi355_engine_set_u32(void* h, size_t d, uint32_t v) { shim* s = h; if (bad(s, d)) return 0; orc_set_u32(s->c, d, v); return 1; }
API int mi355_engine_copy(void* h, size_t d, size_t r) { shim* s = h; if (bad(s, d) || bad(s, r)) return 0; orc_copy(s->c, d, r); return 1; }
API int mi355_engine_prepare(void* h, size_t d, size_t r) { shim* s = h; if (bad(s, d) || bad(s, r)) return 0; orc_set_multiplicand(s->c, d, r); return 1; }
API int mi355_engine_square_mul(void* h, size_t r, uint32_t a) { shim* s = h; if (bad(s, r) || !a) return 0; orc_square_mul(s->c, r, a); return 1; }
API int mi355_engine_mul(void* h, size_t d, size_t r, uint32_t a) { shim* s = h; if (bad(s, d) || bad(s, r) || !a) return 0; orc_mul(s->c, d, r, a); return 1; }
API int mi355_engine_add(void* h, size_t d, size_t r) { shim* s = h; if (bad(s, d) || bad(s, r)) return 0; orc_add(s->c, d, r); return 1; }
API int mi355_engine_sub_reg(void* h, size_t d, size_t r) { shim* s = h; if (bad(s, d) || bad(s, r)) return 0; orc_sub_reg(s->c, d, r); return 1; }
API int mi355_engine_sub_u32(void* h, size_t r, uint32_t v) { shim* s = h; if (bad(s, r)) return 0; orc_sub_u32(s->c, r, v); return 1; }
API int mi355_engine_equal(void* h, size_t a, size_t b, int* out) {
  shim* s = h; if (bad(s, a) || bad(s, b)) return 0;
  const size_t wc = orc_word_count(s->p);
  uint32_t* x = malloc(wc * 4); uint32_t* y = malloc(wc * 4);
  orc_get_words(s->c, a, x, wc); orc_get_words(s->c, b, y, wc);
  *out = memcmp(x, y, wc * 4) == 0;
  free(x); free(y);
  return 1;
}
API int mi355_engine_get_digits(void* h, size_t r, uint64_t* d, size_t n) { shim* s = h; if (bad(s, r) || n != s->n) return 0; orc_get_digits(s->c, r, d); return 1; }
API int mi355_engine_set_digits(void* h, size_t r, const uint64_t* d, size_t n) { shim* s = h; if (bad(s, r) || n != s->n) return 0; orc_set_digits(s->c, r, d); return 1; }
API size_t mi355_engine_register_data_size(void* h) { return ((shim*)h)->n * 8; }
API int mi355_engine_get_data(void* h, size_t r, void* d, size_t sz) { shim* s = h; if (bad(s, r) || sz != s->n * 8) return 0; orc_get_raw(s->c, r, d); return 1; }
API int mi355_engine_set_data(void* h, size_t r, const void* d, size_t sz) { shim* s = h; if (bad(s, r) || sz != s->n * 8) return 0; orc_set_raw(s->c, r, d); return 1; }
API size_t mi355_engine_checkpoint_size(void* h) { shim* s = h; return s->regs * s->n * 8; }
API int mi355_engine_get_checkpoint(void* h, void* d, size_t sz) {
  shim* s = h; if (sz != s->regs * s->n * 8) return 0;
  for (size_t r = 0; r < s->regs; ++r) orc_get_raw(s->c, r, (uint64_t*)d + r * s->n);
  return 1;
}
API int mi355_engine_set_checkpoint(void* h, const void* d, size_t sz) {
  shim* s = h; if (sz != s->regs * s->n * 8) return 0;
  for (size_t r = 0; r < s->regs; ++r) orc_set_raw(s->c, r, (const uint64_t*)d + r * s->n);
  return 1;
}
/* the fused variants as their base-class compositions (include/marin/engine.h:65-131) */
API int mi355_engine_addsub(void* h, size_t so, size_t dout, size_t a, size_t b) {
  shim* s = h; const size_t t = s->regs;
  orc_copy(s->c, t, a); orc_sub_reg(s->c, t, b);
  if (so != a) orc_copy(s->c, so, a);
  orc_add(s->c, so, b); orc_copy(s->c, dout, t);
  return 1;
}
API int mi355_engine_addsub_copy(void* h, size_t s1, size_t d1, size_t s2, size_t d2, size_t a, size_t b) {
  if (!mi355_engine_addsub(h, s1, d1, a, b)) return 0;
  shim* s = h; orc_copy(s->c, s2, s1); orc_copy(s->c, d2, d1);
  return 1;
}
API int mi355_engine_mul_add(void* h, size_t d, size_t ms, size_t as, uint32_t f) {
  shim* s = h; const size_t t = s->regs;
  orc_copy(s->c, t, as); orc_mul(s->c, d, ms, f); orc_add(s->c, d, t);
  return 1;
}
API int mi355_engine_square_mul_copy(void* h, size_t r, size_t cp, uint32_t f) { shim* s = h; orc_square_mul(s->c, r, f); orc_copy(s->c, cp, r); return 1; }
API int mi355_engine_mul_copy(void* h, size_t d, size_t r, size_t cp, uint32_t f) { shim* s = h; orc_mul(s->c, d, r, f); orc_copy(s->c, cp, d); return 1; }
