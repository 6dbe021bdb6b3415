/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's Marin IBDWT squaring path over GF(2^64-2^32+1).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * library (prmers_amd/csrc, libmi355_engine.so) never links, calls or falls back to it.
 *
 * Parity status: PINNED.  Checked (tests/test_oracle_*.py) against
 *   - the reference's own known answers: unit_tests.sh:5-14 (prime / composite exponents),
 *     unit_tests.sh:140-141 (M100003 res64 + res2048), unit_tests.sh:167-177 (11 intermediate
 *     res64 of M11213), tests/test_aevum_reg_adapter.cpp:32-86 (op-level expectations);
 *   - tables produced by the reference's own host headers compiled in place
 *     (oracle/_ref/ref_tables from include/marin/ibdwt.h + arith.h, see oracle/Makefile);
 *   - independent big-integer arithmetic (Python ints / libgmp).
 * The reference's device kernels (OpenCL) cannot execute in the build container (no OpenCL device),
 * so raw register images are not compared, only canonical digits / words / res64.
 */
#ifndef PRMERS_ORACLE_H
#define PRMERS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* field helpers (arith.h:27-72) */
uint64_t orc_mod_add(uint64_t a, uint64_t b);
uint64_t orc_mod_sub(uint64_t a, uint64_t b);
uint64_t orc_mod_mul(uint64_t a, uint64_t b);
uint64_t orc_mod_pow(uint64_t a, uint64_t e);
uint64_t orc_mod_invert(uint64_t a);

/* ibdwt.h:17-43 */
size_t orc_transform_size(uint32_t p);

orc_ctx* orc_create(uint32_t p, size_t reg_count);
void orc_destroy(orc_ctx* c);
size_t orc_size(const orc_ctx* c);                       /* n */
uint32_t orc_exponent(const orc_ctx* c);
void orc_widths(const orc_ctx* c, uint8_t* out);         /* n digit widths, ibdwt.h:127-132 */
void orc_weights(const orc_ctx* c, uint64_t* w, uint64_t* winv); /* natural digit order, ibdwt.h:134-143 */
int orc_threads(void);                                   /* OpenMP threads used by the transforms */
void orc_set_threads(int t);

/* engine operations (engine.h:47-71; engine_gpu.h:1432-1630,1695-1884,2085-2098) */
void orc_set_u32(orc_ctx* c, size_t dst, uint32_t a);
void orc_copy(orc_ctx* c, size_t dst, size_t src);
void orc_square_mul(orc_ctx* c, size_t reg, uint32_t a);
void orc_set_multiplicand(orc_ctx* c, size_t dst, size_t src);
void orc_mul(orc_ctx* c, size_t dst, size_t src, uint32_t a);
void orc_sub_u32(orc_ctx* c, size_t reg, uint32_t a);
void orc_add(orc_ctx* c, size_t dst, size_t src);
void orc_sub_reg(orc_ctx* c, size_t dst, size_t src);

/* digit I/O: d[k] = value | width << 32, strong carry (engine_gpu.h:1534-1561, 1452-1485) */
void orc_get_digits(const orc_ctx* c, size_t src, uint64_t* d);
void orc_set_digits(orc_ctx* c, size_t dst, const uint64_t* d);
/* raw weighted register (engine_gpu.h:2134-2148) */
void orc_get_raw(const orc_ctx* c, size_t src, uint64_t* x);
void orc_set_raw(orc_ctx* c, size_t dst, const uint64_t* x);

/* engine::digit (engine.h:257-295) on an encoded digit vector */
uint64_t orc_digits_res64(const uint64_t* d, size_t n);
int orc_digits_equal_to(const uint64_t* d, size_t n, uint64_t a);
int orc_digits_equal_to_Mp(const uint64_t* d, size_t n);

/* AlgoUtils.hpp:165-223: words = ceil(p/32) little-endian 32-bit words of the digit vector */
size_t orc_word_count(uint32_t p);
void orc_pack_words(const uint64_t* d, size_t n, uint32_t p, uint32_t* words);
void orc_prp3_div9(uint32_t p, uint32_t* words, size_t count);
void orc_format_res64(const uint32_t* words, size_t count, char out[17]);
void orc_format_res2048(const uint32_t* words, size_t count, char out[513]);
/* canonical words of x mod 2^p-1 (all-ones -> 0), as engine::get_mpz would export (engine.h:173-203) */
void orc_get_words(const orc_ctx* c, size_t src, uint32_t* words, size_t count);
void orc_set_words(orc_ctx* c, size_t dst, const uint32_t* words, size_t count);

#ifdef __cplusplus
}
#endif
#endif
