/* libmi355_engine.so -- C ABI of the MI355X-native Goldilocks IBDWT engine.
 *
 * Drop-in boundary for the reference's Marin backend (cherubrock-seb/PrMers).  The shapes are the
 * reference's own engine-plugin ABI (third_party/aevum/src/EngineApi.h:28-59, loaded with dlopen by
 * src/aevum/EngineAevum.cpp:225-243) under the prefix mi355_engine_, so the reference's adapter
 * pattern binds it unchanged; the C++ adapter that serves `-engine-marin` is include/mi355/engine_hip.h
 * (hook: src/marin/gpu.cpp:149).
 *
 * Conventions (same as the reference ABI, EngineApi.cpp:447-517):
 *   - int results: 1 = ok, 0 = failure; the message is in mi355_engine_last_error() (thread-local,
 *     owned by the library).  create() returns NULL on failure.  No exception ever crosses the ABI.
 *   - registers are indices 0..register_count-1; a register holds a residue mod 2^p-1, or -- after
 *     prepare(dst, src) -- an opaque multiplicand image only valid as the src of mul()
 *     (engine.h:52-60, engine_gpu.h:1695-1756).
 *   - residues cross the boundary as little-endian 32-bit words of the canonical value in
 *     [0, 2^p-1), word_count = ceil(p/32)  (EngineApi.cpp:210-218), or as IBDWT digits
 *     value | width << 32 (engine.h:24-25).
 *   - one engine = one HIP device + one stream; calls are asynchronous, sync() / any read drains the
 *     stream (engine_gpu.h:1427-1430).  An engine is not thread-safe; engines are independent.
 *   - there is NO CPU fallback: without a usable HIP device create() fails with a clear message.
 */
#ifndef MI355_ENGINE_H
#define MI355_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#define MI355_ENGINE_VERSION "0.3"   /* one version string: mi355_engine_version(), the result JSON of the callers */
#define MI355_ENGINE_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mi355_engine_handle;

/* ---- the 19 shapes of the reference plugin ABI (EngineApi.h:28-59) ---- */
MI355_ENGINE_API const char* mi355_engine_version(void);                                    /* :28 */
MI355_ENGINE_API const char* mi355_engine_last_error(void);                                 /* :29 */
/* transform plan for an exponent as text ("marin-hip:n=..:m1=..:m2=..:c=.."); needs no GPU.   :30-34 */
MI355_ENGINE_API int mi355_engine_resolve_fft(uint32_t exponent, const char* fft_spec, char* output, size_t output_size);
/* engine::create_gpu(p, reg_count, device, verbose) (engine.h:301, src/marin/gpu.cpp:52).     :36-42
   fft_spec: NULL/"" = automatic, or "m2=<rows>,c=<columns per tile>"; tune_dir is ignored. */
MI355_ENGINE_API mi355_engine_handle mi355_engine_create(uint32_t exponent, size_t register_count, uint32_t device,
                                                         int verbose, const char* fft_spec, const char* tune_dir);
MI355_ENGINE_API void mi355_engine_destroy(mi355_engine_handle handle);                      /* :44 */
MI355_ENGINE_API size_t mi355_engine_transform_size(mi355_engine_handle handle);            /* engine::get_size, engine.h:40 */
MI355_ENGINE_API size_t mi355_engine_word_count(mi355_engine_handle handle);                /* :46 */
MI355_ENGINE_API int mi355_engine_sync(mi355_engine_handle handle);                         /* engine::sync, engine.h:45 */

MI355_ENGINE_API int mi355_engine_set_u32(mi355_engine_handle handle, size_t dst, uint32_t value);   /* engine::set, engine.h:47 */
MI355_ENGINE_API int mi355_engine_set_words(mi355_engine_handle handle, size_t dst, const uint32_t* words, size_t count); /* set_mpz, engine.h:206 */
MI355_ENGINE_API int mi355_engine_get_words(mi355_engine_handle handle, size_t src, uint32_t* words, size_t count);       /* get_mpz, engine.h:173 */
MI355_ENGINE_API int mi355_engine_copy(mi355_engine_handle handle, size_t dst, size_t src);          /* engine::copy, engine.h:49 */
MI355_ENGINE_API int mi355_engine_prepare(mi355_engine_handle handle, size_t dst, size_t src);       /* set_multiplicand, engine.h:53 */
MI355_ENGINE_API int mi355_engine_square_mul(mi355_engine_handle handle, size_t reg, uint32_t factor);      /* engine.h:51 */
MI355_ENGINE_API int mi355_engine_mul(mi355_engine_handle handle, size_t dst, size_t src, uint32_t factor); /* engine.h:60 */
MI355_ENGINE_API int mi355_engine_add(mi355_engine_handle handle, size_t dst, size_t src);           /* engine.h:64 */
MI355_ENGINE_API int mi355_engine_sub_reg(mi355_engine_handle handle, size_t dst, size_t src);       /* engine.h:71 */
MI355_ENGINE_API int mi355_engine_sub_u32(mi355_engine_handle handle, size_t dst, uint32_t value);   /* engine::sub, engine.h:62 */
MI355_ENGINE_API int mi355_engine_equal(mi355_engine_handle handle, size_t lhs, size_t rhs, int* equal_out); /* is_equal, engine.h:148 */

/* ---- fused variants (overridable defaults of the reference's engine, engine.h:65-131; its GPU engine fuses them
        into the carry kernels, kernels/marin.cl:1856-2365, engine_gpu.h:1960-2110).  One sweep each here too. ---- */
/* sum_out = a + b, diff_out = a - b (engine.h:72) */
MI355_ENGINE_API int mi355_engine_addsub(mi355_engine_handle handle, size_t sum_out, size_t diff_out, size_t a, size_t b);
/* the same, each result also written to a second register (engine.h:125) */
MI355_ENGINE_API int mi355_engine_addsub_copy(mi355_engine_handle handle, size_t sum, size_t diff, size_t sum_copy, size_t diff_copy, size_t a, size_t b);
/* dst = dst * mul_src * factor + add_src; mul_src is a multiplicand (engine.h:65) */
MI355_ENGINE_API int mi355_engine_mul_add(mi355_engine_handle handle, size_t dst, size_t mul_src, size_t add_src, uint32_t factor);
/* src = src^2 * factor, dst_copy = src (engine.h:81) */
MI355_ENGINE_API int mi355_engine_square_mul_copy(mi355_engine_handle handle, size_t src, size_t dst_copy, uint32_t factor);
/* dst = dst * src * factor, dst_copy = dst (engine.h:91) */
MI355_ENGINE_API int mi355_engine_mul_copy(mi355_engine_handle handle, size_t dst, size_t src, size_t dst_copy, uint32_t factor);
/* count x { reg = reg^2 * factor; reg -= sub } -- the run of squarings a PRP (sub = 0) or Lucas-Lehmer (sub = 2) loop issues between two
   checks (src/modes/RunPrpOrLlMarin.cpp:338-409: one square_mul, and for LL one sub, per iteration).  Same result as the loop of
   mi355_engine_square_mul / mi355_engine_sub_u32 calls, issued by the library in one call (no per-iteration trip through the FFI; an LL
   subtraction rides on the next front sweep).  (A one-cooperative-launch form for transforms of at most 2^20 words was built and measured
   slower than three launches per squaring; it is not in this library: DESIGN.md 5.2c.) */
MI355_ENGINE_API int mi355_engine_square_mul_n(mi355_engine_handle handle, size_t reg, uint32_t factor, size_t count, uint32_t sub);

/* ---- rest of the engine surface the Marin callers use ---- */
/* engine::get / engine::set(Reg, uint64*) (engine.h:24-25): n digits, value | width << 32, strongly
   carried (engine_gpu.h:1534-1561).  count must equal transform_size(). */
MI355_ENGINE_API int mi355_engine_get_digits(mi355_engine_handle handle, size_t src, uint64_t* digits, size_t count);
MI355_ENGINE_API int mi355_engine_set_digits(mi355_engine_handle handle, size_t dst, const uint64_t* digits, size_t count);
/* engine::digit::res64 (engine.h:257-269) */
MI355_ENGINE_API int mi355_engine_res64(mi355_engine_handle handle, size_t src, uint64_t* res64_out);
/* raw register images: get_register_data_size / get_data / set_data / *checkpoint (engine.h:134-146).
   Images are implementation-defined (as in the reference); sizes must match exactly or the call fails. */
MI355_ENGINE_API size_t mi355_engine_register_data_size(mi355_engine_handle handle);
MI355_ENGINE_API int mi355_engine_get_data(mi355_engine_handle handle, size_t src, void* data, size_t size);
MI355_ENGINE_API int mi355_engine_set_data(mi355_engine_handle handle, size_t dst, const void* data, size_t size);
MI355_ENGINE_API size_t mi355_engine_checkpoint_size(mi355_engine_handle handle);
MI355_ENGINE_API int mi355_engine_get_checkpoint(mi355_engine_handle handle, void* data, size_t size);
MI355_ENGINE_API int mi355_engine_set_checkpoint(mi355_engine_handle handle, const void* data, size_t size);

/* ---- measurement hooks (bench.py; the reference's profiling map, ocl.h:238-247,657-675) ---- */
/* `iters` back-to-back square_mul(reg, factor) [+ sub_u32(sub) when sub != 0: the LL step x^2-2]
   timed with HIP events on the engine's own stream.  total_ms = wall of the whole batch.
   kernel_ms[k] (k < kernel_count, may be NULL) = average duration of kernel k of the chain
   (names from mi355_engine_kernel_name), from per-launch event pairs in a second instrumented batch. */
MI355_ENGINE_API int mi355_engine_time_square_mul(mi355_engine_handle handle, size_t reg, uint32_t factor, uint32_t sub,
                                                  size_t iters, double* total_ms, double* kernel_ms, size_t kernel_count);
MI355_ENGINE_API size_t mi355_engine_kernel_count(mi355_engine_handle handle);
MI355_ENGINE_API const char* mi355_engine_kernel_name(mi355_engine_handle handle, size_t k);
/* device self-test of the GF(2^64-2^32+1) primitives (add, sub, mul, lazy forms, every power-of-two shift, the
   radix-8 butterflies) against 128-bit host arithmetic; 1 = ok, 0 = mismatch or no device (see last_error) */
MI355_ENGINE_API int mi355_engine_selftest(size_t device);
/* algorithmic bytes one squaring moves (48 * n) */
MI355_ENGINE_API size_t mi355_engine_algorithmic_bytes(mi355_engine_handle handle);


/* ---- second field family, first slice ----
   The fused unweight + Garner + carry sweep of the paired-NTT squaring over GF(M61^2) x GF(M31^2) (the reference's FFT3161 carry
   kernel, third_party/aevum/src/cl/carry.cl:506-588): in61 / in31 = the two residues of every scaled, still weighted convolution
   coefficient in logical digit order (transform_words = odd_radix * 2^k of them, odd_radix in {1, 3, 9}); digits_out = the digits in
   base 2^width after the run-wise carry (widths reach 39 bits: plain u64 values); residual_out = ceil(words / 8) carries (0 or a few
   units) still to be added in front of each following run of 8 digits (cyclically).  Host buffers in and out; kernel_ms (may be NULL)
   receives the duration of the sweep. */
MI355_ENGINE_API int mi355_crt_carry(uint32_t exponent, size_t transform_words, uint32_t odd_radix, uint32_t factor, const uint64_t* in61,
                                     const uint32_t* in31, uint64_t* digits_out, uint64_t* residual_out, size_t device, double* kernel_ms);

/* ---- second field family: selected by the fft_spec ----
   mi355_engine_create(exponent, registers, device, verbose, "crt[:odd][:words=N][:h2=K]", NULL) returns an engine that computes over
   GF(M61^2) x GF(M31^2) with a prime-factor axis of radix odd in {1, 3, 9} (default 1): what the reference's Aevum plugin does behind
   the same EngineApi.h shapes (third_party/aevum/src/EngineApi.h:28-59; fft-middle.cl:663-720, pfaunpack.cl:12-56, carry.cl:506-588;
   transform sizes README.md:907-926, e.g. "crt:9" = the radix-9 family, "crt:3:words=6291456").  words = 0 / absent: the smallest
   admissible odd * 2^k.  Served by that engine: the 19 core entry points (create ... equal), get_digits / set_digits (sizes with words of
   at most 32 bits), res64, raw images and checkpoints (12 bytes per word + a kind tag per register), time_square_mul (sub must be 0), kernel_count /
   kernel_name, algorithmic_bytes, describe, the fused register operations (as compositions). */
MI355_ENGINE_API size_t mi355_crt_transform_size(uint32_t exponent, uint32_t odd_radix);
/* the engine's own digits of a crt handle: plain u64 values in base 2^width_j, logical order (canonical != 0: after the strong carry;
   0: as they are on the device, weakly carried) -- test and debugging access; mi355_engine_get_digits / set_digits on such a handle
   use the value | width << 32 form of engine::get and refuse sizes whose words exceed 32 bits */
MI355_ENGINE_API int mi355_crt_get_raw_digits(mi355_engine_handle handle, size_t src, uint64_t* digits, size_t count, int canonical);
MI355_ENGINE_API int mi355_crt_set_raw_digits(mi355_engine_handle handle, size_t dst, const uint64_t* digits, size_t count);
/* plan text of a live engine: "marin-hip:n=...:m1=...:m2=...:c=..." or "crt-hip:n=...:odd=...:m=...:h1=...:h2=...:radix8|generic" */
MI355_ENGINE_API int mi355_engine_describe(mi355_engine_handle handle, char* output, size_t output_size);

#ifdef __cplusplus
}
#endif
#endif
