// gfx950 kernels, register-resident radix-5 column set ("v5"): column tiles of M1 = 1280 = 5 x 256 with C = 4 -- the columns of n = 5 2^21
// (BASELINE configs[3] on the Goldilocks path), 5 2^22 and 5 2^23.  Reference: forward80_0 / backward80_0, kernels/marin.cl:1019-1040,
// engine_gpu.h:1619.  Helpers shared with kernels_v2.hip live in kernels_v2_common.hpp.
#include "kernels_v2_common.hpp"

namespace mi355 {

// ---------------------------------------------------------------------------------------------
// Columns of M1 = 1280 = 5 x 256 (n = 5 * 2^21: BASELINE configs[3] on the Goldilocks path), C = 4 pairs per run: one tile
// of 5120 pairs per work-group of 640 threads, 8 pairs per thread, the radix-5 stage on 512 threads with 10.
// The reference serves this size with forward80_0 / backward80_0 (kernels/marin.cl:1019-1040, engine_gpu.h:1619).
//   i1 = (256 d0 + 5 r) mod 1280 (prime-factor input map, round 4),  r = 64 e1 + 8 e2 + e3;  slot (k0, kr), kr = k1 + 4 k2 + 32 k3, holds the
//   column frequency (256 k0 + 1025 kr) mod 1280 (Shape<J>::PU / PV; rounds 2-3: i1 = 256 d0 + r, frequency k0 + 5 kr, a twiddle after stage A)
//   L   thread t: runs i1 = t, t + 640 (digits -> carry-in -> weight)
//   A   thread u < 512: groups g = u, u + 512 (r = g / 4, c = g % 4): DFT5 over d0 (no twiddle: 5 and 256 are coprime)
//   B1  thread (k0 | e2 | e3 | c/2): DFT4 over e1, twiddle omega_256^(k1 (8 e2 + e3))
//   B2  thread (k0 | k1 | e3 | c):   DFT8 over e2, twiddle omega_64^(k2 e3)
//   B3  thread (k0 | k1 | k2 | c):   DFT8 over e3, then the four-step twiddle chain (ratio omega_m^(1025 x 32 i2)) and the store to
//       row k0 256 + bitrev8(kr), the row order of kernels.hip freq1
// The factor 5 leaves no digit that is uniform over a wavefront, so the seams inside the power-of-two part are table
// multiplications (omega_1280 powers from UT1) instead of the compile-time shifts of the 512 R shapes.  LDS carries one
// plane (8 bytes per slot) at a time: 46 KiB per work-group, so that two of them share a CU.
// The back sweep is the mirror image, ending with the unweighting and the carry along the thread's two runs.
// ---------------------------------------------------------------------------------------------
// Round 4b: the same kernels serve columns of M1 = 2560 = 5 x 512 with C = 2 (template parameter J = 1: 5.8.8.8 instead of 5.4.8.8) -- the
// columns of n = 5 2^22 (exponents of 100 M decimal digits: 2560 x 4096, so that the rows stay 4096 wide) and of n = 5 2^23 (2560 x 8192).
// Same tile (5120 pairs), same thread count; thread t owns four runs of two pairs (i1 = t + 640 d1), the e1 digit has eight values and its
// stage is a radix-8 on one column (c = t & 1) instead of a radix-4 on two.  Reference sizes: engine_gpu.h:1620-1624.
namespace v5 {
constexpr uint32_t kThreads = 640, kTile = 5120;
template <int J> struct Shape {
  static constexpr uint32_t C = 4u >> J, LC = 2 - J, L = 256u << J, M1 = 5 * L, R1 = 4u << J;   // pairs per run, log2 C, M1 / 5, radix of the e1 stage
  static constexpr uint32_t NR = 2u << J, ND = 2 * C, QW = ND / 4;                                // runs per thread, digits per run, uint4 per run
  static constexpr int SA = 2, SAb = J ? -1 : 5, SB1 = 1, SB2 = J ? 3 : 1, SB3 = 3;                 // LDS slot maps per exchange (tools/lds_census5.py); SAb: stage A of the back sweep
  // Prime-factor form of the 5 x L split (Good-Thomas: 5 and L are coprime, so no twiddle stands between the radix-5 stage and the power-of-two
  // part): input i1 = (L d0 + 5 r) mod M1, output slot (k0, kr) holds the frequency (PU k0 + PV kr) mod M1 with PU = L (L^-1 mod 5),
  // PV = 5 (5^-1 mod L) = 1025 for L = 256 and 512.  The engine passes PU / PV to every kernel that needs a column frequency (DevPlan.lab_u/v,
  // kernels.hpp col_label).  Saves the four table products per radix-5 butterfly and direction of the mixed-radix form: C4 front 56.3 -> 53.0 us, back 62.3 -> 58.2,
  // 0.2000 -> 0.1925 ms per squaring; n = 5 2^20 -3 %, 5 2^22 -2.6 % (same-box A/B, profiles/r04_ab_pfa5.txt).
  static constexpr uint32_t PU = J ? 1536u : 256u, PV = 1025u;
};
static_assert((Shape<0>::PU % 5) == 1 && (Shape<0>::PU % 256) == 0 && (Shape<0>::PV % 256) == 1 && (Shape<0>::PV % 5) == 0, "CRT idempotents, L = 256");
static_assert((Shape<1>::PU % 5) == 1 && (Shape<1>::PU % 512) == 0 && (Shape<1>::PV % 512) == 1 && (Shape<1>::PV % 5) == 0, "CRT idempotents, L = 512");
// Launched with 768 threads: twelve waves spread evenly over the four SIMDs of a CU, the last two leave at once.  A work-group of ten waves
// (3 + 3 + 2 + 2) is not placed next to a resident one for up to 17 us after a slot has become free (profiles/r03_probe_c4.md: 44 % of
// the dispatches of a launch, a CU then runs one group for half of its time); twelve are placed within 2 us like the 512-thread groups.
constexpr uint32_t kLaunchThreads = 768;
#if defined(MI355_LDS_ADD3)
constexpr uint32_t kPlaneWords = kTile + kTile / 32 + 8;
#else
constexpr uint32_t kPlaneWords = kTile;
#endif
// the exchange plane, then a copy of the omega_1280 table (10 KiB): the seams of this shape are table multiplications and their roots
// come out of LDS (~100 cycles) instead of L2 (several hundred, exposed at every stage)
constexpr uint32_t lds_bytes(uint32_t m1) { return (kPlaneWords + m1) * 8; }
// LDS slot (8 bytes) of tile element i for an exchange: round 4 gives every exchange its own map i ^ ((i >> S) & 31), conflict-free for
// the lane groups of 64-bit accesses (stores: four groups of 16 lanes on 32 banks, loads: two groups of 32 lanes on 64 banks;
// MI355X_MICROARCH.md, LDS) in both directions of that exchange.  The single skew i + i / 32 of rounds 2-3 left 44 % of the LDS cycles of
// these kernels to bank conflicts (SQ_LDS_BANK_CONFLICT 4.03 M of SQ_LDS_IDX_ACTIVE 9.07 M per launch at C4: the last exchange of the front
// and the first of the back were 4-way conflicted).  -DMI355_LDS_ADD3 restores it for A/B builds.
template <int S>
__device__ __forceinline__ uint32_t ph(uint32_t i) {   // S < 0: identity
#if defined(MI355_LDS_ADD3)
  return i + (i >> 5);
#else
  if constexpr (S < 0) return i;
  else return i ^ ((i >> S) & 31u);
#endif
}
template <uint32_t M1>
__device__ __forceinline__ const uint64_t* stage_roots(const DevPlan& pl, uint64_t* X) {
  uint64_t* R = X + kPlaneWords;
  for (uint32_t i = threadIdx.x; i < M1; i += kThreads) R[i] = pl.UT1[i];
  return R;   // visible after the first barrier of the first exchange
}

// write 8 (or 10) values to element ids wi[], read ids ri[]; one plane after the other
template <int NW, int NR, int S, class WI, class RI>
__device__ __forceinline__ void exchange(uint64_t* X, const v2::P2* in, v2::P2* out, bool writer, bool reader, WI wi, RI ri) {
  v2::lds_barrier();
  if (writer) { _Pragma("unroll") for (int k = 0; k < NW; ++k) X[ph<S>(wi(k))] = in[k].a; }
  v2::lds_barrier();
  uint64_t ta[NR];
  if (reader) { _Pragma("unroll") for (int k = 0; k < NR; ++k) ta[k] = X[ph<S>(ri(k))]; }
  v2::lds_barrier();
  if (writer) { _Pragma("unroll") for (int k = 0; k < NW; ++k) X[ph<S>(wi(k))] = in[k].b; }
  v2::lds_barrier();
  if (reader) { _Pragma("unroll") for (int k = 0; k < NR; ++k) out[k] = {ta[k], X[ph<S>(ri(k))]}; }
}

__device__ __forceinline__ void dft5p(v2::P2 (&x)[5], const uint64_t (&c5)[4], bool inverse) {
  // kernels.hip dft5, on the two planes (Winograd: 4 table multiplications + one shift per plane)
  using v2::P2;
  auto add = [](P2 p, P2 q) { return P2{gf::add(p.a, q.a), gf::add(p.b, q.b)}; };
  auto sub = [](P2 p, P2 q) { return P2{gf::sub(p.a, q.a), gf::sub(p.b, q.b)}; };
  auto mul = [](P2 p, uint64_t w) { return P2{gf::mul(p.a, w), gf::mul(p.b, w)}; };
  const P2 t1 = add(x[1], x[4]), t2 = add(x[2], x[3]), t3 = sub(x[1], x[4]), t4 = sub(x[2], x[3]);
  const P2 t5 = add(t1, t2);
  const P2 A = add(x[0], P2{gf::mul_pow2(t5.a, 94), gf::mul_pow2(t5.b, 94)});
  const P2 m2 = mul(sub(t1, t2), c5[0]);
  const P2 B1 = add(A, m2), B2 = sub(A, m2);
  const P2 m3 = mul(add(t3, t4), c5[1]), m4 = mul(t4, c5[2]), m5 = mul(t3, c5[3]);
  const P2 Pp = add(m3, m4), Q = sub(m5, m3);
  x[0] = add(x[0], t5);
  if (!inverse) { x[1] = add(B1, Pp); x[4] = sub(B1, Pp); x[2] = add(B2, Q); x[3] = sub(B2, Q); }
  else          { x[1] = sub(B1, Pp); x[4] = add(B1, Pp); x[2] = sub(B2, Q); x[3] = add(B2, Q); }
}

template <int J> __device__ __forceinline__ uint32_t brev_l(uint32_t k) { return __brev(k) >> (24 - J); }   // bit reversal over log2 L = 8 + J bits

template <int J>
__global__ void __launch_bounds__(kLaunchThreads, 2) k1_cols5(DevPlan pl, const uint32_t* __restrict__ digits, const uint64_t* __restrict__ cbuf_in, uint32_t sub,
                                                   uint64_t* __restrict__ Wout) {
  using v2::P2;
  using S = Shape<J>;
  constexpr uint32_t C = S::C, LC = S::LC, L = S::L, M1 = S::M1, R1 = S::R1, ND = S::ND;
  if (threadIdx.x >= kThreads) return;   // the two padding waves (kLaunchThreads): ended waves do not count at a barrier
  uint64_t* X = reinterpret_cast<uint64_t*>(v2::smem_v2);
  // tile order: plain with runs of four pairs (C4: 55.6 us against 57 with the XCD-contiguous order), XCD-contiguous with runs of two (the four
  // tiles that share a 128-byte line of the work buffer then meet in one L2: n = 5 2^22 front sweep 178 -> 162 us, same box); MI355_TUNE bit 5 swaps
  const uint32_t t = threadIdx.x, T = ((J == 1) != ((pl.tune & 32) != 0)) ? v2::tile_of_block(pl, PROBE_BLOCK(pl), PROBE_GRID(pl)) : PROBE_BLOCK(pl);
  v2::boost_if_late(pl.boost_tiles);
  PROBE_BEGIN(pl)
  const uint64_t* __restrict__ UT = stage_roots<M1>(pl, X);
  // ---- L: the thread's runs ----
  P2 x[8];
  {
    const uint32_t di = pl.DI[size_t(T) * kThreads + t];
    const uint32_t nowrap = ~di;
#pragma unroll
    for (uint32_t d1 = 0; d1 < S::NR; ++d1) {
      const uint32_t i1 = t + kThreads * d1;
      uint32_t dg[ND];
      {
        const uint4* src = reinterpret_cast<const uint4*>(digits) + (size_t(T) * M1 + i1) * S::QW;
#pragma unroll
        for (uint32_t q = 0; q < S::QW; ++q) { const uint4 v = src[q]; dg[4 * q] = v.x; dg[4 * q + 1] = v.y; dg[4 * q + 2] = v.z; dg[4 * q + 3] = v.w; }
      }
      if (cbuf_in) v2::apply_carry_in<ND>(pl, di, int(d1), v2::carry_in_of(pl, cbuf_in, T, i1), dg);
      const uint64_t tah = gf::half(pl.TA[i1]), tah1 = gf::half(pl.TA[M1 + i1]);
#pragma unroll
      for (uint32_t c = 0; c < C; ++c) {
        const uint32_t idx = d1 * ND + 2 * c;
        x[d1 * C + c] = {gf::mul_u32(tah, dg[2 * c] << ((nowrap >> (2 * idx + 1)) & 1u)), gf::mul_u32(tah1, dg[2 * c + 1] << ((nowrap >> (2 * idx + 3)) & 1u))};
      }
      if (sub != 0 && T == 0 && i1 == 0) x[0].a = gf::sub(x[0].a, uint64_t(sub));
    }
  }
  // ---- A: DFT5 over d0 (tile element id = i1 C + c; group g = r C + c) ----
  P2 y[10];
  exchange<8, 10, S::SA>(X, x, y, true, t < 512,
                  [&](int k) { return (t + kThreads * (uint32_t(k) >> LC)) * C + (uint32_t(k) & (C - 1)); },
                  [&](int k) { const uint32_t g = t + 512u * (k / 5); return ((L * (k % 5) + 5 * (g >> LC)) % M1) * C + (g & (C - 1)); });   // i1 = (L d0 + 5 r) mod M1
  if (t < 512) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      P2 z[5] = {y[5 * q], y[5 * q + 1], y[5 * q + 2], y[5 * q + 3], y[5 * q + 4]};
      dft5p(z, pl.W5c, false);   // (prime-factor form: no twiddle between this stage and the power-of-two part)
#pragma unroll
      for (int k0 = 0; k0 < 5; ++k0) y[5 * q + k0] = z[k0];
    }
  }
  // ---- B1: DFT over e1 (radix 4 on two columns, or radix 8 on one) ----
  {
    const uint32_t chi = t & 1, e3 = (t >> 1) & 7, e2 = (t >> 4) & 7, k0 = t >> 7;
    auto id1 = [&](int k) {   // (k0 | e1 | e2 | e3 | c): register k is (e1, c) = (k >> 1, 2 chi + (k & 1)) or (k, chi)
      const uint32_t e1 = J ? uint32_t(k) : uint32_t(k) >> 1, cc = J ? chi : 2 * chi + (uint32_t(k) & 1);
      return (k0 * L + 64 * e1 + 8 * e2 + e3) * C + cc;
    };
    exchange<10, 8, S::SB1>(X, y, x, t < 512, true, [&](int k) { return 1024u * (k % 5) + t + 512u * (k / 5); }, id1);
    const uint32_t rr = 8 * e2 + e3;
    if constexpr (J == 0) {
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        v2::dft4<false>(x[cl].a, x[2 + cl].a, x[4 + cl].a, x[6 + cl].a);
        v2::dft4<false>(x[cl].b, x[2 + cl].b, x[4 + cl].b, x[6 + cl].b);
      }
#pragma unroll
      for (int k1 = 1; k1 < 4; ++k1) {
        const uint64_t w = UT[5 * k1 * rr];   // omega_L = omega_M1^5
        x[2 * k1] = v2::p2_mul(x[2 * k1], w); x[2 * k1 + 1] = v2::p2_mul(x[2 * k1 + 1], w);
      }
    } else {
      v2::dft8p<false, 2>(x);
#pragma unroll
      for (int k1 = 1; k1 < 8; ++k1) x[k1] = v2::p2_mul(x[k1], UT[5 * k1 * rr]);
      x[0] = {gf::fold(x[0].a), gf::fold(x[0].b)};
    }
    // ---- B2: DFT8 over e2 ----
    P2 z[8];
    const uint32_t c = t & (C - 1), f3 = (t >> LC) & 7, f1 = (t >> (LC + 3)) & (R1 - 1), f0 = t >> 7;   // reader (k0 | k1 | e3 | c)
    exchange<8, 8, S::SB2>(X, x, z, true, true, id1, [&](int k) { return (f0 * L + 64 * f1 + 8 * uint32_t(k) + f3) * C + c; });
    v2::dft8p<false, 1>(z);
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) z[k2] = v2::p2_mul(z[k2], UT[(M1 / 64) * k2 * f3]);   // omega_64 = omega_M1^(M1 / 64)
    z[0] = {gf::fold(z[0].a), gf::fold(z[0].b)};
    // ---- B3: DFT8 over e3 ----
    const uint32_t g2 = f3;   // reader (k0 | k1 | k2 | c): same decode, e3's place holds k2
    exchange<8, 8, S::SB3>(X, z, x, true, true,
                   [&](int k) { return (f0 * L + 64 * f1 + 8 * uint32_t(k) + f3) * C + c; },
                   [&](int k) { return (f0 * L + 64 * f1 + 8 * g2 + uint32_t(k)) * C + c; });
    v2::dft8p<false, 2>(x);
    const uint32_t i2 = C * T + c;
    uint64_t ca = pl.F0f[size_t(T) * kThreads + t];
    const uint64_t B = pl.FBf[i2];
    P2* W = reinterpret_cast<P2*>(Wout);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t row = f0 * L + brev_l<J>(f1 + R1 * g2 + 8 * R1 * j);   // slot (k0, kr = k1 + R1 k2 + 8 R1 k3): column frequency PU k0 + PV kr
      W[size_t(row) * pl.M2 + i2] = {gf::mul(x[j].a, ca), gf::mul(x[j].b, ca)};
      if (j < 7) ca = gf::mul(ca, B);
    }
  }
  PROBE_END(pl)
}

template <int J, bool EXT>
__global__ void __launch_bounds__(kLaunchThreads, 2) k3_cols5(DevPlan pl, const uint64_t* __restrict__ Win, uint32_t* __restrict__ digits, uint64_t* __restrict__ cbuf,
                                                   uint32_t a, uint64_t scale, BackExt ext) {
  using v2::P2;
  using S = Shape<J>;
  constexpr uint32_t C = S::C, LC = S::LC, L = S::L, M1 = S::M1, R1 = S::R1, ND = S::ND;
  if (threadIdx.x >= kThreads) return;   // the two padding waves (kLaunchThreads)
  uint64_t* X = reinterpret_cast<uint64_t*>(v2::smem_v2);
  const uint32_t t = threadIdx.x, T = v2::tile_of_block(pl, PROBE_BLOCK(pl), PROBE_GRID(pl));
  v2::boost_if_late(pl.boost_tiles);
  PROBE_BEGIN(pl)
  const uint64_t* __restrict__ UT = stage_roots<M1>(pl, X);
  const uint32_t c = t & (C - 1), g2 = (t >> LC) & 7, f1 = (t >> (LC + 3)) & (R1 - 1), f0 = t >> 7, f3 = g2;
  P2 x[8], z[8];
  {
    const uint32_t i2 = C * T + c;
    uint64_t ca = pl.F0i[size_t(T) * kThreads + t];
    const uint64_t B = pl.FBi[i2];
    if (scale != 1) ca = gf::mul(ca, scale);
    const P2* W = reinterpret_cast<const P2*>(Win);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = W[size_t(f0 * L + brev_l<J>(f1 + R1 * g2 + 8 * R1 * j)) * pl.M2 + i2];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = {gf::mul(x[j].a, ca), gf::mul(x[j].b, ca)}; if (j < 7) ca = gf::mul(ca, B); }
  }
  v2::dft8p<true, 2>(x);   // k3 -> e3; multiplied by the seam next
  exchange<8, 8, S::SB3>(X, x, z, true, true,
                 [&](int k) { return (f0 * L + 64 * f1 + 8 * g2 + uint32_t(k)) * C + c; },
                 [&](int k) { return (f0 * L + 64 * f1 + 8 * uint32_t(k) + f3) * C + c; });
#pragma unroll
  for (int k2 = 1; k2 < 8; ++k2) { const uint32_t e = (M1 / 64) * k2 * f3; z[k2] = v2::p2_mul(z[k2], UT[e ? M1 - e : 0]); }
  z[0] = {gf::fold(z[0].a), gf::fold(z[0].b)};
  v2::dft8p<true, 2>(z);   // k2 -> e2
  const uint32_t chi = t & 1, e3 = (t >> 1) & 7, e2 = (t >> 4) & 7, k0 = t >> 7;
  auto id1 = [&](int k) {
    const uint32_t e1 = J ? uint32_t(k) : uint32_t(k) >> 1, cc = J ? chi : 2 * chi + (uint32_t(k) & 1);
    return (k0 * L + 64 * e1 + 8 * e2 + e3) * C + cc;
  };
  exchange<8, 8, S::SB2>(X, z, x, true, true, [&](int k) { return (f0 * L + 64 * f1 + 8 * uint32_t(k) + f3) * C + c; }, id1);
  {
    const uint32_t rr = 8 * e2 + e3;
    if constexpr (J == 0) {
      x[0] = {gf::fold(x[0].a), gf::fold(x[0].b)}; x[1] = {gf::fold(x[1].a), gf::fold(x[1].b)};
#pragma unroll
      for (int k1 = 1; k1 < 4; ++k1) {
        const uint32_t e = 5 * k1 * rr;
        const uint64_t w = UT[e ? M1 - e : 0];
        x[2 * k1] = v2::p2_mul(x[2 * k1], w); x[2 * k1 + 1] = v2::p2_mul(x[2 * k1 + 1], w);
      }
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        v2::dft4<true>(x[cl].a, x[2 + cl].a, x[4 + cl].a, x[6 + cl].a);
        v2::dft4<true>(x[cl].b, x[2 + cl].b, x[4 + cl].b, x[6 + cl].b);
      }
    } else {
      x[0] = {gf::fold(x[0].a), gf::fold(x[0].b)};
#pragma unroll
      for (int k1 = 1; k1 < 8; ++k1) { const uint32_t e = 5 * k1 * rr; x[k1] = v2::p2_mul(x[k1], UT[e ? M1 - e : 0]); }
      v2::dft8p<true>(x);   // k1 -> e1, canonical: the k0 = 0 term goes into the radix-5 butterfly as it is
    }
  }
  P2 y[10];
  exchange<8, 10, S::SB1>(X, x, y, true, t < 512, id1, [&](int k) { return 1024u * (k % 5) + t + 512u * (k / 5); });
  if (t < 512) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      P2 w5[5] = {y[5 * q], y[5 * q + 1], y[5 * q + 2], y[5 * q + 3], y[5 * q + 4]};
      dft5p(w5, pl.W5c, true);
#pragma unroll
      for (int d0 = 0; d0 < 5; ++d0) y[5 * q + d0] = w5[d0];
    }
  }
  exchange<10, 8, S::SAb>(X, y, x, t < 512, true,
                  [&](int k) { const uint32_t g = t + 512u * (k / 5); return ((L * (k % 5) + 5 * (g >> LC)) % M1) * C + (g & (C - 1)); },
                  [&](int k) { return (t + kThreads * (uint32_t(k) >> LC)) * C + (uint32_t(k) & (C - 1)); });
  // ---- unweight, x a, carry along the thread's runs ----
  const uint32_t di = pl.DI[size_t(T) * kThreads + t];
#pragma unroll
  for (uint32_t d1 = 0; d1 < S::NR; ++d1) {
    const uint32_t i1 = t + kThreads * d1;
    const uint64_t tai_e = pl.TAi[i1], tai_o = pl.TAi[M1 + i1];
    const uint64_t tai2_e = gf::dbl(tai_e), tai2_o = gf::dbl(tai_o);
    uint32_t ad[ND];
#pragma unroll
    for (uint32_t k = 0; k < ND; ++k) ad[k] = 0;
    if (EXT && ext.add_digits) {
      const uint4* src = reinterpret_cast<const uint4*>(ext.add_digits) + (size_t(T) * M1 + i1) * S::QW;
#pragma unroll
      for (uint32_t q = 0; q < S::QW; ++q) { const uint4 v = src[q]; ad[4 * q] = v.x; ad[4 * q + 1] = v.y; ad[4 * q + 2] = v.z; ad[4 * q + 3] = v.w; }
      if (ext.add_cbuf) v2::apply_carry_in<ND>(pl, di, int(d1), v2::carry_in_of(pl, ext.add_cbuf, T, i1), ad);
    }
    uint64_t carry = 0;
    uint32_t dg[ND];
#pragma unroll
    for (uint32_t k = 0; k < ND; ++k) {
      const uint32_t bits = di >> (2 * (d1 * ND + k));
      const uint32_t width = pl.q + (bits & 1u);
      const bool wrap = (bits & 2u) != 0;
      const P2 v = x[C * d1 + (k >> 1)];
      const uint64_t u = (k & 1) ? gf::mul(v.b, wrap ? tai2_o : tai_o) : gf::mul(v.a, wrap ? tai2_e : tai_e);
      const uint64_t mask = (uint64_t(1) << width) - 1;
      if (a == 1) {
        const uint64_t r = u + carry + (EXT ? ad[k] : 0u);
        dg[k] = __builtin_amdgcn_ubfe(uint32_t(r), 0u, width);
        carry = r >> width;
      } else {
        const uint64_t dlo = u & mask, chi2 = u >> width;
        const uint64_t r = dlo * a + carry + (EXT ? ad[k] : 0u);
        dg[k] = uint32_t(r & mask);
        carry = (r >> width) + chi2 * a;
      }
    }
    uint4* dst = reinterpret_cast<uint4*>(digits) + (size_t(T) * M1 + i1) * S::QW;
#pragma unroll
    for (uint32_t q = 0; q < S::QW; ++q) dst[q] = make_uint4(dg[4 * q], dg[4 * q + 1], dg[4 * q + 2], dg[4 * q + 3]);
    cbuf[size_t(T) * M1 + i1] = carry;
    if (EXT && ext.digits2) {
      uint4* d2 = reinterpret_cast<uint4*>(ext.digits2) + (size_t(T) * M1 + i1) * S::QW;
#pragma unroll
      for (uint32_t q = 0; q < S::QW; ++q) d2[q] = make_uint4(dg[4 * q], dg[4 * q + 1], dg[4 * q + 2], dg[4 * q + 3]);
      ext.cbuf2[size_t(T) * M1 + i1] = carry;
    }
  }
  PROBE_END(pl)
}

// chain starts omega_m^(i2 (PU k0 + PV (k1 + R1 k2))) TB[2 i2] and ratios omega_m^(PV 8 R1 i2) of the B3 thread map (and inverses with TBi)
template <int J>
__global__ void __launch_bounds__(640) k_build_f0(DevPlan pl, uint64_t* __restrict__ f0f, uint64_t* __restrict__ f0i, uint64_t* __restrict__ fbf, uint64_t* __restrict__ fbi) {
  using S = Shape<J>;
  const uint32_t t = threadIdx.x, T = blockIdx.x;
  const uint32_t c = t & (S::C - 1), g2 = (t >> S::LC) & 7, f1 = (t >> (S::LC + 3)) & (S::R1 - 1), f0 = t >> 7, i2 = S::C * T + c;
  const uint64_t ea = uint64_t(i2) * (S::PU * f0 + S::PV * (f1 + S::R1 * g2)) % pl.m;   // i2 x the slot's frequency label (kernels.hpp col_label)
  f0f[size_t(T) * kThreads + t] = gf::mul(v2::tw_lookup(pl, ea), pl.TB[2 * i2]);
  f0i[size_t(T) * kThreads + t] = gf::mul(v2::tw_lookup(pl, ea ? pl.m - ea : 0), pl.TBi[2 * i2]);
  if (t < S::C) {
    const uint64_t eb = uint64_t(i2) * (S::PV * 8 * S::R1) % pl.m;   // the label advances by PV 8 R1 per step of k3
    fbf[i2] = v2::tw_lookup(pl, eb);
    fbi[i2] = v2::tw_lookup(pl, eb ? pl.m - eb : 0);
  }
}
}  // namespace v5

static bool v5_j1(const DevPlan& pl) { return pl.M1 == 2560; }
void v5_pfa(const DevPlan& pl, uint32_t* u, uint32_t* v) { *u = v5_j1(pl) ? v5::Shape<1>::PU : v5::Shape<0>::PU; *v = v5::Shape<0>::PV; }
size_t v5_threads_per_tile() { return v5::kThreads; }
hipError_t v5_configure() {
  for (const void* f : {reinterpret_cast<const void*>(v5::k1_cols5<0>), reinterpret_cast<const void*>(v5::k3_cols5<0, false>), reinterpret_cast<const void*>(v5::k3_cols5<0, true>)}) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(v5::lds_bytes(1280)));
    if (e != hipSuccess) return e;
  }
  for (const void* f : {reinterpret_cast<const void*>(v5::k1_cols5<1>), reinterpret_cast<const void*>(v5::k3_cols5<1, false>), reinterpret_cast<const void*>(v5::k3_cols5<1, true>)}) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(v5::lds_bytes(2560)));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
hipError_t v5_build_fourstep(const DevPlan& pl, uint64_t* f0f, uint64_t* f0i, uint64_t* fbf, uint64_t* fbi, hipStream_t s) {
  if (v5_j1(pl)) hipLaunchKernelGGL(v5::k_build_f0<1>, dim3(pl.M2 / pl.C), dim3(v5::kThreads), 0, s, pl, f0f, f0i, fbf, fbi);
  else hipLaunchKernelGGL(v5::k_build_f0<0>, dim3(pl.M2 / pl.C), dim3(v5::kThreads), 0, s, pl, f0f, f0i, fbf, fbi);
  return hipGetLastError();
}
hipError_t v5_launch_front(const DevPlan& pl, const uint32_t* digits, const uint64_t* cbuf_in, uint32_t sub, uint64_t* W, hipStream_t s) {
  if (v5_j1(pl)) hipLaunchKernelGGL(v5::k1_cols5<1>, dim3(pl.M2 / pl.C), dim3(v5::kLaunchThreads), v5::lds_bytes(2560), s, pl, digits, cbuf_in, sub, W);
  else hipLaunchKernelGGL(v5::k1_cols5<0>, dim3(pl.M2 / pl.C), dim3(v5::kLaunchThreads), v5::lds_bytes(1280), s, pl, digits, cbuf_in, sub, W);
  return hipGetLastError();
}
hipError_t v5_launch_back(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, uint64_t scale, hipStream_t s) {
  if (v5_j1(pl)) hipLaunchKernelGGL((v5::k3_cols5<1, false>), dim3(pl.M2 / pl.C), dim3(v5::kLaunchThreads), v5::lds_bytes(2560), s, pl, W, digits, cbuf, a, scale, BackExt());
  else hipLaunchKernelGGL((v5::k3_cols5<0, false>), dim3(pl.M2 / pl.C), dim3(v5::kLaunchThreads), v5::lds_bytes(1280), s, pl, W, digits, cbuf, a, scale, BackExt());
  return hipGetLastError();
}
hipError_t v5_launch_back_ext(const DevPlan& pl, const uint64_t* W, uint32_t* digits, uint64_t* cbuf, uint32_t a, const BackExt& x, hipStream_t s) {
  if (v5_j1(pl)) hipLaunchKernelGGL((v5::k3_cols5<1, true>), dim3(pl.M2 / pl.C), dim3(v5::kLaunchThreads), v5::lds_bytes(2560), s, pl, W, digits, cbuf, a, uint64_t(1), x);
  else hipLaunchKernelGGL((v5::k3_cols5<0, true>), dim3(pl.M2 / pl.C), dim3(v5::kLaunchThreads), v5::lds_bytes(1280), s, pl, W, digits, cbuf, a, uint64_t(1), x);
  return hipGetLastError();
}
#if defined(MI355_PROBE)
hipError_t v5_probe_launch(const DevPlan& pl, int kind, int grid_mult, int extra_lds, const uint32_t* digits, uint64_t* cbuf, uint64_t* W, uint32_t* dout, hipStream_t s) {
  if (v5_j1(pl)) return hipErrorNotSupported;
  const dim3 g5((pl.M2 / pl.C) * grid_mult), b5(v5::kLaunchThreads);
  const size_t l5 = v5::lds_bytes(1280) + size_t(extra_lds);
  const void* f = kind == 0 ? reinterpret_cast<const void*>(v5::k1_cols5<0>) : reinterpret_cast<const void*>(v5::k3_cols5<0, false>);
  hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(l5));
  if (e != hipSuccess) return e;
  if (kind == 0) hipLaunchKernelGGL(v5::k1_cols5<0>, g5, b5, l5, s, pl, digits, cbuf, 0u, W);
  else hipLaunchKernelGGL((v5::k3_cols5<0, false>), g5, b5, l5, s, pl, W, dout, cbuf, 1u, uint64_t(1), BackExt());
  return hipGetLastError();
}
#endif

}  // namespace mi355
