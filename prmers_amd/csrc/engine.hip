// Engine implementation: buffers, host-side digit I/O, kernel sequencing.
#include "engine.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace mi355 {

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " in " #expr); \
  } while (0)

template <class T>
static const T* upload(unsigned char*& cursor, unsigned char* base, const std::vector<T>& v, std::vector<unsigned char>& host) {
  const size_t off = size_t(cursor - base);
  std::memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
  const T* p = reinterpret_cast<const T*>(cursor);
  cursor += (v.size() * sizeof(T) + 255) & ~size_t(255);
  return p;
}

Engine::Engine(uint32_t p, size_t reg_count, int device, bool verbose, const char* spec)
    : pl_(make_plan(p, spec, true)), device_(device), verbose_(verbose), nregs_(reg_count) {
  if (reg_count == 0) throw std::runtime_error("register_count must be > 0");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    throw std::runtime_error("no HIP device available: the MI355X engine has no CPU fallback");
  if (device < 0 || device >= ndev) throw std::runtime_error("HIP device index out of range");
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));

  reg_bytes_ = pl_.n * 8;  // digits use the first 4n bytes, a multiplicand image all 8n
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&regs_), (nregs_ + 1) * reg_bytes_));
  HIPCHK(hipMemsetAsync(regs_, 0, (nregs_ + 1) * reg_bytes_, stream_));
  slot_.resize(nregs_ + 1);
  for (size_t r = 0; r <= nregs_; ++r) slot_[r] = regs_ + r * reg_bytes_;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&cbuf_), (nregs_ + 4) * pl_.runs() * 8));
  HIPCHK(hipMemsetAsync(cbuf_, 0, (nregs_ + 4) * pl_.runs() * 8, stream_));
  cb_.resize(nregs_);
  for (size_t r = 0; r < nregs_; ++r) cb_[r] = cbuf_ + r * pl_.runs();
  for (size_t r = nregs_; r < nregs_ + 4; ++r) cb_spare_.push_back(cbuf_ + r * pl_.runs());
  kind_.assign(nregs_, kDigits);
  pending_carry_.assign(nregs_, 0);
  pending_sub_.assign(nregs_, 0);

  // one allocation for all tables
  auto padded = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
  std::vector<uint64_t> tah(pl_.TA.size()), tai2(pl_.TAi.size());
  for (size_t i = 0; i < tah.size(); ++i) { tah[i] = gf::half(pl_.TA[i]); tai2[i] = gf::dbl(pl_.TAi[i]); }
  const size_t total = padded(pl_.SA.size() * 4) + padded(pl_.SB.size() * 4) + padded(pl_.TA.size() * 8) * 4 +
                       padded(pl_.TB.size() * 8) * 2 + padded(pl_.TWlo.size() * 8) + padded(pl_.TWhi.size() * 8) +
                       padded(pl_.UT1.size() * 8) + padded(pl_.UT2.size() * 8) + padded(pl_.S2r.size() * 8) * 2 +
                       padded(pl_.S1r.size() * 8) * 2 + 1024;
  HIPCHK(hipMalloc(&tables_, total));
  std::vector<unsigned char> host(total, 0);
  unsigned char* base = static_cast<unsigned char*>(tables_);
  unsigned char* cur = base;
  dp_.SA = upload(cur, base, pl_.SA, host);
  dp_.SB = upload(cur, base, pl_.SB, host);
  dp_.TA = upload(cur, base, pl_.TA, host);
  dp_.TAi = upload(cur, base, pl_.TAi, host);
  dp_.TAh = upload(cur, base, tah, host);
  dp_.TAi2 = upload(cur, base, tai2, host);
  dp_.TB = upload(cur, base, pl_.TB, host);
  dp_.TBi = upload(cur, base, pl_.TBi, host);
  dp_.TWlo = upload(cur, base, pl_.TWlo, host);
  dp_.TWhi = upload(cur, base, pl_.TWhi, host);
  dp_.UT1 = upload(cur, base, pl_.UT1, host);
  dp_.UT2 = upload(cur, base, pl_.UT2, host);
  dp_.S2r = pl_.S2r.empty() ? nullptr : upload(cur, base, pl_.S2r, host);
  dp_.S2ri = pl_.S2ri.empty() ? nullptr : upload(cur, base, pl_.S2ri, host);
  dp_.S1r = pl_.S1r.empty() ? nullptr : upload(cur, base, pl_.S1r, host);
  dp_.S1ri = pl_.S1ri.empty() ? nullptr : upload(cur, base, pl_.S1ri, host);
  HIPCHK(hipMemcpy(tables_, host.data(), total, hipMemcpyHostToDevice));

  dp_.n = uint32_t(pl_.n); dp_.m = uint32_t(pl_.m);
  dp_.M1 = pl_.M1; dp_.M2 = pl_.M2; dp_.L1 = pl_.L1; dp_.logL1 = pl_.logL1; dp_.logM2 = pl_.logM2;
  dp_.r5 = pl_.r5; dp_.C = pl_.C; dp_.logC = 0; while ((1u << dp_.logC) < pl_.C) ++dp_.logC; dp_.q = pl_.q; dp_.t = pl_.t; dp_.twh = pl_.twh;
  dp_.I4 = pl_.I4; dp_.I4inv = pl_.I4inv;
  dp_.DI = nullptr;
  dp_.F0f = dp_.F0i = dp_.FBf = dp_.FBi = nullptr;
  if (!pl_.DI.empty()) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&di_), pl_.DI.size() * 4));
    HIPCHK(hipMemcpy(di_, pl_.DI.data(), pl_.DI.size() * 4, hipMemcpyHostToDevice));
    dp_.DI = di_;
  }
  for (int i = 0; i < 4; ++i) dp_.W5c[i] = pl_.W5c[i];
  { const char* tn = std::getenv("MI355_TUNE"); dp_.tune = tn ? uint32_t(std::atoi(tn)) : 0u; }
  {
    // the register-resident kernels keep two 512-thread work-groups per CU: a launch of more than one round
    // boosts the groups of its last half round (kernels_v2.hip, boost_if_late)
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_));
    const uint32_t slots = 2u * uint32_t(prop.multiProcessorCount);
    const char* bf = std::getenv("MI355_BOOST");           // boosted part of the last round in percent (default 50)
    const uint32_t pct = bf ? uint32_t(std::atoi(bf)) : 50u;
    auto from = [&](size_t grid) { return (!(dp_.tune & 4) && grid > slots) ? uint32_t(grid - size_t(slots) * pct / 100) : ~0u; };
    dp_.boost_rows = from(pl_.M2 == 2048 ? pl_.M1 / 2 : pl_.M1);   // (rows of 2048 go two to a tile on the register-resident row kernel)
    dp_.boost_tiles = from(pl_.tiles());
  }
  HIPCHK(configure_kernels(pl_.lds_front, pl_.lds_mid));
  if (pl_.split5) {
    HIPCHK(configure_split(dp_));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&split_), reg_bytes_));
  }
  {
    // kernel set: the register-resident radix-8 kernels where the shape is served, else the generic
    // set.  MI355_KERNELS=generic|v2rows|v2cols narrows it (A/B tests, debugging).
    const char* ks = std::getenv("MI355_KERNELS");
    const std::string sel = ks ? ks : "v2";
    v2rows_ = (sel == "v2" || sel == "v2rows") && v2_rows_supported(dp_);
    v2cols_ = (sel == "v2" || sel == "v2cols") && v2_cols_supported(dp_);
    dp_.lab_u = 1; dp_.lab_v = pl_.r5; dp_.lab_red = 0;
    if (v2cols_ && v5_cols_shape(dp_)) v5_pfa(dp_, &dp_.lab_u, &dp_.lab_v);   // radix-5 columns in prime-factor form: their own frequency labels
    else if (!v2cols_ && pl_.r5 == 5 && !pl_.split5 && pl_.L1 > 1) {   // the generic radix-5 stage in prime-factor form (kernels.hip lds_radix5)
      uint32_t u = 1; while ((pl_.L1 * u) % 5 != 1) ++u;
      dp_.lab_u = pl_.L1 * u;   // L1 (L1^-1 mod 5); lab_v stays 5
      dp_.lab_red = pl_.M1;     // lab_u blk + 5 rq <= 16 L1 + 5 (L1 - 1) < 5 M1: four conditional subtractions
    }
    {   // every exponent label + M1 k, k < M2, that a row kernel looks up lies inside the two-level root table (plan.hpp TWhi)
      const uint64_t max_label = dp_.lab_red ? uint64_t(pl_.M1) - 1 : uint64_t(dp_.lab_u) * (pl_.r5 - 1) + uint64_t(dp_.lab_v) * (pl_.L1 - 1);
      if (max_label + uint64_t(pl_.M1) * (pl_.M2 - 1) >= (uint64_t(pl_.TWhi.size()) << pl_.twh))
        throw std::runtime_error("internal: column frequency labels beyond the root table");
    }
    if (v2rows_ || v2cols_) HIPCHK(v2_configure());
    if (v2cols_) {   // four-step chain starts and ratios: built once on the device (2 x tiles x 512 + 2 x M2 words)
      const size_t nt = pl_.tiles() * v2_threads_per_tile(dp_);
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&f0_), (2 * nt + 2 * size_t(pl_.M2)) * 8));
      HIPCHK(v2_build_fourstep(dp_, f0_, f0_ + nt, f0_ + 2 * nt, f0_ + 2 * nt + pl_.M2, stream_));
      HIPCHK(hipStreamSynchronize(stream_));
      dp_.F0f = f0_; dp_.F0i = f0_ + nt; dp_.FBf = f0_ + 2 * nt; dp_.FBi = f0_ + 2 * nt + pl_.M2;
    }
#if defined(MI355_EXPERIMENTAL)
    {
      // back sweep + next front sweep in one launch for runs of squarings (kernels_v2.hip k31_cols): opt-in, MI355_CHAIN=1
      const char* ch = std::getenv("MI355_CHAIN");
      if (v2cols_ && ch && ch[0] == '1' && v2_chain_supported(dp_)) {
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&xchain_), pl_.runs() * 8 + 64));
        HIPCHK(hipMemsetAsync(xchain_, 0, pl_.runs() * 8 + 64, stream_));
      }
    }
    if (v2rows_ && v2cols_) chain_tiles_ = v3_chain_tiles(dp_, device_);
    if (chain_tiles_) {
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&chain_x_), size_t(chain_tiles_) * 256 * 8));
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&chain_flags_), (size_t(chain_tiles_) + 1) * 4));
      HIPCHK(hipMemsetAsync(chain_flags_, 0, (size_t(chain_tiles_) + 1) * 4, stream_));
    }
#endif
  }

#if defined(MI355_EXPERIMENTAL)
  {
    // transforms whose tiles are all resident at once: one cooperative launch per run of squarings instead of three launches per
    // squaring (kernels.hip k_coop).  Opt-in (MI355_COOP=1): measured slower on MI355X -- 0.044 ms per squaring at C2 inside one launch
    // against 0.031 ms for three launches; a grid barrier costs what a kernel boundary costs (2.9 vs 2.8 us, tools/microbench_gridsync.hip),
    // the hand-over data has to bypass the XCD's L2, and a cooperative launch itself takes 21.6 us.
    const char* co = std::getenv("MI355_COOP");
    if (!v2rows_ && !v2cols_ && !pl_.split5 && co && co[0] == '1') coop_groups_ = coop_groups(dp_, device_);
    if (coop_groups_) {
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&coop_flags_), (size_t(coop_groups_) + 64) * 4));
      HIPCHK(hipMemsetAsync(coop_flags_, 0, (size_t(coop_groups_) + 64) * 4, stream_));
      const char* cf = std::getenv("MI355_COOP_FAULT");   // test hook: work-group 0 skips its barriers (tests/test_gpu_coop.py)
      coop_fault_ = (cf && cf[0] == '1') ? 1u : 0u;
      const char* cb = std::getenv("MI355_COOP_BATCH");
      if (cb && std::atoi(cb) > 1) coop_batch_ = size_t(std::atoi(cb));
    }
  }
#endif

  // digit widths in natural order (ibdwt.h:127-132), s_j = p*j mod n kept incrementally
  width_.resize(pl_.n);
  uint64_t s = 0;
  for (size_t j = 0; j < pl_.n; ++j) {
    width_[j] = uint8_t(pl_.width_of_s(s));
    s += pl_.t; if (s >= pl_.n) s -= pl_.n;
  }
  { const char* hc = std::getenv("MI355_HOST_CARRY"); host_carry_ = hc && hc[0] == '1'; }
  HIPCHK(hipStreamSynchronize(stream_));
  if (verbose_) std::fprintf(stderr, "[mi355] p=%u %s regs=%zu device=%d\n", p, pl_.describe().c_str(), nregs_, device_);
}

Engine::~Engine() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (regs_) (void)hipFree(regs_);
  if (cbuf_) (void)hipFree(cbuf_);
  if (tables_) (void)hipFree(tables_);
  if (di_) (void)hipFree(di_);
  if (f0_) (void)hipFree(f0_);
  if (split_) (void)hipFree(split_);
  if (canon_) (void)hipFree(canon_);
#if defined(MI355_EXPERIMENTAL)
  if (xchain_) (void)hipFree(xchain_);
  if (chain_x_) (void)hipFree(chain_x_);
  if (chain_flags_) (void)hipFree(chain_flags_);
#endif
#if defined(MI355_EXPERIMENTAL)
  if (coop_flags_) (void)hipFree(coop_flags_);
#endif
  if (stream_) (void)hipStreamDestroy(stream_);
}

void Engine::check_reg(size_t r) const {
  if (r >= nregs_) throw std::runtime_error("register index out of range");
}
void Engine::need_digits(size_t r, const char* op) const {
  check_reg(r);
  if (kind_[r] == kImage) throw std::runtime_error(std::string(op) + ": register holds a multiplicand image, not a residue");
}

void Engine::sync() {
  HIPCHK(hipSetDevice(device_));
  HIPCHK(hipStreamSynchronize(stream_));
  coop_check();
  chain_check();
}

#if defined(MI355_EXPERIMENTAL)
// The fused back + front launches of square_mul_n hand carry words from tile to tile inside the launch; a wait that timed out raised the
// error word and left garbage behind: nothing is read out of the engine after that.
void Engine::chain_check() {
  if (xchain_failed_) throw std::runtime_error("chained squaring kernel: a carry hand-over timed out earlier; the engine's registers are not valid");
  if (xchain_used_) {
    uint32_t err = 0;
    HIPCHK(hipMemcpyAsync(&err, xchain_err(), 4, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    xchain_used_ = false;
    if (err) { xchain_failed_ = true; throw std::runtime_error("chained squaring kernel: carry hand-over timed out"); }
  }
  if (chain_failed_) throw std::runtime_error("chained squaring kernel: a carry hand-over timed out earlier; the engine's registers are not valid");
  if (!chain_used_) return;
  uint32_t err = 0;
  HIPCHK(hipMemcpyAsync(&err, chain_flags_ + chain_tiles_, 4, hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  chain_used_ = false;
  if (err) { chain_failed_ = true; throw std::runtime_error("chained squaring kernel: carry hand-over timed out (tiles not co-resident?)"); }
}
#endif

#if defined(MI355_EXPERIMENTAL)
// The grid barrier of k_coop gives up after 0.2 s and raises the error word; the results of that launch are garbage.
void Engine::coop_check() {
  if (coop_failed_) throw std::runtime_error("cooperative squaring kernel: a grid barrier timed out earlier; the engine's registers are not valid");
  if (!coop_used_) return;
  uint32_t err = 0;
  HIPCHK(hipMemcpyAsync(&err, coop_flags_ + coop_groups_, 4, hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  coop_used_ = false;
  if (err) { coop_failed_ = true; throw std::runtime_error("cooperative squaring kernel: grid barrier timed out (work-groups not co-resident?)"); }
}

void Engine::coop_launch(size_t r, uint32_t a, size_t count, uint32_t sub_next) {
  if (coop_failed_) coop_check();
  while (count) {
    const uint32_t c = uint32_t(std::min<size_t>(count, 1u << 20));
    HIPCHK(launch_coop(dp_, coop_groups_, digits(r), cbuf(r), pending_carry_[r] != 0, work(), a, pending_sub_[r], sub_next, c, coop_flags_,
                       coop_flags_ + coop_groups_, coop_epoch_, coop_fault_, stream_));
    coop_epoch_ += 3 * c - 1;
    pending_carry_[r] = 1;
    pending_sub_[r] = sub_next;
    coop_used_ = true;
    count -= c;
  }
}
#endif

void Engine::normalize(size_t r) {
  if (kind_[r] != kDigits) return;
  if (pending_carry_[r]) {
    HIPCHK(launch_carry_fix(dp_, digits(r), cbuf(r), stream_));
    pending_carry_[r] = 0;
  }
  if (pending_sub_[r]) {
    HIPCHK(launch_sub_small(dp_, digits(r), pending_sub_[r], stream_));
    pending_sub_[r] = 0;
  }
}

// Runs of two digits (C = 1): the run carries go into the digits at once.  A carry word has about w + log2(n) bits (the
// convolution sums of unsigned digits are ~ n 2^(2w-2)); its first digit absorbs w of them and the rest lands on the
// run's second digit, log2(n) - 2 (+ log2 a) bits above its width -- too much for the next squaring once that exceeds w.
// Local carry passes (canon.hip k_relax) take w bits off per pass; as many as it takes to get below the width follow.
void Engine::carry_fix_now(size_t r) {
  HIPCHK(launch_carry_fix(dp_, digits(r), cbuf(r), stream_));
  pending_carry_[r] = 0;
  if (pl_.C >= 2) return;
  int excess = ilog2(pl_.n) + 1 + 4 - 2;   // log2(n) rounded up, factor a up to 15
  const int w = int(pl_.q);                 // the narrower digit width
  while (excess > w - 2) {
    HIPCHK(canon_relax(dp_, pl_.p, digits(r), reinterpret_cast<uint32_t*>(work()), stream_));
    swap_with_work(r);
    excess -= w;
  }
}

void Engine::run_front(size_t r) {
  if (pl_.split5) {   // columns beyond LDS (n = 5 2^26): radix-5 stage through the second work buffer
    normalize(r);
    HIPCHK(launch_front_split(dp_, digits(r), split_, work(), stream_));
    return;
  }
  if (v2cols_) {
    HIPCHK(v2_launch_front(dp_, digits(r), pending_carry_[r] ? cbuf(r) : nullptr, pending_sub_[r], work(), stream_));
  } else if (pl_.C >= 2 && kind_[r] == kDigits && !pending_sub_[r]) {
    HIPCHK(launch_front(dp_, digits(r), pending_carry_[r] ? cbuf(r) : nullptr, work(), stream_));   // does not modify digits(r)
  } else {
    normalize(r);
    HIPCHK(launch_front(dp_, digits(r), nullptr, work(), stream_));
  }
}

void Engine::run_middle(const uint64_t* in, const uint64_t* y, uint64_t* out, int mode, uint32_t sub) {
  if (v2rows_) HIPCHK(v2_launch_middle(dp_, in, y, out, mode, sub, stream_));
  else HIPCHK(launch_middle(dp_, in, y, out, mode, sub, stream_));
}

// work() -> digits(r) (+ run carries in cbuf(r)); the carry fix is deferred to the next front sweep
// when that kernel can fold it in, otherwise applied right away
void Engine::run_back(size_t r, uint32_t a) {
  if (pl_.split5) {
    HIPCHK(launch_back_split(dp_, work(), split_, digits(r), cbuf(r), a, stream_));
    carry_fix_now(r);
    pending_sub_[r] = 0;
    return;
  }
  if (v2cols_) {
    HIPCHK(v2_launch_back(dp_, work(), digits(r), cbuf(r), a, 1, stream_));
    pending_carry_[r] = 1;
  } else {
    HIPCHK(launch_back(dp_, work(), digits(r), cbuf(r), a, stream_));
    if (pl_.C >= 2) pending_carry_[r] = 1;   // the generic front folds the run carries in as well
    else carry_fix_now(r);
  }
  pending_sub_[r] = 0;
}

// ---- host digit I/O -------------------------------------------------------------------------

void Engine::write_values(size_t dst, const std::vector<uint32_t>& natural) {
  // natural order goes up as it is; the tile-major order is made on the device (canon.hip k_scatter)
  HIPCHK(hipSetDevice(device_));
  HIPCHK(hipStreamSynchronize(stream_));
  uint32_t* nat = reinterpret_cast<uint32_t*>(work());
  HIPCHK(hipMemcpy(nat, natural.data(), pl_.n * 4, hipMemcpyHostToDevice));
  HIPCHK(canon_scatter(dp_, pl_.p, nat, digits(dst), stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  kind_[dst] = kDigits;
  pending_carry_[dst] = 0; pending_sub_[dst] = 0;
}

// canonical digits of register r (strong carry with wrap-around, 2^p - 1 -> 0) in natural order, on the device
uint32_t* Engine::canon_digits(size_t r, int slot) {
  need_digits(r, "get");
  HIPCHK(hipSetDevice(device_));
  coop_check();   // nothing is read out of an engine whose one-launch kernel gave up at a grid barrier
  chain_check();
  const size_t sw = canon_scratch_words(dp_);
  if (!canon_) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&canon_), (sw + 2 * pl_.n) * 4));
    HIPCHK(hipMemsetAsync(canon_flags(dp_, canon_), 0, 16 * 4, stream_));
  }
  normalize(r);
  uint32_t* out = canon_ + sw + size_t(slot) * pl_.n;
  HIPCHK(canon_launch(dp_, pl_.p, digits(r), out, canon_, stream_));
  return out;
}

// flags: [0] all ones, [1] chain too wide (fall back), [2] compare differs; clears the sticky ones for the next use
bool Engine::canon_flags_ok(uint32_t (&flags)[4]) {
  uint32_t* df = canon_flags(dp_, canon_);
  HIPCHK(hipMemcpyAsync(flags, df, 16, hipMemcpyDeviceToHost, stream_));
  HIPCHK(hipMemsetAsync(df, 0, 16 * 4, stream_));
  HIPCHK(hipStreamSynchronize(stream_));
  return flags[1] == 0;
}

void Engine::read_values(size_t src, std::vector<uint64_t>& v) {
  if (host_carry_) { read_values_host(src, v); return; }
  uint32_t* d = canon_digits(src, 0);
  uint32_t flags[4];
  stage_.resize(pl_.n);
  HIPCHK(hipMemcpyAsync(stage_.data(), d, pl_.n * 4, hipMemcpyDeviceToHost, stream_));
  if (!canon_flags_ok(flags)) { read_values_host(src, v); return; }
  v.resize(pl_.n);
  if (flags[0]) {   // 2^p - 1: the reference's get() leaves the digits all ones (engine.h:188-196 maps them to 0 later)
    for (size_t k = 0; k < pl_.n; ++k) v[k] = (uint64_t(1) << width_[k]) - 1;
  } else {
    for (size_t k = 0; k < pl_.n; ++k) v[k] = stage_[k];
  }
}

void Engine::read_values_host(size_t src, std::vector<uint64_t>& v) {
  need_digits(src, "get");
  stage_.resize(pl_.n);
  HIPCHK(hipSetDevice(device_));
  coop_check();
  chain_check();
  normalize(src);
  HIPCHK(hipStreamSynchronize(stream_));
  HIPCHK(hipMemcpy(stage_.data(), digits(src), pl_.n * 4, hipMemcpyDeviceToHost));
  v.resize(pl_.n);
  const size_t M2 = pl_.M2, C = pl_.C, M1 = pl_.M1;
  for (size_t i = 0; i < pl_.m; ++i) {
    const size_t i1 = i / M2, i2 = i % M2, T = i2 / C, c = i2 % C;
    const size_t s = ((T * M1 + i1) * C + c) * 2;
    v[2 * i] = stage_[s];
    v[2 * i + 1] = stage_[s + 1];
  }
  // strong carry with wrap-around (engine_gpu.h:1543-1557)
  uint64_t c = 0;
  for (size_t k = 0; k < pl_.n; ++k) {
    const uint64_t t = v[k] + c;
    v[k] = t & ((uint64_t(1) << width_[k]) - 1);
    c = t >> width_[k];
  }
  while (c != 0) {
    for (size_t k = 0; k < pl_.n; ++k) {
      const uint64_t t = v[k] + c;
      v[k] = t & ((uint64_t(1) << width_[k]) - 1);
      c = t >> width_[k];
      if (c == 0) break;
    }
  }
}

void Engine::set_u32(size_t dst, uint32_t value) {
  check_reg(dst);
  HIPCHK(hipSetDevice(device_));
  HIPCHK(hipMemsetAsync(digits(dst), 0, pl_.n * 4, stream_));
  // spread the constant over the first digits (the reference stores it whole in digit 0,
  // engine_gpu.h:1444-1449; same value, but never an over-wide digit)
  if (value) HIPCHK(canon_set_small(dp_, pl_.p, digits(dst), value, stream_));
  kind_[dst] = kDigits;
  pending_carry_[dst] = 0; pending_sub_[dst] = 0;
}

void Engine::set_digits(size_t dst, const uint64_t* d, size_t count) {
  check_reg(dst);
  if (count != pl_.n) throw std::runtime_error("set_digits: count must equal the transform size");
  std::vector<uint32_t> nat(pl_.n);
  for (size_t k = 0; k < pl_.n; ++k) nat[k] = uint32_t(d[k]);
  write_values(dst, nat);
}

void Engine::get_digits(size_t src, uint64_t* d, size_t count) {
  if (count != pl_.n) throw std::runtime_error("get_digits: count must equal the transform size");
  std::vector<uint64_t> v;
  read_values(src, v);
  for (size_t k = 0; k < pl_.n; ++k) d[k] = uint32_t(v[k]) | (uint64_t(width_[k]) << 32);  // engine_gpu.h:1560
}

uint64_t Engine::res64(size_t src) {
  std::vector<uint64_t> v;
  size_t have = pl_.n;
  if (host_carry_) {
    read_values_host(src, v);
  } else {
    // the low 64 bits live in the first few digits: canonicalise on the device, read back only those
    uint32_t* d = canon_digits(src, 0);
    have = std::min<size_t>(pl_.n, 16);
    uint32_t head[16], flags[4];
    HIPCHK(hipMemcpyAsync(head, d, have * 4, hipMemcpyDeviceToHost, stream_));
    if (!canon_flags_ok(flags)) { read_values_host(src, v); have = pl_.n; }
    else { v.resize(have); for (size_t k = 0; k < have; ++k) v[k] = flags[0] ? (uint64_t(1) << width_[k]) - 1 : head[k]; }
  }
  uint64_t r64 = 0; unsigned s = 0;   // engine.h:257-269
  for (size_t k = 0; k < have; ++k) {
    r64 += v[k] << s;
    s += width_[k];
    if (s >= 64) break;
  }
  return r64;
}

void Engine::get_words(size_t src, uint32_t* w, size_t count) {
  if (count != word_count()) throw std::runtime_error("get_words: count must equal word_count()");
  std::vector<uint64_t> v;
  read_values(src, v);
  bool all_ones = true;   // 2^p - 1 == 0 (engine.h:188-196)
  for (size_t k = 0; k < pl_.n && all_ones; ++k) all_ones = (v[k] == (uint64_t(1) << width_[k]) - 1);
  std::memset(w, 0, count * 4);
  if (all_ones) return;
  size_t bit = 0;
  for (size_t k = 0; k < pl_.n; ++k) {
    const size_t i = bit / 32, s = bit % 32;
    const uint64_t x = v[k] << s;
    w[i] |= uint32_t(x);
    if ((x >> 32) && i + 1 < count) w[i + 1] |= uint32_t(x >> 32);
    bit += width_[k];
  }
}

void Engine::set_words(size_t dst, const uint32_t* w, size_t count) {
  check_reg(dst);
  if (count != word_count()) throw std::runtime_error("set_words: count must equal word_count()");
  // bits at and above p are folded back (2^p = 1), so any count-word value is accepted
  std::vector<uint32_t> src(w, w + count);
  src.push_back(0);
  const unsigned top = pl_.p % 32;
  uint64_t fold = 0;
  if (top) { fold = src[count - 1] >> top; src[count - 1] &= (1u << top) - 1; }
  for (size_t i = 0; fold && i < count; ++i) {  // add the folded bits at bit 0
    const uint64_t t = uint64_t(src[i]) + (fold & 0xffffffffu);
    src[i] = uint32_t(t);
    fold = (fold >> 32) + (t >> 32);
  }
  if (top && (src[count - 1] >> top)) {  // the addition rippled past bit p once more
    src[count - 1] &= (1u << top) - 1;
    for (size_t i = 0; i < count; ++i) { if (++src[i] != 0) break; }
  }
  std::vector<uint32_t> nat(pl_.n);
  size_t bit = 0;
  for (size_t k = 0; k < pl_.n; ++k) {   // engine.h:206-232
    const size_t i = bit / 32, s = bit % 32;
    uint64_t u = src[i] >> s;
    if (s != 0) u |= uint64_t(src[i + 1]) << (32 - s);
    nat[k] = uint32_t(u & ((uint64_t(1) << width_[k]) - 1));
    bit += width_[k];
  }
  write_values(dst, nat);
}

bool Engine::equal(size_t lhs, size_t rhs) {
  if (!host_carry_) {
    // both registers canonicalised and compared on the device: 16 bytes cross PCIe (the reference reads both
    // registers back and carries them on the host: engine.h:148-157 via engine_gpu.h:1534-1561)
    uint32_t* a = canon_digits(lhs, 0);
    uint32_t* b = canon_digits(rhs, 1);
    HIPCHK(canon_compare(a, b, uint32_t(pl_.n), canon_flags(dp_, canon_) + 2, stream_));
    uint32_t flags[4];
    if (canon_flags_ok(flags)) return flags[2] == 0;
  }
  std::vector<uint64_t> a, b;
  read_values_host(lhs, a);
  read_values_host(rhs, b);
  auto all_ones = [&](const std::vector<uint64_t>& v) {
    for (size_t k = 0; k < pl_.n; ++k) if (v[k] != (uint64_t(1) << width_[k]) - 1) return false;
    return true;
  };
  if (all_ones(a)) std::fill(a.begin(), a.end(), 0);   // 2^p - 1 == 0
  if (all_ones(b)) std::fill(b.begin(), b.end(), 0);
  return a == b;
}

// ---- register operations -------------------------------------------------------------------

void Engine::copy(size_t dst, size_t src) {
  check_reg(dst); check_reg(src);
  if (dst == src) return;
  HIPCHK(hipSetDevice(device_));
  // the register is copied as it stands: digits with their pending run carries and small subtraction (no carry sweep),
  // a multiplicand image whole
  const size_t bytes = (kind_[src] == kDigits) ? pl_.n * 4 : reg_bytes_;
  HIPCHK(hipMemcpyAsync(slot_[dst], slot_[src], bytes, hipMemcpyDeviceToDevice, stream_));
  if (kind_[src] == kDigits && pending_carry_[src])
    HIPCHK(hipMemcpyAsync(cbuf(dst), cbuf(src), pl_.runs() * 8, hipMemcpyDeviceToDevice, stream_));
  pending_carry_[dst] = (kind_[src] == kDigits) ? pending_carry_[src] : 0;
  pending_sub_[dst] = (kind_[src] != kImage) ? pending_sub_[src] : 0;
  kind_[dst] = kind_[src];
}

void Engine::square_chain(size_t r, uint32_t a, hipEvent_t* ev) {
  if (ev) HIPCHK(hipEventRecord(ev[0], stream_));
#if defined(MI355_EXPERIMENTAL)
  if (coop_groups_ && pending_sub_[r] < (1u << 30)) {   // one launch: the whole squaring sits in slot 0
    coop_launch(r, a, 1, 0);
    if (ev) for (int k = 1; k <= 4; ++k) HIPCHK(hipEventRecord(ev[k], stream_));
    return;
  }
#endif
  run_front(r);
  if (ev) HIPCHK(hipEventRecord(ev[1], stream_));
  run_middle(work(), nullptr, work(), 0, 0);
  if (ev) HIPCHK(hipEventRecord(ev[2], stream_));
  if (v2cols_) {
    HIPCHK(v2_launch_back(dp_, work(), digits(r), cbuf(r), a, 1, stream_));
    if (ev) { HIPCHK(hipEventRecord(ev[3], stream_)); HIPCHK(hipEventRecord(ev[4], stream_)); }
    pending_carry_[r] = 1;
  } else if (pl_.split5) {
    HIPCHK(launch_back_split(dp_, work(), split_, digits(r), cbuf(r), a, stream_));
    if (ev) HIPCHK(hipEventRecord(ev[3], stream_));
    carry_fix_now(r);
    if (ev) HIPCHK(hipEventRecord(ev[4], stream_));
  } else {
    HIPCHK(launch_back(dp_, work(), digits(r), cbuf(r), a, stream_));
    if (ev) HIPCHK(hipEventRecord(ev[3], stream_));
    if (pl_.C >= 2) pending_carry_[r] = 1;
    else carry_fix_now(r);
    if (ev) HIPCHK(hipEventRecord(ev[4], stream_));
  }
  pending_sub_[r] = 0;
}

void Engine::square_mul(size_t r, uint32_t a) {
  need_digits(r, "square_mul");
  if (a == 0) throw std::runtime_error("square_mul: factor must be >= 1");
  HIPCHK(hipSetDevice(device_));
  square_chain(r, a, nullptr);
}

void Engine::square_mul_n(size_t r, uint32_t a, size_t count, uint32_t sub) {
  need_digits(r, "square_mul_n");
  if (a == 0) throw std::runtime_error("square_mul_n: factor must be >= 1");
  if (count == 0) return;
  HIPCHK(hipSetDevice(device_));
#if defined(MI355_EXPERIMENTAL)
  if (coop_groups_ && pending_sub_[r] < (1u << 30) && sub < (1u << 30)) { coop_launch(r, a, count, sub); return; }
#endif
#if defined(MI355_EXPERIMENTAL)
  if (xchain_ && a == 1 && count >= 2 && sub < (1u << 30) && !xchain_failed_) {
    // front | rows | [back + front | rows] x (count - 1) | back: the back sweep of a squaring and the front sweep of the next one are ONE
    // launch (kernels_v2.hip k31_cols); same digits as the loop below
    run_front(r);                                      // consumes pending carries / subtraction
    for (size_t i = 0; i + 1 < count; ++i) {
      run_middle(work(), nullptr, work(), 0, 0);
      HIPCHK(v2_launch_backfront(dp_, work(), sub, xchain_, xchain_err(), xchain_tag_, stream_));
      xchain_tag_ = xchain_tag_ % 4095u + 1u;          // 1 .. 4095: never the tag of the launch before, never the zero of a fresh buffer
      xchain_used_ = true;
    }
    run_middle(work(), nullptr, work(), 0, 0);
    run_back(r, a);
    if (sub) sub_u32(r, sub);
    return;
  }
  if (chain_tiles_ && count >= 2 && sub < (1u << 30) && !chain_failed_) {
    // front | rows | [back + front | rows] x (count - 1) | back: the back sweep of a squaring and the front sweep of the next one are ONE
    // launch on the small shapes (kernels_v3.hip k31_cols256_planes); same digits as the loop below
    run_front(r);                                      // consumes pending carries / subtraction
    for (size_t i = 0; i + 1 < count; ++i) {
      run_middle(work(), nullptr, work(), 0, 0);
      HIPCHK(v3_launch_backfront(dp_, work(), a, sub, chain_x_, chain_flags_, ++chain_epoch_, stream_));
      chain_used_ = true;
      if (chain_epoch_ >= 0x7fff0000u) {               // the flags compare epochs in 31 bits: start over
        HIPCHK(hipMemsetAsync(chain_flags_, 0, size_t(chain_tiles_) * 4, stream_));
        chain_epoch_ = 0;
      }
    }
    run_middle(work(), nullptr, work(), 0, 0);
    run_back(r, a);
    if (sub) sub_u32(r, sub);
    return;
  }
#endif
  for (size_t i = 0; i < count; ++i) { square_chain(r, a, nullptr); if (sub) sub_u32(r, sub); }
}

void Engine::prepare(size_t dst, size_t src) {
  need_digits(src, "set_multiplicand");
  check_reg(dst);
  HIPCHK(hipSetDevice(device_));
  run_front(src);
  run_middle(work(), nullptr, image(dst), 2, 0);
  kind_[dst] = kImage;
  pending_carry_[dst] = 0; pending_sub_[dst] = 0;
}

void Engine::mul(size_t dst, size_t src, uint32_t a) {
  need_digits(dst, "mul");
  check_reg(src);
  if (kind_[src] != kImage) throw std::runtime_error("mul: src must be a multiplicand (set_multiplicand)");
  if (dst == src) throw std::runtime_error("mul: dst and src must differ");
  if (a == 0) throw std::runtime_error("mul: factor must be >= 1");
  HIPCHK(hipSetDevice(device_));
  run_front(dst);
  run_middle(work(), image(src), work(), 1, 0);
  run_back(dst, a);
}

uint64_t* Engine::take_spare_cbuf() {
  if (cb_spare_.empty()) throw std::runtime_error("internal: no spare carry buffer");
  uint64_t* b = cb_spare_.back();
  cb_spare_.pop_back();
  return b;
}
void Engine::adopt_cbuf(size_t r, uint64_t* fresh) {
  cb_spare_.push_back(cb_[r]);
  cb_[r] = fresh;
}

void Engine::digits_ready(size_t r) {
  need_digits(r, "add/sub");
  if (pending_sub_[r]) normalize(r);   // rare: a small subtraction not yet folded into a sweep
}

// sum -> s1 (and s2), difference -> d1 (and d2); -1: not wanted.  One run-wise sweep on pending-carry digits
// (kernels.hip k_linear); the results leave their run carries pending for the next front sweep.
void Engine::linear(long s1, long s2, long d1, long d2, size_t a, size_t b) {
  HIPCHK(hipSetDevice(device_));
  digits_ready(a); digits_ready(b);
  const long outs[4] = {s1, s2, d1, d2};
  for (int i = 0; i < 4; ++i)
    if (outs[i] >= 0) { check_reg(size_t(outs[i])); for (int j = 0; j < i; ++j) if (outs[j] == outs[i]) throw std::runtime_error("addsub: output registers must differ"); }
  LinArgs la;
  la.a = digits(a); la.ca = pending_carry_[a] ? cbuf(a) : nullptr;
  la.b = digits(b); la.cb = pending_carry_[b] ? cbuf(b) : nullptr;
  uint64_t* fresh[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < 4; ++i) if (outs[i] >= 0) fresh[i] = take_spare_cbuf();
  if (s1 >= 0) { la.s1 = digits(size_t(s1)); la.cs1 = fresh[0]; }
  if (s2 >= 0) { la.s2 = digits(size_t(s2)); la.cs2 = fresh[1]; }
  if (d1 >= 0) { la.d1 = digits(size_t(d1)); la.cd1 = fresh[2]; }
  if (d2 >= 0) { la.d2 = digits(size_t(d2)); la.cd2 = fresh[3]; }
  if ((s2 >= 0 && s1 < 0) || (d2 >= 0 && d1 < 0)) throw std::runtime_error("internal: copy output without a primary output");
  HIPCHK(launch_linear(dp_, la, stream_));
  for (int i = 0; i < 4; ++i)
    if (outs[i] >= 0) {
      const size_t r = size_t(outs[i]);
      adopt_cbuf(r, fresh[i]);
      kind_[r] = kDigits; pending_carry_[r] = 1; pending_sub_[r] = 0;
      if (pl_.C < 2) carry_fix_now(r);   // runs of two digits: no deferred fold
    }
}

void Engine::add(size_t dst, size_t src) {
  need_digits(dst, "add"); need_digits(src, "add");
  linear(long(dst), -1, -1, -1, dst, src);
}

void Engine::sub_reg(size_t dst, size_t src) {
  need_digits(dst, "sub_reg"); need_digits(src, "sub_reg");
  linear(-1, -1, long(dst), -1, dst, src);
}

void Engine::addsub(size_t sum_out, size_t diff_out, size_t a, size_t b) {
  need_digits(a, "addsub"); need_digits(b, "addsub");
  linear(long(sum_out), -1, long(diff_out), -1, a, b);
}

void Engine::addsub_copy(size_t sum, size_t diff, size_t sum_copy, size_t diff_copy, size_t a, size_t b) {
  need_digits(a, "addsub_copy"); need_digits(b, "addsub_copy");
  linear(long(sum), long(sum_copy), long(diff), long(diff_copy), a, b);
}

// back sweep of work() into dst with the extras of kernels.hpp BackExt
void Engine::back_ext(size_t dst, uint32_t a, long copy_to, long add_src) {
  BackExt x;
  if (copy_to >= 0 && size_t(copy_to) != dst) { check_reg(size_t(copy_to)); x.digits2 = digits(size_t(copy_to)); x.cbuf2 = cbuf(size_t(copy_to)); }
  if (add_src >= 0) { x.add_digits = digits(size_t(add_src)); x.add_cbuf = pending_carry_[size_t(add_src)] ? cbuf(size_t(add_src)) : nullptr; }
  uint64_t* fresh = take_spare_cbuf();   // the addend may be dst itself: its pending carries are read while the new ones are written
  if (v2cols_) HIPCHK(v2_launch_back_ext(dp_, work(), digits(dst), fresh, a, x, stream_));
  else HIPCHK(launch_back_ext(dp_, work(), digits(dst), fresh, a, x, stream_));
  adopt_cbuf(dst, fresh);
  const size_t outs[2] = {dst, x.digits2 ? size_t(copy_to) : dst};
  for (int i = 0; i < (x.digits2 ? 2 : 1); ++i) {
    const size_t r = outs[i];
    kind_[r] = kDigits; pending_sub_[r] = 0; pending_carry_[r] = 1;
    if (!v2cols_ && pl_.C < 2) carry_fix_now(r);
  }
}

void Engine::square_mul_copy(size_t src, size_t dst_copy, uint32_t a) {
  need_digits(src, "square_mul_copy"); check_reg(dst_copy);
  if (a == 0) throw std::runtime_error("square_mul_copy: factor must be >= 1");
  HIPCHK(hipSetDevice(device_));
  if (dst_copy == src || pl_.split5) { square_mul(src, a); copy(dst_copy, src); return; }
  run_front(src);
  run_middle(work(), nullptr, work(), 0, 0);
  back_ext(src, a, long(dst_copy), -1);
}

void Engine::mul_copy(size_t dst, size_t src, size_t dst_copy, uint32_t a) {
  need_digits(dst, "mul_copy"); check_reg(src); check_reg(dst_copy);
  if (kind_[src] != kImage) throw std::runtime_error("mul_copy: src must be a multiplicand (set_multiplicand)");
  if (dst == src || dst_copy == src) throw std::runtime_error("mul_copy: the multiplicand must differ from the outputs");
  if (a == 0) throw std::runtime_error("mul_copy: factor must be >= 1");
  HIPCHK(hipSetDevice(device_));
  if (dst_copy == dst || pl_.split5) { mul(dst, src, a); copy(dst_copy, dst); return; }
  run_front(dst);
  run_middle(work(), image(src), work(), 1, 0);
  back_ext(dst, a, long(dst_copy), -1);
}

void Engine::mul_add(size_t dst, size_t mul_src, size_t add_src, uint32_t a) {
  need_digits(dst, "mul_add"); check_reg(mul_src); need_digits(add_src, "mul_add");
  if (kind_[mul_src] != kImage) throw std::runtime_error("mul_add: mul_src must be a multiplicand (set_multiplicand)");
  if (dst == mul_src) throw std::runtime_error("mul_add: dst and mul_src must differ");
  if (a == 0) throw std::runtime_error("mul_add: factor must be >= 1");
  HIPCHK(hipSetDevice(device_));
  if (pl_.split5) {   // the split sweeps have no fused variants: the base-class composition (engine.h:65-70)
    if (add_src == dst) throw std::runtime_error("mul_add: add_src == dst needs the fused sweep, which this transform size does not have");
    mul(dst, mul_src, a); add(dst, add_src); return;
  }
  if (add_src != dst) digits_ready(add_src);
  else if (kind_[dst] != kDigits || pending_sub_[dst]) normalize(dst);
  run_front(dst);   // reads digits(dst) (+ pending carries) and leaves them in place
  run_middle(work(), image(mul_src), work(), 1, 0);
  back_ext(dst, a, -1, long(add_src));
}

void Engine::sub_u32(size_t r, uint32_t v) {
  need_digits(r, "sub");
  if (v == 0) return;
  HIPCHK(hipSetDevice(device_));
  if ((v2cols_ || coop_on()) && uint64_t(pending_sub_[r]) + v < (1u << 30)) { pending_sub_[r] += v; return; }  // folded into the next front / middle sweep
  // the small subtraction only touches the digit vector (cyclic borrow), so run carries that are still pending
  // for the next front sweep can stay pending: value = digits + carries - v either way
  if (!(kind_[r] == kDigits && pl_.C >= 2 && !pending_sub_[r])) normalize(r);
  HIPCHK(launch_sub_small(dp_, digits(r), v, stream_));
}

// ---- raw images -----------------------------------------------------------------------------

void Engine::get_data(size_t src, void* data, size_t size) {
  check_reg(src);
  if (size != register_data_size()) throw std::runtime_error("get_data: size mismatch");
  HIPCHK(hipSetDevice(device_));
  coop_check();
  chain_check();
  normalize(src);
  HIPCHK(hipStreamSynchronize(stream_));
  HIPCHK(hipMemcpy(data, slot_[src], reg_bytes_, hipMemcpyDeviceToHost));
  const uint64_t tag = kind_[src];
  std::memcpy(static_cast<unsigned char*>(data) + reg_bytes_, &tag, 8);
}

void Engine::set_data(size_t dst, const void* data, size_t size) {
  check_reg(dst);
  if (size != register_data_size()) throw std::runtime_error("set_data: size mismatch");
  uint64_t tag = 0;
  std::memcpy(&tag, static_cast<const unsigned char*>(data) + reg_bytes_, 8);
  if (tag > 1) throw std::runtime_error("set_data: not an image written by this engine");
  HIPCHK(hipSetDevice(device_));
  HIPCHK(hipStreamSynchronize(stream_));
  HIPCHK(hipMemcpy(slot_[dst], data, reg_bytes_, hipMemcpyHostToDevice));
  kind_[dst] = uint8_t(tag);
  pending_carry_[dst] = 0; pending_sub_[dst] = 0;
}

void Engine::get_checkpoint(void* data, size_t size) {
  if (size != checkpoint_size()) throw std::runtime_error("get_checkpoint: size mismatch");
  for (size_t r = 0; r < nregs_; ++r) get_data(r, static_cast<unsigned char*>(data) + r * register_data_size(), register_data_size());
}

void Engine::set_checkpoint(const void* data, size_t size) {
  if (size != checkpoint_size()) throw std::runtime_error("set_checkpoint: size mismatch");
  for (size_t r = 0; r < nregs_; ++r) set_data(r, static_cast<const unsigned char*>(data) + r * register_data_size(), register_data_size());
}

// ---- measurement ----------------------------------------------------------------------------

namespace {
struct EventPool {   // HIP events owned for the length of one measurement
  std::vector<hipEvent_t> evs;
  hipEvent_t make() {
    hipEvent_t e = nullptr;
    HIPCHK(hipEventCreate(&e));
    evs.push_back(e);
    return e;
  }
  ~EventPool() { for (hipEvent_t e : evs) (void)hipEventDestroy(e); }
};
}  // namespace

const char* Engine::kernel_name(size_t k) {
  static const char* names[kKernels] = {"k_front", "k_middle", "k_back", "k_carry_fix", "k_sub_small", "event_overhead"};
  return k < kKernels ? names[k] : "";
}

// kernel_ms[k]: average duration of kernel k of one squaring (-1: that kernel is not launched on this path),
// from one event between consecutive kernels on the engine's stream, minus the cost of an event record itself
// (kernel_ms[5], measured as the spacing of back-to-back records on the same stream: without the subtraction every
// interval carries one record, ~4-5 us, and a path that launches no kernel between two records shows it as a kernel).
void Engine::time_square_mul(size_t r, uint32_t a, uint32_t sub, size_t iters, double* total_ms, double* kernel_ms, size_t kcount) {
  need_digits(r, "time_square_mul");
  if (a == 0 || iters == 0) throw std::runtime_error("time_square_mul: factor and iters must be >= 1");
  HIPCHK(hipSetDevice(device_));
  EventPool pool;   // destroys its events on every way out (a HIPCHK that throws mid-batch used to leak them)
  const hipEvent_t e0 = pool.make(), e1 = pool.make();
  HIPCHK(hipStreamSynchronize(stream_));
  HIPCHK(hipEventRecord(e0, stream_));
#if defined(MI355_EXPERIMENTAL)
  if (coop_groups_ && coop_batch_ > 1) {   // MI355_COOP_BATCH squarings per launch (what a PRP / LL loop between two checks does)
    for (size_t done = 0; done < iters;) { const size_t c = std::min(coop_batch_, iters - done); square_mul_n(r, a, c, sub); done += c; }
  } else
#endif
  {
    bool as_run = false;   // the run of squarings as the callers issue it (experimental build: back + front in one launch where that is on)
#if defined(MI355_EXPERIMENTAL)
    as_run = xchain_ != nullptr || chain_tiles_ != 0;
#endif
    if (as_run) square_mul_n(r, a, iters, sub);
    else for (size_t i = 0; i < iters; ++i) {
      square_chain(r, a, nullptr);
      if (sub) sub_u32(r, sub);
    }
  }
  HIPCHK(hipEventRecord(e1, stream_));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  if (total_ms) *total_ms = ms;

  if (kernel_ms && kcount) {
    for (size_t k = 0; k < kcount; ++k) kernel_ms[k] = 0;
    const size_t reps = std::min<size_t>(iters, 64);
    const size_t per = 6;
    std::vector<hipEvent_t> ev(reps * per);
    for (auto& x : ev) x = pool.make();
    // cost of one event record: spacing of back-to-back records on a batch of its own (its size does not depend on `iters`)
    double overhead = 0;
    {
      constexpr size_t kBatch = 64, kSkip = 8;   // the first records carry the queue start-up
      std::vector<hipEvent_t> oe(kBatch);
      for (auto& x : oe) x = pool.make();
      for (size_t i = 0; i < kBatch; ++i) HIPCHK(hipEventRecord(oe[i], stream_));
      HIPCHK(hipStreamSynchronize(stream_));
      float t = 0;
      HIPCHK(hipEventElapsedTime(&t, oe[kSkip], oe[kBatch - 1]));
      overhead = double(t) / double(kBatch - 1 - kSkip);
    }
    for (size_t i = 0; i < reps; ++i) {
      square_chain(r, a, &ev[i * per]);
      if (sub) sub_u32(r, sub);
      HIPCHK(hipEventRecord(ev[i * per + 5], stream_));
    }
    HIPCHK(hipStreamSynchronize(stream_));
    for (size_t i = 0; i < reps; ++i)
      for (size_t k = 0; k < 5 && k < kcount; ++k) {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, ev[i * per + k], ev[i * per + k + 1]));
        kernel_ms[k] += double(t) / double(reps);
      }
    // which of the five slots hold a kernel on this path
    const bool fix_now = !v2cols_ && pl_.C < 2;                        // k_carry_fix right after the back sweep
    const bool coop = coop_on();                                       // (experimental build) one launch (slot 0) for the whole squaring
    const bool sub_kernel = sub != 0 && !((v2cols_ || coop) && sub < (1u << 30));          // k_sub_small (else folded into the next front sweep)
    const bool launched[5] = {true, !coop, !coop, fix_now && !coop, sub_kernel};
    for (size_t k = 0; k < 5 && k < kcount; ++k) kernel_ms[k] = launched[k] ? std::max(0.0, kernel_ms[k] - overhead) : -1.0;
    if (kcount > 5) kernel_ms[5] = overhead;
  }
}

#if defined(MI355_PROBE)
void Engine::probe(int kind, int grid_mult, int extra_lds, int boost_pct, size_t iters, double* avg_ms, uint64_t* tl, size_t tl_words) {
  HIPCHK(hipSetDevice(device_));
  if (!v2cols_ || !v2rows_) throw std::runtime_error("probe: needs the register-resident kernels");
  if (grid_mult < 1 || kind < 0 || kind > 2) throw std::runtime_error("probe: bad arguments");
  const size_t base = (kind == 1) ? pl_.M1 : pl_.tiles(), grid = base * size_t(grid_mult);
  if (tl && tl_words < grid * 8) throw std::runtime_error("probe: timeline buffer too small");
  DevPlan d = dp_;
  d.probe = nullptr; d.probe_mod = uint32_t(base);
  {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_));
    const size_t per_cu = extra_lds >= 20 * 1024 ? 1 : 2;
    const size_t slots = per_cu * size_t(prop.multiProcessorCount);
    const int bp = boost_pct < 0 ? -boost_pct : boost_pct;
    const uint32_t from = (bp > 0 && grid > slots) ? uint32_t(grid - slots * size_t(bp) / 100) : ~0u;
    d.boost_rows = d.boost_tiles = from;
  }
  uint64_t* dtl = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&dtl), grid * 64));
  HIPCHK(hipMemset(dtl, 0, grid * 64));
  uint32_t* dout = reinterpret_cast<uint32_t*>(slot_[0]);   // register 0 is scratch here
  auto launch = [&](const DevPlan& dd) { HIPCHK(v2_probe_launch(dd, kind, grid_mult, extra_lds, digits(1 % nregs_), kind == 2 ? cbuf(0) : nullptr, work(), dout, stream_)); };
  for (int w = 0; w < 3; ++w) launch(d);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipStreamSynchronize(stream_));
  HIPCHK(hipEventRecord(e0, stream_));
  for (size_t i = 0; i < iters; ++i) launch(d);
  HIPCHK(hipEventRecord(e1, stream_));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  if (avg_ms) *avg_ms = double(ms) / double(iters ? iters : 1);
  HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1));
  if (tl) {
    d.probe = dtl;
    // (boost_pct < 0: the instrumented launch follows a launch of ANOTHER kernel, as in a squaring, instead of a launch of itself)
    if (boost_pct < 0) { if (kind == 1) HIPCHK(v2_launch_back(dp_, work(), dout, cbuf(0), 1, 1, stream_)); else run_middle(work(), nullptr, work(), 0, 0); }
    launch(d);
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipMemcpy(tl, dtl, grid * 64, hipMemcpyDeviceToHost));
  }
  HIPCHK(hipFree(dtl));
  HIPCHK(v2_configure());   // restore the LDS attributes
}
#endif

}  // namespace mi355
