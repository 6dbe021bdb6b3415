// Caller-side formats around the engine boundary, in C++: what the reference's PRP / LL driver reads and
// writes next to the squaring loop, so that a run on the MI355X engine leaves the files a PrMers user expects.
//   * checkpoint file, version 2, CRC trailer      -- src/modes/RunPrpOrLlMarin.cpp:150-211, include/marin/file.h:40-111
//   * worktodo.txt entries and their rotation      -- src/io/WorktodoParser.cpp:78-400,402-427, RunPrpOrLlMarin.cpp:727-751
//   * result line (PrimeNet-style JSON)            -- src/io/JsonBuilder.cpp:322-472
//   * PRP proof checkpoints (residue files)        -- src/core/ProofSetMarin.cpp:56-122,156-158, ProofManagerMarin.cpp:84-120
//   * words / type-1 residue / hex                  -- include/core/AlgoUtils.hpp:165-223
// Header-only, no dependency on the engine library: everything works on an `engine` (include/mi355/engine_iface.h or the
// reference's include/marin/engine.h) or on plain vectors, so it is testable without a GPU (tests/host/test_caller_formats.cpp).
// Formats are the reference's; the code is not (nothing here is copied: same bytes on disk, different implementation).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include "../mi355_engine.h"
#include <cstring>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mi355 {
namespace formats {

// ---- CRC-32 (reflected 0xEDB88320), as file.h:60-84 and ProofSetMarin's computeCRC32 use it -------------------------
inline uint32_t crc32_update(uint32_t crc, const void* data, size_t len) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t r = i;
      for (int k = 0; k < 8; ++k) r = (r & 1u) ? (r >> 1) ^ 0xedb88320u : r >> 1;
      table[i] = r;
    }
    ready = true;
  }
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~crc;
  for (size_t i = 0; i < len; ++i) c = (c >> 8) ^ table[(c ^ p[i]) & 0xffu];
  return ~c;
}

// ---- residue words (AlgoUtils.hpp:165-223) ---------------------------------------------------------------------------
// digits[i] = value | width << 32 (engine::get): little-endian 32-bit words of the p-bit residue
inline std::vector<uint32_t> pack_words(const std::vector<uint64_t>& digits, uint32_t p) {
  std::vector<uint32_t> out((size_t(p) + 31) / 32, 0u);
  size_t bit = 0;
  for (uint64_t d : digits) {
    const unsigned w = unsigned(d >> 32);
    uint64_t v = uint32_t(d);
    if (w < 32) v &= (uint64_t(1) << w) - 1;
    const size_t i = bit / 32, s = bit % 32;
    if (i < out.size()) {
      const uint64_t x = v << s;
      out[i] |= uint32_t(x);
      if (i + 1 < out.size()) out[i + 1] |= uint32_t(x >> 32);
    }
    bit += w;
  }
  return out;
}
// x / 3 mod 2^p - 1 on the word vector (the residue plus the right multiple of Mp is divisible by 3)
inline void div3_words(uint32_t p, std::vector<uint32_t>& W) {
  uint32_t r3 = 0;
  for (uint32_t w : W) r3 = (r3 + w % 3) % 3;
  uint32_t rem = (3 - r3) % 3;   // Mp == 1 (mod 3) for odd p: adding rem * Mp makes the value divisible by 3; rem * 2^p enters at the top
  const unsigned top = p % 32;
  for (size_t i = W.size(); i-- > 0;) {
    const uint64_t t = (uint64_t(rem) << ((i + 1 == W.size()) ? top : 32)) + W[i];   // the top word holds p mod 32 bits
    W[i] = uint32_t(t / 3);
    rem = uint32_t(t % 3);
  }
}
inline void prp3_div9(uint32_t p, std::vector<uint32_t>& W) { div3_words(p, W); div3_words(p, W); }   // type-1 residue of a base-3 PRP
inline std::string res64_hex(const std::vector<uint32_t>& W) {
  const uint64_t r = (uint64_t(W.size() > 1 ? W[1] : 0) << 32) | (W.empty() ? 0u : W[0]);
  char buf[24];
  std::snprintf(buf, sizeof buf, "%016llX", static_cast<unsigned long long>(r));
  return buf;
}
inline std::string res2048_hex(const std::vector<uint32_t>& W) {
  std::string s;
  char buf[12];
  for (int i = 63; i >= 0; --i) { std::snprintf(buf, sizeof buf, "%08x", size_t(i) < W.size() ? W[size_t(i)] : 0u); s += buf; }
  return s;
}

// ---- checkpoint file (RunPrpOrLlMarin.cpp:190-211): int version = 2, u32 p, u32 mode (1 PRP / 2 LL), u32 backend,
//      u32 iteration, double elapsed, the engine's checkpoint image, u32 (~crc ^ 0xa23777ac) over everything before it ----
constexpr uint32_t kBackendMarinOpenCL = 1, kBackendAevum = 2, kBackendMi355 = 3;   // images are backend-specific (:166-170)
inline std::string checkpoint_name(uint32_t p, bool ll, const std::string& dir = ".") {
  return (std::filesystem::path(dir) / (std::string(ll ? "llunsafe_" : "") + "m_" + std::to_string(p) + ".ckpt")).string();
}
template <class Engine>
bool save_checkpoint(const std::string& path, const Engine& eng, uint32_t p, bool ll, uint32_t iteration, double elapsed) {
  std::vector<char> image(eng.get_checkpoint_size());
  if (!eng.get_checkpoint(image)) return false;
  const std::string fresh = path + ".new", old = path + ".old";
  {
    std::ofstream f(fresh, std::ios::binary);
    if (!f) return false;
    uint32_t crc = 0;
    auto put = [&](const void* d, size_t n) { f.write(static_cast<const char*>(d), std::streamsize(n)); crc = crc32_update(crc, d, n); };
    const int version = 2;
    const uint32_t mode = ll ? 2u : 1u, backend = kBackendMi355;
    put(&version, sizeof version); put(&p, 4); put(&mode, 4); put(&backend, 4); put(&iteration, 4); put(&elapsed, sizeof elapsed);
    put(image.data(), image.size());
    const uint32_t trailer = ~crc ^ 0xa23777acu;
    f.write(reinterpret_cast<const char*>(&trailer), 4);
    if (!f.good()) return false;
  }
  std::error_code ec;
  std::filesystem::remove(old, ec);
  if (std::filesystem::exists(path)) std::filesystem::rename(path, old, ec);   // keep the previous one as .old
  std::filesystem::rename(fresh, path, ec);
  return !ec;
}
// 0: loaded (iteration / elapsed set, registers restored); -1: no file; -2: damaged or for another exponent; -3: another mode / backend
template <class Engine>
int load_checkpoint(const std::string& path, const Engine& eng, uint32_t p, bool ll, uint32_t& iteration, double& elapsed) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return -1;
  uint32_t crc = 0;
  auto get = [&](void* d, size_t n) { f.read(static_cast<char*>(d), std::streamsize(n)); if (f.gcount() != std::streamsize(n)) return false; crc = crc32_update(crc, d, n); return true; };
  int version = 0;
  uint32_t rp = 0, mode = 0, backend = 0;
  if (!get(&version, sizeof version) || !get(&rp, 4) || rp != p || version != 2) return -2;
  if (!get(&mode, 4) || !get(&backend, 4)) return -2;
  if (mode != (ll ? 2u : 1u) || backend != kBackendMi355) return -3;
  if (!get(&iteration, 4) || !get(&elapsed, sizeof elapsed)) return -2;
  std::vector<char> image(eng.get_checkpoint_size());
  if (!get(image.data(), image.size())) return -2;
  const uint32_t want = ~crc ^ 0xa23777acu;
  uint32_t trailer = 0;
  f.read(reinterpret_cast<char*>(&trailer), 4);
  if (f.gcount() != 4 || trailer != want) return -2;
  return eng.set_checkpoint(image) ? 0 : -2;
}

// ---- Gerbicz-Li rollback point next to a checkpoint (the reference keeps itersave / jsave in its backup manager and reloads them on
//      a resume, RunPrpOrLlMarin.cpp:251-255).  <ckpt>.gl: u32 magic "GL3R", u32 iteration of the checkpoint it belongs to, u64 itersave,
//      u64 jsave, u64 checkpass, u32 CRC-32 of the 32 bytes before it.  R4 / R5 (the state those counters name) travel in the checkpoint. ----
constexpr uint32_t kGerbiczMagic = 0x474C3352u;
struct GerbiczState { uint64_t itersave = 0, jsave = 0, checkpass = 0; };
inline bool save_gerbicz_state(const std::string& ckpt_path, uint32_t iteration, const GerbiczState& g) {
  unsigned char b[36];
  std::memcpy(b, &kGerbiczMagic, 4); std::memcpy(b + 4, &iteration, 4);
  std::memcpy(b + 8, &g.itersave, 8); std::memcpy(b + 16, &g.jsave, 8); std::memcpy(b + 24, &g.checkpass, 8);
  const uint32_t crc = crc32_update(0, b, 32);
  std::memcpy(b + 32, &crc, 4);
  const std::string fresh = ckpt_path + ".gl.new";
  { std::ofstream f(fresh, std::ios::binary); f.write(reinterpret_cast<const char*>(b), 36); if (!f.good()) return false; }
  // the side file rotates with the checkpoint (<ckpt>.gl.old belongs to <ckpt>.old), so a resume from the older generation -- a torn main
  // checkpoint, or a crash between the two renames -- still finds the rollback point that generation was verified from
  std::error_code ec;
  if (std::filesystem::exists(ckpt_path + ".gl", ec)) std::filesystem::rename(ckpt_path + ".gl", ckpt_path + ".gl.old", ec);
  std::filesystem::rename(fresh, ckpt_path + ".gl", ec);
  return !ec;
}
// the rollback point of the checkpoint written at `iteration`, from whichever generation of the side file names that iteration
inline bool load_gerbicz_state(const std::string& ckpt_path, uint32_t iteration, GerbiczState& g) {
  for (const char* suffix : {".gl", ".gl.old"}) {
    std::ifstream f(ckpt_path + suffix, std::ios::binary);
    unsigned char b[37];
    f.read(reinterpret_cast<char*>(b), 37);
    if (f.gcount() != 36) continue;
    uint32_t magic, it, crc;
    std::memcpy(&magic, b, 4); std::memcpy(&it, b + 4, 4); std::memcpy(&crc, b + 32, 4);
    if (magic != kGerbiczMagic || it != iteration || crc != crc32_update(0, b, 32)) continue;
    std::memcpy(&g.itersave, b + 8, 8); std::memcpy(&g.jsave, b + 16, 8); std::memcpy(&g.checkpass, b + 24, 8);
    return true;
  }
  return false;
}

// ---- worktodo.txt (WorktodoParser.cpp:98-104,331-348): PRP=[aid,]k,b,n,c[,...] / PRPDC= / Test=[aid,]p[,...] / DoubleCheck= ----
struct WorkEntry { bool valid = false; bool ll = false; uint32_t exponent = 0; std::string aid, raw; };
inline WorkEntry parse_worktodo_line(const std::string& line_in) {
  WorkEntry e;
  std::string line = line_in;
  while (!line.empty() && (line.back() == '\r' || line.back() == '\n' || line.back() == ' ')) line.pop_back();
  e.raw = line;
  const size_t eq = line.find('=');
  if (line.empty() || line[0] == '#' || eq == std::string::npos) return e;
  std::string key = line.substr(0, eq);
  std::transform(key.begin(), key.end(), key.begin(), [](unsigned char c) { return char(std::toupper(c)); });
  if (key == "PRP" || key == "PRPDC") e.ll = false;
  else if (key == "TEST" || key == "DOUBLECHECK") e.ll = true;
  else return e;
  std::vector<std::string> parts;
  std::stringstream ss(line.substr(eq + 1));
  for (std::string tok; std::getline(ss, tok, ',');) {
    const size_t a = tok.find_first_not_of(' '), b = tok.find_last_not_of(' ');
    parts.push_back(a == std::string::npos ? std::string() : tok.substr(a, b - a + 1));
  }
  auto is_hex32 = [](const std::string& s) { return s.size() == 32 && s.find_first_not_of("0123456789abcdefABCDEF") == std::string::npos; };
  size_t i = 0;
  if (i < parts.size() && (parts[i].empty() || parts[i] == "N/A")) ++i;
  if (i < parts.size() && (is_hex32(parts[i]) || parts[i] == "AID" || parts[i] == "N/A")) { if (is_hex32(parts[i])) e.aid = parts[i]; ++i; }
  auto is_num = [](const std::string& s) { return !s.empty() && s.find_first_not_of("0123456789") == std::string::npos; };
  if (parts.size() >= i + 4 && parts[i] == "1" && parts[i + 1] == "2" && is_num(parts[i + 2]) && parts[i + 3] == "-1") {
    e.exponent = uint32_t(std::stoul(parts[i + 2])); e.valid = true;           // k,b,n,c = 1,2,p,-1
  } else if (e.ll && i < parts.size() && is_num(parts[i])) {
    e.exponent = uint32_t(std::stoul(parts[i])); e.valid = true;               // Test=p[,how far factored[,P-1 done]]
  }
  return e;
}
inline WorkEntry first_worktodo_entry(const std::string& path) {
  std::ifstream f(path);
  for (std::string l; std::getline(f, l);) { WorkEntry e = parse_worktodo_line(l); if (e.valid) return e; }
  return WorkEntry();
}
// Rotation after a finished entry (removeFirstProcessed, WorktodoParser.cpp:402-427): the first non-empty line moves to
// save_path (appended), the rest stays.  Returns whether another work line remains (the reference then restarts itself).
inline bool rotate_worktodo(const std::string& path, const std::string& save_path, bool* removed = nullptr) {
  std::ifstream in(path);
  std::vector<std::string> keep;
  bool skipped = false;
  std::string first;
  for (std::string l; std::getline(in, l);) {
    if (!skipped && !l.empty()) { skipped = true; first = l; continue; }
    keep.push_back(l);
  }
  in.close();
  if (removed) *removed = skipped;
  if (!skipped) return false;
  { std::ofstream save(save_path, std::ios::app); save << first << "\n"; }
  const std::string tmp = path + ".tmp";
  { std::ofstream out(tmp); for (const auto& l : keep) out << l << "\n"; }
  std::error_code ec;
  std::filesystem::rename(tmp, path, ec);
  for (const auto& l : keep) if (!l.empty() && l[0] != '#') return true;
  return false;
}

// ---- result line (JsonBuilder.cpp:322-472: key order of the PRP / LL work types) ----------------------------------------
struct ResultInfo {
  uint32_t exponent = 0; bool ll = false, is_prime = false;
  std::string res64, res2048; unsigned gerbicz_errors = 0, fft_length = 0;
  std::string program_version = "mi355-marin-hip " MI355_ENGINE_VERSION, os_name = "Linux", os_arch = "x86_64", user, computer, aid, timestamp;
  unsigned port = 8;
};
inline std::string json_escape(const std::string& s) {
  std::string o = "\"";
  for (char c : s) { if (c == '"' || c == '\\') { o += '\\'; o += c; } else if (c == '\n') o += "\\n"; else o += c; }
  return o + "\"";
}
inline std::string result_json(const ResultInfo& r) {
  std::ostringstream o;
  o << "{\"status\":" << json_escape(r.is_prime ? "P" : "C") << ",\"exponent\":" << r.exponent << ",\"worktype\":" << json_escape(r.ll ? "LL" : "PRP-3")
    << ",\"res64\":" << json_escape(r.res64);
  if (!r.ll) o << ",\"res2048\":" << json_escape(r.res2048) << ",\"residue-type\":1";
  o << ",\"errors\":{\"gerbicz\":" << r.gerbicz_errors << "},\"shift-count\":0";
  if (r.fft_length) o << ",\"fft-length\":" << r.fft_length;
  o << ",\"program\":{\"name\":\"prmers\",\"version\":" << json_escape(r.program_version) << ",\"port\":" << r.port << "}";
  o << ",\"os\":{\"os\":" << json_escape(r.os_name);
  if (!r.os_arch.empty()) o << ",\"architecture\":" << json_escape(r.os_arch);
  o << "}";
  if (!r.user.empty()) o << ",\"user\":" << json_escape(r.user);
  if (!r.computer.empty()) o << ",\"computer\":" << json_escape(r.computer);
  if (!r.aid.empty()) o << ",\"aid\":" << json_escape(r.aid);
  if (!r.timestamp.empty()) o << ",\"timestamp\":" << json_escape(r.timestamp);
  o << "}";
  return o.str();
}

// ---- PRP proof checkpoints (ProofSetMarin.cpp:56-122): the 2^power iterations whose residues a proof of that power needs;
//      each residue is stored as <p>/proof/<iteration>: u32 CRC-32 of the words, then the ceil(p/32) little-endian words ----
class ProofPoints {
 public:
  ProofPoints(uint32_t exponent, uint32_t power, const std::string& base_dir = ".") : p_(exponent), power_(power) {
    dir_ = (std::filesystem::path(base_dir) / std::to_string(exponent) / "proof").string();
    points_.push_back(0);
    uint32_t span = (exponent + 1) / 2;
    for (uint32_t level = 0; level < power; ++level, span = (span + 1) / 2) {
      const size_t have = points_.size();
      for (size_t i = 0; i < have; ++i) points_.push_back(points_[i] + span);
    }
    points_.front() = exponent;   // the residue after all p squarings replaces iteration 0
    std::sort(points_.begin(), points_.end());
  }
  const std::vector<uint32_t>& points() const { return points_; }
  bool should_checkpoint(uint32_t iteration) const { return std::binary_search(points_.begin(), points_.end(), iteration); }
  std::string file_of(uint32_t iteration) const { return (std::filesystem::path(dir_) / std::to_string(iteration)).string(); }
  // words: canonical residue after `iteration` squarings (pack_words of engine::get); no-op for other iterations
  bool save(uint32_t iteration, const std::vector<uint32_t>& words) const {
    if (!should_checkpoint(iteration)) return false;
    std::filesystem::create_directories(dir_);
    std::ofstream f(file_of(iteration), std::ios::binary);
    const uint32_t crc = crc32_update(0, words.data(), words.size() * 4);
    f.write(reinterpret_cast<const char*>(&crc), 4);
    f.write(reinterpret_cast<const char*>(words.data()), std::streamsize(words.size() * 4));
    return f.good();
  }
  std::vector<uint32_t> load(uint32_t iteration) const {
    std::ifstream f(file_of(iteration), std::ios::binary);
    if (!f) throw std::runtime_error("cannot open proof checkpoint " + file_of(iteration));
    uint32_t crc = 0;
    std::vector<uint32_t> words((size_t(p_) + 31) / 32);
    f.read(reinterpret_cast<char*>(&crc), 4);
    f.read(reinterpret_cast<char*>(words.data()), std::streamsize(words.size() * 4));
    if (!f.good() || crc != crc32_update(0, words.data(), words.size() * 4)) throw std::runtime_error("damaged proof checkpoint " + file_of(iteration));
    return words;
  }
  // every residue up to and including iteration `limit` is on disk (ProofSetMarin::isValidTo)
  bool valid_to(uint32_t limit) const {
    for (uint32_t pt : points_) { if (pt > limit) break; if (pt < p_ && !std::filesystem::exists(file_of(pt))) return false; }
    return true;
  }
 private:
  uint32_t p_, power_;
  std::string dir_;
  std::vector<uint32_t> points_;
};

}  // namespace formats
}  // namespace mi355
