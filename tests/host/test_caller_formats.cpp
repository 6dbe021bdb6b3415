// Host-side check of include/mi355/caller_formats.h: prints one JSON object with everything computed, which
// tests/test_host_logic.py compares with the Python mirror (prmers_amd/prp.py, pinned by the reference's golden
// vectors) and with known answers.  usage: t_formats <workdir>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mi355/caller_formats.h"

using namespace mi355::formats;

struct FakeEngine {   // the two checkpoint methods of `engine` (engine.h:142-146)
  mutable std::vector<char> regs;
  size_t get_checkpoint_size() const { return regs.size(); }
  bool get_checkpoint(std::vector<char>& d) const { if (d.size() != regs.size()) return false; d = regs; return true; }
  bool set_checkpoint(const std::vector<char>& d) const { if (d.size() != regs.size()) return false; regs = d; return true; }
};

static std::string jstr(const std::string& s) { return json_escape(s); }

int main(int argc, char** argv) {
  if (argc != 2) return 2;
  const std::string dir = argv[1];
  std::string out = "{";
  // CRC-32 check value
  char buf[32];
  std::snprintf(buf, sizeof buf, "%08X", crc32_update(0, "123456789", 9));
  out += "\"crc_check\":" + jstr(buf);
  // words / div9 / hex of a synthetic digit vector: p = 127, n = 8 (widths 16,16,16,16,16,16,16,15)
  {
    const unsigned w[8] = {16, 16, 16, 16, 16, 16, 16, 15};
    std::vector<uint64_t> d(8);
    for (int i = 0; i < 8; ++i) d[i] = (uint64_t(w[i]) << 32) | uint64_t((0x9e37u * (i + 3) + 77u) & ((1u << w[i]) - 1));
    std::vector<uint32_t> W = pack_words(d, 127);
    out += ",\"res64_raw\":" + jstr(res64_hex(W));
    prp3_div9(127, W);
    out += ",\"res64_div9\":" + jstr(res64_hex(W)) + ",\"res2048_div9\":" + jstr(res2048_hex(W));
  }
  // checkpoint round trip, corruption, wrong mode
  {
    FakeEngine e; e.regs.resize(4096);
    for (size_t i = 0; i < e.regs.size(); ++i) e.regs[i] = char(i * 7 + 1);
    const std::string path = checkpoint_name(9941, false, dir);
    const bool s1 = save_checkpoint(path, e, 9941, false, 1234, 5.5);
    e.regs[10] ^= 1;
    const bool s2 = save_checkpoint(path, e, 9941, false, 2345, 6.5);   // rotates the first one to .old
    FakeEngine f; f.regs.resize(4096);
    uint32_t it = 0; double et = 0;
    const int r_new = load_checkpoint(path, f, 9941, false, it, et);
    const bool same = f.regs == e.regs;
    uint32_t it_old = 0, it_x = 0; double et_old = 0, et_x = 0;
    const int r_old = load_checkpoint(path + ".old", f, 9941, false, it_old, et_old);
    const int r_mode = load_checkpoint(path, f, 9941, true, it_x, et_x);
    const int r_exp = load_checkpoint(path, f, 9949, false, it_x, et_x);
    { std::fstream g(path, std::ios::in | std::ios::out | std::ios::binary); g.seekp(100); char c = 0x55; g.write(&c, 1); }
    const int r_bad = load_checkpoint(path, f, 9941, false, it_x, et_x);
    const int r_none = load_checkpoint(path + ".missing", f, 9941, false, it_x, et_x);
    char b2[256];
    std::snprintf(b2, sizeof b2, ",\"ckpt\":{\"saved\":%d,\"load\":%d,\"iteration\":%u,\"elapsed\":%.1f,\"same\":%d,\"old\":%d,\"old_iteration\":%u,\"wrong_mode\":%d,\"wrong_exponent\":%d,\"corrupt\":%d,\"missing\":%d,\"name_ll\":%s}",
                  int(s1 && s2), r_new, it, et, int(same), r_old, it_old, r_mode, r_exp, r_bad, r_none,
                  jstr(std::filesystem::path(checkpoint_name(607, true, "")).filename().string()).c_str());
    out += b2;
  }
  // worktodo lines
  {
    const char* lines[] = {"PRP=1,2,136279841,-1", "PRP=N/A,1,2,9941,-1,75,0", "PRPDC=0123456789ABCDEF0123456789abcdef,1,2,521,-1", "Test=607",
                           "DoubleCheck=AID,1279,70,1", "Pfactor=1,2,999,-1,70,2", "# PRP=1,2,127,-1", "", "PRP=1,3,127,-1", "Test=N/A,2203,75,1"};
    out += ",\"worktodo\":[";
    for (size_t i = 0; i < sizeof(lines) / sizeof(*lines); ++i) {
      const WorkEntry e = parse_worktodo_line(lines[i]);
      char b3[160];
      std::snprintf(b3, sizeof b3, "%s[%d,%d,%u,%s]", i ? "," : "", int(e.valid), int(e.ll), e.exponent, jstr(e.aid).c_str());
      out += b3;
    }
    out += "]";
    const std::string wt = dir + "/worktodo.txt", sv = dir + "/worktodo_save.txt";
    { std::ofstream f(wt); f << "\nPRP=1,2,127,-1\n# note\nTest=607\n"; }
    bool removed = false;
    const bool more1 = rotate_worktodo(wt, sv, &removed);
    const WorkEntry next = first_worktodo_entry(wt);
    const bool more2 = rotate_worktodo(wt, sv);   // removes "# note" (first non-empty line, as the reference does)
    const bool more3 = rotate_worktodo(wt, sv);
    std::ifstream s(sv); std::string saved, l; while (std::getline(s, l)) saved += l + "|";
    char b4[256];
    std::snprintf(b4, sizeof b4, ",\"rotate\":{\"removed\":%d,\"more1\":%d,\"next\":%u,\"more2\":%d,\"more3\":%d,\"saved\":%s}", int(removed), int(more1), next.exponent,
                  int(more2), int(more3), jstr(saved).c_str());
    out += b4;
  }
  // result lines
  {
    ResultInfo r; r.exponent = 100003; r.is_prime = false; r.res64 = "1CF45E9503C71FD6"; r.res2048 = "ab"; r.gerbicz_errors = 1; r.fft_length = 8192;
    r.program_version = "v"; r.user = "u"; r.computer = "c"; r.aid = "a"; r.timestamp = "t";
    out += ",\"json_prp\":" + jstr(result_json(r));
    r.ll = true; r.is_prime = true; r.user.clear(); r.computer.clear(); r.aid.clear(); r.timestamp.clear();
    out += ",\"json_ll\":" + jstr(result_json(r));
  }
  // proof points
  {
    ProofPoints pp(9941, 3, dir);
    out += ",\"proof_points\":[";
    for (size_t i = 0; i < pp.points().size(); ++i) out += (i ? "," : "") + std::to_string(pp.points()[i]);
    out += "]";
    std::vector<uint32_t> words((9941 + 31) / 32);
    for (size_t i = 0; i < words.size(); ++i) words[i] = uint32_t(i * 2654435761u);
    const uint32_t pt = pp.points()[2];
    const bool saved = pp.save(pt, words), not_a_point = pp.save(pt + 1, words);
    const bool back = pp.load(pt) == words;
    bool corrupt_caught = false;
    { std::fstream g(pp.file_of(pt), std::ios::in | std::ios::out | std::ios::binary); g.seekp(40); char c = 0x11; g.write(&c, 1); }
    try { pp.load(pt); } catch (const std::exception&) { corrupt_caught = true; }
    char b5[200];
    std::snprintf(b5, sizeof b5, ",\"proof\":{\"saved\":%d,\"other_iteration_ignored\":%d,\"round_trip\":%d,\"corruption_caught\":%d,\"valid_to_before\":%d,\"file\":%s}", int(saved),
                  int(!not_a_point), int(back), int(corrupt_caught), int(pp.valid_to(pp.points()[0] - 1)),
                  jstr(std::filesystem::relative(pp.file_of(pt), dir).string()).c_str());
    out += b5;
  }
  out += "}";
  std::puts(out.c_str());
  return 0;
}
