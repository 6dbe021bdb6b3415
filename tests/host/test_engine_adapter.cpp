// The op-level contract the reference checks through engine::create_gpu with its fake plugin
// (tests/test_aevum_reg_adapter.cpp:32-93), here through engine_hip -> C ABI -> MI355X.
// With argument "load": only dlopen + symbol binding (no GPU needed) -- engine_hip's constructor
// must then fail in create() with the library's own message.
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "mi355/engine_hip.h"

static uint64_t value(engine& e, engine::Reg r) {
  mpz_t z; mpz_init(z); e.get_mpz(z, r);
  const uint64_t v = mpz_get_ui(z);
  mpz_clear(z);
  return v;
}
static void expect(uint64_t a, uint64_t b, const char* what) {
  if (a != b) throw std::runtime_error(std::string(what) + ": got " + std::to_string(a) + ", expected " + std::to_string(b));
}

int main(int argc, char** argv) {
  const std::string lib = argc > 2 ? argv[2] : "";
  if (argc > 1 && !std::strcmp(argv[1], "load")) {
    try {
      engine_hip e(31, 8, 0, false, lib);
      std::puts("created");   // a GPU is present
    } catch (const std::exception& ex) {
      std::printf("create failed: %s\n", ex.what());
      return std::strstr(ex.what(), "MI355 create failed") ? 0 : 1;   // symbols bound, create refused
    }
    return 0;
  }
  try {
    constexpr uint32_t p = 31;
    std::unique_ptr<engine> eng(new engine_hip(p, 8, 0, false, lib));
    eng->set(0, 5); eng->set(1, 7);
    eng->set_multiplicand(2, 1);
    eng->mul(0, 2);
    expect(value(*eng, 0), 35, "mul");
    eng->square_mul(0, 3);
    expect(value(*eng, 0), 3675, "square_mul");
    eng->add(0, 1); eng->sub_reg(0, 1); eng->sub(0, 2);
    expect(value(*eng, 0), 3673, "add/sub");
    {   // the run of squarings as one call (an extra of this adapter): 5 Lucas-Lehmer steps from 4 mod 2^31 - 1
      engine_hip* h = static_cast<engine_hip*>(eng.get());
      eng->set(7, 4);
      h->square_mul_n(7, 5, 1, 2);
      uint64_t s = 4; for (int i = 0; i < 5; ++i) s = (s * s + ((uint64_t(1) << 31) - 1) - 2) % ((uint64_t(1) << 31) - 1);
      expect(value(*eng, 7), s, "square_mul_n");
    }
  } catch (const std::exception& ex) {
    std::printf("FAIL %s\n", ex.what());
    return 1;
  }
  try {
    constexpr uint32_t p = 31;
    constexpr uint64_t M = (uint64_t(1) << p) - 1;
    std::unique_ptr<engine> eng(new engine_hip(p, 8, 0, false, lib));
    mpz_t z; mpz_init(z); mpz_set_ui(z, 123); eng->set_mpz(3, z); mpz_clear(z);
    expect(value(*eng, 3), 123, "set_mpz");
    std::vector<char> one(eng->get_register_data_size());
    if (!eng->get_data(one, 3)) return 4;
    eng->set(4, uint32_t(0));
    if (!eng->set_data(4, one)) return 5;
    expect(value(*eng, 4), 123, "register data");
    if (!eng->is_equal(3, 4)) return 10;
    eng->set(4, 124);
    if (eng->is_equal(3, 4)) return 11;
    eng->set(0, 3673);
    std::vector<char> ck(eng->get_checkpoint_size());
    if (!eng->get_checkpoint(ck)) return 6;
    eng->set(0, 1); eng->set(3, 1);
    if (!eng->set_checkpoint(ck)) return 7;
    expect(value(*eng, 0), 3673, "checkpoint r0");
    expect(value(*eng, 3), 123, "checkpoint r3");
    eng->set(5, 9);
    eng->pow(6, 5, 13);
    uint64_t e = 1; for (int i = 0; i < 13; ++i) e = (e * 9) % M;
    expect(value(*eng, 6), e, "pow");
    engine::digit dg(eng.get(), 3);
    if (dg.get_size() != eng->get_size() || dg.res64() != 123) return 8;
    std::vector<char> bad(one.size() - 1);
    if (eng->get_data(bad, 3)) return 12;
    eng->sync();
    std::puts("engine_hip adapter test passed");
    return 0;
  } catch (const std::exception& ex) {
    std::printf("FAIL %s\n", ex.what());
    return 1;
  }
}
