// fuzz_gob.cpp -- host-mode fuzz of pgpu_gob_unpack for the CPU suite (tests/test_wire_asan.py): wire.cpp compiled with
// AddressSanitizer + UBSan and linked with the few symbols it takes from the rest of the library, no GPU call on this path
// (ctx = NULL, PGPU_MEM_HOST).  Seeds: blobs pgpu_gob_pack writes; mutations: byte flips, truncations, 64-bit varints spliced
// in front of every byte position (field deltas, lengths, counts are all attacker-controlled; ADVICE r4 wire.cpp:193).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../paillier_amd/csrc/engine.hpp"

namespace pgi {
thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
[[noreturn]] void api_throw(int code, const std::string& m) { throw ApiError{code, m}; }
void wipe(void* p, size_t n) { memset(p, 0, n); }
}  // namespace pgi

// device helpers wire.cpp launches in PGPU_MEM_DEVICE mode only: never reached here
void launch_gob_emit(const uint8_t*, size_t, const uint32_t*, const uint64_t*, size_t, const uint8_t*, uint32_t, const uint8_t*,
                     uint32_t, const uint8_t*, uint32_t, uint8_t*, hipStream_t) { abort(); }
void launch_be_lengths(const uint8_t*, size_t, size_t, uint32_t*, hipStream_t) { abort(); }
void launch_bytes_gather_be(const uint8_t*, const uint64_t*, const uint32_t*, size_t, uint8_t*, size_t, hipStream_t) { abort(); }

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return rng_state;
}

int main(int argc, char** argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 200000;
  const size_t stride = 64;
  // seeds: one blob per (level, method) with a few magnitudes
  std::vector<std::vector<uint8_t>> seeds;
  for (int level = 0; level < 2; ++level)
    for (int method = 0; method < 3; ++method) {
      uint8_t rows[3 * 64];
      for (size_t i = 0; i < sizeof rows; ++i) rows[i] = (uint8_t)rnd();
      memset(rows, 0, 60);                                                           // a short magnitude
      std::vector<uint8_t> blobs(3 * pgpu_gob_max_bytes(stride));
      size_t offs[4];
      if (pgpu_gob_pack(nullptr, 3, rows, stride, PGPU_MEM_HOST, level, method, blobs.data(), blobs.size(), offs) != PGPU_OK) {
        fprintf(stderr, "pack failed: %s\n", pgi::g_err.c_str());
        return 2;
      }
      for (int i = 0; i < 3; ++i) seeds.emplace_back(blobs.begin() + offs[i], blobs.begin() + offs[i + 1]);
    }
  long ok = 0, rejected = 0;
  std::vector<uint8_t> out(stride);
  for (long it = 0; it < iters; ++it) {
    std::vector<uint8_t> b = seeds[rnd() % seeds.size()];
    const int kind = (int)(rnd() % 6);
    const size_t pos = rnd() % b.size();
    if (kind == 0) b[pos] = (uint8_t)rnd();
    else if (kind == 1) b.resize(pos);
    else if (kind == 2) {                                                             // a 64-bit varint in place of one byte
      uint8_t v[9] = {0xf8};
      const uint64_t x = (rnd() & 1) ? ~0ull - (rnd() % 4) : rnd();
      for (int j = 0; j < 8; ++j) v[1 + j] = (uint8_t)(x >> (8 * (7 - j)));
      b.erase(b.begin() + (long)pos);
      b.insert(b.begin() + (long)pos, v, v + 9);
    } else if (kind == 3) b.insert(b.begin() + (long)pos, (uint8_t)(rnd() % 3 ? 0xff : 0xf7));
    else if (kind == 4) { for (int j = 0; j < 4; ++j) b[rnd() % b.size()] = (uint8_t)rnd(); }
    else b.erase(b.begin() + (long)pos);
    // an exact-size heap copy, so that ASan sees a read past the blob
    uint8_t* heap = (uint8_t*)malloc(b.size() ? b.size() : 1);
    memcpy(heap, b.data(), b.size());
    const size_t offs[2] = {0, b.size()};
    int32_t level = 0, method = 0;
    const int rc = pgpu_gob_unpack(nullptr, 1, heap, offs, out.data(), stride, PGPU_MEM_HOST, &level, &method);
    free(heap);
    if (rc == PGPU_OK) ++ok;
    else if (rc == PGPU_ERR_INVALID) ++rejected;
    else { fprintf(stderr, "unexpected status %d: %s\n", rc, pgi::g_err.c_str()); return 3; }
  }
  printf("fuzz_gob ok: %ld accepted, %ld rejected\n", ok, rejected);
  return ok > 0 && rejected > 0 ? 0 : 4;
}
