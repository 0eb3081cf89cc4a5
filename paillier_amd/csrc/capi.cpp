// capi.cpp -- C ABI (include/paillier_hip.h) of the batched Paillier engine: contexts, keys,
// Montgomery constants, VM program generation and kernel orchestration.  Host code here runs once
// per key or once per batch call; everything per ciphertext happens in kernels.hip on the GPU.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <memory>
#include <mutex>
#include <set>
#include <sys/random.h>
#include <thread>
#include <string>
#include <vector>

#include "../../include/paillier_hip.h"
#include "hostbig.hpp"
#include "kernels.h"

using hostbig::BigU;

namespace {

constexpr int LB = 28;
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

struct HipError { hipError_t e; const char* what; };
#define HIPCHK(x)                                                      \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) throw HipError{e_, #x};                      \
  } while (0)

struct ApiError { int code; std::string msg; };
[[noreturn]] void api_throw(int code, const std::string& m) { throw ApiError{code, m}; }

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// overwrite key material before its storage is released (not elided: volatile stores)
void wipe(void* p, size_t n) {
  volatile unsigned char* v = (volatile unsigned char*)p;
  while (n--) *v++ = 0;
}
template <class T> void wipe_vec(std::vector<T>& v) { if (!v.empty()) wipe(v.data(), v.size() * sizeof(T)); }
// wipes host copies of secret exponents on EVERY exit of the enclosing scope, exceptions included
template <class V> struct WipeOnExit {
  V& v;
  explicit WipeOnExit(V& v_) : v(v_) {}
  ~WipeOnExit() { for (auto& e : v) wipe_vec(e.d); }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// Context: device, stream, workspace pool, profile events
// ------------------------------------------------------------------------------------------------
struct pgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;   // created by pgpu_ctx_create(device, PGPU_STREAM_NEW): destroyed with the context
  // workspace: list of chunks, bump allocated, reset per API call
  struct Chunk { char* p; size_t cap; size_t used; };
  std::vector<Chunk> chunks;
  std::vector<std::vector<uint32_t>> host_keep;  // host buffers that async copies read from
  // profile of the last call
  struct Ev { hipEvent_t a, b; double mads; char name[32]; };
  std::vector<Ev> evs;
  size_t evs_used = 0;
  bool use_asm = true;       // hand-scheduled VM kernels (pgpu_ctx_set_flag("asm", 0) selects the hipcc-generated ones)
  int last_vm_asm = 0;       // number of VM launches of the last call that ran the assembly kernel
  int last_vm_launches = 0;  // number of VM launches of the last call, assembly or compiler-generated
  bool use_pair = true;      // Decrypt ladders mod p^2 on the pair kernel (pgpu_ctx_set_flag("pair", 0): the 2H-limb kernel)
  bool use_triple = true;    // ladders modulo n^3 on the three-digit kernel (pgpu_ctx_set_flag("triple", 0): the 3H-limb kernels)
  bool use_shared_chain = true;   // several shared exponents on ONE base share the chain of squarings (pgpu_partial_decrypt_multi)
  bool use_lift = true;           // level-two Encrypt: r^(n^2) mod n^3 as (r^n mod n^2)^n mod n^3
  size_t lanes_wanted = 0;   // 0: default occupancy target; tests set 1 to run every modulus at its natural shape
  // A second stream for work of a call that depends on no ladder in flight (SideStream below; the DDLEQ prover's
  // per-statement chains run beside its big launches).  pgpu_ctx_set_flag("side", 0): everything on the one stream.
  bool use_nm4 = true;       // per-number 4- / 5-bit window tables of the pair kernels number-major (pgpu_ctx_set_flag("nm4", 0): limb-major, VM_MULV / VM_MULV5)
  bool use_early = true;     // the DDLEQ prover prepares its response for every statement / instance beside the Alpha ladders (pgpu_ctx_set_flag("early", 0): after the hash, for the bit-1 instances)
  bool use_handover = true;  // a power modulo n^2 that is only needed modulo n^2 by the next ladder modulo n^3 stays in pair form: (a0, a1, 0) is its digit form (pgpu_ctx_set_flag("handover", 0): exit and re-entry)
  bool use_muls = true;      // bucket products of the shared chain as VM_MULS where the kernel has it (pgpu_ctx_set_flag("muls", 0): LOAD / MUL / STORE)
  bool use_lanes8 = true;    // shards too small for four lanes per number take the eight-lane pair kernel (pgpu_ctx_set_flag("lanes8", 0): never)
  bool use_side = true;
  hipStream_t side = nullptr;
  hipStream_t side_l[3] = {nullptr, nullptr, nullptr};   // further lanes: chains of small kernels of one CRT half / operand beside the others'
  std::vector<hipEvent_t> sync_evs;
  size_t sync_used = 0;
  hipEvent_t next_sync_ev() {
    if (sync_used == sync_evs.size()) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      sync_evs.push_back(e);
    }
    return sync_evs[sync_used++];
  }

  void bind() { HIPCHK(hipSetDevice(device)); }

  void reset_ws() {
    // Chunks are kept across calls (the same call sequence lands in the same chunks again: no allocation in steady
    // state).  Only a long tail of chunks is consolidated: that costs a device synchronisation and a large hipMalloc
    // (~100 ms), which must not land in a caller's second call.
    if (chunks.size() > 12) {
      HIPCHK(hipStreamSynchronize(stream));
      size_t total = 0;
      for (auto& c : chunks) { total += c.cap; HIPCHK(hipFree(c.p)); }
      chunks.clear();
      Chunk c{nullptr, round_up(total, 1 << 20), 0};
      HIPCHK(hipMalloc((void**)&c.p, c.cap));
      chunks.push_back(c);
    }
    for (auto& c : chunks) c.used = 0;
    sync_used = 0;
    for (auto& h : host_keep) wipe_vec(h);   // ladder programs encode secret exponents (p - 1, q - 1, shares)
    host_keep.clear();
    evs_used = 0;
    last_vm_asm = 0;
    last_vm_launches = 0;
  }
  void* ws(size_t bytes) {
    bytes = round_up(bytes ? bytes : 1, 256);
    for (auto& c : chunks)
      if (c.cap - c.used >= bytes) { void* p = c.p + c.used; c.used += bytes; return p; }
    // geometric growth (a new chunk is at least as large as everything before it, up to 64 GiB of the 288): the list stays short, so
    // the consolidation in reset_ws() -- a free and a multi-GiB hipMalloc inside some later call -- stays rare
    size_t total = 0;
    for (auto& k : chunks) total += k.cap;
    Chunk c{nullptr, std::max(round_up(bytes, 64 << 20), std::min<size_t>(total, (size_t)64 << 30)), bytes};
    HIPCHK(hipMalloc((void**)&c.p, c.cap));
    chunks.push_back(c);
    return c.p;
  }
  template <class T> T* ws_t(size_t n) { return (T*)ws(n * sizeof(T)); }

  uint32_t* upload_words(const std::vector<uint32_t>& v) {
    host_keep.push_back(v);
    uint32_t* d = ws_t<uint32_t>(v.size());
    HIPCHK(hipMemcpyAsync(d, host_keep.back().data(), v.size() * 4, hipMemcpyHostToDevice, stream));
    return d;
  }
  Ev& next_ev() {
    if (evs_used == evs.size()) {
      Ev e;
      HIPCHK(hipEventCreate(&e.a));
      HIPCHK(hipEventCreate(&e.b));
      e.mads = 0;
      e.name[0] = 0;
      evs.push_back(e);
    }
    return evs[evs_used++];
  }
  // zero the workspace (intermediate values of the last call, ladder programs of secret exponents)
  void wipe_ws() {
    if (side) (void)hipStreamSynchronize(side);
    for (auto l : side_l) if (l) (void)hipStreamSynchronize(l);
    for (auto& c : chunks) (void)hipMemsetAsync(c.p, 0, c.cap, stream);
    (void)hipStreamSynchronize(stream);
    for (auto& h : host_keep) wipe_vec(h);
  }
  ~pgpu_ctx() {
    wipe_ws();
    for (auto& c : chunks) (void)hipFree(c.p);
    for (auto& e : evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : sync_evs) (void)hipEventDestroy(e);
    if (side) (void)hipStreamDestroy(side);
    for (auto l : side_l) if (l) (void)hipStreamDestroy(l);
    if (own_stream) (void)hipStreamDestroy(stream);
  }
};

// Work of ONE call on two streams.  Every helper of this file issues to ctx->stream; between enter() and leave() that is the
// side stream.  enter(after) orders the side work behind an event of the main stream (mark()), leave() goes back WITHOUT making
// the main stream wait, join() makes the main stream wait for everything the side stream was given.  Host-side synchronisation
// inside side work (the root of an inversion tree goes through the host) waits for the side stream only: issue the main
// stream's long launch BEFORE entering, and it runs meanwhile.  Disabled (ctx->use_side == false): everything stays on the one
// stream, in program order -- the same results.
struct SideStream {
  pgpu_ctx* c;
  hipStream_t main_stream;
  hipStream_t& s;                           // the lane's stream (created on first use)
  bool on, entered = false, dirty = false;
  // lane 0: work of a call that is independent of its ladders (the prover's per-statement chains).  lanes 1 .. 3: independent
  // CHAINS of small kernels -- the entry into digit form of the q-half next to the p-half's, of operand y next to x's; each such
  // kernel fills a fraction of the chip for tens of microseconds, chains side by side take about the time of one.
  explicit SideStream(pgpu_ctx* c_, int lane = 0) : c(c_), main_stream(c_->stream), s(lane ? c_->side_l[lane - 1] : c_->side), on(c_->use_side) {
    if (on && !s) HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (on && s == main_stream) on = false;   // (nested use of a lane from inside itself: stay in line)
  }
  hipEvent_t mark() {                       // "everything issued to the main stream so far"
    if (!on) return nullptr;
    hipEvent_t e = c->next_sync_ev();
    HIPCHK(hipEventRecord(e, main_stream));
    return e;
  }
  void enter(hipEvent_t after) {
    if (!on) return;
    if (after) HIPCHK(hipStreamWaitEvent(s, after, 0));
    c->stream = s;
    entered = dirty = true;
  }
  void leave() {
    if (!on) return;
    c->stream = main_stream;
    entered = false;
  }
  void join() {
    if (!on || !dirty) return;
    if (entered) leave();
    hipEvent_t e = c->next_sync_ev();
    HIPCHK(hipEventRecord(e, s));
    HIPCHK(hipStreamWaitEvent(main_stream, e, 0));
    dirty = false;
  }
  ~SideStream() {                           // error paths: never leave the context on the side stream or the side stream busy
    if (!on) return;
    c->stream = main_stream;
    if (dirty) (void)hipStreamSynchronize(s);
  }
};

// Up to four independent chains of small kernels side by side:
//     Fork f(ctx);  for (k : {0, 1}) { f.chain(k); ... }  f.join();            (Fork f(ctx, 4): chains 0 .. 3)
// chain(0) is the stream the context was on; chain(k > 0) moves the context to side lane k (ordered behind everything issued before
// the Fork was made; calls with the same k follow each other on their lane); join() comes back and makes the main stream wait
// for every lane.  The chains must not share a buffer that one of them writes.  With the side streams off: plain program order.
struct Fork {
  std::vector<std::unique_ptr<SideStream>> lanes;
  hipEvent_t start = nullptr;
  int cur = 0;
  explicit Fork(pgpu_ctx* c, int n = 2) {
    for (int l = 1; l < n && l <= 3; ++l) lanes.emplace_back(new SideStream(c, l));
    if (!lanes.empty()) start = lanes[0]->mark();
  }
  void chain(int k) {
    k %= (int)lanes.size() + 1;
    if (cur > 0) lanes[(size_t)cur - 1]->leave();
    cur = k;
    if (k > 0) {
      // (a lane whose stream is the one the context is on -- nested use -- stays in line: SideStream switched itself off)
      lanes[(size_t)k - 1]->enter(lanes[(size_t)k - 1]->dirty ? nullptr : start);
    }
  }
  void join() {
    if (cur > 0) lanes[(size_t)cur - 1]->leave();
    cur = 0;
    for (auto& l : lanes) l->join();
  }
};

// ------------------------------------------------------------------------------------------------
// Modulus context: Montgomery constants for one odd modulus on the device
// ------------------------------------------------------------------------------------------------
namespace {

enum { C_R2 = 0, C_R3 = 1, C_ONE_M = 2, C_ONE = 3, C_USER = 4 };

struct ModCtx;
// A modulus N = n^2 whose root n is known can run its shared-exponent ladders on the two-lane pair kernel (GenQ).
struct PairInfo {
  const ModCtx* root = nullptr;       // n
  const uint32_t* consts = nullptr;   // device: n | Cadj (H limbs each) + one word of padding
  int c_rh = -1;                      // index of R_H mod n^2 (plain) in the consts of n^2
  int c_one_pair = -1;                // index of the pair form of 1 (the digits of R_H mod n^2) in the consts of n^2
  const uint32_t* dinv = nullptr;     // n^-1 mod 2^(28 H)
  const uint32_t* n_limbs = nullptr;  // n as H limbs
  // the eight-lane pair kernel (GenQ8: every digit in four lanes of h8 / 4 limbs; h8 = H rounded up to a multiple of 4):
  int h8 = 0;
  const uint32_t* consts8 = nullptr;  // device: n | Cadj, h8 limbs each, + one word of padding
  const uint32_t* tconsts8 = nullptr; // device: [3][2 h8] pair digits of R_h8^2 R_H^-1 (radix R_H -> R_h8), of R_H (back), of R_h8 (= 1)
};

// A modulus N = n^3 whose root n is known can run its ladders on the three-digit kernel (GenQ3): residues as
// a0 + a1 n + a2 n^2 in the lanes of a quad.
struct TripleInfo {
  const ModCtx* root = nullptr;       // n
  const ModCtx* mid = nullptr;        // n^2 (digit split)
  const uint32_t* kconsts = nullptr;  // device: n (padded to an even count) | (C1_i, C2_i) pairs | two words of padding
  const uint32_t* tconsts = nullptr;  // device: constants in digit form, [c][3H]; entry 0 = the digit form of 1
  int c_rh = -1;                      // in the consts of n^3: R_H mod n^3 (plain): entry into the digit form
  int c_exit = -1;                    // in the consts of n^3: R R_H^-1 mod n^3 (plain): exit from it
  const uint32_t* dinv1 = nullptr;    // n^-1 mod 2^(28 WT(n))
  const uint32_t* dinv2 = nullptr;    // n^-1 mod 2^(28 WT(n^2))
  const uint32_t* n_limbs = nullptr;  // n   as WT(n) limbs
  const uint32_t* n2_limbs = nullptr; // n^2 as WT(n^2) limbs
  bool lanes6_only = false;           // the digit does not fit one lane (H > 74): only the two-lanes-per-digit kernel (GenQ6) serves it,
                                      // and that one runs shared-exponent and limb-major per-number programs only
};

struct ModCtx {
  pgpu_ctx* ctx = nullptr;
  PairInfo pairn;
  TripleInfo triple;
  BigU N, R;
  size_t nbits = 0, nbytes = 0;
  int WL = 0, K = 0, WT = 0;
  uint32_t n0inv = 0;
  std::vector<BigU> consts;
  uint32_t* d_nmod = nullptr;
  uint32_t* d_consts = nullptr;
  size_t d_consts_cap = 0;

  static bool pick_shape(size_t bits, int& wl, int& k) {
    // (the widest shape is eight lanes of 42 limbs: 83 limbs in four lanes -- the shape of rounds 1 and 2 -- does not fit the
    // register file next to its accumulators, so it existed in the compiler-generated kernel only, at 31 % of the issue peak)
    struct S { int wl, k; } shapes[] = {{37, 1}, {55, 1}, {74, 1}, {55, 2}, {74, 2}, {55, 4}, {74, 4}, {42, 8}};
    for (auto s : shapes)
      if (bits + 3 <= (size_t)LB * s.wl * s.k) { wl = s.wl; k = s.k; return true; }
    return false;
  }

  void init(pgpu_ctx* c, const BigU& n) {
    ctx = c;
    N = n;
    if (!N.is_odd() || N.bit_length() < 2) api_throw(PGPU_ERR_INVALID, "modulus must be odd and at least 3");
    nbits = N.bit_length();
    nbytes = (nbits + 7) / 8;
    if (!pick_shape(nbits, WL, K)) api_throw(PGPU_ERR_UNSUPPORTED, "modulus wider than 9405 bits is not built");
    WT = WL * K;
    R = hostbig::shl(BigU(1), (size_t)LB * WT);
    uint32_t n0 = N.d[0], x = n0;  // Newton: x = n0^-1 mod 2^32
    for (int i = 0; i < 6; ++i) x *= 2u - n0 * x;
    n0inv = (0u - x) & ((1u << LB) - 1);
    BigU r1 = R % N, r2 = hostbig::mulmod(r1, r1, N), r3 = hostbig::mulmod(r2, r1, N);
    consts = {r2, r3, r1, BigU(1)};
  }
  int add_const(const BigU& v) {  // v < 2N (any value below R works as a Montgomery operand)
    consts.push_back(v);
    return (int)consts.size() - 1;
  }
  // Montgomery form of v
  BigU to_mont(const BigU& v) const { return hostbig::mulmod(v % N, R % N, N); }
  void upload() {
    ctx->bind();
    if (!d_nmod) HIPCHK(hipMalloc((void**)&d_nmod, (size_t)WT * 4));
    std::vector<uint32_t> nl = N.to_limbs(LB, WT);
    HIPCHK(hipMemcpy(d_nmod, nl.data(), nl.size() * 4, hipMemcpyHostToDevice));
    if (consts.size() > d_consts_cap) {
      if (d_consts) HIPCHK(hipFree(d_consts));
      d_consts_cap = consts.size() + 8;
      HIPCHK(hipMalloc((void**)&d_consts, d_consts_cap * WT * 4));
    }
    std::vector<uint32_t> all;
    for (auto& c : consts) {
      auto l = c.to_limbs(LB, WT);
      all.insert(all.end(), l.begin(), l.end());
    }
    HIPCHK(hipMemcpy(d_consts, all.data(), all.size() * 4, hipMemcpyHostToDevice));
  }
  ~ModCtx() {   // a modulus may be secret (p, q and their powers): nothing of it outlives the handle
    if (d_nmod) { (void)hipMemset(d_nmod, 0, (size_t)WT * 4); (void)hipFree(d_nmod); }
    if (d_consts) { (void)hipMemset(d_consts, 0, d_consts_cap * WT * 4); (void)hipFree(d_consts); }
    wipe_vec(N.d);
    for (auto& c : consts) wipe_vec(c.d);
  }
};

// device copy of an arbitrary constant as `w` canonical 28-bit limbs
struct DevLimbs {
  uint32_t* d = nullptr;
  int w = 0;
  void set(const BigU& v, int width) {
    w = width;
    auto l = v.to_limbs(LB, width);
    if (!d) HIPCHK(hipMalloc((void**)&d, (size_t)width * 4));
    HIPCHK(hipMemcpy(d, l.data(), l.size() * 4, hipMemcpyHostToDevice));
  }
  ~DevLimbs() { if (d) { (void)hipMemset(d, 0, (size_t)w * 4); (void)hipFree(d); } }
};

// ------------------------------------------------------------------------------------------------
// VM programs
// ------------------------------------------------------------------------------------------------
// pgpu_ctx_set_flag("fair", 0): no priority bits in the programs.  A measurement switch (A/B runs of the wave-priority scheme),
// deliberately PROCESS-wide -- programs are built without a context at hand -- and atomic: other contexts' threads read it
// while they build programs.  Product code never clears it.
static std::atomic<bool> g_wave_priorities{true};
struct Prog {
  std::vector<uint32_t> w;
  ~Prog() { wipe_vec(w); }   // a ladder program spells out its exponent window by window: p - 1, q - 1, lambda, shares ...
  double montmuls = 0, sqrs = 0;
  bool asm_ok = true;  // only opcodes the assembly kernel implements
  bool has_mulv = false;
  bool wide_gathers = false;  // table opcodes other than the 4-bit VM_MULV
  bool nm_tables = false;     // VM_MULV7 / VM_STORET: among the assembly kernels only the three-digit ones implement them
  bool needs_muls = false;    // VM_MULS: the four- and eight-lane pair kernels only
  bool nm4 = false;           // VM_MULVT (with VM_STORET): the one-lane pair kernel for 37-limb primes only
  bool mulv7 = false;
  uint32_t gather_slots = 1;  // slots a per-number gather spans (table entries + 1): its offsets are 32-bit in the assembly kernels
  void op(uint32_t o, uint32_t arg = 0, uint32_t aux = 0) {
    if (o == VM_SETOFF) asm_ok = false;
    if (o == VM_MULV7 || o == VM_STORET || o == VM_MULVT || o == VM_MULVT5) nm_tables = true;
    if (o == VM_MULVT || o == VM_MULVT5) nm4 = true;
    if (o == VM_MULV7) mulv7 = true;
    if (o == VM_MULS) needs_muls = true;
    if (o == VM_MULCV || o == VM_MULV5 || o == VM_MULV7 || o == VM_STORET || o == VM_MULVT || o == VM_MULVT5) wide_gathers = true;   // (kernels without these opcodes must not get the program)
    if (o == VM_MULV || o == VM_MULV5 || o == VM_MULV7 || o == VM_MULVT || o == VM_MULVT5) {
      has_mulv = true;
      gather_slots = std::max<uint32_t>(gather_slots, o == VM_MULV ? 17u : o == VM_MULVT ? 18u : o == VM_MULV5 ? 33u : o == VM_MULVT5 ? 49u : 129u);
    }
    if (aux >> 22) api_throw(PGPU_ERR_INVALID, "internal: table slot does not fit the instruction word");
    w.push_back(o | (aux << 8));
    w.push_back(arg);
    if (o == VM_SQR || o == VM_MUL || o == VM_MULC || o == VM_MULV || o == VM_MULCV || o == VM_MULV5 || o == VM_MULV7 || o == VM_MULS || o == VM_MULVT || o == VM_MULVT5) montmuls += 1;
    if (o == VM_SQR) sqrs += 1;
  }
  // Bits 30..31 of an instruction word: the priority the wave takes when it gets there -- 3, 2, 1, 0 over four stretches of a
  // long program (by products done), so that of two waves sharing a SIMD the one that leads yields to the one behind
  // (gen_vm_asm.py fair_share; the hipcc kernels ignore the bits).  What is lost is the END of the launch, where the wave
  // that finishes first leaves the other one alone on the SIMD for most of the last stretch -- so the stretches shrink
  // geometrically (80 %, 16 %, 3.2 %, 0.8 %: the wave that yields crawls at ~7 % of the other's speed, so a stretch has to be
  // longer than 7 % of the one before it or the leader would run out of program while the other is still catching up).
  void end() {
    op(VM_END);
    if (montmuls < 256 || !g_wave_priorities.load(std::memory_order_relaxed)) return;
    double done = 0;
    for (size_t i = 0; i + 1 < w.size(); i += 2) {
      const uint32_t o = w[i] & 0xFFu;
      const double f = done / montmuls;
      const uint32_t quarter = f < 0.80 ? 0u : f < 0.96 ? 1u : f < 0.992 ? 2u : 3u;
      w[i] = (w[i] & 0x3FFFFFFFu) | ((3u - quarter) << 30);
      if (o == VM_SQR || o == VM_MUL || o == VM_MULC || o == VM_MULV || o == VM_MULCV || o == VM_MULV5 || o == VM_MULV7 || o == VM_MULS || o == VM_MULVT || o == VM_MULVT5) done += 1;
    }
  }
};

constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;

// x <- Montgomery form of the input held in slot lo (+ hi * R when hi != NO_SLOT); lazy, < 2N
void emit_to_mont(Prog& p, uint32_t lo, uint32_t hi, uint32_t tmp) {
  if (hi != NO_SLOT) {
    p.op(VM_LOAD, hi);
    p.op(VM_MULC, C_R3);   // hi * R^2
    p.op(VM_STORE, tmp);
    p.op(VM_LOAD, lo);
    p.op(VM_MULC, C_R2);   // lo * R
    p.op(VM_ADD, tmp);     // (hi R + lo) R, limbs <= 2^29 + 2
    p.op(VM_MULC, C_ONE_M);  // * R * R^-1: renormalise (lazy < 2N)
  } else {
    p.op(VM_LOAD, lo);
    p.op(VM_MULC, C_R2);
  }
}

// Shared-exponent fixed-window (w = 5) modexp.  Table slots tab .. tab+31.
// post_slot != NO_SLOT: the result is multiplied by mem[post_slot] (a plain residue), which also takes
// it out of Montgomery form; otherwise it is multiplied by the constant 1.  Result (lazy, < 2N) -> out.
void emit_modexp_shared(Prog& p, const BigU& e, uint32_t in_lo, uint32_t in_hi, uint32_t tmp, uint32_t out,
                        uint32_t tab, uint32_t post_slot, bool skip_zero_digits, bool raw = false) {
  const int w = 5;
  if (raw && (e.bit_length() < 64 || !skip_zero_digits)) api_throw(PGPU_ERR_INVALID, "raw ladder needs the sliding-window form");
  if (e.is_zero()) {  // gmp.Int.Exp: y <= 0 -> 1
    if (post_slot != NO_SLOT) { p.op(VM_LOAD, post_slot); } else { p.op(VM_LOADC, C_ONE); }
    p.op(VM_STORE, out);
    return;
  }
  if (raw) p.op(VM_LOAD, in_lo); else emit_to_mont(p, in_lo, in_hi, tmp);
  if (skip_zero_digits && e.bit_length() >= 64) {
    // Sliding window over odd powers: ~bits/(sw+1) products instead of bits/w, half the table.  The operation sequence
    // depends on the exponent's bits -- on the KEY, never on the ciphertexts: every lane of every batch under one key
    // runs the same program, so kernel time carries no per-ciphertext signal (what mpz_powm, the reference's own
    // backend, does too).  Table slot tab+k holds x^(2k+1); x^2 sits in `tmp` during the build (free after the entry).
    const int sw = e.bit_length() >= 700 ? 6 : 5;        // 2^(sw-1) + bits/(sw+1) products: 6 wins from ~700 bits
    const uint32_t nodd = 1u << (sw - 1);
    p.op(VM_STORE, tab + 0);          // x^1
    p.op(VM_SQR);
    p.op(VM_STORE, tmp);              // x^2
    p.op(VM_LOAD, tab + 0);
    for (uint32_t k = 1; k < nodd; ++k) {
      p.op(VM_MUL, tmp);
      p.op(VM_STORE, tab + k);        // x^(2k+1)
    }
    long i = (long)e.bit_length() - 1;
    bool first = true;
    while (i >= 0) {
      if (!e.bit((size_t)i)) { p.op(VM_SQR); --i; continue; }
      long l = std::max<long>(i - sw + 1, 0);
      while (!e.bit((size_t)l)) ++l;                    // window [l, i] ends in a one bit
      uint32_t val = 0;
      for (long b = i; b >= l; --b) val = (val << 1) | (uint32_t)e.bit((size_t)b);
      if (first) {
        p.op(VM_LOAD, tab + (val >> 1));
        first = false;
      } else {
        for (long b = i; b >= l; --b) p.op(VM_SQR);
        p.op(VM_MUL, tab + (val >> 1));
      }
      i = l - 1;
    }
    if (!raw) { if (post_slot != NO_SLOT) p.op(VM_MUL, post_slot); else p.op(VM_MULC, C_ONE); }
    p.op(VM_STORE, out);
    return;
  }
  const size_t ebits = e.bit_length();
  const size_t nwin = (ebits + w - 1) / w;
  auto digit = [&](size_t i_from_top) {
    size_t lo_bit = (nwin - 1 - i_from_top) * w;
    uint32_t d = 0;
    for (int b = 0; b < w; ++b) d |= (uint32_t)e.bit(lo_bit + b) << b;
    return d;
  };
  uint32_t maxd = 0;
  for (size_t i = 0; i < nwin; ++i) maxd = std::max(maxd, digit(i));
  p.op(VM_STORE, tab + 1);
  p.op(VM_LOADC, C_ONE_M);
  p.op(VM_STORE, tab + 0);
  p.op(VM_LOAD, tab + 1);
  for (uint32_t k = 2; k <= maxd; ++k) {
    p.op(VM_MUL, tab + 1);
    p.op(VM_STORE, tab + k);
  }
  p.op(VM_LOAD, tab + digit(0));
  for (size_t i = 1; i < nwin; ++i) {
    for (int s = 0; s < w; ++s) p.op(VM_SQR);
    uint32_t d = digit(i);
    if (d != 0 || !skip_zero_digits) p.op(VM_MUL, tab + d);
  }
  if (post_slot != NO_SLOT) p.op(VM_MUL, post_slot); else p.op(VM_MULC, C_ONE);
  p.op(VM_STORE, out);
}

// Per-number exponent, fixed window w = 4 (7 windows per 28-bit exponent limb).  The exponent limbs are
// the segment's `digits` array [we][nb].  Table slots tab .. tab+15.
// raw_one >= 0: the value in slot in_lo is already in the kernel's working form (pair kernels), raw_one is the constant
// holding 1 in that form; no entry, no exit.
// wb: window bits.  4: VM_MULV (7 windows per limb), table tab .. tab+15.  5: VM_MULV5 (the segment's `digits` must be the
// repacked 25-bit words: windows5_of()), 32 entries.  7: VM_MULV7 (4 windows per limb, no repacking), 128 entries.
// The wide windows pay where a product costs two squarings (the digit kernels): a 4 096-bit exponent takes 586 products
// and a 63 + 63 table at 7 bits, 820 + 30 at 5, 1 024 + 14 at 4.
static int perlane_windows(int we, int wb) { return wb == 4 ? we * 7 : wb == 5 ? (we * LB + 4) / 5 : we * 4; }
static VmOp perlane_op(int wb, bool nm4 = false) { return wb == 4 ? (nm4 ? VM_MULVT : VM_MULV) : wb == 5 ? (nm4 ? VM_MULVT5 : VM_MULV5) : VM_MULV7; }
// table of x^0 .. x^(2^wb - 1) from x in the accumulator; the wide tables square for their even entries (a squaring
// is half a product on the digit kernels).  The 7-bit table is gathered per number, so its 128 slots are number-major
// (VM_STORET / VM_MULV7); the entries the build itself reads back (x and the ones that get squared) are kept limb-major as
// well, in the 64 slots after the table.
static int perlane_table_slots(int wb, bool nm4 = false) { return wb == 7 ? 128 + 64 : nm4 ? (wb == 5 ? 32 + 16 : 16 + 1) : 1 << wb; }
// nm4 (4- / 5-bit windows on the pair kernels): the entries NUMBER-major (VM_STORET, gathered by VM_MULVT / VM_MULVT5) and what the
// build itself reads back limb-major behind them -- x in slot tab + 16 (4 bits), entries 1 .. 15 in slots tab + 32 + k (5 bits).
static void emit_power_table(Prog& p, uint32_t tab, uint32_t one, int wb, bool nm4 = false) {
  if (wb == 5 && nm4) {
    const uint32_t scr = tab + 32;
    p.op(VM_STORET, tab + 1);
    p.op(VM_STORE, scr + 1);
    p.op(VM_LOADC, one);
    p.op(VM_STORET, tab + 0);
    p.op(VM_LOAD, scr + 1);
    for (uint32_t k = 2; k < 32; ++k) {
      if (k % 2 == 0) { p.op(VM_LOAD, scr + k / 2); p.op(VM_SQR); }
      else p.op(VM_MUL, scr + 1);
      p.op(VM_STORET, tab + k);
      if (k < 16) p.op(VM_STORE, scr + k);
    }
    return;
  }
  if (wb == 4 && nm4) {
    const uint32_t scr = tab + 16;
    p.op(VM_STORET, tab + 1);
    p.op(VM_STORE, scr);
    p.op(VM_LOADC, one);
    p.op(VM_STORET, tab + 0);
    p.op(VM_LOAD, scr);
    for (uint32_t k = 2; k < 16; ++k) { p.op(VM_MUL, scr); p.op(VM_STORET, tab + k); }
    return;
  }
  if (wb == 7) {
    const uint32_t scr = tab + 128;
    p.op(VM_STORET, tab + 1);
    p.op(VM_STORE, scr + 1);
    p.op(VM_LOADC, one);
    p.op(VM_STORET, tab + 0);
    p.op(VM_LOAD, scr + 1);
    for (uint32_t k = 2; k < 128; ++k) {
      if (k % 2 == 0) { p.op(VM_LOAD, scr + k / 2); p.op(VM_SQR); }
      else p.op(VM_MUL, scr + 1);
      p.op(VM_STORET, tab + k);
      if (k < 64) p.op(VM_STORE, scr + k);
    }
    return;
  }
  p.op(VM_STORE, tab + 1);
  p.op(VM_LOADC, one);
  p.op(VM_STORE, tab + 0);
  p.op(VM_LOAD, tab + 1);
  for (uint32_t k = 2; k < (1u << wb); ++k) {
    if (wb >= 5 && k % 2 == 0) { p.op(VM_LOAD, tab + k / 2); p.op(VM_SQR); }
    else p.op(VM_MUL, tab + 1);
    p.op(VM_STORE, tab + k);
  }
}

void emit_modexp_perlane(Prog& p, int we, uint32_t in_lo, uint32_t in_hi, uint32_t tmp, uint32_t out, uint32_t tab,
                         uint32_t post_slot, int raw_one = -1, int wb = 4, bool nm4 = false) {
  const uint32_t one = raw_one >= 0 ? (uint32_t)raw_one : (uint32_t)C_ONE_M;
  if (raw_one >= 0) p.op(VM_LOAD, in_lo); else emit_to_mont(p, in_lo, in_hi, tmp);
  emit_power_table(p, tab, one, wb, nm4);
  const int nwin = perlane_windows(we, wb);
  p.op(VM_LOADC, one);
  for (int i = nwin - 1; i >= 0; --i) {
    if (i != nwin - 1) for (int s = 0; s < wb; ++s) p.op(VM_SQR);
    p.op(perlane_op(wb, nm4), (uint32_t)i, tab);
  }
  if (raw_one < 0) { if (post_slot != NO_SLOT) p.op(VM_MUL, post_slot); else p.op(VM_MULC, C_ONE); }
  p.op(VM_STORE, out);
}

// x^(per-number exponent) * y^(shared exponent e) with ONE chain of squarings (interleaved / "Shamir" exponentiation):
// x's fixed windows (wb bits: MULV / MULV5 / MULV7 on the table tab1[0 .. 2^wb - 1]) and y's sliding windows over odd powers
// (tab2[0 .. 2^(sw-1) - 1]; sw = 6, or 7 next to 7-bit windows) hang off the same accumulator.  Costs max(bits) squarings
// instead of the sum: what check^(E^n) * F^(n^2) of the DDLEQ verifier (ddleq.go:143-152) and alpha = ct1^(x^n) * y^(n^2)
// of the prover (ddleq.go:81-87) need.
// x in slot in1, y in slot in2 (plain residues); result (plain, lazy) -> out.
static int dual_sliding_bits(int wb) { return wb == 7 ? 7 : 6; }
void emit_modexp_dual(Prog& p, int we, const BigU& e, uint32_t in1, uint32_t in2, uint32_t tmp, uint32_t out, uint32_t tab1,
                      uint32_t tab2, int raw_one = -1, int wb = 4, bool nm4 = false) {
  // raw_one >= 0: in1 / in2 are already in the kernel's working form (digit kernels), raw_one = the constant holding 1 in
  // that form; no entry, no exit
  const uint32_t one_m = raw_one >= 0 ? (uint32_t)raw_one : (uint32_t)C_ONE_M;
  // tables
  if (raw_one >= 0) p.op(VM_LOAD, in1); else emit_to_mont(p, in1, NO_SLOT, tmp);
  emit_power_table(p, tab1, one_m, wb, nm4);
  const int sw = dual_sliding_bits(wb);
  const uint32_t nodd = 1u << (sw - 1);
  if (raw_one >= 0) p.op(VM_LOAD, in2); else emit_to_mont(p, in2, NO_SLOT, tmp);
  p.op(VM_STORE, tab2 + 0);
  p.op(VM_SQR);
  p.op(VM_STORE, tmp);
  p.op(VM_LOAD, tab2 + 0);
  for (uint32_t k = 1; k < nodd; ++k) { p.op(VM_MUL, tmp); p.op(VM_STORE, tab2 + k); }
  // sliding windows of e: mul_at[l] = table index to multiply in after the squaring of bit l (the window's lowest bit)
  const long nbits = std::max<long>((long)we * LB, (long)e.bit_length());
  std::vector<int> mul_at((size_t)nbits, -1);
  for (long i = (long)e.bit_length() - 1; i >= 0;) {
    if (!e.bit((size_t)i)) { --i; continue; }
    long l = std::max<long>(i - sw + 1, 0);
    while (!e.bit((size_t)l)) ++l;
    uint32_t val = 0;
    for (long b = i; b >= l; --b) val = (val << 1) | (uint32_t)e.bit((size_t)b);
    mul_at[(size_t)l] = (int)(val >> 1);
    i = l - 1;
  }
  const long nwin = perlane_windows(we, wb);
  p.op(VM_LOADC, one_m);
  for (long b = nbits - 1; b >= 0; --b) {
    if (b != nbits - 1) p.op(VM_SQR);
    if (b % wb == 0 && b / wb < nwin) p.op(perlane_op(wb, nm4), (uint32_t)(b / wb), tab1);
    if (mul_at[(size_t)b] >= 0) p.op(VM_MUL, tab2 + (uint32_t)mul_at[(size_t)b]);
  }
  if (raw_one < 0) p.op(VM_MULC, C_ONE);
  p.op(VM_STORE, out);
}

// The general interleaved ladder: x^(per-number exponent) * prod_k y_k^(e_k) with ONE chain of squarings -- one base with
// per-number windows (optional: we == 0 leaves it out) and any number of bases with shared exponents, each with its own
// sliding windows over odd powers.  All inputs already in the kernel's working form (digit kernels), `one` = the constant
// holding 1 in that form.  Sliding windows: 6 bits below 1 500 exponent bits, dual_sliding_bits(wb) above.
struct SharedBase { BigU e; uint32_t in; uint32_t tab; };
struct PerNumberBase { int we; uint32_t in; uint32_t tab; uint32_t first_window; };   // windows first_window.. of the `digits` rows
static int shared_window_bits(const BigU& e, int wb) { return e.bit_length() < 1500 ? 6 : dual_sliding_bits(wb); }
void emit_modexp_multi(Prog& p, const std::vector<PerNumberBase>& pn, int wb, const std::vector<SharedBase>& sh, uint32_t tmp,
                       uint32_t out, uint32_t one, bool nm4 = false) {
  long nbits = 0;
  for (auto& b : pn) {
    p.op(VM_LOAD, b.in);
    emit_power_table(p, b.tab, one, wb, nm4);
    nbits = std::max<long>(nbits, (long)b.we * LB);
  }
  std::vector<std::vector<int>> mul_at(sh.size());
  for (size_t k = 0; k < sh.size(); ++k) {
    const BigU& e = sh[k].e;
    const int sw = shared_window_bits(e, wb);
    const uint32_t nodd = 1u << (sw - 1);
    nbits = std::max<long>(nbits, (long)e.bit_length());
    p.op(VM_LOAD, sh[k].in);
    p.op(VM_STORE, sh[k].tab + 0);
    p.op(VM_SQR);
    p.op(VM_STORE, tmp);
    p.op(VM_LOAD, sh[k].tab + 0);
    for (uint32_t j = 1; j < nodd; ++j) { p.op(VM_MUL, tmp); p.op(VM_STORE, sh[k].tab + j); }
  }
  for (size_t k = 0; k < sh.size(); ++k) {
    const BigU& e = sh[k].e;
    const int sw = shared_window_bits(e, wb);
    mul_at[k].assign((size_t)std::max<long>(nbits, 1), -1);
    for (long i = (long)e.bit_length() - 1; i >= 0;) {
      if (!e.bit((size_t)i)) { --i; continue; }
      long l = std::max<long>(i - sw + 1, 0);
      while (!e.bit((size_t)l)) ++l;
      uint32_t val = 0;
      for (long b = i; b >= l; --b) val = (val << 1) | (uint32_t)e.bit((size_t)b);
      mul_at[k][(size_t)l] = (int)(val >> 1);
      i = l - 1;
    }
  }
  p.op(VM_LOADC, one);
  for (long b = nbits - 1; b >= 0; --b) {
    if (b != nbits - 1) p.op(VM_SQR);
    for (auto& q : pn)
      if (b % wb == 0 && b / wb < perlane_windows(q.we, wb)) p.op(perlane_op(wb, nm4), (uint32_t)(b / wb) + q.first_window, q.tab);
    for (size_t k = 0; k < sh.size(); ++k)
      if (mul_at[k][(size_t)b] >= 0) p.op(VM_MUL, sh[k].tab + (uint32_t)mul_at[k][(size_t)b]);
  }
  p.op(VM_STORE, out);
}

// x^(e_0), x^(e_1), ... for SEVERAL shared exponents on ONE base with a single chain of squarings (right-to-left sliding
// windows, Yao's buckets): the chain x, x^2, x^4, ... is walked once; where a window of e_s starts (a one bit at position j,
// value d = bits [j, j + w), odd), the current power x^(2^j) is multiplied into bucket B_s[(d-1)/2]; at the end
//   x^(e_s) = prod_d B_s[d]^d = R_0 * (R_1 R_2 ... R_(K-1))^2,    R_k = prod_(i >= k) B_s[i]   (d = 2k + 1)
// -- bits/(w+1) + 2 * 2^(w-1) products per exponent beside ONE chain of `bits` squarings, where separate left-to-right
// ladders square `bits` times EACH (three servers' PartialDecrypt of the same ciphertexts: 50 % of the multiplies).  The
// operation sequence depends on the exponents (the key's shares), never on the bases.
// The base is in slot `in` in the kernel's working form (pair digits); one_const = 1 in that form.  Slots: bp, run, acc
// (scratch), out0 + s (results, lazy), bucket0 + s * 2^(w-1) + k.
void emit_multi_exp_shared_base(Prog& p, const std::vector<BigU>& es, uint32_t in, uint32_t bp, uint32_t run, uint32_t acc,
                                uint32_t out0, uint32_t bucket0, int w, uint32_t one_const, bool muls = false) {
  // muls: the kernel has VM_MULS (bucket <- bucket * x with x left in the registers): a bucket product is one load and one
  // store of the bucket -- no parking of the current power in `bp`, no reloading it afterwards
  const uint32_t K = 1u << (w - 1);
  const size_t S = es.size();
  size_t nbits = 0;
  for (auto& e : es) nbits = std::max(nbits, e.bit_length());
  // window starts: at[j] = list of (server, bucket)
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> at(nbits + 1);
  size_t last_start = 0;
  for (size_t s = 0; s < S; ++s) {
    const BigU& e = es[s];
    for (size_t j = 0; j < e.bit_length();) {
      if (!e.bit(j)) { ++j; continue; }
      uint32_t d = 0;
      for (int b = 0; b < w; ++b) if (j + b < e.bit_length()) d |= (uint32_t)e.bit(j + b) << b;
      at[j].push_back({(uint32_t)s, d >> 1});
      last_start = std::max(last_start, j);
      j += w;
    }
  }
  std::vector<char> touched(S * K, 0);
  auto bucket = [&](uint32_t s, uint32_t k) { return bucket0 + s * K + k; };
  p.op(VM_LOAD, in);
  for (size_t j = 0; j <= last_start && nbits; ++j) {
    if (j) p.op(VM_SQR);                                   // x^(2^j)
    bool stored = false, dirty = false;
    for (auto& sk : at[j]) {
      const uint32_t b = bucket(sk.first, sk.second);
      if (muls) {
        p.op(touched[sk.first * K + sk.second] ? VM_MULS : VM_STORE, b);
        touched[sk.first * K + sk.second] = 1;
        continue;
      }
      if (!touched[sk.first * K + sk.second]) {
        if (dirty) { p.op(VM_LOAD, bp); dirty = false; }
        p.op(VM_STORE, b);
        touched[sk.first * K + sk.second] = 1;
      } else {
        if (dirty) { p.op(VM_LOAD, bp); dirty = false; }
        if (!stored) { p.op(VM_STORE, bp); stored = true; }
        p.op(VM_MUL, b);
        p.op(VM_STORE, b);
        dirty = true;
      }
    }
    if (dirty && j < last_start) p.op(VM_LOAD, bp);
  }
  for (size_t s = 0; s < S; ++s) {
    bool run_empty = true, acc_empty = true;
    for (uint32_t k = K - 1; k >= 1; --k) {
      bool x_is_run = false;
      if (touched[s * K + k]) {
        if (run_empty) { p.op(VM_LOAD, bucket((uint32_t)s, k)); run_empty = false; }
        else { p.op(VM_LOAD, run); p.op(VM_MUL, bucket((uint32_t)s, k)); }
        p.op(VM_STORE, run);
        x_is_run = true;
      }
      if (!run_empty) {
        if (acc_empty) { if (!x_is_run) p.op(VM_LOAD, run); acc_empty = false; }
        else { p.op(VM_LOAD, acc); p.op(VM_MUL, run); }
        p.op(VM_STORE, acc);
      }
    }
    // R_0 -> its slot (run, or the bucket itself), then acc^2 * R_0
    uint32_t r0 = NO_SLOT;
    if (touched[s * K]) {
      if (run_empty) r0 = bucket((uint32_t)s, 0);
      else { p.op(VM_LOAD, run); p.op(VM_MUL, bucket((uint32_t)s, 0)); p.op(VM_STORE, run); r0 = run; }
    } else if (!run_empty) r0 = run;
    if (!acc_empty) {
      p.op(VM_LOAD, acc);
      p.op(VM_SQR);
      p.op(VM_MUL, r0);            // (acc non-empty implies run non-empty, so R_0 exists)
    } else if (r0 != NO_SLOT) p.op(VM_LOAD, r0);
    else p.op(VM_LOADC, one_const);                        // e_s = 0: gmp.Int.Exp gives 1
    p.op(VM_STORE, out0 + (uint32_t)s);
  }
}

struct SegSpec {
  const ModCtx* mc;
  const Prog* prog;
  uint32_t* mem;
  const uint32_t* digits;
  // pair kernel (N = p^2 as two base-p digits, gen_vm_asm.py GenP): `pair` = device array p | Cadj (H limbs each),
  // pair_n0inv = -p^-1 mod 2^28, pair_h = H.  mc stays the 2H-limb modulus p^2 (slot size, accounting).
  const uint32_t* pair = nullptr;
  uint32_t pair_n0inv = 0;
  int pair_h = 0;
  int pair_lanes = 1;   // 1: GenP (one lane holds both digits); 2: GenQ (one digit per lane); 4: GenQ4 (two lanes per digit);
                        // 3: GenQ3 (three digits modulo n^3 in a quad; `tconsts` = its constant table, slots are 3H limbs)
  const uint32_t* tconsts = nullptr;
};

// launch one VM kernel with 1 to 3 segments of `nb` numbers each (same modulus shape; s2 only together with s1)
void run_vm(pgpu_ctx* ctx, size_t nb, const SegSpec& s0, const SegSpec* s1, bool profile, size_t launch_nb = 0,
            const SegSpec* s2 = nullptr) {
  const ModCtx* mc = s0.mc;
  if (s1 && (s1->mc->WL != mc->WL || s1->mc->K != mc->K)) api_throw(PGPU_ERR_INVALID, "segment shape mismatch");
  if (s2 && (!s1 || s2->mc->WL != mc->WL || s2->mc->K != mc->K)) api_throw(PGPU_ERR_INVALID, "segment shape mismatch");
  VmArgs a;
  memset(&a, 0, sizeof a);
  const SegSpec* ss[3] = {&s0, s1, s2};
  double montmuls = 0, sqrs = 0;
  for (int i = 0; i < 3; ++i) {
    if (!ss[i]) continue;
    VmSeg& g = a.seg[i];
    g.prog = ctx->upload_words(ss[i]->prog->w);
    g.nmod = ss[i]->pair ? const_cast<uint32_t*>(ss[i]->pair) : ss[i]->mc->d_nmod;
    g.consts = ss[i]->tconsts ? const_cast<uint32_t*>(ss[i]->tconsts) : ss[i]->mc->d_consts;
    g.mem = ss[i]->mem;
    g.digits = ss[i]->digits;
    g.n0inv = ss[i]->pair ? ss[i]->pair_n0inv : ss[i]->mc->n0inv;
    g.nb = (uint32_t)nb;
    montmuls += ss[i]->prog->montmuls;
    sqrs += ss[i]->prog->sqrs;
  }
  if (launch_nb == 0) launch_nb = nb;  // numbers actually launched (<= nb, the row stride of the arrays)
  // Occupancy-aware shape: the same WT limbs can be sliced over more lanes (WL/2 x 2K).  One lane per slice keeps the
  // multiply count but fills the chip when the batch is small.  The natural shape has the cheapest squarings (K == 1:
  // triangular rows; the wave-sliced 2-slice kernels: every limb product once), and a single wave per SIMD already
  // issues at ~88 % of the two-wave rate, so it wins from one wave per SIMD (1024 SIMDs x 64 lanes) upwards; below that
  // the finer slicing wins (tools/occupancy_sweep.py: Decrypt-2048 at 32768: 1.24 M/s natural vs 1.01 M/s re-sliced; at
  // 16384: 0.63 vs 0.88 M/s; Encrypt-2048 at 32768: 295 k vs 271 k; at 16384: 148 k vs 251 k).
  int WL = mc->WL, K = mc->K;
  const bool pair = s0.pair != nullptr;
  if (s1 && pair != (s1->pair != nullptr)) api_throw(PGPU_ERR_INVALID, "segment kind mismatch");
  if (s2 && pair != (s2->pair != nullptr)) api_throw(PGPU_ERR_INVALID, "segment kind mismatch");
  if (pair) {
    WL = s0.pair_lanes == 8 ? s0.pair_h / 4 : (s0.pair_lanes == 4 || s0.pair_lanes == 6) ? s0.pair_h / 2 : s0.pair_h;
    K = s0.pair_lanes == 8 ? 96 : s0.pair_lanes == 6 ? 112 : s0.pair_lanes == 4 ? 64 : s0.pair_lanes == 3 ? 48 : s0.pair_lanes == 2 ? 32 : 16;   // tags of the pair kernels, not lane counts
  } else {
    static const size_t lanes_env = [] { const char* e = getenv("PGPU_LANES_WANTED"); return e ? (size_t)atoll(e) : (size_t)0; }();
    const size_t lanes_wanted = ctx->lanes_wanted ? ctx->lanes_wanted : lanes_env ? lanes_env : (size_t)1024 * 64;
    const size_t segs = s2 ? 3 : s1 ? 2 : 1;
    while (launch_nb * K * segs < lanes_wanted && K < 4 && WL % 2 == 0 && WL / 2 >= 37) { WL /= 2; K *= 2; }
    // PGPU_W74=0 (experiments): the 4-lane slicing instead of the wave-sliced 148-limb kernel
    static const bool w74 = [] { const char* e = getenv("PGPU_W74"); return e ? atoi(e) != 0 : true; }();
    if (WL == 74 && K == 2 && !(w74 && ctx->use_asm)) { WL = 37; K = 4; }
  }
  const uint32_t blocks_per_seg = (uint32_t)(launch_nb * (pair ? (s0.pair_lanes == 3 ? 4 : s0.pair_lanes == 6 ? 8 : s0.pair_lanes) : K) / VM_BLOCK);
  a.seg0_blocks = blocks_per_seg;
  a.seg1_blocks = blocks_per_seg;
  const uint32_t blocks = blocks_per_seg * (s2 ? 3 : s1 ? 2 : 1);
  const bool nm_tables = s0.prog->nm_tables || (s1 && s1->prog->nm_tables) || (s2 && s2->prog->nm_tables);
  const bool nm4 = s0.prog->nm4 || (s1 && s1->prog->nm4) || (s2 && s2->prog->nm4);
  const bool mulv7 = s0.prog->mulv7 || (s1 && s1->prog->mulv7) || (s2 && s2->prog->mulv7);
  if (nm4 && mulv7) api_throw(PGPU_ERR_UNSUPPORTED, "internal: 4-bit and 7-bit number-major tables in one launch");
  // VM_STORET / VM_MULVT / VM_MULVT5: GenP for 37-limb primes, GenQ (two lanes), GenQ4 (four lanes)
  // (the three-digit kernel: VM_MULVT5 beside its VM_MULV7; the host never sends it 4-bit windows)
  const bool nm4_kernel = pair && ((s0.pair_lanes == 1 && s0.pair_h == 37) || s0.pair_lanes == 2 || s0.pair_lanes == 4 || s0.pair_lanes == 3);
  const bool use_asm = ctx->use_asm && vm_asm_available(WL, K) && s0.prog->asm_ok && (!s1 || s1->prog->asm_ok) &&
                       (!s2 || s2->prog->asm_ok) && (!nm_tables || (pair && (mulv7 ? s0.pair_lanes == 3 : nm4 ? nm4_kernel : (s0.pair_lanes == 3 || nm4_kernel)))) &&
                       (uint64_t)nb * ((s0.pair_lanes == 3 || s0.pair_lanes == 6) ? 3 * s0.pair_h : mc->WT) * 4 *
                               std::max(s0.prog->gather_slots, std::max(s1 ? s1->prog->gather_slots : 1u, s2 ? s2->prog->gather_slots : 1u)) < (1ull << 32);
  pgpu_ctx::Ev* ev = nullptr;
  if (profile) {
    ev = &ctx->next_ev();
    // v_mad_u64_u32 lane-ops executed: 2 WT^2 per product; the assembly kernel's K == 1 squaring rows use the
    // symmetry of the square: WL^2 (reduction) + WL(WL-1)/2 + WL (product)
    const double full = 2.0 * mc->WT * mc->WT;
    // squaring rows: K == 1 triangular (WT^2 + WT(WT-1)/2 + WT); K == 2 slice-level symmetry (product part 1.5 WL^2 per lane)
    double sq = full;
    double mulp = full;
    if (pair && (s0.pair_lanes == 3 || s0.pair_lanes == 6)) {   // GenQ3 / GenQ6: a squaring is one pass, a product 6 blocks (its second pass runs half the lanes)
      const double H = s0.pair_h;
      mulp = 12.0 * H * H;
      sq = 8.0 * H * H;
    } else if (pair && (s0.pair_lanes == 4 || s0.pair_lanes == 8)) {   // GenQ4 / GenQ8: a squaring is one pass of H rows of 2 * H/2 multiplies in four lanes; a
      const double H = s0.pair_h;              // product one pass with two multiplier streams (3 * H/2 multiplies a row)
      mulp = 6.0 * H * H;
      sq = 4.0 * H * H;
    } else if (pair && s0.pair_lanes >= 2) {   // GenQ: one / two Montgomery passes modulo n in every digit lane
      const double H = s0.pair_h;
      mulp = 8.0 * H * H;
      sq = 4.0 * H * H;
    } else if (pair) {   // five / three-and-a-half half-width products (see GenP)
      const double H = WL;
      mulp = 5.0 * H * H;
      sq = 3.0 * H * H + 0.5 * H * (H - 1) + H;
    } else
    if (use_asm && K == 1) sq = (double)mc->WT * mc->WT + 0.5 * mc->WT * (mc->WT - 1) + mc->WT;
    else if (use_asm && K == 2 && WL >= 55) sq = 2.0 * WL * WL + WL + 2.0 * mc->WT * WL;   // wave-sliced: every product once
    else if (use_asm && K == 2) sq = (double)mc->WT * mc->WT * (2.0 - 1.0 / (2 * K)) + WL;
    ev->mads = ((montmuls - sqrs) * mulp + sqrs * sq) * (double)launch_nb;
    HIPCHK(hipEventRecord(ev->a, ctx->stream));
  }
  if (pair && !use_asm) api_throw(PGPU_ERR_UNSUPPORTED, "the pair kernel exists in assembly only");
  for (int i = 0; i < 3; ++i)
    if (ss[i] && ss[i]->prog->needs_muls && !(pair && (s0.pair_lanes == 4 || s0.pair_lanes == 8)))
      api_throw(PGPU_ERR_UNSUPPORTED, "internal: VM_MULS on a kernel that does not implement it");
  if (pair && s0.pair_lanes == 1 && s0.pair_h > 37)
    for (int i = 0; i < 3; ++i)
      if (ss[i] && ss[i]->prog->wide_gathers)   // (an opcode a kernel does not know ends its program: refuse, never compute garbage)
        api_throw(PGPU_ERR_UNSUPPORTED, "internal: the one-lane pair kernel for 55-limb primes has 4-bit per-number windows only");
  if (ev) snprintf(ev->name, sizeof ev->name, use_asm ? "vm_asm_%d_%d" : "vm_kernel<%d,%d>", WL, K);
  hipError_t e = use_asm ? launch_vm_asm(WL, K, a, blocks, ctx->stream) : launch_vm(WL, K, a, blocks, ctx->stream);
  if (use_asm) ctx->last_vm_asm++;
  ctx->last_vm_launches++;
  if (e != hipSuccess) throw HipError{e, "launch_vm"};
  if (profile) HIPCHK(hipEventRecord(ev->b, ctx->stream));
}

// stage an operand buffer (host or device, big-endian element-major) and unpack it to `wt` limbs
void unpack_operand(pgpu_ctx* ctx, const uint8_t* buf, size_t stride, size_t nbytes, size_t count, int mem,
                    uint32_t* out, int wt, size_t nb) {
  if (nbytes > stride) api_throw(PGPU_ERR_INVALID, "operand length exceeds its stride");
  if (nbytes * 8 > (size_t)LB * wt + 7) api_throw(PGPU_ERR_INVALID, "operand wider than the modulus supports");
  const uint8_t* d = buf;
  if (mem == PGPU_MEM_HOST) {
    uint8_t* st = (uint8_t*)ctx->ws(stride * count);
    HIPCHK(hipMemcpyAsync(st, buf, stride * count, hipMemcpyHostToDevice, ctx->stream));
    d = st;
  }
  // operands are right-aligned in their stride: the value is the last nbytes of each element
  launch_unpack_be(d + (stride - nbytes), stride, nbytes, count, out, wt, nb, ctx->stream);
}

void pack_result(pgpu_ctx* ctx, const uint32_t* in, int wt, size_t nb, size_t count, uint8_t* out, size_t stride,
                 size_t nbytes, int mem) {
  if (nbytes > stride) api_throw(PGPU_ERR_INVALID, "result length exceeds its stride");
  if (mem == PGPU_MEM_HOST) {
    uint8_t* st = (uint8_t*)ctx->ws(stride * count);
    HIPCHK(hipMemsetAsync(st, 0, stride * count, ctx->stream));
    launch_pack_be(in, wt, nb, count, st + (stride - nbytes), stride, nbytes, ctx->stream);
    HIPCHK(hipMemcpyAsync(out, st, stride * count, hipMemcpyDeviceToHost, ctx->stream));
  } else {
    if (stride != nbytes) HIPCHK(hipMemsetAsync(out, 0, stride * count, ctx->stream));
    launch_pack_be(in, wt, nb, count, out + (stride - nbytes), stride, nbytes, ctx->stream);
  }
}

template <class F> int guarded(F&& f) {
  try {
    f();
    return PGPU_OK;
  } catch (const HipError& e) {
    return fail(PGPU_ERR_HIP, "HIP error: %s (%s)", hipGetErrorString(e.e), e.what);
  } catch (const ApiError& e) {
    return fail(e.code, "%s", e.msg.c_str());
  } catch (const std::exception& e) {
    return fail(PGPU_ERR_INVALID, "%s", e.what());
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Opaque handle types
// ------------------------------------------------------------------------------------------------
struct pgpu_modulus {
  pgpu_ctx* ctx;
  ModCtx mc;
};

struct pgpu_pubkey {
  pgpu_ctx* ctx;
  BigU N, G, H, Kk;
  bool g_is_n_plus_1;
  ModCtx mn, mn2;                 // moduli n, n^2  (level one)
  std::unique_ptr<ModCtx> mn3;    // n^3 (level two), built when the width is supported
  DevLimbs n_limbs;               // n as mn.WT limbs (multiplicand of the closed-form g^m)
  DevLimbs ninv2k;                // n^-1 mod 2^(28 mn.WT): exact division by n (the L function)
  DevLimbs ninv2k_2;              // n^-1 mod 2^(28 mn2.WT): quotients up to n^2 (level two)
  DevLimbs n2_limbs;              // n^2 as mn2.WT limbs
  int c_inv2R = -1;               // 2^-1 * R mod n in mn.consts        (binomial of the level-two g^m)
  int c_ninv2R_2 = -1;            // n * 2^-1 * R mod n^2 in mn2.consts (Damgard-Jurik recovery, paillier.go:326-331)
  struct AltTab { bool built = false; int base = 0; int nwin = 0; size_t kbits = 0; } alt[2];  // fixed-base comb tables of h_s
  struct FixedBase { BigU base; int idx; int nwin; };
  std::vector<FixedBase> fixed_bases;   // comb tables of other fixed bases mod n^2 (verification keys)
  DevLimbs pairn_consts;          // n | Cadj | pad for the two-lane pair kernel (mn2.pairn points here)
  DevLimbs pairn_consts8, pairn_tconsts8;   // the same for the eight-lane pair kernel (76-limb digits) and its three constants
  DevLimbs triple_kconsts;        // n | (C1_i, C2_i) pairs | pad for the three-digit kernel (mn3->triple points here)
  DevLimbs triple_tconsts;        // its constants in digit form
  std::vector<std::pair<int, int>> combine_consts;  // (total servers l, index of (4 (l!)^2)^-1 * R mod n in mn.consts)
};

// order of the unit group modulo pr^3, split as 2^t m for the Montgomery reduction modulo its odd part
struct ExpOrder {
  bool ok = false;
  BigU ord;
  ModCtx modd;           // m
  int t = 0;
  uint32_t minv = 0;     // m^-1 mod 2^t
  DevLimbs m_limbs;      // m as w limbs
  int w = 0;             // limbs of a reduced exponent (< 2 ord)
  void init(pgpu_ctx* ctx, const BigU& pr) {
    ord = pr * pr * (pr - BigU(1));
    t = 0;
    while (!ord.bit((size_t)t)) ++t;
    if (t > 20) return;                                  // (k m must stay below 2^48 per limb in the lift kernel)
    const BigU m = hostbig::shr(ord, (size_t)t);
    modd.init(ctx, m);
    modd.upload();
    uint32_t m0 = m.d[0], x = m0;
    for (int i = 0; i < 6; ++i) x *= 2u - m0 * x;
    minv = x & ((1u << t) - 1u);
    w = (int)((ord.bit_length() + 1 + LB - 1) / LB);
    m_limbs.set(m, w);
    ok = true;
  }
};

struct pgpu_seckey {
  pgpu_ctx* ctx;
  const pgpu_pubkey* pk;
  BigU lambda;
  bool has_crt = false;
  BigU p, q;
  ModCtx mp, mq, mp2, mq2;       // moduli p, q, p^2, q^2
  int c_hpR = -1, c_hqR = -1;    // constants: hp*R mod p in mp, hq*R mod q in mq
  int c_pinvR = -1;              // p^-1 * R mod q in mq
  DevLimbs pinv2k, qinv2k;       // p^-1 mod 2^(28 mp.WT), q^-1 mod 2^(28 mq.WT)
  DevLimbs p_limbs;              // p as mp.WT limbs
  // generic path (reference formula)
  ModCtx smn, smn2;              // the key's own constant tables for n and n^2: secret-derived constants never enter the
                                 // (shared, longer-lived) public key's tables
  int c_muR = -1;                // lambda^-1 mod n, times R mod n, in smn
  DevLimbs n_minus_mu;           // (n - mu) mod n, the answer for c == 0 (L(-1) = -1)
  int c_mu2R = -1;               // lambda^-1 mod n^2, times R mod n^2, in smn2 (level two)
  ~pgpu_seckey() { wipe_vec(lambda.d); wipe_vec(p.d); wipe_vec(q.d); }
  // pair kernel for the p^2 / q^2 ladders (GenP): p | Cadj limb arrays, R_H mod p^2 as a plain constant of mp2 / mq2
  bool has_pair = false;
  bool pair_small2 = false;        // 37-limb primes: below one wave per SIMD the two-lane kernel fills the chip better
  int pair_lanes = 1;              // 1: GenP (both digits in one lane, 37-limb primes); 2: GenQ (one digit per lane: 55 / 74 limbs)
  DevLimbs pair_p, pair_q;
  int c_rh_p2 = -1, c_rh_q2 = -1;
  int c_pk_p2[4] = {-1, -1, -1, -1}, c_pk_q2[4] = {-1, -1, -1, -1};   // pair forms of R_H^(k+2): chunk k of c enters the ladder
  int c_onep_p2 = -1, c_onep_q2 = -1;                                 // pair form of 1 (normalises a lazy pair)
  DevLimbs q_limbs1;               // q as mq.WT limbs (p_limbs is above)
  // level-two CRT over p^3 and q^3
  bool has_crt2 = false;
  ModCtx mp3, mq3;
  int c_qinv_p = -1, c_pinv_q = -1;      // q^-1 mod p in mp, p^-1 mod q in mq, stored plain (x*R (x) c = x*c)
  int c_inv2R_p = -1, c_inv2R_q = -1;    // 2^-1 * R
  int c_q2R = -1, c_p2R = -1;            // q^2 * R mod p^2 in mp2, p^2 * R mod q^2 in mq2
  int c_hp2R = -1, c_hq2R = -1;          // (q (p-1))^-1 * R mod p^2 in mp2, (p (q-1))^-1 * R mod q^2 in mq2
  int c_p2invR = -1;                     // (p^2)^-1 * R mod q^2 in mq2
  int c_p3invR = -1;                     // (p^3)^-1 * R mod q^3 in mq3 (Garner for exponentiations modulo n^3)
  DevLimbs p3_limbs;                     // p^3 as mp3.WT limbs
  DevLimbs pinv2k_2, qinv2k_2;           // p^-1 mod 2^(28 mp2.WT), q^-1 mod 2^(28 mq2.WT)
  DevLimbs q_limbs, p2_limbs;            // q as mq.WT limbs, p^2 as mp2.WT limbs
  DevLimbs q2_limbs;                     // q^2 as mq2.WT limbs
  DevLimbs tkc_p, ttc_p, tkc_q, ttc_q;   // three-digit kernel constants for the ladders modulo p^3 and q^3 (mp3 / mq3 .triple)
  ExpOrder eo_p, eo_q;                   // exponent reduction modulo the orders of the units modulo p^3 / q^3
};

// Exponents of ladders modulo pr^3 can be taken modulo the order of its unit group, ord = pr^2 (pr - 1) = 2^t m: a quarter
// shorter than the exponents modulo n^2 the DDLEQ prover raises to (the holder of the factorisation only).
static BigU order_fixup(const BigU& e, const BigU& ord) {
  if (e < ord) return e;
  BigU r = e % ord;
  if (r < BigU(3)) r = r + ord;      // x^e = 0 for a non-unit x and e >= 3: keep the reduced exponent >= 3 as well
  return r;
}

// Constants of the pair kernel for a prime of H limbs: its limbs, then Cadj -- the multiple of the prime whose limbs
// 0..H-1 can all be taken from [2^28, 2^29), so that Cadj - m is limb-wise non-negative for every quotient m < 2^(28 H).
static std::vector<uint32_t> make_pair_consts(const BigU& pr, int H) {
  BigU D;
  for (int j = 0; j < H; ++j) D = D + hostbig::shl(BigU(1), (size_t)LB * j + LB);
  BigU kq, kr;
  hostbig::divmod(D, pr, kq, kr);
  BigU E = (kq + BigU(1)) * pr - D;          // 0 < E <= prime < 2^(28 H)
  std::vector<uint32_t> v = pr.to_limbs(LB, H), el = E.to_limbs(LB, H);
  for (int j = 0; j < H; ++j) v.push_back(el[j] + (1u << LB));
  return v;
}

// Constants of the three-digit kernel for the root n (H limbs): n padded to an even number of words, then the pairs
// (C1_i, C2_i), then two words of padding (the kernel prefetches one pair past the end).  C1 = k1 n is make_pair_consts'
// Cadj (every limb in [2^28, 2^29)); C2 = -k1 (mod n) shifted into the same limb range: the -C1 n that the first link
// leaves in digit one is -k1 n^2, which C2 cancels in digit two.
static std::vector<uint32_t> make_triple_kconsts(const BigU& n, int H) {
  const int npad = (H + 1) / 2 * 2;
  std::vector<uint32_t> pc = make_pair_consts(n, H);                 // n | C1
  BigU D;
  for (int j = 0; j < H; ++j) D = D + hostbig::shl(BigU(1), (size_t)LB * j + LB);
  BigU c1 = BigU::from_limbs(pc.data() + H, LB, (size_t)H), k1, rem;
  c1.trim();
  hostbig::divmod(c1, n, k1, rem);
  if (!rem.is_zero()) api_throw(PGPU_ERR_INVALID, "internal: C1 is not a multiple of n");
  const BigU e2 = (n - ((k1 + D) % n)) % n;                            // C2 = D + e2 = -k1 (mod n)
  const std::vector<uint32_t> e2l = e2.to_limbs(LB, (size_t)H);
  std::vector<uint32_t> kc((size_t)npad + 2 * H + 2, 0);
  for (int j = 0; j < H; ++j) {
    kc[j] = pc[j];
    kc[(size_t)npad + 2 * j] = pc[H + j];
    kc[(size_t)npad + 2 * j + 1] = e2l[j] + (1u << LB);
  }
  return kc;
}

// Attach the three-digit form to the modulus m3 = root^3: kernel constants, the digit form of 1, entry / exit constants.
// kc / tc own the device arrays; the exact-division inverses and the limb arrays of root and root^2 belong to the caller.
static void setup_triple(ModCtx& m3, const ModCtx& root, const ModCtx& mid, DevLimbs& kc_dev, DevLimbs& tc_dev, const uint32_t* dinv1,
                         const uint32_t* dinv2, const uint32_t* n_limbs, const uint32_t* n2_limbs) {
  const int H = root.WT;
  const BigU& n = root.N;
  const BigU& n3 = m3.N;
  const std::vector<uint32_t> kc = make_triple_kconsts(n, H);
  kc_dev.w = (int)kc.size();
  HIPCHK(hipMalloc((void**)&kc_dev.d, kc.size() * 4));
  HIPCHK(hipMemcpy(kc_dev.d, kc.data(), kc.size() * 4, hipMemcpyHostToDevice));
  // digit form of 1: the digits of R_H mod root^3
  const BigU RH = hostbig::shl(BigU(1), (size_t)LB * H);
  BigU rh = RH % n3, q1, d0, d2, d1;
  hostbig::divmod(rh, n, q1, d0);
  hostbig::divmod(q1, n, d2, d1);
  std::vector<uint32_t> tc;
  for (const BigU* dg : {&d0, &d1, &d2}) {
    auto l = dg->to_limbs(LB, (size_t)H);
    tc.insert(tc.end(), l.begin(), l.end());
  }
  tc_dev.w = (int)tc.size();
  HIPCHK(hipMalloc((void**)&tc_dev.d, tc.size() * 4));
  HIPCHK(hipMemcpy(tc_dev.d, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
  TripleInfo& ti = m3.triple;
  ti.mid = &mid;
  ti.kconsts = kc_dev.d;
  ti.tconsts = tc_dev.d;
  ti.c_rh = m3.add_const(rh);
  BigU rhinv;
  if (!hostbig::modinv(rh, n3, rhinv)) api_throw(PGPU_ERR_INVALID, "internal: R_H is not invertible modulo the cube");
  ti.c_exit = m3.add_const(hostbig::mulmod(m3.R % n3, rhinv, n3));
  ti.dinv1 = dinv1;
  ti.dinv2 = dinv2;
  ti.n_limbs = n_limbs;
  ti.n2_limbs = n2_limbs;
  ti.root = &root;
  m3.upload();
}

// inverse of odd d modulo 2^bits
static BigU inv_mod_pow2(const BigU& d, size_t bits) {
  BigU m = hostbig::shl(BigU(1), bits), out;
  if (!hostbig::modinv(d, m, out)) api_throw(PGPU_ERR_INVALID, "inverse mod 2^k of an even number");
  return out;
}

// Contexts that are alive: a key handle may outlive its context (garbage-collected bindings destroy in any order), so
// nothing dereferences a key's context pointer on the way out without looking here first.
static std::mutex g_live_mu;
static std::set<pgpu_ctx*> g_live_ctx;
static bool ctx_alive(pgpu_ctx* c) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  return g_live_ctx.count(c) != 0;
}

extern "C" {

const char* pgpu_last_error(void) { return g_err.c_str(); }
const char* pgpu_version(void) { return "paillier_hip 0.1 (gfx950, radix-2^28 Montgomery VM)"; }

int pgpu_ctx_create(int device, void* stream, pgpu_ctx** out) {
  if (!out) return fail(PGPU_ERR_INVALID, "null out");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PGPU_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= n) return fail(PGPU_ERR_INVALID, "device %d out of range (have %d)", device, n);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(PGPU_ERR_HIP, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(PGPU_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
  pgpu_ctx* c = new pgpu_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  int rc = guarded([&] {
    c->bind();
    if (stream == PGPU_STREAM_NEW) {
      c->stream = nullptr;
      HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      c->own_stream = true;
    }
  });
  if (rc != PGPU_OK) { delete c; return rc; }
  { std::lock_guard<std::mutex> lk(g_live_mu); g_live_ctx.insert(c); }
  *out = c;
  return PGPU_OK;
}

void pgpu_ctx_destroy(pgpu_ctx* ctx) {
  if (!ctx) return;
  { std::lock_guard<std::mutex> lk(g_live_mu); g_live_ctx.erase(ctx); }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  delete ctx;
}

int pgpu_ctx_set_flag(pgpu_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return fail(PGPU_ERR_INVALID, "null argument");
  if (strcmp(name, "asm") == 0) { ctx->use_asm = value != 0; return PGPU_OK; }
  if (strcmp(name, "pair") == 0) { ctx->use_pair = value != 0; return PGPU_OK; }
  if (strcmp(name, "triple") == 0) { ctx->use_triple = value != 0; return PGPU_OK; }
  if (strcmp(name, "shared_chain") == 0) { ctx->use_shared_chain = value != 0; return PGPU_OK; }
  if (strcmp(name, "lift") == 0) { ctx->use_lift = value != 0; return PGPU_OK; }
  if (strcmp(name, "side") == 0) { ctx->use_side = value != 0; return PGPU_OK; }
  if (strcmp(name, "lanes8") == 0) { ctx->use_lanes8 = value != 0; return PGPU_OK; }
  if (strcmp(name, "muls") == 0) { ctx->use_muls = value != 0; return PGPU_OK; }
  if (strcmp(name, "nm4") == 0) { ctx->use_nm4 = value != 0; return PGPU_OK; }
  if (strcmp(name, "early") == 0) { ctx->use_early = value != 0; return PGPU_OK; }
  if (strcmp(name, "handover") == 0) { ctx->use_handover = value != 0; return PGPU_OK; }
  if (strcmp(name, "fair") == 0) { g_wave_priorities.store(value != 0, std::memory_order_relaxed); return PGPU_OK; }   // process-wide (see above)
  if (strcmp(name, "lanes_wanted") == 0) { ctx->lanes_wanted = value > 0 ? (size_t)value : 0; return PGPU_OK; }
  if (strcmp(name, "cu_partition") == 0) {
    // value = (parts << 16) | part: confine this context's (own) stream to the part-th of `parts` equal slices of the
    // device's compute units.  Kernels of concurrently running contexts otherwise pile up on the same CUs (the dispatcher
    // starts every kernel's workgroups from the same place) and slow each other down instead of using the idle ones.
    if (!ctx->own_stream) return fail(PGPU_ERR_INVALID, "cu_partition needs a context created with PGPU_STREAM_NEW");
    const int parts = value >> 16, part = value & 0xffff;
    if (parts < 1 || part >= parts) return fail(PGPU_ERR_INVALID, "cu_partition: part out of range");
    return guarded([&] {
      ctx->bind();
      hipDeviceProp_t prop;
      HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
      const int ncu = prop.multiProcessorCount;
      std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0);
      const int lo = (int)((long)ncu * part / parts), hi = (int)((long)ncu * (part + 1) / parts);
      for (int i = lo; i < hi; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
      HIPCHK(hipStreamSynchronize(ctx->stream));
      hipStream_t ns = nullptr;
      HIPCHK(hipExtStreamCreateWithCUMask(&ns, (uint32_t)mask.size(), mask.data()));
      HIPCHK(hipStreamDestroy(ctx->stream));
      ctx->stream = ns;
    });
  }
  return fail(PGPU_ERR_INVALID, "unknown flag %s", name);
}

int pgpu_ctx_last_vm_asm(pgpu_ctx* ctx) { return ctx ? ctx->last_vm_asm : 0; }
int pgpu_ctx_last_vm_launches(pgpu_ctx* ctx) { return ctx ? ctx->last_vm_launches : 0; }

const char* pgpu_ctx_last_kernel(pgpu_ctx* ctx) {
  if (!ctx) return "";
  const pgpu_ctx::Ev* best = nullptr;
  for (size_t i = 0; i < ctx->evs_used; ++i)
    if (!best || ctx->evs[i].mads > best->mads) best = &ctx->evs[i];
  return best ? best->name : "";
}

int pgpu_ctx_last_profile(pgpu_ctx* ctx, double* vm_ms, int* vm_launches, double* vm_mads) {
  if (!ctx) return fail(PGPU_ERR_INVALID, "null ctx");
  return guarded([&] {
    ctx->bind();
    double ms = 0, mads = 0;
    for (size_t i = 0; i < ctx->evs_used; ++i) {
      HIPCHK(hipEventSynchronize(ctx->evs[i].b));
      float t = 0;
      HIPCHK(hipEventElapsedTime(&t, ctx->evs[i].a, ctx->evs[i].b));
      ms += t;
      mads += ctx->evs[i].mads;
      // PGPU_PROFILE_DUMP=1 (measurements): one line per profiled VM launch of the last call
      static const bool dump = [] { const char* e = getenv("PGPU_PROFILE_DUMP"); return e && atoi(e) != 0; }();
      if (dump) fprintf(stderr, "[pgpu] launch %2zu %-16s %9.3f ms %14.0f mads  %.3f of peak\n", i, ctx->evs[i].name, t, ctx->evs[i].mads,
                        t > 0 ? ctx->evs[i].mads / (t * 1e-3) / 39.3216e12 : 0.0);
    }
    if (vm_ms) *vm_ms = ms;
    if (vm_launches) *vm_launches = (int)ctx->evs_used;
    if (vm_mads) *vm_mads = mads;
  });
}

// ---- generic modulus ----------------------------------------------------------------------------

int pgpu_modulus_create(pgpu_ctx* ctx, const uint8_t* n_be, size_t n_len, pgpu_modulus** out) {
  if (!ctx || !n_be || !out) return fail(PGPU_ERR_INVALID, "null argument");
  std::unique_ptr<pgpu_modulus> m(new pgpu_modulus());
  int rc = guarded([&] {
    m->ctx = ctx;
    m->mc.init(ctx, BigU::from_be(n_be, n_len));
    m->mc.upload();
  });
  if (rc == PGPU_OK) *out = m.release();
  return rc;
}
void pgpu_modulus_destroy(pgpu_modulus* mod) { delete mod; }
size_t pgpu_modulus_bytes(const pgpu_modulus* mod) { return mod ? mod->mc.nbytes : 0; }

}  // extern "C"

namespace {

// out[i] = base[i]^e mod N on an already-unpacked base array (slot layout described inline).
// Returns the device array of canonical results (WT limbs, limb-major).
// base_wide: the base occupies 2*WT limbs (slots 0 and 1).  post: optional plain multiplicand array.
struct ModexpPlan {
  size_t nb;
  uint32_t* mem;     // slots: 0 in_lo, 1 in_hi, 2 tmp, 3 out, 4 post, 5.. table
  uint32_t* out() const { return mem + 3 * slot_words; }
  uint32_t* in() const { return mem; }
  uint32_t* post() const { return mem + 4 * slot_words; }
  size_t slot_words;
};

ModexpPlan modexp_alloc(pgpu_ctx* ctx, const ModCtx& mc, size_t nb, int table_slots) {
  ModexpPlan pl;
  pl.nb = nb;
  pl.slot_words = (size_t)mc.WT * nb;
  pl.mem = ctx->ws_t<uint32_t>(pl.slot_words * (size_t)(5 + table_slots + 1));   // (+ 1: x limb-major behind a number-major 4-bit table)
  return pl;
}

void reduce_mod(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* in, int w_in, uint32_t* out, size_t nb);

// pl.in() (canonical, < N) ^ e [* pl.post()] mod N = n^2 on the two-lane pair kernel; result lazy in pl.out()
void modexp_pair(pgpu_ctx* ctx, const ModCtx& mc, const ModexpPlan& pl, const BigU* e, const uint32_t* exps, int we,
                 bool use_post, int lanes, uint32_t** raw_out = nullptr) {
  // raw_out: the result stays in PAIR form -- *raw_out = its digits (a0 | a1, 2H limbs, stride nb), F R_H = a0 + a1 n (mod n^2) --
  // for a caller that continues modulo n^3 on the digit kernel (pair_digits of modexp_triple); pl.out() is not written
  const PairInfo& pi = mc.pairn;
  const ModCtx& mn = *pi.root;
  const int H = mn.WT, W2 = mc.WT;
  const size_t nb = pl.nb, S1 = (size_t)H * nb, SW = pl.slot_words;
  uint32_t* mem = pl.mem;
  // (1) X = x R_H mod n^2, canonical, in slot 3
  {
    Prog a;
    a.op(VM_LOAD, 0); a.op(VM_MULC, C_R2); a.op(VM_MULC, (uint32_t)pi.c_rh); a.op(VM_STORE, 3); a.end();
    SegSpec sa{&mc, &a, mem, nullptr};
    run_vm(ctx, nb, sa, nullptr, false);
    launch_canon(mem + 3 * SW, mc.d_nmod, W2, nb, ctx->stream);
  }
  // (2) digits X = X0 + X1 n -> slot 2
  {
    uint32_t* x0 = ctx->ws_t<uint32_t>(S1);
    uint32_t* tb = ctx->ws_t<uint32_t>(SW);
    reduce_mod(ctx, mn, mem + 3 * SW, W2, x0, nb);
    launch_div_exact(mem + 3 * SW, W2, 0, x0, H, tb, pi.dinv, mn.d_nmod, H, mem + 2 * SW + S1, H, nb, nb, nullptr, 0, ctx->stream);
    HIPCHK(hipMemcpyAsync(mem + 2 * SW, x0, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  }
  // (3) the ladder in pair form
  if (lanes == 8 && !exps) {
    // eight lanes per number (GenQ8, shared exponents only): slots of 2 x 76 limbs of their own; the digits are zero-extended,
    // change radix R_74 -> R_76 with the first product of the program and come back with its last
    const int H8 = pi.h8;
    const size_t SW8 = (size_t)2 * H8 * nb;
    uint32_t* m8 = ctx->ws_t<uint32_t>(SW8 * (size_t)(5 + 32));        // pair slots: 2 in, 3 out, 5.. table
    HIPCHK(hipMemsetAsync(m8 + 2 * SW8, 0, SW8 * 4, ctx->stream));
    launch_restride(mem + 2 * SW, nb, nb, nullptr, m8 + 2 * SW8, nb, H, ctx->stream);
    launch_restride(mem + 2 * SW + S1, nb, nb, nullptr, m8 + 2 * SW8 + (size_t)H8 * nb, nb, H, ctx->stream);
    Prog p;
    p.op(VM_LOAD, 2); p.op(VM_MULC, 0); p.op(VM_STORE, 2);
    emit_modexp_shared(p, *e, 2, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    p.op(VM_LOAD, 3); p.op(VM_MULC, 1); p.op(VM_STORE, 3);
    p.end();
    SegSpec sp{&mc, &p, m8, nullptr};
    sp.pair = pi.consts8; sp.pair_n0inv = mn.n0inv; sp.pair_h = H8; sp.pair_lanes = 8; sp.tconsts = pi.tconsts8;
    run_vm(ctx, nb, sp, nullptr, true);
    launch_restride(m8 + 3 * SW8, nb, nb, nullptr, mem + 3 * SW, nb, H, ctx->stream);
    launch_restride(m8 + 3 * SW8 + (size_t)H8 * nb, nb, nb, nullptr, mem + 3 * SW + S1, nb, H, ctx->stream);
  } else {
    Prog p;
    // per-number windows: the table number-major (GenQ / GenQ4 gather a lane's limbs as contiguous bytes)
    if (exps) emit_modexp_perlane(p, we, 2, NO_SLOT, 2, 3, 5, NO_SLOT, pi.c_one_pair, 4, ctx->use_nm4 && (uint64_t)nb * W2 * 4 * 18 < (1ull << 32));
    else emit_modexp_shared(p, *e, 2, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    p.end();
    SegSpec sp{&mc, &p, mem, exps};
    sp.pair = pi.consts; sp.pair_n0inv = mn.n0inv; sp.pair_h = H; sp.pair_lanes = lanes == 8 ? 4 : lanes;
    run_vm(ctx, nb, sp, nullptr, true);
  }
  if (raw_out) {
    *raw_out = mem + 3 * SW;
    return;
  }
  // (4) F~ = F0 + F1 n, out of pair and Montgomery form, times the plain residue in the post slot
  {
    launch_mul_const_add(mem + 3 * SW + S1, H, pi.n_limbs, H, mem + 3 * SW, H, 0, mem + 2 * SW, W2, nb, ctx->stream);
    Prog a;
    a.op(VM_LOAD, 2); a.op(VM_MULC, (uint32_t)pi.c_rh);
    if (use_post) { a.op(VM_MULC, C_R2); a.op(VM_MUL, 4); }
    a.op(VM_STORE, 3); a.end();
    SegSpec sa{&mc, &a, mem, nullptr};
    run_vm(ctx, nb, sa, nullptr, false);
  }
}

// ---- three-digit form for moduli n^3 (GenQ3) ---------------------------------------------------------------------
// Slots of the digit kernel are 3H limbs (a0 | a1 | a2), H = WT(n); the generic kernels' slots of n^3 have WT(n^3) limbs.
struct TriplePlan {
  uint32_t* mem;        // [slot][3H][nb]
  size_t slot_words;    // 3H * nb
  size_t nb;
  int H;
  uint32_t* slot(uint32_t i) const { return mem + (size_t)i * slot_words; }
};

// per-number exponents (28-bit limbs, [we][nb]) -> the 25-bit words VM_MULV5 reads (5 windows of 5 bits each)
const uint32_t* windows5_of(pgpu_ctx* ctx, const uint32_t* exps, int we, size_t nb) {
  const int we5 = (we * LB + 24) / 25;
  uint32_t* out = ctx->ws_t<uint32_t>((size_t)we5 * nb);
  launch_repack_windows5(exps, we, out, we5, nb, ctx->stream);
  return out;
}

// window bits for per-number exponents on the three-digit kernel: 7 while the 128-entry table stays within the kernel's
// 32-bit gather offsets (nb <= 32768 at 2048-bit keys), else 5
int triple_window_bits(size_t nb, int H) { return (uint64_t)nb * 3 * H * 4 * 129 < (1ull << 32) ? 7 : 5; }
const uint32_t* windows5_of(pgpu_ctx* ctx, const uint32_t* exps, int we, size_t nb);
const uint32_t* triple_windows(pgpu_ctx* ctx, const uint32_t* exps, int we, size_t nb, int wb) {
  return wb == 5 ? windows5_of(ctx, exps, we, nb) : exps;
}

bool triple_usable(pgpu_ctx* ctx, const ModCtx& mc, bool allow6 = false) {
  static const bool env_on = [] { const char* e = getenv("PGPU_TRIPLE"); return e ? atoi(e) != 0 : true; }();
  return env_on && mc.triple.root && ctx->use_asm && ctx->use_pair && ctx->use_triple && (allow6 || !mc.triple.lanes6_only);
}

TriplePlan triple_alloc(pgpu_ctx* ctx, const ModCtx& mc, size_t nb, int slots) {
  TriplePlan tp;
  tp.H = mc.triple.root->WT;
  tp.nb = nb;
  tp.slot_words = (size_t)3 * tp.H * nb;
  tp.mem = ctx->ws_t<uint32_t>(tp.slot_words * (size_t)slots);
  return tp;
}

// canonical residue x (WT(n^3) limbs, stride nb) -> digit form of x R_H mod n^3 in slot `slot`
void triple_enter(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, const TriplePlan& tp, uint32_t slot) {
  const TripleInfo& ti = mc.triple;
  const ModCtx &mn = *ti.root, &mn2 = *ti.mid;
  const int H = tp.H, W2 = mn2.WT, W3 = mc.WT;
  const size_t nb = tp.nb, S1 = (size_t)H * nb, S2 = (size_t)W2 * nb, S3 = (size_t)W3 * nb;
  uint32_t* gm = ctx->ws_t<uint32_t>(S3 * 2);       // generic slots: 0 x, 1 X = x R_H mod n^3 (canonical)
  HIPCHK(hipMemcpyAsync(gm, x, S3 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  {
    Prog a;
    a.op(VM_LOAD, 0); a.op(VM_MULC, C_R2); a.op(VM_MULC, (uint32_t)ti.c_rh); a.op(VM_STORE, 1); a.end();
    SegSpec sa{&mc, &a, gm, nullptr};
    run_vm(ctx, nb, sa, nullptr, false);
    launch_canon(gm + S3, mc.d_nmod, W3, nb, ctx->stream);
  }
  uint32_t* X = gm + S3;
  uint32_t* d = tp.slot(slot);
  uint32_t* r2 = ctx->ws_t<uint32_t>(S2);
  uint32_t* Y = ctx->ws_t<uint32_t>(S2);
  uint32_t* tb = ctx->ws_t<uint32_t>(S3);
  reduce_mod(ctx, mn2, X, W3, r2, nb);              // X mod n^2
  reduce_mod(ctx, mn, r2, W2, d, nb);               // X0 = X mod n
  launch_div_exact(X, W3, 0, d, H, tb, ti.dinv2, mn.d_nmod, H, Y, W2, nb, nb, nullptr, 0, ctx->stream);        // Y = (X - X0) / n < n^2
  reduce_mod(ctx, mn, Y, W2, d + S1, nb);           // X1 = Y mod n
  launch_div_exact(Y, W2, 0, d + S1, H, tb, ti.dinv1, mn.d_nmod, H, d + 2 * S1, H, nb, nb, nullptr, 0, ctx->stream);   // X2
}

// pair form (a0 | a1: 2H limbs, stride tp.nb) of a value that matters modulo n^2 only -> digit form (a0, a1, 0) in slot `slot`
void triple_from_pair(pgpu_ctx* ctx, const uint32_t* pair_digits, const TriplePlan& tp, uint32_t slot) {
  const size_t S1 = (size_t)tp.H * tp.nb;
  HIPCHK(hipMemcpyAsync(tp.slot(slot), pair_digits, 2 * S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(tp.slot(slot) + 2 * S1, 0, S1 * 4, ctx->stream));
}

// digit form in slot `slot` (value F R_H) -> canonical F [* post] mod n^3 in `out` (WT(n^3) limbs); post: plain residue
void triple_exit(pgpu_ctx* ctx, const ModCtx& mc, const TriplePlan& tp, uint32_t slot, uint32_t* out, const uint32_t* post) {
  const TripleInfo& ti = mc.triple;
  const ModCtx& mn2 = *ti.mid;
  const int H = tp.H, W2 = mn2.WT, W3 = mc.WT;
  const size_t nb = tp.nb, S1 = (size_t)H * nb, S3 = (size_t)W3 * nb;
  const uint32_t* d = tp.slot(slot);
  uint32_t* t = ctx->ws_t<uint32_t>((size_t)W2 * nb);
  uint32_t* gm = ctx->ws_t<uint32_t>(S3 * 3);       // generic slots: 0 F~, 1 post, 2 out
  launch_mul_const_add(d + S1, H, ti.n_limbs, H, d, H, 0, t, W2, nb, ctx->stream);              // F0 + F1 n
  launch_mul_const_add(d + 2 * S1, H, ti.n2_limbs, W2, t, W2, 0, gm, W3, nb, ctx->stream);      // + F2 n^2  (< 2^(28 WT(n^3)))
  Prog a;
  a.op(VM_LOAD, 0); a.op(VM_MULC, (uint32_t)ti.c_exit);
  if (post) {
    HIPCHK(hipMemcpyAsync(gm + S3, post, S3 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    a.op(VM_MULC, C_R2); a.op(VM_MUL, 1);
  }
  a.op(VM_STORE, 2); a.end();
  SegSpec sa{&mc, &a, gm, nullptr};
  run_vm(ctx, nb, sa, nullptr, false);
  launch_canon(gm + 2 * S3, mc.d_nmod, W3, nb, ctx->stream);
  HIPCHK(hipMemcpyAsync(out, gm + 2 * S3, S3 * 4, hipMemcpyDeviceToDevice, ctx->stream));
}

void triple_run(pgpu_ctx* ctx, const ModCtx& mc, const TriplePlan& tp, const Prog& p, const uint32_t* exps) {
  const TripleInfo& ti = mc.triple;
  SegSpec sp{&mc, &p, tp.mem, exps};
  // two lanes per digit (GenQ6) where the digit does not fit a lane, or where the batch is so small that eight lanes per number
  // still leave every wave a SIMD of its own (one ladder's latency is the run time); not for number-major tables
  const size_t lt = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
  const bool six = ti.lanes6_only || (ctx->use_lanes8 && !p.nm_tables && tp.H % 2 == 0 && vm_asm_available(tp.H / 2, 112) && tp.nb * 8 <= lt);
  sp.pair = ti.kconsts; sp.pair_n0inv = ti.root->n0inv; sp.pair_h = tp.H; sp.pair_lanes = six ? 6 : 3; sp.tconsts = ti.tconsts;
  run_vm(ctx, tp.nb, sp, nullptr, true);
}

// pl.in() (canonical, < n^3) ^ e [* pl.post()] mod n^3 on the three-digit kernel; canonical result in pl.out()
void modexp_triple(pgpu_ctx* ctx, const ModCtx& mc, const ModexpPlan& pl, const BigU* e, const uint32_t* exps, int we,
                   bool use_post, const uint32_t* pair_digits = nullptr) {
  // pair_digits: the base is the result W of a ladder modulo n^2 that is still in pair form, W R_H = a0 + a1 n (mod n^2), and only
  // W mod n^2 matters (the lift: x = x' (mod n^2) implies x^n = x'^n (mod n^3)).  Then (a0, a1, 0) IS the digit form of a valid
  // base: it stands for W'' = (a0 + a1 n) R_H^-1 mod n^3, and W'' = W (mod n^2) because n^2 divides n^3 -- no exit from pair form,
  // no entry into digit form (about 30 small kernels between two ladders that depend on each other): one copy and one memset.
  // digit slots: 0 in, 1 (unused), 2 tmp, 3 out, 5.. table (32 entries: sliding windows of a shared exponent, or the 5-bit
  // windows of per-number exponents -- a product costs two squarings here, so the wider window pays)
  const int wb = (exps && !mc.triple.lanes6_only) ? triple_window_bits(pl.nb, mc.triple.root->WT) : 5;   // (GenQ6: limb-major tables)
  // (5-bit windows -- batches whose 128-entry tables would not fit the 32-bit gather offsets -- on number-major tables as well:
  // VM_MULVT5; the two-lanes-per-digit kernel has limb-major tables only)
  const bool nm5 = exps && wb == 5 && !mc.triple.lanes6_only;
  TriplePlan tp = triple_alloc(ctx, mc, pl.nb, 5 + perlane_table_slots(wb, nm5));
  if (pair_digits) triple_from_pair(ctx, pair_digits, tp, 0);
  else triple_enter(ctx, mc, pl.in(), tp, 0);
  Prog p;
  if (exps) {
    emit_modexp_perlane(p, we, 0, NO_SLOT, 2, 3, 5, NO_SLOT, 0, wb, nm5);
    exps = triple_windows(ctx, exps, we, pl.nb, wb);
  } else {
    emit_modexp_shared(p, *e, 0, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
  }
  p.end();
  triple_run(ctx, mc, tp, p, exps);
  triple_exit(ctx, mc, tp, 3, pl.out(), use_post ? pl.post() : nullptr);
}

void modexp_shared_run(pgpu_ctx* ctx, const ModCtx& mc, const ModexpPlan& pl, const BigU& e, bool wide, bool use_post,
                       bool skip_zero, uint32_t** raw_pair_out = nullptr, const uint32_t* pair_digits_in = nullptr) {
  // raw_pair_out: if the ladder runs on a pair kernel its result may stay in pair form (*raw_pair_out set, pl.out() not written);
  // pair_digits_in: the base of a ladder modulo n^3, still in the pair form of the ladder modulo n^2 before it (modexp_triple)
  if (raw_pair_out) *raw_pair_out = nullptr;
  // two lanes per number from one wave per SIMD upwards; below that four (each digit over two lanes: a squaring is half as
  // long as on the 4-lane 2H-limb kernel, which is what counts when the ladder's latency is the run time)
  if (triple_usable(ctx, mc, true) && !wide && skip_zero && e.bit_length() >= 256) {
    modexp_triple(ctx, mc, pl, &e, nullptr, 0, use_post, pair_digits_in);
    return;
  }
  if (pair_digits_in) api_throw(PGPU_ERR_UNSUPPORTED, "internal: pair digits handed to a ladder that is not on the digit kernel");
  const bool two = pl.nb * 2 >= (ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64);
  if (mc.pairn.root && ctx->use_asm && ctx->use_pair && !wide && skip_zero && e.bit_length() >= 256 &&
      (two || (mc.pairn.root->WT % 2 == 0 && vm_asm_available(mc.pairn.root->WT / 2, 64)))) {
    // (a batch that leaves SIMDs empty even at four lanes per number is bound by one ladder's latency: eight lanes, GenQ8)
    const size_t lt = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
    const bool eight = !two && mc.pairn.consts8 && ctx->use_lanes8 && pl.nb * 8 <= lt;
    modexp_pair(ctx, mc, pl, &e, nullptr, 0, use_post, two ? 2 : eight ? 8 : 4, use_post ? nullptr : raw_pair_out);
    if (!(raw_pair_out && *raw_pair_out)) launch_canon(pl.out(), mc.d_nmod, mc.WT, pl.nb, ctx->stream);
    return;
  }
  Prog p;
  emit_modexp_shared(p, e, 0, wide ? 1 : NO_SLOT, 2, 3, 5, use_post ? 4 : NO_SLOT, skip_zero);
  p.end();
  SegSpec s{&mc, &p, pl.mem, nullptr};
  run_vm(ctx, pl.nb, s, nullptr, true);
  launch_canon(pl.out(), mc.d_nmod, mc.WT, pl.nb, ctx->stream);
}

void modexp_perlane_run(pgpu_ctx* ctx, const ModCtx& mc, const ModexpPlan& pl, const uint32_t* exps, int we, bool wide,
                        bool use_post) {
  if (triple_usable(ctx, mc, true) && !wide && we >= 10 && (uint64_t)pl.nb * (mc.WT + 4) * 4 * 33 < (1ull << 32)) {
    modexp_triple(ctx, mc, pl, nullptr, exps, we, use_post);
    return;
  }
  {
    const bool two = pl.nb * 2 >= (ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64);
    const uint64_t table_bytes = (uint64_t)pl.nb * mc.WT * 4 * 17;      // MULV gathers with 32-bit offsets
    if (mc.pairn.root && mc.pairn.c_one_pair >= 0 && ctx->use_asm && ctx->use_pair && !wide && we >= 10 &&
        table_bytes < (1ull << 32) && (two || (mc.pairn.root->WT % 2 == 0 && vm_asm_available(mc.pairn.root->WT / 2, 64)))) {
      modexp_pair(ctx, mc, pl, nullptr, exps, we, use_post, two ? 2 : 4);
      launch_canon(pl.out(), mc.d_nmod, mc.WT, pl.nb, ctx->stream);
      return;
    }
  }
  Prog p;
  emit_modexp_perlane(p, we, 0, wide ? 1 : NO_SLOT, 2, 3, 5, use_post ? 4 : NO_SLOT);
  p.end();
  SegSpec s{&mc, &p, pl.mem, exps};
  run_vm(ctx, pl.nb, s, nullptr, true);
  launch_canon(pl.out(), mc.d_nmod, mc.WT, pl.nb, ctx->stream);
}

// x mod N for an array of `w_in` <= 2*WT limbs -> canonical WT limbs in `out`
void reduce_mod(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* in, int w_in, uint32_t* out, size_t nb) {
  // slots: 0 lo, 1 hi, 2 tmp, 3 out
  size_t sw = (size_t)mc.WT * nb;
  uint32_t* mem = ctx->ws_t<uint32_t>(sw * 4);
  bool wide = w_in > mc.WT;
  launch_copy_limbs(in, 0, std::min(w_in, mc.WT), mem, mc.WT, nb, ctx->stream);
  if (wide) launch_copy_limbs(in, mc.WT, w_in - mc.WT, mem + sw, mc.WT, nb, ctx->stream);
  Prog p;
  emit_to_mont(p, 0, wide ? 1 : NO_SLOT, 2);
  p.op(VM_MULC, C_ONE);
  p.op(VM_STORE, 3);
  p.end();
  SegSpec s{&mc, &p, mem, nullptr};
  run_vm(ctx, nb, s, nullptr, false);
  launch_canon(mem + 3 * sw, mc.d_nmod, mc.WT, nb, ctx->stream);
  HIPCHK(hipMemcpyAsync(out, mem + 3 * sw, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
}

// Stage an operand that is read modulo N, whatever its stride.  Up to the width of the modulus the bytes go straight into
// WT limbs (a Montgomery operand may be any value below R; `canonical` additionally reduces it below N).  A wider stride
// is unpacked whole and reduced chunk by chunk (Horner over WT-limb chunks), so that no leading byte is silently
// dropped: the reference's Exp / Mul+Mod reduce any value correctly.
void unpack_mod(pgpu_ctx* ctx, const ModCtx& mc, const uint8_t* buf, size_t stride, size_t count, int mem, uint32_t* out,
                size_t nb, bool canonical = false) {
  const int WT = mc.WT;
  if (stride * 8 <= (size_t)LB * WT) {
    if (!canonical) { unpack_operand(ctx, buf, stride, stride, count, mem, out, WT, nb); return; }
    uint32_t* raw = ctx->ws_t<uint32_t>((size_t)WT * nb);
    unpack_operand(ctx, buf, stride, stride, count, mem, raw, WT, nb);
    reduce_mod(ctx, mc, raw, WT, out, nb);
    return;
  }
  const int w_in = (int)((stride * 8 + LB - 1) / LB);
  uint32_t* wide = ctx->ws_t<uint32_t>((size_t)w_in * nb);
  unpack_operand(ctx, buf, stride, stride, count, mem, wide, w_in, nb);
  if (w_in <= 2 * WT) { reduce_mod(ctx, mc, wide, w_in, out, nb); return; }
  const size_t sw = (size_t)WT * nb;
  const int nchunks = (w_in + WT - 1) / WT;
  uint32_t* arr = ctx->ws_t<uint32_t>(2 * sw);   // [chunk | running remainder]: the value chunk + rem * 2^(28 WT)
  launch_copy_limbs(wide, (nchunks - 1) * WT, w_in - (nchunks - 1) * WT, arr + sw, WT, nb, ctx->stream);
  for (int k = nchunks - 2; k >= 0; --k) {
    launch_copy_limbs(wide, k * WT, WT, arr, WT, nb, ctx->stream);
    reduce_mod(ctx, mc, arr, 2 * WT, k ? arr + sw : out, nb);
  }
}

// Batch modular inverse (gmp.Int.ModInverse for a whole batch) by Montgomery's trick arranged as a binary tree so that
// every level is one data-parallel VM launch: products up the tree (number i with number i + half: the upper half of a
// level is copied next to the lower half, so the programs are plain LOAD / MUL / STORE and run on the assembly kernels),
// ONE inversion of the root on the host, inverses down the tree.  3 Montgomery products per element instead of a
// ~2*bits-step extended Euclid per element.  x: canonical, stride nb, `count` valid.  Returns canonical inverses with
// stride nb, or nullptr when the root is not invertible (some element is not a unit).
uint32_t* tree_inverse(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count) {
  const int WT = mc.WT;
  size_t nbt = VM_BLOCK;
  int L = 8;
  while (nbt < count) { nbt <<= 1; ++L; }
  const size_t sw = (size_t)WT * nbt;
  // slots: V_0..V_L, U (upper half of the current level, moved down), I0, I1, A, B, O
  const uint32_t SV = 0, SU = (uint32_t)L + 1, SI0 = SU + 1, SI1 = SI0 + 1, SA = SI0 + 2, SB = SI0 + 3, SO = SI0 + 4;
  uint32_t* mem = ctx->ws_t<uint32_t>(sw * (size_t)(L + 7));
  HIPCHK(hipMemsetAsync(mem + (size_t)SU * sw, 0, sw * 4, ctx->stream));
  auto upper_half = [&](int k, size_t half) {   // U[i] <- V_k[i + half], i < half
    HIPCHK(hipMemcpy2DAsync(mem + (size_t)SU * sw, nbt * 4, mem + (size_t)(SV + k) * sw + half, nbt * 4, half * 4, (size_t)WT,
                            hipMemcpyDeviceToDevice, ctx->stream));
  };
  // V_0 = x (padding lanes = 1), to Montgomery form
  launch_restride(x, nb, count, mc.d_consts + (size_t)C_ONE * WT, mem + SV * sw, nbt, WT, ctx->stream);
  {
    Prog p;
    p.op(VM_LOAD, SV); p.op(VM_MULC, C_R2); p.op(VM_STORE, SV); p.end();
    SegSpec sg{&mc, &p, mem, nullptr};
    run_vm(ctx, nbt, sg, nullptr, false);
  }
  for (int k = 0; k < L; ++k) {  // V_{k+1}[i] = V_k[i] * V_k[i + half]
    const size_t half = nbt >> (k + 1);
    upper_half(k, half);
    Prog p;
    p.op(VM_LOAD, SV + k); p.op(VM_MUL, SU); p.op(VM_STORE, SV + k + 1); p.end();
    SegSpec sg{&mc, &p, mem, nullptr};
    run_vm(ctx, nbt, sg, nullptr, false, std::max<size_t>(VM_BLOCK, half));
  }
  // root: out of Montgomery form, canonical, to the host
  {
    Prog p;
    p.op(VM_LOAD, SV + L); p.op(VM_MULC, C_ONE); p.op(VM_STORE, SO); p.end();
    SegSpec sg{&mc, &p, mem, nullptr};
    run_vm(ctx, nbt, sg, nullptr, false, VM_BLOCK);
    launch_canon(mem + SO * sw, mc.d_nmod, WT, nbt, ctx->stream);
  }
  std::vector<uint8_t> rb(mc.nbytes);
  uint8_t* d_rb = (uint8_t*)ctx->ws(mc.nbytes);
  launch_pack_be(mem + SO * sw, WT, nbt, 1, d_rb, mc.nbytes, mc.nbytes, ctx->stream);
  HIPCHK(hipMemcpyAsync(rb.data(), d_rb, mc.nbytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  BigU root = BigU::from_be(rb.data(), rb.size()), rinv;
  if (!hostbig::modinv(root, mc.N, rinv)) return nullptr;
  uint32_t* d_rinv = ctx->upload_words(mc.to_mont(rinv).to_limbs(LB, WT));
  // every lane of I0 <- root inverse (count = 0: all lanes take the fill value); only lane 0 is consumed
  launch_restride(mem + SI0 * sw, nbt, 0, d_rinv, mem + SI0 * sw, nbt, WT, ctx->stream);
  uint32_t cur = SI0, nxt = SI1;
  for (int k = L - 1; k >= 0; --k) {
    const size_t half = nbt >> (k + 1);
    upper_half(k, half);
    Prog p;
    p.op(VM_LOAD, cur); p.op(VM_MUL, SU); p.op(VM_STORE, SA);         // inverse of V_k[i]        = I_{k+1}[i] V_k[i + half]
    p.op(VM_LOAD, cur); p.op(VM_MUL, SV + k); p.op(VM_STORE, SB);     // inverse of V_k[i + half] = I_{k+1}[i] V_k[i]
    p.end();
    SegSpec sg{&mc, &p, mem, nullptr};
    run_vm(ctx, nbt, sg, nullptr, false, std::max<size_t>(VM_BLOCK, half));
    launch_merge_halves(mem + SA * sw, mem + SB * sw, half, mem + nxt * sw, nbt, WT, ctx->stream);
    std::swap(cur, nxt);
  }
  {
    Prog p;
    p.op(VM_LOAD, cur); p.op(VM_MULC, C_ONE); p.op(VM_STORE, SO); p.end();
    SegSpec sg{&mc, &p, mem, nullptr};
    run_vm(ctx, nbt, sg, nullptr, false);
    launch_canon(mem + SO * sw, mc.d_nmod, WT, nbt, ctx->stream);
  }
  uint32_t* out = ctx->ws_t<uint32_t>((size_t)WT * nb);
  launch_restride(mem + SO * sw, nbt, count, nullptr, out, nb, WT, ctx->stream);
  return out;
}

// Are ALL of x[0 .. count) units modulo N?  The up-sweep of the same product tree and one gcd on the host -- half the
// launches of tree_inverse and no inverses; what the randomness filter needs (utils.go:43: gcd(r, n) = 1) in the
// overwhelmingly likely case that every draw is a unit.
// all_units in two halves for a caller that has something to run meanwhile: begin() issues the product tree (to the stream the
// context is on) and leaves the root's bytes on the device, finish() fetches them, waits and tests the gcd on the host.
struct UnitCheck {
  pgpu_ctx* ctx = nullptr;
  const ModCtx* mc = nullptr;
  uint8_t* d_rb = nullptr;
  hipStream_t st = nullptr;
  std::unique_ptr<SideStream> side;
  bool begun = false;
  void begin(pgpu_ctx* c, const ModCtx& m, const uint32_t* x, size_t nb, size_t count);
  bool finish() {
    std::vector<uint8_t> rb(mc->nbytes);
    HIPCHK(hipMemcpyAsync(rb.data(), d_rb, mc->nbytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (side) side->dirty = false;
    BigU root = BigU::from_be(rb.data(), rb.size()), rinv;
    return hostbig::modinv(root, mc->N, rinv);
  }
};
void UnitCheck::begin(pgpu_ctx* c, const ModCtx& m, const uint32_t* x, size_t nb, size_t count) {
  ctx = c;
  mc = &m;
  st = c->stream;
  begun = true;
  const int WT = m.WT;
  size_t nbt = VM_BLOCK;
  int L = 8;
  while (nbt < count) { nbt <<= 1; ++L; }
  const size_t sw = (size_t)WT * nbt;
  uint32_t* mem = ctx->ws_t<uint32_t>(sw * 3);                      // slots: 0 V, 1 U (upper half moved down), 2 O
  HIPCHK(hipMemsetAsync(mem + sw, 0, sw * 4, ctx->stream));
  launch_restride(x, nb, count, m.d_consts + (size_t)C_ONE * WT, mem, nbt, WT, ctx->stream);
  for (int k = 0; k < L; ++k) {
    const size_t half = nbt >> (k + 1);
    HIPCHK(hipMemcpy2DAsync(mem + sw, nbt * 4, mem + half, nbt * 4, half * 4, (size_t)WT, hipMemcpyDeviceToDevice, ctx->stream));
    Prog p;
    p.op(VM_LOAD, 0); p.op(VM_MUL, 1); p.op(VM_STORE, 0); p.end();
    SegSpec sg{&m, &p, mem, nullptr};
    run_vm(ctx, nbt, sg, nullptr, false, std::max<size_t>(VM_BLOCK, half));
  }
  launch_canon(mem, m.d_nmod, WT, nbt, ctx->stream);
  d_rb = (uint8_t*)ctx->ws(m.nbytes);
  launch_pack_be(mem, WT, nbt, 1, d_rb, m.nbytes, m.nbytes, ctx->stream);
}

bool all_units(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count) {
  const int WT = mc.WT;
  size_t nbt = VM_BLOCK;
  int L = 8;
  while (nbt < count) { nbt <<= 1; ++L; }
  const size_t sw = (size_t)WT * nbt;
  uint32_t* mem = ctx->ws_t<uint32_t>(sw * 3);                      // slots: 0 V, 1 U (upper half moved down), 2 O
  HIPCHK(hipMemsetAsync(mem + sw, 0, sw * 4, ctx->stream));
  launch_restride(x, nb, count, mc.d_consts + (size_t)C_ONE * WT, mem, nbt, WT, ctx->stream);
  // (plain residues multiplied with Montgomery products: every level loses a factor R, all of them units -- the gcd of the
  // root with N is that of the product)
  for (int k = 0; k < L; ++k) {
    const size_t half = nbt >> (k + 1);
    HIPCHK(hipMemcpy2DAsync(mem + sw, nbt * 4, mem + half, nbt * 4, half * 4, (size_t)WT, hipMemcpyDeviceToDevice, ctx->stream));
    Prog p;
    p.op(VM_LOAD, 0); p.op(VM_MUL, 1); p.op(VM_STORE, 0); p.end();
    SegSpec sg{&mc, &p, mem, nullptr};
    run_vm(ctx, nbt, sg, nullptr, false, std::max<size_t>(VM_BLOCK, half));
  }
  launch_canon(mem, mc.d_nmod, WT, nbt, ctx->stream);
  std::vector<uint8_t> rb(mc.nbytes);
  uint8_t* d_rb = (uint8_t*)ctx->ws(mc.nbytes);
  launch_pack_be(mem, WT, nbt, 1, d_rb, mc.nbytes, mc.nbytes, ctx->stream);
  HIPCHK(hipMemcpyAsync(rb.data(), d_rb, mc.nbytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  BigU root = BigU::from_be(rb.data(), rb.size()), rinv;
  return hostbig::modinv(root, mc.N, rinv);
}

// gmp.Int.ModInverse for a batch.  d_bad (device int32[nb], may be null) receives 1 on the lanes that are not units and 0
// elsewhere; those lanes get the result 0 (mpz_invert leaves its result undefined there and the reference never checks).
// One hostile element must not cost the honest ones their answers: when the tree's root cannot be inverted, a per-lane
// binary GCD finds the non-units, 1 is substituted for them and the tree runs again.  With d_bad == nullptr a non-unit
// throws PGPU_ERR_NOT_INVERTIBLE (callers for which a non-unit means the whole call is meaningless).
uint32_t* batch_inverse(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count, int32_t* d_bad = nullptr,
                        bool* any_bad = nullptr) {
  if (any_bad) *any_bad = false;
  if (d_bad) HIPCHK(hipMemsetAsync(d_bad, 0, nb * 4, ctx->stream));
  uint32_t* out = tree_inverse(ctx, mc, x, nb, count);
  if (out) return out;
  if (!d_bad) api_throw(PGPU_ERR_NOT_INVERTIBLE, "ModInverse: an element of the batch is not invertible modulo the modulus");
  const size_t sw = (size_t)mc.WT * nb;
  uint32_t* work = ctx->ws_t<uint32_t>(2 * sw);
  launch_unit_flags(x, mc.d_nmod, mc.WT, nb, count, work, d_bad, ctx->stream);
  uint32_t* x1 = ctx->ws_t<uint32_t>(sw);
  HIPCHK(hipMemcpyAsync(x1, x, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
  launch_select_const(d_bad, mc.d_consts + (size_t)C_ONE * mc.WT, x1, mc.WT, nb, ctx->stream);
  out = tree_inverse(ctx, mc, x1, nb, count);
  if (!out) api_throw(PGPU_ERR_NOT_INVERTIBLE, "ModInverse: internal error (non-unit survived the unit test)");
  uint32_t* zero = ctx->ws_t<uint32_t>((size_t)mc.WT);
  HIPCHK(hipMemsetAsync(zero, 0, (size_t)mc.WT * 4, ctx->stream));
  launch_select_const(d_bad, zero, out, mc.WT, nb, ctx->stream);
  if (any_bad) *any_bad = true;
  return out;
}

// After a batch_inverse with per-lane flags: OR PGPU_LANE_NOT_INVERTIBLE into the caller's status array, or -- when the caller
// passed none -- report the failure through the return code once the outputs have been written (finish_bad_lanes()).
struct BadLanes {
  bool any = false;
  std::vector<int32_t> host;
  void collect(pgpu_ctx* ctx, const int32_t* d_bad, size_t batch, bool any_bad) {
    if (!any_bad) return;
    any = true;
    host.resize(batch);
    HIPCHK(hipMemcpyAsync(host.data(), d_bad, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  void finish(int32_t* status, size_t batch) const {
    if (status) {
      for (size_t i = 0; i < batch; ++i) status[i] = (any && host[i]) ? PGPU_LANE_NOT_INVERTIBLE : PGPU_LANE_OK;
      return;
    }
    if (any)
      api_throw(PGPU_ERR_NOT_INVERTIBLE, "ModInverse of a non-unit: the invertible lanes were computed, the others are zero "
                                         "(pass a status array to get them per lane)");
  }
};

void check_batch_args(const void* a, const void* b, size_t batch) {
  if (!a || !b) api_throw(PGPU_ERR_INVALID, "null buffer");
  if (batch == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
  if (batch > (1u << 26)) api_throw(PGPU_ERR_INVALID, "batch too large");
}

}  // namespace

extern "C" {

int pgpu_modexp(const pgpu_modulus* mod, size_t batch, const uint8_t* base, size_t base_stride, size_t base_len,
                const uint8_t* e, size_t e_len, size_t e_stride, uint8_t* out, size_t out_stride, int mem) {
  if (!mod) return fail(PGPU_ERR_INVALID, "null modulus");
  pgpu_ctx* ctx = mod->ctx;
  const ModCtx& mc = mod->mc;
  return guarded([&] {
    check_batch_args(base, out, batch);
    if (!e) api_throw(PGPU_ERR_INVALID, "null exponent");
    ctx->bind();
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    const bool wide = base_len * 8 > (size_t)LB * mc.WT;
    if (base_len * 8 > (size_t)2 * LB * mc.WT) api_throw(PGPU_ERR_INVALID, "base wider than twice the modulus width");
    const bool perlane = e_stride != 0;
    ModexpPlan pl = modexp_alloc(ctx, mc, nb, perlane ? 16 : 32);
    unpack_operand(ctx, base, base_stride, base_len, batch, mem, pl.in(), wide ? 2 * mc.WT : mc.WT, nb);
    if (!perlane) {
      BigU ev = BigU::from_be(e, e_len);
      modexp_shared_run(ctx, mc, pl, ev, wide, false, true);
    } else {
      int we = (int)((e_len * 8 + LB - 1) / LB);
      if (we < 1) we = 1;
      uint32_t* exps = ctx->ws_t<uint32_t>((size_t)we * nb);
      unpack_operand(ctx, e, e_stride, e_len, batch, mem, exps, we, nb);
      modexp_perlane_run(ctx, mc, pl, exps, we, wide, false);
    }
    pack_result(ctx, pl.out(), mc.WT, nb, batch, out, out_stride, mc.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

/* Test hook: run a raw VM program on raw limb-major slot memory (host arrays of 28-bit limbs).
 * mem_words = nslots * WT * nb uint32.  Used by tests/ to compare the assembly and hipcc kernels per opcode. */
int pgpu_vm_debug_run(const pgpu_modulus* mod, const uint32_t* prog, size_t prog_words, uint32_t* mem_host,
                      size_t nslots, size_t nb, int use_asm, int* wt_out) {
  if (!mod || !prog || !mem_host) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = mod->ctx;
  const ModCtx& mc = mod->mc;
  if (wt_out) *wt_out = mc.WT;
  return guarded([&] {
    if (nb % VM_BLOCK) api_throw(PGPU_ERR_INVALID, "nb must be a multiple of 256");
    ctx->bind();
    ctx->reset_ws();
    size_t words = nslots * (size_t)mc.WT * nb;
    uint32_t* d = ctx->ws_t<uint32_t>(words);
    HIPCHK(hipMemcpyAsync(d, mem_host, words * 4, hipMemcpyHostToDevice, ctx->stream));
    Prog p;
    p.w.assign(prog, prog + prog_words);
    p.asm_ok = true;
    bool saved = ctx->use_asm;
    ctx->use_asm = use_asm != 0;
    SegSpec s{&mc, &p, d, nullptr};
    try { run_vm(ctx, nb, s, nullptr, false); } catch (...) { ctx->use_asm = saved; throw; }
    ctx->use_asm = saved;
    HIPCHK(hipMemcpyAsync(mem_host, d, words * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_pair_debug_run(pgpu_ctx* ctx, const uint8_t* p_be, size_t p_len, int lanes, const uint32_t* prog, size_t prog_words,
                        uint32_t* mem_host, size_t nslots, size_t nb, uint32_t* consts_out, int* h_out) {
  if (!ctx || !p_be || !prog || !mem_host) return fail(PGPU_ERR_INVALID, "null argument");
  return guarded([&] {
    if (nb % VM_BLOCK) api_throw(PGPU_ERR_INVALID, "nb must be a multiple of 256");
    ctx->bind();
    ctx->reset_ws();
    const BigU pr = BigU::from_be(p_be, p_len);
    ModCtx mp, mp2;
    mp.init(ctx, pr);
    mp2.init(ctx, pr * pr);
    if (lanes == 3 || lanes == 6) {
      // three-digit kernel: slots are [3H][nb] (a0 | a1 | a2), the root is `p_be`; constants: one entry, the zero-extended
      // digits given in consts_out on entry are NOT used -- the test passes constants as slots.  lanes = 6: two lanes per digit
      if (lanes == 3 ? (mp.K != 1 || !vm_asm_available(mp.WT, 48)) : (mp.WT % 2 != 0 || !vm_asm_available(mp.WT / 2, 112)))
        api_throw(PGPU_ERR_UNSUPPORTED, "no three-digit kernel for this width");
      const int H = mp.WT;
      if (h_out) *h_out = H;
      ModCtx mp3;
      mp3.init(ctx, pr * pr * pr);
      mp3.upload();
      std::vector<uint32_t> kc = make_triple_kconsts(pr, H);
      if (consts_out) memcpy(consts_out, kc.data(), std::min(kc.size(), (size_t)3 * H) * 4);
      uint32_t* d_kc = ctx->upload_words(kc);
      size_t words = nslots * (size_t)3 * H * nb;
      uint32_t* d = ctx->ws_t<uint32_t>(words);
      HIPCHK(hipMemcpyAsync(d, mem_host, words * 4, hipMemcpyHostToDevice, ctx->stream));
      Prog p;
      p.w.assign(prog, prog + prog_words);
      p.asm_ok = true;
      SegSpec s{&mp3, &p, d, nullptr};
      s.pair = d_kc; s.pair_n0inv = mp.n0inv; s.pair_h = H; s.pair_lanes = lanes; s.tconsts = d;   // constant c = slot c
      bool saved = ctx->use_asm;
      ctx->use_asm = true;
      try { run_vm(ctx, nb, s, nullptr, false); } catch (...) { ctx->use_asm = saved; throw; }
      ctx->use_asm = saved;
      HIPCHK(hipMemcpyAsync(mem_host, d, words * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      return;
    }
    if (lanes != 1 && lanes != 2 && lanes != 4) api_throw(PGPU_ERR_INVALID, "lanes must be 1, 2, 3, 4 or 6");
    if (mp.K != 1 || mp2.WT != 2 * mp.WT ||
        !(lanes == 4 ? (mp.WT % 2 == 0 && vm_asm_available(mp.WT / 2, 64)) : vm_asm_available(mp.WT, lanes == 2 ? 32 : 16)))
      api_throw(PGPU_ERR_UNSUPPORTED, "no pair kernel for this width");
    mp2.upload();
    const int H = mp.WT;
    if (h_out) *h_out = H;
    std::vector<uint32_t> pc = make_pair_consts(pr, H);
    if (consts_out) memcpy(consts_out, pc.data(), pc.size() * 4);
    pc.push_back(0);
    uint32_t* d_pc = ctx->upload_words(pc);
    size_t words = nslots * (size_t)mp2.WT * nb;
    uint32_t* d = ctx->ws_t<uint32_t>(words);
    HIPCHK(hipMemcpyAsync(d, mem_host, words * 4, hipMemcpyHostToDevice, ctx->stream));
    Prog p;
    p.w.assign(prog, prog + prog_words);
    p.asm_ok = true;
    SegSpec s{&mp2, &p, d, nullptr};
    s.pair = d_pc; s.pair_n0inv = mp.n0inv; s.pair_h = H; s.pair_lanes = lanes;
    bool saved = ctx->use_asm;
    ctx->use_asm = true;
    try { run_vm(ctx, nb, s, nullptr, false); } catch (...) { ctx->use_asm = saved; throw; }
    ctx->use_asm = saved;
    HIPCHK(hipMemcpyAsync(mem_host, d, words * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_modinv(const pgpu_modulus* mod, size_t batch, const uint8_t* x, size_t x_stride, size_t x_len, uint8_t* out,
                size_t out_stride, int mem, int32_t* status) {
  if (!mod) return fail(PGPU_ERR_INVALID, "null modulus");
  pgpu_ctx* ctx = mod->ctx;
  const ModCtx& mc = mod->mc;
  return guarded([&] {
    check_batch_args(x, out, batch);
    ctx->bind();
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    const int w_in = (int)((x_len * 8 + LB - 1) / LB);
    if (w_in > 2 * mc.WT) api_throw(PGPU_ERR_INVALID, "operand wider than twice the modulus width");
    uint32_t* xl = ctx->ws_t<uint32_t>((size_t)std::max(w_in, mc.WT) * nb);
    unpack_operand(ctx, x, x_stride, x_len, batch, mem, xl, std::max(w_in, mc.WT), nb);
    uint32_t* xr = ctx->ws_t<uint32_t>((size_t)mc.WT * nb);
    reduce_mod(ctx, mc, xl, std::max(w_in, mc.WT), xr, nb);
    int32_t* d_bad = ctx->ws_t<int32_t>(nb);
    bool any_bad = false;
    uint32_t* inv = batch_inverse(ctx, mc, xr, nb, batch, d_bad, &any_bad);
    BadLanes bl;
    bl.collect(ctx, d_bad, batch, any_bad);
    pack_result(ctx, inv, mc.WT, nb, batch, out, out_stride, mc.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    bl.finish(status, batch);
  });
}

int pgpu_modmul(const pgpu_modulus* mod, size_t batch, const uint8_t* a, size_t a_stride, size_t a_len,
                const uint8_t* b, size_t b_stride, size_t b_len, uint8_t* out, size_t out_stride, int mem) {
  if (!mod) return fail(PGPU_ERR_INVALID, "null modulus");
  pgpu_ctx* ctx = mod->ctx;
  const ModCtx& mc = mod->mc;
  return guarded([&] {
    check_batch_args(a, out, batch);
    if (!b) api_throw(PGPU_ERR_INVALID, "null buffer");
    if (a_len * 8 > (size_t)LB * mc.WT || b_len * 8 > (size_t)LB * mc.WT)
      api_throw(PGPU_ERR_INVALID, "modmul operands must fit the modulus width");
    ctx->bind();
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    size_t sw = (size_t)mc.WT * nb;
    uint32_t* memv = ctx->ws_t<uint32_t>(sw * 3);  // slots: 0 a, 1 b, 2 out
    unpack_operand(ctx, a, a_stride, a_len, batch, mem, memv, mc.WT, nb);
    unpack_operand(ctx, b, b_stride, b_len, batch, mem, memv + sw, mc.WT, nb);
    Prog p;
    p.op(VM_LOAD, 0);
    p.op(VM_MULC, C_R2);
    p.op(VM_MUL, 1);
    p.op(VM_STORE, 2);
    p.end();
    SegSpec s{&mc, &p, memv, nullptr};
    run_vm(ctx, nb, s, nullptr, true);
    launch_canon(memv + 2 * sw, mc.d_nmod, mc.WT, nb, ctx->stream);
    pack_result(ctx, memv + 2 * sw, mc.WT, nb, batch, out, out_stride, mc.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

// ---- keys ---------------------------------------------------------------------------------------

int pgpu_pubkey_create(pgpu_ctx* ctx, const uint8_t* n_be, size_t n_len, const uint8_t* g_be, size_t g_len,
                       const uint8_t* h_be, size_t h_len, const uint8_t* k_be, size_t k_len, pgpu_pubkey** out) {
  if (!ctx || !n_be || !g_be || !out) return fail(PGPU_ERR_INVALID, "null argument");
  std::unique_ptr<pgpu_pubkey> pk(new pgpu_pubkey());
  int rc = guarded([&] {
    ctx->bind();
    pk->ctx = ctx;
    pk->N = BigU::from_be(n_be, n_len);
    pk->G = BigU::from_be(g_be, g_len);
    if (h_be) pk->H = BigU::from_be(h_be, h_len);
    if (k_be) pk->Kk = BigU::from_be(k_be, k_len);
    pk->g_is_n_plus_1 = (pk->G == pk->N + BigU(1));
    BigU n2 = pk->N * pk->N;
    pk->mn.init(ctx, pk->N);
    pk->mn2.init(ctx, n2);
    pk->mn.upload();
    pk->mn2.upload();
    int wl, k;
    BigU n3 = n2 * pk->N;
    if (ModCtx::pick_shape(n3.bit_length(), wl, k)) {
      pk->mn3.reset(new ModCtx());
      pk->mn3->init(ctx, n3);
      pk->mn3->upload();
    }
    pk->n_limbs.set(pk->N, pk->mn.WT);
    pk->ninv2k.set(inv_mod_pow2(pk->N, (size_t)LB * pk->mn.WT), pk->mn.WT);
    pk->ninv2k_2.set(inv_mod_pow2(pk->N, (size_t)LB * pk->mn2.WT), pk->mn2.WT);
    pk->n2_limbs.set(n2, pk->mn2.WT);
    BigU inv2 = hostbig::shr(pk->N + BigU(1), 1);                       // 2^-1 mod n
    BigU inv2_n2 = hostbig::shr(n2 + BigU(1), 1);                       // 2^-1 mod n^2
    pk->c_inv2R = pk->mn.add_const(pk->mn.to_mont(inv2));
    pk->c_ninv2R_2 = pk->mn2.add_const(pk->mn2.to_mont(hostbig::mulmod(pk->N, inv2_n2, n2)));
    if (pk->mn.K == 1 && pk->mn2.WT == 2 * pk->mn.WT && vm_asm_available(pk->mn.WT, 32)) {
      const int H = pk->mn.WT;
      std::vector<uint32_t> pc = make_pair_consts(pk->N, H);
      pc.push_back(0);   // the kernel prefetches one word past Cadj
      pk->pairn_consts.w = (int)pc.size();
      HIPCHK(hipMalloc((void**)&pk->pairn_consts.d, pc.size() * 4));
      HIPCHK(hipMemcpy(pk->pairn_consts.d, pc.data(), pc.size() * 4, hipMemcpyHostToDevice));
      PairInfo& pi = pk->mn2.pairn;
      pi.root = &pk->mn;
      pi.consts = pk->pairn_consts.d;
      const BigU rh = hostbig::shl(BigU(1), (size_t)LB * H) % n2;
      pi.c_rh = pk->mn2.add_const(rh);
      {
        BigU d1, d0;
        hostbig::divmod(rh, pk->N, d1, d0);                       // R_H mod n^2 = d0 + d1 n
        pi.c_one_pair = pk->mn2.add_const(d0 + hostbig::shl(d1, (size_t)LB * H));   // limbs 0..H-1 = d0, H..2H-1 = d1
      }
      pi.dinv = pk->ninv2k.d;
      pi.n_limbs = pk->n_limbs.d;
      // the eight-lane variant: digits of h8 = 76 limbs (74 padded to four slices of 19), Montgomery radix R_76; a number enters
      // with one product by the pair digits of R_76^2 R_74^-1 and leaves with one by those of R_74
      const int h8 = (H + 3) / 4 * 4;
      if (H % 2 == 0 && vm_asm_available(h8 / 4, 96)) {
        std::vector<uint32_t> pc8 = make_pair_consts(pk->N, h8);
        pc8.push_back(0);
        pk->pairn_consts8.w = (int)pc8.size();
        HIPCHK(hipMalloc((void**)&pk->pairn_consts8.d, pc8.size() * 4));
        HIPCHK(hipMemcpy(pk->pairn_consts8.d, pc8.data(), pc8.size() * 4, hipMemcpyHostToDevice));
        const BigU r8 = hostbig::shl(BigU(1), (size_t)LB * h8) % n2;
        BigU rh_inv;
        if (hostbig::modinv(rh, n2, rh_inv)) {
          const BigU vals[3] = {hostbig::mulmod(hostbig::mulmod(r8, r8, n2), rh_inv, n2), rh, r8};
          std::vector<uint32_t> tc;
          for (const BigU& v : vals) {
            BigU d1, d0;
            hostbig::divmod(v, pk->N, d1, d0);
            auto l0 = d0.to_limbs(LB, (size_t)h8), l1 = d1.to_limbs(LB, (size_t)h8);
            tc.insert(tc.end(), l0.begin(), l0.end());
            tc.insert(tc.end(), l1.begin(), l1.end());
          }
          pk->pairn_tconsts8.w = (int)tc.size();
          HIPCHK(hipMalloc((void**)&pk->pairn_tconsts8.d, tc.size() * 4));
          HIPCHK(hipMemcpy(pk->pairn_tconsts8.d, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
          pi.h8 = h8;
          pi.consts8 = pk->pairn_consts8.d;
          pi.tconsts8 = pk->pairn_tconsts8.d;
        }
      }
    }
    const bool one_lane_digit = pk->mn.K == 1 && vm_asm_available(pk->mn.WT, 48);
    const bool two_lane_digit = pk->mn.WT % 2 == 0 && vm_asm_available(pk->mn.WT / 2, 112);     // 3072-bit keys: digits of 110 limbs
    if (pk->mn3 && (one_lane_digit || two_lane_digit) && (size_t)LB * pk->mn3->WT >= n3.bit_length() + 3) {
      setup_triple(*pk->mn3, pk->mn, pk->mn2, pk->triple_kconsts, pk->triple_tconsts, pk->ninv2k.d, pk->ninv2k_2.d, pk->n_limbs.d,
                   pk->n2_limbs.d);
      pk->mn3->triple.lanes6_only = !one_lane_digit;
    }
    pk->mn.upload();
    pk->mn2.upload();
  });
  if (rc == PGPU_OK) *out = pk.release();
  return rc;
}
void pgpu_pubkey_destroy(pgpu_pubkey* pk) { delete pk; }

size_t pgpu_pubkey_plain_bytes(const pgpu_pubkey* pk, int level) {
  if (!pk) return 0;
  return level == PGPU_LEVEL_TWO ? pk->mn2.nbytes : pk->mn.nbytes;
}
size_t pgpu_pubkey_cipher_bytes(const pgpu_pubkey* pk, int level) {
  if (!pk) return 0;
  if (level == PGPU_LEVEL_TWO) return pk->mn3 ? pk->mn3->nbytes : 0;
  return pk->mn2.nbytes;
}

int pgpu_seckey_create(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint8_t* lambda_be, size_t lambda_len,
                       pgpu_seckey** out) {
  if (!ctx || !pk || !lambda_be || !out) return fail(PGPU_ERR_INVALID, "null argument");
  if (ctx != pk->ctx) return fail(PGPU_ERR_INVALID, "the secret key must live on its public key's context (device pointers are shared)");
  std::unique_ptr<pgpu_seckey> sk(new pgpu_seckey());
  int rc = guarded([&] {
    ctx->bind();
    sk->ctx = ctx;
    sk->pk = pk;
    sk->lambda = BigU::from_be(lambda_be, lambda_len);
    const BigU& n = pk->N;
    if (sk->lambda.is_zero()) api_throw(PGPU_ERR_INVALID, "lambda is zero");
    // generic-path constants (paillier.go:298: mu = lambda^-1 mod n)
    BigU mu;
    if (!hostbig::modinv(sk->lambda, n, mu)) api_throw(PGPU_ERR_NOT_INVERTIBLE, "lambda is not invertible mod n");
    sk->smn.init(ctx, n);
    sk->smn2.init(ctx, n * n);
    sk->c_muR = sk->smn.add_const(sk->smn.to_mont(mu));
    sk->smn.upload();
    sk->n_minus_mu.set((n - mu) % n, pk->mn.WT);
    {
      BigU mu2, n2v = n * n;
      if (hostbig::modinv(sk->lambda, n2v, mu2)) sk->c_mu2R = sk->smn2.add_const(sk->smn2.to_mont(mu2));
      sk->smn2.upload();
    }
    // recover p, q from n and lambda = (p-1)(q-1): p + q = n - lambda + 1
    if (hostbig::cmp(n + BigU(1), sk->lambda) > 0) {
      BigU s = n + BigU(1) - sk->lambda;
      BigU s2 = s * s, n4 = hostbig::shl(n, 2);
      if (hostbig::cmp(s2, n4) >= 0) {
        BigU d = hostbig::isqrt(s2 - n4);
        if (d * d == s2 - n4 && !((s + d).is_odd())) {
          BigU p = hostbig::shr(s + d, 1), q = hostbig::shr(s - d, 1);
          if (p * q == n && p.is_odd() && q.is_odd() && !(p == q) && !(q == BigU(1))) {
            sk->p = p;
            sk->q = q;
            sk->has_crt = true;
          }
        }
      }
    }
    if (sk->has_crt) {
      const BigU &p = sk->p, &q = sk->q;
      sk->mp.init(ctx, p);
      sk->mq.init(ctx, q);
      sk->mp2.init(ctx, p * p);
      sk->mq2.init(ctx, q * q);
      if (sk->mp.WL != sk->mq.WL || sk->mp.K != sk->mq.K || sk->mp2.WL != sk->mq2.WL || sk->mp2.K != sk->mq2.K) {
        sk->has_crt = false;  // unbalanced primes: the two CRT halves would need different kernels
      } else {
        // hp = L_p((1+n)^(p-1) mod p^2)^-1 mod p = ((p-1) q)^-1 mod p, same for q
        BigU hp, hq, pinv;
        BigU p1q = hostbig::mulmod(p - BigU(1), q % p, p), q1p = hostbig::mulmod(q - BigU(1), p % q, q);
        if (!hostbig::modinv(p1q, p, hp) || !hostbig::modinv(q1p, q, hq) || !hostbig::modinv(p % q, q, pinv))
          api_throw(PGPU_ERR_INVALID, "CRT constants not invertible");
        sk->c_hpR = sk->mp.add_const(sk->mp.to_mont(hp));
        sk->c_hqR = sk->mq.add_const(sk->mq.to_mont(hq));
        sk->c_pinvR = sk->mq.add_const(sk->mq.to_mont(pinv));
        sk->mp.upload();
        sk->mq.upload();
        sk->mp2.upload();
        sk->mq2.upload();
        sk->pinv2k.set(inv_mod_pow2(p, (size_t)LB * sk->mp.WT), sk->mp.WT);
        sk->qinv2k.set(inv_mod_pow2(q, (size_t)LB * sk->mq.WT), sk->mq.WT);
        sk->p_limbs.set(p, sk->mp.WT);
        sk->q_limbs1.set(q, sk->mq.WT);
        const int pair_tag = vm_asm_available(sk->mp.WT, 16) ? 16 : 32;   // one lane per number (GenP: 37 limbs, GenP2: 55)
        if (sk->mp.K == 1 && sk->mp2.WT == 2 * sk->mp.WT && vm_asm_available(sk->mp.WT, pair_tag)) {
          sk->pair_lanes = pair_tag == 16 ? 1 : 2;
          sk->pair_small2 = sk->pair_lanes == 1 && vm_asm_available(sk->mp.WT, 32);   // two-lane variant for small batches
          const int H = sk->mp.WT;
          auto pair_consts = [&](const BigU& pr) { return make_pair_consts(pr, H); };
          std::vector<uint32_t> vp = pair_consts(p), vq = pair_consts(q);
          vp.push_back(0);   // the two-lane kernel prefetches one word past Cadj
          vq.push_back(0);
          auto put = [&](DevLimbs& d, const std::vector<uint32_t>& v) {
            d.w = (int)v.size();
            HIPCHK(hipMalloc((void**)&d.d, v.size() * 4));
            HIPCHK(hipMemcpy(d.d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
          };
          put(sk->pair_p, vp);
          put(sk->pair_q, vq);
          const BigU RH = hostbig::shl(BigU(1), (size_t)LB * H);
          sk->c_rh_p2 = sk->mp2.add_const(RH % sk->mp2.N);
          sk->c_rh_q2 = sk->mq2.add_const(RH % sk->mq2.N);
          auto pair_const = [&](ModCtx& m2, const BigU& pr, const BigU& v) {     // digits of v mod prime^2 as [d0 | d1]
            BigU d1, d0;
            hostbig::divmod(v % m2.N, pr, d1, d0);
            return m2.add_const(d0 + hostbig::shl(d1, (size_t)LB * H));
          };
          {
            BigU rp = RH % sk->mp2.N, rq = RH % sk->mq2.N, ap = rp, aq = rq;     // R_H^1
            sk->c_onep_p2 = pair_const(sk->mp2, p, ap);
            sk->c_onep_q2 = pair_const(sk->mq2, q, aq);
            for (int k2 = 0; k2 < 4; ++k2) {
              ap = hostbig::mulmod(ap, rp, sk->mp2.N);                            // R_H^(k2+2)
              aq = hostbig::mulmod(aq, rq, sk->mq2.N);
              sk->c_pk_p2[k2] = pair_const(sk->mp2, p, ap);
              sk->c_pk_q2[k2] = pair_const(sk->mq2, q, aq);
            }
          }
          sk->mp2.upload();
          sk->mq2.upload();
          sk->has_pair = true;
        }
        if (pk->mn3 && sk->c_mu2R >= 0) {
          const BigU p2 = p * p, q2 = q * q;
          sk->mp3.init(ctx, p2 * p);
          sk->mq3.init(ctx, q2 * q);
          BigU qinv, pinvq, inv2p, inv2q, hp2, hq2, p2inv;
          const BigU two(2);
          if (sk->mp3.WL == sk->mq3.WL && sk->mp3.K == sk->mq3.K && hostbig::modinv(q % p, p, qinv) &&
              hostbig::modinv(p % q, q, pinvq) && hostbig::modinv(two, p, inv2p) && hostbig::modinv(two, q, inv2q) &&
              hostbig::modinv(hostbig::mulmod(q % p2, p - BigU(1), p2), p2, hp2) &&
              hostbig::modinv(hostbig::mulmod(p % q2, q - BigU(1), q2), q2, hq2) && hostbig::modinv(p2 % q2, q2, p2inv)) {
            sk->c_qinv_p = sk->mp.add_const(qinv);
            sk->c_pinv_q = sk->mq.add_const(pinvq);
            sk->c_inv2R_p = sk->mp.add_const(sk->mp.to_mont(inv2p));
            sk->c_inv2R_q = sk->mq.add_const(sk->mq.to_mont(inv2q));
            sk->c_q2R = sk->mp2.add_const(sk->mp2.to_mont(q2));
            sk->c_p2R = sk->mq2.add_const(sk->mq2.to_mont(p2));
            sk->c_hp2R = sk->mp2.add_const(sk->mp2.to_mont(hp2));
            sk->c_hq2R = sk->mq2.add_const(sk->mq2.to_mont(hq2));
            sk->c_p2invR = sk->mq2.add_const(sk->mq2.to_mont(p2inv));
            {
              BigU p3inv;
              const BigU p3 = p2 * p, q3 = q2 * q;
              if (hostbig::modinv(p3 % q3, q3, p3inv)) {
                sk->c_p3invR = sk->mq3.add_const(sk->mq3.to_mont(p3inv));
                sk->p3_limbs.set(p3, sk->mp3.WT);
              }
            }
            sk->mp.upload();
            sk->mq.upload();
            sk->mp2.upload();
            sk->mq2.upload();
            sk->mp3.upload();
            sk->mq3.upload();
            sk->pinv2k_2.set(inv_mod_pow2(p, (size_t)LB * sk->mp2.WT), sk->mp2.WT);
            sk->qinv2k_2.set(inv_mod_pow2(q, (size_t)LB * sk->mq2.WT), sk->mq2.WT);
            sk->q_limbs.set(q, sk->mq.WT);
            sk->p2_limbs.set(p2, sk->mp2.WT);
            sk->q2_limbs.set(q2, sk->mq2.WT);
            sk->has_crt2 = true;
            // ladders modulo p^3 / q^3 (the DDLEQ prover, level-two CRT) in three-digit form: digits modulo the prime
            if (sk->mp.K == 1 && sk->mq.K == 1 && sk->mp.WT == sk->mq.WT && vm_asm_available(sk->mp.WT, 48) &&
                (size_t)LB * sk->mp3.WT >= sk->mp3.nbits + 3 && (size_t)LB * sk->mq3.WT >= sk->mq3.nbits + 3) {
              setup_triple(sk->mp3, sk->mp, sk->mp2, sk->tkc_p, sk->ttc_p, sk->pinv2k.d, sk->pinv2k_2.d, sk->p_limbs.d, sk->p2_limbs.d);
              setup_triple(sk->mq3, sk->mq, sk->mq2, sk->tkc_q, sk->ttc_q, sk->qinv2k.d, sk->qinv2k_2.d, sk->q_limbs.d, sk->q2_limbs.d);
              sk->eo_p.init(ctx, p);
              sk->eo_q.init(ctx, q);
            }
          }
        }
      }
    }
  });
  if (rc == PGPU_OK) *out = sk.release();
  return rc;
}
void pgpu_seckey_destroy(pgpu_seckey* sk) {
  if (!sk) return;
  if (ctx_alive(sk->ctx)) {   // (a context destroyed earlier wiped its workspace itself)
    (void)hipSetDevice(sk->ctx->device);
    sk->ctx->wipe_ws();       // the workspace may still hold this key's ladder programs and intermediate residues
  }
  delete sk;
}
int pgpu_seckey_has_crt(const pgpu_seckey* sk) { return sk && sk->has_crt; }

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Paillier batch operations
// ------------------------------------------------------------------------------------------------
namespace {

const ModCtx& cipher_mod(const pgpu_pubkey* pk, int level) {
  if (level == PGPU_LEVEL_ONE) return pk->mn2;
  if (level == PGPU_LEVEL_TWO) {
    if (!pk->mn3) api_throw(PGPU_ERR_UNSUPPORTED, "n^3 is wider than the built kernels");
    return *pk->mn3;
  }
  api_throw(PGPU_ERR_INVALID, "bad encryption level");
}

// Level-one decryption, CRT over p^2 and q^2.  c: device array of 2*WT2 limbs (WT2 = mp2.WT) per number.
// Returns device array of mn.WT-limb plaintexts; status bits are OR-ed into d_status.
uint32_t* decrypt1_crt(const pgpu_seckey* sk, const uint32_t* c_limbs, size_t nb, size_t count, int32_t* d_status) {
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx &mp = sk->mp, &mq = sk->mq, &mp2 = sk->mp2, &mq2 = sk->mq2;
  const int W2 = mp2.WT, W1 = mp.WT;
  const size_t S2 = (size_t)W2 * nb, S1 = (size_t)W1 * nb;
  // big VM memory: slots 0,1 = c (lo, hi); P: tmp 2, out 3, table 4..35; Q: tmp 36, out 37, table 38..69; 70..73 chunks of c
  uint32_t* mem = ctx->ws_t<uint32_t>(S2 * 74);
  bool pair_done = false;
  uint32_t *up = mem + 3 * S2, *uq = mem + 37 * S2;
  // (the two-lane kernel needs 2 lanes x 2 halves per ciphertext to fill the chip; below that the finer slicings win)
  const size_t lanes_target = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
  // one lane per number (GenP) from one wave per SIMD upwards; two lanes per number (GenQ) from there down to one wave
  // per SIMD again; below that the ordinary kernels with their finer slicings
  // two lanes per number while they leave every wave a SIMD of its own (both halves: 4 nb lanes); between half a wave and one
  // wave per SIMD at one lane the two-lane kernel would put two waves on most SIMDs: 16.9 ms against 14.5 ms for 20 480 ...
  // 30 720 ciphertexts (tools/decrypt_lanes_probe.py)
  const int pair_lanes_now = (sk->pair_lanes == 1 && sk->pair_small2 && nb * 4 <= lanes_target) ? 2 : sk->pair_lanes;
  // (a two-lane digit pass is 2 H^2 multiplies per lane: shorter than any slicing of the 2H-limb kernels for H <= 55, so it
  // also wins when the batch is latency-bound; for H = 74 the 4-lane slicing has the same length and fills the chip better)
  if (sk->has_pair && ctx->use_asm && ctx->use_pair && (pair_lanes_now == 1 || W1 <= 55 || nb * 4 >= lanes_target)) {
    // The whole ladder runs on the pair kernel (residues mod p^2 as two base-p digits: 58 % of the multiplies of a
    // squaring), entry and exit included:
    //   entry  c = sum_k c_k R_H^k (four H-limb chunks): the pair (c_k, 0) times the pair form of R_H^(k+2) is c_k R_H^k in
    //          pair form; the lazy sums are normalised by a product with the pair form of 1;
    //   exit   one product with the pair (1, 0): digit 0 becomes F mod p (= 1 for every unit) and digit 1 becomes
    //          (F1 + Cadj - m'') R^-1 = (F - 1)/p mod p = L_p -- Paillier's L function falls out of the last Montgomery step.
    pair_done = true;
    const uint32_t CH = 70;
    launch_copy_chunks(c_limbs, W1, 4, mem + (size_t)CH * S2, S2, W2, nb, ctx->stream);
    {
      auto entry = [&](Prog& pr, const int* ck, int onep, uint32_t acc) {
        pr.op(VM_LOAD, CH); pr.op(VM_MULC, (uint32_t)ck[0]); pr.op(VM_STORE, acc);
        for (uint32_t k2 = 1; k2 < 4; ++k2) {
          pr.op(VM_LOAD, CH + k2); pr.op(VM_MULC, (uint32_t)ck[k2]); pr.op(VM_ADD, acc);
          pr.op(VM_MULC, (uint32_t)onep); pr.op(VM_STORE, acc);
        }
      };
      Prog pp, pq;
      entry(pp, sk->c_pk_p2, sk->c_onep_p2, 2);
      emit_modexp_shared(pp, sk->p - BigU(1), 2, NO_SLOT, 2, 3, 4, NO_SLOT, true, true);
      pp.op(VM_LOAD, 3); pp.op(VM_MULC, C_ONE); pp.op(VM_STORE, 3);
      pp.end();
      entry(pq, sk->c_pk_q2, sk->c_onep_q2, 36);
      emit_modexp_shared(pq, sk->q - BigU(1), 36, NO_SLOT, 36, 37, 38, NO_SLOT, true, true);
      pq.op(VM_LOAD, 37); pq.op(VM_MULC, C_ONE); pq.op(VM_STORE, 37);
      pq.end();
      SegSpec sp{&mp2, &pp, mem, nullptr}, sq{&mq2, &pq, mem, nullptr};
      sp.pair = sk->pair_p.d; sp.pair_n0inv = mp.n0inv; sp.pair_h = W1; sp.pair_lanes = pair_lanes_now;
      sq.pair = sk->pair_q.d; sq.pair_n0inv = mq.n0inv; sq.pair_h = W1; sq.pair_lanes = pair_lanes_now;
      run_vm(ctx, nb, sp, &sq, true);
    }
  } else {
    HIPCHK(hipMemcpyAsync(mem, c_limbs, S2 * 2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    Prog pp, pq;
    emit_modexp_shared(pp, sk->p - BigU(1), 0, 1, 2, 3, 4, NO_SLOT, true);
    pp.end();
    emit_modexp_shared(pq, sk->q - BigU(1), 0, 1, 36, 37, 38, NO_SLOT, true);
    pq.end();
    SegSpec sp{&mp2, &pp, mem, nullptr}, sq{&mq2, &pq, mem, nullptr};
    run_vm(ctx, nb, sp, &sq, true);
  }
  // small memory: slots 0 Lp, 1 Lq, 2 mp, 3 mq, 4 B, 5 A, 6 h
  uint32_t* m1 = ctx->ws_t<uint32_t>(S1 * 7);
  if (pair_done) {
    // out slots hold (F mod prime | L): a unit has first digit exactly 1; L is lazy below 2 prime
    launch_flag_not_one(up, W1, nb, count, d_status, PGPU_LANE_NONUNIT, ctx->stream);
    launch_flag_not_one(uq, W1, nb, count, d_status, PGPU_LANE_NONUNIT, ctx->stream);
    launch_canon(up + S1, mp.d_nmod, W1, nb, ctx->stream);
    launch_canon(uq + S1, mq.d_nmod, W1, nb, ctx->stream);
    HIPCHK(hipMemcpyAsync(m1 + 0 * S1, up + S1, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(m1 + 1 * S1, uq + S1, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    launch_canon(up, mp2.d_nmod, W2, nb, ctx->stream);
    launch_canon(uq, mq2.d_nmod, W2, nb, ctx->stream);
    uint32_t* tb = ctx->ws_t<uint32_t>(S2);
    launch_div_exact(up, W2, 1, nullptr, 0, tb, sk->pinv2k.d, mp.d_nmod, W1, m1 + 0 * S1, W1, nb, count, d_status,
                     PGPU_LANE_NONUNIT, ctx->stream);
    launch_div_exact(uq, W2, 1, nullptr, 0, tb, sk->qinv2k.d, mq.d_nmod, W1, m1 + 1 * S1, W1, nb, count, d_status,
                     PGPU_LANE_NONUNIT, ctx->stream);
  }
  Prog a, b, c;
  a.op(VM_LOAD, 0); a.op(VM_MULC, (uint32_t)sk->c_hpR); a.op(VM_STORE, 2); a.end();   // m_p = L_p * h_p mod p
  b.op(VM_LOAD, 1); b.op(VM_MULC, (uint32_t)sk->c_hqR); b.op(VM_STORE, 3); b.end();   // m_q = L_q * h_q mod q
  SegSpec sa{&mp, &a, m1, nullptr}, sb{&mq, &b, m1, nullptr};
  run_vm(ctx, nb, sa, &sb, false);
  // Garner needs ONE integer m_p in both places it is used (B below and the final sum): canonicalise it
  // first.  (A lazy m_p in [p, 2p) here and a reduced one in the sum gave m - p on ~1e-4 of the lanes.)
  launch_canon(m1 + 2 * S1, mp.d_nmod, W1, nb, ctx->stream);
  c.op(VM_LOAD, 2); c.op(VM_MULC, (uint32_t)sk->c_pinvR); c.op(VM_STORE, 4);           // B = m_p * p^-1 mod q
  c.op(VM_LOAD, 3); c.op(VM_MULC, (uint32_t)sk->c_pinvR); c.op(VM_STORE, 5);           // A = m_q * p^-1 mod q
  c.end();
  SegSpec sc{&mq, &c, m1, nullptr};
  run_vm(ctx, nb, sc, nullptr, false);
  launch_canon(m1 + 4 * S1, mq.d_nmod, W1, nb, ctx->stream);
  launch_canon(m1 + 5 * S1, mq.d_nmod, W1, nb, ctx->stream);
  launch_sub_mod(m1 + 5 * S1, m1 + 4 * S1, mq.d_nmod, m1 + 6 * S1, W1, nb, ctx->stream);  // h = A - B mod q
  const int WN = sk->pk->mn.WT;
  uint32_t* res = ctx->ws_t<uint32_t>((size_t)WN * nb);
  // m = m_p + p * h
  launch_mul_const_add(m1 + 6 * S1, W1, sk->p_limbs.d, W1, m1 + 2 * S1, W1, 0, res, WN, nb, ctx->stream);
  return res;
}


// base^e mod n for a holder of the factorisation: ladders modulo p and q (half the width, exponents modulo p - 1 and
// q - 1) in one two-segment launch, then Garner as in decrypt1_crt.  base: canonical, mn.WT limbs.  Returns mn.WT limbs.
uint32_t* pow_n_crt(const pgpu_seckey* sk, const uint32_t* base, const BigU& e, size_t nb) {
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx &mp = sk->mp, &mq = sk->mq;
  const int W1 = mp.WT, WN = sk->pk->mn.WT;
  const size_t S1 = (size_t)W1 * nb;
  // slots per half: 0 in, 2 tmp, 3 out, 5..36 table
  uint32_t *memp = ctx->ws_t<uint32_t>(S1 * 37), *memq = ctx->ws_t<uint32_t>(S1 * 37);
  reduce_mod(ctx, mp, base, WN, memp, nb);
  reduce_mod(ctx, mq, base, WN, memq, nb);
  auto half_exp = [&](const BigU& pr) {
    const BigU ord = pr - BigU(1);
    BigU r = e % ord;
    if (r.is_zero() && !e.is_zero()) r = ord;       // 0^e stays 0 for a base that is a multiple of the prime
    return r;
  };
  Prog pp, pq;
  emit_modexp_shared(pp, half_exp(sk->p), 0, NO_SLOT, 2, 3, 5, NO_SLOT, true);
  pp.end();
  emit_modexp_shared(pq, half_exp(sk->q), 0, NO_SLOT, 2, 3, 5, NO_SLOT, true);
  pq.end();
  SegSpec sp{&mp, &pp, memp, nullptr}, sq{&mq, &pq, memq, nullptr};
  run_vm(ctx, nb, sp, &sq, true);
  // small memory: 2 x_p, 3 x_q, 4 B, 5 A, 6 h
  uint32_t* m1 = ctx->ws_t<uint32_t>(S1 * 7);
  HIPCHK(hipMemcpyAsync(m1 + 2 * S1, memp + 3 * S1, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(m1 + 3 * S1, memq + 3 * S1, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  launch_canon(m1 + 2 * S1, mp.d_nmod, W1, nb, ctx->stream);     // one integer x_p in both places it is used
  launch_canon(m1 + 3 * S1, mq.d_nmod, W1, nb, ctx->stream);
  Prog c;
  c.op(VM_LOAD, 2); c.op(VM_MULC, (uint32_t)sk->c_pinvR); c.op(VM_STORE, 4);           // B = x_p * p^-1 mod q
  c.op(VM_LOAD, 3); c.op(VM_MULC, (uint32_t)sk->c_pinvR); c.op(VM_STORE, 5);           // A = x_q * p^-1 mod q
  c.end();
  SegSpec sc{&mq, &c, m1, nullptr};
  run_vm(ctx, nb, sc, nullptr, false);
  launch_canon(m1 + 4 * S1, mq.d_nmod, W1, nb, ctx->stream);
  launch_canon(m1 + 5 * S1, mq.d_nmod, W1, nb, ctx->stream);
  launch_sub_mod(m1 + 5 * S1, m1 + 4 * S1, mq.d_nmod, m1 + 6 * S1, W1, nb, ctx->stream);  // h = A - B mod q
  uint32_t* res = ctx->ws_t<uint32_t>((size_t)WN * nb);
  launch_mul_const_add(m1 + 6 * S1, W1, sk->p_limbs.d, W1, m1 + 2 * S1, W1, 0, res, WN, nb, ctx->stream);   // x_p + p h
  return res;
}

// base^e mod n^2 for a holder of the factorisation (base < n, canonical, mn.WT limbs): both halves -- modulo p^2 and q^2,
// exponents modulo p (p-1) and q (q-1) -- in pair form on the pair kernel in ONE two-segment launch (the ladder of the
// headline Decrypt with another exponent), then Garner in Z_{q^2}.  Returns mn2.WT canonical limbs.
bool pow_n2_crt_usable(const pgpu_seckey* sk) {
  pgpu_ctx* ctx = sk->ctx;
  return sk->has_pair && sk->has_crt2 && sk->c_rh_p2 >= 0 && sk->c_p2invR >= 0 && ctx->use_asm && ctx->use_pair &&
         sk->pk->mn.WT == sk->mp2.WT && sk->pk->mn2.WT == 2 * sk->mp2.WT;
}
uint32_t* pow_n2_crt(const pgpu_seckey* sk, const uint32_t* base, const BigU& e, size_t nb) {
  pgpu_ctx* ctx = sk->ctx;
  const int H = sk->mp.WT, W2 = sk->mp2.WT, WN2 = sk->pk->mn2.WT;
  const size_t S1 = (size_t)H * nb, S2 = (size_t)W2 * nb;
  uint32_t* xh[2];
  uint32_t* mem[2];
  Prog lad[2];
  Fork in(ctx);                                                         // the q-half's entry chain beside the p-half's
  for (int half = 0; half < 2; ++half) {
    in.chain(half);
    const ModCtx &m1 = half ? sk->mq : sk->mp, &m2 = half ? sk->mq2 : sk->mp2;
    const BigU& pr = half ? sk->q : sk->p;
    // slots (W2 limbs): 0 x, 2 pair form in, 3 out, 5..36 table
    uint32_t* mm = mem[half] = ctx->ws_t<uint32_t>(S2 * 37);
    reduce_mod(ctx, m2, base, sk->pk->mn.WT, mm, nb);
    Prog a;                                                             // X = x R_H mod prime^2
    a.op(VM_LOAD, 0); a.op(VM_MULC, C_R2); a.op(VM_MULC, (uint32_t)(half ? sk->c_rh_q2 : sk->c_rh_p2)); a.op(VM_STORE, 3); a.end();
    SegSpec sa{&m2, &a, mm, nullptr};
    run_vm(ctx, nb, sa, nullptr, false);
    launch_canon(mm + 3 * S2, m2.d_nmod, W2, nb, ctx->stream);
    uint32_t* x0 = ctx->ws_t<uint32_t>(S1);                             // digits X = X0 + X1 prime -> slot 2
    uint32_t* tb = ctx->ws_t<uint32_t>(S2);
    reduce_mod(ctx, m1, mm + 3 * S2, W2, x0, nb);
    launch_div_exact(mm + 3 * S2, W2, 0, x0, H, tb, (half ? sk->qinv2k : sk->pinv2k).d, m1.d_nmod, H, mm + 2 * S2 + S1, H, nb, nb,
                     nullptr, 0, ctx->stream);
    HIPCHK(hipMemcpyAsync(mm + 2 * S2, x0, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    const BigU ord = pr * (pr - BigU(1));
    BigU eh = e;
    if (!(e < ord)) {
      eh = e % ord;
      if (eh < BigU(2)) eh = eh + ord;     // x^e = 0 modulo prime^2 for a multiple of the prime and e >= 2: keep it so
    }
    if (eh.bit_length() < 64) eh = e;
    emit_modexp_shared(lad[half], eh, 2, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    lad[half].end();
  }
  in.join();
  {
    // (small batches on two lanes per number, as Decrypt chooses: a squaring is 37 rows of 74 multiplies instead of the one-lane
    // kernel's 4 810 in a row -- the ladder's latency is the run time there)
    const size_t lanes_target = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
    const int lanes = (sk->pair_lanes == 1 && sk->pair_small2 && nb * 4 <= lanes_target) ? 2 : sk->pair_lanes;
    SegSpec sp{&sk->mp2, &lad[0], mem[0], nullptr}, sq{&sk->mq2, &lad[1], mem[1], nullptr};
    sp.pair = sk->pair_p.d; sp.pair_n0inv = sk->mp.n0inv; sp.pair_h = H; sp.pair_lanes = lanes;
    sq.pair = sk->pair_q.d; sq.pair_n0inv = sk->mq.n0inv; sq.pair_h = H; sq.pair_lanes = lanes;
    run_vm(ctx, nb, sp, &sq, true);
  }
  Fork out(ctx);
  for (int half = 0; half < 2; ++half) {
    out.chain(half);
    const ModCtx& m2 = half ? sk->mq2 : sk->mp2;
    uint32_t* mm = mem[half];
    // F~ = F0 + F1 prime, then out of pair and Montgomery form
    launch_mul_const_add(mm + 3 * S2 + S1, H, (half ? sk->q_limbs1 : sk->p_limbs).d, H, mm + 3 * S2, H, 0, mm + 2 * S2, W2, nb, ctx->stream);
    Prog a;
    a.op(VM_LOAD, 2); a.op(VM_MULC, (uint32_t)(half ? sk->c_rh_q2 : sk->c_rh_p2)); a.op(VM_STORE, 3); a.end();
    SegSpec sa{&m2, &a, mm, nullptr};
    run_vm(ctx, nb, sa, nullptr, false);
    launch_canon(mm + 3 * S2, m2.d_nmod, W2, nb, ctx->stream);
    xh[half] = mm + 3 * S2;
  }
  out.join();
  // Garner: x = x_p + p^2 ((x_q - x_p) p^-2 mod q^2); slots: 0 x_p, 1 x_q, 2 B, 3 A, 4 h
  uint32_t* g = ctx->ws_t<uint32_t>(S2 * 5);
  HIPCHK(hipMemcpyAsync(g, xh[0], S2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(g + S2, xh[1], S2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  Prog c;
  c.op(VM_LOAD, 0); c.op(VM_MULC, (uint32_t)sk->c_p2invR); c.op(VM_STORE, 2);
  c.op(VM_LOAD, 1); c.op(VM_MULC, (uint32_t)sk->c_p2invR); c.op(VM_STORE, 3);
  c.end();
  SegSpec sc{&sk->mq2, &c, g, nullptr};
  run_vm(ctx, nb, sc, nullptr, false);
  launch_canon(g + 2 * S2, sk->mq2.d_nmod, W2, nb, ctx->stream);
  launch_canon(g + 3 * S2, sk->mq2.d_nmod, W2, nb, ctx->stream);
  launch_sub_mod(g + 3 * S2, g + 2 * S2, sk->mq2.d_nmod, g + 4 * S2, W2, nb, ctx->stream);
  uint32_t* res = ctx->ws_t<uint32_t>((size_t)WN2 * nb);
  launch_mul_const_add(g + 4 * S2, W2, sk->p2_limbs.d, W2, g, W2, 0, res, WN2, nb, ctx->stream);
  return res;
}

// The reference's L(u, n) = Div(u - 1, n) (paillier.go:436-440; Euclidean: floor for u >= 1 and -1 for u = 0) for a
// canonical u of `wu` limbs: floor((u-1)/n) = ((u-1) - ((u-1) mod n)) / n -- Montgomery reductions mod n plus one
// exact division.  Returns the quotient (wq limbs: mn.WT for u < n^2, mn2.WT for u < n^3); zf[g] = (u == 0), for which
// the caller substitutes the value that stands for -1.
uint32_t* L_floor(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint32_t* u, int wu, size_t nb, size_t count, int wq,
                  int32_t* zf) {
  const ModCtx &mn = pk->mn, &mn2 = pk->mn2;
  const int W1 = mn.WT, W2 = mn2.WT;
  launch_is_zero(u, wu, nb, zf, ctx->stream);
  uint32_t* v = ctx->ws_t<uint32_t>((size_t)wu * nb);
  launch_sub_one(u, v, wu, nb, ctx->stream);
  uint32_t* r = ctx->ws_t<uint32_t>((size_t)W1 * nb);
  if (wu <= 2 * W1) {
    reduce_mod(ctx, mn, v, wu, r, nb);
  } else {
    if (wu > 2 * W2) api_throw(PGPU_ERR_UNSUPPORTED, "L_floor operand too wide");
    uint32_t* r2 = ctx->ws_t<uint32_t>((size_t)W2 * nb);
    reduce_mod(ctx, mn2, v, wu, r2, nb);     // (u-1) mod n^2
    reduce_mod(ctx, mn, r2, W2, r, nb);      // ... mod n
  }
  uint32_t* q = ctx->ws_t<uint32_t>((size_t)wq * nb);
  uint32_t* tb = ctx->ws_t<uint32_t>((size_t)wu * nb);
  int32_t* st_dummy = ctx->ws_t<int32_t>(nb);
  HIPCHK(hipMemsetAsync(st_dummy, 0, nb * 4, ctx->stream));
  const uint32_t* dinv = (wq == W1) ? pk->ninv2k.d : pk->ninv2k_2.d;
  if (wq != W1 && wq != W2) api_throw(PGPU_ERR_INVALID, "L_floor quotient width");
  launch_div_exact(v, wu, 0, r, W1, tb, dinv, mn.d_nmod, W1, q, wq, nb, count, st_dummy, 2, ctx->stream);
  return q;
}

// m = L(u) * C mod n for canonical u < n^2.  c_const = index of C*R mod n in pk->mn.consts; neg_const = (-C) mod n as
// limbs (the u = 0 answer: L = -1).  Returns mn.WT-limb canonical results.
uint32_t* L_times_const(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint32_t* u, size_t nb, size_t count, const ModCtx& mn,
                        int c_const, const uint32_t* neg_const) {
  const int W1 = mn.WT;
  int32_t* zf = ctx->ws_t<int32_t>(nb);
  uint32_t* q = L_floor(ctx, pk, u, pk->mn2.WT, nb, count, W1, zf);
  size_t s1 = (size_t)W1 * nb;
  uint32_t* m1 = ctx->ws_t<uint32_t>(s1 * 2);
  HIPCHK(hipMemcpyAsync(m1, q, s1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  Prog p;
  p.op(VM_LOAD, 0);
  p.op(VM_MULC, (uint32_t)c_const);
  p.op(VM_STORE, 1);
  p.end();
  SegSpec sg{&mn, &p, m1, nullptr};
  run_vm(ctx, nb, sg, nullptr, false);
  launch_canon(m1 + s1, mn.d_nmod, W1, nb, ctx->stream);
  launch_select_const(zf, neg_const, m1 + s1, W1, nb, ctx->stream);  // u = 0: L = -1
  return m1 + s1;
}

// Level-one decryption by the reference's own formula (paillier.go:292-303), for ANY c (units or not):
//   u = c^lambda mod n^2 ; ml = L(u) ; m = ml * lambda^-1 mod n.
// c: device array of mn2.WT limbs per number.  Returns device array of mn.WT-limb plaintexts.
uint32_t* decrypt1_generic(const pgpu_seckey* sk, const uint32_t* c_limbs, size_t nb, size_t count) {
  pgpu_ctx* ctx = sk->ctx;
  const pgpu_pubkey* pk = sk->pk;
  const ModCtx& mn2 = pk->mn2;
  ModexpPlan pl = modexp_alloc(ctx, mn2, nb, 32);
  HIPCHK(hipMemcpyAsync(pl.in(), c_limbs, (size_t)mn2.WT * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
  modexp_shared_run(ctx, mn2, pl, sk->lambda, false, false, true);   // u, canonical
  return L_times_const(ctx, pk, pl.out(), nb, count, sk->smn, sk->c_muR, sk->n_minus_mu.d);
}

// Level-two (Damgard-Jurik s = 2) decryption by the reference's formula (paillier.go:292-340):
//   a = c^lambda mod n^3
//   i1 = L(a mod n^2)                                   (recoveryAlgorithm j = 1)
//   i  = (L(a) - (|i1| (i1 - 1) mod n^2) * n * 2^-1) mod n^2      (j = 2, k = 2; |i1| from SetBytes(i.Bytes()), :320)
//   m  = i * lambda^-1 mod n^2
// c: device array of mn3.WT limbs.  Returns mn2.WT-limb canonical plaintexts.
uint32_t* decrypt2_generic(const pgpu_seckey* sk, const uint32_t* c_limbs, size_t nb, size_t count) {
  pgpu_ctx* ctx = sk->ctx;
  const pgpu_pubkey* pk = sk->pk;
  if (!pk->mn3 || sk->c_mu2R < 0) api_throw(PGPU_ERR_UNSUPPORTED, "level two is not available for this key");
  const ModCtx &mn = pk->mn, &mn2 = pk->mn2, &mn3 = *pk->mn3;
  const int W1 = mn.WT, W2 = mn2.WT, W3 = mn3.WT;
  ModexpPlan pl = modexp_alloc(ctx, mn3, nb, 32);
  HIPCHK(hipMemcpyAsync(pl.in(), c_limbs, (size_t)W3 * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
  modexp_shared_run(ctx, mn3, pl, sk->lambda, false, false, true);
  const uint32_t* a = pl.out();
  // j = 1
  uint32_t* a2 = ctx->ws_t<uint32_t>((size_t)W2 * nb);
  reduce_mod(ctx, mn2, a, W3, a2, nb);
  int32_t* z1 = ctx->ws_t<int32_t>(nb);
  uint32_t* i1 = L_floor(ctx, pk, a2, W2, nb, count, W1, z1);
  // j = 2
  int32_t* z2 = ctx->ws_t<int32_t>(nb);
  uint32_t* t1 = L_floor(ctx, pk, a, W3, nb, count, W2, z2);
  const size_t s2 = (size_t)W2 * nb;
  uint32_t* n2m1 = ctx->upload_words((mn2.N - BigU(1)).to_limbs(LB, W2));
  uint32_t* n2m2 = ctx->upload_words((mn2.N - BigU(2)).to_limbs(LB, W2));
  uint32_t* one2 = ctx->upload_words(BigU(1).to_limbs(LB, W2));
  launch_select_const(z2, n2m1, t1, W2, nb, ctx->stream);                  // L(0) = -1  ->  n^2 - 1 (mod n^2)
  // slots (mod n^2): 0 |i1|, 1 (i1 - 1) mod n^2, 2 X*n/2, 3 i, 4 m
  uint32_t* mv = ctx->ws_t<uint32_t>(s2 * 5);
  launch_copy_limbs(i1, 0, W1, mv, W2, nb, ctx->stream);                     // zero-extend
  launch_select_const(z1, one2, mv, W2, nb, ctx->stream);                    // i1 = -1: |i1| = 1
  uint32_t* ones = ctx->ws_t<uint32_t>(s2);
  launch_fill_const(one2, ones, W2, nb, ctx->stream);
  launch_copy_limbs(i1, 0, W1, mv + 3 * s2, W2, nb, ctx->stream);
  launch_sub_mod(mv + 3 * s2, ones, mn2.d_nmod, mv + s2, W2, nb, ctx->stream);   // (i1 - 1) mod n^2
  launch_select_const(z1, n2m2, mv + s2, W2, nb, ctx->stream);               // i1 = -1: i1 - 1 = -2
  Prog p;
  p.op(VM_LOAD, 0); p.op(VM_MULC, C_R2); p.op(VM_MUL, 1);                    // X = |i1| (i1-1) mod n^2 (plain)
  p.op(VM_MULC, (uint32_t)pk->c_ninv2R_2);                                   // X * n * 2^-1 mod n^2
  p.op(VM_STORE, 2);
  p.end();
  SegSpec sg{&mn2, &p, mv, nullptr};
  run_vm(ctx, nb, sg, nullptr, false);
  launch_canon(mv + 2 * s2, mn2.d_nmod, W2, nb, ctx->stream);
  launch_sub_mod(t1, mv + 2 * s2, mn2.d_nmod, mv + 3 * s2, W2, nb, ctx->stream);  // i
  Prog q;
  q.op(VM_LOAD, 3); q.op(VM_MULC, (uint32_t)sk->c_mu2R); q.op(VM_STORE, 4); q.end();
  SegSpec sq{&sk->smn2, &q, mv, nullptr};
  run_vm(ctx, nb, sq, nullptr, false);
  launch_canon(mv + 4 * s2, mn2.d_nmod, W2, nb, ctx->stream);
  return mv + 4 * s2;
}

// Level-two decryption, CRT over p^3 and q^3 (the level-one idea of decrypt1_crt carried to s = 2).  For a unit c =
// (1+n)^m r^(n^2) mod n^3:  u_p = c^(p-1) mod p^3 = (1+n)^x with x = m (p-1) mod p^2 (r^(n^2 (p-1)) = 1: the group has
// order p^2 (p-1)), and (1+n)^x = 1 + x n + C(x,2) n^2 (mod p^3) because p^3 | n^3.  So with L_p(u) = (u-1)/p (exact):
//   L_p = x q + C(x,2) q^2 p  (mod p^2)   =>   x1 = L_p q^-1 mod p,   x = (L_p - C(x1,2) p q^2) q^-1  mod p^2
// (C(x,2) p mod p^2 depends on x mod p only), m mod p^2 = x (p-1)^-1, likewise mod q^2, then Garner.  Two 1.5k-bit-wide
// exponentiations with half-length exponents instead of one three times as wide: ~4x fewer limb products than
// paillier.go:292-340 and the same integers out.  c: device array of 2*mp3.WT limbs per number.  Returns mn2.WT-limb
// plaintexts; lanes where a division is not exact (c not a unit) get PGPU_LANE_NONUNIT and are redone by the caller.
uint32_t* decrypt2_crt(const pgpu_seckey* sk, const uint32_t* c_limbs, size_t nb, size_t count, int32_t* d_status) {
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx &mp = sk->mp, &mq = sk->mq, &mp2 = sk->mp2, &mq2 = sk->mq2, &mp3 = sk->mp3, &mq3 = sk->mq3;
  const int W1 = mp.WT, W2 = mp2.WT, W3 = mp3.WT;
  const size_t S1 = (size_t)W1 * nb, S2 = (size_t)W2 * nb, S3 = (size_t)W3 * nb;
  uint32_t *up, *uq;
  if (triple_usable(ctx, mp3) && triple_usable(ctx, mq3) && (sk->p - BigU(1)).bit_length() >= 64 &&
      (sk->q - BigU(1)).bit_length() >= 64) {       // (the raw ladder is the sliding-window form: toy keys keep the generic kernel)
    // c^(p-1) mod p^3 and c^(q-1) mod q^3 on the three-digit kernel (digits modulo the prime), both halves in one launch
    TriplePlan tp = triple_alloc(ctx, mp3, nb, 5 + 32), tq = triple_alloc(ctx, mq3, nb, 5 + 32);
    uint32_t* g = ctx->ws_t<uint32_t>(S3 * 3);
    reduce_mod(ctx, mp3, c_limbs, 2 * W3, g + 2 * S3, nb);
    triple_enter(ctx, mp3, g + 2 * S3, tp, 0);
    reduce_mod(ctx, mq3, c_limbs, 2 * W3, g + 2 * S3, nb);
    triple_enter(ctx, mq3, g + 2 * S3, tq, 0);
    Prog pp, pq;
    emit_modexp_shared(pp, sk->p - BigU(1), 0, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    pp.end();
    emit_modexp_shared(pq, sk->q - BigU(1), 0, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    pq.end();
    SegSpec sp{&mp3, &pp, tp.mem, nullptr}, sq{&mq3, &pq, tq.mem, nullptr};
    sp.pair = mp3.triple.kconsts; sp.pair_n0inv = mp.n0inv; sp.pair_h = tp.H; sp.pair_lanes = 3; sp.tconsts = mp3.triple.tconsts;
    sq.pair = mq3.triple.kconsts; sq.pair_n0inv = mq.n0inv; sq.pair_h = tq.H; sq.pair_lanes = 3; sq.tconsts = mq3.triple.tconsts;
    run_vm(ctx, nb, sp, &sq, true);
    up = g;
    uq = g + S3;
    triple_exit(ctx, mp3, tp, 3, up, nullptr);
    triple_exit(ctx, mq3, tq, 3, uq, nullptr);
  } else {
    uint32_t* mem = ctx->ws_t<uint32_t>(S3 * 70);   // same slot plan as decrypt1_crt
    HIPCHK(hipMemcpyAsync(mem, c_limbs, S3 * 2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    Prog pp, pq;
    emit_modexp_shared(pp, sk->p - BigU(1), 0, 1, 2, 3, 4, NO_SLOT, true);
    pp.end();
    emit_modexp_shared(pq, sk->q - BigU(1), 0, 1, 36, 37, 38, NO_SLOT, true);
    pq.end();
    SegSpec sp{&mp3, &pp, mem, nullptr}, sq{&mq3, &pq, mem, nullptr};
    run_vm(ctx, nb, sp, &sq, true);
    up = mem + 3 * S3;
    uq = mem + 37 * S3;
    launch_canon(up, mp3.d_nmod, W3, nb, ctx->stream);
    launch_canon(uq, mq3.d_nmod, W3, nb, ctx->stream);
  }
  // m2 (W2-limb slots): per side s in {0 (p), 1 (q)}: 5s+0 L, 5s+1 t*prime, 5s+2 w, 5s+3 L - w, 5s+4 m mod prime^2;
  // then 10 B, 11 A, 12 h
  uint32_t* m2 = ctx->ws_t<uint32_t>(S2 * 13);
  uint32_t* tb = ctx->ws_t<uint32_t>(S3);
  launch_div_exact(up, W3, 1, nullptr, 0, tb, sk->pinv2k_2.d, mp.d_nmod, W1, m2 + 0 * S2, W2, nb, count, d_status,
                   PGPU_LANE_NONUNIT, ctx->stream);
  launch_div_exact(uq, W3, 1, nullptr, 0, tb, sk->qinv2k_2.d, mq.d_nmod, W1, m2 + 5 * S2, W2, nb, count, d_status,
                   PGPU_LANE_NONUNIT, ctx->stream);
  // m1 (W1-limb slots): per side 6s+0 L lo, 6s+1 L hi, 6s+2 tmp, 6s+3 x1, 6s+4 x1 - 1, 6s+5 t; 12 = ones
  uint32_t* m1 = ctx->ws_t<uint32_t>(S1 * 13);
  const int whi = std::min(W2 - W1, W1);   // L < prime^2 < 2^(56 W1): limbs above 2 W1 are zero
  for (int s = 0; s < 2; ++s) {
    const uint32_t* L = m2 + (size_t)(5 * s) * S2;
    launch_copy_limbs(L, 0, W1, m1 + (size_t)(6 * s) * S1, W1, nb, ctx->stream);
    launch_copy_limbs(L, W1, whi, m1 + (size_t)(6 * s + 1) * S1, W1, nb, ctx->stream);
  }
  {
    Prog a, b;                                                             // x1 = L * other^-1 mod prime
    emit_to_mont(a, 0, 1, 2); a.op(VM_MULC, (uint32_t)sk->c_qinv_p); a.op(VM_STORE, 3); a.end();
    emit_to_mont(b, 6, 7, 8); b.op(VM_MULC, (uint32_t)sk->c_pinv_q); b.op(VM_STORE, 9); b.end();
    SegSpec sa{&mp, &a, m1, nullptr}, sb{&mq, &b, m1, nullptr};
    run_vm(ctx, nb, sa, &sb, false);
  }
  launch_canon(m1 + 3 * S1, mp.d_nmod, W1, nb, ctx->stream);
  launch_canon(m1 + 9 * S1, mq.d_nmod, W1, nb, ctx->stream);
  launch_fill_const(mp.d_consts + (size_t)C_ONE * W1, m1 + 12 * S1, W1, nb, ctx->stream);
  launch_sub_mod(m1 + 3 * S1, m1 + 12 * S1, mp.d_nmod, m1 + 4 * S1, W1, nb, ctx->stream);
  launch_sub_mod(m1 + 9 * S1, m1 + 12 * S1, mq.d_nmod, m1 + 10 * S1, W1, nb, ctx->stream);
  {
    Prog a, b;                                                             // t = x1 (x1 - 1) / 2 mod prime
    a.op(VM_LOAD, 3); a.op(VM_MULC, C_R2); a.op(VM_MUL, 4); a.op(VM_MULC, (uint32_t)sk->c_inv2R_p); a.op(VM_STORE, 5); a.end();
    b.op(VM_LOAD, 9); b.op(VM_MULC, C_R2); b.op(VM_MUL, 10); b.op(VM_MULC, (uint32_t)sk->c_inv2R_q); b.op(VM_STORE, 11); b.end();
    SegSpec sa{&mp, &a, m1, nullptr}, sb{&mq, &b, m1, nullptr};
    run_vm(ctx, nb, sa, &sb, false);
  }
  launch_canon(m1 + 5 * S1, mp.d_nmod, W1, nb, ctx->stream);
  launch_canon(m1 + 11 * S1, mq.d_nmod, W1, nb, ctx->stream);
  launch_mul_const_add(m1 + 5 * S1, W1, sk->p_limbs.d, W1, nullptr, 0, 0, m2 + 1 * S2, W2, nb, ctx->stream);   // t p < p^2
  launch_mul_const_add(m1 + 11 * S1, W1, sk->q_limbs.d, W1, nullptr, 0, 0, m2 + 6 * S2, W2, nb, ctx->stream);
  {
    Prog a, b;                                                             // w = t prime other^2 mod prime^2
    a.op(VM_LOAD, 1); a.op(VM_MULC, (uint32_t)sk->c_q2R); a.op(VM_STORE, 2); a.end();
    b.op(VM_LOAD, 6); b.op(VM_MULC, (uint32_t)sk->c_p2R); b.op(VM_STORE, 7); b.end();
    SegSpec sa{&mp2, &a, m2, nullptr}, sb{&mq2, &b, m2, nullptr};
    run_vm(ctx, nb, sa, &sb, false);
  }
  launch_canon(m2 + 2 * S2, mp2.d_nmod, W2, nb, ctx->stream);
  launch_canon(m2 + 7 * S2, mq2.d_nmod, W2, nb, ctx->stream);
  launch_sub_mod(m2 + 0 * S2, m2 + 2 * S2, mp2.d_nmod, m2 + 3 * S2, W2, nb, ctx->stream);
  launch_sub_mod(m2 + 5 * S2, m2 + 7 * S2, mq2.d_nmod, m2 + 8 * S2, W2, nb, ctx->stream);
  {
    Prog a, b;                                                             // m mod prime^2 = (L - w) (other (prime-1))^-1
    a.op(VM_LOAD, 3); a.op(VM_MULC, (uint32_t)sk->c_hp2R); a.op(VM_STORE, 4); a.end();
    b.op(VM_LOAD, 8); b.op(VM_MULC, (uint32_t)sk->c_hq2R); b.op(VM_STORE, 9); b.end();
    SegSpec sa{&mp2, &a, m2, nullptr}, sb{&mq2, &b, m2, nullptr};
    run_vm(ctx, nb, sa, &sb, false);
  }
  launch_canon(m2 + 4 * S2, mp2.d_nmod, W2, nb, ctx->stream);   // ONE integer m_p2 for both uses below (see decrypt1_crt)
  {
    Prog c;                                                                // Garner in Z_{q^2}
    c.op(VM_LOAD, 4); c.op(VM_MULC, (uint32_t)sk->c_p2invR); c.op(VM_STORE, 10);
    c.op(VM_LOAD, 9); c.op(VM_MULC, (uint32_t)sk->c_p2invR); c.op(VM_STORE, 11);
    c.end();
    SegSpec sc{&mq2, &c, m2, nullptr};
    run_vm(ctx, nb, sc, nullptr, false);
  }
  launch_canon(m2 + 10 * S2, mq2.d_nmod, W2, nb, ctx->stream);
  launch_canon(m2 + 11 * S2, mq2.d_nmod, W2, nb, ctx->stream);
  launch_sub_mod(m2 + 11 * S2, m2 + 10 * S2, mq2.d_nmod, m2 + 12 * S2, W2, nb, ctx->stream);   // h = (m_q2 - m_p2) / p^2 mod q^2
  const int WN = sk->pk->mn2.WT;
  uint32_t* res = ctx->ws_t<uint32_t>((size_t)WN * nb);
  launch_mul_const_add(m2 + 12 * S2, W2, sk->p2_limbs.d, W2, m2 + 4 * S2, W2, 0, res, WN, nb, ctx->stream);   // m_p2 + p^2 h
  return res;
}

// Level-two decryption of `count` ciphertexts held as wc limbs each (wc = 2*mp3.WT when crt, else mn3.WT): CRT first,
// then the reference formula on the lanes whose ciphertext turned out not to be a unit.  hstat (host, `count` entries,
// may be null) receives the per-lane status.  Synchronises the stream when crt is set.
uint32_t* decrypt2_units_or_generic(const pgpu_seckey* sk, const uint32_t* cl3, int wc, size_t nb, size_t count, bool crt,
                                    int32_t* hstat) {
  pgpu_ctx* ctx = sk->ctx;
  const pgpu_pubkey* pk = sk->pk;
  if (!crt) return decrypt2_generic(sk, cl3, nb, count);
  const int W3 = pk->mn3->WT, W2 = pk->mn2.WT;
  int32_t* d_status = ctx->ws_t<int32_t>(nb);
  HIPCHK(hipMemsetAsync(d_status, 0, nb * 4, ctx->stream));
  uint32_t* res = decrypt2_crt(sk, cl3, nb, count, d_status);
  std::vector<int32_t> st(count);
  HIPCHK(hipMemcpyAsync(st.data(), d_status, count * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<uint32_t> idx;
  for (size_t i = 0; i < count; ++i)
    if (st[i] & PGPU_LANE_NONUNIT) idx.push_back((uint32_t)i);
  if (!idx.empty()) {
    const size_t nbg = round_up(idx.size(), VM_BLOCK);
    uint32_t* d_idx = ctx->upload_words(idx);
    uint32_t* cg = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
    launch_gather(cl3, nb, d_idx, idx.size(), cg, nbg, W3, ctx->stream);   // the low WT(n^3) limbs hold all of c
    uint32_t* rg = decrypt2_generic(sk, cg, nbg, idx.size());
    launch_scatter(rg, nbg, d_idx, idx.size(), res, nb, W2, ctx->stream);
  }
  if (hstat) memcpy(hstat, st.data(), count * 4);
  (void)wc;
  return res;
}

// ---- threshold decryption ------------------------------------------------------------------------------------

// signed host integer for the Lagrange coefficients (thresholdkey.go:91-107)
struct SBig { BigU mag; bool neg = false; };
SBig smul_small(const SBig& a, long long k) {
  SBig r;
  r.mag = a.mag * BigU((uint64_t)(k < 0 ? -k : k));
  r.neg = r.mag.is_zero() ? false : (a.neg != (k < 0));
  return r;
}
// gmp.Int.Div: Euclidean division (remainder in [0, |d|))
SBig sdiv_euclid(const SBig& n, long long d) {
  BigU ad((uint64_t)(d < 0 ? -d : d)), q0, r0;
  hostbig::divmod(n.mag, ad, q0, r0);
  SBig q;
  const bool dneg = d < 0;
  if (!n.neg) { q.mag = q0; q.neg = dneg; }
  else if (r0.is_zero()) { q.mag = q0; q.neg = !dneg; }
  else { q.mag = q0 + BigU(1); q.neg = !dneg; }
  if (q.mag.is_zero()) q.neg = false;
  return q;
}
BigU factorial_big(int n) {
  BigU r(1);
  for (int i = 1; i <= n; ++i) r = r * BigU((uint64_t)i);
  return r;
}

// x <- x^e for a small public exponent, square-and-multiply on the value held in slot `base` (Montgomery form).
void emit_pow_small(Prog& p, const BigU& e, uint32_t base) {
  // caller guarantees e >= 1 and x == mem[base] on entry
  for (size_t i = e.bit_length() - 1; i-- > 0;) {
    p.op(VM_SQR);
    if (e.bit(i)) p.op(VM_MUL, base);
  }
}

}  // namespace

extern "C" {

int pgpu_decrypt(const pgpu_seckey* sk, int level, size_t batch, const uint8_t* c, size_t c_stride, uint8_t* m,
                 size_t m_stride, int mem, int flags, int32_t* status) {
  if (!sk) return fail(PGPU_ERR_INVALID, "null key");
  pgpu_ctx* ctx = sk->ctx;
  return guarded([&] {
    check_batch_args(c, m, batch);
    ctx->bind();
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    const pgpu_pubkey* pk = sk->pk;
    if (level == PGPU_LEVEL_TWO) {
      const ModCtx& mn3 = cipher_mod(pk, level);
      if (c_stride < mn3.nbytes) api_throw(PGPU_ERR_INVALID, "ciphertext stride smaller than the byte length of n^3");
      const bool crt2 = sk->has_crt2 && !(flags & PGPU_DECRYPT_NO_CRT) && 2 * sk->mp3.WT >= mn3.WT;
      const int WC3 = crt2 ? 2 * sk->mp3.WT : mn3.WT;
      uint32_t* cl3 = ctx->ws_t<uint32_t>((size_t)WC3 * nb);
      unpack_operand(ctx, c, c_stride, mn3.nbytes, batch, mem, cl3, WC3, nb);
      std::vector<int32_t> hstat2(batch, 0);
      uint32_t* r2 = decrypt2_units_or_generic(sk, cl3, WC3, nb, batch, crt2, hstat2.data());
      pack_result(ctx, r2, pk->mn2.WT, nb, batch, m, m_stride, pk->mn2.nbytes, mem);
      if (status) memcpy(status, hstat2.data(), batch * 4);
      HIPCHK(hipStreamSynchronize(ctx->stream));
      return;
    }
    if (level != PGPU_LEVEL_ONE) api_throw(PGPU_ERR_INVALID, "bad encryption level");
    const size_t cbytes = pk->mn2.nbytes;
    if (c_stride < cbytes) api_throw(PGPU_ERR_INVALID, "ciphertext stride smaller than the byte length of n^2");
    int32_t* d_status = ctx->ws_t<int32_t>(nb);
    HIPCHK(hipMemsetAsync(d_status, 0, nb * 4, ctx->stream));
    const bool crt = sk->has_crt && !(flags & PGPU_DECRYPT_NO_CRT);
    // CRT consumes c as two chunks of WT(p^2) limbs (c = lo + hi * R_p); WT(n^2) <= 2 WT(p^2) always, with equality for
    // real key sizes and strict inequality for toy keys where every modulus gets the minimum shape: zero-extend.
    const int WG = pk->mn2.WT;
    // (and for 4096-bit keys n^2 takes a wider kernel shape than two chunks of p^2: the unpacked array then has the
    // generic width, CRT reads its first 2 WT(p^2) limb rows -- everything above bit 8192 is zero)
    const int WC = crt ? std::max(2 * sk->mp2.WT, WG) : WG;
    uint32_t* cl = ctx->ws_t<uint32_t>((size_t)WC * nb);
    // the ciphertext is the last cbytes of each element (values >= n^2 are reduced implicitly)
    unpack_operand(ctx, c, c_stride, cbytes, batch, mem, cl, WC, nb);
    uint32_t* res;
    std::vector<int32_t> hstat(batch, 0);
    if (crt) {
      res = decrypt1_crt(sk, cl, nb, batch, d_status);
      // the plaintexts are packed while the status words travel to the host (no idle GPU behind the read-back); only when a
      // lane turns out to be a non-unit are they packed again after its recomputation
      pack_result(ctx, res, pk->mn.WT, nb, batch, m, m_stride, pk->mn.nbytes, mem);
      HIPCHK(hipMemcpyAsync(hstat.data(), d_status, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      std::vector<uint32_t> idx;
      for (size_t i = 0; i < batch; ++i)
        if (hstat[i] & PGPU_LANE_NONUNIT) idx.push_back((uint32_t)i);
      if (idx.empty()) {
        if (status) memcpy(status, hstat.data(), batch * 4);
        return;
      }
      {
        // gcd(c, n) != 1 on these lanes: the CRT shortcut (L exact) does not apply; run the reference formula on them
        const size_t nbg = round_up(idx.size(), VM_BLOCK);
        uint32_t* d_idx = ctx->upload_words(idx);
        uint32_t* cg = ctx->ws_t<uint32_t>((size_t)WG * nbg);
        launch_gather(cl, nb, d_idx, idx.size(), cg, nbg, WG, ctx->stream);   // the low WT(n^2) limbs hold all of c
        uint32_t* rg = decrypt1_generic(sk, cg, nbg, idx.size());
        launch_scatter(rg, nbg, d_idx, idx.size(), res, nb, pk->mn.WT, ctx->stream);
      }
    } else {
      res = decrypt1_generic(sk, cl, nb, batch);
    }
    pack_result(ctx, res, pk->mn.WT, nb, batch, m, m_stride, pk->mn.nbytes, mem);
    if (status) memcpy(status, hstat.data(), batch * 4);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

}  // extern "C"

namespace {

// post <- (1+n)^m mod n^3 = 1 + m n + C(m,2) n^2 (mod n^3) for canonical m < n^2 (mn2.WT limbs)
void gm2_from_reduced(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint32_t* mred, size_t nb, uint32_t* post) {
  const ModCtx &mn = pk->mn, &mn2 = pk->mn2, &mn3 = *pk->mn3;
  const int W1 = mn.WT, W2 = mn2.WT, W3 = mn3.WT;
  // t = C(m,2) mod n = (m mod n) ((m-1) mod n) 2^-1 mod n
  const size_t s1 = (size_t)W1 * nb;
  uint32_t* mv = ctx->ws_t<uint32_t>(s1 * 4);   // slots: 0 m0, 1 (m0-1) mod n, 2 t, 3 ones
  reduce_mod(ctx, mn, mred, W2, mv, nb);
  launch_fill_const(mn.d_consts + (size_t)C_ONE * W1, mv + 3 * s1, W1, nb, ctx->stream);
  launch_sub_mod(mv, mv + 3 * s1, mn.d_nmod, mv + s1, W1, nb, ctx->stream);
  Prog pt;
  pt.op(VM_LOAD, 0); pt.op(VM_MULC, C_R2); pt.op(VM_MUL, 1); pt.op(VM_MULC, (uint32_t)pk->c_inv2R); pt.op(VM_STORE, 2);
  pt.end();
  SegSpec st{&mn, &pt, mv, nullptr};
  run_vm(ctx, nb, st, nullptr, false);
  launch_canon(mv + 2 * s1, mn.d_nmod, W1, nb, ctx->stream);
  uint32_t* tmpa = ctx->ws_t<uint32_t>((size_t)W3 * nb);
  launch_mul_const_add(mred, W2, pk->n_limbs.d, W1, nullptr, 0, 1, tmpa, W3, nb, ctx->stream);            // 1 + m n
  launch_mul_const_add(mv + 2 * s1, W1, pk->n2_limbs.d, W2, tmpa, W3, 0, post, W3, nb, ctx->stream);      // + t n^2
  launch_canon(post, mn3.d_nmod, W3, nb, ctx->stream);                                                   // mod n^3
}

// post <- G^m mod n^(s+1) for G = n + 1 (closed form; paillier.go:213 with the generator the reference always uses):
//   s = 1: 1 + (m mod n) n                       s = 2: 1 + m n + C(m,2) n^2 (mod n^3), m taken mod n^2
// `post` has cipher_mod(level).WT limbs per number.
void build_gm(pgpu_ctx* ctx, const pgpu_pubkey* pk, int level, const uint8_t* m, size_t m_stride, size_t batch, int mem,
              size_t nb, uint32_t* post) {
  const ModCtx &mn = pk->mn, &mn2 = pk->mn2;
  const int W1 = mn.WT, W2 = mn2.WT;
  if (!pk->g_is_n_plus_1) {
    // a caller-supplied generator (PublicKey.G is an exported field): the literal Exp(G, m, n^(s+1)) of paillier.go:213,
    // one exponent per ciphertext, uniform base
    const ModCtx& mc = (level == PGPU_LEVEL_ONE) ? mn2 : *pk->mn3;
    const size_t mlen = m_stride;
    const int we = std::max<int>(1, (int)((mlen * 8 + LB - 1) / LB));
    uint32_t* exps = ctx->ws_t<uint32_t>((size_t)we * nb);
    unpack_operand(ctx, m, m_stride, mlen, batch, mem, exps, we, nb);
    ModexpPlan pl = modexp_alloc(ctx, mc, nb, 16);
    uint32_t* gl = ctx->upload_words((pk->G % mc.N).to_limbs(LB, mc.WT));
    launch_fill_const(gl, pl.in(), mc.WT, nb, ctx->stream);
    modexp_perlane_run(ctx, mc, pl, exps, we, false, false);
    HIPCHK(hipMemcpyAsync(post, pl.out(), (size_t)mc.WT * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return;
  }
  if (level == PGPU_LEVEL_ONE) {
    uint32_t* mred = ctx->ws_t<uint32_t>((size_t)W1 * nb);
    unpack_mod(ctx, mn, m, m_stride, batch, mem, mred, nb, true);   // the generator 1+n has order n: G^m = G^(m mod n)
    launch_mul_const_add(mred, W1, pk->n_limbs.d, W1, nullptr, 0, 1, post, W2, nb, ctx->stream);
    return;
  }
  uint32_t* mred = ctx->ws_t<uint32_t>((size_t)W2 * nb);
  unpack_mod(ctx, mn2, m, m_stride, batch, mem, mred, nb, true);   // 1+n has order n^2 modulo n^3
  gm2_from_reduced(ctx, pk, mred, nb, post);
}

// Fixed-base comb table of h_s for AltEncrypt (paillier.go:416-434: h_1 = (N-H)^N mod N^2, h_2 = (N^2-H)^(N^2) mod N^3):
// entries h^(d 16^i) (Montgomery form), i < ceil(log2(K)/4), d < 16, appended to the ciphertext modulus' constants.
void ensure_alt_table(pgpu_pubkey* pk, int level) {
  pgpu_pubkey::AltTab& t = pk->alt[level];
  if (t.built) return;
  if (pk->H.is_zero() || pk->Kk.is_zero()) api_throw(PGPU_ERR_INVALID, "alternative encryption needs H and K in the public key");
  const size_t kbits = pk->Kk.bit_length() - 1;
  if (!(hostbig::shl(BigU(1), kbits) == pk->Kk)) api_throw(PGPU_ERR_UNSUPPORTED, "K must be a power of two (KeyGen: 2^(secparam/2))");
  ModCtx& mc = (level == PGPU_LEVEL_ONE) ? pk->mn2 : *pk->mn3;
  const BigU ns = (level == PGPU_LEVEL_ONE) ? pk->N : pk->mn2.N;
  if (hostbig::cmp(ns, pk->H) <= 0) api_throw(PGPU_ERR_INVALID, "H must be smaller than n^s");
  BigU base = hostbig::powmod(ns - pk->H, ns, mc.N);
  t.kbits = kbits;
  t.nwin = (int)((kbits + 3) / 4);
  t.base = (int)mc.consts.size();
  const BigU rmod = mc.R % mc.N;
  for (int i = 0; i < t.nwin; ++i) {
    BigU cur(1);
    for (int d = 0; d < 16; ++d) {
      mc.consts.push_back(hostbig::mulmod(cur, rmod, mc.N));   // Montgomery form of base^d
      cur = hostbig::mulmod(cur, base, mc.N);
    }
    base = cur;  // base^16
  }
  mc.upload();
  t.built = true;
}

}  // namespace

extern "C" {

// c = G^m * r^(n^s) mod n^(s+1) with r given as canonical limbs on the device (r_limbs: mc.WT limbs, stride nb) or as a
// byte buffer (r, r_stride, in `mem`)
static void encrypt_core(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* m, size_t m_stride, const uint8_t* r,
                         size_t r_stride, const uint32_t* r_limbs, uint8_t* c, size_t c_stride, int mem) {
  pgpu_ctx* ctx = pk->ctx;
  const ModCtx& mc = cipher_mod(pk, level);
  const size_t nb = round_up(batch, VM_BLOCK);
  ModexpPlan pl = modexp_alloc(ctx, mc, nb, 32);
  build_gm(ctx, pk, level, m, m_stride, batch, mem, nb, pl.post());
  if (level == PGPU_LEVEL_TWO && ctx->use_lift) {
    // r^(n^2) mod n^3 = ((r mod n^2)^n mod n^2)^n mod n^3 for EVERY integer r: x = x' (mod n^k) implies x^n = x'^n (mod n^(k+1))
    // (binomial: the second term of (x' + t n^k)^n is n x'^(n-1) t n^k).  Half of the 4 096 squarings of paillier.go:213's
    // Exp(r, n^2, n^3) move to the modulus n^2, where a squaring costs half as much: level-two Encrypt -20 %.
    const ModCtx& m2 = pk->mn2;
    ModexpPlan p2 = modexp_alloc(ctx, m2, nb, 32);
    if (r_limbs) reduce_mod(ctx, m2, r_limbs, mc.WT, p2.in(), nb);
    else unpack_mod(ctx, m2, r, r_stride, batch, mem, p2.in(), nb, true);
    // (the power stays in pair form when the second ladder runs on the digit kernel of the same n: (a0, a1, 0) is its base)
    uint32_t* raw = nullptr;
    const bool digit_next = ctx->use_handover && triple_usable(ctx, mc, true) && pk->N.bit_length() >= 256 && m2.pairn.root &&
                            mc.triple.root && m2.pairn.root->WT == mc.triple.root->WT;
    modexp_shared_run(ctx, m2, p2, pk->N, false, false, true, digit_next ? &raw : nullptr);   // y = r^n mod n^2
    if (raw) {
      modexp_shared_run(ctx, mc, pl, pk->N, false, true, true, nullptr, raw);                 // y^n * g^m mod n^3
    } else {
      launch_copy_limbs(p2.out(), 0, m2.WT, pl.in(), mc.WT, nb, ctx->stream);
      modexp_shared_run(ctx, mc, pl, pk->N, false, true, true);
    }
    pack_result(ctx, pl.out(), mc.WT, nb, batch, c, c_stride, mc.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return;
  }
  if (r_limbs) HIPCHK(hipMemcpyAsync(pl.in(), r_limbs, (size_t)mc.WT * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
  else unpack_mod(ctx, mc, r, r_stride, batch, mem, pl.in(), nb);
  const BigU& ns = (level == PGPU_LEVEL_ONE) ? pk->N : pk->mn2.N;
  modexp_shared_run(ctx, mc, pl, ns, false, true, true);  // r^(n^s) * g^m mod n^(s+1)  (public exponent: zero windows skipped)
  pack_result(ctx, pl.out(), mc.WT, nb, batch, c, c_stride, mc.nbytes, mem);
  HIPCHK(hipStreamSynchronize(ctx->stream));
}

int pgpu_encrypt_with_r(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* m, size_t m_stride,
                        const uint8_t* r, size_t r_stride, uint8_t* c, size_t c_stride, int mem) {
  if (!pk) return fail(PGPU_ERR_INVALID, "null key");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    check_batch_args(m, c, batch);
    if (!r) api_throw(PGPU_ERR_INVALID, "null buffer");
    if (level != PGPU_LEVEL_ONE && level != PGPU_LEVEL_TWO) api_throw(PGPU_ERR_INVALID, "bad encryption level");
    ctx->bind();
    ctx->reset_ws();
    encrypt_core(pk, level, batch, m, m_stride, r, r_stride, nullptr, c, c_stride, mem);
  });
}

// ---- randomness (utils.go:26-49) --------------------------------------------------------------------------------------
// GetRandomNumber = crypto/rand.Int(rand.Reader, n): k = ceil(bitlen(n)/8) bytes from the operating system's CSPRNG, the
// excess bits of the first byte cleared, accepted when the value is below n -- uniform on [0, n) by rejection.
// GetRandomNumberInMultiplicativeGroup redraws while r = 0 or gcd(r, n) != 1.  Here: getrandom(2) on several host threads
// into one byte matrix (the draws and the r < n rejection are a few ms per 65 536 elements), and the unit test for the whole
// batch on the device -- the batch-inverse tree modulo n: if its single inversion succeeds every r is a unit (the
// overwhelmingly likely case: a non-unit would reveal a factor of n); if not, the per-lane GCD kernel names the lanes to
// redraw.  Randomness never comes from anywhere but the OS.
static void os_random(uint8_t* p, size_t n) {
  while (n) {
    ssize_t got = getrandom(p, std::min<size_t>(n, 1u << 20), 0);
    if (got < 0) {
      if (errno == EINTR) continue;
      api_throw(PGPU_ERR_INVALID, "getrandom failed");
    }
    p += got;
    n -= (size_t)got;
  }
}

static void draw_below(const std::vector<uint8_t>& n_be, uint8_t top_mask, uint8_t* out, size_t count, const uint32_t* only,
                       size_t n_only) {
  const size_t k = n_be.size();
  auto fill = [&](size_t lo, size_t hi) {
    std::vector<uint8_t> pool;
    size_t pos = 0;
    for (size_t j = lo; j < hi; ++j) {
      uint8_t* r = out + (only ? (size_t)only[j] : j) * k;
      for (;;) {
        if (pos + k > pool.size()) { pool.resize(std::max<size_t>(k * 256, k)); os_random(pool.data(), pool.size()); pos = 0; }
        memcpy(r, pool.data() + pos, k);
        pos += k;
        r[0] &= top_mask;
        bool zero = true;
        for (size_t b = 0; b < k; ++b) if (r[b]) { zero = false; break; }
        if (!zero && memcmp(r, n_be.data(), k) < 0) break;      // 0 < r < n
      }
    }
    wipe_vec(pool);
  };
  const size_t total = only ? n_only : count;
  const size_t nthreads = std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)std::thread::hardware_concurrency(), total / 2048 + 1}));
  if (nthreads == 1) { fill(0, total); return; }
  std::vector<std::thread> th;
  std::vector<std::string> errs(nthreads);
  for (size_t t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      try { fill(total * t / nthreads, total * (t + 1) / nthreads); } catch (const ApiError& e) { errs[t] = e.msg; }
    });
  for (auto& t : th) t.join();
  for (auto& e : errs) if (!e.empty()) api_throw(PGPU_ERR_INVALID, e);
}

// `count` uniform elements of Z_n^* as canonical limbs on the device (mn.WT limbs, stride nb); host_out (optional): the
// same values as big-endian bytes, k = byte length of n per element
static uint32_t* random_units_device(const pgpu_pubkey* pk, size_t count, size_t nb, std::vector<uint8_t>* host_out,
                                     UnitCheck* deferred = nullptr) {
  // deferred: the unit test of the draws is only STARTED (on the side stream); the caller runs its ladder on them meanwhile and
  // asks deferred->finish() afterwards -- a non-unit among uniform draws modulo an honest n would be a factor of n, so the
  // test all but never fails, and when it does the caller draws again the careful way
  pgpu_ctx* ctx = pk->ctx;
  const ModCtx& mn = pk->mn;
  std::vector<uint8_t> n_be = pk->N.to_be_min();
  const size_t k = n_be.size();
  const int excess = (int)(k * 8 - pk->N.bit_length());
  const uint8_t top_mask = (uint8_t)(0xFFu >> excess);
  std::vector<uint8_t> buf(count * k);
  draw_below(n_be, top_mask, buf.data(), count, nullptr, 0);
  uint32_t* limbs = ctx->ws_t<uint32_t>((size_t)mn.WT * nb);
  uint8_t* stage = (uint8_t*)ctx->ws(count * k);
  int32_t* d_bad = ctx->ws_t<int32_t>(nb);
  for (int round = 0; round < 64; ++round) {
    HIPCHK(hipMemcpyAsync(stage, buf.data(), count * k, hipMemcpyHostToDevice, ctx->stream));
    launch_unpack_be(stage, k, k, count, limbs, mn.WT, nb, ctx->stream);
    // padding lanes: 1 (a unit), so that the tree sees units only
    launch_restride(limbs, nb, count, mn.d_consts + (size_t)C_ONE * mn.WT, limbs, nb, mn.WT, ctx->stream);
    if (deferred && ctx->use_side) {
      deferred->side.reset(new SideStream(ctx, 0));
      deferred->side->enter(deferred->side->mark());
      deferred->begin(ctx, mn, limbs, nb, count);
      deferred->side->leave();
      break;
    }
    if (all_units(ctx, mn, limbs, nb, count)) break;
    bool any_bad = false;
    (void)batch_inverse(ctx, mn, limbs, nb, count, d_bad, &any_bad);      // names the lanes to redraw
    if (!any_bad) break;
    std::vector<int32_t> bad(count);
    HIPCHK(hipMemcpyAsync(bad.data(), d_bad, count * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> redo;
    for (size_t i = 0; i < count; ++i) if (bad[i]) redo.push_back((uint32_t)i);
    if (redo.empty()) break;
    draw_below(n_be, top_mask, buf.data(), count, redo.data(), redo.size());     // utils.go:46: draw again
    if (round == 63) api_throw(PGPU_ERR_INVALID, "could not draw units modulo n (is n a product of tiny primes?)");
  }
  if (host_out) *host_out = buf;
  wipe_vec(buf);
  return limbs;
}

int pgpu_random_units(const pgpu_pubkey* pk, size_t batch, uint8_t* r_out, size_t r_stride, int mem) {
  if (!pk || !r_out) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    if (batch == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
    ctx->bind();
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    uint32_t* limbs = random_units_device(pk, batch, nb, nullptr);
    pack_result(ctx, limbs, pk->mn.WT, nb, batch, r_out, r_stride, pk->mn.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_encrypt(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* m, size_t m_stride, uint8_t* c, size_t c_stride,
                 uint8_t* r_out, size_t r_stride, int mem) {
  if (!pk) return fail(PGPU_ERR_INVALID, "null key");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    check_batch_args(m, c, batch);
    if (level != PGPU_LEVEL_ONE && level != PGPU_LEVEL_TWO) api_throw(PGPU_ERR_INVALID, "bad encryption level");
    ctx->bind();
    ctx->reset_ws();
    const ModCtx& mc = cipher_mod(pk, level);
    const size_t nb = round_up(batch, VM_BLOCK);
    // the gcd test of the draws runs beside the ladder (side stream); should it ever fail, the call is redone with the test first
    for (int attempt = 0; attempt < 2; ++attempt) {
      UnitCheck chk;
      uint32_t* r1 = random_units_device(pk, batch, nb, nullptr, attempt == 0 ? &chk : nullptr);   // paillier.go:263: r in Z_n^* for either level
      if (r_out) pack_result(ctx, r1, pk->mn.WT, nb, batch, r_out, r_stride, pk->mn.nbytes, mem);
      uint32_t* rw = r1;
      if (mc.WT != pk->mn.WT) {                                            // zero-extend to the width of n^(s+1)
        rw = ctx->ws_t<uint32_t>((size_t)mc.WT * nb);
        launch_copy_limbs(r1, 0, pk->mn.WT, rw, mc.WT, nb, ctx->stream);
      }
      encrypt_core(pk, level, batch, m, m_stride, nullptr, 0, rw, c, c_stride, mem);
      if (!chk.begun || chk.finish()) break;
      ctx->reset_ws();
    }
  });
}

int pgpu_alt_encrypt_with_r(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* m, size_t m_stride,
                            const uint8_t* r, size_t r_stride, uint8_t* c, size_t c_stride, uint8_t* r_reduced, int mem) {
  if (!pk) return fail(PGPU_ERR_INVALID, "null key");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    check_batch_args(m, c, batch);
    if (!r) api_throw(PGPU_ERR_INVALID, "null buffer");
    if (level != PGPU_LEVEL_ONE && level != PGPU_LEVEL_TWO) api_throw(PGPU_ERR_INVALID, "bad encryption level");
    const ModCtx& mc = cipher_mod(pk, level);
    ctx->bind();
    ensure_alt_table(const_cast<pgpu_pubkey*>(pk), level);
    const pgpu_pubkey::AltTab& t = pk->alt[level];
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    const size_t sw = (size_t)mc.WT * nb;
    uint32_t* memv = ctx->ws_t<uint32_t>(sw * 2);   // slots: 0 g^m (post), 1 out
    build_gm(ctx, pk, level, m, m_stride, batch, mem, nb, memv);
    // r mod K: K = 2^kbits, so the low kbits of r (paillier.go:228, which also overwrites the caller's r)
    const int we = (int)((t.kbits + LB - 1) / LB);
    const size_t rbytes = std::min(r_stride, (t.kbits + 7) / 8);
    uint32_t* exps = ctx->ws_t<uint32_t>((size_t)std::max(we, 1) * nb);
    {
      // unpack only the low ceil(kbits/8) bytes of each r, then clear the bits above kbits
      const uint8_t* d = r;
      if (mem == PGPU_MEM_HOST) {
        uint8_t* stg = (uint8_t*)ctx->ws(r_stride * batch);
        HIPCHK(hipMemcpyAsync(stg, r, r_stride * batch, hipMemcpyHostToDevice, ctx->stream));
        d = stg;
      }
      launch_unpack_be(d + (r_stride - rbytes), r_stride, rbytes, batch, exps, we, nb, ctx->stream);
      launch_mask_bits(exps, we, nb, t.kbits, ctx->stream);
    }
    Prog p;
    p.op(VM_LOADC, C_ONE_M);
    for (int i = 0; i < t.nwin; ++i) p.op(VM_MULCV, (uint32_t)i, (uint32_t)t.base);
    p.op(VM_MUL, 0);      // * g^m (plain) -> leaves Montgomery form
    p.op(VM_STORE, 1);
    p.end();
    SegSpec sg{&mc, &p, memv, exps};
    run_vm(ctx, nb, sg, nullptr, true);
    launch_canon(memv + sw, mc.d_nmod, mc.WT, nb, ctx->stream);
    pack_result(ctx, memv + sw, mc.WT, nb, batch, c, c_stride, mc.nbytes, mem);
    if (r_reduced) pack_result(ctx, exps, we, nb, batch, r_reduced, r_stride, std::min(r_stride, (size_t)((t.kbits + 7) / 8)), mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_add_many(const pgpu_pubkey* pk, int level, int n_ops, size_t batch, const uint8_t* const* ops, size_t stride,
                  uint8_t* out, size_t out_stride, int mem) {
  if (!pk || !ops) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    if (n_ops < 1) api_throw(PGPU_ERR_INVALID, "Add needs at least one operand (the reference indexes cts[0])");
    check_batch_args(ops[0], out, batch);
    const ModCtx& mc = cipher_mod(pk, level);
    ctx->bind();
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    const size_t sw = (size_t)mc.WT * nb;
    uint32_t* memv = ctx->ws_t<uint32_t>(sw * (size_t)(n_ops + 1));
    for (int k = 0; k < n_ops; ++k) {
      if (!ops[k]) api_throw(PGPU_ERR_INVALID, "null operand buffer");
      unpack_mod(ctx, mc, ops[k], stride, batch, mem, memv + (size_t)k * sw, nb);
    }
    // operations.go:12-22: accumulator = 1; accumulator = accumulator * c mod n^(s+1) for every operand
    Prog p;
    p.op(VM_LOAD, 0);
    for (int k = 1; k < n_ops; ++k) { p.op(VM_MULC, C_R2); p.op(VM_MUL, (uint32_t)k); }
    if (n_ops == 1) { p.op(VM_MULC, C_R2); p.op(VM_MULC, C_ONE); }   // 1 * c mod n^(s+1): a single operand comes back reduced
    p.op(VM_STORE, (uint32_t)n_ops);
    p.end();
    SegSpec sg{&mc, &p, memv, nullptr};
    run_vm(ctx, nb, sg, nullptr, true);
    launch_canon(memv + (size_t)n_ops * sw, mc.d_nmod, mc.WT, nb, ctx->stream);
    pack_result(ctx, memv + (size_t)n_ops * sw, mc.WT, nb, batch, out, out_stride, mc.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_add(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* a, size_t a_stride, const uint8_t* b,
             size_t b_stride, uint8_t* out, size_t out_stride, int mem) {
  if (a_stride != b_stride) return fail(PGPU_ERR_INVALID, "pgpu_add: both operands must share one stride");
  const uint8_t* ops[2] = {a, b};
  return pgpu_add_many(pk, level, 2, batch, ops, a_stride, out, out_stride, mem);
}

int pgpu_sub_many(const pgpu_pubkey* pk, int level, int n_ops, size_t batch, const uint8_t* const* ops, size_t stride,
                  uint8_t* out, size_t out_stride, int mem, int32_t* status) {
  if (!pk || !ops) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    if (n_ops < 1) api_throw(PGPU_ERR_INVALID, "Sub needs at least one operand (the reference indexes cts[0])");
    check_batch_args(ops[0], out, batch);
    const ModCtx& mc = cipher_mod(pk, level);
    ctx->bind();
    ctx->reset_ws();
    if (n_ops == 1) {
      // operations.go:34-47: accumulator := cts[0].C and the loop body never runs -- the operand comes back UNREDUCED
      if (out_stride < stride) api_throw(PGPU_ERR_INVALID, "single-operand Sub returns its operand: out_stride < stride");
      if (mem == PGPU_MEM_HOST) {
        for (size_t i = 0; i < batch; ++i) {
          memset(out + i * out_stride, 0, out_stride - stride);
          memcpy(out + i * out_stride + (out_stride - stride), ops[0] + i * stride, stride);
        }
      } else {
        HIPCHK(hipMemsetAsync(out, 0, out_stride * batch, ctx->stream));
        HIPCHK(hipMemcpy2DAsync(out + (out_stride - stride), out_stride, ops[0], stride, stride, batch, hipMemcpyDeviceToDevice,
                                ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
      }
      if (status) memset(status, 0, batch * sizeof(int32_t));
      return;
    }
    const size_t nb = round_up(batch, VM_BLOCK);
    const size_t sw = (size_t)mc.WT * nb;
    // slots: 0 minuend, 1..n-1 subtrahends, n denominator / its inverse, n+1 out
    const uint32_t SD = (uint32_t)n_ops, SO = SD + 1;
    uint32_t* memv = ctx->ws_t<uint32_t>(sw * (size_t)(n_ops + 2));
    for (int k = 0; k < n_ops; ++k) {
      if (!ops[k]) api_throw(PGPU_ERR_INVALID, "null operand buffer");
      unpack_mod(ctx, mc, ops[k], stride, batch, mem, memv + (size_t)k * sw, nb);
    }
    // operations.go:43-47 inverts every subtrahend and multiplies the inverses in; the product of the inverses is the
    // inverse of the product, so ONE inversion per ciphertext gives the same canonical residue
    {
      Prog p;
      p.op(VM_LOAD, 1);
      for (int k = 2; k < n_ops; ++k) { p.op(VM_MULC, C_R2); p.op(VM_MUL, (uint32_t)k); }
      p.op(VM_MULC, C_R2); p.op(VM_MULC, C_ONE);      // reduce below 2N whatever the operand was
      p.op(VM_STORE, SD);
      p.end();
      SegSpec sg{&mc, &p, memv, nullptr};
      run_vm(ctx, nb, sg, nullptr, false);
      launch_canon(memv + (size_t)SD * sw, mc.d_nmod, mc.WT, nb, ctx->stream);
    }
    int32_t* d_bad = ctx->ws_t<int32_t>(nb);
    bool any_bad = false;
    uint32_t* dinv = batch_inverse(ctx, mc, memv + (size_t)SD * sw, nb, batch, d_bad, &any_bad);   // operations.go:43 ModInverse
    BadLanes bl;
    bl.collect(ctx, d_bad, batch, any_bad);
    HIPCHK(hipMemcpyAsync(memv + (size_t)SD * sw, dinv, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
    Prog p;
    p.op(VM_LOAD, 0); p.op(VM_MULC, C_R2); p.op(VM_MUL, SD); p.op(VM_STORE, SO); p.end();   // operations.go:44-47
    SegSpec s2{&mc, &p, memv, nullptr};
    run_vm(ctx, nb, s2, nullptr, true);
    launch_canon(memv + (size_t)SO * sw, mc.d_nmod, mc.WT, nb, ctx->stream);
    pack_result(ctx, memv + (size_t)SO * sw, mc.WT, nb, batch, out, out_stride, mc.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    bl.finish(status, batch);
  });
}

int pgpu_sub(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* a, size_t a_stride, const uint8_t* b,
             size_t b_stride, uint8_t* out, size_t out_stride, int mem, int32_t* status) {
  if (a_stride != b_stride) return fail(PGPU_ERR_INVALID, "pgpu_sub: both operands must share one stride");
  const uint8_t* ops[2] = {a, b};
  return pgpu_sub_many(pk, level, 2, batch, ops, a_stride, out, out_stride, mem, status);
}

int pgpu_partial_decrypt(const pgpu_pubkey* pk, int total_servers, const uint8_t* share_be, size_t share_len, size_t batch,
                         const uint8_t* c, size_t c_stride, uint8_t* out, size_t out_stride, int mem) {
  if (!pk || !share_be) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    check_batch_args(c, out, batch);
    if (total_servers < 1) api_throw(PGPU_ERR_INVALID, "total_servers must be positive");
    ctx->bind();
    ctx->reset_ws();
    const ModCtx& mc = pk->mn2;
    const size_t nb = round_up(batch, VM_BLOCK);
    // thresholdkey.go:195: exp = Share * (2 * delta), delta = l!
    std::vector<BigU> ev{BigU::from_be(share_be, share_len) * (BigU(2) * factorial_big(total_servers))};
    WipeOnExit<std::vector<BigU>> wipe_e(ev);
    const BigU& e = ev[0];
    ModexpPlan pl = modexp_alloc(ctx, mc, nb, 32);
    unpack_mod(ctx, mc, c, c_stride, batch, mem, pl.in(), nb);
    modexp_shared_run(ctx, mc, pl, e, false, false, true);
    pack_result(ctx, pl.out(), mc.WT, nb, batch, out, out_stride, mc.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

// Pair-form entry / exit shared by the multi-share forms of PartialDecrypt (N = n^2, root n public).
// entry: canonical residues x (slot 0 of `ent`, 4 slots of mc.WT limbs, stride nb) -> digits X0 | X1 of x R_H in slot 2
static void pair_enter(pgpu_ctx* ctx, const ModCtx& mc, uint32_t* ent, size_t nb) {
  const PairInfo& pi = mc.pairn;
  const ModCtx& mn = *pi.root;
  const int H = mn.WT, W2 = mc.WT;
  const size_t S1 = (size_t)H * nb, SW = (size_t)W2 * nb;
  Prog a;
  a.op(VM_LOAD, 0); a.op(VM_MULC, C_R2); a.op(VM_MULC, (uint32_t)pi.c_rh); a.op(VM_STORE, 3); a.end();
  SegSpec sa{&mc, &a, ent, nullptr};
  run_vm(ctx, nb, sa, nullptr, false);
  launch_canon(ent + 3 * SW, mc.d_nmod, W2, nb, ctx->stream);
  uint32_t* x0 = ctx->ws_t<uint32_t>(S1);
  uint32_t* tb = ctx->ws_t<uint32_t>(SW);
  reduce_mod(ctx, mn, ent + 3 * SW, W2, x0, nb);
  launch_div_exact(ent + 3 * SW, W2, 0, x0, H, tb, pi.dinv, mn.d_nmod, H, ent + 2 * SW + S1, H, nb, nb, nullptr, 0, ctx->stream);
  HIPCHK(hipMemcpyAsync(ent + 2 * SW, x0, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
}
// exit: F~ = F0 + F1 n of slot `out_slot` of pm (stride nb), out of pair and Montgomery form, packed to dst (`count` results;
// slots 2 and 3 of pm are scratch by now)
static uint32_t* pair_leave(pgpu_ctx* ctx, const ModCtx& mc, uint32_t* pm, uint32_t out_slot, size_t nb) {
  const PairInfo& pi = mc.pairn;
  const int H = pi.root->WT, W2 = mc.WT;
  const size_t S1 = (size_t)H * nb, SW = (size_t)W2 * nb;
  launch_mul_const_add(pm + out_slot * SW + S1, H, pi.n_limbs, H, pm + out_slot * SW, H, 0, pm + 2 * SW, W2, nb, ctx->stream);
  Prog a;
  a.op(VM_LOAD, 2); a.op(VM_MULC, (uint32_t)pi.c_rh); a.op(VM_STORE, 3); a.end();
  SegSpec sa{&mc, &a, pm, nullptr};
  run_vm(ctx, nb, sa, nullptr, false);
  launch_canon(pm + 3 * SW, mc.d_nmod, W2, nb, ctx->stream);
  return pm + 3 * SW;                                     // canonical, stride nb
}
static void pair_leave_and_pack(pgpu_ctx* ctx, const ModCtx& mc, uint32_t* pm, uint32_t out_slot, size_t nb, size_t count, uint8_t* dst,
                                size_t out_stride, int mem) {
  pack_result(ctx, pair_leave(ctx, mc, pm, out_slot, nb), mc.WT, nb, count, dst, out_stride, mc.nbytes, mem);
}

// x^(per-number exponent, `we` limbs) * y^(shared exponent e) modulo N = n^2 as ONE interleaved ladder on the pair kernels
// (4-bit windows of the per-number exponent, sliding windows of e).  x, y: canonical residues (mc.WT limbs, stride nb).
// Returns the canonical result, or nullptr when the pair kernels do not serve this key / batch.
static uint32_t* dual_pow_pair(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, const uint32_t* exps, int we, const uint32_t* y,
                               const BigU& e, size_t nb, uint32_t** raw_out = nullptr) {
  // raw_out: the result stays in pair form (a0 | a1, stride nb): *raw_out and the return value point at its digits
  const PairInfo& pi = mc.pairn;
  if (!(pi.root && pi.c_one_pair >= 0 && ctx->use_asm && ctx->use_pair)) return nullptr;
  const int H = pi.root->WT, W2 = mc.WT;
  const size_t lanes_target = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
  const bool two = nb * 2 >= lanes_target;
  if (!two && !(H % 2 == 0 && vm_asm_available(H / 2, 64))) return nullptr;
  // per-number window table number-major (VM_STORET / VM_MULVT5 / VM_MULVT): limb-major, the 16 384-number ladder of the DDLEQ
  // verifier fetched 98 GB of 32-byte sectors for its dword gathers in a 43 ms launch (profiles/r03_bench_traffic.txt)
  const bool nm4 = ctx->use_nm4;
  const int wb = (uint64_t)nb * W2 * 4 * (uint64_t)(perlane_table_slots(5, nm4) + 1) < (1ull << 32) ? 5 : 4;   // gathers with 32-bit offsets
  if ((uint64_t)nb * W2 * 4 * (uint64_t)(perlane_table_slots(4, nm4) + 1) >= (1ull << 32)) return nullptr;
  const uint32_t tab2 = 5 + (uint32_t)perlane_table_slots(wb, nm4);
  const size_t SW = (size_t)W2 * nb;
  uint32_t* pm = ctx->ws_t<uint32_t>(SW * (size_t)(tab2 + 32));           // 0 x, 1 y, 2 tmp, 3 out, 5.. / tab2.. the tables
  Fork fk(ctx);                                                            // y's entry chain beside x's
  for (int k = 0; k < 2; ++k) {
    fk.chain(k);
    uint32_t* ent = ctx->ws_t<uint32_t>(SW * 4);
    HIPCHK(hipMemcpyAsync(ent, k ? y : x, SW * 4, hipMemcpyDeviceToDevice, ctx->stream));
    pair_enter(ctx, mc, ent, nb);
    HIPCHK(hipMemcpyAsync(pm + (size_t)k * SW, ent + 2 * SW, SW * 4, hipMemcpyDeviceToDevice, ctx->stream));
  }
  fk.join();
  Prog pd;
  emit_modexp_dual(pd, we, e, 0, 1, 2, 3, 5, tab2, pi.c_one_pair, wb, nm4);
  pd.end();
  SegSpec sp{&mc, &pd, pm, wb == 5 ? windows5_of(ctx, exps, we, nb) : exps};
  sp.pair = pi.consts; sp.pair_n0inv = pi.root->n0inv; sp.pair_h = H; sp.pair_lanes = two ? 2 : 4;
  run_vm(ctx, nb, sp, nullptr, true);
  if (raw_out) return *raw_out = pm + 3 * SW;
  return pair_leave(ctx, mc, pm, 3, nb);
}

int pgpu_partial_decrypt_multi(const pgpu_pubkey* pk, int total_servers, int n_shares, const uint8_t* const* shares_be,
                               const size_t* share_lens, size_t batch, const uint8_t* c, size_t c_stride, uint8_t* const* outs,
                               size_t out_stride, int mem) {
  if (!pk || !shares_be || !share_lens || !outs) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    if (n_shares < 1 || n_shares > 256 || total_servers < 1) api_throw(PGPU_ERR_INVALID, "bad share count");
    check_batch_args(c, outs[0], batch);
    ctx->bind();
    ctx->reset_ws();
    const ModCtx& mc = pk->mn2;
    const size_t nb = round_up(batch, VM_BLOCK);
    const BigU two_delta = BigU(2) * factorial_big(total_servers);
    std::vector<BigU> es;
    WipeOnExit<std::vector<BigU>> wipe_es(es);             // share * 2 delta: wiped on every exit, the fallbacks and errors too
    for (int k = 0; k < n_shares; ++k) {
      if (!shares_be[k] || !outs[k]) api_throw(PGPU_ERR_INVALID, "null share / output buffer");
      es.push_back(BigU::from_be(shares_be[k], share_lens[k]) * two_delta);       // thresholdkey.go:195
    }
    const PairInfo& pi = mc.pairn;
    const bool pair_ok = pi.root && ctx->use_asm && ctx->use_pair;
    bool all_long = true;
    for (auto& e : es) all_long = all_long && e.bit_length() >= 256;
    if (!pair_ok || !all_long) {                                   // no pair kernel for this key: server after server
      for (int k = 0; k < n_shares; ++k) {
        ModexpPlan pl = modexp_alloc(ctx, mc, nb, 32);
        unpack_mod(ctx, mc, c, c_stride, batch, mem, pl.in(), nb);
        modexp_shared_run(ctx, mc, pl, es[k], false, false, true);
        pack_result(ctx, pl.out(), mc.WT, nb, batch, outs[k], out_stride, mc.nbytes, mem);
      }
      HIPCHK(hipStreamSynchronize(ctx->stream));
      return;
    }
    // Every server raises the SAME ciphertexts to its own exponent: the entry into the pair form is done once, and the
    // ladders of two servers share a launch (two program segments) -- 2 x 16 384 numbers fill the chip with the two-lane
    // kernel, where one server's 16 384 alone need the less efficient four-lane slicing.
    const ModCtx& mn = *pi.root;
    const int H = mn.WT, W2 = mc.WT;
    const size_t S1 = (size_t)H * nb, SW = (size_t)W2 * nb;
    uint32_t* ent = ctx->ws_t<uint32_t>(SW * 4);          // generic slots: 0 x, 1 -, 2 digits (X0 | X1), 3 X
    unpack_mod(ctx, mc, c, c_stride, batch, mem, ent, nb);
    pair_enter(ctx, mc, ent, nb);
    const size_t lanes_target = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
    auto leave_pair_form = [&](uint32_t* pm, uint32_t out_slot, uint8_t* dst) {
      pair_leave_and_pack(ctx, mc, pm, out_slot, nb, batch, dst, out_stride, mem);
    };
    if (n_shares >= 2 && ctx->use_shared_chain && pi.c_one_pair >= 0 && nb * 8 >= lanes_target) {
      // One chain of squarings for all the servers (emit_multi_exp_shared_base): the ciphertexts are the same, only the
      // exponents differ.  A batch that fills at least half the chip on its own takes this path; smaller ones are bound by
      // the length of the operation sequence, where separate ladders side by side (below) are shorter.
      const int w = 7;
      const uint32_t K = 1u << (w - 1);
      const size_t per_server = (size_t)K * SW * 4;
      const int group = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_shares, ((size_t)12 << 30) / per_server));
      // one slot array for every group, sized for the largest (the bump workspace releases nothing before the next call)
      uint32_t* pm = ctx->ws_t<uint32_t>(SW * (size_t)(6 + (size_t)group + (size_t)group * K));
      for (int k0 = 0; k0 < n_shares; k0 += group) {
        const int S = std::min(group, n_shares - k0);
        const uint32_t OUT0 = 6, B0 = 6 + (uint32_t)S;
        HIPCHK(hipMemcpyAsync(pm + 2 * SW, ent + 2 * SW, SW * 4, hipMemcpyDeviceToDevice, ctx->stream));
        Prog pr;
        const int lanes = (nb * 2 >= lanes_target || !(H % 2 == 0 && vm_asm_available(H / 2, 64))) ? 2 : 4;
        emit_multi_exp_shared_base(pr, std::vector<BigU>(es.begin() + k0, es.begin() + k0 + S), 2, 3, 4, 5, OUT0, B0, w,
                                   (uint32_t)pi.c_one_pair, lanes == 4 && ctx->use_muls);
        pr.end();
        SegSpec sg{&mc, &pr, pm, nullptr};
        sg.pair = pi.consts; sg.pair_n0inv = mn.n0inv; sg.pair_h = H; sg.pair_lanes = lanes;
        run_vm(ctx, nb, sg, nullptr, true);
        for (int j = 0; j < S; ++j) leave_pair_form(pm, OUT0 + (uint32_t)j, outs[k0 + j]);
      }
      HIPCHK(hipStreamSynchronize(ctx->stream));
      return;
    }
    for (int k = 0; k < n_shares;) {
      // two servers per launch: 2 x 16 384 numbers x 2 lanes are exactly one wave per SIMD.  (Three segments -- the kernels
      // take up to three -- would be 1.5 waves per SIMD: the SIMDs that got two waves take as long as a full second wave,
      // measured 236 ms against 122 + 76 ms for a pair plus a single.)
      const int left = n_shares - k;
      const int segs = std::min(left, 2);
      const int lanes = ((size_t)segs * nb * 2 >= lanes_target || !(H % 2 == 0 && vm_asm_available(H / 2, 64))) ? 2 : 4;
      uint32_t* pm[3];
      Prog pr[3];
      SegSpec sg[3];
      for (int j = 0; j < segs; ++j) {
        pm[j] = ctx->ws_t<uint32_t>(SW * (size_t)(5 + 32));        // pair slots: 2 in, 3 out, 5.. table
        HIPCHK(hipMemcpyAsync(pm[j] + 2 * SW, ent + 2 * SW, SW * 4, hipMemcpyDeviceToDevice, ctx->stream));
        emit_modexp_shared(pr[j], es[k + j], 2, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
        pr[j].end();
        sg[j] = SegSpec{&mc, &pr[j], pm[j], nullptr};
        sg[j].pair = pi.consts; sg[j].pair_n0inv = mn.n0inv; sg[j].pair_h = H; sg[j].pair_lanes = lanes;
      }
      run_vm(ctx, nb, sg[0], segs >= 2 ? &sg[1] : nullptr, true, 0, segs == 3 ? &sg[2] : nullptr);
      for (int j = 0; j < segs; ++j) leave_pair_form(pm[j], 3, outs[k + j]);
      k += segs;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

// The shard of ONE rank of the sharded threshold flow (paillier_amd/dist.py): the (server, ciphertext) units u = s * batch + i
// of the contiguous, server-major range [unit_begin, unit_end) over ONE ciphertext batch.  Such a range is a partial run of its
// first server, whole runs of the servers between, a partial run of its last server -- so the ciphertext index range splits into
// at most three intervals, each wanted under a fixed SET of shares.  An interval wanted under several shares walks ONE chain of
// squarings for all of them (emit_multi_exp_shared_base: the ciphertexts are the same, only the exponents differ), an interval
// wanted under one share runs its sliding-window ladder; the intervals are program segments of one launch.  (At N = 2 a rank
// holds one server whole and half of the next: half of its ciphertexts need both exponents -- 1.5 ladders' worth of multiplies
// instead of 3 half-batch ladders; pgpu_partial_decrypt_indexed, which is handed one ciphertext row per unit, cannot see that
// two of its rows are the same ciphertext.)  out: unit_end - unit_begin rows in unit order.
int pgpu_partial_decrypt_units(const pgpu_pubkey* pk, int total_servers, int n_shares, const uint8_t* const* shares_be,
                               const size_t* share_lens, size_t batch, const uint8_t* c, size_t c_stride, size_t unit_begin,
                               size_t unit_end, uint8_t* out, size_t out_stride, int mem) {
  if (!pk || !shares_be || !share_lens) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    if (n_shares < 1 || n_shares > 256 || total_servers < 1) api_throw(PGPU_ERR_INVALID, "bad share count");
    check_batch_args(c, out, batch);
    if (unit_begin >= unit_end || unit_end > (size_t)n_shares * batch) api_throw(PGPU_ERR_INVALID, "unit range out of bounds");
    ctx->bind();
    ctx->reset_ws();
    const ModCtx& mc = pk->mn2;
    const size_t nb = round_up(batch, VM_BLOCK);
    const BigU two_delta = BigU(2) * factorial_big(total_servers);
    const int s_first = (int)(unit_begin / batch), s_last = (int)((unit_end - 1) / batch);
    std::vector<BigU> es((size_t)n_shares);
    WipeOnExit<std::vector<BigU>> wipe_es(es);
    for (int k = s_first; k <= s_last; ++k) {
      if (!shares_be[k]) api_throw(PGPU_ERR_INVALID, "null share");
      es[(size_t)k] = BigU::from_be(shares_be[k], share_lens[k]) * two_delta;       // thresholdkey.go:195
    }
    // ciphertext range of server s inside the unit range
    auto lo_of = [&](int sv) { return sv == s_first ? unit_begin - (size_t)sv * batch : (size_t)0; };
    auto hi_of = [&](int sv) { return sv == s_last ? unit_end - (size_t)sv * batch : batch; };
    const PairInfo& pi = mc.pairn;
    bool ok = pi.root && pi.c_one_pair >= 0 && ctx->use_asm && ctx->use_pair && ctx->use_shared_chain;
    for (int k = s_first; k <= s_last; ++k) ok = ok && es[(size_t)k].bit_length() >= 256;
    // intervals of the ciphertext index range and the servers that want each
    struct Interval { size_t b, e; std::vector<int> servers; };
    std::vector<Interval> ivs;
    {
      std::vector<size_t> cuts{0, batch};
      for (int k = s_first; k <= s_last; ++k) { cuts.push_back(lo_of(k)); cuts.push_back(hi_of(k)); }
      std::sort(cuts.begin(), cuts.end());
      cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
      for (size_t t = 0; t + 1 < cuts.size(); ++t) {
        Interval iv{cuts[t], cuts[t + 1], {}};
        for (int k = s_first; k <= s_last; ++k)
          if (lo_of(k) <= iv.b && iv.e <= hi_of(k)) iv.servers.push_back(k);
        if (!iv.servers.empty()) ivs.push_back(iv);
      }
    }
    if (!ok || ivs.size() > 3) {
      // no pair kernel for this key (or a range no rank of the sharded flow produces): server after server
      for (int k = s_first; k <= s_last; ++k) {
        const size_t b = lo_of(k), e = hi_of(k), cnt = e - b, nbk = round_up(cnt, VM_BLOCK);
        ModexpPlan pl = modexp_alloc(ctx, mc, nbk, 32);
        unpack_mod(ctx, mc, c + b * c_stride, c_stride, cnt, mem, pl.in(), nbk);
        modexp_shared_run(ctx, mc, pl, es[(size_t)k], false, false, true);
        pack_result(ctx, pl.out(), mc.WT, nbk, cnt, out + ((size_t)k * batch + b - unit_begin) * out_stride, out_stride, mc.nbytes, mem);
      }
      HIPCHK(hipStreamSynchronize(ctx->stream));
      return;
    }
    const ModCtx& mn = *pi.root;
    const int H = mn.WT, W2 = mc.WT;
    const size_t SW = (size_t)W2 * nb;
    uint32_t* ent = ctx->ws_t<uint32_t>(SW * 4);            // generic slots: 0 x, 1 -, 2 digits (X0 | X1), 3 X
    unpack_mod(ctx, mc, c, c_stride, batch, mem, ent, nb);
    pair_enter(ctx, mc, ent, nb);
    size_t longest = 0;
    for (auto& iv : ivs) longest = std::max(longest, iv.e - iv.b);
    const size_t nbs = round_up(longest, VM_BLOCK), SWs = (size_t)W2 * nbs;
    const size_t lanes_target = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
    int lanes = (ivs.size() * nbs * 2 >= lanes_target || !(H % 2 == 0 && vm_asm_available(H / 2, 64))) ? 2 : 4;
    // A shard so small that even four lanes per number leave SIMDs empty is bound by the LATENCY of one ladder: eight lanes per
    // number (GenQ8: 76-limb digits in four lanes each, 38 multiplies a row and lane instead of 74) while every wave still has
    // a SIMD of its own.  The digits change radix on the way in and out (R_74 <-> R_76: one product each, inside the program).
    if (lanes == 4 && pi.consts8 && ctx->use_lanes8 && ivs.size() * nbs * 8 <= lanes_target) lanes = 8;
    const int Hk = lanes == 8 ? pi.h8 : H;                  // limbs of a digit in the kernel's slots
    const size_t SWk = (size_t)2 * Hk * nbs;                // words of a kernel slot
    const int w = 7;
    const uint32_t K = 1u << (w - 1);
    uint32_t* pm[3];
    Prog pr[3];
    SegSpec sg[3];
    uint32_t out0[3];
    for (size_t t = 0; t < ivs.size(); ++t) {
      const Interval& iv = ivs[t];
      const size_t S = iv.servers.size();
      if (lanes == 8) { pr[t].op(VM_LOAD, 2); pr[t].op(VM_MULC, 0); pr[t].op(VM_STORE, 2); }       // radix R_74 -> R_76
      if (S >= 2) {
        // slots: 2 in, 3 bp, 4 run, 5 acc, 6.. results, then the buckets (64 per server)
        out0[t] = 6;
        const uint32_t B0 = 6 + (uint32_t)S;
        pm[t] = ctx->ws_t<uint32_t>(SWk * (size_t)(B0 + S * K));
        std::vector<BigU> ev;
        WipeOnExit<std::vector<BigU>> wipe_ev(ev);
        for (int k : iv.servers) ev.push_back(es[(size_t)k]);
        emit_multi_exp_shared_base(pr[t], ev, 2, 3, 4, 5, out0[t], B0, w, lanes == 8 ? 2u : (uint32_t)pi.c_one_pair,
                                   lanes >= 4 && ctx->use_muls);
      } else {
        out0[t] = 3;                                        // pair slots: 2 in, 3 out, 5.. table
        pm[t] = ctx->ws_t<uint32_t>(SWk * (size_t)(5 + 32));
        emit_modexp_shared(pr[t], es[(size_t)iv.servers[0]], 2, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
      }
      if (lanes == 8)
        for (size_t j = 0; j < S; ++j) {                     // radix R_76 -> R_74
          pr[t].op(VM_LOAD, out0[t] + (uint32_t)j); pr[t].op(VM_MULC, 1); pr[t].op(VM_STORE, out0[t] + (uint32_t)j);
        }
      pr[t].end();
      const size_t cnt = iv.e - iv.b;
      if (lanes == 8) {
        // digits of 74 limbs -> the kernel's digits of 76 limbs (zero-extended)
        uint32_t* in = pm[t] + 2 * SWk;
        HIPCHK(hipMemsetAsync(in, 0, SWk * 4, ctx->stream));
        launch_restride(ent + 2 * SW + iv.b, nb, cnt, nullptr, in, nbs, H, ctx->stream);
        launch_restride(ent + 2 * SW + (size_t)H * nb + iv.b, nb, cnt, nullptr, in + (size_t)Hk * nbs, nbs, H, ctx->stream);
      } else {
        launch_restride(ent + 2 * SW + iv.b, nb, cnt, nullptr, pm[t] + 2 * SWs, nbs, W2, ctx->stream);
      }
      sg[t] = SegSpec{&mc, &pr[t], pm[t], nullptr};
      sg[t].pair = lanes == 8 ? pi.consts8 : pi.consts; sg[t].pair_n0inv = mn.n0inv; sg[t].pair_h = Hk; sg[t].pair_lanes = lanes;
      if (lanes == 8) sg[t].tconsts = pi.tconsts8;
    }
    run_vm(ctx, nbs, sg[0], ivs.size() >= 2 ? &sg[1] : nullptr, true, 0, ivs.size() == 3 ? &sg[2] : nullptr);
    uint32_t* back = lanes == 8 ? ctx->ws_t<uint32_t>(SWs * 4) : nullptr;     // (pair_leave wants 74-limb digits and two scratch slots)
    for (size_t t = 0; t < ivs.size(); ++t)
      for (size_t j = 0; j < ivs[t].servers.size(); ++j) {
        const size_t u0 = (size_t)ivs[t].servers[j] * batch + ivs[t].b - unit_begin;
        const size_t cnt = ivs[t].e - ivs[t].b;
        if (lanes == 8) {
          const uint32_t* res = pm[t] + (size_t)(out0[t] + j) * SWk;
          launch_restride(res, nbs, nbs, nullptr, back, nbs, H, ctx->stream);
          launch_restride(res + (size_t)Hk * nbs, nbs, nbs, nullptr, back + (size_t)H * nbs, nbs, H, ctx->stream);
          pair_leave_and_pack(ctx, mc, back, 0, nbs, cnt, out + u0 * out_stride, out_stride, mem);
        } else {
          pair_leave_and_pack(ctx, mc, pm[t], out0[t] + (uint32_t)j, nbs, cnt, out + u0 * out_stride, out_stride, mem);
        }
      }
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_partial_decrypt_indexed(const pgpu_pubkey* pk, int total_servers, int n_shares, const uint8_t* const* shares_be,
                                 const size_t* share_lens, size_t batch, const uint8_t* c, size_t c_stride,
                                 const int32_t* share_index, uint8_t* out, size_t out_stride, int mem) {
  if (!pk || !shares_be || !share_lens || !share_index) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    check_batch_args(c, out, batch);
    if (total_servers < 1 || n_shares < 1 || n_shares > 4096) api_throw(PGPU_ERR_INVALID, "bad share count");
    for (size_t i = 0; i < batch; ++i)
      if (share_index[i] < 0 || share_index[i] >= n_shares) api_throw(PGPU_ERR_INVALID, "share index out of range");
    ctx->bind();
    ctx->reset_ws();
    const ModCtx& mc = pk->mn2;
    const size_t nb = round_up(batch, VM_BLOCK);
    // thresholdkey.go:195: exp = Share * (2 * delta) for every share; the ladder takes them as per-number exponents, so the
    // units of SEVERAL servers share one launch (a few thousand ciphertexts per server cannot fill the chip on their own)
    const BigU two_delta = BigU(2) * factorial_big(total_servers);
    std::vector<BigU> exps_big;
    WipeOnExit<std::vector<BigU>> wipe_exps(exps_big);
    size_t ebits = 1;
    for (int k = 0; k < n_shares; ++k) {
      if (!shares_be[k]) api_throw(PGPU_ERR_INVALID, "null share");
      exps_big.push_back(BigU::from_be(shares_be[k], share_lens[k]) * two_delta);
      ebits = std::max(ebits, exps_big.back().bit_length());
    }
    // A shard of the sharded threshold flow is a contiguous, server-major range of units: one, two or three RUNS of units
    // with the same share.  Those are shared-exponent ladders (sliding windows: 618 products where the per-unit form below
    // needs 1 026 and a gather each), side by side as program segments of ONE launch.
    {
      struct Run { int share; size_t b, e; };
      std::vector<Run> runs;
      for (size_t i = 0; i < batch && runs.size() <= 3;) {
        size_t j = i;
        while (j < batch && share_index[j] == share_index[i]) ++j;
        runs.push_back({share_index[i], i, j});
        i = j;
      }
      const PairInfo& pi = mc.pairn;
      bool ok = runs.size() <= 3 && !runs.empty() && runs.back().e == batch && pi.root && ctx->use_asm && ctx->use_pair &&
                ctx->use_shared_chain;
      for (auto& r : runs) ok = ok && exps_big[(size_t)r.share].bit_length() >= 256;
      if (ok) {
        const ModCtx& mn = *pi.root;
        const int H = mn.WT, W2 = mc.WT;
        const size_t SW = (size_t)W2 * nb;
        uint32_t* ent = ctx->ws_t<uint32_t>(SW * 4);
        unpack_mod(ctx, mc, c, c_stride, batch, mem, ent, nb);
        pair_enter(ctx, mc, ent, nb);
        size_t longest = 0;
        for (auto& r : runs) longest = std::max(longest, r.e - r.b);
        const size_t nbs = round_up(longest, VM_BLOCK), SWs = (size_t)W2 * nbs;
        const size_t lanes_target = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
        const int lanes = (runs.size() * nbs * 2 >= lanes_target || !(H % 2 == 0 && vm_asm_available(H / 2, 64))) ? 2 : 4;
        uint32_t* pm[3];
        Prog pr[3];
        SegSpec sg[3];
        for (size_t k = 0; k < runs.size(); ++k) {
          pm[k] = ctx->ws_t<uint32_t>(SWs * (size_t)(5 + 32));      // pair slots: 2 in, 3 out, 5.. table
          launch_restride(ent + 2 * SW + runs[k].b, nb, runs[k].e - runs[k].b, nullptr, pm[k] + 2 * SWs, nbs, W2, ctx->stream);
          emit_modexp_shared(pr[k], exps_big[(size_t)runs[k].share], 2, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
          pr[k].end();
          sg[k] = SegSpec{&mc, &pr[k], pm[k], nullptr};
          sg[k].pair = pi.consts; sg[k].pair_n0inv = mn.n0inv; sg[k].pair_h = H; sg[k].pair_lanes = lanes;
        }
        run_vm(ctx, nbs, sg[0], runs.size() >= 2 ? &sg[1] : nullptr, true, 0, runs.size() == 3 ? &sg[2] : nullptr);
        for (size_t k = 0; k < runs.size(); ++k)
          pair_leave_and_pack(ctx, mc, pm[k], 3, nbs, runs[k].e - runs[k].b, out + runs[k].b * out_stride, out_stride, mem);
        for (auto& e : exps_big) wipe_vec(e.d);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return;
      }
    }
    const int we = (int)((ebits + LB - 1) / LB);
    std::vector<uint32_t> table;
    for (auto& e : exps_big) {
      auto l = e.to_limbs(LB, (size_t)we);
      table.insert(table.end(), l.begin(), l.end());
      wipe_vec(l);
      wipe_vec(e.d);
    }
    uint32_t* d_table = ctx->upload_words(table);
    wipe_vec(table);
    std::vector<uint32_t> idx(share_index, share_index + batch);
    const int32_t* d_idx = (const int32_t*)ctx->upload_words(idx);
    uint32_t* exps = ctx->ws_t<uint32_t>((size_t)we * nb);
    launch_gather_rows(d_table, we, d_idx, batch, exps, nb, ctx->stream);
    ModexpPlan pl = modexp_alloc(ctx, mc, nb, 16);
    unpack_mod(ctx, mc, c, c_stride, batch, mem, pl.in(), nb);
    modexp_perlane_run(ctx, mc, pl, exps, we, false, false);
    pack_result(ctx, pl.out(), mc.WT, nb, batch, out, out_stride, mc.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_combine_partial_decryptions(const pgpu_pubkey* pk, int total_servers, int threshold, int n_shares, const int* ids,
                                     size_t batch, const uint8_t* const* partials, size_t stride, uint8_t* m,
                                     size_t m_stride, int mem, int32_t* status) {
  if (!pk || !ids || !partials) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    // thresholdkey.go:77-89 verifyPartialDecryptions
    if (n_shares < threshold) api_throw(PGPU_ERR_THRESHOLD, "Threshold not meet");
    for (int i = 0; i < n_shares; ++i)
      for (int j = i + 1; j < n_shares; ++j)
        if (ids[i] == ids[j]) api_throw(PGPU_ERR_THRESHOLD, "two shares has been created by the same server");
    if (n_shares < 1 || total_servers < 1) api_throw(PGPU_ERR_INVALID, "bad share count");
    check_batch_args(partials[0], m, batch);
    ctx->bind();
    ctx->reset_ws();
    const ModCtx &mn = pk->mn, &mn2 = pk->mn2;
    const size_t nb = round_up(batch, VM_BLOCK);
    const size_t sw = (size_t)mn2.WT * nb;
    const BigU delta = factorial_big(total_servers);
    // Lagrange coefficients, in the reference's order of operations (thresholdkey.go:91-107)
    std::vector<SBig> two_lambda(n_shares);
    for (int i = 0; i < n_shares; ++i) {
      SBig lam;
      lam.mag = delta;
      for (int j = 0; j < n_shares; ++j)
        if (ids[j] != ids[i]) lam = sdiv_euclid(smul_small(lam, -(long long)ids[j]), (long long)ids[i] - ids[j]);
      two_lambda[i] = smul_small(lam, 2);
    }
    // slots: 0..n-1 partials, n base, n+1 numerator, n+2 denominator
    const uint32_t SB = (uint32_t)n_shares, SNUM = SB + 1, SDEN = SB + 2;
    uint32_t* mem_v = ctx->ws_t<uint32_t>(sw * (size_t)(n_shares + 3));
    for (int i = 0; i < n_shares; ++i) {
      if (!partials[i]) api_throw(PGPU_ERR_INVALID, "null partial buffer");
      unpack_mod(ctx, mn2, partials[i], stride, batch, mem, mem_v + (size_t)i * sw, nb);
    }
    // numerator = prod over lambda_i >= 0 of c_i^(2 lambda_i); denominator = prod over lambda_i < 0 of c_i^|2 lambda_i|.
    // (thresholdkey.go:132-138 inverts each negative factor separately; the product of inverses is the inverse of
    //  the product, so one inversion per ciphertext gives the same canonical residue.)
    Prog p;
    bool have_den = false;
    for (int pass = 0; pass < 2; ++pass) {
      const uint32_t acc = pass == 0 ? SNUM : SDEN;
      bool first = true;
      for (int i = 0; i < n_shares; ++i) {
        const SBig& e = two_lambda[i];
        if ((pass == 1) != e.neg) continue;
        if (e.mag.is_zero()) continue;  // Exp(x, 0) = 1: no contribution
        p.op(VM_LOAD, (uint32_t)i);
        p.op(VM_MULC, C_R2);
        p.op(VM_STORE, SB);
        emit_pow_small(p, e.mag, SB);
        if (!first) p.op(VM_MUL, acc);
        p.op(VM_STORE, acc);
        first = false;
        if (pass == 1) have_den = true;
      }
      if (first) { p.op(VM_LOADC, C_ONE_M); p.op(VM_STORE, acc); }
    }
    // leave Montgomery form: denominator -> canonical (to be inverted); numerator stays in Montgomery form
    p.op(VM_LOAD, SDEN); p.op(VM_MULC, C_ONE); p.op(VM_STORE, SDEN);
    p.end();
    SegSpec sg{&mn2, &p, mem_v, nullptr};
    run_vm(ctx, nb, sg, nullptr, true);
    BadLanes bl;
    if (have_den) {
      launch_canon(mem_v + SDEN * sw, mn2.d_nmod, mn2.WT, nb, ctx->stream);
      int32_t* d_bad = ctx->ws_t<int32_t>(nb);
      bool any_bad = false;
      uint32_t* dinv = batch_inverse(ctx, mn2, mem_v + SDEN * sw, nb, batch, d_bad, &any_bad);   // thresholdkey.go:135 ModInverse
      bl.collect(ctx, d_bad, batch, any_bad);
      HIPCHK(hipMemcpyAsync(mem_v + SDEN * sw, dinv, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    {
      Prog q;   // cprime = numerator(Montgomery) * denominator^-1 (plain)  -> plain residue
      q.op(VM_LOAD, SNUM);
      if (have_den) q.op(VM_MUL, SDEN); else q.op(VM_MULC, C_ONE);
      q.op(VM_STORE, SB);
      q.end();
      SegSpec sq{&mn2, &q, mem_v, nullptr};
      run_vm(ctx, nb, sq, nullptr, false);
      launch_canon(mem_v + SB * sw, mn2.d_nmod, mn2.WT, nb, ctx->stream);
    }
    // thresholdkey.go:143-146: L(cprime) * (4 delta^2)^-1 mod n   (combineSharesConstant, :63-66)
    pgpu_pubkey* pkm = const_cast<pgpu_pubkey*>(pk);
    int cidx = -1;
    for (auto& pr : pkm->combine_consts) if (pr.first == total_servers) cidx = pr.second;
    BigU cconst;
    if (!hostbig::modinv((BigU(4) * delta * delta) % pk->N, pk->N, cconst))
      api_throw(PGPU_ERR_NOT_INVERTIBLE, "4*delta^2 is not invertible mod n");
    if (cidx < 0) {
      cidx = pkm->mn.add_const(pkm->mn.to_mont(cconst));
      pkm->mn.upload();
      pkm->combine_consts.push_back({total_servers, cidx});
    }
    uint32_t* negc = ctx->upload_words(((pk->N - cconst) % pk->N).to_limbs(LB, mn.WT));
    uint32_t* res = L_times_const(ctx, pk, mem_v + SB * sw, nb, batch, pk->mn, cidx, negc);
    pack_result(ctx, res, mn.WT, nb, batch, m, m_stride, mn.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    bl.finish(status, batch);   // a share that is not a unit modulo n^2 (mpz_invert undefined in the reference): flagged per lane
  });
}

int pgpu_random_oracle_digest(pgpu_ctx* ctx, int nparts, const uint8_t* const* parts, const size_t* strides, size_t batch,
                              uint8_t* digests, int mem) {
  if (!ctx || !parts || !strides || !digests) return fail(PGPU_ERR_INVALID, "null argument");
  return guarded([&] {
    if (nparts < 0 || nparts > 6) api_throw(PGPU_ERR_INVALID, "0..6 transcript parts");
    if (batch == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
    ctx->bind();
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    const uint32_t* dp[6];
    int dw[6];
    for (int i = 0; i < nparts; ++i) {
      if (!parts[i]) api_throw(PGPU_ERR_INVALID, "null part");
      dw[i] = std::max<int>(1, (int)((strides[i] * 8 + LB - 1) / LB));
      uint32_t* l = ctx->ws_t<uint32_t>((size_t)dw[i] * nb);
      unpack_operand(ctx, parts[i], strides[i], strides[i], batch, mem, l, dw[i], nb);
      dp[i] = l;
    }
    uint32_t* dg = ctx->ws_t<uint32_t>(8 * nb);
    launch_sha256_transcript(dp, dw, nparts, nb, batch, dg, nullptr, ctx->stream);
    // digests: 32 bytes each, big-endian words
    std::vector<uint32_t> h(8 * nb);
    HIPCHK(hipMemcpyAsync(h.data(), dg, 8 * nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<uint8_t> outb(32 * batch);
    for (size_t g = 0; g < batch; ++g)
      for (int i = 0; i < 8; ++i) {
        uint32_t v = h[(size_t)i * nb + g];
        outb[g * 32 + 4 * i + 0] = (uint8_t)(v >> 24); outb[g * 32 + 4 * i + 1] = (uint8_t)(v >> 16);
        outb[g * 32 + 4 * i + 2] = (uint8_t)(v >> 8);  outb[g * 32 + 4 * i + 3] = (uint8_t)v;
      }
    if (mem == PGPU_MEM_HOST) memcpy(digests, outb.data(), outb.size());
    else HIPCHK(hipMemcpy(digests, outb.data(), outb.size(), hipMemcpyHostToDevice));
  });
}

}  // extern "C"

namespace {
// x^(per-number exponent, W2 limbs) * y^(n^2) mod n^3 as ONE interleaved ladder (emit_modexp_dual): the verifier's
// check^(E^n) * F^(n^2) (ddleq.go:143-152) and NestedRandomize's ct^(a^n) * b^(n^2) (operations.go:108-114).
// x, y: W3-limb arrays (any value below R); returns the canonical result (W3 limbs, stride nb).
uint32_t* dual_pow_n3(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint32_t* x, const uint32_t* exps, const uint32_t* y, size_t nb) {
  const ModCtx &mn2 = pk->mn2, &mn3 = *pk->mn3;
  const int W2 = mn2.WT, W3 = mn3.WT;
  const bool use3 = triple_usable(ctx, mn3) && (uint64_t)nb * (W3 + 4) * 4 * 33 < (1ull << 32);
  ModexpPlan pc = modexp_alloc(ctx, mn3, nb, use3 ? 0 : 48);   // slots: 0 x, 1 y, 2 tmp, 3 out, 5..20 / 21..52 the two tables
  HIPCHK(hipMemcpyAsync(pc.in(), x, pc.slot_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(pc.in() + pc.slot_words, y, pc.slot_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
  if (use3 && ctx->use_lift) {
    // x^e y^(n^2) = (x^(e1) y^n)^n x^(e0) with e = e0 + e1 n, and W^n mod n^3 depends on W mod n^2 only (the lift of
    // encrypt_core: x = x' mod n^k implies x^n = x'^n mod n^(k+1), any integers).  So: W = x^(e1) y^n modulo n^2 -- an
    // interleaved ladder of 2 048 squarings on the pair kernel, half the price of squarings modulo n^3 -- and then
    // W^n x^(e0) modulo n^3, an interleaved ladder of 2 048 squarings where the literal form needs 4 096.
    const ModCtx& mn = pk->mn;
    const int W1 = mn.WT;
    const size_t S1 = (size_t)W1 * nb;
    uint32_t* e0 = ctx->ws_t<uint32_t>(S1);
    uint32_t* e1 = ctx->ws_t<uint32_t>(S1);
    uint32_t* tb = ctx->ws_t<uint32_t>((size_t)W2 * nb);
    reduce_mod(ctx, mn, exps, W2, e0, nb);
    launch_div_exact(exps, W2, 0, e0, W1, tb, pk->ninv2k.d, mn.d_nmod, W1, e1, W1, nb, nb, nullptr, 0, ctx->stream);
    uint32_t* x2 = ctx->ws_t<uint32_t>((size_t)W2 * nb);
    uint32_t* y2 = ctx->ws_t<uint32_t>((size_t)W2 * nb);
    reduce_mod(ctx, mn2, x, W3, x2, nb);
    reduce_mod(ctx, mn2, y, W3, y2, nb);
    // x enters digit form beside the W ladder (a side lane); W itself is handed over in pair form -- (a0, a1, 0) is the digit form
    // of a representative of W mod n^2, which is all the lift needs (modexp_triple) -- where the pair kernel and the digit
    // kernel share the root n
    const int wb = triple_window_bits(nb, mn3.triple.root->WT);
    const uint32_t tab2 = 5 + (uint32_t)perlane_table_slots(wb, wb == 5);
    TriplePlan tp = triple_alloc(ctx, mn3, nb, (int)tab2 + (1 << (dual_sliding_bits(wb) - 1)));
    const bool hand = ctx->use_handover && mn2.pairn.root && mn2.pairn.root->WT == mn3.triple.root->WT;
    Fork ft(ctx, 3);
    ft.chain(2);                                   // (lane 2: dual_pow_pair forks its own two entries over lanes 0 and 1)
    triple_enter(ctx, mn3, pc.in(), tp, 0);
    ft.chain(0);
    uint32_t* raw = nullptr;
    uint32_t* wv = dual_pow_pair(ctx, mn2, x2, e1, W1, y2, pk->N, nb, hand ? &raw : nullptr);
    if (wv) {
      if (raw) {
        triple_from_pair(ctx, raw, tp, 1);
      } else {
        launch_copy_limbs(wv, 0, W2, pc.in() + pc.slot_words, W3, nb, ctx->stream);    // slot 1 <- W, zero-extended
        triple_enter(ctx, mn3, pc.in() + pc.slot_words, tp, 1);
      }
      ft.join();
      Prog pd;
      emit_modexp_dual(pd, W1, pk->N, 0, 1, 2, 3, 5, tab2, 0, wb, wb == 5);
      pd.end();
      triple_run(ctx, mn3, tp, pd, triple_windows(ctx, e0, W1, nb, wb));
      triple_exit(ctx, mn3, tp, 3, pc.out(), nullptr);
      return pc.out();
    }
  }
  if (use3) {
    // the interleaved ladder on the three-digit kernel: residues modulo n^3 as a0 + a1 n + a2 n^2
    const int wb = triple_window_bits(nb, mn3.triple.root->WT);       // 7-bit (or 5-bit) windows of the per-number exponent
    const uint32_t tab2 = 5 + (uint32_t)perlane_table_slots(wb, wb == 5);
    TriplePlan tp = triple_alloc(ctx, mn3, nb, (int)tab2 + (1 << (dual_sliding_bits(wb) - 1)));   // slots: 0 x, 1 y, 2 tmp, 3 out, 5.. / tab2.. the tables
    Fork ft(ctx);
    ft.chain(0);
    triple_enter(ctx, mn3, pc.in(), tp, 0);
    ft.chain(1);
    triple_enter(ctx, mn3, pc.in() + pc.slot_words, tp, 1);
    ft.join();
    Prog pd;
    emit_modexp_dual(pd, W2, mn2.N, 0, 1, 2, 3, 5, tab2, 0, wb, wb == 5);
    pd.end();
    triple_run(ctx, mn3, tp, pd, triple_windows(ctx, exps, W2, nb, wb));
    triple_exit(ctx, mn3, tp, 3, pc.out(), nullptr);
  } else {
    Prog pd;
    emit_modexp_dual(pd, W2, mn2.N, 0, 1, 2, 3, 5, 21);
    pd.end();
    SegSpec sd{&mn3, &pd, pc.mem, exps};
    run_vm(ctx, nb, sd, nullptr, true);
    launch_canon(pc.out(), mn3.d_nmod, W3, nb, ctx->stream);
  }
  return pc.out();
}
}  // namespace

extern "C" {

int pgpu_nested_randomize_with_ab(const pgpu_pubkey* pk, size_t batch, const uint8_t* ct, size_t ct_stride, const uint8_t* a,
                                  const uint8_t* b, size_t ab_stride, uint8_t* out, size_t out_stride, int mem) {
  if (!pk || !ct || !a || !b || !out) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    if (batch == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
    if (!pk->mn3) api_throw(PGPU_ERR_UNSUPPORTED, "n^3 is wider than the built kernels");
    ctx->bind();
    ctx->reset_ws();
    const ModCtx &mn2 = pk->mn2, &mn3 = *pk->mn3;
    const int W2 = mn2.WT, W3 = mn3.WT;
    const size_t nb = round_up(batch, VM_BLOCK);
    // an = a^n mod n^2 (operations.go:108); r = ct^an * b^(n^2) mod n^3 (:109-114)
    ModexpPlan pa_ = modexp_alloc(ctx, mn2, nb, 32);
    unpack_mod(ctx, mn2, a, ab_stride, batch, mem, pa_.in(), nb);
    modexp_shared_run(ctx, mn2, pa_, pk->N, false, false, true);
    uint32_t* x = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    uint32_t* y = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    unpack_mod(ctx, mn3, ct, ct_stride, batch, mem, x, nb);
    unpack_mod(ctx, mn3, b, ab_stride, batch, mem, y, nb);
    uint32_t* res = dual_pow_n3(ctx, pk, x, pa_.out(), y, nb);
    pack_result(ctx, res, W3, nb, batch, out, out_stride, mn3.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    (void)W2;
  });
}

int pgpu_ddleq_verify(const pgpu_pubkey* pk, size_t batch, const uint8_t* ct1, const uint8_t* ct2, size_t ct_stride,
                      const uint8_t* x, const uint8_t* y, size_t xy_stride, const uint8_t* alpha, size_t alpha_stride,
                      const uint8_t* e, size_t e_stride, const uint8_t* f, size_t f_stride, int32_t* ok, int mem) {
  if (!pk || !ct1 || !ct2 || !x || !y || !alpha || !e || !f || !ok) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    if (batch == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
    if (!pk->mn3) api_throw(PGPU_ERR_UNSUPPORTED, "n^3 is wider than the built kernels");
    ctx->bind();
    ctx->reset_ws();
    const ModCtx &mn2 = pk->mn2, &mn3 = *pk->mn3;
    const int W2 = mn2.WT, W3 = mn3.WT;
    const size_t nb = round_up(batch, VM_BLOCK);
    auto wof = [](size_t stride) { return std::max<int>(1, (int)((stride * 8 + LB - 1) / LB)); };
    // transcript operands exactly as given (ddleq.go:136: RandomOracleBit(ct1, ct2, X, Y, Alpha); ct1 is skipped by
    // random_oracle.go:24-26)
    const int wc = std::max(wof(ct_stride), W3), wxy = wof(xy_stride), wa = std::max(wof(alpha_stride), W3);
    uint32_t* c1 = ctx->ws_t<uint32_t>((size_t)wc * nb);
    uint32_t* c2 = ctx->ws_t<uint32_t>((size_t)wc * nb);
    uint32_t* xl = ctx->ws_t<uint32_t>((size_t)wxy * nb);
    uint32_t* yl = ctx->ws_t<uint32_t>((size_t)wxy * nb);
    uint32_t* al = ctx->ws_t<uint32_t>((size_t)wa * nb);
    unpack_operand(ctx, ct1, ct_stride, ct_stride, batch, mem, c1, wc, nb);
    unpack_operand(ctx, ct2, ct_stride, ct_stride, batch, mem, c2, wc, nb);
    unpack_operand(ctx, x, xy_stride, xy_stride, batch, mem, xl, wxy, nb);
    unpack_operand(ctx, y, xy_stride, xy_stride, batch, mem, yl, wxy, nb);
    unpack_operand(ctx, alpha, alpha_stride, alpha_stride, batch, mem, al, wa, nb);
    int32_t* chal = ctx->ws_t<int32_t>(nb);
    HIPCHK(hipMemsetAsync(chal, 0, nb * 4, ctx->stream));
    const uint32_t* parts[4] = {c2, xl, yl, al};
    const int widths[4] = {wc, wxy, wxy, wa};
    launch_sha256_transcript(parts, widths, 4, nb, batch, nullptr, chal, ctx->stream);
    if (wc != W3) api_throw(PGPU_ERR_INVALID, "ciphertext stride must be the byte length of n^3");
    // en = E^n mod n^2 ; fn2 = F^(n^2) mod n^3                                   (ddleq.go:143-144)
    ModexpPlan pe = modexp_alloc(ctx, mn2, nb, 32);
    // E is a residue modulo n^2 in every honest proof (ddleq.go:94-99); a wider field (up to twice the width) is reduced
    // by the ordinary kernel's two-chunk entry, the usual width takes the pair-kernel path
    const bool ewide = e_stride * 8 > (size_t)LB * W2;
    unpack_operand(ctx, e, e_stride, std::min(e_stride, 2 * mn2.nbytes), batch, mem, pe.in(), ewide ? 2 * W2 : W2, nb);
    modexp_shared_run(ctx, mn2, pe, pk->N, ewide, false, true);
    // check = chalBit ? ct2 : ct1 ; check^en * F^(n^2) mod n^3 == alpha           (ddleq.go:138-152)
    // one interleaved ladder: the squarings of check^en and of F^(n^2) are shared (emit_modexp_dual)
    if (f_stride * 8 > (size_t)LB * W3 + 7) api_throw(PGPU_ERR_INVALID, "F wider than n^3");
    uint32_t* chk = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    uint32_t* fl = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    launch_select(chal, c2, c1, chk, W3, nb, ctx->stream);
    unpack_operand(ctx, f, f_stride, f_stride, batch, mem, fl, W3, nb);
    uint32_t* got = dual_pow_n3(ctx, pk, chk, pe.out(), fl, nb);
    int32_t* d_ok = ctx->ws_t<int32_t>(nb);
    if (wa != W3) api_throw(PGPU_ERR_INVALID, "alpha stride must be the byte length of n^3");
    launch_equal(got, al, W3, nb, batch, d_ok, ctx->stream);
    HIPCHK(hipMemcpyAsync(ok, d_ok, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

}  // extern "C"

namespace {

// fixed-base comb table of `base` modulo n^2 covering exponents of up to `ebits` bits; returns (first const index, windows)
const pgpu_pubkey::FixedBase& ensure_fixed_base(pgpu_pubkey* pk, const BigU& base_in, size_t ebits) {
  ModCtx& mc = pk->mn2;
  BigU base = base_in % mc.N;
  const int nwin = (int)((ebits + 3) / 4);
  for (auto& f : pk->fixed_bases)
    if (f.base == base && f.nwin >= nwin) return f;
  pgpu_pubkey::FixedBase f;
  f.base = base;
  f.nwin = nwin;
  f.idx = (int)mc.consts.size();
  const BigU rmod = mc.R % mc.N;
  BigU b = base;
  for (int i = 0; i < nwin; ++i) {
    BigU cur(1);
    for (int d = 0; d < 16; ++d) {
      mc.consts.push_back(hostbig::mulmod(cur, rmod, mc.N));
      cur = hostbig::mulmod(cur, b, mc.N);
    }
    b = cur;
  }
  mc.upload();
  pk->fixed_bases.push_back(f);
  return pk->fixed_bases.back();
}

// x^e mod n^2 for a uniform base with a comb table and per-number exponents (limb-major [we][nb]); result canonical in `out`
void comb_pow(pgpu_ctx* ctx, const ModCtx& mc, const pgpu_pubkey::FixedBase& fb, const uint32_t* exps, int we, size_t nb,
              uint32_t* out) {
  if (we * 7 > fb.nwin + 6) api_throw(PGPU_ERR_INVALID, "fixed-base table narrower than the exponent");   // callers size the table
  const int nwin = std::min(fb.nwin, we * 7);
  size_t sw = (size_t)mc.WT * nb;
  uint32_t* memv = ctx->ws_t<uint32_t>(sw);
  Prog p;
  p.op(VM_LOADC, C_ONE_M);
  for (int i = 0; i < nwin; ++i) p.op(VM_MULCV, (uint32_t)i, (uint32_t)fb.idx);
  p.op(VM_MULC, C_ONE);
  p.op(VM_STORE, 0);
  p.end();
  SegSpec sg{&mc, &p, memv, exps};
  run_vm(ctx, nb, sg, nullptr, true);
  launch_canon(memv, mc.d_nmod, mc.WT, nb, ctx->stream);
  HIPCHK(hipMemcpyAsync(out, memv, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
}

// out = base[i]^(e[i]) mod N, per-number base (WT limbs) and per-number exponent (we limbs); canonical
void perlane_pow(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* base, const uint32_t* exps, int we, size_t nb, uint32_t* out) {
  ModexpPlan pl = modexp_alloc(ctx, mc, nb, 16);
  HIPCHK(hipMemcpyAsync(pl.in(), base, (size_t)mc.WT * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
  modexp_perlane_run(ctx, mc, pl, exps, we, false, false);
  HIPCHK(hipMemcpyAsync(out, pl.out(), (size_t)mc.WT * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
}

// out = a * b mod N (canonical operands, WT limbs)
void modmul_arrays(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* a, const uint32_t* b, size_t nb, uint32_t* out) {
  size_t sw = (size_t)mc.WT * nb;
  uint32_t* memv = ctx->ws_t<uint32_t>(sw * 3);
  HIPCHK(hipMemcpyAsync(memv, a, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(memv + sw, b, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
  Prog p;
  p.op(VM_LOAD, 0); p.op(VM_MULC, C_R2); p.op(VM_MUL, 1); p.op(VM_STORE, 2); p.end();
  SegSpec sg{&mc, &p, memv, nullptr};
  run_vm(ctx, nb, sg, nullptr, false);
  launch_canon(memv + 2 * sw, mc.d_nmod, mc.WT, nb, ctx->stream);
  HIPCHK(hipMemcpyAsync(out, memv + 2 * sw, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
}

// Shared part of prove and verify: the Fiat-Shamir hash E = SHA-256(a || b || c^4 || c_i^2) over UNREDUCED c^4, c_i^2
// (thresholdkey.go:241,248,319-326).  c, ci: canonical W2-limb arrays as given by the caller.  Returns digest words [8][nb].
uint32_t* zkp_hash(pgpu_ctx* ctx, int W2, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* ci, size_t nb,
                   size_t count) {
  uint32_t* c2 = ctx->ws_t<uint32_t>((size_t)2 * W2 * nb);
  uint32_t* c4 = ctx->ws_t<uint32_t>((size_t)4 * W2 * nb);
  uint32_t* ci2 = ctx->ws_t<uint32_t>((size_t)2 * W2 * nb);
  launch_mul_plain(c, W2, c, W2, c2, nb, ctx->stream);
  launch_mul_plain(c2, 2 * W2, c2, 2 * W2, c4, nb, ctx->stream);
  launch_mul_plain(ci, W2, ci, W2, ci2, nb, ctx->stream);
  const uint32_t* parts[4] = {a, b, c4, ci2};
  const int widths[4] = {W2, W2, 4 * W2, 2 * W2};
  uint32_t* dg = ctx->ws_t<uint32_t>(8 * nb);
  launch_sha256_transcript(parts, widths, 4, nb, count, dg, nullptr, ctx->stream);
  return dg;
}

}  // namespace

namespace {

// out = base^e mod N for a shared exponent; base: `wb` limbs (<= 2 WT); canonical result
void shared_pow(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* base, int wb, const BigU& e, size_t nb, uint32_t* out) {
  ModexpPlan pl = modexp_alloc(ctx, mc, nb, 32);
  const bool wide = wb > mc.WT;
  launch_copy_limbs(base, 0, std::min(wb, mc.WT), pl.in(), mc.WT, nb, ctx->stream);
  if (wide) launch_copy_limbs(base, mc.WT, wb - mc.WT, pl.in() + pl.slot_words, mc.WT, nb, ctx->stream);
  modexp_shared_run(ctx, mc, pl, e, wide, false, true);
  HIPCHK(hipMemcpyAsync(out, pl.out(), (size_t)mc.WT * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
}

// W = x^(per-number exponent r1, `we` limbs) [* y^(s1)] modulo p^2 (half 0) and modulo q^2 (half 1): interleaved ladders in
// pair form on the one-lane pair kernel, both halves in ONE two-segment launch.  xs / ys: canonical residues modulo p^2 / q^2
// (mp2.WT limbs, stride nb).  outs[half]: canonical results.  False when the one-lane pair kernel does not serve this key.
bool pow_p2_multi_crt(const pgpu_seckey* sk, const uint32_t* const xs[2], const uint32_t* const r1[2], int we,
                      const uint32_t* const ys[2], const BigU s1[2], size_t nb, uint32_t* outs[2],
                      const uint32_t* const xs_b[2] = nullptr, const uint32_t* const r1_b[2] = nullptr, bool raw = false) {
  // raw: outs[half] = the results still in pair form (a0 | a1, 2H limbs, stride nb) for a caller that continues modulo prime^3
  // on the digit kernel, where (a0, a1, 0) is their digit form (modexp_triple); else canonical residues modulo prime^2
  // xs_b / r1_b: a SECOND base with per-number exponents (the response of the DDLEQ prover: s^(e_s) b^(e_b)), not together with ys
  pgpu_ctx* ctx = sk->ctx;
  if (!(sk->has_pair && sk->pair_lanes == 1 && sk->c_onep_p2 >= 0 && sk->c_onep_q2 >= 0 && sk->c_rh_p2 >= 0 && ctx->use_asm &&
        ctx->use_pair && sk->mp2.WT == 2 * sk->mp.WT && sk->mq2.WT == 2 * sk->mq.WT))
    return false;
  // one lane per number when the two halves fill the chip that way, else two (as Decrypt chooses)
  const size_t lanes_target = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
  // (two lanes per number pay while they still leave every wave a SIMD of its own: above half a wave per SIMD at one lane, two
  // lanes are two waves on most SIMDs -- 1.15 x the one-lane ladder -- and the one-lane kernel needs fewer multiplies)
  const int lanes = (sk->pair_small2 && nb * 4 <= lanes_target) ? 2 : 1;
  const int H = sk->mp.WT, W2 = sk->mp2.WT;
  const size_t S1 = (size_t)H * nb, S2 = (size_t)W2 * nb;
  const int wb = 4;                                           // per-number windows: VM_MULV
  if ((uint64_t)nb * W2 * 4 * 18 >= (1ull << 32)) return false;
  // per-number window tables number-major where the kernel has VM_STORET / VM_MULVT (one lane per number, 37-limb primes): a
  // gather then reads 296 contiguous bytes per lane instead of 74 dwords in 74 different sectors
  const bool nm4 = ctx->use_nm4 && lanes == 1 && H == 37;
  const uint32_t TAB1 = 5, TAB2 = TAB1 + (1u << wb) + (nm4 ? 1u : 0u);
  if (xs_b && (ys || !r1 || !r1_b)) return false;
  uint32_t* mem[2];
  Prog pr[2];
  const uint32_t* dig[2] = {r1 ? r1[0] : nullptr, r1 ? r1[1] : nullptr};
  Fork in(ctx, 4);                                            // the entry chains of both halves and both operands side by side
  for (int half = 0; half < 2; ++half) {
    const ModCtx &m1 = half ? sk->mq : sk->mp, &m2 = half ? sk->mq2 : sk->mp2;
    mem[half] = ctx->ws_t<uint32_t>(S2 * (size_t)(TAB2 + 64));      // 0 x, 1 y, 2 tmp, 3 out, 5.. / TAB2.. the tables (+ 1 each: x limb-major)
    for (int k = 0; k < ((ys || xs_b) ? 2 : 1); ++k) {
      in.chain(2 * half + k);
      uint32_t* ent = ctx->ws_t<uint32_t>(S2 * 4);                   // (buffers of the chain's own)
      uint32_t* x0 = ctx->ws_t<uint32_t>(S1);
      uint32_t* tb = ctx->ws_t<uint32_t>(S2);
      // pair-form entry: X = v R_H mod prime^2, then its digits X0 + X1 prime
      HIPCHK(hipMemcpyAsync(ent, k ? (ys ? ys[half] : xs_b[half]) : xs[half], S2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
      Prog a;
      a.op(VM_LOAD, 0); a.op(VM_MULC, C_R2); a.op(VM_MULC, (uint32_t)(half ? sk->c_rh_q2 : sk->c_rh_p2)); a.op(VM_STORE, 3); a.end();
      SegSpec sa{&m2, &a, ent, nullptr};
      run_vm(ctx, nb, sa, nullptr, false);
      launch_canon(ent + 3 * S2, m2.d_nmod, W2, nb, ctx->stream);
      reduce_mod(ctx, m1, ent + 3 * S2, W2, x0, nb);
      uint32_t* slot = mem[half] + (size_t)k * S2;
      launch_div_exact(ent + 3 * S2, W2, 0, x0, H, tb, (half ? sk->qinv2k : sk->pinv2k).d, m1.d_nmod, H, slot + S1, H, nb, nb, nullptr, 0,
                       ctx->stream);
      HIPCHK(hipMemcpyAsync(slot, x0, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    std::vector<SharedBase> sh;
    if (ys) sh.push_back(SharedBase{s1[half], 1, TAB2});
    std::vector<PerNumberBase> pn;
    if (r1) pn.push_back(PerNumberBase{we, 0, TAB1, 0});
    if (xs_b) {
      // the two exponents of a number one after the other in the rows of `digits` (as pow_n3_crt_two keeps them)
      pn.push_back(PerNumberBase{we, 1, TAB2, (uint32_t)perlane_windows(we, wb)});
      uint32_t* d2 = ctx->ws_t<uint32_t>((size_t)2 * we * nb);
      HIPCHK(hipMemcpyAsync(d2, r1[half], (size_t)we * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(d2 + (size_t)we * nb, r1_b[half], (size_t)we * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
      dig[half] = d2;
    }
    emit_modexp_multi(pr[half], pn, wb, sh, 2, 3, (uint32_t)(half ? sk->c_onep_q2 : sk->c_onep_p2), nm4);
    pr[half].end();
  }
  in.join();
  {
    SegSpec sp{&sk->mp2, &pr[0], mem[0], dig[0]}, sq{&sk->mq2, &pr[1], mem[1], dig[1]};
    sp.pair = sk->pair_p.d; sp.pair_n0inv = sk->mp.n0inv; sp.pair_h = H; sp.pair_lanes = lanes;
    sq.pair = sk->pair_q.d; sq.pair_n0inv = sk->mq.n0inv; sq.pair_h = H; sq.pair_lanes = lanes;
    run_vm(ctx, nb, sp, &sq, true);
  }
  if (raw) {
    outs[0] = mem[0] + 3 * S2;
    outs[1] = mem[1] + 3 * S2;
    return true;
  }
  Fork out(ctx);
  for (int half = 0; half < 2; ++half) {
    out.chain(half);
    const ModCtx& m2 = half ? sk->mq2 : sk->mp2;
    uint32_t* mm = mem[half];
    launch_mul_const_add(mm + 3 * S2 + S1, H, (half ? sk->q_limbs1 : sk->p_limbs).d, H, mm + 3 * S2, H, 0, mm + 2 * S2, W2, nb, ctx->stream);
    Prog a;
    a.op(VM_LOAD, 2); a.op(VM_MULC, (uint32_t)(half ? sk->c_rh_q2 : sk->c_rh_p2)); a.op(VM_STORE, 3); a.end();
    SegSpec sa{&m2, &a, mm, nullptr};
    run_vm(ctx, nb, sa, nullptr, false);
    launch_canon(mm + 3 * S2, m2.d_nmod, W2, nb, ctx->stream);
    outs[half] = mm + 3 * S2;
  }
  out.join();
  return true;
}

// out = base^e mod n^3 for a holder of the factorisation (the DDLEQ prover): two ladders modulo p^3 and q^3 -- half the
// width, the same exponent -- in one two-segment launch, then Garner.  2.6x fewer limb products than the ladder modulo
// n^3, the same canonical residue.  Per-number exponents (exps: we limbs each) or one shared exponent (*e).
void pow_n3_crt(const pgpu_seckey* sk, const uint32_t* base, int wb, const uint32_t* exps, int we, const BigU* e, size_t nb,
                uint32_t* out, const uint32_t* base2 = nullptr, int wb2 = 0) {
  // base2 != nullptr: out = base^(per-number exps) * base2^(*e), one interleaved ladder per half (emit_modexp_dual)
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx &mp3 = sk->mp3, &mq3 = sk->mq3;
  const int W = mp3.WT, W3 = sk->pk->mn3->WT;
  const size_t S = (size_t)W * nb;
  if (triple_usable(ctx, mp3) && triple_usable(ctx, mq3) && (uint64_t)nb * (W + 4) * 4 * 33 < (1ull << 32) &&
      (exps || base2 || e->bit_length() >= 64)) {
    // both halves on the three-digit kernel (digits modulo p and q, 37 limbs for 2048-bit keys): a squaring is 37 rows
    // where the wave-sliced 110-limb kernel has 110 -- what counts for the half-size, latency-bound batches of the response
    const int win = exps ? triple_window_bits(nb, mp3.triple.root->WT) : 5;
    const bool nm5 = exps && win == 5;          // (number-major 5-bit tables: VM_MULVT5)
    const uint32_t tab2 = 5 + (uint32_t)perlane_table_slots(win, nm5);
    const int nslots = base2 ? (int)tab2 + (1 << (dual_sliding_bits(win) - 1)) : 5 + perlane_table_slots(win, nm5);
    uint32_t* g = ctx->ws_t<uint32_t>(S * 6);       // generic slots (W limbs): 0 x_p, 1 x_q, 2 A, 3 B, 4 h, 5 scratch
    // Exponents modulo the orders of the unit groups of p^3 and q^3 (a quarter shorter than exponents modulo n^2): each half
    // gets its own reduced exponents and its own program.  PGPU_EXP_ORDER=0 (experiments) keeps the exponents as given.
    static const bool order_on = [] { const char* v = getenv("PGPU_EXP_ORDER"); return v ? atoi(v) != 0 : true; }();
    const bool reduce_e = order_on && sk->eo_p.ok && sk->eo_q.ok;
    const uint32_t* ex[2] = {exps, exps};
    int wex[2] = {we, we};
    BigU es[2];
    if (e) es[0] = es[1] = *e;
    Fork fe(ctx);                                   // (the q-half's chains of small kernels beside the p-half's, here and below)
    for (int half = 0; half < 2 && reduce_e; ++half) {
      fe.chain(half);
      const ExpOrder& eo = half ? sk->eo_q : sk->eo_p;
      if (exps && (size_t)we * LB > eo.ord.bit_length() + LB && we <= 2 * eo.modd.WT) {
        uint32_t* em = ctx->ws_t<uint32_t>((size_t)eo.modd.WT * nb);
        reduce_mod(ctx, eo.modd, exps, we, em, nb);
        uint32_t* er = ctx->ws_t<uint32_t>((size_t)eo.w * nb);
        launch_exp_order_lift(exps, we, em, eo.modd.WT, eo.m_limbs.d, eo.t, eo.minv, er, eo.w, nb, ctx->stream);
        ex[half] = er;
        wex[half] = eo.w;
      }
      if (e) {
        const BigU r = order_fixup(*e, eo.ord);
        if (r.bit_length() >= 64) es[half] = r;
      }
    }
    fe.join();
    auto garner = [&](const TriplePlan& tp, const TriplePlan& tq) {
      Fork fx(ctx);
      fx.chain(0);
      triple_exit(ctx, mp3, tp, 3, g + 0 * S, nullptr);     // x_p, canonical
      fx.chain(1);
      triple_exit(ctx, mq3, tq, 3, g + 1 * S, nullptr);     // x_q
      fx.join();
      Prog c;
      c.op(VM_LOAD, 0); c.op(VM_MULC, (uint32_t)sk->c_p3invR); c.op(VM_STORE, 3);
      c.op(VM_LOAD, 1); c.op(VM_MULC, (uint32_t)sk->c_p3invR); c.op(VM_STORE, 2);
      c.end();
      SegSpec sc{&mq3, &c, g, nullptr};
      run_vm(ctx, nb, sc, nullptr, false);
      launch_canon(g + 2 * S, mq3.d_nmod, W, nb, ctx->stream);
      launch_canon(g + 3 * S, mq3.d_nmod, W, nb, ctx->stream);
      launch_sub_mod(g + 2 * S, g + 3 * S, mq3.d_nmod, g + 4 * S, W, nb, ctx->stream);                  // h = (x_q - x_p) / p^3 mod q^3
      launch_mul_const_add(g + 4 * S, W, sk->p3_limbs.d, W, g, W, 0, out, W3, nb, ctx->stream);          // x_p + p^3 h
    };
    // p-adic split of the exponents (the lift of encrypt_core, one prime at a time): with r = r0 + r1 p,
    //     x^r = (x^(r1))^p x^(r0)   and   W^p mod p^3 depends on W mod p^2 only,
    // so W = x^(r1) [y^(s1)] is an interleaved ladder of 2 047 squarings modulo p^2 -- on the one-lane pair kernel, 3.5 H^2
    // multiplies a squaring -- and W^p x^(r0) [y^(s0)] an interleaved ladder of 1 024 squarings modulo p^3, where the ladder
    // on the whole reduced exponent squares 3 071 times modulo p^3 (8 H^2 issue slots each).
    const bool reduced_pn = !exps || (ex[0] != exps && ex[1] != exps && wex[0] == sk->eo_p.w && wex[1] == sk->eo_q.w);
    const size_t lanes_target_s = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
    // (below one wave per SIMD for the stage modulo p^2 the ladders are bound by their length, and one ladder is shorter than two)
    // (7-bit windows of r0 while their 128-entry tables fit the 32-bit gather offsets -- 75 000 numbers for 37-limb primes --
    // and 5-bit windows on number-major tables beyond: a big batch keeps the split, it does not fall back to the long ladder)
    if (ctx->use_lift && reduce_e && (exps || base2) && reduced_pn && (win == 7 || nm5) && sk->mp2.WT == 2 * sk->mp.WT && nb * 4 >= lanes_target_s &&
        sk->eo_p.w <= 3 * sk->mp.WT && sk->pinv2k_2.d &&
        (uint64_t)nb * 3 * sk->mp.WT * 4 * (uint64_t)((1u << win) + 1) < (1ull << 32)) {   // (what the gathers address: the number-major entries)
      const int H = sk->mp.WT, W2 = sk->mp2.WT;
      const size_t S1 = (size_t)H * nb, S2 = (size_t)W2 * nb;
      const uint32_t *r0[2] = {nullptr, nullptr}, *r1[2] = {nullptr, nullptr}, *x2[2], *y2[2] = {nullptr, nullptr};
      BigU s0[2], s1[2];
      uint32_t *xr[2], *yr3[2] = {nullptr, nullptr};
      Fork fa(ctx, 4);
      for (int half = 0; half < 2; ++half) {
        fa.chain(2 * half);
        uint32_t* tbx = ctx->ws_t<uint32_t>(S);
        const ModCtx &m1 = half ? sk->mq : sk->mp, &m2 = half ? sk->mq2 : sk->mp2, &m3 = half ? mq3 : mp3;
        if (exps) {
          uint32_t* t2 = ctx->ws_t<uint32_t>(S2);
          uint32_t* d0 = ctx->ws_t<uint32_t>(S1);
          uint32_t* d1 = ctx->ws_t<uint32_t>(S2);
          reduce_mod(ctx, m2, ex[half], wex[half], t2, nb);
          reduce_mod(ctx, m1, t2, W2, d0, nb);                                                       // r0 = r mod prime
          launch_div_exact(ex[half], wex[half], 0, d0, H, tbx, (half ? sk->qinv2k_2 : sk->pinv2k_2).d, m1.d_nmod, H, d1, W2, nb, nb,
                           nullptr, 0, ctx->stream);                                                 // r1 = (r - r0) / prime
          r0[half] = d0;
          r1[half] = d1;
        }
        if (base2) hostbig::divmod(es[half], half ? sk->q : sk->p, s1[half], s0[half]);
        // the bases modulo prime^3 (kept for stage B) and modulo prime^2 (stage A)
        xr[half] = ctx->ws_t<uint32_t>(S);
        reduce_mod(ctx, m3, base, wb, xr[half], nb);
        uint32_t* xx = ctx->ws_t<uint32_t>(S2);
        reduce_mod(ctx, m2, xr[half], W, xx, nb);
        x2[half] = xx;
        if (base2) {
          fa.chain(2 * half + 1);
          uint32_t* yr = ctx->ws_t<uint32_t>(S);
          reduce_mod(ctx, m3, base2, wb2, yr, nb);
          yr3[half] = yr;
          uint32_t* yy = ctx->ws_t<uint32_t>(S2);
          reduce_mod(ctx, m2, yr, W, yy, nb);
          y2[half] = yy;
        }
      }
      fa.join();
      uint32_t* wv[2];
      // (W stays in pair form: the digit kernel of the same prime takes (a0, a1, 0) as W's digit form)
      const bool hand = ctx->use_handover && sk->mp.WT == mp3.triple.root->WT && sk->mq.WT == mq3.triple.root->WT;
      if (pow_p2_multi_crt(sk, x2, exps ? r1 : nullptr, W2, base2 ? y2 : nullptr, s1, nb, wv, nullptr, nullptr, hand)) {
        // stage B: slots 0 x, 1 W, 2 tmp, 3 out, 4 y, 5.. the per-number table (128 + 64), then W's and y's odd powers
        const uint32_t TABW = 5 + (uint32_t)perlane_table_slots(win, nm5), TABY = TABW + 64;
        TriplePlan up = triple_alloc(ctx, mp3, nb, (int)TABY + 64), uq = triple_alloc(ctx, mq3, nb, (int)TABY + 64);
        Prog pb[2];
        Fork fb(ctx, 4);
        for (int half = 0; half < 2; ++half) {
          const ModCtx& m3 = half ? mq3 : mp3;
          const TriplePlan& t = half ? uq : up;
          fb.chain(2 * half);
          triple_enter(ctx, m3, xr[half], t, 0);
          fb.chain(2 * half + 1);
          if (hand) {
            triple_from_pair(ctx, wv[half], t, 1);
          } else {
            uint32_t* wz = ctx->ws_t<uint32_t>(S);
            launch_copy_limbs(wv[half], 0, W2, wz, W, nb, ctx->stream);
            triple_enter(ctx, m3, wz, t, 1);
          }
          std::vector<SharedBase> sh;
          sh.push_back(SharedBase{half ? sk->q : sk->p, 1, TABW});
          if (base2) {
            fb.chain(2 * half);
            triple_enter(ctx, m3, yr3[half], t, 4);
            sh.push_back(SharedBase{s0[half], 4, TABY});
          }
          std::vector<PerNumberBase> pn;
          if (exps) pn.push_back(PerNumberBase{H, 0, 5, 0});
          emit_modexp_multi(pb[half], pn, win, sh, 2, 3, 0, nm5);
          pb[half].end();
        }
        fb.join();
        SegSpec sp{&mp3, &pb[0], up.mem, r0[0] ? triple_windows(ctx, r0[0], H, nb, win) : nullptr},
                sq{&mq3, &pb[1], uq.mem, r0[1] ? triple_windows(ctx, r0[1], H, nb, win) : nullptr};
        sp.pair = mp3.triple.kconsts; sp.pair_n0inv = sk->mp.n0inv; sp.pair_h = up.H; sp.pair_lanes = 3; sp.tconsts = mp3.triple.tconsts;
        sq.pair = mq3.triple.kconsts; sq.pair_n0inv = sk->mq.n0inv; sq.pair_h = uq.H; sq.pair_lanes = 3; sq.tconsts = mq3.triple.tconsts;
        run_vm(ctx, nb, sp, &sq, true);
        garner(up, uq);
        return;
      }
    }
    TriplePlan tp = triple_alloc(ctx, mp3, nb, nslots), tq = triple_alloc(ctx, mq3, nb, nslots);
    Fork fc(ctx);
    for (int half = 0; half < 2; ++half) {
      fc.chain(half);
      const ModCtx& m3 = half ? mq3 : mp3;
      const TriplePlan& t = half ? tq : tp;
      uint32_t* red = half ? ctx->ws_t<uint32_t>(S) : g + 5 * S;       // (a scratch slot per half: the chains run side by side)
      reduce_mod(ctx, m3, base, wb, red, nb);
      triple_enter(ctx, m3, red, t, 0);
      if (base2) {
        reduce_mod(ctx, m3, base2, wb2, red, nb);
        triple_enter(ctx, m3, red, t, 1);
      }
    }
    fc.join();
    Prog pp, pq;
    for (int half = 0; half < 2; ++half) {
      Prog& pr = half ? pq : pp;
      if (base2) emit_modexp_dual(pr, wex[half], es[half], 0, 1, 2, 3, 5, tab2, 0, win, nm5);
      else if (exps) emit_modexp_perlane(pr, wex[half], 0, NO_SLOT, 2, 3, 5, NO_SLOT, 0, win, nm5);
      else emit_modexp_shared(pr, es[half], 0, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
      pr.end();
    }
    SegSpec sp{&mp3, &pp, tp.mem, ex[0] ? triple_windows(ctx, ex[0], wex[0], nb, win) : nullptr},
            sq{&mq3, &pq, tq.mem, ex[1] ? triple_windows(ctx, ex[1], wex[1], nb, win) : nullptr};
    sp.pair = mp3.triple.kconsts; sp.pair_n0inv = sk->mp.n0inv; sp.pair_h = tp.H; sp.pair_lanes = 3; sp.tconsts = mp3.triple.tconsts;
    sq.pair = mq3.triple.kconsts; sq.pair_n0inv = sk->mq.n0inv; sq.pair_h = tq.H; sq.pair_lanes = 3; sq.tconsts = mq3.triple.tconsts;
    run_vm(ctx, nb, sp, &sq, true);
    garner(tp, tq);
    return;
  }
  // slots: P: in 0, in2 1, tmp 2, out 3, tables 4..51;  Q: the same + QO;  A, B, h after them
  const uint32_t QO = 56, SA = 112, SB = 113, SH = 114;
  uint32_t* mem = ctx->ws_t<uint32_t>(S * 115);
  reduce_mod(ctx, mp3, base, wb, mem + 0 * S, nb);
  reduce_mod(ctx, mq3, base, wb, mem + (size_t)QO * S, nb);
  if (base2) {
    reduce_mod(ctx, mp3, base2, wb2, mem + 1 * S, nb);
    reduce_mod(ctx, mq3, base2, wb2, mem + (size_t)(QO + 1) * S, nb);
  }
  {
    Prog pp, pq;
    if (base2) {
      emit_modexp_dual(pp, we, *e, 0, 1, 2, 3, 4, 20);
      emit_modexp_dual(pq, we, *e, QO, QO + 1, QO + 2, QO + 3, QO + 4, QO + 20);
    } else if (exps) {
      emit_modexp_perlane(pp, we, 0, NO_SLOT, 2, 3, 4, NO_SLOT);
      emit_modexp_perlane(pq, we, QO, NO_SLOT, QO + 2, QO + 3, QO + 4, NO_SLOT);
    } else {
      emit_modexp_shared(pp, *e, 0, NO_SLOT, 2, 3, 4, NO_SLOT, true);
      emit_modexp_shared(pq, *e, QO, NO_SLOT, QO + 2, QO + 3, QO + 4, NO_SLOT, true);
    }
    pp.end();
    pq.end();
    SegSpec sp{&mp3, &pp, mem, exps}, sq{&mq3, &pq, mem, exps};
    run_vm(ctx, nb, sp, &sq, true);
  }
  uint32_t *xp = mem + 3 * S, *xq = mem + (size_t)(QO + 3) * S;
  launch_canon(xp, mp3.d_nmod, W, nb, ctx->stream);     // ONE integer x_p for both uses below
  launch_canon(xq, mq3.d_nmod, W, nb, ctx->stream);
  {
    Prog c;
    c.op(VM_LOAD, 3);      c.op(VM_MULC, (uint32_t)sk->c_p3invR); c.op(VM_STORE, SB);
    c.op(VM_LOAD, QO + 3); c.op(VM_MULC, (uint32_t)sk->c_p3invR); c.op(VM_STORE, SA);
    c.end();
    SegSpec sc{&mq3, &c, mem, nullptr};
    run_vm(ctx, nb, sc, nullptr, false);
  }
  launch_canon(mem + (size_t)SA * S, mq3.d_nmod, W, nb, ctx->stream);
  launch_canon(mem + (size_t)SB * S, mq3.d_nmod, W, nb, ctx->stream);
  launch_sub_mod(mem + (size_t)SA * S, mem + (size_t)SB * S, mq3.d_nmod, mem + (size_t)SH * S, W, nb, ctx->stream);  // h = (x_q - x_p) / p^3 mod q^3
  launch_mul_const_add(mem + (size_t)SH * S, W, sk->p3_limbs.d, W, xp, W, 0, out, W3, nb, ctx->stream);              // x_p + p^3 h
}

// out = A^(ea) * B^(eb) mod n^3 for the holder of the factorisation, BOTH exponents per number and already reduced modulo the
// group orders of p^3 and q^3 (ea[half], eb[half]: eo.w limbs): one interleaved ladder per half on the three-digit kernel --
// two per-number window tables hang off one chain of squarings -- then Garner.  A, B: W3-limb residues.  False when the
// three-digit kernels do not serve this key / batch.
// The two bases of the prover's response are per-STATEMENT values (s, b): their residues modulo prime^3 / prime^2 and their digit
// forms can be made for every statement while the Alpha ladders run (side stream), before the challenge bits say which
// instances need them; the response then GATHERS them by statement index instead of entering them between two dependent ladders.
struct PreBases {
  size_t nbs = 0;                       // statements (stride of the arrays below)
  const uint32_t* r3[2][2] = {};        // [half][base]: canonical residues modulo prime^3 (mp3.WT limbs)
  const uint32_t* r2[2][2] = {};        // ... modulo prime^2 (mp2.WT limbs)
  TriplePlan dig[2];                    // [half]: slot 0 = A, slot 1 = B in digit form
  const uint32_t* sti = nullptr;        // device: statement of every number of the call
  size_t cnt = 0;                       // numbers of the call that are real (the rest is padding)
};
bool pre_bases_usable(const pgpu_seckey* sk) {
  return triple_usable(sk->ctx, sk->mp3) && triple_usable(sk->ctx, sk->mq3) && sk->eo_p.ok && sk->eo_q.ok && sk->eo_p.w == sk->eo_q.w;
}
void pre_bases(const pgpu_seckey* sk, const uint32_t* A, const uint32_t* B, size_t nbs, PreBases& pre) {
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx &mp3 = sk->mp3, &mq3 = sk->mq3;
  const int W = mp3.WT, W2 = sk->mp2.WT, W3 = sk->pk->mn3->WT;
  pre.nbs = nbs;
  Fork f(ctx, 4);
  for (int half = 0; half < 2; ++half) {
    const ModCtx &m2 = half ? sk->mq2 : sk->mp2, &m3 = half ? mq3 : mp3;
    pre.dig[half] = triple_alloc(ctx, m3, nbs, 2);
    for (int k = 0; k < 2; ++k) {
      f.chain(2 * half + k);
      uint32_t* r3 = ctx->ws_t<uint32_t>((size_t)W * nbs);
      reduce_mod(ctx, m3, k ? B : A, W3, r3, nbs);
      uint32_t* r2 = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
      reduce_mod(ctx, m2, r3, W, r2, nbs);
      triple_enter(ctx, m3, r3, pre.dig[half], (uint32_t)k);
      pre.r3[half][k] = r3;
      pre.r2[half][k] = r2;
    }
  }
  f.join();
}

bool pow_n3_crt_two(const pgpu_seckey* sk, const uint32_t* A, const uint32_t* const ea[2], const uint32_t* B,
                    const uint32_t* const eb[2], size_t nb, uint32_t* out, const PreBases* pre = nullptr) {
  // pre: A and B are not read; number g of the call is statement pre->sti[g] of the prepared bases
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx &mp3 = sk->mp3, &mq3 = sk->mq3;
  const int W = mp3.WT, W3 = sk->pk->mn3->WT;
  const size_t S = (size_t)W * nb;
  if (!(triple_usable(ctx, mp3) && triple_usable(ctx, mq3) && sk->eo_p.ok && sk->eo_q.ok && sk->eo_p.w == sk->eo_q.w)) return false;
  const int H = mp3.triple.root->WT, we = sk->eo_p.w;
  if (triple_window_bits(nb, H) != 7) return false;
  const int win = 7;
  // The p-adic split of BOTH exponents (what pow_n3_crt does for one): with e = e0 + e1 prime,
  //     A^(ea) B^(eb) = (A^(ea1) B^(eb1))^prime * A^(ea0) B^(eb0)    and    W^prime mod prime^3 depends on W mod prime^2 only,
  // so W is an interleaved ladder of 2 047 squarings modulo prime^2 on the one-lane pair kernel (two per-number exponents, 4-bit
  // windows) and the rest an interleaved ladder of 1 024 squarings modulo prime^3 with two per-number exponents and the shared
  // exponent prime on W -- where the unsplit ladder squares 3 071 times modulo prime^3.  Taken when the stage modulo prime^2
  // fills the chip (the response batch of ProveDDLEQ at secpar 40: half of 61 440 instances).
  {
    const size_t lanes_target_s = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
    const int H1 = sk->mp.WT, W2 = sk->mp2.WT;
    if (ctx->use_lift && sk->mp2.WT == 2 * sk->mp.WT && sk->mq2.WT == 2 * sk->mq.WT && nb * 4 >= lanes_target_s && we <= 3 * H1 &&
        sk->pinv2k_2.d && sk->qinv2k_2.d && (uint64_t)nb * 3 * H1 * 4 * 129 < (1ull << 32) && H1 == H) {
      const size_t S1 = (size_t)H1 * nb, S2 = (size_t)W2 * nb;
      const uint32_t *a0[2], *a1[2], *b0[2], *b1[2], *A2[2], *B2[2];
      uint32_t *Ar[2], *Br[2];
      Fork fa(ctx, 4);                              // (the chains of small kernels of both halves and both bases side by side, here and below)
      for (int half = 0; half < 2; ++half) {
        const ModCtx &m1 = half ? sk->mq : sk->mp, &m2 = half ? sk->mq2 : sk->mp2, &m3 = half ? mq3 : mp3;
        for (int k = 0; k < 2; ++k) {
          fa.chain(2 * half + k);
          uint32_t* tbx = ctx->ws_t<uint32_t>(S);
          const uint32_t* ex = k ? eb[half] : ea[half];
          uint32_t* t2 = ctx->ws_t<uint32_t>(S2);
          uint32_t* d0 = ctx->ws_t<uint32_t>(S1);
          uint32_t* d1 = ctx->ws_t<uint32_t>(S2);
          reduce_mod(ctx, m2, ex, we, t2, nb);
          reduce_mod(ctx, m1, t2, W2, d0, nb);                                                       // e0 = e mod prime
          launch_div_exact(ex, we, 0, d0, H1, tbx, (half ? sk->qinv2k_2 : sk->pinv2k_2).d, m1.d_nmod, H1, d1, W2, nb, nb, nullptr, 0,
                           ctx->stream);                                                             // e1 = (e - e0) / prime
          (k ? b0 : a0)[half] = d0;
          (k ? b1 : a1)[half] = d1;
          uint32_t* r3 = ctx->ws_t<uint32_t>(S);
          uint32_t* r2 = ctx->ws_t<uint32_t>(S2);
          if (pre) {
            launch_gather(pre->r3[half][k], pre->nbs, pre->sti, pre->cnt, r3, nb, W, ctx->stream);
            launch_gather(pre->r2[half][k], pre->nbs, pre->sti, pre->cnt, r2, nb, W2, ctx->stream);
          } else {
            reduce_mod(ctx, m3, k ? B : A, W3, r3, nb);
            reduce_mod(ctx, m2, r3, W, r2, nb);
          }
          (k ? Br : Ar)[half] = r3;
          (k ? B2 : A2)[half] = r2;
        }
      }
      fa.join();
      uint32_t* wv[2];
      const bool hand = ctx->use_handover && sk->mp.WT == mp3.triple.root->WT && sk->mq.WT == mq3.triple.root->WT;
      if (pow_p2_multi_crt(sk, A2, a1, W2, nullptr, nullptr, nb, wv, B2, b1, hand)) {
        // stage B: slots 0 A, 1 B, 2 tmp, 3 out, 4 W, 5.. A's table, then B's, then W's odd powers
        const uint32_t TA = 5, TB = TA + (uint32_t)perlane_table_slots(win), TW = TB + (uint32_t)perlane_table_slots(win);
        TriplePlan up = triple_alloc(ctx, mp3, nb, (int)TW + 64), uq = triple_alloc(ctx, mq3, nb, (int)TW + 64);
        uint32_t* g2 = ctx->ws_t<uint32_t>(S * 6);
        Prog pb[2];
        const uint32_t* dg2[2];
        Fork fb(ctx, 4);
        for (int half = 0; half < 2; ++half) {
          const ModCtx& m3 = half ? mq3 : mp3;
          const TriplePlan& t = half ? uq : up;
          fb.chain(2 * half);
          if (pre) launch_gather(pre->dig[half].slot(0), pre->nbs, pre->sti, pre->cnt, t.slot(0), nb, 3 * t.H, ctx->stream);
          else triple_enter(ctx, m3, Ar[half], t, 0);
          fb.chain(2 * half + 1);
          if (pre) launch_gather(pre->dig[half].slot(1), pre->nbs, pre->sti, pre->cnt, t.slot(1), nb, 3 * t.H, ctx->stream);
          else triple_enter(ctx, m3, Br[half], t, 1);
          fb.chain(2 * half);
          if (hand) {
            triple_from_pair(ctx, wv[half], t, 4);
          } else {
            uint32_t* wz = ctx->ws_t<uint32_t>(S);
            launch_copy_limbs(wv[half], 0, W2, wz, W, nb, ctx->stream);
            triple_enter(ctx, m3, wz, t, 4);
          }
          uint32_t* d2 = ctx->ws_t<uint32_t>((size_t)2 * H1 * nb);
          HIPCHK(hipMemcpyAsync(d2, a0[half], S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
          HIPCHK(hipMemcpyAsync(d2 + S1, b0[half], S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
          dg2[half] = d2;
          std::vector<SharedBase> sh;
          sh.push_back(SharedBase{half ? sk->q : sk->p, 4, TW});
          std::vector<PerNumberBase> pn;
          pn.push_back(PerNumberBase{H1, 0, TA, 0});
          pn.push_back(PerNumberBase{H1, 1, TB, (uint32_t)perlane_windows(H1, win)});
          emit_modexp_multi(pb[half], pn, win, sh, 2, 3, 0);
          pb[half].end();
        }
        fb.join();
        SegSpec sp{&mp3, &pb[0], up.mem, dg2[0]}, sq{&mq3, &pb[1], uq.mem, dg2[1]};
        sp.pair = mp3.triple.kconsts; sp.pair_n0inv = sk->mp.n0inv; sp.pair_h = up.H; sp.pair_lanes = 3; sp.tconsts = mp3.triple.tconsts;
        sq.pair = mq3.triple.kconsts; sq.pair_n0inv = sk->mq.n0inv; sq.pair_h = uq.H; sq.pair_lanes = 3; sq.tconsts = mq3.triple.tconsts;
        run_vm(ctx, nb, sp, &sq, true);
        Fork fx(ctx);
        fx.chain(0);
        triple_exit(ctx, mp3, up, 3, g2 + 0 * S, nullptr);
        fx.chain(1);
        triple_exit(ctx, mq3, uq, 3, g2 + 1 * S, nullptr);
        fx.join();
        Prog c;
        c.op(VM_LOAD, 0); c.op(VM_MULC, (uint32_t)sk->c_p3invR); c.op(VM_STORE, 3);
        c.op(VM_LOAD, 1); c.op(VM_MULC, (uint32_t)sk->c_p3invR); c.op(VM_STORE, 2);
        c.end();
        SegSpec sc{&mq3, &c, g2, nullptr};
        run_vm(ctx, nb, sc, nullptr, false);
        launch_canon(g2 + 2 * S, mq3.d_nmod, W, nb, ctx->stream);
        launch_canon(g2 + 3 * S, mq3.d_nmod, W, nb, ctx->stream);
        launch_sub_mod(g2 + 2 * S, g2 + 3 * S, mq3.d_nmod, g2 + 4 * S, W, nb, ctx->stream);               // h = (x_q - x_p) / p^3 mod q^3
        launch_mul_const_add(g2 + 4 * S, W, sk->p3_limbs.d, W, g2, W, 0, out, W3, nb, ctx->stream);       // x_p + p^3 h
        return true;
      }
    }
  }
  const uint32_t TABA = 5, TABB = TABA + (uint32_t)perlane_table_slots(win);
  TriplePlan tp = triple_alloc(ctx, mp3, nb, (int)TABB + perlane_table_slots(win)), tq = triple_alloc(ctx, mq3, nb, (int)TABB + perlane_table_slots(win));
  uint32_t* g = ctx->ws_t<uint32_t>(S * 6);       // generic slots (W limbs): 0 x_p, 1 x_q, 2 A, 3 B, 4 h, 5 scratch
  Prog pr[2];
  const uint32_t* dg[2];
  Fork fc(ctx);
  for (int half = 0; half < 2; ++half) {
    fc.chain(half);
    const ModCtx& m3 = half ? mq3 : mp3;
    const TriplePlan& t = half ? tq : tp;
    if (pre) {
      launch_gather(pre->dig[half].slot(0), pre->nbs, pre->sti, pre->cnt, t.slot(0), nb, 3 * t.H, ctx->stream);
      launch_gather(pre->dig[half].slot(1), pre->nbs, pre->sti, pre->cnt, t.slot(1), nb, 3 * t.H, ctx->stream);
    } else {
      uint32_t* red = half ? ctx->ws_t<uint32_t>(S) : g + 5 * S;     // (a scratch slot per half: the chains run side by side)
      reduce_mod(ctx, m3, A, W3, red, nb);
      triple_enter(ctx, m3, red, t, 0);
      reduce_mod(ctx, m3, B, W3, red, nb);
      triple_enter(ctx, m3, red, t, 1);
    }
    // the two exponents of a number one after the other in the rows of `digits`: windows 0 .. 4 we - 1 and 4 we .. 8 we - 1
    uint32_t* d2 = ctx->ws_t<uint32_t>((size_t)2 * we * nb);
    HIPCHK(hipMemcpyAsync(d2, ea[half], (size_t)we * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d2 + (size_t)we * nb, eb[half], (size_t)we * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
    dg[half] = d2;
    std::vector<PerNumberBase> pn;
    pn.push_back(PerNumberBase{we, 0, TABA, 0});
    pn.push_back(PerNumberBase{we, 1, TABB, (uint32_t)perlane_windows(we, win)});
    emit_modexp_multi(pr[half], pn, win, {}, 2, 3, 0);
    pr[half].end();
  }
  fc.join();
  SegSpec sp{&mp3, &pr[0], tp.mem, dg[0]}, sq{&mq3, &pr[1], tq.mem, dg[1]};
  sp.pair = mp3.triple.kconsts; sp.pair_n0inv = sk->mp.n0inv; sp.pair_h = tp.H; sp.pair_lanes = 3; sp.tconsts = mp3.triple.tconsts;
  sq.pair = mq3.triple.kconsts; sq.pair_n0inv = sk->mq.n0inv; sq.pair_h = tq.H; sq.pair_lanes = 3; sq.tconsts = mq3.triple.tconsts;
  run_vm(ctx, nb, sp, &sq, true);
  Fork fx(ctx);
  fx.chain(0);
  triple_exit(ctx, mp3, tp, 3, g + 0 * S, nullptr);
  fx.chain(1);
  triple_exit(ctx, mq3, tq, 3, g + 1 * S, nullptr);
  fx.join();
  Prog c;
  c.op(VM_LOAD, 0); c.op(VM_MULC, (uint32_t)sk->c_p3invR); c.op(VM_STORE, 3);
  c.op(VM_LOAD, 1); c.op(VM_MULC, (uint32_t)sk->c_p3invR); c.op(VM_STORE, 2);
  c.end();
  SegSpec sc{&mq3, &c, g, nullptr};
  run_vm(ctx, nb, sc, nullptr, false);
  launch_canon(g + 2 * S, mq3.d_nmod, W, nb, ctx->stream);
  launch_canon(g + 3 * S, mq3.d_nmod, W, nb, ctx->stream);
  launch_sub_mod(g + 2 * S, g + 3 * S, mq3.d_nmod, g + 4 * S, W, nb, ctx->stream);                  // h = (x_q - x_p) / p^3 mod q^3
  launch_mul_const_add(g + 4 * S, W, sk->p3_limbs.d, W, g, W, 0, out, W3, nb, ctx->stream);          // x_p + p^3 h
  return true;
}

// [w][nb] arrays a, b  ->  one [w][2 nb] array (a's numbers first): two independent batches share one launch
uint32_t* concat2(pgpu_ctx* ctx, const uint32_t* a, const uint32_t* b, int w, size_t nb) {
  uint32_t* o = ctx->ws_t<uint32_t>((size_t)w * 2 * nb);
  HIPCHK(hipMemcpy2DAsync(o, 2 * nb * 4, a, nb * 4, nb * 4, (size_t)w, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpy2DAsync(o + nb, 2 * nb * 4, b, nb * 4, nb * 4, (size_t)w, hipMemcpyDeviceToDevice, ctx->stream));
  return o;
}
// half `which` (0 / 1) of a [w][2 nb] array -> [w][nb]
void split2(pgpu_ctx* ctx, const uint32_t* in, int which, int w, size_t nb, uint32_t* out) {
  HIPCHK(hipMemcpy2DAsync(out, nb * 4, in + (size_t)which * nb, 2 * nb * 4, nb * 4, (size_t)w, hipMemcpyDeviceToDevice, ctx->stream));
}

// [w][nba + nbb] <- a ([w][nba]) | b ([w][nbb]): two batches of different sizes side by side in one launch
uint32_t* concat_ab(pgpu_ctx* ctx, const uint32_t* a, size_t nba, const uint32_t* b, size_t nbb, int w) {
  const size_t t = nba + nbb;
  uint32_t* o = ctx->ws_t<uint32_t>((size_t)w * t);
  HIPCHK(hipMemcpy2DAsync(o, t * 4, a, nba * 4, nba * 4, (size_t)w, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpy2DAsync(o + nba, t * 4, b, nbb * 4, nbb * 4, (size_t)w, hipMemcpyDeviceToDevice, ctx->stream));
  return o;
}
// part `which` (0: the first nba numbers, 1: the nbb after them) of a [w][nba + nbb] array
void split_ab(pgpu_ctx* ctx, const uint32_t* in, size_t nba, size_t nbb, int which, int w, uint32_t* out) {
  const size_t t = nba + nbb, n = which ? nbb : nba;
  HIPCHK(hipMemcpy2DAsync(out, n * 4, in + (which ? nba : 0), t * 4, n * 4, (size_t)w, hipMemcpyDeviceToDevice, ctx->stream));
}

uint32_t* zext(pgpu_ctx* ctx, const uint32_t* in, int w, int wo, size_t nb) {
  uint32_t* o = ctx->ws_t<uint32_t>((size_t)wo * nb);
  launch_copy_limbs(in, 0, w, o, wo, nb, ctx->stream);
  return o;
}

}  // namespace

extern "C" {

// EncryptWithR for the holder of the secret key (SecretKey embeds PublicKey in the reference: sk.EncryptWithR is the same
// method): r^n mod n^2 through p^2 and q^2 -- half the width, the exponent modulo p (p - 1) -- on the one-lane pair kernel of
// Decrypt, then Garner and the closed form of G^m.  The same integers as pgpu_encrypt_with_r, a third of its multiplies.
int pgpu_encrypt_with_r_sk(const pgpu_seckey* sk, int level, size_t batch, const uint8_t* m, size_t m_stride, const uint8_t* r,
                           size_t r_stride, uint8_t* c, size_t c_stride, int mem) {
  if (!sk) return fail(PGPU_ERR_INVALID, "null key");
  pgpu_ctx* ctx = sk->ctx;
  const pgpu_pubkey* pk = sk->pk;
  if (level != PGPU_LEVEL_ONE || !pk->g_is_n_plus_1 || !pow_n2_crt_usable(sk))
    return pgpu_encrypt_with_r(pk, level, batch, m, m_stride, r, r_stride, c, c_stride, mem);   // nothing to gain: the public path
  return guarded([&] {
    check_batch_args(m, c, batch);
    if (!r) api_throw(PGPU_ERR_INVALID, "null buffer");
    ctx->bind();
    ctx->reset_ws();
    const ModCtx &mn = pk->mn, &mn2 = pk->mn2;
    const size_t nb = round_up(batch, VM_BLOCK);
    uint32_t* gm = ctx->ws_t<uint32_t>((size_t)mn2.WT * nb);
    build_gm(ctx, pk, level, m, m_stride, batch, mem, nb, gm);                       // 1 + (m mod n) n
    uint32_t* rl = ctx->ws_t<uint32_t>((size_t)mn.WT * nb);
    unpack_mod(ctx, mn, r, r_stride, batch, mem, rl, nb, true);                      // r^n mod n^2 depends on r mod n only
    uint32_t* rn = pow_n2_crt(sk, rl, pk->N, nb);                                    // canonical, mn2.WT limbs
    // c = r^n * g^m mod n^2
    const size_t sw = (size_t)mn2.WT * nb;
    uint32_t* mem2 = ctx->ws_t<uint32_t>(sw * 3);
    HIPCHK(hipMemcpyAsync(mem2, rn, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(mem2 + sw, gm, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
    Prog p;
    p.op(VM_LOAD, 0); p.op(VM_MULC, C_R2); p.op(VM_MUL, 1); p.op(VM_STORE, 2); p.end();
    SegSpec sg{&mn2, &p, mem2, nullptr};
    run_vm(ctx, nb, sg, nullptr, false);
    launch_canon(mem2 + 2 * sw, mn2.d_nmod, mn2.WT, nb, ctx->stream);
    pack_result(ctx, mem2 + 2 * sw, mn2.WT, nb, batch, c, c_stride, mn2.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

// ProveDDLEQ (ddleq.go:27-40) for `n_statements` statements (ct1, ct2, a, b) with `secpar` instances each -- draws x, y
// supplied, instance k of statement j in row j * secpar + k of x / y / alpha / e / f.  What ddleq.go:55-127 recomputes in every
// instance although it depends on the statement only is computed ONCE per statement: the sanity check ct1^(a^n) b^(n^2) == ct2
// (:62-69), a^n (:104), a^-1 (:95) and (a^n)^-1, s = ExtractRandonness(ct1) (:103) and the unit tests of s and b; per instance
// remain x^n, alpha = ct1^(x^n) y^(n^2), the challenge bit and -- for bit 1 -- the response ladder.  The integers are those of
// `secpar` calls of proveDDLEQInstance with the same draws.  secpar = 1 is pgpu_ddleq_prove.
static void ddleq_prove_impl(const pgpu_seckey* sk, size_t S, size_t secpar, const uint8_t* ct1, const uint8_t* ct2, size_t ct_stride,
                             const uint8_t* a, const uint8_t* b, const uint8_t* x, const uint8_t* y, size_t n_stride, uint8_t* alpha,
                             uint8_t* e_out, size_t e_stride, uint8_t* f_out, int mem) {
  pgpu_ctx* ctx = sk->ctx;
  const pgpu_pubkey* pk = sk->pk;
  {
    if (S == 0 || secpar == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
    if (!pk->mn3 || sk->c_mu2R < 0) api_throw(PGPU_ERR_UNSUPPORTED, "level two is not available for this key");
    if (!pk->g_is_n_plus_1) api_throw(PGPU_ERR_UNSUPPORTED, "DDLEQ prover assumes G = N+1");
    ctx->bind();
    ctx->reset_ws();
    const ModCtx &mn = pk->mn, &mn2 = pk->mn2, &mn3 = *pk->mn3;
    const int W1 = mn.WT, W2 = mn2.WT, W3 = mn3.WT;
    // the prover holds the factorisation: exponentiations modulo n^3 go through p^3 and q^3 (pow_n3_crt)
    const bool crt3 = sk->has_crt2 && sk->c_p3invR >= 0 && 2 * sk->mp3.WT >= W3 && ctx->use_pair;
    auto perlane3 = [&](const uint32_t* base, const uint32_t* exps, int we, size_t nbx, uint32_t* outp) {
      if (crt3) pow_n3_crt(sk, base, W3, exps, we, nullptr, nbx, outp);
      else perlane_pow(ctx, mn3, base, exps, we, nbx, outp);
    };
    auto shared3 = [&](const uint32_t* base, int wb, const BigU& ex, size_t nbx, uint32_t* outp) {
      if (crt3) pow_n3_crt(sk, base, wb, nullptr, 0, &ex, nbx, outp);
      else shared_pow(ctx, mn3, base, wb, ex, nbx, outp);
    };
    const size_t batch = S * secpar;                        // instances
    const size_t nbs = round_up(S, VM_BLOCK), nb = round_up(batch, VM_BLOCK);
    if (ct_stride != mn3.nbytes) api_throw(PGPU_ERR_INVALID, "ciphertext stride must be the byte length of n^3");
    if (n_stride * 8 > (size_t)LB * W1 + 7) api_throw(PGPU_ERR_INVALID, "a, b, x, y must fit the width of n");
    auto up = [&](const uint8_t* buf, size_t stride, int w, size_t count, size_t nbx) {
      uint32_t* l = ctx->ws_t<uint32_t>((size_t)w * nbx);
      unpack_operand(ctx, buf, stride, stride, count, mem, l, w, nbx);
      return l;
    };
    // per statement (S numbers, row stride nbs) ...
    uint32_t *c1s = up(ct1, ct_stride, W3, S, nbs), *c2s = up(ct2, ct_stride, W3, S, nbs);
    uint32_t *al = up(a, n_stride, W1, S, nbs), *bl = up(b, n_stride, W1, S, nbs);
    // ... and per instance (S * secpar numbers, row stride nb)
    uint32_t *xl = up(x, n_stride, W1, batch, nb), *yl = up(y, n_stride, W1, batch, nb);
    uint32_t* d_stmt = nullptr;                              // instance -> its statement
    if (secpar > 1) {
      std::vector<uint32_t> st(batch);
      for (size_t i = 0; i < batch; ++i) st[i] = (uint32_t)(i / secpar);
      d_stmt = ctx->upload_words(st);
    }
    auto expand = [&](uint32_t* in, int w) {                 // a per-statement array repeated for the instances of its statement
      if (secpar == 1) return in;
      uint32_t* o = ctx->ws_t<uint32_t>((size_t)w * nb);
      launch_gather(in, nbs, d_stmt, batch, o, nb, w, ctx->stream);
      return o;
    };
    uint32_t *c1 = expand(c1s, W3), *c2 = expand(c2s, W3);
    const BigU &N = pk->N, &N2 = mn2.N;
    // ---- what depends on the STATEMENT only and on no ladder of this call goes to the side stream, beside the big launches
    // (the GPU was busy, but with ~800 small launches in a row between the ladders: 30 of 188 ms per 16 384 instances):
    //   s = ExtractRandonness(ct1) (ddleq.go:103) for every statement -- a latency-bound launch beside the a^n | x^n
    //   launch, which fills half the chip; then, behind a^n, the inversion tree for a^-1 | (a^n)^-1 and the unit test of s b
    //   beside the Alpha ladders.  Which statements have an instance with challenge bit 1 is not known yet: all are done
    //   (the side work is bound by launch latencies, not by its width).
    SideStream side(ctx);
    BigU ns_inv;
    if (!hostbig::modinv(N2 % sk->lambda, sk->lambda, ns_inv)) api_throw(PGPU_ERR_NOT_INVERTIBLE, "n^2 is not invertible mod lambda");
    uint32_t* qs = nullptr;                                  // s per statement (W1 limbs, stride nbs)
    hipEvent_t inputs_ready = side.mark();
    // ---- sanity check (ddleq.go:62-69): ct1^(a^n mod n^2) * b^(n^2) mod n^3 == ct2, else the reference panics -- once per
    // statement.  Independent exponentiations of the same shape share a launch (the chip is filled better and, for the half-size
    // batches of the response, a latency-bound launch is saved outright): a^n (S numbers) | x^n (S secpar numbers), then the
    // sanity value and alpha -- ct1^(a^n) b^(n^2) | ct1^(x^n) y^(n^2) -- and further down s^(a^n) | s^(x^n).
    uint32_t* an = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
    uint32_t* xn = ctx->ws_t<uint32_t>((size_t)W2 * nb);
    {
      uint32_t* ax = concat_ab(ctx, al, nbs, xl, nb, W1);
      uint32_t* axn;
      if (pow_n2_crt_usable(sk)) {
        axn = pow_n2_crt(sk, ax, N, nbs + nb);               // the prover holds p and q
      } else {
        axn = ctx->ws_t<uint32_t>((size_t)W2 * (nbs + nb));
        shared_pow(ctx, mn2, ax, W1, N, nbs + nb, axn);
      }
      split_ab(ctx, axn, nbs, nb, 0, W2, an);
      split_ab(ctx, axn, nbs, nb, 1, W2, xn);
    }
    // Beside the a^n | x^n launch only where that launch leaves the second wave slot of the SIMDs free (one wave per SIMD or
    // less: 16 384 instances at secpar 1): a launch that fills both slots would lose one of them on half the chip to the side
    // launch for its whole length (measured at 32 768 instances: 48 -> 76 ms for 8 ms hidden).  Then s follows on the main stream.
    // (A side launch of a few dozen waves -- the statements of a secpar-40 call -- costs the big launch next to nothing.)
    const size_t lt0 = ctx->lanes_wanted ? ctx->lanes_wanted : (size_t)1024 * 64;
    const bool s_beside = (nbs + nb) * 2 <= lt0 || nbs * 2 * 8 <= lt0;
    if (s_beside) side.enter(inputs_ready);
    {
      // s = ExtractRandonness(ct1) at level two (operations.go:75-91): z = G^(-v) ct1 mod n^3 with v = Decrypt(ct1)
      // (operations.go:81-86) is only ever used modulo n (:88), and G^v = (1 + n)^v = 1 (mod n) whatever v is (the prover
      // requires G = n + 1): z = ct1 (mod n).  No decryption, no G^v, no inversion modulo n^3 -- the same s.
      uint32_t* z2 = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
      reduce_mod(ctx, mn2, c1s, W3, z2, nbs);
      if (sk->has_crt && sk->mp.WT * 2 == W1) {                                // z^nsInv mod n through p and q
        uint32_t* z1 = ctx->ws_t<uint32_t>((size_t)W1 * nbs);
        reduce_mod(ctx, mn, z2, W2, z1, nbs);
        qs = pow_n_crt(sk, z1, ns_inv, nbs);
      } else {
        qs = ctx->ws_t<uint32_t>((size_t)W1 * nbs);
        shared_pow(ctx, mn, z2, W2, ns_inv, nbs, qs);                          // z^nsInv mod n
      }
    }
    if (s_beside) side.leave();
    hipEvent_t an_ready = side.mark();
    uint32_t* bn2 = ctx->ws_t<uint32_t>((size_t)W3 * nbs);
    uint32_t* t3 = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    uint32_t* san = ctx->ws_t<uint32_t>((size_t)W3 * nbs);
    uint32_t* alp = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    if (crt3) {
      // ct1^(a^n) * b^(n^2) and ct1^(x^n) * y^(n^2): one interleaved ladder per number and CRT half, both batches in one launch
      uint32_t* cc2 = concat_ab(ctx, c1s, nbs, c1, nb, W3);
      uint32_t* ee2 = concat_ab(ctx, an, nbs, xn, nb, W2);
      uint32_t* by2 = concat_ab(ctx, bl, nbs, yl, nb, W1);
      uint32_t* o2 = ctx->ws_t<uint32_t>((size_t)W3 * (nbs + nb));
      pow_n3_crt(sk, cc2, W3, ee2, W2, &N2, nbs + nb, o2, by2, W1);
      split_ab(ctx, o2, nbs, nb, 0, W3, san);
      split_ab(ctx, o2, nbs, nb, 1, W3, alp);
    } else {
      shared3(bl, W1, N2, nbs, bn2);
      perlane3(c1s, an, W2, nbs, t3);
      modmul_arrays(ctx, mn3, t3, bn2, nbs, san);
    }
    int32_t* d_ok = ctx->ws_t<int32_t>(nbs);
    launch_equal(san, c2s, W3, nbs, S, d_ok, ctx->stream);
    std::vector<int32_t> hok(S);
    // ---- side stream, behind a^n (the Alpha ladders above are in flight on the main stream): a^-1 and (a^n)^-1 modulo n^2 for
    // every statement from ONE inversion tree (both batches side by side; a non-unit is flagged per lane and matters only if
    // one of its instances draws challenge bit 1), and the unit test of s b for the one-ladder form of the response
    uint32_t *qainv = ctx->ws_t<uint32_t>((size_t)W2 * nbs), *qani = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
    int32_t* d_badinv = ctx->ws_t<int32_t>(2 * nbs);
    bool any_badinv = false, sb_units = false, early = false;
    uint32_t *ge_all = nullptr, *es_all[2] = {nullptr, nullptr}, *eb_all[2] = {nullptr, nullptr};
    PreBases pre;
    const bool one_ladder = crt3 && ctx->use_lift && sk->eo_p.ok && sk->eo_q.ok && sk->eo_p.w == sk->eo_q.w &&
                            W2 <= 2 * sk->eo_p.modd.WT && W2 <= 2 * sk->eo_q.modd.WT;
    side.enter(an_ready);
    {
      uint32_t* a1 = ctx->ws_t<uint32_t>((size_t)W1 * nbs);
      launch_restride(al, nbs, S, mn.d_consts + (size_t)C_ONE * W1, a1, nbs, W1, ctx->stream);      // padding lanes: 1
      uint32_t* a2 = zext(ctx, a1, W1, W2, nbs);
      uint32_t* an1 = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
      launch_restride(an, nbs, S, mn2.d_consts + (size_t)C_ONE * W2, an1, nbs, W2, ctx->stream);
      uint32_t* inv2 = batch_inverse(ctx, mn2, concat2(ctx, a2, an1, W2, nbs), 2 * nbs, 2 * nbs, d_badinv, &any_badinv);
      split2(ctx, inv2, 0, W2, nbs, qainv);
      split2(ctx, inv2, 1, W2, nbs, qani);
      if (one_ladder) {
        uint32_t* sb = ctx->ws_t<uint32_t>((size_t)W1 * nbs);
        modmul_arrays(ctx, mn, qs, bl, nbs, sb);
        launch_restride(sb, nbs, S, mn.d_consts + (size_t)C_ONE * W1, sb, nbs, W1, ctx->stream);
        sb_units = all_units(ctx, mn, sb, nbs, S);
      }
      // The response's per-statement bases (s, b: residues modulo p^3, q^3, p^2, q^2 and digit forms) and, for EVERY instance, its
      // e = x a^-1, e^n = x^n (a^n)^-1 and the two exponents of the one-ladder response modulo the group orders: nothing here
      // depends on the challenge bits, so it is done now, beside the Alpha ladders, and the instances that draw bit 1 gather it
      // afterwards (between the hash and the response ladder there is then a handful of gathers instead of ~150 small kernels).
      // (only where the response ladder is certain to take pow_n3_crt_two's kernels whatever the number of bit-1 instances turns out
      // to be: its 7-bit window tables must fit the gather offsets even if every instance draws bit 1)
      if (one_ladder && sb_units && ctx->use_early && pre_bases_usable(sk) && triple_window_bits(nb, sk->mp3.triple.root->WT) == 7) {
        early = true;
        std::vector<uint32_t> stall(nb, 0);
        for (size_t i = 0; i < batch; ++i) stall[i] = (uint32_t)(i / secpar);
        const uint32_t* d_stall = ctx->upload_words(stall);
        auto per_inst_all = [&](const uint32_t* in, int w) {
          uint32_t* o = ctx->ws_t<uint32_t>((size_t)w * nb);
          launch_gather(in, nbs, d_stall, batch, o, nb, w, ctx->stream);
          return o;
        };
        uint32_t *ainv_a = per_inst_all(qainv, W2), *ani_a = per_inst_all(qani, W2), *gan_a = per_inst_all(an, W2);
        uint32_t* x2a = zext(ctx, xl, W1, W2, nb);
        ge_all = ctx->ws_t<uint32_t>((size_t)W2 * nb);
        modmul_arrays(ctx, mn2, x2a, ainv_a, nb, ge_all);                     // e = x a^-1 mod n^2 (ddleq.go:94-99)
        uint32_t* en_a = ctx->ws_t<uint32_t>((size_t)W2 * nb);
        modmul_arrays(ctx, mn2, xn, ani_a, nb, en_a);                         // e^n = x^n (a^n)^-1
        uint32_t* ls = ctx->ws_t<uint32_t>(nb);
        uint32_t* lb = ctx->ws_t<uint32_t>(nb);
        launch_exp_low_combine(xn, gan_a, en_a, ls, lb, nb, ctx->stream);
        Fork fo(ctx);
        for (int half = 0; half < 2; ++half) {
          fo.chain(half);
          const ExpOrder& eo_ = half ? sk->eo_q : sk->eo_p;
          const ModCtx& mm = eo_.modd;
          const size_t sm = (size_t)mm.WT * nb;
          uint32_t *am = ctx->ws_t<uint32_t>(sm), *em_ = ctx->ws_t<uint32_t>(sm), *xm = ctx->ws_t<uint32_t>(sm),
                   *pm = ctx->ws_t<uint32_t>(sm), *esm = ctx->ws_t<uint32_t>(sm), *ebm = ctx->ws_t<uint32_t>(sm),
                   *zero = ctx->ws_t<uint32_t>(sm);
          HIPCHK(hipMemsetAsync(zero, 0, sm * 4, ctx->stream));
          reduce_mod(ctx, mm, gan_a, W2, am, nb);
          reduce_mod(ctx, mm, en_a, W2, em_, nb);
          reduce_mod(ctx, mm, xn, W2, xm, nb);
          modmul_arrays(ctx, mm, am, em_, nb, pm);                                         // an en mod m
          launch_sub_mod(xm, pm, mm.d_nmod, esm, mm.WT, nb, ctx->stream);                   // xn - an en mod m
          launch_sub_mod(zero, em_, mm.d_nmod, ebm, mm.WT, nb, ctx->stream);                // -en mod m
          uint32_t* e1 = ctx->ws_t<uint32_t>((size_t)eo_.w * nb);
          uint32_t* e2 = ctx->ws_t<uint32_t>((size_t)eo_.w * nb);
          launch_exp_order_lift(ls, 1, esm, mm.WT, eo_.m_limbs.d, eo_.t, eo_.minv, e1, eo_.w, nb, ctx->stream);
          launch_exp_order_lift(lb, 1, ebm, mm.WT, eo_.m_limbs.d, eo_.t, eo_.minv, e2, eo_.w, nb, ctx->stream);
          es_all[half] = e1;
          eb_all[half] = e2;
        }
        fo.join();
        pre_bases(sk, zext(ctx, qs, W1, W3, nbs), zext(ctx, bl, W1, W3, nbs), nbs, pre);
      }
    }
    side.leave();
    // (a device-to-host copy into pageable memory holds the host until the stream has got there: it comes after the side work
    // has been issued, not before)
    HIPCHK(hipMemcpyAsync(hok.data(), d_ok, S * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < S; ++i)
      if (!hok[i]) api_throw(PGPU_ERR_INVALID, "cannot prove re-encryption because inputs are wrong");
    // ---- alpha = ct1^(x^n) * y^(n^2) mod n^3 (ddleq.go:81-87); with CRT it came out of the launch above
    uint32_t* yn2 = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    if (!crt3) {
      shared3(yl, W1, N2, nb, yn2);
      perlane3(c1, xn, W2, nb, t3);
      modmul_arrays(ctx, mn3, t3, yn2, nb, alp);
    }
    // ---- challenge bit = LSB SHA-256(ct2 || x || y || alpha)  (ddleq.go:91; ct1 is skipped: random_oracle.go:24-26)
    int32_t* chal = ctx->ws_t<int32_t>(nb);
    HIPCHK(hipMemsetAsync(chal, 0, nb * 4, ctx->stream));
    const uint32_t* parts[4] = {c2, xl, yl, alp};
    const int widths[4] = {W3, W1, W1, W3};
    launch_sha256_transcript(parts, widths, 4, nb, batch, nullptr, chal, ctx->stream);
    std::vector<int32_t> hch(batch);
    HIPCHK(hipMemcpyAsync(hch.data(), chal, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    side.join();                                             // the per-statement values are needed from here on
    // default outputs: e = x, f = y (chalBit false)
    uint32_t* eo = zext(ctx, xl, W1, W2, nb);
    uint32_t* fo = zext(ctx, yl, W1, W3, nb);
    // instances with challenge bit 1 and the statement each belongs to
    std::vector<uint32_t> idx, sti;
    for (size_t i = 0; i < batch; ++i)
      if (hch[i]) {
        idx.push_back((uint32_t)i);
        sti.push_back((uint32_t)(i / secpar));
      }
    if (!idx.empty()) {
      const size_t cnt = idx.size(), nbg = round_up(cnt, VM_BLOCK);
      if (getenv("PGPU_PROFILE_DUMP")) fprintf(stderr, "[pgpu] prove: %zu of %zu instances drew challenge bit 1 (response batch %zu)\n", cnt, batch, nbg);
      if (any_badinv) {      // ModInverse(a, n^2) of a non-unit a (ddleq.go:95) is undefined in the reference: refuse, as before
        std::vector<int32_t> hb(2 * nbs);
        HIPCHK(hipMemcpyAsync(hb.data(), d_badinv, 2 * nbs * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (uint32_t st : sti)
          if (hb[st] || hb[nbs + st])
            api_throw(PGPU_ERR_NOT_INVERTIBLE, "ModInverse: an element of the batch is not invertible modulo the modulus");
      }
      uint32_t* d_idx = ctx->upload_words(idx);
      uint32_t* d_sti = ctx->upload_words(sti);
      auto gat = [&](const uint32_t* in, int w) {            // per-instance array -> the compacted instances
        uint32_t* o = ctx->ws_t<uint32_t>((size_t)w * nbg);
        launch_gather(in, nb, d_idx, cnt, o, nbg, w, ctx->stream);
        return o;
      };
      auto per_inst = [&](const uint32_t* in, int w) {       // per-statement array -> one entry per compacted instance
        uint32_t* o = ctx->ws_t<uint32_t>((size_t)w * nbg);
        launch_gather(in, nbs, d_sti, cnt, o, nbg, w, ctx->stream);
        return o;
      };
      if (early) {
        // everything but the ladder itself is at hand (side stream, above): gather it for the instances with bit 1
        uint32_t* ge = gat(ge_all, W2);
        const uint32_t* es[2] = {gat(es_all[0], sk->eo_p.w), gat(es_all[1], sk->eo_q.w)};
        const uint32_t* eb[2] = {gat(eb_all[0], sk->eo_p.w), gat(eb_all[1], sk->eo_q.w)};
        pre.sti = d_sti;
        pre.cnt = cnt;
        uint32_t* c5 = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
        if (!pow_n3_crt_two(sk, nullptr, es, nullptr, eb, nbg, c5, &pre)) api_throw(PGPU_ERR_UNSUPPORTED, "internal: the early response path lost its kernel");
        uint32_t* y3 = zext(ctx, gat(yl, W1), W1, W3, nbg);
        uint32_t* gf = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
        modmul_arrays(ctx, mn3, y3, c5, nbg, gf);                             // f = y c mod n^3 (ddleq.go:114)
        launch_scatter(ge, nbg, d_idx, cnt, eo, nb, W2, ctx->stream);
        launch_scatter(gf, nbg, d_idx, cnt, fo, nb, W3, ctx->stream);
      } else {
      uint32_t *gx = gat(xl, W1), *gy = gat(yl, W1), *gxn = gat(xn, W2);
      uint32_t *ainv = per_inst(qainv, W2), *ani = per_inst(qani, W2), *gan = per_inst(an, W2), *sres = per_inst(qs, W1),
               *gb = per_inst(bl, W1);
      // e = x * a^-1 mod n^2 (ddleq.go:94-99)
      uint32_t* x2 = zext(ctx, gx, W1, W2, nbg);
      uint32_t* ge = ctx->ws_t<uint32_t>((size_t)W2 * nbg);
      modmul_arrays(ctx, mn2, x2, ainv, nbg, ge);
      uint32_t* s3 = zext(ctx, sres, W1, W3, nbg);
      // c = ((s^an * b)^en)^-1 * s^xn ; f = y * c mod n^3   (ddleq.go:103-114)
      // en = e^n mod n^2 (ddleq.go:104) = (x a^-1)^n = x^n (a^n)^-1: both powers are at hand, so an inversion replaces the ladder
      uint32_t* en = ctx->ws_t<uint32_t>((size_t)W2 * nbg);
      modmul_arrays(ctx, mn2, gxn, ani, nbg, en);
      uint32_t* c5 = nullptr;
      if (one_ladder) {
        // c = ((s^an b)^en)^-1 s^xn = s^(xn - an en) b^(-en)  (ddleq.go:107-112) whenever s and b are units: ONE interleaved
        // ladder with two per-number exponents, computed modulo the group orders of p^3 and q^3 (ord = 2^t m: the odd part
        // through a Montgomery product modulo m, the 2-part from the lowest limbs, then the CRT lift), instead of the ladders
        // s^an | s^xn, (.)^en and a batch inversion modulo n^3.  A non-unit s or b (the reference's ModInverse is then
        // undefined) keeps the literal sequence below and its error.  (The unit test ran per statement, on the side stream.)
        if (sb_units) {
          const uint32_t *es[2], *eb[2];
          uint32_t* ls = ctx->ws_t<uint32_t>(nbg);
          uint32_t* lb = ctx->ws_t<uint32_t>(nbg);
          launch_exp_low_combine(gxn, gan, en, ls, lb, nbg, ctx->stream);
          Fork fo(ctx);                                                  // the exponents modulo the two group orders side by side
          for (int half = 0; half < 2; ++half) {
            fo.chain(half);
            const ExpOrder& eo_ = half ? sk->eo_q : sk->eo_p;
            const ModCtx& mm = eo_.modd;
            const size_t sm = (size_t)mm.WT * nbg;
            uint32_t *am = ctx->ws_t<uint32_t>(sm), *em_ = ctx->ws_t<uint32_t>(sm), *xm = ctx->ws_t<uint32_t>(sm),
                     *pm = ctx->ws_t<uint32_t>(sm), *esm = ctx->ws_t<uint32_t>(sm), *ebm = ctx->ws_t<uint32_t>(sm),
                     *zero = ctx->ws_t<uint32_t>(sm);
            HIPCHK(hipMemsetAsync(zero, 0, sm * 4, ctx->stream));
            reduce_mod(ctx, mm, gan, W2, am, nbg);
            reduce_mod(ctx, mm, en, W2, em_, nbg);
            reduce_mod(ctx, mm, gxn, W2, xm, nbg);
            modmul_arrays(ctx, mm, am, em_, nbg, pm);                                       // an en mod m
            launch_sub_mod(xm, pm, mm.d_nmod, esm, mm.WT, nbg, ctx->stream);                 // xn - an en mod m
            launch_sub_mod(zero, em_, mm.d_nmod, ebm, mm.WT, nbg, ctx->stream);              // -en mod m
            uint32_t* e1 = ctx->ws_t<uint32_t>((size_t)eo_.w * nbg);
            uint32_t* e2 = ctx->ws_t<uint32_t>((size_t)eo_.w * nbg);
            launch_exp_order_lift(ls, 1, esm, mm.WT, eo_.m_limbs.d, eo_.t, eo_.minv, e1, eo_.w, nbg, ctx->stream);
            launch_exp_order_lift(lb, 1, ebm, mm.WT, eo_.m_limbs.d, eo_.t, eo_.minv, e2, eo_.w, nbg, ctx->stream);
            es[half] = e1;
            eb[half] = e2;
          }
          fo.join();
          uint32_t* b3n = zext(ctx, gb, W1, W3, nbg);
          uint32_t* o = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
          if (pow_n3_crt_two(sk, s3, es, b3n, eb, nbg, o)) c5 = o;
        }
      }
      uint32_t* cc = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
      uint32_t* sx = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
      if (!c5) {
        // s^(a^n) and s^(x^n) (ddleq.go:107,112): same base, independent exponents -> one launch
        uint32_t* ss2 = concat2(ctx, s3, s3, W3, nbg);
        uint32_t* ee2 = concat2(ctx, gan, gxn, W2, nbg);
        uint32_t* o2 = ctx->ws_t<uint32_t>((size_t)W3 * 2 * nbg);
        perlane3(ss2, ee2, W2, 2 * nbg, o2);
        split2(ctx, o2, 0, W3, nbg, cc);
        split2(ctx, o2, 1, W3, nbg, sx);
      }
      if (!c5) {
        uint32_t* b3 = zext(ctx, gb, W1, W3, nbg);
        uint32_t* cb = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
        modmul_arrays(ctx, mn3, cc, b3, nbg, cb);
        perlane3(cb, en, W2, nbg, cc);
        launch_restride(cc, nbg, cnt, mn3.d_consts + (size_t)C_ONE * W3, cc, nbg, W3, ctx->stream);
        uint32_t* ci = batch_inverse(ctx, mn3, cc, nbg, cnt);
        c5 = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
        modmul_arrays(ctx, mn3, ci, sx, nbg, c5);
      }
      uint32_t* y3 = zext(ctx, gy, W1, W3, nbg);
      uint32_t* gf = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
      modmul_arrays(ctx, mn3, y3, c5, nbg, gf);
      launch_scatter(ge, nbg, d_idx, cnt, eo, nb, W2, ctx->stream);
      launch_scatter(gf, nbg, d_idx, cnt, fo, nb, W3, ctx->stream);
      }
    }
    pack_result(ctx, alp, W3, nb, batch, alpha, ct_stride, mn3.nbytes, mem);
    pack_result(ctx, eo, W2, nb, batch, e_out, e_stride, std::min(e_stride, mn2.nbytes), mem);
    pack_result(ctx, fo, W3, nb, batch, f_out, ct_stride, mn3.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
}

int pgpu_ddleq_prove(const pgpu_seckey* sk, size_t batch, const uint8_t* ct1, const uint8_t* ct2, size_t ct_stride,
                     const uint8_t* a, const uint8_t* b, const uint8_t* x, const uint8_t* y, size_t n_stride, uint8_t* alpha,
                     uint8_t* e_out, size_t e_stride, uint8_t* f_out, int mem) {
  if (!sk || !ct1 || !ct2 || !a || !b || !x || !y || !alpha || !e_out || !f_out) return fail(PGPU_ERR_INVALID, "null argument");
  return guarded([&] { ddleq_prove_impl(sk, batch, 1, ct1, ct2, ct_stride, a, b, x, y, n_stride, alpha, e_out, e_stride, f_out, mem); });
}

int pgpu_ddleq_prove_secpar(const pgpu_seckey* sk, size_t n_statements, size_t secpar, const uint8_t* ct1, const uint8_t* ct2,
                            size_t ct_stride, const uint8_t* a, const uint8_t* b, const uint8_t* x, const uint8_t* y, size_t n_stride,
                            uint8_t* alpha, uint8_t* e_out, size_t e_stride, uint8_t* f_out, int mem) {
  if (!sk || !ct1 || !ct2 || !a || !b || !x || !y || !alpha || !e_out || !f_out) return fail(PGPU_ERR_INVALID, "null argument");
  if (secpar && n_statements > ((size_t)1 << 31) / secpar) return fail(PGPU_ERR_INVALID, "n_statements * secpar is too large");
  return guarded([&] {
    ddleq_prove_impl(sk, n_statements, secpar, ct1, ct2, ct_stride, a, b, x, y, n_stride, alpha, e_out, e_stride, f_out, mem);
  });
}

}  // extern "C"

extern "C" {

int pgpu_share_zkp_prove(const pgpu_pubkey* pk, int total_servers, const uint8_t* share_be, size_t share_len,
                         const uint8_t* vkey_be, size_t vkey_len, size_t batch, const uint8_t* c, size_t c_stride,
                         const uint8_t* r, size_t r_stride, uint8_t* dec, size_t dec_stride, uint8_t* e_out, uint8_t* z_out,
                         size_t z_stride, int mem) {
  if (!pk || !share_be || !vkey_be || !c || !r || !dec || !e_out || !z_out) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    if (batch == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
    ctx->bind();
    const ModCtx& mc = pk->mn2;
    const int W2 = mc.WT;
    const BigU share = BigU::from_be(share_be, share_len), delta = factorial_big(total_servers);
    if (r_stride > mc.nbytes + 96) api_throw(PGPU_ERR_INVALID, "r stride larger than the byte length of n^2 plus 96 (r < n^2, thresholdkey.go:233)");
    const pgpu_pubkey::FixedBase fb = ensure_fixed_base(const_cast<pgpu_pubkey*>(pk), BigU::from_be(vkey_be, vkey_len),
                                                        std::max(mc.nbits + 384, r_stride * 8));
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    const size_t sw = (size_t)W2 * nb;
    if (c_stride != mc.nbytes) api_throw(PGPU_ERR_INVALID, "ciphertext stride must be the byte length of n^2");
    uint32_t* cl = ctx->ws_t<uint32_t>(sw);
    unpack_operand(ctx, c, c_stride, c_stride, batch, mem, cl, W2, nb);
    // Decryption = c^(2 delta s_i) mod n^2                                   (thresholdkey.go:229,192-201)
    ModexpPlan pd = modexp_alloc(ctx, mc, nb, 32);
    HIPCHK(hipMemcpyAsync(pd.in(), cl, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
    modexp_shared_run(ctx, mc, pd, share * (BigU(2) * delta), false, false, true);
    // r, a = (c^4)^r mod n^2, b = V^r mod n^2                                (thresholdkey.go:241-245)
    const int wr = std::max<int>(1, (int)((r_stride * 8 + LB - 1) / LB));
    uint32_t* rl = ctx->ws_t<uint32_t>((size_t)wr * nb);
    unpack_operand(ctx, r, r_stride, r_stride, batch, mem, rl, wr, nb);
    uint32_t* c4m = ctx->ws_t<uint32_t>(sw);
    {
      uint32_t* memv = ctx->ws_t<uint32_t>(sw * 2);
      HIPCHK(hipMemcpyAsync(memv, cl, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
      Prog p;
      p.op(VM_LOAD, 0); p.op(VM_MULC, C_R2); p.op(VM_SQR); p.op(VM_SQR); p.op(VM_MULC, C_ONE); p.op(VM_STORE, 1); p.end();
      SegSpec sg{&mc, &p, memv, nullptr};
      run_vm(ctx, nb, sg, nullptr, false);
      launch_canon(memv + sw, mc.d_nmod, W2, nb, ctx->stream);
      HIPCHK(hipMemcpyAsync(c4m, memv + sw, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    uint32_t* a = ctx->ws_t<uint32_t>(sw);
    uint32_t* b = ctx->ws_t<uint32_t>(sw);
    perlane_pow(ctx, mc, c4m, rl, wr, nb, a);
    comb_pow(ctx, mc, fb, rl, wr, nb, b);
    uint32_t* dg = zkp_hash(ctx, W2, a, b, cl, pd.out(), nb, batch);
    // E and Z = r + E * delta * s_i (plain integers, thresholdkey.go:313-317)
    uint32_t* el = ctx->ws_t<uint32_t>(10 * nb);
    launch_digest_to_limbs(dg, el, nb, ctx->stream);
    const BigU ds = delta * share;
    const int wds = std::max<int>(1, (int)((ds.bit_length() + LB - 1) / LB));
    const int wz = std::max(wr, wds + 10) + 1;
    if (z_stride * 8 < (size_t)LB * wz && z_stride * 8 < std::max((size_t)r_stride * 8, ds.bit_length() + 256) + 1)
      api_throw(PGPU_ERR_INVALID, "z stride too small for r + E*delta*share");
    uint32_t* d_ds = ctx->upload_words(ds.to_limbs(LB, wds));
    uint32_t* zl = ctx->ws_t<uint32_t>((size_t)wz * nb);
    launch_mul_const_add(el, 10, d_ds, wds, rl, wr, 0, zl, wz, nb, ctx->stream);
    pack_result(ctx, pd.out(), W2, nb, batch, dec, dec_stride, mc.nbytes, mem);
    pack_result(ctx, zl, wz, nb, batch, z_out, z_stride, std::min(z_stride, (size_t)((LB * wz + 7) / 8)), mem);
    // E as 32 big-endian bytes
    uint32_t* e10 = el;
    pack_result(ctx, e10, 10, nb, batch, e_out, 32, 32, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_share_zkp_verify(const pgpu_pubkey* pk, const uint8_t* vkey_be, size_t vkey_len, const uint8_t* vi_be, size_t vi_len,
                          size_t batch, const uint8_t* c, size_t c_stride, const uint8_t* dec, size_t dec_stride,
                          const uint8_t* e, const uint8_t* z, size_t z_stride, int32_t* ok, int mem) {
  if (!pk || !vkey_be || !vi_be || !c || !dec || !e || !z || !ok) return fail(PGPU_ERR_INVALID, "null argument");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    if (batch == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
    ctx->bind();
    const ModCtx& mc = pk->mn2;
    const int W2 = mc.WT;
    pgpu_pubkey* pkm = const_cast<pgpu_pubkey*>(pk);
    // Z = r + E * l! * s_i with r < n^2, E < 2^256, s_i < n^2: at most ~n^2 bits + 256 + log2(l!) bits.  The stride sizes a
    // table that is kept with the key: bound it (an unbounded caller-chosen stride would grow the key without limit).
    if (z_stride > mc.nbytes + 96) api_throw(PGPU_ERR_INVALID, "z stride larger than the byte length of n^2 plus 96");
    const size_t zbits = std::max(z_stride * 8, mc.nbits + 384);
    const pgpu_pubkey::FixedBase fbV = ensure_fixed_base(pkm, BigU::from_be(vkey_be, vkey_len), zbits);
    const pgpu_pubkey::FixedBase fbI = ensure_fixed_base(pkm, BigU::from_be(vi_be, vi_len), 256);
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    const size_t sw = (size_t)W2 * nb;
    if (c_stride != mc.nbytes || dec_stride != mc.nbytes) api_throw(PGPU_ERR_INVALID, "c / decryption stride must be the byte length of n^2");
    uint32_t* cl = ctx->ws_t<uint32_t>(sw);
    uint32_t* dl = ctx->ws_t<uint32_t>(sw);
    unpack_operand(ctx, c, c_stride, c_stride, batch, mem, cl, W2, nb);
    unpack_operand(ctx, dec, dec_stride, dec_stride, batch, mem, dl, W2, nb);
    const int wz = std::max<int>(1, (int)((z_stride * 8 + LB - 1) / LB));
    uint32_t* zl = ctx->ws_t<uint32_t>((size_t)wz * nb);
    unpack_operand(ctx, z, z_stride, z_stride, batch, mem, zl, wz, nb);
    uint32_t* el = ctx->ws_t<uint32_t>(10 * nb);
    unpack_operand(ctx, e, 32, 32, batch, mem, el, 10, nb);
    // c^4 mod n^2 and c_i^2 mod n^2 in one program (two numbers per lane would need two x registers: two programs)
    uint32_t* c4m = ctx->ws_t<uint32_t>(sw);
    uint32_t* d2m = ctx->ws_t<uint32_t>(sw);
    {
      uint32_t* memv = ctx->ws_t<uint32_t>(sw * 4);
      HIPCHK(hipMemcpyAsync(memv, cl, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(memv + sw, dl, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
      Prog p;
      p.op(VM_LOAD, 0); p.op(VM_MULC, C_R2); p.op(VM_SQR); p.op(VM_SQR); p.op(VM_MULC, C_ONE); p.op(VM_STORE, 2);
      p.op(VM_LOAD, 1); p.op(VM_MULC, C_R2); p.op(VM_SQR); p.op(VM_MULC, C_ONE); p.op(VM_STORE, 3);
      p.end();
      SegSpec sg{&mc, &p, memv, nullptr};
      run_vm(ctx, nb, sg, nullptr, false);
      launch_canon(memv + 2 * sw, mc.d_nmod, W2, nb, ctx->stream);
      launch_canon(memv + 3 * sw, mc.d_nmod, W2, nb, ctx->stream);
      HIPCHK(hipMemcpyAsync(c4m, memv + 2 * sw, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(d2m, memv + 3 * sw, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // a = (c^4)^Z * ((c_i^2)^E)^-1,  b = V^Z * (v_i^E)^-1  mod n^2          (thresholdkey.go:294-311)
    uint32_t* a1 = ctx->ws_t<uint32_t>(sw);
    uint32_t* a2 = ctx->ws_t<uint32_t>(sw);
    uint32_t* b1 = ctx->ws_t<uint32_t>(sw);
    uint32_t* b2 = ctx->ws_t<uint32_t>(sw);
    perlane_pow(ctx, mc, c4m, zl, wz, nb, a1);
    perlane_pow(ctx, mc, d2m, el, 10, nb, a2);
    comb_pow(ctx, mc, fbV, zl, wz, nb, b1);
    comb_pow(ctx, mc, fbI, el, 10, nb, b2);
    // A proof whose Decryption (or v_i) is not a unit has no inverse: mpz_invert leaves the reference's a2 / b2 undefined
    // and the hash comparison fails; here such a lane is rejected (ok = 0) without disturbing the other proofs.
    int32_t* bad_a = ctx->ws_t<int32_t>(nb);
    int32_t* bad_b = ctx->ws_t<int32_t>(nb);
    uint32_t* a2i = batch_inverse(ctx, mc, a2, nb, batch, bad_a);
    uint32_t* av = ctx->ws_t<uint32_t>(sw);
    modmul_arrays(ctx, mc, a1, a2i, nb, av);
    uint32_t* b2i = batch_inverse(ctx, mc, b2, nb, batch, bad_b);
    uint32_t* bv = ctx->ws_t<uint32_t>(sw);
    modmul_arrays(ctx, mc, b1, b2i, nb, bv);
    uint32_t* dg = zkp_hash(ctx, W2, av, bv, cl, dl, nb, batch);
    uint32_t* e2 = ctx->ws_t<uint32_t>(10 * nb);
    launch_digest_to_limbs(dg, e2, nb, ctx->stream);
    int32_t* d_ok = ctx->ws_t<int32_t>(nb);
    launch_equal(e2, el, 10, nb, batch, d_ok, ctx->stream);
    launch_clear_where(bad_a, batch, d_ok, ctx->stream);
    launch_clear_where(bad_b, batch, d_ok, ctx->stream);
    HIPCHK(hipMemcpyAsync(ok, d_ok, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

int pgpu_const_mult(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* c, size_t c_stride,
                    const uint8_t* k, size_t k_len, size_t k_stride, uint8_t* out, size_t out_stride, int mem) {
  if (!pk) return fail(PGPU_ERR_INVALID, "null key");
  pgpu_ctx* ctx = pk->ctx;
  return guarded([&] {
    check_batch_args(c, out, batch);
    if (!k) api_throw(PGPU_ERR_INVALID, "null exponent");
    const ModCtx& mc = cipher_mod(pk, level);
    ctx->bind();
    ctx->reset_ws();
    const size_t nb = round_up(batch, VM_BLOCK);
    const bool perlane = k_stride != 0;
    ModexpPlan pl = modexp_alloc(ctx, mc, nb, perlane ? 16 : 32);
    unpack_mod(ctx, mc, c, c_stride, batch, mem, pl.in(), nb);
    if (!perlane) {
      modexp_shared_run(ctx, mc, pl, BigU::from_be(k, k_len), false, false, true);
    } else {
      int we = std::max<int>(1, (int)((k_len * 8 + LB - 1) / LB));
      uint32_t* exps = ctx->ws_t<uint32_t>((size_t)we * nb);
      unpack_operand(ctx, k, k_stride, k_len, batch, mem, exps, we, nb);
      modexp_perlane_run(ctx, mc, pl, exps, we, false, false);
    }
    pack_result(ctx, pl.out(), mc.WT, nb, batch, out, out_stride, mc.nbytes, mem);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

}  // extern "C"
