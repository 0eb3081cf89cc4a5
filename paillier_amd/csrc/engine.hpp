// engine.hpp -- internal header of libpaillier_hip.so (nothing here is part of the C ABI: that is include/paillier_hip.h).
//
// Translation units:
//   ctx.cpp        contexts, runtime switches, profile of the last call, error text
//   keys.cpp       key handles: moduli and their Montgomery / pair / digit-form constants, CRT material
//   vm_emit.cpp    programs of the big-integer VM and run_vm, the one place that launches a VM kernel
//   modexp.cpp     ladders on the kernel family that fits, reduction, batch inversion; pgpu_modexp / modmul / modinv
//   paillier.cpp   Encrypt / Decrypt / Add / Sub / ConstMult, randomness
//   threshold.cpp  PartialDecrypt (batch forms), Combine, share ZKP
//   ddleq.cpp      NestedRandomize, DDLEQ prove / verify, RandomOracleDigest
//   wire.cpp       the gob wire format of Ciphertext at the ABI (batch <-> flat buffers)
//   debug.cpp      test hooks (include/paillier_hip_debug.h)
//   plan.hpp       every size decision (lanes per number, window widths, split gates): pure functions, CPU-tested
//   kernels.hip / asm_loader.cpp / gen_vm_asm.py   the device side
// Host code here runs once per key or once per batch call; everything per ciphertext happens on the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/paillier_hip.h"
#include "hostbig.hpp"
#include "kernels.h"
#include "plan.hpp"

namespace pgi {

using hostbig::BigU;

constexpr int LB = plan::LB;
extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);

struct HipError { hipError_t e; const char* what; };
#define HIPCHK(x)                                                      \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) throw ::pgi::HipError{e_, #x};               \
  } while (0)

struct ApiError { int code; std::string msg; };
[[noreturn]] void api_throw(int code, const std::string& m);

inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// overwrite key material before its storage is released (not elided: volatile stores)
void wipe(void* p, size_t n);

template <class T> void wipe_vec(std::vector<T>& v) { if (!v.empty()) wipe(v.data(), v.size() * sizeof(T)); }
// wipes host copies of secret exponents on EVERY exit of the enclosing scope, exceptions included
template <class V> struct WipeOnExit {
  V& v;
  explicit WipeOnExit(V& v_) : v(v_) {}
  ~WipeOnExit() { for (auto& e : v) wipe_vec(e.d); }
};

}  // namespace pgi
using namespace pgi;

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
  bool use_exclusive = true; // concurrent small launches of one call take a compute unit each (see exclusive_call; pgpu_ctx_set_flag("exclusive", 0): the dispatcher's placement)
  bool exclusive_call = false;   // set by a protocol function for the length of a call whose concurrent launches together fit the chip's compute units: every
                                 // workgroup then asks for the whole LDS of a CU, so the dispatcher cannot stack the side lanes' workgroups on the CUs the main
                                 // launch runs on (it starts every queue's workgroups from the same CUs: 17.5 -> 28 ms for a^n | x^n of 2 048 instances)
  bool use_base_early = true;  // the links of the prover's side chains (ct1's structure chain, the preparation of the response) run beside the main stream's ladders (pgpu_ctx_set_flag("base_early", 0): they wait for an empty compute unit, i.e. for the end of a ladder)
  bool use_lanes16 = true;       // shared-exponent ladders modulo n^2 of up to 2 048 numbers on sixteen lanes per number (pgpu_ctx_set_flag("lanes16", 0): eight)
  bool use_prime_lanes = true;   // ladders modulo the 37-limb primes of small batches on four lanes per number (pgpu_ctx_set_flag("prime_lanes", 0): one lane, the unrolled kernel)
  bool use_late = true;      // the DDLEQ prover's response for few instances per statement through the structure of the unit group AFTER the hash: b's plaintext for the statements with a bit-1 instance only (pgpu_ctx_set_flag("late", 0): the one-ladder response on s and b themselves)
  bool use_spread = true;    // a main-stream ladder of at most one workgroup per CU asks for just over half a CU's LDS (plan::lds_share; pgpu_ctx_set_flag("spread", 0): the dispatcher's placement)
  bool use_exclusive_short = true;    // short programs of a prover call take a CU per workgroup too (pgpu_ctx_set_flag("exclusive_short", 0): only ladders do; measured equal -- plan::lds_share)
  uint32_t stream_cus = plan::kChipCUs;   // compute units this context's stream may use (pgpu_ctx_set_flag("cu_partition", ...) narrows it)
  bool use_w74 = true;       // 74-limb two-slice moduli on the wave-sliced assembly kernel (pgpu_ctx_set_flag("w74", 0): four lanes of 37 limbs)
  bool use_exp_order = true; // the key holder's exponents modulo p^3, q^3 reduced modulo the group orders (pgpu_ctx_set_flag("exp_order", 0): as given)
  int lds_force = -1;        // >= 0: the LDS request of the next assembly launches (set for the length of a scope by protocol code that places
                             // launches against each other: plan::kLate*; otherwise plan::lds_share decides)
  bool use_background = false;   // the prover's side-lane ladders at wave priority 0 (pgpu_ctx_set_flag("background", 1): measured, no gain -- ddleq.cpp)
  bool background_launch = false; // set around side-lane ladders of a LARGE call (see run_vm): their long programs run at wave priority 0
  bool use_handover = true;  // a power modulo n^2 that is only needed modulo n^2 by the next ladder modulo n^3 stays in pair form: (a0, a1, 0) is its digit form (pgpu_ctx_set_flag("handover", 0): exit and re-entry)
  bool use_muls = true;      // bucket products of the shared chain as VM_MULS where the kernel has it (pgpu_ctx_set_flag("muls", 0): LOAD / MUL / STORE)
  bool use_struct = true;    // the key holder's ct^e y^(n^2) mod n^3 through the structure of the unit group: plaintext of ct, ladders modulo the primes, Teichmueller lift (pgpu_ctx_set_flag("struct", 0): the ladders on ct itself)
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
    pinned_used = 0;
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
  // Page-locked host memory for small results the host reads LATER (flags, a tree's root): a device-to-host copy into pageable memory
  // holds the host until the stream has got there, one into pinned memory is just another command of the stream.  Reset per call.
  uint8_t* pinned_base = nullptr;
  size_t pinned_used = 0;
  static constexpr size_t kPinnedBytes = (size_t)1 << 20;
  void* pinned(size_t bytes) {
    bytes = round_up(bytes ? bytes : 1, 64);
    if (!pinned_base) HIPCHK(hipHostMalloc((void**)&pinned_base, kPinnedBytes, hipHostMallocDefault));
    if (pinned_used + bytes > kPinnedBytes) throw HipError{hipErrorOutOfMemory, "pinned scratch"};
    void* ptr = pinned_base + pinned_used;
    pinned_used += bytes;
    return ptr;
  }
  // The context's four side streams, created TOGETHER and in this order the first time any of them is wanted: the runtime maps a
  // process's streams onto four hardware queues in the order they are created, so that the fourth and fifth stream of a process share
  // a queue -- and which two of a context's lanes that is must not depend on which entry point happened to be called first.  (Created
  // lazily in call order, the verifier's x^(e0) ladder on lane 2 shared a queue with the entry chains on lane 1 whenever a prover call
  // had come first: 118 ms instead of 75 for 2 048 instances.)  In this order lanes 2 and 3 share: the prover's closed form (lane 2)
  // waits for the decryption on lane 3 anyway, and nothing else uses lane 3 beside a ladder on lane 2.
  void ensure_side_streams() {
    if (side) return;
    HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    for (auto& l : side_l) HIPCHK(hipStreamCreateWithFlags(&l, hipStreamNonBlocking));
  }
  ~pgpu_ctx() {
    if (pinned_base) (void)hipHostFree(pinned_base);
    wipe_ws();
    for (auto& c : chunks) (void)hipFree(c.p);
    for (auto& e : evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : sync_evs) (void)hipEventDestroy(e);
    if (side) (void)hipStreamDestroy(side);
    for (auto l : side_l) if (l) (void)hipStreamDestroy(l);
    if (own_stream) (void)hipStreamDestroy(stream);
  }
};

namespace pgi {

// Work of ONE call on two streams.  Every helper of this file issues to ctx->stream; between enter() and leave() that is the
// side stream.  enter(after) orders the side work behind an event of the main stream (mark()), leave() goes back WITHOUT making
// the main stream wait, join() makes the main stream wait for everything the side stream was given.  Host-side synchronisation
// inside side work (the root of an inversion tree goes through the host) waits for the side stream only: issue the main
// stream's long launch BEFORE entering, and it runs meanwhile.  Disabled (ctx->use_side == false): everything stays on the one
// stream, in program order -- the same results.
struct ForceLds {            // ctx->lds_force for the length of a scope (value < 0: no change)
  pgpu_ctx* c; int was;
  ForceLds(pgpu_ctx* c_, int value) : c(c_), was(c_->lds_force) { if (value >= 0) c->lds_force = value; }
  ~ForceLds() { c->lds_force = was; }
};

struct SideStream {
  pgpu_ctx* c;
  hipStream_t main_stream;
  hipStream_t& s;                           // the lane's stream (created on first use)
  bool on, entered = false, dirty = false;
  // lane 0: work of a call that is independent of its ladders (the prover's per-statement chains).  lanes 1 .. 3: independent
  // CHAINS of small kernels -- the entry into digit form of the q-half next to the p-half's, of operand y next to x's; each such
  // kernel fills a fraction of the chip for tens of microseconds, chains side by side take about the time of one.
  explicit SideStream(pgpu_ctx* c_, int lane = 0) : c(c_), main_stream(c_->stream), s(lane ? c_->side_l[lane - 1] : c_->side), on(c_->use_side) {
    if (on && !s) c->ensure_side_streams();
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
  // the sixteen-lane pair kernel (GenQ16: every digit in eight lanes of h16 / 8 limbs): the same constants for digits of h16 limbs
  int h16 = 0;
  const uint32_t* consts16 = nullptr;
  const uint32_t* tconsts16 = nullptr;
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
  // the sixteen-lane variant (GenQ12: four lanes per digit, digits of h12 limbs, radix R_h12): kconsts for h12-limb digits, and constants
  // in digit form [3][3 h12]: 1 (= R_h12), R_h12^2 R_H^-1 (entry from the radix-R_H digit form), R_H (exit)
  int h12 = 0;
  const uint32_t* kconsts12 = nullptr;
  const uint32_t* tconsts12 = nullptr;
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

  // wl, k: a shape other than the narrowest that holds the modulus (the four-lane twins of the 37-limb primes, plan::prime_lanes)
  void init(pgpu_ctx* c, const BigU& n, int wl = 0, int k = 0) {
    ctx = c;
    N = n;
    if (!N.is_odd() || N.bit_length() < 2) api_throw(PGPU_ERR_INVALID, "modulus must be odd and at least 3");
    nbits = N.bit_length();
    nbytes = (nbits + 7) / 8;
    if (wl) {
      if (nbits + 3 > (size_t)LB * wl * k) api_throw(PGPU_ERR_INVALID, "internal: the modulus does not fit the shape asked for");
      WL = wl;
      K = k;
    } else if (!pick_shape(nbits, WL, K)) api_throw(PGPU_ERR_UNSUPPORTED, "modulus wider than 9405 bits is not built");
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
extern std::atomic<bool> g_wave_priorities;
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
    if (o == VM_MULCV || o == VM_MULCV7 || o == VM_MULV5 || o == VM_MULV7 || o == VM_STORET || o == VM_MULVT || o == VM_MULVT5) wide_gathers = true;   // (kernels without these opcodes must not get the program)
    if (o == VM_MULV || o == VM_MULV5 || o == VM_MULV7 || o == VM_MULVT || o == VM_MULVT5) {
      has_mulv = true;
      gather_slots = std::max<uint32_t>(gather_slots, o == VM_MULV ? 17u : o == VM_MULVT ? 18u : o == VM_MULV5 ? 33u : o == VM_MULVT5 ? 49u : 129u);
    }
    if (aux >> 22) api_throw(PGPU_ERR_INVALID, "internal: table slot does not fit the instruction word");
    w.push_back(o | (aux << 8));
    w.push_back(arg);
    if (o == VM_SQR || o == VM_MUL || o == VM_MULC || o == VM_MULV || o == VM_MULCV || o == VM_MULCV7 || o == VM_MULV5 || o == VM_MULV7 || o == VM_MULS || o == VM_MULVT || o == VM_MULVT5) montmuls += 1;
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
    if (!g_wave_priorities.load(std::memory_order_relaxed)) return;
    if (montmuls < 256) {
      // a short program is a link of a chain between dependent ladders: top priority for all of it (next to a side lane's long ladder
      // at priority 3 .. 1 a priority-0 wave only gets the slots the ladder leaves: 3.4 ms for three products, r04 trace)
      for (size_t i = 0; i + 1 < w.size(); i += 2) w[i] = (w[i] & 0x3FFFFFFFu) | (3u << 30);
      return;
    }
    double done = 0;
    for (size_t i = 0; i + 1 < w.size(); i += 2) {
      const uint32_t o = w[i] & 0xFFu;
      const double f = done / montmuls;
      const uint32_t quarter = f < 0.80 ? 0u : f < 0.96 ? 1u : f < 0.992 ? 2u : 3u;
      w[i] = (w[i] & 0x3FFFFFFFu) | ((3u - quarter) << 30);
      if (o == VM_SQR || o == VM_MUL || o == VM_MULC || o == VM_MULV || o == VM_MULCV || o == VM_MULCV7 || o == VM_MULV5 || o == VM_MULV7 || o == VM_MULS || o == VM_MULVT || o == VM_MULVT5) done += 1;
    }
  }
};

constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;

// ---- vm_emit.cpp ---------------------------------------------------------------------------------------------------------------
using plan::perlane_windows;
using plan::perlane_table_slots;
using plan::dual_sliding_bits;
using plan::triple_window_bits;
void emit_to_mont(Prog& p, uint32_t lo, uint32_t hi, uint32_t tmp);
void emit_modexp_shared(Prog& p, const BigU& e, uint32_t in_lo, uint32_t in_hi, uint32_t tmp, uint32_t out, uint32_t tab,
                        uint32_t post_slot, bool skip_zero_digits, bool raw = false);
void emit_modexp_perlane(Prog& p, int we, uint32_t in_lo, uint32_t in_hi, uint32_t tmp, uint32_t out, uint32_t tab,
                         uint32_t post_slot, int raw_one = -1, int wb = 4, bool nm4 = false);
void emit_modexp_dual(Prog& p, int we, const BigU& e, uint32_t in1, uint32_t in2, uint32_t tmp, uint32_t out, uint32_t tab1,
                      uint32_t tab2, int raw_one = -1, int wb = 4, bool nm4 = false);

struct SharedBase { BigU e; uint32_t in; uint32_t tab; };
struct PerNumberBase { int we; uint32_t in; uint32_t tab; uint32_t first_window; };   // windows first_window.. of the `digits` rows
void emit_modexp_multi(Prog& p, const std::vector<PerNumberBase>& pn, int wb, const std::vector<SharedBase>& sh, uint32_t tmp,
                       uint32_t out, uint32_t one, bool nm4 = false);
void emit_multi_exp_shared_base(Prog& p, const std::vector<BigU>& es, uint32_t in, uint32_t bp, uint32_t run, uint32_t acc,
                                uint32_t out0, uint32_t bucket0, int w, uint32_t one_const, bool muls = false);

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

// The moduli the ladders modulo the primes of `nb` numbers run on (plan::prime_lanes): the key's p, q as they are -- one lane per number
// -- or their four-lane twins, whose slots are Hs = 40 limbs wide.  A residue stays an H-limb array either way: the first H rows of a
// twin's slot ARE that array (limb-major; a lazy value below 2 p has nothing in the rows above), and prime_slot_fill zeroes the rows
// above a residue that goes in.
struct PrimeShape {
  const ModCtx* m[2] = {nullptr, nullptr};
  int H = 0, Hs = 0;
};
PrimeShape prime_shape(const pgpu_seckey* sk, size_t nb, int beside = 1);
// x^(e[half]) in pair form modulo p^2 | q^2 on the eight-lane pair kernel, both halves in one launch (plan::crt_pair_lanes8): in[half] /
// out[half] = pair digits (d0 | d1, 2 H limbs, stride nb) in the radix-R_H pair form of the one- and two-lane kernels
bool crt_pair8_usable(const pgpu_seckey* sk, size_t nb);
void crt_pair8_ladders(const pgpu_seckey* sk, const BigU e[2], const uint32_t* const in[2], uint32_t* const out[2], size_t nb);     // beside: plan::prime_lanes
void prime_slot_fill(pgpu_ctx* ctx, const PrimeShape& ps, uint32_t* slot, const uint32_t* src, size_t nb);

// launch one VM kernel with 1 to 3 segments of `nb` numbers each (same modulus shape; s2 only together with s1)
void run_vm(pgpu_ctx* ctx, size_t nb, const SegSpec& s0, const SegSpec* s1, bool profile, size_t launch_nb = 0,
            const SegSpec* s2 = nullptr);
void unpack_operand(pgpu_ctx* ctx, const uint8_t* buf, size_t stride, size_t nbytes, size_t count, int mem,
                    uint32_t* out, int wt, size_t nb);
void pack_result(pgpu_ctx* ctx, const uint32_t* in, int wt, size_t nb, size_t count, uint8_t* out, size_t stride,
                 size_t nbytes, int mem);

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

}  // namespace pgi

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
  struct AltTab { bool built = false; BigU hs; size_t kbits = 0; } alt[2];  // the fixed base h_s of AltEncrypt (its comb table: comb7)
  struct FixedBase { BigU base; int idx; int nwin; };
  std::vector<FixedBase> fixed_bases;   // comb tables of other fixed bases mod n^2 (verification keys)
  // 7-bit comb tables (VM_MULCV7) of fixed bases modulo n^2 / n^3, built on the device and kept there: entry 4 + 128 i + d =
  // base^(d 128^i) in Montgomery form, entries 0 .. 3 = the modulus' standard constants (the programs' LOADC / MULC indices)
  struct Comb7 {
    BigU base;
    int level = 0;
    int nwin = 0;
    uint32_t* d_table = nullptr;
    size_t words = 0;
    Comb7() = default;
    Comb7(const Comb7&) = delete;
    Comb7& operator=(const Comb7&) = delete;
    ~Comb7() { if (d_table) { (void)hipMemset(d_table, 0, words * 4); (void)hipFree(d_table); } }
  };
  std::vector<std::unique_ptr<Comb7>> comb7;
  struct CachedInverse { BigU x, inv; bool unit; };
  std::vector<CachedInverse> inverse_cache;   // inverses modulo n^2 of per-key constants (verification keys), taken once on the host
  DevLimbs pairn_consts;          // n | Cadj | pad for the two-lane pair kernel (mn2.pairn points here)
  DevLimbs pairn_consts16, pairn_tconsts16; // ... and for the sixteen-lane pair kernel (80-limb digits)
  DevLimbs pairn_consts8, pairn_tconsts8;   // the same for the eight-lane pair kernel (76-limb digits) and its three constants
  DevLimbs triple_kconsts;        // n | (C1_i, C2_i) pairs | pad for the three-digit kernel (mn3->triple points here)
  DevLimbs triple_tconsts;        // its constants in digit form
  DevLimbs triple_kconsts12, triple_tconsts12;   // the same for the sixteen-lane three-digit kernel (76-limb digits)
  std::vector<std::pair<int, int>> combine_consts;  // (total servers l, index of (4 (l!)^2)^-1 * R mod n in mn.consts)
};

namespace pgi {
// order of the unit group modulo pr^3 (power = 2: pr^2 (pr - 1)) or modulo pr (power = 0: pr - 1), split as 2^t m for the
// Montgomery reduction modulo its odd part
struct ExpOrder {
  bool ok = false;
  BigU ord;
  ModCtx modd;           // m
  int t = 0;
  uint32_t minv = 0;     // m^-1 mod 2^t
  DevLimbs m_limbs;      // m as w limbs
  int w = 0;             // limbs of a reduced exponent (< 2 ord)
  void init(pgpu_ctx* ctx, const BigU& pr, int power = 2) {
    ord = pr - BigU(1);
    for (int i = 0; i < power; ++i) ord = ord * pr;
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
}  // namespace pgi

struct pgpu_seckey {
  pgpu_ctx* ctx;
  const pgpu_pubkey* pk;
  BigU lambda;
  bool has_crt = false;
  BigU p, q;
  ModCtx mp, mq, mp2, mq2;       // moduli p, q, p^2, q^2
  // the 37-limb primes of a 2048-bit key once more as 40-limb moduli in four lanes of 10 (the standard constants only): ladders
  // modulo the primes of batches that leave most of the chip empty at one lane per number (plan::prime_lanes, PrimeShape)
  bool has_sliced_primes = false;
  ModCtx mp_s, mq_s;
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
  // the eight-lane pair kernel for the halves (digits of h8 = 40 limbs in four lanes of 10, Montgomery radix R_40): prime | Cadj, and the pair
  // digits of R_40^2 R_H^-1 (a number enters with one product by it), of R_H (leaves), of R_40 -- CRT ladders of batches that leave most
  // of the chip empty at two lanes per number (plan::crt_pair_lanes8)
  int pair_h8 = 0;
  DevLimbs pair8_p, pair8_q, pair8t_p, pair8t_q;
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
  // the same ladders with TWO lanes per digit (GenQ6 on 19-limb slices: digits of h6 = 38 limbs, Montgomery radix R_38 per digit) for
  // batches that leave most of the chip empty (plan::crt_triple_lanes6): kconsts for 38-limb digits, and the digit forms of
  // R_38^2 R_H^-1 (entry from the radix-R_H digit form), R_H (exit), R_38 (= 1)
  int triple_h6 = 0;
  DevLimbs tkc6_p, ttc6_p, tkc6_q, ttc6_q;
  ExpOrder eo_p, eo_q;                   // exponent reduction modulo the orders of the units modulo p^3 / q^3
  // The key holder's powers modulo n^3 through the STRUCTURE of the unit group, Z*_{n^3} = <1 + n> x (Teichmueller lifts of Z*_n)
  // (ddleq.cpp struct_pow_n3): ladders modulo the primes with exponents modulo p - 1, q - 1, then the lift
  // omega_p(t) = t (t^(p-1))^z, z = -(p - 1)^-1 in Z_p, by the binomial series modulo p^3
  bool has_lift = false;
  ExpOrder eo1_p, eo1_q;                 // exponent reduction modulo p - 1, q - 1
  BigU n2_mod_p1, n2_mod_q1;             // n^2 mod (p - 1), n^2 mod (q - 1): the shared exponent of y^(n^2) in each half
  int c_lz_p2 = -1, c_lz_q2 = -1;        // z R mod p^2 in mp2, mod q^2 in mq2 (z taken modulo prime^2)
  int c_lz2_p = -1, c_lz2_q = -1;        // z (z - 1) / 2 mod p in mp, mod q in mq, stored plain
};

namespace pgi {

// ---- keys.cpp / ctx.cpp --------------------------------------------------------------------------------------------------------
BigU order_fixup(const BigU& e, const BigU& ord);
std::vector<uint32_t> make_pair_consts(const BigU& pr, int H);
bool make_pair8_consts(const BigU& root, const BigU& mod2, int H, int h8, std::vector<uint32_t>& c8, std::vector<uint32_t>& t8);
std::vector<uint32_t> make_triple_kconsts(const BigU& n, int H);
bool ctx_alive(pgpu_ctx* c);

// ---- modexp.cpp ----------------------------------------------------------------------------------------------------------------

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
struct TriplePlan {
  uint32_t* mem;        // [slot][3H][nb]
  size_t slot_words;    // 3H * nb
  size_t nb;
  int H;
  uint32_t* slot(uint32_t i) const { return mem + (size_t)i * slot_words; }
};
// the ladders x^(e[half]) modulo p^3 | q^3 in digit form, both halves in one launch: slot 0 -> slot 3 of tp / tq (table from slot 5), on one
// lane per digit or -- small batches -- two (plan::crt_triple_lanes6)
// the three-digit kernel with four lanes per digit (GenQ12: slots of 3 x h12 limbs, radix R_h12; constants of mc.triple: 0 one, 1 entry from
// the radix-R_H digit form, 2 exit): a plan of its own, digit forms zero-extended into it / cut back out of it, and the launch
bool triple12_available(pgpu_ctx* ctx, const ModCtx& mc);
TriplePlan triple_alloc12(pgpu_ctx* ctx, const ModCtx& mc, size_t nb, int slots);
void triple_widen12(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* digits, uint32_t* digits12, size_t nb);
void triple_narrow12(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* digits12, uint32_t* digits, size_t nb);
void triple_run12(pgpu_ctx* ctx, const ModCtx& mc, const TriplePlan& t12, const Prog& p, const uint32_t* exps);
void crt_triple_ladders(const pgpu_seckey* sk, const BigU e[2], const TriplePlan& tp, const TriplePlan& tq, size_t nb, int beside = 1);

ModexpPlan modexp_alloc(pgpu_ctx* ctx, const ModCtx& mc, size_t nb, int table_slots);
void reduce_mod(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* in, int w_in, uint32_t* out, size_t nb);
void reduce_mod_wide(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* in, int w_in, uint32_t* out, size_t nb);
void unpack_mod(pgpu_ctx* ctx, const ModCtx& mc, const uint8_t* buf, size_t stride, size_t count, int mem, uint32_t* out,
                size_t nb, bool canonical = false);
void modexp_pair(pgpu_ctx* ctx, const ModCtx& mc, const ModexpPlan& pl, const BigU* e, const uint32_t* exps, int we,
                 bool use_post, int lanes, uint32_t** raw_out = nullptr);
const uint32_t* windows5_of(pgpu_ctx* ctx, const uint32_t* exps, int we, size_t nb);
const uint32_t* triple_windows(pgpu_ctx* ctx, const uint32_t* exps, int we, size_t nb, int wb);
bool triple_usable(pgpu_ctx* ctx, const ModCtx& mc, bool allow6 = false);
TriplePlan triple_alloc(pgpu_ctx* ctx, const ModCtx& mc, size_t nb, int slots);
void triple_enter(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, const TriplePlan& tp, uint32_t slot);
void triple_from_pair(pgpu_ctx* ctx, const uint32_t* pair_digits, const TriplePlan& tp, uint32_t slot);
void triple_exit(pgpu_ctx* ctx, const ModCtx& mc, const TriplePlan& tp, uint32_t slot, uint32_t* out, const uint32_t* post);
void triple_run(pgpu_ctx* ctx, const ModCtx& mc, const TriplePlan& tp, const Prog& p, const uint32_t* exps);
void modexp_triple(pgpu_ctx* ctx, const ModCtx& mc, const ModexpPlan& pl, const BigU* e, const uint32_t* exps, int we,
                   bool use_post, const uint32_t* pair_digits = nullptr);
void modexp_shared_run(pgpu_ctx* ctx, const ModCtx& mc, const ModexpPlan& pl, const BigU& e, bool wide, bool use_post,
                       bool skip_zero, uint32_t** raw_pair_out = nullptr, const uint32_t* pair_digits_in = nullptr);
void modexp_perlane_run(pgpu_ctx* ctx, const ModCtx& mc, const ModexpPlan& pl, const uint32_t* exps, int we, bool wide,
                        bool use_post);
uint32_t* tree_inverse(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count);

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
bool all_units(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count);
// ... in two halves: begin() issues the product tree and the copy of its root into pinned memory (no wait); end() -- after the stream
// that ran begin() has been synchronised -- inverts the root on the host
const uint8_t* all_units_begin(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count);
bool all_units_end(const ModCtx& mc, const uint8_t* root_be);
uint32_t* batch_inverse(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count, int32_t* d_bad = nullptr,
                        bool* any_bad = nullptr);

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

void check_batch_args(const void* a, const void* b, size_t batch);
void pair_enter(pgpu_ctx* ctx, const ModCtx& mc, uint32_t* ent, size_t nb);
uint32_t* pair_leave(pgpu_ctx* ctx, const ModCtx& mc, uint32_t* pm, uint32_t out_slot, size_t nb);
void pair_leave_and_pack(pgpu_ctx* ctx, const ModCtx& mc, uint32_t* pm, uint32_t out_slot, size_t nb, size_t count, uint8_t* dst,
                         size_t out_stride, int mem);
// x^(per-number exponent) * y^e modulo n^2 as one interleaved ladder on the pair kernels; y_ready: y already IS y'^e (a caller that
// ran that ladder beside another one): the per-number ladder and one product
uint32_t* dual_pow_pair(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, const uint32_t* exps, int we, const uint32_t* y,
                        const BigU& e, size_t nb, uint32_t** raw_out = nullptr, bool y_ready = false);
void perlane_pow(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* base, const uint32_t* exps, int we, size_t nb, uint32_t* out);
void modmul_arrays(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* a, const uint32_t* b, size_t nb, uint32_t* out);
void shared_pow(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* base, int wb, const BigU& e, size_t nb, uint32_t* out);

// ---- paillier.cpp --------------------------------------------------------------------------------------------------------------
const ModCtx& cipher_mod(const pgpu_pubkey* pk, int level);
uint32_t* pow_n_crt(const pgpu_seckey* sk, const uint32_t* base, const BigU& e, size_t nb);
bool pow_n2_crt_usable(const pgpu_seckey* sk);
uint32_t* pow_n2_crt(const pgpu_seckey* sk, const uint32_t* base, const BigU& e, size_t nb);
uint32_t* L_times_const(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint32_t* u, size_t nb, size_t count, const ModCtx& mn,
                        int c_const, const uint32_t* neg_const);
uint32_t* decrypt2_crt(const pgpu_seckey* sk, const uint32_t* c_limbs, size_t nb, size_t count, int32_t* d_status);
bool struct_pow_usable(const pgpu_seckey* sk);
void teichmueller_lift(const pgpu_seckey* sk, uint32_t* const t[2], size_t nb, int32_t* d_status, uint32_t* T, int beside = 1);   // beside: plan::crt_triple_lanes6
const pgpu_pubkey::Comb7& ensure_comb7(pgpu_pubkey* pk, int level, const BigU& base, size_t ebits);
void emit_comb7(Prog& p, int we);
constexpr uint32_t kComb7First = 4;   // first table entry of a Comb7 buffer (behind the standard constants)
void gm2_from_reduced(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint32_t* mred, size_t nb, uint32_t* post);

}  // namespace pgi
