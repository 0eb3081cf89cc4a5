// vm_emit.cpp -- programs of the big-integer VM (ladders, window tables, interleaved and shared-chain exponentiations) and the
// one place that launches a VM kernel (run_vm); staging of operands and results at the byte boundary.
#include "engine.hpp"

namespace pgi {

std::atomic<bool> g_wave_priorities{true};

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
                        uint32_t tab, uint32_t post_slot, bool skip_zero_digits, bool raw) {
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

// Per-number exponents: fixed windows of wb bits on a table of x^0 .. x^(2^wb - 1) (window counts and table sizes: plan.hpp).
// wb 4: VM_MULV (7 windows per 28-bit exponent limb).  5: VM_MULV5 (the segment's `digits` must be the repacked 25-bit words:
// windows5_of()).  7: VM_MULV7 (4 windows per limb, no repacking), number-major slots.  The wide windows pay where a product costs
// two squarings (the digit kernels): a 4 096-bit exponent takes 586 products and a 63 + 63 table at 7 bits, 820 + 30 at 5,
// 1 024 + 14 at 4.
static VmOp perlane_op(int wb, bool nm4 = false) { return wb == 4 ? (nm4 ? VM_MULVT : VM_MULV) : wb == 5 ? (nm4 ? VM_MULVT5 : VM_MULV5) : VM_MULV7; }
// Table build from x in the accumulator; the wide tables square for their even entries (a squaring is half a product on the
// digit kernels).  The 7-bit table is gathered per number, so its 128 slots are number-major (VM_STORET / VM_MULV7); the entries
// the build itself reads back (x and the ones that get squared) are kept limb-major as well, in the 64 slots after the table.
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
                         uint32_t post_slot, int raw_one, int wb, bool nm4) {
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
void emit_modexp_dual(Prog& p, int we, const BigU& e, uint32_t in1, uint32_t in2, uint32_t tmp, uint32_t out, uint32_t tab1,
                      uint32_t tab2, int raw_one, int wb, bool nm4) {
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
static int shared_window_bits(const BigU& e, int wb) { return e.bit_length() < 1500 ? 6 : dual_sliding_bits(wb); }
void emit_modexp_multi(Prog& p, const std::vector<PerNumberBase>& pn, int wb, const std::vector<SharedBase>& sh, uint32_t tmp,
                       uint32_t out, uint32_t one, bool nm4) {
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
                                uint32_t out0, uint32_t bucket0, int w, uint32_t one_const, bool muls) {
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

// launch one VM kernel with 1 to 3 segments of `nb` numbers each (same modulus shape; s2 only together with s1)
void run_vm(pgpu_ctx* ctx, size_t nb, const SegSpec& s0, const SegSpec* s1, bool profile, size_t launch_nb,
            const SegSpec* s2) {
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
    if (ctx->background_launch && ss[i]->prog->montmuls >= 256) {
      // a long ladder on a side lane whose result is not needed before the main stream's next ladders are done: priority 0 throughout, so
      // that it takes the issue slots the main stream's waves leave instead of an equal share of their SIMDs
      std::vector<uint32_t> bg(ss[i]->prog->w);
      for (size_t k = 0; k + 1 < bg.size(); k += 2) bg[k] &= 0x3FFFFFFFu;
      g.prog = ctx->upload_words(bg);
      wipe_vec(bg);
    } else {
      g.prog = ctx->upload_words(ss[i]->prog->w);
    }
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
  // lanes per number of the generic kernels: plan::generic_shape
  int WL = mc->WL, K = mc->K;
  const bool pair = s0.pair != nullptr;
  if (s1 && pair != (s1->pair != nullptr)) api_throw(PGPU_ERR_INVALID, "segment kind mismatch");
  if (s2 && pair != (s2->pair != nullptr)) api_throw(PGPU_ERR_INVALID, "segment kind mismatch");
  if (pair) {
    WL = s0.pair_lanes == 16 ? s0.pair_h / 8 : (s0.pair_lanes == 8 || s0.pair_lanes == 12) ? s0.pair_h / 4 : (s0.pair_lanes == 4 || s0.pair_lanes == 6) ? s0.pair_h / 2 : s0.pair_h;
    K = s0.pair_lanes == 16 ? 128 : s0.pair_lanes == 12 ? 160 : s0.pair_lanes == 8 ? 96 : s0.pair_lanes == 6 ? 112 : s0.pair_lanes == 4 ? 64 : s0.pair_lanes == 3 ? 48 : s0.pair_lanes == 2 ? 32 : 16;   // tags of the pair kernels, not lane counts
  } else {
    const plan::GenericShape gs = plan::generic_shape(WL, K, launch_nb, s2 ? 3 : s1 ? 2 : 1, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus),
                                                      ctx->use_w74 && ctx->use_asm);
    WL = gs.WL;
    K = gs.K;
  }
  const uint32_t blocks_per_seg = (uint32_t)(launch_nb * (pair ? (s0.pair_lanes == 3 ? 4 : s0.pair_lanes == 6 ? 8 : s0.pair_lanes == 12 ? 16 : s0.pair_lanes) : K) / VM_BLOCK);
  a.seg0_blocks = blocks_per_seg;
  a.seg1_blocks = blocks_per_seg;
  const uint32_t blocks = blocks_per_seg * (s2 ? 3 : s1 ? 2 : 1);
  const bool nm_tables = s0.prog->nm_tables || (s1 && s1->prog->nm_tables) || (s2 && s2->prog->nm_tables);
  const bool nm4 = s0.prog->nm4 || (s1 && s1->prog->nm4) || (s2 && s2->prog->nm4);
  const bool mulv7 = s0.prog->mulv7 || (s1 && s1->prog->mulv7) || (s2 && s2->prog->mulv7);
  if (nm4 && mulv7) api_throw(PGPU_ERR_UNSUPPORTED, "internal: 4-bit and 7-bit number-major tables in one launch");
  // VM_STORET / VM_MULVT / VM_MULVT5: GenP for 37-limb primes, GenQ (two lanes), GenQ4 (four lanes)
  // (the three-digit kernel: VM_MULVT5 beside its VM_MULV7; the host never sends it 4-bit windows)
  // (and the generic one-lane kernel for 37-limb moduli: the ladders modulo the primes of struct_pow_n3)
  const bool nm4_generic = !pair && ((WL == 37 && K == 1) || (WL == 10 && K == 4));     // (and the primes' four-lane twins: PrimeShape)
  const bool nm4_kernel = nm4_generic || (pair && ((s0.pair_lanes == 1 && s0.pair_h == 37) || s0.pair_lanes == 2 || s0.pair_lanes == 4 || s0.pair_lanes == 3));
  const bool use_asm = ctx->use_asm && vm_asm_available(WL, K) && s0.prog->asm_ok && (!s1 || s1->prog->asm_ok) &&
                       (!s2 || s2->prog->asm_ok) && (!nm_tables || (nm4_generic ? !mulv7 : (pair && (mulv7 ? s0.pair_lanes == 3 : nm4 ? nm4_kernel : (s0.pair_lanes == 3 || nm4_kernel))))) &&
                       plan::gather_fits(nb, (s0.pair_lanes == 3 || s0.pair_lanes == 6 || s0.pair_lanes == 12) ? 3 * s0.pair_h : mc->WT,
                                         (int)std::max(s0.prog->gather_slots, std::max(s1 ? s1->prog->gather_slots : 1u, s2 ? s2->prog->gather_slots : 1u)));
  pgpu_ctx::Ev* ev = nullptr;
  if (profile) {
    ev = &ctx->next_ev();
    // v_mad_u64_u32 lane-ops executed: 2 WT^2 per product; the assembly kernel's K == 1 squaring rows use the
    // symmetry of the square: WL^2 (reduction) + WL(WL-1)/2 + WL (product)
    const double full = 2.0 * mc->WT * mc->WT;
    // squaring rows: K == 1 triangular (WT^2 + WT(WT-1)/2 + WT); K == 2 slice-level symmetry (product part 1.5 WL^2 per lane)
    double sq = full;
    double mulp = full;
    if (pair && (s0.pair_lanes == 3 || s0.pair_lanes == 6 || s0.pair_lanes == 12)) {   // GenQ3 / GenQ6 / GenQ12: a squaring is one pass, a product 6 blocks (its second pass runs half the lanes)
      const double H = s0.pair_h;
      mulp = 12.0 * H * H;
      sq = 8.0 * H * H;
    } else if (pair && (s0.pair_lanes == 4 || s0.pair_lanes == 8 || s0.pair_lanes == 16)) {   // GenQ4 / GenQ8: a squaring is one pass of H rows of 2 * H/2 multiplies in four lanes; a
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
    if (ss[i] && ss[i]->prog->needs_muls && !(pair && (s0.pair_lanes == 4 || s0.pair_lanes == 8 || s0.pair_lanes == 16)))
      api_throw(PGPU_ERR_UNSUPPORTED, "internal: VM_MULS on a kernel that does not implement it");
  if (pair && s0.pair_lanes == 1 && s0.pair_h > 37)
    for (int i = 0; i < 3; ++i)
      if (ss[i] && ss[i]->prog->wide_gathers)   // (an opcode a kernel does not know ends its program: refuse, never compute garbage)
        api_throw(PGPU_ERR_UNSUPPORTED, "internal: the one-lane pair kernel for 55-limb primes has 4-bit per-number windows only");
  if (ev) snprintf(ev->name, sizeof ev->name, use_asm ? "vm_asm_%d_%d" : "vm_kernel<%d,%d>", WL, K);
  // placement by LDS size (plan::lds_share): a CU per workgroup for the small launches of a call whose launches run beside each other, at
  // most one workgroup per CU for a main-stream ladder that fits the CUs its stream may use
  const bool on_side = ctx->stream == ctx->side || ctx->stream == ctx->side_l[0] || ctx->stream == ctx->side_l[1] || ctx->stream == ctx->side_l[2];
  const int lds_share = !use_asm ? 0 : ctx->lds_force >= 0 ? ctx->lds_force : plan::lds_share(blocks, ctx->stream_cus, on_side, ctx->exclusive_call, montmuls, ctx->use_exclusive, ctx->use_spread, ctx->use_exclusive_short);
  hipError_t e = use_asm ? launch_vm_asm(WL, K, a, blocks, ctx->stream, lds_share) : launch_vm(WL, K, a, blocks, ctx->stream);
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

}  // namespace pgi
