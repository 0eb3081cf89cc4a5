// ddleq.cpp -- NestedRandomize (operations.go:96-118), the DDLEQ proofs (ddleq.go:27-153) and RandomOracleDigest
// (random_oracle.go:10-32): interleaved ladders modulo n^3, the key holder's halves modulo p^3 / q^3, the prover.
#include <chrono>
#include <functional>
#include "engine.hpp"

extern "C" {

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

namespace pgi {
// x^(per-number exponent, W2 limbs) * y^(n^2) mod n^3 as ONE interleaved ladder (emit_modexp_dual): the verifier's
// check^(E^n) * F^(n^2) (ddleq.go:143-152) and NestedRandomize's ct^(a^n) * b^(n^2) (operations.go:108-114).
// x, y: W3-limb arrays (any value below R); returns the canonical result (W3 limbs, stride nb).
// (does a batch of nb numbers take the two-ladder form below?  Its callers then run y^n modulo n^2 beside their own first ladder.)
static bool dual_n3_two(pgpu_ctx* ctx, const pgpu_pubkey* pk, size_t nb) {
  const ModCtx &mn2 = pk->mn2, &mn3 = *pk->mn3;
  if (!(triple_usable(ctx, mn3) && plan::triple_batch_fits(nb, mn3.WT) && ctx->use_lift)) return false;
  const int H3 = mn3.triple.root->WT;
  const PairInfo& pi = mn2.pairn;
  const bool six = ctx->use_lanes8 && H3 % 2 == 0 && vm_asm_available(H3 / 2, 112) &&
                   plan::triple_two_lanes_per_digit(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus));
  return six && pi.root && pi.c_one_pair >= 0 && ctx->use_asm && ctx->use_pair && ctx->use_side &&
         plan::dual_n3_two_ladders(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus));
}
// y^n modulo n^2 (y: W3 limbs, any value below R) on a side lane, beside the caller's ladder that produces the per-number exponents;
// join() before dual_pow_n3 uses it
struct YPower {
  SideStream lane;
  uint32_t* yn = nullptr;
  YPower(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint32_t* y, size_t nb, bool wanted) : lane(ctx, 2) {
    if (!wanted || !lane.on) return;
    const ModCtx &mn2 = pk->mn2, &mn3 = *pk->mn3;
    hipEvent_t ready = lane.mark();
    lane.enter(ready);
    uint32_t* y2 = ctx->ws_t<uint32_t>((size_t)mn2.WT * nb);
    reduce_mod(ctx, mn2, y, mn3.WT, y2, nb);
    yn = ctx->ws_t<uint32_t>((size_t)mn2.WT * nb);
    shared_pow(ctx, mn2, y2, mn2.WT, pk->N, nb, yn);
    lane.leave();
  }
  const uint32_t* get() { lane.join(); return yn; }
};
struct ExclusiveScope {
  pgpu_ctx* c; bool was;
  ExclusiveScope(pgpu_ctx* c_, bool on) : c(c_), was(c_->exclusive_call) { if (on) c->exclusive_call = true; }
  ~ExclusiveScope() { c->exclusive_call = was; }
};

uint32_t* dual_pow_n3(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint32_t* x, const uint32_t* exps, const uint32_t* y, size_t nb,
                      const uint32_t* yn2 = nullptr) {
  // yn2 (optional): y^n modulo n^2, canonical -- the two-ladder form then runs x^(e1) alone and multiplies
  const ModCtx &mn2 = pk->mn2, &mn3 = *pk->mn3;
  const int W2 = mn2.WT, W3 = mn3.WT;
  const bool use3 = triple_usable(ctx, mn3) && plan::triple_batch_fits(nb, W3);
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
    // (a batch so small that one ladder's latency is the run time takes two lanes per digit -- GenQ6, limb-major 5-bit tables:
    // 2 048 squarings modulo n^3 in ~25 ms where the one-lane digits take 77 whatever the batch)
    const int H3 = mn3.triple.root->WT;
    const bool six = ctx->use_lanes8 && H3 % 2 == 0 && vm_asm_available(H3 / 2, 112) &&
                     plan::triple_two_lanes_per_digit(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus));
    const int wb = six ? 5 : triple_window_bits(nb, H3);
    const bool nm5 = !six && wb == 5;
    const uint32_t tab2 = 5 + (uint32_t)perlane_table_slots(wb, nm5);
    TriplePlan tp = triple_alloc(ctx, mn3, nb, (int)tab2 + (1 << (dual_sliding_bits(wb) - 1)));
    const bool hand = ctx->use_handover && mn2.pairn.root && mn2.pairn.root->WT == mn3.triple.root->WT;
    // A batch so small that two eight-lane ladders still find a SIMD per wave side by side: x^(e0) does not depend on W, so it runs as a
    // ladder of its own on a side lane -- beside the W ladder modulo n^2 and W^n modulo n^3 -- and one product joins them.  The chain
    // E^n -> W -> W^n is then the run time (2 048 numbers: the interleaved ladder 45 ms, W^n alone 34); launches on CUs of their own.
    const bool two_ladders = six && plan::dual_n3_two_ladders(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus));
    if (two_ladders) {
      ExclusiveScope excl(ctx, true);
      TriplePlan tx = triple_alloc(ctx, mn3, nb, 5 + perlane_table_slots(5, false));
      TriplePlan tw = triple_alloc(ctx, mn3, nb, 5 + 32);
      // The smallest batches: FOUR lanes per digit (GenQ12, a DPP row per number; digits of h12 = 76 limbs in radix R_76).  The ladders run
      // in slots of their own: the digit forms are zero-extended, change radix with the first product of their programs, and the product
      // of the two results comes back to radix R_H with its last.
      const bool twelve = triple12_available(ctx, mn3) &&
                          plan::triple_four_lanes_per_digit(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), true, true, 2);
      TriplePlan tx12{}, tw12{};
      if (twelve) { tx12 = triple_alloc12(ctx, mn3, nb, 5 + perlane_table_slots(5, false)); tw12 = triple_alloc12(ctx, mn3, nb, 5 + 32); }
      auto widen = [&](const uint32_t* d74, uint32_t* d76) { triple_widen12(ctx, mn3, d74, d76, nb); };
      auto run12 = [&](const TriplePlan& t, const Prog& p, const uint32_t* exps) { triple_run12(ctx, mn3, t, p, exps); };
      Fork ft(ctx, 3);
      ft.chain(2);
      triple_enter(ctx, mn3, pc.in(), tx, 0);
      {
        Prog px;
        if (twelve) {
          widen(tx.slot(0), tx12.slot(0));
          px.op(VM_LOAD, 0); px.op(VM_MULC, 1); px.op(VM_STORE, 0);                     // radix R_H -> R_h12
        }
        emit_modexp_perlane(px, W1, 0, NO_SLOT, 2, 3, 5, NO_SLOT, 0, 5, false);
        px.end();
        if (twelve) run12(tx12, px, triple_windows(ctx, e0, W1, nb, 5));
        else triple_run(ctx, mn3, tx, px, triple_windows(ctx, e0, W1, nb, 5));
      }
      ft.chain(0);
      uint32_t* raw = nullptr;
      uint32_t* wv = dual_pow_pair(ctx, mn2, x2, e1, W1, yn2 ? yn2 : y2, pk->N, nb, hand ? &raw : nullptr, yn2 != nullptr);
      if (wv) {
        if (raw) {
          triple_from_pair(ctx, raw, tw, 0);
        } else {
          launch_copy_limbs(wv, 0, W2, pc.in() + pc.slot_words, W3, nb, ctx->stream);    // slot 1 <- W, zero-extended
          triple_enter(ctx, mn3, pc.in() + pc.slot_words, tw, 0);
        }
        Prog pw;
        if (twelve) {
          widen(tw.slot(0), tw12.slot(0));
          pw.op(VM_LOAD, 0); pw.op(VM_MULC, 1); pw.op(VM_STORE, 0);
        }
        emit_modexp_shared(pw, pk->N, 0, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
        pw.end();
        if (twelve) run12(tw12, pw, nullptr);
        else triple_run(ctx, mn3, tw, pw, nullptr);
        ft.join();
        Prog pm;
        if (twelve) {
          HIPCHK(hipMemcpyAsync(tw12.slot(1), tx12.slot(3), tw12.slot_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
          pm.op(VM_LOAD, 3); pm.op(VM_MUL, 1); pm.op(VM_MULC, 2); pm.op(VM_STORE, 3); pm.end();       // ... and back to radix R_H
          run12(tw12, pm, nullptr);
          triple_narrow12(ctx, mn3, tw12.slot(3), tw.slot(3), nb);
        } else {
          HIPCHK(hipMemcpyAsync(tw.slot(1), tx.slot(3), tw.slot_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
          pm.op(VM_LOAD, 3); pm.op(VM_MUL, 1); pm.op(VM_STORE, 3); pm.end();
          triple_run(ctx, mn3, tw, pm, nullptr);
        }
        triple_exit(ctx, mn3, tw, 3, pc.out(), nullptr);
        return pc.out();
      }
      ft.join();                                   // (no pair kernel for this key: the interleaved ladder below)
    }
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
      emit_modexp_dual(pd, W1, pk->N, 0, 1, 2, 3, 5, tab2, 0, wb, nm5);
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
}  // namespace pgi

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
    // (a batch so small that the ladders are latency: b^n modulo n^2 runs beside a^n, on CUs of its own -- dual_pow_n3's two-ladder form)
    const bool two = dual_n3_two(ctx, pk, nb);
    ExclusiveScope excl(ctx, two);
    uint32_t* x = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    uint32_t* y = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    unpack_mod(ctx, mn3, ct, ct_stride, batch, mem, x, nb);
    unpack_mod(ctx, mn3, b, ab_stride, batch, mem, y, nb);
    YPower bn(ctx, pk, y, nb, two);
    ModexpPlan pa_ = modexp_alloc(ctx, mn2, nb, 32);
    unpack_mod(ctx, mn2, a, ab_stride, batch, mem, pa_.in(), nb);
    modexp_shared_run(ctx, mn2, pa_, pk->N, false, false, true);
    uint32_t* res = dual_pow_n3(ctx, pk, x, pa_.out(), y, nb, bn.get());
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
    // (a batch so small that the ladders are latency: F^n modulo n^2 runs beside E^n, on CUs of its own -- dual_pow_n3's two-ladder form)
    if (f_stride * 8 > (size_t)LB * W3 + 7) api_throw(PGPU_ERR_INVALID, "F wider than n^3");
    const bool two = dual_n3_two(ctx, pk, nb);
    ExclusiveScope excl(ctx, two);
    uint32_t* fl = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    unpack_operand(ctx, f, f_stride, f_stride, batch, mem, fl, W3, nb);
    YPower fn(ctx, pk, fl, nb, two);
    ModexpPlan pe = modexp_alloc(ctx, mn2, nb, 32);
    // E is a residue modulo n^2 in every honest proof (ddleq.go:94-99); a wider field (up to twice the width) is reduced
    // by the ordinary kernel's two-chunk entry, the usual width takes the pair-kernel path
    const bool ewide = e_stride * 8 > (size_t)LB * W2;
    unpack_operand(ctx, e, e_stride, std::min(e_stride, 2 * mn2.nbytes), batch, mem, pe.in(), ewide ? 2 * W2 : W2, nb);
    modexp_shared_run(ctx, mn2, pe, pk->N, ewide, false, true);
    // check = chalBit ? ct2 : ct1 ; check^en * F^(n^2) mod n^3 == alpha           (ddleq.go:138-152)
    // one interleaved ladder: the squarings of check^en and of F^(n^2) are shared (emit_modexp_dual)
    uint32_t* chk = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    launch_select(chal, c2, c1, chk, W3, nb, ctx->stream);
    uint32_t* got = dual_pow_n3(ctx, pk, chk, pe.out(), fl, nb, fn.get());
    int32_t* d_ok = ctx->ws_t<int32_t>(nb);
    if (wa != W3) api_throw(PGPU_ERR_INVALID, "alpha stride must be the byte length of n^3");
    launch_equal(got, al, W3, nb, batch, d_ok, ctx->stream);
    HIPCHK(hipMemcpyAsync(ok, d_ok, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

}  // extern "C"

namespace pgi {

uint32_t* zext(pgpu_ctx* ctx, const uint32_t* in, int w, int wo, size_t nb);
uint32_t* concat2(pgpu_ctx* ctx, const uint32_t* a, const uint32_t* b, int w, size_t nb);
void split2(pgpu_ctx* ctx, const uint32_t* in, int which, int w, size_t nb, uint32_t* out);

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
  // (two lanes per number pay while they still leave every wave a SIMD of its own: above half a wave per SIMD at one lane, two
  // lanes are two waves on most SIMDs -- 1.15 x the one-lane ladder -- and the one-lane kernel needs fewer multiplies)
  const int lanes = plan::crt_pair_lanes(1, sk->pair_small2, nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus));
  const int H = sk->mp.WT, W2 = sk->mp2.WT;
  const size_t S1 = (size_t)H * nb, S2 = (size_t)W2 * nb;
  const int wb = 4;                                           // per-number windows: VM_MULV
  if (!plan::pair_nm4_fits(nb, W2)) return false;
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

// Garner for residues modulo p^3 and q^3 (canonical, mp3.WT limbs, stride nb): out = x_p + p^3 ((x_q - x_p) p^-3 mod q^3), WT(n^3) limbs
void garner_n3(const pgpu_seckey* sk, const uint32_t* xp, const uint32_t* xq, size_t nb, uint32_t* out) {
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx& mq3 = sk->mq3;
  const int W = mq3.WT, W3 = sk->pk->mn3->WT;
  const size_t S = (size_t)W * nb;
  uint32_t* g = ctx->ws_t<uint32_t>(S * 5);       // slots: 0 x_p, 1 x_q, 2 A, 3 B, 4 h
  HIPCHK(hipMemcpyAsync(g, xp, S * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(g + S, xq, S * 4, hipMemcpyDeviceToDevice, ctx->stream));
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
}

// ---- powers modulo n^3 through the STRUCTURE of the unit group (round 4) ---------------------------------------------------------
// Z*_{n^3} = <1 + n> x T, T = the Teichmueller lifts omega(u) = u^(n^2) of Z*_n, and the key holder can read both coordinates of
// ct = (1 + n)^m rho^(n^2): m is the level-two plaintext (one CRT decryption PER STATEMENT), and the T-part of any product is
// determined by the product modulo n.  So for X = ct^e * y^(n^2) -- the prover's sanity value ct1^(a^n) b^(n^2) (ddleq.go:62-69)
// and Alpha = ct1^(x^n) y^(n^2) (:81-87) --
//     X = (1 + n)^(m e mod n^2) * omega(X mod n),        X mod p = (ct mod p)^(e mod (p-1)) * (y mod p)^(n^2 mod (p-1)),
// and modulo p^3 the lift is omega_p(t) = t * (t^(p-1))^z with z = -(p-1)^-1 in Z_p: t^(p-1) = 1 + p a, and
// (1 + p a)^z = 1 + z p a + C(z, 2) p^2 a^2 (mod p^3) is three small products.  Per number: one product modulo n^2 and the closed
// form of (1 + n)^k, ONE interleaved ladder of 1 036 squarings modulo p (and q: 37-limb numbers), ONE ladder of 1 023 squarings
// modulo p^3 (and q^3) for t^(p-1), Garner -- 16 M multiply-adds per number and half where the p-adic split of pow_n3_crt needs
// 2 047 squarings modulo p^2 plus 1 024 modulo p^3: 26 M.  The same integers for every unit ct, y; a lane whose ct or y is not a
// unit is flagged (the decryption's exact division per statement, the lift's zero test per number) and the caller falls back to the
// literal ladders.

// omega(t) = t^(n^2) mod n^3 from t modulo the primes: t[half] = canonical residues modulo p / q (mp.WT limbs, stride nb).  Modulo
// pr^3 the lift is t (t^(pr-1))^z, z = -(pr - 1)^-1 in Z_pr: w = t^(pr-1) on the digit kernel (the ladder of the level-two
// decryption, both halves in one launch), a = (w - 1) / pr (exact for every unit t; a t that is 0 modulo the prime is flagged in d_status),
// (1 + pr a)^z = 1 + pr (z a mod pr^2) + pr^2 (C(z,2) a^2 mod pr); then Garner.  T: WT(n^3) limbs, stride nb.
void teichmueller_lift(const pgpu_seckey* sk, uint32_t* const t[2], size_t nb, int32_t* d_status, uint32_t* T, int beside) {
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx &mp3 = sk->mp3, &mq3 = sk->mq3;
  const int H = sk->mp.WT, P2 = sk->mp2.WT, W = mp3.WT;
  const size_t S1 = (size_t)H * nb, S2 = (size_t)P2 * nb, S = (size_t)W * nb;
  TriplePlan tp = triple_alloc(ctx, mp3, nb, 5 + 32), tq = triple_alloc(ctx, mq3, nb, 5 + 32);
  uint32_t* tz[2];
  Fork f2(ctx);
  for (int half = 0; half < 2; ++half) {
    f2.chain(half);
    {                                                    // a t that is a multiple of the prime has no lift: flag the lane
      int32_t* zf = ctx->ws_t<int32_t>(nb);
      launch_is_zero(t[half], H, nb, zf, ctx->stream);
      launch_or_flags(zf, nb, d_status, PGPU_LANE_NONUNIT, ctx->stream);
    }
    tz[half] = zext(ctx, t[half], H, W, nb);
    triple_enter(ctx, half ? mq3 : mp3, tz[half], half ? tq : tp, 0);
  }
  f2.join();
  {
    const BigU es[2] = {sk->p - BigU(1), sk->q - BigU(1)};
    crt_triple_ladders(sk, es, tp, tq, nb, beside);        // (small batches: two lanes per digit)
  }
  // a = (w - 1) / pr (exact for every unit t), c = (1 + pr a)^z = 1 + pr (z a mod pr^2) + pr^2 (C(z,2) a^2 mod pr), tau = t c
  uint32_t* tau[2];
  Fork f3(ctx);
  for (int half = 0; half < 2; ++half) {
    f3.chain(half);
    const ModCtx &m1 = half ? sk->mq : sk->mp, &m2 = half ? sk->mq2 : sk->mp2, &m3 = half ? mq3 : mp3;
    uint32_t* w = ctx->ws_t<uint32_t>(S);
    triple_exit(ctx, m3, half ? tq : tp, 3, w, nullptr);
    uint32_t* v2 = ctx->ws_t<uint32_t>(S2 * 2);                          // slots (P2 limbs): 0 a, 1 z a
    uint32_t* tb = ctx->ws_t<uint32_t>(S);
    // (t^(pr-1) = 1 modulo pr for every unit t: the division is exact and needs no check -- the non-units are the t = 0, flagged above)
    launch_div_exact(w, W, 1, nullptr, 0, tb, (half ? sk->qinv2k_2 : sk->pinv2k_2).d, m1.d_nmod, H, v2, P2, nb, nb, nullptr, 0,
                     ctx->stream);
    {
      Prog a;
      a.op(VM_LOAD, 0); a.op(VM_MULC, (uint32_t)(half ? sk->c_lz_q2 : sk->c_lz_p2)); a.op(VM_STORE, 1); a.end();
      SegSpec sa{&m2, &a, v2, nullptr};
      run_vm(ctx, nb, sa, nullptr, false);
      launch_canon(v2 + S2, m2.d_nmod, P2, nb, ctx->stream);
    }
    uint32_t* v1 = ctx->ws_t<uint32_t>(S1 * 2);                          // slots (H limbs): 0 a mod pr, 1 C(z,2) a^2
    reduce_mod(ctx, m1, v2, P2, v1, nb);
    {
      Prog a;
      a.op(VM_LOAD, 0); a.op(VM_MULC, C_R2); a.op(VM_SQR); a.op(VM_MULC, (uint32_t)(half ? sk->c_lz2_q : sk->c_lz2_p)); a.op(VM_STORE, 1);
      a.end();
      SegSpec sa{&m1, &a, v1, nullptr};
      run_vm(ctx, nb, sa, nullptr, false);
      launch_canon(v1 + S1, m1.d_nmod, H, nb, ctx->stream);
    }
    uint32_t* c1 = ctx->ws_t<uint32_t>(S);
    uint32_t* cf = ctx->ws_t<uint32_t>(S);
    launch_mul_const_add(v2 + S2, P2, (half ? sk->q_limbs : sk->p_limbs).d, H, nullptr, 0, 1, c1, W, nb, ctx->stream);          // 1 + pr (z a)
    launch_mul_const_add(v1 + S1, H, (half ? sk->q2_limbs : sk->p2_limbs).d, P2, c1, W, 0, cf, W, nb, ctx->stream);          // + pr^2 (...)
    launch_canon(cf, m3.d_nmod, W, nb, ctx->stream);
    tau[half] = ctx->ws_t<uint32_t>(S);
    modmul_arrays(ctx, m3, tz[half], cf, nb, tau[half]);
  }
  f3.join();
  garner_n3(sk, tau[0], tau[1], nb, T);
}

struct StructBase {
  const uint32_t* m = nullptr;            // level-two plaintext of ct per statement (mn2.WT limbs, stride nbs)
  const uint32_t* cpr[2] = {nullptr, nullptr};   // ct mod p, ct mod q per statement (mp.WT limbs, stride nbs)
  size_t nbs = 0;
};
bool struct_pow_usable(const pgpu_seckey* sk) {
  pgpu_ctx* ctx = sk->ctx;
  return ctx->use_struct && sk->has_lift && sk->has_crt2 && sk->c_p3invR >= 0 && triple_usable(ctx, sk->mp3) &&
         triple_usable(ctx, sk->mq3) && sk->pk->g_is_n_plus_1 && sk->mp.K == 1 && sk->mq.K == 1 && sk->mp.WT == sk->mq.WT &&
         (sk->p - BigU(1)).bit_length() >= 64 && (sk->q - BigU(1)).bit_length() >= 64 && 2 * sk->mp3.WT >= sk->pk->mn3->WT;
}
// per statement: m = Decrypt_2(ct) through p^3, q^3 and ct modulo the primes.  ct: WT(n^3) limbs, stride nbs; d_status (nbs
// entries, zeroed by the caller) receives PGPU_LANE_NONUNIT for the first `count` statements
// (two steps: the residues feed the ladders modulo the primes, the plaintext only the closed form at the very end -- a caller
// with streams to spare lets the second run beside those ladders)
void struct_base_plaintext(const pgpu_seckey* sk, const uint32_t* ct, size_t nbs, size_t count, int32_t* d_status, StructBase& sb) {
  pgpu_ctx* ctx = sk->ctx;
  sb.nbs = nbs;
  sb.m = decrypt2_crt(sk, zext(ctx, ct, sk->pk->mn3->WT, 2 * sk->mp3.WT, nbs), nbs, count, d_status);
}
void struct_base_residues(const pgpu_seckey* sk, const uint32_t* ct, size_t nbs, StructBase& sb) {
  pgpu_ctx* ctx = sk->ctx;
  const int W3 = sk->pk->mn3->WT, W = sk->mp3.WT, H = sk->mp.WT;
  sb.nbs = nbs;
  for (int half = 0; half < 2; ++half) {
    uint32_t* r3 = ctx->ws_t<uint32_t>((size_t)W * nbs);
    reduce_mod(ctx, half ? sk->mq3 : sk->mp3, ct, W3, r3, nbs);
    uint32_t* r1 = ctx->ws_t<uint32_t>((size_t)H * nbs);
    reduce_mod_wide(ctx, half ? sk->mq : sk->mp, r3, W, r1, nbs);
    sb.cpr[half] = r1;
  }
}
// X[g] = ct[sti[g]]^(e[g]) * y[g]^(n^2) mod n^3 for g < nb (sti == nullptr: number g belongs to statement g).  e: mn2.WT limbs,
// y: mn.WT limbs, out: WT(n^3) limbs, all stride nb.  d_status (nb entries, zeroed by the caller) is flagged where the lift met a
// non-unit.
void struct_pow_n3(const pgpu_seckey* sk, const StructBase& sb, const uint32_t* sti, const uint32_t* e, const uint32_t* y, size_t nb,
                   uint32_t* out, int32_t* d_status, hipEvent_t plaintext_ready = nullptr) {
  // plaintext_ready: sb.m is still being computed on another stream of the context; only the lane of the closed form waits for it
  pgpu_ctx* ctx = sk->ctx;
  const pgpu_pubkey* pk = sk->pk;
  const ModCtx &mn = pk->mn, &mn2 = pk->mn2, &mn3 = *pk->mn3, &mp3 = sk->mp3, &mq3 = sk->mq3;
  const int W1 = mn.WT, W2 = mn2.WT, W3 = mn3.WT, H = sk->mp.WT, P2 = sk->mp2.WT, W = mp3.WT;
  const size_t S1 = (size_t)H * nb, S2 = (size_t)P2 * nb, S = (size_t)W * nb;
  auto per_number = [&](const uint32_t* in, int w) -> const uint32_t* {
    if (!sti) return in;
    uint32_t* o = ctx->ws_t<uint32_t>((size_t)w * nb);
    launch_gather(in, sb.nbs, sti, nb, o, nb, w, ctx->stream);
    return o;
  };
  hipEvent_t ladder_inputs = nullptr;                      // "e and y are there": all the closed form needs of this stream
  if (ctx->use_side) {
    ladder_inputs = ctx->next_sync_ev();
    HIPCHK(hipEventRecord(ladder_inputs, ctx->stream));
  }
  // (1) X modulo the primes: interleaved ladders with the exponents modulo p - 1, q - 1, both halves in one launch
  // slots (H limbs): 0 ct mod pr, 1 y mod pr, 2 tmp, 3 out, 5 .. the windows of e (number-major where the kernel has VM_MULVT: a
  // limb-major gather reads one dword per 32-byte sector), then the 32 odd powers of y
  // (a batch that leaves most SIMDs empty at one lane per number: the primes' four-lane twins, slots of Hs limbs -- PrimeShape; inside a
  // prover call the decryption of ct1 is still running beside this ladder)
  const PrimeShape ps = prime_shape(sk, nb, plaintext_ready ? 2 : 1);
  const size_t Ss = (size_t)ps.Hs * nb;
  const bool nm4 = ctx->use_nm4 && ctx->use_asm && H == 37 && plan::pair_nm4_fits(nb, ps.Hs);
  uint32_t* mem1[2];
  const uint32_t* ex[2];
  Prog lad[2];
  Fork f1(ctx);
  for (int half = 0; half < 2; ++half) {
    f1.chain(half);
    const ExpOrder& eo = half ? sk->eo1_q : sk->eo1_p;
    mem1[half] = ctx->ws_t<uint32_t>(Ss * 54);
    uint32_t* em = ctx->ws_t<uint32_t>((size_t)eo.modd.WT * nb);
    reduce_mod_wide(ctx, eo.modd, e, W2, em, nb);
    uint32_t* er = ctx->ws_t<uint32_t>((size_t)eo.w * nb);
    launch_exp_order_lift(e, W2, em, eo.modd.WT, eo.m_limbs.d, eo.t, eo.minv, er, eo.w, nb, ctx->stream);
    ex[half] = er;
    prime_slot_fill(ctx, ps, mem1[half], per_number(sb.cpr[half], H), nb);
    reduce_mod(ctx, *ps.m[half], y, W1, mem1[half] + Ss, nb);
    emit_modexp_dual(lad[half], eo.w, half ? sk->n2_mod_q1 : sk->n2_mod_p1, 0, 1, 2, 3, 5, 5 + (uint32_t)perlane_table_slots(4, nm4), -1, 4, nm4);
    lad[half].end();
  }
  f1.join();
  {
    SegSpec sp{ps.m[0], &lad[0], mem1[0], ex[0]}, sq{ps.m[1], &lad[1], mem1[1], ex[1]};
    run_vm(ctx, nb, sp, &sq, true);
  }
  // (2) the lift of X mod n
  uint32_t* tt[2];
  for (int half = 0; half < 2; ++half) {
    tt[half] = mem1[half] + 3 * Ss;
    launch_canon(tt[half], (half ? sk->mq : sk->mp).d_nmod, H, nb, ctx->stream);
  }
  uint32_t* T = ctx->ws_t<uint32_t>((size_t)W3 * nb);
  // (ct1's decryption -- a CU per workgroup on half the chip -- may still be running: the lift then counts it as a launch beside itself)
  teichmueller_lift(sk, tt, nb, d_status, T, plaintext_ready ? 2 : 1);
  // (3) the <1 + n> coordinate: k = m e mod n^2, G = (1 + n)^k = 1 + k n + C(k, 2) n^2 -- a chain of small kernels on a lane of its
  // own that waits for the plaintext.  ISSUED here, behind everything the ladder and the lift fork to their lanes: the runtime maps a
  // process's streams onto four hardware queues (this context has five streams), and a chain that waits for the decryption of ct1
  // in front of a chain the ladder needs -- in the same queue, by whatever order the streams were created in -- held the ladder back
  // until the decryption was over (X modulo the primes began 24 ms after a^n | x^n had ended; kernel trace, round 5).  It RUNS as soon
  // as the plaintext is there, beside the lift.
  uint32_t* G = ctx->ws_t<uint32_t>((size_t)W3 * nb);
  SideStream g_lane(ctx, 2);
  g_lane.enter(ladder_inputs);
  if (plaintext_ready) HIPCHK(hipStreamWaitEvent(ctx->stream, plaintext_ready, 0));
  {
    uint32_t* kk = ctx->ws_t<uint32_t>((size_t)W2 * nb);
    modmul_arrays(ctx, mn2, per_number(sb.m, W2), e, nb, kk);
    gm2_from_reduced(ctx, pk, kk, nb, G);
  }
  g_lane.leave();
  g_lane.join();
  // (4) X = (1 + n)^k * omega(X mod n)
  modmul_arrays(ctx, mn3, G, T, nb, out);
}

// ---- the prover's response for challenge bit 1 through the same structure (ProveDDLEQ with secpar > 1) ---------------------------
// c = ((s^an b)^en)^-1 s^xn = s^(E1) b^(E2) modulo n^3 for units (ddleq.go:107-112), E1 = xn - an en, E2 = -en, and both bases
// belong to the STATEMENT: s = ExtractRandonness(ct1), b.  Write the integers s, b as (1 + n)^(mu) omega(.) -- mu = their level-two
// "plaintext", one CRT decryption each per statement --; then
//     c = (1 + n)^(mu_s E1 + mu_b E2 mod n^2) * omega(s^E1 b^E2 mod n):
// per instance two products modulo n^2 and a closed form, ONE ladder modulo p (and q) with two per-number exponents modulo p - 1,
// and ONE lift -- where pow_n3_crt_two squares 3 071 times modulo p^3 (or 2 047 times modulo p^2 and 1 024 modulo p^3) on two
// 128-entry tables.  Worth it from a few instances per statement (plan::response_by_structure).
struct RespBase {
  const uint32_t* mu_b = nullptr;                                  // mn2.WT limbs, stride nbs
  const uint32_t *sp[2] = {nullptr, nullptr}, *bp[2] = {nullptr, nullptr};   // s, b modulo p and q: mp.WT limbs, stride nbs
  size_t nbs = 0;
};
// s, b: canonical residues modulo n (mn.WT limbs, stride nbs); d_status (nbs entries, zeroed) flags non-units among the first `count`.
// Only b's plaintext is taken (one CRT decryption per statement): E1 = xn - an en is a multiple of n^2 AS AN INTEGER -- en is made
// as xn (an)^-1 mod n^2, so an en = xn + j n^2 -- and (1 + n)^(mu_s E1) = 1 whatever mu_s is: s enters through its residues modulo
// the primes alone.  (Round 4 decrypted s as well and multiplied its plaintext by E1 mod n^2 = 0.)
// (two steps, as struct_base_*: the residues feed the ladder modulo the primes, the plaintext only the closed form at the very end -- a
// caller with a lane to spare lets the decryption run beside that ladder)
void resp_base_residues(const pgpu_seckey* sk, const uint32_t* s, const uint32_t* b, size_t nbs, RespBase& rb) {
  pgpu_ctx* ctx = sk->ctx;
  const int W1 = sk->pk->mn.WT, H = sk->mp.WT;
  rb.nbs = nbs;
  for (int half = 0; half < 2; ++half) {
    const ModCtx& m1 = half ? sk->mq : sk->mp;
    uint32_t *x = ctx->ws_t<uint32_t>((size_t)H * nbs), *y = ctx->ws_t<uint32_t>((size_t)H * nbs);
    reduce_mod_wide(ctx, m1, s, W1, x, nbs);
    reduce_mod_wide(ctx, m1, b, W1, y, nbs);
    rb.sp[half] = x;
    rb.bp[half] = y;
  }
}
void resp_base_plaintext(const pgpu_seckey* sk, const uint32_t* b, size_t nbs, size_t count, int32_t* d_status, RespBase& rb) {
  pgpu_ctx* ctx = sk->ctx;
  const int W1 = sk->pk->mn.WT, W = sk->mp3.WT;
  rb.nbs = nbs;
  int32_t* st = ctx->ws_t<int32_t>(nbs);
  HIPCHK(hipMemsetAsync(st, 0, nbs * 4, ctx->stream));
  rb.mu_b = decrypt2_crt(sk, zext(ctx, b, W1, 2 * W, nbs), nbs, nbs, st);
  launch_or_flags(st, count, d_status, PGPU_LANE_NONUNIT, ctx->stream);
}
// ct's and b's level-two plaintexts per statement in ONE decryption of 2 nbs numbers (a prover call that will take the response through
// the structure whatever the challenge bits turn out to be: both are inputs of the call, and a second latency-bound launch of the
// same kernel later -- behind s and the inversions on the side stream -- was the end of the critical path at 4 096 instances)
void struct_and_resp_plaintexts(const pgpu_seckey* sk, const uint32_t* ct, const uint32_t* b, size_t nbs, size_t count, int32_t* d_status_ct,
                                int32_t* d_status_b, StructBase& sb, RespBase& rb) {
  pgpu_ctx* ctx = sk->ctx;
  const int W1 = sk->pk->mn.WT, W2 = sk->pk->mn2.WT, W3 = sk->pk->mn3->WT, W = sk->mp3.WT;
  sb.nbs = rb.nbs = nbs;
  uint32_t* both = concat2(ctx, zext(ctx, ct, W3, 2 * W, nbs), zext(ctx, b, W1, 2 * W, nbs), 2 * W, nbs);
  int32_t* st = ctx->ws_t<int32_t>(2 * nbs);
  HIPCHK(hipMemsetAsync(st, 0, 2 * nbs * 4, ctx->stream));
  const uint32_t* m2 = decrypt2_crt(sk, both, 2 * nbs, 2 * nbs, st);
  uint32_t *m = ctx->ws_t<uint32_t>((size_t)W2 * nbs), *mu = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
  split2(ctx, m2, 0, W2, nbs, m);
  split2(ctx, m2, 1, W2, nbs, mu);
  sb.m = m;
  rb.mu_b = mu;
  launch_or_flags(st, count, d_status_ct, PGPU_LANE_NONUNIT, ctx->stream);
  launch_or_flags(st + nbs, count, d_status_b, PGPU_LANE_NONUNIT, ctx->stream);
}
void resp_base(const pgpu_seckey* sk, const uint32_t* s, const uint32_t* b, size_t nbs, size_t count, int32_t* d_status, RespBase& rb) {
  resp_base_residues(sk, s, b, nbs, rb);
  resp_base_plaintext(sk, b, nbs, count, d_status, rb);
}
// per-instance exponents of the response, for EVERY instance (the challenge bits are not known yet): E1, E2 modulo
// p - 1 and q - 1 (the odd part through a Montgomery product, the 2-part from the lowest limbs, then the CRT lift)
struct RespExps {
  uint32_t* e1p[2] = {nullptr, nullptr};                            // E1 mod (pr - 1): eo1.w limbs
  uint32_t* e2p[2] = {nullptr, nullptr};                            // E2 mod (pr - 1)
};
void resp_exps(const pgpu_seckey* sk, const uint32_t* xn, const uint32_t* an, const uint32_t* en, size_t nb, RespExps& re) {
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx& mn2 = sk->pk->mn2;
  const int W2 = mn2.WT;
  uint32_t* ls = ctx->ws_t<uint32_t>(nb);
  uint32_t* lb = ctx->ws_t<uint32_t>(nb);
  launch_exp_low_combine(xn, an, en, ls, lb, nb, ctx->stream);
  Fork fo(ctx);
  for (int half = 0; half < 2; ++half) {
    fo.chain(half);
    const ExpOrder& eo = half ? sk->eo1_q : sk->eo1_p;
    const ModCtx& mm = eo.modd;
    const size_t sm = (size_t)mm.WT * nb;
    uint32_t *am = ctx->ws_t<uint32_t>(sm), *em = ctx->ws_t<uint32_t>(sm), *xm = ctx->ws_t<uint32_t>(sm), *pm = ctx->ws_t<uint32_t>(sm),
             *esm = ctx->ws_t<uint32_t>(sm), *ebm = ctx->ws_t<uint32_t>(sm), *zero = ctx->ws_t<uint32_t>(sm);
    HIPCHK(hipMemsetAsync(zero, 0, sm * 4, ctx->stream));
    reduce_mod_wide(ctx, mm, an, W2, am, nb);
    reduce_mod_wide(ctx, mm, en, W2, em, nb);
    reduce_mod_wide(ctx, mm, xn, W2, xm, nb);
    modmul_arrays(ctx, mm, am, em, nb, pm);                                            // an en mod m
    launch_sub_mod(xm, pm, mm.d_nmod, esm, mm.WT, nb, ctx->stream);                      // xn - an en mod m
    launch_sub_mod(zero, em, mm.d_nmod, ebm, mm.WT, nb, ctx->stream);                    // -en mod m
    re.e1p[half] = ctx->ws_t<uint32_t>((size_t)eo.w * nb);
    re.e2p[half] = ctx->ws_t<uint32_t>((size_t)eo.w * nb);
    launch_exp_order_lift(ls, 1, esm, mm.WT, eo.m_limbs.d, eo.t, eo.minv, re.e1p[half], eo.w, nb, ctx->stream);
    launch_exp_order_lift(lb, 1, ebm, mm.WT, eo.m_limbs.d, eo.t, eo.minv, re.e2p[half], eo.w, nb, ctx->stream);
  }
  fo.join();
}
// c[g] = s^(E1) b^(E2) mod n^3 for the `nb` (compacted) instances with challenge bit 1: sti[g] = row of instance g's statement in rb (device,
// nb entries), en: e^n modulo n^2 (mn2.WT limbs), e1p / e2p: the exponents modulo p - 1, q - 1 (eo1.w limbs); all stride
// nb.  out: WT(n^3) limbs.  d_status (nb entries, zeroed) is flagged where the lift met a non-unit.
void struct_response(const pgpu_seckey* sk, const RespBase& rb, const uint32_t* sti, const uint32_t* en,
                     const uint32_t* const e1p[2], const uint32_t* const e2p[2], size_t nb, uint32_t* out, int32_t* d_status,
                     const std::function<hipEvent_t()>& plaintext_beside = nullptr) {
  // plaintext_beside: rb.mu_b does not exist yet -- the callback issues its computation on another stream of the context and returns
  // the event behind it.  It is called right BEHIND the launch of the ladder modulo the primes: that ladder is the critical path (the
  // lift waits for it) and, inside a prover call, takes compute units of its own (plan::lds_share: 64 workgroups at 8 192 numbers);
  // the decryption then fills the other CUs and has until the end of the lift.  Issued before the ladder it sits on every CU and the
  // ladder's workgroups wait for empty ones (9 + 8 ms one after the other); with both launches sharing SIMDs each runs at half speed
  // (15.5 and 12.8 ms side by side).  And issued before this function's own chains it ends up in a hardware queue in front of one of
  // them (the runtime maps a process's streams onto four queues).  Only the lane of the closed form waits for the plaintext.
  pgpu_ctx* ctx = sk->ctx;
  const pgpu_pubkey* pk = sk->pk;
  const ModCtx &mn2 = pk->mn2, &mn3 = *pk->mn3;
  const int W2 = mn2.WT, W3 = mn3.WT, H = sk->mp.WT;
  const size_t S1 = (size_t)H * nb;
  auto per_number = [&](const uint32_t* in, int w) {
    uint32_t* o = ctx->ws_t<uint32_t>((size_t)w * nb);
    launch_gather(in, rb.nbs, sti, nb, o, nb, w, ctx->stream);
    return o;
  };
  // c modulo the primes: (s mod pr)^(E1) (b mod pr)^(E2), one chain of squarings, two per-number window tables; both halves in one launch
  const int we = sk->eo1_p.w, wb = 4;
  const PrimeShape ps = prime_shape(sk, nb);                          // (few numbers: four lanes per number, slots of Hs limbs)
  const size_t Ss = (size_t)ps.Hs * nb;
  const bool nm4 = ctx->use_nm4 && ctx->use_asm && H == 37 && plan::pair_nm4_fits(nb, ps.Hs);
  const uint32_t TA = 5, TB = TA + (uint32_t)perlane_table_slots(wb, nm4), NS = TB + (uint32_t)perlane_table_slots(wb, nm4);
  uint32_t* mem1[2];
  const uint32_t* dig[2];
  Prog lad[2];
  {
    Fork f1(ctx);
    for (int half = 0; half < 2; ++half) {
      f1.chain(half);
      mem1[half] = ctx->ws_t<uint32_t>(Ss * (size_t)NS);             // slots (H limbs): 0 s, 1 b, 2 tmp, 3 out, TA.. / TB.. the tables
      prime_slot_fill(ctx, ps, mem1[half], per_number(rb.sp[half], H), nb);
      prime_slot_fill(ctx, ps, mem1[half] + Ss, per_number(rb.bp[half], H), nb);
      // the two exponents of a number one after the other in the rows of `digits`
      uint32_t* d2 = ctx->ws_t<uint32_t>((size_t)2 * we * nb);
      HIPCHK(hipMemcpyAsync(d2, e1p[half], (size_t)we * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(d2 + (size_t)we * nb, e2p[half], (size_t)we * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
      dig[half] = d2;
      Prog& pr = lad[half];
      pr.op(VM_LOAD, 0); pr.op(VM_MULC, C_R2); pr.op(VM_STORE, 0);     // into Montgomery form (the generic kernel's working form)
      pr.op(VM_LOAD, 1); pr.op(VM_MULC, C_R2); pr.op(VM_STORE, 1);
      std::vector<PerNumberBase> pn;
      pn.push_back(PerNumberBase{we, 0, TA, 0});
      pn.push_back(PerNumberBase{we, 1, TB, (uint32_t)perlane_windows(we, wb)});
      emit_modexp_multi(pr, pn, wb, {}, 2, 3, (uint32_t)C_ONE_M, nm4);
      pr.op(VM_LOAD, 3); pr.op(VM_MULC, C_ONE); pr.op(VM_STORE, 3);    // and out of it
      pr.end();
    }
    f1.join();
  }
  {
    SegSpec sp{ps.m[0], &lad[0], mem1[0], dig[0]}, sq{ps.m[1], &lad[1], mem1[1], dig[1]};
    run_vm(ctx, nb, sp, &sq, true);
  }
  const hipEvent_t plaintext_ready = plaintext_beside ? plaintext_beside() : nullptr;
  // the <1 + n> coordinate on a lane of its own, beside the lift: k = - mu_b en mod n^2, G = (1 + n)^k
  uint32_t* G = ctx->ws_t<uint32_t>((size_t)W3 * nb);
  // (with the plaintext still in flight the closed form goes to the stream the decryption is on -- the side stream, which has a hardware
  // queue of its own: on lane 2 it shares a queue with the lane the lift's chains fork to, and the lift would wait for the decryption)
  SideStream g_lane(ctx, plaintext_beside ? 0 : 2);
  g_lane.enter(g_lane.mark());
  if (plaintext_ready) HIPCHK(hipStreamWaitEvent(ctx->stream, plaintext_ready, 0));
  {
    uint32_t *k1 = ctx->ws_t<uint32_t>((size_t)W2 * nb), *k2 = ctx->ws_t<uint32_t>((size_t)W2 * nb), *kk = ctx->ws_t<uint32_t>((size_t)W2 * nb);
    HIPCHK(hipMemsetAsync(k1, 0, (size_t)W2 * nb * 4, ctx->stream));                  // (mu_s E1 = 0 modulo n^2: see resp_base)
    modmul_arrays(ctx, mn2, per_number(rb.mu_b, W2), en, nb, k2);
    launch_sub_mod(k1, k2, mn2.d_nmod, kk, W2, nb, ctx->stream);
    gm2_from_reduced(ctx, pk, kk, nb, G);
  }
  g_lane.leave();
  uint32_t* tt[2];
  for (int half = 0; half < 2; ++half) {
    tt[half] = mem1[half] + 3 * Ss;
    launch_canon(tt[half], (half ? sk->mq : sk->mp).d_nmod, H, nb, ctx->stream);
  }
  uint32_t* T = ctx->ws_t<uint32_t>((size_t)W3 * nb);
  {
    ForceLds lift_lds(ctx, plaintext_beside ? plan::kLateLiftLds : -1);      // (beside the decryption that is still running: plan.hpp)
    teichmueller_lift(sk, tt, nb, d_status, T);
  }
  g_lane.join();
  modmul_arrays(ctx, mn3, G, T, nb, out);
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
  // (plan.hpp: window width of the per-number exponents and the gate of the p-adic split, decided in ONE place)
  const plan::Crt3Ladder lad = plan::crt3_ladder(nb, sk->mp.WT, W, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), exps != nullptr, true);
  if (triple_usable(ctx, mp3) && triple_usable(ctx, mq3) && lad.triple && (exps || base2 || e->bit_length() >= 64)) {
    // both halves on the three-digit kernel (digits modulo p and q, 37 limbs for 2048-bit keys): a squaring is 37 rows
    // where the wave-sliced 110-limb kernel has 110 -- what counts for the half-size, latency-bound batches of the response
    const int win = lad.win;
    const bool nm5 = lad.nm5;                    // (number-major 5-bit tables: VM_MULVT5)
    const uint32_t tab2 = 5 + (uint32_t)perlane_table_slots(win, nm5);
    const int nslots = base2 ? (int)tab2 + (1 << (dual_sliding_bits(win) - 1)) : 5 + perlane_table_slots(win, nm5);
    uint32_t* g = ctx->ws_t<uint32_t>(S * 6);       // generic slots (W limbs): 0 x_p, 1 x_q, 2 A, 3 B, 4 h, 5 scratch
    // Exponents modulo the orders of the unit groups of p^3 and q^3 (a quarter shorter than exponents modulo n^2): each half
    // gets its own reduced exponents and its own program (context flag "exp_order", 0: the exponents as given).
    const bool reduce_e = ctx->use_exp_order && sk->eo_p.ok && sk->eo_q.ok;
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
    // (below one wave per SIMD for the stage modulo p^2 the ladders are bound by their length, and one ladder is shorter than two)
    // (7-bit windows of r0 while their 128-entry tables fit the 32-bit gather offsets -- 75 000 numbers for 37-limb primes --
    // and 5-bit windows on number-major tables beyond: a big batch keeps the split, it does not fall back to the long ladder)
    const bool split_prereq = ctx->use_lift && reduce_e && (exps || base2) && reduced_pn && (win == 7 || nm5) && sk->mp2.WT == 2 * sk->mp.WT &&
                              sk->eo_p.w <= 3 * sk->mp.WT && sk->pinv2k_2.d && mp3.triple.root->WT == sk->mp.WT;
    if (split_prereq && lad.split) {
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
  const plan::Crt3Two two = plan::crt3_two(nb, H, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), true);
  if (!two.usable) return false;
  const int win = 7;
  // The p-adic split of BOTH exponents (what pow_n3_crt does for one): with e = e0 + e1 prime,
  //     A^(ea) B^(eb) = (A^(ea1) B^(eb1))^prime * A^(ea0) B^(eb0)    and    W^prime mod prime^3 depends on W mod prime^2 only,
  // so W is an interleaved ladder of 2 047 squarings modulo prime^2 on the one-lane pair kernel (two per-number exponents, 4-bit
  // windows) and the rest an interleaved ladder of 1 024 squarings modulo prime^3 with two per-number exponents and the shared
  // exponent prime on W -- where the unsplit ladder squares 3 071 times modulo prime^3.  Taken when the stage modulo prime^2
  // fills the chip (the response batch of ProveDDLEQ at secpar 40: half of 61 440 instances).
  {
    const int H1 = sk->mp.WT, W2 = sk->mp2.WT;
    if (ctx->use_lift && sk->mp2.WT == 2 * sk->mp.WT && sk->mq2.WT == 2 * sk->mq.WT && we <= 3 * H1 &&
        sk->pinv2k_2.d && sk->qinv2k_2.d && H1 == H && two.split) {
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

}  // namespace pgi

// ProveDDLEQ (ddleq.go:27-40) for `n_statements` statements (ct1, ct2, a, b) with `secpar` instances each -- draws x, y
// supplied, instance k of statement j in row j * secpar + k of x / y / alpha / e / f.  What ddleq.go:55-127 recomputes in every
// instance although it depends on the statement only is computed ONCE per statement: the sanity check ct1^(a^n) b^(n^2) == ct2
// (:62-69), a^n (:104), a^-1 (:95) and (a^n)^-1, s = ExtractRandonness(ct1) (:103) and the unit tests of s and b; per instance
// remain x^n, alpha = ct1^(x^n) y^(n^2), the challenge bit and -- for bit 1 -- the response ladder.  The integers are those of
// `secpar` calls of proveDDLEQInstance with the same draws.  secpar = 1 is pgpu_ddleq_prove.
//
// One call = one ProveCall: its stages, in the order run() issues them (each stage only ISSUES work; the two places where the host
// waits for the device are read_back() and, inside statement_side_work(), the inversion tree's root):
//   unpack_inputs        operands to limbs; per-statement rows repeated for the instances of their statement
//   powers_of_n          a^n | x^n modulo n^2 in one launch (through the primes and a Teichmueller lift for the key holder)
//   structure_base       side lane: ct1 modulo the primes, the level-two plaintext of ct1 (what the structure path needs per statement)
//   extract_randomness   s = ExtractRandonness(ct1), on the side stream where the a^n | x^n launch leaves room
//   sanity_and_alpha     ct1^(a^n) b^(n^2) | ct1^(x^n) y^(n^2): structure path, literal ladders or (no factorisation) plain ladders;
//                        the comparison with ct2; the hash right behind Alpha
//   statement_side_work  side stream: a^-1 | (a^n)^-1 from one inversion tree, the unit test of s b, the response prepared for EVERY
//                        instance (exponents, bases) before the challenge bits exist
//   read_back            the host learns sanity flags, unit flags and challenge bits; non-units redo the launch literally
//   respond              the instances with challenge bit 1: gather what was prepared, one ladder (or the structure path, or the
//                        reference's literal sequence), scatter E and F
//   write_outputs        Alpha, E, F to the caller's buffers
namespace {

struct HostTrace {      // PGPU_HOST_TRACE=1 (measurements): when the HOST passed each stage of the call (ms from its start) -- where it waited for the device
  bool on;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  std::vector<std::pair<const char*, double>> log;
  HostTrace() {
    static const bool host_trace = [] { const char* e = getenv("PGPU_HOST_TRACE"); return e && atoi(e) != 0; }();
    on = host_trace;
  }
  void operator()(const char* what) {
    if (on) log.emplace_back(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  ~HostTrace() {
    if (log.empty()) return;
    fprintf(stderr, "[pgpu] prove host:");
    for (auto& x : log) fprintf(stderr, " %s %.2f |", x.first, x.second);
    fprintf(stderr, "\n");
  }
};

struct ExclusiveCall {  // which of the call's launches take a CU per workgroup: plan::lds_share
  pgpu_ctx* c;
  explicit ExclusiveCall(pgpu_ctx* c_) : c(c_) { c->exclusive_call = true; }
  ~ExclusiveCall() { c->exclusive_call = false; }
};

struct Background {     // side-lane ladders at wave priority 0 for the length of a scope (flag "background")
  pgpu_ctx* c; bool on;
  Background(pgpu_ctx* c_, bool on_) : c(c_), on(on_) { if (on) c->background_launch = true; }
  ~Background() { if (on) c->background_launch = false; }
};

struct ProveCall {
  // ---- the call
  const pgpu_seckey* sk;
  pgpu_ctx* ctx;
  const pgpu_pubkey* pk;
  const size_t S, secpar;
  const uint8_t *ct1, *ct2, *a, *b, *x, *y;
  const size_t ct_stride, n_stride;
  uint8_t *alpha, *e_out, *f_out;
  const size_t e_stride;
  const int mem;
  const ModCtx &mn, &mn2, &mn3;
  const int W1, W2, W3;
  const bool crt3;                                         // the prover holds the factorisation: exponentiations modulo n^3 go through p^3 and q^3 (pow_n3_crt)
  const size_t batch, nbs, nb, nt;                         // instances; statements / instances padded to whole workgroups; numbers of the two-batch launches
  const BigU &N, &N2;
  HostTrace HT;
  ExclusiveCall exclusive_call;
  SideStream side;                                         // what depends on the statement only and on no ladder of the call
  SideStream base_lane;                                    // (a lane of its own: s keeps the side stream, beside the same launch)
  // ---- operands as limbs: per statement (row stride nbs) and per instance (row stride nb)
  uint32_t *c1s = nullptr, *c2s = nullptr, *al = nullptr, *bl = nullptr, *xl = nullptr, *yl = nullptr;
  uint32_t* d_stmt = nullptr;                              // instance -> its statement
  uint32_t *c1 = nullptr, *c2 = nullptr;                   // ct1, ct2 per instance
  // ---- values of the call
  uint32_t *an = nullptr, *xn = nullptr;                   // a^n per statement, x^n per instance (W2 limbs)
  uint32_t* qs = nullptr;                                  // s per statement (W1 limbs, stride nbs)
  hipEvent_t inputs_ready = nullptr, an_ready = nullptr, residues_ready = nullptr, plaintext_ready = nullptr;
  bool by_struct = false;
  StructBase sbase;
  int32_t *d_st_stmt = nullptr, *d_st_num = nullptr;
  uint32_t *bn2 = nullptr, *t3 = nullptr, *san = nullptr, *alp = nullptr;
  int32_t* d_ok = nullptr;
  int32_t* chal = nullptr;
  bool hash_early = false;
  std::vector<int32_t> hok, hch, hst_stmt, hst_num;
  // the response, prepared before the bits are known
  uint32_t *qainv = nullptr, *qani = nullptr;
  int32_t* d_badinv = nullptr;
  hipEvent_t axn_done = nullptr;                            // a^n | x^n is through its eight-lane ladders (powers_of_n)
  bool any_badinv = false, sb_units = false, early = false, resp_struct = false, resp_late = false, one_ladder = false;
  const uint8_t* sb_root = nullptr;
  hipEvent_t root_ready = nullptr;
  RespBase rbase;
  RespExps rexps;
  uint32_t* en_all = nullptr;
  int32_t* d_st_rstmt = nullptr;
  uint32_t *ge_all = nullptr, *es_all[2] = {nullptr, nullptr}, *eb_all[2] = {nullptr, nullptr};
  PreBases pre;
  uint32_t *eo = nullptr, *fo = nullptr;                   // E, F of every instance

  ProveCall(const pgpu_seckey* sk_, size_t S_, size_t secpar_, const uint8_t* ct1_, const uint8_t* ct2_, size_t ct_stride_, const uint8_t* a_,
            const uint8_t* b_, const uint8_t* x_, const uint8_t* y_, size_t n_stride_, uint8_t* alpha_, uint8_t* e_out_, size_t e_stride_,
            uint8_t* f_out_, int mem_)
      : sk(sk_), ctx(sk_->ctx), pk(sk_->pk), S(S_), secpar(secpar_), ct1(ct1_), ct2(ct2_), a(a_), b(b_), x(x_), y(y_), ct_stride(ct_stride_),
        n_stride(n_stride_), alpha(alpha_), e_out(e_out_), f_out(f_out_), e_stride(e_stride_), mem(mem_), mn(pk->mn), mn2(pk->mn2), mn3(*pk->mn3),
        W1(mn.WT), W2(mn2.WT), W3(mn3.WT),
        crt3(sk->has_crt2 && sk->c_p3invR >= 0 && 2 * sk->mp3.WT >= W3 && ctx->use_pair), batch(S * secpar), nbs(round_up(S, VM_BLOCK)),
        nb(round_up(batch, VM_BLOCK)), nt(nbs + nb), N(pk->N), N2(mn2.N), exclusive_call(ctx), side(ctx), base_lane(ctx, 3) {}

  void perlane3(const uint32_t* base, const uint32_t* exps, int we, size_t nbx, uint32_t* outp) {
    if (crt3) pow_n3_crt(sk, base, W3, exps, we, nullptr, nbx, outp);
    else perlane_pow(ctx, mn3, base, exps, we, nbx, outp);
  }
  void shared3(const uint32_t* base, int wb, const BigU& ex, size_t nbx, uint32_t* outp) {
    if (crt3) pow_n3_crt(sk, base, wb, nullptr, 0, &ex, nbx, outp);
    else shared_pow(ctx, mn3, base, wb, ex, nbx, outp);
  }
  uint32_t* up(const uint8_t* buf, size_t stride, int w, size_t count, size_t nbx) {
    uint32_t* l = ctx->ws_t<uint32_t>((size_t)w * nbx);
    unpack_operand(ctx, buf, stride, stride, count, mem, l, w, nbx);
    return l;
  }
  uint32_t* expand(uint32_t* in, int w) {                  // a per-statement array repeated for the instances of its statement
    if (secpar == 1) return in;
    uint32_t* o = ctx->ws_t<uint32_t>((size_t)w * nb);
    launch_gather(in, nbs, d_stmt, batch, o, nb, w, ctx->stream);
    return o;
  }

  void unpack_inputs() {
    if (ct_stride != mn3.nbytes) api_throw(PGPU_ERR_INVALID, "ciphertext stride must be the byte length of n^3");
    if (n_stride * 8 > (size_t)LB * W1 + 7) api_throw(PGPU_ERR_INVALID, "a, b, x, y must fit the width of n");
    // per statement (S numbers, row stride nbs) ...
    c1s = up(ct1, ct_stride, W3, S, nbs);
    c2s = up(ct2, ct_stride, W3, S, nbs);
    al = up(a, n_stride, W1, S, nbs);
    bl = up(b, n_stride, W1, S, nbs);
    // ... and per instance (S * secpar numbers, row stride nb)
    xl = up(x, n_stride, W1, batch, nb);
    yl = up(y, n_stride, W1, batch, nb);
    if (secpar > 1) {
      std::vector<uint32_t> st(batch);
      for (size_t i = 0; i < batch; ++i) st[i] = (uint32_t)(i / secpar);
      d_stmt = ctx->upload_words(st);
    }
    c1 = expand(c1s, W3);
    c2 = expand(c2s, W3);
    // ---- what depends on the STATEMENT only and on no ladder of this call goes to the side stream, beside the big launches
    // (the GPU was busy, but with ~800 small launches in a row between the ladders: 30 of 188 ms per 16 384 instances):
    //   s = ExtractRandonness(ct1) (ddleq.go:103) for every statement -- a latency-bound launch beside the a^n | x^n
    //   launch, which fills half the chip; then, behind a^n, the inversion tree for a^-1 | (a^n)^-1 and the unit test of s b
    //   beside the Alpha ladders.  Which statements have an instance with challenge bit 1 is not known yet: all are done
    //   (the side work is bound by launch latencies, not by its width).
    inputs_ready = side.mark();
  }

  // ---- sanity check (ddleq.go:62-69): ct1^(a^n mod n^2) * b^(n^2) mod n^3 == ct2, else the reference panics -- once per
  // statement.  Independent exponentiations of the same shape share a launch (the chip is filled better and, for the half-size
  // batches of the response, a latency-bound launch is saved outright): a^n (S numbers) | x^n (S secpar numbers), then the
  // sanity value and alpha -- ct1^(a^n) b^(n^2) | ct1^(x^n) y^(n^2) -- and further down s^(a^n) | s^(x^n).
  void powers_of_n() {
    an = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
    xn = ctx->ws_t<uint32_t>((size_t)W2 * nb);
    uint32_t* ax = concat_ab(ctx, al, nbs, xl, nb, W1);
    uint32_t* axn;
    if (pow_n2_crt_usable(sk)) {
      axn = pow_n2_crt(sk, ax, N, nbs + nb);               // the prover holds p and q
      if (crt_pair8_usable(sk, nbs + nb) && ctx->use_side) {
        // the second stage ran on eight lanes per number: up to a wave on EVERY SIMD for 3.6 ms.  ct1's decryption (a CU per workgroup on
        // half the chip) beside it would make it two rounds of that; it has slack until the end of the lifts and starts behind this launch
        axn_done = ctx->next_sync_ev();
        HIPCHK(hipEventRecord(axn_done, ctx->stream));
      }
    } else {
      axn = ctx->ws_t<uint32_t>((size_t)W2 * (nbs + nb));
      shared_pow(ctx, mn2, ax, W1, N, nbs + nb, axn);
    }
    split_ab(ctx, axn, nbs, nb, 0, W2, an);
    split_ab(ctx, axn, nbs, nb, 1, W2, xn);
    HT("a^n|x^n issued");
  }

  // The structure path's per-statement part (struct_pow_n3 below) -- the level-two plaintext of ct1 through p^3, q^3 and ct1 modulo
  // the primes -- depends on the inputs only: it is issued beside the a^n | x^n launch on a lane of its own (s has the side
  // stream), and runs where wave slots are free: beside that launch at 16 384 instances (one lane per number: one wave per SIMD),
  // beside the ladders modulo the primes otherwise.
  // (flag "background", measurements: the side lanes' ladders of a call that fills the chip at wave priority 0 -- a^n | x^n gates
  // everything behind it, the plaintext of ct1 is needed after the lifts, s after the hash.  Measured with the main launch spread over
  // the CUs (run_vm): 16 384 instances 130.4 ms without, 132.6 +- 4 with -- what a^n | x^n gains, the ladder modulo the primes and the
  // lifts lose to the side launches that are still running beside them; secpar 40 the same within noise.  Off by default.)
  // whether the response will go through the structure with its per-statement part made BEFORE the hash (plan::response_by_structure:
  // from four instances per statement, and for small calls): decided here, before the structure chain of ct1, which then takes b's
  // plaintext along in its decryption
  void plan_response() {
    one_ladder = crt3 && ctx->use_lift && sk->eo_p.ok && sk->eo_q.ok && sk->eo_p.w == sk->eo_q.w && W2 <= 2 * sk->eo_p.modd.WT &&
                 W2 <= 2 * sk->eo_q.modd.WT;
    resp_struct = by_struct && one_ladder && plan::response_by_structure(S, batch, nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus));
  }

  void structure_base() {
    by_struct = crt3 && struct_pow_usable(sk);
    plan_response();
    if (!by_struct) return;
    Background bg(ctx, ctx->use_background);
    // (flag "base_early": the links of this chain keep the kernels' own LDS size -- they run beside the a^n | x^n ladders at once
    // instead of waiting for an EMPTY compute unit, which there is none of before those ladders end; the decryption of ct1 then shares
    // the SIMDs with them: 246 + 183 VGPRs, a wave of each)
    struct LinksBeside {
      pgpu_ctx* c; bool was;
      LinksBeside(pgpu_ctx* c_, bool on) : c(c_), was(c_->use_exclusive_short) { if (on) c->use_exclusive_short = false; }
      ~LinksBeside() { c->use_exclusive_short = was; }
    } links_beside(ctx, ctx->use_base_early);
    d_st_stmt = ctx->ws_t<int32_t>(nbs);
    d_st_num = ctx->ws_t<int32_t>(nt);
    base_lane.enter(inputs_ready);
    HIPCHK(hipMemsetAsync(d_st_stmt, 0, nbs * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(d_st_num, 0, nt * 4, ctx->stream));
    struct_base_residues(sk, c1s, nbs, sbase);
    if (base_lane.on) {
      residues_ready = ctx->next_sync_ev();
      HIPCHK(hipEventRecord(residues_ready, ctx->stream));
    }
    if (axn_done) HIPCHK(hipStreamWaitEvent(ctx->stream, axn_done, 0));
    if (resp_struct) {
      d_st_rstmt = ctx->ws_t<int32_t>(nbs);
      HIPCHK(hipMemsetAsync(d_st_rstmt, 0, nbs * 4, ctx->stream));
      struct_and_resp_plaintexts(sk, c1s, bl, nbs, S, d_st_stmt, d_st_rstmt, sbase, rbase);
    } else {
      struct_base_plaintext(sk, c1s, nbs, S, d_st_stmt, sbase);
    }
    if (base_lane.on) {
      plaintext_ready = ctx->next_sync_ev();
      HIPCHK(hipEventRecord(plaintext_ready, ctx->stream));
    }
    base_lane.leave();
  }

  // s = ExtractRandonness(ct1) at level two (operations.go:75-91): z = G^(-v) ct1 mod n^3 with v = Decrypt(ct1)
  // (operations.go:81-86) is only ever used modulo n (:88), and G^v = (1 + n)^v = 1 (mod n) whatever v is (the prover
  // requires G = n + 1): z = ct1 (mod n).  No decryption, no G^v, no inversion modulo n^3 -- the same s.
  void extract_randomness() {
    BigU ns_inv;
    if (!hostbig::modinv(N2 % sk->lambda, sk->lambda, ns_inv)) api_throw(PGPU_ERR_NOT_INVERTIBLE, "n^2 is not invertible mod lambda");
    // Beside the a^n | x^n launch only where that launch leaves the second wave slot of the SIMDs free (one wave per SIMD or
    // less: 16 384 instances at secpar 1): a launch that fills both slots would lose one of them on half the chip to the side
    // launch for its whole length (measured at 32 768 instances: 48 -> 76 ms for 8 ms hidden).  Then s follows on the main stream.
    // (A side launch of a few dozen waves -- the statements of a secpar-40 call -- costs the big launch next to nothing.)
    const bool s_beside = plan::extract_beside(nbs, nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus));
    if (s_beside) side.enter(inputs_ready);
    {
      Background bg(ctx, ctx->use_background && s_beside);
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
    HT("s issued");
    an_ready = side.mark();
  }

  // one interleaved ladder per number and CRT half on ct1 itself (pow_n3_crt: the p-adic split for batches that fill the chip)
  void literal_ladders() {
    uint32_t* cc2 = concat_ab(ctx, c1s, nbs, c1, nb, W3);
    uint32_t* ee2 = concat_ab(ctx, an, nbs, xn, nb, W2);
    uint32_t* by2 = concat_ab(ctx, bl, nbs, yl, nb, W1);
    uint32_t* o2 = ctx->ws_t<uint32_t>((size_t)W3 * nt);
    pow_n3_crt(sk, cc2, W3, ee2, W2, &N2, nt, o2, by2, W1);
    split_ab(ctx, o2, nbs, nb, 0, W3, san);
    split_ab(ctx, o2, nbs, nb, 1, W3, alp);
  }

  // ---- challenge bit = LSB SHA-256(ct2 || x || y || alpha)  (ddleq.go:91; ct1 is skipped: random_oracle.go:24-26)
  void hash_bits() {
    const uint32_t* parts[4] = {c2, xl, yl, alp};
    const int widths[4] = {W3, W1, W1, W3};
    HIPCHK(hipMemsetAsync(chal, 0, nb * 4, ctx->stream));
    launch_sha256_transcript(parts, widths, 4, nb, batch, nullptr, chal, ctx->stream);
  }

  // ct1^(a^n) * b^(n^2) and ct1^(x^n) * y^(n^2), both batches (S statements | S secpar instances) in the same launches
  void sanity_and_alpha() {
    bn2 = ctx->ws_t<uint32_t>((size_t)W3 * nbs);
    t3 = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    san = ctx->ws_t<uint32_t>((size_t)W3 * nbs);
    alp = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    // Through the structure of the unit group where the key serves it (struct_pow_n3): ct1's plaintext once per statement, then per
    // number a ladder modulo the primes and ONE lift modulo p^3, q^3.  Lanes that meet a non-unit are flagged; if a REAL lane is
    // (never for honest inputs) the literal ladders redo the launch after the host has seen the flags (read_back).
    if (by_struct) {
      // the main stream waits for ct1 modulo the primes only; the decryption (a latency-bound launch when the statements are few) goes
      // on beside the ladders modulo the primes, and the lane of the closed form waits for it
      if (residues_ready) HIPCHK(hipStreamWaitEvent(ctx->stream, residues_ready, 0));
      std::vector<uint32_t> sv(nt, 0);                       // statement of every number: the sanity values | the instances
      for (size_t g = 0; g < S; ++g) sv[g] = (uint32_t)g;
      for (size_t i = 0; i < batch; ++i) sv[nbs + i] = (uint32_t)(i / secpar);
      const uint32_t* d_sv = ctx->upload_words(sv);
      uint32_t* ee2 = concat_ab(ctx, an, nbs, xn, nb, W2);
      uint32_t* by2 = concat_ab(ctx, bl, nbs, yl, nb, W1);
      uint32_t* o2 = ctx->ws_t<uint32_t>((size_t)W3 * nt);
      struct_pow_n3(sk, sbase, d_sv, ee2, by2, nt, o2, d_st_num, plaintext_ready);
      HT("alpha issued");
      base_lane.join();
      split_ab(ctx, o2, nbs, nb, 0, W3, san);
      split_ab(ctx, o2, nbs, nb, 1, W3, alp);
    } else if (crt3) {
      literal_ladders();
    } else {
      shared3(bl, W1, N2, nbs, bn2);
      perlane3(c1s, an, W2, nbs, t3);
      modmul_arrays(ctx, mn3, t3, bn2, nbs, san);
    }
    d_ok = ctx->ws_t<int32_t>(nbs);
    launch_equal(san, c2s, W3, nbs, S, d_ok, ctx->stream);
    hok.resize(S);
    // With the CRT path Alpha is on the main stream by now: the hash follows it at once -- the host is about to wait inside the side
    // section (the inversion tree inverts its root on the host) and would otherwise launch the hash only after that: 8 of 133 ms at
    // 16 384 instances.  The sanity flags are read together with the bits; a failed check discards them.
    chal = ctx->ws_t<int32_t>(nb);
    hash_early = crt3;
    if (hash_early) hash_bits();
    hch.resize(batch);
  }

  // the two exponents of the one-ladder response modulo the group orders of p^3 and q^3 (ord = 2^t m: the odd part through a Montgomery
  // product modulo m, the 2-part from the lowest limbs, then the CRT lift) for `nbx` numbers: es = xn - an en, eb = -en
  void response_exponents(const uint32_t* gxn, const uint32_t* gan, const uint32_t* en, size_t nbx, const uint32_t* es[2], const uint32_t* eb[2]) {
    uint32_t* ls = ctx->ws_t<uint32_t>(nbx);
    uint32_t* lb = ctx->ws_t<uint32_t>(nbx);
    launch_exp_low_combine(gxn, gan, en, ls, lb, nbx, ctx->stream);
    Fork fo_(ctx);                                                     // the exponents modulo the two group orders side by side
    for (int half = 0; half < 2; ++half) {
      fo_.chain(half);
      const ExpOrder& eo_ = half ? sk->eo_q : sk->eo_p;
      const ModCtx& mm = eo_.modd;
      const size_t sm = (size_t)mm.WT * nbx;
      uint32_t *am = ctx->ws_t<uint32_t>(sm), *em_ = ctx->ws_t<uint32_t>(sm), *xm = ctx->ws_t<uint32_t>(sm),
               *pm = ctx->ws_t<uint32_t>(sm), *esm = ctx->ws_t<uint32_t>(sm), *ebm = ctx->ws_t<uint32_t>(sm),
               *zero = ctx->ws_t<uint32_t>(sm);
      HIPCHK(hipMemsetAsync(zero, 0, sm * 4, ctx->stream));
      reduce_mod(ctx, mm, gan, W2, am, nbx);
      reduce_mod(ctx, mm, en, W2, em_, nbx);
      reduce_mod(ctx, mm, gxn, W2, xm, nbx);
      modmul_arrays(ctx, mm, am, em_, nbx, pm);                                         // an en mod m
      launch_sub_mod(xm, pm, mm.d_nmod, esm, mm.WT, nbx, ctx->stream);                   // xn - an en mod m
      launch_sub_mod(zero, em_, mm.d_nmod, ebm, mm.WT, nbx, ctx->stream);                // -en mod m
      uint32_t* e1 = ctx->ws_t<uint32_t>((size_t)eo_.w * nbx);
      uint32_t* e2 = ctx->ws_t<uint32_t>((size_t)eo_.w * nbx);
      launch_exp_order_lift(ls, 1, esm, mm.WT, eo_.m_limbs.d, eo_.t, eo_.minv, e1, eo_.w, nbx, ctx->stream);
      launch_exp_order_lift(lb, 1, ebm, mm.WT, eo_.m_limbs.d, eo_.t, eo_.minv, e2, eo_.w, nbx, ctx->stream);
      es[half] = e1;
      eb[half] = e2;
    }
    fo_.join();
  }

  // ---- side stream, behind a^n (the Alpha ladders are in flight on the main stream): a^-1 and (a^n)^-1 modulo n^2 for
  // every statement from ONE inversion tree (both batches side by side; a non-unit is flagged per lane and matters only if
  // one of its instances draws challenge bit 1), and the unit test of s b for the one-ladder form of the response
  void statement_side_work() {
    qainv = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
    qani = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
    d_badinv = ctx->ws_t<int32_t>(2 * nbs);
    side.enter(an_ready);
    {
      uint32_t* a1 = ctx->ws_t<uint32_t>((size_t)W1 * nbs);
      launch_restride(al, nbs, S, mn.d_consts + (size_t)C_ONE * W1, a1, nbs, W1, ctx->stream);      // padding lanes: 1
      uint32_t* a2 = zext(ctx, a1, W1, W2, nbs);
      uint32_t* an1 = ctx->ws_t<uint32_t>((size_t)W2 * nbs);
      launch_restride(an, nbs, S, mn2.d_consts + (size_t)C_ONE * W2, an1, nbs, W2, ctx->stream);
      uint32_t* inv2 = batch_inverse(ctx, mn2, concat2(ctx, a2, an1, W2, nbs), 2 * nbs, 2 * nbs, d_badinv, &any_badinv);
      HT("inversions");
      split2(ctx, inv2, 0, W2, nbs, qainv);
      split2(ctx, inv2, 1, W2, nbs, qani);
      if (one_ladder) {
        uint32_t* sb = ctx->ws_t<uint32_t>((size_t)W1 * nbs);
        modmul_arrays(ctx, mn, qs, bl, nbs, sb);
        launch_restride(sb, nbs, S, mn.d_consts + (size_t)C_ONE * W1, sb, nbs, W1, ctx->stream);
        // (no wait: the root of the product tree goes to pinned memory and is looked at once the hash is known -- the side lane
        // crawls beside the lifts, and a host that waits for it here issues the response's preparation 35 ms late)
        sb_root = all_units_begin(ctx, mn, sb, nbs, S);
        if (side.on) {
          root_ready = ctx->next_sync_ev();
          HIPCHK(hipEventRecord(root_ready, ctx->stream));
        }
        sb_units = true;                                     // assumed; checked in read_back, before anything uses the prepared response
        HT("unit test");
      }
      // The response's per-statement bases (s, b: residues modulo p^3, q^3, p^2, q^2 and digit forms) and, for EVERY instance, its
      // e = x a^-1, e^n = x^n (a^n)^-1 and the two exponents of the one-ladder response modulo the group orders: nothing here
      // depends on the challenge bits, so it is done now, beside the Alpha ladders, and the instances that draw bit 1 gather it
      // afterwards (between the hash and the response ladder there is then a handful of gathers instead of ~150 small kernels).
      // (only where the response ladder is certain to take pow_n3_crt_two's kernels whatever the number of bit-1 instances turns out
      // to be: its 7-bit window tables must fit the gather offsets even if every instance draws bit 1)
      // From a few instances per statement on, the response goes through the structure of the unit group (struct_response): the
      // per-statement part -- the level-two "plaintexts" of s and b, s and b modulo the primes -- and every instance's exponents
      // are made here as well.
      // With few instances per statement the per-statement part waits for the hash (resp_late): only the statements that HAVE an
      // instance with challenge bit 1 -- half of them at secpar 1 -- get b's plaintext then, beside the ladder modulo the primes;
      // the exponents of every instance are still made here.  One decryption, one ladder modulo the primes and one lift per bit-1
      // instance: 64 M multiply-adds where the ladder on s and b themselves needs 103 M, and three latency-bound stages of ~9 ms
      // (two of them side by side) where that ladder takes 32 ms at 8 192 numbers.
      resp_struct = resp_struct && sb_units;          // (planned in plan_response; b's plaintext came with ct1's: structure_base)
      resp_late = by_struct && one_ladder && sb_units && !resp_struct && ctx->use_late;
      const bool early_cond = one_ladder && sb_units && ctx->use_early && pre_bases_usable(sk) && !resp_late &&
                              plan::early_response_ok(nb, sk->mp3.triple.root->WT);
      if (resp_struct || resp_late || early_cond) {
        early = !resp_struct && !resp_late;
        // (flag "base_early": the links of this preparation run beside the Alpha ladders and the lifts instead of waiting for an EMPTY
        // compute unit -- there is none while the lifts run, and the main stream then waited 4 ms for this lane after Alpha was known
        // at secpar 40)
        struct LinksBeside {
          pgpu_ctx* c; bool was;
          LinksBeside(pgpu_ctx* c_, bool on) : c(c_), was(c_->use_exclusive_short) { if (on) c->use_exclusive_short = false; }
          ~LinksBeside() { c->use_exclusive_short = was; }
        } links_beside(ctx, ctx->use_base_early);
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
        if (resp_struct || resp_late) {
          if (resp_struct) {
            resp_base_residues(sk, qs, bl, nbs, rbase);           // (the plaintext of b: structure_base, on ct1's lane)
            HT("resp_base issued");
          }
          en_all = en_a;
          resp_exps(sk, xn, gan_a, en_a, nb, rexps);
          HT("resp_exps issued");
        } else {
          const uint32_t *es[2], *eb[2];
          response_exponents(xn, gan_a, en_a, nb, es, eb);
          for (int half = 0; half < 2; ++half) {
            es_all[half] = const_cast<uint32_t*>(es[half]);
            eb_all[half] = const_cast<uint32_t*>(eb[half]);
          }
          pre_bases(sk, zext(ctx, qs, W1, W3, nbs), zext(ctx, bl, W1, W3, nbs), nbs, pre);
        }
      }
    }
    side.leave();
  }

  // device int32 array -> the host vector, through the context's page-locked scratch where it fits (four copies into pageable memory
  // are four round trips of ~0.15 ms between the lifts and the response; into pinned memory they are commands of the stream)
  struct ReadBack { int32_t* pinned; std::vector<int32_t>* dst; size_t n; };
  std::vector<ReadBack> pending;
  void fetch(std::vector<int32_t>& dst, const int32_t* dev, size_t n) {
    dst.resize(n);
    int32_t* stage = nullptr;
    if (ctx->pinned_used + n * 4 + 64 <= pgpu_ctx::kPinnedBytes) stage = (int32_t*)ctx->pinned(n * 4);
    HIPCHK(hipMemcpyAsync(stage ? stage : dst.data(), dev, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (stage) pending.push_back({stage, &dst, n});
  }
  void fetched() {                                          // after the stream has been synchronised
    for (auto& r : pending) memcpy(r.dst->data(), r.pinned, r.n * 4);
    pending.clear();
  }

  // the host learns the sanity flags, the unit flags of the structure path and the challenge bits
  void read_back() {
    // (a device-to-host copy into pageable memory holds the host until the stream has got there: it comes after the side work
    // has been issued, not before)
    fetch(hok, d_ok, S);
    if (by_struct) {
      fetch(hst_stmt, d_st_stmt, S);
      fetch(hst_num, d_st_num, nt);
    }
    if (hash_early) fetch(hch, chal, batch);
    HT("side issued");
    // s b a unit for every statement?  The root of the product tree reaches pinned memory long before Alpha is known: its inversion on
    // the host (0.4 ms) happens while the device is still in the lifts.  (Never false for honest inputs; if it is, what was prepared
    // for the one-ladder response is dropped.)
    bool root_checked = false;
    if (sb_root && root_ready) {
      HIPCHK(hipEventSynchronize(root_ready));
      sb_units = all_units_end(mn, sb_root);
      root_checked = true;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    fetched();
    HT("alpha known");
    if (by_struct) {
      bool nonunit = false;
      for (size_t g = 0; g < S; ++g) nonunit = nonunit || hst_stmt[g] || hst_num[g];
      for (size_t i = 0; i < batch; ++i) nonunit = nonunit || hst_num[nbs + i];
      if (nonunit) {      // a ct1, b or y that is not a unit: the structure theorem does not apply -- the reference's formula verbatim
        literal_ladders();
        launch_equal(san, c2s, W3, nbs, S, d_ok, ctx->stream);
        HIPCHK(hipMemcpyAsync(hok.data(), d_ok, S * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (hash_early) {                                  // (Alpha changed: hash it again)
          hash_bits();
          HIPCHK(hipMemcpyAsync(hch.data(), chal, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
      }
    }
    for (size_t i = 0; i < S; ++i)
      if (!hok[i]) api_throw(PGPU_ERR_INVALID, "cannot prove re-encryption because inputs are wrong");
    // ---- alpha = ct1^(x^n) * y^(n^2) mod n^3 (ddleq.go:81-87); with CRT it came out of the launch above
    uint32_t* yn2 = ctx->ws_t<uint32_t>((size_t)W3 * nb);
    if (!crt3) {
      shared3(yl, W1, N2, nb, yn2);
      perlane3(c1, xn, W2, nb, t3);
      modmul_arrays(ctx, mn3, t3, yn2, nb, alp);
    }
    if (!hash_early) {
      hash_bits();
      HIPCHK(hipMemcpyAsync(hch.data(), chal, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    HT("flags checked");
    side.join();                                             // the per-statement values are needed from here on
    if (sb_root) {
      if (!root_checked) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        sb_units = all_units_end(mn, sb_root);
      }
      if (!sb_units) early = resp_struct = resp_late = false;
    }
    HT("hash known");
  }

  // the instances with challenge bit 1 (ddleq.go:93-115): e = x a^-1, f = y c
  void respond() {
    // default outputs: e = x, f = y (chalBit false)
    eo = zext(ctx, xl, W1, W2, nb);
    fo = zext(ctx, yl, W1, W3, nb);
    // instances with challenge bit 1 and the statement each belongs to
    std::vector<uint32_t> idx, sti;
    for (size_t i = 0; i < batch; ++i)
      if (hch[i]) {
        idx.push_back((uint32_t)i);
        sti.push_back((uint32_t)(i / secpar));
      }
    if (idx.empty()) return;
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
    HT("bits sorted");
    uint32_t* d_idx = ctx->upload_words(idx);
    uint32_t* d_sti = ctx->upload_words(sti);
    HT("idx uploaded");
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
    auto finish = [&](uint32_t* ge, uint32_t* gy, uint32_t* c5) {       // f = y c mod n^3 (ddleq.go:114); E, F of the bit-1 instances into place
      uint32_t* y3 = zext(ctx, gy, W1, W3, nbg);
      uint32_t* gf = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
      modmul_arrays(ctx, mn3, y3, c5, nbg, gf);
      launch_scatter(ge, nbg, d_idx, cnt, eo, nb, W2, ctx->stream);
      launch_scatter(gf, nbg, d_idx, cnt, fo, nb, W3, ctx->stream);
    };
    if (resp_struct || resp_late) {
      // through the structure of the unit group: gather the prepared exponents of the bit-1 instances; one ladder modulo the primes, one lift
      uint32_t* ge = gat(ge_all, W2);
      const uint32_t* e1p[2] = {gat(rexps.e1p[0], sk->eo1_p.w), gat(rexps.e1p[1], sk->eo1_q.w)};
      const uint32_t* e2p[2] = {gat(rexps.e2p[0], sk->eo1_p.w), gat(rexps.e2p[1], sk->eo1_q.w)};
      int32_t* d_st_r = ctx->ws_t<int32_t>(nbg);
      HIPCHK(hipMemsetAsync(d_st_r, 0, nbg * 4, ctx->stream));
      std::vector<uint32_t> row(sti);                                    // row of each instance's statement in rbase
      size_t n_rows = S, nbu_late = 0;
      const uint32_t* b_late = nullptr;
      hipEvent_t late_inputs = nullptr;
      // (late: the decryption of b runs on every compute unit the ladder modulo the primes leaves, until the lift is under way: a link of
      // the main stream's chains that asked for an EMPTY compute unit would wait for its end -- 9 ms, measured)
      struct NoShortExclusive {
        pgpu_ctx* c; bool was;
        NoShortExclusive(pgpu_ctx* c_, bool on) : c(c_), was(c_->use_exclusive_short) { if (on) c->use_exclusive_short = false; }
        ~NoShortExclusive() { c->use_exclusive_short = was; }
      } no_short(ctx, resp_late);
      if (resp_late) {
        // the statements that have an instance with bit 1, in order (sti is sorted): their s, b gathered, b decrypted NOW
        std::vector<uint32_t> used;
        for (size_t i = 0; i < cnt; ++i) {
          if (used.empty() || used.back() != sti[i]) used.push_back(sti[i]);
          row[i] = (uint32_t)(used.size() - 1);
        }
        n_rows = used.size();
        const size_t nbu = round_up(n_rows, VM_BLOCK);
        const uint32_t* d_used = ctx->upload_words(used);
        uint32_t *s_u = ctx->ws_t<uint32_t>((size_t)W1 * nbu), *b_u = ctx->ws_t<uint32_t>((size_t)W1 * nbu);
        launch_gather(qs, nbs, d_used, n_rows, s_u, nbu, W1, ctx->stream);
        launch_gather(bl, nbs, d_used, n_rows, b_u, nbu, W1, ctx->stream);
        // (padding rows: 1, a unit with plaintext 0)
        launch_restride(s_u, nbu, n_rows, mn.d_consts + (size_t)C_ONE * W1, s_u, nbu, W1, ctx->stream);
        launch_restride(b_u, nbu, n_rows, mn.d_consts + (size_t)C_ONE * W1, b_u, nbu, W1, ctx->stream);
        d_st_rstmt = ctx->ws_t<int32_t>(nbu);
        HIPCHK(hipMemsetAsync(d_st_rstmt, 0, nbu * 4, ctx->stream));
        resp_base_residues(sk, s_u, b_u, nbu, rbase);
        b_late = b_u;
        nbu_late = nbu;
        late_inputs = side.mark();                                       // b is gathered: all the decryption waits for
        // b's plaintext on a lane of its own, beside the ladder modulo the primes of struct_response (which needs the residues only;
        // its closed form waits for the plaintext)
      }
      row.resize(nbg, 0);                                                // (padding lanes: row 0's bases)
      uint32_t* c5 = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
      std::function<hipEvent_t()> beside;
      if (resp_late)
        beside = [&]() -> hipEvent_t {
          // b's plaintext on the side stream (idle since read_back joined it), beside the ladder modulo the primes
          hipEvent_t ready = nullptr;
          side.enter(late_inputs);
          {
            ForceLds one_per_cu(ctx, plan::kLateDecryptionLds);
            resp_base_plaintext(sk, b_late, nbu_late, n_rows, d_st_rstmt, rbase);
          }
          if (side.on) {
            ready = ctx->next_sync_ev();
            HIPCHK(hipEventRecord(ready, ctx->stream));
          }
          side.leave();
          HT("resp_base issued");
          return ready;
        };
      struct_response(sk, rbase, ctx->upload_words(row), gat(en_all, W2), e1p, e2p, nbg, c5, d_st_r, beside);
      side.join();
      HT("response issued");
      std::vector<int32_t> hr(cnt), hs(n_rows);
      HIPCHK(hipMemcpyAsync(hr.data(), d_st_r, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipMemcpyAsync(hs.data(), d_st_rstmt, n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      bool nonunit = false;
      for (size_t i = 0; i < cnt; ++i) nonunit = nonunit || hr[i] || hs[row[i]];
      if (!nonunit) {
        finish(ge, gat(yl, W1), c5);
        return;
      }
    }
    if (early) {
      // everything but the ladder itself is at hand (side stream, above): gather it for the instances with bit 1
      uint32_t* ge = gat(ge_all, W2);
      const uint32_t* es[2] = {gat(es_all[0], sk->eo_p.w), gat(es_all[1], sk->eo_q.w)};
      const uint32_t* eb[2] = {gat(eb_all[0], sk->eo_p.w), gat(eb_all[1], sk->eo_q.w)};
      pre.sti = d_sti;
      pre.cnt = cnt;
      uint32_t* c5 = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
      HT("gathers issued");
      if (!pow_n3_crt_two(sk, nullptr, es, nullptr, eb, nbg, c5, &pre)) api_throw(PGPU_ERR_UNSUPPORTED, "internal: the early response path lost its kernel");
      HT("response issued");
      finish(ge, gat(yl, W1), c5);
      return;
    }
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
    if (one_ladder && sb_units) {
      // c = ((s^an b)^en)^-1 s^xn = s^(xn - an en) b^(-en)  (ddleq.go:107-112) whenever s and b are units: ONE interleaved
      // ladder with two per-number exponents, computed modulo the group orders of p^3 and q^3 (response_exponents), instead of the
      // ladders s^an | s^xn, (.)^en and a batch inversion modulo n^3.  A non-unit s or b (the reference's ModInverse is then
      // undefined) keeps the literal sequence below and its error.  (The unit test ran per statement, on the side stream.)
      const uint32_t *es[2], *eb[2];
      response_exponents(gxn, gan, en, nbg, es, eb);
      uint32_t* b3n = zext(ctx, gb, W1, W3, nbg);
      uint32_t* o = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
      if (pow_n3_crt_two(sk, s3, es, b3n, eb, nbg, o)) c5 = o;
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
      uint32_t* b3 = zext(ctx, gb, W1, W3, nbg);
      uint32_t* cb = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
      modmul_arrays(ctx, mn3, cc, b3, nbg, cb);
      perlane3(cb, en, W2, nbg, cc);
      launch_restride(cc, nbg, cnt, mn3.d_consts + (size_t)C_ONE * W3, cc, nbg, W3, ctx->stream);
      uint32_t* ci = batch_inverse(ctx, mn3, cc, nbg, cnt);
      c5 = ctx->ws_t<uint32_t>((size_t)W3 * nbg);
      modmul_arrays(ctx, mn3, ci, sx, nbg, c5);
    }
    finish(ge, gy, c5);
  }

  void write_outputs() {
    pack_result(ctx, alp, W3, nb, batch, alpha, ct_stride, mn3.nbytes, mem);
    pack_result(ctx, eo, W2, nb, batch, e_out, e_stride, std::min(e_stride, mn2.nbytes), mem);
    pack_result(ctx, fo, W3, nb, batch, f_out, ct_stride, mn3.nbytes, mem);
    HT("outputs issued");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HT("done");
  }

  void run() {
    unpack_inputs();
    powers_of_n();
    structure_base();
    extract_randomness();
    sanity_and_alpha();
    statement_side_work();
    read_back();
    respond();
    write_outputs();
  }
};

}  // namespace

extern "C" {

static void ddleq_prove_impl(const pgpu_seckey* sk, size_t S, size_t secpar, const uint8_t* ct1, const uint8_t* ct2, size_t ct_stride,
                             const uint8_t* a, const uint8_t* b, const uint8_t* x, const uint8_t* y, size_t n_stride, uint8_t* alpha,
                             uint8_t* e_out, size_t e_stride, uint8_t* f_out, int mem) {
  pgpu_ctx* ctx = sk->ctx;
  const pgpu_pubkey* pk = sk->pk;
  if (S == 0 || secpar == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
  if (!pk->mn3 || sk->c_mu2R < 0) api_throw(PGPU_ERR_UNSUPPORTED, "level two is not available for this key");
  if (!pk->g_is_n_plus_1) api_throw(PGPU_ERR_UNSUPPORTED, "DDLEQ prover assumes G = N+1");
  ctx->bind();
  ctx->reset_ws();
  ProveCall call(sk, S, secpar, ct1, ct2, ct_stride, a, b, x, y, n_stride, alpha, e_out, e_stride, f_out, mem);
  call.run();
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
