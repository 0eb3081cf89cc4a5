// modexp.cpp -- the ladders every protocol is made of: x^e modulo N on the kernel family that fits (generic, pair form for
// moduli with a known root, three-digit form), reduction, batch inversion; gmp.Int.Exp / Mul+Mod / ModInverse at the ABI.
#include "engine.hpp"

namespace pgi {

// out[i] = base[i]^e mod N on an already-unpacked base array (slot layout described inline).
// Returns the device array of canonical results (WT limbs, limb-major).
// base_wide: the base occupies 2*WT limbs (slots 0 and 1).  post: optional plain multiplicand array.
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
                 bool use_post, int lanes, uint32_t** raw_out) {
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
  if ((lanes == 8 || lanes == 16) && !exps) {
    // eight / sixteen lanes per number (GenQ8 / GenQ16, shared exponents only): slots of 2 x 76 / 2 x 80 limbs of their own; the digits
    // are zero-extended, change radix R_74 -> R_76 / R_80 with the first product of the program and come back with its last
    const int H8 = lanes == 16 ? pi.h16 : pi.h8;
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
    sp.pair = lanes == 16 ? pi.consts16 : pi.consts8; sp.pair_n0inv = mn.n0inv; sp.pair_h = H8; sp.pair_lanes = lanes;
    sp.tconsts = lanes == 16 ? pi.tconsts16 : pi.tconsts8;
    run_vm(ctx, nb, sp, nullptr, true);
    launch_restride(m8 + 3 * SW8, nb, nb, nullptr, mem + 3 * SW, nb, H, ctx->stream);
    launch_restride(m8 + 3 * SW8 + (size_t)H8 * nb, nb, nb, nullptr, mem + 3 * SW + S1, nb, H, ctx->stream);
  } else {
    Prog p;
    // per-number windows: the table number-major (GenQ / GenQ4 gather a lane's limbs as contiguous bytes)
    if (exps) emit_modexp_perlane(p, we, 2, NO_SLOT, 2, 3, 5, NO_SLOT, pi.c_one_pair, 4, ctx->use_nm4 && plan::pair_nm4_fits(nb, W2));
    else emit_modexp_shared(p, *e, 2, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    p.end();
    SegSpec sp{&mc, &p, mem, exps};
    sp.pair = pi.consts; sp.pair_n0inv = mn.n0inv; sp.pair_h = H; sp.pair_lanes = lanes >= 8 ? 4 : lanes;
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
// per-number exponents (28-bit limbs, [we][nb]) -> the 25-bit words VM_MULV5 reads (5 windows of 5 bits each)
const uint32_t* windows5_of(pgpu_ctx* ctx, const uint32_t* exps, int we, size_t nb) {
  const int we5 = (we * LB + 24) / 25;
  uint32_t* out = ctx->ws_t<uint32_t>((size_t)we5 * nb);
  launch_repack_windows5(exps, we, out, we5, nb, ctx->stream);
  return out;
}

// window bits for per-number exponents on the three-digit kernel: 7 while the 128-entry table stays within the kernel's
// 32-bit gather offsets (nb <= 32768 at 2048-bit keys), else 5
const uint32_t* triple_windows(pgpu_ctx* ctx, const uint32_t* exps, int we, size_t nb, int wb) {
  return wb == 5 ? windows5_of(ctx, exps, we, nb) : exps;
}

bool triple_usable(pgpu_ctx* ctx, const ModCtx& mc, bool allow6) {
  return mc.triple.root && ctx->use_asm && ctx->use_pair && ctx->use_triple && (allow6 || !mc.triple.lanes6_only);
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

bool triple12_available(pgpu_ctx* ctx, const ModCtx& mc) {
  const TripleInfo& ti = mc.triple;
  return ti.h12 > 0 && ctx->use_asm && ctx->use_lanes8 && ctx->use_lanes16 && vm_asm_available(ti.h12 / 4, 160);
}
TriplePlan triple_alloc12(pgpu_ctx* ctx, const ModCtx& mc, size_t nb, int slots) {
  TriplePlan t;
  t.H = mc.triple.h12;
  t.nb = nb;
  t.slot_words = (size_t)3 * t.H * nb;
  t.mem = ctx->ws_t<uint32_t>(t.slot_words * (size_t)slots);
  return t;
}
void triple_widen12(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* digits, uint32_t* digits12, size_t nb) {
  const int H = mc.triple.root->WT, H12 = mc.triple.h12;
  const size_t S = (size_t)H * nb, S12 = (size_t)H12 * nb;
  HIPCHK(hipMemsetAsync(digits12, 0, 3 * S12 * 4, ctx->stream));
  for (int d = 0; d < 3; ++d) launch_restride(digits + (size_t)d * S, nb, nb, nullptr, digits12 + (size_t)d * S12, nb, H, ctx->stream);
}
void triple_narrow12(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* digits12, uint32_t* digits, size_t nb) {
  const int H = mc.triple.root->WT, H12 = mc.triple.h12;
  const size_t S = (size_t)H * nb, S12 = (size_t)H12 * nb;
  for (int d = 0; d < 3; ++d) launch_restride(digits12 + (size_t)d * S12, nb, nb, nullptr, digits + (size_t)d * S, nb, H, ctx->stream);
}
void triple_run12(pgpu_ctx* ctx, const ModCtx& mc, const TriplePlan& t12, const Prog& p, const uint32_t* exps) {
  const TripleInfo& ti = mc.triple;
  SegSpec sp{&mc, &p, t12.mem, exps};
  sp.pair = ti.kconsts12; sp.pair_n0inv = ti.root->n0inv; sp.pair_h = ti.h12; sp.pair_lanes = 12; sp.tconsts = ti.tconsts12;
  run_vm(ctx, t12.nb, sp, nullptr, true);
}

void triple_run(pgpu_ctx* ctx, const ModCtx& mc, const TriplePlan& tp, const Prog& p, const uint32_t* exps) {
  const TripleInfo& ti = mc.triple;
  SegSpec sp{&mc, &p, tp.mem, exps};
  // two lanes per digit (GenQ6) where the digit does not fit a lane, or where the batch is so small that eight lanes per number
  // still leave every wave a SIMD of its own (one ladder's latency is the run time); not for number-major tables
  const bool six = ti.lanes6_only || (ctx->use_lanes8 && !p.nm_tables && tp.H % 2 == 0 && vm_asm_available(tp.H / 2, 112) &&
                                      plan::triple_two_lanes_per_digit(tp.nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus)));
  sp.pair = ti.kconsts; sp.pair_n0inv = ti.root->n0inv; sp.pair_h = tp.H; sp.pair_lanes = six ? 6 : 3; sp.tconsts = ti.tconsts;
  run_vm(ctx, tp.nb, sp, nullptr, true);
}

// pl.in() (canonical, < n^3) ^ e [* pl.post()] mod n^3 on the three-digit kernel; canonical result in pl.out()
void modexp_triple(pgpu_ctx* ctx, const ModCtx& mc, const ModexpPlan& pl, const BigU* e, const uint32_t* exps, int we,
                   bool use_post, const uint32_t* pair_digits) {
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
  if (!exps && triple12_available(ctx, mc) &&
      plan::triple_four_lanes_per_digit(pl.nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), true, true, 1)) {
    // a lone shared-exponent ladder of a batch that leaves most SIMDs empty (level-two Encrypt's (r^n)^n of up to 4 096 numbers): four
    // lanes per digit -- slots of 3 x 76 limbs of their own, radix R_H -> R_76 with the first product of the program, back with its last
    TriplePlan t12 = triple_alloc12(ctx, mc, pl.nb, 5 + 32);
    triple_widen12(ctx, mc, tp.slot(0), t12.slot(0), pl.nb);
    p.op(VM_LOAD, 0); p.op(VM_MULC, 1); p.op(VM_STORE, 0);
    emit_modexp_shared(p, *e, 0, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    p.op(VM_LOAD, 3); p.op(VM_MULC, 2); p.op(VM_STORE, 3);
    p.end();
    triple_run12(ctx, mc, t12, p, nullptr);
    triple_narrow12(ctx, mc, t12.slot(3), tp.slot(3), pl.nb);
    triple_exit(ctx, mc, tp, 3, pl.out(), use_post ? pl.post() : nullptr);
    return;
  }
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
                       bool skip_zero, uint32_t** raw_pair_out, const uint32_t* pair_digits_in) {
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
  const bool have4 = mc.pairn.root && mc.pairn.root->WT % 2 == 0 && vm_asm_available(mc.pairn.root->WT / 2, 64);
  const int lanes = plan::pair_lanes_shared(pl.nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), have4, have4 && mc.pairn.consts8 && ctx->use_lanes8,
                                            mc.pairn.consts16 && ctx->use_lanes16);
  if (mc.pairn.root && ctx->use_asm && ctx->use_pair && !wide && skip_zero && e.bit_length() >= 256 && plan::pair_kernel_serves(pl.nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), have4)) {
    // (a batch that leaves SIMDs empty even at four lanes per number is bound by one ladder's latency: eight lanes, GenQ8)
    modexp_pair(ctx, mc, pl, &e, nullptr, 0, use_post, lanes, use_post ? nullptr : raw_pair_out);
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
  if (triple_usable(ctx, mc, true) && !wide && we >= 10 && plan::triple_batch_fits(pl.nb, mc.WT)) {
    modexp_triple(ctx, mc, pl, nullptr, exps, we, use_post);
    return;
  }
  {
    const bool have4 = mc.pairn.root && mc.pairn.root->WT % 2 == 0 && vm_asm_available(mc.pairn.root->WT / 2, 64);
    if (mc.pairn.root && mc.pairn.c_one_pair >= 0 && ctx->use_asm && ctx->use_pair && !wide && we >= 10 &&
        plan::pair_mulv_fits(pl.nb, mc.WT) &&                            // MULV gathers with 32-bit offsets
        plan::pair_kernel_serves(pl.nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), have4)) {
      modexp_pair(ctx, mc, pl, nullptr, exps, we, use_post, plan::pair_lanes_2or4(pl.nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), have4));
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

// x mod N for an array of ANY width: Horner over WT-limb chunks from the top, every step one reduce_mod of [chunk | remainder]
void reduce_mod_wide(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* in, int w_in, uint32_t* out, size_t nb) {
  const int WT = mc.WT;
  if (w_in <= 2 * WT) { reduce_mod(ctx, mc, in, w_in, out, nb); return; }
  const size_t sw = (size_t)WT * nb;
  const int nchunks = (w_in + WT - 1) / WT;
  uint32_t* arr = ctx->ws_t<uint32_t>(2 * sw);   // [chunk | running remainder]: the value chunk + rem * 2^(28 WT)
  launch_copy_limbs(in, (nchunks - 1) * WT, w_in - (nchunks - 1) * WT, arr + sw, WT, nb, ctx->stream);
  for (int k = nchunks - 2; k >= 0; --k) {
    launch_copy_limbs(in, k * WT, WT, arr, WT, nb, ctx->stream);
    reduce_mod(ctx, mc, arr, 2 * WT, k ? arr + sw : out, nb);
  }
}

// Stage an operand that is read modulo N, whatever its stride.  Up to the width of the modulus the bytes go straight into
// WT limbs (a Montgomery operand may be any value below R; `canonical` additionally reduces it below N).  A wider stride
// is unpacked whole and reduced chunk by chunk (Horner over WT-limb chunks), so that no leading byte is silently
// dropped: the reference's Exp / Mul+Mod reduce any value correctly.
void unpack_mod(pgpu_ctx* ctx, const ModCtx& mc, const uint8_t* buf, size_t stride, size_t count, int mem, uint32_t* out,
                size_t nb, bool canonical) {
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

const uint8_t* all_units_begin(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count) {
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
  uint8_t* rb = (uint8_t*)ctx->pinned(mc.nbytes);
  uint8_t* d_rb = (uint8_t*)ctx->ws(mc.nbytes);
  launch_pack_be(mem, WT, nbt, 1, d_rb, mc.nbytes, mc.nbytes, ctx->stream);
  HIPCHK(hipMemcpyAsync(rb, d_rb, mc.nbytes, hipMemcpyDeviceToHost, ctx->stream));
  return rb;
}
bool all_units_end(const ModCtx& mc, const uint8_t* root_be) {
  BigU root = BigU::from_be(root_be, mc.nbytes), rinv;
  return hostbig::modinv(root, mc.N, rinv);
}
bool all_units(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count) {
  const uint8_t* rb = all_units_begin(ctx, mc, x, nb, count);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return all_units_end(mc, rb);
}

// gmp.Int.ModInverse for a batch.  d_bad (device int32[nb], may be null) receives 1 on the lanes that are not units and 0
// elsewhere; those lanes get the result 0 (mpz_invert leaves its result undefined there and the reference never checks).
// One hostile element must not cost the honest ones their answers: when the tree's root cannot be inverted, a per-lane
// binary GCD finds the non-units, 1 is substituted for them and the tree runs again.  With d_bad == nullptr a non-unit
// throws PGPU_ERR_NOT_INVERTIBLE (callers for which a non-unit means the whole call is meaningless).
uint32_t* batch_inverse(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, size_t nb, size_t count, int32_t* d_bad,
                        bool* any_bad) {
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
void check_batch_args(const void* a, const void* b, size_t batch) {
  if (!a || !b) api_throw(PGPU_ERR_INVALID, "null buffer");
  if (batch == 0) api_throw(PGPU_ERR_INVALID, "empty batch");
  if (batch > (1u << 26)) api_throw(PGPU_ERR_INVALID, "batch too large");
}

}  // namespace pgi

namespace pgi {

// Pair-form entry / exit shared by the multi-share forms of PartialDecrypt (N = n^2, root n public).
// entry: canonical residues x (slot 0 of `ent`, 4 slots of mc.WT limbs, stride nb) -> digits X0 | X1 of x R_H in slot 2
void pair_enter(pgpu_ctx* ctx, const ModCtx& mc, uint32_t* ent, size_t nb) {
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
uint32_t* pair_leave(pgpu_ctx* ctx, const ModCtx& mc, uint32_t* pm, uint32_t out_slot, size_t nb) {
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
void pair_leave_and_pack(pgpu_ctx* ctx, const ModCtx& mc, uint32_t* pm, uint32_t out_slot, size_t nb, size_t count, uint8_t* dst,
                                size_t out_stride, int mem) {
  pack_result(ctx, pair_leave(ctx, mc, pm, out_slot, nb), mc.WT, nb, count, dst, out_stride, mc.nbytes, mem);
}

// x^(per-number exponent, `we` limbs) * y^(shared exponent e) modulo N = n^2 as ONE interleaved ladder on the pair kernels
// (4-bit windows of the per-number exponent, sliding windows of e).  x, y: canonical residues (mc.WT limbs, stride nb).
// Returns the canonical result, or nullptr when the pair kernels do not serve this key / batch.
uint32_t* dual_pow_pair(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* x, const uint32_t* exps, int we, const uint32_t* y,
                               const BigU& e, size_t nb, uint32_t** raw_out, bool y_ready) {
  // raw_out: the result stays in pair form (a0 | a1, stride nb): *raw_out and the return value point at its digits
  const PairInfo& pi = mc.pairn;
  if (!(pi.root && pi.c_one_pair >= 0 && ctx->use_asm && ctx->use_pair)) return nullptr;
  const int H = pi.root->WT, W2 = mc.WT;
  const bool have4 = H % 2 == 0 && vm_asm_available(H / 2, 64);
  if (!plan::pair_kernel_serves(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), have4)) return nullptr;
  const bool two = plan::pair_lanes_2or4(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), have4) == 2;
  // A batch so small that eight lanes per number still leave every wave a SIMD of its own is bound by the ladder's latency: GenQ8
  // (76-limb digits over four lanes each; the radix changes R_74 <-> R_76 by one product on the way in and out, inside the program;
  // limb-major 5-bit tables -- the gathers of so few numbers are not what the launch waits for).  2 048 numbers: 36.6 -> 2x ms.
  // (and sixteen -- GenQ16, 80-limb digits over eight lanes each -- for the smallest batches: plan::pair_lanes_shared)
  const int wide = (have4 && pi.consts8 && ctx->use_lanes8)
                       ? plan::pair_lanes_shared(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), have4, true, pi.consts16 && ctx->use_lanes16) : 0;
  const bool eight = wide == 8 || wide == 16;
  // per-number window table number-major (VM_STORET / VM_MULVT5 / VM_MULVT): limb-major, the 16 384-number ladder of the DDLEQ
  // verifier fetched 98 GB of 32-byte sectors for its dword gathers in a 43 ms launch (profiles/r03_bench_traffic.txt)
  const bool nm4 = ctx->use_nm4 && !eight;
  const int Hk = wide == 16 ? pi.h16 : eight ? pi.h8 : H;                  // limbs of a digit in the kernel's slots
  const int wb = plan::dual_pair_window_bits(nb, 2 * Hk, nm4);             // gathers with 32-bit offsets
  if (!wb) return nullptr;
  const uint32_t tab2 = 5 + (uint32_t)perlane_table_slots(wb, nm4);
  const size_t SW = (size_t)W2 * nb, SWk = (size_t)2 * Hk * nb;
  uint32_t* pm = ctx->ws_t<uint32_t>(SWk * (size_t)(tab2 + 32));          // 0 x, 1 y, 2 tmp, 3 out, 5.. / tab2.. the tables
  if (eight) HIPCHK(hipMemsetAsync(pm, 0, 2 * SWk * 4, ctx->stream));
  Fork fk(ctx);                                                            // y's entry chain beside x's
  for (int k = 0; k < 2; ++k) {
    fk.chain(k);
    uint32_t* ent = ctx->ws_t<uint32_t>(SW * 4);
    HIPCHK(hipMemcpyAsync(ent, k ? y : x, SW * 4, hipMemcpyDeviceToDevice, ctx->stream));
    pair_enter(ctx, mc, ent, nb);
    if (eight) {                                                           // digits of 74 limbs -> 76 (zero-extended)
      launch_restride(ent + 2 * SW, nb, nb, nullptr, pm + (size_t)k * SWk, nb, H, ctx->stream);
      launch_restride(ent + 2 * SW + (size_t)H * nb, nb, nb, nullptr, pm + (size_t)k * SWk + (size_t)Hk * nb, nb, H, ctx->stream);
    } else {
      HIPCHK(hipMemcpyAsync(pm + (size_t)k * SW, ent + 2 * SW, SW * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  fk.join();
  Prog pd;
  if (eight)
    for (uint32_t k = 0; k < 2; ++k) { pd.op(VM_LOAD, k); pd.op(VM_MULC, 0); pd.op(VM_STORE, k); }          // radix R_74 -> R_76
  if (y_ready) {
    // slot 1 already holds the power of y (in pair form): x^(its own exponent), then one product
    emit_modexp_perlane(pd, we, 0, NO_SLOT, 2, 3, 5, NO_SLOT, eight ? 2 : pi.c_one_pair, wb, nm4);
    pd.op(VM_LOAD, 3); pd.op(VM_MUL, 1); pd.op(VM_STORE, 3);
  } else {
    emit_modexp_dual(pd, we, e, 0, 1, 2, 3, 5, tab2, eight ? 2 : pi.c_one_pair, wb, nm4);
  }
  if (eight) { pd.op(VM_MULC, 1); pd.op(VM_STORE, 3); }                                                      // radix R_76 -> R_74
  pd.end();
  SegSpec sp{&mc, &pd, pm, wb == 5 ? windows5_of(ctx, exps, we, nb) : exps};
  sp.pair = wide == 16 ? pi.consts16 : eight ? pi.consts8 : pi.consts; sp.pair_n0inv = pi.root->n0inv; sp.pair_h = Hk; sp.pair_lanes = eight ? wide : two ? 2 : 4;
  if (eight) sp.tconsts = wide == 16 ? pi.tconsts16 : pi.tconsts8;
  run_vm(ctx, nb, sp, nullptr, true);
  if (eight) {
    uint32_t* back = ctx->ws_t<uint32_t>(SW * 4);                          // (pair_leave wants 74-limb digits and two scratch slots)
    const uint32_t* res = pm + 3 * SWk;
    launch_restride(res, nb, nb, nullptr, back, nb, H, ctx->stream);
    launch_restride(res + (size_t)Hk * nb, nb, nb, nullptr, back + (size_t)H * nb, nb, H, ctx->stream);
    if (raw_out) return *raw_out = back;
    return pair_leave(ctx, mc, back, 0, nb);
  }
  if (raw_out) return *raw_out = pm + 3 * SW;
  return pair_leave(ctx, mc, pm, 3, nb);
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

// out = base^e mod N for a shared exponent; base: `wb` limbs (<= 2 WT); canonical result
void shared_pow(pgpu_ctx* ctx, const ModCtx& mc, const uint32_t* base, int wb, const BigU& e, size_t nb, uint32_t* out) {
  ModexpPlan pl = modexp_alloc(ctx, mc, nb, 32);
  const bool wide = wb > mc.WT;
  launch_copy_limbs(base, 0, std::min(wb, mc.WT), pl.in(), mc.WT, nb, ctx->stream);
  if (wide) launch_copy_limbs(base, mc.WT, wb - mc.WT, pl.in() + pl.slot_words, mc.WT, nb, ctx->stream);
  modexp_shared_run(ctx, mc, pl, e, wide, false, true);
  HIPCHK(hipMemcpyAsync(out, pl.out(), (size_t)mc.WT * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
}

}  // namespace pgi

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

}  // extern "C"
