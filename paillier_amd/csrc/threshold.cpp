// threshold.cpp -- threshold decryption (thresholdkey.go:63-326): PartialDecrypt in its batch forms, CombinePartialDecryptions,
// the share ZKP.
#include "engine.hpp"

namespace pgi {

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

// x^-1 mod n^2 of a per-key constant (a verification key), computed once on the host and kept with the key; false: not a unit
bool cached_inverse(pgpu_pubkey* pk, const BigU& x, BigU& inv) {
  for (auto& e : pk->inverse_cache)
    if (e.x == x) { inv = e.inv; return e.unit; }
  pgpu_pubkey::CachedInverse e;
  e.x = x;
  e.unit = hostbig::modinv(x, pk->mn2.N, e.inv);
  if (pk->inverse_cache.size() >= 64) pk->inverse_cache.erase(pk->inverse_cache.begin());      // (bounded: callers choose the keys)
  pk->inverse_cache.push_back(e);
  inv = e.inv;
  return e.unit;
}

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

// The 7-bit comb table of `base` modulo n^(level + 2) for exponents of up to `ebits` bits (4 windows per 28-bit limb: 1.75 x fewer
// products than the 4-bit table).  82 000 entries for a 4 480-bit exponent: built on the DEVICE -- the host walks the chain
// b_i = base^(128^i) (7 squarings per window), one lane per window multiplies its 127 entries up, a transpose puts them where
// VM_MULCV7 reads them ([entry][WT], shared by the batch) -- and kept with the key.
const pgpu_pubkey::Comb7& ensure_comb7(pgpu_pubkey* pk, int level, const BigU& base_in, size_t ebits) {
  pgpu_ctx* ctx = pk->ctx;
  ModCtx& mc = level == PGPU_LEVEL_TWO ? *pk->mn3 : pk->mn2;
  const BigU base = base_in % mc.N;
  const int nwin = (int)((ebits + LB - 1) / LB) * 4;
  for (auto& c : pk->comb7)
    if (c->level == level && c->base == base && c->nwin >= nwin) return *c;
  std::unique_ptr<pgpu_pubkey::Comb7> c(new pgpu_pubkey::Comb7());
  c->base = base;
  c->level = level;
  c->nwin = nwin;
  const int WT = mc.WT;
  const size_t nbw = round_up((size_t)nwin, VM_BLOCK);
  // b_i in Montgomery form, limb-major [WT][nbw] (padding lanes: 1)
  std::vector<uint32_t> bl((size_t)WT * nbw, 0);
  {
    const auto one = mc.to_mont(BigU(1)).to_limbs(LB, WT);
    for (size_t i = (size_t)nwin; i < nbw; ++i)
      for (int l = 0; l < WT; ++l) bl[(size_t)l * nbw + i] = one[(size_t)l];
    BigU b = base;
    for (int i = 0; i < nwin; ++i) {
      const auto bm = mc.to_mont(b).to_limbs(LB, WT);
      for (int l = 0; l < WT; ++l) bl[(size_t)l * nbw + (size_t)i] = bm[(size_t)l];
      if (i + 1 < nwin)
        for (int k = 0; k < 7; ++k) b = hostbig::mulmod(b, b, mc.N);
    }
  }
  ctx->bind();
  ctx->reset_ws();
  const size_t sw = (size_t)WT * nbw;
  uint32_t* mem = ctx->ws_t<uint32_t>(sw * 129);           // slots: 0 b_i, 1 + d: b_i^d
  HIPCHK(hipMemcpyAsync(mem, bl.data(), sw * 4, hipMemcpyHostToDevice, ctx->stream));
  {
    Prog p;
    p.op(VM_LOADC, C_ONE_M);
    p.op(VM_STORE, 1);
    for (uint32_t d = 1; d < 128; ++d) { p.op(VM_MUL, 0); p.op(VM_STORE, 1 + d); }
    p.end();
    SegSpec sg{&mc, &p, mem, nullptr};
    run_vm(ctx, nbw, sg, nullptr, false);
  }
  for (uint32_t d = 0; d < 128; ++d) launch_canon(mem + (size_t)(1 + d) * sw, mc.d_nmod, WT, nbw, ctx->stream);
  c->words = ((size_t)kComb7First + (size_t)nwin * 128) * (size_t)WT;
  HIPCHK(hipMalloc((void**)&c->d_table, c->words * 4));
  HIPCHK(hipMemcpyAsync(c->d_table, mc.d_consts, (size_t)kComb7First * WT * 4, hipMemcpyDeviceToDevice, ctx->stream));   // C_R2, C_R3, C_ONE_M, C_ONE
  launch_comb7_transpose(mem, nbw, WT, nwin, kComb7First, c->d_table, ctx->stream);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (pk->comb7.size() >= 16) pk->comb7.erase(pk->comb7.begin());     // (bounded: callers choose the bases)
  pk->comb7.push_back(std::move(c));
  return *pk->comb7.back();
}
// x <- x * base^(this number's exponent of `we` limbs): one table product per 7-bit window (the segment's tconsts = the Comb7 buffer)
void emit_comb7(Prog& p, int we) {
  for (int i = 0; i < we * 4; ++i) p.op(VM_MULCV7, (uint32_t)i, kComb7First);
}
// base^e mod n^2 through the 7-bit comb table; exps: limb-major [we][nb]; result canonical in `out`
void comb_pow7(pgpu_ctx* ctx, const ModCtx& mc, const pgpu_pubkey::Comb7& cb, const uint32_t* exps, int we, size_t nb, uint32_t* out) {
  if (we * 4 > cb.nwin) api_throw(PGPU_ERR_INVALID, "fixed-base table narrower than the exponent");
  size_t sw = (size_t)mc.WT * nb;
  uint32_t* memv = ctx->ws_t<uint32_t>(sw);
  Prog p;
  p.op(VM_LOADC, C_ONE_M);
  emit_comb7(p, we);
  p.op(VM_MULC, C_ONE);
  p.op(VM_STORE, 0);
  p.end();
  SegSpec sg{&mc, &p, memv, exps};
  sg.tconsts = cb.d_table;
  run_vm(ctx, nb, sg, nullptr, true);
  launch_canon(memv, mc.d_nmod, mc.WT, nb, ctx->stream);
  HIPCHK(hipMemcpyAsync(out, memv, sw * 4, hipMemcpyDeviceToDevice, ctx->stream));
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

// Shared part of prove and verify: the Fiat-Shamir hash E = SHA-256(a || b || c^4 || c_i^2) over UNREDUCED c^4, c_i^2
// (thresholdkey.go:241,248,319-326).  c, ci: canonical W2-limb arrays as given by the caller.  Returns digest words [8][nb].
uint32_t* zkp_hash(pgpu_ctx* ctx, int W2, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* ci, size_t nb,
                   size_t count) {
  uint32_t* c2 = ctx->ws_t<uint32_t>((size_t)2 * W2 * nb);
  uint32_t* c4 = ctx->ws_t<uint32_t>((size_t)4 * W2 * nb);
  uint32_t* ci2 = ctx->ws_t<uint32_t>((size_t)2 * W2 * nb);
  uint32_t* slo = ctx->ws_t<uint32_t>((size_t)4 * W2 * nb);                    // column sums of the widest product: limbs and carries
  uint64_t* scy = ctx->ws_t<uint64_t>((size_t)4 * W2 * nb);
  launch_mul_plain(c, W2, c, W2, c2, nb, slo, scy, ctx->stream);
  launch_mul_plain(c2, 2 * W2, c2, 2 * W2, c4, nb, slo, scy, ctx->stream);
  launch_mul_plain(ci, W2, ci, W2, ci2, nb, slo, scy, ctx->stream);
  const uint32_t* parts[4] = {a, b, c4, ci2};
  const int widths[4] = {W2, W2, 4 * W2, 2 * W2};
  uint32_t* dg = ctx->ws_t<uint32_t>(8 * nb);
  launch_sha256_transcript(parts, widths, 4, nb, count, dg, nullptr, ctx->stream);
  return dg;
}

}  // namespace pgi

extern "C" {

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
    const size_t lanes_target = plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus);
    const bool have4 = H % 2 == 0 && vm_asm_available(H / 2, 64);
    auto leave_pair_form = [&](uint32_t* pm, uint32_t out_slot, uint8_t* dst) {
      pair_leave_and_pack(ctx, mc, pm, out_slot, nb, batch, dst, out_stride, mem);
    };
    if (n_shares >= 2 && ctx->use_shared_chain && pi.c_one_pair >= 0 && plan::shared_chain_pays(nb, lanes_target)) {
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
        const int lanes = plan::pair_lanes_2or4(nb, lanes_target, have4);
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
      const int lanes = plan::pair_lanes_2or4((size_t)segs * nb, lanes_target, have4);
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
    // One ciphertext range under several shares (a ciphertext-major shard): a single chain would carry every share's window products
    // on its critical path; where the chip has room the shares split into groups with a chain each (plan::shared_chain_groups)
    if (ok && ivs.size() == 1 && ivs[0].servers.size() >= 2) {
      const PairInfo& pi0 = mc.pairn;
      const int H0 = pi0.root->WT;
      const bool have8 = H0 % 2 == 0 && vm_asm_available(H0 / 2, 64) && pi0.consts8 && ctx->use_lanes8;
      const Interval whole = ivs[0];
      const int groups = plan::shared_chain_groups(round_up(whole.e - whole.b, VM_BLOCK), (int)whole.servers.size(),
                                                   plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), have8);
      if (groups > 1) {
        ivs.clear();
        const size_t S = whole.servers.size();
        for (int g = 0; g < groups; ++g) {                 // S = 3 in two groups: {s0}, {s1, s2}
          Interval iv{whole.b, whole.e, {}};
          for (size_t j = S * (size_t)g / groups; j < S * (size_t)(g + 1) / groups; ++j) iv.servers.push_back(whole.servers[j]);
          ivs.push_back(iv);
        }
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
    const bool have4 = H % 2 == 0 && vm_asm_available(H / 2, 64);
    // A shard so small that even four lanes per number leave SIMDs empty is bound by the LATENCY of one ladder: eight lanes per
    // number (GenQ8: 76-limb digits in four lanes each, 38 multiplies a row and lane instead of 74) while every wave still has
    // a SIMD of its own.  The digits change radix on the way in and out (R_74 <-> R_76: one product each, inside the program).
    const int lanes = plan::pair_lanes_shared(ivs.size() * nbs, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), have4, have4 && pi.consts8 && ctx->use_lanes8);
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
        const int lanes = plan::pair_lanes_2or4(runs.size() * nbs, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), H % 2 == 0 && vm_asm_available(H / 2, 64));
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
    const pgpu_pubkey::Comb7& fb = ensure_comb7(const_cast<pgpu_pubkey*>(pk), PGPU_LEVEL_ONE, BigU::from_be(vkey_be, vkey_len),
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
    comb_pow7(ctx, mc, fb, rl, wr, nb, b);                      // V^r: one table product per 7 bits of r
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
    const pgpu_pubkey::Comb7& fbV = ensure_comb7(pkm, PGPU_LEVEL_ONE, BigU::from_be(vkey_be, vkey_len), zbits);
    // b = V^Z (v_i^E)^-1 = V^Z (v_i^-1)^E: the verification key v_i is fixed for the batch, so its inverse is taken ONCE, on the host
    // (kept with the key), and the comb table is that of v_i^-1 -- no inversion tree for b.  A v_i that is not a unit has no inverse:
    // mpz_invert leaves the reference's b undefined and every proof under that key is rejected.
    BigU vi_inv;
    const bool vi_unit = cached_inverse(pkm, BigU::from_be(vi_be, vi_len) % mc.N, vi_inv);
    const pgpu_pubkey::FixedBase fbI = ensure_fixed_base(pkm, vi_unit ? vi_inv : BigU(1), 256);
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
    comb_pow7(ctx, mc, fbV, zl, wz, nb, b1);
    comb_pow(ctx, mc, fbI, el, 10, nb, b2);                     // (v_i^-1)^E
    // A proof whose Decryption is not a unit has no inverse: mpz_invert leaves the reference's a2 undefined and the hash
    // comparison fails; here such a lane is rejected (ok = 0) without disturbing the other proofs.
    int32_t* bad_a = ctx->ws_t<int32_t>(nb);
    uint32_t* a2i = batch_inverse(ctx, mc, a2, nb, batch, bad_a);
    uint32_t* av = ctx->ws_t<uint32_t>(sw);
    modmul_arrays(ctx, mc, a1, a2i, nb, av);
    uint32_t* bv = ctx->ws_t<uint32_t>(sw);
    modmul_arrays(ctx, mc, b1, b2, nb, bv);
    uint32_t* dg = zkp_hash(ctx, W2, av, bv, cl, dl, nb, batch);
    uint32_t* e2 = ctx->ws_t<uint32_t>(10 * nb);
    launch_digest_to_limbs(dg, e2, nb, ctx->stream);
    int32_t* d_ok = ctx->ws_t<int32_t>(nb);
    launch_equal(e2, el, 10, nb, batch, d_ok, ctx->stream);
    launch_clear_where(bad_a, batch, d_ok, ctx->stream);
    if (!vi_unit) HIPCHK(hipMemsetAsync(d_ok, 0, nb * 4, ctx->stream));
    HIPCHK(hipMemcpyAsync(ok, d_ok, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  });
}

}  // extern "C"
