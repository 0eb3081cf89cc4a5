// paillier.cpp -- Encrypt / Decrypt / Add / Sub / ConstMult at both levels (paillier.go:185-372, operations.go:11-64), the
// randomness of Encrypt (utils.go:26-49), and the key holder's powers modulo n, n^2 through p and q.
#include <sys/random.h>

#include "engine.hpp"

namespace pgi {

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
  const size_t lanes_target = plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus);
  // one lane per number (GenP) from one wave per SIMD upwards; two lanes per number (GenQ) from there down to one wave
  // per SIMD again; below that the ordinary kernels with their finer slicings
  // two lanes per number while they leave every wave a SIMD of its own (both halves: 4 nb lanes); between half a wave and one
  // wave per SIMD at one lane the two-lane kernel would put two waves on most SIMDs: 16.9 ms against 14.5 ms for 20 480 ...
  // 30 720 ciphertexts (tools/decrypt_lanes_probe.py)
  const int pair_lanes_now = plan::crt_pair_lanes(sk->pair_lanes, sk->pair_small2, nb, lanes_target);
  // (a two-lane digit pass is 2 H^2 multiplies per lane: shorter than any slicing of the 2H-limb kernels for H <= 55, so it
  // also wins when the batch is latency-bound; for H = 74 the 4-lane slicing has the same length and fills the chip better)
  if (sk->has_pair && ctx->use_asm && ctx->use_pair && plan::crt_pair_usable(pair_lanes_now, W1, nb, lanes_target)) {
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
      auto run_halves = [&](const Prog& pp, const Prog& pq, bool profile) {
        SegSpec sp{&mp2, &pp, mem, nullptr}, sq{&mq2, &pq, mem, nullptr};
        sp.pair = sk->pair_p.d; sp.pair_n0inv = mp.n0inv; sp.pair_h = W1; sp.pair_lanes = pair_lanes_now;
        sq.pair = sk->pair_q.d; sq.pair_n0inv = mq.n0inv; sq.pair_h = W1; sq.pair_lanes = pair_lanes_now;
        run_vm(ctx, nb, sp, &sq, profile);
      };
      if (crt_pair8_usable(sk, nb)) {
        // a batch that leaves most of the chip empty: entry and exit as they are (a handful of products on the two-lane kernel), the
        // LADDERS on the eight-lane pair kernel (plan::crt_pair_lanes8) -- the latency of the ladder is the run time of the call
        Prog pe, qe, px, qx;
        entry(pe, sk->c_pk_p2, sk->c_onep_p2, 2);
        pe.end();
        entry(qe, sk->c_pk_q2, sk->c_onep_q2, 36);
        qe.end();
        run_halves(pe, qe, false);
        const BigU es[2] = {sk->p - BigU(1), sk->q - BigU(1)};
        const uint32_t* in8[2] = {mem + 2 * S2, mem + 36 * S2};
        uint32_t* out8[2] = {mem + 3 * S2, mem + 37 * S2};
        crt_pair8_ladders(sk, es, in8, out8, nb);
        px.op(VM_LOAD, 3); px.op(VM_MULC, C_ONE); px.op(VM_STORE, 3);
        px.end();
        qx.op(VM_LOAD, 37); qx.op(VM_MULC, C_ONE); qx.op(VM_STORE, 37);
        qx.end();
        run_halves(px, qx, false);
      } else {
        Prog pp, pq;
        entry(pp, sk->c_pk_p2, sk->c_onep_p2, 2);
        emit_modexp_shared(pp, sk->p - BigU(1), 2, NO_SLOT, 2, 3, 4, NO_SLOT, true, true);
        pp.op(VM_LOAD, 3); pp.op(VM_MULC, C_ONE); pp.op(VM_STORE, 3);
        pp.end();
        entry(pq, sk->c_pk_q2, sk->c_onep_q2, 36);
        emit_modexp_shared(pq, sk->q - BigU(1), 36, NO_SLOT, 36, 37, 38, NO_SLOT, true, true);
        pq.op(VM_LOAD, 37); pq.op(VM_MULC, C_ONE); pq.op(VM_STORE, 37);
        pq.end();
        run_halves(pp, pq, true);
      }
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
PrimeShape prime_shape(const pgpu_seckey* sk, size_t nb, int beside) {
  pgpu_ctx* ctx = sk->ctx;
  PrimeShape ps;
  ps.H = sk->mp.WT;
  const bool sliced = plan::prime_lanes(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), sk->has_sliced_primes, ctx->use_prime_lanes, beside) == 4;
  ps.m[0] = sliced ? &sk->mp_s : &sk->mp;
  ps.m[1] = sliced ? &sk->mq_s : &sk->mq;
  ps.Hs = ps.m[0]->WT;
  return ps;
}
void prime_slot_fill(pgpu_ctx* ctx, const PrimeShape& ps, uint32_t* slot, const uint32_t* src, size_t nb) {
  HIPCHK(hipMemcpyAsync(slot, src, (size_t)ps.H * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
  if (ps.Hs > ps.H) HIPCHK(hipMemsetAsync(slot + (size_t)ps.H * nb, 0, (size_t)(ps.Hs - ps.H) * nb * 4, ctx->stream));
}

void crt_triple_ladders(const pgpu_seckey* sk, const BigU e[2], const TriplePlan& tp, const TriplePlan& tq, size_t nb, int beside) {
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx &mp3 = sk->mp3, &mq3 = sk->mq3;
  const bool on_side = ctx->stream == ctx->side || ctx->stream == ctx->side_l[0] || ctx->stream == ctx->side_l[1] || ctx->stream == ctx->side_l[2];
  const bool six = ctx->use_asm && plan::crt_triple_lanes6(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), sk->triple_h6 > 0, ctx->use_lanes8,
                                                           std::max(beside, ctx->exclusive_call && on_side ? 2 : 1));
  Prog lad[2];
  if (!six) {
    for (int half = 0; half < 2; ++half) {
      emit_modexp_shared(lad[half], e[half], 0, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
      lad[half].end();
    }
    SegSpec sp{&mp3, &lad[0], tp.mem, nullptr}, sq{&mq3, &lad[1], tq.mem, nullptr};
    sp.pair = mp3.triple.kconsts; sp.pair_n0inv = sk->mp.n0inv; sp.pair_h = tp.H; sp.pair_lanes = 3; sp.tconsts = mp3.triple.tconsts;
    sq.pair = mq3.triple.kconsts; sq.pair_n0inv = sk->mq.n0inv; sq.pair_h = tq.H; sq.pair_lanes = 3; sq.tconsts = mq3.triple.tconsts;
    run_vm(ctx, nb, sp, &sq, true);
    return;
  }
  // two lanes per digit: slots of 3 x h6 limbs of their own (0 in, 2 tmp, 3 out, 5.. the table); the digits zero-extended, radix R_H -> R_h6
  // with the first product of the program and back with its last
  const int H = tp.H, H6 = sk->triple_h6;
  const size_t S1 = (size_t)H * nb, S6 = (size_t)H6 * nb, SW6 = 3 * S6;
  uint32_t* m6[2];
  for (int half = 0; half < 2; ++half) {
    const TriplePlan& t = half ? tq : tp;
    m6[half] = ctx->ws_t<uint32_t>(SW6 * (size_t)(5 + 32));
    HIPCHK(hipMemsetAsync(m6[half], 0, SW6 * 4, ctx->stream));
    for (int d = 0; d < 3; ++d) launch_restride(t.slot(0) + (size_t)d * S1, nb, nb, nullptr, m6[half] + (size_t)d * S6, nb, H, ctx->stream);
    Prog& p = lad[half];
    p.op(VM_LOAD, 0); p.op(VM_MULC, 0); p.op(VM_STORE, 0);
    emit_modexp_shared(p, e[half], 0, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    p.op(VM_LOAD, 3); p.op(VM_MULC, 1); p.op(VM_STORE, 3);
    p.end();
  }
  SegSpec sp{&mp3, &lad[0], m6[0], nullptr}, sq{&mq3, &lad[1], m6[1], nullptr};
  sp.pair = sk->tkc6_p.d; sp.pair_n0inv = sk->mp.n0inv; sp.pair_h = H6; sp.pair_lanes = 6; sp.tconsts = sk->ttc6_p.d;
  sq.pair = sk->tkc6_q.d; sq.pair_n0inv = sk->mq.n0inv; sq.pair_h = H6; sq.pair_lanes = 6; sq.tconsts = sk->ttc6_q.d;
  run_vm(ctx, nb, sp, &sq, true);
  for (int half = 0; half < 2; ++half) {
    const TriplePlan& t = half ? tq : tp;
    for (int d = 0; d < 3; ++d) launch_restride(m6[half] + 3 * SW6 + (size_t)d * S6, nb, nb, nullptr, t.slot(3) + (size_t)d * S1, nb, H, ctx->stream);
  }
}

bool crt_pair8_usable(const pgpu_seckey* sk, size_t nb) {
  pgpu_ctx* ctx = sk->ctx;
  return sk->has_pair && ctx->use_asm && ctx->use_pair &&
         plan::crt_pair_lanes8(nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus), sk->pair_h8 > 0, ctx->use_lanes8, ctx->exclusive_call ? 2 : 1);
}
void crt_pair8_ladders(const pgpu_seckey* sk, const BigU e[2], const uint32_t* const in[2], uint32_t* const out[2], size_t nb) {
  pgpu_ctx* ctx = sk->ctx;
  const int H = sk->mp.WT, H8 = sk->pair_h8;
  const size_t S1 = (size_t)H * nb, SW8 = (size_t)2 * H8 * nb;
  uint32_t* m8[2];
  Prog lad[2];
  for (int half = 0; half < 2; ++half) {
    // pair slots of 2 x H8 limbs: 2 in, 3 out, 5.. the table; the digits zero-extended, radix R_H -> R_H8 with the first product of the
    // program and back with its last
    m8[half] = ctx->ws_t<uint32_t>(SW8 * (size_t)(5 + 32));
    HIPCHK(hipMemsetAsync(m8[half] + 2 * SW8, 0, SW8 * 4, ctx->stream));
    launch_restride(in[half], nb, nb, nullptr, m8[half] + 2 * SW8, nb, H, ctx->stream);
    launch_restride(in[half] + S1, nb, nb, nullptr, m8[half] + 2 * SW8 + (size_t)H8 * nb, nb, H, ctx->stream);
    Prog& p = lad[half];
    p.op(VM_LOAD, 2); p.op(VM_MULC, 0); p.op(VM_STORE, 2);
    emit_modexp_shared(p, e[half], 2, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    p.op(VM_LOAD, 3); p.op(VM_MULC, 1); p.op(VM_STORE, 3);
    p.end();
  }
  SegSpec sp{&sk->mp2, &lad[0], m8[0], nullptr}, sq{&sk->mq2, &lad[1], m8[1], nullptr};
  sp.pair = sk->pair8_p.d; sp.pair_n0inv = sk->mp.n0inv; sp.pair_h = H8; sp.pair_lanes = 8; sp.tconsts = sk->pair8t_p.d;
  sq.pair = sk->pair8_q.d; sq.pair_n0inv = sk->mq.n0inv; sq.pair_h = H8; sq.pair_lanes = 8; sq.tconsts = sk->pair8t_q.d;
  run_vm(ctx, nb, sp, &sq, true);
  for (int half = 0; half < 2; ++half) {
    launch_restride(m8[half] + 3 * SW8, nb, nb, nullptr, out[half], nb, H, ctx->stream);
    launch_restride(m8[half] + 3 * SW8 + (size_t)H8 * nb, nb, nb, nullptr, out[half] + S1, nb, H, ctx->stream);
  }
}

uint32_t* pow_n_crt(const pgpu_seckey* sk, const uint32_t* base, const BigU& e, size_t nb) {
  pgpu_ctx* ctx = sk->ctx;
  const ModCtx &mp = sk->mp, &mq = sk->mq;
  // (on a side lane -- s of the DDLEQ prover -- the ladder runs beside the main stream's and ct1's decryption)
  const bool on_side = ctx->stream == ctx->side || ctx->stream == ctx->side_l[0] || ctx->stream == ctx->side_l[1] || ctx->stream == ctx->side_l[2];
  const PrimeShape ps = prime_shape(sk, nb, on_side ? 4 : 1);
  const int W1 = mp.WT, WN = sk->pk->mn.WT;
  const size_t S1 = (size_t)W1 * nb, Ss = (size_t)ps.Hs * nb;
  // slots per half: 0 in, 2 tmp, 3 out, 5..36 table
  uint32_t *memp = ctx->ws_t<uint32_t>(Ss * 37), *memq = ctx->ws_t<uint32_t>(Ss * 37);
  reduce_mod(ctx, *ps.m[0], base, WN, memp, nb);
  reduce_mod(ctx, *ps.m[1], base, WN, memq, nb);
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
  SegSpec sp{ps.m[0], &pp, memp, nullptr}, sq{ps.m[1], &pq, memq, nullptr};
  run_vm(ctx, nb, sp, &sq, true);
  // small memory: 2 x_p, 3 x_q, 4 B, 5 A, 6 h
  uint32_t* m1 = ctx->ws_t<uint32_t>(S1 * 7);
  HIPCHK(hipMemcpyAsync(m1 + 2 * S1, memp + 3 * Ss, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(m1 + 3 * S1, memq + 3 * Ss, S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
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
// The n-th power through the structure of the unit group (context flag "lift"): Z*_{pr^2} = <1 + pr> x T with T the Teichmueller
// lifts of Z*_pr and omega(u) = u^pr mod pr^2, so for n = p q and pr = p, o = q (or the other way round)
//     x^n = (x^pr)^o = omega(x)^o = omega(x^o mod pr) = t^pr mod pr^2,   t = (x mod pr)^(o mod (pr - 1)) mod pr
// -- a ladder modulo the PRIME (1 024 squarings of 1.5 H^2 on the generic one-lane kernel) and the ladder of the headline Decrypt with
// the exponent pr, instead of 2 047 squarings in pair form: 8.6 M multiply-adds per number and half where the exponent n mod
// pr (pr - 1) needs 12.2 M.  The same integers for EVERY x: a multiple of pr gives t = 0 and 0^pr = 0 = x^n modulo pr^2 (n >= 2).
// power_residues returns the canonical t per half (H limbs, stride nb), or false where the shortcut does not apply.
// k = 1: the residues for x^n; k = 2: for x^(n^2) modulo pr^3 (= omega_3(x^(o^2) mod pr): the same argument one digit further).
bool power_residues_usable(const pgpu_seckey* sk, int k, BigU r[2]) {
  if (!sk->ctx->use_lift || !sk->has_crt || sk->mp.WT != sk->mq.WT) return false;
  for (int half = 0; half < 2; ++half) {
    const BigU &pr = half ? sk->q : sk->p, &o = half ? sk->p : sk->q;
    const BigU ord = pr - BigU(1);
    r[half] = o % ord;
    if (k == 2) r[half] = (r[half] * r[half]) % ord;
    if (r[half].is_zero()) r[half] = ord;
    if (pr.bit_length() < 64 || r[half].bit_length() < 64) return false;       // (toy keys: the literal exponent)
  }
  return true;
}
bool power_residues(const pgpu_seckey* sk, const uint32_t* base, int k, size_t nb, uint32_t* t[2]) {
  pgpu_ctx* ctx = sk->ctx;
  BigU r[2];
  if (!power_residues_usable(sk, k, r)) return false;
  const PrimeShape ps = prime_shape(sk, nb);
  const ModCtx &mp = *ps.m[0], &mq = *ps.m[1];
  const int H = sk->mp.WT, WN = sk->pk->mn.WT;
  const size_t Ss = (size_t)ps.Hs * nb;
  // slots per half (H limbs; Hs on the four-lane twins): 0 in, 2 tmp, 3 out, 5..36 the odd powers of the sliding windows
  uint32_t *memp = ctx->ws_t<uint32_t>(Ss * 37), *memq = ctx->ws_t<uint32_t>(Ss * 37);
  reduce_mod(ctx, mp, base, WN, memp, nb);
  reduce_mod(ctx, mq, base, WN, memq, nb);
  Prog pp, pq;
  emit_modexp_shared(pp, r[0], 0, NO_SLOT, 2, 3, 5, NO_SLOT, true);
  pp.end();
  emit_modexp_shared(pq, r[1], 0, NO_SLOT, 2, 3, 5, NO_SLOT, true);
  pq.end();
  SegSpec sp{&mp, &pp, memp, nullptr}, sq{&mq, &pq, memq, nullptr};
  run_vm(ctx, nb, sp, &sq, true);
  t[0] = memp + 3 * Ss;                       // (the first H rows of the slot: PrimeShape)
  t[1] = memq + 3 * Ss;
  launch_canon(t[0], sk->mp.d_nmod, H, nb, ctx->stream);
  launch_canon(t[1], sk->mq.d_nmod, H, nb, ctx->stream);
  return true;
}

uint32_t* pow_n2_crt(const pgpu_seckey* sk, const uint32_t* base, const BigU& e, size_t nb) {
  pgpu_ctx* ctx = sk->ctx;
  const int H = sk->mp.WT, W2 = sk->mp2.WT, WN2 = sk->pk->mn2.WT;
  const size_t S1 = (size_t)H * nb, S2 = (size_t)W2 * nb;
  uint32_t* xh[2];
  uint32_t* mem[2];
  Prog lad[2];
  BigU ehs[2];
  uint32_t* tl[2] = {nullptr, nullptr};
  const bool lifted = e == sk->pk->N && power_residues(sk, base, 1, nb, tl);      // e == n: t = x^(other prime) modulo each prime, then t^prime
  Fork in(ctx);                                                         // the q-half's entry chain beside the p-half's
  for (int half = 0; half < 2; ++half) {
    in.chain(half);
    const ModCtx &m1 = half ? sk->mq : sk->mp, &m2 = half ? sk->mq2 : sk->mp2;
    const BigU& pr = half ? sk->q : sk->p;
    // slots (W2 limbs): 0 x, 2 pair form in, 3 out, 5..36 table
    uint32_t* mm = mem[half] = ctx->ws_t<uint32_t>(S2 * 37);
    if (lifted) {                                                       // t < prime: its own residue modulo prime^2
      HIPCHK(hipMemsetAsync(mm + S1, 0, (S2 - S1) * 4, ctx->stream));
      HIPCHK(hipMemcpyAsync(mm, tl[half], S1 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
      reduce_mod(ctx, m2, base, sk->pk->mn.WT, mm, nb);
    }
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
    if (lifted) {
      eh = pr;                             // omega(t) = t^prime
    } else {
      if (!(e < ord)) {
        eh = e % ord;
        if (eh < BigU(2)) eh = eh + ord;   // x^e = 0 modulo prime^2 for a multiple of the prime and e >= 2: keep it so
      }
      if (eh.bit_length() < 64) eh = e;
    }
    ehs[half] = eh;
    emit_modexp_shared(lad[half], eh, 2, NO_SLOT, 2, 3, 5, NO_SLOT, true, true);
    lad[half].end();
  }
  in.join();
  if (crt_pair8_usable(sk, nb)) {
    // (small batches: eight lanes per number -- the ladder's latency is the run time)
    const uint32_t* in8[2] = {mem[0] + 2 * S2, mem[1] + 2 * S2};
    uint32_t* out8[2] = {mem[0] + 3 * S2, mem[1] + 3 * S2};
    crt_pair8_ladders(sk, ehs, in8, out8, nb);
  } else {
    // (small batches on two lanes per number, as Decrypt chooses: a squaring is 37 rows of 74 multiplies instead of the one-lane
    // kernel's 4 810 in a row -- the ladder's latency is the run time there)
    const int lanes = plan::crt_pair_lanes(sk->pair_lanes, sk->pair_small2, nb, plan::lanes_target(ctx->lanes_wanted, ctx->stream_cus));
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
    const BigU es[2] = {sk->p - BigU(1), sk->q - BigU(1)};
    crt_triple_ladders(sk, es, tp, tq, nb);
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

}  // namespace pgi

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

namespace pgi {

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

// The fixed base h_s of AltEncrypt (paillier.go:416-434: h_1 = (N-H)^N mod N^2, h_2 = (N^2-H)^(N^2) mod N^3), computed once per key;
// its comb table (7-bit windows, VM_MULCV7) is built on the device by ensure_comb7.
void ensure_alt_table(pgpu_pubkey* pk, int level) {
  pgpu_pubkey::AltTab& t = pk->alt[level];
  if (t.built) return;
  if (pk->H.is_zero() || pk->Kk.is_zero()) api_throw(PGPU_ERR_INVALID, "alternative encryption needs H and K in the public key");
  const size_t kbits = pk->Kk.bit_length() - 1;
  if (!(hostbig::shl(BigU(1), kbits) == pk->Kk)) api_throw(PGPU_ERR_UNSUPPORTED, "K must be a power of two (KeyGen: 2^(secparam/2))");
  ModCtx& mc = (level == PGPU_LEVEL_ONE) ? pk->mn2 : *pk->mn3;
  const BigU ns = (level == PGPU_LEVEL_ONE) ? pk->N : pk->mn2.N;
  if (hostbig::cmp(ns, pk->H) <= 0) api_throw(PGPU_ERR_INVALID, "H must be smaller than n^s");
  t.hs = hostbig::powmod(ns - pk->H, ns, mc.N);
  t.kbits = kbits;
  t.built = true;
}

}  // namespace pgi

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
    const pgpu_pubkey::Comb7& cb = ensure_comb7(const_cast<pgpu_pubkey*>(pk), level, t.hs, t.kbits);
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
    emit_comb7(p, std::max(we, 1));   // h_s^(r mod K): one table product per 7 bits
    p.op(VM_MUL, 0);      // * g^m (plain) -> leaves Montgomery form
    p.op(VM_STORE, 1);
    p.end();
    SegSpec sg{&mc, &p, memv, exps};
    sg.tconsts = cb.d_table;
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

int pgpu_encrypt_with_r_sk(const pgpu_seckey* sk, int level, size_t batch, const uint8_t* m, size_t m_stride, const uint8_t* r,
                           size_t r_stride, uint8_t* c, size_t c_stride, int mem) {
  if (!sk) return fail(PGPU_ERR_INVALID, "null key");
  pgpu_ctx* ctx = sk->ctx;
  const pgpu_pubkey* pk = sk->pk;
  if (level == PGPU_LEVEL_TWO && pk->g_is_n_plus_1 && pk->mn3 && struct_pow_usable(sk)) {
    // Level two for the key holder: r^(n^2) mod n^3 is the Teichmueller lift of r^(n^2) mod n, i.e. modulo p^3 the lift of
    // t = (r mod p)^(q^2 mod (p - 1)) -- a ladder modulo the primes and ONE lift (1 023 squarings modulo p^3, q^3; the DDLEQ prover's
    // teichmueller_lift): 33 M multiply-adds per ciphertext where the public path's (r^n mod n^2)^n mod n^3 needs 166 M.  The same
    // integers for every unit r; an r that shares a factor with n has no lift (the lane is flagged) and the batch takes the public path.
    BigU rr[2];
    if (power_residues_usable(sk, 2, rr)) {
      bool redo = false;
      const int rc = guarded([&] {
        check_batch_args(m, c, batch);
        if (!r) api_throw(PGPU_ERR_INVALID, "null buffer");
        ctx->bind();
        ctx->reset_ws();
        const ModCtx &mn = pk->mn, &mn3 = *pk->mn3;
        const size_t nb = round_up(batch, VM_BLOCK);
        uint32_t* gm = ctx->ws_t<uint32_t>((size_t)mn3.WT * nb);
        build_gm(ctx, pk, level, m, m_stride, batch, mem, nb, gm);                     // 1 + m n + C(m, 2) n^2
        uint32_t* rl = ctx->ws_t<uint32_t>((size_t)mn.WT * nb);
        unpack_mod(ctx, mn, r, r_stride, batch, mem, rl, nb, true);                    // r^(n^2) mod n^3 depends on r mod n only
        uint32_t* t[2];
        if (!power_residues(sk, rl, 2, nb, t)) api_throw(PGPU_ERR_UNSUPPORTED, "internal: power_residues");
        int32_t* d_status = ctx->ws_t<int32_t>(nb);
        HIPCHK(hipMemsetAsync(d_status, 0, nb * 4, ctx->stream));
        uint32_t* T = ctx->ws_t<uint32_t>((size_t)mn3.WT * nb);
        teichmueller_lift(sk, t, nb, d_status, T);
        uint32_t* out = ctx->ws_t<uint32_t>((size_t)mn3.WT * nb);
        modmul_arrays(ctx, mn3, gm, T, nb, out);                                       // c = G^m r^(n^2) mod n^3
        std::vector<int32_t> hst(batch);
        HIPCHK(hipMemcpyAsync(hst.data(), d_status, batch * 4, hipMemcpyDeviceToHost, ctx->stream));
        pack_result(ctx, out, mn3.WT, nb, batch, c, c_stride, mn3.nbytes, mem);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < batch; ++i) redo = redo || hst[i] != 0;
      });
      if (rc != PGPU_OK) return rc;
      if (!redo) return PGPU_OK;
    }
    return pgpu_encrypt_with_r(pk, level, batch, m, m_stride, r, r_stride, c, c_stride, mem);
  }
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
