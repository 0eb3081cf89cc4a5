// keys.cpp -- key handles of the C ABI: moduli, Montgomery / pair / digit-form constants, CRT material of a secret key.  Host
// big-integer work, once per key.
#include "engine.hpp"

namespace pgi {

// Exponents of ladders modulo pr^3 can be taken modulo the order of its unit group, ord = pr^2 (pr - 1) = 2^t m: a quarter
// shorter than the exponents modulo n^2 the DDLEQ prover raises to (the holder of the factorisation only).
BigU order_fixup(const BigU& e, const BigU& ord) {
  if (e < ord) return e;
  BigU r = e % ord;
  if (r < BigU(3)) r = r + ord;      // x^e = 0 for a non-unit x and e >= 3: keep the reduced exponent >= 3 as well
  return r;
}

// Constants of the pair kernel for a prime of H limbs: its limbs, then Cadj -- the multiple of the prime whose limbs
// 0..H-1 can all be taken from [2^28, 2^29), so that Cadj - m is limb-wise non-negative for every quotient m < 2^(28 H).
std::vector<uint32_t> make_pair_consts(const BigU& pr, int H) {
  BigU D;
  for (int j = 0; j < H; ++j) D = D + hostbig::shl(BigU(1), (size_t)LB * j + LB);
  BigU kq, kr;
  hostbig::divmod(D, pr, kq, kr);
  BigU E = (kq + BigU(1)) * pr - D;          // 0 < E <= prime < 2^(28 H)
  std::vector<uint32_t> v = pr.to_limbs(LB, H), el = E.to_limbs(LB, H);
  for (int j = 0; j < H; ++j) v.push_back(el[j] + (1u << LB));
  return v;
}

// Constants of the eight-lane pair kernel for modulus mod2 = root^2, a root of H limbs in digits of h8 limbs (four slices): root | Cadj (+ one
// word of padding), and the pair digits (d0 | d1, h8 limbs each) of R_h8^2 R_H^-1 (entry from the radix-R_H pair form), R_H (exit), R_h8 (= 1)
bool make_pair8_consts(const BigU& root, const BigU& mod2, int H, int h8, std::vector<uint32_t>& c8, std::vector<uint32_t>& t8) {
  c8 = make_pair_consts(root, h8);
  c8.push_back(0);
  const BigU rh = hostbig::shl(BigU(1), (size_t)LB * H) % mod2, r8 = hostbig::shl(BigU(1), (size_t)LB * h8) % mod2;
  BigU rh_inv;
  if (!hostbig::modinv(rh, mod2, rh_inv)) return false;
  const BigU vals[3] = {hostbig::mulmod(hostbig::mulmod(r8, r8, mod2), rh_inv, mod2), rh, r8};
  t8.clear();
  for (const BigU& v : vals) {
    BigU d1, d0;
    hostbig::divmod(v, root, d1, d0);
    auto l0 = d0.to_limbs(LB, (size_t)h8), l1 = d1.to_limbs(LB, (size_t)h8);
    t8.insert(t8.end(), l0.begin(), l0.end());
    t8.insert(t8.end(), l1.begin(), l1.end());
  }
  return true;
}

// Constants of the three-digit kernel for the root n (H limbs): n padded to an even number of words, then the pairs
// (C1_i, C2_i), then two words of padding (the kernel prefetches one pair past the end).  C1 = k1 n is make_pair_consts'
// Cadj (every limb in [2^28, 2^29)); C2 = -k1 (mod n) shifted into the same limb range: the -C1 n that the first link
// leaves in digit one is -k1 n^2, which C2 cancels in digit two.
std::vector<uint32_t> make_triple_kconsts(const BigU& n, int H) {
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

}  // namespace pgi

extern "C" {

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
        std::vector<uint32_t> pc8, tc;
        if (make_pair8_consts(pk->N, n2, H, h8, pc8, tc)) {
          pk->pairn_consts8.w = (int)pc8.size();
          HIPCHK(hipMalloc((void**)&pk->pairn_consts8.d, pc8.size() * 4));
          HIPCHK(hipMemcpy(pk->pairn_consts8.d, pc8.data(), pc8.size() * 4, hipMemcpyHostToDevice));
          pk->pairn_tconsts8.w = (int)tc.size();
          HIPCHK(hipMalloc((void**)&pk->pairn_tconsts8.d, tc.size() * 4));
          HIPCHK(hipMemcpy(pk->pairn_tconsts8.d, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
          pi.h8 = h8;
          pi.consts8 = pk->pairn_consts8.d;
          pi.tconsts8 = pk->pairn_tconsts8.d;
        }
      }
      const int h16 = (H + 7) / 8 * 8;
      if (pi.h8 && vm_asm_available(h16 / 8, 128)) {
        std::vector<uint32_t> pc16, tc;
        if (make_pair8_consts(pk->N, n2, H, h16, pc16, tc)) {
          pk->pairn_consts16.w = (int)pc16.size();
          HIPCHK(hipMalloc((void**)&pk->pairn_consts16.d, pc16.size() * 4));
          HIPCHK(hipMemcpy(pk->pairn_consts16.d, pc16.data(), pc16.size() * 4, hipMemcpyHostToDevice));
          pk->pairn_tconsts16.w = (int)tc.size();
          HIPCHK(hipMalloc((void**)&pk->pairn_tconsts16.d, tc.size() * 4));
          HIPCHK(hipMemcpy(pk->pairn_tconsts16.d, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
          pi.h16 = h16;
          pi.consts16 = pk->pairn_consts16.d;
          pi.tconsts16 = pk->pairn_tconsts16.d;
        }
      }
    }
    const bool one_lane_digit = pk->mn.K == 1 && vm_asm_available(pk->mn.WT, 48);
    const bool two_lane_digit = pk->mn.WT % 2 == 0 && vm_asm_available(pk->mn.WT / 2, 112);     // 3072-bit keys: digits of 110 limbs
    if (pk->mn3 && (one_lane_digit || two_lane_digit) && (size_t)LB * pk->mn3->WT >= n3.bit_length() + 3) {
      setup_triple(*pk->mn3, pk->mn, pk->mn2, pk->triple_kconsts, pk->triple_tconsts, pk->ninv2k.d, pk->ninv2k_2.d, pk->n_limbs.d,
                   pk->n2_limbs.d);
      pk->mn3->triple.lanes6_only = !one_lane_digit;
      {
        // sixteen lanes per number (GenQ12): digits of h12 = H rounded up to four slices
        const int H = pk->mn.WT, h12 = (H + 3) / 4 * 4;
        if (one_lane_digit && two_lane_digit && vm_asm_available(h12 / 4, 160)) {
          const std::vector<uint32_t> k12 = make_triple_kconsts(pk->N, h12);
          const BigU rh = hostbig::shl(BigU(1), (size_t)LB * H) % n3, r12 = hostbig::shl(BigU(1), (size_t)LB * h12) % n3;
          BigU rh_inv;
          if (hostbig::modinv(rh, n3, rh_inv)) {
            const BigU vals[3] = {r12, hostbig::mulmod(hostbig::mulmod(r12, r12, n3), rh_inv, n3), rh};
            std::vector<uint32_t> t12;
            for (const BigU& v : vals) {
              BigU q1, d0, d1, d2;
              hostbig::divmod(v, pk->N, q1, d0);
              hostbig::divmod(q1, pk->N, d2, d1);
              for (const BigU* dg : {&d0, &d1, &d2}) {
                auto l = dg->to_limbs(LB, (size_t)h12);
                t12.insert(t12.end(), l.begin(), l.end());
              }
            }
            auto put = [&](DevLimbs& d, const std::vector<uint32_t>& v) {
              d.w = (int)v.size();
              HIPCHK(hipMalloc((void**)&d.d, v.size() * 4));
              HIPCHK(hipMemcpy(d.d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
            };
            put(pk->triple_kconsts12, k12);
            put(pk->triple_tconsts12, t12);
            pk->mn3->triple.h12 = h12;
            pk->mn3->triple.kconsts12 = pk->triple_kconsts12.d;
            pk->mn3->triple.tconsts12 = pk->triple_tconsts12.d;
          }
        }
      }
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
        if (sk->mp.WL == 37 && sk->mp.K == 1 && p.bit_length() + 3 <= (size_t)LB * 40) {
          sk->mp_s.init(ctx, p, 10, 4);
          sk->mq_s.init(ctx, q, 10, 4);
          sk->mp_s.upload();
          sk->mq_s.upload();
          sk->has_sliced_primes = true;
        }
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
          {
            const int h8 = (H + 3) / 4 * 4;
            std::vector<uint32_t> c8p, t8p, c8q, t8q;
            if (sk->pair_lanes == 1 && vm_asm_available(h8 / 4, 96) && make_pair8_consts(p, sk->mp2.N, H, h8, c8p, t8p) &&
                make_pair8_consts(q, sk->mq2.N, H, h8, c8q, t8q)) {
              put(sk->pair8_p, c8p); put(sk->pair8t_p, t8p);
              put(sk->pair8_q, c8q); put(sk->pair8t_q, t8q);
              sk->pair_h8 = h8;
            }
          }
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
              {
                // two lanes per digit for small batches: digits of h6 = H + 1 limbs in two slices (GenQ6)
                const int H = sk->mp.WT, h6 = (H + 1) / 2 * 2;
                if (H % 2 == 1 && vm_asm_available(h6 / 2, 112)) {
                  auto consts6 = [&](const BigU& pr, const BigU& pr3, DevLimbs& kc, DevLimbs& tc) {
                    const std::vector<uint32_t> k6 = make_triple_kconsts(pr, h6);
                    const BigU rh = hostbig::shl(BigU(1), (size_t)LB * H) % pr3, r6 = hostbig::shl(BigU(1), (size_t)LB * h6) % pr3;
                    BigU rh_inv;
                    if (!hostbig::modinv(rh, pr3, rh_inv)) return false;
                    const BigU vals[3] = {hostbig::mulmod(hostbig::mulmod(r6, r6, pr3), rh_inv, pr3), rh, r6};
                    std::vector<uint32_t> t6;
                    for (const BigU& v : vals) {
                      BigU q1, d0, d1, d2;
                      hostbig::divmod(v, pr, q1, d0);
                      hostbig::divmod(q1, pr, d2, d1);
                      for (const BigU* dg : {&d0, &d1, &d2}) {
                        auto l = dg->to_limbs(LB, (size_t)h6);
                        t6.insert(t6.end(), l.begin(), l.end());
                      }
                    }
                    auto put = [&](DevLimbs& d, const std::vector<uint32_t>& v) {
                      d.w = (int)v.size();
                      HIPCHK(hipMalloc((void**)&d.d, v.size() * 4));
                      HIPCHK(hipMemcpy(d.d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
                    };
                    put(kc, k6);
                    put(tc, t6);
                    return true;
                  };
                  if (consts6(p, sk->mp3.N, sk->tkc6_p, sk->ttc6_p) && consts6(q, sk->mq3.N, sk->tkc6_q, sk->ttc6_q)) sk->triple_h6 = h6;
                }
              }
              sk->eo_p.init(ctx, p);
              sk->eo_q.init(ctx, q);
              // the Teichmueller lift and the exponents modulo p - 1, q - 1 (struct_pow_n3)
              sk->eo1_p.init(ctx, p, 0);
              sk->eo1_q.init(ctx, q, 0);
              if (sk->eo1_p.ok && sk->eo1_q.ok && sk->eo1_p.w == sk->eo1_q.w && sk->eo1_p.modd.WT == sk->mp.WT &&
                  sk->eo1_q.modd.WT == sk->mq.WT) {
                const BigU n2v = n * n;
                sk->n2_mod_p1 = n2v % (p - BigU(1));
                sk->n2_mod_q1 = n2v % (q - BigU(1));
                auto lift_consts = [&](const BigU& pr, const BigU& pr2, ModCtx& m1, ModCtx& m2, int& c_z, int& c_z2) {
                  BigU inv;
                  if (!hostbig::modinv((pr - BigU(1)) % pr2, pr2, inv)) return false;
                  const BigU z = (pr2 - inv) % pr2;                                   // -(pr - 1)^-1 mod pr^2
                  const BigU zz = z.is_zero() ? BigU(0) : hostbig::shr(z * (z - BigU(1)), 1) % pr;   // C(z, 2) mod pr
                  c_z = m2.add_const(m2.to_mont(z));
                  c_z2 = m1.add_const(zz);
                  return true;
                };
                if (lift_consts(p, p2, sk->mp, sk->mp2, sk->c_lz_p2, sk->c_lz2_p) &&
                    lift_consts(q, q2, sk->mq, sk->mq2, sk->c_lz_q2, sk->c_lz2_q)) {
                  sk->mp.upload(); sk->mq.upload(); sk->mp2.upload(); sk->mq2.upload();
                  sk->has_lift = true;
                }
              }
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
