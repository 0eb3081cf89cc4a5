/* CPU ORACLE (test infrastructure only) -- libgmp restatement of sachaservan/paillier's hot path.
 *
 * NOT product code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from this file, and only as the checker / the reported CPU baseline.
 *
 * The reference executes every hot-path operation through github.com/ncw/gmp (version unpinned by the
 * reference: no go.mod), a cgo wrapper over libgmp.  This file performs the SAME libgmp calls in the SAME
 * order as the reference functions cited below, one ciphertext per call, fresh temporaries per call,
 * lambda^-1 recomputed per Decrypt -- no batching, no CRT, no hoisting -- so its timing is what the
 * reference's own arithmetic costs on this host (minus ~100 ns of cgo per call).
 *
 * Parity pin: see oracle/paillier_oracle.py header (toy KATs from the reference's tests; no large KATs
 * exist in the reference).  tests/test_oracle_cross.py checks this file against the Python-int oracle.
 *
 * Buffers: unsigned big-endian, fixed stride, element-major -- the C-ABI operand format.
 */
#include <gmp.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void imp(mpz_t z, const uint8_t* p, size_t n) { mpz_import(z, n, 1, 1, 1, 0, p); }
static void expo(const mpz_t z, uint8_t* p, size_t n) {
  size_t cnt = (mpz_sizeinbase(z, 2) + 7) / 8;
  memset(p, 0, n);
  if (mpz_sgn(z) != 0 && cnt <= n) mpz_export(p + (n - cnt), NULL, 1, 1, 1, 0, z);
}

/* gmp.Int.Exp: y <= 0 -> 1 */
static void gmp_exp(mpz_t z, const mpz_t x, const mpz_t y, const mpz_t m) {
  if (mpz_sgn(y) <= 0) { mpz_set_ui(z, 1); return; }
  mpz_powm(z, x, y, m);
}

/* paillier.go:436-440  L(u, n) = Div(u - 1, n)  (Euclidean; n > 0 so floor) */
static void L_fn(mpz_t out, const mpz_t u, const mpz_t n) {
  mpz_t t;
  mpz_init(t);
  mpz_sub_ui(t, u, 1);
  mpz_fdiv_q(out, t, n);
  mpz_clear(t);
}

/* paillier.go:292-303 SecretKey.Decrypt, level one (s = 1), verbatim call sequence. */
static void decrypt1(mpz_t m, const mpz_t c, const mpz_t n, const mpz_t n2, const mpz_t lambda) {
  mpz_t tmp, amod, ml, mu;
  mpz_inits(tmp, amod, ml, mu, NULL);
  gmp_exp(tmp, c, lambda, n2);          /* :296 */
  mpz_mod(amod, tmp, n2);               /* recoveryAlgorithm :316 (j = 1) */
  L_fn(ml, amod, n);                    /* :318 */
  mpz_invert(mu, lambda, n);            /* :298, recomputed every call */
  mpz_mul(m, ml, mu);                   /* :300 */
  mpz_mod(m, m, n);
  mpz_clears(tmp, amod, ml, mu, NULL);
}

/* returns the number of threads used */
int oracle_decrypt_batch(const uint8_t* n_be, size_t n_len, const uint8_t* lambda_be, size_t l_len, size_t batch,
                         const uint8_t* c, size_t c_stride, uint8_t* m_out, size_t m_stride, int threads) {
  mpz_t n, n2, lambda;
  mpz_inits(n, n2, lambda, NULL);
  imp(n, n_be, n_len);
  imp(lambda, lambda_be, l_len);
  mpz_mul(n2, n, n);
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
    mpz_t ci, mi;
    mpz_inits(ci, mi, NULL);
#pragma omp for schedule(dynamic, 4)
    for (long i = 0; i < (long)batch; ++i) {
      imp(ci, c + (size_t)i * c_stride, c_stride);
      decrypt1(mi, ci, n, n2, lambda);
      expo(mi, m_out + (size_t)i * m_stride, m_stride);
    }
    mpz_clears(ci, mi, NULL);
  }
#else
  (void)threads;
  mpz_t ci, mi;
  mpz_inits(ci, mi, NULL);
  for (size_t i = 0; i < batch; ++i) {
    imp(ci, c + i * c_stride, c_stride);
    decrypt1(mi, ci, n, n2, lambda);
    expo(mi, m_out + i * m_stride, m_stride);
  }
  mpz_clears(ci, mi, NULL);
#endif
  mpz_clears(n, n2, lambda, NULL);
  return used;
}

/* paillier.go:206-218 EncryptWithRAtLevel, level one. */
int oracle_encrypt_batch(const uint8_t* n_be, size_t n_len, const uint8_t* g_be, size_t g_len, size_t batch,
                         const uint8_t* m, size_t m_stride, const uint8_t* r, size_t r_stride, uint8_t* c_out,
                         size_t c_stride, int threads) {
  mpz_t n, n2, g;
  mpz_inits(n, n2, g, NULL);
  imp(n, n_be, n_len);
  imp(g, g_be, g_len);
  mpz_mul(n2, n, n);
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    mpz_t mi, ri, gm, rn, ci;
    mpz_inits(mi, ri, gm, rn, ci, NULL);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (long i = 0; i < (long)batch; ++i) {
      imp(mi, m + (size_t)i * m_stride, m_stride);
      imp(ri, r + (size_t)i * r_stride, r_stride);
      gmp_exp(gm, g, mi, n2);   /* :213 */
      gmp_exp(rn, ri, n, n2);   /* :214 */
      mpz_mul(ci, gm, rn);      /* :216 */
      mpz_mod(ci, ci, n2);
      expo(ci, c_out + (size_t)i * c_stride, c_stride);
    }
    mpz_clears(mi, ri, gm, rn, ci, NULL);
  }
  mpz_clears(n, n2, g, NULL);
  return used;
}

/* gmp.Int.Exp(base, e, mod) for a batch with a shared exponent (ConstMult operations.go:58-64,
 * PartialDecrypt thresholdkey.go:192-201). */
int oracle_modexp_batch(const uint8_t* mod_be, size_t mod_len, const uint8_t* e_be, size_t e_len, size_t batch,
                        const uint8_t* base, size_t b_stride, uint8_t* out, size_t o_stride, int threads) {
  mpz_t n, e;
  mpz_inits(n, e, NULL);
  imp(n, mod_be, mod_len);
  imp(e, e_be, e_len);
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    mpz_t b, o;
    mpz_inits(b, o, NULL);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (long i = 0; i < (long)batch; ++i) {
      imp(b, base + (size_t)i * b_stride, b_stride);
      gmp_exp(o, b, e, n);
      expo(o, out + (size_t)i * o_stride, o_stride);
    }
    mpz_clears(b, o, NULL);
  }
  mpz_clears(n, e, NULL);
  return used;
}


/* ---- Decrypt with CRT: the second CPU-baseline line BASELINE.md section 3 promises ("the same harness with CRT, so
 * that the GPU's algorithmic advantage and its hardware advantage are separable").  NOT the reference's algorithm
 * (paillier.go:292-303 has no CRT; SecretKey does not even keep p, q): the textbook Paillier CRT decryption
 *   m_p = L_p(c^(p-1) mod p^2) h_p mod p, h_p = ((p-1) q)^-1 mod p, likewise q, Garner.
 * Equal to decrypt1 for every unit c (tests/test_oracle_cross.py). */
int oracle_decrypt_crt_batch(const uint8_t* p_be, size_t p_len, const uint8_t* q_be, size_t q_len, size_t batch,
                             const uint8_t* c, size_t c_stride, uint8_t* m_out, size_t m_stride, int threads) {
  mpz_t p, q, p2, q2, p1, q1, hp, hq, pinv, t;
  mpz_inits(p, q, p2, q2, p1, q1, hp, hq, pinv, t, NULL);
  imp(p, p_be, p_len);
  imp(q, q_be, q_len);
  mpz_mul(p2, p, p);
  mpz_mul(q2, q, q);
  mpz_sub_ui(p1, p, 1);
  mpz_sub_ui(q1, q, 1);
  mpz_mul(t, p1, q); mpz_mod(t, t, p); mpz_invert(hp, t, p);
  mpz_mul(t, q1, p); mpz_mod(t, t, q); mpz_invert(hq, t, q);
  mpz_invert(pinv, p, q);
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    mpz_t ci, up, uq, mp, mq, h;
    mpz_inits(ci, up, uq, mp, mq, h, NULL);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (long i = 0; i < (long)batch; ++i) {
      imp(ci, c + (size_t)i * c_stride, c_stride);
      mpz_powm(up, ci, p1, p2);
      mpz_powm(uq, ci, q1, q2);
      mpz_sub_ui(up, up, 1); mpz_fdiv_q(up, up, p); mpz_mul(mp, up, hp); mpz_mod(mp, mp, p);
      mpz_sub_ui(uq, uq, 1); mpz_fdiv_q(uq, uq, q); mpz_mul(mq, uq, hq); mpz_mod(mq, mq, q);
      mpz_sub(h, mq, mp); mpz_mul(h, h, pinv); mpz_mod(h, h, q);
      mpz_mul(h, h, p); mpz_add(h, h, mp);
      expo(h, m_out + (size_t)i * m_stride, m_stride);
    }
    mpz_clears(ci, up, uq, mp, mq, h, NULL);
  }
  mpz_clears(p, q, p2, q2, p1, q1, hp, hq, pinv, t, NULL);
  return used;
}

/* paillier.go:206-218 EncryptWithRAtLevel at level two (s = 2: ns = n^2, ns1 = n^3): c = G^m * r^(n^2) mod n^3, verbatim. */
int oracle_encrypt_l2_batch(const uint8_t* n_be, size_t n_len, const uint8_t* g_be, size_t g_len, size_t batch,
                            const uint8_t* m, size_t m_stride, const uint8_t* r, size_t r_stride, uint8_t* c_out,
                            size_t c_stride, int threads) {
  mpz_t n, n2, n3, g;
  mpz_inits(n, n2, n3, g, NULL);
  imp(n, n_be, n_len);
  imp(g, g_be, g_len);
  mpz_mul(n2, n, n);
  mpz_mul(n3, n2, n);
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    mpz_t mi, ri, gm, rn, ci;
    mpz_inits(mi, ri, gm, rn, ci, NULL);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long i = 0; i < (long)batch; ++i) {
      imp(mi, m + (size_t)i * m_stride, m_stride);
      imp(ri, r + (size_t)i * r_stride, r_stride);
      gmp_exp(gm, g, mi, n3);    /* :213 Exp(G, m, ns1) */
      gmp_exp(rn, ri, n2, n3);   /* :214 Exp(r, ns, ns1) */
      mpz_mul(ci, gm, rn);       /* :216 */
      mpz_mod(ci, ci, n3);
      expo(ci, c_out + (size_t)i * c_stride, c_stride);
    }
    mpz_clears(mi, ri, gm, rn, ci, NULL);
  }
  mpz_clears(n, n2, n3, g, NULL);
  return used;
}

/* operations.go:96-118 NestedRandomize with the draws a, b supplied: an = a^n mod n^2, bn2 = b^(n^2) mod n^3,
 * r = ct^an * bn2 mod n^3 -- three mpz_powm, in the reference's order. */
int oracle_nested_randomize_batch(const uint8_t* n_be, size_t n_len, size_t batch, const uint8_t* ct, size_t ct_stride,
                                  const uint8_t* a, const uint8_t* b, size_t ab_stride, uint8_t* out, size_t out_stride,
                                  int threads) {
  mpz_t n, n2, n3;
  mpz_inits(n, n2, n3, NULL);
  imp(n, n_be, n_len);
  mpz_mul(n2, n, n);
  mpz_mul(n3, n2, n);
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    mpz_t ai, bi, an, bn2, r;
    mpz_inits(ai, bi, an, bn2, r, NULL);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long i = 0; i < (long)batch; ++i) {
      imp(r, ct + (size_t)i * ct_stride, ct_stride);
      imp(ai, a + (size_t)i * ab_stride, ab_stride);
      imp(bi, b + (size_t)i * ab_stride, ab_stride);
      gmp_exp(an, ai, n, n2);      /* :108 */
      gmp_exp(bn2, bi, n2, n3);    /* :109 */
      gmp_exp(r, r, an, n3);       /* :112 */
      mpz_mul(r, r, bn2);          /* :113 */
      mpz_mod(r, r, n3);           /* :114 */
      expo(r, out + (size_t)i * out_stride, out_stride);
    }
    mpz_clears(ai, bi, an, bn2, r, NULL);
  }
  mpz_clears(n, n2, n3, NULL);
  return used;
}

/* Threshold decryption of one ciphertext as the reference's own benchmark drives it (thresholdkey_test.go:396-427): t servers
 * call PartialDecrypt (thresholdkey.go:192-201: c^(2 delta s_i) mod n^2), then CombinePartialDecryptions (:149-161) --
 * computeLambda / updateLambda with Euclidean Div (:91-107), updateCprime / exp with ModInverse for a negative lambda
 * (:118-138), L and combineSharesConstant recomputed per call (:63-66,143-146).  ids: the servers' 1-based IDs; shares
 * concatenated big-endian with stride sh_stride.  partials_out (optional): t rows per ciphertext, server-major per ciphertext. */
int oracle_threshold_decrypt_batch(const uint8_t* n_be, size_t n_len, int total_servers, int t, const int* ids,
                                   const uint8_t* shares, size_t sh_stride, size_t batch, const uint8_t* c, size_t c_stride,
                                   uint8_t* m_out, size_t m_stride, uint8_t* partials_out, int threads) {
  mpz_t n, n2, delta, two_delta;
  mpz_inits(n, n2, delta, two_delta, NULL);
  imp(n, n_be, n_len);
  mpz_mul(n2, n, n);
  mpz_fac_ui(delta, (unsigned long)total_servers);          /* utils.go:17-23 Factorial */
  mpz_mul_ui(two_delta, delta, 2);
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    mpz_t ci, e, cprime, lambda, num, two_l, ret, tmp, l, cst;
    mpz_inits(ci, e, cprime, lambda, num, two_l, ret, tmp, l, cst, NULL);
    mpz_t* dec = (mpz_t*)malloc((size_t)t * sizeof(mpz_t));
    for (int k = 0; k < t; ++k) mpz_init(dec[k]);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long i = 0; i < (long)batch; ++i) {
      imp(ci, c + (size_t)i * c_stride, c_stride);
      for (int k = 0; k < t; ++k) {                          /* PartialDecrypt, one server after the other */
        imp(e, shares + (size_t)k * sh_stride, sh_stride);
        mpz_mul(e, e, two_delta);                            /* :195 exp = Share * (2 delta) */
        gmp_exp(dec[k], ci, e, n2);                          /* :197 */
        if (partials_out) expo(dec[k], partials_out + ((size_t)i * (size_t)t + (size_t)k) * c_stride, c_stride);
      }
      mpz_set_ui(cprime, 1);                                 /* :154 */
      for (int k = 0; k < t; ++k) {
        mpz_set(lambda, delta);                              /* computeLambda :104-112 */
        for (int j = 0; j < t; ++j) {
          if (ids[j] == ids[k]) continue;
          mpz_mul_si(num, lambda, -(long)ids[j]);            /* updateLambda :92 */
          long den = (long)ids[k] - (long)ids[j];
          /* gmp.Int.Div: Euclidean division (remainder in [0, |den|)): floor(num / den) for den > 0, and for den < 0
           * num = q den + r = (-q) |den| + r, i.e. q = -floor(num / |den|) */
          mpz_fdiv_q_ui(lambda, num, (unsigned long)(den > 0 ? den : -den));
          if (den < 0) mpz_neg(lambda, lambda);
        }
        mpz_mul_ui(two_l, lambda, 2);                        /* updateCprime :127 */
        if (mpz_sgn(two_l) < 0) {                            /* exp :134-140 */
          mpz_neg(tmp, two_l);
          gmp_exp(ret, dec[k], tmp, n2);
          mpz_invert(ret, ret, n2);
        } else {
          gmp_exp(ret, dec[k], two_l, n2);
        }
        mpz_mul(cprime, cprime, ret);                        /* :129-130 */
        mpz_mod(cprime, cprime, n2);
      }
      L_fn(l, cprime, n);                                    /* computeDecryption :144 */
      mpz_mul(tmp, delta, delta);                            /* combineSharesConstant :63-66, per call */
      mpz_mul_ui(tmp, tmp, 4);
      mpz_invert(cst, tmp, n);
      mpz_mul(l, cst, l);                                    /* :145 */
      mpz_mod(l, l, n);
      expo(l, m_out + (size_t)i * m_stride, m_stride);
    }
    for (int k = 0; k < t; ++k) mpz_clear(dec[k]);
    free(dec);
    mpz_clears(ci, e, cprime, lambda, num, two_l, ret, tmp, l, cst, NULL);
  }
  mpz_clears(n, n2, delta, two_delta, NULL);
  return used;
}

/* ---- SHA-256 (FIPS 180-4), for the Fiat-Shamir transcripts of random_oracle.go / thresholdkey.go:319-326 ---- */
typedef struct { uint32_t h[8]; uint8_t buf[64]; size_t fill; uint64_t total; } sha256_t;
static const uint32_t SK[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,
  0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,
  0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,0x27b70a85,
  0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,
  0xd192e819,0xd6990624,0xf40e3585,0x106aa070,0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,
  0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(sha256_t* s, const uint8_t* b) {
  uint32_t w[64], a[8];
  for (int i = 0; i < 16; ++i) w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
  for (int i = 16; i < 64; ++i) {
    uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  memcpy(a, s->h, sizeof a);
  for (int i = 0; i < 64; ++i) {
    uint32_t S1 = ROR(a[4], 6) ^ ROR(a[4], 11) ^ ROR(a[4], 25), ch = (a[4] & a[5]) ^ (~a[4] & a[6]);
    uint32_t t1 = a[7] + S1 + ch + SK[i] + w[i];
    uint32_t S0 = ROR(a[0], 2) ^ ROR(a[0], 13) ^ ROR(a[0], 22), mj = (a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]);
    uint32_t t2 = S0 + mj;
    a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
  }
  for (int i = 0; i < 8; ++i) s->h[i] += a[i];
}
static void sha256_init(sha256_t* s) {
  static const uint32_t iv[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
  memcpy(s->h, iv, sizeof iv);
  s->fill = 0; s->total = 0;
}
static void sha256_update(sha256_t* s, const uint8_t* d, size_t n) {
  s->total += n;
  while (n) {
    size_t k = 64 - s->fill; if (k > n) k = n;
    memcpy(s->buf + s->fill, d, k); s->fill += k; d += k; n -= k;
    if (s->fill == 64) { sha256_block(s, s->buf); s->fill = 0; }
  }
}
static void sha256_final(sha256_t* s, uint8_t out[32]) {
  uint64_t bits = s->total * 8;
  uint8_t pad = 0x80;
  sha256_update(s, &pad, 1);
  pad = 0;
  while (s->fill != 56) sha256_update(s, &pad, 1);
  uint8_t l[8];
  for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha256_update(s, l, 8);
  for (int i = 0; i < 8; ++i) { out[4 * i] = s->h[i] >> 24; out[4 * i + 1] = s->h[i] >> 16; out[4 * i + 2] = s->h[i] >> 8; out[4 * i + 3] = s->h[i]; }
}
/* h.Write(x.Bytes()): minimal big-endian magnitude, nothing for zero */
static void sha256_mpz(sha256_t* s, const mpz_t z) {
  if (mpz_sgn(z) == 0) return;
  size_t cnt = (mpz_sizeinbase(z, 2) + 7) / 8;
  uint8_t stack[1024], *b = cnt <= sizeof stack ? stack : (uint8_t*)malloc(cnt);
  mpz_export(b, NULL, 1, 1, 1, 0, z);
  sha256_update(s, b, cnt);
  if (b != stack) free(b);
}

/* random_oracle.go:10-32 RandomOracleBit(ct1, ct2, x, y, alpha): argument 0 (ct1) is skipped */
static int ro_bit4(const mpz_t ct2, const mpz_t x, const mpz_t y, const mpz_t alpha) {
  sha256_t s;
  uint8_t d[32];
  sha256_init(&s);
  sha256_mpz(&s, ct2); sha256_mpz(&s, x); sha256_mpz(&s, y); sha256_mpz(&s, alpha);
  sha256_final(&s, d);
  return d[31] & 1;   /* big.Int(SetBytes(res)) mod 2 */
}

/* ddleq.go:129-153 verifyDDLEQProofInstance for a batch of (statement, instance) pairs.  Strides: ct/alpha/f = s3 bytes,
 * x/y = s1, e = s2.  ok_out[i] = 1/0. */
int oracle_ddleq_verify_batch(const uint8_t* n_be, size_t n_len, size_t batch, const uint8_t* ct1, const uint8_t* ct2, size_t s3,
                              const uint8_t* x, const uint8_t* y, size_t s1, const uint8_t* alpha, const uint8_t* e, size_t s2,
                              const uint8_t* f, int32_t* ok_out, int threads) {
  mpz_t n, n2, n3;
  mpz_inits(n, n2, n3, NULL);
  imp(n, n_be, n_len);
  mpz_mul(n2, n, n);
  mpz_mul(n3, n2, n);
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    mpz_t c1, c2, xi, yi, al, ei, fi, en, fn2, chk;
    mpz_inits(c1, c2, xi, yi, al, ei, fi, en, fn2, chk, NULL);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long i = 0; i < (long)batch; ++i) {
      imp(c1, ct1 + (size_t)i * s3, s3); imp(c2, ct2 + (size_t)i * s3, s3);
      imp(xi, x + (size_t)i * s1, s1); imp(yi, y + (size_t)i * s1, s1);
      imp(al, alpha + (size_t)i * s3, s3); imp(ei, e + (size_t)i * s2, s2); imp(fi, f + (size_t)i * s3, s3);
      int bit = ro_bit4(c2, xi, yi, al);            /* :136 */
      mpz_set(chk, bit ? c2 : c1);                  /* :138-141 */
      gmp_exp(en, ei, n, n2);                       /* :143 */
      gmp_exp(fn2, fi, n2, n3);                     /* :144 */
      gmp_exp(chk, chk, en, n3);                    /* :146 */
      mpz_mul(chk, chk, fn2);                       /* :147-148 */
      mpz_mod(chk, chk, n3);
      ok_out[i] = mpz_cmp(al, chk) == 0;
    }
    mpz_clears(c1, c2, xi, yi, al, ei, fi, en, fn2, chk, NULL);
  }
  mpz_clears(n, n2, n3, NULL);
  return used;
}

/* paillier.go:308-340 recoveryAlgorithm for s = 2 (statement for statement) and :292-303 Decrypt at level two */
static void decrypt2(mpz_t m, const mpz_t c, const mpz_t n, const mpz_t n2, const mpz_t n3, const mpz_t lambda) {
  mpz_t a, i, t1, t2, amod, kf, mu;
  mpz_inits(a, i, t1, t2, amod, kf, mu, NULL);
  gmp_exp(a, c, lambda, n3);                         /* :296 */
  /* j = 1: nj = n, nj1 = n^2 */
  mpz_mod(amod, a, n2);
  L_fn(i, amod, n);                                  /* i = t1 (no inner loop for j = 1) */
  /* j = 2: nj = n^2, nj1 = n^3 */
  mpz_mod(amod, a, n3);
  L_fn(t1, amod, n);
  mpz_abs(t2, i);                                    /* :320 SetBytes(i.Bytes()): magnitude */
  mpz_sub_ui(i, i, 1);                               /* :323 (k = 2) */
  mpz_mul(t2, t2, i); mpz_mod(t2, t2, n2);           /* :324-325 */
  mpz_mul(t2, t2, n);                                /* :326  nk = n^(k-1) = n */
  mpz_set_ui(kf, 2); mpz_invert(kf, kf, n2);         /* :327-328  (k!)^-1 mod nj */
  mpz_mul(t2, t2, kf);                               /* :329 */
  mpz_sub(t2, t1, t2);                               /* :330 */
  mpz_mod(t1, t2, n2);                               /* :331 */
  /* i = t1 */
  mpz_invert(mu, lambda, n2);                        /* :298 */
  mpz_mul(m, t1, mu);
  mpz_mod(m, m, n2);
  mpz_clears(a, i, t1, t2, amod, kf, mu, NULL);
}

/* ddleq.go:55-127 proveDDLEQInstance with the draws x, y supplied (G = n + 1 as KeyGen sets it), for a batch of
 * (statement, instance) pairs.  Returns -1 if a statement fails the sanity check (the reference panics, :68).
 * Strides: ct1/ct2/alpha_out/f_out = s3, a/b/x/y = s1, e_out = s2. */
int oracle_ddleq_prove_batch(const uint8_t* n_be, size_t n_len, const uint8_t* lambda_be, size_t l_len, size_t batch,
                             const uint8_t* ct1, const uint8_t* ct2, size_t s3, const uint8_t* a, const uint8_t* b, const uint8_t* x,
                             const uint8_t* y, size_t s1, uint8_t* alpha_out, uint8_t* e_out, size_t s2, uint8_t* f_out,
                             int32_t* bit_out, int threads) {
  mpz_t n, n2, n3, lambda, g, nsinv;
  mpz_inits(n, n2, n3, lambda, g, nsinv, NULL);
  imp(n, n_be, n_len);
  imp(lambda, lambda_be, l_len);
  mpz_mul(n2, n, n);
  mpz_mul(n3, n2, n);
  mpz_add_ui(g, n, 1);
  int bad = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
  {
    mpz_t c1, c2, ai, bi, xi, yi, an, xn, yn2, al, t, e, f, s, v, gv, z, en, cc, ainv, nsi;
    mpz_inits(c1, c2, ai, bi, xi, yi, an, xn, yn2, al, t, e, f, s, v, gv, z, en, cc, ainv, nsi, NULL);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long i = 0; i < (long)batch; ++i) {
      imp(c1, ct1 + (size_t)i * s3, s3); imp(c2, ct2 + (size_t)i * s3, s3);
      imp(ai, a + (size_t)i * s1, s1); imp(bi, b + (size_t)i * s1, s1);
      imp(xi, x + (size_t)i * s1, s1); imp(yi, y + (size_t)i * s1, s1);
      gmp_exp(an, ai, n, n2);                        /* :62 */
      gmp_exp(t, c1, an, n3);
      gmp_exp(cc, bi, n2, n3);                       /* :63 */
      mpz_mul(t, t, cc); mpz_mod(t, t, n3);          /* :64-65 */
      if (mpz_cmp(t, c2) != 0) {                     /* :67-69 panic */
#ifdef _OPENMP
#pragma omp atomic write
#endif
        bad = 1;
        continue;
      }
      gmp_exp(xn, xi, n, n2);                        /* :81 */
      gmp_exp(yn2, yi, n2, n3);                      /* :82 */
      gmp_exp(al, c1, xn, n3);                       /* :85 */
      mpz_mul(al, al, yn2); mpz_mod(al, al, n3);     /* :86-87 */
      int bit = ro_bit4(c2, xi, yi, al);             /* :91 */
      mpz_set(e, xi);
      mpz_set(f, yi);
      if (bit) {
        mpz_invert(ainv, ai, n2);                    /* :96 */
        mpz_mul(e, e, ainv); mpz_mod(e, e, n2);      /* :97-98 */
        /* s = ExtractRandonness(ct1), level two (operations.go:75-91) */
        mpz_invert(nsi, n2, lambda);                 /* :80 */
        decrypt2(v, c1, n, n2, n3, lambda);          /* :82 */
        gmp_exp(gv, g, v, n3);                       /* :83 */
        mpz_invert(gv, gv, n3);                      /* :84 */
        mpz_mul(z, gv, c1); mpz_mod(z, z, n3);       /* :85-86 */
        gmp_exp(s, z, nsi, n);                       /* :88 */
        gmp_exp(an, ai, n, n2);                      /* ddleq.go:104 */
        gmp_exp(en, e, n, n2);                       /* :105 */
        gmp_exp(cc, s, an, n3);                      /* :107 */
        mpz_mul(cc, cc, bi);                         /* :108 (not reduced) */
        gmp_exp(cc, cc, en, n3);                     /* :109 */
        mpz_invert(cc, cc, n3);                      /* :110 */
        gmp_exp(t, s, xn, n3);
        mpz_mul(cc, cc, t);                          /* :112 */
        mpz_mul(f, f, cc);                           /* :113 */
        mpz_mod(f, f, n3);                           /* :114 */
      }
      expo(al, alpha_out + (size_t)i * s3, s3);
      expo(e, e_out + (size_t)i * s2, s2);
      expo(f, f_out + (size_t)i * s3, s3);
      if (bit_out) bit_out[i] = bit;
    }
    mpz_clears(c1, c2, ai, bi, xi, yi, an, xn, yn2, al, t, e, f, s, v, gv, z, en, cc, ainv, nsi, NULL);
  }
  mpz_clears(n, n2, n3, lambda, g, nsinv, NULL);
  return bad ? -1 : 0;
}

/* ---- round 5: the rows of SURVEY 8(a) that had no CPU figure (Add / Sub / ConstMult, AltEncrypt, the share ZKP) ---------------- */

#ifdef _OPENMP
#define ORACLE_PAR_BEGIN(threads, used) \
  if ((threads) > 0) omp_set_num_threads(threads); \
  _Pragma("omp parallel") {                          \
    _Pragma("omp single") (used) = omp_get_num_threads();
#define ORACLE_FOR _Pragma("omp for schedule(dynamic, 4)")
#else
#define ORACLE_PAR_BEGIN(threads, used) { (void)(threads);
#define ORACLE_FOR
#endif
#define ORACLE_PAR_END }

/* operations.go:11-29 Add(a, b) and :32-55 Sub(a, b) with two operands, level one or two (mod = n^(s+1)).
 * Add: accumulator 1, then Mod(Mul(acc, c.C)) per operand.  Sub: accumulator = a, then ModInverse(b) and Mod(Mul(acc, neg)).
 * sub != 0 selects Sub.  A b that has no inverse: mpz_invert returns 0 and the reference goes on with an undefined value --
 * ok_out[i] = 0 flags the lane and the row is written as zero. */
int oracle_add_sub_batch(const uint8_t* mod_be, size_t mod_len, int sub, size_t batch, const uint8_t* a, size_t a_stride,
                         const uint8_t* b, size_t b_stride, uint8_t* out, size_t o_stride, int32_t* ok_out, int threads) {
  mpz_t ns1;
  mpz_init(ns1);
  imp(ns1, mod_be, mod_len);
  int used = 1;
  ORACLE_PAR_BEGIN(threads, used)
    mpz_t ai, bi, acc, neg;
    mpz_inits(ai, bi, acc, neg, NULL);
    ORACLE_FOR
    for (long i = 0; i < (long)batch; ++i) {
      imp(ai, a + (size_t)i * a_stride, a_stride);
      imp(bi, b + (size_t)i * b_stride, b_stride);
      int ok = 1;
      if (!sub) {
        mpz_set_ui(acc, 1);                                    /* :12 */
        mpz_mul(acc, acc, ai); mpz_mod(acc, acc, ns1);         /* :18-21, first operand */
        mpz_mul(acc, acc, bi); mpz_mod(acc, acc, ns1);         /* second operand */
      } else {
        mpz_set(acc, ai);                                      /* :34 */
        ok = mpz_invert(neg, bi, ns1) != 0;                    /* :43 */
        if (ok) { mpz_mul(acc, acc, neg); mpz_mod(acc, acc, ns1); }   /* :44-47 */
        else mpz_set_ui(acc, 0);
      }
      if (ok_out) ok_out[i] = ok;
      expo(acc, out + (size_t)i * o_stride, o_stride);
    }
    mpz_clears(ai, bi, acc, neg, NULL);
  ORACLE_PAR_END
  mpz_clear(ns1);
  return used;
}

/* operations.go:58-64 ConstMult: Exp(ct.C, k, n^(s+1)).  k_stride == 0: one shared k (BenchmarkConstMul2's shape,
 * operations_test.go:186-198); otherwise k[i] per ciphertext. */
int oracle_const_mult_batch(const uint8_t* mod_be, size_t mod_len, size_t batch, const uint8_t* c, size_t c_stride, const uint8_t* k,
                            size_t k_len, size_t k_stride, uint8_t* out, size_t o_stride, int threads) {
  mpz_t ns1;
  mpz_init(ns1);
  imp(ns1, mod_be, mod_len);
  int used = 1;
  ORACLE_PAR_BEGIN(threads, used)
    mpz_t ci, ki, m;
    mpz_inits(ci, ki, m, NULL);
    ORACLE_FOR
    for (long i = 0; i < (long)batch; ++i) {
      imp(ci, c + (size_t)i * c_stride, c_stride);
      imp(ki, k + (size_t)i * k_stride, k_len);
      gmp_exp(m, ci, ki, ns1);                                 /* :62 */
      expo(m, out + (size_t)i * o_stride, o_stride);
    }
    mpz_clears(ci, ki, m, NULL);
  ORACLE_PAR_END
  mpz_clear(ns1);
  return used;
}

/* paillier.go:221-238 AltEncryptWithRAtLevel at level one: h1 = (N - H)^N mod N^2 once per key (the reference caches it,
 * :416-425), then per ciphertext r.Mod(r, K), gm = Exp(G, m, n^2), hr = Exp(h, r, n^2), c = Mod(Mul(gm, hr), n^2).
 * r_red_out (optional, r_stride bytes per row) receives r mod K -- the reference overwrites the caller's r. */
int oracle_alt_encrypt_batch(const uint8_t* n_be, size_t n_len, const uint8_t* g_be, size_t g_len, const uint8_t* h_be, size_t h_len,
                             const uint8_t* k_be, size_t k_len, size_t batch, const uint8_t* m, size_t m_stride, const uint8_t* r,
                             size_t r_stride, uint8_t* c_out, size_t c_stride, uint8_t* r_red_out, int threads) {
  mpz_t n, n2, g, H, K, h1;
  mpz_inits(n, n2, g, H, K, h1, NULL);
  imp(n, n_be, n_len); imp(g, g_be, g_len); imp(H, h_be, h_len); imp(K, k_be, k_len);
  mpz_mul(n2, n, n);
  mpz_sub(h1, n, H);                                           /* :419 */
  gmp_exp(h1, h1, n, n2);                                      /* :420 */
  int used = 1;
  ORACLE_PAR_BEGIN(threads, used)
    mpz_t mi, ri, gm, hr, ci;
    mpz_inits(mi, ri, gm, hr, ci, NULL);
    ORACLE_FOR
    for (long i = 0; i < (long)batch; ++i) {
      imp(mi, m + (size_t)i * m_stride, m_stride);
      imp(ri, r + (size_t)i * r_stride, r_stride);
      mpz_mod(ri, ri, K);                                      /* :228 */
      gmp_exp(gm, g, mi, n2);                                  /* :233 */
      gmp_exp(hr, h1, ri, n2);                                 /* :234 */
      mpz_mul(ci, gm, hr); mpz_mod(ci, ci, n2);                /* :236 */
      expo(ci, c_out + (size_t)i * c_stride, c_stride);
      if (r_red_out) expo(ri, r_red_out + (size_t)i * r_stride, r_stride);
    }
    mpz_clears(mi, ri, gm, hr, ci, NULL);
  ORACLE_PAR_END
  mpz_clears(n, n2, g, H, K, h1, NULL);
  return used;
}

/* thresholdkey.go:319-326 computeHash(a, b, c4, ci2) = SetBytes(SHA-256(a.Bytes() | b.Bytes() | c4.Bytes() | ci2.Bytes())) */
static void zkp_hash(uint8_t d[32], const mpz_t a, const mpz_t b, const mpz_t c4, const mpz_t ci2) {
  sha256_t s;
  sha256_init(&s);
  sha256_mpz(&s, a); sha256_mpz(&s, b); sha256_mpz(&s, c4); sha256_mpz(&s, ci2);
  sha256_final(&s, d);
}

/* thresholdkey.go:225-257 PartialDecryptionWithZKP with the random r supplied (the reference draws it at :233):
 * Decryption = c^(2 delta s) mod n^2 (PartialDecrypt :192-201); c4 = c^4 and ci2 = Decryption^2 UNREDUCED (Exp(., ., nil));
 * a = c4^r, b = V^r mod n^2; E = hash; Z = r + E delta s (computeZ :313-317).  e_out: 32 bytes per proof. */
int oracle_share_zkp_prove_batch(const uint8_t* n_be, size_t n_len, int total_servers, const uint8_t* share_be, size_t share_len,
                                 const uint8_t* v_be, size_t v_len, size_t batch, const uint8_t* c, size_t c_stride, const uint8_t* r,
                                 size_t r_stride, uint8_t* dec_out, size_t dec_stride, uint8_t* e_out, uint8_t* z_out, size_t z_stride,
                                 int threads) {
  mpz_t n, n2, share, V, delta;
  mpz_inits(n, n2, share, V, delta, NULL);
  imp(n, n_be, n_len); imp(share, share_be, share_len); imp(V, v_be, v_len);
  mpz_mul(n2, n, n);
  mpz_fac_ui(delta, (unsigned long)total_servers);
  int used = 1;
  ORACLE_PAR_BEGIN(threads, used)
    mpz_t ci, ri, ex, dec, c4, a, b, ci2, E, Z;
    mpz_inits(ci, ri, ex, dec, c4, a, b, ci2, E, Z, NULL);
    ORACLE_FOR
    for (long i = 0; i < (long)batch; ++i) {
      uint8_t d[32];
      imp(ci, c + (size_t)i * c_stride, c_stride);
      imp(ri, r + (size_t)i * r_stride, r_stride);
      mpz_mul_ui(ex, delta, 2); mpz_mul(ex, share, ex);        /* :195 */
      gmp_exp(dec, ci, ex, n2);                                /* :199 */
      mpz_pow_ui(c4, ci, 4);                                   /* :241 Exp(c, 4, nil) */
      gmp_exp(a, c4, ri, n2);                                  /* :242 */
      gmp_exp(b, V, ri, n2);                                   /* :245 */
      mpz_pow_ui(ci2, dec, 2);                                 /* :248 */
      zkp_hash(d, a, b, c4, ci2);                              /* :250 */
      imp(E, d, 32);
      mpz_mul(Z, E, delta); mpz_mul(Z, Z, share); mpz_add(Z, ri, Z);   /* :314-316 */
      expo(dec, dec_out + (size_t)i * dec_stride, dec_stride);
      memcpy(e_out + (size_t)i * 32, d, 32);
      expo(Z, z_out + (size_t)i * z_stride, z_stride);
    }
    mpz_clears(ci, ri, ex, dec, c4, a, b, ci2, E, Z, NULL);
  ORACLE_PAR_END
  mpz_clears(n, n2, share, V, delta, NULL);
  return used;
}

/* thresholdkey.go:278-311 VerifyProof: a = (c^4)^Z ((c_i^2)^E)^-1, b = V^Z (v_i^E)^-1 mod n^2 (verifyPart1 / verifyPart2),
 * E == SHA-256(a | b | c^4 | c_i^2).  A (c_i^2)^E without an inverse: mpz_invert fails, the reference's a is undefined --
 * ok_out[i] = 0. */
int oracle_share_zkp_verify_batch(const uint8_t* n_be, size_t n_len, const uint8_t* v_be, size_t v_len, const uint8_t* vi_be,
                                  size_t vi_len, size_t batch, const uint8_t* c, size_t c_stride, const uint8_t* dec, size_t dec_stride,
                                  const uint8_t* e, const uint8_t* z, size_t z_stride, int32_t* ok_out, uint8_t* ab_out, int threads) {
  /* ab_out (optional): a | b of verifyPart1 / verifyPart2 per proof, c_stride bytes each (what TestVerifyPart1/2 pin) */
  mpz_t n, n2, V, vi;
  mpz_inits(n, n2, V, vi, NULL);
  imp(n, n_be, n_len); imp(V, v_be, v_len); imp(vi, vi_be, vi_len);
  mpz_mul(n2, n, n);
  int used = 1;
  ORACLE_PAR_BEGIN(threads, used)
    mpz_t ci, di, E, Z, c4, d2, a1, a2, a, b1, b2, b;
    mpz_inits(ci, di, E, Z, c4, d2, a1, a2, a, b1, b2, b, NULL);
    ORACLE_FOR
    for (long i = 0; i < (long)batch; ++i) {
      uint8_t d[32];
      imp(ci, c + (size_t)i * c_stride, c_stride);
      imp(di, dec + (size_t)i * dec_stride, dec_stride);
      imp(E, e + (size_t)i * 32, 32);
      imp(Z, z + (size_t)i * z_stride, z_stride);
      mpz_pow_ui(c4, ci, 4);                                   /* :294 */
      mpz_pow_ui(d2, di, 2);                                   /* :295 */
      gmp_exp(a1, c4, Z, n2);                                  /* :297 */
      gmp_exp(a2, d2, E, n2);                                  /* :298 */
      int ok = mpz_invert(a2, a2, n2) != 0;                    /* :299 */
      mpz_mul(a, a1, a2); mpz_mod(a, a, n2);                   /* :300 */
      gmp_exp(b1, V, Z, n2);                                   /* :306 */
      gmp_exp(b2, vi, E, n2);                                  /* :307 */
      ok = (mpz_invert(b2, b2, n2) != 0) && ok;                /* :308 */
      mpz_mul(b, b1, b2); mpz_mod(b, b, n2);                   /* :309 */
      zkp_hash(d, a, b, c4, d2);                               /* :281-288 */
      ok_out[i] = ok && memcmp(d, e + (size_t)i * 32, 32) == 0;   /* :290-291 */
      if (ab_out) { expo(a, ab_out + (size_t)i * 2 * c_stride, c_stride); expo(b, ab_out + ((size_t)i * 2 + 1) * c_stride, c_stride); }
    }
    mpz_clears(ci, di, E, Z, c4, d2, a1, a2, a, b1, b2, b, NULL);
  ORACLE_PAR_END
  mpz_clears(n, n2, V, vi, NULL);
  return used;
}

/* paillier.go:292-340 SecretKey.Decrypt at level two (what NestedDecrypt runs first, :344-372) for a batch: decrypt2 above, verbatim. */
int oracle_decrypt_l2_batch(const uint8_t* n_be, size_t n_len, const uint8_t* lambda_be, size_t l_len, size_t batch, const uint8_t* c,
                            size_t c_stride, uint8_t* m_out, size_t m_stride, int threads) {
  mpz_t n, n2, n3, lambda;
  mpz_inits(n, n2, n3, lambda, NULL);
  imp(n, n_be, n_len); imp(lambda, lambda_be, l_len);
  mpz_mul(n2, n, n); mpz_mul(n3, n2, n);
  int used = 1;
  ORACLE_PAR_BEGIN(threads, used)
    mpz_t ci, mi;
    mpz_inits(ci, mi, NULL);
    ORACLE_FOR
    for (long i = 0; i < (long)batch; ++i) {
      imp(ci, c + (size_t)i * c_stride, c_stride);
      decrypt2(mi, ci, n, n2, n3, lambda);
      expo(mi, m_out + (size_t)i * m_stride, m_stride);
    }
    mpz_clears(ci, mi, NULL);
  ORACLE_PAR_END
  mpz_clears(n, n2, n3, lambda, NULL);
  return used;
}

const char* oracle_gmp_version(void) { return gmp_version; }
