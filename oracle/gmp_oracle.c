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

const char* oracle_gmp_version(void) { return gmp_version; }
