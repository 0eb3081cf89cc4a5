/* test_cabi.c -- the drop-in boundary exercised the way cgo would call it: plain C99, caller-owned flat big-endian buffers,
 * int status codes, no C++ and no Python in between.  Calls EVERY pgpu_* batch entry point of include/paillier_hip.h at
 * least once and checks results against vectors the pytest driver (tests/test_gpu_c_abi.py) exports from the committed
 * fixtures (the JSON files under tests/golden) as "key hex hex ..." lines.
 *
 *   gcc -std=c99 -pedantic -Wall -Werror -pthread tests/c/test_cabi.c -Iinclude -Lpaillier_amd -lpaillier_hip -o test_cabi
 *
 * The last section drives TWO contexts from TWO pthreads through the sharded flows of go/sharded.go (ShardedGPU: one context per
 * device, one goroutine per device, slices by shard_slice, the threshold exchange through host memory): the ABI is
 * thread-compatible per context the way cgo will use it.  On the one-GPU test box both contexts sit on device 0.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "paillier_hip.h"

#define MAXV 128
typedef struct { char key[32]; int n; char* hex[MAXV]; } entry_t;
static entry_t g_ent[64];
static int g_nent = 0;

static int fail_line = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s  [%s]\n", __FILE__, __LINE__, #cond, pgpu_last_error()); fail_line = __LINE__; return 1; } } while (0)
#define OK(call) CHECK((call) == PGPU_OK)

static void load(const char* path) {
  FILE* f = fopen(path, "r");
  static char line[1 << 20];
  if (!f) { perror(path); exit(2); }
  while (fgets(line, sizeof line, f)) {
    char* tok = strtok(line, " \n");
    entry_t* e;
    if (!tok) continue;
    e = &g_ent[g_nent++];
    strncpy(e->key, tok, sizeof e->key - 1);
    e->n = 0;
    while ((tok = strtok(NULL, " \n")) != NULL && e->n < MAXV) {
      e->hex[e->n] = (char*)malloc(strlen(tok) + 1);
      strcpy(e->hex[e->n++], tok);
    }
  }
  fclose(f);
}

static const entry_t* get(const char* key) {
  int i;
  for (i = 0; i < g_nent; ++i) if (strcmp(g_ent[i].key, key) == 0) return &g_ent[i];
  fprintf(stderr, "missing vector %s\n", key);
  exit(2);
}

static int hexval(char c) { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; }

/* hex -> big-endian bytes right-aligned in `stride` bytes (what Go builds from gmp.Int.Bytes()) */
static void put(const char* hex, uint8_t* out, size_t stride) {
  size_t len = strlen(hex), i;
  memset(out, 0, stride);
  for (i = 0; i < len; ++i) {
    size_t nib = len - 1 - i;              /* nibble significance */
    if (nib / 2 >= stride) { fprintf(stderr, "value wider than its stride\n"); exit(2); }
    out[stride - 1 - nib / 2] |= (uint8_t)(hexval(hex[i]) << (4 * (nib % 2)));
  }
}

static uint8_t* pack(const entry_t* e, size_t stride) {
  uint8_t* b = (uint8_t*)calloc((size_t)e->n, stride);
  int i;
  for (i = 0; i < e->n; ++i) put(e->hex[i], b + (size_t)i * stride, stride);
  return b;
}

static int same(const uint8_t* got, const entry_t* want, size_t stride) {
  uint8_t* w = pack(want, stride);
  int r = memcmp(got, w, (size_t)want->n * stride) == 0;
  free(w);
  return r;
}

static size_t bytes_of(const char* hex) { return (strlen(hex) + 1) / 2; }

/* ---- go/sharded.go in C: `world` workers, one context (own stream) and one set of key handles each -------------------------- */
static void shard_slice(size_t total, size_t r, size_t w, size_t* b, size_t* e) {     /* paillier_amd/dist.py shard_slice */
  size_t base = total / w, rem = total % w;
  *b = r * base + (r < rem ? r : rem);
  *e = *b + base + (r < rem ? 1 : 0);
}

typedef struct {
  int rank, world, device;
  pthread_barrier_t* bar;
  /* keys (big-endian) */
  const uint8_t *n, *g, *lambda, *tn, *tg; size_t nl, ll, tl;
  const uint8_t* const* shares; const size_t* share_lens;
  /* Decrypt: nd ciphertexts -> plaintexts */
  const uint8_t* dc; size_t nd; uint8_t* dm;
  /* threshold: nt ciphertexts; parts = the exchange buffer in host memory, server-major [3][nt][2 tl]; tm = plaintexts */
  const uint8_t* tc; size_t nt; uint8_t* parts; uint8_t* tm;
  int rc; char err[256];
} worker_t;

#define WOK(call) do { int rc_ = (call); if (rc_ != PGPU_OK) { w->rc = rc_ ? rc_ : -100; snprintf(w->err, sizeof w->err, "line %d: %s", __LINE__, pgpu_last_error()); goto done; } } while (0)

static void* sharded_worker(void* arg) {
  worker_t* w = (worker_t*)arg;
  pgpu_ctx* ctx = NULL; pgpu_pubkey *pk = NULL, *tpk = NULL; pgpu_seckey* sk = NULL;
  size_t b, e, ub, ue, s, tcb = 2 * w->tl;
  static const int ids[3] = {1, 3, 5};
  int crossed = 0;
  w->rc = PGPU_OK;
  WOK(pgpu_ctx_create(w->device, PGPU_STREAM_NEW, &ctx));
  WOK(pgpu_pubkey_create(ctx, w->n, w->nl, w->g, w->nl, NULL, 0, NULL, 0, &pk));
  WOK(pgpu_seckey_create(ctx, pk, w->lambda, w->ll, &sk));
  WOK(pgpu_pubkey_create(ctx, w->tn, w->tl, w->tg, w->tl, NULL, 0, NULL, 0, &tpk));
  /* ShardedSecretKey.DecryptBatch: this worker's slice of the ciphertexts */
  shard_slice(w->nd, (size_t)w->rank, (size_t)w->world, &b, &e);
  if (e > b) WOK(pgpu_decrypt(sk, PGPU_LEVEL_ONE, e - b, w->dc + b * 2 * w->nl, 2 * w->nl, w->dm + b * w->nl, w->nl, PGPU_MEM_HOST, PGPU_DECRYPT_DEFAULT, NULL));
  /* ShardedPublicKey.ThresholdDecryptBatch, step 1: this worker's (server, ciphertext) unit range, straight into the exchange buffer */
  shard_slice(3 * w->nt, (size_t)w->rank, (size_t)w->world, &ub, &ue);
  if (ue > ub) WOK(pgpu_partial_decrypt_units(tpk, 5, 3, w->shares, w->share_lens, w->nt, w->tc, tcb, ub, ue, w->parts + ub * tcb, tcb, PGPU_MEM_HOST));
  /* step 2: the exchange -- every worker's partials are in host memory once all have passed here */
  pthread_barrier_wait(w->bar);
  crossed = 1;
  /* step 3: combine this worker's ciphertext slice */
  shard_slice(w->nt, (size_t)w->rank, (size_t)w->world, &b, &e);
  if (e > b) {
    const uint8_t* cols[3];
    for (s = 0; s < 3; ++s) cols[s] = w->parts + (s * w->nt + b) * tcb;
    WOK(pgpu_combine_partial_decryptions(tpk, 5, 3, 3, ids, e - b, cols, tcb, w->tm + b * w->tl, w->tl, PGPU_MEM_HOST, NULL));
  }
done:
  if (!crossed) pthread_barrier_wait(w->bar);          /* a failed worker must not leave the others waiting */
  pgpu_seckey_destroy(sk);
  pgpu_pubkey_destroy(pk);
  pgpu_pubkey_destroy(tpk);
  pgpu_ctx_destroy(ctx);
  return NULL;
}

int main(int argc, char** argv) {
  pgpu_ctx* ctx = NULL;
  pgpu_pubkey *pk = NULL, *tpk = NULL;
  pgpu_seckey* sk = NULL;
  pgpu_modulus* mod = NULL;
  uint8_t nb[512], gb[512], lb[512], tnb[512];
  size_t nl, ll, pb, cb, cb3, pb2, B;
  uint8_t *m, *r, *c, *out, *a, *b, *k0;
  int32_t st[MAXV];
  double ms, mads;
  int launches;
  const entry_t* e;

  if (argc < 2) { fprintf(stderr, "usage: test_cabi vectors.txt\n"); return 2; }
  load(argv[1]);
  printf("%s\n", pgpu_version());
  OK(pgpu_ctx_create(0, NULL, &ctx));
  CHECK(pgpu_ctx_set_flag(ctx, "asm", 1) == PGPU_OK && pgpu_ctx_set_flag(ctx, "no-such-flag", 1) == PGPU_ERR_INVALID);

  /* ---- keys */
  nl = bytes_of(get("n")->hex[0]); put(get("n")->hex[0], nb, nl);
  put(get("g")->hex[0], gb, nl);
  ll = bytes_of(get("lambda")->hex[0]); put(get("lambda")->hex[0], lb, ll);
  OK(pgpu_pubkey_create(ctx, nb, nl, gb, nl, NULL, 0, NULL, 0, &pk));
  OK(pgpu_seckey_create(ctx, pk, lb, ll, &sk));
  CHECK(pgpu_seckey_has_crt(sk) == 1);
  pb = pgpu_pubkey_plain_bytes(pk, PGPU_LEVEL_ONE); cb = pgpu_pubkey_cipher_bytes(pk, PGPU_LEVEL_ONE);
  pb2 = pgpu_pubkey_plain_bytes(pk, PGPU_LEVEL_TWO); cb3 = pgpu_pubkey_cipher_bytes(pk, PGPU_LEVEL_TWO);
  CHECK(pb == nl && cb == 2 * nl && pb2 == 2 * nl && cb3 == 3 * nl);

  /* ---- EncryptWithR / Decrypt (paillier.go:206-218,292-303) */
  e = get("enc_m"); B = (size_t)e->n;
  m = pack(e, pb); r = pack(get("enc_r"), pb); c = (uint8_t*)calloc(B, cb); out = (uint8_t*)calloc(B, cb3);
  OK(pgpu_encrypt_with_r(pk, PGPU_LEVEL_ONE, B, m, pb, r, pb, c, cb, PGPU_MEM_HOST));
  CHECK(same(c, get("enc_c"), cb));
  OK(pgpu_ctx_last_profile(ctx, &ms, &launches, &mads));
  CHECK(launches >= 1 && mads > 0 && strncmp(pgpu_ctx_last_kernel(ctx), "vm_", 3) == 0 && pgpu_ctx_last_vm_asm(ctx) >= 1 &&
        pgpu_ctx_last_vm_launches(ctx) >= pgpu_ctx_last_vm_asm(ctx));
  OK(pgpu_decrypt(sk, PGPU_LEVEL_ONE, B, c, cb, out, pb, PGPU_MEM_HOST, PGPU_DECRYPT_DEFAULT, st));
  CHECK(memcmp(out, m, B * pb) == 0 && st[0] == PGPU_LANE_OK);
  /* the key holder's EncryptWithR (SecretKey embeds PublicKey, paillier.go:59-62): r^n through p^2, q^2 -- the same ciphertexts */
  { uint8_t* c2 = (uint8_t*)calloc(B, cb);
    OK(pgpu_encrypt_with_r_sk(sk, PGPU_LEVEL_ONE, B, m, pb, r, pb, c2, cb, PGPU_MEM_HOST));
    CHECK(memcmp(c2, c, B * cb) == 0 && same(c2, get("enc_c"), cb));
    free(c2); }
  free(c);
  e = get("dec_c");
  c = pack(e, cb);
  OK(pgpu_decrypt(sk, PGPU_LEVEL_ONE, (size_t)e->n, c, cb, out, pb, PGPU_MEM_HOST, PGPU_DECRYPT_DEFAULT, st));
  CHECK(same(out, get("dec_m"), pb) && st[8] == PGPU_LANE_NONUNIT);         /* vector 8 is c = 0 */
  OK(pgpu_decrypt(sk, PGPU_LEVEL_ONE, (size_t)e->n, c, cb, out, pb, PGPU_MEM_HOST, PGPU_DECRYPT_NO_CRT, NULL));
  CHECK(same(out, get("dec_m"), pb));
  free(c);

  /* ---- Encrypt with library randomness: Decrypt(Encrypt(m)) == m and c == EncryptWithR(m, r_out) */
  c = (uint8_t*)calloc(B, cb);
  { uint8_t* rr = (uint8_t*)calloc(B, pb); uint8_t* c2 = (uint8_t*)calloc(B, cb);
    OK(pgpu_encrypt(pk, PGPU_LEVEL_ONE, B, m, pb, c, cb, rr, pb, PGPU_MEM_HOST));
    OK(pgpu_encrypt_with_r(pk, PGPU_LEVEL_ONE, B, m, pb, rr, pb, c2, cb, PGPU_MEM_HOST));
    CHECK(memcmp(c, c2, B * cb) == 0);
    OK(pgpu_random_units(pk, B, rr, pb, PGPU_MEM_HOST));
    free(rr); free(c2); }
  OK(pgpu_decrypt(sk, PGPU_LEVEL_ONE, B, c, cb, out, pb, PGPU_MEM_HOST, 0, NULL));
  CHECK(memcmp(out, m, B * pb) == 0);
  free(c);

  /* ---- Add / Sub / ConstMult (operations.go:11-64) */
  e = get("add_a"); B = (size_t)e->n;
  a = pack(e, cb); b = pack(get("add_b"), cb);
  OK(pgpu_add(pk, PGPU_LEVEL_ONE, B, a, cb, b, cb, out, cb, PGPU_MEM_HOST));
  CHECK(same(out, get("add_out"), cb));
  { const uint8_t* ops[3]; uint8_t* t = (uint8_t*)calloc(B, cb);
    ops[0] = a; ops[1] = b; ops[2] = b;
    OK(pgpu_add_many(pk, PGPU_LEVEL_ONE, 3, B, ops, cb, t, cb, PGPU_MEM_HOST));          /* a b b */
    ops[0] = t; ops[1] = b;
    OK(pgpu_sub_many(pk, PGPU_LEVEL_ONE, 2, B, ops, cb, t, cb, PGPU_MEM_HOST, st));      /* / b  */
    CHECK(same(t, get("add_out"), cb) && st[0] == PGPU_LANE_OK);
    OK(pgpu_sub(pk, PGPU_LEVEL_ONE, B, t, cb, b, cb, t, cb, PGPU_MEM_HOST, NULL));        /* / b  = a mod n^2 */
    OK(pgpu_add_many(pk, PGPU_LEVEL_ONE, 1, B, (const uint8_t* const*)&a, cb, out, cb, PGPU_MEM_HOST));
    CHECK(memcmp(t, out, B * cb) == 0);
    free(t); }
  k0 = (uint8_t*)calloc(1, pb); put(get("cm_k0")->hex[0], k0, pb);
  OK(pgpu_const_mult(pk, PGPU_LEVEL_ONE, B, a, cb, k0, pb, 0, out, cb, PGPU_MEM_HOST));
  CHECK(same(out, get("cm_out"), cb));

  /* ---- the gmp.Int seam: Exp / Mul+Mod / ModInverse */
  { uint8_t n2b[1024]; const entry_t* n2 = get("n2"); size_t n2l = bytes_of(n2->hex[0]); uint8_t* inv = (uint8_t*)calloc(B, cb);
    put(n2->hex[0], n2b, n2l);
    OK(pgpu_modulus_create(ctx, n2b, n2l, &mod));
    CHECK(pgpu_modulus_bytes(mod) == cb);
    OK(pgpu_modexp(mod, B, a, cb, cb, k0, pb, 0, out, cb, PGPU_MEM_HOST));
    CHECK(same(out, get("cm_out"), cb));
    OK(pgpu_modmul(mod, B, a, cb, cb, b, cb, cb, out, cb, PGPU_MEM_HOST));
    CHECK(same(out, get("add_out"), cb));
    OK(pgpu_modinv(mod, B, b, cb, cb, inv, cb, PGPU_MEM_HOST, st));
    OK(pgpu_modmul(mod, B, out, cb, cb, inv, cb, cb, out, cb, PGPU_MEM_HOST));             /* a b b^-1 = a mod n^2 */
    OK(pgpu_add_many(pk, PGPU_LEVEL_ONE, 1, B, (const uint8_t* const*)&a, cb, inv, cb, PGPU_MEM_HOST));
    CHECK(memcmp(out, inv, B * cb) == 0 && st[0] == PGPU_LANE_OK);
    free(inv); }

  /* ---- level two and AltEncrypt */
  { const entry_t* l2m = get("l2_m"); size_t n2 = (size_t)l2m->n; uint8_t* mm = pack(l2m, pb2); uint8_t* rr = pack(get("l2_r"), pb);
    uint8_t* cc = (uint8_t*)calloc(n2, cb3);
    OK(pgpu_encrypt_with_r(pk, PGPU_LEVEL_TWO, n2, mm, pb2, rr, pb, cc, cb3, PGPU_MEM_HOST));
    CHECK(same(cc, get("l2_c"), cb3));
    OK(pgpu_decrypt(sk, PGPU_LEVEL_TWO, n2, cc, cb3, out, pb2, PGPU_MEM_HOST, 0, NULL));
    CHECK(memcmp(out, mm, n2 * pb2) == 0);
    free(mm); free(rr); free(cc); }
  { pgpu_pubkey* apk = NULL; uint8_t hb[512], kb[512]; size_t hl = bytes_of(get("h")->hex[0]), kl = bytes_of(get("k")->hex[0]);
    uint8_t* cc = (uint8_t*)calloc(B, cb); uint8_t* red = (uint8_t*)calloc(B, pb);
    put(get("h")->hex[0], hb, hl); put(get("k")->hex[0], kb, kl);
    OK(pgpu_pubkey_create(ctx, nb, nl, gb, nl, hb, hl, kb, kl, &apk));
    OK(pgpu_alt_encrypt_with_r(apk, PGPU_LEVEL_ONE, B, m, pb, r, pb, cc, cb, red, PGPU_MEM_HOST));
    OK(pgpu_decrypt(sk, PGPU_LEVEL_ONE, B, cc, cb, out, pb, PGPU_MEM_HOST, 0, NULL));      /* same n: decrypts under sk */
    CHECK(memcmp(out, m, B * pb) == 0);
    pgpu_pubkey_destroy(apk); free(cc); free(red); }

  /* ---- wire format (paillier.go:374-401): Bytes() of the committed ciphertexts and back, host and device-kernel paths agree */
  { size_t W = (size_t)get("enc_c")->n; size_t cap = W * pgpu_gob_max_bytes(cb); uint8_t* blobs = (uint8_t*)calloc(cap, 1); uint8_t* back = (uint8_t*)calloc(W, cb + 8);
    size_t* offs = (size_t*)calloc(W + 1, sizeof(size_t)); int32_t lv[MAXV], me[MAXV]; size_t i;
    uint8_t* encc = pack(get("enc_c"), cb);
    OK(pgpu_gob_pack(ctx, W, encc, cb, PGPU_MEM_HOST, PGPU_LEVEL_ONE, 0, blobs, cap, offs));
    CHECK(offs[0] == 0 && offs[W] <= cap && offs[1] > cb);
    CHECK(blobs[offs[1] - 1] == 0 && blobs[offs[W] - 1] == 0);                          /* every value message ends its struct */
    OK(pgpu_gob_unpack(NULL, W, blobs, offs, back, cb + 8, PGPU_MEM_HOST, lv, me));     /* host buffers: no context needed */
    for (i = 0; i < W; ++i) CHECK(memcmp(back + i * (cb + 8) + 8, encc + i * cb, cb) == 0 && lv[i] == 0 && me[i] == 0);
    CHECK(pgpu_gob_unpack(ctx, W, blobs, offs, back, 8, PGPU_MEM_HOST, NULL, NULL) == PGPU_ERR_INVALID);   /* C wider than the stride */
    CHECK(pgpu_gob_pack(ctx, W, encc, cb, PGPU_MEM_HOST, 0, 0, blobs, 16, offs) == PGPU_ERR_INVALID);       /* buffer too small */
    { size_t zero[2] = {0, 0}; CHECK(pgpu_gob_unpack(ctx, 1, blobs, zero, back, cb, PGPU_MEM_HOST, NULL, NULL) == PGPU_ERR_INVALID); }  /* "no data provided" */
    free(blobs); free(back); free(offs); free(encc); }

  /* ---- threshold decryption (thresholdkey.go:149-201): committed partials and plaintexts */
  { const entry_t* tc = get("t_c"); size_t nt = (size_t)tc->n, tl = bytes_of(get("t_n")->hex[0]), tcb = 2 * tl;
    uint8_t* cc = pack(tc, tcb); uint8_t* parts[3]; const uint8_t* cparts[3]; int ids[3] = {1, 3, 5}; int s;
    uint8_t shb[3][1024]; const uint8_t* shp[3]; size_t shl[3]; int32_t idx[MAXV]; uint8_t* un; uint8_t* uo;
    put(get("t_n")->hex[0], tnb, tl);
    { uint8_t tg[512]; memcpy(tg, tnb, tl); tg[tl - 1] += 1;                                /* G = N + 1 (n is odd) */
      OK(pgpu_pubkey_create(ctx, tnb, tl, tg, tl, NULL, 0, NULL, 0, &tpk)); }
    for (s = 0; s < 3; ++s) {
      const char* sh = get("t_shares")->hex[ids[s] - 1];
      char key[16];
      shl[s] = bytes_of(sh); put(sh, shb[s], shl[s]); shp[s] = shb[s];
      parts[s] = (uint8_t*)calloc(nt, tcb); cparts[s] = parts[s];
      OK(pgpu_partial_decrypt(tpk, 5, shb[s], shl[s], nt, cc, tcb, parts[s], tcb, PGPU_MEM_HOST));
      sprintf(key, "t_part%d", ids[s]);
      CHECK(same(parts[s], get(key), tcb));
    }
    OK(pgpu_combine_partial_decryptions(tpk, 5, 3, 3, ids, nt, cparts, tcb, out, tl, PGPU_MEM_HOST, st));
    CHECK(same(out, get("t_m"), tl) && st[0] == PGPU_LANE_OK);
    CHECK(pgpu_combine_partial_decryptions(tpk, 5, 3, 2, ids, nt, cparts, tcb, out, tl, PGPU_MEM_HOST, NULL) == PGPU_ERR_THRESHOLD);
    /* the units of the three servers in one launch */
    un = (uint8_t*)calloc(3 * nt, tcb); uo = (uint8_t*)calloc(3 * nt, tcb);
    for (s = 0; s < 3; ++s) { size_t i; for (i = 0; i < nt; ++i) { memcpy(un + (s * nt + i) * tcb, cc + i * tcb, tcb); idx[s * nt + i] = s; } }
    OK(pgpu_partial_decrypt_indexed(tpk, 5, 3, shp, shl, 3 * nt, un, tcb, idx, uo, tcb, PGPU_MEM_HOST));
    for (s = 0; s < 3; ++s) CHECK(memcmp(uo + (size_t)s * nt * tcb, parts[s], nt * tcb) == 0);
    /* the unit range of one rank of the sharded flow straight from the ciphertext batch (pgpu_partial_decrypt_units; what
     * paillier_amd/dist.py shard_slice hands rank r of w): rank 0 of 2 -- one server whole and part of the next, the ciphertexts
     * wanted under both shares walk ONE chain of squarings --, rank 7 of 8 -- the tail of the last server --, and the whole job */
    { static const int rw[3][2] = {{0, 2}, {7, 8}, {0, 1}}; int t;
      for (t = 0; t < 3; ++t) {
        size_t total = 3 * nt, w = (size_t)rw[t][1], r = (size_t)rw[t][0], base = total / w, rem = total % w;
        size_t ub = r * base + (r < rem ? r : rem), ue = ub + base + (r < rem ? 1 : 0);
        uint8_t* ro = (uint8_t*)calloc(ue - ub + 1, tcb);
        CHECK(ub < ue);
        OK(pgpu_partial_decrypt_units(tpk, 5, 3, shp, shl, nt, cc, tcb, ub, ue, ro, tcb, PGPU_MEM_HOST));
        CHECK(memcmp(ro, uo + ub * tcb, (ue - ub) * tcb) == 0);
        free(ro);
      }
      CHECK(pgpu_partial_decrypt_units(tpk, 5, 3, shp, shl, nt, cc, tcb, 2, 2, uo, tcb, PGPU_MEM_HOST) == PGPU_ERR_INVALID);           /* empty range */
      CHECK(pgpu_partial_decrypt_units(tpk, 5, 3, shp, shl, nt, cc, tcb, 0, 3 * nt + 1, uo, tcb, PGPU_MEM_HOST) == PGPU_ERR_INVALID);  /* past the job */
    }
    /* one ciphertext batch under the three shares held by one process (pgpu_partial_decrypt_multi) */
    { uint8_t* mo[3]; uint8_t* const* mop = mo;
      for (s = 0; s < 3; ++s) mo[s] = (uint8_t*)calloc(nt, tcb);
      OK(pgpu_partial_decrypt_multi(tpk, 5, 3, shp, shl, nt, cc, tcb, mop, tcb, PGPU_MEM_HOST));
      for (s = 0; s < 3; ++s) { CHECK(memcmp(mo[s], parts[s], nt * tcb) == 0); free(mo[s]); } }
    /* share-decryption proof: prove with the committed r, compare (dec, E, Z), verify */
    { const entry_t* zc = get("z_c"); size_t nz = (size_t)zc->n, zs = tcb + 48; uint8_t* zcb = pack(zc, tcb); uint8_t* zr = pack(get("z_r"), tcb);
      uint8_t* dec = (uint8_t*)calloc(nz, tcb); uint8_t* ee = (uint8_t*)calloc(nz, 32); uint8_t* zz = (uint8_t*)calloc(nz, zs);
      uint8_t vb[1024], vib[1024], s2[1024]; size_t vl = bytes_of(get("t_v")->hex[0]), vil = bytes_of(get("t_vks")->hex[1]);
      size_t s2l = bytes_of(get("t_shares")->hex[1]);
      put(get("t_v")->hex[0], vb, vl); put(get("t_vks")->hex[1], vib, vil); put(get("t_shares")->hex[1], s2, s2l);
      OK(pgpu_share_zkp_prove(tpk, 5, s2, s2l, vb, vl, nz, zcb, tcb, zr, tcb, dec, tcb, ee, zz, zs, PGPU_MEM_HOST));
      CHECK(same(dec, get("z_dec"), tcb) && same(ee, get("z_e"), 32) && same(zz, get("z_z"), zs));
      OK(pgpu_share_zkp_verify(tpk, vb, vl, vib, vil, nz, zcb, tcb, dec, tcb, ee, zz, zs, st, PGPU_MEM_HOST));
      { size_t i; for (i = 0; i < nz; ++i) CHECK(st[i] == 1); }
      ee[31] ^= 1;
      OK(pgpu_share_zkp_verify(tpk, vb, vl, vib, vil, nz, zcb, tcb, dec, tcb, ee, zz, zs, st, PGPU_MEM_HOST));
      CHECK(st[0] == 0 && st[1] == 1);
      /* the transcript hash on its own: E = SHA-256(a || b || c^4 || ci^2) of the committed integers */
      { const uint8_t* hp[4]; size_t hs[4]; uint8_t dg[32 * MAXV]; uint8_t* h0 = pack(get("z_a"), tcb); uint8_t* h1 = pack(get("z_b"), tcb);
        uint8_t* h2 = pack(get("z_c4"), 4 * tcb); uint8_t* h3 = pack(get("z_ci2"), 2 * tcb);
        hp[0] = h0; hp[1] = h1; hp[2] = h2; hp[3] = h3; hs[0] = tcb; hs[1] = tcb; hs[2] = 4 * tcb; hs[3] = 2 * tcb;
        OK(pgpu_random_oracle_digest(ctx, 4, hp, hs, nz, dg, PGPU_MEM_HOST));
        CHECK(same(dg, get("z_e"), 32));
        free(h0); free(h1); free(h2); free(h3); }
      free(zcb); free(zr); free(dec); free(ee); free(zz); }
    free(cc); free(un); free(uo); for (s = 0; s < 3; ++s) free(parts[s]); }

  /* ---- DDLEQ (ddleq.go:55-153): prove with the committed (x, y), compare, verify */
  { const entry_t* d1 = get("d_ct1"); size_t nd = (size_t)d1->n, i;
    uint8_t *c1 = pack(d1, cb3), *c2 = pack(get("d_ct2"), cb3), *da = pack(get("d_a"), pb), *db = pack(get("d_b"), pb),
            *dx = pack(get("d_x"), pb), *dy = pack(get("d_y"), pb);
    uint8_t *al = (uint8_t*)calloc(nd, cb3), *ee = (uint8_t*)calloc(nd, pb2), *ff = (uint8_t*)calloc(nd, cb3);
    OK(pgpu_nested_randomize_with_ab(pk, nd, c1, cb3, da, db, pb, al, cb3, PGPU_MEM_HOST));     /* operations.go:96-118 */
    CHECK(memcmp(al, c2, nd * cb3) == 0);
    OK(pgpu_ddleq_prove(sk, nd, c1, c2, cb3, da, db, dx, dy, pb, al, ee, pb2, ff, PGPU_MEM_HOST));
    CHECK(same(al, get("d_alpha"), cb3) && same(ee, get("d_e"), pb2) && same(ff, get("d_f"), cb3));
    OK(pgpu_ddleq_verify(pk, nd, c1, c2, cb3, dx, dy, pb, al, cb3, ee, pb2, ff, cb3, st, PGPU_MEM_HOST));
    for (i = 0; i < nd; ++i) CHECK(st[i] == 1);
    ff[5] ^= 1;
    OK(pgpu_ddleq_verify(pk, nd, c1, c2, cb3, dx, dy, pb, al, cb3, ee, pb2, ff, cb3, st, PGPU_MEM_HOST));
    CHECK(st[0] == 0 && st[1] == 1);
    CHECK(pgpu_ddleq_prove(sk, 2, c1, c2 + cb3, cb3, da, db, dx, dy, pb, al, ee, pb2, ff, PGPU_MEM_HOST) == PGPU_ERR_INVALID);   /* false statement */
    /* ProveDDLEQ with secpar instances per statement (ddleq.go:27-40, pgpu_ddleq_prove_secpar): the fixture's rows i and i + 4
     * are two instances of the same statement (make_golden.py: statement = i mod 4), so ONE statement (row 0) with secpar 2
     * and the draws of rows 0 and 4 must give the committed rows 0 and 4; nd statements with secpar 1 the committed rows */
    if (nd >= 8) {
      uint8_t *x2 = (uint8_t*)calloc(2, pb), *y2 = (uint8_t*)calloc(2, pb);
      memcpy(x2, dx, pb); memcpy(x2 + pb, dx + 4 * pb, pb); memcpy(y2, dy, pb); memcpy(y2 + pb, dy + 4 * pb, pb);
      OK(pgpu_ddleq_prove_secpar(sk, 1, 2, c1, c2, cb3, da, db, x2, y2, pb, al, ee, pb2, ff, PGPU_MEM_HOST));
      { const entry_t *wa = get("d_alpha"), *we = get("d_e"), *wf = get("d_f"); uint8_t* w = (uint8_t*)calloc(1, cb3); int k;
        for (k = 0; k < 2; ++k) {
          put(wa->hex[4 * k], w, cb3); CHECK(memcmp(al + (size_t)k * cb3, w, cb3) == 0);
          put(we->hex[4 * k], w, pb2); CHECK(memcmp(ee + (size_t)k * pb2, w, pb2) == 0);
          put(wf->hex[4 * k], w, cb3); CHECK(memcmp(ff + (size_t)k * cb3, w, cb3) == 0);
        }
        free(w); }
      free(x2); free(y2);
    }
    OK(pgpu_ddleq_prove_secpar(sk, nd, 1, c1, c2, cb3, da, db, dx, dy, pb, al, ee, pb2, ff, PGPU_MEM_HOST));
    CHECK(same(al, get("d_alpha"), cb3) && same(ee, get("d_e"), pb2) && same(ff, get("d_f"), cb3));
    CHECK(pgpu_ddleq_prove_secpar(sk, 1, 0, c1, c2, cb3, da, db, dx, dy, pb, al, ee, pb2, ff, PGPU_MEM_HOST) == PGPU_ERR_INVALID);
    free(c1); free(c2); free(da); free(db); free(dx); free(dy); free(al); free(ee); free(ff); }

  /* ---- two contexts, two threads: the sharded flows of go/sharded.go against the same fixtures ---------------------------------- */
  { enum { WORLD = 2 };
    const entry_t *dce = get("dec_c"), *tce = get("t_c");
    size_t nd = (size_t)dce->n, nt = (size_t)tce->n, tl = bytes_of(get("t_n")->hex[0]), tcb = 2 * tl; int i, s; int ids3[3] = {1, 3, 5};
    uint8_t* dcb = pack(dce, cb); uint8_t* tcbuf = pack(tce, tcb);
    uint8_t* dm = (uint8_t*)calloc(nd, pb); uint8_t* parts = (uint8_t*)calloc(3 * nt, tcb); uint8_t* tm = (uint8_t*)calloc(nt, tl);
    uint8_t tg[512]; uint8_t shb[3][1024]; const uint8_t* shp[3]; size_t shl[3];
    pthread_barrier_t bar; pthread_t th[WORLD]; worker_t ws[WORLD];
    memcpy(tg, tnb, tl); tg[tl - 1] += 1;
    for (s = 0; s < 3; ++s) { const char* sh = get("t_shares")->hex[ids3[s] - 1]; shl[s] = bytes_of(sh); put(sh, shb[s], shl[s]); shp[s] = shb[s]; }
    CHECK(pthread_barrier_init(&bar, NULL, WORLD) == 0);
    for (i = 0; i < WORLD; ++i) {
      worker_t* w = &ws[i];
      memset(w, 0, sizeof *w);
      w->rank = i; w->world = WORLD; w->device = 0; w->bar = &bar;
      w->n = nb; w->g = gb; w->lambda = lb; w->tn = tnb; w->tg = tg; w->nl = nl; w->ll = ll; w->tl = tl;
      w->shares = shp; w->share_lens = shl;
      w->dc = dcb; w->nd = nd; w->dm = dm; w->tc = tcbuf; w->nt = nt; w->parts = parts; w->tm = tm;
      CHECK(pthread_create(&th[i], NULL, sharded_worker, w) == 0);
    }
    for (i = 0; i < WORLD; ++i) pthread_join(th[i], NULL);
    pthread_barrier_destroy(&bar);
    for (i = 0; i < WORLD; ++i) if (ws[i].rc != PGPU_OK) { fprintf(stderr, "sharded worker %d failed (%d): %s\n", i, ws[i].rc, ws[i].err); return 1; }
    CHECK(same(dm, get("dec_m"), pb));                              /* the slices, concatenated, are the batch's plaintexts */
    CHECK(same(tm, get("t_m"), tl));
    { char key[16]; for (s = 0; s < 3; ++s) { uint8_t* wv; sprintf(key, "t_part%d", ids3[s]); wv = pack(get(key), tcb);
        CHECK(memcmp(parts + (size_t)s * nt * tcb, wv, nt * tcb) == 0); free(wv); } }   /* what went through the exchange */
    free(dcb); free(tcbuf); free(dm); free(parts); free(tm);
  }

  /* ---- error conventions */
  CHECK(pgpu_encrypt_with_r(pk, 7, B, m, pb, r, pb, out, cb, PGPU_MEM_HOST) == PGPU_ERR_INVALID);
  CHECK(pgpu_encrypt_with_r(NULL, 0, B, m, pb, r, pb, out, cb, PGPU_MEM_HOST) == PGPU_ERR_INVALID);
  { uint8_t even[4] = {0, 0, 1, 0}; pgpu_modulus* bad = NULL; CHECK(pgpu_modulus_create(ctx, even, 4, &bad) == PGPU_ERR_INVALID); }

  pgpu_modulus_destroy(mod);
  pgpu_seckey_destroy(sk);
  pgpu_pubkey_destroy(pk);
  pgpu_pubkey_destroy(tpk);
  pgpu_ctx_destroy(ctx);
  free(m); free(r); free(out); free(a); free(b); free(k0);
  printf("c abi ok\n");
  return 0;
}
