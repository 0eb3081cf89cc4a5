/* paillier_hip.h -- C ABI of the MI355X batched Paillier engine (libpaillier_hip.so).
 *
 * This is the drop-in boundary for the hot path of sachaservan/paillier: the reference has no FFI
 * seam of its own; every hot-path call bottoms out in github.com/ncw/gmp
 * (`Exp`, `Mul`+`Mod`, `ModInverse`).  Each entry point below replaces the per-ciphertext loop a
 * Go caller would write around ONE exported reference method, for a whole batch, and cites that
 * method.  INTEGRATION.md shows the cgo binding.
 *
 * Operand format (what Go can produce from gmp.Int.Bytes()): unsigned big-endian integers,
 * ELEMENT-MAJOR with a fixed byte stride per element, left-padded with zero bytes.
 * `mem` says where the buffers live: PGPU_MEM_HOST (copied by the library) or PGPU_MEM_DEVICE
 * (HBM pointers on the context's device; nothing is copied).
 *
 * All functions return PGPU_OK (0) or a negative error; pgpu_last_error() gives the text.
 * Every batch function launches HIP kernels on the context's stream and blocks until the
 * results are in the output buffers.  There is no CPU fallback: without a gfx950 device,
 * pgpu_ctx_create fails with PGPU_ERR_NO_DEVICE.
 */
#ifndef PAILLIER_HIP_H
#define PAILLIER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: only what this header (and paillier_hip_debug.h) declares is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define PGPU_OK 0
#define PGPU_ERR_INVALID (-1)       /* bad argument (size, level, null pointer, even modulus ...) */
#define PGPU_ERR_NO_DEVICE (-2)     /* no HIP device / not gfx950 */
#define PGPU_ERR_HIP (-3)           /* HIP runtime error (see pgpu_last_error) */
#define PGPU_ERR_UNSUPPORTED (-4)   /* modulus width or level not built */
#define PGPU_ERR_NOT_INVERTIBLE (-5)/* ModInverse of a non-unit (undefined in the reference) */
#define PGPU_ERR_THRESHOLD (-6)     /* "Threshold not meet" / duplicate share ids (thresholdkey.go:77-89) */

#define PGPU_MEM_HOST 0
#define PGPU_MEM_DEVICE 1

/* paillier.go:17-23 */
#define PGPU_LEVEL_ONE 0
#define PGPU_LEVEL_TWO 1

/* pgpu_decrypt flags */
#define PGPU_DECRYPT_DEFAULT 0      /* CRT over p^(s+1), q^(s+1) when (n, lambda) factor n, else generic */
#define PGPU_DECRYPT_NO_CRT 1       /* the reference's formula verbatim: c^lambda mod n^(s+1), L, * lambda^-1 */

/* per-lane status bits (int32 array, optional) */
#define PGPU_LANE_OK 0
#define PGPU_LANE_NONUNIT 1         /* gcd(c, n) != 1: lane was recomputed on the generic path */
#define PGPU_LANE_NOT_INVERTIBLE 2  /* a ModInverse operand of this lane is not a unit: its result is 0 (mpz_invert leaves
                                     * the reference's result undefined there); the other lanes are unaffected */

typedef struct pgpu_ctx pgpu_ctx;
typedef struct pgpu_pubkey pgpu_pubkey;
typedef struct pgpu_seckey pgpu_seckey;
typedef struct pgpu_modulus pgpu_modulus;

const char* pgpu_last_error(void);
const char* pgpu_version(void);

/* device = HIP device ordinal; stream = hipStream_t to launch on, NULL for the default stream, or PGPU_STREAM_NEW for a
 * non-blocking stream that the context creates and owns.  A context is thread-compatible (one call at a time); batches too
 * small to fill the GPU (a few thousand ciphertexts: the run time is the latency of one ladder) are overlapped by issuing
 * them from several host threads on several contexts with streams of their own -- the kernels then share the chip. */
#define PGPU_STREAM_NEW ((void*)(intptr_t)-1)
int pgpu_ctx_create(int device, void* stream, pgpu_ctx** out);
void pgpu_ctx_destroy(pgpu_ctx* ctx);
/* Timing of the dominant (modexp VM) kernel of the last batch call, measured with HIP events on
 * the context's stream: milliseconds, number of launches, and 28-bit-limb multiply-adds executed. */
int pgpu_ctx_last_profile(pgpu_ctx* ctx, double* vm_ms, int* vm_launches, double* vm_mads);
/* Runtime switches.  "asm" (default 1): run the hand-scheduled gfx950 assembly VM kernels; 0 selects the
 * hipcc-generated kernels of identical semantics (used by the parity tests to cross-check the two).
 * "pair" (default 1): run the Decrypt ladders modulo p^2 / q^2 on the pair kernel (two base-p digits per residue); 0
 * keeps them on the ordinary 2H-limb kernel.
 * "triple" (default 1): ladders modulo n^3 on the three-digit kernel.  "shared_chain" (default 1): several shares on the
 * same ciphertexts share one chain of squarings (pgpu_partial_decrypt_multi), runs of units under one share become
 * shared-exponent ladders (pgpu_partial_decrypt_indexed).  "lift" (default 1): level-two Encrypt computes r^(n^2) mod n^3
 * as (r^n mod n^2)^n mod n^3, and the key holder's x^n mod n^2 (pgpu_encrypt_with_r_sk, the DDLEQ prover's a^n | x^n) goes through the
 * primes: t = x^q mod p, then the Teichmueller lift t^p mod p^2 (and likewise modulo q^2), a third fewer multiplies than the exponent
 * n mod p (p - 1) in one ladder.  "side" (default 1): a call may issue work that depends on no ladder in flight to a second
 * stream of the context (the DDLEQ prover's per-statement chains run beside its big launches); 0 keeps one stream.
 * "lanes8" (default 1): batches too small to fill the chip at four lanes per number run ladders modulo n^2 on the eight-lane pair
 * kernel (and ladders modulo n^3 on the three-digit kernel with two lanes per digit); "muls" (default 1): bucket products of a shared
 * chain of squarings leave the current power in registers (VM_MULS); "nm4" (default 1): the window tables of per-number exponents
 * on the pair kernels are stored number-major (VM_STORET / VM_MULVT / VM_MULVT5: contiguous gathers), 0: limb-major; "early"
 * (default 1): the DDLEQ prover prepares its response for every statement and instance beside the Alpha ladders and gathers it
 * for the instances whose challenge bit is 1 (0: prepared after the hash, for those instances only); "handover" (default 1): a
 * power modulo n^2 that the next ladder modulo n^3 needs modulo n^2 only goes from the pair kernel to the digit kernel as
 * (a0, a1, 0) without leaving Montgomery / pair form (0: exit and re-entry); "struct" (default 1): the DDLEQ prover computes
 * ct1^e y^(n^2) mod n^3 (sanity value, Alpha) -- and from four instances per statement its response -- through the structure of the
 * unit group: the level-two plaintext of ct1 once per statement, ladders modulo the primes, one Teichmueller lift per number (0:
 * ladders on ct1 itself; non-unit inputs always take those); "late" (default 1): with fewer than four instances per statement (and a batch that
 * fills the chip) the prover's response goes through the structure as well, AFTER the hash: the level-two plaintext of b for the statements
 * that have an instance with challenge bit 1 only, one ladder modulo the primes, one lift (0: one ladder of 3 071 squarings modulo p^3, q^3
 * on s and b themselves); "lanes16" (default 1): ladders modulo n^2 of up to 2 048 numbers -- and the verifier's two ladders modulo n^3 -- run on sixteen lanes per number
 * (0: eight); "prime_lanes" (default 1): the key holder's ladders modulo the primes of a 2048-bit key take four lanes of 10 limbs
 * per number while the batch (up to 8 192 numbers) leaves every wave a SIMD of its own (0: one lane per number throughout); "base_early" (default 1): the links of the prover's side chains -- the per-statement structure chain (ct1 modulo
 * the primes, its plaintext) and the preparation of the response -- run beside the main stream's ladders instead of waiting for an empty
 * compute unit (0: between those ladders; measured 1 % slower and less steady call to call); "exclusive" (default 1): placement by LDS size -- the small concurrent launches
 * of a prover call (up to 128 workgroups) ask for the whole LDS of a compute unit per workgroup, so that the side lanes' workgroups
 * land on idle CUs instead of the ones the main launch runs on, and a main-stream ladder of at most one workgroup per CU asks for more
 * than half of it, so that it spreads over all CUs (0: the dispatcher's placement); "exclusive_short" (default 1): the links between ladders -- programs of a few
 * products -- ask for a whole CU as well (0: only ladders do and the links run beside the main stream's launches; measured equal); "spread" (default 1): the second of those two rules
 * alone -- which inside a prover call also holds for a side lane's ladder of 129 ... 256 workgroups -- (0: such ladders keep the dispatcher's placement; both rules stop at the compute units the context's stream may use, so a
 * context with a "cu_partition" narrower than its launch keeps the dispatcher's placement by itself); "w74" (default 1): moduli of two
 * 74-limb slices on the wave-sliced assembly kernel (0: four lanes of 37 limbs); "exp_order" (default 1): the key holder's exponents
 * modulo p^3, q^3 are reduced modulo the group orders on the device (0: used as given); "background" (default 0): the prover's side-lane
 * ladders run at wave priority 0 (measured: no gain).
 * These names are the ONLY switches of the plan: the library reads no environment variable that changes what is launched (a Go host
 * inherits its process environment from wherever it runs).  PGPU_PROFILE_DUMP / PGPU_HOST_TRACE (include/paillier_hip_debug.h) print
 * timings to stderr and change nothing else.
 * All of these change the work done, never a result (the tests switch them off to compare).
 * "lanes_wanted" (default 0 = fill the chip): the lane count below which a batch is re-sliced over more lanes per
 * number; 1 keeps every modulus on its natural kernel shape whatever the batch size (tests).
 * "fair" (default 1, process-wide): programs carry wave priorities that fall with progress, so that the waves sharing a SIMD
 * finish together (0: A/B measurements).
 * "cu_partition": (parts << 16) | part confines an own-stream context to one slice of the compute units. */
int pgpu_ctx_set_flag(pgpu_ctx* ctx, const char* name, int value);
/* number of VM launches of the last batch call that ran the assembly kernel */
int pgpu_ctx_last_vm_asm(pgpu_ctx* ctx);
/* ... and how many VM launches it made in all (equal to the above when nothing fell back to the compiler-generated kernel) */
int pgpu_ctx_last_vm_launches(pgpu_ctx* ctx);
/* name of the profiled VM launch of the last batch call that executed the most multiply-adds (its dominant kernel), as it
 * appears in a rocprof kernel trace, e.g. "vm_asm_37_16"; "" if the call profiled none.  Valid until the next call. */
const char* pgpu_ctx_last_kernel(pgpu_ctx* ctx);

/* ---- keys ---------------------------------------------------------------------------------- */

/* PublicKey{N,G,H,K} (paillier.go:46-57).  H and K may be NULL/0 when alternative encryption is
 * not used.  n^2, n^3 and all Montgomery constants are precomputed here, immutably (the reference
 * caches them lazily and racily: paillier.go:72-90). */
int pgpu_pubkey_create(pgpu_ctx* ctx, const uint8_t* n_be, size_t n_len, const uint8_t* g_be, size_t g_len,
                       const uint8_t* h_be, size_t h_len, const uint8_t* k_be, size_t k_len, pgpu_pubkey** out);
void pgpu_pubkey_destroy(pgpu_pubkey* pk);
/* byte length of n^s (plaintexts) and n^(s+1) (ciphertexts) for the level: the natural strides */
size_t pgpu_pubkey_plain_bytes(const pgpu_pubkey* pk, int level);
size_t pgpu_pubkey_cipher_bytes(const pgpu_pubkey* pk, int level);

/* SecretKey{PublicKey; Lambda} (paillier.go:60-63).  p and q are recovered from (n, lambda = phi(n))
 * to enable CRT; if lambda is not phi(n) the generic path is used. */
int pgpu_seckey_create(pgpu_ctx* ctx, const pgpu_pubkey* pk, const uint8_t* lambda_be, size_t lambda_len,
                       pgpu_seckey** out);
void pgpu_seckey_destroy(pgpu_seckey* sk);
int pgpu_seckey_has_crt(const pgpu_seckey* sk);

/* ---- Paillier batch operations --------------------------------------------------------------- */

/* PublicKey.EncryptWithRAtLevel (paillier.go:206-218), for i in [0,batch):
 *   c[i] = G^m[i] * r[i]^(n^s) mod n^(s+1)                                                        */
int pgpu_encrypt_with_r(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* m, size_t m_stride,
                        const uint8_t* r, size_t r_stride, uint8_t* c, size_t c_stride, int mem);
/* The same for the holder of the secret key (in the reference SecretKey embeds PublicKey, paillier.go:59-62: sk.EncryptWithR is
 * the same method).  Level one: r^n mod n^2 through the primes -- t = (r mod p)^(q mod (p - 1)), then the Teichmueller lift t^p mod p^2
 * (and likewise modulo q^2): a fifth of the public path's multiplies.  Level two: r^(n^2) mod n^3 as the lift modulo p^3, q^3 of
 * (r mod p)^(q^2 mod (p - 1)): a fifth of the multiplies again; a batch with an r that shares a factor with n takes the public path.
 * Identical ciphertexts for every input; keys without the factorisation take the public path. */
int pgpu_encrypt_with_r_sk(const pgpu_seckey* sk, int level, size_t batch, const uint8_t* m, size_t m_stride,
                           const uint8_t* r, size_t r_stride, uint8_t* c, size_t c_stride, int mem);

/* PublicKey.EncryptAtLevel (paillier.go:258-269) for a batch: for every message a fresh r uniform in Z_n^* is drawn from the
 * operating system's CSPRNG (getrandom(2)) the way utils.go:26-49 does it -- crypto/rand.Int's rejection sampling below n,
 * redrawn while r = 0 or gcd(r, n) != 1 (the gcd test runs on the device for the whole batch) -- then c[i] as
 * pgpu_encrypt_with_r.  r_out (optional, r_stride bytes per element, in `mem`) receives the randomness used. */
int pgpu_encrypt(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* m, size_t m_stride, uint8_t* c, size_t c_stride,
                 uint8_t* r_out, size_t r_stride, int mem);

/* GetRandomNumberInMultiplicativeGroup (utils.go:36-49) for a batch: r_out[i] uniform in Z_n^* (see pgpu_encrypt). */
int pgpu_random_units(const pgpu_pubkey* pk, size_t batch, uint8_t* r_out, size_t r_stride, int mem);

/* PublicKey.AltEncryptWithRAtLevel (paillier.go:221-238): c[i] = G^m[i] * h_s^(r[i] mod K) mod n^(s+1), with
 * h_1 = (N-H)^N mod N^2, h_2 = (N^2-H)^(N^2) mod N^3 (paillier.go:416-434).  h_s is shared by the batch, so the engine
 * uses a fixed-base comb table (no squarings).  The reference overwrites the caller's r with r mod K; pass r_reduced
 * (same stride as r, may be NULL) to receive it.  Requires H and K (a power of two) in the public key. */
int pgpu_alt_encrypt_with_r(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* m, size_t m_stride,
                            const uint8_t* r, size_t r_stride, uint8_t* c, size_t c_stride, uint8_t* r_reduced, int mem);

/* SecretKey.Decrypt (paillier.go:292-303): m[i] = L_s(c[i]^lambda mod n^(s+1)) * lambda^-1 mod n^s.
 * status (optional, host int32[batch]) receives PGPU_LANE_* bits. */
int pgpu_decrypt(const pgpu_seckey* sk, int level, size_t batch, const uint8_t* c, size_t c_stride, uint8_t* m,
                 size_t m_stride, int mem, int flags, int32_t* status);

/* PublicKey.Add on two ciphertext vectors (operations.go:11-29 with two operands):
 *   out[i] = a[i] * b[i] mod n^(s+1)                                                              */
int pgpu_add(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* a, size_t a_stride, const uint8_t* b,
             size_t b_stride, uint8_t* out, size_t out_stride, int mem);

/* PublicKey.Sub on two ciphertext vectors (operations.go:32-55 with two operands):
 *   out[i] = a[i] * b[i]^-1 mod n^(s+1).
 * A b[i] that is not a unit has no inverse (mpz_invert leaves the reference's result undefined and the reference does not
 * check): that lane's result is 0 and status[i] = PGPU_LANE_NOT_INVERTIBLE, every other lane is computed normally.
 * status: optional host int32[batch]; when it is NULL such a lane makes the call return PGPU_ERR_NOT_INVERTIBLE (after
 * the outputs have been written). */
int pgpu_sub(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* a, size_t a_stride, const uint8_t* b,
             size_t b_stride, uint8_t* out, size_t out_stride, int mem, int32_t* status);

/* The variadic forms PublicKey.Add(cts...) / Sub(cts...) (operations.go:11,32), element-wise over a batch: ops[k] is the
 * buffer of the k-th argument (all with the same stride), n_ops >= 1.
 *   add: out[i] = prod_k ops[k][i] mod n^(s+1)   (the accumulator starts at 1: a single operand comes back reduced)
 *   sub: out[i] = ops[0][i] * prod_{k>=1} ops[k][i]^-1 mod n^(s+1); with ONE operand the reference returns cts[0].C
 *        unreduced (operations.go:34): the bytes are copied through unchanged. */
int pgpu_add_many(const pgpu_pubkey* pk, int level, int n_ops, size_t batch, const uint8_t* const* ops, size_t stride,
                  uint8_t* out, size_t out_stride, int mem);
int pgpu_sub_many(const pgpu_pubkey* pk, int level, int n_ops, size_t batch, const uint8_t* const* ops, size_t stride,
                  uint8_t* out, size_t out_stride, int mem, int32_t* status);

/* PublicKey.ConstMult (operations.go:58-64): out[i] = c[i]^k mod n^(s+1).
 * k_stride == 0: one shared k of k_len bytes; otherwise k[i] at k + i*k_stride, k_len bytes each. */
int pgpu_const_mult(const pgpu_pubkey* pk, int level, size_t batch, const uint8_t* c, size_t c_stride,
                    const uint8_t* k, size_t k_len, size_t k_stride, uint8_t* out, size_t out_stride, int mem);

/* ---- threshold decryption (thresholdkey.go) ---------------------------------------------------------- */

/* ThresholdSecretKey.PartialDecrypt (thresholdkey.go:192-201): out[i] = c[i]^(2 * l! * share) mod n^2,
 * l = total_servers.  The share stays on the caller's side of the ABI (big-endian). */
int pgpu_partial_decrypt(const pgpu_pubkey* pk, int total_servers, const uint8_t* share_be, size_t share_len, size_t batch,
                         const uint8_t* c, size_t c_stride, uint8_t* out, size_t out_stride, int mem);

/* ThresholdSecretKey.PartialDecrypt of ONE ciphertext batch by SEVERAL servers (a process that holds more than one share:
 * the shape of the reference's own BenchmarkThresholdDecrypt, thresholdkey_test.go:396-427): outs[k][i] = c[i]^(2 * l! * shares[k]).
 * The entry conversion of the ciphertexts is done once.  From 8 192 ciphertexts up all the shares' ladders share ONE chain of
 * squarings (right-to-left windows into buckets; three servers cost about 1.5 ladders); below that the ladders of two servers
 * share a launch.  Same integers either way. */
int pgpu_partial_decrypt_multi(const pgpu_pubkey* pk, int total_servers, int n_shares, const uint8_t* const* shares_be,
                               const size_t* share_lens, size_t batch, const uint8_t* c, size_t c_stride, uint8_t* const* outs,
                               size_t out_stride, int mem);

/* The shard of one rank of the sharded threshold flow: the (server, ciphertext) units u = s * batch + i of the contiguous,
 * server-major range [unit_begin, unit_end) over ONE batch of ciphertexts (thresholdkey.go:192-201 for each unit; shares[s] is
 * server s's share, only those the range touches are read).  Ciphertexts the range wants under several shares walk ONE chain
 * of squarings for all of them; the rest run their ladders side by side in the same launch.  out: unit_end - unit_begin rows,
 * in unit order.  Same integers as pgpu_partial_decrypt per server.  (The whole unit range of a ciphertext SLICE is the shard of a
 * rank that holds every share: from 8 192 ciphertexts one chain for all shares, below that the shares split into up to three groups
 * with a chain each while every group keeps a SIMD per wave on the eight-lane kernel.) */
int pgpu_partial_decrypt_units(const pgpu_pubkey* pk, int total_servers, int n_shares, const uint8_t* const* shares_be,
                               const size_t* share_lens, size_t batch, const uint8_t* c, size_t c_stride, size_t unit_begin,
                               size_t unit_end, uint8_t* out, size_t out_stride, int mem);

/* ThresholdSecretKey.PartialDecrypt for a batch of (share, ciphertext) UNITS: out[i] = c[i]^(2 * l! * shares[share_index[i]])
 * mod n^2.  One launch serves the units of several decryption servers (or any mix of them): the exponents become per-unit
 * operands of the ladder.  This is what a shard of a threshold batch looks like -- e.g. 16 384 ciphertexts x 3 servers over
 * 8 GPUs = 6 144 units per GPU: per server they could not fill the chip.  share_index: host int32[batch]. */
int pgpu_partial_decrypt_indexed(const pgpu_pubkey* pk, int total_servers, int n_shares, const uint8_t* const* shares_be,
                                 const size_t* share_lens, size_t batch, const uint8_t* c, size_t c_stride,
                                 const int32_t* share_index, uint8_t* out, size_t out_stride, int mem);

/* ThresholdPublicKey.CombinePartialDecryptions (thresholdkey.go:149-161) for a batch of ciphertexts:
 *   m[i] = L( prod_k partials[k][i]^(2 lambda_k) mod n^2 ) * (4 (l!)^2)^-1 mod n
 * ids[k] = server id (1-based) of partials[k]; all partial buffers share `stride`.  Lagrange coefficients follow
 * the reference's sequence of Euclidean divisions; negative exponents use a modular inverse as thresholdkey.go:132-138.
 * PGPU_ERR_THRESHOLD when fewer than `threshold` shares or duplicate ids (thresholdkey.go:77-89).
 * status (optional host int32[batch]): PGPU_LANE_NOT_INVERTIBLE where a share with a negative coefficient is not a unit
 * modulo n^2 (its m[i] is then computed with 0 for the missing inverse); NULL turns such a lane into the return code
 * PGPU_ERR_NOT_INVERTIBLE, as for pgpu_sub. */
int pgpu_combine_partial_decryptions(const pgpu_pubkey* pk, int total_servers, int threshold, int n_shares, const int* ids,
                                     size_t batch, const uint8_t* const* partials, size_t stride, uint8_t* m,
                                     size_t m_stride, int mem, int32_t* status);

/* ---- share-decryption proofs (thresholdkey.go:225-326) ------------------------------------------------- */

/* ThresholdSecretKey.PartialDecryptionWithZKP for a batch, with the random r (< n^2, thresholdkey.go:233) supplied:
 *   dec[i] = c[i]^(2 l! s) ; a = (c^4)^r, b = V^r mod n^2 ; E = SHA-256(a || b || c^4 || dec^2) (unreduced powers) ;
 *   Z = r + E l! s.   e_out: 32 bytes per proof; z_out: z_stride bytes per proof.  V^r uses a fixed-base comb table. */
int pgpu_share_zkp_prove(const pgpu_pubkey* pk, int total_servers, const uint8_t* share_be, size_t share_len,
                         const uint8_t* vkey_be, size_t vkey_len, size_t batch, const uint8_t* c, size_t c_stride,
                         const uint8_t* r, size_t r_stride, uint8_t* dec, size_t dec_stride, uint8_t* e_out, uint8_t* z_out,
                         size_t z_stride, int mem);

/* PartialDecryptionZKP.VerifyProof for a batch of proofs of ONE server (thresholdkey.go:278-311):
 *   ok[i] = ( E == SHA-256( (c^4)^Z (dec^2)^-E || V^Z v_i^-E || c^4 || dec^2 ) ),  v_i = VerificationKeys[ID-1].
 * A proof whose Decryption is not a unit modulo n^2 (e.g. 0, or a multiple of a prime factor) is rejected (ok[i] = 0)
 * without affecting the other proofs of the batch.  z_stride <= byte length of n^2 + 96. */
int pgpu_share_zkp_verify(const pgpu_pubkey* pk, const uint8_t* vkey_be, size_t vkey_len, const uint8_t* vi_be, size_t vi_len,
                          size_t batch, const uint8_t* c, size_t c_stride, const uint8_t* dec, size_t dec_stride,
                          const uint8_t* e, const uint8_t* z, size_t z_stride, int32_t* ok, int mem);

/* ---- proofs (ddleq.go, random_oracle.go) ----------------------------------------------------------- */

/* RandomOracleDigest-style transcripts on the device: digests[i] = SHA-256( Bytes(parts[0][i]) || ... ) where Bytes is
 * gmp.Int.Bytes() (minimal big-endian, zero -> empty; random_oracle.go:20-32, thresholdkey.go:319-326).  The caller
 * passes exactly the integers that are hashed (i.e. WITHOUT the skipped first argument of RandomOracleDigest).
 * parts[k] is a fixed-stride big-endian buffer with strides[k] bytes per element; digests = 32 bytes per element. */
int pgpu_random_oracle_digest(pgpu_ctx* ctx, int nparts, const uint8_t* const* parts, const size_t* strides, size_t batch,
                              uint8_t* digests, int mem);

/* PublicKey.NestedRandomize (operations.go:96-118) with the two random draws a, b in Z_n^* supplied:
 *   out[i] = ct[i]^(a[i]^n mod n^2) * b[i]^(n^2) mod n^3      (ct: level-two ciphertexts; one interleaved ladder per ciphertext) */
int pgpu_nested_randomize_with_ab(const pgpu_pubkey* pk, size_t batch, const uint8_t* ct, size_t ct_stride, const uint8_t* a,
                                  const uint8_t* b, size_t ab_stride, uint8_t* out, size_t out_stride, int mem);

/* PublicKey.verifyDDLEQProofInstance for a batch of (statement, instance) pairs (ddleq.go:129-153), entirely on the
 * device: Fiat-Shamir bit = LSB(SHA-256(ct2||X||Y||Alpha)), check = bit ? ct2 : ct1,
 * ok[i] = (check^(E^n mod n^2) * F^(n^2) mod n^3 == Alpha).  ct/alpha strides = byte length of n^3.  ok: host int32[batch]. */
int pgpu_ddleq_verify(const pgpu_pubkey* pk, size_t batch, const uint8_t* ct1, const uint8_t* ct2, size_t ct_stride,
                      const uint8_t* x, const uint8_t* y, size_t xy_stride, const uint8_t* alpha, size_t alpha_stride,
                      const uint8_t* e, size_t e_stride, const uint8_t* f, size_t f_stride, int32_t* ok, int mem);

/* SecretKey.proveDDLEQInstance for a batch of (statement, instance) pairs (ddleq.go:55-127) with the random draws x, y
 * supplied: sanity check (a false statement returns PGPU_ERR_INVALID where the reference panics), Alpha, Fiat-Shamir
 * bit on the device, and for bit = 1 the response (E, F) through the level-two ExtractRandonness (operations.go:75-91).
 * ct strides = byte length of n^3 (also used for alpha and f_out); a, b, x, y share n_stride; e_stride >= bytes of n^2. */
int pgpu_ddleq_prove(const pgpu_seckey* sk, size_t batch, const uint8_t* ct1, const uint8_t* ct2, size_t ct_stride,
                     const uint8_t* a, const uint8_t* b, const uint8_t* x, const uint8_t* y, size_t n_stride, uint8_t* alpha,
                     uint8_t* e_out, size_t e_stride, uint8_t* f_out, int mem);

/* SecretKey.ProveDDLEQ (ddleq.go:27-40) for a batch of statements: `secpar` instances per statement (soundness 1 - 2^-secpar
 * each), draws supplied.  ct1, ct2, a, b hold n_statements rows; x, y, alpha, e_out, f_out hold n_statements * secpar rows,
 * statement-major (instance k of statement j in row j * secpar + k).  What proveDDLEQInstance (ddleq.go:55-127) recomputes in
 * every instance although it depends on the statement only -- the sanity check (:62-69), a^n, a^-1, ExtractRandonness(ct1)
 * (:103) -- is computed once per statement; the integers are those of `secpar` calls of proveDDLEQInstance with the same
 * draws.  Strides as for pgpu_ddleq_prove, which is the secpar = 1 case.  A false statement returns PGPU_ERR_INVALID. */
int pgpu_ddleq_prove_secpar(const pgpu_seckey* sk, size_t n_statements, size_t secpar, const uint8_t* ct1, const uint8_t* ct2,
                            size_t ct_stride, const uint8_t* a, const uint8_t* b, const uint8_t* x, const uint8_t* y, size_t n_stride,
                            uint8_t* alpha, uint8_t* e_out, size_t e_stride, uint8_t* f_out, int mem);

/* ---- generic modular batch primitives (the gmp.Int seam: Exp / Mul+Mod) ------------------------ */

/* Load an odd modulus (big-endian).  Precomputes -N^-1 mod 2^28, R mod N, R^2 mod N, R^3 mod N. */
int pgpu_modulus_create(pgpu_ctx* ctx, const uint8_t* n_be, size_t n_len, pgpu_modulus** out);
void pgpu_modulus_destroy(pgpu_modulus* mod);
size_t pgpu_modulus_bytes(const pgpu_modulus* mod);

/* gmp.Int.Exp(base, e, N) for a batch: out[i] = base[i]^e mod N (e == 0 -> 1).
 * e_stride == 0: shared exponent; otherwise one exponent per element.  base may be >= N
 * (up to twice the modulus width), as with mpz_powm. */
int pgpu_modexp(const pgpu_modulus* mod, size_t batch, const uint8_t* base, size_t base_stride, size_t base_len,
                const uint8_t* e, size_t e_len, size_t e_stride, uint8_t* out, size_t out_stride, int mem);

/* gmp.Int.ModInverse(x, N) for a batch (Montgomery's trick as a parallel tree: 3 modular products per element and
 * one host inversion per call).  x may be up to twice the modulus width.  Non-units: out[i] = 0 and
 * status[i] = PGPU_LANE_NOT_INVERTIBLE (status: optional host int32[batch]; NULL: PGPU_ERR_NOT_INVERTIBLE is returned after
 * the invertible lanes have been written).  The non-units are found by a per-lane binary GCD on the device, run only when
 * the single inversion of the tree fails. */
int pgpu_modinv(const pgpu_modulus* mod, size_t batch, const uint8_t* x, size_t x_stride, size_t x_len, uint8_t* out,
                size_t out_stride, int mem, int32_t* status);

/* new(gmp.Int).Mod(new(gmp.Int).Mul(a, b), N) for a batch. */
int pgpu_modmul(const pgpu_modulus* mod, size_t batch, const uint8_t* a, size_t a_stride, size_t a_len,
                const uint8_t* b, size_t b_stride, size_t b_len, uint8_t* out, size_t out_stride, int mem);

/* ---- wire format: Ciphertext.Bytes() / PublicKey.NewCiphertextFromBytes (paillier.go:374-401) for a batch ------------------
 * The reference serialises a ciphertext with encoding/gob (a fresh encoder per ciphertext: every blob carries the type
 * definitions of Ciphertext{C *gmp.Int; Level; EncMethod} and of gmp.Int's GobEncoder form, then the value).  These two entry
 * points move a whole batch between that format and the flat fixed-stride big-endian buffers every other entry point takes, so
 * that batches can arrive and leave in the reference's own format.  Blobs live in HOST memory (they come from / go to the
 * network), concatenated, blob i = bytes [offsets[i], offsets[i+1]); the flat buffer lives where `mem` says -- with
 * PGPU_MEM_DEVICE the payload bytes are moved by a kernel and never return to the host (feed pgpu_decrypt(..., PGPU_MEM_DEVICE)
 * directly).  With PGPU_MEM_HOST nothing touches the device and ctx may be NULL.
 * pgpu_gob_unpack: levels / methods (optional host int32[batch]) receive Ciphertext.Level / EncMethod.  Accepts any gob type ids
 * and field order (fields match by name), as Go's decoder does, and skips value fields Ciphertext does not have when they are of
 * gob's basic types or GobEncoder values (extra fields of other types -- nested structs, slices, maps -- are an error; Go would
 * skip those too).  Errors as NewCiphertextFromBytes ("no data provided", malformed data) fail the call with PGPU_ERR_INVALID;
 * so do a value without C, a negative C and a C wider than out_stride.  Blobs are untrusted input: every length, count and
 * field delta is checked against the bytes that are there before it is used (tests/test_wire.py's malformed cases).
 * pgpu_gob_pack: every ciphertext of the batch carries the same Level / EncMethod.  blobs_cap: capacity of `blobs`;
 * batch * pgpu_gob_max_bytes(stride) always suffices.  offsets: host size_t[batch + 1], written.
 * Parity: byte-identical to the restatement in paillier_amd/wire.py, which is pinned to the gob specification's own example and
 * math/big's documented GobEncode layout -- NOT to a Go toolchain (none in the build image: see INTEGRATION.md). */
size_t pgpu_gob_max_bytes(size_t value_bytes);
int pgpu_gob_unpack(pgpu_ctx* ctx, size_t batch, const uint8_t* blobs, const size_t* offsets, uint8_t* out, size_t out_stride,
                    int mem, int32_t* levels, int32_t* methods);
int pgpu_gob_pack(pgpu_ctx* ctx, size_t batch, const uint8_t* in, size_t stride, int mem, int level, int enc_method, uint8_t* blobs,
                  size_t blobs_cap, size_t* offsets);

/* (Test hooks -- raw VM programs, the planning predicates -- are NOT part of this boundary: include/paillier_hip_debug.h.) */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* PAILLIER_HIP_H */
