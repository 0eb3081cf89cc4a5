// batch.go -- batch variants of the reference's exported hot-path methods, one C call per batch.
// Every function cites the scalar method it is the batch form of; results are bit-identical to calling that method in a
// loop (tests of this repository compare the C ABI with a restatement of the reference and with libgmp).
// See gpu.go for build notes (not compiled here: no Go toolchain in the build image).
package paillier

/*
#include <stdint.h>
#include <stdlib.h>
#include "paillier_hip.h"
*/
import "C"

import (
	"errors"
	"unsafe"

	gmp "github.com/ncw/gmp"
)

func cts(values []*gmp.Int, level EncryptionLevel, method EncryptionMethod) []*Ciphertext {
	out := make([]*Ciphertext, len(values))
	for i, v := range values {
		out[i] = &Ciphertext{v, level, method}
	}
	return out
}

func cvals(c []*Ciphertext) []*gmp.Int {
	out := make([]*gmp.Int, len(c))
	for i, x := range c {
		out[i] = x.C
	}
	return out
}

// EncryptWithRBatch: PublicKey.EncryptWithRAtLevel (paillier.go:206-218) for every (m[i], r[i]).
func (k *GPUPublicKey) EncryptWithRBatch(m, r []*gmp.Int, level EncryptionLevel) ([]*Ciphertext, error) {
	if len(m) != len(r) {
		return nil, errors.New("paillier: len(m) != len(r)")
	}
	ms, rs, cs := maxLen(m, k.plainBytes(level)), maxLen(r, k.plainBytes(EncLevelOne)), k.cipherBytes(level)
	mb, rb, out := pack(m, ms), pack(r, rs), make([]byte, len(m)*cs)
	rc := C.pgpu_encrypt_with_r(k.h, C.int(level), C.size_t(len(m)), p8(mb), C.size_t(ms), p8(rb), C.size_t(rs), p8(out),
		C.size_t(cs), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	return cts(unpack(out, cs), level, RegularEncryption), nil
}

// EncryptWithRBatch on the SECRET key: SecretKey embeds PublicKey (paillier.go:59-62), so sk.EncryptWithR is the same
// method; the key holder's r^n goes through p^2 and q^2 (pgpu_encrypt_with_r_sk): identical ciphertexts, a third of the work.
func (s *GPUSecretKey) EncryptWithRBatch(m, r []*gmp.Int, level EncryptionLevel) ([]*Ciphertext, error) {
	if len(m) != len(r) {
		return nil, errors.New("paillier: len(m) != len(r)")
	}
	k := s.pub
	ms, rs, cs := maxLen(m, k.plainBytes(level)), maxLen(r, k.plainBytes(EncLevelOne)), k.cipherBytes(level)
	mb, rb, out := pack(m, ms), pack(r, rs), make([]byte, len(m)*cs)
	rc := C.pgpu_encrypt_with_r_sk(s.h, C.int(level), C.size_t(len(m)), p8(mb), C.size_t(ms), p8(rb), C.size_t(rs), p8(out),
		C.size_t(cs), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	return cts(unpack(out, cs), level, RegularEncryption), nil
}

// EncryptBatch: PublicKey.EncryptAtLevel (paillier.go:258-269): one fresh r in Z_n^* per message, drawn by the library
// from the operating system's CSPRNG exactly as utils.go:36-49 does (uniform below n by rejection, gcd(r, n) = 1).
func (k *GPUPublicKey) EncryptBatch(m []*gmp.Int, level EncryptionLevel) ([]*Ciphertext, error) {
	ms, cs := maxLen(m, k.plainBytes(level)), k.cipherBytes(level)
	mb, out := pack(m, ms), make([]byte, len(m)*cs)
	rc := C.pgpu_encrypt(k.h, C.int(level), C.size_t(len(m)), p8(mb), C.size_t(ms), p8(out), C.size_t(cs), nil, 0, C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	return cts(unpack(out, cs), level, RegularEncryption), nil
}

// AltEncryptWithRBatch: PublicKey.AltEncryptWithRAtLevel (paillier.go:221-238).  As the reference does, r[i] is
// overwritten with r[i] mod K.
func (k *GPUPublicKey) AltEncryptWithRBatch(m, r []*gmp.Int, level EncryptionLevel) ([]*Ciphertext, error) {
	ms, rs, cs := maxLen(m, k.plainBytes(level)), maxLen(r, 1), k.cipherBytes(level)
	mb, rb, out, red := pack(m, ms), pack(r, rs), make([]byte, len(m)*cs), make([]byte, len(m)*rs)
	rc := C.pgpu_alt_encrypt_with_r(k.h, C.int(level), C.size_t(len(m)), p8(mb), C.size_t(ms), p8(rb), C.size_t(rs), p8(out),
		C.size_t(cs), p8(red), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	for i, v := range unpack(red, rs) {
		r[i].Set(v) // paillier.go:228 mutates the caller's r
	}
	return cts(unpack(out, cs), level, AlternativeEncryption), nil
}

// AltEncryptBatch: PublicKey.AltEncryptAtLevel (paillier.go:244-255): one fresh r in Z_n^* per message (the library's draw,
// as in EncryptBatch), then the alternative encryption with it.
func (k *GPUPublicKey) AltEncryptBatch(m []*gmp.Int, level EncryptionLevel) ([]*Ciphertext, error) {
	rs := k.plainBytes(EncLevelOne)
	rb := make([]byte, len(m)*rs)
	if err := status(C.pgpu_random_units(k.h, C.size_t(len(m)), p8(rb), C.size_t(rs), C.PGPU_MEM_HOST)); err != nil {
		return nil, err
	}
	return k.AltEncryptWithRBatch(m, unpack(rb, rs), level)
}

// EncryptZeroBatch / EncryptOneBatch: `count` fresh encryptions of 0 / 1 (paillier.go:272-289).
func (k *GPUPublicKey) EncryptZeroBatch(count int, level EncryptionLevel) ([]*Ciphertext, error) {
	return k.EncryptBatch(constants(count, 0), level)
}
func (k *GPUPublicKey) EncryptOneBatch(count int, level EncryptionLevel) ([]*Ciphertext, error) {
	return k.EncryptBatch(constants(count, 1), level)
}
func constants(count int, v int64) []*gmp.Int {
	out := make([]*gmp.Int, count)
	for i := range out {
		out[i] = gmp.NewInt(v)
	}
	return out
}

// DecryptBatch: SecretKey.Decrypt (paillier.go:292-303) for every ciphertext (all of one level).
func (s *GPUSecretKey) DecryptBatch(c []*Ciphertext) ([]*gmp.Int, error) {
	if len(c) == 0 {
		return nil, nil
	}
	level := c[0].Level
	cs, ps := s.pub.cipherBytes(level), s.pub.plainBytes(level)
	cb, out := pack(cvals(c), maxLen(cvals(c), cs)), make([]byte, len(c)*ps)
	rc := C.pgpu_decrypt(s.h, C.int(level), C.size_t(len(c)), p8(cb), C.size_t(len(cb)/len(c)), p8(out), C.size_t(ps),
		C.PGPU_MEM_HOST, C.PGPU_DECRYPT_DEFAULT, nil)
	if err := status(rc); err != nil {
		return nil, err
	}
	return unpack(out, ps), nil
}

func (k *GPUPublicKey) many(ops [][]*Ciphertext, sub bool) ([]*Ciphertext, error) {
	if len(ops) == 0 || len(ops[0]) == 0 {
		panic("runtime error: index out of range") // operations.go:13 cts[0]
	}
	level, batch := ops[0][0].Level, len(ops[0])
	cs := k.cipherBytes(level)
	stride := cs
	for _, op := range ops {
		stride = maxLen(cvals(op), stride)
	}
	bufs := make([][]byte, len(ops))
	ptrs := (*[1 << 20]*C.uint8_t)(C.malloc(C.size_t(len(ops)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(ptrs))
	for i, op := range ops {
		bufs[i] = pack(cvals(op), stride)
		ptrs[i] = p8(bufs[i])
	}
	os := cs
	if sub && len(ops) == 1 {
		os = stride // Sub with one operand returns it unreduced (operations.go:34)
	}
	out := make([]byte, batch*os)
	var rc C.int
	if sub {
		rc = C.pgpu_sub_many(k.h, C.int(level), C.int(len(ops)), C.size_t(batch), &ptrs[0], C.size_t(stride), p8(out), C.size_t(os),
			C.PGPU_MEM_HOST, nil)
	} else {
		rc = C.pgpu_add_many(k.h, C.int(level), C.int(len(ops)), C.size_t(batch), &ptrs[0], C.size_t(stride), p8(out), C.size_t(os),
			C.PGPU_MEM_HOST)
	}
	if err := status(rc); err != nil {
		return nil, err
	}
	return cts(unpack(out, os), level, MixedEncryption), nil
}

// AddBatch: PublicKey.Add(cts...) (operations.go:11-29), element-wise: out[i] = Add(ops[0][i], ops[1][i], ...).
func (k *GPUPublicKey) AddBatch(ops ...[]*Ciphertext) ([]*Ciphertext, error) { return k.many(ops, false) }

// SubBatch: PublicKey.Sub(cts...) (operations.go:32-55), element-wise.
func (k *GPUPublicKey) SubBatch(ops ...[]*Ciphertext) ([]*Ciphertext, error) { return k.many(ops, true) }

// ConstMultBatch: PublicKey.ConstMult (operations.go:58-64) with one constant per ciphertext (len(ks) == len(c)) or one
// shared constant (len(ks) == 1).  NestedAdd (operations.go:121-127) is ConstMultBatch(ct1, values of ct2).
func (k *GPUPublicKey) ConstMultBatch(c []*Ciphertext, ks []*gmp.Int) ([]*Ciphertext, error) {
	level := c[0].Level
	cs := k.cipherBytes(level)
	cb, out := pack(cvals(c), cs), make([]byte, len(c)*cs)
	kl := maxLen(ks, 1)
	kb := pack(ks, kl)
	kstride := kl
	if len(ks) == 1 {
		kstride = 0
	} else if len(ks) != len(c) {
		return nil, errors.New("paillier: one constant per ciphertext, or exactly one")
	}
	rc := C.pgpu_const_mult(k.h, C.int(level), C.size_t(len(c)), p8(cb), C.size_t(cs), p8(kb), C.size_t(kl), C.size_t(kstride),
		p8(out), C.size_t(cs), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	res := cts(unpack(out, cs), level, c[0].EncMethod)
	return res, nil
}

// ---- threshold decryption (thresholdkey.go) ----------------------------------------------------------------------------

// PartialDecryptBatch: ThresholdSecretKey.PartialDecrypt (thresholdkey.go:192-201) for every ciphertext.
func (k *GPUPublicKey) PartialDecryptBatch(tsk *ThresholdSecretKey, c []*gmp.Int) ([]*PartialDecryption, error) {
	cs := k.cipherBytes(EncLevelOne)
	cb, out, sh := pack(c, cs), make([]byte, len(c)*cs), bytesOf(tsk.Share)
	rc := C.pgpu_partial_decrypt(k.h, C.int(tsk.TotalNumberOfDecryptionServers), p8(sh), C.size_t(len(sh)), C.size_t(len(c)),
		p8(cb), C.size_t(cs), p8(out), C.size_t(cs), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	res := make([]*PartialDecryption, len(c))
	for i, v := range unpack(out, cs) {
		res[i] = &PartialDecryption{tsk.ID, v}
	}
	return res, nil
}

// PartialDecryptMultiBatch: the same ciphertexts under SEVERAL servers' shares held by one process (the shape of the
// reference's BenchmarkThresholdDecrypt): res[k][i] = tsks[k].PartialDecrypt(c[i]).  From 8 192 ciphertexts up the library
// walks ONE chain of squarings for all the shares (pgpu_partial_decrypt_multi): three servers cost about 1.5 ladders.
func (k *GPUPublicKey) PartialDecryptMultiBatch(tsks []*ThresholdSecretKey, c []*gmp.Int) ([][]*PartialDecryption, error) {
	if len(tsks) == 0 {
		return nil, nil
	}
	cs := k.cipherBytes(EncLevelOne)
	cb := pack(c, cs)
	n := len(tsks)
	shp := (*[1 << 20]*C.uint8_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	outp := (*[1 << 20]*C.uint8_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	lens := (*[1 << 20]C.size_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(C.size_t(0)))))
	defer C.free(unsafe.Pointer(shp))
	defer C.free(unsafe.Pointer(outp))
	defer C.free(unsafe.Pointer(lens))
	shs, outs := make([][]byte, n), make([][]byte, n)
	for i, t := range tsks {
		shs[i], outs[i] = bytesOf(t.Share), make([]byte, len(c)*cs)
		shp[i], outp[i], lens[i] = p8(shs[i]), p8(outs[i]), C.size_t(len(shs[i]))
	}
	rc := C.pgpu_partial_decrypt_multi(k.h, C.int(tsks[0].TotalNumberOfDecryptionServers), C.int(n), &shp[0], &lens[0], C.size_t(len(c)),
		p8(cb), C.size_t(cs), &outp[0], C.size_t(cs), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	res := make([][]*PartialDecryption, n)
	for j, t := range tsks {
		res[j] = make([]*PartialDecryption, len(c))
		for i, v := range unpack(outs[j], cs) {
			res[j][i] = &PartialDecryption{t.ID, v}
		}
	}
	return res, nil
}

// CombinePartialDecryptionsBatch: ThresholdPublicKey.CombinePartialDecryptions (thresholdkey.go:149-161) for a batch of
// ciphertexts: shares[k][i] is server k's partial decryption of ciphertext i (every shares[k] from one server).
func (k *GPUPublicKey) CombinePartialDecryptionsBatch(tk *ThresholdPublicKey, shares [][]*PartialDecryption) ([]*gmp.Int, error) {
	if len(shares) == 0 {
		return nil, errors.New("Threshold not meet")
	}
	batch, cs, ps := len(shares[0]), k.cipherBytes(EncLevelOne), k.plainBytes(EncLevelOne)
	ids := make([]C.int, len(shares))
	bufs := make([][]byte, len(shares))
	ptrs := (*[1 << 20]*C.uint8_t)(C.malloc(C.size_t(len(shares)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(ptrs))
	for s, col := range shares {
		vals := make([]*gmp.Int, len(col))
		for i, pd := range col {
			vals[i] = pd.Decryption
		}
		ids[s] = C.int(col[0].ID)
		bufs[s] = pack(vals, cs)
		ptrs[s] = p8(bufs[s])
	}
	out := make([]byte, batch*ps)
	rc := C.pgpu_combine_partial_decryptions(k.h, C.int(tk.TotalNumberOfDecryptionServers), C.int(tk.Threshold), C.int(len(shares)),
		&ids[0], C.size_t(batch), &ptrs[0], C.size_t(cs), p8(out), C.size_t(ps), C.PGPU_MEM_HOST, nil)
	if rc == C.PGPU_ERR_THRESHOLD {
		return nil, errors.New(C.GoString(C.pgpu_last_error())) // the reference's own messages (thresholdkey.go:77-89)
	}
	if err := status(rc); err != nil {
		return nil, err
	}
	return unpack(out, ps), nil
}

// PartialDecryptionWithZKPBatch: ThresholdSecretKey.PartialDecryptionWithZKP (thresholdkey.go:225-257) for every
// ciphertext, with the random r[i] < n^2 (thresholdkey.go:233) supplied by the caller.
func (k *GPUPublicKey) PartialDecryptionWithZKPBatch(tsk *ThresholdSecretKey, c, r []*gmp.Int) ([]*PartialDecryptionZKP, error) {
	cs := k.cipherBytes(EncLevelOne)
	zs := cs + 48
	cb, rb := pack(c, cs), pack(r, cs)
	dec, e, z := make([]byte, len(c)*cs), make([]byte, len(c)*32), make([]byte, len(c)*zs)
	sh, vk := bytesOf(tsk.Share), bytesOf(tsk.VerificationKey)
	rc := C.pgpu_share_zkp_prove(k.h, C.int(tsk.TotalNumberOfDecryptionServers), p8(sh), C.size_t(len(sh)), p8(vk), C.size_t(len(vk)),
		C.size_t(len(c)), p8(cb), C.size_t(cs), p8(rb), C.size_t(cs), p8(dec), C.size_t(cs), p8(e), p8(z), C.size_t(zs), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	ds, es, zz := unpack(dec, cs), unpack(e, 32), unpack(z, zs)
	out := make([]*PartialDecryptionZKP, len(c))
	for i := range c {
		out[i] = &PartialDecryptionZKP{PartialDecryption: PartialDecryption{tsk.ID, ds[i]}, Key: tsk.PublicKey(), E: es[i], Z: zz[i], C: c[i]}
	}
	return out, nil
}

// VerifyProofBatch: PartialDecryptionZKP.VerifyProof (thresholdkey.go:278-311) for proofs of ONE server.
func (k *GPUPublicKey) VerifyProofBatch(tk *ThresholdPublicKey, proofs []*PartialDecryptionZKP) ([]bool, error) {
	if len(proofs) == 0 {
		return nil, nil
	}
	cs := k.cipherBytes(EncLevelOne)
	n := len(proofs)
	c, d, e, z := make([]*gmp.Int, n), make([]*gmp.Int, n), make([]*gmp.Int, n), make([]*gmp.Int, n)
	for i, p := range proofs {
		c[i], d[i], e[i], z[i] = p.C, p.Decryption, p.E, p.Z
	}
	zs := maxLen(z, cs+48)
	cb, db, eb, zb := pack(c, cs), pack(d, cs), pack(e, 32), pack(z, zs)
	vk, vi := bytesOf(tk.VerificationKey), bytesOf(tk.VerificationKeys[proofs[0].ID-1]) // thresholdkey.go:305
	ok := make([]C.int32_t, n)
	rc := C.pgpu_share_zkp_verify(k.h, p8(vk), C.size_t(len(vk)), p8(vi), C.size_t(len(vi)), C.size_t(n), p8(cb), C.size_t(cs), p8(db),
		C.size_t(cs), p8(eb), p8(zb), C.size_t(zs), &ok[0], C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	res := make([]bool, n)
	for i := range ok {
		res[i] = ok[i] != 0
	}
	return res, nil
}

// ---- DDLEQ (ddleq.go) ---------------------------------------------------------------------------------------------

// ProveDDLEQBatch: SecretKey.ProveDDLEQ (ddleq.go:27-40) with the draws supplied: instance i proves statement
// (ct1[i], ct2[i], a[i], b[i]) with randomness (x[i], y[i]).  For one statement with secpar instances, repeat the statement.
// A false statement returns an error where the reference panics (ddleq.go:68).
func (s *GPUSecretKey) ProveDDLEQBatch(ct1, ct2 []*Ciphertext, a, b, x, y []*gmp.Int) ([]*DDLEQProofInstance, error) {
	k := s.pub
	c3, p1, p2 := k.cipherBytes(EncLevelTwo), k.plainBytes(EncLevelOne), k.plainBytes(EncLevelTwo)
	n := len(ct1)
	c1b, c2b := pack(cvals(ct1), c3), pack(cvals(ct2), c3)
	ab, bb, xb, yb := pack(a, p1), pack(b, p1), pack(x, p1), pack(y, p1)
	al, e, f := make([]byte, n*c3), make([]byte, n*p2), make([]byte, n*c3)
	rc := C.pgpu_ddleq_prove(s.h, C.size_t(n), p8(c1b), p8(c2b), C.size_t(c3), p8(ab), p8(bb), p8(xb), p8(yb), C.size_t(p1), p8(al),
		p8(e), C.size_t(p2), p8(f), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	as, es, fs := unpack(al, c3), unpack(e, p2), unpack(f, c3)
	out := make([]*DDLEQProofInstance, n)
	for i := range out {
		out[i] = &DDLEQProofInstance{X: x[i], Y: y[i], Alpha: as[i], E: es[i], F: fs[i]}
	}
	return out, nil
}

// NestedRandomizeWithABBatch: PublicKey.NestedRandomize (operations.go:96-118) with the draws a[i], b[i] supplied (draw them
// with GetRandomNumberInMultiplicativeGroup, as the reference does, to keep its distribution).
func (k *GPUPublicKey) NestedRandomizeWithABBatch(ct []*Ciphertext, a, b []*gmp.Int) ([]*Ciphertext, error) {
	c3, p1 := k.cipherBytes(EncLevelTwo), k.plainBytes(EncLevelOne)
	for _, c := range ct {
		if c.Level != EncLevelTwo {
			panic("can only homomorphically randomize doubly encrypted values") // operations.go:98
		}
	}
	cb, ab, bb, out := pack(cvals(ct), c3), pack(a, p1), pack(b, p1), make([]byte, len(ct)*c3)
	rc := C.pgpu_nested_randomize_with_ab(k.h, C.size_t(len(ct)), p8(cb), C.size_t(c3), p8(ab), p8(bb), C.size_t(p1), p8(out),
		C.size_t(c3), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	return cts(unpack(out, c3), EncLevelTwo, RegularEncryption), nil
}

// VerifyDDLEQBatch: PublicKey.verifyDDLEQProofInstance (ddleq.go:129-153) for every (statement, instance) pair;
// VerifyDDLEQProof (ddleq.go:44-53) is the conjunction over one statement's instances.
func (k *GPUPublicKey) VerifyDDLEQBatch(ct1, ct2 []*Ciphertext, proofs []*DDLEQProofInstance) ([]bool, error) {
	c3, p1, p2 := k.cipherBytes(EncLevelTwo), k.plainBytes(EncLevelOne), k.plainBytes(EncLevelTwo)
	n := len(proofs)
	x, y, al, e, f := make([]*gmp.Int, n), make([]*gmp.Int, n), make([]*gmp.Int, n), make([]*gmp.Int, n), make([]*gmp.Int, n)
	for i, p := range proofs {
		x[i], y[i], al[i], e[i], f[i] = p.X, p.Y, p.Alpha, p.E, p.F
	}
	ok := make([]C.int32_t, n)
	c1b, c2b, xb, yb, ab, eb, fb := pack(cvals(ct1), c3), pack(cvals(ct2), c3), pack(x, p1), pack(y, p1), pack(al, c3), pack(e, p2), pack(f, c3)
	rc := C.pgpu_ddleq_verify(k.h, C.size_t(n), p8(c1b), p8(c2b), C.size_t(c3), p8(xb), p8(yb), C.size_t(p1), p8(ab), C.size_t(c3),
		p8(eb), C.size_t(p2), p8(fb), C.size_t(c3), &ok[0], C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	res := make([]bool, n)
	for i := range ok {
		res[i] = ok[i] != 0
	}
	return res, nil
}

// ---- the gmp.Int seam: Exp / Mul+Mod / ModInverse for a batch under one modulus ---------------------------------------------

// Modulus is an odd modulus loaded on the device (Montgomery constants precomputed).
type Modulus struct {
	g *GPU
	h *C.pgpu_modulus
	n int
}

func (g *GPU) NewModulus(m *gmp.Int) (*Modulus, error) {
	b := bytesOf(m)
	var h *C.pgpu_modulus
	if err := status(C.pgpu_modulus_create(g.ctx, p8(b), C.size_t(len(b)), &h)); err != nil {
		return nil, err
	}
	return &Modulus{g, h, int(C.pgpu_modulus_bytes(h))}, nil
}

func (m *Modulus) Close() { C.pgpu_modulus_destroy(m.h); m.h = nil }

// ExpBatch: new(gmp.Int).Exp(base[i], e[i], N) (len(e) == len(base)) or Exp(base[i], e[0], N) (len(e) == 1).
func (m *Modulus) ExpBatch(base, e []*gmp.Int) ([]*gmp.Int, error) {
	bs := maxLen(base, m.n)
	el := maxLen(e, 1)
	es := el
	if len(e) == 1 {
		es = 0
	}
	bb, eb, out := pack(base, bs), pack(e, el), make([]byte, len(base)*m.n)
	rc := C.pgpu_modexp(m.h, C.size_t(len(base)), p8(bb), C.size_t(bs), C.size_t(bs), p8(eb), C.size_t(el), C.size_t(es), p8(out),
		C.size_t(m.n), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	return unpack(out, m.n), nil
}

// MulBatch: new(gmp.Int).Mod(new(gmp.Int).Mul(a[i], b[i]), N).
func (m *Modulus) MulBatch(a, b []*gmp.Int) ([]*gmp.Int, error) {
	ab, bb, out := pack(a, m.n), pack(b, m.n), make([]byte, len(a)*m.n)
	rc := C.pgpu_modmul(m.h, C.size_t(len(a)), p8(ab), C.size_t(m.n), C.size_t(m.n), p8(bb), C.size_t(m.n), C.size_t(m.n), p8(out),
		C.size_t(m.n), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	return unpack(out, m.n), nil
}

// InvBatch: new(gmp.Int).ModInverse(x[i], N); invertible[i] is false where x[i] is not a unit (its result is 0).
func (m *Modulus) InvBatch(x []*gmp.Int) (inv []*gmp.Int, invertible []bool, err error) {
	xs := maxLen(x, m.n)
	xb, out := pack(x, xs), make([]byte, len(x)*m.n)
	st := make([]C.int32_t, len(x))
	rc := C.pgpu_modinv(m.h, C.size_t(len(x)), p8(xb), C.size_t(xs), C.size_t(xs), p8(out), C.size_t(m.n), C.PGPU_MEM_HOST, &st[0])
	if err = status(rc); err != nil {
		return nil, nil, err
	}
	invertible = make([]bool, len(x))
	for i := range st {
		invertible[i] = st[i]&C.PGPU_LANE_NOT_INVERTIBLE == 0
	}
	return unpack(out, m.n), invertible, nil
}

// RandomOracleDigestBatch: RandomOracleDigest (random_oracle.go:20-32) for a batch of argument tuples, on the device.
// As in the reference, the FIRST argument does not enter the hash: pass cols without it.
func (g *GPU) RandomOracleDigestBatch(cols ...[]*gmp.Int) ([][32]byte, error) {
	n := len(cols[0])
	bufs := make([][]byte, len(cols))
	strides := make([]C.size_t, len(cols))
	ptrs := (*[1 << 20]*C.uint8_t)(C.malloc(C.size_t(len(cols)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(ptrs))
	for i, col := range cols {
		s := maxLen(col, 1)
		bufs[i], strides[i] = pack(col, s), C.size_t(s)
		ptrs[i] = p8(bufs[i])
	}
	out := make([]byte, n*32)
	if err := status(C.pgpu_random_oracle_digest(g.ctx, C.int(len(cols)), &ptrs[0], &strides[0], C.size_t(n), p8(out), C.PGPU_MEM_HOST)); err != nil {
		return nil, err
	}
	res := make([][32]byte, n)
	for i := range res {
		copy(res[i][:], out[i*32:])
	}
	return res, nil
}
