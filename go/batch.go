// batch.go -- batch variants of the reference's exported hot-path methods, one C call per batch.
// Every function cites the scalar method it is the batch form of; results are bit-identical to calling that method in a
// loop (tests of this repository compare the C ABI with a restatement of the reference and with libgmp).
// See gpu.go for build notes (not compiled here: no Go toolchain in the build image).
package paillier

/*
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
	if len(ops) == 0 || len(ops[0]) == 0 {
		panic("runtime error: index out of range") // operations.go:13 cts[0]
	}
	level, batch := ops[0][0].Level, len(ops[0])
	cs := k.cipherBytes(level)
	stride := cs
	for _, op := range ops {
		if len(op) != batch { // the C side reads `batch` rows of every operand
			return nil, errors.New("paillier: operand batches differ in length")
		}
		stride = maxLen(cvals(op), stride)
	}
	if len(ops) == 2 {
		// the common two-operand form: flat buffers as direct arguments (pgpu_add / pgpu_sub), no pointer array, no C copies
		a, b, out := pack(cvals(ops[0]), stride), pack(cvals(ops[1]), stride), make([]byte, batch*cs)
		var rc C.int
		if sub {
			rc = C.pgpu_sub(k.h, C.int(level), C.size_t(batch), p8(a), C.size_t(stride), p8(b), C.size_t(stride), p8(out), C.size_t(cs),
				C.PGPU_MEM_HOST, nil)
		} else {
			rc = C.pgpu_add(k.h, C.int(level), C.size_t(batch), p8(a), C.size_t(stride), p8(b), C.size_t(stride), p8(out), C.size_t(cs),
				C.PGPU_MEM_HOST)
		}
		if err := status(rc); err != nil {
			return nil, err
		}
		return cts(unpack(out, cs), level, MixedEncryption), nil
	}
	in := newCbufs(len(ops)) // C copies of the operands: no Go pointer is stored in C memory
	defer in.free()
	for i, op := range ops {
		in.in(i, pack(cvals(op), stride))
	}
	os := cs
	if sub && len(ops) == 1 {
		os = stride // Sub with one operand returns it unreduced (operations.go:34)
	}
	out := make([]byte, batch*os)
	var rc C.int
	if sub {
		rc = C.pgpu_sub_many(k.h, C.int(level), C.int(len(ops)), C.size_t(batch), in.array(), C.size_t(stride), p8(out), C.size_t(os),
			C.PGPU_MEM_HOST, nil)
	} else {
		rc = C.pgpu_add_many(k.h, C.int(level), C.int(len(ops)), C.size_t(batch), in.array(), C.size_t(stride), p8(out), C.size_t(os),
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
	if len(tsks) == 0 {
		return nil, nil
	}
	cs := k.cipherBytes(EncLevelOne)
	cb := pack(c, cs)
	n := len(tsks)
	shp, outp := newCbufs(n), newCbufs(n) // shares and outputs in C memory: no Go pointer is stored in a C array
	defer shp.free()
	defer outp.free()
	lens := (*[1 << 20]C.size_t)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.size_t(0)))))
	defer C.free(unsafe.Pointer(lens))
	for i, t := range tsks {
		sh := bytesOf(t.Share)
		shp.in(i, sh)
		outp.out(i, len(c)*cs)
		lens[i] = C.size_t(len(sh))
	}
	rc := C.pgpu_partial_decrypt_multi(k.h, C.int(tsks[0].TotalNumberOfDecryptionServers), C.int(n), shp.array(), &lens[0], C.size_t(len(c)),
		p8(cb), C.size_t(cs), outp.array(), C.size_t(cs), C.PGPU_MEM_HOST)
	for i := range tsks { // the C copies of the shares are secrets: clear them before they are freed
		C.memset(unsafe.Pointer(shp.ptrs[i]), 0, C.size_t(shp.lens[i]))
	}
	if err := status(rc); err != nil {
		return nil, err
	}
	res := make([][]*PartialDecryption, n)
	for j, t := range tsks {
		res[j] = make([]*PartialDecryption, len(c))
		for i, v := range unpack(outp.bytes(j, len(c)*cs), cs) {
			res[j][i] = &PartialDecryption{t.ID, v}
		}
	}
	return res, nil
}

// PartialDecryptIndexedBatch: the (server, ciphertext) units of SEVERAL servers in one launch (what one rank of the sharded
// threshold flow holds: a server-major range of units touches at most two servers' shares): res[i] =
// tsks[serverIndex[i]].PartialDecrypt(c[i]).  Runs of units under one share become shared-exponent ladders side by side
// (pgpu_partial_decrypt_indexed).
func (k *GPUPublicKey) PartialDecryptIndexedBatch(tsks []*ThresholdSecretKey, serverIndex []int, c []*gmp.Int) ([]*PartialDecryption, error) {
	defer pin()()
	if len(tsks) == 0 || len(serverIndex) != len(c) {
		return nil, errors.New("paillier: one server index per ciphertext")
	}
	cs := k.cipherBytes(EncLevelOne)
	cb, out := pack(c, cs), make([]byte, len(c)*cs)
	n := len(tsks)
	shp := newCbufs(n)
	defer shp.free()
	lens := (*[1 << 20]C.size_t)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.size_t(0)))))
	defer C.free(unsafe.Pointer(lens))
	for i, t := range tsks {
		sh := bytesOf(t.Share)
		shp.in(i, sh)
		lens[i] = C.size_t(len(sh))
	}
	idx := make([]C.int32_t, len(c))
	for i, s := range serverIndex {
		if s < 0 || s >= n {
			return nil, errors.New("paillier: server index out of range")
		}
		idx[i] = C.int32_t(s)
	}
	rc := C.pgpu_partial_decrypt_indexed(k.h, C.int(tsks[0].TotalNumberOfDecryptionServers), C.int(n), shp.array(), &lens[0],
		C.size_t(len(c)), p8(cb), C.size_t(cs), &idx[0], p8(out), C.size_t(cs), C.PGPU_MEM_HOST)
	for i := range tsks {
		C.memset(unsafe.Pointer(shp.ptrs[i]), 0, C.size_t(shp.lens[i]))
	}
	if err := status(rc); err != nil {
		return nil, err
	}
	res := make([]*PartialDecryption, len(c))
	for i, v := range unpack(out, cs) {
		res[i] = &PartialDecryption{tsks[serverIndex[i]].ID, v}
	}
	return res, nil
}

// PartialDecryptUnitsBatch: what ONE rank of the sharded threshold flow computes (BASELINE config 4 over N GPUs).  The
// (server, ciphertext) units u = s*len(c) + i of the whole job -- server-major over tsks and the ONE ciphertext batch c -- are
// split into contiguous ranges; this call computes the range [unitBegin, unitEnd): res[u - unitBegin] =
// tsks[u / len(c)].PartialDecrypt(c[u % len(c)]) (thresholdkey.go:192-201).  Unlike PartialDecryptIndexedBatch the library sees
// that two units are the SAME ciphertext under two shares: such ciphertexts walk one chain of squarings for both exponents
// (pgpu_partial_decrypt_units; a rank of two holds one server whole and half of the next).  Only the shares of the servers
// the range touches are read.  A holder of EVERY share shards by ciphertexts instead: rank r calls this with its slice of c and the
// slice's whole unit range [0, len(tsks)*len(slice)) and combines locally -- no exchange (paillier_amd/dist.py
// threshold_decrypt_ciphertext_major; the shares of a ciphertext then split into chains of their own as the batch shrinks).
func (k *GPUPublicKey) PartialDecryptUnitsBatch(tsks []*ThresholdSecretKey, c []*gmp.Int, unitBegin, unitEnd int) ([]*PartialDecryption, error) {
	defer pin()()
	n := len(tsks)
	if n == 0 || len(c) == 0 {
		return nil, errors.New("paillier: no shares or no ciphertexts")
	}
	if unitBegin < 0 || unitBegin >= unitEnd || unitEnd > n*len(c) {
		return nil, errors.New("paillier: unit range out of bounds")
	}
	cs := k.cipherBytes(EncLevelOne)
	cb, out := pack(c, cs), make([]byte, (unitEnd-unitBegin)*cs)
	shp := newCbufs(n) // the shares in C memory: no Go pointer is stored in a C array
	defer shp.free()
	lens := (*[1 << 20]C.size_t)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.size_t(0)))))
	defer C.free(unsafe.Pointer(lens))
	for i, t := range tsks {
		sh := bytesOf(t.Share)
		shp.in(i, sh)
		lens[i] = C.size_t(len(sh))
	}
	rc := C.pgpu_partial_decrypt_units(k.h, C.int(tsks[0].TotalNumberOfDecryptionServers), C.int(n), shp.array(), &lens[0],
		C.size_t(len(c)), p8(cb), C.size_t(cs), C.size_t(unitBegin), C.size_t(unitEnd), p8(out), C.size_t(cs), C.PGPU_MEM_HOST)
	for i := range tsks { // the C copies of the shares are secrets: clear them before they are freed
		C.memset(unsafe.Pointer(shp.ptrs[i]), 0, C.size_t(shp.lens[i]))
	}
	if err := status(rc); err != nil {
		return nil, err
	}
	res := make([]*PartialDecryption, unitEnd-unitBegin)
	for j, v := range unpack(out, cs) {
		res[j] = &PartialDecryption{tsks[(unitBegin+j)/len(c)].ID, v}
	}
	return res, nil
}

// CombinePartialDecryptionsBatch: ThresholdPublicKey.CombinePartialDecryptions (thresholdkey.go:149-161) for a batch of
// ciphertexts: shares[k][i] is server k's partial decryption of ciphertext i (every shares[k] from one server).
func (k *GPUPublicKey) CombinePartialDecryptionsBatch(tk *ThresholdPublicKey, shares [][]*PartialDecryption) ([]*gmp.Int, error) {
	defer pin()()
	if len(shares) == 0 {
		return nil, errors.New("Threshold not meet")
	}
	batch, cs, ps := len(shares[0]), k.cipherBytes(EncLevelOne), k.plainBytes(EncLevelOne)
	ids := make([]C.int, len(shares))
	in := newCbufs(len(shares)) // C copies of the servers' columns: no Go pointer is stored in C memory
	defer in.free()
	for s, col := range shares {
		vals := make([]*gmp.Int, len(col))
		for i, pd := range col {
			vals[i] = pd.Decryption
		}
		ids[s] = C.int(col[0].ID)
		in.in(s, pack(vals, cs))
	}
	out := make([]byte, batch*ps)
	rc := C.pgpu_combine_partial_decryptions(k.h, C.int(tk.TotalNumberOfDecryptionServers), C.int(tk.Threshold), C.int(len(shares)),
		&ids[0], C.size_t(batch), in.array(), C.size_t(cs), p8(out), C.size_t(ps), C.PGPU_MEM_HOST, nil)
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
	defer pin()()
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
	defer pin()()
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

// ProveDDLEQBatch: SecretKey.proveDDLEQInstance (ddleq.go:55-127) with the draws supplied: instance i proves statement
// (ct1[i], ct2[i], a[i], b[i]) with randomness (x[i], y[i]).  For whole proofs of secpar instances per statement use
// ProveDDLEQProofBatch below (per-statement work done once).  A false statement returns an error where the reference
// panics (ddleq.go:68).
func (s *GPUSecretKey) ProveDDLEQBatch(ct1, ct2 []*Ciphertext, a, b, x, y []*gmp.Int) ([]*DDLEQProofInstance, error) {
	defer pin()()
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
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
	defer pin()()
	n := len(cols[0])
	strides := make([]C.size_t, len(cols))
	in := newCbufs(len(cols)) // C copies of the columns: no Go pointer is stored in C memory
	defer in.free()
	for i, col := range cols {
		s := maxLen(col, 1)
		strides[i] = C.size_t(s)
		in.in(i, pack(col, s))
	}
	out := make([]byte, n*32)
	if err := status(C.pgpu_random_oracle_digest(g.ctx, C.int(len(cols)), in.array(), &strides[0], C.size_t(n), p8(out), C.PGPU_MEM_HOST)); err != nil {
		return nil, err
	}
	res := make([][32]byte, n)
	for i := range res {
		copy(res[i][:], out[i*32:])
	}
	return res, nil
}

// ---- wire format (paillier.go:374-401) ---------------------------------------------------------------------------------------

// CiphertextsFromBytesBatch: PublicKey.NewCiphertextFromBytes (paillier.go:376-391) for a batch of blobs as Ciphertext.Bytes()
// writes them (encoding/gob, a fresh encoder per ciphertext): one C call walks all the blobs (host threads) and returns the
// values in one flat buffer (pgpu_gob_unpack).  The errors of the reference's decoder ("no data provided", malformed data) fail
// the batch.  To keep the ciphertexts on the device for DecryptBatch-style calls on device buffers use the C entry point with
// PGPU_MEM_DEVICE directly.
func (k *GPUPublicKey) CiphertextsFromBytesBatch(blobs [][]byte) ([]*Ciphertext, error) {
	defer pin()()
	n := len(blobs)
	if n == 0 {
		return nil, nil
	}
	offs := make([]C.size_t, n+1)
	stride, total := 1, 0
	for i, b := range blobs {
		total += len(b)
		offs[i+1] = C.size_t(total)
		if len(b) > stride {
			stride = len(b) // the magnitude of C is shorter than its blob
		}
	}
	cat := make([]byte, 0, total+1)
	for _, b := range blobs {
		cat = append(cat, b...)
	}
	if len(cat) == 0 {
		cat = append(cat, 0)
	}
	out := make([]byte, n*stride)
	levels, methods := make([]C.int32_t, n), make([]C.int32_t, n)
	rc := C.pgpu_gob_unpack(k.g.ctx, C.size_t(n), p8(cat), &offs[0], p8(out), C.size_t(stride), C.PGPU_MEM_HOST, &levels[0], &methods[0])
	if err := status(rc); err != nil {
		return nil, err
	}
	res := make([]*Ciphertext, n)
	for i, v := range unpack(out, stride) {
		res[i] = &Ciphertext{C: v, Level: EncryptionLevel(levels[i]), EncMethod: EncryptionMethod(methods[i])}
	}
	return res, nil
}

// CiphertextBytesBatch: Ciphertext.Bytes() (paillier.go:393-401) for every ciphertext: res[i] is the gob blob of c[i]
// (pgpu_gob_pack; one C call per (Level, EncMethod) group, normally one).
func (k *GPUPublicKey) CiphertextBytesBatch(c []*Ciphertext) ([][]byte, error) {
	defer pin()()
	res := make([][]byte, len(c))
	type key struct {
		l EncryptionLevel
		m EncryptionMethod
	}
	groups := map[key][]int{}
	for i, ct := range c {
		kk := key{ct.Level, ct.EncMethod}
		groups[kk] = append(groups[kk], i)
	}
	for kk, idx := range groups {
		vals := make([]*gmp.Int, len(idx))
		for j, i := range idx {
			vals[j] = c[i].C
		}
		stride := maxLen(vals, 1)
		in := pack(vals, stride)
		cap := len(idx) * int(C.pgpu_gob_max_bytes(C.size_t(stride)))
		blobs := make([]byte, cap)
		offs := make([]C.size_t, len(idx)+1)
		rc := C.pgpu_gob_pack(k.g.ctx, C.size_t(len(idx)), p8(in), C.size_t(stride), C.PGPU_MEM_HOST, C.int(kk.l), C.int(kk.m), p8(blobs),
			C.size_t(cap), &offs[0])
		if err := status(rc); err != nil {
			return nil, err
		}
		for j, i := range idx {
			res[i] = append([]byte(nil), blobs[offs[j]:offs[j+1]]...)
		}
	}
	return res, nil
}

// ---- whole-protocol batch forms (the reference's signatures, one slice element per call of the scalar method) -----------------

func randomUnits(k *GPUPublicKey, count int) ([]*gmp.Int, error) {
	defer pin()()
	rs := k.plainBytes(EncLevelOne)
	rb := make([]byte, count*rs)
	if err := status(C.pgpu_random_units(k.h, C.size_t(count), p8(rb), C.size_t(rs), C.PGPU_MEM_HOST)); err != nil {
		return nil, err
	}
	return unpack(rb, rs), nil
}

// ProveDDLEQProofBatch: SecretKey.ProveDDLEQ (ddleq.go:27-40) for a batch of statements: one *DDLEQProof of `secpar`
// instances per statement (ct1[j], ct2[j], a[j], b[j]).  The draws x, y come from the library's CSPRNG path
// (GetRandomNumberInMultiplicativeGroup, utils.go:26-49: pgpu_random_units); what proveDDLEQInstance (ddleq.go:55-127)
// recomputes in every instance although it depends on the statement only -- the sanity check, a^n, a^-1,
// ExtractRandonness(ct1) -- is computed once per statement (pgpu_ddleq_prove_secpar).  A false statement returns an error
// where the reference panics (ddleq.go:68).
func (s *GPUSecretKey) ProveDDLEQProofBatch(secpar int, ct1, ct2 []*Ciphertext, a, b []*gmp.Int) ([]*DDLEQProof, error) {
	n := len(ct1)
	if len(ct2) != n || len(a) != n || len(b) != n || secpar < 1 {
		return nil, errors.New("paillier: one (ct1, ct2, a, b) per statement and secpar >= 1")
	}
	x, err := randomUnits(s.pub, n*secpar)
	if err != nil {
		return nil, err
	}
	y, err := randomUnits(s.pub, n*secpar)
	if err != nil {
		return nil, err
	}
	return s.ProveDDLEQProofWithXYBatch(secpar, ct1, ct2, a, b, x, y)
}

// ProveDDLEQProofWithXYBatch: the same with the draws supplied, statement-major: x[j*secpar + i], y[j*secpar + i] belong to
// instance i of statement j.
func (s *GPUSecretKey) ProveDDLEQProofWithXYBatch(secpar int, ct1, ct2 []*Ciphertext, a, b, x, y []*gmp.Int) ([]*DDLEQProof, error) {
	defer pin()()
	k := s.pub
	n := len(ct1)
	if len(x) != n*secpar || len(y) != n*secpar {
		return nil, errors.New("paillier: secpar draws (x, y) per statement")
	}
	c3, p1, p2 := k.cipherBytes(EncLevelTwo), k.plainBytes(EncLevelOne), k.plainBytes(EncLevelTwo)
	c1b, c2b := pack(cvals(ct1), c3), pack(cvals(ct2), c3)
	ab, bb, xb, yb := pack(a, p1), pack(b, p1), pack(x, p1), pack(y, p1)
	al, e, f := make([]byte, n*secpar*c3), make([]byte, n*secpar*p2), make([]byte, n*secpar*c3)
	rc := C.pgpu_ddleq_prove_secpar(s.h, C.size_t(n), C.size_t(secpar), p8(c1b), p8(c2b), C.size_t(c3), p8(ab), p8(bb), p8(xb), p8(yb),
		C.size_t(p1), p8(al), p8(e), C.size_t(p2), p8(f), C.PGPU_MEM_HOST)
	if err := status(rc); err != nil {
		return nil, err
	}
	as, es, fs := unpack(al, c3), unpack(e, p2), unpack(f, c3)
	out := make([]*DDLEQProof, n)
	for j := range out {
		p := &DDLEQProof{Instances: make([]*DDLEQProofInstance, secpar)}
		for i := 0; i < secpar; i++ {
			r := j*secpar + i
			p.Instances[i] = &DDLEQProofInstance{X: x[r], Y: y[r], Alpha: as[r], E: es[r], F: fs[r]}
		}
		out[j] = p
	}
	return out, nil
}

// VerifyDDLEQProofBatch: PublicKey.VerifyDDLEQProof (ddleq.go:44-53) for a batch of statements: ok[j] is true when EVERY
// instance of proofs[j] verifies for (ct1[j], ct2[j]); all instances of all statements go to the device as one batch.
func (k *GPUPublicKey) VerifyDDLEQProofBatch(ct1, ct2 []*Ciphertext, proofs []*DDLEQProof) ([]bool, error) {
	if len(ct2) != len(ct1) || len(proofs) != len(ct1) {
		return nil, errors.New("paillier: one proof per statement")
	}
	var f1, f2 []*Ciphertext
	var flat []*DDLEQProofInstance
	for j, p := range proofs {
		for _, in := range p.Instances {
			f1, f2, flat = append(f1, ct1[j]), append(f2, ct2[j]), append(flat, in)
		}
	}
	ok := make([]bool, len(proofs))
	for j := range ok {
		ok[j] = true // a proof without instances verifies (the loop of ddleq.go:46 does not run)
	}
	if len(flat) == 0 {
		return ok, nil
	}
	each, err := k.VerifyDDLEQBatch(f1, f2, flat)
	if err != nil {
		return nil, err
	}
	r := 0
	for j, p := range proofs {
		for range p.Instances {
			ok[j] = ok[j] && each[r]
			r++
		}
	}
	return ok, nil
}

// CombinePartialDecryptionsZKPBatch: ThresholdPublicKey.CombinePartialDecryptionsZKP (thresholdkey.go:164-172) for a batch of
// ciphertexts: shares[k][i] is server k's proof for ciphertext i.  A server whose proof fails FOR A CIPHERTEXT is dropped for
// that ciphertext, as the reference drops it; ciphertexts are regrouped by their surviving server set and combined per group.
// The error of a group ("Threshold not meet", thresholdkey.go:77) is returned as the reference returns it.
func (k *GPUPublicKey) CombinePartialDecryptionsZKPBatch(tk *ThresholdPublicKey, shares [][]*PartialDecryptionZKP) ([]*gmp.Int, error) {
	if len(shares) == 0 {
		return nil, errors.New("Threshold not meet")
	}
	batch := len(shares[0])
	valid := make([][]bool, len(shares))
	for s, col := range shares {
		v, err := k.VerifyProofBatch(tk, col)
		if err != nil {
			return nil, err
		}
		valid[s] = v
	}
	groups := map[string][]int{}
	keyOf := func(i int) string {
		b := make([]byte, len(shares))
		for s := range shares {
			if valid[s][i] {
				b[s] = 1
			}
		}
		return string(b)
	}
	for i := 0; i < batch; i++ {
		groups[keyOf(i)] = append(groups[keyOf(i)], i)
	}
	out := make([]*gmp.Int, batch)
	for key, idxs := range groups {
		var sub [][]*PartialDecryption
		for s := range shares {
			if key[s] == 1 {
				col := make([]*PartialDecryption, len(idxs))
				for j, i := range idxs {
					col[j] = &shares[s][i].PartialDecryption
				}
				sub = append(sub, col)
			}
		}
		res, err := k.CombinePartialDecryptionsBatch(tk, sub)
		if err != nil {
			return nil, err
		}
		for j, i := range idxs {
			out[i] = res[j]
		}
	}
	return out, nil
}

// VerifyDecryptionBatch: ThresholdPublicKey.VerifyDecryption (thresholdkey.go:175-189) for a batch, with its messages.
func (k *GPUPublicKey) VerifyDecryptionBatch(tk *ThresholdPublicKey, encrypted, decrypted []*gmp.Int, shares [][]*PartialDecryptionZKP) error {
	for _, col := range shares {
		for i, p := range col {
			if p.C.Cmp(encrypted[i]) != 0 {
				return errors.New("The encrypted message is not the same than the one in the shares")
			}
		}
	}
	res, err := k.CombinePartialDecryptionsZKPBatch(tk, shares)
	if err != nil {
		return err
	}
	for i, r := range res {
		if r.Cmp(decrypted[i]) != 0 {
			return errors.New("The decrypted message is not the same than the one in the shares")
		}
	}
	return nil
}

// NestedEncryptBatch: PublicKey.NestedEncrypt (paillier.go:200-203) for every message: the level-one ciphertext of m[i]
// (fresh r) becomes the plaintext of a level-two encryption (fresh r).
func (k *GPUPublicKey) NestedEncryptBatch(m []*gmp.Int) ([]*Ciphertext, error) {
	inner, err := k.EncryptBatch(m, EncLevelOne)
	if err != nil {
		return nil, err
	}
	return k.EncryptBatch(cvals(inner), EncLevelTwo)
}

// NestedDecryptBatch: SecretKey.NestedDecrypt (paillier.go:344-355): peel the level-two layer (DecryptNestedCiphertextLayer,
// :359-372), then decrypt at level one; a layer that decrypts to 0 gives 0 (the reference's edge case).
func (s *GPUSecretKey) NestedDecryptBatch(c []*Ciphertext) ([]*gmp.Int, error) {
	for _, ct := range c {
		if ct.Level == EncLevelOne {
			panic("no nested ciphertexts to recover") // paillier.go:362
		}
		if ct.Level != EncLevelTwo {
			panic("not implemented") // paillier.go:371
		}
	}
	layer, err := s.DecryptBatch(c)
	if err != nil {
		return nil, err
	}
	var nz []*Ciphertext
	var at []int
	out := make([]*gmp.Int, len(c))
	for i, v := range layer {
		if v.Cmp(ZeroBigInt) == 0 {
			out[i] = gmp.NewInt(0)
		} else {
			nz, at = append(nz, &Ciphertext{C: v, Level: EncLevelOne, EncMethod: MixedEncryption}), append(at, i)
		}
	}
	if len(nz) > 0 {
		ms, err := s.DecryptBatch(nz)
		if err != nil {
			return nil, err
		}
		for j, i := range at {
			out[i] = ms[j]
		}
	}
	return out, nil
}

// RandomizeBatch: PublicKey.Randomize (operations.go:67-69): ct[i] + a fresh encryption of zero at level one (pk.Encrypt).
func (k *GPUPublicKey) RandomizeBatch(ct []*Ciphertext) ([]*Ciphertext, error) {
	z, err := k.EncryptZeroBatch(len(ct), EncLevelOne)
	if err != nil {
		return nil, err
	}
	return k.AddBatch(ct, z)
}

// NestedRandomizeBatch: PublicKey.NestedRandomize (operations.go:96-118): returns the randomized ciphertexts and the draws
// a, b (from the library's GetRandomNumberInMultiplicativeGroup path) used for each.
func (k *GPUPublicKey) NestedRandomizeBatch(ct []*Ciphertext) ([]*Ciphertext, []*gmp.Int, []*gmp.Int, error) {
	a, err := randomUnits(k, len(ct))
	if err != nil {
		return nil, nil, nil, err
	}
	b, err := randomUnits(k, len(ct))
	if err != nil {
		return nil, nil, nil, err
	}
	out, err := k.NestedRandomizeWithABBatch(ct, a, b)
	return out, a, b, err
}

// ExtractRandonnessBatch: SecretKey.ExtractRandonness (operations.go:75-91) for every ciphertext (all of one level), composed
// from the batch primitives exactly as the reference composes the scalar ones: v = Decrypt(ct), z = G^(-v) ct mod n^(s+1),
// r = z^(ns^-1 mod lambda) mod n.
func (s *GPUSecretKey) ExtractRandonnessBatch(c []*Ciphertext) ([]*gmp.Int, error) {
	if len(c) == 0 {
		return nil, nil
	}
	sk := s.sk
	_, ns, ns1 := sk.getModuliForLevel(c[0].Level)
	nsInv := new(gmp.Int).ModInverse(ns, sk.Lambda)
	v, err := s.DecryptBatch(c)
	if err != nil {
		return nil, err
	}
	g := s.pub.g
	m1, err := g.NewModulus(ns1)
	if err != nil {
		return nil, err
	}
	defer m1.Close()
	mn, err := g.NewModulus(sk.N)
	if err != nil {
		return nil, err
	}
	defer mn.Close()
	gs := make([]*gmp.Int, len(c))
	for i := range gs {
		gs[i] = sk.G
	}
	gv, err := m1.ExpBatch(gs, v) // G^v mod n^(s+1), one exponent per ciphertext
	if err != nil {
		return nil, err
	}
	gvInv, _, err := m1.InvBatch(gv)
	if err != nil {
		return nil, err
	}
	z, err := m1.MulBatch(gvInv, cvals(c))
	if err != nil {
		return nil, err
	}
	return mn.ExpBatch(z, []*gmp.Int{nsInv}) // Exp(z, nsInv, N): z < n^(s+1) is reduced modulo n by the library
}
