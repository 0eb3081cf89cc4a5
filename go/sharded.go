// sharded.go -- several MI355X of one node behind the batch API of batch.go (SURVEY 8(e); BASELINE.json: "ciphertext batches
// shard embarrassingly across the 8 GPUs of one node").  One GPU context per device, one goroutine per device for the length of
// a call, contiguous slices (the rule of paillier_amd/dist.py shard_slice: sizes differ by at most one), results concatenated in
// order -- so every method returns exactly what the one-device method of the same name returns for the whole batch.
//
// Encrypt / Decrypt / proofs are independent units: no exchange.  Threshold decryption is the one flow with an exchange step
// (thresholdkey.go:149-201; the shape of the reference's own multi-server test, thresholdkey_test.go:413-427, where the servers
// live in one process): the (server, ciphertext) units are sharded over the devices, every device computes its unit range in one
// call (PartialDecryptUnitsBatch), the partials meet in host memory -- this process IS the exchange; between processes it is the
// RCCL all-gather of paillier_amd/dist.py -- and every device combines its own ciphertext slice.
//
// The C ABI underneath is thread-compatible per context (one blocking call at a time per pgpu_ctx, any number of contexts in
// flight): tests/c/test_cabi.c drives two contexts from two pthreads through this same flow against the committed fixtures.
// Like gpu.go / batch.go this file is NOT compiled in this repository's build image (no Go toolchain).
package paillier

import (
	"errors"
	"sync"

	gmp "github.com/ncw/gmp"
)

// ShardedGPU owns one context per device.
type ShardedGPU struct{ gpus []*GPU }

// NewShardedGPU opens the listed HIP devices (e.g. []int{0, 1, 2, 3, 4, 5, 6, 7}); a device may be listed twice (two contexts
// with streams of their own on one device -- what the one-GPU rehearsal of tests/c/test_cabi.c does).
func NewShardedGPU(devices []int) (*ShardedGPU, error) {
	if len(devices) == 0 {
		return nil, errors.New("paillier: no devices")
	}
	s := &ShardedGPU{}
	for _, d := range devices {
		g, err := NewGPUOwnStream(d, 0, 0)
		if err != nil {
			s.Close()
			return nil, err
		}
		s.gpus = append(s.gpus, g)
	}
	return s, nil
}

func (s *ShardedGPU) Close() {
	for _, g := range s.gpus {
		g.Close()
	}
	s.gpus = nil
}

// Devices is the number of contexts.
func (s *ShardedGPU) Devices() int { return len(s.gpus) }

// ShardSlice is the contiguous [begin, end) slice of `total` units for device `rank` of `world`: sizes differ by at most one and
// the slices cover [0, total) in order (paillier_amd/dist.py shard_slice; bench.py shards its ranks by the same rule).
func ShardSlice(total, rank, world int) (int, int) {
	base, rem := total/world, total%world
	extra := rank
	if extra > rem {
		extra = rem
	}
	begin := rank*base + extra
	end := begin + base
	if rank < rem {
		end++
	}
	return begin, end
}

// each runs f(rank, begin, end) for every device with a non-empty slice of `total`, one goroutine per device, and returns the
// first error in rank order.  (The batch methods pin their goroutine to an OS thread themselves: the library's last-error text
// is thread-local.)
func (s *ShardedGPU) each(total int, f func(rank, begin, end int) error) error {
	world := len(s.gpus)
	errs := make([]error, world)
	var wg sync.WaitGroup
	for r := 0; r < world; r++ {
		b, e := ShardSlice(total, r, world)
		if e == b {
			continue
		}
		wg.Add(1)
		go func(r, b, e int) {
			defer wg.Done()
			errs[r] = f(r, b, e)
		}(r, b, e)
	}
	wg.Wait()
	for _, err := range errs {
		if err != nil {
			return err
		}
	}
	return nil
}

// ShardedPublicKey is a PublicKey uploaded to every device (the key is replicated; the batches are what is sharded).
type ShardedPublicKey struct {
	sg   *ShardedGPU
	keys []*GPUPublicKey
}

func (s *ShardedGPU) Upload(pk *PublicKey) (*ShardedPublicKey, error) {
	k := &ShardedPublicKey{sg: s}
	for _, g := range s.gpus {
		h, err := g.Upload(pk)
		if err != nil {
			k.Close()
			return nil, err
		}
		k.keys = append(k.keys, h)
	}
	return k, nil
}

func (k *ShardedPublicKey) Close() {
	for _, h := range k.keys {
		h.Close()
	}
	k.keys = nil
}

// ShardedSecretKey is a SecretKey uploaded to every device.
type ShardedSecretKey struct {
	pub  *ShardedPublicKey
	keys []*GPUSecretKey
}

func (k *ShardedPublicKey) UploadSecret(sk *SecretKey) (*ShardedSecretKey, error) {
	s := &ShardedSecretKey{pub: k}
	for _, h := range k.keys {
		x, err := h.UploadSecret(sk)
		if err != nil {
			s.Close()
			return nil, err
		}
		s.keys = append(s.keys, x)
	}
	return s, nil
}

func (s *ShardedSecretKey) Close() {
	for _, h := range s.keys {
		h.Close()
	}
	s.keys = nil
}

// EncryptWithRBatch: PublicKey.EncryptWithRAtLevel (paillier.go:206-218) for every (m[i], r[i]), sliced over the devices.
func (k *ShardedPublicKey) EncryptWithRBatch(m, r []*gmp.Int, level EncryptionLevel) ([]*Ciphertext, error) {
	if len(m) != len(r) {
		return nil, errors.New("paillier: len(m) != len(r)")
	}
	out := make([]*Ciphertext, len(m))
	err := k.sg.each(len(m), func(rank, b, e int) error {
		c, err := k.keys[rank].EncryptWithRBatch(m[b:e], r[b:e], level)
		if err == nil {
			copy(out[b:e], c)
		}
		return err
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}

// EncryptBatch: PublicKey.EncryptAtLevel (paillier.go:258-269), the library drawing r on every device's host side.
func (k *ShardedPublicKey) EncryptBatch(m []*gmp.Int, level EncryptionLevel) ([]*Ciphertext, error) {
	out := make([]*Ciphertext, len(m))
	err := k.sg.each(len(m), func(rank, b, e int) error {
		c, err := k.keys[rank].EncryptBatch(m[b:e], level)
		if err == nil {
			copy(out[b:e], c)
		}
		return err
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}

// DecryptBatch: SecretKey.Decrypt (paillier.go:292-303) for every ciphertext (all of one level), sliced over the devices --
// the headline path of BASELINE.json at N devices.
func (s *ShardedSecretKey) DecryptBatch(c []*Ciphertext) ([]*gmp.Int, error) {
	out := make([]*gmp.Int, len(c))
	err := s.pub.sg.each(len(c), func(rank, b, e int) error {
		m, err := s.keys[rank].DecryptBatch(c[b:e])
		if err == nil {
			copy(out[b:e], m)
		}
		return err
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}

// AddBatch / ConstMultBatch: operations.go:11-29, 58-64 element-wise, sliced over the devices.
func (k *ShardedPublicKey) AddBatch(a, b []*Ciphertext) ([]*Ciphertext, error) {
	if len(a) != len(b) {
		return nil, errors.New("paillier: operand vectors differ in length")
	}
	out := make([]*Ciphertext, len(a))
	err := k.sg.each(len(a), func(rank, lo, hi int) error {
		c, err := k.keys[rank].AddBatch(a[lo:hi], b[lo:hi])
		if err == nil {
			copy(out[lo:hi], c)
		}
		return err
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}

func (k *ShardedPublicKey) ConstMultBatch(c []*Ciphertext, ks []*gmp.Int) ([]*Ciphertext, error) {
	if len(ks) != 1 && len(ks) != len(c) {
		return nil, errors.New("paillier: one shared k or one k per ciphertext")
	}
	out := make([]*Ciphertext, len(c))
	err := k.sg.each(len(c), func(rank, lo, hi int) error {
		kk := ks
		if len(ks) != 1 {
			kk = ks[lo:hi]
		}
		r, err := k.keys[rank].ConstMultBatch(c[lo:hi], kk)
		if err == nil {
			copy(out[lo:hi], r)
		}
		return err
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}

// ProveDDLEQProofBatch: SecretKey.ProveDDLEQ (ddleq.go:27-40) for a batch of statements, the statements split over the devices
// (BASELINE config 5; proofs of different statements are independent: no exchange).
func (s *ShardedSecretKey) ProveDDLEQProofBatch(secpar int, ct1, ct2 []*Ciphertext, a, b []*gmp.Int) ([]*DDLEQProof, error) {
	n := len(ct1)
	if len(ct2) != n || len(a) != n || len(b) != n || secpar < 1 {
		return nil, errors.New("paillier: one (ct1, ct2, a, b) per statement and secpar >= 1")
	}
	out := make([]*DDLEQProof, n)
	err := s.pub.sg.each(n, func(rank, lo, hi int) error {
		p, err := s.keys[rank].ProveDDLEQProofBatch(secpar, ct1[lo:hi], ct2[lo:hi], a[lo:hi], b[lo:hi])
		if err == nil {
			copy(out[lo:hi], p)
		}
		return err
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}

// VerifyDDLEQProofBatch: PublicKey.VerifyDDLEQProof (ddleq.go:44-53) per statement, the statements split over the devices.
func (k *ShardedPublicKey) VerifyDDLEQProofBatch(ct1, ct2 []*Ciphertext, proofs []*DDLEQProof) ([]bool, error) {
	n := len(ct1)
	if len(ct2) != n || len(proofs) != n {
		return nil, errors.New("paillier: one proof per statement")
	}
	out := make([]bool, n)
	err := k.sg.each(n, func(rank, lo, hi int) error {
		ok, err := k.keys[rank].VerifyDDLEQProofBatch(ct1[lo:hi], ct2[lo:hi], proofs[lo:hi])
		if err == nil {
			copy(out[lo:hi], ok)
		}
		return err
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}

// ThresholdDecryptBatch: every ciphertext decrypted by the servers tsks (thresholdkey.go:192-201 PartialDecrypt per server, then
// :149-161 CombinePartialDecryptions), BASELINE config 4 on the devices of this process.
//
//  1. unit u = s*len(c) + i is server tsks[s] on ciphertext c[i]; device r computes the unit range ShardSlice(len(tsks)*len(c), r,
//     world) in ONE call (PartialDecryptUnitsBatch: a range that wants a ciphertext under two shares walks one chain of
//     squarings; only the shares of the servers a range touches are read by that device's call);
//  2. the partials meet in host memory, server-major (the exchange: 512 bytes per unit at 2048 bits);
//  3. device r combines the ciphertext slice ShardSlice(len(c), r, world) (CombinePartialDecryptionsBatch).
//
// Errors are the reference's ("Threshold not meet", "two shares has been created by the same server", thresholdkey.go:77-89).
func (k *ShardedPublicKey) ThresholdDecryptBatch(tk *ThresholdPublicKey, tsks []*ThresholdSecretKey, c []*gmp.Int) ([]*gmp.Int, error) {
	S, B := len(tsks), len(c)
	if S == 0 || B == 0 {
		return nil, errors.New("paillier: no shares or no ciphertexts")
	}
	parts := make([]*PartialDecryption, S*B)
	err := k.sg.each(S*B, func(rank, ub, ue int) error {
		p, err := k.keys[rank].PartialDecryptUnitsBatch(tsks, c, ub, ue)
		if err == nil {
			copy(parts[ub:ue], p)
		}
		return err
	})
	if err != nil {
		return nil, err
	}
	out := make([]*gmp.Int, B)
	err = k.sg.each(B, func(rank, lo, hi int) error {
		cols := make([][]*PartialDecryption, S)
		for s := 0; s < S; s++ {
			cols[s] = parts[s*B+lo : s*B+hi]
		}
		m, err := k.keys[rank].CombinePartialDecryptionsBatch(tk, cols)
		if err == nil {
			copy(out[lo:hi], m)
		}
		return err
	})
	if err != nil {
		return nil, err
	}
	return out, nil
}
