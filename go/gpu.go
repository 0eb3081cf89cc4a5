// gpu.go -- cgo binding of libpaillier_hip.so (include/paillier_hip.h) for package paillier.
//
// Drop this file and batch.go next to paillier.go / operations.go / thresholdkey.go / ddleq.go of
// sachaservan/paillier: they ADD batch variants of the exported methods and leave every existing type and signature
// untouched (the scalar methods keep running on github.com/ncw/gmp).  Build with cgo:
//
//	CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/paillier_amd -lpaillier_hip" go build
//
// NOT compiled in the build image of this repository (no Go toolchain, ncw/gmp not vendored); the C entry points are
// exercised, with the same calling convention -- caller-owned flat big-endian buffers, int status -- by
// tests/c/test_cabi.c, tests/cpp/test_host_mirror.cpp and the ctypes mirror paillier_amd/api.py.
//
// Operand format at the boundary: what gmp.Int.Bytes() gives, left-padded with zeros to a fixed stride, element-major.
// cgo rules honoured here:
//   - a Go slice is passed to C only as a DIRECT argument of one blocking call (cgo pins it for that call) and C keeps no
//     Go pointer afterwards;
//   - no Go pointer is ever STORED in C memory: the entry points that take arrays of buffers (pgpu_add_many, pgpu_sub_many,
//     pgpu_partial_decrypt_multi, pgpu_partial_decrypt_indexed, pgpu_partial_decrypt_units, pgpu_combine_partial_decryptions, pgpu_random_oracle_digest)
//     get C.malloc'd pointer arrays whose entries point at C.malloc'd copies of the operands (cbufs below), and their
//     outputs are copied back with C.GoBytes;
//   - pgpu_last_error() is thread-local in the library: every method that makes a C call pins its goroutine to one OS
//     thread (runtime.LockOSThread) from the call to the reading of the message.
package paillier

/*
#cgo LDFLAGS: -lpaillier_hip
#include <stdint.h>
#include <stdlib.h>
#include "paillier_hip.h"

// PGPU_STREAM_NEW is a cast of -1 to a pointer: made here, in C, so that Go never forms that pointer value itself
static void* pgpu_stream_new(void) { return PGPU_STREAM_NEW; }
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"

	gmp "github.com/ncw/gmp"
)

// GPU is one device context (one HIP stream).  Thread-compatible: one batch call at a time per GPU value; use several
// (NewGPUOwnStream) for concurrent small batches.
type GPU struct{ ctx *C.pgpu_ctx }

// GPUError carries the C ABI's status code (PGPU_ERR_*).
type GPUError struct {
	Code int
	Msg  string
}

func (e *GPUError) Error() string { return fmt.Sprintf("paillier_hip [%d]: %s", e.Code, e.Msg) }

// status turns a C status code into an error.  The caller has pinned its goroutine to an OS thread (pin) before the C call:
// the library's last-error message is thread-local, and a goroutine may otherwise move between two cgo calls.
func status(rc C.int) error {
	if rc == C.PGPU_OK {
		return nil
	}
	return &GPUError{int(rc), C.GoString(C.pgpu_last_error())}
}

// pin locks the calling goroutine to its OS thread and returns the function that undoes it:  defer pin()()
func pin() func() {
	runtime.LockOSThread()
	return runtime.UnlockOSThread
}

// cbufs owns C copies of operand buffers and C output buffers plus a C array of pointers to them: what the entry points
// with `const uint8_t* const*` / `uint8_t* const*` parameters take.  Nothing in it is Go memory.
type cbufs struct {
	ptrs *[1 << 20]*C.uint8_t
	n    int
	lens []int
}

func newCbufs(n int) *cbufs {
	if n >= 1<<20 {
		panic("paillier: too many operand buffers")
	}
	sz := n
	if sz == 0 {
		sz = 1
	}
	return &cbufs{ptrs: (*[1 << 20]*C.uint8_t)(C.calloc(C.size_t(sz), C.size_t(unsafe.Sizeof(uintptr(0))))), n: n, lens: make([]int, n)}
}

// in stores a C copy of b as entry i
func (c *cbufs) in(i int, b []byte) {
	c.ptrs[i] = (*C.uint8_t)(C.CBytes(b))
	c.lens[i] = len(b)
}

// out allocates a zeroed C buffer of n bytes as entry i
func (c *cbufs) out(i, n int) {
	if n == 0 {
		n = 1
	}
	c.ptrs[i] = (*C.uint8_t)(C.calloc(C.size_t(n), 1))
	c.lens[i] = n
}

func (c *cbufs) array() **C.uint8_t { return &c.ptrs[0] }

// bytes copies entry i back into Go memory
func (c *cbufs) bytes(i, n int) []byte { return C.GoBytes(unsafe.Pointer(c.ptrs[i]), C.int(n)) }

func (c *cbufs) free() {
	for i := 0; i < c.n; i++ {
		if c.ptrs[i] != nil {
			C.free(unsafe.Pointer(c.ptrs[i]))
		}
	}
	C.free(unsafe.Pointer(c.ptrs))
}

// NewGPU opens HIP device `device` on the default stream.
func NewGPU(device int) (*GPU, error) {
	defer pin()()
	var ctx *C.pgpu_ctx
	if err := status(C.pgpu_ctx_create(C.int(device), nil, &ctx)); err != nil {
		return nil, err
	}
	return &GPU{ctx}, nil
}

// NewGPUOwnStream opens a context with a stream of its own; part/parts confine it to one slice of the compute units
// (parts == 0: the whole device) so that several contexts run small batches side by side.
func NewGPUOwnStream(device, part, parts int) (*GPU, error) {
	defer pin()()
	var ctx *C.pgpu_ctx
	if err := status(C.pgpu_ctx_create(C.int(device), C.pgpu_stream_new(), &ctx)); err != nil {
		return nil, err
	}
	g := &GPU{ctx}
	if parts > 1 {
		name := C.CString("cu_partition")
		defer C.free(unsafe.Pointer(name))
		if err := status(C.pgpu_ctx_set_flag(ctx, name, C.int(parts<<16|part))); err != nil {
			g.Close()
			return nil, err
		}
	}
	return g, nil
}

func (g *GPU) Close() {
	if g.ctx != nil {
		C.pgpu_ctx_destroy(g.ctx)
		g.ctx = nil
	}
}

// Version is the library's version string (pgpu_version).
func Version() string { return C.GoString(C.pgpu_version()) }

// SetFlag flips a runtime switch of the context (pgpu_ctx_set_flag: "asm", "pair", "side", "lanes_wanted" ...; measurement
// and test switches -- the defaults are the product configuration).
func (g *GPU) SetFlag(name string, value int) error {
	defer pin()()
	cn := C.CString(name)
	defer C.free(unsafe.Pointer(cn))
	return status(C.pgpu_ctx_set_flag(g.ctx, cn, C.int(value)))
}

// Profile of the LAST batch call on this context: HIP-event time of its big-integer VM launches, their number (how many of
// them ran the hand-scheduled assembly kernels) and the 28-bit multiply-adds they executed; Kernel names the dominant one.
type Profile struct {
	VMms         float64
	Launches     int
	AsmLaunch    int
	AllLaunch    int
	MultiplyAdds float64
	Kernel       string
}

func (g *GPU) LastProfile() (Profile, error) {
	defer pin()()
	var ms, mads C.double
	var n C.int
	if err := status(C.pgpu_ctx_last_profile(g.ctx, &ms, &n, &mads)); err != nil {
		return Profile{}, err
	}
	return Profile{float64(ms), int(n), int(C.pgpu_ctx_last_vm_asm(g.ctx)), int(C.pgpu_ctx_last_vm_launches(g.ctx)), float64(mads),
		C.GoString(C.pgpu_ctx_last_kernel(g.ctx))}, nil
}

// ---- packing -------------------------------------------------------------------------------------------------------

func pack(xs []*gmp.Int, stride int) []byte {
	buf := make([]byte, len(xs)*stride)
	for i, x := range xs {
		b := x.Bytes() // minimal big-endian magnitude
		if len(b) > stride {
			panic("paillier: operand wider than its stride")
		}
		copy(buf[(i+1)*stride-len(b):(i+1)*stride], b)
	}
	return buf
}

func unpack(buf []byte, stride int) []*gmp.Int {
	out := make([]*gmp.Int, len(buf)/stride)
	for i := range out {
		out[i] = new(gmp.Int).SetBytes(buf[i*stride : (i+1)*stride])
	}
	return out
}

func maxLen(xs []*gmp.Int, atLeast int) int {
	n := atLeast
	for _, x := range xs {
		if l := len(x.Bytes()); l > n {
			n = l
		}
	}
	return n
}

func p8(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

func bytesOf(x *gmp.Int) []byte {
	b := x.Bytes()
	if len(b) == 0 {
		return []byte{0}
	}
	return b
}

// ---- key handles ---------------------------------------------------------------------------------------------------

// GPUPublicKey is the device-side image of a PublicKey (n^2, n^3 and all Montgomery constants precomputed once).
type GPUPublicKey struct {
	g  *GPU
	pk *PublicKey
	h  *C.pgpu_pubkey
}

// Upload precomputes the key on the device (paillier.go:46-57 fields N, G, H, K).
func (g *GPU) Upload(pk *PublicKey) (*GPUPublicKey, error) {
	defer pin()()
	n, gg := bytesOf(pk.N), bytesOf(pk.G)
	var hb, kb []byte
	if pk.H != nil {
		hb = bytesOf(pk.H)
	}
	if pk.K != nil {
		kb = bytesOf(pk.K)
	}
	var h *C.pgpu_pubkey
	rc := C.pgpu_pubkey_create(g.ctx, p8(n), C.size_t(len(n)), p8(gg), C.size_t(len(gg)), p8(hb), C.size_t(len(hb)), p8(kb),
		C.size_t(len(kb)), &h)
	if err := status(rc); err != nil {
		return nil, err
	}
	return &GPUPublicKey{g, pk, h}, nil
}

func (k *GPUPublicKey) Close() { C.pgpu_pubkey_destroy(k.h); k.h = nil }

func (k *GPUPublicKey) plainBytes(level EncryptionLevel) int {
	return int(C.pgpu_pubkey_plain_bytes(k.h, C.int(level)))
}
func (k *GPUPublicKey) cipherBytes(level EncryptionLevel) int {
	return int(C.pgpu_pubkey_cipher_bytes(k.h, C.int(level)))
}

// GPUSecretKey is the device-side image of a SecretKey: p and q are recovered from (n, lambda = phi(n)) for CRT.
type GPUSecretKey struct {
	pub *GPUPublicKey
	sk  *SecretKey
	h   *C.pgpu_seckey
}

func (k *GPUPublicKey) UploadSecret(sk *SecretKey) (*GPUSecretKey, error) {
	defer pin()()
	l := bytesOf(sk.Lambda)
	var h *C.pgpu_seckey
	if err := status(C.pgpu_seckey_create(k.g.ctx, k.h, p8(l), C.size_t(len(l)), &h)); err != nil {
		return nil, err
	}
	return &GPUSecretKey{k, sk, h}, nil
}

func (s *GPUSecretKey) Close() { C.pgpu_seckey_destroy(s.h); s.h = nil }

// HasCRT reports whether (n, lambda) factored n, i.e. whether Decrypt runs over p^2 and q^2 (pgpu_seckey_has_crt); keys whose
// Lambda is not phi(n) take the reference's formula verbatim.
func (s *GPUSecretKey) HasCRT() bool { return C.pgpu_seckey_has_crt(s.h) != 0 }
