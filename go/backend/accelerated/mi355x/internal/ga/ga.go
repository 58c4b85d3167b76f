//go:build mi355x

// Package ga is the cgo binding of libgnark_amd.so (include/gnark_amd.h): the only place of the mi355x backend that
// imports "C".  Every wrapper passes flat pointers to pointer-free memory (fr/fp element arrays, []bool images) that the
// library has finished with when the call returns, so the default cgocheck is satisfied without pinning; the two
// struct-of-pointers entry points of the ABI (ga_g16_key, ga_plonk_quotient_in) are filled in C memory under a
// runtime.Pinner (see PlonkQuotient).
//
// Replaces the ~20 ICICLE wrappers imported by backend/accelerated/icicle/groth16/bn254/icicle.go:38-46
// (icicle_core, icicle_msm, icicle_g2, icicle_ntt, icicle_vecops, icicle_runtime).
//
// Build: the header is expected at <repo>/include and the library on the linker path, e.g.
//
//	CGO_LDFLAGS="-L/opt/gnark_amd/lib -Wl,-rpath,/opt/gnark_amd/lib" go build -tags=mi355x ./...
package ga

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../../../include
#cgo LDFLAGS: -lgnark_amd
#include <stdlib.h>
#include "gnark_amd.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"sync"
	"unsafe"
)

// Curve ids of the ABI (ecc.ID analogue; only the curves libgnark_amd is built for).
type Curve int

const (
	BN254     Curve = C.GA_BN254
	BLS12_381 Curve = C.GA_BLS12_381
)

// Group ids.
const (
	G1 = C.GA_G1
	G2 = C.GA_G2
)

// MSM flags.
const (
	BasesOnDevice     = uint(C.GA_BASES_ON_DEVICE)
	ScalarsOnDevice   = uint(C.GA_SCALARS_ON_DEVICE)
	ScalarsMontgomery = uint(C.GA_SCALARS_MONTGOMERY)
)

// Vector / point selectors of the staged key builder.
const (
	KeyG1A = C.GA_KEY_G1_A
	KeyG1B = C.GA_KEY_G1_B
	KeyG1Z = C.GA_KEY_G1_Z
	KeyG1K = C.GA_KEY_G1_K
	KeyG2B = C.GA_KEY_G2_B

	KeyG1Alpha = C.GA_KEY_G1_ALPHA
	KeyG1Beta  = C.GA_KEY_G1_BETA
	KeyG1Delta = C.GA_KEY_G1_DELTA
	KeyG2Beta  = C.GA_KEY_G2_BETA
	KeyG2Delta = C.GA_KEY_G2_DELTA
)

// Precompute policy of a pinned key (ga_g16_key.precompute).
const (
	PrecomputeAuto   int32 = 0
	PrecomputeAlways int32 = 1
	PrecomputeNever  int32 = -1
)

func status(rc C.int, what string) error {
	if rc == C.GA_OK {
		return nil
	}
	return fmt.Errorf("gnark_amd: %s: %s (code %d)", what, C.GoString(C.ga_last_error()), int(rc))
}

// call runs f on a locked OS thread so that ga_last_error() (thread-local in the library) is read on the thread that
// produced it.  The library selects its device in every entry point, so no thread affinity is needed beyond that.
func call(what string, f func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	return status(f(), what)
}

// Version returns ga_version().
func Version() string { return C.GoString(C.ga_version()) }

// DeviceCount returns the number of HIP devices visible to the process.
func DeviceCount() (int, error) {
	var n C.int
	if err := call("ga_device_count", func() C.int { return C.ga_device_count(&n) }); err != nil {
		return 0, err
	}
	return int(n), nil
}

// Context is one (process, device) pair: streams, scratch and the per-device prove mutex live behind it
// (icicle.go:77-86 keeps that mutex on the Go side; here it is inside the library).
type Context struct {
	h      *C.ga_ctx
	Device int
}

var (
	ctxMu sync.Mutex
	ctxs  = map[int]*Context{}
)

// ContextFor returns the process-wide context of a device, creating it on first use (the warm-up of
// groth16_icicle.go:33-72 collapses into this).
func ContextFor(device int) (*Context, error) {
	ctxMu.Lock()
	defer ctxMu.Unlock()
	if c, ok := ctxs[device]; ok {
		return c, nil
	}
	c := &Context{Device: device}
	if err := call("ga_ctx_create", func() C.int { return C.ga_ctx_create(C.int(device), &c.h) }); err != nil {
		return nil, err
	}
	ctxs[device] = c
	return c, nil
}

// MemInfo returns the device name and its total / free bytes (runtime.GetAvailableMemory, icicle.go:475).
func (c *Context) MemInfo() (name string, total, free uint64, err error) {
	buf := (*C.char)(C.malloc(256))
	defer C.free(unsafe.Pointer(buf))
	var t, f C.uint64_t
	err = call("ga_device_info", func() C.int { return C.ga_device_info(c.h, buf, 256, &t, &f) })
	return C.GoString(buf), uint64(t), uint64(f), err
}

// SetProfiling switches the per-stage hipEvent timers on or off (ICICLE_STEP_PROFILE, icicle.go:72-75).
func (c *Context) SetProfiling(on bool) error {
	v := C.int(0)
	if on {
		v = 1
	}
	return call("ga_profile_enable", func() C.int { return C.ga_profile_enable(c.h, v) })
}

// ReadProfile returns "stage=ms;..." for the stages recorded since the last reset, and resets.
func (c *Context) ReadProfile() (string, error) {
	const size = 1 << 20
	buf := (*C.char)(C.malloc(size))
	defer C.free(unsafe.Pointer(buf))
	if err := call("ga_profile_read", func() C.int { return C.ga_profile_read(c.h, buf, size) }); err != nil {
		return "", err
	}
	s := C.GoString(buf)
	return s, call("ga_profile_reset", func() C.int { return C.ga_profile_reset(c.h) })
}

// ---- Groth16 key -------------------------------------------------------------------------------------------------

// KeyBuilder stages a proving key on the device vector by vector (ga_g16_builder_*): the replacement of loadG1 /
// loadG1Raw / loadG2 (icicle.go:319-359).  Every method hands ONE Go pointer to C for the duration of the call.
type KeyBuilder struct {
	h *C.ga_g16_builder
}

// NewKeyBuilder starts a key for `curve` with the given domain cardinality and wire count; (shardIndex, shardCount)
// select the base-point range this device keeps (0, 1 = the whole key).
func (c *Context) NewKeyBuilder(curve Curve, domainCardinality, nbWires uint64, shardIndex, shardCount int) (*KeyBuilder, error) {
	b := &KeyBuilder{}
	err := call("ga_g16_builder_create", func() C.int {
		return C.ga_g16_builder_create(c.h, C.int(curve), C.uint64_t(domainCardinality), C.uint64_t(nbWires), C.uint32_t(shardIndex),
			C.uint32_t(shardCount), &b.h)
	})
	if err != nil {
		return nil, err
	}
	return b, nil
}

// Vector uploads one base vector (points: pointer to the first element of a []G1Affine / []G2Affine, n its length) in
// chunks, so that a sharded builder never receives more than it keeps plus one chunk.
func (b *KeyBuilder) Vector(which int, points unsafe.Pointer, n uint64, pointBytes uintptr) error {
	if err := call("ga_g16_builder_reserve", func() C.int { return C.ga_g16_builder_reserve(b.h, C.int(which), C.uint64_t(n)) }); err != nil {
		return err
	}
	const chunk = uint64(1) << 22
	for lo := uint64(0); lo < n; lo += chunk {
		cnt := min(chunk, n-lo)
		p := unsafe.Add(points, uintptr(lo)*pointBytes)
		if err := call("ga_g16_builder_append", func() C.int { return C.ga_g16_builder_append(b.h, C.int(which), p, C.uint64_t(cnt)) }); err != nil {
			return err
		}
	}
	return nil
}

// Point sets one of alpha1, beta1, delta1 (G1Affine) or beta2, delta2 (G2Affine).
func (b *KeyBuilder) Point(which int, affine unsafe.Pointer) error {
	return call("ga_g16_builder_set_point", func() C.int { return C.ga_g16_builder_set_point(b.h, C.int(which), affine) })
}

// Infinity sets pk.InfinityA (which = 0) or pk.InfinityB (which = 1); a Go bool is one byte holding 0 or 1.
func (b *KeyBuilder) Infinity(which int, mask []bool) error {
	if len(mask) == 0 {
		return errors.New("gnark_amd: empty infinity mask")
	}
	p := (*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(mask)))
	return call("ga_g16_builder_set_infinity", func() C.int { return C.ga_g16_builder_set_infinity(b.h, C.int(which), p, C.uint64_t(len(mask))) })
}

// CommitmentKey pins pk.CommitmentKeys[i].Basis / BasisExpSigma (setup.go:276-287; icicle.go:231-261).
func (b *KeyBuilder) CommitmentKey(basis, basisExpSigma unsafe.Pointer, n uint64) error {
	return call("ga_g16_builder_add_commitment_key", func() C.int {
		return C.ga_g16_builder_add_commitment_key(b.h, basis, basisExpSigma, C.uint64_t(n))
	})
}

// KRemove gives the sorted wire ids left out of the K MSM (prove.go:231-235).
func (b *KeyBuilder) KRemove(ids []uint64) error {
	if len(ids) == 0 {
		return nil
	}
	p := (*C.uint64_t)(unsafe.Pointer(unsafe.SliceData(ids)))
	return call("ga_g16_builder_set_k_remove", func() C.int { return C.ga_g16_builder_set_k_remove(b.h, p, C.uint64_t(len(ids))) })
}

// Finish builds the window tables / gather lists and returns the device key; the builder is consumed either way.
func (b *KeyBuilder) Finish(precompute int32) (*ProvingKey, error) {
	pk := &ProvingKey{}
	h := b.h
	b.h = nil
	if err := call("ga_g16_builder_finish", func() C.int { return C.ga_g16_builder_finish(h, C.int32_t(precompute), &pk.h) }); err != nil {
		return nil, err
	}
	return pk, nil
}

// Abandon releases a builder that will not be finished.
func (b *KeyBuilder) Abandon() {
	if b.h != nil {
		C.ga_g16_builder_destroy(b.h)
		b.h = nil
	}
}

// ProvingKey is a device-resident Groth16 proving key (or one shard of it).
type ProvingKey struct {
	h *C.ga_g16_pk
}

// ReadKeyFd loads a key file -- ProvingKey.WriteTo, WriteRawTo or WriteDump output, recognised from the stream -- from an
// open file descriptor straight into HBM (ga_g16_pk_read_fd): the file streams through pinned staging buffers and the
// points are decoded by a device kernel, so the 6-9 GiB key never exists as Go slices.  kRemove is the toRemove list of
// prove.go:231-235 (it comes from the constraint system, not from the key file).
func (c *Context) ReadKeyFd(curve Curve, fd uintptr, precompute int32, shardIndex, shardCount int, kRemove []uint64) (*ProvingKey, uint64, error) {
	pk := &ProvingKey{}
	var used C.uint64_t
	var rem *C.uint64_t
	if len(kRemove) > 0 {
		rem = (*C.uint64_t)(unsafe.Pointer(unsafe.SliceData(kRemove)))
	}
	err := call("ga_g16_pk_read_fd", func() C.int {
		return C.ga_g16_pk_read_fd(c.h, C.int(curve), C.int(fd), C.int32_t(precompute), C.uint32_t(shardIndex), C.uint32_t(shardCount),
			rem, C.uint64_t(len(kRemove)), &pk.h, &used)
	})
	if err != nil {
		return nil, 0, err
	}
	return pk, uint64(used), nil
}

// ParseProof is Proof.ReadFrom on WriteTo / WriteRawTo bytes (ga_g16_proof_unmarshal); proofOut: Ar | Bs | Krs affine.
func ParseProof(curve Curve, data []byte, proofOut, commitmentsOut unsafe.Pointer, maxCommitments int, pokOut unsafe.Pointer) (nbCommitments int, consumed int, err error) {
	var n C.uint32_t
	var used C.size_t
	p := (*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(data)))
	err = call("ga_g16_proof_unmarshal", func() C.int {
		return C.ga_g16_proof_unmarshal(C.int(curve), p, C.size_t(len(data)), proofOut, commitmentsOut, C.uint32_t(maxCommitments), &n, pokOut, &used)
	})
	return int(n), int(used), err
}

// Prove runs computeH, the five MSMs and the (r, s) epilogue: w, a, b, c point to the solver's W, A, B, C;
// out receives Ar | Bs | Krs (G1Affine, G2Affine, G1Affine).
func (pk *ProvingKey) Prove(w, a, b, c unsafe.Pointer, nbConstraints, nbPublic uint64, r, s, out unsafe.Pointer) error {
	return call("ga_g16_prove", func() C.int {
		return C.ga_g16_prove(pk.h, w, a, b, c, C.uint64_t(nbConstraints), C.uint64_t(nbPublic), r, s, out)
	})
}

// OneShotKey names the host images of a proving key for ProveOneShot: pointers to the first elements of pk.G1.A, B, Z, K,
// pk.G2.B ([]G1Affine / []G2Affine), to the five single points, and the infinity masks and K filter of the key.
type OneShotKey struct {
	Curve                       Curve
	DomainCardinality           uint64
	A, B, Z, K, B2              unsafe.Pointer
	LenA, LenB, LenZ, LenK, LenB2 uint64
	Alpha1, Beta1, Delta1       unsafe.Pointer
	Beta2, Delta2               unsafe.Pointer
	InfinityA, InfinityB        []bool
	NbInfinityA, NbInfinityB    uint64
	KRemove                     []uint64
}

// ProveOneShot is one proof on a key that is NOT kept on the device (PinToGPU false, the default here as in
// icicle.go:797-805): ga_g16_prove_oneshot uploads the key as plain vectors while the proof already runs -- every MSM waits
// for its own vector only -- and frees it before it returns.  The ga_g16_key lives in C memory; the Go slices it points to
// are pinned for the duration of the call (runtime.Pinner), nothing is retained afterwards.
func (c *Context) ProveOneShot(key *OneShotKey, w, a, b, cc unsafe.Pointer, nbConstraints, nbPublic uint64, r, s, out unsafe.Pointer) error {
	if len(key.InfinityA) == 0 || len(key.InfinityA) != len(key.InfinityB) {
		return errors.New("gnark_amd: infinity masks must have nbWires entries each")
	}
	var p runtime.Pinner
	defer p.Unpin()
	pin := func(x unsafe.Pointer) unsafe.Pointer {
		if x != nil {
			p.Pin(x)
		}
		return x
	}
	k := (*C.ga_g16_key)(C.calloc(1, C.size_t(unsafe.Sizeof(C.ga_g16_key{}))))
	defer C.free(unsafe.Pointer(k))
	k.curve = C.int(key.Curve)
	k.domain_cardinality = C.uint64_t(key.DomainCardinality)
	k.g1_alpha, k.g1_beta, k.g1_delta = pin(key.Alpha1), pin(key.Beta1), pin(key.Delta1)
	k.g2_beta, k.g2_delta = pin(key.Beta2), pin(key.Delta2)
	k.g1_a, k.len_a = pin(key.A), C.uint64_t(key.LenA)
	k.g1_b, k.len_b = pin(key.B), C.uint64_t(key.LenB)
	k.g1_z, k.len_z = pin(key.Z), C.uint64_t(key.LenZ)
	k.g1_k, k.len_k = pin(key.K), C.uint64_t(key.LenK)
	k.g2_b, k.len_b2 = pin(key.B2), C.uint64_t(key.LenB2)
	k.infinity_a = (*C.uint8_t)(pin(unsafe.Pointer(unsafe.SliceData(key.InfinityA))))
	k.infinity_b = (*C.uint8_t)(pin(unsafe.Pointer(unsafe.SliceData(key.InfinityB))))
	k.nb_wires = C.uint64_t(len(key.InfinityA))
	k.nb_infinity_a, k.nb_infinity_b = C.uint64_t(key.NbInfinityA), C.uint64_t(key.NbInfinityB)
	k.precompute = -1
	k.shard_count = 1
	if len(key.KRemove) > 0 {
		k.k_remove = (*C.uint64_t)(pin(unsafe.Pointer(unsafe.SliceData(key.KRemove))))
		k.len_k_remove = C.uint64_t(len(key.KRemove))
	}
	return call("ga_g16_prove_oneshot", func() C.int {
		return C.ga_g16_prove_oneshot(c.h, k, w, a, b, cc, C.uint64_t(nbConstraints), C.uint64_t(nbPublic), r, s, out)
	})
}

// ProveMulti proves ONE statement over several devices: keys[i] is shard i of len(keys) of the same proving key, each on
// its own device (ga_g16_prove_multi: one host thread per device inside the library, partial sums added on the host).
func ProveMulti(keys []*ProvingKey, w, a, b, c unsafe.Pointer, nbConstraints, nbPublic uint64, r, s, out unsafe.Pointer) error {
	if len(keys) == 0 {
		return errors.New("gnark_amd: no device keys")
	}
	hs := (**C.ga_g16_pk)(C.malloc(C.size_t(len(keys)) * C.size_t(unsafe.Sizeof(uintptr(0))))) // C handles in C memory
	defer C.free(unsafe.Pointer(hs))
	arr := unsafe.Slice(hs, len(keys))
	for i, k := range keys {
		arr[i] = k.h
	}
	return call("ga_g16_prove_multi", func() C.int {
		return C.ga_g16_prove_multi(hs, C.uint32_t(len(keys)), w, a, b, c, C.uint64_t(nbConstraints), C.uint64_t(nbPublic), r, s, out)
	})
}

// Commit runs pk.CommitmentKeys[i].Commit and ProveKnowledge on the pinned bases (prove.go:84,114) with one upload of the
// committed values.
func (pk *ProvingKey) Commit(i int, values unsafe.Pointer, n uint64, commitmentOut, pokOut unsafe.Pointer) error {
	return call("ga_g16_commit", func() C.int {
		return C.ga_g16_commit(pk.h, C.uint32_t(i), values, C.uint64_t(n), commitmentOut, pokOut)
	})
}

// Free releases the device memory of the key (FreeGPUResources, icicle.go:1493-1549); idempotent.
func (pk *ProvingKey) Free() {
	if pk != nil && pk.h != nil {
		C.ga_g16_pk_destroy(pk.h)
		pk.h = nil
	}
}

// ---- MSM / NTT (PLONK) -------------------------------------------------------------------------------------------
//
// Every MSM entry point returns A Jacobian representative of the sum, not a canonical one: the order in which a bucket's points
// are added comes from atomics in the fused sort, so X, Y, Z of two calls on the same inputs may differ by a common scaling while
// the point is the same (include/gnark_amd.h, ga_msm).  gnark-crypto's MultiExp is deterministic for a fixed NbTasks; code on this
// side must therefore never compare, hash or serialise the Jacobian limbs of a result -- convert with FromJacobian (or compare with
// G1Jac.Equal / G2Jac.Equal) first, as the provers in ../../groth16 and ../../plonk do.  tests/test_abi_surface.py checks the shim
// for `==` / `!=` on Jacobian values.

// MSM computes sum scalars[i]*bases[i] into a Jacobian point (G1Jac / G2Jac image).
func (c *Context) MSM(curve Curve, group int, bases, scalars unsafe.Pointer, n uint64, flags uint, outJac unsafe.Pointer) error {
	return call("ga_msm", func() C.int {
		return C.ga_msm(c.h, C.int(curve), C.int(group), bases, scalars, C.size_t(n), C.uint(flags), outJac)
	})
}

// Table is a pinned base vector with its window multiples (ga_msm_table_*): the KZG SRS of a PLONK key.
type Table struct {
	h *C.ga_msm_table
	N uint64
}

// NewTable pins n bases.  batched: the table will mostly serve RunBatch (the KZG SRS of a PLONK key, whose commitments come in
// groups of three): GA_TABLE_BATCHED plans a narrower window.
func (c *Context) NewTable(curve Curve, group int, bases unsafe.Pointer, n uint64, batched bool) (*Table, error) {
	t := &Table{N: n}
	flags := C.uint(0)
	if batched {
		flags = C.GA_TABLE_BATCHED
	}
	err := call("ga_msm_table_create", func() C.int {
		return C.ga_msm_table_create(c.h, C.int(curve), C.int(group), bases, C.size_t(n), flags, &t.h)
	})
	if err != nil {
		return nil, err
	}
	return t, nil
}

// Run multiplies the table's bases by exactly t.N scalars (Montgomery).
func (t *Table) Run(scalars unsafe.Pointer, outJac unsafe.Pointer) error {
	return call("ga_msm_table_run", func() C.int { return C.ga_msm_table_run(t.h, scalars, C.uint(ScalarsMontgomery), outJac) })
}

// RunBatch multiplies the table's bases by k scalar vectors of exactly t.N elements each in ONE pass (ga_msm_table_run_batch):
// outJacs receives k Jacobian points.  The pointer array handed to C lives in C memory for the duration of the call and every
// scalar vector is pinned (cgo: no Go pointer to unpinned Go memory is stored anywhere C can see).
func (t *Table) RunBatch(scalars []unsafe.Pointer, outJacs unsafe.Pointer) error {
	k := len(scalars)
	if k == 0 || k > 16 {
		return fmt.Errorf("ga: batch of %d scalar vectors outside [1, 16]", k)
	}
	arr := (*[16]unsafe.Pointer)(C.malloc(C.size_t(k) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(arr))
	var pin runtime.Pinner
	defer pin.Unpin()
	for i, p := range scalars {
		pin.Pin(p)
		arr[i] = p
	}
	return call("ga_msm_table_run_batch", func() C.int {
		return C.ga_msm_table_run_batch(t.h, (*unsafe.Pointer)(unsafe.Pointer(arr)), C.uint32_t(k), C.uint(ScalarsMontgomery), outJacs)
	})
}

// Open is kzg.Open over the pinned SRS: claimed value and the Jacobian commitment to the quotient.
func (t *Table) Open(poly unsafe.Pointer, n uint64, point, claimedOut, hOutJac unsafe.Pointer) error {
	return call("ga_kzg_open", func() C.int {
		return C.ga_kzg_open(t.h, poly, C.size_t(n), C.uint(ScalarsMontgomery), point, claimedOut, hOutJac)
	})
}

// Free releases the table.
func (t *Table) Free() {
	if t != nil && t.h != nil {
		C.ga_msm_table_destroy(t.h)
		t.h = nil
	}
}

// LinearCombination is out[i] = sum_j scalars[j] * vecs[j][i] over fr for k <= 16 host vectors of n elements each
// (ga_fr_linear_combination): the fold of kzg.BatchOpenSinglePoint.  The pointer array lives in C memory, the vectors are pinned.
func (c *Context) LinearCombination(curve Curve, n uint64, vecs []unsafe.Pointer, scalars, out unsafe.Pointer) error {
	k := len(vecs)
	if k == 0 || k > 16 {
		return fmt.Errorf("ga: linear combination of %d vectors outside [1, 16]", k)
	}
	arr := (*[16]unsafe.Pointer)(C.malloc(C.size_t(k) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(arr))
	var pin runtime.Pinner
	defer pin.Unpin()
	for i, p := range vecs {
		pin.Pin(p)
		arr[i] = p
	}
	return call("ga_fr_linear_combination", func() C.int {
		return C.ga_fr_linear_combination(c.h, C.int(curve), C.uint64_t(n), C.int(k), (*unsafe.Pointer)(unsafe.Pointer(arr)), scalars, out, 0)
	})
}

// JacToAffine converts with the library's host arithmetic (the Go callers normally use curve.G1Affine.FromJacobian).
func JacToAffine(curve Curve, group int, jac, affineOut unsafe.Pointer) error {
	return call("ga_jac_to_affine", func() C.int { return C.ga_jac_to_affine(C.int(curve), C.int(group), jac, affineOut) })
}

// Domain is the fft.Domain analogue.
type Domain struct {
	h           *C.ga_domain
	Cardinality uint64
}

// NewDomain creates the twiddles of a power-of-two domain on the device.
func (c *Context) NewDomain(curve Curve, cardinality uint64) (*Domain, error) {
	d := &Domain{Cardinality: cardinality}
	if err := call("ga_domain_create", func() C.int { return C.ga_domain_create(c.h, C.int(curve), C.uint64_t(cardinality), &d.h) }); err != nil {
		return nil, err
	}
	return d, nil
}

// FFT transforms `Cardinality` fr elements in place in host memory: inverse / decimation (0 DIF, 1 DIT) / onCoset as
// fft.Domain.FFT and FFTInverse take them.
func (d *Domain) FFT(data unsafe.Pointer, inverse bool, decimation int, onCoset bool) error {
	dir, cs := C.int(C.GA_FFT_FORWARD), C.int(0)
	if inverse {
		dir = C.GA_FFT_INVERSE
	}
	if onCoset {
		cs = 1
	}
	return call("ga_fft", func() C.int { return C.ga_fft(d.h, data, dir, C.int(decimation), cs, 0) })
}

// BuildZ is iop.BuildRatioCopyConstraint on the device (ga_plonk_build_z).
func (d *Domain) BuildZ(l, r, o unsafe.Pointer, permutation []int64, beta, gamma, zOut unsafe.Pointer) error {
	p := (*C.int64_t)(unsafe.Pointer(unsafe.SliceData(permutation)))
	return call("ga_plonk_build_z", func() C.int { return C.ga_plonk_build_z(d.h, l, r, o, p, beta, gamma, 0, zOut) })
}

// Free releases the domain.
func (d *Domain) Free() {
	if d != nil && d.h != nil {
		C.ga_domain_destroy(d.h)
		d.h = nil
	}
}

// QuotientInput mirrors ga_plonk_quotient_in with Go-side types; every pointer addresses n fr elements of pointer-free
// memory.  The struct handed to C is built in C memory and the Go arrays are pinned for the duration of the call.
type QuotientInput struct {
	L, R, O, Z, Ql, Qr, Qm, Qo, Qk, S1, S2, S3 unsafe.Pointer
	Qcp, Pi2                                   []unsafe.Pointer
	LagrangeMask                               uint64
	Bl, Br, Bo, Bz                             unsafe.Pointer
	Alpha, Beta, Gamma                         unsafe.Pointer
}

func (in *QuotientInput) toC(p *runtime.Pinner) (*C.ga_plonk_quotient_in, func()) {
	q := (*C.ga_plonk_quotient_in)(C.calloc(1, C.size_t(unsafe.Sizeof(C.ga_plonk_quotient_in{}))))
	pin := func(x unsafe.Pointer) unsafe.Pointer {
		if x != nil {
			p.Pin(x)
		}
		return x
	}
	q.l, q.r, q.o, q.z = pin(in.L), pin(in.R), pin(in.O), pin(in.Z)
	q.ql, q.qr, q.qm, q.qo, q.qk = pin(in.Ql), pin(in.Qr), pin(in.Qm), pin(in.Qo), pin(in.Qk)
	q.s1, q.s2, q.s3 = pin(in.S1), pin(in.S2), pin(in.S3)
	q.bl, q.br, q.bo, q.bz = pin(in.Bl), pin(in.Br), pin(in.Bo), pin(in.Bz)
	q.alpha, q.beta, q.gamma = pin(in.Alpha), pin(in.Beta), pin(in.Gamma)
	q.lagrange_mask = C.uint64_t(in.LagrangeMask)
	q.nb_bsb = C.uint32_t(len(in.Qcp))
	var frees []unsafe.Pointer
	ptrArray := func(src []unsafe.Pointer) *unsafe.Pointer {
		if len(src) == 0 {
			return nil
		}
		a := (*unsafe.Pointer)(C.malloc(C.size_t(len(src)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
		frees = append(frees, unsafe.Pointer(a))
		dst := unsafe.Slice(a, len(src))
		for i, x := range src {
			dst[i] = pin(x)
		}
		return a
	}
	q.qcp = ptrArray(in.Qcp)
	q.pi2 = ptrArray(in.Pi2)
	return q, func() {
		for _, f := range frees {
			C.free(f)
		}
		C.free(unsafe.Pointer(q))
	}
}

// PlonkQuotient is computeNumerator + divideByZH (backend/plonk/bn254/prove.go:841-1123,1287-1350) in one call.
func PlonkQuotient(domain0, domain1 *Domain, in *QuotientInput, hOut unsafe.Pointer) error {
	var p runtime.Pinner
	defer p.Unpin()
	q, free := in.toC(&p)
	defer free()
	return call("ga_plonk_quotient", func() C.int { return C.ga_plonk_quotient(domain0.h, domain1.h, q, hOut) })
}

// PlonkKey holds the coset evaluations of the circuit-constant polynomials (ga_plonk_pk_create).
type PlonkKey struct {
	h *C.ga_plonk_pk
}

// NewPlonkKey pins Ql, Qr, Qm, Qo, S1, S2, S3 and every Qcp of `in` on all cosets.
func NewPlonkKey(domain0, domain1 *Domain, in *QuotientInput) (*PlonkKey, error) {
	var p runtime.Pinner
	defer p.Unpin()
	q, free := in.toC(&p)
	defer free()
	k := &PlonkKey{}
	if err := call("ga_plonk_pk_create", func() C.int { return C.ga_plonk_pk_create(domain0.h, domain1.h, q, &k.h) }); err != nil {
		return nil, err
	}
	return k, nil
}

// Quotient computes the quotient from the per-proof polynomials L, R, O, Z, Qk, Pi2 of `in`.
func (k *PlonkKey) Quotient(in *QuotientInput, hOut unsafe.Pointer) error {
	var p runtime.Pinner
	defer p.Unpin()
	q, free := in.toC(&p)
	defer free()
	return call("ga_plonk_quotient_pinned", func() C.int { return C.ga_plonk_quotient_pinned(k.h, q, hOut) })
}

// Free releases the pinned evaluations.
func (k *PlonkKey) Free() {
	if k != nil && k.h != nil {
		C.ga_plonk_pk_destroy(k.h)
		k.h = nil
	}
}
