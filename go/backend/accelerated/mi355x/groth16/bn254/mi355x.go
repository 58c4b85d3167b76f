//go:build mi355x

package bn254

import (
	"fmt"
	"math/big"
	"os"
	"slices"
	"time"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc"
	curve "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr/hash_to_field"
	"github.com/consensys/gnark/backend"
	"github.com/consensys/gnark/backend/accelerated/mi355x"
	"github.com/consensys/gnark/backend/accelerated/mi355x/internal/ga"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
	cs "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/constraint/solver"
	fcs "github.com/consensys/gnark/frontend/cs"
	"github.com/consensys/gnark/logger"
)

const curveID = ga.BN254

// sliceData is &s[0] without the panic on an empty slice.
func sliceData[T any](s []T) unsafe.Pointer { return unsafe.Pointer(unsafe.SliceData(s)) }

// kRemoveList is the toRemove list of prove.go:231-233 (private committed wires and commitment wires), sorted and
// without repetitions, as the key builder wants it.
func kRemoveList(info constraint.Groth16Commitments) []uint64 {
	parts := info.GetPrivateCommitted()
	parts = append(parts, info.CommitmentIndexes())
	all := slices.Concat(parts...)
	slices.Sort(all)
	all = slices.Compact(all)
	out := make([]uint64, len(all))
	for i, v := range all {
		out[i] = uint64(v)
	}
	return out
}

// effectivePrecompute is the window-table policy a key is pinned with.  A key that is NOT kept on the device between proofs
// (PinToGPU false, the default -- the ICICLE backend's default too: icicle.go:797-805, provingkey.go:37-42) is uploaded as plain
// vectors unless the caller asked for tables explicitly: building 72 GiB of tables (2.4 s for a 2^24 BN254 key) to save 50 ms
// on the one proof that uses them, and freeing them afterwards, would make every default Prove ~17x slower.
func effectivePrecompute(cfg *mi355x.Config, pinned bool) int32 {
	if !pinned && cfg.Precompute == mi355x.PrecomputeAuto {
		return int32(mi355x.PrecomputeNever)
	}
	return int32(cfg.Precompute)
}

// acquire registers a prover on the key and returns the device copy it proves on, pinning the key first if that has not
// happened yet; release is its counterpart.  Between the two the device copy cannot be freed: FreeGPUResources from another
// goroutine (or the un-pinned key's own free-after-proof) is deferred to the LAST prover's release.  Two goroutines may
// therefore prove on one key at the same time (the library runs the second on its own pair of lanes), as the reference's
// accelerated backend allows under its per-device mutex (icicle.go:821-823).
func (pk *ProvingKey) acquire(cfg *mi355x.Config, info constraint.Groth16Commitments) (*deviceInfo, error) {
	pk.setupMu.Lock()
	defer pk.setupMu.Unlock()
	// the sticky pin flag is committed only once the key really is set up under it: a caller that asks for a pinned key while
	// another goroutine is proving on the un-pinned copy gets the "in use" error below and must leave pk.PinToGPU as it was, or the
	// first prover's release() would no longer free its un-pinned device copy
	pin := pk.PinToGPU || cfg.PinToGPU
	if !(pk.deviceInfo != nil && len(pk.InfinityA) == 0) { // a key pinned by PinFromFile has no host copy to (re)pin from
		if err := pk.setupDevicePointersLocked(cfg, info, pin); err != nil {
			return nil, err
		}
	}
	pk.PinToGPU = pin
	pk.users++
	return pk.deviceInfo, nil
}

func (pk *ProvingKey) release() {
	pk.setupMu.Lock()
	defer pk.setupMu.Unlock()
	pk.users--
	if pk.users == 0 && (!pk.PinToGPU || pk.freePending) {
		pk.freeLocked()
		pk.freePending = false
	}
}

// setupDevicePointersLocked pins the key on the configured devices if that has not happened yet (or happened for another
// device set or table policy).  Counterpart of (*ProvingKey).setupDevicePointers, icicle.go:88-264: the Den vector, the coset
// generator and the NTT domain bookkeeping have no Go-side remains -- the library derives them from the cardinality.
// Caller holds setupMu.
func (pk *ProvingKey) setupDevicePointersLocked(cfg *mi355x.Config, info constraint.Groth16Commitments, pin bool) error {
	devices := cfg.DeviceIDs()
	precompute := effectivePrecompute(cfg, pin)
	if pk.deviceInfo != nil {
		if slices.Equal(pk.deviceInfo.devices, devices) && pk.deviceInfo.precompute == precompute {
			return nil
		}
		if pk.users > 0 {
			return fmt.Errorf("proving key is in use on devices %v (precompute %d); cannot re-pin it for devices %v (precompute %d) now",
				pk.deviceInfo.devices, pk.deviceInfo.precompute, devices, precompute)
		}
		pk.freeLocked()
	}
	di := &deviceInfo{devices: slices.Clone(devices), precompute: precompute}
	remove := kRemoveList(info)
	for shard, dev := range devices {
		ctx, err := ga.ContextFor(dev)
		if err != nil {
			di.free()
			return err
		}
		b, err := ctx.NewKeyBuilder(curveID, pk.Domain.Cardinality, uint64(len(pk.InfinityA)), shard, len(devices))
		if err != nil {
			di.free()
			return err
		}
		stage := func() error {
			g1 := unsafe.Sizeof(curve.G1Affine{})
			g2 := unsafe.Sizeof(curve.G2Affine{})
			if err := b.Vector(ga.KeyG1A, sliceData(pk.G1.A), uint64(len(pk.G1.A)), g1); err != nil {
				return err
			}
			if err := b.Vector(ga.KeyG1B, sliceData(pk.G1.B), uint64(len(pk.G1.B)), g1); err != nil {
				return err
			}
			if err := b.Vector(ga.KeyG1Z, sliceData(pk.G1.Z), uint64(len(pk.G1.Z)), g1); err != nil {
				return err
			}
			if err := b.Vector(ga.KeyG1K, sliceData(pk.G1.K), uint64(len(pk.G1.K)), g1); err != nil {
				return err
			}
			if err := b.Vector(ga.KeyG2B, sliceData(pk.G2.B), uint64(len(pk.G2.B)), g2); err != nil {
				return err
			}
			for which, p := range map[int]unsafe.Pointer{
				ga.KeyG1Alpha: unsafe.Pointer(&pk.G1.Alpha), ga.KeyG1Beta: unsafe.Pointer(&pk.G1.Beta), ga.KeyG1Delta: unsafe.Pointer(&pk.G1.Delta),
				ga.KeyG2Beta: unsafe.Pointer(&pk.G2.Beta), ga.KeyG2Delta: unsafe.Pointer(&pk.G2.Delta),
			} {
				if err := b.Point(which, p); err != nil {
					return err
				}
			}
			if err := b.Infinity(0, pk.InfinityA); err != nil {
				return err
			}
			if err := b.Infinity(1, pk.InfinityB); err != nil {
				return err
			}
			// commitment keys live whole on shard 0 only: their MSMs run inside the solver hint, before the sharded part
			if shard == 0 {
				for i := range pk.CommitmentKeys {
					ck := &pk.CommitmentKeys[i]
					if len(ck.Basis) != len(ck.BasisExpSigma) {
						return fmt.Errorf("commitment key %d: len(Basis) != len(BasisExpSigma)", i)
					}
					if err := b.CommitmentKey(sliceData(ck.Basis), sliceData(ck.BasisExpSigma), uint64(len(ck.Basis))); err != nil {
						return err
					}
				}
			}
			return b.KRemove(remove)
		}
		if err := stage(); err != nil {
			b.Abandon()
			di.free()
			return err
		}
		key, err := b.Finish(precompute)
		if err != nil {
			di.free()
			return err
		}
		di.keys = append(di.keys, key)
	}
	pk.deviceInfo = di
	return nil
}

// PinFromFile loads the device copy of the key directly from a key file written by WriteTo, WriteRawTo or WriteDump
// (the format is recognised from the stream): the counterpart of ReadFrom / ReadDump (marshal.go:305-373,449-539) for
// provers that never need the key on the host -- the file goes through pinned staging buffers into HBM and is decoded
// there.  Every configured device reads the file once and keeps its shard.  The embedded native key stays empty, so
// such a key can prove but cannot be re-serialized; commitment information comes from the constraint system.
func (pk *ProvingKey) PinFromFile(path string, cfg *mi355x.Config, info constraint.Groth16Commitments) error {
	pk.setupMu.Lock()
	defer pk.setupMu.Unlock()
	if pk.users > 0 {
		return fmt.Errorf("proving key is in use by %d prover(s); cannot replace its device copy now", pk.users)
	}
	pk.freeLocked()
	devices := cfg.DeviceIDs()
	di := &deviceInfo{devices: slices.Clone(devices), precompute: int32(cfg.Precompute)} // a key read from a file stays pinned: the caller's policy as is
	remove := kRemoveList(info)
	for shard, dev := range devices {
		ctx, err := ga.ContextFor(dev)
		if err != nil {
			di.free()
			return err
		}
		f, err := os.Open(path)
		if err != nil {
			di.free()
			return err
		}
		key, _, err := ctx.ReadKeyFd(curveID, f.Fd(), int32(cfg.Precompute), shard, len(devices), remove)
		f.Close()
		if err != nil {
			di.free()
			return fmt.Errorf("device %d: %w", dev, err)
		}
		di.keys = append(di.keys, key)
	}
	pk.deviceInfo = di
	pk.PinToGPU = true
	return nil
}

func (di *deviceInfo) free() {
	for _, k := range di.keys {
		k.Free()
	}
	di.keys = nil
}

func (pk *ProvingKey) freeLocked() {
	if pk.deviceInfo != nil {
		pk.deviceInfo.free()
		pk.deviceInfo = nil
	}
}

// FreeGPUResources releases the device copy of the key (vectors, window tables, commitment keys).  Idempotent; the
// next Prove pins the key again.  While other goroutines are proving on the key the release is deferred to the last of
// them (the device memory must outlive their proofs).  Counterpart of icicle.go:1493-1549.
func (pk *ProvingKey) FreeGPUResources() {
	pk.setupMu.Lock()
	defer pk.setupMu.Unlock()
	if pk.users > 0 {
		pk.freePending = true
		return
	}
	pk.freeLocked()
}

// oneShot reports whether this proof takes the upload-while-proving path: nothing pinned or asked to be, the whole key on one
// device, plain vectors, no commitments, and no device copy already there (a second prover on a key that IS on the device uses it).
func (pk *ProvingKey) oneShot(cfg *mi355x.Config, info constraint.Groth16Commitments) bool {
	pk.setupMu.Lock()
	defer pk.setupMu.Unlock()
	return !pk.PinToGPU && !cfg.PinToGPU && pk.deviceInfo == nil && len(cfg.DeviceIDs()) == 1 && len(info) == 0 && len(pk.InfinityA) > 0 &&
		effectivePrecompute(cfg, false) == int32(mi355x.PrecomputeNever)
}

// proveOneShot is Prove for a key that stays on the host: solver on the CPU, then ga.ProveOneShot.
func proveOneShot(r1cs *cs.R1CS, pk *ProvingKey, fullWitness witness.Witness, cfg *mi355x.Config, opt backend.ProverConfig) (*groth16_bn254.Proof, error) {
	log := logger.Logger().With().Str("curve", r1cs.CurveID().String()).Str("acceleration", "mi355x").Int("nbConstraints", r1cs.GetNbConstraints()).Str("backend", "groth16").Logger()
	ctx, err := ga.ContextFor(cfg.DeviceIDs()[0])
	if err != nil {
		return nil, err
	}
	_solution, err := r1cs.Solve(fullWitness, opt.SolverOpts...)
	if err != nil {
		return nil, err
	}
	solution := _solution.(*cs.R1CSSolution)
	wireValues := []fr.Element(solution.W)
	start := time.Now()
	var r, s fr.Element // the prover's randomness (prove.go:171-177)
	if _, err := r.SetRandom(); err != nil {
		return nil, err
	}
	if _, err := s.SetRandom(); err != nil {
		return nil, err
	}
	var out struct {
		Ar  curve.G1Affine
		Bs  curve.G2Affine
		Krs curve.G1Affine
	}
	key := &ga.OneShotKey{
		Curve: curveID, DomainCardinality: pk.Domain.Cardinality,
		A: sliceData(pk.G1.A), LenA: uint64(len(pk.G1.A)), B: sliceData(pk.G1.B), LenB: uint64(len(pk.G1.B)),
		Z: sliceData(pk.G1.Z), LenZ: uint64(len(pk.G1.Z)), K: sliceData(pk.G1.K), LenK: uint64(len(pk.G1.K)),
		B2: sliceData(pk.G2.B), LenB2: uint64(len(pk.G2.B)),
		Alpha1: unsafe.Pointer(&pk.G1.Alpha), Beta1: unsafe.Pointer(&pk.G1.Beta), Delta1: unsafe.Pointer(&pk.G1.Delta),
		Beta2: unsafe.Pointer(&pk.G2.Beta), Delta2: unsafe.Pointer(&pk.G2.Delta),
		InfinityA: pk.InfinityA, InfinityB: pk.InfinityB, NbInfinityA: pk.NbInfinityA, NbInfinityB: pk.NbInfinityB,
	}
	err = ctx.ProveOneShot(key, sliceData(wireValues), sliceData([]fr.Element(solution.A)), sliceData([]fr.Element(solution.B)),
		sliceData([]fr.Element(solution.C)), uint64(len(solution.A)), uint64(r1cs.GetNbPublicVariables()), unsafe.Pointer(&r), unsafe.Pointer(&s),
		unsafe.Pointer(&out))
	if err != nil {
		return nil, err
	}
	proof := &groth16_bn254.Proof{Ar: out.Ar, Bs: out.Bs, Krs: out.Krs}
	log.Debug().Dur("took", time.Since(start)).Msg("prover done (one shot: key uploaded while proving)")
	return proof, nil
}

// Prove generates the proof of knowledge of a r1cs with full witness (secret + public part).
//
// The solver, the BSB22 hashing and the Fiat-Shamir fold stay on the CPU exactly as in
// backend/groth16/bn254/prove.go:52-135; everything between "the solver returned W, A, B, C" and "three affine
// points" (prove.go:130-315, icicle.go:981-1341) is one call into the library.
func Prove(r1cs *cs.R1CS, pk *ProvingKey, fullWitness witness.Witness, cfg *mi355x.Config) (*groth16_bn254.Proof, error) {
	opt, err := backend.NewProverConfig(cfg.ProverOpts...)
	if err != nil {
		return nil, fmt.Errorf("new prover config: %w", err)
	}
	if opt.HashToFieldFn == nil {
		opt.HashToFieldFn = hash_to_field.New([]byte(constraint.CommitmentDst))
	}
	log := logger.Logger().With().Str("curve", r1cs.CurveID().String()).Str("acceleration", "mi355x").Int("nbConstraints", r1cs.GetNbConstraints()).Str("backend", "groth16").Logger()

	commitmentInfo := r1cs.CommitmentInfo.(constraint.Groth16Commitments)
	// The default -- a key that is not kept on the device, one device, no window tables asked for, no BSB22 commitment (whose MSMs
	// run inside the solver, before the proof) -- goes through ONE call that uploads the key while the proof already runs and
	// drops it afterwards (ga_g16_prove_oneshot): about the longer of "6 GiB over PCIe" and "the proof" instead of their sum.
	if pk.oneShot(cfg, commitmentInfo) {
		return proveOneShot(r1cs, pk, fullWitness, cfg, opt)
	}
	// pin (if needed) and hold the device copy for the whole proof; an un-pinned key is freed by the last prover's release
	di, err := pk.acquire(cfg, commitmentInfo)
	if err != nil {
		return nil, fmt.Errorf("setup device pointers: %w", err)
	}
	defer pk.release()
	keys := di.keys

	proof := &groth16_bn254.Proof{Commitments: make([]curve.G1Affine, len(commitmentInfo))}
	poks := make([]curve.G1Affine, len(commitmentInfo))
	solverOpts := opt.SolverOpts[:len(opt.SolverOpts):len(opt.SolverOpts)]

	// BSB22 hint (prove.go:72-100, icicle.go:825-890): the commitment and its proof of knowledge are two MSMs over the
	// pinned pedersen bases, done in one device call with one upload of the committed values.
	bsb22ID := solver.GetHintID(fcs.Bsb22CommitmentComputePlaceholder)
	solverOpts = append(solverOpts, solver.OverrideHint(bsb22ID, func(_ *big.Int, in []*big.Int, out []*big.Int) error {
		i := int(in[0].Int64())
		in = in[1:]
		hashed := in[:len(commitmentInfo[i].PublicAndCommitmentCommitted)]
		committed := in[len(hashed):]
		values := make([]fr.Element, len(committed))
		for j, v := range committed {
			values[j].SetBigInt(v)
		}
		if err := keys[0].Commit(i, sliceData(values), uint64(len(values)), unsafe.Pointer(&proof.Commitments[i]), unsafe.Pointer(&poks[i])); err != nil {
			return err
		}
		opt.HashToFieldFn.Write(constraint.SerializeCommitment(proof.Commitments[i].Marshal(), hashed, (fr.Bits-1)/8+1))
		hashBts := opt.HashToFieldFn.Sum(nil)
		opt.HashToFieldFn.Reset()
		nbBuf := fr.Bytes
		if opt.HashToFieldFn.Size() < fr.Bytes {
			nbBuf = opt.HashToFieldFn.Size()
		}
		var res fr.Element
		res.SetBytes(hashBts[:nbBuf])
		res.BigInt(out[0])
		return nil
	}))

	_solution, err := r1cs.Solve(fullWitness, solverOpts...)
	if err != nil {
		return nil, err
	}
	solution := _solution.(*cs.R1CSSolution)
	wireValues := []fr.Element(solution.W)
	start := time.Now()

	// fold the proofs of knowledge with the challenge derived from the commitment wire VALUES (prove.go:118-129)
	if len(commitmentInfo) > 0 {
		serialized := make([]byte, fr.Bytes*len(commitmentInfo))
		for i := range commitmentInfo {
			copy(serialized[fr.Bytes*i:], wireValues[commitmentInfo[i].CommitmentIndex].Marshal())
		}
		challenge, err := fr.Hash(serialized, []byte("G16-BSB22"), 1)
		if err != nil {
			return nil, err
		}
		if _, err = proof.CommitmentPok.Fold(poks, challenge[0], ecc.MultiExpConfig{NbTasks: 1}); err != nil {
			return nil, err
		}
	}

	// the prover's randomness (prove.go:171-177); handed to the library as fr.Element images
	var r, s fr.Element
	if _, err := r.SetRandom(); err != nil {
		return nil, err
	}
	if _, err := s.SetRandom(); err != nil {
		return nil, err
	}

	// layout of the library's proof_out: Ar | Bs | Krs
	var out struct {
		Ar  curve.G1Affine
		Bs  curve.G2Affine
		Krs curve.G1Affine
	}
	nbConstraints := uint64(len(solution.A))
	nbPublic := uint64(r1cs.GetNbPublicVariables())
	w, a, b, c := sliceData(wireValues), sliceData([]fr.Element(solution.A)), sliceData([]fr.Element(solution.B)), sliceData([]fr.Element(solution.C))
	if len(keys) == 1 {
		err = keys[0].Prove(w, a, b, c, nbConstraints, nbPublic, unsafe.Pointer(&r), unsafe.Pointer(&s), unsafe.Pointer(&out))
	} else {
		err = ga.ProveMulti(keys, w, a, b, c, nbConstraints, nbPublic, unsafe.Pointer(&r), unsafe.Pointer(&s), unsafe.Pointer(&out))
	}
	if err != nil {
		return nil, err
	}
	proof.Ar, proof.Bs, proof.Krs = out.Ar, out.Bs, out.Krs

	if cfg.StepProfile {
		for _, dev := range di.devices {
			if ctx, err := ga.ContextFor(dev); err == nil {
				if stages, err := ctx.ReadProfile(); err == nil {
					log.Debug().Int("device", dev).Str("stages_ms", stages).Msg("mi355x step profile")
				}
			}
		}
	}
	log.Debug().Dur("took", time.Since(start)).Msg("prover done")
	return proof, nil
}
