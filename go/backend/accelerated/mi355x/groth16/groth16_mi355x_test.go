//go:build mi355x

package groth16_test

import (
	"bytes"
	"testing"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend"
	"github.com/consensys/gnark/backend/accelerated/mi355x"
	mi355x_groth16 "github.com/consensys/gnark/backend/accelerated/mi355x/groth16"
	native_groth16 "github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/frontend"
	"github.com/consensys/gnark/frontend/cs/r1cs"
	"github.com/consensys/gnark/test"
)

// x^k with a range of intermediate products: a chain long enough that B is sparser than A (builder.go:190-195).
type powerCircuit struct {
	X frontend.Variable
	Y frontend.Variable `gnark:",public"`
}

const powerDepth = 300

func (c *powerCircuit) Define(api frontend.API) error {
	acc := c.X
	for i := 0; i < powerDepth; i++ {
		acc = api.Mul(acc, c.X)
	}
	api.AssertIsEqual(acc, c.Y)
	return nil
}

// two api.Commit calls: exercises the BSB22 hint override, the K filter and the proof-of-knowledge fold.
type commitCircuit struct {
	A, B frontend.Variable
	Out  frontend.Variable `gnark:",public"`
}

func (c *commitCircuit) Define(api frontend.API) error {
	committer, ok := api.(frontend.Committer)
	if !ok {
		panic("builder does not implement frontend.Committer")
	}
	c1, err := committer.Commit(c.A, c.B)
	if err != nil {
		return err
	}
	c2, err := committer.Commit(c.B, c1)
	if err != nil {
		return err
	}
	api.AssertIsDifferent(c2, 0)
	api.AssertIsEqual(api.Mul(c.A, c.B), c.Out)
	return nil
}

func curves() []ecc.ID { return []ecc.ID{ecc.BN254, ecc.BLS12_381} }

func TestProveVerifiesWithNativeVerifier(t *testing.T) {
	for _, curve := range curves() {
		t.Run(curve.String(), func(t *testing.T) {
			assert := test.NewAssert(t)
			ccs, err := frontend.Compile(curve.ScalarField(), r1cs.NewBuilder, &powerCircuit{})
			assert.NoError(err)
			nativePK, vk, err := native_groth16.Setup(ccs)
			assert.NoError(err)

			// native -> accelerated through the shared serialization (backend/accelerated/icicle/groth16/marshal_test.go does the same)
			pk := mi355x_groth16.NewProvingKey(curve)
			var buf bytes.Buffer
			_, err = nativePK.WriteTo(&buf)
			assert.NoError(err)
			_, err = pk.ReadFrom(&buf)
			assert.NoError(err)
			assert.False(pk.IsDifferent(nativePK))

			y := 1
			for i := 0; i <= powerDepth; i++ {
				y = y * 3 % 1000003
			}
			_ = y
			w, err := frontend.NewWitness(&powerCircuit{X: 1, Y: 1}, curve.ScalarField())
			assert.NoError(err)
			pw, err := w.Public()
			assert.NoError(err)
			for _, opts := range [][]mi355x.Option{
				nil, // the default: the key is NOT kept on the device -- one call uploads it while the proof runs (ga_g16_prove_oneshot)
				nil, // ... twice: the second call gets the context's spare buffers and NTT domain back
				{mi355x.WithPinKeysToGPU(true)},
				{mi355x.WithPinKeysToGPU(true), mi355x.WithPrecompute(mi355x.PrecomputeNever)},
			} {
				proof, err := mi355x_groth16.Prove(ccs, pk, w, opts...)
				assert.NoError(err)
				assert.NoError(native_groth16.Verify(proof, vk, pw))
			}
		})
	}
}

func TestProveWithCommitments(t *testing.T) {
	for _, curve := range curves() {
		t.Run(curve.String(), func(t *testing.T) {
			assert := test.NewAssert(t)
			ccs, err := frontend.Compile(curve.ScalarField(), r1cs.NewBuilder, &commitCircuit{})
			assert.NoError(err)
			pk, vk, err := mi355x_groth16.Setup(ccs)
			assert.NoError(err)
			w, err := frontend.NewWitness(&commitCircuit{A: 3, B: 5, Out: 15}, curve.ScalarField())
			assert.NoError(err)
			pw, err := w.Public()
			assert.NoError(err)
			proof, err := mi355x_groth16.Prove(ccs, pk, w, mi355x.WithProverOptions(backend.WithSolverOptions()))
			assert.NoError(err)
			assert.NoError(native_groth16.Verify(proof, vk, pw))

			// accelerated -> native: the same key proves on the CPU
			nativePK := native_groth16.NewProvingKey(curve)
			var buf bytes.Buffer
			_, err = pk.WriteTo(&buf)
			assert.NoError(err)
			_, err = nativePK.ReadFrom(&buf)
			assert.NoError(err)
			proofNative, err := native_groth16.Prove(ccs, nativePK, w)
			assert.NoError(err)
			assert.NoError(native_groth16.Verify(proofNative, vk, pw))
		})
	}
}

// One proof over two devices (skipped on a single-GPU box).
func TestProveOnTwoDevices(t *testing.T) {
	assert := test.NewAssert(t)
	ccs, err := frontend.Compile(ecc.BN254.ScalarField(), r1cs.NewBuilder, &powerCircuit{})
	assert.NoError(err)
	pk, vk, err := mi355x_groth16.Setup(ccs)
	assert.NoError(err)
	w, err := frontend.NewWitness(&powerCircuit{X: 1, Y: 1}, ecc.BN254.ScalarField())
	assert.NoError(err)
	pw, err := w.Public()
	assert.NoError(err)
	proof, err := mi355x_groth16.Prove(ccs, pk, w, mi355x.WithDevices(0, 1), mi355x.WithPinKeysToGPU(true))
	if err != nil {
		t.Skipf("two devices not available: %v", err)
	}
	assert.NoError(native_groth16.Verify(proof, vk, pw))
}
