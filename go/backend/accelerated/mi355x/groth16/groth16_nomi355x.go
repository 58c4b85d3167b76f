//go:build !mi355x

package groth16

import (
	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/accelerated/mi355x"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
)

const noTag = "mi355x backend requested but program compiled without 'mi355x' build tag"

// Prove generates the proof of knowledge of a r1cs with full witness (secret + public part) on the GPU.
func Prove(r1cs constraint.ConstraintSystem, pk groth16.ProvingKey, fullWitness witness.Witness, opts ...mi355x.Option) (groth16.Proof, error) {
	panic(noTag)
}

// Setup generates a proving and verifying key for a given r1cs; the proving key is the accelerated type.
func Setup(r1cs constraint.ConstraintSystem) (groth16.ProvingKey, groth16.VerifyingKey, error) {
	panic(noTag)
}

// DummySetup generates a dummy accelerated proving key for a given circuit (benchmarks and tests).
func DummySetup(r1cs constraint.ConstraintSystem) (groth16.ProvingKey, error) {
	panic(noTag)
}

// NewProvingKey creates an empty accelerated proving key for deserializing into.
func NewProvingKey(curveID ecc.ID) groth16.ProvingKey {
	panic(noTag)
}
