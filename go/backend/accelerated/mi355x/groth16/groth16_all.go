// Package groth16 implements the Groth16 proof system with MI355X acceleration.
package groth16

import (
	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
)

// Verify verifies a Groth16 proof with the native verifier; provided for completeness.
func Verify(proof groth16.Proof, vk groth16.VerifyingKey, publicWitness witness.Witness, opts ...backend.VerifierOption) error {
	return groth16.Verify(proof, vk, publicWitness, opts...)
}

// NewVerifyingKey creates an empty verifying key for deserializing into. Compatible with [groth16.NewVerifyingKey].
func NewVerifyingKey(curveID ecc.ID) groth16.VerifyingKey {
	return groth16.NewVerifyingKey(curveID)
}

// NewProof creates an empty proof for deserializing into. Compatible with [groth16.NewProof].
func NewProof(curveID ecc.ID) groth16.Proof {
	return groth16.NewProof(curveID)
}

// NewCS creates a typed R1CS constraint system for the curve. Compatible with [groth16.NewCS].
func NewCS(curveID ecc.ID) constraint.ConstraintSystem {
	return groth16.NewCS(curveID)
}
