//go:build mi355x

package groth16

import (
	"fmt"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/accelerated/mi355x"
	mi355x_bls12381 "github.com/consensys/gnark/backend/accelerated/mi355x/groth16/bls12-381"
	mi355x_bn254 "github.com/consensys/gnark/backend/accelerated/mi355x/groth16/bn254"
	"github.com/consensys/gnark/backend/groth16"
	groth16_bls12381 "github.com/consensys/gnark/backend/groth16/bls12-381"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
	cs_bls12381 "github.com/consensys/gnark/constraint/bls12-381"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
)

const unsupported = "mi355x backend requested but the curve is not supported (bn254, bls12-381)"

// Prove generates the proof of knowledge of a r1cs with full witness (secret + public part).
//
// The proving key must be the accelerated type: create it with [NewProvingKey] and the serialization methods, or
// with [Setup] / [DummySetup].
func Prove(r1cs constraint.ConstraintSystem, pk groth16.ProvingKey, fullWitness witness.Witness, opts ...mi355x.Option) (groth16.Proof, error) {
	config, err := mi355x.NewConfig(opts...)
	if err != nil {
		return nil, fmt.Errorf("initializing config: %w", err)
	}
	switch _r1cs := r1cs.(type) {
	case *cs_bn254.R1CS:
		_pk, ok := pk.(*mi355x_bn254.ProvingKey)
		if !ok {
			return nil, fmt.Errorf("proving key is %T, expected the mi355x bn254 key (use NewProvingKey + ReadFrom)", pk)
		}
		return mi355x_bn254.Prove(_r1cs, _pk, fullWitness, config)
	case *cs_bls12381.R1CS:
		_pk, ok := pk.(*mi355x_bls12381.ProvingKey)
		if !ok {
			return nil, fmt.Errorf("proving key is %T, expected the mi355x bls12-381 key (use NewProvingKey + ReadFrom)", pk)
		}
		return mi355x_bls12381.Prove(_r1cs, _pk, fullWitness, config)
	default:
		panic(unsupported)
	}
}

// Setup wraps [groth16.Setup]; the returned proving key is the accelerated type (it embeds the native key, so
// the serialization is shared).
func Setup(r1cs constraint.ConstraintSystem) (groth16.ProvingKey, groth16.VerifyingKey, error) {
	switch _r1cs := r1cs.(type) {
	case *cs_bn254.R1CS:
		var pk mi355x_bn254.ProvingKey
		var vk groth16_bn254.VerifyingKey
		if err := groth16_bn254.Setup(_r1cs, &pk.ProvingKey, &vk); err != nil {
			return nil, nil, err
		}
		return &pk, &vk, nil
	case *cs_bls12381.R1CS:
		var pk mi355x_bls12381.ProvingKey
		var vk groth16_bls12381.VerifyingKey
		if err := groth16_bls12381.Setup(_r1cs, &pk.ProvingKey, &vk); err != nil {
			return nil, nil, err
		}
		return &pk, &vk, nil
	default:
		panic(unsupported)
	}
}

// DummySetup wraps [groth16.DummySetup] (all bases equal: a degenerate bucket distribution -- every bucket meets P + P and
// P - P; the library's complete lazy loop handles it at ~1.1x the cost of a key with distinct bases, DESIGN 4.2).
func DummySetup(r1cs constraint.ConstraintSystem) (groth16.ProvingKey, error) {
	switch _r1cs := r1cs.(type) {
	case *cs_bn254.R1CS:
		var pk mi355x_bn254.ProvingKey
		if err := groth16_bn254.DummySetup(_r1cs, &pk.ProvingKey); err != nil {
			return nil, err
		}
		return &pk, nil
	case *cs_bls12381.R1CS:
		var pk mi355x_bls12381.ProvingKey
		if err := groth16_bls12381.DummySetup(_r1cs, &pk.ProvingKey); err != nil {
			return nil, err
		}
		return &pk, nil
	default:
		panic(unsupported)
	}
}

// NewProvingKey creates an empty accelerated proving key for deserializing into; compatible with
// [groth16.NewProvingKey].
func NewProvingKey(curveID ecc.ID) groth16.ProvingKey {
	switch curveID {
	case ecc.BN254:
		return &mi355x_bn254.ProvingKey{}
	case ecc.BLS12_381:
		return &mi355x_bls12381.ProvingKey{}
	default:
		panic(unsupported)
	}
}
