// Package bn254 implements the MI355X-accelerated Groth16 prover for BN254.
//
// bls12-381/ is generated from this package by ../../internal/generator/gen.sh (imports and curve ids only).
package bn254
