// Package bls12381 implements the MI355X-accelerated Groth16 prover for BLS12-381.
//
// Generated from ../bn254 by ../../internal/generator/gen.sh.
package bls12381
