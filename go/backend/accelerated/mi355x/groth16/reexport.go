// Package groth16 is the Groth16 prover of gnark with the MI355X library behind Prove (build tag mi355x).
//
// What needs no device is gnark's own and is re-exported unchanged, so that switching a program from
// backend/groth16 (or from backend/accelerated/icicle/groth16, whose groth16_all.go:13-31 offers the same four names)
// is a change of import path only.
package groth16

import "github.com/consensys/gnark/backend/groth16"

var (
	// Verify is [groth16.Verify]: verification stays on the CPU.
	Verify = groth16.Verify
	// NewVerifyingKey is [groth16.NewVerifyingKey] (an empty key to deserialize into).
	NewVerifyingKey = groth16.NewVerifyingKey
	// NewProof is [groth16.NewProof] (an empty proof to deserialize into).
	NewProof = groth16.NewProof
	// NewCS is [groth16.NewCS] (a typed R1CS to deserialize into).
	NewCS = groth16.NewCS
)
