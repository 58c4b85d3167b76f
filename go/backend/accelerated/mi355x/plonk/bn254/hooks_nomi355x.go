//go:build !mi355x

// Package bn254 holds the device hooks of the PLONK BN254 prover; without the `mi355x` build tag it is empty and the
// native prover runs unchanged.
package bn254
