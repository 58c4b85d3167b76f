//go:build !mi355x

package groth16

import (
	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/accelerated/mi355x"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
)

// A build without the mi355x tag keeps the package's API and refuses to run it: the first call stops the program, as the ICICLE
// package does for its own tag (groth16_noicicle.go:18-48) -- a mis-tagged binary must not fall back to the CPU prover unnoticed.
func unavailable(entry string) {
	panic("mi355x/groth16." + entry + ": this program was built without the 'mi355x' build tag (go build -tags=mi355x)")
}

func Prove(constraint.ConstraintSystem, groth16.ProvingKey, witness.Witness, ...mi355x.Option) (proof groth16.Proof, err error) {
	unavailable("Prove")
	return
}

func Setup(constraint.ConstraintSystem) (pk groth16.ProvingKey, vk groth16.VerifyingKey, err error) {
	unavailable("Setup")
	return
}

func DummySetup(constraint.ConstraintSystem) (pk groth16.ProvingKey, err error) {
	unavailable("DummySetup")
	return
}

func NewProvingKey(ecc.ID) (pk groth16.ProvingKey) {
	unavailable("NewProvingKey")
	return
}
