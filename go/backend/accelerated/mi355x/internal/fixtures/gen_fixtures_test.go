//go:build mi355x_fixtures

// Package fixtures writes golden directories from gnark's OWN CPU prover for the MI355X backend's parity tests
// (tests/gnark_fixture.py consumes them; the layout is documented there and in README.md next to this file).
//
// Needs the test-only hook of groth16_rs_<curve>.patch in backend/groth16/<curve>/prove.go: gnark samples the prover's
// randomness r, s from crypto/rand (prove.go:171-177) and offers no way to fix it (backend.ProverConfig has no field for it,
// backend/backend.go:59-65), so byte-identical proofs from two provers need the hook.  Usage, from a gnark checkout with this
// tree copied in and both patches applied:
//
//	patch -p1 < backend/accelerated/mi355x/internal/fixtures/groth16_rs_bn254.patch
//	patch -p1 < backend/accelerated/mi355x/internal/fixtures/groth16_rs_bls12-381.patch
//	GNARK_AMD_GOLDEN=/path/to/gnark_amd/tests/golden/gnark go test -tags mi355x_fixtures -run TestWriteFixtures ./backend/accelerated/mi355x/internal/fixtures/
//
// No Go toolchain exists in the image this repository is built in: the file has never been compiled; its package-qualified
// identifiers are resolved against the reference by tools/check_go_idents.py (go/IDENTS.json).
package fixtures

import (
	"bytes"
	"encoding/json"
	"fmt"
	"math/big"
	"os"
	"path/filepath"
	"testing"

	"github.com/consensys/gnark-crypto/ecc"
	fr_bls12381 "github.com/consensys/gnark-crypto/ecc/bls12-381/fr"
	fr_bn254 "github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark/backend/groth16"
	groth16_bls12381 "github.com/consensys/gnark/backend/groth16/bls12-381"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/constraint"
	cs_bls12381 "github.com/consensys/gnark/constraint/bls12-381"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/frontend"
	"github.com/consensys/gnark/frontend/cs/r1cs"
)

// cubicCircuit is examples/cubic: x**3 + x + 5 == y.
type cubicCircuit struct {
	X frontend.Variable `gnark:"x"`
	Y frontend.Variable `gnark:",public"`
}

func (c *cubicCircuit) Define(api frontend.API) error {
	x3 := api.Mul(c.X, c.X, c.X)
	api.AssertIsEqual(c.Y, api.Add(x3, c.X, 5))
	return nil
}

// twoCommitments commits twice (BSB22): the second commitment commits to the first.
type twoCommitments struct {
	X frontend.Variable
	Y frontend.Variable `gnark:",public"`
}

func (c *twoCommitments) Define(api frontend.API) error {
	committer, ok := api.(frontend.Committer)
	if !ok {
		return fmt.Errorf("builder does not implement frontend.Committer")
	}
	c0, err := committer.Commit(c.X, c.Y)
	if err != nil {
		return err
	}
	x2 := api.Mul(c.X, c.X)
	c1, err := committer.Commit(x2, c0)
	if err != nil {
		return err
	}
	api.AssertIsDifferent(c0, c1)
	api.AssertIsEqual(api.Mul(x2, c.X), c.Y)
	return nil
}

// squaringChain is refCircuit of backend/groth16/groth16_test.go:120-132 at 2^10 constraints.
type squaringChain struct {
	nbConstraints int
	X             frontend.Variable
	Y             frontend.Variable `gnark:",public"`
}

func (c *squaringChain) Define(api frontend.API) error {
	x := c.X
	for i := 0; i < c.nbConstraints; i++ {
		x = api.Mul(x, x)
	}
	api.AssertIsEqual(x, c.Y)
	return nil
}

type meta struct {
	Curve       string           `json:"curve"`
	NbPublic    int              `json:"nb_public"`
	Commitments []commitmentMeta `json:"commitments"`
	Producer    string           `json:"producer"`
}

type commitmentMeta struct {
	PrivateCommitted             []int `json:"private_committed"`
	PublicAndCommitmentCommitted []int `json:"public_and_commitment_committed"`
	CommitmentIndex              int   `json:"commitment_index"`
}

func be32(v *big.Int) []byte {
	out := make([]byte, 32)
	v.FillBytes(out)
	return out
}

// fixed, non-trivial randomness (any value below both scalar field moduli works)
var fixedR, _ = new(big.Int).SetString("1234567890123456789012345678901234567890123456789012345678901234567", 10)
var fixedS, _ = new(big.Int).SetString("987654321098765432109876543210987654321098765432109876543210987654", 10)

func writeCase(t *testing.T, root, name string, id ecc.ID, circuit, assignment frontend.Circuit) {
	ccs, err := frontend.Compile(id.ScalarField(), r1cs.NewBuilder, circuit)
	if err != nil {
		t.Fatal(err)
	}
	pk, _, err := groth16.Setup(ccs)
	if err != nil {
		t.Fatal(err)
	}
	w, err := frontend.NewWitness(assignment, id.ScalarField())
	if err != nil {
		t.Fatal(err)
	}
	var solution bytes.Buffer
	switch id {
	case ecc.BN254:
		groth16_bn254.TestingHooks.Randomness = func() (r, s fr_bn254.Element) {
			r.SetBigInt(fixedR)
			s.SetBigInt(fixedS)
			return
		}
		// (serialised inside the hook: computeH pads and then drops A, B, C in place, prove.go:134-140)
		groth16_bn254.TestingHooks.Solution = func(sol *cs_bn254.R1CSSolution) { _, _ = sol.WriteTo(&solution) }
		defer func() { groth16_bn254.TestingHooks.Randomness, groth16_bn254.TestingHooks.Solution = nil, nil }()
	case ecc.BLS12_381:
		groth16_bls12381.TestingHooks.Randomness = func() (r, s fr_bls12381.Element) {
			r.SetBigInt(fixedR)
			s.SetBigInt(fixedS)
			return
		}
		groth16_bls12381.TestingHooks.Solution = func(sol *cs_bls12381.R1CSSolution) { _, _ = sol.WriteTo(&solution) }
		defer func() { groth16_bls12381.TestingHooks.Randomness, groth16_bls12381.TestingHooks.Solution = nil, nil }()
	default:
		t.Fatalf("curve %s is not built into the MI355X backend", id)
	}
	proof, err := groth16.Prove(ccs, pk, w)
	if err != nil {
		t.Fatal(err)
	}
	m := meta{Curve: map[ecc.ID]string{ecc.BN254: "bn254", ecc.BLS12_381: "bls12-381"}[id], NbPublic: ccs.GetNbPublicVariables(),
		Commitments: []commitmentMeta{}, Producer: "gnark (backend/groth16 CPU prover) + groth16_rs patch"}
	if infos, ok := ccs.GetCommitments().(constraint.Groth16Commitments); ok {
		for _, ci := range infos {
			m.Commitments = append(m.Commitments, commitmentMeta{PrivateCommitted: ci.PrivateCommitted,
				PublicAndCommitmentCommitted: ci.PublicAndCommitmentCommitted, CommitmentIndex: ci.CommitmentIndex})
		}
	}
	dir := filepath.Join(root, name+"_"+m.Curve)
	if err := os.MkdirAll(dir, 0o755); err != nil {
		t.Fatal(err)
	}
	var pkRaw, proofBin, proofRaw bytes.Buffer
	if _, err := pk.WriteRawTo(&pkRaw); err != nil {
		t.Fatal(err)
	}
	if _, err := proof.WriteTo(&proofBin); err != nil {
		t.Fatal(err)
	}
	if _, err := proof.WriteRawTo(&proofRaw); err != nil {
		t.Fatal(err)
	}
	mj, _ := json.MarshalIndent(m, "", " ")
	files := map[string][]byte{"meta.json": mj, "pk.bin": pkRaw.Bytes(), "solution.bin": solution.Bytes(), "r.bin": be32(fixedR),
		"s.bin": be32(fixedS), "proof.bin": proofBin.Bytes(), "proof.raw": proofRaw.Bytes()}
	for f, data := range files {
		if err := os.WriteFile(filepath.Join(dir, f), data, 0o644); err != nil {
			t.Fatal(err)
		}
	}
}

func TestWriteFixtures(t *testing.T) {
	root := os.Getenv("GNARK_AMD_GOLDEN")
	if root == "" {
		t.Skip("GNARK_AMD_GOLDEN is not set")
	}
	for _, id := range []ecc.ID{ecc.BN254, ecc.BLS12_381} {
		writeCase(t, root, "cubic", id, &cubicCircuit{}, &cubicCircuit{X: 3, Y: 35})
		writeCase(t, root, "two_commitments", id, &twoCommitments{}, &twoCommitments{X: 3, Y: 27})
		y := new(big.Int).Exp(big.NewInt(2), new(big.Int).Lsh(big.NewInt(1), 1<<10), id.ScalarField())
		writeCase(t, root, "squaring_chain_2p10", id, &squaringChain{nbConstraints: 1 << 10}, &squaringChain{nbConstraints: 1 << 10, X: 2, Y: y})
	}
}
