//go:build mi355x

// Package bn254 holds the device hooks of the PLONK BN254 prover: the KZG commitments, the grand product and the
// quotient polynomial of backend/plonk/bn254/prove.go computed by libgnark_amd.  The round structure, the transcript,
// the linearised polynomial and the batch opening's transcript stay in the native prover.  prove.patch (next to this file)
// is the wiring: it adds an `Accelerator` interface and `ProveWithAccelerator` to backend/plonk/bn254/prove.go and routes the
// call sites below through it; *Device implements that interface.  tests/c_abi/plonk_pattern.c replays the resulting call
// sequence against the library.
//
//	prove.go:404-489   commitToLRO               -> Device.CommitLagrangeBatch (3 MSMs over pk.KzgLagrange.G1 in one pass)
//	prove.go:530-540   commitToPolyAndBlinding   -> Device.CommitLagrange
//	prove.go:645-655   iop.BuildRatioCopyConstraint -> Device.BuildRatioCopyConstraint
//	prove.go:558-633   computeNumerator + divideByZH (:841-1123,1287-1350) -> Device.ComputeQuotient
//	prove.go:1263-1285 commitToQuotient          -> Device.CommitBatch (3 MSMs over pk.Kzg.G1 in one pass)
//	prove.go:681,788,827 kzg.Open                -> Device.Open
package bn254

import (
	"fmt"
	"unsafe"

	curve "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark-crypto/ecc/bn254/kzg"
	"github.com/consensys/gnark/backend/accelerated/mi355x/internal/ga"
	plonk_bn254 "github.com/consensys/gnark/backend/plonk/bn254"
)

// Device is the device-side state of one PLONK proving key: both SRS as pinned window tables, the two FFT domains and
// (after PinTrace) the coset evaluations of the circuit-constant polynomials.
type Device struct {
	ctx              *ga.Context
	srs, srsLagrange *ga.Table
	domain0, domain1 *ga.Domain
	trace            *ga.PlonkKey
	nbBsb            int
}

func sliceData[T any](s []T) unsafe.Pointer { return unsafe.Pointer(unsafe.SliceData(s)) }

// NewDevice pins pk.Kzg.G1 and pk.KzgLagrange.G1 (setup.go:88-93) on the device and creates the domains of
// cardinality n and rho*n (prove.go:240-251: rho = 4, or 8 below 6 constraints).
func NewDevice(pk *plonk_bn254.ProvingKey, n uint64, deviceID int) (*Device, error) {
	ctx, err := ga.ContextFor(deviceID)
	if err != nil {
		return nil, err
	}
	d := &Device{ctx: ctx}
	fail := func(err error) (*Device, error) {
		d.Free()
		return nil, err
	}
	if d.srs, err = ctx.NewTable(ga.BN254, ga.G1, sliceData(pk.Kzg.G1), uint64(len(pk.Kzg.G1)), true); err != nil {
		return fail(err)
	}
	if d.srsLagrange, err = ctx.NewTable(ga.BN254, ga.G1, sliceData(pk.KzgLagrange.G1), uint64(len(pk.KzgLagrange.G1)), true); err != nil {
		return fail(err)
	}
	rho := uint64(4)
	if n < 6 {
		rho = 8
	}
	if d.domain0, err = ctx.NewDomain(ga.BN254, n); err != nil {
		return fail(err)
	}
	if d.domain1, err = ctx.NewDomain(ga.BN254, rho*n); err != nil {
		return fail(err)
	}
	return d, nil
}

// Free releases everything the device holds for this key.
func (d *Device) Free() {
	d.trace.Free()
	d.domain1.Free()
	d.domain0.Free()
	d.srsLagrange.Free()
	d.srs.Free()
}

func commit(t *ga.Table, p []fr.Element) (curve.G1Affine, error) {
	var res curve.G1Affine
	if uint64(len(p)) > t.N {
		return res, fmt.Errorf("polynomial has %d coefficients, the pinned SRS %d points", len(p), t.N)
	}
	scalars := p
	if uint64(len(p)) < t.N { // the table MSM takes exactly t.N scalars; zero scalars cost nothing on the device
		scalars = make([]fr.Element, t.N)
		copy(scalars, p)
	}
	var jac curve.G1Jac
	if err := t.Run(sliceData(scalars), unsafe.Pointer(&jac)); err != nil {
		return res, err
	}
	res.FromJacobian(&jac)
	return res, nil
}

// commitBatch is len(ps) kzg.Commit calls over the same pinned SRS in one device pass (commitToLRO's three goroutines,
// prove.go:404-489, and commitToQuotient's three shards, prove.go:1263-1285, become one call each).
func commitBatch(t *ga.Table, ps [][]fr.Element) ([]curve.G1Affine, error) {
	ptrs := make([]unsafe.Pointer, len(ps))
	for i, p := range ps {
		if uint64(len(p)) > t.N {
			return nil, fmt.Errorf("polynomial %d has %d coefficients, the pinned SRS %d points", i, len(p), t.N)
		}
		scalars := p
		if uint64(len(p)) < t.N {
			scalars = make([]fr.Element, t.N)
			copy(scalars, p)
		}
		ptrs[i] = sliceData(scalars)
	}
	jacs := make([]curve.G1Jac, len(ps))
	if err := t.RunBatch(ptrs, unsafe.Pointer(&jacs[0])); err != nil {
		return nil, err
	}
	res := make([]curve.G1Affine, len(ps))
	for i := range jacs {
		res[i].FromJacobian(&jacs[i])
	}
	return res, nil
}

// CommitBatch is kzg.Commit(p, pk.Kzg) for every p of ps (canonical form) in one pass.
func (d *Device) CommitBatch(ps ...[]fr.Element) ([]curve.G1Affine, error) { return commitBatch(d.srs, ps) }

// CommitLagrangeBatch is kzg.Commit(p, pk.KzgLagrange) for every p of ps (Lagrange form) in one pass.
func (d *Device) CommitLagrangeBatch(ps ...[]fr.Element) ([]curve.G1Affine, error) {
	return commitBatch(d.srsLagrange, ps)
}

// Commit is kzg.Commit(p, pk.Kzg): p in canonical form.
func (d *Device) Commit(p []fr.Element) (curve.G1Affine, error) { return commit(d.srs, p) }

// CommitLagrange is kzg.Commit(p, pk.KzgLagrange): p in Lagrange form (regular layout).
func (d *Device) CommitLagrange(p []fr.Element) (curve.G1Affine, error) { return commit(d.srsLagrange, p) }

// Open is kzg.Open(p, point, pk.Kzg).
func (d *Device) Open(p []fr.Element, point fr.Element) (kzg.OpeningProof, error) {
	var res kzg.OpeningProof
	var h curve.G1Jac
	if err := d.srs.Open(sliceData(p), uint64(len(p)), unsafe.Pointer(&point), unsafe.Pointer(&res.ClaimedValue), unsafe.Pointer(&h)); err != nil {
		return res, err
	}
	res.H.FromJacobian(&h)
	return res, nil
}

// BuildRatioCopyConstraint is iop.BuildRatioCopyConstraint([L, R, O], s.trace.S, beta, gamma, Lagrange/Regular, domain0):
// l, r, o are the evaluations on domain0, permutation is s.trace.S; the result is Z in Lagrange form, regular layout.
func (d *Device) BuildRatioCopyConstraint(l, r, o []fr.Element, permutation []int64, beta, gamma fr.Element) ([]fr.Element, error) {
	n := d.domain0.Cardinality
	if uint64(len(l)) != n || uint64(len(r)) != n || uint64(len(o)) != n || uint64(len(permutation)) != 3*n {
		return nil, fmt.Errorf("L, R, O need %d evaluations each and the permutation %d entries", n, 3*n)
	}
	z := make([]fr.Element, n)
	err := d.domain0.BuildZ(sliceData(l), sliceData(r), sliceData(o), permutation, unsafe.Pointer(&beta), unsafe.Pointer(&gamma), sliceData(z))
	return z, err
}

// Polys are the polynomials of the quotient in the order of prove.go:44-59 (without ZS, which is Z shifted): n
// coefficients each, canonical unless the matching bit of LagrangeMask is set (bit order L R O Z Ql Qr Qm Qo Qk S1 S2 S3,
// then Qcp_0, Pi2_0, Qcp_1, ...).
type Polys struct {
	L, R, O, Z, Ql, Qr, Qm, Qo, Qk, S1, S2, S3 []fr.Element
	Qcp, Pi2                                   [][]fr.Element
	LagrangeMask                               uint64
	Bl, Br, Bo                                 []fr.Element // blinding polynomials of L, R, O: 2 coefficients
	Bz                                         []fr.Element // of Z: 3 coefficients
	Alpha, Beta, Gamma                         fr.Element
}

func (p *Polys) input() *ga.QuotientInput {
	in := &ga.QuotientInput{
		L: sliceData(p.L), R: sliceData(p.R), O: sliceData(p.O), Z: sliceData(p.Z),
		Ql: sliceData(p.Ql), Qr: sliceData(p.Qr), Qm: sliceData(p.Qm), Qo: sliceData(p.Qo), Qk: sliceData(p.Qk),
		S1: sliceData(p.S1), S2: sliceData(p.S2), S3: sliceData(p.S3),
		LagrangeMask: p.LagrangeMask,
		Bl: sliceData(p.Bl), Br: sliceData(p.Br), Bo: sliceData(p.Bo), Bz: sliceData(p.Bz),
		Alpha: unsafe.Pointer(&p.Alpha), Beta: unsafe.Pointer(&p.Beta), Gamma: unsafe.Pointer(&p.Gamma),
	}
	for i := range p.Qcp {
		in.Qcp = append(in.Qcp, sliceData(p.Qcp[i]))
	}
	for i := range p.Pi2 {
		in.Pi2 = append(in.Pi2, sliceData(p.Pi2[i]))
	}
	return in
}

// PinTrace evaluates the circuit constants (Ql, Qr, Qm, Qo, S1, S2, S3 and every Qcp) on all cosets once and keeps
// them in HBM -- the precomputation prove.go:1030-1034 rules out on a CPU for its memory footprint.  Only those
// fields of p are read.
func (d *Device) PinTrace(p *Polys) error {
	k, err := ga.NewPlonkKey(d.domain0, d.domain1, p.input())
	if err != nil {
		return err
	}
	d.trace.Free()
	d.trace, d.nbBsb = k, len(p.Qcp)
	return nil
}

// ComputeQuotient returns s.h = divideByZH(computeNumerator()) as rho*n canonical coefficients in regular order;
// h1, h2, h3 of prove.go:691-728 are its slices.  With a pinned trace only L, R, O, Z, Qk, Pi2, the blinding
// polynomials and the challenges of p are read.
func (d *Device) ComputeQuotient(p *Polys) ([]fr.Element, error) {
	h := make([]fr.Element, d.domain1.Cardinality)
	if d.trace != nil {
		if len(p.Pi2) != d.nbBsb {
			return nil, fmt.Errorf("%d committed polynomials for a trace pinned with %d Qcp", len(p.Pi2), d.nbBsb)
		}
		return h, d.trace.Quotient(p.input(), sliceData(h))
	}
	return h, ga.PlonkQuotient(d.domain0, d.domain1, p.input(), sliceData(h))
}

// ComputeQuotientRaw is ComputeQuotient on plain slices -- the method of the prover's Accelerator interface (prove.patch), which
// cannot name this package's types: polys in the order L R O Z Ql Qr Qm Qo Qk S1 S2 S3.
func (d *Device) ComputeQuotientRaw(polys [12][]fr.Element, qcp, pi2 [][]fr.Element, lagrangeMask uint64, bl, br, bo, bz []fr.Element,
	alpha, beta, gamma fr.Element) ([]fr.Element, error) {
	return d.ComputeQuotient(&Polys{
		L: polys[0], R: polys[1], O: polys[2], Z: polys[3], Ql: polys[4], Qr: polys[5], Qm: polys[6], Qo: polys[7], Qk: polys[8],
		S1: polys[9], S2: polys[10], S3: polys[11], Qcp: qcp, Pi2: pi2, LagrangeMask: lagrangeMask,
		Bl: bl, Br: br, Bo: bo, Bz: bz, Alpha: alpha, Beta: beta, Gamma: gamma,
	})
}

// LinearCombination is sum_i coeffs[i] * polys[i] (polynomials of different lengths are padded with zeros): the fold of
// kzg.BatchOpenSinglePoint, 16 polynomials per device pass.
func (d *Device) LinearCombination(polys [][]fr.Element, coeffs []fr.Element) ([]fr.Element, error) {
	if len(polys) == 0 || len(polys) != len(coeffs) {
		return nil, fmt.Errorf("%d polynomials, %d coefficients", len(polys), len(coeffs))
	}
	n := 0
	for _, p := range polys {
		n = max(n, len(p))
	}
	acc := make([]fr.Element, n)
	one := make([]fr.Element, 1)
	one[0].SetOne()
	for lo := 0; lo < len(polys); lo += 15 {
		hi := min(lo+15, len(polys))
		vecs := []unsafe.Pointer{sliceData(acc)} // the running sum is the first term of every pass
		sc := append(one[:1:1], coeffs[lo:hi]...)
		for _, p := range polys[lo:hi] {
			if len(p) < n {
				q := make([]fr.Element, n)
				copy(q, p)
				p = q
			}
			vecs = append(vecs, sliceData(p))
		}
		if err := d.ctx.LinearCombination(ga.BN254, uint64(n), vecs, sliceData(sc), sliceData(acc)); err != nil {
			return nil, err
		}
	}
	return acc, nil
}

// FFT runs fft.Domain.FFT / FFTInverse of domain0 (big = false) or domain1 (big = true) on the device, in place.
func (d *Device) FFT(a []fr.Element, big, inverse bool, decimation int, onCoset bool) error {
	dom := d.domain0
	if big {
		dom = d.domain1
	}
	if uint64(len(a)) != dom.Cardinality {
		return fmt.Errorf("len(a) = %d, domain cardinality %d", len(a), dom.Cardinality)
	}
	return dom.FFT(sliceData(a), inverse, decimation, onCoset)
}
