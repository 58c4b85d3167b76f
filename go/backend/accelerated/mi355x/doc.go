// Package mi355x implements gnark proving backends accelerated by AMD Instinct MI355X GPUs through libgnark_amd
// (hand-written HIP kernels for gfx950 behind the C ABI of include/gnark_amd.h).
//
// It is the MI355X counterpart of [github.com/consensys/gnark/backend/accelerated/icicle] and is used the same way:
//
//	import mi355x_groth16 "github.com/consensys/gnark/backend/accelerated/mi355x/groth16"
//	...
//	pk := mi355x_groth16.NewProvingKey(ecc.BN254)
//	_, err = pk.ReadFrom(r)                       // same serialization as the native key
//	...
//	proof, err := mi355x_groth16.Prove(ccs, pk, witness, mi355x.WithPinKeysToGPU(true))
//	err = groth16.Verify(proof, vk, publicWitness) // the native verifier
//
// Build with the `mi355x` tag and point cgo at the library:
//
//	CGO_LDFLAGS="-L$GNARK_AMD/gnark_amd -Wl,-rpath,$GNARK_AMD/gnark_amd" go build -tags=mi355x ./...
//
// Without the tag every accelerated entry point panics, exactly like the ICICLE backend without `icicle`.
//
// Supported: Groth16 on BN254 and BLS12-381 (including BSB22 commitments), and the KZG / FFT core of PLONK on BN254.
//
// # Differences from the ICICLE backend
//
//   - The proving key is pinned with its window-multiple tables (72 GiB for a 2^24 BN254 key out of 288 GB of HBM);
//     [WithPrecompute] selects the policy.  [WithPinKeysToGPU](false), the default of the ICICLE backend, frees the
//     device copy after every proof and pays the pinning cost each time.
//   - One statement can be proved over several GPUs of a node with [WithDevices]: the key is sharded by base-point
//     range, each device proves its share and the partial sums are added on the host.
//   - There is no backend-library loading step (WithBackend / WithBackendLibrary): the kernels are linked in.
package mi355x
