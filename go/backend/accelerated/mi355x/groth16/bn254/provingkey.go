//go:build mi355x

package bn254

import (
	"sync"

	"github.com/consensys/gnark/backend/accelerated/mi355x/internal/ga"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
)

// deviceInfo is the device-side state of a proving key: one pinned key (or key shard) per device of the
// configuration it was pinned for.
type deviceInfo struct {
	devices    []int             // device ids, in shard order
	keys       []*ga.ProvingKey  // keys[i] = shard i of len(keys), resident on devices[i]
	precompute int32
}

// ProvingKey embeds the native Groth16 proving key, so ReadFrom / WriteTo / ReadDump / WriteDump are the native
// ones and a key serialized by either backend loads in the other (backend/accelerated/icicle/doc.go:51-57 makes
// the same promise).  The device state is created on the first Prove and released by FreeGPUResources, or after
// every proof unless PinToGPU is set.
type ProvingKey struct {
	groth16_bn254.ProvingKey
	*deviceInfo
	setupMu     sync.Mutex // protects deviceInfo, users, freePending and PinToGPU
	users       int        // provers currently inside Prove on this key (acquire / release)
	freePending bool       // FreeGPUResources was called while provers were active: the last one frees
	PinToGPU    bool       // keep the device copy between proofs (default false, like the ICICLE backend)
}
