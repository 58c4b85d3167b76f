package mi355x

import (
	"fmt"

	"github.com/consensys/gnark/backend"
)

// Precompute is the window-table policy of a pinned proving key.
type Precompute int32

const (
	// PrecomputeAuto: for a key kept on the device (WithPinKeysToGPU(true), PinFromFile) build [2^(c*w)]P tables for A, B, K, Z
	// and G2.B when they fit in 85 % of the free HBM; for a key that is uploaded per proof and freed afterwards (the default,
	// as in the ICICLE backend) upload the plain vectors only.
	PrecomputeAuto Precompute = 0
	// PrecomputeAlways builds them or fails.
	PrecomputeAlways Precompute = 1
	// PrecomputeNever keeps the plain affine vectors (6 GiB for a 2^24 BN254 key instead of 72 GiB): pinning takes 0.2 s instead of
	// 2.4 s, a 2^24 proof 194 ms instead of 144 ms -- the choice for fewer than ~50 proofs per key.
	PrecomputeNever Precompute = -1
)

// Config is the configuration of the MI355X backend (the counterpart of icicle.Config).
type Config struct {
	// DeviceID is the HIP device used when Devices is empty.
	DeviceID int
	// Devices, when it holds more than one id, shards ONE proof over these devices (base-point range sharding).
	Devices []int
	// ProverOpts are handed to backend.NewProverConfig (hash-to-field function, solver options, ...).
	ProverOpts []backend.ProverOption
	// PinToGPU keeps the device copy of the proving key between proofs.  Default false, as in the ICICLE backend
	// (the device memory is released after each proof); long-lived provers want true.
	PinToGPU bool
	// Precompute is the window-table policy (see PrecomputeAuto for what the default means for un-pinned keys).
	Precompute Precompute
	// StepProfile logs the per-stage device timings of every proof (ICICLE_STEP_PROFILE of the ICICLE backend).
	StepProfile bool
}

// Option configures the MI355X backend.
type Option func(*Config) error

// NewConfig applies the options over the defaults (device 0, nothing pinned, automatic precomputation).
func NewConfig(opts ...Option) (*Config, error) {
	cfg := Config{}
	for _, o := range opts {
		if o == nil {
			continue
		}
		if err := o(&cfg); err != nil {
			return nil, err
		}
	}
	return &cfg, nil
}

// DeviceIDs returns the devices a proof runs on: Devices if set, else DeviceID alone.
func (c *Config) DeviceIDs() []int {
	if len(c.Devices) > 0 {
		return c.Devices
	}
	return []int{c.DeviceID}
}

// WithDeviceID selects the device of a single-GPU proof.
func WithDeviceID(id int) Option {
	return func(c *Config) error {
		if id < 0 {
			return fmt.Errorf("invalid device id %d", id)
		}
		c.DeviceID = id
		return nil
	}
}

// WithDevices proves one statement over several devices of the node.  Ids must be distinct.
func WithDevices(ids ...int) Option {
	return func(c *Config) error {
		if len(ids) == 0 {
			return fmt.Errorf("no device ids provided")
		}
		seen := map[int]bool{}
		for _, id := range ids {
			if id < 0 || seen[id] {
				return fmt.Errorf("invalid or repeated device id %d", id)
			}
			seen[id] = true
		}
		c.Devices = append([]int(nil), ids...)
		return nil
	}
}

// WithProverOptions sets prover options. See [backend.ProverOption] for details.
func WithProverOptions(opts ...backend.ProverOption) Option {
	return func(c *Config) error {
		if len(opts) == 0 {
			return fmt.Errorf("no prover options provided")
		}
		c.ProverOpts = opts
		return nil
	}
}

// WithPinKeysToGPU keeps the proving key (vectors, window tables, commitment keys) in HBM between proofs.
func WithPinKeysToGPU(pin bool) Option {
	return func(c *Config) error {
		c.PinToGPU = pin
		return nil
	}
}

// WithPrecompute selects the window-table policy of the pinned key.
func WithPrecompute(p Precompute) Option {
	return func(c *Config) error {
		if p < PrecomputeNever || p > PrecomputeAlways {
			return fmt.Errorf("invalid precompute policy %d", p)
		}
		c.Precompute = p
		return nil
	}
}

// WithStepProfile logs per-stage device timings.
func WithStepProfile(on bool) Option {
	return func(c *Config) error {
		c.StepProfile = on
		return nil
	}
}
