"""gnark_amd -- MI355X-native accelerated prover backend for gnark (Groth16 / PLONK hot path).

The product is `libgnark_amd.so` (hand-written HIP for gfx950 behind the C ABI in include/gnark_amd.h); this package
is the thin host-side mirror of the reference's interfaces for that path:
  gnark_amd.ecc      MultiExp                      (gnark-crypto ecc, prove.go:194-283)
  gnark_amd.fft      Domain.FFT / FFTInverse       (gnark-crypto fft, prove.go:346-389)
  gnark_amd.groth16  ProvingKey / Prove / Proof    (backend/accelerated/icicle/groth16)
  gnark_amd.plonk    ComputeQuotient / BuildRatioCopyConstraint (backend/plonk/bn254/prove.go:645-655,841-1123,1287-1350)
There is no CPU fallback anywhere in this package.
"""
from . import _lib, device, ecc, fft, groth16, plonk  # noqa: F401
from ._lib import GnarkAmdError, load  # noqa: F401
from .device import Context, DeviceBuffer  # noqa: F401

__all__ = ["Context", "DeviceBuffer", "GnarkAmdError", "ecc", "fft", "groth16", "load", "plonk"]
