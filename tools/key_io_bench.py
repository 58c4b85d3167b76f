#!/usr/bin/env python3
"""Loading a Groth16 proving key from a FILE into HBM (gnark's WriteTo / WriteRawTo / WriteDump layouts read by
ga_g16_pk_read_fd; compressed points are decoded by the device -- one square root each), at BASELINE size.

  python tools/key_io_bench.py --log-n 24 [--dir /dev/shm]

Per format: file size, seconds to write (the library's writer, host arrays -> file), seconds to read + pin WITHOUT window tables
(precompute = -1: the decode + upload cost alone) and with them, and whether a proof on the loaded key equals the proof on the key
pinned from the host arrays.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bn254")
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--dir", default="/dev/shm")
    args = ap.parse_args()
    from gnark_amd import groth16, synth
    from gnark_amd.device import Context
    ctx = Context(0)
    inst = synth.make_instance(ctx, args.curve, args.log_n, 0xF11E, want_dlogs=False)
    ref = inst.proving_key(ctx, precompute=-1)
    want = groth16.Prove(ref, inst.solution, inst.nb_public, inst.r, inst.s).raw()
    ref.FreeGPUResources()
    for fmt, name in ((groth16.KEY_FORMAT_COMPRESSED, "WriteTo (compressed)"), (groth16.KEY_FORMAT_RAW, "WriteRawTo"), (groth16.KEY_FORMAT_DUMP, "WriteDump")):
        path = os.path.join(args.dir, "ga_key_%d.bin" % fmt)
        out = {"curve": args.curve, "log_n": args.log_n, "format": name}
        try:
            t0 = time.perf_counter()
            with open(path, "wb") as f:
                size = groth16.WriteKey(ctx, args.curve, f, fmt, domain_cardinality=inst.n, **inst.key)
            out["file_gib"] = round(size / 2**30, 3)
            out["write_s"] = round(time.perf_counter() - t0, 2)
            for pre, tag in ((-1, "read_pin_plain_s"), (1, "read_pin_tables_s")):
                t0 = time.perf_counter()
                with open(path, "rb") as f:
                    pk = groth16.ProvingKey.ReadFrom(ctx, args.curve, f, precompute=pre)
                ctx.sync()
                out[tag] = round(time.perf_counter() - t0, 2)
                if pre == -1:
                    out["read_gib_per_s"] = round(size / 2**30 / out[tag], 2)
                    out["proof_equal"] = bool(np.array_equal(groth16.Prove(pk, inst.solution, inst.nb_public, inst.r, inst.s).raw(), want))
                pk.FreeGPUResources()
        finally:
            if os.path.exists(path):
                os.unlink(path)
        print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
