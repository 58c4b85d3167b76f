#!/usr/bin/env python3
"""What clock and power does the GPU hold while a kernel family runs?  (round 4: the bucket kernels sit ~20 % under the sum of their
instruction costs at the microbenchmarked rates; the microbenchmarks are milliseconds long, the kernels run for seconds of a proof
stream -- if the chip power-throttles under sustained integer load, the issue bound has to be priced at the clock it actually holds.)

Samples the amdgpu sysfs sensors (hwmon freq1_input = current shader clock, power1_average / power1_input) -- and `rocm-smi` as a
fallback -- every few milliseconds from a thread while the main thread runs, back to back for --seconds each:
  idle, v_mad microbench (ga_microbench), BN254 G1 table MSM 2^24, BN254 G2, BLS12-381 G1, BLS12-381 G2, computeH 2^24.
Prints one JSON line: per workload the mean / min / max clock in MHz, mean power in W, and the kernel times it saw."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "power1_average", "power1_input", "temp1_input", "freq2_input"):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out.setdefault(name, []).append(p)
    return out


def read_int(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return None


class Sampler(threading.Thread):
    """sysfs sensors every `period` seconds, and between them ga_clock_probe: one wave on its own stream comparing the shader cycle
    counter with the constant-rate one for 2 ms -- the clock as the kernels see it"""

    def __init__(self, paths, ctx, period=0.005):
        super().__init__(daemon=True)
        self.paths, self.ctx, self.period, self.rows, self.stop = paths, ctx, period, [], False

    def run(self):
        import ctypes as C
        mhz = C.c_double()
        while not self.stop:
            row = {k: read_int(v[0]) for k, v in self.paths.items()}
            row["t"] = time.perf_counter()
            if self.ctx.lib.ga_clock_probe(self.ctx.handle, 2000, C.byref(mhz)) == 0:
                row["probe_mhz"] = mhz.value
            self.rows.append(row)
            time.sleep(self.period)


def smi_clock():
    try:
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(r.stdout)
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)[:200]}


def summarize(rows):
    res = {"samples": len(rows)}
    f = [r["freq1_input"] for r in rows if r.get("freq1_input")]
    if f:
        res.update(sclk_mhz_mean=round(sum(f) / len(f) / 1e6, 1), sclk_mhz_min=round(min(f) / 1e6, 1), sclk_mhz_max=round(max(f) / 1e6, 1))
    pm = [r["probe_mhz"] for r in rows if r.get("probe_mhz")]
    if pm:
        res.update(probe_mhz_mean=round(sum(pm) / len(pm), 1), probe_mhz_min=round(min(pm), 1), probe_mhz_max=round(max(pm), 1))
    for k in ("power1_average", "power1_input"):
        p = [r[k] for r in rows if r.get(k)]
        if p:
            res[k + "_w_mean"] = round(sum(p) / len(p) / 1e6, 1)
            res[k + "_w_max"] = round(max(p) / 1e6, 1)
    t = [r["temp1_input"] for r in rows if r.get("temp1_input")]
    if t:
        res["temp_c_max"] = round(max(t) / 1e3, 1)
    return res


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--log-n", type=int, default=24)
    args = ap.parse_args()
    import gnark_amd
    from gnark_amd import _lib, ecc, fft
    ctx = gnark_amd.Context(0)
    lib = ctx.lib
    n = 1 << args.log_n
    paths = sensors()
    out = {"sensors": {k: v[0] for k, v in paths.items()}, "rocm_smi_idle": smi_clock()}

    def measure(name, body):
        body()   # warm-up
        ctx.sync()
        s = Sampler(paths, ctx)
        s.start()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < args.seconds:
            body()
            reps += 1
        ctx.sync()
        el = time.perf_counter() - t0
        s.stop = True
        s.join()
        res = summarize(s.rows[len(s.rows) // 4:])   # (the first quarter: clocks still settling)
        res.update(reps=reps, ms_per_rep=round(el * 1e3 / reps, 3))
        out[name] = res

    measure("idle", lambda: time.sleep(0.2))
    measure("microbench", lambda: ctx.microbench())
    for cid, cname in ((0, "bn254"), (1, "bls12-381")):
        for group, gname in ((_lib.G1, "g1"), (_lib.G2, "g2")):
            words = gnark_amd.device.affine_words(cid, group)
            bases = ctx.malloc(n * words * 8)
            scal = ctx.malloc(n * 32)
            lib.check(lib.ga_gen_bases(ctx.handle, cid, group, 0x5EED0002, n, bases.ptr, None))
            lib.check(lib.ga_gen_scalars(ctx.handle, cid, 0x5EED0001, n, scal.ptr))
            table = ecc.PrecomputedBases(ctx, cid, group, bases, n=n)
            bases.free()
            ctx.profile(True)
            ctx.profile_reset()
            measure("msm_%s_%s" % (cname, gname), lambda: table.MultiExp(scal))
            st = {}
            for k, ms in ctx.profile_read():
                a = st.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += ms
            ctx.profile(False)
            out["msm_%s_%s" % (cname, gname)]["accumulate_ms"] = round(st["msm_accumulate"][1] / st["msm_accumulate"][0], 3)
            table.free()
            scal.free()
    d = fft.Domain(ctx, 0, n)
    a, b, c = (ctx.malloc(n * 32) for _ in range(3))
    for k, buf in enumerate((a, b, c)):
        lib.check(lib.ga_gen_scalars(ctx.handle, 0, 0x1000 + k, n, buf.ptr))
    measure("compute_h_bn254", lambda: lib.check(lib.ga_compute_h(d.handle, a.ptr, b.ptr, c.ptr, n, a.ptr, 1)))
    out["rocm_smi_after"] = smi_clock()
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
