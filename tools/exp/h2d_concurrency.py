#!/usr/bin/env python3
"""How fast are PAGEABLE host-to-device copies on this box: alone, two at once from two threads, and beside a compute-bound kernel?
(round 6: ga_g16_prove_oneshot uploads the key while the proof runs -- is that overlap real?)"""
import json, threading, time
import torch
dev = torch.device("cuda", 0)
GiB = 1 << 30
src = [torch.empty(GiB, dtype=torch.uint8).random_(0, 255) for _ in range(2)]
dst = [torch.empty(GiB, dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(3)]
a = torch.randn(8192, 8192, device=dev)


def copy(k, reps, out):
    with torch.cuda.stream(streams[k]):
        t0 = time.perf_counter()
        for _ in range(reps):
            dst[k].copy_(src[k], non_blocking=True)
            streams[k].synchronize()
        out[k] = (time.perf_counter() - t0) / reps


def busy(seconds, flag):
    with torch.cuda.stream(streams[2]):
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                b = a @ a
            streams[2].synchronize()
            n += 20
        flag.append(n / (time.perf_counter() - t0))


res = {}
o = [0, 0]
copy(0, 2, o)
copy(0, 5, o)
res["one_copy_GB_per_s"] = round(GiB / o[0] / 1e9, 1)
th = [threading.Thread(target=copy, args=(k, 5, o)) for k in range(2)]
t0 = time.perf_counter()
[t.start() for t in th]
[t.join() for t in th]
res["two_copies_each_GB_per_s"] = [round(GiB / x / 1e9, 1) for x in o]
res["two_copies_total_GB_per_s"] = round(10 * GiB / (time.perf_counter() - t0) / 1e9, 1)
f = []
busy(1.0, f)
res["matmuls_per_s_alone"] = round(f[0], 1)
f = []
tb = threading.Thread(target=busy, args=(2.0, f))
tb.start()
time.sleep(0.3)
copy(0, 5, o)
tb.join()
res["one_copy_beside_matmuls_GB_per_s"] = round(GiB / o[0] / 1e9, 1)
res["matmuls_per_s_beside_copy"] = round(f[0], 1)
print(json.dumps(res))
