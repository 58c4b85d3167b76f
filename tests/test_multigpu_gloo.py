"""N>1 path on CPU: 2 processes, `gloo`, both MSM partitionings on the emulation build, checked against the
oracle inside each worker.  On the GPU box the same gnark_amd.multigpu functions run over RCCL (bench.py --gpus N)."""
import os
import subprocess
import sys

from gnark_amd import multigpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 16, 301):
        for world in (1, 2, 3, 8):
            parts = [multigpu.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 3, 5])
def test_gloo_ranks_msm_and_groth16_sharding(emu_lib, world):
    """world 2: rank 0 runs the a and c chains of computeH, rank 1 the b chain; world 3: one chain per rank; world 5: two ranks own no
    chain but still upload their fifth of A, B, C (sliced uploads gathered on the chain owners).  Both MSM
    partitionings, the sharded Groth16 proof (wire-range upload, chains sent to rank 0, h slices scattered) against the oracle's
    proof bytes, and the round-1 replicate-h scheme for comparison."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29731 + world), os.path.join(ROOT, "tests", "_mgpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "MGPU_OK world=%d" % world in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
