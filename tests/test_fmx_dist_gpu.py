"""The N-process freemuxlet path of bench.py (--config 3) on a 1-GPU box: two and three real processes
(torch.distributed.run), each with its own muxgl handle on device 0 holding the rank's row and column slabs, exchange the
cluster-GP rows and the assignments through torch.distributed -- one in-place all_gather_into_tensor per exchange on the
library's own device buffers (gloo stages them here; on a multi-GPU node the same calls run over RCCL, ordered against
the library's stream) -- and must reproduce the single-process, whole-pileup run bit for bit."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "bench.py")
SHAPE = ["--config", "3", "--no-cpu-baseline", "--cells", "400", "--snps", "3000", "--clusters", "5", "--mean-entries", "200",
         "--steps", "4", "--warmup", "0"]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("world", [2, 3])
def test_two_processes_match_one(tmp_path, world):
    one = str(tmp_path / "one.npz")
    many = str(tmp_path / "many.npz")
    j1 = run([sys.executable, PROBE, "--gpus", "1", "--dump", one] + SHAPE)
    jn = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
              "127.0.0.1", "--master-port", str(free_port()), PROBE, "--gpus", str(world), "--dist-backend", "gloo",
              "--single-device", "--dump", many] + SHAPE)
    assert j1["n_gpus"] == 1 and jn["n_gpus"] == world and jn["scaling"] == "strong"
    a, b = np.load(one), np.load(many)
    assert np.array_equal(a["hist"], b["hist"])
    assert a["cells"].tobytes() == b["cells"].tobytes()
    assert (a["cells"]["type"] == 0).sum() > 200


def test_single_rank_over_rccl(tmp_path):
    """What a 1-GPU box can run of the RCCL path: a process group of ONE rank on backend nccl, the rank's slabs (here the
    whole pileup, twice), the in-place all_gather_into_tensor / all_reduce on the library's device buffers, the library's
    stream as torch's current stream (ExternalStream), asynchronous phases, the counters fetched behind the M-step --
    everything the N-GPU run does except moving bytes between devices.  Same records as the plain one-handle run."""
    one = str(tmp_path / "one.npz")
    rccl = str(tmp_path / "rccl.npz")
    run([sys.executable, PROBE, "--gpus", "1", "--dump", one] + SHAPE)
    j = run([sys.executable, PROBE, "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--dump", rccl] + SHAPE)
    assert j["config"]["backend"] == "nccl" and j["n_gpus"] == 1
    ex = j["exchange_ms_rank0"]  # the exchanges are timed with device events (what DESIGN 4.3's model is compared with)
    assert ex and ex["cluster_gp_allgather"] >= 0.0 and ex["assignments_allgather_and_counters"] >= 0.0
    a, b = np.load(one), np.load(rccl)
    assert np.array_equal(a["hist"], b["hist"])
    assert a["cells"].tobytes() == b["cells"].tobytes()
