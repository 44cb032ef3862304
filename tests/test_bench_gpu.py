"""bench.py's one-line JSON contract on a GPU box (short run): the keys the driver reads, the roofline and cpu_baseline
objects, and that the N > 1 launch path (torch.distributed.run, here two ranks on one device over gloo) prints one line
from rank 0 with the whole-job aggregate."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_single_gpu_line():
    d = run([sys.executable, BENCH, "--gpus", "1", "--steps", "50", "--warmup", "5", "--ramp-seconds", "0.2"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 50 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    # two roofs are priced, "bound" names the larger fraction and the top-level numbers are that roof's
    assert rf["bound"] in ("hbm", "fp64_valu") and rf["kernel"] == "demux_quad_kernel"
    top = rf["hbm"] if rf["bound"] == "hbm" else rf["fp64"]
    assert rf["unit"] == top["unit"] and rf["peak"] == top["peak"] and rf["frac"] == top["frac"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["hbm"]["peak"] == 8000.0 and rf["hbm"]["unit"] == "GB/s" and rf["fp64"]["peak"] == 78.6
    for part in (rf, rf["hbm"], rf["fp64"]):  # no fraction above 1 anywhere in the line
        assert part["frac"] is None or 0.0 < part["frac"] <= 1.0
    assert 0.1 < rf["fp64"]["frac"] < 1.0 and 0.1 < rf["fp64"]["issued_over_reference"] < 1.0
    assert rf["traffic"] is None or (rf["traffic"] > 1e8 and rf["hbm"]["frac"] is not None)
    assert rf["traffic"] == rf["hbm"]["traffic_bytes"]
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert cb["parity_max_abs_ll_diff"] < 1e-5
    # value = LLs of the job / step time
    assert abs(d["value"] - 10000 * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    assert d["value"] > 50 * cb["value"]
    st = cb["single_thread"]
    assert st["cores"] == 1 and 0 < st["value"] <= cb["value"] * 1.5 and cb["speedup_over_single_thread"] > 0.5
    # the secondary leg: freemuxlet EM of configs[3] (20 iterations) on the same rank(s)
    fx = d["freemuxlet_em"]
    assert "error" not in fx, fx
    assert fx["scaling"] == "strong" and fx["steps"] == 20 and fx["n_gpus"] == 1 and fx["config"]["clusters"] == 16
    assert fx["config"]["cells"] == 50000 and fx["config"]["snps"] == 100000
    assert abs(fx["value"] - 50000 * 136 / (fx["ms_per_step"] * 1e-3)) / fx["value"] < 1e-9
    assert fx["kernel_ms_rank0_last_iteration"]["estep"] > 0 and fx["roofline"]["kernel"] == "fmx_estep_quad_kernel"
    assert fx["roofline"]["frac"] is not None and 0 < fx["roofline"]["frac"] <= 1.0


def test_two_rank_launch_line():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
             "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "30", "--warmup", "3",
             "--ramp-seconds", "0.1", "--dist-backend", "gloo", "--single-device", "--fmx-leg-steps", "3"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    fx = d["freemuxlet_em"]  # two ranks, slabs, one in-place all-gather per exchange (gloo staging on this box)
    assert "error" not in fx, fx
    assert fx["n_gpus"] == 2 and fx["steps"] == 3 and fx["scaling"] == "strong" and fx["config"]["backend"] == "gloo"
    # whole-job aggregate: both ranks' cells
    assert abs(d["value"] - 2 * 10000 * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
