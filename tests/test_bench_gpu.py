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


def run(cmd, timeout=900):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
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
    assert rf["bound"] in ("hbm", "fp64_valu") and rf["kernel"] == "demux_oct_kernel"
    top = rf["hbm"] if rf["bound"] == "hbm" else rf["fp64"]
    assert rf["unit"] == top["unit"] and rf["peak"] == top["peak"] and rf["frac"] == top["frac"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["hbm"]["peak"] == 8000.0 and rf["hbm"]["unit"] == "GB/s" and rf["fp64"]["peak"] == 78.6
    for part in (rf, rf["hbm"], rf["fp64"]):  # no fraction above 1 anywhere in the line
        assert part["frac"] is None or 0.0 < part["frac"] <= 1.0
    assert 0.1 < rf["fp64"]["frac"] < 1.0 and 0.1 < rf["fp64"]["issued_over_reference"] < 1.0
    assert rf["traffic"] is None or (rf["traffic"] > 1e8 and rf["hbm"]["frac"] is not None)
    assert rf["traffic"] == rf["hbm"]["traffic_bytes"]
    # the recomputable floor of the algorithm as built and the fraction of it the kernel reaches
    fl = rf["floor"]
    assert fl["hypotheses_per_entry"] == 136 and 0.6 < fl["linear_entry_share"] < 0.85
    assert 0 < fl["hbm_ms"] < fl["valu_ms"] < rf["kernel_ms"] and 0.2 < fl["frac_of_floor"] <= 1.0
    assert abs(fl["frac_of_floor"] - max(fl["valu_ms"], fl["hbm_ms"]) / rf["kernel_ms"]) < 1e-12
    assert "sensitivity" not in d and "note_8d" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert cb["parity_max_abs_ll_diff"] < 1e-5
    # the checked records equal the oracle's in every integer field after the product's exact-call pass; the pass's cost
    # over ALL cells of the step is stated next to the timed value, not inside it
    assert set(cb["parity_excuses_used"].values()) == {0} and cb["exact_pass_on_checked_cells"]["cells"] > 0
    ep = d["exact_calls_pass"]
    assert ep["timed_in_value"] is False and ep["ms_all_cells"] > 0 and ep["cells"] > 9000 and ep["deep"] < 20
    assert d["pair_order_ms_all_cells"] == ep["ms_all_cells"]
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
    assert fx["kernel_ms_rank0_last_iteration"]["estep"] > 0 and fx["roofline"]["kernel"] == "fmx_estep_oct_kernel"
    assert fx["roofline"]["frac"] is not None and 0 < fx["roofline"]["frac"] <= 1.0
    # the roofline prices the sweep kernel alone, inside the E-step bracket
    km = fx["kernel_ms_rank0_last_iteration"]
    assert 0 < km["estep_sweep"] <= km["estep"] and abs(fx["roofline"]["kernel_ms"] - km["estep_sweep"]) < 1e-6
    assert fx["cpu_baseline"]["kind"] == "port" and fx["cpu_baseline"]["cores"] == 1 and fx["cpu_baseline"]["value"] > 0
    # ... which doubles as a parity check: sampled cells' E-step / scans / re-assignment against the device's own cluster pileups
    assert fx["cpu_baseline"]["parity_checked_cells"] >= 16 and fx["cpu_baseline"]["parity_max_abs_ll_diff"] < 1e-5
    assert d["ramp"]["untimed_passes"] > 0
    # the north_star shapes, in the same line: configs[2] (demuxlet 100 k x 64 x 200 k, six alphas) ...
    d2 = d["demuxlet_config2"]
    assert "error" not in d2, d2
    assert d2["config"]["cells_per_gpu"] == 100000 and d2["config"]["samples"] == 64 and d2["config"]["snps"] == 200000
    assert d2["config"]["alphas"] == [0.0, 0.1, 0.2, 0.3, 0.4, 0.5] and d2["steps"] == 3 and d2["scaling"] == "weak"
    assert d2["config"]["entries_per_gpu"] > 90_000_000 and d2["roofline"]["kernel"] == "demux_ring_lin_kernel"
    assert abs(d2["value"] - 100000 * (64 + 64 * 63 * 5) / (d2["ms_per_step"] * 1e-3)) / d2["value"] < 1e-9
    assert d2["ms_per_step"] < 60e3  # the north_star's "< 60 s" with room to spare
    assert 0 < d2["roofline"]["frac"] <= 1.0 and d2["cpu_baseline"]["parity_max_abs_ll_diff"] < 1e-5
    assert d2["roofline"]["floor"]["hypotheses_per_entry"] == 18208 and 0.3 < d2["roofline"]["floor"]["frac_of_floor"] <= 1.0
    assert 0.2 < fx["roofline"]["floor"]["frac_of_floor"] <= 1.0 and fx["roofline"]["floor"]["hypotheses_per_entry"] == 136
    # one key, one meaning: in EVERY leg roofline.frac = achieved / peak of the binding roof from issued FP64 flops or counter
    # bytes (fp64.source names counters or model); the fraction of the algorithm's own floor is roofline.floor.frac_of_floor
    for leg in (d, d2, fx, d["freemuxlet_config4"]):
        rl = leg["roofline"]
        assert rl["bound"] in ("fp64_valu", "hbm") and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12
        top = rl["fp64"] if rl["bound"] == "fp64_valu" else rl["hbm"]
        assert rl["frac"] == top["frac"] and rl["achieved"] == top["achieved"] and "frac_is" not in rl
        assert rl["fp64"]["source"].startswith(("pmc", "model")) and 0 < rl["floor"]["frac_of_floor"] <= 1.0
        assert rl["traffic"] is None or "traffic_over_compulsory" not in rl["hbm"] or rl["hbm"]["traffic_over_compulsory"] > 1.0
    # ... and configs[4] (freemuxlet 500 k x 500 k, K = 64)
    f4 = d["freemuxlet_config4"]
    assert "error" not in f4, f4
    assert f4["config"]["cells"] == 500000 and f4["config"]["snps"] == 500000 and f4["config"]["clusters"] == 64
    assert f4["config"]["entries"] > 450_000_000 and f4["steps"] == 2 and f4["scaling"] == "strong"
    assert f4["roofline"]["kernel"] == "fmx_estep_wave_kernel" and 0 < f4["roofline"]["frac"] <= 1.0
    assert f4["cpu_baseline"]["value"] > 0 and f4["cpu_baseline"]["parity_max_abs_ll_diff"] < 1e-5


def test_two_rank_launch_line():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
             "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "30", "--warmup", "3",
             "--ramp-seconds", "0.1", "--dist-backend", "gloo", "--single-device", "--fmx-leg-steps", "3", "--legs", "3"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    fx = d["freemuxlet_em"]  # two ranks, slabs, one in-place all-gather per exchange (gloo staging on this box)
    assert "error" not in fx, fx
    assert fx["n_gpus"] == 2 and fx["steps"] == 3 and fx["scaling"] == "strong" and fx["config"]["backend"] == "gloo"
    # whole-job aggregate: both ranks' cells
    assert abs(d["value"] - 2 * 10000 * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    # a line of an N > 1 run explains itself: weak scaling with the per-rank rate; the freemuxlet leg carries the
    # model's prediction next to the measurement and the exchange times
    assert "weak scaling" in d["scaling_note"] and abs(d["per_rank_value"] * 2 - d["value"]) / d["value"] < 1e-12
    sm = fx["scaling_model"]
    assert sm["kernels_ms_rank0_last_iteration"] > 0 and sm["measured_ms_per_iteration"] == fx["ms_per_step"]
    assert sm["predicted_ms_per_iteration"]["direct"] <= sm["predicted_ms_per_iteration"]["ring"]
    assert "exchange_ms_rank0" in fx


def test_eight_ranks_line_explains_itself():
    """what the driver's first 8-GPU run will print, as far as a 1-GPU box can show it: eight real processes (gloo staging,
    every rank on device 0), a hundredth of the cells, the legs that have an exchange step (configs[3], [4]) and the
    strong-scaling demuxlet leg.  Every N > 1 line names the ranks that answered an all-reduce, the per-rank rates, the
    exchange times and the scaling model next to the measurement."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr",
             "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "8", "--steps", "5", "--warmup", "1",
             "--ramp-seconds", "0", "--dist-backend", "gloo", "--single-device", "--scale", "0.01", "--fmx-leg-steps", "2",
             "--legs", "3,2s,4", "--no-cpu-baseline"], timeout=900)
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["dist"]["world_size"] == 8
    assert len(d["per_rank"]["sweep_kernel_ms"]) == 8 and min(d["per_rank"]["sweep_kernel_ms"]) > 0
    for key in ("freemuxlet_em", "freemuxlet_config4"):
        fx = d[key]
        assert "error" not in fx, fx
        assert fx["n_gpus"] == 8 and fx["scaling"] == "strong" and fx["ranks_seen"] == 8
        assert "exchange_ms_rank0" in fx and "scaling_model" in fx
        assert len(fx["per_rank"]["ms_per_iteration"]) == 8 and min(fx["per_rank"]["ms_per_iteration"]) > 0
        assert fx["scaling_model"]["measured_ms_per_iteration"] == fx["ms_per_step"]
    st = d["demuxlet_config2_strong"]
    assert "error" not in st, st
    assert st["scaling"] == "strong" and st["n_gpus"] == 8 and "ONE job" in st["scaling_note"]
    ents = st["per_rank"]["entries"]
    assert len(ents) == 8 and max(ents) < 1.2 * min(ents)       # ranges balanced by entries
    # the job's LLs, not N x a shard's: 1000 cells x (64 + 64 x 63 x 5) hypotheses per pass
    assert abs(st["value"] - 1000 * 20224 / (st["ms_per_step"] * 1e-3)) / st["value"] < 1e-9


def test_dense_pileup_sensitivity_line():
    """`--dense`: the same shape with 1 + Poisson(2) reads per entry, and once more without the linear-entry form"""
    base = [sys.executable, BENCH, "--steps", "30", "--warmup", "5", "--ramp-seconds", "0.2", "--no-legs", "--no-cpu-baseline",
            "--dense"]
    a = run(base)
    b = run(base + ["--no-linear"])
    for d in (a, b):
        assert d["sensitivity"]["reads_lambda"] == 2.0 and 2.5 < d["sensitivity"]["reads_per_entry"] < 3.5
        assert 0.08 < d["sensitivity"]["linear_entry_share"] < 0.25  # P(one read) = exp(-2) = 0.135
    assert a["sensitivity"]["linear_entry_form"] and not b["sensitivity"]["linear_entry_form"]
    assert b["ms_per_step"] > a["ms_per_step"]  # the linear class still pays on one entry in seven


def test_plain_shell_gpus_n_launches_itself():
    """`python bench.py --gpus 2` from a plain shell (the form the driver uses) starts its own two ranks"""
    d = run([sys.executable, BENCH, "--gpus", "2", "--steps", "10", "--warmup", "2", "--ramp-seconds", "0.1",
             "--dist-backend", "gloo", "--single-device", "--fmx-leg-steps", "2", "--legs", "3", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["freemuxlet_em"]["n_gpus"] == 2
    assert "error" not in d["freemuxlet_em"], d["freemuxlet_em"]
