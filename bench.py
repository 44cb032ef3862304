#!/usr/bin/env python
"""Benchmark of the demuxlet hot path on MI355X -- BASELINE.json's metric on BASELINE.json's configs[1].

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path (entry likelihoods + sample-pair sweep + evidence/scan/call + per-cell records
to the host) over one batch: the whole synthetic pileup of the workload, already resident in HBM.  One process per
GPU; every rank owns its own 10k-cell shard (weak scaling; demuxlet's cells are independent, so there is no data-path
collective -- torch.distributed/RCCL only provides the barrier and the max-over-ranks of the elapsed time).

Prints ONE JSON line (rank 0).  LL = one hypothesis log-likelihood feeding a call: per cell V singlets +
V(V-1)(A-1) ordered doublets.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from popscle_amd import muxgl, synth  # noqa: E402

METRIC = "cell-sample-pair LLs/sec (singlet+doublet), demuxlet 10k cells×16 samples"
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # vector FP64 = half the 157.3 TF FP32 vector rate (same guide); FP64 MFMA runs at the same rate


def algorithmic_bytes_per_entry(V: int, reads_per_entry: float) -> float:
    """12 B (SNP id + read offset) + reads + 24*V (GP row gather)"""
    return 12.0 + reads_per_entry + 24.0 * V


def algorithmic_flops_per_entry(V: int, A: int, reads_per_entry: float) -> float:
    """A*(18V + 7V^2) + reads*A*27"""
    return A * (18.0 * V + 7.0 * V * V) + reads_per_entry * A * 27.0


def sweep_kernel_name(V, alphas):
    """the kernel libmuxgl dispatches for this shape (popscle_amd/csrc/demux_kernels.hip: demux_launch)"""
    if V <= 16 and tuple(alphas) == (0.0, 0.5):
        return "demux_quad_kernel"
    if V <= 16 and sum(1 for a in alphas[1:] if a != 0.5) <= 5 and sum(1 for a in alphas[1:] if a == 0.5) <= 1:
        return "demux_row_kernel"
    if V <= 24 and tuple(alphas) == (0.0, 0.5):
        return "demux_rowx_kernel"
    if V <= 32 and len(alphas) == 2 and alphas[1] == 0.5 and alphas[0] != 0.5:
        return "demux_row2_kernel"
    if V <= 32:
        return "demux_wave32_kernel"
    if V <= 255:
        return "demux_wave_kernel"
    return "demux_sweep_kernel"


def cpu_baseline(p, alphas, gpu_cells, budget_s=15.0):
    """The CPU oracle (the restatement of the reference's loop, kind "port") timed on this host's cores on a bounded
    sample of the same workload.  Also used as a last parity check of the GPU records for the sampled cells."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import parity

    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    V = p.gp.shape[1]
    rng = np.random.default_rng(0)
    probe = np.sort(rng.choice(p.C, min(p.C, 2 * threads), replace=False))
    sub = p.subset_cells(probe)
    t0 = time.perf_counter()
    ob.demux(sub, alphas=alphas, nthreads=threads)
    dt = max(time.perf_counter() - t0, 1e-6)
    n = int(min(p.C, max(len(probe), budget_s / dt * len(probe))))
    pick = np.sort(rng.choice(p.C, n, replace=False))
    sub = p.subset_cells(pick)
    t0 = time.perf_counter()
    want = ob.demux(sub, alphas=alphas, nthreads=threads)
    dt = time.perf_counter() - t0
    lls = n * (V + V * (V - 1) * (len(alphas) - 1))
    rep = parity.compare_demux(gpu_cells[pick], want, alphas)
    return {
        "value": lls / dt, "unit": "LLs/s", "cores": threads, "kind": "port",
        "sample": f"{n} of {p.C} cells of the same workload ({int(sub.nnz)} entries), oracle/muxgl_oracle.c with "
                  f"{threads} OpenMP threads over cells, {dt:.1f} s",
        "entries_per_s": sub.nnz / dt,
        "parity_checked_cells": rep["cells"], "parity_max_abs_ll_diff": rep["max_abs_ll_diff"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--ramp-seconds", type=float, default=1.0,
                    help="untimed passes before the W warmup steps, until the engine clock has ramped up (a 0.6 ms step "
                         "repeated 20 times runs ~10 %% below the sustained clock)")
    ap.add_argument("--config", type=int, default=1, help="index into BASELINE.json configs (1 or 2)")
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the config's cells (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for tests)")
    ap.add_argument("--single-device", action="store_true",
                    help="functional test of the N>1 path on a 1-GPU box: every rank uses device 0 (use with gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dev = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.dist_backend)
    tdev = "cuda" if args.dist_backend == "nccl" else "cpu"

    cfg = synth.CONFIGS[args.config]
    alphas = tuple(cfg["alphas"])
    V, A = cfg["V"], len(alphas)
    # weak scaling: every rank owns a full config-sized shard of cells (its own seed), GP tensor replicated
    C = max(1, int(round(cfg["C"] * args.scale)))
    p = synth.make_pileup(C, cfg["S"], V, seed=synth.BASE_SEED + args.config + 1000 * rank,
                          donor_seed=synth.BASE_SEED + args.config)

    eng = muxgl.Engine(dev)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.demux_set_gp(p.gp, p.has_gp)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        eng.demux_run(alphas, 0.5, want_cells=False)
    for _ in range(args.warmup):
        eng.demux_run(alphas, 0.5, want_cells=False)
    barrier()
    kern_ms = np.zeros(muxgl.T_COUNT)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.demux_run(alphas, 0.5, want_cells=False)
        kern_ms += eng.timing()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(p.C), float(p.nnz)], dtype=torch.float64, device=tdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_cells, total_entries = float(tot[0].item()), float(tot[1].item())
    else:
        total_cells, total_entries = float(p.C), float(p.nnz)

    if rank == 0:
        lls_per_cell = V + V * (V - 1) * (A - 1)
        step_s = elapsed / args.steps
        value = total_cells * lls_per_cell / step_s
        kern_ms /= args.steps
        rpe = p.R / max(p.nnz, 1)
        sweep_s = kern_ms[muxgl.T_DEMUX_SWEEP] * 1e-3
        abytes = algorithmic_bytes_per_entry(V, rpe) * p.nnz
        aflops = algorithmic_flops_per_entry(V, A, rpe) * p.nnz
        achieved = abytes / sweep_s / 1e9 if sweep_s > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(f"config{args.config}", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": METRIC, "value": value, "unit": "LLs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"demuxlet synthetic PLP (BASELINE.json configs[{args.config}]): {C} cells x {V} samples x "
                            f"{cfg['S']} SNPs per GPU, alpha grid {list(alphas)}, {p.nnz} entries, {p.R} reads",
                "cells_per_gpu": C, "samples": V, "snps": cfg["S"], "alphas": list(alphas),
                "entries_per_gpu": int(p.nnz), "lls_per_cell": lls_per_cell, "parallelism": f"cells sharded x{world}",
            },
            "entries_per_s": total_entries / step_s,
            "cells_per_s": total_cells / step_s,
            # quad path: "reduce" is the fused finish kernel (chunk reduction + call + records written to pinned host
            # memory), "call" and "d2h" are then 0; other paths run them as separate launches
            "kernel_ms": {"sweep": float(kern_ms[muxgl.T_DEMUX_SWEEP]), "reduce": float(kern_ms[muxgl.T_DEMUX_REDUCE]),
                          "call": float(kern_ms[muxgl.T_DEMUX_CALL]),
                          "d2h": float(kern_ms[muxgl.T_DEMUX_D2H])},
            "roofline": {
                "bound": "hbm", "kernel": sweep_kernel_name(V, alphas), "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": abytes,
                "fp64_valu": {"achieved": aflops / sweep_s / 1e12 if sweep_s > 0 else 0.0, "peak": FP64_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": (aflops / sweep_s / 1e12 / FP64_PEAK_TFLOPS) if sweep_s > 0 else 0.0},
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(p, alphas, eng.demux_results_view().copy())
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
