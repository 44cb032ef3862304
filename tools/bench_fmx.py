#!/usr/bin/env python
"""freemuxlet EM on N GPUs (BASELINE.json configs[3]: --nsample 16, 50k cells x 100k SNPs, 20 EM iterations), one
process per GPU -- the secondary benchmark next to bench.py (which keeps BASELINE.json's headline metric).

    python tools/bench_fmx.py --gpus 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_fmx.py --gpus N [--iters 20] [--config 3] [--scale 1.0]

Strong scaling: the job is fixed, every rank holds the whole packed pileup and entry likelihoods; the E-step / scans are
sharded by cells, the cluster-GP rows and the ordered M-step by SNPs (popscle_amd/freemuxlet.py).  Per iteration two
all-gathers (RCCL broadcasts on the library's own device buffers) and one small all-reduce.  Prints one JSON line.

--dist-backend gloo --single-device runs the same N-process path on a 1-GPU box (all ranks on device 0, collectives
staged by gloo): used by tests/test_fmx_dist_gpu.py; --dump writes rank 0's final records for comparison."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popscle_amd import freemuxlet, muxgl, synth  # noqa: E402


def cpu_baseline(p, K, clust0, budget_s=10.0):
    """One EM iteration of the CPU oracle (restatement of cmd_cram_freemux2.cpp:375-597, kind "port") on a bounded
    sample of the cells, single-threaded like the reference.  The oracle is only timed here."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    def run(n):
        cells = np.sort(np.random.default_rng(1).choice(p.C, n, replace=False))
        sub = p.subset_cells(cells)
        e = ob.fmx_entry_pileup(sub)
        c0 = np.ascontiguousarray(clust0[cells])
        cplp = ob.fmx_build_cluster_pileup(sub, e, K, c0)
        st = ob.fmx_init_cells(c0)
        t0 = time.perf_counter()
        ob.fmx_iterate(sub, e, K, cplp, st, 0.5, 0.1, nthreads=1)
        return time.perf_counter() - t0, sub.nnz

    n = min(p.C, 50)
    dt, _ = run(n)
    n = int(min(p.C, max(n, budget_s / max(dt, 1e-6) * n)))
    dt, nnz = run(n)
    npairs = K * (K + 1) // 2
    return {"value": n * npairs / dt, "unit": "LLs/s", "cores": 1, "kind": "port",
            "sample": f"one EM iteration over {n} of {p.C} cells ({nnz} entries), oracle/muxgl_oracle.c on one core "
                      f"(the reference is single-threaded), {dt:.1f} s", "entries_per_s": nnz / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--cells", type=int, default=0, help="override the cell count (tests)")
    ap.add_argument("--snps", type=int, default=0, help="override the SNP count (tests)")
    ap.add_argument("--clusters", type=int, default=0, help="override K (tests)")
    ap.add_argument("--mean-entries", type=float, default=800.0)
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--single-device", action="store_true")
    ap.add_argument("--dump", default="")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    cfg = synth.CONFIGS[args.config]
    C = args.cells or max(1, int(round(cfg["C"] * args.scale)))
    S = args.snps or cfg["S"]
    K = args.clusters or cfg["V"]
    # the same job on every rank (same seed): strong scaling
    p = synth.make_pileup(C, S, K, seed=synth.BASE_SEED + args.config, with_gp=False, mean_entries=args.mean_entries,
                          min_entries=min(50, max(1, int(args.mean_entries // 4))))
    eng = muxgl.Engine(dev)
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.fmx_prepare(p.af)
    # a seeded start (--init-cluster style): 90 % of the cells start in their source sample's cluster
    clust0 = np.where(np.random.default_rng(0).random(C) < 0.9, p.truth["s1"], -1).astype(np.int32)
    ex = freemuxlet.TorchExchange(dist, rank, world) if world > 1 else None

    def tensor_on(dev_index):
        return lambda e, which: freemuxlet.engine_exchange_tensor(e, which, dev_index)

    from popscle_amd import shard

    ranges = (shard.cell_shards(p.cell_ptr, world), shard.snp_shards(p.entry_snp, p.S, world))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n, timings=None):
        return freemuxlet.run_em(eng, K, clust0, p.cell_ptr, p.entry_snp, max_iter=n, early_stop=False, exchange=ex,
                                 exchange_tensor=tensor_on(dev), ranges=ranges, timings=timings, sync=barrier)

    if args.warmup:
        run(args.warmup)
    tm = {}
    cells, hist = run(args.iters, tm)
    elapsed = tm["loop_s"]  # exactly args.iters EM iterations, bracketed by barrier + device synchronize on both sides
    setup_s = tm["setup_s"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        npairs = K * (K + 1) // 2
        out = {
            "metric": "freemuxlet EM cell-cluster-pair LLs/sec", "value": C * npairs * args.iters / elapsed,
            "unit": "LLs/s", "n_gpus": world, "steps": args.iters, "warmup": args.warmup,
            "ms_per_step": elapsed / args.iters * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"freemuxlet EM (BASELINE.json configs[{args.config}]): {C} cells x {S} SNPs, "
                                   f"K = {K}, {p.nnz} entries, {args.iters} EM iterations (E-step, scans, "
                                   f"re-assignment, ordered M-step, exchanges)",
                       "cells": C, "snps": S, "clusters": K, "entries": int(p.nnz),
                       "parallelism": f"E-step by cells x{world}, M-step by SNPs x{world}",
                       "backend": args.dist_backend},
            "entries_per_s": p.nnz * args.iters / elapsed,
            "setup_ms": setup_s * 1e3,  # shard tables + initial cluster pileups (muxgl_fmx_set_shard / set_clusters)
            "last_iteration": {"nsingle": hist[-1][0], "namb": hist[-1][1], "nchanged": hist[-1][2]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, K, clust0)
        print(json.dumps(out), flush=True)
        if args.dump:
            np.savez(args.dump, cells=cells, hist=np.array(hist, dtype=np.int64))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
