#!/usr/bin/env python
"""Benchmark of the demuxlet / freemuxlet hot path on MI355X.

    python bench.py --gpus 1 --steps 2000 --warmup 200              # BASELINE.json's metric on configs[1] (headline)
    python bench.py --config 2 --steps 3 --warmup 1                 # demuxlet 100k x 64 x 200k, six alphas
    python bench.py --config 3 --steps 20 --warmup 2                # freemuxlet K=16, 50k x 100k, 20 EM iterations
    python bench.py --gpus N ...                                    # starts N ranks of itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config I]         # the same, launched from outside

One process per GPU.
  * demuxlet (configs 1, 2): a "step" is one full pass of the hot path (entry likelihoods + sample-pair sweep +
    evidence/scan/call + per-cell records to the host) over the whole synthetic pileup of the workload, resident in HBM.
    Every rank owns its own config-sized shard of cells (WEAK scaling; demuxlet's cells are independent, so there is no
    data-path collective -- torch.distributed/RCCL only provides the barrier and the max-over-ranks of the elapsed time).
  * freemuxlet (configs 3, 4): a "step" is one EM iteration (cluster posteriors, E-step, scans, re-assignment, ordered
    M-step and the two all-gathers + one all-reduce between them) of the fixed job (STRONG scaling): the E-step is
    sharded by cells, the ordered M-step by SNPs, each rank holding 2/N of the pileup (popscle_amd/freemuxlet.py).
  * The default run (config 1) also times, on the same ranks and inside the same JSON line, the other BASELINE.json
    configs: "freemuxlet_em" (configs[3], 20 EM iterations), "demuxlet_config2" (configs[2], the north_star's
    100 k x 64 x 200 k shape, 3 passes) and "freemuxlet_config4" (configs[4], 500 k x 500 k, K = 64: 2 EM iterations;
    sharded over the ranks for N > 1) -- each with its own roofline and CPU baseline, each behind a watchdog so that it
    can never cost the headline (--no-legs skips them; --legs 3,2,4 selects).  Their inputs are generated on the GPU
    (synth.make_pileup_device: the numpy generator needs minutes for 5 x 10^8 entries); a sharded leg moves only the
    rank's two slabs to the host.

Prints ONE JSON line (rank 0).  LL = one hypothesis log-likelihood feeding a call: demuxlet V singlets +
V(V-1)(A-1) ordered doublets per cell, freemuxlet K(K+1)/2 per cell and iteration.

Roofline ("the binding fraction is the max of the two"): two roofs are priced for the dominant kernel,
    hbm   bytes that actually crossed the L2<->fabric boundary (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, own passes,
          profiles/traffic.json, quoted only when it was measured on these very sources) / time / 8.0 TB/s -- next to
          the ALGORITHMIC bytes of SURVEY 8d (a gather of cache-resident GP rows: an equivalent rate, not a bandwidth);
    fp64  FP64 operations actually ISSUED (PMC: 64 x (2 FMA + MUL + ADD) wave instructions; a model of the same when no
          fresh counters exist) / time / 78.6 TFLOP/s -- next to the reference's operation count (which the kernels
          undercut by evaluating only the hypotheses the reference reads, mirrored alpha = 0.5 pairs once);
and "bound" names the larger fraction.
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import signal
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from popscle_amd import freemuxlet, muxgl, shard, synth  # noqa: E402
from popscle_amd.build import source_hash  # noqa: E402

METRIC = "cell-sample-pair LLs/sec (singlet+doublet), demuxlet 10k cells×16 samples"
FMX_METRIC = "freemuxlet EM cell-cluster-pair LLs/sec"
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # vector FP64 = half the 157.3 TF FP32 vector rate (same guide); FP64 MFMA runs at the same rate
SIMDS, PEAK_CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 4.0  # 256 CUs x 4 SIMDs; one wave64 VALU instruction = 4 issue cycles


# ---- per-entry work, SURVEY.md 8d -------------------------------------------------------------------------------------
def demux_bytes_per_entry(V, rpe):
    return 12.0 + rpe + 24.0 * V  # SNP id + read offset, reads, GP row gather


def demux_flops_per_entry(V, A, rpe):
    return A * (18.0 * V + 7.0 * V * V) + rpe * A * 27.0  # the reference's own operation count


def demux_issued_flops_model(V, alphas, rpe, f_lin=0.0):
    """FP64 operations the kernels issue per entry (model, used when no fresh PMC pass exists; counted like the counters
    do: FMA = 2, MUL = ADD = 1): a hypothesis of a general entry costs a 3-term dot product (MUL + 2 FMA = 5) and one
    multiply into its product accumulator, a hypothesis of a linear entry (share f_lin: at most one usable read) one FMA
    and that multiply (3); the u-factors 3 dots per sample and alpha; singlets a dot, the sample-0 factor and the
    accumulate; alpha = 0.5 pairs are evaluated once (k < j)."""
    nsym = sum(1 for a in alphas[1:] if a == 0.5)
    nns = len(alphas) - 1 - nsym
    pairs = V * (V - 1)
    hyp = nsym * pairs / 2 + nns * pairs
    return hyp * (3.0 * f_lin + 6.0 * (1.0 - f_lin)) + (nsym + nns) * V * 15.0 + V * 7.0 + rpe * len(alphas) * 27.0


def fmx_bytes_per_entry(K):
    return 76.0 + 24.0 * K  # SNP id + 9 entry likelihoods, cluster-GP row gather


def fmx_flops_per_entry(K):
    return 33.0 * K + 3.5 * K * (K + 1)


def fmx_issued_flops_model(K, f_lin=0.0):
    return K * (K + 1) / 2 * (3.0 * f_lin + 6.0 * (1.0 - f_lin)) + K * 15.0


# ---- the floor: the least work the algorithm AS BUILT must do (DESIGN.md section 6.1 states the same table) -----------
# SURVEY.md 8d's byte / flop figures count every GP-row gather as HBM bytes and every (j, k, n) slot of the reference's
# loop nest as work; the kernels keep the GP tensor in cache and evaluate only the hypotheses the reference READS, a
# linear entry (at most one usable read) at two instructions per hypothesis.  With those figures the fractions exceed 1
# for configs[1] and [3] (reported as `reference_equiv_*`, without a fraction).  The floor below is what a reader can
# recompute: FP64 lane-instructions that cannot be avoided without another algorithm, at 4 issue cycles per wave64
# instruction on 1024 SIMDs at 2.4 GHz, and the bytes that must cross HBM once, at 8 TB/s.
def linear_fraction(entry_rptr, reads):
    """share of the entries with at most one usable read (allele 0/1): the class the kernels sweep at 2 instructions
    per hypothesis (the library also requires the read's quality <= 60, which the synthetic qualities always meet)"""
    rp = np.asarray(entry_rptr, dtype=np.int64)
    if rp.size < 2:
        return 0.0
    cs = np.concatenate(([0], np.cumsum(np.asarray(reads) != 0xFF, dtype=np.int64)))
    return float(np.mean((cs[rp[1:]] - cs[rp[:-1]]) <= 1))


def demux_hypotheses(V, alphas):
    """hypotheses the reference reads per entry: V singlets (j, 0, 0) and, per non-first alpha, the V (V - 1) ordered
    pairs -- an alpha of 0.5 is symmetric in (j, k): evaluated once, mirrored"""
    nsym = sum(1 for a in alphas[1:] if a == 0.5)
    nns = len(alphas) - 1 - nsym
    return V + nns * V * (V - 1) + nsym * V * (V - 1) // 2


def demux_floor(V, alphas, nnz, frac_lin, rpe, S, C, kernel_ms):
    A = len(alphas)
    H = demux_hypotheses(V, alphas)
    # linear entry: one FMA (partner's rho times B_m, plus the lane's A + B_l rho_j) and the product update per
    # hypothesis; the lane's A + B_l rho_j once per sample and alpha.  Other entries: a three-term dot product
    # (MUL + 2 FMA) and the update; u[m] = sum_l g_j[l] pG[l][m] (9 per sample and alpha); the per-read update of
    # cmd_cram_demuxlet.cpp:655-700 (9 products per alpha and read)
    i_lin = 2.0 * H + V * A
    i_gen = 4.0 * H + 9.0 * V * A + 9.0 * A * rpe
    lane_instr = nnz * (frac_lin * i_lin + (1.0 - frac_lin) * i_gen)
    valu_ms = lane_instr / 64.0 * VALU_CYCLES / (SIMDS * PEAK_CLOCK_HZ) * 1e3
    # every input once: 4 B SNP id + 8 B read offset + the reads per entry, the GP tensor, the 160-byte records
    byts = nnz * (12.0 + rpe) + S * V * 24.0 + C * 160.0
    hbm_ms = byts / (HBM_PEAK_GBS * 1e9) * 1e3
    fl = max(valu_ms, hbm_ms)
    return {"hypotheses_per_entry": H, "linear_entry_share": frac_lin,
            "fp64_instr_per_hypothesis": {"linear": 2, "general": 4},
            "lane_instructions": lane_instr, "valu_ms": valu_ms, "compulsory_bytes": byts, "hbm_ms": hbm_ms,
            "frac_of_floor": (fl / kernel_ms) if kernel_ms > 0 else None,
            "note": "floor of the algorithm as built (DESIGN.md 6.1): FP64 issue at 2.4 GHz / compulsory bytes at 8 TB/s"}


def fmx_floor(K, nnz, frac_lin, S, kernel_ms):
    H = K * (K + 1) // 2  # unordered pairs and the K singlets (cmd_cram_freemux2.cpp:440-452)
    i_lin = 2.0 * H + K            # (c0 + c1 E_j) once per cluster
    i_gen = 4.0 * H + 9.0 * K      # u[m] = sum_l P_j[l] glis[l][m]
    lane_instr = nnz * (frac_lin * i_lin + (1.0 - frac_lin) * i_gen)
    valu_ms = lane_instr / 64.0 * VALU_CYCLES / (SIMDS * PEAK_CLOCK_HZ) * 1e3
    # a linear entry streams {c0, c1, snp} (24 B), another one its six distinct likelihoods and its SNP id (52 B);
    # the cluster posteriors once
    byts = nnz * (frac_lin * 24.0 + (1.0 - frac_lin) * 52.0) + S * K * 24.0
    hbm_ms = byts / (HBM_PEAK_GBS * 1e9) * 1e3
    fl = max(valu_ms, hbm_ms)
    return {"hypotheses_per_entry": H, "linear_entry_share": frac_lin,
            "fp64_instr_per_hypothesis": {"linear": 2, "general": 4},
            "lane_instructions": lane_instr, "valu_ms": valu_ms, "compulsory_bytes": byts, "hbm_ms": hbm_ms,
            "frac_of_floor": (fl / kernel_ms) if kernel_ms > 0 else None,
            "note": "floor of the E-step as built (DESIGN.md 6.1): FP64 issue at 2.4 GHz / compulsory bytes at 8 TB/s"}


def demux_sweep_kernel(V, alphas):
    """the kernel libmuxgl dispatches for this shape (popscle_amd/csrc/demux_kernels.hip: demux_launch)"""
    if V <= 32 and tuple(alphas) == (0.0, 0.5):
        return "demux_oct_kernel"  # (eight lanes per entry up to 16 samples, sixteen beyond: demux_oct.hip)
    if V <= 16 and sum(1 for a in alphas[1:] if a != 0.5) <= 5 and sum(1 for a in alphas[1:] if a == 0.5) <= 1:
        return "demux_row_kernel"
    if V <= 32 and len(alphas) == 2 and alphas[1] == 0.5 and alphas[0] != 0.5:
        return "demux_row2_kernel"
    if V <= 32:
        return "demux_wave32_kernel"
    if V <= 64:
        # one kernel walks every entry of a work unit (round 4): demux_ring.hip
        return "demux_ring_lin_kernel"
    if V <= 255:
        return "demux_ring_lin_kernel + demux_wave_kernel"  # diagonal / off-diagonal 64 x 64 blocks of the pair matrix
    return "demux_sweep_kernel"


def fmx_estep_kernel(K):
    if K <= 16:
        return "fmx_estep_oct_kernel"
    if K <= 32:
        return "fmx_estep_row2_kernel"
    return "fmx_estep_wave_kernel"


def pmc_record(config):
    """counters of the dominant kernel for this config from profiles/traffic.json -- only if they were measured on the
    sources this run is built from (the file carries their fingerprint)"""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        d = json.load(open(tfile))
    except Exception:
        return None, "no profiles/traffic.json"
    rec = d.get(f"config{config}")
    if not rec:
        return None, f"profiles/traffic.json has no config{config}"
    if rec.get("source_hash") != source_hash(config):  # fingerprint of the machine code of this config's kernel family
        return None, (f"profiles/traffic.json config{config} is stale (measured on sources {rec.get('source_hash')}, "
                      f"git {rec.get('git_head')})")
    return rec, d.get("measured_with", "")


def roofline(kernel, kern_s, abytes, ref_flops, issued_model_flops, config, scale=1.0):
    """the two roofs of the dominant kernel; `scale` = this run's units / the units of the PMC pass (a reduced run)"""
    rec, note = pmc_record(config)
    if rec and rec.get("units") and scale:
        k = scale / rec["units"]  # counters are per launch over rec["units"] entries
    else:
        k = None
    traffic = rec["hbm_bytes_per_launch"] * k if rec and k and rec.get("hbm_bytes_per_launch") else None
    issued, src = issued_model_flops, "model (popscle_amd per-hypothesis instruction count)"
    valu = None
    if rec and k and rec.get("fma_f64") is not None:
        issued = 64.0 * (2.0 * rec["fma_f64"] + rec.get("mul_f64", 0.0) + rec.get("add_f64", 0.0)) * k
        src = "pmc (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64, " + str(note) + ")"
        valu = rec.get("valu", 0.0) * k or None
    if kern_s <= 0:
        return {"bound": None, "kernel": kernel, "achieved": None, "peak": None, "unit": None, "frac": None,
                "traffic": traffic, "note": "no kernel time was recorded"}
    t = kern_s
    hbm = {"algorithmic_bytes": abytes, "algorithmic_equiv_GBps": abytes / t / 1e9,
           "traffic_bytes": traffic, "traffic_over_algorithmic": (traffic / abytes) if traffic else None,
           "achieved": (traffic / t / 1e9) if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": (traffic / t / 1e9 / HBM_PEAK_GBS) if traffic else None,
           "note": "traffic = PMC FETCH_SIZE x2 + WRITE_SIZE per launch" if traffic else f"no counter traffic: {note}"}
    fp = {"issued_flops": issued, "source": src, "achieved": issued / t / 1e12, "peak": FP64_PEAK_TFLOPS,
          "unit": "TFLOP/s", "frac": issued / t / 1e12 / FP64_PEAK_TFLOPS, "reference_flops": ref_flops,
          "reference_equiv_TFLOPs": ref_flops / t / 1e12, "issued_over_reference": issued / ref_flops if ref_flops else None,
          "valu_issue_frac": (valu * VALU_CYCLES / (SIMDS * PEAK_CLOCK_HZ * t)) if valu else None}
    bound = "hbm" if (hbm["frac"] or 0.0) > fp["frac"] else "fp64_valu"
    top = hbm if bound == "hbm" else fp
    return {"bound": bound, "kernel": kernel, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
            "frac": top["frac"], "traffic": traffic, "kernel_ms": kern_s * 1e3, "hbm": hbm, "fp64": fp}


# ---- CPU baselines: the oracle (restatement of the reference's loops, kind "port") on this host's cores ---------------
def usable_cores():
    """cores this process may actually use: physical cores, capped by the affinity mask and by the cgroup CPU quota (a
    container on a 128-core host with a quota of 16 CPUs runs 128 workers no faster than 16)"""
    try:
        import psutil

        n = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:  # cgroup v2, then v1
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period)))
        except Exception:
            pass
    return int(max(1, min(n, 256)))


def _demux_shard_job(job):
    p, alphas = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    t0 = time.perf_counter()
    ob.demux(p, alphas=alphas, nthreads=1)
    return time.perf_counter() - t0


def cpu_baseline_demux(p, alphas, gpu_cells, budget_s=9.0):
    """(i) ONE thread -- the reference's execution model (it has no threads) -- and (ii) N independent
    single-threaded processes over N cell shards, the reference's documented way to use more cores (--group-list,
    README.md:168), N = physical cores.  Bounded samples of the same workload; the sampled cells double as a last parity
    check of the GPU records."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multiprocessing as mp

    import oracle_binding as ob
    import parity

    V, A = p.gp.shape[1], len(alphas)
    lls_per_cell = V + V * (V - 1) * (A - 1)
    rng = np.random.default_rng(0)
    probe = p.subset_cells(np.sort(rng.choice(p.C, min(p.C, 8), replace=False)))
    t0 = time.perf_counter()
    ob.demux(probe, alphas=alphas, nthreads=1)
    per_cell = max(time.perf_counter() - t0, 1e-6) / probe.C
    # (i) one thread
    n1 = int(min(p.C, max(8, budget_s / per_cell)))
    pick = np.sort(rng.choice(p.C, n1, replace=False))
    sub = p.subset_cells(pick)
    t0 = time.perf_counter()
    want = ob.demux(sub, alphas=alphas, nthreads=1)
    dt1 = time.perf_counter() - t0
    # parity of the checked cells: every integer field EQUAL to the oracle's after the product's exact-call pass (what
    # popscle-amd demuxlet runs before it writes .best; host code, outside the timed steps -- timed over ALL cells below)
    rep = parity.compare_demux(np.ascontiguousarray(gpu_cells[pick]), want, alphas, sub, nthreads=usable_cores())
    single = {"value": n1 * lls_per_cell / dt1, "unit": "LLs/s", "cores": 1, "entries_per_s": sub.nnz / dt1,
              "sample": f"{n1} of {p.C} cells ({int(sub.nnz)} entries), one thread, {dt1:.1f} s"}
    # (ii) N processes, one cell shard each
    ncores = usable_cores()
    nn = int(min(p.C, max(ncores, ncores * budget_s / per_cell)))
    pickn = np.sort(rng.choice(p.C, nn, replace=False))
    shards = [p.subset_cells(pickn[i::ncores]) for i in range(ncores) if len(pickn[i::ncores])]
    ents = int(sum(s.nnz for s in shards))
    with mp.get_context("fork").Pool(len(shards)) as pool:
        t0 = time.perf_counter()
        pool.map(_demux_shard_job, [(s, alphas) for s in shards])
        dtn = time.perf_counter() - t0
    return {
        "value": nn * lls_per_cell / dtn, "unit": "LLs/s", "cores": len(shards), "kind": "port",
        "sample": f"{nn} of {p.C} cells of the same workload ({ents} entries) as {len(shards)} independent single-threaded "
                  f"processes over cell shards (oracle/muxgl_oracle.c; the reference's --group-list parallelisation), "
                  f"{dtn:.1f} s",
        "entries_per_s": ents / dtn, "single_thread": single,
        # what the host really delivered: a quota or shared cores show up here, whatever the core count says
        "speedup_over_single_thread": (ents / dtn) / single["entries_per_s"],
        "note": "LLs/s per core depends on entries per cell (LLs per cell are fixed, work is per entry): this workload "
                "has ~950 entries per cell, BASELINE.md's reference timing (36.6 k LLs/s, 71 k entries/s per core) had 500",
        "parity_checked_cells": rep["cells"], "parity_max_abs_ll_diff": rep["max_abs_ll_diff"],
        # tests/parity.py has no relaxation of "exact calls" any more (all zeros; key kept for readers of older lines)
        "parity_excuses_used": rep["excuses_used"],
        "parity_raw_records_differing": rep["raw_records_differing"], "exact_pass_on_checked_cells": rep["exact_pass"],
    }


def cpu_baseline_fmx(p, K, clust0, budget_s=10.0):
    """One EM iteration of the CPU oracle (restatement of cmd_cram_freemux2.cpp:375-597, kind "port") on a bounded sample of
    the cells, single-threaded: the reference's freemuxlet has no process-level parallelism either (global EM state)."""
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


def fmx_parity_sample(eng, p, K, cells_now, n=24):
    """One more EM iteration on the device, untimed, and the oracle's E-step, scans and re-assignment of a sample of the
    cells against the device's OWN cluster pileups of that moment (as tests/test_large_gpu.py does at configs[4]): the
    freemuxlet legs' counterpart of the demuxlet legs' parity check.  The oracle is the checker only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import parity

    gls, cnt = eng.fmx_cluster_pileup()  # [K][S][9], [K][S][3]: the state the next E-step starts from
    nxt, _ = eng.fmx_iterate(0.5, 0.1)
    pick = np.sort(np.random.default_rng(2).choice(p.C, min(n, p.C), replace=False))
    sub = p.subset_cells(pick)
    se = ob.fmx_entry_pileup(sub)
    cplp = np.zeros((K, p.S), dtype=ob.PLP)
    cplp["gls"] = gls
    cplp["nreads"], cplp["nref"], cplp["nalt"] = cnt[..., 0], cnt[..., 1], cnt[..., 2]
    del gls, cnt
    clust = np.where(cells_now["type"][pick] == 0, cells_now["clust"][pick], -1).astype(np.int32)
    ocells = ob.fmx_init_cells(np.ascontiguousarray(clust))
    ob.fmx_iterate(sub, se, K, cplp, ocells, 0.5, 0.1, nthreads=min(8, usable_cores()))
    # (the checker starts from the DEVICE's cluster pileups, which equal the reference's to ~1e-12 only: a cell whose call
    #  is within rounding reach of that may legitimately differ and is counted, not excused silently)
    rep = parity.compare_fmx(nxt[pick], ocells, resolved=False)
    return {"parity_checked_cells": int(rep["cells"]), "parity_max_abs_ll_diff": float(rep["max_abs_ll_diff"]),
            "parity_excuses_used": rep.get("excuses_used")}


# ---- distributed context --------------------------------------------------------------------------------------------
class Ctx:
    def __init__(self, args):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != max(1, args.gpus):
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        self.dev = 0 if args.single_device else local_rank
        torch.cuda.set_device(self.dev)
        self.backend = args.dist_backend
        # --force-dist: the N-rank code path (process group, slabs, collectives on the library's buffers, stream ordering)
        # with a single rank -- what a 1-GPU box can exercise of the RCCL path
        self.dist_on = self.world > 1 or args.force_dist
        if self.dist_on and self.world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if self.dist_on:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            to = datetime.timedelta(seconds=600)
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.dev), timeout=to)
            else:
                dist.init_process_group(self.backend, timeout=to)
        self.tdev = "cuda" if self.backend == "nccl" else "cpu"
        # every rank of the job must be IN the collective world: an all-reduce of 1 over the backend that carries the
        # exchanges (RCCL under the driver's launch).  A launcher that started fewer processes than --gpus, or ranks that
        # ended up in different worlds, fail here, loudly, instead of producing a plausible line.
        self.ranks_seen = 1
        if self.dist_on:
            t = torch.ones(1, dtype=torch.int64, device=self.tdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.ranks_seen = int(t.item())
            if self.ranks_seen != max(1, args.gpus):
                raise SystemExit(f"[bench rank {self.rank}] {self.ranks_seen} ranks answered the all-reduce over "
                                 f"{self.backend}, --gpus says {args.gpus}")

    def stamp(self, out):
        """what an N > 1 line says about the world it ran in"""
        if out is not None and self.dist_on:
            out["dist"] = {"backend": "rccl (torch nccl)" if self.backend == "nccl" else self.backend,
                           "ranks_seen": self.ranks_seen, "world_size": self.world}
            out["rccl_ranks_seen" if self.backend == "nccl" else "ranks_seen"] = self.ranks_seen
        return out

    def gather_floats(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank"""
        if self.world == 1:
            return [float(x)]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.tdev)
        t[self.rank] = float(x)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def barrier(self):
        if self.dist_on:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, xs):
        if self.world == 1:
            return [float(x) for x in xs]
        t = self.torch.tensor([float(x) for x in xs], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell: start N ranks of this very command under torch.distributed.run (one
    process per GPU, RCCL or gloo rendezvous on 127.0.0.1) and hand their output through; rank 0 prints the JSON line."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def use_device_synth(args, entries_estimate):
    """large inputs are generated on the GPU (torch), small ones and every test-sized run by the numpy generator"""
    if args.synth == "numpy":
        return False
    return args.synth == "device" or entries_estimate > 2.0e7


def free_torch_cache(ctx):
    import gc

    gc.collect()
    ctx.torch.cuda.empty_cache()


# ---- demuxlet leg (configs 1, 2): weak scaling -----------------------------------------------------------------------
def demux_leg(args, ctx, config, steps=None, warmup=None, ramp_seconds=None, cpu_budget_s=9.0, strong=False):
    """strong = False: weak scaling, every rank sweeps a config-sized shard of its own (the driver's contract for the
    headline).  strong = True: ONE config-sized job, the same on every rank count, its cells cut into contiguous ranges
    balanced by entries (shard.cell_shards, what popscle_amd/demuxlet.py does): the N-GPU number then says how much
    faster the same job got."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    ramp_seconds = args.ramp_seconds if ramp_seconds is None else ramp_seconds
    cfg = synth.CONFIGS[config]
    alphas = tuple(cfg["alphas"])
    V, A = cfg["V"], len(alphas)
    # weak scaling: every rank owns a full config-sized shard of cells (its own seed), GP tensor replicated
    C = max(1, int(round(cfg["C"] * args.scale)))
    seeds = dict(seed=synth.BASE_SEED + config + (0 if strong else 1000 * ctx.rank), donor_seed=synth.BASE_SEED + config)
    reads_lambda = args.reads_lambda if args.reads_lambda is not None else (2.0 if args.dense else None)
    if reads_lambda is not None:
        seeds["reads_lambda"] = reads_lambda
    t_gen = time.perf_counter()
    on_device = use_device_synth(args, C * 950.0)
    job_cells = C
    if on_device:
        d = synth.make_pileup_device(C, cfg["S"], V, device=f"cuda:{ctx.dev}", **seeds)
        if strong and ctx.world > 1:
            c0, c1 = shard.cell_shards(d.cell_ptr.cpu().numpy(), ctx.world)[ctx.rank]
            p = d.take_cells(c0, c1)
        else:
            p = d.host()
        del d
        free_torch_cache(ctx)
    else:
        p = synth.make_pileup(C, cfg["S"], V, **seeds)
        if strong and ctx.world > 1:
            p = shard.take_cells(p, *shard.cell_shards(p.cell_ptr, ctx.world)[ctx.rank])
    C = p.C
    gen_s = time.perf_counter() - t_gen
    eng = muxgl.Engine(ctx.dev, muxgl.FLAG_NO_LINEAR_ENTRIES if args.no_linear else 0)
    t_h = time.perf_counter()
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.demux_set_gp(p.gp, p.has_gp)
    handover_s = time.perf_counter() - t_h

    t_ramp = time.perf_counter()
    ramp_passes = 0
    while time.perf_counter() - t_ramp < ramp_seconds:
        eng.demux_run(alphas, 0.5, want_cells=False)
        ramp_passes += 1
    for _ in range(warmup):
        eng.demux_run(alphas, 0.5, want_cells=False)
    ctx.barrier()
    eng.timing_sum(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.demux_run(alphas, 0.5, want_cells=False)  # returns with the stream drained and the records on the host
    ctx.barrier()
    own_s = time.perf_counter() - t0
    elapsed = ctx.max_over_ranks(own_s)
    # hipEvent times of the timed passes (the live kernel durations of the roofline), summed inside the library: a fetch
    # per pass cost 7.6 us of the 350 us step
    kern_ms, n_timed = eng.timing_sum()
    assert n_timed == steps
    total_cells, total_entries = ctx.sum_over_ranks([p.C, p.nnz])
    rank_sweep_ms = ctx.gather_floats(kern_ms[muxgl.T_DEMUX_SWEEP] / steps)
    rank_entries = ctx.gather_floats(p.nnz)
    out = None
    if ctx.rank == 0:
        lls_per_cell = V + V * (V - 1) * (A - 1)
        step_s = elapsed / steps
        kern_ms /= steps
        rpe = p.R / max(p.nnz, 1)
        sweep_s = kern_ms[muxgl.T_DEMUX_SWEEP] * 1e-3
        frac_lin = linear_fraction(p.entry_rptr, p.reads)
        out = {
            "metric": METRIC if config == 1 else f"cell-sample-pair LLs/sec (singlet+doublet), demuxlet BASELINE.json configs[{config}]",
            "value": total_cells * lls_per_cell / step_s, "unit": "LLs/s", "n_gpus": ctx.world, "steps": steps,
            "warmup": warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # untimed passes before the W warmup steps (the engine clock ramps up over a few hundred launches)
            "ramp": {"seconds": ramp_seconds, "untimed_passes": ramp_passes},
            "input": {"generator": "torch on the GPU (synth.make_pileup_device)" if on_device else "numpy (synth.make_pileup)",
                      "generate_s": gen_s, "handover_s": handover_s},
            "config": {
                "workload": f"demuxlet synthetic PLP (BASELINE.json configs[{config}]): {C} cells x {V} samples x "
                            f"{cfg['S']} SNPs per GPU, alpha grid {list(alphas)}, {p.nnz} entries, {p.R} reads",
                "cells_per_gpu": C, "samples": V, "snps": cfg["S"], "alphas": list(alphas),
                "entries_per_gpu": int(p.nnz), "lls_per_cell": lls_per_cell, "parallelism": f"cells sharded x{ctx.world}",
            },
            "entries_per_s": total_entries / step_s, "cells_per_s": total_cells / step_s,
            # oct path: "reduce" is the fused finish kernel (chunk reduction + call + records written to pinned host
            # memory), "call" and "d2h" are then 0; other paths run them as separate launches
            "kernel_ms": {"sweep": float(kern_ms[muxgl.T_DEMUX_SWEEP]), "reduce": float(kern_ms[muxgl.T_DEMUX_REDUCE]),
                          "call": float(kern_ms[muxgl.T_DEMUX_CALL]), "d2h": float(kern_ms[muxgl.T_DEMUX_D2H])},
            "roofline": roofline(demux_sweep_kernel(V, alphas), sweep_s, demux_bytes_per_entry(V, rpe) * p.nnz,
                                 demux_flops_per_entry(V, A, rpe) * p.nnz,
                                 demux_issued_flops_model(V, alphas, rpe, 0.0 if args.no_linear else frac_lin) * p.nnz, config,
                                 scale=float(p.nnz)),
        }
        out["roofline"]["floor"] = demux_floor(V, alphas, float(p.nnz), 0.0 if args.no_linear else frac_lin, rpe, cfg["S"],
                                               C, sweep_s * 1e3)
        if reads_lambda is not None or args.no_linear:
            out["sensitivity"] = {"reads_lambda": 0.3 if reads_lambda is None else reads_lambda, "reads_per_entry": rpe,
                                  "linear_entry_share": frac_lin, "linear_entry_form": not args.no_linear,
                                  "note": "not the BASELINE workload: the default line's entries carry 1 + Poisson(0.3) reads "
                                          "(three quarters linear: two FP64 instructions per hypothesis instead of four)"}
        if ctx.world > 1 and strong:
            out["scaling_note"] = ("strong scaling: ONE job of %d cells, cut into %d contiguous cell ranges balanced by entries "
                                   "(no data-path collective; cells are independent, cmd_cram_demuxlet.cpp:636-1013); "
                                   "value = the job's LLs / max-rank time" % (job_cells, ctx.world))
            out["per_rank"] = {"sweep_kernel_ms": rank_sweep_ms, "entries": rank_entries}
        elif ctx.world > 1:  # every rank swept a full config-sized shard of its own: N x the one-GPU value by construction
            out["scaling_note"] = ("weak scaling: each of the %d ranks owns %d cells (no data-path collective; cells are "
                                   "independent, cmd_cram_demuxlet.cpp:636-1013); value = sum over ranks / max-rank time"
                                   % (ctx.world, C))
            out["per_rank_value"] = out["value"] / ctx.world
            out["per_rank"] = {"sweep_kernel_ms": rank_sweep_ms, "entries": rank_entries}
        out["roofline"]["note_8d"] = ("SURVEY 8d's per-entry bytes / flops (reference_equiv_*) exceed the roofs for V <= 16: they count "
                                      "cache-resident GP-row gathers as HBM bytes and all V*V*A slots of the reference's loop "
                                      "nest as work; `floor` is the recomputable bound of the algorithm as built")
        if not args.no_cpu_baseline and ctx.world == 1:
            raw = eng.demux_results_view().copy()
            out["cpu_baseline"] = cpu_baseline_demux(p, alphas, raw, budget_s=cpu_budget_s)
            # The product's exact-call pass over ALL cells of the step (host threads; what popscle-amd demuxlet runs after
            # muxgl_demux_run and before it writes .best: the printed order of every mirrored alpha = 0.5 pair and every
            # near-tie call in the reference's own arithmetic).  NOT part of `value` / `ms_per_step`: stated here so that
            # nobody has to guess what bit-exact DBL.BEST.GUESS costs next to the kernels.
            t0 = time.perf_counter()
            st = muxgl.demux_exact_calls(p, alphas, raw, 0.5, nthreads=usable_cores())
            ms = (time.perf_counter() - t0) * 1e3
            out["exact_calls_pass"] = dict(st, ms_all_cells=ms, threads=usable_cores(), timed_in_value=False)
            out["pair_order_ms_all_cells"] = ms   # (the name round 5's review asked for; the pass now also settles near ties)
        else:
            out["cpu_baseline"] = None
    eng.close()
    return out


# ---- freemuxlet leg (configs 3, 4): strong scaling -------------------------------------------------------------------
def fmx_leg(args, ctx, config, steps, warmup, cpu_baseline=True, cpu_budget_s=10.0):
    torch, dist = ctx.torch, ctx.dist
    cfg = synth.CONFIGS[config]
    C = args.cells or max(1, int(round(cfg["C"] * args.scale)))
    S = args.snps or cfg["S"]
    K = args.clusters or cfg["V"]
    ordered = ctx.dist_on and ctx.backend == "nccl"
    (c_ranges, per_c), (s_ranges, per_s) = freemuxlet.plan_ranges(C, S, ctx.world)
    # the same job on every rank (same seed): strong scaling.  A rank keeps only its row slab (its cells) and its
    # column slab (its SNPs) on the host and on the device: 2/N of the pileup.
    gen = dict(seed=synth.BASE_SEED + config, with_gp=False, mean_entries=args.mean_entries,
               min_entries=min(50, max(1, int(args.mean_entries // 4))))
    t_gen = time.perf_counter()
    on_device = use_device_synth(args, C * args.mean_entries * 1.2)
    p = rows = cols = None
    if on_device:
        d = synth.make_pileup_device(C, S, K, device=f"cuda:{ctx.dev}", **gen)
        nnz_total, my_entries = d.nnz, float(d.cell_ptr[c_ranges[0][1]] - d.cell_ptr[c_ranges[0][0]])
        usable = torch.cumsum((d.reads != 0xFF).to(torch.int64), 0)
        usable = torch.cat([usable.new_zeros(1), usable])
        frac_lin = float(((usable[d.entry_rptr[1:]] - usable[d.entry_rptr[:-1]]) <= 1).double().mean())
        del usable
        af, truth_s1 = d.af.cpu().numpy(), d.truth["s1"].cpu().numpy()
        if ctx.dist_on:
            rows = d.take_cells(*c_ranges[ctx.rank])
            cols = d.take_snps(*s_ranges[ctx.rank])
        if not ctx.dist_on or (ctx.rank == 0 and cpu_baseline and ctx.world == 1 and not args.no_cpu_baseline):
            p = d.host()
        del d
        free_torch_cache(ctx)
    else:
        p = synth.make_pileup(C, S, K, **gen)
        nnz_total, my_entries = p.nnz, float(p.cell_ptr[c_ranges[0][1]] - p.cell_ptr[c_ranges[0][0]])
        frac_lin = linear_fraction(p.entry_rptr, p.reads)
        af, truth_s1 = p.af, p.truth["s1"]
        if ctx.dist_on:
            rows = shard.take_cells(p, *c_ranges[ctx.rank])
            cols = shard.take_snps(p, *s_ranges[ctx.rank])
    gen_s = time.perf_counter() - t_gen
    eng = muxgl.Engine(ctx.dev, muxgl.FLAG_ASYNC_PHASES if ordered else 0)
    t0 = time.perf_counter()
    if ctx.dist_on:
        eng.set_pileup(S, rows.cell_ptr, rows.entry_snp, rows.entry_rptr, rows.reads)
        eng.fmx_set_column_slab(C, c_ranges[ctx.rank][0], *s_ranges[ctx.rank], *cols)
        slab_bytes = sum(a.nbytes for a in (rows.cell_ptr, rows.entry_snp, rows.entry_rptr, rows.reads) + tuple(cols))
        del rows, cols
    else:
        eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
        slab_bytes = sum(a.nbytes for a in (p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads))
    eng.fmx_prepare(af)
    handover_s = time.perf_counter() - t0
    # a seeded start (--init-cluster style): 90 % of the cells start in their source sample's cluster
    clust0 = np.where(np.random.default_rng(0).random(C) < 0.9, truth_s1, -1).astype(np.int32)
    ex = (freemuxlet.TorchExchange(dist, ctx.rank, ctx.world, device_ordered=ordered, always=True)
          if ctx.dist_on else None)
    stream_ctx = None
    if ordered:
        ext = torch.cuda.ExternalStream(eng.stream(), device=torch.device("cuda", ctx.dev))
        stream_ctx = lambda: torch.cuda.stream(ext)  # noqa: E731

    def run(n, timings=None):
        return freemuxlet.run_em(eng, K, clust0, max_iter=n, early_stop=False, exchange=ex,
                                 exchange_tensor=lambda e, w: freemuxlet.engine_exchange_tensor(e, w, ctx.dev),
                                 per=(per_c, per_s), timings=timings, sync=ctx.barrier, stream_ctx=stream_ctx)

    if warmup:
        run(warmup)
    tm = {}
    cells, hist = run(steps, tm)  # exactly `steps` EM iterations, bracketed by barrier + device synchronize on both sides
    kern = eng.timing()           # kernels of the last iteration on this rank
    elapsed = ctx.max_over_ranks(tm["loop_s"])
    rank_loop_ms = ctx.gather_floats(tm["loop_s"] / steps * 1e3)
    rank_estep_ms = ctx.gather_floats(float(kern[muxgl.T_FMX_ESTEP]))
    out = None
    if ctx.rank == 0:
        npairs = K * (K + 1) // 2
        # the roofline prices the pair-sweep kernel(s) alone -- the kernels the PMC traffic belongs to -- not the whole
        # E-step bracket (which also holds the relayout of the posteriors and the reduction of the chunk partials)
        est_s = float(kern[muxgl.T_FMX_ESTEP_SWEEP] or kern[muxgl.T_FMX_ESTEP]) * 1e-3
        out = {
            "metric": FMX_METRIC, "value": C * npairs * steps / elapsed, "unit": "LLs/s", "n_gpus": ctx.world,
            "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"freemuxlet EM (BASELINE.json configs[{config}]): {C} cells x {S} SNPs, K = {K}, "
                                   f"{nnz_total} entries, {steps} EM iterations (cluster posteriors, E-step, scans, "
                                   f"re-assignment, ordered M-step, exchanges)",
                       "cells": C, "snps": S, "clusters": K, "entries": int(nnz_total),
                       "parallelism": f"E-step by cells x{ctx.world}, ordered M-step by SNPs x{ctx.world}, "
                                      f"2 all-gathers + 1 all-reduce per iteration" if ctx.dist_on else "one GPU",
                       "backend": ctx.backend if ctx.dist_on else None},
            "entries_per_s": nnz_total * steps / elapsed,
            "input": {"generator": "torch on the GPU (synth.make_pileup_device)" if on_device else "numpy (synth.make_pileup)",
                      "generate_s": gen_s, "rank0_host_bytes": int(slab_bytes)},
            "handover_ms": handover_s * 1e3,  # H2D of this rank's slabs + derived tables + entry likelihoods
            "setup_ms": tm["setup_s"] * 1e3,  # initial cluster pileups (muxgl_fmx_set_clusters)
            "kernel_ms_rank0_last_iteration": {"gp": float(kern[muxgl.T_FMX_GP]), "estep": float(kern[muxgl.T_FMX_ESTEP]),
                                               "estep_sweep": float(kern[muxgl.T_FMX_ESTEP_SWEEP]),
                                               "call": float(kern[muxgl.T_FMX_CALL]), "mstep": float(kern[muxgl.T_FMX_MSTEP])},
            "last_iteration": {"nsingle": hist[-1][0], "namb": hist[-1][1], "nchanged": hist[-1][2]},
            # device-event times of the two exchanges per iteration on rank 0 (RCCL only): DESIGN.md 4.3's scaling model
            "exchange_ms_rank0": tm.get("exchange_ms"),
            "roofline": roofline(fmx_estep_kernel(K), est_s, fmx_bytes_per_entry(K) * my_entries,
                                 fmx_flops_per_entry(K) * my_entries, fmx_issued_flops_model(K, frac_lin) * my_entries, config,
                                 scale=my_entries),
        }
        fl = fmx_floor(K, my_entries, frac_lin, S, est_s * 1e3)
        rf = out["roofline"]
        rf["floor"] = fl
        # roofline.frac means the same in every leg of this line: FP64 flops ISSUED (PMC SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 x 64,
        # FMA counted twice) / kernel time / 78.6 TFLOP/s, or counter bytes / time / 8 TB/s when that is larger
        # (roofline.fp64.source says whether the counters or, without a current PMC record, the instruction model supplied the
        # flops).  The fraction of the recomputable floor of the algorithm as built is its own key: roofline.floor.frac_of_floor.
        if rf.get("traffic"):
            rf["hbm"]["traffic_over_compulsory"] = rf["traffic"] / fl["compulsory_bytes"]
            rf["hbm"]["note"] = ("traffic = PMC FETCH_SIZE x2 + WRITE_SIZE per launch: L2-side fabric requests, Infinity-Cache hits "
                                 "included -- row gathers that miss the L2, not compulsory HBM bytes")
        if ctx.world > 1:
            out["per_rank"] = {"ms_per_iteration": rank_loop_ms, "estep_kernel_ms_last_iteration": rank_estep_ms}
            # DESIGN.md 4.3's model next to the measurement: this rank's kernels + the two exchanges (xGMI: point to
            # point, ~70 GB/s usable per link and direction; a small collective ~25 us end to end) + the host's wait
            N = ctx.world
            kern_ms = float(kern[muxgl.T_FMX_GP] + kern[muxgl.T_FMX_ESTEP] + kern[muxgl.T_FMX_CALL] + kern[muxgl.T_FMX_MSTEP])
            slice_b = S * K * 24.0 / N
            x1_direct, x1_ring = 0.025 + slice_b / 70e9 * 1e3, 0.025 + (N - 1) * slice_b / 70e9 * 1e3
            x2 = 0.05
            out["scaling_model"] = {
                "kernels_ms_rank0_last_iteration": kern_ms,
                "exchange1_cluster_gp_allgather_ms": {"bytes_total": S * K * 24.0, "every_peer_on_its_own_link": x1_direct,
                                                      "single_ring": x1_ring},
                "exchange2_assignments_and_counters_ms": x2, "host_wait_ms": 0.02,
                "predicted_ms_per_iteration": {"direct": kern_ms + x1_direct + x2 + 0.02, "ring": kern_ms + x1_ring + x2 + 0.02},
                "measured_ms_per_iteration": elapsed / steps * 1e3,
                "note": "strong scaling: the same job on every rank count; E-step and scans shard by cells, cluster "
                        "posteriors and the ordered M-step by SNPs (DESIGN.md 4.3)"}
        # near-tie calls of the timed iterations: cells the exact path (csrc/fmx_exact.hip) settled inside the timed region, how
        # many of them it decided differently from the kernels, and cells left open (0: every path settles its own)
        out["exact_path"] = dict(zip(("near_tie_cells", "calls_changed", "unresolved"), eng.fmx_exact_stats()))
        if cpu_baseline and ctx.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_fmx(p, K, clust0, budget_s=cpu_budget_s)
            out["cpu_baseline"].update(fmx_parity_sample(eng, p, K, cells))
        else:
            out["cpu_baseline"] = None
        if args.dump:
            np.savez(args.dump, cells=cells, hist=np.array(hist, dtype=np.int64))
    eng.close()
    return out


LEGS = {  # secondary legs of the default run: JSON key, config, (steps, warmup), CPU-baseline budget, watchdog share
    3: ("freemuxlet_em", "fmx"),
    2: ("demuxlet_config2", "demux"),
    4: ("freemuxlet_config4", "fmx"),
    "2s": ("demuxlet_config2_strong", "demux_strong"),  # N > 1 only: the configs[2] job cut over the ranks
}


def guarded_leg(args, ctx, headline, config):
    """A secondary leg of the default run must never cost the headline line: a watchdog prints the line without it
    (rank 0, with the legs finished so far) and ends every rank cleanly if the leg hangs (a collective waiting for a
    rank that failed)."""
    key, kind = LEGS[config]

    def give_up(signum=None, frame=None, why="timeout"):
        if ctx.rank == 0:
            headline[key] = {"error": f"{key} did not finish: {why}"}
            flush_c_stdio()
            print(json.dumps(headline), flush=True)
        os._exit(0)

    signal.signal(signal.SIGALRM, give_up)
    signal.alarm(int(args.leg_timeout))
    try:
        if kind == "fmx":
            st, wu = (args.fmx_leg_steps, 2) if config == 3 else (2, 1)
            leg = fmx_leg(args, ctx, config, st, wu, cpu_baseline=True, cpu_budget_s=5.0)
        elif kind == "demux_strong":
            leg = demux_leg(args, ctx, 2, steps=3, warmup=1, ramp_seconds=0.0, cpu_budget_s=5.0, strong=True)
        else:
            leg = demux_leg(args, ctx, config, steps=3, warmup=1, ramp_seconds=0.0, cpu_budget_s=5.0)
        signal.alarm(0)
        return leg
    except BaseException as ex:  # noqa: BLE001 -- includes a failed collective on this rank
        sys.stderr.write(f"[bench rank {ctx.rank}] {key} failed: {ex!r}\n")
        if ctx.world == 1:
            signal.alarm(0)
            free_torch_cache(ctx)
            return {"error": repr(ex)}
        while True:  # the other ranks may be waiting for us in a collective: leave together, when the alarm fires
            time.sleep(1.0)


def flush_c_stdio():
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 2000 / 3 / 20 / 5 for configs 1 / 2 / 3 / 4")
    ap.add_argument("--warmup", type=int, default=None, help="default: 200 / 1 / 2 / 1")
    ap.add_argument("--ramp-seconds", type=float, default=1.0,
                    help="untimed passes before the W warmup steps, until the engine clock has ramped up (a 0.6 ms step "
                         "repeated 20 times runs ~10 %% below the sustained clock)")
    ap.add_argument("--config", type=int, default=1, help="index into BASELINE.json configs: 1, 2 demuxlet; 3, 4 freemuxlet")
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the config's cells (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", "--no-fmx-leg", dest="no_legs", action="store_true",
                    help="default run: only the headline (configs[1]), none of the secondary legs")
    ap.add_argument("--legs", default=None,
                    help="default run: secondary legs, in order: BASELINE.json config indices 3, 2, 4 and, with N > 1 ranks, "
                         "2s = configs[2] as ONE job cut over the ranks (strong scaling).  Default 3,2,4 (N = 1) / "
                         "3,2,2s,4 (N > 1); given explicitly, the legs also run at --scale != 1 (tests)")
    ap.add_argument("--fmx-leg-steps", type=int, default=20, help="EM iterations of the configs[3] leg")
    ap.add_argument("--leg-timeout", "--fmx-leg-timeout", dest="leg_timeout", type=float, default=300.0,
                    help="watchdog per secondary leg, seconds")
    ap.add_argument("--synth", default="auto", choices=("auto", "numpy", "device"),
                    help="input generator: numpy (seeded, the tests' generator), device (torch on the GPU), "
                         "auto = device beyond 2e7 entries")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for tests)")
    ap.add_argument("--force-dist", action="store_true",
                    help="freemuxlet: run the N-rank code path (slabs, collectives, stream ordering) even with one rank")
    ap.add_argument("--single-device", action="store_true",
                    help="functional test of the N>1 path on a 1-GPU box: every rank uses device 0 (use with gloo)")
    ap.add_argument("--cells", type=int, default=0, help="freemuxlet: override the cell count (tests)")
    ap.add_argument("--snps", type=int, default=0, help="freemuxlet: override the SNP count (tests)")
    ap.add_argument("--clusters", type=int, default=0, help="freemuxlet: override K (tests)")
    ap.add_argument("--mean-entries", type=float, default=800.0)
    ap.add_argument("--dump", default="", help="freemuxlet: write rank 0's final records (tests)")
    ap.add_argument("--dense", action="store_true",
                    help="demuxlet sensitivity run: the same shape with 1 + Poisson(2.0) reads per entry (one entry in seven "
                         "linear instead of three quarters); adds a 'sensitivity' key to the line")
    ap.add_argument("--reads-lambda", type=float, default=None, help="reads per entry = 1 + Poisson(lambda) (default 0.3)")
    ap.add_argument("--no-linear", action="store_true",
                    help="demuxlet: MUXGL_FLAG_NO_LINEAR_ENTRIES -- every entry through the general three-term form")
    args = ap.parse_args()
    if args.config not in synth.CONFIGS:
        raise SystemExit("--config must be 1, 2 (demuxlet) or 3, 4 (freemuxlet)")
    if args.steps is None:
        args.steps = {1: 2000, 2: 3, 3: 20, 4: 5}[args.config]
    if args.warmup is None:
        args.warmup = {1: 200, 2: 1, 3: 2, 4: 1}[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    ctx = Ctx(args)
    if args.config in (1, 2):
        out = demux_leg(args, ctx, args.config)
        legs = args.legs if args.legs is not None else ("3,2,2s,4" if ctx.world > 1 else "3,2,4")
        if args.config == 1 and not args.no_legs and (args.scale == 1.0 or args.legs is not None):
            for cfg in [x.strip() for x in legs.split(",") if x.strip()]:
                cfg = int(cfg) if cfg.isdigit() else cfg
                if cfg not in LEGS:
                    raise SystemExit("--legs: a comma-separated subset of 3,2,2s,4")
                if cfg == "2s" and ctx.world == 1:
                    continue
                leg = ctx.stamp(guarded_leg(args, ctx, out, cfg))
                if ctx.rank == 0:
                    out[LEGS[cfg][0]] = leg
    else:
        out = fmx_leg(args, ctx, args.config, args.steps, args.warmup)
    if ctx.dist_on:  # RCCL writes a version banner through C stdio: get it out of the way, the JSON line comes last
        ctx.barrier()
        flush_c_stdio()
    if ctx.rank == 0:
        print(json.dumps(ctx.stamp(out)), flush=True)
    if ctx.dist_on:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
