"""A pileup as `freemuxlet` meets it when nobody filters barcodes: a few thousand cells among many droplets of one to a
handful of reads.  The droplets' singlet scores are 0 +- rounding noise and many of their greedy / EM decisions are
noise-level ties: the start's order, the greedy pass and every iteration must still be the reference's
(oracle/_ref/libscdrop_ref.so), and the exact paths must not take over the run time.

    python tests/stress_droplets.py [cells] [droplets] [K] [S]      (prints one JSON line)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import parity  # noqa: E402
import ref_binding as rb  # noqa: E402
from popscle_amd import muxgl, synth  # noqa: E402


def mixed(cells, droplets, K, S, seed=77, with_gp=False):
    a = synth.make_pileup(cells, S, K, seed=seed, mean_entries=800, with_gp=with_gp, donor_seed=seed)
    b = synth.make_pileup(droplets, S, K, seed=seed + 1, mean_entries=2.5, sigma=0.9, min_entries=1, with_gp=False,
                          donor_seed=seed, reads_lambda=0.1, other=0.01)
    rng = np.random.default_rng(seed + 2)
    order = rng.permutation(cells + droplets)          # droplets and cells interleaved, as barcodes sort
    lens = np.concatenate([np.diff(a.cell_ptr), np.diff(b.cell_ptr)])[order]
    starts = np.concatenate([a.cell_ptr[:-1], b.cell_ptr[:-1] + a.nnz])[order]
    esnp = np.concatenate([a.entry_snp, b.entry_snp])
    nre = np.concatenate([np.diff(a.entry_rptr), np.diff(b.entry_rptr)])
    rstart = np.concatenate([a.entry_rptr[:-1], b.entry_rptr[:-1] + a.R])
    reads = np.concatenate([a.reads, b.reads])
    eidx = synth._ranges(starts, lens)
    cell_ptr = np.zeros(order.size + 1, dtype=np.int64)
    np.cumsum(lens, out=cell_ptr[1:])
    rl = nre[eidx]
    entry_rptr = np.zeros(eidx.size + 1, dtype=np.int64)
    np.cumsum(rl, out=entry_rptr[1:])
    ridx = synth._ranges(rstart[eidx], rl)
    return synth.Pileup(order.size, S, cell_ptr, esnp[eidx].astype(np.int32), entry_rptr, reads[ridx], a.af, a.gp, a.has_gp,
                        {"G": a.truth["G"]})


def main_demux(argv):
    """demuxlet over the same shape (V samples): records against the reference's, and what the exact-call pass costs"""
    cells, droplets, V, S = (int(x) for x in (argv + ["3000", "30000", "16", "30000"][len(argv):]))
    p = mixed(cells, droplets, V, S, with_gp=True)
    out = dict(kind="demuxlet", cells=cells, droplets=droplets, V=V, S=S, nnz=int(p.nnz))
    t = time.time()
    want, _, _ = rb.RefScl.from_packed(p).demux((0.0, 0.5), doublet_prior=0.5)
    out["reference_s"] = round(time.time() - t, 2)
    eng = muxgl.Engine(0)
    t0 = time.time()
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    eng.demux_set_gp(p.gp, p.has_gp)
    out["handover_s"] = round(time.time() - t0, 3)
    t = time.time()
    got = eng.demux_run((0.0, 0.5), 0.5)
    out["run_s"] = round(time.time() - t, 4)
    t = time.time()
    try:
        rep = parity.compare_demux(got, want, (0.0, 0.5), p)
        out.update(exact_pass=rep["exact_pass"], raw_records_differing=rep["raw_records_differing"],
                   max_abs_ll_diff=rep["max_abs_ll_diff"], differing=0)
    except AssertionError as ex:
        out.update(differing=1, first_error=str(ex)[:300])
    out["exact_pass_and_compare_s"] = round(time.time() - t, 3)
    print(json.dumps(out, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o)), flush=True)
    return out["differing"]


def main(argv):
    if argv and argv[0] == "demux":
        return main_demux(argv[1:])
    cells, droplets, K, S = (int(x) for x in (argv + ["3000", "30000", "8", "30000"][len(argv):]))
    p = mixed(cells, droplets, K, S)
    out = dict(cells=cells, droplets=droplets, K=K, S=S, nnz=int(p.nnz))
    t = time.time()
    ref = rb.RefScl.from_packed(p).freemux2(K)
    out["reference_s"] = round(time.time() - t, 2)
    eng = muxgl.Engine(0)
    t = time.time()
    eng.set_pileup(p.S, p.cell_ptr, p.entry_snp, p.entry_rptr, p.reads)
    out["handover_s"] = round(time.time() - t, 3)
    t = time.time()
    llk0, llk2, ns, nr = eng.fmx_prepare(p.af)
    out["prepare_s"] = round(time.time() - t, 3)
    out["exact_scores"] = int(eng.fmx_score_stats())
    t = time.time()
    clust = eng.fmx_greedy_init(K, llk2 - llk0)
    out["greedy_s"] = round(time.time() - t, 3)
    out["greedy_near_overruled"] = [int(x) for x in eng.fmx_greedy_stats()]
    out["greedy_cells_differing"] = int((clust != ref["clust0"]).sum())
    t = time.time()
    eng.fmx_set_clusters(K, clust)
    out["set_clusters_s"] = round(time.time() - t, 3)
    em = 0.0
    bad = 0
    for it in range(ref["n_iter"]):
        t = time.time()
        cellsr, st = eng.fmx_iterate(0.5, 0.1)   # (records of all cells copied to the host included)
        em += time.time() - t
        try:
            parity.compare_fmx(cellsr, ref["cells"][it])
            assert tuple(st) == tuple(ref["counters"][it]), (st, ref["counters"][it])
        except AssertionError as ex:
            bad += 1
            out.setdefault("first_error", f"iteration {it}: {str(ex)[:300]}")
    out["em_s"] = round(em, 3)
    out["iterations"] = int(ref["n_iter"])
    out["iterations_differing"] = bad
    out["em_exact"] = [int(x) for x in eng.fmx_exact_stats()]
    # the library's calls only (the comparisons with the reference's records are not in it)
    out["device_total_s"] = round(out["handover_s"] + out["prepare_s"] + out["greedy_s"] + out["set_clusters_s"] + em, 3)
    print(json.dumps(out), flush=True)
    return 1 if (bad or out["greedy_cells_differing"]) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
