"""Host-side driver of the freemuxlet EM loop -- the sequential control flow of cmdCramFreemux2
(cmd_cram_freemux2.cpp:373-605) around the libmuxgl phases, for one GPU or for one process per GPU.

Multi-GPU scheme (exact).  Rank r holds two slabs of the packed pileup, 2/N of it in all:
its ROW slab (cells c_ranges[r], every SNP: E-step, scans, re-assignment) and its COLUMN slab (every cell, SNPs
s_ranges[r]: cluster pileups, cluster-GP rows, ordered M-step).  Per iteration:

    iter_gp     -> all-gather of the cluster-GP rows   f64[S][K][3]   (38 MB at config 3, 768 MB at config 4)
    iter_estep  -> all-gather of the assignments       i32[C]         + all-reduce of (nsingle, namb, nchanged)
    iter_mstep

Each exchange is ONE collective (all_gather_into_tensor, in place on the library's own device buffer: the ranges are
equal slices and the buffers carry a little slack).  Over RCCL (backend "nccl") nothing on the host waits between the
phases: the library enqueues on its own stream (MUXGL_FLAG_ASYNC_PHASES), that stream is torch's current stream while
the collectives are issued, and ProcessGroupNCCL orders its communication stream against it with events; the host
blocks once per iteration, on the three reduced counters the reference's early stop reads (:601-604), while the M-step
is already running.  An all-reduce of sufficient statistics would NOT reproduce the reference: merge() clamps after
every cell (sc_drop_seq.h:92-100), so the chain per (cluster, SNP) is evaluated by exactly one rank, in ascending cell id.
"""
from __future__ import annotations

import contextlib
import time

import numpy as np

from . import shard

UNIT_CGP, UNIT_CLUST, UNIT_STAT = 0, 1, 2


class TorchExchange:
    """Collectives over torch.distributed on tensors that alias the engine's exchange buffers.

    device_ordered=True (RCCL): the collectives are enqueued behind the engine's stream and nothing waits on the host.
    device_ordered=False (gloo staging, tests on a 1-GPU box or on CPU): every exchange returns when the data has landed."""

    def __init__(self, dist_module, rank: int, world: int, device_ordered: bool = False, always: bool = False):
        self.dist = dist_module
        self.rank = rank
        self.world = world
        self.device_ordered = device_ordered
        self.always = always  # take the exchange path even with one rank

    def allgather_equal(self, tensor, per: int):
        """tensor: [>= world*per, row]; rank r owns rows [r*per, (r+1)*per).  One collective, in place."""
        if per <= 0:
            return
        if self.world * per > tensor.shape[0]:  # slices of ceil(n / world) units overrun the buffer's XCHG_PAD slack
            raise ValueError(f"in-place all-gather of {self.world} x {per} rows needs {self.world * per} rows, the exchange "
                             f"buffer has {tensor.shape[0]} (slack of muxgl.XCHG_PAD = 64 units: at most 65 ranks)")
        out = tensor[: self.world * per]
        inp = tensor[self.rank * per:(self.rank + 1) * per]
        self.dist.all_gather_into_tensor(out, inp)
        self._landed(tensor)

    def allreduce_sum(self, tensor):
        self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM)
        self._landed(tensor)

    def _landed(self, tensor):
        if not self.device_ordered and getattr(tensor, "is_cuda", False):
            import torch

            torch.cuda.current_stream(tensor.device).synchronize()

    def gather_objects(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


class NoExchange:
    rank, world, device_ordered = 0, 1, False

    def allgather_equal(self, tensor, per):
        pass

    def allreduce_sum(self, tensor):
        pass

    def gather_objects(self, obj):
        return [obj]


def engine_exchange_tensor(eng, which, device_index=0):
    """torch tensor aliasing a libmuxgl device buffer (zero-copy, __cuda_array_interface__), slack included:
    CGP -> f64[S + pad, K*3], CLUST -> i32[C_total + pad, 1], STAT -> i32[4]"""
    import torch

    from . import muxgl

    buf = {UNIT_CGP: muxgl.BUF_CGP, UNIT_CLUST: muxgl.BUF_CLUST, UNIT_STAT: muxgl.BUF_STAT}[which]
    ptr, n = eng.fmx_buffer(buf)
    row = {UNIT_CGP: eng.K * 3, UNIT_CLUST: 1, UNIT_STAT: 1}[which]
    n_all = n + (muxgl.XCHG_PAD * row if which != UNIT_STAT else 0)
    typestr, dtype = ("<f8", torch.float64) if which == UNIT_CGP else ("<i4", torch.int32)

    class _Wrap:
        __cuda_array_interface__ = {"shape": (int(n_all),), "typestr": typestr, "data": (int(ptr), False), "version": 2}

    t = torch.as_tensor(_Wrap(), device=torch.device("cuda", device_index))
    assert t.dtype == dtype and t.data_ptr() == ptr
    return t if which == UNIT_STAT else t.view(-1, row)


def plan_ranges(C: int, S: int, world: int):
    """((cell ranges, cells per rank), (SNP ranges, SNPs per rank)): equal slices, see shard.equal_ranges"""
    return shard.equal_ranges(C, world), shard.equal_ranges(S, world)


def load_rank(eng, p, c_range, s_range):
    """Hand-over of one rank's share of the pileup p: row slab, column slab, entry likelihoods.  Returns the singlet
    scores (llk0, llk2, nsnps, nreads) of the rank's own cells.  (A loader that never materialises the whole pileup on a
    rank would cut the same two slabs at file level: .plp.gz rows are sorted by SNP, then droplet.)"""
    c0, c1 = c_range
    s0, s1 = s_range
    sub = shard.take_cells(p, c0, c1)
    eng.set_pileup(p.S, sub.cell_ptr, sub.entry_snp, sub.entry_rptr, sub.reads)
    eng.fmx_set_column_slab(p.C, c0, s0, s1, *shard.take_snps(p, s0, s1))
    return eng.fmx_prepare(p.af)


def load_rank_from_files(eng, exe, plp_prefix, rank, world, scratch, extra=()):
    """load_rank() from a dsc-pileup data set on disk: the C++ loader (`popscle-amd dump-plp --rank r --world N`) keeps
    only this rank's two slabs while it parses the .plp.gz -- the whole pileup is never held by a rank, on the host or
    on the device.  Returns (scores of the own cells, the slab dictionary)."""
    import subprocess

    from . import plpio

    r = subprocess.run([exe, "dump-plp", "--plp", plp_prefix, "--out", scratch, "--rank", str(rank), "--world", str(world),
                        *extra], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    d = plpio.read_slab_dump(scratch)
    eng.set_pileup(d["S"], *d["rows"])
    eng.fmx_set_column_slab(d["C"], d["c0"], d["s0"], d["s1"], *d["cols"])
    return eng.fmx_prepare(d["af"]), d


def settle_near_ties(engs, gather, doublet_prior, geno_error):
    """The exact path for the calls rounding noise could decide (include/muxgl.h, csrc/fmx_exact.hip), across ranks:
    the SNP lists of the ranks' listed cells are united, every cluster-posterior row is recomputed exactly by the rank
    whose M-step range holds the SNP, all ranks get all rows, and every rank settles its own cells.

    engs: the engines this process drives (one per rank in a real run; several virtual ranks in tests).
    gather(obj) -> list of obj over ALL ranks of the job (identity-like for a process that drives every rank).
    Returns True when an assignment changed anywhere (assignments must be exchanged again, the M-step repeated)."""
    lists = [e.fmx_exact_snps() for e in engs]
    every = [x for part in gather(lists) for x in part]
    uni = np.unique(np.concatenate(every)).astype(np.int32) if every else np.zeros(0, dtype=np.int32)
    K = engs[0].K
    mine = []
    for e in engs:
        rows, owned = e.fmx_exact_rows(uni, doublet_prior, geno_error)
        mine.append((np.flatnonzero(owned), rows[owned]))
    full = np.zeros((uni.size, K, 3))
    seen = np.zeros(uni.size, dtype=bool)
    for part in gather(mine):
        for idx, r in part:
            full[idx] = r
            seen[idx] = True
    if not seen.all():
        raise RuntimeError("near-tie calls: no rank's M-step range holds SNP(s) " + str(uni[~seen][:5].tolist()))
    re = [e.fmx_exact_finish(uni, full, doublet_prior, geno_error)[1] for e in engs]
    return any(x for part in gather(re) for x in part)


def run_em(eng, K, clust0, doublet_prior=0.5, geno_error=0.1, max_iter=10, early_stop=True, exchange=None,
           exchange_tensor=engine_exchange_tensor, log=None, per=None, timings=None, sync=None, stream_ctx=None):
    """EM loop of cmd_cram_freemux2.cpp:373-605 on a prepared engine.  One rank: a plain engine holding the whole
    pileup.  Several ranks: an engine holding the rank's slabs (load_rank), `exchange` a TorchExchange and
    per = (cells per rank, SNPs per rank) of the equal-slice plan the slabs were cut by.  clust0 spans the whole job.
    Returns (records of ALL cells, complete on every rank; per-iteration stats).  stream_ctx: context manager that
    makes the engine's stream torch's current one (device-ordered exchanges)."""
    ex = exchange or NoExchange()
    t_start = time.perf_counter()
    eng.fmx_set_clusters(K, np.ascontiguousarray(clust0, dtype=np.int32))  # :277-288, own SNP range
    multi = ex.world > 1 or getattr(ex, "always", False)  # (always: the exchange path with a single rank, for tests)
    if multi:
        per_c, per_s = per
        t_cgp, t_clust = exchange_tensor(eng, UNIT_CGP), exchange_tensor(eng, UNIT_CLUST)
        t_stat = exchange_tensor(eng, UNIT_STAT)
    ordered = multi and ex.device_ordered
    if ordered:
        import torch

        stat_host = torch.zeros(4, dtype=torch.int32).pin_memory()
        stat_ready = torch.cuda.Event()
    history = []
    # device-ordered exchanges are timed with events on the engine's stream (the collective's own stream is ordered
    # against it on both sides): what the scaling model of DESIGN.md 4.3 is compared with
    marks = [] if (ordered and timings is not None) else None

    def mark():
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)

    if sync:
        sync()
    t_loop = time.perf_counter()
    with (stream_ctx() if stream_ctx else contextlib.nullcontext()):
        for it in range(max_iter):
            if not multi:  # one handle, whole pileup: the three phases as ONE call (no host round trip between them)
                _, stats = eng.fmx_iterate(doublet_prior, geno_error, want_cells=False)
                history.append(stats)
                if log:
                    log(f"iter {it + 1}: {stats[0]} singlets, {eng.C_total - stats[0] - stats[1]} doublets, "
                        f"{stats[1]} ambiguous, {stats[2]} changed")
                if stats[2] == 0 and early_stop:  # :601-604
                    break
                continue
            eng.fmx_iter_gp(doublet_prior, geno_error)
            if multi:
                mark()
                ex.allgather_equal(t_cgp, per_s)
                mark()
            eng.fmx_iter_estep(doublet_prior, geno_error)
            if multi:
                mark()
                ex.allgather_equal(t_clust, per_c)
                ex.allreduce_sum(t_stat)
                mark()
            if ordered:  # counters on their way to the host; the M-step is enqueued before anybody waits for them
                stat_host.copy_(t_stat, non_blocking=True)
                stat_ready.record()
                eng.fmx_iter_mstep()
                stat_ready.synchronize()
                stats4 = tuple(int(x) for x in stat_host[:4].tolist())
            else:
                stats4 = tuple(int(x) for x in t_stat[:4].tolist())
            if stats4[3] > 0:
                # cells whose call is within rounding reach of the kernels' numbers, somewhere in the job: settled in the
                # reference's arithmetic (two small exchanges, this iteration only), then the assignments and the counters
                # once more, and the ordered merge from the corrected assignments
                reassigned = settle_near_ties([eng], ex.gather_objects, doublet_prior, geno_error)
                ex.allgather_equal(t_clust, per_c)
                ex.allreduce_sum(t_stat)
                stats4 = tuple(int(x) for x in t_stat[:4].tolist())
                if reassigned or not ordered:
                    eng.fmx_iter_mstep()
            elif not ordered:
                eng.fmx_iter_mstep()  # :516-517 + :590-596 for the own SNP range
            stats = stats4[:3]
            if hasattr(eng, "fmx_exact_hint"):
                eng.fmx_exact_hint(stats[2])  # nothing moved: the cells settled so far need not be listed again
            history.append(stats)
            if log:
                log(f"iter {it + 1}: {stats[0]} singlets, {eng.C_total - stats[0] - stats[1]} doublets, "
                    f"{stats[1]} ambiguous, {stats[2]} changed")
            if stats[2] == 0 and early_stop:  # :601-604
                break
    if sync:
        sync()
    t_end = time.perf_counter()
    if timings is not None:  # `sync` (a barrier) makes these comparable across ranks
        timings.update(setup_s=t_loop - t_start, loop_s=t_end - t_loop, iterations=len(history))
        if marks:
            torch.cuda.synchronize()
            gp = [marks[i].elapsed_time(marks[i + 1]) for i in range(0, len(marks), 4)]
            cl = [marks[i + 2].elapsed_time(marks[i + 3]) for i in range(0, len(marks), 4)]
            timings.update(exchange_ms={"cluster_gp_allgather": sum(gp) / len(gp),
                                        "assignments_allgather_and_counters": sum(cl) / len(cl)})
    cells, _ = eng.fmx_iter_fetch()  # the rank's own cells
    c0 = eng.cell_base
    parts = ex.gather_objects((c0, c0 + len(cells), cells.tobytes()))
    out = np.zeros(eng.C_total, dtype=cells.dtype)
    for b, e, raw in parts:
        out[b:e] = np.frombuffer(raw, dtype=cells.dtype)
    return out, history
