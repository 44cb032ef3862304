"""End-to-end GPU tests of the popscle-compatible front end: files of the real CEL/VAR/PLP (+VCF) format in,
`.best` / `.lmix` / `.clust1.samples.gz` / `.clust1.vcf.gz` out, compared with rows formatted from the CPU oracle with
the reference's printf formats (cmd_cram_demuxlet.cpp:993-1013, cmd_cram_freemux2.cpp:161,660-665)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import pyplp
from popscle_amd import plpio, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "popscle_amd", "bin", "popscle-amd")
TYPES = {0: "SNG", 1: "DBL", 2: "AMB"}


def tokens_match(a, b):
    """exact, or numerically equal up to one unit in the last printed digit (a 1e-11 LL difference may straddle a
    rounding boundary of %.2lf)"""
    if a == b:
        return True
    pa, pb = a.split(","), b.split(",")
    if len(pa) != len(pb):
        return False
    for x, y in zip(pa, pb):
        if x == y:
            continue
        try:
            fx, fy = float(x), float(y)
        except ValueError:
            return False
        if abs(fx - fy) > 0.0101 * max(1.0, min(abs(fx), abs(fy)) if "e" in x.lower() else 1.0):
            return False
    return True


def assert_rows_match(got_lines, want_lines):
    assert len(got_lines) == len(want_lines)
    for g, w in zip(got_lines, want_lines):
        gt, wt = g.rstrip("\n").split("\t"), w.rstrip("\n").split("\t")
        assert len(gt) == len(wt), (g, w)
        for i, (a, b) in enumerate(zip(gt, wt)):
            assert tokens_match(a, b), f"column {i}: {a!r} vs {b!r}\n got: {g}\nwant: {w}"


def as_pileup(d):
    return synth.Pileup(d["C"], d["S"], d["cell_ptr"], d["entry_snp"], d["entry_rptr"], d["reads"], d["af"],
                        d["gp"] if d["nv"] else None, d["has_gp"] if d["nv"] else None)


@pytest.mark.parametrize("V,alphas,field,shape", [(v, a, f, "cells") for v, a, f in [
    (4, None, "GT"),   # BASELINE configs[0]'s shape: 4-sample GT VCF, default grid
    (6, None, "GT"), (6, (0.0, 0.1, 0.3, 0.5), "GT"), (20, None, "GT"),
    # posteriors from FORMAT/GP (float normalisation) and from FORMAT/PL (the 10-iteration EM); the
    # loader's arithmetic for both is pinned to the reference's own lines in tests/test_oracle_ref.py
    (6, None, "GP"), (6, None, "PL"), (4, (0.0, 0.25, 0.5), "PL"), (20, None, "GP")]] + [
    # unfiltered barcodes: cells among droplets of one to a handful of reads (most hypotheses of a droplet tie exactly)
    (6, None, "GT", "droplets"), (16, None, "GP", "droplets")])
def test_demuxlet_cli(tmp_path, V, alphas, field, shape):
    if shape == "droplets":
        import stress_droplets

        p = stress_droplets.mixed(40, 700, V, 800, seed=5, with_gp=True)
    else:
        p = synth.make_pileup(60, 800, V, seed=5, mean_entries=150, min_entries=20, doublet_frac=0.3)
    prefix = str(tmp_path / "plp")
    plpio.write_plp(prefix, p, seed=5, extra_cells=1)
    vcf = str(tmp_path / "g.vcf.gz")
    G = p.truth["G"].astype(np.int64)
    rng = np.random.default_rng(V)
    onehot = np.eye(3)[G]
    gp = 0.85 * onehot + 0.15 * rng.dirichlet([0.5, 0.5, 0.5], size=G.shape)   # imputation-like posteriors
    pl = np.where(onehot > 0, 0, rng.integers(8, 70, size=G.shape + (3,)))       # caller-like likelihoods
    plpio.write_vcf(vcf, p, G, field=field, missing_frac=0.02, drop_snps=range(0, 800, 37), gp=gp, pl=pl)
    out = str(tmp_path / "out")
    cmd = [BIN, "demuxlet", "--plp", prefix, "--vcf", vcf, "--field", field, "--out", out]
    for a in alphas or ():
        cmd += ["--alpha", str(a)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    al = alphas or (0.0, 0.5)
    d = pyplp.load(prefix, vcf=vcf, field=field)
    if field != "GT":
        hard = pyplp.load(prefix, vcf=vcf, field="GT")
        assert not np.array_equal(hard["gp"], d["gp"])   # the run really read another FORMAT key
    q = as_pileup(d)
    cells = ob.demux(q, alphas=al, doublet_prior=0.5)
    ids = [f"S{v}" for v in range(V)]
    want = ["INT_ID\tBARCODE\tNUM.SNPS\tNUM.READS\tDROPLET.TYPE\tBEST.GUESS\tBEST.LLK\tNEXT.GUESS\tNEXT.LLK\t"
            "DIFF.LLK.BEST.NEXT\tBEST.POSTERIOR\tSNG.POSTERIOR\tSNG.BEST.GUESS\tSNG.BEST.LLK\tSNG.NEXT.GUESS\t"
            "SNG.NEXT.LLK\tSNG.ONLY.POSTERIOR\tDBL.BEST.GUESS\tDBL.BEST.LLK\tDIFF.LLK.SNG.DBL\n"]
    order = sorted(range(d["C"]), key=lambda i: d["bcs"][i].encode())
    for rank, i in enumerate(order):
        c = cells[i]
        if not c["valid"]:
            continue
        want.append("%d\t%s\t%u\t%d\t%s\t%s,%s,%.2f\t%.2f\t%s,%s,%.2f\t%.2f\t%.2f\t%.2g\t%.2g\t%s\t%.2f\t%s\t%.2f\t%.5f\t"
                    "%s,%s,%.2f\t%.2f\t%.2f\n" % (
                        rank, d["bcs"][i], c["nsnps"], d["cell_uniq_reads"][i], TYPES[int(c["type"])],
                        ids[c["jBest"]], ids[c["kBest"]], al[c["aBest"]], c["bestLLK"], ids[c["jNext"]], ids[c["kNext"]],
                        al[c["aNext"]], c["nextLLK"], c["bestLLK"] - c["nextLLK"], c["bestPP"], c["sngPP"],
                        ids[c["sBest"]], c["sngBestLLK"], ids[c["sNext"]], c["sngNextLLK"], c["sngOnlyPP"],
                        ids[c["dBest1"]], ids[c["dBest2"]], al[c["dBestA"]], c["dblBestLLK"],
                        c["sngBestLLK"] - c["dblBestLLK"]))
    got = open(out + ".best").readlines()

    # no canonicalisation: the front end settles the order of a mirrored alpha-0.5 pair and every near-tie call as the
    # reference does (popscle_amd/host/exact_calls.hpp), so every guess column is compared as printed
    assert_rows_match(got, want)
    assert len(got) == 1 + int((cells["valid"] == 1).sum())


@pytest.mark.parametrize("shape", ["cells", "droplets"])
def test_freemuxlet_cli(tmp_path, shape):
    K = 4
    if shape == "cells":
        p = synth.make_pileup(150, 1200, K, seed=8, mean_entries=200, min_entries=30, with_gp=False)
    else:
        # unfiltered barcodes: a hundred cells among 1 500 droplets of one to a handful of reads -- start scores of
        # 0 +- rounding noise, noise-level ties in the greedy pass and in the iterations (tests/stress_droplets.py)
        import stress_droplets

        p = stress_droplets.mixed(100, 1500, K, 5000, seed=31)
    prefix = str(tmp_path / "plp")
    plpio.write_plp(prefix, p, seed=8)
    out = str(tmp_path / "out")
    # (POPSCLE_AMD_CHECK_FORMAT: every sample field of the cluster VCF is also printed with the reference's own
    #  conversion specification and compared with what the writer's fast formatting produced)
    r = subprocess.run([BIN, "freemuxlet", "--plp", prefix, "--nsample", str(K), "--out", out, "--seed", "1"],
                       env=dict(os.environ, POPSCLE_AMD_CHECK_FORMAT="1"),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = pyplp.load(prefix)
    q = as_pileup(d)
    e = ob.fmx_entry_pileup(q)
    llk0, llk2, ns, nr = ob.fmx_cell_scores(q, e)
    want = ["INT_ID\tBARCODE\tNSNPs\tNREADs\tDBL.LLK\tSNG.LLK\tBF.SINGLET\tBF.SINGLET.PER.SNP\n"]
    for i in range(q.C):
        want.append("%d\t%s\t%d\t%d\t%.2f\t%.2f\t%.2f\t%.4f\n" % (i, d["bcs"][i], ns[i], nr[i], llk0[i], llk2[i],
                                                                 llk2[i] - llk0[i], (llk2[i] - llk0[i]) / ns[i]))
    assert_rows_match(open(out + ".lmix").readlines(), want)

    clust = ob.fmx_greedy_init(q, e, K, llk2 - llk0, ob.fmx_sort(llk2 - llk0))
    cplp = ob.fmx_build_cluster_pileup(q, e, K, clust)
    cells = ob.fmx_init_cells(clust)
    for _ in range(10):
        st = ob.fmx_iterate(q, e, K, cplp, cells)
        if st[2] == 0:
            break
    want = ["INT_ID\tBARCODE\tNUM.SNPS\tNUM.READS\tDROPLET.TYPE\tBEST.GUESS\tBEST.LLK\tNEXT.GUESS\tNEXT.LLK\t"
            "DIFF.LLK.BEST.NEXT\tBEST.POSTERIOR\tSNG.POSTERIOR\tSNG.BEST.GUESS\tSNG.BEST.LLK\tSNG.NEXT.GUESS\t"
            "SNG.NEXT.LLK\tSNG.ONLY.POSTERIOR\tDBL.BEST.GUESS\tDBL.BEST.LLK\tDIFF.LLK.SNG.DBL\n"]
    for i in range(q.C):
        c = cells[i]
        want.append("%d\t%s\t%d\t%d\t%s\t%d,%d\t%.2f\t%d,%d\t%.2f\t%.2f\t%.5f\t%.2g\t%d\t%.2f\t%d\t%.2f\t%.5f\t%d,%d\t%.2f\t"
                    "%.2f\n" % (i, d["bcs"][i], ns[i], nr[i], TYPES[int(c["type"])], c["jBest"], c["kBest"], c["bestLLK"],
                                c["jNext"], c["kNext"], c["nextLLK"], c["bestLLK"] - c["nextLLK"], c["bestPP"],
                                c["sngPP"], c["sBest"], c["sngBestLLK"], c["sNext"], c["sngNextLLK"], c["sngOnlyPP"],
                                c["dBest1"], c["dBest2"], c["dblBestLLK"], c["sngBestLLK"] - c["dblBestLLK"]))
    with gzip.open(out + ".clust1.samples.gz", "rt") as f:
        assert_rows_match(f.readlines(), want)

    # .clust1.vcf.gz: header shape, one record per observed SNP, counts and PL/GP from the oracle's cluster pileups
    with gzip.open(out + ".clust1.vcf.gz", "rt") as f:
        lines = f.readlines()
    body = [ln for ln in lines if not ln.startswith("#")]
    observed = np.unique(q.entry_snp)
    assert len(body) == observed.size
    assert lines[0] == "##fileformat=VCFv4.2\n" and lines[2] == "##source=cramore-freemuxlet\n"
    for ln, v in list(zip(body, observed))[::37]:
        t = ln.rstrip("\n").split("\t")
        assert t[1] == str(1000 + 10 * int(v)) and t[8] == "GT:GQ:DP:AD:PL:GP" and len(t) == 9 + K
        for k in range(K):
            f_ = t[9 + k].split(":")
            assert int(f_[2]) == cplp["nreads"][k, v]
            assert f_[3] == f"{cplp['nref'][k, v]},{cplp['nalt'][k, v]}"
            g = cplp["gls"][k, v]
            mx = max(g[0], g[4], g[8])
            pls = [int(-10.0 * np.log10(x / mx)) for x in (g[0], g[4], g[8])]
            got_pl = [int(x) for x in f_[4].split(",")]
            assert all(abs(a - b) <= 1 for a, b in zip(got_pl, pls))  # (int) truncation next to an integer boundary


def test_freemuxlet_cli_init_cluster(tmp_path):
    K = 3
    p = synth.make_pileup(80, 600, K, seed=12, mean_entries=150, min_entries=30, with_gp=False)
    prefix = str(tmp_path / "plp")
    bcs = plpio.write_plp(prefix, p, seed=12)
    init = tmp_path / "init.txt"
    with open(init, "w") as f:
        for i, b in enumerate(bcs):
            if i % 9:
                f.write(f"{b}\t{int(p.truth['s1'][i])}\n")
    out = str(tmp_path / "out")
    r = subprocess.run([BIN, "freemuxlet", "--plp", prefix, "--nsample", str(K), "--out", out, "--init-cluster",
                        str(init), "--aux-files"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with gzip.open(out + ".clust0.samples.gz", "rt") as f:
        rows = f.readlines()[1:]
    got0 = np.array([int(r_.split("\t")[2]) for r_ in rows])
    want0 = np.where(np.arange(p.C) % 9 != 0, p.truth["s1"], -1)
    assert np.array_equal(got0, want0)
    assert os.path.exists(out + ".clust0.vcf.gz") and os.path.exists(out + ".clust1.samples.gz")
    with gzip.open(out + ".clust1.samples.gz", "rt") as f:
        final = f.readlines()[1:]
    types = [r_.split("\t")[4] for r_ in final]
    sng = np.array([t == "SNG" for t in types])
    best = np.array([int(r_.split("\t")[5].split(",")[0]) for r_ in final])
    ok = sng & ~p.truth["is_doublet"]
    assert ok.sum() > 0.7 * p.C and np.array_equal(best[ok], p.truth["s1"][ok])


def test_cli_errors(tmp_path):
    r = subprocess.run([BIN, "freemuxlet", "--plp", "nowhere", "--out", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode != 0 and "Missing required option" in r.stderr
    r = subprocess.run([BIN, "demuxlet", "--plp", str(tmp_path / "nowhere"), "--vcf", str(tmp_path / "no.vcf"), "--out",
                        str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode != 0 and "FATAL ERROR" in r.stderr


def _read(path):
    if path.endswith(".gz"):
        with gzip.open(path, "rt") as f:
            return [ln for ln in f.readlines() if not ln.startswith("##fileDate")]  # wall-clock header line
    return open(path).readlines()


def test_cli_devices_flag_reproduces_the_one_device_outputs(tmp_path):
    """--devices a,b,... (muxgl_config.n_devices; here the same GPU named several times) must write the same files,
    byte for byte, as the one-device run: demuxlet's .best, freemuxlet's .lmix / .clust1.samples.gz / .clust1.vcf.gz --
    with the greedy initial clustering made on the group's first device and the EM on the group."""
    p = synth.make_pileup(90, 900, 5, seed=21, mean_entries=160, min_entries=20)
    prefix = str(tmp_path / "plp")
    plpio.write_plp(prefix, p, seed=21)
    vcf = str(tmp_path / "g.vcf.gz")
    plpio.write_vcf(vcf, p, p.truth["G"].astype(np.int64), missing_frac=0.02)
    outs = {}
    for tag, extra in (("one", []), ("grp", ["--devices", "0,0,0"])):
        o = str(tmp_path / f"d_{tag}")
        r = subprocess.run([BIN, "demuxlet", "--plp", prefix, "--vcf", vcf, "--field", "GT", "--out", o] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        f = str(tmp_path / f"f_{tag}")
        r = subprocess.run([BIN, "freemuxlet", "--plp", prefix, "--nsample", "5", "--out", f, "--seed", "1"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[tag] = (o, f)
    assert _read(outs["one"][0] + ".best") == _read(outs["grp"][0] + ".best")
    for ext in (".lmix", ".clust1.samples.gz", ".clust1.vcf.gz"):
        assert _read(outs["one"][1] + ext) == _read(outs["grp"][1] + ext), ext
    r = subprocess.run([BIN, "demuxlet", "--plp", prefix, "--vcf", vcf, "--field", "GT", "--out", str(tmp_path / "x"), "--devices",
                        "0,x"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "--devices" in r.stderr
    r = subprocess.run([BIN, "freemuxlet-old", "--plp", prefix, "--nsample", "3", "--out", str(tmp_path / "y"), "--devices",
                        "0,0"], capture_output=True, text=True)
    assert r.returncode != 0 and "--devices" in r.stderr
