"""CPU tests of the host side (popscle_amd/host): the CEL/VAR/PLP loader and the VCF -> GP reader of the popscle-amd
front end against an independent Python restatement of load_from_plp / parse_posteriors (tests/pyplp.py), on files of
the real dsc-pileup format.  No GPU: `popscle-amd dump-plp` runs the loader only."""
import os
import subprocess

import numpy as np
import pytest

import pyplp
from popscle_amd import plpio, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "popscle_amd", "bin", "popscle-amd")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(BIN):
        from popscle_amd.build import build_lib

        build_lib()
        subprocess.run(["make", "-C", os.path.join(ROOT, "popscle_amd", "host")], check=True)
    return BIN


def dump(exe, prefix, out, *extra):
    r = subprocess.run([exe, "dump-plp", "--plp", prefix, "--out", out, *extra], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return plpio.read_dump(out)


def same(got, want):
    for k in ("C", "S", "nv", "bcs"):
        assert got[k] == want[k], k
    for k in ("cell_ptr", "entry_snp", "entry_rptr", "reads", "af", "has_gp", "cell_totl_reads", "cell_uniq_reads"):
        assert np.array_equal(got[k], want[k]), k
    assert got["gp"].shape == want["gp"].shape
    assert np.array_equal(got["gp"], want["gp"]), "GP tensor differs (bit-exact expected)"


def make_files(tmp_path, C=12, S=60, V=3, seed=3, deep=True):
    p = synth.make_pileup(C, S, V, seed=seed, mean_entries=25, min_entries=5, reads_lambda=1.5 if deep else 0.3,
                          other=0.05)
    rng = np.random.default_rng(seed)
    raw_bq = rng.integers(2, 41, size=p.R).astype(np.uint8)  # below min-BQ, between, above cap-BQ
    prefix = str(tmp_path / "plp")
    plpio.write_plp(prefix, p, raw_bq=raw_bq, seed=seed, extra_cells=2)
    return p, prefix


def test_loader_without_vcf(exe, tmp_path):
    p, prefix = make_files(tmp_path)
    got = dump(exe, prefix, str(tmp_path / "d.bin"))
    want = pyplp.load(prefix)
    same(got, want)
    assert got["C"] == p.C + 2 and got["nnz"] <= p.nnz
    # the UMI order is the order of the hex STRINGS of a global counter: with > 16 kept bases some entry must be
    # affected ("10" < "9"); make sure the case is exercised
    assert got["R"] > 16


@pytest.mark.parametrize("minbq,capbq", [(13, 20), (2, 40), (30, 35)])
def test_loader_bq_filters(exe, tmp_path, minbq, capbq):
    p, prefix = make_files(tmp_path, seed=5)
    got = dump(exe, prefix, str(tmp_path / "d.bin"), "--min-BQ", str(minbq), "--cap-BQ", str(capbq))
    same(got, pyplp.load(prefix, min_bq=minbq, cap_bq=capbq))
    q = got["reads"][got["reads"] != 0xFF] & 0x7F
    assert q.min() >= minbq and q.max() <= capbq


def test_loader_cell_filters_and_group_list(exe, tmp_path):
    p, prefix = make_files(tmp_path, seed=7)
    got = dump(exe, prefix, str(tmp_path / "d.bin"), "--min-snp", "1")
    same(got, pyplp.load(prefix, min_snp=1))
    assert got["C"] == p.C  # the two extra droplets have NUM.SNP = 0
    bcs = pyplp.load(prefix)["bcs"]
    keep = bcs[::2]
    gl = tmp_path / "groups.txt"
    gl.write_text("\n".join(keep) + "\n")
    got = dump(exe, prefix, str(tmp_path / "d2.bin"), "--group-list", str(gl))
    same(got, pyplp.load(prefix, group_list=keep))
    assert got["bcs"] == keep


@pytest.mark.parametrize("field", ["GT", "GP", "PL"])
def test_loader_with_vcf(exe, tmp_path, field):
    p, prefix = make_files(tmp_path, C=10, S=80, V=4, seed=11, deep=False)
    G = p.truth["G"].astype(np.int64)
    rng = np.random.default_rng(1)
    gp = rng.dirichlet([0.3, 0.3, 0.3], size=G.shape)
    pl = rng.integers(0, 60, size=G.shape + (3,))
    vcf = str(tmp_path / "g.vcf.gz")
    plpio.write_vcf(vcf, p, G, field=field, missing_frac=0.1, drop_snps=[3, 17, 18], gp=gp, pl=pl)
    got = dump(exe, prefix, str(tmp_path / "d.bin"), "--vcf", vcf, "--field", field)
    want = pyplp.load(prefix, vcf=vcf, field=field)
    same(got, want)
    assert got["nv"] == 4 and got["sample_ids"] == ["S0", "S1", "S2", "S3"]
    assert got["has_gp"][[3, 17, 18]].tolist() == [0, 0, 0]
    assert 0 < got["has_gp"].sum() < p.S  # MAC / call-rate filters remove some monomorphic sites too
    rows = got["gp"][got["has_gp"] == 1]
    assert np.allclose(rows.sum(axis=2), 1.0, atol=1e-6)


def test_loader_geno_error_flags(exe, tmp_path):
    p, prefix = make_files(tmp_path, C=8, S=50, V=3, seed=13, deep=False)
    vcf = str(tmp_path / "g.vcf")
    plpio.write_vcf(vcf, p, p.truth["G"].astype(np.int64))
    got = dump(exe, prefix, str(tmp_path / "d.bin"), "--vcf", vcf, "--field", "GT", "--geno-error-offset", "0.05",
               "--geno-error-coeff", "0.5", "--min-mac", "2", "--min-callrate", "0.9")
    want = pyplp.load(prefix, vcf=vcf, field="GT", geno_error_offset=0.05, geno_error_coeff=0.5, min_mac=2,
                      min_callrate=0.9)
    same(got, want)


def test_loader_rejects_malformed_header(exe, tmp_path):
    import gzip

    p, prefix = make_files(tmp_path, seed=17)
    with gzip.open(prefix + ".cel.gz", "rt") as f:
        lines = f.read().split("\n")
    lines[0] = "#DROPLET_ID\tBARCODE\tNUM.READ\tNUM.UMI\tNUM.SNP"  # the outdated 5-column header
    with gzip.open(prefix + ".cel.gz", "wt") as f:
        f.write("\n".join(lines))
    r = subprocess.run([exe, "dump-plp", "--plp", prefix, "--out", str(tmp_path / "x.bin")], capture_output=True,
                       text=True)
    assert r.returncode != 0 and "malformed or outdated" in r.stderr


# ---- the threaded .plp.gz reader (popscle_amd/host/plp_fast.hpp) -----------------------------------------------------

def rewrite_plp(prefix, fn):
    """apply fn(list of data rows) -> list of lines to the rows of prefix.plp.gz (header kept)"""
    import gzip

    with gzip.open(prefix + ".plp.gz", "rt") as f:
        lines = f.read().split("\n")
    hdr, rows = lines[0], [x for x in lines[1:] if x]
    with gzip.open(prefix + ".plp.gz", "wt") as f:
        f.write("\n".join([hdr] + fn(rows)) + "\n")


@pytest.mark.parametrize("order", ["cell_major", "shuffled", "reversed"])
def test_loader_row_order_is_free(exe, tmp_path, order):
    """dsc-pileup writes the table SNP-major; load_from_plp keys everything by (cell, SNP) maps, so any row order must
    give the same pileup except for the UMI counter (= row order) that orders the reads INSIDE an entry -- pyplp
    restates that rule, so it is the judge for every order"""
    p, prefix = make_files(tmp_path, C=15, S=80, seed=23)
    rng = np.random.default_rng(5)

    def fn(rows):
        if order == "cell_major":
            return sorted(rows, key=lambda r: (int(r.split("\t")[0]), int(r.split("\t")[1])))
        if order == "reversed":
            return rows[::-1]
        return [rows[i] for i in rng.permutation(len(rows))]

    rewrite_plp(prefix, fn)
    same(dump(exe, prefix, str(tmp_path / "d.bin")), pyplp.load(prefix))


def test_loader_duplicate_rows_of_one_entry(exe, tmp_path):
    """two rows for the same (droplet, SNP) far apart in the file: the reference merges them into one map node"""
    p, prefix = make_files(tmp_path, C=10, S=40, seed=29)
    rewrite_plp(prefix, lambda rows: rows + rows[:7])
    same(dump(exe, prefix, str(tmp_path / "d.bin")), pyplp.load(prefix))


def test_loader_blank_line_ends_the_file(exe, tmp_path):
    """`while( tsv.read_line() > 0 )` (sc_drop_seq.cpp:346) stops at the first line without fields"""
    p, prefix = make_files(tmp_path, C=10, S=40, seed=31)
    rewrite_plp(prefix, lambda rows: rows[:len(rows) // 2] + ["  \t "] + rows[len(rows) // 2:])
    got = dump(exe, prefix, str(tmp_path / "d.bin"))
    want = pyplp.load(prefix)
    same(got, want)
    assert got["R"] < p.R


def test_loader_whitespace_and_crlf(exe, tmp_path):
    """fields are split on whitespace runs (ksplit, delimiter 0): spaces, repeated tabs, CR before LF, extra columns"""
    p, prefix = make_files(tmp_path, C=10, S=40, seed=37)
    want = pyplp.load(prefix)

    def fn(rows):
        out = []
        for i, r in enumerate(rows):
            a = r.split("\t")
            out.append([" ".join(a), "\t\t".join(a) + "\r", "  " + "\t".join(a) + "\textra"][i % 3])
        return out

    rewrite_plp(prefix, fn)
    same(dump(exe, prefix, str(tmp_path / "d.bin")), want)


@pytest.mark.parametrize("bad,msg", [
    ("3\t5\t01", "has 3 fields"),
    ("999\t5\t0\tI", "DROPLET_ID 999 out of range"),
    ("3\t99999\t0\tI", "SNP_ID 99999 out of range"),
    ("3\t5\t011\tII", "Length are different"),
])
def test_loader_bad_rows_are_fatal(exe, tmp_path, bad, msg):
    p, prefix = make_files(tmp_path, C=10, S=40, seed=41)
    rewrite_plp(prefix, lambda rows: rows[:11] + [bad] + rows[11:])
    r = subprocess.run([exe, "dump-plp", "--plp", prefix, "--out", str(tmp_path / "x.bin")], capture_output=True,
                       text=True)
    assert r.returncode != 0 and msg in r.stderr, r.stderr
    if "fields" in msg:
        assert "line 13 " in r.stderr  # header + 11 rows + the bad one


def test_loader_threads_and_blocks(exe, tmp_path):
    """a table of several inflate blocks (8 MiB each) parsed with 1, 3 and 8 threads gives the same bytes"""
    p = synth.make_pileup(700, 20000, 2, seed=43, mean_entries=900, min_entries=100, reads_lambda=0.6, other=0.02)
    prefix = str(tmp_path / "big")
    plpio.write_plp(prefix, p, seed=1)
    outs = []
    for nt in (1, 3, 8):
        out = str(tmp_path / f"d{nt}.bin")
        r = subprocess.run([exe, "dump-plp", "--plp", prefix, "--out", out], capture_output=True, text=True,
                           env=dict(os.environ, POPSCLE_AMD_THREADS=str(nt)))
        assert r.returncode == 0, r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2]
    got = plpio.read_dump(str(tmp_path / "d8.bin"))
    assert np.array_equal(got["cell_ptr"], p.cell_ptr[: p.C + 1]) and np.array_equal(got["entry_snp"], p.entry_snp)
    # reads: same multiset per entry (the order inside an entry follows the "%x" rule, pinned by the small tests)
    assert got["R"] == p.R and np.array_equal(got["entry_rptr"], p.entry_rptr)


# ---- the BGZF writer behind every "wz" output (.clust1.vcf.gz, .clust1.samples.gz) ------------------------------------

@pytest.mark.parametrize("size", [0, 1, 65279, 65280, 65281, 9_000_000])
def test_bgzf_writer(exe, tmp_path, size):
    """hts_open(..., "wz") writes BGZF: independent gzip members with the BC extra field and an empty end marker.
    The payload must survive, every member must be a valid BGZF block of <= 64 KiB, and the file must end with the
    28-byte EOF block (SAM spec 4.1.2)."""
    import gzip
    import struct

    rng = np.random.default_rng(size)
    words = np.array([b"CLUST", b"0/1", b"\t", b"255,0,12", b"\n", b"0.99999", b":"], dtype=object)
    data = b"".join(words[rng.integers(0, len(words), size=size // 2 + 1)])[:size]
    assert len(data) == size
    src, dst = tmp_path / "in.txt", tmp_path / "out.gz"
    src.write_bytes(data)
    r = subprocess.run([exe, "bgzf", "--in", str(src), "--out", str(dst)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = dst.read_bytes()
    assert gzip.decompress(raw) == data
    eof = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    assert raw.endswith(eof)
    off, total, nblocks = 0, 0, 0
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04" and raw[off + 12:off + 16] == b"BC\x02\x00"
        bsize = struct.unpack("<H", raw[off + 16:off + 18])[0] + 1
        isize = struct.unpack("<I", raw[off + bsize - 4:off + bsize])[0]
        assert isize <= 0xff00
        total += isize
        off += bsize
        nblocks += 1
    assert off == len(raw) and total == size and nblocks == (size + 0xff00 - 1) // 0xff00 + 1


def test_loader_bgzf_input(exe, tmp_path):
    """dsc-pileup writes its tables as BGZF (hts_open "wz"); the loader inflates the blocks in parallel.  Same table as
    plain gzip and as BGZF (re-compressed with the front end's own BGZF writer) must load to the same bytes."""
    import gzip
    import shutil

    p = synth.make_pileup(500, 20000, 2, seed=47, mean_entries=900, min_entries=100, reads_lambda=0.6, other=0.02)
    prefix = str(tmp_path / "gz")
    plpio.write_plp(prefix, p, seed=1)
    ref = str(tmp_path / "ref.bin")
    assert subprocess.run([exe, "dump-plp", "--plp", prefix, "--out", ref]).returncode == 0
    bprefix = str(tmp_path / "bgzf")
    for ext in (".cel.gz", ".var.gz"):
        shutil.copy(prefix + ext, bprefix + ext)
    txt = tmp_path / "plp.txt"
    txt.write_bytes(gzip.open(prefix + ".plp.gz", "rb").read())
    assert txt.stat().st_size > 3 * (1 << 20)  # dozens of 64 KiB blocks
    assert subprocess.run([exe, "bgzf", "--in", str(txt), "--out", bprefix + ".plp.gz"]).returncode == 0
    for nt in (1, 5):
        out = str(tmp_path / f"b{nt}.bin")
        r = subprocess.run([exe, "dump-plp", "--plp", bprefix, "--out", out], capture_output=True, text=True,
                           env=dict(os.environ, POPSCLE_AMD_THREADS=str(nt), POPSCLE_AMD_TIMING="1"))
        assert r.returncode == 0, r.stderr
        assert "bgzf 1" in r.stderr
        assert open(out, "rb").read() == open(ref, "rb").read()
    # a corrupted block is detected (CRC), not parsed
    raw = bytearray(open(bprefix + ".plp.gz", "rb").read())
    raw[len(raw) // 2] ^= 0x55
    open(bprefix + ".plp.gz", "wb").write(bytes(raw))
    r = subprocess.run([exe, "dump-plp", "--plp", bprefix, "--out", str(tmp_path / "x.bin")], capture_output=True,
                       text=True)
    assert r.returncode != 0


def _write_tiny(tmp_path, header_ids, rows, fmt):
    """PLP files with three SNPs at 1:1000/1010/1020 and a hand-written VCF with the given sample columns"""
    p = synth.make_pileup(4, 3, 3, seed=1, mean_entries=3, min_entries=3)
    prefix = str(tmp_path / "plp")
    plpio.write_plp(prefix, p, seed=1)
    vcf = str(tmp_path / "h.vcf")
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n##contig=<ID=1>\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(header_ids) + "\n")
        for i, cols in enumerate(rows):
            f.write(f"1\t{1000 + 10 * i}\t.\tA\tG\t.\tPASS\t.\t{fmt}\t" + "\t".join(cols) + "\n")
    return prefix, vcf


def _mix(gp32, err=0.1):
    """sc_drop_seq.cpp:287-315 on float GPs: promote, avgGP with the 1e-10 pseudo-count, (1-err) gp + err avg"""
    gp = gp32.astype(np.float64)
    avg = np.full(3, 1e-10)
    for v in range(gp.shape[0]):
        avg += gp[v]
    avg /= (avg[0] + avg[1]) + avg[2]
    return (1 - err) * gp + err * avg


def test_vcf_rules_against_hand_derived_rows(exe, tmp_path):
    """A third check of the VCF -> GP rules, independent of tests/pyplp.py: rows derived by hand from
    bcf_filtered_reader.cpp:385-408 (GT: one-hot, missing genotype -> HWE from AC/AN with pseudo-counts, all through
    float) and :412-455 (GP: per-sample normalisation in float; gt_error = 0 as load_from_plp passes it,
    sc_drop_seq.cpp:285), followed by the mixing of sc_drop_seq.cpp:287-315."""
    f32 = np.float32
    # GT: S_a 0/0, S_b 0/1, S_c missing -> acs = (3, 1), an = 4, nalleles = 2:
    #   (0,0): 1 * (3+.5)/5 * (3+.5)/5   (1,0): 2 * (1+.5)/5 * (3+.5)/5   (1,1): 1 * (1+.5)/5 * (1+.5)/5
    prefix, vcf = _write_tiny(tmp_path, ["A", "B", "C"], [["0/0", "0/1", "./."], ["1/1", "1|0", "0/0"], ["0/1", "0/1", "1/1"]],
                              "GT")
    got = dump(exe, prefix, str(tmp_path / "d.bin"), "--vcf", vcf, "--field", "GT")
    miss = np.array([f32(1.0 * (3 + 0.5) / 5.0 * (3 + 0.5) / 5.0), f32(2.0 * (1 + 0.5) / 5.0 * (3 + 0.5) / 5.0),
                     f32(1.0 * (1 + 0.5) / 5.0 * (1 + 0.5) / 5.0)], dtype=f32)
    want0 = _mix(np.array([[1, 0, 0], [0, 1, 0], miss], dtype=f32))
    want1 = _mix(np.array([[0, 0, 1], [0, 1, 0], [1, 0, 0]], dtype=f32))
    assert got["sample_ids"] == ["A", "B", "C"] and got["has_gp"].tolist() == [1, 1, 1]
    assert np.array_equal(got["gp"][0], want0) and np.array_equal(got["gp"][1], want1)
    # GP: float arithmetic, each sample divided by its own float sum
    prefix, vcf = _write_tiny(tmp_path, ["A", "B", "C"],
                              [["0/0:0.8,0.15,0.05", "0/1:0.1,0.6,0.3", "0/1:0.2,0.2,0.2"],
                               ["0/1:0.5,0.5,0", "1/1:0,0.25,0.75", "0/0:1,0,0"],
                               ["0/1:0.3,0.4,0.3", "0/1:0.3,0.4,0.3", "1/1:0.1,0.1,0.8"]], "GT:GP")
    got = dump(exe, prefix, str(tmp_path / "d.bin"), "--vcf", vcf, "--field", "GP")

    def norm(a, b, c):
        a, b, c = f32(a), f32(b), f32(c)
        s = f32(f32(f32(0) + a) + b) + c  # sumgp = 0; sumgp += gps[j]
        return [a / s, b / s, c / s]

    want0 = _mix(np.array([norm(0.8, 0.15, 0.05), norm(0.1, 0.6, 0.3), norm(0.2, 0.2, 0.2)], dtype=f32))
    assert np.array_equal(got["gp"][0], want0)
    assert np.allclose(got["gp"][0][2], 0.9 / 3 + 0.1 * want0.mean(axis=0), atol=0.05)  # sanity of the hand row itself


def test_vcf_sample_subset_is_numbered_in_sorted_id_order(exe, tmp_path):
    """--sm / --sm-list: bcf_filtered_reader.cpp:105-122 walks the requested IDs as a std::set, so the samples are
    numbered in SORTED ID order whatever the order of the VCF's columns or of the flags; an unknown ID is an error."""
    rows = [["0/0", "0/1", "1/1", "0/1"], ["1/1", "0/0", "0/1", "0/0"], ["0/1", "1/1", "0/0", "1/1"]]
    prefix, vcf = _write_tiny(tmp_path, ["zeta", "alpha", "mid", "beta"], rows, "GT")
    got = dump(exe, prefix, str(tmp_path / "d.bin"), "--vcf", vcf, "--field", "GT", "--sm", "zeta", "--sm", "beta",
               "--min-mac", "0", "--min-callrate", "0")
    assert got["sample_ids"] == ["beta", "zeta"] and got["nv"] == 2
    onehot = {"0/0": [1, 0, 0], "0/1": [0, 1, 0], "1/1": [0, 0, 1]}
    for s in range(3):
        want = _mix(np.array([onehot[rows[s][3]], onehot[rows[s][0]]], dtype=np.float32))  # beta = column 3, zeta = column 0
        assert np.array_equal(got["gp"][s], want), s
    lst = tmp_path / "ids.txt"
    lst.write_text("mid\nalpha\n")
    got = dump(exe, prefix, str(tmp_path / "d.bin"), "--vcf", vcf, "--field", "GT", "--sm-list", str(lst), "--min-mac", "0",
               "--min-callrate", "0")
    assert got["sample_ids"] == ["alpha", "mid"]
    r = subprocess.run([exe, "dump-plp", "--plp", prefix, "--out", str(tmp_path / "x.bin"), "--vcf", vcf, "--field", "GT",
                        "--sm", "nobody"], capture_output=True, text=True)
    assert r.returncode != 0 and "Cannot find sample ID nobody" in r.stderr


def test_plp_rules_against_hand_derived_rows(exe, tmp_path):
    """The CEL / PLP rules of load_from_plp derived by hand from sc_drop_seq.cpp:335-380, independent of tests/pyplp.py:
    a base is kept if bq >= minBQ and then capped (:361-363); every kept base is its own UMI named sprintf("%x", numi++)
    with ONE counter over the whole file (:364), and an entry is a std::map keyed by that string, so its reads come out in
    the LEXICOGRAPHIC order of the hex names ("0" < "1" < "10" < "11" < "2" ... "9" < "a" ... "f"); dropped bases do not
    consume a name; cell_uniq_reads counts kept bases, cell_totl_reads too -- unless the CEL row's NUM.UMIwSNP and NUM.SNP
    agree with what was loaded, in which case it is overwritten by the CEL row's NUM.READ (:375-380)."""
    import gzip

    prefix = str(tmp_path / "h")
    with gzip.open(prefix + ".cel.gz", "wt") as f:
        f.write("#DROPLET_ID\tBARCODE\tNUM.READ\tNUM.UMI\tNUM.UMIwSNP\tNUM.SNP\n")
        f.write("0\tAAA-1\t50\t40\t18\t2\n")  # 17 + 1 kept bases at 2 markers: consistent -> totl := 50
        f.write("1\tCCC-1\t30\t20\t5\t1\n")   # 3 kept bases, the row says 5 -> totl stays the kept count
        f.write("2\tGGG-1\t10\t8\t2\t1\n")    # consistent -> totl := 10
    with gzip.open(prefix + ".var.gz", "wt") as f:
        f.write("#SNP_ID\tCHROM\tPOS\tREF\tALT\tAF\n0\t1\t1000\tA\tG\t0.25000\n1\t1\t1010\tA\tG\t0.50000\n")
    # droplet 0 / marker 0: 18 bases, the fourth below --min-BQ 13 (dropped, takes no name), so 17 kept bases named
    # 0..9, a..f, 10; quality of kept base n: 14 + n (--cap-BQ 60 keeps them distinct); alleles 0/1 alternating, one '2'
    al0, bq0, kept = "", "", 0
    for i in range(18):
        if i == 3:
            al0 += "1"
            bq0 += chr(33 + 5)
            continue
        al0 += "2" if kept == 6 else str(kept & 1)
        bq0 += chr(33 + 14 + kept)
        kept += 1
    assert kept == 17
    with gzip.open(prefix + ".plp.gz", "wt") as f:
        f.write("#DROPLET_ID\tSNP_ID\tALLELES\tBASEQS\n")
        f.write(f"0\t0\t{al0}\t{bq0}\n")                            # names 0 .. 10 (hex)
        f.write("1\t0\t101\t" + chr(33 + 20) + chr(33 + 70) + chr(33 + 13) + "\n")  # names 11, 12, 13; 70 capped to 60
        f.write("0\t1\t1\t" + chr(33 + 33) + "\n")                  # name 14
        f.write("2\t1\t00\t" + chr(33 + 13) + chr(33 + 12) + "\n")  # name 15; the second base is below min-BQ
    got = dump(exe, prefix, str(tmp_path / "d.bin"), "--min-BQ", "13", "--cap-BQ", "60")
    assert got["C"] == 3 and got["S"] == 2 and got["bcs"] == ["AAA-1", "CCC-1", "GGG-1"]
    assert got["cell_ptr"].tolist() == [0, 2, 3, 4] and got["entry_snp"].tolist() == [0, 1, 0, 1]
    assert got["entry_rptr"].tolist() == [0, 17, 18, 21, 22]

    def code(n):  # kept base n of droplet 0 / marker 0
        return 0xFF if n == 6 else ((n & 1) << 7) | (14 + n)

    lex = sorted(range(17), key=lambda n: "%x" % n)  # 0, 1, 16, 2, 3, ..., 9, 10 (a), ..., 15 (f)
    assert lex[:4] == [0, 1, 16, 2] and lex[-1] == 15
    want = [code(n) for n in lex] + [(1 << 7) | 33] + [(1 << 7) | 20, 60, (1 << 7) | 13] + [13]
    assert got["reads"].tolist() == want
    assert got["cell_uniq_reads"].tolist() == [18, 3, 1]
    # GGG-1: the CEL row promises 2 kept bases, one was loaded -> no overwrite either
    assert got["cell_totl_reads"].tolist() == [50, 3, 1]


@pytest.mark.parametrize("world", [2, 3, 5])
def test_loader_cuts_a_ranks_slabs_at_file_level(exe, tmp_path, world):
    """`dump-plp --rank r --world N`: the loader keeps only the rank's row slab (its cells) and column slab (its markers)
    while it parses -- and they are exactly the cuts shard.take_cells / take_snps make of the whole pileup, read order
    inside the entries included (the name of a kept base counts the bases of ALL rows, kept or not)."""
    from popscle_amd import shard

    p, prefix = make_files(tmp_path, C=23, S=70, deep=True)
    whole = dump(exe, prefix, str(tmp_path / "whole.bin"), "--min-BQ", "5", "--cap-BQ", "33")
    C, S = whole["C"], whole["S"]
    q = synth.Pileup(C, S, whole["cell_ptr"], whole["entry_snp"], whole["entry_rptr"], whole["reads"], whole["af"])
    (c_ranges, _), (s_ranges, _) = shard.equal_ranges(C, world), shard.equal_ranges(S, world)
    for r in range(world):
        out = str(tmp_path / f"slab{r}.bin")
        sub = subprocess.run([exe, "dump-plp", "--plp", prefix, "--out", out, "--rank", str(r), "--world", str(world),
                              "--min-BQ", "5", "--cap-BQ", "33"], capture_output=True, text=True)
        assert sub.returncode == 0, sub.stderr
        d = plpio.read_slab_dump(out)
        assert (d["c0"], d["c1"]) == c_ranges[r] and (d["s0"], d["s1"]) == s_ranges[r] and d["C"] == C and d["S"] == S
        rows = shard.take_cells(q, *c_ranges[r])
        for a, b in zip(d["rows"], (rows.cell_ptr, rows.entry_snp, rows.entry_rptr, rows.reads)):
            assert np.array_equal(a, b)
        for a, b in zip(d["cols"], shard.take_snps(q, *s_ranges[r])):
            assert np.array_equal(a, b)
        c0, c1 = c_ranges[r]
        assert np.array_equal(d["cell_uniq_reads"][c0:c1], whole["cell_uniq_reads"][c0:c1])
        assert np.array_equal(d["cell_totl_reads"][c0:c1], whole["cell_totl_reads"][c0:c1])
        assert np.array_equal(d["af"], whole["af"])


def test_errors_next_to_the_vcf_thread_exit_cleanly(exe, tmp_path):
    """The VCF merge-join runs on a thread of its own (plp.hpp); fatal() throws.  A user error inside that thread, or on
    the main thread while that thread is alive, must end in `FATAL ERROR` + exit status 1 through main()'s handler,
    not in std::terminate (SIGABRT)."""
    p, prefix = make_files(tmp_path, C=10, S=80, V=4, seed=11, deep=False)
    vcf = str(tmp_path / "g.vcf.gz")
    plpio.write_vcf(vcf, p, p.truth["G"].astype(np.int64), field="GT")

    def run(*extra):
        return subprocess.run([exe, "dump-plp", "--plp", prefix, "--out", str(tmp_path / "x.bin"), "--vcf", vcf, *extra],
                              capture_output=True, text=True)

    r = run("--field", "GP")  # the file carries GT only: raised inside the thread
    assert r.returncode == 1 and "Cannot parse posterior probability" in r.stderr, (r.returncode, r.stderr)
    r = run("--field", "GT", "--geno-error-coeff", "0.5", "--r2-info", "NOPE")  # missing INFO field: inside the thread
    assert r.returncode == 1 and "Cannot extract NOPE" in r.stderr, (r.returncode, r.stderr)
    r = run("--field", "GT", "--rank", "5", "--world", "2")  # checked before the thread starts
    assert r.returncode == 1 and "--rank must be in" in r.stderr, (r.returncode, r.stderr)
    rewrite_plp(prefix, lambda rows: rows[:11] + ["3\t5\t01"] + rows[11:])  # main-thread error, thread alive
    r = run("--field", "GT")
    assert r.returncode == 1 and "has 3 fields" in r.stderr, (r.returncode, r.stderr)
    r = run("--field", "GP")  # both stages fail: the VCF's error is the one the reference would meet first
    assert r.returncode == 1 and "Cannot parse posterior probability" in r.stderr, (r.returncode, r.stderr)


def test_writer_number_formatting_is_printf(exe):
    """The cluster VCF's sample fields are formatted without printf (util.hpp: fmt_int, fmt_g3): the binary's own
    comparison against "%d" / "%.3lg" over random values, exact ties of the third digit and their neighbours."""
    r = subprocess.run([exe, "selftest-fmt", "200000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 differences" in r.stdout
