#!/usr/bin/env python
"""End-to-end wall time of the popscle-amd front end on files of the real dsc-pileup format (the
loader bounds the end-to-end time).  Writes a synthetic CEL/VAR/PLP (+VCF) of the given shape to a scratch directory,
runs `popscle-amd demuxlet` and/or `freemuxlet` with POPSCLE_AMD_TIMING=1 and prints the stage times.

    python tools/e2e_cli.py --cells 10000 --snps 50000 --samples 16 [--freemuxlet K] [--dir /tmp/e2e]
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from popscle_amd import plpio, synth  # noqa: E402

BIN = os.path.join(ROOT, "popscle_amd", "bin", "popscle-amd")


def run(cmd):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, POPSCLE_AMD_TIMING="1"))
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        print(r.stderr[-2000:])
        raise SystemExit(f"{cmd[1]} failed")
    for line in r.stderr.splitlines():
        if line.startswith("TIMING"):
            print("   ", line)
    print(f"    TOTAL wall {dt:.3f} s")
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=10000)
    ap.add_argument("--snps", type=int, default=50000)
    ap.add_argument("--samples", type=int, default=16)
    ap.add_argument("--freemuxlet", type=int, default=0, help="also run freemuxlet with this many clusters")
    ap.add_argument("--freemuxlet-old", type=int, default=0, help="also run freemuxlet-old with this many clusters")
    ap.add_argument("--dir", default="/tmp/e2e")
    ap.add_argument("--bgzf", action="store_true", help="store the PLP table as BGZF, as dsc-pileup does")
    ap.add_argument("--cpp-writer", action="store_true",
                    help="write the data set with `popscle-amd synth-plp` (same generator family, BGZF, seconds instead of "
                         "minutes at 10^8 rows)")
    ap.add_argument("--skip-demuxlet", action="store_true", help="freemuxlet only")
    ap.add_argument("--alpha", action="append", default=[], help="demuxlet --alpha values (default: the command's {0, 0.5})")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    prefix = os.path.join(a.dir, "plp")
    t0 = time.perf_counter()
    if a.cpp_writer:
        r = subprocess.run([BIN, "synth-plp", "--cells", str(a.cells), "--snps", str(a.snps), "--samples", str(a.samples),
                            "--seed", "3", "--out", prefix], capture_output=True, text=True,
                           env=dict(os.environ, POPSCLE_AMD_TIMING="1"))
        if r.returncode != 0:
            raise SystemExit(r.stderr[-2000:])
        for line in r.stderr.splitlines():
            print("   ", line)
        vcf = prefix + ".vcf.gz"
        rows = int(r.stderr.split(" PLP rows")[0].split()[-1])
        print(f"wrote {prefix}.* in {time.perf_counter() - t0:.1f} s: plp.gz {os.path.getsize(prefix + '.plp.gz') / 1e6:.1f} MB, "
              f"vcf.gz {os.path.getsize(vcf) / 1e6:.1f} MB")
        a.bgzf = False
    else:
        p = synth.make_pileup(a.cells, a.snps, a.samples, seed=synth.BASE_SEED + 1)
        plpio.write_plp(prefix, p)
        vcf = os.path.join(a.dir, "donors.vcf.gz")
        plpio.write_vcf(vcf, p, p.truth["G"].astype(np.int64))
        rows = int((np.diff(p.entry_rptr) > 0).sum())
        print(f"wrote {prefix}.* in {time.perf_counter() - t0:.1f} s: {a.cells} droplets, {a.snps} SNPs, {rows} PLP rows, "
              f"{p.R} bases, plp.gz {os.path.getsize(prefix + '.plp.gz') / 1e6:.1f} MB")
    if a.bgzf:  # what dsc-pileup itself writes (hts_open "wz"): blocks inflate in parallel
        import gzip
        import shutil

        with gzip.open(prefix + ".plp.gz", "rb") as f, open(prefix + ".plp.txt", "wb") as g:
            shutil.copyfileobj(f, g, 1 << 24)
        subprocess.run([BIN, "bgzf", "--in", prefix + ".plp.txt", "--out", prefix + ".plp.gz"], check=True)
        os.remove(prefix + ".plp.txt")
        print(f"re-compressed as BGZF: plp.gz {os.path.getsize(prefix + '.plp.gz') / 1e6:.1f} MB")
    if not a.skip_demuxlet:
        print("demuxlet --field GT:")
        alphas = [x for v in a.alpha for x in ("--alpha", v)]
        dt = run([BIN, "demuxlet", "--plp", prefix, "--vcf", vcf, "--field", "GT", "--out", os.path.join(a.dir, "dmx")] + alphas)
        print(f"    => {rows / dt / 1e6:.2f} M PLP rows/s end to end")
    if a.freemuxlet:
        print(f"freemuxlet --nsample {a.freemuxlet}:")
        dt = run([BIN, "freemuxlet", "--plp", prefix, "--nsample", str(a.freemuxlet), "--out", os.path.join(a.dir, "fmx")])
        print(f"    => {rows / dt / 1e6:.2f} M PLP rows/s end to end")
    if a.freemuxlet_old:
        print(f"freemuxlet-old --nsample {a.freemuxlet_old}:")
        run([BIN, "freemuxlet-old", "--plp", prefix, "--nsample", str(a.freemuxlet_old), "--out",
             os.path.join(a.dir, "fmxold")])


if __name__ == "__main__":
    main()
