#!/usr/bin/env python
"""Build profiles/traffic.json from the PMC passes of tools/profile_bench.sh: per BASELINE config, the counters of the
dominant kernel family PER STEP (a step = one demuxlet pass / one EM iteration), stamped with the fingerprint of the
kernel sources they were measured on.  bench.py quotes them only while that fingerprint matches.

    python tools/traffic_update.py --config 1 --prefix gpurun_out/prof_c1 [--config 2 --prefix gpurun_out/prof_c2 ...]

For config N it reads <prefix>_fetch, _write, _f64, _mix (rocprofv3 --pmc output directories, each its own run) and
<prefix>_kt.log (the bench line of the kernel-trace run, for the entries per step)."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from prof_summary import short  # noqa: E402

from popscle_amd.build import source_hash  # noqa: E402

FAMILY = {  # config -> (kernels of the dominant phase, a kernel that runs exactly once per step)
    1: (r"^demux_oct_kernel", r"^demux_oct_finish_kernel"),
    2: (r"^(demux_(wave|ring_lin|entry_pg)|wave_(neutral_pg|gm|combine|lpg)_kernel|ring_lut_kernel)",  # MUXGL_T_DEMUX_SWEEP
        r"^demux_call_wave_kernel"),
    3: (r"^fmx_estep_oct_kernel", r"^fmx_call_kernel"),
    4: (r"^fmx_estep_wave_kernel", r"^fmx_call_kernel"),
}


def sums(d):
    """{kernel: {counter: total over dispatches}}, {kernel: dispatches}"""
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    if not f:
        return tot, {}
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


def per_step(d, fam, anchor, counters):
    tot, n = sums(d)
    steps = sum(c for k, c in n.items() if re.match(anchor, k))
    if not steps:
        return None
    out = {}
    for c in counters:
        out[c] = sum(v.get(c, 0.0) for k, v in tot.items() if re.match(fam, k)) / steps
    out["launches_per_step"] = sum(c for k, c in n.items() if re.match(fam, k)) / steps
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, action="append", required=True)
    ap.add_argument("--prefix", action="append", required=True)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--note", default="tools/profile_bench.sh")
    a = ap.parse_args()
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = None
    try:
        doc = json.load(open(a.out))  # configs that are not named keep their records (and their own fingerprints)
    except Exception:
        doc = {}
    doc.pop("source_hash", None)
    doc.pop("git_head", None)
    doc.update({"measured_with": a.note,
           "what": "counters of the dominant kernel family per step (rocprofv3 --pmc, one counter set per run); "
                   "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (FETCH x2 on gfx950, "
                   "MI355X_MICROARCH.md); fma/mul/add_f64 and valu are wave-level instruction counts; every config record "
                   "carries the fingerprint of the sources of its kernel family (popscle_amd.build.FAMILY_SOURCES)"})
    for cfg, pre in zip(a.config, a.prefix):
        fam, anchor = FAMILY[cfg]
        rec = {"kernels": fam, "source_hash": source_hash(cfg), "git_head": head}
        fe = per_step(pre + "_fetch", fam, anchor, ["FETCH_SIZE"])
        wr = per_step(pre + "_write", fam, anchor, ["WRITE_SIZE"])
        f64 = per_step(pre + "_f64", fam, anchor, ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"])
        mix = per_step(pre + "_mix", fam, anchor, ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"])
        if fe and wr:
            rec.update(fetch_kb=fe["FETCH_SIZE"], write_kb=wr["WRITE_SIZE"],
                       hbm_bytes_per_launch=(2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0,
                       launches_per_step=fe["launches_per_step"])
        if f64:
            rec.update(fma_f64=f64["SQ_INSTS_VALU_FMA_F64"], mul_f64=f64["SQ_INSTS_VALU_MUL_F64"],
                       add_f64=f64["SQ_INSTS_VALU_ADD_F64"])
        if mix:
            rec.update(valu=mix["SQ_INSTS_VALU"], lds=mix["SQ_INSTS_LDS"], salu=mix["SQ_INSTS_SALU"])
        # entries per step of the profiled run, from its bench line
        units = None
        for log in glob.glob(pre + "_*.log"):
            for ln in open(log, errors="replace"):
                if ln.startswith("{"):
                    try:
                        d = json.loads(ln)
                    except Exception:
                        continue
                    c = d.get("config", {})
                    units = c.get("entries_per_gpu") or c.get("entries")
        rec["units"] = units
        rec["units_what"] = "pileup entries per step of the profiled run (counters scale linearly in them)"
        doc[f"config{cfg}"] = rec
    json.dump(doc, open(a.out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
