#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel-trace --stats, and separate --pmc passes) into one small JSON/markdown summary
that can be committed under profiles/.

    python tools/prof_summary.py --kt gpurun_out/prof_kt --fetch gpurun_out/prof_fetch --write gpurun_out/prof_write \
        [--sq gpurun_out/prof_sq] --out profiles/r01_demux_config1

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: bytes = FETCH_SIZE*1024 (x2 on gfx950: the
counter tallies 128-B requests at 64 B) + WRITE_SIZE*1024, each collected in its own --pmc pass.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def load_stats(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f[0])):
        out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                 "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                 "pct": float(r["Percentage"])}
    return out


def load_counters(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    if not f:
        return {}, {}
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                   "lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"]), "wg": int(r["Workgroup_Size"]),
                   "grid": int(r["Grid_Size"])}
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kt")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--smi", default="", help="rocm-smi --json samples taken during the trace pass (profile_bench.sh)")
    a = ap.parse_args()
    stats = load_stats(a.kt) if a.kt else {}
    summary = {"note": a.note, "kernels": {}}
    clock_note = ""
    if a.smi and os.path.exists(a.smi):
        sclk, pw = [], []
        for ln in open(a.smi):
            try:
                d = json.loads(ln)["card0"]
            except Exception:
                continue
            for k, v in d.items():
                m = re.search(r"\((\d+)Mhz\)", str(v))
                if "sclk" in k and m:
                    sclk.append(int(m.group(1)))
                if "Power" in k:
                    try:
                        pw.append(float(str(v).split()[0]))
                    except Exception:
                        pass
        busy = sorted(x for x in sclk if x > 500)
        if busy:
            summary["box_clock"] = {"sclk_mhz_median_busy": busy[len(busy) // 2], "sclk_mhz_max": busy[-1], "samples": len(sclk),
                                    "power_w_max": max(pw) if pw else None}
            clock_note = (f"Box clock during the trace pass (rocm-smi, {len(sclk)} samples every 0.25 s): shader clock median of the "
                          f"busy samples {busy[len(busy) // 2]} MHz, max {busy[-1]} MHz" +
                          (f"; package power max {max(pw):.0f} W" if pw else "") + ".\n\n")
    fetch, meta = load_counters(a.fetch) if a.fetch else ({}, {})
    write, _ = load_counters(a.write) if a.write else ({}, {})
    sq, meta2 = load_counters(a.sq) if a.sq else ({}, {})
    meta.update(meta2)
    for k in sorted(set(stats) | set(fetch) | set(write) | set(sq)):
        e = dict(stats.get(k, {}))
        e.update(meta.get(k, {}))
        if k in fetch and "FETCH_SIZE" in fetch[k]:
            e["FETCH_SIZE_KB"] = fetch[k]["FETCH_SIZE"]
        if k in write and "WRITE_SIZE" in write[k]:
            e["WRITE_SIZE_KB"] = write[k]["WRITE_SIZE"]
        if "FETCH_SIZE_KB" in e or "WRITE_SIZE_KB" in e:
            e["hbm_bytes_per_launch_raw"] = (e.get("FETCH_SIZE_KB", 0.0) + e.get("WRITE_SIZE_KB", 0.0)) * 1024
            e["hbm_bytes_per_launch"] = (2.0 * e.get("FETCH_SIZE_KB", 0.0) + e.get("WRITE_SIZE_KB", 0.0)) * 1024
        if k in sq:
            e["sq"] = sq[k]
        summary["kernels"][k] = e
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(summary, open(a.out + ".json", "w"), indent=1, sort_keys=True)
    with open(a.out + ".md", "w") as f:
        f.write(f"# rocprofv3 summary: {os.path.basename(a.out)}\n\n{a.note}\n\n{clock_note}")
        f.write("The VGPR column is rocprofv3's `VGPR_Count` field, which on gfx950 reads about half of what the compiler "
                "allocates per lane (e.g. 84 for the 164 of `demux_oct_kernel`); the compiler's own numbers -- the ones "
                "occupancy follows -- are asserted in `tests/test_isa.py` and printed by `tools/kres.py`.\n\n")
        f.write("| kernel | calls | avg us | min us | max us | % | VGPR_Count (rocprofv3) | AGPR | LDS B | FETCH KB | WRITE KB | HBM MB/launch (fetch x2) |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k, e in summary["kernels"].items():
            f.write("| {} | {} | {:.1f} | {:.1f} | {:.1f} | {:.1f} | {} | {} | {} | {} | {} | {} |\n".format(
                k, e.get("calls", ""), e.get("avg_us", 0), e.get("min_us", 0), e.get("max_us", 0), e.get("pct", 0),
                e.get("vgpr", ""), e.get("agpr", ""), e.get("lds", ""),
                f"{e['FETCH_SIZE_KB']:.0f}" if "FETCH_SIZE_KB" in e else "",
                f"{e['WRITE_SIZE_KB']:.0f}" if "WRITE_SIZE_KB" in e else "",
                f"{e['hbm_bytes_per_launch'] / 1e6:.1f}" if "hbm_bytes_per_launch" in e else ""))
        for k, e in summary["kernels"].items():
            if "sq" in e:
                f.write(f"\n**{k}** SQ counters (avg per launch): " +
                        ", ".join(f"{c}={v:.4g}" for c, v in sorted(e["sq"].items())) + "\n")
    print(open(a.out + ".md").read())


if __name__ == "__main__":
    main()
