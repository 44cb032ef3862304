#!/usr/bin/env python
"""Per-kernel register/LDS/occupancy table from hipcc -Rpass-analysis=kernel-resource-usage.
usage: python tools/kres.py popscle_amd/csrc/demux_row.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
       "-Iinclude", "-I../../include", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    txt = m.group(1)
    if txt.startswith("Function Name:") or txt.startswith("Name:"):
        if cur: rows.append(cur)
        cur = {"name": txt.split(":",1)[1].strip()}
    elif ":" in txt:
        k, v = txt.split(":",1); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = name.split("(")[0]
    if flt and flt not in name: continue
    print(f"{name[:70]:70s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} spill {r.get('VGPR Spill','?'):>3} "
          f"SGPR {r.get('TotalSGPRs', r.get('SGPRs','?')):>3} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')}")
