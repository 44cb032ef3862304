"""In-tree build of libmuxgl.so (hipcc, gfx950).  Used by __graft_entry__.build() and by developers."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmuxgl.so")


def build_lib(force: bool = False, jobs: int = 3) -> str:
    """Compile every HIP translation unit for gfx950 and link popscle_amd/lib/libmuxgl.so."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True)
    subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], check=True)
    if not os.path.exists(LIB):
        raise RuntimeError("libmuxgl.so was not produced")
    return LIB


if __name__ == "__main__":
    print(build_lib())


# sources the dominant kernel family of a BASELINE config is built from (kernels, the headers they include, the plan
# kernels that cut their work): counters in profiles/traffic.json are stamped per config with a fingerprint of these
FAMILY_SOURCES = {
    1: ["demux_oct.hip", "oct_tiling.hpp", "demux_entry.hpp", "demux_call_body.hpp", "common.hpp", "plan_kernels.hip"],
    2: ["demux_wave.hip", "demux_ring.hip", "demux_kernels.hip", "demux_entry.hpp", "common.hpp", "plan_kernels.hip"],
    3: ["fmx_oct.hip", "oct_tiling.hpp", "fmx_kernels.hip", "common.hpp", "plan_kernels.hip"],
    4: ["fmx_wave.hip", "fmx_kernels.hip", "common.hpp", "plan_kernels.hip"],
}


def source_hash(config: int | None = None) -> str:
    """fingerprint of the kernel sources (all of csrc/*.hip, *.hpp and include/muxgl.h, or the files of one config's
    dominant kernel family): stamps measurements that are only valid for the code they were taken on
    (profiles/traffic.json); works on the GPU box, where there is no .git"""
    import glob
    import hashlib

    h = hashlib.sha256()
    if config is None:
        files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))
        files.append(os.path.join(os.path.dirname(HERE), "include", "muxgl.h"))
    else:
        files = [os.path.join(CSRC, f) for f in FAMILY_SOURCES[config]]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
