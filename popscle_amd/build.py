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


def source_hash() -> str:
    """fingerprint of the kernel sources (csrc/*.hip, *.hpp, include/muxgl.h): stamps measurements that are only valid
    for the code they were taken on (profiles/traffic.json); works on the GPU box, where there is no .git"""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "muxgl.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
