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


# translation units that hold the device code behind a BASELINE config's dominant kernel family (its kernels, the plan
# kernels that cut their work, the kernels that prepare their inputs): the counters in profiles/traffic.json are stamped
# per config with a fingerprint of the gfx950 MACHINE CODE of these units, so host-side edits, comments and declarations
# do not invalidate a measurement, and any change of the instructions does
FAMILY_UNITS = {
    1: ["demux_oct", "plan_kernels"],
    2: ["demux_wave", "demux_ring", "demux_kernels", "plan_kernels"],
    3: ["fmx_oct", "fmx_kernels", "plan_kernels"],
    4: ["fmx_wave", "fmx_kernels", "plan_kernels"],
}
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
HASH_CACHE = os.path.join(HERE, "lib", "code_hashes.json")


def _unit_code_hash(unit: str) -> str:
    """sha256 of the .text and .rodata sections of the gfx950 code object embedded in popscle_amd/lib/<unit>.o"""
    import hashlib
    import tempfile

    obj = os.path.join(HERE, "lib", unit + ".o")
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "u.fatbin"), os.path.join(td, "u.co")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--type=o", "--unbundle",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co],
                       check=True, capture_output=True)
        h = hashlib.sha256()
        for sec in (".text", ".rodata"):
            out = os.path.join(td, "sec.bin")
            subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=" + sec, co, out], check=True)
            h.update(sec.encode())
            h.update(open(out, "rb").read())
    return h.hexdigest()[:16]


def source_hash(config: int | None = None) -> str | None:
    """config given: fingerprint of the machine code of that config's kernel family (FAMILY_UNITS), cached in
    popscle_amd/lib/code_hashes.json next to the objects it was taken from (the cache travels with the built library;
    it is recomputed when an object file is newer).  None when it cannot be determined (no objects, no LLVM tools).
    No config: fingerprint of all kernel SOURCES (csrc/*.hip, *.hpp, include/muxgl.h)."""
    import glob
    import hashlib
    import json

    if config is not None:
        units = FAMILY_UNITS[config]
        objs = [os.path.join(HERE, "lib", u + ".o") for u in units]
        try:
            newest = max(os.path.getmtime(o) for o in objs)
            cache = {}
            if os.path.exists(HASH_CACHE) and os.path.getmtime(HASH_CACHE) >= newest:
                cache = json.load(open(HASH_CACHE))
            if any(u not in cache for u in units):
                cache = {u: _unit_code_hash(u) for us in FAMILY_UNITS.values() for u in us}
                json.dump(cache, open(HASH_CACHE, "w"), indent=1)
        except Exception:
            return None
        return hashlib.sha256("".join(u + cache[u] for u in units).encode()).hexdigest()[:16]
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "muxgl.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
