"""The ISA facts the kernels' designs rely on, checked on the compiler's output (no GPU needed: hipcc cross-compiles).

The sweeps are software pipelines whose performance rests on how hipcc places loads and waits -- things a compiler
upgrade can change without any test of results noticing:
  * demux_oct_kernel (the BASELINE metric's kernel): three waves per SIMD (<= 168 VGPRs), no scratch, 12.5 KB of LDS;
    its loop over the linear entries issues its row and table loads in front of each sweep and never waits for ALL
    outstanding loads (no `s_waitcnt vmcnt(0)`), the records' offsets come through one 64-byte load per step;
  * fmx_estep_wave_kernel / demux_wave kernels: the ring of partner values is read with single `ds_read_b64` at immediate
    offsets (inline asm, common.hpp: a merged `ds_read2_b64` costs 8 LDS cycles instead of 2 + 2).
The hipcc version is printed: profiles/ name the one they were taken with.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "popscle_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def isa(tmp_path_factory, name):
    out = str(tmp_path_factory.mktemp("isa") / (name + ".s"))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
                    "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                    os.path.join(CSRC, name + ".hip")], check=True, capture_output=True)
    return open(out).read()


def kernels(text, pattern):
    """{mangled name: (body, metadata dict)} of the kernels whose mangled name contains `pattern`"""
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        if pattern in m.group(1):
            meta = {k: int(v) for k, v in re.findall(r"\.set " + re.escape(m.group(1)) + r"\.(\w+), (\d+)", text)}
            out[m.group(1)] = (m.group(2), meta)
    return out


def blocks(body):
    """the basic blocks of a kernel body (split at labels), comments stripped"""
    out, cur = [], []
    for line in body.split("\n"):
        line = line.split(";")[0].rstrip()
        if re.match(r"^\.LBB\d+_\d+:", line):
            out.append("\n".join(cur))
            cur = []
        elif line.strip():
            cur.append(line)
    out.append("\n".join(cur))
    return out


@pytest.fixture(scope="module")
def hipcc_version():
    v = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    print("hipcc:", v)
    return v


def test_oct_kernel_resources_and_linear_loop(tmp_path_factory, hipcc_version):
    ks = kernels(isa(tmp_path_factory, "demux_oct"), "demux_oct_kernelILi8ELb")
    assert len(ks) == 2  # eight lanes per entry, <UNIT_S = true / false>
    for name, (body, meta) in ks.items():
        assert meta["num_vgpr"] <= 168 and meta["num_agpr"] == 0, (name, meta)  # three waves per SIMD
        assert meta["private_seg_size"] == 0, "scratch (spills) in the sweep kernel"
        # the loop of the linear entries: the one with DPP rotations and table reads but no LDS writes
        lin = [t for t in blocks(body) if t.count("row_ror:2 ") >= 12 and "ds_read_b128" in t and "ds_write" not in t
               and "v_frexp_mant" not in t.split("s_cbranch")[0]]
        assert len(lin) == 1, [len(t) for t in blocks(body)]
        t = lin[0]
        assert "s_waitcnt vmcnt(0)" not in t, "the linear loop drains its loads"
        rows = len(re.findall(r"global_load_dwordx4", t))
        assert rows == (3 if "ELb1E" in name else 6), rows  # 3 unrolled steps x (1 | 2) row pieces
        assert len(re.findall(r"v_mov_b32_dpp", t)) == 3 * 16  # two rho per rotation, four rotations, two dwords each
        fp = len(re.findall(r"v_(fma|mul|fmac|add)_f64", t))
        assert fp <= 3 * 42 + 45, fp  # <= 42 FP64 per step; the renormalisation rides in the same block
    m = re.search(r"demux_oct_kernelILi8ELb1E.*?LDSByteSize: (\d+)", isa(tmp_path_factory, "demux_oct"), re.S)
    assert m and int(m.group(1)) <= 12800


def test_oct_kernel_sixteen_lanes_per_entry(tmp_path_factory, hipcc_version):
    """16 < V <= 32: the same kernel with sixteen lanes per entry (34 accumulators per lane): two waves per SIMD, its
    lambdas inlined (without always_inline hipcc compiled the batch of sixteen entries as calls through scratch memory),
    seven full rotations + the facing one per linear step, no drain of the loads in the linear loop"""
    ks = kernels(isa(tmp_path_factory, "demux_oct"), "demux_oct_kernelILi16ELb")
    assert len(ks) == 2
    for name, (body, meta) in ks.items():
        assert meta["num_vgpr"] <= 256 and meta["num_agpr"] == 0, (name, meta)
        assert "s_swappc" not in body and "s_setpc" not in body, "a lambda of the sweep was not inlined"
        assert len(re.findall(r"v_(fma|mul|fmac)_f64", body)) > 5000  # three batch variants x sixteen entries, inline
        lin = [t for t in blocks(body) if t.count("row_ror:1 ") >= 12 and "ds_read_b128" in t and "ds_write" not in t
               and "v_frexp_mant" not in t.split("s_cbranch")[0]]
        assert len(lin) == 1, [len(t) for t in blocks(body)]
        t = lin[0]
        assert "s_waitcnt vmcnt(0)" not in t, "the linear loop drains its loads"
        assert len(re.findall(r"v_mov_b32_dpp", t)) == 3 * 32  # two rho per rotation, eight rotations, two dwords each


def test_wave_ring_reads_are_single_b64(tmp_path_factory, hipcc_version):
    text = isa(tmp_path_factory, "fmx_wave")
    ks = kernels(text, "fmx_estep_wave_kernel")
    assert ks
    for name, (body, meta) in ks.items():
        big = max(blocks(body), key=len, default="")
        assert len(re.findall(r"ds_read_b64 v\[\d+:\d+\], v\d+ offset:\d+", big)) >= 24, name
        assert "ds_read2_b64" not in big, f"{name}: merged ring reads"
