"""CPU checks of the drop-in boundary: the in-tree C-ABI library loads and exports every symbol include/muxgl.h
declares, the ctypes struct mirrors match the header's layout, and compute calls fail loudly without a GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from popscle_amd import muxgl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "muxgl.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(muxgl.LIB_PATH):
        from popscle_amd.build import build_lib

        build_lib()
    return muxgl.load_library()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(muxgl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/muxgl.h but not exported by libmuxgl.so"
    # and the binding knows every one of them
    assert set(names) == set(muxgl.SYMBOLS)


def test_version(lib):
    assert lib.muxgl_version() == 3


def test_struct_layouts_match_header(tmp_path):
    """compile a tiny C program against include/muxgl.h and compare sizeof/offsetof with the numpy/ctypes mirrors"""
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "muxgl.h"\n'
        "int main(void){\n"
        'printf("%zu %zu %zu %zu\\n", sizeof(muxgl_demux_cell), sizeof(muxgl_fmx_cell), sizeof(muxgl_demux_params), sizeof(muxgl_fmx_params));\n'
        'printf("%zu %zu %zu\\n", offsetof(muxgl_demux_cell, sngBestLLK), offsetof(muxgl_demux_cell, sngOnlyPP), offsetof(muxgl_fmx_cell, bestLLK));\n'
        'printf("%zu %zu\\n", offsetof(muxgl_demux_params, alpha), offsetof(muxgl_demux_params, doublet_prior));\n'
        'printf("%zu %zu %zu\\n", sizeof(muxgl_config), offsetof(muxgl_config, n_devices), offsetof(muxgl_config, device_ids));\n'
        'muxgl_config c = MUXGL_CONFIG_INIT;\n'
        'printf("%d %zu %zu\\n", (int)c.struct_size, offsetof(muxgl_config, struct_size), offsetof(muxgl_config, device_id));\n'
        "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    vals = list(map(int, out))
    assert vals[0] == muxgl.DEMUX_CELL.itemsize
    assert vals[1] == muxgl.FMX_CELL.itemsize
    assert vals[2] == ctypes.sizeof(muxgl._DemuxParams)
    assert vals[3] == ctypes.sizeof(muxgl._FmxParams)
    assert vals[4] == muxgl.DEMUX_CELL.fields["sngBestLLK"][1]
    assert vals[5] == muxgl.DEMUX_CELL.fields["sngOnlyPP"][1]
    assert vals[6] == muxgl.FMX_CELL.fields["bestLLK"][1]
    assert vals[7] == muxgl._DemuxParams.alpha.offset
    assert vals[8] == muxgl._DemuxParams.doublet_prior.offset
    assert vals[9] == ctypes.sizeof(muxgl._Config)
    assert vals[10] == muxgl._Config.n_devices.offset and vals[11] == muxgl._Config.device_ids.offset
    # the struct names its own size first, so that a caller built against another header is rejected, not misread
    assert vals[12] == vals[9] and vals[13] == 0 == muxgl._Config.struct_size.offset
    assert vals[14] == muxgl._Config.device_id.offset


def test_create_rejects_a_config_of_another_size(lib):
    cfg = muxgl._Config()
    cfg.struct_size = 8  # what a caller built against MUXGL_VERSION 1 would pass
    h = ctypes.c_void_p()
    assert lib.muxgl_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"struct_size" in lib.muxgl_last_error(None)


def test_oracle_and_library_records_share_a_layout():
    import oracle_binding as ob

    # the demuxlet record IS the oracle's (= the reference's variables); the freemuxlet record is the oracle's followed by
    # the two third-largest values the exact path reads, which the reference has no counterpart of
    assert ob.DEMUX_CELL == muxgl.DEMUX_CELL
    lib_t, ora_t = muxgl.FMX_CELL, ob.FMX_CELL
    n = len(ora_t.names)
    assert lib_t.names[:n] == ora_t.names and lib_t.names[n:] == ("sngThirdLLK", "dblThirdLLK")
    for f in ora_t.names:
        assert lib_t.fields[f][:2] == ora_t.fields[f][:2], f
    assert lib_t.itemsize == ora_t.itemsize + 16


def test_no_cpu_fallback(lib):
    """without a HIP device the product path must fail loudly, not route anywhere else"""
    try:
        import torch

        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    with pytest.raises(muxgl.MuxglError, match="no HIP device|no CPU fallback|hip"):
        muxgl.Engine(0)


def test_product_sources_do_not_reference_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "popscle_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "__init__.py" and False, f"{f} mentions the oracle"
