"""popscle_amd -- MI355X-native genotype-likelihood engine for popscle's demuxlet / freemuxlet hot path.

The product is the C-ABI shared library ``popscle_amd/lib/libmuxgl.so`` (HIP kernels for gfx950, declared in
``include/muxgl.h``).  This package holds its sources (``csrc/``), a ctypes binding (``muxgl``), the synthetic
pileup generator used by the tests and the benchmark (``synth``), and the host-side drivers that mirror the
reference commands' flow around the hot path (``demuxlet``, ``freemuxlet``).
"""

__all__ = ["muxgl", "synth"]
