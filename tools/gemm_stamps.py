#!/usr/bin/env python
"""Diagnostic: where a gemm4p_kernel<*, 3> workgroup's cycles go, on REAL data (needs `make -C xna_basecaller_amd/csrc diag`).
Timing-only ablations are no substitute here: the kernel runs at the clock the chip holds under its load, and garbage or zero operands
raise that clock by up to 20 % (MI355X_MICROARCH.md, DVFS give-back).
Run on the GPU box:  XNA_LIBXNACALL=xna_basecaller_amd/libxnacall_diag.so XB_OVERLAP=0 python tools/gemm_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import encoder_shapes, seeded_state_dict  # noqa: E402
from xna_basecaller_amd import _lib  # noqa: E402

N = int(os.environ.get("N", 512))
L = int(os.environ.get("L", 10000))
nb, F = 6, 768
ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N)
keys, shapes = encoder_shapes(F, nb)
ctx.load_state_dict(seeded_state_dict(keys, shapes, 25))
x = np.random.default_rng(0).standard_normal((N, L)).astype(np.float32)
out = (C.c_ulonglong * 8)()
ctx.lib.xb_debug_gemm_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for rep in range(3):
    ctx.basecall_chunks(x, "NACGTXY")
ctx.lib.xb_debug_gemm_stamps(ctx.h, out, 1)
ctx.set_profiling(True)
ctx.reset_stage_times()
ctx.basecall_chunks(x, "NACGTXY")
st = ctx.stage_times()
ctx.lib.xb_debug_gemm_stamps(ctx.h, out, 1)
names = ["prologue (start to the first k-tile)", "top of a k-tile: A landed + barrier", "weight pieces landed (+ first fragment reads issued)",
         "MFMA phases of the k-tiles", "loop end to kernel end (drain, epilogue, stores left the wave)"]
wgs, kt = max(int(out[6]), 1), max(int(out[7]), 1)
tot = sum(out[:5])
print("three-product workgroups: %d (%d of them in an odd wave slot of their SIMD), k-tiles per workgroup: %.1f" % (wgs, out[5], kt / wgs))
print("stage ms:", {k: round(v[0], 2) for k, v in st.items()})
for i, nme in enumerate(names):
    print("%-66s %9.0f cycles per workgroup  %5.1f %%" % (nme, out[i] / wgs, 100.0 * out[i] / max(tot, 1)))
print("total %.0f cycles per workgroup (wave 0's view); MFMA pipe time of a workgroup's 24 x 96 MFMAs of 16 cycles: 36864 (two workgroups share a CU)" % (tot / wgs))
