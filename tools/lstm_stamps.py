#!/usr/bin/env python
"""Diagnostic: per-phase cycle breakdown of the persistent LSTM kernel (needs `make -C xna_basecaller_amd/csrc diag`).
Run on the GPU box:  XNA_LIBXNACALL=xna_basecaller_amd/libxnacall_diag.so python tools/lstm_stamps.py"""
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
nb, F = 5, 768
ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=int(os.environ.get("PREC", 0)))
keys, shapes = encoder_shapes(F, nb)
ctx.load_state_dict(seeded_state_dict(keys, shapes, 25))
x = np.random.default_rng(0).standard_normal((N, L)).astype(np.float32)
ctx.basecall_chunks(x, "NACGTX")
out = (C.c_ulonglong * 10)()
ctx.lib.xb_debug_lstm_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.lib.xb_debug_lstm_stamps(ctx.h, out, 1)
ctx.set_profiling(True)
ctx.reset_stage_times()
ctx.basecall_chunks(x, "NACGTX")
st = ctx.stage_times()
ctx.lib.xb_debug_lstm_stamps(ctx.h, out, 1)
names = ["loop / y stores", "gin issue + group wait", "first piece landed", "piece compute (ds_read+MFMA)",
         "pointwise + h stores issued", "stores drained + barrier", "arrive", "piece DMA wait + barrier"]
gsteps = max(int(out[9]), 1)          # group-steps of workgroup 0 (two per time step when it serves two groups)
steps = gsteps
tot = sum(out[:8])
print("group-steps of workgroup 0: %d, first piece requested early in %d" % (out[9], out[8]))
print("stage ms:", {k: round(v[0], 2) for k, v in st.items()})
for i, nme in enumerate(names):
    print("%-30s %10.0f cycles/step  %5.1f %%" % (nme, out[i] / steps, 100.0 * out[i] / max(tot, 1)))
print("total %.0f cycles/step (s_memtime ticks of 100 MHz? see note) over %d steps" % (tot / steps, steps))
