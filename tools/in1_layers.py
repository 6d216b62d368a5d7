#!/usr/bin/env python
"""Diagnostic: which LSTM input projections tolerate the single-product form (XB_PREC_F16F8_IN1)?  For every layer mask in
XB_IN1_LAYERS: max / rms score error against the fp32 oracle encoder (features 768, nb 6) and, with BENCH=1, the step time."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from xna_basecaller_amd import _lib
from xna_basecaller_amd.synthetic import encoder_shapes, seeded_state_dict

F, nb, L, N = 768, 6, 2500, 6
keys, shapes = encoder_shapes(F, nb)
sd = seeded_state_dict(keys, shapes, seed=25)
x = np.random.default_rng(3).standard_normal((N, L)).astype(np.float32)
ref = oracle.encode(x, sd, F, nb, 3, expand_blanks=False)
masks = [0, 1, 2, 4, 8, 16, 30, 28, 24, 3, 7, 15, 31]
for m in masks:
    os.environ["XB_IN1_LAYERS"] = str(m)
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8_IN1)
    ctx.load_state_dict(sd)
    err = np.abs(ctx.encode(x, expand_blanks=False) - ref)
    ctx.close()
    line = "layers %s: max %.2e rms %.2e" % (format(m, "05b")[::-1], err.max(), np.sqrt((err ** 2).mean()))
    if os.environ.get("BENCH") == "1":
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-chunks", "0",
                              "--precision", "f16f8i"], capture_output=True, text=True, env=dict(os.environ)).stdout
        d = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
        if d:
            line += "  %.1f ms/step" % d[0]["ms_per_step"]
    print(line, flush=True)
