#!/usr/bin/env python
"""Diagnostic: how the score error of the precision modes grows with the scale of the LSTM / linear weights (the seeded
N(0, 1/sqrt(fan_in)) weights times 1, 2, 4), features 768, nb 6, against the fp32 oracle encoder.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from xna_basecaller_amd import _lib
from xna_basecaller_amd.synthetic import encoder_shapes, seeded_state_dict
F, nb, L, N = 768, 6, 2500, 6
keys, shapes = encoder_shapes(F, nb)
base = seeded_state_dict(keys, shapes, seed=25)
x = np.random.default_rng(3).standard_normal((N, L)).astype(np.float32)
for g in (1.0, 2.0, 4.0):
    sd = {k: (v * np.float32(g) if (".rnn." in k or ".linear." in k) else v) for k, v in base.items()}
    ref = oracle.encode(x, sd, F, nb, 3, expand_blanks=False)
    sat = float((np.abs(ref) > 4.9).mean())
    for name, prec, mask in (("f16f8", _lib.XB_PREC_F16F8, 0), ("in1[0-2]", _lib.XB_PREC_F16F8_IN1, 7), ("in1[all]", _lib.XB_PREC_F16F8_IN1, 31), ("f16", _lib.XB_PREC_F16, 0)):
        os.environ["XB_IN1_LAYERS"] = str(mask)
        ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=prec)
        ctx.load_state_dict(sd)
        err = np.abs(ctx.encode(x, expand_blanks=False) - ref)
        ctx.close()
        print("weights x%.0f (|score|>4.9: %.2f)  %-9s max %.2e rms %.2e" % (g, sat, name, err.max(), np.sqrt((err**2).mean())), flush=True)
