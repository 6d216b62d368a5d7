#!/usr/bin/env python
"""Diagnostic: score error (vs the fp32 oracle encoder, features 768, nb 6) of the opt-in arithmetic switches combined:
precision f16f8i with XB_IN1_LAYERS and the int8-limb recurrence XB_LSTM_I8."""
import os, sys
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
for prec, mask, i8 in ((_lib.XB_PREC_F16F8, 0, 0), (_lib.XB_PREC_F16F8, 0, 1), (_lib.XB_PREC_F16F8, 0, 2),
                       (_lib.XB_PREC_F16F8_IN1, 7, 0), (_lib.XB_PREC_F16F8_IN1, 7, 2), (_lib.XB_PREC_F16F8_IN1, 31, 2)):
    os.environ["XB_IN1_LAYERS"] = str(mask)
    os.environ["XB_LSTM_I8"] = str(i8)
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=prec)
    ctx.load_state_dict(sd)
    e = np.abs(ctx.encode(x, expand_blanks=False) - ref)
    ctx.close()
    print("precision %s in1 mask %2d i8 %d: max %.2e rms %.2e" % ("f16f8i" if prec == _lib.XB_PREC_F16F8_IN1 else "f16f8 ", mask, i8,
                                                                e.max(), np.sqrt((e ** 2).mean())), flush=True)
