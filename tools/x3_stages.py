#!/usr/bin/env python
"""GPU tool: which contraction stages of the f16f8 encoder need the three-product (f16x3) arithmetic?  For every stage mask in
XB_X3_STAGES (bits 0-4 input projection of LSTM layer l, 5-9 recurrence of layer l, 10 CRF linear layer, 11 conv3): max / rms
score error against the fp32 oracle encoder on the peaky synthetic model (synthetic.peaky_weights: the trained-like regime) and
on the plain seeded model, features 768, T = 2000.  The oracle is the checker here, as in the tests.
usage: x3_stages.py [mask ...]   (masks in any int() base-0 form; default: the attribution ladder)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                             # noqa: E402
from xna_basecaller_amd import _lib                                        # noqa: E402
from xna_basecaller_amd.synthetic import peaky_weights, seeded_weights    # noqa: E402

IN, REC, LIN, CONV = (lambda l: 1 << l), (lambda l: 1 << (5 + l)), 1 << 10, 1 << 11
LADDER = [0, LIN, LIN | REC(4), LIN | REC(4) | IN(4), LIN | REC(4) | IN(4) | REC(3), LIN | REC(4) | IN(4) | REC(3) | IN(3),
          REC(4), IN(4), REC(4) | IN(4), sum(REC(l) for l in range(5)), sum(IN(l) for l in range(5)), CONV, 0xfff]


def name(m):
    parts = []
    if m & CONV:
        parts.append("conv")
    parts += ["in%d" % l for l in range(5) if m & IN(l)] + ["rec%d" % l for l in range(5) if m & REC(l)]
    if m & LIN:
        parts.append("lin")
    return "+".join(parts) if parts else "none (plain f16f8)"


def main():
    F, nb, L = 768, int(os.environ.get("X3_NB", "6")), 10000
    N = int(os.environ.get("X3_N", "32"))
    masks = [int(a, 0) for a in sys.argv[1:]] or LADDER
    x = np.random.default_rng(25).standard_normal((N, L)).astype(np.float32)
    for label, sd in (("peaky", peaky_weights(F, nb)), ("seeded", seeded_weights(F, nb))):
        t0 = time.time()
        ref = oracle.encode(x, sd, F, nb, 3, expand_blanks=False)
        print("# %s model, nb %d, %d chunks x %d samples (oracle %.0f s)" % (label, nb, N, L, time.time() - t0), flush=True)
        for m in masks:
            os.environ["XB_X3_STAGES"] = str(m)
            ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8)
            ctx.load_state_dict(sd)
            err = np.abs(ctx.encode(x, expand_blanks=False) - ref)
            ctx.close()
            print("%-8s x3 stages 0x%03x %-36s max %.2e  rms %.2e  p99.99 %.2e"
                  % (label, m, name(m), err.max(), np.sqrt((err.astype(np.float64) ** 2).mean()), np.quantile(err, 0.9999)), flush=True)
        os.environ.pop("XB_X3_STAGES", None)
        ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16X3)
        ctx.load_state_dict(sd)
        err = np.abs(ctx.encode(x, expand_blanks=False) - ref)
        ctx.close()
        print("%-8s precision f16x3 %-41s max %.2e  rms %.2e" % (label, "", err.max(), np.sqrt((err.astype(np.float64) ** 2).mean())), flush=True)


if __name__ == "__main__":
    main()
