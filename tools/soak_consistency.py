#!/usr/bin/env python
"""GPU tool: round-3 kernels and schedule against round 2's, batch after batch on fresh signals.
Default build (gemm4p_kernel, same-XCD exchange with plain stores, overlapped schedule, pipelined decode) vs
XB_GEMM4=0 XB_LSTM_LOCAL=0 XB_OVERLAP=0 (gemm8r_kernel, write-through exchange, serial order): the called sequences of every
chunk of every batch must be identical bytes.  BATCHES (default 30) x N (default 512) chunks of 10 000 samples, nb 6 and 5."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xna_basecaller_amd import _lib                                   # noqa: E402
from xna_basecaller_amd.synthetic import peaky_weights, seeded_weights  # noqa: E402


def run(nb, N, sd, signals, env):
    import torch
    os.environ.update(env)
    ctx = _lib.Context(0, nb, 3, 768, 19, 5, 5.0, 2.0, 10000, N, precision=_lib.XB_PREC_F16F8)
    for k in env:
        os.environ.pop(k)
    ctx.load_state_dict(sd)
    alphabet = "NACGTXY"[:nb + 1]
    outs = []
    bufs = [(torch.empty((N, ctx.T), dtype=torch.int8, device="cuda"), torch.empty((N,), dtype=torch.int32, device="cuda"))
            for _ in signals]
    for x, (s, l) in zip(signals, bufs):
        ctx.basecall_chunks_dev(x.data_ptr(), N, alphabet, s.data_ptr(), l.data_ptr())
    ctx.synchronize()
    for s, l in bufs:
        outs.append((s.cpu().numpy(), l.cpu().numpy()))
    ctx.close()
    return outs


def main():
    import torch
    N = int(os.environ.get("N", 512))
    B = int(os.environ.get("BATCHES", 30))
    bad = 0
    for nb, weights in ((6, "peaky"), (5, "seeded")):
        sd = peaky_weights(768, nb) if weights == "peaky" else seeded_weights(768, nb)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(1234 + nb)
        signals = [torch.randn((N, 10000), dtype=torch.float32, device="cuda", generator=gen) for _ in range(B)]
        new = run(nb, N, sd, signals, {})
        old = run(nb, N, sd, signals, {"XB_GEMM4": "0", "XB_LSTM_LOCAL": "0", "XB_OVERLAP": "0"})
        diff = sum(int(not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))) for a, b in zip(new, old))
        called = sum(int(a[1].sum()) for a in new)
        print("nb %d %s weights: %d batches x %d chunks, %d bases called, batches that differ: %d" % (nb, weights, B, N, called, diff), flush=True)
        bad += diff
    print("soak_consistency:", "OK" if bad == 0 else "%d DIFFERENCES" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
