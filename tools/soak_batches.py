#!/usr/bin/env python
"""GPU tool: the fused path at many batch sizes (features 768, chunksize 10 000, 6-base CRF, default precision).
Per batch size: two back-to-back calls in the default (overlapped) schedule must give the same bytes as each other and as the
serial order (XB_OVERLAP=0, one group per workgroup) -- every chunk -- and the per-step time of both schedules is printed with
the per-chunk time relative to batch 512.  Covers what the trailing batch of a real run looks like (any size), the
reference's own `-b 98`, sizes around the group / slab seams (63 | 64 | 65, 511 | 512 | 513, 1023 | 1024 | 1025)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xna_basecaller_amd import _lib                                  # noqa: E402
from xna_basecaller_amd.synthetic import seeded_weights               # noqa: E402


def run(N, sd, d_signal, env, steps=3):
    import torch
    os.environ.update(env)
    ctx = _lib.Context(0, 6, 3, 768, 19, 5, 5.0, 2.0, 10000, N, precision=_lib.XB_PREC_MIXED)
    for k in env:
        os.environ.pop(k)
    ctx.load_state_dict(sd)
    ctx.reserve_pairing()          # room for two co-scheduled calls now, not inside a timed loop
    T = ctx.T
    seqs = [torch.full((N, T), -1, dtype=torch.int8, device="cuda") for _ in range(steps)]
    lens = [torch.full((N,), -1, dtype=torch.int32, device="cuda") for _ in range(steps)]
    ctx.basecall_chunks_dev(d_signal.data_ptr(), N, "NACGTXY", seqs[0].data_ptr(), lens[0].data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for s, l in zip(seqs, lens):
        ctx.basecall_chunks_dev(d_signal.data_ptr(), N, "NACGTXY", s.data_ptr(), l.data_ptr())
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = [s.cpu().numpy() for s in seqs], [l.cpu().numpy() for l in lens]
    ctx.close()
    return dt, out


def main():
    import torch
    sizes = [int(a) for a in sys.argv[1:]] or [1, 7, 63, 64, 65, 98, 130, 200, 384, 448, 500, 511, 512, 513, 640, 1000, 1023, 1024, 1025, 1536]
    sd = seeded_weights(768, 6)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(25)
    full = torch.randn((max(sizes), 10000), dtype=torch.float32, device="cuda", generator=gen)
    ref = None
    bad = 0
    for N in sizes:
        x = full[:N].contiguous()
        dt_o, (so, lo) = run(N, sd, x, {})
        dt_s, (ss, ls) = run(N, sd, x, {"XB_OVERLAP": "0", "XB_LSTM_DUAL": "0"}, steps=2)
        same = all(np.array_equal(so[0], s) for s in so[1:]) and np.array_equal(so[0], ss[0]) and np.array_equal(lo[0], ls[0])
        prefix = ref is None or (np.array_equal(so[0][:min(N, ref[0].shape[0])], ref[0][:min(N, ref[0].shape[0])]))
        if N == 512:
            ref512 = dt_o
        if ref is None or N > ref[0].shape[0]:
            ref = (so[0], lo[0])
        bad += 0 if (same and prefix) else 1
        print("batch %5d : overlapped %7.2f ms/step (%.4f ms per chunk)  serial %7.2f ms/step   schedules agree: %s   chunk results independent of the batch: %s"
              % (N, 1e3 * dt_o, 1e3 * dt_o / N, 1e3 * dt_s, same, prefix), flush=True)
    print("soak:", "OK" if bad == 0 else "%d FAILURES" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
