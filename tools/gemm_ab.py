#!/usr/bin/env python
"""GPU tool: gemm4p_kernel (two workgroups per CU, fragment-major weights, the product's GEMM) against gemm8r_kernel (XB_GEMM4=0).

Both kernels add the same products in the same order per accumulator, so the encoder's scores must be BIT-identical.
Shapes cover the interior fast path, ragged M / N edges and batches that are not multiples of 128 (the member-major
gin epilogue's wrap path).  With --time also the per-stage times of one full-size step under both kernels.

  python tools/gemm_ab.py [--time]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from xna_basecaller_amd import _lib                                          # noqa: E402
from xna_basecaller_amd.synthetic import encoder_shapes, seeded_state_dict     # noqa: E402

PREC = {"f16x3": _lib.XB_PREC_F16X3, "f16": _lib.XB_PREC_F16, "f16f8": _lib.XB_PREC_F16F8, "f16f8i": _lib.XB_PREC_F16F8_IN1}


def encode(gemm4, F, nb, L, N, prec, x, sd, lstm_mode=0):
    os.environ["XB_GEMM4"] = str(int(gemm4))
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=PREC[prec], lstm_mode=lstm_mode)
    ctx.load_state_dict(sd)
    out = ctx.encode(x)
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    args = ap.parse_args()
    _lib.require_gpu()
    bad = 0
    cases = [(32, 4, 300, 5, "f16f8"), (64, 5, 600, 3, "f16f8"), (96, 4, 800, 70, "f16x3"), (96, 5, 400, 9, "f16f8"), (96, 5, 400, 9, "f16"), (128, 6, 400, 2, "f16"), (256, 6, 1000, 130, "f16f8"),
             (768, 6, 1000, 98, "f16f8"), (768, 6, 500, 448, "f16f8"), (768, 5, 500, 513, "f16f8"),
             (768, 6, 500, 256, "f16x3"), (768, 6, 500, 200, "f16f8i"), (768, 6, 500, 40, "f16f8")]
    for F, nb, L, N, prec in cases:
        keys, shapes = encoder_shapes(F, nb)
        sd = seeded_state_dict(keys, shapes, seed=F + nb)
        x = np.random.default_rng(L + N).standard_normal((N, L)).astype(np.float32)
        a = encode(0, F, nb, L, N, prec, x, sd)
        for kern in (1,):
            b = encode(kern, F, nb, L, N, prec, x, sd)
            same = np.array_equal(a, b)
            print("F %4d nb %d L %5d N %4d %-6s gemm4=%d : %s  (max |diff| %.3g, finite %s)"
                  % (F, nb, L, N, prec, kern, "bit-identical" if same else "DIFFERENT", float(np.abs(a - b).max()),
                     bool(np.isfinite(b).all())), flush=True)
            bad += 0 if same else 1
    if args.time:
        import torch
        F, nb, L = 768, 6, 10000
        keys, shapes = encoder_shapes(F, nb)
        sd = seeded_state_dict(keys, shapes, seed=25)
        for N in (512, 448, 98):
            for overlap in ("0", "1"):
                for g4 in (0, 1):
                    os.environ["XB_GEMM4"] = str(g4)
                    os.environ["XB_OVERLAP"] = overlap
                    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8)
                    ctx.load_state_dict(sd)
                    ctx.reserve_pairing()          # room for two co-scheduled calls now, not inside a timed loop
                    T = ctx.T
                    d_signal = torch.randn((N, L), dtype=torch.float32, device="cuda")
                    d_seq = torch.empty((N, T), dtype=torch.int8, device="cuda")
                    d_len = torch.empty((N,), dtype=torch.int32, device="cuda")
                    for _ in range(2):
                        ctx.basecall_chunks_dev(d_signal.data_ptr(), N, "NACGTXY", d_seq.data_ptr(), d_len.data_ptr())
                    ctx.synchronize()
                    ctx.set_profiling(True)
                    ctx.reset_stage_times()
                    K = 4
                    t0 = time.perf_counter()
                    for _ in range(K):
                        ctx.basecall_chunks_dev(d_signal.data_ptr(), N, "NACGTXY", d_seq.data_ptr(), d_len.data_ptr())
                    ctx.synchronize()
                    dt = (time.perf_counter() - t0) / K
                    st = ctx.stage_times()
                    print("N %4d overlap %s gemm4 %d : %.2f ms/step  %s" % (
                        N, overlap, g4, 1e3 * dt, "  ".join("%s %.1f" % (k, v[0] / K) for k, v in st.items())), flush=True)
                    ctx.close()
        os.environ.pop("XB_OVERLAP", None)
    os.environ.pop("XB_GEMM4", None)
    print("gemm_ab:", "OK" if bad == 0 else "%d MISMATCHES" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
