#!/usr/bin/env python
"""GPU tool: end-to-end label identity (GPU encoder + GPU decode vs all-oracle) on the peaky synthetic model
(synthetic.peaky_weights) at the timed size: features 768, T = 2000, N = 512.  Prints, per alphabet: bases per step,
score error (max / rms), label mismatch rate between the GPU path and the all-oracle path on `picks` sampled chunks, and
the per-chunk called-length differences.  The oracle is the checker here, as in the tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                             # noqa: E402
from xna_basecaller_amd import _lib                                        # noqa: E402
from xna_basecaller_amd.synthetic import peaky_weights                     # noqa: E402


def main():
    import torch
    F, L, N = 768, 10000, int(os.environ.get("PEAKY_N", "512"))
    npick = int(os.environ.get("PEAKY_PICKS", "24"))
    for nb in (6, 5):
        for prec in (_lib.XB_PREC_MIXED, _lib.XB_PREC_F16F8, _lib.XB_PREC_F16X3):
            ig, lg, bb = 2.0, 10.0, 2.0
            alphabet = "NACGTXY"[:nb + 1]
            sd = peaky_weights(F, nb, input_gain=ig, linear_gain=lg, blank_bias=bb)
            ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=prec)
            ctx.load_state_dict(sd)
            T = ctx.T
            gen = torch.Generator(device="cuda")
            gen.manual_seed(25)
            d_signal = torch.randn((N, L), dtype=torch.float32, device="cuda", generator=gen)
            d_seq = torch.empty((N, T), dtype=torch.int8, device="cuda")
            d_len = torch.empty((N,), dtype=torch.int32, device="cuda")
            ctx.basecall_chunks_dev(d_signal.data_ptr(), N, alphabet, d_seq.data_ptr(), d_len.data_ptr())
            ctx.synchronize()
            lens = d_len.cpu().numpy()
            picks = np.linspace(0, N - 1, npick).astype(int)
            d_scores = torch.empty((T, N, ctx.C_noblank), dtype=torch.float32, device="cuda")
            ctx.encode_dev(d_signal.data_ptr(), N, False, d_scores.data_ptr())
            ctx.synchronize()
            sc = d_scores[:, picks, :].cpu().numpy()
            x = d_signal[picks].cpu().numpy()
            seqs = d_seq.cpu().numpy()[picks]
            del d_scores
            ctx.close()
            lab_g = oracle.decode(sc, nb, 3, blank_score=2.0)["labels"]           # oracle decode of the GPU's scores
            gseq, _, glen = oracle.pack(lab_g, alphabet)
            same_dec = bool(np.array_equal(glen, lens[picks]) and np.array_equal(gseq, seqs))
            ref = oracle.encode(x, sd, F, nb, 3, expand_blanks=False)
            lab_o = oracle.decode(ref, nb, 3, blank_score=2.0)["labels"]
            _, _, olen = oracle.pack(lab_o, alphabet)
            err = np.abs(ref - sc)
            print("nb %d %s gains (%.1f, %.1f, %.1f): bases/step %.3f (whole batch %.3f)  score err max %.2e rms %.2e  "
                  "GPU decode == oracle decode of GPU scores: %s  label mismatch vs all-oracle %.3e (%d of %d)  "
                  "len diff per chunk: max %d, chunks differing %d of %d"
                  % (nb, {0: "f16x3", 2: "f16f8", 4: "mixed"}[prec], ig, lg, bb, (lab_o != 0).mean(), lens.mean() / T, err.max(), np.sqrt((err ** 2).mean()), same_dec,
                     (lab_o != lab_g).mean(), int((lab_o != lab_g).sum()), lab_o.size,
                     int(np.abs(olen - glen).max()), int((olen != glen).sum()), npick), flush=True)


if __name__ == "__main__":
    main()
