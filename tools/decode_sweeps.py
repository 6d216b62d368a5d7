#!/usr/bin/env python
"""Diagnostic: time of the CRF decode alone (random 5*tanh scores resident in HBM, no blank column = the fused
path's layout).  NB / N / T / XB_DECODE_LPS from the environment; XB_DECODE_STOP=1|2 (diagnostic library only)
stops after sweep 1 / 2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from xna_basecaller_amd import _lib
nb = int(os.environ.get("NB", 5)); N = int(os.environ.get("N", 512)); T = int(os.environ.get("T", 2000))
S = nb ** 3
ctx = _lib.Context(0, nb, 3, 32, 19, 5, 5.0, 2.0, T * 5, N)
g = torch.Generator(device="cuda"); g.manual_seed(1)
ld = (S * nb + 3) // 4 * 4 if os.environ.get("PAD", "0") == "1" else S * nb
sc = 5 * torch.tanh(torch.randn((T, N, S * nb), device="cuda", generator=g))
d_seq = torch.empty((N, T), dtype=torch.int8, device="cuda"); d_len = torch.empty((N,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
ctx.set_profiling(True)
best = 1e9
for rep in range(int(os.environ.get("REPS", 4))):
    ctx.reset_stage_times()
    ctx.decode_dev(sc.data_ptr(), T, N, False, "NACGTXY"[:nb + 1], None, d_seq.data_ptr(), d_len.data_ptr())
    ctx.synchronize()
    best = min(best, ctx.stage_times()["decode"][0])
a_dec = T * N * (3 * S * (nb + 1) * 4 + 7 * S * 4 + 1)
print("NB %d N %d T %d LPS %s stop %s: decode %.3f ms  -> %.0f GB/s algorithmic = %.3f of 8 TB/s" % (
    nb, N, T, os.environ.get("XB_DECODE_LPS", "auto"), os.environ.get("XB_DECODE_STOP", "0"), best,
    a_dec / best / 1e6, a_dec / best / 1e6 / 8000))
