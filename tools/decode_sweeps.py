#!/usr/bin/env python
"""Diagnostic: time of the CRF decode when stopped after sweep 1 / 2 / 3 (needs the diag library)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import random_scores
from xna_basecaller_amd import _lib
nb = int(os.environ.get("NB", 5)); N = int(os.environ.get("N", 512)); T = int(os.environ.get("T", 2000))
S = nb ** 3
ctx = _lib.Context(0, nb, 3, 32, 19, 5, 5.0, 2.0, T * 5, N)
g = torch.Generator(device="cuda"); g.manual_seed(1)
sc = 5 * torch.tanh(torch.randn((T, N, S * nb), device="cuda", generator=g))
d_seq = torch.empty((N, T), dtype=torch.int8, device="cuda"); d_len = torch.empty((N,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
ctx.set_profiling(True)
for rep in range(2):
    ctx.reset_stage_times()
    ctx.decode_dev(sc.data_ptr(), T, N, False, "NACGTXY"[:nb + 1], None, d_seq.data_ptr(), d_len.data_ptr())
    ctx.synchronize()
print("NB", nb, "stop", os.environ.get("XB_DECODE_STOP", "0"), "decode ms", ctx.stage_times()["decode"][0])
