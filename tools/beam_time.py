#!/usr/bin/env python
"""GPU tool: time xb_beam_search_dev (Log scans + beam kernel) on device-resident scores at a bench-sized batch.
  python tools/beam_time.py [N] [nb] [state_len]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xna_basecaller_amd import _lib                                          # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sl = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T = 2000
alphabet = "NACGTXY"[:nb + 1]
ctx = _lib.Context(0, nb, sl, 32, 19, 5, 5.0, 2.0, T * 5, N)
S = nb ** sl
g = torch.Generator(device="cuda").manual_seed(1)
sc = 5.0 * torch.tanh(torch.randn((T, N, S * nb), device="cuda", generator=g))
d_seq = torch.zeros((N, T), dtype=torch.int8, device="cuda")
d_q = torch.zeros((N, T), dtype=torch.int8, device="cuda")
d_mv = torch.zeros((N, T), dtype=torch.uint8, device="cuda")
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.beam_search_dev(sc.data_ptr(), T, N, 0, alphabet, d_seq.data_ptr(), d_q.data_ptr(), d_mv.data_ptr())
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("N %d nb %d sl %d: scans + beam search %.2f ms  (%.3e samples/s, %.2f bases per block)"
          % (N, nb, sl, 1e3 * dt, N * T * 5 / dt, float(d_mv.sum()) / (N * T)), flush=True)
ctx.close()
