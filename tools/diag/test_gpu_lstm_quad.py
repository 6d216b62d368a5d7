"""GPU: the software-pipelined recurrence (csrc/xb_lstm_quad.h: four groups of 32 chunks per workgroup, the gate math of one
group-step between the MFMAs of the next) against lstm_kernel<48, 2, DUAL> (XB_LSTM_QUAD=0), which the oracle parity tests of
the other modules pin.  Same products in the same order, same gate arithmetic: every output must be BIT-identical -- scores,
every layer's output planes, the called sequences -- for full and ragged group counts, several chunk slabs, time-slab
launches, both second-part forms of the layer output (YALT) and both member placements."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))

from conftest import encoder_shapes, seeded_state_dict
from xna_basecaller_amd import _lib

pytestmark = pytest.mark.gpu

# Round 5: the kernel lives in the diagnostic library only (make -C xna_basecaller_amd/csrc diag; run this file with
# XNA_LIBXNACALL=xna_basecaller_amd/libxnacall_diag.so python -m pytest tools/diag/test_gpu_lstm_quad.py); libxnacall.so
# refuses XB_LSTM_QUAD.
if "diag" not in os.environ.get("XNA_LIBXNACALL", ""):
    pytest.skip("needs the diagnostic library (XNA_LIBXNACALL=.../libxnacall_diag.so)", allow_module_level=True)

F, NB = 768, 5


def _run(monkeypatch, quad, n, L, prec, seed=5, layers=(), env=()):
    monkeypatch.setenv("XB_LSTM_QUAD", "1" if quad else "0")
    for k, v in env:
        monkeypatch.setenv(k, v)
    keys, shapes = encoder_shapes(F, NB)
    sd = seeded_state_dict(keys, shapes, seed=seed)
    x = np.random.default_rng(seed).standard_normal((n, L)).astype(np.float32)
    ctx = _lib.Context(0, NB, 3, F, 19, 5, 5.0, 2.0, L, n, precision=prec)
    ctx.load_state_dict(sd)
    out = {"scores": ctx.encode(x)}
    for l in layers:                 # 0 / 1: the output planes of LSTM layers 3 / 4 (hi, second part)
        out["hi%d" % l], out["lo%d" % l] = ctx.debug_layer_output(l, n)
    out["seq"], out["len"] = ctx.basecall_chunks(x, "NACGTX")
    ctx.close()
    for k, _ in env:
        monkeypatch.delenv(k, raising=False)
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k


# n: 1024 = 32 full groups of 32; 600 = 19 groups (workgroup slots with 4 and 3 groups, a ragged last group of 24 chunks);
# 516 = 17 groups, the last with 4 chunks; 1500 = a pipelined slab of 1024 + a tail slab of 476 on lstm_kernel (counter and
# exchange slots of both layouts side by side)
@pytest.mark.parametrize("n,L,prec", [(1024, 600, "mixed"), (1024, 1300, "mixed"), (600, 500, "mixed"), (516, 400, "f16f8"), (1500, 300, "mixed")])
def test_quad_kernel_bit_identical_to_dual(n, L, prec, monkeypatch):
    p = _lib.PRECISIONS[prec]
    a = _run(monkeypatch, False, n, L, p, layers=(0, 1))
    b = _run(monkeypatch, True, n, L, p, layers=(0, 1))
    _same(a, b)


def test_quad_kernel_members_on_several_xcds(monkeypatch):
    """XB_LSTM_SPREAD=1 deals a group's members over all XCDs: the one-XCD proof fails and the exchange stays write-through."""
    a = _run(monkeypatch, False, 640, 300, _lib.PRECISIONS["mixed"])
    b = _run(monkeypatch, True, 640, 300, _lib.PRECISIONS["mixed"], env=(("XB_LSTM_SPREAD", "1"),))
    _same(a, b)


def test_quad_kernel_time_slab_launches(monkeypatch):
    """XB_LSTM_SIGNAL=0: the recurrence runs as one launch per time slab (s_begin > 0, counters carried by sync_base)."""
    a = _run(monkeypatch, False, 1024, 1300, _lib.PRECISIONS["mixed"])
    b = _run(monkeypatch, True, 1024, 1300, _lib.PRECISIONS["mixed"], env=(("XB_LSTM_SIGNAL", "0"),))
    c = _run(monkeypatch, True, 1024, 1300, _lib.PRECISIONS["mixed"], env=(("XB_OVERLAP", "0"),))
    _same(a, b)
    _same(a, c)
