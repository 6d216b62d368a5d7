"""GPU: the int8-limb recurrence (XB_LSTM_I8=1: W_hh rows and h as 16-bit fixed point in two balanced signed digits, the
four digit products on v_mfma_i32_32x32x32_i8, exact int32 sums) against the fp32 oracle encoder, and its launch variants
against each other (same arithmetic -> same bits)."""
import numpy as np
import pytest

import oracle
from xna_basecaller_amd import _lib
from xna_basecaller_amd.synthetic import encoder_shapes, seeded_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("features,N,L", [(768, 150, 1000), (256, 200, 600), (128, 70, 500), (64, 130, 400), (384, 70, 400)])
def test_int8_limb_recurrence(monkeypatch, features, N, L):
    nb = 6
    keys, shapes = encoder_shapes(features, nb)
    sd = seeded_state_dict(keys, shapes, seed=features + 1)
    x = np.random.default_rng(features).standard_normal((N, L)).astype(np.float32)
    picks = sorted({0, 1, 63, 64, N - 1})
    ref = oracle.encode(x[picks], sd, features, nb, 3, expand_blanks=False)
    outs = {}
    for name, i8, mode, dual in [("f16f8", "0", 2, "0"), ("i8", "1", 2, "0"), ("i8_step", "1", 1, "0"), ("i8_dual", "1", 2, "2"),
                                 ("i8x3", "2", 2, "0"), ("i8x3_dual_step", "2", 1, "2")]:
        monkeypatch.setenv("XB_LSTM_I8", i8)
        monkeypatch.setenv("XB_LSTM_DUAL", dual)
        ctx = _lib.Context(0, nb, 3, features, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8, lstm_mode=mode)
        ctx.load_state_dict(sd)
        for rep in range(2 if mode == 2 else 1):
            outs[name] = ctx.encode(x, expand_blanks=False)
        ctx.close()
    e0 = np.abs(outs["f16f8"][:, picks] - ref)
    e1 = np.abs(outs["i8"][:, picks] - ref)
    print("features %d: f16f8 max %.2e rms %.2e | int8 limbs max %.2e rms %.2e" % (
        features, e0.max(), np.sqrt((e0 ** 2).mean()), e1.max(), np.sqrt((e1 ** 2).mean())))
    assert not np.array_equal(outs["i8"], outs["f16f8"])            # the switch took effect
    assert e1.max() < 2e-4, e1.max()
    assert np.array_equal(outs["i8"], outs["i8_step"])
    assert np.array_equal(outs["i8"], outs["i8_dual"])
    e2 = np.abs(outs["i8x3"][:, picks] - ref)
    print("features %d: three products (XB_LSTM_I8=2) max %.2e rms %.2e" % (features, e2.max(), np.sqrt((e2 ** 2).mean())))
    assert e2.max() < 3e-4, e2.max()
    assert np.array_equal(outs["i8x3"], outs["i8x3_dual_step"]) and not np.array_equal(outs["i8x3"], outs["i8"])
