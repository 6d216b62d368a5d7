"""GPU: the mixed-precision encoder (XB_PREC_MIXED and the XB_X3_STAGES stage mask).  An f16f8 context runs chosen
contraction stages in the three-product f16x3 arithmetic; the activation tensor between two stages then carries the second
part (q8 image or fp16 residual) that the CONSUMING stage reads, which for the recurrence means the YALT kernel variants
(lstm_kernel<.., YALT = true>).  Checked: the plumbing (every stage x3 == an f16x3 context, bit for bit), the residual /
q8 planes the YALT kernels write (against the exchange-form planes of the same hidden values), the score tolerance of the
mixes against the oracle, and the invariants every recurrence variant keeps (per-step == persistent, two groups == one)."""
import numpy as np
import pytest

import oracle
from conftest import encoder_shapes, seeded_state_dict
from xna_basecaller_amd import _lib

pytestmark = pytest.mark.gpu

IN, REC, LIN, CONV = (lambda l: 1 << l), (lambda l: 1 << (5 + l)), 1 << 10, 1 << 11
ALL = 0xfff
MIXED = CONV | LIN | sum(IN(l) for l in range(5))      # XB_PREC_MIXED: every feed-forward projection


def _ctx(F, nb, L, N, prec, mask=None, monkeypatch=None, lstm_mode=0):
    if mask is None:
        monkeypatch.delenv("XB_X3_STAGES", raising=False)
    else:
        monkeypatch.setenv("XB_X3_STAGES", str(mask))
    return _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=prec, lstm_mode=lstm_mode)


def _model(F, nb, L, N, seed):
    keys, shapes = encoder_shapes(F, nb)
    sd = seeded_state_dict(keys, shapes, seed=seed)
    x = np.random.default_rng(seed).standard_normal((N, L)).astype(np.float32)
    return sd, x


def _e4m3(b):
    """OCP e4m3 (fn) bytes -> float64."""
    b = b.astype(np.int64)
    s, e, m = (b >> 7) & 1, (b >> 3) & 15, b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (8 + m) * 2.0 ** (e - 10.0))
    return np.where(s == 1, -v, v)


@pytest.mark.parametrize("F,N,L", [(64, 5, 600), (128, 70, 400)])
def test_every_stage_x3_equals_an_f16x3_context(F, N, L, monkeypatch):
    sd, x = _model(F, 6, L, N, seed=F)
    a = _ctx(F, 6, L, N, _lib.XB_PREC_F16X3, None, monkeypatch)
    a.load_state_dict(sd)
    ref = a.encode(x)
    a.close()
    b = _ctx(F, 6, L, N, _lib.XB_PREC_F16F8, ALL, monkeypatch)
    b.load_state_dict(sd)
    got = b.encode(x)
    b.close()
    assert np.array_equal(ref, got)


def test_mixed_precision_is_the_documented_stage_mask(monkeypatch):
    F, N, L = 64, 6, 800
    sd, x = _model(F, 5, L, N, seed=3)
    a = _ctx(F, 5, L, N, _lib.XB_PREC_MIXED, None, monkeypatch)
    a.load_state_dict(sd)
    ra = a.encode(x)
    a.close()
    b = _ctx(F, 5, L, N, _lib.XB_PREC_F16F8, MIXED, monkeypatch)
    b.load_state_dict(sd)
    rb = b.encode(x)
    b.close()
    assert np.array_equal(ra, rb)
    ref = oracle.encode(x, sd, F, 5, 3)
    assert np.abs(ra - ref).max() < 1e-4


@pytest.mark.parametrize("F,N,L,lstm_mode", [(64, 5, 400, 2), (64, 5, 400, 1), (128, 130, 300, 2), (768, 70, 250, 2)])
def test_alternative_layer_output_planes(F, N, L, lstm_mode, monkeypatch):
    """Layer 4 in the q8 arithmetic: its output goes out as hi + q8 image (linear layer f16f8) or hi + fp16 residual (linear layer
    f16x3, the YALT kernel).  Same hidden values either way: hi planes bit-equal, residual == what the image's l8 byte encodes to
    e4m3 accuracy.  The same for layer 3 -> input projection of layer 4, and in the other direction (x3 recurrence, q8 output)."""
    sd, x = _model(F, 6, L, N, seed=11)
    planes = {}
    for tag, mask in (("q8", 0), ("res", LIN | IN(4)), ("x3_res", REC(3) | REC(4) | LIN | IN(4)), ("x3_q8", REC(3) | REC(4))):
        c = _ctx(F, 6, L, N, _lib.XB_PREC_F16F8, mask, monkeypatch, lstm_mode=lstm_mode)
        c.load_state_dict(sd)
        c.encode(x)
        planes[tag] = c.debug_layer_output(0, N)          # layer 3's output: its recurrence arithmetic is the same in "q8" / "res"
        c.close()
    for exch, alt in (("q8", "res"), ("x3_res", "x3_q8")):
        q8tag, restag = (exch, alt) if exch.endswith("q8") else (alt, exch)
        hi_q, img = planes[q8tag]
        hi_r, res = planes[restag]
        assert np.array_equal(hi_q, hi_r)                 # the same h, bit for bit, whichever second part travels with it
        hi = hi_r.view(np.float16).astype(np.float64)
        lo = res.view(np.float16).astype(np.float64)
        assert np.all(np.abs(lo) <= np.abs(hi) * 2.0 ** -10 + 2.0 ** -24)      # a residual of its hi part
        # q8 image: per row and 32 columns [32 x h8 | 32 x l8] in the place of the 32 fp16 residuals (exponent 8)
        b = img.view(np.uint8).reshape(img.shape[0], img.shape[1], F // 32, 64)
        h8 = _e4m3(b[..., :32]).reshape(hi.shape) * 2.0 ** -8
        l8 = _e4m3(b[..., 32:]).reshape(hi.shape) * 2.0 ** -19
        assert np.all(np.abs(h8 - hi) <= np.abs(hi) * 2.0 ** -4 + 2.0 ** -17)
        # (the residual plane is fp16: below 2^-14 it is a subnormal with spacing 2^-24, the l8 byte encodes the unrounded residual)
        assert np.all(np.abs(l8 - lo) <= np.abs(lo) * 2.0 ** -4 + 2.0 ** -23)
        assert np.abs(lo).max() > 0 and np.abs(l8).max() > 0


@pytest.mark.parametrize("mask", [MIXED, IN(4) | REC(4) | LIN, LIN | REC(4), IN(0) | IN(2) | IN(4) | REC(1) | REC(3), REC(0) | REC(2) | REC(4) | IN(1) | IN(3) | LIN,
                                  CONV | IN(0), ALL & ~REC(2)])
def test_stage_mixes_against_the_oracle(mask, monkeypatch):
    F, nb, L, N = 128, 6, 1000, 6
    sd, x = _model(F, nb, L, N, seed=mask)
    ref = oracle.encode(x, sd, F, nb, 3)
    plain = _ctx(F, nb, L, N, _lib.XB_PREC_F16F8, 0, monkeypatch)
    plain.load_state_dict(sd)
    e0 = np.abs(plain.encode(x) - ref).max()
    plain.close()
    outs = []
    for mode in (1, 2):
        c = _ctx(F, nb, L, N, _lib.XB_PREC_F16F8, mask, monkeypatch, lstm_mode=mode)
        c.load_state_dict(sd)
        outs.append(c.encode(x))
        c.close()
    assert np.array_equal(outs[0], outs[1])                # one launch per step == persistent, also for the YALT variants
    e = np.abs(outs[1] - ref).max()
    assert e < 2e-4 and e0 < 2e-4
    assert e <= 1.5 * e0 + 1e-5, (e, e0)                   # more exact stages never make it worse than plain f16f8


@pytest.mark.parametrize("F,N,L", [(64, 130, 300), (768, 130, 150)])
def test_two_groups_per_workgroup_with_mixed_stages(F, N, L, monkeypatch):
    sd, x = _model(F, 6, L, N, seed=5)
    outs = []
    for dual in ("0", "2"):
        monkeypatch.setenv("XB_LSTM_DUAL", dual)
        c = _ctx(F, 6, L, N, _lib.XB_PREC_F16F8, MIXED | REC(2), monkeypatch, lstm_mode=2)
        c.load_state_dict(sd)
        outs.append(c.encode(x))
        c.close()
    assert np.array_equal(outs[0], outs[1])
