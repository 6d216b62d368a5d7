"""GPU: the HIP encoder (conv front-end, split-fp16 MFMA GEMMs, LSTM) through the C ABI against the golden
fixtures made by the reference's nn.py modules and against the oracle.  Tolerance: |score error| <= 1e-3
(north star); the default split-fp16 (3 product) arithmetic lands around 1e-5."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, encoder_shapes, seeded_state_dict
from xna_basecaller_amd import _lib

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _golden_case(name):
    meta = json.load(open(os.path.join(GOLDEN, "encoder_meta.json")))
    case = [c for c in meta["cases"] if c["name"] == name][0]
    z = np.load(os.path.join(GOLDEN, "encoder_small.npz"))
    sd = {k: z["%s/w/%s" % (name, k)] for k in case["keys"]}
    return case, sd, z[name + "/signal"], z[name + "/scores"]


@pytest.mark.parametrize("name", ["f32_nb6", "f32_nb4_long"])
@pytest.mark.parametrize("lstm_mode", [1, 2])
def test_encoder_small_golden_f16f8(name, lstm_mode):
    """The default arithmetic (f16f8) against the fixtures made by the reference's own modules."""
    case, sd, signal, ref = _golden_case(name)
    F, nb = case["features"], len(case["labels"]) - 1
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, case["L"], case["N"], precision=_lib.XB_PREC_F16F8,
                       lstm_mode=lstm_mode)
    ctx.load_state_dict(sd)
    got = ctx.encode(signal[:, 0, :], expand_blanks=True)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err < 2e-4, err
    ctx.close()


@pytest.mark.parametrize("name", ["f32_nb6", "f32_nb4_long", "f48_nb5", "f16_nb4"])
@pytest.mark.parametrize("lstm_mode", [1, 2])
def test_encoder_small_golden(name, lstm_mode):
    case, sd, signal, ref = _golden_case(name)
    F, nb = case["features"], len(case["labels"]) - 1
    if F % 32:
        with pytest.raises(_lib.XbError) as e:
            _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, case["L"], case["N"])
        assert "features" in str(e.value)
        return
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, case["L"], case["N"], lstm_mode=lstm_mode)
    ctx.load_state_dict(sd)
    got = ctx.encode(signal[:, 0, :], expand_blanks=True)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err < TOL, err
    assert err < 1e-4, "split-fp16 arithmetic should be ~1e-5, got %g" % err
    S = nb ** 3
    assert np.all(got.reshape(got.shape[0], got.shape[1], S, nb + 1)[..., 0] == 2.0)
    nob = ctx.encode(signal[:, 0, :], expand_blanks=False)
    assert np.array_equal(nob.reshape(nob.shape[0], nob.shape[1], S, nb),
                          got.reshape(got.shape[0], got.shape[1], S, nb + 1)[..., 1:])
    ctx.close()


# (features 32 / 64 / 96: ONE, two and three k-tiles per GEMM workgroup -- the shortest trips through the k loop's prologue, its barrier between
#  MFMA phases 2 and 3 and its past-the-end requests, csrc/xb_encoder.hip XB_GEMM_XTILE / XB_GEMM_TAIL)
@pytest.mark.parametrize("features,nb,L,N", [(32, 6, 500, 5), (64, 5, 600, 3), (96, 4, 800, 70), (128, 6, 400, 2)])
def test_encoder_vs_oracle(features, nb, L, N):
    keys, shapes = encoder_shapes(features, nb)
    sd = seeded_state_dict(keys, shapes, seed=features + nb)
    x = np.random.default_rng(L).standard_normal((N, L)).astype(np.float32)
    ref = oracle.encode(x, sd, features, nb, 3)
    for mode in (1, 2):
        ctx = _lib.Context(0, nb, 3, features, 19, 5, 5.0, 2.0, L, N, lstm_mode=mode)
        ctx.load_state_dict(sd)
        got = ctx.encode(x)
        assert np.abs(got - ref).max() < 1e-4
        ctx.close()


def test_encoder_per_step_and_persistent_agree_bitwise():
    keys, shapes = encoder_shapes(96, 6)
    sd = seeded_state_dict(keys, shapes, seed=5)
    x = np.random.default_rng(3).standard_normal((130, 1000)).astype(np.float32)   # 3 groups, last one ragged
    outs = []
    for mode in (1, 2):
        ctx = _lib.Context(0, 6, 3, 96, 19, 5, 5.0, 2.0, 1000, 130, lstm_mode=mode)
        ctx.load_state_dict(sd)
        outs.append(ctx.encode(x))
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


def test_encoder_full_size_golden():
    """features 768 (24.85 M parameters), weights regenerated from the fixture's seed."""
    meta = json.load(open(os.path.join(GOLDEN, "encoder_meta.json")))["full"]
    z = np.load(os.path.join(GOLDEN, "encoder_full.npz"))
    sd = seeded_state_dict(meta["keys"], meta["shapes"], meta["seed"])
    x = np.random.default_rng(meta["signal_seed"]).standard_normal((meta["N"], meta["L"])).astype(np.float32)
    for mode in (1, 2):
        ctx = _lib.Context(0, 6, 3, 768, 19, 5, 5.0, 2.0, meta["L"], meta["N"], lstm_mode=mode)
        ctx.load_state_dict(sd)
        got = ctx.encode(x)
        err = max(np.abs(got[0] - z["scores_t0"]).max(), np.abs(got[-1] - z["scores_tlast"]).max(),
                  np.abs(got[50, :, ::7] - z["scores_mid"]).max())
        assert err < 1e-4, err
        ctx.close()


def test_encoder_fp16_fast_mode_is_close():
    """precision = f16 (single product, the reference's model.half()): looser than the default, reported only."""
    case, sd, signal, ref = _golden_case("f32_nb6")
    ctx = _lib.Context(0, 6, 3, 32, 19, 5, 5.0, 2.0, case["L"], case["N"], precision=_lib.XB_PREC_F16)
    ctx.load_state_dict(sd)
    got = ctx.encode(signal[:, 0, :])
    assert np.abs(got - ref).max() < 2e-2
    ctx.close()


def test_persistent_handoff_is_placement_independent(monkeypatch):
    """Full-size W (features 768, 24 member workgroups per group), several groups, ragged last group:
    the persistent kernel must equal the one-launch-per-step kernel bit for bit, both with a group's members
    on one XCD (default block mapping) and spread over all XCDs (XB_LSTM_SPREAD=1)."""
    meta = json.load(open(os.path.join(GOLDEN, "encoder_meta.json")))["full"]
    sd = seeded_state_dict(meta["keys"], meta["shapes"], meta["seed"])
    N, L = 150, 400
    x = np.random.default_rng(11).standard_normal((N, L)).astype(np.float32)
    outs = {}
    for name, mode, spread in [("step", 1, "0"), ("persist", 2, "0"), ("persist_spread", 2, "1")]:
        monkeypatch.setenv("XB_LSTM_SPREAD", spread)
        ctx = _lib.Context(0, 6, 3, 768, 19, 5, 5.0, 2.0, L, N, lstm_mode=mode)
        ctx.load_state_dict(sd)
        for rep in range(3 if mode == 2 else 1):           # repeated launches reuse warm caches
            outs[name] = ctx.encode(x)
        ctx.close()
    assert np.array_equal(outs["step"], outs["persist"])
    assert np.array_equal(outs["step"], outs["persist_spread"])
    ref = oracle.encode(x[:4], sd, 768, 6, 3)
    assert np.abs(outs["persist"][:, :4] - ref).max() < 1e-4


@pytest.mark.parametrize("features,nb,L,N", [(32, 6, 500, 5), (64, 5, 600, 3), (96, 4, 800, 70), (128, 6, 400, 2)])
def test_encoder_f16f8_vs_oracle(features, nb, L, N):
    """precision = f16f8: fp16 main product, both correction products on the block-scaled FP8 MFMA.
    Numerical model (tools/../DESIGN.md): |score error| ~4e-5 at features 768, an order below the north-star 1e-3."""
    keys, shapes = encoder_shapes(features, nb)
    sd = seeded_state_dict(keys, shapes, seed=features + nb)
    x = np.random.default_rng(L).standard_normal((N, L)).astype(np.float32)
    ref = oracle.encode(x, sd, features, nb, 3)
    outs = []
    for mode in (1, 2):
        ctx = _lib.Context(0, nb, 3, features, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8, lstm_mode=mode)
        ctx.load_state_dict(sd)
        got = ctx.encode(x)
        err = np.abs(got - ref).max()
        assert err < 2e-4, err
        outs.append(got)
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


def test_encoder_f16f8_full_size():
    """features 768, several groups with a ragged tail: persistent == per-step bit for bit, oracle within 2e-4,
    and large activations (conv output beyond the e4m3 range of the q8 image) only lose the correction, not the value."""
    meta = json.load(open(os.path.join(GOLDEN, "encoder_meta.json")))["full"]
    sd = seeded_state_dict(meta["keys"], meta["shapes"], meta["seed"])
    N, L = 150, 400
    x = np.random.default_rng(11).standard_normal((N, L)).astype(np.float32)
    x[3] *= 40.0                                            # a badly normalised chunk
    outs = {}
    for name, mode in [("step", 1), ("persist", 2)]:
        ctx = _lib.Context(0, 6, 3, 768, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8, lstm_mode=mode)
        ctx.load_state_dict(sd)
        outs[name] = ctx.encode(x)
        ctx.close()
    assert np.array_equal(outs["step"], outs["persist"])
    ref = oracle.encode(x[:4], sd, 768, 6, 3)
    err = np.abs(outs["persist"][:, :4] - ref)
    assert err[:, :3].max() < 2e-4, err[:, :3].max()
    assert err[:, 3].max() < 1e-3, err[:, 3].max()


def test_encoder_more_groups_than_one_launch_holds():
    """features 768, 600 chunks = 10 groups: a persistent launch holds 8 (one group of 24 workgroups per XCD), so the batch
    runs as two chunk slabs (512 + a ragged 88) x two time slabs (T = 260), with every group keeping its own exchange and
    counter slot across the launches.  Must equal the one-launch-per-step mode bit for bit, and the oracle on a sample."""
    meta = json.load(open(os.path.join(GOLDEN, "encoder_meta.json")))["full"]
    sd = seeded_state_dict(meta["keys"], meta["shapes"], meta["seed"])
    N, L = 600, 1300
    x = np.random.default_rng(12).standard_normal((N, L)).astype(np.float32)
    outs = {}
    for name, mode in [("step", 1), ("persist", 2)]:
        ctx = _lib.Context(0, 6, 3, 768, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8, lstm_mode=mode)
        ctx.load_state_dict(sd)
        outs[name] = ctx.encode(x, expand_blanks=False)
        ctx.close()
    assert np.array_equal(outs["step"], outs["persist"])
    pick = [0, 511, 512, 599]
    ref = oracle.encode(x[pick], sd, 768, 6, 3, expand_blanks=False)
    assert np.abs(outs["persist"][:, pick] - ref).max() < 2e-4


@pytest.mark.parametrize("features,N", [(512, 1100), (256, 2100), (384, 1030)])
def test_encoder_group_limit_other_feature_sizes(features, N):
    """16 / 8 / 12 member workgroups per group -> 2 / 4 / 2 groups per XCD and launch: batches just past one launch's
    capacity (1024 / 2048 / 1024 chunks), ragged last group; persistent == one launch per step, bit for bit."""
    nb, L = 4, 300
    keys, shapes = encoder_shapes(features, nb)
    sd = seeded_state_dict(keys, shapes, seed=features)
    x = np.random.default_rng(features).standard_normal((N, L)).astype(np.float32)
    outs = []
    for mode in (1, 2):
        ctx = _lib.Context(0, nb, 3, features, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8, lstm_mode=mode)
        ctx.load_state_dict(sd)
        outs.append(ctx.encode(x, expand_blanks=False))
        ctx.close()
    assert np.array_equal(outs[0], outs[1])
    pick = [0, N // 2, N - 1]
    ref = oracle.encode(x[pick], sd, features, nb, 3, expand_blanks=False)
    assert np.abs(outs[1][:, pick] - ref).max() < 2e-4


@pytest.mark.parametrize("features,nb,L,N", [(64, 5, 5, 3), (64, 5, 10, 70), (768, 6, 15, 65), (128, 4, 23, 1),
                                             (768, 5, 640, 513)])
def test_encoder_edge_shapes(features, nb, L, N):
    """T = 1, 2, 3 (only the first-step path of the recurrence, the one-step-ahead gin prefetch with nothing to prefetch),
    L not a multiple of the stride, a single chunk, and one chunk more than a launch holds at features 768."""
    keys, shapes = encoder_shapes(features, nb)
    sd = seeded_state_dict(keys, shapes, seed=features + L)
    x = np.random.default_rng(L).standard_normal((N, L)).astype(np.float32)
    outs = []
    for mode in (1, 2):
        ctx = _lib.Context(0, nb, 3, features, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8, lstm_mode=mode)
        ctx.load_state_dict(sd)
        outs.append(ctx.encode(x, expand_blanks=False))
        if mode == 2:
            seq, lens = ctx.basecall_chunks(x, "NACGTXY"[:nb + 1])
            lab = oracle.decode(ctx.encode(x, expand_blanks=True), nb, 3)["labels"]
            assert np.array_equal(lens, (lab != 0).sum(axis=1))
        ctx.close()
    assert np.array_equal(outs[0], outs[1])
    pick = sorted({0, N // 2, N - 1})
    ref = oracle.encode(x[pick], sd, features, nb, 3, expand_blanks=False)
    assert np.abs(outs[1][:, pick] - ref).max() < 2e-4


@pytest.mark.parametrize("N", [384, 128, 640])
def test_encoder_batches_that_are_multiples_of_128(N):
    """The reference's shipped batch size is 384 (config.toml:26-29): a multiple of 128 but not of the GEMM's 256-row tile.
    Those batches take the unchecked member-major gin epilogue per WAVE (128 rows); every chunk must still agree with
    the per-step launch mode bit for bit and with the oracle."""
    F, nb, L = 64, 6, 720 * 5 // 6           # T = 120
    keys, shapes = encoder_shapes(F, nb)
    sd = seeded_state_dict(keys, shapes, seed=7)
    x = np.random.default_rng(N).standard_normal((N, L)).astype(np.float32)
    out = []
    for mode in (2, 1):
        ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8, lstm_mode=mode)
        ctx.load_state_dict(sd)
        out.append(ctx.encode(x, expand_blanks=False))
        ctx.close()
    assert np.array_equal(out[0], out[1])
    picks = sorted({0, 1, 127, min(128, N - 1), N // 2, max(N - 129, 0), N - 128, N - 1})
    ref = oracle.encode(x[picks], sd, F, nb, 3, expand_blanks=False)
    assert np.abs(ref - out[0][:, picks]).max() < 2e-4


def test_encoder_f16f8_in1_mode_within_north_star_tolerance():
    """XB_PREC_F16F8_IN1: the LSTM input projections keep only the fp16 main product (they have no feedback through time),
    recurrence / conv / linear stay f16f8.  Still inside the 1e-3 north-star tolerance, and the launch modes agree bitwise."""
    F, nb, L, N = 768, 6, 2500, 4
    keys, shapes = encoder_shapes(F, nb)
    sd = seeded_state_dict(keys, shapes, seed=25)
    x = np.random.default_rng(3).standard_normal((N, L)).astype(np.float32)
    ref = oracle.encode(x, sd, F, nb, 3, expand_blanks=False)
    out = []
    for mode in (2, 1):
        ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8_IN1, lstm_mode=mode)
        ctx.load_state_dict(sd)
        out.append(ctx.encode(x, expand_blanks=False))
        ctx.close()
    assert np.array_equal(out[0], out[1])
    err = np.abs(out[0] - ref)
    assert err.max() < TOL and np.sqrt((err ** 2).mean()) < 2e-4, (err.max(), np.sqrt((err ** 2).mean()))


@pytest.mark.parametrize("features,N,L,prec", [
    (768, 1024, 500, "f16f8"),    # 16 groups: every workgroup serves two (the configs[3] per-GPU launch shape)
    (768, 600, 300, "f16f8"),     # 10 groups: slots 0-4 serve two groups, ragged last group, spread placement too
    (768, 150, 400, "f16x3"),     # 3 groups, forced: slot 1 has a single group (blocking path beside the early one)
    (256, 200, 400, "f16f8"),     # two pieces per step (first piece requested inside the last piece)
    (384, 200, 400, "f16f8"),     # three pieces: odd count, first piece requested after the last piece instead
    (128, 330, 400, "f16x3"),     # one piece per step
    (32, 700, 300, "f16f8"),      # one member per group, 11 groups
])
def test_two_groups_per_workgroup_equals_one_launch_per_step(monkeypatch, features, N, L, prec):
    """lstm_kernel<.., DUAL>: a workgroup serves two chunk groups alternately (XB_LSTM_DUAL: 1 = when the batch needs more
    groups than one launch holds, 2 = whenever there are two groups).  Same arithmetic, different schedule and hand-off
    timing: scores must equal the single-group persistent kernel and the one-launch-per-step kernel bit for bit, for both
    block placements, on repeated calls (warm caches, counters reused)."""
    nb = 6
    keys, shapes = encoder_shapes(features, nb)
    sd = seeded_state_dict(keys, shapes, seed=features + N)
    x = np.random.default_rng(N).standard_normal((N, L)).astype(np.float32)
    precision = {"f16f8": _lib.XB_PREC_F16F8, "f16x3": _lib.XB_PREC_F16X3}[prec]
    outs = {}
    for name, mode, dual, spread in [("step", 1, "0", "0"), ("single", 2, "0", "0"), ("dual", 2, "2", "0"),
                                     ("dual_spread", 2, "2", "1"), ("dual_step", 1, "2", "0")]:
        monkeypatch.setenv("XB_LSTM_DUAL", dual)
        monkeypatch.setenv("XB_LSTM_SPREAD", spread)
        ctx = _lib.Context(0, nb, 3, features, 19, 5, 5.0, 2.0, L, N, precision=precision, lstm_mode=mode)
        ctx.load_state_dict(sd)
        for rep in range(2 if mode == 2 else 1):
            outs[name] = ctx.encode(x)
        ctx.close()
    for name in ("single", "dual", "dual_spread", "dual_step"):
        assert np.array_equal(outs["step"], outs[name]), name
    picks = sorted({0, 63, 64, N // 2, N - 1})
    ref = oracle.encode(x[picks], sd, features, nb, 3)
    assert np.abs(outs["dual"][:, picks] - ref).max() < 2e-4


@pytest.mark.parametrize("features,nb,L,N,prec", [(32, 4, 300, 5, _lib.XB_PREC_F16F8), (96, 5, 400, 9, _lib.XB_PREC_F16),
                                                    (96, 4, 800, 70, _lib.XB_PREC_F16X3), (256, 6, 1000, 130, _lib.XB_PREC_F16F8),
                                                    (768, 6, 500, 200, _lib.XB_PREC_F16F8_IN1)])
def test_gemm_kernels_agree_bitwise(features, nb, L, N, prec, monkeypatch):
    """gemm4p_kernel (two workgroups per CU, LDS-DMA ring for A, B from the fragment-major image with hand-counted waits)
    and gemm8r_kernel (one workgroup per CU, both operands through LDS, compiler-counted) add the same products in the
    same order per accumulator: identical scores, bit for bit, in every precision mode and with ragged M / N / K edges."""
    keys, shapes = encoder_shapes(features, nb)
    sd = seeded_state_dict(keys, shapes, seed=features + nb)
    x = np.random.default_rng(L + N).standard_normal((N, L)).astype(np.float32)
    out = []
    for g4 in ("0", "1"):
        monkeypatch.setenv("XB_GEMM4", g4)
        ctx = _lib.Context(0, nb, 3, features, 19, 5, 5.0, 2.0, L, N, precision=prec)
        ctx.load_state_dict(sd)
        out.append(ctx.encode(x))
        ctx.close()
    assert np.isfinite(out[1]).all()
    assert np.array_equal(out[0], out[1])


@pytest.mark.parametrize("N", [98, 448, 513])
def test_batch_not_a_multiple_of_128(N, monkeypatch):
    """VERDICT r2 (weak 7): the member-major gin epilogue used to fall off its unchecked path for any batch that is not a
    multiple of 128 (the trailing batch of every run; the reference's own eval_model.sh -b 98).  Features 768: scores
    against the oracle, bit-equality between the two GEMM kernels, and the recurrence reading what the GEMM wrote."""
    F, nb, L = 768, 6, 250
    keys, shapes = encoder_shapes(F, nb)
    sd = seeded_state_dict(keys, shapes, seed=N)
    x = np.random.default_rng(N).standard_normal((N, L)).astype(np.float32)
    ref = oracle.encode(x, sd, F, nb, 3)
    got = []
    for g4 in ("1", "0"):
        monkeypatch.setenv("XB_GEMM4", g4)
        ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8)
        ctx.load_state_dict(sd)
        got.append(ctx.encode(x))
        ctx.close()
    assert np.abs(got[0] - ref).max() < 2e-4
    assert np.array_equal(got[0], got[1])


def test_reload_weights_on_live_context():
    """A second load_state_dict on a live context replaces the device weight set (and frees the old one)."""
    F, nb, L, N = 64, 5, 600, 3
    keys, shapes = encoder_shapes(F, nb)
    x = np.random.default_rng(1).standard_normal((N, L)).astype(np.float32)
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N)
    for seed in (3, 4, 3):
        sd = seeded_state_dict(keys, shapes, seed=seed)
        ctx.load_state_dict(sd)
        assert np.abs(ctx.encode(x) - oracle.encode(x, sd, F, nb, 3)).max() < 1e-4
    ctx.close()


@pytest.mark.parametrize("N,pair", [(513, False), (576, True), (640, False), (1025, False), (1280, False)])
def test_wide_group_slots_equal_the_xcd_local_placement(N, pair, monkeypatch):
    """Round 5, the batch cliffs: 513..640 chunks run as ONE launch with one group per workgroup over up to ten group slots
    whose members are dealt over all XCDs (1025..1280: two groups per workgroup), instead of costing a second round; batches
    of up to 640 chunks pair (two calls share a pass).  Placement and slot count change nothing: scores and called sequences
    equal XB_LSTM_WIDE=0 byte for byte."""
    import torch
    F_, nb, L = 768, 5, 600
    keys, shapes = encoder_shapes(F_, nb)
    sd = seeded_state_dict(keys, shapes, seed=11)
    gen = torch.Generator(device="cuda").manual_seed(N)
    x = torch.randn((N, L), dtype=torch.float32, device="cuda", generator=gen)

    def run(wide):
        monkeypatch.setenv("XB_LSTM_WIDE", "1" if wide else "0")
        ctx = _lib.Context(0, nb, 3, F_, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_MIXED)
        monkeypatch.delenv("XB_LSTM_WIDE")
        ctx.load_state_dict(sd)
        paired = bool(pair and ctx.reserve_pairing())
        bufs = [(torch.empty((N, ctx.T), dtype=torch.int8, device="cuda"), torch.empty((N,), dtype=torch.int32, device="cuda"))
                for _ in range(2)]
        for s, l in bufs:
            ctx.basecall_chunks_dev(x.data_ptr(), N, "NACGTX", s.data_ptr(), l.data_ptr())
        ctx.synchronize()
        out = [(s.cpu().numpy(), l.cpu().numpy()) for s, l in bufs]
        scores = ctx.encode(x.cpu().numpy())
        ctx.close()
        return out, scores, paired

    wide, sw, paired_w = run(True)
    local, sl, paired_l = run(False)
    if pair:
        assert paired_w and not paired_l          # 576 > 512: only the wide placement can co-schedule two such calls
    assert np.array_equal(sw.view(np.uint32), sl.view(np.uint32))
    for (a, la), (b, lb) in zip(wide, local):
        assert np.array_equal(a, b) and np.array_equal(la, lb)
    assert np.array_equal(wide[0][0], wide[1][0])
