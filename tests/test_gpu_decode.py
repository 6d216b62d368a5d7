"""GPU: the HIP CRF decode (through the C ABI) against the oracle -- bit-exact labels / packed sequences."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, random_scores
from xna_basecaller_amd import _lib

pytestmark = pytest.mark.gpu


def _ctx(nb, T, N, sl=3, features=32):
    return _lib.Context(0, nb, sl, features, 19, 5, 5.0, 2.0, T * 5, N)


def _check(ctx, sc, nb, alphabet, sl=3, blank=None):
    seq, lens, labels = ctx.decode(sc, alphabet, want_labels=True)
    ref = oracle.decode(sc, nb, sl, blank_score=blank)["labels"]
    assert np.array_equal(labels, ref), "label mismatches: %d of %d at (chunk, t) %s" % (
        (labels != ref).sum(), ref.size, np.argwhere(labels != ref)[:8].tolist())
    rseq, _, rlens = oracle.pack(ref, alphabet)
    assert np.array_equal(lens, rlens)
    assert np.array_equal(seq, rseq)


@pytest.mark.parametrize("nb", [4, 5, 6])
@pytest.mark.parametrize("with_blank", [True, False])
@pytest.mark.parametrize("lps", [0, 1, 2])
def test_decode_bit_exact_random(nb, with_blank, lps, monkeypatch):
    """lps = lanes per CRF state (0: the library's own choice); every variant must give the same bits."""
    if lps:
        monkeypatch.setenv("XB_DECODE_LPS", str(lps))
    alphabet = "NACGTXY"[:nb + 1]
    T, N = 203, 5                                   # T not a multiple of the prefetch depth
    ctx = _ctx(nb, T, N)
    sc = random_scores(T, N, nb, seed=10 + nb, with_blank=with_blank)
    _check(ctx, sc, nb, alphabet, blank=None if with_blank else 2.0)
    ctx.close()


def test_decode_golden_fixture_scores():
    meta = json.load(open(os.path.join(GOLDEN, "decode_meta.json")))
    z = np.load(os.path.join(GOLDEN, "decode_small.npz"))
    for case in meta["cases"]:
        sc = z[case["name"] + "/scores_f16"].astype(np.float32)
        T, N, _ = sc.shape
        ctx = _ctx(case["nb"], T, N)
        _check(ctx, sc, case["nb"], "".join(case["labels"]))
        ctx.close()


@pytest.mark.parametrize("T", [1, 2, 3, 5, 8])
def test_decode_tiny_T(T):
    ctx = _ctx(6, 8, 3)
    _check(ctx, random_scores(T, 3, 6, seed=T), 6, "NACGTXY")
    ctx.close()


def test_decode_ties_and_extremes():
    nb, T, N = 6, 50, 4
    S, E = nb ** 3, nb + 1
    ctx = _ctx(nb, T, N)
    sc = np.zeros((T, N, S * E), np.float32)                       # every path ties: lowest flat index wins
    _check(ctx, sc, nb, "NACGTXY")
    sc = random_scores(T, N, nb, seed=1)
    sc[:, 1] = np.round(sc[:, 1])                                   # heavy ties on a coarse grid
    sc[:, 2] *= 8.0                                                 # scores far outside tanh range: deep underflow
    sc[:, 3] = -5.0
    sc[:, 3].reshape(T, S, E)[:, :, 0] = 5.0                        # blank dominates: empty call
    _check(ctx, sc, nb, "NACGTXY")
    seq, lens = ctx.decode(sc, "NACGTXY")
    assert lens[3] == 0 and not seq[3].any()
    ctx.close()


def test_decode_full_length_chunks():
    """BASELINE size T=2000 (chunksize 10000), 6-base CRF: oracle on all chunks of a small batch."""
    nb, T, N = 6, 2000, 6
    ctx = _ctx(nb, T, N)
    sc = random_scores(T, N, nb, seed=99)
    _check(ctx, sc, nb, "NACGTXY")
    ctx.close()


@pytest.mark.parametrize("lps", [1, 2])
def test_decode_full_length_odd_stride(lps, monkeypatch):
    """T = 2000 with the 5-base CRF and no blank column: row stride 625 floats (4-byte loads, deepest register
    ring) -- the configuration that exposed reuse of a prefetch register behind an unconsumed load."""
    monkeypatch.setenv("XB_DECODE_LPS", str(lps))
    nb, T, N = 5, 2000, 4
    ctx = _ctx(nb, T, N)
    sc = random_scores(T, N, nb, seed=123, with_blank=False)
    _check(ctx, sc, nb, "NACGTX", blank=2.0)
    ctx.close()


def test_decode_properties_large_batch():
    """Size-independent properties at a larger batch: per-chunk independence and batch-slicing invariance."""
    nb, T, N = 5, 400, 96
    ctx = _ctx(nb, T, N)
    sc = random_scores(T, N, nb, seed=5)
    seq, lens, labels = ctx.decode(sc, "NACGTX", want_labels=True)
    sub = np.ascontiguousarray(sc[:, 17:49])
    seq2, lens2, labels2 = ctx.decode(sub, "NACGTX", want_labels=True)
    assert np.array_equal(labels[17:49], labels2) and np.array_equal(lens[17:49], lens2)
    assert np.array_equal((labels != 0).sum(1), lens)
    ref = oracle.decode(np.ascontiguousarray(sc[:, :8]), nb, 3)["labels"]
    assert np.array_equal(labels[:8], ref)
    ctx.close()


def test_decode_dev_pointers():
    import torch
    nb, T, N = 6, 64, 4
    ctx = _ctx(nb, T, N)
    sc = random_scores(T, N, nb, seed=2)
    d_sc = torch.from_numpy(sc).cuda()
    d_lab = torch.empty((N, T), dtype=torch.int8, device="cuda")
    d_seq = torch.empty((N, T), dtype=torch.int8, device="cuda")
    d_len = torch.empty((N,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.decode_dev(d_sc.data_ptr(), T, N, True, "NACGTXY", d_lab.data_ptr(), d_seq.data_ptr(), d_len.data_ptr())
    ctx.synchronize()
    ref = oracle.decode(sc, nb, 3)["labels"]
    assert np.array_equal(d_lab.cpu().numpy(), ref)
    ctx.close()


def test_abi_error_paths():
    ctx = _ctx(6, 16, 2)
    with pytest.raises(_lib.XbError) as e:
        ctx.decode(random_scores(16, 3, 6), "NACGTXY")              # batch above max_batch
    assert e.value.code == -1 and "max_batch" in str(e.value)
    with pytest.raises(_lib.XbError) as e:
        ctx.encode(np.zeros((1, 80), np.float32))                   # weights never loaded
    assert e.value.code == -4
    with pytest.raises(_lib.XbError):
        ctx.decode(random_scores(17, 2, 6), "NACGTXY")              # T above the context's T
    ctx.close()


def test_decode_random_shapes(monkeypatch):
    """Seeded sweep over small random shapes (T, N, alphabet, blank layout, lanes per state, score scale): every
    combination must be bit-exact -- ring depths (8 / 16), the 64-step label finalisation and the padded tail iterations
    all have their own boundaries."""
    rng = np.random.default_rng(20261004)
    ctxs = {}
    for case in range(40):
        nb = int(rng.integers(4, 7))
        T = int(rng.choice([1, 2, 7, 15, 16, 17, 31, 63, 64, 65, 127, 128, 129, 191, 200, int(rng.integers(3, 260))]))
        N = int(rng.integers(1, 8))
        with_blank = bool(rng.integers(0, 2))
        lps = int(rng.integers(1, 3))
        monkeypatch.setenv("XB_DECODE_LPS", str(lps))
        ctx = ctxs.get(nb)
        if ctx is None:
            ctx = ctxs[nb] = _ctx(nb, 260, 8)
        sc = random_scores(T, N, nb, seed=1000 + case, with_blank=with_blank)
        if case % 5 == 0:
            body = sc.reshape(T, N, nb ** 3, -1)
            body[..., (1 if with_blank else 0):] *= float(rng.choice([0.25, 3.0, 8.0]))     # peaky / flat / far outside tanh range
        _check(ctx, sc, nb, "NACGTXY"[:nb + 1], blank=None if with_blank else 2.0)
    for ctx in ctxs.values():
        ctx.close()


@pytest.mark.parametrize("nb,with_blank,T,N", [(4, True, 77, 3), (5, False, 400, 6), (6, True, 2000, 9), (6, False, 1, 2)])
def test_crf_logz_is_the_oracles_bit_for_bit(nb, with_blank, T, N):
    """xb_crf_logz = CTC_CRF.logZ (crf/model.py:41-46): the Log forward sweep alone.  Same contract arithmetic as the
    oracle, so the partition function must have the same bits; it must also be what the full decode normalises with
    (posteriors of one step sum to 1 -- checked in the oracle tier) and be unchanged by a decode in between."""
    import torch
    ctx = _ctx(nb, T, N)
    sc = random_scores(T, N, nb, seed=3 * nb + T, with_blank=with_blank)
    ref = oracle.decode(sc, nb, 3, blank_score=None if with_blank else 2.0, want=("logz",))["logz"]
    got = ctx.crf_logz(sc)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    ctx.decode(sc, "NACGTXY"[:nb + 1])
    d_sc = torch.from_numpy(sc).cuda()
    d_lz = torch.zeros((N,), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.crf_logz_dev(d_sc.data_ptr(), T, N, with_blank, d_lz.data_ptr())
    ctx.synchronize()
    assert np.array_equal(d_lz.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # linearity property of the partition function: adding a constant c to every score adds T * c
    got2 = ctx.crf_logz(sc + np.float32(0.25)) if with_blank else None
    if got2 is not None:
        assert np.allclose(got2, ref + 0.25 * T, rtol=0, atol=2e-3 * max(T, 1) ** 0.5 + 1e-4)
    with pytest.raises(ValueError):
        ctx.crf_logz(sc[:, :, :-1])
    ctx.close()


def test_model_logz_and_normalise():
    from conftest import make_config
    from xna_basecaller_amd.crf import Model
    model = Model(make_config(32, "NACGTXY")).to("cuda")
    sc = random_scores(120, 4, 6, seed=5, with_blank=True)
    lz = model.logZ(sc)
    ref = oracle.decode(sc, 6, 3, want=("logz",))["logz"]
    assert np.array_equal(lz, ref)
    nrm = model.normalise(sc)
    assert nrm.shape == sc.shape and np.allclose(model.logZ(nrm), 0.0, atol=2e-3)


@pytest.mark.parametrize("nb,with_blank,T,N", [(4, True, 50, 3), (5, True, 333, 4), (5, False, 64, 2), (6, True, 700, 5),
                                               (6, False, 2000, 3)])
def test_crf_scans_are_the_oracles_bit_for_bit(nb, with_blank, T, N):
    """xb_crf_scans: forward / backward scores, logZ and the edge posteriors (CTC_CRF.forward_scores / backward_scores /
    logZ, seqdist posteriors; crf/model.py:41-61) -- every output has the oracle's bits, in every combination of
    requested outputs, and a decode afterwards still gives the oracle's labels (shared workspaces)."""
    import torch
    ctx = _ctx(nb, T, N)
    S, E = nb ** 3, nb + 1
    sc = random_scores(T, N, nb, seed=7 * nb + T, with_blank=with_blank)
    ref = oracle.decode(sc, nb, 3, blank_score=None if with_blank else 2.0, want=("alpha", "beta", "logz", "post"))
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
    for want in (("alpha", "beta", "logz", "post"), ("post",), ("beta",), ("alpha", "logz"), ("logz", "post")):
        got = ctx.crf_scans(sc, want=want)
        for k in want:
            assert got[k].shape == ref[k].shape, k
            assert np.array_equal(bits(got[k]), bits(ref[k])), (want, k, np.abs(got[k] - ref[k]).max())
    # posteriors of a step sum to one
    post = ctx.crf_scans(sc, want=("post",))["post"]
    # (fp32 log domain: the exponent alpha + M + beta - logZ carries the rounding of numbers of size |logZ|)
    assert np.abs(post.reshape(T, N, -1).sum(-1) - 1.0).max() < 4e-6 * max(1.0, float(np.abs(ref["logz"]).max()))
    # device-pointer variant (padded posterior rows)
    ldq = (S * E + 3) & ~3
    d_sc = torch.from_numpy(sc).cuda()
    d_a = torch.zeros((T + 1, N, S), dtype=torch.float32, device="cuda")
    d_b = torch.zeros((T + 1, N, S), dtype=torch.float32, device="cuda")
    d_lz = torch.zeros((N,), dtype=torch.float32, device="cuda")
    d_p = torch.zeros((T, N, ldq), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.crf_scans_dev(d_sc.data_ptr(), T, N, with_blank, d_a.data_ptr(), d_b.data_ptr(), d_lz.data_ptr(), d_p.data_ptr())
    ctx.synchronize()
    assert np.array_equal(bits(d_a.cpu().numpy()), bits(ref["alpha"]))
    assert np.array_equal(bits(d_b.cpu().numpy()), bits(ref["beta"]))
    assert np.array_equal(bits(d_lz.cpu().numpy()), bits(ref["logz"]))
    assert np.array_equal(bits(d_p.cpu().numpy()[:, :, :S * E]), bits(ref["post"]))
    _check(ctx, sc, nb, "NACGTXY"[:nb + 1], blank=None if with_blank else 2.0)
    with pytest.raises(_lib.XbError):
        ctx.crf_scans(sc, want=())
    ctx.close()


def test_model_scan_operators():
    from conftest import make_config
    from xna_basecaller_amd.crf import Model
    model = Model(make_config(32, "NACGTX")).to("cuda")
    sc = random_scores(90, 3, 5, seed=8, with_blank=True)
    ref = oracle.decode(sc, 5, 3, want=("alpha", "beta", "post"))
    assert np.array_equal(model.forward_scores(sc), ref["alpha"])
    assert np.array_equal(model.backward_scores(sc), ref["beta"])
    assert np.array_equal(model.posteriors(sc), ref["post"])
