"""CPU: the oracle (oracle/xna_oracle.c) against the golden fixtures made from the reference's own Python
(tests/golden/make_golden.py) and against independent float64 restatements."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, random_scores, seeded_state_dict


def test_spec_math_accuracy():
    x = np.linspace(-87, 5, 100001).astype(np.float32)
    e = oracle.expf(x)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 2e-7
    # the argument is clamped to [-87, 88]: anything smaller evaluates as exp(-87) = 1.6e-38 (no flush to zero)
    tiny = oracle.expf(np.array([-87.0, -88.0, -1e38, -np.inf], np.float32))
    assert np.all(tiny == tiny[0]) and 1e-38 < tiny[0] < 2e-38
    x = np.exp(np.linspace(np.log(1e-8), np.log(8), 100001)).astype(np.float32)
    l = oracle.logf(x)
    assert np.max(np.abs(l - np.log(x.astype(np.float64)))) < 2e-6


def test_crf_idx_matches_reference():
    z = np.load(os.path.join(GOLDEN, "crf_idx.npz"))
    for nb in (4, 5, 6):
        assert np.array_equal(oracle.crf_idx(nb, 3), z["idx_nb%d" % nb])
        assert int(z["n_score_nb%d" % nb]) == (nb + 1) * nb ** 3
    for sl in (1, 2, 4):
        assert np.array_equal(oracle.crf_idx(4, sl), z["idx_nb4_sl%d" % sl])
    idx = oracle.crf_idx(6, 3)
    assert idx[0].tolist() == [0, 0, 36, 72, 108, 144, 180]
    assert idx[215].tolist() == [215, 35, 71, 107, 143, 179, 215]


def _fp64_posteriors(sc, nb, sl):
    """Independent float64 restatement: posteriors = d logZ / d scores (autograd)."""
    import torch
    T, N, C = sc.shape
    S, E = nb ** sl, nb + 1
    idx = torch.from_numpy(oracle.crf_idx(nb, sl)).long()
    Ms = torch.tensor(sc, dtype=torch.float64).reshape(T, N, S, E).requires_grad_(True)
    a = torch.zeros(N, S, dtype=torch.float64)
    for t in range(T):
        a = torch.logsumexp(Ms[t] + a[:, idx], dim=2)
    logZ = torch.logsumexp(a, dim=1)
    g, = torch.autograd.grad(logZ.sum(), Ms)
    return g.reshape(T, N, C).numpy(), logZ.detach().numpy()


def _backpointer_viterbi(Q, idx):
    T, N, S, E = Q.shape
    lab = np.zeros((N, T), dtype=np.int8)
    for n in range(N):
        v = np.zeros(S)
        bp = np.zeros((T, S), dtype=np.int64)
        for t in range(T):
            cand = Q[t, n] + v[idx]
            bp[t] = cand.argmax(1)
            v = cand.max(1)
        j = v.argmax()
        for t in range(T - 1, -1, -1):
            k = bp[t, j]
            lab[n, t] = k
            j = idx[j, k]
    return lab


@pytest.mark.parametrize("nb", [4, 5, 6])
def test_decode_against_float64_restatement(nb):
    sl, T, N = 3, 64, 3
    sc = random_scores(T, N, nb, sl, seed=nb)
    out = oracle.decode(sc, nb, sl, want=("logz", "post", "alpha", "beta", "qlog"))
    P, logZ = _fp64_posteriors(sc, nb, sl)
    assert np.abs(out["logz"] - logZ).max() < 1e-3
    assert np.abs(out["post"] - P).max() < 1e-4
    assert np.abs(out["post"].sum(2) - 1).max() < 1e-3
    assert np.all(out["alpha"][0] == 0) and np.all(out["beta"][-1] == 0)
    assert np.abs(out["qlog"] - np.log(out["post"].astype(np.float64) + 1e-8)).max() < 2e-6
    lab = _backpointer_viterbi(np.log(P + 1e-8).reshape(T, N, nb ** sl, nb + 1), oracle.crf_idx(nb, sl))
    assert np.array_equal(lab, out["labels"])


def test_scaled_comparison_model_is_float64_grade():
    """oracle.decode_scaled (census comparison model, not the contract): a_t * 2^Ka[t] = exp(alpha_t) of the float64
    logsumexp recursions (crf/model.py:41-46 semantics), and posteriors at fp32 rounding level."""
    import torch
    nb, sl, T, N = 5, 3, 400, 2
    sc = random_scores(T, N, nb, sl, seed=77)
    out = oracle.decode_scaled(sc, nb, sl, want=("avec", "aexp", "post", "logz"))
    S, E = nb ** sl, nb + 1
    idx = torch.from_numpy(oracle.crf_idx(nb, sl)).long()
    Ms = torch.tensor(sc, dtype=torch.float64).reshape(T, N, S, E)
    a = torch.zeros(N, S, dtype=torch.float64)
    alphas = [a]
    for t in range(T):
        a = torch.logsumexp(Ms[t] + a[:, idx], dim=2)
        alphas.append(a)
    alpha = torch.stack(alphas).numpy()
    got = np.log(out["avec"].astype(np.float64)) + out["aexp"][:, :, None] * np.log(2.0)
    assert np.abs(got - alpha).max() < 1e-4
    assert abs(out["aexp"][-1]).min() > 100          # the exponent bookkeeping is exercised
    P, logZ = _fp64_posteriors(sc[:64], nb, sl)
    out = oracle.decode_scaled(sc[:64], nb, sl, want=("post",))
    assert np.abs(out["post"] - P).max() < 2e-6
    assert np.abs(out["post"].sum(2) - 1).max() < 1e-5
    # and the contract (log domain) calls the same labels on this easy case
    assert np.array_equal(oracle.decode(sc, nb, sl)["labels"], oracle.decode_scaled(sc, nb, sl)["labels"])


def test_decode_blank_column_equivalence():
    """(T,N,S*nb) scores + constant blank == (T,N,S*E) scores with the blank column (nn.py:123-130)."""
    for nb in (4, 5, 6):
        a = random_scores(20, 2, nb, seed=3, with_blank=True)
        b = random_scores(20, 2, nb, seed=3, with_blank=False)
        la = oracle.decode(a, nb, 3)["labels"]
        lb = oracle.decode(b, nb, 3, blank_score=2.0)["labels"]
        assert np.array_equal(la, lb)


def test_decode_golden_reference_control_flow():
    """Fixtures: reference decode_batch/viterbi/path_to_str control flow + restated seqdist math (flagged)."""
    meta = json.load(open(os.path.join(GOLDEN, "decode_meta.json")))
    z = np.load(os.path.join(GOLDEN, "decode_small.npz"))
    total = mism = 0
    for case in meta["cases"]:
        name, nb = case["name"], case["nb"]
        sc = z[name + "/scores_f16"].astype(np.float32)
        out = oracle.decode(sc, nb, 3, want=("post",))
        ref = z[name + "/labels"]
        total += ref.size
        mism += int((out["labels"] != ref).sum())
        assert np.abs(out["post"][:, :, ::13] - z[name + "/post_sample"]).max() < 1e-4
        seqs = oracle.decode_batch(sc, case["labels"], 3)
        if np.array_equal(out["labels"], ref):
            assert seqs == case["sequences"]
    # the stand-in's fp32 torch arithmetic and the oracle's differ in rounding; near-ties may flip
    assert mism <= max(1, total // 200), (mism, total)


def test_pack_and_path_to_str():
    labels = np.array([[0, 1, 0, 0, 5, 6, 0, 2], [0, 0, 0, 0, 0, 0, 0, 0], [3, 3, 3, 3, 3, 3, 3, 3]], np.int8)
    seq, qs, lens = oracle.pack(labels, "NACGTXY")
    assert lens.tolist() == [4, 0, 8]
    assert seq[0, :4].tobytes() == b"AXYC" and not seq[0, 4:].any()
    assert qs[0, :4].tolist() == [79] * 4 and not qs[0, 4:].any()
    assert seq[2].tobytes() == b"GGGGGGGG"


def test_encoder_small_golden():
    """Oracle encoder vs the reference's own nn.py modules (fp32 CPU) on seeded small models."""
    meta = json.load(open(os.path.join(GOLDEN, "encoder_meta.json")))
    z = np.load(os.path.join(GOLDEN, "encoder_small.npz"))
    for case in meta["cases"]:
        name, F, labels = case["name"], case["features"], case["labels"]
        sd = {k: z["%s/w/%s" % (name, k)] for k in case["keys"]}
        nb = len(labels) - 1
        scores, lo = oracle.encode(z[name + "/signal"], sd, F, nb, 3, want_lstm_out=True)
        assert scores.shape == z[name + "/scores"].shape
        assert np.abs(scores - z[name + "/scores"]).max() < 2e-5
        assert np.abs(lo - z[name + "/lstm4_out"]).max() < 2e-5
        c3 = oracle.conv1d_silu(oracle.conv1d_silu(oracle.conv1d_silu(
            z[name + "/signal"], sd["encoder.0.conv.weight"], sd["encoder.0.conv.bias"], 1, 2),
            sd["encoder.1.conv.weight"], sd["encoder.1.conv.bias"], 1, 2),
            sd["encoder.2.conv.weight"], sd["encoder.2.conv.bias"], 5, 9)
        assert np.abs(c3 - z[name + "/conv_out"]).max() < 1e-5
        # blank layout: column 0 of every (state, .) row is exactly blank_score
        S = nb ** 3
        assert np.all(scores.reshape(scores.shape[0], scores.shape[1], S, nb + 1)[..., 0] == 2.0)


def test_encoder_full_size_golden():
    meta = json.load(open(os.path.join(GOLDEN, "encoder_meta.json")))["full"]
    z = np.load(os.path.join(GOLDEN, "encoder_full.npz"))
    assert meta["n_params"] == 24854904
    sd = seeded_state_dict(meta["keys"], meta["shapes"], meta["seed"])
    x = np.random.default_rng(meta["signal_seed"]).standard_normal((meta["N"], 1, meta["L"])).astype(np.float32)
    scores = oracle.encode(x, sd, 768, 6, 3)
    assert np.abs(scores[0] - z["scores_t0"]).max() < 5e-5
    assert np.abs(scores[-1] - z["scores_tlast"]).max() < 5e-5
    assert np.abs(scores[50, :, ::7] - z["scores_mid"]).max() < 5e-5
    assert abs(np.abs(scores.astype(np.float64)).sum() - float(z["abs_checksum"])) < 1e-3 * scores.size
