"""CTC-CRF loss scans (crf/model.py:102-135).

CPU tier: the oracle's restatement of seqdist.ctc_simple (parity unpinned: seqdist 0.0.4 is not in the tree) against an
independent float64 torch-autograd restatement -- logZ, its gradient (the restricted posteriors), the Max-semiring
alignment -- and the host-side gather indices against the reference's own index arithmetic written out in numpy.
GPU tier: xb_ctc_logz / xb_ctc_alignments bit-equal to the oracle for nb 4 / 5 / 6 with ragged target lengths, and the
Model-level ctc_loss (value and gradient) against float64 autograd through the whole normalise + CTC expression."""
import numpy as np
import pytest
import torch

import oracle
from conftest import make_config, random_scores


def _targets(rng, N, Lt, nb, sl, lens=None):
    t = rng.integers(1, nb + 1, (N, Lt)).astype(np.int32)
    if lens is None:
        lens = rng.integers(sl, Lt + 1, N)
        lens[0], lens[-1] = Lt, sl                      # the longest and the shortest legal targets
    lens = np.asarray(lens, dtype=np.int32)
    for b in range(N):
        t[b, lens[b]:] = 0
    return t, lens


def _ctc_logz_f64(x, stay_idx, move_idx, npos, semiring="log"):
    """float64 torch restatement over the dense scores x (T, N, C): returns logz (N,) as a differentiable tensor."""
    T, N, _ = x.shape
    n = stay_idx.shape[1]
    out = []
    zero = torch.full((1,), -1e38, dtype=torch.float64)
    for b in range(N):
        a = torch.full((n,), -1e38, dtype=torch.float64)
        a[0] = 0.0
        si = torch.as_tensor(stay_idx[b], dtype=torch.long)
        mi = torch.as_tensor(move_idx[b], dtype=torch.long)
        for t in range(T):
            x0 = a + x[t, b, si]
            x1 = torch.cat([zero, a[:-1] + x[t, b, mi]])
            st = torch.stack([x0, x1])
            a = torch.logsumexp(st, 0) if semiring == "log" else st.max(0).values
        out.append(a[npos[b] - 1])
    return torch.stack(out)


@pytest.mark.parametrize("nb", [4, 5, 6])
def test_oracle_ctc_against_float64_autograd(nb):
    sl, T, N, Lt = 3, 48, 4, 14
    rng = np.random.default_rng(nb)
    sc = random_scores(T, N, nb, seed=nb)
    targets, lens = _targets(rng, N, Lt, nb, sl)
    stay_idx, move_idx = oracle.ctc_indices(targets, nb, sl)
    # the gather indices are the reference's expression (crf/model.py:108-114) written out in numpy
    t0 = np.clip(targets.astype(np.int64) - 1, 0, None)
    n = Lt - (sl - 1)
    ref_stay = sum(t0[:, i:n + i] * nb ** (sl - i - 1) for i in range(sl)) * (nb + 1)
    assert np.array_equal(stay_idx, ref_stay) and np.array_equal(move_idx, ref_stay[:, 1:] + t0[:, :n - 1] + 1)
    o = oracle.ctc_logz(sc, targets, lens, nb, sl, want_grads=True)
    x = torch.tensor(sc, dtype=torch.float64, requires_grad=True)
    lz = _ctc_logz_f64(x, stay_idx, move_idx, lens + 1 - sl)
    lz.sum().backward()
    assert np.abs(o["logz"] - lz.detach().numpy()).max() < 2e-4
    dense = np.zeros(sc.shape, np.float64)
    bi = np.arange(N)[:, None]
    for t in range(T):
        np.add.at(dense[t], (bi, stay_idx), o["stay"][t])
        np.add.at(dense[t], (bi, move_idx), o["move"][t])
    assert np.abs(dense - x.grad.numpy()).max() < 2e-4
    assert np.abs(dense.sum(axis=2) - 1.0).max() < 2e-4          # one edge per time step on every path
    # Max semiring: the alignment is a monotone path of unit steps that ends on the target's last position, and its score
    # is the best path score
    om = oracle.ctc_logz(sc, targets, lens, nb, sl, semiring="max", want_grads=True)
    best = _ctc_logz_f64(torch.tensor(sc, dtype=torch.float64), stay_idx, move_idx, lens + 1 - sl, "max").numpy()
    assert np.abs(om["logz"] - best).max() < 2e-4
    pos = om["stay"].argmax(axis=2)                              # (T, N)
    assert np.all(om["stay"].sum(axis=2) == 1.0) and np.all(pos[0] == 0)
    d = np.diff(pos, axis=0)
    assert np.all((d == 0) | (d == 1))
    assert np.all((pos[-1] == lens - sl) | (pos[-1] == lens - sl - 1))
    # the path's own score, summed in float64, is the reported maximum
    for b in range(N):
        tot = 0.0
        for t in range(T):
            nxt = pos[t + 1, b] if t + 1 < T else lens[b] - sl
            tot += sc[t, b, stay_idx[b, pos[t, b]]] if nxt == pos[t, b] else sc[t, b, move_idx[b, pos[t, b]]]
        assert abs(tot - om["logz"][b]) < 2e-3


def test_oracle_ctc_rejects_bad_lengths():
    sc = random_scores(8, 2, 4)
    t, _ = _targets(np.random.default_rng(0), 2, 6, 4, 3, lens=[6, 6])
    with pytest.raises(ValueError):
        oracle.ctc_logz(sc, t, [6, 2], 4, 3)                     # shorter than state_len
    with pytest.raises(ValueError):
        oracle.ctc_logz(sc, t, [7, 6], 4, 3)                     # longer than the target row


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("nb,T,N,Lt", [(4, 37, 3, 9), (5, 120, 5, 40), (6, 200, 7, 130), (6, 64, 2, 3), (5, 90, 4, 300)])
def test_gpu_ctc_scans_are_the_oracles_bit_for_bit(nb, T, N, Lt):
    from xna_basecaller_amd import _lib
    sl = 3
    rng = np.random.default_rng(Lt + nb)
    sc = random_scores(T, N, nb, seed=T)
    targets, lens = _targets(rng, N, Lt, nb, sl)
    ctx = _lib.Context(0, nb, sl, 32, 19, 5, 5.0, 2.0, T * 5, N)
    got = ctx.ctc_logz(sc, targets, lens, want_grads=True)
    ref = oracle.ctc_logz(sc, targets, lens, nb, sl, want_grads=True)
    assert np.array_equal(got["logz"], ref["logz"])
    assert np.array_equal(got["stay"], ref["stay"]) and np.array_equal(got["move"], ref["move"])
    only = ctx.ctc_logz(sc, targets, lens)                       # forward sweep alone
    assert np.array_equal(only["logz"], ref["logz"])
    al, best = ctx.ctc_alignments(sc, targets, lens)
    refm = oracle.ctc_logz(sc, targets, lens, nb, sl, semiring="max", want_grads=True)
    assert np.array_equal(best, refm["logz"]) and np.array_equal(al, refm["stay"])
    with pytest.raises(_lib.XbError):
        ctx.ctc_logz(sc, targets, np.full(N, sl - 1, np.int32))
    with pytest.raises(_lib.XbError):
        ctx.ctc_logz(sc, np.full_like(targets, nb + 1), lens)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("labels", ["NACGT", "NACGTXY"])
def test_gpu_model_ctc_loss_and_gradient(labels):
    """model.seqdist.ctc_loss (training.py:82's criterion) on the device: the value and d loss / d scores against float64
    autograd through the whole expression  -logZ_ctc(scores - logZ_crf(scores) / T) / length  (crf/model.py:118-131)."""
    from xna_basecaller_amd.crf import Model
    nb, sl, T, N, Lt = len(labels) - 1, 3, 30, 3, 10
    model = Model(make_config(32, labels)).to("cuda")
    rng = np.random.default_rng(7)
    sc = random_scores(T, N, nb, seed=11)
    targets, lens = _targets(rng, N, Lt, nb, sl)
    loss, grad = model.seqdist.ctc_loss(sc, targets, lens, want_grad=True)
    per = model.seqdist.ctc_loss(sc, targets, lens, reduction="none")
    assert abs(float(per.mean()) - float(loss)) < 1e-6
    # float64 restatement: CRF logZ (dense recursion over the idx table) and the CTC lattice
    idx = model.seqdist.idx.numpy().astype(np.int64)
    S, E = nb ** sl, nb + 1
    x = torch.tensor(sc, dtype=torch.float64, requires_grad=True)
    a = torch.zeros((N, S), dtype=torch.float64)
    for t in range(T):
        a = torch.logsumexp(x[t].reshape(N, S, E) + a[:, torch.as_tensor(idx)], dim=2)
    xn = x - (torch.logsumexp(a, dim=1) / T)[None, :, None]
    stay_idx, move_idx = oracle.ctc_indices(targets, nb, sl)
    lz = _ctc_logz_f64(xn, stay_idx, move_idx, lens + 1 - sl)
    ref = (-(lz / torch.as_tensor(lens, dtype=torch.float64))).mean()
    ref.backward()
    assert abs(float(ref.detach()) - float(loss)) < 2e-4
    assert np.abs(grad - x.grad.numpy()).max() < 2e-5
    # clipping zeroes the clipped chunks' gradient; alignments come back one-hot
    lc, gc = model.seqdist.ctc_loss(sc, targets, lens, loss_clip=1e-3, reduction="none", want_grad=True)
    assert np.all(lc <= 1e-3) and np.all(gc[:, per > 1e-3] == 0)
    al = model.seqdist.ctc_viterbi_alignments(sc, targets, lens)
    assert al.shape == (T, N, Lt - sl + 1) and np.all(al.sum(axis=2) == 1)
    st, mv = model.seqdist.prepare_ctc_scores(sc, targets)
    assert st.shape == (T, N, Lt - sl + 1) and mv.shape == (T, N, Lt - sl)
    assert np.array_equal(st[:, 1, 2], sc[:, 1, stay_idx[1, 2]]) and np.array_equal(mv[:, 2, 0], sc[:, 2, move_idx[2, 0]])
