"""`bonito evaluate` (cli/evaluate.py:25-96): the host-side accuracy (a restatement of the reference's
parasail call, util.py:402-424 -- parasail is in no image, so "unpinned") against an independent pure-Python Gotoh with the
same stated tie rules, the validation-data loader, and on the GPU the evaluator itself on a synthetic ctc-data directory."""
import os

import numpy as np
import pytest

from conftest import make_config, seeded_state_dict, encoder_shapes
from xna_basecaller_amd import _lib, data as xdata, util


def _gotoh(ref, seq):
    """Smith-Waterman, affine gaps (8 + 4 (k - 1)), +5 / -4; trace-back preferring diagonal, deletion, insertion."""
    n, m, NEG = len(seq), len(ref), -10 ** 9
    H = [[0] * (m + 1) for _ in range(n + 1)]
    E = [[NEG] * (m + 1) for _ in range(n + 1)]
    F = [[NEG] * (m + 1) for _ in range(n + 1)]
    best, bi, bj = 0, 0, 0
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            E[i][j] = max(E[i][j - 1] - 4, H[i][j - 1] - 8)
            F[i][j] = max(F[i - 1][j] - 4, H[i - 1][j] - 8)
            d = H[i - 1][j - 1] + (5 if seq[i - 1] == ref[j - 1] else -4)
            H[i][j] = max(0, d, E[i][j], F[i][j])
            if H[i][j] > best:
                best, bi, bj = H[i][j], i, j
    c = {"=": 0, "X": 0, "I": 0, "D": 0}
    i, j, st = bi, bj, 0
    while i > 0 and j > 0:
        if st == 0:
            if H[i][j] == 0:
                break
            d = H[i - 1][j - 1] + (5 if seq[i - 1] == ref[j - 1] else -4)
            if H[i][j] == d:
                c["=" if seq[i - 1] == ref[j - 1] else "X"] += 1
                i, j = i - 1, j - 1
            elif H[i][j] == E[i][j]:
                st = 1
            else:
                st = 2
        elif st == 1:
            c["D"] += 1
            st = 0 if E[i][j] == H[i][j - 1] - 8 else 1
            j -= 1
        else:
            c["I"] += 1
            st = 0 if F[i][j] == H[i - 1][j] - 8 else 2
            i -= 1
    # coverage as util.py:410 takes it: len(alignment.traceback.ref) / len(ref) -- the traceback string has one character per
    # alignment COLUMN ('-' where the reference does not take part), so insertions count
    return c, sum(c.values()) / max(m, 1), best


def test_accuracy_against_independent_gotoh():
    rng = np.random.default_rng(4)
    for case in range(60):
        m = int(rng.integers(1, 70))
        ref = "".join(rng.choice(list("ACGTXY"), m))
        seq = list(ref)
        for _ in range(int(rng.integers(0, 8))):                # substitutions, insertions, deletions
            k = int(rng.integers(0, max(len(seq), 1)))
            op = rng.integers(0, 3)
            if op == 0 and seq:
                seq[k] = str(rng.choice(list("ACGTXY")))
            elif op == 1:
                seq.insert(k, str(rng.choice(list("ACGTXY"))))
            elif seq:
                del seq[k]
        seq = "".join(seq) if case % 7 else "".join(rng.choice(list("ACGT"), int(rng.integers(1, 30))))
        if not seq:
            continue
        want, cov, score = _gotoh(ref, seq)
        acc, got = _lib.align_accuracy(ref, seq, want_counts=True)
        assert got == want, (ref, seq)
        den = sum(want.values())
        assert abs(acc - (100.0 * want["="] / den if den else 0.0)) < 1e-9
        # the trace is an alignment of that score: 5 '=' - 4 'X' - (8 per gap + 4 per further gap column)
        assert score >= 5 * want["="] - 4 * want["X"] - 8 * (want["I"] + want["D"])
        assert util.accuracy(ref, seq, min_coverage=cov + 1e-9) == 0.0
        if den:
            assert util.accuracy(ref, seq, min_coverage=cov - 1e-9) == acc
        bal = _lib.align_accuracy(ref, seq, balanced=True)
        d2 = want["="] + want["X"] + want["D"]
        assert abs(bal - (100.0 * (want["="] - want["I"]) / d2 if d2 else 0.0)) < 1e-9
    assert util.accuracy("ACGT" * 10, "ACGT" * 10) == 100.0 and util.accuracy("ACGT", "") == 0.0
    # an insertion-rich call near the gate: 10 reference bases aligned over 12 columns -> coverage 1.2 of a 10-base reference
    acc, c = _lib.align_accuracy("ACGTACGTAC", "ACGTATTCGTAC", want_counts=True)
    assert c == {"=": 10, "X": 0, "I": 2, "D": 0} and util.accuracy("ACGTACGTAC", "ACGTATTCGTAC", min_coverage=1.1) == acc > 0
    with pytest.raises(_lib.XbError):
        _lib.align_accuracy("A" * 9000, "C" * 9000)                 # beyond the bounded DP: refused, not attempted
    assert util.decode_ref(np.array([1, 2, 0, 5, 6, 0]), list("NACGTXY")) == "ACXY"


def test_validation_loader(tmp_path):
    rng = np.random.default_rng(0)
    N, L, R = 200, 50, 12
    np.save(tmp_path / "chunks.npy", rng.standard_normal((N, L)).astype(np.float32))
    np.save(tmp_path / "references.npy", rng.integers(0, 5, (N, R)).astype(np.int16))
    np.save(tmp_path / "reference_lengths.npy", rng.integers(1, R, N).astype(np.int16))
    c, t, l = xdata.load_validation(100, tmp_path)                 # 97 % / 3 % split of the first 100
    assert c.shape == (3, L) and t.shape == (3, R) and l.shape == (3,)
    assert np.array_equal(c, np.load(tmp_path / "chunks.npy")[97:100])
    np.save(tmp_path / "indices.npy", np.arange(N)[::-1].copy())
    c2, _, _ = xdata.load_validation(100, tmp_path)
    assert np.array_equal(c2, np.load(tmp_path / "chunks.npy")[::-1][97:100])
    os.makedirs(tmp_path / "validation")
    for name in ("chunks", "references", "reference_lengths"):
        np.save(tmp_path / "validation" / (name + ".npy"), np.load(tmp_path / (name + ".npy"))[:7])
    c3, t3, l3 = xdata.load_validation(100, tmp_path)
    assert c3.shape == (7, L) and len(l3) == 7


def test_evaluate_argparser_defaults():
    from xna_basecaller_amd.cli import evaluate
    a = evaluate.argparser().parse_args(["model_dir", "--directory", "d"])
    assert (a.device, a.seed, a.weights, a.chunks, a.batchsize, a.beamsize, a.poa, a.min_coverage) == \
        ("cuda", 9, "0", 1000, 96, 5, False, 0.5)                  # cli/evaluate.py:99-116


@pytest.mark.gpu
def test_evaluate_on_a_synthetic_ctc_directory(tmp_path, capsys):
    """The evaluator end to end: model directory (config.toml + weights_1.tar) + ctc-data whose references are the model's own
    calls, a third of them corrupted: accuracies 100 % where untouched, lower where corrupted, the reference's report lines."""
    import torch
    from xna_basecaller_amd import toml_lite
    from xna_basecaller_amd.cli import evaluate
    from xna_basecaller_amd.crf import Model
    F, L, N = 64, 2000, 30
    labels = list("NACGTXY")
    cfg = make_config(F, labels)
    cfg["basecaller"] = {"batchsize": 16, "chunksize": L, "overlap": 100}
    mdir = tmp_path / "model@v1"
    mdir.mkdir()
    (mdir / "config.toml").write_text(toml_lite.dumps(cfg))
    keys, shapes = encoder_shapes(F, 6)
    sd = seeded_state_dict(keys, shapes, seed=5)
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, str(mdir / "weights_1.tar"))
    model = Model(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to("cuda")
    x = np.random.default_rng(3).standard_normal((N, L)).astype(np.float32)
    seq, lens = model.basecall_chunks(x[:, None, :])
    calls = [seq[i, :lens[i]].tobytes().decode() for i in range(N)]
    assert min(lens) > 20
    code = {c: i for i, c in enumerate(labels)}
    refs = np.zeros((N, max(lens) + 5), np.int16)
    for i, s in enumerate(calls):
        r = list(s)
        if i % 3 == 0:                                             # corrupt: drop a tenth, flip a tenth
            r = [c for k, c in enumerate(r) if k % 10 != 3]
            r = [("A" if c != "A" else "C") if k % 10 == 7 else c for k, c in enumerate(r)]
        refs[i, :len(r)] = [code[c] for c in r]
    data = tmp_path / "ctc" / "validation"
    data.mkdir(parents=True)
    np.save(data / "chunks.npy", x)
    np.save(data / "references.npy", refs)
    np.save(data / "reference_lengths.npy", (refs != 0).sum(1).astype(np.int16))
    for name in ("chunks", "references", "reference_lengths"):
        np.save(tmp_path / "ctc" / (name + ".npy"), np.load(data / (name + ".npy"))[:2])
    args = evaluate.argparser().parse_args([str(mdir), "--directory", str(tmp_path / "ctc"), "--weights", "1", "--batchsize", "16"])
    acc = evaluate.main(args)
    out = capsys.readouterr().out
    assert len(acc) == N
    assert all(a == 100.0 for i, a in enumerate(acc) if i % 3) and all(60.0 < a < 95.0 for i, a in enumerate(acc) if i % 3 == 0)
    for line in ("* loading data", "* loading model 1", "* calling", "* decoding refs", "* computing accuracies", "* mean ", "* median ",
                 "* time ", "* samples/s "):
        assert line in out
