import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

SHIPPED_CONFIG = {  # values of the reference's models/xna_r9.4.1_e8_sup@v3.3/config.toml
    "global_norm": {"state_len": 3},
    "qscore": {"bias": 0.3498, "scale": 0.9722},
    "input": {"features": 1},
    "model": {"package": "bonito.crf"},
    "labels": {"labels": ["N", "A", "C", "G", "T", "X", "Y"]},
    "encoder": {"stride": 5, "activation": "swish", "features": 768, "winlen": 19, "scale": 5.0,
                "rnn_type": "lstm", "blank_score": 2.0},
    "basecaller": {"batchsize": 384, "chunksize": 3600, "overlap": 500},
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_config(features=768, labels=("N", "A", "C", "G", "T", "X", "Y")):
    c = json.loads(json.dumps(SHIPPED_CONFIG))
    c["encoder"]["features"] = features
    c["labels"]["labels"] = list(labels)
    return c


def seeded_state_dict(keys, shapes, seed):
    """Same generator as tests/golden/make_golden.py:seeded_state_dict (weights are not stored for the
    full-size fixture, they are regenerated)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, shp in zip(keys, shapes):
        shp = tuple(shp)
        if k.endswith("bias_hh_l0"):
            a = np.zeros(shp, np.float32)
        elif "bias" in k:
            a = np.clip(0.5 * rng.standard_normal(shp), -1, 1).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = (rng.standard_normal(shp) / np.sqrt(fan_in)).astype(np.float32)
        out[k] = a
    return out


def encoder_shapes(features, n_base, state_len=3, winlen=19):
    import oracle
    F = features
    shapes = {"encoder.0.conv.weight": (4, 1, 5), "encoder.0.conv.bias": (4,),
              "encoder.1.conv.weight": (16, 4, 5), "encoder.1.conv.bias": (16,),
              "encoder.2.conv.weight": (F, 16, winlen), "encoder.2.conv.bias": (F,),
              "encoder.9.linear.weight": (n_base ** (state_len + 1), F),
              "encoder.9.linear.bias": (n_base ** (state_len + 1),)}
    for l in range(4, 9):
        shapes["encoder.%d.rnn.weight_ih_l0" % l] = (4 * F, F)
        shapes["encoder.%d.rnn.weight_hh_l0" % l] = (4 * F, F)
        shapes["encoder.%d.rnn.bias_ih_l0" % l] = (4 * F,)
        shapes["encoder.%d.rnn.bias_hh_l0" % l] = (4 * F,)
    keys = list(oracle.STATE_DICT_ORDER)
    return keys, [shapes[k] for k in keys]


def random_scores(T, N, nb, sl=3, seed=0, blank=2.0, with_blank=True):
    """5*tanh(N(0,1)) scores with the constant blank column."""
    rng = np.random.default_rng(seed)
    S, E = nb ** sl, nb + 1
    sc = (5.0 * np.tanh(rng.standard_normal((T, N, S, E)))).astype(np.float32)
    sc[..., 0] = blank
    if with_blank:
        return sc.reshape(T, N, S * E)
    return np.ascontiguousarray(sc[..., 1:]).reshape(T, N, S * nb)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
