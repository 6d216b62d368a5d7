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


from xna_basecaller_amd.synthetic import encoder_shapes, seeded_state_dict  # noqa: E402,F401  (re-exported for the tests)


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
