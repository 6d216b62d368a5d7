"""
Synthetic model weights and geometry helpers (bench.py, smoke(), tests): the reference's weights_N.tar files are not
distributable (download_data.sh:57-67 fetches them), so benchmarks and tests use seeded random weights in the
reference's state-dict layout (encoder.{0,1,2}.conv.*, encoder.{4..8}.rnn.*, encoder.9.linear.*).
"""
import numpy as np

# state-dict keys of the 10-module inference encoder, in order
STATE_DICT_ORDER = (
    ["encoder.%d.conv.%s" % (i, p) for i in (0, 1, 2) for p in ("weight", "bias")]
    + ["encoder.%d.rnn.%s" % (i, p) for i in (4, 5, 6, 7, 8)
       for p in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    + ["encoder.9.linear.weight", "encoder.9.linear.bias"]
)


def encoder_shapes(features, n_base, state_len=3, winlen=19):
    """(keys, shapes) of the inference encoder's 28 tensors in PyTorch layout."""
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
    keys = list(STATE_DICT_ORDER)
    return keys, [shapes[k] for k in keys]


def seeded_state_dict(keys, shapes, seed):
    """N(0, 1/sqrt(fan_in)) weights, clipped N(0, 0.5) biases, bias_hh = 0 (nn.py:209-213); the generator of
    tests/golden/make_golden.py (full-size fixture weights are regenerated, not stored)."""
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


def seeded_weights(features, n_base, seed=25):
    keys, shapes = encoder_shapes(features, n_base)
    return seeded_state_dict(keys, shapes, seed)


def peaky_weights(features, n_base, seed=25, input_gain=2.0, linear_gain=10.0, blank_bias=2.0):
    """Seeded weights whose scores DEPEND on the signal and whose posteriors are peaky -- the regime of a trained
    basecaller, where end-to-end label identity can be asserted (tests/test_gpu_fullsize.py, tools/peaky_parity.py).
    With the plain seeded weights the five random LSTM layers damp the time-varying part of the signal to ~1e-3 of a
    static offset set by the biases: the scores are nearly constant in time, the Viterbi path is all-stay or all-move and
    the posteriors are flat.  Here the LSTM biases are zero (no static offset: the gates sit at their linear point and
    follow the input), the LSTM input projections are scaled by `input_gain`, and the CRF linear layer by `linear_gain` with
    bias -blank_bias (5 tanh saturates on a fifth of the edges; a move needs evidence against the blank score 2.0).  At
    features 768 about 0.45-0.5 bases are called per time step (the reference's real regime is ~0.55: chunksize / 9 bases
    per 2000 steps) and the scores move by ~0.5 from step to step.  The model is ~10x more sensitive to
    rounding than the plain seeded one (a 1e-6 relative change of the signal moves a score by up to 4e-5); measured with the
    oracle, tools/peaky_parity.py prints the numbers."""
    sd = seeded_weights(features, n_base, seed)
    for l in range(4, 9):
        k = "encoder.%d.rnn.weight_ih_l0" % l
        sd[k] = (sd[k] * np.float32(input_gain)).astype(np.float32)
        sd["encoder.%d.rnn.bias_ih_l0" % l] = np.zeros_like(sd["encoder.%d.rnn.bias_ih_l0" % l])
    sd["encoder.9.linear.weight"] = (sd["encoder.9.linear.weight"] * np.float32(linear_gain)).astype(np.float32)
    sd["encoder.9.linear.bias"] = np.full_like(sd["encoder.9.linear.bias"], -np.float32(blank_bias))
    return sd
