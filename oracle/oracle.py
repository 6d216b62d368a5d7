"""
ctypes/numpy front-end of oracle/xna_oracle.c (test infrastructure, see that file's header).

Reference call sites restated (all under /root/reference/ub-bonito/bonito/):
  crf/model.py:31-36, 41-46, 92-100, 215-218   CRF table, logZ, viterbi, path_to_str, decode_batch
  nn.py:57-68, 112-133, 176-193, 216-220      Convolution, LinearCRFEncoder, RNNWrapper/LSTM
  crf/model.py:142-160                         rnn_encoder layer order
  crf/basecall.py:60-67                        left-pack of called sequences
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libxna_oracle.so")
_lib = None

f32p = C.POINTER(C.c_float)
i8p = C.POINTER(C.c_int8)
i32p = C.POINTER(C.c_int32)


def build(force=False):
    """Compile oracle/xna_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "xna_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libxna_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.xo_expf.restype = C.c_float
        _lib.xo_expf.argtypes = [C.c_float]
        _lib.xo_logf.restype = C.c_float
        _lib.xo_logf.argtypes = [C.c_float]
        _lib.xo_decode.restype = C.c_int
        _lib.xo_decode_logdomain.restype = C.c_int
        _lib.xo_decode_scaled.restype = C.c_int
        _lib.xo_lstm.restype = C.c_int
        _lib.xo_linear_crf.restype = C.c_int
        _lib.xo_encode.restype = C.c_int
        _lib.xo_num_threads.restype = C.c_int
        _lib.xo_ctc_logz.restype = C.c_int
        _lib.xo_beam_search.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, typ=f32p):
    return None if a is None else a.ctypes.data_as(typ)


def num_threads():
    return int(lib().xo_num_threads())


def set_num_threads(n):
    lib().xo_set_num_threads(C.c_int(int(n)))


def expf(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().xo_expf_array(_p(x), _p(y), C.c_int64(x.size))
    return y


def logf(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().xo_logf_array(_p(x), _p(y), C.c_int64(x.size))
    return y


def crf_idx(n_base, state_len):
    S, E = n_base ** state_len, n_base + 1
    idx = np.empty((S, E), dtype=np.int32)
    lib().xo_crf_idx(C.c_int(n_base), C.c_int(state_len), _p(idx, i32p))
    return idx


def _decode_args(scores, n_base, state_len, blank_score):
    scores = _f32(scores)
    T, N, Cin = scores.shape
    S, E = n_base ** state_len, n_base + 1
    if Cin == S * E:
        has_blank, blank = 1, 0.0
    elif Cin == S * n_base and blank_score is not None:
        has_blank, blank = 0, float(blank_score)
    else:
        raise ValueError("scores last dim %d matches neither S*E=%d nor S*nb=%d" % (Cin, S * E, S * n_base))
    return scores, T, N, S, E, has_blank, blank


def decode(scores, n_base, state_len, blank_score=None, want=()):
    """
    The decode contract (xna_oracle.c header).  scores (T,N,C) fp32; C = S*(nb+1) (blank column present) or,
    with blank_score given and C = S*nb, the blank is the constant.  Returns dict with 'labels' (N,T) int8 and
    any of want = ('alpha','beta','logz','post','qlog','amax','bmax').
    """
    scores, T, N, S, E, has_blank, blank = _decode_args(scores, n_base, state_len, blank_score)
    out = {"labels": np.empty((N, T), dtype=np.int8)}
    shapes = {"alpha": (T + 1, N, S), "beta": (T + 1, N, S), "logz": (N,), "post": (T, N, S * E),
              "qlog": (T, N, S * E), "amax": (T + 1, N, S), "bmax": (T + 1, N, S)}
    for k in want:
        out[k] = np.zeros(shapes[k], dtype=np.float32)
    rc = lib().xo_decode(_p(scores), C.c_int(T), C.c_int(N), C.c_int(n_base), C.c_int(state_len),
                         C.c_int(has_blank), C.c_float(blank), _p(out["labels"], i8p),
                         _p(out.get("alpha")), _p(out.get("beta")), _p(out.get("logz")),
                         _p(out.get("post")), _p(out.get("qlog")), _p(out.get("amax")), _p(out.get("bmax")))
    if rc:
        raise MemoryError("xo_decode failed")
    return out


def decode_scaled(scores, n_base, state_len, blank_score=None, want=()):
    """
    NOT the contract: the same decode on scaled probabilities (float64-grade accuracy in fp32; census comparison
    model).  want = ('avec','bvec','aexp','bexp','logz','post','qlog','amax','bmax').
    """
    scores, T, N, S, E, has_blank, blank = _decode_args(scores, n_base, state_len, blank_score)
    out = {"labels": np.empty((N, T), dtype=np.int8)}
    shapes = {"avec": (T + 1, N, S), "bvec": (T + 1, N, S), "logz": (N,), "post": (T, N, S * E),
              "qlog": (T, N, S * E), "amax": (T + 1, N, S), "bmax": (T + 1, N, S)}
    for k in want:
        if k in ("aexp", "bexp"):
            out[k] = np.zeros((T + 1, N), dtype=np.int32)
        else:
            out[k] = np.zeros(shapes[k], dtype=np.float32)
    rc = lib().xo_decode_scaled(_p(scores), C.c_int(T), C.c_int(N), C.c_int(n_base), C.c_int(state_len),
                                C.c_int(has_blank), C.c_float(blank), _p(out["labels"], i8p),
                                _p(out.get("avec")), _p(out.get("bvec")), _p(out.get("aexp"), i32p),
                                _p(out.get("bexp"), i32p), _p(out.get("logz")), _p(out.get("post")),
                                _p(out.get("qlog")), _p(out.get("amax")), _p(out.get("bmax")))
    if rc:
        raise MemoryError("xo_decode_scaled failed")
    return out


def decode_logdomain(scores, n_base, state_len, blank_score=None, libm=False, want=(), softmax=False):
    """
    NOT the contract: the plain log-domain fp32 evaluation (what a seqdist-style kernel computes), with the
    polynomial exp/log or libm's (softmax=True: libm with seqdist's per-time-step softmax normalisation of the
    posteriors).  Returns 'labels' and any of want = ('gap','logz','post'); 'gap' (N,T) is the
    winning max-marginal minus the best max-marginal of any edge carrying a different label.
    """
    scores, T, N, S, E, has_blank, blank = _decode_args(scores, n_base, state_len, blank_score)
    out = {"labels": np.empty((N, T), dtype=np.int8)}
    shapes = {"gap": (N, T), "logz": (N,), "post": (T, N, S * E)}
    for k in want:
        out[k] = np.zeros(shapes[k], dtype=np.float32)
    rc = lib().xo_decode_logdomain(_p(scores), C.c_int(T), C.c_int(N), C.c_int(n_base), C.c_int(state_len),
                                   C.c_int(has_blank), C.c_float(blank), C.c_int(2 if softmax else int(bool(libm))),
                                   _p(out["labels"], i8p), _p(out.get("gap")), _p(out.get("logz")),
                                   _p(out.get("post")))
    if rc:
        raise MemoryError("xo_decode_logdomain failed")
    return out


def pack(labels, alphabet):
    """labels (N,T) int8 -> (seq (N,T) int8 left-packed ASCII, qstring (N,T) int8, lens (N,) int32)."""
    labels = np.ascontiguousarray(labels, dtype=np.int8)
    N, T = labels.shape
    seq = np.empty((N, T), dtype=np.int8)
    qs = np.empty((N, T), dtype=np.int8)
    lens = np.empty((N,), dtype=np.int32)
    ab = "".join(alphabet).encode()
    lib().xo_pack(_p(labels, i8p), C.c_int(N), C.c_int(T), C.c_char_p(ab), _p(seq, i8p), _p(qs, i8p),
                  _p(lens, i32p))
    return seq, qs, lens


def decode_batch(scores, alphabet, state_len, blank_score=None):
    """SeqdistModel.decode_batch (crf/model.py:215-218): list of called strings, one per chunk."""
    nb = len(alphabet) - 1
    labels = decode(scores, nb, state_len, blank_score)["labels"]
    seq, _, lens = pack(labels, alphabet)
    return [seq[i, :lens[i]].tobytes().decode() for i in range(seq.shape[0])]


def conv1d_silu(x, w, b, stride, pad):
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    N, Cin, L = x.shape
    Cout, _, K = w.shape
    Lout = (L + 2 * pad - K) // stride + 1
    y = np.empty((N, Cout, Lout), dtype=np.float32)
    lib().xo_conv1d_silu(_p(x), C.c_int(N), C.c_int(Cin), C.c_int(L), _p(w), _p(b), C.c_int(Cout),
                         C.c_int(K), C.c_int(stride), C.c_int(pad), _p(y))
    return y


def lstm(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    x, w_ih, w_hh = _f32(x), _f32(w_ih), _f32(w_hh)
    b_ih = None if b_ih is None else _f32(b_ih)
    b_hh = None if b_hh is None else _f32(b_hh)
    T, N, I = x.shape
    H = w_hh.shape[1]
    y = np.empty((T, N, H), dtype=np.float32)
    rc = lib().xo_lstm(_p(x), C.c_int(T), C.c_int(N), C.c_int(I), C.c_int(H), _p(w_ih), _p(w_hh),
                       _p(b_ih), _p(b_hh), C.c_int(int(reverse)), _p(y))
    if rc:
        raise MemoryError("xo_lstm failed")
    return y


def linear_crf(x, w, b, scale, n_base, blank_score=None, expand_blanks=True):
    x, w, b = _f32(x), _f32(w), _f32(b)
    lead = x.shape[:-1]
    I = x.shape[-1]
    M = int(np.prod(lead))
    O = w.shape[0]
    expand = int(blank_score is not None and expand_blanks)
    Cout = O // n_base * (n_base + 1) if expand else O
    y = np.empty(lead + (Cout,), dtype=np.float32)
    rc = lib().xo_linear_crf(_p(x), C.c_int64(M), C.c_int(I), _p(w), _p(b), C.c_int(O),
                             C.c_float(scale), C.c_int(n_base), C.c_int(expand),
                             C.c_float(0.0 if blank_score is None else blank_score), _p(y))
    if rc:
        raise MemoryError("xo_linear_crf failed")
    return y


# state-dict keys of the 10-module inference encoder, in order
STATE_DICT_ORDER = (
    ["encoder.%d.conv.%s" % (i, p) for i in (0, 1, 2) for p in ("weight", "bias")]
    + ["encoder.%d.rnn.%s" % (i, p) for i in (4, 5, 6, 7, 8)
       for p in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    + ["encoder.9.linear.weight", "encoder.9.linear.bias"]
)


def state_dict_list(state_dict):
    return [_f32(np.asarray(state_dict[k])) for k in STATE_DICT_ORDER]


def encode(signal, state_dict, features, n_base, state_len, winlen=19, stride=5, scale=5.0,
           blank_score=2.0, expand_blanks=True, want_lstm_out=False):
    """signal (N,L) or (N,1,L) fp32 -> scores (T,N,C) fp32 (Model.forward, crf/model.py:212-213)."""
    signal = _f32(signal)
    if signal.ndim == 3:
        signal = signal[:, 0, :]
    signal = np.ascontiguousarray(signal)
    N, L = signal.shape
    ws = state_dict_list(state_dict)
    arr = (f32p * len(ws))(*[_p(w) for w in ws])
    T = (L + 2 * (winlen // 2) - winlen) // stride + 1
    S = n_base ** state_len
    expand = int(blank_score is not None and expand_blanks)
    Cout = S * (n_base + 1) if expand else S * n_base
    scores = np.empty((T, N, Cout), dtype=np.float32)
    lo = np.empty((T, N, features), dtype=np.float32) if want_lstm_out else None
    rc = lib().xo_encode(_p(signal), C.c_int(N), C.c_int(L), arr, C.c_int(features), C.c_int(winlen),
                         C.c_int(stride), C.c_int(n_base), C.c_int(state_len), C.c_float(scale),
                         C.c_float(0.0 if blank_score is None else blank_score), C.c_int(expand),
                         _p(scores), _p(lo))
    if rc:
        raise MemoryError("xo_encode failed")
    return (scores, lo) if want_lstm_out else scores


def ctc_indices(targets, n_base, state_len):
    """prepare_ctc_scores' gather indices (crf/model.py:102-116): (stay_idx (N, n), move_idx (N, n-1)), n = Lt - state_len + 1."""
    t = np.ascontiguousarray(targets, dtype=np.int32)
    N, Lt = t.shape
    n = Lt - (state_len - 1)
    stay = np.zeros((N, n), np.int32)
    move = np.zeros((N, max(n - 1, 0)), np.int32)
    lib().xo_ctc_indices(_p(t, i32p), C.c_int(N), C.c_int(Lt), C.c_int(n_base), C.c_int(state_len), _p(stay, i32p), _p(move, i32p))
    return stay, move


def ctc_logz(scores, targets, target_lengths, n_base, state_len, semiring="log", want_grads=False):
    """seqdist.ctc_simple logZ over the stay / move lattice of the targets (crf/model.py:118-135).  scores (T, N, C) with the
    blank column.  Returns {'logz': (N,)} plus, with want_grads, 'stay' (T, N, n) and 'move' (T, N, n-1): the restricted
    posteriors (Log) or, for semiring 'max', 'stay' = the one-hot Viterbi alignment (viterbi_alignments)."""
    sc = _f32(scores)
    T, N, Cc = sc.shape
    stay_idx, move_idx = ctc_indices(targets, n_base, state_len)
    n = stay_idx.shape[1]
    tl = np.ascontiguousarray(target_lengths, dtype=np.int32)
    logz = np.empty(N, np.float32)
    gs = np.zeros((T, N, n), np.float32) if want_grads else None
    gm = np.zeros((T, N, max(n - 1, 0)), np.float32) if (want_grads and semiring == "log") else None
    rc = lib().xo_ctc_logz(_p(sc), C.c_int(T), C.c_int(N), C.c_int(Cc), _p(stay_idx, i32p), _p(move_idx, i32p), C.c_int(n),
                           _p(tl, i32p), C.c_int(state_len), C.c_int(0 if semiring == "log" else 1), _p(logz), _p(gs), _p(gm))
    if rc:
        raise ValueError("xo_ctc_logz failed (%d): target_lengths outside [state_len, Lt]?" % rc)
    out = {"logz": logz}
    if want_grads:
        out["stay"] = gs
        if gm is not None:
            out["move"] = gm
    return out


def log_beam_cut(beam_cut):
    """log(beam_cut) as the float both sides use (FLT_MAX switches the cut off, as in the published decoder)."""
    import math
    return float(np.float32(math.log(beam_cut))) if beam_cut > 0 else float(np.finfo(np.float32).max)


def beam_search(scores, alphabet, state_len, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0, blank_score=None):
    """
    koi.decode.beam_search as called at crf/basecall.py:43-46, restated (xna_oracle.c: "CRF beam search"; parity unpinned).
    scores (T, N, C) fp32 with or without the blank column.  Returns dict: 'sequence', 'qstring' (N, T) int8 (ASCII at the
    emitting blocks, 0 elsewhere), 'moves' (N, T) uint8, 'score' (N,) float32.
    """
    n_base = len(alphabet) - 1
    scores, T, N, S, E, has_blank, blank = _decode_args(scores, n_base, state_len, blank_score)
    d = decode(scores, n_base, state_len, blank_score=blank_score, want=("alpha", "beta", "logz"))
    out = {"sequence": np.zeros((N, T), dtype=np.int8), "qstring": np.zeros((N, T), dtype=np.int8),
           "moves": np.zeros((N, T), dtype=np.uint8), "score": np.zeros((N,), dtype=np.float32)}
    rc = lib().xo_beam_search(_p(scores), C.c_int(T), C.c_int(N), C.c_int(n_base), C.c_int(state_len), C.c_int(has_blank),
                              C.c_float(blank), _p(d["alpha"]), _p(d["beta"]), _p(d["logz"]), C.c_int(int(beam_width)),
                              C.c_float(log_beam_cut(beam_cut)), C.c_float(scale), C.c_float(offset),
                              C.c_char_p(alphabet.encode("ascii")), _p(out["sequence"], i8p), _p(out["qstring"], i8p),
                              out["moves"].ctypes.data_as(C.POINTER(C.c_uint8)), _p(out["score"]))
    if rc:
        raise ValueError("xo_beam_search failed (%d)" % rc)
    return out
