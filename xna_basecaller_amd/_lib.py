"""
ctypes binding of libxnacall.so (include/xna_basecaller.h).  There is NO fallback: if the
library is missing or no gfx950 GPU is present the product path raises, it never computes on
the CPU (the CPU restatement lives in oracle/ and is test infrastructure only).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XNA_LIBXNACALL", os.path.join(_HERE, "libxnacall.so"))   # override: diagnostic builds only
_lib = None

XB_STAGE_NAMES = ("conv", "lstm_in", "lstm_rec", "linear", "decode")
XB_PREC_F16X3, XB_PREC_F16, XB_PREC_F16F8, XB_PREC_F16F8_IN1, XB_PREC_MIXED = 0, 1, 2, 3, 4
PRECISIONS = {"f16x3": XB_PREC_F16X3, "f16": XB_PREC_F16, "f16f8": XB_PREC_F16F8, "f16f8i": XB_PREC_F16F8_IN1,
              "mixed": XB_PREC_MIXED}

EXPORTS = [
    "xb_ctx_create", "xb_ctx_destroy", "xb_last_error", "xb_device_count", "xb_load_weights",
    "xb_weights_ready", "xb_encode", "xb_encode_dev", "xb_decode", "xb_decode_dev", "xb_crf_logz", "xb_crf_logz_dev", "xb_crf_scans", "xb_crf_scans_dev",
    "xb_basecall_chunks", "xb_basecall_chunks_dev", "xb_synchronize", "xb_set_profiling",
    "xb_get_stage_times", "xb_reset_stage_times", "xb_geometry", "xb_version", "xb_result_stream",
    "xb_submit_chunks", "xb_collect_chunks", "xb_ctc_logz", "xb_ctc_alignments",
    "xb_comm_unique_id", "xb_comm_create", "xb_comm_destroy", "xb_comm_rank", "xb_comm_world", "xb_comm_last_error",
    "xb_gather_called", "xb_comm_fence", "xb_comm_synchronize", "xb_stream_wait_event", "xb_align_accuracy",
    "xb_beam_search", "xb_beam_search_dev", "xb_basecall_chunks_beam", "xb_reserve_pairing", "xb_pairing_active", "xb_debug_layer_output",
]
XB_COMM_ID_BYTES = 128
# xb_status (include/xna_basecaller.h)
XB_OK, XB_ERR_INVALID, XB_ERR_HIP, XB_ERR_NOMEM, XB_ERR_STATE, XB_ERR_DEVICE, XB_ERR_NO_GPU = 0, -1, -2, -3, -4, -5, -6
XB_PIPELINE_SLOTS = 4          # include/xna_basecaller.h


class XbConfig(C.Structure):
    _fields_ = [("n_base", C.c_int32), ("state_len", C.c_int32), ("features", C.c_int32),
                ("winlen", C.c_int32), ("stride", C.c_int32), ("scale", C.c_float),
                ("blank_score", C.c_float), ("chunk_len", C.c_int32), ("max_batch", C.c_int32),
                ("precision", C.c_int32), ("lstm_mode", C.c_int32)]


class XbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libxnacall error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile the HIP sources for gfx950 (csrc/Makefile; hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(no CPU fallback exists for the MI355X path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    lib.xb_ctx_create.argtypes = [C.POINTER(vp), ip, C.POINTER(XbConfig)]
    lib.xb_ctx_destroy.argtypes = [vp]
    lib.xb_ctx_destroy.restype = None
    lib.xb_last_error.argtypes = [vp]
    lib.xb_last_error.restype = C.c_char_p
    lib.xb_version.restype = C.c_char_p
    lib.xb_load_weights.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    lib.xb_weights_ready.argtypes = [vp]
    lib.xb_encode.argtypes = [vp, vp, ip, ip, vp]
    lib.xb_encode_dev.argtypes = [vp, vp, ip, ip, vp]
    lib.xb_decode.argtypes = [vp, vp, ip, ip, ip, C.c_char_p, vp, vp, vp]
    lib.xb_decode_dev.argtypes = [vp, vp, ip, ip, ip, C.c_char_p, vp, vp, vp]
    lib.xb_crf_logz.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.xb_crf_logz_dev.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.xb_crf_scans.argtypes = [vp, vp, ip, ip, ip, vp, vp, vp, vp]
    lib.xb_crf_scans_dev.argtypes = [vp, vp, ip, ip, ip, vp, vp, vp, vp]
    lib.xb_basecall_chunks.argtypes = [vp, vp, ip, C.c_char_p, vp, vp]
    lib.xb_basecall_chunks_dev.argtypes = [vp, vp, ip, C.c_char_p, vp, vp]
    lib.xb_synchronize.argtypes = [vp]
    lib.xb_reserve_pairing.argtypes = [vp]
    lib.xb_pairing_active.argtypes = [vp]
    lib.xb_comm_unique_id.argtypes = [C.c_char_p]
    lib.xb_comm_create.argtypes = [C.POINTER(vp), ip, ip, ip, C.c_char_p]
    lib.xb_comm_destroy.argtypes = [vp]
    lib.xb_comm_destroy.restype = None
    lib.xb_comm_rank.argtypes = [vp]
    lib.xb_comm_world.argtypes = [vp]
    lib.xb_comm_last_error.argtypes = [vp]
    lib.xb_comm_last_error.restype = C.c_char_p
    lib.xb_gather_called.argtypes = [vp, vp, vp, vp, ip, ip, vp, vp]
    lib.xb_comm_fence.argtypes = [vp, vp, ip]
    lib.xb_comm_synchronize.argtypes = [vp]
    lib.xb_stream_wait_event.argtypes = [vp, vp]
    lib.xb_align_accuracy.argtypes = [C.c_char_p, ip, C.c_char_p, ip, C.c_double, ip, C.POINTER(C.c_double), vp]
    lib.xb_ctc_logz.argtypes = [vp, vp, ip, ip, vp, ip, vp, vp, vp, vp]
    fl = C.c_float
    lib.xb_beam_search.argtypes = [vp, vp, ip, ip, ip, C.c_char_p, ip, fl, fl, fl, vp, vp, vp, vp]
    lib.xb_beam_search_dev.argtypes = [vp, vp, ip, ip, ip, C.c_char_p, ip, fl, fl, fl, vp, vp, vp, vp]
    lib.xb_basecall_chunks_beam.argtypes = [vp, vp, ip, C.c_char_p, ip, fl, fl, fl, vp, vp, vp, vp]
    lib.xb_ctc_alignments.argtypes = [vp, vp, ip, ip, vp, ip, vp, vp, vp]
    lib.xb_submit_chunks.argtypes = [vp, ip, vp, ip, C.c_char_p]
    lib.xb_collect_chunks.argtypes = [vp, ip, vp, vp]
    lib.xb_result_stream.argtypes = [vp]
    lib.xb_result_stream.restype = C.c_void_p
    lib.xb_set_profiling.argtypes = [vp, ip]
    lib.xb_get_stage_times.argtypes = [vp, vp, vp]
    lib.xb_reset_stage_times.argtypes = [vp]
    lib.xb_geometry.argtypes = [vp, C.POINTER(ip), C.POINTER(ip), C.POINTER(ip), C.POINTER(ip)]
    lib.xb_debug_layer_output.argtypes = [vp, ip, ip, vp, vp]
    _lib = lib
    return lib


def align_accuracy(ref, seq, balanced=False, min_coverage=0.0, want_counts=False):
    """util.accuracy (util.py:402-424) through xb_align_accuracy: percent identity of the local alignment, 0 below min_coverage."""
    r, q = ref.encode("ascii"), seq.encode("ascii")
    acc = C.c_double()
    counts = (C.c_int32 * 4)()
    rc = load().xb_align_accuracy(r, len(r), q, len(q), float(min_coverage), int(bool(balanced)), C.byref(acc), counts)
    if rc:
        raise XbError(rc, "xb_align_accuracy: bad argument")
    return (acc.value, dict(zip("=XID", counts))) if want_counts else acc.value


def source_digest():
    """sha1 (12 hex digits) over the library's sources (csrc/*.hip, *.h, Makefile and the public header): what bench.py and
    tools/hbm_traffic.py record so that a counter profile is only ever quoted for the code it was collected on."""
    import glob
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile")])
    files.append(os.path.join(os.path.dirname(_HERE), "include", "xna_basecaller.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def device_count():
    return int(load().xb_device_count())


def require_gpu():
    if device_count() < 1:
        raise RuntimeError("no HIP device visible: the xna_basecaller_amd hot path runs on MI355X (gfx950) only")


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a)          # raw device pointer (e.g. torch.Tensor.data_ptr())


class Context:
    """One xb_ctx: a GPU, a stream, the device copies of the weights and all workspaces."""

    def __init__(self, device, n_base, state_len, features, winlen, stride, scale, blank_score,
                 chunk_len, max_batch, precision=XB_PREC_F16X3, lstm_mode=0):
        self.lib = load()
        self.cfg = XbConfig(n_base, state_len, features, winlen, stride, scale, blank_score, chunk_len,
                            max_batch, precision, lstm_mode)
        h = C.c_void_p()
        rc = self.lib.xb_ctx_create(C.byref(h), int(device), C.byref(self.cfg))
        if rc:
            raise XbError(rc, (self.lib.xb_last_error(None) or b"").decode())
        self.h = h
        T, S, Cb, Cn = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.lib.xb_geometry(self.h, C.byref(T), C.byref(S), C.byref(Cb), C.byref(Cn))
        self.T, self.S, self.C_blank, self.C_noblank = T.value, S.value, Cb.value, Cn.value
        self.n_base, self.chunk_len, self.max_batch = n_base, chunk_len, max_batch

    def _check(self, rc):
        if rc:
            raise XbError(rc, (self.lib.xb_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.xb_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, state_dict):
        """state_dict: name -> fp32 array in PyTorch layout (the 28 inference-encoder tensors)."""
        for k, v in state_dict.items():
            a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
            self._check(self.lib.xb_load_weights(self.h, k.encode(), a.ctypes.data, a.size))
        self._check(self.lib.xb_weights_ready(self.h))

    def debug_layer_output(self, which, n):
        """(hi, second) uint16 arrays (T, n, features) of LSTM layer 3 (which = 0) / 4 (1) after the last encode of n chunks."""
        F = self.cfg.features
        hi = np.empty((self.T, n, F), dtype=np.uint16)
        second = np.empty((self.T, n, F), dtype=np.uint16)
        self._check(self.lib.xb_debug_layer_output(self.h, int(which), int(n), hi.ctypes.data, second.ctypes.data))
        return hi, second

    # ---- host-buffer operators ---------------------------------------------------------
    def encode(self, signal, expand_blanks=True):
        signal = np.ascontiguousarray(signal, dtype=np.float32).reshape(-1, self.chunk_len)
        n = signal.shape[0]
        scores = np.empty((self.T, n, self.C_blank if expand_blanks else self.C_noblank), dtype=np.float32)
        self._check(self.lib.xb_encode(self.h, signal.ctypes.data, n, int(bool(expand_blanks)), scores.ctypes.data))
        return scores

    def decode(self, scores, alphabet, has_blank=None, want_labels=False):
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        T, n, Cin = scores.shape
        if has_blank is None:
            has_blank = Cin == self.C_blank
        if Cin != (self.C_blank if has_blank else self.C_noblank):
            raise ValueError("scores last dim %d does not match the model (%d with blanks, %d without)"
                             % (Cin, self.C_blank, self.C_noblank))
        labels = np.empty((n, T), dtype=np.int8) if want_labels else None
        seq = np.empty((n, T), dtype=np.int8)
        lens = np.empty((n,), dtype=np.int32)
        self._check(self.lib.xb_decode(self.h, scores.ctypes.data, T, n, int(bool(has_blank)),
                                       "".join(alphabet).encode(), _ptr(labels), seq.ctypes.data, lens.ctypes.data))
        return (seq, lens, labels) if want_labels else (seq, lens)

    def crf_logz(self, scores, has_blank=None):
        """(T, n, C) scores -> (n,) fp32 log partition function (CTC_CRF.logZ, crf/model.py:41-46)."""
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        T, n, Cin = scores.shape
        if has_blank is None:
            has_blank = Cin == self.C_blank
        if Cin != (self.C_blank if has_blank else self.C_noblank):
            raise ValueError("scores last dim %d does not match the model (%d with blanks, %d without)"
                             % (Cin, self.C_blank, self.C_noblank))
        logz = np.empty((n,), dtype=np.float32)
        self._check(self.lib.xb_crf_logz(self.h, scores.ctypes.data, T, n, int(bool(has_blank)), logz.ctypes.data))
        return logz

    def crf_scans(self, scores, want=("alpha", "beta", "logz", "post"), has_blank=None):
        """(T, n, C) scores -> dict of the Log scans (xb_crf_scans): 'alpha', 'beta' (T+1, n, S), 'logz' (n,),
        'post' (T, n, S*(n_base+1))."""
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        T, n, Cin = scores.shape
        if has_blank is None:
            has_blank = Cin == self.C_blank
        if Cin != (self.C_blank if has_blank else self.C_noblank):
            raise ValueError("scores last dim %d does not match the model (%d with blanks, %d without)"
                             % (Cin, self.C_blank, self.C_noblank))
        S = self.C_blank // (self.n_base + 1)
        shapes = {"alpha": (T + 1, n, S), "beta": (T + 1, n, S), "logz": (n,), "post": (T, n, self.C_blank)}
        out = {k: np.empty(shapes[k], dtype=np.float32) for k in want}
        self._check(self.lib.xb_crf_scans(self.h, scores.ctypes.data, T, n, int(bool(has_blank)), _ptr(out.get("alpha")),
                                          _ptr(out.get("beta")), _ptr(out.get("logz")), _ptr(out.get("post"))))
        return out

    def ctc_logz(self, scores, targets, target_lengths, want_grads=False):
        """xb_ctc_logz: scores (T, n, C_blank), targets (n, Lt) CTC labels, target_lengths (n) -> {'logz': (n,)} and, with
        want_grads, 'stay' (T, n, np) / 'move' (T, n, np - 1), np = Lt - state_len + 1 (the restricted posteriors)."""
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        targets = np.ascontiguousarray(targets, dtype=np.int32)
        tl = np.ascontiguousarray(target_lengths, dtype=np.int32)
        T, n, Cin = scores.shape
        if Cin != self.C_blank or targets.shape[0] != n or tl.shape != (n,):
            raise ValueError("ctc_logz: scores (T, n, %d), targets (n, Lt), target_lengths (n) expected" % self.C_blank)
        Lt = targets.shape[1]
        npos = Lt - (self.cfg.state_len - 1)
        out = {"logz": np.empty((n,), np.float32)}
        if want_grads:
            out["stay"] = np.empty((T, n, max(npos, 0)), np.float32)
            out["move"] = np.empty((T, n, max(npos - 1, 0)), np.float32)
        self._check(self.lib.xb_ctc_logz(self.h, scores.ctypes.data, T, n, targets.ctypes.data, Lt, tl.ctypes.data,
                                         out["logz"].ctypes.data, _ptr(out.get("stay")), _ptr(out.get("move"))))
        return out

    def ctc_alignments(self, scores, targets, target_lengths):
        """xb_ctc_alignments: (alignments (T, n, np) one-hot over target positions, max path score (n,))."""
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        targets = np.ascontiguousarray(targets, dtype=np.int32)
        tl = np.ascontiguousarray(target_lengths, dtype=np.int32)
        T, n, Cin = scores.shape
        if Cin != self.C_blank or targets.shape[0] != n or tl.shape != (n,):
            raise ValueError("ctc_alignments: scores (T, n, %d), targets (n, Lt), target_lengths (n) expected" % self.C_blank)
        Lt = targets.shape[1]
        npos = Lt - (self.cfg.state_len - 1)
        al = np.empty((T, n, max(npos, 0)), np.float32)
        best = np.empty((n,), np.float32)
        self._check(self.lib.xb_ctc_alignments(self.h, scores.ctypes.data, T, n, targets.ctypes.data, Lt, tl.ctypes.data,
                                               al.ctypes.data, best.ctypes.data))
        return al, best

    def crf_scans_dev(self, d_scores, T, n, has_blank, d_alpha=None, d_beta=None, d_logz=None, d_post=None):
        self._check(self.lib.xb_crf_scans_dev(self.h, _ptr(d_scores), T, n, int(bool(has_blank)), _ptr(d_alpha), _ptr(d_beta),
                                              _ptr(d_logz), _ptr(d_post)))

    def crf_logz_dev(self, d_scores, T, n, has_blank, d_logz):
        self._check(self.lib.xb_crf_logz_dev(self.h, _ptr(d_scores), T, n, int(bool(has_blank)), _ptr(d_logz)))

    def basecall_chunks(self, signal, alphabet):
        signal = np.ascontiguousarray(signal, dtype=np.float32).reshape(-1, self.chunk_len)
        n = signal.shape[0]
        seq = np.empty((n, self.T), dtype=np.int8)
        lens = np.empty((n,), dtype=np.int32)
        self._check(self.lib.xb_basecall_chunks(self.h, signal.ctypes.data, n, "".join(alphabet).encode(),
                                                seq.ctypes.data, lens.ctypes.data))
        return seq, lens

    # ---- beam search with qualities and moves (koi.decode.beam_search at crf/basecall.py:43-46) ----------
    def _beam_out(self, n, T):
        return (np.empty((n, T), dtype=np.int8), np.empty((n, T), dtype=np.int8), np.empty((n, T), dtype=np.uint8),
                np.empty((n,), dtype=np.float32))

    def beam_search(self, scores, alphabet, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0):
        """xb_beam_search: scores (T, n, C_noblank | C_blank) -> {'sequence', 'qstring' (n, T) int8, 'moves' (n, T) uint8,
        'score' (n,)}; the stay score of blank-less scores is the context's blank_score."""
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        T, n, Cin = scores.shape
        if Cin not in (self.C_blank, self.C_noblank):
            raise ValueError("scores last dim %d matches neither %d nor %d" % (Cin, self.C_blank, self.C_noblank))
        seq, q, mv, sc = self._beam_out(n, T)
        self._check(self.lib.xb_beam_search(self.h, scores.ctypes.data, T, n, int(Cin == self.C_blank), "".join(alphabet).encode(),
                                            int(beam_width), float(beam_cut), float(scale), float(offset), seq.ctypes.data,
                                            q.ctypes.data, mv.ctypes.data, sc.ctypes.data))
        return {"sequence": seq, "qstring": q, "moves": mv, "score": sc}

    def basecall_chunks_beam(self, signal, alphabet, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0):
        """xb_basecall_chunks_beam: signal (n, chunk_len) -> the same dict as beam_search, scores never leave the device."""
        signal = np.ascontiguousarray(signal, dtype=np.float32).reshape(-1, self.chunk_len)
        n = signal.shape[0]
        seq, q, mv, sc = self._beam_out(n, self.T)
        self._check(self.lib.xb_basecall_chunks_beam(self.h, signal.ctypes.data, n, "".join(alphabet).encode(), int(beam_width),
                                                     float(beam_cut), float(scale), float(offset), seq.ctypes.data, q.ctypes.data,
                                                     mv.ctypes.data, sc.ctypes.data))
        return {"sequence": seq, "qstring": q, "moves": mv, "score": sc}

    def beam_search_dev(self, d_scores, T, n, has_blank, alphabet, d_sequence, d_qstring, d_moves, d_score=None, beam_width=32,
                        beam_cut=100.0, scale=1.0, offset=0.0):
        self._check(self.lib.xb_beam_search_dev(self.h, _ptr(d_scores), int(T), int(n), int(bool(has_blank)),
                                                "".join(alphabet).encode(), int(beam_width), float(beam_cut), float(scale),
                                                float(offset), _ptr(d_sequence), _ptr(d_qstring), _ptr(d_moves), _ptr(d_score)))

    # ---- host pipeline: two batches in flight (xb_submit_chunks / xb_collect_chunks) ----------
    def submit_chunks(self, slot, signal, alphabet):
        signal = np.ascontiguousarray(signal, dtype=np.float32).reshape(-1, self.chunk_len)
        self._check(self.lib.xb_submit_chunks(self.h, int(slot), signal.ctypes.data, signal.shape[0],
                                              "".join(alphabet).encode()))
        return signal.shape[0]

    def collect_chunks(self, slot, n):
        seq = np.empty((n, self.T), dtype=np.int8)
        lens = np.empty((n,), dtype=np.int32)
        self._check(self.lib.xb_collect_chunks(self.h, int(slot), seq.ctypes.data, lens.ctypes.data))
        return seq, lens

    # ---- device-pointer operators (pointers are ints, e.g. torch data_ptr()) -----------------
    def encode_dev(self, d_signal, n, expand_blanks, d_scores):
        self._check(self.lib.xb_encode_dev(self.h, _ptr(d_signal), n, int(bool(expand_blanks)), _ptr(d_scores)))

    def decode_dev(self, d_scores, T, n, has_blank, alphabet, d_labels, d_seq, d_len):
        ab = None if alphabet is None else "".join(alphabet).encode()
        self._check(self.lib.xb_decode_dev(self.h, _ptr(d_scores), T, n, int(bool(has_blank)), ab,
                                           _ptr(d_labels), _ptr(d_seq), _ptr(d_len)))

    def basecall_chunks_dev(self, d_signal, n, alphabet, d_seq, d_len):
        self._check(self.lib.xb_basecall_chunks_dev(self.h, _ptr(d_signal), n, "".join(alphabet).encode(),
                                                    _ptr(d_seq), _ptr(d_len)))

    def synchronize(self):
        self._check(self.lib.xb_synchronize(self.h))

    def reserve_pairing(self):
        """Opt in to the co-scheduling of two asynchronous calls in flight (xb_reserve_pairing: allocates the workspaces for a
        pair); returns whether the context pairs calls from now on."""
        self._check(self.lib.xb_reserve_pairing(self.h))
        return self.pairing_active()

    def pairing_active(self):
        return bool(self.lib.xb_pairing_active(self.h))

    def result_stream(self):
        """hipStream_t (int) producing the outputs of the most recent *_dev call (xb_result_stream)."""
        return int(self.lib.xb_result_stream(self.h) or 0)

    def set_profiling(self, on):
        self._check(self.lib.xb_set_profiling(self.h, int(bool(on))))

    def reset_stage_times(self):
        self._check(self.lib.xb_reset_stage_times(self.h))

    def stage_times(self):
        ms = (C.c_float * 5)()
        ln = (C.c_int64 * 5)()
        self._check(self.lib.xb_get_stage_times(self.h, ms, ln))
        return {k: (float(ms[i]), int(ln[i])) for i, k in enumerate(XB_STAGE_NAMES)}


class Comm:
    """xb_comm: the RCCL communicator of the path's one collective (include/xna_basecaller.h, 'multi-GPU').  Rank 0 draws
    the id with Comm.unique_id() and passes it to the other ranks out of band (dist.exchange_comm_id)."""

    def __init__(self, device, rank, world, comm_id):
        self.lib = load()
        if len(comm_id) != XB_COMM_ID_BYTES:
            raise ValueError("communicator id must be %d bytes" % XB_COMM_ID_BYTES)
        h = C.c_void_p()
        rc = self.lib.xb_comm_create(C.byref(h), int(device), int(rank), int(world), bytes(comm_id))
        if rc:
            raise XbError(rc, (self.lib.xb_comm_last_error(None) or b"").decode())
        self.h, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def unique_id():
        lib = load()
        buf = C.create_string_buffer(XB_COMM_ID_BYTES)
        rc = lib.xb_comm_unique_id(buf)
        if rc:
            raise XbError(rc, (lib.xb_comm_last_error(None) or b"").decode())
        return buf.raw

    def _check(self, rc):
        if rc:
            raise XbError(rc, (self.lib.xb_comm_last_error(self.h) or b"").decode())

    def gather_called(self, ctx, d_seq, d_len, n, T, d_all_seq, d_all_len):
        """All-gather of one batch on the communicator's stream, behind ctx's result stream (device pointers as ints)."""
        self._check(self.lib.xb_gather_called(self.h, ctx.h if ctx is not None else None, _ptr(d_seq), _ptr(d_len), int(n), int(T),
                                              _ptr(d_all_seq), _ptr(d_all_len)))

    def fence(self, ctx, lag=0):
        self._check(self.lib.xb_comm_fence(self.h, ctx.h, int(lag)))

    def synchronize(self):
        self._check(self.lib.xb_comm_synchronize(self.h))

    def close(self):
        if self.h:
            self.lib.xb_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
