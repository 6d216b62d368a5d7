"""
CTC-CRF model with the reference's plugin interface (ub-bonito/bonito/crf/model.py:24-237,
ub-bonito/bonito/nn.py): `Model(config)` exposes .encoder (indexable; [-1].expand_blanks /
.blank_score), .seqdist (.n_base .alphabet .state_len .idx .reverse_complement), .stride,
.alphabet, .config, state_dict()/load_state_dict()/half()/eval()/to()/parameters(),
`model(batch) -> scores (T,N,C)` and `decode_batch(scores) -> list[str]`.

torch is used ONLY as the checkpoint container (parameter names/shapes identical to the
reference so that util.load_model/match_names work unchanged).  No torch op runs in forward or
decode: both call the hand-written HIP kernels through the C ABI (include/xna_basecaller.h).
"""
import os
import sys
import weakref

import numpy as np
import torch

from .. import _lib


class CTC_CRF:
    """State/edge bookkeeping of the CRF (crf/model.py:24-46,78-100); arithmetic is on the GPU."""

    def __init__(self, state_len, alphabet):
        self.alphabet = alphabet
        self.state_len = state_len
        self.n_base = len(alphabet[1:])
        nb, S = self.n_base, self.n_base ** state_len
        idx = np.empty((S, nb + 1), dtype=np.int32)
        idx[:, 0] = np.arange(S)
        # in-edge k >= 1 of state j comes from the state that had base k-1 in front of j's first sl-1 bases
        idx[:, 1:] = (np.arange(nb)[None, :] * nb ** (state_len - 1) + (np.arange(S) // nb)[:, None])
        self.idx = torch.from_numpy(idx)

    def n_score(self):
        return len(self.alphabet) * self.n_base ** self.state_len

    def path_to_str(self, path):
        alphabet = np.frombuffer("".join(self.alphabet).encode(), dtype="u1")
        path = np.asarray(path)
        return alphabet[path[path != 0]].tobytes().decode()

    def compute_transition_probs(self, scores, betas):
        """
        (T,N,C) scores with the blank column and (T+1,N,S) Log backward scores -> (trans_probs (T,N,S,nb+1),
        init_state_probs (N,S)): per source state the probabilities of staying and of emitting each base, the input of
        the reference's beam-search branch (crf/model.py:62-76, crf/basecall.py:33-46).  Edge (j, k >= 1) of destination j
        leaves source idx[j, k] and emits base j % nb; in the (old_state, emitted_base) layout that is
        [src, 1 + base] with src = (k - 1) * hi + j // nb.
        """
        scores = np.asarray(scores, dtype=np.float32)
        betas = np.asarray(betas, dtype=np.float32)
        T, N, _ = scores.shape
        nb, S = self.n_base, self.n_base ** self.state_len
        lp = scores.reshape(T, N, S, nb + 1) + betas[1:, :, :, None]
        moves = lp[..., 1:].transpose(0, 1, 3, 2).reshape(T, N, S, nb)       # (new_state, dropped_base) -> (old_state, emitted_base)
        lp = np.concatenate([lp[..., :1], moves], axis=-1)
        lp = lp - lp.max(axis=-1, keepdims=True)
        tp = np.exp(lp)
        tp /= tp.sum(axis=-1, keepdims=True)
        b0 = betas[0] - betas[0].max(axis=-1, keepdims=True)
        ip = np.exp(b0)
        ip /= ip.sum(axis=-1, keepdims=True)
        return tp, ip

    # ---- CTC-CRF loss scans (crf/model.py:102-135): the lattice arithmetic runs on the device (xb_ctc_logz /
    #      xb_ctc_alignments through the owning Model's context); gathers and scatters of a few MB stay on the host ----
    def _ctc_indices(self, targets):
        """Gather columns of prepare_ctc_scores: (stay_idx (N, n), move_idx (N, n - 1)), n = Lt - state_len + 1."""
        t = np.clip(np.asarray(targets, dtype=np.int64) - 1, 0, None)          # CTC labels (blank = 0) -> zero-based bases
        nb, sl = self.n_base, self.state_len
        n = t.shape[1] - (sl - 1)
        stay = sum(t[:, i:n + i] * nb ** (sl - i - 1) for i in range(sl)) * len(self.alphabet)
        move = stay[:, 1:] + t[:, :n - 1] + 1
        return stay, move

    def prepare_ctc_scores(self, scores, targets):
        """(T,N,C) scores, (N,Lt) targets -> (stay_scores (T,N,n), move_scores (T,N,n-1)) (crf/model.py:102-116)."""
        scores = np.asarray(scores, dtype=np.float32)
        stay, move = self._ctc_indices(targets)
        T = scores.shape[0]
        return (np.take_along_axis(scores, np.broadcast_to(stay[None], (T,) + stay.shape), axis=2),
                np.take_along_axis(scores, np.broadcast_to(move[None], (T,) + move.shape), axis=2))

    def _owner(self):
        m = getattr(self, "_model", None)
        m = m() if m is not None else None
        if m is None:
            raise RuntimeError("the CTC scans run on the device: use the CTC_CRF of a Model (model.seqdist)")
        return m

    def ctc_loss(self, scores, targets, target_lengths, loss_clip=None, reduction="mean", normalise_scores=True,
                 want_grad=False):
        """
        crf/model.py:118-131: loss = -logZ_ctc(normalised scores) / target_length, clipped, mean / none.
        want_grad=True also returns d loss / d scores (T,N,C) -- what the reference obtains by autograd through seqdist:
        -(R - P) / length per chunk, with R the restricted posteriors scattered over (stay_idx, move_idx) and, when the
        scores were normalised here, P = the CRF's edge posteriors (every time step carries exactly one edge of a path,
        so the normalisation's own derivative is -P); zero where the loss was clipped, / N for the mean.
        """
        m = self._owner()
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        T, N, _ = scores.shape
        tl = np.asarray(target_lengths, dtype=np.int32)
        x = m.normalise(scores) if normalise_scores else scores
        ctx = m.context(T * m.stride, N)
        out = ctx.ctc_logz(x, targets, tl, want_grads=want_grad)
        loss = -(out["logz"] / tl.astype(np.float32))
        live = np.ones(N, dtype=bool)
        if loss_clip:
            live = (loss >= 0.0) & (loss <= loss_clip)
            loss = np.clip(loss, 0.0, np.float32(loss_clip))
        if reduction == "mean":
            value = loss.mean(dtype=np.float32)
        elif reduction in ("none", None):
            value = loss
        else:
            raise ValueError("Unknown reduction type {}".format(reduction))
        if not want_grad:
            return value
        stay, move = self._ctc_indices(targets)
        grad = np.zeros_like(scores)
        bi = np.arange(N)[:, None]
        for t in range(T):                                   # repeated k-mers hit the same column: unbuffered adds
            np.add.at(grad[t], (bi, stay), out["stay"][t])
            np.add.at(grad[t], (bi, move), out["move"][t])
        if normalise_scores:
            grad -= m.posteriors(scores)
        w = -(live / tl.astype(np.float32)) / (np.float32(N) if reduction == "mean" else np.float32(1))
        return value, grad * w[None, :, None].astype(np.float32)

    def ctc_viterbi_alignments(self, scores, targets, target_lengths):
        """crf/model.py:133-135: (T,N,n) one-hot alignment of the best path through the targets' stay / move lattice."""
        m = self._owner()
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        T, N, _ = scores.shape
        al, _ = m.context(T * m.stride, N).ctc_alignments(scores, targets, np.asarray(target_lengths, dtype=np.int32))
        return al

    def reverse_complement(self, scores):
        """crf/model.py:78-90 on a host (T,N,C) array: flip time, complement every k-mer index."""
        scores = np.asarray(scores)
        T, N, C = scores.shape
        nb, sl = self.n_base, self.state_len
        x = scores.reshape(T, N, *([nb] * sl), nb + 1)
        blanks = x[..., 0].transpose(0, 1, *range(sl + 1, 1, -1)).reshape(T, N, -1, 1)[::-1, :, ::-1]
        em = x[..., 1:].transpose(0, 1, *range(sl, 1, -1), sl + 2, sl + 1).reshape(T, N, -1, nb)[::-1, :, ::-1, ::-1]
        return np.ascontiguousarray(np.concatenate([blanks, em], axis=-1).reshape(T, N, -1))


class Convolution(torch.nn.Module):
    """Parameter holder for nn.py:57-84 (keys conv.weight / conv.bias)."""

    def __init__(self, insize, size, winlen, stride=1, padding=0, bias=True, activation=None):
        super().__init__()
        if activation != "swish" or not bias:
            raise NotImplementedError("the MI355X path implements Conv1d + bias + swish (rnn_encoder's form)")
        self.conv = torch.nn.Conv1d(insize, size, winlen, stride=stride, padding=padding, bias=bias)
        self.stride = stride


class Permute(torch.nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.dims = dims


class LSTM(torch.nn.Module):
    """Parameter holder for nn.py:176-235 (keys rnn.{weight_ih_l0,weight_hh_l0,bias_ih_l0,bias_hh_l0})."""

    def __init__(self, size, insize, bias=True, reverse=False):
        super().__init__()
        self.rnn = torch.nn.LSTM(size, insize, bias=bias)
        self.reverse = reverse
        with torch.no_grad():
            self.rnn.bias_hh_l0.zero_()
        self.rnn.bias_hh_l0.requires_grad = False


class LinearCRFEncoder(torch.nn.Module):
    """Parameter holder for nn.py:87-153 (keys linear.weight / linear.bias)."""

    def __init__(self, insize, n_base, state_len, bias=True, scale=None, activation=None, blank_score=None,
                 expand_blanks=True, extra_linear=False, drop_rate=0):
        super().__init__()
        if extra_linear:
            raise NotImplementedError("extra_linear is a training-time experiment; not on the MI355X path")
        if activation != "tanh" or scale is None or blank_score is None:
            raise NotImplementedError("the MI355X path implements scale*tanh(Wx+b) with a fixed blank_score")
        self.scale, self.n_base, self.state_len = scale, n_base, state_len
        self.blank_score, self.expand_blanks = blank_score, expand_blanks
        self.linear = torch.nn.Linear(insize, n_base ** (state_len + 1), bias=bias)


def rnn_encoder(n_base, state_len, insize=1, stride=5, winlen=19, activation="swish", rnn_type="lstm",
                features=768, scale=5.0, blank_score=None, expand_blanks=True, extra_linear=False, drop_rate=0,
                drop_rate_bottom=0):
    """Layer list of crf/model.py:142-160 (inference form: dropout layers are identities and are not built)."""
    if rnn_type != "lstm" or insize != 1:
        raise NotImplementedError("only insize=1, rnn_type='lstm' encoders are implemented")
    return torch.nn.Sequential(
        Convolution(insize, 4, 5, padding=2, activation=activation),
        Convolution(4, 16, 5, padding=2, activation=activation),
        Convolution(16, features, winlen, stride=stride, padding=winlen // 2, activation=activation),
        Permute([2, 0, 1]),
        LSTM(features, features, reverse=True), LSTM(features, features),
        LSTM(features, features, reverse=True), LSTM(features, features),
        LSTM(features, features, reverse=True),
        LinearCRFEncoder(features, n_base, state_len, activation="tanh", scale=scale, blank_score=blank_score,
                         expand_blanks=expand_blanks, extra_linear=extra_linear, drop_rate=drop_rate),
    )


def encoder_from_dict(spec, n_base, state_len):
    """
    New-style typed encoder config (crf/model.py:231-232, nn.py:244-259 `from_dict`): {'type': 'serial', 'sublayers':
    [{'type': 'convolution', ...} x3, {'type': 'permute', 'dims': [2, 0, 1]}, {'type': 'lstm', ...} x5,
    {'type': 'linearcrfencoder', ...}]} -- what `to_dict` writes for the rnn_encoder architecture.  Any other layer
    list is outside the hot path (the kernels implement exactly this stack) and is rejected with the reason.
    Returns (encoder, geometry) with geometry = dict(features, winlen, stride, scale, blank_score).
    """
    def bad(why):
        raise NotImplementedError("typed encoder config is not the rnn_encoder stack the MI355X path implements: " + why)

    if spec.get("type") != "serial" or not isinstance(spec.get("sublayers"), list):
        bad("top level must be {'type': 'serial', 'sublayers': [...]}")
    subs = [dict(d) for d in spec["sublayers"] if d.get("type") != "dropout"]      # inference: dropout is the identity
    kinds = [d.get("type") for d in subs]
    if kinds != ["convolution"] * 3 + ["permute"] + ["lstm"] * 5 + ["linearcrfencoder"]:
        bad("layer types %s" % kinds)
    c1, c2, c3 = subs[0:3]
    feats = int(c3["size"])
    want = [dict(insize=1, size=4, winlen=5, stride=1, padding=2), dict(insize=4, size=16, winlen=5, stride=1, padding=2),
            dict(insize=16, size=feats, winlen=int(c3["winlen"]), stride=int(c3["stride"]), padding=int(c3["winlen"]) // 2)]
    for i, (c, w) in enumerate(zip((c1, c2, c3), want)):
        got = {k: int(c.get(k, {"stride": 1, "padding": 0}.get(k, -1))) for k in w}
        if got != w or c.get("activation") != "swish" or not c.get("bias", True):
            bad("convolution %d is %s" % (i, c))
    if list(subs[3].get("dims", [])) != [2, 0, 1]:
        bad("permute dims %s" % subs[3].get("dims"))
    for i, l in enumerate(subs[4:9]):
        if int(l["size"]) != feats or int(l["insize"]) != feats or not l.get("bias", True) or bool(l.get("reverse", False)) != (i % 2 == 0):
            bad("lstm %d is %s (expected size %d, directions reverse,forward,reverse,forward,reverse)" % (i, l, feats))
    lin = subs[9]
    if int(lin["insize"]) != feats or int(lin["n_base"]) != n_base or int(lin["state_len"]) != state_len:
        bad("linearcrfencoder %s does not match labels / state_len" % lin)
    enc = rnn_encoder(n_base, state_len, insize=1, stride=int(c3["stride"]), winlen=int(c3["winlen"]), activation="swish",
                      features=feats, scale=lin.get("scale"), blank_score=lin.get("blank_score"),
                      expand_blanks=lin.get("expand_blanks", True))
    if lin.get("activation") != "tanh" or not lin.get("bias", True):
        bad("linearcrfencoder %s" % lin)
    return enc, dict(features=feats, winlen=int(c3["winlen"]), stride=int(c3["stride"]))


def _device_index(device):
    if isinstance(device, int):
        return device
    s = str(device)
    if s == "cpu":
        raise RuntimeError("xna_basecaller_amd has no CPU path: run with --device cuda[:N] on an MI355X "
                           "(the CPU restatement in oracle/ is test infrastructure only)")
    return int(s.split(":")[1]) if ":" in s else 0


class Model(torch.nn.Module):

    def __init__(self, config):
        super().__init__()
        self.seqdist = CTC_CRF(state_len=config["global_norm"]["state_len"], alphabet=config["labels"]["labels"])
        self.seqdist._model = weakref.ref(self)        # the CTC scans of model.seqdist run on this model's device context
        if "type" in config["encoder"]:          # new-style (typed) config, crf/model.py:231-232
            self.encoder, enc = encoder_from_dict(config["encoder"], self.seqdist.n_base, self.seqdist.state_len)
        else:                                      # old-style: keyword arguments of rnn_encoder
            enc = {k: v for k, v in config["encoder"].items()}
            self.encoder = rnn_encoder(self.seqdist.n_base, self.seqdist.state_len,
                                       insize=config["input"]["features"], **enc)
        self.stride = enc.get("stride", 5)
        self.alphabet = self.seqdist.alphabet
        self.config = config
        self._features = enc.get("features", 768)
        self._winlen = enc.get("winlen", 19)
        self._device = 0
        self._ctx = None
        self._ctx_key = None
        # mixed (default): the feed-forward projections in three fp16 products, the recurrent ones as fp16 product + FP8
        # block-scaled correction products -- |score error| 3.4e-4 max on trained-like (peaky) weights, 3e-5 on seeded ones;
        # f16f8: FP8 corrections everywhere, 1.15x faster, 1.1e-3 / 7e-5 (on the north star's 1e-3 tolerance with peaky weights);
        # f16x3: three fp16 products everywhere, 6e-5 / 6e-6; f16: one product, ~1e-3..2e-3 (what the reference's model.half()
        # computes); f16f8i: f16f8 with single-product LSTM input projections (seeded weights: ~6e-4 max)
        self.precision = _lib.PRECISIONS[
            os.environ.get("XNA_PRECISION", config.get("basecaller", {}).get("precision", "mixed"))]

    # ---- torch.nn.Module surface used by load_model -------------------------------------
    def to(self, device=None, *args, **kwargs):
        if device is not None and not isinstance(device, torch.dtype):
            self._device = _device_index(device)
            self._drop_context()
        return self

    def half(self):
        return self          # arithmetic is fixed by `precision`; the checkpoint stays fp32

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        self._drop_context()
        return out

    def _drop_context(self):
        if self._ctx is not None:
            self._ctx.close()
        self._ctx, self._ctx_key = None, None

    # ---- device context -------------------------------------------------------------------
    def context_is_current(self, chunk_len, batch):
        """True when context(chunk_len, batch) would hand back the live context (nothing in flight is lost)."""
        key = (self._device, int(chunk_len), self.precision)
        return self._ctx is not None and self._ctx_key == key and batch <= self._ctx.max_batch

    def context(self, chunk_len, batch):
        """The xb_ctx for this chunk length; rebuilt when the geometry grows.  Sized for at least the configured
        basecaller batchsize so that a short first batch does not force a rebuild on the next one."""
        key = (self._device, int(chunk_len), self.precision)
        if not self.context_is_current(chunk_len, batch):
            batch = max(int(batch), int(self.config.get("basecaller", {}).get("batchsize", 0) or 0))
            self._drop_context()
            last = self.encoder[-1]
            ctx = _lib.Context(self._device, self.seqdist.n_base, self.seqdist.state_len, self._features,
                               self._winlen, self.stride, float(last.scale), float(last.blank_score),
                               int(chunk_len), int(batch), precision=self.precision)
            sd = {k: v.detach().to(torch.float32).cpu().numpy() for k, v in self.state_dict().items()}
            ctx.load_state_dict(sd)
            self._ctx, self._ctx_key = ctx, key
        return self._ctx

    # ---- operators ----------------------------------------------------------------------
    @staticmethod
    def _as_signal(x):
        if hasattr(x, "detach"):
            x = x.detach().to(torch.float32).cpu().numpy()
        x = np.asarray(x, dtype=np.float32)
        if x.ndim == 3:
            x = x[:, 0, :]
        return np.ascontiguousarray(x)

    def forward(self, x):
        """(N,1,L) signal -> (T,N,C) fp32 scores on the host (crf/model.py:212-213)."""
        sig = self._as_signal(x)
        ctx = self.context(sig.shape[1], sig.shape[0])
        return ctx.encode(sig, expand_blanks=self.encoder[-1].expand_blanks)

    def decode_batch(self, x):
        """(T,N,C) scores -> list of called strings (crf/model.py:215-218)."""
        if hasattr(x, "detach"):
            x = x.detach().to(torch.float32).cpu().numpy()
        x = np.ascontiguousarray(x, dtype=np.float32)
        T, N, _ = x.shape
        ctx = self.context(T * self.stride, N)
        seq, lens = ctx.decode(x, self.alphabet)
        return [seq[i, :lens[i]].tobytes().decode() for i in range(N)]

    def decode(self, x):
        return self.decode_batch(np.asarray(x)[:, None, :])[0]

    def logZ(self, scores):
        """(T,N,C) scores -> (N,) log partition function (CTC_CRF.logZ, crf/model.py:41-46), on the device."""
        if hasattr(scores, "detach"):
            scores = scores.detach().to(torch.float32).cpu().numpy()
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        T, N, _ = scores.shape
        return self.context(T * self.stride, N).crf_logz(scores)

    def _scans(self, scores, want):
        if hasattr(scores, "detach"):
            scores = scores.detach().to(torch.float32).cpu().numpy()
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        T, N, _ = scores.shape
        return self.context(T * self.stride, N).crf_scans(scores, want=want)

    def forward_scores(self, scores):
        """(T,N,C) -> (T+1,N,S) Log forward scores (CTC_CRF.forward_scores, crf/model.py:50-54)."""
        return self._scans(scores, ("alpha",))["alpha"]

    def backward_scores(self, scores):
        """(T,N,C) -> (T+1,N,S) Log backward scores (CTC_CRF.backward_scores, crf/model.py:56-60)."""
        return self._scans(scores, ("beta",))["beta"]

    def posteriors(self, scores):
        """(T,N,C) -> (T,N,S*(n_base+1)) edge posteriors = d logZ / d scores (seqdist `posteriors`, Log semiring)."""
        return self._scans(scores, ("post",))["post"]

    def normalise(self, scores):
        """scores - logZ / T (CTC_CRF.normalise, crf/model.py:48-49)."""
        if hasattr(scores, "detach"):
            scores = scores.detach().to(torch.float32).cpu().numpy()
        scores = np.asarray(scores, dtype=np.float32)
        return scores - (self.logZ(scores) / np.float32(len(scores)))[None, :, None]

    def basecall_chunks(self, batch):
        """Fused encode + decode of a (N,1,L) batch -> (seq (N,T) int8 left-packed ASCII, lens (N,))."""
        sig = self._as_signal(batch)
        ctx = self.context(sig.shape[1], sig.shape[0])
        return ctx.basecall_chunks(sig, self.alphabet)

    def basecall_chunks_beam(self, batch, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0):
        """Fused encode + beam search of a (N,1,L) batch -> {'sequence', 'qstring' (N,T) int8, 'moves' (N,T) uint8, 'score'}:
        koi.decode.beam_search on the model's blank-less scores (crf/basecall.py:33-46), one device call."""
        sig = self._as_signal(batch)
        ctx = self.context(sig.shape[1], sig.shape[0])
        return ctx.basecall_chunks_beam(sig, self.alphabet, beam_width, beam_cut, scale, offset)

    def beam_search(self, scores, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0):
        """(T,N,C) host scores (with or without the blank column) -> the same dict."""
        if hasattr(scores, "detach"):
            scores = scores.detach().to(torch.float32).cpu().numpy()
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        T, N, _ = scores.shape
        return self.context(T * self.stride, N).beam_search(scores, self.alphabet, beam_width, beam_cut, scale, offset)

    def submit_chunks(self, slot, batch):
        """Enqueue the fused encode + decode of a (N,1,L) batch in pipeline slot 0 .. 3 without waiting; returns a handle
        for collect_chunks.  The caller keeps at most one handle per slot in flight."""
        sig = self._as_signal(batch)
        ctx = self.context(sig.shape[1], sig.shape[0])
        self.chunks_submitted = getattr(self, "chunks_submitted", 0) + sig.shape[0]
        return ctx, slot, ctx.submit_chunks(slot, sig, self.alphabet)

    def pipeline_depth(self, chunk_len, n):
        """Batches the host pipeline keeps in flight on the context for (chunk_len, n): 4 when the context co-schedules two
        calls per device pass (pair k+1 is then on the device before the host waits for pair k), else 2.  Opts the context
        in to the pairing (xb_reserve_pairing) the first time it is asked."""
        ctx = self.context(chunk_len, n)
        if not getattr(ctx, "_pairing_asked", False):
            ctx._pairing_asked = True
            try:
                ctx.reserve_pairing()
            except _lib.XbError as e:
                if e.code != _lib.XB_ERR_NOMEM:     # a device / HIP failure inside the reservation is NOT "no room": surface it
                    raise
                sys.stderr.write("> no device memory for co-scheduled pairs of %d chunks: every call runs on its own\n" % n)
        return 4 if ctx.pairing_active() else 2

    @staticmethod
    def collect_chunks(handle):
        ctx, slot, n = handle
        return ctx.collect_chunks(slot, n)
