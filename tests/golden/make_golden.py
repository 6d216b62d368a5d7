#!/usr/bin/env python
"""
Generate the golden fixtures under tests/golden/ by IMPORTING the reference's own Python
(/root/reference/ub-bonito/bonito/*.py) in this container and recording inputs/outputs.

Run here only (the GPU box has no /root/reference):   python tests/golden/make_golden.py

What is reference code and what is not:
  * bonito/nn.py, bonito/util.py (chunk, stitch, batchify, unbatchify, match_names,
    mean_qscore_from_qstring), bonito/fast5.py (trim, med_mad, norm_by_noisiest_section) and
    bonito/crf/model.py (CTC_CRF.idx, n_score, viterbi's argmax %% E, path_to_str,
    rnn_encoder, Model, decode_batch) run AS REFERENCE CODE, loaded by file path.
  * Third-party modules that are absent here and unused by the functions above (toml, koi,
    parasail, ont_fast5_api) are empty placeholders in sys.modules so that the files import.
  * seqdist (ont-seqdist-cuda 0.0.4, CUDA-only, not in the tree) is what computes
    SequenceDist.posteriors / sparse.logZ.  For the decode fixtures it is replaced by the
    torch fp32 restatement (`_LogZ`, `_SequenceDist`) below, so those fixtures are flagged
    "reference control-flow + restated seqdist math" in their metadata and do NOT pin the
    third-party arithmetic ("parity unpinned", see oracle/xna_oracle.c).
No reference source text is stored: fixtures are inputs and outputs only.
"""
import importlib.util
import json
import os
import sys
import types
from collections import OrderedDict, namedtuple

import numpy as np
import torch

REF = "/root/reference/ub-bonito/bonito"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


# --------------------------------------------------------------------------------------
# restated seqdist (torch, fp32) -- NOT reference code
# --------------------------------------------------------------------------------------
semiring = namedtuple("semiring", ("zero", "one", "mul", "sum", "dsum"))


def _max_grad(x, dim=0):
    return torch.zeros_like(x).scatter_(dim, x.argmax(dim, True), 1.0)


Log = semiring(zero=-1e38, one=0.0, mul=torch.add, sum=torch.logsumexp, dsum=torch.softmax)
Max = semiring(zero=-1e38, one=0.0, mul=torch.add,
               sum=(lambda x, dim=0: torch.max(x, dim=dim)[0]), dsum=_max_grad)


class _LogZ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Ms, idx, v0, vT, S):
        T, N, C, NZ = Ms.shape
        idx = idx.long()
        alpha = [v0]
        for t in range(T):
            alpha.append(S.sum(S.mul(Ms[t], alpha[-1][:, idx]), dim=2))
        alpha = torch.stack(alpha)
        ctx.save_for_backward(alpha, Ms, idx, vT)
        ctx.semiring = S
        return S.sum(S.mul(alpha[-1], vT), dim=1)

    @staticmethod
    def backward(ctx, grad):
        alpha, Ms, idx, vT = ctx.saved_tensors
        S = ctx.semiring
        T, N, C, NZ = Ms.shape
        beta = [vT]
        flat = idx.flatten()
        for t in range(T - 1, -1, -1):
            x = S.mul(Ms[t], beta[-1][:, :, None]).reshape(N, -1)   # edge (j,k) -> score + beta[j]
            b = torch.full((N, C), S.zero, dtype=Ms.dtype)
            # sum over edges grouped by their source state
            order = flat.argsort(stable=True)
            xs = x[:, order].reshape(N, C, NZ)
            b = S.sum(xs, dim=2)
            beta.append(b)
        beta = torch.stack(beta[::-1])
        g = S.mul(S.mul(Ms.reshape(T, N, -1), alpha[:-1][:, :, flat]).reshape(T, N, C, NZ),
                  beta[1:, :, :, None])
        g = S.dsum(g.reshape(T, N, -1), dim=2).reshape(T, N, C, NZ)
        return grad[None, :, None, None] * g, None, None, None, None


class _SequenceDist:
    def __init__(self):
        pass

    def posteriors(self, scores, S=Log):
        x = scores.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            y = self.logZ(x, S).sum()
            g, = torch.autograd.grad(y, x)
        return g


def _install_placeholders():
    pkg = types.ModuleType("bonito")
    pkg.__path__ = [REF]
    sys.modules["bonito"] = pkg
    for name in ["toml", "koi", "koi.lstm", "koi.decode", "parasail", "ont_fast5_api",
                 "ont_fast5_api.fast5_interface"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["koi"].lstm = sys.modules["koi.lstm"]
    sys.modules["ont_fast5_api.fast5_interface"].get_fast5_file = None
    sd = types.ModuleType("seqdist")
    sd.sparse = types.ModuleType("seqdist.sparse")
    sd.sparse.logZ = lambda Ms, idx, v0, vT, S: _LogZ.apply(Ms, idx, v0, vT, S)
    sd.ctc_simple = types.ModuleType("seqdist.ctc_simple")
    sd.ctc_simple.logZ_cupy = None
    sd.ctc_simple.viterbi_alignments = None
    sd.core = types.ModuleType("seqdist.core")
    sd.core.SequenceDist, sd.core.Max, sd.core.Log, sd.core.semiring = _SequenceDist, Max, Log, semiring
    for k in ("seqdist", "seqdist.sparse", "seqdist.ctc_simple", "seqdist.core"):
        sys.modules[k] = {"seqdist": sd, "seqdist.sparse": sd.sparse,
                          "seqdist.ctc_simple": sd.ctc_simple, "seqdist.core": sd.core}[k]


SHIPPED_CONFIG = {  # values of models/xna_r9.4.1_e8_sup@v3.3/config.toml:1-29
    "global_norm": {"state_len": 3},
    "qscore": {"bias": 0.3498, "scale": 0.9722},
    "input": {"features": 1},
    "model": {"package": "bonito.crf"},
    "labels": {"labels": ["N", "A", "C", "G", "T", "X", "Y"]},
    "encoder": {"stride": 5, "activation": "swish", "features": 768, "winlen": 19, "scale": 5.0,
                "rnn_type": "lstm", "blank_score": 2.0},
    "basecaller": {"batchsize": 384, "chunksize": 3600, "overlap": 500},
}


def _cfg(features, labels):
    c = json.loads(json.dumps(SHIPPED_CONFIG))
    c["encoder"]["features"] = features
    c["labels"]["labels"] = labels
    return c


def seeded_state_dict(model_state, seed):
    """Deterministic weights: N(0, 1/sqrt(fan_in)), bias_ih N(0,0.5) clipped, bias_hh 0."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for k, v in model_state.items():
        shp = tuple(v.shape)
        if k.endswith("bias_hh_l0"):
            a = np.zeros(shp, np.float32)
        elif "bias" in k:
            a = np.clip(0.5 * rng.standard_normal(shp), -1, 1).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = (rng.standard_normal(shp) / np.sqrt(fan_in)).astype(np.float32)
        out[k] = a
    return out


def make_revcomp():
    """(10) CTC_CRF.reverse_complement (crf/model.py:78-90, pure torch, reference code): index maps for nb 4/5/6 and
    state_len 2/3.  The input holds its own flat index, so the output IS the gather map out[t, n, c] = in[T-1-t, n, map[c]]."""
    _install_placeholders()
    crfpkg = types.ModuleType("bonito.crf")
    crfpkg.__path__ = [REF + "/crf"]
    sys.modules["bonito.crf"] = crfpkg
    if "bonito.nn" not in sys.modules:
        _load("bonito.nn", REF + "/nn.py")
    crf = sys.modules.get("bonito.crf.model") or _load("bonito.crf.model", REF + "/crf/model.py")
    out = {}
    for nb in (4, 5, 6):
        for sl in (2, 3):
            labels = ["N", "A", "C", "G", "T", "X", "Y"][:nb + 1]
            sd = crf.CTC_CRF(sl, labels)
            C = (nb + 1) * nb ** sl
            T, N = 3, 2
            x = torch.arange(T * N * C, dtype=torch.float32).reshape(T, N, C)
            y = sd.reverse_complement(x).numpy().astype(np.int64)
            out["nb%d_sl%d" % (nb, sl)] = y
    np.savez_compressed(os.path.join(OUT, "revcomp.npz"), **out)
    print("revcomp.npz written")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _install_placeholders()
    nn = _load("bonito.nn", REF + "/nn.py")
    util = _load("bonito.util", REF + "/util.py")
    f5 = _load("bonito.fast5", REF + "/fast5.py")
    crfpkg = types.ModuleType("bonito.crf")
    crfpkg.__path__ = [REF + "/crf"]
    sys.modules["bonito.crf"] = crfpkg
    crf = _load("bonito.crf.model", REF + "/crf/model.py")

    # ---- (1) chunk / stitch ------------------------------------------------------------
    cases = []
    table = [(2500, 10000, 500), (10000, 10000, 500), (23000, 10000, 500), (19500, 10000, 500),
             (29000, 10000, 500), (4000, 4000, 500), (7123, 4000, 500), (11500, 4000, 500),
             (3600, 3600, 500), (9999, 3600, 500), (16000, 3600, 500), (650, 200, 50),
             (1000, 200, 0), (403, 200, 50), (199, 200, 50), (350, 200, 50), (4001, 4000, 100)]
    stride = 5
    for (length, L, ov) in table:
        sig = torch.arange(length, dtype=torch.float32) + 1.0       # 0 marks padding
        ch = util.chunk(sig, L, ov)
        n = ch.shape[0]
        first = ch[:, 0, 0].numpy().astype(np.int64) - 1             # start index, -1 = padded
        npad = int((ch[0, 0] == 0).sum())
        T = L // stride
        rows = (torch.arange(n * T, dtype=torch.int32).reshape(n, T) + 1)
        st = util.stitch(rows, L, ov, length, stride)
        st_rev = util.stitch(rows, L, ov, length, stride, reverse=True) if n > 1 else st
        cases.append(dict(length=length, chunksize=L, overlap=ov, stride=stride, n_chunks=int(n),
                          starts=first.tolist(), left_pad=npad,
                          stitch=(st.numpy().astype(np.int64) - 1).tolist(),
                          stitch_reverse=(np.asarray(st_rev).astype(np.int64) - 1).tolist()))
    with open(os.path.join(OUT, "chunk_stitch.json"), "w") as fh:
        json.dump({"source": "bonito/util.py chunk:152-166 stitch:169-188 (reference code, imported)",
                   "cases": cases}, fh)

    # ---- (2) batchify / unbatchify -----------------------------------------------------
    bcases = []
    for lens, bs in [([3, 1, 5, 2, 8, 1], 4), ([1, 1, 1], 8), ([9], 4), ([4, 4, 4], 4), ([2, 7, 3], 5)]:
        items = [(("read%d" % i, 0, n * 10), torch.full((n, 1, 6), float(i))) for i, n in enumerate(lens)]
        batches = list(util.batchify(iter(items), bs))
        keys = [[[list(k[0]), list(k[1])] for k in ks] for ks, v in batches]
        sizes = [int(v.shape[0]) for ks, v in batches]
        un = list(util.unbatchify(iter(batches)))
        bcases.append(dict(n_chunks=lens, batchsize=bs, keys=keys, batch_sizes=sizes,
                           unbatch=[[list(k), int(v.shape[0]), float(v[0, 0, 0])] for k, v in un]))
    with open(os.path.join(OUT, "batchify.json"), "w") as fh:
        json.dump({"source": "bonito/util.py batchify:191-210 unbatchify:213-225 (reference code, imported)",
                   "cases": bcases}, fh)

    # ---- (3) signal preparation --------------------------------------------------------
    rng = np.random.default_rng(7)
    sigs, outs = {}, {}
    for i, n in enumerate([12000, 9000, 6000, 3000, 8001]):
        base = rng.normal(90.0, 12.0, n)
        lead = int(rng.integers(400, min(3000, n // 3)))
        base[:lead] = rng.normal(140.0, 3.0, lead)                   # open-pore-like prefix
        base[lead:lead + 120] += np.linspace(60, 0, 120)
        if i == 3:
            base[1000:2000] = rng.normal(90.0, 1.0, 1000)             # a quiet stall
        raw = np.round(base * 8.0).astype(np.int16)
        offset, rng_, digi = 10, 1443.03, 8192.0
        scaling = rng_ / digi
        scaled = np.array(scaling * (raw + offset), dtype=np.float32)
        t0, tlen = f5.trim(scaled[:8000])
        trimmed = scaled[t0:]
        med, mad = f5.med_mad(trimmed)
        if len(trimmed) > 8000:
            norm = (trimmed - med) / mad
        else:
            norm = f5.norm_by_noisiest_section(trimmed)
        sigs["raw%d" % i] = raw
        outs["trim%d" % i] = np.array([t0, tlen], dtype=np.int64)
        outs["medmad%d" % i] = np.array([med, mad], dtype=np.float64)
        outs["signal%d" % i] = np.asarray(norm, dtype=np.float32)
        outs["noisiest%d" % i] = np.asarray(f5.norm_by_noisiest_section(trimmed[:7000]), dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "signal_prep.npz"), offset=10, range=1443.03,
                        digitisation=8192.0, n=5, **sigs, **outs)

    # ---- (4) CRF table + (6) blank layout ------------------------------------------------
    idxs = {}
    for nb in (4, 5, 6):
        labels = ["N", "A", "C", "G", "T", "X", "Y"][:nb + 1]
        sdist = crf.CTC_CRF(state_len=3, alphabet=labels)
        idxs["idx_nb%d" % nb] = sdist.idx.numpy()
        idxs["n_score_nb%d" % nb] = np.int64(sdist.n_score())
    for sl in (1, 2, 4):
        idxs["idx_nb4_sl%d" % sl] = crf.CTC_CRF(state_len=sl, alphabet=list("NACGT")).idx.numpy()
    np.savez_compressed(os.path.join(OUT, "crf_idx.npz"), **idxs)

    # ---- (5) encoder: reference nn.py modules, fp32 CPU ------------------------------------
    enc = {}
    meta = {"source": "bonito/crf/model.py Model/rnn_encoder + bonito/nn.py (reference code, imported); "
                      "weights seeded by make_golden.seeded_state_dict", "cases": []}
    for name, features, labels, L, N in [("f32_nb6", 32, list("NACGTXY"), 400, 3),
                                         ("f32_nb4_long", 32, list("NACGT"), 2000, 2),
                                         ("f48_nb5", 48, list("NACGTX"), 600, 2),
                                         ("f16_nb4", 16, list("NACGT"), 300, 4)]:
        model = crf.Model(_cfg(features, labels))
        sd = seeded_state_dict(model.state_dict(), seed=features + len(labels))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval()
        x = torch.from_numpy(np.random.default_rng(L).standard_normal((N, 1, L)).astype(np.float32))
        with torch.no_grad():
            scores = model(x)
            h = x
            inter = []
            for li, layer in enumerate(model.encoder):
                h = layer(h)
                if li in (2, 4, 8):
                    inter.append(h.numpy().copy())
        for k, v in sd.items():
            enc["%s/w/%s" % (name, k)] = v
        enc[name + "/signal"] = x.numpy()
        enc[name + "/scores"] = scores.numpy()
        enc[name + "/conv_out"] = inter[0]          # (N,F,T)
        enc[name + "/lstm0_out"] = inter[1]         # (T,N,F)
        enc[name + "/lstm4_out"] = inter[2]
        meta["cases"].append(dict(name=name, features=features, labels=labels, L=L, N=N,
                                  stride=int(model.stride), n_score=int(model.seqdist.n_score()),
                                  keys=list(sd.keys()), shapes=[list(v.shape) for v in sd.values()]))
    np.savez_compressed(os.path.join(OUT, "encoder_small.npz"), **enc)

    # full-size (features 768) single short chunk: only sampled outputs are stored, the weights are
    # regenerated by the same seeded generator in the test.
    model = crf.Model(_cfg(768, list("NACGTXY")))
    sd = seeded_state_dict(model.state_dict(), seed=768)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    x = torch.from_numpy(np.random.default_rng(768).standard_normal((2, 1, 500)).astype(np.float32))
    with torch.no_grad():
        scores = model(x).numpy()
    meta["full"] = dict(features=768, labels=list("NACGTXY"), L=500, N=2, seed=768, signal_seed=768,
                        keys=list(sd.keys()), shapes=[list(v.shape) for v in sd.values()],
                        n_params=int(sum(v.size for v in sd.values())))
    np.savez_compressed(os.path.join(OUT, "encoder_full.npz"), scores_t0=scores[0], scores_tlast=scores[-1],
                        scores_mid=scores[50, :, ::7],
                        checksum=np.float64(scores.astype(np.float64).sum()),
                        abs_checksum=np.float64(np.abs(scores.astype(np.float64)).sum()))

    # dropout-interleaved training encoder -> inference encoder key remap (util.match_names)
    cfg_do = _cfg(32, list("NACGTXY"))
    enc_do = crf.rnn_encoder(6, 3, insize=1, stride=5, winlen=19, activation="swish", rnn_type="lstm",
                             features=32, scale=5.0, blank_score=2.0, drop_rate_bottom=0.1)
    sd_do = OrderedDict(("encoder." + k, v) for k, v in enc_do.state_dict().items())
    model = crf.Model(cfg_do)
    remap = util.match_names(sd_do, model)
    meta["match_names"] = dict(train_keys=list(sd_do.keys()),
                               train_shapes=[list(v.shape) for v in sd_do.values()],
                               model_keys=list(model.state_dict().keys()),
                               remap=[[k, v] for k, v in remap.items()])
    meta["qscore"] = [[q, float(util.mean_qscore_from_qstring(q))] for q in ["", "O" * 10, "!+5?I", "OOOO5"]]
    with open(os.path.join(OUT, "encoder_meta.json"), "w") as fh:
        json.dump(meta, fh)

    # ---- (7) decode: reference control flow + restated seqdist math --------------------------
    dec = {}
    dmeta = {"flag": "reference control-flow + restated seqdist math (fp32 torch stand-in); "
                     "does not pin the third-party arithmetic", "cases": []}
    for nb, T, N, seed in [(4, 120, 3, 11), (5, 100, 2, 12), (6, 100, 2, 13), (6, 40, 1, 14)]:
        labels = list("NACGTXY")[:nb + 1]
        model = crf.Model(_cfg(16, labels))
        S, E = nb ** 3, nb + 1
        rng = np.random.default_rng(seed)
        sc = 5.0 * np.tanh(rng.standard_normal((T, N, S, E)))
        sc = sc.astype(np.float16)
        sc[..., 0] = 2.0
        scores = torch.from_numpy(sc.astype(np.float32).reshape(T, N, S * E))
        seqs = model.decode_batch(scores)
        post = model.seqdist.posteriors(scores.to(torch.float32))
        tb = model.seqdist.viterbi((post + 1e-8).log()).to(torch.int16).T.numpy()
        name = "nb%d_T%d" % (nb, T)
        dec[name + "/scores_f16"] = sc.reshape(T, N, S * E)
        dec[name + "/labels"] = tb.astype(np.int8)
        dec[name + "/post_sample"] = post.numpy()[:, :, ::13]
        dmeta["cases"].append(dict(name=name, nb=nb, T=T, N=N, labels=labels, sequences=seqs))
    np.savez_compressed(os.path.join(OUT, "decode_small.npz"), **dec)
    with open(os.path.join(OUT, "decode_meta.json"), "w") as fh:
        json.dump(dmeta, fh)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print("  %-24s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "revcomp":      # regenerate only that fixture
        make_revcomp()
    else:
        main()
        make_revcomp()
