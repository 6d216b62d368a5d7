"""
Host-side pipeline helpers with the reference's names, argument meaning and results
(ub-bonito/bonito/util.py).  numpy arrays replace torch tensors: nothing here runs torch ops,
the device work lives behind the C ABI (include/xna_basecaller.h).

  chunk            util.py:152-166      stitch        util.py:169-188
  batchify         util.py:191-210      unbatchify    util.py:213-225
  concat/select_range/size  util.py:66-103
  load_symbol      util.py:228-239      match_names   util.py:242-258
  load_model       util.py:261-366      mean_qscore_from_qstring  util.py:124-131
"""
import os
import re
import sys
from collections import OrderedDict
from importlib import import_module
from itertools import groupby
from operator import itemgetter

import numpy as np

from . import toml_lite

__dir__ = os.path.dirname(os.path.realpath(__file__))
__models__ = os.path.join(__dir__, "models")

__all__ = ["chunk", "stitch", "batchify", "unbatchify", "concat", "select_range", "size",
           "load_symbol", "match_names", "load_model", "mean_qscore_from_qstring",
           "column_to_set", "half_supported", "init"]


def init(seed, device):
    """util.py:40-53 -- seed the host RNGs; the HIP path has no nondeterministic kernels."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    if device == "cpu":
        return
    from . import _lib
    _lib.require_gpu()


def half_supported():
    """util.py:105-112.  The MI355X path computes in split-fp16 MFMA with fp32 accumulation
    whatever this returns; kept for interface parity."""
    return False


def mean_qscore_from_qstring(qstring):
    """util.py:124-131: mean of per-base error probabilities, floored at 1e-4 -> phred."""
    if len(qstring) == 0:
        return 0.0
    qs = np.frombuffer(qstring.encode("ascii"), dtype=np.uint8).astype(np.float64) - 33
    mean_err = np.exp(qs * (-np.log(10) / 10.0)).mean()
    return -10 * np.log10(max(mean_err, 1e-4))


def decode_ref(encoded, labels):
    """Integer-coded reference -> string, blanks removed (util.py:134-138)."""
    return "".join(labels[e] for e in encoded if e)


def accuracy(ref, seq, balanced=False, min_coverage=0.0):
    """Percent identity of the local alignment of `seq` against `ref` (util.py:402-424); the alignment itself is the
    library's host-side restatement of the reference's parasail call (xb_align_accuracy)."""
    from . import _lib
    return _lib.align_accuracy(ref, seq, balanced=balanced, min_coverage=min_coverage)


def column_to_set(filename, idx=0, skip_header=False):
    """
    The set of whitespace-separated field `idx` over the lines of `filename` (read-id lists);
    None when there is no such file.  Same contract as util.py:140-149.
    """
    if not filename or not os.path.isfile(filename):
        return None
    with open(filename, "r") as fh:
        lines = fh.read().splitlines()
    return {ln.split()[idx] for ln in lines[1 if skip_header else 0:]}


# ---------------------------------------------------------------------------------------
# type-agnostic helpers (util.py:66-103)
# ---------------------------------------------------------------------------------------

def concat(xs, dim=0):
    x0 = xs[0]
    if isinstance(x0, np.ndarray):
        return np.concatenate(xs, axis=dim)
    if isinstance(x0, list):
        return [x for l in xs for x in l]
    if isinstance(x0, str):
        return "".join(xs)
    if isinstance(x0, dict):
        return {k: concat([x[k] for x in xs], dim) for k in x0.keys()}
    if hasattr(x0, "numpy"):                       # torch tensors handed in by a caller
        import torch
        return torch.cat(xs, dim=dim)
    raise TypeError(type(x0))


def select_range(x, start, end, dim=0):
    if isinstance(x, dict):
        return {k: select_range(v, start, end, dim) for (k, v) in x.items()}
    if dim == 0 or isinstance(x, list):
        return x[start:end]
    return x[(*(slice(None),) * dim, slice(start, end))]


def size(x, dim=0):
    if hasattr(x, "shape"):
        return x.shape[dim]
    if dim == 0:
        return len(x)
    raise TypeError(type(x))


# ---------------------------------------------------------------------------------------
# chunk / stitch
# ---------------------------------------------------------------------------------------

def chunk(signal, chunksize, overlap):
    """
    Read signal (1-D) -> (n_chunks, 1, chunksize).  Short reads are LEFT-padded with zeros;
    otherwise windows of `chunksize` every `chunksize - overlap` samples starting at
    stub = (T - overlap) % (chunksize - overlap), plus signal[:chunksize] in front if stub > 0.
    """
    signal = np.ascontiguousarray(signal)
    T = signal.shape[0]
    if chunksize == 0:
        chunks = signal[None, :]
    elif T < chunksize:
        chunks = np.concatenate([np.zeros(chunksize - T, dtype=signal.dtype), signal])[None, :]
    else:
        step = chunksize - overlap
        stub = (T - overlap) % step
        n = (T - stub - chunksize) // step + 1
        # overlapping windows as a strided VIEW of the signal (no gather, no copy: batchify copies each row once)
        item = signal.strides[0]
        chunks = np.lib.stride_tricks.as_strided(signal[stub:], shape=(n, chunksize), strides=(step * item, item),
                                                 writeable=False)
        if stub > 0:
            chunks = np.concatenate([signal[None, :chunksize], chunks])
    return chunks[:, None, :]


def chunk_starts(length, chunksize, overlap):
    """Start sample of every chunk `chunk` would produce (negative = left padding amount)."""
    if chunksize == 0:
        return np.zeros(1, dtype=np.int64)
    if length < chunksize:
        return np.array([length - chunksize], dtype=np.int64)
    step = chunksize - overlap
    stub = (length - overlap) % step
    n = (length - stub - chunksize) // step + 1
    starts = stub + step * np.arange(n, dtype=np.int64)
    if stub > 0:
        starts = np.concatenate([[0], starts])
    return starts


def stitch(chunks, chunksize, overlap, length, stride, reverse=False):
    """
    Per-read (n_chunks, T) rows -> 1-D.  One chunk: the row as is.  Otherwise drop
    overlap//2 samples (in stride units) at every internal boundary; the first chunk ends at
    (stub + overlap//2)//stride when the read had a stub chunk.  The indices are time-domain
    while compute_scores' rows are left-packed base strings -- reproduced as is
    (SURVEY.md section 8a row 14).
    """
    if isinstance(chunks, dict):
        return {k: stitch(v, chunksize, overlap, length, stride, reverse=reverse) for k, v in chunks.items()}
    if chunks.shape[0] == 1:
        return chunks[0]
    semi = overlap // 2
    start, end = semi // stride, (chunksize - semi) // stride
    stub = (length - overlap) % (chunksize - overlap)
    first_end = (stub + semi) // stride if stub > 0 else end
    n = chunks.shape[0]
    if reverse:
        # literal index forms: with start == 0 the `[:-start]` slices are empty, as in the reference
        parts = [chunks[n - 1][:-start]]
        parts += [chunks[i][-end:-start] for i in range(n - 2, 0, -1)]
        parts += [chunks[0][-first_end:]]
    else:
        parts = [chunks[0, :first_end]] + [chunks[i, start:end] for i in range(1, n - 1)] + [chunks[n - 1, start:]]
    return concat(parts)


# ---------------------------------------------------------------------------------------
# batchify / unbatchify
# ---------------------------------------------------------------------------------------

def batchify(items, batchsize, dim=0):
    """
    (key, value) stream -> (keys, batch) stream with exactly `batchsize` rows per batch
    (last may be short); keys are ((key), (lo, hi)) row ranges inside the batch.
    """
    stack, pos = [], 0
    for k, v in items:
        n = size(v, dim)
        cuts = list(range(batchsize - pos, n, batchsize))
        for lo, hi in zip([0] + cuts, cuts + [n]):
            stack.append(((k, (pos, pos + hi - lo)), select_range(v, lo, hi, dim)))
            pos += hi - lo
            if pos == batchsize:
                ks, vs = zip(*stack)
                yield ks, concat(vs, dim)
                stack, pos = [], 0
    if stack:
        ks, vs = zip(*stack)
        yield ks, concat(vs, dim)


def unbatchify(batches, dim=0):
    """Inverse of batchify: regroup consecutive pieces with the same key (order preserving)."""
    pieces = (
        (k, select_range(v, lo, hi, dim))
        for sub, v in batches
        for k, (lo, hi) in sub
    )
    return (
        (k, concat([v for (_, v) in group], dim))
        for k, group in groupby(pieces, itemgetter(0))
    )


# ---------------------------------------------------------------------------------------
# model loading
# ---------------------------------------------------------------------------------------

def _model_dir(name):
    if not os.path.isdir(name) and os.path.isdir(os.path.join(__models__, name)):
        return os.path.join(__models__, name)
    return name


def load_symbol(config, symbol):
    """
    Resolve `symbol` from the package named by config['model']['package'] (the reference's
    plugin mechanism).  `bonito.crf` -- what shipped configs say -- maps to this package's
    drop-in `xna_basecaller_amd.crf` when ub-bonito itself is not importable.
    """
    if not isinstance(config, dict):
        config = toml_lite.load(os.path.join(_model_dir(config), "config.toml"))
    package = config["model"]["package"]
    try:
        if package.split(".")[0] == "bonito":
            raise ImportError
        imported = import_module(package)
    except ImportError:
        if package in ("bonito.crf", "xna_basecaller_amd.crf", "xnacall.crf"):
            imported = import_module("xna_basecaller_amd.crf")
        else:
            raise
    return getattr(imported, symbol)


def match_names(state_dict, model, skip_layers=()):
    """
    Checkpoint key -> model key by sorting BOTH dicts on (shape, original position) and zipping;
    this is what lets dropout-interleaved training encoders load into the inference encoder.
    """
    def keys_and_shapes(sd):
        rows = sorted((tuple(v.shape), i, k) for i, (k, v) in enumerate(sd.items()))
        rows = [(k, s) for s, i, k in rows if k not in skip_layers]
        return zip(*rows)
    k1, s1 = keys_and_shapes(state_dict)
    k2, s2 = keys_and_shapes(model.state_dict())
    assert s1 == s2, "checkpoint tensors do not match the model's shapes"
    remap = dict(zip(k1, k2))
    return OrderedDict((k, remap[k]) for k in state_dict.keys() if k not in skip_layers)


def _checkpoint_path(dirname, weights):
    """weights_<N>.tar for the requested N, or for the largest N present."""
    if not weights:
        numbers = [int(m.group(1)) for m in
                   (re.search(r"weights_([0-9]+)\.tar$", f) for f in os.listdir(dirname)) if m]
        if not numbers:
            raise FileNotFoundError("no model weights found in '%s'" % dirname)
        weights = max(numbers)
    return os.path.join(dirname, "weights_%s.tar" % weights)


def _first_set(*candidates):
    for c in candidates:
        if c is not None:
            return c


def _remapped_checkpoint(path, model, skip_layers):
    """Checkpoint tensors under the model's own key names (DataParallel 'module.' prefix removed)."""
    import torch  # checkpoint deserialisation only
    saved = torch.load(path, map_location="cpu")
    out = OrderedDict()
    for src, dst in match_names(saved, model, skip_layers).items():
        out[dst.replace("module.", "")] = saved[src]
    return out


def load_model(dirname, device, weights=None, half=None, chunksize=None, batchsize=None,
               overlap=None, quantize=False, use_koi=False, skip_top=False, drop_rate=None,
               drop_rate_bottom=None):
    """
    config.toml + weights_<N>.tar -> Model on `device` (util.py:261-366).  Run parameters:
    command-line flag, else the [basecaller] table, else 4000 / 500 / 64 (a zero chunksize or
    batchsize counts as unset, a zero overlap does not); the newest checkpoint unless
    `weights` names one; keys remapped by match_names; `use_koi` selects the beam-search decoder for
    4-base models with a fixed blank score and is dropped for XNA alphabets (util.py:299-301).
    """
    dirname = _model_dir(dirname)
    checkpoint = _checkpoint_path(dirname, weights)
    config = toml_lite.load(os.path.join(dirname, "config.toml"))

    run = config.setdefault("basecaller", {})
    run["chunksize"] = chunksize or run.get("chunksize", 4000)
    run["batchsize"] = batchsize or run.get("batchsize", 64)
    run["overlap"] = _first_set(overlap, run.get("overlap"), 500)
    run["quantize"] = run.get("quantize") if quantize is None else quantize
    enc = config["encoder"]
    enc["drop_rate"] = _first_set(drop_rate, enc.get("drop_rate"), 0)
    enc["drop_rate_bottom"] = _first_set(drop_rate_bottom, enc.get("drop_rate_bottom"), 0)

    model = load_symbol(config, "Model")(config)

    if use_koi:
        if model.seqdist.n_base != 4:
            sys.stderr.write("[Warning] Setting use_koi to False because n_base != 4.\n")
        elif model.encoder[-1].blank_score is not None:
            # util.py:304-313: koi's encoder hands blank-less scores to koi.decode.beam_search (crf/basecall.py:31-46).  Here the
            # encoder is the same hand-written one either way; what the flag selects is the decoder: xb_basecall_chunks_beam.
            model.encoder[-1].expand_blanks = False

    skip_layers = []
    if skip_top:
        skip_layers = [k for k in model.state_dict() if k.startswith("encoder.9")]
        sys.stderr.write("[WARNING: skipping top layer weights]\n")
    missing, unexpected = model.load_state_dict(
        _remapped_checkpoint(checkpoint, model, skip_layers), strict=not skip_top)
    if unexpected or list(missing) != skip_layers:
        raise RuntimeError("checkpoint does not fit the model: missing %s, unexpected %s"
                           % (list(missing), list(unexpected)))

    if half is None:
        half = half_supported()
    if half:
        model = model.half()
    model.eval()
    model.to(device)
    return model
