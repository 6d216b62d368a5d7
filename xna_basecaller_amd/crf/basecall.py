"""
CRF basecalling pipeline (ub-bonito/bonito/crf/basecall.py): chunk -> batch -> compute_scores ->
unbatch -> stitch -> strings, as five generator stages on background threads with bounded queues
(bonito/multiprocessing.py:20-24,92-122), strict FIFO order.
"""
import queue
from collections import deque
from threading import Thread

import numpy as np

from ..util import chunk, stitch, batchify, unbatchify, mean_qscore_from_qstring


class _ThreadIterator(Thread):
    """Run an iterator on a background thread behind a bounded queue (exceptions are re-raised)."""

    def __init__(self, iterator, maxsize=1):
        super().__init__(daemon=True)
        self.iterator = iterator
        self.queue = queue.Queue(maxsize)

    def __iter__(self):
        self.start()
        while True:
            item = self.queue.get()
            if item is StopIteration:
                break
            if isinstance(item, BaseException):
                raise item
            yield item

    def run(self):
        try:
            for item in self.iterator:
                self.queue.put(item)
            self.queue.put(StopIteration)
        except BaseException as e:  # propagate to the consumer instead of hanging it
            self.queue.put(e)


def thread_iter(iterator, maxsize=1):
    return iter(_ThreadIterator(iterator, maxsize=maxsize))


def stitch_results(results, length, size, overlap, stride, reverse=False):
    """crf/basecall.py:15-24"""
    if isinstance(results, dict):
        return {k: stitch_results(v, length, size, overlap, stride, reverse=reverse) for k, v in results.items()}
    return stitch(results, size, overlap, length, stride, reverse=reverse)


def compute_scores(model, batch, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0, blank_score=2.0,
                   reverse=False):
    """
    crf/basecall.py:27-82: (n,1,L) batch -> {'sequence': int8 (n,T), 'qstring': int8 (n,T), 'moves': bool (n,T)}.
    Viterbi branch (expand_blanks, the only one the reference reaches for XNA alphabets): left-packed ASCII rows, 'O'
    placeholders, no moves -- one fused device call; the per-character Python loops of the reference are gone.
    Beam branch (expand_blanks = False, `koi.decode.beam_search`): bases and quality characters at the blocks that emit,
    real moves -- xb_basecall_chunks_beam, for any alphabet the CRF supports.
    """
    if not model.encoder[-1].expand_blanks:
        own = model.encoder[-1].blank_score
        if own is None or float(own) != float(blank_score):
            raise ValueError("beam search uses the model's fixed blank score (%r); blank_score=%r was asked for"
                             % (own, blank_score))
        if reverse:
            scores = model.seqdist.reverse_complement(model(batch))
            res = model.beam_search(scores, beam_width, beam_cut, scale, offset)
        else:
            res = model.basecall_chunks_beam(batch, beam_width, beam_cut, scale, offset)
        return {"qstring": res["qstring"], "sequence": res["sequence"], "moves": res["moves"].astype(bool)}
    if reverse:
        scores = model.seqdist.reverse_complement(model(batch))
        ctx = model.context(np.asarray(batch).shape[-1], scores.shape[1])
        sequence, _ = ctx.decode(scores, model.alphabet)
    else:
        sequence, _ = model.basecall_chunks(batch)
    return _scores_dict(sequence)


def _scores_dict(sequence):
    """The reference's result layout around the left-packed ASCII rows: dummy quality 'O', no moves."""
    qstring = np.where(sequence != 0, np.int8(ord("O")), np.int8(0)).astype(np.int8)
    return {
        "qstring": qstring,
        "sequence": sequence,
        "moves": np.zeros(sequence.shape, dtype=bool),
    }


def compute_sequences_pipelined(model, batches, reverse=False):
    """
    The device stage of `basecall`: (key, batch) stream -> (key, sequence (n,T) int8 left-packed ASCII) with several batches
    in flight on the device: batch k+1 is submitted (pinned staging, H2D on a copy stream, fused kernels, D2H) before
    batch k's result is waited for, so the GPU never idles while the host unpacks results.  Where the context co-schedules two
    calls per device pass (Model.pipeline_depth: batches of at most 640 chunks) FOUR batches rotate through four staging
    slots -- batch k+3 is submitted before batch k is collected, so pair (k+2, k+3) is on the device, its H2D copies done,
    while the host waits for pair (k, k+1) -- otherwise two.  Results come out in input order, depth - 1 batches late.
    (compute_scores is the same operator, synchronous, with the reference's full result dict.)
    """
    if reverse or not model.encoder[-1].expand_blanks:
        for key, batch in batches:                       # decode of host-side reverse-complemented scores: synchronous
            yield key, compute_scores(model, batch, reverse=reverse)["sequence"]
        return
    pending, slot, depth = deque(), 0, 2
    for key, batch in batches:
        shape = np.asarray(batch).shape
        if pending and not model.context_is_current(shape[-1], shape[0]):
            while pending:                                                # drain before the context is rebuilt
                k, h = pending.popleft()
                yield k, model.collect_chunks(h)[0]
        if not pending:
            depth, slot = model.pipeline_depth(shape[-1], shape[0]), 0
        pending.append((key, model.submit_chunks(slot, batch)))
        slot = (slot + 1) % depth
        if len(pending) == depth:                                         # the slot the next batch goes into
            k, h = pending.popleft()
            yield k, model.collect_chunks(h)[0]
    while pending:
        k, h = pending.popleft()
        yield k, model.collect_chunks(h)[0]


def compute_scores_pipelined(model, batches, reverse=False):
    """compute_scores over a stream of (key, batch), two batches in flight; yields the reference's result dicts."""
    if not model.encoder[-1].expand_blanks:              # beam search: qualities and moves are real, one batch at a time
        for key, batch in batches:
            yield key, compute_scores(model, batch, reverse=reverse)
        return
    for key, sequence in compute_sequences_pipelined(model, batches, reverse=reverse):
        yield key, _scores_dict(sequence)


def to_str(x, encoding="ascii"):
    """koi.decode.to_str: int8 array -> str without the zero padding."""
    x = np.asarray(x)
    return x[x != 0].astype(np.uint8).tobytes().decode(encoding)


def apply_stride_to_moves(model, attrs):
    """crf/basecall.py:85-93"""
    moves = np.array(attrs["moves"], dtype=bool)
    sig_move = np.full(moves.size * model.stride, False)
    sig_move[np.where(moves)[0] * model.stride] = True
    return {
        "qstring": to_str(attrs["qstring"]),
        "sequence": to_str(attrs["sequence"]),
        "sig_move": sig_move,
    }


def _called(model, sequence):
    """
    The per-read result of crf/basecall.py:85-93 from the stitched left-packed row alone.  The Viterbi branch's quality
    string and moves are placeholders that mirror the sequence (crf/basecall.py:60-76: 'O' wherever a base was written,
    no moves), so stitching them separately and then dropping their padding gives exactly 'O' * len(sequence) and an
    all-False signal-move vector of one entry per stitched slot and stride.
    """
    seq = to_str(sequence)
    return {"qstring": "O" * len(seq), "sequence": seq, "sig_move": np.zeros(np.asarray(sequence).size * model.stride, dtype=bool),
            "mean_qscore": 40.0 if seq else 0.0}           # = mean_qscore_from_qstring('O' * n), util.py:124-131


def _called_beam(model, attrs):
    """crf/basecall.py:85-93 on stitched beam-search results, plus the mean quality the writers print."""
    out = apply_stride_to_moves(model, attrs)
    out["mean_qscore"] = mean_qscore_from_qstring(out["qstring"]) if out["qstring"] else 0.0
    return out


def basecall(model, reads, chunksize=4000, overlap=100, batchsize=32, reverse=False):
    """Basecall `reads` (objects with .signal); yields (read, {'sequence','qstring','sig_move'}) in input order."""
    chunks = thread_iter(
        ((read, 0, len(read.signal)), chunk(np.asarray(read.signal, dtype=np.float32), chunksize, overlap))
        for read in reads
    )
    batches = thread_iter(batchify(chunks, batchsize=batchsize))
    if not model.encoder[-1].expand_blanks:
        # the reference's own five stages (crf/basecall.py:96-122): result dicts are unbatched and stitched plane by plane
        scores = thread_iter(compute_scores_pipelined(model, batches, reverse=reverse))
        results = thread_iter(
            (read, stitch_results(attrs, end - start, chunksize, overlap, model.stride, reverse))
            for ((read, start, end), attrs) in unbatchify(scores)
        )
        return thread_iter((read, _called_beam(model, attrs)) for read, attrs in results)
    sequences = thread_iter(compute_sequences_pipelined(model, batches, reverse=reverse))
    results = thread_iter(
        (read, stitch(seq, chunksize, overlap, end - start, model.stride, reverse=reverse))
        for ((read, start, end), seq) in unbatchify(sequences)
    )
    return thread_iter(
        (read, _called(model, stitched))
        for read, stitched in results
    )
