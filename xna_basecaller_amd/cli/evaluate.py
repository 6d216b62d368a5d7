"""
`bonito evaluate MODEL_DIR --directory CTC_DATA` (ub-bonito/bonito/cli/evaluate.py): call the validation chunks with every
requested checkpoint and report the mean / median accuracy against the references, the time and samples/s.

Device work = the reference's `model(data)` + `model.decode_batch(log_probs)` per batch (evaluate.py:57-72): here one fused
call per batch (`Model.basecall_chunks`: the same scores and the same decode without moving the scores to the host).
Accuracy = util.accuracy (host side; xb_align_accuracy restates the parasail call, see csrc/xb_align.hip).  --poa (spoa
consensus over several checkpoints) is outside the MI355X path and refused.
"""
import time
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser
from pathlib import Path

import numpy as np

from ..data import load_validation
from ..util import accuracy, decode_ref, init, load_model


def main(args):
    if args.poa:
        raise SystemExit("> error: --poa (spoa consensus) is not part of the MI355X path")
    init(args.seed, args.device)
    print("* loading data")
    chunks, targets, lengths = load_validation(args.chunks, args.directory)
    chunks = np.asarray(chunks, dtype=np.float32)
    for w in [int(i) for i in args.weights.split(",")]:
        print("* loading model", w)
        model = load_model(args.model_directory, args.device, weights=w)
        print("* calling")
        t0 = time.perf_counter()
        seqs = []
        for b0 in range(0, len(chunks), args.batchsize):
            batch = chunks[b0:b0 + args.batchsize]
            seq, lens = model.basecall_chunks(batch[:, None, :])
            seqs.extend(seq[i, :lens[i]].tobytes().decode() for i in range(len(batch)))
        duration = time.perf_counter() - t0
        print("* decoding refs")
        refs = [decode_ref(t, model.alphabet) for t in targets]
        print("* computing accuracies")
        accuracies = [accuracy(ref, seq, min_coverage=args.min_coverage) if len(seq) else 0. for ref, seq in zip(refs, seqs)]
        print("* mean      %.2f%%" % np.mean(accuracies))
        print("* median    %.2f%%" % np.median(accuracies))
        print("* time      %.2f" % duration)
        print("* samples/s %.2E" % (len(chunks) * chunks.shape[1] / duration))
    return accuracies


def argparser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter, add_help=False)
    parser.add_argument("model_directory")
    parser.add_argument("--directory", type=Path)
    parser.add_argument("--device", default="cuda")
    parser.add_argument("--seed", default=9, type=int)
    parser.add_argument("--weights", default="0", type=str)
    parser.add_argument("--chunks", default=1000, type=int)
    parser.add_argument("--batchsize", default=96, type=int)
    parser.add_argument("--beamsize", default=5, type=int)
    parser.add_argument("--poa", action="store_true", default=False)
    parser.add_argument("--min-coverage", default=0.5, type=float)
    return parser
