"""
`bonito basecaller`-compatible command line (ub-bonito/bonito/cli/basecaller.py:24-196): same positional
arguments, flags, defaults and stderr lines; FASTQ (or, redirected to `*.sam`, unaligned SAM text) on stdout,
`<stdout-stem>_summary.tsv` beside it.

Differences, all outside the hot path: reads come from `*.xsig.npz` signal bundles (no HDF5/VBZ reader in
this image, see reads.py); --reference / --modified-bases / --save-ctc are rejected (mappy / remora /
CTCWriter are not on the north-star path); under torchrun (WORLD_SIZE > 1) reads are sharded over the
ranks and gathered to rank 0 over RCCL before writing.
"""
import os
import sys
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser
from datetime import timedelta
from time import perf_counter

import numpy as np

from .. import dist as xdist
from ..io import Writer, biofmt
from ..reads import get_read_groups, get_reads
from ..util import column_to_set, init, load_model, load_symbol


READ_FIELDS = ("read_id", "run_id", "filename", "channel", "mux", "start", "duration", "template_start",
               "template_duration")


class _CalledRead:
    """What the writer needs of a read that was basecalled on another rank: its metadata and length, not its signal."""

    def __init__(self, fields, tags, n_samples):
        for k, v in zip(READ_FIELDS, fields):
            setattr(self, k, v)
        self._tags = tags
        self.signal = np.broadcast_to(np.float32(0), (n_samples,))     # len() only; no memory behind it

    def tagdata(self):
        return self._tags


def _gathered_results(results, loader, rank, world, window=256):
    """
    Multi-GPU: every rank basecalls its shard (reads i % world == rank); rank 0 receives (index, metadata, sequence)
    of all ranks in windows of `window` reads per rank and yields them in global read order, so nothing but metadata
    and called strings is kept, output starts while later windows are still being basecalled, and the collective is
    entered the same number of times by every rank.  A rank that fails keeps entering the remaining collectives with
    an error marker (its peers are never left blocked), then re-raises; rank 0 raises once it has seen the marker.
    """
    import torch.distributed as tdist
    n_windows = (loader.total + window * world - 1) // (window * world) if loader.total else 0
    it = iter(results)
    failure = None
    held = None
    for w in range(n_windows):
        hi = (w + 1) * window * world                   # global indices below `hi` belong to this window
        batch = []
        while failure is None:
            try:
                item = held if held is not None else next(it)
                held = None
            except StopIteration:
                break
            except BaseException as e:                   # noqa: BLE001 -- carried to rank 0, re-raised below
                failure = e
                break
            read, res = item
            if read.index >= hi:
                held = item
                break
            batch.append((read.index, tuple(getattr(read, k) for k in READ_FIELDS), read.tagdata(), len(read.signal),
                          res["sequence"], res["qstring"]))
        payload = ("error", repr(failure)) if failure is not None else ("ok", batch)
        gathered = [None] * world if rank == 0 else None
        tdist.gather_object(payload, gathered, dst=0)
        if rank == 0 and failure is None:
            bad = [p[1] for p in gathered if p[0] == "error"]
            if bad:
                # keep entering the remaining collectives (the other ranks do), raise at the end
                failure = RuntimeError("a rank failed while basecalling: %s" % "; ".join(bad))
                continue
            merged = sorted((rec for p in gathered for rec in p[1]), key=lambda rec: rec[0])
            for _, fields, tags, n_samples, seq, qstring in merged:
                yield _CalledRead(fields, tags, n_samples), {"sequence": seq, "qstring": qstring}
    if failure is not None:
        raise failure


def reader_procs(world=1):
    """Reader workers of this rank: the reference's 8 (cli/basecaller.py:107-111) when the host has them to give -- the cores
    this process may run on, divided by the ranks that share the node (LOCAL_WORLD_SIZE under torchrun, else the world size),
    one core per rank left for its own pipeline threads; XB_READER_PROCS overrides.  8 ranks x 8 workers on a 64-core node
    would otherwise put 72 busy processes on 64 cores and slow every rank's reader (host budget)."""
    env = os.environ.get("XB_READER_PROCS", "")
    if env:
        return max(1, int(env))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world) or 1))
    return max(1, min(8, cores // local - 1))


def main(args):
    if args.read_ids is not None and not os.path.isfile(args.read_ids):
        raise FileNotFoundError(args.read_ids)
    # the reader pool (8 worker processes, cli/basecaller.py:107-111) is forked BEFORE anything touches the GPU: the shard
    # comes from torchrun's environment alone, and the process group (whose nccl backend selects the device, i.e.
    # initialises HIP and starts runtime threads) is only joined once the pool exists.  Under torchrun every rank only
    # ever loads its own shard of the reads.
    rank, world = xdist.env_rank_world()
    n_proc = reader_procs(world)
    reads = get_reads(args.reads_directory, n_proc=n_proc, recursive=args.recursive,
                      read_ids=column_to_set(args.read_ids), skip=args.skip, limit=args.max_reads,
                      shard=(rank, world) if world > 1 else None)
    rank, world = xdist.init_from_env()

    init(args.seed, args.device)
    device = args.device
    if world > 1 and device == "cuda":
        device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", rank))

    sys.stderr.write(f"> loading model {args.model_directory}\n")
    try:
        model = load_model(args.model_directory, device, weights=int(args.weights), chunksize=args.chunksize,
                           overlap=args.overlap, batchsize=args.batchsize, quantize=args.quantize,
                           use_koi=args.use_koi)
    except FileNotFoundError:
        sys.stderr.write(f"> error: failed to load {args.model_directory}\n")
        exit(1)

    if args.verbose:
        sys.stderr.write(f"> model basecaller params: {model.config['basecaller']}\n")
        sys.stderr.write("> decode algorithm: %s\n" % ("Viterbi" if model.encoder[-1].expand_blanks else "Beam Search"))
        sys.stderr.write(f"> read_ids: {args.read_ids}\n")

    basecall = load_symbol(args.model_directory, "basecall")

    if args.reference or args.modified_bases or args.modified_base_model or args.save_ctc:
        sys.stderr.write("> error: --reference/--modified-bases/--save-ctc are not part of the MI355X path\n")
        exit(1)
    fmt = biofmt(aligned=False)
    sys.stderr.write(f"> outputting {fmt.aligned} {fmt.name}\n")
    if fmt.name not in ("fastq", "sam"):
        sys.stderr.write("> error: FASTQ and SAM text output are implemented (redirect stdout to *.fastq or *.sam); "
                         "BAM / CRAM need htslib\n")
        exit(1)
    # SAM: the header carries one @RG line per (run, model) of the selected reads -- metadata only (cli/basecaller.py:100-106)
    groups = []
    if fmt.name != "fastq" and rank == 0:
        groups = get_read_groups(args.reads_directory, args.model_directory, recursive=args.recursive,
                                 read_ids=column_to_set(args.read_ids), skip=args.skip, n_proc=n_proc)

    results = basecall(model, reads, reverse=args.revcomp,
                       batchsize=model.config["basecaller"]["batchsize"],
                       chunksize=model.config["basecaller"]["chunksize"],
                       overlap=model.config["basecaller"]["overlap"])

    t0 = perf_counter()
    if world > 1:
        results = _gathered_results(results, reads, rank, world)
        if rank != 0:
            for _ in results:                           # drives the local pipeline and the collectives
                pass
            return

    writer = Writer(fmt.mode, results, aligner=None, group_key=args.model_directory, groups=groups)
    writer.start()
    writer.join()
    if writer.error is not None:
        raise writer.error
    duration = perf_counter() - t0
    num_samples = sum(num_samples for read_id, num_samples in writer.log)

    sys.stderr.write(f"> completed reads: {len(writer.log):0,d}\n")
    sys.stderr.write("> duration: %s\n" % timedelta(seconds=np.round(duration)))
    sys.stderr.write("> samples per second %.1E\n" % (num_samples / duration))
    if args.verbose:
        # beyond the reference's lines: what the device stage did -- chunks of chunksize samples, overlaps and stub chunks included
        # (the reference's metric above counts READ samples); tools/cli_e2e.py compares this rate with bench.py's
        chunks = getattr(model, "chunks_submitted", 0)
        sys.stderr.write("> duration (s): %.2f\n" % duration)
        sys.stderr.write("> reads per second: %.0f (reader workers: %d)\n" % (len(writer.log) / duration, n_proc))
        sys.stderr.write("> chunks basecalled: %d x %d samples = %.3E chunk samples per second\n"
                         % (chunks, model.config["basecaller"]["chunksize"], chunks * model.config["basecaller"]["chunksize"] / duration))
    sys.stderr.write("> done\n")


def argparser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter, add_help=False)
    parser.add_argument("model_directory")
    parser.add_argument("reads_directory")
    parser.add_argument("--reference")
    parser.add_argument("--modified-bases", nargs="+")
    parser.add_argument("--modified-base-model")
    parser.add_argument("--read-ids")
    parser.add_argument("--device", default="cuda")
    parser.add_argument("--seed", default=25, type=int)
    parser.add_argument("--weights", default="0", type=str)
    parser.add_argument("--skip", action="store_true", default=False)
    parser.add_argument("--save-ctc", action="store_true", default=False)
    parser.add_argument("--revcomp", action="store_true", default=False)
    parser.add_argument("--recursive", action="store_true", default=False)
    quant_parser = parser.add_mutually_exclusive_group(required=False)
    quant_parser.add_argument("--quantize", dest="quantize", action="store_true")
    quant_parser.add_argument("--no-quantize", dest="quantize", action="store_false")
    quant_parser.add_argument("--no-use-koi", dest="use_koi", action="store_false")
    parser.set_defaults(quantize=None)
    parser.add_argument("--overlap", default=None, type=int)
    parser.add_argument("--chunksize", default=None, type=int)
    parser.add_argument("--batchsize", default=None, type=int,
                        help="chunks per device call; 512 and 1024 are the efficient sizes on MI355X (a recurrence launch serves "
                             "64-chunk groups, 8 or 16 at a time with a group on one XCD, up to 10 or 20 dealt over all XCDs: 513..640 "
                             "and 1025..1280 chunks cost up to 11 %% more per chunk, 641 chunks 25 %% more)")
    parser.add_argument("--max-reads", default=0, type=int)
    parser.add_argument("--min-accuracy", default=0.95, type=float)
    parser.add_argument("--min-coverage", default=0.90, type=float)
    parser.add_argument("--ub-only", action="store_true", default=False)
    parser.add_argument("-v", "--verbose", action="count", default=0)
    return parser
