"""
`bonito basecaller`-compatible command line (ub-bonito/bonito/cli/basecaller.py:24-196): same positional
arguments, flags, defaults and stderr lines; FASTQ on stdout, `<stdout-stem>_summary.tsv` beside it.

Differences, all outside the hot path: reads come from `*.xsig.npz` signal bundles (no HDF5/VBZ reader in
this image, see reads.py); --reference / --modified-bases / --save-ctc are rejected (mappy / remora /
CTCWriter are not on the north-star path); under torchrun (WORLD_SIZE > 1) reads are sharded over the
ranks and gathered to rank 0 over RCCL before writing.
"""
import os
import sys
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser
from datetime import timedelta
from itertools import islice as take
from time import perf_counter

import numpy as np

from .. import dist as xdist
from ..io import Writer, biofmt
from ..reads import get_reads
from ..util import column_to_set, init, load_model, load_symbol


def main(args):
    init(args.seed, args.device)
    rank, world = xdist.init_from_env()
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
        sys.stderr.write("> decode algorithm: Viterbi\n")
        sys.stderr.write(f"> read_ids: {args.read_ids}\n")

    basecall = load_symbol(args.model_directory, "basecall")

    if args.reference or args.modified_bases or args.modified_base_model or args.save_ctc:
        sys.stderr.write("> error: --reference/--modified-bases/--save-ctc are not part of the MI355X path\n")
        exit(1)
    fmt = biofmt(aligned=False)
    sys.stderr.write(f"> outputting {fmt.aligned} {fmt.name}\n")
    if fmt.name != "fastq":
        sys.stderr.write("> error: only FASTQ output is implemented (redirect stdout to *.fastq)\n")
        exit(1)
    if args.read_ids is not None and not os.path.isfile(args.read_ids):
        raise FileNotFoundError(args.read_ids)

    reads = get_reads(args.reads_directory, n_proc=8, recursive=args.recursive,
                      read_ids=column_to_set(args.read_ids), skip=args.skip)
    if args.max_reads:
        reads = take(reads, args.max_reads)

    index_of = {}
    if world > 1:
        def local_reads():
            for i, read in xdist.shard(reads, rank, world):
                index_of[id(read)] = i
                yield read
        reads_in = local_reads()
    else:
        reads_in = reads

    results = basecall(model, reads_in, reverse=args.revcomp,
                       batchsize=model.config["basecaller"]["batchsize"],
                       chunksize=model.config["basecaller"]["chunksize"],
                       overlap=model.config["basecaller"]["overlap"])

    t0 = perf_counter()
    if world > 1:
        # every rank basecalls its shard; one RCCL gather brings (index, read, sequence) to rank 0
        local, by_index = [], {}
        for read, res in results:
            i = index_of[id(read)]
            local.append((i, read.read_id, res["sequence"], res["qstring"]))
            by_index[i] = read
        meta = [(i, r.read_id, r.run_id, r.filename, str(r.channel), int(r.mux), float(r.start), float(r.duration),
                 float(r.template_start), float(r.template_duration), r.tagdata(), len(r.signal))
                for i, r in by_index.items()]
        import torch.distributed as tdist
        metas = [None] * world
        tdist.all_gather_object(metas, meta)
        merged = xdist.gather_called(local, dst=0)
        if rank != 0:
            return
        from ..reads import SyntheticRead
        lookup = {}
        for m in (x for ms in metas for x in ms):
            r = SyntheticRead(m[1], np.zeros(m[11], np.float32), run_id=m[2], filename=m[3], channel=m[4], mux=m[5],
                              start=m[6])
            r.duration, r.template_start, r.template_duration = m[7], m[8], m[9]
            r.tagdata = (lambda tags: (lambda: tags))(m[10])
            lookup[m[0]] = r
        results = ((lookup[i], {"sequence": seq, "qstring": q}) for i, _, seq, q in merged)

    writer = Writer(fmt.mode, results, aligner=None, group_key=args.model_directory)
    writer.start()
    writer.join()
    if writer.error is not None:
        raise writer.error
    duration = perf_counter() - t0
    num_samples = sum(num_samples for read_id, num_samples in writer.log)

    sys.stderr.write(f"> completed reads: {len(writer.log):0,d}\n")
    sys.stderr.write("> duration: %s\n" % timedelta(seconds=np.round(duration)))
    sys.stderr.write("> samples per second %.1E\n" % (num_samples / duration))
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
    parser.add_argument("--batchsize", default=None, type=int)
    parser.add_argument("--max-reads", default=0, type=int)
    parser.add_argument("--min-accuracy", default=0.95, type=float)
    parser.add_argument("--min-coverage", default=0.90, type=float)
    parser.add_argument("--ub-only", action="store_true", default=False)
    parser.add_argument("-v", "--verbose", action="count", default=0)
    return parser
