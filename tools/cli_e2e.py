#!/usr/bin/env python
"""
End-to-end throughput of the drop-in CLI on synthetic reads (run on the GPU box):
writes a model directory (shipped geometry: features 768, 6-base CRF, seeded weights, chunksize 10000, batch 512) and a
signal bundle of READS reads x ~SAMPLES raw samples under /tmp, runs `python -m xna_basecaller_amd basecaller` with
stdout redirected to a .fastq file and reports the CLI's own "> samples per second" (cli/basecaller.py:153-161).
Usage: python tools/cli_e2e.py [--reads 2000] [--samples 50000] [--batch 512]
Round 5: --container fast5 writes multi-read fast5 files (tests/h5write.py: classic HDF5 layout, VBZ) instead of bundles and
--per-file sets the reads per container: the reference's real read shapes are POC ~3 000 samples (ONE left-padded chunk per
read, 4 000 reads per file) and CPLX ~25 000 samples (VERDICT r4 next 2 -> profiles/r05_cli_e2e_shapes.txt).
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--samples", type=int, default=50000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--chunksize", type=int, default=10000)
    ap.add_argument("--features", type=int, default=768)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--container", choices=("npz", "fast5"), default="npz")
    ap.add_argument("--per-file", type=int, default=250)
    ap.add_argument("--spread", type=float, default=0.4, help="read lengths are samples * U(1 - spread, 1 + spread)")
    ap.add_argument("--fuse", default="", help="comma list of XB_FUSE values to run the CLI under, on the same reads (e.g. 1,0)")
    args = ap.parse_args()
    import torch
    from xna_basecaller_amd import reads as xreads, toml_lite
    from xna_basecaller_amd.synthetic import seeded_weights

    work = tempfile.mkdtemp(prefix="xb_e2e_", dir="/tmp")
    model_dir = os.path.join(work, "xna_synth@v1")
    reads_dir = os.path.join(work, "reads")
    os.makedirs(model_dir)
    os.makedirs(reads_dir)
    cfg = {"global_norm": {"state_len": 3}, "qscore": {"bias": 0.3498, "scale": 0.9722}, "input": {"features": 1},
           "model": {"package": "bonito.crf"}, "labels": {"labels": list("NACGTXY")},
           "encoder": {"stride": 5, "activation": "swish", "features": args.features, "winlen": 19, "scale": 5.0,
                       "rnn_type": "lstm", "blank_score": 2.0},
           "basecaller": {"batchsize": args.batch, "chunksize": args.chunksize, "overlap": 500}}
    with open(os.path.join(model_dir, "config.toml"), "w") as fh:
        fh.write(toml_lite.dumps(cfg))
    sd = seeded_weights(args.features, 6)
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, os.path.join(model_dir, "weights_1.tar"))

    rng = np.random.default_rng(3)
    t0 = time.time()
    per_file = args.per_file
    if args.container == "fast5":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from h5write import write_multi_fast5
    for f0 in range(0, args.reads, per_file):
        recs = []
        for i in range(f0, min(f0 + per_file, args.reads)):
            length = int(args.samples * rng.uniform(1.0 - args.spread, 1.0 + args.spread))
            base = rng.normal(90.0, 12.0, length)
            lead = int(rng.integers(300, 900))
            base[:lead] = rng.normal(140.0, 3.0, lead)
            raw = np.round(base * 8.0).astype(np.int16)
            recs.append((raw, dict(read_id="read-%06d" % i, range=1443.03, digitisation=8192.0, offset=10,
                                   sampling_rate=4000.0, run_id="runX", channel_number=str(1 + i % 512),
                                   start_mux=1 + i % 4, read_number=i, start_time=4000 * i, duration=length,
                                   exp_start_time="2021-06-01T10:00:00Z")))
        if args.container == "fast5":
            write_multi_fast5(os.path.join(reads_dir, "batch%04d.fast5" % (f0 // per_file)), recs, vbz=True)
        else:
            xreads.write_bundle(os.path.join(reads_dir, "batch%04d.xsig.npz" % (f0 // per_file)), recs)
    print("wrote %d reads (%s, %d per file) in %.1f s" % (args.reads, args.container, per_file, time.time() - t0), flush=True)

    out = os.path.join(work, "calls.fastq")
    rc, digests = 0, []
    for fuse in (args.fuse.split(",") if args.fuse else [None]):
        env = dict(os.environ)
        if fuse is not None:
            env["XB_FUSE"] = fuse
        t0 = time.time()
        with open(out, "w") as fh:
            r = subprocess.run([sys.executable, "-m", "xna_basecaller_amd", "basecaller", model_dir, reads_dir, "-v"], cwd=ROOT,
                               stdout=fh, stderr=subprocess.PIPE, env=env)
        wall = time.time() - t0
        err = r.stderr.decode()
        print("== XB_FUSE=%s (reads %d x ~%d samples, %s, %d per file, batch %d)"
              % (fuse, args.reads, args.samples, args.container, per_file, args.batch))
        print("\n".join(l for l in err.splitlines() if l.startswith(">") and "model basecaller params" not in l)[-1500:])
        import hashlib
        digests.append(hashlib.sha1(open(out, "rb").read()).hexdigest())
        print("cli wall (incl. start-up and model load): %.1f s, rc %d, fastq %d bytes, sha1 %s"
              % (wall, r.returncode, os.path.getsize(out), digests[-1][:12]), flush=True)
        rc = rc or r.returncode
    if len(set(digests)) > 1:
        print("FASTQ differs between the schedules")
        rc = rc or 1
    if not args.keep:
        shutil.rmtree(work, ignore_errors=True)
    sys.exit(rc)


if __name__ == "__main__":
    main()
