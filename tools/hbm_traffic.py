#!/usr/bin/env python
"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_gemm.sh (PASSES=2) into the per-kernel HBM traffic
summary that bench.py's `roofline.traffic` reads (profiles/*_pmc_hbm_traffic.json).
usage: python tools/hbm_traffic.py gpurun_out/<pmcdir> profiles/rNN_xx_pmc_hbm_traffic.json --nbase 6 --batch 512"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xna_basecaller_amd import _lib           # noqa: E402  (source_digest only: no GPU is touched)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmcdir")
    ap.add_argument("out")
    ap.add_argument("--nbase", type=int, default=6)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--chunksize", type=int, default=10000)
    ap.add_argument("--precision", default="f16f8")
    ap.add_argument("--note", default="")
    ap.add_argument("--steps", type=int, default=2, help="bench steps the PMC passes covered (tools/pmc_gemm.sh: 2)")
    ap.add_argument("--fuse", type=int, default=1, help="the passes ran with two calls co-scheduled per device pass (bench.py default)")
    args = ap.parse_args()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(args.pmcdir + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            name = re.sub(r"^void \(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for name, d in agg.items():
        if not any(k in name for k in ("conv_front", "gemm8r", "gemm4p", "lstm_kernel", "crf_decode")):
            continue
        n = max(len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"]), 1)
        fetch = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
        write = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
        per = (2.0 * fetch + write) * 1024.0            # gfx950: FETCH_SIZE counts 128-B requests at 64 B
        kernels[name] = {"launches": n, "fetch_size_kib": fetch, "write_size_kib": write,
                         "hbm_bytes_per_launch": per, "hbm_bytes_all_launches": per * n}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_gemm.sh with PASSES=2) over "
                     "`bench.py --steps 2 --warmup 0 --cpu-chunks 0`; MI355X. Per-launch means. " + args.note,
           "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for 16-B-per-lane "
                         "loads -> fetch bytes = 2 x FETCH_SIZE KiB x 1024; WRITE_SIZE is exact for 16-B-per-lane stores. "
                         "Infinity-Cache hits are counted.",
           "steps": args.steps,
           "config": {"n_base": args.nbase, "batch_per_gpu": args.batch, "chunksize": args.chunksize,
                      "precision": args.precision, "fuse": args.fuse},
           # the code the counters were collected on: bench.py quotes them only while the library's sources still hash to this
           "source_digest": _lib.source_digest(),
           "schedule_note": "counter collection serialises kernels and runs the recurrence as slab launches (XB_LSTM_SIGNAL off "
                            "under ROCPROF_COUNTER_COLLECTION): the BYTES a step moves are those of the timed schedule, the launch "
                            "count is not -- bench.py divides the bytes per step by its own launches per step",
           "kernels": kernels}
    json.dump(out, open(args.out, "w"), indent=1)
    for k, v in kernels.items():
        print("%-50s launches %3d  %.3f GB per launch" % (k[:50], v["launches"], v["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
