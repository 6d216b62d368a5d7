"""Reader throughput on the reference's real read shapes: N reads of 2-4 k samples (POC: 106-nt templates)
or ~25 k samples (CPLX) in ONE multi-read fast5 (written by tests/h5write.py, VBZ; --libhdf5: by the real libhdf5, deflate) and as
an .xsig.npz bundle; prints reads/s of get_reads(..., n_proc=P).  CPU only."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def records(n, lo, hi, seed=3):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        m = int(rng.integers(lo, hi))
        raw = (rng.standard_normal(m) * 60 + 480).astype(np.int16)
        raw[:200] += 300                                              # an open-pore prefix for trim() to find
        recs.append((raw, {"read_id": "%08x-0000-4000-8000-%012x" % (i, i * 7919), "range": 1437.0, "digitisation": 8192.0,
                           "offset": 6.0, "sampling_rate": 4000.0, "run_id": "run0", "channel_number": str(1 + i % 512),
                           "start_mux": 1 + i % 4, "read_number": i, "start_time": 4000 * i, "duration": m,
                           "exp_start_time": "2021-03-01T10:00:00Z", "sample_id": "poc", "flow_cell_id": "FAK1", "device_id": "MN1"}))
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--lo", type=int, default=2000)
    ap.add_argument("--hi", type=int, default=4000)
    ap.add_argument("--procs", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--dir", default="/tmp/rb")
    ap.add_argument("--libhdf5", action="store_true")
    ap.add_argument("--kinds", nargs="+", default=["fast5", "npz"])
    args = ap.parse_args()
    from xna_basecaller_amd import reads as xreads
    d5, dn = os.path.join(args.dir, "f5_%d_%d" % (args.reads, args.hi)), os.path.join(args.dir, "npz_%d_%d" % (args.reads, args.hi))
    recs = None
    if not os.path.isdir(d5):
        recs = records(args.reads, args.lo, args.hi)
        os.makedirs(d5)
        if args.libhdf5:
            import h5lib
            h5lib.write_multi_fast5(os.path.join(d5, "batch_0.fast5"), recs, chunk=4096)
        else:
            from h5write import write_multi_fast5
            write_multi_fast5(os.path.join(d5, "batch_0.fast5"), recs, vbz=True)
    if not os.path.isdir(dn):
        recs = recs or records(args.reads, args.lo, args.hi)
        os.makedirs(dn)
        xreads.write_bundle(os.path.join(dn, "all.xsig.npz"), recs)
    for kind, d in (("fast5", d5), ("npz", dn)):
        if kind not in args.kinds:
            continue
        for p in args.procs:
            t0 = time.time()
            n = samples = 0
            for r in xreads.get_reads(d, n_proc=p):
                n += 1
                samples += len(r.signal)
            dt = time.time() - t0
            print("%-5s n_proc=%d: %d reads, %.2f s, %.0f reads/s, %.2e samples/s" % (kind, p, n, dt, n / dt, samples / dt), flush=True)


if __name__ == "__main__":
    main()
