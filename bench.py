#!/usr/bin/env python
"""
bench.py -- headline benchmark of the MI355X basecalling hot path.

Metric (BASELINE.json): raw signal samples/s basecalled, chunksize 10 000.
One "step" = one pass of the whole hot path (conv front-end -> 5 LSTM layers -> CRF linear ->
posterior + max-plus decode -> left-packed called sequences) over one batch of synthetic chunks
that is already resident in HBM.  Default workload = BASELINE.json configs[2]: the shipped 6-base CRF
(labels N A C G T X Y), chunksize 10000, batch 512 per GPU (--nbase 5 gives configs[1]).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; every rank basecalls its own batch (reads shard, weak scaling) and the
called sequences are gathered with ONE all-gather per step over RCCL (the path's only exchange: xb_gather_called of the C
ABI, on the communicator's own stream behind an event), so that N = 1 and N > 1 time the same device pipeline.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak


def cpu_baseline(sd, features, n_base, L, chunks, alphabet, repeats=3):
    """The oracle (a port: the reference's own decode is CUDA-only) timed on this host's cores: one warm-up pass, then
    the median of `repeats` timed passes over the same bounded sample (SURVEY.md 8d / BASELINE.md 4).  The oracle is
    imported HERE only: nothing else in this file touches it."""
    import oracle
    # a one-GPU box's CPU share is 16 cores; more OpenMP threads than that only adds overhead here
    oracle.set_num_threads(min(os.cpu_count() or 1, 16))
    x = np.random.default_rng(25).standard_normal((chunks, L)).astype(np.float32)

    def one_pass(xs):
        t0 = time.perf_counter()
        sc = oracle.encode(xs, sd, features, n_base, 3, expand_blanks=False)
        lab = oracle.decode(sc, n_base, 3, blank_score=2.0)["labels"]
        oracle.pack(lab, alphabet)
        return time.perf_counter() - t0

    one_pass(x[:max(1, chunks // 8)])                      # warm-up (page in, OpenMP pool)
    times = sorted(one_pass(x) for _ in range(repeats))
    dt = times[len(times) // 2]
    return {"value": chunks * L / dt, "unit": "samples/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "%d chunks x %d samples, %d-base CRF, fp32 C oracle (OpenMP), median of %d passes: %s s"
                      % (chunks, L, n_base, repeats, "/".join("%.1f" % t for t in times))}


def measured_traffic(kernel_prefix, nb, batch, chunksize, precision, launches_per_step, fused):
    """(HBM bytes per launch, provenance) from the committed PMC passes (profiles/*_pmc_hbm_traffic.json: FETCH_SIZE / WRITE_SIZE
    collected in separate rocprofv3 runs, gfx950 correction applied).  The number is a REPLAY of a profile, not a count taken in
    this run (counters cannot be collected inside the timed region), so it is only quoted when that profile was collected on
    THIS code (the library's source digest, _lib.source_digest) and on this configuration and call pairing; otherwise null,
    with the reason.  The passes count bytes per bench step; they are divided by THIS run's launches per step (counter collection
    runs the recurrence as slab launches, the timed run as one launch per layer: the bytes are the same, the launch count not)."""
    import glob
    from xna_basecaller_amd import _lib
    digest = _lib.source_digest()
    why = "no profile for this configuration under profiles/"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        c = d.get("config", {})
        if (c.get("n_base"), c.get("batch_per_gpu"), c.get("chunksize"), c.get("precision")) != (nb, batch, chunksize, precision):
            continue
        src = {"file": os.path.relpath(path, ROOT), "profile_source_digest": d.get("source_digest"), "running_source_digest": digest}
        if d.get("source_digest") != digest:
            why = "%s was collected on other code (source digest %s, running %s)" % (src["file"], d.get("source_digest"), digest)
            continue
        if bool(c.get("fuse", 1)) != bool(fused):
            why = "%s was collected with%s call pairing" % (src["file"], "" if c.get("fuse", 1) else "out")
            continue
        for name, k in d.get("kernels", {}).items():
            if name.startswith(kernel_prefix) and launches_per_step > 0 and "steps" in d:
                src["kernel"] = name
                return k.get("hbm_bytes_all_launches") / d["steps"] / launches_per_step, src
    return None, {"reason": why, "running_source_digest": digest}


def main():
    # exactly ONE line on stdout (the contract): libraries that print banners from C (RCCL's version header at the first
    # communicator) get stderr as their stdout; the JSON line goes to a duplicate of the real one
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="chunks per GPU per step")
    ap.add_argument("--chunksize", type=int, default=10000)
    ap.add_argument("--nbase", type=int, default=6, choices=[4, 5, 6])
    ap.add_argument("--features", type=int, default=768)
    ap.add_argument("--precision", default="mixed", choices=["mixed", "f16x3", "f16", "f16f8", "f16f8i"],
                    help="mixed (the product default): feed-forward projections in f16x3, recurrent ones in f16f8")
    ap.add_argument("--cpu-chunks", type=int, default=64, help="chunks in the bounded cpu_baseline sample (0 = skip)")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--lstm-mode", type=int, default=0)
    ap.add_argument("--weights", default="seeded", choices=["seeded", "peaky"],
                    help="seeded: N(0, 1/sqrt(fan_in)) weights (flat posteriors: every step calls a base); peaky: "
                         "synthetic.peaky_weights (scores follow the signal, ~0.5 bases per step: a trained model's regime)")
    ap.add_argument("--gather", default="xb", choices=["xb", "torch"],
                    help="N > 1: xb = xb_gather_called over librccl (C ABI); torch = torch.distributed all_gather")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the deferred side-stream gather plumbing even with one rank (harness self-test)")
    args = ap.parse_args()

    import torch
    from xna_basecaller_amd import _lib
    from xna_basecaller_amd import dist as xdist
    from xna_basecaller_amd.synthetic import peaky_weights, seeded_weights

    rank, world = xdist.init_from_env()
    if world != max(args.gpus, 1):
        raise SystemExit("--gpus %d but WORLD_SIZE is %d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    local = int(os.environ.get("LOCAL_RANK", rank))
    _lib.require_gpu()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    nb, L, N, F = args.nbase, args.chunksize, args.batch, args.features
    alphabet = "NACGTXY"[:nb + 1]
    S, E = nb ** 3, nb + 1
    prec = _lib.PRECISIONS[args.precision]
    ctx = _lib.Context(local, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=prec, lstm_mode=args.lstm_mode)
    sd = peaky_weights(F, nb) if args.weights == "peaky" else seeded_weights(F, nb)
    ctx.load_state_dict(sd)
    ctx.reserve_pairing()           # two calls in flight are co-scheduled: their workspaces now, not inside the timed region
    T = ctx.T

    # synthetic signal ~ N(0,1) generated on the device (Philox counter RNG, seeded by (25, rank)): resident in HBM
    gen = torch.Generator(device=dev)
    gen.manual_seed(25 + 1000003 * rank)
    d_signal = torch.randn((N, L), dtype=torch.float32, device=dev, generator=gen)
    # two output buffer sets in rotation: batch k's sequences are gathered (side stream) while batch k+1 computes
    d_seq = [torch.empty((N, T), dtype=torch.int8, device=dev) for _ in range(2)]
    d_len = [torch.empty((N,), dtype=torch.int32, device=dev) for _ in range(2)]
    # The path's one exchange: the gather of called sequences, through the C ABI (xb_comm: RCCL all-gathers on
    # the communicator's own stream behind the context's result stream).  --gather torch keeps the torch.distributed route
    # (the same RCCL underneath) and is also the fallback should the communicator not come up.
    gather, gather_kind = None, "none"
    if world > 1 or args.force_gather:
        if args.gather == "xb":
            try:
                gather, gather_kind = xdist.RcclGather(ctx, local, rank, world), "xb_gather_called (librccl via the C ABI)"
            except Exception as e:                                   # noqa: BLE001 -- reported, then the torch route
                sys.stderr.write("rank %d: xb_comm unavailable (%s); gathering through torch.distributed\n" % (rank, e))
        if world > 1 and args.gather == "xb":                        # all ranks take the same route
            import torch.distributed as tdist
            agree = torch.tensor([1 if gather is not None else 0], dtype=torch.int32, device=dev)
            tdist.all_reduce(agree, op=tdist.ReduceOp.MIN)
            if int(agree.item()) == 0 and gather is not None:
                gather.close()
                gather = None
        if gather is None:
            gather, gather_kind = xdist.DeferredGather(), "torch.distributed all_gather (nccl backend = RCCL)"
    use_xb = isinstance(gather, xdist.RcclGather)
    state = {"k": 0}

    def step():
        k = state["k"]
        state["k"] = k + 1
        b = k & 1
        if use_xb:
            gather.before_batch()                                # the gather that last read buffer set b is done (device-side)
        elif gather is not None:
            out_stream = torch.cuda.ExternalStream(ctx.result_stream(), device=dev)
            ev = gather.consumed(k - 2)                      # the gather that last read this buffer set
            if ev is not None:
                out_stream.wait_event(ev)
        ctx.basecall_chunks_dev(d_signal.data_ptr(), N, alphabet, d_seq[b].data_ptr(), d_len[b].data_ptr())
        if use_xb:
            gather.submit(d_seq[b], d_len[b])                    # starts on the device when batch k is complete
        elif gather is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.ExternalStream(ctx.result_stream(), device=dev))
            gather.submit(d_seq[b], d_len[b], ready)         # starts the gather of batch k-1

    def fence():
        if gather is not None:
            gather.flush()
        ctx.synchronize()
        torch.cuda.synchronize()
        xdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_profiling(True)
    ctx.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as tdist
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        dt = float(tmax.item())
    stages = ctx.stage_times()
    ctx.set_profiling(False)
    called = int(d_len[(state["k"] - 1) & 1].sum().item())

    if rank != 0:
        return
    K = args.steps
    samples = world * N * L * K
    value = samples / dt
    ms_step = 1e3 * dt / K

    # ---- roofline of the dominant kernel (by device time): the LSTM recurrence -----------------------
    # algorithmic FLOPs per launch = one layer's recurrent projection h_{t-1} W_hh^T over the launch's chunks:
    #   LSTM = 9 437 184 FLOP per raw sample for 5 layers x (input + recurrent) projections
    #   -> 943 718.4 FLOP per sample per recurrent projection; x (N*L) samples per launch.
    rec_ms, rec_launches = stages["lstm_rec"]
    flop_launch = 5 * 2.0 * (4 * F) * F * T * N / max(rec_launches / K, 1)   # a layer may run as several time-slab launches
    rec_avg_s = 1e-3 * rec_ms / max(rec_launches, 1)
    rec_tflops = flop_launch / rec_avg_s / 1e12 if rec_avg_s > 0 else 0.0
    # two chunk groups per workgroup: batches above 640 chunks, or two consecutive asynchronous calls of at most
    # 640 chunks co-scheduled by the library (XB_FUSE) -- visible here as half as many launches as calls
    chunks_per_launch_factor = 5.0 * K / max(rec_launches, 1)       # calls served per recurrence launch of a layer (time slabs: < 1)
    fused = chunks_per_launch_factor > 1.5
    # (round 5: 513..640 chunks run with ONE group per workgroup over up to ten group slots dealt over all XCDs, xb_api.hip WIDE)
    wide_single = 512 < N <= 640 and not fused and os.environ.get("XB_LSTM_WIDE", "1") != "0"
    dual = (N > 512 or fused) and not wide_single and os.environ.get("XB_LSTM_DUAL", "1") != "0"
    rec_traffic, rec_traffic_src = measured_traffic("lstm_kernel", nb, N, L, args.precision, rec_launches / float(K), fused)
    # template arguments as rocprofv3 prints them: KS, NSPLIT, DUAL, YALT (mixed: the q8 recurrence writes the fp16 residual the
    # three-product GEMMs read, xb_lstm.hip lstm_kernel)
    yalt = "true" if prec == _lib.XB_PREC_MIXED else "false"
    rec_kernel = "lstm_kernel<%d, %d, %s, %s>" % (F // 16, {0: 3, 1: 1, 2: 2, 3: 2, 4: 2}[prec], "true" if dual else "false", yalt)
    roofline = {"kernel": rec_kernel,
                "bound": "mfma",
                "achieved": rec_tflops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": rec_tflops / MFMA_F16_PEAK_TFLOPS,
                "traffic": rec_traffic, "traffic_source": rec_traffic_src,
                # what a launch has to move: its gin tiles in (fp32, 4F per chunk and step), the layer output out (hi + second part)
                "algorithmic_bytes": 5.0 * K / max(rec_launches, 1) * float(T) * N * (4 * F * 4 + F * 4),
                "avg_launch_ms": 1e3 * rec_avg_s, "launches": rec_launches, "launches_per_step": rec_launches / K,
                "note": "algorithmic fp32-equivalent FLOPs; f16x3 issues 3 fp16 MFMA products per FLOP pair, "
                        "f16f8 one fp16 product + one block-scaled FP8 MFMA (2x rate) for both corrections"}
    # ---- CRF decode: HBM roofline (the north-star target) -------------------------------------------------
    dec_ms, dec_launches = stages["decode"]
    # A_dec bytes per launch (a launch decodes the chunks of one call, or of two co-scheduled calls)
    a_dec = float(T) * N * (3 * S * E * 4 + 7 * S * 4 + 1) * (K / max(dec_launches, 1))
    dec_avg_s = 1e-3 * dec_ms / max(dec_launches, 1)
    dec_gbs = a_dec / dec_avg_s / 1e9 if dec_avg_s > 0 else 0.0
    dec_traffic, dec_traffic_src = measured_traffic("crf_decode_kernel", nb, N, L, args.precision, dec_launches / float(K), fused)
    roofline_decode = {"kernel": "crf_decode_kernel", "bound": "hbm", "achieved": dec_gbs, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": dec_gbs / HBM_PEAK_GBS,
                       "traffic": dec_traffic, "traffic_source": dec_traffic_src, "algorithmic_bytes": a_dec,
                       "avg_launch_ms": 1e3 * dec_avg_s, "launches": dec_launches,
                       "decode_only_samples_per_s": N * L / dec_avg_s if dec_avg_s > 0 else 0.0}

    # ---- the LSTM-input GEMMs: the largest kernel by SUMMED device time (they run slab by slab beside the recurrence, so their
    #      in-situ durations include what the two kernels cost each other; XB_OVERLAP=0 gives the stand-alone figure)
    gin_ms, gin_launches = stages["lstm_in"]
    gemm_flop_step = 5 * 2.0 * T * N * (4 * F) * F                      # 4 718 592 FLOP per raw sample
    gemm_tflops = gemm_flop_step * K / (1e-3 * gin_ms) / 1e12 if gin_ms > 0 else 0.0
    products = {0: 3, 1: 1, 2: 2, 3: 2, 4: 3}[prec]                     # MFMA-rate units per algorithmic FLOP pair (f16f8: fp16 + FP8 at 2x)
    roofline_gemm = {"kernel": "gemm4p_kernel<0, %d>" % {0: 3, 1: 1, 2: 2, 3: 2, 4: 3}[prec], "bound": "mfma",
                     "achieved": gemm_tflops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tflops / MFMA_F16_PEAK_TFLOPS,
                     "mfma_rate_frac": products * gemm_tflops / MFMA_F16_PEAK_TFLOPS,
                     "stage_ms_per_step": gin_ms / K, "launches_per_step": gin_launches / K,
                     "note": "algorithmic fp32-equivalent FLOPs of the five input projections over their summed HIP-event time; "
                             "mfma_rate_frac counts the products the arithmetic issues per FLOP pair"}

    out = {
        "metric": "raw signal samples/sec basecalled, chunksize 10k", "value": value, "unit": "samples/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {0: "f32 (split-f16x3 MFMA, f32 accumulate; CRF decode f32)",
                  1: "f16 MFMA, f32 accumulate; CRF decode f32",
                  2: "f32 (f16 MFMA + FP8 block-scaled correction MFMA, f32 accumulate; CRF decode f32)",
                  3: "f32 (as f16f8, LSTM input projections f16 MFMA only, f32 accumulate; |score err| <= 1e-3; CRF decode f32)",
                  4: "f32 (feed-forward projections split-f16x3 MFMA, recurrent projections f16 MFMA + FP8 block-scaled correction "
                     "MFMA, f32 accumulate; CRF decode f32)"}[prec],
        "data": "synthetic N(0,1) signal chunks generated in HBM; " +
                ("seeded N(0,1/sqrt(fan_in)) weights in the reference state-dict layout" if args.weights == "seeded" else
                 "synthetic.peaky_weights (zero LSTM biases, input gain 2, CRF linear gain 10 / bias -2) in the reference state-dict layout"),
        "config": {"workload": "BASELINE configs[%s]: %d-base CRF (S=%d, C=%d), chunksize %d, batch %d per GPU, features %d"
                               % ({(5, 512): "1", (6, 512): "2", (6, 1024): "3] per-GPU workload [1 of 8 ranks",
                                   (6, 2048): "4] per-GPU workload [1 of 8 ranks"}.get((nb, N), "-"), nb, S, S * E, L, N, F),
                   "n_base": nb, "chunksize": L, "batch_per_gpu": N, "T": T, "parallelism": "reads sharded x%d" % world,
                   "collective": ("all_gather of packed sequences per step on a side stream: " + gather_kind) if gather is not None else "none",
                   "in_flight": ("a step = one asynchronous xb_basecall_chunks_dev of %d chunks; the library co-schedules two consecutive "
                                 "calls through one pass of the encoder and the decode (two chunk groups per recurrence workgroup), "
                                 "XB_FUSE=0 runs every call on its own" % N) if fused else
                                "a step = one asynchronous xb_basecall_chunks_dev of %d chunks, each run on its own" % N},
        "roofline": roofline, "roofline_decode": roofline_decode, "roofline_gemm": roofline_gemm,
        "stage_ms_per_step": {k: v[0] / K for k, v in stages.items()},
        "stage_note": "HIP-event time per stage on its own stream; lstm_in / linear run slab by slab on a second stream "
                      "beside the previous layer's recurrence, so the stages overlap and do not add up to ms_per_step",
        "called_bases_last_step": called, "called_bases_per_time_step": called / float(N * T),
    }
    if world == 1 and args.cpu_chunks > 0:
        out["cpu_baseline"] = cpu_baseline(sd, F, nb, L, args.cpu_chunks, alphabet, args.cpu_repeats)
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()


if __name__ == "__main__":
    main()
