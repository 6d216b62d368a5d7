"""GPU: the reference-shaped pipeline (Model / compute_scores / basecall) end to end against the oracle."""
import numpy as np
import pytest

import oracle
from conftest import encoder_shapes, make_config, seeded_state_dict
from xna_basecaller_amd import util
from xna_basecaller_amd.crf import Model, basecall
from xna_basecaller_amd.crf.basecall import compute_scores
from xna_basecaller_amd.reads import SyntheticRead

pytestmark = pytest.mark.gpu


def _model(features=64, labels="NACGTXY", seed=3):
    import torch
    model = Model(make_config(features, labels))
    keys, shapes = encoder_shapes(features, len(labels) - 1)
    sd = seeded_state_dict(keys, shapes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model.eval().to("cuda"), sd


def test_model_forward_and_decode_batch():
    model, sd = _model()
    x = np.random.default_rng(0).standard_normal((4, 1, 1000)).astype(np.float32)
    scores = model(x)
    assert scores.shape == (200, 4, 1512)
    ref = oracle.encode(x, sd, 64, 6, 3)
    assert np.abs(scores - ref).max() < 2e-4
    # decode the GPU's own scores with both implementations: identical strings
    assert model.decode_batch(scores) == oracle.decode_batch(scores, list("NACGTXY"), 3)


def test_compute_scores_layout():
    model, sd = _model()
    x = np.random.default_rng(1).standard_normal((3, 1, 1000)).astype(np.float32)
    out = compute_scores(model, x)
    assert out["sequence"].shape == (3, 200) and out["sequence"].dtype == np.int8
    assert out["qstring"].dtype == np.int8 and out["moves"].dtype == bool and not out["moves"].any()
    assert np.array_equal(out["qstring"] != 0, out["sequence"] != 0)
    assert set(np.unique(out["qstring"])) <= {0, 79}
    # fused path == forward + decode_batch
    seqs = model.decode_batch(model(x))
    for i in range(3):
        n = int((out["sequence"][i] != 0).sum())
        assert out["sequence"][i, :n].tobytes().decode() == seqs[i]
        assert not out["sequence"][i, n:].any()


def test_basecall_reads_end_to_end():
    """Mixed read lengths (shorter than a chunk, exactly one chunk, stub and no-stub multi-chunk) in input order."""
    model, sd = _model(features=32, labels="NACGTX", seed=9)
    L, ov, bs = 1000, 100, 7
    rng = np.random.default_rng(4)
    lens = [400, 1000, 1900, 2350, 2800, 5100, 999, 3700]
    reads = [SyntheticRead("read%d" % i, rng.standard_normal(n).astype(np.float32)) for i, n in enumerate(lens)]
    got = list(basecall(model, iter(reads), chunksize=L, overlap=ov, batchsize=bs))
    assert [r.read_id for r, _ in got] == [r.read_id for r in reads]
    total = mism = 0
    for read, res in got:
        ch = util.chunk(read.signal, L, ov)
        sc_gpu = model(ch)
        # exact: oracle decode of the GPU scores, packed and stitched by the same host code
        lab = oracle.decode(sc_gpu, 5, 3)["labels"]
        seq, qs, _ = oracle.pack(lab, "NACGTX")
        st = util.stitch(seq, L, ov, len(read.signal), 5)
        expect = st[st != 0].astype(np.uint8).tobytes().decode()
        assert res["sequence"] == expect
        assert res["qstring"] == "O" * len(expect)
        assert res["sig_move"].shape == (util.stitch(seq, L, ov, len(read.signal), 5).size * 5,) and not res["sig_move"].any()
        # end to end vs the all-CPU oracle (encoder differences of ~1e-5 may flip a near-tie)
        lab2 = oracle.decode(oracle.encode(ch, sd, 32, 5, 3), 5, 3)["labels"]
        total += lab.size
        mism += int((lab != lab2).sum())
    assert mism <= total // 500, (mism, total)


@pytest.mark.parametrize("decode_async", ["0", "1"])
def test_pipelined_device_calls_match_blocking_calls(decode_async, monkeypatch):
    """Several xb_basecall_chunks_dev calls in flight (time-slab recurrence with the next layer's GEMM on the second stream; the
    decode of batch k on the main stream -- the default -- or, XB_DECODE_ASYNC=1, on a third stream beside the encoder of
    batch k+1 with ping-pong score buffers), ONE synchronize at the end: every batch must equal what the blocking host entry
    point returns for it.  T = 400 makes the recurrence run as 3 time slabs."""
    monkeypatch.setenv("XB_DECODE_ASYNC", decode_async)
    import torch
    from conftest import encoder_shapes, seeded_state_dict
    from xna_basecaller_amd import _lib

    F, nb, L, N = 128, 5, 2000, 70
    keys, shapes = encoder_shapes(F, nb)
    sd = seeded_state_dict(keys, shapes, seed=9)
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8)
    ctx.load_state_dict(sd)
    alphabet = "NACGTX"
    rng = np.random.default_rng(4)
    batches = [rng.standard_normal((n, L)).astype(np.float32) for n in (70, 33, 70, 1, 64)]
    expect = [ctx.basecall_chunks(x, alphabet) for x in batches]
    dev = torch.device("cuda", 0)
    d_in = [torch.from_numpy(x).to(dev) for x in batches]
    d_seq = [torch.full((x.shape[0], ctx.T), -1, dtype=torch.int8, device=dev) for x in batches]
    d_len = [torch.full((x.shape[0],), -1, dtype=torch.int32, device=dev) for x in batches]
    torch.cuda.synchronize()
    for rep in range(2):
        for x, s, l in zip(d_in, d_seq, d_len):
            ctx.basecall_chunks_dev(x.data_ptr(), x.shape[0], alphabet, s.data_ptr(), l.data_ptr())
        ctx.synchronize()
        for (eseq, elen), s, l in zip(expect, d_seq, d_len):
            assert np.array_equal(l.cpu().numpy(), elen)
            assert np.array_equal(s.cpu().numpy(), eseq)
    ctx.close()


def test_two_calls_in_flight_share_one_pass(monkeypatch):
    """Co-scheduling of two calls in flight (xb_reserve_pairing opts in; contexts of at most 640 chunks at features 768): the first of two
    asynchronous calls is held back and both go through the encoder and the decode as one batch -- half the recurrence launches,
    the same bytes as without; a held call is launched on its own by xb_result_stream / xb_synchronize / any other entry point,
    and a lone last call too.  Without the opt-in (and with XB_FUSE=0 even after it) every call is enqueued on its own."""
    import torch
    from conftest import encoder_shapes, seeded_state_dict
    from xna_basecaller_amd import _lib

    F, nb, L, N = 96, 6, 2000, 130
    keys, shapes = encoder_shapes(F, nb)
    sd = seeded_state_dict(keys, shapes, seed=3)
    alphabet = "NACGTXY"
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    d_in = [torch.from_numpy(rng.standard_normal((N, L)).astype(np.float32)).to(dev) for _ in range(5)]

    def run(fuse, poke, opt_in=True):
        monkeypatch.setenv("XB_FUSE", fuse)
        ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8)
        ctx.load_state_dict(sd)
        assert not ctx.pairing_active()
        if opt_in:
            assert ctx.reserve_pairing() == (fuse == "1")
        d_seq = [torch.full((N, ctx.T), -1, dtype=torch.int8, device=dev) for _ in d_in]
        d_len = [torch.full((N,), -1, dtype=torch.int32, device=dev) for _ in d_in]
        ctx.set_profiling(True)
        ctx.reset_stage_times()
        for k, (x, s, l) in enumerate(zip(d_in, d_seq, d_len)):
            ctx.basecall_chunks_dev(x.data_ptr(), N, alphabet, s.data_ptr(), l.data_ptr())
            if poke and k == 0:
                ctx.result_stream()                      # the held call goes out alone; calls 1+2 and 3+4 pair up
        ctx.synchronize()
        launches = ctx.stage_times()["decode"][1]
        out = [(s.cpu().numpy(), l.cpu().numpy()) for s, l in zip(d_seq, d_len)]
        # a held call is also flushed by an unrelated entry point
        ctx.basecall_chunks_dev(d_in[0].data_ptr(), N, alphabet, d_seq[0].data_ptr(), d_len[0].data_ptr())
        sc = ctx.encode(d_in[1].cpu().numpy())
        assert np.isfinite(sc).all()
        ctx.synchronize()
        assert np.array_equal(d_seq[0].cpu().numpy(), out[0][0])
        ctx.close()
        return out, launches

    ref, n_ref = run("0", False)
    plain, n_plain = run("1", False, opt_in=False)
    got, n_got = run("1", False)
    poked, n_poked = run("1", True)
    assert n_ref == 5 and n_plain == 5 and n_got == 3 and n_poked == 3          # (0,1) (2,3) 4  /  0 (1,2) (3,4)
    for (rs, rl), (gs, gl), (ps, pl), (qs, ql) in zip(ref, got, poked, plain):
        assert np.array_equal(rs, gs) and np.array_equal(rl, gl)
        assert np.array_equal(rs, ps) and np.array_equal(rl, pl)
        assert np.array_equal(rs, qs) and np.array_equal(rl, ql)


def test_held_call_keeps_the_weights_it_was_called_with():
    """a held-back call must run with the weight set (and the profiling state) in force when it was made: loading new
    weights first launches it."""
    import torch
    from conftest import encoder_shapes, seeded_state_dict
    from xna_basecaller_amd import _lib
    F, nb, L, N = 64, 6, 1500, 20
    keys, shapes = encoder_shapes(F, nb)
    sd_a, sd_b = seeded_state_dict(keys, shapes, seed=1), seeded_state_dict(keys, shapes, seed=2)
    alphabet = "NACGTXY"
    x = np.random.default_rng(0).standard_normal((N, L)).astype(np.float32)
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_MIXED)
    ctx.load_state_dict(sd_a)
    want_a = ctx.basecall_chunks(x, alphabet)
    assert ctx.reserve_pairing()
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(x).to(dev)
    d_seq = torch.full((N, ctx.T), -1, dtype=torch.int8, device=dev)
    d_len = torch.full((N,), -1, dtype=torch.int32, device=dev)
    ctx.basecall_chunks_dev(d_x.data_ptr(), N, alphabet, d_seq.data_ptr(), d_len.data_ptr())      # held back
    ctx.load_state_dict(sd_b)                                                                     # launches it with weights A first
    ctx.synchronize()
    assert np.array_equal(d_seq.cpu().numpy(), want_a[0]) and np.array_equal(d_len.cpu().numpy(), want_a[1])
    want_b = ctx.basecall_chunks(x, alphabet)
    assert not np.array_equal(want_a[0], want_b[0])
    # a held call at destruction is launched (its deferred gather would be a collective) and waited for, not dropped
    ctx.basecall_chunks_dev(d_x.data_ptr(), N, alphabet, d_seq.data_ptr(), d_len.data_ptr())
    ctx.close()
    torch.cuda.synchronize()
    assert np.array_equal(d_seq.cpu().numpy(), want_b[0])


def test_submit_collect_pipeline_matches_blocking_calls():
    """xb_submit_chunks / xb_collect_chunks: two batches in flight through the pinned staging slots give what the
    blocking entry point gives, in order; slot misuse is an XB_ERR_STATE, not undefined behaviour."""
    from xna_basecaller_amd import _lib
    F, nb, L, N = 64, 6, 1500, 40
    keys, shapes = encoder_shapes(F, nb)
    sd = seeded_state_dict(keys, shapes, seed=5)
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8)
    ctx.load_state_dict(sd)
    alphabet = "NACGTXY"
    rng = np.random.default_rng(8)
    batches = [rng.standard_normal((n, L)).astype(np.float32) for n in (40, 17, 40, 1, 33, 40)]
    expect = [ctx.basecall_chunks(x, alphabet) for x in batches]
    got, pending, slot = [], None, 0
    for x in batches:
        n = ctx.submit_chunks(slot, x, alphabet)
        x[:] = 0                                  # the caller's buffer is free again as soon as submit returns
        if pending is not None:
            got.append(ctx.collect_chunks(*pending))
        pending, slot = (slot, n), slot ^ 1
    got.append(ctx.collect_chunks(*pending))
    for (es, el), (gs, gl) in zip(expect, got):
        assert np.array_equal(el, gl) and np.array_equal(es, gs)
    # four slots, co-scheduled pairs: submit k+3 before collecting k (the product pipeline's rotation)
    assert ctx.reserve_pairing()
    got, pending, slot = [], [], 0
    ctx.set_profiling(True)
    ctx.reset_stage_times()
    for x in [b.copy() for b in (rng.standard_normal((n, L)).astype(np.float32) for n in (40, 17, 40, 1, 33, 40, 5))]:
        pending.append((slot, ctx.submit_chunks(slot, x, alphabet), x.copy()))
        x[:] = 0
        slot = (slot + 1) % _lib.XB_PIPELINE_SLOTS
        if len(pending) == _lib.XB_PIPELINE_SLOTS:
            s_, n_, x_ = pending.pop(0)
            got.append((ctx.collect_chunks(s_, n_), x_))
    for s_, n_, x_ in pending:
        got.append((ctx.collect_chunks(s_, n_), x_))
    assert ctx.stage_times()["decode"][1] == 4          # 7 batches = 3 pairs + 1
    ctx.set_profiling(False)
    for (gs, gl), x_ in got:
        es, el = ctx.basecall_chunks(x_, alphabet)
        assert np.array_equal(el, gl) and np.array_equal(es, gs)
    with pytest.raises(_lib.XbError) as e:
        ctx.collect_chunks(0, 1)                  # nothing in flight
    assert e.value.code == -4
    with pytest.raises(_lib.XbError) as e:
        ctx.submit_chunks(_lib.XB_PIPELINE_SLOTS, batches[1][:3], alphabet)     # no such slot
    assert e.value.code == -1
    ctx.submit_chunks(1, batches[1][:3] + 1.0, alphabet)
    with pytest.raises(_lib.XbError) as e:
        ctx.submit_chunks(1, batches[1][:3], alphabet)     # slot still in flight
    assert e.value.code == -4
    ctx.collect_chunks(1, 3)
    ctx.close()
