"""GPU: the gather of called sequences through the C ABI (xb_comm / xb_gather_called over librccl), one-rank self-test.
A one-GPU box cannot host two RCCL ranks (one rank per device), so what runs here is everything but the wire: library
loading, id, communicator, the event hand-off from the context's result stream, the two all-gathers on the communicator's
stream, fence and synchronise -- with world = 1 the gathered rows must equal the batch's own, for consecutive batches in
rotating buffers.  The world > 1 rendezvous is covered on the CPU tier (tests/test_dist.py)."""
import numpy as np
import pytest

from conftest import encoder_shapes, seeded_state_dict
from xna_basecaller_amd import _lib
from xna_basecaller_amd import dist as xdist

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pairing", [False, True])
def test_gather_called_one_rank_roundtrip(pairing):
    """pairing: the asynchronous calls are co-scheduled two by two (xb_reserve_pairing, bench.py's schedule), so every other
    gather is asked for while its basecall is still held back and runs right behind that call's launch."""
    import torch
    F, nb, L, N = 64, 6, 1000, 40
    keys, shapes = encoder_shapes(F, nb)
    sd = seeded_state_dict(keys, shapes, seed=3)
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8)
    ctx.load_state_dict(sd)
    if pairing:
        assert ctx.reserve_pairing()
    T = ctx.T
    g = xdist.RcclGather(ctx, 0, 0, 1)
    assert g.comm.rank == 0 and g.comm.world == 1
    xs = [torch.randn((N, L), dtype=torch.float32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(s))
          for s in range(4)]
    d_seq = [torch.empty((N, T), dtype=torch.int8, device="cuda") for _ in range(2)]
    d_len = [torch.empty((N,), dtype=torch.int32, device="cuda") for _ in range(2)]
    got = []
    for k, x in enumerate(xs):
        g.before_batch()
        ctx.basecall_chunks_dev(x.data_ptr(), N, "NACGTXY", d_seq[k & 1].data_ptr(), d_len[k & 1].data_ptr())
        prev = g.submit(d_seq[k & 1], d_len[k & 1])
        if prev is not None:
            g.comm.synchronize()                     # test only: read the previous batch's gathered rows on the host now
            got.append((prev[0].cpu().numpy().copy(), prev[1].cpu().numpy().copy()))
    last = g.flush()
    got.append((last[0].cpu().numpy().copy(), last[1].cpu().numpy().copy()))
    ctx.synchronize()
    for k, x in enumerate(xs):
        seq, lens = ctx.basecall_chunks(x.cpu().numpy(), "NACGTXY")
        assert got[k][0].shape == (1, N, T) and got[k][1].shape == (1, N)
        assert np.array_equal(got[k][1][0], lens) and np.array_equal(got[k][0][0], seq)
    g.close()
    ctx.close()


def test_comm_error_paths():
    with pytest.raises(ValueError):
        _lib.Comm(0, 0, 1, b"short")
    with pytest.raises(_lib.XbError):
        _lib.Comm(0, 2, 2, _lib.Comm.unique_id())      # rank outside the world


def test_destroy_the_communicator_before_the_context_with_a_deferred_gather_pending():
    """a gather asked for while its basecall is held back for pairing is enqueued behind that call's launch.  A C
    caller may destroy the communicator FIRST: xb_comm_destroy launches the held call itself and completes the gather, so the
    context that is destroyed afterwards holds no pointer into freed memory (it used to: xb_ctx_destroy launched the held call
    and its callback entered RCCL on a destroyed communicator).  The gathered rows written on the way are the right ones."""
    import torch
    F, nb, L, N = 64, 6, 1000, 24
    keys, shapes = encoder_shapes(F, nb)
    sd = seeded_state_dict(keys, shapes, seed=4)
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8)
    ctx.load_state_dict(sd)
    assert ctx.reserve_pairing()
    T = ctx.T
    comm = _lib.Comm(0, 0, 1, _lib.Comm.unique_id())
    x = torch.randn((N, L), dtype=torch.float32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    d_seq = torch.zeros((N, T), dtype=torch.int8, device="cuda")
    d_len = torch.zeros((N,), dtype=torch.int32, device="cuda")
    all_seq = torch.zeros((1, N, T), dtype=torch.int8, device="cuda")
    all_len = torch.full((1, N), -1, dtype=torch.int32, device="cuda")
    ctx.basecall_chunks_dev(x.data_ptr(), N, "NACGTXY", d_seq.data_ptr(), d_len.data_ptr())      # held back: nothing enqueued
    comm.gather_called(ctx, d_seq.data_ptr(), d_len.data_ptr(), N, T, all_seq.data_ptr(), all_len.data_ptr())   # deferred
    comm.close()                                   # launches the held call, runs its gather, waits, destroys
    ctx.synchronize()
    seq, lens = ctx.basecall_chunks(x.cpu().numpy(), "NACGTXY")
    assert np.array_equal(all_len.cpu().numpy()[0], lens) and np.array_equal(all_seq.cpu().numpy()[0], seq)
    ctx.close()                                    # nothing held, no callback left
    # the other order: context first (its destroy launches the held call with the communicator alive), then the communicator
    ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, N, precision=_lib.XB_PREC_F16F8)
    ctx.load_state_dict(sd)
    assert ctx.reserve_pairing()
    comm = _lib.Comm(0, 0, 1, _lib.Comm.unique_id())
    all_len.fill_(-1)
    ctx.basecall_chunks_dev(x.data_ptr(), N, "NACGTXY", d_seq.data_ptr(), d_len.data_ptr())
    comm.gather_called(ctx, d_seq.data_ptr(), d_len.data_ptr(), N, T, all_seq.data_ptr(), all_len.data_ptr())
    ctx.close()
    comm.synchronize()
    assert np.array_equal(all_len.cpu().numpy()[0], lens)
    comm.close()
