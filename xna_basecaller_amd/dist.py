"""
Multi-GPU layer: one process per GPU, reads sharded round-robin, no data-path collective, and ONE
gather of the called sequences to rank 0 (RCCL over xGMI via torch.distributed's "nccl" backend on
ROCm; "gloo" on CPU for tests).  The reference has no distributed code at all;
reads are independent units, chunks of one read stay on one rank so that
stitching is local.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["init_from_env", "env_rank_world", "exchange_comm_id", "RcclGather", "rank", "world_size", "shard", "gather_called", "gather_packed", "barrier",
           "DeferredGather"]


def env_rank_world():
    """(rank, world) from torchrun's environment WITHOUT touching torch.distributed or the GPU -- for work that has to
    happen before the device is initialised (forking the reader pool)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return (int(os.environ.get("RANK", "0")), world) if world > 1 else (0, 1)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's env)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def barrier():
    if world_size() > 1:
        dist.barrier()


def shard(items, rank_=None, world=None):
    """Round-robin partition of an iterable of units (reads): unit i goes to rank i % world; yields (i, item)."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    for i, item in enumerate(items):
        if i % w == r:
            yield i, item


def _comm_device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_called(records, dst=0):
    """
    records: list of (global_index, read_id, sequence, qstring) called on this rank.
    Returns on rank `dst` the records of all ranks ordered by global_index (None elsewhere).
    Wire format: all_gather of int64 byte counts, then all_gather of uint8 payloads padded to the
    maximum count (what NCCL/RCCL offers without variable-size collectives).
    """
    w = world_size()
    if w == 1:
        return sorted(records, key=lambda r: r[0])
    lines = ["%d\t%s\t%s\t%s" % (i, rid, seq, q) for i, rid, seq, q in records]
    payload = np.frombuffer("\n".join(lines).encode("ascii"), dtype=np.uint8)
    dev = _comm_device()
    count = torch.tensor([payload.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(w)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    buf = torch.zeros(width, dtype=torch.uint8, device=dev)
    if payload.size:
        buf[:payload.size] = torch.from_numpy(payload.copy()).to(dev)
    out = torch.empty(w * width, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, buf)
    if rank() != dst:
        return None
    out = out.cpu().numpy().reshape(w, width)
    merged = []
    for r in range(w):
        if counts[r] == 0:
            continue
        for line in out[r, :counts[r]].tobytes().decode("ascii").split("\n"):
            i, rid, seq, q = line.split("\t")
            merged.append((int(i), rid, seq, q))
    merged.sort(key=lambda r: r[0])
    return merged


def gather_packed(seq, lens):
    """
    Fixed-shape gather for the synthetic bench: seq (n, T) int8 left-packed + lens (n,) int32 device
    tensors of every rank -> (world, n, T), (world, n) on every rank (all_gather; payload ~T bytes per chunk).
    """
    w = world_size()
    if w == 1:
        return seq[None].clone(), lens[None].clone()       # a copy, like the collective: the inputs may be reused
    # all_gather_into_tensor concatenates along dim 0 (the layout both gloo and nccl accept)
    out_s = torch.empty((w * seq.shape[0],) + tuple(seq.shape[1:]), dtype=seq.dtype, device=seq.device)
    out_l = torch.empty((w * lens.shape[0],) + tuple(lens.shape[1:]), dtype=lens.dtype, device=lens.device)
    dist.all_gather_into_tensor(out_s, seq.contiguous())
    dist.all_gather_into_tensor(out_l, lens.contiguous())
    return out_s.view((w,) + tuple(seq.shape)), out_l.view((w,) + tuple(lens.shape))


class DeferredGather:
    """
    The path's one exchange, taken off the compute stream: batch k's (seq, lens) are gathered while batch k+1 is being
    computed.  `submit` hands in the buffers of the batch just enqueued together with an event that fires when they are
    complete, starts the gather of the PREVIOUS submission on a side stream behind that submission's event, and returns
    the gathered result of the previous submission (None for the first call); `flush` gathers the last one.  After
    `submit(k)` returned, `consumed(k - 1)` is the event to wait for before batch k - 1's buffers are written again
    (two buffer sets in rotation: wait for it just before enqueueing batch k + 1).
    CPU tensors (gloo tests) take the same route without streams or events.
    """

    def __init__(self):
        self._pending = None            # (seq, lens, ready_event, ticket)
        self._consumed = {}
        self._ticket = 0
        self._side = None

    def _gather(self, seq, lens, ready):
        if seq.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=seq.device)
            if ready is not None:
                self._side.wait_event(ready)
            with torch.cuda.stream(self._side):
                out = gather_packed(seq, lens)
                done = torch.cuda.Event()
                done.record(self._side)
            return out, done
        return gather_packed(seq, lens), None

    def submit(self, seq, lens, ready_event=None):
        prev, out = self._pending, None
        self._pending = (seq, lens, ready_event, self._ticket)
        self._ticket += 1
        if prev is not None:
            out, done = self._gather(prev[0], prev[1], prev[2])
            self._consumed[prev[3]] = done
            self._consumed.pop(prev[3] - 2, None)
        return out

    def consumed(self, ticket):
        """Event recorded behind the gather of submission `ticket` (None on CPU or if it has not been gathered)."""
        return self._consumed.get(ticket)

    def flush(self):
        prev, self._pending = self._pending, None
        if prev is None:
            return None
        out, done = self._gather(prev[0], prev[1], prev[2])
        if done is not None:
            done.synchronize()
        return out


def exchange_comm_id(rank_, world, make_id, port_offset=1, timeout=120.0):
    """
    Out-of-band hand-off of the 128-byte communicator id (xb_comm_unique_id) from rank 0 to the other ranks.  With
    torch.distributed up (torchrun) its store carries it; otherwise a one-shot TCP rendezvous on MASTER_ADDR :
    MASTER_PORT + port_offset (rank 0 listens and serves world - 1 connections; the others retry until it is up).
    """
    if world <= 1:
        return make_id()
    if dist.is_available() and dist.is_initialized():
        box = [make_id() if rank_ == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    import socket
    import time
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) + int(port_offset)
    deadline = time.monotonic() + timeout
    if rank_ == 0:
        cid = make_id()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _peer = srv.accept()
                with conn:
                    conn.sendall(cid)
        return cid
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as c:
                buf = b""
                while len(buf) < 128:
                    part = c.recv(128 - len(buf))
                    if not part:
                        raise ConnectionError("short communicator id")
                    buf += part
                return buf
        except (ConnectionError, OSError):
            if time.monotonic() > deadline:
                raise
            time.sleep(0.2)


class RcclGather:
    """
    The gather of called sequences through the C ABI (xb_comm / xb_gather_called: RCCL all-gathers on the communicator's own
    stream behind the context's result stream) -- DeferredGather's job without torch.distributed on the data path.
    `submit` enqueues the gather of the batch just handed in (it starts on the device when that batch is complete, while the
    next batch computes) and returns the PREVIOUS batch's gathered (seq (world, n, T), lens (world, n)) device tensors;
    `before_batch` orders the producer behind the gather that last read the buffer set it is about to overwrite (two sets
    in rotation); `flush` completes the last one.
    """

    def __init__(self, ctx, device_index, rank_, world):
        from . import _lib
        self.ctx, self.world = ctx, int(world)
        cid = exchange_comm_id(rank_, world, _lib.Comm.unique_id)
        self.comm = _lib.Comm(device_index, rank_, world, cid)
        self._out = {}
        self._prev = None
        self._k = 0

    def before_batch(self):
        self.comm.fence(self.ctx, lag=1)

    def submit(self, seq, lens):
        n, T = int(seq.shape[0]), int(seq.shape[1])
        slot = self._k & 1
        key = (slot, n, T)
        if key not in self._out:
            self._out[key] = (torch.empty((self.world, n, T), dtype=torch.int8, device=seq.device),
                              torch.empty((self.world, n), dtype=torch.int32, device=seq.device))
        all_seq, all_len = self._out[key]
        self.comm.gather_called(self.ctx, seq.data_ptr(), lens.data_ptr(), n, T, all_seq.data_ptr(), all_len.data_ptr())
        prev, self._prev = self._prev, (all_seq, all_len)
        self._k += 1
        return prev

    def flush(self):
        self.ctx.result_stream()            # a basecall held back for co-scheduling is launched now, its gather behind it
        self.comm.synchronize()
        prev, self._prev = self._prev, None
        return prev

    def close(self):
        self.comm.close()
