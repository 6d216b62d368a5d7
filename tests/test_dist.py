"""CPU: the N>1 path (round-robin sharding + gather of called sequences) with world_size 2 over gloo."""
import os
import socket
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["XB_ROOT"])
import numpy as np, torch
from xna_basecaller_amd import dist as xd
rank, world = xd.init_from_env(backend="gloo")
assert world == 2
units = ["read%03d" % i for i in range(11)]
mine = list(xd.shard(units))
assert [i for i, _ in mine] == list(range(rank, 11, 2))
recs = [(i, u, "ACGTXY"[: 1 + i % 6] * (1 + i % 3), "O" * ((1 + i % 6) * (1 + i % 3))) for i, u in mine]
if rank == 1:
    recs = recs[::-1]           # arrival order on a rank must not matter
merged = xd.gather_called(recs, dst=0)
if rank == 0:
    assert [m[0] for m in merged] == list(range(11))
    assert [m[1] for m in merged] == units
    for i, rid, seq, q in merged:
        assert seq == "ACGTXY"[: 1 + i % 6] * (1 + i % 3) and len(q) == len(seq)
else:
    assert merged is None
# empty shard on one rank
merged = xd.gather_called([(0, "only", "AC", "OO")] if rank == 0 else [], dst=0)
if rank == 0:
    assert merged == [(0, "only", "AC", "OO")]
# fixed-shape gather used by bench.py
seq = torch.full((3, 8), rank + 65, dtype=torch.int8)
lens = torch.tensor([rank, 2, 3], dtype=torch.int32)
s, l = xd.gather_packed(seq, lens)
assert s.shape == (2, 3, 8) and l.shape == (2, 3)
assert s[0, 0, 0].item() == 65 and s[1, 2, 7].item() == 66 and l[:, 0].tolist() == [0, 1]
# deferred (one batch late) gather used by bench.py and the CLI: results come back in submission order, one call late
dg = xd.DeferredGather()
outs = []
for k in range(5):
    seq = torch.full((2, 4), 10 * k + rank, dtype=torch.int8)
    lens = torch.tensor([k, rank], dtype=torch.int32)
    r = dg.submit(seq, lens)
    assert (r is None) == (k == 0)
    if r is not None:
        outs.append(r)
outs.append(dg.flush())
assert dg.flush() is None
assert len(outs) == 5
for k, (s, l) in enumerate(outs):
    assert s.shape == (2, 2, 4) and l.shape == (2, 2)
    assert s[0, 0, 0].item() == 10 * k and s[1, 1, 3].item() == 10 * k + 1
    assert l[0].tolist() == [k, 0] and l[1].tolist() == [k, 1]
# the CLI's windowed gather of called reads to rank 0 (cli/basecaller.py:_gathered_results)
from xna_basecaller_amd.cli.basecaller import _gathered_results, READ_FIELDS
class FakeRead:
    def __init__(self, i):
        self.index = i
        for k in READ_FIELDS: setattr(self, k, "%s%d" % (k[:2], i))
        self.signal = np.zeros(100 + i, np.float32)
    def tagdata(self): return ["mx:i:%d" % self.index]
class FakeLoader: total = 23
def local(fail_at=None):
    for i in range(rank, 23, 2):
        if fail_at is not None and i == fail_at: raise ValueError("boom at %d" % i)
        yield FakeRead(i), {"sequence": "ACGT"[: 1 + i % 4], "qstring": "O" * (1 + i % 4)}
got = list(_gathered_results(local(), FakeLoader, rank, world, window=3))
if rank == 0:
    assert [r.read_id for r, _ in got] == ["re%d" % i for i in range(23)]
    assert [len(r.signal) for r, _ in got] == [100 + i for i in range(23)] and got[5][0].tagdata() == ["mx:i:5"]
    assert all(res["sequence"] == "ACGT"[: 1 + i % 4] for i, (_, res) in enumerate(got))
else:
    assert got == []
# a rank that fails mid-way: nobody hangs, rank 0 and the failing rank both raise
try:
    list(_gathered_results(local(fail_at=9), FakeLoader, rank, world, window=3))
    raised = False
except (RuntimeError, ValueError) as e:
    raised = True
    assert "boom at 9" in str(e)
assert raised == (rank in (0, 1))
xd.barrier()
print("rank", rank, "ok")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_and_gather_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(_free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, XB_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    try:
        outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    finally:
        for p in procs:                      # never leave a blocked rank behind
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_single_process_paths():
    from xna_basecaller_amd import dist as xd
    assert xd.rank() == 0 and xd.world_size() == 1
    assert list(xd.shard("abc")) == [(0, "a"), (1, "b"), (2, "c")]
    assert xd.gather_called([(2, "b", "A", "O"), (0, "a", "C", "O")]) == [(0, "a", "C", "O"), (2, "b", "A", "O")]


def _id_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from xna_basecaller_amd import dist as xdist
    cid = xdist.exchange_comm_id(rank, world, lambda: bytes(range(128)))
    q.put((rank, cid))


def test_comm_id_rendezvous_without_torch_distributed():
    """The 128-byte communicator id of xb_comm_create travels from rank 0 to the other ranks over a one-shot TCP rendezvous
    on MASTER_ADDR : MASTER_PORT + 1 when no torch.distributed process group exists (a pure C-ABI launcher's route)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_id_worker, args=(r, 3, port, q)) for r in (2, 1, 0)]          # rank 0 comes up LAST
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    assert got == {0: bytes(range(128)), 1: bytes(range(128)), 2: bytes(range(128))}
