#!/usr/bin/env python
"""GPU tool: two builds of libxnacall.so against each other, BYTE for byte -- scores (fp32), called sequences and lengths -- over
the shapes that exercise the recurrence's variants (one / two groups per workgroup, ragged groups, chunk slabs, small feature
sizes, every precision), each build loaded in its own process (XNA_LIBXNACALL).  Used in round 5 to show that the recurrence's
lean-issue rewrite (same arithmetic, fewer instructions) changes no bit:  python tools/lib_ab.py OLD.so NEW.so"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [  # features, nb, chunk_len, n, precision, paired calls, env
    (768, 6, 2000, 512, "mixed", 2, {}),
    (768, 6, 1500, 512, "mixed", 1, {}),
    (768, 5, 1000, 1024, "mixed", 1, {}),
    (768, 6, 1000, 600, "f16f8", 1, {}),
    (768, 6, 1000, 513, "f16x3", 1, {}),
    (768, 6, 800, 1500, "mixed", 1, {}),
    (768, 5, 800, 200, "f16", 1, {}),
    (768, 6, 1000, 512, "mixed", 2, {"XB_OVERLAP": "0"}),
    (768, 6, 1000, 700, "mixed", 1, {"XB_LSTM_SPREAD": "1"}),
    (256, 6, 1000, 300, "mixed", 1, {}),
    (128, 5, 1000, 130, "f16f8", 1, {}),
    (96, 5, 1000, 70, "f16x3", 1, {}),
    (64, 6, 1000, 40, "mixed", 1, {}),
]


def worker():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from xna_basecaller_amd import _lib
    from xna_basecaller_amd.synthetic import peaky_weights, seeded_weights
    out = []
    for F, nb, L, n, prec, calls, env in CASES:
        os.environ.update(env)
        ctx = _lib.Context(0, nb, 3, F, 19, 5, 5.0, 2.0, L, n, precision=_lib.PRECISIONS[prec])
        for k in env:
            os.environ.pop(k)
        ctx.load_state_dict(peaky_weights(F, nb) if F == 768 else seeded_weights(F, nb))
        if calls > 1:
            ctx.reserve_pairing()
        gen = torch.Generator(device="cuda")
        gen.manual_seed(F + n)
        xs = [torch.randn((n, L), dtype=torch.float32, device="cuda", generator=gen) for _ in range(calls)]
        alphabet = "NACGTXY"[:nb + 1]
        bufs = [(torch.empty((n, ctx.T), dtype=torch.int8, device="cuda"), torch.empty((n,), dtype=torch.int32, device="cuda")) for _ in xs]
        for x, (s, l) in zip(xs, bufs):
            ctx.basecall_chunks_dev(x.data_ptr(), n, alphabet, s.data_ptr(), l.data_ptr())
        ctx.synchronize()
        h = hashlib.sha1()
        for s, l in bufs:
            h.update(s.cpu().numpy().tobytes())
            h.update(l.cpu().numpy().tobytes())
        scores = ctx.encode(xs[0].cpu().numpy())
        h.update(np.ascontiguousarray(scores).tobytes())
        out.append(h.hexdigest())
        ctx.close()
    print("DIGESTS " + json.dumps(out))


def main():
    if len(sys.argv) == 2 and sys.argv[1] == "--worker":
        return worker()
    libs = sys.argv[1:3]
    res = []
    for lib in libs:
        env = dict(os.environ, XNA_LIBXNACALL=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("DIGESTS ")]
        if r.returncode or not line:
            print(r.stderr.decode()[-2000:])
            print("lib_ab: %s failed (rc %d)" % (lib, r.returncode))
            return 1
        res.append(json.loads(line[0][8:]))
    bad = 0
    for case, a, b in zip(CASES, res[0], res[1]):
        same = a == b
        bad += not same
        print("%-60s %s" % (str(case), "identical" if same else "DIFFERENT  %s vs %s" % (a[:10], b[:10])))
    print("lib_ab:", "OK -- every byte identical" if not bad else "%d cases differ" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
