#!/usr/bin/env python
"""
Fixture generator for the eval-loop hand-off (eval_model.sh:119-177): basecalls FASTQ -> minimap2 PAF ->
src/tools/analyze_paf.py, which joins the PAF's query names with the FASTQ's record ids (SeqIO.index(reads_filepath)
[read_id], analyze_paf.py:424-425, 591-593; misc/data_io.py:225-246) and reads the PAF with misc/data_io.py:read_paf.

Run in the BUILD container only (it imports the reference's src/misc/data_io.py by file path; h5py and Bio -- absent third
party modules that read_paf itself never touches -- are stubbed in sys.modules first).  What it stores in
tests/golden/evalloop.json is DATA: the FASTQ text this package's Writer produced for six synthetic reads, the PAF text a
`minimap2 -x map-ont -c --cs=short --secondary=no` run would hand over for them (query name = FASTQ header up to the first
whitespace, query length = sequence length; one read unaligned, one on the minus strand), and what the REFERENCE'S read_paf
extracts from that PAF (columns read_id ... mapping_quality plus the cs tag).  tests/test_host.py then pins: the writer
still produces those bytes, every PAF read_id the reference extracted is a FASTQ record id, lengths agree, tags survive.
minimap2 itself is not in any image; nothing of the reference travels.
"""
import importlib.util
import io
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/src"


def synthetic_fastq():
    """(FASTQ text the product's Writer emits for six synthetic reads, [(read_id, sequence)]) -- also what the test re-runs."""
    import numpy as np
    from xna_basecaller_amd import io as xio
    from xna_basecaller_amd import reads as xreads
    rng = np.random.default_rng(11)
    recs = []
    for i in range(6):
        raw = rng.integers(300, 700, 4000 + 500 * i).astype(np.int16)
        recs.append((raw, dict(read_id="%08x-aaaa-4bbb-8ccc-%012d" % (0xABC000 + i, i), run_id="run%02d" % (i % 2), range=1400.0,
                               digitisation=8192.0, offset=10, sampling_rate=4000.0, channel_number=str(100 + i), start_mux=1 + i % 4,
                               read_number=7 * i, start_time=1000 * i, exp_start_time="2021-03-04T05:06:07Z")))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        xreads.write_bundle(os.path.join(d, "evalloop.xsig.npz"), recs)
        rds = list(xreads.get_reads(d))
    seqs = ["".join(rng.choice(list("ACGTXY"), 30 + 7 * i)) for i in range(6)]
    seqs[3] = ""                                              # an empty call: the writer skips it (io.py:444-445)
    out = io.StringIO()
    results = [(r, {"sequence": s, "qstring": "O" * len(s), "sig_move": None}) for r, s in zip(rds, seqs)]
    summ = "/tmp/evalloop_%d_summary.tsv" % os.getpid()
    w = xio.Writer("wfq", iter(results), fd=out, group_key="xna_r9.4.1_e8_sup@v3.3", summary=summ)
    w.run()
    os.remove(summ)
    return out.getvalue(), [(r.read_id, s) for r, s in zip(rds, seqs)]


def paf_text(records):
    """What minimap2 emits per aligned query: 12 mandatory columns + tags (tp, cm, s1, dv, rl, cg, cs)."""
    lines = []
    for k, (rid, seq) in enumerate(records):
        if not seq or k == 4:                                 # the empty call never reaches minimap2; read 4 stays unaligned
            continue
        n = len(seq)
        strand = "-" if k == 1 else "+"
        lines.append("\t".join(str(v) for v in [
            rid, n, 2, n - 1, strand, "POC_target_%d" % (k % 3), 120, 5, 5 + n - 3, n - 5, n - 3, 60,
            "tp:A:P", "cm:i:%d" % (n // 4), "s1:i:%d" % (n - 6), "dv:f:0.0123", "rl:i:0", "cg:Z:%dM" % (n - 3),
            "cs:Z::%d*ag:%d" % (10, n - 14)]))
    return "\n".join(lines) + "\n"


def reference_read_paf(paf_path):
    for name in ("h5py", "Bio", "Bio.SeqIO", "Bio.Seq", "Bio.SeqRecord"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
    sys.modules["Bio.Seq"].Seq = object
    sys.modules["Bio.SeqRecord"].SeqRecord = object
    spec = importlib.util.spec_from_file_location("ref_data_io", os.path.join(REF, "misc", "data_io.py"))
    mod = importlib.util.module_from_spec(spec)
    # the module probes a list of the authors' project directories at import time and raises when none exists
    # (data_io.py:682-693; BASE_DIR only locates a k-mer pore model, unused here): let the first probe succeed
    real_exists = os.path.exists
    os.path.exists = lambda q: True if ("GIS" in str(q) or "xna_basecallers" in str(q)) else real_exists(q)
    try:
        spec.loader.exec_module(mod)
    finally:
        os.path.exists = real_exists
    df = mod.read_paf(paf_path, extra_tags=["cs"])
    return {"columns": list(df.columns), "rows": json.loads(df.to_json(orient="values"))}


def main():
    fastq, records = synthetic_fastq()
    paf = paf_text(records)
    path = "/tmp/evalloop_%d.paf" % os.getpid()
    with open(path, "w") as fh:
        fh.write(paf)
    ref = reference_read_paf(path)
    os.remove(path)
    out = {"note": "inputs (fastq: written by xna_basecaller_amd.io.Writer; paf: minimap2-shaped) and the reference read_paf's view",
           "records": records, "fastq": fastq, "paf": paf, "reference_read_paf": ref}
    with open(os.path.join(HERE, "evalloop.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote evalloop.json:", len(records), "reads,", len(ref["rows"]), "PAF rows; columns", ref["columns"])


if __name__ == "__main__":
    main()
