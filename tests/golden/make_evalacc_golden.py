#!/usr/bin/env python
"""
Fixture generator for the ACCURACY half of the eval loop (eval_model.sh:155-177): FASTQ + minimap2 PAF ->
src/tools/analyze_paf.py -> the UB / DNA error table (README.md:139-143).

Run in the BUILD container only.  It imports, BY FILE PATH and as reference code, src/misc/data_io.py (read_paf),
src/misc/utils.py (compute_all_error_rates_paf and what it calls: parse_cs_flag, compute_read_matches, polish_target_matches,
compute_errors_paf, compute_error_rate_per_pos_paf) and src/tools/analyze_paf.py (compute_stats_error_rate).  Third-party
modules that those functions never touch (Bio, Levenshtein, h5py) are placeholders in sys.modules; data_io.py probes the
authors' project directories at import time (data_io.py:682-693), xna_refs.py wants ./xna_libs -- the probe is answered and the
import runs with the reference tree as working directory; nothing is written there.

Inputs (all DATA, stored in tests/golden/evalacc.json):
  * three POC templates with their UB position `N` -- read from the reference's data file xna_libs/POC/refdb_short.fasta;
  * reads derived from those templates by explicit edit scripts (matches, substitutions, insertions, deletions, the UB called
    as X / Y, as a natural base, deleted, or shifted next to a deletion -- the cases utils.py:661-725 polishes), on both
    strands, with soft-clipped flanks: so the alignment is KNOWN and the PAF row (coordinates, counts, `cs:Z:` in minimap2's
    short form with `*nn` where an UB letter meets the template's N) is what `minimap2 -x map-ont -c --cs=short` reports for
    it -- minimap2 itself is in no image;
  * the FASTQ text this package's Writer produces for those reads (the product's own output format).
Outputs stored: the per-read columns compute_all_error_rates_paf appends (read_acc, target_acc, ub_acc, ub_area_acc,
ub_area_acc_plus, non_ub_area_acc, target_alig_acc, fdr, fpr, true/false pos/neg, ub_area_seq), the per-position error rates
per (target, strand), and compute_stats_error_rate's cuts (only_ub, no_ub, inside / outside the UB area, by distance) -- the
numbers behind "UB accuracy" = 100 - mean(only_ub) and "DNA accuracy" = 100 - mean(no_ub).
tests/test_host.py::test_eval_loop_accuracy_table_from_this_packages_fastq re-creates the FASTQ with the Writer, restates
the metrics (tests/evalloop_metrics.py) and must reproduce every stored number.
"""
import importlib.util
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REFTREE = "/root/reference"

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "X": "Y", "Y": "X", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def build_read(target, ts, te, edits, strand, pre, post):
    """Apply `edits` {target position: op} to target[ts:te] (forward orientation); ops: ('S', base) substitution, ('D',)
    deletion, ('I', bases) insertion BEFORE the position, ('U', letter) the call at an UB position (the template's N).
    Returns (read sequence as it stands in the FASTQ, PAF fields, cs string)."""
    seg, cs, run = [], [], 0
    n_match = n_mis = n_ins = n_del = 0

    def flush():
        nonlocal run
        if run:
            cs.append(":%d" % run)
            run = 0
    pos = ts
    while pos < te:
        ops = edits.get(pos, [])
        ops = ops if isinstance(ops, list) else [ops]
        consumed = False
        for op in ops:
            if op[0] == "I":
                flush()
                seg.append(op[1])
                cs.append("+" + op[1].lower().replace("x", "n").replace("y", "n"))
                n_ins += len(op[1])
            elif op[0] == "D":
                flush()
                cs.append("-" + target[pos].lower())
                n_del += 1
                consumed = True
            elif op[0] in ("S", "U"):
                flush()
                seg.append(op[1])
                cs.append("*%s%s" % (target[pos].lower(), op[1].lower().replace("x", "n").replace("y", "n")))
                n_mis += 1
                consumed = True
        if not consumed:
            assert target[pos] != "N", "an UB position needs an explicit call"
            seg.append(target[pos])
            run += 1
            n_match += 1
        pos += 1
    flush()
    # merge adjacent deletions / insertions the way minimap2 prints them (-acg, +tt)
    merged = []
    for c in cs:
        if merged and c[0] in "+-" and merged[-1][0] == c[0]:
            merged[-1] += c[1:]
        else:
            merged.append(c)
    fwd = "".join(seg)
    if strand == "+":
        read = pre + fwd + post
    else:
        read = pre + revcomp(fwd) + post
    paf = dict(read_length=len(read), read_start=len(pre), read_end=len(pre) + len(fwd), strand=strand,
               target_length=len(target), target_start=ts, target_end=te, n_matches=n_match,
               block_length=n_match + n_mis + n_ins + n_del, mapping_quality=60)
    return read, paf, "".join(merged)


def load_reference():
    for name in ("h5py", "Levenshtein", "Bio", "Bio.SeqIO", "Bio.Seq", "Bio.SeqRecord", "Bio.Align"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
    sys.modules["Bio"].Align = sys.modules["Bio.Align"]
    sys.modules["Bio.Seq"].Seq = object
    sys.modules["Bio.SeqRecord"].SeqRecord = object
    sys.path.insert(0, os.path.join(REFTREE, "src"))
    real_exists, cwd = os.path.exists, os.getcwd()
    os.path.exists = lambda q: True if ("GIS" in str(q) or "xna_basecallers" in str(q)) else real_exists(q)
    os.chdir(REFTREE)
    try:
        import misc.data_io as data_io
        import misc.utils as utils
        spec = importlib.util.spec_from_file_location("ref_analyze_paf", os.path.join(REFTREE, "src", "tools", "analyze_paf.py"))
        analyze = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(analyze)
    finally:
        os.path.exists = real_exists
        os.chdir(cwd)
    return data_io, utils, analyze


def poc_templates(names):
    out, cur = {}, None
    for line in open(os.path.join(REFTREE, "xna_libs", "POC", "refdb_short.fasta")):
        line = line.strip()
        if line.startswith(">"):
            cur = line[1:].split()[0]
        elif cur in names:
            out[cur] = out.get(cur, "") + line
    return out


def design_reads(templates):
    """[(read_id, target_id, sequence, paf fields, cs)]: 11 reads per template, both strands."""
    rng = np.random.default_rng(7)
    reads = []
    for ti, (tid, target) in enumerate(sorted(templates.items())):
        ub = target.index("N")
        L = len(target)
        cases = [
            ("+", 0, L, {ub: ("U", "X")}),                                                     # perfect forward read
            ("-", 0, L, {ub: ("U", "X")}),                                                     # perfect reverse read (the FASTQ holds Y)
            ("+", 3, L - 2, {ub: ("U", "A"), 20: ("S", "G" if target[20] != "G" else "T")}),   # UB called as a natural base
            ("-", 5, L - 7, {ub: ("D",), 40: ("I", "TT")}),                                    # UB deleted, an insertion elsewhere
            ("+", 0, L - 10, {ub: ("U", "Y"), ub - 3: ("D",), ub + 4: ("S", "C" if target[ub + 4] != "C" else "A")}),   # the wrong UB
            ("+", 8, L, {ub - 1: ("D",), ub: ("U", "X"), 30: ("I", "X")}),                     # deletion next to a correct UB, a false UB
            ("-", 0, L, {ub: ("D",), ub + 1: ("S", "X"), 60: ("D",), 61: ("D",)}),             # UB aligned right of its gap: polished
            ("+", 2, L - 1, {ub - 1: ("S", "X"), ub: ("D",), 15: ("S", "A" if target[15] != "A" else "C")}),   # ... left of its gap
            ("+", 0, L, {ub - 1: ("D",), ub: ("S", "T"), ub + 1: ("S", "X")}),                 # GTGG- T Y: the UB one to the right
            ("-", 1, L - 3, {ub - 1: ("S", "X"), ub: ("S", "C"), ub + 1: ("D",)}),             # GTGGY T -: the UB one to the left
            ("+", 0, L, {ub: [("I", "X"), ("D",)]}),                                           # UB inserted beside its gap: NOT polished
        ]
        for ci, (strand, ts, te, edits) in enumerate(cases):
            pre = "".join(rng.choice(list("ACGT"), int(rng.integers(0, 6))))
            post = "".join(rng.choice(list("ACGT"), int(rng.integers(0, 6))))
            seq, paf, cs = build_read(target, ts, te, edits, strand, pre, post)
            reads.append(("%08x-bbbb-4ccc-8ddd-%012d" % (0xE0A000 + ti, ci), tid, seq, paf, cs))
    return reads


def product_fastq(reads):
    """The FASTQ this package's Writer emits for the designed calls (also what the test re-creates)."""
    from xna_basecaller_amd import io as xio
    from xna_basecaller_amd.reads import SyntheticRead
    out = io.StringIO()
    results = []
    for i, (rid, _, seq, _, _) in enumerate(reads):
        r = SyntheticRead(rid, np.zeros(10 * len(seq), np.float32), run_id="evalrun", filename="poc.xsig.npz", channel=str(1 + i),
                          mux=1 + i % 4, read_number=i)
        results.append((r, {"sequence": seq, "qstring": "O" * len(seq), "mean_qscore": 40.0}))
    summ = "/tmp/evalacc_%d_summary.tsv" % os.getpid()
    w = xio.Writer("wfq", iter(results), fd=out, group_key="xna_r9.4.1_e8_sup@v3.3", summary=summ)
    w.run()
    os.remove(summ)
    return out.getvalue()


def paf_text(reads):
    lines = []
    for rid, tid, seq, p, cs in reads:
        lines.append("\t".join(str(v) for v in [
            rid, p["read_length"], p["read_start"], p["read_end"], p["strand"], tid, p["target_length"], p["target_start"],
            p["target_end"], p["n_matches"], p["block_length"], p["mapping_quality"], "tp:A:P", "cm:i:%d" % (p["n_matches"] // 5),
            "s1:i:%d" % p["n_matches"], "dv:f:0.0100", "rl:i:0", "cg:Z:%dM" % (p["target_end"] - p["target_start"]), "cs:Z:" + cs]))
    return "\n".join(lines) + "\n"


def jsonable(v):
    if isinstance(v, (np.floating, float)):
        return None if np.isnan(v) else float(v)
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (list, tuple, np.ndarray)):
        return [jsonable(x) for x in v]
    return v


def main():
    data_io, utils, analyze = load_reference()
    templates = poc_templates(("XNA01", "XNA02", "XNA03"))
    reads = design_reads(templates)
    fastq = product_fastq(reads)
    paf = paf_text(reads)
    path = "/tmp/evalacc_%d.paf" % os.getpid()
    with open(path, "w") as fh:
        fh.write(paf)
    paf_df = data_io.read_paf(path, extra_tags=["cs"])
    os.remove(path)
    # what SeqIO.index(fastq)[read_id].seq gives analyze_paf.py:424-425: the record's sequence line
    recs = fastq.strip().split("\n")
    reads_dict = {h[1:].split()[0]: types.SimpleNamespace(seq=s) for h, s in zip(recs[0::4], recs[1::4])}
    error_rates = utils.compute_all_error_rates_paf(paf_df, templates, reads_dict=reads_dict)
    cols = ["read_id", "target_id", "strand", "read_acc", "target_acc", "ub_area_acc", "ub_area_matches", "ub_area_len",
            "ub_area_seq", "ub_acc", "ub_matches", "ub_len", "ub_area_acc_plus", "non_ub_area_acc", "non_ub_area_matches",
            "non_ub_area_len", "target_alig_acc", "fdr", "fpr", "true_pos", "false_neg", "true_neg", "false_pos"]
    per_read = [{c: jsonable(row[c]) for c in cols} for _, row in paf_df.iterrows()]
    per_pos, cuts = {}, {}
    for (tid, strand), rates in error_rates.items():
        key = "%s/%s" % (tid, strand)
        per_pos[key] = jsonable(np.asarray(rates, dtype=float))
        target = templates[tid]
        x_positions = [i for i, c in enumerate(target) if c == "N"]
        if strand in ("-", "R"):
            x_positions = [len(target) - p - 1 for p in x_positions[::-1]]
        cuts[key] = {k: jsonable(v) for k, v in analyze.compute_stats_error_rate(np.asarray(rates, dtype=float), x_positions,
                                                                                max_dist=4).items()}
    out = {"note": "inputs: POC templates (reference data file), designed reads, the product's FASTQ, the minimap2-shaped PAF; "
                   "outputs: what the reference's compute_all_error_rates_paf / compute_stats_error_rate computed from them",
           "templates": templates, "reads": [[r[0], r[1], r[2]] for r in reads], "fastq": fastq, "paf": paf,
           "per_read": per_read, "error_rate_per_position": per_pos, "error_rate_cuts": cuts}
    with open(os.path.join(HERE, "evalacc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote evalacc.json: %d reads, %d (target, strand) groups" % (len(reads), len(per_pos)))
    for k in sorted(cuts):
        print(k, "UB acc %.1f  DNA acc %.2f" % (100 - np.mean(cuts[k]["only_ub"]), 100 - np.mean(cuts[k]["no_ub"])))


if __name__ == "__main__":
    main()
