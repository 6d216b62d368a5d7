"""
Test infrastructure (like oracle/): a restatement of the reference's UB / DNA accuracy metrics -- src/misc/utils.py
(parse_cs_flag :87-110, compute_read_matches :112-191, polish_target_matches :661-725, compute_errors_paf :727-770,
compute_error_rate_per_pos_paf :772-966, compute_all_error_rates_paf :1198-1253) and src/tools/analyze_paf.py
(compute_stats_error_rate :111-190) -- written from what those functions compute, to check that FASTQ produced by this
package, aligned the way eval_model.sh:128-132 aligns it, yields the accuracy table the reference's own code yields
(tests/golden/evalacc.json holds the reference's numbers; tests/golden/make_evalacc_golden.py made them).
Nothing in the product imports this module.
"""
import re

import numpy as np

_PAIR = dict(zip("ACGTXYN-acgtxyn", "TGCAYXN-tgcayxn"))
_CS_TOKEN = re.compile(r":[0-9]+|\*[a-z]{2}|[=+-][A-Za-z]+|~[a-z]{2}[0-9]+[a-z]{2}")
KMER = 6


def revcomp(seq):
    """Reverse complement with the XNA pair X <-> Y (utils.py:25-31)."""
    return "".join(_PAIR[c] for c in reversed(seq))


def aligned_columns(read_fwd, cs, t_start, t_end, t_len):
    """The read's letter opposite every template position ('-' where nothing is aligned: outside [t_start, t_end) and at
    deletions); inserted read letters are dropped.  `read_fwd` is the aligned part of the read in template orientation."""
    cols = ["-"] * t_start
    at = 0
    for tok in _CS_TOKEN.findall(cs):
        kind, val = tok[0], tok[1:]
        if kind in ":=":
            n = int(val) if kind == ":" else len(val)
            cols.extend(read_fwd[at:at + n])
            at += n
        elif kind == "*":
            cols.append(read_fwd[at])
            at += 1
        elif kind == "+":
            at += len(val)
        elif kind == "-":
            cols.extend("-" * len(val))
        else:
            raise NotImplementedError("introns do not occur in map-ont alignments")
    cols.extend("-" * (t_len - t_end))
    assert at == len(read_fwd) and len(cols) == t_len
    return cols


def polish(cols, template):
    """minimap2 sometimes puts the UB letter beside a gap instead of on the template's UB column; move it there
    (utils.py:661-725, always in template orientation, UB letter X)."""
    out = list(cols)
    last = len(cols) - 1
    for p in (m.start() for m in re.finditer("X", template)):
        here = cols[p]
        if here == "X":
            continue
        if here == "-":
            lo = hi = p
            while lo > 0 and cols[lo - 1] == "-":
                lo -= 1
            while hi < last and cols[hi + 1] == "-":
                hi += 1
            if lo != 0 and cols[lo - 1] == "X":
                out[lo - 1], out[p] = "-", "X"
            elif hi != last and cols[hi + 1] == "X":
                out[hi + 1], out[p] = "-", "X"
        elif cols[p - 1] == "-" and cols[p + 1] == "X":
            out[p - 1], out[p], out[p + 1] = out[p], "X", "-"
        elif cols[p + 1] == "-" and cols[p - 1] == "X":
            out[p + 1], out[p], out[p - 1] = out[p], "X", "-"
    return out


def read_metrics(row, template, read_seq):
    """Per-read error vector (read orientation) and the UB-area metrics of one PAF row.  `template` carries N at its UB
    positions; XNA templates are compared with the UB written as X (utils.py:781-786), PC templates as they are."""
    if not row["target_id"].startswith("PC"):
        template = template.replace("N", "X")
    seg = read_seq[row["read_start"]:row["read_end"]]
    assert len(read_seq) == row["read_length"]
    minus = row["strand"] in ("-", "R")
    if minus:
        seg = revcomp(seg)
    cols = polish(aligned_columns(seg, row["cs"], row["target_start"], row["target_end"], row["target_length"]), template)
    wrong = (np.array(list(template)) != np.array(cols)).astype(float)
    if minus:
        wrong = wrong[::-1]
    n = len(template)
    ubs = [m.start() for m in re.finditer("[NXY]", template)]
    area = np.zeros(n, bool)
    for p in ubs:
        area[max(p + 1 - KMER, 0):p + KMER] = True
    area[ubs] = False
    area_incl = area.copy()
    area_incl[ubs] = True
    area_seq = "".join(np.array(cols)[area_incl])
    aligned = np.zeros(n, bool)
    aligned[row["target_start"]:row["target_end"]] = True
    if minus:
        area_seq = revcomp(area_seq)
        area, area_incl, aligned = area[::-1], area_incl[::-1], aligned[::-1]
        ubs = [n - p - 1 for p in ubs[::-1]]
    ok = wrong == 0
    ub_ok, area_ok = int(ok[ubs].sum()), int(ok[area].sum())
    called_ub = int(np.isin(cols, ["X", "Y"]).sum())
    false_ub = called_ub - ub_ok
    outside = ~area_incl
    span = row["target_end"] - row["target_start"]
    m = {
        "n_matches": float(n - wrong.sum()),
        "ub_area_matches": area_ok, "ub_area_len": int(area.sum()), "ub_area_seq": area_seq,
        "ub_matches": ub_ok, "ub_len": len(ubs),
        "non_ub_area_matches": int(ok[outside].sum()), "non_ub_area_len": int(outside.sum()),
        "target_alig_acc": int(span - wrong[aligned].sum()) / span,
        "fdr": false_ub / called_ub if called_ub else float("nan"),
        "fpr": false_ub / (n - len(ubs)),
        "true_pos": ub_ok, "false_neg": len(ubs) - ub_ok, "true_neg": n - len(ubs) - false_ub, "false_pos": false_ub,
    }
    m["non_ub_area_acc"] = m["non_ub_area_matches"] / m["non_ub_area_len"]
    if ubs:
        m["ub_area_acc"] = area_ok / m["ub_area_len"]
        m["ub_acc"] = ub_ok / len(ubs)
        m["ub_area_acc_plus"] = (area_ok + ub_ok) / (m["ub_area_len"] + len(ubs))
    else:
        m["ub_area_acc"] = m["ub_acc"] = m["ub_area_acc_plus"] = float("nan")
    m["read_acc"] = m["n_matches"] / (row["read_end"] - row["read_start"])
    m["target_acc"] = m["n_matches"] / row["target_length"]
    return wrong, m


def error_rate_cuts(rate, ubs, max_dist=4):
    """Per-position error rates grouped by their relation to the UB positions (analyze_paf.py:111-190)."""
    rate = np.asarray(rate, float)
    n = len(rate)
    is_ub = np.zeros(n, bool)
    is_ub[ubs] = True
    near = np.zeros(n, bool)
    for p in ubs:
        near[max(p + 1 - KMER, 0):p + KMER] = True
    near[ubs] = True
    dist = np.array([min(abs(p - q) for p in ubs) for q in range(n)])
    cuts = {"only_ub": rate[is_ub], "no_ub": rate[~is_ub], "outside_ub_area": rate[~near],
            "inside_ub_area": rate[near & ~is_ub], "ub_and_ub_area": rate[near]}
    for d in range(1, max_dist + 1):
        cuts["dist_ub_d-%d" % d] = rate[dist == d]
    cuts["dist_ub_d-%d+" % (max_dist + 1)] = rate[dist >= max_dist + 1]
    return cuts


def parse_paf(text):
    """PAF rows as dicts: the 12 mandatory columns under the names misc/data_io.py:14-16 gives them, plus the cs tag."""
    names = ["read_id", "read_length", "read_start", "read_end", "strand", "target_id", "target_length", "target_start",
             "target_end", "n_matches", "block_length", "mapping_quality"]
    rows = []
    for line in text.strip().split("\n"):
        f = line.split("\t")
        row = {k: (v if k in ("read_id", "strand", "target_id") else int(v)) for k, v in zip(names, f)}
        row["cs"] = [t for t in f[12:] if t.startswith("cs:Z:")][0][5:]
        rows.append(row)
    return rows
