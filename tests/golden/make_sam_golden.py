#!/usr/bin/env python
"""
Fixture generator for the SAM text output (second half): `bonito basecaller ... > calls.sam`.

Run in the BUILD container only.  It imports the reference's ub-bonito/bonito/io.py and fast5.py BY FILE PATH and calls, as
reference code, `sam_header(groups)` (io.py:87-112), `sam_record(read_id, sequence, qstring, mapping=False, tags)` -- the
unaligned branch, flag 4 (io.py:115-145) -- and `Read.readgroup(model)` / `Read.tagdata()` (fast5.py:106-128).  Absent third
party modules those functions never touch at call time (pysam, pandas' users, ont_fast5_api, the `bonito` package object) are
placeholders in sys.modules; `mappy` is a placeholder that only carries the version string the reference pins
(requirements.txt:2, mappy==2.23), which sam_header prints in its @PG aligner line; `bonito.__version__` is read from the
reference's bonito/__init__.py:8.  What is stored in tests/golden/sam.json is DATA: the inputs (read attributes, sequences,
quality strings, argv) and the strings the reference functions returned.  Aligned records (mapping != False) need a mappy
alignment object, i.e. minimap2: not in any image, not generated.  The bytes pysam 0.18 (`AlignmentFile(fd, 'w',
text=sam_header(groups))`, io.py:391-401) puts around them -- the header text as is, one line per record -- are stated in
xna_basecaller_amd/io.py, not generated here (pysam is in no image).
"""
import importlib.util
import json
import os
import re
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/ub-bonito/bonito"


def _placeholder(name, **attrs):
    m = sys.modules.setdefault(name, types.ModuleType(name))
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def load_reference():
    version = re.search(r"__version__ = '([^']+)'", open(os.path.join(REF, "__init__.py")).read()).group(1)
    for name in ("pysam", "ont_fast5_api", "ont_fast5_api.fast5_interface"):
        _placeholder(name)
    _placeholder("pysam", AlignmentFile=object, AlignmentHeader=object, AlignedSegment=object)
    _placeholder("ont_fast5_api.fast5_interface", get_fast5_file=None)
    # mappy.revcomp is the one mappy FUNCTION the aligned branch of sam_record calls (io.py:133).  mappy is in no image; the
    # stand-in below restates minimap2's complement table (sketch.c seq_comp_table: the IUPAC codes, both cases, every
    # other byte unchanged) -- so the XNA letter X stays X while Y, an IUPAC code, becomes R.  Flagged in sam.json.
    comp = {a: b for a, b in zip("ACGTUMRWSYKVHDBN", "TGCAAKYWSRMBDHVN")}
    comp.update({a.lower(): b.lower() for a, b in list(comp.items())})
    _placeholder("mappy", __version__="2.23", revcomp=lambda seq: "".join(comp.get(c, c) for c in reversed(seq)))
    pkg = _placeholder("bonito", __version__=version)
    pkg.__path__ = []
    _placeholder("bonito.cli")
    _placeholder("bonito.cli.convert", typical_indices=None)
    _placeholder("bonito.util", mean_qscore_from_qstring=None)
    mods = {}
    for name in ("io", "fast5"):
        spec = importlib.util.spec_from_file_location("ref_bonito_" + name, os.path.join(REF, name + ".py"))
        mods[name] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mods[name])
    return mods["io"], mods["fast5"], version


def main():
    rio, rfast5, version = load_reference()
    model = "xna_r9.4.1_e8_sup@v3.3"
    argv = ["basecaller", model, "reads", "--read-ids", "ids.tsv", "--batchsize", "512"]
    reads = []
    for i in range(4):
        reads.append(dict(read_id="%08x-aaaa-4bbb-8ccc-%012d" % (0xABC000 + i, i), run_id="run%02d" % (i % 2),
                          exp_start_time="2021-03-04T05:06:07Z", flow_cell_id="FAK%05d" % (i % 2), device_id="MN%05d" % (i % 2),
                          sample_id="poc_lib_%d" % (i % 2), mux=1 + i % 4, channel=100 + i,
                          start_time="2021-03-04T05:%02d:07Z" % i, read_number=7 * i, filename="batch_%d.fast5" % (i // 2)))
    groups, tag_lists = [], []
    for r in reads:
        ns = types.SimpleNamespace(**r)
        groups.append(rfast5.Read.readgroup(ns, model))
        tag_lists.append(rfast5.Read.tagdata(ns))
    groups = sorted(set(groups))
    old_argv = sys.argv
    sys.argv = ["bonito"] + argv
    try:
        header = rio.sam_header(groups)
    finally:
        sys.argv = old_argv
    seqs = ["ACGTXYACGT", "GGGTTTAXAYA", "T", "ACGTACGTACGTXXYY"]
    quals = ["%&'()*+,-.", "OOOOOOOOOOO", "I", "5555566666777788"]
    records = []
    for r, tags, s, q in zip(reads, tag_lists, seqs, quals):
        full = ["RG:Z:%s_%s" % (r["run_id"], model), "qs:i:%d" % (7 + len(s) % 5)] + list(tags)
        records.append({"read": r, "sequence": s, "qstring": q, "tags": full,
                        "sam_record": rio.sam_record(r["read_id"], s, q, False, tags=full),
                        "sam_record_no_tags": rio.sam_record(r["read_id"], s, q, False)})
    # ---- aligned records (io.py:118-137): the reference's sam_record with a stand-in for the mappy.Alignment object -- the
    # attributes it reads: ctg, r_st, q_st, q_en, strand, mapq, cigar_str, NM, MD.  Both strands, soft clips at neither /
    # one / both ends, with and without tags.
    aligned = []
    cases = [dict(ctg="POC_T1", r_st=0, r_en=10, q_st=0, q_en=10, strand=1, mapq=60, cigar_str="10M", NM=0, MD="10"),
             dict(ctg="POC_T1", r_st=3, r_en=12, q_st=2, q_en=11, strand=1, mapq=37, cigar_str="4M1I4M", NM=2, MD="3A4"),
             dict(ctg="POC_T2", r_st=5, r_en=13, q_st=0, q_en=8, strand=-1, mapq=12, cigar_str="3M1D5M", NM=1, MD="3^C5"),
             dict(ctg="CPLX_7", r_st=100, r_en=112, q_st=1, q_en=13, strand=-1, mapq=0, cigar_str="12M", NM=3, MD="2T4G3A0"),
             dict(ctg="POC_T3", r_st=41, r_en=42, q_st=0, q_en=1, strand=-1, mapq=1, cigar_str="1M", NM=0, MD="1")]
    aseqs = ["ACGTXYACGT", "GGGTTTAXAYA", "ACGTXAYT", "ACGTACGTACGTXXYY", "T"]
    aquals = ["%&'()*+,-.", "OOOOOOOOOOO", "12345678", "5555566666777788", "I"]
    for i, (m, s, q) in enumerate(zip(cases, aseqs, aquals)):
        tags = ["RG:Z:run00_%s" % model, "qs:i:%d" % (9 + i)]
        mp = types.SimpleNamespace(**m)
        aligned.append({"read_id": "aligned-%d" % i, "sequence": s, "qstring": q, "mapping": m, "tags": tags,
                        "sam_record": rio.sam_record("aligned-%d" % i, s, q, mp, tags=tags),
                        "sam_record_no_tags": rio.sam_record("aligned-%d" % i, s, q, mp)})
    out = {"note": "strings returned by the reference's sam_header / sam_record (unaligned, and aligned with a stand-in mapping "
                   "object and a stand-in for mappy.revcomp restating minimap2's complement table) / Read.readgroup / Read.tagdata",
           "aligned": aligned,
           "bonito_version": version, "mappy_version": "2.23", "model": model, "argv": argv, "linesep": os.linesep,
           "groups": groups, "header": header, "records": records}
    with open(os.path.join(HERE, "sam.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote sam.json: header %d bytes, %d groups, %d records" % (len(header), len(groups), len(records)))
    print(header)
    print(records[0]["sam_record"])


if __name__ == "__main__":
    main()
