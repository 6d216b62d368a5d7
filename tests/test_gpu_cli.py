"""GPU: `bonito basecaller`-compatible CLI end to end (BASELINE configs[0] shape:
16 reads, chunksize 4000, small batch): model directory (config.toml + weights_N.tar with a dropout-interleaved
training state dict) + signal bundle -> FASTQ on stdout + <stem>_summary.tsv, compared with the all-oracle pipeline."""
import os
import subprocess
import sys
from collections import OrderedDict

import numpy as np
import pytest

import oracle
from conftest import ROOT, encoder_shapes, make_config, seeded_state_dict
from xna_basecaller_amd import reads as xreads
from xna_basecaller_amd import toml_lite, util

pytestmark = pytest.mark.gpu

# training-time module indices of the dropout-interleaved encoder (crf/model.py:183-201) -> inference indices
TRAIN_INDEX = {0: 0, 1: 2, 2: 4, 4: 7, 5: 9, 6: 11, 7: 13, 8: 15, 9: 16}


def _make_model_dir(path, features, labels, seed):
    import torch
    cfg = make_config(features, labels)
    cfg["model"]["package"] = "bonito.crf"
    cfg["basecaller"] = {"batchsize": 5, "chunksize": 4000, "overlap": 500}
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.toml"), "w") as fh:
        fh.write(toml_lite.dumps(cfg))
    keys, shapes = encoder_shapes(features, len(labels) - 1)
    sd = seeded_state_dict(keys, shapes, seed)
    train = OrderedDict()
    for k, v in sd.items():
        idx = int(k.split(".")[1])
        train["module." + k.replace("encoder.%d." % idx, "encoder.%d." % TRAIN_INDEX[idx])] = torch.from_numpy(v)
    torch.save(train, os.path.join(path, "weights_3.tar"))
    torch.save({k: torch.zeros_like(v) for k, v in train.items()}, os.path.join(path, "weights_1.tar"))  # older checkpoint
    return cfg, sd


def _make_reads(path, n):
    rng = np.random.default_rng(11)
    recs = []
    for i in range(n):
        length = int(rng.integers(3000, 16000))
        base = rng.normal(90.0, 12.0, length)
        lead = int(rng.integers(300, 900))
        base[:lead] = rng.normal(140.0, 3.0, lead)
        raw = np.round(base * 8.0).astype(np.int16)
        recs.append((raw, dict(read_id="read-%02d" % i, range=1443.03, digitisation=8192.0, offset=10,
                               sampling_rate=4000.0, run_id="runX", channel_number=str(100 + i), start_mux=1 + i % 4,
                               read_number=i, start_time=4000 * i, duration=length,
                               exp_start_time="2021-06-01T10:00:00Z")))
    os.makedirs(path, exist_ok=True)
    xreads.write_bundle(os.path.join(path, "batch0.xsig.npz"), recs[: n // 2])
    xreads.write_bundle(os.path.join(path, "batch1.xsig.npz"), recs[n // 2:])


def test_cli_basecaller_end_to_end(tmp_path):
    labels = list("NACGTXY")
    model_dir = str(tmp_path / "xna_test@v1")
    reads_dir = str(tmp_path / "reads")
    cfg, sd = _make_model_dir(model_dir, 64, labels, seed=21)
    _make_reads(reads_dir, 16)
    ids = tmp_path / "ids.tsv"
    ids.write_text("".join("read-%02d\n" % i for i in range(16) if i != 5))
    out = tmp_path / "calls.fastq"
    with open(out, "w") as fh:
        r = subprocess.run([sys.executable, "-m", "xna_basecaller_amd", "basecaller", model_dir, reads_dir,
                            "--read-ids", str(ids), "--batch", "7", "-v"], cwd=ROOT, stdout=fh, stderr=subprocess.PIPE,
                           timeout=600)
    err = r.stderr.decode()
    assert r.returncode == 0, err
    assert "> outputting unaligned fastq" in err and "> samples per second" in err and "> completed reads: 15" in err
    recs = out.read_text().strip().split("\n")
    assert len(recs) == 4 * 15
    summary = (tmp_path / "calls_summary.tsv").read_text().strip().split("\n")
    assert len(summary) == 16 and summary[0].split("\t")[1] == "read_id"

    # oracle pipeline on the same reads (file order, read-id filter applied)
    expect = [rd for rd in xreads.get_reads(reads_dir, read_ids=util.column_to_set(str(ids)))]
    assert [rec[1:].split(" ")[0] for rec in recs[0::4]] == [rd.read_id for rd in expect]
    total = mism = 0
    for rd, hdr, seq, qs in zip(expect, recs[0::4], recs[1::4], recs[3::4]):
        ch = util.chunk(np.asarray(rd.signal, np.float32), 4000, 500)
        lab = oracle.decode(oracle.encode(ch, sd, 64, 6, 3), 6, 3)["labels"]
        packed, _, _ = oracle.pack(lab, "".join(labels))
        st = util.stitch(packed, 4000, 500, len(rd.signal), 5)
        ref = st[st != 0].astype(np.uint8).tobytes().decode()
        assert hdr.startswith("@%s RG:Z:runX_%s\tqs:i:40\tmx:i:" % (rd.read_id, model_dir))
        assert qs == "O" * len(seq)
        total += max(len(ref), 1)
        if seq != ref:
            mism += sum(a != b for a, b in zip(seq, ref)) + abs(len(seq) - len(ref))
    assert mism <= total // 500, (mism, total)       # encoder differences of ~1e-5 may flip a near-tie


def test_cli_reads_fast5_like_bundles(tmp_path):
    """The same reads as multi-read fast5 files (HDF5 + VBZ, parsed by hdf5_lite) and as signal bundles give the same
    FASTQ records (apart from the f5:Z: source-file tag)."""
    from h5write import write_multi_fast5
    labels = list("NACGTXY")
    model_dir = str(tmp_path / "xna_test@v1")
    _make_model_dir(model_dir, 64, labels, seed=21)
    rng = np.random.default_rng(12)
    recs = []
    for i in range(8):
        length = int(rng.integers(3000, 12000))
        base = rng.normal(90.0, 12.0, length)
        base[: int(rng.integers(300, 900))] = 140.0
        recs.append((np.round(base * 8.0).astype(np.int16),
                     dict(read_id="aaaa-%02d" % i, range=1443.03, digitisation=8192.0, offset=10, sampling_rate=4000.0,
                          run_id="runX", channel_number=str(100 + i), start_mux=1 + i % 4, read_number=i,
                          start_time=4000 * i, duration=length, exp_start_time="2021-06-01T10:00:00Z")))
    (tmp_path / "f5").mkdir()
    (tmp_path / "npz").mkdir()
    write_multi_fast5(str(tmp_path / "f5" / "batch_0.fast5"), recs, vbz=True)
    xreads.write_bundle(str(tmp_path / "npz" / "batch_0.xsig.npz"), recs)
    outs = {}
    for kind in ("f5", "npz"):
        out = tmp_path / ("calls_%s.fastq" % kind)
        with open(out, "w") as fh:
            r = subprocess.run([sys.executable, "-m", "xna_basecaller_amd", "basecaller", model_dir, str(tmp_path / kind),
                                "--batch", "6"], cwd=ROOT, stdout=fh, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()
        outs[kind] = out.read_text().replace("f5:Z:batch_0.fast5", "f5:Z:X").replace("f5:Z:batch_0.xsig.npz", "f5:Z:X")
    assert outs["f5"] == outs["npz"] and outs["f5"].count("\n") == 32


def test_cli_sam_text_output_of_the_beam_branch(tmp_path):
    """`bonito basecaller MODEL READS > calls.sam` (io.py:30-49: the extension of stdout's target selects SAM) on a 4-base model,
    whose default decode is the beam search with real quality strings: header (@HD, @PG basecaller, one @RG per run), one
    unaligned record per read (flag 4), the same sequences / qualities / tags as the FASTQ of the same run."""
    labels = list("NACGT")
    model_dir = str(tmp_path / "dna_test@v1")
    reads_dir = str(tmp_path / "reads")
    _make_model_dir(model_dir, 64, labels, seed=5)
    _make_reads(reads_dir, 6)
    outs = {}
    for ext in ("sam", "fastq"):
        out = tmp_path / ("calls." + ext)
        with open(out, "w") as fh:
            r = subprocess.run([sys.executable, "-m", "xna_basecaller_amd", "basecaller", model_dir, reads_dir, "--batch", "7", "-v"],
                               cwd=ROOT, stdout=fh, stderr=subprocess.PIPE, timeout=600)
        err = r.stderr.decode()
        assert r.returncode == 0, err
        assert "> outputting unaligned %s" % ext in err and "> decode algorithm: Beam Search" in err
        outs[ext] = out.read_text()
    lines = outs["sam"].splitlines()
    head = [l for l in lines if l.startswith("@")]
    body = [l for l in lines if not l.startswith("@")]
    assert head[0] == "@HD\tVN:1.5\tSO:unknown\tob:0.0.1"
    assert head[1].startswith("@PG\tID:basecaller\tPN:bonito\tVN:") and ("CL:bonito basecaller %s %s" % (model_dir, reads_dir)) in head[1]
    assert head[2:] == ["@RG\tID:runX_%s\tPL:ONT\tDT:2021-06-01T10:00:00\tPU:\tPM:None\tLB:None\tSM:None\tDS:run_id=runX basecall_model=%s"
                        % (model_dir, model_dir)]
    fq = outs["fastq"].strip().split("\n")
    assert len(body) == 6 == len(fq) // 4
    for rec, hdr, seq, qs in zip(body, fq[0::4], fq[1::4], fq[3::4]):
        f = rec.split("\t")
        assert f[1:9] == ["4", "*", "0", "0", "*", "*", "0", "0"] and f[11] == "NM:i:0"
        assert f[9] == seq and f[10] == qs and len(set(qs)) > 1          # real qualities, not the Viterbi branch's placeholder
        assert "@" + f[0] + " " + "\t".join(f[12:]) == hdr
    assert (tmp_path / "calls_summary.tsv").read_text().count("\n") == 2 * 6 + 1     # both runs appended their rows
