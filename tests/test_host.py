"""CPU: host-side mirror of the reference interface (util / reads / io / toml / model container) against the
golden fixtures produced by the reference's own functions (tests/golden/make_golden.py)."""
import io as _io
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, SHIPPED_CONFIG, make_config
from xna_basecaller_amd import toml_lite, util
from xna_basecaller_amd import io as xio
from xna_basecaller_amd import reads as xreads


def test_chunk_and_stitch_match_reference():
    cases = json.load(open(os.path.join(GOLDEN, "chunk_stitch.json")))["cases"]
    assert len(cases) >= 15
    for c in cases:
        length, L, ov, stride = c["length"], c["chunksize"], c["overlap"], c["stride"]
        sig = np.arange(length, dtype=np.float32) + 1.0
        ch = util.chunk(sig, L, ov)
        assert ch.shape == (c["n_chunks"], 1, L)
        assert (ch[:, 0, 0].astype(np.int64) - 1).tolist() == c["starts"]
        assert int((ch[0, 0] == 0).sum()) == c["left_pad"]
        if length >= L:
            assert util.chunk_starts(length, L, ov).tolist() == c["starts"]
        T = L // stride
        rows = np.arange(c["n_chunks"] * T, dtype=np.int32).reshape(c["n_chunks"], T) + 1
        st = util.stitch(rows, L, ov, length, stride)
        assert (np.asarray(st).astype(np.int64) - 1).tolist() == c["stitch"]
        if c["n_chunks"] > 1:
            st = util.stitch(rows, L, ov, length, stride, reverse=True)
            assert (np.asarray(st).astype(np.int64) - 1).tolist() == c["stitch_reverse"]


def test_stitch_survey_example():
    # len 23000, L 10000, ov 500 -> starts 0, 3500, 13000 and 4600 slots
    assert util.chunk_starts(23000, 10000, 500).tolist() == [0, 3500, 13000]
    rows = np.zeros((3, 2000), np.int8)
    assert util.stitch(rows, 10000, 500, 23000, 5).shape == (4600,)


def test_batchify_unbatchify_match_reference():
    cases = json.load(open(os.path.join(GOLDEN, "batchify.json")))["cases"]
    for c in cases:
        items = [(("read%d" % i, 0, n * 10), np.full((n, 1, 6), float(i), np.float32)) for i, n in enumerate(c["n_chunks"])]
        batches = list(util.batchify(iter(items), c["batchsize"]))
        keys = [[[list(k[0]), list(k[1])] for k in ks] for ks, v in batches]
        assert keys == c["keys"]
        assert [int(v.shape[0]) for ks, v in batches] == c["batch_sizes"]
        un = [[list(k), int(v.shape[0]), float(v[0, 0, 0])] for k, v in util.unbatchify(iter(batches))]
        assert un == c["unbatch"]


def test_unbatchify_dict_values():
    items = [(("a",), np.zeros((3, 1, 4), np.float32)), (("b",), np.ones((4, 1, 4), np.float32))]
    batches = [(ks, {"x": v[:, 0, :2], "y": v[:, 0, 0]}) for ks, v in util.batchify(iter(items), 2)]
    out = list(util.unbatchify(iter(batches)))
    assert [k for k, _ in out] == [("a",), ("b",)]
    assert out[0][1]["x"].shape == (3, 2) and out[1][1]["y"].tolist() == [1, 1, 1, 1]


def test_signal_preparation_matches_reference():
    z = np.load(os.path.join(GOLDEN, "signal_prep.npz"))
    offset, rng, digi = int(z["offset"]), float(z["range"]), float(z["digitisation"])
    for i in range(int(z["n"])):
        raw = z["raw%d" % i]
        scaled = np.array((rng / digi) * (raw + offset), dtype=np.float32)
        t0, tlen = xreads.trim(scaled[:8000])
        assert [t0, tlen] == z["trim%d" % i].tolist()
        trimmed = scaled[t0:]
        med, mad = xreads.med_mad(trimmed)
        assert np.allclose([med, mad], z["medmad%d" % i], rtol=0, atol=0)
        attrs = dict(read_id="r%d" % i, range=rng, digitisation=digi, offset=offset, sampling_rate=4000.0)
        read = xreads.Read(raw, attrs, "x.xsig.npz")
        assert read.signal.dtype == np.float32 or read.signal.dtype == np.float64
        assert np.array_equal(np.asarray(read.signal, np.float32), z["signal%d" % i])
        assert np.array_equal(np.asarray(xreads.norm_by_noisiest_section(trimmed[:7000]), np.float32), z["noisiest%d" % i])


def test_bundle_roundtrip(tmp_path):
    z = np.load(os.path.join(GOLDEN, "signal_prep.npz"))
    recs = []
    for i in range(3):
        recs.append((z["raw%d" % i], dict(read_id="read-%d" % i, range=float(z["range"]), digitisation=float(z["digitisation"]),
                                           offset=int(z["offset"]), sampling_rate=4000.0, run_id="run1", channel_number="7",
                                           start_mux=2, read_number=i, start_time=4000 * i, duration=len(z["raw%d" % i]),
                                           exp_start_time="2021-06-01T10:00:00Z")))
    xreads.write_bundle(str(tmp_path / "a.xsig.npz"), recs)
    got = list(xreads.get_reads(str(tmp_path)))
    assert [r.read_id for r in got] == ["read-0", "read-1", "read-2"]
    assert np.array_equal(np.asarray(got[1].signal, np.float32), z["signal1"])
    assert got[2].tagdata() == ["mx:i:2", "ch:i:7", "st:Z:2021-06-01T10:00:02", "rn:i:2", "f5:Z:a.xsig.npz"]
    only = list(xreads.get_reads(str(tmp_path), read_ids={"read-1"}))
    assert [r.read_id for r in only] == ["read-1"]
    skip = list(xreads.get_reads(str(tmp_path), read_ids={"read-1"}, skip=True))
    assert [r.read_id for r in skip] == ["read-0", "read-2"]


def test_toml_roundtrip_shipped_config():
    text = toml_lite.dumps(SHIPPED_CONFIG)
    assert toml_lite.loads(text) == SHIPPED_CONFIG
    # the reference file's own formatting: trailing comma arrays, spaces inside brackets
    cfg = toml_lite.loads('[labels]\nlabels = [ "N", "A", "C", "G", "T", "X", "Y",]\n\n[encoder]\nscale = 5.0\nstride = 5\n'
                          'rnn_type = "lstm" # comment\nflag = true\n[a.b]\nx = [1,\n 2,\n 3]\n')
    assert cfg["labels"]["labels"] == list("NACGTXY")
    assert cfg["encoder"] == {"scale": 5.0, "stride": 5, "rnn_type": "lstm", "flag": True}
    assert cfg["a"]["b"]["x"] == [1, 2, 3]


def test_match_names_and_model_container():
    meta = json.load(open(os.path.join(GOLDEN, "encoder_meta.json")))
    from xna_basecaller_amd.crf.model import Model
    mm = meta["match_names"]
    model = Model(make_config(32))
    assert list(model.state_dict().keys()) == mm["model_keys"]
    import torch
    sd = {k: torch.zeros(s) for k, s in zip(mm["train_keys"], mm["train_shapes"])}
    remap = util.match_names(sd, model)
    assert [[k, v] for k, v in remap.items()] == mm["remap"]
    # shipped geometry: 24 854 904 parameters, stride 5, 1512 scores, idx table
    full = Model(make_config(768))
    assert sum(p.numel() for p in full.parameters()) == 24854904
    assert full.stride == 5 and full.seqdist.n_score() == 1512
    assert full.seqdist.idx.shape == (216, 7)
    assert full.seqdist.idx[215].tolist() == [215, 35, 71, 107, 143, 179, 215]
    assert full.encoder[-1].expand_blanks and full.encoder[-1].blank_score == 2.0
    z = np.load(os.path.join(GOLDEN, "crf_idx.npz"))
    assert np.array_equal(full.seqdist.idx.numpy(), z["idx_nb6"])
    assert full.seqdist.path_to_str(np.array([0, 1, 0, 6, 5, 0])) == "AYX"


def test_mean_qscore_and_fastq_record():
    meta = json.load(open(os.path.join(GOLDEN, "encoder_meta.json")))
    for q, v in meta["qscore"]:
        assert abs(util.mean_qscore_from_qstring(q) - v) < 1e-9
    assert util.mean_qscore_from_qstring("O" * 12) == pytest.approx(40.0)
    fd = _io.StringIO()
    xio.write_fastq("rid", "ACGX", "OOOO", fd=fd, tags=["RG:Z:run_m", "qs:i:40", "mx:i:1"])
    assert fd.getvalue() == "@rid RG:Z:run_m\tqs:i:40\tmx:i:1\nACGX\n+\nOOOO\n"


def test_writer_fastq_and_summary(tmp_path):
    reads = [xreads.SyntheticRead("r%d" % i, np.zeros(100 * (i + 1), np.float32), run_id="runA") for i in range(3)]
    results = [(reads[0], {"sequence": "ACGT", "qstring": "OOOO"}), (reads[1], {"sequence": "", "qstring": ""}),
               (reads[2], {"sequence": "XY", "qstring": "OO"})]
    fd = _io.StringIO()
    w = xio.Writer("wfq", iter(results), fd=fd, group_key="model_dir", summary=str(tmp_path / "s.tsv"))
    w.start()
    w.join()
    assert w.error is None
    assert w.log == [("r0", 100), ("r2", 300)]          # the empty sequence is skipped
    lines = fd.getvalue().split("\n")
    assert lines[0] == "@r0 RG:Z:runA_model_dir\tqs:i:40\tmx:i:1\tch:i:1\tst:Z:1970-01-01T00:00:00\trn:i:0\tf5:Z:synthetic.xsig.npz"
    assert lines[1:4] == ["ACGT", "+", "OOOO"]
    rows = open(tmp_path / "s.tsv").read().strip().split("\n")
    assert rows[0].split("\t")[:3] == ["filename", "read_id", "run_id"] and len(rows) == 3


def test_summary_table_bytes_follow_the_csv_dialect(tmp_path):
    """The reference writes the summary with csv.writer(delimiter='\\t') on a newline='' file (io.py:322-356): CRLF line
    ends, minimal quoting, floats as str(); an existing file is appended to without a second header."""
    import csv
    rows = [{"filename": "a b.fast5", "read_id": "r\t1", "run_id": 'q"x', "channel": "12", "mux": 3, "start_time": 1.5,
             "duration": 0.25, "template_start": 1.75, "template_duration": 1e-05, "sequence_length_template": 7,
             "mean_qscore_template": 40.0},
            {"filename": "b.fast5", "read_id": "r2", "run_id": "run", "channel": 1, "mux": 1, "start_time": 0.0,
             "duration": 2.0, "template_start": 0.1, "template_duration": 1.9, "sequence_length_template": 120,
             "mean_qscore_template": 39.99999999999999}]
    mine, ref = tmp_path / "mine.tsv", tmp_path / "ref.tsv"
    for half in (rows[:1], rows[1:]):                   # second pass appends to the existing files
        with xio.SummaryTable(mine) as t:
            for r in half:
                t.append(r)
        new = not ref.exists()
        with open(ref, "a", newline="") as fh:
            w = csv.writer(fh, delimiter="\t")
            if new:
                w.writerow(list(xio.SUMMARY_COLUMNS))
            for r in half:
                w.writerow([r.get(k, "-") for k in xio.SUMMARY_COLUMNS])
    assert mine.read_bytes() == ref.read_bytes()
    assert mine.read_bytes().count(b"\r\n") == 3


def test_cli_argparser_defaults():
    from xna_basecaller_amd.cli.basecaller import argparser
    from argparse import ArgumentParser
    p = ArgumentParser(parents=[argparser()])
    a = p.parse_args(["m", "r", "--batch", "98", "--read-ids", "ids.tsv"])      # prefix abbreviation, eval_model.sh:29
    assert a.batchsize == 98 and a.device == "cuda" and a.seed == 25 and a.weights == "0"
    assert a.chunksize is None and a.overlap is None and a.use_koi is True and a.quantize is None


def test_reverse_complement_matches_reference():
    """CTC_CRF.reverse_complement (crf/model.py:78-90) against gather maps recorded from the reference's own code."""
    from xna_basecaller_amd.crf.model import CTC_CRF
    z = np.load(os.path.join(GOLDEN, "revcomp.npz"))
    for nb in (4, 5, 6):
        for sl in (2, 3):
            sd = CTC_CRF(sl, list("NACGTXY"[:nb + 1]))
            C = (nb + 1) * nb ** sl
            x = np.arange(3 * 2 * C, dtype=np.float32).reshape(3, 2, C)
            got = sd.reverse_complement(x)
            assert got.dtype == np.float32 and got.flags["C_CONTIGUOUS"]
            assert np.array_equal(got.astype(np.int64), z["nb%d_sl%d" % (nb, sl)])


def _fast5_records(n, seed=5):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        length = int(rng.integers(2500, 15000))
        base = rng.normal(90.0, 12.0, length)
        base[: int(rng.integers(300, 900))] = 140.0
        raw = np.round(base * 8.0).astype(np.int16)
        if i == 1:
            raw[100:110] = [-32768, 32767, -1, 0, 1, 255, 256, -256, -255, 12345]     # delta / zig-zag extremes
        recs.append((raw, dict(read_id="0a1b2c3d-%04d" % i, range=1443.03, digitisation=8192.0, offset=10,
                               sampling_rate=4000.0, run_id="runF5", channel_number=str(7 + i), start_mux=1 + i % 4,
                               read_number=100 + i, start_time=4000 * i, duration=length,
                               exp_start_time="2021-06-01T10:00:00Z", sample_id="s1", flow_cell_id="FAK1",
                               device_id="MN1")))
    return recs


@pytest.mark.parametrize("vbz,vlen", [(True, False), (False, True)])
def test_fast5_reader_equals_bundle_path(tmp_path, vbz, vlen):
    """get_reads(dir of fast5) yields Reads equal to the bundle path: the file is a multi-read fast5
    written by the test itself in the classic HDF5 layout (tests/h5write.py), VBZ-compressed or plain chunks, fixed or
    variable-length string attributes."""
    from h5write import write_multi_fast5
    recs = _fast5_records(6)
    d5, dn = tmp_path / "f5", tmp_path / "npz"
    d5.mkdir()
    dn.mkdir()
    write_multi_fast5(str(d5 / "batch_0.fast5"), recs[:4], vbz=vbz, vlen_strings=vlen)
    # second file: small chunks and a B-tree fan-out of 3, i.e. multi-level chunk and group B-trees
    write_multi_fast5(str(d5 / "batch_1.fast5"), recs[4:], vbz=vbz, vlen_strings=vlen, chunk=1000, fanout=3)
    xreads.write_bundle(str(dn / "all.xsig.npz"), recs)
    a = sorted(xreads.get_reads(str(d5)), key=lambda r: r.read_id)
    b = sorted(xreads.get_reads(str(dn)), key=lambda r: r.read_id)
    assert [r.read_id for r in a] == [r.read_id for r in b] and len(a) == 6
    for x, y in zip(a, b):
        assert np.array_equal(x.signal, y.signal) and x.signal.dtype == np.float32
        for k in ("run_id", "channel", "mux", "read_number", "start", "duration", "template_start",
                  "template_duration", "start_time", "sample_id", "flow_cell_id", "device_id", "offset", "scaling"):
            assert getattr(x, k) == getattr(y, k), k
        assert x.filename.endswith(".fast5") and x.tagdata()[:4] == y.tagdata()[:4]
    # read-id filter, pool and sharding go through the same job list
    ids = {recs[1][1]["read_id"], recs[5][1]["read_id"]}
    assert sorted(r.read_id for r in xreads.get_reads(str(d5), read_ids=ids, n_proc=2)) == sorted(ids)
    assert [r.index for r in xreads.get_reads(str(d5), shard=(1, 2))] == [1, 3, 5]
    # SAM read groups (fast5.py:236-251): metadata only -- the same @RG lines from the fast5 files and from the bundle
    g5, gn = xreads.get_read_groups(str(d5), "m@v1"), xreads.get_read_groups(str(dn), "m@v1")
    assert g5 == gn == {r.readgroup("m@v1") for r in a} and all(g.startswith("@RG\tID:") for g in g5)
    assert xreads._load_read(xreads.read_jobs(str(d5))[0], meta=True).meta and not hasattr(
        xreads._load_read(xreads.read_jobs(str(d5))[0], meta=True), "signal")


def _need_libhdf5():
    import h5lib
    if h5lib.find() is None:
        pytest.skip("no libhdf5 in this container (the GPU box has none: this test never runs there)")
    return h5lib


def test_hdf5_lite_reads_files_written_by_libhdf5(tmp_path):
    """VERDICT r2 (8 f1): every file hdf5_lite had met was laid out by tests/h5write.py, i.e. by the same reading of the format
    specification.  Here the REAL libhdf5 (1.10.6 of the build container, through ctypes: tests/h5lib.py) writes the files:
    chunked + shuffle + deflate and fletcher32 datasets, 1-D and 2-D, contiguous datasets, scalar / array / fixed- and
    variable-length string attributes (global heap), a group with 60 + links (multi-node symbol-table B-tree), both the
    classic and the 1.10 ("latest") object-header / superblock formats."""
    h5lib = _need_libhdf5()
    from xna_basecaller_amd import hdf5_lite
    x = (np.arange(70000) % 3000 - 1500).astype(np.int16)
    z = np.arange(24, dtype=np.int32).reshape(4, 6)
    for latest in (False, True):
        path = str(tmp_path / ("t%d.h5" % latest))
        w = h5lib.Writer(path, latest=latest)
        g = w.group(w.file, "grp")
        # 1.10 format: chunked datasets get data-layout version 4 (fixed-array / extensible-array chunk indexes), which
        # hdf5_lite refuses (no fast5 writer produces it); contiguous datasets and everything else are read
        w.dataset(g, "x", x, chunks=None if latest else 4096, shuffle=True, deflate=4)
        w.dataset(g, "y", np.linspace(0, 1, 11))
        w.dataset(g, "z", z, chunks=None if latest else (2, 3), deflate=1, fletcher32=True)
        if latest:
            w.dataset(g, "v4", x[:5000], chunks=1000, deflate=1)
        w.attr(g, "note", ("vlen", "variable text"))
        w.attr(g, "fixed", "fixed text")
        w.attr(g, "n", np.int64(-3))
        w.attr(g, "f", np.float32(2.5))
        w.attr(g, "arr", np.arange(5, dtype=np.uint16))
        many = w.group(w.file, "many")
        nsub = 60 if not latest else 6          # 1.10 format: more than 8 links switch to dense storage (fractal heap): unsupported
        for i in range(nsub):
            sub = w.group(many, "sub%03d" % i)
            w.attr(sub, "i", np.int32(i))
        w.close()
        with hdf5_lite.File(path) as f:
            assert sorted(f.keys()) == ["grp", "many"]
            assert np.array_equal(f["grp/x"][:], x) and f["grp/x"].dtype == np.int16 and f["grp/x"].shape == (70000,)
            assert np.allclose(f["grp/y"][:], np.linspace(0, 1, 11)) and np.array_equal(f["grp/z"][:], z)
            a = f["grp"].attrs
            assert a["note"] == "variable text" and a["fixed"] == "fixed text" and a["n"] == -3 and a["f"] == 2.5
            assert np.array_equal(a["arr"], np.arange(5))
            if latest:
                with pytest.raises(hdf5_lite.Hdf5Error, match="layout version 4"):
                    f["grp/v4"][:]
            assert sorted(f["many"].keys()) == ["sub%03d" % i for i in range(nsub)]
            assert all(f["many/sub%03d" % i].attrs["i"] == i for i in range(nsub))
    # what stays outside: dense link storage of the 1.10 format -- refused with a message, never mis-read
    path = str(tmp_path / "dense.h5")
    w = h5lib.Writer(path, latest=True)
    for i in range(20):
        w.group(w.file, "g%02d" % i)
    w.close()
    with pytest.raises(hdf5_lite.Hdf5Error, match="fractal heap"):
        with hdf5_lite.File(path) as f:
            f.keys()


def test_fast5_written_by_libhdf5_equals_bundle_path(tmp_path):
    """A multi-read fast5 in the ont_fast5_api layout written by libhdf5 itself (variable-length string attributes, as h5py
    writes them; signal chunked + shuffle + deflate; 60 top-level groups) -> the same Reads as the bundle path and as the
    same records written by tests/h5write.py.  VBZ stays pinned by hand-computed vectors only (test_vbz_known_vectors): ONT's
    filter plugin is not in any image, so libhdf5 cannot write it."""
    h5lib = _need_libhdf5()
    from h5write import write_multi_fast5
    recs = _fast5_records(7)
    dl, dw, dn = tmp_path / "lib", tmp_path / "own", tmp_path / "npz"
    for d in (dl, dw, dn):
        d.mkdir()
    h5lib.write_multi_fast5(str(dl / "batch.fast5"), recs, vlen_strings=True, chunk=1000, fillers=53)
    write_multi_fast5(str(dw / "batch.fast5"), recs, vbz=True)
    xreads.write_bundle(str(dn / "all.xsig.npz"), recs)
    a, b, c = (sorted(xreads.get_reads(str(d)), key=lambda r: r.read_id) for d in (dl, dw, dn))
    assert len(a) == len(b) == len(c) == 7
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x.signal, z.signal) and np.array_equal(y.signal, z.signal)
        for k in ("read_id", "run_id", "channel", "mux", "read_number", "start", "duration", "template_start",
                  "template_duration", "start_time", "sample_id", "flow_cell_id", "device_id", "offset", "scaling"):
            assert getattr(x, k) == getattr(z, k) == getattr(y, k), k


def test_read_loader_lookahead_is_bounded(tmp_path):
    """Pool.imap has no backpressure: the loader hands jobs out as results are taken, so never more than
    `lookahead` prepared reads exist ahead of a slow consumer; order and content equal the serial path."""
    import time
    from h5write import write_multi_fast5
    recs = _fast5_records(24)
    write_multi_fast5(str(tmp_path / "b.fast5"), recs, vbz=True)
    serial = [r.read_id for r in xreads.get_reads(str(tmp_path))]
    loader = xreads.ReadLoader(str(tmp_path), n_proc=3, lookahead=5)
    got = []
    for r in loader:
        time.sleep(0.01)                                   # a slow device stage
        got.append(r.read_id)
    assert got == serial and 1 <= loader.max_pending <= 5
    assert xreads.ReadLoader(str(tmp_path), n_proc=2).lookahead == 64


def test_env_rank_world(monkeypatch):
    from xna_basecaller_amd import dist as xdist
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert xdist.env_rank_world() == (0, 1)
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "3")
    assert xdist.env_rank_world() == (3, 4)


def test_hdf5_lite_basics(tmp_path):
    from h5write import H5Writer
    from xna_basecaller_amd import hdf5_lite
    w = H5Writer(fanout=2)                                  # two entries per node: three-level trees below
    w._vlen_patches = []
    x = np.arange(10, dtype=np.int32)
    y = (np.arange(7000) % 300 - 150).astype(np.int16)
    dx = w.dataset(x, attrs={"unit": "pA", "gain": np.float32(2.5)})
    dy = w.dataset(y, chunks=2048, vbz=True)
    dz = w.dataset(np.linspace(0, 1, 5))
    many = {"d%02d" % i: w.dataset(np.full(3, i, dtype=np.int16)) for i in range(11)}
    g, _, _ = w.group(dict(many, x=dx, y=dy), attrs={"note": "vlen:variable length text", "n": np.int64(-3)})
    root = w.group({"grp": g, "z": dz}, attrs={"file_version": "2.2"})
    w.finish(str(tmp_path / "t.h5"), root)
    with hdf5_lite.File(str(tmp_path / "t.h5")) as f:
        assert sorted(f.keys()) == ["grp", "z"] and f.attrs["file_version"] == "2.2"
        assert "grp/x" in f and "nope" not in f
        assert np.array_equal(f["grp/x"][:], x) and f["grp/x"].shape == (10,)
        assert f["grp/x"].attrs["unit"] == "pA" and f["grp/x"].attrs["gain"] == 2.5
        assert np.array_equal(f["grp"]["y"][:], y) and f["grp/y"].dtype == np.int16
        assert np.allclose(f["z"][:], np.linspace(0, 1, 5))
        assert f["grp"].attrs["note"] == "variable length text" and f["grp"].attrs["n"] == -3
        assert sorted(f["grp"].keys()) == sorted(["d%02d" % i for i in range(11)] + ["x", "y"])
        assert all(f["grp/d%02d" % i][:].tolist() == [i] * 3 for i in range(11))
    with pytest.raises(hdf5_lite.Hdf5Error):
        (tmp_path / "bad.h5").write_bytes(b"not hdf5" * 100)
        hdf5_lite.File(str(tmp_path / "bad.h5"))
    # VBZ round trip on awkward data (wrap-around differences, odd lengths), both stream versions
    for n in (0, 1, 3, 4, 5, 7, 8, 9, 1000):
        v = np.random.default_rng(n).integers(-32768, 32768, n).astype(np.int16)
        for version in (0, 1):
            for level in (0, 1):
                enc = hdf5_lite.vbz_encode_int16(v, level=level, version=version)
                assert np.array_equal(np.frombuffer(hdf5_lite.vbz_decode(enc, [version, 2, 1, level]), dtype="<i2"), v)


def test_vbz_known_vectors():
    """Streams written out by hand from the format descriptions (StreamVByte README: control bytes first, two bits per
    value = byte count - 1, first value in the low bits; vbz_compression: version 0 widens to 32 bits, zig-zag of the
    difference to the previous sample) -- independent of this package's own encoder.
    samples 100, 101, 99, 400, -5 -> differences 100, 1, -2, 301, -405 -> zig-zag 200, 2, 3, 602, 809."""
    from xna_basecaller_amd import hdf5_lite
    want = np.array([100, 101, 99, 400, -5], dtype="<i2")
    # version 0: keys (0, 0, 0, 1) -> 0x40, (1) -> 0x01; data c8 | 02 | 03 | 5a 02 | 29 03
    v0 = bytes([10, 0, 0, 0, 0x40, 0x01, 0xC8, 0x02, 0x03, 0x5A, 0x02, 0x29, 0x03])
    assert np.array_equal(np.frombuffer(hdf5_lite.vbz_decode(v0, [0, 2, 1, 0]), dtype="<i2"), want)
    # version 1 (svb16): one key bit per value, LSB first: 0, 0, 0, 1, 1 -> 0x18; same data bytes
    v1 = bytes([10, 0, 0, 0, 0x18, 0xC8, 0x02, 0x03, 0x5A, 0x02, 0x29, 0x03])
    assert np.array_equal(np.frombuffer(hdf5_lite.vbz_decode(v1, [1, 2, 1, 0]), dtype="<i2"), want)
    # the two layouts are NOT interchangeable: the version field decides (a version-0 stream read as svb16 is garbage)
    assert not np.array_equal(np.frombuffer(hdf5_lite.vbz_decode(v0, [1, 2, 1, 0]), dtype="<i2"), want)
    # version 0 without the delta / zig-zag flag: plain widened values 1, 256, 65535 -> keys (0, 1, 1) = 0x14
    raw = bytes([6, 0, 0, 0, 0x14, 0x01, 0x00, 0x01, 0xFF, 0xFF])
    assert np.frombuffer(hdf5_lite.vbz_decode(raw, [0, 2, 0, 0]), dtype="<u2").tolist() == [1, 256, 65535]
    # our encoder produces exactly the hand-written version-0 stream
    assert hdf5_lite.vbz_encode_int16(want, level=0, version=0) == v0
    assert hdf5_lite.vbz_encode_int16(want, level=0, version=1) == v1
    # version 1, 4-byte samples: 0/1/2/4-byte variant; values 0, 5, 300, 70000 -> zig-zag of differences 0, 10, 590, 139400
    # codes (0, 1, 2, 3) -> 0xE4; data 0a | 4e 02 | 88 20 02 00
    v4 = bytes([16, 0, 0, 0, 0xE4, 0x0A, 0x4E, 0x02, 0x88, 0x20, 0x02, 0x00])
    assert np.frombuffer(hdf5_lite.vbz_decode(v4, [1, 4, 1, 0]), dtype="<i4").tolist() == [0, 5, 300, 70000]
    with pytest.raises(hdf5_lite.Hdf5Error):
        hdf5_lite.vbz_decode(v0[:-2], [0, 2, 1, 0])          # truncated data


def test_new_style_typed_encoder_config():
    """crf/model.py:231-232: a config whose [encoder] has a 'type' goes through from_dict; the typed form of the
    rnn_encoder stack builds the same model (same state-dict keys and shapes), anything else is rejected."""
    from xna_basecaller_amd.crf.model import Model
    F = 96
    conv = lambda i, o, w, s, p: dict(type="convolution", insize=i, size=o, bias=True, winlen=w, stride=s, padding=p, activation="swish")
    subs = [conv(1, 4, 5, 1, 2), conv(4, 16, 5, 1, 2), conv(16, F, 19, 5, 9), dict(type="permute", dims=[2, 0, 1])]
    subs += [dict(type="lstm", size=F, insize=F, bias=True, reverse=(i % 2 == 0)) for i in range(5)]
    subs += [dict(type="linearcrfencoder", insize=F, n_base=6, state_len=3, bias=True, scale=5.0, activation="tanh", blank_score=2.0)]
    cfg = make_config(F)
    old = Model(cfg)
    cfg2 = make_config(F)
    cfg2["encoder"] = dict(type="serial", sublayers=subs)
    new = Model(cfg2)
    assert [(k, tuple(v.shape)) for k, v in new.state_dict().items()] == [(k, tuple(v.shape)) for k, v in old.state_dict().items()]
    assert new.stride == 5 and new._features == F and new.encoder[-1].blank_score == 2.0 and new.encoder[-1].expand_blanks
    # a TOML round trip of the typed form (array of tables)
    text = toml_lite.dumps(cfg2)
    assert toml_lite.loads(text)["encoder"]["sublayers"][8]["reverse"] is True
    bad = [dict(d) for d in subs]
    bad[5]["reverse"] = True                                   # wrong direction pattern
    cfg3 = make_config(F)
    cfg3["encoder"] = dict(type="serial", sublayers=bad)
    with pytest.raises(NotImplementedError):
        Model(cfg3)


def test_load_model_precedence_and_checkpoint_choice(tmp_path):
    """util.load_model (util.py:261-366): newest weights_N.tar unless one is named, flag > [basecaller] table >
    4000/500/64 (overlap 0 is a value, chunksize 0 is 'unset'), skip_top leaves encoder.9 at its initial values,
    and a checkpoint that does not fit fails loudly.  No GPU: the device context is created lazily."""
    import torch
    from xna_basecaller_amd.crf.model import Model
    cfg = make_config(32)
    cfg["basecaller"] = {"chunksize": 3600, "batchsize": 384}
    d = str(tmp_path / "m")
    os.makedirs(d)
    open(os.path.join(d, "config.toml"), "w").write(toml_lite.dumps(cfg))
    ref = Model(make_config(32))
    g = torch.Generator().manual_seed(5)
    new = {k: torch.randn(v.shape, generator=g) for k, v in ref.state_dict().items()}
    torch.save({k: torch.zeros_like(v) for k, v in new.items()}, os.path.join(d, "weights_2.tar"))
    torch.save({"module." + k: v for k, v in new.items()}, os.path.join(d, "weights_10.tar"))   # 10 > 2 numerically
    with pytest.raises(FileNotFoundError):
        util.load_model(str(tmp_path), "cpu")

    m = util.load_model(d, "cuda:0")
    assert all(torch.equal(v, new[k]) for k, v in m.state_dict().items())
    assert m.config["basecaller"] == {"chunksize": 3600, "batchsize": 384, "overlap": 500, "quantize": False}
    assert m.config["encoder"]["drop_rate"] == 0 and m.config["encoder"]["drop_rate_bottom"] == 0
    m = util.load_model(d, "cuda:0", weights=2, chunksize=0, overlap=0, batchsize=16, quantize=None)
    assert all(float(v.abs().max()) == 0 for v in m.state_dict().values())
    assert m.config["basecaller"] == {"chunksize": 3600, "batchsize": 16, "overlap": 0, "quantize": None}

    torch.save(new, os.path.join(d, "weights_9.tar"))      # skip_top filters by the model's key names (no prefix)
    m = util.load_model(d, "cuda:0", weights=9, skip_top=True, use_koi=True)
    init = Model(make_config(32)).state_dict()
    for k, v in m.state_dict().items():
        if k.startswith("encoder.9"):
            assert not torch.equal(v, new[k]) and v.shape == init[k].shape
        else:
            assert torch.equal(v, new[k])

    assert m.encoder[-1].expand_blanks                      # XNA alphabet: use_koi is dropped, Viterbi (util.py:299-301)

    # 4-base model: use_koi (the CLI's default) selects the beam-search decoder, --no-use-koi the Viterbi one
    d4 = str(tmp_path / "m4")
    os.makedirs(d4)
    cfg4 = make_config(32, labels=("N", "A", "C", "G", "T"))
    open(os.path.join(d4, "config.toml"), "w").write(toml_lite.dumps(cfg4))
    torch.save(Model(cfg4).state_dict(), os.path.join(d4, "weights_1.tar"))
    assert not util.load_model(d4, "cuda:0", use_koi=True).encoder[-1].expand_blanks
    assert util.load_model(d4, "cuda:0", use_koi=False).encoder[-1].expand_blanks

    wide = Model(make_config(64)).state_dict()
    torch.save(wide, os.path.join(d, "weights_11.tar"))
    with pytest.raises((AssertionError, RuntimeError)):
        util.load_model(d, "cuda:0")


def test_column_to_set(tmp_path):
    f = tmp_path / "ids.tsv"
    f.write_text("read_id\tx\nr1\t5\nr2 6\n\tr3\t7\n")
    assert util.column_to_set(str(f), skip_header=True) == {"r1", "r2", "r3"}
    assert util.column_to_set(str(f), idx=1) == {"x", "5", "6", "7"}
    assert util.column_to_set(str(tmp_path / "absent")) is None and util.column_to_set(None) is None


def test_compute_transition_probs_layout():
    """CTC_CRF.compute_transition_probs (crf/model.py:62-76): from the (new_state, dropped_base) edge layout to
    (old_state, emitted_base), softmax over a source state's nb + 1 options -- checked edge by edge with plain loops."""
    from xna_basecaller_amd.crf.model import CTC_CRF
    nb, sl = 4, 3
    crf = CTC_CRF(sl, "NACGT")
    S, hi = nb ** sl, nb ** (sl - 1)
    rng = np.random.default_rng(4)
    T, N = 3, 2
    scores = rng.standard_normal((T, N, S * (nb + 1))).astype(np.float32)
    betas = rng.standard_normal((T + 1, N, S)).astype(np.float32)
    tp, ip = crf.compute_transition_probs(scores, betas)
    assert tp.shape == (T, N, S, nb + 1) and ip.shape == (N, S)
    M = scores.reshape(T, N, S, nb + 1)
    idx = crf.idx.numpy()
    for t in range(T):
        for n in range(N):
            for src in range(0, S, 7):
                w = np.empty(nb + 1)
                w[0] = M[t, n, src, 0] + betas[t + 1, n, src]                      # stay
                for b in range(nb):                                                  # emit base b: destination j
                    j, k = (src % hi) * nb + b, src // hi + 1
                    assert idx[j, k] == src
                    w[1 + b] = M[t, n, j, k] + betas[t + 1, n, j]
                w = np.exp(w - w.max())
                assert np.allclose(tp[t, n, src], w / w.sum(), atol=1e-6)
    e = np.exp(betas[0] - betas[0].max(-1, keepdims=True))
    assert np.allclose(ip, e / e.sum(-1, keepdims=True), atol=1e-6)


def test_eval_loop_handoff_fastq_to_paf_tooling():
    """SURVEY.md 8 f2 (eval_model.sh:119-177): the FASTQ this package writes is what the reference's evaluation chain
    consumes -- minimap2 names a query by the FASTQ header up to the first whitespace, and src/tools/analyze_paf.py joins
    the PAF's read_id column with the FASTQ's record ids (SeqIO.index(...)[read_id]).  tests/golden/evalloop.json holds
    the reference's own read_paf view of a minimap2-shaped PAF for six reads (made by make_evalloop_golden.py in the
    build container); here: the writer still produces the same bytes, and a strict FASTQ parse (Biopython's rules: '@'
    title, sequence, '+' line, quality of equal length in Sanger phred+33) yields the ids, lengths and sequences the
    reference's PAF reader reports."""
    sys_path_golden = os.path.join(GOLDEN)
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_evalloop_golden", os.path.join(sys_path_golden, "make_evalloop_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    fix = json.load(open(os.path.join(GOLDEN, "evalloop.json")))
    fastq, records = gen.synthetic_fastq()
    assert fastq == fix["fastq"] and [list(r) for r in records] == fix["records"]
    assert gen.paf_text(records) == fix["paf"]
    # strict FASTQ parse
    lines = fastq.split("\n")
    assert lines[-1] == "" and (len(lines) - 1) % 4 == 0
    recs = {}
    for i in range(0, len(lines) - 1, 4):
        title, seq, plus, qual = lines[i:i + 4]
        assert title.startswith("@") and plus == "+" and len(seq) == len(qual) > 0
        assert all(33 <= ord(c) <= 126 for c in qual) and set(seq) <= set("ACGTXY")
        rid, _, desc = title[1:].partition(" ")
        tags = dict(t.split(":", 2)[::2] for t in desc.split("\t"))
        assert list(t.split(":")[0] for t in desc.split("\t")) == ["RG", "qs", "mx", "ch", "st", "rn", "f5"]
        assert tags["qs"] == "40" and tags["RG"].endswith("_xna_r9.4.1_e8_sup@v3.3")
        recs[rid] = (seq, [ord(c) - 33 for c in qual])
    called = {rid: s for rid, s in records if s}
    assert {k: v[0] for k, v in recs.items()} == called             # the empty call is not in the FASTQ
    # the reference's PAF reader: every aligned read joins on the FASTQ id, with the FASTQ's length
    ref = fix["reference_read_paf"]
    col = {c: i for i, c in enumerate(ref["columns"])}
    assert ref["columns"][:12] == ["read_id", "read_length", "read_start", "read_end", "strand", "target_id", "target_length",
                                   "target_start", "target_end", "n_matches", "block_length", "mapping_quality"]
    assert len(ref["rows"]) == 4
    for row in ref["rows"]:
        rid = row[col["read_id"]]
        assert rid in recs and row[col["read_length"]] == len(recs[rid][0])
        assert all(q == 46 for q in recs[rid][1])                    # 'O': the Viterbi branch's constant quality (basecall.py:68)
        assert row[col["cs"]].startswith(":10*ag:") and row[col["strand"]] in "+-"


def test_bench_traffic_is_the_committed_bytes_per_step_over_this_runs_launches(tmp_path, monkeypatch):
    """bench.py's roofline.traffic is a REPLAY of committed PMC passes (profiles/*_pmc_hbm_traffic.json), not a count taken in the
    run: it is quoted only for the configuration, the call pairing AND the code the passes were collected on (the
    library's source digest), with its provenance beside it; otherwise null with the reason.  When quoted: the passes' bytes
    per bench step over THIS run's launches per step."""
    import importlib.util
    from conftest import ROOT
    from xna_basecaller_amd import _lib
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    digest = _lib.source_digest()
    assert len(digest) == 12 and digest == _lib.source_digest()
    prof = {"steps": 2, "source_digest": digest, "config": {"n_base": 6, "batch_per_gpu": 512, "chunksize": 10000, "precision": "mixed", "fuse": 1},
            "kernels": {"lstm_kernel<48, 2, true, true>": {"hbm_bytes_all_launches": 3.4e11}}}
    os.makedirs(tmp_path / "profiles")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    path = tmp_path / "profiles" / "r99_pmc_hbm_traffic.json"
    path.write_text(json.dumps(prof))
    got, src = bench.measured_traffic("lstm_kernel", 6, 512, 10000, "mixed", 2.5, True)
    assert got == pytest.approx(3.4e11 / 2 / 2.5)
    assert src["file"] == os.path.join("profiles", "r99_pmc_hbm_traffic.json") and src["profile_source_digest"] == digest
    # another batch size, another pairing, other code: null, and the reason says why
    for args, word in (((6, 513, 10000, "mixed", 2.5, True), "no profile"), ((6, 512, 10000, "mixed", 5.0, False), "pairing")):
        got, src = bench.measured_traffic("lstm_kernel", *args)
        assert got is None and word in src["reason"]
    prof["source_digest"] = "0" * 12
    path.write_text(json.dumps(prof))
    got, src = bench.measured_traffic("lstm_kernel", 6, 512, 10000, "mixed", 2.5, True)
    assert got is None and "other code" in src["reason"] and src["running_source_digest"] == digest


class _FakePipelineModel:
    """Host-logic stand-in for Model in compute_sequences_pipelined: records the submit / collect order, enforces the C ABI's
    slot rule (a slot is collected before it is submitted again) and returns each batch's own first column as its 'sequence'."""

    class _Enc:
        expand_blanks = True

    def __init__(self, depth, max_batch=8):
        self.encoder = [self._Enc()]
        self.depth, self.max_batch = depth, max_batch
        self.geometry = None
        self.busy = {}
        self.log = []
        self.rebuilds = 0

    def context_is_current(self, chunk_len, batch):
        return self.geometry == chunk_len and batch <= self.max_batch

    def pipeline_depth(self, chunk_len, n):
        if not self.context_is_current(chunk_len, n):
            assert not self.busy, "context rebuilt with batches in flight"
            self.geometry = chunk_len
            self.rebuilds += 1
        return self.depth

    def submit_chunks(self, slot, batch):
        assert self.context_is_current(batch.shape[-1], batch.shape[0])
        assert 0 <= slot < self.depth and slot not in self.busy, "slot %d submitted twice" % slot
        self.busy[slot] = np.asarray(batch)[:, 0, :1].astype(np.int8).copy()
        self.log.append(("submit", slot))
        return ("ctx", slot, batch.shape[0])

    def collect_chunks(self, handle):
        _, slot, n = handle
        self.log.append(("collect", slot))
        seq = self.busy.pop(slot)
        return seq, np.ones(n, np.int32)


@pytest.mark.parametrize("depth", [2, 4])
def test_device_stage_keeps_depth_batches_in_flight_in_order(depth):
    """crf/basecall.py:109-111 (compute_scores stage): results in input order; `depth` batches rotate through `depth` staging
    slots (4 where the context co-schedules two calls per pass: pair k+1 is submitted before pair k is collected); a change of
    geometry drains everything first."""
    from xna_basecaller_amd.crf.basecall import compute_sequences_pipelined
    model = _FakePipelineModel(depth)
    sizes = [8, 8, 3, 8, 8, 8, 1] + [8] * 4
    lens = [100] * 7 + [60] * 4                                   # the last four batches need another context
    batches = [("k%d" % i, np.full((n, 1, L), i, np.float32)) for i, (n, L) in enumerate(zip(sizes, lens))]
    out = list(compute_sequences_pipelined(model, iter(batches)))
    assert [k for k, _ in out] == [k for k, _ in batches]
    for i, (_, seq) in enumerate(out):
        assert seq.shape == (sizes[i], 1) and np.all(seq == i)
    assert not model.busy and model.rebuilds == 2
    # steady state: `depth` submits before the first collect, then they alternate; a collect always takes the oldest batch
    first = model.log[:depth + 1]
    assert first == [("submit", s) for s in range(depth)] + [("collect", 0)]
    inflight, peak = 0, 0
    for op, _ in model.log:
        inflight += 1 if op == "submit" else -1
        peak = max(peak, inflight)
    assert peak == depth


def test_sam_text_output_matches_the_reference_functions(tmp_path):
    """SURVEY 8 f3 (second half): `> calls.sam`.  tests/golden/sam.json holds what the reference's own sam_header / sam_record
    (unaligned branch) / Read.readgroup / Read.tagdata returned for four reads (make_sam_golden.py, build container): this
    package's functions return the same strings, and Writer(mode 'w') writes header + one line per record."""
    import io as pyio
    import types
    from xna_basecaller_amd import io as xio
    from xna_basecaller_amd import reads as xreads
    g = json.load(open(os.path.join(GOLDEN, "sam.json")))
    model = g["model"]
    rds = []
    for rec in g["records"]:
        r = types.SimpleNamespace(**rec["read"])
        r.signal = np.zeros(100 + len(rec["sequence"]), np.float32)
        r.start = r.duration = r.template_start = r.template_duration = 0.0
        r.tagdata = (lambda r=r: xreads._read_tags(r))
        rds.append(r)
        assert r.tagdata() == rec["tags"][2:]
        assert xio.sam_record(r.read_id, rec["sequence"], rec["qstring"], False, tags=rec["tags"]) == rec["sam_record"]
        assert xio.sam_record(r.read_id, rec["sequence"], rec["qstring"], None) == rec["sam_record_no_tags"]
    groups = sorted({xreads._read_group(r, model) for r in rds})
    assert groups == g["groups"]
    header = xio.sam_header(groups, version=g["bonito_version"], argv=g["argv"], aligner_version=g["mappy_version"])
    assert header == g["header"]
    # without an aligner (the only case here) the aligner's @PG line is not claimed; everything else is unchanged
    plain = xio.sam_header(groups, version=g["bonito_version"], argv=g["argv"])
    assert plain == "".join(l for l in g["header"].splitlines(True) if not l.startswith("@PG\tID:aligner"))
    # aligned records (io.py:118-137), round 5: the reference's sam_record driven with a stand-in mapping object, both
    # strands, soft clips at neither / one / both ends; mappy.revcomp restated (X stays X, the IUPAC code Y becomes R)
    assert len(g["aligned"]) >= 5 and {a["mapping"]["strand"] for a in g["aligned"]} == {1, -1}
    for a in g["aligned"]:
        m = types.SimpleNamespace(**a["mapping"])
        assert xio.sam_record(a["read_id"], a["sequence"], a["qstring"], m, tags=a["tags"]) == a["sam_record"]
        assert xio.sam_record(a["read_id"], a["sequence"], a["qstring"], m) == a["sam_record_no_tags"]
        f = a["sam_record"].split("\t")
        assert f[1] == ("0" if m.strand == 1 else "16") and int(f[3]) == m.r_st + 1 and len(f[9]) == len(f[10])
    assert xio.revcomp("ACGTXYacgtN") == "NacgtRXACGT" and xio.revcomp("") == ""
    # ... and the writer emits that record for a result that carries a mapping (what an aligner stage would hand it)
    a0 = g["aligned"][2]
    rd0 = types.SimpleNamespace(**g["records"][0]["read"])
    rd0.signal = np.zeros(50, np.float32)
    rd0.start = rd0.duration = rd0.template_start = rd0.template_duration = 0.0
    rd0.read_id = a0["read_id"]
    rd0.tagdata = lambda: []
    buf = pyio.StringIO()
    wa = xio.Writer("w", iter([(rd0, {"sequence": a0["sequence"], "qstring": a0["qstring"], "mean_qscore": 11.0,
                                      "mapping": types.SimpleNamespace(**a0["mapping"])})]),
                    fd=buf, group_key=model, groups=set(), summary=str(tmp_path / "a_summary.tsv"))
    wa.run()
    line = buf.getvalue()[len(xio.sam_header([])):].rstrip("\n")
    assert line.split("\t")[:11] == a0["sam_record_no_tags"].split("\t")[:11] and "\tNM:i:1\tMD:Z:3^C5\t" in line + "\t"
    # the writer: header first, then the records in order, an empty call skipped, summary rows as for FASTQ
    out = pyio.StringIO()
    results = [(r, {"sequence": rec["sequence"], "qstring": rec["qstring"], "mean_qscore": float(rec["tags"][1].split(":")[2])})
               for r, rec in zip(rds, g["records"])]
    results.insert(2, (rds[0], {"sequence": "", "qstring": ""}))
    w = xio.Writer("w", iter(results), fd=out, group_key=model, groups=set(groups), summary=str(tmp_path / "s_summary.tsv"))
    w.run()
    text = out.getvalue()
    assert text.startswith(xio.sam_header(groups))
    body = text[len(xio.sam_header(groups)):].splitlines()
    assert body == [rec["sam_record"] for rec in g["records"]]
    for line in body:                                    # SAM: 11 mandatory columns, flag 4, unmapped placeholders
        f = line.split("\t")
        assert f[1] == "4" and f[2] == "*" and f[5] == "*" and len(f[9]) == len(f[10]) and f[11] == "NM:i:0"
    assert len(open(tmp_path / "s_summary.tsv").read().splitlines()) == 1 + len(g["records"])
    assert [rid for rid, _ in w.log] == [r.read_id for r in rds]
    for mode in ("wb", "wc"):
        with pytest.raises(NotImplementedError):
            xio.Writer(mode, iter([]))


def test_read_groups_from_bundles_without_touching_signals(tmp_path):
    """cli/basecaller.py:100-106 / fast5.py:236-251: the @RG lines of the selected reads, metadata only."""
    from xna_basecaller_amd import reads as xreads
    recs = []
    for i in range(5):
        recs.append((np.zeros(10, np.int16), dict(read_id="r%d" % i, run_id="run%d" % (i % 2), range=1400.0, digitisation=8192.0,
                                                  offset=3, sampling_rate=4000.0, sample_id="lib", flow_cell_id="FC%d" % (i % 2),
                                                  device_id="MN1", exp_start_time="2021-03-04T05:06:07Z")))
    xreads.write_bundle(str(tmp_path / "a.xsig.npz"), recs)
    groups = xreads.get_read_groups(str(tmp_path), "m@v1")
    assert groups == {"@RG\tID:run%d_m@v1\tPL:ONT\tDT:2021-03-04T05:06:07\tPU:FC%d\tPM:MN1\tLB:lib\tSM:lib\tDS:run_id=run%d basecall_model=m@v1"
                      % (k, k, k) for k in (0, 1)}
    assert xreads.get_read_groups(str(tmp_path), "m@v1", read_ids={"r0", "r2"}) == {g for g in groups if "run0" in g}


def test_eval_loop_accuracy_table_from_this_packages_fastq(tmp_path):
    """SURVEY 8 f2, the accuracy half (eval_model.sh:155-177 -> src/tools/analyze_paf.py).  tests/golden/evalacc.json holds
    what the REFERENCE's compute_all_error_rates_paf / compute_stats_error_rate computed (build container) from (a) FASTQ text
    written by this package's Writer for 33 designed calls on three POC templates and (b) the PAF minimap2 reports for them
    (known alignments: UB called right / as a natural base / as the wrong UB / deleted / put beside its gap, both strands).
    Here: the Writer still produces that FASTQ byte for byte, a strict FASTQ parse feeds the restated metrics
    (tests/evalloop_metrics.py), and every number of the reference's table is reproduced -- per read, per template position,
    and the UB / DNA accuracy cuts."""
    import io
    import evalloop_metrics as em
    from xna_basecaller_amd import io as xio
    from xna_basecaller_amd.reads import SyntheticRead
    g = json.load(open(os.path.join(GOLDEN, "evalacc.json")))
    out = io.StringIO()
    results = []
    for i, (rid, _, seq) in enumerate(g["reads"]):
        r = SyntheticRead(rid, np.zeros(10 * len(seq), np.float32), run_id="evalrun", filename="poc.xsig.npz", channel=str(1 + i),
                          mux=1 + i % 4, read_number=i)
        results.append((r, {"sequence": seq, "qstring": "O" * len(seq), "mean_qscore": 40.0}))
    xio.Writer("wfq", iter(results), fd=out, group_key="xna_r9.4.1_e8_sup@v3.3", summary=str(tmp_path / "x_summary.tsv")).run()
    assert out.getvalue() == g["fastq"]
    recs = out.getvalue().strip().split("\n")
    assert len(recs) % 4 == 0 and all(l == "+" for l in recs[2::4]) and all(h.startswith("@") for h in recs[0::4])
    seqs = {h[1:].split()[0]: s for h, s in zip(recs[0::4], recs[1::4])}
    rows = em.parse_paf(g["paf"])
    assert [r["read_id"] for r in rows] == [r["read_id"] for r in g["per_read"]]
    by_group = {}
    for row, want in zip(rows, g["per_read"]):
        wrong, m = em.read_metrics(row, g["templates"][row["target_id"]], seqs[row["read_id"]])
        by_group.setdefault("%s/%s" % (row["target_id"], row["strand"]), []).append(wrong)
        for k, v in want.items():
            if k in ("read_id", "target_id", "strand"):
                continue
            if v is None:
                assert np.isnan(m[k]), (row["read_id"], k)
            elif isinstance(v, str):
                assert m[k] == v, (row["read_id"], k, m[k], v)
            else:
                assert abs(m[k] - v) < 1e-12, (row["read_id"], k, m[k], v)
    assert sorted(by_group) == sorted(g["error_rate_per_position"])
    for key, errs in by_group.items():
        rate = np.mean(errs, axis=0) * 100
        assert np.allclose(rate, g["error_rate_per_position"][key], rtol=0, atol=1e-12)
        tid, strand = key.split("/")
        t = g["templates"][tid]
        ubs = [i for i, c in enumerate(t) if c == "N"]
        if strand == "-":
            ubs = [len(t) - p - 1 for p in ubs[::-1]]
        cuts = em.error_rate_cuts(rate, ubs, max_dist=4)
        assert sorted(cuts) == sorted(g["error_rate_cuts"][key])
        for name, vals in g["error_rate_cuts"][key].items():
            assert np.allclose(cuts[name], vals, rtol=0, atol=1e-12), (key, name)
    # the headline numbers (README.md:139-143 reports them per model): UB accuracy = 100 - mean error at the UB, DNA = elsewhere
    ub_acc = {k: 100 - np.mean(v["only_ub"]) for k, v in g["error_rate_cuts"].items()}
    assert abs(ub_acc["XNA01/+"] - 100 * 4 / 7) < 1e-9 and abs(ub_acc["XNA01/-"] - 75.0) < 1e-9


def test_default_precision_is_mixed_and_the_opt_ins_are_named(monkeypatch):
    """the default arithmetic of Model (hence of the CLI) is `mixed` -- feed-forward projections in three fp16
    products, recurrences in f16f8 -- and the faster / more exact forms are opt-ins by config.toml [basecaller] precision or the
    XNA_PRECISION environment variable."""
    from conftest import make_config
    from xna_basecaller_amd import _lib
    from xna_basecaller_amd.crf import Model
    monkeypatch.delenv("XNA_PRECISION", raising=False)
    assert Model(make_config(32)).precision == _lib.XB_PREC_MIXED == 4
    cfg = make_config(32)
    cfg["basecaller"] = dict(cfg.get("basecaller", {}), precision="f16f8")
    assert Model(cfg).precision == _lib.XB_PREC_F16F8
    monkeypatch.setenv("XNA_PRECISION", "f16x3")
    assert Model(cfg).precision == _lib.XB_PREC_F16X3
    assert set(_lib.PRECISIONS) == {"mixed", "f16x3", "f16", "f16f8", "f16f8i"}
    monkeypatch.setenv("XNA_PRECISION", "fp64")
    with pytest.raises(KeyError):
        Model(cfg)
