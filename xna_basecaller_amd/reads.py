"""
Read objects and signal preparation (ub-bonito/bonito/fast5.py).

  Read (metadata + scaled, trimmed, normalised fp32 signal)  fast5.py:22-128
  trim                                                        fast5.py:149-171
  med_mad                                                     fast5.py:174-180
  norm_by_noisiest_section                                    fast5.py:183-204
  get_reads                                                   fast5.py:284-296

Two containers are read: `*.fast5` (HDF5 + ONT's VBZ filter; neither libhdf5/h5py nor ont_fast5_api exists in this
image, so the files are parsed by the package's own minimal reader, hdf5_lite.py) and a pre-extracted signal bundle
(`*.xsig.npz`, written by `write_bundle`): raw int16 DACs plus the attributes fast5.py reads (channel_id,
tracking_id, Raw attrs).  Everything after the container decode is the reference's arithmetic.
"""
import json
import os
from datetime import datetime, timedelta
from glob import glob
from pathlib import Path

import numpy as np

__all__ = ["Read", "trim", "med_mad", "norm_by_noisiest_section", "get_reads", "ReadLoader", "read_jobs",
           "write_bundle", "SyntheticRead"]


def med_mad(x, factor=1.4826):
    """Robust location / scale: the median and factor * median(|x - median|), plus float32 eps so the scale is never 0
    (fast5.py:174-180)."""
    centre = np.median(x)
    spread = np.median(np.absolute(x - centre)) * factor + np.finfo(np.float32).eps
    return centre, spread


def trim(signal, window_size=40, threshold_factor=2.4, min_elements=3):
    """
    Where the open-pore / adapter prefix ends (fast5.py:149-171), evaluated on whole windows at once.
    The first 10 samples are ignored; the threshold is median + 2.4 * mad of the last 100 windows of the remainder.
    The prefix ends with the first window, from the first "hot" window on (more than `min_elements` samples above the
    threshold), whose last sample is back at or under the threshold.  Returns (start, length of the remainder); start
    is relative to the untrimmed input, capped at the remainder's length, and 10 when no such window exists.
    """
    skip = 10
    body = signal[skip:]
    level, scale = med_mad(body[-(window_size * 100):])
    threshold = level + scale * threshold_factor
    n_win = len(body) // window_size
    if n_win:
        windows = body[:n_win * window_size].reshape(n_win, window_size)
        hot = np.count_nonzero(windows > threshold, axis=1) > min_elements
        if hot.any():
            first = int(np.argmax(hot))
            calm = windows[first:, -1] <= threshold
            if calm.any():
                end = (first + int(np.argmax(calm)) + 1) * window_size
                return min(end + skip, len(body)), len(body)
    return skip, len(body)


def norm_by_noisiest_section(signal, samples=100, threshold=6.0):
    """
    Normalisation of short reads (fast5.py:183-204): med/mad of the widest stretch of 100-sample windows whose
    standard deviation exceeds std(signal) / threshold; of the whole signal when there is no such stretch.
    The stretch is located as the reference does, with scipy's plateau peaks on the 0/1 window mask (its
    left_bases / right_bases define the slice), so the numbers agree bit for bit (tests/golden/signal_prep.npz).
    """
    from scipy.signal import find_peaks
    cutoff = signal.std() / threshold
    n_win = signal.shape[0] // samples
    mask = np.ones(signal.shape)
    if n_win:
        noisy = signal[:n_win * samples].reshape(n_win, samples).std(axis=1) > cutoff
        mask[:n_win * samples] = np.repeat(noisy, samples)
    mask[0] = mask[-1] = 0
    peaks, info = find_peaks(mask, width=(None, None))
    section = signal
    if len(peaks):
        widest = np.argmax(info["widths"])
        section = signal[info["left_bases"][widest]: info["right_bases"][widest]]
    centre, spread = med_mad(section)
    return (signal - centre) / spread


def _parse_time(s):
    s = s.replace("Z", "")
    for fmt in ("%Y-%m-%dT%H:%M:%S.%f", "%Y-%m-%dT%H:%M:%S", "%Y-%m-%d %H:%M:%S"):
        try:
            return datetime.strptime(s, fmt)
        except ValueError:
            pass
    try:
        from dateutil import parser
        return parser.parse(s)
    except Exception:
        return datetime(1970, 1, 1)


def _read_tags(read):
    return ["mx:i:%s" % read.mux, "ch:i:%s" % read.channel, "st:Z:%s" % read.start_time,
            "rn:i:%s" % read.read_number, "f5:Z:%s" % read.filename]


def _read_group(read, model):
    """The read's @RG header line for SAM output (fast5.py:106-120): ID <run_id>_<model>, platform ONT, run start, flow cell,
    device, sample (as library and as sample), and a description naming run and basecall model."""
    fields = (("ID", "%s_%s" % (read.run_id, model)), ("PL", "ONT"), ("DT", getattr(read, "exp_start_time", "")),
              ("PU", getattr(read, "flow_cell_id", "")), ("PM", getattr(read, "device_id", "None")),
              ("LB", getattr(read, "sample_id", "None")), ("SM", getattr(read, "sample_id", "None")),
              ("DS", "run_id=%s basecall_model=%s" % (read.run_id, model)))
    return "\t".join(["@RG"] + ["%s:%s" % kv for kv in fields])


class Read:
    """
    One nanopore read.  `raw` int16 DACs + attributes -> scaled pA -> trimmed -> normalised
    float32 `signal` (fast5.py:88-100).
    """

    def __init__(self, raw, attrs, filename, meta=False):
        self.meta = meta
        self.read_id = attrs["read_id"]
        self.filename = os.path.basename(str(filename))
        self.run_id = attrs.get("run_id", "")
        self.sample_id = attrs.get("sample_id", "None")
        self.exp_start_time = attrs.get("exp_start_time", "1970-01-01T00:00:00").replace("Z", "")
        self.flow_cell_id = attrs.get("flow_cell_id", "")
        self.device_id = attrs.get("device_id", "None")

        self.range = attrs["range"]
        self.digitisation = attrs["digitisation"]
        self.offset = int(attrs["offset"])
        self.sampling_rate = attrs["sampling_rate"]
        self.scaling = attrs["range"] / attrs["digitisation"]

        self.mux = attrs.get("start_mux", 0)
        self.read_number = attrs.get("read_number", 0)
        self.channel = attrs.get("channel_number", "0")
        self.start = attrs.get("start_time", 0) / self.sampling_rate
        self.duration = attrs.get("duration", len(raw) if raw is not None else 0) / self.sampling_rate

        start_time = _parse_time(self.exp_start_time) + timedelta(seconds=self.start)
        self.start_time = start_time.replace(microsecond=0).isoformat()
        if self.meta:
            return

        scaled = np.array(self.scaling * (raw + self.offset), dtype=np.float32)
        trim_start, _ = trim(scaled[:8000])
        scaled = scaled[trim_start:]
        self.template_start = self.start + (1 / self.sampling_rate) * trim_start
        self.template_duration = self.duration - (1 / self.sampling_rate) * trim_start
        if len(scaled) > 8000:
            med, mad = med_mad(scaled)
            self.signal = (scaled - med) / mad
        else:
            self.signal = norm_by_noisiest_section(scaled)

    def __repr__(self):
        return "Read('%s')" % self.read_id

    def readgroup(self, model):
        return _read_group(self, model)

    def tagdata(self):
        """FASTQ header tags of the read (fast5.py:118-128): mux, channel, start time, read number, source file."""
        return _read_tags(self)


class SyntheticRead:
    """A Read-shaped object around an already normalised signal (tests, bench, synthetic shards)."""

    def __init__(self, read_id, signal, run_id="synthetic", filename="synthetic.xsig.npz", channel="1",
                 mux=1, start=0.0, sampling_rate=4000.0, read_number=0):
        self.read_id = read_id
        self.signal = np.ascontiguousarray(signal, dtype=np.float32)
        self.run_id = run_id
        self.filename = filename
        self.channel = channel
        self.mux = mux
        self.start = start
        self.duration = len(self.signal) / sampling_rate
        self.template_start = start
        self.template_duration = self.duration
        self.read_number = read_number
        self.start_time = "1970-01-01T00:00:00"

    def __repr__(self):
        return "SyntheticRead('%s')" % self.read_id

    def readgroup(self, model):
        return _read_group(self, model)

    def tagdata(self):
        return _read_tags(self)


def write_bundle(path, reads):
    """reads: iterable of (raw int16 array, attrs dict with at least read_id/range/digitisation/offset/sampling_rate)."""
    arrays, metas = {}, []
    for i, (raw, attrs) in enumerate(reads):
        arrays["raw_%d" % i] = np.asarray(raw, dtype=np.int16)
        metas.append(attrs)
    arrays["meta"] = np.frombuffer(json.dumps(metas).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrays)


def _bundle_index(filename):
    """[(read_id, position)] of a bundle without touching the signals."""
    with np.load(filename) as z:
        metas = json.loads(bytes(z["meta"]).decode())
    return [(attrs["read_id"], i) for i, attrs in enumerate(metas)]


# ---- fast5 (HDF5) -------------------------------------------------------------------------------------------------
def _fast5_layout(f):
    """'multi' (/read_<id>/Raw/Signal, ont_fast5_api multi-read files) or 'single' (/Raw/Reads/Read_<n>/Signal)."""
    keys = f.keys()
    if any(k.startswith("read_") for k in keys):
        return "multi"
    if "Raw" in keys and "UniqueGlobalKey" in keys:
        return "single"
    raise ValueError("%s: neither a multi-read nor a single-read fast5 file" % f.path)


def _fast5_index(filename):
    """[(read_id, group name)] of a fast5 file, in the file's own (name sorted) order."""
    from . import hdf5_lite
    with hdf5_lite.File(filename) as f:
        if _fast5_layout(f) == "multi":
            return [(k[len("read_"):], k) for k in f.keys() if k.startswith("read_")]
        reads = f["Raw/Reads"]
        return [(str(reads[k].attrs.get("read_id", k)), "Raw/Reads/" + k) for k in reads.keys()]


def _text(v, default=""):
    if v is None:
        return default
    if isinstance(v, (bytes, np.bytes_)):
        return v.decode("ascii", "replace")
    return str(v)


def _fast5_read(filename, group, meta=False):
    """One read of a fast5 file -> Read (meta: attributes only, the signal is not touched -- fast5.py:222-233).  The attribute
    set is the one fast5.py:24-76 consumes."""
    from . import hdf5_lite
    with hdf5_lite.File(filename) as f:
        if group.startswith("Raw/Reads/"):                   # single-read layout
            raw_grp = f[group]
            chan, track = f["UniqueGlobalKey/channel_id"].attrs, f["UniqueGlobalKey/tracking_id"].attrs
            run_id = track.get("run_id", "")
        else:
            node = f[group]
            raw_grp = node["Raw"]
            chan, track = node["channel_id"].attrs, node["tracking_id"].attrs
            run_id = node.attrs.get("run_id", track.get("run_id", ""))
        ra = raw_grp.attrs
        attrs = {
            "read_id": _text(ra.get("read_id", group.split("_", 1)[-1])),
            "run_id": _text(run_id), "sample_id": _text(track.get("sample_id"), "None"),
            "exp_start_time": _text(track.get("exp_start_time"), "1970-01-01T00:00:00"),
            "flow_cell_id": _text(track.get("flow_cell_id")), "device_id": _text(track.get("device_id"), "None"),
            "range": float(chan["range"]), "digitisation": float(chan["digitisation"]), "offset": int(chan["offset"]),
            "sampling_rate": float(chan["sampling_rate"]), "channel_number": _text(chan.get("channel_number"), "0"),
            "start_mux": int(ra.get("start_mux", 0)), "read_number": int(ra.get("read_number", 0)),
            "start_time": int(ra.get("start_time", 0)),
        }
        raw = None if meta else raw_grp["Signal"][:]
        attrs["duration"] = int(ra.get("duration", 0 if raw is None else len(raw)))
        return Read(raw, attrs, filename, meta=meta)


def _load_read(job, meta=False):
    """(filename, key) -> Read; runs in a pool worker (fast5.py:263-270 get_raw_data_for_read)."""
    filename, key = job
    if str(filename).endswith(".fast5"):
        return _fast5_read(filename, key, meta=meta)
    with np.load(filename) as z:
        attrs = json.loads(bytes(z["meta"]).decode())[key]
        return Read(None if meta else z["raw_%d" % key], attrs, filename, meta=meta)


def get_read_groups(directory, model, read_ids=None, skip=False, n_proc=1, recursive=False, cancel=None):
    """The set of @RG header lines of the selected reads (fast5.py:236-251): metadata only, no signal is read."""
    groups = set()
    for job in read_jobs(directory, read_ids=read_ids, skip=skip, recursive=recursive):
        groups.add(_load_read(job, meta=True).readgroup(model))
        if cancel is not None and cancel.is_set():
            break
    return groups


def read_jobs(directory, read_ids=None, skip=False, recursive=False):
    """(filename, key) of every selected read under `directory` -- `*.fast5` files (multi- or single-read HDF5, parsed
    by hdf5_lite) and `*.xsig.npz` signal bundles -- in file then in-file order; metadata only."""
    jobs = []
    for ext, index in ((".fast5", _fast5_index), (".xsig.npz", _bundle_index)):
        pattern = ("**/*" if recursive else "*") + ext
        for fn in sorted(Path(x) for x in glob(directory + "/" + pattern, recursive=True)):
            for rid, key in index(fn):
                if read_ids is None or (rid in read_ids) ^ skip:
                    jobs.append((fn, key))
    return jobs


class ReadLoader:
    """
    Iterator over the selected reads of a directory.  The worker pool is started in the constructor -- create it before
    the GPU is initialised (forking a process that holds a HIP context is best avoided).  Results arrive in job order,
    and the pool stays at most `lookahead` reads (default 4 per worker) ahead of the consumer: jobs are handed out one
    by one as results are taken, so a slow device stage bounds the prepared signals held in host memory (Pool.imap alone
    has no backpressure; the reference calls it once per file, fast5.py:284-296, which caps its backlog at one file).
    """

    def __init__(self, directory, read_ids=None, skip=False, n_proc=1, recursive=False, cancel=None, shard=None,
                 limit=0, lookahead=0):
        jobs = list(enumerate(read_jobs(directory, read_ids=read_ids, skip=skip, recursive=recursive)))
        if limit:
            jobs = jobs[:limit]
        self.total = len(jobs)                      # selected reads over all shards
        if shard is not None:
            rank, world = shard
            jobs = [(i, j) for i, j in jobs if i % world == rank]
        self.jobs, self.cancel, self.pool = jobs, cancel, None
        self.lookahead = max(1, int(lookahead)) if lookahead else 4 * max(1, n_proc)
        self.max_pending = 0                        # high-water mark of reads in flight or waiting (tests)
        if n_proc > 1 and len(jobs) > 1:
            import multiprocessing as mp
            self.pool = mp.get_context("fork" if "fork" in mp.get_all_start_methods() else None).Pool(min(n_proc, len(jobs)))

    def __len__(self):
        return len(self.jobs)

    def __iter__(self):
        from collections import deque
        pending, nxt = deque(), 0
        try:
            for k, (i, job) in enumerate(self.jobs):
                if self.pool is not None:
                    # top the window up, then take the oldest result: at most `lookahead` reads are ever outstanding
                    while nxt < len(self.jobs) and len(pending) < self.lookahead:
                        pending.append(self.pool.apply_async(_load_read, (self.jobs[nxt][1],)))
                        nxt += 1
                    self.max_pending = max(self.max_pending, len(pending))
                    read = pending.popleft().get()
                else:
                    read = _load_read(job)
                read.index = i
                yield read
                if self.cancel is not None and self.cancel.is_set():
                    return
        finally:
            self.close()

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool = None


def get_reads(directory, read_ids=None, skip=False, n_proc=1, recursive=False, cancel=None, shard=None, limit=0):
    """
    All selected reads under `directory` as `Read`s, in file then in-file order (fast5.py:284-296).  The signal
    preparation (inflate, scale, trim, two medians) runs in a pool of `n_proc` worker processes; results come back in
    order.  shard = (rank, world): only reads whose global index i has i % world == rank are ever loaded; every yielded
    read carries that index as `read.index`.
    """
    return ReadLoader(directory, read_ids=read_ids, skip=skip, n_proc=n_proc, recursive=recursive, cancel=cancel,
                      shard=shard, limit=limit)
