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
from collections import OrderedDict, deque
from functools import lru_cache
from datetime import datetime, timedelta
from glob import glob
from pathlib import Path

import numpy as np

__all__ = ["Read", "trim", "med_mad", "norm_by_noisiest_section", "get_reads", "ReadLoader", "read_jobs",
           "write_bundle", "SyntheticRead", "close_containers"]


def _median(x):
    """np.median of a 1-D float array without its Python-side dispatch (four medians per read are a third of the signal
    preparation on short reads): the same partition and the same `mean` of the middle element(s), NaN propagated as
    np.median does -- bit-equal results (tests/golden/signal_prep.npz, test_median_equals_numpy)."""
    n = x.shape[0]
    if x.ndim != 1 or n == 0:
        return np.median(x)
    k = n // 2
    part = np.partition(x, (k - 1, k, n - 1) if n % 2 == 0 else (k, n - 1))       # n - 1: a NaN, if any, ends up last
    if np.isnan(part[-1]):
        return np.median(x)
    return np.mean(part[k - 1:k + 1] if n % 2 == 0 else part[k:k + 1])


def med_mad(x, factor=1.4826):
    """Robust location / scale: the median and factor * median(|x - median|), plus float32 eps so the scale is never 0
    (fast5.py:174-180)."""
    centre = _median(x)
    spread = _median(np.absolute(x - centre)) * factor + np.finfo(np.float32).eps
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


def _widest_plateau(mask):
    """(left base, right base) of the widest run of ones in a 0/1 mask whose ends are 0, or None -- what scipy's
    `find_peaks(mask, width=(None, None))` reports as left_bases / right_bases of the peak with the largest width
    (fast5.py:196-200).  On such a mask every maximal run of ones [l, r] is one plateau peak; its prominence walk stops at
    the nearest zeros (bases l - 1 and r + 1), its width at half prominence is r - l + 1, and argmax takes the first
    of equally wide runs.  Checked against scipy itself in tests/test_host.py."""
    edges = np.diff(mask.astype(np.int8))
    left, right = np.flatnonzero(edges == 1), np.flatnonzero(edges == -1)      # l - 1 and r of every run
    if left.size == 0:
        return None
    widest = int(np.argmax(right - left))
    return int(left[widest]), int(right[widest]) + 1


def norm_by_noisiest_section(signal, samples=100, threshold=6.0):
    """
    Normalisation of short reads (fast5.py:183-204): med/mad of the widest stretch of 100-sample windows whose
    standard deviation exceeds std(signal) / threshold; of the whole signal when there is no such stretch.
    The reference locates the stretch with scipy's plateau peaks on the 0/1 window mask (left_bases / right_bases define
    the slice); `_widest_plateau` computes the same two indices directly, so the numbers agree bit for bit
    (tests/golden/signal_prep.npz).
    """
    cutoff = signal.std() / threshold
    n_win = signal.shape[0] // samples
    mask = np.ones(signal.shape[0], dtype=bool)
    if n_win:
        noisy = signal[:n_win * samples].reshape(n_win, samples).std(axis=1) > cutoff
        mask[:n_win * samples] = np.repeat(noisy, samples)
    mask[0] = mask[-1] = False
    section = signal
    bases = _widest_plateau(mask)
    if bases is not None:
        section = signal[bases[0]:bases[1]]
    centre, spread = med_mad(section)
    return (signal - centre) / spread


@lru_cache(maxsize=256)
def _parse_time(s):
    """exp_start_time -> datetime (one value per run: memoised, strptime costs 50 us per read otherwise)"""
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


# ---- per-process container cache -------------------------------------------------------------------------------------
# A reader worker serves many reads of the same container.  Each open container is indexed ONCE per process and kept (a few
# files, least recently used first out): a multi-read fast5 stays memory-mapped with its root symbol table enumerated, a
# bundle keeps its zip directory and its parsed metadata list.  Keyed by path, validated by (mtime, size) so that a file
# rewritten under the same name is re-opened.  fast5.py:254-296 opens the file per read as well, but through libhdf5, whose
# group lookup is a B-tree search; here the lookup is a dict access on the kept index (or the address carried by the job).
_CACHE_FILES = 8
_containers = OrderedDict()
_containers_pid = os.getpid()


def _container(filename, opener):
    global _containers, _containers_pid
    if _containers_pid != os.getpid():
        # forked: the parent's handles share their file offsets with it (a zip read from two processes at once corrupts
        # both) -- forget them without closing and open this process's own
        _containers, _containers_pid = OrderedDict(), os.getpid()
    key = str(filename)
    st = os.stat(key)
    sig = (st.st_mtime_ns, st.st_size)
    hit = _containers.get(key)
    if hit is not None and hit[0] == sig:
        _containers.move_to_end(key)
        return hit[1]
    if hit is not None:
        _close_container(_containers.pop(key)[1])
    obj = opener(key)
    _containers[key] = (sig, obj)
    while len(_containers) > _CACHE_FILES:
        _close_container(_containers.popitem(last=False)[1][1])
    return obj


def _close_container(obj):
    try:
        obj.close()
    except Exception:
        pass


def close_containers():
    """Drop every container this process keeps open (tests; a long-lived host that is done with a directory)."""
    while _containers:
        _close_container(_containers.popitem()[1][1])


class _Bundle:
    """An open `*.xsig.npz`: the zip directory and the metadata list are parsed once."""

    def __init__(self, path):
        self.z = np.load(path)
        self.metas = json.loads(bytes(self.z["meta"]).decode())

    def close(self):
        self.z.close()


def _open_bundle(filename):
    return _container(filename, _Bundle)


def _open_fast5(filename):
    from . import hdf5_lite
    return _container(filename, hdf5_lite.File)


def _bundle_index(filename):
    """[(read_id, position)] of a bundle without touching the signals."""
    return [(attrs["read_id"], i) for i, attrs in enumerate(_open_bundle(filename).metas)]


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
    f = _open_fast5(filename)
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
    set is the one fast5.py:24-76 consumes.  The file comes from the process's container cache: no per-read open / walk."""
    f = _open_fast5(filename)
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
    b = _open_bundle(filename)
    return Read(None if meta else b.z["raw_%d" % key], b.metas[key], filename, meta=meta)


def _load_run(jobs, meta=False):
    """A run of consecutive jobs (normally of one file) -> [Read]; the unit of work of the reader pool."""
    return [_load_read(job, meta=meta) for job in jobs]


def _read_groups_of(jobs, model):
    return {r.readgroup(model) for r in _load_run(jobs, meta=True)}


def _runs(jobs, run):
    """Consecutive jobs of one container, at most `run` at a time: [(first position, [job, ...])]."""
    out, k = [], 0
    while k < len(jobs):
        e = k + 1
        while e < len(jobs) and e - k < run and jobs[e][0] == jobs[k][0]:
            e += 1
        out.append((k, jobs[k:e]))
        k = e
    return out


def get_read_groups(directory, model, read_ids=None, skip=False, n_proc=1, recursive=False, cancel=None):
    """The set of @RG header lines of the selected reads (fast5.py:236-251): metadata only, no signal is read.  Jobs are
    taken file by file in runs, each container is opened and indexed once; with n_proc > 1 the runs go through a
    pool as in the reference."""
    groups = set()
    runs = [r for _, r in _runs(read_jobs(directory, read_ids=read_ids, skip=skip, recursive=recursive), 256)]
    if n_proc > 1 and len(runs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork" if "fork" in mp.get_all_start_methods() else None).Pool(min(n_proc, len(runs))) as pool:
            for part in pool.imap_unordered(_RunGroups(model), runs):
                groups |= part
                if cancel is not None and cancel.is_set():
                    break
        return groups
    for run in runs:
        groups |= _read_groups_of(run, model)
        if cancel is not None and cancel.is_set():
            break
    return groups


class _RunGroups:
    def __init__(self, model):
        self.model = model

    def __call__(self, jobs):
        return _read_groups_of(jobs, self.model)


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


def _warm_up():
    """One short and one long synthetic read through `Read.__init__` in the calling process.  Done right before a reader pool
    is forked: whatever the signal preparation initialises lazily (numpy's partition / std machinery, `_strptime`, allocator
    arenas) is then inherited by every worker instead of being paid by each -- measured: the first 4 000 short reads of a
    fresh process in 0.6 s instead of 1.7 s with 8 workers (tools/reader_bench.py)."""
    rng = np.random.default_rng(0)
    attrs = {"read_id": "warm-up", "range": 1437.0, "digitisation": 8192.0, "offset": 6, "sampling_rate": 4000.0}
    for n in (3000, 9000):
        Read((rng.standard_normal(n) * 60 + 480).astype(np.int16), attrs, "warm-up")


class ReadLoader:
    """
    Iterator over the selected reads of a directory.  The worker pool is started in the constructor -- create it before
    the GPU is initialised (forking a process that holds a HIP context is best avoided).  Results arrive in job order.
    The unit of work is a RUN of up to `run` consecutive reads of one container (a worker keeps the container open and
    indexed, `_container`; one task message per run instead of per read), and the pool stays at most `lookahead` reads
    (default 32 per worker) ahead of the consumer: runs are handed out as results are taken, so a slow device stage bounds
    the prepared signals held in host memory (Pool.imap alone has no backpressure; the reference calls it once per file,
    fast5.py:284-296, which caps its backlog at one file).
    """

    def __init__(self, directory, read_ids=None, skip=False, n_proc=1, recursive=False, cancel=None, shard=None,
                 limit=0, lookahead=0, run=0):
        jobs = list(enumerate(read_jobs(directory, read_ids=read_ids, skip=skip, recursive=recursive)))
        if limit:
            jobs = jobs[:limit]
        self.total = len(jobs)                      # selected reads over all shards
        if shard is not None:
            rank, world = shard
            jobs = [(i, j) for i, j in jobs if i % world == rank]
        self.jobs, self.cancel, self.pool = jobs, cancel, None
        self.lookahead = max(1, int(lookahead)) if lookahead else 32 * max(1, n_proc)
        # two runs per worker inside the window: one being prepared, one queued behind it
        self.run = max(1, int(run)) if run else max(1, min(16, self.lookahead // (2 * max(1, n_proc))))
        self.max_pending = 0                        # high-water mark of reads in flight or waiting (tests)
        if n_proc > 1 and len(jobs) > 1:
            import multiprocessing as mp
            from . import hdf5_lite
            hdf5_lite.preload()                     # before the fork: no worker resolves libzstd on its own ...
            _warm_up()                              # ... or pays numpy's / datetime's first-call initialisation
            self.pool = mp.get_context("fork" if "fork" in mp.get_all_start_methods() else None).Pool(min(n_proc, len(jobs)))

    def __len__(self):
        return len(self.jobs)

    def _reads(self):
        if self.pool is None:
            for _, job in self.jobs:
                yield _load_read(job)
            return
        runs = _runs([j for _, j in self.jobs], self.run)
        pending, nxt, out = deque(), 0, 0           # out: reads handed to the pool and not yet yielded
        while nxt < len(runs) or pending:
            # top the window up, then take the oldest run: at most `lookahead` reads are ever outstanding
            while nxt < len(runs) and (not pending or out + len(runs[nxt][1]) <= self.lookahead):
                pending.append(self.pool.apply_async(_load_run, (runs[nxt][1],)))
                out += len(runs[nxt][1])
                nxt += 1
            self.max_pending = max(self.max_pending, out)
            for read in pending.popleft().get():
                out -= 1
                yield read

    def __iter__(self):
        try:
            for (i, _), read in zip(self.jobs, self._reads()):
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
