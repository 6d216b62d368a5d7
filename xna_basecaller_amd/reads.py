"""
Read objects and signal preparation (ub-bonito/bonito/fast5.py).

  Read (metadata + scaled, trimmed, normalised fp32 signal)  fast5.py:22-128
  trim                                                        fast5.py:149-171
  med_mad                                                     fast5.py:174-180
  norm_by_noisiest_section                                    fast5.py:183-204
  get_reads                                                   fast5.py:284-296

fast5 is HDF5 + the VBZ filter; neither libhdf5/h5py nor ont_fast5_api exists in this image,
so reads come from a pre-extracted signal bundle (`*.xsig.npz`, written by `write_bundle`):
raw int16 DACs plus the attributes fast5.py reads (channel_id, tracking_id, Raw attrs).
Everything after the HDF5 decode is the reference's arithmetic.
"""
import json
import os
from collections import OrderedDict
from datetime import datetime, timedelta
from glob import glob
from pathlib import Path

import numpy as np

__all__ = ["Read", "trim", "med_mad", "norm_by_noisiest_section", "get_reads", "write_bundle",
           "SyntheticRead"]


def med_mad(x, factor=1.4826):
    """Median and scaled median absolute deviation (+ float32 eps so it is never zero)."""
    med = np.median(x)
    mad = np.median(np.absolute(x - med)) * factor + np.finfo(np.float32).eps
    return med, mad


def trim(signal, window_size=40, threshold_factor=2.4, min_elements=3):
    """
    Find where the open-pore/adapter prefix ends.  Skip 10 samples; threshold = med + 2.4*mad of the
    last 100 windows; walk 40-sample windows: once a window has more than `min_elements` samples
    above threshold, return the end of the first window whose last sample is back under it.
    """
    min_trim = 10
    signal = signal[min_trim:]
    med, mad = med_mad(signal[-(window_size * 100):])
    threshold = med + mad * threshold_factor
    num_windows = len(signal) // window_size
    seen_peak = False
    for pos in range(num_windows):
        end = (pos + 1) * window_size
        window = signal[end - window_size:end]
        if seen_peak or np.count_nonzero(window > threshold) > min_elements:
            seen_peak = True
            if window[-1] > threshold:
                continue
            return min(end + min_trim, len(signal)), len(signal)
    return min_trim, len(signal)


def norm_by_noisiest_section(signal, samples=100, threshold=6.0):
    """
    med/mad normalisation using the widest run of 100-sample windows whose std exceeds
    std(signal)/threshold (short reads, fast5.py:183-204).
    """
    from scipy.signal import find_peaks
    threshold = signal.std() / threshold
    noise = np.ones(signal.shape)
    for idx in np.arange(signal.shape[0] // samples):
        window = slice(idx * samples, (idx + 1) * samples)
        noise[window] = np.where(signal[window].std() > threshold, 1, 0)
    noise[0] = 0
    noise[-1] = 0
    peaks, info = find_peaks(noise, width=(None, None))
    if len(peaks):
        widest = np.argmax(info["widths"])
        med, mad = med_mad(signal[info["left_bases"][widest]: info["right_bases"][widest]])
    else:
        med, mad = med_mad(signal)
    return (signal - med) / mad


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
        self._groupdict = OrderedDict([
            ("ID", f"{self.run_id}_{model}"), ("PL", "ONT"), ("DT", f"{self.exp_start_time}"),
            ("PU", f"{self.flow_cell_id}"), ("PM", f"{self.device_id}"), ("LB", f"{self.sample_id}"),
            ("SM", f"{self.sample_id}"),
            ("DS", "%s" % " ".join([f"run_id={self.run_id}", f"basecall_model={model}"])),
        ])
        return "\t".join(["@RG", *[f"{k}:{v}" for k, v in self._groupdict.items()]])

    def tagdata(self):
        return [
            f"mx:i:{self.mux}",
            f"ch:i:{self.channel}",
            f"st:Z:{self.start_time}",
            f"rn:i:{self.read_number}",
            f"f5:Z:{self.filename}",
        ]


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

    def tagdata(self):
        return [f"mx:i:{self.mux}", f"ch:i:{self.channel}", f"st:Z:{self.start_time}",
                f"rn:i:{self.read_number}", f"f5:Z:{self.filename}"]


def write_bundle(path, reads):
    """reads: iterable of (raw int16 array, attrs dict with at least read_id/range/digitisation/offset/sampling_rate)."""
    arrays, metas = {}, []
    for i, (raw, attrs) in enumerate(reads):
        arrays["raw_%d" % i] = np.asarray(raw, dtype=np.int16)
        metas.append(attrs)
    arrays["meta"] = np.frombuffer(json.dumps(metas).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrays)


def _bundle_reads(filename, read_ids=None, skip=False):
    with np.load(filename) as z:
        metas = json.loads(bytes(z["meta"]).decode())
        for i, attrs in enumerate(metas):
            rid = attrs["read_id"]
            if read_ids is None or (rid in read_ids) ^ skip:
                yield Read(z["raw_%d" % i], attrs, filename)


def get_reads(directory, read_ids=None, skip=False, n_proc=1, recursive=False, cancel=None):
    """All reads of every signal bundle under `directory`, in file then in-file order."""
    pattern = "**/*.xsig.npz" if recursive else "*.xsig.npz"
    for fn in sorted(Path(x) for x in glob(directory + "/" + pattern, recursive=True)):
        for read in _bundle_reads(fn, read_ids=read_ids, skip=skip):
            yield read
            if cancel is not None and cancel.is_set():
                return
