"""
Output side of the basecaller (ub-bonito/bonito/io.py): FASTQ records + the summary TSV.

  biofmt         io.py:30-49      write_fastq   io.py:76-84     summary_file  io.py:148-155
  summary_row    io.py:190-237    CSVLogger     io.py:322-356   Writer.run    io.py:403-445

SAM/BAM/CRAM output, alignment and the CTC writer need pysam/mappy and are outside the
north-star path.
"""
import csv
import os
import sys
from collections import namedtuple
from logging import getLogger
from os.path import realpath, splitext
from threading import Thread

from .util import mean_qscore_from_qstring

logger = getLogger("bonito")
Format = namedtuple("Format", "aligned name mode")


def biofmt(aligned=False):
    mode, name = ("w", "sam") if aligned else ("wfq", "fastq")
    aligned = "aligned" if aligned else "unaligned"
    stdout = realpath("/dev/fd/1")
    if sys.stdout.isatty() or stdout.startswith("/proc"):
        return Format(aligned, name, mode)
    ext = stdout.split(os.extsep)[-1]
    if ext in ["fq", "fastq"]:
        return Format(aligned, "fastq", "wfq")
    elif ext == "bam":
        return Format(aligned, "bam", "wb")
    elif ext == "cram":
        return Format(aligned, "cram", "wc")
    elif ext == "sam":
        return Format(aligned, "sam", "w")
    return Format(aligned, name, mode)


def write_fasta(header, sequence, fd=sys.stdout):
    fd.write(f">{header}\n{sequence}\n")


def write_fastq(header, sequence, qstring, fd=sys.stdout, tags=None, sep="\t"):
    if tags is not None:
        fd.write(f"@{header} {sep.join(tags)}\n")
    else:
        fd.write(f"@{header}\n")
    fd.write(f"{sequence}\n+\n{qstring}\n")


def summary_file():
    stdout = realpath("/dev/fd/1")
    if sys.stdout.isatty() or stdout.startswith("/proc"):
        return "summary.tsv"
    return "%s_summary.tsv" % splitext(stdout)[0]


summary_field_names = [
    "filename", "read_id", "run_id", "channel", "mux", "start_time", "duration", "template_start",
    "template_duration", "sequence_length_template", "mean_qscore_template",
]


def summary_row(read, seqlen, qscore, alignment=False):
    fields = [read.filename, read.read_id, read.run_id, read.channel, read.mux, read.start, read.duration,
              read.template_start, read.template_duration, seqlen, qscore]
    return dict(zip(summary_field_names, fields))


class CSVLogger:
    def __init__(self, filename, sep=","):
        self.filename = str(filename)
        if os.path.exists(self.filename):
            with open(self.filename) as f:
                self.columns = csv.DictReader(f).fieldnames
        else:
            self.columns = None
        self.fh = open(self.filename, "a", newline="")
        self.csvwriter = csv.writer(self.fh, delimiter=sep)
        self.count = 0

    def set_columns(self, columns):
        if self.columns:
            raise Exception("Columns already set")
        self.columns = list(columns)
        self.csvwriter.writerow(self.columns)

    def append(self, row):
        if self.columns is None:
            self.set_columns(row.keys())
        self.csvwriter.writerow([row.get(k, "-") for k in self.columns])
        self.count += 1
        if self.count > 100:
            self.count = 0
            self.fh.flush()

    def close(self):
        self.fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()


class Writer(Thread):
    """Drains the (read, result) iterator: FASTQ to `fd`, one summary row per read, a (read_id, samples) log."""

    def __init__(self, mode, iterator, aligner=None, fd=sys.stdout, duplex=False, ref_fn=None, groups=None,
                 group_key=None, summary=None):
        super().__init__()
        if mode != "wfq" or aligner is not None:
            raise NotImplementedError("only unaligned FASTQ output is on the MI355X path (mode 'wfq')")
        self.fd = fd
        self.log = []
        self.mode = mode
        self.iterator = iterator
        self.fastq = True
        self.group_key = group_key
        self.summary = summary
        self.error = None

    def run(self):
        try:
            with CSVLogger(self.summary or summary_file(), sep="\t") as summary:
                for read, res in self.iterator:
                    seq = res["sequence"]
                    qstring = res.get("qstring", "*")
                    mean_qscore = res.get("mean_qscore", mean_qscore_from_qstring(qstring))
                    samples = len(read.signal)
                    read_id = read.read_id
                    tags = [f"RG:Z:{read.run_id}_{self.group_key}", f"qs:i:{round(mean_qscore)}",
                            *read.tagdata(), *res.get("mods", [])]
                    if len(seq):
                        write_fastq(read_id, seq, qstring, fd=self.fd, tags=tags)
                        summary.append(summary_row(read, len(seq), mean_qscore))
                        self.log.append((read_id, samples))
                    else:
                        logger.warning("> skipping empty sequence %s", read_id)
        except BaseException as e:  # surfaced by the CLI after join()
            self.error = e
            raise
