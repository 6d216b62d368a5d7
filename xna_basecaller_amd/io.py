"""
Output side of the basecaller: FASTQ records on stdout and one summary row per read.

Written from the OUTPUT FORMAT the reference produces (ub-bonito/bonito/io.py), not from its code:

  * stdout format is chosen from the name stdout is redirected to (io.py:30-49): `*.fq` / `*.fastq` / a tty / a pipe
    mean FASTQ; `.sam` means SAM text: the header (io.py:87-112: @HD VN:1.5 SO:unknown ob:0.0.1, @PG basecaller with version
    and command line, the reads' @RG lines) and one UNALIGNED record per read (io.py:115-145 with mapping = False: flag 4,
    `*` reference / CIGAR, NM:i:0, then the same tags as the FASTQ header) -- the reference hands both to pysam
    (`AlignmentFile(fd, 'w', text=header)` / `AlignedSegment.fromstring`, io.py:391-401,432-437), whose SAM text mode prints the
    header text as it is and each record as its own line; pysam is in no image, so the text is written directly.  The
    reference's header also carries `@PG ID:aligner PN:minimap2 VN:<mappy version>` whether or not anything is aligned; here
    that line is written only when an aligner version is given (none exists: mappy is in no image), so an unaligned file does
    not claim an aligner.  `.bam` / `.cram` need htslib and aligned records need mappy: refused;
  * a FASTQ record is  "@<read_id> <tag>\\t<tag>...\\n<sequence>\\n+\\n<qstring>\\n"  (io.py:76-84) with the tags
    RG:Z:<run_id>_<model>  qs:i:<rounded mean q>  mx:i  ch:i  st:Z  rn:i  f5:Z  (io.py:412-419, fast5.py:118-128);
  * the summary is `<stdout stem>_summary.tsv` (`summary.tsv` on a tty or a pipe; io.py:148-155), tab separated with the
    CSV dialect of Python's csv.writer (CRLF line ends, minimal quoting), header
    filename read_id run_id channel mux start_time duration template_start template_duration
    sequence_length_template mean_qscore_template (io.py:158-170, 190-206); appended to if it already exists;
  * reads whose called sequence is empty are skipped with a warning (io.py:444-445); the writer keeps a
    (read_id, samples) log from which the CLI prints samples per second (cli/basecaller.py:153-161).
"""
import os
import sys
from collections import namedtuple
from logging import getLogger
from threading import Thread

from .util import mean_qscore_from_qstring

logger = getLogger("bonito")
Format = namedtuple("Format", "aligned name mode")

SUMMARY_COLUMNS = ("filename", "read_id", "run_id", "channel", "mux", "start_time", "duration", "template_start",
                   "template_duration", "sequence_length_template", "mean_qscore_template")
_MODES = {"fq": ("fastq", "wfq"), "fastq": ("fastq", "wfq"), "sam": ("sam", "w"), "bam": ("bam", "wb"),
          "cram": ("cram", "wc")}


def _stdout_target():
    """Path stdout points at, or None for a terminal / pipe / socket."""
    if sys.stdout.isatty():
        return None
    target = os.path.realpath("/dev/fd/1")
    return None if target.startswith("/proc") else target


def biofmt(aligned=False):
    """Output format implied by stdout's file name: Format(aligned|unaligned, fastq|sam|bam|cram, open mode)."""
    kind = "aligned" if aligned else "unaligned"
    fallback = ("sam", "w") if aligned else ("fastq", "wfq")
    target = _stdout_target()
    ext = target.rsplit(os.extsep, 1)[-1] if target else ""
    name, mode = _MODES.get(ext, fallback)
    return Format(kind, name, mode)


def summary_file():
    target = _stdout_target()
    return "summary.tsv" if target is None else os.path.splitext(target)[0] + "_summary.tsv"


def write_fastq(header, sequence, qstring, fd=sys.stdout, tags=None, sep="\t"):
    title = header if tags is None else "%s %s" % (header, sep.join(tags))
    fd.write("@%s\n%s\n+\n%s\n" % (title, sequence, qstring))


def write_fasta(header, sequence, fd=sys.stdout):
    fd.write(">%s\n%s\n" % (header, sequence))


SAM_SPEC = "0.0.1"          # the reference's __ont_bam_spec__ (io.py:27)


def sam_header(groups, sep="\t", version=None, argv=None, aligner_version=None):
    """The SAM header text (io.py:87-112): @HD, @PG of the basecaller (PN:bonito -- the drop-in's program name -- with `version`
    and the command line `argv`), optionally the aligner's @PG, then the read-group lines; lines joined by os.linesep, one
    trailing newline."""
    from . import __version__
    lines = [sep.join(["@HD", "VN:1.5", "SO:unknown", "ob:%s" % SAM_SPEC]),
             sep.join(["@PG", "ID:basecaller", "PN:bonito", "VN:%s" % (__version__ if version is None else version),
                       "CL:bonito %s" % " ".join(sys.argv[1:] if argv is None else argv)])]
    if aligner_version is not None:
        lines.append(sep.join(["@PG", "ID:aligner", "PN:minimap2", "VN:%s" % aligner_version, "DS:mappy"]))
    return "%s\n" % os.linesep.join(lines + list(groups))


# minimap2's complement table (sketch.c `seq_comp_table`, what `mappy.revcomp` applies): the IUPAC codes in both cases, every
# other byte unchanged.  On the XNA alphabets that means X stays X and Y -- an IUPAC code -- becomes R; the reference calls
# mappy.revcomp on the called sequence as it is (io.py:133), so this is what its aligned records carry.  mappy is in no
# image: the table is restated, and the generator of tests/golden/sam.json uses the same restatement as its stand-in.
_COMP = {a: b for a, b in zip("ACGTUMRWSYKVHDBN", "TGCAAKYWSRMBDHVN")}
_COMP.update({a.lower(): b.lower() for a, b in list(_COMP.items())})
_COMP_TABLE = str.maketrans(_COMP)


def revcomp(sequence):
    """mappy.revcomp: reverse, then complement through minimap2's table."""
    return sequence[::-1].translate(_COMP_TABLE)


def sam_record(read_id, sequence, qstring, mapping=None, tags=None, sep="\t"):
    """One SAM record as text (io.py:115-145).  Unaligned (mapping false): flag 4, no reference, NM:i:0.  Aligned: `mapping` is
    a mappy.Alignment-shaped object (ctg, r_st, q_st, q_en, strand, mapq, cigar_str, NM, MD): flag 0 / 16, 1-based position,
    the CIGAR wrapped in the soft clips of the unaligned query ends (swapped on the reverse strand), the sequence reverse
    complemented on the reverse strand -- and the quality string left as it is, as the reference does.  The formatting is pinned
    by tests/golden/sam.json; producing a mapping needs mappy (`--reference`), which is in no image."""
    if mapping:
        tail = len(sequence) - mapping.q_en
        softclip = ["%sS" % mapping.q_st if mapping.q_st else "", mapping.cigar_str, "%sS" % tail if tail else ""]
        forward = mapping.strand == +1
        record = [read_id, 0 if forward else 16, mapping.ctg, mapping.r_st + 1, mapping.mapq,
                  "".join(softclip if forward else softclip[::-1]), "*", 0, 0,
                  sequence if forward else revcomp(sequence), qstring, "NM:i:%s" % mapping.NM, "MD:Z:%s" % mapping.MD]
    else:
        record = [read_id, 4, "*", 0, 0, "*", "*", 0, 0, sequence, qstring, "NM:i:0"]
    if tags is not None:
        record.extend(tags)
    return sep.join(map(str, record))


def _tsv_field(value):
    """One field in csv.writer's default dialect with a tab delimiter (quote only when needed, double the quotes)."""
    text = "" if value is None else str(value)
    if any(c in text for c in '\t"\r\n'):
        text = '"%s"' % text.replace('"', '""')
    return text


class SummaryTable:
    """Append-only TSV: the header is written when the file is new, otherwise the existing header's columns are kept."""

    def __init__(self, path, columns=SUMMARY_COLUMNS):
        self.path = str(path)
        self.columns = list(columns)
        fresh = not os.path.exists(self.path) or os.path.getsize(self.path) == 0
        if not fresh:
            with open(self.path, newline="") as fh:
                first = fh.readline().rstrip("\r\n")
            if first:
                self.columns = first.split("\t")
        self._fh = open(self.path, "a", newline="")
        self._pending = 0
        if fresh:
            self._line(self.columns)

    def _line(self, fields):
        self._fh.write("\t".join(_tsv_field(f) for f in fields) + "\r\n")

    def append(self, row):
        self._line([row.get(c, "-") for c in self.columns])
        self._pending += 1
        if self._pending > 100:
            self._fh.flush()
            self._pending = 0

    def close(self):
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def summary_row(read, seqlen, qscore):
    values = (read.filename, read.read_id, read.run_id, read.channel, read.mux, read.start, read.duration,
              read.template_start, read.template_duration, seqlen, qscore)
    return dict(zip(SUMMARY_COLUMNS, values))


class Writer(Thread):
    """Drains the (read, result) iterator on its own thread: FASTQ (mode 'wfq') or unaligned SAM text (mode 'w', header first)
    to `fd`, a summary row and a log entry per read."""

    def __init__(self, mode, iterator, aligner=None, fd=sys.stdout, duplex=False, ref_fn=None, groups=None,
                 group_key=None, summary=None):
        super().__init__()
        if mode not in ("wfq", "w") or aligner is not None or duplex:
            raise NotImplementedError("unaligned FASTQ (mode 'wfq') and unaligned SAM text (mode 'w') are on the MI355X path; "
                                      "BAM / CRAM need htslib, aligned records need mappy")
        self.mode, self.fd, self.iterator = mode, fd, iterator
        self.group_key = group_key
        self.groups = sorted(groups) if groups else []
        self.summary = summary
        self.log = []
        self.error = None

    def _emit(self, table, read, res):
        seq = res["sequence"]
        if not len(seq):
            logger.warning("> skipping empty sequence %s", read.read_id)
            return
        qstring = res.get("qstring", "*")
        mean_q = res.get("mean_qscore")
        if mean_q is None:
            mean_q = mean_qscore_from_qstring(qstring)
        tags = ["RG:Z:%s_%s" % (read.run_id, self.group_key), "qs:i:%d" % round(mean_q)]
        tags += list(read.tagdata()) + list(res.get("mods", []))
        if self.mode == "wfq":
            write_fastq(read.read_id, seq, qstring, fd=self.fd, tags=tags)
        else:
            self.fd.write(sam_record(read.read_id, seq, qstring, res.get("mapping", False), tags=tags) + "\n")
        table.append(summary_row(read, len(seq), mean_q))
        self.log.append((read.read_id, len(read.signal)))

    def run(self):
        try:
            with SummaryTable(self.summary or summary_file()) as table:
                if self.mode == "w":
                    self.fd.write(sam_header(self.groups))
                for read, res in self.iterator:
                    self._emit(table, read, res)
        except BaseException as e:  # surfaced by the CLI after join()
            self.error = e
            raise
