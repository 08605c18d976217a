"""Which part of a sequence file a rank reads ("partition by chunk — contiguous byte ranges cut at record
boundaries").

The reference's unit of work is the chunk (bionumpy/io/parser.py:96-171) and records are independent, so N ranks read N
byte ranges of the file.  Where a range begins is decided by ONE function of the file, B(x) = the offset of the first
record that starts at or behind byte x, so that rank r reads exactly the records that start in [B(lo_r), B(hi_r)) with
lo_r = r * S / N, hi_r = lo_{r+1}: every record belongs to exactly one rank, the ranks' records follow each other in file
order, and a record that straddles hi_r is read to its end by rank r (and skipped by rank r + 1).  Nothing is exchanged
to find the cuts: both neighbours evaluate B at the same x.

What starts a record is the reader's own validation rule, applied forwards (bionumpy/io/one_line_buffer.py:156-173,
bionumpy/io/fastq_buffer.py:39-45):

* FASTQ — a line that starts with '@' whose line + 2 starts with '+'.  A quality line may start with '@' too; the line
  two behind it is then the next record's sequence, which never starts with '+', so it is never taken for a header.
* FASTA (two-line and multi-line) — a line that starts with '>'.

Three kinds of file:

* plain — byte ranges as above;
* BGZF (bgzip) — ranges of the COMPRESSED file, moved forward to the next member (their sizes are in the members' ``BC``
  fields), the same B on the inflated text behind the cut; a member header is told from compressed bytes that look like
  one by following the chain of member sizes from it (``bgzf_member_at_or_after``);
* any other gzip stream — cannot be entered anywhere but at its start: every rank inflates everything and keeps the chunks
  i with i % N == r (``chunk_modulo``; the documented fall-back).
"""
import os

import numpy as np

NEWLINE = 10


class Shard:
    """(rank, world[, process group]) of a reader"""

    def __init__(self, rank, world, group=None):
        rank, world = int(rank), int(world)
        if not 0 <= rank < world:
            raise ValueError("shard: rank %d of %d" % (rank, world))
        self.rank, self.world, self.group = rank, world, group

    def __iter__(self):
        return iter((self.rank, self.world))

    def __repr__(self):
        return "Shard(%d of %d)" % (self.rank, self.world)


def resolve_shard(shard):
    """``bnp.open(..., shard=...)`` -> Shard or None.

    Sharding is OPT-IN (round 6): a script written against the reference and launched with torchrun keeps getting
    the whole file on every rank — only the reductions that end in a merge over the ranks (``streamable`` reductions of
    histograms, bincounts, sums) give the whole file's answer from a part per rank; a manual chunk loop or ``.read()`` would
    silently see a 1/N part.  None: the whole file, unless the environment says BNPK_SHARD=auto (or 1); "auto": the ranks of
    ``torch.distributed`` if a process group has been initialised with more than one rank (the job reads every file ONCE, a
    part per rank); False / "off": the whole file; (rank, world) or (rank, world, group): as given; a Shard: itself."""
    if isinstance(shard, Shard):
        return shard if shard.world > 1 else None
    if shard is False or shard == "off":
        return None
    if shard is None and os.environ.get("BNPK_SHARD", "0").lower() not in ("1", "auto"):
        return None
    if shard is None or shard == "auto":
        try:
            import torch.distributed as dist
        except Exception:                                    # noqa: BLE001
            return None
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
            return None
        return Shard(dist.get_rank(), dist.get_world_size())
    s = Shard(*shard)
    return s if s.world > 1 else None


# ---- what starts a record --------------------------------------------------------------------------------------------
class RecordRule:
    def __init__(self, header, plus_line=None):
        self.header = ord(header)
        self.plus_line = plus_line                           # lines behind the header whose first byte must be '+' (FASTQ: 2)

    @classmethod
    def of(cls, buffer_type):
        header = getattr(buffer_type, "_new_entry_marker", None) or buffer_type.HEADER
        return cls(header, 2 if getattr(buffer_type, "_check_plus", False) else None)


def first_record_start(window, at_line_start, rule, eof):
    """index of the first record start in ``window`` (uint8 array) — or None (the window is too short to tell: read
    more), or -1 (there is none up to the end of the file).
    at_line_start: byte 0 of the window is the first byte of a line; eof: the window reaches the end of the file."""
    w = np.asarray(window, dtype=np.uint8)
    n = w.size
    starts = np.flatnonzero(w == NEWLINE) + 1
    if at_line_start:
        starts = np.concatenate([[0], starts])
    starts = starts[starts < n]                              # (a line that begins behind the window: not seen yet)
    if starts.size == 0:
        return -1 if eof else None
    is_header = w[starts] == rule.header
    if rule.plus_line is None:
        hit = np.flatnonzero(is_header)
        if hit.size:
            return int(starts[hit[0]])
        return -1 if eof else None
    d = rule.plus_line
    known = np.arange(starts.size) + d < starts.size         # the line d behind it begins inside the window
    plus = np.zeros(starts.size, dtype=bool)
    plus[:starts.size - d] = w[starts[d:]] == ord("+") if starts.size > d else False
    good = is_header & known & plus
    unsure = is_header & ~known                              # a header candidate whose '+' line is not in the window yet
    first_good = int(np.argmax(good)) if good.any() else starts.size
    first_unsure = int(np.argmax(unsure)) if unsure.any() else starts.size
    if first_good < first_unsure:
        return int(starts[first_good])
    if first_unsure < starts.size and not eof:
        return None
    return -1 if eof else None                               # (at the end of the file an unfinished record starts nothing)


def record_start_at_or_after(read_at, size, x, rule, probe=1 << 16):
    """B(x): offset of the first record of the file that starts at or behind byte x (0 for x <= 0, ``size`` if there is
    none).  read_at(offset, n) -> bytes of the file."""
    if x <= 0:
        return 0
    if x >= size:
        return size
    n = probe
    while True:
        raw = read_at(x - 1, min(n + 1, size - (x - 1)))
        w = np.frombuffer(raw, dtype=np.uint8)
        eof = (x - 1) + w.size >= size
        idx = first_record_start(w[1:], w[0] == NEWLINE, rule, eof)
        if idx is None:
            n *= 4
            continue
        return size if idx < 0 else x + idx


def plain_byte_range(fd, size, shard, rule):
    """[start, stop) of the file that ``shard`` reads: record-aligned on both sides"""
    read_at = lambda off, n: os.pread(fd, n, off)
    lo = size * shard.rank // shard.world
    hi = size * (shard.rank + 1) // shard.world
    start = record_start_at_or_after(read_at, size, lo, rule)
    stop = size if shard.rank == shard.world - 1 else record_start_at_or_after(read_at, size, hi, rule)
    return start, max(start, stop)


# ---- BGZF: cuts between members --------------------------------------------------------------------------------------
def bgzf_member_at_or_after(read_at, size, x, block_size, chain=3):
    """offset of the first BGZF member that starts at or behind compressed byte x (``size`` if none).  Compressed data may
    contain the four magic bytes anywhere, so a candidate counts only if ``chain`` members follow each other from it
    (or the chain reaches the end of the file)."""
    if x <= 0:
        return 0
    pos = x
    while pos < size:
        raw = read_at(pos, min(1 << 17, size - pos))
        at = raw.find(b"\x1f\x8b\x08\x04")
        while at >= 0:
            cand = pos + at
            p, ok = cand, True
            for _ in range(chain):
                if p == size:
                    break
                bs = block_size(read_at(p, min(64, size - p)))
                if bs is None or p + bs > size:
                    ok = False
                    break
                p += bs
            if ok:
                return cand
            at = raw.find(b"\x1f\x8b\x08\x04", at + 1)
        if len(raw) < 4:
            break
        pos += len(raw) - 3                                  # (a magic that straddles two reads)
    return size
