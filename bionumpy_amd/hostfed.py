"""Host-fed FASTQ -> k-mer histogram: text that lives in (page-locked) host memory is streamed to HBM with
``hipMemcpyAsync`` on its own HIP stream while the kernels of the previous chunks run on another.

This is the seam the reference's GPU backend has (``cupy.asanyarray(chunk)`` of the raw uint8 chunk,
bionumpy/cupy_compatible/parser.py:10-17, fed by bionumpy/io/parser.py:192-206), built for a device that decodes
16 GB of text in 13 ms: the link is the bottleneck (2.1 bytes per base over PCIe), so everything else has to hide
behind it.

* the text is cut into chunks at record boundaries by looking at a few hundred bytes around every cut on the host
  (a header line is a line that starts with '@' and whose line after next starts with '+' — the rule the
  validation kernel applies, one_line_buffer.py:156-173, fastq_buffer.py:39-45); no chunk is ever re-read;
* a feeder thread keeps up to ``ring`` chunk buffers in HBM filled: ``hipMemcpyAsync`` on the copy stream, an event
  per buffer in each direction (filled / free); ctypes releases the GIL, so the copies are enqueued while the main
  thread sits in a synchronising call of the compute stream (the census returns totals to the host);
* every chunk is decoded by the fused census + encode kernels straight into the batch's flat 2-bit stream at a
  64-base aligned offset; the <= 63 padding positions between two chunks are marked as read ends, so that no k-mer
  starts on them or spans them (the k-mer start mask is "no read end among the next k - 1 positions");
* when the last chunk of a batch is in, the counting stage (k-mer generation fused with the first radix level, second
  level, finishing kernels) runs on the whole batch — while the feeder is already copying the next batch's chunks.
"""
import ctypes as C
import queue
import threading
import time
from collections import namedtuple

import numpy as np

from ._native import lib, check
from .device import Device, HArray, ptr
from .exceptions import EncodingError, FormatException
from .ops import get_ops, NONE
from .pipeline import BatchStats

HostFedTiming = namedtuple("HostFedTiming", "seconds bytes h2d_seconds h2d_gb_per_s compute_seconds overlap_frac chunks")


def cut_points(text, chunk_bytes, header=ord("@"), plus=ord("+"), window=1 << 16):
    """chunk ends for a FASTQ text (numpy uint8 view): each one the start of a record (or the end of the text), at most
    ``chunk_bytes`` apart.  Only the ``window`` bytes before every tentative cut are looked at."""
    n = text.size
    cuts, at = [], 0
    while n - at > chunk_bytes:
        hi = at + chunk_bytes
        lo = max(at + 1, hi - window)
        found = -1
        while found < 0:
            seg = text[lo:min(hi + 2048, n)]
            nl = np.flatnonzero(seg == 10)
            # line starts inside [lo, hi): after every newline; a record starts where the line begins with the header byte
            # and the line after next begins with '+'
            starts = nl + 1
            for j in range(len(starts) - 1, -1, -1):
                s = starts[j]
                if lo + s > hi or j + 2 >= len(starts) or starts[j + 2] >= len(seg):
                    continue
                if seg[s] == header and seg[starts[j + 2]] == plus:
                    found = lo + int(s)
                    break
            if found < 0:
                if lo == at + 1:
                    raise FormatException("no record boundary in a chunk of %d bytes" % chunk_bytes, line_number=0)
                lo = max(at + 1, lo - 4 * window)
        cuts.append(found)
        at = found
    cuts.append(n)
    return cuts


class HostFedCounter:
    """k-mer histograms of FASTQ batches held in host memory (see the module docstring)."""

    def __init__(self, k, chunk_bytes=1 << 30, ring=6, canonical=False):
        assert 13 < k < 32, "the host-fed path feeds the sparse (k > 13) histogram"
        import torch
        self._t = torch
        self.k, self.canonical = k, canonical
        self.chunk_bytes = int(chunk_bytes)
        self.ops = get_ops()
        self.dev = Device.get()
        self.copy_stream = torch.cuda.Stream()
        self.ring = [torch.empty(self.chunk_bytes + 64, dtype=torch.uint8, device=self.dev.tdev) for _ in range(ring)]
        self.filled = [torch.cuda.Event() for _ in range(ring)]
        self.freed = [torch.cuda.Event() for _ in range(ring)]
        self.free_slots = threading.Semaphore(ring)
        self._stop = threading.Event()
        self._h2d_events = []

    # -- feeder thread: host memory -> ring buffers on the copy stream ---------------------------------------------------
    def _feed(self, batches, out):
        try:
            self._t.cuda.set_device(self.dev.index)
            i = 0
            for b, (text, cuts) in enumerate(batches):
                at = 0
                for c, end in enumerate(cuts):
                    self.free_slots.acquire()
                    if self._stop.is_set():                  # the consumer has gone (an exception, an abandoned generator)
                        out.put(None)
                        return
                    slot = i % len(self.ring)
                    n = end - at
                    s = self.copy_stream.cuda_stream
                    self.copy_stream.wait_event(self.freed[slot])
                    begin, end_ev = self._t.cuda.Event(enable_timing=True), self._t.cuda.Event(enable_timing=True)
                    begin.record(self.copy_stream)
                    check(lib.bnpk_copy_h2d_async(C.c_void_p(self.ring[slot].data_ptr()),
                                                  C.c_void_p(text.__array_interface__["data"][0] + at), n, C.c_void_p(s)))
                    end_ev.record(self.copy_stream)
                    self.filled[slot].record(self.copy_stream)
                    self._h2d_events.append((begin, end_ev, n))
                    out.put((b, slot, n, c == len(cuts) - 1))
                    at = end
                    i += 1
            out.put(None)
        except BaseException as e:                       # (surface the error in the consumer)
            out.put(e)

    # -- main thread ---------------------------------------------------------------------------------------------------
    def run(self, batches):
        """batches: list of page-locked uint8 numpy arrays, each a whole FASTQ text.  Yields ((keys, counts), BatchStats)
        per batch; ``self.timing`` afterwards."""
        t, ops = self._t, self.ops
        plan = [(text, cut_points(text, self.chunk_bytes)) for text in batches]
        for text, cuts in plan:
            assert max(np.diff([0] + cuts)) <= self.chunk_bytes
        q = queue.Queue()
        compute = t.cuda.current_stream()
        for ev in self.freed:
            ev.record(compute)
        self._h2d_events, n_chunks, total_bytes = [], 0, 0
        self._stop.clear()
        self.free_slots = threading.Semaphore(len(self.ring))
        t0 = time.perf_counter()
        feeder = threading.Thread(target=self._feed, args=(plan, q), name="bnpk-feeder", daemon=True)
        feeder.start()
        try:
            yield from self._consume(plan, q, compute, t0)
        finally:
            # however the consumer ends — all batches done, malformed input raising out of _count, the generator dropped
            # half way — the feeder must not stay blocked on a slot (it holds this object and its ring of chunk buffers)
            self._stop.set()
            for _ in self.ring:
                self.free_slots.release()
            feeder.join()
            self.copy_stream.synchronize()

    def _consume(self, plan, q, compute, t0):
        t, ops = self._t, self.ops
        n_chunks, total_bytes = 0, 0
        compute_s, paused = 0.0, 0.0                     # (paused: time the consumer of the generator spends between batches)
        state = None
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            b, slot, n, last = item
            if state is None:                            # a new batch: its flat 2-bit stream and read-end mask
                cap_bases = plan[b][0].size // 2 + 64 * (len(plan[b][1]) + 1)
                state = {"packed": ops._empty(cap_bases // 32 + 4, np.int64), "ends": ops._empty(cap_bases // 64 + 4, np.int64),
                         "bases": 0, "real_bases": 0, "reads": 0, "bytes": 0, "errs": []}
            compute.wait_event(self.filled[slot])
            c0 = time.perf_counter()
            n_reads, n_bases, err = ops.fastq_encode_into(self.ring[slot], n, 4, 1, ord("@"), True, state["packed"],
                                                          state["ends"], state["bases"])
            self.freed[slot].record(compute)
            self.free_slots.release()
            state["errs"].append((err, state["reads"], state["real_bases"]))
            state["reads"] += n_reads
            state["real_bases"] += n_bases
            state["bytes"] += n
            pad_to = (state["bases"] + n_bases + 63) // 64 * 64
            if pad_to > state["bases"] + n_bases:        # padding positions count as read ends: no k-mer starts there
                w, lo_bit = (state["bases"] + n_bases) // 64, (state["bases"] + n_bases) % 64
                bits = ((1 << 64) - 1) ^ ((1 << lo_bit) - 1)
                state["ends"][w] |= bits - (1 << 64) if bits >= (1 << 63) else bits
            state["bases"] = pad_to
            n_chunks += 1
            total_bytes += n
            if last:
                result = self._count(state)
                state = None
                compute_s += time.perf_counter() - c0
                p0 = time.perf_counter()
                yield result
                del result
                paused += time.perf_counter() - p0
            else:
                compute_s += time.perf_counter() - c0
        t.cuda.synchronize()
        wall = time.perf_counter() - t0 - paused
        h2d_s = sum(a.elapsed_time(b) for a, b, _ in self._h2d_events) * 1e-3          # time the copy engine was busy
        rate = total_bytes / h2d_s / 1e9 if h2d_s > 0 else 0.0
        overlap = 0.0 if min(h2d_s, compute_s) <= 0 else max(0.0, min(1.0, (h2d_s + compute_s - wall) / min(h2d_s, compute_s)))
        self.timing = HostFedTiming(wall, total_bytes, h2d_s, rate, compute_s, overlap, n_chunks)

    def _count(self, state):
        ops, k = self.ops, self.k
        # the reference validates the whole buffer before it encodes anything (io/one_line_buffer.py:45-71, then
        # encodings/alphabet_encoding.py:37-45 on the gathered sequences): format errors of any chunk come first; the
        # offset of an invalid base counts the bases of the whole batch, not of its chunk
        cells = [(err.cpu().numpy(), reads_before, bases_before) for err, reads_before, bases_before in state["errs"]]
        for e, reads_before, _ in cells:
            if e[0] != NONE:
                raise FormatException("Expected header line to start with @", line_number=(int(e[0]) + reads_before) * 4)
            if e[1] != NONE:
                raise FormatException("Expected '+' at third line of entry", line_number=2 + (int(e[1]) + reads_before) * 4)
        for e, _, bases_before in cells:
            if e[2] != NONE:
                raise EncodingError("Error when encoding to AlphabetEncoding('ACGT'): invalid character", int(e[2]) + bases_before)
        n_pos = state["bases"]
        packed, ends = HArray(dev=state["packed"]), HArray(dev=state["ends"])
        starts_mask, n_kmers = ops.kmer_starts_from_ends(ends, n_pos, k)
        del ends
        stats = BatchStats(state["reads"], state["real_bases"], n_kmers, state["bytes"])
        skew = 2.0 if self.canonical else 1.0
        levels = ops.radix_plan(int(n_kmers * skew), 2 * k)
        bits = levels[0] if levels else 0
        hashes, cuts = ops.kmers_partitioned(packed, starts_mask, n_pos, n_kmers, k, bits, canonical=self.canonical)
        del packed, starts_mask
        hist = ops.count_sparse(hashes, key_bits=2 * k, consume=True, partition=(cuts, bits) if bits else None, skew=skew)
        return hist, stats
