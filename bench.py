#!/usr/bin/env python
"""bench.py — FASTQ -> k-mer histogram throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path (newline scan + validation, field table, ragged gather + 2-bit
encode, 31-mer hashes generated straight into an MSD radix partition, in-LDS finishing sort + run-length
histogram; for N > 1 plus the key-range all-to-all over RCCL)
over one HBM-resident batch of synthetic FASTQ: ``--reads`` reads of ``--read-len`` bp PER GPU (weak
scaling; default 50 M x 150 bp = BASELINE config 2, 15.8 GB of text, 7.5 Gbases, 6.0 G 31-mers per GPU).
The input is generated on the device before the timed region; nothing is cached between steps.

Rank 0 prints ONE JSON line: value = Gbases/s of the whole job (all GPUs), plus
  roofline     — the dominant kernel of the step: algorithmic bytes per launch / its hipEvent-measured
                 average launch duration vs the 8 TB/s HBM peak (DESIGN.md §5 states the byte model)
  cpu_baseline — the numpy oracle (a statement-by-statement port of the reference's numpy path, which
                 cannot be imported here: npstructures is absent) timed on a bounded sample of the same
                 reads on the host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable

# algorithmic HBM bytes per launch of each kernel, as a function of the batch
def algorithmic_bytes(name, s, read_len, k, n_distinct=None):
    n, bases, kmers, text = s.n_reads, s.n_bases, s.n_kmers, s.n_bytes
    distinct = kmers if n_distinct is None else n_distinct
    return {
        "fastq_census": text,                                 # fused decode, pass 1: read the text once
        "fastq_encode": text + bases / 4 + bases / 8,         # pass 2: read it again, write packed bases + read-end mask
        "kmer_starts_from_ends": bases / 8 + bases / 8,       # read-end mask in, k-mer start mask out
        "byte_census": text,                                  # read the text once
        "byte_positions": text + 8 * 4 * n,                   # read it again, write 4 newline offsets/read
        "validate_entries": 2 * 8 * n + 2 * n,                # two offsets + two probe bytes per read
        "field_table": 2 * 8 * n + 2 * 8 * n,                 # two offsets in, start+len out
        "row_offsets": 8 * n + 8 * n,                         # lens in, offsets out
        "gather_encode_dna": bases + 16 * n + bases / 4,      # sequence bytes + start/offset + packed out
        "kmer_start_mask": 8 * n + bases / 8,                 # offsets in, one bit per base out
        "kmers": bases / 4 + 16 * n + 8 * kmers,              # packed in, offsets, 8 B per k-mer out
        "kmers_partition_hist": bases / 4 + bases / 8,        # packed reads + start mask in, digit counts out
        "kmers_partition_scatter": bases / 4 + bases / 8 + 8 * kmers,   # same input, every hash written once
        "radix_hist": 8 * kmers,                              # read every key
        "radix_scatter": 16 * kmers,                          # read every key, write it to its bucket
        "radix_scatter_claimed": 16 * kmers,                  # the same level without its histogram pass (round 5): places claimed line by line
        "claimed_finalize": 16 * 30 * (kmers / 5722) + 16 * (kmers / 5722),   # ~30 leftover keys per bucket moved + the fill counters scanned
        "finish_sorted": 8 * kmers + 16 * distinct,           # read the bucketed keys, write key + count per distinct key
        "sort_keys": 16 * kmers,                              # (fallback path) read every key once, write it once sorted
        "run_census": 8 * kmers,
        "run_heads": 8 * kmers + 16 * kmers,
        "run_sums": 16 * kmers,
        "count_dense_lds": 8 * kmers,
        "count_dense_global": 8 * kmers,
    }.get(name)


def _oracle_pass(args):
    """one process of the nproc-sharded baseline: the whole oracle path over its shard of the sample"""
    host_text, k = args
    import oracle
    t0 = time.perf_counter()
    res = oracle.scan_one_line_buffer(host_text, oracle.FASTQ)
    starts, lens = res.field_starts[:, 1], res.field_lens[:, 1]
    codes = oracle.encode_dna(oracle.gather_rows(host_text, starts, lens))
    h, _ = oracle.get_kmers(codes, lens, k)
    if k <= 8:
        oracle.count_dense(h, k)
    else:
        oracle.count_sparse(h)
    return codes.size, time.perf_counter() - t0


def cpu_baseline_sharded(host_text, record_bytes, k):
    """the same sample cut into one shard of whole records per host core, one process each (chunks shard
    embarrassingly; the per-shard histograms are not merged, which only flatters the CPU)"""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    n_rec = host_text.size // record_bytes
    per = min(n_rec, 50_000)                               # ~0.2 s of numpy per process: start-up does not dominate
    # every process gets `per` reads of the sample (rotated; the data are i.i.d., the work per shard is the same)
    shards = [host_text[(i * per) % max(n_rec - per + 1, 1) * record_bytes:][:per * record_bytes] for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(len(shards)) as pool:
        done = pool.map(_oracle_pass, [(s, k) for s in shards])
    dt = time.perf_counter() - t0
    return {"value": float(sum(d[0] for d in done) / dt / 1e9), "unit": "Gbases/s", "cores": len(shards), "kind": "port",
            "sample": "%d processes (one per host core) x %d reads each from the same sample, wall %.1f s, shard histograms not merged"
                      % (len(shards), per, dt)}


def cpu_baseline(host_text, k, budget_s=20.0):
    """the oracle (numpy port of the reference path) on a bounded sample of the same reads, 1 thread"""
    import numpy as np
    import oracle
    t0 = time.perf_counter()
    res = oracle.scan_one_line_buffer(host_text, oracle.FASTQ)
    starts, lens = res.field_starts[:, 1], res.field_lens[:, 1]
    codes = oracle.encode_dna(oracle.gather_rows(host_text, starts, lens))
    h, _ = oracle.get_kmers(codes, lens, k)
    if k <= 8:
        oracle.count_dense(h, k)
    else:
        oracle.count_sparse(h)
    dt = time.perf_counter() - t0
    return {"value": float(codes.size / dt / 1e9), "unit": "Gbases/s", "cores": 1, "kind": "port",
            "sample": "%d reads x %d bp of the same synthetic FASTQ (%.1f s of numpy: flatnonzero scan, "
                      "ragged gather, LUT encode, BitArray pack+sliding_window, np.unique)"
                      % (res.n_records, int(lens[0]) if len(lens) else 0, dt)}


def _index_roofline(kernels_ms, n_pairs, n_distinct, build_ms, note=None):
    """config 5 against the HBM peak.  bnpk_index_build (round 6: no library sort on its path) is one partition of (k-mer, row) words
    and a sparse count of them — a dozen small launches at sacCer3's size, so the figure that means something is the WHOLE build: algorithmic bytes = every (k-mer, row)
    pair read once (16 B) and every distinct pair written once (16 B), over the build's wall time; the dominant kernel is named
    with its own time (the group timer "finish_sorted" spans the finishing kernels and is not one of them)"""
    if not kernels_ms:
        return None
    names = [n for n in kernels_ms if n != "finish_sorted"] or list(kernels_ms)
    dom = max(names, key=kernels_ms.get)
    bytes_ = 16 * n_pairs + 16 * n_distinct
    achieved = bytes_ / (build_ms * 1e-3) / 1e9 if build_ms > 0 else None
    return {"bound": "hbm", "kernel": dom, "avg_launch_ms": kernels_ms[dom], "algorithmic_bytes_per_build": bytes_,
            "achieved": None if achieved is None else round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
            "note": note or ("whole build (%d timed launches): one partition of (k-mer, row) words; the yeast genome's 31-mers fill their "
                             "top digits unevenly (buckets 7x the average) and share long prefixes inside a bucket, which is where the "
                             "finishing kernels' time goes; see synthetic_1e9_pairs for the same call at a size where the kernels set "
                             "the time" % len(kernels_ms))}


def host_fed_leg(args, text, expected_sums):
    """the batch copied once into page-locked host memory, then streamed back --host-fed-batches times through
    bionumpy_amd.hostfed.HostFedCounter (1 GiB chunks, copy stream + compute stream); every histogram is checked
    against the checksums of the k-mers in read order"""
    import ctypes as C
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fullsize
    from bionumpy_amd._native import lib, check
    from bionumpy_amd.hostfed import HostFedCounter
    from bionumpy_amd.io.pinned import PinnedBuffer
    n = text.size
    t0 = time.perf_counter()
    pinned = PinnedBuffer(n + 64)
    t_alloc = time.perf_counter() - t0
    check(lib.bnpk_copy_d2h_async(pinned.ptr, C.c_void_p(text.dev().data_ptr()), n, None))
    torch.cuda.synchronize()
    host_text = pinned.array[:n]
    counter = HostFedCounter(args.k, chunk_bytes=1 << 30, ring=6, canonical=args.canonical)
    for (keys, counts), st in counter.run([host_text]):      # warm-up batch (allocator, first touch of the ring)
        del keys, counts
    n_bases, ok = 0, True
    for (keys, counts), st in counter.run([host_text] * args.host_fed_batches):
        n_bases += st.n_bases
        ok = ok and fullsize.histogram_sums(keys, counts) == expected_sums
        del keys, counts
    tm = counter.timing
    assert ok, "host-fed histogram differs from the device-resident one"
    del counter
    pinned.free()
    return {"gbases_per_s": round(n_bases / tm.seconds / 1e9, 3), "h2d_gb_per_s": round(tm.h2d_gb_per_s, 2),
            "overlap_frac": round(tm.overlap_frac, 3), "source": "pinned RAM", "batches": args.host_fed_batches,
            "bytes": tm.bytes, "seconds": round(tm.seconds, 4), "h2d_busy_s": round(tm.h2d_seconds, 4),
            "compute_host_s": round(tm.compute_seconds, 4), "chunks": tm.chunks, "chunk_bytes": 1 << 30,
            "pinned_alloc_s": round(t_alloc, 3), "parity": "checksums of every batch == k-mers in read order",
            "link_bound_gbases_per_s": round(tm.h2d_gb_per_s / (n / n_bases * args.host_fed_batches), 2) if n_bases else None}


def extras(args, ops, dev, main_stats, copy_rate):
    """the secondary workloads under the same clock as `value` (never part of it): S-genome, BASELINE config 3
    (k = 31 minimizers, window 40) and config 5 (sacCer3 index + lookups) — each with its timing, the roofline fraction of
    its dominant kernel and a parity flag.  One GPU only; the headline batch has been released by now."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fullsize
    import oracle
    import bionumpy_amd as bnp
    from bionumpy_amd.pipeline import fastq_kmer_histogram, fastq_minimizers
    out = {}

    def timed(fn, steps):
        r = fn(); del r
        torch.cuda.synchronize()
        dev.prof_enable(True); dev.prof_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
            if _ != steps - 1:
                del r
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        prof = dev.prof_report(); dev.prof_enable(False)
        return r, dt, prof

    def roof(prof, name, nbytes, steps):
        avg_ms = prof[name]["total_ms"] / max(prof[name]["launches"], 1)
        gbs = nbytes / (avg_ms * 1e-3) / 1e9
        return {"kernel": name, "avg_launch_ms": round(avg_ms, 3), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy": round(gbs / copy_rate, 4) if copy_rate else None,
                "algorithmic_bytes_per_launch": int(nbytes)}

    # ---- S-genome: the same 50 M x 150 bp, reads drawn from a 100 Mbp genome (60x coverage): duplicate-heavy k-mers
    try:
        steps = 3
        text = ops.synth_fastq(args.reads, args.read_len, args.seed, 1, args.genome_len, 0)
        (hist, stats), dt, prof = timed(lambda: fastq_kmer_histogram(text, args.k), steps)
        keys, counts = hist
        dom = max(prof, key=lambda n: prof[n]["total_ms"])
        check = fullsize.check_histogram(ops, text, args.reads, args.read_len, args.k, args.seed, 1, args.genome_len, 0, keys, counts)
        out["s_genome"] = {"workload": "synthetic %dbp x %d reads FASTQ (genome of %d bp), k=%d" % (args.read_len, args.reads, args.genome_len, args.k),
                           "ms_per_step": round(dt * 1e3, 2), "steps": steps, "gbases_per_s": round(stats.n_bases / dt / 1e9, 2),
                           "distinct": keys.size, "roofline": roof(prof, dom, algorithmic_bytes(dom, stats, args.read_len, args.k, keys.size), steps),
                           "kernels_ms": {n: round(v["total_ms"] / steps, 2) for n, v in prof.items()},
                           "parity": True, "parity_detail": check}
        del hist, keys, counts, text
    except AssertionError as e:
        out["s_genome"] = {"parity": False, "error": str(e)}
    torch.cuda.empty_cache()

    # ---- config 3: k = 31 minimizers over windows of 40 bases (w = 10 k-mers) of the headline batch's reads
    try:
        steps = 3
        text = ops.synth_fastq(args.reads, args.read_len, args.seed, 0, 0, 0)
        (mins, stats), dt, prof = timed(lambda: fastq_minimizers(text, 31, 40), steps)
        n_out = mins.size
        del mins
        # parity (outside the timing): reads at the start, in the middle and at the end — every minimizer against the
        # row-lookup kernel bnpk_minimizers, sampled reads against the numpy oracle
        rec = 2 * args.read_len + 16
        per = min(args.reads, 2_000_000)
        checked = {"minimizers": 0, "sampled_minimizers_vs_oracle": 0}
        for f in sorted({0, max(0, args.reads // 2 - per // 2), max(0, args.reads - per)}):
            from bionumpy_amd.device import HArray
            part = HArray(dev=text.dev()[f * rec:(f + per) * rec])
            c = fullsize.check_minimizers(ops, part, per, args.read_len, 31, 40, args.seed, 0, 0, first_read=f)
            checked["minimizers"] += c["minimizers"]
            checked["sampled_minimizers_vs_oracle"] += c["sampled_minimizers_vs_oracle"]
        nbytes = 8 * n_out + stats.n_bases / 4 + stats.n_bases / 8
        out["config3_minimizers"] = {"workload": "the same %d reads, k=31 minimizers, window 40" % args.reads, "ms_per_step": round(dt * 1e3, 2),
                                     "steps": steps, "gbases_per_s": round(stats.n_bases / dt / 1e9, 2), "n_minimizers": n_out,
                                     "roofline": roof(prof, "minimizers_flat", nbytes, steps),
                                     "kernels_ms": {n: round(v["total_ms"] / steps, 2) for n, v in prof.items()},
                                     "parity": True, "parity_detail": dict(checked, compared_with="bnpk_minimizers (row-lookup kernel) element for "
                                                                            "element on 3 x %d reads; the numpy oracle on sampled reads" % per)}
        # ---- the same reads through the API objects (what a user of the reference's functions runs): minimizers, a read filter
        # with write-back, a reverse-complement rewrite — parity by what must hold at any size
        try:
            def api_minimizers():
                buf = bnp.FastQBuffer.from_raw_buffer(text)
                return bnp.get_minimizers(bnp.change_encoding(buf.get_field_by_number(1), bnp.DNAEncoding), 31, 40)
            m, dt, prof = timed(api_minimizers, 2)
            fused, _ = fastq_minimizers(text, 31, 40)
            same = bool(torch.equal(m._flat_data().dev(), fused.dev()))
            del m, fused
            out["config3_api_objects"] = {"workload": "FastQBuffer.from_raw_buffer + change_encoding + get_minimizers(31, 40) on the same reads",
                                          "ms_per_step": round(dt * 1e3, 2), "steps": 2, "kernels_ms": {n: round(v["total_ms"] / 2, 2) for n, v in prof.items()},
                                          "parity": same, "parity_detail": "every minimizer == the fused pipeline's"}

            def read_filter():
                chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
                keep = np.mean(chunk.quality, axis=1) >= 0.0
                keep[::3] = False
                return chunk[keep].get_buffer().entry_bytes(), int(keep.sum())
            (kept_text, n_keep), dt, prof = timed(read_filter, 2)
            back = bnp.FastQBuffer.from_raw_buffer(kept_text)
            ok = len(back) == n_keep and n_keep == args.reads - (args.reads + 2) // 3
            out["read_filter_write_back"] = {"workload": "np.mean(chunk.quality, axis=1) mask, every third read dropped, chunk[mask] as text",
                                             "ms_per_step": round(dt * 1e3, 2), "steps": 2, "kept": n_keep, "bytes_out": int(kept_text.size),
                                             "kernels_ms": {n: round(v["total_ms"] / 2, 2) for n, v in prof.items()},
                                             "parity": bool(ok), "parity_detail": "the kept text parses back to exactly the kept entries"}
            del kept_text, back

            def rewrite():
                chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
                rc = bnp.get_reverse_complement(chunk.sequence)
                return bnp.FastQBuffer.from_data(bnp.replace(chunk, sequence=rc))
            new_text, dt, prof = timed(rewrite, 2)
            again = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(new_text))
            orig = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text)).sequence
            orig._compact()
            ok = new_text.size == text.size and bool(torch.equal(bnp.get_reverse_complement(again.sequence)._flat_data().dev(), orig._flat_data().dev()))
            out["reverse_complement_rewrite"] = {"workload": "get_reverse_complement(chunk.sequence) + the chunk's text rebuilt from its fields",
                                                 "ms_per_step": round(dt * 1e3, 2), "steps": 2, "bytes_out": int(new_text.size),
                                                 "kernels_ms": {n: round(v["total_ms"] / 2, 2) for n, v in prof.items()},
                                                 "parity": bool(ok), "parity_detail": "the rewritten text parses back; its sequences reverse-complemented are the original ones"}
            del new_text, again, orig

            # ---- results that stay on the device (round 5): which reads contain a motif, the base composition of the batch
            from bionumpy_amd import synth
            dna = bnp.change_encoding(bnp.FastQBuffer.from_raw_buffer(text).get_field_by_number(1), bnp.DNAEncoding)
            dna._compact()
            motif = "GATTACA"

            def reads_with_motif():
                return bnp.match_string(dna, motif).any(axis=-1)
            mask, dt, prof = timed(reads_with_motif, 3)
            on_device = mask.harray()._np is None
            n_hit = int(mask.sum())
            sample = 20000
            codes = synth.read_codes(sample, args.read_len, args.seed, 0, 0, 0)
            want = np.array([motif in "".join("ACGT"[c] for c in row) for row in codes])
            km7 = bnp.get_kmers(dna, 7)
            km7._compact()
            target = sum("ACGT".index(ch) << (2 * j) for j, ch in enumerate(motif))
            windows = int((km7._flat_data().dev() == target).sum().item())
            hits_total = int(bnp.match_string(dna, motif).sum(axis=None))
            ok = on_device and np.array_equal(np.asarray(mask)[:sample], want) and hits_total == windows and 0 < n_hit <= hits_total
            del km7
            out["match_string_any"] = {"workload": "match_string(reads, %r).any(axis=-1) on the same %d reads (2-bit DNA)" % (motif, args.reads),
                                       "ms_per_step": round(dt * 1e3, 2), "steps": 3, "reads_with_motif": n_hit, "windows_matching": hits_total,
                                       "flags_left_the_device": not on_device, "kernels_ms": {n: round(v["total_ms"] / 3, 2) for n, v in prof.items()},
                                       "parity": bool(ok), "parity_detail": "first %d reads against Python's `in` on the generator's twin; matching windows == 7-mer hashes equal to the motif's (another kernel)" % sample}
            del mask

            def composition():
                return bnp.count_encoded(dna, axis=None)
            comp, dt, prof = timed(composition, 3)
            from bionumpy_amd.encoded_array import packed_words
            n_part = min(int(dna.total()), 1 << 28)
            unpacked = ops.unpack_codes(packed_words(dna._data), n_part, to_ascii=False)
            ref = torch.bincount(unpacked.dev().to(torch.int32), minlength=4).cpu().numpy()
            part = np.asarray(ops.count_bytes(unpacked, 4).host())                                   # (bytes: the other kernel)
            whole = np.asarray(comp.counts)
            ok = int(whole.sum()) == int(dna.total()) and np.array_equal(part, ref) and (n_part < dna.total() or np.array_equal(whole, ref))
            del unpacked
            out["letter_counts"] = {"workload": "count_encoded(reads, axis=None): the base composition of %d bases, from the 2-bit words" % int(dna.total()),
                                    "ms_per_step": round(dt * 1e3, 2), "steps": 3, "counts": [int(x) for x in comp.counts],
                                    "kernels_ms": {n: round(v["total_ms"] / 3, 2) for n, v in prof.items()},
                                    "roofline": roof(prof, "count_packed2", dna.total() / 4, 3),
                                    "parity": bool(ok), "parity_detail": "counts add up to the bases; the byte kernel on the first 2^28 unpacked codes == torch.bincount"}
            del dna
        except AssertionError as e:
            out["config3_api_objects"] = {"parity": False, "error": str(e)}
        del text
    except AssertionError as e:
        out["config3_minimizers"] = {"parity": False, "error": str(e)}
    torch.cuda.empty_cache()

    # ---- (f2) multi-line FASTA at speed: a 1 GB synthetic genome (60-letter lines, records of ~8 Mbp) read through bnp.open
    # (file -> pinned RAM -> HBM -> bnpk_multiline_* -> sequences), then cached as a MemMapEncodedRaggedArray and loaded back
    try:
        import tempfile
        rng = np.random.default_rng(11)
        n_rec, per_rec, width = 128, 8_000_040, 60
        lines = per_rec // width
        body = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(lines, width), dtype=np.int8)]
        rec = np.concatenate([body, np.full((lines, 1), 10, dtype=np.uint8)], axis=1).reshape(-1)
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        fa = os.path.join(tmpdir, "bnpk_bench_%d.fa" % os.getpid())
        with open(fa, "wb") as f:
            for i in range(n_rec):                            # (the same body under every header: the decode does not care)
                f.write(b">chr%d synthetic\n" % i)
                f.write(rec.data)
        fa_bytes = os.path.getsize(fa)

        def read_all():
            n_b = n_r = 0
            for chunk in bnp.open(fa).read_chunks(min_chunk_size=256 << 20):
                seq = chunk.sequence
                n_b += int(seq.total()); n_r += len(seq)
            return n_b, n_r
        read_all()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_b, n_r = read_all()
        torch.cuda.synchronize()
        t_read = time.perf_counter() - t0
        # parity: the rows of one chunk against the bytes they were written from
        first = next(iter(bnp.open(fa).read_chunks(min_chunk_size=32 << 20)))
        row0 = np.asarray(first.sequence[0].raw())
        ok = n_r == n_rec and n_b == n_rec * lines * width and row0.size == lines * width and np.array_equal(row0, body.reshape(-1))
        base = os.path.join(tmpdir, "bnpk_bench_%d_mm" % os.getpid())
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            bnp.MemMapEncodedRaggedArray.create(lambda: (bnp.change_encoding(c.sequence, bnp.DNAEncoding) for c in bnp.open(fa).read_chunks(min_chunk_size=256 << 20)), base)
            t_create = time.perf_counter() - t0
        t0 = time.perf_counter()
        loaded = bnp.MemMapEncodedRaggedArray.load(base)
        km = bnp.count_kmers(loaded, 5)
        torch.cuda.synchronize()
        t_load = time.perf_counter() - t0
        ok = ok and int(np.sum(km.counts)) == n_rec * (lines * width - 4)
        for suffix in ("_data.dat", "_lengths.dat", "_encoding.pkl"):
            if os.path.exists(base + suffix):
                os.remove(base + suffix)
        os.remove(fa)
        out["multiline_fasta"] = {"workload": "synthetic multi-line FASTA, %d records x %d bases in %d-letter lines (%.2f GB)" % (n_rec, lines * width, width, fa_bytes / 1e9),
                                  "read_decode_s": round(t_read, 3), "file_gb_per_s": round(fa_bytes / t_read / 1e9, 2),
                                  "memmap_create_s": round(t_create, 2), "memmap_load_and_count_5mers_s": round(t_load, 3),
                                  "parity": bool(ok), "parity_detail": "records / bases of all chunks, the first row byte for byte, the 5-mers of the loaded cache"}
    except Exception as e:                                   # noqa: BLE001
        out["multiline_fasta"] = {"parity": False if isinstance(e, AssertionError) else None, "error": "%s: %s" % (type(e).__name__, e)}
    torch.cuda.empty_cache()

    # ---- config 5: sacCer3 k = 31 KmerIndex build + lookup of every 31-mer of big.fq.gz (tests/golden/)
    try:
        gold = os.path.join(ROOT, "tests", "golden")
        genome = bnp.open(os.path.join(gold, "sacCer3.fa.gz")).read()
        seqs = bnp.change_encoding(genome.sequence, bnp.DNAEncoding)
        index = bnp.KmerIndex.create_index(seqs, k=31)
        torch.cuda.synchronize()
        dev.prof_enable(True); dev.prof_reset()
        t0 = time.perf_counter()
        for _ in range(3):
            index = bnp.KmerIndex.create_index(seqs, k=31)
        torch.cuda.synchronize()
        t_build = (time.perf_counter() - t0) / 3
        idx_prof = dev.prof_report(); dev.prof_enable(False)
        idx_kernels = {n: round(v["total_ms"] / 3, 3) for n, v in idx_prof.items()}
        q0_pairs = int(seqs.total()) - 30 * len(seqs)          # (k-mer, row) occurrences that go into the index
        reads_fq = bnp.open(os.path.join(gold, "big.fq.gz")).read()
        q = bnp.get_kmers(bnp.change_encoding(reads_fq.sequence, bnp.DNAEncoding), 31)
        q._compact()
        lo, hi = index.get_indices_batch(q._flat_data())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            lo, hi = index.get_indices_batch(q._flat_data())
        torch.cuda.synchronize()
        t_lookup = (time.perf_counter() - t0) / 3
        # parity: the whole index pair for pair and every lookup against the numpy oracle
        raw, res = oracle.open_text(os.path.join(gold, "sacCer3.fa.gz")).read()
        codes = oracle.encode_dna(oracle.gather_rows(raw, res.line_starts, res.line_lens))
        eh, er = oracle.kmer_index_pairs(codes, res.seq_lens, 31)
        qh = q._flat_data().host()
        ok = (np.array_equal(index._keys.host(), eh) and np.array_equal(index._rows.host(), er)
              and np.array_equal(lo.host(), np.searchsorted(eh, qh, side="left")) and np.array_equal(hi.host(), np.searchsorted(eh, qh, side="right")))
        assert ok, "config 5: index or lookups differ from the oracle"
        out["config5_kmer_index"] = {"workload": "sacCer3.fa.gz (%d bases) k=31 KmerIndex + %d lookups (big.fq.gz)" % (int(seqs.total()), qh.size),
                                     "build_ms": round(t_build * 1e3, 2), "lookup_ms": round(t_lookup * 1e3, 3), "index_pairs": int(eh.size),
                                     "kernels_ms": idx_kernels, "roofline": _index_roofline(idx_kernels, int(q0_pairs), int(eh.size), t_build * 1e3), "parity": True,
                                     "parity_detail": "all (kmer, row) pairs and all lookups == oracle.kmer_index_pairs / np.searchsorted"}
        # ---- the same call at a size where the kernels, not the launches, set the time: 10^9 random 31-mers in 100 rows ----------
        try:
            if args.reads < 10_000_000:                      # (a small smoke run of bench.py: not the place for a 10^9-pair build)
                raise RuntimeError("skipped below --reads 10000000")
            from bionumpy_amd.device import HArray
            del index, genome, seqs, q, lo, hi
            torch.cuda.empty_cache()
            n_big, rows_big = 1_000_000_000, 100
            g = torch.Generator(device="cuda"); g.manual_seed(11)
            big_k = torch.randint(0, 1 << 62, (n_big,), dtype=torch.int64, device="cuda", generator=g)
            m3 = n_big // 3
            big_k[0:3 * m3:3] = big_k[1:3 * m3:3].clone()       # a third of the k-mers occur twice (next to each other: mostly in one row)
            big_r = (torch.arange(n_big, dtype=torch.int64, device="cuda") * rows_big) // n_big
            r = ops.unique_pairs(HArray(dev=big_k), HArray(dev=big_r), key_bits=62, n_values=rows_big); del r
            torch.cuda.synchronize()
            dev.prof_reset(); dev.prof_enable(True)
            t0 = time.perf_counter()
            pk, pr = ops.unique_pairs(HArray(dev=big_k), HArray(dev=big_r), key_bits=62, n_values=rows_big)
            torch.cuda.synchronize()
            big_ms = (time.perf_counter() - t0) * 1e3
            prof_big = dev.prof_report(); dev.prof_enable(False)
            kd, rd = pk.dev(), pr.dev()
            # parity by properties: (k-mer, row) strictly increasing; sampled input k-mers are found; a 50 M-pair part == torch.unique
            inc = bool(((kd[1:] > kd[:-1]) | ((kd[1:] == kd[:-1]) & (rd[1:] > rd[:-1]))).all().item())
            probe = torch.randint(0, n_big, (2_000_000,), device="cuda", generator=g)
            pos = torch.searchsorted(kd, big_k[probe]).clamp(max=kd.numel() - 1)
            found = bool((kd[pos] == big_k[probe]).all().item())
            part_k, part_r = big_k[:50_000_000].clone(), big_r[:50_000_000].clone()
            want = int(torch.unique(part_k * rows_big + part_r).numel()) if False else int(torch.unique(torch.stack([part_k, part_r]), dim=1).shape[1])
            sub = ops.unique_pairs(HArray(dev=part_k), HArray(dev=part_r), key_bits=62, n_values=rows_big)
            kernels_big = {name: round(p["total_ms"], 2) for name, p in prof_big.items()}
            out["config5_kmer_index"]["synthetic_1e9_pairs"] = {
                "workload": "%d random 31-mers (a third of them twice) in %d rows: bnpk_index_build" % (n_big, rows_big),
                "build_ms": round(big_ms, 1), "distinct_pairs": int(kd.numel()), "kernels_ms": kernels_big,
                "roofline": _index_roofline(kernels_big, n_big, int(kd.numel()), big_ms,
                                            "whole build: one partition of (k-mer, row) words (first level by the fixed-line scatter, the "
                                            "claiming level, a finishing kernel, the decode of the k-mers' top bits); the same pairs by "
                                            "ranks (option index_pairs 0) take 213 ms, 153 of them random look-ups"),
                "parity": bool(inc and found and sub[0].size == want),
                "parity_detail": "(k-mer, row) strictly increasing; 2 M sampled input k-mers found; the first 50 M pairs: as many distinct as torch.unique"}
            del big_k, big_r, pk, pr, kd, rd, sub, part_k, part_r
        except Exception as e:                                # noqa: BLE001
            out["config5_kmer_index"]["synthetic_1e9_pairs"] = {"parity": None, "error": "%s: %s" % (type(e).__name__, e)}
    except AssertionError as e:
        out["config5_kmer_index"] = {"parity": False, "error": str(e)}
    except Exception as e:                                   # noqa: BLE001  (e.g. the fixtures are not there)
        out["config5_kmer_index"] = {"parity": None, "error": "%s: %s" % (type(e).__name__, e)}
    return out


def virtual_ranks_mode(args, ops, dev, mode):
    """--virtual-ranks N: shard -> k-mers partitioned by the send cuts -> (the exchange's result) -> every rank's key range
    counted; the concatenation is checked against the k-mers of all reads (tests/fullsize.py) and one JSON line is printed"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fullsize
    from bionumpy_amd.device import HArray
    from bionumpy_amd.pipeline import fastq_kmer_histogram_virtual_ranks
    n = args.virtual_ranks
    per = -(-args.reads // n)
    texts = [ops.synth_fastq(min(per, args.reads - r * per), args.read_len, args.seed, mode, args.genome_len, r * per)
             for r in range(n)]
    for _ in range(args.warmup):
        fastq_kmer_histogram_virtual_ranks(texts, args.k, canonical=args.canonical)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hists, stats, received, plan = fastq_kmer_histogram_virtual_ranks(texts, args.k, canonical=args.canonical, with_plan=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    sums = [0, 0, 0, 0]
    for keys, counts in hists:
        sums = [(a + b) & ((1 << 64) - 1) if i else a + b for i, (a, b) in enumerate(zip(sums, fullsize.histogram_sums(keys, counts)))]
    flat = [0, 0, 0, 0]
    for text, st in zip(texts, stats):
        flat = [(a + b) & ((1 << 64) - 1) if i else a + b
                for i, (a, b) in enumerate(zip(flat, fullsize.reads_sums(ops, text, st.n_reads, args.read_len, args.k, args.canonical)))]
    assert sums == flat, "virtual ranks: checksums differ %s vs %s" % (sums, flat)
    bounds = [(int(k.dev()[0]), int(k.dev()[-1])) for k, _ in hists if k.size]
    assert all(a[1] < b[0] for a, b in zip(bounds[:-1], bounds[1:])), "key ranges of the ranks overlap"
    print(json.dumps({"mode": "virtual ranks (one GPU plays N ranks; functional, not a scaling number)", "virtual_ranks": n,
                      "reads_total": args.reads, "k": args.k, "ms_per_step": round(dt * 1e3, 2),
                      "gbases_per_s_one_gpu_doing_all_ranks": round(sum(s.n_bases for s in stats) / dt / 1e9, 3),
                      "plan": plan, "int64_words_received_per_rank": received, "distinct_per_rank": [k.size for k, _ in hists],
                      "parity": "count / sum / sum of squares / mixed sum of all ranks' (key, count) == k-mers in read order; "
                                "key ranges disjoint and ascending"}))


def from_file_mode(args, ops, dev, mode, rank, world):
    """--from-file PATH: BASELINE config 4's shape from an actual FASTQ file.  Every rank calls the reference's stream form

        count_kmers(bnp.open(PATH, shard="auto").read_chunks().sequence, k)

    and nothing else: under torch.distributed ``shard="auto"`` gives every rank its part of the file (byte ranges cut at record
    starts, io/sharding.py) and the reduction finishes with the merge over the ranks.  The file is written first (rank 0,
    outside the timed region: world x --reads synthetic reads) unless it exists.  Exits 4 if the ranks' read counts do not add
    up to the reads of the file, and checks the merged histogram against the k-mers of all parts in read order (checksums,
    all-reduced).  One JSON line; host-fed by construction (file -> pinned RAM -> HBM), so never the headline `value`."""
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fullsize
    import bionumpy_amd as bnp
    path = args.from_file
    total_reads = world * args.reads
    if rank == 0 and not os.path.exists(path):
        piece = 4_000_000
        with open(path + ".tmp", "wb") as f:
            for first in range(0, total_reads, piece):
                t = ops.synth_fastq(min(piece, total_reads - first), args.read_len, args.seed, mode, args.genome_len, first)
                f.write(t.host().tobytes())
                del t
        os.replace(path + ".tmp", path)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    file_bytes = os.path.getsize(path)
    chunk = args.file_chunk_mb << 20

    def step():
        return bnp.count_kmers(bnp.open(path, shard="auto").read_chunks(min_chunk_size=chunk).sequence, args.k)

    for _ in range(args.warmup):
        h = step(); h._keys; del h
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h = step()
        h._keys                                              # (sparse counts are lazy: the timed region ends with a counted result)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # ---- what was read and counted, outside the timed region -------------------------------------------------------
    n_reads = n_bases = 0
    flat = [0, 0, 0, 0]
    for c in bnp.open(path, shard="auto").read_chunks(min_chunk_size=chunk):
        n_reads += len(c)
        seqs = bnp.change_encoding(c.sequence, bnp.DNAEncoding)
        n_bases += int(seqs.total())
        km = bnp.get_kmers(seqs, args.k)
        km._compact()
        hd = km._flat_data().dev()
        s = fullsize._sums(torch, hd)
        flat = [flat[0] + hd.numel()] + [(a + b) & ((1 << 64) - 1) for a, b in zip(flat[1:], s)]
        del km, hd, seqs
    hs = fullsize.histogram_sums(h._keys, h._counts) if isinstance(h, bnp.SparseKmerCounts) else None
    wrap = lambda v: [x - (1 << 64) if x >= (1 << 63) else x for x in v]
    both = torch.tensor([wrap(hs) if hs is not None else [0] * 4, wrap(flat), [n_reads, n_bases, len(h) if hs is not None else 0, 0]],
                        dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(both)
    reads_seen = int(both[2][0].item())
    ok_reads = reads_seen * fullsize.synth.record_bytes(args.read_len) == file_bytes if args.from_file_synthetic else True
    parity = hs is None or bool((both[0] == both[1]).all().item())
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        from bionumpy_amd import parallel
        print(json.dumps({
            "mode": "from-file (count_kmers(bnp.open(f, shard='auto').read_chunks().sequence, k) on every rank; file -> pinned RAM -> HBM: never the headline value)",
            "file": path, "file_bytes": file_bytes, "n_gpus": world, "k": args.k, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(float(tmax.item()) * 1e3, 2),
            "gbases_per_s": round(int(both[2][1].item()) / float(tmax.item()) / 1e9, 3),
            "file_gb_per_s": round(file_bytes / float(tmax.item()) / 1e9, 2),
            "reads_seen_by_all_ranks": reads_seen, "reads_add_up": bool(ok_reads), "distinct_keys": int(both[2][2].item()),
            "collectives": parallel.collectives().name if world > 1 else None,
            "parity_fullsize": bool(parity),
            "parity": "count / sum / sum of squares / mixed sum of all ranks' (key, count) == those of the k-mers of every part in read order (all-reduced)"}))
    if world > 1:
        dist.destroy_process_group()
    if not ok_reads or not parity:
        sys.exit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--mode", choices=["uniform", "genome"], default="uniform")
    ap.add_argument("--genome-len", type=int, default=100_000_000)
    ap.add_argument("--seed", type=int, default=20260925)
    ap.add_argument("--cpu-sample-reads", type=int, default=400_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="play the N-GPU sparse path on this one GPU: the batch is sharded N ways, the exchange is replaced "
                         "by its result, every rank's key range is counted (functional check of the N > 1 kernels, "
                         "not a scaling measurement)")
    ap.add_argument("--from-file", default=None, metavar="PATH",
                    help="count the k-mers of this FASTQ file through the reader, every rank its part (written first — world x "
                         "--reads synthetic reads — if it does not exist); exits 4 if the ranks' reads do not add up")
    ap.add_argument("--file-chunk-mb", type=int, default=256, help="--from-file: chunk size the reader is asked for")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (pinned RAM -> HBM) measurement")
    ap.add_argument("--host-fed-batches", type=int, default=2)
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: accept torch.distributed's collectives if the C-ABI (RCCL) communicator cannot be made; without "
                         "this flag such a run FAILS instead of reporting a number measured on another path")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the secondary workloads (S-genome, config 3, config 5) that follow the timed region of `value`")
    ap.add_argument("--verify", action="store_true", help="check the first reads against the oracle")
    ap.add_argument("--canonical", action="store_true",
                    help="count strand-independent k-mers min(h, rc(h)) (extension; not the headline workload)")
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, "RCCL sees %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus)

    from bionumpy_amd.device import Device, HArray
    from bionumpy_amd.ops import get_ops
    from bionumpy_amd.pipeline import fastq_kmer_histogram
    ops = get_ops()
    dev = Device.get()
    if os.environ.get("BNPK_FINISH_MODE"):                   # (experiments: force a finishing path, include/bnpk.h "finish_mode")
        from bionumpy_amd._native import lib as _lib
        assert _lib.bnpk_set_option(dev.ctx, b"finish_mode", int(os.environ["BNPK_FINISH_MODE"])) == 0
    mode = 0 if args.mode == "uniform" else 1

    if args.from_file:
        args.from_file_synthetic = not os.path.exists(args.from_file)       # (a file we write ourselves: its size says how many reads)
        from_file_mode(args, ops, dev, mode, rank, world)
        return

    # ---- input: generated on the device, resident in HBM before the timed region ------------------------
    text = ops.synth_fastq(args.reads, args.read_len, args.seed, mode, args.genome_len, first_read=rank * args.reads)
    dev.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.virtual_ranks > 1:
        virtual_ranks_mode(args, ops, dev, mode)
        return

    def step():
        hist, stats = fastq_kmer_histogram(text, args.k, canonical=args.canonical)
        return hist, stats

    stats = None
    for _ in range(args.warmup):
        hist, stats = step()
        del hist
    barrier()
    dev.prof_enable(True)
    dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hist, stats = step()
        if _ != args.steps - 1:
            del hist
    barrier()
    dt = time.perf_counter() - t0
    dev.prof_enable(False)
    prof = dev.prof_report()

    # ---- parity of the FULL-SIZE result of the last step (outside the timed region; tests/fullsize.py) ------------
    # sparse: order-independent checksums of the whole histogram against the k-mers in read order (another kernel),
    # plus the k-mers of reads sampled at the start / middle / end computed by the numpy oracle and looked up in the keys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fullsize
    parity = None
    if isinstance(hist, tuple):
        keys, counts = hist
        n_distinct = keys.size
        hs = fullsize.histogram_sums(keys, counts)
        rs = fullsize.reads_sums(ops, text, args.reads, args.read_len, args.k, args.canonical)
        both = torch.tensor([[x - (1 << 64) if x >= (1 << 63) else x for x in hs],
                             [x - (1 << 64) if x >= (1 << 63) else x for x in rs]], dtype=torch.int64, device="cuda")
        if world > 1:
            dist.all_reduce(both)                    # range-partitioned histogram: the sums must match globally (mod 2^64)
        assert bool((both[0] == both[1]).all().item()), "full-size parity: checksums differ %s" % both.tolist()
        sampled = 0
        if world == 1:
            sampled = fullsize.sampled_reads_check(ops, keys, counts, args.reads, args.read_len, args.k, args.seed, mode,
                                                   args.genome_len, rank * args.reads, args.canonical)
        parity = {"ok": True, "kmers_checked": int(both[1][0].item()), "sampled_kmers_vs_oracle": sampled,
                  "checks": "count, sum, sum of squares, sum of mixed hashes (mod 2^64) of (key, count) == the same over "
                            "bnpk_windows_flat in read order; keys strictly increasing; sampled reads' k-mers from the "
                            "numpy oracle found with equal counts"}
    else:
        total_counted = int(hist.dev().sum().item())
        n_distinct = int((hist.dev() > 0).sum().item())
        counted = torch.tensor([total_counted, stats.n_kmers], dtype=torch.int64, device="cuda")
        if world == 1:
            assert int(counted[0]) == int(counted[1]), "histogram does not account for every k-mer: %s" % counted
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    per_rank = None
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        mine = torch.tensor([n_distinct, stats.n_kmers], dtype=torch.int64, device="cuda")
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = {"distinct_keys_held": [int(g[0]) for g in gathered], "kmers_generated": [int(g[1]) for g in gathered]}
    dt = float(tmax.item())

    if world > 1:
        # what ran must be what is claimed: N ranks over the library's own RCCL communicator (every rank checks; the
        # decision to fall back is collective, so they all agree)
        from bionumpy_amd import parallel as _par
        used = _par.collectives().name
        if not used.startswith("bnpk C-ABI") and not args.allow_fallback:
            sys.stderr.write("bench.py: rank %d ran its collectives over %r, not the C-ABI communicator (--allow-fallback accepts that)\n"
                             % (rank, used))
            dist.destroy_process_group()
            sys.exit(3)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    if args.verify:
        import oracle
        m = min(args.reads, 2000)
        sample = text.dev()[:m * (2 * args.read_len + 16)].cpu().numpy()
        sub, _ = fastq_kmer_histogram(HArray(host=sample), args.k, canonical=args.canonical)
        res = oracle.scan_one_line_buffer(sample, oracle.FASTQ)
        codes = oracle.encode_dna(oracle.gather_rows(sample, res.field_starts[:, 1], res.field_lens[:, 1]))
        h, _ = oracle.get_kmers(codes, res.field_lens[:, 1], args.k)
        ek, ec = oracle.count_sparse(oracle.canonical_kmers(h, args.k) if args.canonical else h)
        assert np.array_equal(sub[0].host(), ek) and np.array_equal(sub[1].host(), ec), "verify failed"

    from bionumpy_amd import parallel
    merge = parallel.last
    gbases = world * stats.n_bases * args.steps / dt / 1e9
    # dominant kernel and its roofline
    dom = max(prof, key=lambda kname: prof[kname]["total_ms"]) if prof else None
    roofline = None
    kernels = {}
    for name, p in prof.items():
        avg_ms = p["total_ms"] / max(p["launches"], 1)
        b = algorithmic_bytes(name, stats, args.read_len, args.k, n_distinct)
        kernels[name] = {"ms_per_step": round(p["total_ms"] / args.steps, 3), "launches_per_step": p["launches"] / args.steps,
                         "gbs": None if not b or avg_ms <= 0 else round(b / (avg_ms * 1e-3) / 1e9, 1)}
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the run, so they come from the committed
    # rocprofv3 --pmc passes of this same command (scripts/profile_r04.sh: FETCH_SIZE / WRITE_SIZE in separate passes,
    # FETCH x2 as MI355X_MICROARCH.md prescribes) — and only if that profile was taken from THIS build (hash of the kernel
    # sources) on this workload; otherwise null
    traffic, traffic_source = None, None
    try:
        from bionumpy_amd.csrc.build import _source_hash
        for cand in (("r06_genome_pmc.json", "r05_genome_pmc.json", "r04_genome_pmc.json", "r03_genome_pmc.json") if args.mode == "genome" else ("r06_pmc.json", "r06_k21_pmc.json", "r05_pmc.json", "r05_k21_pmc.json", "r04_pmc.json", "r03_pmc.json")) + ("r02_pmc.json", "r01_pmc.json"):
            path = os.path.join(ROOT, "profiles", cand)
            if not os.path.exists(path):
                continue
            pmc = json.load(open(path))
            cfg = pmc.get("_config", {})
            same_work = (cfg.get("reads_per_gpu"), cfg.get("read_len"), cfg.get("k"), cfg.get("mode"), bool(cfg.get("canonical", False))) == \
                (args.reads, args.read_len, args.k, args.mode, bool(args.canonical)) and world == 1
            if not same_work or pmc.get("_source_hash") != _source_hash():
                continue
            ms_of = lambda n: prof.get(n, {}).get("total_ms", 0.0)
            finisher = "finish_wave" if args.mode == "genome" else "finish_multi" if ms_of("finish.multi") > ms_of("finish.fast") else "finish_fast"
            names = {"finish_sorted": finisher, "radix_scatter": "rp_scatter<mem_source>",
                     "radix_scatter_claimed": "rp_scatter<mem_source, claiming>",
                     "kmers_partition_scatter": "rp_scatter<kmer_source>", "radix_hist": "rp_hist<mem_source>",
                     "fastq_encode": "fq_encode_fast", "fastq_census": "fq_census_fast"}
            rec = pmc.get(names.get(dom, dom)) or pmc.get(dom)
            if rec and "read_bytes_corrected" in rec and "write_bytes" in rec:
                traffic = int(rec["read_bytes_corrected"] + rec["write_bytes"])
                traffic_source = "profiles/%s (source hash %s)" % (cand, pmc["_source_hash"][:12])
                break
    except Exception:
        traffic = None
    measured_peak, copy_rates = None, None
    try:                                                     # what this chip streams at (device copy, read + write bytes)
        import ctypes as C
        from bionumpy_amd._native import lib
        src = torch.empty(1 << 31, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        rates = (C.c_double * 4)()
        if lib.bnpk_copy_rates(dev.ctx, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel(), 5, rates, None) == 0:
            copy_rates = dict(zip(("nontemporal_loop", "plain_loop", "one_float4_per_thread", "four_per_iteration"),
                                  (round(r, 1) for r in rates)))
            measured_peak = max(copy_rates.values())         # the fastest form is what this chip copies at
        del src, dst
    except Exception:
        measured_peak = None
    if dom is not None:
        p = prof[dom]
        avg_ms = p["total_ms"] / max(p["launches"], 1)
        b = algorithmic_bytes(dom, stats, args.read_len, args.k, n_distinct)
        achieved = b / (avg_ms * 1e-3) / 1e9 if b else None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": None if achieved is None else round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": round(avg_ms, 3),
                    "traffic_box": None if traffic is None else "builder (a committed rocprofv3 --pmc profile of this build and workload, not this run)",
                    "measured_copy_gb_per_s": measured_peak, "copy_forms_gb_per_s": copy_rates,
                    "guide_copy_gb_per_s": 6290.0,           # MI355X_MICROARCH.md: float4 copy, 79 % of the spec peak
                    "frac_of_measured_copy": None if (achieved is None or not measured_peak) else round(achieved / measured_peak, 4),
                    "algorithmic_bytes_per_launch": b}
    out = {
        "metric": "Gbases/s FASTQ->k-mer count (k=%d)" % args.k,
        "value": round(gbases, 4), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "synthetic %dbp x %d reads/GPU FASTQ (%s), k=%d get_kmers+count on %dxMI355X"
                               % (args.read_len, args.reads, args.mode, args.k, world),
                   "reads_per_gpu": args.reads, "read_len": args.read_len, "k": args.k, "mode": args.mode,
                   "canonical": bool(args.canonical), "kmers_per_gpu": stats.n_kmers, "distinct_rank0": n_distinct,
                   "histogram": "dense" if args.k <= 13 else "sparse (sorted unique int64 keys + counts)",
                   "parallelism": "chunk-sharded x%d%s" % (world, ", key-range exchange of %s over %s" % (
                       {"keys": "raw hashes in %s steps overlapped with the counting" % merge.get("groups"),
                        "counts": "(key, count) runs of the local histograms"}.get(merge["plan"], "dense bins"),
                       merge["collectives"]) if world > 1 else "")},
        "roofline": roofline,
        "kernels": kernels,
        "per_rank": per_rank,
        "parity_fullsize": bool(parity and parity["ok"]),
        "parity": parity,
        # what bnpk_count_sparse (ONE C-ABI call since round 6: csrc/sparse.hip) reported about the last step's histogram —
        # path 1 = claiming level + strided finish, 2 = plain levels, 3 = library sort; host round trips inside the call
        "planner": getattr(ops, "last_sparse_info", None),
    }
    # ---- the same workload fed from page-locked host memory (never `value`): H2D on its own stream, overlapped ----------
    out["host_fed"] = None
    if world == 1 and not args.no_host_fed and args.k > 13 and isinstance(hist, tuple):
        hist = keys = counts = None                          # (the resident result: 96 GB that the next batches need)
        torch.cuda.empty_cache()
        try:
            out["host_fed"] = host_fed_leg(args, text, rs)
        except AssertionError:                               # a histogram that differs from the resident one fails the run
            raise
        except (RuntimeError, MemoryError, OSError) as e:    # (e.g. the box cannot page-lock 16 GB)
            out["host_fed"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world == 1 and not args.no_cpu_baseline:
        m = min(args.reads, args.cpu_sample_reads)
        sample = text.dev()[:m * (2 * args.read_len + 16)].cpu().numpy()
        out["cpu_baseline"] = cpu_baseline(sample, args.k)
        try:                                                 # the same sample over all host cores, one process per core
            out["cpu_baseline"]["all_cores"] = cpu_baseline_sharded(sample, 2 * args.read_len + 16, args.k)
        except Exception as e:
            out["cpu_baseline"]["all_cores"] = {"error": "%s: %s" % (type(e).__name__, e)}
    else:
        out["cpu_baseline"] = None
    # ---- the secondary workloads, under the same clock (never `value`) ---------------------------------------------------
    out["extra"] = None
    if world == 1 and not args.no_extra and args.k > 13 and args.mode == "uniform" and not args.canonical:
        hist = keys = counts = None
        del text
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        out["extra"] = extras(args, ops, dev, stats, measured_peak)
        out["extra"]["seconds"] = round(time.perf_counter() - t0, 1)
        out["extra_keys"] = sorted(k for k in out["extra"] if k != "seconds")
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
