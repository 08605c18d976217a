"""Measure the non-headline BASELINE.json configs on one MI355X and print one JSON object.

  config 3: 150 bp x N reads, k=31 minimizers (window_size 40 == w=10 k-mers)
  config 5: sacCer3.fa.gz KmerIndex build (k=31) + lookup of every 31-mer of big.fq.gz
  stream  : bnp.open(<synthetic .fq file>).read_chunks(256 MB) -> count_kmers(k=31), pinned H2D path included
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bionumpy_amd as bnp
from bionumpy_amd import synth
from bionumpy_amd.device import Device, HArray
from bionumpy_amd.ops import get_ops

GOLD = os.path.join(ROOT, "tests", "golden")
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ops = get_ops()
dev = Device.get()
out = {}


def sync():
    torch.cuda.synchronize()


# ---- config 3: minimizers ----------------------------------------------------------------------------------
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
sync()
def minimizer_step():
    buf = bnp.FastQBuffer.from_raw_buffer(text)
    seqs = bnp.change_encoding(buf.get_field_by_number(1), bnp.DNAEncoding)
    m = bnp.get_minimizers(seqs, 31, 40)
    return m
m = minimizer_step(); n_out = m.total(); del m
sync(); dev.prof_enable(True); dev.prof_reset()
t0 = time.perf_counter()
for _ in range(2):
    m = minimizer_step(); del m
sync(); dt = (time.perf_counter() - t0) / 2
prof = dev.prof_report(); dev.prof_enable(False)
out["config3_minimizers"] = {"reads": reads, "k": 31, "window_size": 40, "n_minimizers": n_out,
                             "ms_per_step": round(dt * 1e3, 2), "gbases_per_s": round(reads * 150 / dt / 1e9, 2),
                             "kernels_ms": {name: round(v["total_ms"] / 2, 2) for name, v in prof.items()},
                             "minimizers_kernel_ms": round(prof["minimizers_flat"]["total_ms"] / 2, 2),
                             "minimizers_kernel_gbs": round((8 * n_out + reads * 150 / 4 + reads * 150 / 8) /
                                                            (prof["minimizers_flat"]["total_ms"] / 2 * 1e-3) / 1e9, 1)}
from bionumpy_amd.pipeline import fastq_minimizers
m, st = fastq_minimizers(text, 31, 40); assert st.n_kmers == n_out; del m
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fullsize                                            # full-size parity of config 3 (outside the timed regions)
out["config3_parity"] = fullsize.check_minimizers(ops, text, reads, 150, 31, 40, 20260925, 0, 0)
sync(); dev.prof_enable(True); dev.prof_reset()
t0 = time.perf_counter()
for _ in range(2):
    m, st = fastq_minimizers(text, 31, 40); del m
sync(); dt = (time.perf_counter() - t0) / 2
prof = dev.prof_report(); dev.prof_enable(False)
out["config3_minimizers_fused_pipeline"] = {"reads": reads, "k": 31, "window_size": 40, "n_minimizers": st.n_kmers,
                                            "ms_per_step": round(dt * 1e3, 2),
                                            "gbases_per_s": round(reads * 150 / dt / 1e9, 2),
                                            "kernels_ms": {name: round(v["total_ms"] / 2, 2) for name, v in prof.items()}}
del text

# ---- widening: reverse complement of the reads, quality filter + compaction of whole records ------------
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
buf = bnp.FastQBuffer.from_raw_buffer(text)
seqs = bnp.change_encoding(buf.get_field_by_number(1), bnp.DNAEncoding)
seqs._compact(); sync()
def rc_step():
    return bnp.sequence.get_reverse_complement(seqs)
r = rc_step(); del r; sync(); dev.prof_enable(True); dev.prof_reset()
t0 = time.perf_counter()
for _ in range(2):
    r = rc_step(); del r
sync(); dt = (time.perf_counter() - t0) / 2
prof = dev.prof_report(); dev.prof_enable(False)
out["reverse_complement_packed"] = {"reads": reads, "ms_per_step": round(dt * 1e3, 2),
                                    "kernel_ms": round(prof["reverse_complement_packed"]["total_ms"] / 2, 2),
                                    "kernel_gbs": round(2 * reads * 150 / 4 / (prof["reverse_complement_packed"]["total_ms"] / 2 * 1e-3) / 1e9, 1)}
def match_step():
    return bnp.match_string(seqs, "GATTACA")
def match_kernel_only():
    off, n_out = ops.row_offsets(seqs._lens, 7)
    return ops.match_windows(bnp.encoded_array.packed_words(seqs._data), seqs.offsets(), len(seqs), seqs.total(), n_out,
                             [2, 0, 3, 3, 0, 1, 0], True)
h = match_kernel_only(); n_hits = int(h.dev().sum().item()); del h; sync(); dev.prof_enable(True); dev.prof_reset()
for _ in range(2):
    h = match_kernel_only(); del h
sync(); prof = dev.prof_report(); dev.prof_enable(False)
out["match_string_packed"] = {"reads": reads, "pattern": "GATTACA", "hits": n_hits,
                              "kernels_ms": {name: round(v["total_ms"] / 2, 2) for name, v in prof.items()}}
def match_rows_only():
    return ops.match_rows(bnp.encoded_array.packed_words(seqs._data), seqs.offsets(), len(seqs), seqs.total(), [2, 0, 3, 3, 0, 1, 0])
h = match_rows_only(); assert int(h.dev().sum().item()) == n_hits; del h; sync(); dev.prof_enable(True); dev.prof_reset()
for _ in range(2):
    h = match_rows_only(); del h
sync(); prof = dev.prof_report(); dev.prof_enable(False)
out["match_string_packed"]["per_row_counts_ms"] = {name: round(v["total_ms"] / 2, 2) for name, v in prof.items()}
_rng = np.random.default_rng(1)
_m = np.log(_rng.dirichlet(np.ones(4), size=12).T / 0.25)
def pwm_kernel_only():
    off, n_out = ops.row_offsets(seqs._lens, 12)
    return ops.pwm_scores(bnp.encoded_array.packed_words(seqs._data), seqs.offsets(), len(seqs), seqs.total(), n_out, _m)
h = pwm_kernel_only(); del h; sync(); dev.prof_enable(True); dev.prof_reset()
for _ in range(2):
    h = pwm_kernel_only(); del h
sync(); prof = dev.prof_report(); dev.prof_enable(False)
out["pwm_scores_12"] = {"reads": reads, "width": 12,
                        "kernels_ms": {name: round(v["total_ms"] / 2, 2) for name, v in prof.items()}}
del seqs
def filter_step():
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
    q = chunk.quality
    means = np.mean(q, axis=1)
    keep = means >= float(np.median(means[:100000]))      # (synthetic qualities are uniform per read: keeps ~all)
    keep[::3] = False                                           # ... so drop every third read by hand
    kept = chunk[keep].get_buffer().entry_bytes()
    return int(keep.sum()), kept.size
n_kept, n_bytes = filter_step(); sync(); dev.prof_enable(True); dev.prof_reset()
t0 = time.perf_counter()
for _ in range(2):
    filter_step()
sync(); dt = (time.perf_counter() - t0) / 2
prof = dev.prof_report(); dev.prof_enable(False)
out["quality_filter_compaction"] = {"reads": reads, "kept": n_kept, "bytes_out": n_bytes, "ms_per_step": round(dt * 1e3, 2),
                                    "gbases_per_s": round(reads * 150 / dt / 1e9, 2),
                                    "kernels_ms": {name: round(v["total_ms"] / 2, 2) for name, v in prof.items()}}
def rewrite_step():
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
    rc = bnp.sequence.get_reverse_complement(chunk.sequence)
    return bnp.FastQBuffer.from_data(bnp.replace(chunk, sequence=rc)).size
n_bytes = rewrite_step(); sync(); dev.prof_enable(True); dev.prof_reset()
t0 = time.perf_counter()
for _ in range(2):
    rewrite_step()
sync(); dt = (time.perf_counter() - t0) / 2
prof = dev.prof_report(); dev.prof_enable(False)
out["reverse_complement_rewrite_fastq"] = {"reads": reads, "bytes_out": n_bytes, "ms_per_step": round(dt * 1e3, 2),
                                           "kernels_ms": {name: round(v["total_ms"] / 2, 2) for name, v in prof.items()}}
del text, buf

# ---- config 5: sacCer3 index + big.fq.gz lookups ---------------------------------------------------------------
t0 = time.perf_counter()
genome = bnp.open(os.path.join(GOLD, "sacCer3.fa.gz")).read()
seqs = bnp.change_encoding(genome.sequence, bnp.DNAEncoding)
sync(); t_read = time.perf_counter() - t0
t0 = time.perf_counter()
index = bnp.KmerIndex.create_index(seqs, k=31)
sync(); t_build = time.perf_counter() - t0
t0 = time.perf_counter()
index = bnp.KmerIndex.create_index(seqs, k=31)
sync(); t_build2 = time.perf_counter() - t0
reads_fq = bnp.open(os.path.join(GOLD, "big.fq.gz")).read()
q = bnp.get_kmers(bnp.change_encoding(reads_fq.sequence, bnp.DNAEncoding), 31)
q._compact()
t0 = time.perf_counter()
lo, hi = index.get_indices_batch(q._flat_data())
sync(); t_lookup = time.perf_counter() - t0
hits = int((hi.dev() > lo.dev()).sum().item())
out["config5_kmer_index"] = {"genome_rows": len(genome), "genome_bases": int(seqs.total()), "index_pairs": index._keys.size,
                             "read_decode_s": round(t_read, 3), "build_s_first": round(t_build, 4),
                             "build_s": round(t_build2, 4), "queries": int(q.total()),
                             "lookup_s": round(t_lookup, 5), "queries_with_hit": hits}

# ---- streamed file -> 31-mer histogram (host file read + pinned H2D included) ----------------------------------------
n_file = min(reads, 16_000_000)
path = "/tmp/bnpk_stream_test.fq"
synth.fastq_bytes(n_file, 150, 7, 1, 5_000_000).tofile(path)
def stream_count():
    total = None
    for chunk in bnp.open(path).read_chunks(min_chunk_size=256_000_000):
        c = bnp.sequence.count_kmers(chunk.sequence, 31)
        total = c if total is None else total + c
    return total
c = stream_count(); sync()
t0 = time.perf_counter(); c = stream_count(); sync(); dt = time.perf_counter() - t0
out["stream_file_to_histogram"] = {"reads": n_file, "file_bytes": os.path.getsize(path), "chunk_bytes": 256_000_000,
                                   "seconds": round(dt, 3), "gbases_per_s": round(n_file * 150 / dt / 1e9, 3),
                                   "distinct": len(c), "file_gb_per_s": round(os.path.getsize(path) / dt / 1e9, 2)}
os.remove(path)

# ---- the API-path kernels against the HBM peak: algorithmic bytes (what the kernel has to read and write once) per launch ----
F, NL, NB, NR = 316 * reads, 4 * reads, 150 * reads, reads          # file bytes, lines, bases, reads of the synthetic FASTQ
c3, mt, pw = out["config3_minimizers"]["kernels_ms"], out["match_string_packed"]["kernels_ms"], out["pwm_scores_12"]["kernels_ms"]
flt, rw = out["quality_filter_compaction"], out["reverse_complement_rewrite_fastq"]
n_min = out["config3_minimizers"]["n_minimizers"]
table = {
    "byte_census": (c3["byte_census"], F),
    "line_positions": (c3["line_positions"], F + 8 * NL),
    "field_table": (c3["field_table"], 16 * NR + 16 * NR),
    "gather_encode_dna": (c3["gather_encode_dna"], NB + NB // 4 + NB // 8 + 16 * NR),
    "kmer_starts_from_ends": (c3["kmer_starts_from_ends"], 2 * (NB // 8)),
    "minimizers_flat": (c3["minimizers_flat"], 8 * n_min + NB // 4 + NB // 8),
    "match_windows_packed": (mt["match_windows_packed"], NB // 4 + NB // 8 + (NB - 6 * NR)),
    "match_rows_packed (per-row counts, no flags)": (out["match_string_packed"]["per_row_counts_ms"]["match_rows_packed"], NB // 4 + 16 * NR),
    "pwm_scores": (pw["pwm_scores"], NB // 4 + NB // 8 + 8 * (NB - 11 * NR)),
    "reverse_complement_packed": (out["reverse_complement_packed"]["kernel_ms"], 2 * (NB // 4) + 8 * NR),
    "reverse_complement_bytes": (rw["kernels_ms"]["reverse_complement_bytes"], 2 * NB + 8 * NR),
    "gather_rows (kept entries)": (flt["kernels_ms"]["gather_rows"], 2 * flt["bytes_out"] + 16 * flt["kept"]),
    "row_reduce_u8 (quality column where it lies in the text)": (flt["kernels_ms"]["row_reduce_u8"], NB + 24 * NR),
    "join_lines": (rw["kernels_ms"]["join_lines"], 2 * rw["bytes_out"]),
}
out["api_kernels"] = {name: {"ms": ms, "algorithmic_gb": round(b / 1e9, 2), "gb_per_s": round(b / (ms * 1e-3) / 1e9, 1),
                             "frac_of_8_tb_per_s": round(b / (ms * 1e-3) / 8e12, 3)} for name, (ms, b) in table.items() if ms}
print(json.dumps(out))
