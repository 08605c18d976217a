"""Experiment: the big-batch reader's phases (file read in the background thread vs upload + scan in the caller)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bionumpy_amd as bnp
from bionumpy_amd import synth
from bionumpy_amd.io import parser
n_file = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
path = "/tmp/bnpk_stream_test.fq"
synth.fastq_bytes(n_file, 150, 7, 1, 5_000_000).tofile(path)
T = {"fill": 0.0, "parse": 0.0}
orig_fill, orig_parse = parser.NumpyFileReader._fill, parser.NumpyFileReader._parse
def fill(self, target, upload=None):
    t = time.perf_counter(); r = orig_fill(self, target, upload); T["fill"] += time.perf_counter() - t; return r
def parse(self, batch):
    t = time.perf_counter(); r = orig_parse(self, batch); torch.cuda.synchronize(); T["parse"] += time.perf_counter() - t; return r
parser.NumpyFileReader._fill, parser.NumpyFileReader._parse = fill, parse
for rep in range(3):
    T["fill"] = T["parse"] = 0.0
    t0 = time.perf_counter(); n = 0; tc = 0.0; tm = 0.0; total = None
    for ch in bnp.open(path).read_chunks(min_chunk_size=chunk):
        a = time.perf_counter(); c = bnp.sequence.count_kmers(ch.sequence, 31); torch.cuda.synchronize(); tc += time.perf_counter() - a; n += 1
        a = time.perf_counter(); total = c if total is None else total + c; torch.cuda.synchronize(); tm += time.perf_counter() - a
    dt = time.perf_counter() - t0
    print("total %.1f ms (%d chunks): file fill %.1f ms (background thread when read-ahead), parse (upload + scan) %.1f ms, count %.1f ms, merge %.1f ms -> %.2f GB/s, %.2f Gbases/s" % (dt * 1e3, n, T["fill"] * 1e3, T["parse"] * 1e3, tc * 1e3, tm * 1e3, os.path.getsize(path) / dt / 1e9, n_file * 150 / dt / 1e9), flush=True)
os.remove(path)
