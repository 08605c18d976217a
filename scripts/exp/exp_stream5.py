"""Experiment: file -> histogram through the API, host-side time line without device-wide synchronisation."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd import synth
from bionumpy_amd.io import parser
n_file = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256_000_000
path = "/tmp/bnpk_stream_test.fq"
synth.fastq_bytes(n_file, 150, 7, 1, 5_000_000).tofile(path)
T0 = [0.0]
log = []
orig_fill, orig_parse = parser.NumpyFileReader._fill, parser.NumpyFileReader._parse
def fill(self, target, upload=None):
    a = time.perf_counter(); r = orig_fill(self, target, upload); b = time.perf_counter()
    log.append(("fill[%s]" % threading.current_thread().name, a - T0[0], b - T0[0])); return r
def parse(self, batch):
    a = time.perf_counter(); r = orig_parse(self, batch); b = time.perf_counter()
    log.append(("parse", a - T0[0], b - T0[0])); return r
parser.NumpyFileReader._fill, parser.NumpyFileReader._parse = fill, parse
orig_assemble = parser._EarlyUpload.assemble
def assemble(self, i, event, room, first, n, front, got):
    a = time.perf_counter(); event.synchronize(); b = time.perf_counter()
    log.append(("wait for the upload event", a - T0[0], b - T0[0]))
    r = orig_assemble(self, i, event, room, first, n, front, got)
    torch.cuda.current_stream().synchronize(); c = time.perf_counter()
    log.append(("assemble (D2D + small H2D), synced", b - T0[0], c - T0[0]))
    return r
if os.environ.get("BNPK_TRACE_ASSEMBLE"):
    parser._EarlyUpload.assemble = assemble
def run(show):
    del log[:]
    T0[0] = time.perf_counter()
    total = None
    it = iter(bnp.open(path).read_chunks(min_chunk_size=chunk))
    while True:
        a = time.perf_counter()
        try:
            ch = next(it)
        except StopIteration:
            break
        b = time.perf_counter()
        c = bnp.sequence.count_kmers(ch.sequence, 31)
        total = c if total is None else total + c
        d = time.perf_counter()
        log.append(("next", a - T0[0], b - T0[0])); log.append(("count+add", b - T0[0], d - T0[0]))
    e = time.perf_counter(); torch.cuda.synchronize(); f = time.perf_counter()
    log.append(("final sync", e - T0[0], f - T0[0]))
    if show:
        for name, a, b in sorted(log, key=lambda x: x[1]):
            print("%7.2f .. %7.2f ms  %s" % (a * 1e3, b * 1e3, name))
run(False); run(False); run(True)
os.remove(path)
