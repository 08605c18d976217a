"""Differential fuzz of the FASTQ decode: the fast tile kernels (census + encode) against the general ones on random line
structures — line lengths from empty to several tiles, CRLF, trailing incomplete entries, 1-4 lines per entry, any
sequence line, invalid bases and bad header / '+' bytes at random places.  Both must give the same bits or the same error."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device, HArray
from bionumpy_amd.ops import get_ops

ops = get_ops(); dev = Device.get()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from fuzz_text import random_text as make  # noqa: E402


def run(buf, lpe, seq_line, check_plus, encoder):
    assert lib.bnpk_set_option(dev.ctx, b"fastq_encoder", encoder) == 0
    try:
        packed, ends, n_records, n_bases = ops.fastq_encode(HArray(host=buf), buf.size, lpe, seq_line, ord("@"), check_plus)
        return ("ok", n_records, n_bases, packed.host().tobytes(), ends.host().tobytes())
    except Exception as e:                                      # noqa: BLE001
        return ("error", type(e).__name__, str(e), getattr(e, "line_number", None), getattr(e, "offset", None))


t0 = time.time(); n = 0; errors = 0; oks = 0
seed = seed0
while time.time() - t0 < seconds:
    rng = np.random.default_rng(seed)
    buf, lpe, seq_line, check_plus = make(rng)
    if seq_line == 0 and lpe > 1:
        pass
    a = run(buf, lpe, seq_line, check_plus, 1)
    b = run(buf, lpe, seq_line, check_plus, 0)
    if a != b:
        np.save("/tmp/fuzz_fail_%d.npy" % seed, buf)
        print("MISMATCH seed %d lpe %d seq_line %d check_plus %s size %d: %s vs %s" % (seed, lpe, seq_line, check_plus, buf.size,
              a[:3] if a[0] == "ok" else a, b[:3] if b[0] == "ok" else b), flush=True)
        if a[0] == b[0] == "ok":
            pa, pb = np.frombuffer(a[3], dtype=np.uint64), np.frombuffer(b[3], dtype=np.uint64)
            ea, eb = np.frombuffer(a[4], dtype=np.uint64), np.frombuffer(b[4], dtype=np.uint64)
            if pa.size == pb.size:
                d = np.flatnonzero(pa != pb); print("  packed words differ at", d[:5], "of", pa.size)
            if ea.size == eb.size:
                d = np.flatnonzero(ea != eb); print("  end words differ at", d[:5], "of", ea.size)
        errors += 1
        if errors > 5:
            break
    oks += a[0] == "ok"
    n += 1; seed += 1
lib.bnpk_set_option(dev.ctx, b"fastq_encoder", 1)
print("fuzz: %d cases (%d decoded, %d raised alike), %d mismatches, seeds %d..%d" % (n, oks, n - oks - errors, errors, seed0, seed - 1))
