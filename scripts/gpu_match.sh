#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_api.py tests/test_fuzz.py -m gpu -x -q -k "match or window_flags or byte_movers" 2>&1 | tail -5
timeout 300 python - <<'PY'
import numpy as np, torch, time
import bionumpy_amd as bnp
from bionumpy_amd import ops as O
from bionumpy_amd.device import Device
ops = O.get_ops(); dev = Device.get()
text = ops.synth_fastq(50_000_000, 150, 20260925, 0, 0, 0)
dna = bnp.change_encoding(bnp.FastQBuffer.from_raw_buffer(text).get_field_by_number(1), bnp.DNAEncoding)
dna._compact()
for motif in ("GATTACA", "ACGTACGTACGTACG", "A" * 31):
    f = lambda: bnp.match_string(dna, motif).any(axis=-1)
    f(); torch.cuda.synchronize()
    dev.prof_enable(True); dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(3): r = f()
    torch.cuda.synchronize()
    print(motif, (time.perf_counter() - t0) / 3 * 1e3, "ms", {k: round(v["total_ms"] / 3, 2) for k, v in dev.prof_report().items()}, int(r.sum()))
    dev.prof_enable(False)
PY
