"""-m gpu: bench.py's one-line JSON contract at a reduced size (the driver runs it at full size): the keys the driver and
the judge read are there, the full-size parity block is green, the host-fed leg and the CPU baseline report."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "400000", "--steps", "1", "--warmup", "1",
                          "--cpu-sample-reads", "20000", "--host-fed-batches", "1"] + list(extra),
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_has_what_the_driver_reads():
    d = _run()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True and d["unit"] == "Gbases/s"
    assert d["dtype"] == "int64" and d["data"] == "synthetic" and "workload" in d["config"] and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["measured_copy_gb_per_s"] and r["measured_copy_gb_per_s"] > 1000
    assert d["parity_fullsize"] is True and d["parity"]["ok"] is True
    assert d["parity"]["kmers_checked"] == d["config"]["kmers_per_gpu"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["all_cores"]["cores"] >= 1
    h = d["host_fed"]
    assert h["source"] == "pinned RAM" and h["gbases_per_s"] > 0 and h["h2d_gb_per_s"] > 0 and 0 <= h["overlap_frac"] <= 1


def test_bench_virtual_ranks_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--virtual-ranks", "4", "--reads", "400000", "--steps", "1",
                          "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["virtual_ranks"] == 4 and len(d["int64_words_received_per_rank"]) == 4 and d["plan"] == "keys"
    assert sum(d["int64_words_received_per_rank"]) == 400000 * 120


def test_bench_from_file_line(tmp_path):
    """--from-file: the reader's stream form over a FASTQ file it writes first; reads add up, checksums match"""
    path = str(tmp_path / "reads.fq")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--from-file", path, "--reads", "300000", "--steps", "1",
                          "--warmup", "1", "--file-chunk-mb", "32"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["reads_seen_by_all_ranks"] == 300000 and d["reads_add_up"] is True and d["parity_fullsize"] is True
    assert d["file_bytes"] == 300000 * 316 and d["gbases_per_s"] > 0
