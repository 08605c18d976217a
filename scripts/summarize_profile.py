"""Summarise rocprofv3 rocpd databases (gpurun_out/prof/*/bench_results.db) into small text/JSON files
under profiles/.

    python scripts/summarize_profile.py gpurun_out/prof profiles/r01

* <out>_kernel_stats.txt : per-kernel calls / total / average / min / max duration (== --stats)
* <out>_pmc.json         : per-kernel FETCH_SIZE / WRITE_SIZE per launch (KB as reported + corrected
                           bytes: FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md §HBM)
"""
import json
import os
import sqlite3
import sys


def _short(name):
    if "rocprim" in name:
        if "radix_sort_onesweep_iteration" in name:
            return "rocprim::onesweep_iteration"
        if "radix_sort_onesweep_global_offsets" in name:
            return "rocprim::onesweep_global_offsets" + ("#2" if name.rstrip(")").endswith("#2}") else "#1")
        return "rocprim::" + name[-50:]
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    if "rp_hist_kmer_kernel" in name:
        return "rp_hist<kmer_source>"
    if "rp_hist_mem_kernel" in name:
        return "rp_hist<mem_source>"
    for key in ("fq_encode_fast", "fq_census_fast", "fq_zero_edges"):
        if key in name:
            return key
    if "fq_encode_kernel<true>" in name or "fq_census_kernel<true>" in name:
        return name.split("_kernel")[0] + "<handed back>"
    if "rp_ring_kernel" in name:
        return "rp_ring<kmer_source>"
    if "rp_scatter_kernel" in name or "rp_hist_kernel" in name:
        kind = "rp_scatter" if "rp_scatter_kernel" in name else "rp_hist"
        src = "kmer_source" if "kmer_source" in name else "mem_source"
        claim = ", claiming" if kind == "rp_scatter" and name.replace(" ", "").rstrip(">").endswith("true") else ""
        return "%s<%s%s>" % (kind, src, claim)
    for key in ("rp_claimed_tails", "rp_claimed_sizes", "hist_bytes_small", "hist_bytes_rows", "hist_packed2"):
        if key in name:
            return key
    for key in ("prefix_table", "rank_compose", "finish_multi_kernel", "bucket_census", "bucket_list", "window_cuts", "rebase_lines", "copy_plain", "copy_oneshot", "copy_unrolled"):
        if key in name:
            return key.replace("_kernel", "")
    if "byte_positions_kernel<true>" in name or "byte_positions_kernelILb1" in name:
        return "line_positions"
    for key in ("finish_wave_kernel", "finish_dup_kernel", "finish_compact", "fw_finalize", "wf_minimizer", "hist_weighted", "row_reduce_wide"):
        if key in name:
            return key.replace("_kernel", "") + ("<probe>" if key == "finish_wave_kernel" and name.rstrip(">").endswith("true") else "")
    for key in ("fq_census", "fq_encode", "fq_select", "fq_starts", "fq_detect_cr", "wf_generate", "wf_count",
                "finish_fast", "finish_sorted", "finish_check", "finish_collect", "finish_fit", "copy_peak", "mg_tile", "mg_split", "kmer_start_mask", "byte_census", "byte_positions", "validate_entries", "field_table", "scan_reduce", "scan_apply",
                "gather_encode", "kmer_kernel", "run_census", "run_heads", "run_sums", "synth_fastq", "fill_kernel",
                "hist_lds", "hist_global", "finish_runs", "partition_scatter", "partition_hist"):
        if key in name:
            return key + ("<minimizer>" if "true" in name and key == "kmer_kernel" else "")
    if "onesweep" in name:
        return "rocprim::onesweep_iteration" if "iteration" in name else "rocprim::onesweep_histograms"
    if "rocprim" in name:
        return "rocprim::" + name.split("::")[-1][:40]
    return name[-60:]


def kernel_stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(_short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["%-40s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%-40s %8d %14d %12d %12d %12d %6.2f%%" % (k, a[0], a[1], a[1] // a[0], a[2], a[3], 100.0 * a[1] / total))
    return "\n".join(lines) + "\n"


def pmc(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = con.execute("select %s, counter_name, value from counters_collection" % name_col).fetchall()
    agg = {}
    for name, counter, value in rows:
        a = agg.setdefault((_short(name), counter), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(value)
        a[2] = max(a[2], float(value))
    return {"%s|%s" % k: {"launches": v[0], "sum": v[1], "per_launch": v[1] / v[0], "largest": v[2]} for k, v in agg.items()}


def main():
    src, out = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    stats_db = os.path.join(src, "stats", "bench_results.db")
    if os.path.exists(stats_db):
        text = kernel_stats(stats_db)
        bench = os.path.join(src, "bench_stats.json")
        header = "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-extra [args of the profiled line below]\n"
        if os.path.exists(bench):
            header += "# bench line of the profiled run: " + open(bench).read().strip()[:1500] + "\n"
        open(out + "_kernel_stats.txt", "w").write(header + text)
        print(text)
    merged = {}
    for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        db = os.path.join(src, sub, "bench_results.db")
        if os.path.exists(db):
            merged.update(pmc(db))
    if merged:
        table = {}
        for key, v in merged.items():
            kern, counter = key.split("|")
            t = table.setdefault(kern, {})
            t[counter + "_KB_per_launch"] = round(v["per_launch"], 1)
            t[counter + "_KB_largest_launch"] = round(v["largest"], 1)     # (the decode kernels also run on the small parity inputs)
            t["launches"] = v["launches"]
        for kern, t in table.items():
            f = t.get("FETCH_SIZE_KB_per_launch")
            w = t.get("WRITE_SIZE_KB_per_launch")
            if f is not None:
                t["read_bytes_corrected"] = int(f * 1024 * 2)        # gfx950: FETCH_SIZE reports 1/2 of wide reads
            if w is not None:
                t["write_bytes"] = int(w * 1024)
            if t.get("FETCH_SIZE_KB_largest_launch") is not None:
                t["read_bytes_corrected_largest_launch"] = int(t["FETCH_SIZE_KB_largest_launch"] * 1024 * 2)
            if t.get("WRITE_SIZE_KB_largest_launch") is not None:
                t["write_bytes_largest_launch"] = int(t["WRITE_SIZE_KB_largest_launch"] * 1024)
        bench = os.path.join(src, "bench_fetch.json")
        if os.path.exists(bench):                      # the workload the counters were collected on
            try:
                table["_config"] = json.loads(open(bench).read().strip().splitlines()[-1])["config"]
            except Exception:
                pass
        try:                                           # the build the counters belong to (bench.py only trusts a matching one)
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bionumpy_amd.csrc.build import _source_hash
            table["_source_hash"] = _source_hash()
        except Exception:
            pass
        json.dump(table, open(out + "_pmc.json", "w"), indent=1, sort_keys=True)
        print(json.dumps(table, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
