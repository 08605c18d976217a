"""Summary of the rocprofv3 --kernel-trace --memory-copy-trace capture of bench.py's host-fed leg: the hipMemcpyAsync
rows (direction, bytes, duration, GB/s) and how much of the copy time had a kernel running next to it."""
import csv, glob, os, sys

src, out = sys.argv[1], sys.argv[2]
copies, kernels = [], []
def pick(row, *needles):
    for k, v in row.items():
        if any(n in k.lower() for n in needles):
            return v
    return ""


headers = []
for f in glob.glob(os.path.join(src, "**", "*memory_copy_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    if rows:
        headers = list(rows[0].keys())
    for r in rows:
        s, e = int(pick(r, "start")), int(pick(r, "end"))
        size = pick(r, "bytes", "size") or "0"
        copies.append((s, e, pick(r, "direction") or pick(r, "kind"), int(float(size))))
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kernels.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
copies.sort()
kernels.sort()
chunk_bytes = 1 << 30
try:                                                       # (this rocprofv3 prints no size column: the chunks are 1 GiB, bench.py says)
    import json
    hf = json.loads(open(os.path.join(os.path.dirname(src.rstrip("/")), "bench_hostfed.json")).read().strip().splitlines()[-1])["host_fed"]
    chunk_bytes = hf["chunk_bytes"]
except Exception:
    hf = None
if copies and max(c[3] for c in copies) == 0:              # no byte counts in the trace: long host-to-device copies are the chunks
    copies = [(s, e, d, chunk_bytes if (e - s) > 5_000_000 and "HOST_TO_DEVICE" in d.upper().replace(" ", "_") else 0)
              for s, e, d, _ in copies]
big = [c for c in copies if c[3] >= (1 << 28)]
lines = ["# rocprofv3 --kernel-trace --memory-copy-trace -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline",
         "# memory copies of at least 256 MiB (the host-fed leg's chunks; sizes taken as bench.py's chunk size where the trace "
         "has no size column, the last chunk of a batch is shorter): %d rows" % len(big),
         "# bench.py host_fed of the traced run: %s" % (json.dumps(hf) if hf else "?"),
         "%-18s %14s %12s %10s" % ("direction", "bytes", "duration_us", "GB/s")]
for s, e, d, b in (big if len(big) <= 24 else big[:12] + [("...",) * 4] + big[-12:]):
    if s == "...":
        lines.append("...")
        continue
    lines.append("%-18s %14d %12.1f %10.2f" % (d, b, (e - s) / 1e3, b / max(e - s, 1)))
if not big:
    lines.append("(no large copies found; columns of the memory-copy trace: %s; %d rows in all, largest %d bytes)"
                 % (headers, len(copies), max([c[3] for c in copies] or [0])))
h2d = [c for c in big if "HOST_TO_DEVICE" in c[2].upper().replace(" ", "_") or "H2D" in c[2].upper() or "HTOD" in c[2].upper()]
if h2d:
    busy = sum(e - s for s, e, _, _ in h2d)
    lines.append("H2D chunks: %d, %.2f GB in %.1f ms of copy-engine time = %.2f GB/s" %
                 (len(h2d), sum(c[3] for c in h2d) / 1e9, busy / 1e6, sum(c[3] for c in h2d) / max(busy, 1)))
    # overlap: copy time during which at least one kernel was running
    overlap, j = 0, 0
    ks = [(s, e) for s, e, _ in kernels]
    for s, e, _, _ in h2d:
        for ks_, ke_ in ks:
            if ke_ <= s:
                continue
            if ks_ >= e:
                break
            overlap += min(e, ke_) - max(s, ks_)
    lines.append("kernel time that ran during those copies: %.1f ms (kernels overlapping each other are counted once each)" % (overlap / 1e6))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
