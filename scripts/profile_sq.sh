#!/bin/bash
# usage: [MB_MODE=1] [MB_K=21] [SQ_TAG=_genome] profile_sq.sh   (MB_MODE=1: S-genome reads; MB_K: k, default 31)
# SQ counter passes (VALU / LDS / wait cycles) of the pipeline kernels on one 50 M-read batch; summaries -> gpurun_out/sq/
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -o x -- python $REPO/scripts/microbench.py 50000000 1 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc2 -o x -- python $REPO/scripts/microbench.py 50000000 1 > $OUT/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections, json, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/sq"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for pm in ("pmc1", "pmc2"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, pm), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "kernel<true>" in name and ("fq_encode" in name or "fq_census" in name):
                continue                                   # (the list-mode launches over the tiles handed back: empty)
            for key in ("rp_ring", "finish_multi", "finish_wave", "finish_dup", "finish_compact", "finish_fast", "finish_sorted", "rp_scatter", "rp_hist", "fq_encode_fast", "fq_census_fast", "fq_encode", "fq_census"):
                if key in name:
                    short = key + ("<kmer>" if ("kmer_source" in name or "rp_hist_kmer" in name) else "<mem>" if ("mem_source" in name or "rp_hist_mem" in name) else "")
                    agg[short][row["Counter_Name"]] += float(row["Counter_Value"])
                    launches[short + pm].add(row["Dispatch_Id"])
res = {}
for k, c in agg.items():
    n = max(len(launches[k + "pmc1"]), 1)
    d = {name: v / n for name, v in c.items()}
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    d["_per_launch_of"] = n
    d["_wait_any_frac"] = round(d.get("SQ_WAIT_ANY", 0) / wc, 3)
    d["_active_inst_any_frac"] = round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    d["_valu_active_frac_of_wave_cycles"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3)
    d["_lds_bank_conflict_frac_of_lds_active"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / (d.get("SQ_LDS_IDX_ACTIVE", 0) or 1), 3)
    res[k] = d
json.dump(res, open(out + "/sq_counters%s.json" % os.environ.get("SQ_TAG", ""), "w"), indent=1, sort_keys=True)
print(json.dumps({k: {x: y for x, y in v.items() if x.startswith("_")} for k, v in res.items()}, indent=1))
PY
