cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fq
OUT=$GRAFT_REPO_ROOT/gpurun_out/fq
ARGS="${FQ_ARGS:-20000000 150 1,0 2}"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o x -- python $GRAFT_REPO_ROOT/scripts/exp/exp_fq.py $ARGS > $OUT/kt.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -o x -- python $GRAFT_REPO_ROOT/scripts/exp/exp_fq.py $ARGS > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc2 -o x -- python $GRAFT_REPO_ROOT/scripts/exp/exp_fq.py $ARGS > $OUT/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections, os, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/fq"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
def key(name):
    m = re.search(r"(fq_\w+?)(_kernel)?(<\w+>)?\(", name)
    return (m.group(1) + (m.group(3) or "")) if m else None
for pm in ("pmc1", "pmc2"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, pm), recursive=True):
        for row in csv.DictReader(open(f)):
            k = key(row["Kernel_Name"])
            if k: agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("%s/kt/**/*kernel_trace.csv" % out, recursive=True):
    for row in csv.DictReader(open(f)):
        k = key(row["Kernel_Name"])
        if k: dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
with open(out + "/counters.txt", "w") as fo:
    for k, c in sorted(agg.items()):
        d = {n: max(v) for n, v in c.items()}
        waves = d.get("SQ_WAVES", 0) or 1
        line = "%s: ms(min/med of %d)=%.3f/%.3f waves=%.4g VALU/wave=%.0f SALU/wave=%.0f LDS/wave=%.0f VMEM_RD/wave=%.1f lds_idx_active=%.3g bank_conflict=%.3g wave_cycles/wave=%.0f wait_any_frac=%.2f busy_cycles=%.3g" % (
            k, len(dur[k]), min(dur[k]) if dur[k] else -1, sorted(dur[k])[len(dur[k]) // 2] if dur[k] else -1, waves, d.get("SQ_INSTS_VALU", 0) / waves, d.get("SQ_INSTS_SALU", 0) / waves, d.get("SQ_INSTS_LDS", 0) / waves,
            d.get("SQ_INSTS_VMEM_RD", 0) / waves, d.get("SQ_LDS_IDX_ACTIVE", 0), d.get("SQ_LDS_BANK_CONFLICT", 0), d.get("SQ_WAVE_CYCLES", 0) / waves,
            d.get("SQ_WAIT_ANY", 0) / (d.get("SQ_WAVE_CYCLES", 0) or 1), d.get("SQ_BUSY_CYCLES", 0))
        print(line); fo.write(line + "\n")
PY
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/kt
