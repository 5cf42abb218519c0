#!/bin/bash
# Per-round profile of the Gabor STAGE (GPU box): kernel trace of `tools/bench_gabor.py --stage`, then PMC passes (each in its
# own run, --pmc only).  Summaries -> gpurun_out/<tag>_*.txt (copied to profiles/ by hand).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${TAG:-r04_gabor}
mkdir -p $OUT
cd /tmp
CMD="python $R/tools/bench_gabor.py --stage --reps 5"
rm -rf $OUT/${TAG}_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- $CMD > $OUT/${TAG}_trace.log 2>&1
echo "trace rc=$?"
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $OUT/${TAG}_pmc$i
  timeout 300 rocprofv3 --pmc $SET --kernel-include-regex "mh_gabor|mh_dog" --output-format csv -d $OUT/${TAG}_pmc$i -o pmc -- $CMD > $OUT/${TAG}_pmc$i.log 2>&1
  echo "pmc set $i rc=$?"
done
python - <<PY > $OUT/${TAG}_summary.txt
import csv, glob, collections
print("# Gabor stage, one 1920x1080 view per mh_gabor_view call (tools/profile_gabor_round.sh); per-launch averages")
for f in sorted(glob.glob("$OUT/${TAG}_trace/**/*kernel_stats.csv", recursive=True)):
    print("## kernel trace (rocprofv3 --kernel-trace --stats)")
    for r in csv.DictReader(open(f)):
        print("%-60s calls=%s avg_ns=%s total_ns=%s pct=%s" % (r["Name"][:60], r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
print("## PMC (one set per run, --pmc only)")
for f in sorted(glob.glob("$OUT/${TAG}_pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%s,%s,avg=%.6g,launches=%d" % (k[0], k[1], sum(v) / len(v), len(v)))
PY
cat $OUT/${TAG}_summary.txt
