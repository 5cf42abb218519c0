#!/bin/bash
# Per-round profile of one bench.py regime on the GPU box:   bash tools/profile_round.sh <tag> "<bench args>" "<kernel regex>"
#   e.g.  bash tools/profile_round.sh r04_8bit "--codes" "mh_project_taps|mh_topk|mh_search"
# One rocprofv3 kernel trace (ONE HIP stream: kernels do not overlap, durations are per kernel) and PMC passes, each its
# own run with --pmc only.  Summary -> gpurun_out/<tag>_summary.txt (copy into profiles/).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r04}
ARGS=${2:-}
RX=${3:-"mh_project_taps|mh_project_gather|mh_topk|mh_search"}
mkdir -p $OUT
cd /tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary --streams 1 $ARGS"
rm -rf $OUT/${TAG}_trace
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- $CMD > $OUT/${TAG}_trace.log 2>&1
echo "trace rc=$?"
grep "^{\"metric\"" $OUT/${TAG}_trace.log | tail -1 > $OUT/${TAG}_bench.json
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $OUT/${TAG}_pmc$i
  timeout 400 rocprofv3 --pmc $SET --kernel-include-regex "$RX" --output-format csv -d $OUT/${TAG}_pmc$i -o pmc -- \
      python $R/bench.py --steps 4 --warmup 1 --no-cpu --no-secondary --streams 1 $ARGS > $OUT/${TAG}_pmc$i.log 2>&1
  echo "pmc set $i rc=$?"
done
python - <<PY > $OUT/${TAG}_summary.txt
import csv, glob, collections
print("# $TAG: rocprofv3 of \`$CMD\` (tools/profile_round.sh); MI355X, ONE HIP stream; per-launch averages")
for f in sorted(glob.glob("$OUT/${TAG}_trace/**/*kernel_stats.csv", recursive=True)):
    print("## kernel trace (rocprofv3 --kernel-trace --stats)")
    print("kernel,calls,avg_us,min_us,max_us,pct")
    for r in csv.DictReader(open(f)):
        print("%s,%s,%.2f,%.2f,%.2f,%s" % (r["Name"].split("(")[0].replace("void ", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                        float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
print("## PMC (one set per run, --pmc only; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them)")
for f in sorted(glob.glob("$OUT/${TAG}_pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0].replace("void ", "")[-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%s,%s,avg=%.6g,launches=%d" % (k[0], k[1], sum(v) / len(v), len(v)))
PY
cat $OUT/${TAG}_summary.txt
