#!/bin/bash
# PMC passes over the search kernel only (bounded).  usage: bash tools/profile_search.sh <tag> [bench args]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-s}
shift
cd /tmp
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --kernel-include-regex "mh_search" --output-format csv -d $OUT/prof_${TAG}_sq1 -o pmc -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/prof_${TAG}_sq1.log 2>&1
echo rc=$?
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_WAVES_EQ_64 SQ_INST_LEVEL_SMEM \
    --kernel-include-regex "mh_search" --output-format csv -d $OUT/prof_${TAG}_sq2 -o pmc -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/prof_${TAG}_sq2.log 2>&1
echo rc=$?
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/prof_${TAG}_sq*/pmc_counter_collection.csv")):
    agg=collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open(f)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for row in csv.DictReader(open(f)):
        pass
    print(f.split('/')[-2], {k: round(sum(v)/len(v),1) for k,v in agg.items()})
PY
