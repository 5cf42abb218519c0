#!/bin/bash
# PMC passes on the Gabor kernel (GPU box): where do its cycles go?  Outputs under gpurun_out/prof_gabor_*.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQC_DCACHE_[A-Z_]+|SQ_INST_CYCLES_[A-Z_]+|SQ_WAIT_INST_[A-Z_]+|SQ_ACTIVE_INST_[A-Z_]+" | sort -u | tr '\n' ' ' > $OUT/gabor_counters.txt
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_INPUT_VALID_READYB" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $SET --kernel-include-regex "mh_gabor" --output-format csv -d $OUT/prof_gabor_$i -o pmc -- \
      python $R/tools/bench_gabor.py --reps 2 --variant ${GABOR_VARIANT:-mfma2} > $OUT/prof_gabor_$i.log 2>&1
  echo "set $i rc=$?"
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/prof_gabor_*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k[0], k[1], "avg=%.4g" % (sum(v) / len(v)), "n=%d" % len(v))
PY
cat $OUT/gabor_counters.txt
