#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench + PMC passes.
# Outputs land in gpurun_out/prof_*; copy the summaries you want judged into profiles/.
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out
TAG=${1:-r01}
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu > $OUT/prof_${TAG}_bench.log 2>&1
# separate PMC passes (never combined with trace domains other than kernel-trace)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d $OUT/prof_${TAG}_pmc1 -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/prof_${TAG}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_${TAG}_pmc2 -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/prof_${TAG}_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $OUT/prof_${TAG}_pmc3 -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/prof_${TAG}_pmc3.log 2>&1
ls -R $OUT/prof_$TAG* | head -50
