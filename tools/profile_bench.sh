#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench, then PMC passes restricted to our
# kernels.  Every step is bounded by `timeout` (a PMC pass over the whole torch process once hung for 25 min).
# Outputs land in gpurun_out/prof_<tag>*; copy the summaries you want judged into profiles/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r01}
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- \
    python $R/bench.py --steps 20 --warmup 3 --no-cpu > $OUT/prof_${TAG}_bench.log 2>&1
echo "kernel-trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $C --kernel-include-regex "mh_project_gather|mh_search|mh_prep" --output-format csv \
      -d $OUT/prof_${TAG}_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/prof_${TAG}_$C.log 2>&1
  echo "pmc $C rc=$?"
done
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --kernel-include-regex "mh_search" --output-format csv -d $OUT/prof_${TAG}_sq -o pmc -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/prof_${TAG}_sq.log 2>&1
echo "pmc sq rc=$?"
find $OUT/prof_$TAG* -type f | head -40
