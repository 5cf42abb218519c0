#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench, then PMC passes restricted to our
# kernels.  Every step is bounded by `timeout` (a PMC pass over the whole torch process once hung for 25 min).
# Outputs land in gpurun_out/prof_<tag>*; tools/summarize_profile.py <tag> copies the summaries into profiles/.
# PMC passes are separate runs with --pmc only (never combined with trace domains), as the microarch guide prescribes.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r02}
cd /tmp
# single stream: kernel durations are not inflated by the overlap of the two streams the headline uses
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- \
    python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary --streams 1 > $OUT/prof_${TAG}_bench.log 2>&1
echo "kernel-trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $C --kernel-include-regex "mh_project_gather|mh_project_taps|mh_search3" --output-format csv \
      -d $OUT/prof_${TAG}_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --streams 1 > $OUT/prof_${TAG}_$C.log 2>&1
  echo "pmc $C rc=$?"
done
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --kernel-include-regex "mh_search3" --output-format csv -d $OUT/prof_${TAG}_sq -o pmc -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --streams 1 > $OUT/prof_${TAG}_sq.log 2>&1
echo "pmc sq rc=$?"
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_VMEM \
    --kernel-include-regex "mh_search3" --output-format csv -d $OUT/prof_${TAG}_sq2 -o pmc -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --streams 1 > $OUT/prof_${TAG}_sq2.log 2>&1
echo "pmc sq2 rc=$?"
find $OUT/prof_$TAG* -name "*.csv" | head -40
