#!/bin/bash
# Round 5, the 8-bit regime: two views in flight in the select-only search kernel (csrc/pmvo_search.hip: two_views).
# Builds the library with the pairing off / on, at 5 and 4 waves per SIMD, and runs bench.py --codes against each build.
#   bash tools/exp_view_pairs.sh        (GPU box)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/monohair_amd/csrc
L=$R/monohair_amd/lib
OBJS=$(ls $C/*.o | grep -v pmvo_search)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
mkdir -p $R/gpurun_out/view_pairs
i=0
P=-DMH_EXP_VIEW_PAIRS
for defs in "" "$P" "$P -DMH_S3_WAVES_SELECT=4" "$P -DMH_PAIR_TAPS=2" "$P -DMH_PAIR_TAPS=2 -DMH_S3_WAVES_SELECT=4" "-DMH_S3_WAVES_SELECT=4" "$P -DMH_PAIR_TAPS=8 -DMH_S3_WAVES_SELECT=4"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc $FLAGS $defs -c $C/pmvo_search.hip -o /tmp/ps_$i.o
  lib=$L/libmhpmvo_exp_$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib /tmp/ps_$i.o $OBJS -ldl
  for rep in 1 2; do
    python $R/tools/ubench/run_lib.py $lib --no-cpu --no-secondary --codes --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s %8.1f it/s  step %.4f ms  search %.4f ms' % ('${defs:-shipped (one view at a time, 5 waves)}', d['value'], d['ms_per_step'], d['roofline']['launch_ms']))"
  done
  rm -f $lib
done | tee $R/gpurun_out/view_pairs/table.txt
