#!/bin/bash
# Round 5 experiment: the sorted-list tap search (csrc/pmvo_search.hip, -DMH_EXP_SORTED) against the shipped key body.
# Builds the library both ways (and the sorted form at 4 waves per SIMD), runs bench.py with its in-run parity check
# (a differing bit = no line) and prints step / search-kernel times.      bash tools/exp_sorted.sh     (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/monohair_amd/csrc
L=$R/monohair_amd/lib
OBJS=$(ls $C/*.o | grep -v pmvo_search)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
mkdir -p $R/gpurun_out/sorted
i=0
for defs in "" "-DMH_EXP_SORTED" "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc $FLAGS $defs -c $C/pmvo_search.hip -o /tmp/ps_$i.o || continue
  lib=$L/libmhpmvo_exp_$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib /tmp/ps_$i.o $OBJS -ldl
  extra=""; case "$defs" in *MAXIT*) extra="--no-cpu";; esac     # (timing-only builds: no parity check)
  python $R/tools/ubench/run_lib.py $lib --no-secondary --steps 200 --warmup 20 $extra 2>$R/gpurun_out/sorted/err_$i.txt | python -c "
import sys, json
t = sys.stdin.read().strip().splitlines()
try:
    d = json.loads(t[-1])
    print('%-40s %8.1f it/s  step %.4f ms  search %.4f ms  parity %s' % ('${defs:-shipped}', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d.get('parity_check', {}).get('bit_exact')))
except Exception as e:
    print('%-40s FAILED %r' % ('${defs:-shipped}', e))"
  tail -2 $R/gpurun_out/sorted/err_$i.txt | cut -c1-300
  rm -f $lib
done | tee $R/gpurun_out/sorted/table.txt
