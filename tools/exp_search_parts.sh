#!/bin/bash
# Where the time of mh_search3_kernel goes (GPU box; round 4): builds the library with parts of the kernel switched off --
# WRONG RESULTS, timing only -- and runs bench.py's timed loop on ONE stream against each build:
#   -DMH_EXP_NOTAPS   every tap list treated as one tap (no tap loops, no key decode)
#   -DMH_EXP_NOPROJ   the per-(item, view) projection / normalisation replaced by two additions
# and both.  roofline.launch_ms of the four lines gives the split quoted in DESIGN.md section 7.
#   bash tools/exp_search_parts.sh
set -e
EXTRA="$@"   # e.g. --codes: the split of the select-only kernel on 8-bit maps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/monohair_amd/csrc
L=$R/monohair_amd/lib
OBJS=$(ls $C/*.o | grep -v pmvo_search)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
for v in "" NOTAPS NOPROJ "NOTAPS NOPROJ"; do
  tag=$(echo $v | tr -d ' ')
  lib=$L/libmhpmvo.so
  if [ -n "$tag" ]; then
    defs=""; for d in $v; do defs="$defs -DMH_EXP_$d"; done
    /opt/rocm/bin/hipcc $FLAGS $defs -c $C/pmvo_search.hip -o /tmp/ps_$tag.o
    lib=$L/libmhpmvo_exp_$tag.so
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib /tmp/ps_$tag.o $OBJS -ldl
  fi
  python $R/tools/ubench/run_lib.py $lib --no-cpu --no-secondary --streams 1 $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s search launch %.4f ms   (step %.4f ms)' % ('${tag:-full}', d['roofline']['launch_ms'], d['ms_per_step']))"
  [ -n "$tag" ] && rm -f $lib
done
