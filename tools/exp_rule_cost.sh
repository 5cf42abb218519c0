#!/bin/bash
# What following the reference's batch composition costs per iteration (round 5): the headline loop and the 8-bit regime with
# the batch rules on (default) and off.  usage (GPU box, repo root): bash tools/exp_rule_cost.sh [steps]
STEPS=${1:-300}
mkdir -p gpurun_out/rule_cost
for maps in "" "--codes"; do
  for opt in "" "--option reproject_rule=1" "--option sum_block=0" "--option reproject_rule=1 --option sum_block=0"; do
    for rep in 1 2; do
      python bench.py --no-cpu --no-secondary --steps $STEPS --warmup 20 $maps $opt 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-8s %-45s %8.1f it/s  %.4f ms  %s' % ('$maps' or 'fp32', '$opt' or 'default', d['value'], d['ms_per_step'], d.get('kernels_ms')))"
    done
  done
done | tee gpurun_out/rule_cost/table.txt
