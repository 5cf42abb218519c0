#!/bin/bash
# The randomised HIP-vs-oracle sweeps of a round in one gpurun call (exact equality):   bash tools/stress_round.sh <tag> [scale]
# -> gpurun_out/<tag>_stress_sweeps.txt (copy to profiles/).  scale multiplies the minutes of every sweep (default 1 = 34 min).
TAG=${1:-r06}
S=${2:-1}
OUT=gpurun_out/${TAG}_stress_sweeps.txt
mkdir -p gpurun_out
m() { python -c "print(round($1 * $S, 2))"; }
run() {
  local line
  line=$(python "$@" 2>&1 | tail -1)
  printf '%-78s %s\n' "python $*" "$line" | tee -a $OUT
}
echo "# $TAG randomised HIP-vs-oracle sweeps on one MI355X (exact equality), one gpurun call, $(date -u +%F)" > $OUT
run tests/stress_parity.py --minutes $(m 6) --seed 21
run tests/stress_parity.py --minutes $(m 5) --seed 22
run tests/stress_parity.py --minutes $(m 4) --seed 23 --ori-mode mix --body 1
run tests/stress_parity.py --minutes $(m 4) --seed 24 --codes
run tests/stress_parity.py --minutes $(m 2) --seed 25 --codes --body 1
run tests/stress_parity.py --minutes $(m 3) --seed 26 --max-views 300
run tests/stress_refine.py --minutes $(m 6) --seed 27
run tests/stress_more.py --minutes $(m 4) --seed 28
