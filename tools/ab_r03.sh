#!/bin/bash
# A/B of bench.py options on the GPU box: headline + 8-bit leg per option set
R=${GRAFT_REPO_ROOT:-$(pwd)}
for OPT in "$@"; do
  python $R/bench.py --no-cpu --steps 100 $OPT 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); q=d.get('secondary_8bit_maps',{})
print('opts [$OPT] headline', d['value'], '| 8bit', q.get('value'), q.get('kernels_ms'))"
done
