#!/bin/bash
# A/B of bench.py options on the GPU box: headline + 8-bit leg per option set
R=${GRAFT_REPO_ROOT:-$(pwd)}
for OPT in "$@"; do
  python $R/bench.py --no-cpu --steps 100 $OPT 2>&1 | tail -1 > /tmp/ab_line.json
  python - "$OPT" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json")); q = d.get("secondary_8bit_maps", {})
print("opts [%s] headline" % sys.argv[1], d["value"], d.get("kernels_ms"), "| 8bit", q.get("value"), q.get("kernels_ms"))
PY
done
