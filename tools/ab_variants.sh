#!/bin/bash
# A/B of search-kernel variants on the GPU box: tools/ab_variants.sh 256 2256 ...
for v in "$@"; do
  timeout 300 python bench.py --no-cpu --no-secondary --variant $v 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant', $v, d['value'], d['kernels_ms'])"
done
