python -m pytest tests/test_hip_parity.py -q -x 2>&1 | tail -2
for c in 1 2 3; do python bench.py --steps 200 --warmup 5 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('run $c', d['value'], d['ms_per_step'], d['kernels_ms'])"; done
