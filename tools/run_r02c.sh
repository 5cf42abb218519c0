mkdir -p gpurun_out/r02c
python -m pytest tests/test_hip_parity.py tests/test_loaders_golden.py -x -q -m gpu > gpurun_out/r02c/parity.txt 2>&1
tail -3 gpurun_out/r02c/parity.txt
for v in 256 1 2 3 4; do
  python bench.py --steps 100 --warmup 5 --no-cpu --no-secondary --variant $v > gpurun_out/r02c/bench_v$v.json 2> gpurun_out/r02c/bench_v$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r02c/bench_v$v.json"))
print("variant $v", d["value"], d["ms_per_step"], d["kernels_ms"])
PY
done
