mkdir -p gpurun_out/r02i
python -m pytest tests/test_knn_gpu.py tests/test_hip_more.py tests/test_cli_gpu.py -q -m gpu -x > gpurun_out/r02i/tests.txt 2>&1
tail -6 gpurun_out/r02i/tests.txt
python tools/bench_mat.py > gpurun_out/r02i/bench_mat.txt 2>&1; cat gpurun_out/r02i/bench_mat.txt
python tools/refine_stages.py > gpurun_out/r02i/refine_stages.txt 2>&1
grep -E "mh-timing|refine total|pass" gpurun_out/r02i/refine_stages.txt | tail -20
for st in 2 3 4; do python bench.py --steps 100 --warmup 5 --no-cpu --no-secondary --streams $st 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $st', d['value'], d['ms_per_step'])"; done
