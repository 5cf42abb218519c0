mkdir -p gpurun_out/r02j
python -m pytest tests/ -q -m gpu > gpurun_out/r02j/tests.txt 2>&1
tail -6 gpurun_out/r02j/tests.txt
python tools/refine_stages.py > gpurun_out/r02j/refine_stages.txt 2>&1
grep -E "mh-timing|refine total|pass" gpurun_out/r02j/refine_stages.txt | tail -20
python bench.py --steps 100 --warmup 5 > gpurun_out/r02j/bench.json 2> gpurun_out/r02j/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02j/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['secondary_full_pass'], d['kernels_ms'])"
