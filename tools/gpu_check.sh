#!/bin/bash
# What the driver runs at round end, in one gpurun call: the GPU test suite, smoke(), the default bench line.
# usage (from the repo root, on the GPU box): bash tools/gpu_check.sh <tag>
TAG=${1:-check}
mkdir -p gpurun_out/$TAG
python -m pytest tests/ -q -m gpu > gpurun_out/$TAG/gputests.txt 2>&1
tail -3 gpurun_out/$TAG/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python -c "
import json; d=json.load(open('gpurun_out/$TAG/bench.json'))
print(d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], [ (k['kernel'], k['frac']) for k in d['roofline_kernels']], d['secondary_full_pass'], d['cpu_baseline']['value'])"
