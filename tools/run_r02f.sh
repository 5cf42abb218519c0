mkdir -p gpurun_out/r02f
python -m pytest tests/ -q -m gpu > gpurun_out/r02f/gputests.txt 2>&1
tail -5 gpurun_out/r02f/gputests.txt
python bench.py --steps 20 --warmup 3 --no-cpu --no-secondary > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err
tail -c 3000 gpurun_out/r02f/bench.json; tail -5 gpurun_out/r02f/bench.err
