mkdir -p gpurun_out/r02g
python -m pytest tests/test_raster_gpu.py tests/test_cli_gpu.py tests/test_bench_gpu.py -q -m gpu > gpurun_out/r02g/tests.txt 2>&1
tail -15 gpurun_out/r02g/tests.txt
python tools/refine_stages.py > gpurun_out/r02g/refine_stages.txt 2>&1
grep -E "mh-timing|refine total|pass" gpurun_out/r02g/refine_stages.txt | tail -40
