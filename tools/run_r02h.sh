mkdir -p gpurun_out/r02h
python -m pytest tests/test_knn_gpu.py tests/test_cli_gpu.py tests/test_hip_more.py -q -m gpu > gpurun_out/r02h/tests.txt 2>&1
tail -8 gpurun_out/r02h/tests.txt
python tools/refine_stages.py > gpurun_out/r02h/refine_stages.txt 2>&1
grep -E "mh-timing|refine total|pass" gpurun_out/r02h/refine_stages.txt | tail -20
