mkdir -p gpurun_out/r02e
python -m pytest tests/test_hip_parity.py tests/test_hip_more.py tests/test_edge_cases_gpu.py -x -q -m gpu > gpurun_out/r02e/parity.txt 2>&1
tail -3 gpurun_out/r02e/parity.txt
for v in 1 2; do
for st in 1 2; do
  python bench.py --steps 100 --warmup 5 --no-cpu --no-secondary --variant $v --streams $st > gpurun_out/r02e/bench_v${v}_s$st.json 2> gpurun_out/r02e/bench_v${v}_s$st.err
  python -c "
import json
d=json.load(open('gpurun_out/r02e/bench_v${v}_s$st.json'))
print('variant $v streams $st', d['value'], d['ms_per_step'], d['kernels_ms'])"
done
done
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
for v in 1; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02e/trace_v$v -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary --streams 1 --variant $v > $R/gpurun_out/r02e/trace_v$v.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r02e/trace_v*/**/bench_kernel_stats.csv", recursive=True)):
    print(f)
    for r in csv.DictReader(open(f)):
        if "mh_" in r["Name"]:
            print("  %-60s calls %5s avg %9.2f us min %9.2f max %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
