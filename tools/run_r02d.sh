mkdir -p gpurun_out/r02d
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bit_exact" > gpurun_out/r02d/parity.txt 2>&1
tail -2 gpurun_out/r02d/parity.txt
for v in 256 1 2; do
  python bench.py --steps 100 --warmup 5 --no-cpu --no-secondary --variant $v > gpurun_out/r02d/bench_v$v.json 2> gpurun_out/r02d/bench_v$v.err
  python -c "
import json
d=json.load(open('gpurun_out/r02d/bench_v$v.json'))
print('variant $v', d['value'], d['ms_per_step'], d['kernels_ms'])"
done
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
for v in 256 1; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02d/trace_v$v -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary --streams 1 --variant $v > $R/gpurun_out/r02d/trace_v$v.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "mh_search" --output-format csv -d $R/gpurun_out/r02d/pmc_v$v -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --streams 1 --variant $v > $R/gpurun_out/r02d/pmc_v$v.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r02d/trace_v*/**/bench_kernel_stats.csv", recursive=True)):
    print(f)
    for r in csv.DictReader(open(f)):
        if "mh_" in r["Name"]:
            print("  %-60s calls %5s avg %9.2f us min %9.2f max %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
for f in sorted(glob.glob("gpurun_out/r02d/pmc_v*/**/pmc_counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(f)
    for k,v in sorted(agg.items()):
        print("  ", k, round(sum(v)/len(v),1), len(v))
PY
