#!/bin/bash
# Profile of the WHOLE exterior pass (filter -> optimize -> refine -> .mat) at the headline size on the GPU box:
#   bash tools/profile_fullpass.sh <tag>
# 1. stage timers (MH_TIMING=1, device-synchronised) of 4 passes          -> gpurun_out/<tag>_stages.txt
# 2. rocprofv3 --kernel-trace --stats of 3 passes (no stage synchronisation)  -> gpurun_out/<tag>_trace/
# 3. tools/summarize_fullpass.py: per-kernel averages of the LAST pass, GPU idle time between kernels, per stage
#                                                                          -> gpurun_out/<tag>_summary.txt (copy to profiles/)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r06_fullpass}
mkdir -p $OUT
cd /tmp
MH_FULLPASS_PLAIN=1 MH_TIMING=1 timeout 600 python $R/tools/time_full_pass.py 4 > $OUT/${TAG}_stages.txt 2>&1
echo "stages rc=$?"
rm -rf $OUT/${TAG}_trace
MH_FULLPASS_PLAIN=1 MH_TIMING=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- \
    python $R/tools/time_full_pass.py 3 > $OUT/${TAG}_trace.log 2>&1
echo "trace rc=$?"
python $R/tools/summarize_fullpass.py $OUT/${TAG}_trace $OUT/${TAG}_stages.txt > $OUT/${TAG}_summary.txt 2>&1
# HBM traffic of the pass's own kernels (separate --pmc passes, counters only; one pass of the driver each)
RX="mh_refine_loss_maps|mh_filter_|mh_knn_kernel|mh_medoid_kernel"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_pmc_$C
  MH_FULLPASS_PLAIN=1 MH_TIMING=0 timeout 600 rocprofv3 --pmc $C --kernel-include-regex "$RX" --output-format csv -d $OUT/${TAG}_pmc_$C -o pmc -- \
      python $R/tools/time_full_pass.py 1 > $OUT/${TAG}_pmc_$C.log 2>&1
  echo "pmc $C rc=$?"
done
python - <<PY >> $OUT/${TAG}_summary.txt
import csv, glob, collections
print("## PMC of the pass's kernels (--pmc only; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; SUMS over the 4 passes of one call of the driver: divide by 4 for a pass)")
for f in sorted(glob.glob("$OUT/${TAG}_pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%s,%s,sum=%.6g,launches=%d" % (k[0], k[1], sum(v), len(v)))
PY
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
# the raw trace is large: keep the compact last-pass timeline only
python - <<PY
import csv, glob, gzip
for f in glob.glob("$OUT/${TAG}_trace/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    with gzip.open("$OUT/${TAG}_timeline.csv.gz", "wt") as g:
        g.write("kernel,start_us,dur_us,stream,grid,wg\n")
        for r in rows:
            g.write("%s,%.1f,%.1f,%s,%s,%s\n" % (r["Kernel_Name"].split("(")[0].replace("void ", "").replace(",", ";")[:60],
                    (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                    r.get("Stream_Id", r.get("Queue_Id", "")), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
PY
rm -rf $OUT/${TAG}_trace/*/*kernel_trace.csv
tail -60 $OUT/${TAG}_stages.txt
cat $OUT/${TAG}_summary.txt
