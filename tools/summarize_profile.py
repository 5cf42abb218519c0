#!/usr/bin/env python
"""Turn the raw rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<tag>*) into the tracked artefacts under
profiles/: <tag>_kernel_stats.csv, <tag>_pmc_<COUNTER>.csv, <tag>_summary.txt and traffic.json (what bench.py reports
as roofline.traffic).  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled for the 16-byte-per-lane reads of the
gather kernels, as MI355X_MICROARCH.md prescribes for gfx950.
    python tools/summarize_profile.py r01h
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
ALGO = 364860000        # V*N*(12P+20) read + the same written + 12N, V=60 N=5000 P=49
lines = ["# profile summary %s -- rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 "
         "--warmup 3 --no-cpu" % tag,
         "# MI355X (gfx950), synthetic 60 views @ 1920x1080, 5000 points / iteration, patch 7, two HIP streams",
         "# (kernels of the two streams overlap, which inflates their durations; the roofline kernel",
         "#  mh_project_gather_kernel is launched alone, back to back, by bench.py's measurement section)", "",
         "kernel,calls,avg_us,min_us,max_us,pct_of_gpu_time"]
stats = os.path.join(src, "prof_%s" % tag, "bench_kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
dur = {}
for r in csv.DictReader(open(stats)):
    if "mh_" in r["Name"]:
        name = r["Name"].split("(")[0].replace("void ", "")
        dur[name] = float(r["AverageNs"]) / 1e3
        lines.append("%s,%s,%.2f,%.2f,%.2f,%s" % (name, r["Calls"], dur[name], float(r["MinNs"]) / 1e3,
                                                  float(r["MaxNs"]) / 1e3, r["Percentage"]))
lines += ["", "# PMC passes (separate runs, `--pmc <counter> --kernel-include-regex mh_...`, per-launch averages)"]
pmc = collections.defaultdict(list)
for sub, out in (("FETCH_SIZE", "FETCH_SIZE"), ("WRITE_SIZE", "WRITE_SIZE"), ("sq", "SQ_search")):
    f = os.path.join(src, "prof_%s_%s" % (tag, sub), "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    shutil.copy(f, os.path.join(dst, "%s_pmc_%s.csv" % (tag, out)))
    for r in csv.DictReader(open(f)):
        pmc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(pmc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    lines.append("%s,%s,avg=%.1f,launches=%d" % (k, c, sum(v) / len(v), len(v)))
pg = "mh_project_gather_kernel<7>"
if (pg, "FETCH_SIZE") in pmc and (pg, "WRITE_SIZE") in pmc:
    fetch = 2 * 1024 * sum(pmc[(pg, "FETCH_SIZE")]) / len(pmc[(pg, "FETCH_SIZE")])
    write = 1024 * sum(pmc[(pg, "WRITE_SIZE")]) / len(pmc[(pg, "WRITE_SIZE")])
    lines += ["", "# Derived (MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE doubled for 16-B/lane reads):",
              "# %s: fetch = %.1f MB, write = %.1f MB, traffic = %.1f MB per launch" % (pg, fetch / 1e6, write / 1e6,
                                                                                    (fetch + write) / 1e6),
              "#   algorithmic bytes (SURVEY.md §8d) = %.2f MB per launch; avg duration %.1f us -> %.2f TB/s algorithmic "
              "= %.1f %% of 8 TB/s" % (ALGO / 1e6, dur[pg], ALGO / dur[pg] / 1e6, ALGO / dur[pg] / 1e6 / 8 * 100)]
    json.dump({"_comment": "HBM traffic per launch from rocprofv3 PMC passes (tools/profile_bench.sh + "
                           "tools/summarize_profile.py, profiles/%s_summary.txt); FETCH_SIZE doubled per "
                           "MI355X_MICROARCH.md (16-B/lane reads), units KiB*1024" % tag,
               "round": 1, "tag": tag, "workload": "60 views @ 1920x1080, 5000 points, patch 7",
               pg: {"fetch_bytes": int(fetch), "write_bytes": int(write), "traffic_bytes": int(fetch + write)}},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
sk = next((k for k, c in pmc if k.startswith("mh_search_kernel") and c == "SQ_INSTS_VALU"), None)
if sk:
    g = lambda c: sum(pmc[(sk, c)]) / len(pmc[(sk, c)])   # noqa: E731
    lines += ["# %s: %.0f M VALU wave-instructions per launch, VALU-active %.0f M quad-cycles = %.2f cycles per "
              "instruction; waves resident %.0f M quad-cycles" % (sk, g("SQ_INSTS_VALU") / 1e6,
                                                                 g("SQ_ACTIVE_INST_VALU") / 1e6,
                                                                 4 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU"),
                                                                 g("SQ_WAVE_CYCLES") / 1e6)]
open(os.path.join(dst, "%s_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
