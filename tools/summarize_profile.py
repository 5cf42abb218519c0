#!/usr/bin/env python
"""Turn the raw rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<tag>*) into the tracked artefacts under
profiles/: <tag>_kernel_stats.csv, <tag>_pmc_<COUNTER>.csv, <tag>_summary.txt and traffic.json (what bench.py reports as
roofline*.traffic and roofline.valu_issue).  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled for the
16-byte-per-lane reads of the gather kernels, as MI355X_MICROARCH.md prescribes for gfx950.
    python tools/summarize_profile.py r02a
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
V, N, P = 60, 5000, 49
ALGO_PG = 2 * V * N * (12 * P + 20) + 12 * N


def find(sub, name):
    hits = glob.glob(os.path.join(src, "prof_%s%s" % (tag, sub), "**", name), recursive=True)
    return hits[0] if hits else None


def short(n):
    return n.split("(")[0].replace("void ", "").strip()


lines = ["# profile summary %s -- rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 "
         "--warmup 3 --no-cpu --no-secondary --streams 1" % tag,
         "# MI355X (gfx950), synthetic 60 views @ 1920x1080, 5000 points / iteration, patch 7, ONE HIP stream",
         "# (the headline run alternates two streams; here kernels do not overlap, so durations are per kernel)", "",
         "kernel,calls,avg_us,min_us,max_us,pct_of_gpu_time"]
stats = find("", "bench_kernel_stats.csv")
dur = {}
if stats:
    shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
    for r in csv.DictReader(open(stats)):
        if "mh_" in r["Name"]:
            name = short(r["Name"])
            dur[name] = float(r["AverageNs"]) / 1e3
            lines.append("%s,%s,%.2f,%.2f,%.2f,%s" % (name, r["Calls"], dur[name], float(r["MinNs"]) / 1e3,
                                                      float(r["MaxNs"]) / 1e3, r["Percentage"]))
lines += ["", "# PMC passes (separate runs, `--pmc <counters> --kernel-include-regex mh_...`, per-launch averages)"]
pmc = collections.defaultdict(list)
for sub, out in (("_FETCH_SIZE", "FETCH_SIZE"), ("_WRITE_SIZE", "WRITE_SIZE"), ("_sq", "SQ_search"), ("_sq2", "SQ_search2")):
    f = find(sub, "pmc_counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, os.path.join(dst, "%s_pmc_%s.csv" % (tag, out)))
    for r in csv.DictReader(open(f)):
        pmc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(pmc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    lines.append("%s,%s,avg=%.1f,launches=%d" % (k, c, sum(v) / len(v), len(v)))


def avg(k, c):
    v = pmc.get((k, c))
    return sum(v) / len(v) if v else None


facts = {"_comment": "facts from rocprofv3 passes of `python bench.py` (tools/profile_bench.sh + tools/summarize_profile.py, "
                     "profiles/%s_summary.txt); FETCH_SIZE doubled per MI355X_MICROARCH.md (16-B/lane reads), KiB*1024" % tag,
         "round": 2, "tag": tag, "workload": "60 views @ 1920x1080, 5000 points, patch 7"}
lines += ["", "# Derived (MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE doubled for 16-B/lane reads):"]
names = sorted({k for k, c in pmc})
for short, prefix in (("mh_project_gather_kernel<7>", "mh_project_gather_kernel<7"),
                      ("mh_project_taps_kernel<7>", "mh_project_taps_kernel<7"),
                      ("mh_search3_kernel<256>", "mh_search3_kernel<256")):
    k = next((n for n in names if n.startswith(prefix)), None)
    if k is None:
        continue
    fs, ws = avg(k, "FETCH_SIZE"), avg(k, "WRITE_SIZE")
    if fs is None or ws is None:
        continue
    fetch, write = 2 * 1024 * fs, 1024 * ws
    facts[short] = {"fetch_bytes": int(fetch), "write_bytes": int(write), "traffic_bytes": int(fetch + write)}
    lines.append("# %s: fetch = %.1f MB, write = %.1f MB, traffic = %.1f MB per launch; avg duration %.1f us"
                 % (k, fetch / 1e6, write / 1e6, (fetch + write) / 1e6, dur.get(k, float("nan"))))
    if k.startswith("mh_project_gather") and k in dur:
        lines.append("#   algorithmic bytes (SURVEY.md §8d) = %.2f MB per launch -> %.2f TB/s algorithmic = %.1f %% of 8 TB/s"
                     % (ALGO_PG / 1e6, ALGO_PG / dur[k] / 1e6, ALGO_PG / dur[k] / 1e6 / 8 * 100))
sk = next((k for k, c in pmc if k.startswith(("mh_search3_kernel", "mh_search2_kernel")) and c == "SQ_INSTS_VALU"), None)
if sk:
    insts, waves, wcyc = avg(sk, "SQ_INSTS_VALU"), avg(sk, "SQ_WAVES"), avg(sk, "SQ_WAVE_CYCLES")
    t_us = dur.get(sk)
    issue = {"kernel": sk, "valu_wave_instructions_per_launch": int(insts), "waves_per_launch": int(waves or 0)}
    if t_us:
        per_simd_ns = t_us * 1e3 * 1024 / insts          # 256 CU x 4 SIMD
        issue.update(launch_us=round(t_us, 1), ns_per_valu_instruction_per_simd=round(per_simd_ns, 3),
                     note="the tap body's own rate, all SIMDs busy, is 1.02 ns per instruction (tools/ubench/valu3.hip: "
                          "69.5 cycles@2.4GHz per 28 instructions); the ratio is the kernel's VALU issue utilisation",
                     valu_issue_utilisation=round(1.02 / per_simd_ns, 3))
    facts["search_valu_issue"] = issue
    lines.append("# %s: %.0f M VALU wave-instructions per launch over 1024 SIMDs in %.1f us = %.3f ns per instruction per SIMD"
                 % (sk, insts / 1e6, t_us or float("nan"), (t_us or float("nan")) * 1e3 * 1024 / insts))
    if wcyc:
        lines.append("#   waves resident %.0f M quad-cycles (SQ_WAVE_CYCLES), %d waves" % (wcyc / 1e6, waves or 0))
json.dump(facts, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
open(os.path.join(dst, "%s_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
