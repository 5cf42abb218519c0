#!/usr/bin/env python
"""Copy a round's rocprofv3 summaries (tools/profile_round.sh, tools/profile_gabor_round.sh -> gpurun_out/<tag>_summary.txt) into
profiles/ and rebuild profiles/traffic.json (what bench.py reports as roofline*.traffic / mfma_busy / valu_issue) from them.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled for kernels that read 16 bytes per lane (the record gathers), as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950; kernels that read 2-8 bytes per lane are taken as reported
(uncalibrated widths, stated in the summary).
    python tools/summarize_round.py r04      # after the three profile scripts have run on the GPU box with tags r04_*"""
import collections
import json
import os
import re
import shutil
import sys

RND = sys.argv[1] if len(sys.argv) > 1 else "r04"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def parse(tag):
    path = os.path.join(SRC, "%s_summary.txt" % tag)
    if not os.path.exists(path):
        return None, None, None
    dur, pmc = {}, collections.defaultdict(dict)
    for ln in open(path):
        ln = ln.rstrip("\n")
        m = re.match(r"^(.*),(\d+),([\d.]+),([\d.]+),([\d.]+),([\d.e+-]+)$", ln)
        if m and not ln.startswith("kernel,"):
            dur[m.group(1)] = float(m.group(3))
            continue
        m = re.match(r"^(.*),([A-Z_0-9a-z]+),avg=([\d.e+-]+),launches=(\d+)$", ln)
        if m:
            pmc[m.group(1)][m.group(2)] = float(m.group(3))
            continue
        m = re.match(r"^(.*?)\s+calls=(\d+) avg_ns=([\d.]+)", ln)      # the Gabor script's trace lines
        if m:
            dur[m.group(1).split("(")[0].replace("void ", "").strip()] = float(m.group(3)) / 1e3
    return path, dur, pmc


def find(d, prefix):
    return next((k for k in d if k.replace("void ", "").startswith(prefix)), None)


facts = {"_comment": "facts from rocprofv3 passes (tools/profile_round.sh, tools/profile_gabor_round.sh -> tools/summarize_round.py; "
                     "profiles/%s_*_summary.txt)" % RND + "; FETCH_SIZE / WRITE_SIZE in KiB * 1024, FETCH_SIZE doubled for the kernels "
                     "that read 16 B per lane (MI355X_MICROARCH.md)",
         "round": int(RND[1:]), "workload": "60 views @ 1920x1080, 5000 points, patch 7"}
notes = []
for tag, pre, kernels in ((RND + "_main", "", (("mh_project_gather_kernel<7>", "mh_project_gather_kernel<7", 2),
                                            ("mh_project_taps_kernel<7>", "mh_project_taps2_kernel<7", 2),
                                            ("mh_search3_kernel<256>", "mh_search3_kernel<256", 2))),
                          (RND + "_8bit", "8bit:", (("mh_project_taps_kernel<7>", "mh_project_taps_codes_kernel<7", 1),
                                                 ("mh_project_gather_kernel<7>", "mh_project_gather_kernel<7", 2),
                                                 ("mh_search3_kernel<256>", "mh_search3_kernel<256", 2)))):
    path, dur, pmc = parse(tag)
    if path is None:
        continue
    shutil.copy(path, os.path.join(DST, "%s_summary.txt" % tag))
    b = os.path.join(SRC, "%s_bench.json" % tag)
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(DST, "%s_bench.json" % tag))
    for short, prefix, fmul in kernels:
        k = find(pmc, prefix)
        if not k or "FETCH_SIZE" not in pmc[k] or "WRITE_SIZE" not in pmc[k]:
            continue
        fetch, write = fmul * 1024 * pmc[k]["FETCH_SIZE"], 1024 * pmc[k]["WRITE_SIZE"]
        facts[pre + short] = {"fetch_bytes": int(fetch), "write_bytes": int(write), "traffic_bytes": int(fetch + write),
                              "launch_us_trace": dur.get(find(dur, prefix))}
        notes.append("%s%s: fetch %.1f MB (x%d), write %.1f MB, %.1f us" % (pre, k, fetch / 1e6, fmul, write / 1e6,
                                                                          dur.get(find(dur, prefix), float("nan"))))
    k = find(pmc, "mh_search3_kernel<256")
    if k and "SQ_INSTS_VALU" in pmc[k] and find(dur, "mh_search3_kernel<256"):
        t_us, insts = dur[find(dur, "mh_search3_kernel<256")], pmc[k]["SQ_INSTS_VALU"]
        per = t_us * 1e3 * 1024 / insts
        facts[pre + "search_valu_issue" if pre else "search_valu_issue"] = {
            "kernel": k, "valu_wave_instructions_per_launch": int(insts), "waves_per_launch": int(pmc[k].get("SQ_WAVES", 0)),
            "launch_us": round(t_us, 1), "ns_per_valu_instruction_per_simd": round(per, 3),
            "note": "the tap body's own rate, all SIMDs busy, is 1.02 ns per instruction (tools/ubench/valu3.hip); the ratio "
                    "is the kernel's VALU issue utilisation", "valu_issue_utilisation": round(1.02 / per, 3)}
path, dur, pmc = parse(RND + "_gabor")
if path is None:
    # no Gabor profile this round (the kernels did not change): the facts of the last round that made one are carried over, tagged
    try:
        prev = json.load(open(os.path.join(DST, "traffic.json")))
        if "gabor_stage" in prev:
            facts["gabor_stage"] = dict(prev["gabor_stage"], profiled_in_round=prev["gabor_stage"].get("profiled_in_round", prev.get("round")))
    except Exception:
        pass
if path is not None:
    shutil.copy(path, os.path.join(DST, RND + "_gabor_summary.txt"))
    tot = 0.0
    per = {}
    for name in ("mh_dog_vert_kernel", "mh_dog_horz_kernel", "mh_gabor_mfma2_kernel", "mh_gabor_finish_kernel"):
        k = find(pmc, name)
        if not k:
            continue
        f, w = pmc[k].get("FETCH_SIZE"), pmc[k].get("WRITE_SIZE")
        if f is None or w is None:
            continue
        per[name] = int(1024 * (f + w))          # 1-8 B per lane reads: taken as reported
        tot += 1024 * (f + w)
    g = {"traffic_bytes": int(tot), "traffic_bytes_per_kernel": per,
         "bank_traffic_bytes": per.get("mh_gabor_mfma2_kernel"),
         "launch_us_trace": {n: dur.get(find(dur, n)) for n in ("mh_dog_vert_kernel", "mh_dog_horz_kernel",
                                                                "mh_gabor_mfma2_kernel", "mh_gabor_finish_kernel")}}
    k = find(pmc, "mh_gabor_mfma2_kernel")
    if k and "SQ_VALU_MFMA_BUSY_CYCLES" in pmc[k] and "GRBM_GUI_ACTIVE" in pmc[k]:
        cyc = pmc[k]["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
        g["mfma_busy"] = round(pmc[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)
        g["kernel_cycles"] = int(cyc)
        notes.append("gabor: MFMA pipe busy %.1f %% of the kernel's %.2f M cycles" % (100 * g["mfma_busy"], cyc / 1e6))
    facts["gabor_stage"] = g
    notes.append("gabor stage: %.1f MB HBM traffic per view (%s)" % (tot / 1e6, {k: round(v / 1e6, 1) for k, v in per.items()}))
# the work counters of the profiled launches (bench.py quotes this file only when its own launches report the same ones)
at = {"_comment": "the launch's own work counters when these PMC passes were made (bench.py reads the same counters back from "
                  "its launches and drops every field copied from this file when they differ: a stale file cannot decorate "
                  "another workload)"}
for tag, pre in ((RND + "_main", ""), (RND + "_8bit", "8bit:")):
    b = os.path.join(SRC, "%s_bench.json" % tag)
    if os.path.exists(b) and os.path.getsize(b):
        d = json.load(open(b))
        d = d.get("secondary_8bit_maps", d) if (pre and "roofline" not in d) else d
        front = next((k for k in d.get("roofline_kernels", []) if "visible_pairs" in k), {})
        at.update({pre + "visible_pairs": front.get("visible_pairs"), pre + "taps_written": front.get("taps_written"),
                   pre + "pair_evals_executed": d.get("roofline", {}).get("pair_evals_executed")})
if len(at) > 1:
    facts["profiled_at"] = at
json.dump(facts, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print("\n".join(notes))
