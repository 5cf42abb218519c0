#!/usr/bin/env python
"""Summary of a rocprofv3 kernel trace of tools/time_full_pass.py (tools/profile_fullpass.sh):
   python tools/summarize_fullpass.py <trace dir> [stages.txt]
Splits the trace into passes (a pass starts with the votes of filter_negative_points) and the LAST pass into its stages
(filter / optimize / refine+volume) by kernel names, and prints per stage: span on the GPU clock, time with at least one
kernel running, idle time, every kernel's calls / total / average, and the largest idle gaps with their neighbours."""
import collections
import csv
import glob
import sys


def short(n):
    return n.split("(")[0].replace("void ", "")[:64]


def load(d):
    fs = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    return rows


def union_busy(rows):
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def gaps(rows, top=12):
    out, end, last = [], None, None
    for s, e, n in rows:
        if end is not None and s > end:
            out.append((s - end, last, n, end))
        if end is None or e > end:
            end, last = e, n
    out.sort(reverse=True)
    return out[:top]


def report(name, rows):
    if not rows:
        print("## %s: no kernels" % name)
        return
    span = max(e for _, e, _ in rows) - rows[0][0]
    busy = union_busy(rows)
    print("## %s: %d launches, span %.3f ms, >=1 kernel running %.3f ms, idle %.3f ms" % (name, len(rows), span / 1e6, busy / 1e6,
                                                                                     (span - busy) / 1e6))
    acc = collections.OrderedDict()
    for s, e, n in rows:
        a = acc.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    print("kernel,calls,total_ms,avg_us,min_us,max_us")
    for n, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%s,%d,%.3f,%.2f,%.2f,%.2f" % (n, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3))
    print("largest idle gaps (us): after -> before")
    t0 = rows[0][0]
    for g, a, b, at in gaps(rows):
        print("  %.1f us at +%.3f ms: %s -> %s" % (g / 1e3, (at - t0) / 1e6, a, b))


def main():
    rows = load(sys.argv[1])
    if not rows:
        print("no kernel trace found under", sys.argv[1])
        return
    # passes: a run of mh_filter_kernel launches that follows a voxel fit (or the start of the trace)
    starts, seen_vox, prev = [], True, ""
    for i, (s, e, n) in enumerate(rows):
        if n.startswith("mh_filter_kernel") and not prev.startswith("mh_filter_kernel") and seen_vox:
            starts.append(i)
            seen_vox = False
        if n.startswith("mh_voxel_key_kernel"):
            seen_vox = True
        prev = n
    print("# %d kernel launches, %d passes found" % (len(rows), len(starts)))
    if not starts:
        return
    last = rows[starts[-1]:]
    # stages of the last pass
    i_opt = next((i for i, r in enumerate(last) if r[2].startswith("mh_project_taps")), len(last))
    i_ref = max((i for i, r in enumerate(last) if r[2].startswith("mh_search3") or r[2].startswith("mh_search_kernel")),
                default=len(last) - 1) + 1
    print("# LAST pass: %d launches, span %.3f ms" % (len(last), (max(e for _, e, _ in last) - last[0][0]) / 1e6))
    report("filter_negative_points", last[:i_opt])
    report("optimize", last[i_opt:i_ref])
    report("refine + volume", last[i_ref:])
    if len(sys.argv) > 2:
        print("## stage timers of the separate MH_TIMING=1 run (device-synchronised wall time, last pass)")
        lines = open(sys.argv[2]).read().splitlines()
        idx = [i for i, l in enumerate(lines) if l.startswith("---- pass")]
        for l in lines[idx[-1]:] if idx else lines[-40:]:
            if l.startswith("[mh-timing]") or l.startswith("{") or l.startswith("----"):
                print(l[:400])


if __name__ == "__main__":
    main()
