#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace CSV to a compact timeline and say how the kernels of concurrent streams overlap.

    tools/timeline.py <dir with *kernel_trace.csv> [--last-ms 120] [--out compact.csv]

Prints, over the last `--last-ms` of the trace (the timed regions of a bench run): per kernel its launches, mean
duration, resources (VGPRs, LDS, workgroup, grid) and how much of its time it ran ALONE / beside the tree walker /
beside anything else; the union busy time of the GPU and the share during which exactly one / two / more kernels
were resident.  The compact CSV (short name, queue, start, end in us from the window's start) is what gets committed."""
import argparse
import csv
import glob
import os
import re
import sys


def short(name):
    n = name.replace("void ", "").replace("smhip::(anonymous namespace)::", "").replace("smhip::", "")
    n = n.split("(")[0]
    return re.sub(r"\s+", "", n).replace(",", ";")[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--last-ms", type=float, default=120.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--walker", default="SeedWalkKernel")
    a = ap.parse_args()
    files = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        sys.exit("no *kernel_trace.csv under " + a.dir)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            try:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                             r.get("Queue_Id", "?"), r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"),
                             r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                             r.get("Scratch_Size", "?"), r.get("Accum_VGPR_Count", "?"), r.get("SGPR_Count", "?")))
            except (KeyError, ValueError):
                continue
    rows.sort()
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(a.last_ms * 1e6)
    win = [r for r in rows if r[0] >= t0]
    if not win:
        sys.exit("empty window")
    base = win[0][0]
    if a.out:
        with open(a.out, "w") as fh:
            fh.write("kernel,queue,start_us,end_us\n")
            for s, e, n, q, *_ in win:
                fh.write("%s,%s,%.1f,%.1f\n" % (n, q, (s - base) / 1e3, (e - base) / 1e3))
    # sweep: at every instant the set of resident kernels
    ev = []
    for i, (s, e, *_r) in enumerate(win):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active = set()
    last_t = ev[0][0]
    alone = {}
    with_walker = {}
    with_other = {}
    conc = {}
    for t, kind, i in ev:
        dt = t - last_t
        if dt > 0 and active:
            k = len(active)
            conc[min(k, 4)] = conc.get(min(k, 4), 0) + dt
            has_walker = any(a.walker in win[j][2] for j in active)
            for j in active:
                n = win[j][2]
                if k == 1:
                    alone[n] = alone.get(n, 0) + dt
                elif has_walker and a.walker not in n:
                    with_walker[n] = with_walker.get(n, 0) + dt
                else:
                    with_other[n] = with_other.get(n, 0) + dt
        last_t = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    span = (win[-1][1] - base) / 1e6
    busy = sum(conc.values()) / 1e6
    print("window %.2f ms, GPU busy (>= 1 kernel resident) %.2f ms; resident kernels 1 / 2 / 3 / 4+: %s ms" % (
        span, busy, " / ".join("%.2f" % (conc.get(k, 0) / 1e6) for k in (1, 2, 3, 4))))
    per = {}
    for s, e, n, q, vg, lds, wg, grid, scr, avg, sg in win:
        p = per.setdefault(n, {"n": 0, "t": 0, "vgpr": vg, "lds": lds, "wg": wg, "grid": grid, "scratch": scr})
        p["n"] += 1
        p["t"] += e - s
    tot = sum(p["t"] for p in per.values())
    print("%-48s %5s %9s %8s %6s | alone  +walker +other (share of own time) | vgpr lds wg grid scratch" % ("kernel", "calls", "avg us", "tot ms", "%"))
    for n, p in sorted(per.items(), key=lambda kv: -kv[1]["t"])[:40]:
        t = p["t"] or 1
        print("%-48s %5d %9.1f %8.2f %6.1f | %5.2f  %5.2f  %5.2f | %s %s %s %s %s" % (
            n, p["n"], p["t"] / p["n"] / 1e3, p["t"] / 1e6, 100.0 * p["t"] / tot, alone.get(n, 0) / t, with_walker.get(n, 0) / t,
            with_other.get(n, 0) / t, p["vgpr"], p["lds"], p["wg"], p["grid"], p["scratch"]))


if __name__ == "__main__":
    main()
