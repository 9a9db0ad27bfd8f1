"""Turns gpurun_out/prof/ (raw rocprofv3 output of tools/collect_profiles.sh) into the small
committed files under profiles/: <tag>_kernel_stats.csv, <tag>_pmc_per_launch.csv,
<tag>_bench.json and pmc_traffic.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def short(name):
    n = name.replace("void ", "").replace("smhip::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def main(tag):
    os.makedirs(DST, exist_ok=True)
    for sub, suffix in (("kt", ""), ("kt_serial", "_serial")):
        ks = sorted(glob.glob(os.path.join(SRC, sub, "*", "*kernel_stats.csv")), key=os.path.getmtime)
        if not ks:
            continue
        rows = list(csv.DictReader(open(ks[-1])))  # newest run
        with open(os.path.join(DST, f"{tag}_kernel_stats{suffix}.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    per = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(SRC, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        cc = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
        if not cc:
            continue
        cc = cc[-1:]
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(cc[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            for c, val in v.items():
                per[k][c] = val / max(1, len(disp[k]))
    counters = sorted({c for v in per.values() for c in v})
    with open(os.path.join(DST, f"{tag}_pmc_per_launch.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + counters)
        for k in sorted(per, key=lambda k: -per[k].get("SQ_WAVE_CYCLES", 0)):
            w.writerow([k] + ["%.6g" % per[k].get(c, float("nan")) for c in counters])
    seed = next((k for k in per if k.startswith("SeedWalkKernel<false>")), None)
    out = {}
    if seed and "FETCH_SIZE" in per[seed]:
        fetch_kb, write_kb = per[seed]["FETCH_SIZE"], per[seed].get("WRITE_SIZE", 0.0)
        out = {
            "kernel": seed,
            "seed_lookup_hbm_bytes_per_launch": int((fetch_kb + write_kb) * 1024),
            "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
            "upper_bound_bytes_if_every_read_request_were_128B": int((2 * fetch_kb + write_kb) * 1024),
            "note": "rocprofv3 FETCH_SIZE/WRITE_SIZE (KB) per launch, separate --pmc passes; FETCH_SIZE = "
                    "TCC_EA0_RDREQ x 64 B.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads exactly half of a "
                    "wide coalesced stream; this kernel issues 16-byte gathers, for which the counter is "
                    "uncalibrated: node records are fetched as single 64-byte sectors (counted exactly), point blocks "
                    "as 128-byte requests (counted half), so the figure is the uncorrected count and the true "
                    "bytes lie between it and the upper bound given here.",
        }
        json.dump(out, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
    for name in ("bench_default.json", "kt_bench.json", "kt_serial_bench.json"):
        p = os.path.join(SRC, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(DST, f"{tag}_{name}"))
    print("profiles written:", sorted(os.listdir(DST)))
    print(json.dumps(out, indent=1)[:600])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r1")
