"""Turns gpurun_out/prof/ (raw rocprofv3 output of tools/collect_profiles.sh) into the small
committed files under profiles/: <tag>_kernel_stats.csv, <tag>_pmc_per_launch.csv,
<tag>_bench.json and <tag>_pmc_traffic.json (read by bench.py for roofline.traffic)."""
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
    if seed and "TCC_EA0_RDREQ_sum" in per[seed]:
        # profiles/r2_calibration.json: on gfx950 EVERY L2 read miss goes to the fabric as one 128-byte
        # request (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ for 64-byte records, 32-byte records, leaf lines
        # and streams alike), and FETCH_SIZE tallies it at 64 bytes; WRITE_SIZE counts 32-byte write
        # requests exactly.  HBM-side bytes = 128 * read requests + WRITE_SIZE.
        rd = per[seed]["TCC_EA0_RDREQ_sum"]
        write_kb = per[seed].get("WRITE_SIZE", 0.0)
        out = {
            "kernel": seed,
            "seed_lookup_hbm_bytes_per_launch": int(rd * 128 + write_kb * 1024),
            "read_requests_128B": rd, "read_bytes": int(rd * 128),
            "fetch_size_kb_as_reported": per[seed].get("FETCH_SIZE"), "write_size_kb": write_kb,
            "rdreq_128B": per[seed].get("TCC_EA0_RDREQ_128B_sum"), "rdreq_64B": per[seed].get("TCC_EA0_RDREQ_64B_sum"),
            "l2_hit_rate": (per[seed]["TCC_HIT_sum"] / (per[seed]["TCC_HIT_sum"] + per[seed]["TCC_MISS_sum"])
                            if "TCC_HIT_sum" in per[seed] else None),
            "source": f"profiles/{tag}_pmc_per_launch.csv: rocprofv3 --pmc (one counter set per pass, --kernel-trace only) "
                      f"-- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pcie --depth 1; mean per launch of {seed}; "
                      "calibration profiles/r2_calibration.json",
        }
        json.dump(out, open(os.path.join(DST, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    for name in ("bench_default.json", "kt_bench.json", "kt_serial_bench.json"):
        p = os.path.join(SRC, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(DST, f"{tag}_{name}"))
    print("profiles written:", sorted(os.listdir(DST)))
    print(json.dumps(out, indent=1)[:600])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r2")
