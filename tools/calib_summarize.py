"""Reduces the raw rocprofv3 counter CSVs of tools/calibrate_traffic.sh (gpurun_out/calib/) to
per-kernel, per-dispatch counter means and compares the calibration kernels with their known byte
counts.  Writes gpurun_out/calib/summary.json on the GPU box; `python tools/calib_summarize.py commit`
copies the result to profiles/r3_calibration.json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "calib")


def short(name):
    n = name.replace("void ", "").replace("smhip::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def reduce_dir(d):
    per = {}
    cc = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not cc:
        return per
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(cc[-1])):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        per[k] = {c: val / max(1, len(disp[k])) for c, val in v.items()}
        per[k]["_dispatches"] = len(disp[k])
    return per


def durations(d):
    out = collections.defaultdict(list)
    kt = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)
    if not kt:
        return {}
    for r in csv.DictReader(open(kt[-1])):
        out[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    micro, bench, dur = collections.defaultdict(dict), collections.defaultdict(dict), {}
    for d in sorted(glob.glob(os.path.join(SRC, "micro_*"))):
        if os.path.isdir(d):
            for k, v in reduce_dir(d).items():
                micro[k].update(v)
    for d in sorted(glob.glob(os.path.join(SRC, "bench_*"))):
        if os.path.isdir(d):
            for k, v in reduce_dir(d).items():
                bench[k].update(v)
            for k, v in durations(d).items():
                dur.setdefault(k, []).append(v)
    known = json.load(open(os.path.join(SRC, "known_bytes.json")))
    calib = {}
    for k, kb in known.items():
        m = micro.get(k, {})
        row = {"known": kb, "counters": m}
        rd, wr = kb.get("read_bytes", 0), kb.get("write_bytes", 0)
        if rd and "FETCH_SIZE" in m:
            row["FETCH_SIZE_bytes_over_known_read"] = m["FETCH_SIZE"] * 1024 / rd
        if rd and "TCC_EA0_RDREQ_DRAM_32B_sum" in m:
            row["RDREQ_DRAM_32B_x32_over_known_read"] = m["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32 / rd
        if rd and "TCC_EA0_RDREQ_sum" in m:
            row["RDREQ_x64_over_known_read"] = m["TCC_EA0_RDREQ_sum"] * 64 / rd
        if rd and "TCC_EA0_RDREQ_64B_sum" in m and "TCC_EA0_RDREQ_128B_sum" in m:
            row["RDREQ_64Bx64_plus_128Bx128_over_known_read"] = (m["TCC_EA0_RDREQ_64B_sum"] * 64 +
                                                                 m["TCC_EA0_RDREQ_128B_sum"] * 128) / rd
        if rd and "TCC_READ_SECTORS_sum" in m:
            row["READ_SECTORS_x32_over_known_read"] = m["TCC_READ_SECTORS_sum"] * 32 / rd
        if wr and "TCC_WRITE_SECTORS_sum" in m:
            row["WRITE_SECTORS_x32_over_known_write"] = m["TCC_WRITE_SECTORS_sum"] * 32 / wr
        if wr and "WRITE_SIZE" in m:
            row["WRITE_SIZE_bytes_over_known_write"] = m["WRITE_SIZE"] * 1024 / wr
        if wr and "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" in m:
            row["WRREQ_DRAM_32B_x32_over_known_write"] = m["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] * 32 / wr
        calib[k] = row
    walker = {k: v for k, v in bench.items() if k.startswith("SeedWalkKernel") or k in ("SortKernel",) or k.startswith("Chain")}
    out = {"calibration_kernels": calib, "bench_kernels": walker,
           "bench_kernel_mean_ns": {k: sum(v) / len(v) for k, v in dur.items()
                                    if k.startswith("SeedWalkKernel") or k.startswith("Sort") or k.startswith("Chain")}}
    json.dump(out, open(os.path.join(SRC, "summary.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk.endswith("known_read") or kk.endswith("known_write")}
                      for k, v in calib.items()}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "commit":
        shutil.copy(os.path.join(SRC, "summary.json"), os.path.join(ROOT, "profiles", "r3_calibration.json"))
    else:
        main()
