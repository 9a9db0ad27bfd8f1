"""Turns gpurun_out/prof_<tag>/ (raw rocprofv3 output of tools/collect_profiles.sh) into the small
committed files under profiles/: <tag>_kernel_stats[_serial].csv, <tag>_pmc_per_launch.csv, <tag>_*bench.json
and <tag>_pmc_traffic.json (read by bench.py for roofline.traffic when its workload has the shape recorded there).

  on the GPU box : python3 tools/summarize_profiles.py <tag> --reduce-only   (writes gpurun_out/prof_<tag>/reduced/;
                   the raw per-dispatch CSVs are then deleted so that the merge-back stays small)
  here           : python3 tools/summarize_profiles.py <tag>                 (copies reduced/ into profiles/)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "profiles")


def short(name):
    n = name.replace("void ", "").replace("smhip::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def parse_shape(text):
    """bench.py shape arguments -> the dict bench.py compares its own arguments with."""
    shape = {"reads": 10000, "contigs": 16, "contig_len": 750000}
    t = text.split()
    for k, name in (("--reads", "reads"), ("--contigs", "contigs"), ("--contig-len", "contig_len")):
        if k in t:
            shape[name] = int(t[t.index(k) + 1])
    return shape


# (the production search kernel under its names over the rounds; "<2, 128, false>": the set walk of a large batch, round 5
# -- ", true>" is the same walk with a query's visits shared between the lanes of a wave, batches of up to 4 096 chunks)
WALKERS = ("SeedWalkKernel<2, 128, false>", "SeedWalkKernel<2, 128, true>", "SeedWalkKernel<2, 128>", "SeedWalkKernel<2>",
           "SeedWalkKernel<0, 256, false>", "SeedWalkKernel<0, 256>", "SeedWalkKernel<0>", "SeedWalkKernel<false>")


def traffic_json(tag, red, per, shape_txt):
    """<tag>_pmc_traffic.json from the per-launch counter means: the production search kernel of the run -- the set
    walk (<2, ...>) when it ran, else the walk in the reference's order (<0, ...>: large references; "<false>" in rounds 1-2)."""
    seed = next((w for w in WALKERS for k in per if k == w and per[k].get("TCC_EA0_RDREQ_sum", 0) > 1e6), None)
    if not seed:
        return {}
    # profiles/r3_calibration.json: on gfx950 EVERY L2 read miss goes to the fabric as one 128-byte
    # request (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ for 64-byte records, 32-byte records, leaf lines
    # and streams alike), and FETCH_SIZE tallies it at 64 bytes; WRITE_SIZE counts 32-byte write
    # requests exactly.  Fabric-side bytes = 128 * read requests + WRITE_SIZE.
    rd = per[seed]["TCC_EA0_RDREQ_sum"]
    write_kb = per[seed].get("WRITE_SIZE", 0.0)
    ks = os.path.join(red, f"{tag}_kernel_stats_serial.csv")
    avg_ns = None
    if os.path.exists(ks):
        for r in csv.DictReader(open(ks)):
            if r["kernel"] == seed:
                avg_ns = float(r.get("avg_ns_main") or r["avg_ns"])  # (the launches of the batch's size)
    import hashlib
    h = hashlib.sha256()
    for f in ("sigmap_amd/csrc/seed_lookup.hip", "sigmap_amd/csrc/common.h"):
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    prev = os.path.join(red, f"{tag}_pmc_traffic.json")
    sha = h.hexdigest()[:16]
    if os.path.exists(prev):  # (re-derived here from the reduced table: keep the hash of the sources that were profiled)
        try:
            sha = json.load(open(prev)).get("source_sha16") or sha
        except Exception:
            pass
    out = {
        "kernel": seed,
        "source_sha16": sha,
        "shape": parse_shape(shape_txt), "bench_shape_args": shape_txt,
        "seed_lookup_hbm_bytes_per_launch": int(rd * 128 + write_kb * 1024),
        "read_requests_128B": rd, "read_bytes": int(rd * 128),
        "fetch_size_kb_as_reported": per[seed].get("FETCH_SIZE"), "write_size_kb": write_kb,
        "rdreq_128B": per[seed].get("TCC_EA0_RDREQ_128B_sum"), "rdreq_64B": per[seed].get("TCC_EA0_RDREQ_64B_sum"),
        "l2_hit_rate": (per[seed]["TCC_HIT_sum"] / (per[seed]["TCC_HIT_sum"] + per[seed]["TCC_MISS_sum"])
                        if "TCC_HIT_sum" in per[seed] else None),
        "launch_avg_ns_kernel_trace_serial": avg_ns,
        "what": "L2-miss (fabric) bytes: 128 B x TCC_EA0_RDREQ + WRITE_SIZE.  Infinity-Cache (MALL) hits are counted in "
                "these requests, so this is an UPPER bound on HBM traffic",
        "source": f"profiles/{tag}_pmc_per_launch.csv: rocprofv3 --pmc (one counter set per pass, --kernel-trace only) "
                  f"-- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pcie --depth 1 {shape_txt}; mean per launch of "
                  f"{seed} (launches of the batch's size: the small census launch is left out); calibration profiles/r3_calibration.json",
    }
    json.dump(out, open(os.path.join(red, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    return out


def reduce(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    red = os.path.join(src, "reduced")
    os.makedirs(red, exist_ok=True)
    for sub, suffix in (("kt", ""), ("kt_serial", "_serial")):
        ks = sorted(glob.glob(os.path.join(src, sub, "*", "*kernel_stats.csv")), key=os.path.getmtime)
        if not ks:
            continue
        rows = list(csv.DictReader(open(ks[-1])))  # newest run
        # avg_ns_main: the mean over the launches of the batch's size only -- a kernel is also launched on small
        # inputs (the census of the index: 4096 queries through the walker), which rocprof's own average includes;
        # launches shorter than HALF the kernel's median are left out (from the kernel trace, when it is there; round 4's
        # quarter kept the walker's census launch -- 1.96 of 5.6 ms -- in an average quoted as 5.24 ms)
        main = {}
        kt = sorted(glob.glob(os.path.join(src, sub, "*", "*kernel_trace.csv")), key=os.path.getmtime)
        if kt:
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(kt[-1])):
                dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            for k, v in dur.items():
                med = sorted(v)[len(v) // 2]
                kept = [x for x in v if x >= 0.5 * med]
                main[k] = (sum(kept) / len(kept), len(kept))
        with open(os.path.join(red, f"{tag}_kernel_stats{suffix}.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns", "avg_ns_main", "calls_main"])
            for r in rows:
                m = main.get(short(r["Name"]), (float(r["AverageNs"]), int(r["Calls"])))
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"],
                            "%.1f" % m[0], m[1]])
    per = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        cc = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
        if not cc:
            continue
        # per dispatch, then the mean over the dispatches of the batch's size: those whose first counter reaches a
        # quarter of the kernel's median (the census of the index sends 4096 queries through the walker: not a step)
        by_disp = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
        for r in csv.DictReader(open(cc[-1])):
            by_disp[short(r["Kernel_Name"])][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, disp in by_disp.items():
            names = sorted({c for v in disp.values() for c in v})
            lead = names[0]
            vals = sorted(v.get(lead, 0.0) for v in disp.values())
            med = vals[len(vals) // 2]
            kept = [v for v in disp.values() if v.get(lead, 0.0) >= 0.25 * med] or list(disp.values())
            for c in names:
                per[k][c] = sum(v.get(c, 0.0) for v in kept) / len(kept)
    counters = sorted({c for v in per.values() for c in v})
    with open(os.path.join(red, f"{tag}_pmc_per_launch.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + counters)
        for k in sorted(per, key=lambda k: -per[k].get("FETCH_SIZE", 0)):
            w.writerow([k] + ["%.6g" % per[k].get(c, float("nan")) for c in counters])
    shape_txt = open(os.path.join(src, "shape.txt")).read().strip() if os.path.exists(os.path.join(src, "shape.txt")) else ""
    out = traffic_json(tag, red, per, shape_txt)
    for name in ("bench_default.json", "kt_bench.json", "kt_serial_bench.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(red, f"{tag}_{name}"))
    print(json.dumps(out, indent=1)[:900])
    print("reduced:", sorted(os.listdir(red)), "kernels with counters:", len(per))


def commit(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    red = os.path.join(src, "reduced")
    os.makedirs(DST, exist_ok=True)
    pl = os.path.join(red, f"{tag}_pmc_per_launch.csv")
    if os.path.exists(pl):  # (re)derive the traffic summary from the reduced counter table
        per = {}
        for r in csv.DictReader(open(pl)):
            per[r["kernel"]] = {k: float(v) for k, v in r.items() if k != "kernel" and v not in ("", "nan")}
        shape_txt = open(os.path.join(src, "shape.txt")).read().strip() if os.path.exists(os.path.join(src, "shape.txt")) else ""
        traffic_json(tag, red, per, shape_txt)
    for f in sorted(os.listdir(red)):
        shutil.copy(os.path.join(red, f), os.path.join(DST, f))
    print("profiles written:", sorted(f for f in os.listdir(DST) if f.startswith(tag)))


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r3_yeast"
    if "--reduce-only" in sys.argv:
        reduce(tag)
    else:
        if not os.path.isdir(os.path.join(ROOT, "gpurun_out", "prof_" + tag, "reduced")):
            reduce(tag)
        commit(tag)
