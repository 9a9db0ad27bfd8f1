"""Summary lines of bench.py output files (or stdin): value, step times, serial kernel times, walker stats."""
import json
import sys


def show(name, text):
    lines = [l for l in text.strip().splitlines() if l.startswith("{")]
    if not lines:
        print(name, "no JSON line")
        return
    d = json.loads(lines[-1])
    c = d["config"]
    print(name, "value", d["value"], "ms", d["ms_per_step"], "pcie", (d.get("pcie_inclusive") or {}).get("value"),
          "serial", d["serial_steps"]["ms_per_step"], "depth", c.get("batches_in_flight"))
    t = d.get("timing") or {}
    print("  timing", {k: t.get(k) for k in ("repeats", "median_ms_per_step", "min_ms_per_step", "max_ms_per_step", "all_ms_per_step")},
          "same", (c.get("same_reads_every_step") or {}).get("spread", {}).get("all_ms_per_step"),
          "pcie", (d.get("pcie_inclusive") or {}).get("spread", {}).get("all_ms_per_step"))
    print("  serial", {k: v["mean_ms"] for k, v in d["serial_steps"]["kernels"].items()})
    print("  flight", {k: v["mean_ms"] for k, v in c["kernels"].items()})
    print("  walker", c.get("walker_phase_stats"), "fast", c.get("fast_hit_slots_per_query"), "mode", c.get("search_mode"),
          "census", c.get("index_census_hits"), "A/chunk", c.get("anchors_per_chunk"), "ovf_q", c.get("overflow_queries"),
          "agree", c.get("contexts_agree"), "census_same", c.get("census_results_identical"), "flags", c.get("chunks_with_capacity_flags"))
    r = d["roofline"]
    print("  roofline frac", r["frac"], "useful", r.get("useful_bytes_frac"), "launch_ms", r["launch_ms"], "traffic", r["traffic"],
          "step_useful_frac", r.get("step_useful_frac"))


if len(sys.argv) > 1:
    for p in sys.argv[1:]:
        try:
            show(p, open(p).read())
        except Exception as e:  # noqa: BLE001
            print(p, "unreadable:", e)
else:
    show("stdin", sys.stdin.read())
