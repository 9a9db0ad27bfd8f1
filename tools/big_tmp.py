import sys, os, json, subprocess
for cl in (2000000, 6250000):
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--steps", "2", "--contig-len", str(cl)], capture_output=True, text=True).stdout.strip().split("\n")[-1]
    d = json.loads(out); c = d["config"]
    print(cl, d["value"], c["kernels"], "flags", c["chunks_with_capacity_flags"], "ovf", c["overflow_queries"], "Q", c["queries_per_chunk"], "A", c["anchors_per_chunk"])
