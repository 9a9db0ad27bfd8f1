import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "pcie", (d.get("pcie_inclusive") or {}).get("value"), "serial", d["serial_steps"]["ms_per_step"])
print(" serial", {k:v["mean_ms"] for k,v in d["serial_steps"]["kernels"].items()})
print(" flight", {k:v["mean_ms"] for k,v in d["config"]["kernels"].items()})
