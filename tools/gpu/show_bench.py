import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 5))
r = d.get("roofline") or {}
print("per_kernel_ms", r.get("per_kernel_ms"))
if "multi_chain" in d:
    print("multi_chain", round(d["multi_chain"]["value"]), "chains_sweep", [round(c["value"]) for c in d["chains_sweep"]], "config5", round(d["config5"]["value"]))
    oc = d["other_configs"]
    print("config3", [round(x, 4) for x in oc["config3"]["ms_per_sweep_runs"]], "config4", [round(x, 4) for x in oc["config4"]["ms_per_sweep_runs"]])
