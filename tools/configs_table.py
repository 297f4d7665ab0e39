"""Prints a table of a file of bench.py lines (tools/bench_configs.sh): Grays/s per precision, roofline fraction, CPU figure.
usage: tools/configs_table.py profiles/r2_other_configs.jsonl"""
import json
import sys

for line in open(sys.argv[1]):
    o = json.loads(line)
    r = o.get("roofline") or {}
    f32 = o.get("f32") or {}
    own = o.get("f64_own_tree") or {}
    cpu = o.get("cpu_baseline") or {}
    err = (f32.get("image_error_vs_f64") or {})
    print("%-44s f64 %7.2f (%s %.3f)  f32 %7.2f (%.3f)  f64-own %s  cpu %s Mrays/s  f32 mean err %s  build %.2f s upload %.2f s" % (
        o["metric"].replace("Mrays/s ", ""), o["value"] / 1e3, r.get("bound"), r.get("frac") or 0.0,
        (f32.get("value") or 0) / 1e3, ((f32.get("roofline") or {}).get("frac") or 0.0),
        "%.2f" % (own["value"] / 1e3) if own else "-", "%.1f" % cpu["value"] if cpu else "-",
        "%.1e" % max(err["mean_error_rel"]) if err else "-", o["config"]["scene_build_s"], o["config"]["upload_s"]))
