"""Turn gpurun_out/profile_<tag>/ (written by tools/profile_round.sh) into the files kept under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python3 bench.py`
  profiles/<tag>_bench_line.json    the JSON line that run printed
  profiles/<tag>_pmc_<scene>_<precision>.json   per-launch counters of the render kernels (summed over the dispatches of one frame)
  profiles/pmc_traffic.json         HBM bytes per frame by scene_precision_pipeline key, read by bench.py for roofline.traffic
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies a 128-byte request as 64 bytes)."""
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RENDER_KERNELS = ("k_paths_persistent", "k_paths_stepwise", "k_bounce_fused", "k_trace", "k_shade", "k_gen_primary")


def digest(tag, prec, scene, spp, src, dst):
    tot = {}
    launches = {}
    owner = {}  # counter -> the pass it is taken from (a counter collected in two passes must not be added up twice)
    for d in sorted(glob.glob(os.path.join(src, "pmc_%s_*" % prec))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            seen = set()
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    name = r["Kernel_Name"]
                    kern = next((k for k in RENDER_KERNELS if k in name), None)
                    if kern is None:
                        continue
                    t = tot.setdefault(kern, {})
                    if owner.setdefault((kern, r["Counter_Name"]), d) == d:
                        t[r["Counter_Name"]] = t.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    if (r["Dispatch_Id"], kern) not in seen and os.path.basename(d).endswith("_fetch"):
                        seen.add((r["Dispatch_Id"], kern))
                        launches[kern] = launches.get(kern, 0) + 1
                        t["duration_ns"] = t.get("duration_ns", 0.0) + float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                        t["vgpr"] = int(r["VGPR_Count"]); t["sgpr"] = int(r["SGPR_Count"]); t["lds"] = int(r["LDS_Block_Size"])
    if not tot:
        return None
    pipeline = "persistent"
    key = "%s_%s_%s" % (scene, prec, pipeline)
    out = {"tag": tag, "key": key, "launches": launches, "kernels": tot}
    fetch = sum(t.get("FETCH_SIZE", 0.0) for t in tot.values()) * 1024.0 * 2.0
    write = sum(t.get("WRITE_SIZE", 0.0) for t in tot.values()) * 1024.0
    out["hbm_bytes_per_frame"] = {"fetch_corrected": fetch, "write": write, "total": fetch + write}
    for kern, t in tot.items():
        if t.get("SQ_INSTS_VALU") and t.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs issue vector instructions
            cycles = t["GRBM_GUI_ACTIVE"] / 8.0
            t["valu_insts_per_simd_cycle"] = t["SQ_INSTS_VALU"] / (1024.0 * cycles)
            t["salu_insts_per_cu_cycle"] = t.get("SQ_INSTS_SALU", 0.0) / (256.0 * cycles)
            t["clock_ghz"] = cycles / t["duration_ns"]
        if t.get("SQ_THREAD_CYCLES_VALU") and t.get("SQ_ACTIVE_INST_VALU"):
            t["lanes_active_per_valu_cycle"] = t["SQ_THREAD_CYCLES_VALU"] / t["SQ_ACTIVE_INST_VALU"]
    name = "%s_pmc_%s_%s" % (tag, scene, prec)
    with open(os.path.join(dst, name + ".json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    tp = os.path.join(dst, "pmc_traffic.json")
    traffic = {}
    if os.path.exists(tp):
        with open(tp) as f:
            traffic = json.load(f)
    traffic[key] = {"bytes_per_frame": fetch + write, "spp": spp, "fetch_corrected": fetch, "write": write, "from": name}
    with open(tp, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    print(key, json.dumps(out["hbm_bytes_per_frame"]), {k: {m: v for m, v in t.items() if m in ("valu_insts_per_simd_cycle", "salu_insts_per_cu_cycle", "clock_ghz", "lanes_active_per_valu_cycle", "duration_ns")} for k, t in tot.items()})
    return out


def main():
    tag = sys.argv[1]
    scene = sys.argv[2] if len(sys.argv) > 2 else "cornell-box"
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    src = os.path.join(REPO, "gpurun_out", "profile_" + tag)
    dst = os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)
    suffix = "" if scene == "cornell-box" else "_" + scene
    for f in glob.glob(os.path.join(src, "trace", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, tag + suffix + "_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "bench_line.json")):
        shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, tag + suffix + "_bench_line.json"))
    out = {prec: digest(tag, prec, scene, spp, src, dst) for prec in ("f64", "f32")}
    # the bench line of the trace run read roofline.traffic from the digest of an EARLIER profile: bring it to this one's counters
    # (same frame, same sample count, one launch per step)
    line_path = os.path.join(dst, tag + suffix + "_bench_line.json")
    if os.path.exists(line_path):
        with open(line_path) as f:
            line = json.loads(f.read())
        for prec, leg in (("f64", line), ("f32", line.get("f32") or {})):
            if out[prec] and isinstance(leg.get("roofline"), dict) and leg.get("dtype", prec) == prec:
                leg["roofline"]["traffic"] = out[prec]["hbm_bytes_per_frame"]["total"]
        with open(line_path, "w") as f:
            f.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
