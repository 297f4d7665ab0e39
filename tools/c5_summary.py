"""profiles/r3_c5_summary.json: what the million-sphere frame (3840x2160, 16 spp) moves through the memory system under its two
cameras, from the counter digests written by tools/profile_digest.py and the bench lines of the same frames.
usage: python tools/c5_summary.py <across bench line> <above bench line>"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(REPO, "profiles")


def leg(view, prec, line):
    scene = "mega-spheres" + ("-above" if view == "above" else "")
    d = json.load(open(os.path.join(P, "r3_c5_%s_pmc_%s_%s.json" % (view, scene, prec))))
    k = d["kernels"]["k_paths_stepwise"]
    b = line if prec == "f64" else line["f32"]
    rays = line["config"]["rays_per_step"] if prec == "f64" else b["value"] * 1e6 * b["ms_per_step"] * 1e-3
    secs = k["duration_ns"] * 1e-9
    fetch, write = d["hbm_bytes_per_frame"]["fetch_corrected"], d["hbm_bytes_per_frame"]["write"]
    return {
        "kernel_ms_under_counters": 1e3 * secs, "kernel_ms_bench": b["roofline"]["avg_launch_ms"], "Mrays_per_s": b["value"], "rays_per_frame": rays,
        "hbm_fetch_bytes": fetch, "hbm_write_bytes": write, "hbm_bytes_per_ray": (fetch + write) / rays,
        "hbm_GBs": (fetch + write) / secs / 1e9, "hbm_frac_of_8TBs": (fetch + write) / secs / 8e12,
        "l2_requests": k["TCC_REQ_sum"], "l2_hits": k["TCC_HIT_sum"], "l2_misses": k["TCC_MISS_sum"], "l2_hit_rate": k["TCC_HIT_sum"] / k["TCC_REQ_sum"],
        "l1_accesses": k["TCP_TOTAL_CACHE_ACCESSES_sum"], "l1_to_l2_read_requests": k["TCP_TCC_READ_REQ_sum"],
        "requested_bytes_per_ray": (b["roofline"].get("hbm") or b["roofline"])["requested_bytes_per_ray"],
        "valu_insts_per_simd_cycle": k["valu_insts_per_simd_cycle"], "lanes_active_of_64": k["lanes_active_per_valu_cycle"],
        "valu_frac": (b["roofline"].get("valu") or b["roofline"])["frac"],
    }


def main():
    out = {"frame": "mega-spheres, 1 000 001 objects, 3840x2160, 16 spp, depth 50", "views": {}}
    for view, path in (("across", sys.argv[1]), ("above", sys.argv[2])):
        line = json.loads([l for l in open(path) if l.startswith("{")][0])
        with open(os.path.join(P, "r3_c5_%s_bench_line.json" % view), "w") as f:
            f.write(json.dumps(line) + "\n")
        out["views"][view] = {"camera": "(13, 6, 3) looking across the field" if view == "across" else "(0, 1400, 0) looking straight down on all of it",
                              "f64": leg(view, "f64", line), "f32": leg(view, "f32", line)}
    with open(os.path.join(P, "r3_c5_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    for v, d in out["views"].items():
        for p in ("f64", "f32"):
            x = d[p]
            print("%-6s %s: %.0f Mrays/s, HBM %.1f B/ray = %.0f GB/s (%.1f%% of peak), L2 hit %.0f%%, lanes requested %.0f B/ray, valu frac %.3f, %.1f lanes"
                  % (v, p, x["Mrays_per_s"], x["hbm_bytes_per_ray"], x["hbm_GBs"], 100 * x["hbm_frac_of_8TBs"], 100 * x["l2_hit_rate"], x["requested_bytes_per_ray"], x["valu_frac"], x["lanes_active_of_64"]))


if __name__ == "__main__":
    main()
