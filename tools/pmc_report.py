"""Summarise rocprofv3 --pmc passes written by tools/pmc_passes.sh: one row per bounce-kernel dispatch."""
import csv
import glob
import json
import os
import sys


def load(outdir):
    rows = {}  # dispatch order -> {counter: value}
    meta = {}
    for f in sorted(glob.glob(os.path.join(outdir, "*", "*counter_collection.csv"))):
        per = {}
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                d = int(r["Dispatch_Id"])
                per.setdefault(d, {"kernel": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                   "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "grid": int(r["Grid_Size"])})
                per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        # keep only the render kernels, number them in dispatch order
        k = 0
        for d in sorted(per):
            name = per[d]["kernel"]
            if "k_bounce" in name or "k_gen_primary" in name or "k_trace" in name or "k_shade" in name:
                rows.setdefault(k, {}).update({c: v for c, v in per[d].items() if c not in ("kernel", "ns", "vgpr", "sgpr", "grid")})
                meta.setdefault(k, per[d])
                k += 1
    return rows, meta


def main():
    outdir = sys.argv[1]
    rows, meta = load(outdir)
    names = sorted({c for r in rows.values() for c in r})
    print("dispatches:", len(rows), "counters:", " ".join(names))
    sel = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 6, 11, 21, 41]
    for k in sel:
        if k not in rows:
            continue
        r, m = rows[k], meta[k]
        print("--- dispatch %d %s  %.1f us vgpr %d sgpr %d grid %d" % (k, m["kernel"][:60], m["ns"] / 1e3, m["vgpr"], m["sgpr"], m["grid"]))
        for c in names:
            if c in r:
                extra = ""
                if c.startswith("SQ_INSTS") and r.get("SQ_INSTS_VALU"):
                    extra = "  (%.3f per VALU instr)" % (r[c] / r["SQ_INSTS_VALU"])
                if c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_ANY",
                         "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") and r.get("SQ_WAVE_CYCLES"):
                    extra = "  (%.3f of wave cycles)" % (r[c] / r["SQ_WAVE_CYCLES"])
                print("   %-28s %16.0f%s" % (c, r[c], extra))
        if r.get("SQ_THREAD_CYCLES_VALU") and r.get("SQ_ACTIVE_INST_VALU"):
            print("   lanes active per VALU cycle: %.1f of 64" % (r["SQ_THREAD_CYCLES_VALU"] / r["SQ_ACTIVE_INST_VALU"]))
    tot = {c: sum(r.get(c, 0.0) for r in rows.values()) for c in names}
    print("--- totals over the render:", json.dumps(tot))


if __name__ == "__main__":
    main()
