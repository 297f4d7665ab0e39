"""Static instruction mix of one kernel of an ISA listing (hipcc -S / -save-temps): counts by class, so that two builds of a
kernel can be compared line by line.  usage: tools/isa_mix.py <file.s> <substring of the mangled kernel name> [...more substrings]"""
import collections
import re
import sys

CLASSES = [
    ("f64 fma/mul/add", r"v_(fma|fmac|mul|add)_f64"), ("f64 min/max", r"v_(min|max)_f64"), ("f64 cmp", r"v_cmp\w*_f64"),
    ("f64 transcendental", r"v_(rcp|rsq|sqrt)_f64"), ("f64 fix-up/scale/ldexp", r"v_(div_\w+|ldexp|frexp\w*)_f64"),
    ("f32 arith", r"v_(pk_)?(fma|fmac|mul|add|sub|subrev|mad|med3|min|max|min3|max3)_f32"), ("f32 cmp", r"v_cmp\w*_f32"),
    ("f32 transcendental", r"v_(rcp|rsq|sqrt|log|exp|sin|cos)\w*_f32"), ("convert", r"v_cvt_"),
    ("select", r"v_cndmask"), ("move", r"v_(mov|accvgpr|swap)"), ("64-bit int multiply", r"v_mad_[ui]64"),
    ("int multiply", r"v_mul_(lo|hi)_"), ("int cmp", r"v_cmp\w*_[ui](16|32|64)"),
    ("lane access", r"v_(readlane|readfirstlane|writelane|mbcnt|permlane)"), ("other vector", r"v_"),
    ("scalar memory", r"s_(load|buffer_load)"), ("scalar wait/nop", r"s_(waitcnt|nop)"), ("scalar branch", r"s_(cbranch|branch|setpc)"),
    ("other scalar", r"s_"), ("LDS", r"ds_"), ("vector memory", r"(global|flat|buffer|scratch)_"),
]


def mix(lines, want):
    start = None
    for i, l in enumerate(lines):
        if start is None and not l.startswith(("\t", " ", ".")) and l.split(":")[0] and all(w in l.split(":")[0] for w in want) and ":" in l:
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            body = lines[start + 1:i]
            break
    else:
        raise SystemExit("kernel not found")
    counts = collections.Counter()
    for l in body:
        t = l.strip()
        if not t or t.startswith((".", ";")) or t.split(" ;")[0].rstrip().endswith(":"):
            continue
        op = t.split()[0]
        for name, pat in CLASSES:
            if re.match(pat, op):
                counts[name] += 1
                break
        else:
            counts["?" + op] += 1
    return lines[start].split(":")[0], counts


if __name__ == "__main__":
    lines = open(sys.argv[1]).read().split("\n")
    name, c = mix(lines, sys.argv[2:])
    print(name)
    vec = sum(v for k, v in c.items() if k not in ("scalar memory", "scalar wait/nop", "scalar branch", "other scalar", "LDS", "vector memory"))
    print("vector ALU %d, scalar %d, LDS %d, vector memory %d" % (vec, sum(c[k] for k in ("scalar memory", "scalar wait/nop", "scalar branch", "other scalar")),
                                                                  c["LDS"], c["vector memory"]))
    for k, v in c.most_common():
        print("  %-26s %5d" % (k, v))
