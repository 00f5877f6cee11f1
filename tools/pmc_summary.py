#!/usr/bin/env python
"""pmc_summary.py <fetch_dir> <write_dir> <out.json> -- per-kernel HBM bytes per launch from two rocprofv3 PMC passes
(FETCH_SIZE and WRITE_SIZE collected in separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes).
rocprofv3 reports both counters in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request and is doubled here,
WRITE_SIZE is taken as reported."""
import csv
import glob
import json
import os
import sys


def kernel_name(raw):
    """'void (anonymous namespace)::k_slice_t_many<20, true, true>(PcManyRec const*, int, int)' -> 'k_slice_t_many<20, true, true>':
    the return type, namespaces written with parentheses and the argument list go; template arguments stay."""
    s = raw.strip()
    if s.startswith("void "):
        s = s[5:].strip()
    s = s.replace("(anonymous namespace)::", "")
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):          # the argument list opens at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return s[:cut].strip()


def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = kernel_name(r["Kernel_Name"])
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    cmd = sys.argv[4] if len(sys.argv) > 4 else ""
    F, W = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(F) | set(W)):
        nf, sf = F.get(k, [0, 0.0]); nw, sw = W.get(k, [0, 0.0])
        n = max(nf, nw, 1)
        fk, wk = sf / max(nf, 1), sw / max(nw, 1)
        kernels[k] = {"launches": n, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
    json.dump({"command": cmd,
               "unit": "KB per launch as reported by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled in "
                       "`hbm_bytes_per_launch` (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as reported",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
        print(f"{k:40s} {v['launches']:6d} launches  {v['hbm_bytes_per_launch'] / 1e6:10.3f} MB/launch")


if __name__ == "__main__":
    main()
