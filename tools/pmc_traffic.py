#!/usr/bin/env python
"""Per-launch HBM traffic of the GEMM kernel from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), overall and
per launch shape (grid size: the decode step's F/A, B and D launches have their own workgroup counts; the prologue
products have theirs).

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streaming reads,
so it is doubled (MI355X_MICROARCH.md section HBM).  Writes the JSON that bench.py reports as roofline.traffic.
"""
import csv, json, sys
from collections import defaultdict


def rows(d, prefix, counter, kernel):
    out = defaultdict(lambda: [0.0, 0])
    path = f"{d}/{prefix}/{prefix}_counter_collection.csv"
    for r in csv.DictReader(open(path)):
        if any(k in r["Kernel_Name"] for k in kernel.split("|")) and r["Counter_Name"] == counter:
            g = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0) // max(1, int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 256))
            e = out[g]
            e[0] += float(r["Counter_Value"]); e[1] += 1
    return out


def main(d, kernel, out):
    f, w = rows(d, "fetch", "FETCH_SIZE", kernel), rows(d, "write", "WRITE_SIZE", kernel)
    ft, fn = sum(v[0] for v in f.values()), sum(v[1] for v in f.values())
    wt, wn = sum(v[0] for v in w.values()), sum(v[1] for v in w.values())
    res = {"kernel": kernel, "launches_sampled": fn,
           "fetch_bytes_per_launch": 2.0 * ft * 1024 / fn, "write_bytes_per_launch": wt * 1024 / wn,
           "traffic_bytes_per_launch": (2.0 * ft * 1024 / fn) + (wt * 1024 / wn),
           "by_workgroups": {str(g): {"launches": f[g][1], "fetch_MB": round(2.0 * f[g][0] * 1024 / f[g][1] / 1e6, 2),
                                      "write_MB": round(w[g][0] * 1024 / max(w[g][1], 1) / 1e6, 2) if g in w else None}
                             for g in sorted(f, key=lambda g: -f[g][1])},
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                     "`bench.py --steps 3 --warmup 1`; FETCH_SIZE x2 (gfx950 128-B request correction); by_workgroups = the same "
                     "per launch shape (workgroups in the grid)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
