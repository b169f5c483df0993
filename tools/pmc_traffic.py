#!/usr/bin/env python
"""Per-launch HBM traffic of one kernel from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide
coalesced streaming reads, so it is doubled (MI355X_MICROARCH.md §HBM).  Writes the JSON that
bench.py reports as roofline.traffic.
"""
import csv, json, sys

def total(d, prefix, counter, kernel):
    s, n = 0.0, 0
    for r in csv.DictReader(open(f"{d}/{prefix}/{prefix}_counter_collection.csv")):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"]); n += 1
    return s, n

def main(d, kernel, out):
    f, nf = total(d, "fetch", "FETCH_SIZE", kernel)
    w, nw = total(d, "write", "WRITE_SIZE", kernel)
    res = {"kernel": kernel, "launches_sampled": nf,
           "fetch_bytes_per_launch": 2.0 * f * 1024 / nf, "write_bytes_per_launch": w * 1024 / nw,
           "traffic_bytes_per_launch": (2.0 * f * 1024 / nf) + (w * 1024 / nw),
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                     "`bench.py --steps 3 --warmup 1`; FETCH_SIZE x2 (gfx950 128-B request correction)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
