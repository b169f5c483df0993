#!/usr/bin/env python
"""Join rocprofv3 counter_collection.csv + kernel_trace.csv: per kernel-name averages of each counter and duration."""
import csv, sys, collections
def main(d, prefix):
    cc = list(csv.DictReader(open(f"{d}/{prefix}_counter_collection.csv")))
    kt = list(csv.DictReader(open(f"{d}/{prefix}_kernel_trace.csv")))
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in kt}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in cc:
        k = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur_ns"].append(dur.get(r["Dispatch_Id"], 0))
    for k, v in agg.items():
        n = len(v["_dur_ns"])
        print(k)
        for c, vals in sorted(v.items()):
            print("   %-28s avg %.4g  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
