"""Condenses rocprofv3 CSV output (kernel trace / counter collection) into small per-kernel summaries for profiles/."""
import csv, glob, os, sys, collections

d, out = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[name][0] += 1; agg[name][1] += dur
    tot = sum(v[1] for v in agg.values())
    with open(out + "_kernel_trace_summary.csv", "w") as o:
        o.write("kernel,calls,total_us,avg_us,percent\n")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"\"{k}\",{n},{t:.1f},{t / n:.2f},{100 * t / tot:.2f}\n")
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        a = agg[name][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    with open(out + "_counters_summary.csv", "a") as o:
        o.write("kernel,counter,dispatches,sum,avg_per_dispatch\n")
        for k, cs in agg.items():
            for c, (n, v) in cs.items():
                o.write(f"\"{k}\",{c},{n},{v:.1f},{v / n:.1f}\n")
print("summaries written to", out)
