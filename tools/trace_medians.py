"""Per-kernel median / p90 of the dispatch durations in a rocprofv3 --kernel-trace run (its --stats table has average, minimum and
maximum only; under the tracer a 3 us dispatch is stretched by a box-dependent amount, which moves the average of every dispatch alike -
the median says whether an average is outliers or the whole distribution). Adds `median_ns` / `p90_ns` to the kernel_stats rows of
summary_<target>.json.   usage: trace_medians.py <prof dir> <target>"""
import csv, glob, json, statistics, sys

out, t = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(f"{out}/trace_{t}/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur.setdefault(r["Kernel_Name"][:110], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
p = f"{out}/summary_{t}.json"
s = json.load(open(p))
for k in s.get("kernel_stats", []):
    d = sorted(dur.get(k["name"], []))
    if d:
        k["median_ns"] = statistics.median(d)
        k["p90_ns"] = d[min(len(d) - 1, int(0.9 * len(d)))]
json.dump(s, open(p, "w"), indent=1)
