#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and, in SEPARATE passes, the TCC
# byte counters for bench.py's headline workload. Summaries land in gpurun_out/prof_<tag>/.
#   tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --steps 200 --warmup 20 > $OUT/bench_under_trace.json 2> $OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --steps 50 --warmup 5 > $OUT/bench_under_$C.json 2> $OUT/pmc_$C.err
done
python3 - <<PY
import csv, glob, os, json
out = "$OUT"
res = {}
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                            "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])} for r in rows[:6]]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(out + f"/pmc_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_bitmm" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c:
                vals.append(float(r["Counter_Value"]))
    if vals:
        res[c] = {"dispatches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
