#!/bin/bash
# Runs on the GPU box (via gpurun). For every target of tools/profile_targets.py: one rocprofv3 --kernel-trace --stats
# pass, then - in SEPARATE passes, counters never share a run with a trace - --pmc FETCH_SIZE, --pmc WRITE_SIZE and an
# LDS / VALU / MFMA activity pass. The program itself follows `--` (python3, no wrapper). Summaries land in
# gpurun_out/prof_<tag>/summary_<target>.json; copy what is to be judged into profiles/.
# Every pass runs under `timeout 300`: a profiler pass that aborts can sit in its finalisation until the box is taken away (one did, on
# TA_* counters, for 25 minutes).
#   tools/collect_profiles.sh <tag> [targets...]
set -u
TAG=${1:-r06}
shift
TARGETS=${@:-headline popcount w8 gin_single epoch epoch_gin loader pack big big16 big1024}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for T in $TARGETS; do
  REPS=200; [ "$T" = "epoch" -o "$T" = "epoch_gin" -o "$T" = "loader" -o "$T" = "loader_gin" ] && REPS=20; case "$T" in wide*|big*) REPS=50;; esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$T -- python3 $GRAFT_REPO_ROOT/tools/profile_targets.py $T $REPS > $OUT/run_trace_$T.json 2> $OUT/trace_$T.err
  for C in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    N=$(echo $C | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${T}_$N -- python3 $GRAFT_REPO_ROOT/tools/profile_targets.py $T 20 > $OUT/run_pmc_${T}_$N.json 2> $OUT/pmc_${T}_$N.err
  done
  python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from qgtc_ppopp22_amd._build import kernel_source_hash
out, t = "$OUT", "$T"
res = {"target": t, "kernel_source_hash": kernel_source_hash()}
try:
    res["hip_events"] = json.loads([l for l in open(f"{out}/run_trace_{t}.json") if l.startswith("{")][-1])
except Exception as e:
    res["hip_events"] = {"error": str(e)}
for f in glob.glob(f"{out}/trace_{t}/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [{"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                            "max_ns": float(r["MaxNs"]), "total_ns": float(r["TotalDurationNs"]), "pct": float(r["Percentage"])} for r in rows[:12]]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_{t}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if k.startswith("void at::") or "elementwise" in k or "Memset" in k:
            continue
        pmc[k[:110]][r.get("Counter_Name")].append(float(r["Counter_Value"]))
res["pmc_per_dispatch_mean"] = {k: {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()} for k, cs in pmc.items()}
json.dump(res, open(f"{out}/summary_{t}.json", "w"), indent=1)
print(t, "ok", [ (k["name"][:40], k["calls"], k["avg_ns"]) for k in res.get("kernel_stats", [])[:3]])
PY
  python3 $GRAFT_REPO_ROOT/tools/trace_medians.py $OUT $T
done
