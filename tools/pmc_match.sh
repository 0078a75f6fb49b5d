#!/bin/bash
# rocprofv3 PMC pass over the matching microbenchmark (one counter set per run)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_match
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/tools/microbench.py --what match > $OUT/sq.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_match/sq"
files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in agg.items():
    n = max(cnt[k], 1)
    print(k, "dispatches", n, {c: round(x / n, 1) for c, x in v.items()})
PY
