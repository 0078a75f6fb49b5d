#!/bin/bash
# HBM traffic of the loss, transform and dense-regime kernels from the rocprofv3 PMC counters (FETCH_SIZE and
# WRITE_SIZE in passes of their own, kernel trace only), over tools/microbench.py.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_loss
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o run -- python3 $GRAFT_REPO_ROOT/tools/microbench.py --what loss --reps 5 > $OUT/$c.log 2>&1
done
python3 - <<'PY'
import collections, csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_loss"
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(f"{root}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "anonymous namespace" not in k:
                continue
            k = k.split("::")[1].split("(")[0].split("<")[0]
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg:
        out.setdefault(k, {})[name + "_KiB_per_launch"] = agg[k] / cnt[k]
for k, v in out.items():
    v["hbm_bytes_per_launch"] = (2.0 * v.get("FETCH_SIZE_KiB_per_launch", 0.0) + v.get("WRITE_SIZE_KiB_per_launch", 0.0)) * 1024.0
json.dump(out, open(root + "/traffic.json", "w"), indent=1)
for k, v in sorted(out.items()):
    print(k, round(v["hbm_bytes_per_launch"] / 1e6, 1), "MB")
PY
