#!/bin/bash
# Counter passes over bench.py (eager launches, timed region only), one rocprofv3 run per counter set with the kernel
# trace only: HBM traffic (FETCH_SIZE, WRITE_SIZE: a pass each) and the SQ instruction counters of every kernel.
# Writes gpurun_out/pmc_r02/{traffic.json,valu.json}; the summaries are committed under profiles/.
#   traffic: bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled for wide coalesced reads on gfx950
#   valu   : SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES per launch (wave-level instruction counts, summed over the chip)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r02
mkdir -p $OUT
ARGS="--steps 6 --warmup 2 --cpu-sample 0 --lean"
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/write.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o sq -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/sq.log 2>&1
python3 - <<'PY'
import collections, csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_r02"

def short(name):
    if "anonymous namespace" in name:
        return name.split("::")[1].split("(")[0]
    return name.split("(")[0].strip()

def rows_of(sub):
    rows = []
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"])))
    return rows

traffic = {}
for sub in ("fetch", "write"):
    rows = rows_of(sub)
    # the score filter is launched twice per call; its second launch (redo pass, normally a no-op) is listed apart
    top = max([v for k, _, v in rows if k == "filter_kernel"], default=0.0)
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for k, _, v in rows:
        if k == "filter_kernel" and v < 0.05 * top:
            k = "filter_kernel(redo pass)"
        agg[k] += v
        cnt[k] += 1
    for k in agg:
        traffic.setdefault(k, {})[sub + "_KiB_per_launch"] = agg[k] / cnt[k]
        traffic[k]["launches_" + sub] = cnt[k]
for k, v in traffic.items():
    f, w = v.get("fetch_KiB_per_launch", 0.0), v.get("write_KiB_per_launch", 0.0)
    v["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
json.dump(traffic, open(root + "/traffic.json", "w"), indent=1)

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for k, c, v in rows_of("sq"):
    agg[k][c] += v
    if c == "SQ_WAVES":
        cnt[k] += 1
valu = {}
for k, v in agg.items():
    n = max(cnt[k], 1)
    valu[k] = {"launches": n, "valu_insts_per_launch": v.get("SQ_INSTS_VALU", 0.0) / n,
               "salu_insts_per_launch": v.get("SQ_INSTS_SALU", 0.0) / n, "lds_insts_per_launch": v.get("SQ_INSTS_LDS", 0.0) / n,
               "waves_per_launch": v.get("SQ_WAVES", 0.0) / n, "busy_cycles_per_launch": v.get("SQ_BUSY_CYCLES", 0.0) / n}
json.dump(valu, open(root + "/valu.json", "w"), indent=1)
for k in sorted(valu):
    print(k, {a: round(b, 1) for a, b in valu[k].items()}, {a: round(b, 1) for a, b in traffic.get(k, {}).items()})
PY
