#!/bin/bash
# Counter passes over bench.py (eager launches, timed region only), one rocprofv3 run per counter set with the kernel
# trace only: HBM traffic (FETCH_SIZE, WRITE_SIZE: a pass each) and the SQ instruction counters of every kernel; plus the
# kernel statistics of the same command.  Writes gpurun_out/pmc_r03/pmc.json (committed as profiles/r03_pmc.json) and
# gpurun_out/pmc_r03/stats/ (its kernel_stats.csv is committed as profiles/r03_bench_kernel_stats.csv).
#   traffic: bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled for wide coalesced reads on gfx950
#   valu   : SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES per launch (wave-level instruction counts, summed over the chip)
# The file records the git blob hashes of the kernel sources, which bench.py compares with the sources it runs from.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r03
mkdir -p $OUT
ARGS="--steps 6 --warmup 2 --cpu-sample 0 --lean"
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
echo fetch pass done
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/write.log 2>&1
echo write pass done
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o sq -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/sq.log 2>&1
echo sq pass done
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-sample 0 --lean > $OUT/stats.log 2>&1
echo stats pass done
python3 - <<'PY'
import collections, csv, glob, hashlib, json, os
repo = os.environ["GRAFT_REPO_ROOT"]
root = repo + "/gpurun_out/pmc_r03"

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

kernels = {}
for sub in ("fetch", "write"):
    rows = rows_of(sub)
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for k, _, v in rows:
        agg[k] += v
        cnt[k] += 1
    for k in agg:
        kernels.setdefault(k, {})[sub + "_KiB_per_launch"] = agg[k] / cnt[k]
        kernels[k]["launches_" + sub] = cnt[k]
for k, v in kernels.items():
    f, w = v.get("fetch_KiB_per_launch", 0.0), v.get("write_KiB_per_launch", 0.0)
    v["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for k, c, v in rows_of("sq"):
    agg[k][c] += v
    if c == "SQ_WAVES":
        cnt[k] += 1
for k, v in agg.items():
    n = max(cnt[k], 1)
    kernels.setdefault(k, {}).update({
        "launches_sq": n, "valu_insts_per_launch": v.get("SQ_INSTS_VALU", 0.0) / n,
        "salu_insts_per_launch": v.get("SQ_INSTS_SALU", 0.0) / n, "lds_insts_per_launch": v.get("SQ_INSTS_LDS", 0.0) / n,
        "waves_per_launch": v.get("SQ_WAVES", 0.0) / n, "busy_cycles_per_launch": v.get("SQ_BUSY_CYCLES", 0.0) / n})
hashes = {}
for name in ("od_common.h", "od_match.hip", "od_select.hip", "od_nms.hip"):
    data = open(f"{repo}/object_detection_amd/csrc/{name}", "rb").read()
    hashes[name] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
json.dump({"source_hashes": hashes, "command": "bench.py --steps 6 --warmup 2 --cpu-sample 0 --lean (one rocprofv3 --pmc pass per counter set)",
           "kernels": kernels}, open(root + "/pmc.json", "w"), indent=1)
for k in sorted(kernels):
    print(k, {a: round(b, 1) for a, b in kernels[k].items()})
PY
