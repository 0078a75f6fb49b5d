#!/bin/bash
# HBM traffic of the bench kernels from the rocprofv3 PMC counters: FETCH_SIZE and WRITE_SIZE need a pass
# each (TCC slots), each pass in a run of its own with the kernel trace only.  Writes
# gpurun_out/pmc_bench/traffic.json: kernel -> bytes per launch, with the gfx950 correction for wide
# coalesced reads (FETCH_SIZE counts 64 B per 128-B request: doubled) and both counters in KiB.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/write.log 2>&1
python3 - <<'PY'
import collections, csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_bench"
out = {}
for name in ("fetch", "write"):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    rows = []
    for f in glob.glob(f"{root}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").strip()
            if "anonymous namespace" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("::")[1].split("(")[0]
            rows.append((k, float(r["Counter_Value"])))
    # the score filter is launched twice per call; its second launch (redo pass, normally a no-op) is listed apart
    top = max([v for k, v in rows if k == "filter_kernel"], default=0.0)
    for k, v in rows:
        if k == "filter_kernel" and v < 0.05 * top:
            k = "filter_kernel(redo pass)"
        agg[k] += v; cnt[k] += 1
    for k in agg:
        out.setdefault(k, {})[name + "_KiB_per_launch"] = agg[k] / cnt[k]
        out[k]["launches_" + name] = cnt[k]
for k, v in out.items():
    f, w = v.get("fetch_KiB_per_launch", 0.0), v.get("write_KiB_per_launch", 0.0)
    v["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0      # FETCH_SIZE doubled: gfx950 wide-read correction
json.dump(out, open(root + "/traffic.json", "w"), indent=1)
for k, v in sorted(out.items()):
    print(k, {a: round(b, 1) for a, b in v.items()})
PY
