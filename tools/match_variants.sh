#!/bin/bash
# Kernel times of the matching path alone (tools/match_probe.py under rocprofv3 --stats) for build variants of the
# one-launch kernel: waves per SIMD 7 (default) and 6, and the four-launch chain.  Run on the GPU box.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/variants; mkdir -p $O
BASE="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
stats() {  # name, extra env
    cd /tmp
    env $2 true
    ( export $2; timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -o s -- python3 $R/tools/match_probe.py 300 1 > $O/$1.log 2>&1 )
    python3 - "$O/$1" "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "match" in r["Name"]:
            print(sys.argv[2], r["Name"].split("::")[1].split("(")[0], r["Calls"], round(float(r["AverageNs"]) / 1000, 2), "us")
PY
}
stats w7 OD_X=0
stats legacy OD_MATCH_LEGACY=1
for w in $VARIANTS; do
    cd $R/object_detection_amd/csrc && touch od_match.hip && make FLAGS="$BASE -DOD_FUSED_WAVES=$w" > $O/build_$w.log 2>&1
    stats w$w OD_X=0
done
