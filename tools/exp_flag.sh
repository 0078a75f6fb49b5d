#!/bin/bash
# A/B of a compile-time experiment flag on the GPU box: bench with the library as built, then rebuilt with $1.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp; mkdir -p $O
BASE="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
run() {
    timeout -k 10 120 python $R/bench.py --steps 200 --cpu-sample 0 > $O/$1.json 2> $O/$1.err
    python3 -c "
import json
d=json.load(open('$O/$1.json'))
print('$1', round(d['value']), round(d['ms_per_step'],5), [(e['kernel'], round(e['us_alone'],1), round(e['us_in_step'],1)) for e in d['roofline_all'] if e['kernel'].startswith('match') or e['kernel']=='filter'])
"
}
run base1
cd $R/object_detection_amd/csrc && touch od_match.hip od_select.hip od_nms.hip && make FLAGS="$BASE $1" > $O/build.log 2>&1
run flag1
run flag2
cd $R/object_detection_amd/csrc && touch od_match.hip od_select.hip od_nms.hip && make FLAGS="$BASE" > $O/build2.log 2>&1
run base2
