#!/bin/bash
# Experiment: threads per workgroup of the restore kernel x groups per image, inside the overlapped step (on the GPU box).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp; mkdir -p $O
BASE="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
for t in $THREADS; do
    cd $R/object_detection_amd/csrc && touch od_match.hip && make FLAGS="$BASE -DOD_RL_T=$t" > $O/build_$t.log 2>&1
    for g in $GROUPS_LIST; do
        timeout -k 10 120 python $R/bench.py --steps 200 --cpu-sample 0 --match-restore-groups $g > $O/rlt.json 2> $O/rlt.err
        python3 -c "
import json
d=json.load(open('$O/rlt.json'))
print('threads $t groups $g', round(d['value']), round(d['ms_per_step'],5), [(e['kernel'], round(e['us_alone'],1), round(e['us_in_step'],1)) for e in d['roofline_all'] if e['kernel']=='match_restore'])
"
    done
done
