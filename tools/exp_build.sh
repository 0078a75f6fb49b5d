#!/bin/bash
# Build experimental variants of libod_boxes.so: tools/exp_build.sh NAME "-DFLAG ..." [file.hip]
set -e
cd "$(dirname "$0")/../object_detection_amd/csrc"
name=$1; flags=$2; src=${3:-od_match.hip}
mkdir -p ../../build_exp
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -c -o ../../build_exp/$name.o $src
objs=$(ls *.o | grep -v "${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_exp/lib_$name.so $objs ../../build_exp/$name.o
rm ../../build_exp/$name.o
