#!/bin/bash
# usage: tools/build_exp.sh 'run<13,1,3,8,1>(864,11,512,D);' [out_name] [extra -D flags]
set -e
cd "$(dirname "$0")/.."
OUT=${2:-exp_pbs}
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value $3 -DEXP_CASES="$1" -o tools/$OUT tools/exp_pbs.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/exp_build.log || { grep error /tmp/exp_build.log | head; exit 1; }
python "$(dirname "$0")/resource_usage.py" /tmp/exp_build.log | grep pbs_kernel
