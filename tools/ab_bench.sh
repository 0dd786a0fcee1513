#!/bin/bash
# whole-step A/B on ONE device, alternating repeats: bash tools/ab_bench.sh VAR "v1 v2 ..." [reps] [extra bench args]
# e.g. two builds of the library:  bash tools/ab_bench.sh ECGVIT_HIP_LIB "$PWD/ecg-representation-learning_amd/libecgvit_hip_prev.so $PWD/ecg-representation-learning_amd/libecgvit_hip.so"
VAR=${1:-ECGVIT_NT_G}; VALS=${2:-"0 3"}; REPS=${3:-2}; shift 3
for r in $(seq $REPS); do
  for v in $VALS; do
    echo -n "rep $r $VAR=$(basename $v): "
    env $VAR=$v python bench.py --no-cpu-baseline --no-masked --steps 20 --warmup 5 "$@" 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"avg_launch_us": [0-9.]*' | tr '\n' ' '
    echo
  done
done
