#!/bin/bash
# whole-step A/B on ONE device, alternating repeats.
#   bash tools/ab_bench.sh VAR "v1 v2 ..." [reps] [extra bench args]
# VAR starting with "--" is a bench.py flag (e.g. --hip-lib with two builds of the library:
#   bash tools/ab_bench.sh --hip-lib "$PWD/ecg-representation-learning_amd/csrc/build/libecgvit_hip_prev.so $PWD/ecg-representation-learning_amd/libecgvit_hip.so");
# anything else is an environment variable read by the tools build (ECGVIT_NT_G, ECGVIT_NT_DIAG; add --hip-lib .../libecgvit_hip_tools.so).
if [ $# -lt 2 ]; then echo "usage: $0 VAR \"v1 v2 ...\" [reps] [extra bench args]" >&2; exit 2; fi
VAR=$1; VALS=$2; REPS=${3:-2}
shift $(( $# < 3 ? $# : 3 ))
for r in $(seq $REPS); do
  for v in $VALS; do
    echo -n "rep $r $VAR=$(basename $v): "
    if [[ $VAR == --* ]]; then
      python bench.py --no-cpu-baseline --no-masked --no-bf16-saved --no-small --no-fp8-large --steps 20 --warmup 5 $VAR $v "$@" 2>&1
    else
      env $VAR=$v python bench.py --no-cpu-baseline --no-masked --no-bf16-saved --no-small --no-fp8-large --steps 20 --warmup 5 "$@" 2>&1
    fi | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"avg_launch_us": [0-9.]*' | tr '\n' ' '
    echo
  done
done
