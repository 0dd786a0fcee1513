#!/bin/bash
# masked pre-train step A/B on ONE device: csrc/build/libecgvit_hip_prev.so (`make prev` of the baseline sources) with device-resident mask indices (the
# path that validates them with blocking reads) against the shipped library with host-resident ones; usage (GPU box): bash tools/ab_masked.sh
P='import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],2), d["final_loss"])'
for i in 1 2 3; do
  python bench.py --objective masked --steps 12 --warmup 3 --no-cpu-baseline --mask-on-device --hip-lib ecg-representation-learning_amd/csrc/build/libecgvit_hip_prev.so 2>/dev/null | python -c "$P" old
  python bench.py --objective masked --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P" new
done
