#!/bin/bash
# Marginal cost of each phase of the persistent attention backward: seven tools builds of attention.hip, each with ONE phase removed
# (-DECGVIT_ATTN_ABL=mask, bit n = phase n of attention.hip's list; results are WRONG by construction), timed against the shipped library by tools/attn_ab.py.
# usage:  bash tools/attn_ablate.sh build      (here: cross-compiles csrc/build/libecgvit_abl_<n>.so)
#         bash tools/attn_ablate.sh run        (on the GPU box)
set -e
MASKS="${MASKS:-2 4 8 16 32 64 128 62 126 254 256 512 1024 2048 4096 1792 3840 7936}"   # bits 1-7: backward phases, 8-12: forward phases (attention.hip)
C=ecg-representation-learning_amd/csrc
if [ "$1" = build ]; then
  make -C $C -j8 all tools > /dev/null
  for n in $MASKS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -Iinclude -DECGVIT_TOOLS -DECGVIT_ATTN_ABL=$n -c $C/attention.hip -o $C/build/attention.abl$n.o &
    k=$((k+1)); if [ $((k % 4)) = 0 ]; then wait; fi
  done
  wait
  for n in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build/libecgvit_abl_$n.so $(ls $C/build/*.o | grep -v "tools.o\|abl\|build/attention.o") $C/build/attention.abl$n.o
  done
  ls -la $C/build/libecgvit_abl_*.so
else
  declare -A names=([2]="no dQ product (C)" [4]="no dV / dK products (B)" [8]="no vector arithmetic (V)" [16]="no S / dP products (A)" [32]="no dS -> LDS (W)" [64]="no slab stream" [128]="no dQ stores" [62]="no C, B, V, A, W (the skeleton: streams, waits, barriers, flush)" [126]="skeleton without the slab stream" [254]="skeleton without slab stream and dQ stores" [256]="fwd: no softmax / dropout arithmetic" [512]="fwd: no Q.K^T products" [1024]="fwd: no P.V products" [2048]="fwd: no K / V image loads" [4096]="fwd: no output stores" [1792]="fwd: no arithmetic, no products" [3840]="fwd: ... and no K / V loads" [7936]="fwd: ... and no stores")
  for n in $MASKS; do
    echo "---- ablation mask $n: ${names[$n]}"
    python tools/attn_ab.py $C/build/libecgvit_abl_$n.so ecg-representation-learning_amd/libecgvit_hip.so "${@:2}" 2>&1 | grep "fwd:\|bwd:\|Error\|error" | grep $( [ $n -ge 256 ] && echo fwd || echo bwd ) || true
  done
fi
