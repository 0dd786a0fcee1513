#!/bin/bash
# Every hash-tied piece of a round's evidence in ONE gpurun call on ONE device (tools/check_profiles.py holds them to each other):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/evidence_round.sh r05'
# Writes gpurun_out/evidence_<round>/: copy its files into profiles/ (same names) and add the prose headers by hand.
# Order matters: the supervised counter pass first, its summary copied into profiles/ ON THE BOX, then the driver-style bench line -- so
# that the line's roofline.traffic comes from the counter pass of the same sources on the same device.
: "${GRAFT_REPO_ROOT:?run through gpurun}"
RN=${1:-r05}
R=$GRAFT_REPO_ROOT; cd $R
SHA=$(python3 -c "import bench; print(bench.kernel_source_hash())") || exit 1
E=$R/gpurun_out/evidence_$RN; rm -rf $E; mkdir -p $E
pmc() {   # tag, bench args
  local tag=$1; shift
  bash tools/pmc_bench.sh $tag "$@" > $E/pmc_$tag.log 2>&1
  cp gpurun_out/pmc_${tag}_$SHA/summary.json $E/${RN}_pmc_$tag.json && cp gpurun_out/pmc_${tag}_$SHA/kernel_stats.csv $E/${RN}_pmc_${tag}_kernel_stats.csv
}
pmc base_sup
cp $E/${RN}_pmc_base_sup.json $E/${RN}_pmc_base_sup_kernel_stats.csv profiles/
python bench.py > $E/${RN}_bench_line.json 2> $E/bench.err
pmc base_masked --objective masked
pmc small --config small
pmc large_fp8 --config large --patch 10 --dtype fp8 --batch 256
L=$R/ecg-representation-learning_amd
python tools/gemm_ab.py --plain --nt4 --no-old 2>&1 | grep -v amdgpu.ids > $E/${RN}_gemm_shapes.txt
python tools/gemm_ab.py --m 64256 --dim 512 --plain --nt4 --no-old 2>&1 | grep -v amdgpu.ids > $E/${RN}_gemm_shapes_small.txt
python tools/gemm_ab.py --only ffn_ --aux8 --no-old --no-lib 2>&1 | grep -v amdgpu.ids > $E/${RN}_gemm_shapes_aux8.txt
python tools/gemm_ab.py --only ffn_ --aux-ld0 --no-old --no-lib 2>&1 | grep -v amdgpu.ids > $E/${RN}_gemm_shapes_auxld0.txt
# the attention kernels against round 4's library (same process, interleaved), at the base and the large / patch-10 geometry
python tools/attn_ab.py $L/csrc/build/libecgvit_hip_r04.so $L/libecgvit_hip.so 2>&1 | grep -v amdgpu.ids > $E/${RN}_attn_ab.txt
python tools/attn_ab.py $L/csrc/build/libecgvit_hip_r04.so $L/libecgvit_hip.so --n 501 --b 256 --h 16 2>&1 | grep -v amdgpu.ids >> $E/${RN}_attn_ab.txt
python tools/stress.py 150 2>&1 | grep -v amdgpu.ids > $E/${RN}_stress.txt
# whole-line A/B against the round-4 library (bf16 saved tensor: it does not know ECGVIT_EPI_AUX8), alternating on this device
bash tools/ab_r04.sh 3 2>&1 | grep -v amdgpu.ids > $E/${RN}_step_ab.txt
bash tools/steady_stats.sh base > /dev/null 2>&1; cp gpurun_out/steady_base.txt $E/${RN}_steady_base.txt
bash tools/steady_stats.sh small --config small > /dev/null 2>&1; cp gpurun_out/steady_small.txt $E/${RN}_steady_small.txt
bash tools/steady_stats.sh large_fp8 --config large --patch 10 --dtype fp8 --batch 256 > /dev/null 2>&1; cp gpurun_out/steady_large_fp8.txt $E/${RN}_steady_large_fp8.txt
rm -rf gpurun_out/steady_base gpurun_out/steady_small gpurun_out/steady_large_fp8
ls -la $E
python3 -c "
import json; o = json.load(open('$E/${RN}_bench_line.json'))
print(o['value'], o['ms_per_step'], o['roofline']['frac'], o['roofline']['traffic'], o.get('masked', {}).get('value'), o.get('small', {}).get('value'), o.get('fp8_large', {}).get('value'), o['kernel_source_sha16'])"
