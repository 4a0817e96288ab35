#!/bin/bash
# round 3, evidence pass: every A/B the round-3 kernel's defaults rest on, in ONE GPU session (boxes differ by a few %),
# the SQ counters of the default kernel, and the N > 1 bench protocol under torchrun on a real GPU.
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
S="timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256"
echo "== A/B: compiler-scheduled loop (asm 0) vs scheduled asm loop (asm 1); share 8, 2^23 kangaroos, group 64, dp 14"
{ for i in 1 2 3; do $S --asm 0,1 | grep "^asm"; done; echo "-- both distance words (jump distances ~2^56)"; $S --asm 0,1 --jd-bits 56 | grep "^asm"; } 2>&1 | tee $OUT/r03_ab_asm_loop.txt
echo "== A/B: every wave inverts (share 1, 256-thread blocks) vs one inversion per CU (share 8)"
{ for i in 1 2; do $S --asm 1 --shares 1,8 | grep "^asm"; done; } 2>&1 | tee $OUT/r03_ab_share.txt
echo "== A/B: DP records via device buffer + copy (ring 0) vs straight into pinned host memory (ring 1); DP 14 and DP 11"
{ for dp in 14 11; do for i in 1 2; do $S --asm 1 --dp $dp --dp-ring 0,1 | grep "^asm"; done; done; } 2>&1 | tee $OUT/r03_ab_dp_ring.txt
echo "== SQ counters of the default kernel"
bash tools/pmc_sq.sh r03 2>&1 | grep -E "mean|kernel" | tee $OUT/r03_pmc_sq.txt
echo "== N > 1 bench protocol under torchrun on one device: two ranks, rank 0 drives devices 0,0 (one process, one table)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --devices 0,0 --grid 256,128 --steps 10 --warmup 2 2> $OUT/r03_torchrun2_one_process.err | tail -1 | tee $OUT/r03_torchrun2_one_process.json; echo "rc=$?"; tail -3 $OUT/r03_torchrun2_one_process.err
echo "== same launcher, rank 0 sees 1 of 2 devices -> per-rank form (both ranks on the one device)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --grid 256,128 --steps 10 --warmup 2 2> $OUT/r03_torchrun2_per_rank.err | tail -1 | tee $OUT/r03_torchrun2_per_rank.json; echo "rc=$?"; tail -3 $OUT/r03_torchrun2_per_rank.err
