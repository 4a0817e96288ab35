OUT=$PWD/gpurun_out; mkdir -p $OUT
S="timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256"
{ echo "# every wave inverts (share 1, 256-thread blocks) against one inversion per CU (share 8), scheduled loop, both at 2 waves per SIMD"; echo "# (amdgpu_waves_per_eu(2,2): the share-1 kernel used to take 260 VGPRs = ONE wave per SIMD, which made the first version of this file look like +16 %)"; for i in 1 2 3; do $S --asm 1 --shares 1,8 | grep "^asm"; done; echo "# compiler-scheduled loop"; $S --asm 0 --shares 1,8 | grep "^asm"; } 2>&1 | tee $OUT/r03_ab_share.txt
