cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
echo "== bench --gpus 8 on one device under taskset -c 0-7"
taskset -c 0-7 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 10 --warmup 2 > $OUT/r05_bench_8engines_taskset8.json 2> $OUT/r05_bench_8engines_taskset8.err; tail -c 1500 $OUT/r05_bench_8engines_taskset8.json | head -c 1500; echo
cd /tmp
printf "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\nB60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4\n" > in80.txt
echo "== kangaroo_mi355x -gpuId 0,0: two GPU threads of the reference program, one table, the program's own DP (13 for 2^24 kangaroos)"
KNG_STATS=10 timeout 62 stdbuf -o0 -e0 $GRAFT_REPO_ROOT/oracle/_ref/kangaroo_mi355x -t 0 -gpu -gpuId 0,0 in80.txt 2>&1 | tr "\r" "\n" > $OUT/r05_ref_program_two_engines.txt
grep -v "^\[" $OUT/r05_ref_program_two_engines.txt | grep -v "^$" | tail -14; grep "^\[" $OUT/r05_ref_program_two_engines.txt | tail -1
echo "== -d 10 with KNG_TABLE_THREADS=8"
KNG_TABLE_THREADS=8 KNG_STATS=10 timeout 56 stdbuf -o0 -e0 $GRAFT_REPO_ROOT/oracle/_ref/kangaroo_mi355x -t 0 -gpu -d 10 in80.txt 2>&1 | tr "\r" "\n" > $OUT/r05_ref_program_dp10_8threads.txt
grep "SolveKeyGPU_kng" $OUT/r05_ref_program_dp10_8threads.txt | tail -2; grep "^\[" $OUT/r05_ref_program_dp10_8threads.txt | tail -1
