OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "every_walk_kernel or bench_config_total or low_word or exact_path or dp_ring or reference_vectors or walk_vs_oracle" 2>&1 | tail -3
echo "== A/B at the bench config (80-bit: carries 2^-23 per jump): carry leaves the loop (carryexit) vs L2 atomic in the loop (carryatomic)"
{ for i in 1 2 3; do for v in carryexit carryatomic; do echo -n "$v: "; KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 | grep "^asm" | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"; done; done
echo "-- jump distances ~2^55 (configs[3]'s table): both words stream (dsplit 0) vs low word + atomics (dsplit 1)"
for i in 1 2 3; do for ds in 0 1; do echo -n "dsplit $ds: "; timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 --jd-bits 55 --dsplit $ds | grep "^asm" | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"; done; done
echo "-- jump distances ~2^58 (115-bit ranges: the largest the engine picks the layout for)"
for ds in 0 1; do echo -n "dsplit $ds: "; timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 --jd-bits 58 --dsplit $ds | grep "^asm" | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"; done
echo "-- jump distances ~2^62 (125-bit): forced, for the record"
for ds in 0 1; do echo -n "dsplit $ds: "; timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 --jd-bits 62 --dsplit $ds | grep "^asm" | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"; done; } 2>&1 | tee $OUT/r03_ab_carry_atomic.txt
