cd /tmp && export TMPDIR=/tmp
printf "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\nB60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4\n" > in80.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out
KNG_STATS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_prog_prof -o kt -- $GRAFT_REPO_ROOT/oracle/_ref/kangaroo_mi355x -t 0 -gpu -m 0.3 in80.txt > $OUT/r05_prog_prof.txt 2>&1
f=$(find $OUT/r05_prog_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/r05_ref_program_kernel_stats.csv; cat $OUT/r05_ref_program_kernel_stats.csv | head -6
tr "\r" "\n" < $OUT/r05_prog_prof.txt | grep "SolveKeyGPU_kng" | tail -1
rm -rf $OUT/r05_prog_prof
