# rocprofv3 over the reference CLI (kangaroo_mi355x) WHILE IT SAVES: default herd, -d 18, -ws -wi 8, ~30 s.  Shows the walk kernel
# busy for (nearly) the whole loop with the saves inside, and what a save costs the device: one kng_snapshot_pack_kernel per save.
cd /tmp && export TMPDIR=/tmp
printf "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\nB60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4\n" > in80.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -f prof.work
KNG_STATS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_prog_prof -o kt -- $GRAFT_REPO_ROOT/oracle/_ref/kangaroo_mi355x -t 0 -gpu -d 18 -ws -w prof.work -wi 8 -m 0.3 in80.txt > $OUT/r06_prog_prof.txt 2>&1
f=$(find $OUT/r06_prog_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/r06_ref_program_saves_kernel_stats.csv; cat $OUT/r06_ref_program_saves_kernel_stats.csv | head -8
tr "\r" "\n" < $OUT/r06_prog_prof.txt | grep -E "SolveKeyGPU_kng|SaveWork_kng" | tail -6
rm -rf $OUT/r06_prog_prof
