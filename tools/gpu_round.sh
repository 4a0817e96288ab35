#!/bin/bash
# GPU round trip: parity (ours + the reference's own -gpu -check), bench line, rocprof summaries.
# usage: bash tools/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -26 | tee $OUT/${TAG}_pytest_gpu.txt
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/${TAG}_smoke.txt
if [ -x oracle/_ref/kangaroo_hip ]; then
  echo "== reference program on our engine: kangaroo -gpu -check (Check.cpp:467-621)"
  # dp is fixed at 8 and maxFound at 65536 inside Check (Check.cpp:418,492): the herd must stay below
  # 2^18 kangaroos or DPs are dropped and the harness itself derails
  timeout 900 ./oracle/_ref/kangaroo_hip -gpu -g 8,128 -check > $OUT/${TAG}_ref_gpu_check.txt 2>&1; echo "rc=$?" >> $OUT/${TAG}_ref_gpu_check.txt
  tail -12 $OUT/${TAG}_ref_gpu_check.txt
fi
echo "== bench"; python bench.py 2> $OUT/${TAG}_bench.err | tee $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
echo "== bench --gpus 2 on this one device is not possible; host path at the 8-GPU DP rate instead"
[ -x tools/dp_ingest_bench ] && (./tools/dp_ingest_bench --feeders 8 --launches 60 --launch-ms 21; ./tools/dp_ingest_bench --feeders 8 --launches 60) | tee $OUT/${TAG}_dp_ingest.txt
echo "== rocprofv3 kernel trace"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o kt -- python $OLDPWD/bench.py --no-cpu-baseline --no-pipeline --no-secondary --no-pmc > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err)
find $OUT/${TAG}_prof -name "*kernel_stats*" | head -3
for f in $(find $OUT/${TAG}_prof -name "*kernel_stats.csv"); do cp $f $OUT/${TAG}_kernel_stats.csv; done
cat $OUT/${TAG}_kernel_stats.csv 2>/dev/null | head -8
echo "== rocprofv3 PMC passes (counters only)"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-secondary --no-pmc --no-alu-ceiling > /dev/null 2> $OUT/${TAG}_pmc_$C.err)
  f=$(find $OUT/${TAG}_pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f $C | tee $OUT/${TAG}_pmc_$C.txt
done
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
# HBM bytes per launch of the bench's walk kernel from the two passes (FETCH doubled: gfx950 note in MI355X_MICROARCH.md),
# against the figure bench.py quotes from profiles/traffic.json: more than 2 % apart = the recorded figure is stale
python - $OUT/${TAG}_pmc_FETCH_SIZE.txt $OUT/${TAG}_pmc_WRITE_SIZE.txt $OUT/${TAG}_traffic.json $OUT/${TAG}_prof_bench.json <<'PY'
import json, re, sys
sys.path.insert(0, ".")
import bench
def mean(path, kernel):
    for line in open(path):
        if kernel in line:
            return float(re.search(r"mean=([0-9.e+]+)", line).group(1))
    raise SystemExit(f"{kernel} not in {path}")
line = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
k = line["roofline"]["kernel"]          # the kernel the bench actually ran (from the engine's options)
n, group = line["config"]["kangaroos_per_gpu"], line["config"]["group"]
bytes_per_launch = (2 * mean(sys.argv[1], k) + mean(sys.argv[2], k)) * 1024
new = {"kernel": k, "kangaroos": n, "group": group, "hbm_bytes_per_launch": int(round(bytes_per_launch, -7)),
       "bytes_per_jump": round(bytes_per_launch / (n * 64), 1),
       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled per the gfx950 note ({sys.argv[3].split('/')[-1]})",
       "source_blobs": bench.kernel_source_blobs()}   # bench.py quotes the figure only while these files are unchanged
json.dump(new, open(sys.argv[3], "w"), indent=1)
print(f"walk kernel {k}: {bytes_per_launch / 1e9:.2f} GB per launch = {bytes_per_launch / (n * 64):.1f} B/jump -> {sys.argv[3]} (copy to profiles/traffic.json)")
PY
echo "traffic rc=$?"
