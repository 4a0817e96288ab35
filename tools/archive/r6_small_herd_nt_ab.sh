for i in 1 2 3; do
  for v in default plainxy; do
    if [ $v = default ]; then unset KNG_LIB_PATH; else export KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so; fi
    echo "== $v"; timeout 600 python tools/small_herd_ab.py 1 | grep "share 4"
  done
done
