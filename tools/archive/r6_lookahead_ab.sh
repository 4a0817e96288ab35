for la in 24,16,8 12,8,4 36,24,12 48,32,16 16,8,0 24,16,8; do
  echo "## KNG_HT_LOOKAHEAD=$la"; KNG_HT_LOOKAHEAD=$la oracle/_ref/htbench_kng ingestp 240000000 80000000 16 | tail -3
done
