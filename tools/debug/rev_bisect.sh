# which operand sizes' reversed tile order breaks the B = 8 parity test (debug aid)
for e in "NESIE_PW_REV_MB=300" "NESIE_PW_REV_MB=250" "NESIE_PW_REV_MB=150" "NESIE_PW_REV_MB=136" "NESIE_PW_REV_MB=100"; do
  echo "== $e"
  env $e NESIE_PW_REV_WGRAD=0 NESIE_PW_REV_FWD=1 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -s -m gpu -k "metric_batch" 2>&1 | grep -E "flat_rel_l2_hip_vs_cpu.: [0-9.e-]*|worst_parameter.: ..name.: .[a-zA-Z0-9_.]*" -o | head -2
done
