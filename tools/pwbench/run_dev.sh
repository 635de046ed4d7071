for s in 0 1 2 3 7 8 10; do ./tools/pwbench/pwbench_dev 20 $s 0 2>&1 | tail -1; done
