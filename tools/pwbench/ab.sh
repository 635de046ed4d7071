# same-device A/B of two builds of the layer kernel: alternate, 3 rounds, plain (1) and fused (2)
for r in 1 2 3; do for v in burst inter; do for s in 0 1 2; do
  echo "$v shape $s: $(./tools/pwbench/pwbench_$v 40 $s 1 | tail -1 | awk '{print $(NF-1)}') plain  $(./tools/pwbench/pwbench_$v 40 $s 2 | tail -1 | awk '{print $(NF-1)}') fused"
done; done; done
