#!/bin/bash
# build ablation variants of the FPS kernel into /tmp and time them (results are WRONG by design)
set -e
R=$GRAFT_REPO_ROOT
for a in 0 1 2; do
  mkdir -p /tmp/abl$a/nesie_amd
  for f in ball_query group_gather interpolate nesie_lib points_in_boxes sort_vertices; do cp $R/nesie_amd/csrc/$f.o /tmp/abl$a/ 2>/dev/null || true; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -DABL=$a -I$R/nesie_amd/csrc -c $R/tools/abl/fps_abl.hip -o /tmp/abl$a/fps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abl$a/libnesie_hip.so /tmp/abl$a/*.o
  echo "ABL=$a"
  NESIE_LIB=/tmp/abl$a/libnesie_hip.so NESIE_FPS_WAVES=16 B=8 python $R/tools/opbench.py 2>&1 | grep -E "fps 40000"
done
