"""Tune the step's GEMMs with PyTorch TunableOp (rocBLAS + hipBLASLt solution search) and
write the per-shape picks to a CSV that bench.py can load (NESIE_TUNABLEOP_FILE).
usage (GPU box): python tools/tune_gemm.py out.csv [workload] [batch]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/tunableop_results.csv'
workload = sys.argv[2] if len(sys.argv) > 2 else 'pretrain'
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
import torch
import torch.cuda.tunable as tunable
import bench
from nesie_amd.votenet import nesie_votenet_scannet_cfg

tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(out)
tunable.set_max_tuning_duration(8)      # ms per candidate
tunable.set_max_tuning_iterations(20)
t0 = time.time()
done = False


def ticker():
    while not done:
        print(f'[tune] {time.time() - t0:6.0f}s', flush=True)
        time.sleep(30)


threading.Thread(target=ticker, daemon=True).start()
dev = torch.device('cuda:0')
cfg = nesie_votenet_scannet_cfg()['optimizer']
model, step, bucket = bench.build_step(dev, batch, 1000, cfg['lr'], cfg['weight_decay'],
                                       graph=False, workload=workload)
for i in range(2):
    step()
    torch.cuda.synchronize()
    print(f'[tune] step {i} done at {time.time() - t0:.0f}s', flush=True)
tunable.write_file(out)
done = True
print('[tune] wrote', out, 'entries:', len(tunable.get_results()))
