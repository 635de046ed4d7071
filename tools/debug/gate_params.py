"""Debug aid: bench.parity_gate at B scenes with the six worst parameters and the loss terms."""
import sys
sys.path.insert(0, '.')
import torch
import bench
g = bench.parity_gate(torch.device('cuda:0'), 'pretrain', scenes=int(sys.argv[1]) if len(sys.argv) > 1 else 8, backward=True)
print('max loss rel diff', g['max_rel_diff'], g['worst_term'], 'picks agreed', g['own_vote_picks_agreed'], 'taps', g['grid_taps'])
print('flat', g['gradient']['flat_rel_l2_hip_vs_cpu'])
for r in g['gradient']['worst_parameters']:
    print('   %.4f  %s' % (r['max_err_over_max_grad'], r['name']))
