import sys, faulthandler
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from magics_amd import World, scenarios as S, sharded
sc = S.grid_scenario(64, 10, interrobot=True, pitch=2.5, comm_radius=5.0)
cl = sharded.LocalCluster(sc, 2, World)
for sw in cl.ranks:
    print('rank', sw.plan.rank, 'local', len(sw.plan.local), 'ghosts', len(sw.plan.ghosts), 'send', sw.send_counts, 'recv', sw.recv_counts, flush=True)
for sw in cl.ranks:
    sw.sweep_segment(False, 1); sw.synchronize(); print('swept', flush=True)
import torch
for sw in cl.ranks:
    print(hex(sw.send_buf.data_ptr()), sw.send_buf.device, sw.send_buf.numel(), hex(torch.zeros(10,device='cuda').data_ptr()), flush=True)

