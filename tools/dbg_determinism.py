import json, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_grad_enabled(False)
from oracle.golden_inputs import SEED, SMALL_CFG, small_inputs
from oracle import unet_oracle as uo
from utils.utils import instantiate_from_config
man = json.load(open('tests/golden/unet_small_manifest.json'))
sd = uo.seeded_state_dict(man, SEED)
inp = small_inputs()
unet = instantiate_from_config({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG})
unet.load_state_dict({k: v for k, v in sd.items() if 'pluker' not in k and 'epipolar' not in k}, strict=True); unet = unet.cuda()
g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
def rel(a, b): return ((a - b).norm() / b.norm()).item()
y1 = unet(g['x'], g['t'], context=g['ctx_pf'], fs=g['fs'])
y2 = unet(g['x'], g['t'], context=g['ctx_pf'], fs=g['fs'])
print('rerun identical inputs:', rel(y1, y2))
x2 = torch.cat([g['x'], g['x']]); t2 = torch.cat([g['t'], g['t']]); f2 = torch.cat([g['fs'], g['fs']])
y4 = unet(x2, t2, context=[g['ctx_pf'], g['ctx_pf']], fs=f2)
print('pair halves vs single:', rel(y4[:2], y1), rel(y4[2:], y1), 'halves vs each other', rel(y4[:2], y4[2:]))
