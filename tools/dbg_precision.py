import json, os, sys, numpy as np, torch
sys.path.insert(0, '.')
torch.set_grad_enabled(False)
from oracle.golden_inputs import SEED, SMALL_CFG, small_inputs
from oracle import unet_oracle as uo
from utils.utils import instantiate_from_config
fx = dict(np.load('tests/golden/unet_small.npz'))
man = json.load(open('tests/golden/unet_small_manifest.json'))
sd = uo.seeded_state_dict(man, SEED)
inp = small_inputs()
ref = torch.from_numpy(fx['y_nocam_pf'])
with uo.operand_rounding(torch.bfloat16):
    emu = uo.unet_forward(sd, SMALL_CFG, inp['x'], inp['t'], inp['ctx_pf'], inp['fs'], None)
def rel(a,b): return ((a-b).norm()/b.norm()).item(), ((a-b).abs().max()/b.abs().max()).item()
print('emu vs fp32', rel(emu, ref))
if torch.cuda.is_available():
    unet = instantiate_from_config({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG})
    unet.enable_camera_conditioning(dict(origin_h=64, origin_w=64, is_3d_full_attn=False, num_register_tokens=4, attention_resolution=[8,4,2,1], compression_factor=1))
    unet.load_state_dict(sd, strict=True); unet = unet.cuda()
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k,v in inp.items()}
    y = unet(g['x'], g['t'], context=g['ctx_pf'], fs=g['fs']).cpu()
    print('hip vs fp32', rel(y, ref)); print('hip vs emu', rel(y, emu))
