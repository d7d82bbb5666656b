import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R+"/sylph-few-shot-detection_amd"); sys.path.insert(0, R+"/tests")
import torch, torch.nn.functional as F
from test_hip_parity import _engine, _bf16_round
g = torch.Generator().manual_seed(3)
B,H,W = 1, 64, 96
x = _bf16_round(torch.randn(B,3,H,W,generator=g))
w = _bf16_round(torch.randn(64,3,7,7,generator=g)/147**0.5)
scale, shift = torch.rand(64,generator=g)+0.5, torch.randn(64,generator=g)*0.1
eng = _engine("bf16")
so, po = eng.stem_maxpool(x,w,scale,shift)
ref = F.max_pool2d(so.cpu(),3,2,1)
d = (po.cpu()-ref).abs()
print("shape", po.shape, "max diff", d.max().item(), "frac bad", (d>0).float().mean().item())
bad = (d>0).any(dim=1)[0]
torch.set_printoptions(linewidth=250)
print(bad.int())
badc = (d>0).any(dim=2).any(dim=2)[0]
print("bad channels", badc.int().tolist())
