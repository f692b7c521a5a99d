import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.env import VoltageControlBatch
from mapdn_amd.netspec import make_case
case = sys.argv[1] if len(sys.argv) > 1 else "case141"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
net, prof = make_case(case)
scale = {"case33": 0.8, "case141": 0.6, "case322": 0.8}[case]
env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=scale, action_bias=0.0), n_envs=B, device="cuda:0")
env.reset()
act = torch.empty(B, net.n_sgen, device="cuda:0").uniform_(-scale, scale)
w0 = None
for i in range(6):
    r, t, info = env.step(act)
    v = info[0].cpu().numpy()
    print(case, "prologue %d  fwd %d  bwd_first3 %d  bwd_rest %d  loop %d  epilogue %d  iters %d" % tuple(v[:7]), " wall100MHz delta", (v[7] - w0) if w0 else 0)
    w0 = v[7]
