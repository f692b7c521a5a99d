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
acc = []
for i in range(12):
    env.step(act)
    acc.append(env.episode_returns()[8:14].cpu().numpy())
v = np.median(np.array(acc[2:]), axis=0)
print(case, "prologue %d  fwd %d  bwd %d  loop %d  epilogue %d  iters(env) %d  total %d" % (*v, v[0] + v[3] + v[4]))
