import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.env import VoltageControlBatch
from mapdn_amd.netspec import make_case
case = sys.argv[1] if len(sys.argv) > 1 else "case141"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
net, prof = make_case(case)
scale = {"case33": 0.8, "case141": 0.6, "case322": 0.8}[case]
env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=scale, action_bias=0.0), n_envs=B, device="cuda:0")
rng = np.random.default_rng(0)
rows = rng.integers(0, prof.n_rows, B); pv = prof.pv[rows]
qs = rng.uniform(-scale, scale, (B, net.n_sgen)) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
ins = [torch.as_tensor(x, device="cuda:0") for x in (prof.load_p[rows], prof.load_q[rows], pv, qs)]
acc = []
for i in range(12):
    env.solve(*ins)
    acc.append(env.episode_returns()[8:14].cpu().numpy())
v = np.median(np.array(acc[2:]), axis=0)
print(case, "flat %d  light %d  full %d  bwd %d  nLight*100+nFull*10+nRedo %d  iters %d" % tuple(v))
