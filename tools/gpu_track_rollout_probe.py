"""Developer probe: contact / candidate statistics of the TrackEnv bench rollout (default RANDOM reference, U(-1,1) actions)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.track import TrackEnv
B = 4096
env = TrackEnv(num_envs=B, seed=0, autoreset=True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
for k in range(61):
    a = torch.rand((B, env.act_dim), device="cuda", generator=g) * 2 - 1
    torch.cuda.synchronize(); t0 = time.time()
    obs, r, d, info = env.step(a)
    torch.cuda.synchronize(); dt = time.time() - t0
    if k % 10 == 0:
        D = env.batch.read(capi.F_DIAG)
        q = env.view(capi.F_QPOS)
        print(f"step {k}: {dt*1e3:.1f} ms | ncon mean {D[:,1].mean():.1f} max {D[:,1].max()} nefc mean {D[:,0].mean():.0f} | ncand/substep {(D[:,4]&0xFFFF).mean()/5:.0f} | mpr {D[:,5].mean()/5:.0f} | newton it {D[:,2].mean():.1f} | done {int(d.sum())} | arm z {float(q[:,2].mean()):.3f} obj z {float(q[:,31].mean()):.3f}")
