import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv
env = BatchedMyoEnv("myoHandPoseRandom-v0", num_envs=4096, as_torch=False)
env.reset(seed=0)
mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
tot = np.zeros(4, int)
for k in range(10):
    env.batch.bench_rollout(100, 10, 0, mode, 100, None)
    f = env.status()
    for b in range(4): tot[b] += ((f >> b) & 1).sum()
    d = env.batch.read(capi.F_DIAG)
print("flag bits (bad_state, bad_qacc, contact_overflow, cand_overflow) over 1000 steps x 4096 envs:", tot, "ncon max now", d[:,1].max(), "nefc max", d[:,0].max())
