"""Developer probe: step-kernel time under feature switches + solver statistics after a random-action rollout."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv

B = int(os.environ.get("B", 4096))
env = BatchedMyoEnv("myoHandPoseRandom-v0", num_envs=B, as_torch=False)
for name, sw in (("smooth", (1, 1, 1)), ("limits", (1, 0, 1)), ("capsule", (0, 0, 1)), ("all", (0, 0, 0))):
    env.model.set_switch(*sw)
    env.reset(seed=1)
    mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
    env.batch.bench_rollout(30, 10, 0, mode, 100, None)      # settle into the rollout regime
    ms = env.batch.bench_rollout(20, 10, 0, 0, 0, None) / 20
    d = env.batch.read(capi.F_DIAG)
    fl = env.batch.status()
    print(f"{name:8s} step kernel {ms:8.3f} ms/env-step  ({B/ms*1e3:,.0f} env-steps/s) | nefc mean {d[:,0].mean():.1f} max {d[:,0].max()} "
          f"ncon mean {d[:,1].mean():.1f} max {d[:,1].max()} iter mean {d[:,2].mean():.2f} max {d[:,2].max()} | flags: "
          f"bad_state {(fl&1).astype(bool).sum()} bad_qacc {(fl&2).astype(bool).sum()} con_ovf {(fl&4).astype(bool).sum()} cand_ovf {(fl&8).astype(bool).sum()}")
for nsub in (1, 2, 5, 10):
    ms = env.batch.bench_rollout(10, nsub, 0, 0, 0, None) / 10
    print(f"all, nsub={nsub}: {ms:.3f} ms")
