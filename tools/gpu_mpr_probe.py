"""Developer probe: mean MPR work feature (slowest lane's support evaluations + 8, summed over rounds) per substep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv
env = BatchedMyoEnv("myoHandPoseRandom-v0", num_envs=4096, as_torch=False)
env.reset(seed=0)
mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
env.batch.bench_rollout(60, 10, 0, mode, 100, None)
d = env.batch.read(capi.F_DIAG)
print("f_mpr per substep mean", d[:, 5].mean() / 10, " kernel ms", env.batch.last_kernel_ms() / 60)
