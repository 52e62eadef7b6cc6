"""Developer probe (diagnostic build), B=1024 = one wave per SIMD: consecutive env steps, to fit next-step duration on this step's features."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MYO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myosuite_mjx_amd", "libmyo_hip_stamps.so")
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv
B = 1024
out = {}
for env_id in ("myoHandPoseRandom-v0", "myoLegWalk-v0"):
    env = BatchedMyoEnv(env_id, num_envs=B, as_torch=False)
    mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
    env.batch.set_balance(0)
    env.reset(seed=1)
    S, D, EL = [], [], []
    env.batch.bench_rollout(25, 10, 0, mode, env.max_episode_steps, None)
    for rep in range(24):
        env.batch.bench_rollout(1, 10, 0, mode, env.max_episode_steps, None)
        st, ok = capi.read_stamps(env.batch, B)
        S.append(st.copy()); D.append(env.batch.read(capi.F_DIAG).copy()); EL.append(env.batch.read(capi.F_ELAPSED).copy())
    out[env_id + "/stamps"] = np.stack(S); out[env_id + "/diag"] = np.stack(D); out[env_id + "/elapsed"] = np.stack(EL)
np.savez_compressed(os.path.join("gpurun_out", "cost_fit2.npz"), **out)
print("ok")
