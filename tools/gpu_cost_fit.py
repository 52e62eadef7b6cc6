"""Developer probe (diagnostic build): one wave per SIMD (B=1024) so a wave's duration is its own work; dumps per-env stage cycles
and the work features to gpurun_out/cost_fit.npz for fitting the placement cost model offline."""
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
    S, D = [], []
    env.batch.bench_rollout(25, 10, 0, mode, env.max_episode_steps, None)
    for rep in range(6):
        env.batch.bench_rollout(7, 10, 0, mode, env.max_episode_steps, None)
        st, ok = capi.read_stamps(env.batch, B)
        S.append(st.copy()); D.append(env.batch.read(capi.F_DIAG).copy())
    out[env_id + "/stamps"] = np.stack(S); out[env_id + "/diag"] = np.stack(D)
    print(env_id, "mean wave cycles", np.stack(S)[:, :, :10].sum(2).mean())
np.savez_compressed(os.path.join("gpurun_out", "cost_fit.npz"), **out)
