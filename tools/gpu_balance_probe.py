"""Developer probe (diagnostic build): per-wave durations of one step launch, grouped by SIMD; how well does the placement
cost estimate (diag[3]) predict them; what would perfect balancing buy?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MYO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myosuite_mjx_amd", "libmyo_hip_stamps.so")
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv
B = 4096
env = BatchedMyoEnv(os.environ.get("ENV", "myoHandPoseRandom-v0"), num_envs=B, as_torch=False)
mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
for bal in (0, 1):
    env.batch.set_balance(bal)
    env.reset(seed=1)
    env.batch.bench_rollout(40, 10, 0, mode, env.max_episode_steps, None)
    prev_cost = env.batch.read(capi.F_DIAG)[:, 3].astype(np.float64)
    ms = env.batch.bench_rollout(1, 10, 0, mode, env.max_episode_steps, None)
    kms = env.batch.last_kernel_ms()
    st, ok = capi.read_stamps(env.batch, B)
    dur = st[:, :10].sum(1).astype(np.float64)
    hw = st[:, 10].astype(np.int64); xcc = st[:, 11].astype(np.int64) & 0xF
    key = xcc * 100000 + ((hw >> 13) & 7) * 10000 + ((hw >> 12) & 1) * 1000 + ((hw >> 8) & 0xF) * 10 + ((hw >> 4) & 3)
    new_cost = env.batch.read(capi.F_DIAG)[:, 3].astype(np.float64)
    u, inv = np.unique(key, return_inverse=True)
    per_simd_max = np.zeros(len(u)); np.maximum.at(per_simd_max, inv, dur)
    per_simd_sum = np.zeros(len(u)); np.add.at(per_simd_sum, inv, dur)
    print(f"== balance {bal}: kernel {kms:.3f} ms; wave duration cycles mean {dur.mean():,.0f} median {np.median(dur):,.0f} p90 {np.quantile(dur,.9):,.0f} max {dur.max():,.0f}  (max/mean {dur.max()/dur.mean():.2f})")
    print(f"   per-SIMD: waves {np.bincount(inv).min()}..{np.bincount(inv).max()}, last-finisher mean {per_simd_max.mean():,.0f} max {per_simd_max.max():,.0f}; sum-of-durations mean {per_simd_sum.mean():,.0f} max {per_simd_sum.max():,.0f} (max/mean {per_simd_sum.max()/per_simd_sum.mean():.3f})")
    # NOTE: the stamps are written per WORKGROUP index; order[] maps workgroup -> env, so costs must be compared per env
    prio = (st[:, 11].astype(np.int64) >> 8) & 0xFF
    cost = st[:, 11].astype(np.int64) >> 16
    print(f"   corr(wave duration, its cost estimate of THIS step) = {np.corrcoef(dur, cost)[0,1]:.3f}; corr(prev cost, new cost) per env = {np.corrcoef(prev_cost, new_cost)[0,1]:.3f}")
    for p in range(4):
        m = prio == p
        if m.sum():
            print(f"   priority {p}: {m.sum()} waves, duration mean {dur[m].mean():,.0f} max {dur[m].max():,.0f}, cost mean {cost[m].mean():.0f}")
