"""Developer probe (diagnostic build -DMYO_STAMPS=1): which stages make the slowest waves of a 4096-env launch slow -- stage cycles of the slowest
2 % of the waves against the mean wave (headline workload, mid-rollout)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MYO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myosuite_mjx_amd", "libmyo_hip_stamps.so")
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv

B = int(os.environ.get("B", 4096))
NAMES = ["load/check", "kinematics", "tendon+muscle", "dynamics", "narrow phase", "rows", "frames+broad", "newton", "euler", "store"]
env = BatchedMyoEnv("myoHandPoseRandom-v0", num_envs=B, as_torch=False)
mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
env.batch.set_balance(0)      # workgroup w = env w, so that stamps (per workgroup) and work features (per env) line up
env.reset(seed=1)
env.batch.bench_rollout(40, 10, 0, mode, env.max_episode_steps, None)
env.batch.bench_rollout(1, 10, 0, 0, 0, None)
st2, ok = capi.read_stamps(env.batch, 2 * B)
st = st2[:B, :10].astype(float)
tot = st.sum(1)
slow = np.argsort(tot)[-B // 50:]
d = env.batch.read(capi.F_DIAG)
print(f"per-wave cycles: mean {tot.mean():,.0f}  p98 {np.percentile(tot, 98):,.0f}  max {tot.max():,.0f}")
print(f"{'stage':16s} {'mean wave':>12s} {'slowest 2 %':>12s} {'difference':>12s}")
for k, n in enumerate(NAMES):
    print(f"{n:16s} {st[:, k].mean():12,.0f} {st[slow, k].mean():12,.0f} {st[slow, k].mean() - st[:, k].mean():12,.0f}")
print("slowest 2 %%: contacts (last substep) %.1f vs %.1f; candidates / substep %.0f vs %.0f; mpr evals / substep %.1f vs %.1f; newton iterations / substep %.2f vs %.2f" % (
    d[slow, 1].mean(), d[:, 1].mean(), (d[slow, 4] & 0xFFFF).mean() / 10, (d[:, 4] & 0xFFFF).mean() / 10, d[slow, 5].mean() / 10, d[:, 5].mean() / 10,
    (d[slow, 6] >> 16).mean() / 10, (d[:, 6] >> 16).mean() / 10))
# where do the slow waves sit?  HW_REG_HW_ID (gfx9 layout): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
hw = st2[:B, 10].astype(np.int64)
xcc = st2[:B, 11].astype(np.int64) & 0xFF
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ncu = len(np.unique(key))
per_cu = np.bincount(np.unique(key, return_inverse=True)[1])
skey = key * 4 + simd
uniq, inv = np.unique(skey, return_inverse=True)
per_simd = np.bincount(inv)
print(f"census: {ncu} distinct CUs, waves per CU min {per_cu.min()} max {per_cu.max()}; waves per SIMD histogram {np.bincount(per_simd)}")
mean_by_simd_load = [tot[per_simd[inv] == n].mean() if (per_simd[inv] == n).any() else 0 for n in range(per_simd.max() + 1)]
print("mean wave cycles by number of waves on its SIMD:", [f"{n}: {m:,.0f}" for n, m in enumerate(mean_by_simd_load) if m])
print("mean wave cycles by XCD:", [f"{x}: {tot[xcc == x].mean():,.0f}" for x in range(8)])
slow_set = np.zeros(B, bool); slow_set[slow] = True
print("slowest 2 %: waves per SIMD of their SIMD:", np.bincount(per_simd[inv][slow_set]), " XCD:", np.bincount(xcc[slow_set], minlength=8))
simd_max = np.zeros(len(uniq)); np.maximum.at(simd_max, inv, tot)
simd_sum = np.bincount(inv, weights=tot)
print(f"per SIMD: sum of its waves' cycles mean {simd_sum.mean():,.0f} max {simd_sum.max():,.0f}; slowest wave of a SIMD mean {simd_max.mean():,.0f}")
