"""Developer probe (diagnostic build): where do the 4096 workgroups of one launch land (XCC / SE / CU / SIMD)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MYO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myosuite_mjx_amd", "libmyo_hip_stamps.so")
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv
B = 4096
env = BatchedMyoEnv("myoHandPoseRandom-v0", num_envs=B, as_torch=False)
env.batch.set_balance(0)
env.reset(seed=1)
env.batch.bench_rollout(3, 10, 0, 0, 0, None)
st, ok = capi.read_stamps(env.batch, B)
hw = st[:, 10].astype(np.int64); xcc = st[:, 11].astype(np.int64) & 0xF
wave = hw & 0xF; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("first 40 WGs: (xcc, se, sh, cu, simd, wave)")
for w in range(40):
    print(w, xcc[w], se[w], sh[w], cu[w], simd[w], wave[w])
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
u, cnt = np.unique(key, return_counts=True)
print("distinct SIMDs", len(u), "waves per SIMD min/max", cnt.min(), cnt.max())
# which WG ids share SIMD with WG 0?
print("WGs sharing the SIMD of WG0:", np.where(key == key[0])[0][:16])
print("WGs sharing the CU of WG0:", np.where(key // 10 == key[0] // 10)[0][:32])
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "census_key.npy"), key)
