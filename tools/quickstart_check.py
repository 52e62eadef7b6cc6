import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, myosuite_mjx_amd as myo
env = myo.make("myoHandPoseRandom-v0", num_envs=4096)
obs = env.reset(seed=0)
for _ in range(5):
    act = torch.rand((4096, env.act_dim), device=obs.device) * 2 - 1
    obs, rew, terminated, truncated, info = env.step(act)
m = myo.put_model("myohand_pose"); d = myo.make_data(m, 1024)
myo.step(m, d, ctrl=None, nsubsteps=10); qpos = myo.get(d, "qpos")
rng = np.random.default_rng(0); sizes = [108, 32, 32, 32, 32, 78]
np.savez("/tmp/policy.npz", obs_mean=np.zeros(108, np.float32), obs_std=np.ones(108, np.float32),
         **{f"w{i}": rng.normal(0, 0.1, (sizes[i], sizes[i+1])).astype(np.float32) for i in range(5)}, **{f"b{i}": np.zeros(sizes[i+1], np.float32) for i in range(5)})
policy = myo.BraxPolicy.from_npz("/tmp/policy.npz")
policy.act(obs.data_ptr(), 4096, act.data_ptr(), deterministic=False, seed=1, step=0)
tenv = myo.make("MyoHandAirplaneRandom-v0", num_envs=4096)
tobs = tenv.reset()
tobs, trew, tterm, ttrunc, tinfo = tenv.step(torch.rand((4096, tenv.act_dim), device="cuda") * 2 - 1)
torch.cuda.synchronize()
print("quickstart ok", obs.shape, qpos.shape, float(act.abs().max()), tobs.shape, float(trew.mean()))
