"""Tiny driver for rocprofv3: a short random-action rollout of the bench workload (no torch, no oracle)."""
import os, sys
os.environ.setdefault("MYO_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv
B = int(os.environ.get("B", 4096)); steps = int(os.environ.get("STEPS", 20))
if os.environ.get("LANES"): capi.set_lanes(int(os.environ["LANES"]))
env = BatchedMyoEnv(os.environ.get("ENV", "myoHandPoseRandom-v0"), num_envs=B, as_torch=False)
env.reset(seed=1)
mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
env.batch.bench_rollout(30, 10, 0, mode, env.max_episode_steps, None)
ms = env.batch.bench_rollout(steps, 10, 0, mode, env.max_episode_steps, None)
print(f"B={B} steps={steps} ms/step={ms/steps:.3f}")
