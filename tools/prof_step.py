"""Tiny driver for rocprofv3: a short random-action rollout of a bench workload (ENV, B, STEPS from the environment; no oracle, and no torch
except for the MyoDM TrackEnv, whose host class views device rows through it)."""
import os, sys
ENV = os.environ.get("ENV", "myoHandPoseRandom-v0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_mjx_amd.envs import REGISTRY
TRACK = REGISTRY.get(ENV, {}).get("task") == "track"
if not TRACK:
    os.environ.setdefault("MYO_NO_TORCH", "1")
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import make
B = int(os.environ.get("B", 4096)); steps = int(os.environ.get("STEPS", 20))
if os.environ.get("LANES"): capi.set_lanes(int(os.environ["LANES"]))
env = make(ENV, num_envs=B, autoreset=True) if TRACK else make(ENV, num_envs=B, as_torch=False)
env.reset(seed=1)
mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
env.batch.bench_rollout(30, env.frame_skip, 0, mode, env.max_episode_steps, None)
ms = env.batch.bench_rollout(steps, env.frame_skip, 0, mode, env.max_episode_steps, None)
print(f"{ENV} B={B} steps={steps} ms/step={ms/steps:.3f}")
