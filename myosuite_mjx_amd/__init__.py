"""myosuite_mjx_amd -- MI355X-native batched stepper for the MyoSuite env.step() hot path.

Python package name note: the repository brief calls the package `myosuite-mjx_amd`; a hyphen is not
importable, so the directory is `myosuite_mjx_amd`.

    import myosuite_mjx_amd as myo
    env = myo.make("myoHandPoseRandom-v0", num_envs=4096)      # gym-style batched env (envs.py)
    obs = env.reset(seed=0); obs, rwd, term, trunc, info = env.step(action)

    m = myo.put_model("myohand_pose")                           # MJX-flavour functional API (sim.py)
    d = myo.make_data(m, 4096); myo.step(m, d, ctrl, nsubsteps=10)
"""
from .envs import REGISTRY, UNSUPPORTED, BatchedMyoEnv, make  # noqa: F401
from .policy import BraxPolicy  # noqa: F401
from .sim import HipSimScene, get, make_data, put_model, set_, step  # noqa: F401
from . import trace  # noqa: F401  (batched rollouts <-> the reference's Trace logger layout)

__version__ = "0.1.0"
