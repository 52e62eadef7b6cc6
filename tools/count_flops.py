"""Algorithmic flop count of one env step, from the INSTRUMENTED build of the oracle (oracle/flopcount.h; SURVEY.md 8d: the figure
of record).  CPU only.   python tools/count_flops.py [env ...]  ->  profiles/r2_flops_oracle.json

Workload = the bench's: reset states of the env (myoHandPoseRandom-v0: every joint uniform over its range), U(-1,1) actions through
the muscle sigmoid (base_v0.py:83-119), frame_skip substeps per env step; counts are averaged over envs and over the env steps of
a short rollout (so that contact-rich reset poses and settled poses are both in the sample)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_mjx_amd import model as M  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

LIB = os.path.join(ROOT, "oracle", "libmyo_oracle_flops.so")
CASES = {"myoHandPoseRandom-v0": ("myohand_pose", 10, "uniform"), "myoLegWalk-v0": ("myolegs", 10, "key2"), "myoFingerPoseFixed-v0": ("myofinger_v0", 10, "qpos0")}


def count(env_id, n_envs=96, n_steps=12, seed=0):
    asset, frame_skip, init = CASES[env_id]
    m = M.load_asset(asset)
    o = Oracle(m.blob(), lib_path=LIB)
    o.lib.myoo_flops.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    rng = np.random.default_rng(seed)
    buf = (C.c_uint64 * 5)()
    tot = np.zeros(5)
    ncon = nefc = iters = 0
    for e in range(n_envs):
        o.reset()
        if init == "uniform":
            o.set_state(qpos=rng.uniform(m.jnt_range[:, 0], m.jnt_range[:, 1]))
        elif init == "key2":
            o.set_state(qpos=np.asarray(m.key_qpos).reshape(-1, m.nq)[2], qvel=np.asarray(m.key_qvel).reshape(-1, m.nv)[2])
        for s in range(n_steps):
            a = rng.uniform(-1, 1, m.nu)
            o.set_state(ctrl=1.0 / (1.0 + np.exp(-5.0 * (a - 0.5))))
            o.lib.myoo_flops(buf, 1)
            o.step(frame_skip)
            o.lib.myoo_flops(buf, 1)
            tot += np.array(buf[:], float)
            ncon += o.ncon; nefc += o.nefc; iters += o.solver_iter
    k = n_envs * n_steps
    add, mul, div, sq, sp = tot / k
    return dict(env=env_id, model=asset, frame_skip=frame_skip, envs=n_envs, env_steps_each=n_steps,
                per_env_step=dict(add_sub=add, mul=mul, div=div, sqrt=sq, special=sp, total=add + mul + div + sq + sp),
                mean_last_substep=dict(ncon=ncon / k, nefc=nefc / k, newton_iterations=iters / k),
                convention="1 per + - * /, sqrt and transcendental call; comparisons / copies / abs free; dense efc_J rows and dense Newton Cholesky as the oracle executes them")


if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "flops"])
    envs = sys.argv[1:] or list(CASES)
    out = [count(e) for e in envs]
    for r in out:
        print(r["env"], f"{r['per_env_step']['total'] / 1e6:.3f} Mflop per env step", r["mean_last_substep"])
    if not sys.argv[1:]:
        json.dump(out, open(os.path.join(ROOT, "profiles", "r2_flops_oracle.json"), "w"), indent=1)
