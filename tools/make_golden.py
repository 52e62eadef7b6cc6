"""Generate tests/golden/hand_step_golden.npz from the f64 oracle (committed fixture: inputs + expected outputs).

Three state families for myohand_pose: (a) contact-free mid-range poses, (b) poses with joints at / past their limits,
(c) closed-fist states harvested from a random-control rollout (many contacts).  Expected values: state after one
physics substep and after one env step (10 substeps), plus tendon lengths / actuator forces / qacc of the first substep.
Also a pose/reach observation fixture computed with numpy from the reference formulas (pose_v0.py:98-138)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_mjx_amd import model as M
from oracle.oracle import Oracle

m = M.load_asset("myohand_pose")
o = Oracle(m.blob())
rng = np.random.default_rng(20251004)
lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
states = []
for _ in range(16):   # (a)
    states.append((mid + 0.5 * half * rng.uniform(-1, 1, m.nq), rng.normal(0, 0.5, m.nv), rng.uniform(0, 1, m.nu), rng.uniform(0, 1, m.nu)))
for _ in range(16):   # (b)
    q = rng.uniform(lo, hi)
    k = rng.random(m.nq) < 0.3
    q[k] = np.where(rng.random(k.sum()) < 0.5, lo[k] - 0.03, hi[k] + 0.03)
    states.append((q, rng.normal(0, 0.5, m.nv), rng.uniform(0, 1, m.nu), rng.uniform(0, 1, m.nu)))
o.reset()
for s in range(16):   # (c)
    for _ in range(12):
        o.set_state(ctrl=rng.uniform(0.2, 1.0, m.nu))
        o.step(10)
    states.append((o.field("qpos").copy(), o.field("qvel").copy(), o.field("act").copy(), rng.uniform(0, 1, m.nu)))
f32 = lambda a: np.asarray(a, np.float32)
N = len(states)
inp = {k: np.zeros((N, n), np.float32) for k, n in (("qpos", m.nq), ("qvel", m.nv), ("act", m.nu), ("ctrl", m.nu))}
out = {k: np.zeros((N, n)) for k, n in (("qpos1", m.nq), ("qvel1", m.nv), ("act1", m.nu), ("qacc1", m.nv), ("tenlen1", m.nu), ("force1", m.nu),
                                        ("qpos10", m.nq), ("qvel10", m.nv), ("act10", m.nu))}
meta = np.zeros((N, 3), np.int32)
for e, (q, v, a, c) in enumerate(states):
    inp["qpos"][e], inp["qvel"][e], inp["act"][e], inp["ctrl"][e] = f32(q), f32(v), f32(a), f32(c)
    o.reset(); o.set_state(qpos=inp["qpos"][e], qvel=inp["qvel"][e], act=inp["act"][e], ctrl=inp["ctrl"][e])
    o.step(1)
    out["qpos1"][e], out["qvel1"][e], out["act1"][e] = o.field("qpos"), o.field("qvel"), o.field("act")
    out["qacc1"][e], out["tenlen1"][e], out["force1"][e] = o.field("qacc"), o.field("actuator_length"), o.field("actuator_force")
    meta[e] = (o.nefc, o.ncon, o.solver_iter)
    o.step(9)
    out["qpos10"][e], out["qvel10"][e], out["act10"][e] = o.field("qpos"), o.field("qvel"), o.field("act")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hand_step_golden.npz"), meta=meta, **{"in_" + k: v for k, v in inp.items()}, **out)
print("wrote", N, "states; nefc", meta[:, 0], "ncon", meta[:, 1])
