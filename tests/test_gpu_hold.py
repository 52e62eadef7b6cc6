"""GPU tests (pytest -m gpu) for the MyoHand + free object model (envs/myo/assets/hand/myohand_hold.xml) and myoHandObjHoldFixed-v0
(obj_hold_v0.py): a second root link carrying a free joint, ellipsoid-object / hand contacts through the generic convex narrow phase,
the object over the scene's floor plane and pedestal cylinder (world-fixed geoms), on the 36-dof wave kernel.
Tolerances: as for the hand's contact states, with the object's angular velocity (inertia ~5e-5 kg m^2) setting the qvel bound."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hold():
    from myosuite_mjx_amd import model as M
    return M.load_asset("myohand_hold")


def _states(m, N, seed):
    rng = np.random.default_rng(seed)
    f32 = np.float32
    q = np.tile(m.qpos0, (N, 1))
    q[:, :23] = 0
    q[:, 0] = -1.5                                                # palm up (obj_hold_v0.py:63-64)
    q[:, :23] += rng.normal(0, 0.15, (N, 23))
    lo, hi = m.jnt_range[:23, 0], m.jnt_range[:23, 1]
    q[:, :23] = np.clip(q[:, :23], lo + 0.01, hi - 0.01)
    q[:, 23:26] += rng.normal(0, 0.003, (N, 3)) + np.array([0, 0, 0.003])     # resting on / pressed a few mm into the palm
    quat = rng.normal(0, 1, (N, 4))
    q[:, 26:30] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    v = rng.normal(0, 0.5, (N, m.nv))
    v[:, 23:26] *= 0.2
    return q.astype(f32), v.astype(f32), rng.uniform(0, 1, (N, 39)).astype(f32), rng.uniform(0, 1, (N, 39)).astype(f32)


@pytest.mark.parametrize("nsub,tq,tv", [(1, 2e-5, 2e-2), (10, 2e-3, 0.2)])     # 10 substeps: the 0.11 kg object (inertia 1e-4) rolling on finger pads amplifies float32 round-off in single envs; medians below
def test_hand_object_parity(hold, nsub, tq, tv):
    from myosuite_mjx_amd import capi
    from oracle.oracle import Oracle
    m = hold
    assert (m.nq, m.nv, m.nu) == (30, 29, 39) and int(m.hip_flags[0]) == 1
    hm = capi.HipModel(m.blob(), 0)
    o = Oracle(m.blob())
    N = 96
    q, v, act, ctrl = _states(m, N, 3)
    b = capi.HipBatch(hm, N)
    for f, a in ((capi.F_QPOS, q), (capi.F_QVEL, v), (capi.F_ACT, act), (capi.F_CTRL, ctrl)):
        b.write(f, a)
    b.step(None, capi.ACTMAP_NONE, nsub)
    gq, gv, dg, fl = b.read(capi.F_QPOS), b.read(capi.F_QVEL), b.read(capi.F_DIAG), b.status()
    eq, ev, nc, objc = np.zeros(N), np.zeros(N), np.zeros(N, int), np.zeros(N, int)
    obj_geom = m.name2id("geom", "object")
    for e in range(N):
        o.reset()
        o.set_state(qpos=q[e], qvel=v[e], act=act[e], ctrl=ctrl[e])
        assert o.step(nsub) == 0
        eq[e], ev[e], nc[e] = np.abs(gq[e] - o.field("qpos")).max(), np.abs(gv[e] - o.field("qvel")).max(), o.ncon
        objc[e] = sum(1 for c in o.contacts() if int(c[7]) == obj_geom or int(c[8]) == obj_geom)
    same = (fl == 0) & (dg[:, 1] == nc)
    assert same.mean() > 0.9 and (objc > 0).mean() > 0.5 and objc.max() >= 3          # the object really touches the hand
    assert np.abs(np.linalg.norm(gq[:, 26:30], axis=1) - 1).max() < 1e-6               # quaternion of the free object stays unit
    assert eq[same].max() < tq and ev[same].max() < tv, (eq[same].max(), ev[same].max())
    assert np.median(eq[same]) < (2e-6 if nsub == 1 else 3e-5) and np.median(ev[same]) < (2e-3 if nsub == 1 else 5e-3), (np.median(eq[same]), np.median(ev[same]))


def test_object_rolls_off_falls_and_lands(hold):
    """No muscle drive: the object rolls off the open hand, falls 1.4 m and lands on the scene's pedestal (a world-fixed cylinder, top at
    z = 0.015, generic convex narrow phase) or, past its rim, on the floor plane at z = -0.4.  The first second (rolling in the palm, free
    flight) is compared with the oracle env by env; after the 5 m/s impact the motion is chaotic, so the end state is checked for
    physical sanity only: resting on one of the two supports, nothing tunnels."""
    from myosuite_mjx_amd import capi
    from oracle.oracle import Oracle
    m = hold
    hm = capi.HipModel(m.blob(), 0)
    N = 64
    rng = np.random.default_rng(1)
    q = np.tile(m.qpos0, (N, 1)).astype(np.float32)
    q[:, :23] = 0
    q[:, 0] = -1.5
    q[:, 23:25] += rng.normal(0, 0.02, (N, 2)).astype(np.float32)
    b = capi.HipBatch(hm, N)
    b.write(capi.F_QPOS, q)
    orc = []
    for e in range(6):
        o = Oracle(m.blob())
        o.reset()
        o.set_state(qpos=q[e], qvel=np.zeros(m.nv), ctrl=np.zeros(39))
        orc.append(o)
    for k in range(150):
        b.step(None, capi.ACTMAP_NONE, 10)
        if k < 50:
            for o in orc:
                assert o.step(10) == 0
        if k in (10, 30, 49):
            gq = b.read(capi.F_QPOS)
            for e, o in enumerate(orc):
                assert np.abs(gq[e, 23:26] - o.field("qpos")[23:26]).max() < (2e-4 if k == 10 else 5e-3), (k, e)
    gq, gv = b.read(capi.F_QPOS), b.read(capi.F_QVEL)
    assert (b.status() == 0).all() and np.isfinite(gq).all()
    z = gq[:, 25]
    on_pedestal = np.abs(z - (0.015 + 0.030)) < 0.012          # object half-sizes 25 / 36 / 30 mm: resting height depends on its attitude
    on_floor = np.abs(z - (-0.4 + 0.030)) < 0.012
    in_flight_or_hand = ~(on_pedestal | on_floor)
    assert z.min() > -0.4 + 0.015 and (on_pedestal | on_floor).mean() > 0.8, (z.min(), on_pedestal.mean(), on_floor.mean())
    assert in_flight_or_hand.mean() < 0.2


def test_obj_hold_env(hold):
    """myoHandObjHoldFixed-v0: obs 91 = hand qpos 23, hand qvel*dt 23, obj_pos 3, obj_err 3, act 39 (obj_hold_v0.py:15,66-79); reward
    100*(-d) + 4*bonus - 10*drop with goal_th 0.010, drop 0.300 (:96-118); 75-step episodes, reset to the open palm-up hand."""
    import torch
    import myosuite_mjx_amd as myo
    env = myo.make("myoHandObjHoldFixed-v0", num_envs=48, seed=3, autoreset=False)
    obs = env.reset(seed=3)
    assert obs.shape == (48, 91) and env.max_episode_steps == 75 and abs(env.dt - 0.02) < 1e-9
    st = env.get_env_state()
    assert np.allclose(st["qpos"][:, 1:23], 0) and np.allclose(st["qpos"][:, 0], -1.5) and np.allclose(st["qpos"][:, 23:], hold.qpos0[23:], atol=1e-6)
    goal = np.array([-0.240, -0.520, 1.470])
    g = torch.Generator(device="cuda").manual_seed(0)
    for k in range(30):
        obs, rwd, term, trunc, info = env.step(torch.rand((48, 39), device="cuda", generator=g) * 2 - 1)
        if k in (0, 29):
            st = env.get_env_state()
            o = obs.cpu().numpy()
            assert np.allclose(o[:, :23], st["qpos"][:, :23], atol=1e-7) and np.allclose(o[:, 23:46], st["qvel"][:, :23] * 0.02, atol=1e-6)
            assert np.allclose(o[:, 46:49], st["qpos"][:, 23:26], atol=1e-6) and np.allclose(o[:, 49:52], goal - st["qpos"][:, 23:26], atol=1e-6)
            assert np.allclose(o[:, 52:], st["act"], atol=1e-7)
            d = np.linalg.norm(o[:, 49:52], axis=1)
            edge = (np.abs(d - 0.01) < 1e-5) | (np.abs(d - 0.02) < 1e-5) | (np.abs(d - 0.3) < 1e-5)
            ref = -100.0 * d + 4.0 * ((d < 0.02) * 1.0 + (d < 0.01) * 1.0) - 10.0 * (d > 0.3)
            assert np.allclose(rwd.cpu().numpy()[~edge], ref[~edge], atol=2e-4)
            assert np.array_equal(term.cpu().numpy()[~edge], (d > 0.3)[~edge])
    assert (env.status() == 0).all()
    # with auto-reset: dropped objects start a new episode at the reset pose
    env2 = myo.make("myoHandObjHoldFixed-v0", num_envs=48, seed=3)
    env2.reset(seed=3)
    ndone = 0
    for k in range(75):
        obs, rwd, term, trunc, info = env2.step(torch.rand((48, 39), device="cuda", generator=g) * 2 - 1)
        ndone += int(term.sum())
    assert trunc.all() or ndone > 0


def test_per_env_object_size(hold):
    """ObjHoldRandomEnvV0 edits model.geom_size of the object per episode (obj_hold_v0.py:133-139): per-env size of that one collision geom
    (MYO_F_GEOMSIZE) against oracle instances whose model carries the same edit; then the env: sizes and goals re-drawn per episode."""
    import torch
    import myosuite_mjx_amd as myo
    from myosuite_mjx_amd import capi
    from oracle.oracle import Oracle
    m = hold
    og = m.name2id("geom", "object")
    hm = capi.HipModel(m.blob(), 0)
    N = 48
    q, v, act, ctrl = _states(m, N, 7)
    rng = np.random.default_rng(7)
    sizes = rng.uniform(0.020, 0.030, (N, 3)).astype(np.float32)
    b = capi.HipBatch(hm, N)
    b.set_geom_override(og, (0.02,) * 3, (0.03,) * 3)
    for f, a in ((capi.F_QPOS, q), (capi.F_QVEL, v), (capi.F_ACT, act), (capi.F_CTRL, ctrl),
                 (capi.F_GEOMSIZE, np.concatenate([sizes, sizes.max(1, keepdims=True)], 1))):
        b.write(f, a)
    b.step(None, capi.ACTMAP_NONE, 1)
    gq, gv, dg, fl = b.read(capi.F_QPOS), b.read(capi.F_QVEL), b.read(capi.F_DIAG), b.status()
    eq, ev, same, differs = np.zeros(N), np.zeros(N), np.zeros(N, bool), 0
    for e in range(N):
        o = Oracle(m.blob())
        o.set_geom_size(og, sizes[e])
        o.reset()
        o.set_state(qpos=q[e], qvel=v[e], act=act[e], ctrl=ctrl[e])
        assert o.step(1) == 0
        eq[e], ev[e] = np.abs(gq[e] - o.field("qpos")).max(), np.abs(gv[e] - o.field("qvel")).max()
        same[e] = fl[e] == 0 and dg[e, 1] == o.ncon
        o0 = Oracle(m.blob())                       # the unedited model gives a different answer: the override is really in effect
        o0.reset()
        o0.set_state(qpos=q[e], qvel=v[e], act=act[e], ctrl=ctrl[e])
        o0.step(1)
        differs += int(np.abs(o0.field("qvel") - o.field("qvel")).max() > 1e-3)
    assert same.mean() > 0.9 and differs > N // 3
    assert eq[same].max() < 2e-5 and ev[same].max() < 2e-2, (eq[same].max(), ev[same].max())
    env = myo.make("myoHandObjHoldRandom-v0", num_envs=64, seed=2)
    obs = env.reset(seed=2)
    assert obs.shape == (64, 91)
    g0 = env.batch.read(capi.F_GEOMSIZE)
    st = env.get_env_state()
    assert (g0[:, :3] >= 0.02 - 1e-7).all() and (g0[:, :3] <= 0.03 + 1e-7).all() and np.allclose(g0[:, 3], g0[:, :3].max(1)) and g0[:, :3].std() > 0.002
    c = np.asarray(m.qpos0[-7:-4])
    assert (np.abs(st["target"] - c[None, :]) <= 0.03 + 1e-6).all() and st["target"].std(0).min() > 0.01
    gen = torch.Generator(device="cuda").manual_seed(0)
    for k in range(76):                                         # one whole 75-step episode: every env is reset at least once
        env.step(torch.rand((64, 39), device="cuda", generator=gen) * 2 - 1)
    g1 = env.batch.read(capi.F_GEOMSIZE)
    assert (np.abs(g1[:, :3] - g0[:, :3]).max(1) > 1e-4).all() and (g1[:, :3] >= 0.02 - 1e-7).all() and (g1[:, :3] <= 0.03 + 1e-7).all()
    assert (env.status() == 0).all()
