"""GPU tests (pytest -m gpu) for the height-field terrain path: MyoLeg on `myolegs_terrain` (height field of
myolegs.xml:17,22 raised and colliding), myoLeg{Rough,Hilly,Stair}TerrainWalk-v0 (TerrainEnvV0, walk_v0.py:490-671).
Parity is against the oracle's restatement of mjc_ConvexHField (parity unpinned against MuJoCo itself, like the rest of the dynamics).
Tolerances: those of the leg ground-contact tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _grids(N, rng):
    hf = np.zeros((N, 100, 100), np.float32)
    for e in range(N):
        if e % 3 == 0:
            hf[e] = rng.uniform(0, 1, (100, 100)) * 0.08 - 0.02                            # rough (walk_v0.py:563-567)
        elif e % 3 == 1:
            hf[e] = 0.03 + 0.05 * np.sin(np.linspace(0, 40, 100))[:, None] * np.ones((1, 100))   # ridges across the walking direction
        else:
            hf[e] = np.linspace(-0.1, 0.15, 100)[None, :] * np.ones((100, 1))                 # slope along x
    return hf


@pytest.mark.parametrize("nsub,tq,tv", [(1, 2e-4, 0.15), (10, 2e-3, 0.3)])
def test_height_field_parity(terrain, nsub, tq, tv):
    from myosuite_mjx_amd import capi
    from oracle.oracle import Oracle
    m = terrain
    hm = capi.HipModel(m.blob(), 0)
    rng = np.random.default_rng(0)
    N, f32 = 48, np.float32
    kq, kv = np.asarray(m.key_qpos).reshape(-1, m.nq)[2], np.asarray(m.key_qvel).reshape(-1, m.nv)[2]
    q, v = np.tile(kq, (N, 1)), np.tile(kv, (N, 1)) * 0.3
    q[:, 7:] += rng.normal(0, 0.05, (N, m.nq - 7))
    q[:, 2] += rng.uniform(-0.03, 0.03, N)
    q[:, :2] += rng.uniform(-0.5, 0.5, (N, 2))
    v += rng.normal(0, 0.2, (N, m.nv))
    act, ctrl = rng.uniform(0, 1, (N, 80)), rng.uniform(0, 1, (N, 80))
    hf = _grids(N, rng)
    q, v, act, ctrl = q.astype(f32), v.astype(f32), act.astype(f32), ctrl.astype(f32)
    b = capi.HipBatch(hm, N)
    for f, a in ((capi.F_QPOS, q), (capi.F_QVEL, v), (capi.F_ACT, act), (capi.F_CTRL, ctrl), (capi.F_HFIELD, hf.reshape(N, -1))):
        b.write(f, a)
    assert np.array_equal(b.read(capi.F_HFIELD), hf.reshape(N, -1))
    b.step(None, capi.ACTMAP_NONE, nsub)
    gq, gv, dg, fl = b.read(capi.F_QPOS), b.read(capi.F_QVEL), b.read(capi.F_DIAG), b.status()
    hfg = int(m.hfield_dims[2])
    eq, ev, nc, nh = np.zeros(N), np.zeros(N), np.zeros(N, int), np.zeros(N, int)
    for e in range(N):
        o = Oracle(m.blob())
        o.set_hfield(hf[e])
        o.reset()
        o.set_state(qpos=q[e], qvel=v[e], act=act[e], ctrl=ctrl[e])
        assert o.step(nsub) == 0
        eq[e], ev[e], nc[e] = np.abs(gq[e] - o.field("qpos")).max(), np.abs(gv[e] - o.field("qvel")).max(), o.ncon
        nh[e] = sum(1 for c in o.contacts() if int(c[7]) == hfg)
    same = (fl == 0) & (dg[:, 1] == nc)
    assert same.mean() > 0.9 and nh.max() >= 10 and (nh > 0).mean() > 0.3          # prism contacts, up to tens per state
    assert eq[same].max() < tq and ev[same].max() < tv, (eq[same].max(), ev[same].max())
    assert np.median(eq) < 1e-5


def test_terrain_envs(terrain):
    """Reset draws the reference's three terrain shapes per env (ranges, flip convention, per-env randomness for rough); observation = the
    walk observation; done adds the knee condition; rollouts stay finite and the feet stand ON the terrain."""
    import torch
    import myosuite_mjx_amd as myo
    from myosuite_mjx_amd import capi
    B = 32
    for eid, kind in (("myoLegRoughTerrainWalk-v0", "rough"), ("myoLegHillyTerrainWalk-v0", "hilly"), ("myoLegStairTerrainWalk-v0", "stairs")):
        env = myo.make(eid, num_envs=B, seed=4)
        obs = env.reset(seed=4)
        assert obs.shape == (B, 403) and env.max_episode_steps == 1000
        H = env.batch.read(capi.F_HFIELD).reshape(B, 100, 100)
        if kind == "rough":
            assert np.allclose(H.min((1, 2)), -0.02, atol=1e-6) and np.allclose(H.max((1, 2)), 0.06, atol=1e-6)
            assert not np.array_equal(H[0], H[1]) and abs(H.mean() - 0.02) < 2e-3 and abs(H.std() - 0.08 / np.sqrt(12)) < 1e-3
        elif kind == "hilly":
            t = np.arange(7000) * (3 * np.pi / 6999)
            norm = np.concatenate((np.ones(3000), 0.5 + 0.5 * np.cos(t)))
            assert np.allclose(H[3], np.flip(norm.reshape(100, 100) * 0.63, [0, 1]), atol=2e-6)
        else:
            rows = np.concatenate((np.zeros(52), np.repeat(0.1 * np.arange(12) / 3.2, 4)))
            assert np.allclose(H[5], np.flip(np.tile(rows[:, None], (1, 100)) * 2.5, [0, 1]), atol=2e-6)
        g = torch.Generator(device="cuda").manual_seed(0)
        ndone = 0
        for k in range(60):
            obs, rwd, term, trunc, info = env.step(torch.rand((B, 80), device="cuda", generator=g) * 2 - 1)
            ndone += int(term.sum())
            assert torch.isfinite(obs).all() and torch.isfinite(rwd).all()
        st = env.get_env_state()
        assert np.isfinite(st["qpos"]).all() and (env.status() & ~capi.FLAG_CONTACT_OVERFLOW == 0).all()
        # feet heights (obs 80..81) sit above the local terrain, within a step height
        o = obs.cpu().numpy()
        sb = 33 + 34 + 6
        assert (o[:, sb] > -0.03).all() and (o[:, sb + 1] > -0.03).all() and ndone > 0        # random muscle noise does fall over eventually
    # the knee condition (walk_v0.py:660-671): crouches of increasing depth -- done = COM below min_height, or COM closer than 0.61 m to the feet;
    # the plain walk env only knows the first rule
    env_t = myo.make("myoLegRoughTerrainWalk-v0", num_envs=4, seed=1, autoreset=False)
    env_w = myo.make("myoLegWalk-v0", num_envs=4, seed=1, autoreset=False)
    knee = np.array([0.3, 1.2, 1.8, 2.05])
    for env in (env_t, env_w):
        env.reset(seed=1)
        st = env.get_env_state()
        q = st["qpos"].copy()
        for n, val in (("knee_angle_r", knee), ("knee_angle_l", knee), ("hip_flexion_r", 0.6 * knee), ("hip_flexion_l", 0.6 * knee)):
            q[:, terrain.jnt_qposadr[terrain.name2id("joint", n)]] = val
        env.set_env_state({"qpos": q, "qvel": st["qvel"] * 0})
        env.batch.obs()
    sb = 33 + 34 + 6
    for env, with_knee in ((env_t, True), (env_w, False)):
        o = env.view(capi.F_OBS).cpu().numpy()
        h, gap = o[:, sb + 2], o[:, sb + 2] - 0.5 * (o[:, sb] + o[:, sb + 1])
        want = (h < 0.8) | ((gap < 0.61) & with_knee)
        assert np.array_equal(env.view(capi.F_DONE).cpu().numpy()[:, 0] > 0, want), (h, gap)
        if with_knee:
            assert ((gap < 0.61) & (h >= 0.8)).any() and (gap >= 0.61).any()      # the sample exercises the knee rule on its own
