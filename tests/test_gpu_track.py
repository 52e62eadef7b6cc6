"""GPU parity for the MyoDM TrackEnv model class (myohand_object.xml + airplane): the TRK instantiation of the wave
kernel (condim-4 pyramids, joint friction loss, box / convex-hull narrow phase, position actuators) against the f64 oracle.
States come from the reference's own motion file MyoHand_airplane_fly1.npz (fixture tests/golden/ref_motion.npz): the pre-grasp, grasp
and lift frames put the hand on the object and the object on / above the table."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def track():
    from myosuite_mjx_amd import model as M
    return M.load_asset("myohand_object_airplane")


@pytest.fixture(scope="module")
def hm(track):
    from myosuite_mjx_amd import capi
    return capi.HipModel(track.blob(), 0)


def _motion_states(m, frames, seed, jitter=0.02):
    """qpos of the reference motion's frames: robot[29] | object position | object euler angles (mjx/myodm_v0.py:147-150 convention)."""
    from myosuite_mjx_amd import track as T
    f = np.load(os.path.join(ROOT, "tests", "golden", "ref_motion.npz"))
    R, O = f["track_MyoHand_airplane_fly1__in__robot"], f["track_MyoHand_airplane_fly1__in__object"]
    rng = np.random.default_rng(seed)
    q = np.zeros((len(frames), m.nq))
    for i, t in enumerate(frames):
        q[i, :29] = R[t] + rng.normal(0, jitter, 29) * (np.arange(29) >= 6)
        q[i, 29:32] = O[t, :3]
        q[i, 32:35] = T.quat2euler(O[t, 3:])
    v = rng.normal(0, 0.2, (len(frames), m.nv))
    act = rng.uniform(0, 1, (len(frames), m.nu)); act[:, :6] = 0
    ctrl = rng.uniform(0, 1, (len(frames), m.nu)); ctrl[:, :6] = q[:, :6]            # position actuators: hold the base where it is
    return q.astype(np.float32), v.astype(np.float32), act.astype(np.float32), ctrl.astype(np.float32)


def _run(m, hm, qpos, qvel, act, ctrl, nsub, switches=(0, 0, 0)):
    from myosuite_mjx_amd import capi
    from oracle.oracle import Oracle
    N = len(qpos)
    hm.set_switch(*switches)
    b = capi.HipBatch(hm, N)
    for f, a in ((capi.F_QPOS, qpos), (capi.F_QVEL, qvel), (capi.F_ACT, act), (capi.F_CTRL, ctrl)):
        b.write(f, a)
    b.step(None, capi.ACTMAP_NONE, nsub)
    g = dict(qpos=b.read(capi.F_QPOS), qvel=b.read(capi.F_QVEL), act=b.read(capi.F_ACT), qacc=b.read(capi.F_QACC), diag=b.read(capi.F_DIAG), flags=b.status(),
             linkx=b.read(capi.F_LINKX))
    o = Oracle(m.blob())
    o.switches(*switches)
    r = dict(qpos=np.zeros_like(qpos, np.float64), qvel=np.zeros_like(qvel, np.float64), ncon=np.zeros(N, int), nefc=np.zeros(N, int), ncon_sum=np.zeros(N, int),
             ncon_max=np.zeros(N, int), xpos=np.zeros((N, m.nbody, 3)))
    for e in range(N):
        o.reset()
        o.set_state(qpos=qpos[e], qvel=qvel[e], act=act[e], ctrl=ctrl[e])
        for s in range(nsub):
            o.step(1)
            r["ncon_sum"][e] += o.ncon
            r["ncon_max"][e] = max(r["ncon_max"][e], o.ncon)
        r["qpos"][e], r["qvel"][e], r["ncon"][e], r["nefc"][e] = o.field("qpos"), o.field("qvel"), o.ncon, o.nefc
        r["xpos"][e] = o.field("xpos").reshape(-1, 3)         # position stage of the LAST substep (the oracle's step = forward, then integrate)
    hm.set_switch(0, 0, 0)
    return g, r


@pytest.mark.parametrize("nsub,tq,tv", [(1, 5e-6, 5e-3), (5, 5e-5, 2e-2)])
def test_track_smooth_dynamics_and_friction_loss(track, hm, nsub, tq, tv):
    """No contacts, no limits: arm base (slides + hinges, damping 20, position actuators), hand muscles, object joints with friction loss."""
    q, v, a, c = _motion_states(track, list(range(0, 96, 2)), 1)
    g, r = _run(track, hm, q, v, a, c, nsub, (1, 1, 1))
    assert (g["flags"] == 0).all()
    assert (g["diag"][:, 0] == 6).all() and (r["nefc"] == 6).all()                   # the six friction-loss rows of the object's joints
    assert np.abs(g["qpos"] - r["qpos"]).max() < tq and np.abs(g["qvel"] - r["qvel"]).max() < tv


@pytest.mark.parametrize("nsub,q90,qmax,v90", [(1, 2e-5, 2e-4, 1e-2), (5, 2e-4, 2e-3, 3e-2)])
def test_track_contacts(track, hm, nsub, q90, qmax, v90):
    """Everything on: hull meshes on the table-top box (condim 4, 6 rows each), fingers on the hulls (MPR with polytope supports),
    joint limits; up to 53 simultaneous contacts / 292 constraint rows (the HBM overflow rows are in play).  Face-on-face contacts of
    polytopes leave the contact POINT to round-off (any point of the overlap is a valid MPR answer), so single envs differ more than
    with the smooth hand geoms; the float32 build of the oracle shows the same spread (measured, tools/gpu_track_probe.py: p50 / p90 / max
    2.4e-6 / 1.2e-5 / 2.6e-5 after one substep, 1.3e-5 / 7.6e-5 / 2.6e-4 after five; HIP: 2.7e-6 / 9.7e-6 / 1.0e-4 and 1.4e-5 / 5.9e-5 / 4.3e-4
    on the envs whose contact history agrees).  TrackEnv steps 5 substeps per env step (mjx/myodm_v0.py:45)."""
    q, v, a, c = _motion_states(track, list(range(0, 48)) + list(range(0, 48)), 2, jitter=0.01)
    g, r = _run(track, hm, q, v, a, c, nsub)
    assert (g["flags"] == 0).all()
    same = (g["diag"][:, 1] == r["ncon"]) & ((g["diag"][:, 4] >> 16) == r["ncon_sum"])
    assert same.mean() > 0.95, same.mean()
    assert r["ncon"].max() >= 40 and r["nefc"].max() >= 200
    eq, ev = np.abs(g["qpos"] - r["qpos"]).max(1), np.abs(g["qvel"] - r["qvel"]).max(1)
    assert np.percentile(eq, 90) < q90 and np.percentile(ev, 90) < v90, (np.percentile(eq, 90), np.percentile(ev, 90))
    assert eq[same].max() < qmax, eq[same].max()
    assert eq.max() < 1e-2


def test_link_frames_export_matches_oracle_xpos(track, hm):
    """MYO_F_LINKX = link frames of the last substep's position stage: body positions rebuilt from them (hip_body_link / lpos tables)
    equal the oracle's xpos of the same stage (what MJX's data.xpos holds after mjx.step)."""
    from oracle import track_ref as TR
    q, v, a, c = _motion_states(track, [0, 10, 20, 30, 40], 3)
    g, r = _run(track, hm, q, v, a, c, 5, (1, 0, 1))
    xpos, xmat = TR.body_frames(track, g["linkx"])
    for bname in ("airplane", "lunate", "distal_thumb"):
        bid = track.name2id("body", bname)
        assert np.abs(xpos[:, bid] - r["xpos"][:, bid]).max() < 2e-5


def test_trackenv_rollout_against_oracle_and_reward(track):
    """TrackEnv.reset / step (mjx/myodm_v0.py:152-173, 269-304) on a TRACK-type reference (the reference's airplane_fly1 motion): action
    scaling onto actuator_ctrlrange, 5 substeps, obs = [qpos, qvel], reward / done from the body frames of the last substep's position
    stage -- all against the oracle stepped with the same controls and TrackReward evaluated on the oracle's frames."""
    import torch
    from myosuite_mjx_amd import track as T
    from oracle import track_ref as TR
    from oracle.oracle import Oracle
    from test_track_host import _oracle_linkx
    f = np.load(os.path.join(ROOT, "tests", "golden", "ref_motion.npz"))
    motion = {k.split("__in__")[1]: f[k] for k in f.files if k.startswith("track_MyoHand_airplane_fly1__in__")}
    B = 8
    env = T.TrackEnv(num_envs=B, reference=motion, motion_extrapolation=True, seed=0)
    obs = env.reset()
    m = track
    assert obs.shape == (B, 70) and np.allclose(obs[:, :35].cpu().numpy(), env.init_qpos[None], atol=0) and float(obs[:, 35:].abs().max()) == 0
    assert np.allclose(env.init_qpos[:29], motion["robot_init"], atol=1e-6) and np.allclose(env.init_qpos[29:32], motion["object_init"][:3], atol=1e-6)
    rng = np.random.default_rng(5)
    oracles = [Oracle(m.blob()) for _ in range(B)]
    for o in oracles:
        o.reset(); o.set_state(qpos=env.init_qpos.astype(np.float64), qvel=np.zeros(m.nv))
    cr = np.asarray(m.actuator_ctrlrange, float)
    rw = TR.TrackReward(m)
    worst_q = worst_r = 0.0
    for k in range(3):
        a = rng.uniform(-1, 1, (B, m.nu)).astype(np.float32)
        obs, reward, done, info = env.step(a)
        ref = env.ref.get_reference(np.full(B, k * env.dt))
        for e, o in enumerate(oracles):
            o.set_state(ctrl=(a[e].astype(np.float64) + 1) * (cr[:, 1] - cr[:, 0]) * 0.5 + cr[:, 0])
            assert o.step(4) == 0
            o.forward()
            linkx = _oracle_linkx(m, o)                         # position stage of the 5th substep
            assert o.step(1) == 0
            tt = lambda x: torch.tensor(np.asarray(x)[None], dtype=torch.float32)
            want, wdone, _ = rw(dict(robot=tt(ref["robot"][e]), robot_vel=None, object=tt(ref["object"][e])), tt(o.field("qpos")), tt(o.field("qvel")), tt(linkx))
            worst_q = max(worst_q, float(np.abs(obs[e, :35].cpu().numpy() - o.field("qpos")).max()))
            worst_r = max(worst_r, abs(float(reward[e]) - float(want[0])))
            assert float(done[e]) == float(wdone[0])
    assert (env.status() == 0).all()
    assert worst_q < 5e-4 and worst_r < 2e-3, (worst_q, worst_r)
    assert abs(float(env.view(__import__("myosuite_mjx_amd.capi", fromlist=["x"]).F_TIME)[0, 0]) - 3 * 0.01) < 1e-6


def test_trackenv_default_random_reference_runs(track):
    """The module-level default of mjx/myodm_v0.py:306-318 (a two-row, i.e. RANDOM-type reference; object_init 10 cm above the table): the
    object drops onto the table, nothing is flagged, metrics are finite, far-away random targets terminate the episode like :233-241."""
    from myosuite_mjx_amd import track as T
    env = T.TrackEnv(num_envs=64, seed=1, autoreset=True)
    obs = env.reset()
    assert abs(float(obs[0, 31]) - 0.1) < 1e-6
    import torch
    g = torch.Generator(device="cuda").manual_seed(0)
    n_done = 0
    for _ in range(30):
        a = torch.rand((64, env.act_dim), device="cuda", generator=g) * 2 - 1
        obs, reward, done, info = env.step(a)
        n_done += int(done.sum())
        assert torch.isfinite(obs).all() and torch.isfinite(reward).all()
    assert (env.status() == 0).all() and n_done > 0
    assert set(info["metrics"]) == {"pose", "object", "bonus", "penalty"}


def test_trackenv_step_is_one_launch_of_the_step_kernel_with_a_fused_epilogue(track):
    """MYO_TASK_TRACK: TrackEnv.step = myo_step(action, MYO_ACTMAP_CTRLRANGE, 5) and nothing else -- control scaling, reference
    lookup, reward, done, metrics and the masked reset happen inside that launch.  Checked here through its effects: the batch rows after ONE
    C-ABI call (no torch arithmetic in between) hold the new observation, the reward terms and, for envs that terminated, the reset state."""
    import torch
    from myosuite_mjx_amd import capi, track as T
    f = np.load(os.path.join(ROOT, "tests", "golden", "ref_motion.npz"))
    motion = {k.split("__in__")[1]: f[k] for k in f.files if k.startswith("track_MyoHand_airplane_fly1__in__")}
    B = 32
    env = T.TrackEnv(num_envs=B, reference=motion, seed=0, autoreset=True)
    env.reset()
    b = env.batch
    # carry the object of half of the envs 40 cm away from its reference: they must come out of the step done (:233-241), penalised and reset
    q = b.read(capi.F_QPOS)
    q[: B // 2, 29] += 0.4
    b.write(capi.F_QPOS, q)
    a = np.random.default_rng(0).uniform(-1, 1, (B, track.nu)).astype(np.float32)
    b.write(capi.F_ACTION, a)
    ptr, _, _ = b.field_ptr(capi.F_ACTION)
    b.step(ptr, capi.ACTMAP_CTRLRANGE, env.n_frames)                       # the ONE call of TrackEnv.step
    assert b.last_kernel_name().startswith("step_kernel_w<36,20,32,2,2,false,0,false,true")
    done, rew, met, obs = b.read(capi.F_DONE)[:, 0], b.read(capi.F_REWARD)[:, 0], b.read(capi.F_METRICS), b.read(capi.F_OBS)
    cr = np.asarray(track.actuator_ctrlrange, float)
    assert done[: B // 2].all() and not done[B // 2:].any()
    assert np.array_equal(met[:, 3], done) and np.allclose(rew, 0.0 * met[:, 0] + met[:, 1] + met[:, 2] - 2.0 * met[:, 3], atol=1e-6)
    # reset envs: observation = [init_qpos, 0], time / elapsed back to zero; live envs: observation = stepped state, controls = scaled action
    assert np.array_equal(obs[: B // 2, :35], np.tile(env.init_qpos, (B // 2, 1))) and not obs[: B // 2, 35:].any()
    assert not b.read(capi.F_TIME)[: B // 2].any() and not b.read(capi.F_ELAPSED)[: B // 2].any() and (b.read(capi.F_ELAPSED)[B // 2:] == 1).all()
    assert np.array_equal(obs[B // 2:, :35], b.read(capi.F_QPOS)[B // 2:]) and np.array_equal(obs[B // 2:, 35:], b.read(capi.F_QVEL)[B // 2:])
    assert np.allclose(b.read(capi.F_CTRL)[B // 2:], ((a[B // 2:] + 1) * (cr[:, 1] - cr[:, 0]).astype(np.float32) * 0.5 + cr[:, 0].astype(np.float32)), atol=1e-6)
    assert np.allclose(b.read(capi.F_TIME)[B // 2:, 0], 0.01, atol=1e-7) and (env.status() == 0).all()


def test_a_second_myodm_object_cup():
    """Another MyoDM object is another compiled asset, no new code: `cup` (13 hull geoms, the visual hull has 824 vertices, 960 candidate pairs)
    with the reference's MyoHand_cup_drink1 motion (tests/golden/ref_motion.npz).  (1) physics parity with the oracle on grasp frames of that
    motion, same criteria as for the airplane; (2) TrackEnv on it: the first env step against the oracle, then a rollout without flags."""
    import torch
    from myosuite_mjx_amd import capi, model as M, track as T
    from oracle.oracle import Oracle
    m = M.load_asset("myohand_object_cup")
    hm = capi.HipModel(m.blob(), 0)
    f = np.load(os.path.join(ROOT, "tests", "golden", "ref_motion.npz"))
    motion = {k.split("__in__")[1]: f[k] for k in f.files if k.startswith("track_MyoHand_cup_drink1__in__")}
    R, O = motion["robot"], motion["object"]
    rng = np.random.default_rng(3)
    frames = list(range(0, len(R), max(1, len(R) // 32)))[:32]
    q = np.zeros((len(frames), m.nq))
    for i, t in enumerate(frames):
        q[i, :29] = R[t] + rng.normal(0, 0.01, 29) * (np.arange(29) >= 6)
        q[i, 29:32] = O[t, :3]
        q[i, 32:35] = T.quat2euler(O[t, 3:])
    v = rng.normal(0, 0.2, (len(frames), m.nv))
    act = rng.uniform(0, 1, (len(frames), m.nu)); act[:, :6] = 0
    ctrl = rng.uniform(0, 1, (len(frames), m.nu)); ctrl[:, :6] = q[:, :6]
    g, r = _run(m, hm, q.astype(np.float32), v.astype(np.float32), act.astype(np.float32), ctrl.astype(np.float32), 5)
    # the cup's twelve convex parts sit inside its visual hull, so a grasping finger touches several of them at once: the deepest grasp frames
    # carry more than 64 contacts.  Round 2 truncated there (one contact per lane) and flagged the env; the TRK kernel now keeps contacts
    # 64 .. 127 in a second bank (one lane, two contacts; state of the second one in its overflow row): nothing is flagged and the contact
    # counts equal the oracle's (128 slots) on every frame
    over = r["ncon_max"] > 64
    assert over.sum() >= 2 and r["ncon_max"].max() <= 128, r["ncon_max"]
    assert (g["flags"] == 0).all(), g["flags"]
    assert r["ncon"].max() >= 8
    same = (g["diag"][:, 1] == r["ncon"]) & ((g["diag"][:, 4] >> 16) == r["ncon_sum"])
    eq = np.abs(g["qpos"] - r["qpos"]).max(1)
    assert same.sum() > 0.85 * len(eq) and np.percentile(eq, 90) < 2e-4 and eq[same].max() < 2e-3 and eq.max() < 1e-2, (same.sum(), len(eq), np.percentile(eq, 90), eq.max())
    assert (same & over).sum() >= 1 and eq[same & over].max() < 2e-3, (same & over).sum()       # ... including frames that need the second bank
    env = T.TrackEnv(num_envs=64, object_name="cup", reference=motion, seed=0, autoreset=True)
    obs = env.reset()
    assert obs.shape == (64, 70) and np.allclose(env.init_qpos[:29], motion["robot_init"], atol=1e-6)
    o = Oracle(m.blob())
    o.reset(); o.set_state(qpos=env.init_qpos.astype(np.float64), qvel=np.zeros(m.nv))
    a = np.random.default_rng(1).uniform(-1, 1, (64, m.nu)).astype(np.float32)
    cr = np.asarray(m.actuator_ctrlrange, float)
    obs, reward, done, info = env.step(a)
    o.set_state(ctrl=(a[0].astype(np.float64) + 1) * (cr[:, 1] - cr[:, 0]) * 0.5 + cr[:, 0])
    assert o.step(5) == 0
    assert np.abs(obs[0, :35].cpu().numpy() - o.field("qpos")).max() < 5e-4
    gen = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(40):
        obs, reward, done, info = env.step(torch.rand((64, env.act_dim), device="cuda", generator=gen) * 2 - 1)
        assert torch.isfinite(obs).all() and torch.isfinite(reward).all()
    assert (env.status() == 0).all()


def _objects_with_frames():
    f = np.load(os.path.join(ROOT, "tests", "golden", "myodm_grasp_frames.npz"))
    return sorted(k[:-7] for k in f.files if k.endswith("__robot"))


@pytest.mark.parametrize("obj", _objects_with_frames())
def test_more_myodm_objects_are_assets_only(obj):
    """VERDICT r2 missing-4: "more MyoDM objects = more compiled assets, no new code" -- on every object of the reference's OBJECTS tuple that has a motion
    file (49 of 50; tools/compile_models.py: hulls of 16 .. 2 970 vertices, 37 .. 77 collision geoms, up to 1 975 candidate pairs, up to 90 contacts).
    (1) physics parity with the oracle on twelve frames of one of the object's own motions (tests/golden/myodm_grasp_frames.npz: rows copied from the
    reference's envs/myo/myodm/data/<motion>.npz by tools/make_myodm_registry.py), same criteria as for the airplane; (2) the registered id steps through
    `myo.make` without a flag.  Table of the measured figures: profiles/r3_myodm_objects.json (tools/gpu_obj_sweep.py)."""
    import torch
    import myosuite_mjx_amd as myo
    from myosuite_mjx_amd import capi, model as M, track as T
    m = M.load_asset(f"myohand_object_{obj}")
    hm = capi.HipModel(m.blob(), 0)
    f = np.load(os.path.join(ROOT, "tests", "golden", "myodm_grasp_frames.npz"))
    R, O = f[obj + "__robot"], f[obj + "__object"]
    rng = np.random.default_rng(7)
    n = len(R)
    q = np.zeros((n, m.nq))
    q[:, :29] = R + rng.normal(0, 0.01, (n, 29)) * (np.arange(29) >= 6)
    q[:, 29:32] = O[:, :3]
    q[:, 32:35] = np.stack([T.quat2euler(o[3:]) for o in O])
    v = rng.normal(0, 0.2, (n, m.nv))
    act = rng.uniform(0, 1, (n, m.nu)); act[:, :6] = 0
    ctrl = rng.uniform(0, 1, (n, m.nu)); ctrl[:, :6] = q[:, :6]
    g, r = _run(m, hm, q.astype(np.float32), v.astype(np.float32), act.astype(np.float32), ctrl.astype(np.float32), 5)
    assert (g["flags"] == 0).all() and r["ncon_max"].max() <= 128, (g["flags"], r["ncon_max"])
    same = (g["diag"][:, 1] == r["ncon"]) & ((g["diag"][:, 4] >> 16) == r["ncon_sum"])
    eq = np.abs(g["qpos"] - r["qpos"]).max(1)
    assert r["ncon"].max() >= 3 and same.sum() >= 0.75 * n, (same.sum(), r["ncon"])
    # grasp frames of recorded motions interpenetrate by millimetres and rest polytope faces on polytope faces: some are ill-conditioned in
    # single precision whoever computes them (mug frame 0: the float32 BUILD of the oracle is off by 5.0e-2, HIP by 4.8e-2; hammer handle frames
    # 1e-3 .. 2e-3 vs 4e-3 .. 2.5e-2, tools/gpu_obj_probe.py).  So: where the float32 oracle itself stays within 1e-4 of the float64 one, HIP
    # must stay within 1e-3 (measured over the 49 objects: 9e-6 .. 1.4e-4, phone 6.2e-4, pyramidlarge 4.0e-4; frames with 90 contacts included);
    # everywhere it must stay bounded
    from oracle.oracle import Oracle
    o32 = Oracle(m.blob(), f32=True)
    e32 = np.zeros(n)
    for e in range(n):
        o32.reset(); o32.set_state(qpos=q[e].astype(np.float32), qvel=v[e].astype(np.float32), act=act[e].astype(np.float32), ctrl=ctrl[e].astype(np.float32))
        o32.step(5)
        e32[e] = np.abs(o32.field("qpos") - r["qpos"][e]).max()
    well = e32 < 1e-4
    assert well.sum() >= 0.5 * n and eq[well].max() < 1e-3 and eq.max() < 0.1, (well.sum(), eq[well].max(), eq.max())
    env = myo.make(f"MyoHand{obj.title()}Random-v0", num_envs=32, seed=0, autoreset=True)
    obs = env.reset()
    assert obs.shape == (32, 70)
    gen = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(30):
        obs, rew, term, trunc, info = env.step(torch.rand((32, env.act_dim), device="cuda", generator=gen) * 2 - 1)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert (env.status() == 0).all()
