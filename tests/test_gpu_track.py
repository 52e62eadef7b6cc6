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
             xpos=np.zeros((N, m.nbody, 3)))
    for e in range(N):
        o.reset()
        o.set_state(qpos=qpos[e], qvel=qvel[e], act=act[e], ctrl=ctrl[e])
        for s in range(nsub):
            o.step(1)
            r["ncon_sum"][e] += o.ncon
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
    from myosuite_mjx_amd import track as T
    q, v, a, c = _motion_states(track, [0, 10, 20, 30, 40], 3)
    g, r = _run(track, hm, q, v, a, c, 5, (1, 0, 1))
    xpos, xmat = T.body_frames(track, g["linkx"])
    for bname in ("airplane", "lunate", "distal_thumb"):
        bid = track.name2id("body", bname)
        assert np.abs(xpos[:, bid] - r["xpos"][:, bid]).max() < 2e-5
