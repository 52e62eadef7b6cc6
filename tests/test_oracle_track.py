"""Oracle-side checks of the physics MyoDM's TrackEnv model adds (myohand_object.xml + object_sim/airplane):
joint friction loss, condim-4 (torsional) pyramids, box and convex-hull mesh contacts, position actuators on the 6-dof arm base.
No reference goldens exist for dynamics: analytic expectations only -- parity unpinned."""
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def track():
    from myosuite_mjx_amd import model as M
    return M.load_asset("myohand_object_airplane")


def _fresh(track, **edits):
    from myosuite_mjx_amd import blob
    from oracle.oracle import Oracle
    A = {k: np.array(v, copy=True) for k, v in track.arrays.items()}
    for k, f in edits.items():
        f(A[k])
    return Oracle(blob.pack(A))


def test_model_dimensions(track):
    """SURVEY Appendix A, TrackEnv column: 6 arm + 23 hand + 6 object dofs, 39 muscles + 6 position actuators (declared first,
    myohand_tabletop.xml:12-17), object joints with frictionloss 0.001 / armature 0.001 (object_sim/common.xml:12), condim 4 on the
    8 contact hulls (:17), 6 colliding table boxes."""
    m = track
    assert (m.nq, m.nv, m.nu) == (35, 35, 45)
    assert m.names["joint"][:6] == ["ARTx", "ARTy", "ARTz", "ARRx", "ARRy", "ARRz"] and m.names["joint"][-6:] == ["OBJTx", "OBJTy", "OBJTz", "OBJRx", "OBJRy", "OBJRz"]
    assert m.names["actuator"][:6] == ["A_ARTx", "A_ARTy", "A_ARTz", "A_ARRx", "A_ARRy", "A_ARRz"] and (np.asarray(m.actuator_kind)[:6] == 1).all()
    assert np.allclose(m.actuator_gainprm[:6, 0], [175, 175, 175, 150, 150, 150]) and np.allclose(m.actuator_biasprm[:6, 1], [-175, -175, -175, -150, -150, -150])
    assert np.allclose(m.dof_frictionloss[-6:], 0.001) and np.allclose(m.dof_frictionloss[:-6], 0) and np.allclose(m.dof_armature[-6:], 0.001)
    assert np.allclose(m.dof_damping[:6], 20)
    gt = np.asarray(m.geom_type)
    assert (gt == 6).sum() == 6 and (gt == 7).sum() == 9 and (gt == 0).sum() == 2
    hulls = [i for i, n in enumerate(m.names["geom"]) if n.startswith("airplane_contact")]
    assert len(hulls) == 8 and (np.asarray(m.geom_condim)[hulls] == 4).all() and np.allclose(np.asarray(m.geom_friction)[hulls], [1, 0.5, 0.01])
    assert int(np.asarray(m.geom_meshnum).sum()) == len(m.mesh_vert) and 10 <= min(np.asarray(m.geom_meshnum)[hulls]) and max(np.asarray(m.geom_meshnum)[hulls]) <= 64


def test_friction_loss_decelerates_at_f_over_m(track):
    """A dof with frictionloss f and nothing else acting on it: Coulomb-like deceleration f / M_dd until it stops, then it stays put."""
    o = _fresh(track)
    o.switches(1, 1, 1)
    o.lib.myoo_set_switch(o.m, 1, 1, 1)
    m = track
    # no gravity: lift the scene's gravity through the model copy (opt[3])
    o = _fresh(track, opt=lambda a: a.__setitem__(slice(1, 4), 0.0))
    o.switches(1, 1, 1)
    o.reset()
    v0 = 0.02
    qv = np.zeros(m.nv); qv[-6] = v0                     # OBJTx
    o.set_state(qvel=qv, ctrl=np.zeros(m.nu))
    o.forward()
    Mdd = o.full_m(m.nv)[-6, -6]
    acc = 0.001 / Mdd
    n = 200
    for _ in range(n):
        assert o.step(1) == 0
    t = n * m.timestep
    assert abs(o.field("qvel")[-6] - (v0 - acc * t)) < 0.02 * acc * t          # linear decay at f / M
    for _ in range(int(1.2 * (v0 / acc - t) / m.timestep)):
        o.step(1)
    x_stop = o.field("qpos")[-6]
    assert abs(o.field("qvel")[-6]) < 1e-5                                      # stopped ...
    for _ in range(200):
        o.step(1)
    assert abs(o.field("qpos")[-6] - x_stop) < 1e-6                             # ... and stays (quadratic zone holds it)


def test_object_rests_on_the_table_with_condim4_rows(track):
    """Hull meshes against the table-top box: the airplane settles on the table (its lowest hull vertex at z = 0 up to the contact
    softness); every contact of the settled state is object-vs-table and carries 2 (condim - 1) = 6 pyramid rows; the last pair of rows
    is J_normal +- 0.5 * J_spin (0.5 = the torsional coefficient) with J_spin = the relative angular velocity about the contact normal (here: +-1 on OBJRz for a
    vertical normal, 0 on the object's translations)."""
    m = track
    o = _fresh(track)
    o.reset()
    o.set_state(ctrl=np.zeros(m.nu))
    for _ in range(150):
        assert o.step(5) == 0
    o.forward()
    z = o.field("qpos")[-4]
    hulls = [g for g in range(m.ngeom) if int(m.geom_meshnum[g]) > 0]                # hull vertices are stored about each hull's centre (= geom_pos)
    hull_z = min(float(m.geom_pos[g][2] + np.asarray(m.mesh_vert)[int(m.geom_meshadr[g]):int(m.geom_meshadr[g]) + int(m.geom_meshnum[g]), 2].min()) for g in hulls)
    assert abs((0.035 + z) + hull_z) < 3e-3                                      # lowest hull point on the table top (z = 0)
    cons = o.contacts()
    table = {i for i, t in enumerate(np.asarray(m.geom_type)) if t == 6}
    assert len(cons) >= 1 and all(int(c[7]) in table and m.names["geom"][int(c[8])].startswith("airplane") for c in cons)
    nfric = int((np.asarray(m.dof_frictionloss) > 0).sum())
    nlim = o.nefc - nfric - 6 * len(cons)
    assert 0 <= nlim <= 10                                                        # friction rows + a few joint limits + 6 rows per contact
    J = o.field("efc_J").reshape(o.nefc, m.nv)[-6:]                               # rows of the last contact
    c = cons[-1]
    assert c[6] > 0.999                                                           # normal = +z (table top)
    Jn = 0.5 * (J[0] + J[1])
    Jspin = (J[4] - J[5]) / (2 * 0.5)              # geom friction = (sliding 1, torsional 0.5, rolling 0.01): condim 4 adds the torsional pair
    assert np.allclose(0.5 * (J[4] + J[5]), Jn, atol=1e-12)
    assert abs(abs(Jspin[-1]) - 1) < 2e-2 and np.abs(Jspin[-6:-3]).max() < 1e-12 and np.abs(Jspin[:29]).max() < 1e-12
    assert np.allclose((J[0] - J[1]) / 2.0, (J[0] - J[1]) / 2.0) and abs(Jn[-4]) > 0.99          # normal row moves OBJTz


def test_position_actuators_hold_the_arm(track):
    """kp (ctrl - q) on the six arm-base joints (myohand_tabletop.xml:12-17): with ctrl = 0.05 on ARTx the base converges there
    (damping 20 on the joint, gain 175)."""
    m = track
    o = _fresh(track)
    o.reset()
    c = np.zeros(m.nu); c[0] = 0.05
    o.set_state(ctrl=c)
    for _ in range(400):
        assert o.step(5) == 0
    assert abs(o.field("qpos")[0] - 0.05) < 5e-3


def test_resting_height_of_the_objects_against_every_motion_file():
    """The only contact-statics golden the reference holds: each motion file starts with the object lying on the table,
    at a height MuJoCo settled it to (`object_init` z; tests/golden/myodm_motion_inits.json holds the initial postures of all 95 files, copied
    from envs/myo/myodm/data/*.npz).  The oracle, started from that pose, must keep the object there: measured 0.5-0.7 mm lower for both
    compiled objects, on all six of their motion files (airplane -0.0098 -> -0.0104, cup +0.0161 -> +0.0156), i.e. contact softness level."""
    import json
    from myosuite_mjx_amd import model as M
    from myosuite_mjx_amd.track import quat2euler
    from oracle.oracle import Oracle
    inits = json.load(open(os.path.join(ROOT, "tests", "golden", "myodm_motion_inits.json")))
    assert len(inits) == 95
    seen = 0
    for obj in ("airplane", "cup"):
        m = M.load_asset(f"myohand_object_{obj}")
        o = Oracle(m.blob())
        for stem, v in inits.items():
            if f"_{obj}_" not in stem:
                continue
            oi = np.array(v["object_init"])
            q = np.array(m.qpos0, float)
            q[:29], q[29:32], q[32:35] = v["robot_init"], oi[:3], quat2euler(oi[3:])
            ctrl = np.zeros(m.nu); ctrl[:6] = q[:6]                 # the arm base holds its pose (position actuators), muscles relaxed
            o.reset(); o.set_state(qpos=q, qvel=np.zeros(m.nv), ctrl=ctrl)
            assert o.step(150) == 0
            z = o.field("qpos")[31]
            assert o.ncon >= 3 and -1.0e-3 < z - oi[2] < -2.0e-4, (stem, oi[2], z)       # rests on the table, 0.2-1 mm below MuJoCo's height
            v = np.abs(o.field("qvel")[29:35])
            assert v[:3].max() < 1e-2 and v[3:].max() < 0.3, stem                        # ... and (but for a last bit of rocking) at rest
            seen += 1
    assert seen == 6
