"""Model compiler + blob: sizes match SURVEY.md Appendix A; committed asset == fresh compile (container only)."""
import os

import numpy as np
import pytest

from conftest import REFERENCE, needs_reference
from myosuite_mjx_amd import blob


def test_blob_roundtrip():
    a = {"x": np.arange(12, dtype=np.float64).reshape(3, 4), "i": np.array([1, -2, 3], np.int32), "e": np.zeros((0, 3))}
    b = blob.unpack(blob.pack(a))
    assert set(b) == set(a)
    for k in a:
        assert b[k].shape == a[k].shape and np.array_equal(b[k], a[k])


def test_hand_sizes(hand):
    # SURVEY.md Appendix A, column H (myohand_pose.xml)
    assert (hand.nq, hand.nv, hand.nu, hand.na) == (23, 23, 39, 39)
    assert hand.nbody == 39 and hand.ntendon == 44 and hand.nsite == 333
    assert int(hand.sizes[11]) == 116                        # tree-sparse mass matrix non-zeros
    assert int((hand.jnt_limited != 0).sum()) == 23
    assert abs(hand.timestep - 0.002) < 1e-12
    ct = hand.geom_type[(hand.geom_contype != 0) | (hand.geom_conaffinity != 0)]
    assert (ct == 3).sum() == 22 and (ct == 4).sum() == 5    # 22 capsules + 5 ellipsoid pads (+ plane + pedestal)
    hs = dict(zip("nl nlevel nv nu ngt nseg ndl maxnnz nwg ncg npair maxkc".split(), hand.hip_sizes))
    assert hs["nl"] == 17 and hs["nlevel"] == 5 and hs["maxnnz"] <= 8 and hs["maxkc"] <= 8


def test_hand_muscle_defaults(hand):
    # <muscle> defaults (range .75-1.05, lmin .5, lmax 1.6, vmax 1.5, fpmax 1.3, fvmax 1.2, timeconst .01/.04)
    g = hand.actuator_gainprm
    assert np.allclose(g[:, [0, 1, 4, 5, 6, 7, 8]], [0.75, 1.05, 0.5, 1.6, 1.5, 1.3, 1.2])
    assert np.allclose(hand.actuator_dynprm[:, :2], [0.01, 0.04])
    i = hand.name2id("actuator", "ECRL")
    assert g[i, 2] == 337.3 and np.allclose(hand.actuator_lengthrange[i], [0.313191, 0.341072])   # myohand_assets.xml:501


def test_hand_inertia_bounds(hand):
    # compiler boundmass=0.001, boundinertia=1e-4, balanceinertia (myohand_assets.xml:11)
    mv = hand.body_weldid != 0
    assert (hand.body_mass[mv] >= 0.001 - 1e-15).all() and (hand.body_inertia[mv] >= 1e-4 - 1e-15).all()
    I = hand.body_inertia[mv]
    assert (I[:, 0] + I[:, 1] >= I[:, 2] - 1e-12).all()


@needs_reference
def test_asset_matches_fresh_compile(hand):
    from myosuite_mjx_amd import model as M
    fresh = M.from_mjcf(os.path.join(REFERENCE, "envs/myo/assets/hand/myohand_pose.xml"))
    assert set(fresh.arrays) == set(hand.arrays)
    for k, v in fresh.arrays.items():
        assert np.allclose(np.asarray(v, float), np.asarray(hand.arrays[k], float), rtol=0, atol=1e-12), k
    assert fresh.names == hand.names


def test_finger_sizes(finger):
    # SURVEY.md Appendix A, column F (myofinger_v0.xml): degrees -> radians, pulleys, tendon limits, compiler-derived force
    assert (finger.nq, finger.nv, finger.nu, finger.ntendon, finger.nsite) == (4, 4, 5, 5, 20)
    assert np.allclose(np.degrees(finger.jnt_range), [[-25, 25], [-25, 60], [0, 60], [0, 60]])
    assert (finger.tendon_limited == 1).all() and np.allclose(finger.tendon_range, [[0, 0.33]] * 5)
    assert (finger.wrap_type == 2).sum() == 2                     # two pulleys (finger_v0.xml:80,85)
    assert np.allclose(finger.jnt_solimp[:, :3], [0.95, 0.95, 0.1]) and np.allclose(finger.dof_damping, 0.5)
    assert (finger.actuator_gainprm[:, 2] == -1).all() and (finger.actuator_acc0 > 0).all()   # force = scale / acc0
    assert int(finger.hip_sizes[15]) == 7                         # the 7 plane pairs are provably out of reach


@needs_reference
def test_finger_asset_matches_fresh_compile(finger):
    from myosuite_mjx_amd import model as M
    fresh = M.from_mjcf(os.path.join(REFERENCE, "simhive/myo_sim/finger/myofinger_v0.xml"))
    for k, v in fresh.arrays.items():
        assert np.allclose(np.asarray(v, float), np.asarray(finger.arrays[k], float), rtol=0, atol=1e-12), k


@needs_reference
def test_unsupported_feature_raises(tmp_path):
    from myosuite_mjx_amd.mjcf import compile_mjcf
    p = tmp_path / "m.xml"
    p.write_text("<mujoco><worldbody><body><joint type='ball'/><geom size='0.1'/></body></worldbody></mujoco>")
    with pytest.raises(NotImplementedError):
        compile_mjcf(str(p))


def test_sim_scene_style_inline_model(tmp_path):
    """Same shape as the reference's physics/sim_scene_test.py:19-40 (tiny inline hinge chain): compiles and steps."""
    from myosuite_mjx_amd import blob as B
    from myosuite_mjx_amd.mjcf import compile_mjcf
    from myosuite_mjx_amd.setconst import set_constants
    from oracle.oracle import Oracle
    xml = """<mujoco><compiler angle="radian"/><worldbody>
      <body name="main" pos="0 0 1"><joint name="j0" type="hinge" axis="0 1 0" range="-1 1" limited="true"/>
        <geom type="capsule" size="0.05 0.2" pos="0 0 -0.2"/>
        <body pos="0 0 -0.4"><joint name="j1" type="hinge" axis="0 1 0"/><geom type="capsule" size="0.04 0.15" pos="0 0 -0.15"/>
          <body pos="0 0 -0.3"><joint name="j2" type="slide" axis="0 0 1"/><geom type="ellipsoid" size="0.05 0.04 0.03"/></body>
        </body></body></worldbody></mujoco>"""
    p = tmp_path / "chain.xml"
    p.write_text(xml)
    cm = compile_mjcf(str(p))
    set_constants(cm)
    assert list(cm.sizes[:2]) == [3, 3]
    o = Oracle(B.pack(cm.arrays))
    o.set_state(qpos=[0.3, -0.2, 0.01])
    for _ in range(10):
        assert o.step(1) == 0
    assert abs(o.time - 10 * 0.002) < 1e-12 and np.isfinite(o.field("qpos")).all()


@pytest.mark.parametrize("name", ["hand", "finger", "legs"])
def test_lowering_tables_are_consistent(name, request):
    """Host-side invariants of the tables the wave kernel relies on: ancestor-dof chains are root-first and end with the link's own
    dofs; the folded constant tendon lengths plus the remaining segments reproduce the numpy tendon lengths at qpos0; the
    size signature the size-specialised kernel instantiations check."""
    from myosuite_mjx_amd import setconst as sc
    m = request.getfixturevalue(name)
    A = m.arrays
    nl = int(A["hip_sizes"][0])
    par, da, dn = A["hip_link_parent"], A["hip_link_dofadr"], A["hip_link_dofnum"]
    adr, chain = A["hip_link_chain_adr"], A["hip_link_chain"]
    assert len(adr) == nl + 1 and adr[-1] == len(chain)
    for l in range(nl):
        dofs = [int(e) & 255 for e in chain[adr[l]:adr[l + 1]]]
        want, k = [], l
        while k >= 0:
            want = list(range(int(da[k]), int(da[k]) + int(dn[k]))) + want
            k = int(par[k])
        assert dofs == want, (l, dofs, want)
        free_flags = [(int(e) >> 12) & 1 for e in chain[adr[l]:adr[l + 1]]]
        assert all(f == int(A["hip_link_free"][int(A["hip_dof_link"][d])]) for f, d in zip(free_flags, dofs))
    # tendon lengths at qpos0: constant (folded) part + the kernel's segments == numpy tendon routine
    L, _ = sc.tendons(m, np.asarray(m.qpos0, float), want_jac=False) if "want_jac" in sc.tendons.__code__.co_varnames else sc.tendons(m, np.asarray(m.qpos0, float))
    gt = A["hip_gt_tendon"]
    assert (A["hip_gt_len0"] >= 0).all() and (A["hip_gt_len0"] <= np.asarray(L)[gt] + 1e-12).all()
    if name == "hand":
        assert tuple(int(x) for x in A["hip_sizes"][[2, 3, 0, 1, 7, 5, 9, 10]]) == (23, 39, 17, 5, 7, 116, 27, 289)     # Sizes<1> in csrc/myo_kernel_wave.h
    if name == "legs":
        assert tuple(int(x) for x in A["hip_sizes"][[2, 3, 0, 1, 7, 5, 9, 10]]) == (34, 80, 13, 6, 11, 100, 32, 45)     # Sizes<2>


def test_elbow_sizes(elbow):
    # myoelbow_1dof6muscles.xml: one hinge, six muscles + the "error" visual tendon; geoms are contype 1 / conaffinity 0 so nothing
    # collides and the bone meshes only contribute inertia (myoelbow_assets.xml:14-16)
    assert (elbow.nq, elbow.nv, elbow.nu, elbow.ntendon) == (1, 1, 6, 7) and elbow.pair_geom.shape[0] == 0
    assert np.allclose(elbow.jnt_range, [[0, 2.26893]]) and np.allclose(elbow.dof_damping, 0.5) and np.allclose(elbow.dof_armature, 0.01)
    assert abs(elbow.timestep - 0.002) < 1e-12 and (elbow.actuator_has_lengthrange == 1).all()
    assert (elbow.body_mass[-2:] > 1.0).all()                      # humerus / forearm masses from the bone meshes


@needs_reference
def test_elbow_asset_matches_fresh_compile(elbow):
    from myosuite_mjx_amd import model as M
    fresh = M.from_mjcf(os.path.join(REFERENCE, "envs/myo/assets/elbow/myoelbow_1dof6muscles.xml"))
    for k, v in fresh.arrays.items():
        assert np.allclose(np.asarray(v, float), np.asarray(elbow.arrays[k], float), rtol=0, atol=1e-12), k


def test_stateless_actuator_models(exo, motorfinger, hand):
    """<motor> actuators (motorfinger_v0.xml:10-16 on tendons, myoelbow_1dof6muscles_1dofexo_body.xml:174 on the joint): kind 1 =
    stateless affine, gain 1, no bias; the exo model's bone meshes can pair with the scene's floor / pedestal and are kept as
    bounding spheres whose pairs the lowering proves out of reach."""
    assert (hand.actuator_kind == 0).all() and hand.n_muscle == 39
    assert (motorfinger.nu, motorfinger.n_muscle) == (5, 0) and (motorfinger.actuator_kind == 1).all()
    assert (motorfinger.actuator_trntype == 1).all() and np.allclose(motorfinger.actuator_gear, [20, 5, 5, 10, 10])
    assert np.allclose(motorfinger.actuator_ctrlrange, [[-1, 0]] * 5) and (motorfinger.actuator_ctrllimited == 1).all()
    assert np.allclose(motorfinger.actuator_gainprm[:, 0], 1) and np.allclose(motorfinger.actuator_biasprm, 0)
    assert (exo.nu, exo.n_muscle, exo.nv) == (7, 6, 1) and exo.actuator_kind.tolist() == [1, 0, 0, 0, 0, 0, 0]
    assert int(exo.actuator_trntype[0]) == 0 and abs(exo.actuator_gear[0] - 8.5) < 1e-12 and int(exo.actuator_ctrllimited[0]) == 0
    assert exo.hip_act_obs.tolist() == [-1, 0, 1, 2, 3, 4, 5] and int(exo.hip_flags[3]) == 1 and int(exo.hip_flags[4]) == 6
    assert (exo.geom_type == 7).sum() > 10 and int(exo.hip_sizes[10]) == 0 and int(exo.hip_sizes[15]) == exo.pair_geom.shape[0]
    # action re-projection stored with the record (lowering): motorfinger has no muscles -> [-1, 1] onto the ctrlrange;
    # the exo motor rides along with muscles -> passed through
    assert np.allclose(motorfinger.hip_act[:, 5:7], [[0.5, -0.5]] * 5) and np.allclose(exo.hip_act[0, 5:7], [1.0, 0.0])


@needs_reference
@pytest.mark.parametrize("stem,rel", [("myoelbow_1dof6muscles_1dofexo", "envs/myo/assets/elbow/myoelbow_1dof6muscles_1dofexo.xml"),
                                      ("motorfinger_v0", "simhive/myo_sim/finger/motorfinger_v0.xml")])
def test_motor_assets_match_fresh_compile(stem, rel):
    from myosuite_mjx_amd import model as M
    fresh, asset = M.from_mjcf(os.path.join(REFERENCE, rel)), M.load_asset(stem)
    for k, v in fresh.arrays.items():
        assert np.allclose(np.asarray(v, float), np.asarray(asset.arrays[k], float), rtol=0, atol=1e-12), k


def test_hand_object_model():
    """myohand_hold.xml: MyoHand + a free ellipsoid object (second root link with a free joint, condim 1 geom mixing to 3 against the hand's
    geoms) over the scene's floor plane and pedestal cylinder, which stay in the pair table as world-fixed geoms."""
    from myosuite_mjx_amd import model as M
    m = M.load_asset("myohand_hold")
    assert (m.nq, m.nv, m.nu, m.n_muscle) == (30, 29, 39, 39) and int(m.jnt_type[-1]) == 0 and int(m.hip_flags[0]) == 1
    assert np.allclose(m.qpos0[-7:], [-0.235, -0.51, 1.45, 1, 0, 0, 0])
    og, cg = m.name2id("geom", "object"), list(m.hip_cg_geom)
    rows = [r for r in m.hip_pair_i if cg[r[0]] == og or cg[r[1]] == og]
    partners = {int(cg[r[0]]) if cg[r[1]] == og else int(cg[r[1]]) for r in rows}
    assert {0, 1} <= partners and len(partners) >= 28          # floor, pedestal and every collidable hand geom
    assert all(r[5] == 3 for r in rows)                          # condim max(1, 3)
    if os.path.isdir(REFERENCE):
        fresh = M.from_mjcf(os.path.join(REFERENCE, "envs/myo/assets/hand/myohand_hold.xml"))
        for k, v in fresh.arrays.items():
            assert np.allclose(np.asarray(v, float), np.asarray(m.arrays[k], float), rtol=0, atol=1e-12), k


@pytest.mark.parametrize("name", ["hand", "finger", "legs"])
def test_two_phase_kinematics_tables(name, request):
    """hip_kin_*: every non-free link has its 4 + 2 * dofnum vectors exactly once, in its own level, at distinct scratch offsets that fit the
    Hessian scratch the kernel borrows; kinds and indices address the right rows (lowering.py, wave kernel phase 1 / phase 2)."""
    m = request.getfixturevalue(name)
    A = m.arrays
    nl, nlevel, nv = (int(x) for x in A["hip_sizes"][:3])
    adr, vec, base = A["hip_kin_adr"], np.asarray(A["hip_kin_vec"]).reshape(-1, 2), A["hip_kin_base"]
    free, dofadr, dofnum, level_adr = A["hip_link_free"], A["hip_link_dofadr"], A["hip_link_dofnum"], A["hip_level_adr"]
    assert len(adr) == nlevel + 1 and adr[-1] == len(vec)
    nvt = 24 if nv <= 24 else 36
    assert int(A["hip_kin_size"][0]) <= nvt * (nvt + 1)
    seen = set()
    for L in range(nlevel):
        for w0, src in vec[adr[L]:adr[L + 1]]:
            l, kind, ix = int(w0) & 255, (int(w0) >> 8) & 3, int(w0) >> 16
            assert level_adr[L] <= l < level_adr[L + 1] and not free[l]
            assert base[l] <= src < base[l] + 3 * (4 + 2 * dofnum[l]) and (src - base[l]) % 3 == 0
            if kind in (2, 3):
                assert dofadr[l] <= ix < dofadr[l] + dofnum[l]
            elif kind == 0:
                assert 0 <= ix < 3
            assert (l, kind, ix if kind != 1 else 0) not in seen
            seen.add((l, kind, ix if kind != 1 else 0))
    assert len(seen) == sum(4 + 2 * int(dofnum[l]) for l in range(nl) if not free[l])
    assert int(A["hip_kin_size"][1]) == max([int(dofnum[l]) for l in range(nl) if not free[l]] + [0])


def test_specialised_instantiations_know_the_models_dof_trees(hand, legs):
    """The size-specialised kernels build in the dof tree (tree-sparse factorisation, SpecTree<1> / SpecTree<2> in myo_kernel_wave.h);
    myo_model_load falls back to the generic instantiation when a model's dof_parentid differs, so a silent mismatch would only cost speed:
    this keeps the header and the compiled config models in step."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myosuite_mjx_amd", "csrc", "myo_kernel_wave.h")).read()
    for spec, m in ((1, hand), (2, legs)):
        mm = re.search(r"struct SpecTree<%d> \{ static constexpr int nv = (\d+);\s*static constexpr int parent\[\d+\] = \{([^}]*)\}" % spec, src)
        assert mm, spec
        parent = [int(x) for x in mm.group(2).split(",")]
        assert int(mm.group(1)) == len(parent) == len(m.arrays["dof_parentid"])
        assert parent == [int(x) for x in m.arrays["dof_parentid"]]
