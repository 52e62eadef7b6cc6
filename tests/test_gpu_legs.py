"""GPU parity tests (pytest -m gpu) for the MyoLeg model (BASELINE.json config 4: free-floating root, 80 muscles, 14 knee
coupling equalities, plane / capsule / ellipsoid foot contacts): the HIP wave kernel, called through the C ABI, against the
f64 oracle on seeded states around the model's keyframes.

Tolerances (float32 HIP vs float64 oracle after identical float32 inputs):
  one substep  : qpos 5e-6, qvel 5e-3 ; joint-limit case 2e-5 / 2e-2 (0.3 rad of joint noise drives joints far past their
                 limits: |qacc| reaches 1e5 rad/s^2, and qvel = h * qacc carries its 1e-5 relative float32 error)
  one env step : qpos 5e-5, qvel 2e-2 ; joint-limit case 1e-4 / 5e-2
The float32 build of the oracle itself differs from the float64 build by more than this on the same states
(tests/test_oracle.py::test_leg_f32_sensitivity documents the figure), i.e. the tolerances are at the float32 noise floor."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def leghip(legs):
    from myosuite_mjx_amd import capi
    return capi.HipModel(legs.blob(), 0)


def leg_states(m, N, seed, key=2, dz=0.0, jitter=0.05, vel_sigma=0.3):
    rng = np.random.default_rng(seed)
    q = np.tile(np.asarray(m.key_qpos).reshape(-1, m.nq)[key], (N, 1))
    q[:, 7:] += rng.normal(0, jitter, (N, m.nq - 7))
    q[:, 2] += dz + rng.uniform(-0.02, 0.02, N)
    quat = q[:, 3:7] + rng.normal(0, 0.03, (N, 4))
    q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    f32 = np.float32
    return (q.astype(f32), rng.normal(0, vel_sigma, (N, m.nv)).astype(f32), rng.uniform(0, 1, (N, m.nu)).astype(f32),
            rng.uniform(0, 1, (N, m.nu)).astype(f32))


def _run_pair(m, hm, oracle, st, nsub, switches):
    from myosuite_mjx_amd import capi
    qpos, qvel, act, ctrl = st
    N = qpos.shape[0]
    hm.set_switch(*switches)
    oracle.switches(*switches)
    b = capi.HipBatch(hm, N)
    for f, a in ((capi.F_QPOS, qpos), (capi.F_QVEL, qvel), (capi.F_ACT, act), (capi.F_CTRL, ctrl)):
        b.write(f, a)
    b.step(None, capi.ACTMAP_NONE, nsub)
    g = {k: b.read(f) for k, f in dict(qpos=capi.F_QPOS, qvel=capi.F_QVEL, act=capi.F_ACT, qacc=capi.F_QACC, tenlen=capi.F_TENLEN,
                                       force=capi.F_ACTFORCE, diag=capi.F_DIAG).items()}
    g["flags"] = b.status()
    r = {k: np.zeros_like(g[k], dtype=np.float64) for k in ("qpos", "qvel", "act", "qacc", "tenlen", "force")}
    r["ncon"] = np.zeros(N, int)
    r["nefc"] = np.zeros(N, int)
    for e in range(N):
        oracle.reset()
        oracle.set_state(qpos=qpos[e], qvel=qvel[e], act=act[e], ctrl=ctrl[e], warm=np.zeros(m.nv), time=0)
        oracle.step(nsub)
        for k, f in (("qpos", "qpos"), ("qvel", "qvel"), ("act", "act"), ("qacc", "qacc"), ("tenlen", "actuator_length"), ("force", "actuator_force")):
            r[k][e] = oracle.field(f)
        r["ncon"][e] = oracle.ncon
        r["nefc"][e] = oracle.nefc
    hm.set_switch(0, 0, 0)
    oracle.switches(0, 0, 0)
    return g, r


def _check(g, r, tq, tv, min_frac=0.97, tl=2e-5):
    assert (g["flags"] == 0).all()
    ok = g["diag"][:, 1] == r["ncon"]
    assert ok.mean() >= min_frac, f"contact-count mismatches: {(~ok).sum()}"
    assert np.abs(g["qpos"] - r["qpos"])[ok].max() < tq
    assert np.abs(g["qvel"] - r["qvel"])[ok].max() < tv
    assert np.abs(g["act"] - r["act"])[ok].max() < 1e-6
    assert np.abs(g["tenlen"] - r["tenlen"])[ok].max() < tl       # lengths of the last substep: moment arm (<= 0.1 m/rad) x qpos error
    return ok


@pytest.mark.parametrize("nsub,tq,tv", [(1, 2e-6, 2e-3), (10, 2e-5, 5e-3)])
def test_leg_smooth_and_equalities(legs, leghip, legoracle64, nsub, tq, tv):
    """Free-joint kinematics / RNE, 80 wrapped muscles and the 14 knee-coupling equality rows (contacts and limits off)."""
    g, r = _run_pair(legs, leghip, legoracle64, leg_states(legs, 64, nsub, dz=0.5), nsub, (1, 1, 1))
    _check(g, r, tq, tv, 1.0)
    assert (g["diag"][:, 0] == 14).all() and (r["nefc"] == 14).all()


@pytest.mark.parametrize("nsub,tq,tv", [(1, 2e-5, 2e-2), (10, 1e-4, 5e-2)])
def test_leg_joint_limits(legs, leghip, legoracle64, nsub, tq, tv):
    g, r = _run_pair(legs, leghip, legoracle64, leg_states(legs, 64, 10 + nsub, dz=0.5, jitter=0.3), nsub, (1, 0, 1))
    # tendon lengths: up to 11 dofs per tendon x moment arm (<= 0.1 m/rad) x the qpos bound above; the worst of the 64 x 80 values sits at 4-5.3e-5
    # after ten substeps and moves by 10 % with any change of float32 summation order (round 3: 4.6e-5 with the library sincos or the LDS
    # mass-matrix product, 5.3e-5 with both replaced -- the two changes are each below 5e-5 alone), i.e. it is the amplified round-off of the
    # limit-row active set, not a bias
    _check(g, r, tq, tv, 1.0, tl=5e-5 if nsub == 1 else 1e-4)
    assert (g["diag"][:, 0] == r["nefc"]).all()
    assert r["nefc"].max() > 20


@pytest.mark.parametrize("nsub,tq,tv", [(1, 5e-6, 5e-3), (10, 5e-5, 2e-2)])
def test_leg_ground_contacts(legs, leghip, legoracle64, nsub, tq, tv):
    """Feet on the floor: plane-capsule (two contacts per pair), plane-ellipsoid, capsule-capsule and the explicit condim-1 pairs."""
    g, r = _run_pair(legs, leghip, legoracle64, leg_states(legs, 96, 20 + nsub, dz=-0.04), nsub, (0, 0, 0))
    ok = _check(g, r, tq, tv)
    assert r["ncon"].max() >= 8 and (r["ncon"] > 0).mean() > 0.3
    assert (g["diag"][:, 0] == r["nefc"])[ok].mean() > 0.9     # a contact exactly at its margin can flip activity in float32


def test_leg_deep_penetration_stays_close(legs, leghip, legoracle64):
    """Keyframe 0 sunk 5 cm into the floor with 0.2 rad joint noise: up to 21 contacts and 108 rows.  Extreme states amplify
    float32 round-off (the float32 oracle build itself is off by up to 7e-2 here); the HIP path must stay finite, unflagged
    and close in the bulk."""
    g, r = _run_pair(legs, leghip, legoracle64, leg_states(legs, 64, 7, key=0, dz=-0.05, jitter=0.2), 10, (0, 0, 0))
    assert (g["flags"] == 0).all()
    ok = g["diag"][:, 1] == r["ncon"]
    assert ok.mean() > 0.9
    err = np.abs(g["qpos"] - r["qpos"])[ok].max(axis=1)
    assert np.quantile(err, 0.9) < 2e-4 and err.max() < 5e-3
    assert r["ncon"].max() >= 15
