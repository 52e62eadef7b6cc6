"""`<option integrator="RK4">` (north_star: "Euler/RK4 integration"; no reference model selects it, `Model.with_integrator("RK4")` does):
mj_RungeKutta(4) restated in the oracle (runge_kutta4) and in the RK4 instantiations of the wave kernel.  Oracle side: the order of
convergence; GPU side: parity with the oracle on the hand (hinges, contacts) and on the legs (free root joint: quaternion stages)."""
import numpy as np
import pytest


def _hand_state(hand, rng, spread=0.6):
    lo, hi = hand.jnt_range[:, 0], hand.jnt_range[:, 1]
    return (0.5 * (lo + hi) + spread * 0.5 * (hi - lo) * rng.uniform(-1, 1, hand.nq))


def test_oracle_rk4_is_fourth_order_and_euler_first(hand):
    """Smooth dynamics (constraints off): error against a fine-step RK4 solution over 20 ms halves with the step for Euler and drops by
    >= 16x per halving for RK4 (measured 81x and 33x: the muscle curves are only piecewise smooth)."""
    from myosuite_mjx_amd import blob
    from oracle.oracle import Oracle
    rng = np.random.default_rng(0)
    q, v, a, c = _hand_state(hand, rng), rng.normal(0, 0.5, hand.nv), rng.uniform(0, 1, hand.nu), rng.uniform(0, 1, hand.nu)

    def run(model, dt, n):
        A = {k: np.array(x, copy=True) for k, x in model.arrays.items()}
        A["opt"][0] = dt
        o = Oracle(blob.pack(A))
        o.switches(1, 1, 1)
        o.reset(); o.set_state(qpos=q, qvel=v, act=a, ctrl=c)
        for _ in range(n):
            assert o.step(1) == 0
        assert abs(o.time - dt * n) < 1e-12
        return o.field("qpos").copy()
    ref = run(hand.with_integrator("RK4"), 1.25e-4, 160)
    e = {name: [np.abs(run(hand.with_integrator(name), dt, n) - ref).max() for dt, n in ((2e-3, 10), (1e-3, 20), (5e-4, 40))] for name in ("Euler", "RK4")}
    assert 1.7 < e["Euler"][0] / e["Euler"][1] < 2.4 and 1.7 < e["Euler"][1] / e["Euler"][2] < 2.4
    assert e["RK4"][0] / e["RK4"][1] > 16 and e["RK4"][1] / e["RK4"][2] > 16 and e["RK4"][2] < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("switches,nsub,tq,tv", [((1, 1, 1), 1, 5e-6, 2e-3), ((1, 1, 1), 10, 2e-5, 5e-3), ((0, 0, 0), 1, 2e-5, 1e-2), ((0, 0, 0), 10, 2e-4, 5e-2)])
def test_hand_rk4_matches_oracle(hand, switches, nsub, tq, tv):
    from myosuite_mjx_amd import capi
    from oracle.oracle import Oracle
    m = hand.with_integrator("RK4")
    hm = capi.HipModel(m.blob(), 0)
    hm.set_switch(*switches)
    o = Oracle(m.blob())
    o.switches(*switches)
    rng = np.random.default_rng(4)
    N = 128
    q = np.stack([_hand_state(hand, rng, 1.0 if switches == (0, 0, 0) else 0.6) for _ in range(N)]).astype(np.float32)
    v = rng.normal(0, 0.5, (N, hand.nv)).astype(np.float32)
    a, c = rng.uniform(0, 1, (N, hand.nu)).astype(np.float32), rng.uniform(0, 1, (N, hand.nu)).astype(np.float32)
    b = capi.HipBatch(hm, N)
    for f, x in ((capi.F_QPOS, q), (capi.F_QVEL, v), (capi.F_ACT, a), (capi.F_CTRL, c)):
        b.write(f, x)
    b.step(None, capi.ACTMAP_NONE, nsub)
    assert b.last_kernel_name().endswith("false,true>")
    gq, gv, ga, gt = b.read(capi.F_QPOS), b.read(capi.F_QVEL), b.read(capi.F_ACT), b.read(capi.F_TIME)
    assert (b.status() == 0).all() and np.allclose(gt, nsub * 0.002, atol=1e-6)
    eq, ev, ea = [], [], []
    for e in range(N):
        o.reset(); o.set_state(qpos=q[e], qvel=v[e], act=a[e], ctrl=c[e])
        assert o.step(nsub) == 0
        eq.append(np.abs(gq[e] - o.field("qpos")).max()); ev.append(np.abs(gv[e] - o.field("qvel")).max()); ea.append(np.abs(ga[e] - o.field("act")).max())
    eq, ev = np.array(eq), np.array(ev)
    assert max(ea) < 2e-6
    if switches == (1, 1, 1):
        assert eq.max() < tq and ev.max() < tv, (eq.max(), ev.max())
    else:      # contacts: the four stage solves each see their own contact set; strict bound on the 95th percentile, loose on all
        assert np.percentile(eq, 95) < tq and np.percentile(ev, 95) < tv and eq.max() < 5e-3, (np.percentile(eq, 95), eq.max())


@pytest.mark.gpu
def test_legs_rk4_free_joint(legs):
    """Free root joint under RK4: every stage restarts the quaternion from X0 and rotates it by h a omega (body frame)."""
    from myosuite_mjx_amd import capi
    from oracle.oracle import Oracle
    m = legs.with_integrator("RK4")
    hm = capi.HipModel(m.blob(), 0)
    o = Oracle(m.blob())
    rng = np.random.default_rng(6)
    N = 48
    kq, kv = np.asarray(legs.key_qpos).reshape(-1, legs.nq)[2], np.asarray(legs.key_qvel).reshape(-1, legs.nv)[2]
    q = np.tile(kq, (N, 1)); q[:, 7:] += rng.normal(0, 0.03, (N, legs.nq - 7)); q[:, 2] += 0.3          # in the air: equalities + limits only
    quat = rng.normal(0, 1, (N, 4)); q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    v = np.tile(kv, (N, 1)) + rng.normal(0, 0.5, (N, legs.nv)); v[:, 3:6] = rng.normal(0, 3.0, (N, 3))
    a, c = rng.uniform(0, 1, (N, legs.nu)), rng.uniform(0, 1, (N, legs.nu))
    b = capi.HipBatch(hm, N)
    for f, x in ((capi.F_QPOS, q), (capi.F_QVEL, v), (capi.F_ACT, a), (capi.F_CTRL, c)):
        b.write(f, x.astype(np.float32))
    b.step(None, capi.ACTMAP_NONE, 5)
    gq, gv = b.read(capi.F_QPOS), b.read(capi.F_QVEL)
    assert (b.status() == 0).all()
    eq = ev = 0.0
    for e in range(N):
        o.reset(); o.set_state(qpos=q[e].astype(np.float32), qvel=v[e].astype(np.float32), act=a[e].astype(np.float32), ctrl=c[e].astype(np.float32))
        assert o.step(5) == 0
        eq = max(eq, np.abs(gq[e] - o.field("qpos")).max()); ev = max(ev, np.abs(gv[e] - o.field("qvel")).max())
    assert eq < 5e-5 and ev < 2e-2, (eq, ev)
    assert np.abs(np.linalg.norm(gq[:, 3:7], axis=1) - 1).max() < 1e-6
