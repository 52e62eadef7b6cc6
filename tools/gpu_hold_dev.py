"""Dev probe: HIP vs f64 oracle on the MyoHand + free object model (myohand_hold.xml), object resting in / dropped onto the open hand."""
import sys
import numpy as np
sys.path.insert(0, ".")
from myosuite_mjx_amd import capi, model as M
from oracle.oracle import Oracle

m = M.load_asset("myohand_hold")
hm = capi.HipModel(m.blob(), 0)
o = Oracle(m.blob())
rng = np.random.default_rng(0)
N = 128
f32 = np.float32
q = np.tile(m.qpos0, (N, 1))
q[:, :23] = 0
q[:, 0] = -1.5
q[:, :23] += rng.normal(0, 0.15, (N, 23))
lo, hi = m.jnt_range[:23, 0], m.jnt_range[:23, 1]
q[:, :23] = np.clip(q[:, :23], lo + 0.01, hi - 0.01)
q[:, 23:26] += rng.normal(0, 0.003, (N, 3)) + np.array([0, 0, 0.003])
quat = rng.normal(0, 1, (N, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
q[:, 26:30] = quat
v = rng.normal(0, 0.5, (N, m.nv)); v[:, 23:26] *= 0.2
act = rng.uniform(0, 1, (N, 39)); ctrl = rng.uniform(0, 1, (N, 39))
q, v, act, ctrl = q.astype(f32), v.astype(f32), act.astype(f32), ctrl.astype(f32)
for nsub in (1, 10):
    b = capi.HipBatch(hm, N)
    for f, a in ((capi.F_QPOS, q), (capi.F_QVEL, v), (capi.F_ACT, act), (capi.F_CTRL, ctrl)):
        b.write(f, a)
    b.step(None, capi.ACTMAP_NONE, nsub)
    gq, gv, dg, fl = b.read(capi.F_QPOS), b.read(capi.F_QVEL), b.read(capi.F_DIAG), b.status()
    eq, ev, nc = np.zeros(N), np.zeros(N), np.zeros(N, int)
    for e in range(N):
        o.reset()
        o.set_state(qpos=q[e], qvel=v[e], act=act[e], ctrl=ctrl[e])
        rc = o.step(nsub)
        assert rc == 0, rc
        eq[e] = np.abs(gq[e] - o.field("qpos")).max(); ev[e] = np.abs(gv[e] - o.field("qvel")).max(); nc[e] = o.ncon
    same = (fl == 0) & (dg[:, 1] == nc)
    print("nsub", nsub, "flags", np.bincount(fl, minlength=1)[:8], "same ncon", same.mean(), "ncon max", nc.max(), "mean", nc.mean(),
          "obj contacts?", "| qpos err max %.3g med %.3g | qvel err max %.3g med %.3g" % (eq[same].max(), np.median(eq), ev[same].max(), np.median(ev)))
    worst = np.argsort(-eq)[:5]
    print("  worst envs", worst, eq[worst], ev[worst], nc[worst], dg[worst, 1])
    for e in worst[:3] if nsub == 1 else []:
        o.reset(); o.set_state(qpos=q[e], qvel=v[e], act=act[e], ctrl=ctrl[e]); o.step(1)
        dv = gv[e] - o.field("qvel")
        print("   env", e, "worst dofs", np.argsort(-np.abs(dv))[:6], np.sort(-np.abs(dv))[:6], "iters hip/oracle", dg[e, 2], o.solver_iter)
        print("   nefc hip/oracle", dg[e, 0], o.nefc)
        if e == 72:
            ga = b.read(capi.F_QACC)[e]
            print("   qacc hip   ", np.round(ga, 1).tolist())
            print("   qacc oracle", np.round(o.field("qacc"), 1).tolist())
        print("   contacts", [np.round(c, 4).tolist() for c in o.contacts()])
