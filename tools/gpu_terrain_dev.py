"""Dev probe: HIP vs f64 oracle on the MyoLeg terrain model (height-field contacts)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from myosuite_mjx_amd import capi, model as M
from oracle.oracle import Oracle

m = M.load_asset("myolegs_terrain")
hm = capi.HipModel(m.blob(), 0)
rng = np.random.default_rng(0)
N = 48
f32 = np.float32
kq, kv = np.asarray(m.key_qpos).reshape(-1, m.nq)[2], np.asarray(m.key_qvel).reshape(-1, m.nv)[2]
q = np.tile(kq, (N, 1)); v = np.tile(kv, (N, 1)) * 0.3
q[:, 7:] += rng.normal(0, 0.05, (N, m.nq - 7))
q[:, 2] += rng.uniform(-0.03, 0.03, N)
q[:, :2] += rng.uniform(-0.5, 0.5, (N, 2))
v += rng.normal(0, 0.2, (N, m.nv))
act = rng.uniform(0, 1, (N, 80)); ctrl = rng.uniform(0, 1, (N, 80))
hf = np.zeros((N, 100, 100), f32)
for e in range(N):
    kind = e % 3
    if kind == 0: hf[e] = rng.uniform(0, 1, (100, 100)) * 0.08 - 0.02                     # rough (walk_v0.py:563-567)
    elif kind == 1: hf[e] = 0.03 + 0.05 * np.sin(np.linspace(0, 40, 100))[:, None] * np.ones((1, 100))   # ridges
    else: hf[e] = np.linspace(-0.1, 0.15, 100)[None, :] * np.ones((100, 1))                 # slope along x
q, v, act, ctrl = q.astype(f32), v.astype(f32), act.astype(f32), ctrl.astype(f32)
hfg = int(m.hfield_dims[2])
for nsub in (1, 10):
    b = capi.HipBatch(hm, N)
    for f, a in ((capi.F_QPOS, q), (capi.F_QVEL, v), (capi.F_ACT, act), (capi.F_CTRL, ctrl), (capi.F_HFIELD, hf.reshape(N, -1))):
        b.write(f, a)
    b.step(None, capi.ACTMAP_NONE, nsub)
    gq, gv, dg, fl = b.read(capi.F_QPOS), b.read(capi.F_QVEL), b.read(capi.F_DIAG), b.status()
    eq, ev, nc, nh = np.zeros(N), np.zeros(N), np.zeros(N, int), np.zeros(N, int)
    for e in range(N):
        o = Oracle(m.blob()); o.set_hfield(hf[e]); o.reset()
        o.set_state(qpos=q[e], qvel=v[e], act=act[e], ctrl=ctrl[e])
        rc = o.step(nsub)
        assert rc == 0, rc
        eq[e] = np.abs(gq[e] - o.field("qpos")).max(); ev[e] = np.abs(gv[e] - o.field("qvel")).max(); nc[e] = o.ncon
        nh[e] = sum(1 for c in o.contacts() if int(c[7]) == hfg)
    same = (fl == 0) & (dg[:, 1] == nc)
    print("nsub", nsub, "flags", np.bincount(fl, minlength=1)[:20].tolist(), "same ncon %.2f" % same.mean(), "ncon max", nc.max(), "hfield contacts mean %.1f max %d" % (nh.mean(), nh.max()),
          "| qpos err max %.3g med %.3g | qvel err max %.3g med %.3g" % (eq[same].max() if same.any() else -1, np.median(eq), ev[same].max() if same.any() else -1, np.median(ev)))
    print("   ncon hip", dg[:12, 1].tolist(), "oracle", nc[:12].tolist(), "err", np.round(eq[:12], 6).tolist())
