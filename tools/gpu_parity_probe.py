"""Developer probe (GPU box): worst env of a HIP-vs-oracle comparison, followed substep by substep."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from myosuite_mjx_amd import capi, model as M
from oracle.oracle import Oracle
from test_gpu_parity import _states, _run_pair

m = M.load_asset("myohand_pose")
hm = capi.HipModel(m.blob(), 0)
o = Oracle(m.blob()); o32 = Oracle(m.blob(), f32=True)
qpos, qvel, act, ctrl = _states(m, 1024, 13)
g, r = _run_pair(m, hm, o, qpos, qvel, act, ctrl, 10, (0, 0, 0))
eq = np.abs(g["qpos"] - r["qpos"]).max(1)
worst = np.argsort(eq)[::-1][:4]
print("worst envs", worst, eq[worst], "ncon hip", g["diag"][worst, 1], "oracle", r["ncon"][worst])
for e in worst[:2]:
    b = capi.HipBatch(hm, 1)
    for f, a in ((capi.F_QPOS, qpos[e:e+1]), (capi.F_QVEL, qvel[e:e+1]), (capi.F_ACT, act[e:e+1]), (capi.F_CTRL, ctrl[e:e+1])):
        b.write(f, a)
    for oo in (o, o32):
        oo.reset(); oo.set_state(qpos=qpos[e], qvel=qvel[e], act=act[e], ctrl=ctrl[e])
    for s in range(10):
        b.step(None, capi.ACTMAP_NONE, 1)
        o.step(1); o32.step(1)
        d = b.read(capi.F_DIAG)[0]
        gq, gv, ga = b.read(capi.F_QPOS)[0], b.read(capi.F_QVEL)[0], b.read(capi.F_QACC)[0]
        print(f"env {e} substep {s}: hip ncon {d[1]} nefc {d[0]} it {d[2]} | f64 ncon {o.ncon} nefc {o.nefc} it {o.solver_iter} | f32 ncon {o32.ncon} | dq hip {np.abs(gq-o.field('qpos')).max():.2e} f32 {np.abs(o32.field('qpos')-o.field('qpos')).max():.2e}"
              f" | dqvel hip {np.abs(gv-o.field('qvel')).max():.2e} f32 {np.abs(o32.field('qvel')-o.field('qvel')).max():.2e} | dqacc hip {np.abs(ga-o.field('qacc')).max():.2e} |qacc| {np.abs(o.field('qacc')).max():.1f}")
for w in (1, 2, 4):
    rate, ncu = capi.probe_valu(w)
    print(f"valu probe: {w} waves/SIMD: {rate/1e9:.1f} G wave-inst/s on {ncu} CUs = {ncu*4*2.4e9/rate:.2f} cycles/inst/SIMD at 2.4 GHz")
