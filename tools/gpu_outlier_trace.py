"""Developer probe: follows one state of tools/gpu_agreement_sweep.py substep by substep through both kernel instantiations and both oracle builds, and
prints where an instantiation leaves the float64 oracle (velocity / tendon-length / contact-count differences per substep).
usage: python tools/gpu_outlier_trace.py asset N seed env"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_mjx_amd import capi, model as M
from oracle.oracle import Oracle
asset, N, seed, e = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m = M.load_asset(asset)
rng = np.random.default_rng(seed)
lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
qpos = rng.uniform(lo, hi, (N, m.nq)).astype(np.float32)
qvel = rng.normal(0, 0.3, (N, m.nv)).astype(np.float32)
act = rng.uniform(0, 1, (N, m.nu)).astype(np.float32)
o = Oracle(m.blob()); o32 = Oracle(m.blob(), f32=True)
for oo in (o, o32):
    oo.reset(); oo.set_state(qpos=qpos[e].astype(float), qvel=qvel[e].astype(float), act=act[e].astype(float), ctrl=act[e].astype(float))
bs = []
for no_spec in ("0", "1"):
    os.environ["MYO_NO_SPEC"] = no_spec
    hm = capi.HipModel(m.blob(), 0)
    os.environ.pop("MYO_NO_SPEC", None)
    b = capi.HipBatch(hm, 1)
    for f, a in ((capi.F_QPOS, qpos[e:e+1]), (capi.F_QVEL, qvel[e:e+1]), (capi.F_ACT, act[e:e+1]), (capi.F_CTRL, act[e:e+1])):
        b.write(f, a)
    bs.append((hm, b))
print("env", e, "(columns: float32 oracle | size-specialised kernel | generic kernel, each against the float64 oracle)")
for k in range(10):
    o.step(1); o32.step(1)
    line = "  substep %d ncon %2d nefc %3d it %d | f32 oracle dv %.1e it %d |" % (k, o.ncon, o.nefc, o.solver_iter, np.abs(o32.field("qvel") - o.field("qvel")).max(), o32.solver_iter)
    for hm, b in bs:
        b.step(None, capi.ACTMAP_NONE, 1)
        dq = np.abs(b.read(capi.F_QPOS)[0] - o.field("qpos")); dv = np.abs(b.read(capi.F_QVEL)[0] - o.field("qvel"))
        tl = np.abs(b.read(capi.F_TENLEN)[0] - np.asarray(o.field("actuator_length"))[:m.nu]).max()
        line += " dq %.1e dv %.1e (dof %2d) tendon %.1e rows/contacts/iters %s |" % (dq.max(), dv.max(), dv.argmax(), tl, b.read(capi.F_DIAG)[0, :3].tolist())
    print(line)
