"""A/B of the polynomial asin / sin / cos of the inside-wrap solve against the library functions (test_finger_parity's velocity
bound moved from 2e-2 to 3e-2 when they came in).  Build the variant once:
    MYO_HIPCC_EXTRA=-DMYO_EXACT_TRIG=1 python -c "from myosuite_mjx_amd import capi; capi.LIB_PATH = capi.LIB_PATH.replace('libmyo_hip.so', 'libmyo_hip_exacttrig.so'); capi.build_library(force=True)"
then on the GPU box:  python tools/gpu_trig_ab.py   -> the finger and hand parity figures of both libraries side by side."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from myosuite_mjx_amd import capi, model as M
from oracle.oracle import Oracle
out = {}
for name, nsub in (("myofinger_v0", 10), ("myohand_pose", 10)):
    m = M.load_asset(name); hm = capi.HipModel(m.blob(), 0); o = Oracle(m.blob())
    N = 256; rng = np.random.default_rng(1)
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    q = (0.5 * (lo + hi) + 0.45 * (hi - lo) * rng.uniform(-1, 1, (N, m.nq))).astype(np.float32)
    v = rng.normal(0, 0.5, (N, m.nv)).astype(np.float32); a = rng.uniform(0, 1, (N, m.nu)).astype(np.float32); c = rng.uniform(0, 1, (N, m.nu)).astype(np.float32)
    b = capi.HipBatch(hm, N)
    for f, x in ((capi.F_QPOS, q), (capi.F_QVEL, v), (capi.F_ACT, a), (capi.F_CTRL, c)): b.write(f, x)
    b.step(None, capi.ACTMAP_NONE, nsub)
    gq, gv, gl = b.read(capi.F_QPOS), b.read(capi.F_QVEL), b.read(capi.F_TENLEN)
    eq = ev = el = 0.0
    for e in range(N):
        o.reset(); o.set_state(qpos=q[e], qvel=v[e], act=a[e], ctrl=c[e]); o.step(nsub)
        eq = max(eq, float(np.abs(gq[e] - o.field("qpos")).max())); ev = max(ev, float(np.abs(gv[e] - o.field("qvel")).max()))
    out[name] = dict(qpos=eq, qvel=ev)
print(json.dumps(out))
''' % (ROOT, ROOT)
res = {}
for tag, lib in (("polynomial (shipped)", "libmyo_hip.so"), ("library asinf / sincosf", "libmyo_hip_exacttrig.so")):
    p = os.path.join(ROOT, "myosuite_mjx_amd", lib)
    if not os.path.exists(p):
        print("missing", p); continue
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, MYO_HIP_LIB=p), capture_output=True, text=True)
    res[tag] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-400:]
print(json.dumps(res, indent=1))
