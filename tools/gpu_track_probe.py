"""Developer probe (GPU box): HIP vs f64 oracle vs f32 oracle on the TrackEnv model states of tests/test_gpu_track.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from myosuite_mjx_amd import capi, model as M
from oracle.oracle import Oracle
import test_gpu_track as T
m = M.load_asset("myohand_object_airplane")
hm = capi.HipModel(m.blob(), 0)
q, v, a, c = T._motion_states(m, list(range(0, 48)) + list(range(0, 48)), 2, jitter=0.01)
o32 = Oracle(m.blob(), f32=True)
for nsub in (1, 5):
    g, r = T._run(m, hm, q, v, a, c, nsub)
    same = (g["diag"][:, 1] == r["ncon"]) & ((g["diag"][:, 4] >> 16) == r["ncon_sum"])
    eq, ev = np.abs(g["qpos"] - r["qpos"]).max(1), np.abs(g["qvel"] - r["qvel"]).max(1)
    e32 = []
    for e in range(len(q)):
        o32.reset(); o32.set_state(qpos=q[e], qvel=v[e], act=a[e], ctrl=c[e]); o32.step(nsub)
        e32.append(np.abs(o32.field("qpos") - r["qpos"][e]).max())
    e32 = np.array(e32)
    print(f"nsub {nsub}: same {same.mean():.3f} ncon max {r['ncon'].max()} nefc max {r['nefc'].max()} | HIP qpos p50 {np.percentile(eq,50):.2e} p90 {np.percentile(eq,90):.2e} p99 {np.percentile(eq,99):.2e} max {eq.max():.2e} same-max {eq[same].max():.2e}"
          f" | qvel p50 {np.percentile(ev,50):.2e} p90 {np.percentile(ev,90):.2e} max {ev.max():.2e} | f32 oracle qpos p50 {np.percentile(e32,50):.2e} p90 {np.percentile(e32,90):.2e} max {e32.max():.2e}")
    w = np.argsort(eq)[::-1][:5]
    print("   worst", w, eq[w], "ncon", g["diag"][w, 1], r["ncon"][w], "iters", g["diag"][w, 2])
# throughput of the TRK kernel: 4096 envs from the motion's frames, 5 substeps per step
import time
B = 4096
qq, vv, aa, cc = T._motion_states(m, [int(t) for t in np.random.default_rng(0).integers(0, 60, B)], 5)
b = capi.HipBatch(hm, B)
for f, x in ((capi.F_QPOS, qq), (capi.F_QVEL, 0 * vv), (capi.F_ACT, aa), (capi.F_CTRL, cc)):
    b.write(f, x)
for _ in range(5): b.step(None, capi.ACTMAP_NONE, 5)
capi.sync()
t0 = time.time()
n = 40
for _ in range(n): b.step(None, capi.ACTMAP_NONE, 5)
capi.sync()
dt = (time.time() - t0) / n
print(f"TRK kernel: {dt*1e3:.3f} ms per env step (5 substeps) at B={B}: {B/dt/1e6:.3f} M env-steps/s; flags {np.unique(b.status())}, mean ncon {b.read(capi.F_DIAG)[:,1].mean():.2f}")
