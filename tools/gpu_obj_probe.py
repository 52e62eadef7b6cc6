"""Developer probe: HIP vs f64 oracle vs f32 oracle on the grasp frames of one MyoDM object (tests/golden/myodm_grasp_frames.npz)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from myosuite_mjx_amd import capi, model as M, track as T
from oracle.oracle import Oracle
import test_gpu_track as TT
obj = sys.argv[1]; nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = M.load_asset(f"myohand_object_{obj}"); hm = capi.HipModel(m.blob(), 0)
f = np.load(os.path.join(ROOT, "tests", "golden", "myodm_grasp_frames.npz"))
R, O = f[obj + "__robot"], f[obj + "__object"]
rng = np.random.default_rng(7); n = len(R)
q = np.zeros((n, m.nq)); q[:, :29] = R + rng.normal(0, 0.01, (n, 29)) * (np.arange(29) >= 6); q[:, 29:32] = O[:, :3]; q[:, 32:35] = np.stack([T.quat2euler(o[3:]) for o in O])
v = rng.normal(0, 0.2, (n, m.nv)); act = rng.uniform(0, 1, (n, m.nu)); act[:, :6] = 0; ctrl = rng.uniform(0, 1, (n, m.nu)); ctrl[:, :6] = q[:, :6]
q, v, act, ctrl = (x.astype(np.float32) for x in (q, v, act, ctrl))
g, r = TT._run(m, hm, q, v, act, ctrl, nsub)
o32 = Oracle(m.blob(), f32=True)
for e in range(n):
    o32.reset(); o32.set_state(qpos=q[e], qvel=v[e], act=act[e], ctrl=ctrl[e])
    nc32 = 0
    for s in range(nsub):
        o32.step(1); nc32 += o32.ncon
    e_hip = np.abs(g["qpos"][e] - r["qpos"][e]).max(); e_32 = np.abs(o32.field("qpos") - r["qpos"][e]).max()
    print(f"frame {e:2d} ncon f64 {r['ncon'][e]:3d} (sum {r['ncon_sum'][e]:4d}) hip {g['diag'][e,1]:3d} (sum {g['diag'][e,4]>>16:4d}) f32 sum {nc32:4d} | qpos err hip {e_hip:.2e} f32-oracle {e_32:.2e} | nefc {r['nefc'][e]} iters hip {g['diag'][e,2]} flags {g['flags'][e]}")
