"""Developer parity probe: HIP step vs the f64 oracle on seeded random states (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_mjx_amd import model as M, capi
from oracle.oracle import Oracle

def run(m, hm, o, N, nsub, switches, seed, vel_sigma=0.5, label=""):
    rng = np.random.default_rng(seed)
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    qpos = rng.uniform(lo, hi, (N, m.nq)) if not switches[1] or True else None
    if label.startswith("limit"):   # push some joints slightly past their limits
        qpos += rng.normal(0, 0.05, qpos.shape) * (rng.random(qpos.shape) < 0.3)
    qvel = rng.normal(0, vel_sigma, (N, m.nv))
    act = rng.uniform(0, 1, (N, m.nu))
    ctrl = rng.uniform(0, 1, (N, m.nu))
    hm.set_switch(*switches); o.switches(*switches)
    b = capi.HipBatch(hm, N)
    b.write(capi.F_QPOS, qpos); b.write(capi.F_QVEL, qvel); b.write(capi.F_ACT, act); b.write(capi.F_CTRL, ctrl)
    b.step(None, capi.ACTMAP_NONE, nsub)
    g = {k: b.read(f) for k, f in dict(qpos=capi.F_QPOS, qvel=capi.F_QVEL, act=capi.F_ACT, qacc=capi.F_QACC, tenlen=capi.F_TENLEN, force=capi.F_ACTFORCE, diag=capi.F_DIAG).items()}
    flags = b.status()
    ref = {k: np.zeros_like(v, dtype=np.float64) for k, v in g.items() if k != "diag"}
    nefc = np.zeros(N, int); ncon = np.zeros(N, int); its = np.zeros(N, int)
    for e in range(N):
        o.reset(); o.set_state(qpos=qpos[e].astype(np.float32), qvel=qvel[e].astype(np.float32), act=act[e].astype(np.float32), ctrl=ctrl[e].astype(np.float32), warm=np.zeros(m.nv), time=0)
        for s in range(nsub):
            o.step(1)
        ref["qpos"][e] = o.field("qpos"); ref["qvel"][e] = o.field("qvel"); ref["act"][e] = o.field("act"); ref["qacc"][e] = o.field("qacc")
        ref["tenlen"][e] = o.field("actuator_length"); ref["force"][e] = o.field("actuator_force")
        nefc[e] = o.nefc; ncon[e] = o.ncon; its[e] = o.solver_iter
    print(f"--- {label} N={N} nsub={nsub} switches={switches}: oracle nefc mean {nefc.mean():.1f} max {nefc.max()} ncon max {ncon.max()} iter max {its.max()} | hip nefc max {g['diag'][:,0].max()} ncon max {g['diag'][:,1].max()} iter max {g['diag'][:,2].max()} flags {np.bincount(flags, minlength=2)[:8]}")
    ok = (flags == 0) & (g['diag'][:,1] == ncon)
    print(f"   envs compared {ok.sum()} (flagged {np.sum(flags!=0)}, ncon mismatch {np.sum(g['diag'][:,1] != ncon)})")
    for e in np.where((flags == 0) & (g['diag'][:,1] != ncon))[0][:5]:
        print(f"      env {e}: hip ncon {g['diag'][e,1]} nefc {g['diag'][e,0]} | oracle ncon {ncon[e]} nefc {nefc[e]}")
    for k in ref:
        err = np.abs(g[k] - ref[k])[ok]
        worst = np.unravel_index(np.argmax(err), err.shape)
        scale = np.abs(ref[k]).max()
        print(f"   {k:7s} max|err| {err.max():.3e} (ref scale {scale:.3e}) at env {worst[0]} idx {worst[1]}  p99 {np.quantile(err, 0.99):.3e}")
    mism = (g['diag'][:,0] != nefc).sum()
    print(f"   nefc mismatches {mism} / {N}; ncon mismatches {(g['diag'][:,1] != ncon).sum()}")
    return g, ref

if __name__ == "__main__":
    m = M.load_asset("myohand_pose")
    hm = capi.HipModel(m.blob(), 0)
    print("dims", {n: getattr(hm.dims, n) for n, _ in hm.dims._fields_})
    o = Oracle(m.blob())
    N = int(os.environ.get("N", 256))
    t = time.time()
    run(m, hm, o, N, 1, (1, 1, 1), 0, label="smooth-1sub")
    run(m, hm, o, N, 10, (1, 1, 1), 1, label="smooth-10sub")
    run(m, hm, o, N, 1, (1, 0, 1), 2, label="limit-1sub")
    run(m, hm, o, N, 10, (1, 0, 1), 3, label="limit-10sub")
    run(m, hm, o, N, 1, (0, 0, 1), 4, label="capsule-contacts-1sub")
    run(m, hm, o, N, 10, (0, 0, 1), 5, label="capsule-contacts-10sub")
    run(m, hm, o, N, 1, (0, 0, 0), 6, label="all-1sub")
    run(m, hm, o, N, 10, (0, 0, 0), 7, label="all-10sub")
    print("elapsed", time.time() - t)
