"""Developer parity probe for the MyoLeg model: HIP step vs the f64 oracle around the model's keyframes (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_mjx_amd import model as M, capi
from oracle.oracle import Oracle


def leg_states(m, N, seed, key=2, dz=0.0, jitter=0.05, vel_sigma=0.3):
    rng = np.random.default_rng(seed)
    q = np.tile(np.asarray(m.key_qpos).reshape(-1, m.nq)[key], (N, 1))
    q[:, 7:] += rng.normal(0, jitter, (N, m.nq - 7))
    q[:, 2] += dz + rng.uniform(-0.02, 0.02, N)
    quat = q[:, 3:7] + rng.normal(0, 0.03, (N, 4))
    q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    f32 = np.float32
    return q.astype(f32), rng.normal(0, vel_sigma, (N, m.nv)).astype(f32), rng.uniform(0, 1, (N, m.nu)).astype(f32), rng.uniform(0, 1, (N, m.nu)).astype(f32)


def run(m, hm, o, st, nsub, switches, label=""):
    qpos, qvel, act, ctrl = st
    N = qpos.shape[0]
    hm.set_switch(*switches); o.switches(*switches)
    b = capi.HipBatch(hm, N)
    b.write(capi.F_QPOS, qpos); b.write(capi.F_QVEL, qvel); b.write(capi.F_ACT, act); b.write(capi.F_CTRL, ctrl)
    b.step(None, capi.ACTMAP_NONE, nsub)
    g = {k: b.read(f) for k, f in dict(qpos=capi.F_QPOS, qvel=capi.F_QVEL, act=capi.F_ACT, qacc=capi.F_QACC, tenlen=capi.F_TENLEN, force=capi.F_ACTFORCE, diag=capi.F_DIAG).items()}
    flags = b.status()
    ref = {k: np.zeros_like(v, dtype=np.float64) for k, v in g.items() if k != "diag"}
    nefc = np.zeros(N, int); ncon = np.zeros(N, int); its = np.zeros(N, int)
    for e in range(N):
        o.reset(); o.set_state(qpos=qpos[e], qvel=qvel[e], act=act[e], ctrl=ctrl[e], warm=np.zeros(m.nv), time=0)
        o.step(nsub)
        ref["qpos"][e] = o.field("qpos"); ref["qvel"][e] = o.field("qvel"); ref["act"][e] = o.field("act"); ref["qacc"][e] = o.field("qacc")
        ref["tenlen"][e] = o.field("actuator_length"); ref["force"][e] = o.field("actuator_force")
        nefc[e] = o.nefc; ncon[e] = o.ncon; its[e] = o.solver_iter
    print(f"--- {label} N={N} nsub={nsub} switches={switches}: oracle nefc mean {nefc.mean():.1f} max {nefc.max()} ncon mean {ncon.mean():.1f} max {ncon.max()} iter max {its.max()} | hip nefc max {g['diag'][:,0].max()} ncon max {g['diag'][:,1].max()} iter max {g['diag'][:,2].max()} flags {np.bincount(flags, minlength=2)[:16]}")
    ok = (flags == 0) & (g['diag'][:, 1] == ncon)
    print(f"   envs compared {ok.sum()} (flagged {np.sum(flags != 0)}, ncon mismatch {np.sum(g['diag'][:,1] != ncon)}, nefc mismatch {np.sum(g['diag'][:,0] != nefc)})")
    for e in np.where((flags == 0) & (g['diag'][:, 1] != ncon))[0][:5]:
        print(f"      env {e}: hip ncon {g['diag'][e,1]} nefc {g['diag'][e,0]} | oracle ncon {ncon[e]} nefc {nefc[e]}")
    if ok.sum() == 0:
        return g, ref
    for k in ref:
        err = np.abs(g[k] - ref[k])[ok]
        worst = np.unravel_index(np.argmax(err), err.shape)
        print(f"   {k:7s} max|err| {err.max():.3e} (ref scale {np.abs(ref[k]).max():.3e}) at env {worst[0]} idx {worst[1]}  p99 {np.quantile(err, 0.99):.3e}")
    return g, ref


if __name__ == "__main__":
    m = M.load_asset("myolegs")
    hm = capi.HipModel(m.blob(), 0)
    print("dims", {n: getattr(hm.dims, n) for n, _ in hm.dims._fields_})
    o = Oracle(m.blob())
    N = int(os.environ.get("N", 64))
    t = time.time()
    run(m, hm, o, leg_states(m, N, 0, dz=0.5), 1, (1, 1, 1), "air-smooth+eq-1sub")
    run(m, hm, o, leg_states(m, N, 1, dz=0.5), 10, (1, 1, 1), "air-smooth+eq-10sub")
    run(m, hm, o, leg_states(m, N, 2, dz=0.5, jitter=0.3), 1, (1, 0, 1), "air-limits-1sub")
    run(m, hm, o, leg_states(m, N, 3, dz=0.5, jitter=0.3), 10, (1, 0, 1), "air-limits-10sub")
    run(m, hm, o, leg_states(m, N, 4, dz=-0.04), 1, (0, 0, 1), "ground-noellipsoid-1sub")
    run(m, hm, o, leg_states(m, N, 5, dz=-0.04), 1, (0, 0, 0), "ground-all-1sub")
    run(m, hm, o, leg_states(m, N, 6, dz=-0.04), 10, (0, 0, 0), "ground-all-10sub")
    run(m, hm, o, leg_states(m, N, 7, key=0, dz=-0.05, jitter=0.2), 10, (0, 0, 0), "key0-sunk-10sub")
    print("elapsed", time.time() - t)
