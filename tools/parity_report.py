"""SURVEY.md 8d parity protocol, run on the GPU box: N seeded states per config, identical float32 state injected into the
f64 oracle and the HIP path, ONE env step (10 substeps), max-abs-err of qpos / qvel / act and of the observation vector.
Beside each HIP figure the report carries the FLOAT32 FLOOR: the same states through the float32 BUILD of the oracle (same algorithm and
operation order as the f64 checker, `real` = float) against the f64 oracle -- what single precision alone does to these states.
Writes gpurun_out/parity_report.json (copy to profiles/)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_mjx_amd import model as M, capi
from oracle.oracle import Oracle

N = int(os.environ.get("N", 1024))


def hand_like_states(m, N, seed):
    rng = np.random.default_rng(seed)
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    f32 = np.float32
    return (rng.uniform(lo, hi, (N, m.nq)).astype(f32), rng.normal(0, 0.5, (N, m.nv)).astype(f32),
            rng.uniform(0, 1, (N, m.nu)).astype(f32), rng.uniform(0, 1, (N, m.nu)).astype(f32))


def leg_states(m, N, seed):
    rng = np.random.default_rng(seed)
    keys = np.asarray(m.key_qpos).reshape(-1, m.nq)
    q = keys[rng.integers(0, len(keys), N)].copy()
    q[:, 7:] += rng.normal(0, 0.05, (N, m.nq - 7))
    q[:, 2] += rng.uniform(-0.06, 0.04, N)          # from 6 cm into the floor to 4 cm above it
    quat = q[:, 3:7] + rng.normal(0, 0.03, (N, 4))
    q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    f32 = np.float32
    return q.astype(f32), rng.normal(0, 0.5, (N, m.nv)).astype(f32), rng.uniform(0, 1, (N, m.nu)).astype(f32), rng.uniform(0, 1, (N, m.nu)).astype(f32)


def run(name, m, st, nsub=10):
    qpos, qvel, act, ctrl = st
    hm = capi.HipModel(m.blob(), 0)
    o = Oracle(m.blob())
    b = capi.HipBatch(hm, N)
    b.write(capi.F_QPOS, qpos); b.write(capi.F_QVEL, qvel); b.write(capi.F_ACT, act); b.write(capi.F_CTRL, ctrl)
    b.step(None, capi.ACTMAP_NONE, nsub)
    g = {k: b.read(f) for k, f in dict(qpos=capi.F_QPOS, qvel=capi.F_QVEL, act=capi.F_ACT, diag=capi.F_DIAG).items()}
    flags = b.status()
    r = {k: np.zeros((N, g[k].shape[1])) for k in ("qpos", "qvel", "act")}
    ncon = np.zeros(N, int)
    nefc = np.zeros(N, int)
    t0 = time.time()
    for e in range(N):
        o.reset(); o.set_state(qpos=qpos[e], qvel=qvel[e], act=act[e], ctrl=ctrl[e], warm=np.zeros(m.nv), time=0)
        o.step(nsub)
        r["qpos"][e], r["qvel"][e], r["act"][e] = o.field("qpos"), o.field("qvel"), o.field("act")
        ncon[e] = o.ncon
        nefc[e] = o.nefc
    o32 = Oracle(m.blob(), f32=True)
    r32 = {k: np.zeros_like(r[k]) for k in r}
    ncon32 = np.zeros(N, int)
    for e in range(N):
        o32.reset(); o32.set_state(qpos=qpos[e], qvel=qvel[e], act=act[e], ctrl=ctrl[e], warm=np.zeros(m.nv), time=0)
        o32.step(nsub)
        r32["qpos"][e], r32["qvel"][e], r32["act"][e] = o32.field("qpos"), o32.field("qvel"), o32.field("act")
        ncon32[e] = o32.ncon
    same = (g["diag"][:, 1] == ncon) & (flags == 0)
    free = same & (ncon == 0)
    cont = same & (ncon > 0)
    dt = nsub * m.timestep
    out = dict(config=name, N=N, substeps=nsub, flagged=int((flags != 0).sum()), contact_count_mismatch=int((g["diag"][:, 1] != ncon).sum()),
               flag_bits={str(b): int(((flags >> i) & 1).sum()) for i, b in enumerate(("bad_state", "bad_qacc", "contact_overflow", "cand_overflow"))},
               last_substep_row_count_mismatch=int((g["diag"][:, 0] != nefc)[same].sum()),
               envs_contact_free=int(free.sum()), envs_with_contacts=int(cont.sum()), oracle_seconds=round(time.time() - t0, 1))
    for label, mask in (("contact_free", free), ("with_contacts", cont)):
        if mask.sum() == 0:
            continue
        d = {k: np.abs(g[k] - r[k])[mask] for k in r}
        out[label] = {k: dict(max=float(v.max()), p99=float(np.quantile(v.max(axis=1), 0.99)), median=float(np.median(v.max(axis=1)))) for k, v in d.items()}
        m32 = mask & (ncon32 == ncon)
        d32 = {k: np.abs(r32[k] - r[k])[m32] for k in r}
        out[label]["float32_oracle_floor"] = {k: dict(max=float(v.max()), p99=float(np.quantile(v.max(axis=1), 0.99)), median=float(np.median(v.max(axis=1)))) for k, v in d32.items()}
        out[label]["float32_oracle_floor"]["envs"] = int(m32.sum())
        worst = int(np.flatnonzero(mask)[np.argmax(d["qpos"].max(axis=1))])
        out[label]["worst_env"] = dict(env=worst, hip_qpos_err=float(np.abs(g["qpos"] - r["qpos"])[worst].max()), float32_oracle_qpos_err=float(np.abs(r32["qpos"] - r["qpos"])[worst].max()))
        # observation entries built from the state: qpos, qvel * dt, act (pose / reach / walk layouts, Appendix C)
        out[label]["obs_state_entries_max"] = float(max(d["qpos"].max(), d["qvel"].max() * dt, d["act"].max()))
        # the same figures over the envs whose active-row count of the last substep agrees (a limit / contact sitting exactly at its
        # margin can be active in float32 and inactive in float64 or vice versa: a different, equally valid, trajectory)
        m2 = (g["diag"][:, 0] == nefc)[mask]
        if m2.sum():
            out[label]["same_active_set"] = {k: float(v[m2].max()) for k, v in d.items()}
            out[label]["same_active_set"]["envs"] = int(m2.sum())
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    rep = []
    hand = M.load_asset("myohand_pose")
    rep.append(run("MyoHand (myoHandPose*/Reach*), U(jnt_range) states", hand, hand_like_states(hand, N, 0)))
    finger = M.load_asset("myofinger_v0")
    rep.append(run("MyoFinger (myoFingerPose*), U(jnt_range) states", finger, hand_like_states(finger, N, 1)))
    legs = M.load_asset("myolegs")
    rep.append(run("MyoLeg (myoLegWalk-v0), keyframes + noise, floor contact", legs, leg_states(legs, N, 2)))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rep, open("gpurun_out/parity_report.json", "w"), indent=1)
