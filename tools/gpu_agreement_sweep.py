"""Rare-event hunt: many seeded states through two instantiations of the step kernel (size-specialised / generic; they differ in unrolling and in the
tree-sparse Euler factorisation only) -- a disagreement far above float32 round-off marks a state where single precision sits on a knife edge that
float64 does not have (this is how the half-turn wrap candidate of DESIGN.md 3, float safeguards iii, was found).  The outliers are then stepped by the
float64 oracle to see which side is off.  Writes gpurun_out/r3_agreement_sweep.json.   usage: python tools/gpu_agreement_sweep.py [asset] [N] [seed]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_mjx_amd import capi, model as M
from oracle.oracle import Oracle
asset = sys.argv[1] if len(sys.argv) > 1 else "myohand_pose"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
m = M.load_asset(asset)
rng = np.random.default_rng(seed)
if m.nq == m.nv:
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    qpos = rng.uniform(lo, hi, (N, m.nq)).astype(np.float32)
else:
    qpos = np.tile(np.asarray(m.key_qpos).reshape(-1, m.nq)[2], (N, 1)).astype(np.float32)
    qpos[:, 7:] += rng.normal(0, 0.05, (N, m.nq - 7)).astype(np.float32)
    qpos[:, 2] -= 0.03
qvel = rng.normal(0, 0.3, (N, m.nv)).astype(np.float32)
act = rng.uniform(0, 1, (N, m.nu)).astype(np.float32)
out = []
for no_spec in ("0", "1"):
    os.environ["MYO_NO_SPEC"] = no_spec
    hm = capi.HipModel(m.blob(), 0)
    os.environ.pop("MYO_NO_SPEC", None)
    b = capi.HipBatch(hm, N)
    for f, a in ((capi.F_QPOS, qpos), (capi.F_QVEL, qvel), (capi.F_ACT, act), (capi.F_CTRL, act)):
        b.write(f, a)
    b.step(None, capi.ACTMAP_NONE, 10)
    out.append((b.read(capi.F_QPOS), b.read(capi.F_QVEL), b.read(capi.F_DIAG)[:, :3], b.status()))
d = np.abs(out[0][0] - out[1][0]).max(axis=1)
dv = np.abs(out[0][1] - out[1][1]).max(axis=1)
rec = {"asset": asset, "states": N, "seed": seed, "substeps": 10, "flagged": [int((out[0][3] != 0).sum()), int((out[1][3] != 0).sum())],
       "qpos_disagreement": {"p50": float(np.median(d)), "p99": float(np.percentile(d, 99)), "p999": float(np.percentile(d, 99.9)), "max": float(d.max())},
       "states_above_1e-4": int((d > 1e-4).sum()), "states_above_1e-3": int((d > 1e-3).sum()), "outliers": []}
o = Oracle(m.blob())
for e in np.argsort(d)[::-1][:12]:
    if d[e] < 1e-4:
        break
    o.reset(); o.set_state(qpos=qpos[e].astype(float), qvel=qvel[e].astype(float), act=act[e].astype(float), ctrl=act[e].astype(float))
    o.step(10)
    q = o.field("qpos")
    rec["outliers"].append({"env": int(e), "spec_vs_generic": float(d[e]), "spec_vs_oracle": float(np.abs(out[0][0][e] - q).max()), "generic_vs_oracle": float(np.abs(out[1][0][e] - q).max()),
                            "ncon": int(o.ncon), "nefc": int(o.nefc), "diag_spec": out[0][2][e].tolist(), "diag_generic": out[1][2][e].tolist()})
print(json.dumps(rec, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
p = os.path.join(ROOT, "gpurun_out", "r3_agreement_sweep.json")
prev = json.load(open(p)) if os.path.exists(p) else []
json.dump(prev + [rec], open(p, "w"), indent=1)
