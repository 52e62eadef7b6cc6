"""Developer probe / evidence: every MyoDM object with a compiled asset -- physics of the HIP TRK kernel against the float64 oracle (and the float32
build of the oracle as the yardstick of what single precision can deliver) on the twelve grasp frames of tests/golden/myodm_grasp_frames.npz, five
substeps each, and a 30-step random rollout of MyoHand<Object>Random-v0.  Writes gpurun_out/r3_myodm_objects.json (copy to profiles/)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import myosuite_mjx_amd as myo
from myosuite_mjx_amd import capi, model as M, track as T, envs
from oracle.oracle import Oracle
import test_gpu_track as TT
f = np.load(os.path.join(ROOT, "tests", "golden", "myodm_grasp_frames.npz"))
only = sys.argv[1:]
out = []
for obj in envs.MYODM_OBJECTS:
    if only and obj not in only:
        continue
    t0 = time.time()
    rec = {"object": obj}
    try:
        m = M.load_asset(f"myohand_object_{obj}"); hm = capi.HipModel(m.blob(), 0)
        rec.update(ncg=int(len(m.arrays["hip_cg_link"])), npair=int(len(m.arrays["hip_pair_i"])), hull_vertices=int(len(m.arrays["hip_mesh_vert"])))
        if obj + "__robot" in f.files:
            R, O = f[obj + "__robot"], f[obj + "__object"]
            rng = np.random.default_rng(7); n = len(R)
            q = np.zeros((n, m.nq)); q[:, :29] = R + rng.normal(0, 0.01, (n, 29)) * (np.arange(29) >= 6); q[:, 29:32] = O[:, :3]; q[:, 32:35] = np.stack([T.quat2euler(o[3:]) for o in O])
            v = rng.normal(0, 0.2, (n, m.nv)); act = rng.uniform(0, 1, (n, m.nu)); act[:, :6] = 0; ctrl = rng.uniform(0, 1, (n, m.nu)); ctrl[:, :6] = q[:, :6]
            q, v, act, ctrl = (x.astype(np.float32) for x in (q, v, act, ctrl))
            g, r = TT._run(m, hm, q, v, act, ctrl, 5)
            o32 = Oracle(m.blob(), f32=True); e32 = np.zeros(n)
            for e in range(n):
                o32.reset(); o32.set_state(qpos=q[e], qvel=v[e], act=act[e], ctrl=ctrl[e]); o32.step(5)
                e32[e] = np.abs(o32.field("qpos") - r["qpos"][e]).max()
            eq = np.abs(g["qpos"] - r["qpos"]).max(1)
            same = (g["diag"][:, 1] == r["ncon"]) & ((g["diag"][:, 4] >> 16) == r["ncon_sum"])
            well = e32 < 1e-4
            rec.update(motion=str(f[obj + "__motion"]), frames=int(n), flags=int((g["flags"] != 0).sum()), ncon_max=int(r["ncon_max"].max()), contact_counts_equal=int(same.sum()),
                       well_conditioned_frames=int(well.sum()), hip_err_on_well=float(eq[well].max()) if well.any() else None, hip_err_max=float(eq.max()), f32_oracle_err_max=float(e32.max()))
        env = myo.make(f"MyoHand{obj.title()}Random-v0", num_envs=32, seed=0, autoreset=True)
        obs = env.reset()
        gen = torch.Generator(device="cuda").manual_seed(0)
        fin = True
        for _ in range(30):
            obs, rew, term, trunc, info = env.step(torch.rand((32, env.act_dim), device="cuda", generator=gen) * 2 - 1)
            fin = fin and bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all())
        rec.update(rollout_finite=fin, rollout_flags=int((env.status() != 0).sum()))
    except Exception as ex:
        rec["error"] = repr(ex)[:300]
    rec["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r3_myodm_objects.json"), "w"), indent=1)
