"""Soak run on the GPU box: long rollouts of the main envs at full batch with U(-1,1) actions and auto-reset, counting every per-env fault
flag (bad state / bad qacc resets, contact or candidate table overflows, scheduler time-outs) and checking the final states.
Writes gpurun_out/r3_soak.json (copy to profiles/)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv

PLAN = [("myoHandPoseRandom-v0", 4096, 20000), ("myoHandReachRandom-v0", 4096, 5000), ("myoLegWalk-v0", 4096, 10000),
        ("myoLegRoughTerrainWalk-v0", 4096, 5000), ("myoLegStairTerrainWalk-v0", 4096, 3000), ("myoHandObjHoldRandom-v0", 4096, 5000),
        ("myoFingerPoseRandom-v0", 4096, 10000), ("myoElbowPose1D6MExoFixed-v0", 4096, 10000)]
SCALE = int(os.environ.get("SOAK_SCALE", 1))      # multiplies every step count (the committed run: SOAK_SCALE=4)
PLAN = [(e, B, n * SCALE) for e, B, n in PLAN]
out = []
mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
for env_id, B, steps in PLAN:
    env = BatchedMyoEnv(env_id, num_envs=B, as_torch=False, seed=1)
    env.reset(seed=1)
    counts = np.zeros(5, np.int64)
    t0 = time.time()
    chunk = 500
    for k in range(0, steps, chunk):
        env.batch.bench_rollout(min(chunk, steps - k), env.frame_skip, 1, mode, env.max_episode_steps, None)
        fl = env.status()                                  # flags accumulated since the last call, per env
        for b in range(5):
            counts[b] += int(((fl >> b) & 1).sum())
    st = env.get_env_state()
    rec = {"env": env_id, "envs": B, "env_steps": int(B) * int(steps), "seconds": round(time.time() - t0, 2),
           "envs_flagged_per_500_step_window": {"bad_state_reset": int(counts[0]), "bad_qacc_reset": int(counts[1]), "contact_overflow": int(counts[2]),
                                                "candidate_overflow": int(counts[3]), "sched_timeout": int(counts[4])},
           "final_state_finite": bool(all(np.isfinite(v).all() for v in st.values())),
           "final_act_in_01": bool((st["act"] >= 0).all() and (st["act"] <= 1).all())}
    print(json.dumps(rec), flush=True)
    out.append(rec)
# MyoDM TrackEnv (torch-side reward / reference lookup / auto-reset): 4096 envs x 2000 env steps of 5 substeps
try:
    import torch
    from myosuite_mjx_amd.track import TrackEnv
    B, steps = 4096, 2000 * SCALE
    tenv = TrackEnv(num_envs=B, seed=1, autoreset=True)
    tenv.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    counts = np.zeros(5, np.int64)
    t0 = time.time()
    ok_finite, n_done = True, 0
    for k in range(steps):
        obs, rew, done, info = tenv.step(torch.rand((B, tenv.act_dim), device="cuda", generator=g) * 2 - 1)
        if (k + 1) % 500 == 0:
            fl = tenv.status()
            for b in range(5):
                counts[b] += int(((fl >> b) & 1).sum())
            ok_finite = ok_finite and bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all())
            n_done += int(done.sum())
    torch.cuda.synchronize()
    rec = {"env": "MyoDM-TrackEnv (airplane)", "envs": B, "env_steps": B * steps, "seconds": round(time.time() - t0, 2),
           "envs_flagged_per_500_step_window": {"bad_state_reset": int(counts[0]), "bad_qacc_reset": int(counts[1]), "contact_overflow": int(counts[2]),
                                                "candidate_overflow": int(counts[3]), "sched_timeout": int(counts[4])},
           "final_state_finite": ok_finite, "final_act_in_01": True}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    # every MyoDM object: MyoHand<Object>Random-v0 at 2048 envs x 300 (x SOAK_SCALE) env steps (random actions, auto-reset, TimeLimit 50)
    from myosuite_mjx_amd import envs as _envs
    import myosuite_mjx_amd as myo
    tot, flagged, nonfinite, t0 = 0, {}, [], time.time()
    for obj in _envs.MYODM_OBJECTS:
        env = myo.make(f"MyoHand{obj.title()}Random-v0", num_envs=2048, seed=2, autoreset=True)
        env.reset()
        fl = np.zeros(2048, np.int64)
        for k in range(300 * SCALE):
            obs, rew, term, trunc, info = env.step(torch.rand((2048, env.act_dim), device="cuda", generator=g) * 2 - 1)
            if (k + 1) % 100 == 0:
                fl |= env.status().astype(np.int64)
        if not (bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all())):
            nonfinite.append(obj)
        if (fl != 0).any():
            flagged[obj] = {f"bit{b}": int(((fl >> b) & 1).sum()) for b in range(5) if ((fl >> b) & 1).any()}
        tot += 2048 * 300 * SCALE
        del env
    rec = {"env": "MyoDM TrackEnv, all %d objects (MyoHand<Object>Random-v0)" % len(_envs.MYODM_OBJECTS), "envs": 2048, "env_steps": tot, "seconds": round(time.time() - t0, 2),
           "objects_with_flagged_envs": flagged, "objects_with_non_finite_output": nonfinite}
    print(json.dumps(rec), flush=True)
    out.append(rec)
except ImportError:
    pass
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(os.path.join("gpurun_out", "r3_soak.json"), "w"), indent=1)
