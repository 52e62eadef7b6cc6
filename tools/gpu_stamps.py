"""Developer probe (diagnostic build, -DMYO_STAMPS=1): where a substep spends its cycles, per lanes-per-env setting."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MYO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myosuite_mjx_amd", "libmyo_hip_stamps.so")
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv

B = int(os.environ.get("B", 4096))
NAMES = ["load/check", "kinematics", "tendon+muscle", "dynamics(CRB/RNE)", "collision (wave kernel: narrow phase)", "constraint rows", "wave kernel: geom frames + broad phase", "newton", "euler", "store"]
ENV = os.environ.get("ENV", "myoHandPoseRandom-v0")
env = BatchedMyoEnv(ENV, num_envs=B, as_torch=False)
mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
for lanes, bal in ((64, 0), (64, 1)):
    capi.set_lanes(lanes)
    env.batch.set_balance(bal)
    env.reset(seed=1)
    env.batch.bench_rollout(30, 10, 0, mode, env.max_episode_steps, None)
    ms = env.batch.bench_rollout(10, 10, 0, 0, 0, None) / 10
    nwg = B // (64 // lanes)
    st2, ok = capi.read_stamps(env.batch, 2 * nwg)      # second half: finer split of tendon / dynamics / Newton (wave kernel only)
    st, sub = st2[:nwg], st2[nwg:]
    tot = st[:, :10].sum(1)          # columns 10, 11 hold hardware ids, not cycles
    print(f"== lanes/env {lanes} balance {bal}: step kernel {ms:.3f} ms per env-step ({B/ms*1e3:,.0f} env-steps/s); per-WG total cycles mean {tot.mean():,.0f} max {tot.max():,.0f} (10 substeps)")
    for k, n in enumerate(NAMES):
        print(f"   {n:26s} {st[:,k].mean()/10:12,.0f} cycles/substep  {100*st[:,k].mean()/tot.mean():5.1f}%")
    SUBN = ["newton: forces, cost, gradient, convergence test", "newton: Hessian assembly", "newton+euler: build rows, Cholesky, store L", "newton+euler: triangular solves",
            "newton: M*search, J*search", "newton: line search", "newton: start (M*warm, J*warm)", "tendons: site / geom frames + wrap geometry", "tendons: moment arms of the straight pieces", "tendons: gather + muscle",
            "dynamics: cinert, cdof", "dynamics: RNE forward / backward, M assembly"]
    for k, n in enumerate(SUBN):
        if n != "-":
            print(f"      {n:52s} {sub[:,k].mean()/10:12,.0f} cycles/substep  {100*sub[:,k].mean()/tot.mean():5.1f}%")
