"""Developer probe: distribution over envs of the work features of one env step (headline workload, mid-rollout): Newton iterations,
factorisations, line-search evaluations, MPR support evaluations, contacts -- and how they relate to the predicted cost."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_mjx_amd import capi
from myosuite_mjx_amd.envs import BatchedMyoEnv

B = 4096
env = BatchedMyoEnv("myoHandPoseRandom-v0", num_envs=B, as_torch=False)
mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
env.reset(seed=1)
for rep in range(3):
    env.batch.bench_rollout(37, 10, 0, mode, env.max_episode_steps, None)
    d = env.batch.read(capi.F_DIAG)
    cost, cand, ncon_sum, mpr = d[:, 3], d[:, 4] & 0xFFFF, d[:, 4] >> 16, d[:, 5]
    itcon, it, ls, fact = d[:, 6] & 0xFFFF, d[:, 6] >> 16, d[:, 7] & 0xFFFF, d[:, 7] >> 16
    q = lambda x: " ".join(f"{v:7.1f}" for v in np.percentile(x, [50, 90, 99, 100]))
    print(f"-- after {37 * (rep + 1)} env steps: percentiles 50 / 90 / 99 / max over {B} envs, per env step (10 substeps)")
    for n, x in (("newton iterations", it), ("factorisations (newton)", fact), ("line-search evals", ls), ("mpr support evals (slowest lane)", mpr), ("contacts (sum over substeps)", ncon_sum),
                 ("candidates (sum)", cand), ("predicted cost (k-cycles)", cost)):
        print(f"   {n:36s} {q(x)}   mean {x.mean():.1f}")
    heavy = np.argsort(cost)[-40:]
    print("   heaviest 40 envs: iterations mean %.1f, factorisations %.1f, mpr %.1f, contacts/substep %.1f ; all envs: %.1f %.1f %.1f %.1f" %
          (it[heavy].mean(), fact[heavy].mean(), mpr[heavy].mean(), ncon_sum[heavy].mean() / 10, it.mean(), fact.mean(), mpr.mean(), ncon_sum.mean() / 10))
