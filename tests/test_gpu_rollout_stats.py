"""GPU test (pytest -m gpu): SURVEY.md 8d parity protocol for LONG rollouts -- "compare distributions (mean reward, mean |qvel|, contact
count), not trajectories".  The HIP env and the f64 oracle start from the same reset states and receive the same action stream for 60
env steps (no auto-reset); contact-rich muscle dynamics are chaotic, so individual trajectories separate after a few dozen steps,
but the rollout statistics must agree.  Tolerances: a few per cent of each statistic (they measure sampling noise of 192 diverged
trajectories, not a numerical error)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sigmoid(a):
    return 1.0 / (1.0 + np.exp(-5.0 * (a - 0.5)))


@pytest.mark.parametrize("env_id,asset", [("myoHandPoseRandom-v0", "myohand_pose"), ("myoLegWalk-v0", "myolegs")])
def test_rollout_statistics_match_the_oracle(env_id, asset):
    import torch
    import myosuite_mjx_amd as myo
    from myosuite_mjx_amd import capi, model as M
    from oracle.oracle import Oracle
    m = M.load_asset(asset)
    B, K = 192, 60
    env = myo.make(env_id, num_envs=B, seed=12, autoreset=False)
    env.reset(seed=12)
    st = env.get_env_state()
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1, 1, (K, B, m.nu)).astype(np.float32)
    # oracle rollout: batched driver (one env per task over the host's threads), same controls as base_v0.py:87-91 produces
    o = Oracle(m.blob())
    q, v = st["qpos"].astype(np.float64), st["qvel"].astype(np.float64)
    a, w, t = st["act"].astype(np.float64), np.zeros((B, m.nv)), np.zeros((B, 1))
    nth = min(32, os.cpu_count() or 8)
    o_qvel, o_act, o_alive = [], [], np.ones(B, bool)
    g_qvel, g_act, g_ncon = [], [], []
    for k in range(K):
        ctrl = _sigmoid(acts[k].astype(np.float64))
        fl = o.step_batch(q, v, a, w, t, np.ascontiguousarray(ctrl), 10, nth)
        o_alive &= fl == 0
        o_qvel.append(np.abs(v).mean(1)); o_act.append(a.mean(1))
        obs, rwd, term, trunc, info = env.step(torch.as_tensor(acts[k], device="cuda"))
        s2 = env.get_env_state()
        g_qvel.append(np.abs(s2["qvel"]).mean(1)); g_act.append(s2["act"].mean(1)); g_ncon.append(env.batch.read(capi.F_DIAG)[:, 1])
    assert o_alive.mean() > 0.98 and (env.status() & (capi.FLAG_BAD_STATE | capi.FLAG_BAD_QACC) == 0).mean() > 0.98
    o_qvel, g_qvel, o_act, g_act = np.array(o_qvel), np.array(g_qvel), np.array(o_act), np.array(g_act)
    # activations are driven by the shared controls: they agree env by env (first-order filter of the same input)
    assert np.abs(o_act - g_act).max() < 1e-4
    # early steps: trajectories still together
    d0 = np.abs(o_qvel[:3] - g_qvel[:3])
    assert np.median(d0) < 1e-4 and d0.max() < 0.05          # (a contact switching one substep apart moves a single env by ~1e-2)
    # whole rollout: statistics of the diverged ensembles
    mo, mg = o_qvel[10:].mean(), g_qvel[10:].mean()
    assert abs(mo - mg) < 0.04 * mo, (mo, mg)
    so, sg = o_qvel[10:].std(), g_qvel[10:].std()
    assert abs(so - sg) < 0.12 * so, (so, sg)
    # final posture distribution (joint angles, degrees of freedom pooled): mean and spread
    qo, qg = q[:, -m.nv + (6 if asset == "myolegs" else 0):], env.get_env_state()["qpos"][:, -m.nv + (6 if asset == "myolegs" else 0):]
    assert np.abs(qo.mean(0) - qg.mean(0)).max() < 0.12 and abs(qo.std() - qg.std()) < 0.05 * qo.std() + 0.01
    assert np.mean(g_ncon) > 0.5                                        # contact-rich rollouts (hand: finger pads; legs: feet)
