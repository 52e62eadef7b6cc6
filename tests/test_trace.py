"""Trace-format rollouts (myosuite_mjx_amd/trace.py; reference: logger/grouped_datasets.py, envs/env_base.py:905-1010, 643-705).
CPU: trial splitting and the two file layouts.  GPU: a batched rollout logged, saved, loaded and replayed from a logged state."""
import numpy as np
import pytest


def _steps(T1, B):
    rng = np.random.default_rng(0)
    return {"time": np.tile(np.arange(T1)[:, None] * 0.02, (1, B)), "observations": rng.normal(size=(T1, B, 5)),
            "actions": rng.normal(size=(T1, B, 3)), "rewards": rng.normal(size=(T1, B)),
            "env_infos/state/qpos": rng.normal(size=(T1, B, 4))}


def test_split_trials_cuts_each_env_at_its_done_row():
    from myosuite_mjx_amd import trace
    T1, B = 7, 4
    steps = _steps(T1, B)
    done = np.zeros((T1, B), bool)
    done[3:, 1] = True          # env 1 ends at row 3
    done[0, 2] = True           # env 2 is done at its first row
    done[-1] = True             # horizon
    tr = trace.split_trials(steps, done, first_trial=10)
    assert sorted(tr) == ["Trial10", "Trial11", "Trial12", "Trial13"]
    assert [len(tr[f"Trial{10 + e}"]["time"]) for e in range(B)] == [7, 4, 1, 7]
    g = tr["Trial11"]
    assert np.array_equal(g["observations"], steps["observations"][:4, 1]) and np.array_equal(g["env_infos/state/qpos"], steps["env_infos/state/qpos"][:4, 1])
    assert np.isnan(g["actions"][-1]).all() and np.array_equal(g["actions"][:-1], steps["actions"][:3, 1])
    assert set(g) == set(steps)                                   # flat "/"-joined keys, one array per key (a closed Trace group)


@pytest.mark.parametrize("ext", ["pickle", "npz"])
def test_save_load_roundtrip(tmp_path, ext):
    from myosuite_mjx_amd import trace
    done = np.zeros((5, 3), bool)
    done[-1] = True
    root = {"myoHandPoseRandom-v0_rollouts": trace.split_trials(_steps(5, 3), done)}
    p = str(tmp_path / f"t.{ext}")
    trace.save(root, p)
    if ext == "pickle":
        with pytest.raises(ValueError):
            trace.load(p)
    back = trace.load(p, allow_pickle=(ext == "pickle"))
    assert list(back) == list(root)                                # Trace.load takes the first key as the trace name (grouped_datasets.py:428-431)
    for g, grp in root["myoHandPoseRandom-v0_rollouts"].items():
        for k, v in grp.items():
            assert np.array_equal(back["myoHandPoseRandom-v0_rollouts"][g][k], v, equal_nan=True)


@pytest.mark.gpu
def test_batched_rollout_to_trace_and_back(tmp_path):
    import torch
    import myosuite_mjx_amd as myo
    from myosuite_mjx_amd import trace
    B, H = 24, 30
    env = myo.make("myoHandReachRandom-v0", num_envs=B, seed=8, autoreset=False)
    with pytest.raises(ValueError):
        trace.rollout(myo.make("myoHandReachRandom-v0", num_envs=2), horizon=2)
    root = trace.rollout(env, policy=None, horizon=H, seed=8)
    (name, trials), = root.items()
    assert name == "myoHandReachRandom-v0_rollouts" and len(trials) == B
    lens = np.array([len(trials[f"Trial{e}"]["time"]) for e in range(B)])
    assert lens.max() <= H + 1 and lens.min() >= 2
    for e in (0, 5, B - 1):
        g = trials[f"Trial{e}"]
        n = lens[e]
        assert g["observations"].shape == (n, 115) and g["actions"].shape == (n, 39) and np.isnan(g["actions"][-1]).all() and np.isfinite(g["actions"][:-1]).all()
        assert np.allclose(g["time"], np.arange(n) * 0.02, atol=1e-5) and np.allclose(g["env_infos/state/qpos"], g["observations"][:, :23], atol=1e-7)
        assert not g["done"][:-1].any() and (g["done"][-1] or n == H + 1)       # a trial ends at its first done row or at the horizon
        assert np.array_equal(g["rewards"], g["env_infos/rwd_dense"])
    p = str(tmp_path / "rollouts.pickle")
    trace.save(root, p)
    back = trace.load(p, allow_pickle=True)[name]
    # replay: put every env into its logged state at row 3 and apply the logged action -> the logged row 4 (solver warm start is not part of
    # the env state, as in the reference: agreement at solver-tolerance level)
    ok = [e for e in range(B) if lens[e] > 6]
    assert len(ok) > B // 2
    env2 = myo.make("myoHandReachRandom-v0", num_envs=B, seed=1, autoreset=False)
    env2.reset(seed=1)
    trace.set_env_state(env2, [trace.state_row(back[f"Trial{e}"], 3 if e in ok else 0) for e in range(B)])
    act = np.stack([back[f"Trial{e}"]["actions"][3 if e in ok else 0] for e in range(B)])
    obs, rwd, term, trunc, info = env2.step(torch.as_tensor(act, device="cuda"))
    o = obs.cpu().numpy()
    for e in ok:
        assert np.abs(o[e] - back[f"Trial{e}"]["observations"][4]).max() < 2e-3, e
        assert abs(float(rwd[e]) - back[f"Trial{e}"]["rewards"][4]) < 5e-3
