"""Batched rollouts <-> the reference's `Trace` logger format (second half).

The reference records rollouts as a `Trace` (myosuite/logger/grouped_datasets.py:49-435): a dict
`{trace_name: {"Trial<k>": {dataset_key: array[T, ...]}}}` whose groups are closed (stacked, flattened with "/" keys:
utils/dict_utils.py:72-87) and written with `pickle.dump(trace.root, ...)` (:403) or as one h5 group per trial (:392-401).
`MujocoEnv.examine_policy_new` (envs/env_base.py:905-1010) fills it with, per step t: `time`, `observations`, `actions`,
`rewards`, `done` and the flattened `env_infos` (time, rwd_dense, solved, done, state/{time,qpos,qvel,act}: :559-570, 643-677);
the last row of an episode carries NaN actions.

Here B envs step together, so one rollout call yields B trials: `rollout(env, policy, horizon)` steps a `BatchedMyoEnv`
(auto-reset off), keeps the batched per-step arrays on the host and `split_trials` cuts them into per-env groups at each
env's `done`.  `save` writes the pickle layout `Trace.load` reads back (plain dicts and numpy arrays only) or an `.npz`
with "Trial<k>/<key>" names; `load` reads either.  `set_env_state` takes a row of a trial's `env_infos/state/*` to put an
env back into a logged state (envs/env_base.py:679-705).
"""
from __future__ import annotations

import pickle

import numpy as np

STATE_KEYS = ("time", "qpos", "qvel", "act", "target")   # target: where the reference keeps site_pos / body_pos edits of the task


def split_trials(steps: dict, done: np.ndarray, first_trial: int = 0) -> dict:
    """steps: {key: array[T+1, B, ...]} batched records (row t = MDP(t)); done: bool[T+1, B].
    Returns {"Trial<k>": {key: array[T_e, ...]}} with T_e = index of the env's first done row + 1 (or T+1)."""
    T1, B = done.shape
    out = {}
    for e in range(B):
        hit = np.nonzero(done[:, e])[0]
        n = int(hit[0]) + 1 if hit.size else T1
        grp = {}
        for k, v in steps.items():
            a = np.array(v[:n, e])
            if k == "actions":
                a[n - 1] = np.nan                                   # env_base.py:993: the final row has no action
            grp[k] = a
        out[f"Trial{first_trial + e}"] = grp
    return out


def rollout(env, policy=None, horizon=100, seed=None, name=None) -> dict:
    """Roll `env` (BatchedMyoEnv, autoreset=False) for up to `horizon` steps under `policy(obs) -> action[B, nu]` (None: U(-1,1)
    actions from a seeded numpy generator) and return the Trace root dict `{name: {Trial<e>: {...}}}`."""
    if env.autoreset:
        raise ValueError("rollout() needs an env made with autoreset=False: a trial ends at its env's done")
    rng = np.random.default_rng(seed)
    B = env.num_envs
    host = (lambda x: np.array(x.cpu().numpy() if hasattr(x, "cpu") else x))
    obs = env.reset(seed=seed)
    rec = {k: [] for k in ("time", "observations", "actions", "rewards", "done", "env_infos/time", "env_infos/rwd_dense", "env_infos/solved",
                           "env_infos/done", *[f"env_infos/state/{k}" for k in STATE_KEYS])}

    def log(obs, act, rwd, done, solved):
        st = env.get_env_state()
        t = st["time"].reshape(B)
        rec["time"].append(t); rec["observations"].append(host(obs)); rec["actions"].append(act)
        rec["rewards"].append(rwd); rec["done"].append(done)
        rec["env_infos/time"].append(t); rec["env_infos/rwd_dense"].append(rwd); rec["env_infos/solved"].append(solved); rec["env_infos/done"].append(done)
        for k in STATE_KEYS:
            rec[f"env_infos/state/{k}"].append(st[k].reshape(B, -1) if k != "time" else t)

    rwd, done, solved = np.zeros(B, np.float32), np.zeros(B, bool), np.zeros(B, bool)
    for t in range(horizon):
        act = np.asarray(policy(obs), np.float32).reshape(B, env.act_dim) if policy is not None else rng.uniform(-1, 1, (B, env.act_dim)).astype(np.float32)
        log(obs, act, rwd, done, solved)
        if done.all():
            break
        obs, r, d, trunc, info = env.step(env._torch.as_tensor(act, device=obs.device) if env.as_torch else act)
        rwd, solved = host(r).astype(np.float32), host(info["solved"]).astype(bool)
        done = done | host(d).astype(bool)
    log(obs, np.full((B, env.act_dim), np.nan, np.float32), rwd, done, solved)
    steps = {k: np.stack(v) for k, v in rec.items()}
    final_done = steps["done"].copy()
    final_done[-1] = True                                            # horizon reached: every trial ends at the last row at the latest
    return {name or (env.id + "_rollouts"): split_trials(steps, final_done)}


def save(root: dict, path: str) -> None:
    """Write a Trace root dict: *.pickle (what `Trace.load` of the reference reads: grouped_datasets.py:427-433) or *.npz."""
    if path.endswith(".npz"):
        (name, trials), = root.items()
        flat = {"__name__": np.array(name)}
        for g, grp in trials.items():
            for k, v in grp.items():
                flat[f"{g}/{k}"] = v
        np.savez_compressed(path, **flat)
    else:
        with open(path, "wb") as f:
            pickle.dump(root, f)


def load(path: str, allow_pickle: bool = False) -> dict:
    """Read back what `save` wrote.  `.npz` files are read with allow_pickle=False.  The pickle layout (what the reference's Trace.load
    reads) is only opened when the caller passes allow_pickle=True, i.e. vouches that the file is one of its own: unpickling executes
    code from the file, and files of unknown origin -- reference checkpoints included -- must never be read this way."""
    if path.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        name = str(z["__name__"])
        trials = {}
        for key in z.files:
            if key == "__name__":
                continue
            g, k = key.split("/", 1)
            trials.setdefault(g, {})[k] = z[key]
        return {name: trials}
    if not allow_pickle:
        raise ValueError("trace.load: pass allow_pickle=True to read a pickle written by trace.save (never for files of unknown origin)")
    with open(path, "rb") as f:
        return pickle.load(f)


def state_row(trial: dict, t: int) -> dict:
    """The env state logged at row t of a trial, in get_env_state()'s key layout."""
    return {k: np.asarray(trial[f"env_infos/state/{k}"][t]) for k in STATE_KEYS}


def set_env_state(env, states: list) -> None:
    """Put env e into states[e] (dicts from `state_row`), like MujocoEnv.set_env_state (envs/env_base.py:679-705) for a batch."""
    B = env.num_envs
    if len(states) != B:
        raise ValueError("one state per env")
    env.set_env_state({k: np.stack([np.asarray(s[k], np.float32).reshape(-1) for s in states]) for k in STATE_KEYS})
