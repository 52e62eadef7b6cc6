"""GPU tests (pytest -m gpu) for myoLegWalk-v0 (BASELINE.json config 4): the batched env -- sigmoid action map, 10 fused
substeps and the fused observation / reward pass of the wave kernel -- against the f64 oracle stepping the same states and
oracle/walk_ref.py (numpy restatement of walk_v0.py's get_obs_dict / get_reward_dict).  Tolerances: observation entries
1e-4 absolute on O(1) quantities (muscle force / 1000 entries: 2e-3 because forces are O(1e3) N), reward 5e-3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sigmoid(a):
    return 1.0 / (1.0 + np.exp(-5.0 * (a - 0.5)))


def _obs_tol(m):
    tol = np.full((m.nq - 2) + m.nv + 16 + 4 * m.nu, 1e-4)
    base = (m.nq - 2) + m.nv + 16
    tol[base + m.nu:base + 2 * m.nu] = 2e-3      # clipped actuator velocity (O(1..10) m/s, tendon jacobian times qvel)
    tol[base + 2 * m.nu:base + 3 * m.nu] = 2e-3  # force / 1000
    return tol


def test_walk_reset_and_rollout_match_oracle(legs, legoracle64):
    from myosuite_mjx_amd import capi, envs
    from oracle.walk_ref import walk_obs_reward
    m, o = legs, legoracle64
    B, K = 16, 12
    env = envs.make("myoLegWalk-v0", num_envs=B, as_torch=False, autoreset=False)
    assert env.obs_dim == 403 and env.max_episode_steps == 1000 and abs(env.dt - 0.01) < 1e-9
    kq = np.asarray(m.key_qpos).reshape(-1, m.nq)
    kv = np.asarray(m.key_qvel).reshape(-1, m.nv)
    obs = np.asarray(env.reset(seed=3))
    o.reset(); o.switches(0, 0, 0)
    o.set_state(qpos=kq[2], qvel=kv[2], act=np.zeros(m.nu), ctrl=np.zeros(m.nu), warm=np.zeros(m.nv), time=0)
    ref, _, _, _, _ = walk_obs_reward(m, o, 0, env.dt, kq[0][3:7])
    tol = _obs_tol(m)
    assert (np.abs(obs - ref[None, :]) < tol).all(), np.abs(obs - ref[None, :]).max()
    rng = np.random.default_rng(5)
    worst = dict(qpos=0.0, qvel=0.0, obs=0.0, rew=0.0)
    for k in range(K):
        st = env.get_env_state()
        warm = env.batch.read(capi.F_WARMSTART)
        a = rng.uniform(-1, 1, (B, m.nu)).astype(np.float32)
        obs, rew, done, trunc, info = env.step(a)
        obs, rew, done = np.asarray(obs), np.asarray(rew), np.asarray(done)
        post = env.get_env_state()
        assert (env.status() == 0).all()
        for e in range(B):
            # (1) the physics of this env step, from the HIP pre-step state
            o.reset()
            o.set_state(qpos=st["qpos"][e], qvel=st["qvel"][e], act=st["act"][e], ctrl=_sigmoid(a[e].astype(np.float64)).astype(np.float32),
                        warm=warm[e], time=float(st["time"][e, 0]))
            o.step(env.frame_skip)
            worst["qpos"] = max(worst["qpos"], np.abs(o.field("qpos") - post["qpos"][e]).max())
            worst["qvel"] = max(worst["qvel"], np.abs(o.field("qvel") - post["qvel"][e]).max())
            # (2) the observation / reward of the HIP post-step state (steps counter = k, walk_v0.py:334-337)
            o.set_state(qpos=post["qpos"][e], qvel=post["qvel"][e], act=post["act"][e])
            ref, dense, rdone, solved, _ = walk_obs_reward(m, o, k, env.dt, kq[0][3:7])
            d = np.abs(obs[e] - ref)
            assert (d < tol).all(), (k, e, int(np.argmax(d / tol)), d.max())
            worst["obs"] = max(worst["obs"], (d / tol).max())
            assert bool(done[e]) == bool(rdone)
            worst["rew"] = max(worst["rew"], abs(rew[e] - dense))
    assert worst["qpos"] < 1e-4 and worst["qvel"] < 2e-2, worst
    assert worst["rew"] < 5e-3, worst


def test_walk_done_autoreset_and_timelimit(legs):
    """Falling (COM below min_height) ends the episode; the env is reset in place to keyframe 2 and its row holds the new episode's first obs."""
    from myosuite_mjx_amd import capi, envs
    m = legs
    B = 8
    env = envs.make("myoLegWalk-v0", num_envs=B, as_torch=False)
    env.reset(seed=1)
    kq = np.asarray(m.key_qpos).reshape(-1, m.nq)
    st = env.get_env_state()
    st["qpos"][:4, 2] = 0.6                      # half of the envs start with the pelvis 0.4 m lower: COM < 0.8 -> done on the first step
    env.set_env_state(st)
    obs, rew, done, trunc, info = env.step(np.zeros((B, m.nu), np.float32))
    done = np.asarray(done)
    assert done[:4].all() and not done[4:].any()
    assert (np.asarray(rew)[:4] < -50).all()      # the -100 * done term
    post = env.get_env_state()
    assert np.allclose(post["qpos"][:4], kq[2], atol=1e-6)
    elapsed = env.batch.read(capi.F_ELAPSED)[:, 0]
    assert (elapsed[:4] == 0).all() and (elapsed[4:] == 1).all()
    obs = np.asarray(obs)
    assert np.allclose(obs[:4, 0], kq[2][2], atol=1e-6)            # first obs of the new episode: qpos[2] of keyframe 2
    assert np.allclose(obs[:4, (m.nq - 2) + m.nv + 15], 0.0)        # phase_var restarts
    assert np.allclose(obs[4:, (m.nq - 2) + m.nv + 15], 0.0)        # and is (steps=0)/100 for the first step's obs (walk_v0.py:334-337)
    obs2, *_ = env.step(np.zeros((B, m.nu), np.float32))
    assert np.allclose(np.asarray(obs2)[4:, (m.nq - 2) + m.nv + 15], 0.01, atol=1e-7)


def test_walk_torch_views_and_bench_rollout(legs):
    import torch
    from myosuite_mjx_amd import capi, envs
    env = envs.make("myoLegWalk-v0", num_envs=256)
    obs = env.reset(seed=0)
    assert obs.shape == (256, 403) and obs.is_cuda
    a = torch.rand((256, env.act_dim), device=obs.device) * 2 - 1
    for _ in range(3):
        obs, rew, done, trunc, info = env.step(a)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    ms = env.batch.bench_rollout(5, env.frame_skip, 0, capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET, env.max_episode_steps,
                                 torch.cuda.current_stream().cuda_stream)
    assert ms > 0
    assert torch.isfinite(env.view(capi.F_OBS)).all()


def test_leg_stand_env(legs):
    """myoLegStandRandom-v0 (walk_v0.py:13-183 ReachEnvV0 on the legs): reset = keyframe 0 + U(-0.2, 0.2) on each joint's first coordinate,
    clipped to the joint range (root x clipped to its (0, 0) "range"); target = pelvis site of that family of poses + U(+-0.05, +-0.05, 0);
    obs 155 = qpos 35, qvel*dt 34, tip 3, err 3, act 80; reward 10 - d - 10 |qvel dt| + 4 bonus - 100 |act|/80 - 50 (d > 0.44, after 2 dt)."""
    import torch
    import myosuite_mjx_amd as myo
    m = legs
    B = 64
    env = myo.make("myoLegStandRandom-v0", num_envs=B, seed=5, autoreset=False)
    obs = env.reset(seed=5)
    assert obs.shape == (B, 155) and env.max_episode_steps == 150 and abs(env.dt - 0.01) < 1e-9
    st = env.get_env_state()
    k0 = np.asarray(m.key_qpos).reshape(-1, m.nq)[0]
    adr = np.asarray(m.jnt_qposadr)
    dq = st["qpos"] - k0[None, :]
    other = np.setdiff1d(np.arange(m.nq), adr)
    assert np.allclose(dq[:, other], 0, atol=1e-6) and np.allclose(st["qpos"][:, 0], 0, atol=1e-7)      # only first coordinates move; root x -> 0
    lo, hi = m.jnt_range[1:, 0], m.jnt_range[1:, 1]
    q1 = st["qpos"][:, adr[1:]]
    assert (q1 >= lo - 1e-6).all() and (q1 <= hi + 1e-6).all() and (np.abs(dq[:, adr[1:]]) <= 0.2 + 1e-6).all() and dq[:, adr[1:]].std() > 0.05
    tg = st["target"]
    assert np.allclose(tg[:, 2], 0.92, atol=1e-6) and (np.abs(tg[:, :2]) <= 0.05 + 1e-6).all() and tg[:, :2].std() > 0.02
    g = torch.Generator(device="cuda").manual_seed(0)
    for k in range(6):
        obs, rwd, term, trunc, info = env.step(torch.rand((B, 80), device="cuda", generator=g) * 2 - 1)
    st = env.get_env_state()
    o = obs.cpu().numpy()
    assert np.allclose(o[:, :35], st["qpos"], atol=1e-7) and np.allclose(o[:, 35:69], st["qvel"] * 0.01, atol=1e-6)
    assert np.allclose(o[:, 69:72], st["qpos"][:, :3], atol=1e-6) and np.allclose(o[:, 72:75], st["target"] - st["qpos"][:, :3], atol=1e-6)
    assert np.allclose(o[:, 75:], st["act"], atol=1e-7)
    d = np.linalg.norm(o[:, 72:75], axis=1)
    ref = (10.0 - d - 10.0 * np.linalg.norm(o[:, 35:69], axis=1)) + 4.0 * ((d < 0.1) * 1.0 + (d < 0.05) * 1.0) \
        - 100.0 * np.linalg.norm(st["act"], axis=1) / 80 - 50.0 * (d > 0.44)
    edge = (np.abs(d - 0.05) < 1e-5) | (np.abs(d - 0.1) < 1e-5) | (np.abs(d - 0.44) < 1e-5)
    assert np.allclose(rwd.cpu().numpy()[~edge], ref[~edge], atol=2e-4) and np.array_equal(term.cpu().numpy()[~edge], (d > 0.44)[~edge])
    assert (env.status() == 0).all()


def test_walk_reset_type_random(legs):
    """WalkEnvV0.get_randomized_initial_state (walk_v0.py:316-332): keyframe 2 or 3 with probability 1/2 each, qpos += N(0, 0.02) on every
    coordinate except the root height and quaternion, qvel of the chosen keyframe."""
    import myosuite_mjx_amd as myo
    from myosuite_mjx_amd import capi
    B = 2048
    env = myo.make("myoLegWalk-v0", num_envs=B, reset_type="random", as_torch=False)
    env.reset(seed=11)
    q, v = env.batch.read(capi.F_QPOS), env.batch.read(capi.F_QVEL)
    kq, kv = np.asarray(legs.key_qpos).reshape(-1, legs.nq), np.asarray(legs.key_qvel).reshape(-1, legs.nv)
    is3 = np.abs(v - kv[3]).max(1) < 1e-6
    is2 = np.abs(v - kv[2]).max(1) < 1e-6
    assert (is2 | is3).all() and abs(is3.mean() - 0.5) < 0.04                   # one of the two keyframes, half each
    base = np.where(is3[:, None], kq[3], kq[2])
    d = q - base
    assert np.abs(d[:, 2:7]).max() < 1e-6                                       # height and orientation untouched
    noisy = np.r_[0, 1, 7:legs.nq]
    assert abs(d[:, noisy].std() - 0.02) < 0.001 and abs(d[:, noisy].mean()) < 0.001
    obs, rew, term, trunc, info = env.step(np.zeros((B, env.act_dim), np.float32))
    assert np.isfinite(obs).all() and (env.status() == 0).all()
