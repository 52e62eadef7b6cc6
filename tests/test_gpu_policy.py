"""GPU tests (pytest -m gpu) for the policy-inference drop-in: the HIP kernel vs the float64 numpy
statement of the same network, on the tensor shapes of the reference's `mjx_brax_policy` artefact (obs 2 -> 4 x 32 -> 12) and on
the hand env's shapes (obs 108 -> 4 x 32 -> 78); then a closed loop env.step(policy(obs)) that never leaves the device.
Tolerance 2e-5 on actions in [-1, 1] (float32 accumulation over <= 108 inputs).  brax itself is absent: parity unpinned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_policy(rng, obs_dim, act_dim, hidden=(32, 32, 32, 32)):
    sizes = [obs_dim, *hidden, 2 * act_dim]
    ks = [rng.normal(0, 1.0 / np.sqrt(sizes[i]), (sizes[i], sizes[i + 1])).astype(np.float32) for i in range(len(sizes) - 1)]
    bs = [rng.normal(0, 0.1, sizes[i + 1]).astype(np.float32) for i in range(len(sizes) - 1)]
    return rng.normal(0, 1, obs_dim).astype(np.float32), rng.uniform(0.5, 2.0, obs_dim).astype(np.float32), ks, bs


@pytest.mark.parametrize("obs_dim,act_dim,B", [(2, 6, 1000), (108, 39, 4096), (403, 80, 257)])
def test_policy_matches_numpy_reference(obs_dim, act_dim, B):
    import torch
    from myosuite_mjx_amd.policy import BraxPolicy, reference_forward
    rng = np.random.default_rng(obs_dim)
    mean, std, ks, bs = _random_policy(rng, obs_dim, act_dim)
    pol = BraxPolicy(mean, std, ks, bs)
    obs = torch.as_tensor(rng.normal(0, 2, (B, obs_dim)).astype(np.float32), device="cuda")
    act = torch.empty((B, act_dim), dtype=torch.float32, device="cuda")
    pol.act(obs.data_ptr(), B, act.data_ptr(), deterministic=True, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref, loc, scale = reference_forward(obs.cpu().numpy(), mean, std, ks, bs)
    assert np.abs(act.cpu().numpy() - ref).max() < 2e-5
    # sampled actions: atanh(a) - loc must be N(0, scale^2); reproducible for a fixed (seed, step); different across steps
    s1 = torch.empty_like(act); s2 = torch.empty_like(act); s3 = torch.empty_like(act)
    st = torch.cuda.current_stream().cuda_stream
    pol.act(obs.data_ptr(), B, s1.data_ptr(), deterministic=False, seed=7, step=3, stream=st)
    pol.act(obs.data_ptr(), B, s2.data_ptr(), deterministic=False, seed=7, step=3, stream=st)
    pol.act(obs.data_ptr(), B, s3.data_ptr(), deterministic=False, seed=7, step=4, stream=st)
    torch.cuda.synchronize()
    assert torch.equal(s1, s2) and not torch.equal(s1, s3)
    a = s1.cpu().numpy().astype(np.float64)
    ok = np.abs(a) < 0.999
    z = ((np.arctanh(np.clip(a, -0.999999, 0.999999)) - loc) / scale)[ok]
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.03


def test_closed_loop_rollout_stays_on_device():
    import torch
    import myosuite_mjx_amd as myo
    from myosuite_mjx_amd import capi
    from myosuite_mjx_amd.policy import BraxPolicy
    env = myo.make("myoHandPoseFixed-v0", num_envs=512)
    rng = np.random.default_rng(0)
    mean, std, ks, bs = _random_policy(rng, env.obs_dim, env.act_dim)
    pol = BraxPolicy(mean, std, ks, bs)
    obs = env.reset(seed=0)
    act = torch.empty((512, env.act_dim), dtype=torch.float32, device=obs.device)
    st = torch.cuda.current_stream().cuda_stream
    tot = torch.zeros(512, device=obs.device)
    for k in range(20):
        pol.act(obs.data_ptr(), 512, act.data_ptr(), deterministic=False, seed=1, step=k, stream=st)
        obs, rew, done, trunc, info = env.step(act)
        tot += rew
    torch.cuda.synchronize()
    assert torch.isfinite(tot).all() and torch.isfinite(obs).all()
    assert (env.status() == 0).all()


def test_the_real_mjx_brax_policy_weights():
    """the reference's own artefact `mjx_brax_policy` (obs 2 -> 4 x 32 -> 12, observation statistics over 102 400 samples),
    its tensors extracted WITHOUT unpickling (tools/extract_brax_policy.py: symbolic opcode walk, fixture tests/golden/mjx_brax_policy.npz),
    through myo_policy_act vs the float64 statement of the brax network; then the closed loop it was trained for, obs = [qpos, qvel] of the
    one-dof six-muscle elbow -> six muscle excitations, on myoElbowPose1D6MRandom-v0 without leaving the device."""
    import os
    import torch
    import myosuite_mjx_amd as myo
    from myosuite_mjx_amd import capi
    from myosuite_mjx_amd.policy import BraxPolicy, reference_forward
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mjx_brax_policy.npz")
    z = np.load(path)
    assert z["w0"].shape == (2, 32) and z["w4"].shape == (32, 12) and float(z["obs_count"]) == 102400.0
    pol = BraxPolicy.from_npz(path)
    assert (pol.obs_dim, pol.act_dim) == (2, 6)
    rng = np.random.default_rng(0)
    B = 4096
    obs_np = (z["obs_mean"] + 3 * z["obs_std"] * rng.normal(0, 1, (B, 2))).astype(np.float32)
    obs = torch.as_tensor(obs_np, device="cuda")
    act = torch.empty((B, 6), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    pol.act(obs.data_ptr(), B, act.data_ptr(), deterministic=True, stream=st)
    torch.cuda.synchronize()
    ks, bs = [z[f"w{i}"] for i in range(5)], [z[f"b{i}"] for i in range(5)]
    ref, loc, scale = reference_forward(obs_np, z["obs_mean"], z["obs_std"], ks, bs)
    assert np.abs(act.cpu().numpy() - ref).max() < 2e-5 and np.ptp(ref) > 0.5          # a trained, non-trivial map
    env = myo.make("myoElbowPose1D6MRandom-v0", num_envs=256)
    assert env.act_dim == 6 and env.mjmodel.nq == 1
    env.reset(seed=0)
    a = torch.empty((256, 6), dtype=torch.float32, device="cuda")
    lo, hi = float(env.mjmodel.jnt_range[0, 0]), float(env.mjmodel.jnt_range[0, 1])
    for k in range(60):
        o2 = torch.cat((env.view(capi.F_QPOS), env.view(capi.F_QVEL)), 1).contiguous()   # the policy's observation: [qpos, qvel]
        pol.act(o2.data_ptr(), 256, a.data_ptr(), deterministic=False, seed=3, step=k, stream=st)
        obs_env, rew, done, trunc, info = env.step(a)
        assert float(a.abs().max()) <= 1.0
    torch.cuda.synchronize()
    q = env.view(capi.F_QPOS)[:, 0]
    assert torch.isfinite(q).all() and float(q.min()) > lo - 0.1 and float(q.max()) < hi + 0.1 and (env.status() == 0).all()
