"""GPU tests (pytest -m gpu) for the muscle-condition variants (register_env_with_variants,
envs/myo/myobase/__init__.py:14-48): the control the step kernel's action-map stage produces (MYO_F_CTRL) and the resulting
physics against oracle/cond_ref.py (numpy restatement of base_v0.py:83-109 / fatigue.py) and the f64 oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_variant_ids_registered():
    from myosuite_mjx_amd.envs import REGISTRY
    for k in ("myoSarcHandPoseRandom-v0", "myoFatiHandPoseRandom-v0", "myoReafHandPoseRandom-v0", "myoFatiLegWalk-v0", "myoSarcFingerPoseFixed-v0"):
        assert k in REGISTRY
    assert "myoReafLegWalk-v0" not in REGISTRY and "myoReafFingerPoseFixed-v0" not in REGISTRY


def test_fatigue_action_map_matches_reference(hand):
    from myosuite_mjx_amd import capi, envs
    from oracle.cond_ref import Fatigue3CCr, sigmoid_map
    B, K = 8, 40
    env = envs.make("myoFatiHandPoseFixed-v0", num_envs=B, as_torch=False, autoreset=False)
    env.reset(seed=0)
    dyn = np.asarray(hand.actuator_dynprm).reshape(hand.nu, -1)
    ref = Fatigue3CCr(dyn[:, 0], dyn[:, 1], env.dt, (B, hand.nu))
    rng = np.random.default_rng(0)
    for k in range(K):
        a = rng.uniform(-1, 1, (B, hand.nu)).astype(np.float32)
        if k > 20:
            a[:] = -1.0                                  # rest phase: MA must decay through the LR branch
        env.step(a)
        want = ref.compute_act(sigmoid_map(a))
        got = env.batch.read(capi.F_CTRL)
        assert np.abs(got - want).max() < 2e-6, k
    fat = env.batch.read(capi.F_FATIGUE).reshape(B, 3, hand.nu)
    assert np.abs(fat[:, 0] - ref.MA).max() < 2e-6 and np.abs(fat[:, 1] - ref.MR).max() < 2e-6 and np.abs(fat[:, 2] - ref.MF).max() < 1e-6
    assert ref.MF.max() > 1e-4                          # fatigue did accumulate
    assert np.allclose(fat.sum(1), 1.0, atol=1e-5)       # the three compartments always add up to one
    env.reset(seed=0)
    fat = env.batch.read(capi.F_FATIGUE).reshape(B, 3, hand.nu)
    assert not fat[:, 0].any() and (fat[:, 1] == 1).all() and not fat[:, 2].any()


def test_reafferentation_redirects_eip_to_epl(hand):
    from myosuite_mjx_amd import capi, envs
    from oracle.cond_ref import reafferentation_map, sigmoid_map
    env = envs.make("myoReafHandPoseFixed-v0", num_envs=4, as_torch=False, autoreset=False)
    env.reset(seed=0)
    a = np.random.default_rng(1).uniform(-1, 1, (4, hand.nu)).astype(np.float32)
    env.step(a)
    epl, eip = hand.name2id("actuator", "EPL"), hand.name2id("actuator", "EIP")
    want = reafferentation_map(sigmoid_map(a), epl, eip)
    got = env.batch.read(capi.F_CTRL)
    assert np.abs(got - want).max() < 1e-6 and (got[:, eip] == 0).all()


def test_sarcopenia_halves_active_force_and_matches_oracle(hand):
    from myosuite_mjx_amd import capi, envs
    from oracle.cond_ref import sigmoid_map
    from oracle.oracle import Oracle
    B = 16
    weak = hand.with_sarcopenia()
    assert np.allclose(np.asarray(weak.actuator_gainprm).reshape(hand.nu, -1)[:, 2], 0.5 * np.asarray(hand.actuator_gainprm).reshape(hand.nu, -1)[:, 2])
    assert np.allclose(weak.actuator_biasprm, hand.actuator_biasprm)
    env = envs.make("myoSarcHandPoseFixed-v0", num_envs=B, as_torch=False, autoreset=False)
    env.reset(seed=0)
    rng = np.random.default_rng(2)
    a = rng.uniform(-1, 1, (B, hand.nu)).astype(np.float32)
    st = env.get_env_state()
    env.step(a)
    post = env.get_env_state()
    o = Oracle(weak.blob())
    ctrl = sigmoid_map(a).astype(np.float32)
    for e in range(B):
        o.reset()
        o.set_state(qpos=st["qpos"][e], qvel=st["qvel"][e], act=st["act"][e], ctrl=ctrl[e])
        o.step(env.frame_skip)
        assert np.abs(o.field("qpos") - post["qpos"][e]).max() < 1e-4
        assert np.abs(o.field("qvel") - post["qvel"][e]).max() < 2e-2
    # and the healthy model moves differently from the same state and action
    o2 = Oracle(hand.blob())
    o2.reset(); o2.set_state(qpos=st["qpos"][0], qvel=st["qvel"][0], act=st["act"][0], ctrl=ctrl[0]); o2.step(env.frame_skip)
    assert np.abs(o2.field("qpos") - post["qpos"][0]).max() > 1e-3


def test_fatigue_reset_options(hand):
    """BaseV0's fatigue_reset_random / fatigue_reset_vec kwargs (base_v0.py:30-31,120-127 -> CumulativeFatigue.reset, fatigue.py:114-134):
    random: MA = u1 u2, MR = u1 (1 - u2), MF = 1 - u1 per muscle (compartments sum to 1, MF ~ U(0,1)); vector: MF = vec, MR = 1 - vec, MA = 0;
    default: all rested."""
    import myosuite_mjx_amd as myo
    from myosuite_mjx_amd import capi
    nu = hand.nu
    env = myo.make("myoFatiHandPoseFixed-v0", num_envs=512, fatigue_reset_random=True, as_torch=False)
    env.reset(seed=3)
    F = env.batch.read(capi.F_FATIGUE).reshape(512, 3, nu)
    MA, MR, MF = F[:, 0], F[:, 1], F[:, 2]
    assert np.abs(MA + MR + MF - 1).max() < 1e-6 and MA.min() >= 0 and MR.min() >= 0 and MF.min() >= 0
    assert abs(MF.mean() - 0.5) < 0.01 and abs(MF.std() - 12 ** -0.5) < 0.01            # MF = 1 - u1 ~ U(0,1)
    assert abs(MA.mean() - 0.25) < 0.01 and abs((MA / (MA + MR)).mean() - 0.5) < 0.01   # MA = u1 u2, the active share u2 ~ U(0,1)
    vec = np.linspace(0.1, 0.9, nu).astype(np.float32)
    env2 = myo.make("myoFatiHandPoseFixed-v0", num_envs=4, fatigue_reset_vec=vec, as_torch=False)
    env2.reset(seed=0)
    F2 = env2.batch.read(capi.F_FATIGUE).reshape(4, 3, nu)
    assert np.allclose(F2[:, 2], vec) and np.allclose(F2[:, 1], 1 - vec) and not F2[:, 0].any()
    env3 = myo.make("myoFatiHandPoseFixed-v0", num_envs=4, as_torch=False)
    env3.reset(seed=0)
    F3 = env3.batch.read(capi.F_FATIGUE).reshape(4, 3, nu)
    assert not F3[:, 0].any() and (F3[:, 1] == 1).all() and not F3[:, 2].any()
    with pytest.raises(AssertionError):
        myo.make("myoFatiHandPoseFixed-v0", num_envs=4, fatigue_reset_vec=vec[:5])
    with pytest.raises(TypeError):
        myo.make("myoFatiHandPoseFixed-v0", num_envs=4, no_such_kwarg=1)
