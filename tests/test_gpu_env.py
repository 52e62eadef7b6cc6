"""GPU env-level tests (pytest -m gpu): gym-style API contract in the shape of the reference's tests/test_envs.py
(spaces, step returns, determinism atol 1e-5 there / bit-exact here), obs/reward formulas, auto-reset, sharding."""
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _make(env_id, n, **kw):
    import myosuite_mjx_amd as myo
    return myo.make(env_id, num_envs=n, **kw)


def test_pose_env_contract_and_formulas(hand):
    import torch
    env = _make("myoHandPoseRandom-v0", 64, seed=3)
    obs = env.reset(seed=3)
    assert obs.shape == (64, 108) and obs.dtype == torch.float32 and obs.is_cuda      # Appendix C layout
    st = env.get_env_state()
    lo, hi = hand.jnt_range[:, 0], hand.jnt_range[:, 1]
    assert (st["qpos"] >= lo - 1e-6).all() and (st["qpos"] <= hi + 1e-6).all() and st["qpos"].std(0).min() > 0.005 * 0 + 0  # U(jnt_range)
    tl, th = env.spec["target_lo"], env.spec["target_hi"]
    assert (st["target"] >= tl - 1e-6).all() and (st["target"] <= th + 1e-6).all()
    o = obs.cpu().numpy()
    assert np.allclose(o[:, :23], st["qpos"]) and np.allclose(o[:, 23:46], 0) and np.allclose(o[:, 46:69], st["target"] - st["qpos"], atol=1e-6)
    assert np.allclose(o[:, 69:], 0)
    a = torch.rand((64, 39), device=obs.device) * 2 - 1
    obs2, rwd, term, trunc, info = env.step(a)
    st = env.get_env_state()
    o = obs2.cpu().numpy()
    assert np.allclose(o[:, :23], st["qpos"], atol=1e-7) and np.allclose(o[:, 23:46], st["qvel"] * 0.02, atol=1e-7)
    assert np.allclose(o[:, 69:], st["act"], atol=1e-7) and np.allclose(st["time"], 0.02, atol=1e-6)
    # pose_v0.py:111-138
    dist = np.linalg.norm(o[:, 46:69], axis=1)
    actm = np.linalg.norm(st["act"], axis=1) / 39
    ref = -dist + 4.0 * ((dist < 0.7) * 1.0 + (dist < 1.05) * 1.0) - actm - 50.0 * (dist > 2 * np.pi)
    assert np.allclose(rwd.cpu().numpy(), ref, atol=1e-5)
    assert term.dtype == torch.bool and not term.any() and not trunc.any()
    assert (env.status() == 0).all()


def test_action_map_is_muscle_sigmoid(hand):
    import torch
    env = _make("myoHandPoseFixed-v0", 4)
    env.reset()
    a = torch.tensor(np.linspace(-2, 2, 4 * 39).reshape(4, 39), dtype=torch.float32, device="cuda")
    env.step(a)
    from myosuite_mjx_amd import capi
    ctrl = env.batch.read(capi.F_CTRL)
    ac = np.clip(a.cpu().numpy(), -1, 1)                                            # env_base.py:341
    assert np.allclose(ctrl, 1.0 / (1.0 + np.exp(-5.0 * (ac - 0.5))), atol=1e-6)      # base_v0.py:89-91


def test_determinism_and_lane_independence():
    import torch
    e1 = _make("myoHandPoseRandom-v0", 256, seed=5)
    e2 = _make("myoHandPoseRandom-v0", 256, seed=5)
    o1, o2 = e1.reset(seed=5), e2.reset(seed=5)
    assert torch.equal(o1, o2)
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(5):
        a = torch.rand((256, 39), device="cuda", generator=g) * 2 - 1
        r1 = e1.step(a)
        r2 = e2.step(a)
        assert torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])              # bit-exact run to run
    # an env's trajectory does not depend on its position in the batch / its wave-mates
    e3 = _make("myoHandPoseRandom-v0", 256, seed=5, autoreset=False)
    e4 = _make("myoHandPoseRandom-v0", 256, seed=5, autoreset=False)
    e3.reset(seed=5); e4.reset(seed=5)
    st = e3.get_env_state()
    perm = np.random.default_rng(0).permutation(256)
    e4.set_env_state({k: v[perm] for k, v in st.items()})
    a = torch.rand((256, 39), device="cuda", generator=g) * 2 - 1
    o3 = e3.step(a)[0]
    o4 = e4.step(a[torch.as_tensor(perm, device="cuda")])[0]
    assert torch.equal(o3[torch.as_tensor(perm, device="cuda")], o4)


def test_timelimit_autoreset():
    import torch
    from myosuite_mjx_amd import capi
    env = _make("myoHandPoseFixed-v0", 8, seed=1)
    env.reset(seed=1)
    a = torch.zeros((8, 39), device="cuda")
    for k in range(100):
        obs, rwd, term, trunc, info = env.step(a)
        if k < 99:
            assert not trunc.any()
    assert trunc.all() and not term.any()                                            # gym TimeLimit at 100 steps
    assert (env.batch.read(capi.F_ELAPSED) == 0).all() and np.allclose(env.get_env_state()["time"], 0)
    assert np.allclose(obs.cpu().numpy()[:, 23:46], 0)                                # first obs of the new episode


def test_reach_env_against_oracle_sites(hand, oracle64):
    import torch
    env = _make("myoHandReachRandom-v0", 16, seed=2, autoreset=False)
    obs = env.reset(seed=2)
    assert obs.shape == (16, 115)
    a = torch.rand((16, 39), device="cuda") * 2 - 1
    obs, rwd, term, trunc, info = env.step(a)
    st = env.get_env_state()
    o = obs.cpu().numpy()
    tips = [hand.name2id("site", t) for t in ("THtip", "IFtip", "MFtip", "RFtip", "LFtip")]
    for e in range(16):
        oracle64.reset()
        oracle64.set_state(qpos=st["qpos"][e])
        oracle64.fwd_position()
        sx = oracle64.field("site_xpos").reshape(-1, 3)[tips].ravel()
        assert np.abs(o[e, 46:61] - sx).max() < 2e-6                                  # tip_pos, world coordinates
        assert np.abs(o[e, 61:76] - (st["target"][e] - sx)).max() < 2e-6              # reach_err
    dist = np.linalg.norm(o[:, 61:76], axis=1)
    near, far = 0.0125 * 5, 0.034 * 5
    ref = -dist + 4.0 * ((dist < 2 * near) * 1.0 + (dist < near) * 1.0) - 50.0 * 0    # t = 0.02 <= 2*dt: far_th = inf (reach_v0.py:118-122)
    assert np.allclose(rwd.cpu().numpy(), ref, atol=1e-5) and not term.any()
    tl, th = env.spec["target_lo"], env.spec["target_hi"]
    assert (st["target"] >= tl - 1e-6).all() and (st["target"] <= th + 1e-6).all()


def test_sharding_invariance_full_batch():
    """B = 4096 as one batch == two half batches with env_offset (global env id keyed RNG): what bench.py relies on."""
    from myosuite_mjx_amd import capi
    full = _make("myoHandPoseRandom-v0", 4096, seed=9, as_torch=False)
    h0 = _make("myoHandPoseRandom-v0", 2048, seed=9, as_torch=False, env_offset=0)
    h1 = _make("myoHandPoseRandom-v0", 2048, seed=9, as_torch=False, env_offset=2048)
    mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
    for e in (full, h0, h1):
        e.reset(seed=9)
        e.batch.bench_rollout(3, 10, 9, mode, 100, None)
    q = full.batch.read(capi.F_QPOS)
    assert np.array_equal(q[:2048], h0.batch.read(capi.F_QPOS)) and np.array_equal(q[2048:], h1.batch.read(capi.F_QPOS))
    o = full.batch.read(capi.F_OBS)
    assert np.array_equal(o[2048:], h1.batch.read(capi.F_OBS)) and np.isfinite(o).all()
    assert (full.status() & 3 == 0).all()


def test_functional_mjx_style_api(hand):
    import myosuite_mjx_amd as myo
    m = myo.put_model("myohand_pose")
    d = myo.make_data(m, 4)
    myo.step(m, d, np.full((4, 39), 0.3, np.float32), nsubsteps=5)
    assert np.allclose(myo.get(d, "time"), 5 * 0.002, atol=1e-7) and myo.get(d, "qpos").shape == (4, 23)
    assert (myo.get(d, "act") > 0).all()


@pytest.mark.parametrize("as_torch", [True, False])
def test_hip_sim_scene_mirrors_the_backend_abc(hand, oracle64, as_torch, tmp_path):
    """HipSimScene = batched `SimScene` (physics/sim_scene.py:38-209): advance / forward / reset / get_state / set_state,
    checked like Robot.step drives it (data.ctrl[:] = ...; sim.advance(n_frames)) against the oracle -- with device-resident `data`
    (zero-copy torch views: advance is only the launch) and with host mirrors; plus the rest of the ABC (copy_model, save_binary,
    get_mjlib / get_handle, disable_option_context, model.*_name2id of mj_sim_scene.py:110-163)."""
    import torch
    import myosuite_mjx_amd as myo
    B = 6
    sim = myo.HipSimScene("myohand_pose", num_envs=B, as_torch=as_torch)
    assert abs(sim.step_duration - 0.002) < 1e-9 and sim.init_qpos.shape == (23,) and tuple(sim.data.qpos.shape) == (B, 23)
    npy = (lambda t: t.cpu().numpy()) if as_torch else (lambda a: a)
    rng = np.random.default_rng(0)
    lo, hi = hand.jnt_range[:, 0], hand.jnt_range[:, 1]
    q = (0.5 * (lo + hi) + 0.4 * (hi - lo) * rng.uniform(-1, 1, (B, 23))).astype(np.float32)
    v = rng.normal(0, 0.3, (B, 23)).astype(np.float32)
    a = rng.uniform(0, 1, (B, 39)).astype(np.float32)
    sim.set_state(time=np.zeros((B, 1)), qpos=q, qvel=v, act=a)
    ctrl = rng.uniform(0, 1, (B, 39)).astype(np.float32)
    sim.data.ctrl[:] = torch.as_tensor(ctrl, device="cuda") if as_torch else ctrl
    if as_torch:
        p0 = sim.data.qpos.data_ptr()
    sim.advance(substeps=10)
    assert (sim.status() == 0).all() and np.allclose(npy(sim.data.time), 0.02, atol=1e-6)
    if as_torch:
        assert sim.data.qpos.data_ptr() == p0 and sim.data.qpos.is_cuda          # the same device buffer, no copies around the step
    st = sim.get_state()
    for e in range(B):
        oracle64.reset()
        oracle64.set_state(qpos=q[e], qvel=v[e], act=a[e], ctrl=ctrl[e])
        oracle64.step(10)
        assert np.abs(oracle64.field("qpos") - npy(st["qpos"])[e]).max() < 1e-4
        assert np.abs(oracle64.field("qvel") - npy(st["qvel"])[e]).max() < 2e-2
    # disable_option_context(limits / contact) switches the model for the block only
    sim.set_state(qpos=np.tile(hi + 0.05, (B, 1)), qvel=np.zeros((B, 23)))
    with sim.disable_option_context(limits=True, contact=True):
        sim.advance(1)
        free = npy(sim.data.qacc).copy()
    sim.set_state(qpos=np.tile(hi + 0.05, (B, 1)), qvel=np.zeros((B, 23)), time=np.zeros((B, 1)))
    sim.advance(1)
    assert np.abs(npy(sim.data.qacc) - free).max() > 10.0                        # with limits on, the violated joints are pushed back
    sim.reset()
    assert np.allclose(npy(sim.data.qpos), hand.qpos0) and not npy(sim.data.qvel).any() and not npy(sim.data.time).any()
    sim.advance(1)
    assert np.isfinite(npy(sim.data.qpos)).all()
    # the rest of the ABC
    m2 = sim.copy_model()
    assert m2 is not sim.model and np.array_equal(m2.jnt_range, sim.model.jnt_range) and m2.blob() == sim.model.blob()
    path = sim.save_binary(str(tmp_path / "hand.mjb"))
    assert path.endswith(".myob") and open(path, "rb").read() == sim.model.blob()
    assert sim.get_mjlib() is sim.lib and hasattr(sim.lib, "myo_step") and sim.get_handle(sim._batch) == sim._batch.h
    assert sim.model.joint_name2id("mcp2_flexion") == hand.name2id("joint", "mcp2_flexion") and sim.model.site_name2id("IFtip") >= 0
    assert sim.model.actuator_name2id("FDS2") == hand.name2id("actuator", "FDS2") and sim.model.body_name2id("lunate") > 0
    with pytest.raises(ValueError):
        sim.model.camera_name2id("hand_side_inter")
    sim.renderer.refresh_window()
    sim.close()


def test_default_sim_scene_is_driven_with_numpy_like_robot_step(hand, oracle64):
    """the default-constructed HipSimScene (device-resident data) must take the reference's own driving code unchanged --
    Robot.step writes NumPy controls into `sim.data.ctrl` and calls `sim.advance` (robot/robot.py:880-882) -- and must not lose the report
    of an in-kernel reset (`last_flags`, the counterpart of DMSimScene.advance's exception path, mj_sim_scene.py:54-61)."""
    import myosuite_mjx_amd as myo
    B = 4
    sim = myo.HipSimScene("myohand_pose", num_envs=B)
    rng = np.random.default_rng(2)
    ctrl = rng.uniform(0, 1, (B, 39))                      # float64 NumPy, as the reference's robot hands it over
    sim.data.ctrl[:] = ctrl
    sim.data.ctrl[0, :3] = np.array([0.25, 0.5, 0.75])
    ctrl[0, :3] = [0.25, 0.5, 0.75]
    sim.advance(substeps=10)
    assert (sim.last_flags == 0).all() and np.allclose(sim.data.ctrl.cpu().numpy(), ctrl.astype(np.float32))
    oracle64.reset(); oracle64.set_state(ctrl=ctrl[1]); oracle64.step(10)
    assert np.abs(oracle64.field("qpos") - sim.data.qpos[1].cpu().numpy()).max() < 1e-4
    q = sim.data.qpos.cpu().numpy(); q[2, 5] = np.nan
    sim.data.qpos[:] = q                                    # a NaN state: that env is reset inside the kernel ...
    sim.advance(substeps=2)
    fl = sim.last_flags                                     # ... and the reset is reported without an explicit status() call
    assert fl[2] & 1 and (fl[[0, 1, 3]] == 0).all() and np.isfinite(sim.data.qpos.cpu().numpy()).all()
    assert (sim.status()[2] & 1) and (sim.last_flags == 0).all()
    sim.close()


def test_every_registered_env_id_resets_steps_and_is_deterministic():
    """Counterpart of the reference's tests/test_myo.py -> test_envs.py loop over every registered id: construct, seed, reset, one
    small-action step (a = 0.01 * U like test_envs.py:61-64), shapes and finiteness, and the same seed twice gives the same obs / reward."""
    import torch
    from myosuite_mjx_amd import envs
    ids = sorted(envs.REGISTRY)
    assert len(ids) >= 40 and "myoLegWalk-v0" in ids and "myoFatiHandReachRandom-v0" in ids
    for k, env_id in enumerate(ids):
        outs = []
        for rep in range(2):
            try:
                env = envs.make(env_id, num_envs=8, seed=1234)
            except FileNotFoundError:          # MyoDM motion-tracking ids need the reference's motion file (not redistributed): two of them travel as fixtures
                f = np.load(os.path.join(ROOT, "tests", "golden", "ref_motion.npz"))
                stem = envs.REGISTRY[env_id]["motion"][:-4]
                motion = {k.split("__in__")[1]: f[k] for k in f.files if k.startswith(f"track_{stem}__in__")}
                if not motion:
                    break
                env = envs.make(env_id, num_envs=8, seed=1234, reference=motion)
            obs0 = env.reset(seed=1234).clone()
            g = torch.Generator(device="cuda").manual_seed(k)
            a = 0.01 * torch.rand((8, env.act_dim), device="cuda", generator=g)
            obs, rew, term, trunc, info = env.step(a)
            torch.cuda.synchronize()
            assert obs.shape == (8, env.obs_dim) and obs0.shape == obs.shape and rew.shape == (8,), env_id
            assert torch.isfinite(obs).all() and torch.isfinite(rew).all(), env_id
            assert (env.status() == 0).all(), env_id
            outs.append((obs0, obs.clone(), rew.clone()))
        if len(outs) < 2:
            continue
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), env_id


def test_async_bench_rollout_interleaves_with_stream_work():
    """bench.py's N > 1 loop: steps are enqueued without host waits and another stream operation (there: the RCCL obs all-gather,
    here: a device copy of the same buffer) sits between them; kernel times are collected once at the end."""
    import torch
    from myosuite_mjx_amd import capi
    e_sync = _make("myoHandPoseRandom-v0", 512, seed=3)
    e_async = _make("myoHandPoseRandom-v0", 512, seed=3)
    e_sync.reset(seed=3); e_async.reset(seed=3)
    st = torch.cuda.current_stream().cuda_stream
    mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
    ms = e_sync.batch.bench_rollout(12, 10, 0, mode, 100, st)
    k_sync = e_sync.batch.last_kernel_ms()
    obs = e_async.view(capi.F_OBS)
    sink = torch.empty((2 * 512, e_async.obs_dim), device=obs.device)
    for _ in range(12):
        e_async.batch.bench_rollout_async(1, 10, 0, mode, 100, st)
        sink[512:].copy_(obs, non_blocking=True)
    k_async = e_async.batch.last_kernel_ms()
    torch.cuda.synchronize()
    assert ms > 0 and k_sync > 0 and k_async > 0 and 0.5 < k_async / k_sync < 2.0
    assert torch.equal(e_sync.view(capi.F_OBS), e_async.view(capi.F_OBS))          # same steps, same results
    assert torch.equal(sink[512:], obs)
    assert e_async.batch.last_kernel_ms() == k_async                               # nothing pending: value is kept


def test_bench_multi_gpu_loop_with_a_stand_in_collective():
    """bench.py's N > 1 loop (async steps, staging copy, deferred wait on the previous gather) with the RCCL all-gather replaced by a
    side-stream device copy that has the same handle contract (.wait() = stream-side wait)."""
    import importlib.util, os, torch
    from myosuite_mjx_amd import capi
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    env = _make("myoHandPoseRandom-v0", 512, seed=5)
    ref = _make("myoHandPoseRandom-v0", 512, seed=5)
    env.reset(seed=5); ref.reset(seed=5)
    obs = env.view(capi.F_OBS)
    staging = torch.empty_like(obs)
    gathered = torch.zeros((2 * 512, env.obs_dim), device=obs.device)
    side = torch.cuda.Stream()

    class Handle:
        def __init__(self, ev): self.ev = ev
        def wait(self): torch.cuda.current_stream().wait_event(self.ev)

    def gather_async(src):
        ready = torch.cuda.Event(); ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            gathered[512:].copy_(src, non_blocking=True)
            done = torch.cuda.Event(); done.record()
        return Handle(done)

    st = torch.cuda.current_stream().cuda_stream
    mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
    ms, kms = bench.rollout_multi(torch, env, 15, mode, st, obs, staging, gather_async)
    ref.batch.bench_rollout(15, 10, 0, mode, 100, st)
    torch.cuda.synchronize()
    assert ms > 0 and 0 < kms <= ms * 1.05
    assert torch.equal(obs, ref.view(capi.F_OBS)) and torch.equal(gathered[512:], obs)


def test_bench_two_rank_rehearsal():
    """bench.py's N > 1 path end to end (launcher env, env_offset sharding, staging + async all-gather per step, barrier / max-over-ranks
    timing, rank-0 JSON line) with two processes on this one GPU over gloo (MYO_BENCH_REHEARSAL=1): a control-flow check, the real
    multi-GPU run uses RCCL on N devices."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MYO_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--batch", "512"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1                                            # rank 0 prints exactly one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 1024 and d["steps"] == 10 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 1024 * 10 / (d["ms_per_step"] * 10 / 1e3)) < 1e-6 * d["value"]
    assert d["roofline"]["kernel"].startswith("step_kernel_w<24,8,32,1,4,false") and "cpu_baseline" not in d and "rehearsal" in d
    assert d["allgather_ms_rank0"] > 0 and d["allgather_bytes_out"] == 1024 * 108 * 4          # gather time reported separately


def test_bench_one_rank_under_torchrun_matches_the_plain_run():
    """`python bench.py --gpus 1` and the driver's N > 1 launcher form with one process (torchrun --nproc-per-node 1) run
    the same code path and print the same line (workload, kernel, batch, contract keys); only the clocks differ."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--batch", "512", "--no-cpu-baseline"]
    plain = subprocess.run([sys.executable] + args, cwd=root, capture_output=True, text=True, timeout=300)
    tr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", "29519"] + args, cwd=root, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0 and tr.returncode == 0, plain.stderr[-1500:] + tr.stderr[-1500:]
    a, b = (json.loads([ln for ln in o.stdout.splitlines() if ln.startswith('{"metric"')][-1]) for o in (plain, tr))
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "dtype", "data", "vs_baseline", "flagged_envs"):
        assert a[k] == b[k], k
    assert a["config"]["workload"] == b["config"]["workload"] and a["config"]["global_batch"] == b["config"]["global_batch"] == 512
    assert a["roofline"]["kernel"] == b["roofline"]["kernel"] and set(a) == set(b) and "allgather_ms_rank0" not in b
    assert 0.5 < a["value"] / b["value"] < 2.0


def test_overlapped_gather_at_the_eight_gpu_buffer_size():
    """The N = 8 shape of bench.rollout_multi on one GPU: 4096 envs per rank, the gathered observation buffer of 8 x 4096 x 108 floats
    (14.2 MB, SURVEY 8d config 4 quotes 15.1 MB for the 115-wide reach observation) filled by a stand-in collective that writes this rank's
    slot on a side stream while the next step kernel runs; the loop never waits on the host and ends with the last step's observations."""
    import importlib.util, os, torch
    from myosuite_mjx_amd import capi
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    B, world, rank = 4096, 8, 5
    env = _make("myoHandPoseRandom-v0", B, seed=5)
    env.batch.set_env_offset(rank * B)
    env.reset(seed=5)
    obs = env.view(capi.F_OBS)
    staging = torch.empty_like(obs)
    gathered = torch.zeros((world * B, env.obs_dim), device=obs.device)
    assert gathered.numel() * 4 == 8 * 4096 * 108 * 4
    side = torch.cuda.Stream()

    class Handle:
        def __init__(self, ev): self.ev = ev
        def wait(self): torch.cuda.current_stream().wait_event(self.ev)

    def gather_async(src):
        ready = torch.cuda.Event(); ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            for r in range(world):                                  # every slot is written each step, like an all-gather's output
                gathered[r * B:(r + 1) * B].copy_(src, non_blocking=True)
            done = torch.cuda.Event(); done.record()
        return Handle(done)
    st = torch.cuda.current_stream().cuda_stream
    mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
    ms, kms = bench.rollout_multi(torch, env, 12, mode, st, obs, staging, gather_async)
    torch.cuda.synchronize()
    assert 0 < kms <= ms * 1.05 and ms / 12 < 2.5 * kms / 12 + 1.0          # the copies overlap the step kernels instead of serialising with them
    assert torch.equal(gathered[rank * B:(rank + 1) * B], obs) and torch.equal(gathered[:B], obs)


@pytest.mark.parametrize("env_id", ["myoHandPoseRandom-v0", "myoHandObjHoldFixed-v0"])
def test_fused_epilogue_equals_the_three_launches(env_id):
    """myo_bench_rollout's per-step epilogue for state-only tasks is one launch (post_kernel: observation + reward + done, TimeLimit / done
    auto-reset, first observation of the new episodes); it must leave exactly what the separate calls myo_obs, myo_autoreset,
    myo_obs_reset_only leave -- state, observation, episode counters -- over episodes that end by time limit and by done."""
    from myosuite_mjx_amd import capi
    B, K, seed, tl = 256, 60, 5, 25
    a = _make(env_id, B, seed=3, as_torch=False)
    m = _make(env_id, B, seed=3, as_torch=False)
    for e in (a, m):
        e.reset(seed=3)
    mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
    a.batch.bench_rollout(K, a.frame_skip, seed, mode, tl)
    act_ptr = m.batch.field_ptr(capi.F_ACTION)[0]
    for k in range(K):
        m.batch.random_action(act_ptr, seed, k)
        m.batch.step(act_ptr, capi.ACTMAP_MUSCLE_SIGMOID, m.frame_skip)
        m.batch.obs()
        m.batch.autoreset(tl, seed)
        m.batch.obs_reset_only()
    for f in (capi.F_QPOS, capi.F_QVEL, capi.F_ACT, capi.F_TIME, capi.F_TARGET, capi.F_OBS, capi.F_ELAPSED):
        assert np.array_equal(a.batch.read(f), m.batch.read(f)), f
    assert a.batch.read(capi.F_ELAPSED).max() < tl and a.batch.read(capi.F_ELAPSED).min() >= 0
