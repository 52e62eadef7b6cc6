#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched MyoSuite env.step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json metric): myoHandPoseRandom-v0, 4096 envs per GPU (weak scaling), one "step" = one
batched env step = action map + 10 fused physics substeps + obs/reward + TimeLimit/done auto-reset, with
synthetic U(-1,1) actions generated on the device by a counter RNG inside the timed region.  For N > 1 every
step also all-gathers the observations over RCCL (the only collective the path has).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 4096
ENV_ID = "myoHandPoseRandom-v0"
B_ALG = 1560.0          # algorithmic HBM bytes per env-step, MyoHand pose
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3
N_SIMD = 256 * 4        # MI355X_MICROARCH.md: 256 CUs, 4 SIMDs each
CLOCK_HZ = 2.4e9        # max engine clock
VALU_CYCLES = 2.0       # MI355X_MICROARCH.md: a wave64 v_fma_f32 issues over 2 cycles on a SIMD-32 (4 for one wave alone)
# committed rocprofv3 counters of the step kernel per (env, envs per GPU) (tools/prof_all.sh): HBM traffic and VALU instruction counts
PMC_FILES = {("myoHandPoseRandom-v0", 4096): "profiles/r3_pmc_step_kernel_hand.json", ("myoHandPoseRandom-v0", 32768): "profiles/r3_pmc_step_kernel_hand_B32768.json",
             ("myoLegWalk-v0", 4096): "profiles/r3_pmc_step_kernel_legs.json", ("MyoHandAirplaneRandom-v0", 4096): "profiles/r3_pmc_step_kernel_trackenv.json"}
FLOP_FILE = "profiles/r2_flops_oracle.json"             # committed flop count of the oracle's instrumented build (tools/count_flops.py)
TRACK_ALIAS = "MyoDM-TrackEnv"                          # round-2 spelling of the MyoDM TrackEnv extra measurement = MyoHandAirplaneRandom-v0


def flops_per_env_step(env_id):
    """Algorithmic flop per env step: the count emitted by the oracle's instrumented build on this workload."""
    try:
        with open(os.path.join(ROOT, FLOP_FILE)) as f:
            for r in json.load(f):
                if r["env"] == env_id:
                    return r["per_env_step"]["total"]
    except Exception:
        pass
    return None


def cpu_baseline(seconds_target=12.0):
    """The f64 C oracle stepping the same workload on every host core (kind 'port'), bounded sample."""
    import numpy as np
    from myosuite_mjx_amd import model as M
    from myosuite_mjx_amd.envs import REGISTRY
    from oracle.oracle import Oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    m = M.load_asset("myohand_pose")
    o = Oracle(m.blob())
    spec = REGISTRY[ENV_ID]
    rng = np.random.default_rng(0)
    envs_per_core, nsteps = 32, 40
    B = envs_per_core * cores
    qpos = rng.uniform(m.jnt_range[:, 0], m.jnt_range[:, 1], (B, m.nq))
    qvel = np.zeros((B, m.nv)); act = np.zeros((B, m.nu)); warm = np.zeros((B, m.nv)); tm = np.zeros(B)
    t0 = time.perf_counter()
    done_steps = 0
    for s in range(nsteps):
        a = rng.uniform(-1, 1, (B, m.nu))
        ctrl = np.ascontiguousarray(1.0 / (1.0 + np.exp(-5.0 * (a - 0.5))))
        o.step_batch(qpos, qvel, act, warm, tm, ctrl, spec["frame_skip"], cores)
        done_steps += 1
        if time.perf_counter() - t0 > seconds_target:
            break
    el = time.perf_counter() - t0
    return {"value": B * done_steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{B} envs x {done_steps} env-steps (10 substeps each) of {ENV_ID}, f64 C oracle, {cores} threads, {el:.1f}s"}


def rollout_multi(torch, env, nsteps, mode, stream, obs, staging, gather_async):
    """The N > 1 loop: nothing waits on the host.  The observation all-gather of step t (RCCL over xGMI, on the process group's own
    stream) overlaps the step kernel of step t+1: the observations are first copied into `staging`, and the compute stream only waits
    for the previous gather right before `staging` is overwritten again.  `gather_async(src)` starts the collective and returns a
    handle with .wait() (stream-side wait).  Returns (HIP-event ms of the whole loop, HIP-event ms inside the step kernels)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    work = None
    for _ in range(nsteps):
        env.batch.bench_rollout_async(1, env.frame_skip, 0, mode, env.max_episode_steps, stream)
        if work is not None:
            work.wait()                                # gather t-1 has finished reading `staging`
        staging.copy_(obs, non_blocking=True)
        work = gather_async(staging)
    if work is not None:
        work.wait()
    e1.record()
    kms = env.batch.last_kernel_ms()                   # waits for the last step kernel
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), kms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=1, help="repeat the timed region R times and report the median run (SURVEY 8d config 2: 5); default 1 = the plain contract")
    ap.add_argument("--env", default=ENV_ID, help="env id (default: the BASELINE.json headline workload); other ids are extra measurements")
    args = ap.parse_args()
    env_id = "MyoHandAirplaneRandom-v0" if args.env == TRACK_ALIAS else args.env

    import torch
    from myosuite_mjx_amd import capi
    from myosuite_mjx_amd.envs import REGISTRY, make
    is_track = REGISTRY.get(env_id, {}).get("task") == "track"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal on a one-GPU box (tests only): MYO_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo, so that the N > 1 control
        # flow can be exercised without N GPUs; the numbers of such a run mean nothing and are marked in the output
        if os.environ.get("MYO_BENCH_REHEARSAL") == "1":
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP stepper has no CPU fallback")
    torch.cuda.set_device(local)
    B = args.batch
    if is_track:   # MyoDM TrackEnv (mjx/myodm_v0.py): the whole env step, reward / done / masked reset included, is one launch of the step kernel
        env = make(env_id, num_envs=B, device=local, seed=0, autoreset=True)
        env.batch.set_env_offset(rank * B)
    else:
        env = make(env_id, num_envs=B, device=local, seed=0, env_offset=rank * B)
    env.reset(seed=0)
    stream = torch.cuda.current_stream(local).cuda_stream
    mode = capi.BENCH_OBS | capi.BENCH_FRESH_ACTIONS | capi.BENCH_AUTORESET
    obs = env.view(capi.F_OBS)
    gathered = torch.empty((world * B, env.obs_dim), dtype=torch.float32, device=obs.device) if world > 1 else None
    staging = torch.empty_like(obs) if world > 1 else None

    def run(nsteps):
        """nsteps batched env steps; returns (HIP-event ms of all step work, HIP-event ms inside the step kernel) on this rank."""
        if world == 1:
            ms = env.batch.bench_rollout(nsteps, env.frame_skip, 0, mode, env.max_episode_steps, stream)
            return ms, env.batch.last_kernel_ms()
        return rollout_multi(torch, env, nsteps, mode, stream, obs, staging,
                             lambda src: dist.all_gather_into_tensor(gathered, src, async_op=True))

    run(args.warmup)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    runs = []
    for _ in range(max(1, args.repeats)):
        t0 = time.perf_counter()
        ev_ms, kernel_ms_total = run(args.steps)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist:
            t = torch.tensor([el], device=obs.device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        runs.append((el, ev_ms, kernel_ms_total))
    el, ev_ms, kernel_ms_total = sorted(runs)[len(runs) // 2]            # median run (the only run when --repeats 1)
    # N > 1: the observation all-gather on its own ("gather time separately"), outside the timed region: in the
    # rollout it overlaps the next step kernel, here 20 back-to-back gathers are timed with nothing else running
    gather_ms = None
    if dist:
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.all_gather_into_tensor(gathered, staging)
        torch.cuda.synchronize()
        g0.record()
        for _ in range(20):
            dist.all_gather_into_tensor(gathered, staging)
        g1.record()
        torch.cuda.synchronize()
        gather_ms = g0.elapsed_time(g1) / 20
    # dominant kernel: average launch duration from HIP event pairs recorded around every step-kernel launch of the
    # timed region, on the stream it is launched on
    k_ms = kernel_ms_total / args.steps
    flags = env.status()
    # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside this process; the value is
    # the committed rocprofv3 measurement of the same kernel / batch (profiles/, separate FETCH_SIZE / WRITE_SIZE passes)
    traffic = None
    valu_insts = None
    PMC_FILE = PMC_FILES.get((env_id, B))
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            pmc = json.load(f)
        traffic = pmc["traffic"]["hbm_bytes_per_launch_raw"]
        valu_insts = pmc["counters"]["SQ_INSTS_VALU"]["mean_per_launch"]
    except Exception:
        pass
    # measured VALU issue peak of THIS chip at the step kernel's occupancy (4 waves per SIMD): a 20 ms v_fma_f32 probe after the timed region
    valu_peak, valu_peak_by_occupancy = None, None
    if rank == 0:
        try:   # v_fma_f32 issue-rate microbenchmark at 1 / 2 / 4 / 8 waves per SIMD; the step kernel runs at 4
            valu_peak = capi.probe_valu(4, 20000, local)[0]
            valu_peak_by_occupancy = {"4": valu_peak}
            for w in (1, 2, 8):
                try:
                    valu_peak_by_occupancy[str(w)] = capi.probe_valu(w, 20000, local)[0]
                except Exception:
                    pass
        except Exception:
            valu_peak = None
    if rank == 0:
        value = world * B * args.steps / el
        mm = env.mjmodel
        # algorithmic HBM bytes per env-step: SURVEY.md 8d's figure for the headline workload; for other envs the same accounting
        # (state + action read once, state + diagnostics + observation written once)
        # (state + action read once, state + observation + reward/done written once; L-walk: 3 412 B)
        b_alg = B_ALG if env_id == ENV_ID else 4.0 * ((mm.nq + 2 * mm.nv + 2 * mm.nu + 1) + (mm.nq + 2 * mm.nv + mm.nu + 1) + env.obs_dim + 2)
        achieved = b_alg * B / (k_ms * 1e-3) / 1e9
        f_alg = flops_per_env_step(env_id)
        out = {
            "metric": f"env-steps/s (whole node) {env_id} batch {B}", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{env_id}, {B} envs per GPU, frame_skip={env.frame_skip} (dt={env.dt:g}), U(-1,1) device-generated actions, "
                                   + ("reference lookup + obs + reward + done + masked reset fused into the step launch (MYO_TASK_TRACK)" if is_track else
                                      f"obs+reward+TimeLimit({env.max_episode_steps})/done auto-reset inside the timed region"
                                      + (f"; {args.warmup + args.steps} env steps from reset: " + ("a TimeLimit reset of every env falls inside the timed region"
                                          if args.warmup < env.max_episode_steps <= args.warmup + args.steps else "NO TimeLimit reset inside the timed region (done-resets only)")))
                                   + (", RCCL obs all-gather per step (overlapped with the next step)" if world > 1 else ""),
                       "global_batch": world * B, "parallelism": f"env-shard x{world}", "lanes_per_env": 64,
                       "substeps_per_s": value * env.frame_skip},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": env.batch.last_kernel_name(), "kernel_ms": k_ms,
                         "alg_bytes_per_launch": b_alg * B,
                         "note": "path is FP32-VALU / latency bound, not HBM bound (SURVEY.md 8d); fp32 and VALU-issue views alongside",
                         "traffic_source": None if traffic is None else f"committed rocprofv3 profile {PMC_FILE} (separate FETCH_SIZE / WRITE_SIZE passes), not measured in this run",
                         "fp32": None if f_alg is None else {
                             "achieved_tflops": f_alg * B / (k_ms * 1e-3) / 1e12, "peak_tflops": FP32_PEAK_TFLOPS,
                             "frac": f_alg * B / (k_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, "flop_per_env_step": f_alg,
                             "flop_source": f"oracle instrumented build, {FLOP_FILE}"},
                         # VALU issue slots: instruction count = committed rocprofv3 SQ_INSTS_VALU of the same kernel / batch; time = this run's
                         # kernel time; peak = the v_fma_f32 issue rate measured on this chip in this run at 4 waves per SIMD (guide: 2 cycles per
                         # wave64 instruction => 1 229 G/s at 2.4 GHz; the chip holds a lower clock under an all-VALU load)
                         "valu_issue": None if valu_insts is None else {
                             "wave_insts_per_launch": valu_insts, "insts_source": f"committed rocprofv3 profile {PMC_FILE}",
                             "peak_wave_insts_per_s": valu_peak if valu_peak else N_SIMD * CLOCK_HZ / VALU_CYCLES,
                             "peak_source": "measured in this run: myo_probe_valu, 4 waves per SIMD" if valu_peak else "guide: 1024 SIMDs x 2.4 GHz / 2 cycles",
                             "peak_guide": N_SIMD * CLOCK_HZ / VALU_CYCLES, "peak_measured_by_waves_per_simd": valu_peak_by_occupancy,
                             "frac": valu_insts / (k_ms * 1e-3) / (valu_peak if valu_peak else N_SIMD * CLOCK_HZ / VALU_CYCLES)}},
            "event_ms_per_step_rank0": ev_ms / args.steps,
            **({"repeats": len(runs), "value_per_repeat": [world * B * args.steps / r[0] for r in runs]} if len(runs) > 1 else {}),
            **({"allgather_ms_rank0": gather_ms, "allgather_bytes_out": int(gathered.numel() * 4)} if gather_ms is not None else {}),
            **({"rehearsal": "all ranks on one GPU over gloo: control-flow check only"} if os.environ.get("MYO_BENCH_REHEARSAL") == "1" else {}),
            "flagged_envs": int((flags != 0).sum()),
        }
        if not args.no_cpu_baseline and world == 1 and env_id == ENV_ID:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
