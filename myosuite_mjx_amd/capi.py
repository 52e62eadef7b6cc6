"""ctypes binding of libmyo_hip.so (include/myo_hip.h).  There is no CPU fallback: if the
library is missing or no MI355X is visible, loading / model upload raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MYO_HIP_LIB") or os.path.join(_HERE, "libmyo_hip.so")
SRC_PATH = os.path.join(_HERE, "csrc", "myo_hip.hip")

# field ids (myo_field)
(F_QPOS, F_QVEL, F_ACT, F_CTRL, F_WARMSTART, F_TIME, F_TARGET, F_OBS, F_REWARD, F_DONE, F_SOLVED, F_FLAGS, F_DIAG,
 F_QACC, F_TENLEN, F_ACTFORCE, F_SITEXPOS, F_ELAPSED, F_ACTION, F_FATIGUE, F_HFIELD, F_GEOMSIZE, F_LINKX, F_METRICS) = range(24)
INT_FIELDS = (F_FLAGS, F_DIAG, F_ELAPSED)
BENCH_OBS, BENCH_FRESH_ACTIONS, BENCH_AUTORESET = 1, 2, 4
ACTMAP_NONE, ACTMAP_MUSCLE_SIGMOID, ACTMAP_SIGMOID_FATIGUE, ACTMAP_SIGMOID_REAFFERENTATION, ACTMAP_CTRLRANGE = 0, 1, 2, 3, 4
TASK_NONE, TASK_POSE, TASK_REACH = 0, 1, 2
TASK_HOLD = 4
TASK_STAND = 5
TASK_TRACK = 6
FLAG_BAD_STATE, FLAG_BAD_QACC, FLAG_CONTACT_OVERFLOW, FLAG_CAND_OVERFLOW = 1, 2, 4, 8


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nq", "nv", "nu", "na", "nbody", "ntendon", "nsite", "nlink", "obs_dim",
                                       "env_lds_bytes", "lanes_per_env", "ncon_max")] + [("timestep", C.c_float)]


class TaskConfig(C.Structure):
    _fields_ = [("task", C.c_int), ("frame_skip", C.c_int), ("reset_random", C.c_int), ("target_generate", C.c_int),
                ("ntarget", C.c_int), ("ntip", C.c_int), ("tip_site", C.c_int * 8),
                ("pose_thd", C.c_float), ("far_th", C.c_float), ("near_th", C.c_float),
                ("w_pose", C.c_float), ("w_bonus", C.c_float), ("w_act_reg", C.c_float), ("w_penalty", C.c_float),
                ("w_reach", C.c_float),
                ("target_lo", C.POINTER(C.c_float)), ("target_hi", C.POINTER(C.c_float)), ("init_qpos", C.POINTER(C.c_float)),
                ("reset_noise_lo", C.POINTER(C.c_float)), ("reset_noise_hi", C.POINTER(C.c_float)),
                ("reset_clip_lo", C.POINTER(C.c_float)), ("reset_clip_hi", C.POINTER(C.c_float)),
                ("init_qvel", C.POINTER(C.c_float)), ("tip_lpos", C.c_float * 3)]


class WalkConfig(C.Structure):
    """ctypes mirror of `myo_walk_config` (include/myo_hip.h)."""
    _fields_ = [("frame_skip", C.c_int), ("hip_period", C.c_int), ("min_height", C.c_float), ("max_rot", C.c_float),
                ("target_x_vel", C.c_float), ("target_y_vel", C.c_float), ("target_rot", C.c_float * 4),
                ("body_talus_l", C.c_int), ("body_talus_r", C.c_int), ("body_pelvis", C.c_int), ("body_torso", C.c_int),
                ("qadr_hip_flexion_l", C.c_int), ("qadr_hip_flexion_r", C.c_int), ("qadr_joint_angle", C.c_int * 4),
                ("w_vel_reward", C.c_float), ("w_done", C.c_float), ("w_cyclic_hip", C.c_float), ("w_ref_rot", C.c_float),
                ("w_joint_angle_rew", C.c_float), ("init_qpos", C.POINTER(C.c_float)), ("init_qvel", C.POINTER(C.c_float)),
                ("knee_height", C.c_float), ("terrain", C.c_int), ("terrain_scalar_lo", C.c_float), ("terrain_scalar_hi", C.c_float),
                ("init_qpos_alt", C.POINTER(C.c_float)), ("init_qvel_alt", C.POINTER(C.c_float)), ("reset_noise_std", C.c_float)]


TERRAIN_NONE, TERRAIN_ROUGH, TERRAIN_HILLY, TERRAIN_STAIRS = 0, 1, 2, 3


class TrackConfig(C.Structure):
    """ctypes mirror of `myo_track_config` (include/myo_hip.h)."""
    _fields_ = [("n_frames", C.c_int), ("ref_type", C.c_int), ("horizon", C.c_int), ("robot_horizon", C.c_int), ("object_horizon", C.c_int),
                ("robot_dim", C.c_int), ("object_dim", C.c_int), ("motion_extrapolation", C.c_int), ("interpolation_linear", C.c_int),
                ("motion_start_time", C.c_double),
                ("ref_time", C.POINTER(C.c_double)), ("ref_robot", C.POINTER(C.c_double)), ("ref_robot_vel", C.POINTER(C.c_double)),
                ("ref_object", C.POINTER(C.c_double)),
                ("init_qpos", C.POINTER(C.c_float)), ("ctrl_lo", C.POINTER(C.c_float)), ("ctrl_hi", C.POINTER(C.c_float)),
                ("object_link", C.c_int), ("wrist_link", C.c_int),
                ("object_ipos", C.c_float * 3), ("object_imat", C.c_float * 9), ("wrist_ipos", C.c_float * 3),
                ("lift_z", C.c_float), ("obj_err_scale", C.c_float), ("base_err_scale", C.c_float), ("lift_bonus_mag", C.c_float),
                ("qpos_reward_weight", C.c_float), ("qpos_err_scale", C.c_float), ("qvel_reward_weight", C.c_float), ("qvel_err_scale", C.c_float),
                ("obj_fail_thresh", C.c_float), ("base_fail_thresh", C.c_float), ("qpos_fail_thresh", C.c_float),
                ("terminate_obj_fail", C.c_int), ("terminate_pose_fail", C.c_int),
                ("w_pose", C.c_float), ("w_object", C.c_float), ("w_bonus", C.c_float), ("w_penalty", C.c_float),
                ("autoreset", C.c_int), ("max_episode_steps", C.c_int), ("seed", C.c_uint64)]


# -mllvm -disable-machine-licm: the post-ISel loop-invariant code motion hoists constant materialisations (polynomial coefficients, masks) out
# of the 10-substep loop into the kernel prologue, where a dozen of them stay live in VGPRs for the whole kernel and push other values into
# scratch memory; with the pass off the headline kernel needs no register spill at all (35 before; tests/test_kernel_resources.py)
# -fno-slp-vectorize: the SLP vectoriser pairs float operations into v_pk_fma_f32 / v_pk_mul_f32 (1 064 of them in the headline kernel), which need
# 64-bit aligned register pairs: the static VALU count stays the same (the packed instructions are paid for in v_mov shuffles: 1 783 against 1 083
# moves), the register pressure rises (MyoLeg kernel 227 -> 213 VGPRs without it, and the headline kernel's 127 turn into spills with any small
# change), and the kernel is slower: measured +5.3 % at 4096 envs, +4.1 % at 32768 envs without the pass
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O2", "-std=c++17", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=on",
               "-fgpu-flush-denormals-to-zero", "-mllvm", "-disable-machine-licm", "-fno-slp-vectorize", *os.environ.get("MYO_HIPCC_EXTRA", "").split()]


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> libmyo_hip.so next to this file (cross-compiles without a GPU)."""
    csrc = os.path.dirname(SRC_PATH)
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    newest = max(newest, os.path.getmtime(os.path.join(os.path.dirname(os.path.dirname(csrc)), "include", "myo_hip.h")))
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -ffp-contract=on: multiply-adds are fused only where one source expression spells them, not across statements at the optimiser's
    # discretion (hipcc's default "fast").  Two effects, both measured: every template instantiation of the step kernel then rounds
    # identically (a scheduled full-batch launch is bit-identical to one-wave-per-env shards), and the MyoHand kernel is 6 % faster
    # -O2 rather than -O3: measured +0.8 % (hand) to +2 % (finger), neutral on the leg kernels
    # -fgpu-flush-denormals-to-zero: float32 division then lowers to v_rcp_f32 + multiply instead of the frexp / ldexp scaling that keeps
    # denormal quotients exact (about 100 divisions per substep: +3.5 % on the hand kernel, all parity tests unchanged; nothing in this
    # physics lives below 1e-38, the kernels' own guards sit at 1e-15)
    cmd = [hipcc, *HIPCC_FLAGS, "-shared", "-fPIC", "-o", LIB_PATH, SRC_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


def build_diagnostic(variant, force=False, verbose=False):
    """Diagnostic builds of the same sources next to libmyo_hip.so (selected at run time with MYO_HIP_LIB=<path>):
    "poison" (-DMYO_POISON=1): every LDS word of an env's slice starts as a NaN and the scratch memory of every wave slot is filled with NaNs
    before each step launch, so a read of a word the launch never wrote -- or a kernel whose addressing went wrong -- shows up in the parity
    tests (tests/test_gpu_poison.py runs them against this build);
    "stamps" (-DMYO_STAMPS=1): clock64 per stage (tools/gpu_stamps.py)."""
    flag = {"poison": "-DMYO_POISON=1", "stamps": "-DMYO_STAMPS=1"}[variant]
    out = os.path.join(os.path.dirname(LIB_PATH), f"libmyo_hip_{variant}.so")
    csrc = os.path.dirname(SRC_PATH)
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *HIPCC_FLAGS, flag, "-shared", "-fPIC", "-o", out, SRC_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                               "(the HIP stepper has no CPU fallback)")
        # PyTorch-ROCm bundles its own HIP runtime.  Two HIP runtimes in one process do not share the device (the second one
        # reports "no HIP GPUs"), so when torch is installed let it load its runtime first: libmyo_hip.so's libamdhip64
        # dependency then resolves to that same, already loaded, library whichever side touches the GPU first.
        if not os.environ.get("MYO_NO_TORCH"):      # MYO_NO_TORCH=1: torch-free processes (profiling drivers) skip the import
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.myo_last_error.restype = C.c_char_p
        L.myo_model_load.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.myo_model_free.argtypes = [C.c_void_p]
        L.myo_model_dims.argtypes = [C.c_void_p, C.POINTER(Dims)]
        L.myo_model_set_switch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.myo_batch_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.myo_batch_free.argtypes = [C.c_void_p]
        L.myo_batch_size.argtypes = [C.c_void_p]
        L.myo_batch_configure.argtypes = [C.c_void_p, C.POINTER(TaskConfig)]
        L.myo_batch_field.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.myo_batch_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.myo_batch_write.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.myo_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.myo_set_state.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_void_p]
        L.myo_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.myo_obs.argtypes = [C.c_void_p, C.c_void_p]
        L.myo_status.argtypes = [C.c_void_p, C.c_void_p]
        L.myo_random_action.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.myo_sync.argtypes = [C.c_void_p]
        L.myo_bench_rollout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
        L.myo_bench_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.myo_batch_set_geom_override.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.myo_probe_valu.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.myo_bench_last_kernel_name.argtypes = [C.c_void_p]
        L.myo_bench_last_kernel_name.restype = C.c_char_p
        L.myo_obs_only.argtypes = [C.c_void_p, C.c_void_p]
        L.myo_autoreset.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]
        L.myo_set_env_offset.argtypes = [C.c_void_p, C.c_int]
        L.myo_set_lanes.argtypes = [C.c_int]
        L.myo_set_balance.argtypes = [C.c_void_p, C.c_int]
        L.myo_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.myo_batch_configure_walk.argtypes = [C.c_void_p, C.POINTER(WalkConfig)]
        L.myo_batch_configure_track.argtypes = [C.c_void_p, C.POINTER(TrackConfig)]
        L.myo_obs_reset_only.argtypes = [C.c_void_p, C.c_void_p]
        L.myo_batch_set_condition.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.myo_batch_set_fatigue_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class MyoError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise MyoError(f"libmyo_hip error {rc}: {lib().myo_last_error().decode()}")


class HipModel:
    """Device-resident model (mjx.put_model counterpart, mjx/play.py:10)."""

    def __init__(self, blob: bytes, device: int = 0):
        self.h = C.c_void_p()
        _chk(lib().myo_model_load(blob, len(blob), device, C.byref(self.h)))
        self.dims = Dims()
        _chk(lib().myo_model_dims(self.h, C.byref(self.dims)))
        self.device = device

    def set_switch(self, disable_contact=0, disable_limit=0, disable_ellipsoid=0):
        _chk(lib().myo_model_set_switch(self.h, disable_contact, disable_limit, disable_ellipsoid))

    def __del__(self):
        try:
            if self.h:
                lib().myo_model_free(self.h)
        except Exception:
            pass


class HipBatch:
    """B environments' state on the device (vmapped mjx.Data counterpart, mjx/play.py:11)."""

    def __init__(self, model: HipModel, B: int):
        self.model = model
        self.B = B
        self.h = C.c_void_p()
        _chk(lib().myo_batch_create(model.h, B, C.byref(self.h)))
        self._keep = None

    def __del__(self):
        try:
            if self.h:
                lib().myo_batch_free(self.h)
        except Exception:
            pass

    def configure(self, task=TASK_NONE, frame_skip=1, reset_random=0, target_generate=0, target_lo=None, target_hi=None,
                  init_qpos=None, tip_sites=(), pose_thd=0.35, far_th=2 * np.pi, near_th=0.0,
                  w_pose=1.0, w_bonus=4.0, w_act_reg=1.0, w_penalty=50.0, w_reach=1.0, reset_noise=None, reset_clip=None, init_qvel=None,
                  tip_lpos=(0.0, 0.0, 0.0)):
        c = TaskConfig()
        c.task, c.frame_skip, c.reset_random, c.target_generate = task, frame_skip, int(reset_random), int(target_generate)
        lo = np.ascontiguousarray(target_lo if target_lo is not None else [], np.float32)
        hi = np.ascontiguousarray(target_hi if target_hi is not None else lo, np.float32)
        c.ntarget = lo.size
        c.ntip = len(tip_sites)
        for i, s in enumerate(tip_sites):
            c.tip_site[i] = int(s)
        c.pose_thd, c.far_th, c.near_th = pose_thd, far_th, near_th
        c.w_pose, c.w_bonus, c.w_act_reg, c.w_penalty, c.w_reach = w_pose, w_bonus, w_act_reg, w_penalty, w_reach
        iq = np.ascontiguousarray(init_qpos, np.float32) if init_qpos is not None else None
        fp = lambda a: np.ascontiguousarray(a, np.float32)
        extra = [fp(reset_noise[0]), fp(reset_noise[1]), fp(reset_clip[0]), fp(reset_clip[1])] if reset_noise is not None else [None] * 4
        iv = fp(init_qvel) if init_qvel is not None else None
        self._keep = (lo, hi, iq, extra, iv)
        ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None
        c.reset_noise_lo, c.reset_noise_hi, c.reset_clip_lo, c.reset_clip_hi = [ptr(a) for a in extra]
        c.init_qvel = ptr(iv)
        c.tip_lpos = (C.c_float * 3)(*[float(x) for x in tip_lpos])
        c.target_lo = lo.ctypes.data_as(C.POINTER(C.c_float)) if lo.size else None
        c.target_hi = hi.ctypes.data_as(C.POINTER(C.c_float)) if hi.size else None
        c.init_qpos = iq.ctypes.data_as(C.POINTER(C.c_float)) if iq is not None else None
        _chk(lib().myo_batch_configure(self.h, C.byref(c)))

    def configure_walk(self, *, frame_skip, hip_period, min_height, max_rot, target_x_vel, target_y_vel, target_rot, bodies, qadr_hip_flexion,
                       qadr_joint_angle, weights, init_qpos, init_qvel=None, knee_height=0.0, terrain=0, terrain_scalar=(0.0, 0.0),
                       init_qpos_alt=None, init_qvel_alt=None, reset_noise_std=0.0):
        """walk task (WalkEnvV0).  bodies = (talus_l, talus_r, pelvis, torso) body ids; weights = (vel_reward, done, cyclic_hip,
        ref_rot, joint_angle_rew)."""
        c = WalkConfig()
        c.frame_skip, c.hip_period = int(frame_skip), int(hip_period)
        c.min_height, c.max_rot, c.target_x_vel, c.target_y_vel = float(min_height), float(max_rot), float(target_x_vel), float(target_y_vel)
        c.target_rot = (C.c_float * 4)(*[float(x) for x in target_rot])
        c.body_talus_l, c.body_talus_r, c.body_pelvis, c.body_torso = [int(x) for x in bodies]
        c.qadr_hip_flexion_l, c.qadr_hip_flexion_r = [int(x) for x in qadr_hip_flexion]
        c.qadr_joint_angle = (C.c_int * 4)(*[int(x) for x in qadr_joint_angle])
        c.w_vel_reward, c.w_done, c.w_cyclic_hip, c.w_ref_rot, c.w_joint_angle_rew = [float(x) for x in weights]
        iq = np.ascontiguousarray(init_qpos, np.float32)
        iv = np.ascontiguousarray(init_qvel, np.float32) if init_qvel is not None else None
        c.init_qpos = iq.ctypes.data_as(C.POINTER(C.c_float))
        c.init_qvel = iv.ctypes.data_as(C.POINTER(C.c_float)) if iv is not None else None
        c.knee_height, c.terrain = float(knee_height), int(terrain)
        c.terrain_scalar_lo, c.terrain_scalar_hi = float(terrain_scalar[0]), float(terrain_scalar[1])
        iq2 = np.ascontiguousarray(init_qpos_alt, np.float32) if init_qpos_alt is not None else None
        iv2 = np.ascontiguousarray(init_qvel_alt, np.float32) if init_qvel_alt is not None else None
        c.init_qpos_alt = iq2.ctypes.data_as(C.POINTER(C.c_float)) if iq2 is not None else None
        c.init_qvel_alt = iv2.ctypes.data_as(C.POINTER(C.c_float)) if iv2 is not None else None
        c.reset_noise_std = float(reset_noise_std)
        _chk(lib().myo_batch_configure_walk(self.h, C.byref(c)))

    def configure_track(self, *, n_frames, reference, ref_type, init_qpos, ctrl_range, object_link, wrist_link, object_ipos, object_imat, wrist_ipos,
                        lift_z, motion_start_time=0.0, motion_extrapolation=True, interpolation_linear=False, terminate_obj_fail=True,
                        terminate_pose_fail=False, weights=(0.0, 1.0, 1.0, -2.0), autoreset=False, seed=0, max_episode_steps=0,
                        obj_err_scale=50.0, base_err_scale=40.0, lift_bonus_mag=1.0, qpos_reward_weight=0.35, qpos_err_scale=5.0,
                        qvel_reward_weight=0.05, qvel_err_scale=0.1, obj_fail_thresh=0.25, base_fail_thresh=0.25, qpos_fail_thresh=0.75):
        """MyoDM TrackEnv as a fused task of the step kernel (myo_track_config).  reference = dict(time [H], robot [Hr, nr], robot_vel | None,
        object [Ho, no]) in float64 (the reference's own arrays); ref_type 0 FIXED / 1 RANDOM / 2 TRACK; weights = (pose, object, bonus, penalty)."""
        f64 = lambda a: np.ascontiguousarray(a, np.float64)
        pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        pf = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        T, Rb, Ob = f64(reference["time"]), f64(reference["robot"]), f64(reference["object"])
        Rv = f64(reference["robot_vel"]) if reference.get("robot_vel") is not None else None
        c = TrackConfig()
        c.n_frames, c.ref_type = int(n_frames), int(ref_type)
        c.horizon, c.robot_horizon, c.object_horizon = max(Rb.shape[0], Ob.shape[0]), Rb.shape[0], Ob.shape[0]
        assert T.shape[0] >= c.horizon or ref_type != 2, "reference time axis shorter than the motion"
        c.robot_dim, c.object_dim = Rb.shape[1], Ob.shape[1]
        c.motion_extrapolation, c.interpolation_linear, c.motion_start_time = int(motion_extrapolation), int(interpolation_linear), float(motion_start_time)
        c.ref_time, c.ref_robot, c.ref_object = pd(T), pd(Rb), pd(Ob)
        c.ref_robot_vel = pd(Rv) if Rv is not None else None
        iq = np.ascontiguousarray(init_qpos, np.float32)
        cr = np.asarray(ctrl_range, np.float64)
        lo, hi = np.ascontiguousarray(cr[:, 0], np.float32), np.ascontiguousarray(cr[:, 1], np.float32)
        c.init_qpos, c.ctrl_lo, c.ctrl_hi = pf(iq), pf(lo), pf(hi)
        c.object_link, c.wrist_link = int(object_link), int(wrist_link)
        c.object_ipos = (C.c_float * 3)(*[float(x) for x in object_ipos])
        c.object_imat = (C.c_float * 9)(*[float(x) for x in np.asarray(object_imat).ravel()])
        c.wrist_ipos = (C.c_float * 3)(*[float(x) for x in wrist_ipos])
        c.lift_z, c.obj_err_scale, c.base_err_scale, c.lift_bonus_mag = float(lift_z), float(obj_err_scale), float(base_err_scale), float(lift_bonus_mag)
        c.qpos_reward_weight, c.qpos_err_scale, c.qvel_reward_weight, c.qvel_err_scale = float(qpos_reward_weight), float(qpos_err_scale), float(qvel_reward_weight), float(qvel_err_scale)
        c.obj_fail_thresh, c.base_fail_thresh, c.qpos_fail_thresh = float(obj_fail_thresh), float(base_fail_thresh), float(qpos_fail_thresh)
        c.terminate_obj_fail, c.terminate_pose_fail = int(terminate_obj_fail), int(terminate_pose_fail)
        c.w_pose, c.w_object, c.w_bonus, c.w_penalty = [float(x) for x in weights]
        c.autoreset, c.max_episode_steps, c.seed = int(autoreset), int(max_episode_steps), int(seed)
        _chk(lib().myo_batch_configure_track(self.h, C.byref(c)))

    def set_condition(self, frame_skip, epl_actuator=-1, eip_actuator=-1):
        _chk(lib().myo_batch_set_condition(self.h, int(frame_skip), int(epl_actuator), int(eip_actuator)))

    def set_fatigue_reset(self, mode=0, vec=None):
        """Fatigue compartments at reset: 0 rested, 1 random (fatigue_reset_random), 2 `vec` (fatigue_reset_vec); fatigue.py:114-134."""
        v = np.ascontiguousarray(vec, np.float32) if vec is not None else None
        _chk(lib().myo_batch_set_fatigue_reset(self.h, int(mode), v.ctypes.data_as(C.c_void_p) if v is not None else None))

    def set_geom_override(self, geom_id, size_lo=None, size_hi=None):
        """Per-env size of one collision geom, re-drawn ~ U(lo, hi) at every reset of an env (None: off)."""
        if size_lo is None:
            _chk(lib().myo_batch_set_geom_override(self.h, int(geom_id), None, None))
            return
        lo, hi = (C.c_float * 3)(*[float(x) for x in size_lo]), (C.c_float * 3)(*[float(x) for x in size_hi])
        _chk(lib().myo_batch_set_geom_override(self.h, int(geom_id), lo, hi))

    def obs_reset_only(self, stream=None):
        _chk(lib().myo_obs_reset_only(self.h, stream))

    def field_ptr(self, field):
        p, pitch, width = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _chk(lib().myo_batch_field(self.h, field, C.byref(p), C.byref(pitch), C.byref(width)))
        return p.value, pitch.value, width.value

    def read(self, field) -> np.ndarray:
        _, _, width = self.field_ptr(field)
        out = np.empty((self.B, width), np.int32 if field in INT_FIELDS else np.float32)
        _chk(lib().myo_batch_read(self.h, field, out.ctypes.data, out.nbytes))
        return out

    def write(self, field, arr):
        _, _, width = self.field_ptr(field)
        a = np.ascontiguousarray(arr, np.int32 if field in INT_FIELDS else np.float32).reshape(self.B, width)
        _chk(lib().myo_batch_write(self.h, field, a.ctypes.data, a.nbytes))

    def reset(self, mask_ptr=None, seed=0, stream=None):
        _chk(lib().myo_reset(self.h, mask_ptr, seed, stream))

    def step(self, action_ptr=None, actmap=ACTMAP_NONE, nsub=1, stream=None):
        _chk(lib().myo_step(self.h, action_ptr, actmap, nsub, stream))

    def obs(self, stream=None):
        _chk(lib().myo_obs(self.h, stream))

    def status(self) -> np.ndarray:
        out = np.zeros(self.B, np.int32)
        _chk(lib().myo_status(self.h, out.ctypes.data))
        return out

    def random_action(self, action_ptr, seed, step, env_offset=0, stream=None):
        _chk(lib().myo_random_action(self.h, action_ptr, seed, step, env_offset, stream))

    def last_kernel_name(self) -> str:
        return lib().myo_bench_last_kernel_name(self.h).decode()

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _chk(lib().myo_bench_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def obs_only(self, stream=None):
        _chk(lib().myo_obs_only(self.h, stream))

    def autoreset(self, max_episode_steps, seed=0, stream=None):
        _chk(lib().myo_autoreset(self.h, max_episode_steps, seed, stream))

    def set_balance(self, on):
        _chk(lib().myo_set_balance(self.h, int(on)))

    def set_env_offset(self, off):
        _chk(lib().myo_set_env_offset(self.h, off))

    def bench_rollout(self, steps, nsub, seed=0, mode=BENCH_OBS | BENCH_FRESH_ACTIONS | BENCH_AUTORESET, max_episode_steps=100,
                      stream=None) -> float:
        """Runs `steps` env steps on `stream`; returns the HIP-event elapsed milliseconds."""
        ms = C.c_float()
        _chk(lib().myo_bench_rollout(self.h, steps, nsub, seed, mode, max_episode_steps, stream, C.byref(ms)))
        return ms.value

    def bench_rollout_async(self, steps, nsub, seed=0, mode=BENCH_OBS | BENCH_FRESH_ACTIONS | BENCH_AUTORESET, max_episode_steps=100, stream=None):
        """Enqueues `steps` env steps on `stream` without waiting; `last_kernel_ms()` later collects the step-kernel time of all of them."""
        _chk(lib().myo_bench_rollout(self.h, steps, nsub, seed, mode, max_episode_steps, stream, None))


def set_lanes(lanes):
    _chk(lib().myo_set_lanes(lanes))


def read_stamps(batch, nwg):
    out = np.zeros((nwg, 12), np.int64)
    rc = lib().myo_read_stamps(batch.h, out.ctypes.data, nwg)
    if rc < 0:
        _chk(rc)
    return out, rc == 0


def sync(stream=None):
    _chk(lib().myo_sync(stream))


def probe_valu(waves_per_simd, iters=20000, device=0):
    """Measured chip-wide issue rate of wave64 v_fma_f32 (instructions / s) with `waves_per_simd` waves resident per SIMD, and the CU count."""
    r, n = C.c_double(0), C.c_int(0)
    _chk(lib().myo_probe_valu(int(device), int(waves_per_simd), int(iters), C.byref(r), C.byref(n)))
    return r.value, n.value
