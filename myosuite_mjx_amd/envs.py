"""Batched MyoSuite task envs on the HIP stepper, keeping the reference's gym-style API.

Mirrors (paths relative to /root/reference/myosuite/):
  * env ids, kwargs and episode lengths      envs/myo/myobase/__init__.py:258-297,300-413,523-571
  * action map + frame_skip                  envs/myo/base_v0.py:23-59,83-119
  * obs / reward / done / reset, pose task   envs/myo/myobase/pose_v0.py:98-138,141-255
  * obs / reward / done / reset, reach task  envs/myo/myobase/reach_v0.py:88-159
  * obs vector layout and float32 cast       envs/obs_vec_dict.py:86-98
  * gym step contract (obs, rwd, terminated, truncated, info)   envs/env_base.py:335-390

One env object steps `num_envs` environments in one kernel launch; tensors stay on the device
(torch views of the library's buffers, zero copy).  Physics runs only in libmyo_hip.so.
"""
from __future__ import annotations

import numpy as np

from . import capi
from . import model as _model

# ASL pose table (envs/myo/myobase/__init__.py:326-376): target joint vectors of myoHandPose{k}Fixed-v0;
# the per-joint min/max over the ten rows is the target range of myoHandPoseRandom-v0 (:396-399)
ASL_QPOS = np.array([
    [0, 0, 0, 0.5624, 0.28272, -0.75573, -1.309, 1.30045, -0.006982, 1.45492, 0.998897, 1.26466, 0, 1.40604, 0.227795, 1.07614, -0.020944, 1.46103, 0.06284, 0.83263, -0.14399, 1.571, 1.38248],
    [0, 0, 0, 0.0248, 0.04536, -0.7854, -1.309, 0.366605, 0.010473, 0.269258, 0.111722, 1.48459, 0, 1.45318, 1.44532, 1.44532, -0.204204, 1.46103, 1.44532, 1.48459, -0.2618, 1.47674, 1.48459],
    [0, 0, 0, 0.0248, 0.04536, -0.7854, -1.13447, 0.514973, 0.010473, 0.128305, 0.111722, 0.510575, 0, 0.37704, 0.117825, 1.44532, -0.204204, 1.46103, 1.44532, 1.48459, -0.2618, 1.47674, 1.48459],
    [0, 0, 0, 0.3384, 0.25305, 0.01569, -0.0262045, 0.645885, 0.010473, 0.128305, 0.111722, 0.510575, 0, 0.37704, 0.117825, 1.571, -0.036652, 1.52387, 1.45318, 1.40604, -0.068068, 1.39033, 1.571],
    [0, 0, 0, 0.6392, -0.147495, -0.7854, -1.309, 0.637158, 0.010473, 0.128305, 0.111722, 0.510575, 0, 0.37704, 0.117825, 0.306345, -0.010472, 0.400605, 0.133535, 0.21994, -0.068068, 0.274925, 0.01571],
    [0, 0, 0, 0.3384, 0.25305, 0.01569, -0.0262045, 0.645885, 0.010473, 0.128305, 0.111722, 0.510575, 0, 0.37704, 0.117825, 0.306345, -0.010472, 0.400605, 0.133535, 0.21994, -0.068068, 0.274925, 0.01571],
    [0, 0, 0, 0.6392, -0.147495, -0.7854, -1.309, 0.637158, 0.010473, 0.128305, 0.111722, 0.510575, 0, 0.37704, 0.117825, 0.306345, -0.010472, 0.400605, 0.133535, 1.1861, -0.2618, 1.35891, 1.48459],
    [0, 0, 0, 0.524, 0.01569, -0.7854, -1.309, 0.645885, -0.006982, 0.128305, 0.111722, 0.510575, 0, 0.37704, 0.117825, 1.28036, -0.115192, 1.52387, 1.45318, 0.432025, -0.068068, 0.18852, 0.149245],
    [0, 0, 0, 0.428, 0.22338, -0.7854, -1.309, 0.645885, -0.006982, 0.128305, 0.194636, 1.39033, 0, 1.08399, 0.573415, 0.667675, -0.020944, 0, 0.06284, 0.432025, -0.068068, 0.18852, 0.149245],
    [0, 0, 0, 0.5624, 0.28272, -0.75573, -1.309, 1.30045, -0.006982, 1.45492, 0.998897, 0.39275, 0, 0.18852, 0.227795, 0.667675, -0.020944, 0, 0.06284, 0.432025, -0.068068, 0.18852, 0.149245],
])
# myoHandPoseFixed-v0 target (envs/myo/myobase/__init__.py:265-291)
HAND_POSE_FIXED = np.array([0, 0, 0, -0.0904, 0.0824475, -0.681555, -0.514888, 0, -0.013964, -0.0458132, 0, 0.67553,
                            -0.020944, 0.76979, 0.65982, 0, 0, 0, 0, 0.479155, -0.099484, 0.95831, 0])
HAND_TIPS = ("THtip", "IFtip", "MFtip", "RFtip", "LFtip")
# myoHandReach* target spans (envs/myo/myobase/__init__.py:523-571)
_REACH_CENTRE = {"THtip": (-0.165, -0.537, 1.495), "IFtip": (-0.151, -0.547, 1.455), "MFtip": (-0.146, -0.547, 1.447),
                 "RFtip": (-0.148, -0.543, 1.445), "LFtip": (-0.148, -0.528, 1.434)}
_REACH_SPAN = {"THtip": ((-0.020, -0.040, -0.040), (0.040, 0.020, 0.040)),
               "IFtip": ((-0.040, -0.020, -0.010), (0.040, 0.020, 0.010)),
               "MFtip": ((-0.040, -0.020, -0.010), (0.040, 0.020, 0.010)),
               "RFtip": ((-0.040, -0.020, -0.010), (0.040, 0.020, 0.010)),
               "LFtip": ((-0.040, -0.020, -0.010), (0.040, 0.020, 0.010))}


def _pose_spec(target_lo, target_hi, reset_type, target_type, pose_thd=0.7, model="myohand_pose"):
    return dict(model=model, task="pose", max_episode_steps=100, frame_skip=10, normalize_act=True,
                target_lo=np.asarray(target_lo, float), target_hi=np.asarray(target_hi, float),
                reset_type=reset_type, target_type=target_type, pose_thd=pose_thd,
                weights=dict(pose=1.0, bonus=4.0, act_reg=1.0, penalty=50.0))


def _reach_spec(random, far_th):
    lo, hi = [], []
    for tip in HAND_TIPS:
        c = np.array(_REACH_CENTRE[tip])
        a, b = (np.array(_REACH_SPAN[tip][0]), np.array(_REACH_SPAN[tip][1])) if random else (np.zeros(3), np.zeros(3))
        lo.append(c + a)
        hi.append(c + b)
    return dict(model="myohand_pose", task="reach", max_episode_steps=100, frame_skip=10, normalize_act=True,
                target_lo=np.concatenate(lo), target_hi=np.concatenate(hi), tips=HAND_TIPS, far_th=far_th,
                reset_type="init", target_type="generate" if random else "fixed",
                weights=dict(reach=1.0, bonus=4.0, penalty=50.0, act_reg=0.0))


def _reach_box_spec(model, tips, lo, hi, far_th, max_episode_steps=100, frame_skip=10):
    """ReachEnvV0 with target_reach_range given as absolute boxes (reach_v0.py:147-153: the target is always re-drawn at reset)."""
    return dict(model=model, task="reach", max_episode_steps=max_episode_steps, frame_skip=frame_skip, normalize_act=True,
                target_lo=np.asarray(lo, float).ravel(), target_hi=np.asarray(hi, float).ravel(), tips=tuple(tips), far_th=far_th,
                reset_type="init", target_type="generate",
                weights=dict(reach=1.0, bonus=4.0, penalty=50.0, act_reg=0.0))


REGISTRY = {
    "myoHandPoseFixed-v0": _pose_spec(HAND_POSE_FIXED, HAND_POSE_FIXED, "init", "fixed"),
    "myoHandPoseRandom-v0": _pose_spec(ASL_QPOS.min(0), ASL_QPOS.max(0), "random", "generate"),
    "myoHandReachFixed-v0": _reach_spec(False, 0.044),
    "myoHandReachRandom-v0": _reach_spec(True, 0.034),
}
# myoFingerPose*-v0 (envs/myo/myobase/__init__.py:222-253): joints IFadb, IFmcp, IFpip, IFdip; defaults reset "init",
# target "generate", pose_thd 0.35 (pose_v0.py:46-57)
REGISTRY["myoFingerPoseFixed-v0"] = _pose_spec([0, 0, 0.75, 0.75], [0, 0, 0.75, 0.75], "init", "generate", 0.35, "myofinger_v0")
REGISTRY["myoFingerPoseRandom-v0"] = _pose_spec([-0.2, -0.4, 0.1, 0.1], [0.2, 1.0, 1.0, 1.0], "init", "generate", 0.35, "myofinger_v0")
# myoElbowPose1D6M*-v0 (envs/myo/myobase/__init__.py:108-137): 1-dof elbow with 6 muscles, reset "random", pose_thd 0.175
REGISTRY["myoElbowPose1D6MFixed-v0"] = _pose_spec([2.0], [2.0], "random", "generate", 0.175, "myoelbow_1dof6muscles")
REGISTRY["myoElbowPose1D6MRandom-v0"] = _pose_spec([0.0], [2.27], "random", "generate", 0.175, "myoelbow_1dof6muscles")
# motorFinger{Reach,Pose}*-v0 (envs/myo/myobase/__init__.py:55-81,188-219): the finger driven by five tendon motors (ctrlrange -1..0, no
# activation state: observations carry no act block), frame_skip 5, 200-step episodes
REGISTRY["motorFingerReachFixed-v0"] = _reach_box_spec("motorfinger_v0", ("IFtip",), [(0.2, 0.05, 0.20)], [(0.2, 0.05, 0.20)], 0.35, 200, 5)
REGISTRY["motorFingerReachRandom-v0"] = _reach_box_spec("motorfinger_v0", ("IFtip",), [(0.1, -0.1, 0.1)], [(0.27, 0.1, 0.3)], 0.35, 200, 5)
REGISTRY["motorFingerPoseFixed-v0"] = dict(_pose_spec([0, 0, 0.75, 0.75], [0, 0, 0.75, 0.75], "init", "generate", 0.35, "motorfinger_v0"),
                                           max_episode_steps=200, frame_skip=5)
REGISTRY["motorFingerPoseRandom-v0"] = dict(_pose_spec([-0.2, -0.4, 0.1, 0.1], [0.2, 1.0, 1.0, 1.0], "init", "generate", 0.35, "motorfinger_v0"),
                                            max_episode_steps=200, frame_skip=5)
# myoElbowPose1D6MExoFixed-v0 (envs/myo/myobase/__init__.py:140-160): the elbow with an exoskeleton motor on the joint (actuator 0) and
# act_reg weight 5.  (ExoRandom additionally re-draws a body mass per episode -- a per-env model edit, not offered.)
REGISTRY["myoElbowPose1D6MExoFixed-v0"] = dict(_pose_spec([2.0], [2.0], "random", "generate", 0.175, "myoelbow_1dof6muscles_1dofexo"),
                                               weights=dict(pose=1.0, bonus=4.0, act_reg=5.0, penalty=50.0))
# myoHandObjHoldFixed-v0 (envs/myo/myobase/__init__.py:596-604, obj_hold_v0.py:13-118): MyoHand palm up + a free ellipsoid object; goal = the
# model's goal site.  (ObjHoldRandom re-draws the object's geom size per episode: a per-env model edit, not offered.)
REGISTRY["myoHandObjHoldFixed-v0"] = dict(
    model="myohand_hold", task="hold", max_episode_steps=75, frame_skip=10, normalize_act=True, reset_type="init",
    goal=(-0.240, -0.520, 1.470), goal_th=0.010, drop_th=0.300,
    weights=dict(goal_dist=100.0, bonus=4.0, penalty=10.0, act_reg=0.0))
# myoHandObjHoldRandom-v0 (:605-613, ObjHoldRandomEnvV0 obj_hold_v0.py:121-140): goal = object's initial position + U(+-0.03)^3 and the object's
# ellipsoid semi-axes ~ U(0.02, 0.03)^3, both re-drawn per episode (the size is a per-env override of that one geom; mass / inertia stay)
REGISTRY["myoHandObjHoldRandom-v0"] = dict(REGISTRY["myoHandObjHoldFixed-v0"], goal=None, goal_span=0.030, object_size=((0.020,) * 3, (0.030,) * 3))
# myoFingerReach*-v0 (envs/myo/myobase/__init__.py:82-105): IFtip to an absolute target box; far_th = ReachEnvV0's default 0.35
REGISTRY["myoFingerReachFixed-v0"] = _reach_box_spec("myofinger_v0", ("IFtip",), [(0.2, 0.05, 0.20)], [(0.2, 0.05, 0.20)], 0.35)
REGISTRY["myoFingerReachRandom-v0"] = _reach_box_spec("myofinger_v0", ("IFtip",), [(0.1, -0.1, 0.1)], [(0.27, 0.1, 0.3)], 0.35)
for _k in range(10):
    REGISTRY[f"myoHandPose{_k}Fixed-v0"] = _pose_spec(ASL_QPOS[_k], ASL_QPOS[_k], "init", "fixed")
# myoLegWalk-v0 (envs/myo/myobase/__init__.py:443-459; WalkEnvV0 defaults walk_v0.py:187-266)
REGISTRY["myoLegWalk-v0"] = dict(
    model="myolegs", task="walk", max_episode_steps=1000, frame_skip=10, normalize_act=True, reset_type="init",
    min_height=0.8, max_rot=0.8, hip_period=100, target_x_vel=0.0, target_y_vel=1.2, target_rot=None,
    weights=dict(vel_reward=5.0, done=-100.0, cyclic_hip=-10.0, ref_rot=10.0, joint_angle_rew=5.0))
# myoLegStandRandom-v0 (envs/myo/myobase/__init__.py:424-441; walk_v0.py:13-183 ReachEnvV0): keep the pelvis site at a target drawn around its
# position in the (randomised) start pose; joint_random_range (-0.2, 0.2) on every joint's first coordinate, clipped to the joint range
REGISTRY["myoLegStandRandom-v0"] = dict(
    model="myolegs", task="stand", max_episode_steps=150, frame_skip=10, normalize_act=True, reset_type="random", tip="pelvis",
    joint_random_range=(-0.2, 0.2), target_span=((-0.05, -0.05, 0.0), (0.05, 0.05, 0.0)), far_th=0.44, near_th=0.050,
    weights=dict(reach=1.0, bonus=4.0, penalty=50.0, act_reg=1.0))
# myoLeg{Rough,Hilly,Stair}TerrainWalk-v0 (envs/myo/myobase/__init__.py:462-520; TerrainEnvV0, walk_v0.py:490-671): the walk task on a height
# field re-drawn per episode; hilly / stairs are registered with variant "fixed" (height scale 0.63 / 2.5; otherwise U(0.53, 0.73) / U(1.5, 3.5))
for _id, _kind, _sc in (("myoLegRoughTerrainWalk-v0", "rough", (0.0, 0.0)), ("myoLegHillyTerrainWalk-v0", "hilly", (0.63, 0.63)),
                        ("myoLegStairTerrainWalk-v0", "stairs", (2.5, 2.5))):
    REGISTRY[_id] = dict(REGISTRY["myoLegWalk-v0"], model="myolegs_terrain", terrain=_kind, terrain_scalar=_sc, knee_height=0.61)
# muscle-condition variants (register_env_with_variants, envs/myo/myobase/__init__.py:14-48): myoSarc* (sarcopenia), myoFati* (fatigue)
# for every myo* id, myoReaf* (EIP -> EPL tendon transfer) for the myoHand* ids
for _id in [k for k in list(REGISTRY) if k.startswith("myo")]:
    REGISTRY[_id[:3] + "Sarc" + _id[3:]] = dict(REGISTRY[_id], muscle_condition="sarcopenia")
    REGISTRY[_id[:3] + "Fati" + _id[3:]] = dict(REGISTRY[_id], muscle_condition="fatigue")
    if _id.startswith("myoHand"):
        REGISTRY[_id[:3] + "Reaf" + _id[3:]] = dict(REGISTRY[_id], muscle_condition="reafferentation")
# MyoDM (envs/myo/myodm/__init__.py:565-700), served by the MJX flavour of the env (mjx/myodm_v0.py TrackEnv -> track.TrackEnv, one fused
# launch per env step) for the objects that have a compiled asset: MyoHand<Object>Fixed-v0 (a one-row reference), MyoHand<Object>Random-v0 (the
# two-row randomisation range, also mjx/myodm_v0.py's module-level default) and the motion-tracking ids, whose motion file is the reference's
# data/<motion>.npz: pass `reference=<path or dict>` or point MYODM_DATA at a directory holding it (the files are not redistributed here).
# The id tables are data: assets/myodm_tasks.json = the reference's OBJECTS tuple and MyoHand_task_spec entries (tools/make_myodm_registry.py);
# an object is registered when its compiled asset assets/myohand_object_<name>.myob[.gz] is present (tools/compile_models.py: all 50).
def _myodm_tables():
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
    with open(os.path.join(here, "myodm_tasks.json")) as f:
        t = json.load(f)
    have = [o for o in t["objects"] if os.path.exists(os.path.join(here, f"myohand_object_{o}.myob")) or os.path.exists(os.path.join(here, f"myohand_object_{o}.myob.gz"))]
    return tuple(have), [tuple(x) for x in t["tasks"] if x[1] in have]


MYODM_OBJECTS, _MYODM_TASKS = _myodm_tables()
_DOF_ROBOT = 29
for _obj in MYODM_OBJECTS:
    REGISTRY[f"MyoHand{_obj.title()}Fixed-v0"] = dict(task="track", object=_obj, max_episode_steps=50, reference=dict(
        time=np.array([0.0, 4.0]), robot=np.zeros((1, _DOF_ROBOT)), robot_vel=np.zeros((1, _DOF_ROBOT)),
        object_init=np.array([-0.2, -0.2, 0.1, 1.0, 0.0, 0.0, 0.0]), object=np.array([[0.2, 0.2, 0.1, 1.0, 0.0, 0.0, 0.1]])))
    REGISTRY[f"MyoHand{_obj.title()}Random-v0"] = dict(task="track", object=_obj, max_episode_steps=50, reference=dict(
        time=np.array([0.0, 4.0]), robot=np.zeros((2, _DOF_ROBOT)), robot_vel=np.zeros((2, _DOF_ROBOT)),
        object_init=np.array([0.0, 0.0, 0.1, 1.0, 0.0, 0.0, 0.0]),
        object=np.array([[-0.2, -0.2, 0.1, 1.0, 0.0, 0.0, -1.0], [0.2, 0.2, 0.1, 1.0, 0.0, 0.0, 1.0]])))
for _id, _obj, _motion in _MYODM_TASKS:      # MyoHand_task_spec (envs/myo/myodm/__init__.py:25-560): 89 motion-tracking ids
    REGISTRY[_id] = dict(task="track", object=_obj, max_episode_steps=75, motion=_motion)

# registered by the reference but not runnable on the HIP path (DESIGN.md "out of scope")
UNSUPPORTED = {
    # (nothing of the walk family: the terrain envs run on the height-field instantiation of the leg kernel)
    "myoElbowPose1D6MExoRandom-v0": "re-draws the mass of body carry_weight per episode (a per-env model edit)",
}


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency of the stepper)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low = np.full(shape, low, dtype)
        self.high = np.full(shape, high, dtype)
        self.shape = tuple(shape)
        self.dtype = dtype

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        return rng.uniform(self.low, self.high).astype(self.dtype)


class _DevArray:
    """__cuda_array_interface__ holder so torch can view library-owned device memory without a copy."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        self._owner = owner


class BatchedMyoEnv:
    """`num_envs` copies of one MyoSuite task, stepped together on one MI355X.

    step(action[B, nu] in [-1, 1]) -> (obs[B, obs_dim] f32, reward[B], terminated[B] bool, truncated[B] bool, info)
    following envs/env_base.py:335-365.  Finished episodes (done, or max_episode_steps like gym's TimeLimit)
    are reset in place and the returned obs row is the first observation of the new episode.
    """

    # env kwargs of the reference that gym.make forwards to the env class and that are honoured here (others raise)
    ENV_KWARGS = ("reset_type", "fatigue_reset_random", "fatigue_reset_vec")

    def __init__(self, env_id, num_envs=1, device=0, seed=0, env_offset=0, autoreset=True, as_torch=True, **env_kwargs):
        if env_id in UNSUPPORTED:
            raise NotImplementedError(f"{env_id}: {UNSUPPORTED[env_id]}")
        if env_id not in REGISTRY:
            raise KeyError(f"unknown env id {env_id!r}; known: {sorted(REGISTRY)}")
        self.id = env_id
        self.spec = spec = dict(REGISTRY[env_id])
        for k, v in env_kwargs.items():
            if k not in self.ENV_KWARGS:
                raise TypeError(f"{env_id}: unsupported env kwarg {k!r} (supported: {self.ENV_KWARGS})")
            spec[k] = v
        self.num_envs = int(num_envs)
        self.device = device
        self.seed = int(seed)
        self.autoreset = autoreset
        self.as_torch = as_torch
        self.mjmodel = _model.load_asset(spec["model"])
        self.muscle_condition = spec.get("muscle_condition", "")
        if self.muscle_condition == "sarcopenia":                      # base_v0.py:64-68: a model edit
            self.mjmodel = self.mjmodel.with_sarcopenia()
        self.model = capi.HipModel(self.mjmodel.blob(), device)       # raises if there is no GPU / no library
        self.batch = capi.HipBatch(self.model, self.num_envs)
        self.batch.set_env_offset(env_offset)
        m = self.mjmodel
        self.frame_skip = spec["frame_skip"]
        self.dt = m.timestep * self.frame_skip                        # env_base.py:616-617
        self.max_episode_steps = spec["max_episode_steps"]
        w = spec["weights"]
        if spec["task"] == "pose":
            self.batch.configure(task=capi.TASK_POSE, frame_skip=self.frame_skip,
                                 reset_random=spec["reset_type"] == "random", target_generate=spec["target_type"] == "generate",
                                 target_lo=spec["target_lo"], target_hi=spec["target_hi"], init_qpos=m.qpos0,
                                 pose_thd=spec["pose_thd"], far_th=4 * np.pi / 2,
                                 w_pose=w["pose"], w_bonus=w["bonus"], w_act_reg=w["act_reg"], w_penalty=w["penalty"])
            self.obs_dim = 3 * m.nq + m.n_muscle
        elif spec["task"] == "walk":
            key_qpos = np.asarray(m.key_qpos).reshape(-1, m.nq)
            key_qvel = np.asarray(m.key_qvel).reshape(-1, m.nv)
            # walk_v0.py:254 init_qpos = key_qpos[0] (the reference orientation of ref_rot); reset "init" starts from keyframe 2
            # (walk_v0.py:339-349), "random" from keyframe 2 or 3 with N(0, 0.02) noise (:316-332; drawn by the reset kernel)
            if spec["reset_type"] not in ("init", "random"):
                raise NotImplementedError("myoLegWalk: reset_type 'init' (keyframe 2) or 'random' (walk_v0.py:316-332)")
            rnd = spec["reset_type"] == "random"
            jadr = lambda n: int(m.jnt_qposadr[m.name2id("joint", n)])
            self.batch.configure_walk(
                frame_skip=self.frame_skip, hip_period=spec["hip_period"], min_height=spec["min_height"], max_rot=spec["max_rot"],
                target_x_vel=spec["target_x_vel"], target_y_vel=spec["target_y_vel"],
                target_rot=spec["target_rot"] if spec["target_rot"] is not None else key_qpos[0][3:7],
                bodies=[m.name2id("body", n) for n in ("talus_l", "talus_r", "pelvis", "torso")],
                qadr_hip_flexion=[jadr("hip_flexion_l"), jadr("hip_flexion_r")],
                qadr_joint_angle=[jadr(n) for n in ("hip_adduction_l", "hip_adduction_r", "hip_rotation_l", "hip_rotation_r")],
                weights=[w[k] for k in ("vel_reward", "done", "cyclic_hip", "ref_rot", "joint_angle_rew")],
                init_qpos=key_qpos[2], init_qvel=key_qvel[2], knee_height=spec.get("knee_height", 0.0),
                terrain={"rough": capi.TERRAIN_ROUGH, "hilly": capi.TERRAIN_HILLY, "stairs": capi.TERRAIN_STAIRS}.get(spec.get("terrain"), capi.TERRAIN_NONE),
                terrain_scalar=spec.get("terrain_scalar", (0.0, 0.0)),
                init_qpos_alt=key_qpos[3] if rnd else None, init_qvel_alt=key_qvel[3] if rnd else None, reset_noise_std=0.02 if rnd else 0.0)
            self.obs_dim = (m.nq - 2) + m.nv + 16 + 4 * m.nu
        elif spec["task"] == "stand":
            from .mjcf import quat2mat
            key_qpos = np.asarray(m.key_qpos).reshape(-1, m.nq)
            key_qvel = np.asarray(m.key_qvel).reshape(-1, m.nv)
            init = key_qpos[0].astype(float)                              # walk_v0.py:63-64
            adr = np.asarray(m.jnt_qposadr)
            nlo, nhi = np.zeros(m.nq), np.zeros(m.nq)
            clo, chi = np.full(m.nq, -1e30), np.full(m.nq, 1e30)
            nlo[adr], nhi[adr] = spec["joint_random_range"]               # generate_qpos (:152-167): only each joint's first coordinate moves ...
            clo[adr], chi[adr] = m.jnt_range[:, 0], m.jnt_range[:, 1]      # ... and is clipped to jnt_range -- (0, 0) for the unlimited free root: x = 0
            tsid = m.name2id("site", spec["tip"])
            if int(m.hip_site_link[tsid]) != 0:
                raise NotImplementedError("stand task: the tip site must ride on the free root link")
            lpos = np.asarray(m.hip_site_lpos[tsid], float)
            q0 = np.clip(init[:7] + 0.0, np.r_[clo[:1], [-1e30] * 6], np.r_[chi[:1], [1e30] * 6])
            p0 = q0[:3] + quat2mat(q0[3:7] / np.linalg.norm(q0[3:7])) @ lpos     # generate_targets (:140-149): the site in the first random pose
            span = np.asarray(spec["target_span"], float)
            self.batch.configure(task=capi.TASK_STAND, frame_skip=self.frame_skip, reset_random=0, target_generate=1,
                                 target_lo=p0 + span[0], target_hi=p0 + span[1], init_qpos=init, init_qvel=key_qvel[0],
                                 reset_noise=(nlo, nhi), reset_clip=(clo, chi), tip_lpos=lpos,
                                 near_th=spec["near_th"], far_th=spec["far_th"],
                                 w_reach=w["reach"], w_bonus=w["bonus"], w_act_reg=w["act_reg"], w_penalty=w["penalty"])
            self.obs_dim = m.nq + m.nv + 6 + m.n_muscle
        elif spec["task"] == "hold":
            init = np.array(m.qpos0, float)
            init[:-7] = 0.0                                            # obj_hold_v0.py:63-64: fully open hand, palm up
            init[0] = -1.5
            if spec["goal"] is None:                                   # Random: around the object's site at the model's initial pose (:125-131)
                glo, ghi = np.asarray(m.qpos0[-7:-4], float) - spec["goal_span"], np.asarray(m.qpos0[-7:-4], float) + spec["goal_span"]
            else:
                glo = ghi = np.asarray(spec["goal"], float)
            self.batch.configure(task=capi.TASK_HOLD, frame_skip=self.frame_skip, reset_random=0, target_generate=int(spec["goal"] is None),
                                 target_lo=glo, target_hi=ghi, init_qpos=init,
                                 near_th=spec["goal_th"], far_th=spec["drop_th"],
                                 w_reach=w["goal_dist"], w_bonus=w["bonus"], w_act_reg=w["act_reg"], w_penalty=w["penalty"])
            self.obs_dim = (m.nq - 7) + (m.nv - 6) + 6 + m.n_muscle
            if "object_size" in spec:
                self.batch.set_geom_override(m.name2id("geom", "object"), *spec["object_size"])
        else:
            tips = [m.name2id("site", t) for t in spec["tips"]]
            n = len(tips)
            self.batch.configure(task=capi.TASK_REACH, frame_skip=self.frame_skip, reset_random=0,
                                 target_generate=spec["target_type"] == "generate", target_lo=spec["target_lo"],
                                 target_hi=spec["target_hi"], init_qpos=m.qpos0, tip_sites=tips,
                                 far_th=spec["far_th"] * n, near_th=0.0125 * n,
                                 w_reach=w["reach"], w_bonus=w["bonus"], w_act_reg=w["act_reg"], w_penalty=w["penalty"])
            self.obs_dim = 2 * m.nq + 6 * n + m.n_muscle
        self.actmap = capi.ACTMAP_MUSCLE_SIGMOID
        if self.muscle_condition == "fatigue":                         # base_v0.py:70-74, 100-104
            self.actmap = capi.ACTMAP_SIGMOID_FATIGUE
            self.batch.set_condition(self.frame_skip)
            if spec.get("fatigue_reset_random"):                       # base_v0.py:30-31, 120-127 -> fatigue.py:114-134
                if spec.get("fatigue_reset_vec") is not None:
                    raise AssertionError("Cannot use 'fatigue_reset_vec' if fatigue_reset_random=False.")   # the reference's own (oddly worded) assertion
                self.batch.set_fatigue_reset(1)
            elif spec.get("fatigue_reset_vec") is not None:
                vec = np.asarray(spec["fatigue_reset_vec"], np.float32)
                if len(vec) != m.n_muscle:
                    raise AssertionError(f"Invalid length of initial/reset fatigue vector (expected {m.n_muscle}, but obtained {len(vec)}).")
                full = np.zeros(m.nu, np.float32)
                full[np.asarray(m.actuator_kind) == 0] = vec
                self.batch.set_fatigue_reset(2, full)
        elif self.muscle_condition == "reafferentation":               # base_v0.py:76-80, 105-109
            self.actmap = capi.ACTMAP_SIGMOID_REAFFERENTATION
            self.batch.set_condition(self.frame_skip, m.name2id("actuator", "EPL"), m.name2id("actuator", "EIP"))
        self.act_dim = m.nu
        self.action_space = Box(-1.0, 1.0, (m.nu,))                    # env_base.py:101-113 (normalize_act)
        self.observation_space = Box(-10.0, 10.0, (self.obs_dim,))     # env_base.py:172-176
        self._episode_seed = self.seed
        self._views = {}
        self._action_buf = None
        if as_torch:
            import torch
            self._torch = torch
            self._action_buf = torch.empty((self.num_envs, m.nu), dtype=torch.float32, device=f"cuda:{device}")

    # -- zero-copy views ---------------------------------------------------------------------------------
    def view(self, field):
        """torch view (as_torch) or numpy copy of a per-env field."""
        if not self.as_torch:
            return self.batch.read(field)
        if field not in self._views:
            ptr, pitch, width = self.batch.field_ptr(field)
            ts = "<i4" if field in capi.INT_FIELDS else "<f4"
            arr = _DevArray(ptr, (self.num_envs, width), ts, self.batch)
            self._views[field] = self._torch.as_tensor(arr, device=f"cuda:{self.device}")
        return self._views[field]

    def _stream(self):
        if self.as_torch:
            return self._torch.cuda.current_stream(self.device).cuda_stream
        return None

    # -- gym API -------------------------------------------------------------------------------------------
    def reset(self, seed=None):
        if seed is not None:
            self._episode_seed = int(seed)
        s = self._stream()
        self.batch.reset(None, self._episode_seed, s)
        self.batch.obs(s)
        return self.view(capi.F_OBS)

    def step(self, action):
        s = self._stream()
        if self.as_torch:
            a = self._torch.as_tensor(action, dtype=self._torch.float32, device=self._action_buf.device)
            a = self._torch.clamp(a, -1.0, 1.0, out=self._action_buf)      # env_base.py:341 (clip to action space)
            self.batch.step(a.data_ptr(), self.actmap, self.frame_skip, s)
        else:
            a = np.clip(np.ascontiguousarray(action, np.float32).reshape(self.num_envs, self.act_dim), -1, 1)
            # host actions are uploaded into the library's action buffer; the action map runs in the step kernel either way
            self.batch.write(capi.F_ACTION, a)
            self.batch.step(self.batch.field_ptr(capi.F_ACTION)[0], self.actmap, self.frame_skip, s)
        if self.spec["task"] != "walk":      # the walk task's observation / reward pass is fused into the step kernel
            self.batch.obs(s)
        if self.as_torch:
            reward = self.view(capi.F_REWARD)[:, 0].clone()
            done = self.view(capi.F_DONE)[:, 0] > 0
            elapsed = self.view(capi.F_ELAPSED)[:, 0]
            truncated = (elapsed >= self.max_episode_steps) & ~done
            solved = self.view(capi.F_SOLVED)[:, 0] > 0
        else:
            reward = self.batch.read(capi.F_REWARD)[:, 0]
            done = self.batch.read(capi.F_DONE)[:, 0] > 0
            truncated = (self.batch.read(capi.F_ELAPSED)[:, 0] >= self.max_episode_steps) & ~done
            solved = self.batch.read(capi.F_SOLVED)[:, 0] > 0
        if self.autoreset:
            self.batch.autoreset(self.max_episode_steps, self._episode_seed, s)
            self.batch.obs_reset_only(s)
        info = {"solved": solved, "time": self.view(capi.F_TIME)}
        return self.view(capi.F_OBS), reward, done, truncated, info

    # -- state access (env_base.py:643-705 get_env_state / set_env_state) ----------------------------------
    def get_env_state(self):
        return {k: self.batch.read(f) for k, f in (("qpos", capi.F_QPOS), ("qvel", capi.F_QVEL), ("act", capi.F_ACT),
                                                   ("time", capi.F_TIME), ("target", capi.F_TARGET))}

    def set_env_state(self, state):
        for k, f in (("qpos", capi.F_QPOS), ("qvel", capi.F_QVEL), ("act", capi.F_ACT), ("time", capi.F_TIME), ("target", capi.F_TARGET)):
            if k in state:
                self.batch.write(f, state[k])

    def status(self):
        """Per-env int32 flag bits since the last call (capi.FLAG_*); bad states were reset like mj_sim_scene.py:56-61."""
        return self.batch.status()


def _make_track(env_id, num_envs, reference=None, **kw):
    import os
    from .track import TrackEnv
    spec = REGISTRY[env_id]
    if reference is None:
        reference = spec.get("reference")
    if reference is None:                      # a motion-tracking id: the reference's own motion file
        roots = [os.environ.get("MYODM_DATA"), "/root/reference/myosuite/envs/myo/myodm/data"]
        hits = [os.path.join(r, spec["motion"]) for r in roots if r and os.path.exists(os.path.join(r, spec["motion"]))]
        if not hits:
            raise FileNotFoundError(f"{env_id}: motion file {spec['motion']} not found; pass reference=<path or dict> or set MYODM_DATA to the "
                                    "directory of the reference's envs/myo/myodm/data")
        reference = hits[0]
    kw.setdefault("max_episode_steps", spec["max_episode_steps"])
    env = TrackEnv(num_envs=num_envs, object_name=spec["object"], reference=reference, gym_api=True, **kw)
    env.id = env_id
    return env


def make(env_id, num_envs=1, **kw):
    """gym.make counterpart for the batched envs (envs/myo/myobase/__init__.py and envs/myo/myodm/__init__.py register the same ids)."""
    if env_id in REGISTRY and REGISTRY[env_id].get("task") == "track":
        return _make_track(env_id, num_envs, **kw)
    return BatchedMyoEnv(env_id, num_envs=num_envs, **kw)
