/* myo_hip.h -- C ABI of libmyo_hip.so: the MI355X-native batched musculoskeletal stepper.
 *
 * Drop-in boundary for the reference's sim-backend / MJX step path.  Each entry point names
 * the reference interface it replaces (paths relative to /root/reference/myosuite/):
 *
 *   myo_model_load   <- mjx.put_model(mj_model)                 mjx/play.py:10
 *                       DMSimScene._load_simulation             physics/mj_sim_scene.py:28-49
 *   myo_batch_create <- mjx.put_data(m, d) (vmapped Data)        mjx/play.py:11
 *   myo_reset        <- Robot.reset / PoseEnvV0.reset            robot/robot.py:913-998, envs/myo/myobase/pose_v0.py:172-255
 *   myo_set_state /
 *   myo_get_state    <- SimScene.set_state/get_state             physics/sim_scene.py:145-166, envs/env_base.py:643-705
 *   myo_step         <- BaseV0.step -> Robot.step -> sim.advance envs/myo/base_v0.py:83-119, robot/robot.py:844-910,
 *                       -> Physics.step(substeps)                physics/mj_sim_scene.py:51-65 ; jit(mjx.step) mjx/play.py:41,47
 *   myo_obs          <- MujocoEnv.get_obs / get_obs_dict / get_reward_dict / obsdict2obsvec
 *                       envs/env_base.py:392-417, envs/myo/myobase/pose_v0.py:98-138, reach_v0.py:88-144,
 *                       envs/obs_vec_dict.py:86-98
 *   myo_status       <- DMSimScene.advance's exception-and-reset path  physics/mj_sim_scene.py:54-61
 *
 * Conventions: every function returns 0 on success or a negative MYO_E_* code and never throws
 * or aborts; myo_last_error() gives a thread-local message.  All device arrays are env-major
 * float32: element (env e, index i) of a field lives at ptr[e * pitch + i].  Functions taking a
 * `stream` are asynchronous on that hipStream_t (NULL = default stream).  One host thread per
 * myo_batch.  No torch types anywhere in this ABI.
 */
#ifndef MYO_HIP_H
#define MYO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct myo_model myo_model;
typedef struct myo_batch myo_batch;

enum { MYO_OK = 0, MYO_E_ARG = -1, MYO_E_BLOB = -2, MYO_E_HIP = -3, MYO_E_UNSUPPORTED = -4, MYO_E_NOMEM = -5 };

typedef struct myo_dims {
  /* na = MuJoCo's na: actuators with an activation state (muscles).  Stateless actuators (<motor> etc.) still own a slot of the
   * [B][nu]-wide MYO_F_ACT rows (kept at zero) so that actuator indices address every per-actuator field alike */
  int nq, nv, nu, na, nbody, ntendon, nsite, nlink, obs_dim, env_lds_bytes, lanes_per_env, ncon_max;
  float timestep;
} myo_dims;

/* per-env state / output fields (all float32 unless noted) */
typedef enum myo_field {
  MYO_F_QPOS = 0,    /* [B][nq] */
  MYO_F_QVEL,        /* [B][nv] */
  MYO_F_ACT,         /* [B][nu]  activation per actuator (zero for stateless actuators) */
  MYO_F_CTRL,        /* [B][nu]  last applied control (after the action map) */
  MYO_F_WARMSTART,   /* [B][nv]  qacc_warmstart */
  MYO_F_TIME,        /* [B][1] */
  MYO_F_TARGET,      /* [B][ntarget] pose: target joint vector; reach: target tip positions */
  MYO_F_OBS,         /* [B][obs_dim] */
  MYO_F_REWARD,      /* [B][1] dense reward */
  MYO_F_DONE,        /* [B][1] 1.0f if done */
  MYO_F_SOLVED,      /* [B][1] 1.0f if solved */
  MYO_F_FLAGS,       /* [B][1] int32 bit flags, see MYO_FLAG_* */
  MYO_F_DIAG,        /* [B][8] int32 diagnostics: nefc, ncon, solver_iter(max over substeps), ncon_dropped, ... */
  MYO_F_QACC,        /* [B][nv]  qacc of the last substep (diagnostic / parity) */
  MYO_F_TENLEN,      /* [B][nu]  actuator (tendon) lengths of the last substep */
  MYO_F_ACTFORCE,    /* [B][nu]  actuator forces of the last substep */
  MYO_F_SITEXPOS,    /* [B][3*ntip] tip site world positions after the step (reach task) */
  MYO_F_ELAPSED,     /* [B][1] int32 env steps since the last reset (gym TimeLimit counter) */
  MYO_F_ACTION,      /* [B][nu]  library-owned action buffer: write normalised actions here and pass its pointer to myo_step */
  MYO_F_FATIGUE,     /* [B][3*nu] fatigue compartments MA | MR | MF (muscle condition "fatigue") */
  MYO_F_HFIELD,      /* [B][nrow*ncol] height-field elevation per env (terrain models: mjModel.hfield_data, rewritten per episode by
                        TerrainEnvV0.reset, walk_v0.py:563-622); absent (MYO_E_ARG) for models without a colliding height field */
  MYO_F_GEOMSIZE,    /* [B][4] per-env size (3) + bounding radius of the geom named in myo_batch_set_geom_override (absent otherwise) */
  MYO_F_LINKX,       /* [B][12*nlink] world frame (pos 3 + rotation 9, row-major) of every kinematic link as the LAST substep's position stage
                        computed it, i.e. at the state before that substep's integration: what an MJX pipeline_state holds in xpos / xmat
                        after mjx.step (forward, then integrate; mjx/myodm_v0.py:201-232 reads it).  Models of the TrackEnv class only */
  MYO_F_METRICS,     /* [B][4] reward terms of the track task: pose, object, bonus, penalty (mjx/myodm_v0.py:243-262 `rwd_dict`) */
  MYO_F_COUNT
} myo_field;

enum { MYO_FLAG_BAD_STATE = 1, MYO_FLAG_BAD_QACC = 2, MYO_FLAG_CONTACT_OVERFLOW = 4, MYO_FLAG_CAND_OVERFLOW = 8,
       MYO_FLAG_SCHED_TIMEOUT = 16 /* opt-in substep scheduler gave up waiting (raised on env 0); the step is incomplete */ };

/* action -> control map applied inside myo_step (base_v0.py:87-91) */
enum {
  MYO_ACTMAP_NONE = 0,
  MYO_ACTMAP_MUSCLE_SIGMOID = 1,
  MYO_ACTMAP_SIGMOID_FATIGUE = 2,          /* + muscle condition "fatigue": 3CC-r model, envs/myo/fatigue.py:61-108, base_v0.py:100-104 */
  MYO_ACTMAP_SIGMOID_REAFFERENTATION = 3,  /* + EIP -> EPL tendon transfer, base_v0.py:105-109 (ids set with myo_batch_set_condition) */
  MYO_ACTMAP_CTRLRANGE = 4                 /* TrackEnv.step, mjx/myodm_v0.py:272-275: ctrl = (a + 1) (hi - lo) / 2 + lo over actuator_ctrlrange; with
                                              MYO_TASK_TRACK configured this map also switches the task's fused prologue / epilogue on */
};

/* tasks understood by myo_obs / myo_reset */
typedef enum myo_task { MYO_TASK_NONE = 0, MYO_TASK_POSE = 1, MYO_TASK_REACH = 2, MYO_TASK_WALK = 3,
                        MYO_TASK_STAND = 5, /* walk_v0.py:13-183 ReachEnvV0 (myoLegStandRandom-v0): reach with a site of the free root link; obs = qpos, qvel*dt,
                                               tip (3), target - tip (3), act; reward 10 - d - 10 |qvel dt| + bonus - 100 |act|/na - penalty */
                        MYO_TASK_TRACK = 6, /* MyoDM TrackEnv (mjx/myodm_v0.py:14-304): see myo_track_config */
                        MYO_TASK_HOLD = 4 /* ObjHoldFixedEnvV0 (envs/myo/myobase/obj_hold_v0.py:13-118): the model's LAST joint is the free
                                             object; obs = hand qpos, hand qvel*dt, object position, goal - object, act; target = goal (3) */
} myo_task;

typedef struct myo_task_config {
  int task;              /* myo_task */
  int frame_skip;        /* substeps per env step; obs scales qvel by frame_skip*timestep */
  int reset_random;      /* pose: 1 = qpos ~ U(jnt_range) (reset_type "random"), 0 = init_qpos */
  int target_generate;   /* 1 = sample target ~ U(target_lo, target_hi) on reset, 0 = fixed (target_lo) */
  int ntarget;           /* pose: nq; reach: 3*ntip; hold: 3 */
  int ntip;              /* reach: number of tip sites */
  int tip_site[8];       /* reach: site ids (compiled-model numbering) */
  float pose_thd, far_th, near_th;   /* hold: near_th = goal threshold (0.010), far_th = drop distance (0.300) */
  float w_pose, w_bonus, w_act_reg, w_penalty, w_reach;   /* hold: w_reach weighs goal_dist */
  const float* target_lo; /* host pointers, ntarget floats each (copied) */
  const float* target_hi;
  const float* init_qpos; /* host pointer, nq floats (copied); NULL = qpos0 */
  /* optional reset noise (walk_v0.py:152-167 generate_qpos): qpos = clip(init_qpos + U(noise_lo, noise_hi), clip_lo, clip_hi) per qpos
   * entry; host pointers, nq floats each (copied), all four or none (NULL) */
  const float *reset_noise_lo, *reset_noise_hi, *reset_clip_lo, *reset_clip_hi;
  const float* init_qvel; /* host pointer, nv floats (copied) or NULL = zero */
  float tip_lpos[3];      /* stand: the tip site's position in the root link's frame */
} myo_task_config;

/* walk task (WalkEnvV0: envs/myo/myobase/walk_v0.py:187-470, registered as myoLegWalk-v0 in envs/myo/myobase/__init__.py:443-459).
 * Observation = qpos[2:], qvel*dt, com_vel(2), torso xquat(4), feet heights(2), COM height, feet - pelvis(6), phase, muscle
 * length / clipped velocity / clipped force/1000 (nu each), act (nu).  With this task configured, myo_step itself produces
 * MYO_F_OBS / REWARD / DONE / SOLVED at the post-step state (one fused pass); myo_obs recomputes them without stepping. */
typedef struct myo_walk_config {
  int frame_skip, hip_period;
  float min_height, max_rot, target_x_vel, target_y_vel;
  float target_rot[4];                                         /* reference root quaternion (walk_v0.py:425-431) */
  int body_talus_l, body_talus_r, body_pelvis, body_torso;     /* compiled-model body ids */
  int qadr_hip_flexion_l, qadr_hip_flexion_r;                  /* qpos addresses (walk_v0.py:413-420) */
  int qadr_joint_angle[4];                                     /* hip_adduction_l, hip_adduction_r, hip_rotation_l, hip_rotation_r */
  float w_vel_reward, w_done, w_cyclic_hip, w_ref_rot, w_joint_angle_rew;
  const float* init_qpos;                                      /* host, nq floats: reset pose (reset_type "init": key_qpos[2]) */
  const float* init_qvel;                                      /* host, nv floats or NULL (zero) */
  /* TerrainEnvV0 (walk_v0.py:490-671, myoLeg{Rough,Hilly,Stair}TerrainWalk-v0), terrain models only; all zero for WalkEnvV0 */
  float knee_height;                                           /* > 0: also done when COM height - mean feet height < this (0.61, :660-671) */
  int terrain;                                                 /* myo_terrain: elevation grid re-drawn at every reset of an env (:563-622) */
  float terrain_scalar_lo, terrain_scalar_hi;                  /* hilly / stairs height scale ~ U(lo, hi); variant "fixed": lo == hi */
  /* reset_type "random" (WalkEnvV0.get_randomized_initial_state, walk_v0.py:316-332): with probability 1/2 the alternative keyframe
   * (init = key 2, alt = key 3), then qpos += N(0, reset_noise_std) on every coordinate except the root height and quaternion */
  const float* init_qpos_alt;                                  /* host, nq floats or NULL (reset_type "init") */
  const float* init_qvel_alt;                                  /* host, nv floats or NULL */
  float reset_noise_std;                                       /* 0.02 in the reference */
} myo_walk_config;
/* MyoDM TrackEnv (mjx/myodm_v0.py:14-304; the env the reference runs on MJX), models of the TrackEnv class (myohand_object_*).  With this
 * task configured, ONE myo_step(action, MYO_ACTMAP_CTRLRANGE, n_frames) is a whole env step:
 *   ctrl = (action + 1) (hi - lo) / 2 + lo (:272-275); reference row looked up at the pre-step time + motion_start_time (:278-279,
 *   mjx/reference_motion.py:7-313 == logger/reference_motion.py, its arithmetic kept: see `interpolation_linear`); n_frames substeps;
 *   MYO_F_OBS = [qpos, qvel] (:297-304); MYO_F_REWARD / DONE / METRICS = compute_reward on the stepped state with the body frames of the last
 *   substep's position stage (:185-267); with `autoreset`, envs that are done go back to init_qpos / zero velocity, activation, time and
 *   their observation row is the first one of the new episode.
 * myo_reset puts envs at init_qpos (TrackEnv.reset :152-173); myo_obs writes [qpos, qvel] without stepping. */
typedef struct myo_track_config {
  int n_frames;                            /* physics substeps per env step (5, :41-46) */
  int ref_type;                            /* 0 FIXED (1 row), 1 RANDOM (2 rows: low / high), 2 TRACK (> 2 rows) (reference_motion.py:64-75) */
  int horizon, robot_horizon, object_horizon, robot_dim, object_dim;
  int motion_extrapolation;                /* hold the last frame beyond the motion's end */
  int interpolation_linear;                /* 0: the reference's between-frame arithmetic, (1 - b) ** x[i] + b x[i+1] with b = t - T[i] / dt; 1: linear */
  double motion_start_time;
  const double *ref_time, *ref_robot, *ref_robot_vel /* or NULL */, *ref_object;   /* host; [horizon], [robot_horizon][robot_dim] x2, [object_horizon][object_dim] */
  const float* init_qpos;                  /* host, nq */
  const float *ctrl_lo, *ctrl_hi;          /* host, nu: actuator_ctrlrange */
  int object_link, wrist_link;             /* kinematic links carrying the object body and the wrist (lunate) body */
  float object_ipos[3], object_imat[9], wrist_ipos[3];   /* the bodies' inertial frames (xipos / ximat) inside those links */
  float lift_z;                            /* object height that earns the lift bonus (:137-139) */
  float obj_err_scale, base_err_scale, lift_bonus_mag, qpos_reward_weight, qpos_err_scale, qvel_reward_weight, qvel_err_scale;   /* :104-128 */
  float obj_fail_thresh, base_fail_thresh, qpos_fail_thresh;
  int terminate_obj_fail, terminate_pose_fail;
  float w_pose, w_object, w_bonus, w_penalty;   /* DEFAULT_RWD_KEYS_AND_WEIGHTS :16-21 */
  int autoreset;
  int max_episode_steps;                   /* gym TimeLimit of the registered MyoDM ids (envs/myo/myodm/__init__.py:571,649; 0: none): MYO_F_SOLVED = 1 when the
                                              episode is truncated by it (and not done); with `autoreset` such envs are reset as well */
  uint64_t seed;                           /* RANDOM references: draws keyed by (seed, global env id, env step) */
} myo_track_config;

typedef enum myo_terrain { MYO_TERRAIN_NONE = 0, MYO_TERRAIN_ROUGH = 1, MYO_TERRAIN_HILLY = 2, MYO_TERRAIN_STAIRS = 3 } myo_terrain;

const char* myo_last_error(void);
int myo_version(void);

int myo_model_load(const void* blob, size_t nbytes, int device, myo_model** out);
void myo_model_free(myo_model*);
int myo_model_dims(const myo_model*, myo_dims* out);
/* test switches mirroring the oracle's (disable contacts / limits / ellipsoid pairs) */
int myo_model_set_switch(myo_model*, int disable_contact, int disable_limit, int disable_ellipsoid);

int myo_batch_create(const myo_model*, int B, myo_batch** out);
void myo_batch_free(myo_batch*);
int myo_batch_size(const myo_batch*);
int myo_batch_configure(myo_batch*, const myo_task_config* cfg);
int myo_batch_configure_walk(myo_batch*, const myo_walk_config* cfg);
int myo_batch_configure_track(myo_batch*, const myo_track_config* cfg);
/* muscle conditions (envs/myo/base_v0.py:61-80): time step of the fatigue model (frame_skip * timestep) and the actuator ids of
 * the EIP -> EPL tendon transfer (-1: none).  Sarcopenia is a model edit (peak force of gainprm halved) made before myo_model_load. */
int myo_batch_set_condition(myo_batch*, int frame_skip, int epl_actuator, int eip_actuator);
/* fatigue state at reset (CumulativeFatigue.reset, envs/myo/fatigue.py:114-134; BaseV0 kwargs fatigue_reset_vec / fatigue_reset_random,
 * base_v0.py:30-31,120-127): mode 0 all rested (MA 0, MR 1, MF 0); 1 random (u1, u2 ~ U(0,1): MA = u1 u2, MR = u1 (1 - u2), MF = 1 - u1);
 * 2 a given fatigue vector (host, nu floats; MF = vec, MR = 1 - vec, MA = 0) */
int myo_batch_set_fatigue_reset(myo_batch*, int mode, const float* fatigue_vec);
/* per-env size of ONE collision geom (compiled-model geom id), as ObjHoldRandomEnvV0.reset edits model.geom_size per episode
 * (envs/myo/myobase/obj_hold_v0.py:122-140): every reset of an env draws size ~ U(lo, hi) per axis; mass and inertia are untouched, as in
 * the reference (the model is not recompiled).  lo == hi == NULL switches the override off.  Generic large-kernel models only. */
int myo_batch_set_geom_override(myo_batch*, int geom_id, const float* size_lo, const float* size_hi);
/* device pointer + pitch (elements per env row) of a field */
int myo_batch_field(myo_batch*, int field, void** dev_ptr, size_t* pitch, size_t* width);
/* synchronous host copies (tests / plumbing without torch); host buffers are [B][width] */
int myo_batch_read(myo_batch*, int field, void* host, size_t nbytes);
int myo_batch_write(myo_batch*, int field, const void* host, size_t nbytes);

/* reset envs whose mask byte is nonzero (mask == NULL: all); mask is a DEVICE pointer, B bytes */
int myo_reset(myo_batch*, const uint8_t* mask_dev, uint64_t seed, void* stream);
/* overwrite state from device arrays [B][n] (NULL = leave field unchanged) */
int myo_set_state(myo_batch*, const float* qpos, const float* qvel, const float* act, const float* time, void* stream);
/* one env step: ctrl = actmap(action[B][nu]) (action == NULL: keep MYO_F_CTRL), then nsubsteps physics substeps */
int myo_step(myo_batch*, const float* action_dev, int actmap, int nsubsteps, void* stream);
/* observation / reward / done for the configured task into MYO_F_OBS / REWARD / DONE / SOLVED */
int myo_obs(myo_batch*, void* stream);
/* observation vector only (reward / done / solved untouched): used after an auto-reset */
int myo_obs_only(myo_batch*, void* stream);
/* observation vector only, and only for envs whose elapsed-step counter is 0 (i.e. those myo_autoreset just reset) */
int myo_obs_reset_only(myo_batch*, void* stream);
/* gym TimeLimit + done handling (envs/myo/myobase/__init__.py max_episode_steps): reset every env whose done flag is
 * set or whose elapsed env-step counter reached max_episode_steps, with the configured reset / target sampling */
int myo_autoreset(myo_batch*, int max_episode_steps, uint64_t seed, void* stream);
/* global id of this batch's env 0 (multi-GPU sharding: RNG streams are keyed by global env id) */
int myo_set_env_offset(myo_batch*, int env_offset);
/* copy per-env int32 flags to host and clear them */
int myo_status(myo_batch*, int32_t* host_flags);
/* fill action[B][nu] with U(-1,1) from a counter-based generator (seed, step, global env id) */
int myo_random_action(myo_batch*, float* action_dev, uint64_t seed, uint64_t step, int env_offset, void* stream);
int myo_sync(void* stream);
/* cost-sorted workgroup -> env placement for the wave-per-env kernel (default on; speed only, results unchanged) */
int myo_set_balance(myo_batch*, int on);
/* lanes cooperating on one env (16, 32 or 64; default 16 or $MYO_LANES) -- a tuning knob, results are identical up to float round-off */
int myo_set_lanes(int lanes);
/* diagnostic build (-DMYO_STAMPS=1) only: per-workgroup clock64 totals per kernel stage; returns 1 in the normal build */
int myo_read_stamps(myo_batch*, long long* host, int nwg);

/* timing helper for bench.py: run `steps` env steps on `stream` bracketed by HIP events recorded on that same
 * stream; returns elapsed milliseconds in *ms_out.  mode bits select what runs inside the timed region. */
enum { MYO_BENCH_OBS = 1, MYO_BENCH_FRESH_ACTIONS = 2, MYO_BENCH_AUTORESET = 4 };
int myo_bench_rollout(myo_batch*, int steps, int nsubsteps, uint64_t seed, int mode, int max_episode_steps, void* stream, float* ms_out);
/* ms_out == NULL makes myo_bench_rollout asynchronous: launches are enqueued on `stream` and nothing waits (a multi-GPU caller puts
 * its observation all-gather on the same stream between calls).
 * myo_bench_last_kernel_ms: HIP-event milliseconds spent inside the step-kernel launches of the last synchronous call, or of all
 * asynchronous calls since the previous collection (waits for them); events are recorded on the launches' own stream */
int myo_bench_last_kernel_ms(myo_batch*, float* ms_out);
/* roofline probe (no reference counterpart): measured chip-wide issue rate of wave64 v_fma_f32 instructions per second with
 * `waves_per_simd` (1..4, or 8 = two workgroups per CU) waves resident on every SIMD; bench.py prices the step kernel's VALU instruction
 * count against the figure at 4, the occupancy the hand kernel runs at */
int myo_probe_valu(int device, int waves_per_simd, int iters, double* wave_insts_per_s, int* n_cu_out);
/* name of the step-kernel template instantiation the last launch used, spelled as rocprofv3 prints it */
const char* myo_bench_last_kernel_name(const myo_batch*);

/* ---- policy inference: the network family of the reference's `mjx_brax_policy` artefact (a brax PPO
 * policy: running-statistics observation normalisation, MLP with swish hidden layers, tanh-normal head; brax is third-party and
 * absent from the reference tree, so the formulas are restated from its documentation -- parity unpinned).
 *   x = (obs - mean) / std ; h_{l+1} = swish(h_l W_l + b_l) ; (loc, raw) = split(h_L W_L + b_L) ;
 *   action = tanh(loc)                                   (deterministic)
 *          = tanh(loc + (softplus(raw) + 0.001) * eps)    (sampled; eps ~ N(0,1) from the counter RNG (seed, step, global env id))
 * Kernels are row-major [in][out] float32, copied at load time.  obs / action are DEVICE pointers, env-major, e.g. the
 * MYO_F_OBS buffer and the action argument of myo_step: a rollout never leaves the GPU. */
typedef struct myo_policy myo_policy;
int myo_policy_load(int device, int obs_dim, int act_dim, int nlayers, const int* layer_out /* nlayers entries, last = 2*act_dim */,
                    const float* obs_mean, const float* obs_std, const float* const* kernels, const float* const* biases, myo_policy** out);
void myo_policy_free(myo_policy*);
int myo_policy_act(myo_policy*, const float* obs_dev, int B, float* action_dev, int deterministic, uint64_t seed, uint64_t step,
                   int env_offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif
