"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's WalkEnvV0 observation / reward
(/root/reference/myosuite/envs/myo/myobase/walk_v0.py:268-316 get_obs_dict / get_reward_dict, :354-470 helpers) on top of the
C oracle's forward pass.  Used by tests/ and __graft_entry__.smoke() as the checker for the fused HIP observation pass; never
imported by the product path.  Parity unpinned: the reference holds no golden observations for myoLegWalk-v0."""
import numpy as np

WEIGHTS = dict(vel_reward=5.0, done=-100.0, cyclic_hip=-10.0, ref_rot=10.0, joint_angle_rew=5.0)   # walk_v0.py:203-209


def quat2mat00(q):
    """[0,0] entry of myosuite.utils.quat_math.quat2mat (quat_math.py:160-183)."""
    return 1.0 - 2.0 * (q[2] * q[2] + q[3] * q[3]) / float(np.dot(q, q))


def walk_obs_reward(m, o, steps, dt, target_rot, hip_period=100, min_height=0.8, max_rot=0.8, target_x_vel=0.0, target_y_vel=1.2,
                    weights=WEIGHTS):
    """Observation vector (obs_keys order of walk_v0.py:189-201 plus 'act', base_v0.py appends it), dense reward, done, solved for
    the oracle's CURRENT state (o.forward() is run here, like Robot.sensor2sim -> sim.forward(), robot.py:573-598)."""
    o.forward()
    qpos, qvel, act = o.field("qpos").copy(), o.field("qvel").copy(), o.field("act").copy()
    nb = len(m.body_mass)
    xpos = o.field("xpos").reshape(nb, 3)
    xquat = o.field("xquat").reshape(nb, 4)
    xipos = o.field("xipos").reshape(nb, 3)
    cvel = o.field("cvel").reshape(nb, 6)
    mass = np.asarray(m.body_mass)[:, None]
    bid = lambda n: m.name2id("body", n)
    jq = lambda n: qpos[m.jnt_qposadr[m.name2id("joint", n)]]
    com_vel = (np.sum(mass * -cvel, 0) / np.sum(mass))[3:5]                                  # walk_v0.py:438-444
    com = np.sum(mass * xipos, 0) / np.sum(mass)                                              # walk_v0.py:465-470
    feet = np.array([xpos[bid("talus_l")][2], xpos[bid("talus_r")][2]])                     # walk_v0.py:382-393
    rel = np.concatenate([xpos[bid("talus_l")] - xpos[bid("pelvis")], xpos[bid("talus_r")] - xpos[bid("pelvis")]])
    phase = (steps / hip_period) % 1                                                          # walk_v0.py:279
    obs = np.concatenate([qpos[2:], qvel * dt, com_vel, xquat[bid("torso")], feet, [com[2]], rel, [phase],
                          o.field("actuator_length"), np.clip(o.field("actuator_velocity"), -100, 100),
                          np.clip(o.field("actuator_force") / 1000, -100, 100), act])
    vel_reward = np.exp(-np.square(target_y_vel - com_vel[1])) + np.exp(-np.square(target_x_vel - com_vel[0]))   # :395-403
    des = np.array([0.8 * np.cos(phase * 2 * np.pi + np.pi), 0.8 * np.cos(phase * 2 * np.pi)], dtype=np.float32)
    cyclic = np.linalg.norm(des - np.array([jq("hip_flexion_l"), jq("hip_flexion_r")]))     # :405-420
    ref_rot = np.exp(-np.linalg.norm(5.0 * (qpos[3:7] - target_rot)))                          # :422-431
    ang = np.array([jq(n) for n in ("hip_adduction_l", "hip_adduction_r", "hip_rotation_l", "hip_rotation_r")])
    ja = np.exp(-5 * np.mean(np.abs(ang)))                                                     # :371-378
    done = float(com[2] < min_height or abs(quat2mat00(qpos[3:7])) > max_rot)                 # :362-369, 452-463
    rwd = dict(vel_reward=vel_reward, done=done, cyclic_hip=cyclic, ref_rot=ref_rot, joint_angle_rew=ja)
    dense = sum(weights[k] * rwd[k] for k in weights)
    return obs, dense, done, float(vel_reward >= 1.0), rwd
