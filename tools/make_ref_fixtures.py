"""Golden fixtures generated FROM THE REFERENCE, in the build container only.

Nothing under /root/reference travels to the GPU box: this script imports three pure-numpy reference modules *by file path*
(bypassing `myosuite/__init__`, which needs gym), feeds them seeded inputs and commits only inputs + outputs as small data
files under tests/golden/.  It also extracts data the reference's own model files hold (MuJoCo-computed `lengthrange`
values, the edit history its XML comments keep) and evaluates the model files with an XML walk that shares no code with
`myosuite_mjx_amd/mjcf.py`.

    python tools/make_ref_fixtures.py          # writes tests/golden/ref_*.npz / *.json

Reference modules imported (never copied):  myosuite/utils/quat_math.py, myosuite/envs/obs_vec_dict.py,
myosuite/logger/reference_motion.py (the numpy twin of mjx/reference_motion.py, which needs jax; the reference's own
tests/mjx/test_reference_motion.py asserts the two agree)."""
import contextlib
import importlib.util
import io
import json
import math
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MYO_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def _load(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------ quat_math
def quat_math_fixture():
    qm = _load("myosuite/utils/quat_math.py", "ref_quat_math")
    rng = np.random.default_rng(20261004)
    N = 64
    q = rng.normal(size=(N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q2 = rng.normal(size=(N, 4))
    q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    q[0] = [1, 0, 0, 0]
    q[1] = [0, 1, 0, 0]
    q[2] = [math.sqrt(0.5), 0, 0, math.sqrt(0.5)]
    eul = rng.uniform(-math.pi, math.pi, (N, 3))
    eul[:, 1] = rng.uniform(-1.5, 1.5, N)          # away from the gimbal pole (the reference switches formula there)
    vec = rng.normal(size=(N, 3))
    axis = rng.normal(size=(N, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = rng.uniform(-3, 3, N)
    out = dict(q=q, q2=q2, euler=eul, vec=vec, axis=axis, angle=ang)
    out["mulQuat"] = np.stack([qm.mulQuat(a, b) for a, b in zip(q, q2)])
    out["negQuat"] = np.stack([qm.negQuat(a) for a in q])
    out["diffQuat"] = np.stack([qm.diffQuat(a, b) for a, b in zip(q, q2)])
    sp, ax = zip(*[qm.quat2Vel(a, 0.02) for a in q])
    out["quat2Vel_speed"], out["quat2Vel_axis"] = np.array(sp), np.stack(ax)
    sp, ax = zip(*[qm.quatDiff2Vel(a, b, 1) for a, b in zip(q, q2)])
    out["quatDiff2Vel_speed"], out["quatDiff2Vel_axis"] = np.array(sp), np.stack(ax)
    out["axis_angle2quat"] = np.stack([qm.axis_angle2quat(a, t) for a, t in zip(axis, ang)])
    out["euler2mat"] = qm.euler2mat(eul)
    out["euler2quat"] = qm.euler2quat(eul)
    out["quat2mat"] = qm.quat2mat(q)
    out["mat2quat"] = qm.mat2quat(out["quat2mat"])
    out["mat2euler"] = qm.mat2euler(out["euler2mat"])
    out["quat2euler"] = qm.quat2euler(q)
    out["rotVecQuat"] = np.stack([qm.rotVecQuat(v, a) for v, a in zip(vec, q)])
    out["rotVecMatT"] = np.stack([qm.rotVecMatT(v, m) for v, m in zip(vec, out["quat2mat"])])
    out["quat2euler_intrinsic"] = np.stack([qm.quat2euler_intrinsic(a) for a in q])
    out["intrinsic_euler2quat"] = np.stack([qm.intrinsic_euler2quat(e) for e in eul])
    # the walk env's fall test (walk_v0.py:456-466): |(quat2mat(q) @ [1,0,0])[0]|
    out["walk_rot_x"] = np.abs((qm.quat2mat(q) @ np.array([1.0, 0, 0]))[:, 0])
    np.savez_compressed(os.path.join(OUT, "ref_quat_math.npz"), **out)
    return len(out)


# ------------------------------------------------------------------------------------------------ obs vector assembly
def obsvec_fixture():
    ov = _load("myosuite/envs/obs_vec_dict.py", "ref_obs_vec_dict")
    rng = np.random.default_rng(7)
    cases = {
        # key orders: pose_v0.py:15 / reach_v0.py:15 / walk_v0.py:186-198, each with "act" appended (base_v0.py:34-38)
        "hand_pose": (["qpos", "qvel", "pose_err", "act"], dict(qpos=23, qvel=23, pose_err=23, act=39)),
        "hand_reach": (["qpos", "qvel", "tip_pos", "reach_err", "act"], dict(qpos=23, qvel=23, tip_pos=15, reach_err=15, act=39)),
        "finger_pose": (["qpos", "qvel", "pose_err", "act"], dict(qpos=4, qvel=4, pose_err=4, act=5)),
        "leg_walk": (["qpos_without_xy", "qvel", "com_vel", "torso_angle", "feet_heights", "height", "feet_rel_positions",
                      "phase_var", "muscle_length", "muscle_velocity", "muscle_force", "act"],
                     dict(qpos_without_xy=33, qvel=34, com_vel=2, torso_angle=4, feet_heights=2, height=1, feet_rel_positions=(2, 3),
                          phase_var=1, muscle_length=80, muscle_velocity=80, muscle_force=80, act=80)),
    }
    out = {}
    meta = {}
    for name, (keys, dims) in cases.items():
        d = {"time": np.array([0.02])}
        for k, n in dims.items():
            d[k] = rng.normal(size=n)                       # float64 inputs, like sim.data views
        # extra keys present in the dict but not in obs_keys must be ignored
        d["unused_extra"] = rng.normal(size=3)
        o = ov.ObsVecDict()
        t, vec = o.obsdict2obsvec(d, keys)
        assert vec.dtype == np.float32
        for k in dims:
            out[f"{name}__in__{k}"] = d[k]
        out[f"{name}__obsvec"] = vec
        meta[name] = dict(keys=keys, dims={k: int(np.prod(v)) for k, v in dims.items()}, dtype=str(vec.dtype), obs_dim=int(vec.size),
                          key_idx={k: [o.key_idx[k].start, o.key_idx[k].stop] for k in keys})
    np.savez_compressed(os.path.join(OUT, "ref_obsvec.npz"), **out)
    with open(os.path.join(OUT, "ref_obsvec.json"), "w") as f:
        json.dump(meta, f, indent=1)
    return len(cases)


# ------------------------------------------------------------------------------------------------ reference motion lookup
def reference_motion_fixture():
    rm = _load("myosuite/logger/reference_motion.py", "ref_reference_motion")
    data_dir = os.path.join(REF, "myosuite/envs/myo/myodm/data")
    out = {}
    meta = {"cases": []}

    def run(tag, reference, times, extrapolation, seed=None, store_input=True):
        gen = np.random.default_rng(seed) if seed is not None else None
        with contextlib.redirect_stdout(io.StringIO()):
            ref = rm.ReferenceMotion(reference, motion_extrapolation=extrapolation, random_generator=gen)
            ri, oi = ref.get_init()
            rows = []
            for t in times:
                r = ref.get_reference(float(t))
                rows.append(r)
        if store_input:
            for k, v in ref.reference.items():
                if v is not None:
                    out[f"{tag}__in__{k}"] = np.asarray(v)
        out[f"{tag}__times"] = np.asarray(times, float)
        out[f"{tag}__robot"] = np.stack([np.asarray(r.robot, float) for r in rows])
        out[f"{tag}__object"] = np.stack([np.asarray(r.object, float) for r in rows])
        if rows[0].robot_vel is not None:
            out[f"{tag}__robot_vel"] = np.stack([np.asarray(r.robot_vel, float) for r in rows])
        out[f"{tag}__robot_init"], out[f"{tag}__object_init"] = np.asarray(ri, float), np.asarray(oi, float)
        meta["cases"].append(dict(tag=tag, type=ref.type.name, horizon=int(ref.horizon), robot_dim=int(ref.robot_dim), object_dim=int(ref.object_dim),
                                  extrapolation=bool(extrapolation), seed=seed))

    # TRACK: two of the reference's own motion files (envs/myo/myodm/data/*.npz: time[100], robot[100,29], object[100,7], inits)
    for stem in ("MyoHand_airplane_fly1", "MyoHand_cup_drink1"):
        d = {k: v for k, v in np.load(os.path.join(data_dir, stem + ".npz")).items()}
        T = d["time"]
        # exact frames walked in order (the heuristic index cache only moves forward by one), then held past the end (extrapolation)
        times = list(T[:40]) + [float(T[-1]), float(T[-1]) + 0.5]
        run(f"track_{stem}", d, [T[0]] + list(T[1:40]), True)
        run(f"trackend_{stem}", d, [T[-1] + 0.25, T[-1] + 1.0], True, store_input=False)
        # between-frame lookups: the reference's "blend" (logger/reference_motion.py:271-300 == mjx/reference_motion.py:271-279) is
        # blend = time - t[i] / (t[i+1] - t[i]);  robot = (1 - blend) ** robot[i] + blend * robot[i+1];  object linear in blend
        mid = [float(0.5 * (T[0] + T[1])), float(T[1]), float(T[1] + 0.25 * (T[2] - T[1])), float(T[2])]
        run(f"blend_{stem}", d, mid, True, store_input=False)
        del times
    # FIXED and RANDOM: the module-level reference of mjx/myodm_v0.py:308-318 (2 rows => RANDOM) and a 1-row FIXED one
    dof = 29
    rnd = dict(time=np.array([0.0, 4.0]), robot=np.zeros((2, dof)), robot_vel=np.zeros((2, dof)),
               object_init=np.array([0.0, 0.0, 0.1, 1.0, 0.0, 0.0, 0.0]),
               object=np.array([[-0.2, -0.2, 0.1, 1.0, 0.0, 0.0, -1.0], [0.2, 0.2, 0.1, 1.0, 0.0, 0.0, 1.0]]))
    run("random_myodm_default", rnd, [0.0, 0.01, 0.02, 3.0], True, seed=123)
    fix = dict(time=np.array([0.0]), robot=np.linspace(-1, 1, dof)[None], robot_vel=np.zeros((1, dof)),
               object=np.array([[0.1, -0.2, 0.3, 0.5, 0.5, 0.5, 0.5]]))
    run("fixed", fix, [0.0, 0.5, 10.0], False)
    np.savez_compressed(os.path.join(OUT, "ref_motion.npz"), **out)
    with open(os.path.join(OUT, "ref_motion.json"), "w") as f:
        json.dump(meta, f, indent=1)
    return len(meta["cases"])


def reference_motion_all_fixture():
    """Every motion file of envs/myo/myodm/data (95): the reference's answers at five sampled times each -- two exact frames, two times between
    frames (the operator-precedence / power arithmetic of logger/reference_motion.py:271-300) and one past the end (held) -- plus what its
    RANDOM-type constructor makes of a reference WITHOUT robot_init / object_init (logger/reference_motion.py:88-93: scalar means; the mjx twin,
    mjx/reference_motion.py:86-90, takes the robot mean over axis 0)."""
    rm = _load("myosuite/logger/reference_motion.py", "ref_reference_motion")
    data_dir = os.path.join(REF, "myosuite/envs/myo/myodm/data")
    out, stems, rejected = {}, [], []
    for path in sorted(os.listdir(data_dir)):
        if not path.endswith(".npz"):
            continue
        stem = path[:-4]
        d = {k: v for k, v in np.load(os.path.join(data_dir, path)).items()}
        T = d["time"]
        n = len(T)
        i, j = n // 3, (2 * n) // 3
        times = [float(T[i]), float(0.5 * (T[i] + T[i + 1])), float(T[j]), float(T[j] + 0.25 * (T[j + 1] - T[j])), float(T[-1] + 0.3)]
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ref = rm.ReferenceMotion(dict(d), motion_extrapolation=True)
                rows = [ref.get_reference(t) for t in times]              # (the heuristic index cache only ever moves forward: times ascend)
        except AssertionError as e:                                       # files the reference's own check_format refuses (object_init of 6 entries)
            rejected.append(stem)
            continue
        out[stem + "__times"] = np.asarray(times)
        out[stem + "__robot"] = np.stack([np.asarray(r.robot, float) for r in rows])
        out[stem + "__object"] = np.stack([np.asarray(r.object, float) for r in rows])
        stems.append(stem)
    rnd = dict(time=np.array([0.0, 4.0]), robot=np.linspace(-0.3, 0.5, 58).reshape(2, 29), robot_vel=np.zeros((2, 29)),
               object=np.array([[-0.2, -0.2, 0.1, 1.0, 0.0, 0.0, -1.0], [0.2, 0.2, 0.1, 1.0, 0.0, 0.0, 1.0]]))
    with contextlib.redirect_stdout(io.StringIO()):
        ref = rm.ReferenceMotion({k: v.copy() for k, v in rnd.items()}, motion_extrapolation=True, random_generator=np.random.default_rng(1))
        ri, oi = ref.get_init()
    for k, v in rnd.items():
        out["noinit__in__" + k] = v
    out["noinit__robot_init_numpy_twin"], out["noinit__object_init"] = np.asarray(ri, float), np.asarray(oi, float)
    out["stems"], out["rejected"] = np.array(stems), np.array(rejected)
    np.savez_compressed(os.path.join(OUT, "ref_motion_all.npz"), **out)
    return len(stems)


# ------------------------------------------------------------------------------------------------ MyoHand model files: goldens + edit history
HAND_ASSETS = "myosuite/simhive/myo_sim/hand/assets/myohand_assets.xml"
HAND_BODY = "myosuite/simhive/myo_sim/hand/assets/myohand_body.xml"


def _q_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _q_rot(q, v):
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return R @ v


def _elem_quat(at):
    """Orientation attributes used in the hand body file: quat, or euler (radians, intrinsic xyz: MuJoCo's default eulerseq)."""
    if "quat" in at:
        q = np.array(list(map(float, at["quat"].split())))
        return q / np.linalg.norm(q)
    if "euler" in at:
        e = list(map(float, at["euler"].split()))
        q = np.array([1.0, 0, 0, 0])
        for ax, a in zip(range(3), e):
            r = np.zeros(4)
            r[0], r[1 + ax] = math.cos(a / 2), math.sin(a / 2)
            q = _q_mul(q, r)
        return q
    for k in ("axisangle", "xyaxes", "zaxis"):
        assert k not in at, k
    return np.array([1.0, 0, 0, 0])


def hand_static_frames():
    """World positions of every site and wrap geom of the hand body tree at qpos0 (all joints at their reference 0), by a direct walk of
    myohand_body.xml: pos/quat (or euler) composition only.  Shares no code with mjcf.py / lowering.py / the oracle."""
    root = ET.parse(os.path.join(REF, HAND_BODY)).getroot()
    sites, geoms = {}, {}

    def walk(e, pos, quat):
        for c in e:
            if c.tag == "body":
                p = np.array(list(map(float, c.get("pos", "0 0 0").split())))
                walk(c, pos + _q_rot(quat, p), _q_mul(quat, _elem_quat(c.attrib)))
            elif c.tag == "site":
                p = np.array(list(map(float, c.get("pos", "0 0 0").split())))
                sites[c.get("name")] = pos + _q_rot(quat, p)
            elif c.tag == "geom" and c.get("class") == "wrap":
                p = np.array(list(map(float, c.get("pos", "0 0 0").split())))
                gq = _q_mul(quat, _elem_quat(c.attrib))
                geoms[c.get("name")] = np.concatenate([pos + _q_rot(quat, p), gq, [float(c.get("size").split()[0])], [1.0 if c.get("type") == "cylinder" else 0.0]])
    walk(root, np.zeros(3), np.array([1.0, 0, 0, 0]))
    return sites, geoms


def hand_model_fixture():
    body = open(os.path.join(REF, HAND_BODY)).read()
    assets = open(os.path.join(REF, HAND_ASSETS)).read()
    # (1) edit history kept in XML comments: former coordinates of sites that are still live
    live_sites = {m.group(1): [float(x) for x in m.group(2).split()]
                  for m in re.finditer(r'<site\s+name="([^"]+)"\s+pos="([^"]+)"', re.sub(r'<!--.*?-->', '', body, flags=re.S))}
    former = {}
    for cm in re.finditer(r'<!--(.*?)-->', body, re.S):
        line0 = body[:cm.start()].count("\n") + 1
        for s in re.finditer(r'<site\s+name="([^"]+)"\s+pos="([^"]+)"', cm.group(1)):
            name, pos = s.group(1), [float(x) for x in s.group(2).split()]
            if name in live_sites and max(abs(a - b) for a, b in zip(pos, live_sites[name])) > 1e-9:
                former[name] = dict(pos=pos, live=live_sites[name], line=line0 + cm.group(1)[:s.start()].count("\n"))
    # (2) MuJoCo-computed length ranges: the live block (myohand_assets.xml:501-539) and the commented-out older block (:540-578)
    live_txt, old_txt = assets.split('<!-- <muscle name="ECRL"')
    rx = r'<muscle name="(\w+)"[^>]*?lengthrange="([^"]+)"'
    live_lr = {m.group(1): [float(x) for x in m.group(2).split()] for m in re.finditer(rx, live_txt)}
    old_lr = {m.group(1): [float(x) for x in m.group(2).split()] for m in re.finditer(rx, '<muscle name="ECRL"' + old_txt)}
    assert len(live_lr) == 39 and len(old_lr) == 39
    # (3) tendon paths that lost a wrapping geom (commented-out <geom geom=.../> inside a <spatial>)
    removed = []
    for sp in re.finditer(r'<spatial[^>]*name="(\w+)"[^>]*>(.*?)</spatial>', assets, re.S):
        for g in re.finditer(r'<!--\s*<(geom|site) (?:geom|site)="([\w-]+)"', sp.group(2)):
            removed.append([sp.group(1), g.group(2)])
    # (4) former wrapping geoms (commented-out definitions of geoms that are still live under the same name)
    former_geoms = {}
    for cm in re.finditer(r'<!--(.*?)-->', body, re.S):
        line0 = body[:cm.start()].count("\n") + 1
        for g in re.finditer(r'<geom\s+name="(\w+_wrap)"([^>]*)>', cm.group(1)):
            at = dict(re.findall(r'(\w+)="([^"]*)"', g.group(2)))
            if re.search(r'<geom\s+name="%s"' % g.group(1), re.sub(r'<!--.*?-->', '', body, flags=re.S)):
                former_geoms[g.group(1)] = dict(pos=[float(x) for x in at["pos"].split()], quat=[float(x) for x in at["quat"].split()],
                                                size=[float(x) for x in at["size"].split()], type=at.get("type", "sphere"), line=line0)
    # (5) the live tendon paths as written in the file: ordered (kind, name, sidesite) triples per spatial tendon
    paths = {}
    for sp in re.finditer(r'<spatial[^>]*name="(\w+)"[^>]*>(.*?)</spatial>', re.sub(r'<!--.*?-->', '', assets, flags=re.S), re.S):
        paths[sp.group(1)] = [[k, n, sd] for k, n, sd in re.findall(r'<(site|geom)\s+(?:site|geom)="([^"]+)"(?:\s+sidesite="([^"]+)")?', sp.group(2))]
    sites, geoms = hand_static_frames()
    with open(os.path.join(OUT, "myohand_xml_goldens.json"), "w") as f:
        json.dump(dict(source=[HAND_ASSETS, HAND_BODY], lengthrange_live=live_lr, lengthrange_commented=old_lr, former_site_pos=former,
                       removed_wraps=removed, former_geoms=former_geoms, tendon_paths=paths), f, indent=1)
    np.savez_compressed(os.path.join(OUT, "myohand_static_frames.npz"), site_names=np.array(sorted(sites)), site_xpos=np.stack([sites[k] for k in sorted(sites)]),
                        geom_names=np.array(sorted(geoms)), geom_frame=np.stack([geoms[k] for k in sorted(geoms)]))
    return len(former), len(sites), len(geoms)


if __name__ == "__main__":
    if not os.path.isdir(os.path.join(REF, "myosuite")):
        sys.exit("reference tree not present: fixtures are generated in the build container only")
    os.makedirs(OUT, exist_ok=True)
    print("quat_math arrays:", quat_math_fixture())
    print("obsvec cases:", obsvec_fixture())
    print("reference-motion cases:", reference_motion_fixture())
    print("reference-motion files:", reference_motion_all_fixture())
    print("hand model (former sites, sites, wrap geoms):", hand_model_fixture())
