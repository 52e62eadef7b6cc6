"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol the header declares;
the product path fails loudly (no CPU fallback) when no MI355X is present; env registry mirrors the reference."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from myosuite_mjx_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        capi.build_library()
    return capi.LIB_PATH


def test_library_exports_every_declared_symbol(lib_path):
    hdr = open(os.path.join(ROOT, "include", "myo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(myo_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_error_convention_without_gpu(lib_path, hand):
    """Bad arguments return negative codes with a message; without a GPU model upload raises (no silent fallback)."""
    import torch
    from myosuite_mjx_amd import capi
    L = capi.lib()
    assert L.myo_model_load(None, 0, 0, None) == -1 and b"bad arguments" in L.myo_last_error()
    h = ctypes.c_void_p()
    assert L.myo_model_load(b"XXXX" + bytes(32), 36, 0, ctypes.byref(h)) == -2
    if not torch.cuda.is_available():
        with pytest.raises(capi.MyoError):
            capi.HipModel(hand.blob(), 0)
        import myosuite_mjx_amd as myo
        with pytest.raises(capi.MyoError):
            myo.make("myoHandPoseFixed-v0", num_envs=2)


def test_product_does_not_import_oracle():
    """The shipped package must not reference the oracle (test infrastructure)."""
    pkg = os.path.join(ROOT, "myosuite_mjx_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "libmyo_oracle" not in src, f


def test_registry_matches_reference_specs():
    from myosuite_mjx_amd.envs import ASL_QPOS, HAND_POSE_FIXED, REGISTRY, UNSUPPORTED
    for k in ("myoHandPoseFixed-v0", "myoHandPoseRandom-v0", "myoHandReachFixed-v0", "myoHandReachRandom-v0", "myoHandPose3Fixed-v0"):
        assert k in REGISTRY and REGISTRY[k]["max_episode_steps"] == 100 and REGISTRY[k]["frame_skip"] == 10
    r = REGISTRY["myoHandPoseRandom-v0"]
    assert r["reset_type"] == "random" and r["target_type"] == "generate" and r["pose_thd"] == 0.7
    assert np.allclose(r["target_lo"], ASL_QPOS.min(0)) and np.allclose(r["target_hi"], ASL_QPOS.max(0))
    assert r["target_lo"][:3].tolist() == [0, 0, 0] and abs(r["target_hi"][21] - 1.571) < 1e-9
    f = REGISTRY["myoHandPoseFixed-v0"]
    assert np.allclose(f["target_lo"], HAND_POSE_FIXED) and f["reset_type"] == "init"
    rr = REGISTRY["myoHandReachRandom-v0"]
    assert rr["far_th"] == 0.034 and np.allclose(rr["target_lo"][:3], [-0.185, -0.577, 1.455]) and np.allclose(rr["target_hi"][:3], [-0.125, -0.517, 1.535])
    for tid, kind, sc in (("myoLegRoughTerrainWalk-v0", "rough", (0.0, 0.0)), ("myoLegHillyTerrainWalk-v0", "hilly", (0.63, 0.63)),
                          ("myoLegStairTerrainWalk-v0", "stairs", (2.5, 2.5))):      # envs/myo/myobase/__init__.py:462-520, walk_v0.py:569-597
        t = REGISTRY[tid]
        assert t["model"] == "myolegs_terrain" and t["terrain"] == kind and t["terrain_scalar"] == sc and t["knee_height"] == 0.61
        assert t["max_episode_steps"] == 1000 and t["weights"] == REGISTRY["myoLegWalk-v0"]["weights"] and tid not in UNSUPPORTED
    lw = REGISTRY["myoLegWalk-v0"]                               # envs/myo/myobase/__init__.py:443-459, walk_v0.py:203-209
    assert lw["model"] == "myolegs" and lw["max_episode_steps"] == 1000 and lw["frame_skip"] == 10 and lw["reset_type"] == "init"
    assert (lw["min_height"], lw["max_rot"], lw["hip_period"], lw["target_x_vel"], lw["target_y_vel"]) == (0.8, 0.8, 100, 0.0, 1.2)
    assert lw["weights"] == dict(vel_reward=5.0, done=-100.0, cyclic_hip=-10.0, ref_rot=10.0, joint_angle_rew=5.0)
    ff = REGISTRY["myoFingerPoseFixed-v0"]                       # envs/myo/myobase/__init__.py:222-236
    assert ff["model"] == "myofinger_v0" and ff["pose_thd"] == 0.35 and ff["target_lo"].tolist() == [0, 0, 0.75, 0.75]
    eb = REGISTRY["myoElbowPose1D6MRandom-v0"]                   # envs/myo/myobase/__init__.py:123-137
    assert eb["model"] == "myoelbow_1dof6muscles" and eb["pose_thd"] == 0.175 and eb["reset_type"] == "random"
    assert eb["target_lo"].tolist() == [0.0] and eb["target_hi"].tolist() == [2.27] and REGISTRY["myoElbowPose1D6MFixed-v0"]["target_lo"].tolist() == [2.0]
    fr = REGISTRY["myoFingerReachRandom-v0"]                     # envs/myo/myobase/__init__.py:94-105; far_th default reach_v0.py:47
    assert fr["tips"] == ("IFtip",) and fr["far_th"] == 0.35 and fr["target_lo"].tolist() == [0.1, -0.1, 0.1] and fr["target_hi"].tolist() == [0.27, 0.1, 0.3]
    assert "myoSarcElbowPose1D6MFixed-v0" in REGISTRY and "myoFatiFingerReachRandom-v0" in REGISTRY
    ls = REGISTRY["myoLegStandRandom-v0"]                        # envs/myo/myobase/__init__.py:424-441, walk_v0.py:15-21,105
    assert ls["max_episode_steps"] == 150 and ls["joint_random_range"] == (-0.2, 0.2) and ls["far_th"] == 0.44 and ls["near_th"] == 0.05
    assert ls["weights"] == dict(reach=1.0, bonus=4.0, penalty=50.0, act_reg=1.0) and ls["target_span"] == ((-0.05, -0.05, 0.0), (0.05, 0.05, 0.0))
    oh = REGISTRY["myoHandObjHoldFixed-v0"]                      # envs/myo/myobase/__init__.py:596-604, obj_hold_v0.py:16-20,96-97
    assert oh["max_episode_steps"] == 75 and oh["weights"] == dict(goal_dist=100.0, bonus=4.0, penalty=10.0, act_reg=0.0) and (oh["goal_th"], oh["drop_th"]) == (0.010, 0.300)
    ohr = REGISTRY["myoHandObjHoldRandom-v0"]                    # obj_hold_v0.py:121-140
    assert ohr["goal"] is None and ohr["goal_span"] == 0.030 and ohr["object_size"] == ((0.020,) * 3, (0.030,) * 3)
    assert "myoElbowPose1D6MExoRandom-v0" in UNSUPPORTED


@pytest.mark.skipif(not os.path.isdir("/root/reference/myosuite"), reason="reference tree not present")
def test_registry_numbers_against_reference_text():
    """The ASL table / fixed target are data copied from the reference registration; check them against its text."""
    from myosuite_mjx_amd.envs import ASL_QPOS, HAND_POSE_FIXED
    txt = open("/root/reference/myosuite/envs/myo/myobase/__init__.py").read()
    rows = re.findall(r'ASL_qpos\[(\d)\] = \(\s*"([^"]+)"', txt)
    assert len(rows) == 10
    for k, s in rows:
        assert np.allclose(np.array(s.split(" "), float), ASL_QPOS[int(k)])
    blk = txt[txt.index('id="myoHandPoseFixed-v0"'):]
    blk = blk[blk.index("np.array("):blk.index("),", blk.index("np.array("))]
    vals = np.array(re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", blk.replace("np.array(", "")), float)
    assert np.allclose(vals[:23], HAND_POSE_FIXED)


def test_host_rng_twin_is_uniform():
    from myosuite_mjx_amd.shard import u01
    v = np.array([u01(7, a, 3) for a in range(4000)])
    assert 0 <= v.min() and v.max() < 1 and abs(v.mean() - 0.5) < 0.02 and abs(v.var() - 1 / 12) < 0.01
    assert u01(7, 5, 3) == u01(7, 5, 3) and u01(7, 5, 3) != u01(8, 5, 3)


def test_public_header_is_plain_c(tmp_path):
    """include/myo_hip.h is the drop-in boundary: it must compile as C99 (no torch / C++ types in the signatures) and as C++."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "include/myo_hip.h"\nint main(void) { myo_task_config c; myo_walk_config w; myo_dims d; (void)c; (void)w; (void)d; return MYO_F_COUNT > 0 ? 0 : 1; }\n')
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", root, "-c", str(src), "-o", str(tmp_path / "a.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", root, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "b.o")])
