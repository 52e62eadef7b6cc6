"""Reads the MyoDM registration tables of the reference (envs/myo/myodm/__init__.py: the OBJECTS tuple and the MyoHand_task_spec entries -- ids, object
names, motion file names; parsed as text, nothing is imported or executed) and writes them as data:
  myosuite_mjx_amd/assets/myodm_tasks.json      {"objects": [...], "tasks": [[id, object, motion file], ...]}      (what envs.REGISTRY registers)
  tests/golden/myodm_grasp_frames.npz           twelve evenly spaced frames of one motion per object (rows copied from data/<motion>.npz,
                                                numpy.load without pickle): the states tests/test_gpu_track.py checks the objects' physics on
Run in the build container (needs /root/reference); the outputs are committed."""
import json, os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MYO_REFERENCE", "/root/reference")
src = open(os.path.join(REF, "myosuite/envs/myo/myodm/__init__.py")).read()
tasks = re.findall(r'task_spec\(\s*name="([^"]+)",\s*robot="MyoHand",\s*object="([^"]+)",\s*motion="([^"]+)"', src)
objects = re.findall(r'"([^"]+)"', re.search(r'OBJECTS = \((.*?)\)', src, re.S).group(1))
assert len(tasks) == 89 and len(objects) == 50, (len(tasks), len(objects))
data = os.path.join(REF, "myosuite/envs/myo/myodm/data")
assert all(os.path.exists(os.path.join(data, m)) for _, _, m in tasks)
json.dump({"objects": objects, "tasks": [list(t) for t in tasks]}, open(os.path.join(ROOT, "myosuite_mjx_amd", "assets", "myodm_tasks.json"), "w"), indent=0)
out = {}
old = np.load(os.path.join(ROOT, "tests", "golden", "myodm_grasp_frames.npz")) if os.path.exists(os.path.join(ROOT, "tests", "golden", "myodm_grasp_frames.npz")) else None
for obj in objects:
    if old is not None and obj + "__motion" in old.files and f"_{obj}_" in str(old[obj + "__motion"]):       # keep the motion the fixture already used for this object
        motion = str(old[obj + "__motion"]) + ".npz"
    else:
        mine = sorted(m for _, o, m in tasks if o == obj and f"_{obj}_" in m)     # (one entry of the reference's table pairs the wineglass with a piggybank motion)
        if not mine:
            continue
        motion = mine[0]
    d = np.load(os.path.join(data, motion))                     # allow_pickle=False (default)
    T = len(d["robot"])
    fr = np.round(np.linspace(0, T - 1, 12)).astype(int)
    out[obj + "__robot"] = d["robot"][fr]; out[obj + "__object"] = d["object"][fr]; out[obj + "__frames"] = fr; out[obj + "__motion"] = np.array(motion[:-4])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "myodm_grasp_frames.npz"), **out)
print(len(objects), "objects,", len(tasks), "motion ids,", len(out) // 4, "objects with grasp frames")
