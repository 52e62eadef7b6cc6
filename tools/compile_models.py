"""Compile the reference's MJCF config models into committed MYOB blobs.

Runs in the build container only (reads the model *data* files under /root/reference;
they do not travel to the GPU box).  Usage: python tools/compile_models.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_mjx_amd import model as M  # noqa: E402

REF = os.environ.get("MYO_REFERENCE", "/root/reference")
MODELS = {
    "myohand_pose": "myosuite/envs/myo/assets/hand/myohand_pose.xml",
    "myofinger_v0": "myosuite/simhive/myo_sim/finger/myofinger_v0.xml",
    "myolegs": "myosuite/simhive/myo_sim/leg/myolegs.xml",
    "myoelbow_1dof6muscles": "myosuite/envs/myo/assets/elbow/myoelbow_1dof6muscles.xml",
    "myoelbow_1dof6muscles_1dofexo": "myosuite/envs/myo/assets/elbow/myoelbow_1dof6muscles_1dofexo.xml",
    "motorfinger_v0": "myosuite/simhive/myo_sim/finger/motorfinger_v0.xml",
    "myohand_hold": "myosuite/envs/myo/assets/hand/myohand_hold.xml",
    "myolegs_terrain": ("myosuite/simhive/myo_sim/leg/myolegs.xml",),     # height field raised and colliding (TerrainEnvV0)
    # MyoDM TrackEnv (mjx/myodm_v0.py:306-308): myohand_object.xml with OBJECT_NAME -> airplane; meshes collide as convex hulls
    "myohand_object_airplane": ("myosuite/envs/myo/assets/hand/myohand_object.xml", {"OBJECT_NAME": "airplane"}),
    # a second MyoDM object (MyoHand_cup_drink1.npz is the other motion file of tests/golden/ref_motion.npz): same code, another asset
    "myohand_object_cup": ("myosuite/envs/myo/assets/hand/myohand_object.xml", {"OBJECT_NAME": "cup"}),
}
# more MyoDM objects (round 3): one per shape family of simhive/object_sim -- no new code, only assets (stored gzip-compressed: .myob.gz)
for _obj in ("apple", "cubesmall", "duck", "mug", "hammer", "bowl"):
    MODELS[f"myohand_object_{_obj}"] = ("myosuite/envs/myo/assets/hand/myohand_object.xml", {"OBJECT_NAME": _obj}, "gz")
# ... and the rest of the reference's OBJECTS tuple (envs/myo/myodm/__init__.py:586-637): every object MyoDM registers Fixed / Random / motion ids for
for _obj in ("alarmclock", "banana", "binoculars", "camera", "coffeemug", "cubelarge", "cubemedium", "cylinderlarge", "cylindermedium", "cylindersmall",
             "elephant", "eyeglasses", "flashlight", "flute", "gamecontroller", "hand", "headphones", "knife", "lightbulb", "mouse", "phone", "piggybank",
             "pyramidlarge", "pyramidmedium", "pyramidsmall", "scissors", "spherelarge", "spheremedium", "spheresmall", "stamp", "stanfordbunny", "stapler",
             "teapot", "toothbrush", "toothpaste", "toruslarge", "torusmedium", "torussmall", "train", "watch", "waterbottle", "wineglass"):
    MODELS[f"myohand_object_{_obj}"] = ("myosuite/envs/myo/assets/hand/myohand_object.xml", {"OBJECT_NAME": _obj}, "gz")

if __name__ == "__main__":
    only = sys.argv[1:]
    for stem, rel in MODELS.items():
        if only and stem not in only:
            continue
        if isinstance(rel, tuple) and len(rel) >= 2:
            m = M.from_mjcf(os.path.join(REF, rel[0]), replace=rel[1], convex_meshes=True)
        else:
            m = M.from_mjcf(os.path.join(REF, rel[0]), terrain=True) if isinstance(rel, tuple) else M.from_mjcf(os.path.join(REF, rel))
        m.save(os.path.join(M.ASSET_DIR, stem), compress=isinstance(rel, tuple) and len(rel) == 3)
        print(stem, dict(nq=m.nq, nv=m.nv, nu=m.nu, nbody=m.nbody, ntendon=m.ntendon, bytes=len(m.blob())))
