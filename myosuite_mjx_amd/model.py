"""Compiled-model container: arrays (mjModel naming) + names, (de)serialised as a
MYOB blob (`blob.py`) plus a JSON name side-car.

The MJCF sources live in the reference tree, which does not exist on the GPU box, so the
compiled blobs for the config models are committed under `myosuite_mjx_amd/assets/`
(data, produced by `tools/compile_models.py` from the reference's model files)."""
from __future__ import annotations

import json
import os

import numpy as np

from . import blob as _blob

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


class Model:
    def __init__(self, arrays: dict, names: dict, source: str = ""):
        self.arrays = arrays
        self.names = names
        self.source = source
        self._blob = None

    def __getattr__(self, k):
        arrays = self.__dict__.get("arrays", {})
        if k in arrays:
            return arrays[k]
        raise AttributeError(k)

    # sizes, mjModel style
    @property
    def nq(self): return int(self.arrays["sizes"][0])
    @property
    def nv(self): return int(self.arrays["sizes"][1])
    @property
    def nu(self): return int(self.arrays["sizes"][2])
    @property
    def na(self): return int(self.arrays["sizes"][3])     # activation slots of the state arrays (one per actuator)

    @property
    def n_muscle(self):
        """MuJoCo's na for these models: actuators with an activation state (muscles); stateless motors keep an unused slot."""
        return int((np.asarray(self.arrays["actuator_kind"]) == 0).sum())
    @property
    def nbody(self): return int(self.arrays["sizes"][4])
    @property
    def njnt(self): return int(self.arrays["sizes"][5])
    @property
    def ngeom(self): return int(self.arrays["sizes"][6])
    @property
    def nsite(self): return int(self.arrays["sizes"][7])
    @property
    def ntendon(self): return int(self.arrays["sizes"][8])
    @property
    def timestep(self): return float(self.arrays["opt"][0])

    def name2id(self, kind, name):
        return self.names[kind].index(name)

    # mujoco_py-style accessors the reference patches onto MjModel (physics/mj_sim_scene.py:110-163)
    def _n2i(self, kind, name):
        if kind not in self.names or name not in self.names[kind]:
            raise ValueError('No {} with name "{}" exists.'.format(kind, name))
        return self.names[kind].index(name)

    def body_name2id(self, name): return self._n2i("body", name)
    def geom_name2id(self, name): return self._n2i("geom", name)
    def site_name2id(self, name): return self._n2i("site", name)
    def joint_name2id(self, name): return self._n2i("joint", name)
    def actuator_name2id(self, name): return self._n2i("actuator", name)
    def tendon_name2id(self, name): return self._n2i("tendon", name)
    def camera_name2id(self, name): return self._n2i("camera", name)      # (cameras / sensors are not compiled: always "No camera ...")
    def sensor_name2id(self, name): return self._n2i("sensor", name)

    def blob(self) -> bytes:
        if self._blob is None:
            self._blob = _blob.pack(self.arrays)
        return self._blob

    def with_sarcopenia(self) -> "Model":
        """Copy of the model with the muscle condition "sarcopenia" applied (BaseV0.initializeConditions,
        envs/myo/base_v0.py:64-68): the peak force entry of every actuator's gainprm is halved (biasprm is left alone)."""
        arrays = {k: np.array(v, copy=True) for k, v in self.arrays.items()}
        gp = arrays["actuator_gainprm"].reshape(self.nu, -1)
        gp[:, 2] *= 0.5
        if "hip_act" in arrays:
            act = arrays["hip_act"].reshape(self.nu, -1)
            pos = gp[:, 2] >= 0          # a negative entry means "scale / acc0", which the halving does not change
            act[pos, 2] = gp[pos, 2]
        return Model(arrays, self.names, self.source)

    def with_integrator(self, name) -> "Model":
        """Copy of the model with `<option integrator=...>` set: "Euler" (semi-implicit, implicit joint damping) or "RK4" (mj_RungeKutta)."""
        if name not in ("Euler", "RK4"):
            raise ValueError("integrator must be 'Euler' or 'RK4'")
        arrays = {k: np.array(v, copy=True) for k, v in self.arrays.items()}
        arrays["integrator"] = np.array([1 if name == "RK4" else 0], np.int32)
        return Model(arrays, self.names, self.source)

    def save(self, stem, compress=False):
        """MYOB blob + JSON name side-car.  compress=True writes `<stem>.myob.gz` (the hull vertex graphs of the MyoDM objects shrink ~6x)."""
        if compress:
            import gzip
            with open(stem + ".myob.gz", "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", compresslevel=9, mtime=0) as f:
                f.write(self.blob())
        else:
            with open(stem + ".myob", "wb") as f:
                f.write(self.blob())
        with open(stem + ".json", "w") as f:
            json.dump({"names": self.names, "source": os.path.basename(self.source)}, f)

    @classmethod
    def load(cls, stem):
        if os.path.exists(stem + ".myob"):
            with open(stem + ".myob", "rb") as f:
                b = f.read()
        else:
            import gzip
            with gzip.open(stem + ".myob.gz", "rb") as f:
                b = f.read()
        with open(stem + ".json") as f:
            meta = json.load(f)
        m = cls(_blob.unpack(b), meta["names"], meta.get("source", ""))
        m._blob = b
        return m


def from_mjcf(path, terrain=False, replace=None, convex_meshes=False) -> Model:
    """Compile an MJCF file (needs the reference tree; not available on the GPU box)."""
    from .mjcf import compile_mjcf
    from .setconst import set_constants
    from .lowering import lower
    cm = compile_mjcf(path, terrain, replace, convex_meshes)
    set_constants(cm)
    try:
        lower(cm)
    except NotImplementedError as e:     # model is oracle-only for now: myo_model_load will refuse it (no hip_* tables)
        cm.arrays["hip_unsupported"] = np.frombuffer(str(e).encode()[:200].ljust(4, b" "), dtype=np.uint8).astype(np.int32)
    return Model(cm.arrays, cm.names, path)


_ASSETS = {"myohand_pose": "myohand_pose", "myofinger_v0": "myofinger_v0", "myolegs": "myolegs"}


def load_asset(name) -> Model:
    """Load a committed compiled model by stem (e.g. 'myohand_pose')."""
    stem = os.path.join(ASSET_DIR, name)
    if not (os.path.exists(stem + ".myob") or os.path.exists(stem + ".myob.gz")):
        raise FileNotFoundError(f"compiled model {name!r} not found under {ASSET_DIR}; run tools/compile_models.py")
    return Model.load(stem)
