"""A second, minimal binding of the C ABI (include/myo_hip.h) for the backend-agnostic tests: the same calls on libmyo_hip.so (device 0) and on the
CPU twin of the ABI that the oracle exports (oracle/libmyo_oracle_abi.so, device -1).  Test infrastructure: the product's own binding
(myosuite_mjx_amd/capi.py) loads the HIP library only."""
import ctypes as C
import os

import numpy as np

from myosuite_mjx_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB = os.path.join(ROOT, "myosuite_mjx_amd", "libmyo_hip.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "libmyo_oracle_abi.so")


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nq", "nv", "nu", "na", "nbody", "ntendon", "nsite", "nlink", "obs_dim", "env_lds_bytes", "lanes_per_env", "ncon_max")] + [("timestep", C.c_float)]


class Backend:
    def __init__(self, lib_path, device):
        if os.path.abspath(lib_path) == os.path.abspath(HIP_LIB):
            capi.lib()          # (lets torch load its bundled HIP runtime first: two HIP runtimes in one process do not share the device, see capi.lib)
        self.L = C.CDLL(lib_path)
        self.device = device
        L = self.L
        L.myo_last_error.restype = C.c_char_p
        L.myo_model_load.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.myo_model_free.argtypes = [C.c_void_p]
        L.myo_model_dims.argtypes = [C.c_void_p, C.POINTER(Dims)]
        L.myo_batch_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.myo_batch_free.argtypes = [C.c_void_p]
        L.myo_batch_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.myo_batch_write.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.myo_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.myo_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.myo_status.argtypes = [C.c_void_p, C.c_void_p]
        L.myo_sync.argtypes = [C.c_void_p]
        L.myo_batch_configure_walk.argtypes = [C.c_void_p, C.c_void_p]

    def err(self):
        return self.L.myo_last_error().decode()

    def load(self, blob):
        h = C.c_void_p()
        rc = self.L.myo_model_load(blob, len(blob), self.device, C.byref(h))
        assert rc == 0, (rc, self.err())
        d = Dims()
        assert self.L.myo_model_dims(h, C.byref(d)) == 0
        return h, d

    def batch(self, h, B):
        b = C.c_void_p()
        rc = self.L.myo_batch_create(h, B, C.byref(b))
        assert rc == 0, (rc, self.err())
        return b

    def write(self, b, field, a):
        a = np.ascontiguousarray(a, np.float32)
        rc = self.L.myo_batch_write(b, field, a.ctypes.data, a.nbytes)
        assert rc == 0, (rc, self.err())

    def read(self, b, field, shape, dtype=np.float32):
        a = np.zeros(shape, dtype)
        rc = self.L.myo_batch_read(b, field, a.ctypes.data, a.nbytes)
        assert rc == 0, (rc, self.err())
        return a

    def step(self, b, nsub):
        rc = self.L.myo_step(b, None, capi.ACTMAP_NONE, nsub, None)
        assert rc == 0, (rc, self.err())
        self.L.myo_sync(None)

    def status(self, b, B):
        f = np.zeros(B, np.int32)
        assert self.L.myo_status(b, f.ctypes.data) == 0
        return f


def scenario(be, model, seed=3, B=6, nsub=10):
    """The same sequence of ABI calls on either backend: load, dims, create, write a seeded state, step one env step, read back; then a NaN
    in one env's qpos: that env is flagged and reset while the others go on (mj_sim_scene.py:54-61); then an entry point the backend may not
    implement."""
    h, d = be.load(model.blob())
    assert (d.nq, d.nv, d.nu) == (model.nq, model.nv, model.nu) and abs(d.timestep - model.timestep) < 1e-9
    b = be.batch(h, B)
    rng = np.random.default_rng(seed)
    lo, hi = model.jnt_range[:, 0], model.jnt_range[:, 1]
    q = (lo + (hi - lo) * rng.uniform(0.2, 0.8, (B, model.nq))).astype(np.float32)
    v = rng.normal(0, 0.3, (B, model.nv)).astype(np.float32)
    a = rng.uniform(0, 1, (B, model.nu)).astype(np.float32)
    for f, x in ((capi.F_QPOS, q), (capi.F_QVEL, v), (capi.F_ACT, a), (capi.F_CTRL, a)):
        be.write(b, f, x)
    be.step(b, nsub)
    out = dict(qpos=be.read(b, capi.F_QPOS, (B, model.nq)), qvel=be.read(b, capi.F_QVEL, (B, model.nv)), act=be.read(b, capi.F_ACT, (B, model.nu)),
               time=be.read(b, capi.F_TIME, (B, 1)), tenlen=be.read(b, capi.F_TENLEN, (B, model.nu)), flags=be.status(b, B),
               diag=be.read(b, capi.F_DIAG, (B, 8), np.int32))
    q2 = out["qpos"].copy(); q2[2, 0] = np.nan
    be.write(b, capi.F_QPOS, q2)
    be.step(b, 1)
    out["flags_after_nan"] = be.status(b, B)
    out["qpos_after_nan"] = be.read(b, capi.F_QPOS, (B, model.nq))
    out["walk_rc"] = be.L.myo_batch_configure_walk(b, None)
    out["walk_err"] = be.err()
    be.L.myo_batch_free(b); be.L.myo_model_free(h)
    return out, (q, v, a)
