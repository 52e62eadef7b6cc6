"""ctypes binding of the CPU oracle (oracle/myo_oracle.c).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """Compile libmyo_oracle.so / libmyo_oracle_f32.so with gcc (seconds)."""
    tgt = os.path.join(_HERE, "libmyo_oracle.so")
    src = os.path.join(_HERE, "myo_oracle.c")
    if force or not os.path.exists(tgt) or os.path.getmtime(tgt) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return tgt


class Oracle:
    def __init__(self, blob: bytes, f32=False, lib_path=None):
        build()
        self.lib = C.CDLL(lib_path or os.path.join(_HERE, "libmyo_oracle_f32.so" if f32 else "libmyo_oracle.so"))
        L = self.lib
        L.myoo_load.restype = C.c_void_p
        L.myoo_load.argtypes = [C.c_char_p, C.c_size_t]
        L.myoo_make_data.restype = C.c_void_p
        L.myoo_make_data.argtypes = [C.c_void_p]
        L.myoo_field.restype = C.c_void_p
        L.myoo_field.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        for f in ("myoo_forward", "myoo_fwd_position", "myoo_reset"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, f).restype = None
        L.myoo_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.myoo_step.restype = C.c_int
        L.myoo_free.argtypes = [C.c_void_p]
        L.myoo_free_data.argtypes = [C.c_void_p]
        L.myoo_get_time.restype = C.c_double
        L.myoo_get_time.argtypes = [C.c_void_p]
        L.myoo_set_time.argtypes = [C.c_void_p, C.c_double]
        for f in ("myoo_nefc", "myoo_ncon", "myoo_solver_iter", "myoo_warning"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.myoo_contact.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.myoo_full_m.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.myoo_energy.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.myoo_set_switch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.myoo_set_hfield.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.myoo_set_geom_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.myoo_lengthrange.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_double] * 6 + [C.POINTER(C.c_double)]
        L.myoo_lengthrange.restype = C.c_int
        L.myoo_step_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
        self.real = np.float32 if L.myoo_sizeof_real() == 4 else np.float64
        self._blob = blob
        self.m = L.myoo_load(blob, len(blob))
        if not self.m:
            raise RuntimeError("oracle: bad model blob")
        self.d = L.myoo_make_data(self.m)

    def __del__(self):
        try:
            self.lib.myoo_free_data(self.d)
            self.lib.myoo_free(self.m)
        except Exception:
            pass

    def field(self, name) -> np.ndarray:
        """Writable numpy view of a Data field."""
        n = C.c_int(0)
        p = self.lib.myoo_field(self.m, self.d, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        ct = C.c_float if self.real == np.float32 else C.c_double
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n.value,))

    def set_state(self, qpos=None, qvel=None, act=None, ctrl=None, warm=None, time=None):
        for nm, v in (("qpos", qpos), ("qvel", qvel), ("act", act), ("ctrl", ctrl), ("qacc_warmstart", warm)):
            if v is not None:
                self.field(nm)[:] = v
        if time is not None:
            self.lib.myoo_set_time(self.d, float(time))

    def switches(self, disable_contact=0, disable_limit=0, disable_ellipsoid=0):
        self.lib.myoo_set_switch(self.m, disable_contact, disable_limit, disable_ellipsoid)

    def set_geom_size(self, geom_id, size):
        """model.geom_size[geom_id] = size (3 floats), bounding radius updated; this Oracle instance's model only."""
        self.lib.myoo_set_geom_size(self.m, int(geom_id), (C.c_double * 3)(*[float(x) for x in size]))

    def set_hfield(self, data):
        """Elevation grid [nrow, ncol] (mjModel.hfield_data) of the colliding height field."""
        a = np.ascontiguousarray(data, np.float32).ravel()
        self.lib.myoo_set_hfield(self.m, self.d, a.ctypes.data)

    def set_mpr_mode(self, mode):
        """0 = support-plane output (default, what the HIP path implements); 1 = libccd's nearest-point-of-the-portal output at MuJoCo's
        ccd_tolerance 1e-6.  Process-wide switch in the loaded library: restore 0 after use."""
        self.lib.myoo_set_mpr_mode(int(mode))

    def reset(self):
        self.lib.myoo_reset(self.m, self.d)

    def forward(self):
        self.lib.myoo_forward(self.m, self.d)

    def fwd_position(self):
        self.lib.myoo_fwd_position(self.m, self.d)

    def step(self, nsub=1):
        return self.lib.myoo_step(self.m, self.d, nsub)

    @property
    def time(self):
        return self.lib.myoo_get_time(self.d)

    @property
    def nefc(self):
        return self.lib.myoo_nefc(self.d)

    @property
    def ncon(self):
        return self.lib.myoo_ncon(self.d)

    @property
    def solver_iter(self):
        return self.lib.myoo_solver_iter(self.d)

    def contacts(self):
        out = []
        buf = (C.c_double * 9)()
        for i in range(self.ncon):
            self.lib.myoo_contact(self.d, i, buf)
            out.append(np.array(buf[:]))
        return out

    def full_m(self, nv):
        out = np.zeros((nv, nv))
        self.lib.myoo_full_m(self.m, self.d, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def energy(self):
        out = (C.c_double * 2)()
        self.lib.myoo_energy(self.m, self.d, out)
        return out[0], out[1]

    def lengthrange(self, actuator, accel=20.0, maxforce=0.0, timeconst=1.0, timestep=0.01, inttotal=10.0, interval=2.0):
        """MuJoCo's compile-time length-range simulation (mj_setLengthRange, default mjLROpt) for one actuator:
        returns (lo, hi, spread_lo, spread_hi).  Resets the data."""
        out = (C.c_double * 4)()
        rc = self.lib.myoo_lengthrange(self.m, self.d, int(actuator), accel, maxforce, timeconst, timestep, inttotal, interval, out)
        if rc:
            raise RuntimeError("length-range simulation went unstable")
        return tuple(out)

    def step_batch(self, qpos, qvel, act, warm, time, ctrl, nsub, nthreads):
        """In-place batched stepping of env-major float64 arrays [B, n] (CPU baseline driver)."""
        B = qpos.shape[0]
        flags = np.zeros(B, np.int32)
        for a in (qpos, qvel, act, warm, time, ctrl):
            assert a.dtype == np.float64 and a.flags.c_contiguous
        self.lib.myoo_step_batch(self.m, B, qpos.ctypes.data, qvel.ctypes.data, act.ctypes.data, warm.ctypes.data,
                                 time.ctypes.data, ctrl.ctypes.data, nsub, nthreads, flags.ctypes.data)
        return flags
