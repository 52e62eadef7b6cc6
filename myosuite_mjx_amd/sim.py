"""MJX-flavour functional surface over the C ABI (reference: myosuite/mjx/play.py:8-11,41-49).

    mjx.put_model(mj_model)      -> put_model(name_or_Model)   device-resident model
    mjx.put_data(m, d) (vmapped) -> make_data(model, B)         B environments' state on the device
    jit(mjx.step)(m, d)          -> step(model, data, ctrl, nsubsteps)   in place, asynchronous
    mjx.get_data(m, dx)          -> get(data, "qpos")           host numpy copy
"""
from __future__ import annotations

import numpy as np

from . import capi
from . import model as _model

_FIELDS = {"qpos": capi.F_QPOS, "qvel": capi.F_QVEL, "act": capi.F_ACT, "ctrl": capi.F_CTRL, "time": capi.F_TIME,
           "qacc_warmstart": capi.F_WARMSTART, "qacc": capi.F_QACC, "actuator_length": capi.F_TENLEN,
           "actuator_force": capi.F_ACTFORCE, "flags": capi.F_FLAGS, "diag": capi.F_DIAG}


class DeviceModel:
    def __init__(self, mjmodel, device):
        self.mj = mjmodel
        self.hip = capi.HipModel(mjmodel.blob(), device)

    def __getattr__(self, k):
        return getattr(self.mj, k)


def put_model(model, device=0) -> DeviceModel:
    """Upload a compiled model (asset name, or a `model.Model`) to the GPU.  Raises without a GPU."""
    if isinstance(model, str):
        model = _model.load_asset(model)
    return DeviceModel(model, device)


def make_data(m: DeviceModel, batch: int) -> capi.HipBatch:
    return capi.HipBatch(m.hip, batch)


def step(m: DeviceModel, d: capi.HipBatch, ctrl=None, nsubsteps=1, stream=None):
    """Advance every env by `nsubsteps` physics substeps with controls ctrl[B, nu] (host array) or the stored ones."""
    if ctrl is not None:
        d.write(capi.F_CTRL, np.asarray(ctrl, np.float32))
    d.step(None, capi.ACTMAP_NONE, nsubsteps, stream)
    return d


def get(d: capi.HipBatch, field: str) -> np.ndarray:
    return d.read(_FIELDS[field])


def set_(d: capi.HipBatch, field: str, value):
    d.write(_FIELDS[field], value)


class _HostData:
    """Host mirrors of the per-env state, [B, n] float32 (the reference's `sim.data.*` NumPy views, batched)."""

    def __init__(self, B, m):
        self.qpos = np.tile(np.asarray(m.qpos0, np.float32), (B, 1))
        self.qvel = np.zeros((B, m.nv), np.float32)
        self.act = np.zeros((B, m.na), np.float32)
        self.ctrl = np.zeros((B, m.nu), np.float32)
        self.time = np.zeros((B, 1), np.float32)
        self.qacc = np.zeros((B, m.nv), np.float32)


def _dev_tensor_type():
    """torch.Tensor subclass of the device views: `view[idx] = <numpy array / list>` uploads the value, so the reference's
    `sim.data.ctrl[:] = ctrl` (Robot.step, robot/robot.py:880-882, NumPy on its side) works unchanged on device-resident data."""
    import torch

    class _DevTensor(torch.Tensor):
        def __setitem__(self, idx, val):
            if not torch.is_tensor(val) and not np.isscalar(val):
                val = torch.as_tensor(np.asarray(val), device=self.device).to(self.dtype)
            super().__setitem__(idx, val)
    return _DevTensor


class _DeviceData:
    """`sim.data.*` as zero-copy torch views of the library's device buffers ([B, n] float32): reading / writing them IS reading / writing
    the simulation state, like the reference's NumPy views of MjData -- no upload or download around `advance`.  NumPy values may be
    assigned into them (they are uploaded); reading gives torch tensors on the device."""

    def __init__(self, batch, B, device):
        import torch
        from .envs import _DevArray
        self._keep = batch
        T = _dev_tensor_type()

        def view(field):
            ptr, pitch, width = batch.field_ptr(field)
            ts = "<i4" if field in capi.INT_FIELDS else "<f4"
            return torch.as_tensor(_DevArray(ptr, (B, width), ts, batch), device=f"cuda:{device}").as_subclass(T)
        self.qpos, self.qvel, self.act, self.ctrl = view(capi.F_QPOS), view(capi.F_QVEL), view(capi.F_ACT), view(capi.F_CTRL)
        self.time, self.qacc, self.qacc_warmstart = view(capi.F_TIME), view(capi.F_QACC), view(capi.F_WARMSTART)
        self.actuator_length, self.actuator_force = view(capi.F_TENLEN), view(capi.F_ACTFORCE)


class _NullRenderer:
    """The reference's SimScene owns a Renderer (sim_scene.py:92, :107-119); the batched GPU backend has nothing to draw."""

    def close(self): pass
    def refresh_window(self): pass
    def render_to_window(self): pass


class HipSimScene:
    """Batched counterpart of the reference's sim-backend ABC (`physics/sim_scene.py:38-209`, `DMSimScene` in
    `physics/mj_sim_scene.py:28-65`): same method names and meaning, with `[B, n]` state instead of one env's vectors.

        sim = HipSimScene("myohand_pose", num_envs=4096)
        sim.data.ctrl[:] = ...; sim.advance(substeps=10)       # Robot.step (robot/robot.py:880-882)
        sim.get_state() / sim.set_state(time, qpos, qvel, act)  # sim_scene.py:145-166

    `data.*` are device-resident by default (`as_torch=True`): zero-copy torch views of the library's buffers, so `advance` is just the
    kernel launch.  With `as_torch=False` they are host mirrors: `advance` uploads `data.ctrl` (and the state, if
    `set_state` or `mark_dirty()` touched it), steps and downloads the new state.  Envs whose state went non-finite are reset in place
    and reported in `last_flags`, like `DMSimScene.advance` swallowing the physics exception and resetting (`mj_sim_scene.py:54-61`).
    Raises without a GPU (no CPU fallback).  The rest of the ABC -- copy_model, save_binary, upload_height_field, get_mjlib, get_handle,
    disable_option(_context), model.*_name2id -- is provided with its batched / MYOB meaning, see each method.
    """

    def __init__(self, model_handle, num_envs=1, device=0, as_torch=True):
        self.num_envs = int(num_envs)
        self.device = device
        self.as_torch = as_torch
        self.sim = self._load_simulation(model_handle, device)
        self.model = self.sim.mj
        self.data = _DeviceData(self._batch, self.num_envs, device) if as_torch else _HostData(self.num_envs, self.model)
        self.lib = self.get_mjlib()
        self.renderer = self._create_renderer(self.sim)
        self.init_qpos = np.asarray(self.model.qpos0, np.float64).copy()
        self.init_qvel = np.zeros(self.model.nv)
        self._last_flags = np.zeros(self.num_envs, np.int32)
        self._flags_stale = False
        self._dirty = not as_torch
        self._disabled = [0, 0, 0]

    def _load_simulation(self, model_handle, device=0):
        m = put_model(model_handle, device)
        self._batch = make_data(m, self.num_envs)
        return m

    def _create_renderer(self, sim):
        return _NullRenderer()

    @property
    def step_duration(self):
        return float(self.model.timestep)

    # ---- the abstract methods of physics/sim_scene.py:168-209 -------------------------------------------------------------------------
    def copy_model(self):
        """A deep copy of the compiled model (the MjModel of this backend): arrays + names."""
        return _model.Model({k: np.array(v, copy=True) for k, v in self.model.arrays.items()}, {k: list(v) for k, v in self.model.names.items()}, self.model.source)

    def save_binary(self, path: str) -> str:
        """The backend's binary model format is the MYOB blob (+ a JSON name side-car), the counterpart of MuJoCo's .mjb."""
        stem = path[:-5] if path.endswith(".myob") else (path[:-4] if path.endswith(".mjb") else path)
        self.model.save(stem)
        return stem + ".myob"

    def upload_height_field(self, hfield_id: int = 0, data=None):
        """TerrainEnvV0 edits model.hfield_data and calls this to refresh the rendering context (walk_v0.py:624-630); here the elevation grid
        of every env lives on the device (MYO_F_HFIELD) and `data` [B, nrow * ncol] (or one grid for all) is uploaded into it."""
        if data is None:
            return
        n = int(self.model.hfield_dims[0]) * int(self.model.hfield_dims[1])
        a = np.broadcast_to(np.asarray(data, np.float32).reshape(-1, n), (self.num_envs, n))
        self._batch.write(capi.F_HFIELD, np.ascontiguousarray(a))

    def get_mjlib(self):
        """The low-level API of this backend: the ctypes handle of libmyo_hip.so (include/myo_hip.h)."""
        return capi.lib()

    def get_handle(self, value):
        """Native handle (myo_model* / myo_batch*) of a wrapper object, for direct calls into get_mjlib()."""
        return getattr(value, "h", value)

    def disable_option(self, constraint_solver=False, limits=False, contact=False, gravity=False, clamp_ctrl=False, actuation=False):
        """sim_scene.py:121-143.  The kernels implement the contact and limit switches (model-wide); the others have no counterpart."""
        if constraint_solver or gravity or clamp_ctrl or actuation:
            raise NotImplementedError("HipSimScene.disable_option: only `limits` and `contact` are switchable")
        self._disabled = [int(contact or self._disabled[0]), int(limits or self._disabled[1]), self._disabled[2]]
        self.sim.hip.set_switch(*self._disabled)

    def disable_option_context(self, **kwargs):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            saved = list(self._disabled)
            self.disable_option(**kwargs)
            try:
                yield
            finally:
                self._disabled = saved
                self.sim.hip.set_switch(*saved)
        return ctx()

    # ---- stepping ---------------------------------------------------------------------------------------------------------------------
    def mark_dirty(self):
        """Host-mirror mode only: call after editing data.qpos / qvel / act in place."""
        self._dirty = True

    def _push_state(self):
        b = self._batch
        b.write(capi.F_QPOS, self.data.qpos); b.write(capi.F_QVEL, self.data.qvel)
        b.write(capi.F_ACT, self.data.act); b.write(capi.F_TIME, self.data.time)
        self._dirty = False

    def _pull_state(self):
        b = self._batch
        self.data.qpos[:] = b.read(capi.F_QPOS); self.data.qvel[:] = b.read(capi.F_QVEL)
        self.data.act[:] = b.read(capi.F_ACT); self.data.time[:] = b.read(capi.F_TIME)
        self.data.qacc[:] = b.read(capi.F_QACC)

    def _stream(self):
        if self.as_torch:
            import torch
            return torch.cuda.current_stream(self.device).cuda_stream
        return None

    def advance(self, substeps: int = 1, render: bool = False):
        if self.as_torch:                      # device-resident: the controls and the state are already where the kernel reads them
            self._batch.step(None, capi.ACTMAP_NONE, int(substeps), self._stream())
            self._flags_stale = True           # (in-kernel resets of bad states are reported when `last_flags` is next read: one device sync there)
            return
        if self._dirty:
            self._push_state()
        self._batch.write(capi.F_CTRL, self.data.ctrl)
        self._batch.step(None, capi.ACTMAP_NONE, int(substeps))
        self._last_flags = self._batch.status()
        self._pull_state()

    @property
    def last_flags(self):
        """Per-env fault flags of the steps since they were last read (MYO_FLAG_*: envs whose state went bad were reset in place, like
        DMSimScene.advance catching the physics error, mj_sim_scene.py:54-61).  Device-resident mode fetches them here, not inside advance."""
        if getattr(self, "_flags_stale", False):
            self._last_flags = self._last_flags | self._batch.status()
            self._flags_stale = False
        return self._last_flags

    def status(self):
        """Per-env fault flags since the last call, then cleared."""
        out = self.last_flags.copy()
        self._last_flags = np.zeros(self.num_envs, np.int32)
        return out

    def forward(self):
        """Derived quantities are recomputed inside every substep; state-only observations need no extra mj_forward."""
        if not self.as_torch and self._dirty:
            self._push_state()

    def reset(self):
        """mj_resetData for every env: qpos0, zero velocity / activation / time / warm start."""
        if self.as_torch:
            import torch
            self.data.qpos[:] = torch.as_tensor(np.asarray(self.model.qpos0, np.float32), device=self.data.qpos.device)
            for t in (self.data.qvel, self.data.act, self.data.ctrl, self.data.time, self.data.qacc_warmstart):
                t.zero_()
            return
        self.data.qpos[:] = np.asarray(self.model.qpos0, np.float32)
        self.data.qvel[:] = 0; self.data.act[:] = 0; self.data.ctrl[:] = 0; self.data.time[:] = 0
        self._batch.write(capi.F_WARMSTART, np.zeros((self.num_envs, self.model.nv), np.float32))
        self._push_state()

    def get_state(self):
        cp = (lambda t: t.clone()) if self.as_torch else (lambda a: a.copy())
        return dict(time=cp(self.data.time), qpos=cp(self.data.qpos), qvel=cp(self.data.qvel), act=cp(self.data.act))

    def set_state(self, time=None, qpos=None, qvel=None, act=None):
        for name, v in (("time", time), ("qpos", qpos), ("qvel", qvel), ("act", act)):
            if v is None:
                continue
            dst = getattr(self.data, name)
            if self.as_torch:
                import torch
                dst[:] = torch.as_tensor(np.asarray(v, np.float32) if not torch.is_tensor(v) else v, device=dst.device).reshape(dst.shape)
            else:
                dst[:] = np.asarray(v, np.float32).reshape(dst.shape)
        if not self.as_torch:
            self._push_state()

    def close(self):
        self.renderer.close()
        self.data = None
        self._batch = None
