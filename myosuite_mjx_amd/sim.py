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


class HipSimScene:
    """Batched counterpart of the reference's sim-backend ABC (`physics/sim_scene.py:38-209`, `DMSimScene` in
    `physics/mj_sim_scene.py:28-65`): same method names and meaning, with `[B, n]` state instead of one env's vectors.

        sim = HipSimScene("myohand_pose", num_envs=4096)
        sim.data.ctrl[:] = ...; sim.advance(substeps=10)       # Robot.step (robot/robot.py:880-882)
        sim.get_state() / sim.set_state(time, qpos, qvel, act)  # sim_scene.py:145-166

    `data.*` are host mirrors: `advance` uploads `data.ctrl` (and the state, if `set_state` or the user touched it via
    `mark_dirty()`), launches the fused step kernel through the C ABI and downloads the new state.  Envs whose state went
    non-finite are reset in place and reported in `last_flags`, like `DMSimScene.advance` swallowing the physics exception and
    resetting (`mj_sim_scene.py:54-61`).  Raises without a GPU (no CPU fallback).
    """

    def __init__(self, model_handle, num_envs=1, device=0):
        self.num_envs = int(num_envs)
        self.sim = self._load_simulation(model_handle, device)
        self.model = self.sim.mj
        self.data = _HostData(self.num_envs, self.model)
        self.init_qpos = np.asarray(self.model.qpos0, np.float64).copy()
        self.init_qvel = np.zeros(self.model.nv)
        self.last_flags = np.zeros(self.num_envs, np.int32)
        self._dirty = True

    def _load_simulation(self, model_handle, device=0):
        m = put_model(model_handle, device)
        self._batch = make_data(m, self.num_envs)
        return m

    @property
    def step_duration(self):
        return float(self.model.timestep)

    def mark_dirty(self):
        """Call after editing data.qpos / qvel / act in place (the reference's direct `data.qpos[:] = ...` writes)."""
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

    def advance(self, substeps: int = 1, render: bool = False):
        if self._dirty:
            self._push_state()
        self._batch.write(capi.F_CTRL, self.data.ctrl)
        self._batch.step(None, capi.ACTMAP_NONE, int(substeps))
        self.last_flags = self._batch.status()
        self._pull_state()

    def forward(self):
        """Derived quantities are recomputed inside every substep; state-only observations need no extra mj_forward."""
        if self._dirty:
            self._push_state()

    def reset(self):
        """mj_resetData for every env: qpos0, zero velocity / activation / time / warm start."""
        self.data.qpos[:] = np.asarray(self.model.qpos0, np.float32)
        self.data.qvel[:] = 0; self.data.act[:] = 0; self.data.ctrl[:] = 0; self.data.time[:] = 0
        self._batch.write(capi.F_WARMSTART, np.zeros((self.num_envs, self.model.nv), np.float32))
        self._push_state()

    def get_state(self):
        return dict(time=self.data.time.copy(), qpos=self.data.qpos.copy(), qvel=self.data.qvel.copy(), act=self.data.act.copy())

    def set_state(self, time=None, qpos=None, qvel=None, act=None):
        for name, v in (("time", time), ("qpos", qpos), ("qvel", qvel), ("act", act)):
            if v is not None:
                getattr(self.data, name)[:] = np.asarray(v, np.float32).reshape(getattr(self.data, name).shape)
        self._push_state()

    def close(self):
        self._batch = None
