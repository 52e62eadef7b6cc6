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
