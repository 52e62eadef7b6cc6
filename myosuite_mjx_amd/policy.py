"""Batched policy inference on the device: the network family of the reference's `mjx_brax_policy`
artefact -- brax PPO: running-statistics observation normalisation, swish MLP, tanh-normal head -- consuming the env's
observation buffer and writing the action buffer `myo_step` reads, so a rollout never leaves the GPU.

The reference's artefact is a pickle of jax/brax objects; it is never unpickled here (untrusted, and jax/brax are absent).  Its tensors
are read either by `tools/extract_brax_policy.py` -- a symbolic walk of the pickle's opcodes that imports and executes nothing; the result
for the reference's own artefact is committed as tests/golden/mjx_brax_policy.npz -- or, where brax is installed, by
`tools/export_brax_policy.py`.  Both write the plain `.npz` this module loads:
    obs_mean[obs_dim], obs_std[obs_dim], w0[obs_dim,h], b0[h], ..., w{L-1}[h, 2*act_dim], b{L-1}[2*act_dim]   (float32)
The shapes of `mjx_brax_policy` itself (read with pickletools): obs 2 -> 32 -> 32 -> 32 -> 32 -> 12 (6 actions).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class BraxPolicy:
    def __init__(self, obs_mean, obs_std, kernels, biases, device=0):
        L = capi.lib()
        L.myo_policy_load.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.myo_policy_free.argtypes = [C.c_void_p]
        L.myo_policy_act.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        self.kernels = [_f32(k) for k in kernels]
        self.biases = [_f32(b) for b in biases]
        self.obs_mean, self.obs_std = _f32(obs_mean), _f32(obs_std)
        self.obs_dim = int(self.kernels[0].shape[0])
        if self.kernels[-1].shape[1] % 2:
            raise ValueError("last layer must output (loc, scale) pairs")
        self.act_dim = int(self.kernels[-1].shape[1] // 2)
        n = len(self.kernels)
        widths = (C.c_int * n)(*[int(k.shape[1]) for k in self.kernels])
        kp = (C.c_void_p * n)(*[k.ctypes.data for k in self.kernels])
        bp = (C.c_void_p * n)(*[b.ctypes.data for b in self.biases])
        self.h = C.c_void_p()
        capi._chk(L.myo_policy_load(device, self.obs_dim, self.act_dim, n, widths, self.obs_mean.ctypes.data, self.obs_std.ctypes.data,
                                    kp, bp, C.byref(self.h)))

    @classmethod
    def from_npz(cls, path, device=0):
        z = np.load(path, allow_pickle=False)
        n = sum(1 for k in z.files if k.startswith("w"))
        return cls(z["obs_mean"], z["obs_std"], [z[f"w{i}"] for i in range(n)], [z[f"b{i}"] for i in range(n)], device)

    def act(self, obs_ptr, B, action_ptr, deterministic=True, seed=0, step=0, env_offset=0, stream=None):
        """obs_ptr / action_ptr: device pointers (int) of [B, obs_dim] / [B, act_dim] float32 buffers."""
        capi._chk(capi.lib().myo_policy_act(self.h, obs_ptr, int(B), action_ptr, int(bool(deterministic)), seed, step, env_offset, stream))

    def __del__(self):
        try:
            if self.h:
                capi.lib().myo_policy_free(self.h)
                self.h = None
        except Exception:
            pass


def reference_forward(obs, obs_mean, obs_std, kernels, biases):
    """numpy statement of the same network (deterministic action), used by tests as the float64 checker."""
    x = (np.asarray(obs, np.float64) - obs_mean) / obs_std
    for i, (w, b) in enumerate(zip(kernels, biases)):
        x = x @ np.asarray(w, np.float64) + np.asarray(b, np.float64)
        if i + 1 < len(kernels):
            x = x / (1.0 + np.exp(-x))
    loc, raw = np.split(x, 2, axis=-1)
    return np.tanh(loc), loc, np.logaddexp(0.0, raw) + 0.001
