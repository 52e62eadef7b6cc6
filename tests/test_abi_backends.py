"""SURVEY.md 8b, last sentence: "the identical ABI is implemented by the CPU oracle (device = -1) so Python tests are backend-agnostic".
oracle/myo_oracle_abi.c exports every entry point of include/myo_hip.h on top of the float64 oracle (the physics surface implemented, the rest
answering MYO_E_UNSUPPORTED); tests/abi_backend.py drives one scenario through the ABI on whichever library it is given."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
import abi_backend as AB


def _declared():
    src = open(os.path.join(ROOT, "include", "myo_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|void|const char\*)\s+(myo_[a-z_0-9]+)\(", src, re.M)))


def test_oracle_twin_exports_the_whole_abi():
    from oracle.oracle import build
    build()
    assert os.path.exists(AB.ORACLE_LIB)
    sym = subprocess.run(["nm", "-D", "--defined-only", AB.ORACLE_LIB], capture_output=True, text=True, check=True).stdout
    have = set(re.findall(r" T (myo_[a-z_0-9]+)", sym))
    want = _declared()
    assert len(want) >= 39 and not [s for s in want if s not in have]


def test_abi_scenario_on_the_oracle_backend(hand, oracle64):
    """The scenario through the ABI gives what the oracle's own interface gives (same code underneath; float32 at the boundary), a NaN state is
    flagged and reset like the HIP kernels do it, and an unimplemented entry point answers with the error code and a message instead of aborting."""
    from myosuite_mjx_amd import capi
    out, (q, v, a) = AB.scenario(AB.Backend(AB.ORACLE_LIB, -1), hand)
    assert (out["flags"] == 0).all() and np.allclose(out["time"], 10 * hand.timestep, atol=1e-6)
    for e in range(len(q)):
        oracle64.reset(); oracle64.set_state(qpos=q[e], qvel=v[e], act=a[e], ctrl=a[e])
        assert oracle64.step(10) == 0
        assert np.abs(out["qpos"][e] - oracle64.field("qpos")).max() < 1e-6 and np.abs(out["qvel"][e] - oracle64.field("qvel")).max() < 1e-4
        assert out["diag"][e, 0] == oracle64.nefc and out["diag"][e, 1] == oracle64.ncon
    assert out["flags_after_nan"][2] & capi_flag("BAD_STATE") and (np.delete(out["flags_after_nan"], 2) == 0).all()
    assert np.isfinite(out["qpos_after_nan"]).all() and np.allclose(out["qpos_after_nan"][2], hand.qpos0, atol=1e-6)
    assert out["walk_rc"] == -4 and "not implemented" in out["walk_err"]


def capi_flag(name):
    return {"BAD_STATE": 1, "BAD_QACC": 2, "CONTACT_OVERFLOW": 4}[name]


@pytest.mark.gpu
def test_abi_scenario_on_both_backends(hand):
    """The same calls on libmyo_hip.so (device 0) and on the oracle's twin (device -1): same dims, same state after one env step up to the
    float32 parity bounds of tests/test_gpu_parity.py, same constraint / contact counts, same fault behaviour."""
    g, _ = AB.scenario(AB.Backend(AB.HIP_LIB, 0), hand)
    o, _ = AB.scenario(AB.Backend(AB.ORACLE_LIB, -1), hand)
    assert (g["flags"] == 0).all() and (o["flags"] == 0).all()
    assert np.abs(g["qpos"] - o["qpos"]).max() < 5e-5 and np.abs(g["qvel"] - o["qvel"]).max() < 2e-2 and np.abs(g["act"] - o["act"]).max() < 1e-6
    assert np.abs(g["tenlen"] - o["tenlen"]).max() < 2e-5 and np.allclose(g["time"], o["time"], atol=1e-6)
    assert (g["diag"][:, :2] == o["diag"][:, :2]).all()
    assert (g["flags_after_nan"] == o["flags_after_nan"]).all() and g["flags_after_nan"][2] == 1
    assert np.abs(g["qpos_after_nan"][2] - o["qpos_after_nan"][2]).max() < 1e-6
    ok = np.arange(len(g["qpos"])) != 2
    assert np.abs(g["qpos_after_nan"][ok] - o["qpos_after_nan"][ok]).max() < 1e-4
