"""Register / scratch figures of the built step kernels, read from the code object's metadata (no GPU needed).

Round 2 shipped step kernels with 35-45 VGPRs spilled to scratch memory inside divergent code (8.5x wasted HBM traffic, and a class of
order-dependent wrong results on two intermediate code shapes, DESIGN.md section 4 "Diagnostic builds").  Every instantiation of the wave
kernel that a model can be routed to must now be free of vector-register spills and of scratch memory altogether, and must keep the occupancy
its LDS slice is sized for (4 waves per SIMD = 128 registers for the hand class, 2 = 256 for the 36-dof class)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels():
    from myosuite_mjx_amd import capi
    import kernel_resources
    if not os.path.exists(capi.LIB_PATH):
        capi.build_library()
    out = {}
    for r in kernel_resources.resources(capi.LIB_PATH):
        out[subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]] = r
    return out


# <NVT, KC, NC, NTR, WPE, SCHED, SPEC, HF, TRK, RK4>: what myo_hip.hip launch_step can select
WAVE = {
    "headline: MyoHand, size-specialised": ("24, 8, 32, 1, 4, false, 1, false, false, false", 128),
    "hand / finger class, run-time sizes": ("24, 8, 32, 1, 3, false, 0, false, false, false", 128),   # budget of 3 waves per SIMD, lands on 128 registers = 4
    "36-dof class": ("36, 20, 32, 2, 2, false, 0, false, false, false", 256),
    "MyoLeg, size-specialised": ("36, 20, 32, 2, 2, false, 2, false, false, false", 256),
    "36-dof class, substep scheduler": ("36, 20, 32, 2, 2, true, 0, false, false, false", 256),
    "MyoLeg, scheduler (config 5 default)": ("36, 20, 32, 2, 2, true, 2, false, false, false", 256),
    "terrain": ("36, 20, 32, 2, 2, false, 0, true, false, false", 256),
    "terrain, size-specialised": ("36, 20, 32, 2, 2, false, 3, true, false, false", 256),
    "terrain, scheduler": ("36, 20, 32, 2, 2, true, 3, true, false, false", 256),
    "TrackEnv model class (TRK)": ("36, 20, 32, 2, 2, false, 0, false, true, false", 256),
    "RK4 hand class": ("24, 8, 32, 1, 3, false, 0, false, false, true", 168),
    "RK4 36-dof class": ("36, 20, 32, 2, 2, false, 0, false, false, true", 256),
}


@pytest.mark.parametrize("what", sorted(WAVE))
def test_wave_kernel_has_no_spills_and_no_scratch(kernels, what):
    args, vmax = WAVE[what]
    name = f"void step_kernel_w<{args}>"
    assert name in kernels, (name, sorted(k for k in kernels if "step_kernel_w" in k))
    r = kernels[name]
    assert r["vgpr_spill"] == 0 and r["scratch"] == 0, r
    assert r["vgpr"] + r["agpr"] <= vmax, r        # occupancy the LDS slice was sized for


def test_no_other_wave_instantiation_is_built(kernels):
    """Instantiations nobody can reach (round 2: a scheduled hand kernel with 86 spilled VGPRs, two 5-waves-per-SIMD experiments with 87-90) are gone."""
    built = sorted(k for k in kernels if k.startswith("void step_kernel_w<"))
    assert built == sorted(f"void step_kernel_w<{a}>" for a, _ in WAVE.values()), built
