"""Developer probe (diagnostic build, -DMYO_STAMPS=1): stage split of the TRK kernel on TrackEnv-like states."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MYO_HIP_LIB"] = os.path.join(ROOT, "myosuite_mjx_amd", "libmyo_hip_stamps.so")
from myosuite_mjx_amd import capi, model as M
import test_gpu_track as T
m = M.load_asset("myohand_object_airplane")
hm = capi.HipModel(m.blob(), 0)
B = int(os.environ.get("B", 4096))
qq, vv, aa, cc = T._motion_states(m, [int(t) for t in np.random.default_rng(0).integers(0, 60, B)], 5)
b = capi.HipBatch(hm, B)
for f, x in ((capi.F_QPOS, qq), (capi.F_QVEL, 0 * vv), (capi.F_ACT, aa), (capi.F_CTRL, cc)):
    b.write(f, x)
for _ in range(3): b.step(None, capi.ACTMAP_NONE, 5)
capi.sync()
b.step(None, capi.ACTMAP_NONE, 5)
capi.sync()
st2, ok = capi.read_stamps(b, 2 * B)
st = st2[:B]
NAMES = ["load/check", "kinematics", "tendon+muscle", "dynamics(CRB/RNE)", "narrow phase", "constraint rows", "geom frames + broad phase", "newton", "euler", "store"]
tot = st[:, :10].sum(1)
print("dims", hm.dims.env_lds_bytes if hasattr(hm, "dims") else "", "per-WG total cycles mean", tot.mean(), "max", tot.max(), "diag ncon mean", b.read(capi.F_DIAG)[:, 1].mean(), "ncand", (b.read(capi.F_DIAG)[:, 4] & 0xFFFF).mean() / 5, "f_mpr (sum over rounds of the slowest lane's support evaluations) per substep", b.read(capi.F_DIAG)[:, 5].mean() / 5)
for k, n in enumerate(NAMES):
    print(f"   {n:26s} {st[:,k].mean()/5:12,.0f} cycles/substep  {100*st[:,k].mean()/tot.mean():5.1f}%")
sub = st2[B:]
SUBN = ["newton: forces, cost, gradient, convergence test", "newton: Hessian assembly", "newton+euler: build rows, Cholesky, store L", "newton+euler: triangular solves",
        "newton: M*search, J*search", "newton: line search", "newton: start (M*warm, J*warm)"]
for k, n in enumerate(SUBN):
    print(f"      {n:52s} {sub[:,k].mean()/5:12,.0f} cycles/substep  {100*sub[:,k].mean()/tot.mean():5.1f}%")
d = b.read(capi.F_DIAG)
print("   per substep: newton iterations %.2f, factorisations %.2f, line-search evaluations %.2f, constraint rows (last substep) %.1f" %
      ((d[:, 6] >> 16).mean() / 5, (d[:, 7] >> 16).mean() / 5, (d[:, 7] & 0xFFFF).mean() / 5, d[:, 0].mean()))
