"""Run this where jax + brax are installed (NOT in the build container: the artefact is a pickle of jax/brax objects and is never
unpickled there): re-exports a brax PPO policy such as the reference's `mjx_brax_policy` to the plain .npz that
`myosuite_mjx_amd.policy.BraxPolicy.from_npz` loads.

    python tools/export_brax_policy.py /path/to/mjx_brax_policy policy.npz
"""
import sys

import numpy as np


def main(src, dst):
    from brax.io import model                      # third-party; provides load_params
    params = model.load_params(src)                # (RunningStatisticsState, policy params[, value params])
    norm, policy = params[0], params[1]
    out = {"obs_mean": np.asarray(norm.mean, np.float32), "obs_std": np.asarray(norm.std, np.float32)}
    layers = policy["params"]
    for i, name in enumerate(sorted(layers, key=lambda n: int(n.split("_")[-1]))):
        out[f"w{i}"] = np.asarray(layers[name]["kernel"], np.float32)
        out[f"b{i}"] = np.asarray(layers[name]["bias"], np.float32)
    np.savez(dst, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
