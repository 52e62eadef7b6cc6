"""Extract the tensors of the reference's `mjx_brax_policy` artefact WITHOUT unpickling it.  Build container only.

The file is a pickle of jax / brax objects (brax.io.model.save_params).  Nothing in it is executed or imported here: the opcode stream is
read with `pickletools.genops` and interpreted by a symbolic stack machine that never resolves a global and never calls anything --
GLOBAL / REDUCE / NEWOBJ / BUILD only build inert marker tuples.  ndarrays are then recognised structurally
(`numpy._core.multiarray._reconstruct` + state (1, shape, dtype('f4'), fortran, raw bytes)) and decoded with numpy.frombuffer.

    python tools/extract_brax_policy.py [/root/reference/mjx_brax_policy] [tests/golden/mjx_brax_policy.npz]
"""
import os
import pickletools
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class G(tuple):      # ("global", module, name)
    pass


class Obj:           # inert stand-in for anything a REDUCE / NEWOBJ would have created
    def __init__(self, fn, args):
        self.fn, self.args, self.state = fn, args, None


def symbolic_load(data):
    stack, memo, marks = [], {}, []
    for op, arg, pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            pass
        elif n in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8", "SHORT_BINBYTES", "BINBYTES", "BINBYTES8", "BININT", "BININT1", "BININT2", "BINFLOAT", "LONG1"):
            stack.append(arg)
        elif n == "NONE": stack.append(None)
        elif n == "NEWTRUE": stack.append(True)
        elif n == "NEWFALSE": stack.append(False)
        elif n == "MEMOIZE": memo[len(memo)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET"): stack.append(memo[arg])
        elif n in ("BINPUT", "LONG_BINPUT"): memo[arg] = stack[-1]
        elif n == "MARK": marks.append(len(stack))
        elif n == "EMPTY_TUPLE": stack.append(())
        elif n == "EMPTY_DICT": stack.append({})
        elif n == "EMPTY_LIST": stack.append([])
        elif n == "TUPLE1": stack[-1:] = [(stack[-1],)]
        elif n == "TUPLE2": stack[-2:] = [tuple(stack[-2:])]
        elif n == "TUPLE3": stack[-3:] = [tuple(stack[-3:])]
        elif n == "TUPLE":
            k = marks.pop(); stack[k:] = [tuple(stack[k:])]
        elif n == "SETITEM":
            v = stack.pop(); key = stack.pop(); stack[-1][key] = v
        elif n == "SETITEMS":
            k = marks.pop(); items = stack[k:]; del stack[k:]
            for i in range(0, len(items), 2):
                stack[-1][items[i]] = items[i + 1]
        elif n == "APPENDS":
            k = marks.pop(); items = stack[k:]; del stack[k:]; stack[-1].extend(items)
        elif n == "APPEND":
            v = stack.pop(); stack[-1].append(v)
        elif n == "STACK_GLOBAL":
            name = stack.pop(); mod = stack.pop(); stack.append(G(("global", mod, name)))      # NOT resolved
        elif n == "GLOBAL":
            mod, name = arg.split(" "); stack.append(G(("global", mod, name)))
        elif n == "REDUCE":
            args = stack.pop(); fn = stack.pop(); stack.append(Obj(fn, args))                     # NOT called
        elif n == "NEWOBJ":
            args = stack.pop(); cls = stack.pop(); stack.append(Obj(cls, args))
        elif n == "BUILD":
            st = stack.pop(); stack[-1].state = st if isinstance(stack[-1], Obj) else None
        elif n == "STOP":
            return stack[-1]
        else:
            raise ValueError(f"opcode {n} at {pos}: not needed for a brax parameter file, refusing to guess")
    raise ValueError("no STOP")


def as_array(x):
    """Decode an inert ndarray marker (or a jax array wrapping one); None if x is not one."""
    if isinstance(x, Obj) and isinstance(x.fn, G):
        _, mod, name = x.fn
        if name == "_reconstruct" and mod.startswith("numpy") and x.state is not None:
            ver, shape, dt, fortran, raw = x.state
            assert isinstance(dt, Obj) and dt.fn[2] == "dtype" and not fortran
            code = dt.args[0]
            order = dt.state[1] if dt.state else "<"
            return np.frombuffer(raw, dtype=np.dtype(code).newbyteorder(order if order in "<>" else "=")).reshape(shape).copy()
        if name == "_reconstruct_array":           # jax._src.array: (fun, args, arr_state, aval_state)
            inner = Obj(x.args[0], x.args[1]); inner.state = x.args[2]
            return as_array(inner)
    return None


def to_plain(x):
    a = as_array(x)
    if a is not None:
        return a
    if isinstance(x, dict): return {k: to_plain(v) for k, v in x.items()}
    if isinstance(x, (tuple, list)): return [to_plain(v) for v in x]
    if isinstance(x, Obj): return {"__class__": ".".join(x.fn[1:]) if isinstance(x.fn, G) else "?", **({k: to_plain(v) for k, v in x.state.items()} if isinstance(x.state, dict) else {})}
    return x


def main(src, dst):
    tree = to_plain(symbolic_load(open(src, "rb").read()))
    norm, policy = tree[0], tree[1]
    assert norm["__class__"].endswith("RunningStatisticsState")
    out = {"obs_mean": np.asarray(norm["mean"], np.float32), "obs_std": np.asarray(norm["std"], np.float32), "obs_count": np.asarray(norm["count"])}
    layers = policy["params"]
    for i, name in enumerate(sorted(layers, key=lambda n: int(n.split("_")[-1]))):
        out[f"w{i}"] = np.asarray(layers[name]["kernel"], np.float32)
        out[f"b{i}"] = np.asarray(layers[name]["bias"], np.float32)
    np.savez(dst, **out)
    print({k: v.shape for k, v in out.items()})
    print("obs_mean", out["obs_mean"], "obs_std", out["obs_std"], "count", out["obs_count"])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/mjx_brax_policy",
         sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "mjx_brax_policy.npz"))
