"""MJCF subset compiler: XML -> flat arrays (the "compiled model").

The reference never compiles MJCF itself: it hands the XML to MuJoCo
(`myosuite/physics/mj_sim_scene.py:28-49`, `myosuite/mjx/play.py:8-10`).  This
module restates the part of MuJoCo's model compiler that the MyoSuite hand /
finger / leg models exercise, producing arrays named
after MuJoCo's `mjModel` fields so that the oracle (`oracle/myo_oracle.c`) and
the HIP stepper (`myosuite_mjx_amd/csrc/`) read the same numbers.

Supported: <include>, <compiler> (angle, eulerseq, inertiafromgeom,
balanceinertia, boundmass, boundinertia, meshdir, autolimits), <option>,
nested <default> classes + childclass, bodies / inertial / hinge+slide joints /
geoms (plane, sphere, capsule, ellipsoid, cylinder, box, mesh-for-inertia) /
sites, spatial tendons with sphere+cylinder wrapping, side sites and pulleys,
<muscle> and <general dyntype=muscle> actuators, contact excludes, keyframes.
Anything else raises so that an unsupported model fails loudly.
"""
from __future__ import annotations

import math
import os
import struct
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

mjMINVAL = 1e-15
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
GEOM_TYPES = {"plane": 0, "hfield": 1, "sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6, "mesh": 7}
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = range(4)
JNT_TYPES = {"free": 0, "ball": 1, "slide": 2, "hinge": 3}
WRAP_NONE, WRAP_JOINT, WRAP_PULLEY, WRAP_SITE, WRAP_SPHERE, WRAP_CYLINDER = range(6)


# --------------------------------------------------------------------------- math helpers
def _floats(s, n=None):
    v = np.array([float(x) for x in s.replace(",", " ").split()], dtype=np.float64)
    if n is not None and v.size != n:
        raise ValueError(f"expected {n} numbers, got {s!r}")
    return v


def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat_conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def quat_normalize(q):
    n = np.linalg.norm(q)
    if n < mjMINVAL:
        return np.array([1.0, 0, 0, 0])
    return q / n


def quat2mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def mat2quat(R):
    """Rotation matrix -> unit quaternion (w,x,y,z), positive w preferred."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    return quat_normalize(q)


def axisangle2quat(axis, ang):
    axis = np.asarray(axis, float)
    n = np.linalg.norm(axis)
    if n < mjMINVAL:
        return np.array([1.0, 0, 0, 0])
    axis = axis / n
    return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


def z2quat(vec):
    """Quaternion rotating +z onto vec (MuJoCo's fromto convention)."""
    vec = vec / np.linalg.norm(vec)
    axis = np.cross([0, 0, 1.0], vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        axis = np.array([1.0, 0, 0])
    else:
        axis = axis / s
    ang = math.atan2(s, vec[2])
    return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


def euler2quat(e, seq):
    q = np.array([1.0, 0, 0, 0])
    for i in range(3):
        c = seq[i]
        ax = {"x": [1, 0, 0], "y": [0, 1, 0], "z": [0, 0, 1]}[c.lower()]
        qr = axisangle2quat(ax, e[i])
        q = quat_mul(q, qr) if c.islower() else quat_mul(qr, q)
    return quat_normalize(q)


# --------------------------------------------------------------------------- geometry: volumes & inertias
def _read_stl(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:5] == b"solid" and b"facet" in data[:1000]:
        tris = []
        cur = []
        for line in data.decode("ascii", "ignore").splitlines():
            p = line.split()
            if len(p) == 4 and p[0] == "vertex":
                cur.append([float(p[1]), float(p[2]), float(p[3])])
                if len(cur) == 3:
                    tris.append(cur)
                    cur = []
        return np.array(tris, dtype=np.float64)
    n = struct.unpack_from("<I", data, 80)[0]
    arr = np.frombuffer(data, dtype=np.uint8, count=n * 50, offset=84).reshape(n, 50)
    v = arr[:, 12:48].copy().view("<f4").reshape(n, 3, 3)
    return v.astype(np.float64)


def mesh_mass_properties(tris):
    """Volume, centre of mass and unit-density inertia about the COM of a closed
    triangle mesh by signed tetrahedra (exact for a closed oriented surface)."""
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    vol6 = np.einsum("ij,ij->i", a, np.cross(b, c))
    V = vol6.sum() / 6.0
    sign = 1.0 if V >= 0 else -1.0
    V *= sign
    vol6 = vol6 * sign
    com = ((a + b + c) / 4.0 * (vol6 / 6.0)[:, None]).sum(0) / V
    # second moments: integral over tetra (0,a,b,c) of x x^T = vol/20 * (sum_i v_i v_i^T + (sum v)(sum v)^T)
    s = a + b + c
    C = np.zeros((3, 3))
    for v in (a, b, c):
        C += np.einsum("i,ij,ik->jk", vol6 / 120.0, v, v)
    C += np.einsum("i,ij,ik->jk", vol6 / 120.0, s, s)
    C -= V * np.outer(com, com)
    I = np.trace(C) * np.eye(3) - C
    return V, com, I


def geom_volume_inertia(gtype, size):
    """Volume and unit-density inertia (diag, geom frame, about geom centre) of a primitive."""
    if gtype == GEOM_SPHERE:
        r = size[0]
        V = 4.0 / 3.0 * math.pi * r ** 3
        return V, np.full(3, 0.4 * V * r * r)
    if gtype == GEOM_CAPSULE:
        r, h = size[0], size[1]
        Vc = math.pi * r * r * 2 * h
        Vs = 4.0 / 3.0 * math.pi * r ** 3
        V = Vc + Vs
        # cylinder part + two hemispheres shifted by h (+3r/8 to their own COM)
        Ixx = Vc * (3 * r * r + 4 * h * h) / 12.0 + Vs * (0.4 * r * r + h * h + 0.75 * r * h)
        Izz = Vc * r * r / 2.0 + Vs * 0.4 * r * r
        return V, np.array([Ixx, Ixx, Izz])
    if gtype == GEOM_CYLINDER:
        r, h = size[0], size[1]
        V = math.pi * r * r * 2 * h
        Ixx = V * (3 * r * r + 4 * h * h) / 12.0
        return V, np.array([Ixx, Ixx, V * r * r / 2.0])
    if gtype == GEOM_ELLIPSOID:
        a, b, c = size
        V = 4.0 / 3.0 * math.pi * a * b * c
        return V, V / 5.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == GEOM_BOX:
        a, b, c = size
        V = 8 * a * b * c
        return V, V / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    raise ValueError(f"no volume for geom type {gtype}")


def geom_rbound(gtype, size):
    if gtype == GEOM_SPHERE:
        return size[0]
    if gtype == GEOM_CAPSULE:
        return size[0] + size[1]
    if gtype == GEOM_CYLINDER:
        return math.sqrt(size[0] ** 2 + size[1] ** 2)
    if gtype == GEOM_ELLIPSOID:
        return max(size)
    if gtype == GEOM_BOX:
        return float(np.linalg.norm(size))
    if gtype == GEOM_MESH:
        return float(size[0])      # conservative sphere about the geom frame origin (see _finalize)
    if gtype == GEOM_HFIELD:
        return float(np.linalg.norm(size))
    return 0.0


# --------------------------------------------------------------------------- defaults
_ACT_TAGS = ("general", "muscle", "motor", "position", "velocity")


class Defaults:
    def __init__(self, parent=None):
        self.attr = {} if parent is None else {k: dict(v) for k, v in parent.attr.items()}

    def update(self, tag, attrib):
        key = "actuator" if tag in _ACT_TAGS else tag
        self.attr.setdefault(key, {}).update(attrib)

    def get(self, tag):
        key = "actuator" if tag in _ACT_TAGS else tag
        return self.attr.get(key, {})


@dataclass
class CompiledModel:
    """Flat arrays, MuJoCo `mjModel` naming.  All float arrays are float64."""
    arrays: dict = field(default_factory=dict)
    names: dict = field(default_factory=dict)
    source: str = ""

    def __getattr__(self, k):
        a = self.__dict__.get("arrays", {})
        if k in a:
            return a[k]
        raise AttributeError(k)

    def name2id(self, kind, name):
        return self.names[kind].index(name)


# --------------------------------------------------------------------------- the compiler
class _Compiler:
    def __init__(self, path, terrain=False, replace=None, convex_meshes=False):
        self.path = os.path.abspath(path)
        self.replace = dict(replace or {})   # text substitutions applied to every file read (TrackEnv: OBJECT_NAME -> "airplane", mjx/myodm_v0.py:66-80)
        self.convex_meshes = convex_meshes   # colliding mesh geoms carry the vertices of their convex hull (MuJoCo collides meshes as convex hulls)
        self.terrain = terrain      # True: keep the height field's pairs and raise its geom to z = 0 (TerrainEnvV0.reset, walk_v0.py:624-630)
        self.hfields = {}
        self.comp = dict(angle="degree", eulerseq="xyz", inertiafromgeom="auto", balanceinertia=False,
                         boundmass=0.0, boundinertia=0.0, meshdir="", autolimits=True,
                         settotalmass=-1.0, inertiagrouprange=(0, 5))
        self.comp_dir = os.path.dirname(self.path)
        self.opt = dict(timestep=0.002, gravity=np.array([0, 0, -9.81]), tolerance=1e-8, iterations=100,
                        ls_iterations=50, ls_tolerance=0.01, impratio=1.0, cone="pyramidal",
                        solver="Newton", integrator="Euler", o_margin=0.0)
        self.defaults = {"main": Defaults()}
        self.meshes = {}
        self.bodies = []   # dicts
        self.joints = []
        self.geoms = []
        self.sites = []
        self.tendons = []
        self.wraps = []
        self.actuators = []
        self.excludes = []
        self.pairs = []
        self.equalities = []
        self.keys = []

    # ---- XML loading with <include>
    def _parse(self, path):
        if not self.replace:
            return ET.parse(path).getroot()
        with open(path) as f:
            txt = f.read()
        for k, v in self.replace.items():
            txt = txt.replace(k, v)
        return ET.fromstring(txt)

    def _load(self, path):
        root = self._parse(path)
        self._expand(root, os.path.dirname(path))
        return root

    def _expand(self, elem, base):
        i = 0
        children = list(elem)
        out = []
        for ch in children:
            if ch.tag == "include":
                p = os.path.normpath(os.path.join(base, ch.attrib["file"]))
                sub = self._parse(p)
                self._expand(sub, os.path.dirname(p))
                for s in list(sub):
                    s.set("__dir", s.get("__dir", os.path.dirname(p)))
                    out.append(s)
            else:
                self._expand(ch, base)
                if "__dir" not in ch.attrib:
                    ch.set("__dir", base)
                out.append(ch)
        for ch in children:
            elem.remove(ch)
        for ch in out:
            elem.append(ch)

    def _angle(self, v):
        return np.asarray(v, float) * (math.pi / 180.0) if self.comp["angle"] == "degree" else np.asarray(v, float)

    def _bool(self, s):
        return str(s).lower() == "true"

    # ---- sections
    def _do_compiler(self, e):
        a = e.attrib
        for k in ("angle", "eulerseq", "inertiafromgeom"):
            if k in a:
                self.comp[k] = a[k].lower() if k != "eulerseq" else a[k]
        for k in ("balanceinertia", "autolimits"):
            if k in a:
                self.comp[k] = self._bool(a[k])
        for k in ("boundmass", "boundinertia", "settotalmass"):
            if k in a:
                self.comp[k] = float(a[k])
        if "meshdir" in a:
            self.comp["meshdir"] = a["meshdir"]  # resolved against the main file's directory
        if "inertiagrouprange" in a:
            self.comp["inertiagrouprange"] = tuple(int(x) for x in a["inertiagrouprange"].split())

    def _do_option(self, e):
        a = e.attrib
        for k in ("timestep", "tolerance", "ls_tolerance", "impratio", "o_margin"):
            if k in a:
                self.opt[k] = float(a[k])
        for k in ("iterations", "ls_iterations"):
            if k in a:
                self.opt[k] = int(a[k])
        if "gravity" in a:
            self.opt["gravity"] = _floats(a["gravity"], 3)
        for k in ("cone", "solver", "integrator"):
            if k in a:
                self.opt[k] = a[k]
        for ch in e:
            if ch.tag == "flag":
                for k, v in ch.attrib.items():
                    if k != "__dir":
                        self.opt["flag_" + k] = v

    def _do_default(self, e, parent_name):
        name = e.attrib.get("class", "main")
        if name == "main" and parent_name is None:
            d = self.defaults["main"]
        else:
            d = Defaults(self.defaults[parent_name or "main"])
            self.defaults[name] = d
        for ch in e:
            if ch.tag == "default":
                continue
            at = {k: v for k, v in ch.attrib.items() if k != "__dir"}
            d.update(ch.tag, at)
        for ch in e:
            if ch.tag == "default":
                self._do_default(ch, name)

    def _attrs(self, e, tag, childclass):
        cls = e.attrib.get("class", childclass or "main")
        if cls not in self.defaults:
            raise ValueError(f"unknown default class {cls!r}")
        at = dict(self.defaults[cls].get(tag))
        explicit = {k: v for k, v in e.attrib.items() if k not in ("class", "__dir")}
        # orientation specifiers are mutually exclusive: an explicit one overrides a default one
        ori = ("quat", "euler", "axisangle", "xyaxes", "zaxis")
        if any(k in explicit for k in ori) or "fromto" in explicit:
            for k in ori:
                at.pop(k, None)
        at.update(explicit)
        return at

    def _orientation(self, at):
        if "quat" in at:
            return quat_normalize(_floats(at["quat"], 4))
        if "euler" in at:
            return euler2quat(self._angle(_floats(at["euler"], 3)), self.comp["eulerseq"])
        if "axisangle" in at:
            v = _floats(at["axisangle"], 4)
            return axisangle2quat(v[:3], float(self._angle(v[3])))
        if "zaxis" in at:
            return z2quat(_floats(at["zaxis"], 3))
        if "xyaxes" in at:
            v = _floats(at["xyaxes"], 6)
            x = v[:3] / np.linalg.norm(v[:3])
            y = v[3:] - x * np.dot(x, v[3:])
            y /= np.linalg.norm(y)
            return mat2quat(np.stack([x, y, np.cross(x, y)], 1))
        return np.array([1.0, 0, 0, 0])

    def _do_asset(self, e):
        for ch in e:
            if ch.tag == "hfield":
                if "file" in ch.attrib:
                    raise NotImplementedError("hfield file")
                self.hfields[ch.attrib["name"]] = dict(size=_floats(ch.attrib["size"], 4), nrow=int(ch.attrib["nrow"]), ncol=int(ch.attrib["ncol"]))
            if ch.tag == "mesh":
                name = ch.attrib.get("name") or os.path.splitext(os.path.basename(ch.attrib["file"]))[0]
                self.meshes[name] = dict(file=ch.attrib["file"], dir=ch.attrib.get("__dir"),
                                         scale=_floats(ch.attrib.get("scale", "1 1 1"), 3))

    def _mesh_props(self, name):
        m = self.meshes[name]
        if "props" not in m:
            base = self.comp_dir
            p = os.path.normpath(os.path.join(base, self.comp["meshdir"], m["file"]))
            tris = _read_stl(p) * m["scale"][None, None, :]
            m["props"] = mesh_mass_properties(tris)
            m["rmax"] = float(np.linalg.norm(tris.reshape(-1, 3), axis=1).max())   # sphere about the mesh origin containing it
            if self.convex_meshes:
                from scipy.spatial import ConvexHull
                v = np.unique(tris.reshape(-1, 3), axis=0)
                hv = v[np.sort(ConvexHull(v).vertices)]
                # MuJoCo re-centres a mesh on its centre of mass; the convex narrow phase (MPR) needs the geom's centre INSIDE the shape and the
                # broad phase wants a tight sphere about it: the hull is stored about the mean of its vertices (an interior point) and the
                # geom frame origin moves there (geom_pos below)
                m["hull_center"] = hv.mean(0)
                m["hull"] = hv - m["hull_center"]
                m["hull_rmax"] = float(np.linalg.norm(m["hull"], axis=1).max())
        return m["props"]

    def _do_body(self, e, parent, childclass):
        is_world = parent < 0
        bid = len(self.bodies)
        if is_world:
            b = dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), inertial=None,
                     jnts=[], geoms=[], childclass=None)
        else:
            childclass = e.attrib.get("childclass", childclass)
            b = dict(name=e.attrib.get("name", f"body{bid}"), parent=parent,
                     pos=_floats(e.attrib.get("pos", "0 0 0"), 3), quat=self._orientation(e.attrib),
                     inertial=None, jnts=[], geoms=[], childclass=childclass)
        self.bodies.append(b)
        # MuJoCo numbers joints / dofs body by body: a body's own elements come before anything of its child bodies, wherever
        # they appear in the text (myolegs.xml puts <freejoint> after the included child chains)
        for ch in [c for c in e if c.tag != "body"] + [c for c in e if c.tag == "body"]:
            t = ch.tag
            if t == "inertial":
                a = ch.attrib
                ine = dict(pos=_floats(a.get("pos", "0 0 0"), 3), quat=self._orientation(a), mass=float(a["mass"]))
                if "fullinertia" in a:
                    f = _floats(a["fullinertia"], 6)
                    ine["full"] = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                elif "diaginertia" in a:
                    ine["diag"] = _floats(a["diaginertia"], 3)
                else:
                    ine["diag"] = np.zeros(3)      # mass only: a point mass (myotorsorigid_chain.xml head body)
                b["inertial"] = ine
            elif t == "joint" or t == "freejoint":
                at = self._attrs(ch, "joint", childclass) if t == "joint" else dict(ch.attrib, type="free")
                jt = JNT_TYPES[at.get("type", "hinge")]
                if jt in (JNT_BALL,):
                    raise NotImplementedError("ball joints are not used by the MyoSuite config models")
                rng = _floats(at.get("range", "0 0"), 2)
                if jt == JNT_HINGE:
                    rng = self._angle(rng)
                lim = at.get("limited", "auto")
                if lim == "auto":
                    limited = self.comp["autolimits"] and ("range" in at) and rng[0] < rng[1]
                else:
                    limited = self._bool(lim)
                axis = _floats(at.get("axis", "0 0 1"), 3)
                if jt != JNT_FREE:
                    axis = axis / np.linalg.norm(axis)
                ref = float(at.get("ref", 0.0))
                sref = float(at.get("springref", 0.0))
                if jt == JNT_HINGE:
                    ref = float(self._angle(ref))
                    sref = float(self._angle(sref))
                j = dict(name=at.get("name", f"jnt{len(self.joints)}"), type=jt, body=bid,
                         pos=_floats(at.get("pos", "0 0 0"), 3), axis=axis, range=rng, limited=bool(limited),
                         damping=float(at.get("damping", 0)), armature=float(at.get("armature", 0)),
                         stiffness=float(at.get("stiffness", 0)), ref=ref, springref=sref,
                         margin=float(at.get("margin", 0)),
                         solref=_floats(at.get("solreflimit", "0.02 1"), 2),
                         solimp=_solimp(at.get("solimplimit")),
                         frictionloss=float(at.get("frictionloss", 0)))
                j["solref_fri"] = _floats(at.get("solreffriction", "0.02 1"), 2)
                j["solimp_fri"] = _solimp(at.get("solimpfriction"))
                self.joints.append(j)
                b["jnts"].append(len(self.joints) - 1)
            elif t == "geom":
                at = self._attrs(ch, "geom", childclass)
                self._do_geom(at, bid)
            elif t == "site":
                at = self._attrs(ch, "site", childclass)
                self.sites.append(dict(name=at.get("name", f"site{len(self.sites)}"), body=bid,
                                       pos=_floats(at.get("pos", "0 0 0"), 3), quat=self._orientation(at)))
            elif t == "body":
                self._do_body(ch, bid, childclass)
            elif t in ("camera", "light", "include"):
                pass
            else:
                raise NotImplementedError(f"unsupported body child <{t}>")

    def _do_geom(self, at, bid):
        gtype = GEOM_TYPES[at.get("type", "sphere")]
        size = np.zeros(3)
        if "size" in at:
            s = _floats(at["size"])
            size[: s.size] = s
        pos = _floats(at.get("pos", "0 0 0"), 3)
        quat = self._orientation(at)
        if "fromto" in at:
            ft = _floats(at["fromto"], 6)
            vec = ft[3:] - ft[:3]
            size[1] = 0.5 * np.linalg.norm(vec)
            pos = 0.5 * (ft[:3] + ft[3:])
            quat = z2quat(vec)
            if gtype in (GEOM_ELLIPSOID, GEOM_BOX):
                size[2] = size[1]
                size[1] = size[0]
        g = dict(name=at.get("name", ""), type=gtype, body=bid, pos=pos, quat=quat, size=size,
                 contype=int(at.get("contype", 1)), conaffinity=int(at.get("conaffinity", 1)),
                 condim=int(at.get("condim", 3)), margin=float(at.get("margin", 0)), gap=float(at.get("gap", 0)),
                 friction=_pad(_floats(at.get("friction", "1 0.005 0.0001")), [1, 0.005, 0.0001]),
                 solref=_floats(at.get("solref", "0.02 1"), 2), solimp=_solimp(at.get("solimp")),
                 solmix=float(at.get("solmix", 1)), priority=int(at.get("priority", 0)),
                 density=float(at.get("density", 1000)), mass=float(at["mass"]) if "mass" in at else None,
                 group=int(at.get("group", 0)), mesh=at.get("mesh"), hfield=at.get("hfield"))
        if gtype == GEOM_HFIELD:
            hf = self.hfields[g["hfield"]]
            g["size"] = np.array(hf["size"][:3], float)          # (x half-extent, y half-extent, z scale); the base depth lives in hfield_size
            if self.terrain:
                g["pos"] = np.array([pos[0], pos[1], 0.0])
        self.geoms.append(g)
        self.bodies[bid]["geoms"].append(len(self.geoms) - 1)

    def _do_tendon(self, e):
        for sp in e:
            if sp.tag != "spatial":
                raise NotImplementedError(f"tendon <{sp.tag}>")
            at = self._attrs(sp, "tendon", None)
            rng = _floats(at.get("range", "0 0"), 2)
            lim = at.get("limited", "auto")
            limited = (self.comp["autolimits"] and "range" in at and rng[0] < rng[1]) if lim == "auto" else self._bool(lim)
            t = dict(name=at.get("name", f"tendon{len(self.tendons)}"), adr=len(self.wraps), limited=bool(limited),
                     range=rng, stiffness=float(at.get("stiffness", 0)), damping=float(at.get("damping", 0)),
                     margin=float(at.get("margin", 0)), springlength=_floats(at.get("springlength", "-1")),
                     solref=_floats(at.get("solreflimit", "0.02 1"), 2), solimp=_solimp(at.get("solimplimit")),
                     frictionloss=float(at.get("frictionloss", 0)))
            for w in sp:
                if w.tag == "site":
                    self.wraps.append(dict(type=WRAP_SITE, obj=w.attrib["site"], prm=0.0))
                elif w.tag == "geom":
                    self.wraps.append(dict(type=-1, obj=w.attrib["geom"], prm=w.attrib.get("sidesite")))
                elif w.tag == "pulley":
                    self.wraps.append(dict(type=WRAP_PULLEY, obj=None, prm=float(w.attrib["divisor"])))
                else:
                    raise NotImplementedError(f"wrap <{w.tag}>")
            t["num"] = len(self.wraps) - t["adr"]
            self.tendons.append(t)

    def _do_actuator(self, e):
        for a in e:
            if a.tag not in _ACT_TAGS:
                raise NotImplementedError(f"actuator <{a.tag}>")
            at = self._attrs(a, a.tag, None)
            act = dict(name=at.get("name", f"act{len(self.actuators)}"))
            if "tendon" in at:
                act["trntype"], act["target"] = "tendon", at["tendon"]
            elif "joint" in at:
                act["trntype"], act["target"] = "joint", at["joint"]
            else:
                raise NotImplementedError("actuator transmission")
            act["gear"] = _pad(_floats(at.get("gear", "1")), [1, 0, 0, 0, 0, 0])
            ctrlrange = _floats(at.get("ctrlrange", "0 0"), 2)
            cl = at.get("ctrllimited", "auto")
            act["ctrllimited"] = (self.comp["autolimits"] and "ctrlrange" in at and ctrlrange[0] < ctrlrange[1]) \
                if cl == "auto" else self._bool(cl)
            act["ctrlrange"] = ctrlrange
            forcerange = _floats(at.get("forcerange", "0 0"), 2)
            fl = at.get("forcelimited", "auto")
            act["forcelimited"] = (self.comp["autolimits"] and "forcerange" in at and forcerange[0] < forcerange[1]) \
                if fl == "auto" else self._bool(fl)
            act["forcerange"] = forcerange
            act["lengthrange"] = _floats(at.get("lengthrange", "0 0"), 2)
            act["has_lengthrange"] = "lengthrange" in at
            if a.tag == "muscle":
                tc = _floats(at.get("timeconst", "0.01 0.04"), 2)
                rng = _floats(at.get("range", "0.75 1.05"), 2)
                prm = np.array([rng[0], rng[1], float(at.get("force", -1)), float(at.get("scale", 200)),
                                float(at.get("lmin", 0.5)), float(at.get("lmax", 1.6)), float(at.get("vmax", 1.5)),
                                float(at.get("fpmax", 1.3)), float(at.get("fvmax", 1.2))])
                act["dyntype"] = act["gaintype"] = act["biastype"] = "muscle"
                act["dynprm"] = np.array([tc[0], tc[1], float(at.get("tausmooth", 0))])
                act["gainprm"] = prm.copy()
                act["biasprm"] = prm.copy()
            elif a.tag in ("motor", "position", "velocity"):
                # shortcuts for stateless affine actuators: force = gainprm[0] * ctrl + biasprm[0] + biasprm[1] * length + biasprm[2] * velocity
                act["dyntype"], act["gaintype"] = "none", "fixed"
                act["dynprm"] = np.array([1.0, 0, 0])
                gp, bp = np.zeros(9), np.zeros(9)
                if a.tag == "motor":
                    gp[0] = 1.0
                    act["biastype"] = "none"
                elif a.tag == "position":
                    kp, kv = float(at.get("kp", 1)), float(at.get("kv", 0))
                    gp[0], bp[1], bp[2] = kp, -kp, -kv
                    act["biastype"] = "affine"
                else:
                    kv = float(at.get("kv", 1))
                    gp[0], bp[2] = kv, -kv
                    act["biastype"] = "affine"
                act["gainprm"], act["biasprm"] = gp, bp
            else:
                act["dyntype"] = at.get("dyntype", "none")
                act["gaintype"] = at.get("gaintype", "fixed")
                act["biastype"] = at.get("biastype", "none")
                affine = act["dyntype"] == "none" and act["gaintype"] == "fixed" and act["biastype"] in ("none", "affine")
                if not (act["dyntype"] == act["gaintype"] == act["biastype"] == "muscle") and not affine:
                    raise NotImplementedError("general actuators: only muscle-type, or stateless fixed-gain / affine-bias ones")
                act["dynprm"] = _pad(_floats(at.get("dynprm", "1")), [1, 0, 0])[:3]
                act["gainprm"] = _pad(_floats(at.get("gainprm", "1")), [1] + [0] * 8)[:9]
                act["biasprm"] = _pad(_floats(at.get("biasprm", "0")), [0] * 9)[:9]
                if affine and act["biastype"] == "none":
                    act["biasprm"] = np.zeros(9)
            # kind 0: muscle (activation state + FLV curves); kind 1: stateless affine (motor / position / velocity / general)
            act["kind"] = 0 if act["dyntype"] == "muscle" else 1
            self.actuators.append(act)

    def _do_contact(self, e):
        for c in e:
            if c.tag == "exclude":
                self.excludes.append((c.attrib["body1"], c.attrib["body2"]))
            elif c.tag == "pair":
                self.pairs.append({k: v for k, v in c.attrib.items() if k != "__dir"})
            else:
                raise NotImplementedError(c.tag)

    # ---- driver
    def compile(self):
        root = self._load(self.path)
        # pass 1: compiler/option/default/asset (global, order independent)
        for e in root:
            if e.tag == "compiler":
                self._do_compiler(e)
        for e in root:
            if e.tag == "option":
                self._do_option(e)
            elif e.tag == "default":
                self._do_default(e, None)
            elif e.tag == "asset":
                self._do_asset(e)
        # pass 2: worldbody (possibly several, merged)
        self._do_body(ET.Element("worldbody"), -1, None)
        self.bodies[0]["_elem_done"] = True
        for e in root:
            if e.tag == "worldbody":
                self._merge_world(e)
        for e in root:
            if e.tag == "tendon":
                self._do_tendon(e)
            elif e.tag == "actuator":
                self._do_actuator(e)
            elif e.tag == "contact":
                self._do_contact(e)
            elif e.tag == "keyframe":
                for k in e:
                    self.keys.append({kk: vv for kk, vv in k.attrib.items() if kk != "__dir"})
            elif e.tag == "equality":
                for q in e:
                    self.equalities.append((q.tag, {kk: vv for kk, vv in q.attrib.items() if kk != "__dir"}))
            elif e.tag in ("compiler", "option", "default", "asset", "worldbody", "size", "visual", "statistic",
                           "sensor", "custom", "extension"):
                pass
            else:
                raise NotImplementedError(f"unsupported section <{e.tag}>")
        return self._finalize()

    def _merge_world(self, e):
        # children of <worldbody> attach to body 0
        for ch in e:
            t = ch.tag
            if t == "geom":
                self._do_geom(self._attrs(ch, "geom", None), 0)
            elif t == "site":
                at = self._attrs(ch, "site", None)
                self.sites.append(dict(name=at.get("name", f"site{len(self.sites)}"), body=0,
                                       pos=_floats(at.get("pos", "0 0 0"), 3), quat=self._orientation(at)))
            elif t == "body":
                self._do_body(ch, 0, None)
            elif t in ("camera", "light"):
                pass
            else:
                raise NotImplementedError(f"unsupported worldbody child <{t}>")

    # ---- finalisation: numbering, inertias, derived constants
    def _finalize(self):
        # MuJoCo numbers bodies depth-first in document order.  _do_body appended them in
        # document order already, but worldbody-level geoms/sites interleave with bodies, which
        # only affects geom/site ids: renumber geoms and sites grouped by body id (stable).
        nb = len(self.bodies)
        gorder = sorted(range(len(self.geoms)), key=lambda i: self.geoms[i]["body"])
        sorder = sorted(range(len(self.sites)), key=lambda i: self.sites[i]["body"])
        self.geoms = [self.geoms[i] for i in gorder]
        self.sites = [self.sites[i] for i in sorder]
        for b in self.bodies:
            b["geoms"] = []
        for gi, g in enumerate(self.geoms):
            self.bodies[g["body"]]["geoms"].append(gi)
        # joints were appended in body order already (body ids are document order)
        A = {}
        names = dict(body=[b["name"] for b in self.bodies], joint=[j["name"] for j in self.joints],
                     geom=[g["name"] for g in self.geoms], site=[s["name"] for s in self.sites],
                     tendon=[t["name"] for t in self.tendons], actuator=[a["name"] for a in self.actuators])
        njnt = len(self.joints)
        # qpos / dof addressing
        qadr, dadr = 0, 0
        jnt_qposadr, jnt_dofadr = [], []
        for j in self.joints:
            jnt_qposadr.append(qadr)
            jnt_dofadr.append(dadr)
            qadr += {JNT_FREE: 7, JNT_BALL: 4, JNT_SLIDE: 1, JNT_HINGE: 1}[j["type"]]
            dadr += {JNT_FREE: 6, JNT_BALL: 3, JNT_SLIDE: 1, JNT_HINGE: 1}[j["type"]]
        nq, nv = qadr, dadr
        A["body_parentid"] = np.array([b["parent"] for b in self.bodies], np.int32)
        A["body_pos"] = np.stack([b["pos"] for b in self.bodies])
        A["body_quat"] = np.stack([b["quat"] for b in self.bodies])
        A["body_jntadr"] = np.array([b["jnts"][0] if b["jnts"] else -1 for b in self.bodies], np.int32)
        A["body_jntnum"] = np.array([len(b["jnts"]) for b in self.bodies], np.int32)
        body_dofnum = np.zeros(nb, np.int32)
        body_dofadr = np.full(nb, -1, np.int32)
        for bi, b in enumerate(self.bodies):
            for ji in b["jnts"]:
                n = {JNT_FREE: 6, JNT_BALL: 3, JNT_SLIDE: 1, JNT_HINGE: 1}[self.joints[ji]["type"]]
                if body_dofadr[bi] < 0:
                    body_dofadr[bi] = jnt_dofadr[ji]
                body_dofnum[bi] += n
        A["body_dofadr"], A["body_dofnum"] = body_dofadr, body_dofnum
        weld = np.zeros(nb, np.int32)
        rootid = np.zeros(nb, np.int32)
        for bi in range(1, nb):
            p = self.bodies[bi]["parent"]
            weld[bi] = bi if self.bodies[bi]["jnts"] else weld[p]
            rootid[bi] = bi if p == 0 else rootid[p]
        A["body_weldid"], A["body_rootid"] = weld, rootid
        # inertias
        mass = np.zeros(nb)
        ipos = np.zeros((nb, 3))
        iquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
        inertia = np.zeros((nb, 3))
        glo, ghi = self.comp["inertiagrouprange"]
        for bi, b in enumerate(self.bodies):
            if bi == 0:
                continue
            mode = self.comp["inertiafromgeom"]
            use_geoms = mode == "true" or (mode == "auto" and b["inertial"] is None)
            static = weld[bi] == 0
            if use_geoms:
                if static:
                    continue  # never enters the dynamics; skip loading big visual meshes
                m, c, I = 0.0, np.zeros(3), np.zeros((3, 3))
                parts = []
                for gi in b["geoms"]:
                    g = self.geoms[gi]
                    if not (glo <= g["group"] <= ghi):
                        continue
                    R = quat2mat(g["quat"])
                    if g["type"] == GEOM_MESH:
                        V, mc, mI = self._mesh_props(g["mesh"])
                        gm = g["mass"] if g["mass"] is not None else g["density"] * V
                        Ig = R @ (mI * (gm / V)) @ R.T
                        pc = g["pos"] + R @ mc
                    elif g["type"] in (GEOM_PLANE, GEOM_HFIELD):
                        continue
                    else:
                        V, Id = geom_volume_inertia(g["type"], g["size"])
                        gm = g["mass"] if g["mass"] is not None else g["density"] * V
                        Ig = R @ np.diag(Id * (gm / V)) @ R.T
                        pc = g["pos"]
                    parts.append((gm, pc, Ig))
                    m += gm
                    c += gm * pc
                if m > 0:
                    c /= m
                    for gm, pc, Ig in parts:
                        d = pc - c
                        I += Ig + gm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
                mass[bi], ipos[bi] = m, c
                full = I
                have = m > 0
            else:
                ine = b["inertial"]
                mass[bi], ipos[bi] = ine["mass"], ine["pos"]
                if "full" in ine:
                    full = ine["full"]
                    have = True
                else:
                    inertia[bi] = ine["diag"]
                    iquat[bi] = ine["quat"]
                    have = False
            if have:
                w, V = np.linalg.eigh(full)
                # principal axes, largest first (MuJoCo sorts descending), right-handed
                order = np.argsort(-w)
                w, V = w[order], V[:, order]
                if np.linalg.det(V) < 0:
                    V[:, 2] = -V[:, 2]
                inertia[bi] = w
                iquat[bi] = mat2quat(V)
            # bounds, then balance (MuJoCo order)
            mass[bi] = max(mass[bi], self.comp["boundmass"])
            inertia[bi] = np.maximum(inertia[bi], self.comp["boundinertia"])
            a, bb, c3 = inertia[bi]
            if a + bb < c3 or a + c3 < bb or bb + c3 < a:
                if self.comp["balanceinertia"]:
                    inertia[bi] = (a + bb + c3) / 3.0
                elif not static:
                    raise ValueError(f"body {b['name']}: inertia violates A+B>=C")
        A["body_mass"], A["body_ipos"], A["body_iquat"], A["body_inertia"] = mass, ipos, iquat, inertia
        # joints / dofs
        A["jnt_type"] = np.array([j["type"] for j in self.joints], np.int32)
        A["jnt_qposadr"] = np.array(jnt_qposadr, np.int32)
        A["jnt_dofadr"] = np.array(jnt_dofadr, np.int32)
        A["jnt_bodyid"] = np.array([j["body"] for j in self.joints], np.int32)
        A["jnt_pos"] = np.stack([j["pos"] for j in self.joints]) if njnt else np.zeros((0, 3))
        A["jnt_axis"] = np.stack([j["axis"] for j in self.joints]) if njnt else np.zeros((0, 3))
        A["jnt_range"] = np.stack([j["range"] for j in self.joints]) if njnt else np.zeros((0, 2))
        A["jnt_limited"] = np.array([j["limited"] for j in self.joints], np.int32)
        A["jnt_margin"] = np.array([j["margin"] for j in self.joints])
        A["jnt_solref"] = np.stack([j["solref"] for j in self.joints]) if njnt else np.zeros((0, 2))
        A["jnt_solimp"] = np.stack([j["solimp"] for j in self.joints]) if njnt else np.zeros((0, 5))
        A["jnt_stiffness"] = np.array([j["stiffness"] for j in self.joints])
        qpos0 = np.zeros(nq)
        qspring = np.zeros(nq)
        dof_bodyid = np.zeros(nv, np.int32)
        dof_jntid = np.zeros(nv, np.int32)
        dof_armature = np.zeros(nv)
        dof_damping = np.zeros(nv)
        for ji, j in enumerate(self.joints):
            qa, da = jnt_qposadr[ji], jnt_dofadr[ji]
            if j["type"] == JNT_FREE:
                b = self.bodies[j["body"]]
                qpos0[qa:qa + 3] = b["pos"]
                qpos0[qa + 3:qa + 7] = b["quat"]
                qspring[qa:qa + 7] = qpos0[qa:qa + 7]
                n = 6
            else:
                qpos0[qa] = j["ref"]
                qspring[qa] = j["springref"]
                n = 1
            for k in range(n):
                dof_bodyid[da + k] = j["body"]
                dof_jntid[da + k] = ji
                dof_armature[da + k] = j["armature"]
                dof_damping[da + k] = j["damping"]
        A["qpos0"], A["qpos_spring"] = qpos0, qspring
        A["dof_bodyid"], A["dof_jntid"] = dof_bodyid, dof_jntid
        A["dof_armature"], A["dof_damping"] = dof_armature, dof_damping
        A["dof_frictionloss"] = np.array([self.joints[dof_jntid[d]]["frictionloss"] for d in range(nv)])
        A["dof_solref_fri"] = np.stack([self.joints[dof_jntid[d]]["solref_fri"] for d in range(nv)]) if nv else np.zeros((0, 2))
        A["dof_solimp_fri"] = np.stack([self.joints[dof_jntid[d]]["solimp_fri"] for d in range(nv)]) if nv else np.zeros((0, 5))
        # dof parent: previous dof in the same body, else last dof of nearest ancestor with dofs
        dof_parent = np.full(nv, -1, np.int32)
        for d in range(nv):
            b = dof_bodyid[d]
            if d > body_dofadr[b]:
                dof_parent[d] = d - 1
            else:
                p = self.bodies[b]["parent"]
                while p > 0 and body_dofnum[p] == 0:
                    p = self.bodies[p]["parent"]
                if p > 0:
                    dof_parent[d] = body_dofadr[p] + body_dofnum[p] - 1
        A["dof_parentid"] = dof_parent
        madr = np.zeros(nv, np.int32)
        nM = 0
        for d in range(nv):
            madr[d] = nM
            k = d
            while k >= 0:
                nM += 1
                k = dof_parent[k]
        A["dof_Madr"] = madr
        # geoms: keep everything that collides or wraps (visual-only geoms dropped)
        wrap_geoms = {w["obj"] for w in self.wraps if w["type"] == -1}
        keep = [i for i, g in enumerate(self.geoms)
                if (g["contype"] or g["conaffinity"]) or (g["name"] and g["name"] in wrap_geoms)]
        # mesh geoms: visual unless their contype / conaffinity bits can match some other geom's (myoelbow: contype 1, conaffinity 0
        # everywhere -> nothing collides and the bone meshes only contribute inertia)
        def _can_pair(i):
            a = self.geoms[i]
            return any(j != i and ((a["contype"] & b["conaffinity"]) or (b["contype"] & a["conaffinity"])) for j, b in enumerate(self.geoms))
        # A mesh that can pair is kept as a bounding sphere only: the lowering must prove each of its pairs out of reach (it raises
        # otherwise), and the oracle's bounding-sphere filter then never lets one reach a narrow phase (it flags it if one does)
        for i in list(keep):
            if self.geoms[i]["type"] == GEOM_MESH:
                if _can_pair(i):
                    self._mesh_props(self.geoms[i]["mesh"])
                    self.geoms[i]["size"] = np.array([self.meshes[self.geoms[i]["mesh"]]["rmax"], 0.0, 0.0])
                    if self.convex_meshes:      # (after the inertia pass above, which used the file's frame)
                        mm = self.meshes[self.geoms[i]["mesh"]]
                        self.geoms[i] = dict(self.geoms[i], pos=self.geoms[i]["pos"] + quat2mat(self.geoms[i]["quat"]) @ mm["hull_center"],
                                             size=np.array([mm["hull_rmax"], 0.0, 0.0]))
                else:
                    keep.remove(i)
        # convex_meshes: a colliding mesh geom is its convex hull (vertex list in the geom frame; support = best vertex)
        mesh_vert, mesh_adr = [], {}
        if self.convex_meshes:
            for i in keep:
                g = self.geoms[i]
                if g["type"] == GEOM_MESH and g["mesh"] not in mesh_adr:
                    h = self.meshes[g["mesh"]]["hull"]
                    mesh_adr[g["mesh"]] = (sum(len(x) for x in mesh_vert), len(h))
                    mesh_vert.append(h)
        # explicit pairs may name geoms that neither collide dynamically nor wrap: keep them too
        pair_names = {pr[k] for pr in self.pairs for k in ("geom1", "geom2")}
        keep = sorted(set(keep) | {i for i, g in enumerate(self.geoms) if g["name"] and g["name"] in pair_names})
        geoms = [self.geoms[i] for i in keep]
        names["geom"] = [g["name"] for g in geoms]
        ng = len(geoms)
        A["geom_type"] = np.array([g["type"] for g in geoms], np.int32)
        A["geom_bodyid"] = np.array([g["body"] for g in geoms], np.int32)
        A["geom_pos"] = np.stack([g["pos"] for g in geoms]) if ng else np.zeros((0, 3))
        A["geom_quat"] = np.stack([g["quat"] for g in geoms]) if ng else np.zeros((0, 4))
        A["geom_size"] = np.stack([g["size"] for g in geoms]) if ng else np.zeros((0, 3))
        A["geom_contype"] = np.array([g["contype"] for g in geoms], np.int32)
        A["geom_conaffinity"] = np.array([g["conaffinity"] for g in geoms], np.int32)
        A["geom_condim"] = np.array([g["condim"] for g in geoms], np.int32)
        A["geom_priority"] = np.array([g["priority"] for g in geoms], np.int32)
        A["geom_margin"] = np.array([g["margin"] for g in geoms])
        A["geom_gap"] = np.array([g["gap"] for g in geoms])
        A["geom_solmix"] = np.array([g["solmix"] for g in geoms])
        A["geom_friction"] = np.stack([g["friction"] for g in geoms]) if ng else np.zeros((0, 3))
        A["geom_solref"] = np.stack([g["solref"] for g in geoms]) if ng else np.zeros((0, 2))
        A["geom_solimp"] = np.stack([g["solimp"] for g in geoms]) if ng else np.zeros((0, 5))
        A["geom_rbound"] = np.array([geom_rbound(g["type"], g["size"]) for g in geoms])
        A["geom_meshadr"] = np.array([mesh_adr[g["mesh"]][0] if (g["type"] == GEOM_MESH and g["mesh"] in mesh_adr) else -1 for g in geoms], np.int32)
        A["geom_meshnum"] = np.array([mesh_adr[g["mesh"]][1] if (g["type"] == GEOM_MESH and g["mesh"] in mesh_adr) else 0 for g in geoms], np.int32)
        A["mesh_vert"] = np.concatenate(mesh_vert) if mesh_vert else np.zeros((0, 3))
        # sites
        A["site_bodyid"] = np.array([s["body"] for s in self.sites], np.int32)
        A["site_pos"] = np.stack([s["pos"] for s in self.sites]) if self.sites else np.zeros((0, 3))
        # tendons + wraps
        nt = len(self.tendons)
        wtype, wobj, wprm = [], [], []
        for w in self.wraps:
            if w["type"] == WRAP_SITE:
                wtype.append(WRAP_SITE); wobj.append(names["site"].index(w["obj"])); wprm.append(0.0)
            elif w["type"] == WRAP_PULLEY:
                wtype.append(WRAP_PULLEY); wobj.append(-1); wprm.append(w["prm"])
            else:
                gi = names["geom"].index(w["obj"])
                gt = geoms[gi]["type"]
                if gt == GEOM_SPHERE:
                    wtype.append(WRAP_SPHERE)
                elif gt == GEOM_CYLINDER:
                    wtype.append(WRAP_CYLINDER)
                else:
                    raise ValueError(f"wrap geom {w['obj']} must be sphere or cylinder")
                wobj.append(gi)
                wprm.append(float(names["site"].index(w["prm"])) if w["prm"] else -1.0)
        A["wrap_type"] = np.array(wtype, np.int32)
        A["wrap_objid"] = np.array(wobj, np.int32)
        A["wrap_prm"] = np.array(wprm)
        A["tendon_adr"] = np.array([t["adr"] for t in self.tendons], np.int32)
        A["tendon_num"] = np.array([t["num"] for t in self.tendons], np.int32)
        A["tendon_limited"] = np.array([t["limited"] for t in self.tendons], np.int32)
        A["tendon_range"] = np.stack([t["range"] for t in self.tendons]) if nt else np.zeros((0, 2))
        A["tendon_margin"] = np.array([t["margin"] for t in self.tendons])
        A["tendon_stiffness"] = np.array([t["stiffness"] for t in self.tendons])
        A["tendon_damping"] = np.array([t["damping"] for t in self.tendons])
        A["tendon_solref"] = np.stack([t["solref"] for t in self.tendons]) if nt else np.zeros((0, 2))
        A["tendon_solimp"] = np.stack([t["solimp"] for t in self.tendons]) if nt else np.zeros((0, 5))
        for t in self.tendons:
            if t["frictionloss"] != 0:
                raise NotImplementedError("tendon frictionloss")
        # actuators
        nu = len(self.actuators)
        A["actuator_trnid"] = np.array(
            [names["tendon"].index(a["target"]) if a["trntype"] == "tendon" else names["joint"].index(a["target"])
             for a in self.actuators], np.int32)
        A["actuator_trntype"] = np.array([1 if a["trntype"] == "tendon" else 0 for a in self.actuators], np.int32)
        A["actuator_gear"] = np.array([a["gear"][0] for a in self.actuators])
        A["actuator_dynprm"] = np.stack([a["dynprm"] for a in self.actuators]) if nu else np.zeros((0, 3))
        A["actuator_gainprm"] = np.stack([a["gainprm"] for a in self.actuators]) if nu else np.zeros((0, 9))
        A["actuator_biasprm"] = np.stack([a["biasprm"] for a in self.actuators]) if nu else np.zeros((0, 9))
        A["actuator_ctrllimited"] = np.array([a["ctrllimited"] for a in self.actuators], np.int32)
        A["actuator_ctrlrange"] = np.stack([a["ctrlrange"] for a in self.actuators]) if nu else np.zeros((0, 2))
        A["actuator_forcelimited"] = np.array([a["forcelimited"] for a in self.actuators], np.int32)
        A["actuator_forcerange"] = np.stack([a["forcerange"] for a in self.actuators]) if nu else np.zeros((0, 2))
        A["actuator_lengthrange"] = np.stack([a["lengthrange"] for a in self.actuators]) if nu else np.zeros((0, 2))
        A["actuator_has_lengthrange"] = np.array([a["has_lengthrange"] for a in self.actuators], np.int32)
        A["actuator_acc0"] = np.zeros(nu)
        A["actuator_kind"] = np.array([a["kind"] for a in self.actuators], np.int32)
        # collision pair table (static part of mj_collision's filtering)
        excl = set()
        for b1, b2 in self.excludes:
            i1, i2 = names["body"].index(b1), names["body"].index(b2)
            excl.add((min(i1, i2), max(i1, i2)))
        pairs = []
        for g1 in range(ng):
            for g2 in range(g1 + 1, ng):
                a, b = geoms[g1], geoms[g2]
                if not ((a["contype"] & b["conaffinity"]) or (b["contype"] & a["conaffinity"])):
                    continue
                b1, b2 = a["body"], b["body"]
                w1, w2 = weld[b1], weld[b2]
                if w1 == w2:
                    continue
                wp1 = weld[self.bodies[w1]["parent"]] if w1 else 0
                wp2 = weld[self.bodies[w2]["parent"]] if w2 else 0
                if w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
                    continue
                if (min(b1, b2), max(b1, b2)) in excl:
                    continue
                pairs.append((g1, g2))
        # heightfield geoms: kept as geoms, but their pairs are dropped (myoLegWalk-v0 parks the terrain below the floor plane,
        # envs/myo/myobase/walk_v0.py:257-261); recorded so that callers can see it
        nhf = sum(1 for (a, b) in pairs if geoms[a]["type"] == GEOM_HFIELD or geoms[b]["type"] == GEOM_HFIELD)
        hf_geoms = [i for i, g in enumerate(geoms) if g["type"] == GEOM_HFIELD]
        if len(hf_geoms) > 1:
            raise NotImplementedError("more than one height field geom")
        if self.terrain:
            # height-field model variant (TerrainEnvV0): pairs kept, hfield geom first in each pair (mj_collision orders by geom type)
            nhf = 0
            g = geoms[hf_geoms[0]]
            if not np.allclose(g["quat"], [1, 0, 0, 0]) or self.bodies[g["body"]]["parent"] >= 0 and weld[g["body"]] != 0:
                raise NotImplementedError("height field geom must be world-fixed and axis-aligned")
            pairs = [(b, a) if geoms[b]["type"] == GEOM_HFIELD else (a, b) for (a, b) in pairs]
            for (a, b) in pairs:
                if geoms[a]["type"] == GEOM_HFIELD and geoms[b]["type"] not in (GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER):
                    raise NotImplementedError("height field against a non-convex-primitive geom")
        else:
            pairs = [(a, b) for (a, b) in pairs if geoms[a]["type"] != GEOM_HFIELD and geoms[b]["type"] != GEOM_HFIELD]
        A["dropped_hfield_pairs"] = np.array([nhf], np.int32)
        hf = self.hfields[geoms[hf_geoms[0]]["hfield"]] if hf_geoms else None
        A["hfield_size"] = np.array(hf["size"], float) if hf else np.zeros(4)            # x, y half-extents, z scale, base depth
        A["hfield_dims"] = np.array([hf["nrow"], hf["ncol"], hf_geoms[0] if (hf and self.terrain) else -1] if hf else [0, 0, -1], np.int32)
        # explicit <contact><pair>: bypass the contype / parent filters; condim etc. from the pair element
        pair_condim = [-1] * len(pairs)          # -1: derive from the geoms (dynamic pair)
        for pr in self.pairs:
            g1, g2 = names["geom"].index(pr["geom1"]), names["geom"].index(pr["geom2"])
            for k in pr:
                if k not in ("geom1", "geom2", "condim", "name"):
                    raise NotImplementedError(f"<pair {k}=...>")
            pairs.append((g1, g2))
            pair_condim.append(int(pr.get("condim", 3)))
        A["pair_geom"] = np.array(pairs, np.int32).reshape(-1, 2)
        A["pair_condim"] = np.array(pair_condim, np.int32)
        # keyframes
        kq = []
        for k in self.keys:
            q = qpos0.copy()
            if "qpos" in k:
                q = _floats(k["qpos"], nq)
            kq.append(q)
        A["key_qpos"] = np.stack(kq) if kq else np.zeros((0, nq))
        A["key_qvel"] = np.stack([_floats(k["qvel"], nv) if "qvel" in k else np.zeros(nv) for k in self.keys]) if kq else np.zeros((0, nv))
        # options
        o = self.opt
        if o["cone"] != "pyramidal" or o["solver"] != "Newton" or o["integrator"] not in ("Euler", "RK4"):
            raise NotImplementedError("only pyramidal cones, the Newton solver and Euler / RK4 integration are restated")
        A["integrator"] = np.array([1 if o["integrator"] == "RK4" else 0], np.int32)
        A["opt"] = np.array([o["timestep"], o["gravity"][0], o["gravity"][1], o["gravity"][2], o["tolerance"],
                             float(o["iterations"]), float(o["ls_iterations"]), o["ls_tolerance"], o["impratio"],
                             0.0])  # last slot: stat.meaninertia, filled by setconst
        A["sizes"] = np.array([nq, nv, nu, nu, nb, njnt, ng, len(self.sites), nt, len(wtype), len(pairs), nM,
                               len(self.equalities)], np.int32)
        # equality constraints: joint couplings q1 - q1_0 = poly(q2 - q2_0)
        eq_j1, eq_j2, eq_data, eq_solref, eq_solimp = [], [], [], [], []
        for tag, at in self.equalities:
            if tag != "joint":
                raise NotImplementedError(f"equality <{tag}>")
            if at.get("active", "true") != "true":
                continue
            eq_j1.append(names["joint"].index(at["joint1"]))
            eq_j2.append(names["joint"].index(at["joint2"]) if "joint2" in at else -1)
            eq_data.append(_pad(_floats(at.get("polycoef", "0 1 0 0 0")), [0, 1, 0, 0, 0]))
            eq_solref.append(_floats(at.get("solref", "0.02 1"), 2))
            eq_solimp.append(_solimp(at.get("solimp")))
        ne = len(eq_j1)
        A["eq_obj1id"] = np.array(eq_j1, np.int32)
        A["eq_obj2id"] = np.array(eq_j2, np.int32)
        A["eq_data"] = np.stack(eq_data) if ne else np.zeros((0, 5))
        A["eq_solref"] = np.stack(eq_solref) if ne else np.zeros((0, 2))
        A["eq_solimp"] = np.stack(eq_solimp) if ne else np.zeros((0, 5))
        A["sizes"][12] = ne
        cm = CompiledModel(arrays=A, names=names, source=self.path)
        return cm


def _solimp(s):
    v = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
    if s:
        f = _floats(s)
        v[: f.size] = f
    return v


def _pad(v, default):
    out = np.array(default, float)
    out[: min(v.size, out.size)] = v[: out.size]
    return out


def compile_mjcf(path, terrain=False, replace=None, convex_meshes=False) -> CompiledModel:
    """Compile an MJCF file into flat arrays (see module docstring).  terrain=True: the model variant the reference's TerrainEnvV0
    creates at reset (height-field geom raised to z = 0 and colliding); False: its pairs are dropped (myoLegWalk-v0 parks it at -10 m)."""
    return _Compiler(path, terrain, replace, convex_meshes).compile()
