"""Attribution of every MyoHand `lengthrange` golden (myohand_assets.xml:501-539) that the oracle does not reproduce on the
shipped geometry.  Build container only (reads the reference's model files).

The reference's XML keeps its own edit history in comments: former site coordinates next to the live ones, former wrapping
geoms, wrapping entries removed from tendon paths, and a whole older block of MuJoCo-computed ranges (:540-578).  For each
muscle this tool recompiles the model with parts of that history restored (text patches applied in memory, nothing is
written into the reference tree) and searches the revision on which the oracle reproduces the stored range.

    python tools/lengthrange_history.py [muscle ...]   ->  tests/golden/myohand_lengthrange_attribution.json (+ table on stdout)"""
import io
import itertools
import json
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from myosuite_mjx_amd import blob, mjcf  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from test_oracle import _extremum  # noqa: E402

REF = "/root/reference/myosuite"
TOP = REF + "/envs/myo/assets/hand/myohand_pose.xml"
BODY = os.path.normpath(REF + "/simhive/myo_sim/hand/assets/myohand_body.xml")
ASSETS = os.path.normpath(REF + "/simhive/myo_sim/hand/assets/myohand_assets.xml")
GOLD = json.load(open(os.path.join(ROOT, "tests/golden/myohand_xml_goldens.json")))


class _PatchedET:
    """Stand-in for the `ET` name inside mjcf.py: files listed in `texts` are parsed from memory."""
    def __init__(self, texts):
        self.texts = texts

    def parse(self, path):
        p = os.path.normpath(path)
        if p in self.texts:
            return ET.ElementTree(ET.fromstring(self.texts[p]))
        return ET.parse(path)

    def __getattr__(self, k):
        return getattr(ET, k)


def compile_variant(sites=(), geoms=(), wraps=(), drop=()):
    body, assets = open(BODY).read(), open(ASSETS).read()
    for s in sites:                       # former coordinates of a live site
        old = GOLD["former_site_pos"][s]["pos"]
        body, n = re.subn(r'(?<!<!-- )(<site\s+name="%s"\s+pos=")[^"]+(")' % re.escape(s), lambda m: m.group(1) + " ".join(map(str, old)) + m.group(2), body)
        # the substitution also rewrites the commented copy: harmless (same value)
        assert n >= 1, s
    for g in geoms:                       # former definition of a live wrapping geom
        d = GOLD["former_geoms"][g]
        new = '<geom name="%s" pos="%s" quat="%s" class="wrap" size="%s" type="%s"/>' % (
            g, " ".join(map(str, d["pos"])), " ".join(map(str, d["quat"])), " ".join(map(str, d["size"])), d["type"])
        nocom = re.sub(r'<!--.*?-->', lambda m: " " * len(m.group(0)), body, flags=re.S)
        m = re.search(r'<geom\s+name="%s"[^>]*>' % g, nocom)
        body = body[:m.start()] + new + body[m.end():]
    for t, g in wraps:                    # wrapping entry removed from a tendon path: uncomment it
        sp = re.search(r'(<spatial[^>]*name="%s"[^>]*>)(.*?)(</spatial>)' % t, assets, re.S)
        inner = re.sub(r'<!--\s*(<(?:geom|site) (?:geom|site)="%s"[^>]*/>)\s*-->' % g, r'\1', sp.group(2))
        assets = assets[:sp.start(2)] + inner + assets[sp.end(2):]
    for t, g in drop:                     # wrapping entry written in the later editor's syntax (<geom ...></geom>): take it out again
        sp = re.search(r'(<spatial[^>]*name="%s"[^>]*>)(.*?)(</spatial>)' % t, assets, re.S)
        inner = re.sub(r'<geom geom="%s"[^>]*></geom>' % g, '', sp.group(2))
        assets = assets[:sp.start(2)] + inner + assets[sp.end(2):]
    if any(g == "PL_ellipsoid_wrap" for _, g in wraps):
        body = re.sub(r'<!--\s*(<geom name="PL_ellipsoid_wrap"[^>]*/>)\s*-->', r'\1', body)
    saved = mjcf.ET
    mjcf.ET = _PatchedET({BODY: body, ASSETS: assets})
    try:
        cm = mjcf.compile_mjcf(TOP)
    finally:
        mjcf.ET = saved
    return cm


def ranges(cm, name):
    A = dict(cm.arrays)
    nb, nv, nt = len(A["body_parentid"]), len(A["dof_bodyid"]), len(A["tendon_adr"])
    A.setdefault("body_invweight0", np.ones((nb, 2)))
    A.setdefault("dof_invweight0", np.ones(nv))
    A.setdefault("tendon_invweight0", np.ones(nt))
    A.setdefault("actuator_acc0", np.ones(len(A["actuator_trnid"])))
    stderr = os.dup(2)
    o = Oracle(blob.pack(A))
    i = cm.names["actuator"].index(name)
    t = int(np.asarray(A["actuator_trnid"]).reshape(len(cm.names["actuator"]), -1)[i].ravel()[0])

    class _M:
        jnt_range = np.asarray(A["jnt_range"], float).reshape(-1, 2)
        qpos0 = np.asarray(A["qpos0"], float)
        ntendon, nv = nt, len(A["dof_bodyid"])
    os.close(stderr)
    return _extremum(o, _M, t, +1), _extremum(o, _M, t, -1)


def err(lo, hi, ref):
    return max(abs(lo - ref[0]), abs(hi - ref[1])) / (ref[1] - ref[0])


def main(only):
    base = compile_variant()
    snames = base.names["site"]
    out = {}
    for name in base.names["actuator"]:
        if only and name not in only:
            continue
        live, old = GOLD["lengthrange_live"][name], GOLD["lengthrange_commented"][name]
        path = GOLD["tendon_paths"][name + "_tendon"]
        psites = [n for k, n, _ in path if k == "site"] + [sd for k, _, sd in path if k == "geom" and sd]
        fs = [s for s in psites if s in GOLD["former_site_pos"]]
        fg = [n for k, n, _ in path if k == "geom" and n in GOLD["former_geoms"]]
        rw = [tuple(w) for w in GOLD["removed_wraps"] if w[0] == name + "_tendon" and w[1] != "FCU_torus_wrap"]    # (that geom no longer exists)
        rw = list(dict.fromkeys(rw))
        lo, hi = ranges(base, name)
        rec = dict(shipped=dict(lo=lo, hi=hi, err_live=err(lo, hi, live), err_commented=err(lo, hi, old)), former_sites=fs, former_geoms=fg, removed_wraps=rw)
        late = [(name + "_tendon", n) for k, n, _ in path if k == "geom" and n.startswith("Elbow_PT_")]
        rec["late_wraps"] = late
        best = None
        site_sets = [()] + ([tuple(fs)] if fs else [])
        if 0 < len(fs) <= 6:
            site_sets = [c for k in range(len(fs) + 1) for c in itertools.combinations(fs, k)]
        for ss in site_sets:
            for gs in ([()] + ([tuple(fg)] if fg else [])):
                for ws, dr in itertools.product([c for k in range(len(rw) + 1) for c in itertools.combinations(rw, k)], [()] + ([tuple(late)] if late else [])):
                    if not (ss or gs or ws or dr):
                        continue
                    try:
                        cm = compile_variant(ss, gs, ws, dr)
                        lo2, hi2 = ranges(cm, name)
                    except Exception as e:      # a variant that does not compile (side site gone) is simply not a candidate
                        continue
                    for which, ref in (("live", live), ("commented", old)):
                        e = err(lo2, hi2, ref)
                        if best is None or e < best["err"]:
                            best = dict(err=e, reproduces=which, sites=list(ss), geoms=list(gs), wraps=[list(w) for w in ws], dropped=[list(w) for w in dr], lo=lo2, hi=hi2,
                                        all_former=(len(ss) == len(fs)))
        rec["best_restored"] = best
        out[name] = rec
        b = best or {}
        print(f"{name:7s} shipped err live {rec['shipped']['err_live']:.3f} commented {rec['shipped']['err_commented']:.3f} | restored: "
              f"{b.get('err', float('nan')):.3f} vs {b.get('reproduces')} sites={b.get('sites')} geoms={b.get('geoms')} wraps={b.get('wraps')} dropped={b.get('dropped')}", flush=True)
    if not only:
        with open(os.path.join(ROOT, "tests", "golden", "myohand_lengthrange_attribution.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
