"""Static audit of VGPR spills against the EXEC mask they are stored / reloaded under (no GPU needed).

    hipcc ... --save-temps  ->  *.s ;  python tools/isa_spill_audit.py file.s [kernel-name-substring] [-v]

A VGPR spill is `scratch_store_dword ... ; N-byte Folded Spill`; it writes only the lanes enabled in EXEC at that point.  The hazard looked
for ("Diagnostic builds"): a value DEFINED under a wide mask, STORED to its spill slot under a narrower one (inside a divergent
region) and RELOADED outside that region: the lanes that were off at the store come back as whatever the scratch slot held before --
residue of earlier kernels, hence order-dependent results.

The compiler's structurised control flow narrows EXEC with `s_and_saveexec_b64 sN, cond` (if), flips it with `s_andn2_saveexec_b64 sN, sN` /
`s_or_saveexec_b64` (else) and widens it again with `s_or_b64 exec, exec, sN` (end of the region); `s_xor_b64 sM, exec, sN` right after the
save moves the saved mask to sM.  Along the linear listing the stack of open save registers therefore describes how narrow EXEC is: a
position whose stack is a proper extension of another position's stack runs under a subset of that position's lanes.  For every spill slot
the script lists, per store, the stack at the store and at the (textually) preceding definition of the stored register, and per reload the
stack at the reload, and flags the triples  def-stack < store-stack  (stored under a narrower mask than defined)  with a reload whose stack
does not extend the store's region (= runs with lanes the store never wrote).  Back edges can make the textual "preceding definition" the
wrong one; flagged triples are printed with line numbers for reading, not decided."""
import re
import sys


def functions(path):
    name, body = None, []
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            m = re.match(r"^(_Z\w+):", line)
            if m and name is None:
                name, body = m.group(1), []
            elif name is not None:
                if line.startswith(".Lfunc_end"):
                    yield name, body
                    name = None
                else:
                    body.append((ln, line.rstrip("\n")))


SAVE = re.compile(r"\s+s_(and|andn2|or|xor)_saveexec_b64\s+(s\[\d+:\d+\]|vcc),\s*(s\[\d+:\d+\]|vcc|exec|-?\d+)")
MOVE = re.compile(r"\s+s_xor_b64\s+(s\[\d+:\d+\]),\s*exec,\s*(s\[\d+:\d+\])")
RESTORE = re.compile(r"\s+s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\]|vcc)")
SPILL = re.compile(r"\s+scratch_store_dword(?:x(\d))?\s+off,\s*v\[?(\d+)(?::\d+)?\]?,\s*off(?:\s+offset:(\d+))?\s*;\s*(\d+)-byte Folded Spill")
RELOAD = re.compile(r"\s+scratch_load_dword(?:x(\d))?\s+v\[?(\d+)(?::\d+)?\]?,\s*off,\s*off(?:\s+offset:(\d+))?\s*;\s*(\d+)-byte Folded Reload")
VDEF = re.compile(r"\s+(?:v_\w+|ds_read\w*|ds_bpermute\w*|ds_permute\w*|ds_swizzle\w*|global_load\w*|global_atomic\w*|flat_load\w*|buffer_load\w*|scratch_load\w*)\s+v\[?(\d+)(?::(\d+))?\]?")


def audit(name, body, verbose=False):
    stack = []              # (register, line of the save) of the open regions, innermost last
    vdef = {}               # vgpr -> (line, stack ids) of its textually last definition
    slots = {}
    for ln, line in body:
        m = SAVE.match(line)
        if m:
            kind, reg, src = m.group(1), m.group(2), m.group(3)
            if kind in ("andn2", "or") and src == reg and any(r == reg for r, _ in stack):
                pass        # else-flip of an open region: same region, complementary lanes; stays open until the s_or_b64
            else:
                stack.append((reg, ln))
            continue
        m = MOVE.match(line)
        if m and stack and stack[-1][0] == m.group(2):
            stack[-1] = (m.group(1), stack[-1][1])
            continue
        m = RESTORE.match(line)
        if m:
            for i in range(len(stack) - 1, -1, -1):
                if stack[i][0] == m.group(1):
                    del stack[i:]
                    break
            continue
        ids = tuple(l for _, l in stack)
        m = SPILL.match(line)
        if m:
            n, v0, off, size = int(m.group(1) or 1), int(m.group(2)), int(m.group(3) or 0), int(m.group(4))
            for k in range(size // 4):
                slots.setdefault(off + 4 * k, []).append(("S", ln, ids, vdef.get(v0 + k, (0, ()))))
            continue
        m = RELOAD.match(line)
        if m:
            v0, off, size = int(m.group(2)), int(m.group(3) or 0), int(m.group(4))
            for k in range(size // 4):
                slots.setdefault(off + 4 * k, []).append(("R", ln, ids, None))
                vdef[v0 + k] = (ln, ids)
            continue
        m = VDEF.match(line)
        if m:
            a = int(m.group(1)); b = int(m.group(2)) if m.group(2) else a
            for v in range(a, b + 1):
                vdef[v] = (ln, ids)

    def prefix(a, b):       # a is a (not necessarily proper) prefix of b: b runs under a subset of a's lanes
        return len(a) <= len(b) and b[:len(a)] == a

    hazards = []
    for off in sorted(slots):
        ev = slots[off]
        for s in (e for e in ev if e[0] == "S"):
            dline, dids = s[3]
            if not (prefix(dids, s[2]) and len(dids) < len(s[2])):
                continue     # stored under the mask it was defined under (or wider): every defined lane is in the slot
            for r in (e for e in ev if e[0] == "R"):
                if not prefix(s[2], r[2]):          # the reload is not inside the store's region: lanes outside the store mask are read
                    hazards.append((off, dline, dids, s[1], s[2], r[1], r[2]))
    ns = sum(1 for v in slots.values() for e in v if e[0] == "S")
    nr = sum(1 for v in slots.values() for e in v if e[0] == "R")
    print(f"{name[:118]}\n  spilled dwords {len(slots)}, store sites {ns}, reload sites {nr}; store-narrower-than-def with an outside reload: {len(hazards)}")
    if verbose:
        for off, dl, di, sl, si, rl, ri in hazards:
            print(f"    slot {off:4d}: def @{dl} depth {len(di)} | store @{sl} depth {len(si)} (regions opened at {list(si[len(di):])}) | reload @{rl} depth {len(ri)}")
    return len(slots), hazards


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "-v"]
    sub = args[1] if len(args) > 1 else "step_kernel_w"
    for name, body in functions(args[0]):
        if sub in name:
            audit(name, body, "-v" in sys.argv)
