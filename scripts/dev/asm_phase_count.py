"""Static instruction counts between the `; XXX_BEGIN` / `; XXX_END` markers of a -DMPC_ASM_MARK=1 -S build of mpc_capi.hip
(usage: asm_phase_count.py file.s kernel-name-substring).  Counts by class: VALU (v_*), of which DPP, division helpers, transcendental; SALU; LDS; readlane / writelane;
waitcnt / nop.  Loops are counted once (static)."""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
# the function body: from the label line "<mangled>:" to its .amdhsa_kernel
start = end = None
for i, l in enumerate(lines):
    if l.endswith(":") and key in l and not l.startswith("\t") and start is None and ".L" not in l: start = i
    if ".amdhsa_kernel" in l and key in l: end = i; break
body = lines[start:end]
print(f"{key}: lines {start}..{end}")
stack = {}; res = collections.OrderedDict()
def cls(op):
    c = []
    if op.startswith("v_"):
        c.append("valu")
        if "readlane" in op or "writelane" in op or "readfirstlane" in op: c.append("lane_x")
        if op.startswith(("v_div_", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_ldexp", "v_frexp")): c.append("div/trans")
        if op.startswith("v_cmp"): c.append("cmp")
        if op.startswith("v_cndmask"): c.append("cndmask")
        if op.startswith(("v_accvgpr", "v_mov")): c.append("mov")
    elif op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier")): c.append("wait/nop")
        elif op.startswith(("s_cbranch", "s_branch")): c.append("branch")
        else: c.append("salu")
    elif op.startswith("ds_"): c.append("lds")
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c.append("vmem")
    else: c.append("other")
    return c
for l in body:
    t = l.strip()
    m = re.match(r";\s*([A-Z0-9_]+)_(BEGIN|END)$", t)
    if m:
        name, what = m.groups()
        if what == "BEGIN": stack[name] = collections.Counter()
        else:
            c = stack.pop(name, None)
            if c is not None: res.setdefault(name, []).append(c)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    op = t.split()[0]
    dpp = "row_" in t or "quad_perm" in t or "wave_" in t
    for c in stack.values():
        c["total"] += 1
        for k in cls(op): c[k] += 1
        if dpp: c["dpp"] += 1
for name, cs in res.items():
    for c in cs:
        print(f"{name:12s} " + "  ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
