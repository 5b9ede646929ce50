"""developer tool: where a kernel's spill traffic sits.  Reads device assembly built with -DMPC_ASM_MARK and counts, between consecutive phase markers (in layout order),
instructions / scratch loads / scratch stores / v_readlane+v_writelane (SGPR spills) / v_accvgpr moves.  usage: asm_spill_map.py file.s"""
import re, sys, collections
cur = "(prologue)"; order = []; cnt = collections.OrderedDict()
for line in open(sys.argv[1]):
    s = line.strip()
    m = re.match(r";\s*([A-Z_0-9]+_(BEGIN|END))\s*$", s)
    if m:
        cur = m.group(1)
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        continue
    c = cnt.setdefault(cur, [0, 0, 0, 0, 0])
    c[0] += 1
    op = s.split()[0]
    if op.startswith("scratch_load"): c[1] += 1
    elif op.startswith("scratch_store"): c[2] += 1
    elif op in ("v_readlane_b32", "v_writelane_b32"): c[3] += 1
    elif op.startswith("v_accvgpr"): c[4] += 1
print(f"{'after marker':22s} {'insts':>7s} {'sld':>6s} {'sst':>6s} {'lane':>6s} {'acc':>6s}")
for k, c in cnt.items():
    print(f"{k:22s} {c[0]:7d} {c[1]:6d} {c[2]:6d} {c[3]:6d} {c[4]:6d}")
t = [sum(c[i] for c in cnt.values()) for i in range(5)]
print(f"{'total':22s} {t[0]:7d} {t[1]:6d} {t[2]:6d} {t[3]:6d} {t[4]:6d}")
