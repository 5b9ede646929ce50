#!/usr/bin/env python
"""Generates mpc_local_planner_amd/csrc/mpc_dpp_blocks.inc: the inline-asm blocks of the register-resident Riccati sweep
and of the forward recurrence (DP-ALU DPP broadcasts, v_fmac_f64_dpp ... row_newbcast:n).

Why blocks: the compiler cannot see the DPP operand inside inline asm, so it does not insert the wait states that the
'VALU write -> DPP read' hazard needs (2).  Every block therefore starts with `s_nop 1` (any copy the register allocator
places in front of the block is covered), and inside a block no DPP source is written less than 3 instructions earlier.
usage: python scripts/gen_dpp_blocks.py > mpc_local_planner_amd/csrc/mpc_dpp_blocks.inc"""

DPP = " row_mask:0xf bank_mask:0xf"


def fmac(sfx, acc, src, own, lane):
    return f"v_fmac_{sfx}_dpp %[{acc}], %[{src}], %[{own}] row_newbcast:{lane}{DPP}"


def emit(name, lines, outs, ins, sfx):
    body = "\\n\\t".join(lines)
    o = ", ".join(f'[{n}] "{c}"({e})' for n, c, e in outs)
    i = ", ".join(f'[{n}] "v"({e})' for n, e in ins)
    return f'    asm("{body}"\n        : {o}\n        : {i});\n'


def block_t1(sfx):
    L = ["s_nop 1"]
    for i in range(6):
        L.append(f"v_mul_{sfx} %[t{i}], %[v{i}], %[ec]")
    for lane, coef in ((0, "g0"), (1, "g1"), (2, "g2"), (3, "e3"), (4, "e4")):
        for i in range(6):
            L.append(fmac(sfx, f"t{i}", f"v{i}", coef, lane))
    for m in range(3):
        L.append(fmac(sfx, "om", f"g{m}", f"v{m}", 8))
    outs = [(f"t{i}", "=&v", f"t[{i}]") for i in range(6)] + [("om", "+&v", "om")]
    ins = [(f"v{i}", f"V[{i}]") for i in range(6)] + [("g0", "G[0]"), ("g1", "G[1]"), ("g2", "G[2]"), ("ec", "ec"), ("e3", "E3"), ("e4", "E4")]
    return emit("t1", L, outs, ins, sfx)


def block_h(sfx):
    L = ["s_nop 1"]
    for m in range(3):
        for row, lane in (("h6", 6), ("h7", 7), ("h5", 5), ("h2", 2)):
            if row == "h2" and m == 2:
                continue
            L.append(fmac(sfx, row, f"g{m}", f"t{m}", lane))
    outs = [("h2", "+&v", "h[2]"), ("h5", "+&v", "h[5]"), ("h6", "+&v", "h[6]"), ("h7", "+&v", "h[7]")]
    ins = [("g0", "G[0]"), ("g1", "G[1]"), ("g2", "G[2]"), ("t0", "t[0]"), ("t1", "t[1]"), ("t2", "t[2]")]
    return emit("h", L, outs, ins, sfx)


def block_r(sfx):
    mov = "v_mov_b64_dpp" if sfx == "f64" else "v_mov_b32_dpp"
    L = ["s_nop 1", f"{mov} %[r00], %[h6] row_newbcast:6{DPP}", f"{mov} %[r01], %[h6] row_newbcast:7{DPP}", f"{mov} %[r11], %[h7] row_newbcast:7{DPP}"]
    outs = [("r00", "=&v", "R00"), ("r01", "=&v", "R01"), ("r11", "=&v", "R11")]
    ins = [("h6", "h[6]"), ("h7", "h[7]")]
    return emit("r", L, outs, ins, sfx)


def block_v(sfx):
    L = ["s_nop 1"]
    for src, k in (("h6", "k0"), ("h7", "k1")):
        for a in range(3):
            L.append(fmac(sfx, f"w{a}", src, k, 9 + a))
    for src, k in (("h6", "k0"), ("h7", "k1")):
        for i in range(6):
            L.append(fmac(sfx, f"v{i}", src, k, i))
    outs = [(f"v{i}", "+&v", f"V[{i}]") for i in range(6)] + [(f"w{a}", "+&v", f"wn[{a}]") for a in range(3)]
    ins = [("h6", "h[6]"), ("h7", "h[7]"), ("k0", "nK0"), ("k1", "nK1")]
    return emit("v", L, outs, ins, sfx)


def block_fwd(sfx):
    # s = q0 + q1 xi[0] + q3 xi[2] + q5 xi[4];  s2 = q2 xi[1] + q4 xi[3];  s += s2;  xn = s + q6 s[3] + q7 s[4]
    mov = "v_mov_b64" if sfx == "f64" else "v_mov_b32"
    L = ["s_nop 1",
         fmac(sfx, "s", "xi", "q1", 0), fmac(sfx, "s2", "xi", "q2", 1), fmac(sfx, "s", "xi", "q3", 2),
         fmac(sfx, "s2", "xi", "q4", 3), fmac(sfx, "s", "xi", "q5", 4),
         f"v_add_{sfx} %[s], %[s], %[s2]",
         "s_nop 1",
         f"{mov} %[xn], %[s]",
         fmac(sfx, "xn", "s", "q6", 3), fmac(sfx, "xn", "s", "q7", 4)]
    outs = [("s", "+&v", "s"), ("s2", "+&v", "s2"), ("xn", "=&v", "xn")]
    ins = [("xi", "xi")] + [(f"q{j}", f"q[{j}]") for j in range(1, 8)]
    return emit("fwd", L, outs, ins, sfx)



# ---------------------------------------------------------------------------------------------------------------------------------------
# Partitioned ("parallel-in-time") sweep: the four 16-lane DPP rows work on four time segments of the horizon (mpc_wave.hpp::backward_pit).
# Lane sets that hold the six columns of a 6 x 6 block: LA (the sweep's own P columns) and LB; a combine step reads its value function from
# one set and leaves the next one in the other, so there are two variants of every block that touches them.
LA = [0, 1, 2, 3, 4, 5]
LB = [6, 7, 12, 13, 14, 15]


def block_v5(sfx):
    """block_v for a segment sweep: FIVE border columns (lanes 9..13: the costate of (x, u_prev) at the segment's end; lane 14 carries the trivial
    dt column) and omega accumulated straight into `om` (lanes 9 + a):  om[a] += nkappa_j Su[j][a]  =  k_j (lane 8) * h_{6+j} (own)"""
    L = ["s_nop 1", fmac(sfx, "om", "k0", "h6", 8), fmac(sfx, "om", "k1", "h7", 8)]
    for src, k in (("h6", "k0"), ("h7", "k1")):
        for a in range(5):
            L.append(fmac(sfx, f"w{a}", src, k, 9 + a))
    for src, k in (("h6", "k0"), ("h7", "k1")):
        for i in range(6):
            L.append(fmac(sfx, f"v{i}", src, k, i))
    outs = [(f"v{i}", "+&v", f"V[{i}]") for i in range(6)] + [(f"w{a}", "+&v", f"wn[{a}]") for a in range(5)] + [("om", "+&v", "om")]
    ins = [("h6", "h[6]"), ("h7", "h[7]"), ("k0", "nK0"), ("k1", "nK1")]
    return emit("v5", L, outs, ins, sfx)


def block_cwv(sfx):
    """WV = W [P+ | p+ | S+]: W (5 x 5, rows 0..4) as the LEFT factor, its column m in lane m of wt{i}; the value function's columns stay where they are"""
    mov = "v_mov_b64" if sfx == "f64" else "v_mov_b32"
    L = ["s_nop 1"] + [f"{mov} %[x{i}], 0" for i in range(5)]
    for m in range(5):
        for i in range(5):
            L.append(fmac(sfx, f"x{i}", f"wt{i}", f"vp{m}", m))
    outs = [(f"x{i}", "=&v", f"WV[{i}]") for i in range(5)]
    ins = [(f"wt{i}", f"Wt[{i}]") for i in range(5)] + [(f"vp{m}", f"Vp[{m}]") for m in range(5)]
    return emit("cwv", L, outs, ins, sfx)


def block_cgj(sfx, Lp):
    """Gauss-Jordan without pivoting on the rows a0..a5 of [M | S' | y | Z] (M = I - W P+ in the lanes Lp, its last row is e_5 exactly); the smallest
    |pivot| goes to `wp` (the caller falls back to the serial sweep when it is tiny)"""
    movd = "v_mov_b64_dpp" if sfx == "f64" else "v_mov_b32_dpp"
    L = ["s_nop 1"]
    for j in range(5):
        L.append(f"{movd} %[pv], %[a{j}] row_newbcast:{Lp[j]}{DPP}")
        L.append(f"v_min_{sfx} %[wp], %[wp], |%[pv]|")
        L.append(f"v_rcp_{sfx} %[r], %[pv]")
        L.append("s_nop 1")
        for _ in range(2 if sfx == "f64" else 1):
            L.append(f"v_fma_{sfx} %[e], -%[pv], %[r], 1.0")
            L.append(f"v_fma_{sfx} %[r], %[e], %[r], %[r]")
        L.append(f"v_mul_{sfx} %[na], -%[a{j}], %[r]")
        L.append(f"v_mul_{sfx} %[a{j}], %[a{j}], %[r]")
        order = [i for i in ([j + 1] + [q for q in range(5) if q not in (j, j + 1)]) if i < 5 and i != j]      # the next pivot row first: its DPP read is then >= 3 instructions away
        for i in order:
            L.append(fmac(sfx, f"a{i}", f"a{i}", "na", Lp[j]))
    L.append(f"v_mul_{sfx} %[na], %[a5], -1.0")
    L.append("s_nop 0")
    for i in range(5):
        L.append(fmac(sfx, f"a{i}", f"a{i}", "na", Lp[5]))
    outs = [(f"a{i}", "+&v", f"A[{i}]") for i in range(6)] + [("wp", "+&v", "wpiv"), ("pv", "=&v", "gj_pv"), ("r", "=&v", "gj_r"), ("e", "=&v", "gj_e"), ("na", "=&v", "gj_na")]
    return emit("cgj", L, outs, [], sfx)


def block_cu(sfx, Lp):
    """U += P+ [X | y | Z]: P+ as the left factor, its column m in lane Lp[m] of vp{i}"""
    L = ["s_nop 1"]
    for m in range(6):
        for i in range(6):
            L.append(fmac(sfx, f"u{i}", f"vp{i}", f"a{m}", Lp[m]))
    outs = [(f"u{i}", "+&v", f"U[{i}]") for i in range(6)]
    ins = [(f"vp{i}", f"Vp[{i}]") for i in range(6)] + [(f"a{m}", f"A[{m}]") for m in range(6)]
    return emit("cu", L, outs, ins, sfx)


def block_cra(sfx):
    """Ra += S U: S (6 x 6, its last column is e_5) as the left factor, column m in lane m of st{i}"""
    L = ["s_nop 1"]
    for m in range(6):
        for i in range(6):
            L.append(fmac(sfx, f"r{i}", f"st{i}", f"u{m}", m))
    outs = [(f"r{i}", "+&v", f"Ra[{i}]") for i in range(6)]
    ins = [(f"st{i}", f"St[{i}]") for i in range(6)] + [(f"u{m}", f"U[{m}]") for m in range(6)]
    return emit("cra", L, outs, ins, sfx)


def block_cwn(sfx):
    """W+[a][b] += S+[:, a]' Z[:, b] (lanes 9 + b of wp{a}), om+[a] += S+[:, a]' y (lanes 9 + a): the goal border of the value function"""
    L = ["s_nop 1"]
    for m in range(6):
        L.append(fmac(sfx, "om", f"a{m}", f"vp{m}", 8))
    for a in range(3):
        for m in range(6):
            L.append(fmac(sfx, f"wp{a}", f"vp{m}", f"a{m}", 9 + a))
    outs = [(f"wp{a}", "+&v", f"Wp[{a}]") for a in range(3)] + [("om", "+&v", "omp")]
    ins = [(f"vp{m}", f"Vp[{m}]") for m in range(6)] + [(f"a{m}", f"A[{m}]") for m in range(6)]
    return emit("cwn", L, outs, ins, sfx)


def block_bx(sfx, Lc, first):
    """boundary matrix-vector product with wave-uniform coefficients: acc_i = M_i[lane 8] + sum_j M_i[Lc[j]] xi_j + M_i[Lc[5]] dd + sum_a M_i[9 + a] nu_a, i < 5
    (M = the post-elimination tile [X | y | Z] for the boundary state, [P+ | p+ | S+] for the boundary costate); first: xi = 0"""
    movd = "v_mov_b64_dpp" if sfx == "f64" else "v_mov_b32_dpp"
    L = ["s_nop 1"] + [f"{movd} %[c{i}], %[m{i}] row_newbcast:8{DPP}" for i in range(5)]
    terms = ([] if first else [(Lc[j], f"xi{j}") for j in range(5)]) + [(Lc[5], "dd")] + [(9 + a, f"nu{a}") for a in range(3)]
    for lane, coef in terms:
        for i in range(5):
            L.append(fmac(sfx, f"c{i}", f"m{i}", coef, lane))
    outs = [(f"c{i}", "=&v", f"acc[{i}]") for i in range(5)]
    ins = [(f"m{i}", f"M[{i}]") for i in range(5)] + ([] if first else [(f"xi{j}", f"xi[{j}]") for j in range(5)]) + [("dd", "dd")] + [(f"nu{a}", f"nu[{a}]") for a in range(3)]
    return emit("bx", L, outs, ins, sfx)


print("// GENERATED by scripts/gen_dpp_blocks.py -- do not edit.  Inline-asm DPP blocks (see the generator for the hazard rules).")
for nm, fn, sig in (("MPC_DPP_BLOCK_T1", block_t1, ""), ("MPC_DPP_BLOCK_H", block_h, ""), ("MPC_DPP_BLOCK_R", block_r, ""),
                    ("MPC_DPP_BLOCK_V", block_v, ""), ("MPC_DPP_BLOCK_FWD", block_fwd, ""),
                    ("MPC_DPP_BLOCK_V5", block_v5, ""), ("MPC_DPP_BLOCK_CWV", block_cwv, ""),
                    ("MPC_DPP_BLOCK_CGJ_A", lambda x: block_cgj(x, LA), ""), ("MPC_DPP_BLOCK_CGJ_B", lambda x: block_cgj(x, LB), ""),
                    ("MPC_DPP_BLOCK_CU_A", lambda x: block_cu(x, LA), ""), ("MPC_DPP_BLOCK_CU_B", lambda x: block_cu(x, LB), ""),
                    ("MPC_DPP_BLOCK_CRA", block_cra, ""), ("MPC_DPP_BLOCK_CWN", block_cwn, ""),
                    ("MPC_DPP_BLOCK_BX_A", lambda x: block_bx(x, LA, False), ""), ("MPC_DPP_BLOCK_BX_B", lambda x: block_bx(x, LB, False), ""),
                    ("MPC_DPP_BLOCK_BX0_A", lambda x: block_bx(x, LA, True), ""), ("MPC_DPP_BLOCK_BX0_B", lambda x: block_bx(x, LB, True), "")):
    txt = "if constexpr (sizeof(T) == 8) {\n" + fn("f64") + "} else {\n" + fn("f32") + "}"
    print(f"#define {nm} \\")
    print(" \\\n".join(txt.split("\n")))
    print()
