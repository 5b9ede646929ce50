"""The algebra of the partitioned ("parallel-in-time") Riccati sweep of csrc/mpc_wave.hpp, checked on random stage-structured LQ problems against a dense
solve of the same KKT system (no GPU needed).

The Newton system of an interior-point iteration is an LQ problem over the augmented stage state xi_k = (dx_k, du_{k-1}, ddt) in R^6 with control du_k in R^2:
    xi_{k+1} = F_k [xi_k; u_k] + c_k,   stage cost 1/2 z' H_k z + h_k' z  (z = [xi; u]),   terminal cost + terminal equality rows S_N' xi_N + om_N = 0 (multiplier nu).
The SERIAL sweep carries the value function V_k(xi, nu) = 1/2 xi' P xi + p' xi + nu' (S' xi + om) + 1/2 nu' W nu backwards over all N stages.
The PARTITIONED sweep cuts the horizon into 4 segments.  Over a segment [a, b) whose terminal value function is not known yet, the same recursion started from
(P, p, S, W, om) = (0, 0, I, 0, 0) yields the segment's SCATTERING element -- the map (xi_a, lam_b) -> (lam_a, xi_b):
    lam_a = P xi_a + S lam_b + p,        xi_b = S' xi_a + W lam_b + om
(the border "multiplier" of the segment is the costate at its end).  Elements are then folded right to left into value functions at the segment boundaries,
    M = I - W P+,  xi_b = X xi_a + y + Z nu  with  X = M^-1 S',  y = M^-1 (W p+ + om),  Z = M^-1 W S+,
    P_a = P + S P+ X,   p_a = p + S (p+ + P+ y),   S_a = S (S+ + P+ Z),   W_a = W+ + S+' Z,   om_a = om+ + S+' y,
the root solve at stage 0 gives (ddt, nu), the boundary states / costates follow from (X, y, Z) and (P+, p+, S+), and every segment runs its forward recurrence
on its own with the gains of its own backward sweep.  Here: both ways give the same step as numpy.linalg.solve on the assembled KKT matrix."""
import numpy as np
import pytest


def random_lq(rng, N, nfix=3, convex=True):
    """stage data in the structure of the kernel's LQ problem (mpc_core.hpp::riccati_step): F = [[A_x 0 f B], [0 0 0 I2], [0 0 1 0]]"""
    st = []
    for k in range(N):
        F = np.zeros((6, 8))
        F[:3, :3] = np.eye(3); F[0, 2], F[1, 2] = rng.normal(size=2) * 0.3
        F[:3, 5] = rng.normal(size=3) * 0.5
        F[:3, 6:8] = rng.normal(size=(3, 2)) * 0.4
        F[3, 6] = F[4, 7] = 1.0
        F[5, 5] = 1.0
        c = np.zeros(6); c[:3] = rng.normal(size=3) * 0.1
        L = rng.normal(size=(8, 8)) * 0.4
        H = L @ L.T * (1.0 if convex else 0.3) + np.diag(rng.uniform(0.05, 0.5, 8))
        if not convex:
            H[2, 2] -= 0.4
        h = rng.normal(size=8)
        st.append((F, c, H, h))
    PN = np.diag(rng.uniform(0.0, 1.0, 6)); pN = rng.normal(size=6) * 0.2
    SN = np.zeros((6, nfix));
    for i in range(nfix):
        SN[i, i] = 1.0; PN[i, i] = 0.0; pN[i] = 0.0
    WN = -1e-9 * np.eye(nfix); omN = rng.normal(size=nfix) * 0.1
    return st, (PN, pN, SN, WN, omN)


def dense_solve(st, term, want_kkt=False):
    """variables: xi_0[5] (= ddt; the other five components of xi_0 are fixed to 0), u_0, xi_1, u_1, ..., xi_N, then lam_1..lam_N (dynamics), nu"""
    N = len(st); PN, pN, SN, WN, omN = term; m = SN.shape[1]
    nz = 1 + 2 * N + 6 * N; ne = 6 * N + m
    ix = lambda k: slice(1 + 2 * N + 6 * (k - 1), 1 + 2 * N + 6 * k)          # xi_k, k >= 1
    iu = lambda k: slice(1 + 2 * k, 3 + 2 * k)
    K = np.zeros((nz + ne, nz + ne)); r = np.zeros(nz + ne)

    def zidx(k):        # indices of z_k = [xi_k; u_k] in the variable vector (xi_0: only component 5 is a variable)
        if k == 0:
            return [None] * 5 + [0] + list(range(iu(0).start, iu(0).stop))
        return list(range(ix(k).start, ix(k).stop)) + list(range(iu(k).start, iu(k).stop))
    for k, (F, c, H, h) in enumerate(st):
        zi = zidx(k)
        for a in range(8):
            if zi[a] is None: continue
            r[zi[a]] -= h[a]
            for b in range(8):
                if zi[b] is None: continue
                K[zi[a], zi[b]] += H[a, b]
        row = nz + 6 * k       # xi_{k+1} - F z_k - c = 0
        for i in range(6):
            K[row + i, ix(k + 1).start + i] = 1.0; K[ix(k + 1).start + i, row + i] = 1.0
            for b in range(8):
                if zi[b] is None: continue
                K[row + i, zi[b]] -= F[i, b]; K[zi[b], row + i] -= F[i, b]
            r[row + i] = c[i]
    xs = ix(N)
    K[xs, xs] += PN; r[xs] -= pN
    for j in range(m):
        K[nz + 6 * N + j, xs] = SN[:, j]; K[xs, nz + 6 * N + j] = SN[:, j]
        K[nz + 6 * N + j, nz + 6 * N + j] = WN[j, j]
        r[nz + 6 * N + j] = -omN[j]
    if want_kkt:
        return K, nz, ne
    sol = np.linalg.solve(K, r)
    xi = np.zeros((N + 1, 6)); xi[0, 5] = sol[0]
    for k in range(1, N + 1):
        xi[k] = sol[ix(k)]
    u = np.stack([sol[iu(k)] for k in range(N)])
    return xi, u, sol[nz + 6 * N:]


def backward(st, V):
    """serial Riccati over the stages `st` (in order), from the value function V = (P, p, S, W, om) at their end; returns V at their start and the gains"""
    P, p, S, W, om = [a.copy() for a in V]
    gains = []
    for (F, c, H, h) in reversed(st):
        Hh = H + F.T @ P @ F
        hh = h + F.T @ (p + P @ c)
        Sh = F.T @ S
        om = om + S.T @ c
        Ri = np.linalg.inv(Hh[6:, 6:])
        K, kap, Kn = Ri @ Hh[6:, :6], Ri @ hh[6:], Ri @ Sh[6:]
        P = Hh[:6, :6] - Hh[:6, 6:] @ K
        p = hh[:6] - Hh[:6, 6:] @ kap
        S = Sh[:6] - Hh[:6, 6:] @ Kn
        W = W - Sh[6:].T @ Kn
        om = om - Sh[6:].T @ kap
        gains.append((K, kap, Kn))
    return (P, p, S, W, om), gains[::-1]


def forward(st, gains, xi0, nu):
    xi, us = [xi0], []
    for (F, c, H, h), (K, kap, Kn) in zip(st, gains):
        u = -(K @ xi[-1] + kap + Kn @ nu)
        us.append(u)
        xi.append(F @ np.concatenate([xi[-1], u]) + c)
    return np.array(xi), np.array(us)


def root(V, m):
    """stage 0: xi_0 = (0, 0, 0, 0, 0, dd); stationarity in dd and the terminal rows"""
    P, p, S, W, om = V
    A = np.zeros((1 + m, 1 + m)); b = np.zeros(1 + m)
    A[0, 0] = P[5, 5]; A[0, 1:] = S[5]; b[0] = -p[5]
    A[1:, 0] = S[5]; A[1:, 1:] = W; b[1:] = -om
    s = np.linalg.solve(A, b)
    return s[0], s[1:]


def combine(E, Vp):
    """segment element E = (P, p, S, W, om) (6-dimensional border) with the value function Vp at its end -> value function at its start + (X, y, Z)"""
    P, p, S, W, om = E
    Pp, pp, Sp, Wp, omp = Vp
    M = np.eye(6) - W @ Pp
    X = np.linalg.solve(M, S.T); y = np.linalg.solve(M, W @ pp + om); Z = np.linalg.solve(M, W @ Sp)
    return (P + S @ Pp @ X, p + S @ (pp + Pp @ y), S @ (Sp + Pp @ Z), Wp + Sp.T @ Z, omp + Sp.T @ y), (X, y, Z)


@pytest.mark.parametrize("N,convex", [(49, True), (49, False), (12, True), (119, True), (30, False)])
def test_partitioned_sweep_equals_serial_sweep_equals_dense_solve(N, convex):
    rng = np.random.default_rng(7 + N)
    st, term = random_lq(rng, N, convex=convex)
    xi_d, u_d, nu_d = dense_solve(st, term)
    # serial
    V0, gains = backward(st, term)
    dd, nu = root(V0, 3)
    xi0 = np.zeros(6); xi0[5] = dd
    xi_s, u_s = forward(st, gains, xi0, nu)
    scale = max(1.0, np.abs(xi_d).max(), np.abs(u_d).max())
    assert np.abs(xi_s - xi_d).max() < 1e-8 * scale and np.abs(u_s - u_d).max() < 1e-8 * scale and np.abs(nu - nu_d).max() < 1e-7 * max(1.0, np.abs(nu_d).max())
    # partitioned: 4 segments, the last one ends in the terminal value function, the others in the identity border
    b = [0] + [round(N * s / 4) for s in (1, 2, 3)] + [N]
    I6 = (np.zeros((6, 6)), np.zeros(6), np.eye(6), np.zeros((6, 6)), np.zeros(6))
    seg = [backward(st[b[s]:b[s + 1]], term if s == 3 else I6) for s in range(4)]
    # the dt costate column of a mid-segment element is trivial: S[:, 5] = e_5, W[5, :] = W[:, 5] = 0 and its gains vanish (what lets the kernel carry 5 border columns)
    for s in range(3):
        (P, p, S, W, om), g = seg[s]
        assert np.allclose(S[:, 5], np.eye(6)[5]) and not W[5].any() and not W[:, 5].any() and all(not Kn[:, 5].any() for (_, _, Kn) in g)
    V = [None] * 4; maps = [None] * 3
    V[3] = seg[3][0]
    for s in (2, 1, 0):
        V[s], maps[s] = combine(seg[s][0], V[s + 1])
    for a, c in zip(V[0], V0):
        assert np.abs(a - c).max() < 1e-7 * max(1.0, np.abs(c).max())
    dd2, nu2 = root(V[0], 3)
    xb = [np.zeros(6)]; xb[0][5] = dd2
    for s in range(3):
        X, y, Z = maps[s]
        xb.append(X @ xb[-1] + y + Z @ nu2)
    lam = [None] + [V[s][0] @ xb[s] + V[s][1] + V[s][2] @ nu2 for s in (1, 2, 3)]          # costates at the boundaries b_1..b_3
    xi_p = np.zeros_like(xi_d); u_p = np.zeros_like(u_d)
    for s in range(4):
        xs, us = forward(st[b[s]:b[s + 1]], seg[s][1], xb[s], nu2 if s == 3 else lam[s + 1])
        xi_p[b[s]:b[s + 1] + 1] = xs; u_p[b[s]:b[s + 1]] = us
    assert np.abs(xi_p - xi_d).max() < 1e-7 * scale and np.abs(u_p - u_d).max() < 1e-7 * scale


# ---------------------------------------------------------------- inertia (r04)
def neg_pivots(A):
    """negative eigenvalues of a symmetric matrix by Jacobi's signature rule: the negative pivots of the elimination without exchanges (None when a pivot vanishes)"""
    A = np.array(A, float); n = A.shape[0]; neg = 0
    scale = np.abs(A).max()
    for c in range(n):
        pv = A[c, c]
        if not abs(pv) > 1e-13 * scale:
            return None
        neg += pv < 0
        A[c + 1:, c:] -= np.outer(A[c + 1:, c] / pv, A[c, c:])
    return neg


def backward_neg(st, V):
    """backward() + the negative eigenvalues of the control pivots R_k it eliminates"""
    P, p, S, W, om = [a.copy() for a in V]
    neg = 0
    for (F, c, H, h) in reversed(st):
        Hh = H + F.T @ P @ F
        Sh = F.T @ S
        R = Hh[6:, 6:]
        det = R[0, 0] * R[1, 1] - R[0, 1] ** 2
        neg += 1 if det < 0 else (2 if R[0, 0] < 0 else 0)
        Ri = np.linalg.inv(R)
        K, Kn = Ri @ Hh[6:, :6], Ri @ Sh[6:]
        P = Hh[:6, :6] - Hh[:6, 6:] @ K
        S = Sh[:6] - Hh[:6, 6:] @ Kn
        W = W - Sh[6:].T @ Kn
    return (P, p, S, W, om), neg


def root_neg(V, m):
    P, p, S, W, om = V
    A = np.zeros((1 + m, 1 + m))
    A[1:, 1:] = W; A[0, 0] = P[5, 5]; A[0, 1:] = S[5]; A[1:, 0] = S[5]
    order = list(range(1, 1 + m)) + [0]                      # nu first, then dt (mpc_core.hpp::riccati_root)
    return neg_pivots(A[np.ix_(order, order)])


@pytest.mark.parametrize("N,seed", [(49, 1), (49, 2), (49, 3), (30, 4), (119, 5), (12, 6), (49, 7), (49, 8)])
def test_inertia_of_the_kkt_matrix_from_the_sweeps(N, seed):
    """Ipopt accepts a factorisation when the KKT matrix has one negative eigenvalue per equality row.  SERIAL sweep: the count is the sum over the stages of the negative
    eigenvalues of the control pivots R_k plus those of the (dt, nu) root system (every eliminated pair (xi_{k+1}, lambda_k) contributes six positive and six negative
    ones whatever the curvature; Haynsworth's inertia additivity) -- what mpc_core.hpp::riccati_step / riccati_root count.  PARTITIONED sweep: a mid-segment's pivots are those of
    its own cost-to-go from the identity border, and every combine eliminates the pair (lambda_b, xi_b) at a boundary: the block [[W, -I], [-I, P+]] over the five components
    that have a costate column (the dt component passes through) has the inertia In(W) + In(P+ - W^-1).  Both counts equal the dense matrix's on random LQ problems, convex and
    not (where the count is wrong both say by how much)."""
    rng = np.random.default_rng(100 + seed)
    st, term = random_lq(rng, N, convex=False)
    # more or less curvature removed: from the right inertia to several negative directions
    shift = [0.0, 0.3, 0.8, 1.5, 0.0, 0.5, 2.5, 0.1][seed - 1]
    st = [(F, c, H - shift * np.diag([0, 0, 1, 0, 0, 0, 1, 1.0]), h) for (F, c, H, h) in st]
    K, nz, ne = dense_solve(st, term, want_kkt=True)
    ev = np.linalg.eigvalsh(K)
    dense_extra = int((ev < 0).sum()) - ne                  # negative eigenvalues beyond the one per equality row
    assert np.abs(ev).min() > 1e-10
    # serial
    V0, neg = backward_neg(st, term)
    serial_extra = neg + root_neg(V0, 3) - 3
    assert serial_extra == dense_extra
    # partitioned
    b = [0] + [round(N * s / 4) for s in (1, 2, 3)] + [N]
    I6 = (np.zeros((6, 6)), np.zeros(6), np.eye(6), np.zeros((6, 6)), np.zeros(6))
    seg = [backward_neg(st[b[s]:b[s + 1]], term if s == 3 else I6) for s in range(4)]
    total = sum(sg[1] for sg in seg)
    V = seg[3][0]
    for s in (2, 1, 0):
        W5, P5 = seg[s][0][3][:5, :5], V[0][:5, :5]
        nw, ng = neg_pivots(W5), neg_pivots(P5 - np.linalg.inv(W5))
        assert nw is not None and ng is not None
        total += nw + ng - 5
        V, _ = combine(seg[s][0], V)
    pit_extra = total + root_neg(V, 3) - 3
    assert pit_extra == dense_extra
    if seed == 1:
        assert dense_extra == 0
    if seed in (3, 4, 7):
        assert dense_extra > 0
