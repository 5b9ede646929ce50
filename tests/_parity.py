"""Parity ACCOUNTING shared by the -m gpu tests (no thresholds on fractions)."""
import numpy as np


def account(tag, ocfg, inputs, r, oracle_out, tol=1e-4, obstacles=None, max_rows=None, kkt_tol=1e-6, min_match=0.7, start_x=None):
    """Every converged device instance is either within `tol` of the C oracle's result (same KKT point: 'match') or is shown to be a
    KKT point of the reference-form NLP on its own (oracle/kkt_check.py: feasibility, stationarity and complementarity <= kkt_tol;
    'other_kkt': a line-search tie or a regularisation decision flipped and the iterate sequences parted ways, or another candidate
    initial trajectory won).  Prints the counts and the objective differences, asserts that nothing is left unclassified.
    A floor on the matching share keeps the test meaningful (a device that never agreed with the oracle but only produced KKT points would pass the
    classification alone): of the instances that converged on BOTH sides at least `min_match` must match.
    Returns (match mask, list of other_kkt indices)."""
    from oracle import kkt_check as KC
    x0, xf, up, dtp = inputs
    xo, uo, do, st, it = oracle_out[:5]
    B = x0.shape[0]
    err = np.maximum(np.abs(r.x - xo).reshape(B, -1).max(1), np.abs(r.u - uo).reshape(B, -1).max(1))
    err = np.maximum(err, np.abs(r.dt - do))
    conv = r.status == 0
    match = conv & (st == 0) & (err < tol)
    rest = np.nonzero(conv & ~match)[0]
    # start_x: the state trajectories the answers' solves started from where that is not the reference's cold start (results of candidate initial trajectories with
    # clearance rows: the rows of a solve are associated on the trajectory it starts from, in the reference as in the product)
    res = KC.kkt_many(ocfg, x0, xf, up, dtp, r.x, r.u, r.dt, rest, obstacles=obstacles, max_rows=max_rows, start_x=start_x)
    good = lambda i: KC.is_kkt_point(res[i], kkt_tol, kkt_tol, kkt_tol)
    other = [i for i in rest if good(i)]
    bad = [i for i in rest if not good(i)]
    dobj = [res[i]["objective"] - (ocfg.n - 1) * do[i] for i in other if st[i] == 0] if ocfg.objective == 0 else []
    print(f"[{tag}] B={B}: device converged {int(conv.sum())}, oracle converged {int((st == 0).sum())}, same status {float((r.status == st).mean()):.4f}; "
          f"match(<{tol:g}) {int(match.sum())} (median |d| {np.median(err[match]) if match.any() else float('nan'):.1e}), other_kkt {len(other)}, unclassified {len(bad)}; "
          f"objective(device) - objective(oracle) over other_kkt with a converged oracle: n={len(dobj)}"
          + (f", min {min(dobj):+.3e}, median {np.median(dobj):+.3e}, max {max(dobj):+.3e}" if dobj else "")
          + f"; worst other_kkt feas/stat/comp = {max([res[i]['feas'] for i in other], default=0):.1e}/{max([res[i]['stat'] for i in other], default=0):.1e}/{max([res[i]['comp'] for i in other], default=0):.1e}")
    assert not bad, [(int(i), res[i]) for i in bad[:5]]
    both = int((conv & (st == 0)).sum())
    assert both == 0 or match.sum() >= min_match * both, (tag, int(match.sum()), both, min_match)
    return match, other
