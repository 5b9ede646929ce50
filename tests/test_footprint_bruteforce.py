"""An INDEPENDENT check of the teb footprint / obstacle geometry: teb's distance functions are not in
/root/reference, oracle/se2_nlp.py::footprint_distance restates them from their published semantics and oracle/mpc_oracle.c (which the
device mirrors line by line) restates them a second time -- but both are closed-form point / segment arithmetic written by the same hand.
Here the distance is measured with no formula at all: both OUTLINES (footprint placed at the pose; obstacle) are sampled densely and the
smallest pairwise point distance is taken.  That number is >= the true boundary-to-boundary distance and exceeds it by at most the sample
spacing, for every shape pairing, with crossing edges (-> ~0) and with one shape inside the other (teb has no inside test: it is the distance
between the outlines) -- exactly the semantics claimed for `footprint_distance`.  Circles enter as their centre with the radius subtracted
afterwards (teb: `distance - radius`), which is the one thing no sampling can contradict.

No code is shared with `footprint_distance` / `_dist_*`: only numpy broadcasting below."""
import numpy as np
import pytest

from oracle import se2_nlp as R

H = 2.5e-3          # sample spacing along every edge
POLY = (0.25, -0.05, 0.18, -0.05, 0.18, -0.18, -0.19, -0.18, -0.25, 0.0, -0.19, 0.18, 0.18, 0.18, 0.18, 0.05, 0.25, 0.05)
FOOTPRINTS = {
    "point": (R.FOOTPRINT_POINT, ()),
    "circle": (R.FOOTPRINT_CIRCLE, (0.22,)),
    "line": (R.FOOTPRINT_LINE, (0.0, 0.0, 0.4, 0.0)),
    "two_circles": (R.FOOTPRINT_TWO_CIRCLES, (0.2, 0.15, 0.2, 0.12)),
    "polygon": (R.FOOTPRINT_POLYGON, POLY),
}


def _sample_loop(verts, closed):
    """dense points on the edges of a vertex list (1 vertex: the point; 2: one edge; closed: edge loop)"""
    v = np.asarray(verts, float).reshape(-1, 2)
    if len(v) == 1:
        return v.copy()
    ne = len(v) if (closed and len(v) > 2) else len(v) - 1
    pts = []
    for e in range(ne):
        a, b = v[e], v[(e + 1) % len(v)]
        m = max(2, int(np.ceil(np.linalg.norm(b - a) / H)) + 1)
        t = np.linspace(0.0, 1.0, m)[:, None]
        pts.append(a + t * (b - a))
    return np.concatenate(pts)


def _min_pair(A, B):
    best = np.inf
    for i in range(0, len(A), 512):
        d = A[i:i + 512, None, :] - B[None, :, :]
        best = min(best, float(np.sqrt((d * d).sum(-1).min())))
    return best


def _brute(kind, params, pose, okind, overts, orad):
    """sampled-outline distance with the radii of circular shapes subtracted afterwards; returns a list of (value, footprint part) whose minimum
    is the footprint's distance (two circles: the closer of the two)"""
    c, s = np.cos(pose[2]), np.sin(pose[2])
    Rm = np.array([[c, -s], [s, c]])
    place = lambda p: pose[:2] + np.asarray(p, float).reshape(-1, 2) @ Rm.T
    O = _sample_loop(overts, closed=(okind == R.OBST_POLYGON))
    orad = orad if okind == R.OBST_CIRCLE else 0.0
    if kind == R.FOOTPRINT_POINT:
        parts = [(place([0.0, 0.0]), 0.0)]
    elif kind == R.FOOTPRINT_CIRCLE:
        parts = [(place([0.0, 0.0]), params[0])]
    elif kind == R.FOOTPRINT_LINE:
        parts = [(_sample_loop(place(params), closed=False), 0.0)]
    elif kind == R.FOOTPRINT_TWO_CIRCLES:
        fo, fr, ro, rr = params
        parts = [(place([fo, 0.0]), fr), (place([-ro, 0.0]), rr)]
    else:
        parts = [(_sample_loop(place(params), closed=True), 0.0)]
    return min(_min_pair(F, O) - rad - orad for F, rad in parts)


def _random_obstacle(rng, okind):
    c = rng.uniform(-1.2, 1.2, 2)
    if okind == R.OBST_POINT:
        return c[None, :], 0.0
    if okind == R.OBST_CIRCLE:
        return c[None, :], float(rng.uniform(0.05, 0.3))
    if okind == R.OBST_LINE:
        return np.stack([c, c + rng.uniform(-0.8, 0.8, 2)]), 0.0
    k = int(rng.integers(3, 7))
    ang = np.sort(rng.uniform(0, 2 * np.pi, k))
    return c + rng.uniform(0.2, 0.6) * np.stack([np.cos(ang), np.sin(ang)], 1), 0.0


@pytest.mark.parametrize("name", sorted(FOOTPRINTS))
def test_reference_form_distance_agrees_with_sampled_outlines(name, c_oracle):
    """>= 500 random poses per footprint over the four obstacle kinds: footprint_distance within [brute - H, brute]; and the C oracle's row
    (the code the device mirrors) gives the same number as footprint_distance for the footprints it evaluates through `footprint_row`."""
    kind, params = FOOTPRINTS[name]
    rng = np.random.default_rng(20260925 + kind)
    cfg = R.config_carlike_min_time(20)
    cfg.footprint_kind, cfg.footprint_params = kind, params
    if kind == R.FOOTPRINT_CIRCLE:
        cfg.footprint_radius = params[0]
    ob = c_oracle.obst_from_nlp_config(cfg, 1, 6, 4)
    n = crossing = inside = 0
    worst = 0.0
    for okind in (R.OBST_POINT, R.OBST_CIRCLE, R.OBST_LINE, R.OBST_POLYGON):
        for _ in range(26):
            verts, rad = _random_obstacle(rng, okind)
            o = R.Obstacle(okind, verts, rad)
            for _ in range(5):
                pose = np.array([*rng.uniform(-1.5, 1.5, 2), rng.uniform(-np.pi, np.pi)])
                d = R.footprint_distance(kind, params, pose, o)
                b = _brute(kind, params, pose, okind, verts, rad)
                assert b - H - 1e-12 <= d <= b + 1e-12, (name, okind, pose, d, b)
                worst = max(worst, b - d)
                dc = c_oracle.footprint_row(ob, pose, verts, rad)[0]
                assert abs(dc - d) < 1e-12, (name, okind, dc, d)
                n += 1
                crossing += d == 0.0
    # the cases that decide what "teb semantics" means, placed by hand
    sq = np.array([[-0.6, -0.6], [0.6, -0.6], [0.6, 0.6], [-0.6, 0.6]])
    for pose in (np.array([0.0, 0.0, 0.3]), np.array([0.1, -0.05, 2.0])):          # the whole footprint INSIDE a polygon: distance between the outlines, not 0
        o = R.Obstacle(R.OBST_POLYGON, sq, 0.0)
        d = R.footprint_distance(kind, params, pose, o)
        b = _brute(kind, params, pose, R.OBST_POLYGON, sq, 0.0)
        assert d > 0.05 and b - H - 1e-12 <= d <= b + 1e-12, (name, "inside", d, b)
        inside += 1
        n += 1
    if kind in (R.FOOTPRINT_LINE, R.FOOTPRINT_POLYGON):                              # an obstacle edge through the footprint: crossing edges -> 0
        for okind, verts in ((R.OBST_LINE, np.array([[0.1, -0.5], [0.12, 0.5]])), (R.OBST_POLYGON, np.array([[0.1, -0.5], [0.9, -0.5], [0.12, 0.5]]))):
            pose = np.array([0.0, 0.0, 0.0])
            d = R.footprint_distance(kind, params, pose, R.Obstacle(okind, verts, 0.0))
            b = _brute(kind, params, pose, okind, verts, 0.0)
            assert d == 0.0 and b <= H, (name, "crossing", d, b)
            crossing += 1
            n += 1
    assert n >= 500
    print(f"{name}: {n} poses, footprint_distance within [sampled - {H}, sampled] everywhere (largest gap {worst:.1e}), {crossing} crossing, {inside} inside")
