"""Builds .ptn pattern text (the format brisk-descriptor-extractor.cc:180-291 parses: N, N x {x y sigma}, S, S x {i j},
L, L x {i j}) from the default pattern's data table (oracle/default_pattern.inc), optionally modified - test input for
the custom-pattern constructors."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def default_tables():
    txt = open(os.path.join(ROOT, "oracle", "default_pattern.inc")).read()

    def block(name):
        body = txt[txt.index(name):]
        body = body[body.index("{") + 1:body.index("};")]
        return [[float(v.rstrip("f")) for v in row.split(",")] for row in re.findall(r"\{([^{}]*)\}", body)]
    pts = np.array(block("brisk_default_points"), np.float64)
    sp = np.array(block("brisk_default_short_pairs"), np.int64)
    lp = np.array(block("brisk_default_long_pairs"), np.int64)
    assert pts.shape == (66, 3) and sp.shape == (384, 2) and lp.shape == (856, 2)
    return pts, sp, lp


def ptn_text(points, short_pairs, long_pairs):
    out = ["%d" % len(points)]
    out += ["%.9g %.9g %.9g" % (p[0], p[1], p[2]) for p in points]
    out.append("%d" % len(short_pairs))
    out += ["%d %d" % (a, b) for a, b in short_pairs]
    out.append("%d" % len(long_pairs))
    out += ["%d %d" % (a, b) for a, b in long_pairs]
    return "\n".join(out) + "\n"


def custom_pattern(seed=0, sigma_factor=1.0, drop_points=0):
    """A valid pattern that differs from the default one: jittered sample positions, scaled smoothing radii, optionally
    fewer points (the pairs that use a dropped point are re-pointed; the 384 short pairs the format requires stay)."""
    pts, sp, lp = default_tables()
    rng = np.random.default_rng(seed)
    pts = pts.copy()
    pts[:, :2] += rng.normal(0, 0.15, (len(pts), 2))
    pts[:, 2] *= sigma_factor
    n = len(pts) - drop_points
    pts = pts[:n]
    sp = np.where(sp >= n, sp % n, sp)
    lp = np.where(lp >= n, lp % n, lp)
    sp[sp[:, 0] == sp[:, 1], 1] = (sp[sp[:, 0] == sp[:, 1], 1] + 1) % n
    lp[lp[:, 0] == lp[:, 1], 1] = (lp[lp[:, 0] == lp[:, 1], 1] + 1) % n
    return ptn_text(pts, sp, lp)
