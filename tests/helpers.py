"""Shared helpers for the tests (fixtures loading, canonical hit ordering)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


def load_coins():
    img = np.load(os.path.join(GOLDEN_DIR, "coins.npz"))["image"]
    assert img.shape == (303, 384) and img.dtype == np.uint8 and int(img.sum()) == 11269333
    return img


def coin_templates(image):
    small = image[37:37 + 38, 80:80 + 41]
    big = image[14:14 + 59, 302:302 + 65]
    return small, big


def hits_json(hits):
    return [[h[0], [int(v) for v in h[1]], float(np.float32(h[2]))] for h in hits]


def canon(hits):
    """Order-free canonical form (the reference's cross-template / tie order is thread timing)."""
    return sorted(hits_json(hits), key=lambda h: (-h[2], h[0], h[1]))


def assert_hits_equal(got, expected, tol=1e-4, ordered=True):
    got = hits_json(got) if got and not isinstance(got[0], list) else got
    if not ordered:
        got = sorted(got, key=lambda h: (h[0], h[1]))
        expected = sorted(expected, key=lambda h: (h[0], h[1]))
    assert len(got) == len(expected), (len(got), len(expected))
    for g, e in zip(got, expected):
        assert g[0] == e[0] and list(g[1]) == list(e[1]), (g, e)
        assert abs(g[2] - e[2]) <= tol * max(1.0, abs(e[2])), (g, e)
