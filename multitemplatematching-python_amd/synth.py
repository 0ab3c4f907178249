"""
Seeded synthetic workloads for the BASELINE.json configs.

Pure numpy, no dependency on the product or the oracle: used by tests, bench.py and the golden
generator, and byte-for-byte reproducible across numpy versions because every pixel comes from a
counter-based splitmix64 hash of (seed, stream, index) instead of a library RNG.

Image: i.i.d. uniform uint8 background.  Templates: i.i.d. uniform uint8.  Every unit is planted
once exactly (score 1.0) and ``noisy_per_unit`` times with additive uniform noise of increasing
amplitude (normalised correlation roughly 0.95 / 0.87 / 0.78), each plant in its own grid cell so
that planted boxes never overlap.  Background NCC at 64x64 is ~N(0, 1/64): far below 0.5.

As in the reference tutorials (tutorials/Tutorial2-Template_Augmentation.ipynb:313) rotations and
scales are extra entries appended to ``listTemplates`` by the caller.
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wraps modulo 2^64)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream(seed, stream, n):
    base = splitmix64(np.array([(int(seed) << 32) ^ int(stream)], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + base
    return splitmix64(idx)


def rand_u8(seed, stream, shape):
    n = int(np.prod(shape))
    return (_stream(seed, stream, n) >> np.uint64(56)).astype(np.uint8).reshape(shape)


def rand_int(seed, stream, n, lo, hi):
    """n integers in [lo, hi)."""
    r = _stream(seed, stream, n) >> np.uint64(11)
    return (lo + (r % np.uint64(hi - lo)).astype(np.int64)).astype(np.int64)


def smooth_u8(seed, shape, scales=(4, 16, 64), weights=(1.0, 1.5, 2.0), noise=0.15):
    """uint8 image with the statistics of a photograph - neighbouring pixels correlated: box-filtered white noise at
    several scales plus a little sensor noise, stretched to 0..255.  Score maps of such images are smooth: at the
    reference's default threshold (0.5) a template has thousands of pixels above it, where a white-noise image
    (rand_u8, make_workload) has none besides the planted copies."""
    rng = np.random.default_rng(seed)
    acc = np.zeros(shape, np.float64)
    for k, wgt in zip(scales, weights):
        n = rng.standard_normal((shape[0] + k, shape[1] + k))
        c = np.cumsum(np.cumsum(n, axis=0), axis=1)
        acc += wgt * (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k])[:shape[0], :shape[1]] / k
    acc += noise * rng.standard_normal(shape)
    acc = (acc - acc.min()) / (acc.max() - acc.min())
    return np.round(acc * 255).astype(np.uint8)


def cut_templates(seed, image, n, side):
    """[(label, crop)] - n side x side crops of `image` at seeded positions."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        y, x = int(rng.integers(0, image.shape[0] - side)), int(rng.integers(0, image.shape[1] - side))
        out.append(("t%d" % i, image[y:y + side, x:x + side].copy()))
    return out


def _resize_area(a, side):
    """Own area-average resize of a square uint8 array to side x side (no cv2), in exact integer
    arithmetic so that every numpy/BLAS build produces the same bytes: output pixel i integrates
    the source interval [i*n/side, (i+1)*n/side); overlaps are multiples of 1/side."""
    n = a.shape[0]
    w = np.zeros((side, n), dtype=np.int64)          # overlap lengths * side; rows sum to n
    for i in range(side):
        lo, hi = i * n, (i + 1) * n                  # in units of 1/side source pixels
        for j in range(lo // side, min((hi + side - 1) // side, n)):
            w[i, j] = max(0, min(hi, (j + 1) * side) - max(lo, j * side))
    num = w @ a.astype(np.int64) @ w.T               # exact: < 2^63
    den = n * n
    return np.clip((2 * num + den) // (2 * den), 0, 255).astype(np.uint8)


def _disc_mask(side):
    yy, xx = np.mgrid[0:side, 0:side]
    c = (side - 1) / 2.0
    return (((yy - c) ** 2 + (xx - c) ** 2) <= (side / 2.0) ** 2).astype(np.uint8) * 255


def make_workload(seed, image_hw, n_base, templ=64, rotations=1, scales=None, masked=False,
                  noisy_per_unit=3, channels=1):
    """Returns (image, listTemplates, plants).

    listTemplates: [(label, uint8 array[, mask])] with n_base * rotations (or * len(scales)) units.
    plants: list of (label, (x, y, w, h), noise_amplitude) for every planted copy.
    """
    H, W = image_hw
    shape = (H, W) if channels == 1 else (H, W, channels)
    image = rand_u8(seed, 0, shape)
    units = []
    for b in range(n_base):
        tshape = (templ, templ) if channels == 1 else (templ, templ, channels)
        base = rand_u8(seed, 1000 + b, tshape)
        if scales is not None:
            for s in scales:
                side = int(s)
                t = base if side == templ else _resize_area(base, side)
                if masked:
                    units.append(("%d_s%d" % (b, side), t, _disc_mask(side)))
                else:
                    units.append(("%d_s%d" % (b, side), t))
        else:
            for k in range(rotations):
                t = np.ascontiguousarray(np.rot90(base, k))
                units.append(("%d_%d" % (b, 90 * k), t))
    max_side = max(u[1].shape[0] for u in units)
    cell = max_side + 32
    gx, gy = W // cell, H // cell
    n_plants = len(units) * (1 + noisy_per_unit)
    if gx * gy < n_plants:
        raise ValueError("image too small to plant %d copies (grid %dx%d)" % (n_plants, gx, gy))
    # a seeded permutation of the grid cells
    keys = _stream(seed, 7, gx * gy)
    cells = np.argsort(keys, kind="stable")[:n_plants]
    amps = [0, 40, 70, 100, 55, 85][:1 + noisy_per_unit]
    plants = []
    p = 0
    for ui, u in enumerate(units):
        t = u[1]
        th, tw = t.shape[:2]
        for ai, amp in enumerate(amps):
            cy, cx = divmod(int(cells[p]), gx)
            jx = int(rand_int(seed, 20000 + p, 1, 0, cell - tw + 1)[0])
            jy = int(rand_int(seed, 30000 + p, 1, 0, cell - th + 1)[0])
            x, y = cx * cell + jx, cy * cell + jy
            if amp == 0:
                patch = t
            else:
                noise = rand_u8(seed, 40000 + p, t.shape).astype(np.int32)
                noise = (noise * (2 * amp + 1)) // 256 - amp
                patch = np.clip(t.astype(np.int32) + noise, 0, 255).astype(np.uint8)
            if len(u) >= 3:      # masked: only the pixels under the mask are planted
                m = u[2] > 0
                region = image[y:y + th, x:x + tw]
                region[m] = patch[m]
            else:
                image[y:y + th, x:x + tw] = patch
            plants.append((u[0], (x, y, tw, th), amp))
            p += 1
    return image, units, plants


CONFIGS = {
    # BASELINE.json configs[1..4]; configs[0] is the coins/plumbing case (tests/golden).
    "cfg2": dict(seed=2, image_hw=(1080, 1920), n_base=8, templ=64),
    "cfg3": dict(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, rotations=4),
    "cfg3_32": dict(seed=3, image_hw=(2160, 3840), n_base=32, templ=64),
    "cfg4": dict(seed=4, image_hw=(2160, 3840), n_base=256, templ=64, noisy_per_unit=1),
    "cfg5": dict(seed=5, image_hw=(4320, 7680), n_base=16, templ=64,
                 scales=(32, 56, 80, 104, 128), masked=True),
}


def make_config(name, **override):
    kw = dict(CONFIGS[name])
    kw.update(override)
    return make_workload(**kw)
