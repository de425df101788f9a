"""Seeded synthetic video generator (BASELINE.md §3 / SURVEY.md §8d): there are no clips in the tree or on the box."""
import numpy as np


def make_scene(width, height, depth=8, seed=4321, tile=96, vmax=9, sigma=3.0):
    """BASELINE.md §3 generator: low-pass random texture; the source frame is the reference moved per `tile` x `tile`
    tile by its own vector in [-vmax, vmax]^2 plus Gaussian noise.  Returns dict(src=, ref=) of (height, width) arrays."""
    rng = np.random.default_rng(seed)
    pmax = (1 << depth) - 1
    pad = vmax + 8
    H, W = height + 2 * pad, width + 2 * pad
    base = rng.random((H // 8 + 3, W // 8 + 3))
    up = np.kron(base, np.ones((8, 8)))[:H + 16, :W + 16]
    k = np.ones(9) / 9.0
    up = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, up)
    up = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, up)[8:8 + H, 8:8 + W]
    up = (up - up.min()) / max(up.max() - up.min(), 1e-9)
    fine = rng.normal(0, 6.0 * pmax / 255.0, (H, W))
    big = np.clip(up * pmax * 0.8 + pmax * 0.1 + fine, 0, pmax)
    ref = big[pad:pad + height, pad:pad + width]
    src = np.empty_like(ref)
    for y0 in range(0, height, tile):
        for x0 in range(0, width, tile):
            dy, dx = int(rng.integers(-vmax, vmax + 1)), int(rng.integers(-vmax, vmax + 1))
            y1, x1 = min(y0 + tile, height), min(x0 + tile, width)
            src[y0:y1, x0:x1] = big[pad + y0 + dy:pad + y1 + dy, pad + x0 + dx:pad + x1 + dx]
    src = src + rng.normal(0, sigma * pmax / 255.0, src.shape)
    dt = np.uint8 if depth == 8 else np.uint16
    return {"src": np.clip(np.rint(src), 0, pmax).astype(dt), "ref": np.clip(np.rint(ref), 0, pmax).astype(dt)}


def make_clip(path, width, height, frames, seed=4321, tile=96, vmax=9, sigma=3.0, fade=False, csp="i420", depth=8):
    """Write an 8-bit I420 clip: a textured background whose tiles keep moving with their own constant velocity
    (half rate) + per-frame noise; chroma = 128 + 0.3 * (luma - 128) subsampled.  fade: the picture fades in from
    40 % to full brightness over the clip (weighted prediction has something to find).  csp: "i420" (default), "i422", "i444" or "i400"
    (x265 --input-csp): how the two chroma planes are subsampled / whether they exist.  depth 10: 16-bit little-endian samples (x265
    --input-depth 10), the 8-bit picture times four plus two more bits of noise."""
    rng = np.random.default_rng(seed)
    sy, sx = {"i420": (2, 2), "i422": (1, 2), "i444": (1, 1), "i400": (0, 0)}[csp]
    pad = vmax * frames // 2 + 16
    big = make_scene(width + 2 * pad, height + 2 * pad, 8, seed, tile, 0, 0.0)["ref"].astype(np.float64)
    ty, tx = (height + tile - 1) // tile, (width + tile - 1) // tile
    vel = rng.integers(-vmax, vmax + 1, size=(ty, tx, 2))
    with open(path, "wb") as f:
        for t in range(frames):
            luma = np.empty((height, width))
            for j in range(ty):
                for i in range(tx):
                    dy, dx = (vel[j, i] * t) // 2
                    y0, x0 = j * tile, i * tile
                    y1, x1 = min(y0 + tile, height), min(x0 + tile, width)
                    luma[y0:y1, x0:x1] = big[pad + y0 + dy:pad + y1 + dy, pad + x0 + dx:pad + x1 + dx]
            if fade:
                luma = luma * (0.4 + 0.6 * t / max(frames - 1, 1))
            luma = np.clip(np.rint(luma + rng.normal(0, sigma, luma.shape)), 0, 255)
            if depth == 8:
                out = lambda a: a.astype(np.uint8).tobytes()                                      # noqa: E731
            else:
                out = lambda a: (a.astype(np.uint16) * 4 + rng.integers(0, 4, a.shape, dtype=np.uint16)).astype("<u2").tobytes()   # noqa: E731
            f.write(out(luma))
            if sy:
                c = np.clip(np.rint(128 + 0.3 * (luma[::sy, ::sx] - 128)), 0, 255)
                f.write(out(c))
                f.write(out(np.ascontiguousarray(255 - c if csp != "i420" else c)))


def chroma_of(luma, depth, gain):
    """4:2:0 chroma plane derived from luma: mid + gain * (2x2-mean(luma) - mid)."""
    mid = 1 << (depth - 1)
    h, w = luma.shape
    m = luma.astype(np.float64).reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
    dt = np.uint8 if depth == 8 else np.uint16
    return np.clip(np.rint(mid + gain * (m - mid)), 0, (1 << depth) - 1).astype(dt)


def make_scene_yuv(width, height, depth=8, seed=4321, **kw):
    """make_scene plus Cb / Cr planes (half resolution) for source and reference."""
    sc = make_scene(width, height, depth, seed, **kw)
    rng = np.random.default_rng(seed + 99)
    out = {"src": sc["src"], "ref": sc["ref"]}
    for name, gain in (("cb", 0.3), ("cr", -0.2)):
        for k in ("src", "ref"):
            c = chroma_of(sc[k], depth, gain).astype(np.int64)
            c += rng.integers(-2, 3, c.shape) * (1 << (depth - 8))
            out[k + "_" + name] = np.clip(c, 0, (1 << depth) - 1).astype(sc[k].dtype)
    return out
