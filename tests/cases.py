"""Seeded test-vector generator shared by the CPU pinning tests, the golden-digest fixtures and the GPU parity tests.

Input distributions follow the reference TestBench: random / all-min / all-max pixel buffers
(source/test/pixelharness.cpp:31-80), residuals in [-PIXEL_MAX, PIXEL_MAX] and full-range int16 for inverse
transforms (source/test/mbdstharness.cpp:53-86), unaligned origins and strides (ipfilterharness.cpp:72-108) —
but with FIXED seeds (the reference seeds from time(), testbench.cpp:142).
"""
import hashlib

import numpy as np

from backends import PU_SIZES, CU_SIZES

TU_SIZES = [4, 8, 16, 32]
# primitives.h:80-90, indexed by LumaPU; 2x2 has no filter entry in the reference (ipfilter.cpp:419-470)
CHROMA_PU_420 = [(w // 2, h // 2) for (w, h) in PU_SIZES if (w, h) != (4, 4)]
MODES = ["rand", "min", "max"]


def pix_buf(rng, mode, shape, depth):
    dt = np.uint8 if depth == 8 else np.uint16
    pmax = (1 << depth) - 1
    if mode == "rand":
        return rng.integers(0, pmax + 1, size=shape, dtype=np.int64).astype(dt)
    return np.full(shape, 0 if mode == "min" else pmax, dt)


def short_buf(rng, mode, shape, lo, hi):
    if mode == "rand":
        return rng.integers(lo, hi + 1, size=shape, dtype=np.int64).astype(np.int16)
    return np.full(shape, lo if mode == "min" else hi, np.int16)


def gen_cases(depth, seed=1234, reps=2):
    """Yield (label, method_name, args) tuples; args are ready for Orc/Ref methods of tests/backends.py."""
    rng = np.random.default_rng(seed + depth)
    pmax = (1 << depth) - 1
    S = 96  # buffer stride, deliberately not a multiple of 64

    def origin(margin=0, room=64):
        return (int(rng.integers(margin, S - room - margin)), int(rng.integers(margin, S - room - margin)))

    combos = [("rand", "rand")] * reps + [("min", "max"), ("max", "min"), ("max", "max")]
    for (ma, mb) in combos:
        a = pix_buf(rng, ma, (S + 80, S), depth)
        b = pix_buf(rng, mb, (S + 80, S), depth)
        fenc = np.zeros((64, 64), a.dtype)
        fenc[:, :] = pix_buf(rng, ma, (64, 64), depth)
        tag = "%s-%s" % (ma, mb)
        for (w, h) in PU_SIZES:
            ao, bo = origin(), origin()
            yield ("sad %dx%d %s" % (w, h, tag), "sad", (w, h, a, ao, b, bo))
            yield ("satd %dx%d %s" % (w, h, tag), "satd", (w, h, a, ao, b, bo))
            offs = [origin() for _ in range(4)]
            yield ("sad_x3 %dx%d %s" % (w, h, tag), "sad_xn", (w, h, fenc, b, offs[:3]))
            yield ("sad_x4 %dx%d %s" % (w, h, tag), "sad_xn", (w, h, fenc, b, offs))
            yield ("pixelavg %dx%d %s" % (w, h, tag), "pixelavg_pp", (w, h, a, ao, b, bo))
            yield ("p2s %dx%d %s" % (w, h, tag), "p2s", (w, h, a, ao))
        for size in CU_SIZES:
            ao, bo = origin(), origin()
            yield ("sa8d %d %s" % (size, tag), "sa8d", (size, a, ao, b, bo))
            yield ("sse_pp %d %s" % (size, tag), "sse_pp", (size, a, ao, b, bo))
            yield ("psy_cost %d %s" % (size, tag), "psy_cost_pp", (size, a, ao, b, bo))
            yield ("var %d %s" % (size, tag), "var", (size, a, ao))
            yield ("sub_ps %d %s" % (size, tag), "sub_ps", (size, a, ao, b, bo))

    # int16 families
    for mode in ["rand"] * reps + ["min", "max"]:
        r = short_buf(rng, mode, (S + 80, S), -pmax, pmax)        # residual-range
        r2 = short_buf(rng, mode, (S + 80, S), -pmax, pmax)
        full = short_buf(rng, mode, (S + 80, S), -32768, 32767)   # full int16 range
        p = pix_buf(rng, mode, (S + 80, S), depth)
        avg = short_buf(rng, mode, (S + 80, S), -16384, 16383)    # addAvg operand range (pixelharness.cpp:41)
        avg2 = short_buf(rng, "rand", (S + 80, S), -16384, 16383)
        for size in CU_SIZES:
            ao, bo = origin(), origin()
            yield ("sse_ss %d %s" % (size, mode), "sse_ss", (size, r, ao, r2, bo))
            yield ("ssd_s %d %s" % (size, mode), "ssd_s", (size, r, ao))
            yield ("add_ps %d %s" % (size, mode), "add_ps", (size, p, ao, r, bo))
        for (w, h) in PU_SIZES:
            ao, bo = origin(), origin()
            yield ("addAvg %dx%d %s" % (w, h, mode), "addAvg", (w, h, avg, ao, avg2, bo))
        for size in TU_SIZES:
            ao = origin()
            yield ("dct %d %s" % (size, mode), "dct", (size, r, ao))
            flat = np.ascontiguousarray(full[:size, :size]).reshape(-1).copy()
            yield ("idct %d %s" % (size, mode), "idct", (size, flat))
            lim = (1 << (depth + 4)) - 1                           # mbufidct range, mbdstharness.cpp:83
            flat2 = short_buf(rng, "rand", (size * size,), -lim, lim)
            yield ("idct-typ %d %s" % (size, mode), "idct", (size, flat2))
            yield ("cpy2Dto1D_shl %d %s" % (size, mode), "cpy2Dto1D_shl", (size, r, ao, int(rng.integers(0, 4))))
            yield ("cpy2Dto1D_shr %d %s" % (size, mode), "cpy2Dto1D_shr", (size, r, ao, int(rng.integers(1, 5))))
            flat3 = np.ascontiguousarray(r[:size, :size]).reshape(-1).copy()
            yield ("cpy1Dto2D_shl %d %s" % (size, mode), "cpy1Dto2D_shl", (size, flat3, int(rng.integers(0, 4))))
            yield ("cpy1Dto2D_shr %d %s" % (size, mode), "cpy1Dto2D_shr", (size, flat3, int(rng.integers(1, 5))))
            sparse = (flat3 * (rng.integers(0, 3, size=flat3.shape) == 0)).astype(np.int16)
            yield ("count_nonzero %d %s" % (size, mode), "count_nonzero", (size, sparse))
            sp2 = (r * (rng.integers(0, 3, size=r.shape) == 0)).astype(np.int16)
            yield ("copy_cnt %d %s" % (size, mode), "copy_cnt", (size, sp2, ao))
        ao = origin()
        yield ("dst4 %s" % mode, "dst4", (r, ao))
        yield ("idst4 %s" % mode, "idst4", (np.ascontiguousarray(full[:4, :4]).reshape(-1).copy(),))

    # quant / dequant (parameter ranges from common/quant.cpp:465-466, :567, :621)
    for rep in range(reps + 2):
        for size in TU_SIZES:
            n = size * size
            log2n = size.bit_length() - 1
            coef = short_buf(rng, "rand", (n,), -32768, 32767) if rep else short_buf(rng, "rand", (n,), -pmax * 8, pmax * 8)
            qp = int(rng.integers(0, 52))
            # quantCoeff = quantScales[qp%6] for flat lists (common/scalinglist.cpp:129) or arbitrary below 1<<15
            qs = [26214, 23302, 20560, 18396, 16384, 14564][qp % 6]
            qc = np.full(n, qs, np.int32) if rep % 2 == 0 else rng.integers(1, 1 << 15, size=n, dtype=np.int64).astype(np.int32)
            qbits = 14 + qp // 6 + (15 - depth - log2n)
            add = (171 if rep % 2 else 85) << (qbits - 9)
            yield ("quant %d #%d" % (size, rep), "quant", (coef, qc, qbits, add))
            yield ("nquant %d #%d" % (size, rep), "nquant", (coef, qc, qbits, 1 << (qbits - 1)))
            q = short_buf(rng, "rand", (n,), -32768, 32767) if rep == 0 else short_buf(rng, "rand", (n,), -512, 512)
            shift = 20 - 14 - (15 - depth - log2n)                 # QUANT_IQUANT_SHIFT - QUANT_SHIFT - transformShift
            scale = [40, 45, 51, 57, 64, 72][qp % 6] << (qp // 6)
            if shift >= 1 and scale < 32768:
                yield ("dequant_normal %d #%d" % (size, rep), "dequant_normal", (q, scale, shift))
            dq = rng.integers(1, 72 * 16 + 1, size=n, dtype=np.int64).astype(np.int32)
            per = qp // 6
            yield ("dequant_scaling %d #%d" % (size, rep), "dequant_scaling", (q, dq, per, max(shift, 0)))
            rs = rng.integers(0, 1 << 20, size=n, dtype=np.int64).astype(np.uint32)
            off = rng.integers(0, 1 << 10, size=n, dtype=np.int64).astype(np.uint16)
            yield ("denoise %d #%d" % (size, rep), "denoise_dct", (coef, rs, off))
            resi = short_buf(rng, "rand", (n,), -32768, 32767)
            fe = short_buf(rng, "rand", (n,), -32768, 32767)
            cgx, cgy = int(rng.integers(0, size // 4)), int(rng.integers(0, size // 4))
            blk = cgy * 4 * size + cgx * 4
            psy = int(rng.integers(0, 1 << 20))
            for kind in ("nonpsy", "psy", "psy1", "psy2"):
                yield ("rdoq-%s %d #%d" % (kind, size, rep), "rdoquant", (kind, size, resi, fe, psy, blk))

    # interpolation: luma (8 tap) all PU sizes, chroma 4:2:0 (4 tap) for the chroma PU sizes
    for mode in ["rand"] * reps + ["min", "max"]:
        p = pix_buf(rng, mode, (S + 88, S), depth)
        sh = short_buf(rng, mode, (S + 88, S), -8192, 8191 if depth == 8 else 8191)   # 14-bit intermediates
        for chroma, sizes in ((0, PU_SIZES), (1, CHROMA_PU_420)):
            nph = 8 if chroma else 4
            for (w, h) in sizes:
                so = origin(margin=8, room=72)
                idx, idy = int(rng.integers(0 if chroma else 1, nph)), int(rng.integers(1, nph))
                lab = "%s %dx%d %s" % ("chroma" if chroma else "luma", w, h, mode)
                for kind in ("hpp", "vpp", "vps"):
                    yield ("%s %s" % (kind, lab), "interp", (kind, chroma, w, h, p, so, idx))
                yield ("hps %s" % lab, "interp", ("hps", chroma, w, h, p, so, idx, 0, 0))
                yield ("hps-ext %s" % lab, "interp", ("hps", chroma, w, h, p, so, idx, 0, 1))
                yield ("vsp %s" % lab, "interp", ("vsp", chroma, w, h, sh, so, idx))
                yield ("vss %s" % lab, "interp", ("vss", chroma, w, h, sh, so, idx))
                if not chroma:
                    yield ("hvpp %s" % lab, "interp", ("hvpp", chroma, w, h, p, so, max(idx, 1), idy))


    # intra prediction (source/test/intrapredharness.cpp: random neighbour lines, every mode, bFilter 0/1) and the lowres
    # downscale — appended last so the cases above keep their rng stream (and their golden digests)
    for mode in ["rand"] * reps + ["min", "max"]:
        for n in TU_SIZES:
            nb = pix_buf(rng, mode, (4 * n + 1,), depth)
            nbf = pix_buf(rng, mode, (4 * n + 1,), depth)
            yield ("intra_filter %d %s" % (n, mode), "intra_filter", (n, nb))
            for m in range(35):
                for bf in (0, 1):
                    yield ("intra_pred %d m%d f%d %s" % (n, m, bf, mode), "intra_pred", (n, m, nb, bf))
            for bl in (0, 1):
                yield ("intra_allangs %d l%d %s" % (n, bl, mode), "intra_allangs", (n, nb, nbf, bl))
        src = pix_buf(rng, mode, (70, 2 * 40 + 8), depth)
        yield ("frame_init_lowres %s" % mode, "frame_init_lowres", (src, (2, 1), 40, 33))


    # weighted prediction (pixelharness.cpp check_weightp / check_weightpUni: w0 in [0, 127], shift includes the 14-bit correction),
    # the 64x64 intra-scan downscales and transposes — appended last like the intra cases
    for mode in ["rand"] * reps + ["min", "max"]:
        p = pix_buf(rng, mode, (S + 80, S), depth)
        sh = short_buf(rng, mode, (S + 80, S), -8192, 8191)
        corr = 14 - depth
        for (w, h) in ((16, 16), (64, 24), (48, 64), (32, 7)):
            ao = origin(room=64)
            w0, shift, offset = int(rng.integers(1, 128)), int(rng.integers(0, 7)) + corr, int(rng.integers(-20, 21)) * (1 << (depth - 8))
            rnd = (1 << (shift - 1)) if shift else 0
            yield ("weight_pp %dx%d %s" % (w, h, mode), "weight_pp", (p, ao, w, h, w0, rnd, shift, offset))
            yield ("weight_sp %dx%d %s" % (w, h, mode), "weight_sp", (sh, ao, w, h, w0, rnd, shift, offset))
        yield ("scale1D %s" % mode, "scale1d_128to64", (pix_buf(rng, mode, (256,), depth),))
        big = pix_buf(rng, mode, (S + 80, S), depth)
        yield ("scale2D %s" % mode, "scale2d_64to32", (big, origin(room=64)))
        for size in (4, 8, 16, 32, 64):
            yield ("transpose %d %s" % (size, mode), "transpose", (size, big, origin(room=64)))


def textured_frame(rng, h, w, depth, sigma=3.0):
    """Low-pass random texture + noise: SADs then have a meaningful minimum (BASELINE.md §3 generator, scaled down)."""
    pmax = (1 << depth) - 1
    base = rng.random((h // 8 + 3, w // 8 + 3))
    up = np.kron(base, np.ones((8, 8)))[:h + 16, :w + 16]
    k = np.ones(9) / 9.0
    up = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, up)
    up = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, up)[8:8 + h, 8:8 + w]
    up = (up - up.min()) / max(up.max() - up.min(), 1e-9)
    img = up * pmax * 0.8 + pmax * 0.1 + rng.normal(0, sigma * (pmax / 255.0), (h, w))
    return np.ascontiguousarray(np.clip(np.rint(img), 0, pmax).astype(np.uint8 if depth == 8 else np.uint16))


def me_scene(depth, seed, H=160, W=192, margin=80):
    """A padded reference plane and a source plane that is the reference shifted per 48x40 tile + noise."""
    rng = np.random.default_rng(seed)
    ref = textured_frame(rng, H + 2 * margin, W + 2 * margin, depth)
    src = np.zeros_like(ref)
    pmax = (1 << depth) - 1
    th, tw = 40, 48                      # tile size: each tile moves by its own vector in [-9, 9]^2
    for y0 in range(margin, margin + H, th):
        for x0 in range(margin, margin + W, tw):
            dy, dx = int(rng.integers(-9, 10)), int(rng.integers(-9, 10))
            y1, x1 = min(y0 + th, margin + H), min(x0 + tw, margin + W)
            src[y0:y1, x0:x1] = ref[y0 + dy:y1 + dy, x0 + dx:x1 + dx]
    noise = rng.normal(0, 2.0 * (pmax / 255.0), src.shape)
    src = np.clip(np.rint(src.astype(np.float64) + noise), 0, pmax).astype(ref.dtype)
    return np.ascontiguousarray(ref), np.ascontiguousarray(src), margin


def digest(result):
    """sha256 over the canonical bytes of a scalar / array / tuple result."""
    h = hashlib.sha256()

    def feed(x):
        if isinstance(x, tuple):
            for e in x:
                feed(e)
        elif isinstance(x, np.ndarray):
            h.update(str(x.dtype).encode() + str(x.shape).encode())
            h.update(np.ascontiguousarray(x).tobytes())
        else:
            h.update(b"i" + str(int(x)).encode())
    feed(result)
    return h.hexdigest()[:16]


def lowres_scene(depth, seed, H=136, W=200, margin=80):
    """A source plane with PicYuv-style extended margins (the lookahead's downscale reads past the picture edge)."""
    rng = np.random.default_rng(seed)
    pic = textured_frame(rng, H, W, depth, sigma=4.0)
    return np.ascontiguousarray(np.pad(pic, ((margin, margin), (margin, margin + 8)), mode="edge")), margin


def lookahead_scene(depth, seed, H=136, W=200, margin=80):
    """Two padded source pictures: the second is the first moved per 48x40 tile (vectors in [-9, 9]^2) plus noise — what the
    lookahead's lowres search sees between consecutive frames."""
    rng = np.random.default_rng(seed)
    pmax = (1 << depth) - 1
    big = textured_frame(rng, H + 32, W + 32, depth, sigma=2.0)
    p0 = big[16:16 + H, 16:16 + W]
    p1 = np.zeros_like(p0)
    th, tw = 40, 48
    for y0 in range(0, H, th):
        for x0 in range(0, W, tw):
            dy, dx = int(rng.integers(-9, 10)), int(rng.integers(-9, 10))
            y1, x1 = min(y0 + th, H), min(x0 + tw, W)
            p1[y0:y1, x0:x1] = big[16 + y0 + dy:16 + y1 + dy, 16 + x0 + dx:16 + x1 + dx]
    p1 = np.clip(np.rint(p1.astype(np.float64) + rng.normal(0, 2.0 * (pmax / 255.0), p1.shape)), 0, pmax).astype(p0.dtype)
    pad = ((margin, margin), (margin, margin + 8))
    return np.ascontiguousarray(np.pad(p0, pad, mode="edge")), np.ascontiguousarray(np.pad(p1, pad, mode="edge")), margin


def me_scene_yuv(depth, seed, H=160, W=192, margin=80):
    """me_scene plus Cb / Cr planes at half resolution (half margins) that move with the luma."""
    ref, src, m = me_scene(depth, seed, H, W, margin)
    rng = np.random.default_rng(seed + 5)
    mid = 1 << (depth - 1)
    pmax = (1 << depth) - 1

    def chroma(y, gain):
        sub = (y[0::2, 0::2].astype(np.int64) + y[1::2, 0::2] + y[0::2, 1::2] + y[1::2, 1::2] + 2) >> 2
        c = mid + np.rint(gain * (sub - mid)).astype(np.int64) + rng.integers(-2, 3, sub.shape) * (1 << (depth - 8))
        return np.ascontiguousarray(np.clip(c, 0, pmax).astype(y.dtype))
    return (ref, chroma(ref, 0.6), chroma(ref, -0.5)), (src, chroma(src, 0.6), chroma(src, -0.5)), m


def lookahead_scene3(depth, seed, H=136, W=200, margin=80):
    """Three consecutive padded pictures (constant per-tile motion across them) for the B-frame cost pass."""
    rng = np.random.default_rng(seed)
    pmax = (1 << depth) - 1
    big = textured_frame(rng, H + 64, W + 64, depth, sigma=2.0)
    th, tw = 40, 48
    vec = {(y0, x0): (int(rng.integers(-7, 8)), int(rng.integers(-7, 8))) for y0 in range(0, H, th) for x0 in range(0, W, tw)}
    pics = []
    for t in range(3):
        p = np.zeros((H, W), big.dtype)
        for (y0, x0), (dy, dx) in vec.items():
            y1, x1 = min(y0 + th, H), min(x0 + tw, W)
            p[y0:y1, x0:x1] = big[32 + y0 + t * dy:32 + y1 + t * dy, 32 + x0 + t * dx:32 + x1 + t * dx]
        p = np.clip(np.rint(p.astype(np.float64) + rng.normal(0, 2.0 * (pmax / 255.0), p.shape)), 0, pmax).astype(big.dtype)
        pics.append(np.ascontiguousarray(np.pad(p, ((margin, margin), (margin, margin + 8)), mode="edge")))
    return pics, margin


def umh_scenes(depth):
    """Scenes that reach every branch of the UMH search (encoder/motion.cpp:946-1130): the early-termination tests need blocks that match
    well at the predictor (SAD_THRESH), the `cross_start = range + 2` branch needs a cost that is flat under the radius-1 diamond and the
    radius-2 octagon but better a few pixels along an axis (stripes of period 3..6), and the adaptive range needs poor matches too.
    Yields (ref, src, margin, H, W, (dy, dx) true shift)."""
    pmax = (1 << depth) - 1
    sc = 1 << (depth - 8)
    dt = np.uint8 if depth == 8 else np.uint16
    m, H, W = 96, 160, 192
    yy, xx = np.mgrid[0:H + 2 * m, 0:W + 2 * m]
    for k, sigma in enumerate((0.0, 0.7, 3.0, 25.0)):                      # smooth blocks + noise of growing strength
        rng = np.random.default_rng(4000 + 10 * depth + k)
        base = rng.integers(0, 256, size=((H + 2 * m) // 8 + 2, (W + 2 * m) // 8 + 2)).astype(np.float64)
        big = np.kron(base, np.ones((8, 8)))[:H + 2 * m, :W + 2 * m]
        big = (big + np.roll(big, 1, 0) + np.roll(big, 1, 1) + np.roll(big, 3, 0) + np.roll(big, 3, 1)) / 5
        ref = np.clip(np.rint((big + rng.normal(0, 2, big.shape)) * sc), 0, pmax)
        shift = (int(rng.integers(-9, 10)), int(rng.integers(-9, 10)))
        src = np.clip(np.rint(np.roll(ref, shift, (0, 1)) + rng.normal(0, sigma * sc, ref.shape)), 0, pmax)
        yield np.ascontiguousarray(ref.astype(dt)), np.ascontiguousarray(src.astype(dt)), m, H, W, shift
    for k, (per, axis) in enumerate(((3, 0), (3, 1), (5, 0), (4, 1))):      # stripes: local minima every `per` pixels along one axis
        rng = np.random.default_rng(4100 + 10 * depth + k)
        stripes = (((xx if axis == 0 else yy) % per) == 0) * 60.0
        smooth = 100 + 20 * np.sin(xx / 37.0) + 20 * np.cos(yy / 29.0)
        ref = np.clip(np.rint((smooth + stripes) * sc), 0, pmax)
        shift = [0, 0]
        shift[1 - axis] = per * int(rng.choice([-2, -1, 1, 2]))
        src = np.clip(np.rint(np.roll(ref, tuple(shift), (0, 1)) + rng.normal(0, 0.6 * sc, ref.shape)), 0, pmax)
        yield np.ascontiguousarray(ref.astype(dt)), np.ascontiguousarray(src.astype(dt)), m, H, W, (shift[0], shift[1], per, axis)


def umh_groups(depth, groups_per_scene=6, npu=8):
    """Batches of PUs for the scenes above: each group is one (w, h, merange, subme, qp, numCand) with npu PUs, predictors near the true
    motion half of the time.  Yields (scene_index, ref, src, group dict)."""
    sizes = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (32, 16), (8, 16), (64, 32), (24, 32), (16, 12)]
    for si, (ref, src, m, H, W, info) in enumerate(umh_scenes(depth)):
        rng = np.random.default_rng(4200 + 100 * depth + si)
        dy, dx = info[0], info[1]
        per = info[2] if len(info) == 4 else 0
        for gi in range(groups_per_scene):
            w, h = sizes[int(rng.integers(0, len(sizes)))]
            merange = int(rng.choice([8, 16, 32, 57]))
            numCand = int(rng.integers(0, 4))
            g = dict(w=w, h=h, merange=merange, subme=int(rng.choice([0, 2, 3, 7])), qp=int(rng.choice([22, 37])), numCand=numCand,
                     pus=[], mins=[], maxs=[], mvps=[], cands=[])
            for _ in range(npu):
                bx = m + int(rng.integers(0, (W - w) // 4 + 1)) * 4
                by = m + int(rng.integers(0, (H - h) // 4 + 1)) * 4
                if per:                                 # predictor a whole number of periods away from the truth (or on it)
                    off = per * int(rng.integers(-1, 2))
                    qmvp = (-dx * 4 + (4 * off if info[3] == 0 else 0), -dy * 4 + (4 * off if info[3] == 1 else 0))
                elif rng.integers(0, 2):
                    qmvp = (-dx * 4 + int(rng.integers(-6, 7)), -dy * 4 + int(rng.integers(-6, 7)))
                else:
                    qmvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                mvmin = [(qmvp[0] >> 2) - merange, (qmvp[1] >> 2) - merange]
                mvmax = [(qmvp[0] >> 2) + merange, (qmvp[1] >> 2) + merange]
                k = int(rng.integers(0, 4))
                if k == 0:
                    mvmax[1] = max(min(mvmax[1], int(rng.integers(0, 6))), mvmin[1])        # frame-parallel row lag
                if k == 1:
                    mvmin[1] = min(max(mvmin[1], int(rng.integers(-3, 4))), mvmax[1])
                g["pus"].append((bx, by)); g["mins"].append(tuple(mvmin)); g["maxs"].append(tuple(mvmax)); g["mvps"].append(qmvp)
                g["cands"].append([((-dx * 4 + int(rng.integers(-8, 9)), -dy * 4 + int(rng.integers(-8, 9))) if rng.integers(0, 2)
                                    else (int(rng.integers(-60, 61)), int(rng.integers(-60, 61)))) for _ in range(numCand)])
            yield si, ref, src, g


def coef_cases(seed=77):
    """Cases for the coefficient-scan cost primitives, drawn as test/pixelharness.cpp:1705-2060 draws them (two thirds zeros, mostly negative
    levels, every scan type and TU size, random CABAC states 2..124).  Yields (label, method, args) for Orc / Ref / Hip methods."""
    rng = np.random.default_rng(seed)

    def coeffs(n, density):
        v = rng.integers(1, 0x7FFF, size=n).astype(np.int32)
        v[rng.random(n) >= density] = 0
        v[rng.random(n) < 0.8] *= -1
        small = rng.random(n) < 0.5
        v[small] = np.sign(v[small]) * rng.integers(1, 4, size=int(small.sum()))
        return v.astype(np.int16)
    for stype in range(3):
        for log2 in (2, 3, 4, 5):
            size = 1 << log2
            for rep in range(6):
                density = [0.33, 0.05, 0.6, 0.01, 0.2, 1.0][rep]
                tu = coeffs(size * size, density).reshape(size, size)
                if not tu.any():
                    tu[-1, -1] = -1
                if rep == 3:                                   # a single coefficient somewhere
                    tu[:] = 0
                    tu[int(rng.integers(0, size)), int(rng.integers(0, size))] = int(rng.choice([-1, 1, 300, -32768]))
                yield ("scanPosLast t%d %dx%d #%d" % (stype, size, size, rep), "scan_pos_last", (log2, stype, tu.copy()))
                ncg = (size // 4) ** 2
                for k in range(3):
                    cgx, cgy = int(rng.integers(0, size // 4)), int(rng.integers(0, size // 4))
                    if tu[cgy * 4:cgy * 4 + 4, cgx * 4:cgx * 4 + 4].any():
                        yield ("findPosFirstLast t%d %dx%d #%d.%d" % (stype, size, size, rep, k), "find_pos_first_last", (tu.copy(), cgx, cgy, stype))
                    cg = int(rng.integers(0, ncg))
                    off = int(rng.integers(0, 16))
                    ctx = rng.integers(2, 125, size=64).astype(np.uint8)
                    offset = 0 if log2 == 2 else (9 if log2 == 3 else 12)
                    yield ("costCoeffNxN t%d %dx%d #%d.%d" % (stype, size, size, rep, k), "cost_coeff_nxn",
                           (tu.copy(), log2, stype, cg, off, int(rng.integers(0, 4)), offset, ctx))
    for rep in range(120):
        a = rng.integers(0, 0x8000, size=16).astype(np.uint16)
        a[rng.random(16) < 0.6] = 1
        if rep % 3 == 0:
            a = np.minimum(a, rng.integers(1, 60, size=16)).astype(np.uint16)
        nnz = int(rng.integers(1, 17))
        first = 0
        while first < 8 and a[first] < 2:
            first += 1
        if first < nnz or rep % 2:
            yield ("costCoeffRemain #%d" % rep, "cost_coeff_remain", (a.copy(), max(nnz, min(first, 15) + 1), min(first, 15)))
    for rep in range(120):
        vals = []
        for _ in range(8):
            v = int(rng.integers(0, 0x8000))
            v = 0 if v < 0x7FFF // 3 else (1 if v < 0x7FFF * 2 // 3 else (2 if v < 0x7FFF * 3 // 4 else v))
            if v:
                vals.append(v)
        if not vals:
            vals = [1]
        a = np.zeros(16, np.uint16)
        a[:len(vals)] = vals
        ctx = rng.integers(2, 125, size=8).astype(np.uint8)
        yield ("costC1C2Flag #%d" % rep, "cost_c1c2_flag", (a, len(vals), ctx, int(rng.integers(0, 4)) + 4))


def loop_cases(depth, seed=91):
    """Cases for the in-loop filter primitives, drawn as test/pixelharness.cpp draws them (random pictures incl. all-min / all-max, tc and
    masks over their ranges, sign buffers in {-1, 0, 1}, CTU-sized statistics blocks with ragged ends).  Yields (label, method, args)."""
    rng = np.random.default_rng(seed + depth)
    pmax = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    H, W = 96, 112

    def pic(mode):
        if mode == "rand":
            return rng.integers(0, pmax + 1, size=(H, W)).astype(dt)
        if mode == "smooth":                        # neighbouring samples often equal or one apart: every edge class occurs
            base = rng.integers(0, pmax - 8, size=(H // 4 + 1, W // 4 + 1))
            return (np.kron(base, np.ones((4, 4), np.int64))[:H, :W] + rng.integers(0, 3, size=(H, W))).astype(dt)
        return np.full((H, W), 0 if mode == "min" else pmax, dt)
    signs = lambda n: rng.integers(-1, 2, size=n).astype(np.int8)      # noqa: E731
    eo = lambda: rng.integers(-7, 8, size=5).astype(np.int8)           # noqa: E731
    for mi, kind_of_picture in enumerate(("rand", "smooth", "rand", "smooth", "min", "max")):
        p = pic(kind_of_picture)
        mode = "%s%d" % (kind_of_picture, mi)
        for rep in range(4):
            pos = (int(rng.integers(8, 16)), int(rng.integers(8, 16)))
            tc = int(rng.integers(0, pmax))
            for d in (0, 1):
                yield ("pelFilterLumaStrong dir%d %s #%d" % (d, mode, rep), "pel_filter_luma_strong", (p, pos, d, tc & int(rng.integers(-1, pmax)), tc & int(rng.integers(-1, pmax))))
                yield ("pelFilterChroma dir%d %s #%d" % (d, mode, rep), "pel_filter_chroma", (p, pos, d, int(rng.integers(0, 30)) << (depth - 8), -int(rng.integers(0, 2)), -int(rng.integers(0, 2))))
            w = int(rng.choice([8, 16, 17, 32, 33, 48, 64]))
            yield ("sign %s #%d" % (mode, rep), "sao_sign", (np.ascontiguousarray(p[pos[0], :w + 5]), np.ascontiguousarray(p[pos[0] + 1, :w + 5])))
            yield ("saoCuOrgE0 %s w%d #%d" % (mode, w, rep), "sao_e0", (p, pos, eo(), w, signs(2)))
            yield ("saoCuOrgE1 %s w%d #%d" % (mode, w, rep), "sao_e1", (p, pos, signs(w), eo(), w, 1))
            yield ("saoCuOrgE1_2Rows %s w%d #%d" % (mode, w, rep), "sao_e1", (p, pos, signs(w), eo(), w, 2))
            yield ("saoCuOrgE2 %s w%d #%d" % (mode, w, rep), "sao_e2", (p, pos, signs(w + 1), signs(w), eo(), w))
            s0 = int(rng.integers(0, 2))
            yield ("saoCuOrgE3 %s w%d #%d" % (mode, w, rep), "sao_e3", (p, pos, signs(w + 1), eo(), s0, w - int(rng.integers(0, 2))))
            yield ("saoCuOrgB0 %s w%d #%d" % (mode, w, rep), "sao_b0", (p, pos, rng.integers(-7, 8, size=32).astype(np.int8), w, int(rng.integers(1, 65))))
            diff = rng.integers(-pmax, pmax + 1, size=(64, 64)).astype(np.int16)
            for kind in range(5):
                endX = 64 - int(rng.integers(0, 5)) - (1 if kind >= 3 else 0)
                endY = 64 - int(rng.integers(0, 4)) - (1 if kind >= 1 else 0)
                if rep == 3:
                    endX, endY = int(rng.integers(1, 20)), int(rng.integers(1, 6))
                ncls = 32 if kind == 0 else 5
                yield ("saoCuStats%s %s %dx%d #%d" % (["BO", "E0", "E1", "E2", "E3"][kind], mode, endX, endY, rep), "sao_stats",
                       (kind, diff, p, pos, endX, endY, rng.integers(0, 1 << 20, size=ncls).astype(np.int32), rng.integers(0, 1 << 20, size=ncls).astype(np.int32),
                        signs(endX + 2), signs(endX + 2)))


def deblock_cases(depth, seed=55):
    """Inner edges of one 64x64 CTU for Deblock::edgeFilterLuma / edgeFilterChroma: pictures that are smooth across the edge (so the beta / strong
    decisions go both ways), random boundary strengths, QPs 10..51 per unit, slice offsets, optional lossless flags.  Yields (label, method, args)."""
    rng = np.random.default_rng(seed + depth)
    pmax = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    H, W = 96, 112
    sc = 1 << (depth - 8)
    for rep in range(10):
        # blocky content: 8x8 blocks of nearly flat values with small steps between neighbours, plus a little noise
        step = [2, 6, 15, 40][rep % 4]
        base = np.cumsum(rng.integers(-step, step + 1, size=(H // 8 + 1, W // 8 + 1)), axis=1) + 128
        y = np.clip((np.kron(base, np.ones((8, 8), np.int64))[:H, :W] + rng.integers(-1, 2, size=(H, W)) * (rep % 3 == 0)) * sc, 0, pmax).astype(dt)
        cb = np.clip((np.kron(base[:H // 16 + 1, :W // 16 + 1], np.ones((8, 8), np.int64))[:H // 2, :W // 2] + rng.integers(-2, 3, size=(H // 2, W // 2))) * sc, 0, pmax).astype(dt)
        cr = np.clip(pmax - cb.astype(np.int64) + rng.integers(-3, 4, size=cb.shape) * sc, 0, pmax).astype(dt)
        ctu = (16, 24)
        for edgeDir in (0, 1):
            for edge in (2, 4, 8, 14):
                bs = rng.integers(0, 3, size=(16, 16)).astype(np.uint8)
                qp = rng.integers(10, 52, size=(16, 16)).astype(np.int8) if rep % 2 else np.full((16, 16), int(rng.integers(20, 45)), np.int8)
                bypass = rng.integers(0, 2, size=(16, 16)).astype(np.uint8) if rep % 5 == 4 else None
                offs = (int(rng.integers(-3, 4)), int(rng.integers(-3, 4)), int(rng.integers(-6, 7)), int(rng.integers(-6, 7))) if rep % 3 == 2 else (0, 0, 0, 0)
                chroma = 1 if edge % 4 == 0 else 0                 # chroma edges sit on the 8-sample chroma grid
                yield ("deblock dir%d edge%d #%d" % (edgeDir, edge, rep), "deblock_ctu_edge",
                       ((y, cb, cr), ctu, edgeDir, edge, bs, qp, bypass, offs[0], offs[1], offs[2], offs[3], 1, chroma))


def weight_scenes(depth, seed=300):
    """Pairs of pictures for the lookahead's weighted-prediction analysis: the second is the first faded (gain / offset) and moved a little,
    plus pairs that should NOT be weighted (same brightness).  Yields (label, src0, src1, margin, H, W, (fencSsd, fencSum, refSsd, refSum))."""
    for k, (gain, off, H, W) in enumerate(((0.8, 6, 136, 200), (1.25, -10, 136, 200), (1.0, 0, 136, 200), (0.6, 30, 144, 176), (1.0, 12, 66, 50),
                                           (0.3, 90, 200, 320), (1.9, -60, 136, 200), (1.02, 0, 136, 200))):
        s0, s1, m = lookahead_scene(depth, seed + 10 * depth + k, H, W)
        sc = 1 << (depth - 8)
        pmax = (1 << depth) - 1
        f = np.clip(np.rint(s1.astype(np.float64) * gain + off * sc), 0, pmax).astype(s1.dtype)
        stats = []
        for pic in (f, s0):
            core = pic[m:m + H, m:m + W].astype(np.int64)
            n = core.size
            sm, sq = int(core.sum()), int((core * core).sum())
            stats += [(sq - (sm * sm + n // 2) // n) // 4, sm // 4]
        # the statistics are inputs of the analysis (AQ leaves them in the Lowres); weightsAnalyse divides the sum by the LOWRES area
        # (slicetype.cpp:892), so sums of that scale are what lets the test reach its weighting branch
        yield ("fade gain %.2f off %d %dx%d" % (gain, off, W, H), s0, f, m, H, W, tuple(stats))


def aq_cases(depth):
    """calcAdaptiveQuantFrame cases: (label, yuv planes, origin, w, h, qgSize, aqMode, aqStrength, weightp) over picture sizes with ragged
    block edges, both quantisation-group sizes, the three AQ modes, strengths incl. 0, and the statistics-only path (AQ off, weightp on)."""
    out = []
    k = 0
    for (w, h) in ((200, 136), (176, 144), (66, 50), (320, 200)):
        (ry, rcb, rcr), (sy, scb, scr), m = me_scene_yuv(depth, 900 + depth + w, H=h, W=w, margin=80)
        for (qg, mode, strength, wp) in ((16, 1, 1.0, 1), (16, 2, 1.0, 1), (16, 3, 0.8, 0), (8, 2, 1.0, 1), (8, 1, 1.5, 0), (8, 3, 0.6, 1), (16, 0, 0.0, 1),
                                         (16, 2, 0.0, 1)):
            out.append(("aq %dx%d qg%d mode%d s%.1f wp%d" % (w, h, qg, mode, strength, wp), (sy, scb, scr), (m, m), w, h, qg, mode, strength, wp))
            k += 1
    return out


def cutree_cases(seed=70):
    """Inputs for the CU-tree propagation step: random per-block costs (some blocks intra: inter cost >= intra cost), vectors that land
    inside, on the edge of and outside the frame, P and B (both lists, one list), referenced or not, saturating reference costs.
    Yields (label, args of Orc / Ref / Hip .cutree_propagate)."""
    rng = np.random.default_rng(seed)
    out = []
    for k, (w, h) in enumerate(((200, 136), (176, 144), (66, 50), (320, 200), (1920, 1080))):
        for rep in range(4):
            wcu, hcu = (w // 2 + 7) // 8, (h // 2 + 7) // 8
            ncu = wcu * hcu
            qg = 16 if rep != 2 else 8
            isP = rep % 2
            intra = rng.integers(1, 4000, size=ncu).astype(np.int32)
            inter = np.minimum(rng.integers(0, 5000, size=ncu), 0x3FFF)
            lists = rng.integers(1, 4, size=ncu) if not isP else np.ones(ncu, np.int64)
            lowres = (inter | (lists << 14)).astype(np.uint16)
            invq = rng.integers(60, 1200, size=ncu).astype(np.int32)
            span = 40 if rep < 3 else 3000
            mvs0 = rng.integers(-span, span + 1, size=(ncu, 2)).astype(np.int32)
            mvs1 = rng.integers(-span, span + 1, size=(ncu, 2)).astype(np.int32)
            mvs0[rng.random(ncu) < 0.3] = 0
            propIn = rng.integers(0, 30000, size=ncu).astype(np.uint16)
            ref0 = rng.integers(0, 65535 if rep == 1 else 20000, size=ncu).astype(np.uint16)
            ref1 = rng.integers(0, 20000, size=ncu).astype(np.uint16)
            nq = ncu * (4 if qg == 8 else 1)
            qpAq = rng.normal(0, 2, size=nq)
            out.append(("cutree %dx%d #%d" % (w, h, rep),
                        (w, h, qg, (30000, 1001) if rep % 2 else (25, 1), [0.04, 0.0333, 1.5, 0.001][rep], isP, 0 if rep == 3 else 1, 1 if rep == 0 else 0,
                         propIn, intra, lowres, invq, mvs0, mvs1, ref0, ref1, 0.6, qpAq)))
    return out
