"""GPU parity of the reference-picture mirrors (x265hip_refpic_*, x265_amd/csrc/refpic.hip): rows published band by band, the 15 fractional
planes that arrive in host memory must equal the oracle's luma_hpp / luma_vpp / luma_hvpp at every pixel of the rows the mirror says are
ready — and nothing may be reported ready whose 8-tap support was not final yet (checked by poisoning the unpublished rows)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_planes(o, buf, depth):
    """{phase: plane} over the whole buffer except its outermost 4 rows / columns, by the oracle's block filters (64x64 tiles)."""
    R, S = buf.shape
    out = {}
    for yf in range(4):
        for xf in range(4):
            if not (xf | yf):
                continue
            pl = np.zeros_like(buf)
            for y0 in range(4, R - 4, 64):
                for x0 in range(4, S - 4, 64):
                    bh, bw = min(64, R - 4 - y0), min(64, S - 4 - x0)
                    if not yf:
                        blk = o.interp("hpp", 0, bw, bh, buf, (y0, x0), xf)
                    elif not xf:
                        blk = o.interp("vpp", 0, bw, bh, buf, (y0, x0), yf)
                    else:
                        blk = o.interp("hvpp", 0, bw, bh, buf, (y0, x0), xf, yf)
                    pl[y0:y0 + bh, x0:x0 + bw] = blk
            out[yf * 4 + xf] = pl
    return out


@pytest.mark.parametrize("depth,w,h", [(8, 192, 136), (10, 136, 72), (8, 328, 200)])
def test_rows_arrive_band_by_band_and_match_the_filters(depth, w, h):
    from backends import Orc
    from x265_amd import hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    mx, my, cu = 96, 80, 64
    S, R = w + 2 * mx, ((h + cu - 1) // cu) * cu + 2 * my
    rng = np.random.default_rng(5 + depth + w)
    dt = hp.pix_dtype(depth)
    final = rng.integers(0, 1 << depth, size=(R, S)).astype(dt)
    want = _oracle_planes(Orc(depth), np.ascontiguousarray(final[:h + 2 * my]), depth)
    # the encoder's buffer: rows below the published ones hold garbage that must never leak into a published phase row
    buf = np.full((R, S), (1 << depth) - 1, dt)
    rp = L.x265hip_refpic_create(depth, w, h, S, mx, my, R, buf.ctypes.data)
    assert rp, L.x265hip_last_error()
    try:
        planes = {p: np.ctypeslib.as_array(C.cast(L.x265hip_refpic_plane(rp, p), C.POINTER(C.c_uint8 if depth == 8 else C.c_uint16)), shape=(R, S)) for p in range(1, 16)}
        assert L.x265hip_refpic_rows_ready(rp) < -1000
        nrows = (h + cu - 1) // cu
        for r in range(nrows):
            rows_final = h if r == nrows - 1 else (r + 1) * cu
            upto = my + rows_final + (my if r == nrows - 1 else 0)
            buf[:upto] = final[:upto]
            hp.check(L.x265hip_refpic_rows_final(rp, rows_final))
            hp.check(L.x265hip_refpic_wait(rp))
            ready = L.x265hip_refpic_rows_ready(rp)
            assert ready == upto - 4 - my
            for p in range(1, 16):
                got = planes[p][4:my + ready, 4:S - 4]
                assert np.array_equal(got, want[p][4:my + ready, 4:S - 4]), (p, r)
        assert L.x265hip_refpic_rows_ready(rp) == h + my - 4
        # a new picture in the same buffer: nothing valid until its rows are published, then the new picture's planes
        hp.check(L.x265hip_refpic_reset(rp))
        assert L.x265hip_refpic_rows_ready(rp) < -1000
        final2 = np.ascontiguousarray(final[::-1])
        buf[:] = final2
        hp.check(L.x265hip_refpic_rows_final(rp, h))
        hp.check(L.x265hip_refpic_wait(rp))
        want2 = _oracle_planes(Orc(depth), np.ascontiguousarray(final2[:h + 2 * my]), depth)
        ready = L.x265hip_refpic_rows_ready(rp)
        assert ready == h + my - 4
        for p in (1, 4, 5, 15):
            assert np.array_equal(planes[p][4:my + ready, 4:S - 4], want2[p][4:my + ready, 4:S - 4]), p
    finally:
        L.x265hip_refpic_destroy(rp)


@pytest.mark.parametrize("depth,w,h", [(8, 200, 136), (10, 72, 64), (12, 328, 200)])
def test_source_energy_planes_match_the_oracle(depth, w, h):
    """x265hip_source_energy: the source half of psyCost_pp (pixel.cpp:726-757) for every aligned 8x8 and 4x4 block of a plane — sa8d_8x8 / satd_4x4
    against a zero block minus a quarter of the block sum — against the oracle's restatements, on random and on extreme pictures."""
    from backends import Orc
    from x265_amd import hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    o = Orc(depth)
    rng = np.random.default_rng(11 + depth)
    dt = hp.pix_dtype(depth)
    S = w + 24
    for kind in ("random", "max", "checker"):
        if kind == "random":
            pic = rng.integers(0, 1 << depth, size=(h + 3, S)).astype(dt)
        elif kind == "max":
            pic = np.full((h + 3, S), (1 << depth) - 1, dt)
        else:
            pic = (((np.add.outer(np.arange(h + 3), np.arange(S)) & 1) * ((1 << depth) - 1))).astype(dt)
        bw, bh = w // 8, h // 8
        e8, e4 = np.zeros((bh, bw), np.int32), np.zeros((bh * 2, bw * 2), np.int32)
        hp.check(L.x265hip_source_energy(depth, pic.ctypes.data, S, w, h, e8.ctypes.data, e4.ctypes.data))
        zero = np.zeros((8, 8), dt)
        for by in range(bh):
            for bx in range(bw):
                blk = np.ascontiguousarray(pic[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8])
                assert e8[by, bx] == o.sa8d(8, blk, (0, 0), zero, (0, 0)) - (int(blk.sum()) >> 2), (kind, bx, by)
                for q in range(4):
                    b4 = np.ascontiguousarray(blk[(q >> 1) * 4:(q >> 1) * 4 + 4, (q & 1) * 4:(q & 1) * 4 + 4])
                    assert e4[by * 2 + (q >> 1), bx * 2 + (q & 1)] == o.satd(4, 4, b4, (0, 0), zero, (0, 0)) - (int(b4.sum()) >> 2), (kind, bx, by, q)
