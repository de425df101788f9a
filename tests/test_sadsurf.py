"""SAD surfaces (include/x265hip.h x265hip_sadsurf_*; x265_amd/csrc/sadsurf.hip): the integer-pel SADs of MotionEstimate::motionEstimate
(reference source/encoder/motion.cpp:246-330, :752-756) as tables built on the device.

CPU tier: the restatement (oracle/x265_oracle_sadsurf.inc, driven through the emulated ABI of tests/support/libx265hip_emul.so) against the
REAL reference's sad<N, N> (oracle/_ref/libx265ref8.so through backends.Ref; the pinned oracle primitive where /root/reference was not
available at build time) — sampled entries of every window — and the windows' legality.  GPU tier: the device surfaces against the
restatement, origin for origin and entry for entry, on pictures with ragged edges, with the reference picture arriving in bands before and
after the surface is attached, and several surfaces on one reference."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
EMUL = os.path.join(ROOT, "tests", "support", "libx265hip_emul.so")
MX, MY = 96, 80            # PicYuv's luma margins at CTU 64 (picyuv.cpp:87-89)
WIN = 16


def _emul(hp):
    if not os.path.exists(EMUL):
        pytest.skip("tests/support/libx265hip_emul.so not built (make -C oracle emul)")
    em = C.CDLL(EMUL)
    for name, (res, args) in hp.PROTOTYPES.items():
        if name.startswith(("x265hip_refpic_", "x265hip_srcpic_", "x265hip_sadsurf_")) or name in ("x265hip_last_error", "x265hip_places", "x265hip_peer_stats"):
            fn = getattr(em, name)
            fn.restype, fn.argtypes = res, args
    return em


def _pictures(w, h, seed, count=2, MX=MX, MY=MY, ctu=64, depth=8):
    """A padded reference picture (as the encoder's recon buffer: margins replicated) and `count` source pictures = the reference moved per
    48 x 40 tile by up to +-20 pixels plus noise."""
    rng = np.random.default_rng(seed)
    from cases import textured_frame
    big = textured_frame(rng, h + 128, w + 128, 8, sigma=2.0)
    ref = big[64:64 + h, 64:64 + w]
    rows = ((h + ctu - 1) // ctu) * ctu + 2 * MY          # picyuv.cpp:95-98
    stride = ((w + 2 * MX + 63) // 64) * 64
    buf = np.zeros((rows, stride), np.uint8)
    buf[:h + 2 * MY, :w + 2 * MX] = np.pad(ref, ((MY, MY), (MX, MX)), mode="edge")
    srcs = []
    for k in range(count):
        s = np.zeros_like(ref)
        for y0 in range(0, h, 40):
            for x0 in range(0, w, 48):
                dy, dx = int(rng.integers(-20, 21)), int(rng.integers(-20, 21))
                y1, x1 = min(y0 + 40, h), min(x0 + 48, w)
                s[y0:y1, x0:x1] = big[64 + y0 + dy:64 + y1 + dy, 64 + x0 + dx:64 + x1 + dx]
        srcs.append(np.ascontiguousarray(np.clip(np.rint(s + rng.normal(0, 2.0, s.shape)), 0, 255).astype(np.uint8)))
    if depth > 8:
        # the same pictures with `depth` bits per sample: scaled up, the new low bits random
        sh = depth - 8
        buf = (buf.astype(np.uint16) << sh) | rng.integers(0, 1 << sh, buf.shape, dtype=np.uint16)
        buf[:h + 2 * MY, :w + 2 * MX] = np.pad(buf[MY:MY + h, MX:MX + w], ((MY, MY), (MX, MX)), mode="edge")
        srcs = [np.ascontiguousarray((s.astype(np.uint16) << sh) | rng.integers(0, 1 << sh, s.shape, dtype=np.uint16)) for s in srcs]
    return buf, stride, rows, srcs


def _read_view(hp, L, ss, w, h):
    """{level: (origins [by][bx][2] int16, tables [by][bx][256] uint32)} out of the chunked host layout."""
    v = C.cast(L.x265hip_sadsurf_get_view(ss), C.POINTER(hp.SadSurfView)).contents
    assert v.ctuRowsReady[0] == (h + 63) // 64, v.ctuRowsReady[0]
    out = {}
    for l in (0, 1, 2, 3):
        lv = v.level[l]
        if not lv.origin:
            continue                   # level not built
        n = 8 << l
        assert (lv.blocksX, lv.blocksY, lv.blocksPerCtuRow) == (w // n, h // n, 64 // n)
        org = np.zeros((lv.blocksY, lv.blocksX, 2), np.int16)
        tab = np.zeros((lv.blocksY, lv.blocksX, WIN * WIN), np.uint32)
        et = np.uint16 if lv.entryBytes == 2 else np.uint32
        for by in range(lv.blocksY):
            r, j = divmod(by, lv.blocksPerCtuRow)
            k0 = j * lv.blocksX
            o = (C.c_int16 * (2 * lv.blocksX)).from_address(lv.origin + r * v.ctuRowPitch + 4 * k0)
            org[by] = np.frombuffer(o, np.int16).reshape(lv.blocksX, 2)
            t = (C.c_char * (lv.blocksX * WIN * WIN * lv.entryBytes)).from_address(lv.table + r * v.ctuRowPitch + k0 * WIN * WIN * lv.entryBytes)
            tab[by] = np.frombuffer(t, et).reshape(lv.blocksX, WIN * WIN)
        sub = None
        if lv.subpel:
            sub = np.zeros((lv.blocksY, lv.blocksX, 49), np.uint32)
            for by in range(lv.blocksY):
                r, j = divmod(by, lv.blocksPerCtuRow)
                k0 = j * lv.blocksX
                t = (C.c_uint32 * (lv.blocksX * 49)).from_address(lv.subpel + r * v.ctuRowPitch + k0 * 49 * 4)
                sub[by] = np.frombuffer(t, np.uint32).reshape(lv.blocksX, 49)
        out[l] = (org, tab, sub)
    return out


def _run(hp, L, w, h, seed, S, lam, bands, attach_after, levels=15, MX=MX, MY=MY, ctu=64, depth=8):
    """Drive one library: reference rows arrive in `bands` (picture rows); source k is attached after band attach_after[k] (-1: before any)."""
    buf, stride, rows, srcs = _pictures(w, h, seed, count=len(attach_after), MX=MX, MY=MY, ctu=ctu, depth=depth)
    rp = L.x265hip_refpic_create(depth, w, h, stride, MX, MY, rows, buf.ctypes.data)
    assert rp, L.x265hip_last_error()
    sps, sss = [], [None] * len(srcs)
    for s in srcs:
        sp = L.x265hip_srcpic_create(depth, w, h)
        assert sp, L.x265hip_last_error()
        assert L.x265hip_srcpic_upload(sp, s.ctypes.data, s.shape[1]) == 0
        sps.append(sp)

    def attach(k):
        sss[k] = L.x265hip_sadsurf_attach_levels(sps[k], rp, S, lam, levels)
        assert sss[k], L.x265hip_last_error()

    for k, a in enumerate(attach_after):
        if a < 0:
            attach(k)
    for i, r in enumerate(bands):
        assert L.x265hip_refpic_rows_final(rp, r) == 0
        for k, a in enumerate(attach_after):
            if a == i:
                attach(k)
    assert L.x265hip_refpic_wait(rp) == 0, L.x265hip_last_error()
    views = [_read_view(hp, L, ss, w, h) for ss in sss]
    for ss in sss:
        L.x265hip_sadsurf_release(ss)
    L.x265hip_refpic_wait(rp)
    L.x265hip_refpic_destroy(rp)
    for sp in sps:
        L.x265hip_srcpic_destroy(sp)
    return views, buf, stride, srcs


@pytest.mark.parametrize("depth", [8, 10])
def test_restatement_entries_are_the_reference_sad_and_windows_are_legal(depth):
    import x265_amd.hipprim as hp
    import backends
    em = _emul(hp)
    try:
        o = backends.Ref(depth)        # the real sad<N, N> (pixel.cpp:40-55) out of oracle/_ref/libx265ref{8,10}.so
    except Exception:
        o = backends.Orc(depth)        # pinned to it by tests/test_oracle_vs_ref.py
    w, h, S = 200, 136, 16
    views, buf, stride, srcs = _run(hp, em, w, h, 5, S, 9 * 20, [64, 128, h], [-1], levels=15 if depth == 8 else 14, depth=depth)
    rng = np.random.default_rng(1)
    for l, (org, tab, _) in views[0].items():
        n = 8 << l
        for by in range(org.shape[0]):
            for bx in range(org.shape[1]):
                ox, oy = int(org[by, bx, 0]), int(org[by, bx, 1])
                x, y = bx * n, by * n
                if l == 0 and (x // 16 >= w // 16 or y // 16 >= h // 16):
                    assert (ox, oy) == (-32768, -32768)          # no 16x16 parent inside the picture: no window
                    continue
                if l == 0:
                    assert (ox, oy) == tuple(int(v) for v in views[0][1][0][by // 2, bx // 2])       # the parent's window
                assert -S <= ox <= S - WIN and -S <= oy <= S - WIN
                assert x + ox >= -MX and x + ox + WIN - 1 + n <= w + MX and y + oy >= -MY and y + oy + WIN - 1 + n <= h + MY
                for k in rng.integers(0, WIN * WIN, 24):
                    j, i = divmod(int(k), WIN)
                    want = o.sad(n, n, srcs[0], (y, x), buf, (MY + y + oy + j, MX + x + ox + i))
                    assert tab[by, bx, k] == want, (l, bx, by, i, j)


GPU_CASES = [
    # w, h, seed, S, lambda20, bands (picture rows final), attach_after per source
    (200, 136, 11, 32, 180, [64, 128, 136], [-1, 1]),
    (416, 240, 12, 32, 0, [240], [-1, 0]),
    (352, 288, 13, 16, 400, [64, 128, 192, 256, 288], [0, 2, 4]),
    (72, 200, 14, 32, 180, [128, 200], [-1]),
    (1280, 720, 15, 32, 180, [192, 448, 720], [1]),
    # --ctu 16: PicYuv pads 48 x 32 only and its buffer ends 32 rows below the picture — the search window reaches beyond both (fuzz seed 59: a GPU
    # memory fault before the staging loads were clamped into the buffer)
    (200, 136, 16, 32, 180, [48, 96, 136], [-1, 1], 48, 32, 16),
    (72, 40, 17, 24, 180, [16, 40], [-1], 48, 32, 16),
    # 16-bit pictures (Main10 / Main12 builds): the v_sad_u16 kernel, u32 surfaces, range <= 16
    (200, 136, 18, 16, 180, [64, 128, 136], [-1, 1], 96, 80, 64, 10),
    (416, 240, 19, 16, 0, [240], [-1, 0], 96, 80, 64, 12),
    (200, 136, 20, 12, 400, [48, 96, 136], [0], 48, 32, 16, 10),
    (1280, 720, 21, 16, 180, [192, 448, 720], [1], 96, 80, 64, 10),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(GPU_CASES)))
def test_device_surfaces_match_restatement(case):
    import x265_amd.hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    em = _emul(hp)
    w, h, seed, S, lam, bands, attach_after, *geom = GPU_CASES[case]
    geom = dict(zip(("MX", "MY", "ctu", "depth"), geom))
    levels = 14 if case == 1 or geom.get("depth", 8) > 8 else 15       # one 8-bit case without the 8x8 windows (the product's default layout); 16-bit never has them
    got, *_ = _run(hp, L, w, h, seed, S, lam, bands, attach_after, levels, **geom)
    want, *_ = _run(hp, em, w, h, seed, S, lam, bands, attach_after, levels, **geom)
    for k in range(len(attach_after)):
        assert sorted(got[k]) == sorted(want[k]) == ([1, 2, 3] if levels == 14 else [0, 1, 2, 3])
        for l in got[k]:
            assert np.array_equal(got[k][l][0], want[k][l][0]), ("origins", k, l)
            assert np.array_equal(got[k][l][1], want[k][l][1]), ("tables", k, l)


# ---- sub-pel SATD tables (round 4) ---------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("depth", [8, 10])
def test_subpel_entries_are_the_reference_filters_and_satd(depth):
    """The restatement's sub-pel entries (oracle/x265_oracle_sadsurf.inc orc_sadsurf_subpel_rows, through the emulated ABI) against what
    MotionEstimate::subpelCompare (reference encoder/motion.cpp:1571-1600) does with the REAL primitives of oracle/_ref: the block itself (integer
    vector) or luma_hpp / luma_vpp / luma_hvpp of it, then satd against the source block."""
    import x265_amd.hipprim as hp
    import backends
    em = _emul(hp)
    try:
        o = backends.Ref(depth)
    except Exception:
        o = backends.Orc(depth)
    w, h, S = 200, 136, 16
    views, buf, stride, srcs = _run(hp, em, w, h, 9, S, 9 * 20, [64, 128, h], [-1], levels=30, depth=depth)
    rng = np.random.default_rng(2)
    checked = 0
    for l in (1, 2, 3):
        org, tab, sub = views[0][l]
        assert sub is not None
        n = 8 << l
        for by in range(org.shape[0]):
            for bx in range(org.shape[1]):
                cx, cy = int(org[by, bx, 0]) + WIN // 2, int(org[by, bx, 1]) + WIN // 2
                for v in rng.choice(49, 4, replace=False):
                    qx, qy = 4 * cx + int(v) % 7 - 3, 4 * cy + int(v) // 7 - 3
                    ry, rx = MY + by * n + (qy >> 2), MX + bx * n + (qx >> 2)
                    fx, fy = qx & 3, qy & 3
                    if not (fx | fy):
                        blk = np.ascontiguousarray(buf[ry:ry + n, rx:rx + n])
                    elif not fy:
                        blk = o.interp("hpp", 0, n, n, buf, (ry, rx), fx)
                    elif not fx:
                        blk = o.interp("vpp", 0, n, n, buf, (ry, rx), fy)
                    else:
                        blk = o.interp("hvpp", 0, n, n, buf, (ry, rx), fx, fy)
                    want = o.satd(n, n, srcs[0], (by * n, bx * n), blk, (0, 0))
                    assert int(sub[by, bx, int(v)]) == want, (l, bx, by, int(v), int(sub[by, bx, int(v)]), want)
                    checked += 1
    assert checked > 400


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(8, 200, 136, 32, [64, 128, 136]), (8, 416, 240, 32, [240]), (10, 200, 136, 16, [64, 136]), (12, 136, 72, 16, [72])])
def test_device_subpel_tables_match_restatement(case):
    """x265_amd/csrc/sadsurf.hip subpel_satd_kernel against the restatement, every entry of every block of levels 1..3; the reference picture in bands and
    whole; two source pictures, one attached late."""
    import x265_amd.hipprim as hp
    depth, w, h, S, bands = case
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    em = _emul(hp)
    got, *_ = _run(hp, L, w, h, 23, S, 200, bands, [-1, len(bands) - 1 if len(bands) > 1 else -1], levels=30, depth=depth)
    want, *_ = _run(hp, em, w, h, 23, S, 200, bands, [-1, len(bands) - 1 if len(bands) > 1 else -1], levels=30, depth=depth)
    n = 0
    for k in range(2):
        for l in (1, 2, 3):
            assert np.array_equal(got[k][l][0], want[k][l][0]) and np.array_equal(got[k][l][1], want[k][l][1]), ("windows", k, l)
            assert got[k][l][2] is not None and want[k][l][2] is not None
            assert np.array_equal(got[k][l][2], want[k][l][2]), ("sub-pel", k, l, np.argwhere(got[k][l][2] != want[k][l][2])[:4])
            n += got[k][l][2].size
    assert n > 1000


@pytest.mark.gpu
def test_device_subpel_tables_first_form_at_8_bit():
    """8- and 10-bit pictures take subpel_satd_kernel_lds (round 6); the first form, subpel_satd_kernel<uint8_t / uint16_t at 10 bit>, stays reachable (X265HIP_SUBPEL_LDS=0, strides that are no
    multiple of 4) and stays pinned: the 8- and 10-bit cases of the test above in a process of their own with the switch set (the library reads it once)."""
    import subprocess
    import sys
    env = dict(os.environ, X265HIP_SUBPEL_LDS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "test_device_subpel_tables_match_restatement and (case0 or case1 or case2)",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "3 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bound_encoder_serves_subpel_candidates_from_the_tables(tmp_path):
    """CPU tier: the motion-search seam with the emulated ABI's sub-pel tables (pictures up to 416 x 240 there): filter + satd pairs of subpelCompare answered
    from the tables, every answer recomputed by the reference's functions (X265HIP_VERIFY), bitstream identical to the unmodified reference's."""
    import re
    import test_places as tp
    ref, emul = tp._need("x265_8bit"), tp._need("x265_emul_8bit")
    from x265_amd.synth import make_clip
    yuv = str(tmp_path / "clip.yuv")
    w, h, frames = 416, 240, 12
    make_clip(yuv, w, h, frames, seed=41)
    want, _ = tp._encode(ref, yuv, w, h, frames, str(tmp_path / "ref.hevc"), {})
    for env in ({"X265HIP_VERIFY": "1"}, {}):
        got, err = tp._encode(emul, yuv, w, h, frames, str(tmp_path / "emul.hevc"), env)
        assert got == want, "bitstreams differ (%s)" % env
        m = re.search(r"sadplanes: (\d+) sub-pel SATDs of the motion search", err)
        assert m and int(m.group(1)) > 2000, err[-800:]


def test_bound_encoder_serves_rectangular_pus_as_sums_of_squares(tmp_path):
    """CPU tier: --rect --amp searches (presets slow / slower): a 32x16 ... 48x64 PU's integer-pel SADs as sums of its 16x16 / 32x32 squares' window entries,
    each sum recomputed by the reference's sad (X265HIP_VERIFY), bitstream identical."""
    import re
    import test_places as tp
    ref, emul = tp._need("x265_8bit"), tp._need("x265_emul_8bit")
    from x265_amd.synth import make_clip
    yuv = str(tmp_path / "clip.yuv")
    w, h, frames = 416, 240, 6
    make_clip(yuv, w, h, frames, seed=43)
    extra = ["--rect", "--amp", "--me", "star", "--subme", "3", "--no-weightp"]
    want, _ = tp._encode(ref, yuv, w, h, frames, str(tmp_path / "ref.hevc"), {}, extra)
    got, err = tp._encode(emul, yuv, w, h, frames, str(tmp_path / "emul.hevc"), {"X265HIP_VERIFY": "1"}, extra)
    assert got == want, "bitstreams differ"
    m = re.search(r"rectangular / asymmetric PUs: (\d+) searches, (\d+) integer-pel SADs served", err)
    assert m and int(m.group(1)) > 500 and int(m.group(2)) > 5000, err[-800:]


# ---- surfaces of several reference pictures in one launch (round 4) --------------------------------------------------------------------------------
def _two_reference_pictures(L, hp, w, h, S, lam, levels, seeds):
    """one source picture per reference picture, both references complete, the surfaces attached back to back; -> (views, launches of this run)"""
    refs, srcs, sss = [], [], []
    before = (C.c_uint64 * 4)()
    L.x265hip_sadsurf_stats(C.byref(before, 0), C.byref(before, 8), C.byref(before, 16), C.byref(before, 24))
    for seed in seeds:
        buf, stride, rows, s = _pictures(w, h, seed, count=1)
        rp = L.x265hip_refpic_create(8, w, h, stride, MX, MY, rows, buf.ctypes.data)
        assert rp, L.x265hip_last_error()
        assert L.x265hip_refpic_rows_final(rp, h) == 0 and L.x265hip_refpic_wait(rp) == 0
        sp = L.x265hip_srcpic_create(8, w, h)
        assert sp and L.x265hip_srcpic_upload(sp, s[0].ctypes.data, s[0].shape[1]) == 0
        refs.append((rp, buf)); srcs.append((sp, s[0]))
    for (rp, _), (sp, _) in zip(refs, srcs):
        ss = L.x265hip_sadsurf_attach_levels(sp, rp, S, lam, levels)
        assert ss, L.x265hip_last_error()
        sss.append(ss)
    for rp, _ in refs:
        assert L.x265hip_refpic_wait(rp) == 0, L.x265hip_last_error()
    views = [_read_view(hp, L, ss, w, h) for ss in sss]
    after = (C.c_uint64 * 4)()
    L.x265hip_sadsurf_stats(C.byref(after, 0), C.byref(after, 8), C.byref(after, 16), C.byref(after, 24))
    for ss in sss:
        L.x265hip_sadsurf_release(ss)
    for rp, _ in refs:
        L.x265hip_refpic_wait(rp)
        L.x265hip_refpic_destroy(rp)
    for sp, _ in srcs:
        L.x265hip_srcpic_destroy(sp)
    return views, int(after[2] - before[2])


@pytest.mark.gpu
def test_device_surfaces_of_three_reference_pictures_share_a_launch():
    """Attach jobs that reach the worker within X265HIP_SADSURF_GATHER_US are built by ONE launch whichever reference picture each follows (a job carries
    its own reference pointer): same origins and tables as the restatement builds one surface at a time.  In a process of its own: the window is read
    once per process."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "three-references"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, X265HIP_SADSURF_GATHER_US="300000"))
    assert r.returncode == 0, (r.stdout[-600:], r.stderr[-1200:])
    assert "launches 1" in r.stdout, r.stdout[-400:]


if __name__ == "__main__" and sys.argv[1:] == ["three-references"]:
    import x265_amd.hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    em = _emul(hp)
    w, h, S, lam, levels, seeds = 416, 240, 32, 180, 15, (31, 32, 33)
    got, launches = _two_reference_pictures(L, hp, w, h, S, lam, levels, seeds)
    want, _ = _two_reference_pictures(em, hp, w, h, S, lam, levels, seeds)
    for k in range(len(seeds)):
        assert sorted(got[k]) == sorted(want[k]) == [0, 1, 2, 3]
        for l in got[k]:
            assert np.array_equal(got[k][l][0], want[k][l][0]), ("origins", k, l)
            assert np.array_equal(got[k][l][1], want[k][l][1]), ("tables", k, l)
    print("three surfaces on three reference pictures: launches %d" % launches)
