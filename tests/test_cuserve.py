"""CU residual quad-tree jobs (include/x265hip.h x265hip_cuserve_*; x265_amd/csrc/cuserve.hip): the transform arithmetic of an inter CU —
Quant::transformNxN (reference source/common/quant.cpp:397-470, with signBitHidingHDQ :246-395) and Quant::invtransformNxN (:543-603) for
every transform unit Search::estimateResidualQT (encoder/search.cpp:3178-3560) may try — as ONE job handed to the device.

CPU tier: the restatement (oracle/x265_oracle_rqt.c) against the REAL Quant class of the reference (oracle/_ref/libx265ref*.so, ref_shim.cpp
ref_transform_nxn / ref_invtransform_nxn): sizes 8-32, Y / Cb / Cr QPs, I and P rounding offsets, sign hiding on and off, residuals from
noise-like to nearly empty; the job built from it through the emulated ABI (tests/support/libx265hip_emul.so).
GPU tier: the device's job against the restatement's, unit for unit (numSig, levels, reconstructed residual, both distortions), CU 16 / 32 /
64, 8 / 10 / 12 bit, with and without chroma, one or two transform sizes, resident-server and launch-per-job hand-off."""
import ctypes as C
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po   # noqa: E402  (checker only)

EMUL = os.path.join(ROOT, "tests", "support", "libx265hip_emul.so")
vp, i32, u32 = C.c_void_p, C.c_int, C.c_uint32


def _orc():
    L = po.oracle()
    L.orc_transform_nxn.restype, L.orc_transform_nxn.argtypes = u32, [vp, C.c_ssize_t, vp, vp, i32, i32, i32, i32, i32, i32, i32]
    L.orc_invtransform_nxn.restype, L.orc_invtransform_nxn.argtypes = None, [vp, C.c_ssize_t, vp, i32, i32, i32, i32, u32]
    for sfx in ("8", "16"):
        f = getattr(L, "orc_cujob_run_" + sfx)
        f.restype, f.argtypes = i32, [vp, vp, vp, vp, vp, u32]
    L.orc_saojob_run_8.restype, L.orc_saojob_run_8.argtypes = i32, [vp, vp, vp, vp, u32]
    L.orc_intrajob_run.restype, L.orc_intrajob_run.argtypes = i32, [vp, vp, vp, vp, u32]
    for sfx in ("8", "16"):
        f = getattr(L, "orc_intra_filter_" + sfx)
        f.restype, f.argtypes = None, [i32, vp, vp]
    return L


def _ref(depth):
    if not po.ref_available(depth):
        pytest.skip("oracle/_ref/libx265ref%d.so not built (make -C oracle ref)" % depth)
    R = po.ref(depth)
    R.ref_transform_nxn.restype, R.ref_transform_nxn.argtypes = u32, [vp, u32, vp, vp, i32, i32, i32, i32, i32]
    R.ref_invtransform_nxn.restype, R.ref_invtransform_nxn.argtypes = None, [vp, u32, vp, i32, i32, i32, u32]
    R.ref_qp_param.restype, R.ref_qp_param.argtypes = None, [i32, vp]
    R.ref_dct.restype, R.ref_dct.argtypes = None, [i32, vp, vp, C.c_ssize_t]
    return R


def _residual(rng, n, depth, kind):
    pmax = (1 << depth) - 1
    if kind == 0:      # TestBench's distribution (mbdstharness.cpp:63-77): uniform over the whole range
        r = rng.integers(-pmax, pmax + 1, (n, n))
    elif kind == 1:    # what a good prediction leaves: small noise
        r = np.rint(rng.normal(0, 3 << (depth - 8), (n, n)))
    elif kind == 2:    # smooth ramp + a few spikes: a handful of low-frequency levels
        yy, xx = np.mgrid[0:n, 0:n]
        r = ((xx - n / 2) * rng.uniform(-2, 2) + (yy - n / 2) * rng.uniform(-2, 2)) * (1 << (depth - 8))
        k = rng.integers(0, n, (4, 2))
        r[k[:, 0], k[:, 1]] += rng.integers(-40, 41, 4) * (1 << (depth - 8))
    else:              # a single step edge
        r = np.zeros((n, n))
        r[:, rng.integers(1, n):] = rng.integers(-30, 31) * (1 << (depth - 8))
    return np.ascontiguousarray(np.clip(r, -pmax, pmax).astype(np.int16))


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_transform_restatement_matches_the_reference_quant_class(depth):
    """orc_transform_nxn / orc_invtransform_nxn == Quant::transformNxN / ::invtransformNxN of the reference, bit for bit."""
    O, R = _orc(), _ref(depth)
    rng = np.random.default_rng(100 + depth)
    bdOff = 6 * (depth - 8)
    n_cases = hidden = 0
    for log2n in (3, 4, 5):
        n = 1 << log2n
        for ttype in (0, 1, 2):
            for qp in (bdOff + 4, bdOff + 17, bdOff + 22, bdOff + 27, bdOff + 32, bdOff + 38, bdOff + 45, bdOff + 51):
                qparam = (C.c_int32 * 4)()
                R.ref_qp_param(qp, qparam)
                rem, per, qs, dqs = qparam[0], qparam[1], qparam[2], qparam[3]
                assert (rem, per) == (qp % 6, qp // 6)
                for sliceI in (0, 1):
                    for signHide in (0, 1):
                        for kind in (0, 1, 2, 3):
                            stride = n + 8 * int(rng.integers(0, 3))
                            resi = np.zeros((n, stride), np.int16)
                            resi[:, :n] = _residual(rng, n, depth, kind)
                            c_o, c_r = np.zeros(n * n, np.int16), np.zeros(n * n, np.int16)
                            d_o, d_r = np.zeros(n * n, np.int16), np.zeros(n * n, np.int16)
                            ns_r = R.ref_transform_nxn(resi.ctypes.data, stride, c_r.ctypes.data, d_r.ctypes.data, log2n, ttype, qp, sliceI, signHide)
                            ns_o = O.orc_transform_nxn(resi.ctypes.data, stride, c_o.ctypes.data, d_o.ctypes.data, log2n, depth, rem, per, qs, 171 if sliceI else 85,
                                                       signHide)
                            label = (log2n, ttype, qp, sliceI, signHide, kind)
                            assert ns_o == ns_r and np.array_equal(c_o, c_r) and np.array_equal(d_o, d_r), label
                            assert ns_o == int(np.count_nonzero(c_o)), label
                            if signHide:
                                c_plain = np.zeros(n * n, np.int16)
                                O.orc_transform_nxn(resi.ctypes.data, stride, c_plain.ctypes.data, d_o.ctypes.data, log2n, depth, rem, per, qs, 171 if sliceI else 85, 0)
                                hidden += int(not np.array_equal(c_plain, c_o))
                            if ns_r:
                                b_o, b_r = np.full((n, stride), 77, np.int16), np.full((n, stride), 77, np.int16)
                                R.ref_invtransform_nxn(b_r.ctypes.data, stride, c_r.ctypes.data, log2n, ttype, qp, ns_r)
                                O.orc_invtransform_nxn(b_o.ctypes.data, stride, c_o.ctypes.data, log2n, depth, per, dqs, ns_o)
                                assert np.array_equal(b_o, b_r), label
                            n_cases += 1
    assert n_cases == 3 * 3 * 8 * 2 * 2 * 4
    assert hidden > 100, hidden          # the sign-hiding branch really moved levels


# ---- jobs ------------------------------------------------------------------------------------------------------------------------------------

def _job_header(hp, log2cu, tr_max, tr_min, chroma, depth, qps, sliceI, signHide, coef=0, source_dct=0):
    j = hp.CuJob()
    j.coefMode, j.sourceDct = coef, source_dct
    j.log2CUSize, j.log2TrMax, j.log2TrMin, j.chroma, j.bitDepth = log2cu, tr_max, tr_min, chroma, depth
    j.quantOffset, j.signHide = (171 if sliceI else 85), signHide
    quant = [26214, 23302, 20560, 18396, 16384, 14564]      # the flat matrix entries (scalinglist.cpp:129-130; checked against the reference above)
    dequant = [40, 45, 51, 57, 64, 72]
    for p in range(3):
        j.qpRem[p], j.qpPer[p] = qps[p] % 6, qps[p] // 6
        j.quantScale[p], j.dequantScale[p] = quant[qps[p] % 6], dequant[qps[p] % 6]
    return j


def _job_pixels(rng, log2cu, chroma, depth, kind):
    """source + prediction, Y then Cb, Cr, as the job's pixel block"""
    N = 1 << log2cu
    pmax = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    planes = [(N, N)] + ([(N // 2, N // 2)] * 2 if chroma else [])
    src, prd = [], []
    for (h, w) in planes:
        base = rng.integers(0, pmax + 1, (h, w)) if kind == 0 else np.clip(np.rint(rng.normal(pmax / 2, pmax / 6, (h, w))), 0, pmax)
        if kind == 0:
            p = rng.integers(0, pmax + 1, (h, w))
        else:
            p = np.clip(base + np.rint(rng.normal(0, (2 + 3 * kind) * (1 << (depth - 8)), (h, w))), 0, pmax)
        src.append(base.astype(dt).ravel())
        prd.append(p.astype(dt).ravel())
    return np.ascontiguousarray(np.concatenate(src + prd))


def _layout(hp, j):
    """[(s, plane, tx, ty, unitIndex, elemOffset, n)] exactly as include/x265hip.h's inline helpers lay a job out"""
    hi, lo = min(5, j.log2TrMax, j.log2CUSize), max(4, j.log2TrMin)
    planes = 3 if j.chroma else 1
    N2 = 1 << (2 * j.log2CUSize)
    per_level = N2 + N2 // 2 if j.chroma else N2
    out, base = [], 0
    for s in range(hi, lo - 1, -1):
        per = 1 << (j.log2CUSize - s)
        for plane in range(planes):
            n = 1 << (s - 1 if plane else s)
            for ty in range(per):
                for tx in range(per):
                    t = ty * per + tx
                    eo = (hi - s) * per_level + (t * n * n if plane == 0 else N2 + (N2 // 4 if plane == 2 else 0) + t * n * n)
                    out.append((s, plane, tx, ty, base + plane * per * per + t, eo, n))
        base += planes * per * per
    return out


def _oracle_job(hp, O, j, pix):
    units = (hp.CuJobUnit * hp.CUJOB_MAX_UNITS)()
    levels, resi = np.zeros(hp.CUJOB_MAX_ELEMS, np.int16), np.zeros(hp.CUJOB_MAX_ELEMS, np.int16)
    fn = O.orc_cujob_run_8 if j.bitDepth == 8 else O.orc_cujob_run_16
    done = fn(C.byref(j), pix.ctypes.data, C.byref(units), levels.ctypes.data, resi.ctypes.data, 1)
    return done, units, levels, resi


def _run_on(hp, L, cs, slot, j, pix, timeout=20.0):
    job, pixels, units, levels, resi = vp(), vp(), vp(), vp(), vp()
    hp.check(L.x265hip_cuserve_slot(cs, slot, C.byref(job), C.byref(pixels), C.byref(units), C.byref(levels), C.byref(resi)))
    C.memmove(job, C.byref(j), C.sizeof(j))
    C.memmove(pixels, pix.ctypes.data, pix.nbytes)
    seq = u32()
    hp.check(L.x265hip_cuserve_submit(cs, slot, C.byref(seq)))
    lay = _layout(hp, j)
    un = C.cast(units, C.POINTER(hp.CuJobUnit))
    t0 = time.time()
    while any(un[k[4]].ready != seq.value or un[k[4]].readyInv != seq.value for k in lay):
        pk = L.x265hip_cuserve_poke(cs, slot)              # 0 / 1: wait on (1: the server is on its way); negative: failed
        if pk < 0:
            hp.check(pk)
        assert time.time() - t0 < timeout, "job not finished after %.0f s" % timeout
    lv = np.ctypeslib.as_array(C.cast(levels, C.POINTER(C.c_int16)), (hp.CUJOB_MAX_ELEMS,)).copy()
    rs = np.ctypeslib.as_array(C.cast(resi, C.POINTER(C.c_int16)), (hp.CUJOB_MAX_ELEMS,)).copy()
    return [(un[k[4]].numSig, un[k[4]].zeroDist, un[k[4]].codedDist, un[k[4]].codedEnergy) for k in lay], lv, rs


def _compare(hp, j, got, want_units, want_levels, want_resi, label):
    heads, lv, rs = got
    n_units = coded = 0
    for k, (s, plane, tx, ty, ui, eo, n) in enumerate(_layout(hp, j)):
        w = want_units[ui]
        assert heads[k][0] == w.numSig, (label, "numSig", s, plane, tx, ty, heads[k][0], w.numSig)
        assert heads[k][1] == w.zeroDist, (label, "zeroDist", s, plane, tx, ty)
        assert np.array_equal(lv[eo:eo + n * n], want_levels[eo:eo + n * n]), (label, "levels", s, plane, tx, ty)
        if j.coefMode:
            # the `levels` block holds the residual's transform coefficients; a luma unit's `resi` block the source block's (sourceDct)
            assert w.numSig == 0
            if j.sourceDct and plane == 0:
                assert np.array_equal(rs[eo:eo + n * n], want_resi[eo:eo + n * n]), (label, "source transform", s, plane, tx, ty)
                coded += 1
        elif w.numSig:
            assert heads[k][2] == w.codedDist, (label, "codedDist", s, plane, tx, ty)
            assert heads[k][3] == w.codedEnergy, (label, "codedEnergy", s, plane, tx, ty, heads[k][3], w.codedEnergy)
            assert np.array_equal(rs[eo:eo + n * n], want_resi[eo:eo + n * n]), (label, "resi", s, plane, tx, ty)
            coded += 1
        n_units += 1
    return n_units, coded


JOB_SHAPES = [  # log2CU, trMax, trMin, chroma
    (5, 5, 5, 1), (6, 5, 5, 1), (4, 5, 4, 1), (5, 5, 4, 1), (6, 5, 4, 1), (5, 5, 2, 0), (6, 5, 5, 0), (4, 4, 2, 1), (6, 4, 4, 1)]


def _cases(depth, seed):
    rng = np.random.default_rng(seed)
    bd = 6 * (depth - 8)
    for shape in JOB_SHAPES:
        for kind in (0, 1, 2):
            for rep in range(2):
                qy = int(rng.integers(bd + 10, bd + 46))
                qps = (qy, max(0, qy - int(rng.integers(0, 7))), max(0, qy - int(rng.integers(0, 7))))
                yield shape, kind, qps, int(rng.integers(0, 2)), int(rng.integers(0, 4) > 0), rng


def test_emulated_jobs_are_the_restatement():
    """tests/support/libx265hip_emul.so's cuserve (what the CPU-tier encodes of test_x265_dropin.py / test_encoder_fuzz.py run on) through the same
    submit / poll protocol as the device."""
    from x265_amd import hipprim as hp
    if not os.path.exists(EMUL):
        pytest.skip("tests/support/libx265hip_emul.so not built (make -C oracle emul)")
    em = C.CDLL(EMUL)
    for name, (res, args) in hp.PROTOTYPES.items():
        if name.startswith("x265hip_cuserve_"):
            fn = getattr(em, name)
            fn.restype, fn.argtypes = res, args
    em.x265hip_last_error.restype = C.c_char_p
    O = _orc()
    cs = vp()
    assert em.x265hip_cuserve_open(3, 0, C.byref(cs)) == 0
    total = 0
    for depth in (8, 10):
        for shape, kind, qps, sliceI, signHide, rng in _cases(depth, 7 + depth):
            j = _job_header(hp, *shape, depth, qps, sliceI, signHide, coef=int(total % 4 == 3), source_dct=int(total % 8 == 3))
            pix = _job_pixels(rng, shape[0], shape[3], depth, kind)
            done, wu, wl, wr = _oracle_job(hp, O, j, pix)
            n, coded = _compare(hp, j, _run_on(_Chk, em, cs, total % 3, j, pix), wu, wl, wr, (depth, shape, kind, qps))
            assert n == done
            total += 1
    assert em.x265hip_cuserve_close(cs) == 0
    assert total == 2 * len(JOB_SHAPES) * 6


@pytest.mark.parametrize("bits", [8, 10])
def test_bound_encoder_on_emulated_jobs_serves_and_stays_byte_identical(tmp_path, bits):
    """The binding end to end on the CPU tier, at both pixel widths: jobs served, distortions answered out of them (Main10 / Main12: through cu[].sse_ss,
    which x265_setup_primitives aliases the luma sse_pp to after the binding's table set-up — a wrapper on sse_pp alone is silently lost there, and the
    reconstructions the binding no longer computes were then read stale: found by tests/test_encoder_fuzz.py seed 203 in round 4), the body's dead
    sub_ps / add_ps calls put off, jobs submitted ahead at the merge candidate's skip evaluation adopted — and the reference's bytes, also with
    X265HIP_VERIFY recomputing every served value."""
    import re, subprocess, sys
    sys.path.insert(0, ROOT)
    ref, emul = os.path.join(ROOT, "oracle", "_ref", "x265_%dbit" % bits), os.path.join(ROOT, "oracle", "_ref", "x265_emul_%dbit" % bits)
    if not (os.path.exists(ref) and os.path.exists(emul)):
        pytest.skip("oracle/_ref encoders not built (make -C oracle ref emul)")
    from x265_amd.synth import make_clip
    yuv = str(tmp_path / "clip.yuv")
    make_clip(yuv, 384, 256, 6, seed=77, depth=bits)
    args = ["--input", yuv, "--input-res", "384x256", "--input-depth", str(bits), "--fps", "30", "--frames", "6", "--preset", "medium", "--hash", "1", "--pools", "4", "-F", "2"]
    want = str(tmp_path / "ref.hevc")
    assert subprocess.run([ref] + args + ["-o", want], capture_output=True, timeout=600).returncode == 0
    for verify in (False, True):
        got = str(tmp_path / ("emul%d.hevc" % verify))
        env = dict(os.environ, X265HIP_VERBOSE="1")
        env.pop("X265HIP", None)
        if verify:
            env["X265HIP_VERIFY"] = "1"
        r = subprocess.run([emul] + args + ["-o", got], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-800:]
        assert open(got, "rb").read() == open(want, "rb").read(), "bitstream differs (bits %d, verify %d)" % (bits, verify)
        m = re.search(r"cuserve: (\d+) sse_pp and (\d+) psy-cost \(source, reconstruction\) answers", r.stderr)
        assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0, r.stderr[-1200:]
        m = re.search(r"cuserve: (\d+) jobs left ahead of their scope.*?; (\d+) of them were the job", r.stderr)
        assert m and int(m.group(1)) > 0 and m.group(1) == m.group(2), r.stderr[-1200:]
        if not verify:
            m = re.search(r"cuserve: (\d+) sub_ps and (\d+) add_ps calls of those CUs put off", r.stderr)
            assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0, r.stderr[-1200:]


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_coefficient_mode_restatement_is_the_reference_transform(depth):
    """coefMode jobs (RDOQ presets: the host quantises): a unit's `levels` block == the reference's cu[].dct of (source - prediction), a luma unit's
    `resi` block == cu[].dct of the source block as int16 — what Quant::transformNxN leaves in m_resiDctCoeff / m_fencDctCoeff (quant.cpp:432, :436-442)"""
    from x265_amd import hipprim as hp
    O, R = _orc(), _ref(depth)
    rng = np.random.default_rng(900 + depth)
    checked = 0
    for shape in ((5, 5, 4, 1), (6, 5, 5, 1), (4, 4, 4, 1)):
        for kind in (0, 1, 2):
            j = _job_header(hp, *shape, depth, (30 + 6 * (depth - 8),) * 3, 0, 1, coef=1, source_dct=1)
            pix = _job_pixels(rng, shape[0], shape[3], depth, kind)
            done, wu, wl, wr = _oracle_job(hp, O, j, pix)
            N = 1 << shape[0]
            planes = [N * N] + ([N * N // 4] * 2 if shape[3] else [])
            half = sum(planes)
            for (s, plane, tx, ty, ui, eo, n) in _layout(hp, j):
                pw = N if plane == 0 else N // 2
                base = sum(planes[:plane])
                src = pix[base:base + pw * pw].reshape(pw, pw)[ty * n:(ty + 1) * n, tx * n:(tx + 1) * n].astype(np.int32)
                prd = pix[half + base:half + base + pw * pw].reshape(pw, pw)[ty * n:(ty + 1) * n, tx * n:(tx + 1) * n].astype(np.int32)
                resi = np.ascontiguousarray((src - prd).astype(np.int16))
                want = np.zeros(n * n, np.int16)
                R.ref_dct(int(np.log2(n)) - 2, resi.ctypes.data, want.ctypes.data, n)
                assert np.array_equal(wl[eo:eo + n * n], want), (depth, shape, kind, s, plane, tx, ty)
                assert wu[ui].numSig == 0 and wu[ui].zeroDist == int(((src - prd) ** 2).sum())
                if plane == 0:
                    s16 = np.ascontiguousarray(src.astype(np.int16))
                    R.ref_dct(int(np.log2(n)) - 2, s16.ctypes.data, want.ctypes.data, n)
                    assert np.array_equal(wr[eo:eo + n * n], want), (depth, shape, kind, "source", s, tx, ty)
                checked += 1
    assert checked >= 90


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_device_coefficient_jobs_match_the_restatement(mode):
    """coefMode jobs on the MI355X (what the RDOQ presets hand over: the MFMA transforms of every unit's residual, and of the luma units' source blocks for
    psy-rdoq) against the restatement; 8 / 10 / 12 bit, with and without sourceDct, slots reused, mixed with ordinary jobs on the same service"""
    from x265_amd import hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    O = _orc()
    cs = vp()
    hp.check(L.x265hip_cuserve_open(4, mode, C.byref(cs)))
    try:
        total = units = sources = 0
        for depth in (8, 10, 12):
            for shape, kind, qps, sliceI, signHide, rng in _cases(depth, 60 + depth):
                variant = total % 3                               # 0: coefficients + source transform, 1: coefficients only, 2: an ordinary job in between
                j = _job_header(hp, *shape, depth, qps, sliceI, signHide, coef=int(variant != 2), source_dct=int(variant == 0))
                pix = _job_pixels(rng, shape[0], shape[3], depth, kind)
                done, wu, wl, wr = _oracle_job(hp, O, j, pix)
                n, c = _compare(hp, j, _run_on(hp, L, cs, total % 4, j, pix), wu, wl, wr, (mode, depth, shape, kind, qps, variant))
                assert n == done
                total += 1; units += n
                if variant == 0:
                    sources += c
        assert units > 1000 and sources > 100
    finally:
        hp.check(L.x265hip_cuserve_close(cs))


# ---- SAO statistics jobs (x265hip_saojob) ---------------------------------------------------------------------------------------------------------

def _sao_job(hp, rng, planes, full):
    """a job as SAO::calcSaoStatsCTU's seam would build it: plane sizes up to 64x64 (partial CTUs at the picture's edges), rectangles as the reference's
    start / end expressions produce them (left / above unavailable: start 1; right / bottom edge or the not-yet-deblocked margin: end shortened)"""
    j = hp.SaoCtuJob()
    j.bitDepth, j.planes, j.eo23 = 8, planes, int(rng.integers(0, 4) > 0)
    blocks = []
    for b in range(planes):
        luma = b == 0 and planes != 2
        w = (64 if luma else 32) if full else int(rng.integers(1, (64 if luma else 32) + 1))
        h = (64 if luma else 32) if full else int(rng.integers(1, (64 if luma else 32) + 1))
        po = 0 if luma else 2
        right, bottom, left, above = (int(rng.integers(0, 3) == 0) for _ in range(4))
        j.plane[b].w, j.plane[b].h = w, h
        rects = [
            (0, 0, w if right else w - 5 + po, h if bottom else h - 4 + po),
            (left, 0, w - 1 if right else w - 5 + po, h - 4 + po),
            (0, above, w if right else w - 5 + po, h - 1 if bottom else h - 4 + po),
            (left, above, w - 1 if right else w - 5 + po, h - 1 if bottom else h - 4 + po),
            (left, above, w - 1 if right else w - 5 + po, h - 1 if bottom else h - 4 + po)]
        for c, (x0, y0, x1, y1) in enumerate(rects):
            if x1 <= x0 or y1 <= y0:
                x0 = y0 = x1 = y1 = 0
            j.plane[b].x0[c], j.plane[b].y0[c], j.plane[b].x1[c], j.plane[b].y1[c] = x0, y0, x1, y1
        kind = int(rng.integers(0, 3))
        base = rng.integers(0, 256, (h + 1, w + 1)) if kind == 0 else np.clip(np.rint(rng.normal(128, 30, (h + 1, w + 1))), 0, 255)
        if kind == 2:
            base = (base // 16) * 16                                      # flat steps: many equal neighbours (the zero-sign category)
        src = np.clip(base[1:, 1:] + np.rint(rng.normal(0, 4, (h, w))), 0, 255)
        blocks += [base.astype(np.uint8).ravel(), src.astype(np.uint8).ravel()]
    return j, np.ascontiguousarray(np.concatenate(blocks))


def _oracle_sao(hp, O, j, pix):
    units = (hp.CuJobUnit * hp.CUJOB_MAX_UNITS)()
    out = np.zeros(2 * hp.SAOJOB_STATS_ENTRIES, np.int32)
    assert O.orc_saojob_run_8(C.byref(j), pix.ctypes.data, C.byref(units), out.ctypes.data, 1) == j.planes
    return out


def _run_sao_on(hp, L, cs, slot, j, pix, timeout=20.0):
    job, pixels, units, levels, resi = vp(), vp(), vp(), vp(), vp()
    hp.check(L.x265hip_cuserve_slot(cs, slot, C.byref(job), C.byref(pixels), C.byref(units), C.byref(levels), C.byref(resi)))
    C.memmove(pixels, pix.ctypes.data, pix.nbytes)
    seq = u32()
    hp.check(L.x265hip_cuserve_submit_sao(cs, slot, C.byref(j), C.byref(seq)))
    un = C.cast(units, C.POINTER(hp.CuJobUnit))
    t0 = time.time()
    while any(un[b].ready != seq.value for b in range(j.planes)):
        pk = L.x265hip_cuserve_poke(cs, slot)
        if pk < 0:
            hp.check(pk)
        assert time.time() - t0 < timeout, "SAO job not finished after %.0f s" % timeout
    return np.ctypeslib.as_array(C.cast(levels, C.POINTER(C.c_int32)), (2 * hp.SAOJOB_STATS_ENTRIES,)).copy()


def _same_sao(j, got, want, label):
    n = 160 * j.planes
    E = 3 * 5 * 32
    assert np.array_equal(got[:n], want[:n]), (label, "sums", np.nonzero(got[:n] != want[:n])[0][:8])
    assert np.array_equal(got[E:E + n], want[E:E + n]), (label, "counts", np.nonzero(got[E:E + n] != want[E:E + n])[0][:8])
    return int(want[E:E + n].sum())


def test_sao_job_restatement_counts_what_the_reference_counts():
    """orc_saojob_run_8 calls the pinned saoCuStats* restatements in SAO::calcSaoStatsCTU's order; here its numbers are checked against a direct statement of
    what that is — per class, every sample of the rectangle classified by its two neighbours — so that the restatement and the device, which is written the
    direct way, are not the same code twice.  (Against the REAL calcSaoStatsCTU: test_bound_encoder_sao_statistics_jobs..., under X265HIP_VERIFY.)"""
    from x265_amd import hipprim as hp
    O = _orc()
    rng = np.random.default_rng(31)
    eo = [1, 2, 0, 3, 4]
    nb = {1: ((0, 1), (0, -1)), 2: ((1, 0), (-1, 0)), 3: ((1, 1), (-1, -1)), 4: ((1, -1), (-1, 1))}
    for it in range(60):
        j, pix = _sao_job(hp, rng, int(rng.integers(1, 4)), it % 4 == 0)
        want = _oracle_sao(hp, O, j, pix)
        at = 0
        for b in range(j.planes):
            w, h = j.plane[b].w, j.plane[b].h
            rec = pix[at:at + (w + 1) * (h + 1)].reshape(h + 1, w + 1).astype(np.int32); at += (w + 1) * (h + 1)
            src = pix[at:at + w * h].reshape(h, w).astype(np.int32); at += w * h
            for c in range(5):
                if c >= 3 and not j.eo23:
                    assert not want[b * 160 + c * 32: b * 160 + c * 32 + 32].any()
                    continue
                sums, cnts = np.zeros(32, np.int64), np.zeros(32, np.int64)
                for y in range(j.plane[b].y0[c], j.plane[b].y1[c]):
                    for x in range(j.plane[b].x0[c], j.plane[b].x1[c]):
                        v = rec[y + 1, x + 1]
                        if c == 0:
                            k = v >> 3
                        else:
                            (ay, ax), (by, bx) = nb[c]
                            k = eo[int(np.sign(v - rec[y + 1 + ay, x + 1 + ax])) + int(np.sign(v - rec[y + 1 + by, x + 1 + bx])) + 2]
                        sums[k] += src[y, x] - v
                        cnts[k] += 1
                assert np.array_equal(want[b * 160 + c * 32: b * 160 + c * 32 + 32], sums), (it, b, c)
                assert np.array_equal(want[480 + b * 160 + c * 32: 480 + b * 160 + c * 32 + 32], cnts), (it, b, c)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_device_sao_jobs_match_the_restatement(mode):
    """SAO statistics jobs on the MI355X against the restatement: 1-3 planes, full and partial CTUs, every availability pattern, eo23 on and off; mixed with
    CU jobs on the same slots (the two job kinds share the resident server)"""
    from x265_amd import hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    O = _orc()
    cs = vp()
    hp.check(L.x265hip_cuserve_open(4, mode, C.byref(cs)))
    try:
        rng = np.random.default_rng(77 + mode)
        measured = 0
        for it in range(240):
            if it % 5 == 4:
                j = _job_header(hp, 5, 5, 4, 1, 8, (30, 29, 29), 0, 1)
                pix = _job_pixels(rng, 5, 1, 8, 1)
                done, wu, wl, wr = _oracle_job(hp, O, j, pix)
                _compare(hp, j, _run_on(hp, L, cs, it % 4, j, pix), wu, wl, wr, ("cu job between SAO jobs", it))
                continue
            j, pix = _sao_job(hp, rng, int(rng.integers(1, 4)), it % 3 == 0)
            measured += _same_sao(j, _run_sao_on(hp, L, cs, it % 4, j, pix), _oracle_sao(hp, O, j, pix), (mode, it))
        assert measured > 500000
    finally:
        hp.check(L.x265hip_cuserve_close(cs))


def test_emulated_sao_jobs_are_the_restatement():
    from x265_amd import hipprim as hp
    if not os.path.exists(EMUL):
        pytest.skip("tests/support/libx265hip_emul.so not built (make -C oracle emul)")
    em = C.CDLL(EMUL)
    for name, (res, args) in hp.PROTOTYPES.items():
        if name.startswith("x265hip_cuserve_"):
            fn = getattr(em, name)
            fn.restype, fn.argtypes = res, args
    O = _orc()
    cs = vp()
    assert em.x265hip_cuserve_open(2, 0, C.byref(cs)) == 0
    rng = np.random.default_rng(5)
    for it in range(40):
        j, pix = _sao_job(hp, rng, int(rng.integers(1, 4)), it % 3 == 0)
        _same_sao(j, _run_sao_on(_Chk, em, cs, it % 2, j, pix), _oracle_sao(hp, O, j, pix), it)
    assert em.x265hip_cuserve_close(cs) == 0


@pytest.mark.parametrize("extra", [[], ["--limit-sao"], ["--sao-non-deblock"], ["--slices", "2"], ["--ctu", "32"], ["--preset", "slow"]], ids=lambda e: "-".join(x.strip("-") for x in e) or "medium")
def test_bound_encoder_sao_statistics_jobs_stay_byte_identical(tmp_path, extra):
    """SAO::calcSaoStatsCTU served by jobs (emulated ABI), 328x200: partial CTUs on the right and at the bottom.  X265HIP_VERIFY runs the reference's own body
    beside every served plane and aborts on a different sum or count."""
    import re, subprocess, sys
    sys.path.insert(0, ROOT)
    ref, emul = os.path.join(ROOT, "oracle", "_ref", "x265_8bit"), os.path.join(ROOT, "oracle", "_ref", "x265_emul_8bit")
    if not (os.path.exists(ref) and os.path.exists(emul)):
        pytest.skip("oracle/_ref encoders not built (make -C oracle ref emul)")
    from x265_amd.synth import make_clip
    yuv = str(tmp_path / "clip.yuv")
    make_clip(yuv, 328, 200, 6, seed=78)
    args = ["--input", yuv, "--input-res", "328x200", "--fps", "30", "--frames", "6", "--preset", "medium", "--hash", "1", "--pools", "4", "-F", "2"] + extra
    want, got = str(tmp_path / "ref.hevc"), str(tmp_path / "emul.hevc")
    assert subprocess.run([ref] + args + ["-o", want], capture_output=True, timeout=600).returncode == 0
    r = subprocess.run([emul] + args + ["-o", got], capture_output=True, text=True, timeout=600, env=dict(os.environ, X265HIP="require", X265HIP_VERBOSE="1", X265HIP_VERIFY="1"))
    assert r.returncode == 0, r.stderr[-800:]
    assert open(got, "rb").read() == open(want, "rb").read()
    m = re.search(r"saostats: SAO statistics of (\d+) CTU planes .*? in (\d+) jobs, (\d+) planes on the host", r.stderr)
    assert m and int(m.group(1)) > 100 and int(m.group(3)) == 0, r.stderr[-800:]


# ---- intra scan jobs (x265hip_intrajob) -----------------------------------------------------------------------------------------------------------

def _intra_job(hp, rng, log2n, depth, wild):
    """a job as Search::checkIntraInInter's seam builds it: the unfiltered neighbour line of a textured block (or, `wild`, unrelated samples: large costs,
    clipping edge gradients), its [1 2 1] filtered twin (oracle primitive), the source block"""
    n = 1 << log2n
    pix = np.uint8 if depth == 8 else np.uint16
    pmax = (1 << depth) - 1
    line = 4 * n + 16
    j = hp.IntraScanJob()
    j.bitDepth, j.mark, j.log2Size = depth, hp.INTRAJOB_MARK, log2n
    base = int(rng.integers(0, pmax - 40))
    fenc = (base + rng.integers(0, 40, (n, n)) + np.add.outer(np.arange(n), np.arange(n)) // 3).clip(0, pmax).astype(pix)
    raw = rng.integers(0, pmax + 1, 4 * n + 1).astype(pix) if wild else (base + rng.integers(0, 40, 4 * n + 1)).clip(0, pmax).astype(pix)
    O = _orc()
    flt = np.zeros(4 * n + 1, pix)
    getattr(O, "orc_intra_filter_%d" % (8 if depth == 8 else 16))(n, raw.ctypes.data, flt.ctypes.data)
    blob = np.zeros(2 * line + n * n, pix)
    blob[:4 * n + 1] = raw
    blob[line:line + 4 * n + 1] = flt
    blob[2 * line:] = fenc.ravel()
    return j, blob


def _oracle_intra(hp, O, j, pix):
    units = (hp.CuJobUnit * hp.CUJOB_MAX_UNITS)()
    out = np.zeros(35, np.int32)
    assert O.orc_intrajob_run(C.byref(j), pix.ctypes.data, C.byref(units), out.ctypes.data, 1) == 35
    return out


def _run_intra_on(hp, L, cs, slot, j, pix, timeout=20.0):
    job, pixels, units, levels, resi = vp(), vp(), vp(), vp(), vp()
    hp.check(L.x265hip_cuserve_slot(cs, slot, C.byref(job), C.byref(pixels), C.byref(units), C.byref(levels), C.byref(resi)))
    C.memmove(pixels, pix.ctypes.data, pix.nbytes)
    seq = u32()
    hp.check(L.x265hip_cuserve_submit_intra(cs, slot, C.byref(j), C.byref(seq)))
    un = C.cast(units, C.POINTER(hp.CuJobUnit))
    t0 = time.time()
    while un[0].ready != seq.value:
        pk = L.x265hip_cuserve_poke(cs, slot)
        if pk < 0:
            hp.check(pk)
        assert time.time() - t0 < timeout, "intra scan job not finished after %.0f s" % timeout
    _intra_ticks.setdefault(j.log2Size, []).append(int(un[0].fwdTicks))
    return np.ctypeslib.as_array(C.cast(levels, C.POINTER(C.c_int32)), (35,)).copy()


_intra_ticks = {}    # log2 size -> the device's own clock, doorbell seen -> costs out (100 MHz ticks), of the jobs run so far


def test_intra_job_restatement_is_the_pinned_primitives():
    """orc_intrajob_run against the composition the table-primitive test pins (tests/test_hip_parity.py::test_intra_mode_scan_matches_oracle): prediction of
    each mode from the line the filter flags pick, then sa8d — here through the numpy-facing oracle wrappers"""
    from x265_amd import hipprim as hp
    from backends import Orc
    O = _orc()
    rng = np.random.default_rng(5)
    for depth in (8, 10):
        o = Orc(depth)
        for log2n in (3, 4, 5):
            n = 1 << log2n
            j, blob = _intra_job(hp, rng, log2n, depth, False)
            got = _oracle_intra(hp, O, j, blob)
            line = 4 * n + 16
            raw, flt, fenc = blob[:4 * n + 1], blob[line:line + 4 * n + 1], blob[2 * line:].reshape(n, n)
            for mode in range(35):
                pred = o.intra_pred(n, mode, flt if o.intra_uses_filtered(n, mode) else raw, 1 if n <= 16 else 0)
                assert got[mode] == o.sa8d(n, fenc, (0, 0), pred, (0, 0)), (depth, n, mode)


def test_emulated_intra_jobs_are_the_restatement():
    from x265_amd import hipprim as hp
    if not os.path.exists(EMUL):
        pytest.skip("tests/support/libx265hip_emul.so not built (make -C oracle emul)")
    E = C.CDLL(EMUL)
    for name, (res, args) in hp.PROTOTYPES.items():
        if name.startswith("x265hip_cuserve_"):
            fn = getattr(E, name)
            fn.restype, fn.argtypes = res, args
    O = _orc()
    cs = vp()
    assert E.x265hip_cuserve_open(2, 0, C.byref(cs)) == 0
    try:
        rng = np.random.default_rng(6)
        for it in range(12):
            j, blob = _intra_job(hp, rng, 3 + it % 3, (8, 10, 12)[it % 3 if it > 5 else 0], it % 4 == 3)
            assert np.array_equal(_run_intra_on(_Chk, E, cs, it % 2, j, blob), _oracle_intra(hp, O, j, blob)), it
    finally:
        assert E.x265hip_cuserve_close(cs) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_device_intra_jobs_match_the_restatement(mode):
    """intra scan jobs on the MI355X against the restatement: 8x8 / 16x16 / 32x32, 8 / 10 / 12 bit, textured and unrelated neighbours; mixed with CU jobs
    and SAO statistics jobs on the same slots"""
    from x265_amd import hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    O = _orc()
    cs = vp()
    hp.check(L.x265hip_cuserve_open(4, mode, C.byref(cs)))
    try:
        rng = np.random.default_rng(91 + mode)
        scans = 0
        for it in range(180):
            if it % 7 == 5:
                j = _job_header(hp, 5, 5, 4, 1, 8, (30, 29, 29), 0, 1)
                pix = _job_pixels(rng, 5, 1, 8, 1)
                done, wu, wl, wr = _oracle_job(hp, O, j, pix)
                _compare(hp, j, _run_on(hp, L, cs, it % 4, j, pix), wu, wl, wr, ("cu job between intra jobs", it))
                continue
            if it % 7 == 6:
                sj, spix = _sao_job(hp, rng, 3, True)
                _same_sao(sj, _run_sao_on(hp, L, cs, it % 4, sj, spix), _oracle_sao(hp, O, sj, spix), ("sao job between intra jobs", it))
                continue
            j, blob = _intra_job(hp, rng, 3 + it % 3, (8, 10, 12)[(it // 3) % 3], it % 5 == 4)
            got, want = _run_intra_on(hp, L, cs, it % 4, j, blob), _oracle_intra(hp, O, j, blob)
            assert np.array_equal(got, want), (mode, it, j.log2Size, j.bitDepth, np.nonzero(got != want)[0][:6], got[:4], want[:4])
            scans += 1
        assert scans > 100
        print("intra scan jobs, device time (doorbell seen -> costs out), mode %d: " % mode +
              ", ".join("%dx%d median %.1f us" % (1 << k, 1 << k, sorted(v)[len(v) // 2] / 100.0) for k, v in sorted(_intra_ticks.items())))
        _intra_ticks.clear()
    finally:
        hp.check(L.x265hip_cuserve_close(cs))


@pytest.mark.parametrize("extra,env", [([], {}), (["--bframes", "0"], {}), (["--fast-intra"], {}), (["--constrained-intra"], {}), (["--no-strong-intra-smoothing", "--rd", "2"], {}),
                                       (["--b-intra", "--preset", "slow"], {}), (["--ctu", "32"], {"X265HIP_INTRASCAN_AHEAD": "0", "X265HIP_INTRASCAN_SYNC_MIN": "4"})],
                         ids=lambda v: ("-".join(x.strip("-") for x in v) or "medium") if isinstance(v, list) else ("sync" if v else "ahead"))
def test_bound_encoder_intra_scan_jobs_stay_byte_identical(tmp_path, extra, env):
    """Search::checkIntraInInter's 35-mode scan served by jobs (emulated ABI).  X265HIP_VERIFY: the table slots also do the reference's work and every cost
    the job returns is compared with the sa8d the reference measures, mode by mode; jobs submitted ahead must be adopted whenever the intra try comes."""
    import re, subprocess, sys
    sys.path.insert(0, ROOT)
    ref, emul = os.path.join(ROOT, "oracle", "_ref", "x265_8bit"), os.path.join(ROOT, "oracle", "_ref", "x265_emul_8bit")
    if not (os.path.exists(ref) and os.path.exists(emul)):
        pytest.skip("oracle/_ref encoders not built (make -C oracle ref emul)")
    from x265_amd.synth import make_clip
    yuv = str(tmp_path / "clip.yuv")
    make_clip(yuv, 416, 240, 10, seed=80)
    args = ["--input", yuv, "--input-res", "416x240", "--fps", "30", "--frames", "10", "--preset", "medium", "--hash", "1", "--pools", "4", "-F", "2"] + extra
    want, got = str(tmp_path / "ref.hevc"), str(tmp_path / "emul.hevc")
    assert subprocess.run([ref] + args + ["-o", want], capture_output=True, timeout=600).returncode == 0
    for verify in (({"X265HIP_VERIFY": "1"}, {}) if not extra else ({"X265HIP_VERIFY": "1"},)):      # (the plain run once: VERIFY makes the slots do the host's work too)
        r = subprocess.run([emul] + args + ["-o", got], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, X265HIP="require", X265HIP_VERBOSE="1", **verify, **env))
        assert r.returncode == 0, r.stderr[-800:]
        assert open(got, "rb").read() == open(want, "rb").read()
        m = re.search(r"intrascan: the 35-mode sa8d scans of (\d+) blocks .*? in (\d+) jobs, .*?; (\d+) jobs left ahead when predInterSearch (?:returned|was entered), (\d+) of them adopted", r.stderr)
        assert m and int(m.group(1)) > 20, r.stderr[-800:]
        if not env:
            # (nearly) every intra try found its job ahead, and it was the right one (with --limit-refs the seam predicts the try from the split's trace)
            assert int(m.group(4)) >= 0.9 * int(m.group(1)), r.stderr[-800:]


@pytest.mark.parametrize("extra", [[], ["--bframes", "0", "--rd", "4"], ["--preset", "slow"]], ids=lambda e: "-".join(x.strip("-") for x in e) or "medium")
def test_bound_encoder_inter_candidate_jobs_ahead_stay_byte_identical(tmp_path, extra):
    """X265HIP_CUSERVE_SPEC_INTER=1 (the default since round 6's second session): the 2Nx2N inter candidate's job leaves when Search::predInterSearch returns and is adopted by
    encodeResAndCalcRdInterCU sample for sample (analysis.cpp:1421-1611).  Every such job must be the one wanted; with rectangular partitions (preset slow)
    none may leave."""
    import re, subprocess, sys
    sys.path.insert(0, ROOT)
    ref, emul = os.path.join(ROOT, "oracle", "_ref", "x265_8bit"), os.path.join(ROOT, "oracle", "_ref", "x265_emul_8bit")
    if not (os.path.exists(ref) and os.path.exists(emul)):
        pytest.skip("oracle/_ref encoders not built (make -C oracle ref emul)")
    from x265_amd.synth import make_clip
    yuv = str(tmp_path / "clip.yuv")
    make_clip(yuv, 416, 240, 8, seed=79)
    args = ["--input", yuv, "--input-res", "416x240", "--fps", "30", "--frames", "8", "--preset", "medium", "--hash", "1", "--pools", "4", "-F", "2"] + extra
    want, got = str(tmp_path / "ref.hevc"), str(tmp_path / "emul.hevc")
    assert subprocess.run([ref] + args + ["-o", want], capture_output=True, timeout=600).returncode == 0
    r = subprocess.run([emul] + args + ["-o", got], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, X265HIP="require", X265HIP_VERBOSE="1", X265HIP_VERIFY="1", X265HIP_CUSERVE_SPEC_INTER="1"))
    assert r.returncode == 0, r.stderr[-800:]
    assert open(got, "rb").read() == open(want, "rb").read()
    m = re.search(r"(\d+) jobs left ahead of their scope when predInterSearch returned .*?; (\d+) of them were the job", r.stderr)
    if "slow" in extra:
        assert not m, r.stderr[-800:]
    else:
        assert m and int(m.group(1)) > 50 and m.group(1) == m.group(2), r.stderr[-800:]


@pytest.mark.parametrize("extra,bits", [(["--preset", "slow"], 8), (["--preset", "slower", "--rd", "6"], 10), (["--preset", "slow", "--ctu", "32", "--psy-rdoq", "0"], 8)],
                         ids=["slow", "slower-rd6-main10", "slow-ctu32-nopsyrdoq"])
def test_bound_encoder_inverse_jobs_behind_rdoq_stay_byte_identical(tmp_path, extra, bits):
    """the RDOQ presets (BASELINE configs[2] / [3]'s): Quant::rdoQuant makes a luma 32x32 unit's levels on the host, the unit's inverse half leaves as an
    x265hip_cujob of its own (coefMode X265HIP_CUJOB_INVERSE) and Quant::invtransformNxN collects it — residual, sse_pp and psy energy of the reconstruction.
    Here over the emulated ABI with X265HIP_VERIFY=1: same bytes as the unmodified encoder, jobs really left and were collected; and switched off
    (X265HIP_CUSERVE_INVERSE=0) none leaves."""
    import re, subprocess, sys
    sys.path.insert(0, ROOT)
    ref, emul = os.path.join(ROOT, "oracle", "_ref", "x265_%dbit" % bits), os.path.join(ROOT, "oracle", "_ref", "x265_emul_%dbit" % bits)
    if not (os.path.exists(ref) and os.path.exists(emul)):
        pytest.skip("oracle/_ref encoders not built (make -C oracle ref emul)")
    from x265_amd.synth import make_clip
    yuv = str(tmp_path / "clip.yuv")
    make_clip(yuv, 416, 240, 6, seed=83)
    args = ["--input", yuv, "--input-res", "416x240", "--fps", "30", "--frames", "6", "--hash", "1", "--pools", "4", "-F", "2"] + extra
    want, got = str(tmp_path / "ref.hevc"), str(tmp_path / "emul.hevc")
    assert subprocess.run([ref] + args + ["-o", want], capture_output=True, timeout=900).returncode == 0
    for off in (False, True):
        # (the default is on for 8-bit builds only: profiles/r06_v1_configs3_4k_main10_slower_ab.txt; here both settings are explicit)
        env = dict(os.environ, X265HIP="require", X265HIP_VERBOSE="1", X265HIP_VERIFY="1", X265HIP_CUSERVE_INVERSE="0" if off else "1")
        r = subprocess.run([emul] + args + ["-o", got], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-800:]
        assert open(got, "rb").read() == open(want, "rb").read()
        m = re.search(r"cuserve: (\d+) inverse jobs .*? left when the levels were made, (\d+) never collected", r.stderr)
        served = re.search(r"(\d+) forward transform\+quant units and (\d+) inverse units served", r.stderr)
        if off:
            assert not m and served and int(served.group(2)) == 0, r.stderr[-800:]
        else:
            assert m and int(m.group(1)) > 20 and int(m.group(2)) * 10 <= int(m.group(1)), r.stderr[-800:]
            assert served and int(served.group(2)) == int(m.group(1)) - int(m.group(2)), r.stderr[-800:]


class _Chk:
    """hp-like holder for _run_on over the emulated library (no HipError there)"""
    from x265_amd import hipprim as _hp
    CuJobUnit, CUJOB_MAX_ELEMS, SAOJOB_STATS_ENTRIES = _hp.CuJobUnit, _hp.CUJOB_MAX_ELEMS, _hp.SAOJOB_STATS_ENTRIES

    @staticmethod
    def check(rc):
        assert rc == 0, rc


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_device_jobs_match_the_restatement(mode):
    """every unit of every job: numSig, levels, reconstructed residual, both distortions; 8 / 10 / 12 bit; slots reused, sequence numbers running"""
    from x265_amd import hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    O = _orc()
    cs = vp()
    hp.check(L.x265hip_cuserve_open(4, mode, C.byref(cs)))
    try:
        total = units = coded = 0
        for depth in (8, 10, 12):
            for shape, kind, qps, sliceI, signHide, rng in _cases(depth, 40 + depth):
                j = _job_header(hp, *shape, depth, qps, sliceI, signHide)
                pix = _job_pixels(rng, shape[0], shape[3], depth, kind)
                done, wu, wl, wr = _oracle_job(hp, O, j, pix)
                n, c = _compare(hp, j, _run_on(hp, L, cs, total % 4, j, pix), wu, wl, wr, (mode, depth, shape, kind, qps, sliceI, signHide))
                assert n == done
                total += 1; units += n; coded += c
        jobs, starts, ns = C.c_uint64(), C.c_uint64(), C.c_uint64()
        hp.check(L.x265hip_cuserve_stats(cs, C.byref(jobs), C.byref(starts), C.byref(ns)))
        assert jobs.value == total and ns.value > 0
        assert coded > units // 3
        if mode == 0:
            # the resident server leaves by itself when idle and comes back for the next job
            time.sleep(0.05)
            j = _job_header(hp, 5, 5, 5, 1, 8, (30, 29, 29), 0, 1)
            pix = _job_pixels(np.random.default_rng(5), 5, 1, 8, 1)
            done, wu, wl, wr = _oracle_job(hp, O, j, pix)
            _compare(hp, j, _run_on(hp, L, cs, 0, j, pix), wu, wl, wr, "after idling")
            hp.check(L.x265hip_cuserve_stats(cs, C.byref(jobs), C.byref(starts), C.byref(ns)))
            assert starts.value >= 2, starts.value
    finally:
        hp.check(L.x265hip_cuserve_close(cs))


@pytest.mark.gpu
def test_a_seventeenth_resident_service_is_refused_not_deadlocked():
    """x265hip_cuserve_open keeps at most sixteen services with a resident server: the next one is an error (it used to close the half-made service while
    holding the lock that close takes: a deadlock instead of X265HIP_EINVAL), and the sixteen stay usable"""
    import threading
    from x265_amd import hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    opened, result = [], {}
    try:
        for _ in range(16):
            cs = vp()
            hp.check(L.x265hip_cuserve_open(1, 0, C.byref(cs)))
            opened.append(cs)
        def one_more():
            cs = vp()
            result["rc"] = L.x265hip_cuserve_open(1, 0, C.byref(cs))
        t = threading.Thread(target=one_more, daemon=True)
        t.start()
        t.join(20)
        assert not t.is_alive(), "the seventeenth x265hip_cuserve_open never returned"
        assert result["rc"] == -1, result                       # X265HIP_EINVAL
        O = _orc()
        j = _job_header(hp, 5, 5, 5, 1, 8, (30, 29, 29), 0, 1)
        pix = _job_pixels(np.random.default_rng(11), 5, 1, 8, 1)
        done, wu, wl, wr = _oracle_job(hp, O, j, pix)
        _compare(hp, j, _run_on(hp, L, opened[-1], 0, j, pix), wu, wl, wr, "sixteenth service")
    finally:
        for cs in opened:
            hp.check(L.x265hip_cuserve_close(cs))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_device_inverse_jobs_match_the_restatement(mode):
    """x265hip_cujob::coefMode == X265HIP_CUJOB_INVERSE (what the RDOQ presets could hand over behind Quant::rdoQuant): the inverse half alone of one 32x32 luma
    unit, for levels made elsewhere.  The levels are the restatement's own for a whole job on the same pixels; the inverse job's reconstructed residual, codedDist,
    codedEnergy, zeroDist and numSig must be that job's (8 / 10 / 12 bit)."""
    from x265_amd import hipprim as hp
    L = hp.lib()
    hp.check(L.x265hip_init(0))
    O = _orc()
    cs = vp()
    hp.check(L.x265hip_cuserve_open(2, mode, C.byref(cs)))
    try:
        n_done = 0
        for depth in (8, 10, 12):
            rng = np.random.default_rng(900 + depth)
            bd = 6 * (depth - 8)
            for kind in (0, 1, 2, 1, 2, 1):
                qy = int(rng.integers(bd + 12, bd + 40))
                j = _job_header(hp, 5, 5, 5, 0, depth, (qy, qy, qy), 0, int(rng.integers(0, 2)))
                pix = _job_pixels(rng, 5, 0, depth, kind)
                done, wu, wl, wr = _oracle_job(hp, O, j, pix)
                assert done == 1
                if not wu[0].numSig:
                    continue
                ji = _job_header(hp, 5, 5, 5, 0, depth, (qy, qy, qy), 0, 0, coef=8)
                blob = np.frombuffer(pix.tobytes() + wl[:1024].tobytes(), np.uint8).copy()
                job, pixels, units, levels, resi = vp(), vp(), vp(), vp(), vp()
                slot = n_done % 2
                hp.check(L.x265hip_cuserve_slot(cs, slot, C.byref(job), C.byref(pixels), C.byref(units), C.byref(levels), C.byref(resi)))
                C.memmove(job, C.byref(ji), C.sizeof(ji))
                C.memmove(pixels, blob.ctypes.data, blob.nbytes)
                seq = u32()
                hp.check(L.x265hip_cuserve_submit(cs, slot, C.byref(seq)))
                un = C.cast(units, C.POINTER(hp.CuJobUnit))
                t0 = time.time()
                while un[0].ready != seq.value or un[0].readyInv != seq.value:
                    pk = L.x265hip_cuserve_poke(cs, slot)
                    if pk < 0:
                        hp.check(pk)
                    assert time.time() - t0 < 20.0, "inverse job not finished"
                rs = np.ctypeslib.as_array(C.cast(resi, C.POINTER(C.c_int16)), (1024,)).copy()
                label = (mode, depth, kind, qy)
                assert un[0].numSig == int(np.count_nonzero(wl[:1024])), label
                assert un[0].zeroDist == wu[0].zeroDist and un[0].codedDist == wu[0].codedDist and un[0].codedEnergy == wu[0].codedEnergy, label
                assert np.array_equal(rs, wr[:1024]), label
                n_done += 1
        assert n_done >= 9, n_done
        # an inverse job is one 32x32 luma unit: anything else is refused
        bad = _job_header(hp, 6, 5, 5, 0, 8, (30, 30, 30), 0, 0, coef=8)
        job = vp()
        hp.check(L.x265hip_cuserve_slot(cs, 0, C.byref(job), None, None, None, None))
        C.memmove(job, C.byref(bad), C.sizeof(bad))
        seq = u32()
        assert L.x265hip_cuserve_submit(cs, 0, C.byref(seq)) == -1
    finally:
        hp.check(L.x265hip_cuserve_close(cs))
