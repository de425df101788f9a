"""Uniform numpy-level front ends over the CPU checkers (tests only):

  Orc(depth)  -> oracle/libx265oracle.so  (our plain-C restatement)
  Ref(depth)  -> oracle/_ref/libx265ref{8,10}.so (the real reference C primitives + MotionEstimate)

Both expose the same methods so a parity test is `assert same(Orc(d).f(*a), Ref(d).f(*a))`.
Blocks are passed as C-contiguous 2-D numpy arrays plus an (y, x) origin; stride = array width.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402

PU_SIZES = po.PU_SIZES
CU_SIZES = [4, 8, 16, 32, 64]
ptr = po.ptr


def part_of(w, h):
    return PU_SIZES.index((w, h))


def cu_of(size):
    return CU_SIZES.index(size)


class _Base:
    def __init__(self, depth):
        self.depth = depth
        self.pix = po.pix_dtype(depth)
        self.pmax = (1 << depth) - 1


def entropy_state_bits_fixture():
    """x265_entropyStateBits[128] as dumped from the reference build by tests/golden/make_golden.py (data, not an algorithm)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "entropy_state_bits.json")
    return np.array(json.load(open(path))["entropyStateBits"], np.uint32)


def scan_order_py(stype, log2):
    """The scan of a (1 << log2) TU by the rule of the standard (4x4 groups in `stype` order, 16x16 / 32x32 always diagonal); used by the
    test wrappers to place jobs, and checked against the oracle / reference tables in the tests."""
    def pos(t, n):
        out = []
        if t == 0:
            x = y = 0
            while len(out) < n * n:
                while y >= 0:
                    if x < n and y < n:
                        out.append((x, y))
                    y -= 1
                    x += 1
                y, x = x, 0
        elif t == 1:
            out = [(x, y) for y in range(n) for x in range(n)]
        else:
            out = [(x, y) for x in range(n) for y in range(n)]
        return out
    size = 1 << log2
    t = 0 if log2 > 3 else stype
    cg = pos(t, size // 4) if size > 4 else [(0, 0)]
    inner = pos(t, 4)
    return np.array([(cy * 4 + py) * size + cx * 4 + px for (cx, cy) in cg for (px, py) in inner], np.uint16)


def sig_ctx_table(log2, pattern):
    """tabSigCtx of codeCoeffNxN (entropy.cpp table_cnt; HEVC 9.3.4.2.5): the significance-context increment of each position of a 4x4
    group — the fixed map of a 4x4 TU, otherwise by which neighbouring groups are coded (pattern bit 0: right, bit 1: below)."""
    if log2 == 2:
        return np.array([0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8], np.uint8)
    t = np.zeros(16, np.uint8)
    for y in range(4):
        for x in range(4):
            if pattern == 0:
                v = 2 if x + y == 0 else (1 if x + y < 3 else 0)
            elif pattern == 1:
                v = 2 if y == 0 else (1 if y == 1 else 0)
            elif pattern == 2:
                v = 2 if x == 0 else (1 if x == 1 else 0)
            else:
                v = 2
            t[y * 4 + x] = v
    return t


class Orc(_Base):
    name = "oracle"

    def __init__(self, depth):
        super().__init__(depth)
        self.L = po.oracle()
        self.s = po.sfx(depth)

    def _f(self, name):
        return getattr(self.L, "%s_%s" % (name, self.s))

    # ---- pixel compare
    def sad(self, w, h, a, ao, b, bo):
        return self._f("orc_sad")(ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1], w, h)

    def sad_xn(self, w, h, fenc, ref, offs):
        res = np.zeros(len(offs), np.int32)
        ps = [ptr(ref, *o) for o in offs]
        if len(offs) == 3:
            self._f("orc_sad_x3")(ptr(fenc), ps[0], ps[1], ps[2], ref.shape[1], w, h, ptr(res))
        else:
            self._f("orc_sad_x4")(ptr(fenc), ps[0], ps[1], ps[2], ps[3], ref.shape[1], w, h, ptr(res))
        return res

    def satd(self, w, h, a, ao, b, bo):
        return self._f("orc_satd")(ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1], w, h)

    def sa8d(self, size, a, ao, b, bo):
        return self._f("orc_sa8d")(ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1], size)

    def sse_pp(self, size, a, ao, b, bo):
        return self._f("orc_sse_pp")(ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1], size, size)

    def sse_ss(self, size, a, ao, b, bo):
        return self.L.orc_sse_ss(ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1], size, size)

    def ssd_s(self, size, a, ao):
        return self.L.orc_ssd_s(ptr(a, *ao), a.shape[1], size)

    def psy_cost_pp(self, size, a, ao, b, bo):
        return self._f("orc_psy_cost_pp")(ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1], size)

    def var(self, size, a, ao):
        return self._f("orc_var")(ptr(a, *ao), a.shape[1], size)

    # ---- block arithmetic (outputs returned as fresh dense arrays)
    def sub_ps(self, size, a, ao, b, bo):
        d = np.zeros((size, size), np.int16)
        self._f("orc_sub_ps")(ptr(d), size, ptr(a, *ao), ptr(b, *bo), a.shape[1], b.shape[1], size, size)
        return d

    def add_ps(self, size, a, ao, r, ro):
        d = np.zeros((size, size), self.pix)
        self._f("orc_add_ps")(ptr(d), size, ptr(a, *ao), ptr(r, *ro), a.shape[1], r.shape[1], size, size, self.depth)
        return d

    def addAvg(self, w, h, a, ao, b, bo):
        d = np.zeros((h, w), self.pix)
        self._f("orc_addAvg")(ptr(a, *ao), ptr(b, *bo), ptr(d), a.shape[1], b.shape[1], w, w, h, self.depth)
        return d

    def pixelavg_pp(self, w, h, a, ao, b, bo):
        d = np.zeros((h, w), self.pix)
        self._f("orc_pixelavg_pp")(ptr(d), w, ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1], w, h)
        return d

    def p2s(self, w, h, a, ao):
        d = np.zeros((h, w), np.int16)
        self._f("orc_p2s")(ptr(a, *ao), a.shape[1], ptr(d), w, w, h, self.depth)
        return d

    def cpy2Dto1D_shl(self, size, a, ao, shift):
        d = np.zeros(size * size, np.int16)
        self.L.orc_cpy2Dto1D_shl(ptr(d), ptr(a, *ao), a.shape[1], shift, size)
        return d

    def cpy2Dto1D_shr(self, size, a, ao, shift):
        d = np.zeros(size * size, np.int16)
        self.L.orc_cpy2Dto1D_shr(ptr(d), ptr(a, *ao), a.shape[1], shift, size)
        return d

    def cpy1Dto2D_shl(self, size, a, shift):
        d = np.zeros((size, size), np.int16)
        self.L.orc_cpy1Dto2D_shl(ptr(d), ptr(a), size, shift, size)
        return d

    def cpy1Dto2D_shr(self, size, a, shift):
        d = np.zeros((size, size), np.int16)
        self.L.orc_cpy1Dto2D_shr(ptr(d), ptr(a), size, shift, size)
        return d

    def copy_cnt(self, size, a, ao):
        d = np.zeros(size * size, np.int16)
        n = self.L.orc_copy_cnt(ptr(d), ptr(a, *ao), a.shape[1], size)
        return d, n

    def count_nonzero(self, size, a):
        return self.L.orc_count_nonzero(ptr(a), size)

    # ---- transforms
    def dct(self, size, a, ao):
        d = np.zeros(size * size, np.int16)
        self.L.orc_dct(size.bit_length() - 1, ptr(a, *ao), ptr(d), a.shape[1], self.depth)
        return d

    def idct(self, size, a):
        d = np.zeros((size, size), np.int16)
        self.L.orc_idct(size.bit_length() - 1, ptr(a), ptr(d), size, self.depth)
        return d

    def dst4(self, a, ao):
        d = np.zeros(16, np.int16)
        self.L.orc_dst4(ptr(a, *ao), ptr(d), a.shape[1], self.depth)
        return d

    def idst4(self, a):
        d = np.zeros((4, 4), np.int16)
        self.L.orc_idst4(ptr(a), ptr(d), 4, self.depth)
        return d

    def quant(self, coef, qc, qbits, add):
        n = coef.size
        du = np.zeros(n, np.int32)
        q = np.zeros(n, np.int16)
        ns = self.L.orc_quant(ptr(coef), ptr(qc), ptr(du), ptr(q), qbits, add, n)
        return q, du, ns

    def nquant(self, coef, qc, qbits, add):
        n = coef.size
        q = np.zeros(n, np.int16)
        ns = self.L.orc_nquant(ptr(coef), ptr(qc), ptr(q), qbits, add, n)
        return q, ns

    def dequant_normal(self, q, scale, shift):
        c = np.zeros(q.size, np.int16)
        self.L.orc_dequant_normal(ptr(q), ptr(c), q.size, scale, shift)
        return c

    def dequant_scaling(self, q, dq, per, shift):
        c = np.zeros(q.size, np.int16)
        self.L.orc_dequant_scaling(ptr(q), ptr(dq), ptr(c), q.size, per, shift)
        return c

    def denoise_dct(self, coef, ressum, offset):
        c, r = coef.copy(), ressum.copy()
        self.L.orc_denoise_dct(ptr(c), ptr(r), ptr(offset), c.size)
        return c, r

    def rdoquant(self, kind, size, resi, fenc, psyscale, blkpos):
        log2n = size.bit_length() - 1
        cu_ = np.full(size * size, 7, np.int64)
        tot = np.array([11, 13], np.int64)
        ps = np.array([psyscale], np.int64)
        tu, tr = po.vp(tot.ctypes.data), po.vp(tot.ctypes.data + 8)
        if kind == "nonpsy":
            self.L.orc_nonpsy_rdoquant(log2n, ptr(resi), ptr(cu_), tu, tr, blkpos, self.depth)
        elif kind == "psy":
            self.L.orc_psy_rdoquant(log2n, ptr(resi), ptr(fenc), ptr(cu_), tu, tr, ptr(ps), blkpos, self.depth)
        elif kind == "psy1":
            self.L.orc_psy_rdoquant_1p(log2n, ptr(resi), ptr(cu_), tu, tr, blkpos, self.depth)
        else:
            self.L.orc_psy_rdoquant_2p(log2n, ptr(resi), ptr(fenc), ptr(cu_), tu, tr, ptr(ps), blkpos, self.depth)
        return cu_, tot

    # ---- interpolation; src origin must leave >= 3 (luma) / 1 (chroma) pixels of margin before it
    def interp(self, kind, chroma, w, h, src, so, idx, idy=0, ext=0):
        N = 4 if chroma else 8
        d = self.depth
        if kind == "hpp":
            out = np.zeros((h, w), self.pix)
            self._f("orc_interp_horiz_pp")(N, ptr(src, *so), src.shape[1], ptr(out), w, w, h, idx, d)
        elif kind == "hps":
            rows = h + (N - 1 if ext else 0)
            out = np.zeros((rows, w), np.int16)
            self._f("orc_interp_horiz_ps")(N, ptr(src, *so), src.shape[1], ptr(out), w, w, h, idx, ext, d)
        elif kind == "vpp":
            out = np.zeros((h, w), self.pix)
            self._f("orc_interp_vert_pp")(N, ptr(src, *so), src.shape[1], ptr(out), w, w, h, idx, d)
        elif kind == "vps":
            out = np.zeros((h, w), np.int16)
            self._f("orc_interp_vert_ps")(N, ptr(src, *so), src.shape[1], ptr(out), w, w, h, idx, d)
        elif kind == "vsp":
            out = np.zeros((h, w), self.pix)
            self._f("orc_interp_vert_sp")(N, ptr(src, *so), src.shape[1], ptr(out), w, w, h, idx, d)
        elif kind == "vss":
            out = np.zeros((h, w), np.int16)
            self.L.orc_interp_vert_ss(N, ptr(src, *so), src.shape[1], ptr(out), w, w, h, idx)
        elif kind == "hvpp":
            out = np.zeros((h, w), self.pix)
            self._f("orc_interp_hv_pp")(N, ptr(src, *so), src.shape[1], ptr(out), w, w, h, idx, idy, d)
        else:
            raise ValueError(kind)
        return out

    # ---- motion estimation
    def mvcost_table(self, qp):
        return po.mvcost_table(qp, self.depth)

    def motion_estimate(self, refplane, fencplane, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, method, subme, qp):
        fenc = np.zeros((64, 64), self.pix)
        fenc[:h, :w] = fencplane[by:by + h, bx:bx + w]
        cost = self.mvcost_table(qp)
        a = [np.array(v, np.int32) for v in (mvmin, mvmax, qmvp)]
        cand = np.array(mvc, np.int32).reshape(-1)
        out = np.zeros(2, np.int32)
        c = self._f("orc_motion_estimate")(ptr(refplane), refplane.shape[1], bx, by, ptr(fenc), w, h,
                                           ptr(a[0]), ptr(a[1]), ptr(a[2]), len(mvc), ptr(cand) if len(mvc) else None,
                                           merange, method, subme,
                                           po.vp(cost.ctypes.data + 2 * po.MVCOST_CENTRE), self.depth, ptr(out))
        return c, (int(out[0]), int(out[1]))

    SEA_WINDOWS = ((32, 32), (32, 24), (32, 8), (24, 32), (16, 16), (16, 12), (16, 4), (12, 16), (8, 32), (8, 8), (4, 16), (4, 4))

    def integral_planes(self, buf, pad=None):
        """The twelve window-sum planes of --me sea over a padded picture buffer: plane[k][y, x] = sum of the w_k x h_k window at (x, y)."""
        out = []
        for (w, h) in self.SEA_WINDOWS:
            o = np.zeros(buf.shape, np.uint32)
            self._f("orc_integral_plane")(ptr(buf), buf.shape[1], buf.shape[0], w, h, ptr(o))
            out.append(o)
        return out

    def motion_estimate_sea(self, refplane, fencplane, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, subme, qp, planes=None):
        import ctypes as C
        fenc = np.zeros((64, 64), self.pix)
        fenc[:h, :w] = fencplane[by:by + h, bx:bx + w]
        # the reference's source cache is the CU's 64x64 block: what lies right of / below the PU is the picture (AMP shapes read it)
        blk = fencplane[by:by + 64, bx:bx + 64]
        fenc[:blk.shape[0], :blk.shape[1]] = blk
        cost = self.mvcost_table(qp)
        planes = planes if planes is not None else self.integral_planes(refplane)
        arr = (C.c_void_p * 12)(*[p.ctypes.data for p in planes])
        a = [np.array(v, np.int32) for v in (mvmin, mvmax, qmvp)]
        cand = np.array(mvc, np.int32).reshape(-1)
        out = np.zeros(2, np.int32)
        c = self._f("orc_motion_estimate_sea")(ptr(refplane), refplane.shape[1], arr, bx, by, ptr(fenc), w, h, ptr(a[0]), ptr(a[1]), ptr(a[2]),
                                               len(mvc), ptr(cand) if len(mvc) else None, merange, subme,
                                               po.vp(cost.ctypes.data + 2 * po.MVCOST_CENTRE), self.depth, ptr(out))
        return c, (int(out[0]), int(out[1]))

    # ---- intra prediction / lookahead lowres
    def intra_filter(self, n, nb):
        out = np.zeros(4 * n + 1, self.pix)
        self._f("orc_intra_filter")(n, ptr(nb), ptr(out))
        return out

    def intra_pred(self, n, mode, nb, bfilter):
        out = np.zeros((n, n), self.pix)
        self._f("orc_intra_pred")(n, mode, ptr(out), n, ptr(nb), bfilter, self.depth)
        return out

    def intra_uses_filtered(self, n, mode):
        return self._f("orc_intra_uses_filtered")(n, mode)

    def intra_allangs(self, n, nb, nbf, bluma):
        out = np.zeros((33, n, n), self.pix)
        self._f("orc_intra_allangs")(n, po.vp(out.ctypes.data), ptr(nb), ptr(nbf), bluma, self.depth)
        return out

    def frame_init_lowres(self, src, origin, w, h):
        """src: padded 2-D plane, origin = (y, x) of the picture; w, h = lowres size. Returns four (h, w) planes."""
        out = [np.zeros((h, w), self.pix) for _ in range(4)]
        self._f("orc_frame_init_lowres")(ptr(src, *origin), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), src.shape[1], w, w, h)
        return tuple(out)

    def lowres_pass(self, src, origin, w, h, mx, my):
        """Lowres::init restated: downscale, extend the four planes, intra estimate. Same returns as Ref.lowres_pass."""
        lw, lh = ((w // 2 + 7) // 8) * 8, ((h // 2 + 7) // 8) * 8
        stride = lw + 2 * mx
        stride += (32 - stride % 32) % 32
        planes = [np.ascontiguousarray(np.pad(p, ((my, my), (mx, stride - lw - mx)), mode="edge"))
                  for p in self.frame_init_lowres(src, origin, lw, lh)]
        est, cost, mode, rows = self.lowres_intra_estimate(planes[0], (my, mx), lw // 8, lh // 8)
        return est, cost, mode, rows, planes, (stride, lw, lh)

    def lowres_intra_estimate(self, plane, origin, wcu, hcu):
        cost = np.zeros(wcu * hcu, np.int32)
        mode = np.zeros(wcu * hcu, np.uint8)
        rows = np.zeros(hcu, np.int32)
        est = self._f("orc_lowres_intra_estimate")(ptr(plane, *origin), plane.shape[1], wcu, hcu, self.depth, ptr(cost), ptr(mode), ptr(rows))
        return est, cost, mode, rows

    def cutree_propagate(self, w, h, qgSize, fps, avgDuration, isP, referenced, wbp, propIn, intra, lowresCosts, invq, mvs0, mvs1, ref0, ref1, qCompress, qpAq):
        """Lookahead::estimateCUPropagate then cuTreeFinish for frame b.  Arrays per 8x8 lowres block; qpAq per quantisation group.
        Returns (refCosts0, refCosts1, qpCuTreeOffset)."""
        import ctypes as C
        L = po.oracle()
        wcu, hcu = (w // 2 + 7) // 8, (h // 2 + 7) // 8
        r0, r1 = np.array(ref0, np.uint16), np.array(ref1, np.uint16)
        f = L.orc_cutree_propagate
        f.restype, f.argtypes = None, [po.i32] * 4 + [C.c_double] + [po.i32] * 4 + [po.vp] * 8
        f(wcu, hcu, fps[0], fps[1], avgDuration, 1, 1 if isP else 2, referenced, wbp, ptr(propIn), ptr(intra), ptr(lowresCosts), ptr(invq), ptr(mvs0), ptr(mvs1),
          ptr(r0), ptr(r1))
        out = np.array(qpAq, np.float64)
        pfin = np.array(propIn, np.uint16)
        if not referenced:
            pfin[:wcu] = 0                 # the reference zeroes (and re-uses) the first row of an unreferenced frame's propagateCost (:2656-2658)
        g = L.orc_cutree_finish
        g.restype, g.argtypes = None, [po.i32] * 5 + [C.c_double] * 3 + [po.vp] * 5
        g(wcu, hcu, qgSize, fps[0], fps[1], avgDuration, qCompress, 0.0, ptr(intra), ptr(invq), ptr(pfin), ptr(qpAq), ptr(out))
        return r0, r1, out

    def aq_frame(self, yuv, origin, w, h, qgSize, aqMode, aqStrength, weightp):
        """LookaheadTLD::calcAdaptiveQuantFrame on a padded 4:2:0 picture yuv = (Y, Cb, Cr), luma origin (y, x) (chroma at half).
        Returns (blockCount, qpAqOffset float64[], invQscaleFactor int32[], invQscaleFactor8x8 int32[], wpStats uint64[6])."""
        import ctypes as C
        lw, lh = ((w // 2 + 7) // 8) * 8, ((h // 2 + 7) // 8) * 8
        nmax = (lw // 8) * (lh // 8) * 4
        qp, inv, inv8, st = np.zeros(nmax, np.float64), np.zeros(nmax, np.int32), np.zeros(nmax, np.int32), np.zeros(6, np.uint64)
        fn = self._f("orc_aq_frame")
        fn.restype = po.i32
        fn.argtypes = [po.vp, po.vp, po.vp, po.ip, po.ip, po.i32, po.i32, po.i32, po.i32, C.c_double, po.i32, po.vp, po.vp, po.vp, po.vp, po.i32]
        n = fn(ptr(yuv[0], *origin), ptr(yuv[1], origin[0] // 2, origin[1] // 2), ptr(yuv[2], origin[0] // 2, origin[1] // 2), yuv[0].shape[1], yuv[1].shape[1],
               w, h, qgSize, aqMode, aqStrength, weightp, ptr(qp), ptr(inv), ptr(inv8), ptr(st), self.depth)
        return int(n), qp, inv, inv8, st

    def lookahead_cost_p_weightp(self, src0, src1, origin, w, h, mx, my, stats):
        """The P-frame cost pass with --weightp: weightsAnalyse, then the pass against the weighted planes when it decided to weight.
        Returns lookahead_cost_p's tuple + (isWeighted,)."""
        import ctypes as C
        isw, wpl = self.weights_analyse(src0, src1, origin, w, h, mx, my, *stats)
        _, _, _, _, pl0, (stride, lw, lh) = self.lowres_pass(src0, origin, w, h, mx, my)
        _, icost, _, _, pl1, _ = self.lowres_pass(src1, origin, w, h, mx, my)
        if isw:
            pl0 = wpl
        wcu, hcu = lw // 8, lh // 8
        ncu = wcu * hcu
        mvs, mvc = np.zeros((ncu, 2), np.int32), np.zeros(ncu, np.int32)
        lc, rows, imb = np.zeros(ncu, np.uint16), np.zeros(hcu, np.int32), np.zeros(1, np.int32)
        tab = po.mvcost_table(12 + 6 * (self.depth - 8), self.depth)
        refs = (C.c_void_p * 4)(*[ptr(p, my, mx).value for p in pl0])
        est = self._f("orc_lookahead_cost_p")(ptr(pl1[0], my, mx), refs, stride, wcu, hcu, hcu, 1, self.depth,
                                              ptr(icost), po.vp(tab.ctypes.data + 2 * po.MVCOST_CENTRE), ptr(mvs), ptr(mvc), ptr(lc), ptr(rows), ptr(imb))
        return int(est), mvs, mvc, lc, rows, int(imb[0]), icost, int(isw)

    def weights_analyse(self, src0, src1, origin, w, h, mx, my, fencSsd, fencSum, refSsd, refSum):
        """LookaheadTLD::weightsAnalyse for frame 1 (src1) against frame 0 (src0).  Returns (isWeighted, [4 weighted padded lowres planes]);
        the chosen (scale, denominator, offset) are left in self.last_weights."""
        import ctypes as C
        _, _, _, _, pl0, (stride, lw, lh) = self.lowres_pass(src0, origin, w, h, mx, my)
        _, icost, _, _, pl1, _ = self.lowres_pass(src1, origin, w, h, mx, my)
        out = [np.zeros_like(p) for p in pl0]
        chosen = np.zeros(3, np.int32)
        fn = self._f("orc_weights_analyse")
        fn.restype = po.i32
        fn.argtypes = [po.vp, po.vp, po.vp, po.ip, po.ip, po.i32, po.i32, po.i32, po.vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, po.vp, po.i32]
        rb = (C.c_void_p * 4)(*[ptr(p).value for p in pl0])
        ob = (C.c_void_p * 4)(*[ptr(p).value for p in out])
        isw = fn(ptr(pl1[0], my, mx), rb, ob, stride, my * stride + mx, lh + 2 * my, lw, lh, ptr(icost), fencSsd, fencSum, refSsd, refSum, ptr(chosen), self.depth)
        self.last_weights = tuple(int(v) for v in chosen)
        for p in out:
            p[:, lw + 2 * mx:] = 0                      # the stride-alignment tail of a row is not part of the padded plane
        return int(isw), out

    def lookahead_cost_p(self, src0, src1, origin, w, h, mx, my, rows_per_slice, num_slices):
        """Lowres::init of both pictures, intra estimate of the second, then the P-frame cost pass (frame 1 referencing frame 0).
        Returns (costEst, mvs[ncu,2], mvCosts, lowresCosts, rowSatds, intraMbs, intraCost)."""
        import ctypes as C
        _, _, _, _, pl0, (stride, lw, lh) = self.lowres_pass(src0, origin, w, h, mx, my)
        _, icost, _, _, pl1, _ = self.lowres_pass(src1, origin, w, h, mx, my)
        wcu, hcu = lw // 8, lh // 8
        ncu = wcu * hcu
        mvs, mvc = np.zeros((ncu, 2), np.int32), np.zeros(ncu, np.int32)
        lc, rows, imb = np.zeros(ncu, np.uint16), np.zeros(hcu, np.int32), np.zeros(1, np.int32)
        tab = po.mvcost_table(12 + 6 * (self.depth - 8), self.depth)
        refs = (C.c_void_p * 4)(*[ptr(p, my, mx).value for p in pl0])
        est = self._f("orc_lookahead_cost_p")(ptr(pl1[0], my, mx), refs, stride, wcu, hcu, rows_per_slice, num_slices, self.depth,
                                              ptr(icost), po.vp(tab.ctypes.data + 2 * po.MVCOST_CENTRE), ptr(mvs), ptr(mvc), ptr(lc), ptr(rows), ptr(imb))
        return int(est), mvs, mvc, lc, rows, int(imb[0]), icost

    def motion_estimate_chroma(self, ref, src, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, method, subme, qp):
        """ref / src: (Y, Cb, Cr) padded planes (chroma at half size, half margins); (bx, by) luma position in the padded plane."""
        fenc = np.zeros((64, 64), self.pix)
        fenc[:h, :w] = src[0][by:by + h, bx:bx + w]
        fc = [np.zeros((32, 32), self.pix) for _ in range(2)]
        for k in range(2):
            fc[k][:h // 2, :w // 2] = src[1 + k][by // 2:by // 2 + h // 2, bx // 2:bx // 2 + w // 2]
        cost = self.mvcost_table(qp)
        a = [np.array(v, np.int32) for v in (mvmin, mvmax, qmvp)]
        cand = np.array(mvc, np.int32).reshape(-1)
        out = np.zeros(2, np.int32)
        c = self._f("orc_motion_estimate_chroma")(ptr(ref[0]), ref[0].shape[1], ptr(ref[1]), ptr(ref[2]), ref[1].shape[1], bx, by,
                                                  ptr(fenc), ptr(fc[0]), ptr(fc[1]), w, h, ptr(a[0]), ptr(a[1]), ptr(a[2]),
                                                  len(mvc), ptr(cand) if len(mvc) else None, merange, method, subme,
                                                  po.vp(cost.ctypes.data + 2 * po.MVCOST_CENTRE), self.depth, ptr(out))
        return c, (int(out[0]), int(out[1]))

    def lookahead_cost_b(self, src0, src1, src2, origin, w, h, mx, my, rows_per_slice, num_slices, prefill_l0):
        """B-frame cost of picture 1 between pictures 0 and 2 (estimateFrameCost(p0=0, p1=2, b=1)).  prefill_l0: list 0 was already
        searched by the P estimate (0, 1, 1) and is reused.  Returns (costEst scaled by 100/130, mvs0, mvCosts0, mvs1, mvCosts1,
        lowresCosts, rowSatds)."""
        import ctypes as C
        pl = [self.lowres_pass(s_, origin, w, h, mx, my) for s_ in (src0, src1, src2)]
        stride, lw, lh = pl[0][5]
        icost = pl[1][1]
        wcu, hcu = lw // 8, lh // 8
        ncu = wcu * hcu
        mvs = [np.zeros((ncu, 2), np.int32) for _ in range(2)]
        mvc = [np.zeros(ncu, np.int32) for _ in range(2)]
        lc, rows, imb = np.zeros(ncu, np.uint16), np.zeros(hcu, np.int32), np.zeros(1, np.int32)
        tab = po.mvcost_table(12 + 6 * (self.depth - 8), self.depth)
        tabp = po.vp(tab.ctypes.data + 2 * po.MVCOST_CENTRE)
        refs0 = (C.c_void_p * 4)(*[ptr(p, my, mx).value for p in pl[0][4]])
        refs1 = (C.c_void_p * 4)(*[ptr(p, my, mx).value for p in pl[2][4]])
        fenc = ptr(pl[1][4][0], my, mx)
        do = np.array([1, 1], np.int32)
        if prefill_l0:
            # (the reference shim pre-fills through singleCost(), i.e. the serial, unsliced loop)
            self._f("orc_lookahead_cost_p")(fenc, refs0, stride, wcu, hcu, hcu, 1, self.depth, ptr(icost), tabp,
                                            ptr(mvs[0]), ptr(mvc[0]), ptr(lc), ptr(rows), ptr(imb))
            do[0] = 0
        est = self._f("orc_lookahead_cost_b")(fenc, refs0, refs1, stride, wcu, hcu, rows_per_slice, num_slices, self.depth, tabp, ptr(do),
                                              ptr(mvs[0]), ptr(mvc[0]), ptr(mvs[1]), ptr(mvc[1]), ptr(lc), ptr(rows))
        return int(est) * 100 // 130, mvs[0], mvc[0], mvs[1], mvc[1], lc, rows

    def pred_inter_bi(self, ref0, ref1, bx, by, w, h, mv0, mv1):
        """ref0 / ref1 = (Y, Cb, Cr) padded planes; (bx, by) absolute luma position.  Returns (Y[h,w], Cb, Cr)."""
        import ctypes as C
        L = po.oracle()
        fn = getattr(L, "orc_pred_inter_bi_%s" % self.s)
        fn.restype = None
        fn.argtypes = [po.vp, po.vp, po.ip, po.ip, po.i32, po.i32, po.i32, po.i32, po.vp, po.vp, po.vp, po.ip, po.vp, po.vp, po.ip, po.i32]
        r0 = (C.c_void_p * 3)(*[ptr(p).value for p in ref0])
        r1 = (C.c_void_p * 3)(*[ptr(p).value for p in ref1])
        y, cb, cr = np.zeros((h, w), self.pix), np.zeros((h // 2, w // 2), self.pix), np.zeros((h // 2, w // 2), self.pix)
        a0, a1 = np.array(mv0, np.int32), np.array(mv1, np.int32)          # named: a temporary would be freed before the call reads it
        fn(r0, r1, ref0[0].shape[1], ref0[1].shape[1], bx, by, w, h, ptr(a0), ptr(a1), ptr(y), w, ptr(cb), ptr(cr), w // 2, self.depth)
        return y, cb, cr

    def motion_compensation(self, ref0, ref1, bx, by, w, h, mv0, mv1, wp0, wp1, sliceP=0, uniList=0):
        """Predict::motionCompensation for one PU: ref1 None = uni-prediction; wp = [(inputWeight, inputOffset, log2WeightDenom, wtPresent)] * 3
        per list or None (weighted prediction off).  Returns (Y[h,w], Cb, Cr)."""
        import ctypes as C
        L = po.oracle()
        fn = getattr(L, "orc_motion_compensation_%s" % self.s)
        fn.restype = None
        fn.argtypes = [po.vp, po.vp, po.ip, po.ip, po.i32, po.i32, po.i32, po.i32, po.vp, po.vp, po.vp, po.vp, po.vp, po.ip, po.vp, po.vp, po.ip, po.i32]
        r0 = (C.c_void_p * 3)(*[ptr(p).value for p in ref0])
        r1 = (C.c_void_p * 3)(*[ptr(p).value for p in ref1]) if ref1 is not None else None
        H, S = ref0[0].shape
        y, cb, cr = np.zeros((H, S), self.pix), np.zeros(ref0[1].shape, self.pix), np.zeros(ref0[1].shape, self.pix)
        a0 = np.array(mv0, np.int32)
        a1 = np.array(mv1 if mv1 is not None else (0, 0), np.int32)
        w0 = np.array(wp0, np.int32).reshape(-1) if wp0 is not None else None
        w1 = np.array(wp1, np.int32).reshape(-1) if wp1 is not None else None
        fn(r0, r1, S, ref0[1].shape[1], bx, by, w, h, ptr(a0), ptr(a1), ptr(w0) if w0 is not None else None, ptr(w1) if w1 is not None else None,
           ptr(y), S, ptr(cb), ptr(cr), ref0[1].shape[1], self.depth)
        return (np.ascontiguousarray(y[by:by + h, bx:bx + w]), np.ascontiguousarray(cb[by // 2:(by + h) // 2, bx // 2:(bx + w) // 2]),
                np.ascontiguousarray(cr[by // 2:(by + h) // 2, bx // 2:(bx + w) // 2]))

    # ---- coefficient-scan cost primitives (dct.cpp:757-1006)
    def _cf(self, name, restype, argtypes):
        import ctypes as C
        fn = getattr(po.oracle(), name)
        fn.restype, fn.argtypes = restype, argtypes
        return fn

    def scan_order(self, stype, log2):
        out = np.zeros(1 << (2 * log2), np.uint16)
        self._cf("orc_scan_order", None, [po.i32, po.i32, po.vp])(stype, log2, ptr(out))
        return out

    def scan4x4(self, stype):
        return self.scan_order(stype, 2)

    def entropy_state_bits(self):
        return entropy_state_bits_fixture()

    def scan_pos_last(self, log2, stype, coeff):
        import ctypes as C
        scan = self.scan_order(stype, log2)
        sign, flag, num = np.zeros(64, np.uint16), np.zeros(64, np.uint16), np.zeros(64, np.uint8)
        c = np.ascontiguousarray(coeff, np.int16).reshape(-1)
        last = self._cf("orc_scanPosLast", po.i32, [po.vp, po.vp, po.vp, po.vp, po.vp, po.i32])(ptr(scan), ptr(c), ptr(sign), ptr(flag), ptr(num),
                                                                                              int(np.count_nonzero(c)))
        return int(last), sign, flag, num

    def find_pos_first_last(self, tu, cgx, cgy, stype):
        import ctypes as C
        tu = np.ascontiguousarray(tu, np.int16)
        scan = self.scan4x4(stype)
        return int(self._cf("orc_findPosFirstLast", C.c_uint32, [po.vp, po.ip, po.vp])(ptr(tu, cgy * 4, cgx * 4), tu.shape[1], ptr(scan)))

    def cost_coeff_nxn(self, tu, log2, stype, cgIdx, scanPosSigOff, pattern, offset, ctx):
        """One coefficient group (the cgIdx-th of the TU's scan) from scanPosSigOff down: (bits, absCoeff[16] with the unwritten tail zero, ctx after)."""
        import ctypes as C
        tu = np.ascontiguousarray(tu, np.int16)
        scan, scan4 = self.scan_order(stype, log2), self.scan4x4(stype if log2 <= 3 else 0)
        base = int(scan[cgIdx * 16])
        mask = 0
        for k in range(scanPosSigOff + 1):
            mask = mask * 2 + int(tu.reshape(-1)[scan[cgIdx * 16 + k]] != 0)
        absC, ctx2 = np.zeros(16, np.uint16), np.array(ctx, np.uint8)
        tab = sig_ctx_table(log2, pattern)
        bits = self._cf("orc_costCoeffNxN", C.c_uint32, [po.vp, po.vp, po.ip, po.vp, po.vp, C.c_uint32, po.vp, po.i32, po.i32, po.i32, po.vp])
        sb = self.entropy_state_bits()
        first = 1 if scanPosSigOff < 15 else 0
        r = bits(ptr(scan4), ptr(tu.reshape(-1), 0, base), tu.shape[1], ptr(absC, 0, first), ptr(tab), mask, ptr(ctx2), offset, scanPosSigOff, cgIdx * 16, ptr(sb))
        return int(r), absC, ctx2

    def cost_coeff_remain(self, absCoeff, numNonZero, idx):
        import ctypes as C
        a = np.ascontiguousarray(absCoeff, np.uint16)
        return int(self._cf("orc_costCoeffRemain", C.c_uint32, [po.vp, po.i32, po.i32])(ptr(a), numNonZero, idx))

    def cost_c1c2_flag(self, absCoeff, numC1Flag, ctx, ctxOffset):
        import ctypes as C
        a, ctx2, sb = np.ascontiguousarray(absCoeff, np.uint16), np.array(ctx, np.uint8), self.entropy_state_bits()
        r = self._cf("orc_costC1C2Flag", C.c_uint32, [po.vp, po.ip, po.vp, po.ip, po.vp])(ptr(a), numC1Flag, ptr(ctx2), ctxOffset, ptr(sb))
        return int(r), ctx2

    # ---- in-loop filter primitives (loopfilter.cpp, sao.cpp:1762-1925).  pos = (y, x) of the primitive's `rec` / `src` pointer in `plane`
    def _lf(self, name, argtypes):
        fn = getattr(po.oracle(), "%s_%s" % (name, self.s))
        fn.restype, fn.argtypes = None, argtypes
        return fn

    def deblock_ctu_edge(self, planes, ctu, edgeDir, edge, bs, qp, bypass, betaDiv2, tcDiv2, cbOff, crOff, doLuma=1, doChroma=1):
        """One inner edge of the 64x64 CTU at luma position ctu = (y, x): bs / qp / bypass are [16, 16] arrays per 4x4 unit (raster).
        Restated as the unit loops of Deblock::edgeFilterLuma / edgeFilterChroma over the per-unit oracle functions."""
        y, cb, cr = [p.copy() for p in planes]
        S, SC = y.shape[1], cb.shape[1]
        fl = self._lf("orc_deblock_luma_unit", [po.vp, po.ip, po.ip] + [po.i32] * 9)
        fc = self._lf("orc_deblock_chroma_unit", [po.vp, po.vp, po.ip, po.ip] + [po.i32] * 10)
        chk = 0 if bypass is None else 1
        byp = bypass if bypass is not None else np.zeros((16, 16), np.uint8)
        if doLuma:
            for idx in range(16):
                (qx, qy), (px, py) = ((edge, idx), (edge - 1, idx)) if edgeDir == 0 else ((idx, edge), (idx, edge - 1))
                pos = (ctu[0] + qy * 4, ctu[1] + qx * 4)
                step, off = (S, 1) if edgeDir == 0 else (1, S)
                fl(ptr(y, *pos), step, off, int(bs[qy, qx]), int(qp[py, px]), int(qp[qy, qx]), chk, int(byp[py, px]), int(byp[qy, qx]), betaDiv2, tcDiv2, self.depth)
        if doChroma:
            for idx in range(8):
                (qx, qy), (px, py) = ((edge, 2 * idx), (edge - 1, 2 * idx)) if edgeDir == 0 else ((2 * idx, edge), (2 * idx, edge - 1))
                pos = (ctu[0] // 2 + (idx * 4 if edgeDir == 0 else edge * 2), ctu[1] // 2 + (edge * 2 if edgeDir == 0 else idx * 4))
                step, off = (SC, 1) if edgeDir == 0 else (1, SC)
                fc(ptr(cb, *pos), ptr(cr, *pos), step, off, int(bs[qy, qx]), int(qp[py, px]), int(qp[qy, qx]), chk, int(byp[py, px]), int(byp[qy, qx]), tcDiv2,
                   cbOff, crOff, self.depth)
        return y, cb, cr

    def pel_filter_luma_strong(self, plane, pos, edgeDir, tcP, tcQ):
        p = plane.copy()
        step, off = (p.shape[1], 1) if edgeDir == 0 else (1, p.shape[1])
        self._lf("orc_pelFilterLumaStrong", [po.vp, po.ip, po.ip, po.i32, po.i32])(ptr(p, *pos), step, off, tcP, tcQ)
        return p

    def pel_filter_chroma(self, plane, pos, edgeDir, tc, maskP, maskQ):
        p = plane.copy()
        step, off = (p.shape[1], 1) if edgeDir == 0 else (1, p.shape[1])
        self._lf("orc_pelFilterChroma", [po.vp, po.ip, po.ip, po.i32, po.i32, po.i32, po.i32])(ptr(p, *pos), step, off, tc, maskP, maskQ, self.depth)
        return p

    def sao_sign(self, a, b):
        d = np.zeros(len(a), np.int8)
        self._lf("orc_saoSign", [po.vp, po.vp, po.vp, po.i32])(ptr(d), ptr(a), ptr(b), len(a))
        return d

    def sao_e0(self, plane, pos, offsetEo, width, signLeft):
        p, eo, sl = plane.copy(), np.array(offsetEo, np.int8), np.array(signLeft, np.int8)
        self._lf("orc_saoCuOrgE0", [po.vp, po.vp, po.i32, po.vp, po.ip, po.i32])(ptr(p, *pos), ptr(eo), width, ptr(sl), p.shape[1], self.depth)
        return p

    def sao_e1(self, plane, pos, up, offsetEo, width, rows):
        p, eo, u = plane.copy(), np.array(offsetEo, np.int8), np.array(up, np.int8)
        self._lf("orc_saoCuOrgE1", [po.vp, po.vp, po.vp, po.ip, po.i32, po.i32, po.i32])(ptr(p, *pos), ptr(u), ptr(eo), p.shape[1], width, rows, self.depth)
        return p, u

    def sao_e2(self, plane, pos, bufft, buff1, offsetEo, width):
        p, eo, bt, b1 = plane.copy(), np.array(offsetEo, np.int8), np.array(bufft, np.int8), np.array(buff1, np.int8)
        self._lf("orc_saoCuOrgE2", [po.vp, po.vp, po.vp, po.vp, po.i32, po.ip, po.i32])(ptr(p, *pos), ptr(bt), ptr(b1), ptr(eo), width, p.shape[1], self.depth)
        return p, bt

    def sao_e3(self, plane, pos, upfull, offsetEo, startX, endX):
        p, eo, u = plane.copy(), np.array(offsetEo, np.int8), np.array(upfull, np.int8)
        self._lf("orc_saoCuOrgE3", [po.vp, po.vp, po.vp, po.ip, po.i32, po.i32, po.i32])(ptr(p, *pos), ptr(u, 0, 1), ptr(eo), p.shape[1], startX, endX, self.depth)
        return p, u

    def sao_b0(self, plane, pos, offset32, w, h):
        p, o = plane.copy(), np.array(offset32, np.int8)
        self._lf("orc_saoCuOrgB0", [po.vp, po.vp, po.i32, po.i32, po.ip, po.i32])(ptr(p, *pos), ptr(o), w, h, p.shape[1], self.depth)
        return p

    def sao_stats(self, kind, diff, plane, pos, endX, endY, stats, count, up1full, uptfull):
        """kind 0 BO, 1..4 E0..E3; diff [64, 64] int16; up*full carry one element before index 0.  Returns (stats, count, up1full, uptfull)."""
        st, ct, u1, ut = np.array(stats, np.int32), np.array(count, np.int32), np.array(up1full, np.int8), np.array(uptfull, np.int8)
        d = np.ascontiguousarray(diff, np.int16)
        S = plane.shape[1]
        if kind == 0:
            self._lf("orc_saoCuStatsBO", [po.vp, po.vp, po.ip, po.i32, po.i32, po.vp, po.vp, po.i32])(ptr(d), ptr(plane, *pos), S, endX, endY, ptr(st), ptr(ct), self.depth)
        elif kind == 1:
            self._lf("orc_saoCuStatsE0", [po.vp, po.vp, po.ip, po.i32, po.i32, po.vp, po.vp])(ptr(d), ptr(plane, *pos), S, endX, endY, ptr(st), ptr(ct))
        elif kind == 2:
            self._lf("orc_saoCuStatsE1", [po.vp, po.vp, po.ip, po.vp, po.i32, po.i32, po.vp, po.vp])(ptr(d), ptr(plane, *pos), S, ptr(u1, 0, 1), endX, endY, ptr(st), ptr(ct))
        elif kind == 3:
            self._lf("orc_saoCuStatsE2", [po.vp, po.vp, po.ip, po.vp, po.vp, po.i32, po.i32, po.vp, po.vp])(ptr(d), ptr(plane, *pos), S, ptr(u1, 0, 1), ptr(ut, 0, 1), endX, endY,
                                                                                                          ptr(st), ptr(ct))
        else:
            self._lf("orc_saoCuStatsE3", [po.vp, po.vp, po.ip, po.vp, po.i32, po.i32, po.vp, po.vp])(ptr(d), ptr(plane, *pos), S, ptr(u1, 0, 1), endX, endY, ptr(st), ptr(ct))
        return st, ct, u1, ut

    # ---- weighted prediction, downscales, transpose
    def weight_pp(self, a, ao, w, h, w0, rnd, shift, offset):
        d = np.zeros_like(a)
        self._f("orc_weight_pp")(ptr(a, *ao), ptr(d, *ao), a.shape[1], w, h, w0, rnd, shift, offset, self.depth)
        return np.ascontiguousarray(d[ao[0]:ao[0] + h, ao[1]:ao[1] + w])

    def weight_sp(self, a, ao, w, h, w0, rnd, shift, offset):
        d = np.zeros((h, w), self.pix)
        self._f("orc_weight_sp")(ptr(a, *ao), ptr(d), a.shape[1], w, w, h, w0, rnd, shift, offset, self.depth)
        return d

    def scale1d_128to64(self, line):
        d = np.zeros(128, self.pix)
        self._f("orc_scale1d_128to64")(ptr(d), ptr(line))
        return d

    def scale2d_64to32(self, a, ao):
        d = np.zeros((32, 32), self.pix)
        self._f("orc_scale2d_64to32")(ptr(d), ptr(a, *ao), a.shape[1])
        return d

    def transpose(self, size, a, ao):
        d = np.zeros((size, size), self.pix)
        self._f("orc_transpose")(ptr(d), ptr(a, *ao), a.shape[1], size)
        return d


class Ref(_Base):
    name = "reference"

    def __init__(self, depth):
        super().__init__(depth)
        self.L = po.ref(depth)

    def sad(self, w, h, a, ao, b, bo):
        return self.L.ref_sad(part_of(w, h), ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1])

    def sad_xn(self, w, h, fenc, ref, offs):
        res = np.zeros(len(offs), np.int32)
        ps = [ptr(ref, *o) for o in offs]
        if len(offs) == 3:
            self.L.ref_sad_x3(part_of(w, h), ptr(fenc), ps[0], ps[1], ps[2], ref.shape[1], ptr(res))
        else:
            self.L.ref_sad_x4(part_of(w, h), ptr(fenc), ps[0], ps[1], ps[2], ps[3], ref.shape[1], ptr(res))
        return res

    def satd(self, w, h, a, ao, b, bo):
        return self.L.ref_satd(part_of(w, h), ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1])

    def sa8d(self, size, a, ao, b, bo):
        return self.L.ref_sa8d(cu_of(size), ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1])

    def sse_pp(self, size, a, ao, b, bo):
        return self.L.ref_sse_pp(cu_of(size), ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1])

    def sse_ss(self, size, a, ao, b, bo):
        return self.L.ref_sse_ss(cu_of(size), ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1])

    def ssd_s(self, size, a, ao):
        return self.L.ref_ssd_s(cu_of(size), ptr(a, *ao), a.shape[1])

    def psy_cost_pp(self, size, a, ao, b, bo):
        return self.L.ref_psy_cost_pp(cu_of(size), ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1])

    def var(self, size, a, ao):
        return self.L.ref_var(cu_of(size), ptr(a, *ao), a.shape[1])

    def sub_ps(self, size, a, ao, b, bo):
        d = np.zeros((size, size), np.int16)
        self.L.ref_sub_ps(cu_of(size), ptr(d), size, ptr(a, *ao), ptr(b, *bo), a.shape[1], b.shape[1])
        return d

    def add_ps(self, size, a, ao, r, ro):
        d = np.zeros((size, size), self.pix)
        self.L.ref_add_ps(cu_of(size), ptr(d), size, ptr(a, *ao), ptr(r, *ro), a.shape[1], r.shape[1])
        return d

    def addAvg(self, w, h, a, ao, b, bo):
        d = np.zeros((h, w), self.pix)
        self.L.ref_addAvg(part_of(w, h), ptr(a, *ao), ptr(b, *bo), ptr(d), a.shape[1], b.shape[1], w)
        return d

    def pixelavg_pp(self, w, h, a, ao, b, bo):
        d = np.zeros((h, w), self.pix)
        self.L.ref_pixelavg_pp(part_of(w, h), ptr(d), w, ptr(a, *ao), a.shape[1], ptr(b, *bo), b.shape[1])
        return d

    def p2s(self, w, h, a, ao):
        d = np.zeros((h, w), np.int16)
        self.L.ref_p2s(0, part_of(w, h), ptr(a, *ao), a.shape[1], ptr(d), w)
        return d

    def cpy2Dto1D_shl(self, size, a, ao, shift):
        d = np.zeros(size * size, np.int16)
        self.L.ref_cpy2Dto1D_shl(cu_of(size), ptr(d), ptr(a, *ao), a.shape[1], shift)
        return d

    def cpy2Dto1D_shr(self, size, a, ao, shift):
        d = np.zeros(size * size, np.int16)
        self.L.ref_cpy2Dto1D_shr(cu_of(size), ptr(d), ptr(a, *ao), a.shape[1], shift)
        return d

    def cpy1Dto2D_shl(self, size, a, shift):
        d = np.zeros((size, size), np.int16)
        self.L.ref_cpy1Dto2D_shl(cu_of(size), ptr(d), ptr(a), size, shift)
        return d

    def cpy1Dto2D_shr(self, size, a, shift):
        d = np.zeros((size, size), np.int16)
        self.L.ref_cpy1Dto2D_shr(cu_of(size), ptr(d), ptr(a), size, shift)
        return d

    def copy_cnt(self, size, a, ao):
        d = np.zeros(size * size, np.int16)
        n = self.L.ref_copy_cnt(cu_of(size), ptr(d), ptr(a, *ao), a.shape[1])
        return d, n

    def count_nonzero(self, size, a):
        return self.L.ref_count_nonzero(cu_of(size), ptr(a))

    def dct(self, size, a, ao):
        d = np.zeros(size * size, np.int16)
        self.L.ref_dct(cu_of(size), ptr(a, *ao), ptr(d), a.shape[1])
        return d

    def idct(self, size, a):
        d = np.zeros((size, size), np.int16)
        self.L.ref_idct(cu_of(size), ptr(a), ptr(d), size)
        return d

    def dst4(self, a, ao):
        d = np.zeros(16, np.int16)
        self.L.ref_dst4(ptr(a, *ao), ptr(d), a.shape[1])
        return d

    def idst4(self, a):
        d = np.zeros((4, 4), np.int16)
        self.L.ref_idst4(ptr(a), ptr(d), 4)
        return d

    def quant(self, coef, qc, qbits, add):
        n = coef.size
        du = np.zeros(n, np.int32)
        q = np.zeros(n, np.int16)
        ns = self.L.ref_quant(ptr(coef), ptr(qc), ptr(du), ptr(q), qbits, add, n)
        return q, du, ns

    def nquant(self, coef, qc, qbits, add):
        n = coef.size
        q = np.zeros(n, np.int16)
        ns = self.L.ref_nquant(ptr(coef), ptr(qc), ptr(q), qbits, add, n)
        return q, ns

    def dequant_normal(self, q, scale, shift):
        c = np.zeros(q.size, np.int16)
        self.L.ref_dequant_normal(ptr(q), ptr(c), q.size, scale, shift)
        return c

    def dequant_scaling(self, q, dq, per, shift):
        c = np.zeros(q.size, np.int16)
        self.L.ref_dequant_scaling(ptr(q), ptr(dq), ptr(c), q.size, per, shift)
        return c

    def denoise_dct(self, coef, ressum, offset):
        c, r = coef.copy(), ressum.copy()
        self.L.ref_denoise_dct(ptr(c), ptr(r), ptr(offset), c.size)
        return c, r

    def rdoquant(self, kind, size, resi, fenc, psyscale, blkpos):
        cu = cu_of(size)
        cu_ = np.full(size * size, 7, np.int64)
        tot = np.array([11, 13], np.int64)
        ps = np.array([psyscale], np.int64)
        tu, tr = po.vp(tot.ctypes.data), po.vp(tot.ctypes.data + 8)
        if kind == "nonpsy":
            self.L.ref_nonpsy_rdoquant(cu, ptr(resi), ptr(cu_), tu, tr, blkpos)
        elif kind == "psy":
            self.L.ref_psy_rdoquant(cu, ptr(resi), ptr(fenc), ptr(cu_), tu, tr, ptr(ps), blkpos)
        elif kind == "psy1":
            self.L.ref_psy_rdoquant_1p(cu, ptr(resi), ptr(cu_), tu, tr, blkpos)
        else:
            self.L.ref_psy_rdoquant_2p(cu, ptr(resi), ptr(fenc), ptr(cu_), tu, tr, ptr(ps), blkpos)
        return cu_, tot

    def interp(self, kind, chroma, w, h, src, so, idx, idy=0, ext=0):
        # chroma tables are indexed by the LUMA partition whose 4:2:0 chroma block is (w, h)
        part = part_of(w * 2, h * 2) if chroma else part_of(w, h)
        N = 4 if chroma else 8
        c = 1 if chroma else 0
        L = self.L
        if kind == "hpp":
            out = np.zeros((h, w), self.pix)
            L.ref_interp_hpp(c, part, ptr(src, *so), src.shape[1], ptr(out), w, idx)
        elif kind == "hps":
            out = np.zeros((h + (N - 1 if ext else 0), w), np.int16)
            L.ref_interp_hps(c, part, ptr(src, *so), src.shape[1], ptr(out), w, idx, ext)
        elif kind == "vpp":
            out = np.zeros((h, w), self.pix)
            L.ref_interp_vpp(c, part, ptr(src, *so), src.shape[1], ptr(out), w, idx)
        elif kind == "vps":
            out = np.zeros((h, w), np.int16)
            L.ref_interp_vps(c, part, ptr(src, *so), src.shape[1], ptr(out), w, idx)
        elif kind == "vsp":
            out = np.zeros((h, w), self.pix)
            L.ref_interp_vsp(c, part, ptr(src, *so), src.shape[1], ptr(out), w, idx)
        elif kind == "vss":
            out = np.zeros((h, w), np.int16)
            L.ref_interp_vss(c, part, ptr(src, *so), src.shape[1], ptr(out), w, idx)
        elif kind == "hvpp":
            out = np.zeros((h, w), self.pix)
            L.ref_interp_hvpp(part, ptr(src, *so), src.shape[1], ptr(out), w, idx, idy)
        else:
            raise ValueError(kind)
        return out

    def mvcost_table(self, qp):
        t = np.zeros(4 * 32768 + 1, np.uint16)
        self.L.ref_mvcost_table(qp, ptr(t))
        return t

    def motion_estimate(self, refplane, fencplane, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, method, subme, qp):
        assert refplane.shape == fencplane.shape
        a = [np.array(v, np.int32) for v in (mvmin, mvmax, qmvp)]
        cand = np.array(mvc, np.int32).reshape(-1)
        out = np.zeros(2, np.int32)
        c = self.L.ref_motion_estimate(ptr(refplane), ptr(fencplane), refplane.shape[1], bx, by, w, h,
                                       ptr(a[0]), ptr(a[1]), ptr(a[2]), len(mvc), ptr(cand) if len(mvc) else None,
                                       merange, method, subme, qp, ptr(out))
        return c, (int(out[0]), int(out[1]))

    def integral_planes(self, buf, pad):
        """The reference's own integral_inith / integral_initv primitives driven as FrameFilter::computeMEIntegral drives them; pad = (padY, padX)."""
        pady, padx = pad
        planes = np.zeros((12,) + buf.shape, np.uint32)
        self.L.ref_integral_planes(ptr(buf, pady, padx), buf.shape[1], buf.shape[0] - 2 * pady, padx, pady, ptr(planes), buf.shape[0] * buf.shape[1])
        return [planes[k] for k in range(12)]

    def motion_estimate_sea(self, refplane, fencplane, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, subme, qp, planes=None, pad=(80, 96)):
        assert refplane.shape == fencplane.shape
        pady, padx = pad
        if planes is None:
            planes = self.integral_planes(refplane, pad)
        pl = np.ascontiguousarray(np.stack(planes))
        a = [np.array(v, np.int32) for v in (mvmin, mvmax, qmvp)]
        cand = np.array(mvc, np.int32).reshape(-1)
        out = np.zeros(2, np.int32)
        # the shim addresses planes from the picture origin: hand it origin-relative PU coordinates
        c = self.L.ref_motion_estimate_sea(ptr(refplane, pady, padx), ptr(fencplane, pady, padx), refplane.shape[1], ptr(pl), refplane.shape[0] * refplane.shape[1],
                                           padx, pady, bx - padx, by - pady, w, h, ptr(a[0]), ptr(a[1]), ptr(a[2]), len(mvc), ptr(cand) if len(mvc) else None,
                                           merange, subme, qp, ptr(out))
        return c, (int(out[0]), int(out[1]))

    # ---- intra prediction / lookahead lowres
    def intra_filter(self, n, nb):
        out = np.zeros(4 * n + 1, self.pix)
        self.L.ref_intra_filter(cu_of(n), ptr(nb), ptr(out))
        return out

    def intra_pred(self, n, mode, nb, bfilter):
        out = np.zeros((n, n), self.pix)
        self.L.ref_intra_pred(cu_of(n), mode, ptr(out), n, ptr(nb), bfilter)
        return out

    def intra_uses_filtered(self, n, mode):
        return 1 if self.L.ref_intra_filter_flags(mode) & n else 0

    def intra_allangs(self, n, nb, nbf, bluma):
        out = np.zeros((33, n, n), self.pix)
        self.L.ref_intra_allangs(cu_of(n), po.vp(out.ctypes.data), ptr(nb), ptr(nbf), bluma)
        return out

    def frame_init_lowres(self, src, origin, w, h):
        out = [np.zeros((h, w), self.pix) for _ in range(4)]
        self.L.ref_frame_init_lowres(ptr(src, *origin), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), src.shape[1], w, w, h)
        return tuple(out)

    def lowres_pass(self, src, origin, w, h, mx, my):
        """The real Lowres::create/init + LookaheadTLD::lowresIntraEstimate on a padded source plane.
        Returns (costEst, intraCost, intraMode, rowSatds, planes[4] padded 2-D, (lumaStride, width, lines))."""
        geom = np.zeros(4, np.int64)
        lw, lh = ((w // 2 + 7) // 8) * 8, ((h // 2 + 7) // 8) * 8
        stride = lw + 2 * mx
        stride += (32 - stride % 32) % 32
        planesize = stride * (lh + 2 * my)
        planes = np.zeros(4 * planesize, self.pix)
        ncu = (lw // 8) * (lh // 8)
        cost = np.zeros(ncu, np.int32)
        mode = np.zeros(ncu, np.uint8)
        rows = np.zeros(lh // 8, np.int32)
        est = self.L.ref_lowres_intra_estimate(ptr(src, *origin), src.shape[1], w, h, mx, my, ptr(geom), ptr(planes), planes.size,
                                               ptr(cost), ptr(mode), ptr(rows))
        assert est >= 0 and (int(geom[0]), int(geom[1]), int(geom[2]), int(geom[3])) == (stride, lw, lh, planesize), geom
        pl = [planes[i * planesize:(i + 1) * planesize].reshape(lh + 2 * my, stride) for i in range(4)]
        return est, cost, mode, rows, pl, (stride, lw, lh)

    def cutree_propagate(self, w, h, qgSize, fps, avgDuration, isP, referenced, wbp, propIn, intra, lowresCosts, invq, mvs0, mvs1, ref0, ref1, qCompress, qpAq):
        m = 80
        pic = np.zeros((h + 2 * m, w + 2 * m), self.pix)
        r0, r1 = np.array(ref0, np.uint16), np.array(ref1, np.uint16)
        out = np.zeros(len(qpAq), np.float64)
        n = self.L.ref_cutree_propagate(ptr(pic, m, m), pic.shape[1], w, h, m, m, qgSize, fps[0], fps[1], float(avgDuration), int(isP), int(referenced), int(wbp),
                                        ptr(propIn), ptr(intra), ptr(lowresCosts), ptr(invq), ptr(mvs0), ptr(mvs1), ptr(r0), ptr(r1), 1, float(qCompress),
                                        ptr(qpAq), ptr(out), len(qpAq))
        assert n == len(intra)
        return r0, r1, out

    def aq_frame(self, yuv, origin, w, h, qgSize, aqMode, aqStrength, weightp):
        lw, lh = ((w // 2 + 7) // 8) * 8, ((h // 2 + 7) // 8) * 8
        nmax = (lw // 8) * (lh // 8) * 4
        qp, inv, inv8, st = np.zeros(nmax, np.float64), np.zeros(nmax, np.int32), np.zeros(nmax, np.int32), np.zeros(6, np.uint64)
        n = self.L.ref_aq_frame(ptr(yuv[0], *origin), ptr(yuv[1], origin[0] // 2, origin[1] // 2), ptr(yuv[2], origin[0] // 2, origin[1] // 2), yuv[0].shape[1],
                                yuv[1].shape[1], w, h, origin[1], origin[0], qgSize, aqMode, float(aqStrength), weightp, ptr(qp), ptr(inv), ptr(inv8), ptr(st))
        return int(n), qp, inv, inv8, st

    def lookahead_cost_p_weightp(self, src0, src1, origin, w, h, mx, my, stats):
        lw, lh = ((w // 2 + 7) // 8) * 8, ((h // 2 + 7) // 8) * 8
        wcu, hcu = lw // 8, lh // 8
        ncu = wcu * hcu
        mvs, mvc = np.zeros((ncu, 2), np.int32), np.zeros(ncu, np.int32)
        lc, rows, imb = np.zeros(ncu, np.uint16), np.zeros(hcu, np.int32), np.zeros(1, np.int32)
        icost, isw = np.zeros(ncu, np.int32), np.zeros(1, np.int32)
        st = np.array(stats, np.uint64)
        est = self.L.ref_lookahead_cost_p_weightp(ptr(src0, *origin), ptr(src1, *origin), src0.shape[1], w, h, mx, my, ptr(st), ptr(mvs), ptr(mvc), ptr(lc),
                                                  ptr(rows), ptr(imb), ptr(icost), ptr(isw))
        return int(est), mvs, mvc, lc, rows, int(imb[0]), icost, int(isw[0])

    def weights_analyse(self, src0, src1, origin, w, h, mx, my, fencSsd, fencSum, refSsd, refSum):
        lw, lh = ((w // 2 + 7) // 8) * 8, ((h // 2 + 7) // 8) * 8
        stride = lw + 2 * mx
        stride += (32 - stride % 32) % 32
        planesize = stride * (lh + 2 * my)
        planes = np.zeros(4 * planesize, self.pix)
        delta = np.zeros(1, np.float64)
        isw = self.L.ref_weights_analyse(ptr(src0, *origin), ptr(src1, *origin), src0.shape[1], w, h, mx, my, fencSsd, fencSum, refSsd, refSum,
                                         ptr(planes), planes.size, ptr(delta))
        assert isw >= 0
        self.last_cost_delta = float(delta[0])
        out = [planes[i * planesize:(i + 1) * planesize].reshape(lh + 2 * my, stride).copy() for i in range(4)]
        for p in out:
            p[:, lw + 2 * mx:] = 0
        return int(isw), out

    def lookahead_cost_p(self, src0, src1, origin, w, h, mx, my, rows_per_slice, num_slices):
        lw, lh = ((w // 2 + 7) // 8) * 8, ((h // 2 + 7) // 8) * 8
        wcu, hcu = lw // 8, lh // 8
        ncu = wcu * hcu
        mvs, mvc = np.zeros((ncu, 2), np.int32), np.zeros(ncu, np.int32)
        lc, rows, imb = np.zeros(ncu, np.uint16), np.zeros(hcu, np.int32), np.zeros(1, np.int32)
        icost = np.zeros(ncu, np.int32)
        assert src0.shape == src1.shape
        est = self.L.ref_lookahead_cost_p(ptr(src0, *origin), ptr(src1, *origin), src0.shape[1], w, h, mx, my, rows_per_slice, num_slices,
                                          ptr(mvs), ptr(mvc), ptr(lc), ptr(rows), ptr(imb), ptr(icost))
        return int(est), mvs, mvc, lc, rows, int(imb[0]), icost

    def motion_estimate_chroma(self, ref, src, bx, by, w, h, mvmin, mvmax, qmvp, mvc, merange, method, subme, qp):
        a = [np.array(v, np.int32) for v in (mvmin, mvmax, qmvp)]
        cand = np.array(mvc, np.int32).reshape(-1)
        out = np.zeros(2, np.int32)
        assert ref[0].shape == src[0].shape and ref[1].shape == src[1].shape
        c = self.L.ref_motion_estimate_chroma(ptr(ref[0]), ptr(ref[1]), ptr(ref[2]), ptr(src[0]), ptr(src[1]), ptr(src[2]),
                                              ref[0].shape[1], ref[1].shape[1], bx, by, w, h, ptr(a[0]), ptr(a[1]), ptr(a[2]),
                                              len(mvc), ptr(cand) if len(mvc) else None, merange, method, subme, qp, ptr(out))
        return c, (int(out[0]), int(out[1]))

    def lookahead_cost_b(self, src0, src1, src2, origin, w, h, mx, my, rows_per_slice, num_slices, prefill_l0):
        lw, lh = ((w // 2 + 7) // 8) * 8, ((h // 2 + 7) // 8) * 8
        wcu, hcu = lw // 8, lh // 8
        ncu = wcu * hcu
        mvs = [np.zeros((ncu, 2), np.int32) for _ in range(2)]
        mvc = [np.zeros(ncu, np.int32) for _ in range(2)]
        lc, rows = np.zeros(ncu, np.uint16), np.zeros(hcu, np.int32)
        est = self.L.ref_lookahead_cost_b(ptr(src0, *origin), ptr(src1, *origin), ptr(src2, *origin), src0.shape[1], w, h, mx, my,
                                          rows_per_slice, num_slices, int(prefill_l0), ptr(mvs[0]), ptr(mvc[0]), ptr(mvs[1]), ptr(mvc[1]),
                                          ptr(lc), ptr(rows))
        return int(est), mvs[0], mvc[0], mvs[1], mvc[1], lc, rows

    def pred_inter_bi(self, ref0, ref1, bx, by, w, h, mv0, mv1):
        y, cb, cr = np.zeros((h, w), self.pix), np.zeros((h // 2, w // 2), self.pix), np.zeros((h // 2, w // 2), self.pix)
        a0, a1 = np.array(mv0, np.int32), np.array(mv1, np.int32)
        self.L.ref_pred_inter_bi(ptr(ref0[0]), ptr(ref0[1]), ptr(ref0[2]), ptr(ref1[0]), ptr(ref1[1]), ptr(ref1[2]), ref0[0].shape[1],
                                 ref0[1].shape[1], bx, by, w, h, ptr(a0), ptr(a1), ptr(y), ptr(cb), ptr(cr))
        return y, cb, cr

    def motion_compensation(self, ref0, ref1, bx, by, w, h, mv0, mv1, wp0, wp1, sliceP=0, uniList=0):
        y, cb, cr = np.zeros((h, w), self.pix), np.zeros((h // 2, w // 2), self.pix), np.zeros((h // 2, w // 2), self.pix)
        a0 = np.array(mv0, np.int32)
        a1 = np.array(mv1 if mv1 is not None else (0, 0), np.int32)
        w0 = np.array(wp0, np.int32).reshape(-1) if wp0 is not None else None
        w1 = np.array(wp1, np.int32).reshape(-1) if wp1 is not None else None
        r1 = [ptr(p) for p in ref1] if ref1 is not None else [None, None, None]
        self.L.ref_motion_compensation(ptr(ref0[0]), ptr(ref0[1]), ptr(ref0[2]), r1[0], r1[1], r1[2], ref0[0].shape[1], ref0[1].shape[1],
                                       bx, by, w, h, ptr(a0), ptr(a1), ptr(w0) if w0 is not None else None, ptr(w1) if w1 is not None else None,
                                       int(sliceP), int(uniList), ptr(y), ptr(cb), ptr(cr))
        return y, cb, cr

    # ---- coefficient-scan cost primitives (dct.cpp:757-1006)
    def scan_order(self, stype, log2):
        out = np.zeros(1 << (2 * log2), np.uint16)
        self.L.ref_scan_order(stype, log2 - 2, ptr(out))
        return out

    def scan4x4(self, stype):
        out = np.zeros(16, np.uint16)
        self.L.ref_scan4x4(stype, ptr(out))
        return out

    def entropy_state_bits(self):
        out = np.zeros(128, np.uint32)
        self.L.ref_entropy_state_bits(ptr(out))
        return out

    def scan_pos_last(self, log2, stype, coeff):
        scan, scan4 = self.scan_order(stype, log2), self.scan4x4(stype if log2 <= 3 else 0)
        sign, flag, num = np.zeros(64, np.uint16), np.zeros(64, np.uint16), np.zeros(64, np.uint8)
        c = np.ascontiguousarray(coeff, np.int16).reshape(-1)
        last = self.L.ref_scanPosLast(ptr(scan), ptr(c), ptr(sign), ptr(flag), ptr(num), int(np.count_nonzero(c)), ptr(scan4), 1 << log2)
        return int(last), sign, flag, num

    def find_pos_first_last(self, tu, cgx, cgy, stype):
        tu = np.ascontiguousarray(tu, np.int16)
        scan = self.scan4x4(stype)
        return int(self.L.ref_findPosFirstLast(ptr(tu, cgy * 4, cgx * 4), tu.shape[1], ptr(scan)))

    def cost_coeff_nxn(self, tu, log2, stype, cgIdx, scanPosSigOff, pattern, offset, ctx):
        tu = np.ascontiguousarray(tu, np.int16)
        scan, scan4 = self.scan_order(stype, log2), self.scan4x4(stype if log2 <= 3 else 0)
        base = int(scan[cgIdx * 16])
        mask = 0
        for k in range(scanPosSigOff + 1):
            mask = mask * 2 + int(tu.reshape(-1)[scan[cgIdx * 16 + k]] != 0)
        absC, ctx2 = np.zeros(16, np.uint16), np.array(ctx, np.uint8)
        tab = sig_ctx_table(log2, pattern)
        first = 1 if scanPosSigOff < 15 else 0
        r = self.L.ref_costCoeffNxN(ptr(scan4), ptr(tu.reshape(-1), 0, base), tu.shape[1], ptr(absC, 0, first), ptr(tab), mask, ptr(ctx2), offset, scanPosSigOff,
                                    cgIdx * 16)
        return int(r), absC, ctx2

    def cost_coeff_remain(self, absCoeff, numNonZero, idx):
        a = np.ascontiguousarray(absCoeff, np.uint16)
        return int(self.L.ref_costCoeffRemain(ptr(a), numNonZero, idx))

    def cost_c1c2_flag(self, absCoeff, numC1Flag, ctx, ctxOffset):
        a, ctx2 = np.ascontiguousarray(absCoeff, np.uint16), np.array(ctx, np.uint8)
        r = self.L.ref_costC1C2Flag(ptr(a), numC1Flag, ptr(ctx2), ctxOffset)
        return int(r), ctx2

    # ---- in-loop filter primitives
    def deblock_ctu_edge(self, planes, ctu, edgeDir, edge, bs, qp, bypass, betaDiv2, tcDiv2, cbOff, crOff, doLuma=1, doChroma=1):
        y, cb, cr = [p.copy() for p in planes]
        b8, q8 = np.ascontiguousarray(bs, np.uint8), np.ascontiguousarray(qp, np.int8)
        t8 = np.ascontiguousarray(bypass, np.uint8) if bypass is not None else None
        self.L.ref_deblock_ctu_edge(ptr(y, *ctu), ptr(cb, ctu[0] // 2, ctu[1] // 2), ptr(cr, ctu[0] // 2, ctu[1] // 2), y.shape[1], cb.shape[1], edgeDir, edge,
                                    ptr(b8), ptr(q8), ptr(t8) if t8 is not None else None, betaDiv2, tcDiv2, cbOff, crOff, doLuma, doChroma)
        return y, cb, cr

    def pel_filter_luma_strong(self, plane, pos, edgeDir, tcP, tcQ):
        p = plane.copy()
        step, off = (p.shape[1], 1) if edgeDir == 0 else (1, p.shape[1])
        self.L.ref_pelFilterLumaStrong(edgeDir, ptr(p, *pos), step, off, tcP, tcQ)
        return p

    def pel_filter_chroma(self, plane, pos, edgeDir, tc, maskP, maskQ):
        p = plane.copy()
        step, off = (p.shape[1], 1) if edgeDir == 0 else (1, p.shape[1])
        self.L.ref_pelFilterChroma(edgeDir, ptr(p, *pos), step, off, tc, maskP, maskQ)
        return p

    def sao_sign(self, a, b):
        d = np.zeros(len(a), np.int8)
        self.L.ref_saoSign(ptr(d), ptr(a), ptr(b), len(a))
        return d

    def sao_e0(self, plane, pos, offsetEo, width, signLeft):
        p, eo, sl = plane.copy(), np.array(offsetEo, np.int8), np.array(signLeft, np.int8)
        self.L.ref_saoCuOrgE0(ptr(p, *pos), ptr(eo), width, ptr(sl), p.shape[1])
        return p

    def sao_e1(self, plane, pos, up, offsetEo, width, rows):
        p, eo, u = plane.copy(), np.array(offsetEo, np.int8), np.array(up, np.int8)
        self.L.ref_saoCuOrgE1(ptr(p, *pos), ptr(u), ptr(eo), p.shape[1], width, rows)
        return p, u

    def sao_e2(self, plane, pos, bufft, buff1, offsetEo, width):
        p, eo, bt, b1 = plane.copy(), np.array(offsetEo, np.int8), np.array(bufft, np.int8), np.array(buff1, np.int8)
        self.L.ref_saoCuOrgE2(ptr(p, *pos), ptr(bt), ptr(b1), ptr(eo), width, p.shape[1])
        return p, bt

    def sao_e3(self, plane, pos, upfull, offsetEo, startX, endX):
        p, eo, u = plane.copy(), np.array(offsetEo, np.int8), np.array(upfull, np.int8)
        self.L.ref_saoCuOrgE3(ptr(p, *pos), ptr(u, 0, 1), ptr(eo), p.shape[1], startX, endX)
        return p, u

    def sao_b0(self, plane, pos, offset32, w, h):
        p, o = plane.copy(), np.array(offset32, np.int8)
        self.L.ref_saoCuOrgB0(ptr(p, *pos), ptr(o), w, h, p.shape[1])
        return p

    def sao_stats(self, kind, diff, plane, pos, endX, endY, stats, count, up1full, uptfull):
        st, ct, u1, ut = np.array(stats, np.int32), np.array(count, np.int32), np.array(up1full, np.int8), np.array(uptfull, np.int8)
        d = np.ascontiguousarray(diff, np.int16)
        S = plane.shape[1]
        if kind == 0:
            self.L.ref_saoCuStatsBO(ptr(d), ptr(plane, *pos), S, endX, endY, ptr(st), ptr(ct))
        elif kind == 1:
            self.L.ref_saoCuStatsE0(ptr(d), ptr(plane, *pos), S, endX, endY, ptr(st), ptr(ct))
        elif kind == 2:
            self.L.ref_saoCuStatsE1(ptr(d), ptr(plane, *pos), S, ptr(u1, 0, 1), endX, endY, ptr(st), ptr(ct))
        elif kind == 3:
            self.L.ref_saoCuStatsE2(ptr(d), ptr(plane, *pos), S, ptr(u1, 0, 1), ptr(ut, 0, 1), endX, endY, ptr(st), ptr(ct))
        else:
            self.L.ref_saoCuStatsE3(ptr(d), ptr(plane, *pos), S, ptr(u1, 0, 1), endX, endY, ptr(st), ptr(ct))
        return st, ct, u1, ut

    # ---- weighted prediction, downscales, transpose
    def weight_pp(self, a, ao, w, h, w0, rnd, shift, offset):
        d = np.zeros_like(a)
        self.L.ref_weight_pp(ptr(a, *ao), ptr(d, *ao), a.shape[1], w, h, w0, rnd, shift, offset)
        return np.ascontiguousarray(d[ao[0]:ao[0] + h, ao[1]:ao[1] + w])

    def weight_sp(self, a, ao, w, h, w0, rnd, shift, offset):
        d = np.zeros((h, w), self.pix)
        self.L.ref_weight_sp(ptr(a, *ao), ptr(d), a.shape[1], w, w, h, w0, rnd, shift, offset)
        return d

    def scale1d_128to64(self, line):
        d = np.zeros(128, self.pix)
        self.L.ref_scale1d_128to64(ptr(d), ptr(line))
        return d

    def scale2d_64to32(self, a, ao):
        d = np.zeros((32, 32), self.pix)
        self.L.ref_scale2d_64to32(ptr(d), ptr(a, *ao), a.shape[1])
        return d

    def transpose(self, size, a, ao):
        d = np.zeros((size, size), self.pix)
        self.L.ref_transpose(cu_of(size), ptr(d), ptr(a, *ao), a.shape[1])
        return d


def same(x, y):
    """Bit-exact comparison of scalars / arrays / tuples thereof."""
    if isinstance(x, tuple):
        return len(x) == len(y) and all(same(a, b) for a, b in zip(x, y))
    if isinstance(x, np.ndarray):
        return x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y)
    return int(x) == int(y)
