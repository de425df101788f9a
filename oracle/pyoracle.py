"""pyoracle — TEST INFRASTRUCTURE ONLY.

ctypes loaders for (1) oracle/libx265oracle.so, our plain-C restatement of the x265 C primitives, and
(2) oracle/_ref/libx265ref{8,10,12}.so, the REAL reference compiled by oracle/Makefile.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under x265_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libx265oracle.so")
REF_DIR = os.path.join(HERE, "_ref")

# LumaPU enum order, reference common/primitives.h:41-55
PU_SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 8), (16, 8), (8, 16), (32, 16), (16, 32),
            (64, 32), (32, 64), (16, 12), (12, 16), (16, 4), (4, 16), (32, 24), (24, 32), (32, 8), (8, 32),
            (64, 48), (48, 64), (64, 16), (16, 64)]

vp, ip, i32, u32, i64, u64 = C.c_void_p, C.c_ssize_t, C.c_int, C.c_uint32, C.c_int64, C.c_uint64


def build_oracle():
    """Compile the C restatement (gcc, < 2 s). Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def build_ref():
    """Compile the real reference from /root/reference (only possible where that tree exists)."""
    subprocess.check_call(["make", "-s", "-j8", "-C", HERE, "ref"])


def ptr(a, y=0, x=0):
    """Address of element (y, x) of a C-contiguous 1-D/2-D numpy array."""
    if a.ndim == 1:
        return vp(a.ctypes.data + x * a.itemsize)
    assert a.flags["C_CONTIGUOUS"]
    return vp(a.ctypes.data + (y * a.shape[1] + x) * a.itemsize)


def pix_dtype(depth):
    return np.uint8 if depth == 8 else np.uint16


def sfx(depth):
    return "8" if depth == 8 else "16"


_oracle = None


def oracle():
    global _oracle
    if _oracle is not None:
        return _oracle
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    L = C.CDLL(ORACLE_SO)
    for s in ("8", "16"):
        def f(name, res, args):
            fn = getattr(L, "%s_%s" % (name, s))
            fn.restype, fn.argtypes = res, args
        f("orc_sad", i32, [vp, ip, vp, ip, i32, i32])
        f("orc_sad_x3", None, [vp, vp, vp, vp, ip, i32, i32, vp])
        f("orc_sad_x4", None, [vp, vp, vp, vp, vp, ip, i32, i32, vp])
        f("orc_satd", i32, [vp, ip, vp, ip, i32, i32])
        f("orc_sa8d", i32, [vp, ip, vp, ip, i32])
        f("orc_sa8d8", i32, [vp, ip, vp, ip, i32, i32])
        f("orc_sse_pp", u64, [vp, ip, vp, ip, i32, i32])
        f("orc_psy_cost_pp", i32, [vp, ip, vp, ip, i32])
        f("orc_var", u64, [vp, ip, i32])
        f("orc_weight_pp", None, [vp, vp, ip, i32, i32, i32, i32, i32, i32, i32])
        f("orc_weight_sp", None, [vp, vp, ip, ip, i32, i32, i32, i32, i32, i32, i32])
        f("orc_scale1d_128to64", None, [vp, vp])
        f("orc_scale2d_64to32", None, [vp, vp, ip])
        f("orc_transpose", None, [vp, vp, ip, i32])
        f("orc_sub_ps", None, [vp, ip, vp, vp, ip, ip, i32, i32])
        f("orc_add_ps", None, [vp, ip, vp, vp, ip, ip, i32, i32, i32])
        f("orc_calcresidual", None, [vp, vp, vp, ip, i32])
        f("orc_addAvg", None, [vp, vp, vp, ip, ip, ip, i32, i32, i32])
        f("orc_pixelavg_pp", None, [vp, ip, vp, ip, vp, ip, i32, i32])
        f("orc_copy_pp", None, [vp, ip, vp, ip, i32, i32])
        f("orc_copy_sp", None, [vp, ip, vp, ip, i32, i32])
        f("orc_copy_ps", None, [vp, ip, vp, ip, i32, i32])
        f("orc_interp_horiz_pp", None, [i32, vp, ip, vp, ip, i32, i32, i32, i32])
        f("orc_interp_horiz_ps", None, [i32, vp, ip, vp, ip, i32, i32, i32, i32, i32])
        f("orc_interp_vert_pp", None, [i32, vp, ip, vp, ip, i32, i32, i32, i32])
        f("orc_interp_vert_ps", None, [i32, vp, ip, vp, ip, i32, i32, i32, i32])
        f("orc_interp_vert_sp", None, [i32, vp, ip, vp, ip, i32, i32, i32, i32])
        f("orc_interp_hv_pp", None, [i32, vp, ip, vp, ip, i32, i32, i32, i32, i32])
        f("orc_p2s", None, [vp, ip, vp, ip, i32, i32, i32])
        f("orc_motion_estimate", i32, [vp, ip, i32, i32, vp, i32, i32, vp, vp, vp, i32, vp, i32, i32, i32, vp, i32, vp])
        f("orc_integral_plane", None, [vp, ip, i32, i32, i32, vp])
        f("orc_motion_estimate_sea", i32, [vp, ip, vp, i32, i32, vp, i32, i32, vp, vp, vp, i32, vp, i32, i32, vp, i32, vp])
        f("orc_subpel_compare", i32, [vp, ip, i32, i32, vp, i32, i32, i32, i32, i32, i32])
        f("orc_motion_estimate_chroma", i32, [vp, ip, vp, vp, ip, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, i32, i32, i32, vp, i32, vp])
        f("orc_intra_filter", None, [i32, vp, vp])
        f("orc_intra_pred", None, [i32, i32, vp, ip, vp, i32, i32])
        f("orc_intra_uses_filtered", i32, [i32, i32])
        f("orc_intra_allangs", None, [i32, vp, vp, vp, i32, i32])
        f("orc_frame_init_lowres", None, [vp, vp, vp, vp, vp, ip, ip, i32, i32])
        f("orc_lowres_intra_estimate", i32, [vp, ip, i32, i32, i32, vp, vp, vp])
        f("orc_lookahead_cost_b", i64, [vp, vp, vp, ip, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp])
        f("orc_lookahead_cost_p", i64, [vp, vp, ip, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp])

    def g(name, res, args):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    g("orc_sse_ss", u64, [vp, ip, vp, ip, i32, i32])
    g("orc_ssd_s", u64, [vp, ip, i32])
    g("orc_blockfill_s", None, [vp, ip, C.c_int16, i32])
    g("orc_cpy2Dto1D_shl", None, [vp, vp, ip, i32, i32])
    g("orc_cpy2Dto1D_shr", None, [vp, vp, ip, i32, i32])
    g("orc_cpy1Dto2D_shl", None, [vp, vp, ip, i32, i32])
    g("orc_cpy1Dto2D_shr", None, [vp, vp, ip, i32, i32])
    g("orc_copy_ss", None, [vp, ip, vp, ip, i32, i32])
    g("orc_count_nonzero", i32, [vp, i32])
    g("orc_copy_cnt", u32, [vp, vp, ip, i32])
    g("orc_interp_vert_ss", None, [i32, vp, ip, vp, ip, i32, i32, i32])
    g("orc_dct", None, [i32, vp, vp, ip, i32])
    g("orc_idct", None, [i32, vp, vp, ip, i32])
    g("orc_dst4", None, [vp, vp, ip, i32])
    g("orc_idst4", None, [vp, vp, ip, i32])
    g("orc_quant", u32, [vp, vp, vp, vp, i32, i32, i32])
    g("orc_nquant", u32, [vp, vp, vp, i32, i32, i32])
    g("orc_dequant_normal", None, [vp, vp, i32, i32, i32])
    g("orc_dequant_scaling", None, [vp, vp, vp, i32, i32, i32])
    g("orc_denoise_dct", None, [vp, vp, vp, i32])
    g("orc_nonpsy_rdoquant", None, [i32, vp, vp, vp, vp, u32, i32])
    g("orc_psy_rdoquant", None, [i32, vp, vp, vp, vp, vp, vp, u32, i32])
    g("orc_psy_rdoquant_1p", None, [i32, vp, vp, vp, vp, u32, i32])
    g("orc_psy_rdoquant_2p", None, [i32, vp, vp, vp, vp, vp, vp, u32, i32])
    g("orc_luma_filter", C.POINTER(C.c_int16), [i32])
    g("orc_chroma_filter", C.POINTER(C.c_int16), [i32])
    g("orc_dct_matrix", C.POINTER(C.c_int16), [i32])
    g("orc_mvcost_table", None, [i32, i32, vp])
    g("orc_partition_from_sizes", i32, [i32, i32])
    _oracle = L
    return L


_refs = {}


def ref_available(depth):
    return os.path.exists(os.path.join(REF_DIR, "libx265ref%d.so" % (depth if depth in (8, 10, 12) else 10)))


def ref(depth):
    """The real reference build for `depth` (8, 10 or 12). Raises FileNotFoundError if oracle/_ref was not built."""
    key = depth if depth in (8, 10, 12) else 10
    if key in _refs:
        return _refs[key]
    path = os.path.join(REF_DIR, "libx265ref%d.so" % key)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    L = C.CDLL(path)

    def g(name, res, args):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    g("ref_depth", i32, [])
    g("ref_sad", i32, [i32, vp, ip, vp, ip])
    g("ref_sad_x3", None, [i32, vp, vp, vp, vp, ip, vp])
    g("ref_sad_x4", None, [i32, vp, vp, vp, vp, vp, ip, vp])
    g("ref_satd", i32, [i32, vp, ip, vp, ip])
    g("ref_sa8d", i32, [i32, vp, ip, vp, ip])
    g("ref_chroma_satd", i32, [i32, vp, ip, vp, ip])
    g("ref_chroma_sa8d", i32, [i32, vp, ip, vp, ip])
    g("ref_sse_pp", u64, [i32, vp, ip, vp, ip])
    g("ref_sse_ss", u64, [i32, vp, ip, vp, ip])
    g("ref_ssd_s", u64, [i32, vp, ip])
    g("ref_psy_cost_pp", i32, [i32, vp, ip, vp, ip])
    g("ref_var", u64, [i32, vp, ip])
    g("ref_weight_pp", None, [vp, vp, ip, i32, i32, i32, i32, i32, i32])
    g("ref_weight_sp", None, [vp, vp, ip, ip, i32, i32, i32, i32, i32, i32])
    g("ref_scale1d_128to64", None, [vp, vp])
    g("ref_scale2d_64to32", None, [vp, vp, ip])
    g("ref_transpose", None, [i32, vp, vp, ip])
    g("ref_sub_ps", None, [i32, vp, ip, vp, vp, ip, ip])
    g("ref_add_ps", None, [i32, vp, ip, vp, vp, ip, ip])
    g("ref_calcresidual", None, [i32, vp, vp, vp, ip])
    g("ref_addAvg", None, [i32, vp, vp, vp, ip, ip, ip])
    g("ref_pixelavg_pp", None, [i32, vp, ip, vp, ip, vp, ip])
    g("ref_copy_pp", None, [i32, vp, ip, vp, ip])
    g("ref_copy_sp", None, [i32, vp, ip, vp, ip])
    g("ref_copy_ps", None, [i32, vp, ip, vp, ip])
    g("ref_copy_ss", None, [i32, vp, ip, vp, ip])
    g("ref_blockfill_s", None, [i32, vp, ip, C.c_int16])
    g("ref_cpy2Dto1D_shl", None, [i32, vp, vp, ip, i32])
    g("ref_cpy2Dto1D_shr", None, [i32, vp, vp, ip, i32])
    g("ref_cpy1Dto2D_shl", None, [i32, vp, vp, ip, i32])
    g("ref_cpy1Dto2D_shr", None, [i32, vp, vp, ip, i32])
    g("ref_copy_cnt", u32, [i32, vp, vp, ip])
    g("ref_count_nonzero", i32, [i32, vp])
    g("ref_dct", None, [i32, vp, vp, ip])
    g("ref_idct", None, [i32, vp, vp, ip])
    g("ref_dst4", None, [vp, vp, ip])
    g("ref_idst4", None, [vp, vp, ip])
    g("ref_quant", u32, [vp, vp, vp, vp, i32, i32, i32])
    g("ref_nquant", u32, [vp, vp, vp, i32, i32, i32])
    g("ref_dequant_normal", None, [vp, vp, i32, i32, i32])
    g("ref_dequant_scaling", None, [vp, vp, vp, i32, i32, i32])
    g("ref_denoise_dct", None, [vp, vp, vp, i32])
    g("ref_nonpsy_rdoquant", None, [i32, vp, vp, vp, vp, u32])
    g("ref_psy_rdoquant", None, [i32, vp, vp, vp, vp, vp, vp, u32])
    g("ref_psy_rdoquant_1p", None, [i32, vp, vp, vp, vp, u32])
    g("ref_psy_rdoquant_2p", None, [i32, vp, vp, vp, vp, vp, vp, u32])
    g("ref_interp_hpp", None, [i32, i32, vp, ip, vp, ip, i32])
    g("ref_interp_hps", None, [i32, i32, vp, ip, vp, ip, i32, i32])
    g("ref_interp_vpp", None, [i32, i32, vp, ip, vp, ip, i32])
    g("ref_interp_vps", None, [i32, i32, vp, ip, vp, ip, i32])
    g("ref_interp_vsp", None, [i32, i32, vp, ip, vp, ip, i32])
    g("ref_interp_vss", None, [i32, i32, vp, ip, vp, ip, i32])
    g("ref_interp_hvpp", None, [i32, vp, ip, vp, ip, i32, i32])
    g("ref_p2s", None, [i32, i32, vp, ip, vp, ip])
    g("ref_dct_matrix", C.POINTER(C.c_int16), [i32])
    g("ref_luma_filter", C.POINTER(C.c_int16), [i32])
    g("ref_chroma_filter", C.POINTER(C.c_int16), [i32])
    g("ref_partition_from_sizes", i32, [i32, i32])
    g("ref_mvcost_table", None, [i32, vp])
    g("ref_motion_estimate", i32, [vp, vp, ip, i32, i32, i32, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp])
    g("ref_integral_planes", None, [vp, ip, i32, i32, i32, vp, i64])
    g("ref_motion_estimate_sea", i32, [vp, vp, ip, vp, i64, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, i32, i32, i32, vp])
    g("ref_motion_estimate_chroma", i32, [vp, vp, vp, vp, vp, vp, ip, ip, i32, i32, i32, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp])
    g("ref_pred_inter_bi", None, [vp, vp, vp, vp, vp, vp, ip, ip, i32, i32, i32, i32, vp, vp, vp, vp, vp])
    u16p = u8p = vp
    g("ref_scan_order", None, [i32, i32, vp])
    g("ref_scan4x4", None, [i32, vp])
    g("ref_entropy_state_bits", None, [vp])
    g("ref_scanPosLast", i32, [vp, vp, vp, vp, vp, i32, vp, i32])
    g("ref_findPosFirstLast", C.c_uint32, [vp, ip, vp])
    g("ref_costCoeffNxN", C.c_uint32, [vp, vp, ip, vp, vp, C.c_uint32, vp, i32, i32, i32])
    g("ref_costCoeffRemain", C.c_uint32, [vp, i32, i32])
    g("ref_costC1C2Flag", C.c_uint32, [vp, ip, vp, ip])
    g("ref_lookahead_cost_p_weightp", C.c_int64, [vp, vp, ip, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp])
    g("ref_cutree_propagate", i32, [vp, ip, i32, i32, i32, i32, i32, i32, i32, C.c_double, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, C.c_double, vp, vp, i32])
    g("ref_aq_frame", i32, [vp, vp, vp, ip, ip, i32, i32, i32, i32, i32, i32, C.c_double, i32, vp, vp, vp, vp])
    g("ref_weights_analyse", i32, [vp, vp, ip, i32, i32, i32, i32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_int64, vp])
    g("ref_deblock_ctu_edge", None, [vp, vp, vp, ip, ip, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32])
    g("ref_pelFilterLumaStrong", None, [i32, vp, ip, ip, i32, i32])
    g("ref_pelFilterChroma", None, [i32, vp, ip, ip, i32, i32, i32])
    g("ref_saoSign", None, [vp, vp, vp, i32])
    g("ref_saoCuOrgE0", None, [vp, vp, i32, vp, ip])
    g("ref_saoCuOrgE1", None, [vp, vp, vp, ip, i32, i32])
    g("ref_saoCuOrgE2", None, [vp, vp, vp, vp, i32, ip])
    g("ref_saoCuOrgE3", None, [vp, vp, vp, ip, i32, i32])
    g("ref_saoCuOrgB0", None, [vp, vp, i32, i32, ip])
    g("ref_saoCuStatsBO", None, [vp, vp, ip, i32, i32, vp, vp])
    g("ref_saoCuStatsE0", None, [vp, vp, ip, i32, i32, vp, vp])
    g("ref_saoCuStatsE1", None, [vp, vp, ip, vp, i32, i32, vp, vp])
    g("ref_saoCuStatsE2", None, [vp, vp, ip, vp, vp, i32, i32, vp, vp])
    g("ref_saoCuStatsE3", None, [vp, vp, ip, vp, i32, i32, vp, vp])
    g("ref_motion_compensation", None, [vp, vp, vp, vp, vp, vp, ip, ip, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp])
    g("ref_intra_filter", None, [i32, vp, vp])
    g("ref_intra_pred", None, [i32, i32, vp, ip, vp, i32])
    g("ref_intra_allangs", None, [i32, vp, vp, vp, i32])
    g("ref_intra_filter_flags", i32, [i32])
    g("ref_frame_init_lowres", None, [vp, vp, vp, vp, vp, ip, ip, i32, i32])
    g("ref_lowres_intra_estimate", i32, [vp, ip, i32, i32, i32, i32, vp, vp, i64, vp, vp, vp])
    g("ref_lookahead_cost_b", i64, [vp, vp, vp, ip, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp])
    g("ref_lookahead_cost_p", i64, [vp, vp, ip, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp])
    assert L.ref_depth() == key
    _refs[key] = L
    return L


MVCOST_CENTRE = 2 * 32768


def mvcost_table(qp, depth):
    """u16[4*32768+1] lambda-scaled MVD cost row (restated BitCost::setQP); index MVCOST_CENTRE is MVD 0."""
    t = np.zeros(4 * 32768 + 1, dtype=np.uint16)
    oracle().orc_mvcost_table(qp, depth, ptr(t))
    return t
