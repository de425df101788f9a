"""ctypes binding of libx265hip.so (the C ABI in include/x265hip.h) — plumbing only.

The product is the HIP library; this module just loads it, declares the prototypes and offers small helpers to move
numpy arrays to and from device memory.  There is NO fallback: if the library is missing, or no MI355X-class GPU is
present when a compute entry point is called, an exception is raised.  Nothing here imports or calls oracle/.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libx265hip.so")

vp, i32, i64, u32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64

# name -> (restype, argtypes); mirrors include/x265hip.h one to one (tests/test_abi.py checks the header against this)
PROTOTYPES = {
    "x265hip_init": (i32, [i32]),
    "x265hip_device_count": (i32, []),
    "x265hip_last_error": (C.c_char_p, []),
    "x265hip_version": (C.c_char_p, []),
    "x265hip_malloc": (i32, [C.POINTER(vp), C.c_size_t]),
    "x265hip_free": (i32, [vp]),
    "x265hip_memcpy_h2d": (i32, [vp, vp, C.c_size_t, vp]),
    "x265hip_memcpy_d2h": (i32, [vp, vp, C.c_size_t, vp]),
    "x265hip_memcpy_d2d": (i32, [vp, vp, C.c_size_t, vp]),
    "x265hip_memset": (i32, [vp, i32, C.c_size_t, vp]),
    "x265hip_stream_create": (i32, [C.POINTER(vp)]),
    "x265hip_stream_destroy": (i32, [vp]),
    "x265hip_stream_sync": (i32, [vp]),
    "x265hip_event_create": (i32, [C.POINTER(vp)]),
    "x265hip_event_destroy": (i32, [vp]),
    "x265hip_event_record": (i32, [vp, vp]),
    "x265hip_event_elapsed_ms": (i32, [vp, vp, C.POINTER(C.c_float)]),
    "x265hip_pixcmp_batch": (i32, [i32, i32, i32, i32, vp, i64, vp, i64, vp, vp, i32, vp, vp]),
    "x265hip_sad_xn_batch": (i32, [i32, i32, i32, i32, vp, i64, vp, i64, vp, vp, i32, vp, vp]),
    "x265hip_sse_pp_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, vp, i32, vp, vp]),
    "x265hip_sse_ss_batch": (i32, [i32, i32, vp, i64, vp, i64, vp, vp, i32, vp, vp]),
    "x265hip_sub_ps_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, i32, vp]),
    "x265hip_add_ps_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, i32, vp]),
    "x265hip_addavg_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, i32, vp]),
    "x265hip_pixelavg_pp_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, i32, vp]),
    "x265hip_copy_batch": (i32, [i32, i32, i32, i32, vp, i64, vp, i64, vp, vp, i32, vp]),
    "x265hip_p2s_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, vp, i32, vp]),
    "x265hip_dct_batch": (i32, [i32, i32, i32, vp, i64, vp, vp, i32, vp]),
    "x265hip_idct_batch": (i32, [i32, i32, i32, vp, vp, i64, vp, i32, vp]),
    "x265hip_quant_batch": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "x265hip_nquant_batch": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "x265hip_dequant_normal": (i32, [vp, vp, i64, i32, i32, vp]),
    "x265hip_dequant_scaling_batch": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "x265hip_count_nonzero_batch": (i32, [vp, i32, i32, vp, vp]),
    "x265hip_cpy_shift_batch": (i32, [i32, i32, vp, vp, i64, vp, i32, i32, vp]),
    "x265hip_copy_cnt_batch": (i32, [i32, vp, vp, i64, vp, i32, vp, vp]),
    "x265hip_blockfill_s_batch": (i32, [i32, vp, i64, vp, vp, i32, vp]),
    "x265hip_denoise_dct_batch": (i32, [vp, vp, vp, i32, i32, vp]),
    "x265hip_rdoq_cost_batch": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "x265hip_interp_batch": (i32, [i32, i32, i32, i32, i32, vp, i64, vp, i64, vp, vp, vp, i32, i32, vp]),
    "x265hip_motion_estimate_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, vp, vp, vp, i32, vp, i32, i32, i32,
                                            vp, i32, i32, vp, vp, vp]),
    "x265hip_build_subpel_planes": (i32, [i32, vp, i64, i32, i32, i32, i32, vp, i64, vp]),
    "x265hip_motion_estimate_planes_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32, vp, i32, i32, i32,
                                                   vp, i32, i32, vp, vp, vp]),
    "x265hip_residual_chain_batch": (i32, [i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32, i32, i32, i32,
                                           vp, vp, vp, i32, vp]),
    "x265hip_set_search_range_batch": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
    "x265hip_pred_inter_luma_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, vp, i32, vp]),
    "x265hip_extend_border": (i32, [i32, vp, i64, i32, i32, i32, i32, vp]),
    "x265hip_mvcost_table": (i32, [i32, i32, vp, i32]),
    "x265hip_framepass_create": (i32, [i32, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]),
    "x265hip_framepass_destroy": (i32, [vp]),
    "x265hip_framepass_run_yuv_b": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "x265hip_framepass_run": (i32, [vp, vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, vp]),
    "x265hip_framepass_run_yuv": (i32, [vp, vp, vp, vp, vp, i32, i32, vp]),
    "x265hip_pred_inter_chroma_batch": (i32, [i32, i32, i32, vp, vp, i64, vp, vp, i64, vp, vp, i32, vp]),
    "x265hip_framepass_output": (i32, [vp, i32, i32, C.POINTER(vp), C.POINTER(i32)]),
    "x265hip_framepass_set_profiling": (i32, [vp, i32]),
    "x265hip_framepass_stage_ms": (i32, [vp, C.POINTER(C.c_float)]),
    "x265hip_call_count": (C.c_ulonglong, []),
    "x265hip_call_pixcmp": (i32, [i32, i32, i32, i32, vp, i64, vp, i64, vp]),
    "x265hip_call_sad_xn": (i32, [i32, i32, i32, i32, vp, vp, i64, vp]),
    "x265hip_call_sse_pp": (i32, [i32, i32, i32, vp, i64, vp, i64, vp]),
    "x265hip_call_sse_ss": (i32, [i32, i32, vp, i64, vp, i64, vp]),
    "x265hip_call_dct": (i32, [i32, i32, i32, vp, vp, i64]),
    "x265hip_call_idct": (i32, [i32, i32, i32, vp, vp, i64]),
    "x265hip_call_quant": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "x265hip_call_nquant": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "x265hip_call_dequant_normal": (i32, [vp, vp, i32, i32, i32]),
    "x265hip_call_dequant_scaling": (i32, [vp, vp, vp, i32, i32, i32]),
    "x265hip_call_interp": (i32, [i32, i32, i32, i32, i32, vp, i64, vp, i64, i32, i32, i32]),
    "x265hip_call_sub_ps": (i32, [i32, i32, i32, vp, i64, vp, vp, i64, i64]),
    "x265hip_call_add_ps": (i32, [i32, i32, i32, vp, i64, vp, vp, i64, i64]),
    "x265hip_call_addavg": (i32, [i32, i32, i32, vp, vp, vp, i64, i64, i64]),
    "x265hip_call_pixelavg_pp": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, i64]),
    "x265hip_call_copy": (i32, [i32, i32, i32, i32, vp, i64, vp, i64]),
    "x265hip_call_p2s": (i32, [i32, i32, i32, vp, i64, vp, i64]),
    "x265hip_call_cpy_shift": (i32, [i32, i32, vp, vp, i64, i32]),
    "x265hip_call_copy_cnt": (i32, [i32, vp, vp, i64, vp]),
    "x265hip_call_count_nonzero": (i32, [i32, vp, vp]),
    "x265hip_call_blockfill_s": (i32, [i32, vp, i64, C.c_int16]),
    "x265hip_call_denoise_dct": (i32, [vp, vp, vp, i32]),
    "x265hip_call_rdoq_cost": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, u32]),
    "x265hip_var_batch": (i32, [i32, i32, vp, i64, vp, i32, vp, vp]),
    "x265hip_weight_pp": (i32, [i32, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    "x265hip_weight_sp": (i32, [i32, vp, vp, i64, i64, i32, i32, i32, i32, i32, i32, vp]),
    "x265hip_scale1d_128to64_batch": (i32, [i32, vp, vp, i32, vp]),
    "x265hip_scale2d_64to32_batch": (i32, [i32, vp, i64, vp, vp, i32, vp]),
    "x265hip_transpose_batch": (i32, [i32, i32, vp, i64, vp, vp, i32, vp]),
    "x265hip_call_var": (i32, [i32, i32, vp, i64, vp]),
    "x265hip_call_weight_pp": (i32, [i32, vp, vp, i64, i32, i32, i32, i32, i32, i32]),
    "x265hip_call_weight_sp": (i32, [i32, vp, vp, i64, i64, i32, i32, i32, i32, i32, i32]),
    "x265hip_call_scale1d_128to64": (i32, [i32, vp, vp]),
    "x265hip_call_scale2d_64to32": (i32, [i32, vp, vp, i64]),
    "x265hip_call_transpose": (i32, [i32, i32, vp, vp, i64]),
    "x265hip_intra_pred_batch": (i32, [i32, i32, vp, vp, vp, vp, vp, i64, i32, vp]),
    "x265hip_intra_allangs_batch": (i32, [i32, i32, vp, vp, vp, i32, vp, i32, vp]),
    "x265hip_intra_filter_batch": (i32, [i32, i32, vp, vp, vp, vp, i32, vp]),
    "x265hip_pred_inter_bi_batch": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "x265hip_set_entropy_state_bits": (i32, [vp]),
    "x265hip_scan_pos_last_batch": (i32, [i32, i32, vp, i32, vp, vp, vp, vp, vp]),
    "x265hip_find_pos_first_last_batch": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "x265hip_cost_coeff_nxn_batch": (i32, [vp, vp, i32, vp, i32, vp, vp, vp]),
    "x265hip_cost_coeff_remain_batch": (i32, [vp, vp, vp, i32, vp, vp]),
    "x265hip_cost_c1c2_flag_batch": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    "x265hip_call_scan_pos_last": (i32, [i32, i32, vp, vp, vp, vp, i32, vp]),
    "x265hip_call_find_pos_first_last": (i32, [vp, i64, i32, vp]),
    "x265hip_call_cost_coeff_nxn": (i32, [i32, vp, i64, vp, vp, C.c_uint32, vp, i32, i32, i32, vp]),
    "x265hip_call_cost_coeff_remain": (i32, [vp, i32, i32, vp]),
    "x265hip_call_cost_c1c2_flag": (i32, [vp, i64, vp, i64, vp]),
    "x265hip_pel_filter_luma_strong_batch": (i32, [i32, vp, vp, i64, i64, vp, vp, i32, vp]),
    "x265hip_pel_filter_chroma_batch": (i32, [i32, vp, vp, i64, i64, vp, vp, vp, i32, vp]),
    "x265hip_sao_sign": (i32, [i32, vp, vp, vp, i32, vp]),
    "x265hip_deblock_luma_batch": (i32, [i32, vp, i64, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "x265hip_deblock_chroma_batch": (i32, [i32, vp, vp, i64, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "x265hip_sao_apply_batch": (i32, [i32, i32, vp, i64, vp, vp, i32, vp]),
    "x265hip_sao_stats_batch": (i32, [i32, i32, vp, vp, i64, vp, vp, i32, vp, vp, vp]),
    "x265hip_call_pel_filter_luma_strong": (i32, [i32, vp, i64, i64, i32, i32]),
    "x265hip_call_pel_filter_chroma": (i32, [i32, vp, i64, i64, i32, i32, i32]),
    "x265hip_call_sao_sign": (i32, [i32, vp, vp, vp, i32]),
    "x265hip_call_sao_apply": (i32, [i32, i32, vp, i64, i32, i32, vp, vp, vp, vp]),
    "x265hip_call_sao_stats": (i32, [i32, i32, vp, vp, i64, vp, vp, i32, i32, vp, vp]),
    "x265hip_lookahead_weight_cost_batch": (i32, [i32, vp, vp, i64, i32, i32, vp, vp, i32, vp, vp]),
    "x265hip_lookahead_weights_analyse": (i32, [i32, vp, vp, i64, i64, i64, i32, i32, i32, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp, vp]),
    "x265hip_cutree_propagate": (i32, [i32, i32, i32, i32, C.c_double, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "x265hip_aq_block_energy": (i32, [i32, vp, i32, i32, i32, vp, vp, vp]),
    "x265hip_lookahead_aq_frame": (i32, [i32, vp, i32, i32, i32, i32, C.c_double, i32, vp, vp, vp, vp, vp, vp]),
    "x265hip_motion_compensation_batch": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "x265hip_intra_scan_batch": (i32, [i32, i32, vp, vp, vp, vp, i64, vp, i32, vp, vp]),
    "x265hip_frame_init_lowres": (i32, [i32, vp, i64, vp, vp, vp, vp, i64, i32, i32, vp]),
    "x265hip_lowres_init": (i32, [i32, vp, i64, C.POINTER(vp), i64, i32, i32, i32, i32, vp]),
    "x265hip_lowres_intra_estimate": (i32, [i32, vp, i64, i32, i32, vp, vp, vp, vp, vp]),
    "x265hip_motion_estimate_chroma_batch": (i32, [i32, i32, i32, vp, i64, vp, vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp, i32, vp, i32, i32, i32,
                                                   vp, i32, i32, vp, vp, vp]),
    "x265hip_lookahead_cost_p_batch": (i32, [i32, vp, i32, i64, i64, i32, i32, i32, i32, vp, u32, vp, vp]),
    "x265hip_lookahead_bidir_batch": (i32, [i32, vp, i32, i64, i64, i32, i32, vp, vp]),
    "x265hip_lookahead_pcost_batch": (i32, [vp, i32, i32, i32, vp, vp]),
    "x265hip_build_integral_planes": (i32, [i32, vp, i64, i32, vp, i64, vp, vp]),
    "x265hip_motion_estimate_sea_batch": (i32, [i32, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32, vp, i32, i32, vp, i32, i32, vp, vp, vp]),
    "x265hip_la_create": (vp, [vp]),
    "x265hip_la_create_at": (vp, [i32, vp]),
    "x265hip_la_destroy": (None, [vp]),
    "x265hip_la_set_frame": (i32, [vp, i32, vp, vp, vp]),
    "x265hip_la_put_vectors": (i32, [vp, i32, i32, i32, vp, vp]),
    "x265hip_la_has_vectors": (i32, [vp, i32, i32, i32]),
    "x265hip_la_weights_analyse": (i32, [vp, i32, i32, u64, u64, u64, u64, vp, vp, vp]),
    "x265hip_la_estimate_batch": (i32, [vp, vp, i32, i32, i32]),
    "x265hip_la_estimate_batch_ahead": (i32, [vp, vp, i32, i32, i32, vp, i32]),
    "x265hip_la_has_ahead": (i32, [vp, i32, i32, i32, i32, i32, i32]),
    "x265hip_la_stats": (i32, [vp, vp, vp, vp]),
    "x265hip_la_stats_ahead": (i32, [vp, vp, vp, vp, vp]),
    "x265hip_source_energy": (i32, [i32, vp, i64, i32, i32, vp, vp]),
    "x265hip_refpic_create": (vp, [i32, i32, i32, i64, i32, i32, i32, vp]),
    "x265hip_refpic_destroy": (None, [vp]),
    "x265hip_refpic_reset": (i32, [vp]),
    "x265hip_refpic_rows_final": (i32, [vp, i32]),
    "x265hip_refpic_plane": (vp, [vp, i32]),
    "x265hip_refpic_rows_ready": (i32, [vp]),
    "x265hip_refpic_rows_ready_ptr": (vp, [vp]),
    "x265hip_refpic_wait": (i32, [vp]),
    "x265hip_srcpic_create": (vp, [i32, i32, i32]),
    "x265hip_srcpic_upload": (i32, [vp, vp, i64]),
    "x265hip_srcpic_destroy": (None, [vp]),
    "x265hip_sadsurf_attach": (vp, [vp, vp, i32, i32]),
    "x265hip_sadsurf_get_view": (vp, [vp]),
    "x265hip_sadsurf_release": (None, [vp]),
    "x265hip_sadsurf_stats": (i32, [vp, vp, vp, vp]),
    "x265hip_cuserve_open": (i32, [i32, i32, C.POINTER(vp)]),
    "x265hip_cuserve_open_at": (i32, [i32, i32, i32, C.POINTER(vp)]),
    "x265hip_cuserve_close": (i32, [vp]),
    "x265hip_cuserve_slot": (i32, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "x265hip_cuserve_submit": (i32, [vp, i32, C.POINTER(u32)]),
    "x265hip_cuserve_poke": (i32, [vp, i32]),
    "x265hip_cuserve_submit_sao": (i32, [vp, i32, vp, vp]),
    "x265hip_cuserve_submit_intra": (i32, [vp, i32, vp, vp]),
    "x265hip_cuserve_stats": (i32, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
    "x265hip_device_time": (i32, [i32, vp, vp, vp]),
    "x265hip_sadsurf_attach_levels": (vp, [vp, vp, i32, i32, i32]),
    "x265hip_places": (i32, [i32, vp]),
    "x265hip_peer_stats": (i32, [vp, vp, vp]),
    "x265hip_refpic_create_at": (vp, [i32, i32, i32, i32, i64, i32, i32, i32, vp]),
    "x265hip_srcpic_create_at": (vp, [i32, i32, i32, i32]),
    "x265hip_call_intra_pred": (i32, [i32, i32, i32, i32, vp, i64, vp]),
    "x265hip_call_intra_allangs": (i32, [i32, i32, vp, vp, vp, i32]),
    "x265hip_call_intra_filter": (i32, [i32, i32, vp, vp]),
    "x265hip_call_frame_init_lowres": (i32, [i32, vp, i64, vp, vp, vp, vp, i64, i32, i32]),
}

CMP_SAD, CMP_SATD, CMP_SA8D, CMP_SA8D8, CMP_PSY = 0, 1, 2, 3, 4


class SaoJob(C.Structure):
    """x265hip_sao_job (include/x265hip.h)"""
    _fields_ = [("recOff", C.c_int64), ("aux0", C.c_int64), ("aux1", C.c_int64), ("width", C.c_int32), ("height", C.c_int32), ("startX", C.c_int32),
                ("offsets", C.c_int8 * 32), ("signLeft", C.c_int8 * 2), ("reserved", C.c_int8 * 2)]


class SaoStatsJob(C.Structure):
    """x265hip_sao_stats_job (include/x265hip.h)"""
    _fields_ = [("diffOff", C.c_int64), ("recOff", C.c_int64), ("aux0", C.c_int64), ("aux1", C.c_int64), ("endX", C.c_int32), ("endY", C.c_int32)]


class CoeffGroupJob(C.Structure):
    """x265hip_coeff_group_job (include/x265hip.h)"""
    _fields_ = [("coeffOffset", C.c_int64), ("trSize", C.c_int32), ("scanType", C.c_int32), ("scanFlagMask", C.c_uint32), ("offset", C.c_int32),
                ("scanPosSigOff", C.c_int32), ("subPosBase", C.c_int32), ("tabSigCtx", C.c_uint8 * 16)]


class WeightParam(C.Structure):
    """x265hip_weight_param (include/x265hip.h): one plane's WeightParam of the slice header"""
    _fields_ = [("inputWeight", C.c_int32), ("inputOffset", C.c_int32), ("log2WeightDenom", C.c_int32), ("wtPresent", C.c_int32)]


class LookaheadPair(C.Structure):
    """x265hip_lookahead_pair (include/x265hip.h)"""
    _fields_ = [("fenc", vp), ("ref", vp), ("intraCost", vp), ("mvs", vp), ("mvCosts", vp), ("lowresCosts", vp), ("rowSatds", vp), ("sync", vp),
                ("invQscale", vp), ("bidirList", C.c_int32), ("sliceGeom", C.c_int32)]


class LaConfig(C.Structure):
    """x265hip_la_config (include/x265hip.h)"""
    _fields_ = [("depth", C.c_int32), ("width", C.c_int32), ("lines", C.c_int32), ("stride", C.c_int64), ("planeElems", C.c_int64), ("padOffset", C.c_int64),
                ("widthInCU", C.c_int32), ("heightInCU", C.c_int32), ("maxDist", C.c_int32), ("numSlots", C.c_int32)]


class LaEstimate(C.Structure):
    """x265hip_la_estimate (include/x265hip.h)"""
    _fields_ = [("b", C.c_int32), ("p0", C.c_int32), ("p1", C.c_int32), ("dist0", C.c_int32), ("dist1", C.c_int32), ("search0", C.c_int32),
                ("search1", C.c_int32), ("weightedId", C.c_int32), ("mvs0", vp), ("mvCosts0", vp), ("mvs1", vp), ("mvCosts1", vp), ("lowresCosts", vp),
                ("rowSatds", vp), ("costEst", C.c_int64), ("costEstAq", C.c_int64), ("intraMbs", C.c_int32), ("reserved", C.c_int32)]


class SadSurfLevel(C.Structure):
    """x265hip_sadsurf_level (include/x265hip.h)"""
    _fields_ = [("blocksX", C.c_int32), ("blocksY", C.c_int32), ("entryBytes", C.c_int32), ("blocksPerCtuRow", C.c_int32), ("origin", vp), ("table", vp), ("subpel", vp)]


class SadSurfView(C.Structure):
    """x265hip_sadsurf_view (include/x265hip.h)"""
    _fields_ = [("level", SadSurfLevel * 4), ("ctuRowPitch", C.c_int64), ("ctuRowsReady", C.POINTER(C.c_int))]


class CuJob(C.Structure):
    """x265hip_cujob (include/x265hip.h): the header of a CU residual quad-tree job"""
    _fields_ = [("log2CUSize", u32), ("log2TrMax", u32), ("log2TrMin", u32), ("chroma", u32), ("bitDepth", u32), ("quantOffset", u32), ("signHide", u32),
                ("reserved", u32), ("qpRem", C.c_int32 * 3), ("qpPer", C.c_int32 * 3), ("quantScale", C.c_int32 * 3), ("dequantScale", C.c_int32 * 3),
                ("coefMode", u32), ("sourceDct", u32)]


class SaoJobPlane(C.Structure):
    _fields_ = [("w", C.c_uint16), ("h", C.c_uint16), ("x0", C.c_uint8 * 5), ("y0", C.c_uint8 * 5), ("x1", C.c_uint8 * 5), ("y1", C.c_uint8 * 5)]


class SaoCtuJob(C.Structure):
    """x265hip_saojob (include/x265hip.h): the SAO statistics of one CTU as a job of the CU-job service"""
    _fields_ = [("bitDepth", u32), ("planes", u32), ("eo23", u32), ("reserved", u32), ("plane", SaoJobPlane * 3)]


SAOJOB_STATS_ENTRIES = 3 * 5 * 32


class IntraScanJob(C.Structure):
    """x265hip_intrajob (include/x265hip.h): the 35-mode sa8d scan of one block as a job of the CU-job service"""
    _fields_ = [("bitDepth", u32), ("mark", u32), ("log2Size", u32), ("reserved", u32)]


INTRAJOB_MARK = 0x100


class CuJobUnit(C.Structure):
    """x265hip_cujob_unit (include/x265hip.h): one transform unit's result header"""
    _fields_ = [("ready", u32), ("numSig", u32), ("zeroDist", u64), ("codedDist", u64), ("readyInv", u32), ("fwdTicks", u32), ("codedEnergy", u32), ("reserved", u32 * 3)]


CUJOB_MAX_UNITS = 60
CUJOB_MAX_ELEMS = 2 * 6144
CUJOB_PIXEL_BYTES = 2 * 6144 * 2


class LaSearch(C.Structure):
    """x265hip_la_search (include/x265hip.h): a list search handed over ahead of the reference's request"""
    _fields_ = [("b", C.c_int32), ("ref", C.c_int32), ("list", C.c_int32), ("dist", C.c_int32), ("bidir", C.c_int32), ("weightedId", C.c_int32),
                ("numRowsPerSlice", C.c_int32), ("numSlices", C.c_int32)]


class LookaheadBFrame(C.Structure):
    """x265hip_lookahead_bframe (include/x265hip.h)"""
    _fields_ = [("fenc", vp), ("ref0", vp), ("ref1", vp), ("mvs0", vp), ("mvs1", vp), ("mvCosts0", vp), ("mvCosts1", vp), ("lowresCosts", vp),
                ("rowSatds", vp), ("invQscale", vp)]
IF_HPP, IF_HPS, IF_VPP, IF_VPS, IF_VSP, IF_VSS, IF_HVPP = range(7)
DIA_SEARCH, HEX_SEARCH, UMH_SEARCH, STAR_SEARCH, FULL_SEARCH = 0, 1, 2, 3, 5      # x265.h X265_*_SEARCH


class HipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libx265hip.so (built by __graft_entry__.build() / x265_amd/csrc/Makefile). Fails loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipError("x265_amd/libx265hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(code):
    if code != 0:
        raise HipError("libx265hip error %d: %s" % (code, lib().x265hip_last_error().decode(errors="replace")))


def pix_dtype(depth):
    return np.uint8 if depth == 8 else np.uint16


class DevBuf:
    """A device allocation owned by libx265hip (hipMalloc); optionally initialised from a numpy array."""

    def __init__(self, arr=None, nbytes=None, dtype=None, shape=None):
        L = lib()
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            nbytes, dtype, shape = arr.nbytes, arr.dtype, arr.shape
        self.nbytes, self.dtype, self.shape = int(nbytes), np.dtype(dtype) if dtype is not None else None, shape
        p = vp()
        check(L.x265hip_malloc(C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value
        if arr is not None and self.nbytes:
            check(L.x265hip_memcpy_h2d(self.ptr, arr.ctypes.data, self.nbytes, None))
            check(L.x265hip_stream_sync(None))

    @classmethod
    def empty(cls, shape, dtype):
        shape = tuple(np.atleast_1d(shape).tolist()) if not isinstance(shape, tuple) else shape
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return cls(nbytes=n, dtype=dtype, shape=shape)

    @classmethod
    def zeros(cls, shape, dtype):
        b = cls.empty(shape, dtype)
        check(lib().x265hip_memset(b.ptr, 0, b.nbytes, None))
        return b

    def at(self, elem_offset):
        """Device address of element `elem_offset` (flat index)."""
        return self.ptr + int(elem_offset) * self.dtype.itemsize

    def get(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            check(lib().x265hip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, None))
        return out

    def free(self):
        if self.ptr:
            lib().x265hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def dev_i32(values):
    return DevBuf(np.asarray(values, np.int32).reshape(-1))


class Timer:
    """HIP-event timer on a given stream handle (int / None)."""

    def __init__(self, stream=None):
        L = lib()
        self.stream = stream
        a, b = vp(), vp()
        check(L.x265hip_event_create(C.byref(a)))
        check(L.x265hip_event_create(C.byref(b)))
        self.a, self.b = a.value, b.value

    def start(self):
        check(lib().x265hip_event_record(self.a, self.stream))

    def stop_ms(self):
        L = lib()
        check(L.x265hip_event_record(self.b, self.stream))
        ms = C.c_float()
        check(L.x265hip_event_elapsed_ms(self.a, self.b, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib().x265hip_event_destroy(self.a)
            lib().x265hip_event_destroy(self.b)
        except Exception:
            pass
