"""ctypes binding of libmt355.so (the C ABI declared in include/mt355.h).

The product path has NO fallback: if the HIP library is missing or a call
fails, this module raises.  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmt355.so')
WGRAD_BATCH_MAX = 64      # MT_WGRAD_BATCH_MAX (include/mt355.h)

c_f32p = C.c_void_p
i32, i64, f32 = C.c_int32, C.c_int64, C.c_float


class View(C.Structure):
    _fields_ = [('p', C.c_void_p), ('sn', i64), ('sh', i64), ('sw', i64),
                ('N', i32), ('H', i32), ('W', i32), ('C', i32),
                ('n_in', i32), ('sg', i64)]


class Conv(C.Structure):
    _fields_ = [('x', View), ('y', View), ('wt', C.c_void_p), ('bias', C.c_void_p),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32),
                ('transposed', i32), ('ups', i32), ('act', i32), ('alpha', f32),
                ('res', View), ('beta', f32), ('dsrc', View), ('dact', i32),
                ('dact_c0', i32), ('accumulate', i32), ('precision', i32), ('algo', i32), ('wt_layout', i32),
                ('x_amax', C.c_void_p), ('w_amax', C.c_void_p), ('y_amax', C.c_void_p)]


class WgradTile(C.Structure):       # mt_wgrad_tile: a 64 x 64 channel tile made of four freely chosen 32-channel groups
    _fields_ = [('a', C.c_uint8 * 2), ('b', C.c_uint8 * 2), ('quads', C.c_uint8), ('bias', C.c_uint8),
                ('reserved', C.c_uint8 * 2)]


WGRAD_TILES_MAX = 8


class Wgrad(C.Structure):
    _fields_ = [('d', View), ('g', View), ('KH', i32), ('KW', i32),
                ('stride', i32), ('pad', i32), ('ups', i32), ('splits', i32),
                ('slabs', C.c_void_p), ('precision', i32), ('algo', i32), ('with_bias', i32),
                ('d_amax', C.c_void_p), ('g_amax', C.c_void_p), ('skip_tiles', C.c_uint32), ('d_terms', i32),
                ('n_tiles', i32), ('tiles', WgradTile * WGRAD_TILES_MAX)]


class ConvPlan(C.Structure):       # mt_conv_plan
    _fields_ = [('path', i32), ('wt_layout', i32), ('tiny_in_ok', i32), ('tiny_out_ok', i32), ('products', i32),
                ('wino_num', i32), ('wino_den', i32), ('kernel', C.c_char * 96)]


class WgradPlan(C.Structure):      # mt_wgrad_plan
    _fields_ = [('halo', i32), ('pc', i32), ('thin', i32), ('batchable', i32), ('tile_rows', i32), ('ntiles', i32),
                ('blocks', i32), ('bias_tap_ok', i32), ('d_terms', i32), ('tiny_ok', i32), ('products', i32),
                ('tile_list_ok', i32), ('kernel', C.c_char * 96)]


PATH_IGEMM, PATH_HALO, PATH_PLANES = 0, 1, 2
WT_ROWS, WT_HALO, WT_PLANES = 0, 1, 2


class WarpLayout(C.Structure):
    _fields_ = [('B', i32), ('F', i32), ('C', i32), ('H', i32), ('W', i32),
                ('xb', i64), ('xc', i64), ('xf', i64), ('vb', i64), ('vf', i64),
                ('ob', i64), ('oc', i64), ('of', i64), ('wb', i64), ('wf', i64)]


class AmaxSeg(C.Structure):
    _fields_ = [('p', C.c_void_p), ('n', i64), ('out', C.c_void_p)]


class PackJob(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('w_amax', C.c_void_p),
                ('sa', i64), ('sb', i64), ('st', i64),
                ('A', i32), ('B', i32), ('T', i32), ('AP', i32), ('BP', i32),
                ('precision', i32), ('flip', i32), ('kind', i32), ('BP_total', i32), ('b_off', i32),
                ('frag_order', i32)]


class ReduceSub(C.Structure):
    _fields_ = [('a0', i32), ('A', i32), ('B', i32), ('sa', i64), ('sb', i64), ('st', i64),
                ('dst', C.c_void_p), ('db', C.c_void_p)]


class PackSrc(C.Structure):
    _fields_ = [('p', C.c_void_p), ('sb', i64), ('sr', i64), ('sc', i64),
                ('sy', i64), ('sx', i64), ('C', i32), ('c_off', i32),
                ('scale', C.c_void_p), ('shift', C.c_void_p)]


ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
PREC_F32, PREC_BF16X3, PREC_BF16X6, PREC_F16X3 = 0, 1, 2, 3
_P = C.POINTER
_vp = C.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    'mt_version': [],
    'mt_last_error': [],
    'mt_init': [],
    'mt_amax_scratch_reset': [_vp],
    'mt_conv_run': [_P(Conv), _vp],
    'mt_conv_run_naive': [_P(Conv), _vp],
    'mt_conv_plan_query': [_P(Conv), i32, i32, _P(ConvPlan)],
    'mt_wgrad_plan_query': [_P(Wgrad), i32, i32, _P(WgradPlan)],
    'mt_conv_tiny_in_run': [_P(Conv), _vp, i32, i64, i64, i64, i32, _vp],
    'mt_conv_tiny_out_run': [_P(Conv), _vp, i32, i64, i64, i64, i32, i32, i32, _vp],
    'mt_wgrad_run': [_P(Wgrad), _vp],
    'mt_wgrad_run_batch': [_vp, i32, _vp],
    'mt_wgrad_tiny_run': [_P(Wgrad), i32, _vp],
    'mt_wgrad_slab_floats': [_P(Wgrad)],
    'mt_wgrad_reduce': [_vp, i32, i32, i32, i32, i32, _vp, i32, i32, i64, i64, i64, i32, f32, _vp, _vp],
    'mt_wgrad_reduce_multi': [_vp, i32, i32, i32, i32, i32, _vp, i32, i32, f32, _vp],
    'mt_bias_grad': [_P(View), _vp, i32, _vp, i32, _vp],
    'mt_weight_pack': [_vp, i32, i32, i32, i64, i64, i64, _vp, i32, i32, _vp],
    'mt_weight_pack_split': [_vp, i32, i32, i32, i64, i64, i64, _vp, i32, i32, i32, i32, _vp, _vp],
    'mt_weight_pack_halo': [_vp, i32, i32, i32, i64, i64, i64, _vp, i32, i32, i32, i32, _vp, _vp],
    'mt_weight_pack_halo_part': [_vp, i32, i32, i32, i64, i64, i64, _vp, i32, i32, i32, i32, _vp, i32, i32, _vp],
    'mt_weight_pack_planes_wanted': [i32, i32, i32, i32],
    'mt_weight_pack_planes_elems': [i32, i32, i32, i32],
    'mt_weight_pack_planes': [_vp, i32, i32, i32, i64, i64, i64, _vp, i32, i32, i32, i32, _vp, _vp],
    'mt_absmax': [_P(View), _vp, _vp],
    'mt_absmax_segments': [_vp, i32, _vp],
    'mt_weight_pack_jobs': [_vp, _vp, i32, _vp],
    'mt_weight_pack_halo_frag_order': [i32, i32, i32, i32],
    'mt_absmax_flat': [_vp, i64, _vp, _vp],
    'mt_weight_pack_halo_elems': [i32, i32, i32, i32],
    'mt_pad_vector': [_vp, i32, _vp, i32, _vp],
    'mt_nchw_to_nhwc4': [_vp, i64, i64, i64, i64, i32, i32, i32, i32, _vp, _vp, _P(View), i32, _vp],
    'mt_nhwc4_to_nchw': [_P(View), i32, _vp, i64, i64, i64, i64, i32, _vp, _vp, i32, _vp],
    'mt_pack_pairs': [_P(PackSrc), i32, i32, i32, _P(View), _vp],
    'mt_compose_clamp_fwd': [_P(View), _vp, _vp, _vp, _vp, i32, i32, _vp, _vp, _vp],
    'mt_compose_clamp_bwd': [_P(View), _vp, _vp, _vp, i32, i32, _vp, _vp, _P(View), _vp],
    'mt_maxpool2x2': [_P(View), _P(View), _vp],
    'mt_sumpool': [_P(View), _P(View), i32, _vp],
    'mt_axpby': [_P(View), f32, _P(View), f32, _P(View), _vp, _vp],
    'mt_dact': [_P(View), _P(View), i32, _P(View), _vp],
    'mt_corr_masked': [_P(View), _P(View), _vp, _vp, i32, i32, _vp, _vp, _vp],
    'mt_rownorm': [_P(View), _vp, _vp, _vp],
    'mt_softmax3d': [_vp, i32, i32, i32, _vp, _vp],
    'mt_warp_fwd': [_vp, _vp, _vp, _P(WarpLayout), i32, _vp, _vp, _vp, _vp],
    'mt_warp_bwd_grid': [_vp, _vp, _vp, _P(WarpLayout), i32, _vp, _vp],
    'mt_resize_fwd': [_vp, i64, i64, i64, i32, i32, i32, _vp, i64, i64, i64, i32, i32, i32, i32, _vp],
    'mt_resize_bilinear_bwd': [_vp, i64, i64, i64, i32, i32, i32, _vp, i64, i64, i64, i32, i32, i32, _vp],
    'mt_l1_fwd': [_vp, _vp, _vp, _vp, i32, i32, i32, i64, i32, _vp, _vp],
    'mt_l1_bwd': [_vp, _vp, _vp, _vp, i32, i32, i32, i64, _vp, _vp, _vp, i32, _vp],
    'mt_sobel_l1_fwd': [_vp, _vp, i32, i32, i32, _vp, _vp],
    'mt_sobel_l1_bwd': [_vp, _vp, i32, i32, i32, _vp, _vp, _vp],
    'mt_adam_step': [_vp, _vp, _vp, _vp, i64, f32, f32, f32, f32, f32, f32, _vp],
    'mt_adam_step_dev': [_vp, _vp, _vp, _vp, i64, f32, f32, f32, f32, _vp, _vp],
    'mt_mask_dilate_fill': [_vp, _vp, _vp, i32, i32, i32, i32, i32, i32, i32, _vp, _vp, _vp],
    'mt_psnr_sums': [_vp, _vp, i32, i64, _vp, _vp],
    'mt_ssim_sums': [_vp, _vp, i32, i32, i32, i32, C.c_double, C.c_double, C.c_double, _vp, _vp],
    'mt_comm_unique_id': [C.c_char_p, _vp],
    'mt_comm_init': [C.c_char_p, _vp, i32, i32, _P(_vp)],
    'mt_comm_info': [_vp, _P(i32), _P(i32)],
    'mt_comm_destroy': [_vp],
    'mt_allreduce_sum_f32': [_vp, _vp, i64, _vp],
}
_RESTYPES = {'mt_last_error': C.c_char_p, 'mt_wgrad_slab_floats': i64, 'mt_weight_pack_halo_elems': i64,
             'mt_weight_pack_planes_elems': i64}

_lib = None


def load():
    """Load libmt355.so; raise loudly if it is missing (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP extension is not built. Run '
            '`python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a symbol is missing
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    if torch.cuda.is_available():
        # the library's one device allocation (reduction scratch), made here -- eagerly, outside any graph capture --
        # instead of inside the first launch
        with torch.cuda.device(torch.cuda.current_device()):
            rc = lib.mt_init()
        if rc != 0:
            raise MT355Error(f'mt_init failed ({rc}): {lib.mt_last_error().decode()}')
    return lib


class MT355Error(RuntimeError):
    pass


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MT355Error(f'{name} failed ({rc}): {lib.mt_last_error().decode()}')


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
