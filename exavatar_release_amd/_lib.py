"""ctypes binding of ``libexa_raster.so`` (C ABI declared in ``include/exa_raster.h``).

The library is the product: if it is missing or fails to load this module raises -- there is no
CPU / PyTorch fallback path anywhere in the package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EXA_RASTER_LIB: developer override to load an experimental build of the same ABI (tools/build_variant.sh, tools/gpu_ab.sh)
LIB_PATH = os.environ.get('EXA_RASTER_LIB') or os.path.join(_HERE, 'libexa_raster.so')

c_float_p = ctypes.c_void_p      # device pointers travel as plain addresses
c_void_p = ctypes.c_void_p


class ExaRasterSettings(ctypes.Structure):
    _fields_ = [
        ('image_height', ctypes.c_int32),
        ('image_width', ctypes.c_int32),
        ('tanfovx', ctypes.c_float),
        ('tanfovy', ctypes.c_float),
        ('bg', c_void_p),
        ('scale_modifier', ctypes.c_float),
        ('viewmatrix', c_void_p),
        ('projmatrix', c_void_p),
        ('sh_degree', ctypes.c_int32),
        ('campos', c_void_p),
        ('prefiltered', ctypes.c_int32),
        ('debug', ctypes.c_int32),
    ]


class ExaRasterWorkspaceSizes(ctypes.Structure):
    _fields_ = [
        ('geom_bytes', ctypes.c_uint64),
        ('tile_bytes', ctypes.c_uint64),
        ('bin_bytes', ctypes.c_uint64),
        ('grad_bytes', ctypes.c_uint64),
    ]


class ExaRasterHeader(ctypes.Structure):
    _fields_ = [('num_rendered', ctypes.c_uint32), ('overflow', ctypes.c_uint32), ('max_tile_list', ctypes.c_uint32),
                ('num_visible', ctypes.c_uint32), ('num_instances', ctypes.c_uint32), ('active_cells', ctypes.c_uint32),
                ('num_tile_instances', ctypes.c_uint32)]


class ExaRasterForwardJob(ctypes.Structure):
    """One render of a batched forward call (include/exa_raster.h)."""
    _fields_ = [
        ('settings', ctypes.POINTER(ExaRasterSettings)),
        ('P', ctypes.c_int32), ('sh_M', ctypes.c_int32),
        ('means3D', c_void_p), ('shs', c_void_p), ('colors_precomp', c_void_p), ('opacities', c_void_p),
        ('scales', c_void_p), ('rotations', c_void_p), ('cov3D_precomp', c_void_p),
        ('radii', c_void_p),
        ('geom_ws', c_void_p), ('tile_ws', c_void_p),
        ('bin_ws', c_void_p), ('capacity', ctypes.c_uint64),
        ('out_color', c_void_p), ('out_depth', c_void_p), ('out_alpha', c_void_p),
        ('keep_sorted_keys', ctypes.c_int32),
        ('host_header', c_void_p), ('header_tag', ctypes.c_uint32),
        ('is_vis', c_void_p),
    ]


class ExaRasterComposeJob(ctypes.Structure):
    """One composite render of two finished renders of the same camera (include/exa_raster.h)."""
    _fields_ = [
        ('settings', ctypes.POINTER(ExaRasterSettings)),
        ('P_a', ctypes.c_int32), ('P_b', ctypes.c_int32),
        ('geom_a', c_void_p), ('tile_a', c_void_p), ('bin_a', c_void_p), ('capacity_a', ctypes.c_uint64),
        ('geom_b', c_void_p), ('tile_b', c_void_p), ('bin_b', c_void_p), ('capacity_b', ctypes.c_uint64),
        ('tile_ws', c_void_p), ('bin_ws', c_void_p), ('capacity', ctypes.c_uint64),
        ('out_color', c_void_p), ('out_depth', c_void_p), ('out_alpha', c_void_p),
        ('host_header', c_void_p), ('header_tag', ctypes.c_uint32),
        ('a_color', c_void_p), ('a_depth', c_void_p), ('a_alpha', c_void_p), ('a_bg', c_void_p),
        ('radii_a', c_void_p), ('radii_b', c_void_p), ('radii_out', c_void_p),
        ('is_vis_a', c_void_p), ('is_vis_b', c_void_p), ('is_vis_out', c_void_p),
    ]


class ExaRasterBackwardJob(ctypes.Structure):
    """One render of a batched backward call (include/exa_raster.h)."""
    _fields_ = [
        ('settings', ctypes.POINTER(ExaRasterSettings)),
        ('P', ctypes.c_int32), ('sh_M', ctypes.c_int32),
        ('means3D', c_void_p), ('shs', c_void_p), ('colors_precomp', c_void_p), ('opacities', c_void_p),
        ('scales', c_void_p), ('rotations', c_void_p), ('cov3D_precomp', c_void_p),
        ('radii', c_void_p),
        ('geom_ws', c_void_p), ('tile_ws', c_void_p), ('bin_ws', c_void_p), ('capacity', ctypes.c_uint64),
        ('dL_dcolor', c_void_p), ('dL_ddepth', c_void_p), ('dL_dalpha', c_void_p),
        ('grad_ws', c_void_p),
        ('dL_dmeans2D', c_void_p), ('dL_dmeans3D', c_void_p), ('dL_dcolors', c_void_p), ('dL_dopacity', c_void_p),
        ('dL_dscales', c_void_p), ('dL_drotations', c_void_p), ('dL_dsh', c_void_p), ('dL_dcov3D', c_void_p),
        ('densify_grad_accum', c_void_p), ('densify_track_cnt', c_void_p), ('densify_radius_max', c_void_p),
        ('grad_first', ctypes.c_int32),
        ('compose_geom_a', c_void_p), ('compose_P_a', ctypes.c_int32), ('compose_capacity_b', ctypes.c_uint64),
        ('dL_dcolor_indirect', c_void_p),
        ('accumulate', ctypes.c_int32),
        ('used_slots', ctypes.c_uint32),
    ]


STORE_CTX, STAGE_NO_BLEND, STAGE_BLEND_ONLY, STAGE_NO_SORT, STAGE_SORT_ONLY = 1, 2, 4, 8, 16       # bits of `store_ctx` (include/exa_raster.h, EXA_RASTER_STAGE_*)

# symbol -> (restype, argtypes); must list every function include/exa_raster.h declares
_I32 = ctypes.c_int32
_U64 = ctypes.c_uint64
_SP = ctypes.POINTER(ExaRasterSettings)
SIGNATURES = {
    'exa_raster_version': (ctypes.c_int, []),
    'exa_raster_last_error': (ctypes.c_char_p, []),
    'exa_raster_workspace_sizes': (ctypes.c_int, [_I32, _I32, _I32, _U64, ctypes.POINTER(ExaRasterWorkspaceSizes)]),
    'exa_raster_forward_bin': (ctypes.c_int, [_SP, _I32, _I32] + [c_void_p] * 7 + [c_void_p, c_void_p, c_void_p, c_void_p]),
    'exa_raster_forward_render': (ctypes.c_int, [_SP, _I32, c_void_p, c_void_p, c_void_p, _U64,
                                                 c_void_p, c_void_p, c_void_p, _I32, c_void_p]),
    'exa_raster_forward': (ctypes.c_int, [_SP, _I32, _I32] + [c_void_p] * 7 + [c_void_p, c_void_p, c_void_p, c_void_p,
                                                                                 _U64, c_void_p, c_void_p, c_void_p,
                                                                                 _I32, c_void_p]),
    'exa_raster_backward': (ctypes.c_int, [_SP, _I32, _I32] + [c_void_p] * 7 + [c_void_p, c_void_p, c_void_p, c_void_p,
                                                                                  _U64, c_void_p, c_void_p, c_void_p,
                                                                                  c_void_p] + [c_void_p] * 8
                            + [c_void_p]),
    'exa_raster_forward_bin_batch': (ctypes.c_int, [ctypes.POINTER(ExaRasterForwardJob), _I32, c_void_p]),
    'exa_raster_forward_render_batch': (ctypes.c_int, [ctypes.POINTER(ExaRasterForwardJob), _I32, _I32, c_void_p]),
    'exa_raster_forward_batch': (ctypes.c_int, [ctypes.POINTER(ExaRasterForwardJob), _I32, _I32, c_void_p]),
    'exa_raster_backward_batch': (ctypes.c_int, [ctypes.POINTER(ExaRasterBackwardJob), _I32, _I32, c_void_p]),
    'exa_raster_compose_sizes': (ctypes.c_int, [_I32, _I32, _U64, _U64, ctypes.POINTER(ExaRasterWorkspaceSizes)]),
    'exa_raster_forward_compose_batch': (ctypes.c_int, [ctypes.POINTER(ExaRasterComposeJob), _I32, _I32, c_void_p]),
    'exa_raster_read_header_async': (ctypes.c_int, [c_void_p, c_void_p, c_void_p]),
    'exa_raster_read_header_full_async': (ctypes.c_int, [c_void_p, c_void_p, c_void_p]),
    'exa_raster_host_device_pointer': (ctypes.c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    'exa_raster_header_status': (ctypes.c_int, [ctypes.POINTER(ExaRasterHeader)]),
    'exa_raster_camera_block': (ctypes.c_int, [c_void_p, c_void_p, ctypes.POINTER(ctypes.c_float), c_void_p, c_void_p,
                                               c_void_p, c_void_p, ctypes.c_float, ctypes.c_float, c_void_p,
                                               ctypes.c_uint32, c_void_p]),
    'exa_raster_select_row': (ctypes.c_int, [c_void_p, _I32, _I32, c_void_p, c_void_p, c_void_p]),
    'exa_raster_store_pointers': (ctypes.c_int, [c_void_p, ctypes.POINTER(c_void_p), _I32, c_void_p]),
    'exa_raster_mark_visible': (ctypes.c_int, [_SP, _I32, c_void_p, c_void_p, c_void_p]),
    'exa_raster_densify_stats': (ctypes.c_int, [_I32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'exa_ssim_forward': (ctypes.c_int, [_I32, _I32, _I32] + [c_void_p] * 7),
    'exa_ssim_backward': (ctypes.c_int, [_I32, _I32, _I32] + [c_void_p] * 8),
    'exa_photo_loss_blocks': (ctypes.c_int64, [_I32, _I32, _I32, _I32]),
    'exa_photo_loss_forward': (ctypes.c_int, [_I32] * 4 + [ctypes.POINTER(ctypes.c_int32)] + [c_void_p] * 7),
    'exa_photo_loss_grad': (ctypes.c_int, [_I32] * 4 + [ctypes.POINTER(ctypes.c_int32)] + [c_void_p] * 4 +
                            [ctypes.c_float, ctypes.c_float] + [c_void_p] * 3),
    'exa_l1_forward': (ctypes.c_int, [_I32] * 4 + [ctypes.POINTER(ctypes.c_int32)] + [c_void_p] * 6),
    'exa_l1_backward': (ctypes.c_int, [_I32] * 4 + [ctypes.POINTER(ctypes.c_int32)] + [c_void_p] * 7),
    'exa_raster_timing_enable': (ctypes.c_int, [_I32]),
    'exa_raster_timing_read': (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), _I32]),
    'exa_raster_timing_name': (ctypes.c_char_p, [_I32]),
}
TIMING_SLOTS = 9

_lib = None


def load():
    """Load the shared library (once). ``torch`` must be imported first so that the HIP runtime that
    torch bundles (SONAME libamdhip64.so.7) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (ensures torch's libamdhip64 is already mapped)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'exavatar_release_amd: %s is missing. Build it with `python -m exavatar_release_amd.build` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.exa_raster_version() < 139:
        raise RuntimeError('exavatar_release_amd: libexa_raster.so is too old')
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().exa_raster_last_error()
        raise RuntimeError('%s (status %d)' % (msg.decode() if msg else 'exa_raster error', rc))


def workspace_sizes(P, W, H, capacity):
    out = ExaRasterWorkspaceSizes()
    check(load().exa_raster_workspace_sizes(P, W, H, capacity, ctypes.byref(out)))
    return out


def timing_enable(on):
    check(load().exa_raster_timing_enable(int(bool(on))))


def timing_read():
    """{kernel name: milliseconds} of the library's most recent launches (timing must be enabled)."""
    lib = load()
    buf = (ctypes.c_float * TIMING_SLOTS)()
    check(lib.exa_raster_timing_read(buf, TIMING_SLOTS))
    return {lib.exa_raster_timing_name(i).decode(): float(buf[i]) for i in range(TIMING_SLOTS) if buf[i] >= 0}
