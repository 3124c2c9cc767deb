"""ctypes binding of libsmr_hip.so (include/smr.h).

This is the only way Python reaches the renderer: there is no CPU fallback.  If the HIP
library is missing or fails to load, importing callers get an ImportError — loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMR_LIB") or os.path.join(_HERE, "libsmr_hip.so")  # SMR_LIB: A/B builds (tools/variant.sh)

SMR_OK, SMR_ERR_INVALID, SMR_ERR_OOM, SMR_ERR_INTERNAL = 0, -1, -2, -3
MODE_GPU_OPTIMIZED, MODE_CPU_OPTIMIZED = 0, 1
PX_RGBA8, PX_RGBA16F, PX_R8, PX_RG8 = 0, 1, 2, 3
(FRAME_PLANAR_YUV420, FRAME_PLANAR_YUV422, FRAME_PLANAR_YUV444, FRAME_PLANAR_YUVJ420, FRAME_UYVY422, FRAME_YUYV422,
 FRAME_NV12, FRAME_BGRA, FRAME_ARGB, FRAME_RGBA) = range(10)
MAX_MASKS = 20
NO_SOURCE = 0xFFFFFFFF
SOURCE_NONE, SOURCE_SURFACE, SOURCE_FRAME, SOURCE_OPAQUE_SURFACE = 0, 1, 2, 3
SHADER_GAUSSIAN_BLUR, SHADER_GRADIENT, SHADER_RED_BORDER, SHADER_CIRCLE_LAYOUT = 0, 1, 2, 3
SHADER_FADE_TO_BALL, SHADER_LAYOUT_PLANES, SHADER_COLOR_BY_TEXTURE_COUNT, SHADER_SILLY = 4, 5, 6, 7
SHADER_MAX_SOURCES = 16

# every symbol include/smr.h declares (checked by tests/test_abi.py without a GPU)
EXPORTS = [
    "smr_ctx_create", "smr_ctx_destroy", "smr_last_error", "smr_sync", "smr_timer_start", "smr_timer_stop",
    "smr_profile_enable", "smr_profile_read", "smr_profile_reset",
    "smr_surface_create", "smr_surface_wrap", "smr_surface_destroy", "smr_surface_info_get", "smr_surface_upload",
    "smr_surface_download", "smr_surface_clear",
    "smr_frame_create", "smr_frame_destroy", "smr_frame_upload", "smr_frame_download",
    "smr_host_alloc", "smr_host_free", "smr_frame_upload_async", "smr_frame_download_async",
    "smr_frame_to_rgba", "smr_add_premultiplied_alpha", "smr_remove_premultiplied_alpha",
    "smr_rgba_to_frame", "smr_frame_fill_black",
    "smr_resample_plan_make", "smr_resample", "smr_resample_pass", "smr_downsample", "smr_rescale_bilinear", "smr_frame_preprocess", "smr_debug_kernel_launches",
    "smr_apply_layouts", "smr_render_layouts", "smr_ingest_resample", "smr_ingest_resample_batch", "smr_blit_glyphs", "smr_builtin_shader",
    "smr_scene_create", "smr_scene_destroy", "smr_scene_last_error", "smr_scene_register_image", "smr_scene_set_text_measurer", "smr_scene_update",
    "smr_scene_parse",
    "smr_scene_node_count", "smr_scene_node_info", "smr_scene_node_children", "smr_scene_node_layouts",
    "smr_cubic_bezier_easing", "smr_bounce_easing", "smr_parse_color", "smr_ctx_mode", "smr_ctx_set_option",
    "smr_renderer_create", "smr_renderer_destroy", "smr_renderer_last_error", "smr_renderer_register_input",
    "smr_renderer_unregister_input", "smr_renderer_register_image", "smr_renderer_register_shader", "smr_renderer_update_scene",
    "smr_renderer_unregister_output", "smr_renderer_node_count", "smr_renderer_node_info", "smr_renderer_set_text", "smr_renderer_set_text_measurer",
    "smr_renderer_render", "smr_renderer_add_lane", "smr_renderer_sync",
    "smr_comm_create_local", "smr_comm_unique_id", "smr_comm_create_rank", "smr_comm_destroy", "smr_comm_world", "smr_comm_rank",
    "smr_comm_last_error", "smr_gather_tiles",
    "smr_fontbook_create", "smr_fontbook_destroy", "smr_fontbook_last_error", "smr_fontbook_add_file", "smr_fontbook_add_memory",
    "smr_fontbook_add_dir", "smr_fontbook_count", "smr_fontbook_measure", "smr_fontbook_rasterise", "smr_renderer_set_fontbook",
    "smr_abi_version", "smr_build_flags", "smr_sizeof_layout",
]
NO_RESOLUTION = 0xFFFFFFFF
NODE_INPUT_STREAM, NODE_LAYOUT, NODE_TEXT, NODE_IMAGE, NODE_SHADER = 0, 1, 2, 3, 4


class SceneNode(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("parent", C.c_int32), ("n_children", C.c_uint32), ("width", C.c_uint32),
                ("height", C.c_uint32), ("ref_id", C.c_char_p), ("id", C.c_char_p), ("payload", C.c_char_p)]


class Mask(C.Structure):
    _fields_ = [("radius", C.c_float * 4), ("top", C.c_float), ("left", C.c_float), ("width", C.c_float), ("height", C.c_float)]


class Layout(C.Structure):
    _fields_ = [
        ("top", C.c_float), ("left", C.c_float), ("width", C.c_float), ("height", C.c_float),
        ("rotation_degrees", C.c_float),
        ("border_radius", C.c_float * 4),
        ("type", C.c_uint32),
        ("source_index", C.c_uint32),
        ("color", C.c_float * 4),
        ("border_color", C.c_float * 4),
        ("border_width", C.c_float),
        ("crop", C.c_float * 4),
        ("blur_radius", C.c_float),
        ("masks_len", C.c_uint32),
        ("masks", Mask * MAX_MASKS),
    ]


class Frame(C.Structure):
    _fields_ = [("format", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("planes", C.c_void_p * 3)]


class SurfaceInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("owned", C.c_uint32),
                ("pitch", C.c_size_t), ("dptr", C.c_void_p)]


class ResamplePlan(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("levels", C.c_int32 * 2), ("reduced_w", C.c_int32), ("reduced_h", C.c_int32),
        ("axis", C.c_int32 * 2), ("scale", C.c_float * 2), ("offset", C.c_float * 2), ("perp_offset", C.c_int32 * 2),
        ("mid_w", C.c_int32), ("mid_h", C.c_int32),
    ]


class Glyph(C.Structure):
    _fields_ = [("dst_x", C.c_int32), ("dst_y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("atlas_x", C.c_int32), ("atlas_y", C.c_int32), ("color", C.c_float * 4)]


class TextParams(C.Structure):
    _fields_ = [("text", C.c_char_p), ("font_family", C.c_char_p), ("style", C.c_char_p), ("weight", C.c_char_p), ("wrap", C.c_char_p),
                ("align", C.c_char_p), ("font_size", C.c_float), ("line_height", C.c_float), ("max_width", C.c_float), ("max_height", C.c_float)]


class TextRun(C.Structure):
    _fields_ = [("glyphs", C.POINTER(Glyph)), ("n_glyphs", C.c_uint32), ("atlas", C.POINTER(C.c_uint8)), ("atlas_w", C.c_uint32), ("atlas_h", C.c_uint32)]


TEXT_MEASURE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(TextParams), C.POINTER(C.c_float), C.POINTER(C.c_uint32))


class Source(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("surface", C.c_void_p), ("frame", C.POINTER(Frame))]


class GaussianBlurParams(C.Structure):
    _fields_ = [("sigma", C.c_float)]


class InputFrame(C.Structure):
    _fields_ = [("input_id", C.c_char_p), ("frame", C.POINTER(Frame)), ("pts_ns", C.c_int64)]


class OutputFrame(C.Structure):
    _fields_ = [("output_id", C.c_char_p), ("frame", C.POINTER(Frame)), ("ctx", C.c_void_p)]


_lib = None


def load():
    """Load libsmr_hip.so.  Raises ImportError when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m smelter_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise ImportError(f"cannot load {LIB_PATH}: {e}") from e
    P, I, U, F = C.c_void_p, C.c_int, C.c_uint32, C.c_float
    PP = C.POINTER(C.c_void_p)
    sig = {
        "smr_ctx_create": ([I, U, U, P, PP], I),
        "smr_ctx_destroy": ([P], None),
        "smr_last_error": ([P], C.c_char_p),
        "smr_sync": ([P], I),
        "smr_timer_start": ([P], I),
        "smr_timer_stop": ([P, C.POINTER(F)], I),
        "smr_profile_enable": ([P, I], I),
        "smr_profile_read": ([P, I, C.POINTER(F), C.POINTER(U)], I),
        "smr_profile_reset": ([P], I),
        "smr_surface_create": ([P, U, U, U, PP], I),
        "smr_surface_wrap": ([P, P, C.c_size_t, U, U, U, PP], I),
        "smr_surface_destroy": ([P, P], None),
        "smr_surface_info_get": ([P, C.POINTER(SurfaceInfo)], I),
        "smr_surface_upload": ([P, P, P, C.c_size_t], I),
        "smr_surface_download": ([P, P, P, C.c_size_t], I),
        "smr_surface_clear": ([P, P], I),
        "smr_frame_create": ([P, U, U, U, C.POINTER(Frame)], I),
        "smr_frame_destroy": ([P, C.POINTER(Frame)], None),
        "smr_frame_upload": ([P, C.POINTER(Frame), PP], I),
        "smr_frame_download": ([P, C.POINTER(Frame), PP], I),
        "smr_host_alloc": ([P, C.c_size_t, PP], I),
        "smr_host_free": ([P, P], None),
        "smr_frame_upload_async": ([P, C.POINTER(Frame), PP], I),
        "smr_frame_download_async": ([P, C.POINTER(Frame), PP], I),
        "smr_frame_to_rgba": ([P, C.POINTER(Frame), P], I),
        "smr_add_premultiplied_alpha": ([P, P, P], I),
        "smr_remove_premultiplied_alpha": ([P, P, P], I),
        "smr_rgba_to_frame": ([P, P, C.POINTER(Frame)], I),
        "smr_frame_fill_black": ([P, C.POINTER(Frame)], I),
        "smr_resample_plan_make": ([U, U, C.POINTER(F), U, U, C.POINTER(ResamplePlan)], I),
        "smr_resample": ([P, P, C.POINTER(F), P], I),
        "smr_resample_pass": ([P, P, I, F, F, I, P], I),
        "smr_downsample": ([P, P, U, U, P], I),
        "smr_rescale_bilinear": ([P, P, P], I),
        "smr_apply_layouts": ([P, P, C.POINTER(Layout), U, PP, U], I),
        "smr_render_layouts": ([P, C.POINTER(Layout), U, C.POINTER(Source), U, U, U, C.POINTER(Frame), P], I),
        "smr_ingest_resample": ([P, C.POINTER(Frame), C.POINTER(F), P], I),
        "smr_ingest_resample_batch": ([P, C.POINTER(C.POINTER(Frame)), C.POINTER(F), PP, U, C.POINTER(I)], I),
        "smr_blit_glyphs": ([P, P, C.POINTER(F), C.POINTER(Glyph), U, P, U, U], I),
        "smr_builtin_shader": ([P, U, P, C.c_size_t, PP, U, P, F], I),
        "smr_scene_create": ([PP], I),
        "smr_scene_destroy": ([P], None),
        "smr_scene_last_error": ([P], C.c_char_p),
        "smr_scene_register_image": ([P, C.c_char_p, U, U], I),
        "smr_scene_set_text_measurer": ([P, TEXT_MEASURE_FN, P], I),
        "smr_renderer_set_text_measurer": ([P, TEXT_MEASURE_FN, P], I),
        "smr_scene_update": ([P, C.c_char_p, U, U], I),
        "smr_scene_parse": ([P, C.c_char_p, C.POINTER(C.c_char_p)], I),
        "smr_scene_node_count": ([P], I),
        "smr_scene_node_info": ([P, I, C.POINTER(SceneNode)], I),
        "smr_scene_node_children": ([P, I, C.POINTER(C.c_int32), U], I),
        "smr_scene_node_layouts": ([P, I, C.c_int64, C.POINTER(U), U, U, C.POINTER(Layout), U, C.POINTER(U), C.POINTER(U),
                                    C.POINTER(U)], I),
        "smr_cubic_bezier_easing": ([C.c_double] * 5, C.c_double),
        "smr_bounce_easing": ([C.c_double], C.c_double),
        "smr_parse_color": ([C.c_char_p, C.POINTER(C.c_uint8)], I),
        "smr_ctx_mode": ([P], U),
        "smr_ctx_set_option": ([P, U, C.c_int32], I),
        "smr_frame_preprocess": ([P, P, U, U, P, C.c_size_t], I),
        "smr_debug_kernel_launches": ([P, U, C.POINTER(C.c_uint64)], I),
        "smr_comm_create_local": ([PP, U, PP], I),
        "smr_comm_unique_id": ([C.POINTER(C.c_uint8)], I),
        "smr_comm_create_rank": ([P, U, U, C.POINTER(C.c_uint8), PP], I),
        "smr_comm_destroy": ([P], None),
        "smr_comm_world": ([P], U),
        "smr_comm_rank": ([P], U),
        "smr_comm_last_error": ([P], C.c_char_p),
        "smr_gather_tiles": ([P, U, C.POINTER(U), PP, PP, U], I),
        "smr_renderer_create": ([P, C.c_int64, PP], I),
        "smr_renderer_destroy": ([P], None),
        "smr_renderer_last_error": ([P], C.c_char_p),
        "smr_renderer_register_input": ([P, C.c_char_p], I),
        "smr_renderer_unregister_input": ([P, C.c_char_p], I),
        "smr_renderer_register_image": ([P, C.c_char_p, P, U, U], I),
        "smr_renderer_register_shader": ([P, C.c_char_p, U], I),
        "smr_renderer_update_scene": ([P, C.c_char_p, U, U, U, C.c_char_p], I),
        "smr_renderer_unregister_output": ([P, C.c_char_p], I),
        "smr_renderer_node_count": ([P, C.c_char_p], I),
        "smr_renderer_node_info": ([P, C.c_char_p, I, C.POINTER(SceneNode)], I),
        "smr_renderer_set_text": ([P, C.c_char_p, I, C.POINTER(F), C.POINTER(Glyph), U, P, U, U], I),
        "smr_renderer_render": ([P, C.c_int64, C.POINTER(InputFrame), U, C.POINTER(OutputFrame), U, C.POINTER(U)], I),
        "smr_renderer_add_lane": ([P, P], I),
        "smr_renderer_sync": ([P], I),
        "smr_fontbook_create": ([PP], I),
        "smr_fontbook_destroy": ([P], None),
        "smr_fontbook_last_error": ([P], C.c_char_p),
        "smr_fontbook_add_file": ([P, C.c_char_p], I),
        "smr_fontbook_add_memory": ([P, P, C.c_size_t], I),
        "smr_fontbook_add_dir": ([P, C.c_char_p], I),
        "smr_fontbook_count": ([P], U),
        "smr_fontbook_measure": ([P, C.POINTER(TextParams), C.POINTER(F), C.POINTER(U)], I),
        "smr_fontbook_rasterise": ([P, C.POINTER(TextParams), U, U, C.POINTER(F), C.POINTER(TextRun)], I),
        "smr_renderer_set_fontbook": ([P, P], I),
        "smr_abi_version": ([], U),
        "smr_build_flags": ([], U),
        "smr_sizeof_layout": ([], U),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    assert lib.smr_sizeof_layout() == C.sizeof(Layout), "smr_layout ABI mismatch"
    _lib = lib
    return lib
