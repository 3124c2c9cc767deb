/*
 * smr.h — C ABI of libsmr_hip: the MI355X (gfx950) scene rasteriser that replaces the
 * wgpu back half of smelter-render (everything below scene/flatten).
 *
 * What binds to it: smelter-render's InputTexture / ResampledChild / LayoutShader /
 * OutputTexture / FramePreProcessor (see INTEGRATION.md for the Rust `extern "C"`
 * block a maintainer adds).  Each entry point names the reference interface it
 * replaces (paths relative to the smelter repository root).
 *
 * Conventions
 *   - every function returns 0 (SMR_OK) or a negative smr_status; the message is
 *     available from smr_last_error(ctx) (ctx-owned, valid until the next call).
 *     Error classes mirror smelter-render/src/wgpu.rs:78-97 WgpuError::{Validation,
 *     OutOfMemory, Internal}.
 *   - no global state, no callbacks.  A ctx may be used from any thread but calls must
 *     be externally serialised (smelter-render already holds one Mutex around the
 *     renderer: smelter-render/src/state.rs:55).
 *   - all work is enqueued on the ctx stream; only *_download, smr_sync and
 *     smr_timer_stop block the host.
 *   - surfaces live in HBM as pitched 2-D arrays (pitch is a multiple of 256 B);
 *     host buffers handed to upload/download are tightly packed unless a pitch is
 *     given (reference: wgpu/texture/base.rs:61-77, output_texture.rs:105-108).
 *   - RGBA8 node surfaces hold premultiplied alpha.  In SMR_MODE_GPU_OPTIMIZED their
 *     bytes are sRGB-encoded and every filter / blend happens in linear light
 *     (Rgba8UnormSrgb + two views: wgpu/texture/rgba_multiview.rs:17-49,
 *     state/node_texture.rs:104-142); in SMR_MODE_CPU_OPTIMIZED they are plain unorm.
 */
#ifndef SMR_H
#define SMR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMR_API __attribute__((visibility("default")))

typedef struct smr_ctx smr_ctx;
typedef struct smr_surface smr_surface;

typedef enum smr_status {
    SMR_OK = 0,
    SMR_ERR_INVALID = -1,  /* WgpuError::Validation  */
    SMR_ERR_OOM = -2,      /* WgpuError::OutOfMemory */
    SMR_ERR_INTERNAL = -3, /* WgpuError::Internal    */
} smr_status;

/* smelter-render/src/types.rs:8-18 RenderingMode (WebGl is wasm-only: out of scope) */
typedef enum smr_mode { SMR_MODE_GPU_OPTIMIZED = 0, SMR_MODE_CPU_OPTIMIZED = 1 } smr_mode;

typedef enum smr_pixel_format {
    SMR_PX_RGBA8 = 0,   /* node texture (wgpu Rgba8UnormSrgb / Rgba8Unorm)          */
    SMR_PX_RGBA16F = 1, /* resampler intermediate (layout/resampler.rs:25-28)       */
    SMR_PX_R8 = 2,      /* Y / U / V plane (wgpu/texture/planar_yuv.rs:111-129)     */
    SMR_PX_RG8 = 3,     /* NV12 chroma plane (wgpu/texture/nv12.rs)                 */
} smr_pixel_format;

/* smelter-render/src/types.rs:27-40 FrameData */
typedef enum smr_frame_format {
    SMR_FRAME_PLANAR_YUV420 = 0,
    SMR_FRAME_PLANAR_YUV422 = 1,
    SMR_FRAME_PLANAR_YUV444 = 2,
    SMR_FRAME_PLANAR_YUVJ420 = 3,
    SMR_FRAME_UYVY422 = 4,
    SMR_FRAME_YUYV422 = 5,
    SMR_FRAME_NV12 = 6,
    SMR_FRAME_BGRA = 7,
    SMR_FRAME_ARGB = 8,
    SMR_FRAME_RGBA = 9, /* Rgba8UnormWgpuTexture: straight alpha, premultiplied on ingest */
} smr_frame_format;

/* One video frame resident in HBM: up to three pitched plane surfaces.
 * planes[] per format: planar YUV {Y,U,V}; NV12 {Y (R8), UV (RG8)}; packed formats {data}.
 * UYVY/YUYV planes are RGBA8 surfaces of width/2 texels (wgpu/texture/interleaved_yuv422.rs). */
typedef struct smr_frame {
    uint32_t format; /* smr_frame_format */
    uint32_t width;
    uint32_t height;
    smr_surface *planes[3];
} smr_frame;

typedef struct smr_surface_info {
    uint32_t width, height, format;
    uint32_t owned; /* 0 when wrapped around caller memory */
    size_t pitch;   /* bytes per row */
    void *dptr;     /* device pointer */
} smr_surface_info;

#define SMR_MAX_MASKS 20            /* layout/params.rs:15 */
#define SMR_DEFAULT_MAX_LAYOUTS 100 /* layout.rs:23 */
#define SMR_NO_SOURCE 0xffffffffu

/* layout/layout.rs:47-55 Mask + params.rs:292-300 byte order */
typedef struct smr_mask {
    float radius[4]; /* top_left, top_right, bottom_right, bottom_left */
    float top, left, width, height;
} smr_mask;

/* POD mirror of RenderLayout (layout.rs:58-96) in the conventions of
 * ParamsBindGroups::update (layout/params.rs:169-334): colours are premultiplied
 * and already mode-converted by the host (wgpu/utils.rs:51-81). */
typedef struct smr_layout {
    float top, left, width, height;
    float rotation_degrees;
    float border_radius[4]; /* tl, tr, br, bl */
    uint32_t type;          /* 0 texture (ChildNode), 1 colour, 2 box shadow */
    uint32_t source_index;  /* texture: index into sources[]; SMR_NO_SOURCE = empty 1x1 */
    float color[4];
    float border_color[4];
    float border_width;
    float crop[4]; /* top, left, width, height (layout.rs:39-45) */
    float blur_radius;
    uint32_t masks_len;
    smr_mask masks[SMR_MAX_MASKS];
} smr_layout;

/* Resampler pass plan (layout/resampler.rs:36-145, 305-378) */
typedef struct smr_resample_plan {
    int32_t kind;      /* 0 direct (layout shader samples the source 1:1), 1 single pass, 2 separable */
    int32_t levels[2]; /* box pre-decimation levels [horizontal, vertical] */
    int32_t reduced_w, reduced_h;
    int32_t axis[2]; /* per pass, in execution order: 0 horizontal, 1 vertical */
    float scale[2];
    float offset[2];
    int32_t perp_offset[2];
    int32_t mid_w, mid_h; /* separable: f16 intermediate size */
} smr_resample_plan;

/* One glyph quad of a text node (transformations/text_renderer.rs:92-132). */
typedef struct smr_glyph {
    int32_t dst_x, dst_y;
    int32_t w, h;
    int32_t atlas_x, atlas_y;
    float color[4]; /* straight RGBA 0..1, gamma-encoded */
} smr_glyph;

/* Source of a texture layout for the fused render entry point. */
typedef enum smr_source_kind {
    SMR_SOURCE_NONE = 0,
    SMR_SOURCE_SURFACE = 1,
    SMR_SOURCE_FRAME = 2,
    SMR_SOURCE_OPAQUE_SURFACE = 3 /* an RGBA8 surface the caller knows to be alpha == 255 everywhere (e.g. a tile produced by
                                     smr_ingest_resample on another GPU): lets the compositor treat it as a base layer */
} smr_source_kind;
typedef struct smr_source {
    uint32_t kind;
    const smr_surface *surface; /* RGBA8 node surface (text, image, shader output, ...) */
    const smr_frame *frame;     /* raw input frame: colour conversion is fused into the resampler */
} smr_source;

/* ---- context ---------------------------------------------------------------------
 * replaces WgpuCtx::new (smelter-render/src/wgpu/ctx.rs:34-107) + RendererOptions
 * {rendering_mode, max_layouts_count} (state.rs:43-52).
 * hip_stream: an existing hipStream_t to enqueue on, or NULL to let the ctx create one. */
SMR_API int smr_ctx_create(int hip_device, uint32_t mode, uint32_t max_layouts, void *hip_stream, smr_ctx **out);
SMR_API void smr_ctx_destroy(smr_ctx *ctx);
SMR_API const char *smr_last_error(const smr_ctx *ctx);
SMR_API uint32_t smr_ctx_mode(const smr_ctx *ctx);  /* smr_mode the context was created with (RenderingMode, types.rs:8-18) */
SMR_API int smr_sync(smr_ctx *ctx);                 /* device.poll(wait) — render_loop.rs:177-183 */
/* Context options (RendererOptions has no counterpart: these select between equivalent implementations).
 *   SMR_OPT_INGEST_IMPL         how wave A of smr_render_layouts / smr_ingest_resample* turns a Y'CbCr frame into its dst-sized tile:
 *       SMR_INGEST_AUTO      what the reference does, both halves fast (default): every frame goes through the exact converter
 *                            (k_yuv420_to_rgba / k_yuv_to_rgba_batch: the WGSL operation sequence value for value — the node texture's bytes
 *                            ARE the reference's, tests/test_emu_convert.py, tests/test_gpu_parity.py) into its RGBA8 node texture, one launch
 *                            for all frames of a call, and the matrix-core kernel (k_ingest_wave) resamples the nodes, one launch for all of
 *                            them; the general pass kernels where a plan leaves the kernel's windows (smr_debug_kernel_launches tells).
 *                            Within 1 LSB of the reference END TO END on every content class.
 *       SMR_INGEST_VALU_F32  exact f32 everywhere: bit-identical to the pass-per-launch kernels (smr_frame_to_rgba + smr_resample)
 *       SMR_INGEST_MFMA_F16, SMR_INGEST_MFMA_F16_NODE  aliases of AUTO (names of earlier revisions, kept for callers)
 *       (5, the conversion fused into the resampler with a folded-FMA BT.709 matrix — within one code per stage but not within 1 LSB end to end on
 *        adversarial content — is not part of the ABI any more: laboratory builds only, -DSMR_LAB)
 *       (3, round 2's workgroup-pipelined kernel k_ingest_mfma, was retired in round 4: smr_ctx_set_option rejects it)
 *     The matrix-core resampler keeps every quantisation point of the reference (u8 node texture, f16 between the passes,
 *     layout/resampler.rs:25-28, u8 sRGB tile); its operands are f16 pairs (texels and the weights of both passes), accumulated in f32: within
 *     1 LSB — on every byte of every content class — of resample.wgsl's passes applied to the node texture.
 *   SMR_OPT_INGEST_STRIP_WIDTH  strip width of the f32 kernel: 0 = chosen per job (default), 32 or 64 (tests, profiling)
 *   SMR_OPT_DIRECT_OUTPUT       1: when smr_render_layouts sees the same layout list again (a scene at rest), the pixels the
 *                               compositor would only copy from a freshly resampled input are converted to Y'CbCr by the resampling
 *                               kernel itself and their RGBA8 form is never stored (HBM traffic per frame 1.5x instead of 2.9x the
 *                               algorithmic bytes, at the price of vector-ALU time in a kernel that is instruction-bound: DESIGN.md); 0 (default): always through
 *                               the RGBA8 tile.  Same output bytes either way. */
typedef enum smr_ingest_impl {
    SMR_INGEST_AUTO = 0, SMR_INGEST_VALU_F32 = 1, SMR_INGEST_MFMA_F16 = 2, /* 3: retired */ SMR_INGEST_MFMA_F16_NODE = 4 /* 5: laboratory builds only */
} smr_ingest_impl;
/*   SMR_OPT_CONVERT_IMPL        which kernels smr_frame_to_rgba (InputTexture::convert_to_node_texture) runs — all of them produce the same bytes:
 *       SMR_CONVERT_AUTO       the block converters: k_yuv420_to_rgba (4:2:0 planar / NV12: a thread per 4 x 4 block, shared chroma work) and
 *                              k_yuv_to_rgba_batch (4:2:2, 4:4:4, packed YUV, BGRA / ARGB), the pass kernels for what those leave (default)
 *       SMR_CONVERT_GENERAL    one kernel per WGSL pass, one launch per frame, coordinates and divisions as the shader writes them (tests: the
 *                              block converters are held to these bit for bit)
 *       SMR_CONVERT_BLOCK_4X2  k_yuv_to_rgba_batch for 4:2:0 frames too (round 3's converter; A/B) */
typedef enum smr_convert_impl { SMR_CONVERT_AUTO = 0, SMR_CONVERT_GENERAL = 1, SMR_CONVERT_BLOCK_4X2 = 2 } smr_convert_impl;
/*   SMR_OPT_COMPACT_NODES       1 (default): the node texture of a 4:2:0 frame that only the matrix-core resampler reads is written as RGB12 — 12 bytes
 *                               per four pixels (R x 4, G x 4, B x 4; alpha is 1 for every Y'CbCr frame and is not stored): a quarter less traffic on
 *                               either side of the intermediate the default route is bound by; 0: always RGBA8.  Same codes, same tiles bit for bit. */
/*   SMR_OPT_PLANE_SOURCE        1: a planar 4:2:0 / NV12 frame whose layout plan lies inside the matrix-core resampler's class windows (scales around 1.5, 2
 *                               and 3: every input of the benchmark scenes) is read by that kernel as it is — the wave converts each chunk of its window with
 *                               the exact converter's own block arithmetic (the node texture's bytes, bit for bit) into LDS and resamples from there: the
 *                               reference's node texture never exists in memory (half the route's traffic, one launch fewer; the same pixels within the
 *                               resampler's 1 LSB).  0 (default): the converter writes the node texture (RGB12 with SMR_OPT_COMPACT_NODES) and the
 *                               resampler reads it back — faster on every measured workload: a wave converts its window's overlaps again (1.45 x) at two
 *                               waves per SIMD, the converter kernel converts every pixel once at six. */
/*   SMR_OPT_FUSED_KERNELS       1 (default): smr_render_layouts / smr_ingest_resample* run the fused kernels (waves A and B); 0: one general kernel per pass
 *                               of the reference (convert, resample passes, apply_layouts, rgba_to_yuv) — what the tests hold the fused kernels to.
 *   SMR_OPT_COMPOSE_SELECT      1 (default): compositor tiles in which every pixel is a plain copy from the topmost layer that holds it (the seams of a
 *                               grid of opaque 1:1 layers) are copied; 0: they take the compositing path.  Same output bytes either way.
 *   SMR_OPT_SHARED_DEVICE       1: other contexts run kernels on this device at the same time (a renderer's lanes: frames in flight).  The matrix-core
 *                               resampler then occupies two waves per SIMD instead of all it could hold (it is as fast from two: its time is a
 *                               dependency chain), which leaves a SIMD's remaining registers to the other lane's converter / compositor waves:
 *                               +2 % frames per second with two lanes, -3 % with one.  0 (default).  smr_renderer_add_lane sets it on every lane.
 * No option is read from the environment: a product build of the library calls getenv nowhere (laboratory builds, -DSMR_LAB, read their A/B knobs there). */
typedef enum smr_option {
    SMR_OPT_INGEST_IMPL = 0, SMR_OPT_INGEST_STRIP_WIDTH = 1, SMR_OPT_DIRECT_OUTPUT = 2, SMR_OPT_CONVERT_IMPL = 3, SMR_OPT_COMPACT_NODES = 4,
    SMR_OPT_FUSED_KERNELS = 5, SMR_OPT_COMPOSE_SELECT = 6, SMR_OPT_SHARED_DEVICE = 7, SMR_OPT_PLANE_SOURCE = 8
} smr_option;
SMR_API int smr_ctx_set_option(smr_ctx *ctx, uint32_t option, int32_t value);
SMR_API int smr_timer_start(smr_ctx *ctx);          /* hipEvent on the ctx stream */
SMR_API int smr_timer_stop(smr_ctx *ctx, float *ms); /* records, synchronises, returns elapsed ms */
/* Per-kernel-class timing with HIP events on the ctx stream (off by default).
 * stage ids: 0 ingest/convert, 1 resample, 2 layouts/compose, 3 output convert, 4 fused ingest+resample,
 * 5 fused compose+output. Accumulates ms and launch counts until reset. */
SMR_API int smr_profile_enable(smr_ctx *ctx, int enable);
SMR_API int smr_profile_read(smr_ctx *ctx, int stage, float *total_ms, uint32_t *launches);
SMR_API int smr_profile_reset(smr_ctx *ctx);
/* Which kernels a context has launched since it was created (always counted, no events): the tests assert the path a resample plan x
 * input format takes; a host can watch for scenes that leave the fast paths. */
typedef enum smr_kernel_id {
    SMR_KERNEL_INGEST_WAVE = 0,      /* k_ingest_wave with the fused conversion (laboratory builds only: always 0 in a product build) */
    SMR_KERNEL_INGEST_WAVE_RGBA = 1, /* k_ingest_wave on an RGBA8 / box-reduced RGBA16F node texture (every frame after smr_frame_to_rgba; surfaces) */
    SMR_KERNEL_FRAME_TO_RGBA_420 = 2, /* k_yuv420_to_rgba, the 4:2:0 planar / NV12 block converter: its launches (counted by FRAME_TO_RGBA as well) */
    SMR_KERNEL_INGEST_VALU = 3,      /* k_ingest_resample: fused conversion + Lanczos, every pass in f32 */
    SMR_KERNEL_RESAMPLE_GENERAL = 4, /* smr_resample on a node texture: box pre-reduction and one or two Lanczos pass kernels */
    SMR_KERNEL_FRAME_TO_RGBA = 5,    /* the input converters (InputTexture::convert_to_node_texture): launches — a block-converter launch takes up to 16 frames */
    SMR_KERNEL_COMPOSE_OUTPUT = 6,   /* k_compose_output: layout shader + output conversion */
    SMR_KERNEL_APPLY_LAYOUTS = 7,    /* k_apply_layouts: the general compositor */
    SMR_KERNEL_COUNT_ = 8
} smr_kernel_id;
SMR_API int smr_debug_kernel_launches(const smr_ctx *ctx, uint32_t kernel, uint64_t *count);

/* ---- surfaces (NodeTexture / wgpu::Texture; state/node_texture.rs:11-163) -------- */
SMR_API int smr_surface_create(smr_ctx *ctx, uint32_t w, uint32_t h, uint32_t format, smr_surface **out);
/* Device memory the caller owns (a decoder's output, a torch tensor) as a surface, in place.  pitch >= w * bytes per texel; rows and the base
 * 4-byte aligned for the block converters, 16-byte aligned for the matrix-core resampler's node textures (other alignments take the general
 * kernels).  What the library READS of a wrapped surface:
 *   - 4:2:0 / NV12 planes of a frame (the input converter): the texels of each row and nothing else — every wrapped plane, tight rows
 *     (pitch == bytes per row) or wide ones, is converted by the build of the block converter that requests nothing behind a row's last
 *     column (k_yuv420_to_rgba_tight: the same bytes, ~2 % more instructions), so pitch * (h - 1) + row bytes of allocation suffice;
 *   - RGBA8 / RGBA16F surfaces handed to the resampler or the compositor: whole 16-byte groups inside a row's pitch — the allocation must
 *     back every row out to the pitch rounded down to 16 bytes, the LAST ROW TOO.
 * Planes the library allocates itself (smr_surface_create / smr_frame_create: every allocation ends with 16 spare bytes) take the plain
 * converter. */
SMR_API int smr_surface_wrap(smr_ctx *ctx, void *dptr, size_t pitch, uint32_t w, uint32_t h, uint32_t format,
                             smr_surface **out);
SMR_API void smr_surface_destroy(smr_ctx *ctx, smr_surface *s);
SMR_API int smr_surface_info_get(const smr_surface *s, smr_surface_info *out);
/* queue.write_texture (wgpu/texture/base.rs:61-77); host_pitch 0 = tight rows */
SMR_API int smr_surface_upload(smr_ctx *ctx, smr_surface *s, const void *host, size_t host_pitch);
/* copy_texture_to_buffer + map + strip padding (texture/base.rs:97-118, output_texture.rs:85-113) */
SMR_API int smr_surface_download(smr_ctx *ctx, const smr_surface *s, void *host, size_t host_pitch);
SMR_API int smr_surface_clear(smr_ctx *ctx, smr_surface *s);

/* ---- frames (InputTexture upload side: state/input_texture.rs:69-201) ------------ */
SMR_API int smr_frame_create(smr_ctx *ctx, uint32_t format, uint32_t w, uint32_t h, smr_frame *out);
SMR_API void smr_frame_destroy(smr_ctx *ctx, smr_frame *f);
SMR_API int smr_frame_upload(smr_ctx *ctx, const smr_frame *f, const void *const host_planes[3]);
SMR_API int smr_frame_download(smr_ctx *ctx, const smr_frame *f, void *const host_planes[3]);
/* Pinned host memory and stream-ordered copies (no staging, no implicit synchronisation): what replaces the reference's
 * queue.write_texture staging (input_texture.rs:69-201) and the padded read-back + repack (output_texture.rs:68-113) for hosts
 * that can keep frames in buffers from smr_host_alloc.  Host buffers must stay untouched until smr_sync has returned. */
SMR_API int smr_host_alloc(smr_ctx *ctx, size_t bytes, void **out);
SMR_API void smr_host_free(smr_ctx *ctx, void *p);
SMR_API int smr_frame_upload_async(smr_ctx *ctx, const smr_frame *f, const void *const host_planes[3]);
SMR_API int smr_frame_download_async(smr_ctx *ctx, const smr_frame *f, void *const host_planes[3]);

/* ---- a3: InputTexture::convert_to_node_texture (input_texture.rs:203-219) ---------
 * wgpu/format/{planar_yuv,nv12,interleaved_uyvy,interleaved_yuyv,bgra,argb}_to_rgba.{rs,wgsl};
 * RGBA frames go through add_premultiplied_alpha (input_texture/rgba_texture.rs:38-63). */
SMR_API int smr_frame_to_rgba(smr_ctx *ctx, const smr_frame *in, smr_surface *node);
SMR_API int smr_add_premultiplied_alpha(smr_ctx *ctx, const smr_surface *src, smr_surface *dst);
SMR_API int smr_remove_premultiplied_alpha(smr_ctx *ctx, const smr_surface *src, smr_surface *dst);

/* ---- a11: read_outputs (render_loop.rs:59-230) -------------------------------------
 * RgbaToYuvConverter::convert (format/rgba_to_yuv.rs:67-117) for planar 420/422/444,
 * RgbaToNv12Converter::convert (format/rgba_to_nv12.rs:54-82) for NV12. */
SMR_API int smr_rgba_to_frame(smr_ctx *ctx, const smr_surface *node, const smr_frame *out);
/* PlanarYuvTextures::fill_with_color(BLACK) (wgpu/texture/planar_yuv.rs:190-202) */
SMR_API int smr_frame_fill_black(smr_ctx *ctx, const smr_frame *out);

/* ---- a7/a8: resampler (layout/resampler.rs) ---------------------------------------- */
/* ResampledChild::is_needed + pass planning; pure host function. crop = {top,left,width,height}. */
SMR_API int smr_resample_plan_make(uint32_t src_w, uint32_t src_h, const float crop[4], uint32_t dst_w, uint32_t dst_h,
                                   smr_resample_plan *out);
/* ResampledChild::render: box pre-reduce + 1 or 2 Lanczos3 passes. Returns the plan kind
 * (0 = direct: dst untouched) or a negative status. */
SMR_API int smr_resample(smr_ctx *ctx, const smr_surface *src, const float crop[4], smr_surface *dst);
/* single passes (resample.wgsl / downsample.wgsl), exposed for tests and the Rust seam */
SMR_API int smr_resample_pass(smr_ctx *ctx, const smr_surface *src, int axis, float scale, float offset, int perp_offset,
                              smr_surface *dst);
SMR_API int smr_downsample(smr_ctx *ctx, const smr_surface *src, uint32_t fx, uint32_t fy, smr_surface *dst);
/* format/rgba_rescale.wgsl via FramePreProcessor::rescale_node_texture (frame_pre_processor.rs:117-132) */
SMR_API int smr_rescale_bilinear(smr_ctx *ctx, const smr_surface *src, smr_surface *dst);
/* a15: FramePreProcessor::process_to_bytes (state/frame_pre_processor.rs:84-107) in one call — the frame (device-resident, any
 * smr_frame_format) through convert_to_node_texture into an RGBA8 node texture at its own resolution, the optional bilinear rescale
 * to dst_w x dst_h (0 x 0 = none; filtering in linear light in GpuOptimized mode, as rescale_node_texture does it) and the
 * read-back as tightly packed rows (host_pitch 0) or rows host_pitch bytes apart.  Blocks until the bytes are in `host`.
 * The intermediate surfaces belong to the context and are reused while the resolutions repeat (as the reference's instance does). */
SMR_API int smr_frame_preprocess(smr_ctx *ctx, const smr_frame *in, uint32_t dst_w, uint32_t dst_h, void *host, size_t host_pitch);

/* ---- a9/a10: LayoutShader::render (layout/shader.rs:93-167 + apply_layouts.wgsl) ---
 * Clears target to transparent, draws layouts back to front with premultiplied OVER.
 * sources[i] may be NULL (sampled as the 1x1 transparent texture, layout.rs:204-212). */
SMR_API int smr_apply_layouts(smr_ctx *ctx, smr_surface *target, const smr_layout *layouts, uint32_t n,
                              const smr_surface *const *sources, uint32_t n_sources);

/* ---- fused hot path: LayoutNode::render + read_outputs in (at most) two kernel waves --
 * Equivalent to: for every texture layout resample_scaled_children (layout.rs:238-278) from the
 * raw frame / surface, LayoutShader::render, then rgba_to_yuv / rgba_to_nv12 into `out`
 * (or a copy into `out_rgba` when out == NULL).  Bit-identical to the unfused sequence with SMR_INGEST_VALU_F32,
 * within 1 LSB of it with the matrix-core resampler (SMR_OPT_INGEST_IMPL). */
SMR_API int smr_render_layouts(smr_ctx *ctx, const smr_layout *layouts, uint32_t n, const smr_source *sources,
                               uint32_t n_sources, uint32_t out_w, uint32_t out_h, const smr_frame *out,
                               smr_surface *out_rgba);

/* Per-input half of the above, for sharding inputs across GPUs: InputTexture::convert_to_node_texture
 * (input_texture.rs:203-219) + ResampledChild::render (resampler.rs:305-378) in one step, frame -> dst-sized
 * RGBA8 tile.  Returns the plan kind (0 = direct: dst untouched) or a negative status. */
SMR_API int smr_ingest_resample(smr_ctx *ctx, const smr_frame *in, const float crop[4], smr_surface *dst);
/* All inputs of a shard in one launch: crops = n x {top, left, width, height}; kinds[i] (optional) = plan kind of input i. */
SMR_API int smr_ingest_resample_batch(smr_ctx *ctx, const smr_frame *const *in, const float *crops, smr_surface *const *dst, uint32_t n,
                                      int *kinds);

/* ---- multi-GPU exchange step (no counterpart in the reference, which owns one wgpu device) -------
 * Inputs are sharded over GPUs (input i -> GPU i mod N); each GPU turns its inputs into dst-sized tiles with
 * smr_ingest_resample_batch (the per-input independence of render_loop.rs:24-41, layout.rs:250-275), the tiles are gathered
 * on the GPU that composes (smr_render_layouts with the tiles as SMR_SOURCE_OPAQUE_SURFACE sources).  Everything is
 * stream-ordered on the contexts' streams: no call below blocks the host.
 *   smr_comm_create_local  ONE process driving n devices (what smelter-core's renderer thread would hold,
 *                          smelter-core/src/pipeline/instance.rs:435-503): peer copies over xGMI, one link per sender.
 *   smr_comm_create_rank   one process per GPU: RCCL point-to-point.  Rank 0 makes an id with smr_comm_unique_id and hands the
 *                          128 bytes to the others by whatever means the host has.
 * smr_gather_tiles: tile i was produced on rank owner[i] in src[i]; afterwards dst[i] on `root` holds it (tiles the root owns
 * are skipped).  Local comm: one call moves everything.  Rank comm: every rank makes the same call; src[i] is read on its
 * owner only, dst[i] written on the root only (other entries may be NULL there); in rank mode a tile travels as one block of
 * pitch * h bytes, so src[i] and dst[i] must both have the canonical pitch (w * bytes per pixel rounded up to 256, what
 * smr_surface_create gives) — anything else is SMR_ERR_INVALID. */
typedef struct smr_comm smr_comm;
#define SMR_COMM_ID_BYTES 128
SMR_API int smr_comm_create_local(smr_ctx *const *ctxs, uint32_t n, smr_comm **out);
SMR_API int smr_comm_unique_id(uint8_t id[SMR_COMM_ID_BYTES]);
SMR_API int smr_comm_create_rank(smr_ctx *ctx, uint32_t world, uint32_t rank, const uint8_t id[SMR_COMM_ID_BYTES], smr_comm **out);
SMR_API void smr_comm_destroy(smr_comm *comm);
SMR_API uint32_t smr_comm_world(const smr_comm *comm);
SMR_API uint32_t smr_comm_rank(const smr_comm *comm);
SMR_API const char *smr_comm_last_error(const smr_comm *comm);
SMR_API int smr_gather_tiles(smr_comm *comm, uint32_t root, const uint32_t *owner, const smr_surface *const *src, smr_surface *const *dst,
                             uint32_t n);

/* ---- a12: text (transformations/text_renderer.rs:72-167, 236-368) -------------------
 * TextRendererNode::render's blit: clear `target` to bg (a shader colour: convert_to_shader_color of the node's background), then
 * every glyph quad's coverage from the A8 atlas in the quad's colour, premultiplied OVER. */
SMR_API int smr_blit_glyphs(smr_ctx *ctx, smr_surface *target, const float bg[4], const smr_glyph *glyphs, uint32_t n,
                            const uint8_t *atlas_host, uint32_t atlas_w, uint32_t atlas_h);

/* The host half of the text node — font database, line layout, glyph rasteriser — behind the ABI (host code, no GPU):
 *   TextRendererCtx::new / add_font, Renderer::register_font (state.rs:168-171)  -> smr_fontbook_add_file / _add_memory / _add_dir
 *   TextRendererCtx::layout_text: Buffer::set_text + set_wrap + set_size + shape_until_scroll, get_text_resolution (text_renderer.rs:282-368)
 *                                                                                -> smr_fontbook_measure (a smr_text_measure_fn: pass the book as `user`)
 *   TextRendererNode::render's prepare(): the laid-out buffer as atlas + quads   -> smr_fontbook_rasterise
 * The reference does all of this with glyphon / cosmic-text / swash (third-party, not in its tree).  This library's reader is its own:
 * TrueType `glyf` outlines (simple + composite), `cmap` formats 0 / 4 / 6 / 12 / 13, `hmtx` advances, pair kerning from the GPOS `kern`
 * feature (PairPos 1 and 2, extension lookups), explicit newlines, Wrap::None / Glyph / Word, Left / Center / Right alignment (Justified
 * is laid out as Left), a line's ascent + descent centred in its line box, exact-area coverage at the fractional pen position.  NOT
 * restated: GSUB (ligatures, contextual alternates), mark positioning, bidi, font fallback, hinting, CFF outlines — glyph shapes and
 * sub-pixel positions are this library's, not glyphon's: PARITY FOR TEXT PIXELS IS UNPINNED (no reference artefact holds them in-tree).
 * Fonts are loaded by path or from memory; the reference bundles Inter (smelter-render/fonts/ *.ttf), hosts register the same files. */
typedef struct smr_fontbook smr_fontbook;
/* What a shaper lays out: TextComponent's text, font attributes and wrap mode (text_renderer.rs:174-233) inside max_width x max_height.
 * Strings are the reference's variant names ("Normal" | "Italic" | "Oblique", "Thin" .. "Black", "None" | "Glyph" | "Word",
 * "Left" | "Right" | "Justified" | "Center"). */
typedef struct smr_text_params {
    const char *text, *font_family, *style, *weight, *wrap, *align;
    float font_size, line_height;
    float max_width, max_height;
} smr_text_params;
typedef struct smr_text_run {
    const smr_glyph *glyphs; /* owned by the font book: valid until its next smr_fontbook_rasterise / destroy */
    uint32_t n_glyphs;
    const uint8_t *atlas;    /* A8 coverage, atlas_w x atlas_h, tight rows */
    uint32_t atlas_w, atlas_h;
} smr_text_run;
SMR_API int smr_fontbook_create(smr_fontbook **out);
SMR_API void smr_fontbook_destroy(smr_fontbook *book);
SMR_API const char *smr_fontbook_last_error(const smr_fontbook *book);
SMR_API int smr_fontbook_add_file(smr_fontbook *book, const char *path);                     /* fontdb::Source::File */
SMR_API int smr_fontbook_add_memory(smr_fontbook *book, const uint8_t *data, size_t size);   /* fontdb::Source::Binary (copied) */
SMR_API int smr_fontbook_add_dir(smr_fontbook *book, const char *dir);                       /* every *.ttf below dir, by path; returns how many, < 0 if none */
SMR_API uint32_t smr_fontbook_count(const smr_fontbook *book);
/* lays params->text out at font_size with params->wrap inside max_width; widest line (pixels) and line count — the signature of
 * smr_text_measure_fn with `user` = the font book.  Refused (SMR_ERR_INVALID, message in smr_fontbook_last_error): text that is not UTF-8,
 * font_size outside (0, 100000] or not finite, a line_height that is not finite; smr_fontbook_add_* refuse a face whose tables do not
 * hold together (directory, head / hhea / maxp / hmtx / cmap / loca / glyf bounds, unitsPerEm outside 16 .. 16384).  Fonts and text
 * are untrusted input: tests/san drives this API with corrupted faces under AddressSanitizer. */
SMR_API int smr_fontbook_measure(void *book, const smr_text_params *params, float *widest_line, uint32_t *line_count);
/* the glyph run of a Text node of width x height pixels for smr_blit_glyphs / smr_renderer_set_text; `color` = straight RGBA 0..1 */
SMR_API int smr_fontbook_rasterise(smr_fontbook *book, const smr_text_params *params, uint32_t width, uint32_t height, const float color[4],
                                   smr_text_run *out);

/* ---- a13 stand-in: built-in "shader" kernels (user WGSL is out of scope) ------------ */
/* Built-in kernels for ShaderNode::render (transformations/shader/node.rs:71-89, shader/pipeline.rs:81-141).  Arbitrary user
 * WGSL needs a compiler this library does not carry; what ships are the shaders the reference keeps in its own tree,
 * restated as HIP kernels with the node's semantics: the target is cleared to transparent, one full-target plane is drawn per
 * source texture (plane_id 0..n-1; a single plane with plane_id -1 when there are none), premultiplied-alpha OVER, each plane
 * stored to the RGBA8 target before the next is blended.  BaseShaderParameters {plane_id, time, output_resolution,
 * texture_count} (shader/base_params.rs:7-12) are supplied by the library from time_s, dst and n_src.
 *   GAUSSIAN_BLUR        params = smr_gaussian_blur_params                                     (no reference counterpart)
 *   GRADIENT             integration-tests/src/render_tests/yuv_tests/gradient.wgsl
 *   RED_BORDER           render_tests/shader/red_border.wgsl
 *   CIRCLE_LAYOUT        render_tests/shader/circle_layout.wgsl; params = smr_circle_layout[n_src] (ShaderParam::to_bytes order)
 *   FADE_TO_BALL         render_tests/shader/fade_to_ball.wgsl
 *   LAYOUT_PLANES        render_tests/shader/layout_planes.wgsl
 *   COLOR_BY_TEXTURE_COUNT  render_tests/shader/color_output_with_texture_count.wgsl
 *   SILLY                integration-tests/examples/silly.wgsl */
typedef enum smr_builtin_shader_id {
    SMR_SHADER_GAUSSIAN_BLUR = 0,
    SMR_SHADER_GRADIENT = 1,
    SMR_SHADER_RED_BORDER = 2,
    SMR_SHADER_CIRCLE_LAYOUT = 3,
    SMR_SHADER_FADE_TO_BALL = 4,
    SMR_SHADER_LAYOUT_PLANES = 5,
    SMR_SHADER_COLOR_BY_TEXTURE_COUNT = 6,
    SMR_SHADER_SILLY = 7
} smr_builtin_shader_id;
#define SMR_SHADER_MAX_SOURCES 16 /* binding_array<texture_2d<f32>, 16> */
typedef struct smr_circle_layout { uint32_t left_px, top_px, width_px, height_px; float background_color[4]; } smr_circle_layout;
typedef struct smr_gaussian_blur_params { float sigma; } smr_gaussian_blur_params;
SMR_API int smr_builtin_shader(smr_ctx *ctx, uint32_t id, const void *params, size_t params_size,
                               const smr_surface *const *src, uint32_t n_src, smr_surface *dst, float time_s);

/* ---- a5/a6: host scene engine (no GPU work) ------------------------------------------
 * Scene JSON (the smelter-api component schema, smelter-api/src/video/component.rs) -> stateful
 * component tree with transitions (smelter-render/src/scene/scene_state.rs:74-127) -> render graph
 * (IntermediateNode::build_tree, scene_state.rs:148-196) -> per frame the flattened smr_layout list
 * of one layout node (LayoutProvider::layouts + NestedLayout::flatten, transformations/layout.rs:176-184,
 * layout/flatten.rs:10-22).  One smr_scene == one output of the reference's SceneState. */
typedef struct smr_scene smr_scene;

typedef enum smr_node_kind {
    SMR_NODE_INPUT_STREAM = 0, /* NodeParams::InputStream */
    SMR_NODE_LAYOUT = 1,       /* NodeParams::Layout (View / Rescaler / Tiles root) */
    SMR_NODE_TEXT = 2,         /* NodeParams::Text */
    SMR_NODE_IMAGE = 3,        /* NodeParams::Image */
    SMR_NODE_SHADER = 4        /* NodeParams::Shader */
} smr_node_kind;

typedef struct smr_scene_node {
    uint32_t kind;        /* smr_node_kind */
    int32_t parent;       /* -1 for the root */
    uint32_t n_children;
    uint32_t width, height; /* intrinsic size (0,0 for an input stream that has not rendered yet) */
    const char *ref_id;   /* input_id / image_id / shader_id, "" for layouts and text; valid until the next update */
    const char *id;       /* component id or "" */
    const char *payload;  /* Text: the string; otherwise "" */
} smr_scene_node;

#define SMR_NO_RESOLUTION 0xffffffffu /* child_wh width: the child has no texture (node_texture.rs state() == None) */

SMR_API int smr_scene_create(smr_scene **out);
SMR_API void smr_scene_destroy(smr_scene *scene);
SMR_API const char *smr_scene_last_error(const smr_scene *scene);
/* Fitted text (TextDimensions::Fitted / FittedColumn, text_renderer.rs:282-346): the renderer sizes a Text node without
 * "width"/"height" by the reference's rule (get_text_resolution, text_renderer.rs:348-368) —
 *     width  = max over the laid-out lines of ceil(line width)             (FittedColumn: the given width)
 *     height = trunc(lines * ceil(line_height) + font_size / 5)
 * — from the line metrics the caller's shaper reports through this callback (glyphon / cosmic-text in the reference, third-party):
 * lay `text` out at `font_size` with wrapping `wrap` ("None" | "Glyph" | "Word") inside max_width x max_height and return the
 * widest line and the number of lines.  Return non-zero to fail the scene update.  Without a measurer such nodes are refused. */
typedef int (*smr_text_measure_fn)(void *user, const smr_text_params *params, float *widest_line, uint32_t *line_count);
SMR_API int smr_scene_set_text_measurer(smr_scene *scene, smr_text_measure_fn fn, void *user);
/* Renderer::register_renderer(Image) as far as sizing goes (scene/image_component.rs) */
SMR_API int smr_scene_register_image(smr_scene *scene, const char *image_id, uint32_t width, uint32_t height);
/* Renderer::update_scene (state.rs:177-189).  On error the previous scene stays active. */
SMR_API int smr_scene_update(smr_scene *scene, const char *scene_json, uint32_t out_width, uint32_t out_height);
/* The API-to-scene conversion alone: `impl TryFrom<Component> for scene::Component`
 * (smelter-api/src/video/component_into.rs:8-432) with its defaults, validation and error strings.  *out_json is the
 * converted component tree as canonical JSON (owned by the scene, valid until the next call); the active scene is untouched.
 * Pinned by smelter-api/tests/scene_deserialization.rs (tests/golden/scene_api_vectors.json). */
SMR_API int smr_scene_parse(smr_scene *scene, const char *scene_json, const char **out_json);
SMR_API int smr_scene_node_count(const smr_scene *scene); /* nodes in pre-order, 0 is the root */
SMR_API int smr_scene_node_info(const smr_scene *scene, int node, smr_scene_node *out);
SMR_API int smr_scene_node_children(const smr_scene *scene, int node, int32_t *out, uint32_t cap);
/* LayoutNode::render up to the flattened list.  child_wh = {w0,h0,w1,h1,...} one pair per child node in order
 * (w == SMR_NO_RESOLUTION: no texture).  mode selects convert_to_shader_color's sRGB handling (wgpu/utils.rs:51-72).
 * Writes min(*n_out, cap) layouts; also advances the scene clock used by the next smr_scene_update
 * (SceneState::register_render_event, scene_state.rs:59-67). */
SMR_API int smr_scene_node_layouts(smr_scene *scene, int node, int64_t pts_ns, const uint32_t *child_wh, uint32_t n_children,
                                   uint32_t mode, smr_layout *out, uint32_t cap, uint32_t *n_out, uint32_t *out_width,
                                   uint32_t *out_height);
/* scene/transition/{cubic_bezier,bounce}.rs and smelter-api/src/video/color.rs, exported for the parity tests */
SMR_API double smr_cubic_bezier_easing(double progress, double x1, double y1, double x2, double y2);
SMR_API double smr_bounce_easing(double progress);
SMR_API int smr_parse_color(const char *text, uint8_t rgba[4]);

/* ---- a14 (+ a2, a4): the renderer ---------------------------------------------------
 * `Renderer` of smelter-render/src/state.rs:96-252 over the scene engine and the kernels above:
 *   Renderer::new              -> smr_renderer_create (stream_fallback_timeout: state.rs:43-52, render_loop.rs:29)
 *   register_input / unregister_input / unregister_output (state.rs:102-121)
 *   register_renderer(Image)   -> smr_renderer_register_image (bitmap pixels; decoding is the caller's)
 *   register_renderer(Shader)  -> smr_renderer_register_shader (a built-in kernel id; user WGSL is out of scope)
 *   update_scene               -> smr_renderer_update_scene(output_id, resolution, OutputFrameFormat, scene JSON) (state.rs:177-189)
 *   render(FrameSet<InputId>)  -> smr_renderer_render: populate_inputs (stale frames dropped), depth-first walk of every output's
 *                                 render graph (input refs, images, text, shader and nested layout nodes), read_outputs fused
 *                                 into the root layout node (state.rs:220-252, render_loop.rs:19-230)
 * Text nodes: with a font book (smr_renderer_set_fontbook) the renderer lays out, rasterises and draws them itself at every update_scene;
 * without one the caller supplies each Text node's glyph run once after every update_scene (smr_renderer_node_info lists the nodes,
 * smr_renderer_set_text takes the run) — text_renderer.rs renders once per update either way.
 * Output frames live in HBM, two per output, alternating: a returned frame stays valid until the render after the next. */
typedef struct smr_renderer smr_renderer;
typedef struct smr_input_frame {
    const char *input_id;
    const smr_frame *frame; /* resident in HBM (smr_frame_upload or wrapped decoder output) */
    int64_t pts_ns;
} smr_input_frame;
typedef struct smr_output_frame {
    const char *output_id;
    const smr_frame *frame; /* owned by the renderer; valid until the lane that rendered it has rendered two more frames */
    smr_ctx *ctx;           /* the context (stream) the frame's GPU work was issued on: sync / download through this one */
} smr_output_frame;

SMR_API int smr_renderer_create(smr_ctx *ctx, int64_t stream_fallback_timeout_ns /* < 0: 500 ms */, smr_renderer **out);
SMR_API void smr_renderer_destroy(smr_renderer *r);
SMR_API const char *smr_renderer_last_error(const smr_renderer *r);
SMR_API int smr_renderer_register_input(smr_renderer *r, const char *input_id);
SMR_API int smr_renderer_unregister_input(smr_renderer *r, const char *input_id);
SMR_API int smr_renderer_register_image(smr_renderer *r, const char *image_id, const uint8_t *rgba_straight, uint32_t width, uint32_t height);
SMR_API int smr_renderer_register_shader(smr_renderer *r, const char *shader_id, uint32_t builtin_id);
/* An update is refused as a whole — the previous scene stays active — for everything the scene definition can get wrong: validation errors,
 * unknown shader ids, and (with a font book) Text nodes no run can be made of (empty font book, text that is not UTF-8, absurd sizes).  One
 * failure is reported AFTER the new scene is in place: a device allocation that fails while the output's frames or a Text node's surface are
 * created (SMR_ERR_OOM): the new scene is active, the nodes that could not be drawn render transparent. */
SMR_API int smr_renderer_update_scene(smr_renderer *r, const char *output_id, uint32_t width, uint32_t height, uint32_t output_format,
                                      const char *scene_json);
SMR_API int smr_renderer_unregister_output(smr_renderer *r, const char *output_id);
SMR_API int smr_renderer_node_count(const smr_renderer *r, const char *output_id);
SMR_API int smr_renderer_node_info(smr_renderer *r, const char *output_id, int node, smr_scene_node *out);
SMR_API int smr_renderer_set_text_measurer(smr_renderer *r, smr_text_measure_fn fn, void *user);
/* TextRendererCtx for this renderer: with a font book (not owned; it must outlive the renderer) fitted Text nodes are measured with it
 * and EVERY Text node is laid out, rasterised and drawn by smr_renderer_update_scene itself — once per update, as
 * TextRendererNode::render does — in the node's colour over its background_color.  smr_renderer_set_text still replaces a node's run.
 * NULL detaches the book (and its measurer). */
SMR_API int smr_renderer_set_fontbook(smr_renderer *r, smr_fontbook *book);
SMR_API int smr_renderer_set_text(smr_renderer *r, const char *output_id, int node, const float bg[4], const smr_glyph *glyphs, uint32_t n,
                                  const uint8_t *atlas_host, uint32_t atlas_w, uint32_t atlas_h);
SMR_API int smr_renderer_render(smr_renderer *r, int64_t pts_ns, const smr_input_frame *inputs, uint32_t n_inputs,
                                smr_output_frame *outputs, uint32_t cap, uint32_t *n_outputs);
/* Frames in flight.  The reference submits one frame at a time (one wgpu queue); here a renderer may own several lanes — extra
 * contexts on the same device (own HIP stream, own scratch).  Consecutive smr_renderer_render calls rotate through the lanes, so
 * the latency-bound tail of one frame overlaps the next frame's kernels, while the scene stays ONE state: one update_scene call,
 * one pts sequence, transitions identical to the single-lane renderer.  Each lane keeps its own pair of output frames and node
 * surfaces; smr_output_frame.ctx says which context produced a frame.  smr_renderer_sync waits for every lane. */
SMR_API int smr_renderer_add_lane(smr_renderer *r, smr_ctx *ctx);
SMR_API int smr_renderer_sync(smr_renderer *r);

/* ABI version: a host checks smr_abi_version() == SMR_ABI_VERSION when it loads the library.
 *   1  rounds 1 - 4.
 *   2  SMR_INGEST_MFMA_F16_FUSED (ingest implementation 5, the fused one-code-per-stage conversion) left the product enum: a product build
 *      rejects 5, a laboratory build (smr_build_flags() & 1) keeps it under an internal name.  Kernel counter slot 2, once
 *      SMR_KERNEL_INGEST_MFMA_WG, counts SMR_KERNEL_FRAME_TO_RGBA_420.  smr_text_params moved in front of the font-book entry points.  The
 *      SMR_CONVERT_GENERAL / SMR_DISABLE_FUSED / SMR_COMPOSE_SELECT environment knobs are smr_ctx_set_option options in a product build
 *      (the environment is read by laboratory builds only).
 * The two removed names are kept as macros that do not compile, so that a source written against version 1 fails where it uses them
 * instead of silently meaning something else. */
#define SMR_ABI_VERSION 2
#define SMR_INGEST_MFMA_F16_FUSED SMR_REMOVED_IN_ABI_2__the_fused_conversion_is_a_laboratory_route__use_SMR_INGEST_AUTO
#define SMR_KERNEL_INGEST_MFMA_WG SMR_REMOVED_IN_ABI_2__slot_2_counts_SMR_KERNEL_FRAME_TO_RGBA_420
SMR_API uint32_t smr_abi_version(void);
/* bit 0: a laboratory build (-DSMR_LAB: environment knobs, the fused-conversion route, profiling hooks); 0 for a product build */
SMR_API uint32_t smr_build_flags(void);
SMR_API uint32_t smr_sizeof_layout(void);

#ifdef __cplusplus
}
#endif
#endif /* SMR_H */
