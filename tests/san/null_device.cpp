// TEST INFRASTRUCTURE — not part of the product, never linked into libsmr_hip.so.  A stand-in for the GPU half of the C ABI (the entry
// points smelter_amd/csrc/host/renderer.cpp calls), so that the renderer — registry, render-graph walk, lanes, per-node surfaces, text
// nodes, error paths — runs on a machine without a GPU under AddressSanitizer + UBSan (tests/san/renderer_fuzz.cpp).
// It renders nothing.  Every entry point checks its arguments the way the library does (same status codes), READS every byte it is
// handed (layout lists, glyph runs, atlases, parameter blocks: a short buffer is an ASan report) and dereferences every surface and
// frame (a destroyed one is a use-after-free report).  Surfaces remember the context that made them; `null_device_live_surfaces()`
// says how many were never destroyed.  `null_device_fail_after(n)` makes the n-th allocation from now fail with SMR_ERR_OOM: the
// renderer's error paths run too.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "smr.h"

struct smr_ctx {
    uint32_t mode = 0;
    std::string err;
    int options[16] = {0};
    long calls = 0;
};
struct smr_surface {
    uint32_t w = 0, h = 0, format = 0;
    size_t pitch = 0;
    std::vector<uint8_t> mem;
    smr_ctx *maker = nullptr;
    uint64_t magic = 0x5355524641434521ull;
};

static long g_live_surfaces = 0, g_fail_countdown = -1;
static volatile uint64_t g_sink;

extern "C" long null_device_live_surfaces() { return g_live_surfaces; }
extern "C" void null_device_fail_after(long n) { g_fail_countdown = n; }

static int fail(smr_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}
static void touch(const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    uint64_t s = 0;
    for (size_t i = 0; i < n; i++) s += b[i];
    g_sink = s;
}
static bool alive(const smr_surface *s) { return s && s->magic == 0x5355524641434521ull; }  // (reading a freed surface is the sanitizer's report)
static size_t bpp(uint32_t format) { return format == SMR_PX_RGBA8 ? 4 : format == SMR_PX_RGBA16F ? 8 : format == SMR_PX_RG8 ? 2 : 1; }

static bool frame_ok(const smr_frame *f) {
    if (!f || !f->width || !f->height || f->format > SMR_FRAME_RGBA) return false;
    const int planes = f->format <= SMR_FRAME_PLANAR_YUVJ420 ? 3 : f->format == SMR_FRAME_NV12 ? 2 : 1;
    for (int p = 0; p < planes; p++)
        if (!alive(f->planes[p])) return false;
    const smr_surface *y = f->planes[0];
    if (f->format == SMR_FRAME_UYVY422 || f->format == SMR_FRAME_YUYV422) return y->w == (f->width + 1) / 2 && y->h == f->height;
    return y->w == f->width && y->h == f->height;
}

extern "C" {

int smr_ctx_create(int hip_device, uint32_t mode, uint32_t max_layouts, void *hip_stream, smr_ctx **out) {
    (void)hip_device; (void)max_layouts; (void)hip_stream;
    if (!out || mode > SMR_MODE_CPU_OPTIMIZED) return SMR_ERR_INVALID;
    *out = new smr_ctx();
    (*out)->mode = mode;
    return SMR_OK;
}
void smr_ctx_destroy(smr_ctx *ctx) { delete ctx; }
const char *smr_last_error(const smr_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
uint32_t smr_ctx_mode(const smr_ctx *ctx) { return ctx ? ctx->mode : 0; }
int smr_sync(smr_ctx *ctx) {
    if (!ctx) return SMR_ERR_INVALID;
    ctx->calls++;
    return SMR_OK;
}
int smr_ctx_set_option(smr_ctx *ctx, uint32_t option, int32_t value) {
    if (!ctx || option >= 16) return SMR_ERR_INVALID;
    ctx->options[option] = value;
    return SMR_OK;
}

int smr_surface_create(smr_ctx *ctx, uint32_t w, uint32_t h, uint32_t format, smr_surface **out) {
    if (!ctx || !out || !w || !h || format > SMR_PX_RG8) return fail(ctx, SMR_ERR_INVALID, "smr_surface_create: invalid argument");
    if (w > 16384 || h > 16384) return fail(ctx, SMR_ERR_INVALID, "smr_surface_create: larger than 16384 x 16384");
    if (g_fail_countdown >= 0 && g_fail_countdown-- == 0) return fail(ctx, SMR_ERR_OOM, "null device: injected allocation failure");
    smr_surface *s = new smr_surface();
    s->w = w; s->h = h; s->format = format; s->maker = ctx;
    s->pitch = ((size_t)w * bpp(format) + 255) & ~(size_t)255;
    s->mem.assign(64, 0);  // (a token allocation: nothing is rendered)
    g_live_surfaces++;
    *out = s;
    return SMR_OK;
}
void smr_surface_destroy(smr_ctx *ctx, smr_surface *s) {
    (void)ctx;
    if (!s) return;
    if (!alive(s)) { fprintf(stderr, "null device: smr_surface_destroy of something that is not a live surface\n"); abort(); }
    s->magic = 0;
    g_live_surfaces--;
    delete s;
}
int smr_surface_info_get(const smr_surface *s, smr_surface_info *out) {
    if (!alive(s) || !out) return SMR_ERR_INVALID;
    out->width = s->w; out->height = s->h; out->format = s->format; out->owned = 1; out->pitch = s->pitch; out->dptr = (void *)s->mem.data();
    return SMR_OK;
}
int smr_surface_upload(smr_ctx *ctx, smr_surface *s, const void *host, size_t host_pitch) {
    if (!ctx || !alive(s) || !host) return fail(ctx, SMR_ERR_INVALID, "smr_surface_upload: invalid argument");
    const size_t row = (size_t)s->w * bpp(s->format), pitch = host_pitch ? host_pitch : row;
    if (pitch < row) return fail(ctx, SMR_ERR_INVALID, "smr_surface_upload: host pitch below the row size");
    for (uint32_t y = 0; y < s->h; y++) touch((const uint8_t *)host + (size_t)y * pitch, row);
    return SMR_OK;
}

int smr_frame_create(smr_ctx *ctx, uint32_t format, uint32_t w, uint32_t h, smr_frame *out) {
    if (!ctx || !out || !w || !h || format > SMR_FRAME_RGBA) return fail(ctx, SMR_ERR_INVALID, "smr_frame_create: invalid argument");
    memset(out, 0, sizeof(*out));
    out->format = format; out->width = w; out->height = h;
    uint32_t pw[3] = {w, 0, 0}, ph[3] = {h, 0, 0}, pf[3] = {SMR_PX_R8, SMR_PX_R8, SMR_PX_R8};
    int planes = 1;
    switch (format) {
    case SMR_FRAME_PLANAR_YUV420: case SMR_FRAME_PLANAR_YUVJ420: planes = 3; pw[1] = pw[2] = (w + 1) / 2; ph[1] = ph[2] = (h + 1) / 2; break;
    case SMR_FRAME_PLANAR_YUV422: planes = 3; pw[1] = pw[2] = (w + 1) / 2; ph[1] = ph[2] = h; break;
    case SMR_FRAME_PLANAR_YUV444: planes = 3; pw[1] = pw[2] = w; ph[1] = ph[2] = h; break;
    case SMR_FRAME_NV12: planes = 2; pw[1] = (w + 1) / 2; ph[1] = (h + 1) / 2; pf[1] = SMR_PX_RG8; break;
    case SMR_FRAME_UYVY422: case SMR_FRAME_YUYV422: pw[0] = (w + 1) / 2; pf[0] = SMR_PX_RGBA8; break;
    default: pf[0] = SMR_PX_RGBA8; break;
    }
    for (int p = 0; p < planes; p++) {
        const int rc = smr_surface_create(ctx, pw[p], ph[p], pf[p], &out->planes[p]);
        if (rc < 0) {
            for (int q = 0; q < p; q++) { smr_surface_destroy(ctx, out->planes[q]); out->planes[q] = nullptr; }
            return rc;
        }
    }
    return SMR_OK;
}
void smr_frame_destroy(smr_ctx *ctx, smr_frame *f) {
    if (!f) return;
    for (int p = 0; p < 3; p++) {
        if (f->planes[p]) smr_surface_destroy(ctx, f->planes[p]);
        f->planes[p] = nullptr;
    }
}

int smr_frame_to_rgba(smr_ctx *ctx, const smr_frame *in, smr_surface *node) {
    if (!ctx || !frame_ok(in) || !alive(node)) return fail(ctx, SMR_ERR_INVALID, "smr_frame_to_rgba: invalid argument");
    if (node->format != SMR_PX_RGBA8 || node->w != in->width || node->h != in->height) return fail(ctx, SMR_ERR_INVALID, "smr_frame_to_rgba: the node's size is not the frame's");
    return SMR_OK;
}
int smr_add_premultiplied_alpha(smr_ctx *ctx, const smr_surface *src, smr_surface *dst) {
    if (!ctx || !alive(src) || !alive(dst) || src->w != dst->w || src->h != dst->h) return fail(ctx, SMR_ERR_INVALID, "smr_add_premultiplied_alpha: invalid argument");
    return SMR_OK;
}
int smr_rgba_to_frame(smr_ctx *ctx, const smr_surface *node, const smr_frame *out) {
    if (!ctx || !alive(node) || !frame_ok(out)) return fail(ctx, SMR_ERR_INVALID, "smr_rgba_to_frame: invalid argument");
    if (node->format != SMR_PX_RGBA8 || node->w != out->width || node->h != out->height) return fail(ctx, SMR_ERR_INVALID, "smr_rgba_to_frame: the node's size is not the frame's");
    return SMR_OK;
}
int smr_frame_fill_black(smr_ctx *ctx, const smr_frame *out) {
    if (!ctx || !frame_ok(out)) return fail(ctx, SMR_ERR_INVALID, "smr_frame_fill_black: invalid argument");
    return SMR_OK;
}
int smr_rescale_bilinear(smr_ctx *ctx, const smr_surface *src, smr_surface *dst) {
    if (!ctx || !alive(src) || !alive(dst) || src == dst) return fail(ctx, SMR_ERR_INVALID, "smr_rescale_bilinear: invalid argument");
    if (src->format != SMR_PX_RGBA8 || dst->format != SMR_PX_RGBA8) return fail(ctx, SMR_ERR_INVALID, "smr_rescale_bilinear: RGBA8 surfaces only");
    return SMR_OK;
}

int smr_render_layouts(smr_ctx *ctx, const smr_layout *layouts, uint32_t n, const smr_source *sources, uint32_t n_sources, uint32_t out_w,
                       uint32_t out_h, const smr_frame *out, smr_surface *out_rgba) {
    if (!ctx || (n && !layouts) || (n_sources && !sources) || !out_w || !out_h || (!out == !out_rgba))
        return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: invalid argument");
    touch(layouts, (size_t)n * sizeof(smr_layout));
    touch(sources, (size_t)n_sources * sizeof(smr_source));
    for (uint32_t i = 0; i < n; i++) {
        const smr_layout &L = layouts[i];
        if (L.type > 2 || L.masks_len > SMR_MAX_MASKS) return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: malformed layout");
        if (L.type == 0 && L.source_index != SMR_NO_SOURCE && L.source_index >= n_sources) return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: source index out of range");
    }
    for (uint32_t k = 0; k < n_sources; k++) {
        const smr_source &S = sources[k];
        if (S.kind == SMR_SOURCE_NONE) continue;
        if (S.kind == SMR_SOURCE_FRAME) { if (!frame_ok(S.frame)) return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: malformed source frame"); }
        else if (S.kind == SMR_SOURCE_SURFACE || S.kind == SMR_SOURCE_OPAQUE_SURFACE) {
            if (!alive(S.surface) || S.surface->format != SMR_PX_RGBA8) return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: a source surface must be RGBA8");
            if (S.surface == out_rgba) return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: the target is one of its own sources");
        } else return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: unknown source kind");
    }
    if (out) {
        if (!frame_ok(out) || out->width != out_w || out->height != out_h) return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: the output frame's size is not out_w x out_h");
        if (out->format != SMR_FRAME_PLANAR_YUV420 && out->format != SMR_FRAME_PLANAR_YUV422 && out->format != SMR_FRAME_PLANAR_YUV444 && out->format != SMR_FRAME_NV12)
            return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: output format");
    } else if (!alive(out_rgba) || out_rgba->format != SMR_PX_RGBA8 || out_rgba->w != out_w || out_rgba->h != out_h) {
        return fail(ctx, SMR_ERR_INVALID, "smr_render_layouts: the RGBA target's size is not out_w x out_h");
    }
    ctx->calls++;
    return SMR_OK;
}

int smr_blit_glyphs(smr_ctx *ctx, smr_surface *target, const float bg[4], const smr_glyph *glyphs, uint32_t n, const uint8_t *atlas_host, uint32_t atlas_w,
                    uint32_t atlas_h) {
    if (!ctx || !alive(target) || !bg || (n && (!glyphs || !atlas_host))) return fail(ctx, SMR_ERR_INVALID, "smr_blit_glyphs: invalid argument");
    touch(bg, 16);
    touch(glyphs, (size_t)n * sizeof(smr_glyph));
    if (n) touch(atlas_host, (size_t)atlas_w * atlas_h);
    for (uint32_t i = 0; i < n; i++) {
        const smr_glyph &g = glyphs[i];
        if (g.w <= 0 || g.h <= 0 || g.dst_x < 0 || g.dst_y < 0 || (int64_t)g.dst_x + g.w > (int64_t)target->w || (int64_t)g.dst_y + g.h > (int64_t)target->h)
            return fail(ctx, SMR_ERR_INVALID, "smr_blit_glyphs: a quad leaves the target");
        if (g.atlas_x < 0 || g.atlas_y < 0 || (int64_t)g.atlas_x + g.w > (int64_t)atlas_w || (int64_t)g.atlas_y + g.h > (int64_t)atlas_h)
            return fail(ctx, SMR_ERR_INVALID, "smr_blit_glyphs: a quad leaves the atlas");
    }
    return SMR_OK;
}

int smr_builtin_shader(smr_ctx *ctx, uint32_t id, const void *params, size_t params_size, const smr_surface *const *src, uint32_t n_src, smr_surface *dst,
                       float time_s) {
    (void)time_s;
    if (!ctx || id > SMR_SHADER_SILLY || !alive(dst) || (n_src && !src) || (params_size && !params) || n_src > SMR_SHADER_MAX_SOURCES)
        return fail(ctx, SMR_ERR_INVALID, "smr_builtin_shader: invalid argument");
    touch(params, params_size);
    for (uint32_t i = 0; i < n_src; i++)
        if (!alive(src[i]) || src[i] == dst) return fail(ctx, SMR_ERR_INVALID, "smr_builtin_shader: malformed source");
    if (id == SMR_SHADER_GAUSSIAN_BLUR && (n_src < 1 || params_size < sizeof(smr_gaussian_blur_params) || src[0]->w != dst->w || src[0]->h != dst->h))
        return fail(ctx, SMR_ERR_INVALID, "smr_builtin_shader: gaussian blur wants one source of the target's size and a sigma");
    return SMR_OK;
}

}  // extern "C"
