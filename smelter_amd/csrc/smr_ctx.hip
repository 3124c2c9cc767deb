// smr_ctx.hip — context, pitched surfaces, frames, transfers, timers.
//
// Replaces: WgpuCtx (smelter-render/src/wgpu/ctx.rs:34-107), wgpu texture wrappers
// (wgpu/texture/*.rs), queue.write_texture (texture/base.rs:61-77) and the padded-row
// read-back (texture/base.rs:97-118, state/output_texture.rs:85-113).
#include "smr_internal.h"
#include "smr_tables.h"

#include <cmath>
#include <cstdlib>

int smr_fail(smr_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

int smr_check_hip(smr_ctx *ctx, hipError_t e, const char *what) {
    if (e == hipSuccess) return SMR_OK;
    int code = (e == hipErrorOutOfMemory) ? SMR_ERR_OOM
               : (e == hipErrorInvalidValue ? SMR_ERR_INVALID : SMR_ERR_INTERNAL);
    (void)hipGetLastError();
    return smr_fail(ctx, code, "%s: %s", what, hipGetErrorString(e));
}

void *smr_scratch(smr_ctx *ctx, int slot, size_t bytes) {
    if ((size_t)slot >= ctx->scratch.size()) ctx->scratch.resize(slot + 1);
    auto &s = ctx->scratch[slot];
    if (s.bytes >= bytes && s.ptr) return s.ptr;
    if (s.ptr) {
        // the old buffer may still be in use by enqueued kernels
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(s.ptr);
        s.ptr = nullptr;
        s.bytes = 0;
    }
    size_t want = (bytes + 4095) & ~(size_t)4095;
    hipError_t e = hipMalloc(&s.ptr, want);
    if (e != hipSuccess) {
        smr_check_hip(ctx, e, "hipMalloc(scratch)");
        s.ptr = nullptr;
        return nullptr;
    }
    s.bytes = want;
    return s.ptr;
}

static int surface_create_with(smr_ctx *ctx, u32 w, u32 h, u32 format, size_t headroom, smr_surface **out);

smr_surface *smr_cached_surface(smr_ctx *ctx, size_t slot, u32 w, u32 h, u32 fmt) {
    if (slot >= ctx->surf_cache.size()) ctx->surf_cache.resize(slot + 1, nullptr);
    smr_surface *&s = ctx->surf_cache[slot];
    if (s && s->w == w && s->h == h && s->fmt == fmt) return s;
    // NodeTexture::ensure_size re-creates the texture on every size change; an animated layout changes the size of its
    // resample target every frame, and hipFree / hipMalloc synchronise the device — so a scratch surface is re-described in
    // place while the new size fits its allocation (work on the stream is ordered, the previous frame is done with it by then)
    const u32 bpp = bytes_per_px(fmt);
    const size_t pitch = ((size_t)w * bpp + 255) & ~(size_t)255;
    if (s && bpp && w && h && w <= 7682 * 2 && h <= 4320 * 2 && pitch * h + SMR_SURFACE_TAIL <= s->capacity) {
        s->w = w; s->h = h; s->fmt = fmt; s->pitch = pitch;
        return s;
    }
    // a slot that outgrew its allocation once takes 25 % headroom, so a layout that keeps growing through a transition does
    // not reallocate on every frame
    const size_t headroom = s ? (pitch * h) / 4 : 0;
    if (s) {
        smr_surface_destroy(ctx, s);
        s = nullptr;
    }
    if (surface_create_with(ctx, w, h, fmt, headroom, &s) != SMR_OK) return nullptr;
    return s;
}

static hipEvent_t take_event(smr_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

StageScope::StageScope(smr_ctx *c, int s) : ctx(c), stage(s) {
    if (!ctx->profiling) return;
    a = take_event(ctx);
    b = take_event(ctx);
    if (a) (void)hipEventRecord(a, ctx->stream);
}

static void drain_profile(smr_ctx *ctx);

StageScope::~StageScope() {
    if (!ctx->profiling || !a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->pending.push_back({a, b, stage});
    // the runtime's pool of timestamp signals is finite: a few thousand recorded-but-unread events stall hipEventRecord for good
    // (seen with ~2 400 outstanding), so the pairs are read back and recycled in batches
    if (ctx->pending.size() >= 256) drain_profile(ctx);
}

static void drain_profile(smr_ctx *ctx) {
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->stage_ms[p.stage] += ms;
            ctx->stage_launches[p.stage] += 1;
        }
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->pending.clear();
}

extern "C" {

uint32_t smr_abi_version(void) { return SMR_ABI_VERSION; }
uint32_t smr_build_flags(void) { return SMR_LAB_BUILD ? 1u : 0u; }
uint32_t smr_ctx_mode(const smr_ctx *ctx) { return ctx ? ctx->mode : 0u; }
uint32_t smr_sizeof_layout(void) { return (uint32_t)sizeof(smr_layout); }

int smr_ctx_create(int hip_device, uint32_t mode, uint32_t max_layouts, void *hip_stream, smr_ctx **out) {
    if (!out) return SMR_ERR_INVALID;
    *out = nullptr;
    if (mode > SMR_MODE_CPU_OPTIMIZED) return SMR_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || hip_device < 0 || hip_device >= n) return SMR_ERR_INVALID;
    if (hipSetDevice(hip_device) != hipSuccess) return SMR_ERR_INTERNAL;
    smr_ctx *ctx = new smr_ctx();
    ctx->device = hip_device;
    ctx->mode = mode;
    ctx->max_layouts = max_layouts ? max_layouts : SMR_DEFAULT_MAX_LAYOUTS;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, hip_device) == hipSuccess && cus > 0) ctx->cu_count = cus;
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return SMR_ERR_INTERNAL;
        }
        ctx->own_stream = true;
    }
    // (local buffers: two threads creating contexts at once must not share a half-built table)
    float tables[SMR_TABLE_FLOATS];
    u32 lut16[SMR_LUT16_WORDS];  // decode table as an f16 pair per entry (hi, lo = t - hi): the A operand of the matrix-core resamplers
    if (!smr_build_tables(tables, lut16)) {
        smr_ctx_destroy(ctx);
        return SMR_ERR_INTERNAL;
    }
    memcpy(ctx->h_tables, tables, sizeof(tables));
#ifdef SMR_LAB  // A/B knobs of laboratory builds (tools/variant.sh -DSMR_LAB): a product build reads nothing from the environment
    if (const char *e = getenv("SMR_INGEST_IMPL")) {  // tools / A-B runs: "valu" or "mfma"; smr_ctx_set_ingest_impl overrides
        if (!strcmp(e, "valu")) ctx->ingest_impl = SMR_INGEST_VALU_F32;
        else if (!strcmp(e, "mfma")) ctx->ingest_impl = SMR_INGEST_MFMA_F16;
        else if (!strcmp(e, "mfma_node")) ctx->ingest_impl = SMR_INGEST_MFMA_F16_NODE;
        else if (!strcmp(e, "fused")) ctx->ingest_impl = SMR_INGEST_LAB_FUSED;
    }
    if (const char *e = getenv("SMR_CONVERT_GENERAL")) ctx->convert_impl = (e[0] && e[0] != '0') ? SMR_CONVERT_GENERAL : SMR_CONVERT_AUTO;  // (read once: tools)
    if (const char *e = getenv("SMR_CONVERT_LDS_PAD")) ctx->convert_lds_pad = (u32)atoi(e);
    if (const char *e = getenv("SMR_COMPACT_NODES")) ctx->compact_nodes = atoi(e) != 0;  // (tools / A-B)
    if (const char *e = getenv("SMR_PLANE_SOURCE")) ctx->plane_source = atoi(e) != 0;
    if (const char *e = getenv("SMR_INGEST_TW")) ctx->force_tw = atoi(e);  // (tools; smr_ctx_set_option overrides)
    if (const char *e = getenv("SMR_WAVE_NODE82")) ctx->wave_node82 = atoi(e) != 0;
    if (const char *e = getenv("SMR_RGB12_CLS82")) ctx->rgb12_cls82 = atoi(e) != 0;
    if (const char *e = getenv("SMR_CONVERT_WG_PER_CU")) ctx->convert_wg_per_cu = atoi(e);
    if (const char *e = getenv("SMR_INGEST_RESERVE_CUS")) ctx->ingest_reserve_cus = atoi(e);
    if (const char *e = getenv("SMR_INGEST_WG_PER_CU")) ctx->ingest_wg_per_cu = atoi(e);
    if (const char *e = getenv("SMR_NO_PACK_REUSE")) ctx->no_pack_reuse = atoi(e) != 0;
    if (const char *e = getenv("SMR_INGEST_MIN_ROWS")) ctx->ingest_min_rows = atoi(e);
    ctx->debug_ingest = getenv("SMR_DEBUG_INGEST") != nullptr;
    if (const char *e = getenv("SMR_COMPOSE_SELECT")) ctx->compose_select = atoi(e) != 0;
#endif
    if (hipMalloc((void **)&ctx->d_tables, sizeof(tables)) != hipSuccess ||
        hipMemcpy(ctx->d_tables, tables, sizeof(tables), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void **)&ctx->d_lut16, sizeof(lut16)) != hipSuccess ||
        hipMemcpy(ctx->d_lut16, lut16, sizeof(lut16), hipMemcpyHostToDevice) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) {
        smr_ctx_destroy(ctx);
        return SMR_ERR_INTERNAL;
    }
    *out = ctx;
    return SMR_OK;
}

void smr_ctx_destroy(smr_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    drain_profile(ctx);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto &s : ctx->scratch)
        if (s.ptr) (void)hipFree(s.ptr);
    for (auto *s : ctx->surf_cache)
        if (s) smr_surface_destroy(ctx, s);
    for (auto &t : ctx->weight_tables)
        if (t.dev) (void)hipFree(t.dev);
    for (auto &t : ctx->mfma_tables)
        if (t.dev) (void)hipFree(t.dev);
    if (ctx->d_lut16) (void)hipFree(ctx->d_lut16);
    for (auto &l : ctx->layout_ring) {
        if (l.host) (void)hipHostFree(l.host);
        if (l.dev) (void)hipFree(l.dev);
        if (l.done) (void)hipEventDestroy(l.done);
    }
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    for (TileClassMap &m : ctx->class_maps) {
        if (m.d_class) (void)hipFree(m.d_class);
        if (m.d_direct) (void)hipFree(m.d_direct);
        if (m.d_list) (void)hipFree(m.d_list);
        if (m.h_count) (void)hipHostFree(m.h_count);
        if (m.count_ev) (void)hipEventDestroy(m.count_ev);
    }
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int smr_ctx_set_option(smr_ctx *ctx, uint32_t option, int32_t value) {
    if (!ctx) return SMR_ERR_INVALID;
    switch (option) {
    case SMR_OPT_INGEST_IMPL:
        if (value < 0 || value > SMR_INGEST_LAB_FUSED || value == 3) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ctx_set_option: unknown ingest implementation %d", value);
        if (value == SMR_INGEST_LAB_FUSED && !SMR_LAB_BUILD)
            return smr_fail(ctx, SMR_ERR_INVALID, "smr_ctx_set_option: ingest implementation 5 (fused conversion) exists in laboratory builds only (-DSMR_LAB)");
        ctx->ingest_impl = (u32)value;
        return SMR_OK;
    case SMR_OPT_CONVERT_IMPL:
        if (value < 0 || value > SMR_CONVERT_BLOCK_4X2) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ctx_set_option: unknown converter implementation %d", value);
        ctx->convert_impl = (u32)value;
        return SMR_OK;
    case SMR_OPT_COMPACT_NODES:
        ctx->compact_nodes = value != 0;
        return SMR_OK;
    case SMR_OPT_PLANE_SOURCE:
        ctx->plane_source = value != 0;
        return SMR_OK;
    case SMR_OPT_FUSED_KERNELS:
        ctx->fused_disabled = value != 0 ? 0 : 1;
        return SMR_OK;
    case SMR_OPT_COMPOSE_SELECT:
        ctx->compose_select = value != 0;
        return SMR_OK;
    case SMR_OPT_SHARED_DEVICE:
        ctx->shared_device = value != 0;
        return SMR_OK;
    case SMR_OPT_DIRECT_OUTPUT:
        ctx->direct_output = value != 0;
        return SMR_OK;
    case SMR_OPT_INGEST_STRIP_WIDTH:
        if (value != 0 && value != 32 && value != 64) return smr_fail(ctx, SMR_ERR_INVALID, "smr_ctx_set_option: strip width %d", value);
        ctx->force_tw = value;
        return SMR_OK;
    default:
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_ctx_set_option: unknown option %u", option);
    }
}

const char *smr_last_error(const smr_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int smr_sync(smr_ctx *ctx) {
    SMR_ENTER(ctx);
    if (!ctx) return SMR_ERR_INVALID;
    SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SMR_OK;
}

int smr_timer_start(smr_ctx *ctx) {
    SMR_ENTER(ctx);
    if (!ctx) return SMR_ERR_INVALID;
    SMR_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    return SMR_OK;
}

int smr_timer_stop(smr_ctx *ctx, float *ms) {
    SMR_ENTER(ctx);
    if (!ctx || !ms) return SMR_ERR_INVALID;
    SMR_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    SMR_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
    SMR_HIP(ctx, hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
    return SMR_OK;
}

int smr_profile_enable(smr_ctx *ctx, int enable) {
    SMR_ENTER(ctx);
    if (!ctx) return SMR_ERR_INVALID;
    if (!enable) {
        SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        drain_profile(ctx);
    }
    ctx->profiling = enable != 0;
    return SMR_OK;
}

int smr_profile_read(smr_ctx *ctx, int stage, float *total_ms, uint32_t *launches) {
    SMR_ENTER(ctx);
    if (!ctx || stage < 0 || stage >= SMR_NUM_STAGES) return SMR_ERR_INVALID;
    SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_profile(ctx);
    if (total_ms) *total_ms = ctx->stage_ms[stage];
    if (launches) *launches = ctx->stage_launches[stage];
    return SMR_OK;
}

int smr_debug_kernel_launches(const smr_ctx *ctx, uint32_t kernel, uint64_t *count) {
    if (!ctx || !count || kernel >= SMR_KERNEL_COUNT_) return SMR_ERR_INVALID;
    *count = ctx->kernel_launches[kernel];
    return SMR_OK;
}

int smr_profile_reset(smr_ctx *ctx) {
    SMR_ENTER(ctx);
    if (!ctx) return SMR_ERR_INVALID;
    SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_profile(ctx);
    for (int i = 0; i < SMR_NUM_STAGES; i++) {
        ctx->stage_ms[i] = 0.f;
        ctx->stage_launches[i] = 0;
    }
    return SMR_OK;
}

// ---------------------------------------------------------------------------- surfaces
int smr_surface_create(smr_ctx *ctx, uint32_t w, uint32_t h, uint32_t format, smr_surface **out) {
    return surface_create_with(ctx, w, h, format, 0, out);
}

static int surface_create_with(smr_ctx *ctx, u32 w, u32 h, u32 format, size_t headroom, smr_surface **out) {
    if (!ctx || !out) return SMR_ERR_INVALID;
    *out = nullptr;
    u32 bpp = bytes_per_px(format);
    if (!bpp) return smr_fail(ctx, SMR_ERR_INVALID, "smr_surface_create: unknown pixel format %u", format);
    // MAX_NODE_RESOLUTION, smelter-render/src/types.rs:146-149
    if (w == 0 || h == 0 || w > 7682 * 2 || h > 4320 * 2)
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_surface_create: bad size %ux%u", w, h);
    SMR_HIP(ctx, hipSetDevice(ctx->device));
    smr_surface *s = new smr_surface();
    s->w = w;
    s->h = h;
    s->fmt = format;
    s->pitch = ((size_t)w * bpp + 255) & ~(size_t)255;
    s->owned = true;
    s->capacity = s->pitch * h + headroom + SMR_SURFACE_TAIL;
    hipError_t e = hipMalloc(&s->ptr, s->capacity);
    if (e != hipSuccess) {
        delete s;
        return smr_check_hip(ctx, e, "hipMalloc(surface)");
    }
    *out = s;
    return SMR_OK;
}

int smr_surface_wrap(smr_ctx *ctx, void *dptr, size_t pitch, uint32_t w, uint32_t h, uint32_t format,
                     smr_surface **out) {
    SMR_ENTER(ctx);
    if (!ctx || !out || !dptr) return SMR_ERR_INVALID;
    u32 bpp = bytes_per_px(format);
    if (!bpp || w == 0 || h == 0 || pitch < (size_t)w * bpp || (pitch % 4) != 0 || ((uintptr_t)dptr % 16) != 0)
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_surface_wrap: bad geometry (pitch %zu, %ux%u fmt %u)", pitch, w, h,
                        format);
    smr_surface *s = new smr_surface();
    s->ptr = dptr;
    s->pitch = pitch;
    s->w = w;
    s->h = h;
    s->fmt = format;
    s->owned = false;
    *out = s;
    return SMR_OK;
}

void smr_surface_destroy(smr_ctx *ctx, smr_surface *s) {
    if (!s) return;
    if (s->owned && s->ptr) {
        if (ctx) {
            (void)hipSetDevice(ctx->device);
            (void)hipStreamSynchronize(ctx->stream);
        }
        (void)hipFree(s->ptr);
    }
    delete s;
}

int smr_surface_info_get(const smr_surface *s, smr_surface_info *out) {
    if (!s || !out) return SMR_ERR_INVALID;
    out->width = s->w;
    out->height = s->h;
    out->format = s->fmt;
    out->owned = s->owned ? 1 : 0;
    out->pitch = s->pitch;
    out->dptr = s->ptr;
    return SMR_OK;
}

int smr_surface_upload(smr_ctx *ctx, smr_surface *s, const void *host, size_t host_pitch) {
    SMR_ENTER(ctx);
    if (!ctx || !s || !host) return SMR_ERR_INVALID;
    size_t row = (size_t)s->w * bytes_per_px(s->fmt);
    if (host_pitch == 0) host_pitch = row;
    if (host_pitch < row) return smr_fail(ctx, SMR_ERR_INVALID, "smr_surface_upload: host pitch too small");
    SMR_HIP(ctx, hipMemcpy2DAsync(s->ptr, s->pitch, host, host_pitch, row, s->h, hipMemcpyHostToDevice, ctx->stream));
    // the host buffer is pageable caller memory: make the call safe to return from
    SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SMR_OK;
}

int smr_surface_download(smr_ctx *ctx, const smr_surface *s, void *host, size_t host_pitch) {
    SMR_ENTER(ctx);
    if (!ctx || !s || !host) return SMR_ERR_INVALID;
    size_t row = (size_t)s->w * bytes_per_px(s->fmt);
    if (host_pitch == 0) host_pitch = row;
    if (host_pitch < row) return smr_fail(ctx, SMR_ERR_INVALID, "smr_surface_download: host pitch too small");
    SMR_HIP(ctx, hipMemcpy2DAsync(host, host_pitch, s->ptr, s->pitch, row, s->h, hipMemcpyDeviceToHost, ctx->stream));
    SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SMR_OK;
}

int smr_surface_clear(smr_ctx *ctx, smr_surface *s) {
    SMR_ENTER(ctx);
    if (!ctx || !s) return SMR_ERR_INVALID;
    SMR_HIP(ctx, hipMemsetAsync(s->ptr, 0, s->pitch * s->h, ctx->stream));
    return SMR_OK;
}

// ------------------------------------------------------------------------------ frames
// plane geometry per FrameData variant (wgpu/texture/planar_yuv.rs:65-84, nv12.rs,
// interleaved_yuv422.rs, bgra_linear.rs, argb_linear.rs)
static int plane_geometry(u32 format, u32 w, u32 h, u32 pw[3], u32 ph[3], u32 pf[3]) {
    for (int i = 0; i < 3; i++) pw[i] = ph[i] = 0, pf[i] = SMR_PX_R8;
    switch (format) {
    case SMR_FRAME_PLANAR_YUV420:
    case SMR_FRAME_PLANAR_YUVJ420:
        pw[0] = w; ph[0] = h; pw[1] = pw[2] = w / 2; ph[1] = ph[2] = h / 2; return 3;
    case SMR_FRAME_PLANAR_YUV422:
        pw[0] = w; ph[0] = h; pw[1] = pw[2] = w / 2; ph[1] = ph[2] = h; return 3;
    case SMR_FRAME_PLANAR_YUV444:
        pw[0] = w; ph[0] = h; pw[1] = pw[2] = w; ph[1] = ph[2] = h; return 3;
    case SMR_FRAME_NV12:
        pw[0] = w; ph[0] = h; pw[1] = w / 2; ph[1] = h / 2; pf[1] = SMR_PX_RG8; return 2;
    case SMR_FRAME_UYVY422:
    case SMR_FRAME_YUYV422:
        pw[0] = w / 2; ph[0] = h; pf[0] = SMR_PX_RGBA8; return 1;
    case SMR_FRAME_BGRA:
    case SMR_FRAME_ARGB:
    case SMR_FRAME_RGBA:
        pw[0] = w; ph[0] = h; pf[0] = SMR_PX_RGBA8; return 1;
    default: return 0;
    }
}

// A caller-filled smr_frame (planes may come from smr_surface_wrap): every plane must have the geometry and pixel format its
// FrameData variant implies, or the kernels would read / write outside it.  (1x1 placeholders stand for empty chroma planes.)
int smr_validate_frame(smr_ctx *ctx, const smr_frame *f, const char *what) {
    if (!f) return smr_fail(ctx, SMR_ERR_INVALID, "%s: null frame", what);
    u32 pw[3], ph[3], pf[3];
    const int n = plane_geometry(f->format, f->width, f->height, pw, ph, pf);
    if (n == 0) return smr_fail(ctx, SMR_ERR_INVALID, "%s: unknown frame format %u", what, f->format);
    if (f->width == 0 || f->height == 0) return smr_fail(ctx, SMR_ERR_INVALID, "%s: empty frame", what);
    for (int i = 0; i < n; i++) {
        const smr_surface *s = f->planes[i];
        if (!s || !s->ptr) return smr_fail(ctx, SMR_ERR_INVALID, "%s: plane %d is missing", what, i);
        const u32 w = pw[i] ? pw[i] : 1, h = ph[i] ? ph[i] : 1;
        if (s->w != w || s->h != h || s->fmt != pf[i] || s->pitch < (size_t)w * bytes_per_px(pf[i]))
            return smr_fail(ctx, SMR_ERR_INVALID, "%s: plane %d is %ux%u fmt %u pitch %zu, a %ux%u frame of format %u needs %ux%u fmt %u", what, i, s->w,
                            s->h, s->fmt, s->pitch, f->width, f->height, f->format, w, h, pf[i]);
    }
    return SMR_OK;
}

int smr_frame_create(smr_ctx *ctx, uint32_t format, uint32_t w, uint32_t h, smr_frame *out) {
    SMR_ENTER(ctx);
    if (!ctx || !out) return SMR_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    u32 pw[3], ph[3], pf[3];
    int n = plane_geometry(format, w, h, pw, ph, pf);
    if (n == 0) return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_create: unknown frame format %u", format);
    out->format = format;
    out->width = w;
    out->height = h;
    for (int i = 0; i < n; i++) {
        // chroma planes of 1-pixel-wide/high frames would be empty: keep a 1x1 plane so kernels stay valid
        int rc = smr_surface_create(ctx, pw[i] ? pw[i] : 1, ph[i] ? ph[i] : 1, pf[i], &out->planes[i]);
        if (rc != SMR_OK) {
            smr_frame_destroy(ctx, out);
            return rc;
        }
    }
    return SMR_OK;
}

void smr_frame_destroy(smr_ctx *ctx, smr_frame *f) {
    if (!f) return;
    for (int i = 0; i < 3; i++) {
        if (f->planes[i]) smr_surface_destroy(ctx, f->planes[i]);
        f->planes[i] = nullptr;
    }
}

int smr_frame_upload(smr_ctx *ctx, const smr_frame *f, const void *const host_planes[3]) {
    SMR_ENTER(ctx);
    if (!ctx || !f || !host_planes) return SMR_ERR_INVALID;
    u32 pw[3], ph[3], pf[3];
    int n = plane_geometry(f->format, f->width, f->height, pw, ph, pf);
    for (int i = 0; i < n; i++) {
        if (!f->planes[i] || !host_planes[i]) return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_upload: missing plane %d", i);
        if (pw[i] == 0 || ph[i] == 0) continue;
        const smr_surface *s = f->planes[i];
        size_t row = (size_t)pw[i] * bytes_per_px(pf[i]);
        SMR_HIP(ctx, hipMemcpy2DAsync(s->ptr, s->pitch, host_planes[i], row, row, ph[i], hipMemcpyHostToDevice, ctx->stream));
    }
    SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SMR_OK;
}

// Pinned host memory + stream-ordered copies: the zero-staging variant of InputTexture::upload / download_buffer for hosts that
// can place decoder output / encoder input in buffers from smr_host_alloc.  The copies are enqueued on the ctx stream and
// return at once; the host buffers must stay untouched until smr_sync (or a later blocking call) has returned.
int smr_host_alloc(smr_ctx *ctx, size_t bytes, void **out) {
    SMR_ENTER(ctx);
    if (!ctx || !out || !bytes) return SMR_ERR_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) return smr_fail(ctx, SMR_ERR_OOM, "smr_host_alloc: %zu B of pinned memory", bytes);
    return SMR_OK;
}
void smr_host_free(smr_ctx *ctx, void *p) {
    (void)ctx;
    if (p) (void)hipHostFree(p);
}
static int frame_copy_async(smr_ctx *ctx, const smr_frame *f, void *const host_planes[3], bool to_device, const char *what) {
    SMR_ENTER(ctx);
    if (!ctx || !f || !host_planes) return SMR_ERR_INVALID;
    u32 pw[3], ph[3], pf[3];
    int n = plane_geometry(f->format, f->width, f->height, pw, ph, pf);
    for (int i = 0; i < n; i++) {
        if (!f->planes[i] || !host_planes[i]) return smr_fail(ctx, SMR_ERR_INVALID, "%s: missing plane %d", what, i);
        if (pw[i] == 0 || ph[i] == 0) continue;
        const smr_surface *s = f->planes[i];
        size_t row = (size_t)pw[i] * bytes_per_px(pf[i]);
        if (to_device) SMR_HIP(ctx, hipMemcpy2DAsync(s->ptr, s->pitch, host_planes[i], row, row, ph[i], hipMemcpyHostToDevice, ctx->stream));
        else SMR_HIP(ctx, hipMemcpy2DAsync(host_planes[i], row, s->ptr, s->pitch, row, ph[i], hipMemcpyDeviceToHost, ctx->stream));
    }
    return SMR_OK;
}
int smr_frame_upload_async(smr_ctx *ctx, const smr_frame *f, const void *const host_planes[3]) {
    return frame_copy_async(ctx, f, (void *const *)host_planes, true, "smr_frame_upload_async");
}
int smr_frame_download_async(smr_ctx *ctx, const smr_frame *f, void *const host_planes[3]) {
    return frame_copy_async(ctx, f, host_planes, false, "smr_frame_download_async");
}

int smr_frame_download(smr_ctx *ctx, const smr_frame *f, void *const host_planes[3]) {
    SMR_ENTER(ctx);
    if (!ctx || !f || !host_planes) return SMR_ERR_INVALID;
    u32 pw[3], ph[3], pf[3];
    int n = plane_geometry(f->format, f->width, f->height, pw, ph, pf);
    for (int i = 0; i < n; i++) {
        if (!f->planes[i] || !host_planes[i]) return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_download: missing plane %d", i);
        if (pw[i] == 0 || ph[i] == 0) continue;
        const smr_surface *s = f->planes[i];
        size_t row = (size_t)pw[i] * bytes_per_px(pf[i]);
        SMR_HIP(ctx, hipMemcpy2DAsync(host_planes[i], row, s->ptr, s->pitch, row, ph[i], hipMemcpyDeviceToHost, ctx->stream));
    }
    SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SMR_OK;
}

}  // extern "C"
