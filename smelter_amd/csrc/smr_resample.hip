// smr_resample.hip — Lanczos3 separable resampler, box pre-decimation, bilinear rescale.
//
// Replaces smelter-render/src/transformations/layout/resampler.rs (pass planning and
// ResampledChild::render), resample.wgsl, downsample.wgsl, and wgpu/format/rgba_rescale.wgsl.
//
// These are the general, pass-per-launch kernels (any source format, any crop, any plan);
// the hot configuration (YUV frame -> scaled tile) runs through smr_fused.hip instead.
// Weights follow resample.wgsl:50-86 exactly: sin/cos rotation recurrence in f32, one
// weight table per output coordinate, built once per workgroup in LDS and shared by the
// 64 threads that walk the perpendicular axis.
#include "smr_resample_dev.h"

#include <cmath>

namespace {

// One workgroup = 64 x 4 output pixels (lanes run along x so every global access is row-contiguous).
// Horizontal pass: 64 weight tables (one per x); vertical pass: 4 tables (one per y).
constexpr int TILE_X = 64;
constexpr int TILE_Y = 4;

struct PassParams {
    int axis;  // 0 horizontal, 1 vertical
    float scale, offset;
    int perp_offset;
    int src_pxi, dst_pxi;
};

__global__ __launch_bounds__(TILE_X *TILE_Y) void k_resample_pass(SurfView src, SurfView dst, PassParams p,
                                                                  const float *__restrict__ tables) {
    __shared__ float s_w[TILE_X][MAX_TAPS + 1];  // +1: conflict-free per-lane rows
    __shared__ float s_wsum[TILE_X];
    __shared__ int s_first[TILE_X];

    const float *dec = tables, *thr = tables + 256;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = blockIdx.x * TILE_X + lx;
    const int y = blockIdx.y * TILE_Y + ly;
    const int o_idx = p.axis == 0 ? lx : ly;            // which weight table this thread uses
    const int out_coord = p.axis == 0 ? x : y;
    const int out_len = p.axis == 0 ? dst.w : dst.h;

    const int taps = lanczos_taps(p.scale);

    // --- weight tables: one thread per output coordinate (resample.wgsl:44-86) ---
    const bool builder = p.axis == 0 ? (ly == 0) : (lx == 0);
    if (builder && out_coord < out_len) {
        float wsum;
        s_first[o_idx] = lanczos_weights(out_coord, p.scale, p.offset, taps, s_w[o_idx], &wsum);
        s_wsum[o_idx] = wsum;
    }
    __syncthreads();
    if (x >= dst.w || y >= dst.h) return;

    const int first = s_first[o_idx];
    const int max_src = (p.axis == 1 ? src.h : src.w) - 1;
    const int perp = clampi((p.axis == 1 ? x : y) + p.perp_offset, 0, (p.axis == 1 ? src.w : src.h) - 1);

    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < taps; t++) {
        float wgt = s_w[o_idx][t];
        int s = clampi(first + t, 0, max_src);
        float4 tx = (p.axis == 1) ? load_texel(src, p.src_pxi, perp, s, dec) : load_texel(src, p.src_pxi, s, perp, dec);
        // contracted multiply-add, as the reference's drivers compile `sum += texel * weight`
        sum.x = __builtin_fmaf(tx.x, wgt, sum.x);
        sum.y = __builtin_fmaf(tx.y, wgt, sum.y);
        sum.z = __builtin_fmaf(tx.z, wgt, sum.z);
        sum.w = __builtin_fmaf(tx.w, wgt, sum.w);
    }
    const float ws = s_wsum[o_idx];
    store_texel(dst, p.dst_pxi, x, y, make_float4(sum.x / ws, sum.y / ws, sum.z / ws, sum.w / ws), thr);
}

// downsample.wgsl:27-40 — 2^k x 2^l box mean into Rgba16Float
__global__ __launch_bounds__(256) void k_downsample(SurfView src, SurfView dst, int fx, int fy, int src_pxi,
                                                    const float *__restrict__ tables) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dst.w || y >= dst.h) return;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < fy; dy++) {
        for (int dx = 0; dx < fx; dx++) {
            int sx = clampi(x * fx + dx, 0, src.w - 1);
            int sy = clampi(y * fy + dy, 0, src.h - 1);
            float4 t = load_texel(src, src_pxi, sx, sy, tables);
            sum.x = sum.x + t.x; sum.y = sum.y + t.y; sum.z = sum.z + t.z; sum.w = sum.w + t.w;
        }
    }
    float n = (float)(unsigned)(fx * fy);
    store_texel(dst, PXI_RGBA16F, x, y, make_float4(sum.x / n, sum.y / n, sum.z / n, sum.w / n), tables + 256);
}

// The same box mean for large factors (a 1080p child laid out into a few pixels — a tile growing from nothing in a transition —
// reduces by up to 512 x 512): one wave per output pixel.  The reference's invocation adds its fy * fx texels one after the other
// and f32 addition does not reassociate, so the order of the additions is kept (row by row, left to right: bit-identical to
// k_downsample); what is parallel is the fetching — 64 texels per step are loaded and decoded side by side into LDS, then four
// lanes (one per channel) run the dependent add chains over them while the next 512 texels are in flight.
// 1920x1080 -> 1x1 (12 reduced pixels of 262 144 texels each): 54 ms with one thread per pixel, 3.5 ms here.
__global__ __launch_bounds__(64) void k_downsample_wave(SurfView src, SurfView dst, int fx, int fy, int src_pxi,
                                                        const float *__restrict__ tables) {
    __shared__ float s_t[4][64];  // channel-major: lane c walks s_t[c][0..n)
    const int x = blockIdx.x, y = blockIdx.y, lane = threadIdx.x;
    constexpr int G = 8;                          // 64-texel batches fetched together: 512 texels of one source row per group
    const int gpr = (fx + 64 * G - 1) / (64 * G);  // groups per row
    const int groups = fy * gpr;
    auto fetch = [&](int gi, float4 (&t)[G]) {
        const int dy = gi / gpr, dx0 = (gi - dy * gpr) * 64 * G;
        const int sy = clampi(y * fy + dy, 0, src.h - 1);
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int dx = dx0 + g * 64 + lane;
            if (dx < fx) t[g] = load_texel(src, src_pxi, clampi(x * fx + dx, 0, src.w - 1), sy, tables);
        }
    };
    auto fence = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    float sum = 0.0f;
    float4 cur[G], nxt[G];
    fetch(0, cur);
    for (int gi = 0; gi < groups; gi++) {
        if (gi + 1 < groups) fetch(gi + 1, nxt);  // the next group's loads are in flight while this one is added up
        const int dx0 = (gi % gpr) * 64 * G;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int n = min(64, fx - dx0 - g * 64);
            if (n <= 0) break;
            if (lane < n) { s_t[0][lane] = cur[g].x; s_t[1][lane] = cur[g].y; s_t[2][lane] = cur[g].z; s_t[3][lane] = cur[g].w; }
            fence();
            if (lane < 4) {
                if (n == 64) {
#pragma unroll
                    for (int i = 0; i < 64; i++) sum = sum + s_t[lane][i];
                } else {
                    for (int i = 0; i < n; i++) sum = sum + s_t[lane][i];
                }
            }
            fence();
        }
#pragma unroll
        for (int g = 0; g < G; g++) cur[g] = nxt[g];
    }
    const float n = (float)(unsigned)(fx * fy);
    const float m = sum / n;
    const float4 o = make_float4(__shfl(m, 0), __shfl(m, 1), __shfl(m, 2), __shfl(m, 3));
    if (lane == 0) store_texel(dst, PXI_RGBA16F, x, y, o, tables + 256);
}

void launch_downsample(smr_ctx *ctx, const SurfView &src, const SurfView &dst, int fx, int fy, int src_pxi) {
    if ((long long)fx * fy >= 256) {
        hipLaunchKernelGGL(k_downsample_wave, dim3(dst.w, dst.h, 1), dim3(64), 0, ctx->stream, src, dst, fx, fy, src_pxi, ctx->d_tables);
    } else {
        dim3 grid((dst.w + 63) / 64, (dst.h + 3) / 4, 1);
        hipLaunchKernelGGL(k_downsample, grid, dim3(256), 0, ctx->stream, src, dst, fx, fy, src_pxi, ctx->d_tables);
    }
}

// rgba_rescale.wgsl:24-27
__global__ __launch_bounds__(256) void k_rescale_bilinear(SurfView src, SurfView dst, int pxi, const float *__restrict__ tables) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dst.w || y >= dst.h) return;
    float4 o = sample_rgba_bilinear(src, pxi, ((float)x + 0.5f) / (float)dst.w, ((float)y + 0.5f) / (float)dst.h, tables);
    store_texel(dst, pxi, x, y, o, tables + 256);
}

// ---- pass planning (host): smelter-render/src/transformations/layout/resampler.rs:36-145 ----
struct AxisMapping {
    int axis;
    float crop_offset, crop_len;
    int dst_len;
    float scale() const { return crop_len / (float)dst_len; }
};

int predecimate_levels(const AxisMapping &m) {
    // ((scale / KERNEL_BUDGET).log2().ceil().max(0.0) as u32).min(MAX_PREDECIMATE_LEVELS)
    float v = ceilf(log2f(m.scale() / 4.0f));
    if (std::isnan(v)) v = 0.0f;
    if (v < 0.0f) v = 0.0f;
    unsigned u = v >= 4294967296.0f ? 0xffffffffu : (unsigned)v;
    return u > 16 ? 16 : (int)u;
}

bool is_same_px(float a, float b) { return fabsf(a - b) < 0.001f; }

bool as_direct(const AxisMapping &m, int *perp) {
    bool direct = is_same_px(m.crop_len, (float)m.dst_len) && is_same_px(m.crop_offset, roundf(m.crop_offset));
    if (direct) *perp = (int)roundf(m.crop_offset);
    return direct;
}

int plan_passes(const AxisMapping maps[2], smr_resample_plan *p) {
    int ph = 0, pv = 0;
    bool dh = as_direct(maps[0], &ph), dv = as_direct(maps[1], &pv);
    if (dh && dv) return 0;
    auto set = [&](int slot, const AxisMapping &m, int perp) {
        p->axis[slot] = m.axis;
        p->scale[slot] = m.scale();
        p->offset[slot] = m.crop_offset;
        p->perp_offset[slot] = perp;
    };
    if (!dh && dv) { set(0, maps[0], pv); return 1; }
    if (dh && !dv) { set(0, maps[1], ph); return 1; }
    // stronger shrink first, to minimise the intermediate size and tap count
    int first = maps[1].scale() > maps[0].scale() ? 1 : 0;
    set(0, maps[first], 0);
    set(1, maps[1 - first], 0);
    return 2;
}

int pxi_of(const smr_ctx *ctx, const smr_surface *s) {
    if (s->fmt == SMR_PX_RGBA16F) return PXI_RGBA16F;
    return ctx->srgb() ? PXI_RGBA8_SRGB : PXI_RGBA8_UNORM;
}

int launch_pass(smr_ctx *ctx, const smr_surface *src, int axis, float scale, float offset, int perp_offset, smr_surface *dst) {
    PassParams p;
    p.axis = axis;
    p.scale = scale;
    p.offset = offset;
    p.perp_offset = perp_offset;
    p.src_pxi = pxi_of(ctx, src);
    p.dst_pxi = pxi_of(ctx, dst);
    dim3 grid((dst->w + TILE_X - 1) / TILE_X, (dst->h + TILE_Y - 1) / TILE_Y, 1);
    hipLaunchKernelGGL(k_resample_pass, grid, dim3(TILE_X * TILE_Y), 0, ctx->stream, view_of(src), view_of(dst), p, ctx->d_tables);
    return smr_check_hip(ctx, hipGetLastError(), "k_resample_pass");
}

}  // namespace

extern "C" {

int smr_resample_plan_make(uint32_t src_w, uint32_t src_h, const float crop[4], uint32_t dst_w, uint32_t dst_h,
                           smr_resample_plan *p) {
    if (!crop || !p || dst_w == 0 || dst_h == 0) return SMR_ERR_INVALID;
    memset(p, 0, sizeof(*p));
    AxisMapping maps[2] = {{0, crop[1], crop[2], (int)dst_w}, {1, crop[0], crop[3], (int)dst_h}};
    p->reduced_w = (int)src_w;
    p->reduced_h = (int)src_h;
    smr_resample_plan tmp;
    memset(&tmp, 0, sizeof(tmp));
    if (plan_passes(maps, &tmp) == 0) return 0;  // ResampledChild::is_needed == false
    for (int a = 0; a < 2; a++) p->levels[a] = predecimate_levels(maps[a]);
    const int fx = 1 << p->levels[0], fy = 1 << p->levels[1];
    if (fx != 1 || fy != 1) {
        p->reduced_w = ((int)src_w + fx - 1) / fx;
        p->reduced_h = ((int)src_h + fy - 1) / fy;
    }
    AxisMapping residual[2];
    for (int a = 0; a < 2; a++) {
        float factor = (float)(1u << p->levels[a]);
        residual[a] = maps[a];
        residual[a].crop_offset = maps[a].crop_offset / factor;
        residual[a].crop_len = maps[a].crop_len / factor;
    }
    p->kind = plan_passes(residual, p);
    if (p->kind == 0) return SMR_ERR_INTERNAL;  // the reference panics here (resampler.rs:345-346)
    if (p->kind == 2) {
        if (p->axis[0] == 0) { p->mid_w = residual[0].dst_len; p->mid_h = p->reduced_h; }
        else { p->mid_w = p->reduced_w; p->mid_h = residual[1].dst_len; }
    }
    return p->kind;
}

int smr_resample_pass(smr_ctx *ctx, const smr_surface *src, int axis, float scale, float offset, int perp_offset,
                      smr_surface *dst) {
    SMR_ENTER(ctx);
    if (!ctx || !src || !dst) return SMR_ERR_INVALID;
    if ((src->fmt != SMR_PX_RGBA8 && src->fmt != SMR_PX_RGBA16F) || (dst->fmt != SMR_PX_RGBA8 && dst->fmt != SMR_PX_RGBA16F))
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_resample_pass: surfaces must be RGBA8 or RGBA16F");
    if (axis != 0 && axis != 1) return smr_fail(ctx, SMR_ERR_INVALID, "smr_resample_pass: bad axis %d", axis);
    if (!(scale > 0.0f) || ceilf(6.0f * fmaxf(scale, 1.0f)) + 1 > MAX_TAPS)
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_resample_pass: scale %f outside (0, %f]", scale, (MAX_TAPS - 1) / 6.0);
    StageScope scope(ctx, SMR_STAGE_RESAMPLE);
    return launch_pass(ctx, src, axis, scale, offset, perp_offset, dst);
}

int smr_downsample(smr_ctx *ctx, const smr_surface *src, uint32_t fx, uint32_t fy, smr_surface *dst) {
    SMR_ENTER(ctx);
    if (!ctx || !src || !dst || fx == 0 || fy == 0) return SMR_ERR_INVALID;
    if (dst->fmt != SMR_PX_RGBA16F || (src->fmt != SMR_PX_RGBA8 && src->fmt != SMR_PX_RGBA16F))
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_downsample: src RGBA8/RGBA16F, dst RGBA16F");
    if (dst->w != (src->w + fx - 1) / fx || dst->h != (src->h + fy - 1) / fy)
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_downsample: dst must be ceil(src / factor)");
    StageScope scope(ctx, SMR_STAGE_RESAMPLE);
    launch_downsample(ctx, view_of(src), view_of(dst), (int)fx, (int)fy, pxi_of(ctx, src));
    return smr_check_hip(ctx, hipGetLastError(), "k_downsample");
}

int smr_resample(smr_ctx *ctx, const smr_surface *src, const float crop[4], smr_surface *dst) {
    SMR_ENTER(ctx);
    if (!ctx || !src || !crop || !dst) return SMR_ERR_INVALID;
    if (src->fmt != SMR_PX_RGBA8 || dst->fmt != SMR_PX_RGBA8)
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_resample: node surfaces must be RGBA8");
    smr_resample_plan plan;
    int kind = smr_resample_plan_make(src->w, src->h, crop, dst->w, dst->h, &plan);
    if (kind < 0) return smr_fail(ctx, kind, "smr_resample: box reduction left no residual scale");
    if (kind == 0) return 0;
    ctx->kernel_launches[SMR_KERNEL_RESAMPLE_GENERAL]++;
    StageScope scope(ctx, SMR_STAGE_RESAMPLE);
    const smr_surface *cur = src;
    smr_surface reduced, mid;
    const int fx = 1 << plan.levels[0], fy = 1 << plan.levels[1];
    if (fx != 1 || fy != 1) {
        reduced.w = (u32)plan.reduced_w;
        reduced.h = (u32)plan.reduced_h;
        reduced.fmt = SMR_PX_RGBA16F;
        reduced.pitch = ((size_t)reduced.w * 8 + 255) & ~(size_t)255;
        reduced.ptr = smr_scratch(ctx, 0, reduced.pitch * reduced.h);
        if (!reduced.ptr) return SMR_ERR_OOM;
        launch_downsample(ctx, view_of(src), view_of(&reduced), fx, fy, pxi_of(ctx, src));
        cur = &reduced;
    }
    int last = 0;
    if (kind == 2) {
        mid.w = (u32)plan.mid_w;
        mid.h = (u32)plan.mid_h;
        mid.fmt = SMR_PX_RGBA16F;
        mid.pitch = ((size_t)mid.w * 8 + 255) & ~(size_t)255;
        mid.ptr = smr_scratch(ctx, 1, mid.pitch * mid.h);
        if (!mid.ptr) return SMR_ERR_OOM;
        int rc = launch_pass(ctx, cur, plan.axis[0], plan.scale[0], plan.offset[0], plan.perp_offset[0], &mid);
        if (rc != SMR_OK) return rc;
        cur = &mid;
        last = 1;
    }
    int rc = launch_pass(ctx, cur, plan.axis[last], plan.scale[last], plan.offset[last], plan.perp_offset[last], dst);
    if (rc != SMR_OK) return rc;
    return kind;
}

int smr_rescale_bilinear(smr_ctx *ctx, const smr_surface *src, smr_surface *dst) {
    SMR_ENTER(ctx);
    if (!ctx || !src || !dst) return SMR_ERR_INVALID;
    if (src->fmt != SMR_PX_RGBA8 || dst->fmt != SMR_PX_RGBA8)
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_rescale_bilinear: surfaces must be RGBA8");
    StageScope scope(ctx, SMR_STAGE_RESAMPLE);
    dim3 grid((dst->w + 63) / 64, (dst->h + 3) / 4, 1);
    hipLaunchKernelGGL(k_rescale_bilinear, grid, dim3(256), 0, ctx->stream, view_of(src), view_of(dst), pxi_of(ctx, src),
                       ctx->d_tables);
    return smr_check_hip(ctx, hipGetLastError(), "k_rescale_bilinear");
}


// FramePreProcessor::process_to_bytes (state/frame_pre_processor.rs:84-107): upload_and_convert_to_node_texture, the optional
// rescale_node_texture, download — the three public passes chained on the context's own scratch surfaces.
int smr_frame_preprocess(smr_ctx *ctx, const smr_frame *in, uint32_t dst_w, uint32_t dst_h, void *host, size_t host_pitch) {
    SMR_ENTER(ctx);
    if (!ctx || !in || !host) return SMR_ERR_INVALID;
    if ((dst_w == 0) != (dst_h == 0)) return smr_fail(ctx, SMR_ERR_INVALID, "smr_frame_preprocess: target %ux%u (both zero = no rescale)", dst_w, dst_h);
    if (int rc = smr_validate_frame(ctx, in, "smr_frame_preprocess")) return rc;
    constexpr size_t SLOT_PRE_NODE = SMR_SLOT_PRE_NODE, SLOT_PRE_SCALED = SMR_SLOT_PRE_SCALED;
    smr_surface *node = smr_cached_surface(ctx, SLOT_PRE_NODE, in->width, in->height, SMR_PX_RGBA8);
    if (!node) return SMR_ERR_OOM;
    int rc = smr_frame_to_rgba(ctx, in, node);
    if (rc != SMR_OK) return rc;
    const smr_surface *out = node;
    if (dst_w) {
        smr_surface *scaled = smr_cached_surface(ctx, SLOT_PRE_SCALED, dst_w, dst_h, SMR_PX_RGBA8);
        if (!scaled) return SMR_ERR_OOM;
        rc = smr_rescale_bilinear(ctx, node, scaled);
        if (rc != SMR_OK) return rc;
        out = scaled;
    }
    return smr_surface_download(ctx, out, host, host_pitch);
}

}  // extern "C"
