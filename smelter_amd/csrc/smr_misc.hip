// smr_misc.hip — text-node glyph blit and the built-in "shader" kernels.
//
// Replaces the per-scene-update raster side of TextRendererNode::render
// (smelter-render/src/transformations/text_renderer.rs:72-167: Clear(background), then the
// glyph quads alpha-blended from the coverage atlas) and stands in for ShaderNode::render
// (transformations/shader/node.rs:71-89) with a library of built-in kernels — arbitrary
// user WGSL is out of scope (no naga->HIP compiler).
#include "smr_internal.h"

#include <cmath>

int smr_launch_plane_shader(smr_ctx *ctx, uint32_t id, const void *params, size_t params_size, const smr_surface *const *src, uint32_t n_src,
                            smr_surface *dst, float time_s);

namespace {

__device__ __forceinline__ float srgb_to_linear_dev(float c) {
    // wgpu/utils.rs:74-81 evaluated on the host in f64 -> passed in already converted
    return c;
}

// One thread per target pixel; glyph quads applied in order (painter's), OVER, with the
// target re-quantised to RGBA8 after every glyph like a render-target store.
__global__ __launch_bounds__(256) void k_blit_glyphs(SurfView target, float4 bg, const smr_glyph *__restrict__ glyphs,
                                                     const float4 *__restrict__ glyph_rgb, int n, const u8 *__restrict__ atlas,
                                                     int aw, int srgb, const float *__restrict__ tables) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= target.w || y >= target.h) return;
    const float *dec = tables, *thr = tables + 256;
    u32 r, g, b, a;
    if (srgb) { r = srgb_encode8(bg.x, thr); g = srgb_encode8(bg.y, thr); b = srgb_encode8(bg.z, thr); }
    else { r = unorm8(bg.x); g = unorm8(bg.y); b = unorm8(bg.z); }
    a = unorm8(bg.w);
    for (int i = 0; i < n; i++) {
        const smr_glyph &gl = glyphs[i];
        const int gx = x - gl.dst_x, gy = y - gl.dst_y;
        if (gx < 0 || gy < 0 || gx >= gl.w || gy >= gl.h) continue;
        const float cov = (float)atlas[(size_t)(gl.atlas_y + gy) * aw + (gl.atlas_x + gx)] / 255.0f;
        const float al = gl.color[3] * cov;
        const float inv = 1.0f - al;
        const float4 col = glyph_rgb[i];
        if (srgb) {
            r = srgb_encode8(col.x * al + dec[r] * inv, thr);
            g = srgb_encode8(col.y * al + dec[g] * inv, thr);
            b = srgb_encode8(col.z * al + dec[b] * inv, thr);
        } else {
            r = unorm8(col.x * al + ((float)r / 255.0f) * inv);
            g = unorm8(col.y * al + ((float)g / 255.0f) * inv);
            b = unorm8(col.z * al + ((float)b / 255.0f) * inv);
        }
        a = unorm8(al + ((float)a / 255.0f) * inv);
    }
    *(u32 *)(target.ptr + (size_t)y * target.pitch + (size_t)x * 4) = r | (g << 8) | (b << 16) | (a << 24);
}

// Separable gaussian: one axis per launch, RGBA8 (node encoding) between the passes.  A workgroup owns a BW x BH block of the target:
// the taps' weights (one expf each) and the block's source footprint — BH rows of BW + 2 r texels, or BH + 2 r rows of BW — are
// computed / fetched and decoded once into LDS, then every pixel sums its 2 r + 1 taps from there in tap order (the order, the
// running weight sum and the final division are those of the one-thread-per-pixel loop this replaces: 22 -> 5 us per axis on
// configs[4]'s 960x540 layer, profiles/r03_gauss.txt).
constexpr int G_MAX_RADIUS = 192;  // ceil(3 * 64)
template <int BW, int BH, int AXIS>
__global__ __launch_bounds__(256) void k_gauss_axis(SurfView src, SurfView dst, float sigma, int radius, int pxi, const float *__restrict__ tables) {
    static_assert(BW * BH == 256, "one thread per pixel of the block");
    extern __shared__ __attribute__((aligned(16))) float4 s_tex[];
    __shared__ float s_w[2 * G_MAX_RADIUS + 1];
    __shared__ float s_tab[SMR_TABLE_FLOATS];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * BW, y0 = blockIdx.y * BH;
    for (int i = tid; i <= 2 * radius; i += 256) {
        const int k = i - radius;
        s_w[i] = sigma > 0.0f ? expf(-((float)k * (float)k) / (2.0f * sigma * sigma)) : (k == 0 ? 1.0f : 0.0f);
    }
    for (int i = tid; i < SMR_TABLE_FLOATS; i += 256) s_tab[i] = tables[i];
    __syncthreads();
    const int SW = AXIS == 0 ? BW + 2 * radius : BW, SH = AXIS == 0 ? BH : BH + 2 * radius;
    for (int i = tid; i < SW * SH; i += 256) {
        const int ly = i / SW, lx = i - ly * SW;
        const int sx = clampi(x0 + lx - (AXIS == 0 ? radius : 0), 0, src.w - 1), sy = clampi(y0 + ly - (AXIS == 1 ? radius : 0), 0, src.h - 1);
        s_tex[i] = load_texel(src, pxi, sx, sy, s_tab);
    }
    __syncthreads();
    const int ly = tid / BW, lx = tid - ly * BW;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= dst.w || y >= dst.h) return;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    float ws = 0.0f;
    const float4 *t0 = &s_tex[ly * SW + lx];  // tap 0 of this pixel: radius texels to the left / above
    const int step = AXIS == 0 ? 1 : SW;
    for (int i = 0; i <= 2 * radius; i++) {
        const float wt = s_w[i];
        const float4 t = t0[i * step];
        sum.x = sum.x + t.x * wt; sum.y = sum.y + t.y * wt; sum.z = sum.z + t.z * wt; sum.w = sum.w + t.w * wt;
        ws = ws + wt;
    }
    store_texel(dst, pxi, x, y, make_float4(sum.x / ws, sum.y / ws, sum.z / ws, sum.w / ws), s_tab + 256);
}

template <int BW, int BH, int AXIS>
void launch_gauss_axis(smr_ctx *ctx, const SurfView &src, const SurfView &dst, float sigma, int radius, int pxi) {
    const size_t lds = (size_t)(AXIS == 0 ? (BW + 2 * radius) * BH : BW * (BH + 2 * radius)) * sizeof(float4);
    dim3 grid((dst.w + BW - 1) / BW, (dst.h + BH - 1) / BH, 1);
    hipLaunchKernelGGL((k_gauss_axis<BW, BH, AXIS>), grid, dim3(256), lds, ctx->stream, src, dst, sigma, radius, pxi, ctx->d_tables);
}

double srgb_to_linear_f64(double c) {
    if (c < 0.04045) return c / 12.92;
    return pow((c + 0.055) / 1.055, 2.4);
}

}  // namespace

extern "C" {

int smr_blit_glyphs(smr_ctx *ctx, smr_surface *target, const float bg[4], const smr_glyph *glyphs, uint32_t n,
                    const uint8_t *atlas_host, uint32_t atlas_w, uint32_t atlas_h) {
    SMR_ENTER(ctx);
    if (!ctx || !target || !bg || (n && (!glyphs || !atlas_host))) return SMR_ERR_INVALID;
    if (target->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_blit_glyphs: target must be RGBA8");
    for (u32 i = 0; i < n; i++) {
        const smr_glyph &g = glyphs[i];
        if (g.w < 0 || g.h < 0 || g.atlas_x < 0 || g.atlas_y < 0 || (u32)(g.atlas_x + g.w) > atlas_w || (u32)(g.atlas_y + g.h) > atlas_h)
            return smr_fail(ctx, SMR_ERR_INVALID, "smr_blit_glyphs: glyph %u reads outside the %ux%u atlas", i, atlas_w, atlas_h);
    }
    // Text nodes render once per scene update (text_renderer.rs:73-75,166): a blocking upload is fine here.
    const size_t g_bytes = ((size_t)n * sizeof(smr_glyph) + 255) & ~(size_t)255;
    const size_t c_bytes = ((size_t)n * sizeof(float4) + 255) & ~(size_t)255;
    const size_t a_bytes = (size_t)atlas_w * atlas_h;
    u8 *dev = (u8 *)smr_scratch(ctx, 2, g_bytes + c_bytes + a_bytes + 256);
    if (!dev) return SMR_ERR_OOM;
    std::vector<float4> cols(n ? n : 1);
    for (u32 i = 0; i < n; i++) {
        float c[3];
        for (int k = 0; k < 3; k++) {
            float v = glyphs[i].color[k];
            v = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
            c[k] = ctx->srgb() ? (float)srgb_to_linear_f64((double)v) : v;
        }
        cols[i] = make_float4(c[0], c[1], c[2], 0.0f);
    }
    if (n) {
        SMR_HIP(ctx, hipMemcpyAsync(dev, glyphs, (size_t)n * sizeof(smr_glyph), hipMemcpyHostToDevice, ctx->stream));
        SMR_HIP(ctx, hipMemcpyAsync(dev + g_bytes, cols.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        SMR_HIP(ctx, hipMemcpyAsync(dev + g_bytes + c_bytes, atlas_host, a_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    dim3 grid((target->w + 63) / 64, (target->h + 3) / 4, 1);
    hipLaunchKernelGGL(k_blit_glyphs, grid, dim3(256), 0, ctx->stream, view_of(target), make_float4(bg[0], bg[1], bg[2], bg[3]),
                       (const smr_glyph *)dev, (const float4 *)(dev + g_bytes), (int)n, (const u8 *)(dev + g_bytes + c_bytes),
                       (int)atlas_w, ctx->srgb() ? 1 : 0, ctx->d_tables);
    SMR_HIP(ctx, hipGetLastError());
    SMR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host vectors above go out of scope
    return SMR_OK;
}

int smr_builtin_shader(smr_ctx *ctx, uint32_t id, const void *params, size_t params_size, const smr_surface *const *src,
                       uint32_t n_src, smr_surface *dst, float time_s) {
    SMR_ENTER(ctx);
    if (!ctx || !dst) return SMR_ERR_INVALID;
    switch (id) {
    case SMR_SHADER_GAUSSIAN_BLUR: {
        if (!params || params_size < sizeof(smr_gaussian_blur_params) || n_src < 1 || !src || !src[0])
            return smr_fail(ctx, SMR_ERR_INVALID, "gaussian blur: needs smr_gaussian_blur_params and one source");
        const smr_surface *s = src[0];
        if (s->fmt != SMR_PX_RGBA8 || dst->fmt != SMR_PX_RGBA8 || s->w != dst->w || s->h != dst->h)
            return smr_fail(ctx, SMR_ERR_INVALID, "gaussian blur: RGBA8 surfaces of equal size");
        float sigma = ((const smr_gaussian_blur_params *)params)->sigma;
        if (!(sigma >= 0.0f) || sigma > 64.0f) return smr_fail(ctx, SMR_ERR_INVALID, "gaussian blur: sigma %f outside [0, 64]", sigma);
        int radius = (int)ceilf(3.0f * sigma);
        smr_surface tmp;
        tmp.w = s->w; tmp.h = s->h; tmp.fmt = SMR_PX_RGBA8;
        tmp.pitch = ((size_t)s->w * 4 + 255) & ~(size_t)255;
        tmp.ptr = smr_scratch(ctx, 3, tmp.pitch * tmp.h);
        if (!tmp.ptr) return SMR_ERR_OOM;
        const int pxi = ctx->srgb() ? PXI_RGBA8_SRGB : PXI_RGBA8_UNORM;
        StageScope scope(ctx, SMR_STAGE_LAYOUT);
        // rows: 64 x 4 blocks (28.7 KB of staged texels at the largest radius); columns: the widest block whose footprint stays small
        launch_gauss_axis<64, 4, 0>(ctx, view_of(s), view_of(&tmp), sigma, radius, pxi);
        if (radius <= 28) launch_gauss_axis<32, 8, 1>(ctx, view_of(&tmp), view_of(dst), sigma, radius, pxi);
        else if (radius <= 56) launch_gauss_axis<16, 16, 1>(ctx, view_of(&tmp), view_of(dst), sigma, radius, pxi);
        else launch_gauss_axis<8, 32, 1>(ctx, view_of(&tmp), view_of(dst), sigma, radius, pxi);  // (<= 53 KB at radius 192)
        SMR_HIP(ctx, hipGetLastError());
        return SMR_OK;
    }
    case SMR_SHADER_GRADIENT:
    case SMR_SHADER_RED_BORDER:
    case SMR_SHADER_CIRCLE_LAYOUT:
    case SMR_SHADER_FADE_TO_BALL:
    case SMR_SHADER_LAYOUT_PLANES:
    case SMR_SHADER_COLOR_BY_TEXTURE_COUNT:
    case SMR_SHADER_SILLY:
        return smr_launch_plane_shader(ctx, id, params, params_size, src, n_src, dst, time_s);  // smr_shaders.hip
    default:
        // RegisterRendererError for unknown shaders; arbitrary WGSL is not supported by this build
        return smr_fail(ctx, SMR_ERR_INVALID, "smr_builtin_shader: unknown built-in shader id %u", id);
    }
}

}  // extern "C"
