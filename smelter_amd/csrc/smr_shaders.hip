// smr_shaders.hip — the reference's in-tree WGSL shaders as built-in kernels of the shader node (include/smr.h:
// smr_builtin_shader ids 1..7).  ShaderNode::render / ShaderPipeline::render (transformations/shader/node.rs:71-89,
// shader/pipeline.rs:81-141): clear the target to transparent, draw one plane per source texture with BaseShaderParameters
// {plane_id, time, output_resolution, texture_count}, premultiplied-alpha blending, sRGB (GpuOptimized) render target.
//
// One thread per target pixel walks the planes in order: the vertex stage of each shader is an axis-aligned scale + translate
// of the unit quad (inverted here to find the fragment's tex_coords), the fragment stage is the WGSL body restated line for
// line; after every plane the running colour is quantised to RGBA8 exactly where the render-target store would.
#include "smr_internal.h"

#include <cmath>

namespace {

struct ShaderArgs {
    SurfView dst;
    SurfView src[SMR_SHADER_MAX_SOURCES];
    int n_src, id, pxi;
    float time;
    smr_circle_layout circles[SMR_SHADER_MAX_SOURCES];
};

__device__ __forceinline__ float smoothstep_dev(float e0, float e1, float x) {
    // WGSL smoothstep: t = clamp((x - e0) / (e1 - e0), 0, 1); t * t * (3 - 2 t)  (edges may be given high-to-low)
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

// vertex stage: clip-space position of the unit quad's corner (x, y) = position * (sx, sy) + (cx, cy)
__device__ __forceinline__ void plane_transform(const ShaderArgs &a, int plane, float &sx, float &sy, float &cx, float &cy) {
    sx = 1.0f; sy = 1.0f; cx = 0.0f; cy = 0.0f;
    if (a.id == SMR_SHADER_CIRCLE_LAYOUT) {  // circle_layout.wgsl vs_main
        const smr_circle_layout &c = a.circles[plane < 0 ? 0 : plane];
        const float W = (float)a.dst.w, H = (float)a.dst.h;
        sx = (float)c.width_px / W;
        sy = (float)c.height_px / H;
        cx = (((float)c.left_px + ((float)c.width_px / 2.0f)) / W) * 2.0f - 1.0f;
        cy = 1.0f - (((float)c.top_px + ((float)c.height_px / 2.0f)) / H) * 2.0f;
    } else if (a.id == SMR_SHADER_LAYOUT_PLANES && plane != -1) {  // layout_planes.wgsl vs_main
        sx = 0.5f; sy = 0.5f;
        if (plane == 0) { cx = -0.5f; cy = 0.5f; }
        else if (plane == 1) { cx = 0.5f; cy = 0.5f; }
        else if (plane == 2) { cx = -0.5f; cy = -0.5f; }
        else if (plane == 3) { cx = 0.5f; cy = -0.5f; }
    }
}

__device__ __forceinline__ float4 tex(const ShaderArgs &a, int i, float u, float v, const float *__restrict__ dec) {
    if (i < 0 || i >= a.n_src || !a.src[i].ptr) return make_float4(0.f, 0.f, 0.f, 0.f);
    return sample_rgba_bilinear(a.src[i], a.pxi, u, v, dec);
}

// fragment stage: premultiplied RGBA in the target's blending space
__device__ __forceinline__ float4 fragment(const ShaderArgs &a, int plane, float u, float v, float fx, float fy, const float *__restrict__ dec) {
    switch (a.id) {
    case SMR_SHADER_GRADIENT:  // gradient.wgsl:36-38
        return make_float4(u, 0.0f, 0.0f, 1.0f);
    case SMR_SHADER_RED_BORDER: {  // red_border.wgsl:40-52
        const float4 sample = tex(a, 0, u, v, dec);
        const float border = 50.0f;
        if (fx > border && fx < (float)a.dst.w - border && fy > border && fy < (float)a.dst.h - border) return sample;
        return make_float4(1.0f, 0.0f, 0.0f, 1.0f);
    }
    case SMR_SHADER_CIRCLE_LAYOUT: {  // circle_layout.wgsl:58-72
        const smr_circle_layout &c = a.circles[plane < 0 ? 0 : plane];
        const float du = u - 0.5f, dv = v - 0.5f;
        const float in_circle = sqrtf(du * du + dv * dv) < 0.5f ? 1.0f : 0.0f;
        const float4 s = tex(a, plane, u, v, dec);
        return make_float4(s.x * in_circle + c.background_color[0] * (1.0f - in_circle), s.y * in_circle + c.background_color[1] * (1.0f - in_circle),
                           s.z * in_circle + c.background_color[2] * (1.0f - in_circle), s.w * in_circle + c.background_color[3] * (1.0f - in_circle));
    }
    case SMR_SHADER_FADE_TO_BALL: {  // fade_to_ball.wgsl:38-52
        const float4 s = tex(a, 0, u, v, dec);
        const float radius = a.time / 5.0f, eps = 0.15f;
        const float du = u - 0.5f, dv = v - 0.5f;
        const float t = smoothstep_dev(radius + eps, radius - eps, sqrtf(du * du + dv * dv));
        return make_float4(s.x * t, s.y * t, s.z * t, s.w * t);
    }
    case SMR_SHADER_LAYOUT_PLANES:  // layout_planes.wgsl:55-61
        if (plane == -1) return make_float4(1.0f, 0.0f, 0.0f, 1.0f);
        return tex(a, plane, u, v, dec);
    case SMR_SHADER_COLOR_BY_TEXTURE_COUNT:  // color_output_with_texture_count.wgsl:42-50
        if (a.n_src == 0) return make_float4(1.0f, 0.0f, 0.0f, 1.0f);
        if (a.n_src == 1) return make_float4(0.0f, 1.0f, 0.0f, 1.0f);
        return make_float4(0.0f, 0.0f, 1.0f, 1.0f);
    case SMR_SHADER_SILLY: {  // examples/silly.wgsl:34-54
        if (a.n_src != 1) return make_float4(0.f, 0.f, 0.f, 0.f);
        const float pi = 3.14159f;
        const float effect_radius = fabsf(sinf(a.time) / 2.0f);
        const float effect_angle = 2.0f * pi * fabsf(sinf(a.time) / 2.0f);
        const float du = u - 0.5f, dv = v - 0.5f;
        const float len = sqrtf(du * du + dv * dv);
        const float angle = atan2f(dv, du) + effect_angle * smoothstep_dev(effect_radius, 0.0f, len);
        return tex(a, 0, len * cosf(angle) + 0.5f, len * sinf(angle) + 0.5f, dec);
    }
    default:
        return make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__global__ __launch_bounds__(256) void k_shader_planes(const ShaderArgs a, const float *__restrict__ tables) {
    const float *dec = tables, *thr = tables + 256;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.dst.w || y >= a.dst.h) return;
    const float W = (float)a.dst.w, H = (float)a.dst.h;
    const float fx = (float)x + 0.5f, fy = (float)y + 0.5f;          // @builtin(position).xy
    const float X = fx / W * 2.0f - 1.0f, Y = 1.0f - fy / H * 2.0f;  // the pixel centre in clip space
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);                     // LoadOp::Clear(TRANSPARENT)
    const int first = a.n_src == 0 ? -1 : 0, last = a.n_src == 0 ? -1 : a.n_src - 1;
    for (int plane = first; plane <= last; plane++) {
        float sx, sy, cx, cy;
        plane_transform(a, plane, sx, sy, cx, cy);
        if (!(sx > 0.0f) || !(sy > 0.0f)) continue;  // a degenerate plane covers no pixel centre
        const float qx = (X - cx) / sx, qy = (Y - cy) / sy;  // position within the unit quad [-1, 1]^2
        // coverage: pixel centre inside the quad; a centre exactly on an edge belongs to the quad whose left / top edge it is
        if (!(qx >= -1.0f && qx < 1.0f && qy > -1.0f && qy <= 1.0f)) continue;
        const float u = (qx + 1.0f) * 0.5f, v = (1.0f - qy) * 0.5f;  // plane.rs:11-28: (1, -1) <-> tex (1, 1)
        const float4 f = fragment(a, plane, u, v, fx, fy, dec);
        const float k = 1.0f - f.w;  // PREMULTIPLIED_ALPHA_BLENDING (common_pipeline.rs:125)
        float4 o = make_float4(f.x + acc.x * k, f.y + acc.y * k, f.z + acc.z * k, f.w + acc.w * k);
        // render-target store, then what the next plane's blend reads back
        store_texel(a.dst, a.pxi, x, y, o, thr);
        acc = load_texel(a.dst, a.pxi, x, y, dec);
    }
    if (acc.x == 0.f && acc.y == 0.f && acc.z == 0.f && acc.w == 0.f) *(u32 *)(a.dst.ptr + (size_t)y * a.dst.pitch + (size_t)x * 4) = 0u;
}

}  // namespace

int smr_launch_plane_shader(smr_ctx *ctx, uint32_t id, const void *params, size_t params_size, const smr_surface *const *src, uint32_t n_src,
                            smr_surface *dst, float time_s) {
    if (dst->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_builtin_shader: the target must be RGBA8");
    if (n_src > SMR_SHADER_MAX_SOURCES) return smr_fail(ctx, SMR_ERR_INVALID, "smr_builtin_shader: at most %d sources", SMR_SHADER_MAX_SOURCES);
    ShaderArgs a;
    memset(&a, 0, sizeof(a));
    a.dst = view_of(dst);
    a.n_src = (int)n_src;
    a.id = (int)id;
    a.pxi = ctx->srgb() ? PXI_RGBA8_SRGB : PXI_RGBA8_UNORM;
    a.time = time_s;
    for (uint32_t i = 0; i < n_src; i++) {
        if (src && src[i]) {
            if (src[i]->fmt != SMR_PX_RGBA8) return smr_fail(ctx, SMR_ERR_INVALID, "smr_builtin_shader: source %u is not RGBA8", i);
            a.src[i] = view_of(src[i]);
        }
    }
    if (id == SMR_SHADER_CIRCLE_LAYOUT) {
        const size_t need = (size_t)(n_src ? n_src : 1) * sizeof(smr_circle_layout);
        if (!params || params_size < need) return smr_fail(ctx, SMR_ERR_INVALID, "circle_layout: needs one smr_circle_layout per source (%zu B)", need);
        memcpy(a.circles, params, need < sizeof(a.circles) ? need : sizeof(a.circles));
    }
    dim3 grid((dst->w + 63) / 64, (dst->h + 3) / 4, 1);
    StageScope scope(ctx, SMR_STAGE_LAYOUT);
    hipLaunchKernelGGL(k_shader_planes, grid, dim3(256), 0, ctx->stream, a, ctx->d_tables);
    SMR_HIP(ctx, hipGetLastError());
    return SMR_OK;
}
