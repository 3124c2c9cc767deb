// smr_layout_dev.h — device-side layout evaluation shared by the general compositor
// (smr_layout.hip) and the fused compose+output kernel (smr_fused_compose.h).
//
// Mirrors apply_layouts.wgsl:127-377 one-for-one; see the oracle (oracle/smr_oracle.c,
// layout_fragment / orc_apply_layouts) for the same maths on the CPU.  Shortcuts taken here are exact:
//   * "solid region" test — inside the rect (and every parent mask) inset past radius / border / AA band, every
//     smoothstep of the fragment stage evaluates to exactly 1, so the fragment is the base value itself;
//   * texel-aligned 1:1 blits have bilinear weights of exactly (1,0): one fetch instead of four;
//   * x / b is evaluated as the correctly rounded quotient from a precomputed RN(1/b) and two FMAs;
//   * sqrt(m*m) == m, and (x - e0) / 1 == x - e0.
#pragma once

#include <algorithm>
#include <cmath>

#include "smr_internal.h"

constexpr int LAYOUT_TILE_W = 32;
constexpr int LAYOUT_TILE_H = 8;
constexpr int MAX_LAYOUT_WORDS = 32;  // up to 1024 layouts per node

// parent mask + the inset of its solid region
struct DevMask {
    float radius[4];  // tl, tr, br, bl
    float top, left, width, height;
    float inset;      // inside the mask rect inset by this much (corner squares excepted) smoothstep(-.5,.5,-sdf) == 1 exactly
    float corner;     // side of the four corner squares in which the SDF is curved (0: none beyond the inset band)
    float pad[2];
};

// Compact per-layout record consumed by the kernels (wave-uniform: read through scalar loads).
struct alignas(16) DevLayout {
    float top, left, width, height;
    float cs, sn;         // cos / sin of rotation_degrees (vertex stage, wgsl:99-110)
    float radius[4];      // tl, tr, br, bl
    float color[4];
    float border_color[4];
    float crop[4];        // top, left, width, height
    float border_width, blur;
    float qw, qh, cx, cy; // rasterised quad: size and centre (shadow quads are grown by blur)
    float rqw, rqh;       // RN(1/qw), RN(1/qh)
    float rtw, rth;       // RN(1/tex_w), RN(1/tex_h)
    u32 type;
    u32 masks_off, masks_len;
    int bx0, by0, bx1, by1;  // pixel bounding box [x0,x1) x [y0,y1)
    int src_kind;         // 0 none (1x1 transparent), 1 RGBA8 surface, 2 RGBA8 surface known to be fully opaque
    int src_index;
    int tex_w, tex_h;
    SurfView src;
    int flags;            // DL_* below
    float inset;          // inside the rect inset by this much (corner squares excepted) the fragment equals its base value
    int ix, iy;           // DL_ALIGNED: texel = pixel - (ix, iy)
    u32 solid_px;         // DL_COLOR_OPAQUE: the colour as the render-target store would encode it
    float corner;         // side of the four corner squares in which the SDF is curved (0: none beyond the inset band)
};

enum {
    DL_UNROTATED = 1,     // rotation is exactly 0 (cs == 1, sn == 0)
    DL_ALIGNED = 2,       // texture layout that is a 1:1 texel-aligned blit (bilinear weights are exactly (1,0))
    DL_COLOR_OPAQUE = 4,  // colour / shadow layout whose premultiplied colour has alpha == 1
};

struct PackedLayouts {
    const DevLayout *layouts = nullptr;  // device
    const DevMask *masks = nullptr;      // device
    DevLayout *host_layouts = nullptr;   // pinned host copy (valid until the slot is reused)
    DevMask *host_masks = nullptr;
    int n = 0;
    int n_masks = 0;
    LayoutSlot *slot = nullptr;
    void *extra_host = nullptr;  // caller-defined parameter block riding in the same slot
    void *extra_dev = nullptr;
    size_t copy_bytes = 0;
    bool reused = false;  // smr_pack_commit found the previous frame's identical device copy: nothing was queued
};

// One layout of the POD list in the compact device form (rotation, quad and bounding box precomputed on the host in f32, as the vertex stage
// does per draw: apply_layouts.wgsl:127-157; the classification helpers of the fused compositor).  Host code: smr_pack_layouts and — so that
// the compositor kernels can be run on the CPU against the oracle — the lane emulator of tests/emu.  `thr`: the context's sRGB encode
// thresholds (tables + 256); `hm` / `mo`: the mask array and the next free entry in it.
inline void smr_pack_one_layout(const smr_layout &L, const SurfView *src_views, const int *src_kind, u32 n_sources, int out_w, int out_h, bool srgb,
                                const float *thr, DevLayout &D, DevMask *hm, u32 &mo) {
    const float DEG = 0.017453292519943295f;
    memset(&D, 0, sizeof(D));
    D.top = L.top; D.left = L.left; D.width = L.width; D.height = L.height;
    for (int k = 0; k < 4; k++) {
        D.radius[k] = L.border_radius[k];
        D.color[k] = L.color[k];
        D.border_color[k] = L.border_color[k];
        D.crop[k] = L.crop[k];
    }
    D.type = L.type;
    D.border_width = L.border_width;
    D.blur = L.blur_radius;
    D.masks_off = mo;
    D.masks_len = L.masks_len > SMR_MAX_MASKS ? SMR_MAX_MASKS : L.masks_len;
    for (u32 m = 0; m < D.masks_len; m++) {
        const smr_mask &K = L.masks[m];
        DevMask &DM = hm[mo++];
        memset(&DM, 0, sizeof(DM));
        for (int k = 0; k < 4; k++) DM.radius[k] = K.radius[k];
        DM.top = K.top; DM.left = K.left; DM.width = K.width; DM.height = K.height;
        // solid region of smoothstep(-.5, .5, -sdf): the rect inset by .5 minus the four corner squares of side max radius
        // (+ rounding slack unless every quantity is a multiple of 1/2 below 2^15: then each f32 operation of the SDF is exact)
        const float mr = fmaxf(fmaxf(K.radius[0], K.radius[1]), fmaxf(K.radius[2], K.radius[3]));
        auto hi = [](float v) { return v * 2.0f == floorf(v * 2.0f) && fabsf(v) < 32768.0f; };
        const bool mexact = hi(K.radius[0]) && hi(K.radius[1]) && hi(K.radius[2]) && hi(K.radius[3]) && hi(K.left) && hi(K.top) &&
                            hi(K.width) && hi(K.height);
        const float mslack = mexact ? 0.0f : 0.015625f;
        DM.inset = 0.5f + mslack;
        DM.corner = (mr + mslack > DM.inset) ? mr + mslack : 0.0f;
    }
    float qleft = L.left, qtop = L.top, qw = L.width, qh = L.height;
    if (L.type == 2) {  // box shadow quad grown by blur on each side (apply_layouts.wgsl:216-229)
        qleft = L.left - L.blur_radius; qtop = L.top - L.blur_radius;
        qw = L.width + 2.0f * L.blur_radius; qh = L.height + 2.0f * L.blur_radius;
    }
    D.qw = qw; D.qh = qh;
    D.cx = qleft + qw / 2.0f; D.cy = qtop + qh / 2.0f;
    float ang = L.rotation_degrees * DEG;
    D.cs = cosf(ang); D.sn = sinf(ang);
    D.src_kind = 0;
    D.tex_w = 1; D.tex_h = 1;
    D.src_index = -1;
    if (L.type == 0 && L.source_index < n_sources && src_kind[L.source_index] != 0) {
        D.src = src_views[L.source_index];
        D.src_kind = src_kind[L.source_index];
        D.tex_w = D.src.w; D.tex_h = D.src.h;
        D.src_index = (int)L.source_index;
    }
    D.rqw = 1.0f / D.qw; D.rqh = 1.0f / D.qh;
    D.rtw = 1.0f / (float)D.tex_w; D.rth = 1.0f / (float)D.tex_h;
    // ---- classification helpers for the fused compose kernel
    D.flags = (D.cs == 1.0f && D.sn == 0.0f) ? DL_UNROTATED : 0;
    // Solid region: inside the rect inset by m >= radius on every side the SDF is <= -m, i.e. edge_distance >= m.
    // m must also reach the point where every smoothstep saturates at exactly 1:
    //   no border: smoothstep(-.5,.5,ed) -> ed >= .5;  texture border: smoothstep(bw-.5,bw+.5,ed) -> ed >= bw+.5;
    //   colour border: smoothstep(bw,bw+1,ed) -> ed >= bw+1;  shadow: smoothstep(-b/2,b/2,ed) -> ed >= b/2.
    float rmax = fmaxf(fmaxf(L.border_radius[0], L.border_radius[1]), fmaxf(L.border_radius[2], L.border_radius[3]));
    float need = 0.5f;
    if (L.type == 2) need = fmaxf(L.blur_radius / 2.0f, 0.5f);  // >= .5 keeps the region inside the half-open quad coverage
    else if (L.border_width >= 1.0f) need = L.border_width + (L.type == 0 ? 0.5f : 1.0f);
    // Outside the four corner squares (side = max radius) the SDF is the plain distance to the nearest straight edge
    // (smr_layout_dev.h, rect_solid_box), so only those squares and the `need` band along the edges are not solid.
    auto half_int = [](float v) { return v * 2.0f == floorf(v * 2.0f) && fabsf(v) < 32768.0f; };
    const bool exact = half_int(L.border_radius[0]) && half_int(L.border_radius[1]) && half_int(L.border_radius[2]) &&
                       half_int(L.border_radius[3]) && half_int(L.left) && half_int(L.top) && half_int(L.width) && half_int(L.height) &&
                       half_int(need) && half_int(qleft) && half_int(qw) && half_int(qtop) && half_int(qh);
    const float slack = exact ? 0.0f : 0.015625f;  // 1/64 px for f32 rounding in the SDF
    D.inset = need + slack;
    D.corner = (rmax + slack > D.inset) ? rmax + slack : 0.0f;
    if (L.type == 0 && D.src_kind != 0 && (D.flags & DL_UNROTATED) && L.crop[0] == 0.0f && L.crop[1] == 0.0f &&
        L.crop[2] == (float)D.tex_w && L.crop[3] == (float)D.tex_h && L.width == (float)D.tex_w && L.height == (float)D.tex_h &&
        L.left == floorf(L.left) && L.top == floorf(L.top) && fabsf(L.left) < 65536.0f && fabsf(L.top) < 65536.0f) {
        D.flags |= DL_ALIGNED;
        D.ix = (int)L.left;
        D.iy = (int)L.top;
    }
    if (L.type != 0 && L.color[3] == 1.0f) {
        D.flags |= DL_COLOR_OPAQUE;
                auto enc = [&](float x) -> u32 {
            if (srgb) {
                if (!(x > 0.0f)) return 0u;
                return (u32)(std::upper_bound(thr + 1, thr + 256, x) - (thr + 1));  // #{i in 1..255 : thr[i] <= x}
            }
            x = !(x > 0.0f) ? 0.0f : (x > 1.0f ? 1.0f : x);
            return (u32)(int)(x * 255.0f + 0.5f);
        };
        D.solid_px = enc(L.color[0]) | (enc(L.color[1]) << 8) | (enc(L.color[2]) << 16) | (255u << 24);
    }
    const bool finite = std::isfinite(qleft) && std::isfinite(qtop) && std::isfinite(qw) && std::isfinite(qh) && std::isfinite(D.cs) &&
                        std::isfinite(D.sn);
    if (!(qw > 0.0f) || !(qh > 0.0f) || L.type > 2 || !finite) {
        // (a quad with a NaN / infinite corner rasterises to nothing; it must not reach the float -> int conversions below)
        D.bx0 = D.by0 = 0; D.bx1 = D.by1 = -1;  // never binned
    } else if (D.flags & DL_UNROTATED) {
        // pixel x is covered iff qleft <= x + .5 < qleft + qw (layout_covers), i.e. qleft - .5 <= x < qleft + qw - .5.
        // Tight bounds matter: a one-pixel margin makes every neighbour of a tile-aligned rect "touch" the next tile column.
        // 1/64 px of slack unless the quad is on half-integers (then the coverage arithmetic is exact in f32).
        const float s = (half_int(qleft) && half_int(qtop) && half_int(qw) && half_int(qh)) ? 0.0f : 0.015625f;
        auto clampi_h = [](float v, int lo, int hi) { return v < (float)lo ? lo : (v > (float)hi ? hi : (int)v); };
        D.bx0 = clampi_h(ceilf(qleft - 0.5f - s), 0, out_w);
        D.bx1 = clampi_h(ceilf(qleft + qw - 0.5f + s), 0, out_w);
        D.by0 = clampi_h(ceilf(qtop - 0.5f - s), 0, out_h);
        D.by1 = clampi_h(ceilf(qtop + qh - 0.5f + s), 0, out_h);
    } else {
        float ex = fabsf(D.cs) * qw / 2.0f + fabsf(D.sn) * qh / 2.0f;
        float ey = fabsf(D.sn) * qw / 2.0f + fabsf(D.cs) * qh / 2.0f;
        auto clampi_h = [](float v, int lo, int hi) { return v < (float)lo ? lo : (v > (float)hi ? hi : (int)v); };
        D.bx0 = clampi_h(floorf(D.cx - ex - 1.0f), 0, out_w);
        D.bx1 = clampi_h(ceilf(D.cx + ex + 1.0f), 0, out_w);
        D.by0 = clampi_h(floorf(D.cy - ey - 1.0f), 0, out_h);
        D.by1 = clampi_h(ceilf(D.cy + ey + 1.0f), 0, out_h);
    }

}

int smr_pack_layouts(smr_ctx *ctx, const smr_layout *layouts, u32 n, const SurfView *src_views, const int *src_kind,
                     u32 n_sources, int out_w, int out_h, size_t extra_bytes, PackedLayouts *out);
int smr_pack_commit(smr_ctx *ctx, PackedLayouts *p);
// launches the general compositor kernel over a committed parameter pack
int smr_launch_apply_layouts(smr_ctx *ctx, smr_surface *target, const PackedLayouts *p);
int smr_pack_done(smr_ctx *ctx, PackedLayouts *p);

#ifdef __HIPCC__

// Cooperative binning: set bit i when layout i's bounding box touches the tile.
__device__ __forceinline__ void bin_layouts(u32 *s_bits, const DevLayout *__restrict__ layouts, int n, int x0, int y0, int x1,
                                            int y1, int tid, int nthreads) {
    if (tid < MAX_LAYOUT_WORDS) s_bits[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += nthreads) {
        const DevLayout &L = layouts[i];
        if (L.bx0 < x1 && L.bx1 > x0 && L.by0 < y1 && L.by1 > y0) atomicOr(&s_bits[i >> 5], 1u << (i & 31));
    }
    __syncthreads();
}

__device__ __forceinline__ float smoothstepf(float e0, float e1, float x) {
    if (e0 == e1) return x >= e1 ? 1.0f : 0.0f;  // degenerate edge (blur 0): step
    const float d = e1 - e0;
    float t = (d == 1.0f) ? (x - e0) : (x - e0) / d;  // v / 1 == v
    t = clampf(t, 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

// roundedRectSDF, apply_layouts.wgsl:246-256
__device__ __forceinline__ float rounded_rect_sdf(float dx, float dy, float sw, float sh, const float *radius) {
    float hx = sw / 2.0f, hy = sh / 2.0f;
    float rx, ry;
    if (dx < 0.0f) { rx = radius[0]; ry = radius[3]; } else { rx = radius[1]; ry = radius[2]; }
    float r = (dy < 0.0f) ? ry : rx;
    float qx = fabsf(dx) - hx + r, qy = fabsf(dy) - hy + r;
    float mx = qx > 0.0f ? qx : 0.0f, my = qy > 0.0f ? qy : 0.0f;
    float m = qx > qy ? qx : qy;
    float inner = m < 0.0f ? m : 0.0f;
    // length(max(q, 0)): with one component zero, sqrt(m*m + 0) == m exactly (correctly rounded sqrt of a rounded square)
    float len = (my == 0.0f) ? mx : ((mx == 0.0f) ? my : sqrtf(mx * mx + my * my));
    return inner + len - r;
}

// Is pixel (px,py) inside layout L's quad?  Outputs the interpolated varyings.
__device__ __forceinline__ bool layout_covers(const DevLayout &L, int px, int py, float &fx, float &fy, float &lx, float &ly) {
    fx = (float)px + 0.5f;
    fy = (float)py + 0.5f;
    float dx = fx - L.cx, dy = -(fy - L.cy);
    if (L.flags & DL_UNROTATED) {  // cs == 1, sn == 0: 1*dx + 0*dy == dx exactly
        lx = dx;
        ly = dy;
    } else {
        lx = L.cs * dx + L.sn * dy;
        ly = -L.sn * dx + L.cs * dy;
    }
    if (!(lx >= -L.qw / 2.0f && lx < L.qw / 2.0f)) return false;
    if (!(-ly >= -L.qh / 2.0f && -ly < L.qh / 2.0f)) return false;
    return true;
}

// Solid region of a rounded rect for the box of pixel centres [cx0,cx1] x [cy0,cy1].
// roundedRectSDF with q = |d| - half + r:  outside the four r x r corner squares at least one of q.x, q.y is <= 0, the
// length() term collapses to the other component (or 0) and the SDF is the plain distance to the nearest straight edge:
// edge_distance = min(half_w - |dx|, half_h - |dy|).  So inside the rect inset by `inset` (the distance at which every
// smoothstep of the fragment stage saturates at exactly 1) and outside the corner squares the fragment is the base value.
// Both carry 1/64 px of slack for f32 rounding unless every quantity is exactly representable (host: smr_pack_layouts).
__device__ __forceinline__ bool rect_solid_box(float left, float top, float width, float height, float inset, float corner,
                                               float cx0, float cy0, float cx1, float cy1) {
    if (!(left + inset <= cx0 && cx1 <= left + width - inset && top + inset <= cy0 && cy1 <= top + height - inset)) return false;
    if (corner > 0.0f) {
        const bool near_x = cx0 < left + corner || cx1 > left + width - corner;
        const bool near_y = cy0 < top + corner || cy1 > top + height - corner;
        if (near_x && near_y) return false;
    }
    return true;
}

// Does the axis-aligned box of pixel centres [cx0,cx1] x [cy0,cy1] lie in the solid region of L and of all its masks?
__device__ __forceinline__ bool layout_solid_box(const DevLayout &L, const DevMask *__restrict__ masks, float cx0, float cy0,
                                                 float cx1, float cy1) {
    if (!(L.flags & DL_UNROTATED)) return false;
    bool solid = rect_solid_box(L.left, L.top, L.width, L.height, L.inset, L.corner, cx0, cy0, cx1, cy1);
    for (u32 m = 0; solid && m < L.masks_len; m++) {
        const DevMask &K = masks[L.masks_off + m];
        solid = rect_solid_box(K.left, K.top, K.width, K.height, K.inset, K.corner, cx0, cy0, cx1, cy1);
    }
    return solid;
}

__device__ __forceinline__ bool layout_base_opaque(const DevLayout &L) {
    return (L.type == 0) ? (L.src_kind == 2) : ((L.flags & DL_COLOR_OPAQUE) != 0);
}

// Fragment stage (apply_layouts.wgsl:258-377).  `sample` is fetched by the caller for texture layouts.
__device__ __forceinline__ float4 layout_fragment(const DevLayout &L, const DevMask *__restrict__ masks, float fx, float fy,
                                                  float lx, float ly, float4 sample) {
    float mask_alpha = 1.0f;
    for (u32 i = 0; i < L.masks_len; i++) {
        const DevMask &m = masks[L.masks_off + i];
        // inside the mask's solid region the factor is exactly 1
        if (rect_solid_box(m.left, m.top, m.width, m.height, m.inset, m.corner, fx, fy, fx, fy)) continue;
        float dx = m.left + (m.width / 2.0f) - fx;
        float dy = m.top + (m.height / 2.0f) - fy;
        float dist = rounded_rect_sdf(dx, dy, m.width, m.height, m.radius);
        mask_alpha = mask_alpha * smoothstepf(-0.5f, 0.5f, -dist);
    }
    float edge_distance = -rounded_rect_sdf(lx, ly, L.width, L.height, L.radius);
    const float bw = L.border_width;
    float4 o;
    if (L.type == 0) {
        if (bw < 1.0f) {
            float ca = smoothstepf(-0.5f, 0.5f, edge_distance);
            o = make_float4(sample.x * ca * mask_alpha, sample.y * ca * mask_alpha, sample.z * ca * mask_alpha,
                            sample.w * ca * mask_alpha);
        } else if (mask_alpha < 0.01f) {
            o = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (edge_distance > bw / 2.0f) {
            float ba = smoothstepf(bw - 0.5f, bw + 0.5f, edge_distance);
            float ia = 1.0f - ba;
            o = make_float4((L.border_color[0] * ia + sample.x * ba) * mask_alpha, (L.border_color[1] * ia + sample.y * ba) * mask_alpha,
                            (L.border_color[2] * ia + sample.z * ba) * mask_alpha, (L.border_color[3] * ia + sample.w * ba) * mask_alpha);
        } else {
            float ca = smoothstepf(-0.5f, 0.5f, edge_distance);
            o = make_float4(L.border_color[0] * ca * mask_alpha, L.border_color[1] * ca * mask_alpha,
                            L.border_color[2] * ca * mask_alpha, L.border_color[3] * ca * mask_alpha);
        }
    } else if (L.type == 1) {
        if (bw < 1.0f) {
            float ca = smoothstepf(-0.5f, 0.5f, edge_distance);
            o = make_float4(L.color[0] * ca * mask_alpha, L.color[1] * ca * mask_alpha, L.color[2] * ca * mask_alpha,
                            L.color[3] * ca * mask_alpha);
        } else if (edge_distance > bw / 2.0f) {
            float ba = smoothstepf(bw, bw + 1.0f, edge_distance);
            float ia = 1.0f - ba;
            o = make_float4((L.border_color[0] * ia + L.color[0] * ba) * mask_alpha, (L.border_color[1] * ia + L.color[1] * ba) * mask_alpha,
                            (L.border_color[2] * ia + L.color[2] * ba) * mask_alpha, (L.border_color[3] * ia + L.color[3] * ba) * mask_alpha);
        } else {
            float ca = smoothstepf(-0.5f, 0.5f, edge_distance);
            o = make_float4(L.border_color[0] * ca * mask_alpha, L.border_color[1] * ca * mask_alpha,
                            L.border_color[2] * ca * mask_alpha, L.border_color[3] * ca * mask_alpha);
        }
    } else {
        float ba = smoothstepf(-L.blur / 2.0f, L.blur / 2.0f, edge_distance) * mask_alpha;
        o = make_float4(L.color[0] * ba, L.color[1] * ba, L.color[2] * ba, L.color[3] * ba);
    }
    return o;
}

// PREMULTIPLIED_ALPHA_BLENDING (wgpu/common_pipeline.rs:125) onto an RGBA8 target texel,
// re-encoded the way the render-target store does (sRGB in GpuOptimized mode).
__device__ __forceinline__ u32 blend_store(u32 acc, float4 frag, int srgb, const float *__restrict__ dec,
                                           const float *__restrict__ thr) {
    // (a zero fragment leaves the bytes unchanged: encode(decode(b) * 1 + 0) == b, so no early-out is needed)
    const float inv = 1.0f - frag.w;
    const u32 r8 = acc & 0xff, g8 = (acc >> 8) & 0xff, b8 = (acc >> 16) & 0xff, a8 = acc >> 24;
    u32 r, g, b;
    if (srgb) {
        r = srgb_encode8(frag.x + dec[r8] * inv, thr);
        g = srgb_encode8(frag.y + dec[g8] * inv, thr);
        b = srgb_encode8(frag.z + dec[b8] * inv, thr);
    } else {
        r = unorm8(frag.x + ((float)r8 / 255.0f) * inv);
        g = unorm8(frag.y + ((float)g8 / 255.0f) * inv);
        b = unorm8(frag.z + ((float)b8 / 255.0f) * inv);
    }
    const u32 a = unorm8(frag.w + ((float)a8 / 255.0f) * inv);
    return r | (g << 8) | (b << 16) | (a << 24);
}

// textureSample of a texture layout at the interpolated tex_coords (vertex stage crop transform, wgsl:159-172)
__device__ __forceinline__ float4 layout_texture_sample(const DevLayout &L, int px, int py, float lx, float ly, int srgb,
                                                        const float *__restrict__ dec) {
    if (L.src_kind == 0) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int pxi = srgb ? PXI_RGBA8_SRGB : PXI_RGBA8_UNORM;
    if (L.flags & DL_ALIGNED) {
        // 1:1 blit on whole texels: the sample position is a texel centre, bilinear weights are exactly (1,0)
        return load_texel(L.src, pxi, clampi(px - L.ix, 0, L.tex_w - 1), clampi(py - L.iy, 0, L.tex_h - 1), dec);
    }
    float u01 = div_cr(lx, L.qw, L.rqw) + 0.5f, v01 = 0.5f - div_cr(ly, L.qh, L.rqh);
    float tu = div_cr(L.crop[1] + u01 * L.crop[2], (float)L.tex_w, L.rtw);
    float tv = div_cr(L.crop[0] + v01 * L.crop[3], (float)L.tex_h, L.rth);
    return sample_rgba_bilinear(L.src, pxi, tu, tv, dec);
}

// One layout applied to one pixel: coverage, varyings, texture fetch, fragment, blend.
__device__ __forceinline__ u32 composite_layout(u32 acc, const DevLayout &L, const DevMask *__restrict__ masks, int px, int py,
                                                int srgb, const float *__restrict__ dec, const float *__restrict__ thr) {
    float fx, fy, lx, ly;
    if (!layout_covers(L, px, py, fx, fy, lx, ly)) return acc;
    float4 frag;
    if (layout_solid_box(L, masks, fx, fy, fx, fy)) {
        // every smoothstep of the fragment stage is exactly 1 here: the fragment is the base value
        frag = (L.type == 0) ? layout_texture_sample(L, px, py, lx, ly, srgb, dec)
                             : make_float4(L.color[0], L.color[1], L.color[2], L.color[3]);
    } else {
        float4 sample = make_float4(0.f, 0.f, 0.f, 0.f);
        if (L.type == 0) sample = layout_texture_sample(L, px, py, lx, ly, srgb, dec);
        frag = layout_fragment(L, masks, fx, fy, lx, ly, sample);
    }
    return blend_store(acc, frag, srgb, dec, thr);
}

// The same for a pixel known to lie in the layout's solid region.
__device__ __forceinline__ u32 composite_layout_solid(u32 acc, const DevLayout &L, int px, int py, int srgb,
                                                      const float *__restrict__ dec, const float *__restrict__ thr) {
    float4 frag;
    if (L.type == 0) {
        float fx, fy, lx, ly;
        layout_covers(L, px, py, fx, fy, lx, ly);
        frag = layout_texture_sample(L, px, py, lx, ly, srgb, dec);
    } else {
        frag = make_float4(L.color[0], L.color[1], L.color[2], L.color[3]);
    }
    return blend_store(acc, frag, srgb, dec, thr);
}

// The filter and store of a bilinear sample of an opaque RGBA8 texture onto a cleared pixel, given the footprint's four texels and the
// 8-bit sub-texel weights of the second column / row: sample_rgba_bilinear's operations in their order on the three colour channels
// (alpha is exactly 1: every texel's is, and 8-bit weights sum to 1 exactly), then the render-target store.
__device__ __forceinline__ u32 filter_opaque_quad(u32 ta, u32 tb, u32 tc, u32 td, float fx, float fy, int srgb, const float *__restrict__ dec,
                                                  const float *__restrict__ thr) {
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    u32 out = 0xff000000u;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const u32 ba = (ta >> (8 * ch)) & 0xffu, bb = (tb >> (8 * ch)) & 0xffu, bc = (tc >> (8 * ch)) & 0xffu, bd = (td >> (8 * ch)) & 0xffu;
        float a, b, c, d;
        if (srgb) { a = dec[ba]; b = dec[bb]; c = dec[bc]; d = dec[bd]; }
        else { a = (float)ba / 255.0f; b = (float)bb / 255.0f; c = (float)bc / 255.0f; d = (float)bd / 255.0f; }
        const float o = (a * gx + b * fx) * gy + (c * gx + d * fx) * fy;
        out |= (srgb ? srgb_encode8(o, thr) : unorm8(o)) << (8 * ch);
    }
    return out;
}

// composite_layout_solid for the one case k_classify_tiles calls TC_SAMPLED: an opaque (src_kind 2) unrotated texture layer that is not
// a 1:1 blit, onto a cleared pixel, in the layer's solid region.  Every texel's alpha is 1, the bilinear weights of a sample sum to 1
// exactly (8-bit sub-texel fractions: 1 - f is exact), so the fragment's alpha is exactly 1, the blend keeps nothing of the
// destination (dst * 0) and the stored alpha is 255: only the three colour channels are fetched, filtered — the operations of
// sample_rgba_bilinear in their order — and encoded.
__device__ __forceinline__ u32 composite_sampled_opaque(const DevLayout &L, int px, int py, int srgb, const float *__restrict__ dec,
                                                        const float *__restrict__ thr) {
    float fx_, fy_, lx, ly;
    layout_covers(L, px, py, fx_, fy_, lx, ly);
    const float u01 = div_cr(lx, L.qw, L.rqw) + 0.5f, v01 = 0.5f - div_cr(ly, L.qh, L.rqh);
    const float tu = div_cr(L.crop[1] + u01 * L.crop[2], (float)L.tex_w, L.rtw);
    const float tv = div_cr(L.crop[0] + v01 * L.crop[3], (float)L.tex_h, L.rth);
    const SurfView &s = L.src;
    const float sx = tu * (float)s.w - 0.5f, sy = tv * (float)s.h - 0.5f;
    const float fx0 = floorf(sx), fy0 = floorf(sy);
    const float fx = subtexel(sx - fx0), fy = subtexel(sy - fy0);
    const int x0 = clampi((int)fx0, 0, s.w - 1), x1 = clampi((int)fx0 + 1, 0, s.w - 1);
    const int y0 = clampi((int)fy0, 0, s.h - 1), y1 = clampi((int)fy0 + 1, 0, s.h - 1);
    // (row offsets as 24-bit multiplies: rows and pitches are below 2^24, a surface below 4 GiB — a 32 x 32 multiply is quarter rate)
    const u32 o0 = (u32)__umul24((u32)y0, s.pitch), o1 = (u32)__umul24((u32)y1, s.pitch);
    const u32 ta = g_ld_u32(s.ptr + (o0 + 4u * (u32)x0)), tb = g_ld_u32(s.ptr + (o0 + 4u * (u32)x1));  // (global loads: smr_internal.h)
    const u32 tc = g_ld_u32(s.ptr + (o1 + 4u * (u32)x0)), td = g_ld_u32(s.ptr + (o1 + 4u * (u32)x1));
    return filter_opaque_quad(ta, tb, tc, td, fx, fy, srgb, dec, thr);
}

// The same for a 4 x 2 block of pixels (columns px0 .. px0 + 3, rows py0, py0 + 1) of such a tile.  The layer is unrotated, so everything a
// sample position's x half is made of — varying, texture coordinate, texel pair, 8-bit sub-texel weight — depends on the pixel's column
// alone and the y half on its row alone: four column records and two row records serve the eight pixels (per pixel the coordinate
// arithmetic was done twice over: ~40 of its ~100 vector instructions).  The expressions are composite_sampled_opaque's, value for value.
struct SampledAxis {
    u32 o0, o1;  // byte offsets of the two texels (x: 4 * column; y: row * pitch)
    float f, g;  // weight of the second texel (8-bit sub-texel fraction) and 1 - f
};
__device__ __forceinline__ SampledAxis sampled_axis_x(const DevLayout &L, int px) {
    const float fx = (float)px + 0.5f;
    const float lx = fx - L.cx;  // (DL_UNROTATED: 1 * dx + 0 * dy == dx exactly)
    const float u01 = div_cr(lx, L.qw, L.rqw) + 0.5f;
    const float tu = div_cr(L.crop[1] + u01 * L.crop[2], (float)L.tex_w, L.rtw);
    const float sx = tu * (float)L.src.w - 0.5f;
    const float fx0 = floorf(sx);
    SampledAxis a;
    a.f = subtexel(sx - fx0);
    a.g = 1.0f - a.f;
    a.o0 = 4u * (u32)clampi((int)fx0, 0, L.src.w - 1);
    a.o1 = 4u * (u32)clampi((int)fx0 + 1, 0, L.src.w - 1);
    return a;
}
__device__ __forceinline__ SampledAxis sampled_axis_y(const DevLayout &L, int py) {
    const float fy = (float)py + 0.5f;
    const float ly = -(fy - L.cy);
    const float v01 = 0.5f - div_cr(ly, L.qh, L.rqh);
    const float tv = div_cr(L.crop[0] + v01 * L.crop[3], (float)L.tex_h, L.rth);
    const float sy = tv * (float)L.src.h - 0.5f;
    const float fy0 = floorf(sy);
    SampledAxis a;
    a.f = subtexel(sy - fy0);
    a.g = 1.0f - a.f;
    // (row offsets as 24-bit multiplies: rows and pitches are below 2^24, a surface below 4 GiB)
    a.o0 = (u32)__umul24((u32)clampi((int)fy0, 0, L.src.h - 1), L.src.pitch);
    a.o1 = (u32)__umul24((u32)clampi((int)fy0 + 1, 0, L.src.h - 1), L.src.pitch);
    return a;
}
// (wave-uniform when the kernel runs on the device; the lane emulator has no wave votes and lets every lane choose for itself)
__device__ __forceinline__ bool sampled_block_all(bool v) {
#ifdef SMR_EMU
    return v;
#else
    return __ballot(!v) == 0ull;
#endif
}
__device__ __forceinline__ void composite_sampled_opaque_block(const DevLayout &L, int px0, int py0, int srgb, const float *__restrict__ dec,
                                                               const float *__restrict__ thr, u32 (&out)[8]) {
    SampledAxis X[4], Y[2];
#pragma unroll
    for (int q = 0; q < 4; q++) X[q] = sampled_axis_x(L, px0 + q);
#pragma unroll
    for (int r = 0; r < 2; r++) Y[r] = sampled_axis_y(L, py0 + r);
    const u8 *base = L.src.ptr;
    // A texture shown at its own size and a fractional position — every tile of a grid whose cells swap places — steps one texel per
    // pixel: neighbouring pixels' footprints share a column, the block's two rows share a texel row.  The block then needs 5 x 3 texels,
    // not 8 x 4: each is fetched and decoded once, and the horizontal half of the filter, (a * gx + b * fx), is evaluated once per texel
    // row and pixel column — the same operations on the same values as below, the middle row's used by both pixel rows.
    const bool unit = X[0].o1 == X[1].o0 && X[1].o1 == X[2].o0 && X[2].o1 == X[3].o0 && Y[0].o1 == Y[1].o0;
    if (sampled_block_all(unit)) {
        const u32 col[5] = {X[0].o0, X[1].o0, X[2].o0, X[3].o0, X[3].o1}, row[3] = {Y[0].o0, Y[0].o1, Y[1].o1};
        u32 t[3][5];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 5; c++) t[r][c] = g_ld_u32(base + (row[r] + col[c]));
        float h[3][4][3];  // [texel row][pixel column][channel]
#pragma unroll
        for (int r = 0; r < 3; r++) {
            float d[5][3];
#pragma unroll
            for (int c = 0; c < 5; c++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const u32 b = (t[r][c] >> (8 * ch)) & 0xffu;
                    d[c][ch] = srgb ? dec[b] : (float)b / 255.0f;
                }
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) h[r][q][ch] = d[q][ch] * X[q].g + d[q + 1][ch] * X[q].f;
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u32 o = 0xff000000u;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const float v = h[r][q][ch] * Y[r].g + h[r + 1][q][ch] * Y[r].f;
                    o |= (srgb ? srgb_encode8(v, thr) : unorm8(v)) << (8 * ch);
                }
                out[r * 4 + q] = o;
            }
        return;
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {  // (four pixels at a time: their texel fetches overlap)
        u32 ta[4], tb[4], tc[4], td[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ta[q] = g_ld_u32(base + (Y[r].o0 + X[q].o0)); tb[q] = g_ld_u32(base + (Y[r].o0 + X[q].o1));
            tc[q] = g_ld_u32(base + (Y[r].o1 + X[q].o0)); td[q] = g_ld_u32(base + (Y[r].o1 + X[q].o1));
        }
#pragma unroll
        for (int q = 0; q < 4; q++) out[r * 4 + q] = filter_opaque_quad(ta[q], tb[q], tc[q], td[q], X[q].f, Y[r].f, srgb, dec, thr);
    }
}

#endif  // __HIPCC__
