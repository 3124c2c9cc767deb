// smr_fused_compose.h — wave B of the hot path: k_compose_output (included by smr_fused.hip only).
//
// LayoutShader::render + RgbaToYuvConverter / RgbaToNv12Converter in one launch.  A 256-thread workgroup
// owns a 128x16 pixel tile; each thread a 4x2 pixel block (one u32 of Y per row, two chroma samples).
// The layout list is classified twice, both times with the exact "solid region" test of smr_layout_dev.h:
//   per tile   (one thread per layout): touch / solid / start = last solid layer whose base value is opaque;
//   per thread (its 4x2 block): the start search continues above the tile's start layer.
// Compositing begins at the start layer — every earlier layer is overwritten (dst * (1 - 1) == 0 exactly).
// A start layer that is a 1:1 texel-aligned opaque texture (the resampled video tile) is a byte copy; a tile
// that nothing else touches needs no tables and no blending arithmetic: texels -> Y'CbCr.
// Everything else runs the same per-pixel code as the general compositor (smr_layout_dev.h).
#pragma once

#include "smr_convert_dev.h"
#include "smr_layout_dev.h"

namespace {

constexpr int B_TILE_W = 128, B_TILE_H = 16;  // pixels; 32 x 8 threads, one 4x2 pixel block each
constexpr int B_MAX_LAYOUTS = 48;             // LDS-resident layout list (larger lists take the general compositor)
constexpr int B_MAX_MASKS = 96;
static_assert(sizeof(DevLayout) % 16 == 0 && sizeof(DevMask) % 16 == 0, "LDS copies move 16 B words");

__device__ __forceinline__ void classify_layouts(u32 *s_touch, u32 *s_solid, int *s_start, const DevLayout *__restrict__ layouts,
                                                 const DevMask *__restrict__ masks, int n, int x0, int y0, int x1, int y1, int tid,
                                                 int nthreads) {
    if (tid < MAX_LAYOUT_WORDS) { s_touch[tid] = 0; s_solid[tid] = 0; }
    if (tid == 0) *s_start = -1;
    __syncthreads();
    const float cx0 = (float)x0 + 0.5f, cx1 = (float)x1 - 0.5f, cy0 = (float)y0 + 0.5f, cy1 = (float)y1 - 0.5f;
    for (int i = tid; i < n; i += nthreads) {
        const DevLayout &L = layouts[i];
        if (!(L.bx0 < x1 && L.bx1 > x0 && L.by0 < y1 && L.by1 > y0)) continue;
        atomicOr(&s_touch[i >> 5], 1u << (i & 31));
        if (!layout_solid_box(L, masks, cx0, cy0, cx1, cy1)) continue;
        atomicOr(&s_solid[i >> 5], 1u << (i & 31));
        if (layout_base_opaque(L)) atomicMax(s_start, i);
    }
    __syncthreads();
}

// fills the 4x2 block from an opaque base layer (dst is irrelevant)
__device__ __forceinline__ void fill_from_base(u32 acc[8], const DevLayout &L, int px0, int py0, int srgb, const float *__restrict__ dec,
                                               const float *__restrict__ thr) {
    if (L.type != 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = L.solid_px;
    } else if (L.flags & DL_ALIGNED) {
        // 1:1 blit of an opaque texture: bilinear weights are exactly (1,0), decode -> encode is the identity
        const u8 *r0 = L.src.ptr + (size_t)(py0 - L.iy) * L.src.pitch + (size_t)(px0 - L.ix) * 4;
        const u8 *r1 = r0 + L.src.pitch;
        if ((((uintptr_t)r0) & 15) == 0 && (L.src.pitch & 15) == 0) {
            const uint4 t0 = *(const uint4 *)r0, t1 = *(const uint4 *)r1;
            acc[0] = t0.x; acc[1] = t0.y; acc[2] = t0.z; acc[3] = t0.w;
            acc[4] = t1.x; acc[5] = t1.y; acc[6] = t1.z; acc[7] = t1.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) { acc[c] = ((const u32 *)r0)[c]; acc[4 + c] = ((const u32 *)r1)[c]; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = composite_layout_solid(0u, L, px0 + (k & 3), py0 + (k >> 2), srgb, dec, thr);
    }
}

// NV = 0: planar Y,U,V (4:2:0); NV = 1: NV12 (Y + interleaved UV)
template <int NV>
__global__ __launch_bounds__(256) void k_compose_output(SurfView yp, SurfView up, SurfView vp, int W, int H,
                                                        const DevLayout *__restrict__ layouts_g, const DevMask *__restrict__ masks_g,
                                                        int n, int n_masks, int srgb_and_ablate, const float *__restrict__ tables) {
    __shared__ u32 s_touch[MAX_LAYOUT_WORDS], s_solid[MAX_LAYOUT_WORDS];
    __shared__ int s_start, s_general;
    __shared__ float s_tab[SMR_TABLE_FLOATS];
    // the whole layout list lives in LDS for the lifetime of the workgroup: one coalesced copy instead of a
    // dependent scalar-memory round trip per field per layer per wave
    __shared__ __attribute__((aligned(16))) DevLayout s_lay[B_MAX_LAYOUTS];
    __shared__ __attribute__((aligned(16))) DevMask s_mask[B_MAX_MASKS];
    {
        const uint4 *gl = (const uint4 *)layouts_g;
        uint4 *ll = (uint4 *)s_lay;
        for (int i = threadIdx.x; i < n * (int)(sizeof(DevLayout) / 16); i += 256) ll[i] = gl[i];
        const uint4 *gm = (const uint4 *)masks_g;
        uint4 *lm = (uint4 *)s_mask;
        for (int i = threadIdx.x; i < n_masks * (int)(sizeof(DevMask) / 16); i += 256) lm[i] = gm[i];
    }
    const DevLayout *layouts = s_lay;
    const DevMask *masks = s_mask;
    const int srgb = srgb_and_ablate & 1;
    const int ablate = srgb_and_ablate >> 8;  // profiling only (SMR_ABLATE bits 8..): 1 dispatch only, 2 classify only,
                                              // 4 no per-thread start search, 8 base layer only
    if (ablate & 1) return;
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * B_TILE_W, ty0 = blockIdx.y * B_TILE_H;
    if (tid == 0) s_general = 0;
    classify_layouts(s_touch, s_solid, &s_start, layouts, masks, n, tx0, ty0, min(tx0 + B_TILE_W, W), min(ty0 + B_TILE_H, H), tid, 256);
    const int start = s_start;
    // does any layer of this tile need blending arithmetic (and therefore the sRGB tables)?
    for (int i = tid; i < n; i += 256) {
        if (!((s_touch[i >> 5] >> (i & 31)) & 1u) || i < start) continue;
        const DevLayout &L = layouts[i];
        const bool copy_layer = i == start && (L.type != 0 || (L.flags & DL_ALIGNED));
        if (!copy_layer) s_general = 1;
    }
    __syncthreads();
    const bool general = s_general != 0;
    if (general) {
        for (int i = tid; i < SMR_TABLE_FLOATS; i += 256) s_tab[i] = tables[i];
        __syncthreads();
    }
    const float *dec = s_tab, *thr = s_tab + 256;
    if (ablate & 2) return;

    const int px0 = tx0 + 4 * (tid & 31), py0 = ty0 + 2 * (tid >> 5);
    if (px0 >= W || py0 >= H) return;  // W % 4 == 0, H % 2 == 0: a block is entirely inside or outside

    u32 acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // [row][col]: acc[r * 4 + c]
    const int words = (n + 31) >> 5;
    // ---- per-thread start: the topmost layer above the tile's start whose solid region contains this 4x2 block
    int start_t = start;
    if (general && !(ablate & 4)) {
        const float bx0 = (float)px0 + 0.5f, bx1 = (float)px0 + 3.5f, by0 = (float)py0 + 0.5f, by1 = (float)py0 + 1.5f;
        for (int wi = words - 1; wi >= (start < 0 ? 0 : (start >> 5)) && start_t == start; wi--) {
            u32 bits = s_touch[wi];
            if (start >= 0 && wi == (start >> 5)) bits &= ~((2u << (start & 31)) - 1u);  // strictly above start
            while (bits) {
                const int b = 31 - __builtin_clz(bits);
                bits &= ~(1u << b);
                const int li = (wi << 5) + b;
                const DevLayout &L = layouts[li];
                if (layout_base_opaque(L) && layout_solid_box(L, masks, bx0, by0, bx1, by1)) { start_t = li; break; }
            }
        }
    }
    for (int wi = start_t < 0 ? 0 : (start_t >> 5); wi < words; wi++) {
        u32 bits = s_touch[wi];
        if (start_t >= 0 && wi == (start_t >> 5)) bits &= ~((1u << (start_t & 31)) - 1u);
        const u32 solid_bits = s_solid[wi];
        while (bits) {
            const int b = __builtin_ctz(bits);
            const int li = (wi << 5) + b;
            bits &= bits - 1;
            const DevLayout &L = layouts[li];
            if (li == start_t) {
                fill_from_base(acc, L, px0, py0, srgb, dec, thr);
            } else if (ablate & 8) {
            } else if ((solid_bits >> b) & 1u) {
#pragma unroll
                for (int k = 0; k < 8; k++) acc[k] = composite_layout_solid(acc[k], L, px0 + (k & 3), py0 + (k >> 2), srgb, dec, thr);
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) acc[k] = composite_layout(acc[k], L, masks, px0 + (k & 3), py0 + (k >> 2), srgb, dec, thr);
            }
        }
    }

    // RGBA -> Y'CbCr on the raw (gamma-encoded) bytes: rgba_to_yuv.wgsl:26-54 / rgba_to_nv12.wgsl:24-52
    float4 c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = unpack_unorm(acc[k]);
    u32 yrow0 = 0, yrow1 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        yrow0 |= unorm8(yuv_component(c[k], 0)) << (8 * k);
        yrow1 |= unorm8(yuv_component(c[4 + k], 0)) << (8 * k);
    }
    *(u32 *)(yp.ptr + (size_t)py0 * yp.pitch + px0) = yrow0;
    *(u32 *)(yp.ptr + (size_t)(py0 + 1) * yp.pitch + px0) = yrow1;
    // chroma: the bilinear tap at the chroma texel centre = weights (1/2, 1/2) x (1/2, 1/2)
    const float fx = 0.5f, gx = 1.0f - fx, fy = 0.5f, gy = 1.0f - fy;
    u32 uv[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const float4 &p00 = c[2 * j], &p01 = c[2 * j + 1], &p10 = c[4 + 2 * j], &p11 = c[4 + 2 * j + 1];
        float4 m;
        m.x = (p00.x * gx + p01.x * fx) * gy + (p10.x * gx + p11.x * fx) * fy;
        m.y = (p00.y * gx + p01.y * fx) * gy + (p10.y * gx + p11.y * fx) * fy;
        m.z = (p00.z * gx + p01.z * fx) * gy + (p10.z * gx + p11.z * fx) * fy;
        m.w = 0.0f;
        uv[j][0] = unorm8(yuv_component(m, 1));
        uv[j][1] = unorm8(yuv_component(m, 2));
    }
    const int cx = px0 >> 1, cy = py0 >> 1;
    if (NV == 0) {
        *(u16 *)(up.ptr + (size_t)cy * up.pitch + cx) = (u16)(uv[0][0] | (uv[1][0] << 8));
        *(u16 *)(vp.ptr + (size_t)cy * vp.pitch + cx) = (u16)(uv[0][1] | (uv[1][1] << 8));
    } else {
        *(u32 *)(up.ptr + (size_t)cy * up.pitch + (size_t)cx * 2) = uv[0][0] | (uv[0][1] << 8) | (uv[1][0] << 16) | (uv[1][1] << 24);
    }
}

}  // namespace
