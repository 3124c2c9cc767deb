// smr_internal.h — shared host/device definitions of libsmr_hip (gfx950 only).
#pragma once


#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "smr.h"

// Laboratory builds (-DSMR_LAB: tools/variant.sh, SMR_LAB=1 python -m smelter_amd.build): the A/B knobs read from the environment at context
// creation, the fused-conversion builds of k_ingest_wave (ingest implementation 5: within one code per stage, NOT within 1 LSB end to end on
// adversarial content, hence not in include/smr.h) and the kernels' ablation / timing hooks.  A product build has none of them.
constexpr int SMR_INGEST_LAB_FUSED = 5;
#ifdef SMR_LAB
constexpr bool SMR_LAB_BUILD = true;
#else
constexpr bool SMR_LAB_BUILD = false;
#endif

// Rows of the longest band of a tile that needs compositing (k_compose_output, smr_fused_compose.h: the workgroup's LDS pixel state).
#ifndef SMR_COMPOSE_BAND_ROWS
#define SMR_COMPOSE_BAND_ROWS 8
#endif


typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;

// sRGB table block (layout documented at srgb_encode8 below)
#define SMR_TABLE_FLOATS 932
#define SMR_ENC_OFFSET_FROM_THR 260
#define SMR_ENC_ENTRIES 1664
// lut16 block (device: ctx->d_lut16): [0,256) the decode table as (f16 hi | f16 lo << 16) | [256, 256 + SMR_ENC_ENTRIES) the encode table of the
// matrix-core resampler: per estimate bucket (code of its lowest x) << 16 | 0xffff - (offset of the one threshold inside it - 1; 0xffff = none)
#define SMR_LUT16_WORDS (256 + SMR_ENC_ENTRIES)

#define SMR_NUM_STAGES 8
enum {
    SMR_STAGE_INGEST = 0,
    SMR_STAGE_RESAMPLE = 1,
    SMR_STAGE_LAYOUT = 2,
    SMR_STAGE_OUTPUT = 3,
    SMR_STAGE_FUSED_INGEST = 4,
    SMR_STAGE_FUSED_COMPOSE = 5,
};

// Every allocation of the library ends with this many spare bytes: the 4:2:0 block converter reads (and ignores) up to a dword past the last
// chroma row's end when a row's bytes fill its pitch exactly (smr_convert.hip, conv_420_ok).
constexpr size_t SMR_SURFACE_TAIL = 16;

struct smr_surface {
    void *ptr = nullptr;
    size_t pitch = 0;
    u32 w = 0, h = 0, fmt = 0;
    bool owned = false;
    size_t capacity = 0;  // bytes behind ptr when owned (cached scratch surfaces are re-described in place while they fit)

};

// Device-side view of a surface.
struct SurfView {
    u8 *ptr;
    u32 pitch;
    int w, h;
};

// ---- device memory through GLOBAL instructions.  A pointer the compiler cannot trace back to a kernel argument — rebuilt from integers
// (v_readfirstlane round trips), loaded from a record in memory or LDS, selected between two bases — is a generic pointer, and generic
// accesses are FLAT instructions: they count in BOTH wait counters (a wait for an LDS table gather then also waits for every load still on its
// way from memory: the prefetch of the next block buys nothing), may return out of order (the compiler waits with vmcnt(0): a loop that stores
// waits for its own stores) and resolve their address space per access.  Every surface the kernels touch is device memory: these say so.
#if defined(__HIPCC__) && !defined(SMR_EMU)
typedef u32 g_u32x2 __attribute__((ext_vector_type(2)));
typedef u32 g_u32x3 __attribute__((ext_vector_type(3)));
typedef u32 g_u32x4 __attribute__((ext_vector_type(4)));
#define SMR_GLOBAL_PTR(T, p) ((__attribute__((address_space(1))) T *)(uintptr_t)(p))
__device__ __forceinline__ u32 g_ld_u32(const void *p) { return *SMR_GLOBAL_PTR(const u32, p); }
__device__ __forceinline__ uint2 g_ld_u32x2(const void *p) { const g_u32x2 v = *SMR_GLOBAL_PTR(const g_u32x2, p); return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint4 g_ld_u32x4(const void *p) { const g_u32x4 v = *SMR_GLOBAL_PTR(const g_u32x4, p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void g_st_u32(void *p, u32 v) { *SMR_GLOBAL_PTR(u32, p) = v; }
__device__ __forceinline__ void g_st_u32x3(void *p, u32 a, u32 b, u32 c) { g_u32x3 v; v.x = a; v.y = b; v.z = c; *SMR_GLOBAL_PTR(g_u32x3, p) = v; }
__device__ __forceinline__ void g_st_u32x4(void *p, uint4 q) { g_u32x4 v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w; *SMR_GLOBAL_PTR(g_u32x4, p) = v; }
#else  // (host code and the lane emulator: plain accesses)
static inline u32 g_ld_u32(const void *p) { return *(const u32 *)p; }
static inline uint2 g_ld_u32x2(const void *p) { return *(const uint2 *)p; }
static inline uint4 g_ld_u32x4(const void *p) { return *(const uint4 *)p; }
static inline void g_st_u32(void *p, u32 v) { *(u32 *)p = v; }
static inline void g_st_u32x3(void *p, u32 a, u32 b, u32 c) { ((u32 *)p)[0] = a; ((u32 *)p)[1] = b; ((u32 *)p)[2] = c; }
static inline void g_st_u32x4(void *p, uint4 q) { *(uint4 *)p = q; }
#endif

// one pinned-host + device staging slot of the per-call layout parameter ring
struct LayoutSlot {
    void *host = nullptr;
    void *dev = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    bool busy = false;
    bool unfenced = false;      // frames that REUSED this slot's device copy were queued after `done` was last recorded (smr_pack_done records no event for
                                // them: a marker packet per frame costs ~5 us on the stream): recycling the slot then waits for the stream instead
    size_t resident_bytes = 0;  // leading bytes of host that the device copy holds (0 = none): smr_pack_commit's reuse
    uint64_t last_reuse = 0;    // ctx->pack_clock when a frame last reused the device copy
};

struct StagePending {
    hipEvent_t a, b;
    int stage;
};

// Tile classes of one layout list (k_classify_tiles, smr_fused_compose.h), kept while the list repeats
struct TileClassMap {
    std::vector<uint8_t> key;      // the layout list (and output size) the classes were computed for
    bool ready = false;
    void *d_class = nullptr;       // TileClass per 128x16 output tile
    uint8_t *d_direct = nullptr;   // direct output: the layer wave A writes the tile for, 0xff = none
    void *d_list = nullptr;        // TileList: the tiles that need compositing
    size_t n = 0;
    uint32_t *h_count = nullptr;   // (pinned) the list's length, read back once per classification
    hipEvent_t count_ev = nullptr;
    bool count_pending = false, count_known = false;
    uint64_t class_serial = 0, count_serial = 0;  // classifications so far / the one the copy in flight belongs to
    int counter = 0;               // TileList::count[counter] is this classification's (a ring, zeroed as a whole when it wraps)
    uint64_t last_use = 0;
};

struct smr_ctx {
    int device = 0;
    u32 mode = 0;
    u32 max_layouts = SMR_DEFAULT_MAX_LAYOUTS;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // device tables: [0..255] sRGB decode, [256..512] encode thresholds (257 entries)
    float *d_tables = nullptr;
    float h_tables[SMR_TABLE_FLOATS] = {0};  // host copy (colour pre-encoding in the layout packer)

    // timers
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool profiling = false;
    std::vector<StagePending> pending;
    std::vector<hipEvent_t> event_pool;
    float stage_ms[SMR_NUM_STAGES] = {0};
    u32 stage_launches[SMR_NUM_STAGES] = {0};

    // per-call layout parameters travel through a ring of pinned staging slots
    std::vector<LayoutSlot> layout_ring;
    size_t layout_ring_next = 0;
    uint64_t pack_clock = 0;     // packs staged so far (LayoutSlot::last_reuse)
    int layout_last = -1;           // ring index of the pack committed last (its device copy is reused by an identical pack)
    bool no_pack_reuse = false;     // SMR_NO_PACK_REUSE (A/B, tests)
    unsigned long long pack_reused = 0;
    unsigned long long kernel_launches[SMR_KERNEL_COUNT_] = {};  // smr_debug_kernel_launches
    // scratch owned by the ctx (resampler intermediates, fused tiles)
    struct Scratch { void *ptr = nullptr; size_t bytes = 0; };
    std::vector<Scratch> scratch;  // indexed slots, grown on demand

    // size-keyed reusable surfaces (NodeTexture::ensure_size, state/node_texture.rs:22-42)
    std::vector<smr_surface *> surf_cache;

    // device-resident Lanczos weight tables, keyed by (scale, offset, n); see smr_fused.hip
    struct WeightTable {
        float scale = 0.f, offset = 0.f;
        int n = 0, taps = 0;
        void *dev = nullptr;  // int first[n]; float wsum[n]; float w[n * taps]
        size_t bytes = 0;
        uint64_t last_use = 0, last_call = 0;
    };
    std::vector<WeightTable> weight_tables;
    uint64_t weight_clock = 0;
    uint64_t weight_call = 0;  // id of the public call building jobs: tables it already handed out are never evicted
    // Lanczos weight bands in MFMA B-operand layout (smr_ingest_wave.h), keyed by (axis, scale, offset, n_dst, n_src)
    struct MfmaTable {
        float scale = 0.f, offset = 0.f;
        int n_dst = 0, n_src = 0, axis = 0, K = 0, max_span = 0;
        int ngm = 0;          // axis 0: column groups of the widest strip footprint (0 = not computed yet)
        void *dev = nullptr;  // int2 meta[n_tiles] (padded to 16 B); uint4 frag[n_tiles][K][64]
        size_t bytes = 0, meta_bytes = 0;
        uint64_t last_use = 0, last_call = 0;
    };
    std::vector<MfmaTable> mfma_tables;
    struct PendingBand { float scale, offset; int taps, n_dst, n_src, axis, K, n_tiles; void *meta, *frag; };
    std::vector<PendingBand> pending_bands;  // bands allocated by the current call, built in one launch before the kernel that reads them
    struct MfmaOccupancy { int kernel; size_t lds; int per_cu; };
    std::vector<MfmaOccupancy> mfma_occupancy;  // resident k_ingest_wave workgroups per CU, per (kernel build, LDS bytes)
    u32 *d_lut16 = nullptr;      // 256 x (f16 hi | f16 lo << 16) of the sRGB decode table
    u32 ingest_impl = 0;         // smr_ingest_impl
    bool wave_node82 = true;     // SMR_WAVE_NODE82=0 (A/B): node textures at scales around 2 on the generic build of k_ingest_wave instead of the <8, 2> class
    bool rgb12_cls82 = false;    // SMR_RGB12_CLS82=1 (A/B): RGB12 node textures for that class too
    int convert_wg_per_cu = 0;   // SMR_CONVERT_WG_PER_CU (laboratory builds): resident workgroups per CU of the persistent converter launch; 0 = by batch size (smr_convert.hip)
    u32 convert_lds_pad = 0;     // SMR_CONVERT_LDS_PAD (laboratory builds): extra dynamic LDS per workgroup of the block converter, i.e. a cap on its resident workgroups per CU
    u32 convert_impl = 0;        // smr_convert_impl (SMR_OPT_CONVERT_IMPL; SMR_CONVERT_GENERAL in the environment sets 1 at creation)
    bool mfma_attr_set = false;  // hipFuncSetAttribute is per device: kept per ctx, not per process
    bool wave_attr_set = false;
    bool valu_attr_set = false;
    int ingest_reserve_cus = -1; // SMR_INGEST_RESERVE_CUS (profiling), read once per ctx
    int ingest_wg_per_cu = 0;    // SMR_INGEST_WG_PER_CU (laboratory builds): cap on resident k_ingest_wave workgroups per CU, 0 = as many as fit
    bool shared_device = false;  // SMR_OPT_SHARED_DEVICE: other contexts' kernels run beside this one's (k_ingest_wave then takes two waves per SIMD)
    bool debug_ingest = false;   // SMR_DEBUG_INGEST: print the launch geometry
    int cu_count = 256;       // compute units of the device (MI355X: 256), sizes the fused ingest grid
    int fused_disabled = 0;   // SMR_OPT_FUSED_KERNELS = 0: the general pass-per-launch kernels instead of waves A / B (tests)
    int ablate = 0;           // SMR_ABLATE (laboratory builds: profiling experiments only)
    bool ablate_read = false;
    bool compose_select = true;  // SMR_COMPOSE_SELECT=0 (tests, profiling): no TC_SELECT tiles — seams between opaque 1:1 layers take the compositing path
    std::vector<u32> compose_bitmap;  // compose_predict's scratch
    int ingest_min_rows = 0;     // SMR_INGEST_MIN_ROWS (profiling): least tile rows per wave of k_ingest_wave (0: the default, 1)
    bool plane_source = false;   // SMR_OPT_PLANE_SOURCE (off by default: measured slower): 4:2:0 frames inside the matrix-core kernel's class windows are converted in the kernel itself (exactly: smr_convert_420.h's blocks through LDS) — no node texture in memory
    bool compact_nodes = true;   // SMR_OPT_COMPACT_NODES: node textures that only the matrix-core resampler reads are RGB12 (12 bytes per four pixels), not RGBA8
    bool direct_output = false;  // SMR_OPT_DIRECT_OUTPUT: wave A writes Y'CbCr for the compositor's copy tiles of a scene at rest
    std::vector<uint8_t> class_key_scratch;
    std::vector<TileClassMap> class_maps;  // tile classes of the last few layout lists (smr_fused.hip)
    uint64_t class_clock = 0;
    int force_tw = 0;         // SMR_INGEST_TW (tests / profiling): strip width of k_ingest_resample, 0 = automatic

    bool srgb() const { return mode == SMR_MODE_GPU_OPTIMIZED; }
};

// ctx-owned surface for `slot`, (re)allocated when the requested geometry changes; nullptr on error
smr_surface *smr_cached_surface(smr_ctx *ctx, size_t slot, u32 w, u32 h, u32 fmt);

int smr_fail(smr_ctx *ctx, int code, const char *fmt, ...);
int smr_check_hip(smr_ctx *ctx, hipError_t e, const char *what);
void *smr_scratch(smr_ctx *ctx, int slot, size_t bytes);  // nullptr on OOM (error set)
extern "C" int smr_validate_frame(smr_ctx *ctx, const smr_frame *f, const char *what);  // plane geometry / formats against the frame's format
// smr_frame_to_rgba for several frames at once (smr_convert.hip): planar 4:2:0 / 4:2:2 / 4:4:4 and NV12 frames share one launch per 16
// rgb12 (may be null): rgb12[i] != 0 -> nodes[i] is an R8 surface of 3 * width x height, the node texture as 12-byte groups of four pixels
// (smr_convert_420.h) — only for frames smr_conv_rgb12_ok() accepts, and only for nodes that nothing but the matrix-core resampler reads
int smr_frames_to_rgba_batch(smr_ctx *ctx, const smr_frame *const *in, smr_surface *const *nodes, u32 n, const u8 *rgb12 = nullptr);
bool smr_conv_rgb12_ok(const smr_ctx *ctx, const smr_frame *in);

// surface-cache slots (ctx->surf_cache), one table for every translation unit: disjoint for up to SMR_SLOT_MAX_LAYOUTS layouts and
// SMR_SLOT_MAX_SOURCES sources per call (smr_render_layouts clamps / rejects beyond)
constexpr size_t SMR_SLOT_MAX_LAYOUTS = 1024, SMR_SLOT_MAX_SOURCES = 1024;
constexpr size_t SMR_SLOT_TARGET = 0;
constexpr size_t SMR_SLOT_INGEST_NODE = 1;            // smr_ingest_resample's node texture
constexpr size_t SMR_SLOT_INGEST_NODE_RGB12 = 2;      // ... as RGB12
constexpr size_t SMR_SLOT_PRE_NODE = 3, SMR_SLOT_PRE_SCALED = 4;   // smr_frame_preprocess
constexpr size_t SMR_SLOT_TRANSPOSED_SINGLE = 8;      // smr_ingest_resample's own four (.. 11)
constexpr size_t SMR_SLOT_NODE0 = 16;                                              // + source index: RGBA8 node textures
constexpr size_t SMR_SLOT_NODE_RGB12_0 = SMR_SLOT_NODE0 + SMR_SLOT_MAX_SOURCES;    // + source index: RGB12 node textures
constexpr size_t SMR_SLOT_TILE0 = SMR_SLOT_NODE_RGB12_0 + SMR_SLOT_MAX_SOURCES;    // + layout index: resampled tiles
constexpr size_t SMR_SLOT_REDUCED0 = SMR_SLOT_TILE0 + SMR_SLOT_MAX_LAYOUTS;        // + layout index: box-reduced RGBA16F nodes
constexpr size_t SMR_SLOT_TRANSPOSED0 = SMR_SLOT_REDUCED0 + SMR_SLOT_MAX_LAYOUTS;  // + 4 * layout index: transposed planes / node and tile of a vertical-first plan
constexpr size_t SMR_SLOT_END = SMR_SLOT_TRANSPOSED0 + 4 * SMR_SLOT_MAX_LAYOUTS;
static_assert(SMR_SLOT_TRANSPOSED_SINGLE + 4 <= SMR_SLOT_NODE0 && SMR_SLOT_PRE_SCALED < SMR_SLOT_TRANSPOSED_SINGLE, "surface-cache slot ranges overlap");

// stage-timing helper: brackets kernel launches of one class with HIP events when profiling.
struct StageScope {
    smr_ctx *ctx;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    StageScope(smr_ctx *c, int s);
    ~StageScope();
};

// Every public entry point that touches the device binds the context's device first: a ctx may be used from any thread, and the
// calling thread's current HIP device is whatever that thread last set (smr.h: "each call does hipSetDevice").
#define SMR_ENTER(ctx)                                                                   \
    do {                                                                                 \
        if ((ctx) != nullptr) {                                                          \
            hipError_t e_dev__ = hipSetDevice((ctx)->device);                            \
            if (e_dev__ != hipSuccess) return smr_check_hip((ctx), e_dev__, "hipSetDevice"); \
        }                                                                                \
    } while (0)

#define SMR_HIP(ctx, call)                                                    \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return smr_check_hip((ctx), e__, #call);       \
    } while (0)

static inline SurfView view_of(const smr_surface *s) {
    SurfView v;
    v.ptr = (u8 *)s->ptr;
    v.pitch = (u32)s->pitch;
    v.w = (int)s->w;
    v.h = (int)s->h;
    return v;
}

static inline u32 bytes_per_px(u32 fmt) {
    switch (fmt) {
    case SMR_PX_RGBA8: return 4;
    case SMR_PX_RGBA16F: return 8;
    case SMR_PX_R8: return 1;
    case SMR_PX_RG8: return 2;
    default: return 0;
    }
}

// ------------------------------------------------------------------ device helpers
#ifdef __HIPCC__

// Pixel interpretation used by filter kernels (same numbering as the oracle).
enum { PXI_RGBA8_SRGB = 0, PXI_RGBA8_UNORM = 1, PXI_RGBA16F = 2 };

__device__ __forceinline__ float clampf(float x, float lo, float hi) {
    // WGSL clamp: min(max(x, lo), hi); NaN -> lo
    if (!(x > lo)) return lo;
    if (x > hi) return hi;
    return x;
}
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

__device__ __forceinline__ u32 unorm8(float x) {
    x = clampf(x, 0.0f, 1.0f);
    return (u32)(int)(x * 255.0f + 0.5f);
}

// sRGB encode as the monotone step function u8 = #{i : thr[i] <= x}.  Table block layout
// (SMR_TABLE_FLOATS floats, built in smr_ctx_create, copied to LDS by the hot kernels):
//   [0,256)    decode LUT            [256,513)  thr[0..256] (thr[0] = -inf, thr[256] = +inf)
//   [516,932)  enc: 1664 bytes, enc[((bits(x) - bits(2^-13)) >> 16)] = code of the bucket's lowest x
// A bucket (7 mantissa bits) straddles at most two thresholds (checked when the table is built),
// so the estimate needs at most two upward fix-up steps: exact, branch-free, no transcendental.
__device__ __forceinline__ u32 srgb_encode8(float x, const float *__restrict__ thr) {
    // branch-free (independent encodes overlap their table latencies): the estimate index is taken from x clamped
    // into [2^-13, 1); below 2^-13 (< thr[1]; also NaN, negatives) the bucket code is 0 and no threshold is reached,
    // at and above 1 the last bucket's code steps up to 255 through thr[255] (thr[256] = +inf ends the count).
    const u8 *enc = (const u8 *)(thr + SMR_ENC_OFFSET_FROM_THR);
    const float xc = fminf(fmaxf(x, 1.220703125e-4f), 0.99999994f);
    u32 c = enc[(__float_as_uint(xc) - 0x39000000u) >> 16];
    c += thr[c + 1] <= x ? 1u : 0u;
    return c;
}

// a / b, correctly rounded, from rb = RN(1/b): q0 = RN(a*rb); r = a - q0*b (exact, FMA); q = RN(q0 + r*rb)
// (Markstein; holds for normal operands unless b's significand is all ones).
__device__ __forceinline__ float div_cr(float a, float b, float rb) {
    float q0 = a * rb;
    float r = __builtin_fmaf(-q0, b, a);
    return __builtin_fmaf(r, rb, q0);
}

__device__ __forceinline__ float subtexel(float f) { return floorf(f * 256.0f + 0.5f) / 256.0f; }

__device__ __forceinline__ float4 load_texel(const SurfView &s, int pxi, int x, int y, const float *__restrict__ dec) {
    float4 o;
    if (pxi == PXI_RGBA16F) {
        const uint2 raw = *(const uint2 *)(s.ptr + (size_t)y * s.pitch + (size_t)x * 8);
        __half2 lo = *(const __half2 *)&raw.x, hi = *(const __half2 *)&raw.y;
        float2 a = __half22float2(lo), b = __half22float2(hi);
        o = make_float4(a.x, a.y, b.x, b.y);
    } else {
        const u32 raw = *(const u32 *)(s.ptr + (size_t)y * s.pitch + (size_t)x * 4);
        u32 r = raw & 0xff, g = (raw >> 8) & 0xff, b = (raw >> 16) & 0xff, a = raw >> 24;
        if (pxi == PXI_RGBA8_SRGB) {
            o.x = dec[r]; o.y = dec[g]; o.z = dec[b];
        } else {
            o.x = (float)r / 255.0f; o.y = (float)g / 255.0f; o.z = (float)b / 255.0f;
        }
        o.w = (float)a / 255.0f;
    }
    return o;
}

__device__ __forceinline__ void store_texel(const SurfView &s, int pxi, int x, int y, float4 v,
                                            const float *__restrict__ thr) {
    if (pxi == PXI_RGBA16F) {
        __half2 lo = __floats2half2_rn(v.x, v.y), hi = __floats2half2_rn(v.z, v.w);
        uint2 raw;
        raw.x = *(const u32 *)&lo;
        raw.y = *(const u32 *)&hi;
        *(uint2 *)(s.ptr + (size_t)y * s.pitch + (size_t)x * 8) = raw;
    } else {
        u32 r, g, b;
        if (pxi == PXI_RGBA8_SRGB) {
            r = srgb_encode8(v.x, thr); g = srgb_encode8(v.y, thr); b = srgb_encode8(v.z, thr);
        } else {
            r = unorm8(v.x); g = unorm8(v.y); b = unorm8(v.z);
        }
        u32 a = unorm8(v.w);
        *(u32 *)(s.ptr + (size_t)y * s.pitch + (size_t)x * 4) = r | (g << 8) | (b << 16) | (a << 24);
    }
}

// textureSample of one channel of an 8-bit plane (comps interleaved channels), bilinear,
// clamp-to-edge, 8-bit sub-texel weights.  Returns the unorm value.
__device__ __forceinline__ float sample_plane_bilinear(const SurfView &p, int comps, int c, float u, float v) {
    float sx = u * (float)p.w - 0.5f;
    float sy = v * (float)p.h - 0.5f;
    float fx0 = floorf(sx), fy0 = floorf(sy);
    float fx = subtexel(sx - fx0), fy = subtexel(sy - fy0);
    int x0 = clampi((int)fx0, 0, p.w - 1), x1 = clampi((int)fx0 + 1, 0, p.w - 1);
    int y0 = clampi((int)fy0, 0, p.h - 1), y1 = clampi((int)fy0 + 1, 0, p.h - 1);
    const u8 *r0 = p.ptr + (size_t)y0 * p.pitch, *r1 = p.ptr + (size_t)y1 * p.pitch;
    float a = (float)r0[x0 * comps + c] / 255.0f;
    float b = (float)r0[x1 * comps + c] / 255.0f;
    float cc = (float)r1[x0 * comps + c] / 255.0f;
    float d = (float)r1[x1 * comps + c] / 255.0f;
    float top = a * (1.0f - fx) + b * fx;
    float bot = cc * (1.0f - fx) + d * fx;
    return top * (1.0f - fy) + bot * fy;
}

// textureSample of an RGBA8 (or RGBA16F) surface, bilinear + clamp, texels decoded per `pxi`.
__device__ __forceinline__ float4 sample_rgba_bilinear(const SurfView &s, int pxi, float u, float v,
                                                       const float *__restrict__ dec) {
    float sx = u * (float)s.w - 0.5f, sy = v * (float)s.h - 0.5f;
    float fx0 = floorf(sx), fy0 = floorf(sy);
    float fx = subtexel(sx - fx0), fy = subtexel(sy - fy0);
    int x0 = clampi((int)fx0, 0, s.w - 1), x1 = clampi((int)fx0 + 1, 0, s.w - 1);
    int y0 = clampi((int)fy0, 0, s.h - 1), y1 = clampi((int)fy0 + 1, 0, s.h - 1);
    float4 a = load_texel(s, pxi, x0, y0, dec), b = load_texel(s, pxi, x1, y0, dec);
    float4 c = load_texel(s, pxi, x0, y1, dec), d = load_texel(s, pxi, x1, y1, dec);
    float4 o;
    float gx = 1.0f - fx, gy = 1.0f - fy;
    o.x = (a.x * gx + b.x * fx) * gy + (c.x * gx + d.x * fx) * fy;
    o.y = (a.y * gx + b.y * fx) * gy + (c.y * gx + d.y * fx) * fy;
    o.z = (a.z * gx + b.z * fx) * gy + (c.z * gx + d.z * fx) * fy;
    o.w = (a.w * gx + b.w * fx) * gy + (c.w * gx + d.w * fx) * fy;
    return o;
}

#endif  // __HIPCC__
