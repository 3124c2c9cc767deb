// smr_tables.h — the sRGB table block every context builds once (layout: smr_internal.h, srgb_encode8) and the f16-pair decode
// table of the matrix-core resamplers.  Header-only so that the lane emulator of tests/emu builds the very same tables.
#pragma once

#include <cmath>
#include <cstring>

#include "smr_internal.h"

static inline double smr_srgb_to_linear_f64(double c) {
    // smelter-render/src/wgpu/utils.rs:74-81
    if (c < 0.04045) return c / 12.92;
    return pow((c + 0.055) / 1.055, 2.4);
}

// tables[SMR_TABLE_FLOATS]: decode LUT | thr[257] | pad | encode estimate bytes; lut16[SMR_LUT16_WORDS]: (f16 hi | f16 lo << 16), lo = t - hi, then the encode buckets (smr_internal.h).
// Returns false if an estimate bucket straddles more than two codes (the one-step fix-up of srgb_encode8 would not suffice).
static inline bool smr_build_tables(float *tables, u32 *lut16) {
    memset(tables, 0, sizeof(float) * SMR_TABLE_FLOATS);
    for (int i = 0; i < 256; i++) tables[i] = (float)smr_srgb_to_linear_f64((double)i / 255.0);
    float *thr = tables + 256;
    thr[0] = -INFINITY;
    for (int i = 1; i < 256; i++) thr[i] = (float)smr_srgb_to_linear_f64(((double)i - 0.5) / 255.0);
    thr[256] = INFINITY;
    // encode estimate table: code of the lowest float of every (exponent, 7-bit mantissa) bucket in [2^-13, 1)
    u8 *enc = (u8 *)(thr + SMR_ENC_OFFSET_FROM_THR);
    auto code_of = [&](float x) {
        int c = 0;
        while (c < 255 && thr[c + 1] <= x) c++;
        return c;
    };
    for (u32 idx = 0; idx < SMR_ENC_ENTRIES; idx++) {
        u32 lo_bits = 0x39000000u + (idx << 16), hi_bits = lo_bits + 0xffffu;
        float lo, hi;
        memcpy(&lo, &lo_bits, 4);
        memcpy(&hi, &hi_bits, 4);
        const int cl = code_of(lo), chh = code_of(hi);
        if (chh - cl > 1) return false;
        enc[idx] = (u8)cl;
        // the same bucket for k_ingest_wave's one-gather encode: at most one threshold lies inside a bucket (just checked), strictly above
        // its lowest x; code = cl + (low 16 bits of x > `above` = that threshold's low 16 bits - 1).  The entry holds cl in its upper half
        // and 0xffff - above in its lower half: entry + (low 16 bits of x) carries into bit 16 exactly when the code steps up — one add
        // instead of a compare and a conditional increment, and the code sits in byte 2 of the sum, where v_perm_b32 picks it up (w_encode_sum)
        u32 above = 0xffffu;
        if (chh > cl) {
            u32 tb;
            memcpy(&tb, &thr[cl + 1], 4);
            if (tb <= lo_bits || tb > hi_bits) return false;
            above = tb - lo_bits - 1u;
        }
        lut16[256 + idx] = ((u32)cl << 16) | (0xffffu - above);
    }
    for (int i = 0; i < 256; i++) {
        const _Float16 hi = (_Float16)tables[i];
        const _Float16 lo = (_Float16)(tables[i] - (float)hi);
        u16 hb, lb;
        memcpy(&hb, &hi, 2);
        memcpy(&lb, &lo, 2);
        lut16[i] = (u32)hb | ((u32)lb << 16);
    }
    return true;
}
