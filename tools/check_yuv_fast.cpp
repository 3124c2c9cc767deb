// tools/check_yuv_fast.cpp — exhaustive proof that the output conversion's FAST path (smr_convert_dev.h: yuv_fast_*) never
// stores a byte that differs from the reference sequence (rgba_to_yuv.wgsl:26-54 as yuv_byte() / unorm_of_byte() restate it)
// without raising its guard flag.
//
//   fast:   yuvfast::convert<plane>() of smelter_amd/csrc/smr_yuv_fast.h — THE header the kernels include —
//           x = fma(R, k_r, fma(G, k_g, fma(B, k_b, k_0 + delta)))      (R, G, B: the bytes — or, for chroma, the 2x2 block's byte sums)
//           byte = trunc(x);   flag = fract(x) < 2 delta   ->  flagged lanes recompute with the reference sequence
//   exact:  the f32 sequence of the WGSL pass, operation by operation (compiled here with -ffp-contract=off)
//
// Luma is a function of one pixel's three bytes: all 2^24 triples are checked.
// Chroma is a function of the three channels' 2x2 means m = ((a + b) + (c + d)) / 4 of byte / 255 values; m depends on the four bytes and their
// order only through a handful of f32 values per byte sum (rounding of the three additions).  The tool enumerates, per sum S in 0..1020, the
// SET of values m can take over all 256^4 ordered quadruples, then checks every combination (S_r, S_g, S_b) x (every m_r, m_g, m_b of the sets):
// the complete domain of the block conversion.  Prints the guard band actually needed (largest distance from a code boundary at which a
// largest deviation of the fast value from the sequence's value before truncation next to the guard band the kernels use.
//
// build:  g++ -O2 -fopenmp -ffp-contract=off -I smelter_amd/csrc -o /tmp/check_yuv_fast tools/check_yuv_fast.cpp      run: /tmp/check_yuv_fast   (~30 s on 8 cores)
// (tests/test_yuv_fast.py runs the luma proof and a slice of the chroma proof in the CPU suite: `check_yuv_fast quick`)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smr_yuv_fast.h"

#define YUV_FAST_DELTA (yuvfast::DELTA)

static float unorm_of_byte(uint32_t b) {
    const float a = (float)b;
    return fmaf(a, 1.0f / 255.0f, a * -1.1641532182693481e-10f);
}
static uint32_t yuv_byte(float r, float g, float b, int plane, float *pre) {
    float comp;
    if (plane == 0) {
        const float y = r * 0.2126f + g * 0.7152f + b * 0.0722f;
        comp = (y * 0.85882352941f) + (16.0f / 255.0f);
    } else if (plane == 1) {
        const float u = r * -0.1146f + g * -0.3854f + b * 0.5f;
        comp = ((u + 0.5f) * 0.87843137254f) + (16.0f / 255.0f);
    } else {
        const float v = r * 0.5f + g * -0.4542f + b * -0.0458f;
        comp = ((v + 0.5f) * 0.87843137254f) + (16.0f / 255.0f);
    }
    const float t = comp * 255.0f + 0.5f;
    if (pre) *pre = t;
    return (uint32_t)(int)t;
}

static inline uint32_t fast_convert(int plane, float r, float g, float b, bool *flag) {
    return plane == 0 ? yuvfast::convert<0>(r, g, b, flag) : plane == 1 ? yuvfast::convert<1>(r, g, b, flag) : yuvfast::convert<2>(r, g, b, flag);
}
static inline float fast_x(int plane, float r, float g, float b) {  // the value convert() truncates (for the deviation statistics only)
    const yuvfast::K k = yuvfast::constants(plane);
    return fmaf(r, k.kr, fmaf(g, k.kg, fmaf(b, k.kb, k.k0)));
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");  // the CPU suite: all of luma, every 16th red sum of chroma
    int bad = 0;
    // ------------------------------------------------------------------ luma: every (R, G, B)
    {
        const yuvfast::K F = yuvfast::constants(0);
        printf("luma   k = %.9g %.9g %.9g  k0 = %.9g\n", F.kr, F.kg, F.kb, F.k0);
        long long flagged = 0, mism_flagged = 0, mism_unflagged = 0;
        double maxdev = 0.0;
        float n[256];
        for (int i = 0; i < 256; i++) n[i] = unorm_of_byte((uint32_t)i);
#pragma omp parallel for reduction(+ : flagged, mism_flagged, mism_unflagged) reduction(max : maxdev)
        for (int r = 0; r < 256; r++)
            for (int g = 0; g < 256; g++)
                for (int b = 0; b < 256; b++) {
                    float pre;
                    const uint32_t want = yuv_byte(n[r], n[g], n[b], 0, &pre);
                    const float x = fast_x(0, (float)r, (float)g, (float)b);
                    bool flag;
                    const uint32_t got = fast_convert(0, (float)r, (float)g, (float)b, &flag);
                    const double dev = fabs(((double)x - YUV_FAST_DELTA) - (double)pre);
                    if (dev > maxdev) maxdev = dev;
                    flagged += flag;
                    if (got != want) {
                        if (flag) mism_flagged++; else mism_unflagged++;
                    }
                }
        printf("luma   2^24 triples: flagged %lld (%.4f %%), of them with another byte than the sequence's %lld, mismatches NOT flagged %lld; |fast - sequence| <= %.3g, guard band %.3g\n",
               flagged, 100.0 * (double)flagged / 16777216.0, mism_flagged, mism_unflagged, maxdev, YUV_FAST_DELTA);
        if (mism_unflagged) bad = 1;
    }
    // ------------------------------------------------------------------ chroma: the sets of block means per byte sum
    enum { NS = 1021, MAXSET = 24 };
    static float set[NS][MAXSET];
    static int nset[NS];
    {
        float n[256];
        for (int i = 0; i < 256; i++) n[i] = unorm_of_byte((uint32_t)i);
        memset(nset, 0, sizeof(nset));
        // (a + b) takes few distinct values per pair sum; enumerate pair sums first: P[s] = set of f32 values of n[a] + n[b] with a + b = s
        enum { NP = 511, MAXP = 16 };
        static float pset[NP][MAXP];
        static int npset[NP];
        memset(npset, 0, sizeof(npset));
        for (int a = 0; a < 256; a++)
            for (int b = 0; b < 256; b++) {
                const float v = n[a] + n[b];
                const int s = a + b;
                int k;
                for (k = 0; k < npset[s]; k++) if (pset[s][k] == v) break;
                if (k == npset[s]) {
                    if (npset[s] >= MAXP) { fprintf(stderr, "pair set overflow\n"); return 2; }
                    pset[s][npset[s]++] = v;
                }
            }
        int maxp = 0;
        for (int s = 0; s < NP; s++) if (npset[s] > maxp) maxp = npset[s];
        for (int s0 = 0; s0 < NP; s0++)
            for (int s1 = 0; s1 < NP; s1++)
                for (int i = 0; i < npset[s0]; i++)
                    for (int j = 0; j < npset[s1]; j++) {
                        const float m = (pset[s0][i] + pset[s1][j]) * 0.25f;
                        const int s = s0 + s1;
                        int k;
                        for (k = 0; k < nset[s]; k++) if (set[s][k] == m) break;
                        if (k == nset[s]) {
                            if (nset[s] >= MAXSET) { fprintf(stderr, "mean set overflow\n"); return 2; }
                            set[s][nset[s]++] = m;
                        }
                    }
        int maxs = 0; long long tot = 0;
        for (int s = 0; s < NS; s++) { if (nset[s] > maxs) maxs = nset[s]; tot += nset[s]; }
        printf("chroma the 2x2 mean of byte / 255 takes at most %d values per pair sum, %d per block sum (%.2f on average)\n", maxp, maxs, (double)tot / NS);
    }
    for (int plane = 1; plane <= 2; plane++) {
        const yuvfast::K F = yuvfast::constants(plane);
        printf("plane %d k = %.9g %.9g %.9g  k0 = %.9g\n", plane, F.kr, F.kg, F.kb, F.k0);
        long long total = 0, flagged = 0, mism_flagged = 0, mism_unflagged = 0;
        double maxdev = 0.0;
        const int step = quick ? 16 : 1;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total, flagged, mism_flagged, mism_unflagged) reduction(max : maxdev)
        for (int sr = 0; sr < NS; sr += step)
            for (int sg = 0; sg < NS; sg++)
                for (int sb = 0; sb < NS; sb++) {
                    const float x = fast_x(plane, (float)sr, (float)sg, (float)sb);
                    bool flag;
                    const uint32_t got = fast_convert(plane, (float)sr, (float)sg, (float)sb, &flag);
                    const double xf = (double)x - YUV_FAST_DELTA;
                    for (int i = 0; i < nset[sr]; i++)
                        for (int j = 0; j < nset[sg]; j++)
                            for (int k = 0; k < nset[sb]; k++) {
                                float pre;
                                const uint32_t want = yuv_byte(set[sr][i], set[sg][j], set[sb][k], plane, &pre);
                                const double dev = fabs(xf - (double)pre);
                                if (dev > maxdev) maxdev = dev;
                                total++;
                                flagged += flag;
                                if (got != want) {
                                    if (flag) mism_flagged++; else mism_unflagged++;
                                }
                            }
                }
        printf("plane %d %lld combinations: flagged %.4f %%, of them with another byte than the sequence's %lld, mismatches NOT flagged %lld; |fast - sequence| <= %.3g, guard band %.3g\n",
               plane, total, 100.0 * (double)flagged / (double)total, mism_flagged, mism_unflagged, maxdev, YUV_FAST_DELTA);
        if (mism_unflagged) bad = 1;
    }
    printf(bad ? "FAILED\n" : "OK: the fast path never disagrees with the reference sequence outside its guard band\n");
    return bad;
}
