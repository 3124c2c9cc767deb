/* render_scene.c — the C ABI from plain C: scene JSON + frames in HBM -> output frame (include/smr.h).
 *
 *   gcc -std=c11 -Iinclude examples/render_scene.c -o render_scene -Lsmelter_amd -l:libsmr_hip.so -Wl,-rpath,$PWD/smelter_amd -lm
 *   ./render_scene out.yuv          # 4 synthetic 640x360 inputs tiled into 1280x720 planar YUV420
 *
 * Mirrors what smelter-core does with smelter-render: Renderer::new, register_input, update_scene, render
 * (smelter-render/src/state.rs:96-193). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smr.h"

#define CHECK(call, what, errfn)                                          \
    do {                                                                  \
        int rc_ = (call);                                                 \
        if (rc_ < 0) {                                                    \
            fprintf(stderr, "%s failed (%d): %s\n", what, rc_, errfn);    \
            return 1;                                                     \
        }                                                                 \
    } while (0)

static const char *SCENE =
    "{\"type\": \"tiles\", \"background_color\": \"#101018FF\", \"margin\": 8, \"children\": ["
    " {\"type\": \"rescaler\", \"border_radius\": 16, \"child\": {\"type\": \"input_stream\", \"input_id\": \"cam0\"}},"
    " {\"type\": \"rescaler\", \"border_radius\": 16, \"child\": {\"type\": \"input_stream\", \"input_id\": \"cam1\"}},"
    " {\"type\": \"rescaler\", \"border_radius\": 16, \"child\": {\"type\": \"input_stream\", \"input_id\": \"cam2\"}},"
    " {\"type\": \"rescaler\", \"border_radius\": 16, \"child\": {\"type\": \"input_stream\", \"input_id\": \"cam3\"}}]}";

int main(int argc, char **argv) {
    enum { N = 4, IW = 640, IH = 360, OW = 1280, OH = 720 };
    smr_ctx *ctx = NULL;
    if (smr_ctx_create(0, SMR_MODE_GPU_OPTIMIZED, 0, NULL, &ctx) != 0) {
        fprintf(stderr, "no HIP device (there is no CPU fallback)\n");
        return 2;
    }
    smr_renderer *r = NULL;
    CHECK(smr_renderer_create(ctx, -1, &r), "smr_renderer_create", smr_last_error(ctx));

    /* inputs: a moving gradient per camera, planar YUV 4:2:0 resident in HBM */
    smr_frame frames[N];
    smr_input_frame set[N];
    static const char *ids[N] = {"cam0", "cam1", "cam2", "cam3"};
    unsigned char *y = malloc(IW * IH), *u = malloc(IW * IH / 4), *v = malloc(IW * IH / 4);
    for (int i = 0; i < N; i++) {
        for (int py = 0; py < IH; py++)
            for (int px = 0; px < IW; px++) y[py * IW + px] = (unsigned char)(16 + ((px + 40 * i) * 219 / IW + py / 4) % 220);
        memset(u, 90 + 30 * i, IW * IH / 4);
        memset(v, 200 - 35 * i, IW * IH / 4);
        CHECK(smr_frame_create(ctx, SMR_FRAME_PLANAR_YUV420, IW, IH, &frames[i]), "smr_frame_create", smr_last_error(ctx));
        const void *planes[3] = {y, u, v};
        CHECK(smr_frame_upload(ctx, &frames[i], planes), "smr_frame_upload", smr_last_error(ctx));
        CHECK(smr_renderer_register_input(r, ids[i]), "register_input", smr_renderer_last_error(r));
        set[i].input_id = ids[i];
        set[i].frame = &frames[i];
        set[i].pts_ns = 0;
    }
    CHECK(smr_renderer_update_scene(r, "out", OW, OH, SMR_FRAME_PLANAR_YUV420, SCENE), "update_scene", smr_renderer_last_error(r));

    smr_output_frame outs[1];
    uint32_t n_out = 0;
    CHECK(smr_renderer_render(r, 0, set, N, outs, 1, &n_out), "render", smr_renderer_last_error(r));
    unsigned char *oy = malloc(OW * OH), *ou = malloc(OW * OH / 4), *ov = malloc(OW * OH / 4);
    void *oplanes[3] = {oy, ou, ov};
    CHECK(smr_frame_download(ctx, outs[0].frame, oplanes), "smr_frame_download", smr_last_error(ctx));

    unsigned long sum = 0;
    for (int i = 0; i < OW * OH; i++) sum += oy[i];
    printf("rendered %ux%u from %d inputs: mean luma %.2f\n", OW, OH, N, (double)sum / (OW * OH));
    if (argc > 1) {
        FILE *f = fopen(argv[1], "wb");
        if (f) {
            fwrite(oy, 1, OW * OH, f);
            fwrite(ou, 1, OW * OH / 4, f);
            fwrite(ov, 1, OW * OH / 4, f);
            fclose(f);
        }
    }
    for (int i = 0; i < N; i++) smr_frame_destroy(ctx, &frames[i]);
    smr_renderer_destroy(r);
    smr_ctx_destroy(ctx);
    free(y); free(u); free(v); free(oy); free(ou); free(ov);
    return 0;
}
