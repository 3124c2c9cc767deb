/* render_scene.c — the C ABI from plain C: scene JSON + frames in HBM -> output frame (include/smr.h).
 *
 *   gcc -std=c11 -Iinclude examples/render_scene.c -o render_scene -Lsmelter_amd -l:libsmr_hip.so -Wl,-rpath,$PWD/smelter_amd -lm
 *   ./render_scene out.yuv [font dir]   # 4 synthetic 640x360 inputs tiled into 1280x720 planar YUV420, a text label per tile
 *
 * Text: the library's own font book (smr_fontbook_*: TrueType reader, layout, rasteriser — no Python, no third-party shaper) loaded from
 * `font dir` ($SMR_FONT_DIR, /usr/share/fonts/truetype); the reference bundles Inter (smelter-render/fonts/), pass that directory to use it.
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

/* BASELINE configs[2]'s shape at a quarter of the size: Tiles{ View{ Rescaler{InputStream} border_radius, label View{ Text } } } */
#define TILE(cam, label)                                                                                                                        \
    " {\"type\": \"view\", \"background_color\": \"#101018FF\", \"children\": ["                                                               \
    "   {\"type\": \"rescaler\", \"border_radius\": 16, \"child\": {\"type\": \"input_stream\", \"input_id\": \"" cam "\"}},"                          \
    "   {\"type\": \"view\", \"background_color\": \"#00000080\", \"border_radius\": 8, \"left\": 16, \"bottom\": 16, \"width\": 200, \"height\": 36,"     \
    "    \"padding_horizontal\": 8, \"padding_vertical\": 4, \"children\": ["                                                                     \
    "     {\"type\": \"text\", \"text\": \"" label "\", \"font_size\": 22, \"color\": \"#FFE080FF\"}]}]}"
static const char *SCENE_WITH_LABELS =
    "{\"type\": \"tiles\", \"background_color\": \"#202030FF\", \"margin\": 8, \"children\": ["
    TILE("cam0", "CAM 0 LIVE") "," TILE("cam1", "CAM 1 LIVE") "," TILE("cam2", "CAM 2 - R\u00e9gie") "," TILE("cam3", "CAM 3 LIVE") "]}";
static const char *SCENE_PLAIN =
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
    /* TextRendererCtx: a font book for the renderer — fitted Text nodes are measured with it and every Text node is laid out,
     * rasterised and drawn by update_scene itself (text_renderer.rs:72-167, 282-368) */
    smr_fontbook *fonts = NULL;
    const char *font_dir = argc > 2 ? argv[2] : getenv("SMR_FONT_DIR") ? getenv("SMR_FONT_DIR") : "/usr/share/fonts/truetype";
    int n_fonts = 0, n_text = 0;
    if (smr_fontbook_create(&fonts) == 0 && (n_fonts = smr_fontbook_add_dir(fonts, font_dir)) > 0)
        CHECK(smr_renderer_set_fontbook(r, fonts), "set_fontbook", smr_renderer_last_error(r));
    else
        fprintf(stderr, "no TrueType fonts below %s (%s): rendering without labels\n", font_dir, fonts ? smr_fontbook_last_error(fonts) : "no font book");
    CHECK(smr_renderer_update_scene(r, "out", OW, OH, SMR_FRAME_PLANAR_YUV420, n_fonts > 0 ? SCENE_WITH_LABELS : SCENE_PLAIN), "update_scene",
          smr_renderer_last_error(r));
    for (int i = 0, n = smr_renderer_node_count(r, "out"); i < n; i++) {
        smr_scene_node info;
        if (smr_renderer_node_info(r, "out", i, &info) == 0 && info.kind == SMR_NODE_TEXT) {
            n_text++;
            printf("text node %d: \"%s\" fitted to %ux%u\n", i, info.payload, info.width, info.height);
        }
    }

    smr_output_frame outs[1];
    uint32_t n_out = 0;
    CHECK(smr_renderer_render(r, 0, set, N, outs, 1, &n_out), "render", smr_renderer_last_error(r));
    unsigned char *oy = malloc(OW * OH), *ou = malloc(OW * OH / 4), *ov = malloc(OW * OH / 4);
    void *oplanes[3] = {oy, ou, ov};
    CHECK(smr_frame_download(ctx, outs[0].frame, oplanes), "smr_frame_download", smr_last_error(ctx));

    unsigned long sum = 0;
    for (int i = 0; i < OW * OH; i++) sum += oy[i];
    printf("rendered %ux%u from %d inputs: mean luma %.2f, %d text nodes drawn with %d font faces\n", OW, OH, N, (double)sum / (OW * OH), n_text, n_fonts);
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
    if (fonts) smr_fontbook_destroy(fonts);
    smr_ctx_destroy(ctx);
    free(y); free(u); free(v); free(oy); free(ou); free(ov);
    return 0;
}
