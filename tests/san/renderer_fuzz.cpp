// TEST INFRASTRUCTURE — not part of the product.  The renderer (smelter_amd/csrc/host/renderer.cpp: registry, update_scene, the
// depth-first walk of every output's render graph, lanes, per-node surfaces, text nodes, error paths) with the scene engine and the
// text pipeline, compiled by g++ under AddressSanitizer + UBSan and linked against tests/san/null_device.cpp instead of the GPU half
// of the library.  A random but deterministic sequence of everything a host can do through smr_renderer_*: scenes of the corpus
// and mutated ones on several outputs, frames of every format / fresh, stale, missing, for unknown inputs / on one to three lanes,
// inputs and outputs unregistered in between, images, shaders, glyph runs, a font book, allocation failures injected into the
// device.  A refused call is an answer; memory errors, undefined behaviour, leaked device surfaces fail the run.
//   renderer_fuzz CORPUS_DIR ITERATIONS SEED
#include <dirent.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "smr.h"

extern "C" long null_device_live_surfaces();
extern "C" void null_device_fail_after(long n);

static uint64_t rng_state = 0x2545F4914F6CDD1Dull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

static std::string slurp(const std::string &path) {
    std::string out;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return out;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return out;
}
static std::vector<std::string> list_dir(const std::string &dir, const char *suffix) {
    std::vector<std::string> out;
    DIR *d = opendir(dir.c_str());
    if (!d) return out;
    while (dirent *e = readdir(d)) {
        const std::string name = e->d_name;
        const size_t ls = strlen(suffix);
        if (name.size() > ls && name.compare(name.size() - ls, ls, suffix) == 0) out.push_back(dir + "/" + name);
    }
    closedir(d);
    std::sort(out.begin(), out.end());
    return out;
}

// scenes the render-test corpus has none of: text (fitted and fixed), images (own size and scaled), shader nodes with parameters and
// children, a blur over a nested layout, roots that are not layouts, a root layout of another size than the output
static const char *EXTRA_SCENES[] = {
    R"({"type":"text","text":"Hello, World!","font_size":40.0,"font_family":"Inter","color":"#FFFFFFFF","background_color":"#00800080"})",
    R"({"type":"view","children":[{"type":"text","text":"fixed box\nsecond line","font_size":22.0,"width":300,"height":90,"wrap":"word","align":"center"},
        {"type":"text","id":"t2","text":"AVATAR To. Régie — cam 1","font_size":31.5,"max_width":200,"wrap":"glyph","weight":"bold"}]})",
    R"({"type":"image","image_id":"image_1"})",
    R"({"type":"view","children":[{"type":"rescaler","child":{"type":"image","image_id":"image_1"}},{"type":"image","image_id":"image_2","width":77,"height":33}]})",
    R"({"type":"shader","shader_id":"s1","resolution":{"width":320,"height":180},"children":[{"type":"input_stream","input_id":"input_1"},{"type":"image","image_id":"image_1"}],
        "shader_param":{"type":"struct","value":[{"field_name":"a","type":"f32","value":0.5},{"field_name":"b","type":"list","value":[{"type":"u32","value":7},{"type":"i32","value":-3}]}]}})",
    R"({"type":"view","children":[{"type":"shader","shader_id":"gaussian_blur","resolution":{"width":640,"height":360},"shader_param":{"type":"struct","value":[{"field_name":"sigma","type":"f32","value":4.0}]},
        "children":[{"type":"view","width":200,"height":100,"background_color":"#FF000080","children":[{"type":"input_stream","input_id":"input_2"}]}]},
        {"type":"rescaler","child":{"type":"input_stream","input_id":"input_1"}}]})",
    R"({"type":"input_stream","input_id":"input_1"})",
    R"({"type":"view","width":300,"height":200,"background_color":"#112233FF","children":[{"type":"input_stream","input_id":"input_3"}]})",
    R"({"type":"tiles","id":"tl","transition":{"duration_ms":400},"children":[{"type":"input_stream","input_id":"input_1","id":"a"},{"type":"input_stream","input_id":"input_2","id":"b"},
        {"type":"text","text":"label","font_size":18.0,"id":"c"},{"type":"view","id":"d","border_radius":20,"border_width":3,"border_color":"#FFFFFFFF","box_shadow":[{"offset_x":4,"offset_y":4,"blur_radius":8,"color":"#00000080"}]}]})",
    R"({"type":"shader","shader_id":"unregistered","resolution":{"width":64,"height":64}})",
};

static const char *HOSTILE_NUMBERS[] = {"0", "-1", "1e30", "-1e30", "1e-30", "1e308", "1e999", "99999999999999999999", "0.5", "2147483648", "4294967296", "16777217", "7683", "4321", "null", "true", "\"12\""};

static std::string mutate(const std::string &src) {
    std::string s = src;
    const int rounds = 1 + (int)below(2);
    for (int r = 0; r < rounds && !s.empty(); r++) {
        switch ((rnd() & 1) ? 0 : 1 + below(4)) {
        case 0: {  // a number made hostile (the structure stays: the renderer gets to see it)
            std::vector<size_t> starts;
            for (size_t i = 1; i < s.size(); i++)
                if ((isdigit((unsigned char)s[i]) || s[i] == '-') && (s[i - 1] == ':' || s[i - 1] == ' ' || s[i - 1] == '[' || s[i - 1] == ',')) starts.push_back(i);
            if (starts.empty()) break;
            const size_t a = starts[below(starts.size())];
            size_t b = a;
            while (b < s.size() && (isdigit((unsigned char)s[b]) || s[b] == '-' || s[b] == '+' || s[b] == '.' || s[b] == 'e' || s[b] == 'E')) b++;
            s.replace(a, b - a, HOSTILE_NUMBERS[below(sizeof HOSTILE_NUMBERS / sizeof *HOSTILE_NUMBERS)]);
            break;
        }
        case 1: s[below(s.size())] = (char)rnd(); break;
        case 2: { const size_t a = below(s.size()), n = below(s.size() - a + 1); s.erase(a, n > 40 ? 40 : n); break; }
        case 3: {  // the scene as a child of something else
            const char *pre[] = {"{\"type\":\"view\",\"children\":[", "{\"type\":\"rescaler\",\"child\":", "{\"type\":\"tiles\",\"children\":[{\"type\":\"view\"},",
                                 "{\"type\":\"shader\",\"shader_id\":\"s1\",\"resolution\":{\"width\":99,\"height\":51},\"children\":["};
            const char *post[] = {"]}", "}", "]}", "]}"};
            const size_t k = below(4);
            s = std::string(pre[k]) + s + post[k];
            break;
        }
        case 4: {  // input ids swapped for ones nobody registered / images for missing ones
            const size_t p = s.find("input_", below(s.size()));
            if (p != std::string::npos) s.replace(p, 6, "nobody");
            break;
        }
        }
    }
    return s;
}

struct Frames {
    smr_ctx *ctx;
    std::vector<smr_frame> pool;  // frames of assorted formats and sizes, resident on the (null) device
    Frames(smr_ctx *c) : ctx(c) {
        const uint32_t sizes[][2] = {{1920, 1080}, {1280, 720}, {640, 360}, {641, 359}, {2, 2}, {1, 1}, {3840, 2160}, {180, 200}};
        for (uint32_t fmt = 0; fmt <= SMR_FRAME_RGBA; fmt++)
            for (int k = 0; k < 3; k++) {
                const auto &wh = sizes[below(sizeof sizes / sizeof *sizes)];
                smr_frame f;
                if (smr_frame_create(ctx, fmt, wh[0], wh[1], &f) == 0) pool.push_back(f);
            }
    }
    ~Frames() { for (auto &f : pool) smr_frame_destroy(ctx, &f); }
};

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: renderer_fuzz CORPUS_DIR ITERATIONS SEED\n"); return 2; }
    const std::string corpus = argv[1];
    const long iterations = atol(argv[2]);
    rng_state ^= (uint64_t)atoll(argv[3]) * 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 8; i++) rnd();

    std::vector<std::string> scenes;
    for (const auto &p : list_dir(corpus, ".json")) scenes.push_back(slurp(p));
    for (const char *s : EXTRA_SCENES) for (int k = 0; k < 12; k++) scenes.push_back(s);  // (weighted: the corpus is two hundred layout scenes)
    std::vector<std::string> fonts;
    for (const auto &p : list_dir(corpus, ".ttf")) fonts.push_back(slurp(p));

    std::map<std::string, long> refusals;  // HOST_FUZZ_REASONS=1: why calls were refused, printed at the end
    long updates_ok = 0, updates_refused = 0, renders_ok = 0, renders_refused = 0, frames_out = 0, text_runs = 0, injected = 0;
    const char *trace = getenv("HOST_FUZZ_TRACE");

    for (int round = 0; round < 4; round++) {  // four renderers one after another: creation and teardown are part of the test
        smr_ctx *ctx[3];
        for (auto &c : ctx) smr_ctx_create(0, (uint32_t)(round & 1), 100, nullptr, &c);
        smr_fontbook *book = nullptr;
        if (!fonts.empty()) {
            smr_fontbook_create(&book);
            for (const auto &f : fonts) smr_fontbook_add_memory(book, (const uint8_t *)f.data(), f.size());
        }
        {
            Frames frames(ctx[0]);
            smr_renderer *r = nullptr;
            smr_renderer_create(ctx[0], (rnd() & 1) ? -1 : (int64_t)(rnd() % 2000000000ull), &r);
            size_t lanes = 1;
            char id[32];
            for (int i = 1; i <= 15; i++) { snprintf(id, sizeof id, "input_%d", i); if (i != 9) smr_renderer_register_input(r, id); }
            std::vector<uint8_t> bitmap(320 * 240 * 4, 0x7f);
            smr_renderer_register_image(r, "image_1", bitmap.data(), 320, 240);
            smr_renderer_register_image(r, "image_2", bitmap.data(), 77, 33);
            smr_renderer_register_image(r, "image_1", bitmap.data(), 10, 10);  // (a second registration is refused)
            smr_renderer_register_shader(r, "s1", SMR_SHADER_LAYOUT_PLANES);
            smr_renderer_register_shader(r, "custom", SMR_SHADER_SILLY);
            smr_renderer_register_shader(r, "too_new", 99);
            if (book && (round & 2) == 0) smr_renderer_set_fontbook(r, book);
            int64_t pts = 0;
            for (long it = 0; it < iterations / 4; it++) {
                const size_t action = below(100);
                snprintf(id, sizeof id, "out_%zu", below(3));
                if (action < 30) {
                    std::string scene = scenes[below(scenes.size())];
                    if (rnd() & 1) scene = mutate(scene);
                    if (trace) { FILE *f = fopen(trace, "wb"); if (f) { fwrite(scene.data(), 1, scene.size(), f); fclose(f); } }
                    const uint32_t w = (rnd() & 31) == 0 ? (uint32_t)below(3) : 2 * (uint32_t)(1 + below(1000)), h = (rnd() & 31) == 0 ? 0 : 2 * (uint32_t)(1 + below(600));
                    const uint32_t formats[] = {SMR_FRAME_PLANAR_YUV420, SMR_FRAME_PLANAR_YUV420, SMR_FRAME_NV12, SMR_FRAME_PLANAR_YUV422, SMR_FRAME_PLANAR_YUV444, SMR_FRAME_RGBA, SMR_FRAME_BGRA};
                    if (smr_renderer_update_scene(r, id, w, h, formats[below(7)], scene.c_str()) == 0) {
                        updates_ok++;
                        // a host without a font book supplies the runs of the Text nodes itself
                        const int n = smr_renderer_node_count(r, id);
                        for (int node = -1; node <= n; node++) {
                            smr_scene_node info;
                            if (smr_renderer_node_info(r, id, node, &info) != 0 || info.kind != SMR_NODE_TEXT || (rnd() & 1)) continue;
                            volatile size_t sink = strlen(info.payload) + strlen(info.id) + strlen(info.ref_id);
                            (void)sink;
                            std::vector<smr_glyph> glyphs(below(6));
                            std::vector<uint8_t> atlas(64 * 32, 200);
                            for (auto &g : glyphs) {
                                g.w = 1 + (int)below(16); g.h = 1 + (int)below(16);
                                g.dst_x = (int)below(info.width ? info.width : 1); g.dst_y = (int)below(info.height ? info.height : 1);
                                if ((rnd() & 7) == 0) g.dst_x = -5;  // (a quad outside the node: refused by the blit, not a crash)
                                g.atlas_x = (int)below(48); g.atlas_y = (int)below(16);
                                for (float &c : g.color) c = 1.0f;
                            }
                            const float bg[4] = {0, 0, 0, 0};
                            if (smr_renderer_set_text(r, id, node, bg, glyphs.data(), (uint32_t)glyphs.size(), atlas.data(), 64, 32) == 0) text_runs++;
                        }
                    } else {
                        updates_refused++;
                        refusals["update: " + std::string(smr_renderer_last_error(r)).substr(0, 60)]++;
                    }
                } else if (action < 85) {
                    std::vector<smr_input_frame> in(below(10));
                    std::vector<std::string> ids(in.size());
                    for (size_t k = 0; k < in.size(); k++) {
                        const size_t which = 1 + below(16);
                        ids[k] = which == 16 ? "never_registered" : "input_" + std::to_string(which);
                        in[k].input_id = (rnd() & 63) == 0 ? nullptr : ids[k].c_str();
                        in[k].frame = (rnd() & 31) == 0 || frames.pool.empty() ? nullptr : &frames.pool[below(frames.pool.size())];
                        in[k].pts_ns = (rnd() & 7) == 0 ? pts - (int64_t)(rnd() % 3000000000ull) : pts;
                    }
                    smr_output_frame out[4];
                    uint32_t n_out = 0;
                    const uint32_t cap = (uint32_t)below(5);
                    if (smr_renderer_render(r, pts, in.data(), (uint32_t)in.size(), cap ? out : nullptr, cap, &n_out) == 0) {
                        renders_ok++;
                        for (uint32_t k = 0; k < n_out && k < cap; k++) {
                            volatile size_t s = strlen(out[k].output_id) + out[k].frame->width + (out[k].ctx != nullptr);
                            (void)s;
                            smr_surface_info si;
                            if (smr_surface_info_get(out[k].frame->planes[0], &si) == 0) frames_out++;  // (the frame is the renderer's and alive)
                        }
                    } else {
                        renders_refused++;
                        refusals["render: " + std::string(smr_renderer_last_error(r)).substr(0, 90)]++;
                    }
                    pts += (rnd() & 15) == 0 ? (int64_t)(rnd() % 5000000000ull) : 33333333;
                } else if (action < 88) {
                    smr_renderer_unregister_output(r, id);
                } else if (action < 91) {
                    snprintf(id, sizeof id, "input_%zu", 1 + below(15));
                    if (rnd() & 1) smr_renderer_unregister_input(r, id); else smr_renderer_register_input(r, id);
                } else if (action < 93) {
                    if (lanes < 3 && smr_renderer_add_lane(r, ctx[lanes]) == 0) lanes++;
                    smr_renderer_add_lane(r, ctx[0]);  // (already a lane: refused)
                } else if (action < 95) {
                    smr_renderer_sync(r);
                } else if (action < 97) {
                    if (book) smr_renderer_set_fontbook(r, (rnd() & 1) ? book : nullptr);
                } else {
                    null_device_fail_after((long)below(6));  // one of the next few device allocations fails
                    injected++;
                }
            }
            null_device_fail_after(-1);
            if (rnd() & 1) for (int k = 0; k < 3; k++) { snprintf(id, sizeof id, "out_%d", k); smr_renderer_unregister_output(r, id); }
            smr_renderer_destroy(r);
        }
        if (book) smr_fontbook_destroy(book);
        for (auto &c : ctx) smr_ctx_destroy(c);
        if (null_device_live_surfaces() != 0) {
            fprintf(stderr, "renderer_fuzz: %ld device surfaces were never destroyed (round %d)\n", null_device_live_surfaces(), round);
            return 1;
        }
    }
    // null arguments are errors, not crashes
    (void)smr_renderer_create(nullptr, 0, nullptr);
    (void)smr_renderer_last_error(nullptr);
    (void)smr_renderer_render(nullptr, 0, nullptr, 0, nullptr, 0, nullptr);
    (void)smr_renderer_update_scene(nullptr, "x", 2, 2, 0, "{}");
    smr_renderer_destroy(nullptr);

    if (getenv("HOST_FUZZ_REASONS"))
        for (const auto &kv : refusals) fprintf(stderr, "%8ld  %s\n", kv.second, kv.first.c_str());
    printf("{\"updates_ok\": %ld, \"updates_refused\": %ld, \"renders_ok\": %ld, \"renders_refused\": %ld, \"frames_out\": %ld, \"text_runs\": %ld, \"injected_failures\": %ld}\n",
           updates_ok, updates_refused, renders_ok, renders_refused, frames_out, text_runs, injected);
    return 0;
}
