// TEST INFRASTRUCTURE — not part of the product.  The host side of the library (scene engine: scene*.cpp; text pipeline: text*.cpp)
// compiled by g++ with AddressSanitizer + UndefinedBehaviorSanitizer and driven through its C ABI (include/smr.h) with
//   1. every scene of the corpus as it is (the reference's own API vectors and render-test scenes, written out by tests/test_host_sanitizers.py),
//   2. mutated scenes: truncations, byte flips, hostile numbers, deleted / repeated slices, deep nesting, long strings,
//   3. fonts as they are and corrupted (table directory, cmap, loca / glyf, hmtx, GPOS), with hostile text parameters.
// A scene or a font may be REJECTED (that is an answer); what may not happen is a memory error, undefined behaviour or a hang — the
// sanitizers abort the process and the test fails.  Deterministic: xorshift from the seed on the command line.
//   host_fuzz CORPUS_DIR ITERATIONS SEED
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "smr.h"
#include "text.h"  // (smr_text::accumulate_edge: the one internal the harness calls directly)

static uint64_t rng_state = 88172645463325252ull;
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

struct Counts {
    long scenes_ok = 0, scenes_rejected = 0, layouts = 0, fonts_ok = 0, fonts_rejected = 0, runs = 0, glyphs = 0;
} counts;

static smr_fontbook *g_book = nullptr;  // the measurer of Text nodes

// HOST_FUZZ_TRACE=path: the input about to run is written there first (what to look at after a hang or an abort)
static void trace_input(const char *what, const std::string &data) {
    static const char *path = getenv("HOST_FUZZ_TRACE");
    if (!path) return;
    FILE *f = fopen(path, "wb");
    if (!f) return;
    fprintf(f, "%s\n", what);
    fwrite(data.data(), 1, data.size(), f);
    fclose(f);
}

// everything a host does with a scene after an update: the node graph, every node's layouts at a few times
static void walk(smr_scene *sc, const int64_t *pts, int n_pts) {
    const int n = smr_scene_node_count(sc);
    for (int node = -1; node <= n; node++) {  // (-1 and n: out of range on purpose)
        smr_scene_node info;
        const int rc = smr_scene_node_info(sc, node, &info);
        if (rc != 0) { (void)smr_scene_last_error(sc); continue; }
        // the strings are the scene's: touch every byte
        volatile size_t sink = strlen(info.id) + strlen(info.ref_id) + strlen(info.payload);
        (void)sink;
        int32_t kids[8];
        const int nk = smr_scene_node_children(sc, node, kids, 8);
        if (info.kind != SMR_NODE_LAYOUT) continue;
        std::vector<uint32_t> wh(2 * (size_t)(nk > 0 ? nk : 0));
        for (int k = 0; k < nk; k++) {
            const uint64_t r = rnd();
            wh[2 * k] = (r & 7) == 0 ? SMR_NO_RESOLUTION : (uint32_t)(1 + (r >> 8) % 1920);
            wh[2 * k + 1] = (uint32_t)(1 + (r >> 24) % 1080);
        }
        for (int t = 0; t < n_pts; t++) {
            std::vector<smr_layout> out(64);
            uint32_t n_out = 0, w = 0, h = 0;
            const uint32_t cap = (uint32_t)(rnd() & 1 ? out.size() : 3);  // (a short buffer: the count is still the whole list's)
            if (smr_scene_node_layouts(sc, node, pts[t], wh.data(), (uint32_t)(nk > 0 ? nk : 0), (uint32_t)(rnd() & 1), out.data(), cap, &n_out, &w, &h) == 0)
                counts.layouts += n_out;
        }
    }
}

static void run_scene(const std::string &json, uint32_t w, uint32_t h, bool keep_previous) {
    static smr_scene *persistent = nullptr;  // updates on top of one another exercise the transition state (ids that stay, go, come back)
    smr_scene *sc = nullptr;
    if (keep_previous) {
        if (!persistent) {
            smr_scene_create(&persistent);
            smr_scene_register_image(persistent, "image_1", 320, 240);
            if (g_book) smr_scene_set_text_measurer(persistent, smr_fontbook_measure, g_book);
        }
        sc = persistent;
    } else {
        smr_scene_create(&sc);
        smr_scene_register_image(sc, "image_1", 320, 240);
        if (g_book) smr_scene_set_text_measurer(sc, smr_fontbook_measure, g_book);
    }
    trace_input("scene", json);
    const char *canon = nullptr;
    if (smr_scene_parse(sc, json.c_str(), &canon) == 0) { volatile size_t s = strlen(canon); (void)s; }
    static int64_t clock_ns = 0;
    if (smr_scene_update(sc, json.c_str(), w, h) == 0) {
        counts.scenes_ok++;
        const int64_t pts[5] = {clock_ns, clock_ns + 250000000ll, clock_ns + 1000000000ll, clock_ns + (int64_t)(rnd() % 20000000000ull),
                                (rnd() & 15) == 0 ? -(int64_t)(rnd() % 1000000000ull) : clock_ns + 20000000000ll};
        walk(sc, pts, 5);
        clock_ns += 500000000ll;
    } else {
        counts.scenes_rejected++;
        volatile size_t s = strlen(smr_scene_last_error(sc));
        (void)s;
        const int64_t pts[1] = {clock_ns};
        walk(sc, pts, 1);  // the previous scene is still there
    }
    if (!keep_previous) smr_scene_destroy(sc);
}

static const char *HOSTILE_NUMBERS[] = {"0", "-1", "-0.0", "1e30", "-1e30", "1e-30", "1e308", "1e999", "-1e999", "99999999999999999999", "0.5", "2147483648",
                                        "4294967296", "-2147483649", "1e-320", "16777217", "NaN", "Infinity", "null", "true", "\"12\"", "[]", "{}"};

static std::string mutate(const std::string &src) {
    std::string s = src;
    const int rounds = 1 + (int)below(3);
    for (int r = 0; r < rounds; r++) {
        if (s.empty()) break;
        switch ((rnd() & 1) ? 2 : below(9)) {  // (half of the mutations keep the structure and only make a number hostile)
        case 0: s.resize(below(s.size())); break;                                       // truncation
        case 1: s[below(s.size())] = (char)rnd(); break;                                // a flipped byte
        case 2: {                                                                       // a number replaced by a hostile one
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
        case 3: { const size_t a = below(s.size()), n = below(s.size() - a + 1); s.erase(a, n > 64 ? 64 : n); break; }                       // a slice deleted
        case 4: { const size_t a = below(s.size()), n = 1 + below(48); s.insert(below(s.size()), s.substr(a, n)); break; }                   // a slice repeated elsewhere
        case 5: {                                                                       // nesting: the scene as the only child of N views
            const size_t depth = (rnd() & 7) == 0 ? 1 + below(3000) : 1 + below(40);
            std::string pre, post;
            for (size_t i = 0; i < depth; i++) { pre += "{\"type\":\"view\",\"children\":["; post += "]}"; }
            s = pre + s + post;
            break;
        }
        case 6: {                                                                       // a string value grown long / emptied / made non-ASCII
            const size_t q = s.find('"', below(s.size()));
            if (q == std::string::npos) break;
            const size_t e = s.find('"', q + 1);
            if (e == std::string::npos) break;
            const char *fill[] = {"", "\\u0000", "\\ud800", "\xf0\x9f\x98\x80", "\xff\xfe", "\\", "#GGGGGGGG", "#FF", "1e9px"};
            std::string v = fill[below(sizeof fill / sizeof *fill)];
            if ((rnd() & 3) == 0) v.assign(1 + below(20000), 'W');
            s.replace(q + 1, e - q - 1, v);
            break;
        }
        case 7: {                                                                       // brackets unbalanced
            const char br[] = "{}[],:\"";
            s.insert(below(s.size()), 1, br[below(sizeof br - 1)]);
            break;
        }
        case 8: {                                                                       // ids that collide / transitions on everything
            const size_t p = s.find("\"type\"", below(s.size()));
            if (p != std::string::npos) s.insert(p, "\"id\":\"same\",\"transition\":{\"duration_ms\":" + std::string(HOSTILE_NUMBERS[below(12)]) + "},");
            break;
        }
        }
    }
    return s;
}

static const char *TEXTS[] = {"", " ", "\n", "Hello, World!", "AV To WA fi ffl", "R\xc3\xa9gie  \xe2\x80\x94  cam 1", "line one\nline two\n\nline four",
                              "\xf0\x9f\x98\x80 emoji \xe6\xbc\xa2\xe5\xad\x97", "\xff\xfe\xfd bad utf8 \xc3", "a\tb\r\nc", "word word word word word word word word word word word word"};
static const char *WRAPS[] = {"None", "Glyph", "Word", "bogus", ""};
static const char *ALIGNS[] = {"Left", "Right", "Justified", "Center", "bogus"};
static const char *WEIGHTS[] = {"Normal", "Bold", "Thin", "Black", "ExtraLight", "bogus"};
static const char *STYLES[] = {"Normal", "Italic", "Oblique", "bogus"};
static const float SIZES[] = {0.0f, -1.0f, 0.5f, 7.0f, 12.0f, 25.0f, 50.0f, 96.0f, 400.0f, 1e6f, 1e30f, NAN, INFINITY};
static const float BOXES[] = {0.0f, 1.0f, 17.5f, 170.0f, 640.0f, 4096.0f, -5.0f, 1e9f, NAN, INFINITY};

static void run_text(smr_fontbook *book, int n) {
    for (int i = 0; i < n; i++) {
        std::string text = TEXTS[below(sizeof TEXTS / sizeof *TEXTS)];
        if ((rnd() & 15) == 0) { text.clear(); for (size_t k = 0, m = below(400); k < m; k++) text.push_back((char)rnd()); for (auto &c : text) if (!c) c = ' '; }
        if ((rnd() & 31) == 0) text.assign(5000, 'm');
        smr_text_params p;
        p.text = text.c_str();
        p.font_family = (rnd() & 3) ? "Inter" : ((rnd() & 1) ? "DejaVu Sans" : "no such family");
        p.style = STYLES[below(4)];
        p.weight = WEIGHTS[below(6)];
        p.wrap = WRAPS[below(5)];
        p.align = ALIGNS[below(5)];
        p.font_size = SIZES[(rnd() & 3) ? 3 + below(5) : below(sizeof SIZES / sizeof *SIZES)];
        p.line_height = (rnd() & 3) ? p.font_size * 1.2f : SIZES[below(sizeof SIZES / sizeof *SIZES)];
        p.max_width = BOXES[(rnd() & 3) ? 2 + below(4) : below(sizeof BOXES / sizeof *BOXES)];
        p.max_height = BOXES[(rnd() & 3) ? 2 + below(4) : below(sizeof BOXES / sizeof *BOXES)];
        float widest = 0;
        uint32_t lines = 0;
        if (smr_fontbook_measure(book, &p, &widest, &lines) != 0) { volatile size_t s = strlen(smr_fontbook_last_error(book)); (void)s; }
        const uint32_t w = (uint32_t)(rnd() % 700), h = (uint32_t)(rnd() % 200);
        const float color[4] = {1.0f, 0.5f, 0.25f, 1.0f};
        smr_text_run run;
        if (smr_fontbook_rasterise(book, &p, w, h, color, &run) == 0) {
            counts.runs++;
            counts.glyphs += run.n_glyphs;
            // the run is the book's: read every quad and every atlas byte a blit would read
            uint64_t sum = 0;
            for (uint32_t g = 0; g < run.n_glyphs; g++) {
                const uint8_t *q = (const uint8_t *)&run.glyphs[g];
                for (size_t b = 0; b < sizeof(smr_glyph); b++) sum += q[b];
            }
            for (size_t b = 0; b < (size_t)run.atlas_w * run.atlas_h; b++) sum += run.atlas[b];
            volatile uint64_t sink = sum;
            (void)sink;
        } else {
            volatile size_t s = strlen(smr_fontbook_last_error(book));
            (void)s;
        }
    }
}

static uint32_t be32(const std::string &d, size_t o) { return o + 4 <= d.size() ? ((uint32_t)(uint8_t)d[o] << 24) | ((uint32_t)(uint8_t)d[o + 1] << 16) | ((uint32_t)(uint8_t)d[o + 2] << 8) | (uint8_t)d[o + 3] : 0; }
static uint32_t be16(const std::string &d, size_t o) { return o + 2 <= d.size() ? ((uint32_t)(uint8_t)d[o] << 8) | (uint8_t)d[o + 1] : 0; }

static std::string corrupt_font(const std::string &src) {
    std::string f = src;
    const uint32_t n_tables = be16(f, 4);
    const int rounds = 1 + (int)below(6);
    for (int r = 0; r < rounds; r++) {
        switch (below(6)) {
        case 0: f[below(f.size())] = (char)rnd(); break;                                                  // anywhere
        case 1: if (n_tables) f[12 + below((size_t)n_tables * 16)] = (char)rnd(); break;                  // the table directory (tags, offsets, lengths)
        case 2: f.resize(below(f.size())); if (f.size() < 12) f.resize(12, 0); break;                     // truncated
        default: {                                                                                        // inside one table (small tables get hit as often as glyf)
            if (!n_tables) break;
            const size_t rec = 12 + below(n_tables) * 16;
            const uint32_t off = be32(f, rec + 8), len = be32(f, rec + 12);
            if (!len || off >= f.size()) break;
            const size_t span = (size_t)len < f.size() - off ? len : f.size() - off;
            const size_t head = (rnd() & 1) ? (span < 64 ? span : 64) : span;  // (headers of a table decide the most)
            const size_t at = off + below(head);
            const int k = 1 + (int)below(4);
            for (int i = 0; i < k && at + i < f.size(); i++) f[at + i] = (rnd() & 3) ? (char)rnd() : (char)0xff;
            break;
        }
        }
    }
    return f;
}

int main(int argc, char **argv) {
    if (argc == 3 && !strcmp(argv[1], "--font")) {  // replay one font (a HOST_FUZZ_TRACE file without its first line)
        const std::string f = slurp(argv[2]);
        smr_fontbook *book = nullptr;
        smr_fontbook_create(&book);
        if (smr_fontbook_add_memory(book, (const uint8_t *)f.data(), f.size()) == 0) run_text(book, 50);
        else printf("rejected: %s\n", smr_fontbook_last_error(book));
        smr_fontbook_destroy(book);
        return 0;
    }
    if (argc == 3 && !strcmp(argv[1], "--edges")) {  // the rasteriser's edge accumulation on edges that END on the bitmap's left / right column
        // (x advanced row by row used to arrive a rounding error past the edge's own end: column -1 — ADVICE round 5; every accumulator here is
        //  exactly w * h + 1 doubles, so AddressSanitizer sees the first write outside it)
        const long n = atol(argv[2]);
        long done = 0;
        for (long i = 0; i < n; i++) {
            const int w = 1 + (int)below(6), h = 1 + (int)below(4);
            std::vector<double> acc((size_t)w * h + 1, 0.0);
            auto u = [&]() { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); };
            const double xmax = (double)(w - 1);  // (rasterise_glyph sizes the bitmap so that every x lies in [0, w - 1])
            double x0 = u() * xmax, x1 = (rnd() & 1) ? 0.0 : xmax, y0 = u() * h, y1 = u() * h;
            if (rnd() & 1) y1 = y0 + (u() - 0.5) * 1e-3;          // nearly horizontal, inside one row
            if (rnd() & 3) { std::swap(x0, x1); std::swap(y0, y1); }
            if (i == 0) { x0 = 0.301; y0 = 0.1; x1 = 0.0; y1 = 0.10037; }  // the advisor's case (w >= 2 there)
            if (i == 0) { std::vector<double> a4((size_t)4 * 2 + 1, 0.0); smr_text::accumulate_edge(a4, 4, 2, x0, y0, x1, y1); done++; continue; }
            y0 = std::min(std::max(y0, 0.0), (double)h); y1 = std::min(std::max(y1, 0.0), (double)h);
            smr_text::accumulate_edge(acc, w, h, x0, y0, x1, y1);
            done++;
        }
        printf("{\"edges\": %ld}\n", done);
        return 0;
    }
    if (argc == 3 && !strcmp(argv[1], "--scene")) {  // replay one scene
        run_scene(slurp(argv[2]), 640, 360, false);
        return 0;
    }
    if (argc < 4) { fprintf(stderr, "usage: host_fuzz CORPUS_DIR ITERATIONS SEED | --font FILE | --scene FILE\n"); return 2; }
    const std::string corpus = argv[1];
    const long iterations = atol(argv[2]);
    rng_state ^= (uint64_t)atoll(argv[3]) * 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 8; i++) rnd();

    std::vector<std::string> scenes, fonts;
    for (const auto &p : list_dir(corpus, ".json")) scenes.push_back(slurp(p));
    for (const auto &p : list_dir(corpus, ".ttf")) fonts.push_back(slurp(p));
    if (scenes.empty()) { fprintf(stderr, "no scenes in %s\n", corpus.c_str()); return 2; }

    if (!fonts.empty()) {
        smr_fontbook_create(&g_book);
        for (const auto &f : fonts)
            if (smr_fontbook_add_memory(g_book, (const uint8_t *)f.data(), f.size()) == 0) counts.fonts_ok++;
        (void)smr_fontbook_count(g_book);
    }

    // 1. the corpus as it is: alone, then as successive updates of one scene
    for (const auto &s : scenes) run_scene(s, 640, 360, false);
    for (const auto &s : scenes) run_scene(s, 1920, 1080, true);
    const long corpus_ok = counts.scenes_ok;

    // 2. mutated scenes
    for (long i = 0; i < iterations; i++) {
        const std::string m = mutate(scenes[below(scenes.size())]);
        const uint32_t w = (rnd() & 15) == 0 ? (uint32_t)(rnd() % 3) : 2 * (uint32_t)(1 + rnd() % 2000), h = (rnd() & 15) == 0 ? 0u : 2 * (uint32_t)(1 + rnd() % 1200);
        run_scene(m, w, h, (rnd() & 1) != 0);
    }

    // 3. text: the fonts as they are, then corrupted
    if (g_book) run_text(g_book, (int)(iterations / 4 + 50));
    for (long i = 0; i < iterations / 8 && !fonts.empty(); i++) {
        const std::string bad = corrupt_font(fonts[below(fonts.size())]);
        trace_input("font", bad);
        smr_fontbook *book = nullptr;
        smr_fontbook_create(&book);
        if (smr_fontbook_add_memory(book, (const uint8_t *)bad.data(), bad.size()) == 0) {
            counts.fonts_ok++;
            run_text(book, 6);
        } else {
            counts.fonts_rejected++;
            volatile size_t s = strlen(smr_fontbook_last_error(book));
            (void)s;
        }
        smr_fontbook_destroy(book);
    }
    // null arguments are errors, not crashes
    (void)smr_scene_update(nullptr, "{}", 1, 1);
    (void)smr_scene_last_error(nullptr);
    (void)smr_fontbook_add_memory(nullptr, nullptr, 0);
    (void)smr_fontbook_last_error(nullptr);
    (void)smr_fontbook_measure(nullptr, nullptr, nullptr, nullptr);
    if (g_book) { (void)smr_fontbook_add_memory(g_book, nullptr, 16); (void)smr_fontbook_add_file(g_book, nullptr); (void)smr_fontbook_add_dir(g_book, "/nonexistent"); smr_fontbook_destroy(g_book); }
    smr_fontbook_destroy(nullptr);
    smr_scene_destroy(nullptr);

    printf("{\"corpus_scenes\": %zu, \"corpus_ok\": %ld, \"scenes_ok\": %ld, \"scenes_rejected\": %ld, \"layouts\": %ld, \"fonts_ok\": %ld, \"fonts_rejected\": %ld, "
           "\"runs\": %ld, \"glyphs\": %ld}\n",
           scenes.size(), corpus_ok, counts.scenes_ok, counts.scenes_rejected, counts.layouts, counts.fonts_ok, counts.fonts_rejected, counts.runs, counts.glyphs);
    return 0;
}
