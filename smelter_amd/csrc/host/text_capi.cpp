// text_capi.cpp — extern "C" surface of the text pipeline (include/smr.h, "a12"): font book, measuring, glyph runs.
#include <cstring>

#include "text.h"

static int fb_fail(smr_fontbook *b, const std::string &msg) {
    if (b) b->error = msg;
    return -1;
}

extern "C" {

SMR_API int smr_fontbook_create(smr_fontbook **out) {
    if (!out) return -1;
    *out = new smr_fontbook();
    return 0;
}
SMR_API void smr_fontbook_destroy(smr_fontbook *book) { delete book; }
SMR_API const char *smr_fontbook_last_error(const smr_fontbook *book) { return book ? book->error.c_str() : "null font book"; }
SMR_API uint32_t smr_fontbook_count(const smr_fontbook *book) { return book ? (uint32_t)book->book.size() : 0u; }

SMR_API int smr_fontbook_add_file(smr_fontbook *book, const char *path) {
    if (!book || !path) return fb_fail(book, "smr_fontbook_add_file: null argument");
    std::string err;
    return book->book.add_file(path, err) ? 0 : fb_fail(book, err);
}

SMR_API int smr_fontbook_add_memory(smr_fontbook *book, const uint8_t *data, size_t size) {
    if (!book || !data || !size) return fb_fail(book, "smr_fontbook_add_memory: null argument");
    std::string err;
    return book->book.add_memory(std::vector<uint8_t>(data, data + size), err) ? 0 : fb_fail(book, err);
}

SMR_API int smr_fontbook_add_dir(smr_fontbook *book, const char *dir) {
    if (!book || !dir) return fb_fail(book, "smr_fontbook_add_dir: null argument");
    std::string err;
    const int n = book->book.add_dir(dir, err);
    if (!n) return fb_fail(book, err.empty() ? std::string("no TrueType fonts below ") + dir : err);
    return n;
}

SMR_API int smr_fontbook_measure(void *user, const smr_text_params *params, float *widest_line, uint32_t *line_count) {
    smr_fontbook *book = (smr_fontbook *)user;
    if (!book || !params || !widest_line || !line_count) return 1;
    std::string err;
    if (!smr_text::measure(book->book, *params, *widest_line, *line_count, err)) {
        book->error = err;
        return 1;
    }
    return 0;
}

SMR_API int smr_fontbook_rasterise(smr_fontbook *book, const smr_text_params *params, uint32_t width, uint32_t height, const float color[4],
                                   smr_text_run *out) {
    if (!book || !params || !color || !out) return fb_fail(book, "smr_fontbook_rasterise: null argument");
    std::string err;
    if (!smr_text::rasterise(book->book, *params, width, height, color, book->run, err)) return fb_fail(book, err);
    out->glyphs = book->run.glyphs.data();
    out->n_glyphs = (uint32_t)book->run.glyphs.size();
    out->atlas = book->run.atlas.data();
    out->atlas_w = book->run.atlas_w;
    out->atlas_h = book->run.atlas_h;
    return 0;
}

}  // extern "C"
