// TEST INFRASTRUCTURE — not part of the product.  Buffers for the emulated kernels' "device memory" with the page after (or before)
// them unmapped: a load or store that leaves the allocation is a segmentation fault instead of a silent read of slack.  What the
// library promises about memory it does not own (include/smr.h, smr_surface_wrap): an allocation covers pitch * h bytes and no kernel
// reads past a row's pitch — with a plane of exactly pitch * h bytes ending at the guard page, the last row's last block proves it.
//   emu_set_guard(mode, min_pitch)   mode 0: heap buffers with slack (the default)   1: the buffer ENDS at a guard page   2: it STARTS behind one
//                                    min_pitch != 0: the smallest pitch the host code lets through instead of the allocator's 256-byte pitch
#pragma once
#include <sys/mman.h>
#include <unistd.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

static int emu_guard_mode = 0, emu_min_pitch = 0;
extern "C" void emu_set_guard(int mode, int min_pitch) { emu_guard_mode = mode; emu_min_pitch = min_pitch; }

struct GuardBuf {
    unsigned char *map = nullptr, *ptr = nullptr;
    size_t map_len = 0, size = 0;
    GuardBuf() = default;
    GuardBuf(const GuardBuf &) = delete;
    GuardBuf &operator=(const GuardBuf &) = delete;
    GuardBuf(GuardBuf &&o) noexcept { *this = static_cast<GuardBuf &&>(o); }
    GuardBuf &operator=(GuardBuf &&o) noexcept {
        release();
        map = o.map; ptr = o.ptr; map_len = o.map_len; size = o.size;
        o.map = o.ptr = nullptr; o.map_len = o.size = 0;
        return *this;
    }
    ~GuardBuf() { release(); }
    void release() {
        if (map) munmap(map, map_len);
        else free(ptr);
        map = ptr = nullptr;
    }
    // `bytes` usable bytes filled with `fill`, the start aligned to `align` (a power of two <= 256)
    void alloc(size_t bytes, unsigned char fill, size_t align = 16) {
        release();
        size = bytes;
        if (!emu_guard_mode) {
#ifdef SMR_EMU_ASAN  // (an instrumented build: the exact size, AddressSanitizer's red zones around it)
            if (posix_memalign((void **)&ptr, 256, bytes ? bytes : 1)) abort();
            memset(ptr, fill, bytes);
#else
            if (posix_memalign((void **)&ptr, 256, bytes + 64)) abort();  // (slack: the pre-guard behaviour)
            memset(ptr, fill, bytes + 64);
#endif
            return;
        }
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        const size_t body = (bytes + align - 1 + page - 1) / page * page;
        map_len = body + 2 * page;
        map = (unsigned char *)mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (map == MAP_FAILED) abort();
        memset(map, fill, map_len);
        if (emu_guard_mode == 1) ptr = map + page + ((body - bytes) & ~(align - 1));  // the end on (or within `align` of) the guard page
        else ptr = map + page;
        // (mode 1: callers pass sizes that are multiples of `align` — pitch * h — so that the last byte is the page's last)
        mprotect(map, page, PROT_NONE);
        mprotect(map + page + body, page, PROT_NONE);
    }
};
