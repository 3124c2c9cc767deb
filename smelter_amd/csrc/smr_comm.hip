// smr_comm.hip — the exchange step of the multi-GPU path behind the C ABI: dst-sized RGBA8 tiles travel from the GPU that
// resampled them to the GPU that composes (the reference has a single wgpu device and no counterpart —
// the per-input independence this shards on is smelter-render/src/state/render_loop.rs:24-41, transformations/layout.rs:250-275).
//
// Two ways to own several GPUs, one gather entry point:
//   * local  — one process, one smr_ctx per device (what smelter-core's single renderer thread would hold,
//              smelter-core/src/pipeline/instance.rs:435-503): peer copies, hipMemcpy2DAsync on the owner's stream, the root's
//              stream waits on an event per sender.  Each peer has its own xGMI link to the root.
//   * ranks  — one process per GPU: RCCL point-to-point (ncclSend / ncclRecv in one group) on the ctx stream.  librccl is
//              opened on first use, so a single-GPU host never needs it.
// Everything is stream-ordered: a tile is sent after the ingest kernel that wrote it and composed after it arrived; nothing
// here synchronises the host.
#include "smr_internal.h"

#include <dlfcn.h>
#include <mutex>

#include <cstdlib>

namespace {

// ---- the slice of the RCCL API this file uses (rccl.h is not needed at build time; same ABI as NCCL 2)
typedef void *ncclComm_t;
struct ncclUniqueId { char internal[SMR_COMM_ID_BYTES]; };
enum { ncclSuccess = 0, ncclUint8 = 1 };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

const Rccl *rccl(std::string *err) {
    static Rccl r;
    static std::once_flag once;
    static bool ok = false;
    static std::string why;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            const char *e = dlerror();  // (one call: dlerror() clears the message it returns)
            why = std::string("librccl.so could not be loaded: ") + (e ? e : "?");
        } else {
            bool all = true;
            auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p) { all = false; why = std::string("librccl.so lacks ") + n; } return p; };
            r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
            r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
            r.Send = (decltype(r.Send))sym("ncclSend");
            r.Recv = (decltype(r.Recv))sym("ncclRecv");
            r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
            ok = all;
        }
    });
    if (!ok && err) *err = why;
    return ok ? &r : nullptr;
}

}  // namespace

struct smr_comm {
    uint32_t world = 1, rank = 0;
    bool local = false;
    std::vector<smr_ctx *> ctxs;   // local: one per rank; ranks: {own ctx}
    std::vector<hipEvent_t> sent;  // local: one per rank, recorded after its copies
    ncclComm_t nccl = nullptr;
    std::string err;
};

static int comm_fail(smr_comm *c, smr_ctx *ctx, int code, const std::string &msg) {
    if (c) c->err = msg;
    if (ctx) ctx->err = msg;
    return code;
}

extern "C" {

int smr_comm_create_local(smr_ctx *const *ctxs, uint32_t n, smr_comm **out) {
    if (!ctxs || !n || !out) return SMR_ERR_INVALID;
    *out = nullptr;
    for (uint32_t i = 0; i < n; i++)
        if (!ctxs[i]) return SMR_ERR_INVALID;
    smr_comm *c = new smr_comm();
    c->world = n;
    c->local = true;
    c->ctxs.assign(ctxs, ctxs + n);
    c->sent.assign(n, nullptr);
    for (uint32_t i = 0; i < n; i++) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess || hipEventCreateWithFlags(&c->sent[i], hipEventDisableTiming) != hipSuccess) {
            int rc = comm_fail(nullptr, ctxs[i], SMR_ERR_INTERNAL, "smr_comm_create_local: event creation failed");
            smr_comm_destroy(c);
            return rc;
        }
        for (uint32_t j = 0; j < n; j++) {
            if (ctxs[j]->device == ctxs[i]->device) continue;
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, ctxs[i]->device, ctxs[j]->device);
            if (!can) {
                int rc = comm_fail(nullptr, ctxs[i], SMR_ERR_INVALID, "smr_comm_create_local: devices " + std::to_string(ctxs[i]->device) + " and " +
                                                                       std::to_string(ctxs[j]->device) + " have no peer access");
                smr_comm_destroy(c);
                return rc;
            }
            hipError_t e = hipDeviceEnablePeerAccess(ctxs[j]->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                int rc = smr_check_hip(ctxs[i], e, "hipDeviceEnablePeerAccess");
                smr_comm_destroy(c);
                return rc;
            }
            (void)hipGetLastError();
        }
    }
    *out = c;
    return SMR_OK;
}

int smr_comm_unique_id(uint8_t id[SMR_COMM_ID_BYTES]) {
    if (!id) return SMR_ERR_INVALID;
    std::string why;
    const Rccl *r = rccl(&why);
    if (!r) return SMR_ERR_INTERNAL;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return SMR_ERR_INTERNAL;
    memcpy(id, u.internal, SMR_COMM_ID_BYTES);
    return SMR_OK;
}

int smr_comm_create_rank(smr_ctx *ctx, uint32_t world, uint32_t rank, const uint8_t id[SMR_COMM_ID_BYTES], smr_comm **out) {
    if (!ctx || !out || !id || !world || rank >= world) return SMR_ERR_INVALID;
    *out = nullptr;
    SMR_ENTER(ctx);
    std::string why;
    const Rccl *r = rccl(&why);
    if (!r) return smr_fail(ctx, SMR_ERR_INTERNAL, "smr_comm_create_rank: %s", why.c_str());
    smr_comm *c = new smr_comm();
    c->world = world;
    c->rank = rank;
    c->ctxs = {ctx};
    ncclUniqueId u;
    memcpy(u.internal, id, SMR_COMM_ID_BYTES);
    int rc = r->CommInitRank(&c->nccl, (int)world, u, (int)rank);
    if (rc != ncclSuccess) {
        delete c;
        return smr_fail(ctx, SMR_ERR_INTERNAL, "ncclCommInitRank: %s", r->GetErrorString(rc));
    }
    *out = c;
    return SMR_OK;
}

void smr_comm_destroy(smr_comm *c) {
    if (!c) return;
    if (c->nccl) {
        const Rccl *r = rccl(nullptr);
        if (r) (void)r->CommDestroy(c->nccl);
    }
    for (size_t i = 0; i < c->sent.size(); i++)
        if (c->sent[i]) {
            (void)hipSetDevice(c->ctxs[i]->device);
            (void)hipEventDestroy(c->sent[i]);
        }
    delete c;
}

uint32_t smr_comm_world(const smr_comm *c) { return c ? c->world : 0; }
uint32_t smr_comm_rank(const smr_comm *c) { return c ? c->rank : 0; }
const char *smr_comm_last_error(const smr_comm *c) { return c ? c->err.c_str() : "null comm"; }

// Tile i was produced on rank owner[i] in src[i]; after the call — in stream order on the root's ctx — dst[i] on the root
// holds it.  Entries whose owner is the root are skipped (the root composes from its own src[i]).
//   local comm: one call moves everything (src[i] lives on ctxs[owner[i]], dst[i] on ctxs[root]).
//   rank comm:  every rank makes the same call; a rank reads src[i] only for the tiles it owns, the root writes dst[i] only.
int smr_gather_tiles(smr_comm *c, uint32_t root, const uint32_t *owner, const smr_surface *const *src, smr_surface *const *dst, uint32_t n) {
    if (!c || (n && (!owner || !src || !dst)) || root >= c->world) return SMR_ERR_INVALID;
    smr_ctx *me = c->local ? c->ctxs[root] : c->ctxs[0];
    for (uint32_t i = 0; i < n; i++) {
        if (owner[i] >= c->world) return comm_fail(c, me, SMR_ERR_INVALID, "smr_gather_tiles: tile " + std::to_string(i) + " has no owner");
        if (owner[i] == root) continue;
        const bool sends = c->local || c->rank == owner[i], recvs = c->local || c->rank == root;
        if ((sends && !src[i]) || (recvs && !dst[i])) return comm_fail(c, me, SMR_ERR_INVALID, "smr_gather_tiles: tile " + std::to_string(i) + " is missing");
        if (sends && recvs && (src[i]->w != dst[i]->w || src[i]->h != dst[i]->h || src[i]->fmt != dst[i]->fmt))
            return comm_fail(c, me, SMR_ERR_INVALID, "smr_gather_tiles: tile " + std::to_string(i) + " differs between owner and root");
    }
    if (c->local) {
        // owner's stream: wait until the root is done with the destination (its previous compose), copy, signal
        std::vector<u8> used(c->world, 0);
        hipEvent_t root_free = c->sent[root];
        SMR_HIP(me, hipSetDevice(me->device));
        SMR_HIP(me, hipEventRecord(root_free, me->stream));
        for (uint32_t i = 0; i < n; i++) {
            if (owner[i] == root) continue;
            smr_ctx *o = c->ctxs[owner[i]];
            SMR_HIP(o, hipSetDevice(o->device));
            if (!used[owner[i]]) SMR_HIP(o, hipStreamWaitEvent(o->stream, root_free, 0));
            used[owner[i]] = 1;
            SMR_HIP(o, hipMemcpy2DAsync(dst[i]->ptr, dst[i]->pitch, src[i]->ptr, src[i]->pitch, (size_t)src[i]->w * bytes_per_px(src[i]->fmt), src[i]->h,
                                        hipMemcpyDeviceToDevice, o->stream));
        }
        for (uint32_t r = 0; r < c->world; r++) {
            if (!used[r]) continue;
            smr_ctx *o = c->ctxs[r];
            SMR_HIP(o, hipSetDevice(o->device));
            SMR_HIP(o, hipEventRecord(c->sent[r], o->stream));
            SMR_HIP(me, hipSetDevice(me->device));
            SMR_HIP(me, hipStreamWaitEvent(me->stream, c->sent[r], 0));
        }
        SMR_HIP(me, hipSetDevice(me->device));
        return SMR_OK;
    }
    SMR_ENTER(me);
    const Rccl *r = rccl(nullptr);
    if (!r) return comm_fail(c, me, SMR_ERR_INTERNAL, "smr_gather_tiles: librccl is not loaded");
    bool any = false;
    for (uint32_t i = 0; i < n; i++)
        any = any || (owner[i] != root && (c->rank == root || c->rank == owner[i]));
    if (!any) return SMR_OK;
    // Sender and receiver are different processes: the byte counts of a send / recv pair must agree without either side seeing
    // the other's surface, so both sides require the library's own pitch rule (rows padded to 256 bytes, smr_surface_create).
    for (uint32_t i = 0; i < n; i++) {
        if (owner[i] == root) continue;
        const smr_surface *s = c->rank == owner[i] ? src[i] : (c->rank == root ? dst[i] : nullptr);
        if (!s) continue;
        const size_t canon = (((size_t)s->w * bytes_per_px(s->fmt)) + 255) & ~(size_t)255;
        if (s->pitch != canon)
            return comm_fail(c, me, SMR_ERR_INVALID, "smr_gather_tiles: tile " + std::to_string(i) + " has pitch " + std::to_string(s->pitch) +
                                                         ", the rank-mode gather needs the canonical pitch " + std::to_string(canon) + " on both sides");
    }
    int rc = r->GroupStart();
    for (uint32_t i = 0; i < n && rc == ncclSuccess; i++) {
        if (owner[i] == root) continue;
        // pitched surface = one contiguous block of pitch * h bytes (row padding travels too: same pitch rule on both sides)
        if (c->rank == owner[i]) rc = r->Send(src[i]->ptr, src[i]->pitch * src[i]->h, ncclUint8, (int)root, c->nccl, me->stream);
        else if (c->rank == root) rc = r->Recv(dst[i]->ptr, dst[i]->pitch * dst[i]->h, ncclUint8, (int)owner[i], c->nccl, me->stream);
    }
    const int rc2 = r->GroupEnd();
    if (rc != ncclSuccess || rc2 != ncclSuccess)
        return comm_fail(c, me, SMR_ERR_INTERNAL, std::string("smr_gather_tiles: ") + r->GetErrorString(rc != ncclSuccess ? rc : rc2));
    return SMR_OK;
}

}  // extern "C"
