// lv_multi.hip -- one frame over the GPUs of a node, behind the C-ABI (SURVEY.md 8b "Multi-GPU = one context per device,
// driven by one host thread ... issuing async work", 8e).  No torch, no second process:
//
//   lv_create_multi(devices[n])  one lv_ctx per device ("ranks"; rank 0 = the handle the caller holds) + one RCCL communicator
//                                per rank (ncclCommInitAll, librccl resolved with dlopen so that the library still loads where
//                                RCCL is not installed).  Every setter of the C-ABI called on the handle is repeated on the
//                                other ranks: each GPU holds the full scene replica + its own LBVH (5 M segments = 0.8 GB of 288 GB).
//   lv_render* on the handle     the requested pixels are cut into tiles (64 x 64 along a Morton order for a rectangle, or the
//                                caller's tile list), the tiles are dealt over the ranks -- round robin, or by the measured cost
//                                of the previous frame (lv_multi_rebalance: RTAO hit pixels per tile, longest processing time
//                                first) --, every rank renders its list with the usual kernels on its own stream, and ONE
//                                gather brings the RGBA8 tiles to rank 0: an ncclSend / ncclRecv group over xGMI (8.3 MB per
//                                1080p frame in total), followed by one scatter kernel that puts every tile where the caller
//                                asked for it.  Pixels are independent (seeds use global pixel coordinates), so the frame is
//                                byte-identical to the single-GPU frame whatever the deal.
//   transport "memcpy"           hipMemcpyPeerAsync + events instead of RCCL: the fallback without librccl, and the way to run
//                                several ranks on ONE device (RCCL refuses duplicate devices) -- tests use it to exercise deal,
//                                gather order and scatter with 2-4 ranks on the single test GPU, next to a 1-rank RCCL run.
// The reference is single-GPU (sgl::AppSettings::getPrimaryDevice() everywhere); this file has no counterpart there.
#include <dlfcn.h>

#include <algorithm>
#include <numeric>

#include <rccl/rccl.h>

#include "lv_internal.h"

struct LvRccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct LvMulti {
    std::vector<lv_ctx*> ranks;                 // ranks[0] = the handle
    bool rcclTransport = true;
    LvRccl rccl;
    std::vector<ncclComm_t> comms;
    // the deal of the last tile list
    std::vector<uint32_t> tiles;                // 2 per tile, the caller's order
    uint32_t tileW = 0, tileH = 0;
    std::vector<uint32_t> owner;                // tile -> rank
    std::vector<std::vector<uint32_t>> owned;   // rank -> tile indices (ascending)
    std::vector<std::vector<uint32_t>> ownedXY; // rank -> origins of its tiles
    bool dealByCost = false;
    // buffers: per rank its tile-major output; rank 0 the rank-major concatenation + the scatter table + the image
    std::vector<LvDeviceBuffer> tileBuf;
    LvDeviceBuffer gatherBuf, scatterTable, imageBuf;
    std::vector<uint32_t> scatterHost;
    bool scatterUploaded = false;
    std::vector<hipEvent_t> evRendered;         // memcpy transport: rank r's copy has been queued behind its kernels
    hipEvent_t evConsumed = nullptr;            // rank 0 has scattered the previous frame out of gatherBuf
    bool consumedValid = false;
    uint64_t frames = 0;
};

namespace {

// scatter: gathered tile g (rank-major order) -> where the caller wants it.  table[3 g] = tile origin x, y; table[3 g + 2] = index in
// the caller's tile list.  IMAGE: into the (x0, y0, w, h) rectangle, row-major; else tile-major at the caller's index.
template <bool IMAGE>
__global__ __launch_bounds__(LV_BLOCK) void k_scatter_tiles(const uint32_t* __restrict__ gathered, const uint32_t* __restrict__ table,
                                                            uint32_t numTiles, uint32_t tileW, uint32_t tileH, uint32_t x0, uint32_t y0,
                                                            uint32_t w, uint32_t h, uint32_t* __restrict__ out) {
    const uint64_t gid = uint64_t(blockIdx.x) * LV_BLOCK + threadIdx.x;
    const uint32_t per = tileW * tileH;
    if (gid >= uint64_t(numTiles) * per) return;
    const uint32_t g = uint32_t(gid / per), rem = uint32_t(gid % per);
    const uint32_t px = rem % tileW, py = rem / tileW;
    const uint32_t v = gathered[gid];
    if (IMAGE) {
        const uint32_t x = table[3 * g] + px, y = table[3 * g + 1] + py;
        if (x >= x0 && y >= y0 && x - x0 < w && y - y0 < h) out[size_t(y - y0) * w + (x - x0)] = v;
    } else {
        out[size_t(table[3 * g + 2]) * per + rem] = v;
    }
}

inline uint64_t part1by1(uint64_t v) {
    v &= 0xFFFFFFFFull;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

int loadRccl(lv_ctx* ctx, LvRccl& R) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (R.lib) break;
    }
    if (!R.lib) return lv_fail(ctx, LV_E_STATE, "librccl not found (%s): use transport \"memcpy\"", dlerror());
#define LV_SYM(field, sym)                                                                              \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.lib, sym));                                   \
    if (!R.field) return lv_fail(ctx, LV_E_STATE, "librccl has no symbol %s", sym)
    LV_SYM(CommInitAll, "ncclCommInitAll");
    LV_SYM(CommDestroy, "ncclCommDestroy");
    LV_SYM(GroupStart, "ncclGroupStart");
    LV_SYM(GroupEnd, "ncclGroupEnd");
    LV_SYM(Send, "ncclSend");
    LV_SYM(Recv, "ncclRecv");
    LV_SYM(GetErrorString, "ncclGetErrorString");
#undef LV_SYM
    return LV_OK;
}

#define LV_NCCL(ctx, M, expr)                                                                                     \
    do {                                                                                                          \
        ncclResult_t _r = (expr);                                                                                 \
        if (_r != ncclSuccess) return lv_fail(ctx, LV_E_HIP, "%s failed: %s", #expr, (M)->rccl.GetErrorString(_r)); \
    } while (0)

} // namespace

// Longest processing time first: tiles in order of falling cost, each to the rank with the least cost so far (ties: fewer tiles,
// then lower rank); equal costs keep the list order, so costs == nullptr gives the round-robin deal.  Deterministic.
extern "C" int lv_tile_deal(const double* costs, uint32_t num_tiles, uint32_t num_ranks, uint32_t* out_owner) {
    if (!out_owner || num_ranks == 0) return LV_E_INVALID;
    if (!costs) {
        for (uint32_t i = 0; i < num_tiles; i++) out_owner[i] = i % num_ranks;
        return LV_OK;
    }
    std::vector<uint32_t> order(num_tiles);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return costs[a] > costs[b]; });
    std::vector<double> load(num_ranks, 0.0);
    std::vector<uint32_t> count(num_ranks, 0u);
    for (uint32_t t : order) {
        uint32_t best = 0;
        for (uint32_t r = 1; r < num_ranks; r++)
            if (load[r] < load[best] || (load[r] == load[best] && count[r] < count[best])) best = r;
        out_owner[t] = best;
        load[best] += costs[t];
        count[best]++;
    }
    return LV_OK;
}

// Origins of the tile x tile squares covering the rectangle, along a Morton order of the tile grid (neighbouring tiles land on
// different ranks under the round-robin deal, and every rank gets a share of the dense middle of the picture).
extern "C" uint32_t lv_make_tiles(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t tile, uint32_t* out_xy, uint32_t capacity) {
    if (tile == 0 || w == 0 || h == 0) return 0;
    const uint32_t nx = (w + tile - 1) / tile, ny = (h + tile - 1) / tile;
    const uint64_t n = uint64_t(nx) * ny;
    if (!out_xy || capacity < n) return uint32_t(n);
    std::vector<std::pair<uint64_t, uint32_t>> keys;
    keys.reserve(n);
    for (uint32_t gy = 0; gy < ny; gy++)
        for (uint32_t gx = 0; gx < nx; gx++) keys.push_back({part1by1(gx) | (part1by1(gy) << 1), gy * nx + gx});
    std::stable_sort(keys.begin(), keys.end());
    for (size_t i = 0; i < keys.size(); i++) {
        out_xy[2 * i] = x0 + (keys[i].second % nx) * tile;
        out_xy[2 * i + 1] = y0 + (keys[i].second / nx) * tile;
    }
    return uint32_t(n);
}

int lv_multi_create(lv_ctx* handle, const int* devices, int numDevices, const char* transport) {
    LvMulti* M = new LvMulti();
    handle->multi = M;
    M->ranks.push_back(handle);
    M->rcclTransport = !(transport && std::string(transport) == "memcpy");
    // per-rank arrays sized before anything can fail: lv_multi_destroy walks them for every rank created so far
    M->tileBuf.resize(size_t(numDevices));
    M->owned.resize(size_t(numDevices));
    M->ownedXY.resize(size_t(numDevices));
    M->evRendered.assign(size_t(numDevices), nullptr);
    for (int r = 1; r < numDevices; r++) {
        int err = 0;
        lv_ctx* p = lv_create(devices[r], &err);
        if (!p) return lv_fail(handle, err ? err : LV_E_HIP, "lv_create failed for device %d", devices[r]);
        M->ranks.push_back(p);
    }
    const size_t n = M->ranks.size();
    for (size_t r = 0; r < n; r++) {
        LV_HIP(handle, hipSetDevice(M->ranks[r]->device));
        LV_HIP(handle, hipEventCreateWithFlags(&M->evRendered[r], hipEventDisableTiming));
    }
    LV_HIP(handle, hipSetDevice(handle->device));
    LV_HIP(handle, hipEventCreateWithFlags(&M->evConsumed, hipEventDisableTiming));
    if (M->rcclTransport) {
        int rc;
        if ((rc = loadRccl(handle, M->rccl))) return rc;
        for (size_t a = 0; a < n; a++)
            for (size_t b = a + 1; b < n; b++)
                if (M->ranks[a]->device == M->ranks[b]->device)
                    return lv_fail(handle, LV_E_INVALID, "device %d appears twice: RCCL needs distinct devices (transport \"memcpy\" allows it)",
                                   M->ranks[a]->device);
        std::vector<int> devs(n);
        for (size_t r = 0; r < n; r++) devs[r] = M->ranks[r]->device;
        M->comms.assign(n, nullptr);
        LV_NCCL(handle, M, M->rccl.CommInitAll(M->comms.data(), int(n), devs.data()));
    } else {
        // peer access for the direct copies (ignored where it is already enabled / the same device)
        for (size_t r = 1; r < n; r++) {
            if (M->ranks[r]->device == handle->device) continue;
            (void)hipSetDevice(handle->device);
            (void)hipDeviceEnablePeerAccess(M->ranks[r]->device, 0);
            (void)hipSetDevice(M->ranks[r]->device);
            (void)hipDeviceEnablePeerAccess(handle->device, 0);
        }
        (void)hipGetLastError();
    }
    return LV_OK;
}

void lv_multi_destroy(lv_ctx* handle) {
    LvMulti* M = handle->multi;
    if (!M) return;
    for (lv_ctx* c : M->ranks) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    if (M->rcclTransport && M->rccl.CommDestroy)
        for (ncclComm_t c : M->comms)
            if (c) (void)M->rccl.CommDestroy(c);
    for (size_t r = 0; r < M->ranks.size(); r++) {
        (void)hipSetDevice(M->ranks[r]->device);
        lv_buf_free(M->tileBuf[r]);
        if (M->evRendered[r]) (void)hipEventDestroy(M->evRendered[r]);
    }
    (void)hipSetDevice(handle->device);
    lv_buf_free(M->gatherBuf);
    lv_buf_free(M->scatterTable);
    lv_buf_free(M->imageBuf);
    if (M->evConsumed) (void)hipEventDestroy(M->evConsumed);
    for (size_t r = 1; r < M->ranks.size(); r++) lv_destroy(M->ranks[r]);
    handle->multi = nullptr;
    delete M;
}

int lv_multi_num_ranks(const lv_ctx* handle) { return handle->multi ? int(handle->multi->ranks.size()) : 1; }
lv_ctx* lv_multi_rank(lv_ctx* handle, int r) {
    if (!handle->multi) return r == 0 ? handle : nullptr;
    return (r >= 0 && size_t(r) < handle->multi->ranks.size()) ? handle->multi->ranks[size_t(r)] : nullptr;
}

static void applyDeal(LvMulti* M, const std::vector<uint32_t>& owner) {
    const size_t n = M->ranks.size();
    M->owner = owner;
    for (size_t r = 0; r < n; r++) { M->owned[r].clear(); M->ownedXY[r].clear(); }
    for (uint32_t i = 0; i < owner.size(); i++) {
        M->owned[owner[i]].push_back(i);
        M->ownedXY[owner[i]].push_back(M->tiles[2 * i]);
        M->ownedXY[owner[i]].push_back(M->tiles[2 * i + 1]);
    }
    // scatter table in gather (rank-major) order
    M->scatterHost.clear();
    for (size_t r = 0; r < n; r++)
        for (uint32_t i : M->owned[r]) {
            M->scatterHost.push_back(M->tiles[2 * i]);
            M->scatterHost.push_back(M->tiles[2 * i + 1]);
            M->scatterHost.push_back(i);
        }
    M->scatterUploaded = false;
}

// One frame over the ranks.  image == true: `out` (device memory of rank 0) is the (x0, y0, w, h) rectangle, row-major; else the
// caller's tile-major layout.
int lv_multi_render(lv_ctx* handle, int mode, const uint32_t* tilesXY, uint32_t numTiles, uint32_t tileW, uint32_t tileH, bool image,
                    uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, void* outDevice) {
    LvMulti* M = handle->multi;
    const size_t n = M->ranks.size();
    const size_t tileBytes = size_t(tileW) * tileH * 4;
    // a new tile list gets a fresh round-robin deal; the same list keeps its (possibly cost-weighted) deal
    const bool same = M->tiles.size() == 2 * size_t(numTiles) && M->tileW == tileW && M->tileH == tileH &&
                      std::equal(M->tiles.begin(), M->tiles.end(), tilesXY);
    if (!same) {
        M->tiles.assign(tilesXY, tilesXY + 2 * size_t(numTiles));
        M->tileW = tileW;
        M->tileH = tileH;
        std::vector<uint32_t> owner(numTiles);
        lv_tile_deal(nullptr, numTiles, uint32_t(n), owner.data());
        applyDeal(M, owner);
        M->dealByCost = false;
    }
    int rc;
    LV_HIP(handle, hipSetDevice(handle->device));
    if ((rc = lv_buf_reserve(handle, M->gatherBuf, size_t(numTiles) * tileBytes))) return rc;
    if ((rc = lv_buf_reserve(handle, M->scatterTable, M->scatterHost.size() * 4))) return rc;
    if (!M->scatterUploaded) {
        LV_HIP(handle, hipStreamSynchronize(handle->stream)); // the previous table may still be read
        LV_HIP(handle, hipMemcpyAsync(M->scatterTable.ptr, M->scatterHost.data(), M->scatterHost.size() * 4, hipMemcpyHostToDevice,
                                      handle->stream));
        LV_HIP(handle, hipStreamSynchronize(handle->stream));
        M->scatterUploaded = true;
    }
    // ---- every rank renders its tiles on its own stream (asynchronous: the host only queues work)
    for (size_t r = 0; r < n; r++) {
        lv_ctx* c = M->ranks[r];
        const uint32_t cnt = uint32_t(M->owned[r].size());
        if (cnt == 0) {
            // A rank without a tile in this frame (fewer tiles than ranks) still has temporal state that must stay in step with the
            // other ranks': the RTAO seed counter aoGlobalFrameNumber, the SVGF history (every rank denoises the whole viewport),
            // lastFrameViewProj.  It renders the list's first tile into its own buffer; the result is not gathered.
            if (numTiles == 0) continue;
            LV_HIP(handle, hipSetDevice(c->device));
            if ((rc = lv_buf_reserve(c, M->tileBuf[r], tileBytes))) return lv_fail(handle, rc, "rank %zu: %s", r, c->lastError.c_str());
            if ((rc = lv_frame_render(c, mode, tilesXY, 1u, tileW, tileH, M->tileBuf[r].ptr)))
                return lv_fail(handle, rc, "rank %zu (device %d): %s", r, c->device, c->lastError.c_str());
            continue;
        }
        LV_HIP(handle, hipSetDevice(c->device));
        if ((rc = lv_buf_reserve(c, M->tileBuf[r], size_t(cnt) * tileBytes))) return lv_fail(handle, rc, "rank %zu: %s", r, c->lastError.c_str());
        if ((rc = lv_frame_render(c, mode, M->ownedXY[r].data(), cnt, tileW, tileH, M->tileBuf[r].ptr)))
            return r == 0 ? rc : lv_fail(handle, rc, "rank %zu (device %d): %s", r, c->device, c->lastError.c_str());
    }
    // ---- ONE gather of RGBA8 tiles to rank 0
    char* gather = (char*)M->gatherBuf.ptr;
    if (M->rcclTransport) {
        LV_NCCL(handle, M, M->rccl.GroupStart());
        size_t off = 0;
        for (size_t r = 0; r < n; r++) {
            const size_t bytes = M->owned[r].size() * tileBytes;
            if (bytes == 0) continue;
            LV_NCCL(handle, M, M->rccl.Send(M->tileBuf[r].ptr, bytes, ncclUint8, 0, M->comms[r], M->ranks[r]->stream));
            LV_NCCL(handle, M, M->rccl.Recv(gather + off, bytes, ncclUint8, int(r), M->comms[0], handle->stream));
            off += bytes;
        }
        LV_NCCL(handle, M, M->rccl.GroupEnd());
    } else {
        size_t off = 0;
        for (size_t r = 0; r < n; r++) {
            lv_ctx* c = M->ranks[r];
            const size_t bytes = M->owned[r].size() * tileBytes;
            if (bytes == 0) continue;
            LV_HIP(handle, hipSetDevice(c->device));
            // gatherBuf must have been consumed by the previous frame's scatter before it is overwritten
            if (M->consumedValid && c != handle) LV_HIP(handle, hipStreamWaitEvent(c->stream, M->evConsumed, 0));
            LV_HIP(handle, hipMemcpyPeerAsync(gather + off, handle->device, M->tileBuf[r].ptr, c->device, bytes, c->stream));
            if (c != handle) {
                LV_HIP(handle, hipEventRecord(M->evRendered[r], c->stream));
                LV_HIP(handle, hipStreamWaitEvent(handle->stream, M->evRendered[r], 0));
            }
            off += bytes;
        }
    }
    // ---- scatter on rank 0
    LV_HIP(handle, hipSetDevice(handle->device));
    const uint64_t threads = uint64_t(numTiles) * tileW * tileH;
    const uint32_t blocks = uint32_t((threads + LV_BLOCK - 1) / LV_BLOCK);
    if (image)
        k_scatter_tiles<true><<<blocks, LV_BLOCK, 0, handle->stream>>>((const uint32_t*)M->gatherBuf.ptr, (const uint32_t*)M->scatterTable.ptr,
                                                                       numTiles, tileW, tileH, x0, y0, w, h, (uint32_t*)outDevice);
    else
        k_scatter_tiles<false><<<blocks, LV_BLOCK, 0, handle->stream>>>((const uint32_t*)M->gatherBuf.ptr, (const uint32_t*)M->scatterTable.ptr,
                                                                        numTiles, tileW, tileH, 0, 0, 0, 0, (uint32_t*)outDevice);
    LV_HIP(handle, hipGetLastError());
    LV_HIP(handle, hipEventRecord(M->evConsumed, handle->stream));
    M->consumedValid = true;
    M->frames++;
    return LV_OK;
}

// Rectangle -> 64 x 64 tiles along a Morton order -> lv_multi_render -> row-major image.
int lv_multi_render_rect(lv_ctx* handle, int mode, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, void* outDevice) {
    const uint32_t T = 64;
    const uint32_t n = lv_make_tiles(x0, y0, w, h, T, nullptr, 0);
    std::vector<uint32_t> xy(2 * size_t(n));
    lv_make_tiles(x0, y0, w, h, T, xy.data(), n);
    return lv_multi_render(handle, mode, xy.data(), n, T, T, true, x0, y0, w, h, outDevice);
}

// Re-deal the tiles of the last frame by measured cost: per tile the RTAO hit pixels (x samples) its rank counted + a fixed cost
// per tile (the primary rays), longest processing time first.  Synchronises every rank (read-backs): call it between frames.
int lv_multi_rebalance_impl(lv_ctx* handle, double baseCostPerTile) {
    LvMulti* M = handle->multi;
    if (!M) return LV_OK;
    if (M->tiles.empty()) return lv_fail(handle, LV_E_STATE, "lv_multi_rebalance: nothing rendered yet");
    const uint32_t numTiles = uint32_t(M->tiles.size() / 2);
    std::vector<double> cost(numTiles, 0.0);
    bool any = false;
    for (size_t r = 0; r < M->ranks.size(); r++) {
        lv_ctx* c = M->ranks[r];
        if (M->owned[r].empty()) continue;
        uint32_t n = 0, per = 0;
        int rc = lv_get_ao_tile_costs(c, nullptr, 0, &n, &per);   // how many counters (tiles with an EAW halo have more groups)
        if (rc == LV_E_STATE) continue; // the last frame ran no RTAO pass: nothing to weigh, keep the fixed cost
        if (rc) return lv_fail(handle, rc, "rank %zu: %s", r, c->lastError.c_str());
        std::vector<uint32_t> counts(std::max<size_t>(n, 1));
        rc = lv_get_ao_tile_costs(c, counts.data(), uint32_t(counts.size()), &n, &per);
        if (rc == LV_E_CAPACITY || rc == LV_E_STATE) continue;     // no usable cost data for this rank: its tiles keep the fixed cost
        if (rc) return lv_fail(handle, rc, "rank %zu: %s", r, c->lastError.c_str());
        if (per == 0 || n != per * M->owned[r].size()) continue;
        for (size_t i = 0; i < M->owned[r].size(); i++) {
            double s = 0.0;
            for (uint32_t k = 0; k < per; k++) s += double(counts[i * per + k]);
            cost[M->owned[r][i]] = s * double(c->opt.aoSamplesPerFrame);
            any = true;
        }
    }
    for (double& c : cost) c += baseCostPerTile;
    std::vector<uint32_t> owner(numTiles);
    lv_tile_deal(any ? cost.data() : nullptr, numTiles, uint32_t(M->ranks.size()), owner.data());
    for (lv_ctx* c : M->ranks) {
        (void)hipSetDevice(c->device);
        LV_HIP(handle, hipStreamSynchronize(c->stream));
    }
    applyDeal(M, owner);
    M->dealByCost = any;
    return LV_OK;
}

int lv_multi_get_deal(lv_ctx* handle, uint32_t* outOwner, uint32_t capacity, uint32_t* outCount) {
    LvMulti* M = handle->multi;
    const uint32_t n = M ? uint32_t(M->owner.size()) : 0u;
    if (outCount) *outCount = n;
    if (outOwner) {
        if (capacity < n) return lv_fail(handle, LV_E_CAPACITY, "the deal has %u tiles", n);
        for (uint32_t i = 0; i < n; i++) outOwner[i] = M->owner[i];
    }
    return LV_OK;
}
