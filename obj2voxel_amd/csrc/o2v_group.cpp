// o2v_group.cpp -- one process, N GPUs: a context and a host thread per GPU, the grid sharded by z-slab.
//
// This is what one obj2voxel_voxelize() call does when more than one device is named (O2V_DEVICES), where the reference
// hands its 64^3 chunks to a pool of worker threads (src/obj2voxel.cpp:467-520, src/threading.hpp): the "workers" are
// GPUs, each owning one z-slab, and the per-rank work is o2v_hip_voxelize_sharded (o2v_device.hip).  The ranks combine
// their planning data with RCCL over xGMI; if RCCL cannot be used (a device listed twice - only sensible for tests on
// a single-GPU machine - or librccl missing) the same collectives run over shared host memory between the threads.
#include "o2v_comm.hpp"
#include "o2v_device_internal.hpp"

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace {

// Collectives between the threads of one process over host memory: every rank deposits its buffer, the last one to
// arrive reduces, everybody copies the result out.  Two barrier phases per collective.
struct SharedExchange {
    std::mutex m;
    std::condition_variable cv;
    uint32_t world = 1, arrived = 0, generation = 0;
    bool aborted = false;  // a rank failed outside a collective: nobody may wait for it any more
    std::vector<void *> bufs;

    // calls `last` on exactly one thread once all ranks have deposited `buf`; returns after it has run (false: aborted)
    bool rendezvous(uint32_t rank, void *buf, const std::function<void(std::vector<void *> &)> &last)
    {
        std::unique_lock<std::mutex> lock{m};
        if (aborted) return false;
        bufs[rank] = buf;
        const uint32_t gen = generation;
        if (++arrived == world) {
            last(bufs);
            arrived = 0;
            ++generation;
            cv.notify_all();
        }
        else {
            cv.wait(lock, [&] { return generation != gen || aborted; });
        }
        return !aborted;
    }
    void abort()
    {
        std::lock_guard<std::mutex> lock{m};
        aborted = true;
        arrived = 0;
        cv.notify_all();
    }
    void reset()
    {
        std::lock_guard<std::mutex> lock{m};
        aborted = false;
        arrived = 0;
    }
};

struct RankLink {
    SharedExchange *x;
    uint32_t rank;
};

template <typename T, typename Op>
int shared_allreduce(void *user, T *buf, size_t n, Op op)
{
    RankLink *l = static_cast<RankLink *>(user);
    return l->x->rendezvous(l->rank, buf, [&](std::vector<void *> &bufs) {
        T *first = static_cast<T *>(bufs[0]);
        for (uint32_t r = 1; r < l->x->world; ++r)
            for (size_t i = 0; i < n; ++i) first[i] = op(first[i], static_cast<T *>(bufs[r])[i]);
        for (uint32_t r = 1; r < l->x->world; ++r) std::memcpy(bufs[r], first, n * sizeof(T));
    }) ? 0 : 1;
}
int shared_min_u32(void *u, uint32_t *b, size_t n) { return shared_allreduce(u, b, n, [](uint32_t a, uint32_t c) { return std::min(a, c); }); }
int shared_max_u32(void *u, uint32_t *b, size_t n) { return shared_allreduce(u, b, n, [](uint32_t a, uint32_t c) { return std::max(a, c); }); }
int shared_sum_u64(void *u, uint64_t *b, size_t n) { return shared_allreduce(u, b, n, [](uint64_t a, uint64_t c) { return a + c; }); }
int shared_allgather(void *user, void *buf, size_t bytes)
{
    RankLink *l = static_cast<RankLink *>(user);
    return l->x->rendezvous(l->rank, buf, [&](std::vector<void *> &bufs) {
        for (uint32_t src = 0; src < l->x->world; ++src)
            for (uint32_t dst = 0; dst < l->x->world; ++dst)
                if (src != dst)
                    std::memcpy(static_cast<char *>(bufs[dst]) + src * bytes, static_cast<char *>(bufs[src]) + src * bytes, bytes);
    }) ? 0 : 1;
}
int shared_broadcast(void *user, void *buf, size_t bytes, int root)
{
    RankLink *l = static_cast<RankLink *>(user);
    return l->x->rendezvous(l->rank, buf, [&](std::vector<void *> &bufs) {
        for (uint32_t dst = 0; dst < l->x->world; ++dst)
            if ((int) dst != root) std::memcpy(bufs[dst], bufs[root], bytes);
    }) ? 0 : 1;
}

}  // namespace

struct o2v_hip_group {
    std::vector<int> devices;
    std::vector<o2v_hip_ctx *> ctx;
    std::vector<o2v_hip_comm *> comm;
    SharedExchange exchange;
    std::vector<RankLink> links;
    bool rccl = false;
    std::string err;

    // runs fn(rank) on one thread per rank (rank 0 on the caller's) and returns the first non-zero result
    int on_every_rank(const std::function<int(uint32_t)> &fn)
    {
        const uint32_t n = (uint32_t) ctx.size();
        std::vector<int> rc(n, 0);
        std::vector<std::thread> threads;
        exchange.reset();
        // a rank that fails must not leave the others waiting for it in a host-memory collective
        auto run = [&](uint32_t r) {
            rc[r] = fn(r);
            if (rc[r] && !rccl) exchange.abort();
        };
        for (uint32_t r = 1; r < n; ++r) threads.emplace_back([&, r] { run(r); });
        run(0);
        for (std::thread &t : threads) t.join();
        // every failing rank's message (the rank whose failure the others merely heard about is among them)
        int first = O2V_HIP_OK;
        std::string all;
        for (uint32_t r = 0; r < n; ++r)
            if (rc[r]) {
                if (!first) first = rc[r];
                all += (all.empty() ? "" : "; ") + ("rank " + std::to_string(r) + " (device " + std::to_string(devices[r]) + "): " + o2v_hip_last_error(ctx[r]));
            }
        if (first) err = all;
        return first;
    }
};

extern "C" {

// (lives here rather than beside o2v_hip_alloc_pinned: the device translation unit's hash identifies the kernels a profile
// was taken with, and this is host plumbing)
void *o2v_hip_alloc_pinned_on(int device, size_t bytes)
{
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    return o2v_hip_alloc_pinned(bytes);
}


// The shared-memory exchange on its own (no GPU): n threads run the callbacks self-test against each other, several rounds.
int o2v_hip_group_exchange_selftest(uint32_t n_threads)
{
    if (!n_threads) return O2V_HIP_ERR_BAD_ARGUMENT;
    SharedExchange x;
    x.world = n_threads;
    x.bufs.assign(n_threads, nullptr);
    std::vector<RankLink> links(n_threads);
    std::vector<int> rc(n_threads, 0);
    std::vector<std::thread> threads;
    for (uint32_t r = 0; r < n_threads; ++r) {
        links[r] = RankLink{&x, r};
        threads.emplace_back([&, r] {
            o2v_hip_comm_callbacks cb{&links[r], shared_min_u32, shared_max_u32, shared_sum_u64, shared_allgather, shared_broadcast};
            for (int round = 0; round < 20 && !rc[r]; ++round) rc[r] = o2v_hip_comm_callbacks_selftest(&cb, (int) r, (int) n_threads);
        });
    }
    for (std::thread &t : threads) t.join();
    for (int v : rc)
        if (v) return v;
    return 0;
}

// The RCCL code path of an in-process group without its GPUs: the unique id, one thread per rank joining the communicator
// (ncclCommInitRank blocks until every rank has), every collective of o2v_hip_comm with patterns whose result is known in
// closed form, teardown.  No device is selected and the buffers are host memory, so it needs a librccl that works on host
// memory - the threads-and-memcpy stand-in of tests/mock/mock_rccl.c, named by O2V_RCCL_LIB.  This is how the start-up,
// id exchange and teardown of an 8-rank group are executed before the first run on an 8-GPU node.  0 = everything behaved.
int o2v_hip_group_rccl_selftest(uint32_t n_ranks)
{
    if (!n_ranks) return O2V_HIP_ERR_BAD_ARGUMENT;
    uint8_t id[O2V_HIP_COMM_ID_BYTES];
    std::string err;
    if (!o2v::rccl_unique_id(id, err)) {
        std::fprintf(stderr, "[o2v] rccl selftest: %s\n", err.c_str());
        return 100;
    }
    std::vector<o2v_hip_comm *> comm(n_ranks, nullptr);
    std::vector<std::string> errs(n_ranks);
    std::vector<int> rc(n_ranks, 0);
    {
        std::vector<std::thread> threads;
        for (uint32_t r = 0; r < n_ranks; ++r)
            threads.emplace_back([&, r] { comm[r] = o2v::make_rccl_comm(id, (int) r, (int) n_ranks, -1, errs[r]); });
        for (std::thread &t : threads) t.join();
    }
    int result = 0;
    for (uint32_t r = 0; r < n_ranks; ++r)
        if (!comm[r]) {
            std::fprintf(stderr, "[o2v] rccl selftest, rank %u: %s\n", r, errs[r].c_str());
            result = 101;
        }
    if (!result) {
        std::vector<std::thread> threads;
        for (uint32_t r = 0; r < n_ranks; ++r)
            threads.emplace_back([&, r] {
                o2v_hip_comm *c = comm[r];
                const uint32_t W = n_ranks, R = r;
                for (int round = 0; round < 10 && !rc[r]; ++round) {
                    // (the same patterns as o2v_hip_comm_callbacks_selftest: bounds min / max, histogram sum, gathers, broadcast)
                    uint32_t mn[3] = {0x80000000u + R, 0xfffffff0u - R, 7u + R}, mx[3] = {0x80000000u + R, 0xfffffff0u - R, 7u + R};
                    if (c->allreduce_min_u32(mn, 3, nullptr) || c->allreduce_max_u32(mx, 3, nullptr)) { rc[r] = 1; break; }
                    if (mn[0] != 0x80000000u || mn[1] != 0xfffffff0u - (W - 1) || mn[2] != 7u) { rc[r] = 2; break; }
                    if (mx[0] != 0x80000000u + (W - 1) || mx[1] != 0xfffffff0u || mx[2] != 7u + (W - 1)) { rc[r] = 3; break; }
                    std::vector<unsigned long long> sum(2048);
                    for (size_t i = 0; i < sum.size(); ++i) sum[i] = (unsigned long long) i * 1000003ull + R + (i == 5 ? (1ull << 40) : 0);
                    if (c->allreduce_sum_u64(sum.data(), sum.size(), nullptr)) { rc[r] = 4; break; }
                    for (size_t i = 0; i < sum.size() && !rc[r]; ++i)
                        if (sum[i] != W * ((unsigned long long) i * 1000003ull + (i == 5 ? (1ull << 40) : 0)) + (unsigned long long) W * (W - 1) / 2) rc[r] = 5;
                    const size_t per = 24;
                    std::vector<unsigned char> gather(per * W, 0xee);
                    for (size_t i = 0; i < per; ++i) gather[R * per + i] = (unsigned char) (R * 31 + i);
                    if (c->allgather(gather.data(), per, nullptr)) { rc[r] = 6; break; }
                    for (uint32_t q = 0; q < W && !rc[r]; ++q)
                        for (size_t i = 0; i < per; ++i)
                            if (gather[q * per + i] != (unsigned char) (q * 31 + i)) rc[r] = 7;
                    const int root = (int) W - 1 - (round % (int) W);
                    std::vector<unsigned char> bc(1000);
                    for (size_t i = 0; i < bc.size(); ++i) bc[i] = (unsigned char) ((int) R == root ? i * 7 : 0);
                    if (c->broadcast(bc.data(), bc.size(), root, nullptr)) { rc[r] = 8; break; }
                    for (size_t i = 0; i < bc.size() && !rc[r]; ++i)
                        if (bc[i] != (unsigned char) (i * 7)) rc[r] = 9;
                }
            });
        for (std::thread &t : threads) t.join();
        for (uint32_t r = 0; r < n_ranks; ++r)
            if (rc[r] && !result) result = rc[r];
    }
    for (o2v_hip_comm *c : comm) delete c;  // ncclCommDestroy
    return result;
}

int o2v_hip_group_create(const int *devices, uint32_t n_devices, o2v_hip_group **out)
{
    if (!devices || !n_devices || !out) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out = nullptr;
    o2v_hip_group *g = new o2v_hip_group;
    g->devices.assign(devices, devices + n_devices);
    g->ctx.assign(n_devices, nullptr);
    g->comm.assign(n_devices, nullptr);
    for (uint32_t r = 0; r < n_devices; ++r) {
        const int rc = o2v_hip_create(devices[r], &g->ctx[r]);
        if (rc) {
            o2v_hip_group_destroy(g);
            return rc;
        }
    }
    // RCCL wants every rank on its own device
    const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == n_devices;
    const char *force = std::getenv("O2V_GROUP_COMM");  // "host": shared host memory even where RCCL would work
    if (n_devices > 1 && distinct && !(force && std::strcmp(force, "host") == 0)) {
        uint8_t id[O2V_HIP_COMM_ID_BYTES];
        std::string err;
        if (o2v::rccl_unique_id(id, err)) {
            // ncclCommInitRank blocks until every rank has joined: one thread per rank
            std::vector<std::string> errs(n_devices);
            std::vector<std::thread> threads;
            for (uint32_t r = 0; r < n_devices; ++r)
                threads.emplace_back([&, r] { g->comm[r] = o2v::make_rccl_comm(id, (int) r, (int) n_devices, devices[r], errs[r]); });
            for (std::thread &t : threads) t.join();
            g->rccl = std::all_of(g->comm.begin(), g->comm.end(), [](o2v_hip_comm *c) { return c != nullptr; });
            if (!g->rccl) {
                for (o2v_hip_comm *&c : g->comm) {
                    delete c;
                    c = nullptr;
                }
                std::fprintf(stderr, "[o2v] RCCL communicators could not be created (%s); the ranks exchange through host memory\n",
                             errs[0].c_str());
            }
        }
        else {
            std::fprintf(stderr, "[o2v] %s; the ranks exchange through host memory\n", err.c_str());
        }
    }
    if (!g->rccl) {
        g->exchange.world = n_devices;
        g->exchange.bufs.assign(n_devices, nullptr);
        g->links.resize(n_devices);
        for (uint32_t r = 0; r < n_devices; ++r) {
            g->links[r] = RankLink{&g->exchange, r};
            o2v_hip_comm_callbacks cb{&g->links[r], shared_min_u32, shared_max_u32, shared_sum_u64, shared_allgather, shared_broadcast};
            g->comm[r] = o2v::make_callback_comm(cb, (int) r, (int) n_devices);
        }
    }
    *out = g;
    return O2V_HIP_OK;
}

void o2v_hip_group_destroy(o2v_hip_group *g)
{
    if (!g) return;
    for (o2v_hip_comm *c : g->comm) delete c;
    for (o2v_hip_ctx *c : g->ctx)
        if (c) o2v_hip_destroy(c);
    delete g;
}

uint32_t o2v_hip_group_size(const o2v_hip_group *g) { return g ? (uint32_t) g->ctx.size() : 0u; }
o2v_hip_ctx *o2v_hip_group_ctx(o2v_hip_group *g, uint32_t rank) { return g && rank < g->ctx.size() ? g->ctx[rank] : nullptr; }
const char *o2v_hip_group_comm_kind(const o2v_hip_group *g) { return g && !g->comm.empty() && g->comm[0] ? g->comm[0]->kind() : "none"; }
const char *o2v_hip_group_last_error(const o2v_hip_group *g) { return g ? g->err.c_str() : "null group"; }

int o2v_hip_group_set_triangles(o2v_hip_group *g, const float *verts, const float *uvs, const uint32_t *types,
                                const float *colors, const int32_t *texids, uint64_t count, int upload_mode)
{
    if (!g || (count && !verts)) return O2V_HIP_ERR_BAD_ARGUMENT;
    const uint32_t n = (uint32_t) g->ctx.size();
    if (upload_mode == O2V_HIP_UPLOAD_H2D || n == 1 || count == 0) {
        // every GPU pulls the arrays over its own PCIe link, all at once
        return g->on_every_rank([&](uint32_t r) { return o2v_hip_set_triangles(g->ctx[r], verts, uvs, types, colors, texids, count); });
    }
    // one host-to-device copy, then GPU to GPU over xGMI
    int rc = o2v_hip_set_triangles(g->ctx[0], verts, uvs, types, colors, texids, count);
    if (rc) {
        g->err = std::string("rank 0: ") + o2v_hip_last_error(g->ctx[0]);
        return rc;
    }
    const o2v::TriHints hints = o2v::ctx_tri_hints(g->ctx[0]);
    const o2v::TriBuffers src = o2v::ctx_tri_buffers(g->ctx[0]);
    // Every rank allocates before any rank enters a collective: a rank whose allocation fails would otherwise return while
    // the others wait for it in the broadcast (on_every_rank reports the failure after all ranks are back).
    rc = g->on_every_rank([&](uint32_t r) -> int {
        return r == 0 ? O2V_HIP_OK
                      : o2v::ctx_alloc_triangles(g->ctx[r], count, uvs != nullptr, types != nullptr, colors != nullptr, texids != nullptr);
    });
    if (rc) return rc;
    return g->on_every_rank([&](uint32_t r) -> int {
        o2v_hip_ctx *c = g->ctx[r];
        (void) hipSetDevice(o2v::ctx_device(c));
        const o2v::TriBuffers dst = o2v::ctx_tri_buffers(c);
        hipStream_t s = o2v::ctx_stream(c);
        struct Part {
            void *dst;
            const void *src;
            size_t bytes;
        } parts[5] = {{dst.verts, src.verts, count * 36}, {dst.uvs, src.uvs, count * 24}, {dst.types, src.types, count * 4},
                      {dst.colors, src.colors, count * 12}, {dst.texids, src.texids, count * 4}};
        for (const Part &p : parts) {
            if (!p.src) continue;
            if (upload_mode == O2V_HIP_UPLOAD_BROADCAST) {
                // rank 0's buffer is the root's send buffer, every other rank's its receive buffer (collective: all ranks call)
                if (g->comm[r]->broadcast(p.dst, p.bytes, 0, s)) return O2V_HIP_ERR_HIP;
            }
            else if (r != 0) {
                if (hipMemcpyPeerAsync(p.dst, o2v::ctx_device(c), p.src, g->devices[0], p.bytes, s) != hipSuccess) return O2V_HIP_ERR_HIP;
            }
        }
        return r == 0 ? (hipStreamSynchronize(s) == hipSuccess ? O2V_HIP_OK : O2V_HIP_ERR_HIP) : o2v::ctx_finish_triangles(c, hints.any_textured, &hints);
    });
}

int o2v_hip_group_set_textures(o2v_hip_group *g, const o2v_hip_texture *textures, uint32_t count)
{
    if (!g) return O2V_HIP_ERR_BAD_ARGUMENT;
    return g->on_every_rank([&](uint32_t r) { return o2v_hip_set_textures(g->ctx[r], textures, count); });
}

int o2v_hip_group_voxelize(o2v_hip_group *g, const o2v_hip_params *params, uint64_t *out_counts, uint32_t *out_cuts)
{
    if (!g || !params || !out_counts) return O2V_HIP_ERR_BAD_ARGUMENT;
    const uint32_t n = (uint32_t) g->ctx.size();
    std::vector<uint64_t> counts((size_t) n * n);
    std::vector<uint32_t> cuts((size_t) n * (n + 1));
    const int rc = g->on_every_rank([&](uint32_t r) {
        uint64_t mine = 0;
        return o2v_hip_voxelize_sharded(g->ctx[r], n > 1 ? g->comm[r] : nullptr, params, &mine, &counts[(size_t) r * n], &cuts[(size_t) r * (n + 1)]);
    });
    if (rc) return rc;
    for (uint32_t r = 0; r < n; ++r) out_counts[r] = counts[r];  // every rank gathered the same list; rank 0's copy
    if (out_cuts)
        for (uint32_t r = 0; r <= n; ++r) out_cuts[r] = cuts[r];
    return O2V_HIP_OK;
}

}  // extern "C"
