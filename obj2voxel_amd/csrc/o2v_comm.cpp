// o2v_comm.cpp -- RCCL-backed and callback-backed implementations of o2v_hip_comm (see o2v_comm.hpp).
#include "o2v_comm.hpp"

#include <rccl/rccl.h>

#include <cstdlib>
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace {

// ---- librccl, loaded on first use ------------------------------------------------------------------------------
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;   // (optional)
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // A process that already carries an RCCL must keep using that one: torch ships its own copy under the plain
        // name "librccl.so", and dlopen by that name returns the loaded object.  RTLD_LOCAL: the symbols of whatever is
        // loaded here must not capture the nccl* references of libraries loaded later.
        // O2V_RCCL_LIB=<path or soname>: load that library instead (also how the tests force the failure path)
        const char *forced = std::getenv("O2V_RCCL_LIB");
        std::string last_error;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) {
            if (forced && forced[0]) name = forced;
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            const char *e = dlerror();  // (one call: dlerror() clears the state it returns)
            last_error = e ? e : "unknown error";
            if (forced && forced[0]) break;
        }
        if (!api.handle) {
            api.err = std::string("librccl could not be loaded: ") + last_error;
            return;
        }
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(api.handle, n);
            if (!p) {
                ok = false;
                api.err = std::string("librccl lacks ") + n;
            }
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.handle, "ncclCommAbort"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) {
            dlclose(api.handle);
            api.handle = nullptr;
        }
    });
    return api.handle ? &api : nullptr;
}

struct RcclComm final : o2v_hip_comm {
    RcclApi *api = nullptr;
    ncclComm_t comm = nullptr;
    int device = 0;

    ~RcclComm() override
    {
        if (comm) {
            if (device >= 0) (void) hipSetDevice(device);
            // (ncclCommDestroy waits for the communicator's outstanding work: with a collective stuck on a stream it would never return)
            if (!poisoned) (void) api->CommDestroy(comm);
            else if (api->CommAbort) (void) api->CommAbort(comm);
        }
    }
    const char *kind() const override { return "rccl"; }
    int check(ncclResult_t r, const char *what)
    {
        if (r == ncclSuccess) return O2V_HIP_OK;
        err = std::string(what) + ": " + api->GetErrorString(r);
        return O2V_HIP_ERR_HIP;
    }
    int allreduce_min_u32(uint32_t *d, size_t n, hipStream_t s) override
    {
        return check(api->AllReduce(d, d, n, ncclUint32, ncclMin, comm, s), "ncclAllReduce(min)");
    }
    int allreduce_max_u32(uint32_t *d, size_t n, hipStream_t s) override
    {
        return check(api->AllReduce(d, d, n, ncclUint32, ncclMax, comm, s), "ncclAllReduce(max)");
    }
    int allreduce_sum_u64(unsigned long long *d, size_t n, hipStream_t s) override
    {
        return check(api->AllReduce(d, d, n, ncclUint64, ncclSum, comm, s), "ncclAllReduce(sum)");
    }
    int allgather(void *d, size_t bytes_per_rank, hipStream_t s) override
    {
        // in place: the send buffer is this rank's part of the receive buffer
        const char *mine = static_cast<const char *>(d) + (size_t) rank * bytes_per_rank;
        return check(api->AllGather(mine, d, bytes_per_rank, ncclUint8, comm, s), "ncclAllGather");
    }
    int broadcast(void *d, size_t bytes, int root, hipStream_t s) override
    {
        return check(api->Broadcast(d, d, bytes, ncclUint8, root, comm, s), "ncclBroadcast");
    }
};

// ---- host callbacks ------------------------------------------------------------------------------------------------
struct CallbackComm final : o2v_hip_comm {
    o2v_hip_comm_callbacks cb{};
    std::vector<unsigned char> host;

    const char *kind() const override { return "callbacks"; }
    template <typename F>
    int staged(void *d, size_t bytes, hipStream_t s, const char *what, F &&op)
    {
        host.resize(bytes);
        if (hipMemcpyAsync(host.data(), d, bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            err = std::string(what) + ": device to host copy failed";
            return O2V_HIP_ERR_HIP;
        }
        if (op(host.data()) != 0) {
            err = std::string(what) + ": the collective callback failed";
            return O2V_HIP_ERR_HIP;
        }
        if (hipMemcpyAsync(d, host.data(), bytes, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            err = std::string(what) + ": host to device copy failed";
            return O2V_HIP_ERR_HIP;
        }
        return O2V_HIP_OK;
    }
    int allreduce_min_u32(uint32_t *d, size_t n, hipStream_t s) override
    {
        return staged(d, n * 4, s, "allreduce_min_u32", [&](void *h) { return cb.allreduce_min_u32(cb.user, static_cast<uint32_t *>(h), n); });
    }
    int allreduce_max_u32(uint32_t *d, size_t n, hipStream_t s) override
    {
        return staged(d, n * 4, s, "allreduce_max_u32", [&](void *h) { return cb.allreduce_max_u32(cb.user, static_cast<uint32_t *>(h), n); });
    }
    int allreduce_sum_u64(unsigned long long *d, size_t n, hipStream_t s) override
    {
        return staged(d, n * 8, s, "allreduce_sum_u64", [&](void *h) { return cb.allreduce_sum_u64(cb.user, static_cast<uint64_t *>(h), n); });
    }
    int allgather(void *d, size_t bytes_per_rank, hipStream_t s) override
    {
        return staged(d, bytes_per_rank * (size_t) world, s, "allgather", [&](void *h) { return cb.allgather(cb.user, h, bytes_per_rank); });
    }
    int broadcast(void *d, size_t bytes, int root, hipStream_t s) override
    {
        return staged(d, bytes, s, "broadcast", [&](void *h) { return cb.broadcast(cb.user, h, bytes, root); });
    }
};

}  // namespace

namespace o2v {

bool rccl_unique_id(uint8_t id[O2V_HIP_COMM_ID_BYTES], std::string &err)
{
    static_assert(sizeof(ncclUniqueId) == O2V_HIP_COMM_ID_BYTES, "unique id size");
    RcclApi *api = rccl_api();
    if (!api) {
        err = rccl_api() ? "" : "librccl is not available";
        return false;
    }
    ncclUniqueId uid;
    const ncclResult_t r = api->GetUniqueId(&uid);
    if (r != ncclSuccess) {
        err = std::string("ncclGetUniqueId: ") + api->GetErrorString(r);
        return false;
    }
    std::memcpy(id, &uid, sizeof(uid));
    return true;
}

o2v_hip_comm *make_rccl_comm(const uint8_t id[O2V_HIP_COMM_ID_BYTES], int rank, int world, int device, std::string &err)
{
    RcclApi *api = rccl_api();
    if (!api) {
        err = "librccl is not available";
        return nullptr;
    }
    // (device < 0: no device is selected - the host-only self-test of this code path, o2v_hip_group_rccl_selftest)
    if (device >= 0 && hipSetDevice(device) != hipSuccess) {
        err = "hipSetDevice failed";
        return nullptr;
    }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    RcclComm *c = new RcclComm;
    c->api = api;
    c->rank = rank;
    c->world = world;
    c->device = device;
    // ncclCommInitRank returns when all `world` ranks have called it with the same id.  A node where one rank never does (a
    // process that died, two ranks given the same GPU, another id) would leave the others waiting for ever: the call runs on a
    // thread of its own and is given o2v::comm_timeout_seconds() to return; after that the rank fails with a message (the
    // thread stays behind, blocked in RCCL, until the process ends).
    struct Init {
        std::mutex m;
        std::condition_variable cv;
        bool done = false;
        ncclResult_t result = ncclSuccess;
        ncclComm_t comm = nullptr;
    };
    auto st = std::make_shared<Init>();
    try {
        std::thread([st, api, uid, world, rank, device] {
            if (device >= 0) (void) hipSetDevice(device);  // (a new thread's current device is device 0)
            ncclComm_t comm = nullptr;
            const ncclResult_t r = api->CommInitRank(&comm, world, uid, rank);
            std::lock_guard<std::mutex> lock(st->m);
            st->result = r;
            st->comm = comm;
            st->done = true;
            st->cv.notify_all();
        }).detach();
    }
    catch (const std::system_error &) {
        // (no thread to be had: call it here, without the guard)
        st->result = api->CommInitRank(&st->comm, world, uid, rank);
        st->done = true;
    }
    {
        std::unique_lock<std::mutex> lock(st->m);
        const double limit = o2v::comm_timeout_seconds();
        if (!st->cv.wait_for(lock, std::chrono::duration<double>(limit), [&] { return st->done; })) {
            err = "ncclCommInitRank did not return within " + std::to_string((int) limit) + " s (rank " + std::to_string(rank) + " of " + std::to_string(world) +
                  "): is every rank of the job running, each on its own GPU, with rank 0's unique id?  (O2V_COMM_TIMEOUT_S sets the limit)";
            delete c;
            return nullptr;
        }
    }
    if (st->result != ncclSuccess) {
        err = std::string("ncclCommInitRank: ") + api->GetErrorString(st->result);
        c->comm = nullptr;
        delete c;
        return nullptr;
    }
    c->comm = st->comm;
    return c;
}

double comm_timeout_seconds()
{
    if (const char *e = std::getenv("O2V_COMM_TIMEOUT_S")) {
        const double v = std::atof(e);
        if (v > 0.0) return v;
    }
    return 120.0;
}

// hipStreamSynchronize with a limit: false (and `err` set) if the stream's work - a collective that waits for a rank that never
// arrives - is still pending after comm_timeout_seconds()
bool stream_wait_limited(hipStream_t s, const char *what, std::string &err)
{
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = comm_timeout_seconds();
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return true;
        if (q != hipErrorNotReady) {
            err = std::string(what) + ": " + hipGetErrorString(q);
            return false;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            err = std::string(what) + " did not complete within " + std::to_string((int) limit) +
                  " s: a rank of the job has not reached it (O2V_COMM_TIMEOUT_S sets the limit)";
            return false;
        }
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));  // (the first ~ms is polled: the usual wait is microseconds)
    }
}

o2v_hip_comm *make_callback_comm(const o2v_hip_comm_callbacks &cb, int rank, int world)
{
    CallbackComm *c = new CallbackComm;
    c->cb = cb;
    c->rank = rank;
    c->world = world;
    return c;
}

}  // namespace o2v

extern "C" {

int o2v_hip_comm_unique_id(uint8_t id[O2V_HIP_COMM_ID_BYTES])
{
    std::string err;
    if (!id) return O2V_HIP_ERR_BAD_ARGUMENT;
    return o2v::rccl_unique_id(id, err) ? O2V_HIP_OK : O2V_HIP_ERR_HIP;
}

int o2v_hip_comm_create_rccl(const uint8_t id[O2V_HIP_COMM_ID_BYTES], int rank, int world, int device, o2v_hip_comm **out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return O2V_HIP_ERR_BAD_ARGUMENT;
    std::string err;
    *out = o2v::make_rccl_comm(id, rank, world, device, err);
    if (!*out) std::fprintf(stderr, "[o2v] RCCL communicator: %s\n", err.c_str());
    return *out ? O2V_HIP_OK : O2V_HIP_ERR_HIP;
}

int o2v_hip_comm_create_callbacks(const o2v_hip_comm_callbacks *cb, int rank, int world, o2v_hip_comm **out)
{
    if (!cb || !out || world < 1 || rank < 0 || rank >= world || !cb->allreduce_min_u32 || !cb->allreduce_max_u32 ||
        !cb->allreduce_sum_u64 || !cb->allgather || !cb->broadcast)
        return O2V_HIP_ERR_BAD_ARGUMENT;
    *out = o2v::make_callback_comm(*cb, rank, world);
    return O2V_HIP_OK;
}

// Drives every callback with patterns whose result is known in closed form (host memory only: no GPU needed).
int o2v_hip_comm_callbacks_selftest(const o2v_hip_comm_callbacks *cb, int rank, int world)
{
    if (!cb || world < 1 || rank < 0 || rank >= world) return O2V_HIP_ERR_BAD_ARGUMENT;
    const uint32_t W = (uint32_t) world, R = (uint32_t) rank;
    // min / max of uint32 values with the top bit set (the order-preserving float encoding has it for positive floats)
    uint32_t mn[3] = {0x80000000u + R, 0xfffffff0u - R, 7u + R}, mx[3] = {0x80000000u + R, 0xfffffff0u - R, 7u + R};
    if (cb->allreduce_min_u32(cb->user, mn, 3) || cb->allreduce_max_u32(cb->user, mx, 3)) return 1;
    if (mn[0] != 0x80000000u || mn[1] != 0xfffffff0u - (W - 1) || mn[2] != 7u) return 2;
    if (mx[0] != 0x80000000u + (W - 1) || mx[1] != 0xfffffff0u || mx[2] != 7u + (W - 1)) return 3;
    std::vector<uint64_t> sum(2048);
    for (size_t i = 0; i < sum.size(); ++i) sum[i] = (uint64_t) i * 1000003ull + R + (i == 5 ? (1ull << 40) : 0);
    if (cb->allreduce_sum_u64(cb->user, sum.data(), sum.size())) return 4;
    for (size_t i = 0; i < sum.size(); ++i)
        if (sum[i] != W * ((uint64_t) i * 1000003ull + (i == 5 ? (1ull << 40) : 0)) + (uint64_t) W * (W - 1) / 2) return 5;
    const size_t per = 24;
    std::vector<unsigned char> gather(per * W, 0xee);
    for (size_t i = 0; i < per; ++i) gather[R * per + i] = (unsigned char) (R * 31 + i);
    if (cb->allgather(cb->user, gather.data(), per)) return 6;
    for (uint32_t r = 0; r < W; ++r)
        for (size_t i = 0; i < per; ++i)
            if (gather[r * per + i] != (unsigned char) (r * 31 + i)) return 7;
    const int root = world - 1;
    std::vector<unsigned char> bc(1000);
    for (size_t i = 0; i < bc.size(); ++i) bc[i] = (unsigned char) (rank == root ? i * 7 : 0);
    if (cb->broadcast(cb->user, bc.data(), bc.size(), root)) return 8;
    for (size_t i = 0; i < bc.size(); ++i)
        if (bc[i] != (unsigned char) (i * 7)) return 9;
    return 0;
}

void o2v_hip_comm_destroy(o2v_hip_comm *comm) { delete comm; }
const char *o2v_hip_comm_kind(const o2v_hip_comm *comm) { return comm ? comm->kind() : "none"; }
const char *o2v_hip_comm_last_error(const o2v_hip_comm *comm) { return comm ? comm->err.c_str() : "null communicator"; }

}  // extern "C"
