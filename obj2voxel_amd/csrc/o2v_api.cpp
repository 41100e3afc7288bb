// o2v_api.cpp -- host side of the drop-in C API (include/obj2voxel.h).
//
// Mirrors the reference's instance / setter / voxelize layer (src/obj2voxel.cpp:142-173, :578-637, :645-1003)
// and replaces its chunk loop with one call sequence into the HIP pipeline through the C-ABI of o2v_hip.h.
// There is no CPU voxelization path in this library: without a usable GPU obj2voxel_voxelize() logs an error
// and returns OBJ2VOXEL_ERR_DEVICE.
#include "o2v_io.hpp"

#include "../../include/o2v_hip.h"
#include "../../include/obj2voxel.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <string>
#include <vector>

// (OBJ2VOXEL_ERR_DEVICE - no device, HIP failure, device out of memory - is declared in include/obj2voxel.h)

using namespace o2v;

// ---- opaque API types ------------------------------------------------------------------------------------

// reference src/triangle.hpp:148-167 (wrapper around voxelio::Image)
struct obj2voxel_texture {
    std::vector<uint8_t> pixels;
    size_t width = 0, height = 0, channels = 0;
    uint32_t wrap = 1;  // REPEAT is the default (include/obj2voxel.h:346-347)
    bool loaded() const { return !pixels.empty(); }
};

// reference src/triangle.hpp:170-195; filled only through obj2voxel_set_triangle_*
// "implementation-defined" in the public header (include/obj2voxel.h): the cached-triangle record itself, so that a
// triangle callback fills the staging object the cache reads from (reference triangle.hpp:170-195)
struct obj2voxel_triangle : o2v::HostTriangle {
    // Where a setter puts the vertices: the object's own v, or - while a callback source is drained on the one-GPU path -
    // the triangle's place in the page-locked block the device copies from (one 36-byte copy per triangle instead of two).
    float *v_out = v;
    bool v_set = false;
};

namespace {

// ---- logging (reference obj2voxel.cpp:639-682; process-global like the reference) ------------------------

obj2voxel_enum_t g_log_level = OBJ2VOXEL_LOG_LEVEL_INFO;  // constants.hpp:21 (release default)
obj2voxel_log_callback *g_log_callback = nullptr;
void *g_log_callback_data = nullptr;
std::mutex g_log_mutex;

}  // namespace

namespace o2v {

void log_message(int level, const std::string &msg)
{
    if (level > g_log_level) return;
    obj2voxel_log_callback *callback;
    void *callback_data;
    {
        // the callback runs outside the lock: it may itself call obj2voxel_set_log_callback
        std::lock_guard<std::mutex> lock{g_log_mutex};
        callback = g_log_callback;
        callback_data = g_log_callback_data;
    }
    if (callback && callback(callback_data, msg.c_str(), (obj2voxel_enum_t) level)) return;
    static std::mutex print_mutex;
    std::lock_guard<std::mutex> lock{print_mutex};
    static const char *names[] = {"", "ERROR", "WARNING", "INFO", "DEBUG"};
    std::fprintf(level <= OBJ2VOXEL_LOG_LEVEL_WARNING ? stderr : stdout, "[obj2voxel] [%s] %s\n",
                 names[level < 0 || level > 4 ? 0 : level], msg.c_str());
}

}  // namespace o2v

namespace {

#define O2V_ASSERT(cond, msg)                                                              \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            std::fprintf(stderr, "[obj2voxel] assertion failed: %s (%s)\n", #cond, msg);   \
            std::abort();                                                                  \
        }                                                                                  \
    } while (0)

enum class IoKind { MISSING, FILE, MEMORY, CALLBACK };

}  // namespace

// reference src/obj2voxel.cpp:142-173
struct obj2voxel_instance {
    IoKind input_kind = IoKind::MISSING;
    const char *input_path = nullptr;  // borrowed, read at voxelize time like the reference (:714-720)
    FileFormat input_format = FileFormat::UNKNOWN;
    obj2voxel_triangle_callback *input_callback = nullptr;
    void *input_callback_data = nullptr;

    IoKind output_kind = IoKind::MISSING;
    const char *output_path = nullptr;
    FileFormat output_format = FileFormat::UNKNOWN;
    obj2voxel_voxel_callback *output_callback = nullptr;
    void *output_callback_data = nullptr;

    obj2voxel_texture *default_texture = nullptr;
    float mesh_bounds[6] = {0, 0, 0, 0, 0, 0};
    bool bounds_known = false;
    obj2voxel_enum_t strategy = OBJ2VOXEL_MAX_STRATEGY;
    uint32_t output_resolution = 0;
    uint32_t supersampling = 1;
    bool parallel = false;
    int unit_transform[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

    std::unique_ptr<VoxelSink> sink;

    // worker compatibility (reference :957-1003): workers only register and wait
    std::mutex worker_mutex;
    std::condition_variable worker_cv;
    uint32_t worker_count = 0;
    uint32_t exit_tokens = 0;
    bool workers_stopped = false;
    bool done = false;
};

namespace {

// wall-clock phases of one obj2voxel_voxelize call, logged at DEBUG level
struct PhaseClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double lap_ms()
    {
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
        return ms;
    }
};

struct CallbackTriangleSource final : TriangleSource {
    obj2voxel_triangle_callback *callback;
    void *data;
    CallbackTriangleSource(obj2voxel_triangle_callback *cb, void *d) : callback{cb}, data{d} {}
    const HostTriangle *next() override
    {
        // like `CachedTriangle triangle{}` (obj2voxel.cpp:585) the staging object is zeroed once and reused,
        // so fields a setter does not write keep their previous value
        return callback(data, &staging) ? &staging : nullptr;
    }
    obj2voxel_triangle staging{};
    bool is_callback() const override { return true; }
};

struct CallbackVoxelSink final : VoxelSink {
    obj2voxel_voxel_callback *callback;
    void *data;
    bool good = true;
    CallbackVoxelSink(obj2voxel_voxel_callback *cb, void *d) : callback{cb}, data{d} {}
    bool can_write() const override { return good; }
    void write(uint32_t *voxels, size_t count) override
    {
        written += count;
        good &= callback(data, voxels, count);  // reference io.cpp:638-653
    }
    void finalize() override {}
};

std::unique_ptr<TriangleSource> open_input(obj2voxel_instance &inst)
{
    switch (inst.input_kind) {
    case IoKind::CALLBACK:
        return std::unique_ptr<TriangleSource>{new CallbackTriangleSource{inst.input_callback, inst.input_callback_data}};
    case IoKind::FILE:
        switch (inst.input_format) {
        case FileFormat::OBJ: return open_obj_file(inst.input_path, inst.default_texture);
        case FileFormat::STL: return open_stl_file(inst.input_path);
        default: return nullptr;
        }
    default: return nullptr;
    }
}

std::unique_ptr<VoxelSink> open_output(obj2voxel_instance &inst)
{
    switch (inst.output_kind) {
    case IoKind::CALLBACK:
        return std::unique_ptr<VoxelSink>{new CallbackVoxelSink{inst.output_callback, inst.output_callback_data}};
    case IoKind::FILE: return open_file_sink(inst.output_path, inst.output_format, inst.output_resolution);
    case IoKind::MEMORY: return open_memory_sink(inst.output_format, inst.output_resolution);
    default: return nullptr;
    }
}

// ---- triangle cache ------------------------------------------------------------------------------------------
// The reference caches every triangle as a 104-byte CachedTriangle (obj2voxel.cpp:122-132,585-600); here the source is
// drained straight into the flat arrays of the device C-ABI.  Arrays only some triangle types need are created when
// the first such triangle arrives (earlier entries zero), so a material-less mesh uploads 36 bytes per triangle.
struct MeshArrays {
    std::vector<float> verts, uvs, colors;
    std::vector<uint32_t> types;
    std::vector<int32_t> texids;
    std::map<const obj2voxel_texture *, int32_t> tex_index;
    std::vector<const obj2voxel_texture *> tex_list;
    uint64_t n = 0;

    void push(const HostTriangle &t)
    {
        verts.insert(verts.end(), t.v, t.v + 9);
        if (t.type != O2V_HIP_TRI_MATERIALLESS && types.empty()) types.assign(n, O2V_HIP_TRI_MATERIALLESS);
        if (!types.empty() || t.type != O2V_HIP_TRI_MATERIALLESS) types.push_back(t.type);
        if (t.type == O2V_HIP_TRI_UNTEXTURED && colors.empty()) colors.assign(n * 3, 0.f);
        if (!colors.empty() || t.type == O2V_HIP_TRI_UNTEXTURED) colors.insert(colors.end(), t.color, t.color + 3);
        if (t.type == O2V_HIP_TRI_TEXTURED) {
            O2V_ASSERT(t.texture != nullptr && t.texture->loaded(), "textured triangle without a loaded texture");
            if (uvs.empty()) {
                uvs.assign(n * 6, 0.f);
                texids.assign(n, 0);
            }
        }
        if (!uvs.empty() || t.type == O2V_HIP_TRI_TEXTURED) {
            int32_t id = 0;
            if (t.type == O2V_HIP_TRI_TEXTURED) {
                auto it = tex_index.find(t.texture);
                if (it == tex_index.end()) {
                    it = tex_index.emplace(t.texture, (int32_t) tex_list.size()).first;
                    tex_list.push_back(t.texture);
                }
                id = it->second;
            }
            uvs.insert(uvs.end(), t.t, t.t + 6);
            texids.push_back(id);
        }
        ++n;
    }
};

// The same for a single-GPU session, without the intermediate host copy: the source is drained straight into the
// page-locked staging blocks of the device context (o2v_hip_begin / commit / end_triangles), each block being copied to
// the device while the next one is filled.
struct StreamedUpload {
    o2v_hip_ctx *ctx = nullptr;
    o2v_hip_staging st{};
    uint64_t in_block = 0, total = 0;
    uint32_t arrays = 0;  // O2V_HIP_ARRAY_* the mesh has so far
    bool any_textured = false, failed = false;
    std::map<const obj2voxel_texture *, int32_t> tex_index;
    std::vector<const obj2voxel_texture *> tex_list;

    bool begin(o2v_hip_ctx *c)
    {
        ctx = c;
        return o2v_hip_begin_triangles(ctx, &st) == O2V_HIP_OK;
    }
    // an optional array appears: the triangles of this block so far get its default (earlier blocks: on the device)
    void activate(uint32_t which)
    {
        if (o2v_hip_stage_arrays(ctx, which, &st) != O2V_HIP_OK) {
            failed = true;
            return;
        }
        if (which & O2V_HIP_ARRAY_TYPES) std::fill(st.types, st.types + in_block, (uint32_t) O2V_HIP_TRI_MATERIALLESS);
        if (which & O2V_HIP_ARRAY_COLORS) std::fill(st.colors, st.colors + in_block * 3, 0.f);
        if (which & O2V_HIP_ARRAY_UVS) std::fill(st.uvs, st.uvs + in_block * 6, 0.f);
        if (which & O2V_HIP_ARRAY_TEXIDS) std::fill(st.texids, st.texids + in_block, 0);
        arrays |= which;
    }
    void commit()
    {
        if (o2v_hip_commit_triangles(ctx, in_block, arrays, &st) != O2V_HIP_OK) failed = true;
        total += in_block;
        in_block = 0;
    }
    float *vertex_slot() { return st.verts + in_block * 9; }
    void push(const HostTriangle &t)
    {
        if (failed) return;
        std::memcpy(vertex_slot(), t.v, sizeof(t.v));
        push_rest(t);
    }
    /// the triangle's vertices are in vertex_slot() already
    void push_rest(const HostTriangle &t)
    {
        if (t.type != O2V_HIP_TRI_MATERIALLESS && !(arrays & O2V_HIP_ARRAY_TYPES)) activate(O2V_HIP_ARRAY_TYPES);
        if (t.type == O2V_HIP_TRI_UNTEXTURED && !(arrays & O2V_HIP_ARRAY_COLORS)) activate(O2V_HIP_ARRAY_COLORS);
        if (t.type == O2V_HIP_TRI_TEXTURED) {
            O2V_ASSERT(t.texture != nullptr && t.texture->loaded(), "textured triangle without a loaded texture");
            if (!(arrays & O2V_HIP_ARRAY_UVS)) activate(O2V_HIP_ARRAY_UVS | O2V_HIP_ARRAY_TEXIDS);
            any_textured = true;
        }
        if (failed) return;  // an array could not be page-locked
        if (arrays & O2V_HIP_ARRAY_TYPES) st.types[in_block] = t.type;
        if (arrays & O2V_HIP_ARRAY_COLORS) std::memcpy(st.colors + in_block * 3, t.color, sizeof(t.color));
        if (arrays & O2V_HIP_ARRAY_UVS) {
            int32_t id = 0;
            if (t.type == O2V_HIP_TRI_TEXTURED) {
                auto it = tex_index.find(t.texture);
                if (it == tex_index.end()) {
                    it = tex_index.emplace(t.texture, (int32_t) tex_list.size()).first;
                    tex_list.push_back(t.texture);
                }
                id = it->second;
            }
            std::memcpy(st.uvs + in_block * 6, t.t, sizeof(t.t));
            st.texids[in_block] = id;
        }
        if (++in_block == st.capacity) commit();
    }
    /// Pulls the rest of a callback source, the setters writing each triangle's vertices where the device copies them from
    /// (t holds the triangle pushed last).  A callback that returns true without calling a setter repeats the previous
    /// vertices, as the reference's reused CachedTriangle does (obj2voxel.cpp:585-588).  The loop for triangles that are
    /// nothing but vertices keeps its cursor in registers: at 870 k triangles the per-triangle cost is the drop-in's wall time.
    void drain(obj2voxel_triangle_callback *callback, void *data, obj2voxel_triangle &t)
    {
        float last[9];
        std::memcpy(last, t.v, sizeof(last));
        const float *previous = last;
        float *slot = vertex_slot();
        const float *end = st.verts + st.capacity * 9;
        while (!failed) {
            t.v_out = slot;
            t.v_set = false;
            if (!callback(data, &t)) break;
            if (!t.v_set) std::memcpy(slot, previous, sizeof(last));
            previous = slot;
            if (t.type == O2V_HIP_TRI_MATERIALLESS && !arrays) {
                slot += 9;
                if (slot != end) continue;
                in_block = st.capacity;
                commit();
            }
            else {
                in_block = uint64_t(slot - st.verts) / 9;
                push_rest(t);
            }
            if (in_block == 0) {  // the block went to the device and its memory will be written again
                std::memcpy(last, previous, sizeof(last));
                previous = last;
            }
            slot = vertex_slot();
            end = st.verts + st.capacity * 9;
        }
        if (!failed) in_block = uint64_t(slot - st.verts) / 9;
        t.v_out = t.v;
    }
    bool finish()
    {
        if (in_block && !failed) commit();
        return !failed && o2v_hip_end_triangles(ctx, any_textured ? 1u : 0u) == O2V_HIP_OK;
    }
};

constexpr uint64_t kReadBackBatch = 1u << 20;  // (x, y, z, argb) records per sink call

// ---- device sessions -----------------------------------------------------------------------------------------------
// What one obj2voxel_voxelize call drives: one GPU (a context), or - if the environment names several devices
// (O2V_DEVICES=0,1,2,3 or O2V_DEVICES=all) - an in-process group of GPUs with the grid sharded by z-slab
// (include/o2v_hip.h, multi-GPU section), where the reference hands its chunks to a worker pool
// (src/obj2voxel.cpp:467-520).
struct Session {
    std::vector<int> devices;
    o2v_hip_ctx *ctx = nullptr;      // one device
    o2v_hip_group *group = nullptr;  // several
    uint32_t *pinned[2] = {nullptr, nullptr};  // read-back staging (pinned: D2H runs at link rate and asynchronously)
    uint64_t pinned_records = 0;
    // Page-locking the two read-back buffers takes ~2 ms: a new session does it on a thread of its own while the triangle
    // source is drained (wait_for_pinned() before the first use).
    std::thread pinning;
    void pin_now()
    {
        // (on the session's first device: a new thread's current device is device 0, which this process may not be meant to touch)
        for (uint32_t *&p : pinned)
            if (!p) p = static_cast<uint32_t *>(o2v_hip_alloc_pinned_on(devices[0], kReadBackBatch * 16));
        pinned_records = pinned[0] && pinned[1] ? kReadBackBatch : 0;
    }
    void start_pinning()
    {
        try {
            pinning = std::thread{[this] { pin_now(); }};
        }
        catch (const std::system_error &) {
            // (no thread to be had: the buffers are made when they are first needed, wait_for_pinned)
        }
    }
    // true if both read-back buffers exist; a session whose buffers could not be made earlier (by the thread, or because the
    // host was short of lockable memory at that time) tries again, here, every time they are needed
    bool wait_for_pinned()
    {
        if (pinning.joinable()) pinning.join();
        if (!pinned_records) pin_now();
        return pinned_records != 0;
    }

    uint32_t ranks() const { return group ? o2v_hip_group_size(group) : 1u; }
    o2v_hip_ctx *rank_ctx(uint32_t r) { return group ? o2v_hip_group_ctx(group, r) : ctx; }
    const char *last_error() const { return group ? o2v_hip_group_last_error(group) : o2v_hip_last_error(ctx); }
    ~Session()
    {
        if (pinning.joinable()) pinning.join();
        for (uint32_t *p : pinned)
            if (p) o2v_hip_free_pinned(p);
        if (group) o2v_hip_group_destroy(group);
        if (ctx) o2v_hip_destroy(ctx);
    }
};

std::vector<int> requested_devices()
{
    std::vector<int> devices;
    if (const char *env = std::getenv("O2V_DEVICES")) {
        if (std::strcmp(env, "all") == 0) {
            for (int d = 0; d < o2v_hip_device_count(); ++d) devices.push_back(d);
        }
        else {
            for (const char *q = env; *q;) {
                char *end = nullptr;
                const long d = std::strtol(q, &end, 10);
                if (end == q) break;
                devices.push_back((int) d);
                q = *end == ',' ? end + 1 : end;
            }
        }
    }
    if (devices.empty()) {
        int device = 0;
        if (const char *env = std::getenv("O2V_DEVICE")) device = std::atoi(env);
        devices.push_back(device);
    }
    return devices;
}

// Instances are single-use (reference obj2voxel.cpp:604-606,635) but device sessions are not: one per process is kept and
// reused, so repeated voxelizations do not pay for creating contexts and allocating the dense grids again.  It holds on
// to device memory (4 + 8 bytes per cell of the largest grid so far); o2v_release_cached_device_memory() (o2v_hip.h) or
// O2V_CONTEXT_CACHE=0 give it back.
std::mutex g_session_mutex;
Session *g_cached_session = nullptr;  // intentionally never destroyed at exit (HIP may already be torn down)
bool g_cached_busy = false;

Session *acquire_session(const std::vector<int> &devices, bool &from_cache, std::string &why)
{
    {
        std::lock_guard<std::mutex> lock{g_session_mutex};
        if (g_cached_session && !g_cached_busy && g_cached_session->devices == devices) {
            g_cached_busy = true;
            from_cache = true;
            return g_cached_session;
        }
    }
    from_cache = false;
    Session *s = new Session;
    s->devices = devices;
    const int rc = devices.size() > 1 ? o2v_hip_group_create(devices.data(), (uint32_t) devices.size(), &s->group)
                                      : o2v_hip_create(devices[0], &s->ctx);
    if (rc != O2V_HIP_OK) {
        // the reason, before the session that knows it is gone
        std::string list;
        for (int d : devices) list += (list.empty() ? "" : ",") + std::to_string(d);
        const int n_dev = o2v_hip_device_count();
        why = "device(s) " + list + " of " + std::to_string(n_dev) + " visible: code " + std::to_string(rc);
        for (int d : devices)
            if (d < 0 || d >= n_dev) why += "; device index " + std::to_string(d) + " does not exist (O2V_DEVICE / O2V_DEVICES)";
        if (s->group) why += std::string("; ") + o2v_hip_group_last_error(s->group);
        if (rc == O2V_HIP_ERR_OUT_OF_MEMORY) why += "; out of device memory";
        delete s;
        return nullptr;
    }
    s->start_pinning();
    return s;
}

void release_session(Session *s, bool from_cache)
{
    if (!s) return;
    const char *cache = std::getenv("O2V_CONTEXT_CACHE");
    bool keep = !(cache && cache[0] == '0');
    // a session whose grids grew beyond 32 GiB (a voxelization that ran in memory-sized z-slabs holds most of the device)
    // is not worth keeping: the next, smaller job would find the device full
    for (uint32_t r = 0; keep && r < s->ranks(); ++r) {
        o2v_hip_stats st{};
        o2v_hip_get_stats(s->rank_ctx(r), &st);
        if (st.grid_bytes > (32ull << 30)) keep = false;
    }
    {
        std::lock_guard<std::mutex> lock{g_session_mutex};
        if (from_cache) {
            g_cached_busy = false;
            if (keep) return;
            g_cached_session = nullptr;
        }
        else if (keep && !g_cached_session) {  // the first finished voxelization donates its session to the cache
            g_cached_session = s;
            g_cached_busy = false;
            return;
        }
    }
    delete s;  // caching is off, or a concurrent voxelization on another thread used a temporary session
}

// A session on its way: acquire_session() on the calling thread or on one of its own (see voxelize()).  Whoever takes the
// session releases it; one that nobody took (the call failed before it was needed) is released here.
struct PendingSession {
    std::vector<int> devices;
    Session *session = nullptr;
    bool from_cache = false, started = false, taken = false, background = false;
    std::string why;
    double create_ms = 0.0;
    std::thread worker;
    void run()
    {
        const auto t0 = std::chrono::steady_clock::now();
        session = acquire_session(devices, from_cache, why);
        create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    void start(bool on_a_thread)
    {
        started = true;
        devices = requested_devices();
        if (on_a_thread) {
            try {
                worker = std::thread{[this] { run(); }};
                background = true;
                return;
            }
            catch (const std::system_error &) {
                // (no thread to be had: on this one)
            }
        }
        run();
    }
    Session *take()
    {
        if (worker.joinable()) worker.join();
        taken = true;
        return session;
    }
    ~PendingSession()
    {
        if (worker.joinable()) worker.join();
        if (session && !taken) release_session(session, from_cache);
    }
};

// The GPU leg of voxelize_specialized (reference obj2voxel.cpp:467-520): bounds, transform, per-triangle
// voxelization, colour combine and packing all happen on the device(s); the host only moves data.
obj2voxel_error_t voxelize_on_device(obj2voxel_instance &inst, Session *session, const std::vector<int> &devices, MeshArrays *mesh_ptr,
                                     const std::vector<const obj2voxel_texture *> &tex_list, uint64_t T, PhaseClock &clock)
{
    auto device_error = [&](const char *what) {
        log_message(OBJ2VOXEL_LOG_LEVEL_ERROR, std::string(what) + ": " + session->last_error());
        return OBJ2VOXEL_ERR_DEVICE;
    };

    std::vector<o2v_hip_texture> tex_desc;
    for (const obj2voxel_texture *t : tex_list)
        tex_desc.push_back(o2v_hip_texture{t->pixels.data(), (uint32_t) t->width, (uint32_t) t->height,
                                           (uint32_t) t->channels, t->wrap});
    if (session->group) {
        MeshArrays &mesh = *mesh_ptr;
        const float *uvs = mesh.uvs.empty() ? nullptr : mesh.uvs.data();
        const uint32_t *types = mesh.types.empty() ? nullptr : mesh.types.data();
        const float *colors = mesh.colors.empty() ? nullptr : mesh.colors.data();
        const int32_t *texids = mesh.texids.empty() ? nullptr : mesh.texids.data();
        int upload = O2V_HIP_UPLOAD_H2D;
        if (const char *env = std::getenv("O2V_UPLOAD"))
            upload = std::strcmp(env, "broadcast") == 0 ? O2V_HIP_UPLOAD_BROADCAST : std::strcmp(env, "peer") == 0 ? O2V_HIP_UPLOAD_PEER : O2V_HIP_UPLOAD_H2D;
        if (!tex_desc.empty() && o2v_hip_group_set_textures(session->group, tex_desc.data(), (uint32_t) tex_desc.size()) != O2V_HIP_OK)
            return device_error("uploading textures failed");
        if (o2v_hip_group_set_triangles(session->group, mesh.verts.data(), uvs, types, colors, texids, T, upload) != O2V_HIP_OK)
            return device_error("uploading triangles failed");
        mesh = MeshArrays{};  // the devices hold the triangles now
    }
    else {
        // the triangles were streamed to the device while the source was drained
        if (!tex_desc.empty() && o2v_hip_set_textures(session->ctx, tex_desc.data(), (uint32_t) tex_desc.size()) != O2V_HIP_OK)
            return device_error("uploading textures failed");
    }
    const double ms_upload = clock.lap_ms();

    o2v_hip_params params{};
    params.resolution = inst.output_resolution;
    params.supersampling = inst.supersampling;
    params.strategy = inst.strategy;
    for (int i = 0; i < 9; ++i) params.unit_transform[i] = inst.unit_transform[i];
    params.bounds_known = inst.bounds_known ? 1u : 0u;
    for (int i = 0; i < 6; ++i) params.bounds[i] = inst.mesh_bounds[i];
    params.z_begin = params.z_end = 0;
    if (g_log_level >= OBJ2VOXEL_LOG_LEVEL_DEBUG) params.flags |= O2V_HIP_FLAG_STAGE_TIMES;  // (log_pipeline prints the device time)

    const uint32_t n_ranks = session->ranks();
    std::vector<uint64_t> counts(n_ranks, 0);

    // Hand the (x, y, z, argb) records to the sink in batches (reference obj2voxel.cpp:298-303; the callback may be
    // invoked any number of times, in any order), rank by rank.  Two pinned staging buffers: while the sink consumes one
    // batch the next one is already on its way from the device.
    constexpr uint64_t kBatch = kReadBackBatch;
    double ms_device = 0.0, ms_sink = 0.0;
    auto drain_to_sink = [&]() -> obj2voxel_error_t {
        struct Batch {
            uint32_t rank;
            uint64_t first, n;
        };
        std::vector<Batch> batches;
        for (uint32_t r = 0; r < n_ranks; ++r)
            for (uint64_t first = 0; first < counts[r]; first += kBatch) batches.push_back({r, first, std::min<uint64_t>(kBatch, counts[r] - first)});
        auto start_read = [&](size_t k) {
            const Batch &b = batches[k];
            return o2v_hip_read_voxels_async(session->rank_ctx(b.rank), session->pinned[k & 1], b.first, b.n) == O2V_HIP_OK;
        };
        if (!session->wait_for_pinned()) return device_error("allocating read-back staging failed");
        {
            uint64_t total = 0;
            for (uint64_t cnt : counts) total += cnt;
            inst.sink->expect(total);
        }
        if (!batches.empty() && !start_read(0)) return device_error("reading voxels failed");
        for (size_t k = 0; k < batches.size(); ++k) {
            if (!inst.sink->can_write()) break;
            if (o2v_hip_read_voxels_wait(session->rank_ctx(batches[k].rank)) != O2V_HIP_OK) return device_error("reading voxels failed");
            if (k + 1 < batches.size() && !start_read(k + 1)) return device_error("reading voxels failed");
            inst.sink->write(session->pinned[k & 1], batches[k].n);
        }
        if (!inst.sink->can_write()) {
            log_message(OBJ2VOXEL_LOG_LEVEL_ERROR, "Voxelization failed because of IO error");
            return OBJ2VOXEL_ERR_IO_ERROR_DURING_VOXEL_WRITE;
        }
        return OBJ2VOXEL_ERR_OK;
    };
    auto log_pipeline = [&]() {
        for (uint32_t r = 0; r < n_ranks; ++r) {
            o2v_hip_timings tm{};
            o2v_hip_get_timings(session->rank_ctx(r), &tm);
            log_message(OBJ2VOXEL_LOG_LEVEL_DEBUG, "device " + std::to_string(devices[r]) + " pipeline: " + std::to_string(tm.total_ms) + " ms, " +
                                                       std::to_string(counts[r]) + " voxels, " + std::to_string(tm.passes) + " pass(es)" +
                                                       (n_ranks > 1 ? ", plan " + std::to_string(tm.plan_ms) + " ms (collectives " +
                                                                          std::to_string(tm.collective_ms) + " ms)"
                                                                    : std::string()));
        }
    };

    if (session->group) {
        if (o2v_hip_group_voxelize(session->group, &params, counts.data(), nullptr) != O2V_HIP_OK)
            return device_error("device voxelization failed");
        ms_device = clock.lap_ms();
        log_pipeline();
        const obj2voxel_error_t rc_sink = drain_to_sink();
        if (rc_sink != OBJ2VOXEL_ERR_OK) return rc_sink;
        ms_sink = clock.lap_ms();
    }
    else {
        // One GPU.  The dense grids of the whole resolution may not fit the device (4 + 8 bytes per cell: 8192^3, the reference
        // README's showcase resolution, would take 6.6 TB), where the reference's sparse VoxelMap just grows (util.hpp:179-208):
        // the grid is then voxelized as consecutive z-slabs as thick as the free memory allows - the reference's chunk
        // mechanism again (obj2voxel.cpp:226-243, voxelization.cpp:440-444) - each slab's records going to the sink before the
        // next slab starts.  The mesh bounds and the z extents of the triangle blocks (which let a slab skip the blocks it
        // cannot meet) are computed once, by the slab plan.
        // A pass handles a box of at most 65 535 samples per axis (voxel coordinates travel in 16-bit fields relative to it,
        // include/o2v_hip.h), where the reference carries u32 coordinates (src/util.hpp:185-196): a finer grid is cut into x / y
        // tiles the same way - every output voxel belongs to exactly one tile.
        const uint32_t ss = params.supersampling ? params.supersampling : 1u;
        const uint32_t tile = (65535u / ss) & ~3u;
        const bool tiled = params.resolution > tile;
        if (tiled)
            log_message(OBJ2VOXEL_LOG_LEVEL_DEBUG, "resolution " + std::to_string(params.resolution) + ": x / y tiles of " + std::to_string(tile) + " voxels");
        for (uint32_t y0 = 0; y0 < params.resolution; y0 += tile)
            for (uint32_t x0 = 0; x0 < params.resolution; x0 += tile) {
                if (tiled) {
                    params.x_begin = x0;
                    params.x_end = std::min<uint64_t>(params.resolution, (uint64_t) x0 + tile);
                    params.y_begin = y0;
                    params.y_end = std::min<uint64_t>(params.resolution, (uint64_t) y0 + tile);
                }
                uint32_t layers = 0;
                if (o2v_hip_max_slab_layers(session->ctx, &params, &layers) != O2V_HIP_OK) return device_error("querying device memory failed");
                if (const char *force = std::getenv("O2V_TEST_SLAB_LAYERS")) layers = std::min<uint32_t>(layers, (uint32_t) std::atoi(force));  // test hook
                if (layers == 0) {
                    log_message(OBJ2VOXEL_LOG_LEVEL_ERROR, "resolution " + std::to_string(params.resolution) + ": not even one 4-layer slab of the dense grid fits the device memory");
                    return OBJ2VOXEL_ERR_DEVICE;
                }
                if (layers < params.resolution && !tiled) {
                    const uint32_t n_slabs = (params.resolution + layers - 1) / layers;
                    log_message(OBJ2VOXEL_LOG_LEVEL_DEBUG, "the dense grid does not fit the device: " + std::to_string(n_slabs) + " z-slabs of " +
                                                               std::to_string(layers) + " layers");
                    uint32_t cuts[2];
                    float bounds[6];
                    if (o2v_hip_plan_slabs(session->ctx, &params, 1, cuts, bounds) != O2V_HIP_OK) return device_error("device slab plan failed");
                    params.bounds_known = 1;
                    for (int i = 0; i < 6; ++i) params.bounds[i] = bounds[i];
                }
                for (uint32_t z0 = 0; z0 < params.resolution; z0 += layers) {
                    params.z_begin = layers < params.resolution ? z0 : 0;
                    params.z_end = layers < params.resolution ? std::min<uint32_t>(params.resolution, z0 + layers) : 0;
                    if (o2v_hip_voxelize(session->ctx, &params, &counts[0]) != O2V_HIP_OK) return device_error("device voxelization failed");
                    ms_device += clock.lap_ms();
                    log_pipeline();
                    const obj2voxel_error_t rc_sink = drain_to_sink();
                    if (rc_sink != OBJ2VOXEL_ERR_OK) return rc_sink;
                    ms_sink += clock.lap_ms();
                }
            }
    }
    log_message(OBJ2VOXEL_LOG_LEVEL_DEBUG, "host phases: session + upload " + std::to_string(ms_upload) + " ms, device call(s) " +
                                               std::to_string(ms_device) + " ms, read back + sink " + std::to_string(ms_sink) + " ms");
    log_message(OBJ2VOXEL_LOG_LEVEL_INFO, "Voxelized " + std::to_string(T) + " triangles, writing any buffered voxels ...");
    inst.sink->finalize();
    if (!inst.sink->can_write()) return OBJ2VOXEL_ERR_IO_ERROR_DURING_VOXEL_WRITE;
    log_message(OBJ2VOXEL_LOG_LEVEL_INFO, "All " + std::to_string(inst.sink->written) + " voxels written");
    return OBJ2VOXEL_ERR_OK;
}

// reference obj2voxel.cpp:602-637 (precondition order: done, input, output, resolution) and :578-600
obj2voxel_error_t voxelize(obj2voxel_instance &inst)
{
    if (inst.done) return OBJ2VOXEL_ERR_DOUBLE_VOXELIZATION;
    if (inst.input_kind == IoKind::MISSING) {
        log_message(OBJ2VOXEL_LOG_LEVEL_ERROR, "No input was specified");
        return OBJ2VOXEL_ERR_NO_INPUT;
    }
    if (inst.output_kind == IoKind::MISSING) {
        log_message(OBJ2VOXEL_LOG_LEVEL_ERROR, "No output was specified");
        return OBJ2VOXEL_ERR_NO_OUTPUT;
    }
    if (inst.output_resolution == 0) {
        log_message(OBJ2VOXEL_LOG_LEVEL_ERROR, "No resolution was specified");
        return OBJ2VOXEL_ERR_NO_RESOLUTION;
    }
    // A file input is parsed as a whole when it is opened, and a new process' first device session costs as much again (HIP
    // runtime start, context, code object): neither needs the other, so the session is made on a thread meanwhile.  (A CLI run
    // is one process per model, src/main.cpp:147-200: it always pays for a new session.)  A callback source has nothing to
    // overlap with; its session is made when its first triangle is there, as before.
    PendingSession pending;
    if (inst.input_kind == IoKind::FILE) pending.start(true);
    std::unique_ptr<TriangleSource> input = open_input(inst);
    if (!input) return OBJ2VOXEL_ERR_IO_ERROR_ON_OPEN_INPUT_FILE;
    inst.sink = open_output(inst);
    if (!inst.sink) return OBJ2VOXEL_ERR_IO_ERROR_ON_OPEN_OUTPUT_FILE;

    log_message(OBJ2VOXEL_LOG_LEVEL_DEBUG, "Caching triangles ...");
    PhaseClock clock;
    const HostTriangle *tri = input->next();
    if (!tri) {
        // reference obj2voxel.cpp:590-594 (no device is needed for an empty model)
        log_message(OBJ2VOXEL_LOG_LEVEL_WARNING, "Model has no triangles, aborting and writing empty voxel model");
        inst.sink->finalize();
        const obj2voxel_error_t empty_result = inst.sink->can_write() ? OBJ2VOXEL_ERR_OK : OBJ2VOXEL_ERR_IO_ERROR_DURING_VOXEL_WRITE;
        if (inst.output_kind != IoKind::MEMORY) inst.sink.reset();
        inst.done = true;
        return empty_result;
    }
    // The device session comes before the rest of the source: with one GPU the triangles are drained straight into its
    // staging memory.
    if (!pending.started) pending.start(false);
    Session *session = pending.take();
    const std::vector<int> &devices = pending.devices;
    const bool from_cache = pending.from_cache;
    const std::string &why = pending.why;
    if (!session) {
        log_message(OBJ2VOXEL_LOG_LEVEL_ERROR, "No usable MI355X (gfx950) device (" + why + "): the GPU voxelization path cannot run "
                                               "and this library has no CPU fallback");
        if (inst.output_kind != IoKind::MEMORY) inst.sink.reset();
        inst.done = true;
        return OBJ2VOXEL_ERR_DEVICE;
    }
    struct Guard {
        Session *s;
        bool from_cache;
        ~Guard() { release_session(s, from_cache); }
    } guard{session, from_cache};
    if (!from_cache)
        log_message(OBJ2VOXEL_LOG_LEVEL_DEBUG, "host phases: creating the device session " + std::to_string(pending.create_ms) + " ms" +
                                                   (pending.background ? " on a thread beside the input's parsing, waited for " + std::to_string(clock.lap_ms()) + " ms"
                                                                       : std::string()));

    MeshArrays mesh;
    StreamedUpload stream;
    uint64_t n_tris = 0;
    bool upload_ok = true;
    if (session->group) {
        for (; tri; tri = input->next()) mesh.push(*tri);
        n_tris = mesh.n;
    }
    else {
        upload_ok = stream.begin(session->ctx);
        if (upload_ok) {
            if (input->is_callback()) {
                auto *source = static_cast<CallbackTriangleSource *>(input.get());
                stream.push(*tri);
                stream.drain(source->callback, source->data, source->staging);
            }
            else {
                for (; tri; tri = input->next()) stream.push(*tri);
            }
            upload_ok = stream.finish();
        }
        n_tris = stream.total;
    }
    log_message(OBJ2VOXEL_LOG_LEVEL_DEBUG, "host phases: draining the triangle source " + std::to_string(clock.lap_ms()) + " ms");

    obj2voxel_error_t result;
    if (!upload_ok) {
        log_message(OBJ2VOXEL_LOG_LEVEL_ERROR, std::string("uploading triangles failed: ") + session->last_error());
        result = OBJ2VOXEL_ERR_DEVICE;
    }
    else {
        log_message(OBJ2VOXEL_LOG_LEVEL_INFO, "Cached model with " + std::to_string(n_tris) + " triangles");
        result = voxelize_on_device(inst, session, devices, &mesh, session->group ? mesh.tex_list : stream.tex_list, n_tris, clock);
    }
    if (inst.output_kind != IoKind::MEMORY) inst.sink.reset();
    inst.done = true;
    return result;
}

}  // namespace

// ---- exported API ---------------------------------------------------------------------------------------

// o2v_mesh (include/o2v_hip.h): a triangle file drained into the flat arrays of the device C-ABI by the same readers and the
// same cache code obj2voxel_voxelize() uses.  The source stays alive: it owns the textures of an OBJ's materials.
struct o2v_mesh {
    std::unique_ptr<TriangleSource> source;
    MeshArrays arrays;
};

extern "C" {

int o2v_mesh_load_file(const char *path, const char *type, o2v_mesh **out)
{
    if (!path || !out) return O2V_HIP_ERR_BAD_ARGUMENT;
    *out = nullptr;
    const FileFormat f = detect_format(path, type);
    std::unique_ptr<TriangleSource> src;
    if (f == FileFormat::OBJ) src = open_obj_file(path, nullptr);
    else if (f == FileFormat::STL) src = open_stl_file(path);
    if (!src) return O2V_HIP_ERR_BAD_ARGUMENT;
    o2v_mesh *m = new o2v_mesh;
    m->source = std::move(src);
    while (const HostTriangle *t = m->source->next()) m->arrays.push(*t);
    *out = m;
    return O2V_HIP_OK;
}

void o2v_mesh_free(o2v_mesh *mesh) { delete mesh; }

uint64_t o2v_mesh_arrays(const o2v_mesh *mesh, const float **verts, const float **uvs, const uint32_t **types, const float **colors,
                         const int32_t **texids, uint32_t *n_textures)
{
    if (!mesh) return 0;
    const MeshArrays &a = mesh->arrays;
    if (verts) *verts = a.verts.empty() ? nullptr : a.verts.data();
    if (uvs) *uvs = a.uvs.empty() ? nullptr : a.uvs.data();
    if (types) *types = a.types.empty() ? nullptr : a.types.data();
    if (colors) *colors = a.colors.empty() ? nullptr : a.colors.data();
    if (texids) *texids = a.texids.empty() ? nullptr : a.texids.data();
    if (n_textures) *n_textures = (uint32_t) a.tex_list.size();
    return a.n;
}

int o2v_mesh_texture(const o2v_mesh *mesh, uint32_t index, o2v_hip_texture *out)
{
    if (!mesh || !out || index >= mesh->arrays.tex_list.size()) return O2V_HIP_ERR_BAD_ARGUMENT;
    const obj2voxel_texture *t = mesh->arrays.tex_list[index];
    *out = o2v_hip_texture{t->pixels.data(), (uint32_t) t->width, (uint32_t) t->height, (uint32_t) t->channels, t->wrap};
    return O2V_HIP_OK;
}

obj2voxel_instance *obj2voxel_alloc(void) { return new obj2voxel_instance; }

void obj2voxel_free(obj2voxel_instance *instance)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    delete instance;
}

void obj2voxel_set_log_level(obj2voxel_enum_t level)
{
    O2V_ASSERT(level <= OBJ2VOXEL_LOG_LEVEL_DEBUG, "invalid log level");
    g_log_level = level;
}

obj2voxel_enum_t obj2voxel_get_log_level(void) { return g_log_level; }

void obj2voxel_set_log_callback(obj2voxel_log_callback *callback, void *callback_data)
{
    std::lock_guard<std::mutex> lock{g_log_mutex};
    g_log_callback = callback;
    g_log_callback_data = callback_data;
}

void obj2voxel_set_resolution(obj2voxel_instance *instance, uint32_t resolution)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(resolution != 0, "resolution must not be zero");
    instance->output_resolution = resolution;
}

void obj2voxel_set_supersampling(obj2voxel_instance *instance, uint32_t level)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(level == 1 || level == 2, "supersampling level must be 1 or 2");
    instance->supersampling = level;
}

void obj2voxel_set_color_strategy(obj2voxel_instance *instance, obj2voxel_enum_t strategy)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(strategy < 2, "invalid colour strategy");
    instance->strategy = strategy;
}

void obj2voxel_set_texture(obj2voxel_instance *instance, obj2voxel_texture *texture)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(texture != nullptr, "null texture");
    instance->default_texture = texture;
}

void obj2voxel_set_input_file(obj2voxel_instance *instance, const char *file, const char *type)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(file != nullptr, "null file");
    FileFormat f = detect_format(file, type);
    O2V_ASSERT(f != FileFormat::UNKNOWN, "unrecognised input file type");
    instance->input_kind = IoKind::FILE;
    instance->input_path = file;
    instance->input_format = f;
}

void obj2voxel_set_input_callback(obj2voxel_instance *instance, obj2voxel_triangle_callback *callback, void *callback_data)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(callback != nullptr, "null callback");
    instance->input_kind = IoKind::CALLBACK;
    instance->input_callback = callback;
    instance->input_callback_data = callback_data;
}

void obj2voxel_set_output_file(obj2voxel_instance *instance, const char *file, const char *type)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(file != nullptr, "null file");
    FileFormat f = detect_format(file, type);
    O2V_ASSERT(f != FileFormat::UNKNOWN, "unrecognised output file type");
    instance->output_kind = IoKind::FILE;
    instance->output_path = file;
    instance->output_format = f;
}

void obj2voxel_set_output_memory(obj2voxel_instance *instance, const char *type)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(type != nullptr, "null type");
    FileFormat f = detect_format(nullptr, type);
    O2V_ASSERT(f != FileFormat::UNKNOWN, "unrecognised output type");
    instance->output_kind = IoKind::MEMORY;
    instance->output_path = nullptr;
    instance->output_format = f;
}

void obj2voxel_set_output_callback(obj2voxel_instance *instance, obj2voxel_voxel_callback *callback, void *callback_data)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(callback != nullptr, "null callback");
    instance->output_kind = IoKind::CALLBACK;
    instance->output_callback = callback;
    instance->output_callback_data = callback_data;
}

void obj2voxel_set_parallel(obj2voxel_instance *instance, bool enabled)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    instance->parallel = enabled;
}

void obj2voxel_set_unit_transform(obj2voxel_instance *instance, const int transform[9])
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(transform != nullptr, "null transform");
    std::memcpy(instance->unit_transform, transform, sizeof(instance->unit_transform));
}

void obj2voxel_set_mesh_boundaries(obj2voxel_instance *instance, const float bounds[6])
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(bounds != nullptr, "null bounds");
    for (int i = 0; i < 6; ++i) O2V_ASSERT(bounds[i] - bounds[i] == 0.f, "mesh boundaries must be finite");
    for (int i = 0; i < 3; ++i) O2V_ASSERT(bounds[i] <= bounds[i + 3], "lower mesh bound must be <= upper bound");
    std::memcpy(instance->mesh_bounds, bounds, sizeof(instance->mesh_bounds));
    instance->bounds_known = true;
}

uint32_t obj2voxel_get_resolution(obj2voxel_instance *instance)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    return instance->output_resolution;
}

uint32_t obj2voxel_get_chunk_size(obj2voxel_instance *instance)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    return 64;  // reference constants.hpp:10
}

const obj2voxel_byte_t *obj2voxel_get_output_memory(obj2voxel_instance *instance, size_t *out_size)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    O2V_ASSERT(instance->sink != nullptr || instance->output_kind != IoKind::MEMORY,
               "accessing output memory before voxelization");
    if (instance->output_kind != IoKind::MEMORY) return nullptr;
    const ByteBuffer *bytes = instance->sink->memory();
    O2V_ASSERT(bytes != nullptr, "memory sink without buffer");
    *out_size = bytes->size;
    return bytes->bytes;
}

void obj2voxel_set_triangle_basic(obj2voxel_triangle *triangle, const float vertices[9])
{
    triangle->type = O2V_HIP_TRI_MATERIALLESS;
    std::memcpy(triangle->v_out, vertices, sizeof(triangle->v));
    triangle->v_set = true;
}

void obj2voxel_set_triangle_colored(obj2voxel_triangle *triangle, const float vertices[9], const float color[3])
{
    // the reference stores the colour but marks the triangle MATERIALLESS (obj2voxel.cpp:828-837): white
    triangle->type = O2V_HIP_TRI_MATERIALLESS;
    std::memcpy(triangle->v_out, vertices, sizeof(triangle->v));
    triangle->v_set = true;
    std::memcpy(triangle->color, color, sizeof(triangle->color));
}

void obj2voxel_set_triangle_textured(obj2voxel_triangle *triangle, const float vertices[9], const float textures[6],
                                     obj2voxel_texture *texture)
{
    triangle->type = O2V_HIP_TRI_TEXTURED;
    std::memcpy(triangle->v_out, vertices, sizeof(triangle->v));
    triangle->v_set = true;
    std::memcpy(triangle->t, textures, sizeof(triangle->t));
    triangle->texture = texture;
}

obj2voxel_texture *obj2voxel_texture_alloc(void) { return new obj2voxel_texture; }

void obj2voxel_texture_free(obj2voxel_texture *texture)
{
    O2V_ASSERT(texture != nullptr, "null texture");
    delete texture;
}

bool obj2voxel_texture_load_from_file(obj2voxel_texture *texture, const char *file, const char *type)
{
    O2V_ASSERT(texture != nullptr, "null texture");
    O2V_ASSERT(file != nullptr, "null file");
    if (detect_format(file, type) != FileFormat::PNG) return false;
    std::vector<uint8_t> bytes;
    if (!read_whole_file(file, bytes)) return false;
    return obj2voxel_texture_load_from_memory(texture, bytes.data(), bytes.size(), "png");
}

bool obj2voxel_texture_load_from_memory(obj2voxel_texture *texture, const obj2voxel_byte_t *data, size_t size,
                                        const char *type)
{
    O2V_ASSERT(texture != nullptr, "null texture");
    O2V_ASSERT(data != nullptr, "null data");
    if (detect_format(nullptr, type) != FileFormat::PNG) return false;
    std::vector<uint8_t> argb;
    size_t w = 0, h = 0;
    std::string err;
    if (!decode_png_argb(data, size, argb, w, h, err)) {
        log_message(OBJ2VOXEL_LOG_LEVEL_WARNING, "PNG decode failed: " + err);
        return false;
    }
    texture->pixels = std::move(argb);
    texture->width = w;
    texture->height = h;
    texture->channels = 4;
    return true;
}

bool obj2voxel_texture_load_pixels(obj2voxel_texture *texture, const obj2voxel_byte_t *pixels, size_t width,
                                   size_t height, size_t channels)
{
    O2V_ASSERT(texture != nullptr, "null texture");
    O2V_ASSERT(pixels != nullptr, "null pixels");
    O2V_ASSERT(channels == 3 || channels == 4, "channels must be 3 (RGB) or 4 (ARGB)");
    texture->pixels.assign(pixels, pixels + width * height * channels);
    texture->width = width;
    texture->height = height;
    texture->channels = channels;
    return true;
}

void obj2voxel_teture_set_uv_mode(obj2voxel_texture *texture, obj2voxel_enum_t mode)
{
    O2V_ASSERT(texture != nullptr && texture->loaded(), "can't set UV mode of empty texture");
    texture->wrap = mode == OBJ2VOXEL_UV_CLAMP ? 0u : 1u;
}

void obj2voxel_texture_get_meta(obj2voxel_texture *texture, size_t *out_width, size_t *out_height, size_t *out_channels)
{
    O2V_ASSERT(texture != nullptr && texture->loaded(), "can't get metadata of empty texture");
    *out_width = texture->width;
    *out_height = texture->height;
    *out_channels = texture->channels;
}

void obj2voxel_texture_get_pixels(obj2voxel_texture *texture, obj2voxel_byte_t *out_pixels)
{
    O2V_ASSERT(texture != nullptr && texture->loaded(), "can't get pixels of empty texture");
    O2V_ASSERT(out_pixels != nullptr, "null output");
    std::memcpy(out_pixels, texture->pixels.data(), texture->pixels.size());
}

obj2voxel_error_t obj2voxel_voxelize(obj2voxel_instance *instance)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    return voxelize(*instance);
}

// The GPU path retires the CPU worker pool (reference src/threading.hpp, obj2voxel.cpp:957-985): a worker only
// registers, so that get_worker_count reflects it, and parks until stop_workers hands it an exit token.
void obj2voxel_run_worker(obj2voxel_instance *instance)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    std::unique_lock<std::mutex> lock{instance->worker_mutex};
    if (instance->workers_stopped) return;
    ++instance->worker_count;
    instance->worker_cv.wait(lock, [&] { return instance->exit_tokens != 0; });
    --instance->exit_tokens;
}

void obj2voxel_stop_workers(obj2voxel_instance *instance)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    std::lock_guard<std::mutex> lock{instance->worker_mutex};
    instance->workers_stopped = true;
    instance->exit_tokens += instance->worker_count;  // one EXIT per registered worker (reference :993-995)
    instance->worker_count = 0;
    instance->worker_cv.notify_all();
}

uint32_t obj2voxel_get_worker_count(obj2voxel_instance *instance)
{
    O2V_ASSERT(instance != nullptr, "null instance");
    std::lock_guard<std::mutex> lock{instance->worker_mutex};
    return instance->worker_count;
}

}  // extern "C"

// Extension (declared in include/o2v_hip.h): gives the cached device session - contexts, dense grids, work buffers,
// staging memory - back to the system.  The next obj2voxel_voxelize call creates a new one.
extern "C" void o2v_release_cached_device_memory(void)
{
    Session *s = nullptr;
    {
        std::lock_guard<std::mutex> lock{g_session_mutex};
        if (g_cached_session && !g_cached_busy) {
            s = g_cached_session;
            g_cached_session = nullptr;
        }
    }
    delete s;
}

// texture accessors for o2v_io.cpp (OBJ loader creates textures it owns)
namespace o2v {
obj2voxel_texture *texture_new() { return new obj2voxel_texture; }
void texture_delete(obj2voxel_texture *t) { delete t; }
bool texture_set_argb(obj2voxel_texture *t, std::vector<uint8_t> &&argb, size_t w, size_t h)
{
    t->pixels = std::move(argb);
    t->width = w;
    t->height = h;
    t->channels = 4;
    t->wrap = 1;
    return true;
}
}  // namespace o2v
