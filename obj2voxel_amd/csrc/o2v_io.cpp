// o2v_io.cpp -- triangle sources (binary STL, Wavefront OBJ + MTL, PNG textures) and voxel sinks
// (VL32, PLY, XYZRGB streamed; QEF, VOX buffered; file or memory).  Formats as documented in the reference's
// README.adoc:210-264 and, for the two palette formats, in their publishers' specifications.
#include "o2v_io.hpp"

#include <sys/mman.h>

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <utility>

namespace o2v {

namespace {

enum { LOG_ERROR = 1, LOG_WARNING = 2, LOG_INFO = 3, LOG_DEBUG = 4 };

std::string lower(std::string s)
{
    for (char &c : s) c = (char) std::tolower((unsigned char) c);
    return s;
}

}  // namespace

FileFormat detect_format(const char *path, const char *type)
{
    std::string ext;
    if (type) ext = type;
    else if (path) {
        std::string p{path};
        size_t dot = p.find_last_of('.');
        if (dot == std::string::npos) return FileFormat::UNKNOWN;
        ext = p.substr(dot + 1);
    }
    ext = lower(ext);
    if (ext == "obj") return FileFormat::OBJ;
    if (ext == "stl") return FileFormat::STL;
    if (ext == "vl32") return FileFormat::VL32;
    if (ext == "ply") return FileFormat::PLY;
    if (ext == "xyzrgb" || ext == "xyz") return FileFormat::XYZRGB;
    if (ext == "qef") return FileFormat::QEF;
    if (ext == "vox") return FileFormat::VOX;
    if (ext == "png") return FileFormat::PNG;
    return FileFormat::UNKNOWN;
}

bool read_whole_file(const char *path, std::vector<uint8_t> &out)
{
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) {
        std::fclose(f);
        return false;
    }
    out.resize((size_t) n);
    size_t got = n ? std::fread(out.data(), 1, (size_t) n, f) : 0;
    std::fclose(f);
    return got == (size_t) n;
}

// ---- PNG ---------------------------------------------------------------------------------------------------

namespace {

uint32_t be32(const uint8_t *p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

static bool decode_png_argb_unguarded(const uint8_t *data, size_t size, std::vector<uint8_t> &argb, size_t &width, size_t &height,
                                      std::string &err);

// (an allocation failure must not escape through the extern "C" texture loaders)
bool decode_png_argb(const uint8_t *data, size_t size, std::vector<uint8_t> &argb, size_t &width, size_t &height,
                     std::string &err)
{
    try {
        return decode_png_argb_unguarded(data, size, argb, width, height, err);
    }
    catch (const std::bad_alloc &) {
        err = "out of memory";
        return false;
    }
}

static bool decode_png_argb_unguarded(const uint8_t *data, size_t size, std::vector<uint8_t> &argb, size_t &width, size_t &height,
                                      std::string &err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 || std::memcmp(data, sig, 8) != 0) {
        err = "not a PNG file";
        return false;
    }
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool have_ihdr = false;
    while (pos + 12 <= size) {
        uint32_t len = be32(data + pos);
        const uint8_t *tag = data + pos + 4;
        const uint8_t *body = data + pos + 8;
        if (pos + 12 + (size_t) len > size) {
            err = "truncated chunk";
            return false;
        }
        if (!std::memcmp(tag, "IHDR", 4) && len >= 13) {
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
            have_ihdr = true;
        }
        else if (!std::memcmp(tag, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(tag, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(tag, "IEND", 4)) break;
        pos += 12 + (size_t) len;
    }
    if (!have_ihdr || w == 0 || h == 0) {
        err = "missing IHDR";
        return false;
    }
    if ((uint64_t) w * h > (1ull << 28)) {  // 268 M texels: a hostile header must not force a huge allocation
        err = "image too large";
        return false;
    }
    if (interlace) {
        err = "interlaced PNG is not supported";
        return false;
    }
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    // bit depths the PNG specification allows per colour type: grey 1/2/4/8/16, palette 1/2/4/8, the others 8/16
    const bool depth_ok = ctype == 0   ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                          : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                       : (depth == 8 || depth == 16);
    if (!channels || !depth_ok) {
        err = "unsupported colour type / bit depth";
        return false;
    }
    const size_t bpp_bits = (size_t) channels * depth;
    const size_t stride = (w * bpp_bits + 7) / 8;
    const size_t bpp = std::max<size_t>(1, bpp_bits / 8);
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf) raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong) idat.size()) != Z_OK || raw_len != raw.size()) {
        err = "zlib inflate failed";
        return false;
    }
    // undo the scanline filters
    std::vector<uint8_t> img(stride * h);
    for (size_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[y * (stride + 1)];
        const uint8_t *in = &raw[y * (stride + 1) + 1];
        uint8_t *out = &img[y * stride];
        const uint8_t *up = y ? &img[(y - 1) * stride] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= bpp ? out[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = in[x];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: v += paeth(a, b, c); break;
            default: err = "bad filter type"; return false;
            }
            out[x] = (uint8_t) v;
        }
    }
    argb.resize((size_t) w * h * 4);
    for (size_t y = 0; y < h; ++y) {
        const uint8_t *row = &img[y * stride];
        for (size_t x = 0; x < w; ++x) {
            uint8_t r = 0, g = 0, b = 0, a = 255;
            auto sample = [&](size_t idx) -> uint8_t {
                if (depth == 8) return row[idx];
                if (depth == 16) return row[idx * 2];
                size_t bit = idx * depth;
                uint8_t v = (row[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
                return ctype == 3 ? v : (uint8_t) (v * 255 / ((1 << depth) - 1));
            };
            switch (ctype) {
            case 0: r = g = b = sample(x); break;
            case 2: r = sample(x * 3); g = sample(x * 3 + 1); b = sample(x * 3 + 2); break;
            case 3: {
                size_t i = sample(x);
                if (i * 3 + 2 < palette.size()) {
                    r = palette[i * 3];
                    g = palette[i * 3 + 1];
                    b = palette[i * 3 + 2];
                }
                if (i < trns.size()) a = trns[i];
                break;
            }
            case 4: r = g = b = sample(x * 2); a = sample(x * 2 + 1); break;
            case 6: r = sample(x * 4); g = sample(x * 4 + 1); b = sample(x * 4 + 2); a = sample(x * 4 + 3); break;
            }
            uint8_t *o = &argb[(y * w + x) * 4];
            o[0] = a; o[1] = r; o[2] = g; o[3] = b;
        }
    }
    width = w;
    height = h;
    return true;
}

// ---- STL (reference io.cpp:395-435; binary only, like the reference) ------------------------------------------

namespace {

struct VectorTriangleSource final : TriangleSource {
    std::vector<HostTriangle> tris;
    size_t index = 0;
    std::vector<obj2voxel_texture *> owned_textures;
    ~VectorTriangleSource() override
    {
        for (auto *t : owned_textures) texture_delete(t);
    }
    const HostTriangle *next() override { return index < tris.size() ? &tris[index++] : nullptr; }
};

}  // namespace

std::unique_ptr<TriangleSource> open_stl_file(const char *path)
{
    std::vector<uint8_t> bytes;
    if (!read_whole_file(path, bytes)) {
        log_message(LOG_ERROR, std::string("Failed to open STL file: \"") + path + "\"");
        return nullptr;
    }
    if (bytes.size() < 84) {
        log_message(LOG_ERROR, "Binary STL file must start with a header of 80 characters and a triangle count");
        return nullptr;
    }
    if (std::memcmp(bytes.data(), "solid", 5) == 0) {
        log_message(LOG_ERROR, "The given file is an ASCII STL file which is not supported");
        return nullptr;
    }
    uint32_t count;
    std::memcpy(&count, bytes.data() + 80, 4);
    if (bytes.size() < 84 + (size_t) count * 50) {
        log_message(LOG_ERROR, "Unexpected EOF when reading STL triangles");
        return nullptr;
    }
    auto src = std::make_unique<VectorTriangleSource>();
    src->tris.resize(count);
    for (uint32_t i = 0; i < count; ++i) {
        // 12 floats (normal, v0, v1, v2) + u16 attribute.  All three vertices of record i are used (the
        // reference's StlTriangleStream::next, io.cpp:171-187, is off by one vertex; not reproduced).
        HostTriangle &t = src->tris[i];
        std::memset(&t, 0, sizeof(t));
        std::memcpy(t.v, bytes.data() + 84 + (size_t) i * 50 + 12, 36);
        t.type = 1;  // MATERIALLESS
    }
    return src;
}

// ---- OBJ + MTL (the subset obj2voxel consumes through tinyobjloader: io.cpp:244-312,351-393) -----------------

namespace {

struct Material {
    std::string name;
    float kd[3] = {0.6f, 0.6f, 0.6f};  // tinyobjloader's default diffuse is 0.6 when Kd is absent
    std::string map_kd;
};

std::string dir_of(const std::string &p)
{
    size_t s = p.find_last_of("/\\");
    return s == std::string::npos ? std::string{} : p.substr(0, s + 1);
}

void load_mtl(const std::string &path, std::vector<Material> &materials, std::map<std::string, int> &by_name)
{
    std::ifstream in(path);
    if (!in) {
        log_message(LOG_WARNING, "Material file \"" + path + "\" not found");
        return;
    }
    std::string line;
    Material *cur = nullptr;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string key;
        if (!(ss >> key)) continue;
        if (key == "newmtl") {
            Material m;
            ss >> m.name;
            by_name[m.name] = (int) materials.size();
            materials.push_back(m);
            cur = &materials.back();
        }
        else if (!cur) continue;
        else if (key == "Kd") ss >> cur->kd[0] >> cur->kd[1] >> cur->kd[2];
        else if (key == "map_Kd") {
            std::string rest;
            std::getline(ss, rest);
            size_t b = rest.find_first_not_of(" \t");
            size_t e = rest.find_last_not_of(" \t\r");
            if (b != std::string::npos) cur->map_kd = rest.substr(b, e - b + 1);
        }
    }
}

}  // namespace

std::unique_ptr<TriangleSource> open_obj_file(const char *path, const obj2voxel_texture *default_texture)
{
    std::ifstream in(path);
    if (!in) {
        log_message(LOG_ERROR, std::string("Failed to open OBJ file: \"") + path + "\"");
        return nullptr;
    }
    const std::string base = dir_of(path);
    std::vector<float> pos, tex;
    std::vector<Material> materials;
    std::map<std::string, int> material_by_name;
    std::map<std::string, obj2voxel_texture *> textures;
    auto src = std::make_unique<VectorTriangleSource>();
    int cur_material = -1;
    std::string line;
    struct Idx {
        long v, t;
    };
    std::vector<Idx> face;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key;
        if (!(ss >> key)) continue;
        if (key == "v") {
            float x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            pos.insert(pos.end(), {x, y, z});
        }
        else if (key == "vt") {
            float u = 0, v = 0;
            ss >> u >> v;
            tex.insert(tex.end(), {u, v});
        }
        else if (key == "mtllib") {
            std::string name;
            ss >> name;
            load_mtl(base + name, materials, material_by_name);
        }
        else if (key == "usemtl") {
            std::string name;
            ss >> name;
            auto it = material_by_name.find(name);
            cur_material = it == material_by_name.end() ? -1 : it->second;
        }
        else if (key == "f") {
            face.clear();
            std::string item;
            while (ss >> item) {
                Idx ix{0, 0};
                ix.v = std::strtol(item.c_str(), nullptr, 10);
                size_t s1 = item.find('/');
                if (s1 != std::string::npos && s1 + 1 < item.size() && item[s1 + 1] != '/')
                    ix.t = std::strtol(item.c_str() + s1 + 1, nullptr, 10);
                const long nv = (long) (pos.size() / 3), nt = (long) (tex.size() / 2);
                ix.v = ix.v < 0 ? nv + ix.v : ix.v - 1;
                ix.t = ix.t < 0 ? nt + ix.t : ix.t - 1;  // -1 when absent
                if (ix.v < 0 || ix.v >= nv) {
                    log_message(LOG_ERROR, "OBJ face references a vertex that does not exist");
                    return nullptr;
                }
                if (ix.t >= nt) ix.t = -1;
                face.push_back(ix);
            }
            // fan triangulation, as tinyobjloader does for convex polygons
            for (size_t k = 1; k + 1 < face.size(); ++k) {
                const Idx tri[3] = {face[0], face[k], face[k + 1]};
                HostTriangle t;
                std::memset(&t, 0, sizeof(t));
                bool has_uv = true;
                for (int c = 0; c < 3; ++c) {
                    std::memcpy(&t.v[c * 3], &pos[(size_t) tri[c].v * 3], 12);
                    if (tri[c].t >= 0) std::memcpy(&t.t[c * 2], &tex[(size_t) tri[c].t * 2], 8);
                    else has_uv = false;
                }
                if (!has_uv) std::memset(t.t, 0, sizeof(t.t));
                // material decision: reference io.cpp:277-303
                const Material *m = cur_material < 0 ? nullptr : &materials[(size_t) cur_material];
                if (!m) {
                    if (has_uv && default_texture) {
                        t.type = 3;
                        t.texture = default_texture;
                    }
                    else t.type = 1;
                }
                else {
                    const obj2voxel_texture *tx = nullptr;
                    if (has_uv && !m->map_kd.empty()) {
                        auto it = textures.find(m->map_kd);
                        if (it == textures.end()) {
                            std::string file = m->map_kd;
                            std::replace(file.begin(), file.end(), '\\', '/');
                            std::vector<uint8_t> bytes, argb;
                            size_t w = 0, h = 0;
                            std::string err;
                            obj2voxel_texture *made = nullptr;
                            if (!read_whole_file((base + file).c_str(), bytes) && !read_whole_file(file.c_str(), bytes))
                                log_message(LOG_WARNING, "Failed to open texture file \"" + file + "\" of material \"" + m->name + "\"");
                            else if (!decode_png_argb(bytes.data(), bytes.size(), argb, w, h, err))
                                log_message(LOG_WARNING, "Failed to decode texture \"" + file + "\": " + err);
                            else {
                                made = texture_new();
                                texture_set_argb(made, std::move(argb), w, h);
                                src->owned_textures.push_back(made);
                                log_message(LOG_INFO, "Loaded texture \"" + file + "\"");
                            }
                            it = textures.emplace(m->map_kd, made).first;
                        }
                        tx = it->second;
                    }
                    if (tx) {
                        t.type = 3;
                        t.texture = tx;
                    }
                    else {
                        t.type = 2;
                        std::memcpy(t.color, m->kd, sizeof(t.color));
                    }
                }
                src->tris.push_back(t);
            }
        }
    }
    return src;
}

// ---- sinks ---------------------------------------------------------------------------------------------------

ByteBuffer::~ByteBuffer() { std::free(bytes); }
bool ByteBuffer::reserve(size_t more)
{
    if (size + more <= capacity) return true;
    // (large blocks are moved by remapping their pages, not by copying: growth is cheap; a grown block of 2 MiB or more asks
    // for huge pages - a fresh gigabyte is a quarter of a million page faults otherwise)
    const size_t want = std::max<size_t>(size + more, capacity + capacity / 2);
    void *p = std::realloc(bytes, want);
    if (!p) return false;
    bytes = static_cast<uint8_t *>(p);
    capacity = want;
#ifdef MADV_HUGEPAGE
    if (want >= (2u << 20)) {
        const uintptr_t a = (reinterpret_cast<uintptr_t>(bytes) + 4095u) & ~(uintptr_t) 4095u, e = (reinterpret_cast<uintptr_t>(bytes) + want) & ~(uintptr_t) 4095u;
        if (e > a) (void) madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
    }
#endif
    return true;
}
uint8_t *ByteBuffer::append(size_t n)
{
    if (!reserve(n)) return nullptr;
    uint8_t *at = bytes + size;
    size += n;
    return at;
}

namespace {

// Writes through a FILE* or into a byte buffer.
struct ByteOut {
    std::FILE *file = nullptr;
    ByteBuffer mem;
    bool ok = true;
    ~ByteOut()
    {
        if (file) std::fclose(file);
    }
    void put(const void *p, size_t n)
    {
        if (!ok || !n) return;
        if (file) ok = std::fwrite(p, 1, n, file) == n;
        else {
            uint8_t *dst = mem.append(n);
            ok = dst != nullptr;
            if (dst) std::memcpy(dst, p, n);
        }
    }
    void patch(size_t offset, const void *p, size_t n)
    {
        if (!ok) return;
        if (file) {
            long cur = std::ftell(file);
            ok = std::fseek(file, (long) offset, SEEK_SET) == 0 && std::fwrite(p, 1, n, file) == n &&
                 std::fseek(file, cur, SEEK_SET) == 0;
        }
        else std::memcpy(mem.bytes + offset, p, n);
    }
    void flush()
    {
        if (file && ok) ok = std::fflush(file) == 0;
    }
};

void put_be32(uint8_t *o, uint32_t v)
{
    o[0] = (uint8_t) (v >> 24);
    o[1] = (uint8_t) (v >> 16);
    o[2] = (uint8_t) (v >> 8);
    o[3] = (uint8_t) v;
}

// VL32: big-endian (x, y, z) int32 then a, r, g, b bytes (README.adoc:233-252).  PLY: 300-byte header followed
// by the same records (README.adoc:214-231).  XYZRGB: one "x y z r g b" text line per voxel.
struct ListSink final : VoxelSink {
    ByteOut out;
    FileFormat format;
    bool is_memory;
    bool finalized = false;
    static constexpr size_t kPlyHeader = 300;
    size_t count_offset = 0;
    ListSink(FileFormat f, bool memory) : format{f}, is_memory{memory} {}

    void begin()
    {
        if (format != FileFormat::PLY) return;
        std::string h = "ply\nformat binary_big_endian 1.0\nelement vertex ";
        count_offset = h.size();
        h += "00000000000000000000\n";  // patched in finalize(); fixed width keeps the header at 300 bytes
        h += "property int x\nproperty int y\nproperty int z\n"
             "property uchar alpha\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n";
        const std::string tail = "end_header\n";
        std::string pad = "comment ";
        const size_t used = h.size() + tail.size() + pad.size() + 1;
        pad += std::string(kPlyHeader > used ? kPlyHeader - used : 0, ' ');
        pad += "\n";
        h += pad + tail;
        out.put(h.data(), h.size());
    }
    bool can_write() const override { return out.ok; }
    void write(uint32_t *voxels, size_t count) override
    {
        written += count;
        if (format == FileFormat::XYZRGB) {
            std::string s;
            char buf[96];
            for (size_t i = 0; i < count; ++i) {
                const uint32_t *v = voxels + i * 4;
                int n = std::snprintf(buf, sizeof(buf), "%u %u %u %u %u %u\n", v[0], v[1], v[2], (v[3] >> 16) & 255u,
                                      (v[3] >> 8) & 255u, v[3] & 255u);
                s.append(buf, (size_t) n);
            }
            out.put(s.data(), s.size());
            return;
        }
        // big-endian words: into the memory sink's buffer directly, or in place and out to the file (VoxelSink::write: the batch is
        // the sink's to clobber - it is the read-back staging buffer, overwritten by the next batch anyway)
        static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "the byte swap below makes big-endian words on a little-endian host");
        if (out.file) {
            for (size_t i = 0; i < count * 4; ++i) voxels[i] = __builtin_bswap32(voxels[i]);
            out.put(voxels, count * 16);
            return;
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(out.mem.append(count * 16));  // (16-byte records after a 0- or 300-byte header: 4-byte aligned)
        out.ok = out.ok && dst != nullptr;
        if (!dst) return;
        for (size_t i = 0; i < count * 4; ++i) dst[i] = __builtin_bswap32(voxels[i]);
    }
    void expect(size_t voxels) override
    {
        if (!out.file && format != FileFormat::XYZRGB && !out.mem.reserve(voxels * 16)) {
            // (said here, where it happens: a failed sink otherwise only shows as an IO error after the voxelization)
            if (out.ok) log_message(LOG_ERROR, "out of memory: the output buffer for " + std::to_string(voxels) + " voxels (" + std::to_string(voxels * 16) + " bytes) could not be reserved");
            out.ok = false;
        }
    }
    void finalize() override
    {
        if (finalized) return;
        finalized = true;
        if (format == FileFormat::PLY) {
            char digits[32];
            std::snprintf(digits, sizeof(digits), "%020llu", (unsigned long long) written);
            out.patch(count_offset, digits, 20);
        }
        out.flush();
    }
    const ByteBuffer *memory() const override { return is_memory ? &out.mem : nullptr; }
};

// ---- palette formats -----------------------------------------------------------------
// QEF and VOX index a colour table, so every voxel is buffered (16 bytes each, as README.adoc:274-275 says of the
// reference) and the file is written in finalize(), like the reference's VoxelioVoxelSink does for paletted writers
// (src/io.cpp:575-633).  The encoders themselves live in the absent voxel-io module; what is written here follows the
// published formats (Qubicle Exchange Format 0.2; MagicaVoxel .vox version 150) and is not byte-pinned to voxel-io.

// Colour table of at most `limit` entries for `colors` (distinct ARGB values, ascending) with voxel counts `weight`:
// median cut in RGB - the box with the largest weighted extent is split at the weighted median of its longest axis;
// every box is represented by its weighted mean.  map[i] receives the table index of colors[i].
void reduce_palette(const std::vector<uint32_t> &colors, const std::vector<uint64_t> &weight, size_t limit,
                    std::vector<uint32_t> &table, std::vector<uint32_t> &map)
{
    const size_t n = colors.size();
    map.assign(n, 0);
    table.clear();
    if (n <= limit) {
        table = colors;
        for (size_t i = 0; i < n; ++i) map[i] = (uint32_t) i;
        return;
    }
    struct Box {
        size_t begin, end;  // range in `order`
        uint64_t score;
        int axis;
    };
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t) i;
    auto channel = [&](uint32_t idx, int axis) { return (int) ((colors[idx] >> (16 - 8 * axis)) & 255u); };
    auto measure = [&](Box &b) {
        int lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0};
        uint64_t w = 0;
        for (size_t k = b.begin; k < b.end; ++k) {
            for (int a = 0; a < 3; ++a) {
                const int c = channel(order[k], a);
                lo[a] = std::min(lo[a], c);
                hi[a] = std::max(hi[a], c);
            }
            w += weight[order[k]];
        }
        b.axis = 0;
        for (int a = 1; a < 3; ++a)
            if (hi[a] - lo[a] > hi[b.axis] - lo[b.axis]) b.axis = a;
        const int extent = hi[b.axis] - lo[b.axis];
        b.score = b.end - b.begin < 2 || extent == 0 ? 0 : (uint64_t) extent * w;
    };
    std::vector<Box> boxes{Box{0, n, 0, 0}};
    measure(boxes[0]);
    while (boxes.size() < limit) {
        size_t best = 0;
        for (size_t k = 1; k < boxes.size(); ++k)
            if (boxes[k].score > boxes[best].score) best = k;
        if (boxes[best].score == 0) break;
        Box b = boxes[best];
        std::sort(order.begin() + (long) b.begin, order.begin() + (long) b.end, [&](uint32_t x, uint32_t y) {
            const int cx = channel(x, b.axis), cy = channel(y, b.axis);
            return cx != cy ? cx < cy : colors[x] < colors[y];
        });
        uint64_t total = 0, run = 0;
        for (size_t k = b.begin; k < b.end; ++k) total += weight[order[k]];
        size_t cut = b.begin + 1;
        for (size_t k = b.begin; k + 1 < b.end; ++k) {
            run += weight[order[k]];
            cut = k + 1;
            if (run * 2 >= total) break;
        }
        Box lo_box{b.begin, cut, 0, 0}, hi_box{cut, b.end, 0, 0};
        measure(lo_box);
        measure(hi_box);
        boxes[best] = lo_box;
        boxes.push_back(hi_box);
    }
    for (size_t k = 0; k < boxes.size(); ++k) {
        uint64_t sum[3] = {0, 0, 0}, w = 0;
        for (size_t i = boxes[k].begin; i < boxes[k].end; ++i) {
            const uint64_t wi = weight[order[i]];
            for (int a = 0; a < 3; ++a) sum[a] += wi * (uint64_t) channel(order[i], a);
            w += wi;
            map[order[i]] = (uint32_t) k;
        }
        uint32_t c = 0xFF000000u;
        for (int a = 0; a < 3; ++a) c |= (uint32_t) ((sum[a] + w / 2) / w) << (16 - 8 * a);
        table.push_back(c);
    }
}

struct PaletteSink final : VoxelSink {
    ByteOut out;
    FileFormat format;
    bool is_memory;
    bool finalized = false;
    uint32_t resolution;
    std::vector<uint32_t> voxels;  // (x, y, z, argb) as received

    PaletteSink(FileFormat f, bool memory, uint32_t res) : format{f}, is_memory{memory}, resolution{res} {}

    bool can_write() const override { return out.ok; }
    void write(uint32_t *v, size_t count) override
    {
        written += count;
        voxels.insert(voxels.end(), v, v + count * 4);
    }
    const ByteBuffer *memory() const override { return is_memory ? &out.mem : nullptr; }

    // distinct colours (ascending) with their voxel counts; index[i] = position of voxel i's colour
    void collect_colors(std::vector<uint32_t> &colors, std::vector<uint64_t> &weight, std::vector<uint32_t> &index) const
    {
        const size_t n = voxels.size() / 4;
        colors.resize(n);
        for (size_t i = 0; i < n; ++i) colors[i] = voxels[i * 4 + 3];
        std::sort(colors.begin(), colors.end());
        colors.erase(std::unique(colors.begin(), colors.end()), colors.end());
        weight.assign(colors.size(), 0);
        index.resize(n);
        for (size_t i = 0; i < n; ++i) {
            index[i] = (uint32_t) (std::lower_bound(colors.begin(), colors.end(), voxels[i * 4 + 3]) - colors.begin());
            weight[index[i]] += 1;
        }
    }

    void write_qef()
    {
        // Qubicle Exchange Format 0.2: three header lines, "sizeX sizeY sizeZ", colour count, one "r g b" line
        // (0..1) per colour, then one "x y z colourIndex mask" line per voxel.  mask: 2 = -x face visible, 4 = +x,
        // 8 = +y, 16 = -y, 32 = +z, 64 = -z (a face is visible if the neighbour cell is empty); 0 = enclosed.
        std::vector<uint32_t> colors, index;
        std::vector<uint64_t> weight;
        collect_colors(colors, weight, index);
        const size_t n = voxels.size() / 4;
        std::vector<uint64_t> keys(n);
        auto key = [](uint32_t x, uint32_t y, uint32_t z) { return (uint64_t) x | ((uint64_t) y << 21) | ((uint64_t) z << 42); };
        for (size_t i = 0; i < n; ++i) keys[i] = key(voxels[i * 4], voxels[i * 4 + 1], voxels[i * 4 + 2]);
        std::sort(keys.begin(), keys.end());
        auto filled = [&](int64_t x, int64_t y, int64_t z) {
            if (x < 0 || y < 0 || z < 0) return false;
            return std::binary_search(keys.begin(), keys.end(), key((uint32_t) x, (uint32_t) y, (uint32_t) z));
        };
        std::string s = "Qubicle Exchange Format\nVersion 0.2\nwww.minddesk.com\n";
        char buf[128];
        std::snprintf(buf, sizeof(buf), "%u %u %u\n%zu\n", resolution, resolution, resolution, colors.size());
        s += buf;
        for (uint32_t c : colors) {
            std::snprintf(buf, sizeof(buf), "%f %f %f\n", (double) ((c >> 16) & 255u) / 255.0, (double) ((c >> 8) & 255u) / 255.0,
                          (double) (c & 255u) / 255.0);
            s += buf;
        }
        out.put(s.data(), s.size());
        s.clear();
        for (size_t i = 0; i < n; ++i) {
            const int64_t x = voxels[i * 4], y = voxels[i * 4 + 1], z = voxels[i * 4 + 2];
            const unsigned mask = (filled(x - 1, y, z) ? 0u : 2u) | (filled(x + 1, y, z) ? 0u : 4u) | (filled(x, y + 1, z) ? 0u : 8u) |
                                  (filled(x, y - 1, z) ? 0u : 16u) | (filled(x, y, z + 1) ? 0u : 32u) | (filled(x, y, z - 1) ? 0u : 64u);
            const int len = std::snprintf(buf, sizeof(buf), "%lld %lld %lld %u %u\n", (long long) x, (long long) y, (long long) z,
                                          index[i], mask);
            s.append(buf, (size_t) len);
            if (s.size() > (1u << 20)) {
                out.put(s.data(), s.size());
                s.clear();
            }
        }
        out.put(s.data(), s.size());
    }

    // ---- MagicaVoxel ----
    static void put_le32(std::vector<uint8_t> &b, uint32_t v)
    {
        for (int k = 0; k < 4; ++k) b.push_back((uint8_t) (v >> (8 * k)));
    }
    static void put_str(std::vector<uint8_t> &b, const std::string &str)
    {
        put_le32(b, (uint32_t) str.size());
        b.insert(b.end(), str.begin(), str.end());
    }
    static void put_dict(std::vector<uint8_t> &b, const std::vector<std::pair<std::string, std::string>> &kv)
    {
        put_le32(b, (uint32_t) kv.size());
        for (const auto &e : kv) {
            put_str(b, e.first);
            put_str(b, e.second);
        }
    }
    static void put_chunk(std::vector<uint8_t> &b, const char *id, const std::vector<uint8_t> &content)
    {
        b.insert(b.end(), id, id + 4);
        put_le32(b, (uint32_t) content.size());
        put_le32(b, 0);
        b.insert(b.end(), content.begin(), content.end());
    }

    void write_vox()
    {
        // .vox version 150: MAIN { (SIZE XYZI)* [nTRN nGRP (nTRN nSHP)*] RGBA }.  A model holds at most 256^3 cells with
        // byte coordinates, so a larger grid becomes one model per non-empty 256^3 block, placed by a scene graph
        // (a translation node's "_t" is the world position of the model's centre floor(size / 2)).  Colour index i of
        // XYZI refers to RGBA entry i - 1; index 0 is reserved, leaving 255 usable colours (README.adoc:253-257).
        // Axes are written as they are (MagicaVoxel shows z up).
        std::vector<uint32_t> colors, index, table, map;
        std::vector<uint64_t> weight;
        collect_colors(colors, weight, index);
        reduce_palette(colors, weight, 255, table, map);
        if (colors.size() > 255)
            log_message(LOG_INFO, "VOX: reduced " + std::to_string(colors.size()) + " colours to a palette of " +
                                      std::to_string(table.size()));
        const size_t n = voxels.size() / 4;
        const uint32_t blocks = (resolution + 255u) / 256u;
        std::map<uint32_t, std::vector<uint8_t>> models;  // block id -> XYZI payload without the count
        for (size_t i = 0; i < n; ++i) {
            const uint32_t x = voxels[i * 4], y = voxels[i * 4 + 1], z = voxels[i * 4 + 2];
            const uint32_t id = ((z >> 8) * blocks + (y >> 8)) * blocks + (x >> 8);
            std::vector<uint8_t> &m = models[id];
            m.push_back((uint8_t) x);
            m.push_back((uint8_t) y);
            m.push_back((uint8_t) z);
            m.push_back((uint8_t) (map[index[i]] + 1u));
        }
        if (models.empty()) models[0];  // an empty grid is still one (empty) model
        std::vector<uint8_t> body;
        auto block_size = [&](uint32_t b) { return std::min<uint32_t>(256u, resolution - b * 256u); };
        for (const auto &m : models) {
            const uint32_t bx = m.first % blocks, by = (m.first / blocks) % blocks, bz = m.first / (blocks * blocks);
            std::vector<uint8_t> c;
            put_le32(c, block_size(bx));
            put_le32(c, block_size(by));
            put_le32(c, block_size(bz));
            put_chunk(body, "SIZE", c);
            c.clear();
            put_le32(c, (uint32_t) (m.second.size() / 4));
            c.insert(c.end(), m.second.begin(), m.second.end());
            put_chunk(body, "XYZI", c);
        }
        if (models.size() > 1 || blocks > 1) {
            std::vector<uint8_t> c;
            put_le32(c, 0);  // root transform
            put_dict(c, {});
            put_le32(c, 1);
            put_le32(c, 0xffffffffu);
            put_le32(c, 0xffffffffu);
            put_le32(c, 1);
            put_dict(c, {});
            put_chunk(body, "nTRN", c);
            c.clear();
            put_le32(c, 1);  // group of all models
            put_dict(c, {});
            put_le32(c, (uint32_t) models.size());
            for (uint32_t k = 0; k < models.size(); ++k) put_le32(c, 2u + 2u * k);
            put_chunk(body, "nGRP", c);
            uint32_t k = 0;
            for (const auto &m : models) {
                const uint32_t bx = m.first % blocks, by = (m.first / blocks) % blocks, bz = m.first / (blocks * blocks);
                const std::string t = std::to_string(bx * 256u + block_size(bx) / 2u) + " " +
                                      std::to_string(by * 256u + block_size(by) / 2u) + " " +
                                      std::to_string(bz * 256u + block_size(bz) / 2u);
                c.clear();
                put_le32(c, 2u + 2u * k);
                put_dict(c, {});
                put_le32(c, 3u + 2u * k);
                put_le32(c, 0xffffffffu);
                put_le32(c, 0);
                put_le32(c, 1);
                put_dict(c, {{"_t", t}});
                put_chunk(body, "nTRN", c);
                c.clear();
                put_le32(c, 3u + 2u * k);
                put_dict(c, {});
                put_le32(c, 1);
                put_le32(c, k);
                put_dict(c, {});
                put_chunk(body, "nSHP", c);
                ++k;
            }
        }
        {
            std::vector<uint8_t> c;
            for (uint32_t i = 0; i < 256; ++i) {
                const uint32_t argb = i < table.size() ? table[i] : 0u;
                c.push_back((uint8_t) (argb >> 16));
                c.push_back((uint8_t) (argb >> 8));
                c.push_back((uint8_t) argb);
                c.push_back((uint8_t) (argb >> 24));
            }
            put_chunk(body, "RGBA", c);
        }
        std::vector<uint8_t> file{'V', 'O', 'X', ' '};
        put_le32(file, 150);
        file.insert(file.end(), {'M', 'A', 'I', 'N'});
        put_le32(file, 0);
        put_le32(file, (uint32_t) body.size());
        out.put(file.data(), file.size());
        out.put(body.data(), body.size());
    }

    void finalize() override
    {
        if (finalized) return;
        finalized = true;
        if (!out.ok) return;
        if (format == FileFormat::QEF) write_qef();
        else write_vox();
        out.flush();
        voxels.clear();
        voxels.shrink_to_fit();
    }
};

std::unique_ptr<VoxelSink> make_list_sink(FileFormat format, const char *path, uint32_t resolution)
{
    const bool paletted = format == FileFormat::QEF || format == FileFormat::VOX;
    if (!paletted && format != FileFormat::VL32 && format != FileFormat::PLY && format != FileFormat::XYZRGB) {
        log_message(LOG_ERROR, "Not an output format (this build writes VL32, PLY, XYZRGB, QEF and VOX)");
        return nullptr;
    }
    std::FILE *file = nullptr;
    if (path) {
        file = std::fopen(path, "wb");
        if (!file) {
            log_message(LOG_ERROR, std::string("Failed to open output file: \"") + path + "\"");
            return nullptr;
        }
    }
    if (paletted) {
        auto sink = std::make_unique<PaletteSink>(format, path == nullptr, resolution);
        sink->out.file = file;
        return sink;
    }
    auto sink = std::make_unique<ListSink>(format, path == nullptr);
    sink->out.file = file;
    sink->begin();
    return sink;
}

}  // namespace

std::unique_ptr<VoxelSink> open_file_sink(const char *path, FileFormat format, uint32_t resolution)
{
    return make_list_sink(format, path, resolution);
}
std::unique_ptr<VoxelSink> open_memory_sink(FileFormat format, uint32_t resolution)
{
    return make_list_sink(format, nullptr, resolution);
}

}  // namespace o2v
