// o2v_io.hpp -- host-side triangle sources and voxel sinks (the steps either side of the hot path).
//
// Mirrors the reference's ITriangleStream / IVoxelSink pair (src/io.hpp:29-92).  These are plumbing around
// the GPU path: file parsing and encoding stay on the CPU.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

struct obj2voxel_texture;

namespace o2v {

enum class FileFormat { UNKNOWN, OBJ, STL, VL32, PLY, XYZRGB, QEF, VOX, PNG };

/// Format from an explicit extension (without dot) or, if type is null, from the path's extension.
FileFormat detect_format(const char *path, const char *type);

void log_message(int level, const std::string &msg);

/// What the reference caches per triangle (obj2voxel_triangle, src/triangle.hpp:170-195).
struct HostTriangle {
    float v[9];
    float t[6];
    uint32_t type;
    float color[3];
    const obj2voxel_texture *texture;
};

struct TriangleSource {  // reference io.hpp:29-67
    virtual ~TriangleSource() = default;
    /// The next triangle (valid until the following call), or null at the end of the stream.
    virtual const HostTriangle *next() = 0;
    /// true for the source that pulls from an obj2voxel_triangle_callback (it can be drained without the copy next() implies)
    virtual bool is_callback() const { return false; }
};

/// The bytes of a memory sink (obj2voxel_get_output_memory, include/obj2voxel.h): grown with realloc, never zero-filled - a
/// 8192^3 model is hundreds of megabytes of VL32 records, and value-initialising them (std::vector) costs as much as writing them.
struct ByteBuffer {
    uint8_t *bytes = nullptr;
    size_t size = 0, capacity = 0;
    ByteBuffer() = default;
    ByteBuffer(const ByteBuffer &) = delete;
    ByteBuffer &operator=(const ByteBuffer &) = delete;
    ~ByteBuffer();
    /// room for `more` further bytes (false: out of memory)
    bool reserve(size_t more);
    /// `n` uninitialised bytes at the end, to be written by the caller; null if out of memory
    uint8_t *append(size_t n);
};

struct VoxelSink {  // reference io.hpp:69-92
    size_t written = 0;
    virtual ~VoxelSink() = default;
    virtual bool can_write() const = 0;
    /// `voxels` holds count (x, y, z, argb) quadruples.  The batch is consumed: a sink may clobber it (the file sinks byte-swap
    /// it in place), so a caller that wants to hand one batch to two sinks must copy it.
    virtual void write(uint32_t *voxels, size_t count) = 0;
    virtual void finalize() = 0;
    /// The in-memory bytes of a memory sink, else null.
    virtual const ByteBuffer *memory() const { return nullptr; }
    /// Announces that about `voxels` more voxels are on their way (a memory sink makes room for them at once).
    virtual void expect(size_t voxels) { (void) voxels; }
};

std::unique_ptr<TriangleSource> open_stl_file(const char *path);
std::unique_ptr<TriangleSource> open_obj_file(const char *path, const obj2voxel_texture *default_texture);

std::unique_ptr<VoxelSink> open_file_sink(const char *path, FileFormat format, uint32_t resolution);
std::unique_ptr<VoxelSink> open_memory_sink(FileFormat format, uint32_t resolution);

bool read_whole_file(const char *path, std::vector<uint8_t> &out);
/// Decodes a non-interlaced PNG into 8-bit ARGB (4 bytes per pixel, alpha first).
bool decode_png_argb(const uint8_t *data, size_t size, std::vector<uint8_t> &argb, size_t &width, size_t &height,
                     std::string &err);

// implemented in o2v_api.cpp, where obj2voxel_texture is complete
obj2voxel_texture *texture_new();
void texture_delete(obj2voxel_texture *t);
bool texture_set_argb(obj2voxel_texture *t, std::vector<uint8_t> &&argb, size_t w, size_t h);

}  // namespace o2v
