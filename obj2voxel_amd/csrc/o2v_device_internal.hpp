// o2v_device_internal.hpp -- what the in-process multi-GPU group (o2v_group.cpp) needs from a device context beyond the
// C-ABI of include/o2v_hip.h: its stream, its triangle arrays as device pointers (so that they can be filled by a
// GPU-to-GPU copy or an RCCL broadcast instead of a host copy) and the per-mesh hints.  Not part of the public interface.
#pragma once

#include "../../include/o2v_hip.h"

#include <hip/hip_runtime.h>

namespace o2v {

struct TriBuffers {
    float *verts, *uvs;
    uint32_t *types;
    float *colors;
    int32_t *texids;
    uint64_t count;
};

struct TriHints {
    bool any_textured;
    float bounds[6];
    float max_tri_extent;
    uint32_t ext_hist[256];  // triangles by the binary (biased) exponent of their extent
};

hipStream_t ctx_stream(o2v_hip_ctx *ctx);
int ctx_device(const o2v_hip_ctx *ctx);
// Sizes the triangle arrays for `count` triangles without filling them; optional arrays the mesh lacks are released.
int ctx_alloc_triangles(o2v_hip_ctx *ctx, uint64_t count, bool uvs, bool types, bool colors, bool texids);
TriBuffers ctx_tri_buffers(o2v_hip_ctx *ctx);
TriHints ctx_tri_hints(const o2v_hip_ctx *ctx);
int ctx_finish_triangles(o2v_hip_ctx *ctx, bool any_textured, const TriHints *hints);

}  // namespace o2v
