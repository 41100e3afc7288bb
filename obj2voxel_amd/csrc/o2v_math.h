// o2v_math.h -- exact float32 building blocks shared by host C++ and the gfx950 kernels.
//
// Every function here reproduces, operation for operation and in the same order, the float32 arithmetic of
// the reference's hot path (cited as file:line relative to the reference tree).  The translation units that
// include this header MUST be compiled with -ffp-contract=off: a fused multiply-add changes voxel colours
// and, in rare cases, occupancy.  Division and sqrt must be correctly rounded
// (hipcc's default, -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define O2V_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define O2V_HD inline
#endif

namespace o2v {

// constants.hpp:10-15, voxelization.cpp:15
constexpr uint32_t kSubdivisionVolumeLimit = 512;
constexpr float kDiagonalityLimit = 0.5f;
constexpr float kEpsilon = 1.0f / (1 << 16);
constexpr float kPlaneDistanceLimit = 2.0f;  // voxelization.cpp:435

// triangle.hpp:21-29
enum TriangleType : uint32_t { kTriNone = 0, kTriMaterialless = 1, kTriUntextured = 2, kTriTextured = 3 };

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};

O2V_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
O2V_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

// voxel-io dot (absent module; restated as a sequential sum starting from zero)
O2V_HD float dot(V3 a, V3 b)
{
    float r = 0;
    r += a.x * b.x;
    r += a.y * b.y;
    r += a.z * b.z;
    return r;
}
O2V_HD V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
O2V_HD float comp(V3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
O2V_HD float fmin2(float a, float b) { return (b < a) ? b : a; }  // std::min
O2V_HD float fmax2(float a, float b) { return (a < b) ? b : a; }  // std::max

// Correctly rounded sqrt.  On gfx950 plain sqrtf() gets the IEEE expansion (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn() lowers to the bare approximate v_sqrt_f32 and must
// not be used here (it changed triangle areas by 1 ulp, visible as 1-LSB BLEND colour differences).
O2V_HD float sqrt_rn(float x) { return sqrtf(x); }
O2V_HD float floor_f(float x) { return floorf(x); }
O2V_HD float abs_f(float x) { return fabsf(x); }

// util.hpp:122-146
O2V_HD float length(V3 a) { return sqrt_rn(dot(a, a)); }
O2V_HD V3 normalize(V3 a)
{
    float l = length(a);
    return {a.x / l, a.y / l, a.z / l};
}
O2V_HD V3 mix(V3 a, V3 b, float t)
{
    float s = 1 - t;
    return {s * a.x + t * b.x, s * a.y + t * b.y, s * a.z + t * b.z};
}
O2V_HD V2 mix(V2 a, V2 b, float t)
{
    float s = 1 - t;
    return {s * a.x + t * b.x, s * a.y + t * b.y};
}

// triangle.hpp:59-106
O2V_HD V3 tri_normal(V3 v0, V3 v1, V3 v2) { return cross(v1 - v0, v2 - v0); }
O2V_HD float tri_area(V3 v0, V3 v1, V3 v2) { return length(tri_normal(v0, v1, v2)) / 2; }
O2V_HD V3 tri_min(V3 a, V3 b, V3 c)
{
    return {fmin2(a.x, fmin2(b.x, c.x)), fmin2(a.y, fmin2(b.y, c.y)), fmin2(a.z, fmin2(b.z, c.z))};
}
O2V_HD V3 tri_max(V3 a, V3 b, V3 c)
{
    return {fmax2(a.x, fmax2(b.x, c.x)), fmax2(a.y, fmax2(b.y, c.y)), fmax2(a.z, fmax2(b.z, c.z))};
}

// floor(x).cast<u32>() (triangle.hpp:91-100).  Negative / huge inputs are outside the reference's contract
// (UB there); both compilers used here saturate or wrap without trapping.
O2V_HD uint32_t floor_u32(float x) { return (uint32_t) floor_f(x); }

// voxelization.cpp:335-347
O2V_HD bool roughly_axis_aligned(V3 v0, V3 v1, V3 v2)
{
    const float s = 0.5773502691896257645091487805019574556476017512701268760186023264f;
    V3 n = tri_normal(v0, v1, v2);
    V3 an = {abs_f(n.x), abs_f(n.y), abs_f(n.z)};
    V3 nn = normalize(an);
    float d = dot(nn, V3{s, s, s});
    float d01 = (d - s) / (1 - s);
    return d01 < kDiagonalityLimit;
}

// ---- affine transform (util.hpp:212-281) ------------------------------------------------------------------
struct Affine {
    V3 m[3];
    V3 t;
};

O2V_HD Affine affine_scale(float s, V3 t) { return {{{s, 0, 0}, {0, s, 0}, {0, 0, s}}, t}; }
O2V_HD V3 affine_col(const Affine &a, uint32_t j) { return {comp(a.m[0], j), comp(a.m[1], j), comp(a.m[2], j)}; }
O2V_HD Affine affine_mul(const Affine &l, const Affine &r)
{
    Affine o;
    float mm[3][3], tt[3];
    for (uint32_t i = 0; i < 3; ++i) {
        for (uint32_t j = 0; j < 3; ++j) mm[i][j] = dot(l.m[i], affine_col(r, j));
        tt[i] = dot(l.m[i], r.t);
    }
    for (uint32_t i = 0; i < 3; ++i) o.m[i] = {mm[i][0], mm[i][1], mm[i][2]};
    o.t = {tt[0] + l.t.x, tt[1] + l.t.y, tt[2] + l.t.z};
    return o;
}
O2V_HD V3 affine_apply(const Affine &a, V3 v)
{
    float x = dot(a.m[0], v), y = dot(a.m[1], v), z = dot(a.m[2], v);
    return {x + a.t.x, y + a.t.y, z + a.t.z};
}

// computeMeshTransform, obj2voxel.cpp:370-402
O2V_HD Affine compute_mesh_transform(V3 mesh_min, V3 mesh_max, uint32_t sample_res, const int32_t unit[9])
{
    const float kAntiBleed = 0.5f;
    V3 size = mesh_max - mesh_min;
    float max_axis = fmax2(size.x, fmax2(size.y, size.z));
    float sample_scale = (float) sample_res - kAntiBleed;

    Affine result = affine_scale(1, V3{-mesh_min.x, -mesh_min.y, -mesh_min.z});
    result = affine_mul(affine_scale(2.f / max_axis, V3{-1.f, -1.f, -1.f}), result);
    Affine u;
    for (uint32_t i = 0; i < 3; ++i) u.m[i] = {(float) unit[i * 3], (float) unit[i * 3 + 1], (float) unit[i * 3 + 2]};
    u.t = {1.f, 1.f, 1.f};
    result = affine_mul(u, result);
    result = affine_mul(affine_scale(sample_scale / 2, V3{kAntiBleed / 2, kAntiBleed / 2, kAntiBleed / 2}), result);
    return result;
}

// ---- weighted values (util.hpp:150-172) -------------------------------------------------------------------
struct WUv {
    float w, u, v;
};
struct WCol {
    float w, r, g, b;
};
O2V_HD WUv wmix(WUv l, WUv r)
{
    float ws = l.w + r.w;
    return {ws, (l.w * l.u + r.w * r.u) / ws, (l.w * l.v + r.w * r.v) / ws};
}
O2V_HD WCol wmix(WCol l, WCol r)
{
    float ws = l.w + r.w;
    return {ws, (l.w * l.r + r.w * r.r) / ws, (l.w * l.g + r.w * r.g) / ws, (l.w * l.b + r.w * r.b) / ws};
}
O2V_HD WCol wmax(WCol l, WCol r) { return l.w > r.w ? l : r; }
// combineFunction(new, existing): voxelization.cpp:56-69,520-523
O2V_HD WCol wcombine(uint32_t blend, WCol fresh, WCol existing) { return blend ? wmix(fresh, existing) : wmax(fresh, existing); }

// Color32{rgb}.argb() (obj2voxel.cpp:294-295; voxel-io conversion restated as truncation, alpha 0xFF)
O2V_HD uint32_t pack_argb(float r, float g, float b)
{
    uint32_t R = (uint8_t) (r * 255), G = (uint8_t) (g * 255), B = (uint8_t) (b * 255);
    return 0xFF000000u | (R << 16) | (G << 8) | B;
}

}  // namespace o2v
