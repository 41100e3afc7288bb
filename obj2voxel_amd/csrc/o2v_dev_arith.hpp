// o2v_dev_arith.hpp -- correctly rounded float32 division in fewer instructions (k_voxelize's clip loop).
//
// Part of the device code of o2v_device.hip, which includes this file inside its anonymous namespace.  Not a stand-alone
// header.
//
// The reference divides in float32 (C++ operator/: IEEE 754, round to nearest even) where a triangle is cut by a voxel
// plane (the cut parameter, voxelization.cpp:262-266,305-318) and in the running uv mean of the pieces (textureCenter() / 3
// and mix's division by the weight sum, triangle.hpp:129-132, util.hpp:160-165).  hipcc expands `a / b` to v_div_scale x 2,
// v_rcp_f32, seven v_fma, v_div_fmas and v_div_fixup: twelve instructions, one of them quarter rate.  The clip loop of
// k_voxelize<true> ran ten to eleven such divisions per iteration - a quarter of its vector instructions.
// The forms below give the same bits for the operands they are used on:
//
//   third(x)       x / 3 for EVERY float32 x (zeros, subnormals, infinities and NaNs included): the product with the double
//                  nearest to 1/3, rounded back.  x / 3 is never closer to a rounding boundary of float32 than a sixth of an
//                  ulp (3 j = 2 m + 3 has no solution in odd-sized steps: a boundary is a half-integer multiple of the ulp),
//                  while the double product is off by 2^-52 relative, so both roundings pick the same float.  Checked for all
//                  2^32 inputs on the device (k_check_third, tests/test_gpu_arith.py) and for a sample of every exponent on
//                  the host (tests/test_host_arith.py).  Three instructions.
//   div_lean(n, d) the compiler's own sequence without v_div_scale / v_div_fmas' rescaling / v_div_fixup.  Those only act when
//                  an operand or the quotient is zero, subnormal, infinite, NaN or within 2^+-~100 of the format's limits; for
//                  operands in the middle of the range they pass their inputs through, so what remains is the same
//                  arithmetic and the same result.  The ranges it is used on (see lean_ok in o2v_dev_k2_voxelize.hpp) lie
//                  far inside the box of exponent pairs for which k_check_div finds no difference on the device
//                  (tests/test_gpu_arith.py maps all 256 x 256 pairs of biased exponents).
//   LeanRecip      the refined reciprocal of div_lean, shared by the two divisions of the uv mean (same divisor).
//
// Exact mode (O2V_HIP_FLAG_EXACT_CLIP) never takes the lean forms of the divisions by a variable: every leaf is then `not
// small`, so the fast-vs-exact tests compare them with the compiler's division on whole workloads.

// ---- selects with an explicit lane mask --------------------------------------------------------------------------------
// hipcc turns `c ? x : y` into v_cndmask_b32 and shrinks it to the two-operand encoding (VOP2, mask in VCC) whenever it can.
// On gfx950 a VOP2 v_cndmask that directly follows another one takes 16 .. 23 cycles of its SIMD instead of 4 (measured:
// tools/ubench/valu_rates.hip, profiles/r03/valu_rates.json: fifteen in a row after one compare 16.3 cycles each; the VOP3
// encoding with the mask in an SGPR pair - or in VCC - 4.2; alternating encodings 4.1) - and a piece of the clip loop is
// fifteen floats that are selected together (sel_piece), which the compiler emitted as runs of fifteen.  These helpers pin
// the three-operand encoding: the condition becomes a lane mask in an SGPR pair (lane_mask: the compare writes it
// directly), the select names it.  Values only; same result as the conditional expression.  (On the whole kernel the
// effect is small - configs[3] 14.4 -> 14.3 ms: its wavefronts wait on each other's instruction latencies, see DESIGN.md.)
__device__ __forceinline__ unsigned long long lane_mask(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ float vsel(unsigned long long take_x, float x, float y)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(y), "v"(x), "s"(take_x));
    return r;
}
__device__ __forceinline__ uint32_t vsel(unsigned long long take_x, uint32_t x, uint32_t y)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(y), "v"(x), "s"(take_x));
    return r;
}
__device__ __forceinline__ V3 vsel(unsigned long long take_x, V3 x, V3 y) { return V3{vsel(take_x, x.x, y.x), vsel(take_x, x.y, y.y), vsel(take_x, x.z, y.z)}; }
__device__ __forceinline__ V2 vsel(unsigned long long take_x, V2 x, V2 y) { return V2{vsel(take_x, x.x, y.x), vsel(take_x, x.y, y.y)}; }

__device__ __forceinline__ float third(float x) { return (float) ((double) x * (1.0 / 3.0)); }

struct LeanRecip {
    float d, r;
    __device__ __forceinline__ explicit LeanRecip(float divisor) : d(divisor)
    {
        const float r0 = __builtin_amdgcn_rcpf(divisor);
        const float e = __builtin_fmaf(-divisor, r0, 1.0f);
        r = __builtin_fmaf(e, r0, r0);
    }
    __device__ __forceinline__ float divide(float n) const
    {
        float q = n * r;
        float e = __builtin_fmaf(-d, q, n);
        q = __builtin_fmaf(e, r, q);
        e = __builtin_fmaf(-d, q, n);
        return __builtin_fmaf(e, r, q);
    }
};
__device__ __forceinline__ float div_lean(float n, float d) { return LeanRecip(d).divide(n); }

// ---- device self checks (debug entry points o2v_hip_debug_check_third / _check_div) ---------------------------------------
// (the reference forms are kept from being folded with the forms under test by the volatile round trip)
__global__ __launch_bounds__(256) void k_check_third(unsigned long long *out /*[2]: mismatches, first bad input + 1*/)
{
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256u + threadIdx.x; i < (1ull << 32); i += (uint64_t) gridDim.x * 256u) {
        const float x = __uint_as_float((uint32_t) i);
        volatile float three = 3.0f;
        const float want = x / three, got = third(x);
        const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
        if (!same) {
            bad += 1;
            atomicMin(&out[1], i + 1);
        }
    }
    if (bad) atomicAdd(&out[0], bad);
}

// out[en * 256 + ed]: among `samples` pairs with biased exponents en (numerator) and ed (divisor), random signs and
// mantissas (the all-zero and all-one mantissas included), how many quotients of div_lean differ from n / d
__global__ __launch_bounds__(256) void k_check_div(uint32_t *out, uint32_t samples, uint64_t seed)
{
    const uint32_t en = blockIdx.x >> 8, ed = blockIdx.x & 255u;
    uint32_t bad = 0;
    for (uint32_t s = threadIdx.x; s < samples; s += 256u) {
        uint64_t h = seed + ((uint64_t) blockIdx.x << 32) + s;
        h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
        uint32_t mn = (uint32_t) h & 0x7fffffu, md = (uint32_t) (h >> 23) & 0x7fffffu;
        if (s < 4u) mn = (s & 1u) ? 0x7fffffu : 0u, md = (s & 2u) ? 0x7fffffu : 0u;
        const uint32_t sn = (uint32_t) (h >> 46) & 1u, sd = (uint32_t) (h >> 47) & 1u;
        const float n = __uint_as_float((sn << 31) | (en << 23) | mn), d = __uint_as_float((sd << 31) | (ed << 23) | md);
        volatile float dv = d;
        const float want = n / dv, got = div_lean(n, d);
        if (__float_as_uint(want) != __float_as_uint(got) && !(want != want && got != got)) bad += 1;
    }
    if (bad) atomicAdd(&out[blockIdx.x], bad);
}
