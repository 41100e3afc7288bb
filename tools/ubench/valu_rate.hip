// Microbenchmark: wave64 VALU issue rate on gfx950 for the instruction mix k_voxelize is made of
// (v_add_f32 / v_mul_f32 / v_cndmask_b32 / IEEE division), at 1, 2, 4 and 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + (float) (threadIdx.x + i);
    float b = seed * 0.5f + 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = a[i] * b + 1.0f;            // v_mul + v_add (no contraction)
                else if (MODE == 1) a[i] = a[i] > b ? a[(i + 1) & 7] : a[i] + b;  // v_cmp + v_cndmask + v_add
                else if (MODE == 2) a[i] = a[i] / (b + (float) i);  // IEEE division
                else a[i] = (a[i] + b) + a[(i + 7) & 7];            // dependent-ish chain of adds
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int instr_per_inner, float *d_out, int cus)
{
    for (int blocks_per_cu : {1, 2, 4, 8}) {  // 256 threads = 4 waves per block -> 1, 2, 4, 8 waves per SIMD
        const int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d_out, 10, 1.0f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d_out, iters, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr_per_simd = (double) blocks_per_cu * iters * 16 * 8 * instr_per_inner;
        std::printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name,
                    blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
    }
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *d_out;
    hipMalloc(&d_out, (size_t) cus * 8 * 256 * sizeof(float));
    run<0>("v_mul_f32 + v_add_f32", 2, d_out, cus);
    run<1>("v_cmp + v_cndmask + v_add", 3, d_out, cus);
    run<3>("v_add + v_add (chained)", 2, d_out, cus);
    run<2>("IEEE f32 division (as 1)", 1, d_out, cus);
    return 0;
}
