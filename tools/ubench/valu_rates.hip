// Developer tool: issue cost of the instructions k_voxelize's clip loop is made of, on the GPU it runs on.
// Every kernel runs a loop of 64 copies of one instruction (dependent: each reads the previous result; independent: eight
// rotating destinations) on W wavefronts per SIMD; reported is SIMD cycles per instruction = time x clock / (instructions of
// one wavefront x W), i.e. 1 / throughput per SIMD, at an assumed 2.4 GHz.
//   make -C tools/ubench && tools/ubench/_build/valu_rates        (inline asm clobbers scc: the loop counter lives in SGPRs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// dependent chain: op dst=v0 <- f(v0, ...); independent: 8 destinations v0..v7 each own chain
#define KERNEL_DEP(NAME, ASM)                                                                                  \
    __global__ void NAME(float *out, int iters)                                                                \
    {                                                                                                          \
        float a = out[threadIdx.x & 1] + 1.5f, b = 1.0001f, c = 0.5f;                                           \
        double da = a, db = 1.0001;                                                                            \
        for (int i = 0; i < iters; ++i) { asm volatile(REP64(ASM) : "+v"(a), "+v"(da) : "v"(b), "v"(c), "v"(db) : "vcc", "scc", "s20", "s21"); }   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + (float) da;                                            \
    }
#define KERNEL_IND(NAME, A0, A1, A2, A3)                                                                        \
    __global__ void NAME(float *out, int iters)                                                                \
    {                                                                                                          \
        float a0 = out[threadIdx.x & 1] + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f, c = 0.5f;    \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = 1.0001;                                                 \
        for (int i = 0; i < iters; ++i) {                                                                      \
            asm volatile(REP8(REP8(A0 A1 A2 A3) ) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c), "v"(db) : "vcc", "scc", "s20", "s21", "s22", "s23"); \
        }                                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float) (d0 + d1 + d2 + d3);            \
    }

// %0 = a (float), %1 = da (double), %2 = b, %3 = c, %4 = db
KERNEL_DEP(dep_mul_f32, "v_mul_f32 %0, %0, %2\n")
KERNEL_DEP(dep_fma_f32, "v_fma_f32 %0, %0, %2, %3\n")
KERNEL_DEP(dep_cndmask, "v_cndmask_b32 %0, %0, %2, vcc\n")
KERNEL_DEP(dep_rcp_f32, "v_rcp_f32 %0, %0\n")
KERNEL_DEP(dep_mul_f64, "v_mul_f64 %1, %1, %4\n")
KERNEL_DEP(dep_fma_f64, "v_fma_f64 %1, %1, %4, %4\n")
KERNEL_DEP(dep_cvt_f64_f32_and_back, "v_cvt_f64_f32 %1, %0\nv_cvt_f32_f64 %0, %1\n")
KERNEL_DEP(dep_div_fixup, "v_div_fixup_f32 %0, %0, %2, %3\n")
KERNEL_DEP(dep_div_scale, "v_div_scale_f32 %0, vcc, %0, %2, %0\n")
KERNEL_DEP(dep_div_fmas, "v_div_fmas_f32 %0, %0, %2, %3\n")
KERNEL_DEP(dep_min3, "v_min3_f32 %0, %0, %2, %3\n")
KERNEL_DEP(dep_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %2\nv_cndmask_b32 %0, %0, %3, vcc\n")
KERNEL_DEP(dep_cmp_sgpr_cnd, "v_cmp_lt_f32 s[20:21], %0, %2\nv_cndmask_b32 %0, %0, %3, s[20:21]\n")
KERNEL_DEP(dep_readlane_writelane, "v_readlane_b32 s20, %0, 3\nv_writelane_b32 %0, s20, 5\n")

// independent: %0..%3 floats, %4..%7 doubles, %8 = b, %9 = c, %10 = db
KERNEL_IND(ind_mul_f32, "v_mul_f32 %0, %0, %8\n", "v_mul_f32 %1, %1, %8\n", "v_mul_f32 %2, %2, %8\n", "v_mul_f32 %3, %3, %8\n")
KERNEL_IND(ind_fma_f32, "v_fma_f32 %0, %0, %8, %9\n", "v_fma_f32 %1, %1, %8, %9\n", "v_fma_f32 %2, %2, %8, %9\n", "v_fma_f32 %3, %3, %8, %9\n")
KERNEL_IND(ind_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n", "v_cndmask_b32 %1, %1, %8, vcc\n", "v_cndmask_b32 %2, %2, %8, vcc\n", "v_cndmask_b32 %3, %3, %8, vcc\n")
KERNEL_IND(ind_cndmask_e64, "v_cndmask_b32 %0, %0, %8, s[20:21]\n", "v_cndmask_b32 %1, %1, %8, s[20:21]\n", "v_cndmask_b32 %2, %2, %8, s[22:23]\n", "v_cndmask_b32 %3, %3, %8, s[22:23]\n")
KERNEL_IND(ind_rcp_f32, "v_rcp_f32 %0, %0\n", "v_rcp_f32 %1, %1\n", "v_rcp_f32 %2, %2\n", "v_rcp_f32 %3, %3\n")
KERNEL_IND(ind_mul_f64, "v_mul_f64 %4, %4, %10\n", "v_mul_f64 %5, %5, %10\n", "v_mul_f64 %6, %6, %10\n", "v_mul_f64 %7, %7, %10\n")
KERNEL_IND(ind_cvt_f64_f32, "v_cvt_f64_f32 %4, %0\n", "v_cvt_f64_f32 %5, %1\n", "v_cvt_f64_f32 %6, %2\n", "v_cvt_f64_f32 %7, %3\n")
KERNEL_IND(ind_cvt_f32_f64, "v_cvt_f32_f64 %0, %4\n", "v_cvt_f32_f64 %1, %5\n", "v_cvt_f32_f64 %2, %6\n", "v_cvt_f32_f64 %3, %7\n")
KERNEL_IND(ind_div_scale, "v_div_scale_f32 %0, vcc, %0, %8, %0\n", "v_div_scale_f32 %1, vcc, %1, %8, %1\n", "v_div_scale_f32 %2, vcc, %2, %8, %2\n", "v_div_scale_f32 %3, vcc, %3, %8, %3\n")
KERNEL_IND(ind_div_fixup, "v_div_fixup_f32 %0, %0, %8, %9\n", "v_div_fixup_f32 %1, %1, %8, %9\n", "v_div_fixup_f32 %2, %2, %8, %9\n", "v_div_fixup_f32 %3, %3, %8, %9\n")
KERNEL_IND(ind_min3, "v_min3_f32 %0, %0, %8, %9\n", "v_min3_f32 %1, %1, %8, %9\n", "v_min3_f32 %2, %2, %8, %9\n", "v_min3_f32 %3, %3, %8, %9\n")
KERNEL_IND(ind_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %8\n", "v_cmp_lt_f32 vcc, %1, %8\n", "v_cmp_lt_f32 vcc, %2, %8\n", "v_cmp_lt_f32 vcc, %3, %8\n")
KERNEL_IND(ind_cmp_sgpr, "v_cmp_lt_f32 s[20:21], %0, %8\n", "v_cmp_lt_f32 s[22:23], %1, %8\n", "v_cmp_lt_f32 s[20:21], %2, %8\n", "v_cmp_lt_f32 s[22:23], %3, %8\n")
KERNEL_IND(ind_pk_mul_f32, "v_pk_mul_f32 %4, %4, %10\n", "v_pk_mul_f32 %5, %5, %10\n", "v_pk_mul_f32 %6, %6, %10\n", "v_pk_mul_f32 %7, %7, %10\n")
KERNEL_IND(ind_pk_fma_f32, "v_pk_fma_f32 %4, %4, %10, %10\n", "v_pk_fma_f32 %5, %5, %10, %10\n", "v_pk_fma_f32 %6, %6, %10, %10\n", "v_pk_fma_f32 %7, %7, %10, %10\n")
KERNEL_IND(ind_salu, "s_and_b64 s[20:21], s[20:21], s[22:23]\n", "s_or_b64 s[22:23], s[20:21], s[22:23]\n", "s_and_b64 s[20:21], s[20:21], s[22:23]\n", "s_or_b64 s[22:23], s[20:21], s[22:23]\n")
KERNEL_IND(ind_valu_salu_mix, "v_mul_f32 %0, %0, %8\n", "s_and_b64 s[20:21], s[20:21], s[22:23]\n", "v_mul_f32 %2, %2, %8\n", "s_or_b64 s[22:23], s[20:21], s[22:23]\n")
KERNEL_IND(ind_alignbit, "v_alignbit_b32 %0, %0, %8, 31\n", "v_alignbit_b32 %1, %1, %8, 31\n", "v_alignbit_b32 %2, %2, %8, 31\n", "v_alignbit_b32 %3, %3, %8, 31\n")
KERNEL_IND(ind_mov, "v_mov_b32 %0, %8\n", "v_mov_b32 %1, %8\n", "v_mov_b32 %2, %8\n", "v_mov_b32 %3, %8\n")
KERNEL_IND(ind_readlane, "v_readlane_b32 s20, %0, 3\n", "v_readlane_b32 s21, %1, 3\n", "v_readlane_b32 s22, %2, 3\n", "v_readlane_b32 s23, %3, 3\n")
KERNEL_IND(ind_dpp_mov, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n", "v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n", "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n", "v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")

// vcc written once by a compare, then read by N selects
KERNEL_IND(cmp_then_cnd3, "v_cmp_lt_f32 vcc, %0, %8\n", "v_cndmask_b32 %1, %1, %8, vcc\n", "v_cndmask_b32 %2, %2, %8, vcc\n", "v_cndmask_b32 %3, %3, %8, vcc\n")
__global__ void cmp_then_cnd15(float *out, int iters)
{
    float a0 = out[threadIdx.x & 1] + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %4\n" "v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                          "v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                          "v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                          "v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                          "v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
// vcc written by the scalar unit, then read by selects
KERNEL_IND(salu_vcc_then_cnd3, "s_and_b64 vcc, s[20:21], s[22:23]\n", "v_cndmask_b32 %1, %1, %8, vcc\n", "v_cndmask_b32 %2, %2, %8, vcc\n", "v_cndmask_b32 %3, %3, %8, vcc\n")
KERNEL_IND(ind_add_f32, "v_add_f32 %0, %0, %8\n", "v_add_f32 %1, %1, %8\n", "v_add_f32 %2, %2, %8\n", "v_add_f32 %3, %3, %8\n")
KERNEL_IND(ind_sub_f32_2src, "v_sub_f32 %0, %1, %2\n", "v_sub_f32 %1, %2, %3\n", "v_sub_f32 %2, %3, %0\n", "v_sub_f32 %3, %0, %1\n")
KERNEL_IND(ind_fmac_f32, "v_fmac_f32 %0, %8, %9\n", "v_fmac_f32 %1, %8, %9\n", "v_fmac_f32 %2, %8, %9\n", "v_fmac_f32 %3, %8, %9\n")
KERNEL_IND(ind_max_f32, "v_max_f32 %0, %0, %8\n", "v_max_f32 %1, %1, %8\n", "v_max_f32 %2, %2, %8\n", "v_max_f32 %3, %3, %8\n")
KERNEL_IND(ind_and_b32, "v_and_b32 %0, %0, %8\n", "v_and_b32 %1, %1, %8\n", "v_and_b32 %2, %2, %8\n", "v_and_b32 %3, %3, %8\n")
KERNEL_IND(ind_add_u32, "v_add_u32 %0, %0, %8\n", "v_add_u32 %1, %1, %8\n", "v_add_u32 %2, %2, %8\n", "v_add_u32 %3, %3, %8\n")
KERNEL_IND(ind_lshl_or, "v_lshl_or_b32 %0, %0, 1, %8\n", "v_lshl_or_b32 %1, %1, 1, %8\n", "v_lshl_or_b32 %2, %2, 1, %8\n", "v_lshl_or_b32 %3, %3, 1, %8\n")
KERNEL_IND(ind_saveexec, "s_and_saveexec_b64 s[20:21], vcc\n", "v_mul_f32 %1, %1, %8\n", "s_or_b64 exec, exec, s[20:21]\n", "v_mul_f32 %3, %3, %8\n")

// opcodes of the clip loop that round 3 priced by a relative (tools/isa_hist.py: PRICE_KEYS)
KERNEL_IND(ind_or_b32, "v_or_b32 %0, %0, %8\n", "v_or_b32 %1, %1, %8\n", "v_or_b32 %2, %2, %8\n", "v_or_b32 %3, %3, %8\n")
KERNEL_IND(ind_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0\n", "v_lshlrev_b32 %1, 3, %1\n", "v_lshlrev_b32 %2, 3, %2\n", "v_lshlrev_b32 %3, 3, %3\n")
KERNEL_IND(ind_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0\n", "v_lshrrev_b32 %1, 3, %1\n", "v_lshrrev_b32 %2, 3, %2\n", "v_lshrrev_b32 %3, 3, %3\n")
KERNEL_IND(ind_or3_b32, "v_or3_b32 %0, %0, %8, %9\n", "v_or3_b32 %1, %1, %8, %9\n", "v_or3_b32 %2, %2, %8, %9\n", "v_or3_b32 %3, %3, %8, %9\n")
KERNEL_IND(ind_and_or_b32, "v_and_or_b32 %0, %0, %8, %9\n", "v_and_or_b32 %1, %1, %8, %9\n", "v_and_or_b32 %2, %2, %8, %9\n", "v_and_or_b32 %3, %3, %8, %9\n")
KERNEL_IND(ind_bfe_u32, "v_bfe_u32 %0, %0, 3, 6\n", "v_bfe_u32 %1, %1, 3, 6\n", "v_bfe_u32 %2, %2, 3, 6\n", "v_bfe_u32 %3, %3, 3, 6\n")
KERNEL_IND(ind_max3_f32, "v_max3_f32 %0, %0, %8, %9\n", "v_max3_f32 %1, %1, %8, %9\n", "v_max3_f32 %2, %2, %8, %9\n", "v_max3_f32 %3, %3, %8, %9\n")
KERNEL_IND(ind_cvt_f32_u32, "v_cvt_f32_u32 %0, %0\n", "v_cvt_f32_u32 %1, %1\n", "v_cvt_f32_u32 %2, %2\n", "v_cvt_f32_u32 %3, %3\n")
KERNEL_IND(ind_mul_u32_u24, "v_mul_u32_u24 %0, %0, %8\n", "v_mul_u32_u24 %1, %1, %8\n", "v_mul_u32_u24 %2, %2, %8\n", "v_mul_u32_u24 %3, %3, %8\n")
KERNEL_IND(ind_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0\n", "v_mbcnt_hi_u32_b32 %1, -1, %1\n", "v_mbcnt_lo_u32_b32 %2, -1, %2\n", "v_mbcnt_hi_u32_b32 %3, -1, %3\n")
KERNEL_IND(ind_lshl_add_u64, "v_lshl_add_u64 %4, %4, 3, %5\n", "v_lshl_add_u64 %5, %5, 3, %6\n", "v_lshl_add_u64 %6, %6, 3, %7\n", "v_lshl_add_u64 %7, %7, 3, %4\n")
KERNEL_IND(ind_cmp_u32, "v_cmp_eq_u32 s[20:21], %0, %8\n", "v_cmp_eq_u32 s[22:23], %1, %8\n", "v_cmp_eq_u32 s[20:21], %2, %8\n", "v_cmp_eq_u32 s[22:23], %3, %8\n")

#define KERNEL_SEQ(NAME, BODY)                                                                                 \
    __global__ void NAME(float *out, int iters)                                                                \
    {                                                                                                          \
        float a0 = out[threadIdx.x & 1] + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f;              \
        float m0 = a0 + 4, m1 = a0 + 5, m2 = a0 + 6, m3 = a0 + 7;                                               \
        for (int i = 0; i < iters; ++i) {                                                                      \
            asm volatile(REP8(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(b) : "vcc", "scc", "s20", "s21", "s22", "s23"); \
        }                                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + m0 + m1 + m2 + m3;                     \
    }
#define MUL7 "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n"
#define MAX7 "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n"
// 16 instructions per body
KERNEL_SEQ(seq_cmp_max7_cndvcc_max7, "v_cmp_lt_f32 vcc, %0, %8\n" MAX7 "v_cndmask_b32 %1, %1, %8, vcc\n" MAX7)
KERNEL_SEQ(seq_cmp_max7_cnde64_max7, "v_cmp_lt_f32 s[20:21], %0, %8\n" MAX7 "v_cndmask_b32 %1, %1, %8, s[20:21]\n" MAX7)
KERNEL_SEQ(seq_max16, "v_max_f32 %7, %7, %8\n" MAX7 "v_max_f32 %7, %7, %8\n" MAX7)
KERNEL_SEQ(seq_cmp_cnd15_e64vcc, "v_cmp_lt_f32 vcc, %0, %8\n"
           "v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n"
           "v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n"
           "v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n")
KERNEL_SEQ(seq_cmp_cnd15_e32vcc, "v_cmp_lt_f32 vcc, %0, %8\n"
           "v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n"
           "v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n"
           "v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n")
KERNEL_SEQ(seq_cmp_cnd15_e64sgpr, "v_cmp_lt_f32 s[20:21], %0, %8\n"
           "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n"
           "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n"
           "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n")
// two alternating masks (as the rotation of split_cut uses them)
KERNEL_SEQ(seq_cnd_two_sgpr_masks, "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_gt_f32 s[22:23], %0, %8\n"
           "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[22:23]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[22:23]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[22:23]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n"
           "v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]\n")
KERNEL_SEQ(seq_cnd_vcc_and_sgpr_masks, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_gt_f32 s[22:23], %0, %8\n"
           "v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, s[22:23]\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, s[22:23]\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, s[22:23]\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
           "v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]\n")

// ---- operand banks -------------------------------------------------------------------------------------------------
// The loops above let the compiler pick the registers.  These name them: a VGPR's bank is its number modulo 4, and an
// instruction whose source operands sit in one bank could need extra read cycles.  If the 4-cycle class (v_fma, v_max, compares,
// selects) were an artefact of such conflicts in the loops above, the "distinct banks" form would run at the 2-cycle rate of
// v_mul / v_add; if it is the hardware's rate for these opcodes, both forms read the same.
#define BANK_REGS "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35"
#define KERNEL_BANK(NAME, A0, A1, A2, A3)                                                                        \
    __global__ void NAME(float *out, int iters)                                                                \
    {                                                                                                          \
        float r = out[threadIdx.x & 1] + 1.5f;                                                                  \
        asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 1.0\n v_mov_b32 v24, %0\n v_mov_b32 v25, 1.0\n" \
                     "v_mov_b32 v26, 0.5\n v_mov_b32 v27, 1.0\n v_mov_b32 v28, %0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 0.5\n v_mov_b32 v31, 1.0\n"  \
                     "v_mov_b32 v32, %0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 0.5\n v_mov_b32 v35, 1.0\n" : : "v"(r) : BANK_REGS);               \
        for (int i = 0; i < iters; ++i) { asm volatile(REP8(REP8(A0 A1 A2 A3)) : : : BANK_REGS, "vcc", "s20", "s21", "s22", "s23"); }               \
        asm volatile("v_add_f32 %0, v20, v24\n v_add_f32 %0, %0, v28\n v_add_f32 %0, %0, v32\n" : "=v"(r) : : BANK_REGS);                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                        \
    }
// (destination = first source, as in the loops above; the other sources: banks 1 and 2 / the destination's own bank 0)
KERNEL_BANK(bank_fma_distinct, "v_fma_f32 v20, v20, v21, v22\n", "v_fma_f32 v24, v24, v25, v26\n", "v_fma_f32 v28, v28, v29, v30\n", "v_fma_f32 v32, v32, v33, v34\n")
KERNEL_BANK(bank_fma_same, "v_fma_f32 v20, v20, v24, v28\n", "v_fma_f32 v24, v24, v28, v32\n", "v_fma_f32 v28, v28, v32, v20\n", "v_fma_f32 v32, v32, v20, v24\n")
KERNEL_BANK(bank_max_distinct, "v_max_f32 v20, v20, v21\n", "v_max_f32 v24, v24, v25\n", "v_max_f32 v28, v28, v29\n", "v_max_f32 v32, v32, v33\n")
KERNEL_BANK(bank_max_same, "v_max_f32 v20, v20, v24\n", "v_max_f32 v24, v24, v28\n", "v_max_f32 v28, v28, v32\n", "v_max_f32 v32, v32, v20\n")
KERNEL_BANK(bank_mul_distinct, "v_mul_f32 v20, v20, v21\n", "v_mul_f32 v24, v24, v25\n", "v_mul_f32 v28, v28, v29\n", "v_mul_f32 v32, v32, v33\n")
KERNEL_BANK(bank_mul_same, "v_mul_f32 v20, v20, v24\n", "v_mul_f32 v24, v24, v28\n", "v_mul_f32 v28, v28, v32\n", "v_mul_f32 v32, v32, v20\n")
KERNEL_BANK(bank_cmp_distinct, "v_cmp_lt_f32 s[20:21], v20, v21\n", "v_cmp_lt_f32 s[22:23], v24, v25\n", "v_cmp_lt_f32 s[20:21], v28, v29\n", "v_cmp_lt_f32 s[22:23], v32, v33\n")
KERNEL_BANK(bank_cndmask_distinct, "v_cndmask_b32_e64 v20, v20, v21, s[20:21]\n", "v_cndmask_b32_e64 v24, v24, v25, s[22:23]\n", "v_cndmask_b32_e64 v28, v28, v29, s[20:21]\n", "v_cndmask_b32_e64 v32, v32, v33, s[22:23]\n")
KERNEL_BANK(bank_pk_fma_distinct, "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[26:27]\n", "v_pk_fma_f32 v[24:25], v[24:25], v[22:23], v[26:27]\n", "v_pk_fma_f32 v[28:29], v[28:29], v[30:31], v[34:35]\n", "v_pk_fma_f32 v[32:33], v[32:33], v[30:31], v[34:35]\n")

struct Case { const char *name; void (*fn)(float *, int); int per_rep; };
#define C1(n) {#n, n, 64}
#define C2(n) {#n, n, 128}

int main()
{
    const Case cases[] = {{"seq_cmp_max7_cndvcc_max7", seq_cmp_max7_cndvcc_max7, 128}, {"seq_cmp_max7_cnde64_max7", seq_cmp_max7_cnde64_max7, 128}, {"seq_max16", seq_max16, 128},
                          {"seq_cmp_cnd15_e64vcc", seq_cmp_cnd15_e64vcc, 128}, {"seq_cmp_cnd15_e32vcc", seq_cmp_cnd15_e32vcc, 128}, {"seq_cmp_cnd15_e64sgpr", seq_cmp_cnd15_e64sgpr, 128},
                          {"seq_cnd_two_sgpr_masks", seq_cnd_two_sgpr_masks, 128}, {"seq_cnd_vcc_and_sgpr_masks", seq_cnd_vcc_and_sgpr_masks, 128},{"cmp_then_cnd3", cmp_then_cnd3, 256}, {"cmp_then_cnd15", cmp_then_cnd15, 128}, {"salu_vcc_then_cnd3", salu_vcc_then_cnd3, 256},
                          {"ind_add_f32", ind_add_f32, 256}, {"ind_sub_f32_2src", ind_sub_f32_2src, 256}, {"ind_fmac_f32", ind_fmac_f32, 256}, {"ind_max_f32", ind_max_f32, 256},
                          {"ind_and_b32", ind_and_b32, 256}, {"ind_add_u32", ind_add_u32, 256}, {"ind_lshl_or", ind_lshl_or, 256}, {"ind_saveexec", ind_saveexec, 256},
                          {"ind_salu", ind_salu, 256}, {"ind_valu_salu_mix", ind_valu_salu_mix, 256}, {"ind_alignbit", ind_alignbit, 256},
                          {"ind_mov", ind_mov, 256}, {"ind_readlane", ind_readlane, 256}, {"ind_dpp_mov", ind_dpp_mov, 256},C1(dep_mul_f32), C1(dep_fma_f32), C1(dep_cndmask), C1(dep_rcp_f32), C1(dep_mul_f64), C1(dep_fma_f64),
                          C2(dep_cvt_f64_f32_and_back), C1(dep_div_fixup), C1(dep_div_scale), C1(dep_div_fmas), C1(dep_min3), C2(dep_cmp_cnd),
                          C2(dep_cmp_sgpr_cnd), C2(dep_readlane_writelane),
                          {"ind_mul_f32", ind_mul_f32, 256}, {"ind_fma_f32", ind_fma_f32, 256}, {"ind_cndmask", ind_cndmask, 256},
                          {"ind_cndmask_e64", ind_cndmask_e64, 256}, {"ind_rcp_f32", ind_rcp_f32, 256}, {"ind_mul_f64", ind_mul_f64, 256},
                          {"ind_cvt_f64_f32", ind_cvt_f64_f32, 256}, {"ind_cvt_f32_f64", ind_cvt_f32_f64, 256}, {"ind_div_scale", ind_div_scale, 256},
                          {"ind_div_fixup", ind_div_fixup, 256}, {"ind_min3", ind_min3, 256}, {"ind_cmp_vcc", ind_cmp_vcc, 256},
                          {"ind_cmp_sgpr", ind_cmp_sgpr, 256}, {"ind_pk_mul_f32", ind_pk_mul_f32, 256}, {"ind_pk_fma_f32", ind_pk_fma_f32, 256},
                          {"ind_salu", ind_salu, 256}, {"ind_valu_salu_mix", ind_valu_salu_mix, 256}, {"ind_alignbit", ind_alignbit, 256},
                          {"ind_mov", ind_mov, 256}, {"ind_readlane", ind_readlane, 256}, {"ind_dpp_mov", ind_dpp_mov, 256},
                          {"ind_or_b32", ind_or_b32, 256}, {"ind_lshlrev_b32", ind_lshlrev_b32, 256}, {"ind_lshrrev_b32", ind_lshrrev_b32, 256},
                          {"ind_or3_b32", ind_or3_b32, 256}, {"ind_and_or_b32", ind_and_or_b32, 256}, {"ind_bfe_u32", ind_bfe_u32, 256},
                          {"ind_max3_f32", ind_max3_f32, 256}, {"ind_cvt_f32_u32", ind_cvt_f32_u32, 256}, {"ind_mul_u32_u24", ind_mul_u32_u24, 256},
                          {"ind_mbcnt", ind_mbcnt, 256}, {"ind_lshl_add_u64", ind_lshl_add_u64, 256}, {"ind_cmp_u32", ind_cmp_u32, 256},
                          {"bank_fma_distinct", bank_fma_distinct, 256}, {"bank_fma_same", bank_fma_same, 256}, {"bank_max_distinct", bank_max_distinct, 256},
                          {"bank_max_same", bank_max_same, 256}, {"bank_mul_distinct", bank_mul_distinct, 256}, {"bank_mul_same", bank_mul_same, 256},
                          {"bank_cmp_distinct", bank_cmp_distinct, 256}, {"bank_cndmask_distinct", bank_cndmask_distinct, 256},
                          {"bank_pk_fma_distinct", bank_pk_fma_distinct, 256}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = 2.4;
    float *out;
    hipMalloc(&out, (size_t) cus * 4 * 16 * 64 * sizeof(float));
    hipMemset(out, 0, (size_t) cus * 4 * 16 * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 500;
    printf("{\"device\": \"%s\", \"cus\": %d, \"assumed_ghz\": %.1f, \"unit\": \"SIMD cycles per instruction\", \"results\": {\n", prop.gcnArchName, cus, ghz);
    bool first = true;
    for (const Case &c : cases) {
        printf("%s  \"%s\": {", first ? "" : ",\n", c.name);
        first = false;
        bool f2 = true;
        for (int w : {1, 4, 8}) {
            // 4 * w wavefronts per CU, w per SIMD: one workgroup per CU, two of 16 wavefronts each for w = 8 (all kernels use few
            // registers, so eight wavefronts fit a SIMD)
            const dim3 grid(w == 8 ? 2 * cus : cus), block(w == 8 ? 1024 : 64 * 4 * w);
            hipLaunchKernelGGL(c.fn, grid, block, 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(c.fn, grid, block, 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double cyc = ms * 1e-3 * ghz * 1e9 / ((double) iters * c.per_rep * w);
            printf("%s\"w%d\": %.2f", f2 ? "" : ", ", w, cyc);
            f2 = false;
        }
        printf("}"); fflush(stdout);
    }
    printf("\n}}\n");
    return 0;
}
