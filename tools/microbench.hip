// microbench.hip - issue-rate microbenchmarks behind DESIGN.md §5 (standalone: hipcc --offload-arch=gfx950 -O3).
//
//  (1) clk per wave-instruction of the integer VALU ops k_detect is built from, at 1..8 waves per SIMD
//      (one workgroup per CU, 4*W waves, every wave runs an unrolled stream of independent instructions;
//      cycles from s_memtime, so the result does not depend on the DVFS clock)
//  (3) LDS window reads with the row pitch of the k_detect tile (72 vs 80 bytes)
//  (4) the box's own streaming ceiling: float4 copy / read-only pass over 2 GiB
// Prints one JSON object.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

#define ITERS 1024
#define UNROLL 16

// One asm block per 16 instructions (two passes over 8 independent accumulators).  Round 2 issued every instruction
// from its own asm statement: hipcc's hazard recognizer then pads each one with `s_nop 0` (29 s_nop per 32 VALU in the
// ISA of every k_op_* loop, tools/kernel ISA dump in DESIGN.md 5), which is what made one wave alone look like 8.5 clocks
// per instruction and `v_lshlrev_b32 v, 1, v` look half rate.  Inside one asm string nothing is inserted.
#define STREAM8(ASM)                                                                                        \
  for (int it = 0; it < ITERS; ++it) {                                                                      \
    asm volatile(ASM8(ASM) ASM8(ASM)                                                                        \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                 : "v"(x), "v"(y) : "vcc");                                                                 \
  }
// one workgroup per CU (dynamic LDS > half of the CU's), 4*W waves: W waves per SIMD, every wave runs the stream
#define ASM8(A) A
#define DEF_OP(ID, ASM)                                                                                     \
  __global__ void __launch_bounds__(1024) k_op_##ID(unsigned long long* cyc, int* sink, int seed) {         \
    extern __shared__ int dyn[];                                                                            \
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    int x = a0 ^ 0x55, y = (a0 >> 3) | 0x00010001;                                                          \
    __syncthreads();                                                                                        \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    STREAM8(ASM)                                                                                            \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;        \
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678) sink[0] = 1;                                 \
  }

DEF_OP(0, "v_and_b32 %0, %0, %8\n" "v_and_b32 %1, %1, %8\n" "v_and_b32 %2, %2, %8\n" "v_and_b32 %3, %3, %8\n" "v_and_b32 %4, %4, %8\n" "v_and_b32 %5, %5, %8\n" "v_and_b32 %6, %6, %8\n" "v_and_b32 %7, %7, %8\n")
DEF_OP(1, "v_or_b32 %0, %0, %8\n" "v_or_b32 %1, %1, %8\n" "v_or_b32 %2, %2, %8\n" "v_or_b32 %3, %3, %8\n" "v_or_b32 %4, %4, %8\n" "v_or_b32 %5, %5, %8\n" "v_or_b32 %6, %6, %8\n" "v_or_b32 %7, %7, %8\n")
DEF_OP(2, "v_add_u32 %0, %0, %8\n" "v_add_u32 %1, %1, %8\n" "v_add_u32 %2, %2, %8\n" "v_add_u32 %3, %3, %8\n" "v_add_u32 %4, %4, %8\n" "v_add_u32 %5, %5, %8\n" "v_add_u32 %6, %6, %8\n" "v_add_u32 %7, %7, %8\n")
DEF_OP(3, "v_sub_u32 %0, %0, %8\n" "v_sub_u32 %1, %1, %8\n" "v_sub_u32 %2, %2, %8\n" "v_sub_u32 %3, %3, %8\n" "v_sub_u32 %4, %4, %8\n" "v_sub_u32 %5, %5, %8\n" "v_sub_u32 %6, %6, %8\n" "v_sub_u32 %7, %7, %8\n")
DEF_OP(4, "v_lshlrev_b32 %0, 1, %0\n" "v_lshlrev_b32 %1, 1, %1\n" "v_lshlrev_b32 %2, 1, %2\n" "v_lshlrev_b32 %3, 1, %3\n" "v_lshlrev_b32 %4, 1, %4\n" "v_lshlrev_b32 %5, 1, %5\n" "v_lshlrev_b32 %6, 1, %6\n" "v_lshlrev_b32 %7, 1, %7\n")
DEF_OP(5, "v_lshrrev_b32 %0, %8, %0\n" "v_lshrrev_b32 %1, %8, %1\n" "v_lshrrev_b32 %2, %8, %2\n" "v_lshrrev_b32 %3, %8, %3\n" "v_lshrrev_b32 %4, %8, %4\n" "v_lshrrev_b32 %5, %8, %5\n" "v_lshrrev_b32 %6, %8, %6\n" "v_lshrrev_b32 %7, %8, %7\n")
DEF_OP(6, "v_max_i32 %0, %0, %8\n" "v_max_i32 %1, %1, %8\n" "v_max_i32 %2, %2, %8\n" "v_max_i32 %3, %3, %8\n" "v_max_i32 %4, %4, %8\n" "v_max_i32 %5, %5, %8\n" "v_max_i32 %6, %6, %8\n" "v_max_i32 %7, %7, %8\n")
DEF_OP(7, "v_max_u32 %0, %0, %8\n" "v_max_u32 %1, %1, %8\n" "v_max_u32 %2, %2, %8\n" "v_max_u32 %3, %3, %8\n" "v_max_u32 %4, %4, %8\n" "v_max_u32 %5, %5, %8\n" "v_max_u32 %6, %6, %8\n" "v_max_u32 %7, %7, %8\n")
DEF_OP(8, "v_min_u32 %0, %0, %8\n" "v_min_u32 %1, %1, %8\n" "v_min_u32 %2, %2, %8\n" "v_min_u32 %3, %3, %8\n" "v_min_u32 %4, %4, %8\n" "v_min_u32 %5, %5, %8\n" "v_min_u32 %6, %6, %8\n" "v_min_u32 %7, %7, %8\n")
DEF_OP(9, "v_max_f32 %0, %0, %8\n" "v_max_f32 %1, %1, %8\n" "v_max_f32 %2, %2, %8\n" "v_max_f32 %3, %3, %8\n" "v_max_f32 %4, %4, %8\n" "v_max_f32 %5, %5, %8\n" "v_max_f32 %6, %6, %8\n" "v_max_f32 %7, %7, %8\n")
DEF_OP(10, "v_add_f32 %0, %0, %8\n" "v_add_f32 %1, %1, %8\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %8\n" "v_add_f32 %4, %4, %8\n" "v_add_f32 %5, %5, %8\n" "v_add_f32 %6, %6, %8\n" "v_add_f32 %7, %7, %8\n")
DEF_OP(11, "v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_fma_f32 %3, %3, %8, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_fma_f32 %6, %6, %8, %9\n" "v_fma_f32 %7, %7, %8, %9\n")
DEF_OP(12, "v_max3_i32 %0, %0, %8, %9\n" "v_max3_i32 %1, %1, %8, %9\n" "v_max3_i32 %2, %2, %8, %9\n" "v_max3_i32 %3, %3, %8, %9\n" "v_max3_i32 %4, %4, %8, %9\n" "v_max3_i32 %5, %5, %8, %9\n" "v_max3_i32 %6, %6, %8, %9\n" "v_max3_i32 %7, %7, %8, %9\n")
DEF_OP(13, "v_max3_u32 %0, %0, %8, %9\n" "v_max3_u32 %1, %1, %8, %9\n" "v_max3_u32 %2, %2, %8, %9\n" "v_max3_u32 %3, %3, %8, %9\n" "v_max3_u32 %4, %4, %8, %9\n" "v_max3_u32 %5, %5, %8, %9\n" "v_max3_u32 %6, %6, %8, %9\n" "v_max3_u32 %7, %7, %8, %9\n")
DEF_OP(14, "v_max3_f32 %0, %0, %8, %9\n" "v_max3_f32 %1, %1, %8, %9\n" "v_max3_f32 %2, %2, %8, %9\n" "v_max3_f32 %3, %3, %8, %9\n" "v_max3_f32 %4, %4, %8, %9\n" "v_max3_f32 %5, %5, %8, %9\n" "v_max3_f32 %6, %6, %8, %9\n" "v_max3_f32 %7, %7, %8, %9\n")
DEF_OP(15, "v_min3_f32 %0, %0, %8, %9\n" "v_min3_f32 %1, %1, %8, %9\n" "v_min3_f32 %2, %2, %8, %9\n" "v_min3_f32 %3, %3, %8, %9\n" "v_min3_f32 %4, %4, %8, %9\n" "v_min3_f32 %5, %5, %8, %9\n" "v_min3_f32 %6, %6, %8, %9\n" "v_min3_f32 %7, %7, %8, %9\n")
DEF_OP(16, "v_med3_i32 %0, %0, %8, %9\n" "v_med3_i32 %1, %1, %8, %9\n" "v_med3_i32 %2, %2, %8, %9\n" "v_med3_i32 %3, %3, %8, %9\n" "v_med3_i32 %4, %4, %8, %9\n" "v_med3_i32 %5, %5, %8, %9\n" "v_med3_i32 %6, %6, %8, %9\n" "v_med3_i32 %7, %7, %8, %9\n")
DEF_OP(17, "v_bfe_u32 %0, %8, 8, 8\n" "v_bfe_u32 %1, %8, 8, 8\n" "v_bfe_u32 %2, %8, 8, 8\n" "v_bfe_u32 %3, %8, 8, 8\n" "v_bfe_u32 %4, %8, 8, 8\n" "v_bfe_u32 %5, %8, 8, 8\n" "v_bfe_u32 %6, %8, 8, 8\n" "v_bfe_u32 %7, %8, 8, 8\n")
DEF_OP(18, "v_perm_b32 %0, %8, %9, %0\n" "v_perm_b32 %1, %8, %9, %1\n" "v_perm_b32 %2, %8, %9, %2\n" "v_perm_b32 %3, %8, %9, %3\n" "v_perm_b32 %4, %8, %9, %4\n" "v_perm_b32 %5, %8, %9, %5\n" "v_perm_b32 %6, %8, %9, %6\n" "v_perm_b32 %7, %8, %9, %7\n")
DEF_OP(19, "v_alignbit_b32 %0, %8, %9, %0\n" "v_alignbit_b32 %1, %8, %9, %1\n" "v_alignbit_b32 %2, %8, %9, %2\n" "v_alignbit_b32 %3, %8, %9, %3\n" "v_alignbit_b32 %4, %8, %9, %4\n" "v_alignbit_b32 %5, %8, %9, %5\n" "v_alignbit_b32 %6, %8, %9, %6\n" "v_alignbit_b32 %7, %8, %9, %7\n")
DEF_OP(20, "v_cndmask_b32 %0, %0, %8, vcc\n" "v_cndmask_b32 %1, %1, %8, vcc\n" "v_cndmask_b32 %2, %2, %8, vcc\n" "v_cndmask_b32 %3, %3, %8, vcc\n" "v_cndmask_b32 %4, %4, %8, vcc\n" "v_cndmask_b32 %5, %5, %8, vcc\n" "v_cndmask_b32 %6, %6, %8, vcc\n" "v_cndmask_b32 %7, %7, %8, vcc\n")
DEF_OP(21, "v_add3_u32 %0, %0, %8, %9\n" "v_add3_u32 %1, %1, %8, %9\n" "v_add3_u32 %2, %2, %8, %9\n" "v_add3_u32 %3, %3, %8, %9\n" "v_add3_u32 %4, %4, %8, %9\n" "v_add3_u32 %5, %5, %8, %9\n" "v_add3_u32 %6, %6, %8, %9\n" "v_add3_u32 %7, %7, %8, %9\n")
DEF_OP(22, "v_lshl_or_b32 %0, %0, 1, %8\n" "v_lshl_or_b32 %1, %1, 1, %8\n" "v_lshl_or_b32 %2, %2, 1, %8\n" "v_lshl_or_b32 %3, %3, 1, %8\n" "v_lshl_or_b32 %4, %4, 1, %8\n" "v_lshl_or_b32 %5, %5, 1, %8\n" "v_lshl_or_b32 %6, %6, 1, %8\n" "v_lshl_or_b32 %7, %7, 1, %8\n")
DEF_OP(23, "v_and_or_b32 %0, %0, %8, %9\n" "v_and_or_b32 %1, %1, %8, %9\n" "v_and_or_b32 %2, %2, %8, %9\n" "v_and_or_b32 %3, %3, %8, %9\n" "v_and_or_b32 %4, %4, %8, %9\n" "v_and_or_b32 %5, %5, %8, %9\n" "v_and_or_b32 %6, %6, %8, %9\n" "v_and_or_b32 %7, %7, %8, %9\n")
DEF_OP(24, "v_or3_b32 %0, %0, %8, %9\n" "v_or3_b32 %1, %1, %8, %9\n" "v_or3_b32 %2, %2, %8, %9\n" "v_or3_b32 %3, %3, %8, %9\n" "v_or3_b32 %4, %4, %8, %9\n" "v_or3_b32 %5, %5, %8, %9\n" "v_or3_b32 %6, %6, %8, %9\n" "v_or3_b32 %7, %7, %8, %9\n")
DEF_OP(25, "v_sad_u8 %0, %8, %9, %0\n" "v_sad_u8 %1, %8, %9, %1\n" "v_sad_u8 %2, %8, %9, %2\n" "v_sad_u8 %3, %8, %9, %3\n" "v_sad_u8 %4, %8, %9, %4\n" "v_sad_u8 %5, %8, %9, %5\n" "v_sad_u8 %6, %8, %9, %6\n" "v_sad_u8 %7, %8, %9, %7\n")
DEF_OP(26, "v_mul_u32_u24 %0, %0, %8\n" "v_mul_u32_u24 %1, %1, %8\n" "v_mul_u32_u24 %2, %2, %8\n" "v_mul_u32_u24 %3, %3, %8\n" "v_mul_u32_u24 %4, %4, %8\n" "v_mul_u32_u24 %5, %5, %8\n" "v_mul_u32_u24 %6, %6, %8\n" "v_mul_u32_u24 %7, %7, %8\n")
DEF_OP(27, "v_mad_u32_u24 %0, %0, %8, %9\n" "v_mad_u32_u24 %1, %1, %8, %9\n" "v_mad_u32_u24 %2, %2, %8, %9\n" "v_mad_u32_u24 %3, %3, %8, %9\n" "v_mad_u32_u24 %4, %4, %8, %9\n" "v_mad_u32_u24 %5, %5, %8, %9\n" "v_mad_u32_u24 %6, %6, %8, %9\n" "v_mad_u32_u24 %7, %7, %8, %9\n")
DEF_OP(28, "v_mul_lo_u32 %0, %0, %8\n" "v_mul_lo_u32 %1, %1, %8\n" "v_mul_lo_u32 %2, %2, %8\n" "v_mul_lo_u32 %3, %3, %8\n" "v_mul_lo_u32 %4, %4, %8\n" "v_mul_lo_u32 %5, %5, %8\n" "v_mul_lo_u32 %6, %6, %8\n" "v_mul_lo_u32 %7, %7, %8\n")
DEF_OP(29, "v_cvt_f32_ubyte1 %0, %8\n" "v_cvt_f32_ubyte1 %1, %8\n" "v_cvt_f32_ubyte1 %2, %8\n" "v_cvt_f32_ubyte1 %3, %8\n" "v_cvt_f32_ubyte1 %4, %8\n" "v_cvt_f32_ubyte1 %5, %8\n" "v_cvt_f32_ubyte1 %6, %8\n" "v_cvt_f32_ubyte1 %7, %8\n")
DEF_OP(30, "v_cvt_f32_u32 %0, %0\n" "v_cvt_f32_u32 %1, %1\n" "v_cvt_f32_u32 %2, %2\n" "v_cvt_f32_u32 %3, %3\n" "v_cvt_f32_u32 %4, %4\n" "v_cvt_f32_u32 %5, %5\n" "v_cvt_f32_u32 %6, %6\n" "v_cvt_f32_u32 %7, %7\n")
DEF_OP(31, "v_bcnt_u32_b32 %0, %8, %0\n" "v_bcnt_u32_b32 %1, %8, %1\n" "v_bcnt_u32_b32 %2, %8, %2\n" "v_bcnt_u32_b32 %3, %8, %3\n" "v_bcnt_u32_b32 %4, %8, %4\n" "v_bcnt_u32_b32 %5, %8, %5\n" "v_bcnt_u32_b32 %6, %8, %6\n" "v_bcnt_u32_b32 %7, %8, %7\n")
DEF_OP(32, "v_pk_max_u16 %0, %0, %8\n" "v_pk_max_u16 %1, %1, %8\n" "v_pk_max_u16 %2, %2, %8\n" "v_pk_max_u16 %3, %3, %8\n" "v_pk_max_u16 %4, %4, %8\n" "v_pk_max_u16 %5, %5, %8\n" "v_pk_max_u16 %6, %6, %8\n" "v_pk_max_u16 %7, %7, %8\n")
DEF_OP(33, "v_pk_min_u16 %0, %0, %8\n" "v_pk_min_u16 %1, %1, %8\n" "v_pk_min_u16 %2, %2, %8\n" "v_pk_min_u16 %3, %3, %8\n" "v_pk_min_u16 %4, %4, %8\n" "v_pk_min_u16 %5, %5, %8\n" "v_pk_min_u16 %6, %6, %8\n" "v_pk_min_u16 %7, %7, %8\n")
DEF_OP(34, "v_pk_max_i16 %0, %0, %8\n" "v_pk_max_i16 %1, %1, %8\n" "v_pk_max_i16 %2, %2, %8\n" "v_pk_max_i16 %3, %3, %8\n" "v_pk_max_i16 %4, %4, %8\n" "v_pk_max_i16 %5, %5, %8\n" "v_pk_max_i16 %6, %6, %8\n" "v_pk_max_i16 %7, %7, %8\n")
DEF_OP(35, "v_pk_add_u16 %0, %0, %8\n" "v_pk_add_u16 %1, %1, %8\n" "v_pk_add_u16 %2, %2, %8\n" "v_pk_add_u16 %3, %3, %8\n" "v_pk_add_u16 %4, %4, %8\n" "v_pk_add_u16 %5, %5, %8\n" "v_pk_add_u16 %6, %6, %8\n" "v_pk_add_u16 %7, %7, %8\n")
DEF_OP(36, "v_pk_sub_i16 %0, %0, %8\n" "v_pk_sub_i16 %1, %1, %8\n" "v_pk_sub_i16 %2, %2, %8\n" "v_pk_sub_i16 %3, %3, %8\n" "v_pk_sub_i16 %4, %4, %8\n" "v_pk_sub_i16 %5, %5, %8\n" "v_pk_sub_i16 %6, %6, %8\n" "v_pk_sub_i16 %7, %7, %8\n")

typedef void (*op_kernel_t)(unsigned long long*, int*, int);
struct OpDesc { const char* name; op_kernel_t fn; int ninstr; };
static const OpDesc g_ops[] = {
    {"v_and_b32", k_op_0, 1},
    {"v_or_b32", k_op_1, 1},
    {"v_add_u32", k_op_2, 1},
    {"v_sub_u32", k_op_3, 1},
    {"v_lshlrev_b32", k_op_4, 1},
    {"v_lshrrev_b32_v", k_op_5, 1},
    {"v_max_i32", k_op_6, 1},
    {"v_max_u32", k_op_7, 1},
    {"v_min_u32", k_op_8, 1},
    {"v_max_f32", k_op_9, 1},
    {"v_add_f32", k_op_10, 1},
    {"v_fma_f32", k_op_11, 1},
    {"v_max3_i32", k_op_12, 1},
    {"v_max3_u32", k_op_13, 1},
    {"v_max3_f32", k_op_14, 1},
    {"v_min3_f32", k_op_15, 1},
    {"v_med3_i32", k_op_16, 1},
    {"v_bfe_u32", k_op_17, 1},
    {"v_perm_b32", k_op_18, 1},
    {"v_alignbit_b32", k_op_19, 1},
    {"v_cndmask_b32", k_op_20, 1},
    {"v_add3_u32", k_op_21, 1},
    {"v_lshl_or_b32", k_op_22, 1},
    {"v_and_or_b32", k_op_23, 1},
    {"v_or3_b32", k_op_24, 1},
    {"v_sad_u8", k_op_25, 1},
    {"v_mul_u32_u24", k_op_26, 1},
    {"v_mad_u32_u24", k_op_27, 1},
    {"v_mul_lo_u32", k_op_28, 1},
    {"v_cvt_f32_ubyte1", k_op_29, 1},
    {"v_cvt_f32_u32", k_op_30, 1},
    {"v_bcnt_u32_b32", k_op_31, 1},
    {"v_pk_max_u16", k_op_32, 1},
    {"v_pk_min_u16", k_op_33, 1},
    {"v_pk_max_i16", k_op_34, 1},
    {"v_pk_add_u16", k_op_35, 1},
    {"v_pk_sub_i16", k_op_36, 1},
};

// LDS window reads as in k_detect phase A: thread (cg = t & 15, rg = t >> 4) reads 3 dwords of 10 rows
template <int PITCH>
__global__ void __launch_bounds__(256) k_lds(unsigned long long* cyc, int* sink) {
  __shared__ unsigned tile[70 * PITCH / 4];
  for (int i = threadIdx.x; i < 70 * PITCH / 4; i += 256) tile[i] = i * 2654435761u;
  __syncthreads();
  const int lx = (threadIdx.x & 15), ly = (threadIdx.x >> 4) * 4;
  unsigned s = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const volatile unsigned* p = &tile[(ly + r) * (PITCH / 4) + lx];
      s += p[0] + p[1] + p[2];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (s == 0x12345678u) sink[0] = 1;
}

__global__ void __launch_bounds__(256) k_copy4(const float4* __restrict__ in, float4* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_read4(const float4* __restrict__ in, float* __restrict__ out, long n) {
  float s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float4 v = in[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 1.2345f) out[0] = s;
}


struct IssueResult { double clk_memtime; double ns_wall; };
static IssueResult run_issue(const OpDesc& op, int waves_per_simd, unsigned long long* d_cyc, int* d_sink, int ncu) {
  const int threads = 64 * 4 * waves_per_simd;
  const size_t lds = 100 * 1024;
  CHECK(hipFuncSetAttribute((const void*)op.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  std::vector<unsigned long long> h(ncu * threads / 64);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  IssueResult best{1e30, 1e30};
  const double n = (double)ITERS * UNROLL * op.ninstr;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(op.fn, dim3(ncu), dim3(threads), lds, 0, d_cyc, d_sink, rep);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    if (rep) {
      best.clk_memtime = std::min(best.clk_memtime, (double)h[h.size() / 2] / n);
      best.ns_wall = std::min(best.ns_wall, (double)ms * 1e6 / n);
    }
  }
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  unsigned long long* d_cyc;
  int* d_sink;
  CHECK(hipMalloc(&d_cyc, 8 * 1024 * 64));
  CHECK(hipMalloc(&d_sink, 64));
  printf("{\"device\": \"%s\", \"cus\": %d, \"note\": \"per op and waves/SIMD: [s_memtime ticks per instruction of one wave, SIMD ticks per instruction, wall ns per instruction per SIMD (includes launch ~ few us)]\", \"issue\": {", prop.gcnArchName, ncu);
  const int W[] = {1, 2, 4, 8};
  const int nops = (int)(sizeof(g_ops) / sizeof(g_ops[0]));
  for (int op = 0; op < nops; ++op) {
    printf("%s\"%s\": {", op ? ", " : "", g_ops[op].name);
    for (int wi = 0; wi < 4; ++wi) {
      if (W[wi] * 256 > 1024 && false) continue;
      const int w = W[wi] > 4 ? 4 : W[wi];
      if (W[wi] > 4) break;
      const IssueResult r = run_issue(g_ops[op], w, d_cyc, d_sink, ncu);
      printf("%s\"%dw\": [%.3f, %.3f, %.3f]", wi ? ", " : "", w, r.clk_memtime, r.clk_memtime / w, r.ns_wall / w);
    }
    printf("}");
  }
  printf("}, \"lds_window_reads_clk_per_wave_30_reads\": {");
  {
    std::vector<unsigned long long> h(ncu * 4 * 4);
    for (int pi = 0; pi < 2; ++pi) {
      for (int occ = 1; occ <= 4; occ *= 2) {
        if (pi == 0) hipLaunchKernelGGL(k_lds<72>, dim3(ncu * occ), dim3(256), 0, 0, d_cyc, d_sink);
        else hipLaunchKernelGGL(k_lds<80>, dim3(ncu * occ), dim3(256), 0, 0, d_cyc, d_sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d_cyc, (size_t)ncu * occ * 4 * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.begin() + ncu * occ * 4);
        printf("%s\"pitch%d_wg_per_cu%d\": %.1f", (pi || occ > 1) ? ", " : "", pi ? 80 : 72, occ, (double)h[ncu * occ * 2] / 256.0);
      }
    }
  }
  printf("}, \"stream_GBps\": {");
  {
    const long bytes = 2L << 30, n = bytes / 16;
    float4 *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(k_copy4, dim3(ncu * 16), dim3(256), 0, 0, a, b, n);
        else hipLaunchKernelGGL(k_read4, dim3(ncu * 16), dim3(256), 0, 0, a, (float*)b, n);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, ms);
      }
      printf("%s\"%s\": %.1f", mode ? ", " : "", mode ? "float4_read_2GiB" : "float4_copy_2GiB_read_plus_write",
             (mode ? 1.0 : 2.0) * bytes / (best * 1e-3) / 1e9);
    }
  }
  printf("}}\n");
  return 0;
}
