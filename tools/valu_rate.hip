// valu_rate.hip — issue cost of the vector instructions the step kernels are made of, measured on gfx950 (MI355X).
//
// "settle the fp64 question with evidence".  One workgroup on one CU with 4 / 8 / 16 waves = 1 / 2 / 4 waves
// per SIMD; every wave runs ITER iterations of a block of 32 instructions of ONE class, either independent (8 accumulators,
// issue rate) or dependent (one accumulator, latency), between two s_memtime reads.  Reported: shader clocks per
// wave-instruction as one wave sees them (its own stream) and per SIMD (clocks / instructions of all waves of that SIMD),
// i.e. what the pipe sustains.  Build + run: tools/valu_rate.py.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <algorithm>

#define ITER 512

#define REP4(x) x x x x

// 8 independent accumulators, 4 rounds = 32 instructions per iteration
#define INDEP64(op)                                                                                       \
  asm volatile(REP4(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
                    op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n") \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)            \
               : "v"(b), "v"(c))
#define DEP64(op)                                                                                         \
  asm volatile(REP4(op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" \
                    op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n" op " %0, %0, %8, %9\n") \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)            \
               : "v"(b), "v"(c))
// two-operand forms (dst, src0, src1)
#define INDEP2(op)                                                                                        \
  asm volatile(REP4(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n"           \
                    op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n")           \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)            \
               : "v"(b), "v"(c))
#define DEP2(op)                                                                                          \
  asm volatile(REP4(op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n"           \
                    op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n")           \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)            \
               : "v"(b), "v"(c))
// one-operand forms (dst, src)
#define INDEP1(op)                                                                                        \
  asm volatile(REP4(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n"                           \
                    op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n")                           \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)            \
               : "v"(b), "v"(c))
#define DEP1(op)                                                                                          \
  asm volatile(REP4(op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n"                           \
                    op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n")                           \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)            \
               : "v"(b), "v"(c))

#define KERNEL(name, T, INIT, BODY)                                                         \
  __global__ __launch_bounds__(1024) void name(unsigned long long *out, T *sink) {        \
    T a0 = (T)(INIT + threadIdx.x), a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3;        \
    T a4 = a0 + (T)4, a5 = a0 + (T)5, a6 = a0 + (T)6, a7 = a0 + (T)7;                       \
    T b = (T)1, c = (T)0;                                                                   \
    asm volatile("" : "+v"(b), "+v"(c));                                                    \
    __syncthreads();                                                                        \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                             \
    _Pragma("unroll 1") for (int i = 0; i < ITER; ++i) { BODY; }                            \
    asm volatile("s_nop 0" ::: "memory");                                                   \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                             \
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                           \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == (T)-12345) sink[0] = a0;                   \
  }

KERNEL(k_fma_f64_i, double, 1.0, INDEP64("v_fma_f64"))
KERNEL(k_fma_f64_d, double, 1.0, DEP64("v_fma_f64"))
KERNEL(k_mul_f64_i, double, 1.0, INDEP2("v_mul_f64"))
KERNEL(k_mul_f64_d, double, 1.0, DEP2("v_mul_f64"))
KERNEL(k_add_f64_i, double, 1.0, INDEP2("v_add_f64"))
KERNEL(k_add_f64_d, double, 1.0, DEP2("v_add_f64"))
KERNEL(k_max_f64_i, double, 1.0, INDEP2("v_max_f64"))
KERNEL(k_floor_f64_i, double, 1.0, INDEP1("v_floor_f64"))
KERNEL(k_rcp_f64_i, double, 1.0, INDEP1("v_rcp_f64"))
KERNEL(k_sqrt_f64_i, double, 1.0, INDEP1("v_sqrt_f64"))
KERNEL(k_fma_f32_i, float, 1.0f, INDEP64("v_fma_f32"))
KERNEL(k_fma_f32_d, float, 1.0f, DEP64("v_fma_f32"))
KERNEL(k_mul_f32_i, float, 1.0f, INDEP2("v_mul_f32"))
KERNEL(k_add_f32_i, float, 1.0f, INDEP2("v_add_f32"))
KERNEL(k_pk_fma_f32_i, double, 1.0, INDEP64("v_pk_fma_f32"))
KERNEL(k_mul_u24_i, unsigned int, 3u, INDEP2("v_mul_u32_u24"))
KERNEL(k_mul_u24_d, unsigned int, 3u, DEP2("v_mul_u32_u24"))
KERNEL(k_mul_lo_u32_i, unsigned int, 3u, INDEP2("v_mul_lo_u32"))
KERNEL(k_add_u32_i, unsigned int, 3u, INDEP2("v_add_u32"))
KERNEL(k_add_u32_d, unsigned int, 3u, DEP2("v_add_u32"))
KERNEL(k_lshl_b32_i, unsigned int, 3u, INDEP2("v_lshlrev_b32"))
KERNEL(k_and_b32_i, unsigned int, 3u, INDEP2("v_and_b32"))
KERNEL(k_mad_u32_u24_i, unsigned int, 3u, INDEP64("v_mad_u32_u24"))
KERNEL(k_or_b32_i, unsigned int, 3u, INDEP2("v_or_b32"))
KERNEL(k_xor_b32_i, unsigned int, 3u, INDEP2("v_xor_b32"))
KERNEL(k_lshr_b32_i, unsigned int, 3u, INDEP2("v_lshrrev_b32"))
KERNEL(k_ashr_i32_i, unsigned int, 3u, INDEP2("v_ashrrev_i32"))
KERNEL(k_min_u32_i, unsigned int, 3u, INDEP2("v_min_u32"))
KERNEL(k_max_i32_i, unsigned int, 3u, INDEP2("v_max_i32"))
KERNEL(k_sub_u32_i, unsigned int, 3u, INDEP2("v_sub_u32"))
KERNEL(k_mov_b32_i, unsigned int, 3u, INDEP1("v_mov_b32"))
KERNEL(k_add3_u32_i, unsigned int, 3u, INDEP64("v_add3_u32"))
KERNEL(k_lshl_add_u32_i, unsigned int, 3u, INDEP64("v_lshl_add_u32"))
KERNEL(k_and_or_b32_i, unsigned int, 3u, INDEP64("v_and_or_b32"))
KERNEL(k_bfe_u32_i, unsigned int, 3u, INDEP64("v_bfe_u32"))
KERNEL(k_cvt_f32_i32_i, unsigned int, 3u, INDEP1("v_cvt_f32_i32"))
KERNEL(k_cvt_i32_f32_i, float, 1.0f, INDEP1("v_cvt_i32_f32"))
KERNEL(k_floor_f32_i, float, 1.0f, INDEP1("v_floor_f32"))
KERNEL(k_rcp_f32_i, float, 1.0f, INDEP1("v_rcp_f32"))
KERNEL(k_mov_b64_i, double, 1.0, INDEP1("v_mov_b64"))

// compares write a lane mask (vcc): 8 compares per group on 8 register pairs
__global__ __launch_bounds__(1024) void k_cmp_f64_i(unsigned long long *out, double *sink) {
  double a0 = 1.0 + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 5.0;
  asm volatile("" : "+v"(b));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i)
    asm volatile(REP4("v_cmp_le_f64 vcc, %0, %4\n v_cmp_le_f64 vcc, %1, %4\n v_cmp_le_f64 vcc, %2, %4\n v_cmp_le_f64 vcc, %3, %4\n"
                      "v_cmp_le_f64 vcc, %0, %4\n v_cmp_le_f64 vcc, %1, %4\n v_cmp_le_f64 vcc, %2, %4\n v_cmp_le_f64 vcc, %3, %4\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (a0 + a1 + a2 + a3 == -12345.0) sink[0] = a0;
}
__global__ __launch_bounds__(1024) void k_cmp_f32_i(unsigned long long *out, float *sink) {
  float a0 = 1.0f + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 5.0f;
  asm volatile("" : "+v"(b));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i)
    asm volatile(REP4("v_cmp_le_f32 vcc, %0, %4\n v_cmp_le_f32 vcc, %1, %4\n v_cmp_le_f32 vcc, %2, %4\n v_cmp_le_f32 vcc, %3, %4\n"
                      "v_cmp_le_f32 vcc, %0, %4\n v_cmp_le_f32 vcc, %1, %4\n v_cmp_le_f32 vcc, %2, %4\n v_cmp_le_f32 vcc, %3, %4\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (a0 + a1 + a2 + a3 == -12345.0f) sink[0] = a0;
}
// select by a lane mask (vcc stays what the prologue left in it)
__global__ __launch_bounds__(1024) void k_cndmask_i(unsigned long long *out, unsigned int *sink) {
  unsigned int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 5;
  asm volatile("" : "+v"(b));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i)
    asm volatile(REP4("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                      "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (a0 + a1 + a2 + a3 == 12345u) sink[0] = a0;
}
__global__ __launch_bounds__(1024) void k_cmp_u32_i(unsigned long long *out, unsigned int *sink) {
  unsigned int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 5;
  asm volatile("" : "+v"(b));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4\n"
                      "v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (a0 + a1 + a2 + a3 == 12345u) sink[0] = a0;
}
// LDS: a dependent chain of reads (latency) and independent reads (rate)
__global__ __launch_bounds__(1024) void k_ds_read_b32_d(unsigned long long *out, unsigned int *sink) {
  __shared__ unsigned int buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = (unsigned int)(((i * 4) + 256) & 16383);
  __syncthreads();
  unsigned int a = (threadIdx.x * 4) & 16383;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i) {
    unsigned int v;
    asm volatile(REP4("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, %0\n"
                      "ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, %0\n"
                      "ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, %0\n"
                      "ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, %0\n")
                 : "=&v"(v), "+v"(a));
  }
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (a == 12345u) sink[0] = a;
}
// conversions f64 <-> i32 (the march's cell index): dst and src differ in width
__global__ __launch_bounds__(1024) void k_cvt_i32_f64_i(unsigned long long *out, double *sink) {
  double a0 = 1.0 + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i)
    asm volatile(REP4("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7\n"
                      "v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7\n")
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (r0 + r1 + r2 + r3 == -12345) sink[0] = a0;
}
__global__ __launch_bounds__(1024) void k_cvt_f32_f64_i(unsigned long long *out, double *sink) {
  double a0 = 1.0 + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i)
    asm volatile(REP4("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                      "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n")
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (r0 + r1 + r2 + r3 == -12345.0f) sink[0] = a0;
}
// scalar instructions share the wave's issue slot with the vector ones: their own cost
__global__ __launch_bounds__(1024) void k_salu_i(unsigned long long *out, int *sink) {
  int s0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i)
    asm volatile(REP4("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n"
                      "s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n")
                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (s0 + s1 + s2 + s3 == -12345) sink[0] = s0;
}
// a vector and a scalar instruction alternating: do they issue in the same cycle group or one after the other?
__global__ __launch_bounds__(1024) void k_valu_salu_mix(unsigned long long *out, int *sink) {
  int s0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), s1 = s0 + 1;
  unsigned int a0 = threadIdx.x, a1 = a0 + 1, b = 3;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i)
    asm volatile(REP4("v_add_u32 %0, %0, %4\n s_add_u32 %2, %2, 3\n v_add_u32 %1, %1, %4\n s_add_u32 %3, %3, 3\n"
                      "v_add_u32 %0, %0, %4\n s_add_u32 %2, %2, 3\n v_add_u32 %1, %1, %4\n s_add_u32 %3, %3, 3\n")
                 : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : "v"(b) : "scc");
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if ((int)(a0 + a1) + s0 + s1 == -12345) sink[0] = s0;
}
// f64 and f32 alternating (a mixed-precision march): 16 + 16 per group
__global__ __launch_bounds__(1024) void k_f64_f32_mix(unsigned long long *out, double *sink) {
  double a0 = 1.0 + threadIdx.x, a1 = a0 + 1, b = 1.0, c = 0.0;
  float f0 = 1.0f + threadIdx.x, f1 = f0 + 1, g = 1.0f, h = 0.0f;
  asm volatile("" : "+v"(b), "+v"(c), "+v"(g), "+v"(h));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i)
    asm volatile(REP4("v_fma_f64 %0, %0, %4, %5\n v_fma_f32 %2, %2, %6, %7\n v_fma_f64 %1, %1, %4, %5\n v_fma_f32 %3, %3, %6, %7\n"
                      "v_fma_f64 %0, %0, %4, %5\n v_fma_f32 %2, %2, %6, %7\n v_fma_f64 %1, %1, %4, %5\n v_fma_f32 %3, %3, %6, %7\n")
                 : "+v"(a0), "+v"(a1), "+v"(f0), "+v"(f1) : "v"(b), "v"(c), "v"(g), "v"(h));
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (a0 + a1 + f0 + f1 == -12345.0) sink[0] = a0;
}

struct Case {
  const char *name;
  void (*launch)(int waves, unsigned long long *out, void *sink);
};
#define CASE(k, T) {#k, [](int waves, unsigned long long *out, void *sink) { hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, (T *)sink); }}

int main(int argc, char **argv) {
  const bool json = argc > 1 && !strcmp(argv[1], "--json");
  std::vector<Case> cases = {
      CASE(k_fma_f64_i, double), CASE(k_fma_f64_d, double), CASE(k_mul_f64_i, double), CASE(k_mul_f64_d, double),
      CASE(k_add_f64_i, double), CASE(k_add_f64_d, double), CASE(k_max_f64_i, double), CASE(k_floor_f64_i, double),
      CASE(k_rcp_f64_i, double), CASE(k_sqrt_f64_i, double), CASE(k_cmp_f64_i, double), CASE(k_cvt_i32_f64_i, double),
      CASE(k_cvt_f32_f64_i, double), CASE(k_fma_f32_i, float), CASE(k_fma_f32_d, float), CASE(k_mul_f32_i, float),
      CASE(k_add_f32_i, float), CASE(k_pk_fma_f32_i, double), CASE(k_cmp_f32_i, float), CASE(k_mul_u24_i, unsigned int),
      CASE(k_mul_u24_d, unsigned int), CASE(k_mul_lo_u32_i, unsigned int), CASE(k_mad_u32_u24_i, unsigned int),
      CASE(k_add_u32_i, unsigned int), CASE(k_add_u32_d, unsigned int), CASE(k_lshl_b32_i, unsigned int),
      CASE(k_and_b32_i, unsigned int), CASE(k_or_b32_i, unsigned int), CASE(k_xor_b32_i, unsigned int), CASE(k_lshr_b32_i, unsigned int),
      CASE(k_ashr_i32_i, unsigned int), CASE(k_min_u32_i, unsigned int), CASE(k_max_i32_i, unsigned int), CASE(k_sub_u32_i, unsigned int),
      CASE(k_mov_b32_i, unsigned int), CASE(k_add3_u32_i, unsigned int), CASE(k_lshl_add_u32_i, unsigned int), CASE(k_and_or_b32_i, unsigned int),
      CASE(k_bfe_u32_i, unsigned int), CASE(k_cvt_f32_i32_i, unsigned int), CASE(k_cvt_i32_f32_i, float), CASE(k_floor_f32_i, float),
      CASE(k_rcp_f32_i, float), CASE(k_mov_b64_i, double), CASE(k_cndmask_i, unsigned int), CASE(k_cmp_u32_i, unsigned int),
      CASE(k_ds_read_b32_d, unsigned int), CASE(k_salu_i, int),
      CASE(k_valu_salu_mix, int), CASE(k_f64_f32_mix, double)};
  unsigned long long *out;
  void *sink;
  hipMalloc(&out, 16 * sizeof(unsigned long long));
  hipMalloc(&sink, 64);
  const int ninst = ITER * 32;
  if (json) printf("{\"iter\": %d, \"insts_per_wave\": %d, \"cases\": [\n", ITER, ninst);
  else printf("%-20s %8s | %-28s | %-28s\n", "kernel", "", "clocks / wave-instruction (one wave's own stream)", "clocks / instruction per SIMD");
  bool first = true;
  for (const Case &cs : cases) {
    double own[3], simd[3];
    int wi = 0;
    for (int waves : {4, 8, 16}) {
      unsigned long long h[16];
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {  // the slowest wave of the fastest repetition
        cs.launch(waves, out, sink);
        hipDeviceSynchronize();
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int w = 0; w < waves; ++w) mx = std::max(mx, h[w]);
        best = std::min(best, (double)mx);
      }
      own[wi] = best / ninst;
      simd[wi] = best / (ninst * (waves / 4.0));
      ++wi;
    }
    if (json) {
      printf("%s  {\"kernel\": \"%s\", \"clocks_per_inst_own\": {\"1\": %.3f, \"2\": %.3f, \"4\": %.3f}, \"clocks_per_inst_simd\": {\"1\": %.3f, \"2\": %.3f, \"4\": %.3f}}",
             first ? "" : ",\n", cs.name, own[0], own[1], own[2], simd[0], simd[1], simd[2]);
      first = false;
    } else {
      printf("%-20s waves/SIMD 1,2,4 | %8.2f %8.2f %8.2f | %8.2f %8.2f %8.2f\n", cs.name, own[0], own[1], own[2], simd[0], simd[1], simd[2]);
    }
  }
  if (json) printf("\n]}\n");
  if (hipGetLastError() != hipSuccess) {
    fprintf(stderr, "HIP error\n");
    return 1;
  }
  return 0;
}
