// valu_rate.hip -- issue cost of the VALU / LDS instructions the fused kernel is made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/hwprobe/valu_rate.hip -o tools/hwprobe/valu_rate && tools/hwprobe/valu_rate
// For each instruction: a wave runs ITER x 64 copies on 8 independent register chains; 1024 x W waves (W per SIMD).
// Reported: shader cycles per instruction per SIMD (s_memtime of wave 0 / instructions issued by the SIMD's W waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *cyc, int iters) {
    __shared__ uint32_t lds[4096];
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 9, a5 = a0 + 11, a6 = a0 ^ 13, a7 = a0 | 64;
    uint32_t b = blockIdx.x + 1, c = threadIdx.x & 3;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    uint32_t la = (threadIdx.x * 4) & 4095;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 1) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 2) { REP8(asm volatile("v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n v_mad_i32_i24 %4, %4, %8, %9\n v_mad_i32_i24 %5, %5, %8, %9\n v_mad_i32_i24 %6, %6, %8, %9\n v_mad_i32_i24 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 3) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 4) { REP8(asm volatile("v_alignbyte_b32 %0, %0, %8, %9\n v_alignbyte_b32 %1, %1, %8, %9\n v_alignbyte_b32 %2, %2, %8, %9\n v_alignbyte_b32 %3, %3, %8, %9\n v_alignbyte_b32 %4, %4, %8, %9\n v_alignbyte_b32 %5, %5, %8, %9\n v_alignbyte_b32 %6, %6, %8, %9\n v_alignbyte_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 5) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 6) { REP8(asm volatile("v_cmp_eq_u32 s[20:21], %0, %8\n v_cmp_eq_u32 s[22:23], %1, %8\n v_cmp_eq_u32 s[24:25], %2, %8\n v_cmp_eq_u32 s[26:27], %3, %8\n v_cmp_eq_u32 s[20:21], %4, %8\n v_cmp_eq_u32 s[22:23], %5, %8\n v_cmp_eq_u32 s[24:25], %6, %8\n v_cmp_eq_u32 s[26:27], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 7) { REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n v_readlane_b32 s24, %4, 11\n v_readlane_b32 s25, %5, 13\n v_readlane_b32 s26, %6, 15\n v_readlane_b32 s27, %7, 17" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 8) { REP8(asm volatile("v_writelane_b32 %0, s4, 3\n v_writelane_b32 %1, s4, 5\n v_writelane_b32 %2, s4, 7\n v_writelane_b32 %3, s4, 9\n v_writelane_b32 %4, s4, 11\n v_writelane_b32 %5, s4, 13\n v_writelane_b32 %6, s4, 15\n v_writelane_b32 %7, s4, 17" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 9) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 10) { REP8(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 11) { REP8(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 12) { REP8(asm volatile("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %5, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 13) { REP8(asm volatile("v_lshl_or_b32 %0, %0, %8, %9\n v_lshl_or_b32 %1, %1, %8, %9\n v_lshl_or_b32 %2, %2, %8, %9\n v_lshl_or_b32 %3, %3, %8, %9\n v_lshl_or_b32 %4, %4, %8, %9\n v_lshl_or_b32 %5, %5, %8, %9\n v_lshl_or_b32 %6, %6, %8, %9\n v_lshl_or_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(b));) }
        if (OP == 14) { REP8(asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1" : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");) }
        if (OP == 15) { REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(la));) }
        if (OP == 16) { REP8(asm volatile("ds_write_b8 %8, %0\n ds_write_b8 %8, %1 offset:1\n ds_write_b8 %8, %2 offset:2\n ds_write_b8 %8, %3 offset:3\n ds_write_b8 %8, %4 offset:256\n ds_write_b8 %8, %5 offset:257\n ds_write_b8 %8, %6 offset:258\n ds_write_b8 %8, %7 offset:259\n s_waitcnt lgkmcnt(0)" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(la));) }
        if (OP == 17) { REP8(asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:1\n ds_read2_b32 %1, %4 offset0:64 offset1:65\n ds_read2_b32 %2, %4 offset0:128 offset1:129\n ds_read2_b32 %3, %4 offset0:192 offset1:193\n s_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)&a0), "=v"(*(uint64_t*)&a2), "=v"(*(uint64_t*)&a4), "=v"(*(uint64_t*)&a6) : "v"(la));) }
        if (OP == 18) { REP8(asm volatile("ds_write_b32 %8, %0\n ds_write_b32 %8, %1 offset:256\n ds_write_b32 %8, %2 offset:512\n ds_write_b32 %8, %3 offset:768\n ds_write_b32 %8, %4 offset:1024\n ds_write_b32 %8, %5 offset:1280\n ds_write_b32 %8, %6 offset:1536\n ds_write_b32 %8, %7 offset:1792\n s_waitcnt lgkmcnt(0)" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(la));) }
        if (OP == 19) { asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21"); REP8(asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21");) }
        if (OP == 20) {  REP8(asm volatile("v_bfi_b32 %0, %9, %0, %8\n v_bfi_b32 %1, %9, %1, %8\n v_bfi_b32 %2, %9, %2, %8\n v_bfi_b32 %3, %9, %3, %8\n v_bfi_b32 %4, %9, %4, %8\n v_bfi_b32 %5, %9, %5, %8\n v_bfi_b32 %6, %9, %6, %8\n v_bfi_b32 %7, %9, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 21) {  REP8(asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 22) {  REP8(asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 23) {  REP8(asm volatile("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 24) {  REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 25) {  REP8(asm volatile("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n v_bfe_u32 %4, %4, 3, 8\n v_bfe_u32 %5, %5, 3, 8\n v_bfe_u32 %6, %6, 3, 8\n v_bfe_u32 %7, %7, 3, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 26) {  REP8(asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 27) {  REP8(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 1, %6\n v_lshrrev_b32 %7, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 28) {  REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 29) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 30) {  REP8(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 31) {  REP8(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 32) {  REP8(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n v_add_u32 %4, 0x12345, %4\n v_add_u32 %5, 0x12345, %5\n v_add_u32 %6, 0x12345, %6\n v_add_u32 %7, 0x12345, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 33) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 34) {  REP8(asm volatile("v_cmp_eq_u32_e32 vcc, %0, %8\n v_cmp_eq_u32_e32 vcc, %1, %8\n v_cmp_eq_u32_e32 vcc, %2, %8\n v_cmp_eq_u32_e32 vcc, %3, %8\n v_cmp_eq_u32_e32 vcc, %4, %8\n v_cmp_eq_u32_e32 vcc, %5, %8\n v_cmp_eq_u32_e32 vcc, %6, %8\n v_cmp_eq_u32_e32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 35) {  REP8(asm volatile("v_cmp_eq_u32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cmp_eq_u32_e32 vcc, %1, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cmp_eq_u32_e32 vcc, %2, %8\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cmp_eq_u32_e32 vcc, %3, %8\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cmp_eq_u32_e32 vcc, %4, %8\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cmp_eq_u32_e32 vcc, %5, %8\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_eq_u32_e32 vcc, %6, %8\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cmp_eq_u32_e32 vcc, %7, %8\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 36) {  REP8(asm volatile("v_cmp_eq_u32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cmp_eq_u32_e64 s[20:21], %1, %8\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cmp_eq_u32_e64 s[20:21], %2, %8\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cmp_eq_u32_e64 s[20:21], %3, %8\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cmp_eq_u32_e64 s[20:21], %4, %8\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cmp_eq_u32_e64 s[20:21], %5, %8\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cmp_eq_u32_e64 s[20:21], %6, %8\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cmp_eq_u32_e64 s[20:21], %7, %8\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21");) }
        if (OP == 37) {  REP8(asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %8, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %8, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %8, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %8, vcc\n v_addc_co_u32_e32 %4, vcc, %4, %8, vcc\n v_addc_co_u32_e32 %5, vcc, %5, %8, vcc\n v_addc_co_u32_e32 %6, vcc, %6, %8, vcc\n v_addc_co_u32_e32 %7, vcc, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 38) {  REP8(asm volatile("v_add_co_u32_e32 %0, vcc, %0, %8\n v_add_co_u32_e32 %1, vcc, %1, %8\n v_add_co_u32_e32 %2, vcc, %2, %8\n v_add_co_u32_e32 %3, vcc, %3, %8\n v_add_co_u32_e32 %4, vcc, %4, %8\n v_add_co_u32_e32 %5, vcc, %5, %8\n v_add_co_u32_e32 %6, vcc, %6, %8\n v_add_co_u32_e32 %7, vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 39) {  REP8(asm volatile("v_cmp_eq_u32_e32 vcc, %0, %8\n v_add_u32 %9, %9, %8\n v_add_u32 %9, %9, %8\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cmp_eq_u32_e32 vcc, %1, %8\n v_add_u32 %9, %9, %8\n v_add_u32 %9, %9, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cmp_eq_u32_e32 vcc, %2, %8\n v_add_u32 %9, %9, %8\n v_add_u32 %9, %9, %8\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cmp_eq_u32_e32 vcc, %3, %8\n v_add_u32 %9, %9, %8\n v_add_u32 %9, %9, %8\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cmp_eq_u32_e32 vcc, %4, %8\n v_add_u32 %9, %9, %8\n v_add_u32 %9, %9, %8\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cmp_eq_u32_e32 vcc, %5, %8\n v_add_u32 %9, %9, %8\n v_add_u32 %9, %9, %8\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_eq_u32_e32 vcc, %6, %8\n v_add_u32 %9, %9, %8\n v_add_u32 %9, %9, %8\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cmp_eq_u32_e32 vcc, %7, %8\n v_add_u32 %9, %9, %8\n v_add_u32 %9, %9, %8\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) }
        if (OP == 40) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n s_nop 4\n v_cndmask_b32_e32 %1, %1, %8, vcc\n s_nop 4\n v_cndmask_b32_e32 %2, %2, %8, vcc\n s_nop 4\n v_cndmask_b32_e32 %3, %3, %8, vcc\n s_nop 4\n v_cndmask_b32_e32 %4, %4, %8, vcc\n s_nop 4\n v_cndmask_b32_e32 %5, %5, %8, vcc\n s_nop 4\n v_cndmask_b32_e32 %6, %6, %8, vcc\n s_nop 4\n v_cndmask_b32_e32 %7, %7, %8, vcc\n s_nop 4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
        if (OP == 41) {  REP8(asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 42) {  REP8(asm volatile("v_dot2_i32_i16 %0, %0, %8, %9\n v_dot2_i32_i16 %1, %1, %8, %9\n v_dot2_i32_i16 %2, %2, %8, %9\n v_dot2_i32_i16 %3, %3, %8, %9\n v_dot2_i32_i16 %4, %4, %8, %9\n v_dot2_i32_i16 %5, %5, %8, %9\n v_dot2_i32_i16 %6, %6, %8, %9\n v_dot2_i32_i16 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 43) {  REP8(asm volatile("v_mad_i32_i16 %0, %0, %8, %9\n v_mad_i32_i16 %1, %1, %8, %9\n v_mad_i32_i16 %2, %2, %8, %9\n v_mad_i32_i16 %3, %3, %8, %9\n v_mad_i32_i16 %4, %4, %8, %9\n v_mad_i32_i16 %5, %5, %8, %9\n v_mad_i32_i16 %6, %6, %8, %9\n v_mad_i32_i16 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 44) {  REP8(asm volatile("v_med3_i32 %0, %0, %8, %9\n v_med3_i32 %1, %1, %8, %9\n v_med3_i32 %2, %2, %8, %9\n v_med3_i32 %3, %3, %8, %9\n v_med3_i32 %4, %4, %8, %9\n v_med3_i32 %5, %5, %8, %9\n v_med3_i32 %6, %6, %8, %9\n v_med3_i32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 45) {  REP8(asm volatile("v_pk_min_i16 %0, %0, %8\n v_pk_min_i16 %1, %1, %8\n v_pk_min_i16 %2, %2, %8\n v_pk_min_i16 %3, %3, %8\n v_pk_min_i16 %4, %4, %8\n v_pk_min_i16 %5, %5, %8\n v_pk_min_i16 %6, %6, %8\n v_pk_min_i16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int OP>
void run(const char *name, int per_group) {
    uint32_t *out; unsigned long long *cyc, h, hh[2];
    hipMalloc(&out, 4 * 256 * 2048 * 4); hipMalloc(&cyc, 16);
    const int iters = 64;
    printf("%-22s", name);
    for (int W : {1, 2, 4, 8}) {             // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD; W blocks per CU
        k<OP><<<256 * W, 256>>>(out, cyc, iters);          // warm
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<OP><<<256 * W, 256>>>(out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost); h = hh[0];
        const double n = (double)iters * 8 * per_group;   // instructions per wave
        printf("  W=%d: wave0 %5.2f cyc/inst @%4.2f GHz, kernel %6.1f us |", W, (double)h / n, (double)h / ((double)hh[1] * 10.0), ms * 1e3);
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main(int argc, char **argv) {
    if (argc > 1) {
    run<33>("v_cndmask e64 (vcc op)", 8);
    run<34>("v_cmp_eq_u32_e32 ->vcc", 8);
    run<35>("cmp_e32+cndmask_e32 vcc", 16);
    run<36>("cmp_e64+cndmask_e64 sgpr", 16);
    run<37>("v_addc_co_u32_e32", 8);
    run<38>("v_add_co_u32_e32", 8);
    run<39>("cmp_e32 + 2 add + cndmask_e32", 32);
    run<40>("v_cndmask_e32 + s_nop 4 between", 8);
        return 0;
    }
    run<0>("v_add_u32", 8); run<1>("v_and_b32", 8); run<2>("v_mad_i32_i24", 8); run<3>("v_mul_lo_u32", 8);
    run<4>("v_alignbyte_b32", 8); run<5>("v_cndmask_b32", 8); run<6>("v_cmp_eq_u32 -> sgpr", 8);
    run<7>("v_readlane_b32", 8); run<8>("v_writelane_b32", 8); run<9>("v_fma_f32", 8); run<10>("v_perm_b32", 8);
    run<11>("v_lshlrev_b32", 8); run<12>("v_add_u32_sdwa", 8); run<13>("v_lshl_or_b32", 8); run<14>("s_add_u32", 8);
    run<15>("ds_read_b32 (x8+wait)", 8); run<16>("ds_write_b8 (x8+wait)", 8); run<17>("ds_read2_b32 (x4+wait)", 4);
    run<18>("ds_write_b32 (x8+wait)", 8);
    run<19>("v_cndmask_b32 e64 sgpr", 8);
    run<20>("v_bfi_b32", 8);
    run<21>("v_and_or_b32", 8);
    run<22>("v_or_b32", 8);
    run<23>("v_sub_u32", 8);
    run<24>("v_mov_b32", 8);
    run<25>("v_bfe_u32", 8);
    run<26>("v_add3_u32", 8);
    run<27>("v_lshrrev_b32", 8);
    run<28>("v_xor_b32", 8);
    run<29>("v_cndmask vcc (vcc set)", 8);
    run<30>("v_min_u32", 8);
    run<31>("v_mul_u32_u24", 8);
    run<32>("v_add_u32 x2 lit", 8);
    run<41>("v_pk_max_i16", 8);
    run<45>("v_pk_min_i16", 8);
    run<42>("v_dot2_i32_i16", 8);
    run<43>("v_mad_i32_i16", 8);
    run<44>("v_med3_i32", 8);
    return 0;
}
