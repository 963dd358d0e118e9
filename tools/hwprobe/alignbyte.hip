// Hardware probe (gfx950): does v_alignbyte_b32 use only src2[1:0]?  does ds_read2_b32 honour the low 2 address bits?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef const uint32_t __attribute__((address_space(3))) *lds_u32_ptr;
__global__ void k(uint32_t *out) {
    __shared__ uint32_t buf[64];
    buf[threadIdx.x] = 0x03020100u + 0x04040404u * threadIdx.x;   // byte b of the array holds value b
    __syncthreads();
    uint32_t sh = threadIdx.x;                                    // 0..63
    out[threadIdx.x] = __builtin_amdgcn_alignbyte(0x07060504u, 0x03020100u, sh);
    // read2 at a misaligned byte address: bytes 8+lane&3 ...
    uint32_t addr = (uint32_t)(uintptr_t)(lds_u32_ptr)buf + 8 + (threadIdx.x & 3);
    uint32_t lo, hi;
    asm volatile("ds_read2_b32 %0, %2 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t *)&lo) : "v"(0), "v"(addr));
    uint64_t v; asm volatile("ds_read2_b32 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[64 + threadIdx.x] = (uint32_t)v; out[128 + threadIdx.x] = (uint32_t)(v >> 32);
}
int main() {
    uint32_t *d; hipMalloc(&d, 192 * 4);
    k<<<1, 64>>>(d);
    uint32_t h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("alignbyte sh=%d -> %08x\n", i, h[i]);
    printf("alignbyte sh=35 -> %08x, sh=63 -> %08x\n", h[35], h[63]);
    for (int i = 0; i < 4; ++i) printf("read2 at byte 8+%d -> %08x %08x\n", i, h[64 + i], h[128 + i]);
    return 0;
}
