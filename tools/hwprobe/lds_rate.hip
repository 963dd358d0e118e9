// lds_rate.hip -- what the CU's LDS pipe charges for the access patterns of the fused kernel's P2 / P4 / P5, on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/hwprobe/lds_rate.hip -o tools/hwprobe/lds_rate && tools/hwprobe/lds_rate
// 1024 workgroups of 256 threads with 36 KiB of LDS each: exactly 4 per CU = 16 wavefronts per CU (4 per SIMD), all resident,
// every wavefront issuing the same LDS instruction ITER x 8 times (8 in flight, then s_waitcnt).  Reported: shader cycles per
// instruction PER CU (s_memtime of one wavefront / (its instructions x 16 wavefronts)) -- the pipe's throughput cost of the
// pattern when every wavefront of the CU is in the same phase, which is the situation of a single-round launch.
// Patterns (V = 7: a view is 49 cells, lanes 0..48 active where noted):
//   w16+w8   the P4 of round 3: one aligned 2-byte store + one byte store per cell at byte 3*lane          (2 instructions)
//   w32u     ONE 4-byte store per cell at byte 3*lane (unaligned; the 4th byte is the next cell's first)   -- also checked for
//            correctness: does an unaligned ds_write_b32 land where the address says?
//   w32q     ONE aligned 4-byte store by 3 lanes of every 4 (12 bytes = 4 cells per quad)
//   w32      aligned 4-byte store, all lanes
//   r128b    ds_read_b128, every lane the same address (the view-record broadcast of P2)
//   r16g     ds_read_u16 gather of a 7x7 window out of a 16-wide tile of 2-byte cells (P2)
//   r128     ds_read_b128 at 16*lane (P5)
//   r32      ds_read_b32 at 4*lane
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define REP8(x) x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *cyc, int iters) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *base = lds + wave * 9216;                                       // a private 9 KiB slice per wavefront
    for (int i = lane; i < 9216 / 4; i += 64) reinterpret_cast<uint32_t *>(base)[i] = i * 0x01010101u;
    __syncthreads();
    const uint32_t b0 = (uint32_t)(uintptr_t)base;                           // LDS byte address of the slice
    uint32_t v0 = lane * 0x00010203u + 0x40414243u, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    uint32_t acc = 0;
    const bool act = lane < 49;
    const uint32_t a3 = b0 + 3 * lane;                                       // byte 3*lane
    const uint32_t par = (3 * lane) & 1;                                     // P4: the aligned 2-byte part starts at 3*lane + par
    const uint32_t a16 = a3 + par, a8 = a3 + 2 * (par ^ 1);
    const uint32_t aq = b0 + 4 * (3 * (lane >> 2) + (lane & 3));             // quad q writes dwords 3q .. 3q+2
    const bool actq = (lane & 3) != 3 && lane < 52;
    const uint32_t a4 = b0 + 4 * lane, a128 = b0 + 16 * lane;
    const uint32_t ag = b0 + 2 * (((lane / 7) + 3) * 16 + (lane % 7) + 5);   // window at (5, 3) of a 16-wide tile
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0 && act) { REP8(asm volatile("ds_write_b16 %0, %2\n ds_write_b8_d16_hi %1, %2" :: "v"(a16), "v"(a8), "v"(v0) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 1 && act) { REP8(asm volatile("ds_write_b32 %0, %1" :: "v"(a3), "v"(v0) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 2 && actq) { REP8(asm volatile("ds_write_b32 %0, %1" :: "v"(aq), "v"(v0) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 3) { REP8(asm volatile("ds_write_b32 %0, %1" :: "v"(a4), "v"(v0) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 4) { uint32_t r[4]; REP8(asm volatile("ds_read_b128 %0, %1" : "=v"(*(__uint128_t *)r) : "v"(b0) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); acc += r[0]; }
        if (OP == 5 && act) { uint32_t r; REP8(asm volatile("ds_read_u16 %0, %1" : "=v"(r) : "v"(ag) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); acc += r; }
        if (OP == 6) { uint32_t r[4]; REP8(asm volatile("ds_read_b128 %0, %1" : "=v"(*(__uint128_t *)r) : "v"(a128) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); acc += r[0]; }
        if (OP == 7) { uint32_t r; REP8(asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(a4) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); acc += r; }
        if (OP == 8 && act) { REP8(asm volatile("ds_write_b16 %0, %1" :: "v"(a16), "v"(v0) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 9) { uint32_t r[2]; REP8(asm volatile("ds_read_b64 %0, %1" : "=v"(*(uint64_t *)r) : "v"(b0) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); acc += r[0]; }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + v1 + v2 + v3;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// does an unaligned ds_write_b32 store its four bytes at the byte address given?  (lane i: bytes 3i .. 3i+3, increasing lanes
// in increasing instruction order would make byte 3i+3 the next lane's -- here every lane writes its own value, one lane at a time)
__global__ void check_unaligned(uint8_t *out) {
    __shared__ uint8_t buf[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) buf[i] = 0xee;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)buf + 3 * lane;
    const uint32_t v = 0x03020100u + 0x10101010u * (lane & 15);
    if (lane < 49 && (lane & 1) == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)");
    __syncthreads();
    if (lane < 49 && (lane & 1) == 1) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = buf[i];
}

template <int OP>
void run(const char *name, int insts_per_rep) {
    uint32_t *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 36864, 0, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 36864, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * insts_per_rep;
    printf("%-8s %6.2f cycles per instruction per CU (16 wavefronts issuing; wave 0: %.1f cycles per instruction of its own; kernel %.1f us)\n",
           name, (double)h / (n * 16), (double)h / n, ms * 1e3);
    hipFree(out); hipFree(cyc);
}

int main() {
    uint8_t *d, hb[256];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(check_unaligned, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(hb, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 49; ++i)                                            // bytes 3i..3i+2 are lane i's; byte 3i+3 is overwritten by lane i+1 (odd after even)
        for (int b = 0; b < 3; ++b) {
            // even lanes wrote first, odd lanes second: an even lane's bytes 0..2 survive unless the PREVIOUS odd lane's 4th byte hit byte 0
            const int owner = i;
            uint8_t want = (uint8_t)(b + 0x10 * (owner & 15));
            if (b == 0 && (i & 1) == 0 && i > 0) want = (uint8_t)(3 + 0x10 * ((i - 1) & 15));   // (the odd lane before it wrote its 4th byte later)
            if (hb[3 * i + b] != want) { if (bad < 8) printf("   byte %d: got %02x want %02x\n", 3 * i + b, hb[3 * i + b], want); ++bad; }
        }
    printf("unaligned ds_write_b32 at byte 3*lane: %s (%d mismatches)\n", bad ? "NOT byte-addressed as assumed" : "lands at the byte address", bad);
    run<0>("w16+w8", 2); run<8>("w16", 1); run<1>("w32u", 1); run<2>("w32q", 1); run<3>("w32", 1);
    run<4>("r128b", 1); run<9>("r64b", 1); run<5>("r16g", 1); run<6>("r128", 1); run<7>("r32", 1);
    return 0;
}
