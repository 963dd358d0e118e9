// mgx_aux.hip -- the kernels either side of the fused step, gfx950.
//
//   mgx_one_hot     OneHotObsWrapper.one_hot        multigrid/wrappers.py:158-190   (HBM write-bound: 3 B in, 21 B out)
//   mgx_full_obs    FullyObsWrapper.observation     multigrid/wrappers.py:48-58     (grid transpose-copy + agent overlay)
//   mgx_reset_done  vector-env auto-reset from a pool of pre-generated layouts (build-defined; the reference has no
//                   batching: its user calls reset() when is_done(), multigrid/base.py:250-301, 534-539)
//
// All three are streaming kernels: 16-byte loads and stores, the byte shuffling in between goes through LDS.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <limits.h>
#include <stdint.h>

#include "mgx_rules.h"

extern "C" void mgx_internal_set_hip_error(int e);      // mgx_kernels.hip: what mgx_last_hip_error() reports

namespace {

using namespace mgx;


typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const uint32_t __attribute__((address_space(3))) *lds_u32_ptr;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ void wave_sync() {          // LDS traffic inside ONE wavefront is in order: compiler fence only
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// one_hot: out[cell][d0 + d1 + d2] = 1 at {x0, d0 + x1, d0 + d1 + x2}.
// A workgroup takes chunks of 1024 cells: (1) each thread loads 12 bytes = 4 cells and leaves their D-bit one-hot masks
// in LDS, (2) each thread assembles 16 output bytes at a time from the masks of the 1-2 cells they span (any D works)
// and stores them as one vector.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kOhCells = 1024;

__device__ __forceinline__ uint32_t one_hot_mask(uint32_t c, int d0, int d1, int d2) {
    const uint32_t p0 = c & 0xffu, p1 = (c >> 8) & 0xffu, p2 = (c >> 16) & 0xffu;
    uint32_t m = 0;
    m |= (p0 < (uint32_t)d0) ? (1u << p0) : 0u;
    m |= (p1 < (uint32_t)d1) ? (1u << (d0 + p1)) : 0u;
    m |= (p2 < (uint32_t)d2) ? (1u << (d0 + d1 + p2)) : 0u;
    return m;
}

__global__ __launch_bounds__(256) void one_hot_kernel(const uint8_t *__restrict__ x, int64_t n_cells, int d0, int d1, int d2,
                                                      uint32_t inv_D, uint8_t *__restrict__ out) {
    __shared__ __align__(16) uint32_t masks[kOhCells + 32];
    const int D = d0 + d1 + d2;                                   // 3 <= D <= 32;  inv_D = ceil(2^32 / D)
    const int64_t nchunks = (n_cells + kOhCells - 1) / kOhCells;
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 3) == 0;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t c_base = chunk * kOhCells;
        const int ncell = (int)min((int64_t)kOhCells, n_cells - c_base);
        const uint8_t *src = x + c_base * 3;
        uint32_t c0, c1, c2, c3;
        if (aligned) {                                            // 12 bytes per thread; past the end reads zeros
            // (the range is rounded up to the aligned dword that holds the last valid byte: the check is per dword)
            const u32x3 w = __builtin_amdgcn_raw_buffer_load_b96(make_rsrc(src, (ncell * 3 + 3) & ~3), threadIdx.x * 12, 0, 0);
            c0 = w.x; c1 = (w.x >> 24) | (w.y << 8); c2 = (w.y >> 16) | (w.z << 16); c3 = w.z >> 8;
        } else {
            uint32_t c[4];
            for (int k = 0; k < 4; ++k) {
                const int i = threadIdx.x * 4 + k;
                c[k] = i < ncell ? load_obs_cell(src + i * 3) : 0u;
            }
            c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3];
        }
        u32x4 m;
        m.x = one_hot_mask(c0, d0, d1, d2); m.y = one_hot_mask(c1, d0, d1, d2);
        m.z = one_hot_mask(c2, d0, d1, d2); m.w = one_hot_mask(c3, d0, d1, d2);
        reinterpret_cast<u32x4 *>(masks)[threadIdx.x] = m;
        __syncthreads();
        const int total = ncell * D;                              // output bytes of this chunk
        uint8_t *dst = out + c_base * D;                          // 16-byte aligned: c_base is a multiple of 1024
        for (int o = threadIdx.x * 16; o < total; o += 256 * 16) {
            const int cq = (int)__umulhi((uint32_t)o, inv_D);     // o / D, exact for o < 2^20
            const int k0 = o - cq * D;
            uint32_t bits = masks[cq] >> k0;                      // bits of up to 16 consecutive output bytes
            int have = D - k0, c = cq + 1;
            while (have < 16) { bits |= masks[c] << have; have += D; ++c; }
            u32x4 v;                                              // spread 4 bits -> 4 bytes of 0/1
            v.x = (((bits >> 0) & 0xfu) * 0x00204081u) & 0x01010101u;
            v.y = (((bits >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
            v.z = (((bits >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
            v.w = (((bits >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
            if (o + 16 <= total) {
                *reinterpret_cast<u32x4 *>(dst + o) = v;
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (int b = 0; o + b < total; ++b) dst[o + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// full_obs: img = grid.encode() (a copy of Grid.state, (W,H,3) indexed [x][y]); img[agent.pos] = agent.encode() for
// every agent in index order, terminated or not (wrappers.py:52-54).
// Every wavefront is autonomous (as in the fused kernel): it takes G consecutive envs, loads their [y][x] tiles into LDS
// with 16-byte vectors, writes each cell to its [x][y] place in a second LDS buffer (one lane per cell), overlays the
// agents there, and streams that buffer out with 16-byte vectors.  Input and output of an env have the same size, so the
// two buffers share one 16-byte skew.
// ---------------------------------------------------------------------------------------------------------------
// CB (template): bytes per grid cell -- 1 compact, 2 packed, 3 the reference's triples.  (As a run-time argument -- round 5's first
// form -- every cell of the transposition went through the format's branches and the loop through one LDS round trip per
// iteration: 0.55 -> 0.42 of the roofline.  Now: the format fixed per instantiation, four cells in flight per lane, shifts
// instead of the two reciprocal multiplies when W is a power of two.)
template <int CB>
__global__ __launch_bounds__(256) void full_obs_kernel(int W, int H, int A, int G, int wave_lds, int in_buf, uint32_t inv_W,
                                                       uint32_t inv_H, uint32_t inv_HW, int64_t batch, const uint8_t *__restrict__ grid,
                                                       const uint8_t *__restrict__ agents, uint8_t *__restrict__ out) {
    constexpr int cb = CB;
    extern __shared__ __align__(16) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t e0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * G;
    if (e0 >= batch) return;
    const int Gc = (int)min((int64_t)G, batch - e0);
    const int HW = H * W, HWB = HW * cb, HW3 = HW * 3;
    // input: packed cells [y][x], 2 bytes each (compact cells: 1); output: (type, color, state) bytes [x][y] -- each with its own 16-byte skew
    const int64_t g0 = e0 * HWB, gtotal = batch * (int64_t)HWB;
    const int64_t ga = g0 & ~(int64_t)15;
    const int iskew = (int)(g0 - ga);
    const int ilen = iskew + Gc * HWB;                             // staged input bytes, from the aligned start
    const int64_t o0 = e0 * HW3;
    const int64_t oa = o0 & ~(int64_t)15;
    const int oskew = (int)(o0 - oa);
    const int olen = oskew + Gc * HW3;
    uint8_t *in_raw = lds + wave * wave_lds, *out_raw = in_raw + in_buf;
    const int lane16 = lane * 16;
    // (1) tile -> LDS; lanes past the end of the tensor read zeros (never used)
    // (range rounded up to the aligned 16 bytes that hold the tensor's last byte: the range check is per dword)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(grid + ga, (int)min((gtotal - ga + 15) & ~(int64_t)15, (int64_t)INT_MAX & ~15));
    for (int rel = lane16; rel < ilen; rel += 1024 * 4) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, rel + 1024 * u, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (rel + 1024 * u < ilen) *reinterpret_cast<u32x4 *>(in_raw + rel + 1024 * u) = v[u];
    }
    wave_sync();
    // (2) one lane per cell: [y][x] packed -> [x][y] bytes
    const uint8_t *in_cells = in_raw + iskew;
    uint8_t *out_cells = out_raw + oskew;
    const int ncell = Gc * HW;
    const bool pow2 = (W & (W - 1)) == 0 && (HW & (HW - 1)) == 0;                         // (wave-uniform)
    const int shW = 31 - __builtin_clz((unsigned)W), shHW = 31 - __builtin_clz((unsigned)HW);
    const bool pow2h = (H & (H - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int shH = 31 - __builtin_clz((unsigned)H);
    auto move = [&](const int i, const uint32_t c) {
        int e, r, y, xx;
        if (pow2) { e = i >> shHW; r = i & (HW - 1); y = r >> shW; xx = r & (W - 1); }
        else {
            e = (int)__umulhi((uint32_t)i, inv_HW);                                         // i / HW  (i < 2^16)
            r = i - e * HW;
            y = (int)__umulhi((uint32_t)r, inv_W); xx = r - y * W;                          // r / W
        }
        // (one aligned 2-byte store + one byte store instead of three byte stores, as the fused kernel's P4 does, measured SLOWER here:
        // 0.358 against 0.317 ms at 1 M envs -- the parity selects cost more than the third store)
        uint8_t *d = out_cells + e * HW3 + (xx * H + y) * 3;
        d[0] = (uint8_t)c; d[1] = (uint8_t)(c >> 8); d[2] = (uint8_t)(c >> 16);
    };
    int i = lane;
#ifndef MGX_FULL_OBS_BY_OUTPUT
#define MGX_FULL_OBS_BY_OUTPUT 1
#endif
    // (round 6) OUTPUT order when the wavefront's slice of the output starts on a dword: a lane owns four consecutive output cells --
    // twelve bytes, three aligned dwords -- and gathers their packed cells from [y][x]: one LDS read per cell and three dword writes
    // per four cells instead of three byte writes per cell, the (e, x, y) decomposition once per four cells (same box: profiles/
    // r6_full_obs.txt).  The ragged last cells and slices that start off a dword keep the input-ordered pass below.
    if (MGX_FULL_OBS_BY_OUTPUT && (oskew & 3) == 0) {
        const int ngrp = ncell >> 2;
        auto gather4 = [&](const int k, uint32_t (&c)[4]) {
            const int o = 4 * k;
            int e, r, xx, y;
            if (pow2h) { e = o >> shHW; r = o & (HW - 1); xx = r >> shH; y = r & (H - 1); }
            else {
                e = (int)__umulhi((uint32_t)o, inv_HW); r = o - e * HW;
                xx = (int)__umulhi((uint32_t)r, inv_H); y = r - xx * H;
            }
            int src = e * HW + y * W + xx;                              // cell index [e][y][x]
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c[u] = load_cell_shown(cb, in_cells + src * cb);
                ++y; src += W;
                if (y == H) { y = 0; ++xx; src += 1 - HW; if (xx == W) { xx = 0; src += HW - W; } }   // next column / next env
            }
        };
        auto put4 = [&](const int k, const uint32_t (&c)[4]) {
            uint32_t *d = reinterpret_cast<uint32_t *>(out_cells + 12 * k);
            d[0] = c[0] | (c[1] << 24);
            d[1] = (c[1] >> 8) | (c[2] << 16);
            d[2] = (c[2] >> 16) | (c[3] << 8);
        };
        int k = lane;
        for (; k + 64 < ngrp; k += 128) {                                                  // eight cells in flight per lane
            uint32_t c0[4], c1[4];
            gather4(k, c0); gather4(k + 64, c1);
            put4(k, c0); put4(k + 64, c1);
        }
        for (; k < ngrp; k += 64) { uint32_t c0[4]; gather4(k, c0); put4(k, c0); }
        const int o = 4 * ngrp + lane;                                                     // (<= 3 OUTPUT cells left: the last env's last column)
        if (o < ncell) {
            int e, r, xx, y;
            if (pow2h) { e = o >> shHW; r = o & (HW - 1); xx = r >> shH; y = r & (H - 1); }
            else {
                e = (int)__umulhi((uint32_t)o, inv_HW); r = o - e * HW;
                xx = (int)__umulhi((uint32_t)r, inv_H); y = r - xx * H;
            }
            store_obs_cell(out_cells + 3 * o, load_cell_shown(cb, in_cells + (e * HW + y * W + xx) * cb));
        }
        i = ncell;
    }
    for (; i + 192 < ncell; i += 256) {                                                    // four cells in flight per lane
        uint32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = load_cell_shown(cb, in_cells + (i + 64 * u) * cb);   // (a box's content is not part of Grid.state)
#pragma unroll
        for (int u = 0; u < 4; ++u) move(i + 64 * u, c[u]);
    }
    for (; i < ncell; i += 64) move(i, load_cell_shown(cb, in_cells + i * cb));
    wave_sync();
    // (3) agents, index order: a later agent overwrites an earlier one on the same cell, so only the last one writes
    const uint64_t *rows = reinterpret_cast<const uint64_t *>(agents) + e0 * A;
    for (int s = lane; s < Gc * A; s += 64) {
        const int e = s / A, ai = s - e * A;
        const uint64_t r = rows[s];
        const uint32_t pos = ((uint32_t)r >> 16) & 0xffffu;
        bool shadowed = false;
        for (int j = ai + 1; j < A; ++j) shadowed |= ((((uint32_t)rows[e * A + j]) >> 16) & 0xffffu) == pos;
        const int x = row_x(r), y = row_y(r);
        if (!shadowed && x < W && y < H)
            store_obs_cell(out_cells + e * HW3 + (x * H + y) * 3, (uint32_t)T_AGENT | ((uint32_t)(r & 0xffffu) << 8));
    }
    wave_sync();
    // (4) stream out
    uint8_t *gdst = out + oa;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(gdst, olen);
    for (int rel = lane16; rel < olen; rel += 1024) {
        if ((rel >= oskew) & (rel + 16 <= olen)) {
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(out_raw + rel), ro, rel, 0, 0);
        } else {
            const int lo_b = max(rel, oskew), hi_b = min(rel + 16, olen);
#pragma clang loop vectorize(disable) unroll(disable)
            for (int B = lo_b; B < hi_b; ++B) gdst[B] = out_raw[B];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pack / unpack: (type, color, state) bytes <-> packed 16-bit cells (include/mgx.h).  One thread converts 8 cells: 24 bytes in
// three dword pairs on one side, one 16-byte vector on the other.
// ---------------------------------------------------------------------------------------------------------------
// W, H > 0: the cells are whole env grids [b][y][x] and bad[1] counts the cells of every env's OUTER RING that are not the
// reference's WALL = (wall, grey, 0) -- the precondition of every kernel that reads the grid (include/mgx.h)
// C8: the output is COMPACT one-byte cells (include/mgx.h: MgxCell8); bad[0] then also counts a state on anything but a door or an
// agent overlay, which that format cannot hold
template <typename OutT, bool C8>
__global__ __launch_bounds__(256) void pack_grid_kernel(const uint8_t *__restrict__ c3, int64_t n, OutT *__restrict__ out,
                                                        int32_t *bad, int W, int H) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;
    int nbad = 0, nring = 0;
    OutT v[8];
    int x = 0, y = 0;
    if (W > 0) { const int r = (int)(i0 % ((int64_t)W * H)); y = r / W; x = r - y * W; }
    // 8 cells = 24 bytes: three 8-byte loads when the chunk is whole and aligned (byte loads otherwise)
    uint32_t w6[6];
    const bool whole = i0 + 8 <= n && (reinterpret_cast<uintptr_t>(c3) & 7) == 0;
    if (whole) {
        const u32x2 *src = reinterpret_cast<const u32x2 *>(c3 + i0 * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) { const u32x2 t = src[k]; w6[2 * k] = t.x; w6[2 * k + 1] = t.y; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t c;
        if (whole) {                                                     // bytes [3k, 3k + 3) of the 24
            const int d = (3 * k) >> 2, sh = 8 * ((3 * k) & 3);
            c = (sh <= 8 ? w6[d] >> sh : (w6[d] >> sh) | (w6[d + 1] << (32 - sh))) & 0xffffffu;
        } else {
            c = (i0 + k < n) ? load_obs_cell(c3 + (i0 + k) * 3) : 0u;
        }
        // the state byte's upper six bits: a box's content (kind | colour << 3, mgx_rules.h), nothing on any other cell
        const uint32_t content = (c >> 18) & 0x3fu;
        nbad += ((c & 0xf0u) != 0) | (((c >> 8) & 0xf8u) != 0)
              | ((content != 0) & (((c & 0xffu) != (uint32_t)T_BOX) | ((content & 7u) == 0) | ((content >> 3) > 5u)));
        if constexpr (C8) {
            nbad += content != 0;                                        // (the compact format has no room for one)
            const uint32_t t = c & 0xffu;
            nbad += (((c >> 16) & 3u) != 0) & (t != (uint32_t)T_DOOR) & (t != (uint32_t)T_AGENT);
            nbad += (t > 10u) & (t < 16u);                               // (types 11..15 would alias the joint codes)
            v[k] = (OutT)cell8_pack(c);
        } else {
            v[k] = (OutT)cell_pack(c);
        }
        if (W > 0) {
            const bool ring = (x == 0) | (x == W - 1) | (y == 0) | (y == H - 1);
            nring += (i0 + k < n) & ring & (c != CELL_WALL);
            if (++x == W) { x = 0; if (++y == H) y = 0; }
        }
    }
    if (nring && bad) atomicAdd(bad + 1, nring);
    if (C8 && i0 + 8 <= n && (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
        u32x2 w;
        w.x = v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        w.y = v[4] | ((uint32_t)v[5] << 8) | ((uint32_t)v[6] << 16) | ((uint32_t)v[7] << 24);
        *reinterpret_cast<u32x2 *>(out + i0) = w;
    } else if (!C8 && i0 + 8 <= n && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        u32x4 w;
        w.x = v[0] | ((uint32_t)v[1] << 16); w.y = v[2] | ((uint32_t)v[3] << 16);
        w.z = v[4] | ((uint32_t)v[5] << 16); w.w = v[6] | ((uint32_t)v[7] << 16);
        *reinterpret_cast<u32x4 *>(out + i0) = w;
    } else {
        for (int k = 0; k < 8 && i0 + k < n; ++k) out[i0 + k] = v[k];
    }
    if (nbad && bad) atomicAdd(bad, nbad);
}

// mgx_check_grid: the state preconditions of the kernels, counted (include/mgx.h).  One thread looks at 8 cells and at one
// agent row.
template <typename CellT, bool C8>
__global__ __launch_bounds__(256) void check_grid_kernel(const CellT *__restrict__ cells, int64_t n, int W, int H,
                                                         const uint8_t *__restrict__ agents, int64_t n_rows, int A, int32_t *bad) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = tid * 8;
    int nfmt = 0, nring = 0, nag = 0;
    int64_t first = INT64_MAX;
    if (i0 < n) {
        const int64_t HW = (int64_t)W * H;
        const int64_t b0 = i0 / HW;
        int r = (int)(i0 - b0 * HW), y = r / W, x = r - y * W;
        int64_t b = b0;
        for (int k = 0; k < 8 && i0 + k < n; ++k) {
            const uint32_t p = cells[i0 + k];
            bool fmt;
            if constexpr (C8) {
                // a code the reference has (no door / overlay state 3: tcode 13..15 are agent overlays, never stored in a grid's
                // own cells but valid in the format), a colour it has, and the opaque bit what the (type, state) says
                const uint32_t u = cell8_unpack(p);
                fmt = (((p >> 4) & 7u) > 5u) | (cell8_pack(u) != p);
            } else {
                const uint32_t t = p & 0xfu, c = (p >> 8) & 0x7u, st = (p >> 12) & 0x3u;
                // content bits only on a box (a kind, a colour the reference has), values the reference has, and the opaque bit
                // what the (type, state) says (obs.py:46-63)
                const uint32_t full = cell_unpack_full(p), content = (full >> 18) & 0x3fu;
                fmt = ((content != 0) & ((t != (uint32_t)T_BOX) | ((content & 7u) == 0) | ((content >> 3) > 5u)))
                    | (t > (uint32_t)T_AGENT) | (c > 5u) | (st > 2u) | (cell_pack(full) != p);
            }
            const bool ring = (x == 0) | (x == W - 1) | (y == 0) | (y == H - 1);
            const bool rbad = ring & (p != (C8 ? CELL8_WALL : CELL16_WALL));
            nfmt += fmt; nring += rbad;
            if ((fmt | rbad) && b < first) first = b;
            if (++x == W) { x = 0; if (++y == H) { y = 0; ++b; } }
        }
    }
    if (tid < n_rows) {
        const uint8_t *row = agents + tid * MGX_AGENT_STRIDE;
        const int ax = row[AG_X], ay = row[AG_Y];
        // inside the walls (never on the ring: nothing can stand on a wall), a direction, 0/1 terminated, a packable carried cell
        const bool abad = (ax < 1) | (ax > W - 2) | (ay < 1) | (ay > H - 2) | (row[AG_DIR] > 3) | (row[AG_TERM] > 1) | (row[AG_COLOR] > 5)
                        | (row[AG_CARRY] > (uint8_t)T_AGENT) | (row[AG_CARRY + 1] > 5) | ((row[AG_CARRY + 2] & 3) > 2)
                        | ((row[AG_CARRY + 2] >> 2) != 0 && (C8 || row[AG_CARRY] != (uint8_t)T_BOX || ((row[AG_CARRY + 2] >> 2) & 7) == 0
                                                             || (row[AG_CARRY + 2] >> 5) > 5));      // (a carried box's content)
        nag += abad;
        if (abad && tid / A < first) first = tid / A;
    }
    if (nfmt) atomicAdd(bad + 0, nfmt);
    if (nring) atomicAdd(bad + 1, nring);
    if (nag) atomicAdd(bad + 2, nag);
    if (first != INT64_MAX) atomicMin(bad + 3, (int32_t)(first > INT32_MAX ? INT32_MAX : first));
}

__global__ __launch_bounds__(256) void unpack_grid_kernel(const uint16_t *__restrict__ in, int64_t n, uint8_t *__restrict__ c3) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;
    if (i0 + 8 <= n && ((reinterpret_cast<uintptr_t>(in) & 15) | (reinterpret_cast<uintptr_t>(c3) & 7)) == 0) {
        // one 16-byte load, 24 bytes out as three 8-byte stores
        const u32x4 p = *reinterpret_cast<const u32x4 *>(in + i0);
        const uint32_t pw[4] = {p.x, p.y, p.z, p.w};
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = cell_unpack_full((pw[k >> 1] >> (16 * (k & 1))) & 0xffffu);   // (with a box's content)
        uint32_t o[6];
        o[0] = c[0] | (c[1] << 24);               o[1] = (c[1] >> 8) | (c[2] << 16);   o[2] = (c[2] >> 16) | (c[3] << 8);
        o[3] = c[4] | (c[5] << 24);               o[4] = (c[5] >> 8) | (c[6] << 16);   o[5] = (c[6] >> 16) | (c[7] << 8);
        u32x2 *dst = reinterpret_cast<u32x2 *>(c3 + i0 * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[k] = u32x2{o[2 * k], o[2 * k + 1]};
        return;
    }
    for (int k = 0; k < 8 && i0 + k < n; ++k) store_obs_cell(c3 + (i0 + k) * 3, cell_unpack_full(in[i0 + k]));
}

__global__ __launch_bounds__(256) void unpack_grid8_kernel(const uint8_t *__restrict__ in, int64_t n, uint8_t *__restrict__ c3) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;
    if (i0 + 8 <= n && ((reinterpret_cast<uintptr_t>(in) & 7) | (reinterpret_cast<uintptr_t>(c3) & 7)) == 0) {
        const u32x2 p = *reinterpret_cast<const u32x2 *>(in + i0);        // 8 cells in, 24 bytes out as three 8-byte stores
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = cell8_unpack(((k < 4 ? p.x : p.y) >> (8 * (k & 3))) & 0xffu);
        uint32_t o[6];
        o[0] = c[0] | (c[1] << 24);               o[1] = (c[1] >> 8) | (c[2] << 16);   o[2] = (c[2] >> 16) | (c[3] << 8);
        o[3] = c[4] | (c[5] << 24);               o[4] = (c[5] >> 8) | (c[6] << 16);   o[5] = (c[6] >> 16) | (c[7] << 8);
        u32x2 *dst = reinterpret_cast<u32x2 *>(c3 + i0 * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[k] = u32x2{o[2 * k], o[2 * k + 1]};
        return;
    }
    for (int k = 0; k < 8 && i0 + k < n; ++k) store_obs_cell(c3 + (i0 + k) * 3, cell8_unpack(in[i0 + k]));
}

// ---------------------------------------------------------------------------------------------------------------
// reset_done: every env whose episode is over (all agents terminated, or step_count >= max_steps: base.py:534-539)
// is re-initialised from layout pool[(global_env + episode * stride) mod K]; step_count := 0, episode += 1.
// One lane per env decides; the wavefront then copies the layouts of its finished envs with all 64 lanes, in the widest
// unit the layout size allows.
// ---------------------------------------------------------------------------------------------------------------
template <typename VecT>
__device__ __forceinline__ void copy_layouts(int n, int units, uint32_t inv_units, int64_t e0, int64_t env_bytes,
                                             const int *l_env, const int *l_lay, const uint8_t *pool, uint8_t *dstbase,
                                             int lane) {
    if (units >= 64) {                                            // big layouts: env by env, lanes over its vectors
        for (int j = 0; j < n; ++j) {
            const VecT *s = reinterpret_cast<const VecT *>(pool + (int64_t)l_lay[j] * env_bytes);
            VecT *d = reinterpret_cast<VecT *>(dstbase + (e0 + l_env[j]) * env_bytes);
            for (int v = lane; v < units; v += 64) d[v] = s[v];
        }
    } else {                                                      // small layouts: (env, vector) pairs flattened over the lanes
        const int total = n * units;
        for (int k = lane; k < total; k += 64) {
            // k / units, k < 4096 (units == 1 -- one agent: ceil(2^32 / 1) does not fit the 32-bit reciprocal, which arrived as 0 and
            // sent every finished env's row to the first one: found by tests/test_instantiations.py, round 6)
            const int j = units == 1 ? k : (int)__umulhi((uint32_t)k, inv_units), v = k - j * units;
            const VecT *s = reinterpret_cast<const VecT *>(pool + (int64_t)l_lay[j] * env_bytes);
            VecT *d = reinterpret_cast<VecT *>(dstbase + (e0 + l_env[j]) * env_bytes);
            d[v] = s[v];
        }
    }
}

template <typename VecT>
__global__ __launch_bounds__(256) void reset_done_kernel(int HWB, int A, int max_steps, int64_t batch, int64_t first_env,
                                                         int K, uint32_t inv_units, uint32_t inv_A,
                                                         const uint8_t *__restrict__ pool_grid,
                                                         const uint8_t *__restrict__ pool_agents,
                                                         const uint8_t *__restrict__ pool_aux, uint8_t *grid,
                                                         uint8_t *agents, int32_t *step_count, uint8_t *aux,
                                                         int32_t *episode, uint8_t *was_reset) {
    __shared__ int s_env[4][64], s_lay[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t e0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (e0 >= batch) return;
    const int64_t b = e0 + lane;
    bool done = false;
    int layout = 0;
    if (b < batch) {
        const uint64_t *rows = reinterpret_cast<const uint64_t *>(agents) + b * A;
        bool all_term = true;
        for (int a = 0; a < A; ++a) all_term &= row_term(rows[a]);
        done = all_term || step_count[b] >= max_steps;
        if (done) {
            const int ep = episode[b];
            // a fixed odd stride walks the whole pool before repeating; depends only on the GLOBAL env index
            layout = (int)((uint64_t)(first_env + b + (int64_t)ep * 7919) % (uint64_t)K);
            episode[b] = ep + 1;
            step_count[b] = 0;
        }
        if (was_reset) was_reset[b] = (uint8_t)done;
    }
    const uint64_t mask = __builtin_amdgcn_ballot_w64(done);
    if (mask == 0) return;
    const int n = __builtin_popcountll(mask);
    if (done) {
        const int pos = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        s_env[wave][pos] = lane; s_lay[wave][pos] = layout;
    }
    wave_sync();
    const int *l_env = s_env[wave], *l_lay = s_lay[wave];
    copy_layouts<VecT>(n, HWB / (int)sizeof(VecT), inv_units, e0, HWB, l_env, l_lay, pool_grid, grid, lane);
    copy_layouts<uint64_t>(n, A, inv_A, e0, (int64_t)A * MGX_AGENT_STRIDE, l_env, l_lay, pool_agents, agents, lane);   // 8-byte rows
    if (aux && pool_aux)
        for (int j = lane; j < n; j += 64)
            reinterpret_cast<uint4 *>(aux)[e0 + l_env[j]] = reinterpret_cast<const uint4 *>(pool_aux)[l_lay[j]];
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent stepping, the producer's side (include/mgx.h: MgxPersistent; the consumer is MODE 3 of the fused kernel).
// A granule = {tag << 32 | 4 action bytes}, written by ONE relaxed agent-scope 8-byte store (write-through): the data is the
// flag (guide: publish/consume recipe R2).  Every poll is a relaxed agent-scope load; every spin is bounded.
// ---------------------------------------------------------------------------------------------------------------
// the 4 action bytes of granule q of an env (agents 4q .. 4q+3; agents beyond A: "absent")
__device__ __forceinline__ uint32_t granule_bytes(const int8_t *act_env, int A, int q) {
    if ((A & 3) == 0) return *reinterpret_cast<const uint32_t *>(act_env + 4 * q);       // (the tensor is 4-byte aligned: host check)
    uint32_t bytes = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ai = 4 * q + j;
        bytes |= (ai < A ? (uint32_t)(uint8_t)act_env[ai] : 0xffu) << (8 * j);
    }
    return bytes;
}
__device__ __forceinline__ uint64_t make_granule(const int8_t *act_env, int A, int q, uint32_t tag) {
    return ((uint64_t)tag << 32) | granule_bytes(act_env, A, q);
}

__global__ __launch_bounds__(256) void persistent_post_kernel(const int8_t *__restrict__ actions, int64_t n_granules, int A, int gpe,
                                                              uint32_t inv_gpe, uint32_t tag, uint64_t *granules) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_granules) return;
    const int64_t b = gpe == 1 ? g : (int64_t)(((uint64_t)g * inv_gpe) >> 32);     // g / gpe (n_granules < 2^31: checked by the host)
    const int q = (int)(g - b * gpe);
    __hip_atomic_store(granules + g, make_granule(actions + b * A, A, q, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all of done[0 .. waves) >= step?  One workgroup; returns false on timeout (uniform over the workgroup).  Every wavefront polls
// its own share of the flags on its own (relaxed agent loads, one ballot per pass: no workgroup barrier inside the loop -- it
// would add its ~0.2 us to every pass, i.e. to the latency of the hand-off); ONE barrier when all have seen theirs.
// NO LDS (round 6): a persistent launch that holds the whole chip -- C4's 65536 envs, 8 wavefronts x 20112 bytes per CU -- leaves no
// LDS granule free, and a kernel that asks for four bytes of it would wait for that launch to end.  The "somebody timed out" word
// the wavefronts of the workgroup share is `fail` = ctrl[4] in memory instead: sticky, which is what a dead hand-shake is.
__device__ __forceinline__ bool wait_all_done(const uint32_t *done, int waves, uint32_t step, uint32_t timeout_ticks, uint32_t *fail) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    const __amdgpu_buffer_rsrc_t drs = make_rsrc(done, waves * 4);
    const int stride = (int)blockDim.x;
    for (uint32_t spin = 1;; ++spin) {
        bool ok = true;
        for (int w0 = 0; w0 < waves; w0 += 8 * stride) {                  // 8 independent loads per thread in flight (sc1: aux 16)
            uint32_t f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = __builtin_amdgcn_raw_buffer_load_b32(drs, (int)threadIdx.x * 4, (w0 + k * stride) * 4, 16);
#pragma unroll
            for (int k = 0; k < 8; ++k) ok &= (w0 + k * stride + (int)threadIdx.x >= waves) | (f[k] >= step);
        }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        if ((spin & 31u) == 0) {
            const bool late = __builtin_amdgcn_s_memrealtime() - t0 > (uint64_t)timeout_ticks;
            if (late || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    return __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

__global__ __launch_bounds__(256) void persistent_wait_kernel(const uint32_t *done, int waves, uint32_t step, uint32_t *ctrl,
                                                              uint32_t timeout_ticks) {
    if (!wait_all_done(done, waves, step, timeout_ticks, ctrl + 4) && threadIdx.x == 0) atomicAdd(ctrl + 1, 1u);
}

// The recorded sequence's next step is known while the current one runs: its action bytes are fetched into registers behind the
// post (FAST = A is a multiple of 4: granule g's four bytes are the g-th dword of the step's action tensor; kFeedPre granules per
// thread, i.e. batches up to 8192 granules; whatever does not fit -- and every other A -- is fetched at post time).
// Workgroups of 256 threads = one wavefront per SIMD of a CU: they must fit BESIDE the persistent launch's wavefronts wherever
// they are (a 1024-thread workgroup needs four wavefronts per SIMD at once: next to two 140-VGPR wavefronts per SIMD there is
// no CU on the chip that can take it -- measured: the hand-shake then runs into its timeout).  One workgroup per slice of 2048
// granules (at most 32): 8-byte write-through stores leave ONE CU at ~15 GB/s (16384 granules from one workgroup: 9 us per
// step), so the post is spread; every workgroup sees ALL flags before it posts its slice (the lock-step of a real policy).
constexpr int kFeedPre = 8, kFeedThreads = 256, kFeedSlice = kFeedPre * kFeedThreads, kFeedMaxWgs = 32;
// GB: action bytes a granule takes from the tensor when they are contiguous and aligned -- 4 (A a multiple of 4), 2 (A = 2),
// 1 (A = 1): granule g's bytes are then the g-th GB-byte word of the step's action tensor; 0 = any other A
template <int GB>
__global__ __launch_bounds__(kFeedThreads) void persistent_feed_kernel(const int8_t *__restrict__ actions, int T, int64_t batch, int A, int gpe,
                                                                       uint32_t inv_gpe, uint64_t *granules, const uint32_t *done, int waves,
                                                                       uint32_t *ctrl, uint32_t timeout_ticks, uint64_t *trace) {
    const int64_t ng_all = batch * gpe, BA = batch * A;
    // this workgroup's slice of the granules: [g_lo, g_lo + ng)
    const int64_t per_wg = ((ng_all + gridDim.x - 1) / gridDim.x + 1) & ~(int64_t)1;
    const int64_t g_lo = (int64_t)blockIdx.x * per_wg;
    const int64_t ng = g_lo >= ng_all ? 0 : (ng_all - g_lo < per_wg ? ng_all - g_lo : per_wg);
    granules += g_lo;
    const bool lead = blockIdx.x == 0;
    if (!lead) trace = nullptr;
    constexpr bool FAST = GB != 0;
    const int npre = FAST ? (int)(ng < (int64_t)kFeedThreads * kFeedPre ? ng : (int64_t)kFeedThreads * kFeedPre) : 0;   // granules via registers
    const __amdgpu_buffer_rsrc_t grs = make_rsrc(granules, npre * 8);
    uint32_t pre[kFeedPre];
    auto fetch = [&](int t) {
        if constexpr (FAST) {
            const __amdgpu_buffer_rsrc_t ars = make_rsrc(actions + (int64_t)t * BA + g_lo * GB, npre * GB);
#pragma unroll
            for (int k = 0; k < kFeedPre; ++k) {            // (agents beyond A: "absent" = 0xff)
                if constexpr (GB == 4) pre[k] = __builtin_amdgcn_raw_buffer_load_b32(ars, threadIdx.x * 4, kFeedThreads * 4 * k, 0);
                else if constexpr (GB == 2) pre[k] = 0xffff0000u | (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(ars, threadIdx.x * 2, kFeedThreads * 2 * k, 0);
                else pre[k] = 0xffffff00u | (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(ars, threadIdx.x, kFeedThreads * k, 0);
            }
        }
    };
    fetch(0);
    for (int t = 0; t < T; ++t) {
        // the outputs of step t (counted from 1) complete?  (nothing to wait for before the first post)
        if (t > 0 && !wait_all_done(done, waves, (uint32_t)t, timeout_ticks, ctrl + 4)) {
            if (threadIdx.x == 0 && lead) { atomicAdd(ctrl + 1, 1u); __hip_atomic_store(ctrl + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            return;
        }
        if (trace && threadIdx.x == 0) trace[2 * t] = __builtin_amdgcn_s_memrealtime();
        const uint32_t tag = (uint32_t)t + 1u;
        if constexpr (FAST) {
            // one aligned 8-byte write-through store per granule (sc1, aux 16); lanes past the buffer's end are dropped
#pragma unroll
            for (int k = 0; k < kFeedPre; ++k) {
                const u32x2 v = {pre[k], tag};
                __builtin_amdgcn_raw_buffer_store_b64(v, grs, threadIdx.x * 8, kFeedThreads * 8 * k, 16);
            }
        }
        const int8_t *act_t = actions + (int64_t)t * BA + (FAST ? g_lo * GB : 0);
        int64_t gdone = npre;
        if constexpr (FAST) {                 // batches beyond one register chunk: fetched now, 32 loads in flight per thread
            for (; gdone < ng; gdone += kFeedThreads * kFeedPre) {
                const int n = (int)(ng - gdone < (int64_t)kFeedThreads * kFeedPre ? ng - gdone : (int64_t)kFeedThreads * kFeedPre);
                const __amdgpu_buffer_rsrc_t ars = make_rsrc(act_t + gdone * GB, n * GB), grs2 = make_rsrc(granules + gdone, n * 8);
                uint32_t w[kFeedPre];
#pragma unroll
                for (int k = 0; k < kFeedPre; ++k) {
                    if constexpr (GB == 4) w[k] = __builtin_amdgcn_raw_buffer_load_b32(ars, threadIdx.x * 4, kFeedThreads * 4 * k, 0);
                    else if constexpr (GB == 2) w[k] = 0xffff0000u | (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(ars, threadIdx.x * 2, kFeedThreads * 2 * k, 0);
                    else w[k] = 0xffffff00u | (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(ars, threadIdx.x, kFeedThreads * k, 0);
                }
#pragma unroll
                for (int k = 0; k < kFeedPre; ++k) {
                    const u32x2 v = {w[k], tag};
                    __builtin_amdgcn_raw_buffer_store_b64(v, grs2, threadIdx.x * 8, kFeedThreads * 8 * k, 16);
                }
            }
        }
        for (int64_t g = threadIdx.x + gdone; g < ng; g += kFeedThreads) {          // (any other A: by env and granule index)
            const int64_t gg = g_lo + g;
            const int64_t b = gpe == 1 ? gg : (int64_t)(((uint64_t)gg * inv_gpe) >> 32);
            __hip_atomic_store(granules + g, make_granule(act_t + b * A, A, (int)(gg - b * gpe), tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (trace) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) trace[2 * t + 1] = __builtin_amdgcn_s_memrealtime();
        }
        if (t + 1 < T) fetch(t + 1);
    }
    // (the last step's outputs: so that "the feed kernel has ended" means "the rollout is complete")
    if (!lead) return;
    if (!wait_all_done(done, waves, (uint32_t)T, timeout_ticks, ctrl + 4) && threadIdx.x == 0) atomicAdd(ctrl + 1, 1u);
    if (trace && threadIdx.x == 0) trace[2 * T] = __builtin_amdgcn_s_memrealtime();
}

int finish_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mgx_internal_set_hip_error((int)e); return MGX_ERR_LAUNCH; }
    return MGX_OK;
}

inline bool misaligned(const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

}  // namespace

extern "C" {

static int granule_geometry(const MgxSpec *spec, int64_t batch, int &gpe, uint32_t &inv_gpe, int64_t &ng) {
    if (!spec || batch < 1 || spec->num_agents < 1 || spec->num_agents > MGX_MAX_AGENTS) return MGX_ERR_INVALID_ARGUMENT;
    gpe = (spec->num_agents + 3) / 4;
    ng = batch * gpe;
    if (ng >= ((int64_t)1 << 31)) return MGX_ERR_UNSUPPORTED;
    inv_gpe = (uint32_t)((((uint64_t)1 << 32) + gpe - 1) / gpe);           // ceil(2^32 / gpe): g / gpe exact for g < 2^31, gpe <= 8
    return MGX_OK;
}

int mgx_persistent_post(const MgxSpec *spec, int64_t batch, const int8_t *actions, uint32_t step, uint64_t *action_granules,
                        void *stream) {
    int gpe = 0; uint32_t inv = 0; int64_t ng = 0;
    const int rc = granule_geometry(spec, batch, gpe, inv, ng);
    if (rc) return rc;
    if (!actions || !action_granules || step == 0 || misaligned(action_granules, 8)) return MGX_ERR_INVALID_ARGUMENT;
    if ((spec->num_agents & 3) == 0 && misaligned(actions, 4)) return MGX_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(persistent_post_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       actions, ng, spec->num_agents, gpe, inv, step, action_granules);
    return finish_launch();
}

int mgx_persistent_wait(const uint32_t *done, int32_t waves, uint32_t step, uint32_t *ctrl, int32_t timeout_ms, void *stream) {
    if (!done || !ctrl || waves < 1 || timeout_ms < 1 || timeout_ms > 30000) return MGX_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(persistent_wait_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), done, (int)waves, step, ctrl,
                       (uint32_t)timeout_ms * 100000u);
    return finish_launch();
}

int mgx_persistent_feed(const MgxSpec *spec, int64_t batch, const int8_t *actions, int32_t steps, const MgxPersistent *p,
                        int32_t waves, uint64_t *trace, void *stream) {
    int gpe = 0; uint32_t inv = 0; int64_t ng = 0;
    const int rc = granule_geometry(spec, batch, gpe, inv, ng);
    if (rc) return rc;
    if (!actions || !p || !p->action_granules || !p->done || !p->ctrl || steps < 1 || waves < 1 || p->timeout_ms < 1
        || p->timeout_ms > 30000 || misaligned(trace, 8) || ((spec->num_agents & 3) == 0 && misaligned(actions, 4)))
        return MGX_ERR_INVALID_ARGUMENT;
    const int A_ = spec->num_agents;
    auto kern = (A_ & 3) == 0 ? persistent_feed_kernel<4> : A_ == 2 ? persistent_feed_kernel<2> : A_ == 1 ? persistent_feed_kernel<1>
                                                                                                   : persistent_feed_kernel<0>;
    int64_t nwg = (ng + kFeedSlice - 1) / kFeedSlice;
    if (nwg > kFeedMaxWgs) nwg = kFeedMaxWgs;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(kFeedThreads), 0, static_cast<hipStream_t>(stream), actions, (int)steps, batch,
                       spec->num_agents, gpe, inv, const_cast<uint64_t *>(p->action_granules), p->done, (int)waves, p->ctrl,
                       (uint32_t)p->timeout_ms * 100000u, trace);
    return finish_launch();
}

int mgx_one_hot(const uint8_t *cells, int64_t n_cells, const int32_t *dim_sizes, uint8_t *out, void *stream) {
    if (n_cells < 0 || !dim_sizes) return MGX_ERR_INVALID_ARGUMENT;
    const int d0 = dim_sizes[0], d1 = dim_sizes[1], d2 = dim_sizes[2];
    if (d0 < 1 || d1 < 1 || d2 < 1) return MGX_ERR_INVALID_ARGUMENT;
    const int D = d0 + d1 + d2;
    if (D > 32) return MGX_ERR_UNSUPPORTED;
    if (n_cells == 0) return MGX_OK;
    if (!cells || !out || misaligned(out, 16)) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t chunks = (n_cells + kOhCells - 1) / kOhCells;
    const int64_t blocks = chunks < 256 * 16 ? chunks : 256 * 16;
    const uint32_t inv_D = (uint32_t)(((1ull << 32) + D - 1) / D);
    hipLaunchKernelGGL(one_hot_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), cells,
                       n_cells, d0, d1, d2, inv_D, out);
    return finish_launch();
}

int mgx_full_obs(const MgxSpec *spec, int64_t batch, const MgxCell *grid, const uint8_t *agents, uint8_t *out,
                 void *stream) {
    if (!spec || batch < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->width < 3 || spec->height < 3 || spec->num_agents < 1) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->width > 255 || spec->height > 255) return MGX_ERR_UNSUPPORTED;
    const int HW = spec->width * spec->height;
    if (spec->cell_bytes < 0 || spec->cell_bytes > 3) return MGX_ERR_INVALID_ARGUMENT;
    const int cb = spec->cell_bytes == 1 ? 1 : (spec->cell_bytes == 3 ? 3 : kCellBytes);
    if ((cb + 3) * HW + 2 * 48 > 64 * 1024) return MGX_ERR_UNSUPPORTED;
    if (batch == 0) return MGX_OK;
    if (!grid || !agents || !out || misaligned(agents, 8) || misaligned(grid, 16) || misaligned(out, 16))
        return MGX_ERR_INVALID_ARGUMENT;
    int G = (6 * 1024) / (HW * 3);                                // ~10 KiB of LDS per wavefront
    if (G < 1) G = 1;
    if (G * HW > 65535) G = 65535 / HW;
    while (G > 1 && (batch + G - 1) / G < 4096) G = (G + 1) / 2;  // small batches: spread over the chip
    const int in_buf = (G * HW * cb + 15 + 16 + 15) & ~15;                 // skew + over-read pad
    const int out_buf = (G * HW * 3 + 15 + 16 + 15) & ~15;
    const int wave_lds = in_buf + out_buf;
    int wpb = 4;
    while (wpb > 1 && wpb * wave_lds > 64 * 1024) wpb >>= 1;
    const int64_t nwaves = (batch + G - 1) / G;
    const int64_t blocks = (nwaves + wpb - 1) / wpb;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    const uint32_t inv_W = (uint32_t)(((1ull << 32) + spec->width - 1) / spec->width);
    const uint32_t inv_H = (uint32_t)(((1ull << 32) + spec->height - 1) / spec->height);
    const uint32_t hw = (uint32_t)HW;
    const uint32_t inv_HW = (uint32_t)(((1ull << 32) + hw - 1) / hw);
    auto *kern = cb == 1 ? full_obs_kernel<1> : cb == 3 ? full_obs_kernel<3> : full_obs_kernel<2>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * wpb), (size_t)(wpb * wave_lds),
                       static_cast<hipStream_t>(stream), spec->width, spec->height, spec->num_agents, G, wave_lds, in_buf, inv_W,
                       inv_H, inv_HW, batch, reinterpret_cast<const uint8_t *>(grid), agents, out);
    return finish_launch();
}

int mgx_pack_grid(const uint8_t *cells3, int64_t n_cells, MgxCell *packed, int32_t *bad, void *stream) {
    if (n_cells < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (n_cells == 0) return MGX_OK;
    if (!cells3 || !packed || misaligned(packed, 2) || misaligned(bad, 4)) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t blocks = (n_cells + 2047) / 2048;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((pack_grid_kernel<uint16_t, false>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), cells3,
                       n_cells, packed, bad, 0, 0);
    return finish_launch();
}

int mgx_pack_grid_env(const uint8_t *cells3, int64_t batch, int32_t height, int32_t width, MgxCell *packed, int32_t *bad,
                      void *stream) {
    if (batch < 0 || height < 3 || width < 3 || height > 255 || width > 255) return MGX_ERR_INVALID_ARGUMENT;
    if (batch == 0) return MGX_OK;
    const int64_t n_cells = batch * height * width;
    if (!cells3 || !packed || misaligned(packed, 2) || misaligned(bad, 4)) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t blocks = (n_cells + 2047) / 2048;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((pack_grid_kernel<uint16_t, false>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), cells3,
                       n_cells, packed, bad, (int)width, (int)height);
    return finish_launch();
}

int mgx_pack_grid8_env(const uint8_t *cells3, int64_t batch, int32_t height, int32_t width, MgxCell8 *packed, int32_t *bad,
                       void *stream) {
    if (batch < 0 || height < 3 || width < 3 || height > 255 || width > 255) return MGX_ERR_INVALID_ARGUMENT;
    if (batch == 0) return MGX_OK;
    const int64_t n_cells = batch * height * width;
    if (!cells3 || !packed || misaligned(bad, 4)) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t blocks = (n_cells + 2047) / 2048;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((pack_grid_kernel<uint8_t, true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), cells3,
                       n_cells, packed, bad, (int)width, (int)height);
    return finish_launch();
}

int mgx_unpack_grid8(const MgxCell8 *packed, int64_t n_cells, uint8_t *cells3, void *stream) {
    if (n_cells < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (n_cells == 0) return MGX_OK;
    if (!cells3 || !packed) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t blocks = (n_cells + 2047) / 2048;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(unpack_grid8_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), packed, n_cells, cells3);
    return finish_launch();
}

int mgx_check_grid(const MgxSpec *spec, int64_t batch, const MgxCell *grid, const uint8_t *agents, int32_t *bad, void *stream) {
    if (!spec || batch < 0 || spec->width < 3 || spec->height < 3 || spec->width > 255 || spec->height > 255
        || spec->num_agents < 1 || spec->num_agents > MGX_MAX_AGENTS)
        return MGX_ERR_INVALID_ARGUMENT;
    if (batch == 0) return MGX_OK;
    if (spec->cell_bytes < 0 || spec->cell_bytes > 3) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->cell_bytes == 3) return MGX_ERR_UNSUPPORTED;           // (byte grids are checked where they are packed: MgxStepArgs.grid_bad)
    const bool c8 = spec->cell_bytes == 1;
    if (!grid || !bad || (!c8 && misaligned(grid, 2)) || misaligned(bad, 4)) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t n_cells = batch * spec->height * spec->width, n_rows = agents ? batch * spec->num_agents : 0;
    const int64_t work = std::max((n_cells + 7) / 8, n_rows);
    const int64_t blocks = (work + 255) / 256;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    if (c8)
        hipLaunchKernelGGL((check_grid_kernel<uint8_t, true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           reinterpret_cast<const uint8_t *>(grid), n_cells, (int)spec->width, (int)spec->height, agents, n_rows,
                           (int)spec->num_agents, bad);
    else
        hipLaunchKernelGGL((check_grid_kernel<uint16_t, false>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), grid,
                           n_cells, (int)spec->width, (int)spec->height, agents, n_rows, (int)spec->num_agents, bad);
    return finish_launch();
}

int mgx_unpack_grid(const MgxCell *packed, int64_t n_cells, uint8_t *cells3, void *stream) {
    if (n_cells < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (n_cells == 0) return MGX_OK;
    if (!cells3 || !packed || misaligned(packed, 2)) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t blocks = (n_cells + 2047) / 2048;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(unpack_grid_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), packed, n_cells,
                       cells3);
    return finish_launch();
}

int mgx_reset_done(const MgxSpec *spec, int64_t batch, int64_t first_env, int32_t pool_size, const MgxCell *pool_grid_c,
                   const uint8_t *pool_agents, const uint8_t *pool_aux, MgxCell *grid_c, uint8_t *agents,
                   int32_t *step_count, uint8_t *aux, int32_t *episode, uint8_t *was_reset, void *stream) {
    if (!spec || batch < 0 || pool_size < 1 || first_env < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (batch == 0) return MGX_OK;
    const uint8_t *pool_grid = reinterpret_cast<const uint8_t *>(pool_grid_c);
    uint8_t *grid = reinterpret_cast<uint8_t *>(grid_c);
    if (!pool_grid || !pool_agents || !grid || !agents || !step_count || !episode) return MGX_ERR_INVALID_ARGUMENT;
    if (misaligned(agents, 8) || misaligned(pool_agents, 8) || misaligned(aux, 16) || misaligned(pool_aux, 16)
        || ((spec->cell_bytes == 0 || spec->cell_bytes == 2) && (misaligned(grid, 2) || misaligned(pool_grid, 2))))
        return MGX_ERR_INVALID_ARGUMENT;
    if (spec->cell_bytes < 0 || spec->cell_bytes > 3) return MGX_ERR_INVALID_ARGUMENT;
    const int HW3 = spec->width * spec->height * (spec->cell_bytes == 1 ? 1 : (spec->cell_bytes == 3 ? 3 : kCellBytes));   // bytes of one env's grid
    const int64_t blocks = (batch + 255) / 256;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    // widest copy unit that divides the layout size and the base addresses
    int unit = 16;
    while (unit > 1 && (HW3 % unit || misaligned(grid, unit) || misaligned(pool_grid, unit))) unit >>= 1;
    const int units = HW3 / unit;
    const uint32_t inv_units = (uint32_t)(((1ull << 32) + units - 1) / units);
    const uint32_t inv_A = (uint32_t)(((1ull << 32) + spec->num_agents - 1) / spec->num_agents);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define MGX_RESET(T)                                                                                                   \
    hipLaunchKernelGGL(reset_done_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, HW3, spec->num_agents,           \
                       spec->max_steps, batch, first_env, pool_size, inv_units, inv_A, pool_grid, pool_agents, pool_aux, grid, \
                       agents, step_count, aux, episode, was_reset)
    switch (unit) {
    case 16: MGX_RESET(uint4); break;
    case 8:  MGX_RESET(uint64_t); break;
    case 4:  MGX_RESET(uint32_t); break;
    case 2:  MGX_RESET(uint16_t); break;
    default: MGX_RESET(uint8_t); break;
    }
#undef MGX_RESET
    return finish_launch();
}

}  // extern "C"
