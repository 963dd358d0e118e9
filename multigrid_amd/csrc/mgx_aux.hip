// mgx_aux.hip -- the kernels either side of the fused step, gfx950.
//
//   mgx_one_hot     OneHotObsWrapper.one_hot        multigrid/wrappers.py:158-190   (HBM write-bound: 3 B in, 21 B out)
//   mgx_full_obs    FullyObsWrapper.observation     multigrid/wrappers.py:48-58     (grid transpose-copy + agent overlay)
//   mgx_reset_done  vector-env auto-reset from a pool of pre-generated layouts (build-defined; the reference has no
//                   batching: its user calls reset() when is_done(), multigrid/base.py:250-301, 534-539)
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "mgx_rules.h"

namespace {

using namespace mgx;

int g_aux_hip_error = 0;

// ---------------------------------------------------------------------------------------------------------------
// one_hot: out[cell][d0 + d1 + d2] = 1 at {x0, d0 + x1, d0 + d1 + x2}.  One thread per 16 output bytes (a vector never
// spans more than two cells when the channel count is >= 16; the general path handles any count).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cell_bits(const uint8_t *x, int64_t cell, int64_t n_cells, int d0, int d1, int d2) {
    if (cell >= n_cells) return 0;
    const uint8_t *p = x + cell * 3;
    uint32_t m = 0;
    if (p[0] < d0) m |= 1u << p[0];
    if (p[1] < d1) m |= 1u << (d0 + p[1]);
    if (p[2] < d2) m |= 1u << (d0 + d1 + p[2]);
    return m;
}

__global__ __launch_bounds__(256) void one_hot_kernel(const uint8_t *__restrict__ x, int64_t n_cells, int d0, int d1, int d2,
                                                      uint8_t *__restrict__ out) {
    const int D = d0 + d1 + d2;                                   // <= 32
    const int64_t total = n_cells * D;
    for (int64_t o = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; o < total;
         o += (int64_t)gridDim.x * blockDim.x * 16) {
        const int64_t c0 = o / D;
        const int k0 = (int)(o - c0 * D);
        // the bits of up to 16 consecutive output bytes, gathered from consecutive cells
        uint32_t bits = 0;
        int have = 0;
        int64_t c = c0;
        int k = k0;
        while (have < 16) {
            const uint32_t m = cell_bits(x, c, n_cells, d0, d1, d2) >> k;
            bits |= (m << have) & 0xffffu;
            have += D - k;
            k = 0;
            ++c;
        }
        uint4 v;                                                  // spread 4 bits -> 4 bytes of 0/1
        v.x = (((bits >> 0) & 0xfu) * 0x00204081u) & 0x01010101u;
        v.y = (((bits >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
        v.z = (((bits >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
        v.w = (((bits >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
        if (o + 16 <= total) {
            *reinterpret_cast<uint4 *>(out + o) = v;
        } else {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int b = 0; o + b < total; ++b) out[o + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// full_obs: img = grid.encode() (a copy of Grid.state, (W,H,3) indexed [x][y]); img[agent.pos] = agent.encode() for
// every agent in index order, terminated or not (wrappers.py:52-54).  One workgroup per env: transpose through LDS.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void full_obs_kernel(int W, int H, int A, int64_t batch, const uint8_t *__restrict__ grid,
                                                       const uint8_t *__restrict__ agents, uint8_t *__restrict__ out) {
    extern __shared__ __align__(16) uint8_t lds[];               // W*H*3 bytes in OUTPUT order [x][y][c]
    const int HW = H * W;
    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        const uint8_t *g = grid + b * HW * 3;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {      // i = y*W + x in the product layout
            const int y = i / W, x = i - y * W;
            uint8_t *d = lds + (x * H + y) * 3;
            d[0] = g[i * 3]; d[1] = g[i * 3 + 1]; d[2] = g[i * 3 + 2];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint64_t *rows = reinterpret_cast<const uint64_t *>(agents) + b * A;
            for (int a = 0; a < A; ++a) {
                const uint64_t r = rows[a];
                const int x = row_x(r), y = row_y(r);
                if (x < W && y < H) store_cell(lds + (x * H + y) * 3, (uint32_t)T_AGENT | ((uint32_t)(r & 0xffffu) << 8));
            }
        }
        __syncthreads();
        uint8_t *o = out + b * HW * 3;
        for (int i = threadIdx.x; i < HW * 3; i += blockDim.x) o[i] = lds[i];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// reset_done: every env whose episode is over (all agents terminated, or step_count >= max_steps: base.py:534-539)
// is re-initialised from layout pool[(global_env + episode * stride) mod K]; step_count := 0, episode += 1.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reset_done_kernel(int HW3, int A, int max_steps, int64_t batch, int64_t first_env,
                                                         int K, const uint8_t *__restrict__ pool_grid,
                                                         const uint8_t *__restrict__ pool_agents,
                                                         const uint8_t *__restrict__ pool_aux, uint8_t *grid,
                                                         uint8_t *agents, int32_t *step_count, uint8_t *aux,
                                                         int32_t *episode, uint8_t *was_reset) {
    __shared__ int s_done, s_layout;
    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        if (threadIdx.x == 0) {
            const uint64_t *rows = reinterpret_cast<const uint64_t *>(agents) + b * A;
            bool all_term = true;
            for (int a = 0; a < A; ++a) all_term &= row_term(rows[a]);
            const int done = all_term || step_count[b] >= max_steps;
            s_done = done;
            if (done) {
                const int ep = episode[b];
                // a fixed odd stride walks the whole pool before repeating; depends only on the GLOBAL env index
                s_layout = (int)((uint64_t)(first_env + b + (int64_t)ep * 7919) % (uint64_t)K);
                episode[b] = ep + 1;
                step_count[b] = 0;
            }
            if (was_reset) was_reset[b] = (uint8_t)done;
        }
        __syncthreads();
        if (s_done) {
            const uint8_t *sg = pool_grid + (int64_t)s_layout * HW3;
            uint8_t *dg = grid + b * HW3;
            for (int i = threadIdx.x; i < HW3; i += blockDim.x) dg[i] = sg[i];
            for (int i = threadIdx.x; i < A * MGX_AGENT_STRIDE; i += blockDim.x)
                agents[b * A * MGX_AGENT_STRIDE + i] = pool_agents[(int64_t)s_layout * A * MGX_AGENT_STRIDE + i];
            if (aux && pool_aux && threadIdx.x < MGX_AUX_BYTES)
                aux[b * MGX_AUX_BYTES + threadIdx.x] = pool_aux[(int64_t)s_layout * MGX_AUX_BYTES + threadIdx.x];
        }
        __syncthreads();
    }
}

int finish_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_aux_hip_error = (int)e; return MGX_ERR_LAUNCH; }
    return MGX_OK;
}

}  // namespace

extern "C" {

int mgx_one_hot(const uint8_t *cells, int64_t n_cells, const int32_t *dim_sizes, uint8_t *out, void *stream) {
    if (n_cells < 0 || !dim_sizes) return MGX_ERR_INVALID_ARGUMENT;
    const int d0 = dim_sizes[0], d1 = dim_sizes[1], d2 = dim_sizes[2];
    if (d0 < 1 || d1 < 1 || d2 < 1) return MGX_ERR_INVALID_ARGUMENT;
    if (d0 + d1 + d2 > 32) return MGX_ERR_UNSUPPORTED;
    if (n_cells == 0) return MGX_OK;
    if (!cells || !out || (reinterpret_cast<uintptr_t>(out) & 15)) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t vectors = (n_cells * (d0 + d1 + d2) + 15) / 16;
    int64_t blocks = (vectors + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(one_hot_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), cells,
                       n_cells, d0, d1, d2, out);
    return finish_launch();
}

int mgx_full_obs(const MgxSpec *spec, int64_t batch, const uint8_t *grid, const uint8_t *agents, uint8_t *out,
                 void *stream) {
    if (!spec || batch < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->width < 3 || spec->height < 3 || spec->num_agents < 1) return MGX_ERR_INVALID_ARGUMENT;
    const int lds = spec->width * spec->height * 3;
    if (lds > 64 * 1024) return MGX_ERR_UNSUPPORTED;
    if (batch == 0) return MGX_OK;
    if (!grid || !agents || !out || (reinterpret_cast<uintptr_t>(agents) & 7)) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t blocks = batch < 256 * 16 ? batch : 256 * 16;
    hipLaunchKernelGGL(full_obs_kernel, dim3((unsigned)blocks), dim3(256), (size_t)lds, static_cast<hipStream_t>(stream),
                       spec->width, spec->height, spec->num_agents, batch, grid, agents, out);
    return finish_launch();
}

int mgx_reset_done(const MgxSpec *spec, int64_t batch, int64_t first_env, int32_t pool_size, const uint8_t *pool_grid,
                   const uint8_t *pool_agents, const uint8_t *pool_aux, uint8_t *grid, uint8_t *agents,
                   int32_t *step_count, uint8_t *aux, int32_t *episode, uint8_t *was_reset, void *stream) {
    if (!spec || batch < 0 || pool_size < 1 || first_env < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (batch == 0) return MGX_OK;
    if (!pool_grid || !pool_agents || !grid || !agents || !step_count || !episode) return MGX_ERR_INVALID_ARGUMENT;
    if (reinterpret_cast<uintptr_t>(agents) & 7) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t blocks = batch < 256 * 32 ? batch : 256 * 32;
    hipLaunchKernelGGL(reset_done_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       spec->width * spec->height * 3, spec->num_agents, spec->max_steps, batch, first_env, pool_size,
                       pool_grid, pool_agents, pool_aux, grid, agents, step_count, aux, episode, was_reset);
    return finish_launch();
}

}  // extern "C"
