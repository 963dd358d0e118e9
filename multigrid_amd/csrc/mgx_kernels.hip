// mgx_kernels.hip -- gfx950 (MI355X, CDNA4, wave64) kernels + the C ABI of libmgx.so (include/mgx.h).
//
// One fused kernel does a whole MultiGridEnv.step for a chunk of G environments per 256-thread workgroup:
//
//   P0  coalesced 16-byte loads of the chunk's (G,H,W,3) uint8 grid bytes and packed agent rows into LDS
//   P1  one lane per env: step_count += 1, PCG64 visiting order, handle_actions on the LDS tile (dirty cells
//       are written straight back to HBM, 3 bytes each), agent overlay for rendering, env post-step hook,
//       reward / terminated / truncated / rng / step_count written out          (multigrid/base.py:303-476)
//   P2  one WAVEFRONT per agent view, one lane per view cell: rotate-to-facing gather from the LDS tile,
//       out-of-bounds -> wall, own cell -> carried object, see-behind bit per lane, __ballot -> 64-bit row
//       masks, cells staged in LDS in image order                              (multigrid/utils/obs.py:130-233)
//   P3  one lane per view: bit-parallel line-of-sight flood on the ballot masks (carry-propagation closed
//       form of the sequential sweeps)                                         (multigrid/utils/obs.py:235-273)
//   P4  wavefront per view again: cells whose visibility bit is clear become UNSEEN (obs.py:95-100)
//   P5  flat, dword-coalesced store of the chunk's (G,A,v,v,3) observation bytes, agent rows and directions
//
// Pure integer / byte work: no MFMA.  The bound is HBM bytes, so the design goals are: every HBM
// byte touched once, 16-byte loads, 4-byte coalesced stores, and as few VALU instructions per view cell as
// possible.  Workgroups touch disjoint memory, so the blockIdx -> XCD mapping needs no swizzle.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "mgx_rules.h"

namespace {

using namespace mgx;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

struct KernelArgs {
    MgxSpec sp;
    int64_t batch;
    uint8_t *grid;
    uint8_t *agents;
    uint64_t *rng;
    int32_t *step_count;
    const int8_t *actions;
    const uint8_t *target;
    uint8_t *obs;
    uint8_t *dir;
    double *reward;
    uint8_t *terminated;
    uint8_t *truncated;
    int32_t *err;
    int32_t G;          // envs per workgroup
    // LDS carve (byte offsets, all 16-byte aligned)
    int32_t off_tile, off_ag, off_stage, off_sb, off_vis, off_ord, off_rnd;
};

struct LdsPlan {
    int32_t off_tile, off_ag, off_stage, off_sb, off_vis, off_ord, off_rnd, total;
};

inline int align16(int x) { return (x + 15) & ~15; }

LdsPlan plan_lds(const MgxSpec &sp, int G) {
    const int V = sp.view_size, A = sp.num_agents;
    const int nw = (V * V + 63) / 64;
    const int nv = G * A;
    LdsPlan p;
    int o = 0;
    p.off_tile = o;  o = align16(o + G * sp.height * sp.width * 3 + 16 + 16);   // head misalignment + tail vector
    p.off_ag = o;    o = align16(o + nv * MGX_AGENT_STRIDE);
    p.off_stage = o; o = align16(o + nv * V * V * 4 + 4);                        // +1 cell read by the packer
    p.off_sb = o;    o = align16(o + nv * nw * 8);
    p.off_vis = o;   o = align16(o + nv * nw * 8);
    p.off_ord = o;   o = align16(o + nv);
    p.off_rnd = o;   o = align16(o + nv * 8);
    p.total = o;
    return p;
}

// Envs per workgroup: enough views (G*A >= 64) to fill the one-lane-per-view and one-lane-per-env phases,
// bounded by an LDS budget that still lets several workgroups share a CU (160 KiB LDS per CU).
int choose_G(const MgxSpec &sp, int64_t batch) {
    const int A = sp.num_agents;
    int G = (64 + A - 1) / A;
    if (G < 1) G = 1;
    if (G > 64) G = 64;
    while (G > 1 && plan_lds(sp, G).total > 40 * 1024) G = (G + 1) / 2;
    // small batches: keep at least ~one workgroup per CU
    while (G > 1 && (batch + G - 1) / G < 256 && G * A > 16) G = (G + 1) / 2;
    return G;
}

template <int V, bool DO_STEP>
__global__ __launch_bounds__(kThreads) void mgx_fused_kernel(const KernelArgs a) {
    constexpr int V2 = V * V;
    constexpr int NW = (V2 + 63) / 64;          // 64-bit mask words per view
    constexpr int NIT = NW;                      // wave passes per view
    extern __shared__ __align__(16) uint8_t lds[];

    const MgxSpec &sp = a.sp;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = sp.width, H = sp.height, A = sp.num_agents;
    const int HW3 = H * W * 3;
    const int64_t e0 = (int64_t)blockIdx.x * a.G;
    const int Gc = (int)min((int64_t)a.G, a.batch - e0);     // envs in this chunk
    const int NVc = Gc * A;                                   // views in this chunk

    // ------------------------------------------------------------------ P0: HBM -> LDS
    const int64_t g0 = e0 * HW3, g1 = g0 + (int64_t)Gc * HW3;       // chunk byte range in `grid`
    const int64_t gtotal = a.batch * (int64_t)HW3;
    const int64_t ga = g0 & ~(int64_t)15;
    uint8_t *tile_raw = lds + a.off_tile;                            // holds global bytes [ga, ...)
    uint8_t *tile = tile_raw + (int)(g0 - ga);                       // env e's cells at tile + e*HW3
    for (int64_t vo = ga + 16 * tid; vo < g1; vo += 16 * kThreads) {
        uint8_t *dst = tile_raw + (int)(vo - ga);
        if (vo + 16 <= gtotal) {
            *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(a.grid + vo);
        } else {
            for (int k = 0; k < 16 && vo + k < gtotal; ++k) dst[k] = a.grid[vo + k];
        }
    }
    uint8_t *ag = lds + a.off_ag;
    {
        const uint64_t *src = reinterpret_cast<const uint64_t *>(a.agents) + e0 * A;
        uint64_t *dst = reinterpret_cast<uint64_t *>(ag);
        for (int t = tid; t < NVc; t += kThreads) dst[t] = src[t];
    }
    __syncthreads();

    // ------------------------------------------------------------------ P1: one lane per env
    for (int e = tid; e < Gc; e += kThreads) {
        const int64_t b = e0 + e;
        uint8_t *etile = tile + e * HW3;
        uint8_t *eag = ag + e * A * MGX_AGENT_STRIDE;
        if (DO_STEP) {
            const int32_t sc = a.step_count[b] + 1;                             // base.py:333
            a.step_count[b] = sc;
            uint64_t r4[4] = {0, 0, 0, 0};
            if (A > 1) { r4[0] = a.rng[b * 4 + 0]; r4[1] = a.rng[b * 4 + 1]; r4[2] = a.rng[b * 4 + 2]; r4[3] = a.rng[b * 4 + 3]; }
            double *rew = a.reward + b * A;
            uint8_t *ggrid = a.grid + b * HW3;
            auto dirty = [=](int off) {
                ggrid[off] = etile[off]; ggrid[off + 1] = etile[off + 1]; ggrid[off + 2] = etile[off + 2];
            };
            const int rc = handle_actions(sp, etile, eag, r4, sc, a.actions + b * A,
                                          lds + a.off_ord + e * A,
                                          reinterpret_cast<uint64_t *>(lds + a.off_rnd) + e * A, rew, dirty);
            if (A > 1) { a.rng[b * 4 + 0] = r4[0]; a.rng[b * 4 + 1] = r4[1]; }
            if (rc != 0 && a.err) { atomicAdd(a.err, 1); atomicMin(a.err + 1, (int32_t)min(b, (int64_t)INT_MAX)); }
            overlay_agents(sp, etile, eag);                                     // uses pre-hook `terminated` (Q2)
            post_step_hook(sp, eag, a.target ? a.target + b * 4 : eag, sc, rew);
            for (int i = 0; i < A; ++i) a.terminated[b * A + i] = eag[i * MGX_AGENT_STRIDE + AG_TERM];  // base.py:338
            a.truncated[b] = (uint8_t)(sc >= sp.max_steps);                      // base.py:339
        } else {
            overlay_agents(sp, etile, eag);
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ P2: one wavefront per view
    uint32_t *stage = reinterpret_cast<uint32_t *>(lds + a.off_stage);       // [view][i*V + j] packed cells
    uint64_t *sbw = reinterpret_cast<uint64_t *>(lds + a.off_sb);            // [view][NW]
    uint64_t *visw = reinterpret_cast<uint64_t *>(lds + a.off_vis);          // [view][NW]
    const uint32_t tile_addr = (uint32_t)(a.off_tile + (int)(g0 - ga));      // LDS byte address of env 0 cell 0

    // lane constants: cell k = lane + 64*it  <->  image[i][j], k = j*V + i (depth-row major, so each ballot
    // word holds whole visibility rows); lateral offset la = i - V/2, forward distance fw = V-1-j.
    int la[NIT], fw[NIT], q[NIT];
    bool act[NIT], own[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = lane + 64 * it;
        const int j = k / V, i = k - j * V;
        act[it] = k < V2;
        la[it] = i - V / 2;
        fw[it] = V - 1 - j;
        q[it] = i * V + j;
        own[it] = (i == V / 2) && (j == V - 1);
    }

    for (int view = wave; view < NVc; view += kWaves) {
        const int e = view / A;
        const uint2 s2 = *reinterpret_cast<const uint2 *>(ag + view * MGX_AGENT_STRIDE);
        const uint32_t s_lo = __builtin_amdgcn_readfirstlane(s2.x), s_hi = __builtin_amdgcn_readfirstlane(s2.y);
        const int d = (s_lo >> 8) & 0xff, x = (s_lo >> 16) & 0xff, y = s_lo >> 24;
        const uint32_t carry = s_hi >> 8;                                   // type | color<<8 | state<<16
        const int dx = dir_dx(d), dy = dir_dy(d);
        const uint32_t ebase = tile_addr + (uint32_t)(e * HW3);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            // world cell seen at image[i][j]: pos + fw*forward + la*right, right = (-dy, dx)   (obs.py:182-202)
            const int wx = x + fw[it] * dx - la[it] * dy;
            const int wy = y + fw[it] * dy + la[it] * dx;
            const bool inb = ((unsigned)wx < (unsigned)W) & ((unsigned)wy < (unsigned)H);
            const uint32_t addr = ebase + (inb ? (uint32_t)((wy * W + wx) * 3) : 0u);
            const uint32_t *p = reinterpret_cast<const uint32_t *>(lds + (addr & ~3u));
            uint32_t c = __builtin_amdgcn_alignbyte(p[1], p[0], addr & 3u) & 0xffffffu;
            c = inb ? c : CELL_WALL;                                            // obs.py:199-202
            c = own[it] ? carry : c;                                            // obs.py:207
            const uint64_t m = __ballot(act[it] && see_behind(c));              // obs.py:211-233
            if (lane == 0) sbw[view * NW + it] = m;
            if (act[it]) stage[view * V2 + q[it]] = c;
        }
    }
    __syncthreads();

    if (!sp.see_through_walls) {                                                // obs.py:95-100
        // -------------------------------------------------------------- P3: one lane per view
        for (int view = tid; view < NVc; view += kThreads) {
            uint64_t sb[NW], vis[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) sb[k] = sbw[view * NW + k];
            vis_mask<V, NW>(sb, vis);
#pragma unroll
            for (int k = 0; k < NW; ++k) visw[view * NW + k] = vis[k];
        }
        __syncthreads();
        // -------------------------------------------------------------- P4: wavefront per view
        for (int view = wave; view < NVc; view += kWaves) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const uint64_t m = visw[view * NW + it];
                if (act[it] && !((m >> lane) & 1)) stage[view * V2 + q[it]] = CELL_UNSEEN;
            }
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ P5: LDS -> HBM, dword-coalesced
    {
        const int64_t o0 = e0 * (int64_t)(A * V2 * 3), o1 = o0 + (int64_t)NVc * V2 * 3;
        const int64_t oa = o0 & ~(int64_t)3;
        for (int64_t D = oa + 4 * tid; D < o1; D += 4 * kThreads) {
            const int64_t lo_b = max(D, o0), hi_b = min(D + 4, o1);
            // bytes [D, D+4) of the obs stream = bytes rel.. of the staged cells, 3 bytes per cell
            const int rel = (int)(lo_b - o0);
            const int q0 = rel / 3, r = rel - q0 * 3;
            const uint32_t c0 = stage[q0], c1 = stage[q0 + 1];
            const uint32_t lo = c0 | (c1 << 24), hi = c1 >> 8;
            const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)r);   // bytes lo_b, lo_b+1, ...
            if (lo_b == D && hi_b == D + 4) {
                *reinterpret_cast<uint32_t *>(a.obs + D) = w;
            } else {
                for (int k = 0; k < (int)(hi_b - lo_b); ++k) a.obs[lo_b + k] = (uint8_t)(w >> (8 * k));
            }
        }
        for (int t = tid; t < NVc; t += kThreads) {
            const uint64_t row = reinterpret_cast<const uint64_t *>(ag)[t];
            if (DO_STEP) reinterpret_cast<uint64_t *>(a.agents)[e0 * A + t] = row;
            if (a.dir) a.dir[e0 * A + t] = (uint8_t)(row >> 8);                   // base.py:359, 372
        }
    }
}

int g_last_hip_error = 0;

template <bool DO_STEP>
int launch(const KernelArgs &ka, int lds_bytes, int64_t nwg, hipStream_t stream) {
    void (*kern)(const KernelArgs) = nullptr;
    switch (ka.sp.view_size) {
    case 3:  kern = mgx_fused_kernel<3, DO_STEP>;  break;
    case 5:  kern = mgx_fused_kernel<5, DO_STEP>;  break;
    case 7:  kern = mgx_fused_kernel<7, DO_STEP>;  break;
    case 9:  kern = mgx_fused_kernel<9, DO_STEP>;  break;
    case 11: kern = mgx_fused_kernel<11, DO_STEP>; break;
    case 13: kern = mgx_fused_kernel<13, DO_STEP>; break;
    case 15: kern = mgx_fused_kernel<15, DO_STEP>; break;
    default: return MGX_ERR_UNSUPPORTED;
    }
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return MGX_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(kThreads), (size_t)lds_bytes, stream, ka);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return MGX_ERR_LAUNCH; }
    return MGX_OK;
}

int check_spec(const MgxSpec *sp, int64_t batch) {
    if (!sp || batch < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (sp->view_size < 3 || !(sp->view_size & 1)) return MGX_ERR_INVALID_ARGUMENT;   // agent.py:78-79
    if (sp->width < 3 || sp->height < 3 || sp->num_agents < 1 || sp->max_steps < 1) return MGX_ERR_INVALID_ARGUMENT;
    if (sp->view_size > MGX_MAX_VIEW || sp->num_agents > MGX_MAX_AGENTS) return MGX_ERR_UNSUPPORTED;
    if (sp->width > 255 || sp->height > 255) return MGX_ERR_UNSUPPORTED;             // positions are uint8
    if (sp->env_kind != MGX_KIND_EMPTY && sp->env_kind != MGX_KIND_BLOCKEDUNLOCKPICKUP) return MGX_ERR_UNSUPPORTED;
    if (plan_lds(*sp, 1).total > 160 * 1024) return MGX_ERR_UNSUPPORTED;
    return MGX_OK;
}

int fill_args(KernelArgs &ka, const MgxSpec *sp, int64_t batch, int &lds_bytes, int64_t &nwg) {
    ka.sp = *sp;
    ka.batch = batch;
    ka.G = choose_G(*sp, batch);
    const LdsPlan p = plan_lds(*sp, ka.G);
    ka.off_tile = p.off_tile; ka.off_ag = p.off_ag; ka.off_stage = p.off_stage; ka.off_sb = p.off_sb;
    ka.off_vis = p.off_vis; ka.off_ord = p.off_ord; ka.off_rnd = p.off_rnd;
    lds_bytes = p.total;
    nwg = (batch + ka.G - 1) / ka.G;
    if (nwg > INT_MAX) return MGX_ERR_UNSUPPORTED;
    return MGX_OK;
}

inline bool misaligned(const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

}  // namespace

extern "C" {

int mgx_abi_version(void) { return MGX_ABI_VERSION; }

const char *mgx_error_string(int code) {
    switch (code) {
    case MGX_OK: return "ok";
    case MGX_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MGX_ERR_UNKNOWN_ACTION: return "unknown action";
    case MGX_ERR_UNSUPPORTED: return "configuration outside compiled limits";
    case MGX_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown error";
    }
}

int mgx_last_hip_error(void) { return g_last_hip_error; }

int mgx_launch_info(const MgxSpec *spec, int64_t batch, MgxLaunchInfo *out) {
    int rc = check_spec(spec, batch);
    if (rc) return rc;
    if (!out) return MGX_ERR_INVALID_ARGUMENT;
    KernelArgs ka{};
    int lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, lds, nwg);
    if (rc) return rc;
    out->envs_per_workgroup = ka.G;
    out->threads_per_workgroup = kThreads;
    out->workgroups = (int32_t)nwg;
    out->lds_bytes = lds;
    return MGX_OK;
}

int mgx_gen_obs(const MgxSpec *spec, int64_t batch, const uint8_t *grid, const uint8_t *agents,
                uint8_t *obs, uint8_t *dir, void *stream) {
    int rc = check_spec(spec, batch);
    if (rc) return rc;
    if (batch == 0) return MGX_OK;
    if (!grid || !agents || !obs) return MGX_ERR_INVALID_ARGUMENT;
    if (misaligned(grid, 16) || misaligned(agents, 8) || misaligned(obs, 4)) return MGX_ERR_INVALID_ARGUMENT;
    KernelArgs ka{};
    int lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, lds, nwg);
    if (rc) return rc;
    ka.grid = const_cast<uint8_t *>(grid);
    ka.agents = const_cast<uint8_t *>(agents);
    ka.obs = obs;
    ka.dir = dir;
    return launch<false>(ka, lds, nwg, static_cast<hipStream_t>(stream));
}

int mgx_step(const MgxSpec *spec, int64_t batch, uint8_t *grid, uint8_t *agents, uint64_t *rng,
             int32_t *step_count, const int8_t *actions, const uint8_t *target,
             uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
             int32_t *err, void *stream) {
    int rc = check_spec(spec, batch);
    if (rc) return rc;
    if (batch == 0) return MGX_OK;
    if (!grid || !agents || !step_count || !actions || !obs || !reward || !terminated || !truncated)
        return MGX_ERR_INVALID_ARGUMENT;
    if (spec->num_agents > 1 && !rng) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->env_kind == MGX_KIND_BLOCKEDUNLOCKPICKUP && !target) return MGX_ERR_INVALID_ARGUMENT;
    if (misaligned(grid, 16) || misaligned(agents, 8) || misaligned(obs, 4) || misaligned(rng, 8)
        || misaligned(reward, 8) || misaligned(step_count, 4) || misaligned(err, 4))
        return MGX_ERR_INVALID_ARGUMENT;
    KernelArgs ka{};
    int lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, lds, nwg);
    if (rc) return rc;
    ka.grid = grid; ka.agents = agents; ka.rng = rng; ka.step_count = step_count; ka.actions = actions;
    ka.target = target; ka.obs = obs; ka.dir = dir; ka.reward = reward; ka.terminated = terminated;
    ka.truncated = truncated; ka.err = err;
    return launch<true>(ka, lds, nwg, static_cast<hipStream_t>(stream));
}

}  // extern "C"
