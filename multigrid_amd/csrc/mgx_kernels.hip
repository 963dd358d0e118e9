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
    int32_t dbg;        // debug: bit p set = skip phase p (profiling only, mgx_debug_skip_phases)
    // LDS carve (byte offsets, all 16-byte aligned)
    int32_t off_tile, off_rows, off_act, off_rng, off_rnd, off_ord, off_rew, off_rec, off_mask, off_stage;
};

struct LdsPlan {
    int32_t off_tile, off_rows, off_act, off_rng, off_rnd, off_ord, off_rew, off_rec, off_mask, off_stage, total;
};

inline int align16(int x) { return (x + 15) & ~15; }

// per-view record written by P1d and read (broadcast) by the view's wavefront in P2
struct ViewRec { int32_t origin, stepF, stepL; uint32_t carry; };     // 16 bytes

LdsPlan plan_lds(const MgxSpec &sp, int G) {
    const int V = sp.view_size, A = sp.num_agents;
    const int nw = (V * V + 63) / 64;
    const int nv = G * A;
    LdsPlan p;
    int o = 0;
    p.off_tile = o;  o = align16(o + G * sp.height * sp.width * 3 + 16 + 16);   // head misalignment + tail vector
    p.off_rows = o;  o = align16(o + nv * MGX_AGENT_STRIDE);
    p.off_act = o;   o = align16(o + nv);
    p.off_rng = o;   o = align16(o + G * 32);
    p.off_rnd = o;   o = align16(o + nv * 8);
    p.off_ord = o;   o = align16(o + nv);
    p.off_rew = o;   o = align16(o + nv * 8);
    p.off_rec = o;   o = align16(o + nv * (int)sizeof(ViewRec));
    p.off_mask = o;  o = align16(o + nv * nw * 8 * 2);                           // [view][NW] in-bounds, then see-behind/vis
    p.off_stage = o; o = align16(o + nv * V * V * 4 + 32);                       // + cells over-read by the packer
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

__device__ const JumpTable kJump{};

template <int V, bool DO_STEP>
__global__ __launch_bounds__(kThreads) void mgx_fused_kernel(const KernelArgs a) {
    constexpr int V2 = V * V;
    constexpr int NW = (V2 + 63) / 64;          // 64-bit mask words per view
    constexpr int NIT = NW;                      // wave passes per view
    extern __shared__ __align__(16) uint8_t lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.sp.width, H = a.sp.height, A = a.sp.num_agents;
    const int HW3 = H * W * 3;
    const int64_t e0 = (int64_t)blockIdx.x * a.G;
    const int Gc = (int)min((int64_t)a.G, a.batch - e0);     // envs in this chunk
    const int NVc = Gc * A;                                   // views in this chunk
    const int64_t v0 = e0 * A;                                // first (env, agent) row of the chunk

    uint64_t *rows = reinterpret_cast<uint64_t *>(lds + a.off_rows);          // [view] packed agent rows
    int8_t *acts = reinterpret_cast<int8_t *>(lds + a.off_act);               // [view]
    uint64_t *rngs = reinterpret_cast<uint64_t *>(lds + a.off_rng);           // [env][4]
    uint64_t *rnd = reinterpret_cast<uint64_t *>(lds + a.off_rnd);            // [view] 53-bit draws
    uint8_t *ord = lds + a.off_ord;                                            // [view] visiting order per env
    double *rew = reinterpret_cast<double *>(lds + a.off_rew);                // [view]
    ViewRec *rec = reinterpret_cast<ViewRec *>(lds + a.off_rec);              // [view]
    uint64_t *inbw = reinterpret_cast<uint64_t *>(lds + a.off_mask);          // [view][NW] in-bounds lanes
    uint64_t *sbw = inbw + (size_t)a.G * A * NW;                               // [view][NW] see-behind, then visible
    uint32_t *stage = reinterpret_cast<uint32_t *>(lds + a.off_stage);        // [view][i*V + j] packed cells

    // ------------------------------------------------------------------ P0: HBM -> LDS, all loads in flight at once
    const int64_t g0 = e0 * HW3, g1 = g0 + (int64_t)Gc * HW3;       // chunk byte range in `grid`
    const int64_t gtotal = a.batch * (int64_t)HW3;
    const int64_t ga = g0 & ~(int64_t)15;
    uint8_t *tile_raw = lds + a.off_tile;                            // holds global bytes [ga, ...)
    const int tile_skew = (int)(g0 - ga);
    uint8_t *tile = tile_raw + tile_skew;                            // env e's cells at tile + e*HW3
    if (!(a.dbg & 1)) {
        constexpr int U = 4;
        for (int64_t base = ga + 16 * tid; base < g1; base += (int64_t)16 * kThreads * U) {
            uint4 v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t vo = base + (int64_t)16 * kThreads * u;
                ok[u] = vo < g1 && vo + 16 <= gtotal;
                v[u] = make_uint4(0, 0, 0, 0);
                if (ok[u]) v[u] = *reinterpret_cast<const uint4 *>(a.grid + vo);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t vo = base + (int64_t)16 * kThreads * u;
                if (ok[u]) *reinterpret_cast<uint4 *>(tile_raw + (int)(vo - ga)) = v[u];
            }
        }
        if (g1 == gtotal && (gtotal & 15)) {                      // last, partial 16-byte vector of the tensor
            const int64_t t0 = gtotal & ~(int64_t)15;
            for (int k = tid; k < (int)(gtotal & 15); k += kThreads) tile_raw[(int)(t0 - ga) + k] = a.grid[t0 + k];
        }
    }
    for (int t = tid; t < NVc; t += kThreads) {
        rows[t] = reinterpret_cast<const uint64_t *>(a.agents)[v0 + t];
        rew[t] = 0.0;                                                            // base.py:393
        if (DO_STEP) acts[t] = a.actions[v0 + t];
    }
    if (DO_STEP && A > 1)
        for (int t = tid; t < Gc * 4; t += kThreads) rngs[t] = a.rng[e0 * 4 + t];
    __syncthreads();

    const StepCfg cf = make_cfg(a.sp);
    if (DO_STEP && !(a.dbg & 2)) {
        if (A > 1) {
            // -------------------------------------------------------------- P1a: one lane per (env, agent): its draw
            for (int t = tid; t < NVc; t += kThreads) {
                const int e = t / A, ai = t - e * A;
                uint64_t s_lo, s_hi;
                rnd[t] = pcg64_draw_at(rngs + e * 4, kJump.w[ai + 1], s_lo, s_hi);    // base.py:399
                if (ai == A - 1) { a.rng[(e0 + e) * 4 + 0] = s_lo; a.rng[(e0 + e) * 4 + 1] = s_hi; }
            }
            __syncthreads();
            // -------------------------------------------------------------- P1b: argsort by ranking
            for (int t = tid; t < NVc; t += kThreads) {
                const int e = t / A, ai = t - e * A;
                ord[e * A + draw_rank(rnd + e * A, A, ai)] = (uint8_t)ai;
            }
            __syncthreads();
        }
        // ------------------------------------------------------------------ P1c: one lane per env, LDS only
        for (int e = tid; e < Gc; e += kThreads) {
            const int64_t b = e0 + e;
            uint8_t *etile = tile + e * HW3;
            uint64_t *erows = rows + e * A;
            double *erew = rew + e * A;
            const int32_t sc = a.step_count[b] + 1;                              // base.py:333
            a.step_count[b] = sc;
            uint8_t *ggrid = a.grid + b * HW3;
            auto dirty = [=](int off) {
                ggrid[off] = etile[off]; ggrid[off + 1] = etile[off + 1]; ggrid[off + 2] = etile[off + 2];
            };
            const int rc = handle_actions(cf, etile, erows, acts + e * A, ord + e * A, erew, sc, dirty);
            if (rc != 0 && a.err) { atomicAdd(a.err, 1); atomicMin(a.err + 1, (int32_t)min(b, (int64_t)INT_MAX)); }
            overlay_agents(cf, etile, erows);                                    // uses pre-hook `terminated` (Q2)
            post_step_hook(cf, a.sp.env_kind, erows, a.target ? a.target + b * 4 : etile, sc, erew);
            a.truncated[b] = (uint8_t)(sc >= cf.max_steps);                      // base.py:339
        }
    } else {
        for (int e = tid; e < Gc; e += kThreads) overlay_agents(cf, tile + e * HW3, rows + e * A);
    }
    __syncthreads();

    // ------------------------------------------------------------------ P1d: one lane per view: geometry + outputs
    const uint32_t tile_addr = (uint32_t)(a.off_tile + tile_skew);           // LDS byte address of env 0 cell 0
    for (int t = tid; t < NVc; t += kThreads) {
        const int e = t / A;
        const uint64_t row = rows[t];
        const ViewGeom g = view_geom<V>(W, H, row_x(row), row_y(row), row_dir(row));
        ViewRec r;
        r.origin = (int32_t)tile_addr + e * HW3 + g.origin;
        r.stepF = g.stepF; r.stepL = g.stepL; r.carry = row_carry(row);
        rec[t] = r;
        uint64_t m[NW];
        inbounds_mask<V, NW>(g, m);
#pragma unroll
        for (int k = 0; k < NW; ++k) inbw[t * NW + k] = m[k];
        if (DO_STEP) {
            reinterpret_cast<uint64_t *>(a.agents)[v0 + t] = row;
            a.reward[v0 + t] = rew[t];
            a.terminated[v0 + t] = (uint8_t)row_term(row);                       // base.py:338 (+ env hook)
        }
        if (a.dir) a.dir[v0 + t] = (uint8_t)row_dir(row);                        // base.py:359, 372
    }
    __syncthreads();

    // ------------------------------------------------------------------ P2: one wavefront per view
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

    if (!(a.dbg & 4))
    for (int view = wave; view < NVc; view += kWaves) {
        const ViewRec r = rec[view];                                            // broadcast read
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const uint64_t inb_m = inbw[view * NW + it];
            // (readfirstlane returns a signed int: cast before widening)
            const uint64_t inb_s = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(inb_m >> 32)) << 32)
                                 | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)inb_m);
            const bool inb = (inb_s >> lane) & 1;
            // world cell seen at image[i][j]: pos + fw*forward + la*right                     (obs.py:182-202)
            uint32_t addr = (uint32_t)(r.origin + fw[it] * r.stepF + la[it] * r.stepL);
            addr = inb ? addr : (uint32_t)r.origin;
            const uint32_t *p = reinterpret_cast<const uint32_t *>(lds + (addr & ~3u));
            uint32_t c = __builtin_amdgcn_alignbyte(p[1], p[0], addr & 3u) & 0xffffffu;
            c = inb ? c : CELL_WALL;                                            // obs.py:199-202
            c = own[it] ? r.carry : c;                                          // obs.py:207
            const uint64_t m = __ballot(act[it] && see_behind(c));              // obs.py:211-233
            if (lane == 0) sbw[view * NW + it] = m;
            if (act[it]) stage[view * V2 + q[it]] = c;
        }
    }
    __syncthreads();

    if (!a.sp.see_through_walls) {                                              // obs.py:95-100
        // -------------------------------------------------------------- P3: one lane per view
        if (!(a.dbg & 8))
        for (int view = tid; view < NVc; view += kThreads) {
            uint64_t sb[NW], vis[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) sb[k] = sbw[view * NW + k];
            vis_mask<V, NW>(sb, vis);
#pragma unroll
            for (int k = 0; k < NW; ++k) sbw[view * NW + k] = vis[k];
        }
        __syncthreads();
        // -------------------------------------------------------------- P4: wavefront per view
        if (!(a.dbg & 16))
        for (int view = wave; view < NVc; view += kWaves) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const uint64_t m = sbw[view * NW + it];
                if (act[it] && !((m >> lane) & 1)) stage[view * V2 + q[it]] = CELL_UNSEEN;
            }
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ P5: LDS -> HBM, 16-byte coalesced stores
    if (!(a.dbg & 32)) {
        const int64_t o0 = v0 * (int64_t)(V2 * 3), o1 = o0 + (int64_t)NVc * V2 * 3;
        const int64_t oa = o0 & ~(int64_t)15;
        for (int64_t D = oa + 16 * tid; D < o1; D += 16 * kThreads) {
            if (D >= o0 && D + 16 <= o1) {
                // bytes [D, D+16) of the obs stream = bytes rel.. of the staged cells, 3 bytes per cell
                const int rel = (int)(D - o0);
                const int q0 = rel / 3, r = rel - q0 * 3;
                const uint32_t *sp = stage + q0;
                const uint32_t c0 = sp[0], c1 = sp[1], c2 = sp[2], c3 = sp[3], c4 = sp[4], c5 = sp[5], c6 = sp[6];
                // the cells' 3-byte encodings as a byte stream, in dwords
                const uint32_t w0 = c0 | (c1 << 24), w1 = (c1 >> 8) | (c2 << 16), w2 = (c2 >> 16) | (c3 << 8);
                const uint32_t w3 = c4 | (c5 << 24), w4 = (c5 >> 8) | (c6 << 16);
                uint4 out;
                out.x = __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)r);
                out.y = __builtin_amdgcn_alignbyte(w2, w1, (uint32_t)r);
                out.z = __builtin_amdgcn_alignbyte(w3, w2, (uint32_t)r);
                out.w = __builtin_amdgcn_alignbyte(w4, w3, (uint32_t)r);
                *reinterpret_cast<uint4 *>(a.obs + D) = out;
            } else {
                const int64_t lo_b = max(D, o0), hi_b = min(D + 16, o1);
                for (int64_t B = lo_b; B < hi_b; ++B) {
                    const int rel = (int)(B - o0);
                    const int q0 = rel / 3, r = rel - q0 * 3;
                    a.obs[B] = (uint8_t)(stage[q0] >> (8 * r));
                }
            }
        }
    }
}

int g_last_hip_error = 0;
int g_debug_skip = 0;

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
    ka.dbg = g_debug_skip;
    const LdsPlan p = plan_lds(*sp, ka.G);
    ka.off_tile = p.off_tile; ka.off_rows = p.off_rows; ka.off_act = p.off_act; ka.off_rng = p.off_rng;
    ka.off_rnd = p.off_rnd; ka.off_ord = p.off_ord; ka.off_rew = p.off_rew; ka.off_rec = p.off_rec;
    ka.off_mask = p.off_mask; ka.off_stage = p.off_stage;
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

// Profiling aid (not part of the product ABI): bit p set = the fused kernel skips phase Pp.  Results are
// then meaningless; tools/phase_probe.py uses it to attribute kernel time to phases.
void mgx_debug_skip_phases(int mask) { g_debug_skip = mask; }

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
    if (misaligned(grid, 16) || misaligned(agents, 8) || misaligned(obs, 16)) return MGX_ERR_INVALID_ARGUMENT;
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
    if (misaligned(grid, 16) || misaligned(agents, 8) || misaligned(obs, 16) || misaligned(rng, 8)
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
