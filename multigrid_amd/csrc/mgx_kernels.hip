// mgx_kernels.hip -- the C ABI of libmgx.so (include/mgx.h): argument checks, launch geometry, dispatch to the fused
// kernel's per-view-size translation units (mgx_fused.h / mgx_fused_inst.hip).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "mgx_fused.h"

#if MGX_SINGLE_TU      // (tools' builds that need the kernels and the host code in one module, e.g. -DMGX_SPANS=1: kernel-side
                       // globals); -DMGX_ONLY_V=<v> keeps just that view size (a third of the build time)
#define MGX_TU_HAS(v) (!defined(MGX_ONLY_V) || MGX_ONLY_V == (v))
#if MGX_TU_HAS(3)
#define MGX_INST_V 3
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#endif
#if MGX_TU_HAS(5)
#define MGX_INST_V 5
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#endif
#if MGX_TU_HAS(7)
#define MGX_INST_V 7
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#endif
#if MGX_TU_HAS(9)
#define MGX_INST_V 9
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#endif
#if MGX_TU_HAS(11)
#define MGX_INST_V 11
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#endif
#if MGX_TU_HAS(13)
#define MGX_INST_V 13
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#endif
#if MGX_TU_HAS(15)
#define MGX_INST_V 15
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#endif
#endif

namespace {

using namespace mgx;
using namespace mgx_fused;

// mode 0: gen_obs, 1: one step, 2: rollout
int launch(int mode, const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream, int *occupancy = nullptr);

int g_last_hip_error = 0;

// A HIP call of ours failed: remember its code for mgx_last_hip_error() AND take it off HIP's own per-thread "last error".  HIP keeps
// a failed call's code sticky until somebody reads it with hipGetLastError(); PyTorch reads it after every launch of its own
// (C10_HIP_KERNEL_LAUNCH_CHECK) -- so an entry point that reports a failure through its return code but leaves the sticky state set
// makes the caller's NEXT torch op raise "device kernel image is invalid" (found by the test-suite: a refused mgx_shape_register,
// then torch.zeros).  Every failure path of this library that is not a kernel launch (those read the state themselves) goes here.
int hip_failed(hipError_t e) {
    g_last_hip_error = (int)e;
    (void)hipGetLastError();
    return MGX_ERR_LAUNCH;
}
#if MGX_DEBUG_KNOBS
int g_debug_skip = 0;
int g_debug_G = 0;
int g_debug_wpb = 0;
int g_debug_grp = 0;
int g_debug_lds_pad = 0;
#else
constexpr int g_debug_skip = 0, g_debug_G = 0, g_debug_wpb = 0, g_debug_grp = 0, g_debug_lds_pad = 0;
#endif

#if MGX_BOUNDS_CHECK
int32_t *g_bounds = nullptr;        // device: [0] violations, [1] last site (checked build only; allocated on first launch)
#endif
#if MGX_SPANS
// launches since mgx_debug_span_reset(): {span_base, wavefronts, batch, first_env} each (tools/chain_overlap.py)
constexpr int kMaxSpanLaunches = 4096;
long long g_span_launches[kMaxSpanLaunches][4];
int g_span_nlaunch = 0;
int g_span_next = 0;
#endif

int launch(int mode, const KernelArgs &ka_in, int threads, int lds_bytes, int64_t nwg, hipStream_t stream, int *occupancy) {
    KernelArgs ka = ka_in;
    lds_bytes += g_debug_lds_pad;
#if MGX_BOUNDS_CHECK
    if (!g_bounds) {                      // (normally made by mgx_abi_version(), which the binding calls when it loads the library)
        hipError_t eb = hipMalloc(reinterpret_cast<void **>(&g_bounds), 8);
        if (eb == hipSuccess) eb = hipMemset(g_bounds, 0, 8);
        if (eb != hipSuccess) return hip_failed(eb);
    }
    ka.bounds = g_bounds;
#endif
#if MGX_SPANS
    if (!occupancy) {
        const long long nwaves = nwg * (threads / 64);
        ka.span_base = g_span_next;
        if (g_span_nlaunch < kMaxSpanLaunches) {
            long long *r = g_span_launches[g_span_nlaunch++];
            r[0] = g_span_next; r[1] = nwaves; r[2] = ka.batch; r[3] = ka.first_env;
        }
        g_span_next = (int)std::min<long long>((long long)kSpanCap, g_span_next + nwaves);
    }
#endif
    switch (ka.sp.view_size) {
#define MGX_CASE(V) case V: return launch_v##V(mode, ka, threads, lds_bytes, nwg, stream, &g_last_hip_error, occupancy);
#if defined(MGX_ONLY_V)
#define MGX_CASE_ONLY2(V) MGX_CASE(V)
#define MGX_CASE_ONLY(V) MGX_CASE_ONLY2(V)
    MGX_CASE_ONLY(MGX_ONLY_V)
#else
    MGX_FOR_EACH_VIEW(MGX_CASE)
#endif
#undef MGX_CASE
    default: return MGX_ERR_UNSUPPORTED;
    }
}

int check_spec(const MgxSpec *sp, int64_t batch, bool roll = false, bool one_hot = false, bool obs_only = false) {
    if (!sp || batch < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (sp->view_size < 3 || !(sp->view_size & 1)) return MGX_ERR_INVALID_ARGUMENT;   // agent.py:78-79
    if (sp->width < 3 || sp->height < 3 || sp->num_agents < 1 || sp->max_steps < 1) return MGX_ERR_INVALID_ARGUMENT;
    if (sp->view_size > MGX_MAX_VIEW || sp->num_agents > MGX_MAX_AGENTS) return MGX_ERR_UNSUPPORTED;
    if (sp->width > 255 || sp->height > 255) return MGX_ERR_UNSUPPORTED;             // positions are uint8
    if (sp->env_kind < MGX_KIND_EMPTY || sp->env_kind > MGX_KIND_RULES) return MGX_ERR_UNSUPPORTED;
    if (sp->cell_bytes < 0 || sp->cell_bytes > 3) return MGX_ERR_INVALID_ARGUMENT;
    // compact cells (include/mgx.h: MgxCell8) and byte grids (cell_bytes = 3): the plain step and gen_obs; rollouts and one-hot
    // output keep the 16-bit cells
    // (round 6: compact cells also take the hook-free STEP with one-hot output; gen_obs with one-hot output stays two launches)
    // (... and the hook-free rollout / persistent launch)
    if ((sp->cell_bytes == 1 || sp->cell_bytes == 3) && (roll || one_hot)
        && !(sp->cell_bytes == 1 && (one_hot != roll) && !obs_only && sp->env_kind == MGX_KIND_EMPTY))
        return MGX_ERR_UNSUPPORTED;
    if (wave_lds_bytes(*sp, 1, roll, one_hot, obs_only) > kLdsPerCU) return MGX_ERR_UNSUPPORTED;   // one env must fit one CU's LDS (the
                                                                                     // rollout carve is the larger one)
    return MGX_OK;
}

// Which RESIDENT shape (mgx_fused.h: kShapes 7 / 8 / 9; 0: the ordinary 32-slot kernels) a rollout / persistent launch of (spec,
// batch) takes.  Empty-16x16 x 4 agents, 7x7 views, 16-bit cells; by batch (tools/rollout_probe.py, profiles/r6_resident.txt):
//   up to 16384 envs   the 32-slot kernels: one wavefront or two per SIMD, as fast
//   up to 49152        kShapes 7 -- one slice of 16 envs per wavefront, 13216 B of LDS: 12 wavefronts per CU
//   up to 65536 (C4)   kShapes 9 -- the same slice in 9872 B (tile rows share their wall ring) and <= 128 VGPRs: SIXTEEN wavefronts per
//                      CU, so C4's 4096 wavefronts are ONE resident round (15.3 us per step against 17.2 for shape 7's round and a third)
//   beyond             kShapes 7 again (rollouts run in rounds)
// kShapes 8 (two slices, 20112 B: 8 per CU) was the first form that held C4 (19.7-20.1 us): the persistent launch of 32768 < batch
// <= 65536 envs takes it (see below)
// (MGX_RESIDENT_SHAPE=0/7/8/9 overrides; MGX_RESIDENT_SLICES=1/2 = shapes 7 / 8: the tests force them at small batches).
int resident_shape(const MgxSpec &sp, int64_t batch, bool persist) {
    const FixedShape &f = kShapes[kShapeResident1];
    if (sp.width != f.W || sp.height != f.H || sp.num_agents != f.A || sp.view_size != f.V || sp.env_kind != MGX_KIND_EMPTY
        || cell_bytes_of(sp) != f.cb || sp.cell_bytes == 3)
        return 0;
    if (const char *e = getenv("MGX_RESIDENT_SHAPE")) {                  // (read per call: the tests and tools switch it)
        if (*e) { const int k = atoi(e); return (k == kShapeResident1 || k == kShapeResident2 || k == kShapeResident4) ? k : 0; }
    }
    if (const char *e = getenv("MGX_RESIDENT_SLICES")) {
        if (*e) { const int n = atoi(e); return n <= 0 ? 0 : (n == 1 ? kShapeResident1 : kShapeResident2); }
    }
    if (batch <= 16384) return 0;
    // 12 wavefronts per CU step faster than 16 while they hold the batch (49152 envs: 10.5 against 12.3 us per step), and a
    // rollout beyond one round of shape 9 is as well off in rounds of shape 7 (98304: 22.7 / 24.1, 131072: 29.6 / 31.2).
    // The persistent launch cannot run in rounds and must leave registers for the kernels that feed it: 16 wavefronts x 128 VGPRs
    // are a SIMD's whole file (mgx_persistent_post / _wait would never be scheduled), so beyond shape 7 it takes the two-slice
    // shape 8 -- 8 wavefronts per CU x 144 VGPRs
    // (... and its residency check counts 4 workgroups = 8 wavefronts per CU, whatever the shape: 32768 envs of shape 7.  Twelve
    // persistent wavefronts of shape 7 per CU -- 4 workgroups x 3, 456 of a SIMD's 512 VGPRs -- were tried at 49152 envs: the hand-shake
    // kernels were not scheduled beside them and every wavefront timed out, profiles/r6_resident.txt)
    if (persist) return batch <= 32768 ? kShapeResident1 : kShapeResident2;
    if (batch <= 49152) return kShapeResident1;
    return batch <= 65536 ? kShapeResident4 : kShapeResident1;
}

// `step_plain`: the launch is the plain one-step kernel (mode 1 without one-hot / generation): the only one the small-group
// latency instantiations exist for (mgx_fused.h: has_small_groups)
int fill_args(KernelArgs &ka, const MgxSpec *sp, int64_t batch, int &threads, int &lds_bytes, int64_t &nwg,
              bool roll = false, bool one_hot = false, bool obs_only = false, bool step_plain = false, bool persist = false) {
    ka.sp = *sp;
    ka.batch = batch;
    ka.Gw = g_debug_G > 0 ? g_debug_G : choose_Gw(*sp, batch, roll, one_hot, obs_only);
    const int max_gw = slots_per_wave(sp->view_size, roll || obs_only || sp->cell_bytes == 1) / sp->num_agents;
    if (ka.Gw > max_gw) ka.Gw = max_gw;
    if (ka.Gw < 1) ka.Gw = 1;
    while (ka.Gw > 1 && wave_lds_bytes(*sp, ka.Gw, roll, one_hot, obs_only) > kLdsPerCU) --ka.Gw;
    ka.grp = kGroup;
    // latency regime: a wavefront that owns ONE small group of view slots (4 or 8) -- fewer slots of P2/P4/P5 on each wave's
    // instruction chain, more wavefronts per SIMD to run them side by side (mgx_fused.h: choose_group)
    if (step_plain && has_small_groups(sp->view_size, 1, false, false) && sp->num_agents <= 8 && sp->cell_bytes != 1 && sp->cell_bytes != 3) {
        const int g = g_debug_grp > 0 ? g_debug_grp : (g_debug_G > 0 ? kGroup : choose_group(*sp, batch));
        if (g < kGroup && sp->num_agents <= g) { ka.grp = g; ka.Gw = g / sp->num_agents; }
    }
    ka.dbg = g_debug_skip;
    // a grid tensor beyond half of the 256 MiB Infinity Cache is streamed (nt tile loads): mgx_fused.h, P0.  (One of exactly half --
    // C5's 32768 compact 64x64 grids -- is still better left to the caches: 60.4 against 61.1-61.6 us with nt loads.)
    ka.flags = (batch * (int64_t)sp->width * sp->height * grid_cell_bytes_of(*sp) > (int64_t)128 << 20) ? 1 : 0;
    ka.vpw = slots_in_use(*sp, ka.Gw, roll || obs_only, ka.grp);
    ka.inv_A = (65536 + sp->num_agents - 1) / sp->num_agents;
    ka.wave_lds = wave_lds_bytes(*sp, ka.Gw, roll, one_hot, obs_only, ka.grp);
    ka.ns = 0; ka.rshape = 0;
    // RESIDENT shapes of the rollout / persistent kernels (mgx_fused.h: kShapes[].ns > 0; round 6): at the batches where the 32-slot
    // rollout kernel no longer keeps every env on the chip, Empty-16x16 x 4 agents takes 64 view slots per wavefront and, beyond what
    // 12 such wavefronts per CU hold, two slices of 16 envs per wavefront (8 per CU: 65536 envs, C4, in 2048 wavefronts)
    if (roll && !one_hot && !MGX_NO_FIXED_SHAPES && g_debug_G <= 0) {
        const int rs = resident_shape(*sp, batch, persist);
        if (rs > 0) {
            const FixedShape &f = kShapes[rs];
            ka.ns = f.ns; ka.rshape = rs; ka.Gw = f.Gw; ka.vpw = shape_slots(f);
            ka.wave_lds = shape_carve(f, true).total();
        }
    }
    struct { int total; } p{ka.wave_lds};
    // wavefronts bundled per workgroup: ONE once the chip is full several times over -- single-wavefront workgroups pack a CU's
    // 160 KiB of LDS tightest and refill a CU one wavefront at a time (round 4, same box: C5 78.1 / 79.0 / 82.2 us for 1 / 2 / 4,
    // 262 144 envs of C4 59.2 / 60.5 / 60.5, 1 M envs 219.3 / 220.5 / 222.9); below that, fewer and larger workgroups launch
    // faster (9.9 vs 10.5 us at 4096 envs; C4's 4096 wavefronts: 18.9 for 4 against 19.2 for 1)
    // (... for the plain step's wavefronts, which hold more than 8 KiB of LDS each; the gen_obs kernel's small slices and the
    // one-hot step stay at 2: 1 M envs gen_obs 203 us for 1 against 190-193 for 2, fused one-hot step 1.03-1.06 ms against 0.99)
    int wpb = ((batch + ka.Gw - 1) / ka.Gw >= 16384) ? ((p.total > 8192 && !one_hot && !obs_only) ? 1 : 2) : 4;
    if (ka.ns > 0) wpb = ka.rshape == kShapeResident4 ? 4 : 2;   // (resident shapes: 12 / 8 / 16 wavefronts per CU as 6 / 4 / 4 workgroups)
    while (wpb > 1 && wpb * p.total > 64 * 1024) wpb >>= 1;
    if (g_debug_wpb > 0) wpb = g_debug_wpb;
    threads = 64 * wpb;
    lds_bytes = wpb * p.total;
    const int64_t nwaves = (batch + (int64_t)ka.Gw * std::max(ka.ns, 1) - 1) / ((int64_t)ka.Gw * std::max(ka.ns, 1));
    // latency regime: LDS-DMA tile loads, 32 view slots with unpacked cell registers (mgx_fused.h: DMA instantiations)
    // (compact cells have no latency instantiation: their launches take the throughput kernel at any size)
    if ((nwaves <= 2048 || ka.grp < kGroup) && !(ka.flags & 1) && ka.Gw * sp->num_agents <= kSlotsLatency && sp->cell_bytes != 1
        && sp->cell_bytes != 3)
        ka.flags |= 2;
    if (!(ka.flags & 2) && ka.grp < kGroup) return MGX_ERR_UNSUPPORTED;     // (cannot happen: small groups imply a small grid tensor)
    nwg = (nwaves + wpb - 1) / wpb;
    if (nwg > INT_MAX) return MGX_ERR_UNSUPPORTED;
    return MGX_OK;
}

inline bool misaligned(const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

// runtime-compiled shape instantiations (mgx_shape_register); never removed: a launch may hold a pointer into the vector's elements,
// so they live in a list of stable nodes
struct JitNode { mgx_fused::JitShape s; int device; JitNode *next; };
std::mutex g_jit_mutex;
JitNode *g_jit_head = nullptr;

}  // namespace

namespace mgx_fused {
const JitShape *jit_shape_lookup(const KernelArgs &ka, bool hooks) {
    if (!g_jit_head) return nullptr;                                   // (the common case: one load, no lock)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    for (const JitNode *n = g_jit_head; n; n = n->next) {
        const FixedShape &f = n->s.f;
        if (n->device == dev && ka.sp.view_size == f.V && ((ka.flags & 2) != 0) == f.dma && ((ka.flags & 1) != 0) == f.stream
            && ka.sp.width == f.W && ka.sp.height == f.H && ka.sp.num_agents == f.A && ka.Gw == f.Gw && hooks == f.hooks
            && ka.vpw == n->s.vpw && ka.wave_lds == n->s.wave_lds && ka.grp == kGroup)
            return &n->s;
    }
    return nullptr;
}
}  // namespace mgx_fused

extern "C" int mgx_internal_stage_candidates(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, const int32_t *episode,
                                             void *stream);      // mgx_layout_gen.hip (not in include/mgx.h)

extern "C" {

int mgx_abi_version(void) {
#if MGX_BOUNDS_CHECK
    // (checked build: the violation counter is allocated HERE -- the process' first launch may be one that a stream capture records
    // (tools/fuzz_generate.py opens some cases with a captured block), and an allocation + memset inside a capture is an error.  On a box
    // without a GPU the calls fail and nothing is kept.)
    if (!g_bounds) {
        int32_t *p = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&p), 8) == hipSuccess) {
            if (hipMemset(p, 0, 8) == hipSuccess) g_bounds = p; else (void)hipFree(p);
        }
        (void)hipGetLastError();
    }
#endif
    return MGX_ABI_VERSION;
}

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

#if MGX_DEBUG_KNOBS
// Profiling aid, only in the tools' build (lib/libmgx_dbg.so; never in libmgx.so): bit p set = the fused kernel skips
// phase Pp.  Results are then meaningless; tools/graph_phase.py uses it to attribute kernel time to phases.
void mgx_debug_skip_phases(int mask) { g_debug_skip = mask; }
void mgx_debug_set_envs_per_wavefront(int G) { g_debug_G = G; }
void mgx_debug_set_waves_per_workgroup(int n) { g_debug_wpb = n; }
void mgx_debug_set_lds_pad(int bytes) { g_debug_lds_pad = bytes; }   // unused LDS bytes per workgroup: limits the wavefronts a CU holds at a time
void mgx_debug_set_group(int g) { g_debug_grp = g; }               // 4 / 8: force the small-group latency instantiation, 16: forbid it
#endif
#if MGX_BOUNDS_CHECK
// Checked build only: LDS accesses of the fused kernel that fell outside their wavefront's slice since the last call
// (out[0]) and the site id of the last one (out[1]); synchronises the device.  Returns 0, or -1 on a HIP error.
int mgx_debug_bounds_violations(int32_t *out2) {
    out2[0] = out2[1] = 0;
    if (!g_bounds) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(out2, g_bounds, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return hipMemset(g_bounds, 0, 8) == hipSuccess ? 0 : -1;
}
#endif
#if MGX_SPANS
// Wave spans: every launch since the last reset owns records [span_base, span_base + wavefronts) of g_span.
void mgx_debug_span_reset(void) { g_span_next = 0; g_span_nlaunch = 0; }
int mgx_debug_span_launches(long long *out4, int max_launches) {          // -> number of launches; out4[i] = {base, waves, batch, first_env}
    const int n = g_span_nlaunch < max_launches ? g_span_nlaunch : max_launches;
    for (int i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) out4[4 * i + k] = g_span_launches[i][k];
    return n;
}
int mgx_debug_read_span(unsigned long long *out, int first, int count) {  // records [first, first + count), [begin, end] each
    if (first < 0 || count < 0 || first + count > kSpanCap) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 2 * count,
                            sizeof(unsigned long long) * 2 * first) != hipSuccess) return -1;
    for (int i = 0; i < count; ++i) out[2 * i + 1] &= 0x0fffffffffffffffull;   // (bits 60..: the wave's path flags, below)
    return 0;
}
// which rare paths the wavefronts of records [first, first + count) took: 1 = auto-reset, 2 = sequential fallback, 4 = success /
// failure events, 8 = cell writes (mgx_fused_body.inc: MGX_SPAN_FLAG)
int mgx_debug_read_span_flags(unsigned char *out, int first, int count) {
    if (first < 0 || count < 0 || first + count > kSpanCap) return -1;
    std::vector<unsigned long long> tmp(2 * (size_t)count);
    if (hipMemcpyFromSymbol(tmp.data(), HIP_SYMBOL(g_span), sizeof(unsigned long long) * 2 * count,
                            sizeof(unsigned long long) * 2 * first) != hipSuccess) return -1;
    for (int i = 0; i < count; ++i) out[i] = (unsigned char)(tmp[2 * i + 1] >> 60);
    return 0;
}
#endif
#if MGX_TIMESTAMPS
int mgx_debug_read_stamps(unsigned long long *out64, long long wave) {   // reads the last launch's stamps, selects the next wave
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    unsigned long long zero[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof zero) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_wave), &wave, sizeof wave) == hipSuccess ? 0 : -1;
}
#endif

int mgx_launch_info(const MgxSpec *spec, int64_t batch, MgxLaunchInfo *out) {
    int rc = check_spec(spec, batch);
    if (rc) return rc;
    if (!out) return MGX_ERR_INVALID_ARGUMENT;
    KernelArgs ka{};
    int threads = 0, lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, threads, lds, nwg, false, false, false, true);     // (the plain step's geometry)
    if (rc) return rc;
    out->envs_per_wavefront = ka.Gw;
    out->envs_per_workgroup = ka.Gw * (threads / 64);
    out->threads_per_workgroup = threads;
    out->workgroups = (int32_t)nwg;
    out->lds_bytes = lds;
    out->slots_per_group = ka.grp;
    out->fixed_shape = match_fixed_shape(ka, spec->env_kind != MGX_KIND_EMPTY);
    if (!out->fixed_shape && jit_shape_lookup(ka, spec->env_kind != MGX_KIND_EMPTY)) out->fixed_shape = MGX_SHAPE_RUNTIME_COMPILED;
    return MGX_OK;
}

int mgx_rollout_info(const MgxSpec *spec, int64_t batch, int32_t persistent, MgxRolloutInfo *out) {
    int rc = check_spec(spec, batch, true, false);
    if (rc) return rc;
    if (!out) return MGX_ERR_INVALID_ARGUMENT;
    KernelArgs ka{};
    int threads = 0, lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, threads, lds, nwg, true, false, false, false, persistent != 0);
    if (rc) return rc;
    const int64_t epw = (int64_t)ka.Gw * std::max(ka.ns, 1);
    out->envs_per_slice = ka.Gw;
    out->slices = std::max(ka.ns, 1);
    out->wavefronts = (int32_t)((batch + epw - 1) / epw);
    out->threads_per_workgroup = threads;
    out->workgroups = (int32_t)nwg;
    out->lds_bytes = lds;
    out->resident_shape = match_resident_shape(ka, spec->env_kind != MGX_KIND_EMPTY);
    return MGX_OK;
}

// ---- runtime-compiled shape specialisation (include/mgx.h: MgxShapeKey) -------------------------------------------------
int mgx_shape_key(const MgxSpec *spec, int64_t batch, MgxShapeKey *key) {
    int rc = check_spec(spec, batch);
    if (rc) return rc;
    if (!key || batch < 1) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->cell_bytes == 1 || spec->cell_bytes == 3) return MGX_ERR_UNSUPPORTED;            // (runtime-compiled shapes: 16-bit cells)
    KernelArgs ka{};
    int threads = 0, lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, threads, lds, nwg, false, false, false, true);
    if (rc) return rc;
    const bool hooks = spec->env_kind != MGX_KIND_EMPTY;
    key->width = spec->width; key->height = spec->height; key->num_agents = spec->num_agents; key->envs_per_wavefront = ka.Gw;
    key->hooks = hooks ? 1 : 0; key->view_size = spec->view_size; key->dma = (ka.flags & 2) ? 1 : 0; key->stream = (ka.flags & 1) ? 1 : 0;
    key->kernel_args_bytes = (int32_t)sizeof(KernelArgs);
    key->built_in = match_fixed_shape(ka, hooks);
    key->registered = jit_shape_lookup(ka, hooks) ? 1 : 0;
    return MGX_OK;
}

int mgx_shape_register(const MgxShapeKey *key, const void *code_object, size_t bytes) {
    if (!key || !code_object || bytes == 0) return MGX_ERR_INVALID_ARGUMENT;
    if (key->kernel_args_bytes != (int32_t)sizeof(KernelArgs)) return MGX_ERR_INVALID_ARGUMENT;   // built for another library
    if (key->view_size < 3 || key->view_size > MGX_MAX_VIEW || !(key->view_size & 1) || key->num_agents < 1
        || key->num_agents > MGX_MAX_AGENTS || key->envs_per_wavefront < 1)
        return MGX_ERR_INVALID_ARGUMENT;
    JitShape js{};
    js.f = FixedShape{key->width, key->height, key->num_agents, key->envs_per_wavefront, key->hooks != 0, key->view_size, key->dma != 0,
                      key->stream != 0};
    js.vpw = shape_slots(js.f);
    js.wave_lds = make_carve(js.f.W, js.f.H, js.f.A, js.f.V, js.f.Gw, js.vpw, false, js.f.hooks, false, kGroup).total();
    int dev = 0;
    hipModule_t mod = nullptr;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipModuleLoadData(&mod, code_object);
    if (e == hipSuccess) e = hipModuleGetFunction(&js.fn[0], mod, "mgx_jit_step");
    if (e == hipSuccess) e = hipModuleGetFunction(&js.fn[1], mod, "mgx_jit_step_ar");
    int32_t *d_size = nullptr; size_t gbytes = 0; int32_t built_for = 0;
    if (e == hipSuccess) e = hipModuleGetGlobal(reinterpret_cast<hipDeviceptr_t *>(&d_size), &gbytes, mod, "mgx_jit_kernel_args_bytes");
    if (e == hipSuccess) e = hipMemcpy(&built_for, d_size, sizeof built_for, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        if (mod) (void)hipModuleUnload(mod);
        return hip_failed(e);
    }
    if (built_for != (int32_t)sizeof(KernelArgs)) { (void)hipModuleUnload(mod); return MGX_ERR_INVALID_ARGUMENT; }   // compiled from other headers
    if (js.wave_lds * 4 > 64 * 1024)
        for (int k = 0; k < 2; ++k)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(js.fn[k]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    g_jit_head = new JitNode{js, dev, g_jit_head};
    return MGX_OK;
}

static int gen_obs_common(bool one_hot, const MgxSpec *spec, int64_t batch, const MgxCell *grid, const uint8_t *agents,
                          uint8_t *obs, uint8_t *dir, void *stream) {
    int rc = check_spec(spec, batch, false, one_hot, true);
    if (rc) return rc;
    if (batch == 0) return MGX_OK;
    if (!grid || !agents || !obs) return MGX_ERR_INVALID_ARGUMENT;
    if (misaligned(grid, 16) || misaligned(agents, 8) || misaligned(obs, 16)) return MGX_ERR_INVALID_ARGUMENT;
    KernelArgs ka{};
    int threads = 0, lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, threads, lds, nwg, false, one_hot, true);
    if (rc) return rc;
    ka.grid = reinterpret_cast<uint8_t *>(const_cast<MgxCell *>(grid));
    ka.agents = const_cast<uint8_t *>(agents);
    ka.obs = obs;
    ka.dir = dir;
    ka.T = 1;
    return launch(one_hot ? 4 : 0, ka, threads, lds, nwg, static_cast<hipStream_t>(stream));
}

int mgx_gen_obs(const MgxSpec *spec, int64_t batch, const MgxCell *grid, const uint8_t *agents,
                uint8_t *obs, uint8_t *dir, void *stream) {
    return gen_obs_common(false, spec, batch, grid, agents, obs, dir, stream);
}

int mgx_gen_obs_one_hot(const MgxSpec *spec, int64_t batch, const MgxCell *grid, const uint8_t *agents,
                        uint8_t *obs_one_hot, uint8_t *dir, void *stream) {
    return gen_obs_common(true, spec, batch, grid, agents, obs_one_hot, dir, stream);
}

// The one implementation behind every step entry point (include/mgx.h: MgxStepArgs).  `occupancy`: query only (mgx_sub_shards).
static int step_common(const MgxSpec *spec, int64_t batch, const MgxStepArgs &sa, void *stream, int *occupancy = nullptr) {
    const bool roll = sa.steps != 1, one_hot = sa.one_hot != 0;
    const MgxAutoReset *ar = sa.auto_reset;
    const MgxLayoutGen *gen = sa.generate;
    int rc = check_spec(spec, batch, roll, one_hot);
    if (rc) return rc;
    if (sa.steps < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (batch == 0 || sa.steps == 0) return MGX_OK;
    if (!occupancy) {
        if (!sa.grid || !sa.agents || !sa.step_count || !sa.actions || !sa.obs || !sa.reward || !sa.terminated || !sa.truncated)
            return MGX_ERR_INVALID_ARGUMENT;
        if (spec->num_agents > 1 && !sa.rng) return MGX_ERR_INVALID_ARGUMENT;
        if (spec->env_kind != MGX_KIND_EMPTY && !sa.aux) return MGX_ERR_INVALID_ARGUMENT;
        if (misaligned(sa.grid, 16) || misaligned(sa.agents, 8) || misaligned(sa.obs, 16) || misaligned(sa.rng, 8)
            || misaligned(sa.reward, 8) || misaligned(sa.step_count, 4) || misaligned(sa.err, 4) || misaligned(sa.aux, 16))
            return MGX_ERR_INVALID_ARGUMENT;
    }
    KernelArgs ka{};
    if (ar) {
        if (!occupancy) {
            if (ar->pool_size < 1 || ar->first_env < 0 || !ar->pool_grid || !ar->pool_agents || !ar->episode)
                return MGX_ERR_INVALID_ARGUMENT;
            if (spec->env_kind != MGX_KIND_EMPTY && !ar->pool_aux) return MGX_ERR_INVALID_ARGUMENT;
            if (misaligned(ar->pool_agents, 8) || misaligned(ar->pool_aux, 16) || misaligned(ar->episode, 4))
                return MGX_ERR_INVALID_ARGUMENT;
        }
        ka.pool_size = ar->pool_size; ka.first_env = ar->first_env;
        ka.pool_magic = ar->pool_size > 1 ? ~0ull / (uint64_t)ar->pool_size + 1ull : 0ull; ka.pool_grid = reinterpret_cast<const uint8_t *>(ar->pool_grid);
        ka.pool_agents = ar->pool_agents; ka.pool_aux = ar->pool_aux; ka.episode = ar->episode;
        ka.was_reset = ar->was_reset;
        if (occupancy && !ka.pool_grid) ka.pool_grid = reinterpret_cast<const uint8_t *>(spec);   // (selects the AR instantiation)
    }
    int threads = 0, lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, threads, lds, nwg, roll, one_hot, false, !roll && !one_hot && !gen);
    if (rc) return rc;
    ka.grid = reinterpret_cast<uint8_t *>(sa.grid); ka.agents = sa.agents; ka.rng = sa.rng; ka.step_count = sa.step_count;
    ka.actions = sa.actions; ka.hook_order = sa.hook_order;
    ka.aux = sa.aux; ka.obs = sa.obs; ka.dir = sa.dir; ka.reward = sa.reward; ka.terminated = sa.terminated;
    ka.truncated = sa.truncated; ka.err = sa.err;
    ka.grid_bad = spec->cell_bytes == 3 ? sa.grid_bad : nullptr;
    if (misaligned(ka.grid_bad, 4)) return MGX_ERR_INVALID_ARGUMENT;
    ka.T = roll ? sa.steps : 1;
    int mode = (roll ? 2 : 1) | (one_hot ? 4 : 0);
    if (gen) {
        if (ar) return MGX_ERR_INVALID_ARGUMENT;
        if (spec->cell_bytes == 1 || spec->cell_bytes == 3) return MGX_ERR_UNSUPPORTED;        // (device-side generation writes 16-bit cells)
        if (roll) {
            // steps = T with generation: T launches of the one-step kernel over the [t] slices.  (Generation writes the HBM state;
            // the one-launch rollout keeps the state in LDS between its steps, so the two do not combine into a single launch --
            // a rollout in one launch restarts its finished envs from the layout pool: auto_reset.)  Same results as T calls.
            if (occupancy) { MgxStepArgs one = sa; one.steps = 1; return step_common(spec, batch, one, stream, occupancy); }
            const int64_t BA = batch * spec->num_agents, V2 = (int64_t)spec->view_size * spec->view_size * (one_hot ? 21 : 3);
            MgxLayoutGen gen_t = *gen;
            if (gen_t.stage.external) gen_t.stage.external = 2;                 // (the generator launches below share the steps' stream)
            for (int32_t t = 0; t < sa.steps; ++t) {
                MgxStepArgs one = sa;
                one.steps = 1;
                gen_t.stage.phase = gen->stage.phase + t;                       // (the staging protocol counts the steps)
                one.generate = &gen_t;
                one.actions = sa.actions + t * BA;
                one.hook_order = sa.hook_order ? sa.hook_order + t * BA : nullptr;
                one.obs = sa.obs + t * BA * V2;
                one.dir = sa.dir ? sa.dir + t * BA : nullptr;
                one.reward = sa.reward + t * BA;
                one.terminated = sa.terminated + t * BA;
                one.truncated = sa.truncated + t * batch;
                one.was_reset = sa.was_reset ? sa.was_reset + t * batch : nullptr;
                rc = step_common(spec, batch, one, stream, nullptr);
                if (rc) return rc;
                // staging with the generator between the steps (MgxGenStage.external): every lead / 2 steps, as the callers of the
                // one-step form do it
                if (gen->stage.external && gen->stage.tag) {
                    const int every = gen->stage.lead >= 4 ? gen->stage.lead / 2 : 1;
                    if ((gen_t.stage.phase + 1) % every == 0) {
                        rc = mgx_stage_generate(spec, batch, &gen_t, sa.rng, sa.episode, stream);
                        if (rc) return rc;
                    }
                }
            }
            return MGX_OK;
        }
        if (!occupancy) {
            if (!gen->blank || !gen->gen_state || !sa.episode || !sa.rng) return MGX_ERR_INVALID_ARGUMENT;
            if (misaligned(gen->gen_state, 8) || misaligned(sa.episode, 4)) return MGX_ERR_INVALID_ARGUMENT;
        }
        if (spec->width > 254 || spec->height > 254) return MGX_ERR_UNSUPPORTED;
        rc = mgx_gen::check_layout_gen(spec, gen);
        if (rc) return rc;
        ka.gen = *gen; ka.episode = sa.episode; ka.was_reset = sa.was_reset;
        ka.gen_first_wg = INT64_MAX;
        // staged generation (include/mgx.h: MgxGenStage): generator wavefronts behind the step's own workgroups, one lane per env
        MgxGenStage &st = ka.gen.stage;
        const bool staged = st.grid && st.agents && st.words && st.tag && (st.aux || spec->env_kind == MGX_KIND_EMPTY)
                            && spec->num_agents > 1 && ka.wave_lds >= 128 * spec->num_agents;
        if (staged) {
            if (misaligned(st.grid, 4) || misaligned(st.agents, 8) || misaligned(st.aux, 16) || misaligned(st.words, 8)
                || misaligned(st.tag, 16))
                return MGX_ERR_INVALID_ARGUMENT;
            if (st.lead < 0 || st.lead >= spec->max_steps) return MGX_ERR_INVALID_ARGUMENT;
            if (st.lead < 2) st.lead = 2;
            if (st.external < 0 || st.external > 2) return MGX_ERR_INVALID_ARGUMENT;
            // candidates (ABI 10): only with the generator launches between the steps, only what the generator kind offers
            if (st.candidates < 0 || (st.candidates > 0 && (st.external != 2 || st.candidates != mgx_gen::stage_candidates(gen))))
                return MGX_ERR_INVALID_ARGUMENT;
            if (!st.external) {                                  // generator wavefronts behind the step's own workgroups
                const int wpb = threads / 64;
                const int64_t gen_waves = (batch + 63) / 64;
                ka.gen_first_wg = nwg;
                nwg += (gen_waves + wpb - 1) / wpb;
                if (nwg > INT_MAX) return MGX_ERR_UNSUPPORTED;
            }
        } else {
            st = MgxGenStage{};
        }
        mode |= 8;
    }
    return launch(mode, ka, threads, lds, nwg, static_cast<hipStream_t>(stream), occupancy);
}

static MgxStepArgs step_args(MgxCell *grid, uint8_t *agents, uint64_t *rng, int32_t *step_count, const int8_t *actions,
                             uint8_t *aux, uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                             int32_t *err) {
    MgxStepArgs sa{};
    sa.grid = grid; sa.agents = agents; sa.rng = rng; sa.step_count = step_count; sa.aux = aux; sa.actions = actions;
    sa.obs = obs; sa.dir = dir; sa.reward = reward; sa.terminated = terminated; sa.truncated = truncated; sa.err = err;
    sa.steps = 1;
    return sa;
}

int mgx_step_ex(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, void *stream) {
    if (!args) return MGX_ERR_INVALID_ARGUMENT;
    return step_common(spec, batch, *args, stream);
}

int mgx_step(const MgxSpec *spec, int64_t batch, MgxCell *grid, uint8_t *agents, uint64_t *rng,
             int32_t *step_count, const int8_t *actions, uint8_t *aux,
             uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
             int32_t *err, void *stream) {
    return step_common(spec, batch, step_args(grid, agents, rng, step_count, actions, aux, obs, dir, reward, terminated, truncated, err), stream);
}

int mgx_rollout(const MgxSpec *spec, int64_t batch, int32_t steps, MgxCell *grid, uint8_t *agents, uint64_t *rng,
                int32_t *step_count, const int8_t *actions, uint8_t *aux,
                uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                int32_t *err, void *stream) {
    MgxStepArgs sa = step_args(grid, agents, rng, step_count, actions, aux, obs, dir, reward, terminated, truncated, err);
    if (steps == 0) return check_spec(spec, batch, true);
    if (steps < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (steps == 1) return step_common(spec, batch, sa, stream);          // (one step IS the step kernel: same results)
    sa.steps = steps;
    return step_common(spec, batch, sa, stream);
}

int mgx_step_autoreset(const MgxSpec *spec, int64_t batch, const MgxAutoReset *ar, MgxCell *grid, uint8_t *agents,
                       uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                       uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                       int32_t *err, void *stream) {
    if (!ar) return MGX_ERR_INVALID_ARGUMENT;
    MgxStepArgs sa = step_args(grid, agents, rng, step_count, actions, aux, obs, dir, reward, terminated, truncated, err);
    sa.auto_reset = ar;
    return step_common(spec, batch, sa, stream);
}

int mgx_rollout_autoreset(const MgxSpec *spec, int64_t batch, int32_t steps, const MgxAutoReset *ar, MgxCell *grid,
                          uint8_t *agents, uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                          uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                          int32_t *err, void *stream) {
    if (!ar || steps < 0) return MGX_ERR_INVALID_ARGUMENT;
    MgxStepArgs sa = step_args(grid, agents, rng, step_count, actions, aux, obs, dir, reward, terminated, truncated, err);
    sa.auto_reset = ar;
    if (steps == 0) return check_spec(spec, batch, true);
    sa.steps = steps;
    return step_common(spec, batch, sa, stream);
}

int mgx_step_one_hot(const MgxSpec *spec, int64_t batch, const MgxAutoReset *ar, MgxCell *grid, uint8_t *agents,
                     uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                     uint8_t *obs_one_hot, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                     int32_t *err, void *stream) {
    MgxStepArgs sa = step_args(grid, agents, rng, step_count, actions, aux, obs_one_hot, dir, reward, terminated, truncated, err);
    sa.auto_reset = ar; sa.one_hot = 1;
    return step_common(spec, batch, sa, stream);
}

int mgx_stage_generate(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, const uint64_t *rng, const int32_t *episode,
                       void *stream) {
    int rc = check_spec(spec, batch);
    if (rc) return rc;
    if (batch == 0) return MGX_OK;
    if (spec->cell_bytes == 1 || spec->cell_bytes == 3) return MGX_ERR_UNSUPPORTED;
    if (!gen || !rng || !episode || !gen->blank || !gen->gen_state) return MGX_ERR_INVALID_ARGUMENT;
    const MgxGenStage &gs = gen->stage;
    if (!gs.grid || !gs.agents || !gs.words || !gs.tag || (!gs.aux && spec->env_kind != MGX_KIND_EMPTY) || spec->num_agents < 2)
        return MGX_ERR_INVALID_ARGUMENT;
    if (misaligned(gs.grid, 4) || misaligned(gs.agents, 8) || misaligned(gs.aux, 16) || misaligned(gs.words, 8) || misaligned(gs.tag, 16)
        || misaligned(gen->gen_state, 8) || misaligned(rng, 8) || misaligned(episode, 4))
        return MGX_ERR_INVALID_ARGUMENT;
    if (spec->width > 254 || spec->height > 254) return MGX_ERR_UNSUPPORTED;
    rc = mgx_gen::check_layout_gen(spec, gen);
    if (rc) return rc;
    KernelArgs ka{};
    int threads = 0, lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, threads, lds, nwg, false, false, false, false);     // (the generated step's carve: its LDS slice)
    if (rc) return rc;
    if (ka.wave_lds < 128 * spec->num_agents) return MGX_ERR_UNSUPPORTED;
    ka.rng = const_cast<uint64_t *>(rng);
    ka.episode = const_cast<int32_t *>(episode);
    ka.gen = *gen;
    ka.gen.stage.phase = -2;                             // (no request carries phase + 1 = -1: every pending one is served)
    ka.gen.stage.lead = ka.gen.stage.lead < 2 ? 2 : ka.gen.stage.lead;
    ka.gen_first_wg = 0;                                 // every workgroup of this launch is a generator workgroup
    ka.T = 1;
    if (gs.candidates < 0 || (gs.candidates > 0 && (gs.external != 2 || gs.candidates != mgx_gen::stage_candidates(gen))))
        return MGX_ERR_INVALID_ARGUMENT;
    if (gs.candidates > 0) return mgx_internal_stage_candidates(spec, batch, gen, episode, stream);   // (a kernel of its own: mgx_layout_gen.hip)
    const int wpb = threads / 64;
    const int64_t gen_waves = (batch + 63) / 64;
    nwg = (gen_waves + wpb - 1) / wpb;
    if (nwg > INT_MAX) return MGX_ERR_UNSUPPORTED;
    return launch(1 | 8, ka, threads, lds, nwg, static_cast<hipStream_t>(stream));
}

int mgx_step_generate(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, MgxCell *grid, uint8_t *agents,
                      uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                      uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                      int32_t *err, int32_t *episode, uint8_t *was_reset, void *stream) {
    if (!gen) return MGX_ERR_INVALID_ARGUMENT;
    MgxStepArgs sa = step_args(grid, agents, rng, step_count, actions, aux, obs, dir, reward, terminated, truncated, err);
    sa.generate = gen; sa.episode = episode; sa.was_reset = was_reset;
    return step_common(spec, batch, sa, stream);
}

// ---- sub-shard stepping (include/mgx.h) ------------------------------------------------------------------------------
static constexpr int64_t kSubShardAlign = 64;      // envs: keeps every block's tensors 16-byte aligned and its wavefronts' tiles as in the whole

static void sub_shard_cuts(int64_t B, int parts, int64_t *cuts /* [parts + 1] */) {
    cuts[0] = 0;
    for (int i = 1; i < parts; ++i) {
        int64_t c = ((B * i / parts + kSubShardAlign - 1) / kSubShardAlign) * kSubShardAlign;
        cuts[i] = c > B ? B : c;
    }
    cuts[parts] = B;
}

int mgx_sub_shards(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, int32_t *parts) {
    if (!parts) return MGX_ERR_INVALID_ARGUMENT;
    *parts = 1;
    MgxStepArgs sa{};
    if (args) sa = *args; else sa.steps = 1;
    if (sa.steps != 1) return MGX_OK;                                  // rollouts: one launch, nothing to chain
    int rc = check_spec(spec, batch, false, sa.one_hot != 0);
    if (rc) return rc;
    if (batch == 0) return MGX_OK;
    int dev = 0;
    hipDeviceProp_t prop;
    {
        hipError_t ed = hipGetDevice(&dev);
        if (ed == hipSuccess) ed = hipGetDeviceProperties(&prop, dev);
        if (ed != hipSuccess) return hip_failed(ed);
    }
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    const int simds = 4 * cus;                                         // CDNA: four SIMDs per CU
    KernelArgs ka{};
    int threads = 0, lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, threads, lds, nwg, false, sa.one_hot != 0, false, !sa.one_hot && !sa.generate);
    if (rc) return rc;
    if (ka.grp < kGroup) return MGX_OK;                                // the latency regime: one launch
    const int64_t nwaves = (batch + ka.Gw - 1) / ka.Gw;
    // below two wavefronts per SIMD (or a view per lane of two waves per SIMD) a launch is a lone wave's instruction chain:
    // splitting it only adds launches
    if (nwaves < 2 * (int64_t)simds || batch * spec->num_agents < 64 * (int64_t)simds) return MGX_OK;
    // Two chains.  Round 3 answered 4 when the whole batch is about ONE round of resident wavefronts (C4: 65536 envs = 4096 wavefronts
    // at 16 per CU); measured again in round 5 on two boxes and over graph lengths (profiles/r5_chain_policy.txt): four chains of
    // C4 run anywhere between 15.3 and 17.7 us per step of the batch -- their advantage hangs on how fast the host side of a graph
    // replay feeds four queues, and it is gone on graphs of 4000+ kernel nodes (18.2-18.6, the "regression" of round 4's bench
    // line, which timed 1000-step graphs where round 3's timed 260-step ones) -- while two chains stay at 16.0-16.7 us on every
    // box and graph length up to 2 x 1000 nodes (lock step: 18.6-18.8); C5: 47.0 / 47.6 / 48.1 us for 2 / 3 / 4 chains (61.5 in
    // lock step).  The policy is the form that holds.
    *parts = 2;
    const int64_t max_parts = batch / kSubShardAlign;
    if (*parts > max_parts) *parts = (int32_t)(max_parts < 1 ? 1 : max_parts);
    return MGX_OK;
}

int mgx_step_chains(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, int32_t parts, void *const *streams,
                    void *fork_event) {
    if (!args || !spec || parts < 1 || parts > 64 || !streams || args->steps != 1) return MGX_ERR_INVALID_ARGUMENT;
    if (parts > 1 && batch / kSubShardAlign < parts) return MGX_ERR_INVALID_ARGUMENT;
    int64_t cuts[65];
    sub_shard_cuts(batch, parts, cuts);
    const int64_t A = spec->num_agents, HW = (int64_t)spec->width * spec->height, V2 = (int64_t)spec->view_size * spec->view_size;
    const int64_t obs_cell = args->one_hot ? 21 : 3;
    for (int k = 0; k < parts; ++k) {
        const int64_t lo = cuts[k], n = cuts[k + 1] - lo;
        if (n <= 0) continue;
        hipStream_t st = static_cast<hipStream_t>(streams[k]);
        if (fork_event) {
            const hipError_t ew = hipStreamWaitEvent(st, static_cast<hipEvent_t>(fork_event), 0);
            if (ew != hipSuccess) return hip_failed(ew);
        }
        MgxStepArgs sa = *args;
        MgxAutoReset ar;
        MgxLayoutGen gen;
        sa.grid = args->grid ? reinterpret_cast<MgxCell *>(reinterpret_cast<uint8_t *>(args->grid) + lo * HW * grid_cell_bytes_of(*spec)) : nullptr;
        sa.agents = args->agents ? args->agents + lo * A * MGX_AGENT_STRIDE : nullptr;
        sa.rng = args->rng ? args->rng + lo * 4 : nullptr;
        sa.step_count = args->step_count ? args->step_count + lo : nullptr;
        sa.aux = args->aux ? args->aux + lo * MGX_AUX_BYTES : nullptr;
        sa.actions = args->actions ? args->actions + lo * A : nullptr;
        sa.hook_order = args->hook_order ? args->hook_order + lo * A : nullptr;
        sa.obs = args->obs ? args->obs + lo * A * V2 * obs_cell : nullptr;
        sa.dir = args->dir ? args->dir + lo * A : nullptr;
        sa.reward = args->reward ? args->reward + lo * A : nullptr;
        sa.terminated = args->terminated ? args->terminated + lo * A : nullptr;
        sa.truncated = args->truncated ? args->truncated + lo : nullptr;
        sa.episode = args->episode ? args->episode + lo : nullptr;
        sa.was_reset = args->was_reset ? args->was_reset + lo : nullptr;
        if (args->auto_reset) {
            ar = *args->auto_reset;
            ar.first_env += lo;
            ar.episode = ar.episode ? ar.episode + lo : nullptr;
            ar.was_reset = ar.was_reset ? ar.was_reset + lo : nullptr;
            sa.auto_reset = &ar;
        }
        if (args->generate) {
            gen = *args->generate;
            gen.gen_state = gen.gen_state ? gen.gen_state + lo * 6 : nullptr;
            MgxGenStage &st = gen.stage;
            const int64_t K = st.candidates > 0 ? st.candidates : 1;         // (candidates: K slots per env, 6 words each)
            st.grid = st.grid ? st.grid + lo * K * HW : nullptr;
            st.agents = st.agents ? st.agents + lo * K * A * MGX_AGENT_STRIDE : nullptr;
            st.aux = st.aux ? st.aux + lo * K * MGX_AUX_BYTES : nullptr;
            st.words = st.words ? st.words + lo * (st.candidates > 0 ? 6 * K : 12) : nullptr;
            st.tag = st.tag ? st.tag + lo * 4 : nullptr;
            sa.generate = &gen;
        }
        const int rc = step_common(spec, n, sa, st);
        if (rc) return rc;
    }
    return MGX_OK;
}

// ---- persistent stepping (include/mgx.h: MgxPersistent) --------------------------------------------------------------
// Geometry + residency of the persistent launch: the rollout kernel's carve, every wavefront resident at once.
static int persistent_geometry(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, KernelArgs &ka, int &threads, int &lds,
                               int64_t &nwg) {
    int rc = check_spec(spec, batch, true, false);
    if (rc) return rc;
    if (batch < 1) return MGX_ERR_INVALID_ARGUMENT;
    if (args && (args->one_hot || args->generate || args->hook_order)) return MGX_ERR_UNSUPPORTED;
    const MgxAutoReset *ar = args ? args->auto_reset : nullptr;
    if (ar) ka.pool_grid = reinterpret_cast<const uint8_t *>(ar->pool_grid ? ar->pool_grid : reinterpret_cast<const MgxCell *>(spec));
    rc = fill_args(ka, spec, batch, threads, lds, nwg, true, false, false, false, true);
    if (rc) return rc;
    ka.T = 1;
    int occ = 0;
    rc = launch(3, ka, threads, lds, nwg, nullptr, &occ);              // workgroups of THIS instantiation one CU holds
    if (rc) return rc;
    int dev = 0;
    hipDeviceProp_t prop;
    {
        hipError_t ed = hipGetDevice(&dev);
        if (ed == hipSuccess) ed = hipGetDeviceProperties(&prop, dev);
        if (ed != hipSuccess) return hip_failed(ed);
    }
    // (the occupancy API is optimistic for kernels with many SGPRs -- guide: "Residency and cooperative launch" -- and a
    // workgroup that is admitted but not resident would hang the hand-shake until its timeout: at most 4 per CU are counted)
    const int64_t resident = (int64_t)(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1) * std::min(occ, 4);
    if (nwg > resident) return MGX_ERR_UNSUPPORTED;
    return MGX_OK;
}

int mgx_persistent_waves(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, int32_t *waves) {
    if (!waves) return MGX_ERR_INVALID_ARGUMENT;
    KernelArgs ka{};
    int threads = 0, lds = 0; int64_t nwg = 0;
    const int rc = persistent_geometry(spec, batch, args, ka, threads, lds, nwg);
    if (rc) return rc;
    const int64_t epw = (int64_t)ka.Gw * std::max(ka.ns, 1);     // envs per wavefront (resident shapes: all of its slices)
    *waves = (int32_t)((batch + epw - 1) / epw);            // (the wavefronts that own envs: the ones that publish a flag)
    return MGX_OK;
}

int mgx_step_persistent(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, const MgxPersistent *p, void *stream) {
    if (!args || !p) return MGX_ERR_INVALID_ARGUMENT;
    const MgxStepArgs &sa = *args;
    KernelArgs ka{};
    int threads = 0, lds = 0; int64_t nwg = 0;
    int rc = persistent_geometry(spec, batch, args, ka, threads, lds, nwg);
    if (rc) return rc;
    if (!sa.grid || !sa.agents || !sa.step_count || !sa.obs || !sa.reward || !sa.terminated || !sa.truncated)
        return MGX_ERR_INVALID_ARGUMENT;
    if (spec->num_agents > 1 && !sa.rng) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->env_kind != MGX_KIND_EMPTY && !sa.aux) return MGX_ERR_INVALID_ARGUMENT;
    if (misaligned(sa.grid, 16) || misaligned(sa.agents, 8) || misaligned(sa.obs, 16) || misaligned(sa.rng, 8)
        || misaligned(sa.reward, 8) || misaligned(sa.step_count, 4) || misaligned(sa.err, 4) || misaligned(sa.aux, 16))
        return MGX_ERR_INVALID_ARGUMENT;
    if (!p->action_granules || !p->done || !p->ctrl || p->max_steps < 1 || p->timeout_ms < 1 || p->timeout_ms > 30000
        || misaligned(p->action_granules, 8) || misaligned(p->done, 4) || misaligned(p->ctrl, 4))
        return MGX_ERR_INVALID_ARGUMENT;
    if (const MgxAutoReset *ar = sa.auto_reset) {
        if (ar->pool_size < 1 || ar->first_env < 0 || !ar->pool_grid || !ar->pool_agents || !ar->episode) return MGX_ERR_INVALID_ARGUMENT;
        if (spec->env_kind != MGX_KIND_EMPTY && !ar->pool_aux) return MGX_ERR_INVALID_ARGUMENT;
        if (misaligned(ar->pool_agents, 8) || misaligned(ar->pool_aux, 16) || misaligned(ar->episode, 4)) return MGX_ERR_INVALID_ARGUMENT;
        ka.pool_size = ar->pool_size; ka.first_env = ar->first_env;
        ka.pool_magic = ar->pool_size > 1 ? ~0ull / (uint64_t)ar->pool_size + 1ull : 0ull;
        ka.pool_grid = reinterpret_cast<const uint8_t *>(ar->pool_grid);
        ka.pool_agents = ar->pool_agents; ka.pool_aux = ar->pool_aux; ka.episode = ar->episode; ka.was_reset = ar->was_reset;
    }
    ka.grid = reinterpret_cast<uint8_t *>(sa.grid); ka.agents = sa.agents; ka.rng = sa.rng; ka.step_count = sa.step_count;
    ka.aux = sa.aux; ka.obs = sa.obs; ka.dir = sa.dir; ka.reward = sa.reward; ka.terminated = sa.terminated;
    ka.truncated = sa.truncated; ka.err = sa.err;
    ka.T = p->max_steps;
    ka.granules = p->action_granules; ka.done = p->done; ka.pctrl = p->ctrl;
    ka.timeout_ticks = (uint32_t)p->timeout_ms * 100000u;              // s_memrealtime: 100 MHz
#if MGX_SPANS
    {   // (spans build: one record per (step, wavefront) of this launch instead of one per wavefront)
        const long long nrec = (long long)nwg * (threads / 64) * std::min<long long>(p->max_steps, 4096);
        const int base = g_span_next;
        rc = launch(3, ka, threads, lds, nwg, static_cast<hipStream_t>(stream));
        g_span_next = (int)std::min<long long>((long long)kSpanCap, base + nrec);
        return rc;
    }
#endif
    return launch(3, ka, threads, lds, nwg, static_cast<hipStream_t>(stream));
}

void mgx_internal_set_hip_error(int e) { g_last_hip_error = e; }   // (not in include/mgx.h: mgx_layout_gen.hip / mgx_aux.hip report
                                                                    // their failed launches through mgx_last_hip_error() too)

}  // extern "C"
