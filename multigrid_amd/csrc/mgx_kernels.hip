// mgx_kernels.hip -- the C ABI of libmgx.so (include/mgx.h): argument checks, launch geometry, dispatch to the fused
// kernel's per-view-size translation units (mgx_fused.h / mgx_fused_inst.hip).
#include "mgx_fused.h"

#if MGX_SINGLE_TU      // (tools' builds that need the kernels and the host code in one module, e.g. -DMGX_TIMESTAMPS=1)
#define MGX_INST_V 3
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#define MGX_INST_V 5
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#define MGX_INST_V 7
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#define MGX_INST_V 9
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#define MGX_INST_V 11
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#define MGX_INST_V 13
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#define MGX_INST_V 15
#include "mgx_fused_inst.hip"
#undef MGX_INST_V
#endif

namespace {

using namespace mgx;
using namespace mgx_fused;

// mode 0: gen_obs, 1: one step, 2: rollout
int launch(int mode, const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream);

int g_last_hip_error = 0;
#if MGX_DEBUG_KNOBS
int g_debug_skip = 0;
int g_debug_G = 0;
int g_debug_wpb = 0;
#else
constexpr int g_debug_skip = 0, g_debug_G = 0, g_debug_wpb = 0;
#endif

#if MGX_BOUNDS_CHECK
int32_t *g_bounds = nullptr;        // device: [0] violations, [1] last site (checked build only; allocated on first launch)
#endif

int launch(int mode, const KernelArgs &ka_in, int threads, int lds_bytes, int64_t nwg, hipStream_t stream) {
    KernelArgs ka = ka_in;
#if MGX_BOUNDS_CHECK
    if (!g_bounds) {
        if (hipMalloc(reinterpret_cast<void **>(&g_bounds), 8) != hipSuccess) return MGX_ERR_LAUNCH;
        if (hipMemset(g_bounds, 0, 8) != hipSuccess) return MGX_ERR_LAUNCH;
    }
    ka.bounds = g_bounds;
#endif
    switch (ka.sp.view_size) {
#define MGX_CASE(V) case V: return launch_v##V(mode, ka, threads, lds_bytes, nwg, stream, &g_last_hip_error);
    MGX_FOR_EACH_VIEW(MGX_CASE)
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
    if (sp->env_kind < MGX_KIND_EMPTY || sp->env_kind > MGX_KIND_LOCKEDHALLWAY) return MGX_ERR_UNSUPPORTED;
    if (wave_lds_bytes(*sp, 1, roll, one_hot, obs_only) > kLdsPerCU) return MGX_ERR_UNSUPPORTED;   // one env must fit one CU's LDS (the
                                                                                     // rollout carve is the larger one)
    return MGX_OK;
}

int fill_args(KernelArgs &ka, const MgxSpec *sp, int64_t batch, int &threads, int &lds_bytes, int64_t &nwg,
              bool roll = false, bool one_hot = false, bool obs_only = false) {
    ka.sp = *sp;
    ka.batch = batch;
    ka.Gw = g_debug_G > 0 ? g_debug_G : choose_Gw(*sp, batch, roll, one_hot, obs_only);
    const int max_gw = slots_per_wave(sp->view_size, roll || obs_only) / sp->num_agents;
    if (ka.Gw > max_gw) ka.Gw = max_gw;
    if (ka.Gw < 1) ka.Gw = 1;
    while (ka.Gw > 1 && wave_lds_bytes(*sp, ka.Gw, roll, one_hot, obs_only) > kLdsPerCU) --ka.Gw;
    ka.dbg = g_debug_skip;
    // a grid tensor beyond half of the 256 MiB Infinity Cache is streamed (nt tile loads): mgx_fused.h, P0
    ka.flags = (batch * (int64_t)sp->width * sp->height * kCellBytes > (int64_t)128 << 20) ? 1 : 0;
    ka.vpw = slots_in_use(*sp, ka.Gw, roll || obs_only);
    ka.inv_A = (65536 + sp->num_agents - 1) / sp->num_agents;
    ka.wave_lds = wave_lds_bytes(*sp, ka.Gw, roll, one_hot, obs_only);
    struct { int total; } p{ka.wave_lds};
    // wavefronts bundled per workgroup: 2 packs a CU's 160 KiB of LDS tighter than 4 once the chip is full (measured
    // 403 vs 425 us at 1M envs); below that, fewer and larger workgroups launch faster (9.9 vs 10.5 us at 4096 envs)
    int wpb = ((batch + ka.Gw - 1) / ka.Gw >= 16384) ? 2 : 4;
    while (wpb > 1 && wpb * p.total > 64 * 1024) wpb >>= 1;
    if (g_debug_wpb > 0) wpb = g_debug_wpb;
    threads = 64 * wpb;
    lds_bytes = wpb * p.total;
    const int64_t nwaves = (batch + ka.Gw - 1) / ka.Gw;
    // latency regime: LDS-DMA tile loads, 32 view slots with unpacked cell registers (mgx_fused.h: DMA instantiations)
    if (nwaves <= 2048 && !(ka.flags & 1) && ka.Gw * sp->num_agents <= kSlotsLatency) ka.flags |= 2;
    nwg = (nwaves + wpb - 1) / wpb;
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

#if MGX_DEBUG_KNOBS
// Profiling aid, only in the tools' build (lib/libmgx_dbg.so; never in libmgx.so): bit p set = the fused kernel skips
// phase Pp.  Results are then meaningless; tools/phase_probe.py uses it to attribute kernel time to phases.
void mgx_debug_skip_phases(int mask) { g_debug_skip = mask; }
void mgx_debug_set_envs_per_wavefront(int G) { g_debug_G = G; }
void mgx_debug_set_waves_per_workgroup(int n) { g_debug_wpb = n; }
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
#if MGX_TIMESTAMPS
int mgx_debug_read_span(unsigned long long *out, int nwaves) {            // [nwaves][2] of the last launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 2 * nwaves) == hipSuccess ? 0 : -1;
}
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
    rc = fill_args(ka, spec, batch, threads, lds, nwg);
    if (rc) return rc;
    out->envs_per_wavefront = ka.Gw;
    out->envs_per_workgroup = ka.Gw * (threads / 64);
    out->threads_per_workgroup = threads;
    out->workgroups = (int32_t)nwg;
    out->lds_bytes = lds;
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

static int step_common(bool roll, bool one_hot, const MgxSpec *spec, int64_t batch, int32_t steps, const MgxAutoReset *ar,
                       const MgxLayoutGen *gen, int32_t *gen_episode, uint8_t *gen_was_reset,
                       MgxCell *grid, uint8_t *agents, uint64_t *rng, int32_t *step_count, const int8_t *actions,
                       uint8_t *aux, uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated,
                       uint8_t *truncated, int32_t *err, void *stream) {
    int rc = check_spec(spec, batch, roll, one_hot);
    if (rc) return rc;
    if (steps < 0 || (roll && one_hot)) return MGX_ERR_INVALID_ARGUMENT;
    if (batch == 0 || steps == 0) return MGX_OK;
    if (!grid || !agents || !step_count || !actions || !obs || !reward || !terminated || !truncated)
        return MGX_ERR_INVALID_ARGUMENT;
    if (spec->num_agents > 1 && !rng) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->env_kind != MGX_KIND_EMPTY && !aux) return MGX_ERR_INVALID_ARGUMENT;
    if (misaligned(grid, 16) || misaligned(agents, 8) || misaligned(obs, 16) || misaligned(rng, 8)
        || misaligned(reward, 8) || misaligned(step_count, 4) || misaligned(err, 4) || misaligned(aux, 16))
        return MGX_ERR_INVALID_ARGUMENT;
    KernelArgs ka{};
    if (ar) {
        if (ar->pool_size < 1 || ar->first_env < 0 || !ar->pool_grid || !ar->pool_agents || !ar->episode)
            return MGX_ERR_INVALID_ARGUMENT;
        if (spec->env_kind != MGX_KIND_EMPTY && !ar->pool_aux) return MGX_ERR_INVALID_ARGUMENT;
        if (misaligned(ar->pool_agents, 8) || misaligned(ar->pool_aux, 16) || misaligned(ar->episode, 4))
            return MGX_ERR_INVALID_ARGUMENT;
        ka.pool_size = ar->pool_size; ka.first_env = ar->first_env; ka.pool_grid = reinterpret_cast<const uint8_t *>(ar->pool_grid);
        ka.pool_agents = ar->pool_agents; ka.pool_aux = ar->pool_aux; ka.episode = ar->episode;
        ka.was_reset = ar->was_reset;
    }
    int threads = 0, lds = 0; int64_t nwg = 0;
    rc = fill_args(ka, spec, batch, threads, lds, nwg, roll, one_hot);
    if (rc) return rc;
    ka.grid = reinterpret_cast<uint8_t *>(grid); ka.agents = agents; ka.rng = rng; ka.step_count = step_count; ka.actions = actions;
    ka.aux = aux; ka.obs = obs; ka.dir = dir; ka.reward = reward; ka.terminated = terminated;
    ka.truncated = truncated; ka.err = err;
    ka.T = roll ? steps : 1;
    if (gen) {
        if (roll || one_hot || ar || !gen->blank || !gen->gen_state || !gen_episode || !rng) return MGX_ERR_INVALID_ARGUMENT;
        if (misaligned(gen->gen_state, 8) || misaligned(gen_episode, 4)) return MGX_ERR_INVALID_ARGUMENT;
        if (spec->width > 254 || spec->height > 254) return MGX_ERR_UNSUPPORTED;
        switch (gen->kind) {
        case MGX_GEN_EMPTY_FIXED:
            if (spec->env_kind != MGX_KIND_EMPTY || gen->start_x < 0 || gen->start_x >= spec->width || gen->start_y < 0
                || gen->start_y >= spec->height || gen->start_dir < 0 || gen->start_dir > 3)
                return MGX_ERR_INVALID_ARGUMENT;
            break;
        case MGX_GEN_EMPTY_RANDOM:
            if (spec->env_kind != MGX_KIND_EMPTY) return MGX_ERR_INVALID_ARGUMENT;
            break;
        case MGX_GEN_BLOCKEDUNLOCKPICKUP:
            if (spec->env_kind != MGX_KIND_BLOCKEDUNLOCKPICKUP || gen->room_size < 4
                || spec->width != 2 * gen->room_size - 1 || spec->height != gen->room_size)
                return MGX_ERR_INVALID_ARGUMENT;
            break;
        default: return MGX_ERR_UNSUPPORTED;
        }
        ka.gen = *gen; ka.episode = gen_episode; ka.was_reset = gen_was_reset;
        return launch(9, ka, threads, lds, nwg, static_cast<hipStream_t>(stream));
    }
    return launch(roll ? 2 : (one_hot ? 5 : 1), ka, threads, lds, nwg, static_cast<hipStream_t>(stream));
}

int mgx_step(const MgxSpec *spec, int64_t batch, MgxCell *grid, uint8_t *agents, uint64_t *rng,
             int32_t *step_count, const int8_t *actions, uint8_t *aux,
             uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
             int32_t *err, void *stream) {
    return step_common(false, false, spec, batch, 1, nullptr, nullptr, nullptr, nullptr, grid, agents, rng, step_count, actions, aux, obs, dir, reward,
                       terminated, truncated, err, stream);
}

int mgx_rollout(const MgxSpec *spec, int64_t batch, int32_t steps, MgxCell *grid, uint8_t *agents, uint64_t *rng,
                int32_t *step_count, const int8_t *actions, uint8_t *aux,
                uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                int32_t *err, void *stream) {
    return step_common(true, false, spec, batch, steps, nullptr, nullptr, nullptr, nullptr, grid, agents, rng, step_count, actions, aux, obs, dir, reward,
                       terminated, truncated, err, stream);
}

int mgx_step_autoreset(const MgxSpec *spec, int64_t batch, const MgxAutoReset *ar, MgxCell *grid, uint8_t *agents,
                       uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                       uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                       int32_t *err, void *stream) {
    if (!ar) return MGX_ERR_INVALID_ARGUMENT;
    return step_common(false, false, spec, batch, 1, ar, nullptr, nullptr, nullptr, grid, agents, rng, step_count, actions, aux, obs, dir, reward,
                       terminated, truncated, err, stream);
}

int mgx_rollout_autoreset(const MgxSpec *spec, int64_t batch, int32_t steps, const MgxAutoReset *ar, MgxCell *grid,
                          uint8_t *agents, uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                          uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                          int32_t *err, void *stream) {
    if (!ar) return MGX_ERR_INVALID_ARGUMENT;
    return step_common(true, false, spec, batch, steps, ar, nullptr, nullptr, nullptr, grid, agents, rng, step_count, actions, aux, obs, dir, reward,
                       terminated, truncated, err, stream);
}

int mgx_step_one_hot(const MgxSpec *spec, int64_t batch, const MgxAutoReset *ar, MgxCell *grid, uint8_t *agents,
                     uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                     uint8_t *obs_one_hot, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                     int32_t *err, void *stream) {
    return step_common(false, true, spec, batch, 1, ar, nullptr, nullptr, nullptr, grid, agents, rng, step_count, actions, aux, obs_one_hot, dir, reward,
                       terminated, truncated, err, stream);
}

int mgx_step_generate(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, MgxCell *grid, uint8_t *agents,
                      uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                      uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                      int32_t *err, int32_t *episode, uint8_t *was_reset, void *stream) {
    if (!gen) return MGX_ERR_INVALID_ARGUMENT;
    return step_common(false, false, spec, batch, 1, nullptr, gen, episode, was_reset, grid, agents, rng, step_count, actions,
                       aux, obs, dir, reward, terminated, truncated, err, stream);
}

}  // extern "C"
