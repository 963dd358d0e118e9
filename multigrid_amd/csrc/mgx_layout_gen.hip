// mgx_layout_gen.hip -- on-device episode starts: mgx_reset_generate.
//
// The reference generates an episode's initial state in MultiGridEnv.reset -> _gen_grid (multigrid/base.py:250-301) by
// rejection sampling against the grid with numpy draws.  Here every finished env of the batch does the same ON THE DEVICE,
// one lane per env, draw for draw:
//   numpy Generator(PCG64).integers(lo, hi)   numpy/random/_bounded_integers.pyx _rand_int64 -> distributions.c
//                                             random_bounded_uint64_fill -> buffered_bounded_lemire_uint32 over PCG64's
//                                             next_uint32 (pcg64.h pcg64_next32: low half first, high half buffered)
//   place_obj / place_agent                   multigrid/base.py:604-697
//   reject_next_to, place_in_room, RoomGrid.place_agent      multigrid/core/roomgrid.py:45-50, 238-259, 376-404
//   BlockedUnlockPickupEnv._gen_grid          multigrid/envs/blockedunlockpickup.py:142-164 (door row drawn from env.np_random,
//                                             roomgrid.py:104-106 -- the same stream the action order is drawn from)
//   EmptyEnv._gen_grid                        multigrid/envs/empty.py:151-170 (fixed start, or place_agent over the grid)
// Two generators per env, as in the reference: the construction-time generator that drives every
// _rand_* placement draw (gen_state[0..4]) and env.np_random (the `rng` tensor + its 32-bit buffer, gen_state[5]).
// Given generators in the same state this produces byte for byte what multigrid_amd/layouts.py (pinned by the reset
// fixtures of the real reference) and oracle/mgx_layout_oracle.c produce; tests/test_layout_gen.py.
//
// Resets are rare (an env restarts every few hundred steps), so the kernel is a scan: one lane per env tests `done`
// (16 B of agent rows + 4 B per env), a wavefront that has finished envs copies the blank room layout into their grids
// with all 64 lanes, and the owning lanes then run the serial placement, which only reads the L2-resident blank template
// and the handful of objects they placed themselves (kept in registers).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "mgx_layout_gen.h"

extern "C" void mgx_internal_set_hip_error(int e);      // mgx_kernels.hip: what mgx_last_hip_error() reports

namespace {

using namespace mgx;
using namespace mgx_gen;

struct GenArgs {
    MgxSpec sp;
    int64_t batch;
    MgxLayoutGen gen;
    uint8_t *grid;
    uint8_t *agents;
    uint64_t *rng;
    int32_t *step_count;
    uint8_t *aux;
    int32_t *episode;
    uint8_t *was_reset;
};

__global__ __launch_bounds__(64) void reset_generate_kernel(const GenArgs a) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x;
    const int W = a.sp.width, H = a.sp.height, A = a.sp.num_agents, HWB = H * W * kCellBytes;
    const int64_t e0 = (int64_t)blockIdx.x * 64;
    const int64_t b = e0 + lane;
    bool done = false;
    if (b < a.batch) {
        const uint64_t *rows = reinterpret_cast<const uint64_t *>(a.agents) + b * A;
        bool all_term = true;
        for (int k = 0; k < A; ++k) all_term &= row_term(rows[k]);
        done = all_term || a.step_count[b] >= a.sp.max_steps;                        // base.py:534-539
        if (a.was_reset) a.was_reset[b] = (uint8_t)done;
    }
    const uint64_t mask = __builtin_amdgcn_ballot_w64(done);
    if (mask == 0) return;
    copy_blank(a.gen, a.grid, e0, HWB, mask, lane);                                  // (1) all lanes
    // (2) overwrites cells that OTHER lanes of this wavefront have just stored: their stores must have left the wave first
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0x0F70);                                              // vmcnt(0)
    __builtin_amdgcn_wave_barrier();
    if (!done) return;
    NpGen lay, npr;                                                                  // (2) the owning lane
    uint64_t *gs = a.gen.gen_state + b * 6;
    for (int k = 0; k < 4; ++k) { lay.s[k] = gs[k]; npr.s[k] = a.rng[b * 4 + k]; }
    lay.buf = gs[4]; npr.buf = gs[5];
    const uint4 aux = generate_episode(a.gen, W, H, A, lay, npr, lds + lane * (2 * A), a.grid + b * HWB,
                                       reinterpret_cast<uint64_t *>(a.agents) + b * A);
    if (a.aux) reinterpret_cast<uint4 *>(a.aux)[b] = aux;
    for (int k = 0; k < 4; ++k) { gs[k] = lay.s[k]; a.rng[b * 4 + k] = npr.s[k]; }
    gs[4] = lay.buf; gs[5] = npr.buf;
    a.step_count[b] = 0;                                                             // base.py:292
    a.episode[b] += 1;
}


// MgxGenStage.candidates (include/mgx.h): kGroupLanes lanes per (env, value of the generator's env.np_random draw) -- a group makes ONE
// candidate, its lanes the tries of every place_obj call at once (mgx_layout_gen.h: GroupCtx; rounds 5: one lane per candidate, the
// launch as long as the unluckiest lane of 64).  An env whose candidate k is not its CURRENT episode's successor gets it made from the
// layout stream as that episode's generation left it.  Launched between two steps on their stream (external == 2): nothing reads or
// writes a slot beside this launch, the kernel boundary publishes it.  KIND is a template argument so that each instantiation
// carries ONE generator's code.
template <int KIND>
__global__ __launch_bounds__(64) void stage_candidates_kernel(const GenArgs a) {
    extern __shared__ uint8_t lds[];
    constexpr int GS = kGroupLanes, CPW = 64 / GS;                                   // candidates per wavefront
    const int lane = threadIdx.x, grp = lane / GS, j = lane % GS;
    const int W = a.sp.width, H = a.sp.height, A = a.sp.num_agents, HWB = H * W * kCellBytes;
    const MgxGenStage &st = a.gen.stage;
    const int K = st.candidates;
    const int64_t s = (int64_t)blockIdx.x * CPW + grp, b = s / K;
    const int k = (int)(s - b * K);
    const bool go = b < a.batch && st.tag[b * 4 + k] != a.episode[b];
    if (__builtin_amdgcn_ballot_w64(go) == 0) return;
    MgxLayoutGen gen = a.gen;
    gen.kind = KIND;
    uint8_t *const cand_grid = reinterpret_cast<uint8_t *>(st.grid) + s * HWB;
    // (the layout stream's words and the lane's jump constants are requested BEFORE the blank copy: one memory round trip for the three)
    NpGen lay, npr;
    GroupCtx gc;
    {
        const uint64_t *gs = gen.gen_state + (go ? b : 0) * 6;
        for (int q = 0; q < 4; ++q) { lay.s[q] = gs[q]; npr.s[q] = 0; }
        lay.buf = gs[4]; npr.buf = 0;                                                // (npr: not drawn from -- door_row is given)
        group_init(gc, lane, lay);
    }
    if (go) {                                                                        // the blank layout: the group copies its own
        const uint8_t *blank = reinterpret_cast<const uint8_t *>(gen.blank);
        if (((HWB | (int)(uintptr_t)blank | (int)(uintptr_t)st.grid) & 3) == 0) {
            for (int i = j; i < HWB / 4; i += GS)
                reinterpret_cast<uint32_t *>(cand_grid)[i] = reinterpret_cast<const uint32_t *>(blank)[i];
        } else {
            for (int i = j; i < HWB; i += GS) cand_grid[i] = blank[i];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                           // (every lane overwrites cells the group's other lanes stored)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_wave_barrier();
    if (!go) return;
    const uint4 naux = generate_episode<true>(gen, W, H, A, lay, npr, lds + grp * (2 * A), cand_grid,
                                              reinterpret_cast<uint64_t *>(st.agents) + s * A, k + 1, &gc);
    if (j != 0) return;
    if (st.aux) reinterpret_cast<uint4 *>(st.aux)[s] = naux;
    uint64_t *const words = st.words + s * 6;
    for (int q = 0; q < 4; ++q) words[q] = lay.s[q];
    words[4] = lay.buf;
    st.tag[b * 4 + k] = a.episode[b];
}

inline bool misaligned(const void *p, uintptr_t al) { return (reinterpret_cast<uintptr_t>(p) & (al - 1)) != 0; }

}  // namespace

// mgx_stage_generate with MgxGenStage.candidates > 0 (mgx_kernels.hip validated the arguments)
extern "C" int mgx_internal_stage_candidates(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, const int32_t *episode,
                                             void *stream) {
    constexpr int64_t cpw = 64 / kGroupLanes;
    const int64_t blocks = (batch * gen->stage.candidates + cpw - 1) / cpw;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    GenArgs ga{*spec, batch, *gen, nullptr, nullptr, nullptr, nullptr, nullptr, const_cast<int32_t *>(episode), nullptr};
    const size_t lds = (size_t)(64 * 2 * spec->num_agents);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    switch (gen->kind) {
    case MGX_GEN_EMPTY_FIXED: hipLaunchKernelGGL(stage_candidates_kernel<MGX_GEN_EMPTY_FIXED>, dim3((unsigned)blocks), dim3(64), lds, hs, ga); break;
    case MGX_GEN_EMPTY_RANDOM: hipLaunchKernelGGL(stage_candidates_kernel<MGX_GEN_EMPTY_RANDOM>, dim3((unsigned)blocks), dim3(64), lds, hs, ga); break;
    case MGX_GEN_BLOCKEDUNLOCKPICKUP: hipLaunchKernelGGL(stage_candidates_kernel<MGX_GEN_BLOCKEDUNLOCKPICKUP>, dim3((unsigned)blocks), dim3(64), lds, hs, ga); break;
    case MGX_GEN_REDBLUEDOORS: hipLaunchKernelGGL(stage_candidates_kernel<MGX_GEN_REDBLUEDOORS>, dim3((unsigned)blocks), dim3(64), lds, hs, ga); break;
    case MGX_GEN_LOCKEDHALLWAY: hipLaunchKernelGGL(stage_candidates_kernel<MGX_GEN_LOCKEDHALLWAY>, dim3((unsigned)blocks), dim3(64), lds, hs, ga); break;
    default: return MGX_ERR_UNSUPPORTED;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mgx_internal_set_hip_error((int)e); return MGX_ERR_LAUNCH; }
    return MGX_OK;
}

extern "C" {

int mgx_reset_generate(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, MgxCell *grid, uint8_t *agents,
                       uint64_t *rng, int32_t *step_count, uint8_t *aux, int32_t *episode, uint8_t *was_reset, void *stream) {
    if (!spec || !gen || batch < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->width < 3 || spec->height < 3 || spec->num_agents < 1 || spec->num_agents > MGX_MAX_AGENTS)
        return MGX_ERR_INVALID_ARGUMENT;
    if (spec->width > 254 || spec->height > 254) return MGX_ERR_UNSUPPORTED;         // positions are bytes, 0xff = off the grid
    if (spec->cell_bytes == 1) return MGX_ERR_UNSUPPORTED;                           // (generation writes 16-bit cells)
    if (batch == 0) return MGX_OK;
    if (!gen->blank || !gen->gen_state || !grid || !agents || !rng || !step_count || !episode) return MGX_ERR_INVALID_ARGUMENT;
    if (misaligned(agents, 8) || misaligned(rng, 8) || misaligned(gen->gen_state, 8) || misaligned(aux, 16)
        || misaligned(step_count, 4) || misaligned(episode, 4))
        return MGX_ERR_INVALID_ARGUMENT;
    const int rc = check_layout_gen(spec, gen);
    if (rc) return rc;
    if (spec->env_kind != MGX_KIND_EMPTY && !aux) return MGX_ERR_INVALID_ARGUMENT;
    const int64_t blocks = (batch + 63) / 64;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    GenArgs ga{*spec, batch, *gen, reinterpret_cast<uint8_t *>(grid), agents, rng, step_count, aux, episode, was_reset};
    hipLaunchKernelGGL(reset_generate_kernel, dim3((unsigned)blocks), dim3(64), (size_t)(64 * 2 * spec->num_agents),
                       static_cast<hipStream_t>(stream), ga);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mgx_internal_set_hip_error((int)e); return MGX_ERR_LAUNCH; }
    return MGX_OK;
}

}  // extern "C"
