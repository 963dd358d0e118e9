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

#include "mgx_rules.h"

namespace {

using namespace mgx;

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

// numpy PCG64 + its next_uint32 buffer: s = {state_lo, state_hi, inc_lo, inc_hi}, buf = has_uint32 << 32 | uinteger
struct NpGen {
    uint64_t s[4];
    uint64_t buf;
};

__device__ __forceinline__ uint32_t np_next32(NpGen &g) {
    if (g.buf >> 32) { const uint32_t r = (uint32_t)g.buf; g.buf = r; return r; }      // has_uint32 = 0 (uinteger stays)
    typedef unsigned __int128 u128;
    const u128 mult = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    const u128 st = (((u128)g.s[1] << 64) | g.s[0]) * mult + (((u128)g.s[3] << 64) | g.s[2]);
    g.s[0] = (uint64_t)st; g.s[1] = (uint64_t)(st >> 64);
    const uint64_t x = g.s[1] ^ g.s[0];
    const unsigned rot = (unsigned)(g.s[1] >> 58);
    const uint64_t next = (x >> rot) | (x << ((64u - rot) & 63u));
    g.buf = (1ull << 32) | (next >> 32);
    return (uint32_t)next;
}

// Generator.integers(lo, hi) for int64 scalars with hi - lo <= 2^32 (Lemire, distributions.c)
__device__ __forceinline__ int np_integers(NpGen &g, int lo, int hi) {
    const uint32_t rng = (uint32_t)(hi - 1 - lo);
    if (rng == 0) return lo;                                         // no draw
    const uint32_t rng_excl = rng + 1u;
    uint64_t m = (uint64_t)np_next32(g) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) { m = (uint64_t)np_next32(g) * rng_excl; leftover = (uint32_t)m; }
    }
    return lo + (int)(m >> 32);
}

constexpr int kMaxObjects = 4;

// What the owning lane knows about its env while placing: the blank template (global, read-only), the objects placed so
// far (registers) and the agents' positions (LDS, 2 bytes per agent).
struct Placer {
    const uint8_t *blank;
    int W, H, A;
    int n_obj;
    uint32_t obj_pos[kMaxObjects];        // x | y << 8
    uint32_t obj_cell[kMaxObjects];       // type | color << 8 | state << 16
    uint8_t *apos;                        // LDS: [A][2] (x, y); 0xff = not on the grid (-1)

    __device__ int type_at(int x, int y) const {
        int t = blank[(y * W + x) * 3];
        const uint32_t p = (uint32_t)x | ((uint32_t)y << 8);
#pragma unroll
        for (int k = 0; k < kMaxObjects; ++k) t = (k < n_obj && obj_pos[k] == p) ? (int)(obj_cell[k] & 0xff) : t;
        return t;
    }
    __device__ void put(int x, int y, uint32_t cell) {
        const uint32_t p = (uint32_t)x | ((uint32_t)y << 8);
#pragma unroll
        for (int k = 0; k < kMaxObjects; ++k)
            if (k == n_obj) { obj_pos[k] = p; obj_cell[k] = cell; }
        ++n_obj;
    }
    // base.py:604-669 place_obj: returns the position as x | y << 8
    __device__ uint32_t place(NpGen &g, int tx, int ty, int sw, int sh, bool next_to) const {
        tx = max(tx, 0); ty = max(ty, 0);
        const int xhi = min(tx + sw, W), yhi = min(ty + sh, H);
        for (;;) {
            const int x = np_integers(g, tx, xhi), y = np_integers(g, ty, yhi);
            if (type_at(x, y) != T_EMPTY) continue;                                  // grid.get(*pos) is not None
            bool bad = false;
            for (int a = 0; a < A; ++a) {
                const int ax = apos[2 * a] == 0xff ? -1 : apos[2 * a], ay = apos[2 * a + 1] == 0xff ? -1 : apos[2 * a + 1];
                const int dx = x - ax, dy = y - ay;
                bad |= (dx == 0) & (dy == 0);                                        // an agent stands there
                bad |= next_to & (dx * dx + dy * dy <= 1);                           // reject_next_to: norm <= 1
            }
            if (bad) continue;
            return (uint32_t)x | ((uint32_t)y << 8);
        }
    }
};

__global__ __launch_bounds__(64) void reset_generate_kernel(const GenArgs a) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x;
    const int W = a.sp.width, H = a.sp.height, A = a.sp.num_agents, HW3 = H * W * 3;
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
    // (1) the blank layout into every finished env's grid, all lanes copying (dwords when everything is aligned)
    const bool dw = ((HW3 | (int)(uintptr_t)a.gen.blank | (int)(uintptr_t)a.grid) & 3) == 0;
    for (uint64_t m = mask; m != 0; m &= m - 1) {
        uint8_t *dst = a.grid + (e0 + __builtin_ctzll(m)) * HW3;
        if (dw) {
            for (int i = lane; i < HW3 / 4; i += 64)
                reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(a.gen.blank)[i];
        } else {
            for (int i = lane; i < HW3; i += 64) dst[i] = a.gen.blank[i];
        }
    }
    if (!done) return;
    // (2) the owning lane: Agent.reset (agent.py:120-133) + _gen_grid
    NpGen lay, npr;
    uint64_t *gs = a.gen.gen_state + b * 6;
    for (int k = 0; k < 4; ++k) { lay.s[k] = gs[k]; npr.s[k] = a.rng[b * 4 + k]; }
    lay.buf = gs[4]; npr.buf = gs[5];
    Placer P;
    P.blank = a.gen.blank; P.W = W; P.H = H; P.A = A; P.n_obj = 0;
    P.apos = lds + lane * (2 * A);
    for (int k = 0; k < kMaxObjects; ++k) { P.obj_pos[k] = 0xffffffffu; P.obj_cell[k] = 0; }
    uint8_t *grid = a.grid + b * HW3;
    uint64_t *rows = reinterpret_cast<uint64_t *>(a.agents) + b * A;
    auto write_row = [&](int i, int x, int y, int d) {                              // colours cycle (constants.py:77-82)
        rows[i] = (uint64_t)(i % 6) | ((uint64_t)d << 8) | ((uint64_t)x << 16) | ((uint64_t)y << 24) | ((uint64_t)CELL_EMPTY << 40);
    };
    uint4 aux = {0, 0, 0, 0};
    if (a.gen.kind == MGX_GEN_EMPTY_FIXED) {                                        // empty.py:164-167
        for (int i = 0; i < A; ++i) write_row(i, a.gen.start_x, a.gen.start_y, a.gen.start_dir);
    } else if (a.gen.kind == MGX_GEN_EMPTY_RANDOM) {                                // empty.py:168-169: place_agent(agent)
        for (int i = 0; i < A; ++i) { P.apos[2 * i] = 0xff; P.apos[2 * i + 1] = 0xff; }
        for (int i = 0; i < A; ++i) {
            const uint32_t p = P.place(lay, 0, 0, W, H, false);
            P.apos[2 * i] = (uint8_t)p; P.apos[2 * i + 1] = (uint8_t)(p >> 8);
            write_row(i, p & 0xff, p >> 8, np_integers(lay, 0, 4));
        }
    } else {                                                                         // blockedunlockpickup.py:142-164
        const int rs = a.gen.room_size;
        for (int i = 0; i < A; ++i) { P.apos[2 * i] = (uint8_t)((rs - 1) + rs / 2); P.apos[2 * i + 1] = (uint8_t)(rs / 2); }   // roomgrid.py:232-236
        const uint32_t box_color = (uint32_t)np_integers(lay, 0, 6);
        uint32_t p = P.place(lay, rs - 1, 0, rs, rs, true);                          // box in the right room
        P.put(p & 0xff, p >> 8, (uint32_t)T_BOX | (box_color << 8));
        const uint32_t door_color = (uint32_t)np_integers(lay, 0, 6);
        const int door_x = rs - 1, door_y = np_integers(npr, 1, rs - 1);            // roomgrid.py:104-106: env.np_random
        P.put(door_x, door_y, (uint32_t)T_DOOR | (door_color << 8) | ((uint32_t)S_LOCKED << 16));
        P.put(door_x - 1, door_y, (uint32_t)T_BALL | ((uint32_t)np_integers(lay, 0, 6) << 8));
        p = P.place(lay, 0, 0, rs, rs, true);                                        // key in the left room
        P.put(p & 0xff, p >> 8, (uint32_t)T_KEY | (door_color << 8));
        for (int i = 0; i < A; ++i) {                                                // roomgrid.py:376-404
            for (;;) {
                P.apos[2 * i] = 0xff; P.apos[2 * i + 1] = 0xff;
                p = P.place(lay, 0, 0, rs, rs, false);
                const int x = (int)(p & 0xff), y = (int)(p >> 8);
                P.apos[2 * i] = (uint8_t)x; P.apos[2 * i + 1] = (uint8_t)y;
                const int d = np_integers(lay, 0, 4);
                const int t = P.type_at(x + dir_dx(d), y + dir_dy(d));
                if (t == T_EMPTY || t == T_WALL) { write_row(i, x, y, d); break; }
            }
        }
        for (int k = 0; k < kMaxObjects; ++k) {
            uint8_t *c = grid + (((P.obj_pos[k] >> 8) & 0xff) * W + (P.obj_pos[k] & 0xff)) * 3;
            c[0] = (uint8_t)P.obj_cell[k]; c[1] = (uint8_t)(P.obj_cell[k] >> 8); c[2] = (uint8_t)(P.obj_cell[k] >> 16);
        }
        aux.x = (uint32_t)T_BOX | (box_color << 8);                                  // the target box `self.obj` (include/mgx.h)
    }
    if (a.aux) reinterpret_cast<uint4 *>(a.aux)[b] = aux;
    for (int k = 0; k < 4; ++k) { gs[k] = lay.s[k]; a.rng[b * 4 + k] = npr.s[k]; }
    gs[4] = lay.buf; gs[5] = npr.buf;
    a.step_count[b] = 0;                                                             // base.py:292
    a.episode[b] += 1;
}

int g_gen_hip_error = 0;

inline bool misaligned(const void *p, uintptr_t al) { return (reinterpret_cast<uintptr_t>(p) & (al - 1)) != 0; }

}  // namespace

extern "C" {

int mgx_reset_generate(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, uint8_t *grid, uint8_t *agents,
                       uint64_t *rng, int32_t *step_count, uint8_t *aux, int32_t *episode, uint8_t *was_reset, void *stream) {
    if (!spec || !gen || batch < 0) return MGX_ERR_INVALID_ARGUMENT;
    if (spec->width < 3 || spec->height < 3 || spec->num_agents < 1 || spec->num_agents > MGX_MAX_AGENTS)
        return MGX_ERR_INVALID_ARGUMENT;
    if (spec->width > 254 || spec->height > 254) return MGX_ERR_UNSUPPORTED;         // positions are bytes, 0xff = off the grid
    if (batch == 0) return MGX_OK;
    if (!gen->blank || !gen->gen_state || !grid || !agents || !rng || !step_count || !episode) return MGX_ERR_INVALID_ARGUMENT;
    if (misaligned(agents, 8) || misaligned(rng, 8) || misaligned(gen->gen_state, 8) || misaligned(aux, 16)
        || misaligned(step_count, 4) || misaligned(episode, 4))
        return MGX_ERR_INVALID_ARGUMENT;
    switch (gen->kind) {
    case MGX_GEN_EMPTY_FIXED:
        if (gen->start_x < 0 || gen->start_x >= spec->width || gen->start_y < 0 || gen->start_y >= spec->height
            || gen->start_dir < 0 || gen->start_dir > 3)
            return MGX_ERR_INVALID_ARGUMENT;
        break;
    case MGX_GEN_EMPTY_RANDOM:
        break;
    case MGX_GEN_BLOCKEDUNLOCKPICKUP:
        if (gen->room_size < 4 || spec->width != 2 * gen->room_size - 1 || spec->height != gen->room_size || !aux)
            return MGX_ERR_INVALID_ARGUMENT;
        break;
    default:
        return MGX_ERR_UNSUPPORTED;
    }
    const int64_t blocks = (batch + 63) / 64;
    if (blocks > INT_MAX) return MGX_ERR_UNSUPPORTED;
    GenArgs ga{*spec, batch, *gen, grid, agents, rng, step_count, aux, episode, was_reset};
    hipLaunchKernelGGL(reset_generate_kernel, dim3((unsigned)blocks), dim3(64), (size_t)(64 * 2 * spec->num_agents),
                       static_cast<hipStream_t>(stream), ga);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_gen_hip_error = (int)e; return MGX_ERR_LAUNCH; }
    return MGX_OK;
}

}  // extern "C"
