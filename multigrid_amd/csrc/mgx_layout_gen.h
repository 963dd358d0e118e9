// mgx_layout_gen.h -- device-side episode generation shared by mgx_layout_gen.hip (mgx_reset_generate: its own launch) and
// the fused step kernel (mgx_step_generate: the finished envs of a step are regenerated in the tail of the same launch).
// What is restated (numpy's Generator.integers, place_obj, the two _gen_grid) and how it is pinned: mgx_layout_gen.hip.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "mgx_rules.h"

namespace mgx_gen {

using namespace mgx;

// Is (spec, gen) a combination the generators implement?  Shared by mgx_reset_generate and the fused step (MGX_OK or an error).
// (... and does a layout EXIST?  The reference's place_obj samples positions until one fits, without a bound (base.py:604-669,
// max_tries = inf): in a room too small for what goes into it, it never returns -- and neither would the lane that runs it here,
// which on a GPU means a hung device.  Such specs are refused: MGX_ERR_UNSUPPORTED.  Only those: a room that is exactly FULL is
// accepted -- the rejection sampling still ends with probability 1, as in the reference (round 6; ADVICE r5: the first form of
// this check asked for a spare cell and turned away configurations the reference runs).  The demands, by generator:
//   Empty-Random      the interior minus the goal holds the A agents (place_agent: distinct free cells, empty.py:164-170)
//   BlockedUnlockPickup  the left room holds the ball in front of the door, the key and the A agents (blockedunlockpickup.py:142-164;
//                     an agent's front cell must be free or a wall: with two objects in the room some direction always is)
//   RedBlueDoors      the middle room holds the A agents (redbluedoors.py:150-156)
//   LockedHallway     the hallway holds up to max_hallway_keys keys and the A agents; a side room up to max_keys_per_room keys for
//                     every time its colour comes up -- once with at most 6 rooms, ceil(rooms / 6) times beyond (the reference keys
//                     its rooms by door colour, locked_hallway.py:168-190)
//   Playground        unchanged (12 objects may draw one room, away from the agents' start; the reference itself gives up after
//                     1000 tries there: roomgrid.py:255))
inline int check_layout_gen(const MgxSpec *spec, const MgxLayoutGen *gen) {
    const int W = spec->width, H = spec->height, rs = gen->room_size, A = spec->num_agents;
    const int room = (rs - 2) * (rs - 2);                       // free cells of one room
    switch (gen->kind) {
    case MGX_GEN_EMPTY_RANDOM:
        if ((W - 2) * (H - 2) - 1 < A) return MGX_ERR_UNSUPPORTED;
        break;
    case MGX_GEN_BLOCKEDUNLOCKPICKUP:
        if (rs >= 4 && room < A + 2) return MGX_ERR_UNSUPPORTED;
        break;
    case MGX_GEN_REDBLUEDOORS:
        if ((W / 2 - 2) * (H - 2) < A) return MGX_ERR_UNSUPPORTED;
        break;
    case MGX_GEN_LOCKEDHALLWAY: {
        if (rs >= 4 && (rs - 2) * (H - 2) < A + gen->max_hallway_keys) return MGX_ERR_UNSUPPORTED;
        const int rooms = rs >= 4 ? 2 * ((H - 1) / (rs - 1)) : 0;
        if (rs >= 4 && room < gen->max_keys_per_room * ((rooms + 5) / 6)) return MGX_ERR_UNSUPPORTED;
        break;
    }
    case MGX_GEN_PLAYGROUND:
        if (rs >= 4 && room < 12 + A + 1) return MGX_ERR_UNSUPPORTED;                  // all 12 objects may draw the same room
        break;
    default:
        break;
    }
    switch (gen->kind) {
    case MGX_GEN_EMPTY_FIXED:
        if (spec->env_kind != MGX_KIND_EMPTY || gen->start_x < 0 || gen->start_x >= W || gen->start_y < 0 || gen->start_y >= H
            || gen->start_dir < 0 || gen->start_dir > 3)
            return MGX_ERR_INVALID_ARGUMENT;
        return MGX_OK;
    case MGX_GEN_EMPTY_RANDOM:
        return spec->env_kind == MGX_KIND_EMPTY ? MGX_OK : MGX_ERR_INVALID_ARGUMENT;
    case MGX_GEN_BLOCKEDUNLOCKPICKUP:
        return (spec->env_kind == MGX_KIND_BLOCKEDUNLOCKPICKUP && rs >= 4 && W == 2 * rs - 1 && H == rs) ? MGX_OK : MGX_ERR_INVALID_ARGUMENT;
    case MGX_GEN_REDBLUEDOORS:
        return (spec->env_kind == MGX_KIND_REDBLUEDOORS && W == 2 * H && W >= 8) ? MGX_OK : MGX_ERR_INVALID_ARGUMENT;
    case MGX_GEN_LOCKEDHALLWAY: {
        if (spec->env_kind != MGX_KIND_LOCKEDHALLWAY || rs < 4 || W != 3 * (rs - 1) + 1 || (H - 1) % (rs - 1) != 0)
            return MGX_ERR_INVALID_ARGUMENT;
        const int rows = (H - 1) / (rs - 1);
        if (rows < 1 || gen->max_hallway_keys < 1 || gen->max_keys_per_room < 1) return MGX_ERR_INVALID_ARGUMENT;
        return rows <= 8 ? MGX_OK : MGX_ERR_UNSUPPORTED;                 // at most 16 rooms (the hook's 16-bit door mask)
    }
    case MGX_GEN_PLAYGROUND: {
        if (spec->env_kind != MGX_KIND_EMPTY || rs < 4 || (W - 1) % (rs - 1) != 0 || (H - 1) % (rs - 1) != 0)
            return MGX_ERR_INVALID_ARGUMENT;
        const int rooms = ((W - 1) / (rs - 1)) * ((H - 1) / (rs - 1));
        return (rooms >= 1 && rooms <= 16) ? MGX_OK : MGX_ERR_UNSUPPORTED;
    }
    default:
        return MGX_ERR_UNSUPPORTED;
    }
}

// MgxGenStage.candidates this generator supports: one per value of its single env.np_random draw (0: not available)
inline int stage_candidates(const MgxLayoutGen *gen) {
    switch (gen->kind) {
    case MGX_GEN_EMPTY_FIXED: case MGX_GEN_EMPTY_RANDOM: case MGX_GEN_REDBLUEDOORS: case MGX_GEN_LOCKEDHALLWAY: return 1;
    case MGX_GEN_BLOCKEDUNLOCKPICKUP: return gen->room_size - 2 <= 4 ? gen->room_size - 2 : 0;
    default: return 0;
    }
}

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

// distributions.c random_interval (max <= 0xffffffff): masked rejection over next_uint32 -- what Generator.shuffle draws
__device__ __forceinline__ uint32_t np_interval(NpGen &g, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max, value;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    do { value = np_next32(g) & mask; } while (value > max);
    return value;
}

// ---------------------------------------------------------------------------------------------------------------
// Group-cooperative rejection sampling (round 6; stage_candidates_kernel, mgx_layout_gen.hip).  place_obj draws (x, y) until the
// cell fits: a serial loop of unknown length per lane, and a wavefront of 64 such lanes runs as long as its unluckiest one -- 6-7
// tries per call where one lane alone needs 1.2 (profiles/r5_candidates.txt).  Here kGroupLanes lanes share ONE candidate: lane j
// evaluates try j of the call -- PCG64 is an LCG, so the state j tries ahead is mult^j * state + (1 + ... + mult^(j-1)) * inc,
// one 128-bit multiply-add with per-lane constants -- and the group adopts the first try that fits (a ballot) together with the
// generator state behind it.  Everything else of the generation runs redundantly on the group's lanes (same values, same stores).
// A try is exactly two 32-bit draws = one 64-bit word, whichever half the stream stands at -- unless Lemire's bounded draw
// re-samples (probability < range / 2^32 per draw): a lane that sees that reports it, and the group re-runs the call serially.
constexpr int kGroupLanes = 8;

struct GroupCtx {
    int j, base;                  // this lane's try index; the group's first lane in the wavefront
    uint64_t a_lo, a_hi;          // mult^j
    uint64_t d_lo, d_hi;          // (1 + mult + ... + mult^(j-1)) * inc of THIS stream
};

__device__ __forceinline__ uint64_t pcg64_output(uint64_t lo, uint64_t hi) {        // XSL-RR of the (new) state
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

__device__ __forceinline__ void group_init(GroupCtx &c, int lane, const NpGen &g) {
    // {mult^j, 1 + mult + ... + mult^(j-1)} mod 2^128, mult = 0x2360ED051FC65DA44385DF649FCCF645 (PCG_DEFAULT_MULTIPLIER_128)
    static constexpr uint64_t T[kGroupLanes][4] = {
        {0x0000000000000001ull, 0x0000000000000000ull, 0x0000000000000000ull, 0x0000000000000000ull},
        {0x4385DF649FCCF645ull, 0x2360ED051FC65DA4ull, 0x0000000000000001ull, 0x0000000000000000ull},
        {0x529ED9EB20E0AE99ull, 0x17BCE35BDF69743Cull, 0x4385DF649FCCF646ull, 0x2360ED051FC65DA4ull},
        {0xEB5AE837ED42153Dull, 0x25F041404BD80E82ull, 0x9624B94FC0ADA4DFull, 0x3B1DD060FF2FD1E0ull},
        {0xD194DFBE42D45771ull, 0xF4DD417327DB7A9Bull, 0x817FA187ADEFBA1Cull, 0x610E11A14B07E063ull},
        {0x1712DD28EC4E2775ull, 0x16C406E9FBE6C01Full, 0x53148145F0C4118Dull, 0x55EB531472E35AFFull},
        {0x81AB1C97E7371089ull, 0x19B2ADD48DEFCDA8ull, 0x6A275E6EDD123902ull, 0x6CAF59FE6ECA1B1Eull},
        {0x3D56ECCA7FE71AEDull, 0x9C49933E0E7F5995ull, 0xEBD27B06C449498Bull, 0x866207D2FCB9E8C6ull}};
    typedef unsigned __int128 u128;
    c.j = lane & (kGroupLanes - 1);
    c.base = lane & ~(kGroupLanes - 1);
    c.a_lo = T[c.j][0]; c.a_hi = T[c.j][1];
    const u128 d = (((u128)T[c.j][3] << 64) | T[c.j][2]) * (((u128)g.s[3] << 64) | g.s[2]);
    c.d_lo = (uint64_t)d; c.d_hi = (uint64_t)(d >> 64);
}

// the generator as c.j tries (two 32-bit draws each) later: j whole words further, the same half pending (has_uint32 keeps its value;
// `uinteger` is always the high half of the last word made, pending or spent: distributions.h next_uint32)
__device__ __forceinline__ void group_skip(NpGen &t, const GroupCtx &c) {
    if (c.j == 0) return;
    typedef unsigned __int128 u128;
    const u128 st = (((u128)t.s[1] << 64) | t.s[0]) * (((u128)c.a_hi << 64) | c.a_lo) + (((u128)c.d_hi << 64) | c.d_lo);
    t.s[0] = (uint64_t)st; t.s[1] = (uint64_t)(st >> 64);
    t.buf = (t.buf & (1ull << 32)) | (pcg64_output(t.s[0], t.s[1]) >> 32);
}

// np_integers that REPORTS a re-sample instead of making it (the value is then meaningless)
__device__ __forceinline__ int np_integers_once(NpGen &g, int lo, int hi, bool &resample) {
    const uint32_t rng = (uint32_t)(hi - 1 - lo);
    const uint32_t rng_excl = rng + 1u;
    const uint64_t m = (uint64_t)np_next32(g) * rng_excl;
    const uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) resample |= leftover < (0xFFFFFFFFu - rng) % rng_excl;
#if defined(MGX_BOUNDS_CHECK) && MGX_BOUNDS_CHECK
    // (the checked build, lib/libmgx_chk.so: one draw in 32 CLAIMS a re-sample -- in the product that is one in ~10^9, never seen by a
    // test.  The group then re-runs the call serially from the same state, which must give the same layout: tests/test_checked_build.py)
    resample |= ((leftover >> 10) & 31u) == 0u;
#endif
    return lo + (int)(m >> 32);
}

__device__ __forceinline__ uint32_t lane_read32(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
__device__ __forceinline__ uint64_t lane_read64(uint64_t v, int src_lane) {
    return (uint64_t)lane_read32((uint32_t)v, src_lane) | ((uint64_t)lane_read32((uint32_t)(v >> 32), src_lane) << 32);
}

// numpy Generator.shuffle of a Python list of n <= 21 colours (RandomMixin._rand_perm, multigrid/utils/random.py:75-83: the
// untyped path, `for i in reversed(range(1, n)): j = random_interval(bitgen, i); x[i], x[j] = x[j], x[i]`), the list kept as 3-bit
// fields of ONE 64-bit register (field k = bits [3k, 3k+3)): no per-lane array, no scratch memory
__device__ __forceinline__ uint32_t field3(uint64_t seq, int k) { return (uint32_t)(seq >> (3 * k)) & 7u; }
__device__ __forceinline__ uint64_t np_shuffle3(NpGen &g, uint64_t seq, int n) {
    for (int i = n - 1; i >= 1; --i) {
        const int j = (int)np_interval(g, (uint32_t)i);
        const uint64_t d = (uint64_t)(field3(seq, i) ^ field3(seq, j));
        seq ^= (d << (3 * i)) ^ (d << (3 * j));                       // swap fields i and j (i == j: d ^ d = 0)
    }
    return seq;
}

// The env's grid in memory, for the generators whose objects do not fit a handful of registers (LockedHallway: up to 16 doors +
// 16 keys; Playground: 12 doors + 12 objects): the lane reads back what it (and copy_blank, before the fence) wrote -- past the
// L1, with its own stores retired first.
struct GridIO {
    uint8_t *grid;
    int W;
    __device__ int type_at(int x, int y) const {
        const uint16_t *p = reinterpret_cast<const uint16_t *>(grid + (y * W + x) * kCellBytes);
        return (int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xfu);
    }
    __device__ void set(int x, int y, uint32_t cell) const {
        store_cell(grid + (y * W + x) * kCellBytes, cell);
        __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0): a later type_at of this cell must see it
    }
};

constexpr int kMaxObjects = 4;

// What the owning lane knows about its env while placing: the blank template (global, read-only), the objects placed so
// far (registers) and the agents' positions (LDS, 2 bytes per agent).
template <bool GROUP> struct PlacerGroup {};
template <> struct PlacerGroup<true> { GroupCtx gc; };

template <bool GROUP = false>
struct Placer : PlacerGroup<GROUP> {
    int kind, rs;                         // the blank layout is a closed form of the kind: no memory round trip per test
    int W, H, A;
    int n_obj;
    uint32_t obj_pos[kMaxObjects];        // x | y << 8
    uint32_t obj_cell[kMaxObjects];       // type | color << 8 | state << 16
    uint8_t *apos;                        // LDS: [A][2] (x, y); 0xff = not on the grid (-1)

    __device__ int type_at(int x, int y) const {
        // the blank layout (what copy_blank wrote): RoomGrid walls (roomgrid.py:203-218: rooms of rs x rs sharing walls) for
        // BlockedUnlockPickup; border walls + the goal at (W-2, H-2) for EmptyEnv (empty.py:156-162)
        // ... and for RedBlueDoors the walls of the middle room, wall_rect(W/4, 0, W/2, H) (redbluedoors.py:148-153)
        const bool border = (x == 0) | (y == 0) | (x == W - 1) | (y == H - 1);
        const bool inner = ((kind == MGX_GEN_BLOCKEDUNLOCKPICKUP) & (x == rs - 1))
                         | ((kind == MGX_GEN_REDBLUEDOORS) & ((x == W / 4) | (x == W / 4 + W / 2 - 1)));
        int t = (border | inner) ? (int)T_WALL : (int)T_EMPTY;
        if ((kind == MGX_GEN_EMPTY_FIXED || kind == MGX_GEN_EMPTY_RANDOM) && x == W - 2 && y == H - 2) t = T_GOAL;
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
    __device__ bool agents_reject(int x, int y, bool next_to) const {
        bool bad = false;
        for (int a = 0; a < A; ++a) {
            const int ax = apos[2 * a] == 0xff ? -1 : apos[2 * a], ay = apos[2 * a + 1] == 0xff ? -1 : apos[2 * a + 1];
            const int dx = x - ax, dy = y - ay;
            bad |= (dx == 0) & (dy == 0);
            bad |= next_to & (dx * dx + dy * dy <= 1);
        }
        return bad;
    }
    // GROUP: the group's lanes evaluate tries 0 .. kGroupLanes - 1 of the call at once (see GroupCtx); `type_of(x, y)` is the grid
    // test of the serial forms below.  Returns false when the call must be made serially (a re-sample before the first fit).
    template <class F>
    __device__ bool place_group(NpGen &g, int tx, int xhi, int ty, int yhi, bool next_to, F type_of, uint32_t &pos) const {
        if constexpr (GROUP) {
            const GroupCtx &c = this->gc;
            for (;;) {
                NpGen t = g;
                group_skip(t, c);
                bool resample = false;
                const int x = np_integers_once(t, tx, xhi, resample), y = np_integers_once(t, ty, yhi, resample);
                const bool fits = !resample && type_of(x, y) == T_EMPTY && !agents_reject(x, y, next_to);
                const uint32_t all = (1u << kGroupLanes) - 1u;
                const uint32_t fit8 = (uint32_t)(__builtin_amdgcn_ballot_w64(fits) >> c.base) & all;
                const uint32_t rs8 = (uint32_t)(__builtin_amdgcn_ballot_w64(resample) >> c.base) & all;
                const int f = fit8 ? __builtin_ctz(fit8) : kGroupLanes - 1;          // the try the group goes on from
                if (rs8 & ((2u << f) - 1u)) return false;                               // (a re-sample shifts every later try)
                const int src = c.base + f;
                g.s[0] = lane_read64(t.s[0], src); g.s[1] = lane_read64(t.s[1], src); g.buf = lane_read64(t.buf, src);
                pos = lane_read32((uint32_t)x | ((uint32_t)y << 8), src);
                if (fit8) return true;
            }
        }
        return false;
    }
    // the same against the grid in memory (GridIO) instead of the closed form + register objects
    __device__ uint32_t place_io(NpGen &g, const GridIO &io, int tx, int ty, int sw, int sh, bool next_to) const {
        tx = max(tx, 0); ty = max(ty, 0);
        const int xhi = min(tx + sw, W), yhi = min(ty + sh, H);
        if constexpr (GROUP) {
            uint32_t pos;
            if (xhi - tx >= 2 && yhi - ty >= 2 && place_group(g, tx, xhi, ty, yhi, next_to, [&](int x, int y) { return io.type_at(x, y); }, pos))
                return pos;
        }
        for (;;) {
            const int x = np_integers(g, tx, xhi), y = np_integers(g, ty, yhi);
            if (io.type_at(x, y) != T_EMPTY) continue;
            bool bad = false;
            for (int a = 0; a < A; ++a) {
                const int ax = apos[2 * a] == 0xff ? -1 : apos[2 * a], ay = apos[2 * a + 1] == 0xff ? -1 : apos[2 * a + 1];
                const int dx = x - ax, dy = y - ay;
                bad |= (dx == 0) & (dy == 0);
                bad |= next_to & (dx * dx + dy * dy <= 1);
            }
            if (bad) continue;
            return (uint32_t)x | ((uint32_t)y << 8);
        }
    }
    // base.py:604-669 place_obj: returns the position as x | y << 8
    __device__ uint32_t place(NpGen &g, int tx, int ty, int sw, int sh, bool next_to) const {
        tx = max(tx, 0); ty = max(ty, 0);
        const int xhi = min(tx + sw, W), yhi = min(ty + sh, H);
        if constexpr (GROUP) {                     // (a range of one cell draws nothing: such a try is not "two draws" -- serial)
            uint32_t pos;
            if (xhi - tx >= 2 && yhi - ty >= 2 && place_group(g, tx, xhi, ty, yhi, next_to, [&](int x, int y) { return type_at(x, y); }, pos))
                return pos;
        }
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


// Agent.reset (agent.py:120-133) + _gen_grid for ONE env, run by one lane.  `grid` / `rows`: the env's slices in HBM (the
// blank layout is already in `grid`); `apos`: 2 * A bytes of LDS scratch.  Returns the env's new hook state.
// `door_row` > 0: the value of the generator's ONE draw from env.np_random is given (a candidate of MgxGenStage.candidates: `npr` is
// not touched); 0: it is drawn.
// GROUP: run by the kGroupLanes lanes of a group in step (`gc`: the lane's GroupCtx) -- every lane computes and stores the same
// values; only the place_obj tries differ per lane (Placer::place_group).
template <bool GROUP = false>
__device__ __forceinline__ uint4 generate_episode(const MgxLayoutGen &gen, int W, int H, int A, NpGen &lay, NpGen &npr,
                                                  uint8_t *apos, uint8_t *grid, uint64_t *rows, int door_row = 0,
                                                  const GroupCtx *gc = nullptr) {
    Placer<GROUP> P;
    if constexpr (GROUP) P.gc = *gc;
    P.kind = gen.kind; P.rs = gen.room_size; P.W = W; P.H = H; P.A = A; P.n_obj = 0;
    P.apos = apos;
    for (int k = 0; k < kMaxObjects; ++k) { P.obj_pos[k] = 0xffffffffu; P.obj_cell[k] = 0; }
    auto write_row = [&](int i, int x, int y, int d) {                              // colours cycle (constants.py:77-82)
        rows[i] = (uint64_t)(i % 6) | ((uint64_t)d << 8) | ((uint64_t)x << 16) | ((uint64_t)y << 24) | ((uint64_t)CELL_EMPTY << 40);
    };
    uint4 aux = {0, 0, 0, 0};
    if (gen.kind == MGX_GEN_EMPTY_FIXED) {                                          // empty.py:164-167
        for (int i = 0; i < A; ++i) write_row(i, gen.start_x, gen.start_y, gen.start_dir);
    } else if (gen.kind == MGX_GEN_EMPTY_RANDOM) {                                  // empty.py:168-169: place_agent(agent)
        for (int i = 0; i < A; ++i) { P.apos[2 * i] = 0xff; P.apos[2 * i + 1] = 0xff; }
        for (int i = 0; i < A; ++i) {
            const uint32_t p = P.place(lay, 0, 0, W, H, false);
            P.apos[2 * i] = (uint8_t)p; P.apos[2 * i + 1] = (uint8_t)(p >> 8);
            write_row(i, p & 0xff, p >> 8, np_integers(lay, 0, 4));
        }
    } else if (gen.kind == MGX_GEN_REDBLUEDOORS) {                                   // redbluedoors.py:142-168
        const int rx0 = W / 4, rw = W / 2;                                           // room_top = (width // 4, 0), size (width // 2, height)
        for (int i = 0; i < A; ++i) { P.apos[2 * i] = 0xff; P.apos[2 * i + 1] = 0xff; }   // Agent.reset: pos = (-1, -1)
        for (int i = 0; i < A; ++i) {                                                // place_agent(agent, top=room_top, size=room_size)
            const uint32_t p = P.place(lay, rx0, 0, rw, H, false);
            P.apos[2 * i] = (uint8_t)p; P.apos[2 * i + 1] = (uint8_t)(p >> 8);
            write_row(i, p & 0xff, p >> 8, np_integers(lay, 0, 4));
        }
        const int ry = np_integers(lay, 1, H - 1);                                   // red door, left wall of the room
        const int by = np_integers(lay, 1, H - 1);                                   // blue door, right wall
        const int bx = rx0 + rw - 1;
        store_cell(grid + (ry * W + rx0) * kCellBytes, (uint32_t)T_DOOR | (0u << 8) | ((uint32_t)S_CLOSED << 16));     // Color.red
        store_cell(grid + (by * W + bx) * kCellBytes, (uint32_t)T_DOOR | (2u << 8) | ((uint32_t)S_CLOSED << 16));      // Color.blue
        aux.x = (uint32_t)bx | ((uint32_t)by << 8) | ((uint32_t)rx0 << 16) | ((uint32_t)ry << 24);   // include/mgx.h: blue, red; [4] = 0
    } else if (gen.kind == MGX_GEN_LOCKEDHALLWAY) {                                  // locked_hallway.py:152-201
        const GridIO io{grid, W};
        const int rs = gen.room_size, nrows = (H - 1) / (rs - 1), n = 2 * nrows;
        for (int i = 0; i < A; ++i) {                                                // roomgrid.py:232-236: all agents in the middle
            P.apos[2 * i] = (uint8_t)((rs - 1) + rs / 2); P.apos[2 * i + 1] = (uint8_t)((nrows / 2) * (rs - 1) + rs / 2);
        }
        const int Lc = 6 * ((n + 5) / 6);
        uint64_t seq = 0;
        for (int k = 0; k < Lc; ++k) seq |= (uint64_t)(k % 6) << (3 * k);            // list(Color) * ceil(n / 6)
        seq = np_shuffle3(lay, seq, Lc);                                             // _rand_perm(...)[:num_rooms]
        uint64_t doors = np_shuffle3(lay, n >= 21 ? seq : (seq & ((1ull << (3 * n)) - 1ull)), n);   // door_colors = _rand_perm(color_sequence)
        uint32_t room_of_color = 0xffffffu, distinct = 0;                            // 4-bit field per colour: row * 2 + side
        int top = n;
        for (int row = 0; row < nrows; ++row)
            for (int side = 0; side < 2; ++side) {                                   // (LEFT, right), (RIGHT, left)
                const uint32_t color = field3(doors, --top);                         // door_colors.pop()
                room_of_color = (room_of_color & ~(0xfu << (4 * color))) | ((uint32_t)(row * 2 + side) << (4 * color));
                distinct |= 1u << color;
                io.set(side ? 2 * (rs - 1) : rs - 1, row * (rs - 1) + (rs - 1) / 2,
                       (uint32_t)T_DOOR | (color << 8) | ((uint32_t)S_LOCKED << 16));
            }
        const int nhk = np_integers(lay, 1, gen.max_hallway_keys + 1);
        for (int t = 0; t < nhk && t < n; ++t) {                                     // keys in the hallway (column 1, whole height)
            const uint32_t p = P.place_io(lay, io, rs - 1, 0, rs, H, false);
            io.set(p & 0xff, p >> 8, (uint32_t)T_KEY | (field3(seq, t) << 8));
        }
        int ki = nhk;
        while (ki < n) {                                                             // keys in the rooms
            const uint32_t r = (room_of_color >> (4 * field3(seq, ki - 1))) & 0xfu;
            const int row = (int)(r >> 1), side = (int)(r & 1u);
            const int nrk = np_integers(lay, 1, gen.max_keys_per_room + 1);
            const int stop = min(ki + nrk, n);                                       // color_sequence[ki : ki + nrk]
            for (int t = ki; t < stop; ++t) {
                const uint32_t p = P.place_io(lay, io, side ? 2 * (rs - 1) : 0, row * (rs - 1), rs, rs, false);
                io.set(p & 0xff, p >> 8, (uint32_t)T_KEY | (field3(seq, t) << 8));
                ++ki;
            }
        }
        for (int i = 0; i < A; ++i) {                                                // MultiGridEnv.place_agent in the hallway
            P.apos[2 * i] = 0xff; P.apos[2 * i + 1] = 0xff;
            const uint32_t p = P.place_io(lay, io, rs - 1, 0, rs, H, false);
            P.apos[2 * i] = (uint8_t)p; P.apos[2 * i + 1] = (uint8_t)(p >> 8);
            write_row(i, p & 0xff, p >> 8, np_integers(lay, 0, 4));
        }
        if (n <= 6) {                                                                // include/mgx.h: explicit door positions, sorted by (x, y)
            uint64_t lo = (uint64_t)n, hi = 0;                                       // bytes 0..7 / 8..15 (no indexed array: no scratch)
            for (int k = 0; k < n; ++k) {
                const int side = k / nrows, row = k % nrows, b0 = 2 + 2 * k;
                const uint64_t xy = (uint64_t)(side ? 2 * (rs - 1) : rs - 1) | ((uint64_t)(row * (rs - 1) + (rs - 1) / 2) << 8);
                if (b0 < 8) lo |= xy << (8 * b0); else hi |= xy << (8 * (b0 - 8));   // (b0 is even: a pair never straddles)
            }
            aux.x = (uint32_t)lo; aux.y = (uint32_t)(lo >> 32); aux.z = (uint32_t)hi; aux.w = (uint32_t)(hi >> 32);
        } else {                                                                     // geometric format: room_size, len(self.rooms)
            aux.x = (uint32_t)(0x80 | n) | ((uint32_t)rs << 24);
            aux.y = (uint32_t)__builtin_popcount(distinct);
        }
    } else if (gen.kind == MGX_GEN_PLAYGROUND) {                                     // playground.py:122-137 over roomgrid.py:203-452
        const GridIO io{grid, W};
        const int rs = gen.room_size, ncols = (W - 1) / (rs - 1), nrows = (H - 1) / (rs - 1), R = nrows * ncols;
        for (int i = 0; i < A; ++i) {
            P.apos[2 * i] = (uint8_t)((ncols / 2) * (rs - 1) + rs / 2); P.apos[2 * i + 1] = (uint8_t)((nrows / 2) * (rs - 1) + rs / 2);
        }
        uint64_t dbits = 0;                                   // bit 16 d + r: room r (<= 16 rooms) has a door in direction d
        const uint32_t all = (1u << R) - 1u;
        for (int itr = 0; itr < 5000; ++itr) {                                       // connect_all (roomgrid.py:406-452)
            uint32_t seen = 1u;                                                      // bfs from room (0, 0): rooms reachable through doors
            for (int pass = 0; pass < R; ++pass) {
                const uint32_t nx = seen | ((seen & (uint32_t)(dbits & 0xffff)) << 1) | ((seen & (uint32_t)((dbits >> 16) & 0xffff)) << ncols)
                                  | ((seen & (uint32_t)((dbits >> 32) & 0xffff)) >> 1) | ((seen & (uint32_t)((dbits >> 48) & 0xffff)) >> ncols);
                if (nx == seen) break;
                seen = nx;
            }
            if ((seen & all) == all) break;
            const int col = np_integers(lay, 0, ncols), row = np_integers(lay, 0, nrows), d = np_integers(lay, 0, 4);
            const int ncol = col + dir_dx(d), nrow = row + dir_dy(d), r = row * ncols + col;
            if (ncol < 0 || ncol >= ncols || nrow < 0 || nrow >= nrows || ((dbits >> (16 * d + r)) & 1ull)) continue;
            const uint32_t color = (uint32_t)np_integers(lay, 0, 6);                 // _rand_elem(door_colors)
            const int left = col * (rs - 1), topy = row * (rs - 1), right = left + rs - 1, bottom = topy + rs - 1;
            int dx, dy;                                                              // Room.set_door_pos(dir, random=env.np_random)
            if (d == 0) { dx = right; dy = np_integers(npr, topy + 1, bottom); }
            else if (d == 1) { dx = np_integers(npr, left + 1, right); dy = bottom; }
            else if (d == 2) { dx = left; dy = np_integers(npr, topy + 1, bottom); }
            else { dx = np_integers(npr, left + 1, right); dy = topy; }
            io.set(dx, dy, (uint32_t)T_DOOR | (color << 8) | ((uint32_t)S_CLOSED << 16));
            dbits |= 1ull << (16 * d + r);
            dbits |= 1ull << (16 * ((d + 2) & 3) + nrow * ncols + ncol);
        }
        for (int k = 0; k < 12; ++k) {                                               // 12 random objects, each in a random room
            const int col = np_integers(lay, 0, ncols), row = np_integers(lay, 0, nrows);
            const uint32_t kind = (uint32_t)T_KEY + (uint32_t)np_integers(lay, 0, 3);      // ['key', 'ball', 'box']
            const uint32_t color = (uint32_t)np_integers(lay, 0, 6);
            const uint32_t p = P.place_io(lay, io, col * (rs - 1), row * (rs - 1), rs, rs, true);   // place_in_room: reject_next_to
            io.set(p & 0xff, p >> 8, kind | (color << 8));
        }
        for (int i = 0; i < A; ++i) {                                                // RoomGrid.place_agent: a random room
            const int col = np_integers(lay, 0, ncols), row = np_integers(lay, 0, nrows);
            for (;;) {
                P.apos[2 * i] = 0xff; P.apos[2 * i + 1] = 0xff;
                const uint32_t p = P.place_io(lay, io, col * (rs - 1), row * (rs - 1), rs, rs, false);
                const int x = (int)(p & 0xff), y = (int)(p >> 8);
                P.apos[2 * i] = (uint8_t)x; P.apos[2 * i + 1] = (uint8_t)y;
                const int d = np_integers(lay, 0, 4);
                const int t = io.type_at(x + dir_dx(d), y + dir_dy(d));
                if (t == T_EMPTY || t == T_WALL) { write_row(i, x, y, d); break; }
            }
        }
    } else {                                                                         // blockedunlockpickup.py:142-164
        const int rs = gen.room_size;
        for (int i = 0; i < A; ++i) { P.apos[2 * i] = (uint8_t)((rs - 1) + rs / 2); P.apos[2 * i + 1] = (uint8_t)(rs / 2); }   // roomgrid.py:232-236
        const uint32_t box_color = (uint32_t)np_integers(lay, 0, 6);
        uint32_t p = P.place(lay, rs - 1, 0, rs, rs, true);                          // box in the right room
        P.put(p & 0xff, p >> 8, (uint32_t)T_BOX | (box_color << 8));
        const uint32_t door_color = (uint32_t)np_integers(lay, 0, 6);
        const int door_x = rs - 1, door_y = door_row > 0 ? door_row : np_integers(npr, 1, rs - 1);   // roomgrid.py:104-106: env.np_random
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
            store_cell(grid + (((P.obj_pos[k] >> 8) & 0xff) * W + (P.obj_pos[k] & 0xff)) * kCellBytes, P.obj_cell[k]);
        }
        aux.x = (uint32_t)T_BOX | (box_color << 8);                                  // the target box `self.obj` (include/mgx.h)
    }
    return aux;
}

// the blank layout into the grid of every env of `mask` (bit = env e0 + bit), all 64 lanes copying
__device__ __forceinline__ void copy_blank(const MgxLayoutGen &gen, uint8_t *grid_base, int64_t e0, int HWB, uint64_t mask, int lane) {
    const uint8_t *blank = reinterpret_cast<const uint8_t *>(gen.blank);
    const bool dw = ((HWB | (int)(uintptr_t)blank | (int)(uintptr_t)grid_base) & 3) == 0;
    for (uint64_t m = mask; m != 0; m &= m - 1) {
        uint8_t *dst = grid_base + (e0 + __builtin_ctzll(m)) * HWB;
        if (dw) {
            for (int i = lane; i < HWB / 4; i += 64)
                reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(blank)[i];
        } else {
            for (int i = lane; i < HWB; i += 64) dst[i] = blank[i];
        }
    }
}

}  // namespace mgx_gen
