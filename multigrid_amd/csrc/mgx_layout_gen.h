// mgx_layout_gen.h -- device-side episode generation shared by mgx_layout_gen.hip (mgx_reset_generate: its own launch) and
// the fused step kernel (mgx_step_generate: the finished envs of a step are regenerated in the tail of the same launch).
// What is restated (numpy's Generator.integers, place_obj, the two _gen_grid) and how it is pinned: mgx_layout_gen.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx_rules.h"

namespace mgx_gen {

using namespace mgx;

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


// Agent.reset (agent.py:120-133) + _gen_grid for ONE env, run by one lane.  `grid` / `rows`: the env's slices in HBM (the
// blank layout is already in `grid`); `apos`: 2 * A bytes of LDS scratch.  Returns the env's new hook state.
__device__ __forceinline__ uint4 generate_episode(const MgxLayoutGen &gen, int W, int H, int A, NpGen &lay, NpGen &npr,
                                                  uint8_t *apos, uint8_t *grid, uint64_t *rows) {
    Placer P;
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
    } else {                                                                         // blockedunlockpickup.py:142-164
        const int rs = gen.room_size;
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
