// mgx_rules.h -- per-environment integer rules shared by the HIP kernels (one lane = one env / one view).
//
// Everything here is straight-line integer code on plain pointers, so the same functions run inside the
// gfx950 kernels (pointers into LDS) and, compiled by g++ into tests/hostshim, against the CPU oracle in
// the `-m "not gpu"` suite.  There is no CPU product path: libmgx.so only ever launches the kernels.
//
// Semantics follow ini/multigrid (paths relative to the reference root); see SURVEY.md App. A.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#endif

#include "../../include/mgx.h"

#if defined(__HIPCC__)
#define MGX_HD __host__ __device__ __forceinline__
#else
#define MGX_HD inline
#endif

namespace mgx {

// multigrid/core/constants.py:34-48, 91-97 ; multigrid/core/actions.py:5-15
enum : int { T_UNSEEN = 0, T_EMPTY = 1, T_WALL = 2, T_FLOOR = 3, T_DOOR = 4, T_KEY = 5, T_BALL = 6, T_BOX = 7,
             T_GOAL = 8, T_LAVA = 9, T_AGENT = 10 };
enum : int { S_OPEN = 0, S_CLOSED = 1, S_LOCKED = 2 };
enum : int { ACT_LEFT = 0, ACT_RIGHT = 1, ACT_FORWARD = 2, ACT_PICKUP = 3, ACT_DROP = 4, ACT_TOGGLE = 5, ACT_DONE = 6 };
// packed agent row (include/mgx.h)
enum : int { AG_COLOR = 0, AG_DIR = 1, AG_X = 2, AG_Y = 3, AG_TERM = 4, AG_CARRY = 5 };

// A cell as the rules see it ("logical" cell, also the obs / carried-object byte order), little-endian in 24 bits:
// type | color << 8 | state << 16.
// (round 5) A BOX may hold an object (multigrid/core/world_object.py:574-605: Box.contains; Box.toggle replaces the box by it).
// Neither Grid.state nor an observation shows the content -- it is an attribute of the Python object -- so it travels in the six
// bits of the state BYTE that the 2-bit state leaves free: bits [23:18] of the logical cell = content kind [20:18] | content
// colour [23:21].  Kinds: 0 nothing, 1 key, 2 ball, 3 goal, 4 floor, 5 lava, 6 wall, 7 door (closed, unlocked: what Door(color)
// constructs); a box in a box and doors in other states are refused where grids enter (world.Box).  Everything that SHOWS a cell
// (observations, full_obs, Grid.state) takes the low 18 bits: kCellShown.
constexpr uint32_t kCellShown = 0x0003ffffu;
constexpr uint32_t CELL_EMPTY = 1u;                       // (1,0,0)  world_object.py:131-137
constexpr uint32_t CELL_WALL = 2u | (5u << 8);            // (2,5,0)  obs.py:14
constexpr uint32_t CELL_UNSEEN = 0u;                      // (0,0,0)  obs.py:15

// A cell as the GRID stores it (HBM and the LDS tile; include/mgx.h "packed cell"), 16 bits:
//   [3:0] type   [10:8] color   [13:12] state   [15] opaque = !see_behind(cell)
//   [7:4], [11], [14]: a box's content (kind [6:4], colour {[7], [11], [14]}), zero on every other cell
// Two bytes instead of three cut the grid stream by a third and make every cell an aligned 16-bit LDS access (no dword pair
// + v_alignbyte); the opaque bit is the sign of the sign-extended load, so the see-behind ballot of the gather is ONE compare.
constexpr int kCellBytes = MGX_CELL_BYTES;
constexpr uint32_t CELL16_WALL = 0x8502u;                 // cell_pack(CELL_WALL)

// multigrid/core/constants.py:21-30 DIR_TO_VEC, branch-free: 0:(1,0) 1:(0,1) 2:(-1,0) 3:(0,-1)
MGX_HD int dir_dx(int d) { return (d == 0) - (d == 2); }
MGX_HD int dir_dy(int d) { return (d == 1) - (d == 3); }

// multigrid/utils/obs.py:46-63 see_behind, on a logical cell
MGX_HD uint32_t cell_state(uint32_t c) { return (c >> 16) & 3u; }
MGX_HD bool see_behind(uint32_t c) {
    const uint32_t t = c & 0xffu;
    return !(t == T_WALL || (t == T_DOOR && cell_state(c) != S_OPEN));
}
// What a toggled box leaves on its cell (world_object.py:599-605: env.grid.set(*pos, self.contains)): the content as a cell,
// EMPTY for a box that holds nothing.  Kind -> type by a nibble table; only the door kind has a state (closed).
MGX_HD uint32_t box_content_cell(uint32_t box) {
    const uint32_t kind = (box >> 18) & 7u, color = (box >> 21) & 7u;
    const uint32_t type = (0x42938651u >> (4u * kind)) & 0xfu;            // empty, key, ball, goal, floor, lava, wall, door
    return type | (color << 8) | ((kind == 7u ? (uint32_t)S_CLOSED : 0u) << 16);
}

// ---------------------------------------------------------------------------------------------------
// numpy Generator(PCG64).random(): 128-bit LCG step, XSL-RR output of the NEW state, top 53 bits.
// rng = [state_lo, state_hi, inc_lo, inc_hi].  Returns (u64 >> 11); doubles compare like these integers.
// ---------------------------------------------------------------------------------------------------
MGX_HD uint64_t pcg64_next53(uint64_t &s_lo, uint64_t &s_hi, uint64_t inc_lo, uint64_t inc_hi) {
    typedef unsigned __int128 u128;
    const u128 mult = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    u128 state = (((u128)s_hi << 64) | s_lo) * mult + (((u128)inc_hi << 64) | inc_lo);
    s_lo = (uint64_t)state;
    s_hi = (uint64_t)(state >> 64);
    const uint64_t x = s_hi ^ s_lo;
    const unsigned rot = (unsigned)(s_hi >> 58);
    return ((x >> rot) | (x << ((64u - rot) & 63u))) >> 11;
}

// ---------------------------------------------------------------------------------------------------
// Visibility: closed form of get_vis_mask (multigrid/utils/obs.py:235-273), one view per caller.
// Row j (depth) is a v-bit mask, bit i = lateral index.  See SURVEY.md App. A.5.
// ---------------------------------------------------------------------------------------------------
MGX_HD uint32_t brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

// bits [pos, pos+V) of a multi-word little-endian bit string
template <int V, int NW>
MGX_HD uint32_t get_bits(const uint64_t (&w)[NW], int pos) {
    const int k = pos >> 6, o = pos & 63;
    uint64_t r = w[k] >> o;
    if (o + V > 64 && k + 1 < NW) r |= w[k + 1] << (64 - o);
    return (uint32_t)r & ((1u << V) - 1u);
}

template <int V, int NW>
MGX_HD void or_bits(uint64_t (&w)[NW], int pos, uint32_t bits) {
    const int k = pos >> 6, o = pos & 63;
    w[k] |= (uint64_t)bits << o;
    if (o + V > 64 && k + 1 < NW) w[k + 1] |= (uint64_t)bits >> (64 - o);
}

// sb: see-behind bits of the (unmasked) view, bit j*V + i for image[i][j].  Returns visibility in the same
// bit order.  The agent sits at (i, j) = (V/2, V-1) (obs.py:252).
//
// Both sweep directions of a row are done by ONE carry: a row is kept "mirrored" in a 32-bit word, the row itself in
// bits [0,V) and its bit-reversal in bits [32-V,32) (x | brev32(x)).  Adding the seed bits to the transparency bits
// lets the carry run through each transparent run and stop on the first blocker -- upwards in the low copy, which is
// downwards for the mirrored copy.  brev32 of the result maps each half onto the other, so `u | brev32(u)` completes
// both copies.  (V <= 15, so the two copies and the carry bit between them never touch.)
template <int V, int NW>
MGX_HD void vis_mask(const uint64_t (&sb)[NW], uint64_t (&vis)[NW]) {
    constexpr uint32_t F = (1u << V) - 1u;
    for (int k = 0; k < NW; ++k) vis[k] = 0;
    uint32_t init = 1u << (V / 2);
    init |= brev32(init);
    // The bits between the two copies (the "gap", [V, 32-V)) may fill with junk -- the carry out of the low copy stops at
    // bit V because s is zero there, the 3-wide spread leaks one bit into each end of the gap -- and nothing cleans it:
    // brev32 maps the gap onto itself, `& s` and `& F` drop it wherever a value is used.
#pragma unroll
    for (int j = V - 1; j >= 0; --j) {
        const uint32_t s0 = get_bits<V, NW>(sb, j * V);
        const uint32_t s = s0 | brev32(s0);                          // mirrored transparency of row j
        uint32_t u = ((s + (init & s)) ^ s) | init;                  // flood from the seeds, both directions
        u |= brev32(u);                                               // each copy gets the other direction's result
        or_bits<V, NW>(vis, j * V, u & F);
        const uint32_t p = u & s;                                     // visible AND transparent
        init = p | (p << 1) | (p >> 1);                               // lights (i-1, i, i+1) of row j-1 (both copies)
    }
}

// ---------------------------------------------------------------------------------------------------
// Packed agent row as one 64-bit word (little-endian view of the 8 bytes in include/mgx.h):
//   [0:8) color  [8:16) dir  [16:24) x  [24:32) y  [32:40) terminated  [40:64) carried cell (type|color<<8|state<<16)
// ---------------------------------------------------------------------------------------------------
MGX_HD int row_dir(uint64_t r) { return (int)((r >> 8) & 0xff); }
MGX_HD int row_x(uint64_t r) { return (int)((r >> 16) & 0xff); }
MGX_HD int row_y(uint64_t r) { return (int)((r >> 24) & 0xff); }
MGX_HD bool row_term(uint64_t r) { return ((r >> 32) & 0xff) != 0; }
MGX_HD uint32_t row_carry(uint64_t r) { return (uint32_t)(r >> 40); }
MGX_HD uint64_t row_set_dir(uint64_t r, int d) { return (r & ~0xff00ull) | ((uint64_t)(d & 0xff) << 8); }
MGX_HD uint64_t row_set_pos(uint64_t r, int x, int y) {
    return (r & ~0xffff0000ull) | ((uint64_t)(x & 0xff) << 16) | ((uint64_t)(y & 0xff) << 24);
}
MGX_HD uint64_t row_set_carry(uint64_t r, uint32_t c) { return (r & 0xffffffffffull) | ((uint64_t)c << 40); }

// "does any lane of the wavefront ...": box contents are rare, so the instructions that carry them sit behind a wave-uniform
// branch (one ballot) instead of on every agent's chain; on the host (tests/hostshim) it is the condition itself
#if defined(__HIP_DEVICE_COMPILE__)
#define MGX_ANY_LANE(cond) (__builtin_amdgcn_ballot_w64(cond) != 0)
#else
#define MGX_ANY_LANE(cond) (cond)
#endif

// logical <-> packed.  Values outside the packed ranges (type > 15, color > 7, state > 3) do not occur in the reference
// (types 0-10, colors 0-5, states 0-2 / directions 0-3) and are refused where grids enter (mgx_pack_grid reports them).
MGX_HD uint32_t cell_pack(uint32_t c) {
    uint32_t p = (c & 0x070fu) | ((c >> 4) & 0x3000u) | (see_behind(c) ? 0u : 0x8000u);
    if (MGX_ANY_LANE((c >> 18) != 0))                                     // content: [21:18] -> [7:4], [22] -> [11], [23] -> [14]
        p |= ((c >> 14) & 0x00f0u) | ((c >> 11) & 0x0800u) | ((c >> 9) & 0x4000u);
    return p;
}
// what a cell SHOWS (observations, full_obs): (type, color, state)
MGX_HD uint32_t cell_unpack(uint32_t p) { return (p & 0x070fu) | ((p & 0x3000u) << 4); }
// ... and with a box's content, for the rules
// (-DMGX_BOX_CONTENTS=0 / -DMGX_RULES_KIND=0: A/B builds without the round-5 additions to the per-agent phase, tools/variant_bench.sh)
#ifndef MGX_BOX_CONTENTS
#define MGX_BOX_CONTENTS 1
#endif
#ifndef MGX_RULES_KIND
#define MGX_RULES_KIND 1
#endif
MGX_HD uint32_t cell_unpack_full(uint32_t p) {
#if MGX_BOX_CONTENTS
    return cell_unpack(p) | ((p & 0x00f0u) << 14) | ((p & 0x0800u) << 11) | ((p & 0x4000u) << 9);
#else
    return cell_unpack(p);
#endif
}
// the agent overlay cell (10, color, dir) of a packed agent row (obs.py:163-173); never opaque
MGX_HD uint32_t agent_cell16(uint64_t row) {
    return (uint32_t)T_AGENT | (((uint32_t)row & 0x7u) << 8) | ((((uint32_t)row >> 8) & 0x3u) << 12);
}

// COMPACT cells (include/mgx.h: MgxCell8, MgxSpec.cell_bytes = 1): one byte, type and state coded jointly --
//   [3:0] tcode: 0..10 = type with state 0 | 11, 12 = door closed, locked | 13..15 = agent overlay facing 1..3   [6:4] color   [7] opaque
// Only doors and the agent overlay have a state in the reference, so nothing is lost; mgx_pack_grid8_env counts anything else.
constexpr uint32_t CELL8_WALL = 0xD2u;                    // cell8_pack(CELL_WALL)
MGX_HD uint32_t cell8_pack(uint32_t c) {
    const uint32_t t = c & 0xffu, st = (c >> 16) & 3u;
    const uint32_t tc = (st == 0u) ? t : ((t == (uint32_t)T_DOOR) ? 10u + st : 12u + st);
    return (tc & 15u) | (((c >> 8) & 7u) << 4) | (see_behind(c) ? 0u : 0x80u);
}
MGX_HD uint32_t cell8_unpack(uint32_t p) {
    const uint32_t tc = p & 15u;
    const bool hi = tc > 10u, door = tc < 13u;
    const uint32_t type = hi ? (door ? (uint32_t)T_DOOR : (uint32_t)T_AGENT) : tc;
    const uint32_t st = hi ? (door ? tc - 10u : tc - 12u) : 0u;
    return type | (((p >> 4) & 7u) << 8) | (st << 16);
}
MGX_HD uint32_t agent_cell8(uint64_t row) {
    const uint32_t d = ((uint32_t)row >> 8) & 3u;
    return (d == 0u ? (uint32_t)T_AGENT : 12u + d) | (((uint32_t)row & 7u) << 4);
}

// grid cells (tile / HBM): 2 bytes, aligned
MGX_HD uint32_t load_cell16(const uint8_t *p) { return *reinterpret_cast<const uint16_t *>(p); }
MGX_HD void store_cell16(uint8_t *p, uint32_t c16) { *reinterpret_cast<uint16_t *>(p) = (uint16_t)c16; }
MGX_HD uint32_t load_cell(const uint8_t *p) { return cell_unpack_full(load_cell16(p)); }
MGX_HD void store_cell(uint8_t *p, uint32_t c) { store_cell16(p, cell_pack(c)); }
// ... in the engine's cell format: `cb` = bytes per cell, 2 (MgxCell) or 1 (MgxCell8).  In the kernels it is a compile-time
// constant of the instantiation, so the selects fold away.
MGX_HD uint32_t load_cell_raw(int cb, const uint8_t *p) { return cb == 1 ? (uint32_t)*p : load_cell16(p); }
MGX_HD void store_cell_raw(int cb, uint8_t *p, uint32_t raw) { if (cb == 1) *p = (uint8_t)raw; else store_cell16(p, raw); }
MGX_HD uint32_t load_cell(int cb, const uint8_t *p) { return cb == 1 ? cell8_unpack(*p) : cell_unpack_full(load_cell16(p)); }
// (cb == 3: the reference's byte triples, as mgx_full_obs reads a byte grid)
MGX_HD uint32_t load_cell_shown(int cb, const uint8_t *p) {
    return cb == 1 ? cell8_unpack(*p)
                   : (cb == 3 ? (((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)) & 0x0003ffffu) : cell_unpack(load_cell16(p)));
}
MGX_HD void store_cell(int cb, uint8_t *p, uint32_t c) { if (cb == 1) *p = (uint8_t)cell8_pack(c); else store_cell16(p, cell_pack(c)); }
MGX_HD uint32_t agent_cell_raw(int cb, uint64_t row) { return cb == 1 ? agent_cell8(row) : agent_cell16(row); }
// observation cells: the reference's 3 bytes (type, color, state), any alignment
MGX_HD uint32_t load_obs_cell(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
MGX_HD void store_obs_cell(uint8_t *p, uint32_t c) { p[0] = (uint8_t)c; p[1] = (uint8_t)(c >> 8); p[2] = (uint8_t)(c >> 16); }

// base.py:598-602 `1 - 0.9 * (step_count / max_steps)` in Python float arithmetic: three correctly rounded
// IEEE-754 binary64 operations, never contracted into an fma (the library is built with -ffp-contract=off).
MGX_HD double reward_value(int32_t step_count, int32_t max_steps) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double q = __ddiv_rn((double)step_count, (double)max_steps);
    return __dsub_rn(1.0, __dmul_rn(0.9, q));
#else
    volatile double q = (double)step_count / (double)max_steps;
    volatile double p = 0.9 * q;
    return 1.0 - p;
#endif
}

// ---------------------------------------------------------------------------------------------------
// PCG64 jump-ahead: state after k steps = state * M^k + inc * (M^(k-1) + ... + M + 1)  (mod 2^128), so the A
// draws of one step (multigrid/base.py:399) are computed by A lanes in parallel instead of a serial chain.
// kJump[k] = { M^k lo, M^k hi, C_k lo, C_k hi }.
// ---------------------------------------------------------------------------------------------------
struct JumpTable {
    uint64_t w[MGX_MAX_AGENTS + 1][4];
    constexpr JumpTable() : w{} {
        typedef unsigned __int128 u128;
        const u128 mult = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
        u128 m = 1, c = 0;
        for (int k = 0; k <= MGX_MAX_AGENTS; ++k) {
            w[k][0] = (uint64_t)m; w[k][1] = (uint64_t)(m >> 64);
            w[k][2] = (uint64_t)c; w[k][3] = (uint64_t)(c >> 64);
            c = c * mult + 1;      // C_{k+1} = C_k * M + 1
            m = m * mult;
        }
    }
};

// k-th next draw (k >= 1) from the state in rng[0..3]; also returns the advanced state.
MGX_HD uint64_t pcg64_draw_at(const uint64_t *rng, const uint64_t *jump_k, uint64_t &s_lo, uint64_t &s_hi) {
    typedef unsigned __int128 u128;
    const u128 s = ((u128)rng[1] << 64) | rng[0], inc = ((u128)rng[3] << 64) | rng[2];
    const u128 mk = ((u128)jump_k[1] << 64) | jump_k[0], ck = ((u128)jump_k[3] << 64) | jump_k[2];
    const u128 st = s * mk + inc * ck;
    s_lo = (uint64_t)st; s_hi = (uint64_t)(st >> 64);
    const uint64_t x = s_hi ^ s_lo;
    const unsigned rot = (unsigned)(s_hi >> 58);
    return ((x >> rot) | (x << ((64u - rot) & 63u))) >> 11;
}

// The state n units further on, a unit being the jump `unit` = (M^k, C_k) of k draws: binary powering of the affine map
// s -> M^k s + C_k inc (the maps commute: all are powers of one LCG step), 2 log2(n) compositions instead of n steps.
MGX_HD void pcg64_advance(uint64_t &s_lo, uint64_t &s_hi, uint64_t inc_lo, uint64_t inc_hi, const uint64_t *unit, uint32_t n) {
    typedef unsigned __int128 u128;
    u128 bm = ((u128)unit[1] << 64) | unit[0], bc = ((u128)unit[3] << 64) | unit[2], am = 1, ac = 0;
    for (; n != 0; n >>= 1) {
        if (n & 1u) { ac = ac * bm + bc; am = am * bm; }
        bc = bc * bm + bc;
        bm = bm * bm;
    }
    const u128 st = (((u128)s_hi << 64) | s_lo) * am + (((u128)inc_hi << 64) | inc_lo) * ac;
    s_lo = (uint64_t)st; s_hi = (uint64_t)(st >> 64);
}

// rank of draw `a` among the env's A draws = its position in argsort (stable; ties ~2^-53)   base.py:399
MGX_HD int draw_rank(const uint64_t *rnd, int A, int a) {
    const uint64_t ra = rnd[a];
    int rank = 0;
#pragma unroll 4                                                            // (4 LDS reads in flight instead of a round trip each)
    for (int b = 0; b < A; ++b) { const uint64_t rb = rnd[b]; rank += (rb < ra) | ((rb == ra) & (b < a)); }
    return rank;
}

// ---------------------------------------------------------------------------------------------------
// handle_actions for ONE env (multigrid/base.py:378-476 with on_success 478-507 / on_failure 509-532).
//   tile : H*W packed cells [y][x]       rows : A packed agent rows      act : A int8
//   ord  : A bytes, visiting order      rew  : A doubles, pre-zeroed (base.py:393)
//   dirty(off): called with the byte offset of every tile cell this step changed (tile already updated).
// Returns 0, or MGX_ERR_UNKNOWN_ACTION at the first invalid action in visiting order.
// ---------------------------------------------------------------------------------------------------
struct StepCfg {
    int W, H, A, max_steps;
    bool allow_overlap, joint_reward, success_any, failure_any;
    int cb;               // bytes per grid cell: 2 (MgxCell) or 1 (MgxCell8)
    int P;                // row pitch of the TILE in cells: W, or W - 1 in the resident kernels whose tile shares the WALL ring between
                          // rows and envs (mgx_fused.h: kShapes[].pitch): cell (x, y) lies at (y * P + x) * cb
};

MGX_HD StepCfg make_cfg(const MgxSpec &sp) {
    return StepCfg{sp.width, sp.height, sp.num_agents, sp.max_steps, sp.allow_agent_overlap != 0,
                   sp.joint_reward != 0, sp.success_any != 0, sp.failure_any != 0, sp.cell_bytes == 1 ? 1 : kCellBytes, sp.width};
}

MGX_HD void set_terminated(uint64_t *rows, int A, int i, bool all) {
    uint8_t *b = reinterpret_cast<uint8_t *>(rows);
    if (all) { for (int a = 0; a < A; ++a) b[a * MGX_AGENT_STRIDE + AG_TERM] = 1; }
    else b[i * MGX_AGENT_STRIDE + AG_TERM] = 1;
}

// base.py:478-507
MGX_HD void on_success(const StepCfg &cf, uint64_t *rows, int i, int32_t step_count, double *rew) {
    set_terminated(rows, cf.A, i, cf.success_any);
    const double r = reward_value(step_count, cf.max_steps);
    if (cf.joint_reward) { for (int a = 0; a < cf.A; ++a) rew[a] = r; }
    else rew[i] = r;
}

// base.py:426-427, 454-455: any agent, terminated or not
MGX_HD bool agent_present(const uint64_t *rows, int A, int x, int y) {
    const uint32_t want = (uint32_t)x | ((uint32_t)y << 8);
    bool hit = false;
#pragma unroll 4
    for (int a = 0; a < A; ++a) hit |= (((uint32_t)rows[a] >> 16) & 0xffffu) == want;
    return hit;
}

// What ONE agent's action does, evaluated against the current tile and agent rows (multigrid/base.py:403-474 for one
// iteration of the loop).  Written as straight-line predicated code: the lanes of a wavefront hold different agents
// taking different actions, so every branch that ANY lane takes is paid by all.
struct AgentEval {
    uint64_t nrow;        // the agent's row after the action
    uint32_t ncell;       // the front cell after the action
    int off;              // byte offset of the front cell in the tile
    bool go;              // the action is executed (agent present, not terminated, action known)
    bool bad;             // unknown action value (base.py:473-474)
    bool reads_cell;      // the outcome depends on the front cell's content
    bool writes;          // the front cell changes (pickup / drop / toggle)
    bool moved;           // the agent's position changes
    bool success;         // it stepped on a goal  (base.py:433-434)
    bool failure;         // it stepped on lava    (base.py:435-436)
    bool used_presence;   // the outcome depended on where the other agents stand (base.py:425-429, 453-456)
    bool unstale;         // it toggled the RedBlueDoors blue door whose object state had diverged from Grid.state
};

// `stale_off`: byte offset of a door whose WorldObj is CLOSED although Grid.state (the tile) says open, or -1.  Only
// RedBlueDoors produces one: its hook closes the blue door object without grid.update (envs/redbluedoors.py:185,
// SURVEY App. C Q9); the rules act on the object (Grid.get), observations on Grid.state.
MGX_HD AgentEval eval_agent(const StepCfg &cf, const uint8_t *tile, const uint64_t *rows, int action, uint64_t row,
                            bool alive, int stale_off = -1) {
    const int A = cf.A;
    AgentEval ev;
    // base.py:403-404 absent, 408-409 terminated
    const bool live = alive & (action >= 0) & !row_term(row);
    ev.bad = live & (action > ACT_DONE);                                    // base.py:473-474
    ev.go = live & (action <= ACT_DONE);
    const int d = row_dir(row), x = row_x(row), y = row_y(row);
    const int fx = x + dir_dx(d), fy = y + dir_dy(d);                       // agent.py:111-118
    const bool inb = ((unsigned)fx < (unsigned)cf.W) & ((unsigned)fy < (unsigned)cf.H);  // walled grids: always
    ev.off = inb ? (fy * cf.P + fx) * cf.cb : 0;
    const uint32_t raw = load_cell_raw(cf.cb, tile + ev.off);
    const uint32_t cell = cf.cb == 1 ? cell8_unpack(raw) : cell_unpack(raw);       // what the cell shows (a box's content: below)
    const uint32_t type = cell & 0xff, gstate = cell_state(cell);
    const bool stale = inb & (ev.off == stale_off);
    const uint32_t state = stale ? (uint32_t)S_CLOSED : gstate;             // the WorldObj's own state
    const uint32_t carry = row_carry(row), ctype = carry & 0xff;
    const bool on_cell = ev.go & inb;
    ev.reads_cell = on_cell & (action >= ACT_FORWARD) & (action <= ACT_TOGGLE);
    // base.py:412-417 turns
    const int nd = (action == ACT_LEFT) ? ((d + 3) & 3) : ((action == ACT_RIGHT) ? ((d + 1) & 3) : d);
    // base.py:420-436 forward; world_object.py:197-201, 287, 314, 339, 452 can_overlap
    const bool overlap = (type == T_EMPTY) | (type == T_GOAL) | (type == T_FLOOR) | (type == T_LAVA)
                       | ((type == T_DOOR) & (state == S_OPEN));
    bool fwd = on_cell & (action == ACT_FORWARD) & overlap;
    // base.py:449-459 drop
    bool drop = on_cell & (action == ACT_DROP) & (ctype != T_EMPTY) & (type == T_EMPTY);
    ev.used_presence = (fwd & !cf.allow_overlap) | drop;
    if (ev.used_presence) {                                                  // base.py:425-429, 453-456
        const bool present = agent_present(rows, A, fx, fy);
        fwd = fwd & (cf.allow_overlap | !present);
        drop = drop & !present;
    }
    // base.py:439-446 pickup; world_object.py:518, 556, 587 can_pickup
    const bool pick = on_cell & (action == ACT_PICKUP) & (ctype == T_EMPTY)
                    & ((type == T_KEY) | (type == T_BALL) | (type == T_BOX));
    // base.py:462-467 toggle; world_object.py:458-474 Door.toggle, 599-605 Box.toggle (the box is replaced by what it holds)
    const bool tog = on_cell & (action == ACT_TOGGLE);
    const bool unlock = (ctype == T_KEY) & (((carry >> 8) & 0xff) == ((cell >> 8) & 0xff));
    const uint32_t ns = (state == S_LOCKED) ? (unlock ? (uint32_t)S_OPEN : state)
                                            : ((state == S_OPEN) ? (uint32_t)S_CLOSED : (uint32_t)S_OPEN);
    const bool door = tog & (type == T_DOOR) & (ns != gstate);               // Door.toggle ends with grid.update
    const bool box = tog & (type == T_BOX);

    uint32_t ncell = cell;
    ncell = door ? ((cell & 0xffffu) | (ns << 16)) : ncell;
    ncell = (pick | box) ? CELL_EMPTY : ncell;
    ncell = drop ? carry : ncell;                                            // (a carried box takes its content along: `carry` has it)
    uint32_t ncarry = pick ? cell : (drop ? CELL_EMPTY : carry);
    // a box that holds something (world_object.py:574-605): picked up with its content, replaced by it when toggled
    if (MGX_ANY_LANE((cf.cb != 1) & ((raw & 0x48f0u) != 0) & (pick | box))) {
        const uint32_t full = cell_unpack_full(raw);
        ncell = box ? box_content_cell(full) : ncell;
        ncarry = pick ? full : ncarry;
    }
    ev.ncell = ncell;
    uint64_t nrow = row_set_dir(row, nd);
    nrow = row_set_pos(nrow, fwd ? fx : x, fwd ? fy : y);
    ev.nrow = row_set_carry(nrow, ncarry);
    ev.unstale = tog & stale;                                               // Door.toggle + grid.update: object == grid again
    ev.writes = door | pick | box | drop | ev.unstale;                      // (an unstale 'writes' the cell's meaning)
    ev.moved = fwd;
    ev.success = fwd & (type == T_GOAL);
    ev.failure = fwd & (type == T_LAVA);
    return ev;
}

// The reference's loop (base.py:402-474): agents act one after the other in `ord`, each seeing the previous ones' effects.
// `stale` points at the env's stale-door flag (aux[4] of a RedBlueDoors env) or is NULL.
MGX_HD int stale_offset(const StepCfg &cf, const uint8_t *aux, int env_kind) {
    return (env_kind == MGX_KIND_REDBLUEDOORS && aux[4]) ? (aux[1] * cf.P + aux[0]) * cf.cb : -1;
}

// ONE iteration of that loop: agent `i` takes its turn against the CURRENT tile and rows and its effects are committed.
// Returns true for an unknown action (base.py:473-474).  `alive` = no unknown action has been met earlier in the visiting order
// (the reference has raised by then).  The kernel's fallback runs it with one lane per env whose turn it is (mgx_fused_body.inc,
// P1c); handle_actions below is the same loop on one thread.
template <class Dirty>
MGX_HD bool agent_turn(const StepCfg &cf, uint8_t *tile, uint64_t *rows, int i, int action, bool alive, double *rew,
                       int32_t step_count, Dirty dirty, uint8_t *aux = nullptr, int env_kind = MGX_KIND_EMPTY) {
    const int so = aux ? stale_offset(cf, aux, env_kind) : -1;
    const AgentEval ev = eval_agent(cf, tile, rows, action, rows[i], alive, so);
    if (ev.go) rows[i] = ev.nrow;
    if (ev.unstale) aux[4] = 0;
    if (ev.writes) { store_cell(cf.cb, tile + ev.off, ev.ncell); dirty(ev.off); }
    if (ev.success) on_success(cf, rows, i, step_count, rew);                // base.py:433-434
    if (ev.failure) set_terminated(rows, cf.A, i, cf.failure_any);           // base.py:435-436, 509-532
    return ev.bad;
}

template <class Dirty>
MGX_HD int handle_actions(const StepCfg &cf, uint8_t *tile, uint64_t *rows, const int8_t *act,
                          const uint8_t *ord, double *rew, int32_t step_count, Dirty dirty,
                          uint8_t *aux = nullptr, int env_kind = MGX_KIND_EMPTY, int k0 = 0) {
    const int A = cf.A;
    int rc = 0;
    for (int k = k0; k < A; ++k) {                   // k0 = 1: the first visited agent's turn has been committed by the caller
        const int i = (A == 1) ? 0 : (ord[k] & 0x1f);        // (the kernel marks entries in bits 6, 7)
        // (after an unknown action the reference has raised: the later agents do nothing)
        if (agent_turn(cf, tile, rows, i, act[i], rc == 0, rew, step_count, dirty, aux, env_kind)) rc = MGX_ERR_UNKNOWN_ACTION;
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Order-free fast path.  All A agents of an env are evaluated at once against the PRE-step state (one lane per agent).
// The result equals the sequential loop above whenever no agent's inputs could have been changed by another agent's
// action in the same step.  Sufficient conditions, checked per env:
//   (1) no action is unknown (the reference aborts the loop there);
//   (2) no agent writes a cell that another agent's outcome depends on (its front cell, when its action reads it);
//   (3) no outcome depended on the other agents' positions (drop; forward without agent overlap) while any agent moved.
// Each agent's own row is only touched by its own action, so with (1)-(3) every agent sees exactly the inputs it would
// see in the sequential loop.  Envs that fail a condition are simply run through handle_actions() instead.
//
// on_success / on_failure stay inside the path.  An event that terminates only its own agent (mode 'all') changes
// nobody else's inputs.  An event that terminates EVERY agent (mode 'any') is where the visiting order matters: agents
// visited after it find themselves terminated and do nothing (base.py:408-409).  So the agents' ranks in the visiting
// order are computed for such an env, the first episode-ending event in that order is the cutoff, agents up to and
// including it act as evaluated, the later ones only receive the terminated flag.  Rewards are assignments of one value
// (base.py:500-507), so they commute.
// ---------------------------------------------------------------------------------------------------------------
MGX_HD bool event_ends_all(const StepCfg &cf, const AgentEval &ev) {
    return (ev.success & cf.success_any) | (ev.failure & cf.failure_any);
}
MGX_HD bool event_ends_self(const StepCfg &cf, const AgentEval &ev) {
    return (ev.success & !cf.success_any) | (ev.failure & !cf.failure_any);
}

// Position in the visiting order of the first agent whose event ends the episode for all (A if there is none).
// `ord` = the visiting order (ord[k] = agent visited k-th), m_ends_all: bit j = agent j has such an event.
MGX_HD int event_cutoff(const uint8_t *ord, uint64_t m_ends_all, int A) {
    int cut = A;
    for (int k = A - 1; k >= 0; --k) cut = ((m_ends_all >> ord[k]) & 1ull) ? k : cut;
    return cut;
}

MGX_HD bool spec_cell_conflict(const int32_t *woff /* [A]: cell written by agent j, or -1 */, int A, int ai,
                               const AgentEval &ev) {
    bool c = false;
#pragma unroll 4
    for (int j = 0; j < A; ++j) c |= (j != ai) & (woff[j] == ev.off);
    return c & ev.reads_cell;
}

// m_*: bit j = agent j of this env
MGX_HD bool spec_needs_fallback(uint64_t m_bad, uint64_t m_conflict, uint64_t m_presence, uint64_t m_moved) {
    return (m_bad != 0) | (m_conflict != 0) | ((m_presence != 0) & (m_moved != 0));
}

// The sequential fallback, shortened (round 4).  In the reference's loop the agent visited at rank r sees the effects of the agents
// ranked before it.  Its order-free evaluation (against the PRE-step state) is therefore still exact when none of THOSE changed
// an input of it: its front cell (when its action reads it), or -- when its outcome used the other agents' positions -- anybody's
// position.  `prefix_blocked` says whether agent `i` (rank `my_rank`) is NOT such an agent; events and unknown actions always
// block (the sequential loop owns their bookkeeping).  The env's cutoff = the lowest blocked rank: the agents ranked below it are
// committed in parallel with the order-free results, the loop starts at the cutoff (one lane walking through 16 agents costs
// ~0.85 us each, and a launch lasts as long as its slowest wavefront: C5).
//   ord: the env's visiting order (ord[k] & 0x1f = agent visited k-th)   woff: [A] cell offset each agent writes or -1 (readable
//   garbage when !woff_valid: no agent of the wavefront writes at all)   m_moved: bit j = agent j of THIS env moves
MGX_HD bool prefix_blocked(const AgentEval &ev, int my_rank, const uint8_t *ord, const int32_t *woff, bool woff_valid,
                            uint64_t m_moved, int A) {
    bool block = ev.bad | ev.success | ev.failure;
    (void)A;
    for (int k = 0; k < my_rank; ++k) {
        const int j = ord[k] & 0x1f;                         // (bits 6, 7: the kernel's marks; agents are < MGX_MAX_AGENTS = 32)
        block |= (ev.reads_cell & woff_valid & (woff[j] == ev.off)) | (ev.used_presence & (bool)((m_moved >> j) & 1ull));
    }
    return block;
}

// The env subclasses' step() post-hooks, run after the base step on the CLEAN tile (no agent overlay) with the
// post-step agent rows; `aux` is the env's 16-byte hook state (include/mgx.h).  The observation is not affected (the
// reference renders before the hook); `terminated` / `reward` outputs are.
//   BlockedUnlockPickup  envs/blockedunlockpickup.py:166-175
//   RedBlueDoors         envs/redbluedoors.py:170-187
//   LockedHallway        envs/locked_hallway.py:203-227
// `order`: the two hooks above iterate `actions.items()`, i.e. the insertion order of the caller's dict; A agent indices in that
// order, or nullptr for ascending index (a dict built in agent order).  It matters when two agents toggle the same door in one
// step (who ends the episode / who is paid for the unlock).  BlockedUnlockPickup iterates self.agents (ascending).
template <class Dirty>
MGX_HD void post_step_hook(const StepCfg &cf, int env_kind, uint8_t *tile, uint64_t *rows, const int8_t *act,
                           uint8_t *aux, int32_t step_count, double *rew, Dirty dirty, const uint8_t *order = nullptr) {
    const int A = cf.A;
    if (env_kind == MGX_KIND_BLOCKEDUNLOCKPICKUP) {
        const uint32_t want = (uint32_t)aux[0] | ((uint32_t)aux[1] << 8);
        for (int a = 0; a < A; ++a)
            if ((row_carry(rows[a]) & 0xffffu) == want) on_success(cf, rows, a, step_count, rew);
    } else if (env_kind == MGX_KIND_REDBLUEDOORS) {
        const int boff = (aux[1] * cf.P + aux[0]) * cf.cb, roff = (aux[3] * cf.P + aux[2]) * cf.cb;
        for (int ko = 0; ko < A; ++ko) {                                        // `for agent_id, action in actions.items()`
            const int a = order ? (int)order[ko] : ko;
            if (a >= A || act[a] != ACT_TOGGLE) continue;
            const uint64_t r = rows[a];
            const int d = row_dir(r), fx = row_x(r) + dir_dx(d), fy = row_y(r) + dir_dy(d);
            if (fx != aux[0] || fy != aux[1]) continue;                         // fwd_obj == self.blue_door
            if (cell_state(load_cell(cf.cb, tile + boff)) != S_OPEN || aux[4]) continue;                   // ... and self.blue_door.is_open (the OBJECT)
            if (cell_state(load_cell(cf.cb, tile + roff)) == S_OPEN) {
                on_success(cf, rows, a, step_count, rew);
            } else {
                set_terminated(rows, A, a, cf.failure_any);                      // on_failure
                aux[4] = 1;                              // blue_door.is_open = False, no grid.update: Grid.state stays open
            }
        }
    } else if (env_kind == MGX_KIND_LOCKEDHALLWAY) {
        // aux[0] < 0x80: the explicit format (n <= 6 doors with their positions); aux[0] & 0x80: the geometric format for
        // more rooms -- doors sit in the middle of their room's wall (add_door(..., rand_pos=False)), so door k = (row k/2,
        // side k%2) is at x = side ? 2(rs-1) : rs-1, y = row(rs-1) + (rs-1)/2 (roomgrid.py:108: (top + bottom) // 2); 16-bit unlocked mask in aux[1], aux[2];
        // aux[3] = rs; aux[4] = len(self.rooms) (a dict keyed by colour: < n when colours repeat, include/mgx.h)
        const bool geo = (aux[0] & 0x80) != 0;
        const int nd = aux[0] & 0x7f, rs = aux[3];
        const int target = geo ? (int)aux[4] : nd;
        uint32_t mask = geo ? ((uint32_t)aux[1] | ((uint32_t)aux[2] << 8)) : (uint32_t)aux[1];
        for (int ko = 0; ko < A; ++ko) {                                        // `for agent_id, action in actions.items()`
            const int a = order ? (int)order[ko] : ko;
            if (a >= A || act[a] != ACT_TOGGLE) continue;
            const uint64_t r = rows[a];
            const int d = row_dir(r), fx = row_x(r) + dir_dx(d), fy = row_y(r) + dir_dy(d);
            if ((unsigned)fx >= (unsigned)cf.W || (unsigned)fy >= (unsigned)cf.H) continue;
            const uint32_t c = load_cell(cf.cb, tile + (fy * cf.P + fx) * cf.cb);
            if ((c & 0xff) != T_DOOR || cell_state(c) == S_LOCKED) continue;                   // isinstance(Door) and not is_locked
            int k = -1;
            if (geo) {
                const int side = (fx == 2 * (rs - 1)) ? 1 : ((fx == rs - 1) ? 0 : -1);
                const int yy = fy - (rs - 1) / 2;
                if (side >= 0 && yy >= 0 && yy % (rs - 1) == 0) k = 2 * (yy / (rs - 1)) + side;
                if (k >= nd) k = -1;
            } else {
                for (int j = 0; j < nd; ++j)
                    if (aux[2 + 2 * j] == fx && aux[3 + 2 * j] == fy) { k = j; break; }
            }
            if (k < 0) continue;
            if (!((mask >> k) & 1u)) {                                           // not yet in self.unlocked_doors
                mask |= 1u << k;
                const double rv = reward_value(step_count, cf.max_steps);
                if (cf.joint_reward) { for (int b = 0; b < A; ++b) rew[b] += rv; }   // `+=`, not `=`
                else rew[a] += rv;
            }
        }
        aux[1] = (uint8_t)mask;
        if (geo) aux[2] = (uint8_t)(mask >> 8);
        int cnt = 0;
        for (int k = 0; k < nd; ++k) cnt += (mask >> k) & 1u;
        aux[15] = (uint8_t)(cnt == target);      // len(unlocked_doors) == len(rooms): `terminations` only, not agent state
#if MGX_RULES_KIND
    } else if (env_kind == MGX_KIND_RULES) {
        // the declared hook of a user-defined env (include/mgx.h: MGX_KIND_RULES): rule by rule in table order
        const int n = aux[0] < MGX_MAX_RULES ? aux[0] : MGX_MAX_RULES;
        for (int k = 0; k < n; ++k) {
            const uint8_t *r = aux + 1 + 5 * k;
            const int op = r[0], effect = r[3], cond = r[4];
            for (int ko = 0; ko < A; ++ko) {
                bool hit = false;
                int a = ko;
                if (op == MGX_RULE_CARRIES) {                                    // `for agent in self.agents: if carrying == obj`
                    hit = (row_carry(rows[a]) & 0xffffu) == ((uint32_t)r[1] | ((uint32_t)r[2] << 8));
                } else if (op == MGX_RULE_TOGGLES_AT) {                          // `for agent_id, action in actions.items()`
                    a = order ? (int)order[ko] : ko;
                    if (a >= A || act[a] != ACT_TOGGLE) continue;
                    const uint64_t row = rows[a];
                    const int d = row_dir(row), fx = row_x(row) + dir_dx(d), fy = row_y(row) + dir_dy(d);
                    hit = (fx == r[1]) & (fy == r[2]);
                    if (hit && cond != MGX_COND_ALWAYS) {
                        const uint32_t c = load_cell(cf.cb, tile + (r[2] * cf.P + r[1]) * cf.cb);
                        const bool open = ((c & 0xff) == T_DOOR) & (cell_state(c) == S_OPEN);
                        hit = (c & 0xff) == T_DOOR && (cond == MGX_COND_DOOR_OPEN ? open : !open);
                    }
                }
                if (!hit) continue;
                if (effect == MGX_EFFECT_SUCCESS) on_success(cf, rows, a, step_count, rew);
                else if (effect == MGX_EFFECT_FAILURE) set_terminated(rows, A, a, cf.failure_any);
            }
        }
#endif
    }
}

// obs.py:163-173: overlay every non-terminated agent's (10, color, dir) on the tile, ascending index.
MGX_HD void overlay_agents(const StepCfg &cf, uint8_t *tile, const uint64_t *rows) {
    if (cf.A <= 1) return;
    for (int a = 0; a < cf.A; ++a) {
        const uint64_t r = rows[a];
        if (row_term(r)) continue;
        const int x = row_x(r), y = row_y(r);
        if (x >= cf.W || y >= cf.H) continue;
        store_cell_raw(cf.cb, tile + (y * cf.P + x) * cf.cb, agent_cell_raw(cf.cb, r));
    }
}

// Same overlay, one call per agent (one lane per agent): where several live agents share a cell the reference's
// ascending loop leaves the highest index visible, so only that agent writes.  Returns the cell's byte offset or -1.
MGX_HD int overlay_offset(const StepCfg &cf, const uint64_t *rows, int ai) {
    if (cf.A <= 1) return -1;
    const uint64_t r = rows[ai];
    const uint32_t pos = ((uint32_t)r >> 16) & 0xffffu;
    bool shadowed = false;
#pragma unroll 4
    for (int j = 0; j < cf.A; ++j) {
        const uint64_t o = rows[j];
        shadowed |= (j > ai) & !row_term(o) & ((((uint32_t)o >> 16) & 0xffffu) == pos);
    }
    const int x = row_x(r), y = row_y(r);
    if (row_term(r) | shadowed | (x >= cf.W) | (y >= cf.H)) return -1;
    return (y * cf.P + x) * cf.cb;
}

// Layout of a restarting env (include/mgx.h: MgxAutoReset): (first_env + b + episode * 7919) mod K.  The 64-bit remainder costs
// ~150 scalar instructions on the ONE wavefront of the launch that restarts an env -- and a launch lasts as long as its slowest
// wavefront -- so the common case (pool of at most 2^19 layouts, env index below 2^32) takes three exact 32-bit remainders by
// multiplication instead (Lemire, "Faster remainder by direct computation", 2019: a mod d = hi64(((M * a) mod 2^64) * d),
// M = ceil(2^64 / d), exact for every 32-bit a and d); anything else takes the division.
MGX_HD uint32_t fastmod_u32(uint32_t a, uint64_t M, uint32_t d) {
    const uint64_t low = M * a;                                             // mod 2^64
    const uint64_t hi_part = (low >> 32) * d, lo_part = (low & 0xffffffffull) * d;
    return (uint32_t)((hi_part + (lo_part >> 32)) >> 32);
}
MGX_HD int pool_index(int64_t first_env, int64_t b, int32_t ep, int32_t K, uint64_t M) {
    if (K == 1) return 0;
    const uint64_t g = (uint64_t)(first_env + b);
    if (K <= (1 << 19) && (g >> 32) == 0 && ep >= 0) {
        const uint32_t k = (uint32_t)K;
        const uint32_t x = fastmod_u32((uint32_t)g, M, k) + fastmod_u32((uint32_t)ep, M, k) * (7919u % k);   // < 2^32 (K <= 2^19)
        return (int)fastmod_u32(x, M, k);
    }
    return (int)((uint64_t)(first_env + b + (int64_t)ep * 7919) % (uint64_t)K);
}

// ---------------------------------------------------------------------------------------------------
// View geometry.  image[i][j] shows world cell  pos + fw*forward + la*right,  fw = V-1-j, la = i - V/2,
// right = (-dy, dx)  (equivalent to obs.py:182-202 + get_view_exts 275-316).  In tile byte offsets:
//     off(i, j) = origin + fw*stepF + la*stepL
// and the in-bounds region is a rectangle in (fw, la): fw <= fmax, imin <= i <= imax.
// ---------------------------------------------------------------------------------------------------
struct ViewGeom { int origin, stepF, stepL, fmax, imin, imax; };

template <int V>
MGX_HD ViewGeom view_geom(int W, int H, int x, int y, int d, int cb = kCellBytes, int pitch = 0) {
    const int dx = dir_dx(d), dy = dir_dy(d), h = V / 2;
    const int P = pitch ? pitch : W;                      // (the tile's row pitch: StepCfg::P)
    ViewGeom g;
    g.origin = (y * P + x) * cb;
    g.stepF = (dy * P + dx) * cb;
    g.stepL = (dx * P - dy) * cb;
    // room ahead and to both sides, as selects (one lane per view: the lanes hold all four directions)
    const bool d0 = d == 0, d1 = d == 1, d2 = d == 2;
    g.fmax = d0 ? W - 1 - x : (d1 ? H - 1 - y : (d2 ? x : y));
    const int lmin = d0 ? -y : (d1 ? x - (W - 1) : (d2 ? y - (H - 1) : -x));
    const int lmax = d0 ? H - 1 - y : (d1 ? x : (d2 ? y : W - 1 - x));
    g.imin = lmin + h < 0 ? 0 : lmin + h;
    g.imax = lmax + h > V - 1 ? V - 1 : lmax + h;
    return g;
}

// The same rectangle for the CLAMPED gather (round 3).  Every env's outer ring of cells is WALL = (wall, grey, 0) -- the
// reference's envs all start from Grid.wall_rect(0, 0, W, H) (core/grid.py:183-218) and nothing can change a wall, a
// precondition of include/mgx.h that layouts.check_walled enforces on import -- and WALL is exactly what obs.py:199-202 shows
// for a world cell outside the grid.  A view cell outside the grid may therefore read the nearest cell INSIDE it: clamp
// (fw, la) to the rectangle, one packed-i16 max + min per cell instead of a 64-bit mask per view (two v_readlane + a select
// per cell), and the tile offset is one dot product with the packed steps:
//     off = origin + dot2((fw', la'), (stepF, stepL)),   (fw', la') = min(max((fw, la), lo), hi)
// all three as i16 pairs, low half = forward.  An agent that stands outside the grid (no valid state has one) sees only walls:
// `valid` is false and the caller points the view at its WALL cell.
struct ViewClamp { uint32_t steps, lo, hi; bool valid; };

template <int V>
MGX_HD ViewClamp view_clamp(const ViewGeom &g, int W, int H, int x, int y) {
    constexpr int h = V / 2;
    ViewClamp c;
    const int fhi = g.fmax > V - 1 ? V - 1 : g.fmax;
    c.steps = ((uint32_t)g.stepF & 0xffffu) | ((uint32_t)g.stepL << 16);
    c.lo = (uint32_t)(g.imin - h) << 16;                                     // forward: from 0
    c.hi = ((uint32_t)fhi & 0xffffu) | ((uint32_t)(g.imax - h) << 16);
    c.valid = (x < W) & (y < H) & (g.fmax >= 0) & (g.imax >= g.imin);
    return c;
}

// (host restatement of the three instructions, for tests/hostshim: v_pk_max_i16, v_pk_min_i16, v_dot2_i32_i16)
MGX_HD int clamped_offset(int origin, const ViewClamp &c, int fw, int la) {
    auto lo16 = [](uint32_t v) { return (int)(int16_t)(v & 0xffffu); };
    auto hi16 = [](uint32_t v) { return (int)(int16_t)(v >> 16); };
    int f = fw > lo16(c.lo) ? fw : lo16(c.lo), l = la > hi16(c.lo) ? la : hi16(c.lo);
    f = f < lo16(c.hi) ? f : lo16(c.hi); l = l < hi16(c.hi) ? l : hi16(c.hi);
    return origin + f * lo16(c.steps) + l * hi16(c.steps);
}

// in-bounds bit mask in lane order k = j*V + i
template <int V, int NW>
MGX_HD void inbounds_mask(const ViewGeom &g, uint64_t (&m)[NW]) {
    const bool any = g.imax >= g.imin;
    const uint32_t cols = any ? (((2u << g.imax) - 1u) & ~((1u << g.imin) - 1u)) : 0u;
    const int jmin = (V - 1 - g.fmax) < 0 ? 0 : (V - 1 - g.fmax);
    if constexpr (NW == 1) {
        // the column pattern repeated in every row by doubling, then the rows nearer than jmin cut off
        uint64_t r = cols;
        r |= r << V;
        r |= r << (2 * V);
        if (V > 4) r |= r << (4 * V);
        constexpr uint64_t kAll = (V * V >= 64) ? ~0ull : ((1ull << ((V * V) & 63)) - 1ull);
        m[0] = (jmin >= V) ? 0ull : (r & kAll & (~0ull << (jmin * V)));
    } else {
        for (int k = 0; k < NW; ++k) m[k] = 0;
#pragma unroll
        for (int j = 0; j < V; ++j)
            if (j >= jmin) or_bits<V, NW>(m, j * V, cols);
    }
}

}  // namespace mgx
