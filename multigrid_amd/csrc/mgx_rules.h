// mgx_rules.h -- per-environment integer rules shared by the HIP kernels (one lane = one env / one view).
//
// Everything here is straight-line integer code on plain pointers, so the same functions run inside the
// gfx950 kernels (pointers into LDS) and, compiled by g++ into tests/hostshim, against the CPU oracle in
// the `-m "not gpu"` suite.  There is no CPU product path: libmgx.so only ever launches the kernels.
//
// Semantics follow ini/multigrid (paths relative to the reference root); see SURVEY.md App. A.
#pragma once
#include <stdint.h>

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

// A cell packed little-endian into 24 bits: type | color << 8 | state << 16.
constexpr uint32_t CELL_EMPTY = 1u;                       // (1,0,0)  world_object.py:131-137
constexpr uint32_t CELL_WALL = 2u | (5u << 8);            // (2,5,0)  obs.py:14
constexpr uint32_t CELL_UNSEEN = 0u;                      // (0,0,0)  obs.py:15

// multigrid/core/constants.py:21-30 DIR_TO_VEC, branch-free: 0:(1,0) 1:(0,1) 2:(-1,0) 3:(0,-1)
MGX_HD int dir_dx(int d) { return (d == 0) - (d == 2); }
MGX_HD int dir_dy(int d) { return (d == 1) - (d == 3); }

// multigrid/utils/obs.py:46-63 see_behind, on a packed cell
MGX_HD bool see_behind(uint32_t c) {
    const uint32_t t = c & 0xffu;
    return !(t == T_WALL || (t == T_DOOR && ((c >> 16) & 0xffu) != S_OPEN));
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

// 1-D flood of seed bits m through transparent cells s (both v-bit), lighting the first blocker on each side:
// what the forward sweep (obs.py:257-262) and backward sweep (obs.py:265-271) do to one row.
template <int V>
MGX_HD uint32_t flood_row(uint32_t m, uint32_t s) {
    constexpr uint32_t F = (1u << V) - 1u;
    const uint32_t up = (m | ((s + (m & s)) ^ s)) & F;          // carry runs up through the transparent run
    const uint32_t mr = brev32(m) >> (32 - V), sr = brev32(s) >> (32 - V);
    const uint32_t dn = (mr | ((sr + (mr & sr)) ^ sr)) & F;     // same, towards lower bits
    return up | (brev32(dn) >> (32 - V));
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
template <int V, int NW>
MGX_HD void vis_mask(const uint64_t (&sb)[NW], uint64_t (&vis)[NW]) {
    constexpr uint32_t F = (1u << V) - 1u;
    for (int k = 0; k < NW; ++k) vis[k] = 0;
    uint32_t init = 1u << (V / 2);
#pragma unroll
    for (int j = V - 1; j >= 0; --j) {
        const uint32_t s = get_bits<V, NW>(sb, j * V);
        const uint32_t m = flood_row<V>(init, s);
        or_bits<V, NW>(vis, j * V, m);
        const uint32_t p = m & s;                                  // visible AND transparent
        init = (p | (p << 1) | (p >> 1)) & F;                      // lights (i-1, i, i+1) of row j-1
    }
}

// ---------------------------------------------------------------------------------------------------
// handle_actions + hooks for ONE env (multigrid/base.py:378-532, envs/blockedunlockpickup.py:166-175).
//   tile   : H*W*3 bytes, [y][x][c]                ag : A*8 bytes (packed agent rows)
//   rng    : 4 words, advanced in place            actions : A int8
//   ord    : A bytes of scratch (visiting order)   rnd : A u64 of scratch (the drawn 53-bit values)
//   rew    : A doubles (stride 1), fully written   dirty(off) : called with the byte offset of every
//            tile cell this step changed (after the tile bytes are updated).
// Returns 0, or MGX_ERR_UNKNOWN_ACTION at the first invalid action in visiting order.
// ---------------------------------------------------------------------------------------------------
MGX_HD uint32_t load_cell(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
MGX_HD void store_cell(uint8_t *p, uint32_t c) { p[0] = (uint8_t)c; p[1] = (uint8_t)(c >> 8); p[2] = (uint8_t)(c >> 16); }

// base.py:598-602 `1 - 0.9 * (step_count / max_steps)` in Python float arithmetic: three correctly rounded
// IEEE-754 binary64 operations, never contracted into an fma.
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

// base.py:478-507
MGX_HD void on_success(const MgxSpec &sp, uint8_t *ag, int i, int32_t step_count, double *rew) {
    const int A = sp.num_agents;
    if (sp.success_any) { for (int a = 0; a < A; ++a) ag[a * MGX_AGENT_STRIDE + AG_TERM] = 1; }
    else ag[i * MGX_AGENT_STRIDE + AG_TERM] = 1;
    const double r = reward_value(step_count, sp.max_steps);
    if (sp.joint_reward) { for (int a = 0; a < A; ++a) rew[a] = r; }
    else rew[i] = r;
}

// base.py:509-532
MGX_HD void on_failure(const MgxSpec &sp, uint8_t *ag, int i) {
    const int A = sp.num_agents;
    if (sp.failure_any) { for (int a = 0; a < A; ++a) ag[a * MGX_AGENT_STRIDE + AG_TERM] = 1; }
    else ag[i * MGX_AGENT_STRIDE + AG_TERM] = 1;
}

// base.py:426-427, 454-455: any agent, terminated or not
MGX_HD bool agent_present(const uint8_t *ag, int A, int x, int y) {
    bool hit = false;
    for (int a = 0; a < A; ++a)
        hit |= (ag[a * MGX_AGENT_STRIDE + AG_X] == x) & (ag[a * MGX_AGENT_STRIDE + AG_Y] == y);
    return hit;
}

// base.py:396-399: order = argsort(np_random.random(A)); stable ranking (ties ~2^-53)
MGX_HD void draw_order(int A, uint64_t *rng, uint64_t *rnd, uint8_t *ord) {
    if (A == 1) { ord[0] = 0; return; }
    uint64_t s_lo = rng[0], s_hi = rng[1];
    const uint64_t i_lo = rng[2], i_hi = rng[3];
    for (int a = 0; a < A; ++a) rnd[a] = pcg64_next53(s_lo, s_hi, i_lo, i_hi);
    rng[0] = s_lo; rng[1] = s_hi;
    for (int a = 0; a < A; ++a) {
        const uint64_t ra = rnd[a];
        int rank = 0;
        for (int b = 0; b < A; ++b) { const uint64_t rb = rnd[b]; rank += (rb < ra) | ((rb == ra) & (b < a)); }
        ord[rank] = (uint8_t)a;
    }
}

template <class Dirty>
MGX_HD int handle_actions(const MgxSpec &sp, uint8_t *tile, uint8_t *ag, uint64_t *rng, int32_t step_count,
                          const int8_t *actions, uint8_t *ord, uint64_t *rnd, double *rew, Dirty dirty) {
    const int W = sp.width, H = sp.height, A = sp.num_agents;
    for (int a = 0; a < A; ++a) rew[a] = 0.0;                                   // base.py:393
    draw_order(A, rng, rnd, ord);
    for (int k = 0; k < A; ++k) {
        const int i = ord[k];
        const int action = actions[i];
        if (action < 0) continue;                                                // base.py:403-404
        uint8_t *s = ag + i * MGX_AGENT_STRIDE;
        if (s[AG_TERM]) continue;                                                // base.py:408-409
        if (action > ACT_DONE) return MGX_ERR_UNKNOWN_ACTION;                    // base.py:473-474
        const int d = s[AG_DIR];
        if (action == ACT_LEFT) { s[AG_DIR] = (uint8_t)((d + 3) & 3); continue; }   // base.py:412-413
        if (action == ACT_RIGHT) { s[AG_DIR] = (uint8_t)((d + 1) & 3); continue; }  // base.py:416-417
        if (action == ACT_DONE) continue;                                        // base.py:470-471
        const int fx = s[AG_X] + dir_dx(d), fy = s[AG_Y] + dir_dy(d);           // agent.py:111-118
        if ((unsigned)fx >= (unsigned)W || (unsigned)fy >= (unsigned)H) continue; // walled grids: never taken
        const int off = (fy * W + fx) * 3;
        uint8_t *cp = tile + off;
        const uint32_t cell = load_cell(cp);
        const int type = cell & 0xff, state = (cell >> 16) & 0xff;
        const uint32_t carry = load_cell(s + AG_CARRY);
        if (action == ACT_FORWARD) {                                             // base.py:420-436
            const bool overlap = type == T_EMPTY || type == T_GOAL || type == T_FLOOR || type == T_LAVA
                              || (type == T_DOOR && state == S_OPEN);            // world_object.py:197-201,287,314,339,452
            if (!overlap) continue;
            if (!sp.allow_agent_overlap && agent_present(ag, A, fx, fy)) continue;
            s[AG_X] = (uint8_t)fx; s[AG_Y] = (uint8_t)fy;
            if (type == T_GOAL) on_success(sp, ag, i, step_count, rew);
            if (type == T_LAVA) on_failure(sp, ag, i);
        } else if (action == ACT_PICKUP) {                                       // base.py:439-446
            const bool can_pickup = type == T_KEY || type == T_BALL || type == T_BOX;  // world_object.py:518,556,587
            if (can_pickup && (carry & 0xff) == T_EMPTY) {
                store_cell(s + AG_CARRY, cell);
                store_cell(cp, CELL_EMPTY);
                dirty(off);
            }
        } else if (action == ACT_DROP) {                                         // base.py:449-459
            if ((carry & 0xff) != T_EMPTY && type == T_EMPTY && !agent_present(ag, A, fx, fy)) {
                store_cell(cp, carry);
                store_cell(s + AG_CARRY, CELL_EMPTY);
                dirty(off);
            }
        } else {                                                                 // toggle, base.py:462-467
            if (type == T_DOOR) {                                                // world_object.py:458-474
                int ns = state;
                if (state == S_LOCKED) {
                    if ((carry & 0xff) == T_KEY && ((carry >> 8) & 0xff) == ((cell >> 8) & 0xff)) ns = S_OPEN;
                } else {
                    ns = (state == S_OPEN) ? S_CLOSED : S_OPEN;
                }
                if (ns != state) { cp[2] = (uint8_t)ns; dirty(off); }
            } else if (type == T_BOX) {                                          // world_object.py:599-605
                store_cell(cp, CELL_EMPTY);                                      // `contains` is None in scope
                dirty(off);
            }
        }
    }
    return 0;
}

// envs/blockedunlockpickup.py:166-175, run AFTER the observation has been rendered.
MGX_HD void post_step_hook(const MgxSpec &sp, uint8_t *ag, const uint8_t *target, int32_t step_count, double *rew) {
    if (sp.env_kind != MGX_KIND_BLOCKEDUNLOCKPICKUP) return;
    const int A = sp.num_agents;
    for (int a = 0; a < A; ++a) {
        const uint8_t *c = ag + a * MGX_AGENT_STRIDE + AG_CARRY;
        if (c[0] == target[0] && c[1] == target[1]) on_success(sp, ag, a, step_count, rew);
    }
}

// obs.py:163-173: overlay every non-terminated agent's (10, color, dir) on the tile, ascending index.
MGX_HD void overlay_agents(const MgxSpec &sp, uint8_t *tile, const uint8_t *ag) {
    const int A = sp.num_agents;
    if (A <= 1) return;
    for (int a = 0; a < A; ++a) {
        const uint8_t *s = ag + a * MGX_AGENT_STRIDE;
        if (s[AG_TERM]) continue;
        if (s[AG_X] >= sp.width || s[AG_Y] >= sp.height) continue;
        uint8_t *c = tile + (s[AG_Y] * sp.width + s[AG_X]) * 3;
        c[0] = T_AGENT; c[1] = s[AG_COLOR]; c[2] = s[AG_DIR];
    }
}

}  // namespace mgx
