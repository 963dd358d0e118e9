/*
 * mgx_layout_oracle.c -- CPU restatement of the reference's episode-start generation for the env classes the device
 * generates on its own (mgx_reset_generate): EmptyEnv and BlockedUnlockPickupEnv.
 *
 * TEST INFRASTRUCTURE, like mgx_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * What is restated, draw for draw:
 *   numpy Generator(PCG64).integers(lo, hi) for int64 scalars -- numpy/random/_bounded_integers.pyx (_rand_int64) ->
 *     distributions.c random_bounded_uint64_fill -> buffered_bounded_lemire_uint32 over PCG64's next_uint32, which hands out
 *     the two halves of one 64-bit output (low half first, the high half buffered: pcg64.h pcg64_next32).  numpy is a
 *     dependency of the build itself and is present on both boxes; tests/test_layout_gen.py checks this restatement against
 *     numpy itself on mixed integers()/random() call sequences.
 *   multigrid/base.py:604-697       place_obj / place_agent (rejection sampling: non-empty cell, agent on the cell, reject_fn)
 *   multigrid/core/roomgrid.py:45-50 reject_next_to;  :238-259 place_in_room;  :376-404 place_agent (not facing an object)
 *   multigrid/envs/blockedunlockpickup.py:142-164  _gen_grid: box (right room), locked door (row from env.np_random,
 *                                   roomgrid.py:104-106), ball in front of the door, key (left room), agents (left room)
 *   multigrid/envs/empty.py:151-170 _gen_grid: fixed start, or place_agent over the whole grid
 * pinned through multigrid_amd/layouts.py (the host restatement the reset fixtures of the real reference pin): given numpy
 * generators in the same state, both produce the same bytes (tests/test_layout_gen.py).
 *
 * Generator state as the device keeps it: u64[5] = [state_lo, state_hi, inc_lo, inc_hi, buf], buf = has_uint32 << 32 | uinteger.
 */
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;

static uint64_t lay_pcg64_next64(uint64_t s[5]) {
    const u128 mult = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    u128 state = ((((u128)s[1]) << 64) | s[0]) * mult + ((((u128)s[3]) << 64) | s[2]);
    s[0] = (uint64_t)state; s[1] = (uint64_t)(state >> 64);
    const uint64_t x = s[1] ^ s[0];
    const unsigned rot = (unsigned)(s[1] >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

/* numpy/random/src/pcg64/pcg64.h pcg64_next32 */
static uint32_t lay_next32(uint64_t s[5]) {
    if (s[4] >> 32) { const uint32_t r = (uint32_t)s[4]; s[4] = r; return r; }     /* has_uint32 = 0; uinteger stays, as in numpy */
    const uint64_t next = lay_pcg64_next64(s);
    s[4] = (1ULL << 32) | (next >> 32);
    return (uint32_t)next;
}

/* Generator.integers(lo, hi), int64 scalars, hi - lo <= 2^32: distributions.c buffered_bounded_lemire_uint32 */
static int64_t lay_integers(uint64_t s[5], int64_t lo, int64_t hi) {
    const uint32_t rng = (uint32_t)(hi - 1 - lo);
    if (rng == 0) return lo;                                    /* no draw */
    const uint32_t rng_excl = rng + 1u;
    uint64_t m = (uint64_t)lay_next32(s) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) { m = (uint64_t)lay_next32(s) * rng_excl; leftover = (uint32_t)m; }
    }
    return lo + (int64_t)(m >> 32);
}

void mgo_np_integers(uint64_t s[5], int64_t lo, int64_t hi, int64_t n, int64_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = lay_integers(s, lo, hi);
}

enum { LT_EMPTY = 1, LT_WALL = 2, LT_DOOR = 4, LT_KEY = 5, LT_BALL = 6, LT_BOX = 7, LS_LOCKED = 2 };
static const int LDX[4] = {1, 0, -1, 0}, LDY[4] = {0, 1, 0, -1};

typedef struct { int W, H, A; uint8_t *grid; int ax[32], ay[32], adir[32]; } Lay;

static int lay_type(const Lay *L, int x, int y) { return L->grid[((size_t)y * L->W + x) * 3]; }
static void lay_set(Lay *L, int x, int y, int t, int c, int st) {
    uint8_t *p = L->grid + ((size_t)y * L->W + x) * 3; p[0] = (uint8_t)t; p[1] = (uint8_t)c; p[2] = (uint8_t)st;
}

/* base.py:604-669 place_obj; reject_next_to (roomgrid.py:45-50) when `next_to`.  Returns the position. */
static void lay_place(Lay *L, uint64_t *rng, int tx, int ty, int sw, int sh, int next_to, int *px, int *py) {
    if (tx < 0) tx = 0;
    if (ty < 0) ty = 0;
    const int xhi = tx + sw < L->W ? tx + sw : L->W, yhi = ty + sh < L->H ? ty + sh : L->H;
    for (;;) {
        const int x = (int)lay_integers(rng, tx, xhi), y = (int)lay_integers(rng, ty, yhi);
        if (lay_type(L, x, y) != LT_EMPTY) continue;
        int bad = 0;
        for (int a = 0; a < L->A; ++a) {
            if (L->ax[a] == x && L->ay[a] == y) bad = 1;
            const int dx = x - L->ax[a], dy = y - L->ay[a];
            if (next_to && dx * dx + dy * dy <= 1) bad = 1;              /* np.linalg.norm(pos - agent_pos) <= 1 */
        }
        if (bad) continue;
        *px = x; *py = y;
        return;
    }
}

static void lay_pack_agents(const Lay *L, uint8_t *agents) {
    for (int a = 0; a < L->A; ++a) {
        uint8_t *r = agents + a * 8;
        r[0] = (uint8_t)(a % 6); r[1] = (uint8_t)L->adir[a]; r[2] = (uint8_t)L->ax[a]; r[3] = (uint8_t)L->ay[a];
        r[4] = 0; r[5] = LT_EMPTY; r[6] = 0; r[7] = 0;
    }
}

/* blockedunlockpickup.py:142-164.  `grid` comes in as the blank room layout (walls only), u8[H][W][3]. */
int mgo_bup_layout(int room_size, int A, uint64_t lay_rng[5], uint64_t np_rng[5], uint8_t *grid, uint8_t *agents,
                   uint8_t *aux) {
    const int rs = room_size;
    Lay L; L.W = 2 * rs - 1; L.H = rs; L.A = A; L.grid = grid;
    for (int a = 0; a < A; ++a) { L.adir[a] = 0; L.ax[a] = (rs - 1) + rs / 2; L.ay[a] = rs / 2; }   /* roomgrid.py:232-236 */
    int x, y;
    const int box_color = (int)lay_integers(lay_rng, 0, 6);
    lay_place(&L, lay_rng, rs - 1, 0, rs, rs, 1, &x, &y);                       /* box in the right room */
    lay_set(&L, x, y, LT_BOX, box_color, 0);
    const int door_color = (int)lay_integers(lay_rng, 0, 6);
    const int door_x = rs - 1, door_y = (int)lay_integers(np_rng, 1, rs - 1);   /* roomgrid.py:104-106: env.np_random */
    lay_set(&L, door_x, door_y, LT_DOOR, door_color, LS_LOCKED);
    lay_set(&L, door_x - 1, door_y, LT_BALL, (int)lay_integers(lay_rng, 0, 6), 0);
    lay_place(&L, lay_rng, 0, 0, rs, rs, 1, &x, &y);                            /* key in the left room */
    lay_set(&L, x, y, LT_KEY, door_color, 0);
    for (int a = 0; a < A; ++a) {                                               /* roomgrid.py:376-404 */
        for (;;) {
            L.ax[a] = -1; L.ay[a] = -1;
            lay_place(&L, lay_rng, 0, 0, rs, rs, 0, &x, &y);
            L.ax[a] = x; L.ay[a] = y;
            L.adir[a] = (int)lay_integers(lay_rng, 0, 4);
            const int t = lay_type(&L, x + LDX[L.adir[a]], y + LDY[L.adir[a]]);
            if (t == LT_EMPTY || t == LT_WALL) break;
        }
    }
    lay_pack_agents(&L, agents);
    memset(aux, 0, 16);
    aux[0] = LT_BOX; aux[1] = (uint8_t)box_color; aux[2] = 0;
    return 0;
}

/* redbluedoors.py:142-168.  `grid` comes in as the blank layout (outer walls + the walls of the middle room), u8[H][W][3],
 * W = 2 * size, H = size.  All draws come from the construction-time generator. */
int mgo_rbd_layout(int size, int A, uint64_t lay_rng[5], uint8_t *grid, uint8_t *agents, uint8_t *aux) {
    Lay L; L.W = 2 * size; L.H = size; L.A = A; L.grid = grid;
    const int rx0 = L.W / 4, rw = L.W / 2;                                     /* room_top = (width // 4, 0), room_size = (width // 2, height) */
    for (int a = 0; a < A; ++a) { L.adir[a] = -1; L.ax[a] = -1; L.ay[a] = -1; }   /* Agent.reset */
    for (int a = 0; a < A; ++a) {                                               /* place_agent(agent, top=room_top, size=room_size) */
        int x, y;
        lay_place(&L, lay_rng, rx0, 0, rw, L.H, 0, &x, &y);
        L.ax[a] = x; L.ay[a] = y;
        L.adir[a] = (int)lay_integers(lay_rng, 0, 4);
    }
    const int ry = (int)lay_integers(lay_rng, 1, L.H - 1);                      /* red door in the left wall */
    lay_set(&L, rx0, ry, LT_DOOR, 0 /* red */, 1 /* closed */);
    const int bx = rx0 + rw - 1, by = (int)lay_integers(lay_rng, 1, L.H - 1);   /* blue door in the right wall */
    lay_set(&L, bx, by, LT_DOOR, 2 /* blue */, 1);
    lay_pack_agents(&L, agents);
    memset(aux, 0, 16);
    aux[0] = (uint8_t)bx; aux[1] = (uint8_t)by; aux[2] = (uint8_t)rx0; aux[3] = (uint8_t)ry;
    return 0;
}

/* empty.py:151-170 with agent_start_pos=None: place_agent over the whole grid.  `grid` = walls + goal. */
int mgo_empty_random_layout(int W, int H, int A, uint64_t lay_rng[5], uint8_t *grid, uint8_t *agents) {
    Lay L; L.W = W; L.H = H; L.A = A; L.grid = grid;
    for (int a = 0; a < A; ++a) { L.adir[a] = -1; L.ax[a] = -1; L.ay[a] = -1; }   /* Agent.reset, agent.py:120-133 */
    for (int a = 0; a < A; ++a) {
        int x, y;
        lay_place(&L, lay_rng, 0, 0, W, H, 0, &x, &y);
        L.ax[a] = x; L.ay[a] = y;
        L.adir[a] = (int)lay_integers(lay_rng, 0, 4);
    }
    lay_pack_agents(&L, agents);
    return 0;
}
