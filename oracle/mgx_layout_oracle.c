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

/* numpy Generator.shuffle on a Python list (the untyped path of _generator.pyx: `for i in reversed(range(1, n)): j =
 * random_interval(bitgen, i); x[i], x[j] = x[j], x[i]`) with distributions.c random_interval: masked rejection over next_uint32
 * (max <= 0xffffffff).  RandomMixin._rand_perm, multigrid/utils/random.py:75-83. */
static uint32_t lay_interval(uint64_t s[5], uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max, value;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    while ((value = (lay_next32(s) & mask)) > max) {}
    return value;
}
static void lay_shuffle(uint64_t s[5], int *x, int n) {
    for (int i = n - 1; i >= 1; --i) {
        const int j = (int)lay_interval(s, (uint32_t)i);
        const int t = x[i]; x[i] = x[j]; x[j] = t;
    }
}
void mgo_np_shuffle(uint64_t s[5], int64_t *x, int64_t n) {            /* (test hook: against numpy itself) */
    int tmp[64];
    for (int i = 0; i < n; ++i) tmp[i] = (int)x[i];
    lay_shuffle(s, tmp, (int)n);
    for (int i = 0; i < n; ++i) x[i] = tmp[i];
}

/* base.py:671-697 place_agent: pos = (-1,-1); place_obj(None, top, size); optional random direction */
static void lay_place_agent(Lay *L, uint64_t *rng, int a, int tx, int ty, int sw, int sh) {
    int x, y;
    L->ax[a] = -1; L->ay[a] = -1;
    lay_place(L, rng, tx, ty, sw, sh, 0, &x, &y);
    L->ax[a] = x; L->ay[a] = y;
    L->adir[a] = (int)lay_integers(rng, 0, 4);
}

/* locked_hallway.py:152-201.  `grid` comes in as the blank layout: the RoomGrid walls (3 columns of rooms, num_rooms / 2
 * rows) with the hallway's inner walls removed (remove_wall, roomgrid.py:333-374).  All draws come from the construction-time
 * generator (the doors sit mid-wall: rand_pos=False).  aux = the hook state of include/mgx.h. */
int mgo_lh_layout(int num_rooms, int room_size, int max_hallway_keys, int max_keys_per_room, int A, uint64_t lay_rng[5],
                  uint8_t *grid, uint8_t *agents, uint8_t *aux) {
    const int rs = room_size, rows = num_rooms / 2, n = num_rooms;
    Lay L; L.W = 3 * (rs - 1) + 1; L.H = rows * (rs - 1) + 1; L.A = A; L.grid = grid;
    for (int a = 0; a < A; ++a) { L.adir[a] = 0; L.ax[a] = (rs - 1) + rs / 2; L.ay[a] = (rows / 2) * (rs - 1) + rs / 2; }   /* roomgrid.py:232-236 */
    int seq[24], doors[16], room_of_color[6] = {-1, -1, -1, -1, -1, -1};
    const int Lc = 6 * ((n + 5) / 6);
    for (int k = 0; k < Lc; ++k) seq[k] = k % 6;                                /* list(Color) * ceil(n / 6) */
    lay_shuffle(lay_rng, seq, Lc);                                              /* _rand_perm(...)[:num_rooms] */
    for (int k = 0; k < n; ++k) doors[k] = seq[k];
    lay_shuffle(lay_rng, doors, n);                                             /* door_colors = _rand_perm(color_sequence) */
    int top = n;
    for (int row = 0; row < rows; ++row)
        for (int side = 0; side < 2; ++side) {                                  /* (LEFT, right), (RIGHT, left) */
            const int color = doors[--top];                                     /* door_colors.pop() */
            room_of_color[color] = row * 2 + side;                              /* self.rooms[color] = room (later rooms overwrite) */
            lay_set(&L, side ? 2 * (rs - 1) : rs - 1, row * (rs - 1) + (rs - 1) / 2, LT_DOOR, color, LS_LOCKED);
        }
    const int nhk = (int)lay_integers(lay_rng, 1, max_hallway_keys + 1);
    int x, y;
    for (int t = 0; t < nhk && t < n; ++t) {                                    /* keys in the hallway: column 1, the whole height */
        lay_place(&L, lay_rng, rs - 1, 0, rs, L.H, 0, &x, &y);
        lay_set(&L, x, y, LT_KEY, seq[t], 0);
    }
    int ki = nhk;
    while (ki < n) {                                                            /* keys in the rooms */
        const int r = room_of_color[seq[ki - 1]], row = r >> 1, side = r & 1;
        const int nrk = (int)lay_integers(lay_rng, 1, max_keys_per_room + 1);
        const int stop = ki + nrk < n ? ki + nrk : n;                           /* color_sequence[ki : ki + nrk] */
        for (int t = ki; t < stop; ++t) {
            lay_place(&L, lay_rng, side ? 2 * (rs - 1) : 0, row * (rs - 1), rs, rs, 0, &x, &y);
            lay_set(&L, x, y, LT_KEY, seq[t], 0);
            ++ki;
        }
    }
    for (int a = 0; a < A; ++a) lay_place_agent(&L, lay_rng, a, rs - 1, 0, rs, L.H);   /* MultiGridEnv.place_agent in the hallway */
    lay_pack_agents(&L, agents);
    memset(aux, 0, 16);
    if (n <= 6) {                                                               /* explicit door positions, sorted by (x, y) */
        aux[0] = (uint8_t)n;
        for (int k = 0; k < n; ++k) {
            const int side = k / rows, row = k % rows;
            aux[2 + 2 * k] = (uint8_t)(side ? 2 * (rs - 1) : rs - 1); aux[3 + 2 * k] = (uint8_t)(row * (rs - 1) + (rs - 1) / 2);
        }
    } else {
        int distinct = 0;
        for (int c = 0; c < 6; ++c) distinct += room_of_color[c] >= 0;
        aux[0] = (uint8_t)(0x80 | n); aux[3] = (uint8_t)rs; aux[4] = (uint8_t)distinct;   /* len(self.rooms) */
    }
    return 0;
}

/* playground.py:122-137 over RoomGrid (roomgrid.py:203-452).  `grid` = the walls of num_rows x num_cols rooms.  Door positions
 * come from env.np_random (Room.set_door_pos, roomgrid.py:104-124), everything else from the construction-time generator. */
int mgo_playground_layout(int room_size, int num_rows, int num_cols, int A, uint64_t lay_rng[5], uint64_t np_rng[5],
                          uint8_t *grid, uint8_t *agents) {
    const int rs = room_size, R = num_rows * num_cols;
    Lay L; L.W = num_cols * (rs - 1) + 1; L.H = num_rows * (rs - 1) + 1; L.A = A; L.grid = grid;
    for (int a = 0; a < A; ++a) { L.adir[a] = 0; L.ax[a] = (num_cols / 2) * (rs - 1) + rs / 2; L.ay[a] = (num_rows / 2) * (rs - 1) + rs / 2; }
    unsigned char door[64][4];
    memset(door, 0, sizeof door);
    static const int NDX[4] = {1, 0, -1, 0}, NDY[4] = {0, 1, 0, -1};
    for (int itr = 0; itr < 5000; ++itr) {                                      /* connect_all, roomgrid.py:406-452 */
        unsigned long long seen = 1ull;                                         /* bfs from room (0, 0) over the doors */
        for (int pass = 0; pass < R; ++pass)
            for (int r = 0; r < R; ++r)
                if ((seen >> r) & 1ull)
                    for (int d = 0; d < 4; ++d)
                        if (door[r][d]) seen |= 1ull << ((r / num_cols + NDY[d]) * num_cols + (r % num_cols + NDX[d]));
        if (seen == ((R >= 64) ? ~0ull : ((1ull << R) - 1ull))) break;
        const int col = (int)lay_integers(lay_rng, 0, num_cols), row = (int)lay_integers(lay_rng, 0, num_rows);
        const int d = (int)lay_integers(lay_rng, 0, 4);
        const int ncol = col + NDX[d], nrow = row + NDY[d], r = row * num_cols + col;
        if (ncol < 0 || ncol >= num_cols || nrow < 0 || nrow >= num_rows || door[r][d]) continue;
        const int color = (int)lay_integers(lay_rng, 0, 6);                     /* _rand_elem(door_colors) */
        const int left = col * (rs - 1), top = row * (rs - 1), right = left + rs - 1, bottom = top + rs - 1;
        int dx, dy;                                                             /* Room.set_door_pos(dir, random=np_random) */
        if (d == 0) { dx = right; dy = (int)lay_integers(np_rng, top + 1, bottom); }
        else if (d == 1) { dx = (int)lay_integers(np_rng, left + 1, right); dy = bottom; }
        else if (d == 2) { dx = left; dy = (int)lay_integers(np_rng, top + 1, bottom); }
        else { dx = (int)lay_integers(np_rng, left + 1, right); dy = top; }
        lay_set(&L, dx, dy, LT_DOOR, color, 1 /* closed */);
        door[r][d] = 1; door[nrow * num_cols + ncol][(d + 2) % 4] = 1;
    }
    int x, y;
    for (int k = 0; k < 12; ++k) {                                              /* 12 random objects */
        const int col = (int)lay_integers(lay_rng, 0, num_cols), row = (int)lay_integers(lay_rng, 0, num_rows);
        const int kind = LT_KEY + (int)lay_integers(lay_rng, 0, 3);             /* ['key', 'ball', 'box'] */
        const int color = (int)lay_integers(lay_rng, 0, 6);
        lay_place(&L, lay_rng, col * (rs - 1), row * (rs - 1), rs, rs, 1, &x, &y);   /* place_in_room: reject_next_to */
        lay_set(&L, x, y, kind, color, 0);
    }
    for (int a = 0; a < A; ++a) {                                               /* RoomGrid.place_agent: a random room */
        const int col = (int)lay_integers(lay_rng, 0, num_cols), row = (int)lay_integers(lay_rng, 0, num_rows);
        for (;;) {
            lay_place_agent(&L, lay_rng, a, col * (rs - 1), row * (rs - 1), rs, rs);
            const int t = lay_type(&L, L.ax[a] + LDX[L.adir[a]], L.ay[a] + LDY[L.adir[a]]);
            if (t == LT_EMPTY || t == LT_WALL) break;
        }
    }
    lay_pack_agents(&L, agents);
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
