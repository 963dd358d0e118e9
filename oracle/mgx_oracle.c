/*
 * mgx_oracle.c -- CPU restatement of ini/multigrid's step/observation path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; nothing under multigrid_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py replays every .npz fixture in tests/golden/ --
 * vectors produced by running the real reference in the build container (oracle/gen_golden.py) -- through
 * mgo_step_ref()/mgo_gen_obs_ref() and requires bit-equality of observations, rewards, terminations,
 * truncations, post-step grid/agent state and the PCG64 stream.  The reference itself ships no tests,
 * golden vectors or known answers for this path, so those fixtures are the pin.
 *
 * Each function follows the reference function cited above it (paths relative to the reference root).
 * The algorithm is restated in the reference's own shapes: grid_state is (W,H,3) indexed [x][y][c],
 * agent_state is (A,9) = [type,color,dir,x,y,terminated,carry_type,carry_color,carry_state], all "int"
 * (int64 here, as numpy's default).  The *_batch entry points at the bottom convert the product's packed
 * uint8 layout (include/mgx.h) to these shapes env by env so tests can compare tensors directly.
 *
 * Third-party arithmetic on the path: numpy's Generator(PCG64).random() and ndarray.argsort()
 * (multigrid/base.py:399).  PCG64 is restated from its published definition (O'Neill, PCG XSL-RR 128/64,
 * default 128-bit multiplier) and pinned against numpy itself in tests/test_oracle_golden.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* multigrid/core/constants.py:34-48 (Type), 91-97 (State), 100-107 (Direction) */
enum { T_UNSEEN = 0, T_EMPTY = 1, T_WALL = 2, T_FLOOR = 3, T_DOOR = 4, T_KEY = 5, T_BALL = 6, T_BOX = 7,
       T_GOAL = 8, T_LAVA = 9, T_AGENT = 10 };
enum { S_OPEN = 0, S_CLOSED = 1, S_LOCKED = 2 };
enum { C_GREY = 5 };
/* multigrid/core/actions.py:5-15 */
enum { A_LEFT = 0, A_RIGHT = 1, A_FORWARD = 2, A_PICKUP = 3, A_DROP = 4, A_TOGGLE = 5, A_DONE = 6 };
/* multigrid/core/agent.py:222-232 */
enum { AS_TYPE = 0, AS_COLOR = 1, AS_DIR = 2, AS_X = 3, AS_Y = 4, AS_TERMINATED = 5, AS_CARRY = 6, AS_DIM = 9 };
/* multigrid/core/constants.py:21-30 */
static const int DIR_TO_VEC[4][2] = { {1, 0}, {0, 1}, {-1, 0}, {0, -1} };
/* multigrid/utils/obs.py:14-15 */
static const int64_t WALL_ENCODING[3] = { T_WALL, C_GREY, 0 };
static const int64_t UNSEEN_ENCODING[3] = { T_UNSEEN, 0, 0 };
static const int64_t EMPTY_ENCODING[3] = { T_EMPTY, 0, 0 }; /* world_object.py:131-137 */

enum { KIND_EMPTY = 0, KIND_BLOCKEDUNLOCKPICKUP = 1, KIND_REDBLUEDOORS = 2, KIND_LOCKEDHALLWAY = 3, KIND_RULES = 4 };

/* Per-env hook state `aux` (int64[16]), mirroring the attributes the env subclasses keep:
 *   BlockedUnlockPickup  [0..2] = self.obj encoding                        (envs/blockedunlockpickup.py:147)
 *   RedBlueDoors         [0,1] = blue door (x,y), [2,3] = red door (x,y),  (envs/redbluedoors.py:158-168)
 *                        [4]   = 1 while the blue Door OBJECT is closed but grid.state still says open: the hook
 *                                closes the object without grid.update (redbluedoors.py:185)
 *   LockedHallway        [0] = number of doors, [1] = bit mask of doors already in self.unlocked_doors,
 *                        [2+2i, 3+2i] = door i (x,y), [15] = 1 if the hook reported all agents terminated  */
#define MGO_AUX 16

typedef struct {
    int32_t width, height, num_agents, view_size, max_steps;
    int32_t see_through_walls, allow_agent_overlap, joint_reward;
    int32_t success_any;  /* success_termination_mode == 'any'  (base.py:97) */
    int32_t failure_any;  /* failure_termination_mode == 'any'  (base.py:98) */
    int32_t env_kind;     /* KIND_* : which env-specific step() hook runs */
} MgoSpec;

#define MGO_ERR_UNKNOWN_ACTION (-2)
#define MGO_MAX_AGENTS 64
#define MGO_MAX_VIEW 15

/* ------------------------------------------------------------------------------------------------
 * numpy Generator(PCG64): pcg_setseq_128 step + XSL-RR output, then next_double = (u64 >> 11) * 2^-53.
 * state words: s[0] = state_lo, s[1] = state_hi, s[2] = inc_lo, s[3] = inc_hi.
 * ---------------------------------------------------------------------------------------------- */
typedef unsigned __int128 u128;

static uint64_t pcg64_next_u64(uint64_t s[4]) {
    const u128 mult = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    u128 state = ((u128)s[1] << 64) | s[0];
    u128 inc = ((u128)s[3] << 64) | s[2];
    state = state * mult + inc;
    s[0] = (uint64_t)state;
    s[1] = (uint64_t)(state >> 64);
    uint64_t hi = s[1], lo = s[0];
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
}

static double pcg64_next_double(uint64_t s[4]) {
    return (double)(pcg64_next_u64(s) >> 11) * (1.0 / 9007199254740992.0);
}

/* exported for the PCG64-vs-numpy known-answer test */
void mgo_pcg64_random(uint64_t s[4], int64_t n, double *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = pcg64_next_double(s);
}

/* multigrid/base.py:396-399: order = (0,) if one agent else np_random.random(size=A).argsort().
 * Insertion sort (stable) -- what numpy's introsort does for n <= 16; ties have probability ~2^-53. */
static void draw_order(uint64_t rng[4], int A, int *order) {
    double r[MGO_MAX_AGENTS];
    if (A == 1) { order[0] = 0; return; }
    for (int i = 0; i < A; ++i) { r[i] = pcg64_next_double(rng); order[i] = i; }
    for (int i = 1; i < A; ++i) {
        int oi = order[i], j = i - 1;
        while (j >= 0 && r[order[j]] > r[oi]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = oi;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Observation path: multigrid/utils/obs.py
 * ---------------------------------------------------------------------------------------------- */

/* obs.py:46-63 see_behind */
static int see_behind(const int64_t *world_obj) {
    if (world_obj[0] == T_WALL) return 0;
    if (world_obj[0] == T_DOOR && world_obj[2] != S_OPEN) return 0;
    return 1;
}

/* obs.py:275-316 get_view_exts */
static void get_view_exts(const int64_t *agent_state, int A, int v, int64_t *topX, int64_t *topY) {
    for (int a = 0; a < A; ++a) {
        const int64_t *s = agent_state + (size_t)a * AS_DIM;
        int64_t x = s[AS_X], y = s[AS_Y];
        topX[a] = 0; topY[a] = 0;   /* np.zeros for an unplaced agent (dir == -1) */
        switch ((int)s[AS_DIR]) {
        case 0: topX[a] = x;             topY[a] = y - v / 2;     break; /* right */
        case 1: topX[a] = x - v / 2;     topY[a] = y;             break; /* down  */
        case 2: topX[a] = x - v + 1;     topY[a] = y - v / 2;     break; /* left  */
        case 3: topX[a] = x - v / 2;     topY[a] = y - v + 1;     break; /* up    */
        }
    }
}

/* obs.py:130-209 gen_obs_grid.  scratch: W*H*3 int64 (grid_encoding copy). */
static void gen_obs_grid(const int64_t *grid_state, const int64_t *agent_state, int W, int H, int A, int v,
                         int64_t *scratch, int64_t *obs_grid) {
    const int64_t *grid_encoding = grid_state;
    if (A > 1) {                                                      /* obs.py:163-173 */
        memcpy(scratch, grid_state, sizeof(int64_t) * (size_t)W * H * 3);
        for (int a = 0; a < A; ++a) {
            const int64_t *s = agent_state + (size_t)a * AS_DIM;
            if (!s[AS_TERMINATED]) {
                int64_t *c = scratch + ((size_t)s[AS_X] * H + s[AS_Y]) * 3;
                c[0] = s[AS_TYPE]; c[1] = s[AS_COLOR]; c[2] = s[AS_DIR];
            }
        }
        grid_encoding = scratch;
    }
    int64_t topX[MGO_MAX_AGENTS], topY[MGO_MAX_AGENTS];
    get_view_exts(agent_state, A, v, topX, topY);                    /* obs.py:176-177 */
    for (int a = 0; a < A; ++a) {                                     /* obs.py:180-202 */
        const int64_t *s = agent_state + (size_t)a * AS_DIM;
        int rot = (int)((s[AS_DIR] + 1) % 4);
        int64_t *img = obs_grid + (size_t)a * v * v * 3;
        for (int i = 0; i < v; ++i) {
            for (int j = 0; j < v; ++j) {
                int64_t x = topX[a] + i, y = topY[a] + j;
                int i_rot = i, j_rot = j;
                if (rot == 1)      { i_rot = j;         j_rot = v - i - 1; }
                else if (rot == 2) { i_rot = v - i - 1; j_rot = v - j - 1; }
                else if (rot == 3) { i_rot = v - j - 1; j_rot = i; }
                const int64_t *src = (x >= 0 && x < W && y >= 0 && y < H)
                    ? grid_encoding + ((size_t)x * H + y) * 3 : WALL_ENCODING;
                int64_t *dst = img + ((size_t)i_rot * v + j_rot) * 3;
                memcpy(dst, src, 3 * sizeof(int64_t));
                dst[2] &= 3;                                          /* (a box's content is not part of Grid.state: below) */
            }
        }
        /* obs.py:207: the agent sees what it carries at its own cell */
        memcpy(img + ((size_t)(v / 2) * v + (v - 1)) * 3, s + AS_CARRY, 3 * sizeof(int64_t));
        img[((size_t)(v / 2) * v + (v - 1)) * 3 + 2] &= 3;
    }
}

/* obs.py:235-273 get_vis_mask (with 211-233 get_see_behind_mask): the sequential sweep, verbatim. */
static void get_vis_mask(const int64_t *obs_grid, int A, int v, uint8_t *vis_mask) {
    uint8_t sb[MGO_MAX_VIEW * MGO_MAX_VIEW];
    for (int a = 0; a < A; ++a) {
        const int64_t *img = obs_grid + (size_t)a * v * v * 3;
        uint8_t *vis = vis_mask + (size_t)a * v * v;
        for (int i = 0; i < v; ++i)
            for (int j = 0; j < v; ++j)
                sb[i * v + j] = (uint8_t)see_behind(img + ((size_t)i * v + j) * 3);
        memset(vis, 0, (size_t)v * v);
        vis[(v / 2) * v + (v - 1)] = 1;
        for (int j = v - 1; j >= 0; --j) {
            for (int i = 0; i < v - 1; ++i) {                         /* forward pass */
                if (vis[i * v + j] && sb[i * v + j]) {
                    vis[(i + 1) * v + j] = 1;
                    if (j > 0) { vis[(i + 1) * v + j - 1] = 1; vis[i * v + j - 1] = 1; }
                }
            }
            for (int i = v - 1; i > 0; --i) {                         /* backward pass */
                if (vis[i * v + j] && sb[i * v + j]) {
                    vis[(i - 1) * v + j] = 1;
                    if (j > 0) { vis[(i - 1) * v + j - 1] = 1; vis[i * v + j - 1] = 1; }
                }
            }
        }
    }
}

/* obs.py:65-102 gen_obs_grid_encoding.  Single env, reference layout.  out: (A,v,v,3) int64. */
int mgo_gen_obs_ref(const int64_t *grid_state, const int64_t *agent_state, int W, int H, int A, int v,
                    int see_through_walls, int64_t *out) {
    if (A < 1 || A > MGO_MAX_AGENTS || v < 3 || v > MGO_MAX_VIEW || !(v & 1)) return -1;
    int64_t *scratch = (int64_t *)malloc(sizeof(int64_t) * (size_t)W * H * 3);
    uint8_t *vis = (uint8_t *)malloc((size_t)A * v * v);
    gen_obs_grid(grid_state, agent_state, W, H, A, v, scratch, out);
    get_vis_mask(out, A, v, vis);
    if (!see_through_walls)
        for (size_t k = 0; k < (size_t)A * v * v; ++k)
            if (!vis[k]) memcpy(out + k * 3, UNSEEN_ENCODING, 3 * sizeof(int64_t));
    free(scratch); free(vis);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Action path: multigrid/base.py
 * ---------------------------------------------------------------------------------------------- */

/* base.py:598-602 _reward: 1 - 0.9 * (step_count / max_steps), Python float arithmetic (no fma). */
static double reward_value(int64_t step_count, int64_t max_steps) {
    volatile double q = (double)step_count / (double)max_steps;
    volatile double p = 0.9 * q;
    return 1.0 - p;
}

/* base.py:478-507 on_success */
static void on_success(const MgoSpec *sp, int64_t *agent_state, int i, int64_t step_count, double *rewards) {
    int A = sp->num_agents;
    if (sp->success_any) { for (int a = 0; a < A; ++a) agent_state[(size_t)a * AS_DIM + AS_TERMINATED] = 1; }
    else agent_state[(size_t)i * AS_DIM + AS_TERMINATED] = 1;
    double r = reward_value(step_count, sp->max_steps);
    if (sp->joint_reward) { for (int a = 0; a < A; ++a) rewards[a] = r; }
    else rewards[i] = r;
}

/* base.py:509-532 on_failure */
static void on_failure(const MgoSpec *sp, int64_t *agent_state, int i) {
    int A = sp->num_agents;
    if (sp->failure_any) { for (int a = 0; a < A; ++a) agent_state[(size_t)a * AS_DIM + AS_TERMINATED] = 1; }
    else agent_state[(size_t)i * AS_DIM + AS_TERMINATED] = 1;
}

/* world_object.py:197-201 (base), 287-291 Goal, 314-318 Floor, 339-343 Lava, 452-456 Door.can_overlap;
 * an empty cell is `None` in base.py:423. */
static int can_overlap(const int64_t *c) {
    return c[0] == T_EMPTY || c[0] == T_GOAL || c[0] == T_FLOOR || c[0] == T_LAVA
        || (c[0] == T_DOOR && c[2] == S_OPEN);
}
/* world_object.py:203-207 (base False), 518-522 Key, 556-560 Ball, 587-591 Box */
static int can_pickup(const int64_t *c) { return c[0] == T_KEY || c[0] == T_BALL || c[0] == T_BOX; }

/* Box.contains (multigrid/core/world_object.py:574-585): an attribute of the Python object that neither Grid.state nor an
 * observation shows.  This restatement keeps it where the product's byte layouts do (include/mgx.h "BOX CONTENTS"): in the upper
 * bits of the box cell's STATE value, state | kind << 2 | colour << 5, kind = 0 nothing, 1 key, 2 ball, 3 goal, 4 floor,
 * 5 lava, 6 wall, 7 door (closed, as Door(color) constructs it).  It moves with the cell through pickup / drop (the reference
 * moves the object), is masked wherever the reference reads Grid.state (observations, full_obs), and Box.toggle
 * (world_object.py:599-605: env.grid.set(*pos, self.contains)) puts it on the grid. */
static void box_content(const int64_t *box, int64_t *out) {
    static const int64_t kind_type[8] = {T_EMPTY, T_KEY, T_BALL, T_GOAL, T_FLOOR, T_LAVA, T_WALL, T_DOOR};
    const int64_t kind = (box[2] >> 2) & 7, color = (box[2] >> 5) & 7;
    out[0] = kind_type[kind]; out[1] = kind ? color : 0; out[2] = (kind == 7) ? S_CLOSED : 0;
}

/* base.py:426-427 / 454-455: any agent (terminated or not, any index) standing on (x,y) */
static int agent_present(const int64_t *agent_state, int A, int64_t x, int64_t y) {
    for (int a = 0; a < A; ++a)
        if (agent_state[(size_t)a * AS_DIM + AS_X] == x && agent_state[(size_t)a * AS_DIM + AS_Y] == y) return 1;
    return 0;
}

/* base.py:378-476 handle_actions.  actions[a] < 0 means "agent key absent" (base.py:403-404).
 * Returns 0, or MGO_ERR_UNKNOWN_ACTION at the first invalid action in visiting order (base.py:473-474;
 * earlier agents in the order have already acted, exactly as when the reference raises mid-loop). */
static int handle_actions(const MgoSpec *sp, int64_t *grid_state, int64_t *agent_state, uint64_t rng[4],
                          int64_t step_count, const int8_t *actions, double *rewards, int *order_out, int64_t *aux) {
    const int W = sp->width, H = sp->height, A = sp->num_agents;
    int order[MGO_MAX_AGENTS];
    for (int a = 0; a < A; ++a) rewards[a] = 0.0;                     /* base.py:393 */
    draw_order(rng, A, order);                                         /* base.py:396-399 */
    if (order_out) for (int a = 0; a < A; ++a) order_out[a] = order[a];
    for (int k = 0; k < A; ++k) {
        int i = order[k];
        int action = actions[i];
        if (action < 0) continue;                                      /* base.py:403-404 */
        int64_t *s = agent_state + (size_t)i * AS_DIM;
        if (s[AS_TERMINATED]) continue;                                /* base.py:408-409 */
        /* agent.py:111-118 front_pos -> utils/misc.py:7-13 */
        int64_t fx = s[AS_X], fy = s[AS_Y];
        if (s[AS_DIR] >= 0 && s[AS_DIR] < 4) { fx += DIR_TO_VEC[s[AS_DIR]][0]; fy += DIR_TO_VEC[s[AS_DIR]][1]; }
        int in_bounds = fx >= 0 && fx < W && fy >= 0 && fy < H;       /* shipped envs are walled: always true */
        int64_t *cell = in_bounds ? grid_state + ((size_t)fx * H + fy) * 3 : NULL;
        /* The rules below act on the WorldObj the grid holds (Grid.get, grid.py:102-117).  Its state equals
         * grid.state except for the RedBlueDoors blue door after the hook closed the object only (Q9). */
        const int stale = sp->env_kind == KIND_REDBLUEDOORS && aux && aux[4] && cell && fx == aux[0] && fy == aux[1];
        int64_t obj[3] = {0, 0, 0};
        if (cell) { obj[0] = cell[0]; obj[1] = cell[1]; obj[2] = stale ? S_CLOSED : cell[2]; }
        switch (action) {
        case A_LEFT:  s[AS_DIR] = (s[AS_DIR] + 3) % 4; break;          /* base.py:412-413 */
        case A_RIGHT: s[AS_DIR] = (s[AS_DIR] + 1) % 4; break;          /* base.py:416-417 */
        case A_FORWARD:                                                /* base.py:420-436 */
            if (cell && can_overlap(obj)) {
                if (!sp->allow_agent_overlap && agent_present(agent_state, A, fx, fy)) break;
                s[AS_X] = fx; s[AS_Y] = fy;
                if (cell[0] == T_GOAL) on_success(sp, agent_state, i, step_count, rewards);
                if (cell[0] == T_LAVA) on_failure(sp, agent_state, i);
            }
            break;
        case A_PICKUP:                                                 /* base.py:439-446 */
            if (cell && can_pickup(cell) && s[AS_CARRY] == T_EMPTY) {
                memcpy(s + AS_CARRY, cell, 3 * sizeof(int64_t));
                memcpy(cell, EMPTY_ENCODING, 3 * sizeof(int64_t));    /* grid.py:95-98 */
            }
            break;
        case A_DROP:                                                   /* base.py:449-459 */
            if (cell && s[AS_CARRY] != T_EMPTY && cell[0] == T_EMPTY
                && !agent_present(agent_state, A, fx, fy)) {
                memcpy(cell, s + AS_CARRY, 3 * sizeof(int64_t));
                memcpy(s + AS_CARRY, EMPTY_ENCODING, 3 * sizeof(int64_t)); /* agent.py:337-346 */
            }
            break;
        case A_TOGGLE:                                                 /* base.py:462-467 */
            if (!cell) break;
            if (cell[0] == T_DOOR) {                                   /* world_object.py:458-474 Door.toggle */
                if (obj[2] == S_LOCKED) {
                    if (s[AS_CARRY] == T_KEY && s[AS_CARRY + 1] == cell[1]) cell[2] = S_OPEN;
                } else {
                    cell[2] = (obj[2] == S_OPEN) ? S_CLOSED : S_OPEN;      /* grid.update: state := object */
                    if (stale) aux[4] = 0;
                }
            } else if (cell[0] == T_BOX) {                             /* world_object.py:599-605 Box.toggle */
                int64_t content[3];
                box_content(cell, content);                           /* env.grid.set(*pos, self.contains); None -> empty */
                memcpy(cell, content, 3 * sizeof(int64_t));
            }
            break;
        case A_DONE: break;                                            /* base.py:470-471 */
        default: return MGO_ERR_UNKNOWN_ACTION;                        /* base.py:473-474 */
        }
    }
    return 0;
}

/* base.py:303-346 step (+ envs/blockedunlockpickup.py:166-175 post-step hook).  Single env, reference
 * layout.  target: (3,) encoding of the BlockedUnlockPickup target box (ignored for KIND_EMPTY).
 * hook_order: the order in which the RedBlueDoors / LockedHallway hooks visit the agents = the insertion order of the
 * caller's `actions` dict (`for agent_id, action in actions.items()`, redbluedoors.py:176, locked_hallway.py:210): A agent
 * indices, or NULL for ascending index (a dict built in agent order).  Agents absent from the dict carry action -1 and are
 * skipped wherever they are listed. */
int mgo_step_ref(const MgoSpec *sp, int64_t *grid_state, int64_t *agent_state, uint64_t rng[4],
                 int64_t *step_count, const int8_t *actions, int64_t *target /* aux[MGO_AUX] */,
                 int64_t *obs, int64_t *direction, double *rewards, uint8_t *terminated, uint8_t *truncated,
                 int *order_out, const uint8_t *hook_order) {
    const int A = sp->num_agents;
    if (A < 1 || A > MGO_MAX_AGENTS) return -1;
    *step_count += 1;                                                  /* base.py:333 */
    int rc = handle_actions(sp, grid_state, agent_state, rng, *step_count, actions, rewards, order_out, target);
    if (rc) return rc;
    rc = mgo_gen_obs_ref(grid_state, agent_state, sp->width, sp->height, A, sp->view_size,   /* base.py:337 */
                         sp->see_through_walls, obs);
    if (rc) return rc;
    for (int a = 0; a < A; ++a) {
        direction[a] = agent_state[(size_t)a * AS_DIM + AS_DIR];       /* base.py:359, 372 */
        terminated[a] = (uint8_t)(agent_state[(size_t)a * AS_DIM + AS_TERMINATED] != 0);  /* base.py:338 */
    }
    *truncated = (uint8_t)(*step_count >= sp->max_steps);              /* base.py:339 */
    if (sp->env_kind == KIND_BLOCKEDUNLOCKPICKUP) {                    /* blockedunlockpickup.py:170-173 */
        for (int a = 0; a < A; ++a) {
            const int64_t *c = agent_state + (size_t)a * AS_DIM + AS_CARRY;
            /* `carrying == self.obj` is object identity; the target box is the only box in the layout,
             * so identity == (type, color) equality (asserted when layouts are imported). */
            if (c[0] == target[0] && c[1] == target[1]) {
                on_success(sp, agent_state, a, *step_count, rewards);
                /* on_success writes the terminations dict it is handed (base.py:498-501) */
                if (sp->success_any) { for (int b = 0; b < A; ++b) terminated[b] = 1; }
                else terminated[a] = 1;
            }
        }
    }
    if (sp->env_kind == KIND_RULES) {
        /* A user-defined env's step() hook in its declared form (include/mgx.h: MGX_KIND_RULES): the two shapes the reference's own
         * hooks have -- `if agent.state.carrying == self.obj: on_success` (blockedunlockpickup.py:170-173) and `if action == toggle
         * and fwd_obj == self.door [and self.door.is_open]: on_failure / on_success` (redbluedoors.py:176-185) -- as a table:
         * aux[0] = n, rule k = aux[1 + 5k ..] = { op, a, b, effect, cond }.  on_success / on_failure are the reference's
         * (base.py:478-532), restated above. */
        const int64_t *aux = target;
        const int H = sp->height, n = aux[0] < 3 ? (int)aux[0] : 3;
        for (int k = 0; k < n; ++k) {
            const int64_t *r = aux + 1 + 5 * k;
            for (int ko = 0; ko < A; ++ko) {
                int a = ko, hit = 0;
                if (r[0] == 1) {                                       /* CARRIES(type, colour): `for agent in self.agents` */
                    const int64_t *c = agent_state + (size_t)a * AS_DIM + AS_CARRY;
                    hit = c[0] == r[1] && c[1] == r[2];
                } else if (r[0] == 2) {                                /* TOGGLES_AT(x, y): `for agent_id, action in actions.items()` */
                    a = hook_order ? hook_order[ko] : ko;
                    if (a >= A || actions[a] != A_TOGGLE) continue;
                    const int64_t *s = agent_state + (size_t)a * AS_DIM;
                    int64_t fx = s[AS_X], fy = s[AS_Y];
                    if (s[AS_DIR] >= 0 && s[AS_DIR] < 4) { fx += DIR_TO_VEC[s[AS_DIR]][0]; fy += DIR_TO_VEC[s[AS_DIR]][1]; }
                    hit = fx == r[1] && fy == r[2];
                    if (hit && r[4] != 0) {                            /* `and self.door.is_open` / `and not ...` */
                        const int64_t *c = grid_state + ((size_t)r[1] * H + r[2]) * 3;
                        const int open = c[0] == T_DOOR && (c[2] & 3) == S_OPEN;
                        hit = c[0] == T_DOOR && (r[4] == 1 ? open : !open);
                    }
                }
                if (!hit) continue;
                if (r[3] == 1) {
                    on_success(sp, agent_state, a, *step_count, rewards);
                    if (sp->success_any) { for (int b = 0; b < A; ++b) terminated[b] = 1; } else terminated[a] = 1;
                } else if (r[3] == 2) {
                    on_failure(sp, agent_state, a);
                    if (sp->failure_any) { for (int b = 0; b < A; ++b) terminated[b] = 1; } else terminated[a] = 1;
                }
            }
        }
    }
    if (sp->env_kind == KIND_REDBLUEDOORS) {                           /* redbluedoors.py:170-187 */
        int64_t *aux = target;
        const int W = sp->width, H = sp->height;
        for (int k = 0; k < A; ++k) {                                  /* `for agent_id, action in actions.items()` */
            const int a = hook_order ? hook_order[k] : k;
            if (a >= A || actions[a] != A_TOGGLE) continue;            /* absent (-1) or another action */
            const int64_t *s = agent_state + (size_t)a * AS_DIM;
            int64_t fx = s[AS_X], fy = s[AS_Y];
            if (s[AS_DIR] >= 0 && s[AS_DIR] < 4) { fx += DIR_TO_VEC[s[AS_DIR]][0]; fy += DIR_TO_VEC[s[AS_DIR]][1]; }
            if (fx != aux[0] || fy != aux[1]) continue;                /* fwd_obj == self.blue_door */
            int64_t *blue = grid_state + ((size_t)aux[0] * H + aux[1]) * 3;
            const int64_t *red = grid_state + ((size_t)aux[2] * H + aux[3]) * 3;
            const int blue_open = !aux[4] && blue[2] == S_OPEN;        /* the OBJECT's state */
            if (!blue_open) continue;
            if (red[2] == S_OPEN) {
                on_success(sp, agent_state, a, *step_count, rewards);
                if (sp->success_any) { for (int b = 0; b < A; ++b) terminated[b] = 1; } else terminated[a] = 1;
            } else {
                on_failure(sp, agent_state, a);
                if (sp->failure_any) { for (int b = 0; b < A; ++b) terminated[b] = 1; } else terminated[a] = 1;
                aux[4] = 1;                                            /* blue_door.is_open = False, no grid.update */
            }
        }
        (void)W;
    }
    if (sp->env_kind == KIND_LOCKEDHALLWAY) {                          /* locked_hallway.py:203-227 */
        int64_t *aux = target;
        const int H = sp->height;
        /* aux[0] & 0x80: the geometric door format for more than 6 rooms (the doors sit mid-wall: add_door(rand_pos=False),
         * Room.set_door_pos: y = (top + bottom) // 2 = row (rs-1) + (rs-1) // 2, roomgrid.py:104-108, 116-118):
         * mask in aux[1] | aux[2] << 8, aux[3] = room_size, aux[4] = len(self.rooms) -- a dict keyed by colour, so
         * fewer than the number of doors when colours repeat (locked_hallway.py:166-176) */
        const int geo = ((int)aux[0] & 0x80) != 0;
        const int nd = (int)aux[0] & 0x7f, rs = (int)aux[3];
        const int n_rooms = geo ? (int)aux[4] : nd;
        int64_t mask = geo ? (aux[1] | (aux[2] << 8)) : aux[1];
        for (int ko = 0; ko < A; ++ko) {                               /* `for agent_id, action in actions.items()` */
            const int a = hook_order ? hook_order[ko] : ko;
            if (a >= A || actions[a] != A_TOGGLE) continue;
            const int64_t *s = agent_state + (size_t)a * AS_DIM;
            int64_t fx = s[AS_X], fy = s[AS_Y];
            if (s[AS_DIR] >= 0 && s[AS_DIR] < 4) { fx += DIR_TO_VEC[s[AS_DIR]][0]; fy += DIR_TO_VEC[s[AS_DIR]][1]; }
            if (fx < 0 || fx >= sp->width || fy < 0 || fy >= H) continue;
            const int64_t *c = grid_state + ((size_t)fx * H + fy) * 3;
            if (c[0] != T_DOOR || c[2] == S_LOCKED) continue;          /* isinstance(Door) and not is_locked */
            int d = -1;
            if (geo) {
                for (int k = 0; k < nd; ++k) {
                    const int64_t dx = (k & 1) ? 2 * (rs - 1) : rs - 1, dy = (int64_t)(k >> 1) * (rs - 1) + (rs - 1) / 2;   /* roomgrid.py:108: (top + bottom) // 2 */
                    if (dx == fx && dy == fy) { d = k; break; }
                }
            } else {
                for (int k = 0; k < nd; ++k) if (aux[2 + 2 * k] == fx && aux[3 + 2 * k] == fy) { d = k; break; }
            }
            if (d < 0 || (mask & (1 << d))) continue;                  /* not a room door / already in self.unlocked_doors */
            mask |= (1 << d);
            const double r = reward_value(*step_count, sp->max_steps);
            if (sp->joint_reward) { for (int b = 0; b < A; ++b) rewards[b] += r; }   /* `+=`, not `=` */
            else rewards[a] += r;
        }
        aux[1] = mask & 0xff;
        if (geo) aux[2] = (mask >> 8) & 0xff;
        int cnt = 0;
        for (int d = 0; d < nd; ++d) cnt += (int)((mask >> d) & 1);
        aux[15] = (cnt == n_rooms);                                    /* len(unlocked_doors) == len(rooms) */
        if (aux[15]) for (int b = 0; b < A; ++b) terminated[b] = 1;    /* the returned dict only, not agent state */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Batched wrappers over the product's packed layout (include/mgx.h):
 *   grid u8[B,H,W,3] ([y][x]); agents u8[B,A,8] = [color,dir,x,y,terminated,carry_type,carry_color,
 *   carry_state]; rng u64[B,4] = [state_lo,state_hi,inc_lo,inc_hi]; step_count i32[B]; actions i8[B,A];
 *   aux u8[B,16] (include/mgx.h); obs u8[B,A,v,v,3]; dir u8[B,A]; reward f64[B,A]; terminated u8[B,A]; truncated u8[B].
 * ---------------------------------------------------------------------------------------------- */
static void unpack_env(const MgoSpec *sp, const uint8_t *grid, const uint8_t *agents,
                       int64_t *grid_state, int64_t *agent_state) {
    const int W = sp->width, H = sp->height, A = sp->num_agents;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c)
                grid_state[((size_t)x * H + y) * 3 + c] = grid[((size_t)y * W + x) * 3 + c];
    for (int a = 0; a < A; ++a) {
        const uint8_t *p = agents + (size_t)a * 8;
        int64_t *s = agent_state + (size_t)a * AS_DIM;
        s[AS_TYPE] = T_AGENT; s[AS_COLOR] = p[0]; s[AS_DIR] = p[1]; s[AS_X] = p[2]; s[AS_Y] = p[3];
        s[AS_TERMINATED] = p[4]; s[AS_CARRY] = p[5]; s[AS_CARRY + 1] = p[6]; s[AS_CARRY + 2] = p[7];
    }
}

static void pack_env(const MgoSpec *sp, const int64_t *grid_state, const int64_t *agent_state,
                     uint8_t *grid, uint8_t *agents) {
    const int W = sp->width, H = sp->height, A = sp->num_agents;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c)
                grid[((size_t)y * W + x) * 3 + c] = (uint8_t)grid_state[((size_t)x * H + y) * 3 + c];
    for (int a = 0; a < A; ++a) {
        uint8_t *p = agents + (size_t)a * 8;
        const int64_t *s = agent_state + (size_t)a * AS_DIM;
        p[0] = (uint8_t)s[AS_COLOR]; p[1] = (uint8_t)s[AS_DIR]; p[2] = (uint8_t)s[AS_X]; p[3] = (uint8_t)s[AS_Y];
        p[4] = (uint8_t)s[AS_TERMINATED];
        p[5] = (uint8_t)s[AS_CARRY]; p[6] = (uint8_t)s[AS_CARRY + 1]; p[7] = (uint8_t)s[AS_CARRY + 2];
    }
}

/* multigrid/wrappers.py:158-190 OneHotObsWrapper.one_hot: x (h,w,3) int -> out (h,w,sum(dim_sizes)) uint8.
 * Restated over a flat list of cells (the (h,w) loops only enumerate cells). */
int mgo_one_hot(const int64_t *x, int64_t n_cells, const int64_t *dim_sizes, uint8_t *out) {
    const int64_t D = dim_sizes[0] + dim_sizes[1] + dim_sizes[2];
    memset(out, 0, (size_t)(n_cells * D));
    int64_t dim_offset = 0;
    for (int d = 0; d < 3; ++d) {
        for (int64_t c = 0; c < n_cells; ++c) {
            int64_t k = dim_offset + x[c * 3 + d];
            out[c * D + k] = 1;
        }
        dim_offset += dim_sizes[d];
    }
    return 0;
}

/* multigrid/wrappers.py:48-58 FullyObsWrapper.observation: img = grid.encode() (grid.py:310-322: state.copy());
 * for agent in agents: img[agent.state.pos] = agent.encode() (agent.py:135-148).  Reference layout (W,H,3). */
int mgo_full_obs(const int64_t *grid_state, const int64_t *agent_state, int W, int H, int A, int64_t *img) {
    memcpy(img, grid_state, sizeof(int64_t) * (size_t)W * H * 3);
    for (size_t k = 0; k < (size_t)W * H; ++k) img[3 * k + 2] &= 3;    /* (Grid.state holds no box content) */
    for (int a = 0; a < A; ++a) {
        const int64_t *s = agent_state + (size_t)a * AS_DIM;
        int64_t *c = img + ((size_t)s[AS_X] * H + s[AS_Y]) * 3;
        c[0] = T_AGENT; c[1] = s[AS_COLOR]; c[2] = s[AS_DIR];
    }
    (void)W;
    return 0;
}

int mgo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* returns 0 on success; on an unknown action returns MGO_ERR_UNKNOWN_ACTION and
 * stores the lowest offending env index in *err_env. */
int mgo_step_batch(const MgoSpec *sp, int64_t B, uint8_t *grid, uint8_t *agents, uint64_t *rng,
                   int32_t *step_count, const int8_t *actions, const uint8_t *target,
                   uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                   int64_t *err_env, int nthreads, const uint8_t *hook_order /* u8[B,A] or NULL */) {
    const int W = sp->width, H = sp->height, A = sp->num_agents, v = sp->view_size;
    const size_t gsz = (size_t)W * H * 3, osz = (size_t)A * v * v * 3;
    int64_t bad = -1;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        int64_t *gs = (int64_t *)malloc(sizeof(int64_t) * gsz);
        int64_t *as = (int64_t *)malloc(sizeof(int64_t) * (size_t)A * AS_DIM);
        int64_t *ob = (int64_t *)malloc(sizeof(int64_t) * osz);
        int64_t dirs[MGO_MAX_AGENTS], tgt[MGO_AUX];
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            unpack_env(sp, grid + b * gsz, agents + (size_t)b * A * 8, gs, as);
            int64_t sc = step_count[b];
            for (int k = 0; k < MGO_AUX; ++k) tgt[k] = target ? target[b * MGO_AUX + k] : 0;
            int rc = mgo_step_ref(sp, gs, as, rng + b * 4, &sc, actions + (size_t)b * A, tgt, ob, dirs,
                                  reward + (size_t)b * A, terminated + (size_t)b * A, truncated + b, NULL,
                                  hook_order ? hook_order + (size_t)b * A : NULL);
            step_count[b] = (int32_t)sc;
            pack_env(sp, gs, as, grid + b * gsz, agents + (size_t)b * A * 8);
            if (sp->env_kind == KIND_LOCKEDHALLWAY && target) {
                ((uint8_t *)target)[b * MGO_AUX + 1] = (uint8_t)tgt[1];
                ((uint8_t *)target)[b * MGO_AUX + 2] = (uint8_t)tgt[2];
                ((uint8_t *)target)[b * MGO_AUX + 15] = (uint8_t)tgt[15];
            }
            if (sp->env_kind == KIND_REDBLUEDOORS && target) ((uint8_t *)target)[b * MGO_AUX + 4] = (uint8_t)tgt[4];
            if (rc) {
#pragma omp critical
                { if (bad < 0 || b < bad) bad = b; }
                continue;
            }
            for (size_t k = 0; k < osz; ++k) obs[b * osz + k] = (uint8_t)ob[k];
            for (int a = 0; a < A; ++a) dir[(size_t)b * A + a] = (uint8_t)dirs[a];
        }
        free(gs); free(as); free(ob);
    }
    if (err_env) *err_env = bad;
    return bad >= 0 ? MGO_ERR_UNKNOWN_ACTION : 0;
}

int mgo_gen_obs_batch(const MgoSpec *sp, int64_t B, const uint8_t *grid, const uint8_t *agents,
                      uint8_t *obs, uint8_t *dir, int nthreads) {
    const int W = sp->width, H = sp->height, A = sp->num_agents, v = sp->view_size;
    const size_t gsz = (size_t)W * H * 3, osz = (size_t)A * v * v * 3;
    int fail = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        int64_t *gs = (int64_t *)malloc(sizeof(int64_t) * gsz);
        int64_t *as = (int64_t *)malloc(sizeof(int64_t) * (size_t)A * AS_DIM);
        int64_t *ob = (int64_t *)malloc(sizeof(int64_t) * osz);
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            unpack_env(sp, grid + b * gsz, agents + (size_t)b * A * 8, gs, as);
            if (mgo_gen_obs_ref(gs, as, W, H, A, v, sp->see_through_walls, ob)) { fail = 1; continue; }
            for (size_t k = 0; k < osz; ++k) obs[b * osz + k] = (uint8_t)ob[k];
            if (dir) for (int a = 0; a < A; ++a) dir[(size_t)b * A + a] = (uint8_t)as[(size_t)a * AS_DIM + AS_DIR];
        }
        free(gs); free(as); free(ob);
    }
    return fail ? -1 : 0;
}
