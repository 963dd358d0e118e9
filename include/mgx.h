/*
 * mgx.h -- C ABI of the MI355X-native batched MultiGrid step/observation engine (libmgx.so).
 *
 * The reference (ini/multigrid) is pure Python and has no FFI of its own: the seam this
 * library replaces is the pair of Python call sites
 *     multigrid/base.py:361-366   image = gen_obs_grid_encoding(grid.state, agent_states, view_size, see_through_walls)
 *     multigrid/base.py:333-340   step_count += 1; rewards = handle_actions(actions); gen_obs(); terminations; truncated
 * batched over B independent environments.  Every entry point takes plain device pointers and sizes, returns 0 or
 * a negative MGX_ERR_* code, never throws, never allocates, never frees and never retains a pointer.  All work is
 * enqueued on the HIP stream passed by the caller (a hipStream_t cast to void*; NULL = the default stream); the
 * functions do not synchronise.
 *
 * Tensor layouts (all contiguous, device memory):
 *   grid        u16[B, H, W]      PACKED cells (MgxCell): cell (x, y) of env b at (b*H + y)*W + x -- the [y][x] transpose of
 *                                 the reference's Grid.state (W,H,3) (multigrid/core/grid.py:54), each (type, color, state)
 *                                 triple in 16 bits:
 *                                     [3:0] type   [10:8] color   [13:12] state   [15] opaque
 *                                     [6:4] + {[7], [11], [14]}: what a BOX holds (ABI 9, below); zero on every other cell
 *                                 opaque = 1 for a cell one cannot see through (multigrid/utils/obs.py:46-63: a wall, or a door
 *                                 that is not open); it is part of the format (the kernels keep it up to date on every cell
 *                                 they write).  Two bytes per cell instead of the reference's three int64: a third less grid
 *                                 traffic and aligned 16-bit cell accesses on the device.  mgx_pack_grid / mgx_unpack_grid
 *                                 convert from / to u8[B,H,W,3] (type, color, state) bytes on the device; values the 16 bits
 *                                 cannot hold (type > 15, color > 7, state > 3 -- the reference has types 0-10, colors 0-5,
 *                                 states 0-2 and directions 0-3) are reported by mgx_pack_grid.
 *                                 BOX CONTENTS (ABI 9).  A box may hold an object (multigrid/core/world_object.py:574-605:
 *                                 Box(color, contains); Box.toggle replaces the box by its content).  Neither Grid.state nor an
 *                                 observation shows it, so it rides in bits the triple leaves free: in the (type, color, state)
 *                                 BYTES (mgx_pack_grid* input, mgx_unpack_grid output, an agent row's carried cell) the state
 *                                 byte of a box is  state | kind << 2 | content colour << 5,  kind = 0 nothing, 1 key, 2 ball,
 *                                 3 goal, 4 floor, 5 lava, 6 wall, 7 door (closed and unlocked, as Door(color) constructs it);
 *                                 in the packed cell: kind in [6:4], colour in {[7], [11], [14]}.  The kernels carry it through
 *                                 pickup / drop and put the content on the grid when the box is toggled; observations and
 *                                 mgx_full_obs show the box as (box, color, 0) whatever it holds.  A box in a box and doors in
 *                                 other states are not representable (the host refuses them: multigrid_amd.world.Box).
 *                                 PRECONDITION: the outer ring of cells (x = 0, x = W-1, y = 0, y = H-1) of every env is the
 *                                 reference's WALL = (wall, grey, 0) (multigrid/utils/obs.py:14).  Every env of the reference
 *                                 starts from Grid.wall_rect(0, 0, W, H) (multigrid/core/grid.py:183-218; envs/empty.py:158,
 *                                 core/roomgrid.py:203-218) and no action can change a wall, so it holds for every reachable
 *                                 state.  The kernels use it twice: a front cell outside the grid never has to be tested
 *                                 (the agent cannot stand on the ring), and a view cell outside the grid -- which
 *                                 obs.py:199-202 shows as WALL -- is read from the ring cell next to it (the gather clamps its
 *                                 coordinates instead of carrying an in-bounds mask per view).  The host side checks it on
 *                                 import (multigrid_amd/layouts.py: check_walled).
 *   agents      u8 [B, A, 8]      packed AgentState row (multigrid/core/agent.py:222-232, 72 B -> 8 B):
 *                                 [0]=color [1]=dir [2]=x [3]=y [4]=terminated [5]=carry.type [6]=carry.color
 *                                 [7]=carry.state (a carried box: | its content << 2, see "grid") ; "carrying nothing" = the
 *                                 empty cell (1,0,0) (agent.py:337-346).
 *   rng         u64[B, 4]         per-env numpy PCG64 state of `env.np_random`: [state_lo, state_hi, inc_lo, inc_hi].
 *                                 Advanced by A draws per step when A > 1 (multigrid/base.py:396-399).
 *   step_count  i32[B]            multigrid/base.py:292, 333
 *   actions     i8 [B, A]         multigrid/core/actions.py:5-15 (0..6); -1 = agent absent from the actions dict
 *                                 (multigrid/base.py:403-404); any other value = MGX_ERR_UNKNOWN_ACTION (base.py:473).
 *   aux         u8 [B, 16]        the env subclass' own attributes that its step() hook uses (in/out; may be NULL for
 *                                 MGX_KIND_EMPTY):
 *                                   BlockedUnlockPickup  [0..2] = encoding of the target box `self.obj`
 *                                                        (multigrid/envs/blockedunlockpickup.py:147, 172)
 *                                   RedBlueDoors         [0,1] = blue door (x,y), [2,3] = red door (x,y)
 *                                                        (multigrid/envs/redbluedoors.py:158-168); [4] = 1 while the blue
 *                                                        Door OBJECT is closed although Grid.state (= `grid`) still says
 *                                                        open: the hook closes it without grid.update
 *                                                        (redbluedoors.py:185).  The rules act on the object, obs on
 *                                                        `grid`; updated by the step
 *                                   LockedHallway        [0] = number of doors, [1] = bit k set once door k is in
 *                                                        `self.unlocked_doors` (updated by the step), [2+2k, 3+2k] =
 *                                                        door k (x,y), [15] = 1 when the last step reported every agent
 *                                                        terminated (multigrid/envs/locked_hallway.py:203-227)
 *                                                        With more than 6 rooms: [0] = 0x80 | number of doors (<= 16),
 *                                                        [1], [2] = unlocked mask (16 bits), [3] = room_size, [4] =
 *                                                        len(self.rooms) (rooms are keyed by door colour: repeated
 *                                                        colours count once, and that count ends the episode); door k =
 *                                                        (row k/2, side k%2) sits mid-wall at x = side ? 2(rs-1) : rs-1,
 *                                                        y = row (rs-1) + (rs-1)/2 (add_door(..., rand_pos=False):
 *                                                        (top + bottom) // 2, multigrid/core/roomgrid.py:108)

 *   obs         u8 [B, A, v, v, 3] image[i][j][c] exactly as multigrid/utils/obs.py:65-102 returns it
 *   dir         u8 [B, A]         obs['direction'] (multigrid/base.py:359, 372)
 *   reward      f64[B, A]         multigrid/base.py:393, 503-507, 598-602 (bit-identical Python float arithmetic)
 *   terminated  u8 [B, A]         multigrid/base.py:338 (+ env hook)
 *   truncated   u8 [B]            multigrid/base.py:339-340
 *   err         i32[2]            err[0] += number of envs that met an unknown action; err[1] = min such env index.
 *                                 The caller initialises it to {0, INT32_MAX} and inspects it after synchronising.
 */
#ifndef MGX_H
#define MGX_H

#if !defined(__HIPCC_RTC__)      /* (hipRTC has the fixed-width integer types built in and no system headers) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_ABI_VERSION 11

enum {
    MGX_OK = 0,
    MGX_ERR_INVALID_ARGUMENT = -1, /* NULL pointer, bad spec (even / <3 view size: multigrid/core/agent.py:78-79) */
    MGX_ERR_UNKNOWN_ACTION = -2,   /* reported through `err`, mirrors ValueError at multigrid/base.py:473-474 */
    MGX_ERR_UNSUPPORTED = -3,      /* spec outside the compiled limits (MGX_MAX_AGENTS, MGX_MAX_VIEW, LDS budget) */
    MGX_ERR_LAUNCH = -4            /* hipLaunchKernel failed; see mgx_last_hip_error() */
};

enum { MGX_KIND_EMPTY = 0, MGX_KIND_BLOCKEDUNLOCKPICKUP = 1, MGX_KIND_REDBLUEDOORS = 2, MGX_KIND_LOCKEDHALLWAY = 3,
       MGX_KIND_RULES = 4 };     /* ABI 9: a declarative step() hook, see MgxHookRule */

/* (ABI 9) MGX_KIND_RULES: the step() post-hook of a USER-DEFINED env, declared instead of compiled in.  The reference's envs
 * end their episodes in a `step` override that runs after the base step: `if agent.state.carrying == self.obj:
 * self.on_success(...)` (multigrid/envs/blockedunlockpickup.py:166-175), `if action == toggle and fwd_obj == self.blue_door ...:
 * self.on_failure(...)` (multigrid/envs/redbluedoors.py:170-187).  Both shapes as a table in the env's `aux`:
 *     aux[0] = n, the number of rules (<= MGX_MAX_RULES); rule k = the five bytes aux[1 + 5k ..]: { op, a, b, effect, cond }
 *       op      MGX_RULE_CARRIES    an agent carries an object of type a and colour b   (`agent.state.carrying == self.obj`)
 *               MGX_RULE_TOGGLES_AT an agent's action this step was `toggle` and the cell in front of it is (x, y) = (a, b)
 *                                   (`fwd_obj == self.door` for an object that does not move)
 *       effect  MGX_EFFECT_SUCCESS  on_success(agent): terminated (mode 'any': every agent), reward (multigrid/base.py:478-507)
 *               MGX_EFFECT_FAILURE  on_failure(agent)                                        (multigrid/base.py:509-532)
 *       cond    TOGGLES_AT only: MGX_COND_ALWAYS, MGX_COND_DOOR_OPEN / MGX_COND_DOOR_SHUT = only while the cell at (x, y) holds a
 *               door that is open / not open AFTER the step (`self.door.is_open`)
 * Evaluated after the base step on the post-step state, rule by rule in table order; within a CARRIES rule the agents in index
 * order (`for agent in self.agents`), within a TOGGLES_AT rule in `hook_order` (`for agent_id, action in actions.items()`).  As with
 * the compiled hooks the observation of the step is rendered BEFORE the hook; `terminated` / `reward` reflect it.  BlockedUnlockPickup
 * is the one-rule table { CARRIES, box, colour, SUCCESS } (tests/test_rule_hooks.py holds it against the compiled kind). */
#define MGX_MAX_RULES 3
enum { MGX_RULE_CARRIES = 1, MGX_RULE_TOGGLES_AT = 2 };
enum { MGX_EFFECT_SUCCESS = 1, MGX_EFFECT_FAILURE = 2 };
enum { MGX_COND_ALWAYS = 0, MGX_COND_DOOR_OPEN = 1, MGX_COND_DOOR_SHUT = 2 };

#define MGX_AUX_BYTES 16

typedef uint16_t MgxCell;        /* packed grid cell, see "grid" above */
#define MGX_CELL_BYTES 2

/* (ABI 9) COMPACT cells: the grid as ONE byte per cell, for large grids -- MgxSpec.cell_bytes = 1.  A step streams an env's whole
 * grid for its agents' views, so on a 64x64 grid (BASELINE.json configs[4]) the grid is half of all the bytes the step moves and
 * its LDS tile decides how many envs a wavefront can own: at one byte per cell the traffic falls by a third and a wavefront
 * steps two such envs instead of one.  A (type, color, state) triple holds 4 + 3 + 2 bits, but only doors (state) and the agent
 * overlay (direction) use the third field, so type and state are coded JOINTLY:
 *     [3:0] tcode   0..10 = the reference's type index with state 0 (4 = an OPEN door, 10 = an agent facing right);
 *                   11 = closed door, 12 = locked door; 13, 14, 15 = agent facing down / left / up
 *     [6:4] color   [7] opaque (as in MgxCell: a wall, or a door that is not open)
 * WALL = (wall, grey, 0) is 0xD2.  Wherever this header says `MgxCell *grid` (grid, pool_grid) an engine whose spec has
 * cell_bytes = 1 takes MgxCell8[B,H,W] through the same parameter (cast).  Served in this format: mgx_gen_obs, mgx_step,
 * mgx_step_autoreset, mgx_step_ex (steps = 1, no one_hot, no generate), mgx_step_chains / mgx_sub_shards, mgx_reset_done,
 * mgx_full_obs, mgx_check_grid, mgx_launch_info and the conversions mgx_pack_grid8_env / mgx_unpack_grid8; every other entry
 * point returns MGX_ERR_UNSUPPORTED for such a spec (rollouts, one-hot output, device-side generation and persistent stepping
 * keep the 16-bit cells).  Same results bit for bit in either format (tests/test_compact_cells.py). */
typedef uint8_t MgxCell8;

/* (ABI 9) BYTE grids: MgxSpec.cell_bytes = 3 -- the grid tensors (grid, pool_grid) ARE the reference's triples, u8[B,H,W,3]
 * (type, color, state | a box's content << 2), the tensor BASELINE.json's north star names.  The step / gen_obs kernels pack them
 * into their 16-bit LDS tile as the bytes arrive and write changed cells back as three bytes, so a caller that holds its state in
 * the reference's form pays the conversion inside the step's own launch (C4, 65536 envs: 23.6 us against 18.4 on packed cells)
 * instead of a pack and an unpack launch around it (55..62 us).  Served: mgx_gen_obs, mgx_step, mgx_step_autoreset, mgx_step_ex (steps = 1, no one_hot, no generate),
 * mgx_step_chains / mgx_sub_shards, mgx_reset_done, mgx_full_obs, mgx_launch_info; MgxStepArgs.grid_bad receives what
 * mgx_pack_grid_env would have counted.  Everything else: MGX_ERR_UNSUPPORTED.  Same results bit for bit. */

#define MGX_MAX_AGENTS 32
#define MGX_MAX_VIEW 15
#define MGX_AGENT_STRIDE 8

/* Constructor-level configuration of one env class: multigrid/base.py:85-103 plus the env-specific step hook. */
typedef struct MgxSpec {
    int32_t width;               /* grid width  W  (multigrid/base.py:152-155) */
    int32_t height;              /* grid height H */
    int32_t num_agents;          /* A */
    int32_t view_size;           /* v, odd, 3..MGX_MAX_VIEW (multigrid/core/agent.py:78-80) */
    int32_t max_steps;           /* multigrid/base.py:91 */
    int32_t see_through_walls;   /* multigrid/base.py:92 (agents[0]'s value is used for all: base.py:364-365) */
    int32_t allow_agent_overlap; /* multigrid/base.py:95 */
    int32_t joint_reward;        /* multigrid/base.py:96 */
    int32_t success_any;         /* success_termination_mode == 'any' (multigrid/base.py:97) */
    int32_t failure_any;         /* failure_termination_mode == 'any' (multigrid/base.py:98) */
    int32_t env_kind;            /* MGX_KIND_*: which subclass step() hook runs after the base step */
    int32_t cell_bytes;          /* ABI 9: the grid's cell format: 0 or 2 = MgxCell (16 bits), 1 = MgxCell8 (compact, see above),
                                    3 = the reference's byte triples u8[B,H,W,3] (see above) */
} MgxSpec;

/* Launch geometry chosen for (spec, batch); for diagnostics, benchmarks and DESIGN.md tables. */
typedef struct MgxLaunchInfo {
    int32_t envs_per_wavefront;
    int32_t envs_per_workgroup;
    int32_t threads_per_workgroup;
    int32_t workgroups;
    int32_t lds_bytes;
    int32_t slots_per_group;     /* view slots a wavefront gathers / stages as one block: 16, or 4 / 8 in the latency regime */
    int32_t fixed_shape;         /* ABI 7: > 0 = the plain step of this (spec, batch) runs a shape-specialised instantiation of the
                                    latency family (W, H, A and the envs per wavefront are compile-time constants: the shapes
                                    BASELINE.json names); 0 = the generic kernel.  Same results either way */
} MgxLaunchInfo;

int mgx_abi_version(void);
const char *mgx_error_string(int code);
int mgx_last_hip_error(void);

/* Replaces gen_obs_grid_encoding (multigrid/utils/obs.py:65-102) as called from MultiGridEnv.gen_obs
 * (multigrid/base.py:361-366), for B envs.  `dir` may be NULL. */
int mgx_gen_obs(const MgxSpec *spec, int64_t batch, const MgxCell *grid, const uint8_t *agents,
                uint8_t *obs, uint8_t *dir, void *stream);

/* Replaces MultiGridEnv.step (multigrid/base.py:303-346: step_count += 1, handle_actions 378-476, gen_obs
 * 348-376, terminations, truncations) and the BlockedUnlockPickupEnv.step post-hook
 * (multigrid/envs/blockedunlockpickup.py:166-175, redbluedoors.py:170-187, locked_hallway.py:203-227), for B envs, in one
 * fused kernel launch.
 * grid / agents / rng / step_count are updated in place. */
int mgx_step(const MgxSpec *spec, int64_t batch, MgxCell *grid, uint8_t *agents, uint64_t *rng,
             int32_t *step_count, const int8_t *actions, uint8_t *aux,
             uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
             int32_t *err, void *stream);

/* `steps` consecutive mgx_step calls in ONE launch, bit-identical to calling mgx_step `steps` times: every wavefront
 * keeps its envs' grid tile, agent rows, PCG64 state and step counts in LDS between steps, so the per-step HBM
 * traffic is just the actions in and the outputs out, and there is no launch or state round trip per step.
 * For rollouts whose actions do not depend on the observations (scripted / random policies, the benchmark of
 * BASELINE.json); a policy in the loop needs mgx_step.
 *   actions i8[steps,B,A]   obs u8[steps,B,A,v,v,3]   dir u8[steps,B,A]   reward f64[steps,B,A]
 *   terminated u8[steps,B,A]   truncated u8[steps,B]        (state tensors as in mgx_step, updated once at the end) */
int mgx_rollout(const MgxSpec *spec, int64_t batch, int32_t steps, MgxCell *grid, uint8_t *agents, uint64_t *rng,
                int32_t *step_count, const int8_t *actions, uint8_t *aux,
                uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                int32_t *err, void *stream);

/* Grid format conversion on the device (the reference's Grid.state holds (type, color, state) triples):
 *   mgx_pack_grid    cells3 u8[n_cells, 3] -> packed u16[n_cells]; bad[0] (i32, device, may be NULL; the caller zeroes it)
 *                    += the number of cells with a value the packed format cannot hold (they are stored truncated)
 *   mgx_unpack_grid  packed u16[n_cells] -> cells3 u8[n_cells, 3]
 * n_cells = B*H*W for a whole grid tensor (both layouts are [b][y][x]). */
int mgx_pack_grid(const uint8_t *cells3, int64_t n_cells, MgxCell *packed, int32_t *bad, void *stream);
int mgx_unpack_grid(const MgxCell *packed, int64_t n_cells, uint8_t *cells3, void *stream);
/* (ABI 8) mgx_pack_grid for whole env grids cells3 u8[B,H,W,3]: bad is i32[2] (may be NULL; the caller zeroes it) --
 *   bad[0] += cells with a value the packed format cannot hold, bad[1] += cells of an env's outer ring that are not the
 *   reference's WALL = (wall, grey, 0) (multigrid/utils/obs.py:14; the PRECONDITION under "grid" above).
 * mgx_check_grid: the same preconditions for state that is already on the device in the packed format, opt-in (one cheap
 * kernel; nothing else in this library re-checks what it is handed):
 *   bad i32[4], the caller initialises it to {0, 0, 0, INT32_MAX}:  bad[0] += cells that are not a valid packed cell (reserved
 *   bits, type > 10, color > 5, state > 2, opaque bit inconsistent with (type, state): multigrid/utils/obs.py:46-63),
 *   bad[1] += outer-ring cells that are not WALL, bad[2] += agent rows outside the walls / with direction > 3 / with a carried
 *   cell the format cannot hold (agents may be NULL: not checked), bad[3] = min index of an env with a violation. */
int mgx_pack_grid_env(const uint8_t *cells3, int64_t batch, int32_t height, int32_t width, MgxCell *packed, int32_t *bad,
                      void *stream);
int mgx_check_grid(const MgxSpec *spec, int64_t batch, const MgxCell *grid, const uint8_t *agents, int32_t *bad, void *stream);
/* (ABI 9) the same conversions for COMPACT cells (MgxCell8, spec->cell_bytes = 1): bad[0] also counts what only the compact
 * format cannot hold (a state on anything but a door / an agent overlay). */
int mgx_pack_grid8_env(const uint8_t *cells3, int64_t batch, int32_t height, int32_t width, MgxCell8 *packed, int32_t *bad,
                       void *stream);
int mgx_unpack_grid8(const MgxCell8 *packed, int64_t n_cells, uint8_t *cells3, void *stream);

/* Geometry of the plain step's launch (mgx_step / mgx_step_autoreset; mgx_gen_obs bundles as many envs per wavefront or fewer). */
int mgx_launch_info(const MgxSpec *spec, int64_t batch, MgxLaunchInfo *out);

/* ABI 11: geometry of the launches that keep the envs' state in LDS between steps -- mgx_rollout* (persistent = 0) and
 * mgx_step_persistent (persistent = 1).
 * resident_shape > 0: one of the library's RESIDENT instantiations -- 64 view slots per wavefront, `slices` groups of `envs_per_slice`
 * envs stepped one after the other by the same wavefront.  Empty-16x16 x 4 agents from 16385 envs up (the value names the shape):
 *   7  one slice of 16 envs, 13216 B of LDS: 12 wavefronts per CU, 49152 envs resident at once -- mgx_rollout* up to that batch and
 *      beyond 65536 (in rounds), mgx_step_persistent up to 32768 envs (its residency check counts 8 wavefronts per CU)
 *   9  one slice in 9872 B and <= 128 VGPRs (the tile's rows share the grid's wall ring): 16 wavefronts per CU -- mgx_rollout* of 49153
 *      to 65536 envs: BASELINE.json configs[3] is ONE resident round of 4096 wavefronts
 *   8  two slices: 2048 wavefronts of 2 x 16 envs, 8 per CU -- mgx_step_persistent of 32769 to 65536 envs (a persistent launch must
 *      leave registers for the kernels that feed it; it owns every CU's LDS: what runs beside it must do without)
 * 0: the ordinary rollout kernels (32 view slots, one slice).  Same results either way, bit for bit. */
typedef struct MgxRolloutInfo {
    int32_t envs_per_slice;
    int32_t slices;              /* per wavefront; envs per wavefront = envs_per_slice * slices */
    int32_t wavefronts;          /* that own envs (mgx_step_persistent: the flags of MgxPersistent.done) */
    int32_t threads_per_workgroup;
    int32_t workgroups;
    int32_t lds_bytes;           /* per workgroup */
    int32_t resident_shape;
} MgxRolloutInfo;
int mgx_rollout_info(const MgxSpec *spec, int64_t batch, int32_t persistent, MgxRolloutInfo *out);

/* Replaces OneHotObsWrapper.one_hot (multigrid/wrappers.py:158-190; the wrapper RLlib registration applies to every
 * env, multigrid/rllib/__init__.py:110-111) for a flat array of cells:
 *   cells u8[n_cells, 3]  ->  out u8[n_cells, D],  D = dim_sizes[0] + dim_sizes[1] + dim_sizes[2]  (<= 32),
 *   out[c, off_d + cells[c, d]] = 1, off = (0, dim_sizes[0], dim_sizes[0] + dim_sizes[1]); everything else 0.
 * The reference uses dim_sizes = (11, 6, 4) (wrappers.py:139-140).  `out` must be 16-byte aligned. */
int mgx_one_hot(const uint8_t *cells, int64_t n_cells, const int32_t *dim_sizes, uint8_t *out, void *stream);

/* Replaces FullyObsWrapper.observation (multigrid/wrappers.py:48-58): out u8[B, W, H, 3] = Grid.state ([x][y], the
 * reference's own orientation) with every agent's (10, color, dir) written at its position in index order,
 * terminated agents included. */
int mgx_full_obs(const MgxSpec *spec, int64_t batch, const MgxCell *grid, const uint8_t *agents, uint8_t *out,
                 void *stream);

/* Vector-env auto-reset (build-defined: the reference has no batching; its caller tests is_done(), base.py:534-539,
 * and calls reset(), base.py:250-301).  Every env b whose episode is over -- all agents terminated or
 * step_count >= max_steps -- is re-initialised from a pool of K pre-generated layouts
 * (pool_grid u16[K,H,W] packed cells, pool_agents u8[K,A,8], pool_aux u8[K,16] or NULL):
 *   layout = (first_env + b + episode[b] * 7919) mod K;  step_count[b] = 0;  episode[b] += 1;  was_reset[b] = 1.
 * The env's PCG64 stream is left running, as an unseeded reset() does for Empty envs.  `was_reset` may be NULL. */
int mgx_reset_done(const MgxSpec *spec, int64_t batch, int64_t first_env, int32_t pool_size, const MgxCell *pool_grid,
                   const uint8_t *pool_agents, const uint8_t *pool_aux, MgxCell *grid, uint8_t *agents,
                   int32_t *step_count, uint8_t *aux, int32_t *episode, uint8_t *was_reset, void *stream);

/* The two calls below fuse mgx_reset_done into the step: an env whose episode ended with the PREVIOUS step (all agents
 * terminated or step_count >= max_steps, base.py:534-539) is first re-initialised from the layout pool exactly as
 * mgx_reset_done defines, then the step is applied to the fresh state -- bit-identical to mgx_reset_done followed by
 * mgx_step, without the second launch.  `was_reset` (u8[B], or u8[steps,B] for the rollout; may be NULL) reports which envs
 * restarted before each step. */
typedef struct MgxAutoReset {
    int64_t first_env;            /* global index of env 0 of this shard */
    int32_t pool_size;            /* K >= 1 */
    const MgxCell *pool_grid;     /* u16[K,H,W] packed cells */
    const uint8_t *pool_agents;   /* u8[K,A,8] */
    const uint8_t *pool_aux;      /* u8[K,16]; NULL for MGX_KIND_EMPTY */
    int32_t *episode;             /* i32[B], in/out */
    uint8_t *was_reset;           /* out, may be NULL */
} MgxAutoReset;

int mgx_step_autoreset(const MgxSpec *spec, int64_t batch, const MgxAutoReset *ar, MgxCell *grid, uint8_t *agents,
                       uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                       uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                       int32_t *err, void *stream);

int mgx_rollout_autoreset(const MgxSpec *spec, int64_t batch, int32_t steps, const MgxAutoReset *ar, MgxCell *grid,
                          uint8_t *agents, uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                          uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                          int32_t *err, void *stream);

/* On-device episode starts.  Every env whose episode is over (the mgx_reset_done test) is
 * re-initialised by running the reference's own generation ON THE DEVICE, one lane per env, draw for draw:
 * Agent.reset + _gen_grid of multigrid/base.py:250-301 with place_obj / place_agent (base.py:604-697), RoomGrid's
 * place_in_room / reject_next_to / place_agent (multigrid/core/roomgrid.py:45-50, 238-259, 376-404) and numpy's
 * Generator(PCG64).integers (Lemire's method over PCG64's buffered 32-bit outputs).  Two generators per env, as in the
 * reference (its placement draws come from a construction-time generator, only Room.set_door_pos uses env.np_random,
 * roomgrid.py:104-106):
 *   gen_state u64[B,6]  [0..3] = the placement generator's PCG64 state (state_lo, state_hi, inc_lo, inc_hi),
 *                       [4] = its 32-bit buffer (has_uint32 << 32 | uinteger), [5] = the 32-bit buffer of env.np_random,
 *                       whose PCG64 words are the `rng` tensor of mgx_step.  In/out.
 *   blank     u16[H,W]  (packed cells) the grid before any object or agent is placed: the rooms' walls (RoomGrid, roomgrid.py:203-218),
 *                       or border walls + goal at (W-2, H-2) (EmptyEnv, multigrid/envs/empty.py:156-162).  It is what gets
 *                       copied into a restarted env's grid; the placement tests use the same layout in closed form, so it
 *                       must be exactly that (multigrid_amd.layouts.roomgrid_blank / empty_blank produce it)
 * Kinds:  MGX_GEN_EMPTY_FIXED          EmptyEnv with agent_start_pos / agent_start_dir (empty.py:164-167): no draws
 *         MGX_GEN_EMPTY_RANDOM         EmptyEnv with agent_start_pos=None: place_agent over the whole grid (empty.py:168-169)
 *         MGX_GEN_BLOCKEDUNLOCKPICKUP  multigrid/envs/blockedunlockpickup.py:142-164 (room_size; also writes aux[0..2])
 *         MGX_GEN_LOCKEDHALLWAY        multigrid/envs/locked_hallway.py:152-201 (room_size, max_hallway_keys, max_keys_per_room; the grid
 *                                      is 3 columns of rooms x num_rooms / 2 rows, num_rooms <= 16; numpy's Generator.shuffle for
 *                                      _rand_perm; writes the whole aux; `blank` = multigrid_amd.layouts.lockedhallway_blank)
 *         MGX_GEN_PLAYGROUND           multigrid/envs/playground.py:122-137 (room_size; rooms x rooms from the grid size, at most 16
 *                                      rooms; connect_all's doors are placed with env.np_random, roomgrid.py:104-124; `blank` =
 *                                      multigrid_amd.layouts.roomgrid_blank; spec->env_kind = MGX_KIND_EMPTY: no hook)
 *         MGX_GEN_REDBLUEDOORS         multigrid/envs/redbluedoors.py:142-168 (grid 2*size x size: agents placed in the middle
 *                                      room, then the red and the blue door rows drawn; writes aux[0..4]; `blank` = the outer
 *                                      walls + the room's walls, multigrid_amd.layouts.redbluedoors_blank)
 * A spec whose rooms have no cell to spare for what the generator places in them (e.g. a BlockedUnlockPickup room_size of 4 with two
 * agents, a Playground of 3x3-cell rooms for its 12 objects) is refused with MGX_ERR_UNSUPPORTED: the reference's place_obj samples
 * positions without a bound (base.py:604-669) and would never return -- here that lane would hang the device.
 * Given generators in the same state the result is byte-identical to the reference's reset() (pinned through
 * multigrid_amd/layouts.py and the reference's reset fixtures).  step_count := 0, episode += 1, was_reset (may be NULL). */
enum { MGX_GEN_EMPTY_FIXED = 0, MGX_GEN_EMPTY_RANDOM = 1, MGX_GEN_BLOCKEDUNLOCKPICKUP = 2, MGX_GEN_REDBLUEDOORS = 3,
       MGX_GEN_LOCKEDHALLWAY = 4, MGX_GEN_PLAYGROUND = 5 };

/* Staged generation (ABI 7, optional; used by mgx_step_generate / mgx_step_ex only).  A launch lasts as long as its slowest
 * wavefront, and the serial placement of one episode (~6500 cycles on one lane) in the tail of the step made the wavefront that
 * holds a finished env exactly that -- every step, at thousands of envs.  Most episodes of a random-action rollout end by
 * TRUNCATION, which is known in advance: two steps before an env's step_count reaches max_steps the step takes a snapshot of the
 * env's PCG64 state (the reference's handle_actions draws exactly A numbers per step, multigrid/base.py:396-399, so the state at
 * reset time is the snapshot advanced by A draws per remaining step), and in the NEXT launch a few extra wavefronts -- beside the
 * step's own, not behind them -- generate that env's next episode from it into the staging slot below.  When the episode then
 * ends by truncation its slot is adopted (a copy); an env that ends earlier (success / failure: its np_random is somewhere else)
 * finds no slot for its episode and is generated in the tail as before.  Same results bit for bit either way: the slot's content is
 * the same function of (gen_state, np_random at reset time) that the tail computes.  The slots are a cache, not state: any content
 * with tag -1 is valid (all of it may be dropped at any time).  All pointers NULL = no staging.
 *   grid    MgxCell[B,H,W]   agents u8[B,A,8]   aux u8[B,16] (may be NULL for MGX_KIND_EMPTY)
 *   words   u64[B,12]   [0..5] gen_state after the staged generation, [6..9] rng (env.np_random) after it,
 *                       [10..11] the snapshot: PCG64 state (lo, hi)
 *   tag     i32[B,4]    [0] the episode whose successor the slot holds (-1: none), [1] snapshot: episode, [2] snapshot: step_count,
 *                       [3] snapshot request: phase + 1 of the launch that took it, 0 = none / served
 *   phase   the caller adds 1 for every step it issues (any wrap-around is fine; a constant phase only disables the staging) */
typedef struct MgxGenStage {
    MgxCell *grid;
    uint8_t *agents;
    uint8_t *aux;
    uint64_t *words;
    int32_t *tag;
    int32_t phase;
    int32_t lead;                 /* ABI 8: the snapshot is taken `lead` steps before the truncation (0 = 2, the least the in-launch
                                     generators need; more when the generator runs beside the steps: see `external`) */
    int32_t external;             /* ABI 8: 1 = the step's launch carries NO generator wavefronts: the caller runs them as their own
                                     launches on a stream beside the steps' (mgx_stage_generate), a few steps ahead of the truncation --
                                     off the step's critical path altogether.  2 = the same, and the caller enqueues those launches on
                                     the steps' OWN stream, between two steps: no generator ever runs beside a step, so the step publishes
                                     its snapshot with plain stores (the kernel boundary orders them) instead of an agent-scope release --
                                     which writes the XCD's L2 back, ~1.5 us for the wavefront that takes a snapshot.  Do NOT pass 2 with
                                     generator launches on another stream. */
    int32_t candidates;           /* ABI 10: K > 0 = EVERY episode end an adoption (needs external == 2).  The layout stream that drives
                                     place_obj is the construction-time generator: it does not depend on WHEN the
                                     episode ends -- only the draws from env.np_random do, and the generators make at most one such draw
                                     (BlockedUnlockPickup: the door's row, roomgrid.py:104-106; the Empty / RedBlueDoors / LockedHallway
                                     generators none).  So the NEXT episode is generated while the current one runs, once per possible
                                     value of that draw: K = room_size - 2 candidates (<= 4) for BlockedUnlockPickup, 1 for the others
                                     (MGX_GEN_PLAYGROUND draws many times: not available).  mgx_stage_generate fills the candidates of every
                                     env whose slots are not its current episode's (tag[k] != episode; no snapshot, no request, `lead`
                                     only sets how often the caller launches it); a step that ENDS an episode -- truncation or success /
                                     failure alike -- makes the draw from np_random as it is then (registers) and adopts the matching
                                     candidate, requested as soon as the step's rules have run and stored in the tail; an env whose
                                     candidates are not there yet is generated in the tail.  Same results bit for bit.  The slots then are
                                       grid MgxCell[B,K,H,W]   agents u8[B,K,A,8]   aux u8[B,K,16]   words u64[B,K,6] (gen_state[0..4] after
                                       that candidate's generation)   tag i32[B,4]: [k] = the episode whose successor candidate k is
                                     (-1: none; any content with all tags -1 is valid) */
} MgxGenStage;

typedef struct MgxLayoutGen {
    int32_t kind;
    int32_t room_size;                    /* MGX_GEN_BLOCKEDUNLOCKPICKUP, MGX_GEN_LOCKEDHALLWAY, MGX_GEN_PLAYGROUND */
    int32_t start_x, start_y, start_dir;  /* MGX_GEN_EMPTY_FIXED */
    int32_t max_hallway_keys;             /* MGX_GEN_LOCKEDHALLWAY (locked_hallway.py:107) */
    int32_t max_keys_per_room;            /* MGX_GEN_LOCKEDHALLWAY (locked_hallway.py:108) */
    const MgxCell *blank;
    uint64_t *gen_state;
    MgxGenStage stage;                    /* ABI 7: staged generation of truncation resets (all NULL: off) */
} MgxLayoutGen;

int mgx_reset_generate(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, MgxCell *grid, uint8_t *agents,
                       uint64_t *rng, int32_t *step_count, uint8_t *aux, int32_t *episode, uint8_t *was_reset, void *stream);

/* (ABI 8) The staging slots' generator as a launch of its own: serves every pending snapshot request of `gen->stage` (one lane
 * per env; envs without a request cost one load).  Enqueue it on a stream BESIDE the one the steps run on, ordered behind the step
 * that took the snapshots (an event): with gen->stage.external = 1 and a lead of a few steps the next episodes of the envs about to
 * truncate are ready when their truncating step adopts them, and no step ever waits for a generator.  A slot that is not ready in
 * time is simply not adopted (that env is generated in the tail of its step, as without staging): same results either way.
 * rng = the env.np_random words of mgx_step (read: the stream increments), episode as in mgx_step_generate. */
int mgx_stage_generate(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, const uint64_t *rng, const int32_t *episode,
                       void *stream);

/* mgx_step followed by mgx_reset_generate, in ONE launch: the envs whose episode ends with this step (all agents terminated
 * or step_count >= max_steps after it) are regenerated in the tail of the step's own kernel; the outputs are the step's
 * (the terminal observation); the state tensors come out holding the next episode's start.  Bit-identical to the two calls.
 * `was_reset` (may be NULL) = the envs that were regenerated. */
int mgx_step_generate(const MgxSpec *spec, int64_t batch, const MgxLayoutGen *gen, MgxCell *grid, uint8_t *agents,
                      uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                      uint8_t *obs, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                      int32_t *err, int32_t *episode, uint8_t *was_reset, void *stream);

/* The step / gen_obs with the observation written ONE-HOT encoded: what OneHotObsWrapper.one_hot
 * (multigrid/wrappers.py:158-190, dim sizes (11, 6, 4)) makes of obs['image'], fused into the same launch -- the
 * wrapper RLlib registration applies to every env (multigrid/rllib/__init__.py:110-111).  Bit-identical to
 * mgx_gen_obs / mgx_step[_autoreset] followed by mgx_one_hot, without the 3-byte observation's round trip through HBM:
 *   obs_one_hot u8[B, A, v, v, 21]   obs_one_hot[b,a,i,j, c0] = obs_one_hot[.., 11 + c1] = obs_one_hot[.., 17 + c2] = 1
 *                                    for obs[b,a,i,j] = (c0, c1, c2); everything else 0.  16-byte aligned.
 * `ar` may be NULL (no auto-reset).  Everything else as in mgx_step / mgx_step_autoreset. */
int mgx_gen_obs_one_hot(const MgxSpec *spec, int64_t batch, const MgxCell *grid, const uint8_t *agents,
                        uint8_t *obs_one_hot, uint8_t *dir, void *stream);
int mgx_step_one_hot(const MgxSpec *spec, int64_t batch, const MgxAutoReset *ar, MgxCell *grid, uint8_t *agents,
                     uint64_t *rng, int32_t *step_count, const int8_t *actions, uint8_t *aux,
                     uint8_t *obs_one_hot, uint8_t *dir, double *reward, uint8_t *terminated, uint8_t *truncated,
                     int32_t *err, void *stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * The general form of the step (ABI 6).  Every mgx_step* / mgx_rollout* entry point above is this call with some of the
 * options set; options combine freely unless noted.  Replaces multigrid/base.py:303-346 (+ the env subclass' step hook,
 * + base.py:250-301 reset for finished envs when `auto_reset` / `generate` is given, + wrappers.py:158-190 when `one_hot`).
 *   steps        1 = one step; T > 1 = T consecutive steps in one launch (mgx_rollout: actions / hook_order / every output and
 *                was_reset carry a leading [T] axis, the state is written back once)
 *   one_hot      obs is u8[.., A, v, v, 21] (mgx_step_one_hot)
 *   auto_reset   finished envs restart from the layout pool BEFORE the step (mgx_step_autoreset), or NULL
 *   generate     the envs whose episode ends WITH the step are regenerated on the device right after it
 *                (mgx_step_generate; needs `episode`, optional `was_reset`; not together with `auto_reset`), or NULL.  With
 *                steps = T the call enqueues T launches of the step kernel over the [t] slices (generation writes the state in
 *                device memory, which the one-launch rollout keeps in LDS between its steps): same results as T calls; every
 *                slice of `obs` must then be 16-byte aligned (B * A * v * v * 3 a multiple of 16, else MGX_ERR_INVALID_ARGUMENT)
 *   hook_order   u8[B, A] or NULL: the order in which the env subclass' step hook visits the agents -- the reference's hooks
 *                iterate `actions.items()`, i.e. the insertion order of the caller's dict (multigrid/envs/redbluedoors.py:176,
 *                locked_hallway.py:210); row b lists agent indices in that order (a permutation of 0..A-1; agents absent from
 *                the dict have action -1 and are skipped wherever they are listed).  NULL = ascending index, which is what a
 *                dict built in agent order gives.  Only RedBlueDoors / LockedHallway read it (BlockedUnlockPickup's hook
 *                iterates self.agents, blockedunlockpickup.py:170).
 */
typedef struct MgxStepArgs {
    /* state, in/out */
    MgxCell *grid;
    uint8_t *agents;
    uint64_t *rng;               /* may be NULL when A == 1 */
    int32_t *step_count;
    uint8_t *aux;                /* may be NULL for MGX_KIND_EMPTY */
    /* inputs */
    const int8_t *actions;
    const uint8_t *hook_order;   /* may be NULL */
    /* outputs */
    uint8_t *obs;
    uint8_t *dir;                /* may be NULL */
    double *reward;
    uint8_t *terminated;
    uint8_t *truncated;
    int32_t *err;                /* may be NULL */
    /* options */
    int32_t steps;
    int32_t one_hot;
    const MgxAutoReset *auto_reset;
    const MgxLayoutGen *generate;
    int32_t *episode;            /* `generate`: i32[B], in/out */
    uint8_t *was_reset;          /* `generate`: u8[B] (or [T,B]), may be NULL */
    int32_t *grid_bad;           /* ABI 9, byte grids (MgxSpec.cell_bytes = 3) only, may be NULL: i32[2], the caller zeroes it --
                                    [0] += cell values the packed format cannot hold, [1] += outer-ring cells that are not WALL
                                    (the counters of mgx_pack_grid_env; nothing synchronises) */
} MgxStepArgs;

int mgx_step_ex(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, void *stream);

/* Sub-shard stepping.  A launch that fills the chip in one round of wavefronts first loads (no wave has data to work on), then
 * computes, then drains, and the next step's launch cannot start before the last wave has gone; envs are independent, so the
 * same step can be issued as `parts` launches over consecutive blocks of the batch on `parts` streams, and consecutive calls
 * then form `parts` independent CHAINS of launches whose bubbles are filled by the other chains' waves (C4: 18.6 -> 16.0-16.7 us per
 * step of 65536 envs as two chains).  Blocks are cut at multiples of 64 envs; per-env seeds, auto-reset layouts and generated episodes follow
 * the global env index (auto_reset->first_env + block offset), so the results are bit-identical to mgx_step_ex on the whole batch.
 *   streams      `parts` HIP streams (hipStream_t cast to void*), one per chain
 *   fork_event   a hipEvent_t (cast to void*) the caller recorded on the stream that produced `actions`, or NULL: every chain's
 *                stream waits for it before its launch.  The call does NOT join: the outputs of block k are complete when
 *                streams[k] reaches this point (the caller makes its consumer wait on the streams it needs).
 * steps must be 1.  mgx_sub_shards() suggests `parts` for (spec, batch, options of `args`) on the current device: 1 when the
 * launch is less than two wavefronts per SIMD of the chip (splitting only shortens short launches), else 2 (round 5: two chains
 * hold their gain on every box and graph length measured -- C4 18.6 -> 16.0-16.7 us, C5 61.5 -> 47 us --, four swing between
 * 15.3 and 18.6 with the host side of the graph replay: profiles/r5_chain_policy.txt).  `args` may be NULL (plain step). */
int mgx_step_chains(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, int32_t parts, void *const *streams,
                    void *fork_event);
int mgx_sub_shards(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, int32_t *parts);

/* ---------------------------------------------------------------------------------------------------------------------
 * Shape specialisation for ANY shape, at run time (ABI 8).  The library carries shape-specialised instantiations of the step
 * kernel for the shapes BASELINE.json names (MgxLaunchInfo.fixed_shape > 0); a launch in the latency regime is a lone wavefront's
 * instruction chain, where a compile-time (W, H, A, envs per wavefront) is worth 10-14 %.  Every other shape -- the reference
 * registers 17 env ids (multigrid/envs/__init__.py:38-52) and users define their own -- gets the same treatment by compiling
 * csrc/mgx_fused.h for its geometry at run time (hipRTC; multigrid_amd/jit.py does it and caches the code object) and registering
 * the result: mgx_shape_key says what to compile for (spec, batch), mgx_shape_register loads the code object -- which must define
 * `mgx_jit_step`, `mgx_jit_step_ar` (extern "C" kernels taking the library's kernel-argument struct) and the i32 global
 * `mgx_jit_kernel_args_bytes` -- for the CURRENT device; later launches of the plain step with exactly that geometry run it
 * (mgx_step / mgx_step_autoreset / mgx_step_ex without one_hot / generate / steps > 1).  Same results bit for bit: it is the same
 * source.  Registrations live until the process ends. */
#define MGX_SHAPE_RUNTIME_COMPILED 100     /* MgxLaunchInfo.fixed_shape of a registered runtime-compiled shape */
typedef struct MgxShapeKey {
    int32_t width, height, num_agents, envs_per_wavefront, hooks, view_size, dma, stream;   /* the MGX_JIT_SHAPE initialiser, in order */
    int32_t kernel_args_bytes;   /* sizeof the kernel-argument struct of THIS library (checked against the code object's) */
    int32_t built_in;            /* out: > 0 = the library has its own instantiation for this geometry (nothing to compile) */
    int32_t registered;          /* out: a runtime-compiled one is registered for it on the current device */
} MgxShapeKey;
int mgx_shape_key(const MgxSpec *spec, int64_t batch, MgxShapeKey *key);
int mgx_shape_register(const MgxShapeKey *key, const void *code_object, size_t bytes);

/* ---------------------------------------------------------------------------------------------------------------------
 * Persistent stepping (ABI 8): closed-loop stepping without a kernel boundary per step.  Replaces the loop a caller of the
 * reference runs around multigrid/base.py:303-346 -- `obs = env.step(policy(obs))`, e.g. multigrid/rllib/__init__.py:59-63 -- for
 * batches in the latency regime (a launch of the step is a lone wavefront's instruction chain per SIMD: BASELINE.json's
 * Empty-16x16 x 4096 envs, the 8192-env share of an 8-GPU node, BlockedUnlockPickup x 16384), where the dependent-launch
 * boundary (~1.5 us) and the reload of the env state (~0.6 us) are a third of a step.
 *
 * mgx_step_persistent enqueues ONE launch of the rollout kernel (mgx_rollout: every wavefront keeps its envs' grid tile, agent
 * rows, PCG64 state and step counts in LDS) that stays resident for up to `max_steps` steps and, per step t = 1, 2, ...:
 *   (1) waits until the actions of step t are there.  They are handed over as GRANULES, one aligned 8-byte word per 4 agents of
 *       an env -- action_granules u64[B, ceil(A/4)], granule q of env b: bits [31:0] = the action bytes of agents 4q..4q+3
 *       (i8, little-endian, -1 = absent, bytes of agents >= A ignored), bits [63:32] = the tag t.  The producer writes each
 *       granule with ONE agent-scope 8-byte store (mgx_persistent_post does that from an ordinary i8[B,A] action tensor): the
 *       data is the flag, a wavefront re-reads only its own granules, no fence on either side;
 *   (2) applies the step exactly as mgx_step does (same results bit for bit, tests/test_persistent.py);
 *   (3) writes obs / dir / reward / terminated / truncated (/ was_reset) of step t THROUGH to memory -- the same buffers every
 *       step -- and then sets done[w] = t for its wavefront w.  The outputs of step t are complete once done[w] >= t for every
 *       w < waves (mgx_persistent_waves); a kernel launched after that has been observed (mgx_persistent_wait) reads them.
 * The producer may write the granules of step t+1 only after it has seen the outputs of step t complete (one action buffer,
 * one set of output buffers: the hand-shake is the double buffer).  The env state tensors of `args` (grid, agents, rng,
 * step_count, aux) are NOT current while the launch runs; they are written back when it ends: after `max_steps` steps, or after
 * the step during which ctrl[0] became non-zero (stop request), or when a wavefront has waited `timeout_ms` for its granules
 * (ctrl[1] counts those: the producer went away).  Every wait in these kernels is bounded by `timeout_ms`.
 *   ctrl u32[8], zeroed by the caller except [3] = UINT32_MAX: [0] in: stop request; [1] out: wavefronts (or waiters) that timed
 *        out; [2] out: wavefronts that have left; [3] out: min over them of the steps they completed; [4] a waiter of the
 *        hand-shake kernels timed out (shared by their wavefronts in place of LDS, which a chip-filling launch leaves none of)
 *   done u32[waves], zeroed by the caller
 * Options of `args`: auto_reset (layout pool) yes; one_hot, generate, hook_order no (MGX_ERR_UNSUPPORTED); steps is ignored.
 * Every wavefront of the launch must be resident at once: MGX_ERR_UNSUPPORTED when (spec, batch) needs more workgroups than the
 * device holds (use mgx_step for such batches: they are throughput-bound, not boundary-bound).  The launch must run on a stream
 * of its own: whatever produces the granules runs beside it. */
#define MGX_PERSIST_CTRL_WORDS 8
typedef struct MgxPersistent {
    const uint64_t *action_granules;  /* u64[B, ceil(A/4)], written by the producer */
    uint32_t *done;                   /* u32[waves] */
    uint32_t *ctrl;                   /* u32[MGX_PERSIST_CTRL_WORDS] */
    int32_t max_steps;                /* >= 1 */
    int32_t timeout_ms;               /* 1 .. 30000 */
} MgxPersistent;

int mgx_persistent_waves(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, int32_t *waves);
int mgx_step_persistent(const MgxSpec *spec, int64_t batch, const MgxStepArgs *args, const MgxPersistent *p, void *stream);
/* actions i8[B,A] (as mgx_step takes them) -> the granules of step `step` (counted from 1); stream-ordered behind whatever
 * produced `actions` on `stream`. */
int mgx_persistent_post(const MgxSpec *spec, int64_t batch, const int8_t *actions, uint32_t step, uint64_t *action_granules,
                        void *stream);
/* Returns (in stream order) once done[w] >= step for every w < waves, or after timeout_ms (then ctrl[1] += 1). */
int mgx_persistent_wait(const uint32_t *done, int32_t waves, uint32_t step, uint32_t *ctrl, int32_t timeout_ms, void *stream);
/* A stand-in policy for benchmarks and tests: a few resident workgroups (256 threads per 2048 granules) that play a recorded
 * action sequence actions i8[T,B,A]
 * through the closed-loop hand-shake -- for t = 1..T: wait for step t-1's outputs to be complete, post the granules of step t --
 * i.e. the shortest producer there can be.  trace (u64[2T + 1], may be NULL): s_memrealtime ticks (100 MHz): [2(t-1)] = step t's
 * predecessor seen complete, [2(t-1) + 1] = step t's granules posted, [2T] = step T seen complete. */
int mgx_persistent_feed(const MgxSpec *spec, int64_t batch, const int8_t *actions, int32_t steps, const MgxPersistent *p,
                        int32_t waves, uint64_t *trace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */
