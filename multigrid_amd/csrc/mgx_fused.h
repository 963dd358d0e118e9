// mgx_fused.h -- the fused gfx950 (MI355X, CDNA4, wave64) kernel behind the C ABI of libmgx.so (include/mgx.h).
// Included by mgx_fused_inst.hip (one translation unit per view size: the instantiations compile in parallel) and by
// mgx_kernels.hip (the C ABI: argument checks, launch geometry, dispatch to the view size's translation unit).
//
// One fused kernel does a whole MultiGridEnv.step for the batch.  Every WAVEFRONT is autonomous: it owns Gw consecutive
// envs (<= 64 agent views, "slots": slots_per_wave) and a private LDS slice and runs all phases for them without a workgroup
// barrier:
//
//   P0   buffer_load_dwordx4 of the wave's (Gw,H,W) packed 16-bit grid cells, packed agent rows, actions, PCG64 words and
//        step counts; one s_waitcnt; LDS stores
//   P1a  lane = (env, agent): that agent's PCG64 draw by jump-ahead                            (multigrid/base.py:396-399)
//   P1s  lane = (env, agent): order-free evaluation of every action against the pre-step state, committed when the
//        env's agents cannot have influenced each other; otherwise P1b (rank argsort of the draws) + P1c (lane = env:
//        the reference's sequential handle_actions loop on the LDS tile)                       (base.py:378-476)
//        then the agent overlay offsets, the env subclass' post-step hook, step_count / truncated
//   P1d  lane = view: view geometry record, in-bounds lane mask, stores of agent rows / reward / terminated / dir
//   P2   lane = view CELL, slots in straight-line blocks of 16: rotate-to-facing gather from the LDS tile (one aligned 16-bit
//        read per cell), out-of-bounds -> wall, see-behind ballot (the cell's opaque bit) -> 64-bit row mask deposited in lane s;
//        cells stay in registers, two slots per register in the throughput instantiations (multigrid/utils/obs.py:130-233)
//   P3   lane = view: bit-parallel line-of-sight flood on the ballot masks (closed form of the sequential sweeps,
//        obs.py:235-273); own cell := carried object (obs.py:207)
//   P4   lane = cell: cells whose visibility bit is clear become UNSEEN (obs.py:95-100); packed cell -> (type, color, state),
//        3 bytes each into the obs byte layout in LDS (the rotate/transpose)
//   P5   ds_read_b128 -> buffer_store_dwordx4 of the (Gw,A,v,v,3) observation bytes
//
// Pure integer / byte work: no MFMA.  The roof is HBM bytes; what the kernel is actually bound by is the number of
// VALU instructions per view, so the rules of the house are: every HBM byte touched once, 16-byte
// loads and stores through buffer resources (no per-lane predicates, no 64-bit VALU addressing), per-view data to the
// cell lanes as LDS broadcasts or SGPR masks, and as few VALU instructions per slot as possible.  Workgroups touch
// disjoint memory, so the blockIdx -> XCD mapping needs no swizzle.
#pragma once
// (hipRTC -- the runtime compilation of shape-specialised kernels, multigrid_amd/jit.py -- has the HIP runtime built in and no
// system headers)
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#else
namespace __hip_internal {}
using namespace __hip_internal;          // (where hipRTC keeps the <stdint.h> names)
typedef unsigned long uintptr_t;
#ifndef offsetof
#define offsetof(t, m) __builtin_offsetof(t, m)
#endif
#ifndef INT_MAX
#define INT_MAX 2147483647
#endif
#ifndef INT64_MAX
#define INT64_MAX 9223372036854775807LL
#endif
#endif

#include "mgx_rules.h"
#include "mgx_layout_gen.h"

namespace mgx_fused {

using namespace mgx;

constexpr int kMaxThreads = 256;

struct KernelArgs {
    MgxSpec sp;
    int64_t batch;
    uint8_t *grid;
    uint8_t *agents;
    uint64_t *rng;
    int32_t *step_count;
    const int8_t *actions;
    const uint8_t *hook_order;   // u8[B,A] or null: visiting order of the RedBlueDoors / LockedHallway hooks (include/mgx.h)
    uint8_t *aux;
    uint8_t *obs;
    uint8_t *dir;
    double *reward;
    uint8_t *terminated;
    uint8_t *truncated;
    int32_t *err;
    int32_t Gw;         // envs per wavefront
    int32_t dbg;        // debug: bit p set = skip phase p (profiling only, mgx_debug_skip_phases)
    int32_t flags;      // bit 0: the grid does not fit the Infinity Cache -- launch the STREAM instantiation (nt tile loads)
                        // bit 1: few wavefronts (latency regime) -- launch the DMA instantiation (LDS-DMA tile loads)
    int32_t T;          // steps per launch (mgx_rollout), 1 otherwise
    // per-wavefront LDS slice: its stride and the slot count its carve is derived from (LdsCarve below)
    int32_t wave_lds;
    int32_t vpw;
    int32_t grp;        // slots per gather / staging group of the instantiation to launch: 16, or 4 / 8 (small-group latency family)
    int32_t inv_A;      // ceil(2^16 / A): (lane * inv_A) >> 16 == lane / A for lane < 64
    int32_t ns;         // rollout / persistent launches: slices of Gw envs per wavefront (LdsCarve); 0 = the ordinary kernels, 1 / 2 = a
                        // resident shape (kShapes[].ns): host-side geometry, the kernels have it as a constant
    int32_t rshape;     // ... which one (kShapes index; fill_args chose it)
    // fused auto-reset (mgx_step_autoreset / mgx_rollout_autoreset; include/mgx.h: MgxAutoReset)
    int32_t pool_size;
    int64_t first_env;
    uint64_t pool_magic;    // ceil(2^64 / pool_size): x mod pool_size for 32-bit x without a division (pool_index below)
    const uint8_t *pool_grid;
    const uint8_t *pool_agents;
    const uint8_t *pool_aux;
    int32_t *episode;
    uint8_t *was_reset;
    MgxLayoutGen gen;   // mgx_step_generate: finished envs are regenerated in the tail of the launch (template flag GEN)
    int64_t gen_first_wg;   // GEN with staging (include/mgx.h: MgxGenStage): workgroups from this index on are GENERATOR wavefronts, one
                            // lane per env, that serve the snapshot requests of earlier launches; INT64_MAX = none
    // persistent stepping (MODE 3, include/mgx.h: MgxPersistent): the launch stays resident for up to T steps, takes each step's
    // actions as tagged 8-byte granules and publishes a per-wavefront flag once the step's outputs are in memory
    const uint64_t *granules;   // u64[B, ceil(A/4)]
    uint32_t *done;             // u32[wavefronts]
    uint32_t *pctrl;            // u32[MGX_PERSIST_CTRL_WORDS]
    uint32_t timeout_ticks;     // s_memrealtime ticks (100 MHz) a wavefront waits for its granules before it gives up
    int32_t *grid_bad;  // byte grids (MgxSpec.cell_bytes = 3; MgxStepArgs.grid_bad): i32[2] counters or NULL -- [0] += cell values the packed
                        // format cannot hold, [1] += outer-ring cells that are not WALL (what mgx_pack_grid_env reports)
    int32_t *bounds;    // -DMGX_BOUNDS_CHECK=1 builds: [0] += LDS accesses outside the wavefront's slice, [1] = last site id
    int32_t span_base;  // -DMGX_TIMESTAMPS=1 builds: first record of this launch in g_span (tools/span_probe.py, tools/chain_overlap.py)
};

// gfx950's LDS does take a short access at any byte address (hipcc emits ds_write_b16 for an align-1 store), but measured
// it is far slower than aligned accesses (round 1, 3-byte cells: fused step 371 us aligned, 602 us with unaligned 16-bit
// writes in P4).  Kept for the record only.
#ifndef MGX_UA_WRITE
#define MGX_UA_WRITE 0
#endif
#ifndef MGX_EARLY_ARGS
#define MGX_EARLY_ARGS 2     // 1: the latency family only (DMA instantiations); 2: every instantiation with views <= 7x7 (the C4
                             // throughput kernel: 18.63-18.83 -> 18.45-18.54 us, three same-box passes; 9x9 and up: the compiler crashes on it)
#endif
#ifndef MGX_LATE_ARGS
#define MGX_LATE_ARGS 1
#endif
#ifndef MGX_P4_B16
#define MGX_P4_B16 1
#endif
#ifndef MGX_P4_PERM
#define MGX_P4_PERM 1     // P4: one v_perm_b32 per slot makes the store-ready bytes (0: the widening chain + per-store shifts of round 2)
#endif
#ifndef MGX_WRITELANE_NOP
#define MGX_WRITELANE_NOP 0
#endif
#ifndef MGX_LDS_DMA
#define MGX_LDS_DMA 0
#endif
#ifndef MGX_DRAWS_FIRST
#define MGX_DRAWS_FIRST 0
#endif
#ifndef MGX_BUF_STORE
#define MGX_BUF_STORE 1
#endif
// -DMGX_DEBUG_KNOBS=1 (the tools' build, `python -m multigrid_amd.build --debug-knobs` -> lib/libmgx_dbg.so): phase
// skipping and launch-geometry overrides for profiling.  The product library has neither the exports nor the branches.
#ifndef MGX_DEBUG_KNOBS
#define MGX_DEBUG_KNOBS 0
#endif
#if MGX_DEBUG_KNOBS
#define MGX_DBG(bits) (a.dbg & (bits))
#else
#define MGX_DBG(bits) 0
#endif

// -DMGX_BOUNDS_CHECK=1 (the checked build, `python -m multigrid_amd.build --checked` -> lib/libmgx_chk.so; SURVEY.md
// section 5 "LDS bounds asserts"): every computed LDS address of the fused kernel is tested against the wavefront's own
// slice [slice, slice + wave_lds) before it is used; violations are counted in KernelArgs::bounds and read back with
// mgx_debug_bounds_violations().  The product library has none of it.
#ifndef MGX_BOUNDS_CHECK
#define MGX_BOUNDS_CHECK 0
#endif
#if MGX_BOUNDS_CHECK
#define MGX_CHECK_LDS_ADDR(site, addr, bytes)                                                              \
    do {                                                                                                   \
        const uint32_t a_ = (uint32_t)(addr), lo_ = (uint32_t)(wave * a.wave_lds);                         \
        if (a_ < lo_ || a_ + (uint32_t)(bytes) > lo_ + (uint32_t)a.wave_lds) {                             \
            atomicAdd(a.bounds, 1); a.bounds[1] = (site);                                                  \
        }                                                                                                  \
    } while (0)
#define MGX_CHECK_LDS_PTR(site, ptr, bytes) MGX_CHECK_LDS_ADDR(site, (uint32_t)(uintptr_t)(lds_u32_ptr)(const void *)(ptr) - (uint32_t)(uintptr_t)(lds_u32_ptr)(const void *)lds, bytes)
#else
#define MGX_CHECK_LDS_ADDR(site, addr, bytes) ((void)0)
#define MGX_CHECK_LDS_PTR(site, ptr, bytes) ((void)0)
#endif

// A kernel argument fetched where it is used (s_load from the kernarg segment) instead of living in SGPRs from the
// kernel's first instruction on: the fused kernel is short of SGPRs, and every spilled one costs VALU
// v_writelane / v_readlane instructions on a VALU-bound kernel.  Only for the fields used late and rarely.
template <typename T>
__device__ __forceinline__ T kernarg_at(size_t offset) {
    typedef const char __attribute__((address_space(4))) *cptr;
    typedef const T __attribute__((address_space(4))) *tptr;
    return *(tptr)((cptr)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}
#if MGX_LATE_ARGS
#define MGX_LATE(field) kernarg_at<decltype(KernelArgs::field)>(offsetof(KernelArgs, field))
#else
#define MGX_LATE(field) (a.field)
#endif

// A launch lasts as long as its slowest wavefront, and the slowest is the one that took a rare path (restart, events, sequential
// fallback): from there on it asks for issue priority over the wavefronts it shares its SIMD with, which finish early anyway.
#ifndef MGX_RARE_PRIO
#define MGX_RARE_PRIO 1
#endif
#if MGX_RARE_PRIO
#define MGX_RARE_PATH_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define MGX_RARE_PATH_PRIO() ((void)0)
#endif

// per-view record written by P1d and read (broadcast) by the wavefront in P2
// origin: LDS address of the agent's own cell; steps / lo / hi: mgx_rules.h ViewClamp (packed i16 pairs, low half = forward)
struct alignas(16) ViewRec { int32_t origin; uint32_t steps, lo, hi; };           // 16 bytes

typedef const uint32_t __attribute__((address_space(3))) *lds_u32_ptr;
typedef const uint16_t __attribute__((address_space(3))) *lds_u16_ptr;
typedef const int8_t __attribute__((address_space(3))) *lds_i8_ptr;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef const u32_unaligned __attribute__((address_space(3))) *lds_u32_ua_ptr;
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;

#ifndef MGX_SLOTS_SMALL_VIEW
#define MGX_SLOTS_SMALL_VIEW 64
#endif
#ifndef MGX_NO_BIG_PERSIST
#define MGX_NO_BIG_PERSIST 0
#endif
#ifndef MGX_NO_FIXED_SHAPES
#define MGX_NO_FIXED_SHAPES 0     // 1: build without the shape-specialised instantiations (kShapes below): A/B builds
#endif
constexpr int kSlotsSmallView = MGX_SLOTS_SMALL_VIEW;
// cache policy bits of the obs stores (raw buffer store `aux`: 1 = sc0, 2 = nt, 16 = sc1 on gfx94x/gfx950)
#ifndef MGX_OBS_AUX
#define MGX_OBS_AUX 2       // nt: the observation is written once and read by another kernel (C4 -4.6 %, C3 -2 %, C5 -1.7 %)
#endif
#ifndef MGX_OBS_AUX_CACHED
#define MGX_OBS_AUX_CACHED MGX_OBS_AUX   // ... of the instantiations whose grid tensor fits the Infinity Cache (!STREAM)
#endif
#ifndef MGX_OH_AUX
#define MGX_OH_AUX 0        // one-hot observation stores: default policy (nt measured 6 % slower on this 7x larger write stream)
#endif
#ifndef MGX_IN_AUX
#define MGX_IN_AUX 0        // small state loads (agent rows, PCG64 words, step counts, actions) of the STREAM instantiations
#endif
#ifndef MGX_OUT_AUX
#define MGX_OUT_AUX 2       // small per-agent outputs (rows, reward, terminated, dir): nt (round 3: fewer dirty lines for the end-of-kernel
                            // write-back, C4 19.30 -> 18.82 us over three alternating runs; C2 / C5 unchanged)
#endif
// the env lanes' own small stores (PCG64 words, step count, truncated, was_reset): 1 = non-temporal too
#ifndef MGX_STATE_NT
#define MGX_STATE_NT 0
#endif
#if MGX_STATE_NT
#define MGX_STORE_SMALL(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define MGX_STORE_SMALL(ptr, val) (*(ptr) = (val))
#endif
// cache policy bits of the grid tile loads (same encoding)
#ifndef MGX_TILE_AUX
#define MGX_TILE_AUX 0
#endif
#ifndef MGX_ROUND
#define MGX_ROUND 16
#endif
constexpr int kRound = MGX_ROUND;        // view slots whose obs bytes are staged in LDS at a time (P4/P5)

// View slots per wavefront.  Views of up to 7x7 (one lane pass per view): 64 slots in the throughput instantiations, whose
// cell registers hold two slots each (the per-agent phases then run on all 64 lanes: their cost per view halves); 32 in the
// latency instantiations (DMA: launches of <= 2048 wavefronts use at most 32 slots anyway, and the packing is two more
// instructions on a lone wave's chain) and in the rollout kernel.
// (the rollout kernel, whose step loop already holds ~160 VGPRs, stays at 32)
// `narrow`: the rollout and the gen_obs-only kernels (little per-agent work to amortise: measured 3-5 % faster at 32)
inline int slots_per_wave(int view_size, bool narrow = false) { return (view_size <= 7 && !narrow) ? kSlotsSmallView : 32; }
constexpr int kSlotsLatency = 32;        // DMA instantiations

// Carve of ONE wavefront's LDS slice (byte offsets, all multiples of 16).  Everything is a closed form of
// (vpw, nw, Gw, A, tile bytes, round bytes) so the kernel recomputes an offset where it needs it instead of carrying
// fifteen of them in SGPRs from the kernel arguments.  Per-slot arrays first (vpw = slots in use, a multiple of 16).
// (the members are always_inline: left to the inliner's budget the big GEN kernels called tile() / out() out of line, which put the
// whole struct in scratch memory and every wavefront of the launch through its set-up)
// COMPACT cells (include/mgx.h: MgxCell8): the wavefront keeps a 256-entry decode table in its slice, indexed by the SIGNED cell
// byte -- entry (int8)b holds cell8_unpack(b) = type | color << 8 | state << 16 -- so that P2 turns a gathered byte into the
// observation's three bytes with one more LDS read (the address is one v_lshl_add from the sign-extended byte the opaque test
// reads anyway) instead of the ~8 VALU instructions of the arithmetic decode.
constexpr int kLutBytes = 256 * 4;
// SLICES (round 6; ns > 1, rollout / persistent kernels of the resident shapes only -- kShapes[].ns): a wavefront owns `ns` groups of
// Gw envs and steps them ONE AFTER THE OTHER, every step.  What lives across steps exists once per slice (agent rows, PCG64 words, step
// counts, hook state, the grid tile: the `s` argument of the accessors); everything a step only needs while it runs -- view records /
// draws / rewards, written-cell offsets, actions, visiting order, the jump table, the WALL cell -- exists once per wavefront, and the
// P4/P5 staging of a round lies over the view records + the written-cell offsets (both dead once P2 has gathered the cells; a round is
// then 8 view slots).  C4: 1488 + 2 x 9344 = 20176 bytes for 32 envs -> 8 wavefronts per CU hold 65536 envs' tiles on the chip.
struct LdsCarve {
    int vpw, nw, Gw, A, tile_bytes, round_bytes;   // round_bytes: P4/P5 staging of one round (obs bytes, or one-hot cell masks)
    bool roll;      // mgx_rollout: tile and PCG64 state live across steps (no aliasing of the tile, rng kept in LDS)
    bool has_aux;   // env kinds with hook state
    bool c8;        // compact cells: + the decode table
    int ns = 1;     // slices per wavefront (above)
    bool sliced = false;   // the layout described above (shared block + per-slice blocks, staging over the view records): every ns > 1
                           // kernel, and the one-slice kernel that must fit 16 wavefronts per CU (kShapes 9)
    __host__ __device__ __attribute__((always_inline)) int tile_pad() const { return (tile_bytes & 15) ? 32 : 0; }
    // -- ns > 1: the shared block [0, sh_end()), then one block of pstride() bytes per slice --
    __host__ __device__ __attribute__((always_inline)) int sh_end() const { return 22 * vpw + 32 * A + 16; }
    __host__ __device__ __attribute__((always_inline)) int pstride() const {
        return 8 * vpw + 32 * Gw + ((4 * Gw + 15) & ~15) + (has_aux ? 16 * Gw : 0) + tile_bytes + tile_pad();
    }
    __host__ __device__ __attribute__((always_inline)) int pers(int s) const { return sh_end() + s * pstride(); }
    __host__ __device__ __attribute__((always_inline)) int rows(int s = 0) const { return sliced ? pers(s) : 0; }   // u64  [vpw]
    // -- per-step temporaries, all dead once P2 has gathered the cells --
    // (the view records, written in P1d, lie over the draws and the rewards, both dead by then)
    __host__ __device__ __attribute__((always_inline)) int rec() const { return sliced ? 0 : 8 * vpw; }            // ViewRec [vpw]  (P1d -> P2)
    __host__ __device__ __attribute__((always_inline)) int rnd() const { return rec(); }                            // u64  [vpw]     (P1a -> P1b), same space
    __host__ __device__ __attribute__((always_inline)) int rew() const { return rec() + 8 * vpw; }                  // f64  [vpw]     (P0 -> hooks), same space
    __host__ __device__ __attribute__((always_inline)) int woff() const { return rec() + 16 * vpw; }                // i32  [vpw]     (P1s)
    __host__ __device__ __attribute__((always_inline)) int temps_end() const { return woff() + 4 * vpw; }
    // -- state that lives across phases / steps --
    __host__ __device__ __attribute__((always_inline)) int act() const { return temps_end(); }                      // i8   [vpw]
    __host__ __device__ __attribute__((always_inline)) int ord() const { return act() + vpw; }                      // u8   [vpw]
    __host__ __device__ __attribute__((always_inline)) int rng(int s = 0) const { return sliced ? pers(s) + 8 * vpw : ord() + vpw; }   // u64  [Gw][4]   (rollout only)
    __host__ __device__ __attribute__((always_inline)) int scnt(int s = 0) const { return rng(s) + (roll ? 32 * Gw : 0); }    // i32  [Gw]
    __host__ __device__ __attribute__((always_inline)) int aux(int s = 0) const { return scnt(s) + ((4 * Gw + 15) & ~15); }   // u8   [Gw][16]  (hook envs only)
    __host__ __device__ __attribute__((always_inline)) int own_jump() const { return sliced ? ord() + vpw : aux() + (has_aux ? 16 * Gw : 0); }
    __host__ __device__ __attribute__((always_inline)) int jump() const { return own_jump(); }                      // u64  [A][4]: k = 1..A   (rollout only: the
    __host__ __device__ __attribute__((always_inline)) int wall() const { return own_jump() + (roll ? 32 * A : 0); }   // one-step kernels keep them in registers)
                                                                                     // wall: one WALL cell + the dword after it
    // P4/P5 staging of one round's obs bytes (skew + pad).  One-step kernels put it over the tile, which is dead once
    // P2 has gathered the cells; the rollout keeps the tile and uses the (equally dead) temporaries' space + its own.
    __host__ __device__ __attribute__((always_inline)) int out_bytes() const { return (round_bytes + 32 + 15) & ~15; }
    __host__ __device__ __attribute__((always_inline)) int lut() const { return wall() + 16; }                         // u32 [256] (compact cells only)
    __host__ __device__ __attribute__((always_inline)) int own_out() const { return lut() + (c8 ? kLutBytes : 0); }
    __host__ __device__ __attribute__((always_inline)) int tile(int s = 0) const {                                      // grid bytes, skew + over-read
        return sliced ? aux(s) + (has_aux ? 16 * Gw : 0) : (roll ? own_out() + out_bytes() : own_out());
    }
    __host__ __device__ __attribute__((always_inline)) int out() const { return sliced ? rec() : (roll ? own_out() : tile()); }
    __host__ __device__ __attribute__((always_inline)) int total() const {
        if (sliced) return (pers(ns) + 15) & ~15;
        // (the tile's 32 bytes of skew + over-read exist only when a wavefront's grid bytes are not whole 16-byte vectors: a 64x64
        // env's 8 KiB tile + 512 bytes of slots is then exactly 17 x 512 bytes of LDS)
        const int t = tile_bytes + tile_pad() > out_bytes() || roll ? tile_bytes + tile_pad() : out_bytes();
        return (tile() + t + 15) & ~15;
    }
    // (ns > 1: the staging must fit the records + the written-cell offsets it lies over)
    __host__ __device__ __attribute__((always_inline)) bool slices_ok() const { return !sliced || (roll && !c8 && out_bytes() <= 20 * vpw); }
};

// one_hot: the round's staging holds one 32-bit one-hot mask per cell (+ a pad dword either side) instead of 3 obs bytes
// `round`: slots staged per P4/P5 round (kRound, or the group size of a small-group latency instantiation)
__host__ __device__ __attribute__((always_inline)) inline LdsCarve make_carve(int W, int H, int A, int V, int Gw, int vpw, bool roll, bool has_aux,
                                               bool one_hot = false, int round = kRound, int cb = kCellBytes, int ns = 1, bool sliced = false,
                                               int pitch = 0) {
    // (pitch: a tile whose rows and envs SHARE the WALL ring -- row pitch W - 1, env stride (W - 1)(H - 1) cells, W more at the end)
    const int tile_bytes = pitch ? (Gw * pitch * (H - 1) + W) * cb : Gw * H * W * cb;
    return LdsCarve{vpw, (V * V + 63) / 64, Gw, A, tile_bytes, one_hot ? round * V * V * 4 + 16 : round * V * V * 3,
                    roll, has_aux, cb == 1, ns, sliced || ns > 1};
}
inline int cell_bytes_of(const MgxSpec &sp) { return sp.cell_bytes == 1 ? 1 : kCellBytes; }        // the LDS tile's cells
inline int grid_cell_bytes_of(const MgxSpec &sp) { return sp.cell_bytes == 3 ? 3 : cell_bytes_of(sp); }   // the HBM tensors' cells

// `grp`: the group size the launch's instantiation is compiled for (16, or 4 / 8: KernelArgs::grp)
inline int slots_in_use(const MgxSpec &sp, int Gw, bool narrow = false, int grp = 16) {
    int vpw = (Gw * sp.num_agents + grp - 1) / grp * grp;     // (the kernel is compiled for slots_per_wave(V) slots)
    const int cap = slots_per_wave(sp.view_size, narrow || sp.cell_bytes == 1);   // (compact cells: one decoded cell per register)
    return vpw > cap ? cap : vpw;
}

inline int wave_lds_bytes(const MgxSpec &sp, int Gw, bool roll = false, bool one_hot = false, bool obs_only = false, int grp = 16) {
    return make_carve(sp.width, sp.height, sp.num_agents, sp.view_size, Gw, slots_in_use(sp, Gw, roll || obs_only, grp), roll,
                      sp.env_kind != MGX_KIND_EMPTY, one_hot, grp, cell_bytes_of(sp)).total();
}

constexpr int kGroupSlots = 16;           // (== kGroup, defined with the gather below)
constexpr int kLdsPerCU = 160 * 1024;
#ifndef MGX_LDS_WAVE_BUDGET
#define MGX_LDS_WAVE_BUDGET (12 * 1024)
#endif
constexpr int kLdsWaveBudget = MGX_LDS_WAVE_BUDGET;     // keeps >= 12 wavefronts per CU resident

// Envs per wavefront: as many as fit the wave's view slots and its LDS budget; fewer when the batch is too small
// to give every SIMD of the chip a few wavefronts (then latency, not throughput, is what matters).
inline int choose_Gw(const MgxSpec &sp, int64_t batch, bool roll = false, bool one_hot = false, bool obs_only = false) {
    int Gw = slots_per_wave(sp.view_size, roll || obs_only || sp.cell_bytes == 1) / sp.num_agents;
    if (Gw < 1) Gw = 1;
    while (Gw > 1 && wave_lds_bytes(sp, Gw, roll, one_hot, obs_only) > kLdsWaveBudget) --Gw;
    // a power of two: otherwise the tile's 16-byte vectors and the output rows stop lining up with the wave's lanes (round 2:
    // C4 at 14 envs per wave 30.4 us, 12: 22.1, 16: 20.8)
    while (Gw > 2 && (Gw & (Gw - 1)) != 0) Gw &= Gw - 1;
    // latency regime (fewer than two wavefronts per SIMD): one full group of 16 view slots per wave is the optimum whatever the
    // batch -- fewer slots per wave means more waves, each paying the fixed per-wave phases again (tools/group_sweep.py, round 3:
    // C2 shape at 1024 envs: 4 envs per wave 5.92 us, 2: 6.45, 1: 6.54; BlockedUnlockPickup at 8192 envs: 8 envs per wave 8.58 us,
    // 4: 10.05; more than 16 slots only once every SIMD has its two waves: C2 shape at 4096 envs, 8 envs per wave: 7.53 vs 6.27)
    while (Gw > 1 && Gw * sp.num_agents > kGroupSlots && (batch + Gw - 1) / Gw < 2048) Gw = (Gw + 1) / 2;
    return Gw;
}

// Group size of the latency family (mgx_fused_body.inc: GRP): 16 = the ordinary instantiations -- always, see has_small_groups
// below; the tools' build overrides it with mgx_debug_set_group().
inline int choose_group(const MgxSpec &sp, int64_t batch) {
    (void)sp; (void)batch;
    return 16;
}

// Shape-specialised instantiations of the latency family (template parameter SHAPE of mgx_fused_kernel; 0 = the shape is read
// from the kernel arguments).  A lone wavefront issues one instruction per ~5 cycles whatever it is, so the scalar arithmetic a
// runtime (W, H, A, envs per wavefront) costs -- LDS carve offsets, the row pitch, the lane -> (env, agent) split, loop bounds -- is
// on its chain like everything else: ~1000 of its ~2000 static scalar instructions disappear when the shape is a constant, and the
// agent loops unroll.  Measured with in-kernel spans (tools/span_probe.py, round 3): C2 wave 4.24 -> 3.78 us (step 6.6 -> 5.95),
// BlockedUnlockPickup at 16384 envs 5.08 -> 4.45 us (step 9.6 -> 8.2); the throughput instantiation of C4 gains 1 % (its scalar work
// runs beside four waves' VALU work) and has none.  The table holds the shapes BASELINE.json names, at the envs-per-wavefront
// choose_Gw gives them in the latency regime; every other shape, and these at other launch geometries, run the generic kernels.
struct FixedShape { int W, H, A, Gw; bool hooks; int V; bool dma, stream; int cb = kCellBytes; int ns = 0; int pitch = 0; };   // (dma / stream: the
                                                  // instantiation family, launch_mode; cb: bytes per grid cell; ns > 0: a RESIDENT shape of the
                                                  // rollout / persistent kernels -- 64 view slots, ns slices per wavefront, LdsCarve)
constexpr FixedShape kShapes[] = {
    {0, 0, 0, 0, false, 0, false, false},
    {16, 16, 4, 4, false, 7, true, false},      // 1: MultiGrid-Empty-16x16 x 4 agents, up to 8192 envs (C2; C4's share of an 8-GPU node)
    {16, 16, 4, 8, false, 7, true, false},      // 2: the same at 16384 envs (C4's share of a 4-GPU node, one sub-shard of the pipelined C4)
    {11, 6, 2, 8, true, 7, true, false},        // 3: MultiGrid-BlockedUnlockPickup x 2 agents (C3)
    {64, 64, 16, 1, false, 9, false, true},     // 4: the 64x64 grid x 16 agents, 9x9 views of C5 (one env per wavefront, streamed grids:
                                                //    issue-bound at ~4 wavefronts per SIMD, so it gains less: 84.7 -> 81.5 us)
    {64, 64, 16, 2, false, 9, false, true, 1},  // 5: the same on COMPACT cells (round 5): two envs per wavefront -- the per-agent phases run
                                                //    on 32 lanes instead of 16 and the 8 KiB of tiles are again 17 wavefronts per CU
    {64, 64, 16, 2, false, 9, false, false, 1}, // 6: ... with the grid left to the caches: C5's 32768 compact grids are 128 MiB, half
                                                //    of the Infinity Cache, and re-read from there (measured: -1.5 % against nt loads)
    // 7, 8 (round 6): the RESIDENT forms of Empty-16x16 x 4 agents for mgx_rollout* / mgx_step_persistent at C4's batch -- 64 view slots
    // per wavefront with two slots per cell register, as the throughput step kernel; 7: one slice of 16 envs (13216 B of LDS: 12
    // wavefronts per CU, up to 49152 envs resident), 8: TWO slices = 32 envs per wavefront (20112 B: 8 wavefronts per CU = 2048
    // wavefronts hold the 65536 envs of C4, tiles and all, for the whole launch)
    {16, 16, 4, 16, false, 7, false, false, kCellBytes, 1},
    {16, 16, 4, 16, false, 7, false, false, kCellBytes, 2},
    // 9 (round 6): one slice in 9872 B of LDS and 127 VGPRs -> SIXTEEN wavefronts per CU, four per SIMD: all 4096 wavefronts of C4's
    // 65536 envs resident in one round.  The bytes come from the tile: every env's outer ring is WALL (include/mgx.h PRECONDITION), so
    // column W - 1 of row y and column 0 of row y + 1, and the last row of env e and the first of env e + 1, can be the SAME cells --
    // row pitch 15, env stride 225 cells: 7232 B instead of 8192 -- with the per-step temporaries laid out as for the sliced kernels;
    // the registers from asking for the occupancy (amdgpu_waves_per_eu(4): the compiler rematerialises instead of holding)
    {16, 16, 4, 16, false, 7, false, false, kCellBytes, 1, 15},
#ifdef MGX_JIT_SHAPE
    {MGX_JIT_SHAPE},                            // 5: ANY other shape, compiled at run time (hipRTC) from these same headers with its
                                                //    launch geometry as MGX_JIT_SHAPE (multigrid_amd/jit.py, mgx_shape_register)
#endif
};
constexpr int kNumShapes = (int)(sizeof(kShapes) / sizeof(kShapes[0]));
constexpr int shape_slots(const FixedShape &f) { return (f.Gw * f.A + 15) / 16 * 16; }   // == slots_in_use() (all entries: <= 32 slots, the
                                                                                         // resident shapes 64)
constexpr int kShapeResident1 = 7, kShapeResident2 = 8, kShapeResident4 = 9;
constexpr int kRoundResident = 8;        // P4/P5 round of the sliced resident kernels (their staging lies over the view records)
constexpr bool shape_sliced(const FixedShape &f) { return f.ns > 1 || f.pitch != 0; }
constexpr int shape_round(const FixedShape &f) { return shape_sliced(f) ? kRoundResident : kRound; }
__host__ __device__ __attribute__((always_inline)) inline LdsCarve shape_carve(const FixedShape &f, bool roll) {
    return make_carve(f.W, f.H, f.A, f.V, f.Gw, shape_slots(f), roll, f.hooks, false, shape_round(f), f.cb, f.ns > 0 ? f.ns : 1,
                      shape_sliced(f), f.pitch);
}

static __device__ const JumpTable kJump{};

// -DMGX_MARKERS=1 (tools/isa_phase_count.py): comment lines in the assembly that delimit the phases
// -DMGX_TIMESTAMPS=1 (tools/stamp_probe.py): wavefront `g_stamp_wave` records the shader clock at every marker; implies MGX_SPANS
// -DMGX_SPANS=1 (tools/chain_overlap.py, tools/span_probe.py): every wavefront records s_memrealtime at its first and after its
//   last instruction -- two scalar clock reads and two stores per wave, nothing else changes (lib/libmgx_spans.so)
#ifndef MGX_SPANS
#define MGX_SPANS (MGX_TIMESTAMPS + 0)
#endif
#if MGX_SPANS
constexpr int kSpanCap = 1 << 18;
static __device__ unsigned long long g_span[2 * kSpanCap];   // [span_base + wave][begin, end] in s_memrealtime ticks (100 MHz): every
                                                             // launch gets its own block of records (KernelArgs::span_base, handed out
                                                             // by the host in launch order -- for a captured graph: capture order), so
                                                             // a replayed graph of several chains leaves one timeline per node
#endif
#if MGX_MARKERS
#define MGX_MARK(name) asm volatile("; MGX_MARK " name ::: "memory")
#elif MGX_TIMESTAMPS
static __device__ unsigned long long g_stamps[64];
static __device__ long long g_stamp_wave = 0;
#define MGX_MARK(name)                                                                                   \
    do {                                                                                                 \
        if (wid == g_stamp_wave && lane == 0 && stamp_i < 64) g_stamps[stamp_i] = __builtin_readcyclecounter(); \
        ++stamp_i;                                                                                       \
    } while (0)
#else
#define MGX_MARK(name) ((void)0)
#endif

// LDS traffic between lanes of ONE wavefront needs no s_barrier (the LDS executes a wave's operations in order);
// this only stops the compiler from moving LDS accesses across the phase boundary.
__device__ __forceinline__ void wave_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// a*b + c on the full-rate 24-bit multiplier (hipcc turns the C expression into quarter-rate v_mul_lo_u32 /
// v_mad_u64_u32 here)
__device__ __forceinline__ int mad24(int a, int b, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// lane `s` of `old` := the wave-uniform value `sval` (v_writelane_b32; clang has no builtin for it)
#ifndef MGX_ASM_WRITELANE
#define MGX_ASM_WRITELANE 1
#endif
__device__ __forceinline__ uint32_t set_lane(uint32_t old, uint32_t sval, const int s) {
#if MGX_ASM_WRITELANE
    // (the data operand of v_writelane has no software hazard: only an SGPR used as LANE SELECT after a VALU write needs
    // wait states; the lane is an immediate here.  -DMGX_WRITELANE_NOP=1 restores the conservative s_nop of round 1.)
#if MGX_WRITELANE_NOP
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(s));
#else
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(s));
#endif
    return old;
#else
    return __builtin_amdgcn_inverse_ballot_w64(1ull << s) ? sval : old;
#endif
}

// Raw buffer resource over `bytes` bytes at `base` (wave-uniform).  Lanes whose offset falls outside read zeros and
// their stores are dropped, so the bulk copies need neither per-lane predicates nor 64-bit VALU address arithmetic.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

// lanes whose cell has state byte (bits 16..23) == 0, in one VALU instruction (hipcc does the type compares with an
// SDWA byte select but spends an extra v_and_b32 on this one)
__device__ __forceinline__ uint64_t state_is_open(uint32_t c) {
    uint64_t m;
    asm("v_cmp_eq_u32_sdwa %0, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD" : "=s"(m) : "v"(c), "v"(0u));
    return m;
}

// Remainder packing (round 5).  A 9x9 view is 64 + 17 cells: its second lane pass used 17 of 64 lanes.  The 17-cell remainders
// of THREE views now share one pass (3 x 17 = 51 lanes): lane l of such a pass is cell 64 + l % 17 of the pass's view l / 17, so
// 16 views take 16 + 6 passes through P2 / P4 instead of 32.  Per-view data that used to be wave-uniform in the second pass
// becomes per-lane: the view record is read at a lane-varying address (three distinct addresses per instruction), the staging
// address carries the view's offset; the see-behind ballot of a shared pass is split by fixed lane ranges on the scalar unit
// (one v_writelane per view, as before), and P4 joins three views' 17-bit visibility words into the pass's lane predicate.
// Only 9x9 views have a remainder small enough (3 R <= 64); every other view size keeps one view per pass.
#ifndef MGX_PACK3
#define MGX_PACK3 1
#endif
template <int V> constexpr bool kPack3 = (MGX_PACK3 != 0) && (MGX_P4_B16 != 0) && (MGX_P4_PERM != 0) && (V * V > 64) && (V * V <= 128)
                                         && 3 * (V * V - 64) <= 64;
constexpr int pack3_passes(int n) { return (n + 2) / 3; }      // remainder passes of a block of n view slots

template <int V, int NIT>
struct LaneConst {          // cell k = lane + 64*it  <->  image[i][j], k = j*V + i
    uint32_t pk[NIT];       // (fw, la) as an i16 pair: forward distance V-1-j in the low half, lateral offset i - V/2 in the high
    int q3[NIT];
    bool act[NIT], own[NIT];
    // kPack3<V>: entry [1] describes the lane's cell of a SHARED remainder pass -- cell 64 + lane % R of view `sub` = lane / R of
    // the pass (R = V*V - 64); q3[1] includes the view's offset sub * V*V*3 in the staging area, act[1] = lane < 3 R
    int sub16;              // sub * sizeof(ViewRec): the lane's record relative to the pass's first view
};

// ---- P2 for slots [S0, S0+N): one lane per cell: rotate-to-facing gather from the LDS tile, out-of-bounds -> wall
// (obs.py:182-202); see-behind ballot (obs.py:211-233) deposited in lane s of sbLo/sbHi.  The agent's own cell still
// shows the grid here; lane s patches the carried object in afterwards (P3: its see-behind bit, P4: its bytes).
// Straight-line over the N slots (no per-slot branch) so that their LDS round trips overlap.
// One gathered cell: the packed 16 bits (zero-extended), or -- compact cells -- the SIGN-extended byte: either way the opaque
// bit is the sign of the low 16 bits, which is what the see-behind compare reads.
template <bool C8>
__device__ __forceinline__ uint32_t gather_read(uint32_t addr) {
    if constexpr (C8) return (uint32_t)(int32_t)*(lds_i8_ptr)(uintptr_t)addr;
    else return (uint32_t)*(lds_u16_ptr)(uintptr_t)addr;
}
// (type, color, state) -> the 21-bit one-hot mask of OneHotObsWrapper (bit type | bit 11 + color | bit 17 + state, dims (11, 6, 4):
// multigrid/wrappers.py:139-140, 158-190); out-of-range values set no bit, as mgx_one_hot
__device__ __forceinline__ uint32_t one_hot_mask21(uint32_t c) {
    const uint32_t p0 = min(c & 0xffu, 31u), p1 = min((c >> 8) & 0xffu, 31u), p2 = min((c >> 16) & 0xffu, 31u);
    return ((1u << p0) & 0x7ffu) | (((1u << p1) & 0x3fu) << 11) | (((1u << p2) & 0xfu) << 17);
}
// compact cells: byte -> (type, color, state) through the wavefront's decode table (LdsCarve::lut; lut_mid = the LDS address of
// entry 0, entries -128..127 around it)
__device__ __forceinline__ uint32_t lut_decode(uint32_t raw, uint32_t lut_mid) {
    return *(lds_u32_ptr)(uintptr_t)(lut_mid + (raw << 2));
}

template <int V, int NW, int S0, int N, int VPW, bool HALF, bool C8 = false>
__device__ __forceinline__ void gather_group(const KernelArgs &a, const int wave, const ViewRec *rec,
                                             const LaneConst<V, NW> &lc, uint32_t (&cell)[HALF ? VPW / 2 : VPW][NW],
                                             uint32_t (&sbLo)[NW], uint32_t (&sbHi)[NW], const uint32_t lut_mid = 0) {
    static_assert(!(HALF && C8), "two cells per register: 16-bit cells only");
    constexpr int V2 = V * V;
    // (the records are fetched half a group at a time: all N of them at once would hold 4 N registers across the block)
    constexpr int NH = N >= 16 ? N / 2 : N;
    uint32_t raw[N][NW];
#pragma unroll
    for (int h0 = 0; h0 < N; h0 += NH) {
        ViewRec r[NH];
#pragma unroll
        for (int n = 0; n < NH; ++n) r[n] = rec[S0 + h0 + n];                // broadcast reads
#pragma unroll
        for (int n = 0; n < NH; ++n) {
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                // world cell seen at image[i][j]: pos + fw*forward + la*right, with (fw, la) clamped to the part of the view
                // that lies inside the grid -- a cell outside it reads the border WALL cell next to it, which is what
                // obs.py:199-202 shows there (mgx_rules.h: ViewClamp).  Three VALU instructions per cell, no per-view lane mask.
                uint32_t t, addr;
                asm("v_pk_max_i16 %0, %1, %2" : "=v"(t) : "v"(lc.pk[it]), "v"(r[n].lo));
                asm("v_pk_min_i16 %0, %1, %2" : "=v"(t) : "v"(t), "v"(r[n].hi));
                asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(addr) : "v"(t), "v"(r[n].steps), "v"(r[n].origin));
                if (lc.act[it]) MGX_CHECK_LDS_ADDR(4, addr, C8 ? 1 : 2);
                // one aligned 16-bit read per cell (ds_read_u16; compact cells: ds_read_i8)
                raw[h0 + n][it] = gather_read<C8>(addr);
            }
        }
        if (h0 + NH < N) __builtin_amdgcn_sched_barrier(0);
    }
    [[maybe_unused]] uint32_t dec[C8 ? N : 1][NW];
    if constexpr (C8) {
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                if (lc.act[it]) MGX_CHECK_LDS_ADDR(9, lut_mid + (raw[n][it] << 2), 4);
                dec[n][it] = lut_decode(raw[n][it], lut_mid);
            }
    }
    // obs.py:46-63 see_behind as a lane mask: the cell's opaque bit is the sign of its 16 bits -- ONE compare per cell, whose
    // SGPR pair goes straight into lane s of sbLo / sbHi.  HARDWARE HAZARD (found on gfx950, not in the ISA manual's table,
    // which lists only the lane-select operand): a v_writelane_b32 whose DATA operand is an SGPR (or VCC) written by the
    // immediately preceding VALU instruction deposits the register's OLD value.  So the sequence is software-pipelined by
    // hand -- compare of cell k, then the two writelanes of cell k-1 -- in one asm block per cell: three instructions lie
    // between a compare and the writelanes that read it, and no s_nop is spent (one after the last compare of the group).
    uint64_t pend = 0;
#pragma unroll
    for (int k = 0; k < N * NW; ++k) {
        const int n = k / NW, it = k - n * NW, pn = (k - 1) / NW, pit = (k - 1) - pn * NW;
        constexpr uint64_t kAll = ~0ull;
        if constexpr (HALF) {
            if ((n & 1) == 0) cell[(S0 + n) >> 1][it] = raw[n][it] | (raw[n + 1][it] << 16);    // two slots' cells per register
        } else if constexpr (C8) {
            cell[S0 + n][it] = dec[n][it];
        } else {
            cell[S0 + n][it] = raw[n][it];
        }
        uint64_t cur;
        // (a pass with at most 32 cells -- the 17 cells of the second pass of a 9x9 view -- has no high word: one writelane)
        const bool prev_hi = V2 - 64 * pit > 32;
        if (k == 0) {
            asm volatile("v_cmp_lt_i16_e64 %0, -1, %1\n\ts_nop 1" : "=s"(cur) : "v"(raw[n][it]));   // (only one compare follows it)
        } else if (prev_hi) {
            asm volatile("v_cmp_lt_i16_e64 %0, -1, %3\n\tv_writelane_b32 %1, %4, %6\n\tv_writelane_b32 %2, %5, %6"
                         : "=&s"(cur), "+v"(sbLo[pit]), "+v"(sbHi[pit])
                         : "v"(raw[n][it]), "s"((uint32_t)pend), "s"((uint32_t)(pend >> 32)), "n"(S0 + pn));
        } else {
            asm volatile("v_cmp_lt_i16_e64 %0, -1, %2\n\ts_nop 0\n\tv_writelane_b32 %1, %3, %4"
                         : "=&s"(cur), "+v"(sbLo[pit])
                         : "v"(raw[n][it]), "s"((uint32_t)pend), "n"(S0 + pn));
        }
        const uint64_t act_mask = (V2 - 64 * it >= 64) ? kAll : ((1ull << ((V2 - 64 * it) & 63)) - 1ull);
        pend = cur & act_mask;                                               // (an SALU op, or nothing)
    }
    {
        constexpr int pn = (N * NW - 1) / NW, pit = (N * NW - 1) - pn * NW;
        if constexpr (V2 - 64 * pit > 32)
            asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                         : "+v"(sbLo[pit]), "+v"(sbHi[pit]) : "s"((uint32_t)pend), "s"((uint32_t)(pend >> 32)), "n"(S0 + pn));
        else
            asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(sbLo[pit]) : "s"((uint32_t)pend), "n"(S0 + pn));
    }
}

// ---- P2 for slots [S0, S0+N) of a view size with packed remainders (kPack3<V>): one full pass per view (cells 0..63), then one
// pass per THREE views for their remainders.  Remainder pass p of the block covers slots S0 + 3p .. S0 + 3p + 2 and leaves its
// cells in cell[pk3_reg(S0, N, p)][1], one view's cell per lane, in the low half of the register (never two per register).
template <int N> constexpr int pk3_reg(int S0, int p) { return S0 / N * pack3_passes(N) + p; }

template <int V, int NW, int S0, int N, int VPW, bool HALF, bool C8 = false>
__device__ __forceinline__ void gather_group_pk3(const KernelArgs &a, const int wave, const ViewRec *rec,
                                                 const LaneConst<V, NW> &lc, uint32_t (&cell)[HALF ? VPW / 2 : VPW][NW],
                                                 uint32_t (&sbLo)[NW], uint32_t (&sbHi)[NW], const uint32_t lut_mid = 0) {
    static_assert(NW == 2, "packed remainders: views of 65..128 cells");
    static_assert(!(HALF && C8), "two cells per register: 16-bit cells only");
    constexpr int R = V * V - 64, NP = pack3_passes(N);
    constexpr int NH = N >= 16 ? N / 2 : N;
    uint32_t raw[N], rawR[NP];
#pragma unroll
    for (int h0 = 0; h0 < N; h0 += NH) {
        ViewRec r[NH];
#pragma unroll
        for (int n = 0; n < NH; ++n) r[n] = rec[S0 + h0 + n];                // broadcast reads
#pragma unroll
        for (int n = 0; n < NH; ++n) {
            uint32_t t, addr;
            asm("v_pk_max_i16 %0, %1, %2" : "=v"(t) : "v"(lc.pk[0]), "v"(r[n].lo));
            asm("v_pk_min_i16 %0, %1, %2" : "=v"(t) : "v"(t), "v"(r[n].hi));
            asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(addr) : "v"(t), "v"(r[n].steps), "v"(r[n].origin));
            MGX_CHECK_LDS_ADDR(4, addr, C8 ? 1 : 2);
            raw[h0 + n] = gather_read<C8>(addr);
        }
        if (h0 + NH < N) __builtin_amdgcn_sched_barrier(0);
    }
    // the remainder passes: every lane reads the record of ITS view (the last pass of the block may hold fewer than three views:
    // the lanes of the missing ones re-read the last view's record and are dropped from the ballot)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        constexpr int kLast = N - 3 * (NP - 1);                                  // views of the block's last pass: 1..3
        const int ns = p == NP - 1 ? kLast : 3;
        const int sub16 = ns == 3 ? lc.sub16 : min(lc.sub16, (ns - 1) * (int)sizeof(ViewRec));
        const ViewRec r = *reinterpret_cast<const ViewRec *>(reinterpret_cast<const uint8_t *>(rec + S0 + 3 * p) + sub16);
        uint32_t t, addr;
        asm("v_pk_max_i16 %0, %1, %2" : "=v"(t) : "v"(lc.pk[1]), "v"(r.lo));
        asm("v_pk_min_i16 %0, %1, %2" : "=v"(t) : "v"(t), "v"(r.hi));
        asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(addr) : "v"(t), "v"(r.steps), "v"(r.origin));
        if (lc.act[1]) MGX_CHECK_LDS_ADDR(4, addr, C8 ? 1 : 2);
        rawR[p] = gather_read<C8>(addr);
    }
    [[maybe_unused]] uint32_t dec[C8 ? N : 1], decR[C8 ? NP : 1];
    if constexpr (C8) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
            MGX_CHECK_LDS_ADDR(9, lut_mid + (raw[n] << 2), 4);
            dec[n] = lut_decode(raw[n], lut_mid);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (lc.act[1]) MGX_CHECK_LDS_ADDR(9, lut_mid + (rawR[p] << 2), 4);
            decR[p] = lut_decode(rawR[p], lut_mid);
        }
    }
    // see-behind ballots of the full passes: compare of view n, then the two writelanes of view n-1 (the v_writelane data hazard,
    // gather_group below)
    uint64_t pend = 0;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        if constexpr (HALF) {
            if ((n & 1) == 0) cell[(S0 + n) >> 1][0] = raw[n] | (raw[n + 1] << 16);
        } else if constexpr (C8) {
            cell[S0 + n][0] = dec[n];
        } else {
            cell[S0 + n][0] = raw[n];
        }
        uint64_t cur;
        if (n == 0) {
            asm volatile("v_cmp_lt_i16_e64 %0, -1, %1\n\ts_nop 1" : "=s"(cur) : "v"(raw[n]));
        } else {
            asm volatile("v_cmp_lt_i16_e64 %0, -1, %3\n\tv_writelane_b32 %1, %4, %6\n\tv_writelane_b32 %2, %5, %6"
                         : "=&s"(cur), "+v"(sbLo[0]), "+v"(sbHi[0])
                         : "v"(raw[n]), "s"((uint32_t)pend), "s"((uint32_t)(pend >> 32)), "n"(S0 + n - 1));
        }
        pend = cur;
    }
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                 : "+v"(sbLo[0]), "+v"(sbHi[0]) : "s"((uint32_t)pend), "s"((uint32_t)(pend >> 32)), "n"(S0 + N - 1));
    // ... of the shared passes: the pass's 64-bit ballot is cut into its views' R-bit words on the scalar unit, so every v_writelane
    // reads an SGPR that a SCALAR instruction wrote (no hazard)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        constexpr int kLast = N - 3 * (NP - 1);
        const int ns = p == NP - 1 ? kLast : 3;
        if constexpr (C8) cell[pk3_reg<N>(S0, p)][1] = decR[p];
        else cell[pk3_reg<N>(S0, p)][1] = rawR[p];
        uint64_t cur;
        asm volatile("v_cmp_lt_i16_e64 %0, -1, %1" : "=s"(cur) : "v"(rawR[p]));
        constexpr uint32_t kWord = (1u << R) - 1u;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < ns) {
                const uint32_t part = (uint32_t)(cur >> (R * q)) & kWord;
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(sbLo[1]) : "s"(part), "n"(S0 + 3 * p + q));
            }
        }
    }
    (void)sbHi;
}

#ifndef MGX_GROUP
#define MGX_GROUP 16
#endif
constexpr int kGroup = MGX_GROUP;      // slots gathered (P2) / written (P4) as one straight-line block

template <int V, int NW, int VPW, bool HALF, int G, bool C8 = false, int S0 = 0>
__device__ __forceinline__ void gather_all(const KernelArgs &a, const int wave, int NVc, const ViewRec *rec,
                                           const LaneConst<V, NW> &lc, uint32_t (&cell)[HALF ? VPW / 2 : VPW][NW],
                                           uint32_t (&sbLo)[NW], uint32_t (&sbHi)[NW], const uint32_t lut_mid = 0) {
    if constexpr (S0 < VPW) {
        // whole groups only: P1d pads the records of a ragged last group with views of nothing (all lanes outside the grid)
        if (S0 < NVc) {
            if constexpr (kPack3<V>) gather_group_pk3<V, NW, S0, G, VPW, HALF, C8>(a, wave, rec, lc, cell, sbLo, sbHi, lut_mid);
            else gather_group<V, NW, S0, G, VPW, HALF, C8>(a, wave, rec, lc, cell, sbLo, sbHi, lut_mid);
        }
        gather_all<V, NW, VPW, HALF, G, C8, S0 + G>(a, wave, NVc, rec, lc, cell, sbLo, sbHi, lut_mid);
    }
}

// Every wavefront is autonomous: it owns Gw consecutive envs (<= VPW agent views) and a private LDS slice, and
// runs all phases for them without any workgroup barrier.  A workgroup is just a bundle of such wavefronts.
// MODE 0: gen_obs only.  MODE 1: one step.  MODE 2: a.T consecutive steps with the envs' state kept in LDS between
// steps (mgx_rollout); per-step outputs go to the [t] slices of the output tensors, the state is written back once.
// HOOKS: the env kind has a post-step hook and 16 bytes of hook state (every kind but EMPTY).  The EMPTY instantiation
// drops that code and its SGPRs.
// AR: fused auto-reset -- an env whose episode ended with the previous step restarts from the layout pool before this
// step's actions are applied (== mgx_reset_done followed by the step, in one launch).
// OH: the observation is written one-hot encoded, u8[B,A,V,V,21] (OneHotObsWrapper, multigrid/wrappers.py:158-190, dims
// (11, 6, 4)): P4 leaves a 21-bit mask per cell in LDS, P5 expands mask bits to 0/1 bytes, 16 at a time.
// GEN: the envs whose episode ends with this step are regenerated in the tail of the launch (== mgx_reset_generate run right
// after the step: the reference's _gen_grid on the device, mgx_layout_gen.h).
// STREAM: the grid tensor is larger than the Infinity Cache can keep between steps: non-temporal tile loads.
// DMA: the tile is loaded HBM -> LDS by LDS-DMA (small launches: P0).
// C8: the grid is held as COMPACT one-byte cells (include/mgx.h: MgxCell8; MgxSpec.cell_bytes = 1): one-step and gen_obs kernels of
// the throughput / streamed families only.
// B3: the grid tensors are the reference's byte triples u8[B,H,W,3] (MgxSpec.cell_bytes = 3), packed into the 16-bit tile by P0.
template <int V, int MODE, bool HOOKS, bool AR, bool OH = false, bool GEN = false, bool STREAM = false, bool DMA = (MGX_LDS_DMA != 0),
          int GRP = kGroup, int SHAPE = 0, bool C8 = false, bool B3 = false>
__global__ __launch_bounds__(kMaxThreads) void mgx_fused_kernel(const KernelArgs a) {
#include "mgx_fused_body.inc"
}

// The one-slice resident kernel that must leave FOUR wavefronts per SIMD (kShapes 9): its own entry point carries the occupancy request
// (128 VGPRs; left to itself the allocator takes 146-153 for this body: profiles/r6_resident.txt).
template <int MODE, bool AR>
__global__ __launch_bounds__(kMaxThreads) __attribute__((amdgpu_waves_per_eu(4)))
void mgx_resident_kernel(const KernelArgs a) {
    constexpr int V = 7, GRP = kGroup, SHAPE = kShapeResident4;
    constexpr bool HOOKS = false, OH = false, GEN = false, STREAM = false, DMA = false, C8 = false, B3 = false;
#include "mgx_fused_body.inc"
}

// gen_obs for views up to 7x7 is a pure stream and wants the 6th wavefront per SIMD, i.e. <= 80 VGPRs: left to itself the
// register allocator lands on 78 or 86 depending on unrelated code (measured: 196 vs 204 us at 1M envs).  Its own entry point
// carries the occupancy request; the step kernels are issue-bound and take the registers they want (forcing them costs
// spills).  (A shared __device__ function for the body perturbs the other kernels' allocation by ~10 VGPRs: hence the include.)
template <int V, bool OH, bool STREAM, bool DMA, bool C8 = false, bool B3 = false>
__global__ __launch_bounds__(kMaxThreads) __attribute__((amdgpu_waves_per_eu(6)))
void mgx_obs_kernel(const KernelArgs a) {
    constexpr int MODE = 0, GRP = kGroup, SHAPE = 0;
    constexpr bool HOOKS = false, AR = false, GEN = false;
#include "mgx_fused_body.inc"
}

// The kernel instantiation for (V, mode, hooks, auto-reset) and its launch.  `hip_err` receives the HIP error code of a
// failed launch (mgx_last_hip_error).
// The small-group latency instantiations (GRP 4 / 8: a wavefront owns ONE group of 4 or 8 view slots) exist only in the tools'
// build (-DMGX_DEBUG_KNOBS=1, lib/libmgx_dbg.so; mgx_debug_set_group), for the step kernel of views up to 9x9.  Measured
// (tools/group_sweep.py, profiles/r3_small_group_sweep.txt): they do NOT pay -- every additional wavefront on a SIMD costs its
// whole instruction stream in issue slots (~1.5 us per 1024 wavefronts of this kernel), so halving a wave's slots and doubling
// the wavefronts only moves the work: C2 6.27 us (4 envs per wave, 16 slots) vs 6.80 (2 envs, group of 8) vs 8.80 (1 env, group
// of 4); only below one 4-slot wave per SIMD (1024 envs of C2 shape) is the group of 4 ahead, by 0.1 us.  The product library
// does not carry them.
constexpr bool has_small_groups(int V, int MODE, bool OH, bool GEN) {
    return MGX_DEBUG_KNOBS != 0 && MODE == 1 && V <= 9 && !OH && !GEN;
}

// Which entry of kShapes, if any, the launch geometry the host derived for the plain step of the latency family matches exactly
// (0 = none: the generic kernel).  Also what mgx_launch_info reports.
// `persist`: the persistent step kernel (MODE 3: the rollout's carve, no DMA / STREAM families) has instantiations for the latency
// shapes too -- it exists for exactly those launches
inline int match_fixed_shape(const KernelArgs &ka, bool hooks, bool persist = false) {
    if (MGX_NO_FIXED_SHAPES || ka.grp != kGroup) return 0;
    for (int k = 1; k < kNumShapes; ++k) {
        const FixedShape &f = kShapes[k];
        if (f.ns > 0) continue;                                                  // (resident shapes: match_resident_shape)
        if (persist ? !f.dma : (((ka.flags & 2) != 0) != f.dma || ((ka.flags & 1) != 0) != f.stream)) continue;
        if (ka.sp.view_size == f.V
            && ka.sp.width == f.W && ka.sp.height == f.H && ka.sp.num_agents == f.A && ka.Gw == f.Gw && hooks == f.hooks
            && ka.vpw == shape_slots(f)
            && cell_bytes_of(ka.sp) == f.cb
            && ka.wave_lds == make_carve(f.W, f.H, f.A, f.V, f.Gw, shape_slots(f), persist, f.hooks, false, kGroup, f.cb).total())
            return k;
    }
    return 0;
}

// ... and which RESIDENT shape (kShapes[].ns > 0) the geometry of a rollout / persistent launch is: fill_args chose it (ka.ns), this
// re-checks every number the instantiation was compiled for.
inline int match_resident_shape(const KernelArgs &ka, bool hooks) {
    if (MGX_NO_FIXED_SHAPES || ka.grp != kGroup || ka.ns < 1 || ka.rshape < 1 || ka.rshape >= kNumShapes) return 0;
    const FixedShape &f = kShapes[ka.rshape];
    if (f.ns == ka.ns && ka.sp.view_size == f.V && ka.sp.width == f.W && ka.sp.height == f.H && ka.sp.num_agents == f.A && ka.Gw == f.Gw
        && hooks == f.hooks && ka.vpw == shape_slots(f) && cell_bytes_of(ka.sp) == f.cb && ka.wave_lds == shape_carve(f, true).total())
        return ka.rshape;
    return 0;
}

#if !defined(__HIPCC_RTC__)      // host side: the launchers (a runtime-compiled translation unit holds kernels only)
// Shape-specialised kernels compiled at run time and handed to mgx_shape_register (mgx_kernels.hip): looked up by the launch geometry
// the host derived, exactly like the built-in kShapes entries.  fn[ar]: the plain step without / with the fused auto-reset.
struct JitShape { FixedShape f; int vpw, wave_lds; hipFunction_t fn[2]; };
const JitShape *jit_shape_lookup(const KernelArgs &ka, bool hooks);
// Compact cells (C8): the plain step / gen_obs of the throughput and streamed families -- hooks and auto-reset included, no one-hot,
// generation, rollout or latency (LDS-DMA) instantiation: fill_args never asks for one on such a spec.
// (... and, with C8 = false / B3 = true, the byte-grid family: the same set of kernels for MgxSpec.cell_bytes = 3)
template <int V, int MODE, bool STREAM, bool C8 = true, bool OH = false>
inline int launch_compact(const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream, int *hip_err, int *occupancy) {
    constexpr bool B3 = !C8;
    if constexpr (MODE > 1 && C8 && !OH && !STREAM) {
        // compact cells, round 6: the hook-free rollout / persistent kernels (32 view slots, the tile resident as bytes)
        if (ka.grp != kGroup || ka.vpw > 32) return MGX_ERR_INVALID_ARGUMENT;
        if (ka.sp.env_kind != MGX_KIND_EMPTY) return MGX_ERR_UNSUPPORTED;
        const bool ar = ka.pool_grid != nullptr;
        void (*kern)(const KernelArgs) = ar ? mgx_fused_kernel<V, MODE, false, true, false, false, false, false, kGroup, 0, true, false>
                                            : mgx_fused_kernel<V, MODE, false, false, false, false, false, false, kGroup, 0, true, false>;
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) { *hip_err = (int)e; (void)hipGetLastError(); return MGX_ERR_LAUNCH; }
        }
        if (occupancy) {
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, reinterpret_cast<const void *>(kern), threads, (size_t)lds_bytes);
            if (e != hipSuccess) { *hip_err = (int)e; (void)hipGetLastError(); return MGX_ERR_LAUNCH; }
            return MGX_OK;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(threads), (size_t)lds_bytes, stream, ka);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { *hip_err = (int)e; return MGX_ERR_LAUNCH; }
        return MGX_OK;
    } else if constexpr (MODE > 1 || (OH && (MODE != 1 || !C8))) {
        return MGX_ERR_UNSUPPORTED;
    } else {
        if constexpr (!STREAM && C8) {    // (byte grids have no streamed family: a u8[B,H,W,3] tensor beyond the Infinity Cache is
                                          // loaded with the default policy -- 35 instantiations nobody asked for, round 6)
            if (ka.flags & 1) return launch_compact<V, MODE, true, C8, OH>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
        }
        if (ka.grp != kGroup || (ka.flags & 2) || (C8 && ka.vpw > 32)) return MGX_ERR_INVALID_ARGUMENT;     // (C8: 32 view slots: one decoded cell per register)
        void (*kern)(const KernelArgs) = nullptr;
        const bool hooks = MODE != 0 && ka.sp.env_kind != MGX_KIND_EMPTY;
        const bool ar = MODE != 0 && ka.pool_grid != nullptr;
        constexpr bool S = MODE != 0;
        if constexpr (C8 && !OH && MODE == 1 && V == 9 && STREAM && !MGX_NO_FIXED_SHAPES) {
            if (match_fixed_shape(ka, hooks) == 5)
                kern = ar ? mgx_fused_kernel<V, 1, false, true, false, false, true, false, kGroup, 5, true>
                          : mgx_fused_kernel<V, 1, false, false, false, false, true, false, kGroup, 5, true>;
        }
        if constexpr (C8 && !OH && MODE == 1 && V == 9 && !STREAM && !MGX_NO_FIXED_SHAPES) {
            if (match_fixed_shape(ka, hooks) == 6)
                kern = ar ? mgx_fused_kernel<V, 1, false, true, false, false, false, false, kGroup, 6, true>
                          : mgx_fused_kernel<V, 1, false, false, false, false, false, false, kGroup, 6, true>;
        }
        if constexpr (OH) {
            // compact cells, one-hot output (round 6): the hook-free step (big grids are Empty-style arenas; the hook envs are small
            // and keep the 16-bit cells)
            if (hooks) return MGX_ERR_UNSUPPORTED;
            kern = ar ? mgx_fused_kernel<V, 1, false, true, true, false, STREAM, false, kGroup, 0, true, false>
                      : mgx_fused_kernel<V, 1, false, false, true, false, STREAM, false, kGroup, 0, true, false>;
        }
        if (!kern) {
            // (the byte-grid gen_obs holds its conversion's staging registers: it takes the step kernels' entry point, without the
            // occupancy request of mgx_obs_kernel that would make it spill)
            if constexpr (MODE == 0 && V <= 7 && !B3)
                kern = mgx_obs_kernel<V, false, STREAM, false, C8, B3>;
            else
                kern = hooks ? (ar ? mgx_fused_kernel<V, MODE, S, S, false, false, STREAM, false, kGroup, 0, C8, B3>
                                   : mgx_fused_kernel<V, MODE, S, false, false, false, STREAM, false, kGroup, 0, C8, B3>)
                             : (ar ? mgx_fused_kernel<V, MODE, false, S, false, false, STREAM, false, kGroup, 0, C8, B3>
                                   : mgx_fused_kernel<V, MODE, false, false, false, false, STREAM, false, kGroup, 0, C8, B3>);
        }
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) { *hip_err = (int)e; (void)hipGetLastError(); return MGX_ERR_LAUNCH; }
        }
        if (occupancy) {
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, reinterpret_cast<const void *>(kern), threads,
                                                                        (size_t)lds_bytes);
            if (e != hipSuccess) { *hip_err = (int)e; (void)hipGetLastError(); return MGX_ERR_LAUNCH; }
            return MGX_OK;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(threads), (size_t)lds_bytes, stream, ka);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { *hip_err = (int)e; return MGX_ERR_LAUNCH; }
        return MGX_OK;
    }
}

template <int V, int MODE, bool OH, bool GEN = false, bool STREAM = false, bool DMA = false, int GRP = kGroup>
inline int launch_mode(const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream, int *hip_err, int *occupancy) {
    if (ka.sp.cell_bytes == 1 || ka.sp.cell_bytes == 3) {  // compact cells / byte grids: their own, smaller families
        if constexpr (!OH && !GEN && !STREAM && !DMA && GRP == kGroup) {
            if (ka.sp.cell_bytes == 1) return launch_compact<V, MODE, false, true>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
            return launch_compact<V, MODE, false, false>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
        } else if constexpr (OH && MODE == 1 && !GEN && !STREAM && !DMA && GRP == kGroup) {
            if (ka.sp.cell_bytes == 1) return launch_compact<V, MODE, false, true, true>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
            return MGX_ERR_UNSUPPORTED;
        } else return MGX_ERR_UNSUPPORTED;
    }
    if constexpr (!STREAM && !DMA && MODE < 2 && !GEN) {    // (rollouts read the tile once per launch; GEN: small envs)
        if (ka.flags & 1) return launch_mode<V, MODE, OH, GEN, true, false>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
        if constexpr (!OH) {
            if (ka.flags & 2) {
                if constexpr (has_small_groups(V, MODE, OH, GEN)) {
                    if (ka.grp == 4) return launch_mode<V, MODE, OH, GEN, false, true, 4>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
                    if (ka.grp == 8) return launch_mode<V, MODE, OH, GEN, false, true, 8>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
                }
                return launch_mode<V, MODE, OH, GEN, false, true>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
            }
        }
    }
    // GEN (round 3): the plain generated step at 7x7 views also has its latency instantiation (LDS-DMA tile, one cell per register)
    if constexpr (GEN && !DMA && !STREAM && MODE == 1 && !OH && V == 7 && GRP == kGroup) {
        if (ka.flags & 2) return launch_mode<V, MODE, OH, true, false, true>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
    }
    if (ka.grp != GRP) return MGX_ERR_INVALID_ARGUMENT;       // (the host-side carve was made for another group size)
    void (*kern)(const KernelArgs) = nullptr;
    const bool hooks = MODE != 0 && ka.sp.env_kind != MGX_KIND_EMPTY;        // (gen_obs never runs a hook)
    const bool ar = MODE != 0 && ka.pool_grid != nullptr;
    constexpr bool S = MODE != 0;
    // the shape-specialised instantiations (kShapes): the plain step of the latency family at 7x7 views, picked only when the
    // launch geometry the host derived is exactly the one the instantiation was compiled for
    if constexpr (MODE == 1 && !OH && !GEN && GRP == kGroup && !MGX_NO_FIXED_SHAPES) {
        const int shape = match_fixed_shape(ka, hooks);          // (its V / dma / stream are this instantiation's: fill_args set the flags)
        if constexpr (V == 7 && DMA && !STREAM) {
            switch (shape) {
            case 1: kern = ar ? mgx_fused_kernel<V, 1, false, true, false, false, false, true, kGroup, 1> : mgx_fused_kernel<V, 1, false, false, false, false, false, true, kGroup, 1>; break;
            case 2: kern = ar ? mgx_fused_kernel<V, 1, false, true, false, false, false, true, kGroup, 2> : mgx_fused_kernel<V, 1, false, false, false, false, false, true, kGroup, 2>; break;
            case 3: kern = ar ? mgx_fused_kernel<V, 1, true, true, false, false, false, true, kGroup, 3> : mgx_fused_kernel<V, 1, true, false, false, false, false, true, kGroup, 3>; break;
            default: break;
            }
        }
        if constexpr (V == 9 && !DMA && STREAM) {
            if (shape == 4) kern = ar ? mgx_fused_kernel<V, 1, false, true, false, false, true, false, kGroup, 4> : mgx_fused_kernel<V, 1, false, false, false, false, true, false, kGroup, 4>;
        }
    }
    if constexpr (MODE == 3 && !OH && !GEN && V == 7 && !MGX_NO_FIXED_SHAPES) {
        // the persistent step kernel at the latency shapes (C2 / C4's 8-GPU and 4-GPU shares, C3)
        switch (match_fixed_shape(ka, hooks, true)) {
        case 1: kern = ar ? mgx_fused_kernel<V, 3, false, true, false, false, false, false, kGroup, 1> : mgx_fused_kernel<V, 3, false, false, false, false, false, false, kGroup, 1>; break;
        case 2: kern = ar ? mgx_fused_kernel<V, 3, false, true, false, false, false, false, kGroup, 2> : mgx_fused_kernel<V, 3, false, false, false, false, false, false, kGroup, 2>; break;
        case 3: kern = ar ? mgx_fused_kernel<V, 3, true, true, false, false, false, false, kGroup, 3> : mgx_fused_kernel<V, 3, true, false, false, false, false, false, kGroup, 3>; break;
        default: break;
        }
    }
    if constexpr ((MODE == 2 || MODE == 3) && !OH && !GEN && V == 7 && !MGX_NO_FIXED_SHAPES) {
        // the resident forms of the C4 shape (kShapes 7 / 8: 64 view slots, one or two slices of 16 envs per wavefront)
        switch (match_resident_shape(ka, hooks)) {
        case kShapeResident1: kern = ar ? mgx_fused_kernel<V, MODE, false, true, false, false, false, false, kGroup, kShapeResident1>
                                        : mgx_fused_kernel<V, MODE, false, false, false, false, false, false, kGroup, kShapeResident1>; break;
        case kShapeResident2: kern = ar ? mgx_fused_kernel<V, MODE, false, true, false, false, false, false, kGroup, kShapeResident2>
                                        : mgx_fused_kernel<V, MODE, false, false, false, false, false, false, kGroup, kShapeResident2>; break;
        case kShapeResident4:
            // (rollouts only.  A persistent launch of 4 x 128 VGPRs per SIMD would leave the producer / consumer kernels of its own
            // hand-shake no register to run in -- resident_shape() keeps that launch on shapes 7 / 8)
            if constexpr (MODE == 2) kern = ar ? mgx_resident_kernel<2, true> : mgx_resident_kernel<2, false>;
            else return MGX_ERR_INVALID_ARGUMENT;
            break;
        default: if (ka.ns > 0) return MGX_ERR_INVALID_ARGUMENT;               // (fill_args chose a geometry no instantiation has)
        }
    } else if (ka.ns > 0) return MGX_ERR_INVALID_ARGUMENT;
    if constexpr (MODE == 1 && !OH && !GEN && GRP == kGroup && !MGX_NO_FIXED_SHAPES) {
        if (!kern && !occupancy) {
            if (const JitShape *js = jit_shape_lookup(ka, hooks)) {          // a runtime-compiled instantiation of this very geometry
                size_t arg_size = sizeof(KernelArgs);
                void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<KernelArgs *>(&ka), HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_size,
                                  HIP_LAUNCH_PARAM_END};
                hipError_t e = hipModuleLaunchKernel(js->fn[ar ? 1 : 0], (unsigned)nwg, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds_bytes,
                                                     stream, nullptr, config);
                if (e != hipSuccess) { *hip_err = (int)e; (void)hipGetLastError(); return MGX_ERR_LAUNCH; }
                return MGX_OK;
            }
        }
    }
    if (!kern) {
        if constexpr (MODE == 0 && V <= 7 && !GEN) {
            kern = mgx_obs_kernel<V, OH, STREAM, DMA>;
        } else if constexpr (GEN) {                                          // (generation replaces the pool pick-up)
            if constexpr (DMA) {      // ... and, for BlockedUnlockPickup x 2 (C3 with its episodes generated on the device), its shape
                const int shape = MGX_NO_FIXED_SHAPES ? 0 : match_fixed_shape(ka, hooks);
                kern = shape == 3 ? mgx_fused_kernel<V, 1, true, false, false, true, false, true, kGroup, 3>
                                  : (hooks ? mgx_fused_kernel<V, MODE, S, false, OH, true, false, true> : mgx_fused_kernel<V, MODE, false, false, OH, true, false, true>);
            } else
            kern = hooks ? mgx_fused_kernel<V, MODE, S, false, OH, true> : mgx_fused_kernel<V, MODE, false, false, OH, true>;
        } else {
            kern = hooks ? (ar ? mgx_fused_kernel<V, MODE, S, S, OH, false, STREAM, DMA, GRP> : mgx_fused_kernel<V, MODE, S, false, OH, false, STREAM, DMA, GRP>)
                         : (ar ? mgx_fused_kernel<V, MODE, false, S, OH, false, STREAM, DMA, GRP> : mgx_fused_kernel<V, MODE, false, false, OH, false, STREAM, DMA, GRP>);
        }
    }
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) { *hip_err = (int)e; (void)hipGetLastError(); return MGX_ERR_LAUNCH; }   // (off HIP's sticky state too)
    }
    if (occupancy) {        // query only (mgx_sub_shards): workgroups of this instantiation that one CU holds at a time
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, reinterpret_cast<const void *>(kern), threads,
                                                                    (size_t)lds_bytes);
        if (e != hipSuccess) { *hip_err = (int)e; (void)hipGetLastError(); return MGX_ERR_LAUNCH; }
        return MGX_OK;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(threads), (size_t)lds_bytes, stream, ka);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *hip_err = (int)e; return MGX_ERR_LAUNCH; }
    return MGX_OK;
}

template <int V>
inline int launch_view(int mode, const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream,
                       int *hip_err, int *occupancy) {
    switch (mode) {                     // mode | 4: one-hot observations; | 8: tail generation (one step only)
    case 0: return launch_mode<V, 0, false>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
    case 1: return launch_mode<V, 1, false>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
    case 2: return launch_mode<V, 2, false>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
    case 3:                                                                                                // persistent stepping
        // (hipcc 7.2 crashes at -O3 on the persistent kernels of the three largest views with the bounds checks / the debug knobs
        // compiled in.  The checked and the tools' builds compile those units at -O2 (multigrid_amd/build.py: flags) and carry
        // them; only the single-translation-unit timestamps build, which wants -O3 code for its stamps, leaves them out.)
        if constexpr (MGX_NO_BIG_PERSIST != 0 && V >= 11) return MGX_ERR_UNSUPPORTED;
        else return launch_mode<V, 3, false>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
    case 4: return launch_mode<V, 0, true>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
    case 5: return launch_mode<V, 1, true>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
    case 6: return launch_mode<V, 2, true>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
    case 9: return launch_mode<V, 1, false, true>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);   // | 8: generate
    case 13: return launch_mode<V, 1, true, true>(ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
    default: return MGX_ERR_INVALID_ARGUMENT;
    }
}

// One translation unit per view size (mgx_fused_inst.hip, -DMGX_INST_V=<V>) defines its launcher:
#define MGX_FOR_EACH_VIEW(X) X(3) X(5) X(7) X(9) X(11) X(13) X(15)
#define MGX_DECLARE_LAUNCHER(V) \
    int launch_v##V(int mode, const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream, int *hip_err, \
                    int *occupancy);
MGX_FOR_EACH_VIEW(MGX_DECLARE_LAUNCHER)
#undef MGX_DECLARE_LAUNCHER

#endif  // !__HIPCC_RTC__

}  // namespace mgx_fused
