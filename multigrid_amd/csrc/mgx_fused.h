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
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

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
    int32_t inv_A;      // ceil(2^16 / A): (lane * inv_A) >> 16 == lane / A for lane < 64
    // fused auto-reset (mgx_step_autoreset / mgx_rollout_autoreset; include/mgx.h: MgxAutoReset)
    int32_t pool_size;
    int64_t first_env;
    const uint8_t *pool_grid;
    const uint8_t *pool_agents;
    const uint8_t *pool_aux;
    int32_t *episode;
    uint8_t *was_reset;
    MgxLayoutGen gen;   // mgx_step_generate: finished envs are regenerated in the tail of the launch (template flag GEN)
    int32_t *bounds;    // -DMGX_BOUNDS_CHECK=1 builds: [0] += LDS accesses outside the wavefront's slice, [1] = last site id
};

// gfx950's LDS does take a short access at any byte address (hipcc emits ds_write_b16 for an align-1 store), but measured
// it is far slower than aligned accesses (round 1, 3-byte cells: fused step 371 us aligned, 602 us with unaligned 16-bit
// writes in P4).  Kept for the record only.
#ifndef MGX_UA_WRITE
#define MGX_UA_WRITE 0
#endif
#ifndef MGX_LATE_ARGS
#define MGX_LATE_ARGS 1
#endif
#ifndef MGX_P4_B16
#define MGX_P4_B16 1
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

// per-view record written by P1d and read (broadcast) by the wavefront in P2
struct ViewRec { int32_t origin, stepF, stepL; uint32_t carry; };     // 16 bytes

typedef const uint32_t __attribute__((address_space(3))) *lds_u32_ptr;
typedef const uint16_t __attribute__((address_space(3))) *lds_u16_ptr;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef const u32_unaligned __attribute__((address_space(3))) *lds_u32_ua_ptr;
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;

#ifndef MGX_SLOTS_SMALL_VIEW
#define MGX_SLOTS_SMALL_VIEW 64
#endif
constexpr int kSlotsSmallView = MGX_SLOTS_SMALL_VIEW;
// cache policy bits of the obs stores (raw buffer store `aux`: 1 = sc0, 2 = nt, 16 = sc1 on gfx94x/gfx950)
#ifndef MGX_OBS_AUX
#define MGX_OBS_AUX 2       // nt: the observation is written once and read by another kernel (C4 -4.6 %, C3 -2 %, C5 -1.7 %)
#endif
#ifndef MGX_OH_AUX
#define MGX_OH_AUX 0        // one-hot observation stores: default policy (nt measured 6 % slower on this 7x larger write stream)
#endif
#ifndef MGX_IN_AUX
#define MGX_IN_AUX 0        // small state loads (agent rows, PCG64 words, step counts, actions) of the STREAM instantiations
#endif
#ifndef MGX_OUT_AUX
#define MGX_OUT_AUX 0       // small per-agent outputs (rows, reward, terminated, dir)
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
struct LdsCarve {
    int vpw, nw, Gw, A, tile_bytes, round_bytes;   // round_bytes: P4/P5 staging of one round (obs bytes, or one-hot cell masks)
    bool roll;      // mgx_rollout: tile and PCG64 state live across steps (no aliasing of the tile, rng kept in LDS)
    bool has_aux;   // env kinds with hook state
    __host__ __device__ int rows() const { return 0; }                               // u64  [vpw]
    // -- per-step temporaries, all dead once P2 has gathered the cells --
    // (the view records, written in P1d, lie over the draws and the rewards, both dead by then)
    __host__ __device__ int rec() const { return 8 * vpw; }                          // ViewRec [vpw]  (P1d -> P2)
    __host__ __device__ int rnd() const { return rec(); }                            // u64  [vpw]     (P1a -> P1b), same space
    __host__ __device__ int rew() const { return 16 * vpw; }                         // f64  [vpw]     (P0 -> hooks), same space
    __host__ __device__ int woff() const { return 24 * vpw; }                        // i32  [vpw]     (P1s)
    __host__ __device__ int temps_end() const { return woff() + 4 * vpw; }
    // -- state that lives across phases / steps --
    __host__ __device__ int act() const { return temps_end(); }                      // i8   [vpw]
    __host__ __device__ int ord() const { return act() + vpw; }                      // u8   [vpw]
    __host__ __device__ int rng() const { return ord() + vpw; }                      // u64  [Gw][4]   (rollout only)
    __host__ __device__ int scnt() const { return rng() + (roll ? 32 * Gw : 0); }    // i32  [Gw]
    __host__ __device__ int aux() const { return scnt() + ((4 * Gw + 15) & ~15); }   // u8   [Gw][16]  (hook envs only)
    __host__ __device__ int own_jump() const { return aux() + (has_aux ? 16 * Gw : 0); }
    __host__ __device__ int jump() const { return own_jump(); }                      // u64  [A][4]: k = 1..A   (rollout only: the
    __host__ __device__ int wall() const { return own_jump() + (roll ? 32 * A : 0); }   // one-step kernels keep them in registers)
                                                                                     // wall: one WALL cell + the dword after it
    // P4/P5 staging of one round's obs bytes (skew + pad).  One-step kernels put it over the tile, which is dead once
    // P2 has gathered the cells; the rollout keeps the tile and uses the (equally dead) temporaries' space + its own.
    __host__ __device__ int out_bytes() const { return (round_bytes + 32 + 15) & ~15; }
    __host__ __device__ int own_out() const { return wall() + 16; }
    __host__ __device__ int tile() const { return roll ? own_out() + out_bytes() : own_out(); }   // grid bytes, skew + over-read
    __host__ __device__ int out() const { return roll ? own_out() : tile(); }
    __host__ __device__ int total() const {
        const int t = tile_bytes + 32 > out_bytes() || roll ? tile_bytes + 32 : out_bytes();
        return (tile() + t + 15) & ~15;
    }
};

// one_hot: the round's staging holds one 32-bit one-hot mask per cell (+ a pad dword either side) instead of 3 obs bytes
__host__ __device__ inline LdsCarve make_carve(int W, int H, int A, int V, int Gw, int vpw, bool roll, bool has_aux,
                                               bool one_hot = false) {
    return LdsCarve{vpw, (V * V + 63) / 64, Gw, A, Gw * H * W * kCellBytes, one_hot ? kRound * V * V * 4 + 16 : kRound * V * V * 3,
                    roll, has_aux};
}

inline int slots_in_use(const MgxSpec &sp, int Gw, bool narrow = false) {
    int vpw = (Gw * sp.num_agents + 15) & ~15;     // (the kernel is compiled for slots_per_wave(V) slots)
    return vpw > slots_per_wave(sp.view_size, narrow) ? slots_per_wave(sp.view_size, narrow) : vpw;
}

inline int wave_lds_bytes(const MgxSpec &sp, int Gw, bool roll = false, bool one_hot = false, bool obs_only = false) {
    return make_carve(sp.width, sp.height, sp.num_agents, sp.view_size, Gw, slots_in_use(sp, Gw, roll || obs_only), roll,
                      sp.env_kind != MGX_KIND_EMPTY, one_hot).total();
}

constexpr int kLdsPerCU = 160 * 1024;
#ifndef MGX_LDS_WAVE_BUDGET
#define MGX_LDS_WAVE_BUDGET (12 * 1024)
#endif
constexpr int kLdsWaveBudget = MGX_LDS_WAVE_BUDGET;     // keeps >= 12 wavefronts per CU resident

// Envs per wavefront: as many as fit the wave's view slots and its LDS budget; fewer when the batch is too small
// to give every SIMD of the chip a few wavefronts (then latency, not throughput, is what matters).
inline int choose_Gw(const MgxSpec &sp, int64_t batch, bool roll = false, bool one_hot = false, bool obs_only = false) {
    int Gw = slots_per_wave(sp.view_size, roll || obs_only) / sp.num_agents;
    if (Gw < 1) Gw = 1;
    while (Gw > 1 && wave_lds_bytes(sp, Gw, roll, one_hot, obs_only) > kLdsWaveBudget) --Gw;
    while (Gw > 4 && (batch + Gw - 1) / Gw < 2048) Gw = (Gw + 1) / 2;      // measured: 4 envs/wave is the latency optimum
    while (Gw > 1 && (batch + Gw - 1) / Gw < 512) Gw = (Gw + 1) / 2;       // tiny batches: spread over the chip
    return Gw;
}

static __device__ const JumpTable kJump{};

// -DMGX_MARKERS=1 (tools/isa_phase_count.py): comment lines in the assembly that delimit the phases
// -DMGX_TIMESTAMPS=1 (tools/stamp_probe.py): wavefront `g_stamp_wave` records the shader clock at every marker
#if MGX_MARKERS
#define MGX_MARK(name) asm volatile("; MGX_MARK " name ::: "memory")
#elif MGX_TIMESTAMPS
static __device__ unsigned long long g_stamps[64];
static __device__ unsigned long long g_span[2 * 16384];      // [wave][begin, end] in s_memrealtime ticks (100 MHz), first 16384 waves
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

template <int V, int NIT>
struct LaneConst {          // cell k = lane + 64*it  <->  image[i][j], k = j*V + i
    int la[NIT], fw[NIT], q3[NIT];
    bool act[NIT], own[NIT];
};

// ---- P2 for slots [S0, S0+N): one lane per cell: rotate-to-facing gather from the LDS tile, out-of-bounds -> wall
// (obs.py:182-202); see-behind ballot (obs.py:211-233) deposited in lane s of sbLo/sbHi.  The agent's own cell still
// shows the grid here; lane s patches the carried object in afterwards (P3: its see-behind bit, P4: its bytes).
// Straight-line over the N slots (no per-slot branch) so that their LDS round trips overlap.
template <int V, int NW, int S0, int N, int VPW, bool HALF>
__device__ __forceinline__ void gather_group(const KernelArgs &a, const int wave, const uint32_t wall_addr, const ViewRec *rec,
                                             const uint32_t (&inbLo)[NW], const uint32_t (&inbHi)[NW],
                                             const LaneConst<V, NW> &lc, uint32_t (&cell)[HALF ? VPW / 2 : VPW][NW],
                                             uint32_t (&sbLo)[NW], uint32_t (&sbHi)[NW]) {
    constexpr int V2 = V * V;
    ViewRec r[N];
    uint64_t inbm[N][NW];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        r[n] = rec[S0 + n];                                                  // broadcast reads
#pragma unroll
        for (int it = 0; it < NW; ++it)                                      // lane S0+n made this slot's mask in P1d
            inbm[n][it] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(inbHi[it], S0 + n) << 32)
                        | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(inbLo[it], S0 + n);
    }
    uint32_t raw[N][NW];
    bool inb[N][NW];
#pragma unroll
    for (int n = 0; n < N; ++n) {
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            inb[n][it] = __builtin_amdgcn_inverse_ballot_w64(inbm[n][it]);
            // world cell seen at image[i][j]: pos + fw*forward + la*right; lanes looking outside the grid read the
            // wavefront's WALL cell instead (obs.py:199-202)
            const int off = mad24(lc.fw[it], r[n].stepF, mad24(lc.la[it], r[n].stepL, r[n].origin));
            const uint32_t addr = inb[n][it] ? (uint32_t)off : wall_addr;
            if (lc.act[it]) MGX_CHECK_LDS_ADDR(4, addr, 2);
            // one aligned 16-bit read per cell (ds_read_u16)
            raw[n][it] = (uint32_t)*(lds_u16_ptr)(uintptr_t)addr;
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
        } else {
            cell[S0 + n][it] = raw[n][it];
        }
        uint64_t cur;
        if (k == 0) {
            asm volatile("v_cmp_lt_i16_e64 %0, -1, %1\n\ts_nop 1" : "=s"(cur) : "v"(raw[n][it]));   // (only one compare follows it)
        } else {
            asm volatile("v_cmp_lt_i16_e64 %0, -1, %3\n\tv_writelane_b32 %1, %4, %6\n\tv_writelane_b32 %2, %5, %6"
                         : "=&s"(cur), "+v"(sbLo[pit]), "+v"(sbHi[pit])
                         : "v"(raw[n][it]), "s"((uint32_t)pend), "s"((uint32_t)(pend >> 32)), "n"(S0 + pn));
        }
        const uint64_t act_mask = (V2 - 64 * it >= 64) ? kAll : ((1ull << ((V2 - 64 * it) & 63)) - 1ull);
        pend = cur & act_mask;                                               // (an SALU op, or nothing)
    }
    {
        constexpr int pn = (N * NW - 1) / NW, pit = (N * NW - 1) - pn * NW;
        asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                     : "+v"(sbLo[pit]), "+v"(sbHi[pit]) : "s"((uint32_t)pend), "s"((uint32_t)(pend >> 32)), "n"(S0 + pn));
    }
}

#ifndef MGX_GROUP
#define MGX_GROUP 16
#endif
constexpr int kGroup = MGX_GROUP;      // slots gathered (P2) / written (P4) as one straight-line block

template <int V, int NW, int VPW, bool HALF, int S0 = 0>
__device__ __forceinline__ void gather_all(const KernelArgs &a, const int wave, int NVc, const uint32_t wall_addr, const ViewRec *rec,
                                           const uint32_t (&inbLo)[NW], const uint32_t (&inbHi)[NW],
                                           const LaneConst<V, NW> &lc, uint32_t (&cell)[HALF ? VPW / 2 : VPW][NW],
                                           uint32_t (&sbLo)[NW], uint32_t (&sbHi)[NW]) {
    if constexpr (S0 < VPW) {
        // whole groups only: P1d pads the records of a ragged last group with views of nothing (all lanes outside the grid)
        if (S0 < NVc) gather_group<V, NW, S0, kGroup, VPW, HALF>(a, wave, wall_addr, rec, inbLo, inbHi, lc, cell, sbLo, sbHi);
        gather_all<V, NW, VPW, HALF, S0 + kGroup>(a, wave, NVc, wall_addr, rec, inbLo, inbHi, lc, cell, sbLo, sbHi);
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
template <int V, int MODE, bool HOOKS, bool AR, bool OH = false, bool GEN = false, bool STREAM = false, bool DMA = (MGX_LDS_DMA != 0)>
__global__ __launch_bounds__(kMaxThreads) void mgx_fused_kernel(const KernelArgs a) {
    constexpr bool DO_STEP = MODE != 0;
    const int env_kind = HOOKS ? a.sp.env_kind : (int)MGX_KIND_EMPTY;
    constexpr bool ROLL = MODE == 2;
    constexpr int V2 = V * V;
    constexpr int NW = (V2 + 63) / 64;          // 64-bit mask words per view = lane passes per view
    constexpr bool HALF = !DMA && MODE == 1;     // cell registers hold two slots' packed cells each (throughput step kernels)
    constexpr int VPW = (V <= 7 && HALF) ? kSlotsSmallView : 32;   // view slots per wavefront (== slots_per_wave; DMA: kSlotsLatency)
    extern __shared__ __align__(16) uint8_t lds[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = a.sp.width, H = a.sp.height, A = a.sp.num_agents;
    const int HWB = H * W * kCellBytes;                     // bytes of one env's grid (packed cells)
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const int64_t e0 = wid * a.Gw;
    if (e0 >= a.batch) return;
    const int Gc = (int)min((int64_t)a.Gw, a.batch - e0);    // envs of this wavefront
    const int NVc = Gc * A;                                   // its views (<= VPW)
    const int64_t v0 = e0 * A;                                // first (env, agent) row

    uint8_t *L = lds + wave * a.wave_lds;
#if MGX_TIMESTAMPS
    int stamp_i = 0;
    MGX_MARK("start");
    if (lane == 0 && wid < 16384) g_span[2 * wid] = __builtin_amdgcn_s_memrealtime();
#endif
    const LdsCarve cv = make_carve(W, H, A, V, a.Gw, a.vpw, ROLL, HOOKS, OH);
    constexpr int kInAux = STREAM ? MGX_IN_AUX : 0;                           // the small state loads of P0
    uint64_t *rows = reinterpret_cast<uint64_t *>(L + cv.rows());             // [slot] packed agent rows
    ViewRec *rec = reinterpret_cast<ViewRec *>(L + cv.rec());                 // [slot]
    int8_t *acts = reinterpret_cast<int8_t *>(L + cv.act());                  // [slot]
    uint64_t *rngs = reinterpret_cast<uint64_t *>(L + cv.rng());              // [env][4]
    uint64_t *rnd = reinterpret_cast<uint64_t *>(L + cv.rnd());               // [slot] 53-bit draws
    uint8_t *ord = L + cv.ord();                                               // [slot] visiting order per env
    double *rew = reinterpret_cast<double *>(L + cv.rew());                   // [slot]
    int32_t *scnt = reinterpret_cast<int32_t *>(L + cv.scnt());               // [env]
    uint4 *auxl = reinterpret_cast<uint4 *>(L + cv.aux());                    // [env] 16-byte hook state (include/mgx.h)
    uint64_t *jump = reinterpret_cast<uint64_t *>(L + cv.jump());             // [A+1][4]

    MGX_MARK("P0");
    // ------------------------------------------------------------------ P0: HBM -> LDS, all loads in flight at once
    const int64_t g0 = e0 * HWB, g1 = g0 + (int64_t)Gc * HWB;       // byte range of these envs in `grid`
    const int64_t gtotal = a.batch * (int64_t)HWB;
    const int64_t ga = g0 & ~(int64_t)15;
    uint8_t *tile_raw = L + cv.tile();                              // holds global bytes [ga, ...)
    const int tile_skew = (int)(g0 - ga);
    uint8_t *tile = tile_raw + tile_skew;                            // env e's cells at tile + e*HWB
    // Every HBM load of the wavefront is issued here, back to back, into registers; then ONE unconditional
    // s_waitcnt vmcnt(0); then the LDS stores.  (Waiting under the same lane predicates as the loads makes hipcc's
    // waitcnt pass believe loads may still be pending and sprinkle vmcnt(0) -- which on CDNA also waits for every
    // older global STORE -- over the rest of the kernel.)
    constexpr int U = 8;                                                    // 8 KiB of tile per pass
    const uint8_t *gsrc = a.grid + ga;
    const int len = (int)(g1 - ga);
    const int avail = (int)min(gtotal - ga, (int64_t)INT_MAX);               // bytes readable from gsrc
    const int grec = MGX_DBG(1) ? 0 : min((len + 15) & ~15, avail);
    const __amdgpu_buffer_rsrc_t grsrc = make_rsrc(gsrc, grec);
    const int lane16 = 16 * lane;
    const int env_of_lane = (lane * a.inv_A) >> 16, agent_of_lane = lane - env_of_lane * A;   // slot `lane` = (env, agent)
    // (1) what the draws (P1a) need goes out first: this lane's env's PCG64 words and its agent's jump-ahead constants
    u32x4 in_rngA = {0, 0, 0, 0}, in_rngB = {0, 0, 0, 0};                    // (ROLL: two envs' halves, copied to LDS)
    u32x4 in_jmpA = {0, 0, 0, 0}, in_jmpB = {0, 0, 0, 0};
    if (DO_STEP && A > 1) {
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(a.rng + e0 * 4, Gc * 32);
        if (ROLL) {                                                              // env-major copy: lane l holds words 2l, 2l+1
            in_rngA = __builtin_amdgcn_raw_buffer_load_b128(rr, lane16, 0,kInAux);
            for (int t = lane; t < A * 4; t += 64) jump[t] = kJump.w[1][t];          // constants for k = 1..A, re-read every step
        } else {                                                                 // same address for the A lanes of an env
            in_rngA = __builtin_amdgcn_raw_buffer_load_b128(rr, env_of_lane * 32, 0,kInAux);
            in_rngB = __builtin_amdgcn_raw_buffer_load_b128(rr, env_of_lane * 32 + 16, 0,kInAux);
            const u32x4 *jk = reinterpret_cast<const u32x4 *>(&kJump.w[1 + agent_of_lane][0]);   // agent k draws k+1 ahead
            in_jmpA = jk[0]; in_jmpB = jk[1];
        }
    }
    u32x2 in_row = {0, 0};
    uint32_t in_scnt = 0;
    u32x4 in_aux = {0, 0, 0, 0};
    uint8_t in_act = 0;
    in_row = __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(a.agents + v0 * 8, NVc * 8), lane * 8, 0,kInAux);
    uint32_t in_ep = 0;                                                      // AR: env `lane`'s episode count
    if (DO_STEP) {
        if (AR && !ROLL) in_ep = __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(a.episode + e0, Gc * 4), lane * 4, 0, 0);
        in_scnt = __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(a.step_count + e0, Gc * 4), lane * 4, 0,kInAux);
        in_act = __builtin_amdgcn_raw_buffer_load_b8(make_rsrc(a.actions + v0, NVc), lane, 0,kInAux);
        if (cv.has_aux) in_aux = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.aux + e0 * MGX_AUX_BYTES, Gc * MGX_AUX_BYTES), lane16, 0, 0);
    }
    // (2) the tile
    [[maybe_unused]] u32x4 tv[U], tv2[U];
    const bool big_tile = len > 1024 * U;                                   // e.g. one 64x64 env per wavefront
    // Cache policy of the tile loads (template flag STREAM, chosen at launch): a grid tensor that fits the 256 MiB Infinity
    // Cache is re-read from there by the next step and is best left to the default policy; one that does not is a pure
    // stream and is loaded non-temporal (measured: 1 M envs -3.5 % step / -6.5 % gen_obs and C5 -4 % with nt, C4 +3 %).
    // (A launch-time branch around the two forms cost 3-4 % everywhere: the loads must stay in the straight-line burst.)
    constexpr int kTileAux = STREAM ? 2 : MGX_TILE_AUX;
    // DMA instantiations: the tile goes HBM -> LDS directly (buffer_load_dwordx4 ... lds: wave-uniform LDS base in M0 + 16
    // bytes per lane, which is exactly the tile's layout): no staging VGPRs (78 instead of 95: the register peak of the
    // step kernel was this burst), no ds_write pass.  Lanes past the wave's bytes are masked off -- an out-of-range lane
    // would still write its zeros into LDS.  Shorter for a lone wave (-3..5 % up to 2048 wavefronts), but the LDS-DMA path
    // has less throughput than loads + ds_write_b128 (+4 % at 65536 envs): chosen at launch by the number of wavefronts.
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
#define MGX_TILE_BURST(dst, base)                                                                                     \
    if constexpr (DMA) {                                                                                              \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                                                 \
            if (lane16 + (base) + 1024 * u < len)                                                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(grsrc, (lds_void_ptr)(tile_raw + (base) + 1024 * u), 16, lane16, \
                                                         (base) + 1024 * u, 0, kTileAux);                            \
    } else {                                                                                                          \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                                                 \
            dst[u] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, lane16 + 1024 * (u & 3), (base) + 4096 * (u >> 2), kTileAux); \
    }
#if !MGX_DRAWS_FIRST
    MGX_TILE_BURST(tv, 0)
#endif
    // (3) P1a of the one-step kernels, while the tile is still on its way: one lane per (env, agent), that agent's draw
    // by jump-ahead (base.py:399); the env's stream after A draws goes straight back to HBM
    uint64_t my_rng[4];
    my_rng[0] = ((uint64_t)in_rngA.y << 32) | in_rngA.x; my_rng[1] = ((uint64_t)in_rngA.w << 32) | in_rngA.z;
    my_rng[2] = ((uint64_t)in_rngB.y << 32) | in_rngB.x; my_rng[3] = ((uint64_t)in_rngB.w << 32) | in_rngB.z;
    uint64_t my_draw = 0;
    if (DO_STEP && !ROLL && A > 1 && !MGX_DBG((2 | 128)) && lane < NVc) {
        uint64_t jk[4];
        jk[0] = ((uint64_t)in_jmpA.y << 32) | in_jmpA.x; jk[1] = ((uint64_t)in_jmpA.w << 32) | in_jmpA.z;
        jk[2] = ((uint64_t)in_jmpB.y << 32) | in_jmpB.x; jk[3] = ((uint64_t)in_jmpB.w << 32) | in_jmpB.z;
        uint64_t s_lo, s_hi;
        my_draw = pcg64_draw_at(my_rng, jk, s_lo, s_hi);
        if (agent_of_lane == A - 1) { uint64_t *dst = a.rng + (e0 + env_of_lane) * 4; dst[0] = s_lo; dst[1] = s_hi; }
    }
#if MGX_DRAWS_FIRST
    asm volatile("" ::: "memory");
    MGX_TILE_BURST(tv, 0)
#endif
    // (3b) big tiles: a second burst under the same wait (requested here, once the draws' inputs are dead, so that the
    // register peak of P0 stays below that of the gather)
    if (big_tile) {
        MGX_TILE_BURST(tv2, 1024 * U)
    }
#undef MGX_TILE_BURST
    // (4) auto-reset test of the one-step kernels, also under the wait (build-defined, include/mgx.h): one lane per env
    // tests base.py:534-539 on the state the previous step left
    uint64_t reset_mask0 = 0;
    if (AR && DO_STEP && !ROLL) {
        const uint64_t row0 = ((uint64_t)in_row.y << 32) | in_row.x;
        const uint64_t alive = __builtin_amdgcn_ballot_w64(lane < NVc && !row_term(row0));     // bit = (env, agent) slot
        bool done = false;
        if (lane < Gc) {
            const uint64_t amask = (A >= 64) ? ~0ull : ((1ull << A) - 1ull);
            done = (((alive >> mad24(lane, A, 0)) & amask) == 0) | ((int32_t)in_scnt >= a.sp.max_steps);
            if (a.was_reset) a.was_reset[e0 + lane] = (uint8_t)done;
        }
        reset_mask0 = __builtin_amdgcn_ballot_w64(done);
    }
    // (5) the small inputs are here long before the tile: their LDS stores go first
    const uint32_t wall_addr = (uint32_t)(wave * a.wave_lds + cv.wall());
    if (lane == 0) *reinterpret_cast<uint32_t *>(L + cv.wall()) = CELL16_WALL;
    if (lane < NVc) {
        if (DO_STEP && !ROLL && A > 1) rnd[lane] = my_draw;                      // (only the sequential fallback reads the draws)
        reinterpret_cast<u32x2 *>(rows)[lane] = in_row;
        rew[lane] = 0.0;                                                         // base.py:393
        if (DO_STEP) acts[lane] = (int8_t)in_act;
    }
    if (DO_STEP) {
        if (A > 1 && ROLL && lane < Gc * 2) reinterpret_cast<u32x4 *>(rngs)[lane] = in_rngA;
        if (lane < Gc) { scnt[lane] = (int32_t)in_scnt; if (cv.has_aux) reinterpret_cast<u32x4 *>(auxl)[lane] = in_aux; }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0), for every lane
    if constexpr (!DMA) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (lane16 + 1024 * u < len) {
                MGX_CHECK_LDS_PTR(1, tile_raw + lane16 + 1024 * u, 16);
                *reinterpret_cast<u32x4 *>(tile_raw + lane16 + 1024 * u) = tv[u];
            }
    }
    if (big_tile) {
        if constexpr (!DMA) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (lane16 + 1024 * (U + u) < len) *reinterpret_cast<u32x4 *>(tile_raw + lane16 + 1024 * (U + u)) = tv2[u];
        }
        for (int rel = lane16 + 2048 * U; rel < len; rel += 1024)               // tiles larger than two bursts (16 KiB)
            *reinterpret_cast<u32x4 *>(tile_raw + rel) = __builtin_amdgcn_raw_buffer_load_b128(grsrc, rel, 0, kTileAux);
    }
    if (g1 == gtotal && (gtotal & 15)) {                          // last, partial 16-byte vector of the tensor
        const int64_t t0 = gtotal & ~(int64_t)15;
        for (int k = lane; k < (int)(gtotal & 15); k += 64) tile_raw[(int)(t0 - ga) + k] = a.grid[t0 + k];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                     // (the loops above may have loaded)
    wave_sync();

    MGX_MARK("P0end");
    // lane constants: cell k = lane + 64*it  <->  image[i][j], k = j*V + i (depth-row major, so each ballot
    // word holds whole visibility rows); lateral offset la = i - V/2, forward distance fw = V-1-j.
    auto lane_consts = [&]() {
        LaneConst<V, NW> c;
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const int k = lane + 64 * it;
            const int j = k / V, i = k - j * V;
            c.act[it] = k < V2;
            c.la[it] = i - V / 2;
            c.fw[it] = V - 1 - j;
            c.q3[it] = (i * V + j) * 3;
            c.own[it] = (i == V / 2) && (j == V - 1);
        }
        return c;
    };
    LaneConst<V, NW> lc_roll;
    if (ROLL) lc_roll = lane_consts();            // hoisted out of the step loop; the one-step kernels make them late

    const StepCfg cf = make_cfg(a.sp);
    const int T = ROLL ? a.T : 1;
    const int64_t BA = a.batch * A;
    int8_t next_act = 0;                                                     // ROLL: next step's action, in flight
    if (ROLL && lane < NVc) next_act = a.actions[v0 + lane];
    for (int t = 0; t < T; ++t) {
    const int64_t tv0 = (int64_t)t * BA + v0;                                // this step's (env, agent) output rows
    if (ROLL) {
        if (lane < NVc) { acts[lane] = next_act; rew[lane] = 0.0; }
        if (t + 1 < T && lane < NVc) next_act = a.actions[(int64_t)(t + 1) * BA + v0 + lane];
        wave_sync();
    }
    uint32_t ovl_saved = 0;                                                  // ROLL: clean cell under this agent's overlay
    int ovl_off = -1;
    // this lane's agent row, carried in registers through the step (one-step kernels have it from P0); re-read from LDS
    // only after something else may have changed it (a reset, the sequential fallback, an env hook)
    uint64_t cur_row = ROLL ? (lane < NVc ? rows[lane] : 0ull) : (((uint64_t)in_row.y << 32) | in_row.x);
    MGX_MARK("AR");
    uint64_t reset_mask = 0;                                                 // AR: envs (bit = env of the wave) restarted now
    if (AR && DO_STEP) {
        // -------------------------------------------------------------- auto-reset: a finished env takes the pool layout
        // (first_env + b + episode * 7919) mod K, step_count 0, episode + 1 -- the definition mgx_reset_done implements
        if (ROLL) {
            const uint64_t alive = __builtin_amdgcn_ballot_w64(lane < NVc && !row_term(cur_row));   // bit = (env, agent) slot
            bool done = false;
            if (lane < Gc) {
                const uint64_t amask = (A >= 64) ? ~0ull : ((1ull << A) - 1ull);
                done = (((alive >> mad24(lane, A, 0)) & amask) == 0) | (scnt[lane] >= cf.max_steps);
                if (a.was_reset) a.was_reset[(int64_t)t * a.batch + e0 + lane] = (uint8_t)done;
            }
            reset_mask = __builtin_amdgcn_ballot_w64(done);
        } else {
            reset_mask = reset_mask0;                                            // (tested in P0, under the load wait)
        }
        if (reset_mask != 0) {                                                   // rare: a few envs per thousand steps
            for (uint64_t m = reset_mask; m != 0; m &= m - 1) {
                const int e = __builtin_ctzll(m);                                // wave-uniform
                const int64_t b = e0 + e;
                int32_t *p_ep = MGX_LATE(episode);
                const int32_t K = MGX_LATE(pool_size);
                // (one-step kernels fetched the episode counts in P0: no dependent load in front of the copy)
                const int32_t ep = ROLL ? p_ep[b] : (int32_t)__builtin_amdgcn_readlane(in_ep, e);
                const int lay = (K == 1) ? 0 : (int)((uint64_t)(MGX_LATE(first_env) + b + (int64_t)ep * 7919) % (uint64_t)K);
                wave_sync();
                if (lane == 0) p_ep[b] = ep + 1;
                const uint8_t *sg = MGX_LATE(pool_grid) + (int64_t)lay * HWB;
                uint8_t *etile = tile + e * HWB;
                uint8_t *gg = MGX_LATE(grid) + b * HWB;
                const uint32_t etile_addr = (uint32_t)(uintptr_t)(lds_u32_ptr)etile;
                if (((HWB | etile_addr | (uint32_t)reinterpret_cast<uintptr_t>(sg) | (uint32_t)reinterpret_cast<uintptr_t>(gg)) & 3u) == 0) {
#pragma unroll 4
                    for (int i = lane; i < HWB / 4; i += 64) {                   // dwords: several loads in flight per lane
                        const uint32_t v = reinterpret_cast<const uint32_t *>(sg)[i];
                        reinterpret_cast<uint32_t *>(etile)[i] = v;
                        if (!ROLL) reinterpret_cast<uint32_t *>(gg)[i] = v;      // (the rollout writes its tile back at the end)
                    }
                } else {
                    for (int i = lane; i < HWB; i += 64) {                       // bytes: layouts of any size / alignment
                        const uint8_t v = sg[i];
                        etile[i] = v;
                        if (!ROLL) gg[i] = v;
                    }
                }
                const uint64_t *sa = reinterpret_cast<const uint64_t *>(MGX_LATE(pool_agents)) + (int64_t)lay * A;
                for (int j = lane; j < A; j += 64) rows[e * A + j] = sa[j];
                if (lane == 0) {
                    scnt[e] = 0;
                    if (HOOKS) {
                        const uint4 x = reinterpret_cast<const uint4 *>(MGX_LATE(pool_aux))[lay];
                        auxl[e] = x;
                        if (!ROLL) reinterpret_cast<uint4 *>(MGX_LATE(aux))[b] = x;
                    }
                }
            }
            wave_sync();
            if (!ROLL && lane < Gc && ((reset_mask >> lane) & 1ull)) in_scnt = 0;
        }
    }
    if (AR && reset_mask != 0 && lane < NVc) cur_row = rows[lane];
    double my_rew = 0.0;                                                     // this agent's reward (base.py:393), in a register
    if (DO_STEP && !MGX_DBG(2)) {
        const bool in = lane < NVc;
        // (fetched now so that the s_load latency hides behind P1a / P1s)
        int32_t *const p_step_count = ROLL ? nullptr : MGX_LATE(step_count);
        uint8_t *const p_truncated = MGX_LATE(truncated);
        MGX_MARK("P1a");
        if (ROLL && A > 1 && !MGX_DBG(128)) {
            // -------------------------------------------------------------- P1a (rollout; the one-step kernels did it in P0):
            // one lane per (env, agent): its draw
            if (in) {
                const int e = env_of_lane, ai = agent_of_lane;
                uint64_t s_lo, s_hi;
                rnd[lane] = pcg64_draw_at(rngs + e * 4, jump + ai * 4, s_lo, s_hi);         // base.py:399
                if (ai == A - 1) { rngs[e * 4 + 0] = s_lo; rngs[e * 4 + 1] = s_hi; }        // (every lane has read it: in-order LDS)
            }
        }
        MGX_MARK("P1s");
        // ------------------------------------------------------------------ P1s: one lane per (env, agent): order-free
        // evaluation of every agent's action against the pre-step state (mgx_rules.h: conditions (1)-(3))
        int32_t *woff = reinterpret_cast<int32_t *>(L + cv.woff());             // [slot]
        AgentEval ev{};
        uint8_t *mytile = tile + env_of_lane * HWB;
        if (in) MGX_CHECK_LDS_PTR(2, mytile, HWB);
        if (in && !MGX_DBG(64)) {
            const int so = (env_kind == MGX_KIND_REDBLUEDOORS)
                               ? stale_offset(cf, reinterpret_cast<const uint8_t *>(auxl + env_of_lane), env_kind) : -1;
            ev = eval_agent(cf, mytile, rows + env_of_lane * A, ROLL ? (int)acts[lane] : (int)(int8_t)in_act, cur_row, true, so);
        }
        // condition (2) needs the cells the other agents write: exchanged through LDS only when somebody writes at all
        bool conf = false;
        if (__builtin_amdgcn_ballot_w64(in && ev.writes) != 0) {
            if (in) woff[lane] = ev.writes ? ev.off : -1;
            wave_sync();
            conf = in && spec_cell_conflict(woff + env_of_lane * A, A, agent_of_lane, ev);
        }
        // the common step has none of this in the whole wavefront: one ballot decides whether the masks are needed at all
        bool fb = false, does_act = in;
        uint64_t m_ends = 0, m_evt = 0, genv = 0;
        if (__builtin_amdgcn_ballot_w64(in && (ev.bad | conf | ev.used_presence | ev.success | ev.failure)) != 0) {
            const uint64_t m_bad = __builtin_amdgcn_ballot_w64(in && ev.bad);
            const uint64_t m_conf = __builtin_amdgcn_ballot_w64(conf);
            const uint64_t m_pres = __builtin_amdgcn_ballot_w64(in && ev.used_presence);
            const uint64_t m_moved = __builtin_amdgcn_ballot_w64(in && ev.moved);
            m_ends = __builtin_amdgcn_ballot_w64(in && event_ends_all(cf, ev));
            m_evt = __builtin_amdgcn_ballot_w64(in && (ev.success | ev.failure));
            const uint64_t amask = (A >= 64) ? ~0ull : ((1ull << A) - 1ull);
            genv = in ? (amask << (env_of_lane * A)) : 0ull;                     // the lanes of this lane's env
            fb = in && spec_needs_fallback(m_bad & genv, m_conf & genv, m_pres & genv, m_moved & genv);
            // an event that ends the episode for every agent: only the agents visited up to it act (mgx_rules.h)
            does_act = in && !fb;
            if (A > 1 && m_ends != 0) {                                          // rare, wave-uniform
                int my_rank = 0;
                if (in) {
                    my_rank = draw_rank(rnd + env_of_lane * A, A, agent_of_lane);
                    ord[env_of_lane * A + my_rank] = (uint8_t)agent_of_lane;
                }
                wave_sync();
                if (in && (m_ends & genv) != 0)
                    does_act = does_act && my_rank <= event_cutoff(ord + env_of_lane * A, (m_ends & genv) >> (env_of_lane * A), A);
            }
        }
        if (does_act) {                                                              // commit
            if (ev.go) { rows[lane] = ev.nrow; cur_row = ev.nrow; }
            if (ev.unstale) reinterpret_cast<uint8_t *>(auxl + env_of_lane)[4] = 0;
            if (ev.writes) {
                store_cell(mytile + ev.off, ev.ncell);
                if (!ROLL) {                                                     // ROLL writes the whole tile back at the end
                    store_cell(MGX_LATE(grid) + (e0 + env_of_lane) * HWB + ev.off, ev.ncell);
                }
            }
        }
        if (m_evt != 0) {                                                        // on_success / on_failure of the agents that acted
            const uint64_t m_succ = __builtin_amdgcn_ballot_w64(does_act && ev.success);
            if (in && !fb) {
                if (cf.joint_reward ? (m_succ & genv) != 0 : (does_act && ev.success))
                    my_rew = reward_value(scnt[env_of_lane] + 1, cf.max_steps);          // base.py:500-507, 598-602
                if ((does_act && event_ends_self(cf, ev)) || (m_ends & genv) != 0) {          // base.py:478-498, 509-532
                    cur_row |= 1ull << 32;
                    rows[lane] = cur_row;
                }
            }
        }
        MGX_MARK("P1s_end");
        const uint64_t fbw = __builtin_amdgcn_ballot_w64(fb);                   // envs that need the sequential loop
        wave_sync();
        if (fbw != 0) {
            if (A > 1) {
                // ---------------------------------------------------------- P1b: argsort by ranking
                if (in) {
                    const int e = env_of_lane, ai = agent_of_lane;
                    ord[e * A + draw_rank(rnd + e * A, A, ai)] = (uint8_t)ai;
                }
                wave_sync();
            }
            // -------------------------------------------------------------- P1c: one lane per env: the reference's loop
            const bool mine = lane < Gc && (lane * A < 64) && ((fbw >> (lane * A)) & 1ull);
            if (mine) {
                const int e = lane;
                const int64_t b = e0 + e;
                uint8_t *etile = tile + e * HWB;
                uint8_t *ggrid = MGX_LATE(grid) + b * HWB;
                auto dirty = [=](int off) {
                    if (!ROLL) store_cell16(ggrid + off, load_cell16(etile + off));
                };
                const int rc = handle_actions(cf, etile, rows + e * A, acts + e * A, ord + e * A, rew + e * A,
                                              scnt[e] + 1, dirty, reinterpret_cast<uint8_t *>(auxl + e), env_kind);
                int32_t *errp = MGX_LATE(err);
                if (rc != 0 && errp) { atomicAdd(errp, 1); atomicMin(errp + 1, (int32_t)min(b, (int64_t)INT_MAX)); }
            }
            wave_sync();
        }
        MGX_MARK("P1hook");
        // ------------------------------------------------------------------ overlay offsets (pre-hook `terminated`, SURVEY
        // App. C Q2), then one lane per env: counters + the env subclass' hook on the clean tile, then the overlay itself
        const int ovl = (in && !MGX_DBG(512)) ? overlay_offset(cf, rows + env_of_lane * A, agent_of_lane) : -1;
        if (HOOKS && in && !fb) rew[lane] = my_rew;                              // (the hooks assign to / add onto the base rewards)
        wave_sync();
        if (lane < Gc) {
            const int e = lane;
            const int64_t b = e0 + e;
            const int32_t sc = (ROLL ? scnt[e] : (int32_t)in_scnt) + 1;          // base.py:333
            if (ROLL) scnt[e] = sc; else p_step_count[b] = sc;
            uint8_t *etile = tile + e * HWB;
            uint8_t *ggrid = MGX_LATE(grid) + b * HWB;
            auto dirty = [=](int off) {
                if (!ROLL) store_cell16(ggrid + off, load_cell16(etile + off));
            };
            uint8_t *eaux = reinterpret_cast<uint8_t *>(auxl + e);
            post_step_hook(cf, env_kind, etile, rows + e * A, acts + e * A, eaux, sc, rew + e * A, dirty);
            if (!ROLL && cv.has_aux) {                                           // the hook state the step may change
                uint8_t *gaux = MGX_LATE(aux) + b * MGX_AUX_BYTES;
                if (env_kind == MGX_KIND_LOCKEDHALLWAY) { gaux[1] = eaux[1]; gaux[2] = eaux[2]; gaux[15] = eaux[15]; }
                if (env_kind == MGX_KIND_REDBLUEDOORS) gaux[4] = eaux[4];
            }
            p_truncated[(int64_t)t * a.batch + b] = (uint8_t)(sc >= cf.max_steps);   // base.py:339
        }
        wave_sync();
        if ((HOOKS || fbw != 0) && in) {                                         // (a hook / the fallback may have terminated it
            cur_row = rows[lane];                                                // and rewarded it: their results are in LDS)
            if (HOOKS || fb) my_rew = rew[lane];
        }
        if (ROLL && ovl >= 0) ovl_saved = load_cell16(mytile + ovl);
        ovl_off = ovl;
        wave_sync();
        if (ovl >= 0) {
            MGX_CHECK_LDS_PTR(3, mytile + ovl, 2);
            store_cell16(mytile + ovl, agent_cell16(cur_row));
        }
    } else {
        const int off = (lane < NVc) ? overlay_offset(cf, rows + env_of_lane * A, agent_of_lane) : -1;
        wave_sync();
        if (off >= 0) store_cell16(tile + env_of_lane * HWB + off, agent_cell16(cur_row));
    }
    wave_sync();

    MGX_MARK("P1d");
    // ------------------------------------------------------------------ P1d: one lane per view: geometry + outputs
    const uint32_t tile_addr = (uint32_t)(wave * a.wave_lds + cv.tile() + tile_skew);   // LDS address of env 0 cell 0
    uint32_t my_carry = 0;                                                   // slot `lane`: what its agent carries
    uint32_t inbLo[NW], inbHi[NW];                                           // slot `lane`: its in-bounds lanes (P2 reads them
#pragma unroll                                                               // with v_readlane; padding slots: none in bounds)
    for (int k = 0; k < NW; ++k) { inbLo[k] = 0; inbHi[k] = 0; }
    // (the output pointers are fetched first so that the s_load latency hides behind the geometry arithmetic)
    uint8_t *const p_dir = MGX_LATE(dir);
    uint8_t *const p_agents = DO_STEP ? MGX_LATE(agents) : nullptr;
    double *const p_reward = DO_STEP ? MGX_LATE(reward) : nullptr;
    uint8_t *const p_term = DO_STEP ? MGX_LATE(terminated) : nullptr;
    if (lane < NVc) {
        const int e = env_of_lane;
        const uint64_t row = cur_row;
        const ViewGeom g = view_geom<V>(W, H, row_x(row), row_y(row), row_dir(row));
        ViewRec r;
        r.origin = (int32_t)tile_addr + e * HWB + g.origin;
        r.stepF = g.stepF; r.stepL = g.stepL; r.carry = 0;
        my_carry = row_carry(row);
        rec[lane] = r;
        uint64_t m[NW];
        inbounds_mask<V, NW>(g, m);
#pragma unroll
        for (int k = 0; k < NW; ++k) { inbLo[k] = (uint32_t)m[k]; inbHi[k] = (uint32_t)(m[k] >> 32); }
        if (DO_STEP) {
            const u32x2 rowv = {(uint32_t)row, (uint32_t)(row >> 32)};
            if (!ROLL) __builtin_amdgcn_raw_buffer_store_b64(rowv, make_rsrc(p_agents + v0 * 8, NVc * 8), lane * 8, 0,MGX_OUT_AUX);
            const uint64_t rbits = __builtin_bit_cast(uint64_t, my_rew);
            const u32x2 rewv = {(uint32_t)rbits, (uint32_t)(rbits >> 32)};
            __builtin_amdgcn_raw_buffer_store_b64(rewv, make_rsrc(p_reward + tv0, NVc * 8), lane * 8, 0,MGX_OUT_AUX);
            const bool forced = env_kind == MGX_KIND_LOCKEDHALLWAY && reinterpret_cast<const uint8_t *>(auxl + e)[15];
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(row_term(row) | forced),                    // base.py:338 (+ env hook)
                                                 make_rsrc(p_term + tv0, NVc), lane, 0, MGX_OUT_AUX);
        }
        if (p_dir) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)row_dir(row), make_rsrc(p_dir + tv0, NVc), lane, 0,MGX_OUT_AUX);   // base.py:359, 372
    } else if (lane < ((NVc + kGroup - 1) & ~(kGroup - 1))) {                // padding slots of the last gather group
        ViewRec r;
        r.origin = (int32_t)wall_addr; r.stepF = 0; r.stepL = 0; r.carry = 0;
        rec[lane] = r;
    }
    wave_sync();

    MGX_MARK("P2");
    // ------------------------------------------------------------------ P2: the wavefront renders its views, one lane per cell
    const LaneConst<V, NW> lc = ROLL ? lc_roll : lane_consts();
    uint32_t cell[HALF ? VPW / 2 : VPW][NW];     // registers: every slot's packed cells, one per lane (and pass); HALF: slots 2p
                                                 // and 2p+1 in the halves of one register; P4 reads only the gathered ones (s < NVc)
    uint32_t sbLo[NW], sbHi[NW];                 // lane s holds the see-behind ballot of slot s
#pragma unroll
    for (int k = 0; k < NW; ++k) { sbLo[k] = 0; sbHi[k] = 0; }
    if (!MGX_DBG(4)) gather_all<V, NW, VPW, HALF>(a, wave, NVc, wall_addr, rec, inbLo, inbHi, lc, cell, sbLo, sbHi);
    if (ROLL) {                                                              // take the overlay off again: the tile persists
        wave_sync();
        if (ovl_off >= 0) store_cell16(tile + env_of_lane * HWB + ovl_off, ovl_saved);
    }

    MGX_MARK("P3");
    // ------------------------------------------------------------------ P3: lane s floods the visibility of slot s
    const bool masked = !a.sp.see_through_walls;                                // obs.py:95-100
    uint32_t visLo[NW], visHi[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) { visLo[k] = 0xffffffffu; visHi[k] = 0xffffffffu; }
    if (masked && !MGX_DBG(8)) {
        uint64_t sb[NW], vis[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) sb[k] = ((uint64_t)sbHi[k] << 32) | sbLo[k];
        constexpr int kOwn = (V - 1) * V + V / 2;                               // the agent's own cell: image[V/2][V-1]
        sb[kOwn >> 6] = (sb[kOwn >> 6] & ~(1ull << (kOwn & 63)))              // ... shows what it carries (obs.py:207)
                      | ((uint64_t)see_behind(my_carry) << (kOwn & 63));
        vis_mask<V, NW>(sb, vis);
#pragma unroll
        for (int k = 0; k < NW; ++k) { visLo[k] = (uint32_t)vis[k]; visHi[k] = (uint32_t)(vis[k] >> 32); }
    }

    MGX_MARK("P4");
    // slot s's packed cell in this lane (HALF: bits 16.. of an even slot's value are the odd slot's cell -- cell_unpack's masks
    // drop them)
    auto slot_cell = [&](const int s, const int it) -> uint32_t {
        if constexpr (HALF) return (s & 1) ? cell[s >> 1][it] >> 16 : cell[s >> 1][it];
        else return cell[s][it];
    };
    if constexpr (OH) {
        // -------------------------------------------------------------- P4'/P5' (one-hot output) in rounds of kRound slots:
        // P4' masks each cell and leaves its one-hot bit mask (bit t | bit 11+c | bit 17+s) at its image position in LDS;
        // P5' turns the masks of the 1-2 cells that each run of 16 output bytes spans into 0/1 bytes and streams them out.
        constexpr int D = 21;
        constexpr uint32_t kInvD = 0xFFFFFFFFu / D + 1u;                        // ceil(2^32 / 21): x / 21 exact for x < 2^20
        const int64_t h0 = tv0 * (int64_t)(V2 * D), h1 = h0 + (int64_t)NVc * V2 * D;
        const int oh_skew = (int)(h0 & 15);
        uint32_t *masks = reinterpret_cast<uint32_t *>(L + cv.out()) + 1;      // masks[-1] and masks[cells] are readable pads
        auto one_hot_mask = [](uint32_t c) -> uint32_t {                        // out-of-range values set no bit (as mgx_one_hot)
            const uint32_t p0 = min(c & 0xffu, 31u), p1 = min((c >> 8) & 0xffu, 31u), p2 = min((c >> 16) & 0xffu, 31u);
            return ((1u << p0) & 0x7ffu) | (((1u << p1) & 0x3fu) << 11) | (((1u << p2) & 0xfu) << 17);
        };
        if (lane == 0) masks[-1] = 0;
#pragma unroll
        for (int r0 = 0; r0 < VPW; r0 += kRound) {
            if (r0 < NVc) {
#pragma unroll
                for (int it = 0; it < NW; ++it) {
                    if (lc.act[it]) {
                        uint32_t *d0 = masks + (lc.q3[it] / 3);                  // q3 = 3 * (i*V + j)
#pragma unroll
                        for (int g0 = 0; g0 < kRound; g0 += kGroup) {
                            if (r0 + g0 < NVc) {
#pragma unroll
                                for (int sl = g0; sl < g0 + kGroup; ++sl) {
                                    const int s = r0 + sl;
                                    const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(visHi[it], s) << 32)
                                                     | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(visLo[it], s);
                                    const uint32_t c = cell_unpack(__builtin_amdgcn_inverse_ballot_w64(m) ? slot_cell(s, it) : CELL_UNSEEN);
                                    MGX_CHECK_LDS_PTR(7, d0 + sl * V2, 4);
                                    d0[sl * V2] = one_hot_mask(c);
                                }
                            }
                        }
                    }
                }
                wave_sync();
                if (lane >= r0 && lane < r0 + kRound && lane < NVc)             // own cell := carried object (obs.py:207)
                    masks[(lane - r0) * V2 + (V / 2) * V + (V - 1)] = one_hot_mask(my_carry);
                wave_sync();
                MGX_MARK("P5");
                const int64_t ro0 = h0 + (int64_t)r0 * (V2 * D);                // this round's output bytes [ro0, ro1)
                const int rbytes = (int)min((int64_t)kRound * V2 * D, h1 - ro0);
                const int rlen = oh_skew + rbytes;                              // from the aligned start
                uint8_t *gdst = MGX_LATE(obs) + (ro0 - oh_skew);
                const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(gdst, rlen);
                for (int rel = lane16; rel < rlen; rel += 1024) {
                    // output bytes [rel, rel + 16) of the aligned run = round bytes x .. x + 15, x = rel - skew (may be < 0:
                    // shifted by one cell so that the division stays in the positives; masks[-1] is a pad)
                    const uint32_t xs = (uint32_t)(rel - oh_skew + D);
                    const uint32_t cq = __umulhi(xs, kInvD);                    // cell + 1
                    const uint32_t k0 = xs - cq * D;                            // first bit inside that cell: 0..20
                    MGX_CHECK_LDS_PTR(8, masks + (int)cq - 1, 8);
                    const uint32_t bits = (masks[(int)cq - 1] >> k0) | (masks[(int)cq] << (D - k0));   // >= 22 valid bits
                    u32x4 v;
                    v.x = (((bits >> 0) & 0xfu) * 0x00204081u) & 0x01010101u;   // 4 bits -> 4 bytes of 0/1
                    v.y = (((bits >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
                    v.z = (((bits >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
                    v.w = (((bits >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
                    if ((rel + 16 <= rlen) & (rel >= oh_skew)) {
                        __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, rel, 0, MGX_OH_AUX);
                    } else {                                                    // ragged head / tail of the wave's bytes
                        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                        const int lo_b = max(rel, oh_skew), hi_b = min(rel + 16, rlen);
#pragma clang loop vectorize(disable) unroll(disable)
                        for (int B = lo_b; B < hi_b; ++B) gdst[B] = (uint8_t)(w[(B - rel) >> 2] >> (8 * ((B - rel) & 3)));
                    }
                }
                wave_sync();
                MGX_MARK("P5end");
            }
        }
    } else {
    // ------------------------------------------------------------------ P4/P5 in rounds of kRound slots:
    // P4 masks each cell and transposes it into the obs byte layout in LDS, P5 streams the round to HBM in 16-byte vectors
    const int64_t o0 = tv0 * (int64_t)(V2 * 3), o1 = o0 + (int64_t)NVc * V2 * 3;  // this wave's obs bytes (of step t)
    const int out_skew = (int)(o0 & 15);                                        // the same for every round
    uint8_t *out_raw = L + cv.out();                                          // obs bytes [oa_r, ...) of round r
    uint8_t *outb = out_raw + out_skew;
    constexpr int kRoundBytes = kRound * V2 * 3;                                // multiple of 16
#pragma unroll
    for (int r0 = 0; r0 < VPW; r0 += kRound) {
        if (r0 < NVc) {
            if (!MGX_DBG(16)) {
#pragma unroll
                for (int it = 0; it < NW; ++it) {
                    if (lc.act[it]) {
                        uint8_t *d0 = outb + lc.q3[it];
#if MGX_P4_B16
                        // A cell's 3 bytes go out as one ALIGNED 2-byte store + one byte store (an LDS store costs the same
                        // VGPR -> LDS transfer whatever its width, so 2 stores instead of 3).  A view is an odd number of
                        // bytes, so the parity of a cell's first byte alternates from slot to slot: even slots use [0], odd [1].
                        const uint32_t par0 = (uint32_t)(out_skew + lc.q3[it]) & 1u;
                        uint8_t *const d16[2] = {d0 + par0, d0 + (par0 ^ 1u)};                  // the 2-byte part: bytes 0-1 or 1-2
                        uint8_t *const d8[2] = {d0 + 2u * (par0 ^ 1u), d0 + 2u * par0};         // the other byte: 2 or 0
                        const uint32_t sh16[2] = {8u * par0, 8u * (par0 ^ 1u)}, sh8[2] = {16u * (par0 ^ 1u), 16u * par0};
#endif
                        // whole groups of kGroup slots, like P2 (padding slots write junk into staging space that P5
                        // never copies): straight-line code whose readlane -> select -> write chains overlap
#pragma unroll
                        for (int g0 = 0; g0 < kRound; g0 += kGroup) {
                            if (r0 + g0 < NVc) {
#pragma unroll
                                for (int sl = g0; sl < g0 + kGroup; ++sl) {
                                    const int s = r0 + sl;
                                    // (see_through_walls: the masks are all ones -- no branch, it would fence the schedule)
                                    const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(visHi[it], s) << 32)
                                                     | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(visLo[it], s);
                                    // packed cell -> the observation's (type, color, state) bytes
                                    const uint32_t c = cell_unpack(__builtin_amdgcn_inverse_ballot_w64(m) ? slot_cell(s, it) : CELL_UNSEEN);
                                    [[maybe_unused]] uint8_t *d = d0 + sl * (V2 * 3);
                                    MGX_CHECK_LDS_PTR(5, d, 3);
#if MGX_P4_B16
                                    *reinterpret_cast<uint16_t *>(d16[sl & 1] + sl * (V2 * 3)) = (uint16_t)(c >> sh16[sl & 1]);
                                    d8[sl & 1][sl * (V2 * 3)] = (uint8_t)(c >> sh8[sl & 1]);
#elif MGX_UA_WRITE
                                    *reinterpret_cast<u16_unaligned *>(d) = (uint16_t)c;    // ds_write_b16 at any byte address
                                    d[2] = (uint8_t)(c >> 16);                              // ds_write_b8_d16_hi
#else
                                    d[0] = (uint8_t)c; d[1] = (uint8_t)(c >> 8); d[2] = (uint8_t)(c >> 16);
#endif
                                }
                            }
                        }
                    }
                }
                wave_sync();
                // lane r0+sl: its agent's own cell shows the carried object (obs.py:207; always visible, obs.py:252)
                if (lane >= r0 && lane < r0 + kRound && lane < NVc)
                    store_obs_cell(outb + (lane - r0) * (V2 * 3) + ((V / 2) * V + (V - 1)) * 3, my_carry);
            }
            wave_sync();
            MGX_MARK("P5");
            if (!MGX_DBG(32)) {
                const int64_t ro0 = o0 + (int64_t)r0 * (V2 * 3);                // this round's obs bytes [ro0, ro1)
                const int rlen = out_skew + (int)min((int64_t)kRoundBytes, o1 - ro0);   // staged bytes, from the aligned start
                uint8_t *gdst = MGX_LATE(obs) + (ro0 - out_skew);
                const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(gdst, rlen);
                constexpr int kPasses = (kRoundBytes + 15 + 1023) / 1024;
#pragma unroll
                for (int k = 0; k < kPasses; ++k) {
                    const int rel = lane16 + 1024 * k;
                    if (rel < rlen) MGX_CHECK_LDS_PTR(6, out_raw + rel, 16);
                    if ((rel + 16 <= rlen) & (rel >= out_skew)) {
#if MGX_BUF_STORE
                        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(out_raw + rel), orsrc, rel, 0, MGX_OBS_AUX);
#else
                        *reinterpret_cast<u32x4 *>(gdst + rel) = *reinterpret_cast<const u32x4 *>(out_raw + rel);
#endif
                    } else if (rel < rlen) {                                    // ragged head / tail of the wave's bytes
                        const int lo_b = max(rel, out_skew), hi_b = min(rel + 16, rlen);
#pragma clang loop vectorize(disable) unroll(disable)
                        for (int B = lo_b; B < hi_b; ++B) gdst[B] = out_raw[B];
                    }
                }
            }
            wave_sync();
            MGX_MARK("P5end");
        }
    }
    }   // if !OH
    }   // for t

    if constexpr (GEN && MODE == 1) {
        // -------------------------------------------------------------- the envs whose episode ended with THIS step
        // (base.py:534-539 on the post-step state, which is still in LDS) start their next one here: Agent.reset + _gen_grid
        // by the env's lane, straight into the HBM state (every store of the step itself precedes it in program order)
        bool done_now = false;
        if (lane < Gc) {
            bool all_term = true;
            for (int k = 0; k < A; ++k) all_term &= row_term(rows[lane * A + k]);
            done_now = all_term | (scnt[lane] + 1 >= cf.max_steps);
            uint8_t *wr = MGX_LATE(was_reset);
            if (wr) wr[e0 + lane] = (uint8_t)done_now;
        }
        const uint64_t gmask = __builtin_amdgcn_ballot_w64(done_now);
        if (gmask != 0) {                                                        // rare
#define MGX_LATE_GEN(f) kernarg_at<decltype(MgxLayoutGen::f)>(offsetof(KernelArgs, gen) + offsetof(MgxLayoutGen, f))
            MgxLayoutGen gen;
            gen.kind = MGX_LATE_GEN(kind); gen.room_size = MGX_LATE_GEN(room_size);
            gen.start_x = MGX_LATE_GEN(start_x); gen.start_y = MGX_LATE_GEN(start_y); gen.start_dir = MGX_LATE_GEN(start_dir);
            gen.blank = MGX_LATE_GEN(blank); gen.gen_state = MGX_LATE_GEN(gen_state);
#undef MGX_LATE_GEN
            mgx_gen::copy_blank(gen, MGX_LATE(grid), e0, HWB, gmask, lane);
            if (done_now) {
                const int64_t b = e0 + lane;
                mgx_gen::NpGen lay, npr;
                uint64_t *gs = gen.gen_state + b * 6, *rg = MGX_LATE(rng) + b * 4;
                for (int k = 0; k < 4; ++k) {
                    lay.s[k] = gs[k];
                    // (this launch advanced the env's PCG64 words in P0: read them past the L1)
                    npr.s[k] = __hip_atomic_load(rg + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                lay.buf = gs[4]; npr.buf = gs[5];
                const uint4 naux = mgx_gen::generate_episode(gen, W, H, A, lay, npr, L + cv.rec() + lane * (2 * A),
                                                             MGX_LATE(grid) + b * HWB,
                                                             reinterpret_cast<uint64_t *>(MGX_LATE(agents)) + b * A);
                if (HOOKS) reinterpret_cast<uint4 *>(MGX_LATE(aux))[b] = naux;
                for (int k = 0; k < 4; ++k) { gs[k] = lay.s[k]; rg[k] = npr.s[k]; }
                gs[4] = lay.buf; gs[5] = npr.buf;
                MGX_LATE(step_count)[b] = 0;                                     // base.py:292
                MGX_LATE(episode)[b] += 1;
            }
        }
    }

#if MGX_TIMESTAMPS
    __builtin_amdgcn_s_waitcnt(0);
    MGX_MARK("end");
    if (lane == 0 && wid < 16384) g_span[2 * wid + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (ROLL) {
        // ------------------------------------------------------------------ state write-back, once per launch
        for (int rel = 16 * lane; rel < len; rel += 16 * 64) {                  // the tile, as it was loaded
            const int64_t D = ga + rel;
            if (D >= g0 && D + 16 <= g1) {
                *reinterpret_cast<uint4 *>(a.grid + D) = *reinterpret_cast<const uint4 *>(tile_raw + rel);
            } else {
                const int64_t lo_b = max(D, g0), hi_b = min(D + 16, g1);
                for (int64_t B = lo_b; B < hi_b; ++B) a.grid[B] = tile_raw[(int)(B - ga)];
            }
        }
        if (lane < NVc) reinterpret_cast<uint64_t *>(a.agents)[v0 + lane] = rows[lane];
        if (A > 1) {
            if (lane < Gc * 4 && (lane & 3) < 2) a.rng[e0 * 4 + lane] = rngs[lane];
            if (lane + 64 < Gc * 4 && (lane & 3) < 2) a.rng[e0 * 4 + lane + 64] = rngs[lane + 64];
        }
        if (lane < Gc) {
            a.step_count[e0 + lane] = scnt[lane];
            if (HOOKS && (AR || env_kind >= MGX_KIND_REDBLUEDOORS)) reinterpret_cast<uint4 *>(a.aux)[e0 + lane] = auxl[lane];
        }
    }
}

// The kernel instantiation for (V, mode, hooks, auto-reset) and its launch.  `hip_err` receives the HIP error code of a
// failed launch (mgx_last_hip_error).
template <int V, int MODE, bool OH, bool GEN = false, bool STREAM = false, bool DMA = false>
inline int launch_mode(const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream, int *hip_err) {
    if constexpr (!STREAM && !DMA && MODE != 2 && !GEN) {    // (rollouts read the tile once per launch; GEN: small envs)
        if (ka.flags & 1) return launch_mode<V, MODE, OH, GEN, true, false>(ka, threads, lds_bytes, nwg, stream, hip_err);
        if constexpr (!OH) {
            if (ka.flags & 2) return launch_mode<V, MODE, OH, GEN, false, true>(ka, threads, lds_bytes, nwg, stream, hip_err);
        }
    }
    void (*kern)(const KernelArgs) = nullptr;
    const bool hooks = MODE != 0 && ka.sp.env_kind != MGX_KIND_EMPTY;        // (gen_obs never runs a hook)
    const bool ar = MODE != 0 && ka.pool_grid != nullptr;
    constexpr bool S = MODE != 0;
    if constexpr (GEN) {                                                     // (generation replaces the pool pick-up)
        kern = hooks ? mgx_fused_kernel<V, MODE, S, false, OH, true> : mgx_fused_kernel<V, MODE, false, false, OH, true>;
    } else {
        kern = hooks ? (ar ? mgx_fused_kernel<V, MODE, S, S, OH, false, STREAM, DMA> : mgx_fused_kernel<V, MODE, S, false, OH, false, STREAM, DMA>)
                     : (ar ? mgx_fused_kernel<V, MODE, false, S, OH, false, STREAM, DMA> : mgx_fused_kernel<V, MODE, false, false, OH, false, STREAM, DMA>);
    }
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) { *hip_err = (int)e; return MGX_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(threads), (size_t)lds_bytes, stream, ka);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { *hip_err = (int)e; return MGX_ERR_LAUNCH; }
    return MGX_OK;
}

template <int V>
inline int launch_view(int mode, const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream,
                       int *hip_err) {
    switch (mode) {                     // mode | 4: one-hot observations (gen_obs and one step only); | 8: tail generation
    case 0: return launch_mode<V, 0, false>(ka, threads, lds_bytes, nwg, stream, hip_err);
    case 1: return launch_mode<V, 1, false>(ka, threads, lds_bytes, nwg, stream, hip_err);
    case 2: return launch_mode<V, 2, false>(ka, threads, lds_bytes, nwg, stream, hip_err);
    case 4: return launch_mode<V, 0, true>(ka, threads, lds_bytes, nwg, stream, hip_err);
    case 5: return launch_mode<V, 1, true>(ka, threads, lds_bytes, nwg, stream, hip_err);
    case 9: return launch_mode<V, 1, false, true>(ka, threads, lds_bytes, nwg, stream, hip_err);   // | 8: generate
    default: return MGX_ERR_INVALID_ARGUMENT;
    }
}

// One translation unit per view size (mgx_fused_inst.hip, -DMGX_INST_V=<V>) defines its launcher:
#define MGX_FOR_EACH_VIEW(X) X(3) X(5) X(7) X(9) X(11) X(13) X(15)
#define MGX_DECLARE_LAUNCHER(V) \
    int launch_v##V(int mode, const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream, int *hip_err);
MGX_FOR_EACH_VIEW(MGX_DECLARE_LAUNCHER)
#undef MGX_DECLARE_LAUNCHER

}  // namespace mgx_fused
