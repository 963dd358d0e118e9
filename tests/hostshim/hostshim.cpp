// hostshim.cpp -- TEST INFRASTRUCTURE.  Compiles multigrid_amd/csrc/mgx_rules.h (the per-env / per-view integer
// rules the HIP kernels are built from) for the host with g++, so the `-m "not gpu"` suite can check them against
// the oracle without a GPU.  This is not a product path: libmgx.so never runs these on the CPU.
//
// The functions below replay, lane by lane in scalar code, what the fused kernel's phases do for ONE env:
//   shim_step_env : P1a (jump-ahead PCG64 draws) + P1b (rank -> order) + P1c (handle_actions, overlay, hook)
//   shim_obs_env  : P1d (view geometry, in-bounds mask) + P2 (gather, see-behind bits) + P3 (visibility flood)
//                   + P4 (mask) + byte packing
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../multigrid_amd/csrc/mgx_rules.h"

using namespace mgx;

static const JumpTable kJump{};

extern "C" int shim_step_env(const MgxSpec *sp, uint8_t *tile /* H*W packed cells, updated, NOT overlaid */,
                             uint8_t *tile_overlaid /* out: tile with agent overlay (render input) */,
                             uint64_t *rows /* A, updated */, const int8_t *act, uint64_t *rng /* 4, updated */,
                             int32_t *step_count, uint8_t *aux /* 16, updated */, double *rew /* A out */,
                             uint8_t *terminated /* A out */, uint8_t *truncated, uint8_t *order_out,
                             int32_t *n_dirty, int force_serial, const uint8_t *hook_order /* A, or NULL */) {
    const StepCfg cf = make_cfg(*sp);
    const int A = cf.A, HWB = cf.H * cf.W * cf.cb;          // (cf.cb: 2 = MgxCell, 1 = the compact MgxCell8; include/mgx.h)
    std::vector<uint64_t> rnd(A);
    std::vector<uint8_t> ord(A, 0);
    for (int a = 0; a < A; ++a) rew[a] = 0.0;
    if (A > 1) {
        uint64_t s_lo = 0, s_hi = 0;
        for (int ai = 0; ai < A; ++ai) rnd[ai] = pcg64_draw_at(rng, kJump.w[ai + 1], s_lo, s_hi);
        rng[0] = s_lo; rng[1] = s_hi;                       // lane A-1 holds the state after A steps
        for (int ai = 0; ai < A; ++ai) ord[draw_rank(rnd.data(), A, ai)] = (uint8_t)ai;
    }
    for (int a = 0; a < A; ++a) order_out[a] = ord[a];
    const int32_t sc = *step_count + 1;
    *step_count = sc;
    int nd = 0;
    auto dirty = [&](int) { ++nd; };
    int rc = 0;
    // the kernel's order-free fast path (P1s), lane by lane; envs that fail its conditions take the sequential loop
    std::vector<AgentEval> ev(A);
    std::vector<int32_t> woff(A);
    uint64_t m_bad = 0, m_conf = 0, m_pres = 0, m_moved = 0, m_ends_all = 0;
    for (int ai = 0; ai < A; ++ai) {
        ev[ai] = eval_agent(cf, tile, rows, act[ai], rows[ai], true, stale_offset(cf, aux, sp->env_kind));
        woff[ai] = ev[ai].writes ? ev[ai].off : -1;
    }
    for (int ai = 0; ai < A; ++ai) {
        if (ev[ai].bad) m_bad |= 1ull << ai;
        if (event_ends_all(cf, ev[ai])) m_ends_all |= 1ull << ai;
        if (spec_cell_conflict(woff.data(), A, ai, ev[ai])) m_conf |= 1ull << ai;
        if (ev[ai].used_presence) m_pres |= 1ull << ai;
        if (ev[ai].moved) m_moved |= 1ull << ai;
    }
    const bool fallback = spec_needs_fallback(m_bad, m_conf, m_pres, m_moved) || force_serial;
    if (!fallback) {
        // lane by lane, as the kernel: who acts (rank <= cutoff), commit, then the events of the agents that acted
        const int cut = (m_ends_all != 0 && A > 1) ? event_cutoff(ord.data(), m_ends_all, A) : A;
        std::vector<bool> acts(A);
        bool joint_success = false;
        for (int ai = 0; ai < A; ++ai) {
            const int rank = (A > 1) ? draw_rank(rnd.data(), A, ai) : 0;
            acts[ai] = rank <= cut;
            joint_success |= acts[ai] & ev[ai].success;
        }
        const double r = reward_value(sc, cf.max_steps);
        for (int ai = 0; ai < A; ++ai) {
            if (acts[ai]) {
                if (ev[ai].go) rows[ai] = ev[ai].nrow;
                if (ev[ai].unstale) aux[4] = 0;
                if (ev[ai].writes) { store_cell(cf.cb, tile + ev[ai].off, ev[ai].ncell); dirty(ev[ai].off); }
            }
            if (cf.joint_reward ? joint_success : (acts[ai] & ev[ai].success)) rew[ai] = r;
            if ((acts[ai] & event_ends_self(cf, ev[ai])) | (m_ends_all != 0))
                reinterpret_cast<uint8_t *>(rows)[ai * MGX_AGENT_STRIDE + AG_TERM] = 1;
        }
    } else if (force_serial || A == 1) {
        rc = handle_actions(cf, tile, rows, act, ord.data(), rew, sc, dirty, aux, sp->env_kind);
    } else {
        // the kernel's shortened fallback: the agents ranked below the first blocked one commit with the order-free results, the
        // reference's loop starts at that cutoff (mgx_rules.h: prefix_blocked)
        bool any_writes = false;
        for (int ai = 0; ai < A; ++ai) any_writes |= ev[ai].writes;
        int cut = A;
        for (int ai = 0; ai < A; ++ai) {
            const int rank = draw_rank(rnd.data(), A, ai);
            if (prefix_blocked(ev[ai], rank, ord.data(), woff.data(), any_writes, m_moved, A) && rank < cut) cut = rank;
        }
        for (int k = 0; k < cut; ++k) {
            const int ai = ord[k];
            if (ev[ai].go) rows[ai] = ev[ai].nrow;
            if (ev[ai].unstale) aux[4] = 0;
            if (ev[ai].writes) { store_cell(cf.cb, tile + ev[ai].off, ev[ai].ncell); dirty(ev[ai].off); }
        }
        rc = handle_actions(cf, tile, rows, act, ord.data(), rew, sc, dirty, aux, sp->env_kind, cut);
    }
    *n_dirty = fallback ? -nd - 1 : nd;          // negative = the sequential loop ran
    // the kernel's per-agent overlay (one lane per agent): offsets from the PRE-hook rows, cells written after the hook
    std::vector<int> ovl(A);
    std::vector<uint64_t> pre_rows(rows, rows + A);
    for (int ai = 0; ai < A; ++ai) ovl[ai] = overlay_offset(cf, rows, ai);

    for (int a = 0; a < A; ++a) terminated[a] = 0;
    post_step_hook(cf, sp->env_kind, tile, rows, act, aux, sc, rew, dirty, hook_order);   // on the clean tile, like the kernel
    const bool forced = sp->env_kind == MGX_KIND_LOCKEDHALLWAY && aux[15];
    for (int a = 0; a < A; ++a) terminated[a] = (uint8_t)(row_term(rows[a]) | forced);
    std::memcpy(tile_overlaid, tile, HWB);
    for (int ai = 0; ai < A; ++ai)
        if (ovl[ai] >= 0) store_cell_raw(cf.cb, tile_overlaid + ovl[ai], agent_cell_raw(cf.cb, rows[ai]));
    {   // must equal the reference's ascending loop (overlay_agents) run on the pre-hook rows
        std::vector<uint8_t> ref(tile, tile + HWB);
        overlay_agents(cf, ref.data(), pre_rows.data());
        if (std::memcmp(ref.data(), tile_overlaid, HWB) != 0) return -99;
    }
    *truncated = (uint8_t)(sc >= cf.max_steps);
    return rc;
}

template <int V>
static void obs_env(const MgxSpec *sp, const uint8_t *tile, const uint64_t *rows, uint8_t *obs) {
    constexpr int V2 = V * V, NW = (V2 + 63) / 64;
    const int A = sp->num_agents, W = sp->width, H = sp->height, cb = make_cfg(*sp).cb;
    for (int a = 0; a < A; ++a) {
        const uint64_t row = rows[a];
        const ViewGeom g = view_geom<V>(W, H, row_x(row), row_y(row), row_dir(row), cb);
        uint64_t inb[NW], sb[NW], vis[NW];
        inbounds_mask<V, NW>(g, inb);
        const ViewClamp vc = view_clamp<V>(g, W, H, row_x(row), row_y(row));   // what the kernel's P2 gathers with
        uint32_t cells[V2];
        for (int k = 0; k < NW; ++k) sb[k] = 0;
        for (int k = 0; k < V2; ++k) {                       // "lane" k
            const int j = k / V, i = k - j * V, la = i - V / 2, fw = V - 1 - j;
            const bool in = (inb[k >> 6] >> (k & 63)) & 1;
            // (the kernel: the clamped offset, or the WALL cell for an agent outside the grid; must agree with the mask form)
            // (what a cell SHOWS: a box's content is not part of it -- the kernel's P4 masks the same bits)
            uint32_t c = vc.valid ? load_cell_shown(cb, tile + clamped_offset(g.origin, vc, fw, la)) : (uint32_t)CELL_WALL;
            const uint32_t c_mask = in ? load_cell_shown(cb, tile + g.origin + fw * g.stepF + la * g.stepL) : (uint32_t)CELL_WALL;
            if (c != c_mask) std::abort();
            if (vc.valid) {      // the gather's see-behind test is the raw cell's opaque bit (sign of the 16 bits / of the byte)
                const uint32_t raw = load_cell_raw(cb, tile + clamped_offset(g.origin, vc, fw, la));
                if ((((raw >> (cb == 1 ? 7 : 15)) & 1u) != 0) == see_behind(c)) std::abort();
            }
            if (i == V / 2 && j == V - 1) c = row_carry(row) & kCellShown;
            if (see_behind(c)) sb[k >> 6] |= 1ull << (k & 63);
            cells[i * V + j] = c;
        }
        if (!sp->see_through_walls) {
            vis_mask<V, NW>(sb, vis);
            for (int k = 0; k < V2; ++k) {
                const int j = k / V, i = k - j * V;
                if (!((vis[k >> 6] >> (k & 63)) & 1)) cells[i * V + j] = CELL_UNSEEN;
            }
        }
        for (int qq = 0; qq < V2; ++qq) store_obs_cell(obs + ((size_t)a * V2 + qq) * 3, cells[qq]);
    }
}

extern "C" int shim_obs_env(const MgxSpec *sp, const uint8_t *tile_overlaid, const uint64_t *rows, uint8_t *obs) {
    switch (sp->view_size) {
    case 3: obs_env<3>(sp, tile_overlaid, rows, obs); break;
    case 5: obs_env<5>(sp, tile_overlaid, rows, obs); break;
    case 7: obs_env<7>(sp, tile_overlaid, rows, obs); break;
    case 9: obs_env<9>(sp, tile_overlaid, rows, obs); break;
    case 11: obs_env<11>(sp, tile_overlaid, rows, obs); break;
    case 13: obs_env<13>(sp, tile_overlaid, rows, obs); break;
    case 15: obs_env<15>(sp, tile_overlaid, rows, obs); break;
    default: return -1;
    }
    return 0;
}

// gen_obs for a state that has not been overlaid yet (reset observation)
extern "C" int shim_overlay(const MgxSpec *sp, uint8_t *tile, const uint64_t *rows) {
    overlay_agents(make_cfg(*sp), tile, rows);
    return 0;
}

// the restart's layout index (mgx_rules.h: pool_index -- exact remainders by multiplication) for the test against big-int Python
extern "C" int shim_pool_index(int64_t first_env, int64_t b, int32_t ep, int32_t K) {
    const uint64_t M = K > 1 ? ~0ull / (uint64_t)K + 1ull : 0ull;
    return pool_index(first_env, b, ep, K, M);
}


// the staged generator's fast-forward (mgx_rules.h: pcg64_advance): state[0..1] after n units of k draws each, inc = state[2..3]
extern "C" void shim_pcg64_advance(uint64_t *state4, int k, uint32_t n) {
    const uint64_t unit[4] = {kJump.w[k][0], kJump.w[k][1], kJump.w[k][2], kJump.w[k][3]};
    pcg64_advance(state4[0], state4[1], state4[2], state4[3], unit, n);
}
