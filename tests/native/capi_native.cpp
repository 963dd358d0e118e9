// capi_native.cpp -- TEST INFRASTRUCTURE.  Drives libmgx.so through its C ABI (include/mgx.h) from plain C++ with the HIP
// runtime only -- no Python, no torch -- the way a non-Python host would bind it, and checks every output and the
// post-step state of mgx_step / mgx_gen_obs against the CPU oracle (oracle/mgx_oracle.c, linked in as the checker).
// Built by tests/test_capi_native.py (hipcc); prints "native capi ok" and exits 0 on success.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mgx.h"

// the oracle's batch entry points (oracle/mgx_oracle.c); its spec struct has the layout of MgxSpec
extern "C" int mgo_step_batch(const MgxSpec *sp, int64_t B, uint8_t *grid, uint8_t *agents, uint64_t *rng,
                              int32_t *step_count, const int8_t *actions, const uint8_t *target, uint8_t *obs, uint8_t *dir,
                              double *reward, uint8_t *terminated, uint8_t *truncated, int64_t *err_env, int nthreads,
                              const uint8_t *hook_order);
extern "C" int mgo_gen_obs_batch(const MgxSpec *sp, int64_t B, const uint8_t *grid, const uint8_t *agents, uint8_t *obs,
                                 uint8_t *dir, int nthreads);

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MGX_OK_(x) do { int r_ = (x); if (r_ != MGX_OK) { std::fprintf(stderr, "%s: %s\n", #x, mgx_error_string(r_)); return 3; } } while (0)

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (uint32_t)(g_s >> 32); }

template <typename T> struct Dev {
    T *p = nullptr; size_t n = 0;
    int alloc(size_t count) { n = count; return (int)hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T) + 16); }
    int up(const std::vector<T> &h) { return (int)hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice); }
    int down(std::vector<T> &h) const { h.resize(n); return (int)hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost); }
};

int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? std::atoll(argv[1]) : 3001;
    const int T = argc > 2 ? std::atoi(argv[2]) : 12;
    MgxSpec sp{};
    sp.width = 13; sp.height = 11; sp.num_agents = 3; sp.view_size = 7; sp.max_steps = 9; sp.see_through_walls = 0;
    sp.allow_agent_overlap = 1; sp.joint_reward = 0; sp.success_any = 1; sp.failure_any = 0; sp.env_kind = MGX_KIND_EMPTY;
    if (mgx_abi_version() != MGX_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    const int W = sp.width, H = sp.height, A = sp.num_agents, V = sp.view_size;
    const size_t gsz = (size_t)W * H * 3, osz = (size_t)A * V * V * 3;
    // random walled grids with walls, goals, lava, doors, keys, balls; agents on empty cells
    std::vector<uint8_t> grid(B * gsz), agents((size_t)B * A * 8, 0);
    std::vector<uint64_t> rng((size_t)B * 4);
    std::vector<int32_t> sc(B, 0);
    static const uint8_t kinds[8][3] = {{2, 5, 0}, {8, 1, 0}, {9, 0, 0}, {4, 2, 1}, {4, 3, 0}, {5, 1, 0}, {6, 2, 0}, {3, 4, 0}};
    for (int64_t b = 0; b < B; ++b) {
        uint8_t *g = &grid[b * gsz];
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
            uint8_t *c = g + (y * W + x) * 3;
            const bool border = x == 0 || y == 0 || x == W - 1 || y == H - 1;
            if (border) { c[0] = 2; c[1] = 5; c[2] = 0; }
            else if (rnd() % 100 < 22) { const uint8_t *k = kinds[rnd() % 8]; c[0] = k[0]; c[1] = k[1]; c[2] = k[2]; }
            else { c[0] = 1; c[1] = 0; c[2] = 0; }
        }
        for (int a = 0; a < A; ++a) {
            int x, y;
            do { x = 1 + rnd() % (W - 2); y = 1 + rnd() % (H - 2); } while (g[(y * W + x) * 3] != 1);
            uint8_t *r = &agents[((size_t)b * A + a) * 8];
            r[0] = (uint8_t)(a % 6); r[1] = (uint8_t)(rnd() % 4); r[2] = (uint8_t)x; r[3] = (uint8_t)y; r[4] = 0; r[5] = 1; r[6] = 0; r[7] = 0;
        }
        for (int k = 0; k < 4; ++k) rng[b * 4 + k] = ((uint64_t)rnd() << 32) | rnd();
        rng[b * 4 + 2] |= 1;                                         // PCG64 increments are odd
    }
    Dev<uint8_t> d_cells3, d_agents, d_obs, d_dir, d_term, d_trunc; Dev<uint64_t> d_rng; Dev<int32_t> d_sc, d_err; Dev<int8_t> d_act;
    Dev<double> d_rew; Dev<MgxCell> d_grid;       // the device holds packed cells (include/mgx.h)
    HIP_OK((hipError_t)d_cells3.alloc(B * gsz)); HIP_OK((hipError_t)d_grid.alloc((size_t)B * W * H)); HIP_OK((hipError_t)d_agents.alloc((size_t)B * A * 8)); HIP_OK((hipError_t)d_rng.alloc((size_t)B * 4));
    HIP_OK((hipError_t)d_sc.alloc(B)); HIP_OK((hipError_t)d_err.alloc(2)); HIP_OK((hipError_t)d_act.alloc((size_t)B * A));
    HIP_OK((hipError_t)d_obs.alloc(B * osz)); HIP_OK((hipError_t)d_dir.alloc((size_t)B * A)); HIP_OK((hipError_t)d_rew.alloc((size_t)B * A));
    HIP_OK((hipError_t)d_term.alloc((size_t)B * A)); HIP_OK((hipError_t)d_trunc.alloc(B));
    HIP_OK((hipError_t)d_cells3.up(grid)); HIP_OK((hipError_t)d_agents.up(agents)); HIP_OK((hipError_t)d_rng.up(rng)); HIP_OK((hipError_t)d_sc.up(sc));
    std::vector<int32_t> err0 = {0, INT32_MAX}; HIP_OK((hipError_t)d_err.up(err0));
    hipStream_t stream; HIP_OK(hipStreamCreate(&stream));
    // the reference's (type, color, state) triples -> packed cells, on the device; err[0] doubles as the bad-cell counter
    MGX_OK_(mgx_pack_grid(d_cells3.p, B * (int64_t)W * H, d_grid.p, d_err.p, stream));
    HIP_OK(hipStreamSynchronize(stream));
    { std::vector<int32_t> e; HIP_OK((hipError_t)d_err.down(e)); if (e[0] != 0) { std::fprintf(stderr, "unpackable cells\n"); return 9; } }

    std::vector<uint8_t> h_obs, h_dir, h_term, h_trunc, h_grid, h_agents, o_obs(B * osz), o_dir((size_t)B * A), o_term((size_t)B * A), o_trunc(B);
    std::vector<double> h_rew, o_rew((size_t)B * A); std::vector<uint64_t> h_rng; std::vector<int32_t> h_sc; std::vector<int8_t> act((size_t)B * A);
    // gen_obs of the initial state
    MGX_OK_(mgx_gen_obs(&sp, B, d_grid.p, d_agents.p, d_obs.p, d_dir.p, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK((hipError_t)d_obs.down(h_obs)); HIP_OK((hipError_t)d_dir.down(h_dir));
    if (mgo_gen_obs_batch(&sp, B, grid.data(), agents.data(), o_obs.data(), o_dir.data(), 8)) return 4;
    if (h_obs != o_obs || h_dir != o_dir) { std::fprintf(stderr, "gen_obs differs from the oracle\n"); return 5; }
    for (int t = 0; t < T; ++t) {
        for (auto &a : act) { const uint32_t r = rnd() % 16; a = (int8_t)(r < 7 ? r : (r == 15 ? -1 : 2)); }   // forward-heavy, some absent
        HIP_OK((hipError_t)d_act.up(act));
        MGX_OK_(mgx_step(&sp, B, d_grid.p, d_agents.p, d_rng.p, d_sc.p, d_act.p, nullptr, d_obs.p, d_dir.p, d_rew.p, d_term.p,
                         d_trunc.p, d_err.p, stream));
        HIP_OK(hipStreamSynchronize(stream));
        int64_t bad = -1;
        if (mgo_step_batch(&sp, B, grid.data(), agents.data(), rng.data(), sc.data(), act.data(), nullptr, o_obs.data(), o_dir.data(),
                           o_rew.data(), o_term.data(), o_trunc.data(), &bad, 8, nullptr)) return 6;
        HIP_OK((hipError_t)d_obs.down(h_obs)); HIP_OK((hipError_t)d_dir.down(h_dir)); HIP_OK((hipError_t)d_rew.down(h_rew)); HIP_OK((hipError_t)d_term.down(h_term));
        MGX_OK_(mgx_unpack_grid(d_grid.p, B * (int64_t)W * H, d_cells3.p, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK((hipError_t)d_trunc.down(h_trunc)); HIP_OK((hipError_t)d_cells3.down(h_grid)); HIP_OK((hipError_t)d_agents.down(h_agents));
        HIP_OK((hipError_t)d_rng.down(h_rng)); HIP_OK((hipError_t)d_sc.down(h_sc));
        const bool same = h_obs == o_obs && h_dir == o_dir && h_term == o_term && h_trunc == o_trunc && h_grid == grid
                       && h_agents == agents && h_rng == rng && h_sc == sc
                       && std::memcmp(h_rew.data(), o_rew.data(), o_rew.size() * sizeof(double)) == 0;
        if (!same) { std::fprintf(stderr, "step %d differs from the oracle\n", t); return 7; }
    }
    std::vector<int32_t> h_err; HIP_OK((hipError_t)d_err.down(h_err));
    if (h_err[0] != 0) { std::fprintf(stderr, "unexpected unknown-action report\n"); return 8; }
    std::printf("native capi ok: %lld envs x %d steps bit-exact vs the oracle through the C ABI (v%d)\n", (long long)B, T, mgx_abi_version());
    return 0;
}
