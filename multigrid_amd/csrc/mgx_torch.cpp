// mgx_torch.cpp -- the compiled PyTorch-ROCm operator library over the C ABI of libmgx.so (include/mgx.h):
//     TORCH_LIBRARY(mgx, ...)  +  TORCH_LIBRARY_IMPL(mgx, CUDA, ...)        (CUDA dispatch key = HIP on ROCm)
// built in-tree as multigrid_amd/lib/libmgx_torch.so (multigrid_amd/build.py: build_torch_lib) and loaded by
// multigrid_amd/ops.py with torch.ops.load_library -- or by any C++ program that links libtorch: the ops are in the dispatcher
// without a Python interpreter.  No device code here: every op validates its tensors, allocates its outputs and calls ONE entry
// point of libmgx.so on torch's current HIP stream of the tensors' device.  Only the CUDA key is registered: CPU tensors raise
// from the dispatcher (there is no CPU implementation to fall back to).
//
// The seam these ops replace in the reference (ini/multigrid): multigrid/base.py:361-366 (gen_obs_grid_encoding call) and
// multigrid/base.py:333-340 (the body of MultiGridEnv.step); wrappers: multigrid/wrappers.py:48-58, 158-190.
//
// `grid` / `pool_grid` are packed cells (int16 [B,H,W], MgxCell bit patterns).  The same ops accept the reference's form,
// (type, color, state) bytes uint8 [B,H,W,3]: packed on the way in (mgx_pack_grid) and, for the mutating ops, unpacked back into
// the caller's tensor on the way out (mgx_unpack_grid) -- two more streaming kernels, NO host synchronisation.  What such a grid
// may get wrong -- a value the packed format cannot hold, or an outer ring that is not the reference's WALL (the precondition of
// include/mgx.h) -- is counted by the pack kernel and reported DEFERRED, the way the device reports its own faults: the counts
// travel to pinned host memory behind the pack, and the next mgx op on that device (or torch.ops.mgx.check_errors(), which
// waits) raises once they have arrived.  State that is already packed is checked on request: torch.ops.mgx.check_grid.
//
// Out-variants (`step_out`, `step_autoreset_out`, `step_one_hot_out`, `gen_obs_out`): the caller owns the output tensors -- no
// allocation per call (five at::empty were most of the op's host time), same checks, same launch.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#include <torch/library.h>

#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <tuple>
#include <vector>

#include "../../include/mgx.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

MgxSpec spec_from(at::IntArrayRef v) {
    TORCH_CHECK(v.size() == 11 || v.size() == 12, "mgx: spec must have 11 ints (struct MgxSpec, include/mgx.h; a 12th = cell_bytes), got ",
                v.size());
    MgxSpec s{};
    // (the operator library works on the 16-bit MgxCell tensors; the compact one-byte format is served by the C ABI itself)
    TORCH_CHECK(v.size() == 11 || v[11] == 0 || v[11] == MGX_CELL_BYTES, "mgx: the torch ops take 16-bit cells (cell_bytes 0 or 2), got ", v[11]);
    s.width = (int32_t)v[0]; s.height = (int32_t)v[1]; s.num_agents = (int32_t)v[2]; s.view_size = (int32_t)v[3];
    s.max_steps = (int32_t)v[4]; s.see_through_walls = (int32_t)v[5]; s.allow_agent_overlap = (int32_t)v[6];
    s.joint_reward = (int32_t)v[7]; s.success_any = (int32_t)v[8]; s.failure_any = (int32_t)v[9]; s.env_kind = (int32_t)v[10];
    return s;
}

// every tensor of a call lives on ONE device: the first one checked sets it (DeviceGuard below), the others must match -- a
// mix of cuda:0 and cuda:1 tensors would otherwise end in an illegal memory access inside the kernel
thread_local c10::Device t_device(c10::DeviceType::CPU);
thread_local bool t_device_set = false;

void want(const Tensor &t, const char *name, at::ScalarType dtype, at::IntArrayRef shape = {}, bool check_shape = false) {
    TORCH_CHECK(t.is_cuda(), "mgx: `", name, "` must live on a HIP device (got ", t.device(), "); there is no CPU path");
    TORCH_CHECK(!t_device_set || t.device() == t_device, "mgx: `", name, "` is on ", t.device(), " but the call's grid is on ", t_device,
                ": every tensor of a call must live on one device");
    TORCH_CHECK_TYPE(t.scalar_type() == dtype, "mgx: `", name, "` must be ", dtype, ", got ", t.scalar_type());
    TORCH_CHECK_VALUE(t.is_contiguous(), "mgx: `", name, "` must be contiguous");
    if (check_shape) TORCH_CHECK_VALUE(t.sizes() == shape, "mgx: `", name, "` must have shape ", shape, ", got ", t.sizes());
}

void check(int rc, const char *what) {
    TORCH_CHECK(rc == MGX_OK, what, ": ", mgx_error_string(rc), " (code ", rc, ", hip error ", mgx_last_hip_error(), ")");
}

// (a failed HIP call stays HIP's sticky "last error" until somebody reads it, and torch reads it after its own launches: take it off
// before raising, or the caller's next torch op reports OUR failure a second time, as its own)
bool hip_ok(hipError_t e) {
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

void *stream_of(const Tensor &t) { return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

struct DeviceGuard {
    c10::hip::HIPGuardMasqueradingAsCUDA g;
    explicit DeviceGuard(const Tensor &t) : g(t.device()) { t_device = t.device(); t_device_set = true; }
    ~DeviceGuard() { t_device_set = false; }
};

// ---- deferred report of what a byte grid got wrong (see the header comment) ------------------------------------------------
struct GridFaults {
    int32_t *dev = nullptr;        // i32[2] on the device: [0] unpackable values, [1] ring cells that are not WALL (accumulating)
    int32_t *host = nullptr;       // pinned copy
    hipEvent_t ev = nullptr;       // recorded behind the copy
    bool pending = false;
    int unreported = 0;            // fused byte-grid steps (the kernel counts into `dev`) since the last queued report: the copy to the
                                   // host is queued with the FIRST step of every grid tensor (a bad grid is reported by the call after
                                   // the one that was handed it, not 31 steps later: ADVICE r5) and then every 32nd (it is a DMA op
                                   // on the step's stream), and by check_errors(), which waits
    const void *last_grid = nullptr;   // the grid tensor of the previous fused byte-grid step on this device
};
std::mutex g_faults_mutex;
std::map<int, GridFaults> g_faults;

void raise_faults(GridFaults &f, void *stream) {
    const int32_t unpackable = f.host[0], ring = f.host[1];
    f.pending = false;
    if (unpackable == 0 && ring == 0) return;
    f.host[0] = f.host[1] = 0;
    (void)hipMemsetAsync(f.dev, 0, 8, (hipStream_t)stream);
    TORCH_CHECK(false, "mgx: an earlier call was handed a byte grid with ", ring, " outer-ring cell(s) that are not WALL = (wall, grey, 0) "
                "and ", unpackable, " cell value(s) the packed format cannot hold (type > 15, color > 7 or state > 3); its results are "
                "undefined.  The outer ring of every env's grid must be the reference's WALL (include/mgx.h; every _gen_grid of the "
                "reference starts from Grid.wall_rect(0, 0, W, H))");
}

// before a call: has an earlier byte grid's report arrived?  (never blocks; `wait`: torch.ops.mgx.check_errors)
void poll_faults(int device, void *stream, bool wait = false) {
    std::lock_guard<std::mutex> lock(g_faults_mutex);
    auto it = g_faults.find(device);
    if (it == g_faults.end()) return;
    GridFaults &f = it->second;
    if (wait && f.unreported > 0) {                     // counts of fused byte-grid steps that have not been sent for yet
        TORCH_CHECK(hip_ok(hipMemcpyAsync(f.host, f.dev, 8, hipMemcpyDeviceToHost, (hipStream_t)stream))
                    && hip_ok(hipEventRecord(f.ev, (hipStream_t)stream)), "mgx: could not queue the grid check's report");
        f.pending = true;
        f.unreported = 0;
    }
    if (!f.pending) return;
    if (wait) {
        TORCH_CHECK(hip_ok(hipEventSynchronize(f.ev)), "mgx: hipEventSynchronize failed");
    } else if (!hip_ok(hipEventQuery(f.ev))) {          // (hipErrorNotReady is not an error: and must not stay behind as one)
        return;
    }
    raise_faults(f, stream);
}

// the pack kernel's counters of this device, and -- behind the pack -- their trip to the host
GridFaults &faults_of(int device) {
    GridFaults &f = g_faults[device];
    if (!f.dev) {
        TORCH_CHECK(hip_ok(hipMalloc((void **)&f.dev, 8)) && hip_ok(hipMemset(f.dev, 0, 8))
                    && hip_ok(hipHostMalloc((void **)&f.host, 8, hipHostMallocDefault))
                    && hip_ok(hipEventCreateWithFlags(&f.ev, hipEventDisableTiming)), "mgx: could not set up the grid check");
        f.host[0] = f.host[1] = 0;
    }
    return f;
}

template <class T> T *ptr(const Tensor &t) { return reinterpret_cast<T *>(t.data_ptr()); }
template <class T> T *ptr(const OptTensor &t) { return t.has_value() ? reinterpret_cast<T *>(t->data_ptr()) : nullptr; }

// ---- grid format ------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> pack_grid(const Tensor &cells3) {
    want(cells3, "cells3", at::kByte);
    TORCH_CHECK_VALUE(cells3.dim() >= 1 && cells3.size(-1) == 3, "mgx: pack_grid expects (type, color, state) bytes in the last axis");
    DeviceGuard g(cells3);
    auto shape = cells3.sizes().vec(); shape.pop_back();
    Tensor out = at::empty(shape, cells3.options().dtype(at::kShort));
    Tensor bad = at::zeros({1}, cells3.options().dtype(at::kInt));
    check(mgx_pack_grid(ptr<const uint8_t>(cells3), out.numel(), ptr<MgxCell>(out), ptr<int32_t>(bad), stream_of(cells3)), "mgx_pack_grid");
    return {out, bad};
}

Tensor unpack_grid(const Tensor &grid) {
    want(grid, "grid", at::kShort);
    DeviceGuard g(grid);
    auto shape = grid.sizes().vec(); shape.push_back(3);
    Tensor out = at::empty(shape, grid.options().dtype(at::kByte));
    check(mgx_unpack_grid(ptr<const MgxCell>(grid), grid.numel(), ptr<uint8_t>(out), stream_of(grid)), "mgx_unpack_grid");
    return out;
}

// the grid as the device wants it: packed cells.  A byte grid is packed here (no host sync); `bytes` tells the caller whether it
// has to unpack the result back into the caller's tensor.
Tensor as_cells(const Tensor &grid, const char *name, bool &bytes) {
    bytes = grid.scalar_type() == at::kByte;
    if (!bytes) return grid;
    want(grid, name, at::kByte);
    TORCH_CHECK_VALUE(grid.dim() == 4 && grid.size(-1) == 3, "mgx: a byte `", name, "` must be uint8 [B,H,W,3]");
    auto shape = grid.sizes().vec(); shape.pop_back();
    Tensor out = at::empty(shape, grid.options().dtype(at::kShort));
    void *st = stream_of(grid);
    std::lock_guard<std::mutex> lock(g_faults_mutex);
    GridFaults &f = faults_of(grid.device().index());
    check(mgx_pack_grid_env(ptr<const uint8_t>(grid), grid.size(0), (int32_t)grid.size(1), (int32_t)grid.size(2), ptr<MgxCell>(out), f.dev, st),
          "mgx_pack_grid_env");
    TORCH_CHECK(hip_ok(hipMemcpyAsync(f.host, f.dev, 8, hipMemcpyDeviceToHost, (hipStream_t)st))
                && hip_ok(hipEventRecord(f.ev, (hipStream_t)st)), "mgx: could not queue the grid check's report");
    f.pending = true;
    return out;
}

void cells_back(const Tensor &cells, const Tensor &grid_bytes) {
    check(mgx_unpack_grid(ptr<const MgxCell>(cells), cells.numel(), ptr<uint8_t>(grid_bytes), stream_of(cells)), "mgx_unpack_grid");
}

int64_t check_state(const MgxSpec &sc, const Tensor &cells, const Tensor &agents) {
    const int64_t B = cells.dim() > 0 ? cells.size(0) : 0;
    want(cells, "grid", at::kShort, {B, sc.height, sc.width}, true);
    want(agents, "agents", at::kByte, {B, sc.num_agents, 8}, true);
    return B;
}

// ---- gen_obs ---------------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> gen_obs_any(const Tensor &grid, const Tensor &agents, at::IntArrayRef spec, bool one_hot,
                                       const Tensor *out_obs = nullptr, const Tensor *out_dirs = nullptr) {
    const MgxSpec sc = spec_from(spec);
    DeviceGuard g(grid);
    poll_faults(grid.device().index(), stream_of(grid));
    bool bytes;
    const Tensor cells = as_cells(grid, "grid", bytes);
    const int64_t B = check_state(sc, cells, agents);
    const int64_t A = sc.num_agents, v = sc.view_size;
    Tensor obs, dirs;
    if (out_obs) {
        want(*out_obs, "obs", at::kByte, {B, A, v, v, one_hot ? 21 : 3}, true);
        want(*out_dirs, "dir", at::kByte, {B, A}, true);
        obs = *out_obs; dirs = *out_dirs;
    } else {
        obs = at::empty({B, A, v, v, one_hot ? 21 : 3}, agents.options());
        dirs = at::empty({B, A}, agents.options());
    }
    if (one_hot)
        check(mgx_gen_obs_one_hot(&sc, B, ptr<const MgxCell>(cells), ptr<const uint8_t>(agents), ptr<uint8_t>(obs), ptr<uint8_t>(dirs),
                                  stream_of(cells)), "mgx_gen_obs_one_hot");
    else
        check(mgx_gen_obs(&sc, B, ptr<const MgxCell>(cells), ptr<const uint8_t>(agents), ptr<uint8_t>(obs), ptr<uint8_t>(dirs),
                          stream_of(cells)), "mgx_gen_obs");
    return {obs, dirs};
}
std::tuple<Tensor, Tensor> gen_obs(const Tensor &grid, const Tensor &agents, at::IntArrayRef spec) { return gen_obs_any(grid, agents, spec, false); }
std::tuple<Tensor, Tensor> gen_obs_one_hot(const Tensor &grid, const Tensor &agents, at::IntArrayRef spec) { return gen_obs_any(grid, agents, spec, true); }
void gen_obs_out(const Tensor &grid, const Tensor &agents, at::IntArrayRef spec, Tensor obs, Tensor dirs) {
    gen_obs_any(grid, agents, spec, obs.dim() == 5 && obs.size(4) == 21, &obs, &dirs);
}

// ---- step family: one implementation over MgxStepArgs / mgx_step_ex ----------------------------------------------------
struct StepOut { Tensor obs, dirs, reward, terminated, truncated, was_reset; };

StepOut step_any(const Tensor &grid, const Tensor &agents, const Tensor &rng, const Tensor &step_count, const Tensor &actions,
                 const OptTensor &aux, const Tensor &err, at::IntArrayRef spec, int64_t T /* 0 = one step */, bool one_hot,
                 const OptTensor &pool_grid, const OptTensor &pool_agents, const OptTensor &pool_aux, const OptTensor &episode,
                 int64_t first_env, const OptTensor &hook_order, const char *what, const StepOut *pre = nullptr) {
    MgxSpec sc = spec_from(spec);
    DeviceGuard g(grid);
    poll_faults(grid.device().index(), stream_of(grid));
    bool bytes, pool_bytes = false;
    // A byte grid u8[B,H,W,3] -- the reference's own form -- of a plain step goes to the kernel AS IT IS (MgxSpec.cell_bytes = 3: the
    // step packs it into its LDS tile while loading it and writes changed cells back as bytes, include/mgx.h): no pack launch in
    // front, no unpack launch behind.  Rollouts and the one-hot output keep the conversion around the call; so does a byte grid
    // with a PACKED layout pool (the kernel takes both in one format).
    const bool fused_bytes = grid.scalar_type() == at::kByte && T == 0 && !one_hot
                             && (!pool_grid.has_value() || pool_grid->scalar_type() == at::kByte);
    Tensor cells;
    int64_t B;
    if (fused_bytes) {
        bytes = false;
        cells = grid;
        B = grid.dim() > 0 ? grid.size(0) : 0;
        want(grid, "grid", at::kByte, {B, sc.height, sc.width, 3}, true);
        want(agents, "agents", at::kByte, {B, sc.num_agents, 8}, true);
        sc.cell_bytes = 3;
    } else {
        cells = as_cells(grid, "grid", bytes);
        B = check_state(sc, cells, agents);
    }
    const int64_t A = sc.num_agents, v = sc.view_size;
    want(rng, "rng", at::kLong, {B, 4}, true);
    want(step_count, "step_count", at::kInt, {B}, true);
    if (T > 0) want(actions, "actions", at::kChar, {T, B, A}, true);
    else want(actions, "actions", at::kChar, {B, A}, true);
    want(err, "err", at::kInt, {2}, true);
    if (aux.has_value()) want(*aux, "aux", at::kByte, {B, 16}, true);
    else TORCH_CHECK_VALUE(sc.env_kind == MGX_KIND_EMPTY, "mgx: this env kind needs `aux` (the env subclass' hook state, include/mgx.h)");
    if (hook_order.has_value()) {
        if (T > 0) want(*hook_order, "hook_order", at::kByte, {T, B, A}, true);
        else want(*hook_order, "hook_order", at::kByte, {B, A}, true);
    }
    std::vector<int64_t> lead = T > 0 ? std::vector<int64_t>{T, B} : std::vector<int64_t>{B};
    auto shape = [&](std::initializer_list<int64_t> tail) { auto s = lead; s.insert(s.end(), tail); return s; };
    StepOut o;
    if (pre) {                                  // out-variant: the caller's tensors, checked like every other argument
        o = *pre;
        want(o.obs, "obs", at::kByte, shape({A, v, v, one_hot ? 21 : 3}), true);
        want(o.dirs, "dir", at::kByte, shape({A}), true);
        want(o.reward, "reward", at::kDouble, shape({A}), true);
        want(o.terminated, "terminated", at::kByte, shape({A}), true);
        want(o.truncated, "truncated", at::kByte, lead, true);
    } else {
        o.obs = at::empty(shape({A, v, v, one_hot ? 21 : 3}), agents.options());
        o.dirs = at::empty(shape({A}), agents.options());
        o.reward = at::empty(shape({A}), agents.options().dtype(at::kDouble));
        o.terminated = at::empty(shape({A}), agents.options());
        o.truncated = at::empty(lead, agents.options());
    }
    MgxStepArgs sa{};
    sa.grid = ptr<MgxCell>(cells); sa.agents = ptr<uint8_t>(agents); sa.rng = ptr<uint64_t>(rng); sa.step_count = ptr<int32_t>(step_count);
    sa.aux = ptr<uint8_t>(aux); sa.actions = ptr<const int8_t>(actions); sa.hook_order = ptr<const uint8_t>(hook_order);
    sa.obs = ptr<uint8_t>(o.obs); sa.dir = ptr<uint8_t>(o.dirs); sa.reward = ptr<double>(o.reward);
    sa.terminated = ptr<uint8_t>(o.terminated); sa.truncated = ptr<uint8_t>(o.truncated); sa.err = ptr<int32_t>(err);
    sa.steps = T > 0 ? (int32_t)T : 1;
    sa.one_hot = one_hot ? 1 : 0;
    MgxAutoReset ar{};
    Tensor pool_cells;
    if (pool_grid.has_value()) {
        if (fused_bytes) pool_cells = *pool_grid;
        else pool_cells = as_cells(*pool_grid, "pool_grid", pool_bytes);
        const int64_t K = pool_cells.dim() > 0 ? pool_cells.size(0) : 0;
        if (fused_bytes) want(pool_cells, "pool_grid", at::kByte, {K, sc.height, sc.width, 3}, true);
        else want(pool_cells, "pool_grid", at::kShort, {K, sc.height, sc.width}, true);
        TORCH_CHECK_VALUE(pool_agents.has_value() && episode.has_value(), "mgx: auto-reset needs pool_agents and episode");
        want(*pool_agents, "pool_agents", at::kByte, {K, A, 8}, true);
        if (pool_aux.has_value()) want(*pool_aux, "pool_aux", at::kByte, {K, 16}, true);
        want(*episode, "episode", at::kInt, {B}, true);
        if (pre) want(o.was_reset, "was_reset", at::kByte, lead, true);
        else o.was_reset = at::empty(lead, agents.options());
        ar.first_env = first_env; ar.pool_size = (int32_t)K; ar.pool_grid = ptr<const MgxCell>(pool_cells);
        ar.pool_agents = ptr<const uint8_t>(*pool_agents); ar.pool_aux = ptr<const uint8_t>(pool_aux);
        ar.episode = ptr<int32_t>(*episode); ar.was_reset = ptr<uint8_t>(o.was_reset);
        sa.auto_reset = &ar;
    } else if (!pre) {
        o.was_reset = at::zeros(lead, agents.options());
    }
    if (fused_bytes) {          // the kernel counts what the pack kernel would have (ring, unpackable values); the report is deferred
        void *st = stream_of(grid);
        GridFaults *fp = nullptr;
        bool report = false;
        {   // (the lock covers the bookkeeping only, not the launch: std::map nodes do not move)
            std::lock_guard<std::mutex> lock(g_faults_mutex);
            GridFaults &f = faults_of(grid.device().index());
            fp = &f;
            const void *g = grid.data_ptr();
            report = f.last_grid != g || (f.unreported & 31) == 0;
            f.last_grid = g;
            f.unreported = report ? 1 : f.unreported + 1;
        }
        sa.grid_bad = fp->dev;
        check(mgx_step_ex(&sc, B, &sa, st), what);
        if (report) {
            std::lock_guard<std::mutex> lock(g_faults_mutex);
            TORCH_CHECK(hip_ok(hipMemcpyAsync(fp->host, fp->dev, 8, hipMemcpyDeviceToHost, (hipStream_t)st))
                        && hip_ok(hipEventRecord(fp->ev, (hipStream_t)st)), "mgx: could not queue the grid check's report");
            fp->pending = true;
        }
        return o;
    }
    check(mgx_step_ex(&sc, B, &sa, stream_of(cells)), what);
    if (bytes) cells_back(cells, grid);
    return o;
}

using Out5 = std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor>;
using Out6 = std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>;

Out5 step(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, const Tensor &actions, OptTensor aux, Tensor err,
          at::IntArrayRef spec) {
    StepOut o = step_any(grid, agents, rng, step_count, actions, aux, err, spec, 0, false, {}, {}, {}, {}, 0, {}, "mgx_step");
    return {o.obs, o.dirs, o.reward, o.terminated, o.truncated};
}

Out5 step_ordered(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, const Tensor &actions, const Tensor &hook_order,
                  OptTensor aux, Tensor err, at::IntArrayRef spec) {
    StepOut o = step_any(grid, agents, rng, step_count, actions, aux, err, spec, 0, false, {}, {}, {}, {}, 0, hook_order, "mgx_step");
    return {o.obs, o.dirs, o.reward, o.terminated, o.truncated};
}

Out6 step_autoreset(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, const Tensor &actions, OptTensor aux, Tensor err,
                    const Tensor &pool_grid, const Tensor &pool_agents, const OptTensor &pool_aux, Tensor episode,
                    int64_t first_env, at::IntArrayRef spec) {
    StepOut o = step_any(grid, agents, rng, step_count, actions, aux, err, spec, 0, false, pool_grid, pool_agents, pool_aux, episode,
                         first_env, {}, "mgx_step_autoreset");
    return {o.obs, o.dirs, o.reward, o.terminated, o.truncated, o.was_reset};
}

Out5 rollout(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, const Tensor &actions, OptTensor aux, Tensor err,
             at::IntArrayRef spec) {
    TORCH_CHECK_VALUE(actions.dim() == 3, "mgx: rollout expects actions[T, B, A]");
    StepOut o = step_any(grid, agents, rng, step_count, actions, aux, err, spec, actions.size(0), false, {}, {}, {}, {}, 0, {},
                         "mgx_rollout");
    return {o.obs, o.dirs, o.reward, o.terminated, o.truncated};
}

Out5 rollout_one_hot(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, const Tensor &actions, OptTensor aux, Tensor err,
                     at::IntArrayRef spec) {
    TORCH_CHECK_VALUE(actions.dim() == 3, "mgx: rollout_one_hot expects actions[T, B, A]");
    StepOut o = step_any(grid, agents, rng, step_count, actions, aux, err, spec, actions.size(0), true, {}, {}, {}, {}, 0, {},
                         "mgx_rollout (one-hot)");
    return {o.obs, o.dirs, o.reward, o.terminated, o.truncated};
}

Out6 step_one_hot(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, const Tensor &actions, OptTensor aux, Tensor err,
                  const OptTensor &pool_grid, const OptTensor &pool_agents, const OptTensor &pool_aux, OptTensor episode,
                  int64_t first_env, at::IntArrayRef spec) {
    StepOut o = step_any(grid, agents, rng, step_count, actions, aux, err, spec, 0, true, pool_grid, pool_agents, pool_aux, episode,
                         first_env, {}, "mgx_step_one_hot");
    return {o.obs, o.dirs, o.reward, o.terminated, o.truncated, o.was_reset};
}

// ---- bound step: every pointer of the call resolved and checked ONCE ---------------------------------------------------------
// A policy-in-the-loop caller steps the SAME tensors thousands of times; the out-variants still push 13-19 arguments through the
// dispatcher and re-check every one of them per call (6.4 us of host time against a 5.5 us kernel at 4096 envs).  bind_step checks
// them once, keeps the tensors alive and returns a handle; step_bound(handle, actions) checks the actions and launches -- the step
// writes the bound outputs.  Packed-cell state only (a byte grid is converted per call: use the out-variants for that).
struct BoundStep {
    MgxSpec sc{};
    int64_t B = 0, A = 0;
    MgxStepArgs sa{};
    MgxAutoReset ar{};
    c10::Device device{c10::DeviceType::CPU};
    std::vector<Tensor> keep;          // every bound tensor: the launch reads / writes their storage
    bool live = false;
};
// A handle = slot | generation << 20.  Slots are reused; the generation a slot was handed out with is part of the handle, so a
// stale handle (kept past its unbind_step) is refused instead of silently stepping whatever was bound into the slot since.  A
// call copies the shared_ptr under the lock and holds it for its duration: a concurrent unbind_step cannot free the struct, or
// the tensors it keeps alive, under a launch that is being set up.
std::mutex g_bound_mutex;
std::vector<std::shared_ptr<BoundStep>> g_bound;
std::vector<int64_t> g_bound_gen;
constexpr int64_t kSlotBits = 20;

std::shared_ptr<BoundStep> bound_at(int64_t handle, const char *who) {
    const int64_t slot = handle & ((int64_t(1) << kSlotBits) - 1), gen = handle >> kSlotBits;
    std::lock_guard<std::mutex> lock(g_bound_mutex);
    TORCH_CHECK_VALUE(handle >= 0 && slot < (int64_t)g_bound.size() && g_bound[slot] && g_bound_gen[slot] == gen,
                      "mgx: ", who, ": no such handle ", handle, " (never bound, or unbound since)");
    return g_bound[slot];
}

int64_t bind_step(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, OptTensor aux, Tensor err, at::IntArrayRef spec,
                  Tensor obs, Tensor dirs, Tensor reward, Tensor terminated, Tensor truncated, const OptTensor &pool_grid,
                  const OptTensor &pool_agents, const OptTensor &pool_aux, OptTensor episode, int64_t first_env, OptTensor was_reset,
                  bool one_hot) {
    auto b = std::make_shared<BoundStep>();
    b->sc = spec_from(spec);
    const MgxSpec &sc = b->sc;
    DeviceGuard g(grid);
    TORCH_CHECK_TYPE(grid.scalar_type() == at::kShort, "mgx: bind_step takes the packed grid i16[B,H,W] (mgx_pack_grid), got ", grid.scalar_type());
    const int64_t B = check_state(sc, grid, agents), A = sc.num_agents, v = sc.view_size;
    b->B = B; b->A = A; b->device = grid.device();
    want(rng, "rng", at::kLong, {B, 4}, true);
    want(step_count, "step_count", at::kInt, {B}, true);
    want(err, "err", at::kInt, {2}, true);
    if (aux.has_value()) want(*aux, "aux", at::kByte, {B, 16}, true);
    else TORCH_CHECK_VALUE(sc.env_kind == MGX_KIND_EMPTY, "mgx: this env kind needs `aux` (the env subclass' hook state, include/mgx.h)");
    want(obs, "obs", at::kByte, {B, A, v, v, one_hot ? 21 : 3}, true);
    want(dirs, "dir", at::kByte, {B, A}, true);
    want(reward, "reward", at::kDouble, {B, A}, true);
    want(terminated, "terminated", at::kByte, {B, A}, true);
    want(truncated, "truncated", at::kByte, {B}, true);
    MgxStepArgs &sa = b->sa;
    sa.grid = ptr<MgxCell>(grid); sa.agents = ptr<uint8_t>(agents); sa.rng = ptr<uint64_t>(rng); sa.step_count = ptr<int32_t>(step_count);
    sa.aux = ptr<uint8_t>(aux); sa.obs = ptr<uint8_t>(obs); sa.dir = ptr<uint8_t>(dirs); sa.reward = ptr<double>(reward);
    sa.terminated = ptr<uint8_t>(terminated); sa.truncated = ptr<uint8_t>(truncated); sa.err = ptr<int32_t>(err);
    sa.steps = 1; sa.one_hot = one_hot ? 1 : 0;
    b->keep = {grid, agents, rng, step_count, err, obs, dirs, reward, terminated, truncated};
    if (aux.has_value()) b->keep.push_back(*aux);
    if (pool_grid.has_value()) {
        const int64_t K = pool_grid->dim() > 0 ? pool_grid->size(0) : 0;
        want(*pool_grid, "pool_grid", at::kShort, {K, sc.height, sc.width}, true);
        TORCH_CHECK_VALUE(pool_agents.has_value() && episode.has_value(), "mgx: auto-reset needs pool_agents and episode");
        want(*pool_agents, "pool_agents", at::kByte, {K, A, 8}, true);
        if (pool_aux.has_value()) want(*pool_aux, "pool_aux", at::kByte, {K, 16}, true);
        want(*episode, "episode", at::kInt, {B}, true);
        if (was_reset.has_value()) want(*was_reset, "was_reset", at::kByte, {B}, true);
        MgxAutoReset &ar = b->ar;
        ar.first_env = first_env; ar.pool_size = (int32_t)K; ar.pool_grid = ptr<const MgxCell>(*pool_grid);
        ar.pool_agents = ptr<const uint8_t>(*pool_agents); ar.pool_aux = ptr<const uint8_t>(pool_aux);
        ar.episode = ptr<int32_t>(*episode); ar.was_reset = ptr<uint8_t>(was_reset);
        sa.auto_reset = &b->ar;                                    // (the record is heap-allocated and never moves)
        b->keep.push_back(*pool_grid); b->keep.push_back(*pool_agents); b->keep.push_back(*episode);
        if (pool_aux.has_value()) b->keep.push_back(*pool_aux);
        if (was_reset.has_value()) b->keep.push_back(*was_reset);
    }
    b->live = true;
    std::lock_guard<std::mutex> lock(g_bound_mutex);
    for (size_t i = 0; i < g_bound.size(); ++i)
        if (!g_bound[i]) { g_bound[i] = std::move(b); return (int64_t)i | (++g_bound_gen[i] << kSlotBits); }
    TORCH_CHECK((int64_t)g_bound.size() < (int64_t(1) << kSlotBits), "mgx: bind_step: too many bound steps");
    g_bound.push_back(std::move(b));
    g_bound_gen.push_back(0);
    return (int64_t)g_bound.size() - 1;
}

void step_bound(int64_t handle, const Tensor &actions) {
    const std::shared_ptr<BoundStep> b = bound_at(handle, "step_bound");      // (held until the launch is enqueued)
    TORCH_CHECK(actions.device() == b->device, "mgx: `actions` is on ", actions.device(), " but the bound state is on ", b->device);
    TORCH_CHECK_TYPE(actions.scalar_type() == at::kChar, "mgx: `actions` must be int8, got ", actions.scalar_type());
    TORCH_CHECK_VALUE(actions.is_contiguous() && actions.dim() == 2 && actions.size(0) == b->B && actions.size(1) == b->A,
                      "mgx: `actions` must be a contiguous int8 tensor [", b->B, ", ", b->A, "], got ", actions.sizes());
    c10::hip::HIPGuardMasqueradingAsCUDA g(b->device);
    MgxStepArgs sa = b->sa;                                         // (a copy: concurrent callers of one handle do not share `actions`)
    sa.actions = ptr<const int8_t>(actions);
    check(mgx_step_ex(&b->sc, b->B, &sa, stream_of(actions)), "mgx_step (bound)");
}

void unbind_step(int64_t handle) {
    (void)bound_at(handle, "unbind_step");
    std::lock_guard<std::mutex> lock(g_bound_mutex);
    g_bound[handle & ((int64_t(1) << kSlotBits) - 1)].reset();
}

// out-variants: nothing is allocated; the outputs are written into the caller's tensors
void step_out(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, const Tensor &actions, OptTensor aux, Tensor err,
              at::IntArrayRef spec, Tensor obs, Tensor dirs, Tensor reward, Tensor terminated, Tensor truncated) {
    const StepOut pre{obs, dirs, reward, terminated, truncated, Tensor()};
    step_any(grid, agents, rng, step_count, actions, aux, err, spec, 0, false, {}, {}, {}, {}, 0, {}, "mgx_step", &pre);
}

void step_autoreset_out(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, const Tensor &actions, OptTensor aux, Tensor err,
                        const Tensor &pool_grid, const Tensor &pool_agents, const OptTensor &pool_aux, Tensor episode,
                        int64_t first_env, at::IntArrayRef spec, Tensor obs, Tensor dirs, Tensor reward, Tensor terminated,
                        Tensor truncated, Tensor was_reset) {
    const StepOut pre{obs, dirs, reward, terminated, truncated, was_reset};
    step_any(grid, agents, rng, step_count, actions, aux, err, spec, 0, false, pool_grid, pool_agents, pool_aux, episode, first_env, {},
             "mgx_step_autoreset", &pre);
}

void step_one_hot_out(Tensor grid, Tensor agents, Tensor rng, Tensor step_count, const Tensor &actions, OptTensor aux, Tensor err,
                      const OptTensor &pool_grid, const OptTensor &pool_agents, const OptTensor &pool_aux, OptTensor episode,
                      int64_t first_env, at::IntArrayRef spec, Tensor obs, Tensor dirs, Tensor reward, Tensor terminated,
                      Tensor truncated, OptTensor was_reset) {
    TORCH_CHECK_VALUE(!pool_grid.has_value() || was_reset.has_value(), "mgx: step_one_hot_out with a layout pool needs `was_reset`");
    const StepOut pre{obs, dirs, reward, terminated, truncated, was_reset.has_value() ? *was_reset : Tensor()};
    step_any(grid, agents, rng, step_count, actions, aux, err, spec, 0, true, pool_grid, pool_agents, pool_aux, episode, first_env, {},
             "mgx_step_one_hot", &pre);
}

// ---- the kernels' preconditions, on request -----------------------------------------------------------------------------------
// bad i32[4] on the device: [0] cells that are not a valid packed cell, [1] outer-ring cells that are not WALL, [2] agent rows
// outside the walls / malformed, [3] first env with a violation (INT32_MAX: none).  No synchronisation: the caller reads it.
Tensor check_grid(const Tensor &grid, const Tensor &agents, at::IntArrayRef spec) {
    const MgxSpec sc = spec_from(spec);
    DeviceGuard g(grid);
    const int64_t B = check_state(sc, grid, agents);
    Tensor bad = at::zeros({4}, grid.options().dtype(at::kInt));
    bad.select(0, 3).fill_(INT32_MAX);
    check(mgx_check_grid(&sc, B, ptr<const MgxCell>(grid), ptr<const uint8_t>(agents), ptr<int32_t>(bad), stream_of(grid)), "mgx_check_grid");
    return bad;
}

// waits for the report of the last byte grid handed to an op on `device` and raises if it had faults (see the header comment)
void check_errors(int64_t device) {
    c10::hip::HIPGuardMasqueradingAsCUDA g(c10::Device(c10::DeviceType::CUDA, (c10::DeviceIndex)device));
    poll_faults((int)device, (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA((c10::DeviceIndex)device).stream(), true);
}

// ---- either side of the path ---------------------------------------------------------------------------------------------
Tensor one_hot(const Tensor &cells, at::IntArrayRef dim_sizes) {
    want(cells, "cells", at::kByte);
    TORCH_CHECK_VALUE(cells.dim() >= 1 && cells.size(-1) == 3 && dim_sizes.size() == 3, "mgx: one_hot expects cells[..., 3] and three dim sizes");
    DeviceGuard g(cells);
    const int32_t ds[3] = {(int32_t)dim_sizes[0], (int32_t)dim_sizes[1], (int32_t)dim_sizes[2]};
    auto shape = cells.sizes().vec(); shape.back() = (int64_t)ds[0] + ds[1] + ds[2];
    Tensor out = at::empty(shape, cells.options());
    check(mgx_one_hot(ptr<const uint8_t>(cells), cells.numel() / 3, ds, ptr<uint8_t>(out), stream_of(cells)), "mgx_one_hot");
    return out;
}

Tensor full_obs(const Tensor &grid, const Tensor &agents, at::IntArrayRef spec) {
    const MgxSpec sc = spec_from(spec);
    DeviceGuard g(grid);
    poll_faults(grid.device().index(), stream_of(grid));
    bool bytes;
    const Tensor cells = as_cells(grid, "grid", bytes);
    const int64_t B = check_state(sc, cells, agents);
    Tensor out = at::empty({B, sc.width, sc.height, 3}, agents.options());
    check(mgx_full_obs(&sc, B, ptr<const MgxCell>(cells), ptr<const uint8_t>(agents), ptr<uint8_t>(out), stream_of(cells)), "mgx_full_obs");
    return out;
}

int64_t abi_version() { return mgx_abi_version(); }

}  // namespace

TORCH_LIBRARY(mgx, m) {
    m.def("gen_obs(Tensor grid, Tensor agents, int[] spec) -> (Tensor, Tensor)");
    m.def("step(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor actions, "
          "Tensor(f!)? aux, Tensor(e!) err, int[] spec) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("step_ordered(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor actions, "
          "Tensor hook_order, Tensor(f!)? aux, Tensor(e!) err, int[] spec) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("step_autoreset(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor actions, "
          "Tensor(f!)? aux, Tensor(e!) err, Tensor pool_grid, Tensor pool_agents, Tensor? pool_aux, Tensor(g!) episode, "
          "int first_env, int[] spec) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("rollout(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor actions, "
          "Tensor(f!)? aux, Tensor(e!) err, int[] spec) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("rollout_one_hot(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor actions, "
          "Tensor(f!)? aux, Tensor(e!) err, int[] spec) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("gen_obs_one_hot(Tensor grid, Tensor agents, int[] spec) -> (Tensor, Tensor)");
    m.def("step_one_hot(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor actions, "
          "Tensor(f!)? aux, Tensor(e!) err, Tensor? pool_grid, Tensor? pool_agents, Tensor? pool_aux, Tensor(g!)? episode, "
          "int first_env, int[] spec) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("gen_obs_out(Tensor grid, Tensor agents, int[] spec, Tensor(a!) obs, Tensor(b!) dir) -> ()");
    m.def("step_out(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor actions, Tensor(f!)? aux, "
          "Tensor(e!) err, int[] spec, Tensor(g!) obs, Tensor(h!) dir, Tensor(i!) reward, Tensor(j!) terminated, Tensor(k!) truncated) -> ()");
    m.def("step_autoreset_out(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor actions, "
          "Tensor(f!)? aux, Tensor(e!) err, Tensor pool_grid, Tensor pool_agents, Tensor? pool_aux, Tensor(g!) episode, int first_env, "
          "int[] spec, Tensor(h!) obs, Tensor(i!) dir, Tensor(j!) reward, Tensor(k!) terminated, Tensor(l!) truncated, "
          "Tensor(m!) was_reset) -> ()");
    m.def("step_one_hot_out(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor actions, "
          "Tensor(f!)? aux, Tensor(e!) err, Tensor? pool_grid, Tensor? pool_agents, Tensor? pool_aux, Tensor(g!)? episode, int first_env, "
          "int[] spec, Tensor(h!) obs, Tensor(i!) dir, Tensor(j!) reward, Tensor(k!) terminated, Tensor(l!) truncated, "
          "Tensor(m!)? was_reset) -> ()");
    m.def("bind_step(Tensor(a!) grid, Tensor(b!) agents, Tensor(c!) rng, Tensor(d!) step_count, Tensor(e!)? aux, Tensor(f!) err, "
          "int[] spec, Tensor(g!) obs, Tensor(h!) dir, Tensor(i!) reward, Tensor(j!) terminated, Tensor(k!) truncated, "
          "Tensor? pool_grid, Tensor? pool_agents, Tensor? pool_aux, Tensor(l!)? episode, int first_env, Tensor(m!)? was_reset, "
          "bool one_hot) -> int");
    m.def("step_bound(int handle, Tensor actions) -> ()");
    m.def("unbind_step(int handle) -> ()", &unbind_step);
    m.def("check_grid(Tensor grid, Tensor agents, int[] spec) -> Tensor");
    m.def("check_errors(int device) -> ()", &check_errors);
    m.def("one_hot(Tensor cells, int[] dim_sizes) -> Tensor");
    m.def("full_obs(Tensor grid, Tensor agents, int[] spec) -> Tensor");
    m.def("pack_grid(Tensor cells3) -> (Tensor, Tensor)");
    m.def("unpack_grid(Tensor grid) -> Tensor");
    m.def("abi_version() -> int", &abi_version);
}

TORCH_LIBRARY_IMPL(mgx, CUDA, m) {
    m.impl("gen_obs", &gen_obs);
    m.impl("step", &step);
    m.impl("step_ordered", &step_ordered);
    m.impl("step_autoreset", &step_autoreset);
    m.impl("rollout", &rollout);
    m.impl("rollout_one_hot", &rollout_one_hot);
    m.impl("gen_obs_one_hot", &gen_obs_one_hot);
    m.impl("step_one_hot", &step_one_hot);
    m.impl("gen_obs_out", &gen_obs_out);
    m.impl("step_out", &step_out);
    m.impl("step_autoreset_out", &step_autoreset_out);
    m.impl("step_one_hot_out", &step_one_hot_out);
    m.impl("bind_step", &bind_step);
    m.impl("step_bound", &step_bound);
    m.impl("check_grid", &check_grid);
    m.impl("one_hot", &one_hot);
    m.impl("full_obs", &full_obs);
    m.impl("pack_grid", &pack_grid);
    m.impl("unpack_grid", &unpack_grid);
}
