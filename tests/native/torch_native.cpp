// torch_native.cpp -- TEST INFRASTRUCTURE.  The compiled operator library (multigrid_amd/lib/libmgx_torch.so, csrc/mgx_torch.cpp)
// used from plain C++: no Python interpreter anywhere.  dlopen()s the library (its static initialisers register
// TORCH_LIBRARY(mgx) with the dispatcher), looks the ops up by schema and calls them.
//   torch_native            schemas present, mgx::abi_version() through the dispatcher           (no GPU needed)
//   torch_native gpu        + mgx::pack_grid / unpack_grid / gen_obs on HIP tensors; gen_obs compared with the same call
//                             made directly on the C ABI (mgx_gen_obs) -- prints "torch native ok"
// Built by tests/test_torch_native.py with g++ against the installed torch.
#include <ATen/ATen.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <tuple>

#include "mgx.h"

int main(int argc, char **argv) {
    const char *lib = argc > 1 ? argv[1] : "libmgx_torch.so";
    const bool gpu = argc > 2 && std::strcmp(argv[2], "gpu") == 0;
    if (!dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) { std::fprintf(stderr, "dlopen %s: %s\n", lib, dlerror()); return 1; }
    auto &d = c10::Dispatcher::singleton();
    for (const char *name : {"mgx::gen_obs", "mgx::step", "mgx::step_ordered", "mgx::step_autoreset", "mgx::rollout", "mgx::gen_obs_one_hot",
                             "mgx::step_one_hot", "mgx::one_hot", "mgx::full_obs", "mgx::pack_grid", "mgx::unpack_grid"})
        if (!d.findSchema({name, ""})) { std::fprintf(stderr, "%s is not registered\n", name); return 2; }
    const int64_t abi = d.findSchemaOrThrow("mgx::abi_version", "").typed<int64_t()>().call();
    if (abi != MGX_ABI_VERSION) { std::fprintf(stderr, "ABI %lld != %d\n", (long long)abi, MGX_ABI_VERSION); return 3; }
    std::printf("schemas ok, abi %lld\n", (long long)abi);
    if (!gpu) return 0;

    const int64_t B = 257, W = 9, H = 7, A = 3, V = 5;
    auto u8 = at::TensorOptions().dtype(at::kByte).device(at::kCUDA, 0);
    at::Tensor cells3 = at::ones({B, H, W, 3}, u8) * at::tensor({1, 0, 0}, u8);          // empty cells ...
    cells3.select(1, 0).copy_(at::tensor({2, 5, 0}, u8)); cells3.select(1, H - 1).copy_(at::tensor({2, 5, 0}, u8));   // ... walled
    cells3.select(2, 0).copy_(at::tensor({2, 5, 0}, u8)); cells3.select(2, W - 1).copy_(at::tensor({2, 5, 0}, u8));
    cells3.index_put_({at::indexing::Slice(), 3, 4}, at::tensor({4, 2, 1}, u8));           // a closed blue door
    auto pack = d.findSchemaOrThrow("mgx::pack_grid", "").typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor &)>();
    auto unpack = d.findSchemaOrThrow("mgx::unpack_grid", "").typed<at::Tensor(const at::Tensor &)>();
    auto [grid, bad] = pack.call(cells3);
    if (bad.item<int>() != 0 || !at::equal(unpack.call(grid), cells3)) { std::fprintf(stderr, "pack / unpack round trip failed\n"); return 4; }
    at::Tensor agents = at::zeros({B, A, 8}, u8);
    agents.select(2, 1).copy_(at::arange(B * A, u8.dtype(at::kLong)).remainder(4).view({B, A}).to(at::kByte));   // dirs
    agents.select(2, 2).fill_(2); agents.select(2, 3).fill_(3); agents.select(2, 5).fill_(1);                    // (2,3), carrying nothing
    const std::vector<int64_t> spec = {W, H, A, V, 100, 0, 1, 0, 1, 0, 0};
    auto gen_obs = d.findSchemaOrThrow("mgx::gen_obs", "").typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor &, const at::Tensor &, at::IntArrayRef)>();
    auto [obs, dirs] = gen_obs.call(grid, agents, spec);
    auto [obs_b, dirs_b] = gen_obs.call(cells3, agents, spec);                              // the byte-grid form of the same op
    MgxSpec sp{(int32_t)W, (int32_t)H, (int32_t)A, (int32_t)V, 100, 0, 1, 0, 1, 0, MGX_KIND_EMPTY};
    at::Tensor obs2 = at::empty_like(obs), dirs2 = at::empty_like(dirs);
    const int rc = mgx_gen_obs(&sp, B, (const MgxCell *)grid.data_ptr(), (const uint8_t *)agents.data_ptr(), (uint8_t *)obs2.data_ptr(),
                               (uint8_t *)dirs2.data_ptr(), nullptr);
    if (rc != MGX_OK) { std::fprintf(stderr, "mgx_gen_obs: %s\n", mgx_error_string(rc)); return 5; }
    if (!at::equal(obs.cpu(), obs2.cpu()) || !at::equal(dirs.cpu(), dirs2.cpu()) || !at::equal(obs_b, obs)) { std::fprintf(stderr, "op != C ABI\n"); return 6; }
    if (obs.select(1, 0).select(0, 0).sum().item<int64_t>() == 0) { std::fprintf(stderr, "empty observation\n"); return 7; }
    std::printf("torch native ok\n");
    return 0;
}
