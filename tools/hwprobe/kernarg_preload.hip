// Does kernel-argument preloading (-mllvm -amdgpu-kernarg-preload-count=N: the first N dwords of the explicit arguments arrive in
// SGPRs with the wavefront instead of through an s_load) shorten a short dependent kernel?  A chain of tiny launches in a hipGraph,
// each doing what the fused kernel's prologue does: fetch pointers from the kernel arguments, then a dependent global load -> store.
// Build twice (with / without the flag) and compare:  hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=14] ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void step(long n, const int *a, const int *b, const int *c, int *out, int *out2, int *out3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[i] = a[i] + b[i] + c[i]; out2[i] = out3[i] + 1; }
}
int main() {
    const long n = 1024 * 64;
    int *buf; hipMalloc(&buf, n * 6 * sizeof(int)); hipMemset(buf, 0, n * 6 * sizeof(int));
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int t = 0; t < 1000; ++t) hipLaunchKernelGGL(step, dim3(n / 256), dim3(256), 0, s, n, buf, buf + n, buf + 2 * n, buf + 3 * n, buf + 4 * n, buf + 5 * n);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, s);
        for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%.3f us per launch\n", ms * 1e3 / 10000);
    }
    return 0;
}
