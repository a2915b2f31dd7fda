// Probe (round 6): what does a device-side grid barrier cost on MI355X next to a kernel boundary?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_probe scripts/probes/barrier_probe.hip && /tmp/barrier_probe
// Phase p: workgroup g writes buf[p & 1][g * 256 + t] = value(p, g, t); barrier; reads the row of workgroup (g + 97) % G written in
// phase p and checks it (cross-XCD visibility), accumulates.  Compared with the same phases as a chain of kernels (plain launches
// and one hipGraph).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__device__ __forceinline__ float val(int p, int g, int t) { return (float)((p * 131 + g * 7 + t) & 1023); }

__global__ __launch_bounds__(256) void persistent(int P, float* buf, unsigned* counter, int* bad, float* out) {
    const int G = gridDim.x, g = blockIdx.x, t = threadIdx.x;
    float acc = 0.f;
    for (int p = 0; p < P; ++p) {
        float* b = buf + (size_t)(p & 1) * G * 256;
        b[g * 256 + t] = val(p, g, t);
        grid_barrier(counter, (unsigned)(p + 1) * G);
        const int src = (g + 97) % G;
        const float v = b[src * 256 + t];
        if (v != val(p, src, t)) atomicAdd(bad, 1);
        acc += v;
        // (the write of phase p + 1 goes to the other half: no second barrier needed before it; the write of phase p + 2 into this
        // half is behind barrier p + 1, which every reader of phase p has passed)
    }
    out[g * 256 + t] = acc;
}
__global__ __launch_bounds__(256) void phase_kernel(int p, float* buf, int* bad, float* out) {
    const int G = gridDim.x, g = blockIdx.x, t = threadIdx.x;
    float* b = buf + (size_t)(p & 1) * G * 256;
    float* prev = buf + (size_t)((p + 1) & 1) * G * 256;
    const int src = (g + 97) % G;
    if (p > 0) {
        const float v = prev[src * 256 + t];
        if (v != val(p - 1, src, t)) atomicAdd(bad, 1);
        out[g * 256 + t] += v;
    }
    b[g * 256 + t] = val(p, g, t);
}

int main(int argc, char** argv) {
    const int P = 2000;
    for (int G : {64, 128, 256, 512}) {
        float *buf, *out; unsigned* counter; int* bad;
        CK(hipMalloc(&buf, sizeof(float) * 2 * G * 256)); CK(hipMalloc(&out, sizeof(float) * G * 256));
        CK(hipMalloc(&counter, 4)); CK(hipMalloc(&bad, 4));
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(out, 0, sizeof(float) * G * 256));
        hipStream_t s; CK(hipStreamCreate(&s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(counter, 0, 4, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(persistent, dim3(G), dim3(256), 0, s, P, buf, counter, bad, out);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("G=%3d  persistent: %.3f us per phase (bad=%d)", G, ms * 1e3 / P, hb);
        CK(hipMemset(bad, 0, 4));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int p = 0; p < P; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, s, p, buf, bad, out);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("   launches: %.3f us per phase", ms * 1e3 / P);
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < 200; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, s, p, buf, bad, out);
        CK(hipStreamEndCapture(s, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("   graph: %.3f us per phase (bad=%d)\n", ms * 1e3 / 2000, hb);
    }
    return 0;
}
