// Does a kernel launched with hipExtAnyOrderLaunch (no AQL barrier bit) start while its predecessor in the same
// stream is still running on this part?  A: 256 workgroups of busy work, each bumps a counter when done.
// B: one workgroup that spins on the counter (bounded), then consumes it.  Compare A;B per-iteration time with B
// launched normally and with the flag.  Build: hipcc -O3 --offload-arch=gfx950 anyorder_probe.hip -o anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_a(float* sink, unsigned* counter, int iters) {
    float v = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 1e-4f);
    if (v == 12345.678f) sink[0] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void k_b(unsigned* counter, unsigned target, unsigned* out, unsigned long long* spins) {
    if (threadIdx.x == 0) {
        unsigned long long n = 0;
        unsigned c;
        while ((c = __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < target && n < 4000000ull) {
            __builtin_amdgcn_s_sleep(8);
            ++n;
        }
        out[0] = c;
        spins[0] += n;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    float* sink; unsigned *counter, *out; unsigned long long* spins;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&spins, 8));
    CK(hipMemset(counter, 0, 4)); CK(hipMemset(spins, 0, 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_wg = 256, reps = 300;
    for (int iters : {200, 2000, 8000}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            unsigned long long sp = 0;
            for (int trial = 0; trial < 4; ++trial) {
                CK(hipMemsetAsync(spins, 0, 8, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) {
                    hipLaunchKernelGGL(k_a, dim3(n_wg), dim3(256), 0, st, sink, counter, iters);
                    if (mode == 0)
                        hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, st, counter, (unsigned)n_wg, out, spins);
                    else
                        hipExtLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch,
                                              counter, (unsigned)n_wg, out, spins);
                }
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(&sp, spins, 8, hipMemcpyDeviceToHost));
            }
            unsigned o; CK(hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost));
            printf("iters %5d  %-26s %7.2f us per A;B pair   (B saw %u, spin loops per B %.1f)\n", iters,
                   mode ? "B with hipExtAnyOrderLaunch" : "B launched normally", best * 1e3f / reps, o, (double)sp / reps);
            fflush(stdout);
        }
    }
    return 0;
}
