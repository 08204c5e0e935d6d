// What would one dense layer cost inside a persistent per-block kernel?  
// C clusters (one per fold model) of W workgroups run L "layer phases" inside ONE launch.  A phase = every workgroup publishes
// `pub` bytes (sc1 write-through stores), signals a per-cluster counter (agent-scope atomic add after s_waitcnt vmcnt(0) + workgroup
// barrier), waits until all W workgroups of its cluster have signalled (one lane polls with sc1 loads + s_sleep; BOUNDED: a phase
// that does not complete within ~2^22 polls sets an error flag and every workgroup leaves), then reads everybody's `pub` bytes with
// sc1 loads (an all-gather of W * pub bytes per workgroup).  The counter is monotonic (phase p waits for (p + 1) * W), so there is
// nothing to reset.  Reported: microseconds per phase = the synchronisation + exchange cost a per-layer hand-off would pay, to be
// compared with today's ~1.5-2 us kernel boundary + ~2.4 us launch skeleton + dependent-load chain per launch.
//   hipcc --offload-arch=gfx950 -O3 -o cluster_handoff cluster_handoff.hip && ./cluster_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k(float* buf, unsigned* counters, unsigned* err, int W, int L, int pub_floats, float* sink) {
    const int c = blockIdx.y, w = blockIdx.x, tid = threadIdx.x;
    float* mine = buf + ((size_t)c * W + w) * pub_floats;
    const float* all = buf + (size_t)c * W * pub_floats;
    unsigned* cnt = counters + c * 64;                // one 256-B line per cluster
    __shared__ int ok;
    float acc = 0.f;
    for (int p = 0; p < L; ++p) {
        for (int i = tid; i < pub_floats; i += 256) __hip_atomic_store(&mine[i], (float)(p + w + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(p + 1) * W;
            int spins = 0, good = 1;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { good = 0; atomicExch(err, 1u); break; }
            }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) good = 0;
            ok = good;
        }
        __syncthreads();
        if (!ok) return;                              // every wave reaches this exit: the grid drains
        const int n = W * pub_floats;                // all-gather: 8 independent sc1 loads in flight per lane
        for (int i0 = tid; i0 < n; i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = i0 + 256 * j < n ? __hip_atomic_load(&all[i0 + 256 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    float *buf, *sink; unsigned *cnt, *err;
    hipMalloc(&buf, 64 << 20); hipMalloc(&sink, 4); hipMalloc(&cnt, 64 * 256); hipMalloc(&err, 4);
    const int L = 200;
    printf("%-28s %s\n", "clusters x workgroups, bytes", "us per phase (publish + arrive + wait + all-gather)");
    const int Ws[] = {4, 8, 16, 32, 64}, Cs[] = {1, 3, 5}, pubs[] = {256, 512, 2048};    // floats: 1 KB, 2 KB, 8 KB per workgroup
    for (int pub : pubs)
        for (int C : Cs)
            for (int W : Ws) {
                if (C * W > 256) continue;            // one workgroup per CU at most: every workgroup is resident, nobody waits for a slot
                hipMemset(cnt, 0, 64 * 256); hipMemset(err, 0, 4);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(k, dim3(W, C), dim3(256), 0, 0, buf, cnt, err, W, 8, pub, sink);       // warm-up
                hipDeviceSynchronize();
                hipMemset(cnt, 0, 64 * 256);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(W, C), dim3(256), 0, 0, buf, cnt, err, W, L, pub, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
                printf("%d x %2d, %5d B each          %6.2f%s\n", C, W, pub * 4, ms * 1e3 / L, e ? "  (TIMED OUT)" : "");
            }
    return 0;
}
