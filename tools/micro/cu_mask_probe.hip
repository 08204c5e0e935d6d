// Which physical CUs does a hipExtStreamCreateWithCUMask bit enable on MI355X?  Launches many short workgroups on a masked stream and
// records (XCC_ID, SE_ID, CU_ID) of each; prints the enabled set per mask.   hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <map>

__global__ void probe(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char* name, const std::vector<int>& bits, int ncu) {
    const int nwords = (ncu + 31) / 32;
    std::vector<uint32_t> mask(nwords, 0);
    for (int b : bits) mask[b / 32] |= 1u << (b % 32);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, nwords, mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int nwg = 8192;
    uint32_t* d; hipMalloc(&d, nwg * 8); hipMemsetAsync(d, 0xff, nwg * 8, st);
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(64), 0, st, d, 20000);
    hipStreamSynchronize(st);
    std::vector<uint32_t> h(2 * nwg); hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per_xcc;     // xcc -> set of (se << 4 | cu)
    for (int i = 0; i < nwg; ++i) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert((se << 5) | (sh << 4) | cu);
    }
    int total = 0;
    printf("%s (%zu bits):", name, bits.size());
    for (auto& kv : per_xcc) {
        total += (int)kv.second.size();
        printf("  xcc%d: %zu CUs [", kv.first, kv.second.size());
        std::map<int, int> per_se;
        for (int v : kv.second) per_se[v >> 5]++;
        for (auto& s : per_se) printf("se%d:%d ", s.first, s.second);
        printf("]");
    }
    printf("  total %d\n", total);
    hipFree(d); hipStreamDestroy(st);
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    printf("CUs %d\n", ncu);
    auto range = [](int a, int b, int step = 1) { std::vector<int> v; for (int i = a; i < b; i += step) v.push_back(i); return v; };
    run("all", range(0, ncu), ncu);
    run("bits 0..31", range(0, 32), ncu);
    run("bits 0..7", range(0, 8), ncu);
    run("bits 224..255", range(224, 256), ncu);
    run("every 8th", range(0, ncu, 8), ncu);
    run("every 8th + 1", range(1, ncu, 8), ncu);
    run("bits 0..223", range(0, 224), ncu);
    run("bit 0", range(0, 1), ncu);
    run("bit 1", range(1, 2), ncu);
    run("bit 8", range(8, 9), ncu);
    run("bit 32", range(32, 33), ncu);
    run("bit 64", range(64, 65), ncu);
    return 0;
}
