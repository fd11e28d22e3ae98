// which CUs does a CU-masked stream run on?  hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.cpp -o scratch/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
__global__ void probe(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    // stay resident a little so that the launch spreads over every CU the mask allows
    for (int i = 0; i < 20; i++) __builtin_amdgcn_s_sleep(127);
}
static void run(const char *name, const unsigned *mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int nb = 4096;
    unsigned *d, *h = new unsigned[nb * 2];
    hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h, d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, int>> per;   // xcc -> (se,sh,cu) -> count
    for (int i = 0; i < nb; i++) per[h[2 * i + 1] & 15][(h[2 * i] >> 8) & 0xff]++;
    int total = 0;
    printf("%s:", name);
    for (auto &x : per) { printf(" xcc%u:%zu", x.first, x.second.size()); total += (int)x.second.size(); }
    printf("  = %d CUs\n", total);
    unsigned got[8];
    if (hipExtStreamGetCUMask(s, 8, got) == hipSuccess) printf("    mask read back: %08x %08x %08x %08x %08x %08x %08x %08x\n", got[0], got[1], got[2], got[3], got[4], got[5], got[6], got[7]);
    hipFree(d); delete[] h; hipStreamDestroy(s);
}
int main() {
    unsigned m[8];
    auto lowbits = [&](int n) { memset(m, 0, sizeof m); for (int b = 0; b < n; b++) m[b / 32] |= 1u << (b % 32); };
    for (int n : {8, 16, 32, 48, 64, 96, 128, 192, 224, 256}) { lowbits(n); char nm[64]; snprintf(nm, 64, "low %d bits", n); run(nm, m); }
    // k low bits in each 32-bit word
    for (int k : {4, 8, 16}) { for (int w = 0; w < 8; w++) m[w] = (k == 32 ? 0xffffffffu : ((1u << k) - 1)); char nm[64]; snprintf(nm, 64, "%d low bits per word", k); run(nm, m); }
    // complement layouts
    for (int n : {32, 64}) { lowbits(256); for (int b = 0; b < n; b++) m[b / 32] &= ~(1u << (b % 32)); char nm[64]; snprintf(nm, 64, "all but low %d bits", n); run(nm, m); }
    for (int k : {8}) { for (int w = 0; w < 8; w++) m[w] = ~((1u << k) - 1); char nm[64]; snprintf(nm, 64, "all but %d low bits per word", k); run(nm, m); }
    // every 4th bit
    memset(m, 0, sizeof m); for (int b = 0; b < 256; b += 4) m[b / 32] |= 1u << (b % 32); run("every 4th bit", m);
    return 0;
}
