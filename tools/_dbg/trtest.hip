#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr_elems, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    int a = addr_elems[threadIdx.x];
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int h[64]; short o[256]; int* d; short* dout;
    hipMalloc(&d, 256); hipMalloc(&dout, 512);
    for (int variant = 0; variant < 2; ++variant) {
        for (int l = 0; l < 64; ++l) h[l] = variant == 0 ? l * 4 : (l * 100);   // v0: contiguous 8B per lane; v1: 100-elem stride
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d, dout);
        hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
        printf("variant %d\n", variant);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    }
    return 0;
}
