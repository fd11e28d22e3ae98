// rotate-yolov3_amd/csrc/yolo_decode.h -- the per-row arithmetic of YOLOLayer.forward (model/models.py:198-221), shared by the stand-alone
// decode kernels (yolo.hip) and the head conv that decodes from its own accumulators (conv_pw.hip MODE 4): one definition, the same
// instructions in both, so the fused path's rows are the stand-alone kernel's rows bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace ryolo_detail {

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// A row of `no` floats to global memory.  io / p rows are 28 B apart (no = 7): 4-byte aligned only, so the compiler splits a plain
// copy into seven dword stores per row -- 14 store instructions per decoded row with 4 B per lane each.  The types below carry their
// true (4-byte) alignment, which global memory supports: one 16-B and one 12-B store per row.
typedef float f32x4_u4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3_u4 __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ void store_row(float *__restrict__ dst, const float *v, int no) {
    if (no == 7) {
        *(f32x4_u4 *)dst = f32x4_u4{v[0], v[1], v[2], v[3]};
        *(f32x3_u4 *)(dst + 4) = f32x3_u4{v[4], v[5], v[6]};
    } else {
        for (int k = 0; k < no; k++) dst[k] = v[k];
    }
}

template <int MAXNO = 96>
__device__ __forceinline__ void decode_row(const float *v, int no, int x, int y, float aw, float ah, float aa,
                                           float stride, float cf, int arc, float *__restrict__ dst) {
    float o[MAXNO];
    float bx = (sigmoidf(v[0]) + (float)x) * stride;
    float by = (sigmoidf(v[1]) + (float)y) * stride;
    float bw = (expf(v[2]) * aw) * stride;
    float bh = (expf(v[3]) * ah) * stride;
    const float ba = atanf(v[4]) + aa;
    bh = bh / cf;
    bw = bw - bh * (cf - 1.f);
    o[0] = bx; o[1] = by; o[2] = bw; o[3] = bh; o[4] = ba;
    if (arc == 0) {
#pragma unroll
        for (int k = 5; k < MAXNO; k++)
            if (k < no) o[k] = sigmoidf(v[k]);
    } else if (arc == 1) {
        o[5] = 1.f;
#pragma unroll
        for (int k = 6; k < MAXNO; k++)
            if (k < no) o[k] = sigmoidf(v[k]);
    } else {
        float mx = -3.4e38f;
        for (int k = 5; k < no; k++) mx = fmaxf(mx, v[k]);
        float sum = 0.f;
        for (int k = 5; k < no; k++) sum += expf(v[k] - mx);
        for (int k = 6; k < no; k++) o[k] = expf(v[k] - mx) / sum;
        o[5] = 1.f;
    }
    if (no == 7) o[6] = 1.f;
    store_row(dst, o, no);
}

}  // namespace ryolo_detail
