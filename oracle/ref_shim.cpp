// oracle/ref_shim.cpp -- C entry points around the REFERENCE's own rotated-IoU arithmetic.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file is never compiled on its own: oracle/Makefile
// pipes lines [#define DIVUP .. end of devRotateIoU] of
//   /root/reference/utils/nms/src/rotate_polygon_nms_kernel.cu   (kernel.cu:19-260)
// into g++ (with -D__device__= so the CUDA qualifier reads as nothing) and appends this shim, so the
// functions trangle_area/area/reorder_pts/inter2line/in_rect/inter_pts/convert_region/inter/devRotateIoU
// below are the reference's text, compiled where it lies; no reference source is written into the repo.
// The rest of kernel.cu (THC allocation, the __global__ tile kernel, the torch glue) needs headers this
// image lacks and is NOT built; the driver below restates only the tile/scan structure (kernel.cu:262-308,
// :358-383) around the reference's devRotateIoU.
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>

extern "C" {

float ref_rotate_iou(const float* r1, const float* r2) { return devRotateIoU(r1, r2); }

void ref_convert_region(float* pts, const float* region) { convert_region(pts, region); }

void ref_riou_matrix(const float* b1, int n1, int s1, const float* b2, int n2, int s2, float* out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n1; i++)
        for (int j = 0; j < n2; j++)
            out[(size_t)i * n2 + j] = devRotateIoU(b1 + (size_t)i * s1, b2 + (size_t)j * s2);
}

static inline uint32_t score_key(float s) {
    uint32_t b; std::memcpy(&b, &s, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// greedy NMS: stable descending sort, mask bits = devRotateIoU(box_i, box_j) > thr with the
// higher-scored box as FIRST argument (kernel.cu:301), lazily per kept row, ascending original indices out.
int ref_rnms(const float* dets, int n, int stride, float thr, int64_t* keep_out) {
    if (n <= 0) return 0;
    std::vector<int32_t> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return score_key(dets[(size_t)a * stride + 5]) > score_key(dets[(size_t)b * stride + 5]);
    });
    std::vector<float> sorted((size_t)n * 6);
    for (int i = 0; i < n; i++) std::memcpy(&sorted[(size_t)i * 6], dets + (size_t)order[i] * stride, 24);
    std::vector<unsigned char> removed(n, 0);
    int k = 0;
    for (int i = 0; i < n; i++) {
        if (removed[i]) continue;
        keep_out[k++] = order[i];
        const float* bi = &sorted[(size_t)i * 6];
#pragma omp parallel for schedule(static) if (n - i > 2048)
        for (int j = i + 1; j < n; j++) {
            if (removed[j]) continue;
            if (devRotateIoU(bi, &sorted[(size_t)j * 6]) > thr) removed[j] = 1;
        }
    }
    std::sort(keep_out, keep_out + k);
    return k;
}

}  // extern "C"
