// rotate-yolov3_amd/csrc/skewiou.hip -- rotated IoU of the EVALUATION path: the value the reference's skew_bbox_iou returns
// (utils/utils.py:290-320 -> get_rotated_coors :702-725 -> skewiou :663-699, shapely polygon intersection in fp64), which
// test.py:146 thresholds to mark a prediction correct.  It is NOT the arithmetic of the native NMS kernel (rnms.hip keeps
// that one bit for bit, including its behaviour on coincident boxes, where it can report 1/3 for IoU(A, A)): here the four
// corners are computed in fp64 with get_rotated_coors' rotation matrix (OpenCV getRotationMatrix2D of angle -a about the
// centre) and the intersection of the two convex quadrilaterals is an fp64 Sutherland-Hodgman clip + shoelace areas --
// the exact geometry, to fp64 rounding.  One pair per lane; the polygon (<= 8 vertices) lives in registers / scratch-free
// local arrays with fully unrolled loops.  Replaces the per-pair Python + shapely loop by one launch per image.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ryolo.h"

namespace {

struct Quad { double x[4], y[4]; };

__device__ __forceinline__ void corners(const float *b, Quad &q) {
    // get_rotated_coors: (xmin,ymin) (xmin,ymax) (xmax,ymax) (xmax,ymin) mapped by R = getRotationMatrix2D((cx,cy), -a*180/pi, 1)
    //   R = [[al, be, (1-al)cx - be*cy], [-be, al, be*cx + (1-al)cy]], al = cos(-a), be = sin(-a)
    const double cx = b[0], cy = b[1], w = b[2], h = b[3], a = b[4];
    const double al = cos(-a), be = sin(-a);
    const double r02 = (1.0 - al) * cx - be * cy, r12 = be * cx + (1.0 - al) * cy;
    const double xs[2] = {cx - w * 0.5, cx + w * 0.5}, ys[2] = {cy - h * 0.5, cy + h * 0.5};
    const int ix[4] = {0, 0, 1, 1}, iy[4] = {0, 1, 1, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double tx = xs[ix[k]], ty = ys[iy[k]];
        q.x[k] = tx * al + ty * be + r02;
        q.y[k] = -tx * be + ty * al + r12;
    }
}

__device__ __forceinline__ double quad_area2(const Quad &q) {   // twice the signed area
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int n = (k + 1) & 3;
        s += q.x[k] * q.y[n] - q.x[n] * q.y[k];
    }
    return s;
}

__device__ double skew_iou(const float *b1, const float *b2) {
    Quad p, c;
    corners(b1, p);
    corners(b2, c);
    double a1 = quad_area2(p), a2 = quad_area2(c);
    if (a1 < 0) {      // make both counter-clockwise
        double t;
        t = p.x[1]; p.x[1] = p.x[3]; p.x[3] = t;
        t = p.y[1]; p.y[1] = p.y[3]; p.y[3] = t;
        a1 = -a1;
    }
    if (a2 < 0) {
        double t;
        t = c.x[1]; c.x[1] = c.x[3]; c.x[3] = t;
        t = c.y[1]; c.y[1] = c.y[3]; c.y[3] = t;
        a2 = -a2;
    }
    if (a1 == 0.0 || a2 == 0.0) return 0.0;                 // skewiou: "if poly1.area == 0 or poly2.area == 0: return 0"
    // Sutherland-Hodgman: clip p by the four half planes of c.  A convex quadrilateral clipped by 4 lines has <= 8 vertices.
    double px[8], py[8], qx[8], qy[8];
    int n = 4;
#pragma unroll
    for (int k = 0; k < 4; k++) { px[k] = p.x[k]; py[k] = p.y[k]; }
    for (int e = 0; e < 4 && n > 0; e++) {
        const double ax = c.x[e], ay = c.y[e];
        const double ex = c.x[(e + 1) & 3] - ax, ey = c.y[(e + 1) & 3] - ay;
        int m = 0;
        for (int j = 0; j < n; j++) {
            const int j2 = j + 1 == n ? 0 : j + 1;
            const double sp = ex * (py[j] - ay) - ey * (px[j] - ax);
            const double sq = ex * (py[j2] - ay) - ey * (px[j2] - ax);
            if (sp >= 0.0 && m < 8) { qx[m] = px[j]; qy[m] = py[j]; m++; }
            if (((sp > 0.0 && sq < 0.0) || (sp < 0.0 && sq > 0.0)) && m < 8) {
                const double t = sp / (sp - sq);
                qx[m] = px[j] + t * (px[j2] - px[j]);
                qy[m] = py[j] + t * (py[j2] - py[j]);
                m++;
            }
        }
        n = m;
        for (int j = 0; j < n; j++) { px[j] = qx[j]; py[j] = qy[j]; }
    }
    double inter2 = 0.0;
    for (int j = 0; j < n; j++) {
        const int j2 = j + 1 == n ? 0 : j + 1;
        inter2 += px[j] * py[j2] - px[j2] * py[j];
    }
    inter2 = fabs(inter2);
    const double uni2 = a1 + a2 - inter2;
    if (uni2 == 0.0) return 0.0;
    return inter2 / uni2;
}

__global__ void skew_iou_pairs_kernel(const float *b1, int s1, const float *b2, int s2, int n, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)skew_iou(b1 + (size_t)i * s1, b2 + (size_t)i * s2);
}

__global__ void skew_iou_matrix_kernel(const float *b1, int n1, int s1, const float *b2, int n2, int s2, float *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < n2) out[(size_t)i * n2 + j] = (float)skew_iou(b1 + (size_t)i * s1, b2 + (size_t)j * s2);
}

}  // namespace

extern "C" {

int ryolo_skew_iou_pairs(const float *b1, int stride1, const float *b2, int stride2, int n, float *out, void *stream_) {
    if (n < 0) return RYOLO_EINVAL;
    if (n == 0) return RYOLO_OK;
    if (!b1 || !b2 || !out || stride1 < 5 || stride2 < 5) return RYOLO_EINVAL;
    hipLaunchKernelGGL(skew_iou_pairs_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream_, b1, stride1, b2, stride2,
                       n, out);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

int ryolo_skew_iou_matrix(const float *b1, int n1, int stride1, const float *b2, int n2, int stride2, float *out, void *stream_) {
    if (n1 < 0 || n2 < 0) return RYOLO_EINVAL;
    if (n1 == 0 || n2 == 0) return RYOLO_OK;
    if (!b1 || !b2 || !out || stride1 < 5 || stride2 < 5 || n1 > 65535) return RYOLO_EINVAL;
    hipLaunchKernelGGL(skew_iou_matrix_kernel, dim3((n2 + 127) / 128, n1), dim3(128), 0, (hipStream_t)stream_, b1, n1, stride1, b2,
                       n2, stride2, out);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

}  // extern "C"
