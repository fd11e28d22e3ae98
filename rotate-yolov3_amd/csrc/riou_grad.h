// rotate-yolov3_amd/csrc/riou_grad.h -- rotated IoU of two boxes (cx, cy, w, h, a) AND its gradient with respect to the five
// parameters of the first box, one pair per lane, no polygon buffers.
//
// The reference has no rotated-IoU loss (model/loss.py:322 is the axis-aligned wh_iou); its rotated IoU is the shapely
// route of utils/utils.py:663-725 (get_rotated_coors + skewiou), forward only.  This is the build's differentiable form of
// that same geometric quantity, written around two boundary integrals instead of a clipped vertex list:
//
//   area(P n T)       = 1/2 * closed-curve integral of (x dy - y dx) over the boundary of P n T, and for convex P, T that
//                       boundary is (edges of P inside T) + (edges of T inside P): eight segment-vs-rectangle slab clips.
//   d area / d theta  = integral over (boundary of P) n T of (velocity of the boundary point under theta) . (outward normal)
//                       (Reynolds transport; T is fixed).  For a rectangle the normal velocities are constants or linear in
//                       the arc length, so with l_k = length of edge k inside T and [s0_k, s1_k] its tangential interval
//                       measured from the edge midpoint:
//                         d/dc  = sum_k n_k l_k            d/dw = (l_0 + l_2)/2        d/dh = (l_1 + l_3)/2
//                         d/da  = -1/2 sum_k (s1_k^2 - s0_k^2)
//   IoU = A / U, U = w h + w' h' - A   =>   dIoU = (dA (U + A) - A d(w h)) / U^2.
//
// Everything is evaluated in P's own frame (P = [-w/2, w/2] x [-h/2, h/2]); T enters through the relative angle and the
// rotated centre offset, so two identical boxes give exactly parallel edges (sin 0 = 0) and the tie rule of slab() counts
// each coincident edge once: IoU(A, A) = 1 exactly.
// Angle convention of get_rotated_coors (utils/utils.py:702-725): local (lx, ly) -> c + (lx cos a - ly sin a, lx sin a + ly cos a).
#pragma once
#include <hip/hip_runtime.h>

namespace ryolo_riou {

// intersect [s0, s1] with { s : |q0 + s dq| <= lim }.  An edge lying ON the slab boundary (dq == 0, |q0| == lim) belongs to
// the boundary of the intersection only when both rectangles lie on the same side of it, i.e. when the edge's outward normal
// (nq = its component along this slab axis) points out of the slab too: q0 * nq > 0.  Edges of P against T pass their
// normal; edges of T against P pass nq = 0 (never inside on a tie) -- so coincident edges of two identical boxes are
// integrated exactly once and the shared edge of two boxes touching from outside not at all.
__host__ __device__ __forceinline__ void slab(float q0, float dq, float lim, float nq, float &s0, float &s1) {
    if (dq == 0.f) {
        const float a = fabsf(q0);
        if (!(a < lim || (a == lim && q0 * nq > 0.f))) s1 = s0 - 1.f;
        return;
    }
    const float inv = 1.f / dq;
    const float ta = (-lim - q0) * inv, tb = (lim - q0) * inv;
    s0 = fmaxf(s0, fminf(ta, tb));
    s1 = fminf(s1, fmaxf(ta, tb));
}

// returns IoU(P, T); g[0..4] = d IoU / d (cx, cy, w, h, a) of P.  Degenerate boxes (w or h <= 0): 0 and zero gradient
// (skewiou returns 0 when either polygon has no area, utils/utils.py:671-673).
__host__ __device__ __forceinline__ float riou_fwd_bwd(const float *__restrict__ P, const float *__restrict__ T, float *__restrict__ g) {
    g[0] = g[1] = g[2] = g[3] = g[4] = 0.f;
    const float w = P[2], h = P[3], w2 = T[2], h2 = T[3];
    if (!(w > 0.f) || !(h > 0.f) || !(w2 > 0.f) || !(h2 > 0.f)) return 0.f;
    const float ha = 0.5f * w, hb = 0.5f * h, ha2 = 0.5f * w2, hb2 = 0.5f * h2;
    float sn, cs, sd, cd;
    sincosf(P[4], &sn, &cs);
    sincosf(T[4] - P[4], &sd, &cd);
    // T's centre in P's frame, T's axes in P's frame
    const float ox = T[0] - P[0], oy = T[1] - P[1];
    const float dx = ox * cs + oy * sn, dy = -ox * sn + oy * cs;
    const float ux = cd, uy = sd, vx = -sd, vy = cd;

    float A2 = 0.f;                       // twice the intersection area
    float len[4], mom = 0.f;              // l_k and sum_k (s1^2 - s0^2)
    // ---- edges of P (counter-clockwise: normals +x, +y, -x, -y) clipped to T (closed)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float p0x = (k < 2) ? ha : -ha, p0y = (k == 1 || k == 2) ? hb : -hb;      // start corner of edge k
        const float ex = (k == 1) ? -w : ((k == 3) ? w : 0.f), ey = (k == 0) ? h : ((k == 2) ? -h : 0.f);
        const float L = (k & 1) ? w : h;
        const float rx = p0x - dx, ry = p0y - dy;
        float s0 = 0.f, s1 = 1.f;
        const float nx = (k == 0) ? 1.f : ((k == 2) ? -1.f : 0.f), ny = (k == 1) ? 1.f : ((k == 3) ? -1.f : 0.f);   // outward normal
        slab(rx * ux + ry * uy, ex * ux + ey * uy, ha2, nx * ux + ny * uy, s0, s1);
        slab(rx * vx + ry * vy, ex * vx + ey * vy, hb2, nx * vx + ny * vy, s0, s1);
        len[k] = 0.f;
        if (s1 > s0) {
            const float ax = p0x + s0 * ex, ay = p0y + s0 * ey, bx = p0x + s1 * ex, by = p0y + s1 * ey;
            A2 += ax * by - ay * bx;
            len[k] = (s1 - s0) * L;
            const float t0 = (s0 - 0.5f) * L, t1 = (s1 - 0.5f) * L;
            mom += t1 * t1 - t0 * t0;
        }
    }
    // ---- edges of T (counter-clockwise) clipped to P (open at coincident edges)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float su = (k < 2) ? ha2 : -ha2, sv = (k == 1 || k == 2) ? hb2 : -hb2;
        const float p0x = dx + su * ux + sv * vx, p0y = dy + su * uy + sv * vy;
        // edge directions: +v', -u', -v', +u' scaled by the edge length
        const float ex = (k == 0) ? h2 * vx : ((k == 1) ? -w2 * ux : ((k == 2) ? -h2 * vx : w2 * ux));
        const float ey = (k == 0) ? h2 * vy : ((k == 1) ? -w2 * uy : ((k == 2) ? -h2 * vy : w2 * uy));
        float s0 = 0.f, s1 = 1.f;
        slab(p0x, ex, ha, 0.f, s0, s1);
        slab(p0y, ey, hb, 0.f, s0, s1);
        if (s1 > s0) {
            const float ax = p0x + s0 * ex, ay = p0y + s0 * ey, bx = p0x + s1 * ex, by = p0y + s1 * ey;
            A2 += ax * by - ay * bx;
        }
    }
    const float ap = w * h, at = w2 * h2;
    float A = fminf(fmaxf(0.5f * A2, 0.f), fminf(ap, at));
    const float U = ap + at - A;
    if (!(U > 0.f)) return 0.f;
    const float iou = A / U;
    // d A in P's frame, centre part rotated back to image axes
    const float dlx = len[0] - len[2], dly = len[1] - len[3];
    const float dA0 = dlx * cs - dly * sn, dA1 = dlx * sn + dly * cs;
    const float dA2 = 0.5f * (len[0] + len[2]), dA3 = 0.5f * (len[1] + len[3]);
    const float dA4 = -0.5f * mom;
    const float k1 = (U + A) / (U * U), k2 = A / (U * U);
    g[0] = dA0 * k1;
    g[1] = dA1 * k1;
    g[2] = dA2 * k1 - k2 * h;
    g[3] = dA3 * k1 - k2 * w;
    g[4] = dA4 * k1;
    return iou;
}

}  // namespace ryolo_riou
