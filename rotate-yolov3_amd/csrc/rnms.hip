// rotate-yolov3_amd/csrc/rnms.hip -- rotated IoU + greedy rotated NMS for gfx950 (MI355X), wave64.
//
// Replaces utils/nms/src/rotate_polygon_nms_kernel.cu of the reference (tile kernel :262-308, IoU :22-260,
// host driver :323-384).  Not a translation: the reference evaluates every one of the n^2 ordered pairs with
// one divergent thread per pair, copies the n x n/64 bit matrix to the host and scans it there.  Here:
//
//   K0  keys      one thread per box: order-preserving radix key of the score                    (HBM-bound, 24 B/box)
//       sort      hipcub radix sort of (key, index) pairs (stable)                               (library; 8 B/box/pass)
//   K1  corners   one thread per SORTED box: corners (fp64 sincos, correctly rounded to fp32),
//                 area, centre, padded circumradius -> three float4 SoA arrays                   (48 B/box written)
//   K2  mask      ONE WAVEFRONT PER ONE OR TWO 64x64 TILES of the upper triangle (two consecutive column tiles of a block row from
//                 128 block rows on).  Phase 1: every lane owns one column box and runs the 6-flop bounding-circle reject against the
//                 64 row boxes (row box broadcast with v_readlane); survivors are compacted with ballot+mbcnt into a 128-entry
//                 LDS ring.  Phase 1b: whenever 64 survivors are queued, 64 lanes run a separating-axis test (four edge
//                 directions, margin) on one pair each and compact what is left into a second ring.  Phase 2: whenever 64
//                 candidates are queued there, all 64 lanes run the exact polygon IoU on one candidate each, so the divergent
//                 per-pair code runs on dense wavefronts instead of ~4 %-occupied ones.  A ring entry names (tile, row,
//                 column); a pair's two boxes are fetched BY INDEX from K1's arrays (three 16-B loads per box; rounds 2-5 pulled
//                 them out of the owning lanes with eighteen ds_bpermute per pair and stage: profiles/r06_pmc_rnms.txt).  Result
//                 bits are OR-ed into 64 TRANSPOSED words per tile (word c = which rows suppress column c).  Every tile gets a
//                 16-byte SUMMARY: the number of suppressing pairs and, up to 7 of them, the pairs themselves (row << 6 | col);
//                 only tiles with more pairs also store the coalesced 512-B word tile.
//   K3  scan      one 1024-thread workgroup, PANELS of 4 block rows per barrier.  Wave 0 carries the serial chain:
//                 it resolves the panel's diagonal tiles (one ballot per row THAT HAS a suppressing pair, not 64 steps)
//                 and folds the panel's rows into the columns of this and the next panel, from tiles the other waves
//                 staged in LDS one step earlier.  Waves 1-15 run one panel behind: one LANE per (row, column) tile,
//                 16-byte summaries prefetched a step ahead, each listed pair tested against the row's keep word and
//                 OR-ed into the column's suppression word with an LDS atomic (dense tiles: one ballot per tile).
//                 Tail: scatter keep flags to original indices and compact them in ascending order.
//
// Bit-exactness contract (checked in tests/test_rnms_gpu.py against oracle/riou_oracle.c, which is pinned to the
// reference arithmetic): every fp32 operation of devRotateIoU is reproduced in the reference's order with
// IEEE + - * / sqrt, no FMA contraction (this TU is built with -ffp-contract=off and correctly rounded
// divide/sqrt), corners from the shared "correctly rounded sincos" definition, 24-slot point buffers.
// The bounding-circle reject only skips pairs the reference arithmetic scores exactly 0 (or NaN), i.e. never > thr >= 0:
//   * it is switched off for thr < 0 (there IoU == 0 already suppresses);
//   * circles separated with the padded radii put every corner of one box a positive distance from the other box, far
//     above the rounding of in_rect's dot products (kernel.cu:134-160) -- no corner is reported inside;
//   * inter2line (kernel.cu:90-132) can then only report a point for two edges that are COLLINEAR to within its rounding
//     noise (|error of a triangle area| <= 4u D^2, u = 2^-24, D = largest point distance), and its points lie on that line;
//     two boxes whose edges are all longer than 2e-3 D have at most one such edge pair per direction, so at most 2 points
//     come back and area() of fewer than 3 points is exactly 0 (kernel.cu:26-33, :232-249);
//   * boxes that break the premise -- an edge of length 0 (w or h == 0: in_rect degenerates to 0 >= 0 and reports EVERY
//     point of the plane inside, the reference then returns area/0 = inf and such a box suppresses boxes anywhere), an
//     edge shorter than 2e-3 of the call's extent (bounding box of all centres + the largest diagonal), NaN/inf fields --
//     get an infinite radius in K1 and are never rejected: all their pairs take the exact path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <mutex>

#include "../../include/ryolo.h"
#include "conv_common.h"      // the tuning-switch registry (ryolo_detail::tune)

#pragma clang fp contract(off)

namespace {

constexpr int WAVE = 64;
constexpr int MAX_PTS = 24;   // definition (b) of oracle/riou_oracle.c
constexpr int FAST_PTS = 8;   // LDS fast path capacity; more points -> generic path

// ------------------------------------------------------------------------------------------------
// correctly rounded sincos (definition (a)): fp64 Cody-Waite + Taylor/Horner, plain IEEE ops only.
__device__ __forceinline__ void sincosf_cr(float a, float &s_out, float &c_out) {
    double x = (double)a;
    if (!(fabs(x) <= 1.0e6)) {
        if (!(fabs(x) <= 3.5e38)) {
            s_out = (float)(x - x);
            c_out = (float)(x - x);
            return;
        }
        x = x - 6.28318530717958623200e+00 * trunc(x / 6.28318530717958623200e+00);
    }
    double kd = rint(x * 6.36619772367581382433e-01);
    double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    double z = r * r;
    double ps = 1.0 / 355687428096000.0;
    ps = -1.0 / 1307674368000.0 + z * ps;
    ps = 1.0 / 6227020800.0 + z * ps;
    ps = -1.0 / 39916800.0 + z * ps;
    ps = 1.0 / 362880.0 + z * ps;
    ps = -1.0 / 5040.0 + z * ps;
    ps = 1.0 / 120.0 + z * ps;
    ps = -1.0 / 6.0 + z * ps;
    double sr = r + r * (z * ps);
    double pc = 1.0 / 20922789888000.0;
    pc = -1.0 / 87178291200.0 + z * pc;
    pc = 1.0 / 479001600.0 + z * pc;
    pc = -1.0 / 3628800.0 + z * pc;
    pc = 1.0 / 40320.0 + z * pc;
    pc = -1.0 / 720.0 + z * pc;
    pc = 1.0 / 24.0 + z * pc;
    pc = -0.5 + z * pc;
    double cr = 1.0 + z * pc;
    int q = (int)(((long long)kd) & 3);
    double s, c;
    if (q == 0) { s = sr; c = cr; }
    else if (q == 1) { s = cr; c = -sr; }
    else if (q == 2) { s = -sr; c = -cr; }
    else { s = -cr; c = sr; }
    s_out = (float)s;
    c_out = (float)c;
}

// corners in the reference's pts[] order (convert_region, kernel.cu:196-229: written in reverse)
struct Quad {
    float x[4], y[4];
};

__device__ __forceinline__ void convert_region(const float cx, const float cy, const float w, const float h,
                                               const float angle, Quad &q) {
    float a_sin, a_cos;
    sincosf_cr(angle, a_sin, a_cos);
    const float px[4] = {-w / 2, w / 2, w / 2, -w / 2};
    const float py[4] = {-h / 2, -h / 2, h / 2, h / 2};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        q.x[3 - i] = a_cos * px[i] - a_sin * py[i] + cx;
        q.y[3 - i] = a_sin * px[i] + a_cos * py[i] + cy;
    }
}

__device__ __forceinline__ float tri_area(float ax, float ay, float bx, float by, float cx, float cy) {
    return ((ax - cx) * (by - cy) - (ay - cy) * (bx - cx)) * 0.5f;   // "/ 2.0" of kernel.cu:23, exact
}

// in_rect (kernel.cu:134-160) with the rectangle-only terms hoisted (same values, computed once)
struct RectFrame {
    float ax, ay, ab0, ab1, ad0, ad1, abab, adad;
};
__device__ __forceinline__ RectFrame make_frame(const Quad &q) {
    RectFrame f;
    f.ax = q.x[0]; f.ay = q.y[0];
    f.ab0 = q.x[1] - q.x[0]; f.ab1 = q.y[1] - q.y[0];
    f.ad0 = q.x[3] - q.x[0]; f.ad1 = q.y[3] - q.y[0];
    f.abab = f.ab0 * f.ab0 + f.ab1 * f.ab1;
    f.adad = f.ad0 * f.ad0 + f.ad1 * f.ad1;
    return f;
}
__device__ __forceinline__ bool in_rect(float px, float py, const RectFrame &f) {
    const float ap0 = px - f.ax, ap1 = py - f.ay;
    const float abap = f.ab0 * ap0 + f.ab1 * ap1;
    const float adap = f.ad0 * ap0 + f.ad1 * ap1;
    return f.abab >= abap && abap >= 0 && f.adad >= adap && adap >= 0;
}

// Point sink abstraction: the candidate-point list of inter_pts (kernel.cu:162-194).
// LdsSink: FAST_PTS slots per lane in LDS, layout [slot][lane] (conflict-free: lane = bank).
struct LdsSink {
    float *bx, *by, *bk;   // wave-private bases, already offset by lane
    int n;
    __device__ __forceinline__ void push(float x, float y) {
        if (n < FAST_PTS) { bx[n * WAVE] = x; by[n * WAVE] = y; }
        n++;
    }
    __device__ __forceinline__ float X(int i) const { return bx[i * WAVE]; }
    __device__ __forceinline__ float Y(int i) const { return by[i * WAVE]; }
    __device__ __forceinline__ float K(int i) const { return bk[i * WAVE]; }
    __device__ __forceinline__ void setX(int i, float v) { bx[i * WAVE] = v; }
    __device__ __forceinline__ void setY(int i, float v) { by[i * WAVE] = v; }
    __device__ __forceinline__ void setK(int i, float v) { bk[i * WAVE] = v; }
};
// LocalSink: MAX_PTS slots in private memory (generic path: pairs kernels and >FAST_PTS overflow)
struct LocalSink {
    float ax[MAX_PTS], ay[MAX_PTS], ak[MAX_PTS];
    int n;
    __device__ __forceinline__ void push(float x, float y) {
        if (n < MAX_PTS) { ax[n] = x; ay[n] = y; }
        n++;
    }
    __device__ __forceinline__ float X(int i) const { return ax[i]; }
    __device__ __forceinline__ float Y(int i) const { return ay[i]; }
    __device__ __forceinline__ float K(int i) const { return ak[i]; }
    __device__ __forceinline__ void setX(int i, float v) { ax[i] = v; }
    __device__ __forceinline__ void setY(int i, float v) { ay[i] = v; }
    __device__ __forceinline__ void setK(int i, float v) { ak[i] = v; }
};

// inter_pts (kernel.cu:162-194): vertices first (interleaved box1/box2), then the 16 edge pairs (i outer).
// The 48 triangle areas of the reference collapse to 32 distinct expressions:
//   area_abc(i,j) = T1[i][j] = tri(p1_i, p1_{i+1}, p2_j);   area_abd(i,j) = T1[i][j+1]
//   area_cda(i,j) = T2[j][i] = tri(p2_j, p2_{j+1}, p1_i);   area_cdb     = area_cda + area_abc - area_abd
template <class Sink>
__device__ __forceinline__ void collect_points(const Quad &p1, const Quad &p2, Sink &s) {
    const RectFrame f1 = make_frame(p1), f2 = make_frame(p2);
    s.n = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (in_rect(p1.x[i], p1.y[i], f2)) s.push(p1.x[i], p1.y[i]);
        if (in_rect(p2.x[i], p2.y[i], f1)) s.push(p2.x[i], p2.y[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int i1 = (i + 1) & 3;
        const float ax = p1.x[i], ay = p1.y[i], bx = p1.x[i1], by = p1.y[i1];
        float t1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) t1[j] = tri_area(ax, ay, bx, by, p2.x[j], p2.y[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int j1 = (j + 1) & 3;
            const float area_abc = t1[j], area_abd = t1[j1];
            if (area_abc * area_abd >= 0) continue;
            const float area_cda = tri_area(p2.x[j], p2.y[j], p2.x[j1], p2.y[j1], ax, ay);
            const float area_cdb = area_cda + area_abc - area_abd;
            if (area_cda * area_cdb >= 0) continue;
            const float t = area_cda / (area_abd - area_abc);
            const float dx = t * (bx - ax);
            const float dy = t * (by - ay);
            s.push(ax + dx, ay + dy);
        }
    }
}

// reorder_pts (kernel.cu:35-89) + area (kernel.cu:26-33) on n points held by the sink
template <class Sink>
__device__ __forceinline__ float order_and_area(Sink &s, int n) {
    if (n < 3) return 0.0f;   // the fan loop of kernel.cu:29 does not run; reordering has no effect on 0
    float c0 = 0.0f, c1 = 0.0f;
    for (int i = 0; i < n; i++) { c0 += s.X(i); c1 += s.Y(i); }
    c0 /= (float)n;
    c1 /= (float)n;
    for (int i = 0; i < n; i++) {
        float v0 = s.X(i) - c0, v1 = s.Y(i) - c1;
        const float d = sqrtf(v0 * v0 + v1 * v1);
        v0 = v0 / d;
        v1 = v1 / d;
        if (v1 < 0) v0 = -2 - v0;
        s.setK(i, v0);
    }
    for (int i = 1; i < n; ++i) {
        if (s.K(i - 1) > s.K(i)) {
            const float temp = s.K(i), tx = s.X(i), ty = s.Y(i);
            int j = i;
            while (j > 0 && s.K(j - 1) > temp) {
                s.setK(j, s.K(j - 1));
                s.setX(j, s.X(j - 1));
                s.setY(j, s.Y(j - 1));
                j--;
            }
            s.setK(j, temp);
            s.setX(j, tx);
            s.setY(j, ty);
        }
    }
    float area = 0.0f;
    const float x0 = s.X(0), y0 = s.Y(0);
    for (int i = 0; i < n - 2; i++)
        area += fabsf(tri_area(x0, y0, s.X(i + 1), s.Y(i + 1), s.X(i + 2), s.Y(i + 2)));
    return area;
}

// devRotateIoU (kernel.cu:251-260), generic path (private 24-slot buffers)
__device__ __noinline__ float riou_generic(const Quad &p1, float area1, const Quad &p2, float area2) {
    LocalSink s;
    collect_points(p1, p2, s);
    const int n = s.n < MAX_PTS ? s.n : MAX_PTS;
    const float area_inter = order_and_area(s, n);
    return area_inter / (area1 + area2 - area_inter);
}

// fast path: FAST_PTS LDS slots; returns false (and leaves `iou` untouched) when more points showed up
__device__ __forceinline__ bool riou_fast(const Quad &p1, float area1, const Quad &p2, float area2,
                                          float *bx, float *by, float *bk, float &iou) {
    LdsSink s{bx, by, bk, 0};
    collect_points(p1, p2, s);
    if (s.n > FAST_PTS) return false;
    const float area_inter = order_and_area(s, s.n);
    iou = area_inter / (area1 + area2 - area_inter);
    return true;
}

__device__ __forceinline__ uint32_t score_key(float s) {
    const uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ------------------------------------------------------------------------------------------------ K0
// ascending sort of ~key == descending score; stable radix sort keeps the lower index first on ties
__global__ void rnms_keys_kernel(const float *__restrict__ dets, int n, int row_stride, uint32_t *keys,
                                 int32_t *idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ~score_key(dets[(size_t)i * row_stride + 5]);
    idx[i] = i;
}

// ------------------------------------------------------------------------------------------------ K1
// extent of the call for the reject's premise: ext[0..4] = order-preserving keys of max cx, max -cx, max cy, max -cy,
// max half diagonal (atomicMax from a zeroed buffer; NaNs skipped)
__global__ void rnms_extent_kernel(const float *__restrict__ dets, int n, int row_stride, uint32_t *__restrict__ ext) {
    uint32_t k[5] = {0u, 0u, 0u, 0u, 0u};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {   // few blocks: 5 atomics per wave
        const float *r = dets + (size_t)i * row_stride;
        const float cx = r[0], cy = r[1], hd = 0.5f * sqrtf(r[2] * r[2] + r[3] * r[3]);
        if (cx == cx) { k[0] = max(k[0], score_key(cx)); k[1] = max(k[1], score_key(-cx)); }
        if (cy == cy) { k[2] = max(k[2], score_key(cy)); k[3] = max(k[3], score_key(-cy)); }
        if (hd == hd) k[4] = max(k[4], score_key(hd));
    }
#pragma unroll
    for (int j = 0; j < 5; j++) {
        uint32_t v = k[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
        if ((threadIdx.x & (WAVE - 1)) == 0 && v) atomicMax(ext + j, v);
    }
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void rnms_corners_kernel(const float *__restrict__ dets, int n, int row_stride,
                                    const int32_t *__restrict__ order, const uint32_t *__restrict__ ext, float4 *P0,
                                    float4 *P1, float4 *AUX) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = dets + (size_t)(order ? order[i] : i) * row_stride;
    const float cx = r[0], cy = r[1], w = r[2], h = r[3], a = r[4];
    Quad q;
    convert_region(cx, cy, w, h, a, q);
    P0[i] = make_float4(q.x[0], q.y[0], q.x[1], q.y[1]);
    P1[i] = make_float4(q.x[2], q.y[2], q.x[3], q.y[3]);
    // padded circumradius: half diagonal, +1e-5 relative, +1e-5 * coordinate magnitude (>= 50x the corner
    // rounding error of convert_region); NaN/inf propagate and disable the reject for this box.
    const float hd = 0.5f * sqrtf(w * w + h * h);
    float rad = hd * 1.00001f + 1.0e-5f * (fabsf(cx) + fabsf(cy) + hd);
    // premise of the reject (file header): both edges of the fp32 corner quad (what in_rect / inter2line see) at least
    // 2e-3 of the call's extent; a 0 key means "no finite value seen" -> infinite extent -> nothing is rejected
    float dmax = __builtin_inff();
    if (ext[0] && ext[1] && ext[2] && ext[3] && ext[4]) {
        const float ex = key_to_float(ext[0]) + key_to_float(ext[1]), ey = key_to_float(ext[2]) + key_to_float(ext[3]);
        dmax = (sqrtf(ex * ex + ey * ey) + 2.f * key_to_float(ext[4])) * 1.001f;
    }
    const float abx = q.x[1] - q.x[0], aby = q.y[1] - q.y[0], adx = q.x[3] - q.x[0], ady = q.y[3] - q.y[0];
    const float l2 = fminf(abx * abx + aby * aby, adx * adx + ady * ady);
    if (!(l2 >= 4.0e-6f * dmax * dmax) || !(l2 > 0.f)) rad = __builtin_inff();
    AUX[i] = make_float4(cx, cy, rad, w * h);
}

// ------------------------------------------------------------------------------------------------ K2
constexpr int MASK_WAVES = 4;                       // waves per workgroup
constexpr int SUMM_MAX = 7;                         // pairs listed in a tile's 16-byte summary
constexpr unsigned SUMM_DENSE = 0xffffu;            // summary count of a tile stored as 64 column words
constexpr int QCAP = 128;                           // candidate rings

// tiles of the upper triangle in row-major order: tile id t <-> (rb, cb >= rb)
__device__ __forceinline__ long long tile_base(int rb, int W) { return (long long)rb * W - (long long)rb * (rb - 1) / 2; }

// The mask kernel.  A wave owns CT consecutive column tiles (64 x 64 pairs each) of one block row: bounding-circle reject on all 64 lanes
// per row, survivors compacted into an LDS ring, separating-axis test on dense lanes, survivors into a second ring, exact IoU on dense
// lanes; both rings are flushed once per CT tiles.  A ring entry names (column tile, row, column); the two boxes of a pair are fetched by
// index from the SoA arrays (three 16-B loads per box: L1 / L2 hits on the otherwise idle vector-memory pipe).
// Round 6 rebuilt this stage on its counters (profiles/r06_pmc_rnms.txt): rounds 2-5 kept the tile's 128 boxes in registers and fetched a
// pair with eighteen wave shuffles (ds_bpermute) per stage, one tile per wave -- the VALU issue port of every SIMD taken 99 % of the time,
// 1.24e8 LDS instructions and 2.2e7 cycles of LDS bank conflicts per launch.  Index loads: 50 000 boxes 3.07 -> 2.74 ms with one tile per
// wave, 2.59 ms with two (fewer partial ring flushes); equal at 500 - 4 096 boxes (profiles/r06_nms_multi_tile.txt).  The arithmetic per pair
// is the same code on the same operands: same bits.
// Jobs: grid.y = block row rb, (grid.x rotated by rb) * MASK_WAVES + wave = k, column tiles rb + k*CT ...; jobs past the row's end exit at once.
template <int CT>
struct __attribute__((aligned(16))) MaskMultiLds {
    float bx[FAST_PTS * WAVE];
    float by[FAST_PTS * WAVE];
    float bk[FAST_PTS * WAVE];
    unsigned long long colmask[CT][WAVE];
    unsigned short queue[QCAP];      // circle survivors: ct << 12 | row << 6 | col
    unsigned short queue2[QCAP];     // separating-axis survivors
};

template <int CT>
__global__ void __launch_bounds__(MASK_WAVES *WAVE)
rnms_mask_multi_kernel(int n, float thr, const float4 *__restrict__ P0, const float4 *__restrict__ P1,
                       const float4 *__restrict__ AUX, unsigned long long *__restrict__ tiles, uint4 *__restrict__ summ,
                       const int32_t *__restrict__ seg_off, long long seg_tile_stride,
                       unsigned long long *__restrict__ eval_counter, int allow_reject, int row_base) {
    static_assert(CT >= 1 && CT <= 4, "ring entries keep the column tile in two bits");
    __shared__ MaskMultiLds<CT> lds_all[MASK_WAVES];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x >> 6;
    if (seg_off) {             // segmented call: blockIdx.z = segment
        const int s = blockIdx.z;
        const int lo = seg_off[s];
        n = seg_off[s + 1] - lo;
        P0 += lo; P1 += lo; AUX += lo;
        tiles += (size_t)s * seg_tile_stride * WAVE;
        summ += (size_t)s * seg_tile_stride;
    }
    const int W = (n + WAVE - 1) / WAVE;
    const int rb = blockIdx.y + row_base;       // (row_base: the block rows of this launch start here -- ryolo_rnms splits large calls in two)
    // (grid.x rotated by the row: the jobs that exist have small k, and with grid.x a multiple of 8 the round-robin of workgroups over the
    //  8 XCDs would send every row's work to the same few XCDs -- W = 256 / 512 ran 6-22 % SLOWER than one tile per wave before this)
    const int kx = ((int)blockIdx.x + rb) % (int)gridDim.x;
    const int cb0 = rb + (kx * MASK_WAVES + wv) * CT;
    if (n <= 0 || rb >= W || cb0 >= W) return;          // (wave-uniform)
    const int nct = min(CT, W - cb0);
    MaskMultiLds<CT> &L = lds_all[wv];
    const int row0 = rb * WAVE;
    const int row_size = min(n - row0, WAVE);

    const float4 rowaux = AUX[min(row0 + lane, n - 1)];     // lane r: (cx, cy, padded radius, area) of row box r
    float4 colaux[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        colaux[ct] = AUX[min((cb0 + ct) * WAVE + lane, n - 1)];
        L.colmask[ct][lane] = 0ull;
    }
    float *bx = L.bx + lane, *by = L.by + lane, *bk = L.bk + lane;
    int head = 0, tail = 0, head2 = 0, tail2 = 0;         // wave-uniform ring indices

    auto load_pair = [&](unsigned e, Quad &q1, float &a1, float &rad1, Quad &q2, float &a2, float &rad2) __attribute__((always_inline)) {
        const int ct = (int)(e >> 12), r = (int)((e >> 6) & 63u), c = (int)(e & 63u);
        const int ri = row0 + r, ci = (cb0 + ct) * WAVE + c;      // (entries are in range by construction)
        const float4 r0 = P0[ri], r1 = P1[ri], ra = AUX[ri];
        const float4 c0 = P0[ci], c1 = P1[ci], ca = AUX[ci];
        q1.x[0] = r0.x; q1.y[0] = r0.y; q1.x[1] = r0.z; q1.y[1] = r0.w; q1.x[2] = r1.x; q1.y[2] = r1.y; q1.x[3] = r1.z; q1.y[3] = r1.w;
        q2.x[0] = c0.x; q2.y[0] = c0.y; q2.x[1] = c0.z; q2.y[1] = c0.w; q2.x[2] = c1.x; q2.y[2] = c1.y; q2.x[3] = c1.z; q2.y[3] = c1.w;
        a1 = ra.w; a2 = ca.w; rad1 = ra.z; rad2 = ca.z;
    };
    auto run_exact = [&](int count) __attribute__((always_inline)) {
        const bool active = lane < count;
        const unsigned e = L.queue2[(head2 + (active ? lane : 0)) & (QCAP - 1)];
        Quad q1, q2;
        float a1, a2, r1, r2;
        load_pair(e, q1, a1, r1, q2, a2, r2);     // box_i (higher score) is the FIRST argument, kernel.cu:301
        if (active) {
            float iou;
            if (!riou_fast(q1, a1, q2, a2, bx, by, bk, iou)) iou = riou_generic(q1, a1, q2, a2);
            if (iou > thr) atomicOr(&L.colmask[e >> 12][e & 63u], 1ull << ((e >> 6) & 63u));
        }
        if (eval_counter && lane == 0) atomicAdd(eval_counter, (unsigned long long)count);   // measurement only (bench.py)
        head2 += count;
    };
    // Second reject, on dense lanes: a separating axis among the four edge directions, with a margin of 1.5e-3 of the pair's
    // extent D = 2 (rad_i + rad_j) (every point of the two boxes lies within D of every other: the circles overlap here).
    // Premise and proof as for the circle reject (file header): the projections are in_rect's own dot products
    // (kernel.cu:134-160), so no corner of the far box is reported inside the near one, the near box's corners are >= 1e-3 D
    // away from the far box, and inter2line can only report points for edge pairs collinear within its noise (2.4e-4 D for
    // boxes that pass K1's edge guard) -- at most two of them, whose polygon has area exactly 0.  Boxes outside the premise
    // carry an infinite radius: D = inf makes the margin infinite and nothing is separated.  NaN compares false.
    auto separated = [&](const Quad &A, const Quad &B, float D) __attribute__((always_inline)) -> bool {
        bool sep = false;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i1 = k == 0 ? 1 : 3, i2 = k == 0 ? 3 : 1;
            const float ex = A.x[i1] - A.x[0], ey = A.y[i1] - A.y[0];
            const float ox = A.x[i2] - A.x[0], oy = A.y[i2] - A.y[0];
            const float ee = ex * ex + ey * ey;
            const float m = 1.5e-3f * D * fmaxf(fabsf(ex), fabsf(ey)) + fabsf(ex * ox + ey * oy);
            float tmin = 3.4e38f, tmax = -3.4e38f;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float t = (B.x[q] - A.x[0]) * ex + (B.y[q] - A.y[0]) * ey;
                tmin = fminf(tmin, t);
                tmax = fmaxf(tmax, t);
            }
            sep = sep || (tmin > ee + m) || (tmax < -m);
        }
        return sep;
    };
    auto run_sat = [&](int count) __attribute__((always_inline)) {
        const bool active = lane < count;
        const unsigned e = L.queue[(head + (active ? lane : 0)) & (QCAP - 1)];
        Quad q1, q2;
        float a1, a2, r1, r2;
        load_pair(e, q1, a1, r1, q2, a2, r2);
        const float D = 2.f * (r1 + r2);
        const bool keep = active && !(allow_reject && (separated(q1, q2, D) || separated(q2, q1, D)));
        const unsigned long long m = __ballot(keep);
        if (m) {
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (keep) L.queue2[(tail2 + pos) & (QCAP - 1)] = (unsigned short)e;
            tail2 += __popcll(m);
        }
        head += count;
        if (tail2 - head2 >= WAVE) run_exact(WAVE);
    };

#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        if (ct >= nct) break;                       // (wave-uniform)
        const int col_size = min(n - (cb0 + ct) * WAVE, WAVE);
        const bool col_ok = lane < col_size;
        const bool diag = (cb0 + ct == rb);
        for (int r = 0; r < row_size; r++) {
            const float rcx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rowaux.x), r));
            const float rcy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rowaux.y), r));
            const float rrad = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rowaux.z), r));
            const float dx = rcx - colaux[ct].x, dy = rcy - colaux[ct].y;
            const float d2 = dx * dx + dy * dy;
            const float lim = rrad + colaux[ct].z;
            const bool reject = allow_reject && d2 > lim * lim;   // NaN or an infinite radius anywhere -> not rejected
            const bool cand = col_ok && !reject && (!diag || lane > r);
            const unsigned long long m = __ballot(cand);
            if (m) {
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                if (cand) L.queue[(tail + pos) & (QCAP - 1)] = (unsigned short)((ct << 12) | (r << 6) | lane);
                tail += __popcll(m);
                if (tail - head >= WAVE) run_sat(WAVE);
            }
        }
    }
    if (tail - head > 0) run_sat(tail - head);
    if (tail2 - head2 > 0) run_exact(tail2 - head2);

    // tile summary, one per column tile: halfword 0 = number of suppressing pairs (SUMM_DENSE: more than SUMM_MAX, the word tile is stored),
    // halfwords 1..7 = the pairs as row << 6 | col
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        if (ct >= nct) break;
        const long long t = tile_base(rb, W) + (cb0 + ct - rb);
        const unsigned long long word = L.colmask[ct][lane];
        int hits = __popcll(word);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o);
        unsigned long long slo = 0ull, shi = 0ull;
        if (hits > SUMM_MAX) {
            tiles[t * WAVE + lane] = word;
            slo = SUMM_DENSE;
        } else if (hits > 0) {
            unsigned long long m = __ballot(word != 0ull);
            int slot = 1;
            while (m) {                                     // wave-uniform: <= SUMM_MAX iterations in total
                const int c = __builtin_ctzll(m);
                m &= m - 1;
                unsigned long long w = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(word >> 32), c) << 32) |
                                       (unsigned)__builtin_amdgcn_readlane((int)word, c);
                while (w) {
                    const int r = __builtin_ctzll(w);
                    w &= w - 1;
                    const unsigned long long e = (unsigned long long)((r << 6) | c);
                    if (slot < 4) slo |= e << (16 * slot); else shi |= e << (16 * (slot - 4));
                    slot++;
                }
            }
            slo |= (unsigned long long)hits;
        }
        if (lane == 0) summ[t] = make_uint4((unsigned)slo, (unsigned)(slo >> 32), (unsigned)shi, (unsigned)(shi >> 32));
    }
}

// ------------------------------------------------------------------------------------------------ K3
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_WAVES = SCAN_THREADS / WAVE;
constexpr int PANEL = 4;                                        // block rows resolved per barrier
constexpr int STAGE_TILES = PANEL * (PANEL + 1) / 2 + PANEL * PANEL;   // wave 0's tiles per step: in-panel + next panel
constexpr int APPLY_LANES = (SCAN_WAVES - 1) * WAVE;
constexpr int APPLY_PREFETCH = 4;                               // summary rounds held in registers one step ahead

__device__ __forceinline__ unsigned summ_entry(const uint4 &s, int k) {   // halfword k (1..7)
    const unsigned w = (k >> 1) == 0 ? s.x : ((k >> 1) == 1 ? s.y : ((k >> 1) == 2 ? s.z : s.w));
    return (k & 1) ? (w >> 16) : (w & 0xffffu);
}

// wave 0's tile j of step p: j < PANEL*(PANEL+1)/2 -> in-panel (row i, column c >= i, row-major), then PANEL x PANEL next-panel
__device__ __forceinline__ void stage_tile_rc(int j, int &ri, int &ci) {
    constexpr int NIN = PANEL * (PANEL + 1) / 2;
    if (j < NIN) {
        int r = 0, base = 0;
        while (j >= base + (PANEL - r)) { base += PANEL - r; r++; }
        ri = r; ci = r + (j - base);
    } else {
        ri = (j - NIN) / PANEL; ci = PANEL + (j - NIN) % PANEL;
    }
}
__device__ __forceinline__ int stage_slot(int ri, int ci) {      // inverse of stage_tile_rc
    constexpr int NIN = PANEL * (PANEL + 1) / 2;
    if (ci < PANEL) return ri * PANEL - ri * (ri - 1) / 2 + (ci - ri);
    return NIN + ri * PANEL + (ci - PANEL);
}

__global__ void __launch_bounds__(SCAN_THREADS)
rnms_scan_kernel(int n, const unsigned long long *__restrict__ tiles, const uint4 *__restrict__ summ,
                 const int32_t *__restrict__ order, unsigned char *__restrict__ flags,
                 int64_t *__restrict__ keep_out, int32_t *__restrict__ num_keep, const int32_t *__restrict__ seg_off,
                 long long seg_tile_stride, int p_begin, int p_end, unsigned long long *__restrict__ state) {
    // p_begin / p_end / state (round 6): the panel steps [p_begin, p_end) of the scan, p_end < 0 = to the last panel.  A launch that stops
    // early leaves remv[] and keepw[] in `state` ([2][W] words); a launch with p_begin > 0 picks them up.  ryolo_rnms runs the first 60 %
    // of the steps on a second stream while the mask kernel still computes the block rows below them (launch_rnms_split).
    extern __shared__ __attribute__((aligned(16))) unsigned long long smem[];
    if (seg_off) {             // segmented call: one workgroup per segment, keep flags only (sorted order = input order)
        const int s = blockIdx.x;
        const int lo = seg_off[s];
        n = seg_off[s + 1] - lo;
        tiles += (size_t)s * seg_tile_stride * WAVE;
        summ += (size_t)s * seg_tile_stride;
        flags += lo;
        if (n <= 0) return;
    }
    const int W = (n + WAVE - 1) / WAVE;
    const int NP = (W + PANEL - 1) / PANEL;
    const int PE = p_end < 0 || p_end > NP ? NP : p_end;    // steps of this launch: [p_begin, PE)
    unsigned long long *remv = smem;                      // [W]  suppression bits per block (sorted order)
    unsigned long long *keepw = smem + W;                 // [W]  keep bits per block
    unsigned long long *stage = smem + 2 * W;             // [2][STAGE_TILES][WAVE] column words of wave 0's tiles
    unsigned long long *stage_rows = stage + 2 * STAGE_TILES * WAVE;   // [2][STAGE_TILES] rows with a suppressing pair
    int *wsum = (int *)(stage_rows + 2 * STAGE_TILES);    // [SCAN_WAVES + 1]
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x >> 6;
    if (p_begin > 0) {
        for (int i = threadIdx.x; i < W; i += SCAN_THREADS) { remv[i] = state[i]; keepw[i] = state[W + i]; }
    } else {
        for (int i = threadIdx.x; i < W; i += SCAN_THREADS) remv[i] = 0ull;
    }

    // ---- helper state (waves 1..15)
    // staging: tile j = (wv - 1) + 15 * u of wave 0's set, u < 2; its summary is loaded one step before it is expanded
    constexpr int STAGE_PER_WAVE = (STAGE_TILES + SCAN_WAVES - 2) / (SCAN_WAVES - 1);
    uint4 st_s[STAGE_PER_WAVE];
    // applying: summaries of the (row, column) tiles this lane handles in the NEXT step
    uint4 ap_s[APPLY_PREFETCH];

    auto stage_tile_index = [&](int p, int j, long long &tid) -> bool {   // global tile id of wave 0's tile j at step p
        int ri, ci;
        stage_tile_rc(j, ri, ci);
        const int b = p * PANEL + ri, c = p * PANEL + ci;
        if (b >= W || c >= W) return false;
        tid = tile_base(b, W) + (c - b);
        return true;
    };
    auto load_stage_summaries = [&](int p) {          // for step p
#pragma unroll
        for (int u = 0; u < STAGE_PER_WAVE; u++) {
            const int j = (wv - 1) + (SCAN_WAVES - 1) * u;
            long long tid;
            st_s[u] = make_uint4(0u, 0u, 0u, 0u);
            if (j < STAGE_TILES && p < PE && stage_tile_index(p, j, tid)) st_s[u] = summ[tid];
        }
    };
    auto expand_stage = [&](int p) {                  // st_s (loaded for step p) -> stage[p & 1]
        unsigned long long *sg = stage + (size_t)(p & 1) * STAGE_TILES * WAVE;
        unsigned long long *sr = stage_rows + (size_t)(p & 1) * STAGE_TILES;
#pragma unroll
        for (int u = 0; u < STAGE_PER_WAVE; u++) {
            const int j = (wv - 1) + (SCAN_WAVES - 1) * u;
            if (j >= STAGE_TILES) continue;
            const uint4 s4 = st_s[u];
            const unsigned cnt = s4.x & 0xffffu;
            unsigned long long word = 0ull, rows = 0ull;
            if (cnt == SUMM_DENSE) {
                long long tid;
                stage_tile_index(p, j, tid);
                word = tiles[tid * WAVE + lane];
                // rows with any bit: OR over the 64 column words
                unsigned long long o = word;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) o |= __shfl_xor(o, d);
                rows = o;
            } else {
                for (unsigned k = 1; k <= cnt; k++) {
                    const unsigned e = summ_entry(s4, (int)k);
                    rows |= 1ull << (e >> 6);
                    if ((int)(e & 63u) == lane) word |= 1ull << (e >> 6);
                }
            }
            sg[j * WAVE + lane] = word;
            if (lane == 0) sr[j] = rows;
        }
    };
    // applier geometry of step p: rows of panel p-1, columns from the start of panel p+1
    auto apply_item = [&](int p, int i, int &b, int &c) -> bool {
        const int c0 = (p + 1) * PANEL;
        const int ncols = W - c0;
        if (p < 1 || ncols <= 0 || i >= PANEL * ncols) return false;
        const int j = i / ncols;
        b = (p - 1) * PANEL + j;
        c = c0 + (i - j * ncols);
        return true;
    };
    auto load_apply_summaries = [&](int p) {
#pragma unroll
        for (int u = 0; u < APPLY_PREFETCH; u++) {
            int b, c;
            ap_s[u] = make_uint4(0u, 0u, 0u, 0u);
            if (apply_item(p, (wv - 1) * WAVE + lane + u * APPLY_LANES, b, c)) ap_s[u] = summ[tile_base(b, W) + (c - b)];
        }
    };
    auto apply_one = [&](const uint4 &s4, bool valid, int b, int c) {
        const unsigned cnt = valid ? (s4.x & 0xffffu) : 0u;
        const unsigned long long keep = valid ? keepw[b] : 0ull;
        if (cnt != SUMM_DENSE) {
            for (unsigned k = 1; k <= cnt; k++) {
                const unsigned e = summ_entry(s4, (int)k);
                if ((keep >> (e >> 6)) & 1ull) atomicOr(&remv[c], 1ull << (e & 63u));
            }
        }
        // dense tiles: the whole wave takes them one at a time (64 column words, one ballot)
        unsigned long long dm = __ballot(cnt == SUMM_DENSE);
        while (dm) {
            const int src = __builtin_ctzll(dm);
            dm &= dm - 1;
            const int bb = __shfl(b, src), cc = __shfl(c, src);
            const unsigned long long kk = keepw[bb];
            const unsigned long long w = tiles[(tile_base(bb, W) + (cc - bb)) * WAVE + lane];
            const unsigned long long sup = __ballot((w & kk) != 0ull);
            if (lane == 0 && sup) atomicOr(&remv[cc], sup);
        }
    };

    // ---- prologue: stage step 0, prefetch step 1's summaries
    if (wv > 0) {
        load_stage_summaries(p_begin);
        expand_stage(p_begin);
        load_stage_summaries(p_begin + 1);
        load_apply_summaries(p_begin);  // step 0 has no rows to fold yet: zeros
    }
    __syncthreads();

    for (int p = p_begin; p < PE; p++) {
        if (wv == 0) {
            // ---- the serial chain: resolve the panel's diagonal tiles, fold its rows into this and the next panel's columns
            const unsigned long long *sg = stage + (size_t)(p & 1) * STAGE_TILES * WAVE;
            const unsigned long long *sr = stage_rows + (size_t)(p & 1) * STAGE_TILES;
            // the step's 26 row masks in ONE LDS read (lane j holds tile j's), then v_readlane per tile: the chain below would
            // otherwise wait for an LDS round trip per tile
            const unsigned long long myrows = lane < STAGE_TILES ? sr[lane] : 0ull;
            auto rows_of = [&](int j) __attribute__((always_inline)) {
                return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(myrows >> 32), j) << 32) |
                       (unsigned)__builtin_amdgcn_readlane((int)myrows, j);
            };
#pragma unroll
            for (int ri = 0; ri < PANEL; ri++) {
                const int b = p * PANEL + ri;
                if (b >= W) break;
                unsigned long long removed = remv[b];
                const int row_size = min(n - b * WAVE, WAVE);
                if (row_size < WAVE) removed |= ~0ull << row_size;
                {
                    const int j = stage_slot(ri, ri);
                    unsigned long long rows = rows_of(j);
                    if (rows) {
                        const unsigned long long dword = sg[j * WAVE + lane];
                        // a row's suppression only reaches HIGHER columns, so bit k of `removed` is final when row k is visited
                        while (rows) {
                            const int k = __builtin_ctzll(rows);
                            rows &= rows - 1;
                            if (!((removed >> k) & 1ull)) removed |= __ballot((dword >> k) & 1ull);
                        }
                    }
                }
                const unsigned long long keep = ~removed;
                if (lane == 0) keepw[b] = keep;
#pragma unroll
                for (int ci = ri + 1; ci < 2 * PANEL; ci++) {
                    const int c = p * PANEL + ci;
                    if (c >= W) break;
                    const int j = stage_slot(ri, ci);
                    if (rows_of(j) & keep) {
                        const unsigned long long w = sg[j * WAVE + lane];
                        const unsigned long long sup = __ballot((w & keep) != 0ull);
                        if (lane == 0 && sup) atomicOr(&remv[c], sup);
                    }
                }
            }
        } else {
            // ---- helpers: expand the tiles wave 0 needs next step, then fold panel p-1's rows into every later column
            if (p + 1 < PE) expand_stage(p + 1);
            load_stage_summaries(p + 2);
            uint4 cur[APPLY_PREFETCH];
#pragma unroll
            for (int u = 0; u < APPLY_PREFETCH; u++) cur[u] = ap_s[u];
            load_apply_summaries(p + 1);
            const int c0 = (p + 1) * PANEL;
            const int total = (p >= 1 && W > c0) ? PANEL * (W - c0) : 0;
            const int g = (wv - 1) * WAVE + lane;
#pragma unroll
            for (int u = 0; u < APPLY_PREFETCH; u++) {
                if (u * APPLY_LANES >= total) break;                 // wave-uniform
                int b = 0, c = 0;
                const bool valid = apply_item(p, g + u * APPLY_LANES, b, c);
                apply_one(cur[u], valid, b, c);
            }
            for (int i0 = APPLY_PREFETCH * APPLY_LANES; i0 < total; i0 += APPLY_LANES) {   // very long rows: not prefetched
                int b = 0, c = 0;
                const bool valid = apply_item(p, g + i0, b, c);
                uint4 s4 = make_uint4(0u, 0u, 0u, 0u);
                if (valid) s4 = summ[tile_base(b, W) + (c - b)];
                apply_one(s4, valid, b, c);
            }
        }
        __syncthreads();
    }

    if (PE < NP) {             // an early stop: hand remv / keepw to the launch that continues
        for (int i = threadIdx.x; i < W; i += SCAN_THREADS) { state[i] = remv[i]; state[W + i] = keepw[i]; }
        return;
    }
    // tail: keep flags in ORIGINAL index space, then ascending compaction
    for (int i = threadIdx.x; i < n; i += SCAN_THREADS)
        flags[order ? order[i] : i] = (unsigned char)((keepw[i >> 6] >> (i & 63)) & 1ull);
    if (seg_off) return;       // the caller compacts
    __threadfence_block();
    __syncthreads();
    const int per = (n + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(threadIdx.x * per, n), hi = min(lo + per, n);
    int cnt = 0;
    for (int i = lo; i < hi; i++) cnt += flags[i];
    // block exclusive scan of cnt: wave scan + wave sums
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == WAVE - 1) wsum[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < SCAN_WAVES; i++) { const int v = wsum[i]; wsum[i] = acc; acc += v; }
        wsum[SCAN_WAVES] = acc;
        *num_keep = acc;
    }
    __syncthreads();
    int pos = wsum[wv] + incl - cnt;
    for (int i = lo; i < hi; i++)
        if (flags[i]) keep_out[pos++] = (int64_t)i;
}

inline void scan_allow_big_lds() {        // remv + keepw of a 262144-box call (64 KiB) + the stage exceed the 64 KiB default
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void *)rnms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        done = true;
    }
}

inline size_t scan_smem_bytes(int W) {
    return sizeof(unsigned long long) * (2 * (size_t)W + 2 * (size_t)STAGE_TILES * WAVE + 2 * STAGE_TILES) +
           sizeof(int) * (SCAN_WAVES + 2);
}

// ------------------------------------------------------------------------------------------------ IoU kernels
__device__ __forceinline__ void load_quad(const float *r, Quad &q, float &area) {
    convert_region(r[0], r[1], r[2], r[3], r[4], q);
    area = r[2] * r[3];
}

__global__ void riou_pairs_kernel(const float *__restrict__ b1, int s1, const float *__restrict__ b2, int s2, int n,
                                  float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Quad q1, q2;
    float a1, a2;
    load_quad(b1 + (size_t)i * s1, q1, a1);
    load_quad(b2 + (size_t)i * s2, q2, a2);
    out[i] = riou_generic(q1, a1, q2, a2);
}

__global__ void riou_matrix_kernel(const float *__restrict__ b1, int n1, int s1, const float *__restrict__ b2, int n2,
                                   int s2, float *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n2 || i >= n1) return;
    Quad q1, q2;
    float a1, a2;
    load_quad(b1 + (size_t)i * s1, q1, a1);
    load_quad(b2 + (size_t)j * s2, q2, a2);
    out[(size_t)i * n2 + j] = riou_generic(q1, a1, q2, a2);
}

// ------------------------------------------------------------------------------------------------ host side
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RnmsLayout {
    size_t keys_in, keys_out, idx_in, order, p0, p1, aux, summ, tiles, flags, ext, state, cub, total;
    size_t cub_bytes;
    long long ntiles;
};

RnmsLayout rnms_layout(int n) {
    RnmsLayout L{};
    const long long W = ((long long)n + WAVE - 1) / WAVE;
    L.ntiles = W * (W + 1) / 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    L.keys_in = take(sizeof(uint32_t) * (size_t)n);
    L.keys_out = take(sizeof(uint32_t) * (size_t)n);
    L.idx_in = take(sizeof(int32_t) * (size_t)n);
    L.order = take(sizeof(int32_t) * (size_t)n);
    L.p0 = take(sizeof(float4) * (size_t)n);
    L.p1 = take(sizeof(float4) * (size_t)n);
    L.aux = take(sizeof(float4) * (size_t)n);
    L.summ = take(sizeof(uint4) * (size_t)L.ntiles);
    L.tiles = take(sizeof(unsigned long long) * WAVE * (size_t)L.ntiles);
    L.flags = take((size_t)n);
    L.ext = take(8 * sizeof(uint32_t));
    L.state = take(sizeof(unsigned long long) * 2 * (size_t)W);        // remv / keepw between the two launches of a split scan
    size_t cub_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                       (const int32_t *)nullptr, (int32_t *)nullptr, n, 0, 32, (hipStream_t)0);
    L.cub_bytes = cub_bytes;
    L.cub = take(cub_bytes);
    L.total = off;
    return L;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH; }

unsigned long long *g_pair_counter = nullptr;   // measurement hook, see ryolo_rnms_count_pairs
// Column tiles per wave of the mask kernel: two from 128 block rows (n > 8128), one below.  Whole call on the SURVEY 8(d) distribution
// (profiles/r06_nms_multi_tile.txt): 16 384 boxes 0.455 -> 0.448 ms, 50 000 2.74 -> 2.59, 100 000 9.9 -> 9.2 with two; at 500 - 4 096 boxes
// (the per-image sets of the detection path) one tile per wave is faster (0.100 vs 0.121 ms: few waves, long tails); four lose to two
// everywhere.  RYOLO_RNMS_TILES = 1 | 2 (ryolo_set_tuning) forces one: the tests run every edge case through both.
inline int mask_column_tiles(int W) {
    const char *e = ryolo_detail::tune(ryolo_detail::TUNE_RNMS_TILES);
    if (e && (e[0] == '1' || e[0] == '2') && e[1] == 0) return e[0] - '0';
    return W >= 128 ? 2 : 1;
}

// W = block rows of the (largest) set; one wave per (block row, group of CT column tiles); grid.z = segment
// block rows [row_lo, row_hi) of the upper triangle (row_hi < 0: to the last row); the longest row of the range has W - row_lo tiles
void launch_mask(int n, float thr, const float4 *P0, const float4 *P1, const float4 *AUX, unsigned long long *tiles, uint4 *summ, int W,
                 int num_segments, const int32_t *seg_off, long long seg_tile_stride, hipStream_t stream, int row_lo = 0, int row_hi = -1) {
    const int ct = mask_column_tiles(W);
    if (row_hi < 0 || row_hi > W) row_hi = W;
    if (row_hi <= row_lo) return;
    const int kj = (W - row_lo + ct - 1) / ct;
    const dim3 grid((unsigned)((kj + MASK_WAVES - 1) / MASK_WAVES), (unsigned)(row_hi - row_lo), (unsigned)num_segments);
    if (ct == 2)
        hipLaunchKernelGGL(rnms_mask_multi_kernel<2>, grid, dim3(MASK_WAVES * WAVE), 0, stream, n, thr, P0, P1, AUX, tiles, summ, seg_off,
                           seg_tile_stride, g_pair_counter, thr < 0.f ? 0 : 1, row_lo);
    else
        hipLaunchKernelGGL(rnms_mask_multi_kernel<1>, grid, dim3(MASK_WAVES * WAVE), 0, stream, n, thr, P0, P1, AUX, tiles, summ, seg_off,
                           seg_tile_stride, g_pair_counter, thr < 0.f ? 0 : 1, row_lo);
}

// ---- the scan of a large call under the mask kernel (round 6).  The scan is one workgroup walking 4-row panels (0.53 ms of the 2.63-ms
// call on 50 000 boxes); panel p only needs the block ROWS up to its own.  So the mask kernel runs as two launches -- rows [0, k) and
// [k, W) -- and the first 60 % of the panel steps run on a second stream beside the second launch (events order the three pieces; no
// flags, no spinning); the rest follows on the caller's stream.  60 %: 0.53 x = 2.0 (1 - x)^2, the point where the hidden part of the scan is
// as long as the mask work left to hide it under.  Same kernels, same tiles, same step order: the keep list does not change.
struct SidePool {
    hipStream_t s = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    int state = 0;                 // 0 not tried, 1 ready, -1 unavailable
};
static std::mutex g_side_mutex;
static SidePool g_side[16];
// the second stream + two events of the current device, created on first use (not while `stream` is being captured into a graph:
// stream / event creation is illegal there -- such a call runs unsplit until a call outside a capture has created them)
static SidePool *side_pool(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SidePool &sp = g_side[dev];
    if (sp.state == 0) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
        const bool ok = hipStreamCreateWithFlags(&sp.s, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&sp.a, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&sp.b, hipEventDisableTiming) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        sp.state = ok ? 1 : -1;
    }
    return sp.state == 1 ? &sp : nullptr;
}
constexpr int SPLIT_MIN_ROWS = 320;      // block rows (n > 20 416) from which the split pays (16 384 boxes: 0.495 vs 0.477 ms; 24 000: 0.750 vs 0.791)

}  // namespace

extern "C" {

const char *ryolo_strerror(int code) {
    switch (code) {
        case RYOLO_OK: return "ok";
        case RYOLO_EINVAL: return "invalid argument";
        case RYOLO_ELAUNCH: return "HIP launch failed (no gfx950 device, or a previous asynchronous error)";
        case RYOLO_ETOOBIG: return "too many boxes for one rotated-NMS call";
        default: return "unknown ryolo error";
    }
}

int ryolo_abi_version(void) { return 2; }

void ryolo_rnms_count_pairs(uint64_t *device_counter) { g_pair_counter = (unsigned long long *)device_counter; }


size_t ryolo_rnms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    if (n > RYOLO_RNMS_MAX_BOXES) return 0;
    return rnms_layout(n).total;
}

int ryolo_rnms(const float *dets, int n, int row_stride, float thr, int64_t *keep_out, int32_t *num_keep,
               void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || !num_keep) return RYOLO_EINVAL;
    if (n == 0) {
        return hipMemsetAsync(num_keep, 0, sizeof(int32_t), stream) == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
    }
    if (!dets || !keep_out || !workspace || row_stride < 6) return RYOLO_EINVAL;
    if (n > RYOLO_RNMS_MAX_BOXES) return RYOLO_ETOOBIG;
    const RnmsLayout L = rnms_layout(n);
    if (workspace_bytes < L.total) return RYOLO_EINVAL;
    char *ws = (char *)workspace;
    uint32_t *keys_in = (uint32_t *)(ws + L.keys_in), *keys_out = (uint32_t *)(ws + L.keys_out);
    int32_t *idx_in = (int32_t *)(ws + L.idx_in), *order = (int32_t *)(ws + L.order);
    float4 *P0 = (float4 *)(ws + L.p0), *P1 = (float4 *)(ws + L.p1), *AUX = (float4 *)(ws + L.aux);
    uint4 *summ = (uint4 *)(ws + L.summ);
    unsigned char *flags = (unsigned char *)(ws + L.flags);
    unsigned long long *tiles = (unsigned long long *)(ws + L.tiles);

    const int tb = 256, nb = (n + tb - 1) / tb;
    hipLaunchKernelGGL(rnms_keys_kernel, dim3(nb), dim3(tb), 0, stream, dets, n, row_stride, keys_in, idx_in);
    size_t cub_bytes = L.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub_bytes, keys_in, keys_out, idx_in, order, n, 0, 32,
                                           stream) != hipSuccess)
        return RYOLO_ELAUNCH;
    uint32_t *ext = (uint32_t *)(ws + L.ext);
    if (hipMemsetAsync(ext, 0, 8 * sizeof(uint32_t), stream) != hipSuccess) return RYOLO_ELAUNCH;
    hipLaunchKernelGGL(rnms_extent_kernel, dim3(nb < 32 ? nb : 32), dim3(tb), 0, stream, dets, n, row_stride, ext);
    hipLaunchKernelGGL(rnms_corners_kernel, dim3(nb), dim3(tb), 0, stream, dets, n, row_stride, order, ext, P0, P1, AUX);
    const int W = (n + WAVE - 1) / WAVE;
    const size_t smem = scan_smem_bytes(W);
    scan_allow_big_lds();
    unsigned long long *state = (unsigned long long *)(ws + L.state);
    bool split = W >= SPLIT_MIN_ROWS;
    {
        const char *e = ryolo_detail::abl_env("RYOLO_RNMS_SPLIT");      // measurement build: 0 = one mask launch, one scan launch (A/B timing)
        if (e && e[0] == '0') split = false;
    }
    if (split) {
        std::lock_guard<std::mutex> lock(g_side_mutex);                // the pool's events are shared by the host threads of a process
        SidePool *sp = side_pool(stream);
        if (sp) {
            const int NP = (W + PANEL - 1) / PANEL, pe = NP * 3 / 5, k = pe * PANEL;
            // the caller's stream: rows [0, k), then the first panel steps; the second stream: rows [k, W) as soon as the first launch is done.
            // (The other way round -- scan beside a mask launch already resident -- the scan's one 1024-thread workgroup found no CU with 16
            //  free wave slots and 100 KiB of LDS until the mask launch had drained: no overlap at all.  Here the scan is the next packet of
            //  the stream that just finished, the mask launch waits for an event on another queue: the scan is placed first.)
            launch_mask(n, thr, P0, P1, AUX, tiles, summ, W, 1, nullptr, 0ll, stream, 0, k);
            bool ok = hipEventRecord(sp->a, stream) == hipSuccess;
            hipLaunchKernelGGL(rnms_scan_kernel, dim3(1), dim3(SCAN_THREADS), smem, stream, n, tiles, summ, order, flags,
                               keep_out, num_keep, (const int32_t *)nullptr, 0ll, 0, pe, state);
            ok = ok && hipStreamWaitEvent(sp->s, sp->a, 0) == hipSuccess;
            launch_mask(n, thr, P0, P1, AUX, tiles, summ, W, 1, nullptr, 0ll, sp->s, k, W);
            ok = ok && hipEventRecord(sp->b, sp->s) == hipSuccess && hipStreamWaitEvent(stream, sp->b, 0) == hipSuccess;
            hipLaunchKernelGGL(rnms_scan_kernel, dim3(1), dim3(SCAN_THREADS), smem, stream, n, tiles, summ, order, flags,
                               keep_out, num_keep, (const int32_t *)nullptr, 0ll, pe, -1, state);
            return ok ? check_launch() : RYOLO_ELAUNCH;
        }
    }
    launch_mask(n, thr, P0, P1, AUX, tiles, summ, W, 1, nullptr, 0ll, stream);
    hipLaunchKernelGGL(rnms_scan_kernel, dim3(1), dim3(SCAN_THREADS), smem, stream, n, tiles, summ, order, flags,
                       keep_out, num_keep, (const int32_t *)nullptr, 0ll, 0, -1, state);
    return check_launch();
}

// ---- segmented NMS: S independent sets laid out back to back, each ALREADY sorted by score (descending, the order
// in which the greedy scan visits them).  One launch of each kernel for all sets: grid.y / grid.x = segment.
static inline long long seg_tiles(int max_seg_len) {
    const long long W = ((long long)max_seg_len + WAVE - 1) / WAVE;
    return W * (W + 1) / 2;
}

size_t ryolo_rnms_segmented_workspace_bytes(int m, int num_segments, int max_seg_len) {
    if (m <= 0 || num_segments <= 0 || max_seg_len <= 0 || max_seg_len > RYOLO_RNMS_MAX_BOXES) return 0;
    const size_t nt = (size_t)seg_tiles(max_seg_len) * (size_t)num_segments;
    return 3 * align256(sizeof(float4) * (size_t)m) + align256(sizeof(uint4) * nt) + align256(sizeof(unsigned long long) * WAVE * nt) +
           align256(8 * sizeof(uint32_t));
}

int ryolo_rnms_segmented(const float *dets, int m, int row_stride, const int32_t *seg_offsets, int num_segments,
                         int max_seg_len, float thr, unsigned char *keep_flags, void *workspace, size_t workspace_bytes,
                         void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (m < 0 || num_segments < 0) return RYOLO_EINVAL;
    if (m == 0 || num_segments == 0) return RYOLO_OK;
    if (!dets || !seg_offsets || !keep_flags || !workspace || row_stride < 6 || max_seg_len <= 0) return RYOLO_EINVAL;
    if (max_seg_len > RYOLO_RNMS_MAX_BOXES) return RYOLO_ETOOBIG;
    if (workspace_bytes < ryolo_rnms_segmented_workspace_bytes(m, num_segments, max_seg_len)) return RYOLO_EINVAL;
    const long long nt1 = seg_tiles(max_seg_len);
    char *ws = (char *)workspace;
    float4 *P0 = (float4 *)ws; ws += align256(sizeof(float4) * (size_t)m);
    float4 *P1 = (float4 *)ws; ws += align256(sizeof(float4) * (size_t)m);
    float4 *AUX = (float4 *)ws; ws += align256(sizeof(float4) * (size_t)m);
    uint4 *summ = (uint4 *)ws; ws += align256(sizeof(uint4) * (size_t)nt1 * num_segments);
    unsigned long long *tiles = (unsigned long long *)ws;
    const int tb = 256, nb = (m + tb - 1) / tb;
    uint32_t *ext = (uint32_t *)(tiles + (size_t)nt1 * num_segments * WAVE);
    if (hipMemsetAsync(ext, 0, 8 * sizeof(uint32_t), stream) != hipSuccess) return RYOLO_ELAUNCH;
    hipLaunchKernelGGL(rnms_extent_kernel, dim3(nb < 32 ? nb : 32), dim3(tb), 0, stream, dets, m, row_stride, ext);
    hipLaunchKernelGGL(rnms_corners_kernel, dim3(nb), dim3(tb), 0, stream, dets, m, row_stride, (const int32_t *)nullptr, ext,
                       P0, P1, AUX);
    const int W = (max_seg_len + WAVE - 1) / WAVE;
    launch_mask(0, thr, P0, P1, AUX, tiles, summ, W, num_segments, seg_offsets, nt1, stream);
    const size_t smem = scan_smem_bytes(W);
    scan_allow_big_lds();
    hipLaunchKernelGGL(rnms_scan_kernel, dim3((unsigned)num_segments), dim3(SCAN_THREADS), smem, stream, 0, tiles, summ,
                       (const int32_t *)nullptr, keep_flags, (int64_t *)nullptr, (int32_t *)nullptr, seg_offsets, nt1, 0, -1,
                       (unsigned long long *)nullptr);
    return check_launch();
}

int ryolo_riou_pairs(const float *b1, int stride1, const float *b2, int stride2, int n, float *out, void *stream_) {
    if (n < 0) return RYOLO_EINVAL;
    if (n == 0) return RYOLO_OK;
    if (!b1 || !b2 || !out || stride1 < 5 || stride2 < 5) return RYOLO_EINVAL;
    hipLaunchKernelGGL(riou_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, b1, stride1, b2,
                       stride2, n, out);
    return check_launch();
}

int ryolo_riou_matrix(const float *b1, int n1, int stride1, const float *b2, int n2, int stride2, float *out,
                      void *stream_) {
    if (n1 < 0 || n2 < 0) return RYOLO_EINVAL;
    if (n1 == 0 || n2 == 0) return RYOLO_OK;
    if (!b1 || !b2 || !out || stride1 < 5 || stride2 < 5 || n1 > 65535) return RYOLO_EINVAL;
    hipLaunchKernelGGL(riou_matrix_kernel, dim3((n2 + 255) / 256, n1), dim3(256), 0, (hipStream_t)stream_, b1, n1,
                       stride1, b2, n2, stride2, out);
    return check_launch();
}

}  // extern "C"
