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
//   K2  mask      ONE WAVEFRONT PER 64x64 TILE of the upper triangle.  Phase 1: every lane owns one column box
//                 and runs the 6-flop bounding-circle reject against the 64 row boxes (row box broadcast with
//                 v_readlane); survivors are compacted with ballot+mbcnt into a 128-entry LDS ring.  Phase 2:
//                 whenever 64 candidates are queued, all 64 lanes run the exact polygon IoU on one candidate
//                 each (operands fetched from the owning lanes with ds_bpermute), so the divergent
//                 per-pair code runs on dense wavefronts instead of ~4 %-occupied ones.  Result bits are
//                 OR-ed into 64 TRANSPOSED words (word c = which rows suppress column c) and stored as one
//                 coalesced 512-B tile, only if non-zero, plus one occupancy byte per tile.
//   K3  scan      one 1024-thread workgroup walks the 64-box blocks in order: resolves the diagonal tile with
//                 a wave-uniform 64-step ballot loop, then each wave turns a whole off-diagonal tile into its
//                 64-bit suppression word with ONE ballot ((colword & keep) != 0), skipping empty tiles.
//                 Tail: scatter keep flags to original indices and compact them in ascending order.
//
// Bit-exactness contract (checked in tests/test_rnms_gpu.py against oracle/riou_oracle.c, which is pinned to the
// reference arithmetic): every fp32 operation of devRotateIoU is reproduced in the reference's order with
// IEEE + - * / sqrt, no FMA contraction (this TU is built with -ffp-contract=off and correctly rounded
// divide/sqrt), corners from the shared "correctly rounded sincos" definition, 24-slot point buffers.
// The bounding-circle reject only skips pairs for which the reference arithmetic yields exactly 0 intersection
// points (circles separated by a margin 50x the worst corner rounding), i.e. IoU == 0 <= thr.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "../../include/ryolo.h"

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
__global__ void rnms_corners_kernel(const float *__restrict__ dets, int n, int row_stride,
                                    const int32_t *__restrict__ order, float4 *P0, float4 *P1, float4 *AUX) {
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
    const float rad = hd * 1.00001f + 1.0e-5f * (fabsf(cx) + fabsf(cy) + hd);
    AUX[i] = make_float4(cx, cy, rad, w * h);
}

// ------------------------------------------------------------------------------------------------ K2
struct BoxRegs {
    float4 p0, p1, aux;
};
__device__ __forceinline__ void fetch_box(const BoxRegs &mine, int src_lane, Quad &q, float &area) {
    q.x[0] = __shfl(mine.p0.x, src_lane); q.y[0] = __shfl(mine.p0.y, src_lane);
    q.x[1] = __shfl(mine.p0.z, src_lane); q.y[1] = __shfl(mine.p0.w, src_lane);
    q.x[2] = __shfl(mine.p1.x, src_lane); q.y[2] = __shfl(mine.p1.y, src_lane);
    q.x[3] = __shfl(mine.p1.z, src_lane); q.y[3] = __shfl(mine.p1.w, src_lane);
    area = __shfl(mine.aux.w, src_lane);
}

constexpr int MASK_WAVES = 4;                       // waves per workgroup, each owns one tile
constexpr int QCAP = 128;                           // candidate ring (entries: row << 6 | col)
struct __attribute__((aligned(16))) MaskWaveLds {
    float bx[FAST_PTS * WAVE];
    float by[FAST_PTS * WAVE];
    float bk[FAST_PTS * WAVE];
    unsigned long long colmask[WAVE];
    unsigned short queue[QCAP];
};

// tiles of the upper triangle in row-major order: tile id t <-> (rb, cb >= rb)
__device__ __forceinline__ long long tile_base(int rb, int W) { return (long long)rb * W - (long long)rb * (rb - 1) / 2; }

__global__ void __launch_bounds__(MASK_WAVES *WAVE)
rnms_mask_kernel(int n, float thr, const float4 *__restrict__ P0, const float4 *__restrict__ P1,
                 const float4 *__restrict__ AUX, unsigned long long *__restrict__ tiles,
                 unsigned char *__restrict__ occ, long long ntiles, const int32_t *__restrict__ seg_off,
                 long long seg_tile_stride, unsigned long long *__restrict__ eval_counter) {
    __shared__ MaskWaveLds lds_all[MASK_WAVES];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x >> 6;
    const long long t = (long long)blockIdx.x * MASK_WAVES + wv;
    if (seg_off) {             // segmented call: blockIdx.y = segment, boxes [seg_off[s], seg_off[s+1]) are one NMS set
        const int s = blockIdx.y;
        const int lo = seg_off[s];
        n = seg_off[s + 1] - lo;
        const long long Ws = ((long long)n + WAVE - 1) / WAVE;
        ntiles = Ws * (Ws + 1) / 2;
        P0 += lo; P1 += lo; AUX += lo;
        tiles += (size_t)s * seg_tile_stride * WAVE;
        occ += (size_t)s * seg_tile_stride;
    }
    if (t >= ntiles) return;   // whole wave exits together (t is wave-uniform)
    MaskWaveLds &L = lds_all[wv];

    const int W = (n + WAVE - 1) / WAVE;
    // invert tile_base: rb = floor(((2W+1) - sqrt((2W+1)^2 - 8t)) / 2), then fix up
    int rb = (int)(((2.0 * W + 1.0) - sqrt((2.0 * W + 1.0) * (2.0 * W + 1.0) - 8.0 * (double)t)) * 0.5);
    if (rb < 0) rb = 0;
    if (rb > W - 1) rb = W - 1;
    while (rb > 0 && tile_base(rb, W) > t) rb--;
    while (rb + 1 < W && tile_base(rb + 1, W) <= t) rb++;
    const int cb = rb + (int)(t - tile_base(rb, W));

    const int row0 = rb * WAVE, col0 = cb * WAVE;
    const int row_size = min(n - row0, WAVE);
    const int col_size = min(n - col0, WAVE);

    BoxRegs rowb, colb;   // lane r holds row box r, lane c holds column box c
    {
        const int ri = min(row0 + lane, n - 1), ci = min(col0 + lane, n - 1);
        rowb.p0 = P0[ri]; rowb.p1 = P1[ri]; rowb.aux = AUX[ri];
        colb.p0 = P0[ci]; colb.p1 = P1[ci]; colb.aux = AUX[ci];
    }
    L.colmask[lane] = 0ull;

    float *bx = L.bx + lane, *by = L.by + lane, *bk = L.bk + lane;
    int head = 0, tail = 0;   // wave-uniform ring indices
    const bool diag = (rb == cb);
    const bool col_ok = lane < col_size;

    auto run_candidates = [&](int count) {
        // lanes [0,count) each take one queued (row, col) pair and run the exact IoU
        const bool active = lane < count;
        const unsigned e = L.queue[(head + (active ? lane : 0)) & (QCAP - 1)];
        const int r = (int)(e >> 6), c = (int)(e & 63u);
        Quad q1, q2;
        float a1, a2;
        fetch_box(rowb, r, q1, a1);   // box_i (higher score) is the FIRST argument, kernel.cu:301
        fetch_box(colb, c, q2, a2);
        if (active) {
            float iou;
            if (!riou_fast(q1, a1, q2, a2, bx, by, bk, iou)) iou = riou_generic(q1, a1, q2, a2);
            if (iou > thr) atomicOr(&L.colmask[c], 1ull << r);
        }
        if (eval_counter && lane == 0) atomicAdd(eval_counter, (unsigned long long)count);   // measurement only (bench.py)
        head += count;
    };

    for (int r = 0; r < row_size; r++) {
        const float rcx = __shfl(rowb.aux.x, r), rcy = __shfl(rowb.aux.y, r), rrad = __shfl(rowb.aux.z, r);
        const float dx = rcx - colb.aux.x, dy = rcy - colb.aux.y;
        const float d2 = dx * dx + dy * dy;
        const float lim = rrad + colb.aux.z;
        const bool reject = d2 > lim * lim;            // NaN anywhere -> not rejected
        const bool cand = col_ok && !reject && (!diag || lane > r);
        const unsigned long long m = __ballot(cand);
        if (m) {
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (cand) L.queue[(tail + pos) & (QCAP - 1)] = (unsigned short)((r << 6) | lane);
            tail += __popcll(m);
            if (tail - head >= WAVE) run_candidates(WAVE);
        }
    }
    if (tail - head > 0) run_candidates(tail - head);

    const unsigned long long word = L.colmask[lane];
    const bool any = __ballot(word != 0ull) != 0ull;
    if (any) tiles[t * WAVE + lane] = word;
    if (lane == 0) occ[t] = any ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ K3
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_WAVES = SCAN_THREADS / WAVE;

__global__ void __launch_bounds__(SCAN_THREADS)
rnms_scan_kernel(int n, const unsigned long long *__restrict__ tiles, const unsigned char *__restrict__ occ,
                 const int32_t *__restrict__ order, unsigned char *__restrict__ flags,
                 int64_t *__restrict__ keep_out, int32_t *__restrict__ num_keep, const int32_t *__restrict__ seg_off,
                 long long seg_tile_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long smem[];
    if (seg_off) {             // segmented call: one workgroup per segment, keep flags only (sorted order = input order)
        const int s = blockIdx.x;
        const int lo = seg_off[s];
        n = seg_off[s + 1] - lo;
        tiles += (size_t)s * seg_tile_stride * WAVE;
        occ += (size_t)s * seg_tile_stride;
        flags += lo;
        if (n <= 0) return;
    }
    const int W = (n + WAVE - 1) / WAVE;
    unsigned long long *remv = smem;        // [W]   suppression bits per block (sorted order)
    unsigned long long *keepw = smem + W;   // [W]   keep bits per block
    int *wsum = (int *)(smem + 2 * W);      // [SCAN_WAVES + 1]
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < W; i += SCAN_THREADS) remv[i] = 0ull;

    for (int b = 0; b < W; b++) {
        const long long base = tile_base(b, W);
        // issue this block's loads before the barrier: they do not depend on the serial state
        const unsigned char docc = occ[base];
        unsigned long long dword = 0ull;
        if (docc) dword = tiles[base * WAVE + lane];
        // lane l of wave wv looks at tile column t = b + 1 + wv + SCAN_WAVES * l (first round prefetched)
        unsigned long long tmask0;
        {
            const int tcol = b + 1 + wv + SCAN_WAVES * lane;
            tmask0 = __ballot((tcol < W) && occ[base + (tcol - b)] != 0);
        }
        __syncthreads();   // remv[b] is final: every earlier block has been folded in
        unsigned long long removed = remv[b];
        const int row_size = min(n - b * WAVE, WAVE);
        if (row_size < WAVE) removed |= ~0ull << row_size;
        unsigned long long keep = 0ull;
        if (docc) {
            for (int k = 0; k < WAVE; k++) {   // wave-uniform serial resolve of the diagonal tile
                if (!((removed >> k) & 1ull)) {
                    keep |= 1ull << k;
                    removed |= __ballot((dword >> k) & 1ull);
                }
            }
        } else {
            keep = ~removed;
        }
        if (threadIdx.x == 0) keepw[b] = keep;
        // off-diagonal tiles of block row b owned by this wave
        for (int l0 = 0; b + 1 + wv + SCAN_WAVES * l0 < W; l0 += WAVE) {   // <= 4 rounds (W <= 4096)
            unsigned long long m;
            if (l0 == 0) m = tmask0;
            else {
                const int tcol = b + 1 + wv + SCAN_WAVES * (l0 + lane);
                m = __ballot((tcol < W) && occ[base + (tcol - b)] != 0);
            }
            while (m) {
                // up to 4 tile loads in flight
                int l[4];
                unsigned long long w[4];
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (m) {
                        l[u] = __builtin_ctzll(m);
                        m &= m - 1;
                        const int tcol = b + 1 + wv + SCAN_WAVES * (l0 + l[u]);
                        w[u] = tiles[(base + (tcol - b)) * WAVE + lane];
                        cnt = u + 1;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (u < cnt) {
                        const int tcol = b + 1 + wv + SCAN_WAVES * (l0 + l[u]);
                        const unsigned long long sup = __ballot((w[u] & keep) != 0ull);
                        if (lane == 0) remv[tcol] |= sup;   // tcol is owned by exactly one wave per block
                    }
                }
            }
        }
    }
    __syncthreads();

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
    size_t keys_in, keys_out, idx_in, order, p0, p1, aux, occ, tiles, flags, cub, total;
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
    L.occ = take((size_t)L.ntiles);
    L.tiles = take(sizeof(unsigned long long) * WAVE * (size_t)L.ntiles);
    L.flags = take((size_t)n);
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
    unsigned char *occ = (unsigned char *)(ws + L.occ), *flags = (unsigned char *)(ws + L.flags);
    unsigned long long *tiles = (unsigned long long *)(ws + L.tiles);

    const int tb = 256, nb = (n + tb - 1) / tb;
    hipLaunchKernelGGL(rnms_keys_kernel, dim3(nb), dim3(tb), 0, stream, dets, n, row_stride, keys_in, idx_in);
    size_t cub_bytes = L.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub_bytes, keys_in, keys_out, idx_in, order, n, 0, 32,
                                           stream) != hipSuccess)
        return RYOLO_ELAUNCH;
    hipLaunchKernelGGL(rnms_corners_kernel, dim3(nb), dim3(tb), 0, stream, dets, n, row_stride, order, P0, P1, AUX);
    const long long nblk = (L.ntiles + MASK_WAVES - 1) / MASK_WAVES;
    hipLaunchKernelGGL(rnms_mask_kernel, dim3((unsigned)nblk), dim3(MASK_WAVES * WAVE), 0, stream, n, thr, P0, P1,
                       AUX, tiles, occ, L.ntiles, (const int32_t *)nullptr, 0ll, g_pair_counter);
    const int W = (n + WAVE - 1) / WAVE;
    const size_t smem = sizeof(unsigned long long) * 2 * (size_t)W + sizeof(int) * (SCAN_WAVES + 2);
    hipLaunchKernelGGL(rnms_scan_kernel, dim3(1), dim3(SCAN_THREADS), smem, stream, n, tiles, occ, order, flags,
                       keep_out, num_keep, (const int32_t *)nullptr, 0ll);
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
    return 3 * align256(sizeof(float4) * (size_t)m) + align256(nt) + align256(sizeof(unsigned long long) * WAVE * nt);
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
    unsigned char *occ = (unsigned char *)ws; ws += align256((size_t)nt1 * num_segments);
    unsigned long long *tiles = (unsigned long long *)ws;
    const int tb = 256, nb = (m + tb - 1) / tb;
    hipLaunchKernelGGL(rnms_corners_kernel, dim3(nb), dim3(tb), 0, stream, dets, m, row_stride, (const int32_t *)nullptr,
                       P0, P1, AUX);
    const long long nblk = (nt1 + MASK_WAVES - 1) / MASK_WAVES;
    hipLaunchKernelGGL(rnms_mask_kernel, dim3((unsigned)nblk, (unsigned)num_segments), dim3(MASK_WAVES * WAVE), 0, stream, 0,
                       thr, P0, P1, AUX, tiles, occ, 0ll, seg_offsets, nt1, g_pair_counter);
    const int W = (max_seg_len + WAVE - 1) / WAVE;
    const size_t smem = sizeof(unsigned long long) * 2 * (size_t)W + sizeof(int) * (SCAN_WAVES + 2);
    hipLaunchKernelGGL(rnms_scan_kernel, dim3((unsigned)num_segments), dim3(SCAN_THREADS), smem, stream, 0, tiles, occ,
                       (const int32_t *)nullptr, keep_flags, (int64_t *)nullptr, (int32_t *)nullptr, seg_offsets, nt1);
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
