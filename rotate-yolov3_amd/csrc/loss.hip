// rotate-yolov3_amd/csrc/loss.hip -- the YOLO training loss of one head and its gradient, on device (gfx950).
//
// Replaces, for the 'default' arcs, what the reference computes with ~100 small ATen ops + autograd per head in
// compute_loss (model/loss.py:266-367): objectness BCE over ALL cells (:346-348), and for the positive (anchor, target)
// candidates smooth-L1 on sigmoid(xy) and atan(angle)+anchor (:314-323), the wh-IoU term (:322, utils/utils.py:346-361)
// and the class BCE (:329-333).  Candidate selection (build_targets, :161-258) stays in the fixed-shape tensor
// formulation of model/loss_static.py; this file consumes its [na, NT] weight grid.
//
//   dense kernel     : d loss / d p for every element (zero except the objectness column, sigmoid(x) * obj / cells) and the
//                      sum of softplus(x) = BCE against an all-zero target; one coalesced read + write of the head. HBM-bound.
//   positives kernel : one thread per candidate with weight 1: gathers the cell's `no` logits, adds the regression /
//                      class terms and gradients (atomics: two candidates may share a cell), and -- once per distinct
//                      cell, decided by an atomicOr on a bitmap -- the objectness correction from target 0 to target 1.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ryolo.h"
#include "riou_grad.h"

namespace {

__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float sl1(float d) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float sl1_grad(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f); }

// Focal wrapper of the reference (model/loss.py:126-146: loss *= alpha * (1.000001 - exp(-loss)) ** gamma, alpha = 1) around an
// element's base loss L: value fl = L * q^gamma with q = 1.000001 - e^-L, and f = d fl / d L = q^gamma + L * gamma * q^(gamma-1) * e^-L.
struct Focal { float fl, f; };
__device__ __forceinline__ Focal focal_of(float L, float gamma, bool on) {
    if (!on) return Focal{L, 1.f};
    const float e = expf(-L), q = 1.000001f - e, m = powf(q, gamma);
    return Focal{L * m, m + L * gamma * (m / q) * e};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int NO>     // NO == 7 (nc = 1): the modulo arithmetic below folds to multiplies; 0: generic
__global__ void __launch_bounds__(256)
yolo_loss_dense_kernel(const float *__restrict__ p, long long n4, long long total, int no_rt, float coef,
                       float *__restrict__ dp, float *__restrict__ items) {
    const int no = NO ? NO : no_rt;
    float acc = 0.f;
    if (blockIdx.x == 0 && threadIdx.x < (int)(total - n4 * 4)) {      // the 0..3 elements past the last float4
        const long long f = n4 * 4 + threadIdx.x;
        const bool obj = f % no == 5;
        dp[f] = obj ? coef * sigmoidf(p[f]) : 0.f;
        if (obj) acc += softplusf(p[f]);
    }
    const unsigned uno = (unsigned)no;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = ((const float4 *)p)[i];
        // residue of the flat element index modulo `no` in 32-bit arithmetic; no >= 6 > 4, so at most ONE of the four
        // elements is an objectness logit: one sigmoid / softplus per float4 (evaluating them per element under
        // divergence made this pass ALU-bound at 1.8 TB/s)
        const unsigned rem = i < 0x7fffffffll ? (unsigned)i % uno : (unsigned)(i % no);
        const unsigned r0 = (4u * rem) % uno;
        const unsigned k = (5u + uno - r0) % uno;                // position of the objectness logit, if < 4
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < 4u) {
            const float x = k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
            const float e = __expf(-fabsf(x));
            const float sg = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            acc += fmaxf(x, 0.f) + __logf(1.f + e);
            const float gval = coef * sg;
            if (k == 0) o.x = gval; else if (k == 1) o.y = gval; else if (k == 2) o.z = gval; else o.w = gval;
        }
        ((float4 *)dp)[i] = o;
    }
    // one atomic per WORKGROUP: 32 k same-address atomics (one per wave of an 8192-block grid) serialise in L2 for longer
    // than the streaming pass itself takes
    __shared__ float wsum[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (t != 0.f) atomicAdd(items + 0, t * coef);
    }
}

struct PosParams {
    const float *p;
    float *dp;
    const float *w;                 // [na, NT]
    const long long *b, *gj, *gi, *cls;   // [NT]
    const float *txy, *twh, *ta;    // [NT,2], [NT,2], [NT]
    const float *av;                // [na,3] anchor (w, h, angle) in grid units
    const float *npos;              // device scalar: number of positives of this head
    unsigned *bitmap;               // one bit per cell (bs*na*ny*nx), zeroed by the caller
    float *items;                   // [4]: lobj, lcls, lreg (already weighted); [3] untouched
    int bs, na, ny, nx, no, NT, nc;
    float giou, reg_w, cls_w, cls_pw, obj_coef, obj_pw;
    float cls_coef;                 // uCE: cls_w / cells
    int iou_mode;                   // 0: axis-aligned wh_iou (the reference's term), 1: rotated IoU of the decoded box
    int focal;                      // RYOLO_ARC_FOCAL: every criterion but the IoU term wrapped (loss.py:284-286)
    int unified;                    // 0 'default' (objectness + class terms), 1 uBCE (loss.py:350-354), 2 uCE (:356-360)
    float gamma;                    // hyp['fl_gamma']
    unsigned *clsmap;               // uBCE with nc > 1: one bit per (cell, class), zeroed by the caller (behind `bitmap`)
    int pixmajor;                   // bitmap bit of a cell: 0 = the cell index (anchor-major, like p), 1 = pixel * na + anchor (NHWC dense pass)
};

__global__ void __launch_bounds__(256) yolo_loss_pos_kernel(const PosParams q) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    float l_obj = 0.f, l_cls = 0.f, l_reg = 0.f;
    if (idx < q.na * q.NT && q.w[idx] > 0.f) {
        const int a = idx / q.NT, t = idx % q.NT;
        const float n = fmaxf(q.npos[0], 1.f);
        const long long cell = (((long long)q.b[t] * q.na + a) * q.ny + q.gj[t]) * q.nx + q.gi[t];
        const float *ps = q.p + cell * q.no;
        float *dps = q.dp + cell * q.no;
        const float aw = q.av[a * 3 + 0], ah = q.av[a * 3 + 1], aa = q.av[a * 3 + 2];
        // xy: smooth-L1 on sigmoid, mean over n*2
        {
            const float rw = q.reg_w / (2.f * n);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const float s = sigmoidf(ps[k]);
                const float d = s - q.txy[t * 2 + k];
                const Focal fo = focal_of(sl1(d), q.gamma, q.focal != 0);
                l_reg += rw * fo.fl;
                atomicAdd(dps + k, rw * fo.f * sl1_grad(d) * s * (1.f - s));
            }
        }
        // angle: 2 * smooth-L1(atan(raw) + anchor - target), mean over n
        {
            const float raw = ps[4];
            const float d = atanf(raw) + aa - q.ta[t];
            const float rw = 2.f * q.reg_w / n;
            const Focal fo = focal_of(sl1(d), q.gamma, q.focal != 0);
            l_reg += rw * fo.fl;
            atomicAdd(dps + 4, rw * fo.f * sl1_grad(d) / (1.f + raw * raw));
        }
        // wh: giou * (1 - wh_iou(target, pred)), mean over n; pred = min(exp(raw), 1e3) * anchor
        if (q.iou_mode == 0) {
            const float ew = __expf(ps[2]), eh = __expf(ps[3]);
            const float pw = fminf(ew, 1e3f) * aw, ph = fminf(eh, 1e3f) * ah;
            const float tw = q.twh[t * 2], th = q.twh[t * 2 + 1];
            const float mw = fminf(tw, pw), mh = fminf(th, ph);
            const float inter = mw * mh;
            const float uni = (tw * th + 1e-16f) + pw * ph - inter;
            const float iou = inter / uni;
            const float rw = q.reg_w * q.giou / n;
            l_reg += rw * (1.f - iou);
            // d min(t, p) / d p: 1 below the target, 1/2 at a tie (ATen's minimum), 0 above
            const float gw = pw < tw ? 1.f : (pw == tw ? 0.5f : 0.f), gh = ph < th ? 1.f : (ph == th ? 0.5f : 0.f);
            const float dI_w = gw * mh, dI_h = gh * mw;
            const float diou_w = (dI_w * uni - inter * (ph - dI_w)) / (uni * uni);
            const float diou_h = (dI_h * uni - inter * (pw - dI_h)) / (uni * uni);
            const float dpw = ew <= 1e3f ? ew * aw : 0.f, dph = eh <= 1e3f ? eh * ah : 0.f;
            atomicAdd(dps + 2, -rw * diou_w * dpw);
            atomicAdd(dps + 3, -rw * diou_h * dph);
        } else {
            // riou (the build's extension, hyp['riou'] = 1): giou * (1 - rotated IoU(decoded box, target)), mean over n; the
            // decoded box is the pbox of loss.py:314-318 (cell offset, anchor-scaled size, anchor angle + atan), all five
            // parameters receive the polygon-overlap gradient (riou_grad.h)
            const float sx = sigmoidf(ps[0]), sy = sigmoidf(ps[1]);
            const float ew = __expf(ps[2]), eh = __expf(ps[3]);
            const float raw = ps[4];
            const float P[5] = {sx, sy, fminf(ew, 1e3f) * aw, fminf(eh, 1e3f) * ah, atanf(raw) + aa};
            const float T[5] = {q.txy[t * 2], q.txy[t * 2 + 1], q.twh[t * 2], q.twh[t * 2 + 1], q.ta[t]};
            float g[5];
            const float iou = ryolo_riou::riou_fwd_bwd(P, T, g);
            const float rw = q.reg_w * q.giou / n;
            l_reg += rw * (1.f - iou);
            atomicAdd(dps + 0, -rw * g[0] * sx * (1.f - sx));
            atomicAdd(dps + 1, -rw * g[1] * sy * (1.f - sy));
            atomicAdd(dps + 2, ew <= 1e3f ? -rw * g[2] * ew * aw : 0.f);
            atomicAdd(dps + 3, eh <= 1e3f ? -rw * g[3] * eh * ah : 0.f);
            atomicAdd(dps + 4, -rw * g[4] / (1.f + raw * raw));
        }
        // the NHWC dense pass looks the anchors of ONE pixel up together: its bitmap is pixel-major (a wave's 72 lookups then fall in
        // one cache line instead of 72 lines a plane apart)
        const long long bcell = q.pixmajor ? (((long long)q.b[t] * q.ny + q.gj[t]) * q.nx + q.gi[t]) * q.na + a : cell;
        const unsigned bit = 1u << (bcell & 31);
        if (q.unified == 0) {
            // classes (nc > 1): BCE with pos_weight against the one-hot class, mean over n*nc
            if (q.nc > 1) {
                const float cw = q.cls_w / (n * (float)q.nc);
                const int tc = (int)q.cls[t];
                for (int k = 0; k < q.nc; k++) {
                    const float x = ps[6 + k];
                    const float y = k == tc ? 1.f : 0.f;
                    const Focal fo = focal_of(q.cls_pw * y * softplusf(-x) + (1.f - y) * softplusf(x), q.gamma, q.focal != 0);
                    l_cls += cw * fo.fl;
                    atomicAdd(dps + 6 + k, cw * fo.f * (sigmoidf(x) * (1.f - y + q.cls_pw * y) - q.cls_pw * y));
                }
            }
            // objectness: the first candidate to claim the cell moves its target from 0 to 1
            const unsigned old = atomicOr(q.bitmap + (bcell >> 5), bit);
            if (!(old & bit)) {
                const float x = ps[5];
                const float sg = sigmoidf(x);
                const Focal f1 = focal_of(q.obj_pw * softplusf(-x), q.gamma, q.focal != 0), f0 = focal_of(softplusf(x), q.gamma, q.focal != 0);
                l_obj = q.obj_coef * (f1.fl - f0.fl);
                atomicAdd(dps + 5, q.obj_coef * (f1.f * q.obj_pw * (sg - 1.f) - f0.f * sg));
            }
        } else if (q.unified == 1) {
            // uBCE (loss.py:350-354): BCE over the class logits of ALL cells against t (1 at the positives' class), mean over
            // cells * nc, added to lobj.  The dense pass charges every logit against 0; the first candidate to claim a
            // (cell, class) moves that target to 1.  `bitmap` only marks the cell as touched (the dense pass merges dp there).
            atomicOr(q.bitmap + (bcell >> 5), bit);
            const int tc = q.nc > 1 ? (int)q.cls[t] : 0;
            const long long cc = cell * (q.nc > 1 ? q.nc : 1) + tc;
            const unsigned cb = 1u << (cc & 31);
            if (!(atomicOr(q.clsmap + (cc >> 5), cb) & cb)) {
                const float x = ps[6 + tc];
                const float sg = sigmoidf(x);
                const float cu = q.obj_coef / (float)q.nc;         // obj_coef = obj_w / cells
                const Focal f1 = focal_of(softplusf(-x), q.gamma, q.focal != 0), f0 = focal_of(softplusf(x), q.gamma, q.focal != 0);
                l_obj = cu * (f1.fl - f0.fl);
                atomicAdd(dps + 6 + tc, cu * (f1.f * (sg - 1.f) - f0.f * sg));
            }
        } else {
            // uCE (loss.py:356-360): cross entropy over (background, classes) = logits 5 .. 5+nc of ALL cells, target 0 except
            // tcls + 1 at the positives, mean over cells, added to lcls.  The dense pass charges every cell against the
            // background; the first candidate to claim a cell moves its target (the reference's index_put_ is undefined for
            // two targets of different class in one cell; the first claimant wins here).
            const unsigned old = atomicOr(q.bitmap + (bcell >> 5), bit);
            if (!(old & bit)) {
                const int nl = q.nc + 1, c = (int)q.cls[t] + 1;
                float mx = ps[5];
                for (int k = 1; k < nl; k++) mx = fmaxf(mx, ps[5 + k]);
                float se = 0.f;
                for (int k = 0; k < nl; k++) se += expf(ps[5 + k] - mx);
                const float lse = mx + logf(se);
                const Focal f1 = focal_of(lse - ps[5 + c], q.gamma, q.focal != 0), f0 = focal_of(lse - ps[5], q.gamma, q.focal != 0);
                l_cls = q.cls_coef * (f1.fl - f0.fl);
                for (int k = 0; k < nl; k++) {
                    const float sm = expf(ps[5 + k] - lse);
                    atomicAdd(dps + 5 + k, q.cls_coef * (f1.f * (sm - (k == c ? 1.f : 0.f)) - f0.f * (sm - (k == 0 ? 1.f : 0.f))));
                }
            }
        }
    }
    l_obj = wave_sum(l_obj);
    l_cls = wave_sum(l_cls);
    l_reg = wave_sum(l_reg);
    if ((threadIdx.x & 63) == 0) {
        if (l_obj != 0.f) atomicAdd(q.items + 0, l_obj);
        if (l_cls != 0.f) atomicAdd(q.items + 1, l_cls);
        if (l_reg != 0.f) atomicAdd(q.items + 2, l_reg);
    }
}

// ---- dense pass of the training step, NHWC in / NHWC out.  The three passes the engine used to run around the positives
// kernel (objectness loss + fp32 d loss/d p over [bs, na, ny, nx, no]; the autograd scale; fp32 -> NHWC bf16 for the head
// conv's backward) read and wrote 3.5 GB per bs-64 step.  This kernel reads the head the conv wrote (bf16 NHWC, channel =
// a*no + k -- the same values as p) and writes the head gradient in the layout the backward consumes: objectness column
// coef * sigmoid(x), every other entry 0, plus -- for the few cells the positives kernel touched (bitmap) -- its fp32
// contributions, which are read from the sparse buffer `dp` and ZEROED again, so that buffer stays all-zero between steps.
typedef __attribute__((ext_vector_type(8))) __bf16 loss_bf16x8;
template <int NO>
__global__ void __launch_bounds__(256)
yolo_loss_dense_nhwc_kernel(const __bf16 *__restrict__ head, int head_cs, long long npix, int plane, int na, int no_rt, float coef,
                            float *__restrict__ dp, const unsigned *__restrict__ bitmap, __bf16 *__restrict__ hg, int hg_cs,
                            float *__restrict__ items) {
    const int no = NO ? NO : no_rt;
    const int cpr = na * no / 8;
    float acc = 0.f;
    // A wave owns one pixel per trip (lane = 16-B chunk of its channels, 64 chunks per sweep).  The pass was ALU-bound twice over:
    // the first version decomposed a flat index with two 64-bit divisions per chunk (~250 instructions per 16 bytes moved); the
    // second tested `column == 5` per element, and with 63 lanes in 7 different phases every element's sigmoid / softplus / IEEE
    // division ran for the whole wave -- ~750 instructions per chunk, 1.9 TB/s.  A chunk of 8 channels holds at most two
    // objectness columns (one when no >= 8) at positions known from its first column: they are selected, evaluated once each, and
    // scattered back; the positives' contributions (rare) take a wave-divergent slow path.
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    auto one = [&](long long pix, int c8, const loss_bf16x8 &v) {
        const int a0 = c8 / no, k0 = c8 - a0 * no;       // anchor / column of the chunk's first channel
        int e1 = 5 - k0;
        if (e1 < 0) e1 += no;                            // k0 + e1 == 5 (mod no): first objectness column, >= 8: none in this chunk
        const int e2 = e1 + no;                          // a second one only when no < 8
        // the chunk as four dwords: only the (at most two) objectness logits are converted, and the output row is assembled from the
        // two bf16 results -- every other channel's gradient is an exact zero
        const uint4 raw = __builtin_bit_cast(uint4, v);
        auto logit = [&](int e) {
            const int w = e >> 1;
            unsigned d = raw.x;
            d = w == 1 ? raw.y : d;
            d = w == 2 ? raw.z : d;
            d = w == 3 ? raw.w : d;
            return __builtin_bit_cast(float, (e & 1) ? (d & 0xffff0000u) : (d << 16));
        };
        auto objectness = [&](float x) {
            const float ex = __expf(-fabsf(x));
            const float r = __builtin_amdgcn_rcpf(1.f + ex);
            acc += fmaxf(x, 0.f) + __logf(1.f + ex);
            return coef * (x >= 0.f ? r : ex * r);
        };
        float g1 = 0.f, g2 = 0.f;
        if (e1 < 8) g1 = objectness(logit(e1));
        if (e2 < 8) g2 = objectness(logit(e2));
        auto place = [&](float g, int e, unsigned (&ow)[4]) {
            const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)g) << ((e & 1) * 16);
#pragma unroll
            for (int w = 0; w < 4; w++) ow[w] |= (e >> 1) == w ? bits : 0u;
        };
        unsigned ow[4] = {0u, 0u, 0u, 0u};
        if (e1 < 8) place(g1, e1, ow);
        if (e2 < 8) place(g2, e2, ow);
        // cells the positives kernel touched (pixel-major bitmap: the pixel's anchors are consecutive bits)
        const long long bit0 = pix * na + a0;
        const int nspan = (k0 + 7) / no;                 // further anchors this chunk reaches into
        unsigned hits = 0;
        for (int j = 0; j <= nspan; j++) hits |= ((bitmap[(bit0 + j) >> 5] >> ((bit0 + j) & 31)) & 1u) << j;
        if (hits) {
            const unsigned n = (unsigned)pix / (unsigned)plane;      // (npix < 2^31, host check)
            const long long cell0 = ((long long)n * na) * plane + ((unsigned)pix - n * (unsigned)plane);     // cell of anchor 0 at this pixel
            int j = 0, k = k0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float o = (e == e1 ? g1 : 0.f) + (e == e2 ? g2 : 0.f);
                if ((hits >> j) & 1u) {
                    float *src = dp + (cell0 + (long long)(a0 + j) * plane) * no + k;
                    o += *src;
                    *src = 0.f;
                }
                const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)o);
                ow[e >> 1] = (e & 1) ? (ow[e >> 1] & 0xffffu) | (bits << 16) : (ow[e >> 1] & 0xffff0000u) | bits;
                if (++k == no) {
                    k = 0;
                    j++;
                }
            }
        }
        *(uint4 *)(hg + pix * hg_cs + c8) = uint4{ow[0], ow[1], ow[2], ow[3]};
    };
    // (two or four pixels in flight per wave measured slower -- 122.8 / 129.2 us against 120.9 per launch: the occupancy they cost is
    // worth more than the extra loads in flight)
    for (long long pix = wave0; pix < npix; pix += nwaves)
        for (int chunk = lane; chunk < cpr; chunk += 64) one(pix, chunk * 8, *(const loss_bf16x8 *)(head + pix * head_cs + chunk * 8));
    __shared__ float wsum[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (t != 0.f) atomicAdd(items + 0, t * coef);
    }
}

// ---- dense pass of the focal / unified arcs (loss.py:284-286, :350-360), both layouts: one thread per cell (anchor fastest, so a
// wave reads and writes consecutive bytes of a pixel's channels).  Not tuned like the 'default' passes above (2-byte accesses): the
// reference's training arcs for this repository are the default ones; these keep every arc on the HIP path.
//   unified 0 (Fdefault): objectness logit against 0, focal-wrapped;   1 (uBCE): the nc class logits against 0, mean over cells*nc;
//   2 (uCE): cross entropy of logits 5..5+nc against the background class, mean over cells (added to lcls).
// NHWC mode (head != nullptr): reads the bf16 head, merges + re-zeroes the positives' sparse contributions where the bitmap says
// so, writes the bf16 head gradient.  fp32 mode: reads p, writes dp (the positives kernel adds onto it afterwards).
__global__ void __launch_bounds__(256)
yolo_loss_dense_arc_kernel(const __bf16 *__restrict__ head, int head_cs, const float *__restrict__ p, long long cells, long long npix,
                           int plane, int na, int no, int nc, int focal, int unified, float gamma, float coef, float *__restrict__ dp,
                           const unsigned *__restrict__ bitmap, __bf16 *__restrict__ hg, int hg_cs, float *__restrict__ items) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x) {
        float x[32], g[32];                                   // no <= 32 here (host check)
        long long cell;
        long long pix = 0;
        int a = 0;
        if (head) {
            pix = i / na;
            a = (int)(i - pix * na);
            const long long n = pix / plane;
            cell = (n * na + a) * plane + (pix - n * plane);
            for (int k = 0; k < no; k++) x[k] = (float)head[pix * head_cs + a * no + k];
        } else {
            cell = i;
            for (int k = 0; k < no; k++) x[k] = p[cell * no + k];
        }
        for (int k = 0; k < no; k++) g[k] = 0.f;
        if (unified == 0) {
            const Focal fo = focal_of(softplusf(x[5]), gamma, focal != 0);
            acc += fo.fl;
            g[5] = coef * fo.f * sigmoidf(x[5]);
        } else if (unified == 1) {
            const int ncl = nc > 1 ? nc : 1;
            for (int k = 0; k < ncl; k++) {
                const Focal fo = focal_of(softplusf(x[6 + k]), gamma, focal != 0);
                acc += fo.fl;
                g[6 + k] = coef * fo.f * sigmoidf(x[6 + k]);
            }
        } else {
            const int nl = nc + 1;
            float mx = x[5];
            for (int k = 1; k < nl; k++) mx = fmaxf(mx, x[5 + k]);
            float se = 0.f;
            for (int k = 0; k < nl; k++) se += expf(x[5 + k] - mx);
            const float lse = mx + logf(se);
            const Focal fo = focal_of(lse - x[5], gamma, focal != 0);
            acc += fo.fl;
            for (int k = 0; k < nl; k++) g[5 + k] = coef * fo.f * (expf(x[5 + k] - lse) - (k == 0 ? 1.f : 0.f));
        }
        if (head) {
            if ((bitmap[i >> 5] >> (i & 31)) & 1u) {         // pixel-major bit index = pix * na + a = i
                for (int k = 0; k < no; k++) {
                    g[k] += dp[cell * no + k];
                    dp[cell * no + k] = 0.f;
                }
            }
            for (int k = 0; k < no; k++) hg[pix * hg_cs + a * no + k] = (__bf16)g[k];
        } else {
            for (int k = 0; k < no; k++) dp[cell * no + k] = g[k];
        }
    }
    __shared__ float wsum[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (t != 0.f) atomicAdd(items + (unified == 2 ? 1 : 0), t * coef);
    }
}

// buf *= g[0] unless g[0] == 1 (the upstream gradient of loss.backward()): the whole grid returns after one scalar load
__global__ void scale_bf16_if_kernel(const float *__restrict__ g, __bf16 *__restrict__ buf, int cs, long long npix, int C) {
    const float sc = g[0];
    if (sc == 1.f) return;
    const int cpr = C / 8;
    const long long total = npix * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / cpr;
        const int c8 = (int)(i - pix * cpr) * 8;
        loss_bf16x8 v = *(loss_bf16x8 *)(buf + pix * cs + c8);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (__bf16)((float)v[e] * sc);
        *(loss_bf16x8 *)(buf + pix * cs + c8) = v;
    }
}

// ---- build_targets (model/loss.py:161-258) in the fixed-shape form of model/loss_static.py: one thread per target walks
// the head-major [heads x anchors] candidate table once -- wh-IoU against every anchor, the angle gate, and for a target
// no anchor accepted the best-anchor fallback (first maximum picks the head, the tie with the smallest angle offset
// picks the anchor).  Same fp32 operations in the same order as the tensor formulation (exact-equality tested).
struct BuildTargetsParams {
    const float *tpad;             // [NT, 7] (img, cls, x, y, w, h, a), padded
    const unsigned char *valid;    // [NT]
    int NT, nheads, na;
    float iou_t, ang_t, cf;
    const float *ng[RYOLO_MAX_HEADS];        // [2] grid size (nx, ny) per head
    const float *av[RYOLO_MAX_HEADS];        // [na, 3] anchor (w, h, angle) in grid units per head
    float *w[RYOLO_MAX_HEADS];               // out [na, NT] 0/1
    long long *idx[RYOLO_MAX_HEADS];         // out [4, NT]: b, cls, gj, gi
    float *box[RYOLO_MAX_HEADS];             // out: txy [NT,2] | twh [NT,2] | ta [NT]  (the layout ryolo_yolo_loss reads)
    float *npos[RYOLO_MAX_HEADS];            // out [1] (zeroed by the caller): number of positives
};

__global__ void __launch_bounds__(256) build_targets_kernel(const BuildTargetsParams q) {
    // every loop over heads is fully unrolled with a guard: the per-head pointers are read from the kernel arguments with
    // STATIC indices (a run-time index into the by-value struct sends the whole struct through scratch), anchors come
    // through uniform (scalar) loads, and the accept counts stay in registers instead of being re-read from `w`
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= q.NT) return;
    const bool ok = q.valid[t] != 0;
    const float *r = q.tpad + (size_t)t * 7;
    const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    float tw = r[4], th = r[5];
    const float ta = r[6];
    float gw[RYOLO_MAX_HEADS], gh[RYOLO_MAX_HEADS];
#pragma unroll
    for (int h = 0; h < RYOLO_MAX_HEADS; h++) {
        if (h >= q.nheads) break;
        // Q7: the context rescale is applied once per head, cumulatively
        tw = tw + th * (q.cf - 1.f);
        th = th * q.cf;
        const float nx = q.ng[h][0], ny = q.ng[h][1];
        gw[h] = tw * nx;
        gh[h] = th * ny;
        const float gx = r2 * nx, gy = r3 * ny;
        q.idx[h][0 * q.NT + t] = (long long)r0;
        q.idx[h][1 * q.NT + t] = (long long)r1;
        q.idx[h][2 * q.NT + t] = (long long)gy;
        q.idx[h][3 * q.NT + t] = (long long)gx;
        q.box[h][t * 2 + 0] = gx - floorf(gx);                 // txy [NT, 2]
        q.box[h][t * 2 + 1] = gy - floorf(gy);
        q.box[h][2 * q.NT + t * 2 + 0] = gw[h];                // twh [NT, 2]
        q.box[h][2 * q.NT + t * 2 + 1] = gh[h];
        q.box[h][4 * q.NT + t] = ta;                           // ta [NT]
    }
    const float half_pi = 0.5f * 3.14159265358979323846f, pi = 3.14159265358979323846f;
    // accept flags, running maximum with its first position, tie with the smallest angle offset
    bool covered = false;
    float best = -1.f, best_ao = 0.f;
    int first_idx = 0, pick_idx = 0;
    float cnt[RYOLO_MAX_HEADS];
    const float *av_last = q.av[0];                            // Q2: the angle gate uses the LAST head's anchor angles
#pragma unroll
    for (int h = 1; h < RYOLO_MAX_HEADS; h++)
        if (h < q.nheads) av_last = q.av[h];
#pragma unroll
    for (int h = 0; h < RYOLO_MAX_HEADS; h++) {
        cnt[h] = 0.f;
        if (h >= q.nheads) continue;
        const float *av = q.av[h];
        float *wh = q.w[h];
        for (int a = 0; a < q.na; a++) {
            const float aw = av[a * 3], ah = av[a * 3 + 1];
            const float inter = fminf(aw, gw[h]) * fminf(ah, gh[h]);
            const float iou = inter / ((aw * ah + 1e-16f) + gw[h] * gh[h] - inter);
            float ao = fabsf(ta - av_last[a * 3 + 2]);
            if (ao > half_pi) ao = pi - ao;
            const bool acc = ok && iou > q.iou_t && ao < q.ang_t;
            wh[(size_t)a * q.NT + t] = acc ? 1.f : 0.f;
            cnt[h] += acc ? 1.f : 0.f;
            covered = covered || acc;
            if (iou > best) { best = iou; first_idx = h * q.na + a; pick_idx = first_idx; best_ao = ao; }
            else if (iou == best && ao < best_ao) { pick_idx = h * q.na + a; best_ao = ao; }
        }
    }
    const bool fallback = ok && !covered;                      // head of the FIRST maximum, anchor of the tie-broken one
    const int lid = first_idx / q.na, fa = pick_idx % q.na;
#pragma unroll
    for (int h = 0; h < RYOLO_MAX_HEADS; h++) {
        if (h >= q.nheads) continue;
        if (fallback && h == lid) {
            q.w[h][(size_t)fa * q.NT + t] = 1.f;
            cnt[h] += 1.f;
        }
        if (cnt[h] != 0.f) atomicAdd(q.npos[h], cnt[h]);       // positives per head (integers in fp32: order-independent)
    }
}

// rotated IoU + gradient for n independent pairs, one pair per lane (the eager loss / tests; the fused loss calls the same
// device function from yolo_loss_pos_kernel)
__global__ void __launch_bounds__(256) riou_pairs_grad_kernel(const float *__restrict__ pbox, const float *__restrict__ tbox,
                                                              int n, float *__restrict__ iou, float *__restrict__ grad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float P[5], T[5], g[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { P[k] = pbox[(size_t)i * 5 + k]; T[k] = tbox[(size_t)i * 5 + k]; }
    iou[i] = ryolo_riou::riou_fwd_bwd(P, T, g);
    if (grad) {
#pragma unroll
        for (int k = 0; k < 5; k++) grad[(size_t)i * 5 + k] = g[k];
    }
}

}  // namespace

extern "C" {

int ryolo_riou_loss_pairs(const float *pbox, const float *tbox, int n, float *iou, float *grad, void *stream) {
    if (n < 0 || (n > 0 && (!pbox || !tbox || !iou))) return RYOLO_EINVAL;
    if (n == 0) return RYOLO_OK;
    hipLaunchKernelGGL(riou_pairs_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pbox, tbox, n, iou, grad);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}


int ryolo_build_targets(const float *tpad, const unsigned char *valid, int NT, int nheads, int na, const float *const *ng,
                        const float *const *anchor_vec, float iou_t, float ang_t, float context_factor, float *const *w,
                        long long *const *idx, float *const *box, float *const *npos, void *stream) {
    if (!tpad || !valid || NT <= 0 || nheads <= 0 || nheads > RYOLO_MAX_HEADS || na <= 0 || !ng || !anchor_vec || !w || !idx ||
        !box || !npos)
        return RYOLO_EINVAL;
    BuildTargetsParams q;
    q.tpad = tpad; q.valid = valid; q.NT = NT; q.nheads = nheads; q.na = na;
    q.iou_t = iou_t; q.ang_t = ang_t; q.cf = context_factor;
    for (int h = 0; h < nheads; h++) {
        if (!ng[h] || !anchor_vec[h] || !w[h] || !idx[h] || !box[h] || !npos[h]) return RYOLO_EINVAL;
        q.ng[h] = ng[h]; q.av[h] = anchor_vec[h]; q.w[h] = w[h]; q.idx[h] = idx[h]; q.box[h] = box[h]; q.npos[h] = npos[h];
    }
    hipLaunchKernelGGL(build_targets_kernel, dim3((NT + 255) / 256), dim3(256), 0, (hipStream_t)stream, q);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

size_t ryolo_yolo_loss_bitmap_bytes(long long cells) { return cells <= 0 ? 0 : (size_t)((cells + 31) / 32) * 4; }

static int launch_positives(const float *p, float *dp, int bs, int na, int ny, int nx, int no, int nc, const float *w, int NT,
                            const long long *b, const long long *gj, const long long *gi, const long long *cls,
                            const float *txy, const float *twh, const float *ta, const float *anchor_vec, const float *npos,
                            float giou, float reg_w, float cls_w, float cls_pw, float coef, float obj_pw, int iou_mode,
                            unsigned *bitmap, float *items, hipStream_t stream, int arc = 0, float gamma = 0.f, int pixmajor = 0) {
    PosParams q;
    q.pixmajor = pixmajor;
    q.focal = (arc & RYOLO_ARC_FOCAL) ? 1 : 0;
    q.unified = (arc & RYOLO_ARC_UBCE) ? 1 : ((arc & RYOLO_ARC_UCE) ? 2 : 0);
    q.gamma = gamma;
    {
        const long long cells = (long long)bs * na * ny * nx;
        q.clsmap = bitmap + (cells + 31) / 32;               // uBCE: (cell, class) claims live behind the cell bitmap
        q.cls_coef = cls_w / (float)cells;
    }
    q.p = p; q.dp = dp; q.w = w; q.b = b; q.gj = gj; q.gi = gi; q.cls = cls; q.txy = txy; q.twh = twh; q.ta = ta;
    q.av = anchor_vec; q.npos = npos; q.bitmap = bitmap; q.items = items;
    q.bs = bs; q.na = na; q.ny = ny; q.nx = nx; q.no = no; q.NT = NT; q.nc = nc;
    q.giou = giou; q.reg_w = reg_w; q.cls_w = cls_w; q.cls_pw = cls_pw; q.obj_coef = coef; q.obj_pw = obj_pw; q.iou_mode = iou_mode;
    const int cand = na * NT;
    hipLaunchKernelGGL(yolo_loss_pos_kernel, dim3((cand + 255) / 256), dim3(256), 0, stream, q);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

static int arc_ok(int arc, int no, int nc) {
    if (arc & ~(RYOLO_ARC_FOCAL | RYOLO_ARC_UBCE | RYOLO_ARC_UCE)) return 0;
    if ((arc & RYOLO_ARC_UBCE) && (arc & RYOLO_ARC_UCE)) return 0;
    if ((arc & (RYOLO_ARC_UBCE | RYOLO_ARC_UCE)) && no < 6 + nc) return 0;
    return arc == 0 || no <= 32;                              // the per-cell pass of the non-default arcs keeps a cell in registers
}

size_t ryolo_yolo_loss_bitmap_bytes_arc(long long cells, int nc, int arc) {
    if (cells <= 0) return 0;
    size_t bytes = (size_t)((cells + 31) / 32) * 4;
    if (arc & RYOLO_ARC_UBCE) bytes += (size_t)((cells * (nc > 1 ? nc : 1) + 31) / 32) * 4;
    return bytes;
}

int ryolo_yolo_loss_nhwc(const void *head, int head_cstride, const float *p, int bs, int na, int ny, int nx, int no, int nc,
                         const float *w, int NT, const long long *b, const long long *gj, const long long *gi,
                         const long long *cls, const float *txy, const float *twh, const float *ta, const float *anchor_vec,
                         const float *npos, float giou, float reg_w, float cls_w, float cls_pw, float obj_w, float obj_pw,
                         int iou_mode, unsigned *bitmap, float *dp_sparse, void *head_grad, int head_grad_cstride, float *items,
                         void *stream_) {
    return ryolo_yolo_loss_nhwc_arc(head, head_cstride, p, bs, na, ny, nx, no, nc, w, NT, b, gj, gi, cls, txy, twh, ta, anchor_vec, npos,
                                    giou, reg_w, cls_w, cls_pw, obj_w, obj_pw, iou_mode, 0, 0.f, bitmap, dp_sparse, head_grad,
                                    head_grad_cstride, items, stream_);
}

int ryolo_yolo_loss_nhwc_arc(const void *head, int head_cstride, const float *p, int bs, int na, int ny, int nx, int no, int nc,
                             const float *w, int NT, const long long *b, const long long *gj, const long long *gi,
                             const long long *cls, const float *txy, const float *twh, const float *ta, const float *anchor_vec,
                             const float *npos, float giou, float reg_w, float cls_w, float cls_pw, float obj_w, float obj_pw,
                             int iou_mode, int arc, float fl_gamma, unsigned *bitmap, float *dp_sparse, void *head_grad,
                             int head_grad_cstride, float *items, void *stream_) {
    if (!arc_ok(arc, no, nc)) return RYOLO_EINVAL;
    if (!head || !p || !w || !b || !gj || !gi || !cls || !txy || !twh || !ta || !anchor_vec || !npos || !bitmap || !dp_sparse ||
        !head_grad || !items)
        return RYOLO_EINVAL;
    if (bs <= 0 || na <= 0 || ny <= 0 || nx <= 0 || no < 6 + (nc > 1 ? nc : 0) || NT <= 0 || (iou_mode != 0 && iou_mode != 1))
        return RYOLO_EINVAL;
    const int C = na * no;
    if ((C & 7) || (head_cstride & 7) || (head_grad_cstride & 7) || head_cstride < C || head_grad_cstride < C ||
        ((uintptr_t)head & 15) || ((uintptr_t)head_grad & 15))
        return RYOLO_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const long long cells = (long long)bs * na * ny * nx, npix = (long long)bs * ny * nx;
    const float coef = obj_w / (float)cells;
    const int rc = launch_positives(p, dp_sparse, bs, na, ny, nx, no, nc, w, NT, b, gj, gi, cls, txy, twh, ta, anchor_vec, npos,
                                    giou, reg_w, cls_w, cls_pw, coef, obj_pw, iou_mode, bitmap, items, stream, arc, fl_gamma, 1);
    if (rc != RYOLO_OK) return rc;
    long long nb = (npix * (C / 8) + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    if (arc != 0) {
        const int unified = (arc & RYOLO_ARC_UBCE) ? 1 : ((arc & RYOLO_ARC_UCE) ? 2 : 0);
        const float cf = unified == 1 ? coef / (float)(nc > 1 ? nc : 1) : (unified == 2 ? cls_w / (float)cells : coef);
        long long nbc = (cells + 255) / 256;
        if (nbc > 4096) nbc = 4096;
        hipLaunchKernelGGL(yolo_loss_dense_arc_kernel, dim3((unsigned)nbc), dim3(256), 0, stream, (const __bf16 *)head, head_cstride,
                           (const float *)nullptr, cells, npix, ny * nx, na, no, nc, (arc & RYOLO_ARC_FOCAL) ? 1 : 0, unified, fl_gamma,
                           cf, dp_sparse, bitmap, (__bf16 *)head_grad, head_grad_cstride, items);
        return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
    }
    if (npix >= 0x7fffffffll) return RYOLO_EINVAL;
    nb = (npix + 3) / 4;
    if (nb > 8192) nb = 8192;
    if (no == 7)
        hipLaunchKernelGGL(yolo_loss_dense_nhwc_kernel<7>, dim3((unsigned)nb), dim3(256), 0, stream, (const __bf16 *)head,
                           head_cstride, npix, ny * nx, na, no, coef, dp_sparse, bitmap, (__bf16 *)head_grad, head_grad_cstride,
                           items);
    else
        hipLaunchKernelGGL(yolo_loss_dense_nhwc_kernel<0>, dim3((unsigned)nb), dim3(256), 0, stream, (const __bf16 *)head,
                           head_cstride, npix, ny * nx, na, no, coef, dp_sparse, bitmap, (__bf16 *)head_grad, head_grad_cstride,
                           items);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

int ryolo_scale_bf16_if(const float *g, void *buf, int cstride, long long npix, int C, void *stream) {
    if (!g || !buf || npix <= 0 || C <= 0 || (C & 7) || (cstride & 7) || cstride < C) return RYOLO_EINVAL;
    long long nb = (npix * (C / 8) + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(scale_bf16_if_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g, (__bf16 *)buf, cstride, npix, C);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

int ryolo_yolo_loss(const float *p, int bs, int na, int ny, int nx, int no, int nc, const float *w, int NT,
                    const long long *b, const long long *gj, const long long *gi, const long long *cls, const float *txy,
                    const float *twh, const float *ta, const float *anchor_vec, const float *npos, float giou, float reg_w,
                    float cls_w, float cls_pw, float obj_w, float obj_pw, int iou_mode, unsigned *bitmap, float *dp,
                    float *items, void *stream_) {
    return ryolo_yolo_loss_arc(p, bs, na, ny, nx, no, nc, w, NT, b, gj, gi, cls, txy, twh, ta, anchor_vec, npos, giou, reg_w, cls_w, cls_pw,
                               obj_w, obj_pw, iou_mode, 0, 0.f, bitmap, dp, items, stream_);
}

int ryolo_yolo_loss_arc(const float *p, int bs, int na, int ny, int nx, int no, int nc, const float *w, int NT,
                        const long long *b, const long long *gj, const long long *gi, const long long *cls, const float *txy,
                        const float *twh, const float *ta, const float *anchor_vec, const float *npos, float giou, float reg_w,
                        float cls_w, float cls_pw, float obj_w, float obj_pw, int iou_mode, int arc, float fl_gamma, unsigned *bitmap,
                        float *dp, float *items, void *stream_) {
    if (!arc_ok(arc, no, nc)) return RYOLO_EINVAL;
    if (!p || !w || !b || !gj || !gi || !cls || !txy || !twh || !ta || !anchor_vec || !npos || !bitmap || !dp || !items)
        return RYOLO_EINVAL;
    if (bs <= 0 || na <= 0 || ny <= 0 || nx <= 0 || no < 6 + (nc > 1 ? nc : 0) || NT <= 0 || (iou_mode != 0 && iou_mode != 1))
        return RYOLO_EINVAL;
    const long long cells = (long long)bs * na * ny * nx, total = cells * no;
    if (((uintptr_t)p | (uintptr_t)dp) & 15) return RYOLO_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const float coef = obj_w / (float)cells;
    const long long n4 = total / 4;
    long long nb = (n4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    if (arc != 0) {
        const int unified = (arc & RYOLO_ARC_UBCE) ? 1 : ((arc & RYOLO_ARC_UCE) ? 2 : 0);
        const float cf = unified == 1 ? coef / (float)(nc > 1 ? nc : 1) : (unified == 2 ? cls_w / (float)cells : coef);
        long long nbc = (cells + 255) / 256;
        if (nbc > 4096) nbc = 4096;
        hipLaunchKernelGGL(yolo_loss_dense_arc_kernel, dim3((unsigned)nbc), dim3(256), 0, stream, (const __bf16 *)nullptr, 0, p, cells,
                           0ll, 1, na, no, nc, (arc & RYOLO_ARC_FOCAL) ? 1 : 0, unified, fl_gamma, cf, dp, (const unsigned *)nullptr,
                           (__bf16 *)nullptr, 0, items);
    } else if (no == 7) hipLaunchKernelGGL(yolo_loss_dense_kernel<7>, dim3((unsigned)nb), dim3(256), 0, stream, p, n4, total, no, coef, dp, items);
    else hipLaunchKernelGGL(yolo_loss_dense_kernel<0>, dim3((unsigned)nb), dim3(256), 0, stream, p, n4, total, no, coef, dp, items);
    if (hipGetLastError() != hipSuccess) return RYOLO_ELAUNCH;
    return launch_positives(p, dp, bs, na, ny, nx, no, nc, w, NT, b, gj, gi, cls, txy, twh, ta, anchor_vec, npos, giou, reg_w, cls_w,
                            cls_pw, coef, obj_pw, iou_mode, bitmap, items, stream, arc, fl_gamma);
}

}  // extern "C"
