// Layer 0's whole backward in ONE pass over dy (the autograd of /root/reference/model/models.py:49-66 for the first
// Conv2d + BatchNorm2d + LeakyReLU block: 3 -> 32 channels, 3x3 / stride 1 / pad 1, on the 8-channel padded NHWC input).
//
// The two-pass form (ryolo_conv0_bn_bwd + the stem weight-gradient kernel) reads dy twice, writes dz (1.5 GB at bs 64 / 608^2) and
// reads it back for the weight gradient: 8.6 GB, 1.63 ms per step -- at the HBM rate for that structure.  But nothing downstream
// needs dz itself: layer 0 has no data gradient, and its weight gradient is LINEAR in dz,
//     dz = scale * (g - S1/M - xhat * S2/M),   g = dy * act'(u),  xhat = (z - mean) * invstd,  S1 = sum g,  S2 = sum g * xhat
//     dW[co][k] = sum_p dz[p][co] * x[p][k]
//               = scale_co * ( G[co][k] - (S1_co / M) * Sx[k] - (S2_co * invstd_co / M) * (Z[co][k] - mean_co * Sx[k]) )
// with G = sum_p g x, Z = sum_p z x, Sx = sum_p x (k = (tap, input channel): 27 real columns).  All of them are sums over the
// pixels, so one pass that recomputes z from x (as the two-pass form does), forms g, and accumulates S1, S2, S3 (the slope
// gradient), G, Z and Sx is enough; a small finalise applies the formula.  Traffic: dy + x once = 1.9 GB.
//
// Kernel.  A wave walks groups of 16 consecutive pixels.  The conv is recomputed TRANSPOSED -- D[pixel][co] = x_patch * W^T, the
// x fragment (lane = pixel, 8 channels of one tap) as the A operand and the filter fragment as B -- so that a lane ends up with
// FOUR PIXELS of ONE output channel: exactly the A-operand layout (row = co, k = pixel) of v_mfma_f32_16x16x16_bf16 for the
// pixel contraction G += g^T x, Z += z^T x.  Its B operand (k = pixel, column = (tap, ci)) is gathered from the x fragments the
// wave has just loaded, through a wave-private LDS tile ([tap][pixel][8 ch], 272-B tap pitch: conflict-free 2-byte reads), with
// the 27 real columns packed into two 16-column fragments.  dy is read in the transposed layout directly (2-byte loads: every
// 32-B sector of dy is still fetched exactly once).  Requests run two groups ahead of the arithmetic.
// Partial sums: one fp32 row of 2176 values per workgroup (the four waves are combined in wave order through LDS), summed over
// the workgroups in a fixed order by the finalise kernels -- bit-reproducible, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ryolo.h"
#include "conv_common.h"

namespace {
using namespace ryolo_detail;

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

constexpr int C0_ROW = 2 * 32 * 32 + 32 + 96;      // G[32][32] | Z[32][32] | Sx[32] | S1[32] | S2[32] | S3[32]
constexpr int C0_XS = 12 * 272;                    // bytes of a wave's x tile: 12 tap slots (9 real) x (16 pixels x 16 B + 16 B pad)

struct C0Params {
    const __bf16 *x; unsigned x_bytes;             // [N, H, W, 8] bf16
    const __bf16 *w; int Kpad;                     // forward-packed filter [32][Kpad], k = tap * 8 + channel
    const __bf16 *dy; unsigned dy_bytes; int dy_cs;
    const float *scale, *shift, *mean;             // [32]
    const float *slope;                            // device scalar (leaky / PReLU) or nullptr
    int H, W, M, gpw;                              // M = N*H*W pixels, groups of 16 pixels per wave
    float *part;                                   // [gridDim.x][C0_ROW]
};

template <int ACT>
__global__ void __launch_bounds__(256) conv0_bwd_fused_kernel(const C0Params p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * C0_XS];
    static_assert(4 * C0_XS >= C0_ROW * 4, "the combine row reuses the x tiles");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (scalar: the group offsets below are SGPR operands)
    const int fr = lane & 15, g = lane >> 4;
    const long long wave_id = (long long)blockIdx.x * 4 + wave;
    const long long g0 = wave_id * p.gpw;
    const long long groups = ((long long)p.M + 15) / 16;
    const int n_it = g0 >= groups ? 0 : (int)(groups - g0 < p.gpw ? groups - g0 : p.gpw);      // (a wave without work still joins the combine)
    char *xs = smem + wave * C0_XS;

    bf16x8 wfr[2][3];                              // B operand of the transposed conv: column = co (16 cf + fr), k = 32 ks + 8 g ..
#pragma unroll
    for (int cf = 0; cf < 2; cf++)
#pragma unroll
        for (int ks = 0; ks < 3; ks++) wfr[cf][ks] = *(const bf16x8 *)(p.w + (size_t)(cf * 16 + fr) * p.Kpad + ks * 32 + g * 8);
    float sc[2], sh[2], mu[2];
#pragma unroll
    for (int cf = 0; cf < 2; cf++) {
        sc[cf] = p.scale[cf * 16 + fr];
        sh[cf] = p.shift[cf * 16 + fr];
        mu[cf] = p.mean[cf * 16 + fr];
    }
    const float slope = p.slope ? p.slope[0] : 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.x) - (size_t)(p.W + 1) * 8, 0, p.x_bytes + (unsigned)(p.W + 1) * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.dy), 0, p.dy_bytes, 0x00020000);
#endif
    // x fragments: lane (pixel fr, tap 4 ks + g); taps >= 9 are K padding
    int dkh[3], dkw[3];
    bool tok[3];
#pragma unroll
    for (int ks = 0; ks < 3; ks++) {
        const int tap = ks * 4 + g;
        dkh[ks] = ((tap * 11) >> 5) - 1;
        dkw[ks] = tap - 3 * ((tap * 11) >> 5) - 1;
        tok[ks] = tap < 9;
    }
    // gather of the pixel-contraction's B operand: lane (column k'' = 16 j + fr = tap * 3 + ci, pixels 4 g .. 4 g + 3)
    int xb_off[2];
    bool xb_ok[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int kk = 16 * j + fr, tap = kk / 3, ci = kk - 3 * tap;
        xb_ok[j] = kk < 27;
        xb_off[j] = (xb_ok[j] ? tap : 9) * 272 + (4 * g) * 16 + ci * 2;
    }
    f32x4 G[2][2], Z[2][2];
#pragma unroll
    for (int cf = 0; cf < 2; cf++)
#pragma unroll
        for (int j = 0; j < 2; j++) G[cf][j] = Z[cf][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 SX[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float b1[2] = {0.f, 0.f}, b2[2] = {0.f, 0.f}, b3[2] = {0.f, 0.f};

    int m = (int)(g0 * 16) + fr;                   // this lane's pixel of the group being REQUESTED
    int wo, ho;
    {
        const int mc = m < p.M ? m : p.M - 1;
        const int t = mc / p.W;
        wo = mc - t * p.W;
        ho = t - (t / p.H) * p.H;
    }
    // Addresses.  Pixel (img, hi, wi) of the NHWC input has the linear index m + dkh * W + dkw, so a request is a wave-uniform group
    // base (the scalar offset operand) plus LANE-CONSTANT vector offsets; the x descriptor starts (W + 1) pixels in front of the tensor so
    // that the offsets of the taps above / left of the pixel are not negative (such lanes are masked or inside the tensor).  Only the
    // padding test needs the lane's coordinates.  dy: rows past M fall outside the descriptor (zeros).
    int vx[3], vd[2][4];
#pragma unroll
    for (int ks = 0; ks < 3; ks++) vx[ks] = (fr + (dkh[ks] + 1) * p.W + (dkw[ks] + 1)) * 16;
#pragma unroll
    for (int cf = 0; cf < 2; cf++)
#pragma unroll
        for (int r = 0; r < 4; r++) vd[cf][r] = ((4 * g + r) * p.dy_cs + cf * 16 + fr) * 2;
    long long gcur = g0;                           // group being requested (wave-uniform)
    auto request = [&](u4(&x)[3], unsigned(&d)[2][4], bool live) {      // x fragments + dy (transposed layout) of the current group; then + 16 pixels
        const int lim = live ? p.M : 0;            // (one scalar select: with `live &&` in every address the compiler threads the whole request into branches)
        const int sx_off = live ? (int)(gcur * 256) : 0, sd_off = live ? (int)(gcur * 16 * p.dy_cs * 2) : 0;
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            const bool ok = m < lim && tok[ks] && (unsigned)(ho + dkh[ks]) < (unsigned)p.H && (unsigned)(wo + dkw[ks]) < (unsigned)p.W;
#if defined(__HIP_DEVICE_COMPILE__)
            x[ks] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? vx[ks] : (int)0x80000000, sx_off, 0);
#endif
        }
#pragma unroll
        for (int cf = 0; cf < 2; cf++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
#if defined(__HIP_DEVICE_COMPILE__)
                d[cf][r] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(drs, live ? vd[cf][r] : (int)0x80000000, sd_off, 0);
#endif
            }
        gcur++;
        m += 16;                                   // W >= 16: at most one row wrap
        wo += 16;
        const bool wrap = wo >= p.W;
        wo -= wrap ? p.W : 0;
        ho += wrap ? 1 : 0;
        const bool wrap2 = ho >= p.H;
        ho = wrap2 ? 0 : ho;
    };
    auto finish = [&](const u4(&x)[3], const unsigned(&d)[2][4]) {
        // the x tile for the gather goes out first: its LDS latency hides under the conv's MFMAs
#pragma unroll
        for (int ks = 0; ks < 3; ks++) *(u4 *)(xs + (4 * ks + g) * 272 + fr * 16) = x[ks];
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 3; ks++)
#pragma unroll
            for (int cf = 0; cf < 2; cf++)       // D[pixel 4 g + r][co 16 cf + fr]
                acc[cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x[ks]), wfr[cf][ks], acc[cf], 0, 0, 0);
        bf16x4 ga[2], za[2];
#pragma unroll
        for (int cf = 0; cf < 2; cf++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                // z as the forward used it (rounded to bf16), u = z * scale + shift, g = dy * act'(u); a pixel past M has x = dy = 0: no contribution
                const __bf16 zb = (__bf16)acc[cf][r];
                const float zf = (float)zb;
                const float dd = __builtin_bit_cast(float, d[cf][r] << 16);
                const float u = zf * sc[cf] + sh[cf];
                float gg = dd;
                if constexpr (ACT == RYOLO_ACT_LEAKY) {
                    if (u <= 0.f) {
                        gg = dd * slope;
                        b3[cf] += dd * u;
                    }
                } else if constexpr (ACT == RYOLO_ACT_MISH) {
                    const float e = __expf(fminf(u, 20.f)), n1 = (1.f + e) * (1.f + e), t = (n1 - 1.f) / (n1 + 1.f);
                    gg = dd * (t + u * (1.f - t * t) * (e / (1.f + e)));
                }
                b1[cf] += gg;
                b2[cf] += gg * (zf - mu[cf]);
                ga[cf][r] = (__bf16)gg;
                za[cf][r] = zb;
            }
        bf16x4 xb[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned short h = *(const unsigned short *)(xs + xb_off[j] + r * 16);
                xb[j][r] = __builtin_bit_cast(__bf16, h);       // (columns >= 27 read the all-zero slot of tap 9)
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int cf = 0; cf < 2; cf++)
#pragma unroll
            for (int j = 0; j < 2; j++) {          // D[co 16 cf + 4 g + r][k'' 16 j + fr] += sum over the group's 16 pixels
                G[cf][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ga[cf]), __builtin_bit_cast(s16x4, xb[j]), G[cf][j], 0, 0, 0);
                Z[cf][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, za[cf]), __builtin_bit_cast(s16x4, xb[j]), Z[cf][j], 0, 0, 0);
            }
#pragma unroll
        for (int j = 0; j < 2; j++)                // Sx: a row of ones as the A operand (every row of SX is the column sum; the VALU has no slack, the MFMA pipe does)
            SX[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(s16x4{0x3f80, 0x3f80, 0x3f80, 0x3f80}, __builtin_bit_cast(s16x4, xb[j]), SX[j], 0, 0, 0);
#endif
    };
    u4 xa[3], xb_[3], xc[3];
    unsigned da[2][4], db[2][4], dc[2][4];      // (one dy value per register: packed pairs made the compiler wait for each 2-byte load in turn)
    request(xa, da, 0 < n_it);
    request(xb_, db, 1 < n_it);
    for (int it = 0; it < n_it; it += 3) {
        request(xc, dc, it + 2 < n_it);
        finish(xa, da);
        request(xa, da, it + 3 < n_it);
        if (it + 1 < n_it) finish(xb_, db);
        request(xb_, db, it + 4 < n_it);
        if (it + 2 < n_it) finish(xc, dc);
    }
    // the four pixel groups (g) of a channel / column: VALU lane sums, every lane of the class ends with the total
#pragma unroll
    for (int cf = 0; cf < 2; cf++) {
        b1[cf] = lane_xor_sum<32>(lane_xor_sum<16>(b1[cf]));
        b2[cf] = lane_xor_sum<32>(lane_xor_sum<16>(b2[cf]));
        b3[cf] = lane_xor_sum<32>(lane_xor_sum<16>(b3[cf]));
    }
    // workgroup row: the waves add in wave order (fixed), then one coalesced store
    float *row = (float *)smem;
    __syncthreads();                               // every wave is done with its x tile
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int cf = 0; cf < 2; cf++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int idx = (16 * cf + 4 * g + r) * 32 + 16 * j + fr;
                        row[idx] = (w ? row[idx] : 0.f) + G[cf][j][r];
                        row[1024 + idx] = (w ? row[1024 + idx] : 0.f) + Z[cf][j][r];
                    }
            if (g == 0) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int i = 16 * c + fr;
                    row[2048 + i] = (w ? row[2048 + i] : 0.f) + SX[c][0];      // (lanes g == 0 hold rows 0..3 of SX: all equal to the column sum)
                    row[2080 + i] = (w ? row[2080 + i] : 0.f) + b1[c];
                    row[2112 + i] = (w ? row[2112 + i] : 0.f) + b2[c];
                    row[2144 + i] = (w ? row[2144 + i] : 0.f) + b3[c];
                }
            }
        }
        __syncthreads();
    }
    float *out = p.part + (size_t)blockIdx.x * C0_ROW;
    for (int i = threadIdx.x; i < C0_ROW; i += 256) out[i] = row[i];
}

// rows [R][C0_ROW] -> totals [C0_ROW] (fp64), fixed order: block = 32 columns x 32 row lanes
__global__ void __launch_bounds__(1024) conv0_bwd_rows_kernel(const float *__restrict__ part, int R, double *__restrict__ tot) {
    __shared__ double red[32][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double a = 0.0;
    if (c < C0_ROW) {
        for (int r0 = ry; r0 < R; r0 += 32 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = r0 + 32 * u < R ? part[(size_t)(r0 + 32 * u) * C0_ROW + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; u++) a += (double)v[u];
        }
    }
    red[ry][cx] = a;
    __syncthreads();
    if (ry != 0 || c >= C0_ROW) return;
    for (int k = 1; k < 32; k++) a += red[k][cx];
    tot[c] = a;
}

// totals -> dgamma += S2 * invstd, dbeta += S1, dslope += sum_c S3 (fixed order), dW (OIHW fp32, the 3 real input channels)
__global__ void __launch_bounds__(1024) conv0_bwd_finish_kernel(const double *__restrict__ tot, const float *__restrict__ scale,
                                                                const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                double inv_count, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                float *__restrict__ dslope, float *__restrict__ dw, int cin_real,
                                                                int accumulate) {
    const int t = threadIdx.x;
    if (t < 32) {
        const float s1 = (float)tot[2080 + t], s2 = (float)(tot[2112 + t] * (double)invstd[t]);
        if (dgamma) dgamma[t] += s2;
        if (dbeta) dbeta[t] += s1;
        if (dslope) {
            float v = (float)tot[2144 + t];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
            if (t == 0) dslope[0] += v;
        }
    }
    if (dw && t < 32 * 27) {
        const int co = t / 27, kk = t - 27 * co, tap = kk / 3, ci = kk - 3 * tap;
        if (ci < cin_real) {
            const double S1 = tot[2080 + co], S2 = tot[2112 + co] * (double)invstd[co];     // sum g, sum g * xhat
            const double Gv = tot[co * 32 + kk], Zv = tot[1024 + co * 32 + kk], Sx = tot[2048 + kk];
            const double v = (double)scale[co] * (Gv - S1 * inv_count * Sx - S2 * inv_count * (double)invstd[co] * (Zv - (double)mean[co] * Sx));
            const size_t dst = ((size_t)co * cin_real + ci) * 9 + tap;
            dw[dst] = accumulate ? dw[dst] + (float)v : (float)v;
        }
    }
}

inline int c0_grid() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    return 4 * cus;
}

}  // namespace

extern "C" {

size_t ryolo_conv0_bn_bwd_wgrad_workspace_bytes(void) { return ((size_t)c0_grid() * C0_ROW * 4 + 255) / 256 * 256 + (size_t)C0_ROW * 8; }

int ryolo_conv0_bn_bwd_wgrad(const ryolo_conv_desc *d, const void *x, const void *w_packed, const void *dy, int dy_cstride,
                             const float *scale, const float *shift, const float *mean, const float *invstd, int act, const float *slope,
                             float *dgamma, float *dbeta, float *dslope, float *grad_oihw, int cin_real, int accumulate, void *workspace,
                             size_t workspace_bytes, void *stream_) {
    if (!ryolo_conv0_recompute_supported(d) || !x || !w_packed || !dy || !scale || !shift || !mean || !invstd || !workspace) return RYOLO_EINVAL;
    if (workspace_bytes < ryolo_conv0_bn_bwd_wgrad_workspace_bytes() || (dy_cstride & 7) || dy_cstride < 32) return RYOLO_EINVAL;
    if (act < 0 || act > 2 || (act == RYOLO_ACT_LEAKY && !slope) || cin_real < 1 || cin_real > 3) return RYOLO_EINVAL;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)w_packed | (uintptr_t)workspace) & 15) return RYOLO_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    C0Params p;
    p.x = (const __bf16 *)x; p.w = (const __bf16 *)w_packed; p.Kpad = (72 + 63) / 64 * 64;
    p.dy = (const __bf16 *)dy; p.dy_cs = dy_cstride;
    p.scale = scale; p.shift = shift; p.mean = mean; p.slope = act == RYOLO_ACT_LEAKY ? slope : nullptr;
    p.H = d->H; p.W = d->W;
    const long long M = (long long)d->N * d->H * d->W;
    if (M <= 0 || (M + 64) * (long long)dy_cstride * 2 >= 0x7fffff00ll || d->W < 16) return RYOLO_EINVAL;
    p.M = (int)M;
    p.x_bytes = (unsigned)((unsigned long long)M * 16ull);
    p.dy_bytes = (unsigned)((((unsigned long long)M - 1) * dy_cstride + 32) * 2ull);
    const long long groups = (M + 15) / 16;
    int grid = c0_grid();
    long long gpw = (groups + (long long)grid * 4 - 1) / ((long long)grid * 4);
    if (gpw < 1) gpw = 1;
    grid = (int)((groups + gpw * 4 - 1) / (gpw * 4));
    p.gpw = (int)gpw;
    p.part = (float *)workspace;
    double *tot = (double *)((char *)workspace + ((size_t)c0_grid() * C0_ROW * 4 + 255) / 256 * 256);
    if (act == RYOLO_ACT_LEAKY) hipLaunchKernelGGL(conv0_bwd_fused_kernel<RYOLO_ACT_LEAKY>, dim3(grid), dim3(256), 0, stream, p);
    else if (act == RYOLO_ACT_MISH) hipLaunchKernelGGL(conv0_bwd_fused_kernel<RYOLO_ACT_MISH>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(conv0_bwd_fused_kernel<RYOLO_ACT_LINEAR>, dim3(grid), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(conv0_bwd_rows_kernel, dim3((C0_ROW + 31) / 32), dim3(1024), 0, stream, p.part, grid, tot);
    hipLaunchKernelGGL(conv0_bwd_finish_kernel, dim3(1), dim3(1024), 0, stream, tot, scale, mean, invstd, 1.0 / (double)M, dgamma, dbeta,
                       act == RYOLO_ACT_LEAKY ? dslope : nullptr, grad_oihw, cin_real, accumulate);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

}  // extern "C"
