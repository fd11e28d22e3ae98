// rotate-yolov3_amd/csrc/conv_tw.hip -- implicit-GEMM convolution tile for gfx950 with TWO independent workgroups per CU.
//
// Same operator and operand layout as conv_mp.hip (model/models.py:49-66 conv -> BN -> activation, :281-282 shortcut).
// conv_mp.hip runs ONE 8-wave workgroup per CU: its K loop sustains ~65 % of the MFMA rate, but all eight waves reach the
// epilogue together, so for the 18-K-tile layers a third of the tile time (scale/shift/activation/residual/convert is
// VALU-bound, ~800 instructions per wave) has no MFMA under it.  Here a workgroup is 4 waves on a 128-pixel x 256-channel
// tile (the same 128 x 64 wave tile: 12 ds_read_b128 per 32 MFMAs), K step 32, three LDS stages of 24 KiB -> 72 KiB, so TWO
// workgroups fit a CU and drift apart: one's epilogue and load prologue run under the other's MFMAs.  The price is that
// each workgroup stages the whole 256-channel weight tile for half the pixels (L2->LDS fill x1.5).
//   * operands HBM->LDS by 16-B direct-to-LDS loads issued as inline assembly (invisible to the compiler's waitcnt pass, see
//     train.hip), every wave issues exactly NLD = 6 per stage (out-of-range ones with the buffer's out-of-bounds offset);
//   * the fills of K step k+2 are issued under the MFMAs of step k and retired by a counted s_waitcnt vmcnt(6); ONE barrier
//     per K step: it proves stage k has landed for every wave and that everyone is done reading the buffer of step k-1,
//     which stage k+2 overwrites;
//   * LDS image of a stage = [row][4 x 16-B slots] (64-B rows), slot ^= (row >> 2) & 3: the 16 rows a ds_read_b128 service
//     group touches sit on distinct banks; the XOR is applied on the source address of the fill and on the read;
//   * epilogue = conv_mp.hip's register-direct one (permlane regroup, whole 64-B half lines per store).
// Stride 1, C_in % 32 == 0, C_out % 256 == 0, dense output placement, no statistics (the training forward stays on conv_mp).
//
// STATUS (round 2): an EXPERIMENT, reachable only through tile code 24 (tests, tools/mp_tune.py) -- measured SLOWER than
// conv_mp on the 3x3 layers it was written for (bs 32: 128->256@76^2 134 vs 121 us, 256->512@38^2 123 vs 113, 512->1024@19^2
// 127 vs 102; only the 1x1 1024->512@19^2 gains, 26 vs 30 us).  A workgroup alone on a CU needs ~1300 cycles per K step for
// 512 cycles of MFMA (barrier + fills + 12 fragment reads + MFMAs in sequence, all four waves in the same state), and two
// resident workgroups each slow to ~2770 cycles: they do not fill each other's gaps, MFMA 37 %, LDS 41 % busy.  The
// epilogue overlap this design buys is smaller than what the 64-deep, 4-phase, staggered K loop of conv_mp is worth.
// (Its Mish epilogue used to store wrong values at tile rows 92-95 of two channel pairs: the store-data hazard described at
// conv_common.h buffer_store16_soff(), which this kernel's register allocation happened to hit first.)
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

using namespace ryolo_detail;

namespace {

constexpr int TW_BM = 128, TW_BN = 256, TW_BK = 32, TW_NST = 3;
constexpr int TW_XS = TW_BM * TW_BK * 2, TW_WS = TW_BN * TW_BK * 2, TW_STAGE = TW_XS + TW_WS;   // 8 KiB + 16 KiB
constexpr int TW_LDS = TW_NST * TW_STAGE;
constexpr int TW_NLD = 6;                                 // direct-to-LDS loads per wave per stage: 2 activation + 4 weight pieces

template <int N> using ic = std::integral_constant<int, N>;

typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 tw_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// s_nop: one wait state between the SALU write of M0 and the LDS-DMA instruction that reads it
__device__ __forceinline__ void tw_load_lds16(i32x4 rsrc, unsigned lds_addr /* wave-uniform */, int voffset, int soffset /* uniform */) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voffset), "s"(rsrc),
                 "s"(soffset)
                 : "memory", "m0");
#endif
}

// two workgroups per CU = two waves per SIMD: the register budget is 256 (arch + accumulation), not the 512 a lone 4-wave group could take
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_tw_kernel(const ConvParams p) {
    constexpr int PF = TW_BM / 16;        // 8 pixel fragments per wave (every wave sees all 128 pixels)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [TW_NST][X 128 x 64 B | W 256 x 64 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = channel quarter
    const int frow = lane & 15, fk = lane >> 4, fr4 = fk * 4;

    // tile of this workgroup: XCD-contiguous ranges of the tile list (channel tile fastest), like the other conv kernels
    int b = blockIdx.x;
    {
        const int per = gridDim.x >> 3;
        if (b < per * 8) b = (b & 7) * per + (b >> 3);
    }
    const int nt = p.nt;
    const int n_t = b % nt, m_t = b / nt;
    const int m0 = m_t * TW_BM, n0 = n_t * TW_BN;

    // ---- staging bookkeeping.  A piece is 16 tile rows x 64 B; lane l fills 16-B slot (l & 3) of row (l >> 2).
    const int prow = lane >> 2, pslot = lane & 3;
    int x_off[2];
    unsigned x_mask[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = (wave * 2 + i) * 16 + prow;                 // tile pixel row
        const int chunk = pslot ^ ((r >> 2) & 3);
        const int m = m0 + r;
        unsigned mk = 0;
        int off = 0;
        if (m < p.M) {
            int wo, ho, img;
            split_pixel(m, p.Wo, p.Ho, p.magic_wo, p.magic_ho, p.use_magic, wo, ho, img);
            const int hi0 = ho - p.pad, wi0 = wo - p.pad;
            off = (((img * p.H + hi0) * p.W + wi0) * p.in_cs) * 2 + chunk * 16;
#pragma unroll
            for (int t = 0; t < 9; t++) {     // branch-free (a scalar branch per tap and piece adds up in the tile prologue)
                const int hi = hi0 + p.tap_dy[t], wi = wi0 + p.tap_dx[t];
                const unsigned in = (unsigned)(t < p.ntaps) & (unsigned)((unsigned)hi < (unsigned)p.H) & (unsigned)((unsigned)wi < (unsigned)p.W);
                mk |= in << t;
            }
        }
        x_off[i] = off;
        x_mask[i] = mk;
    }
    int w_off[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = (wave * 4 + i) * 16 + prow;                 // tile channel row
        const int chunk = pslot ^ ((r >> 2) & 3);
        w_off[i] = ((n0 + r) * p.Kpad + chunk * 8) * 2;
    }
    int lane_tapoff;                        // lane t keeps the byte offset of tap t
    {
        const int t = lane < p.ntaps ? lane : 0;
        lane_tapoff = ((p.tap_dy[t] * p.W + p.tap_dx[t]) * p.in_cs) * 2;
    }
    const i32x4 rs_x = tw_rsrc(p.x, p.x_bytes), rs_w = tw_rsrc(p.w, p.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const int cin_bytes = p.Cin * 2;
    const int KT = p.Kpad / TW_BK, KREAL = (p.ntaps * p.Cin) / TW_BK;

    // K position of the stage being issued (advanced once per stage())
    int s_tap = 0, s_cb = 0, s_kt = 0;
    auto stage = [&](int buf) __attribute__((always_inline)) {    // exactly TW_NLD loads per wave, in or out of range
        const unsigned xb = lds0 + buf * TW_STAGE, wb = xb + TW_XS;
        const bool live = s_kt < KREAL;                           // K padding / look-ahead past the end: zeros
        const int tap = __builtin_amdgcn_readfirstlane(live ? s_tap : 0);
        const int tapoff = __builtin_amdgcn_readlane(lane_tapoff, tap) + s_cb;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const bool ok = live && ((x_mask[i] >> tap) & 1u);
            tw_load_lds16(rs_x, xb + (wave * 2 + i) * 1024, ok ? x_off[i] + tapoff : (int)0x80000000, 0);
        }
        const int wsoff = __builtin_amdgcn_readfirstlane(s_kt * (TW_BK * 2));
#pragma unroll
        for (int i = 0; i < 4; i++)
            tw_load_lds16(rs_w, wb + (wave * 4 + i) * 1024, s_kt < KT ? w_off[i] : (int)0x80000000, s_kt < KT ? wsoff : 0);
        s_cb += TW_BK * 2;
        s_kt++;
        if (s_cb >= cin_bytes) { s_cb = 0; s_tap++; }
        s_cb = __builtin_amdgcn_readfirstlane(s_cb);
        s_kt = __builtin_amdgcn_readfirstlane(s_kt);
        s_tap = __builtin_amdgcn_readfirstlane(s_tap);
    };

    // fragment read offsets inside a stage: row * 64 + swizzled 16-B slot (the swizzle key (row >> 2) & 3 only depends on frow)
    const int sw = ((fk ^ ((frow >> 2) & 3)) << 4) + frow * 64;
    const int xrd = sw, wrd = TW_XS + (wave * 64) * 64 + sw;

    f32x4 acc[4][PF];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int f = 0; f < PF; f++) acc[c][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0);
    stage(1);
    int cur = 0, nxt = 2;
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < KT; kt++) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TW_NLD) : "memory");
        __builtin_amdgcn_s_barrier();
        stage(nxt);
        const char *base = smem + cur * TW_STAGE;
        bf16x8 wf[4], xf[PF];
#pragma unroll
        for (int c = 0; c < 4; c++) wf[c] = *(const bf16x8 *)(base + wrd + c * 1024);
#pragma unroll
        for (int f = 0; f < PF; f++) xf[f] = *(const bf16x8 *)(base + xrd + f * 1024);
#pragma unroll
        for (int f = 0; f < PF; f++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xf[f], acc[c][f], 0, 0, 0);
        cur = cur == TW_NST - 1 ? 0 : cur + 1;
        nxt = nxt == TW_NST - 1 ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the look-ahead fills behind the last K step

    // ---------------------------------------------------------------- epilogue (registers -> global, no LDS): conv_mp.hip's
    const float slope = p.slope;
    auto run_epilogue = [&](auto ACTc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(ACTc)::value;
        const int chq = n0 + wave * 64;                   // first channel of this wave's quarter
        const int mrow = m0 + frow;
#if defined(__HIP_DEVICE_COMPILE__)
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.y), 0, p.res ? p.res_bytes : 0u, 0x00020000);
#endif
        const int yoff0 = (mrow * p.out_cs + chq + fk * 8) * 2, roff0 = (mrow * p.res_cs + chq + fk * 8) * 2;
        const int ystep = 16 * p.out_cs * 2, rstep = 16 * p.res_cs * 2;
        u32x4 rv[PF][2];
        if (p.res) {
#pragma unroll
            for (int f = 0; f < PF; f++) {
                const int voff = (mrow + f * 16) < p.M ? roff0 : (int)0x80000000;
#if defined(__HIP_DEVICE_COMPILE__)
                rv[f][0] = __builtin_amdgcn_raw_buffer_load_b128(rrs, voff, f * rstep, 0);
                rv[f][1] = __builtin_amdgcn_raw_buffer_load_b128(rrs, voff + 64, f * rstep, 0);
#endif
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            f32x4 sc[2], sh[2];
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                sc[cc] = *(const f32x4 *)(p.scale + chq + (2 * h + cc) * 16 + fr4);
                sh[cc] = *(const f32x4 *)(p.shift + chq + (2 * h + cc) * 16 + fr4);
            }
#pragma unroll
            for (int f = 0; f < PF; f++) {
                const bool ok = (mrow + f * 16) < p.M;
                unsigned R[2][2];
#pragma unroll
                for (int cc = 0; cc < 2; cc++) {
                    const int c = 2 * h + cc;
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float v = acc[c][f][r] * sc[cc][r] + sh[cc][r];
                        if constexpr (ACT == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                        else if constexpr (ACT == 3) v = fmaxf(v, v * slope);   // leaky with slope <= 1
                        else if constexpr (ACT == RYOLO_ACT_MISH) v = mish(v);
                        o[r] = (__bf16)v;
                    }
                    const uint2 u = __builtin_bit_cast(uint2, o);
                    R[cc][0] = u.x;
                    R[cc][1] = u.y;
                }
                // regroup the 8-B units of a pixel so the four lanes of the pixel hold 16 B each of 64 CONTIGUOUS bytes
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("s_nop 4" : "+v"(R[0][0]), "+v"(R[0][1]), "+v"(R[1][0]), "+v"(R[1][1]));
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    auto s1 = __builtin_amdgcn_permlane32_swap(R[0][d], R[1][d], false, false);
                    auto s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                    R[0][d] = s2[0];
                    R[1][d] = s2[1];
                }
#endif
                u32x4 out = u32x4{R[0][0], R[0][1], R[1][0], R[1][1]};
                if (p.res) {
                    bf16x8 a = __builtin_bit_cast(bf16x8, out);
                    const bf16x8 bb = __builtin_bit_cast(bf16x8, rv[f][h]);
#pragma unroll
                    for (int e = 0; e < 8; e++) a[e] = (__bf16)((float)a[e] + (float)bb[e]);
                    out = __builtin_bit_cast(u32x4, a);
                }
#if defined(__HIP_DEVICE_COMPILE__)
                buffer_store16_soff(out, yrs, (ok ? yoff0 : (int)0x80000000) + 64 * h, f * ystep);
#endif
            }
        }
    };
    if (p.act == RYOLO_ACT_LEAKY && p.slope <= 1.f) run_epilogue(ic<3>{});
    else if (p.act == RYOLO_ACT_LEAKY) run_epilogue(ic<RYOLO_ACT_LEAKY>{});
    else if (p.act == RYOLO_ACT_MISH) run_epilogue(ic<RYOLO_ACT_MISH>{});
    else run_epilogue(ic<RYOLO_ACT_LINEAR>{});
}

inline unsigned tw_magic_u32(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

}  // namespace

namespace ryolo_detail {

bool conv_tw_eligible(const ConvParams &p) {
    return p.fast && !p.taps2 && p.ups == 1 && p.stride == 1 && p.os == 1 && !p.stat_part && (p.Cin % TW_BK) == 0 &&
           (p.Cout % TW_BN) == 0 && p.ntaps >= 1 && p.ntaps <= 9 && p.Kpad >= p.ntaps * p.Cin && p.Kpad >= 2 * TW_BK &&
           (p.Kpad % TW_BK) == 0;
}

int launch_conv_tw(ConvParams &p, hipStream_t stream) {
    if (!conv_tw_eligible(p)) return RYOLO_EINVAL;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)conv_tw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS) != hipSuccess)
            return RYOLO_ELAUNCH;
        attr_done = true;
        if (getenv("RYOLO_TW_DEBUG")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)conv_tw_kernel, 256, TW_LDS);
            fprintf(stderr, "conv_tw: resident workgroups per CU = %d (LDS %d B)\n", nb, TW_LDS);
        }
    }
    const int mt = (p.M + TW_BM - 1) / TW_BM;
    p.nt = p.Cout / TW_BN;
    const long long T = (long long)mt * p.nt;
    const long long dmax = p.Wo > p.Ho ? p.Wo : p.Ho, mpad = (long long)mt * TW_BM;
    if (mpad * dmax >= 0x100000000ll || T > 0x7fffffffll) return RYOLO_EINVAL;
    p.use_magic = 1;
    p.magic_wo = tw_magic_u32(p.Wo);
    p.magic_ho = tw_magic_u32(p.Ho);
    p.ntiles = (int)T;
    const unsigned long long yb = (((unsigned long long)p.N * p.OH * p.OW - 1) * p.out_cs + p.Cout) * 2ull;
    const unsigned long long rb = p.res ? (((unsigned long long)p.N * p.OH * p.OW - 1) * p.res_cs + p.Cout) * 2ull : 0ull;
    if (yb >= 0x7fffff00ull || rb >= 0x7fffff00ull) return RYOLO_EINVAL;
    p.y_bytes = (unsigned)yb;
    p.res_bytes = (unsigned)rb;
    hipLaunchKernelGGL(conv_tw_kernel, dim3((unsigned)T), dim3(256), TW_LDS, stream, p);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

}  // namespace ryolo_detail
