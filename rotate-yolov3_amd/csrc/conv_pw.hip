// rotate-yolov3_amd/csrc/conv_pw.hip -- the 1x1 ("pointwise") convolutions of the Darknet-53 stack on gfx950, WEIGHT-STATIONARY:
// the filter slice of a wave lives in its registers for the whole launch and the activation rows stream through a deep LDS ring.
//
// Operator: the same fused block as conv.hip (model/models.py:49-66 conv -> BN -> PReLU / Mish, :281-282 shortcut, :93-94
// upsample) for ksize 1 / stride 1, and its stride-1 data gradient (autograd of the same lines).
//
// Why a kernel of its own (profiles/r03_bench_per_op_events.txt): a 1x1 layer is a GEMM with a huge M
// (pixels), a small N and a K loop of only 4..16 steps of 64.  On the 128 x 128 tile of conv.hip every K step is one HBM round
// trip (double buffer, one stage in flight, ~1.1 us issued -> landed under load) and the 38^2 / 19^2 layers have at most 1.4
// tiles per workgroup, so the launch is a chain of exposed latencies: 512->256@38^2 ran at 0.17 of MFMA AND 0.39 of HBM.
//   * weights: wave w of the workgroup owns output channels n0 + w*CF*16 .. and keeps their whole K extent as MFMA A-operand
//     fragments in registers (CF*K/32 x 4 VGPRs: 64 for 256->128, 128 for 512->256 and for 1024->(128 of 512)); they are read
//     from L2 once per workgroup instead of once per tile, need no LDS, no barrier and no refill;
//   * activations: a row block (BMS = 64 or 128 pixels) enters LDS as KT units of BMS x 64 channels (16-B direct-to-LDS
//     buffer loads, lane-linear writes, XOR swizzle on the source address) in a ring of RING units; RING-1 units are always in
//     flight per workgroup (88 KB per CU for the 8-wave shapes) and the stream never stops at a row-block boundary: the fills
//     of the next row block are issued while this one is multiplied and stored.  One counted s_waitcnt + one s_barrier per unit;
//   * every wave reads ALL rows of a unit (the waves split the channels), so a fragment read feeds CF MFMAs and the per-CU LDS
//     read traffic is NW x the activation stream -- at most half of the LDS bandwidth for every shape served;
//   * epilogue from registers (no LDS): scale/shift/act -> bf16 -> v_permlane16_swap so that every lane owns a 16-B run ->
//     buffer store; BatchNorm statistics (training forward), the accumulate operand (data gradient into a running gradient) and the
//     folded BatchNorm-backward reduce keep their per-channel state in registers for the whole launch (a lane's channels never
//     change) and leave through DPP row sums once per workgroup;
//   * work split: XCD x (= blockIdx & 7) owns a contiguous chunk of row blocks; inside an XCD the workgroups are (row slot,
//     channel block) pairs, channel block fastest, so that the NB workgroups that read the same rows run on one L2.
// Accumulation order over K is that of conv_igemm_kernel (K ascending, two 32-wide MFMAs per 64-wide unit): outputs are
// bit-identical to the tile it replaces (tests/test_conv_gpu.py::test_conv_pw_*).
//
// In-order queue arithmetic (vector-memory operations retire in issue order on gfx9; stores and loads share vmcnt).  Iteration
// u = (row block i, unit kt) does: [wait for unit u] [barrier] [kt == 0: request the row block's accumulate / z rows] [fill unit
// u + RING - 1] [MFMAs] [kt == KT-1: epilogue stores].  Unit u's fill was issued in iteration u - (RING-1); younger than it at
// the wait are the fills of units u+1 .. u+RING-2, the stores of every epilogue in iterations u-(RING-1) .. u-1 and the row-block
// requests of iterations u-(RING-1)+1 .. u-1.  kt is unrolled, so these counts are compile-time per kt once the window lies
// inside the launch (i >= I0); the first I0 row blocks count only the operations of their own row block (fewer allowed in
// flight than there are: never unsafe).
#include <type_traits>
#include <utility>

#include "conv_common.h"
#include "yolo_decode.h"

using namespace ryolo_detail;

namespace {

template <int N> using ic = std::integral_constant<int, N>;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(ic<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// The counted waits below are arithmetic on the ORDER in which this wave's vector-memory operations were issued.  The compiler is free
// to reorder independent loads (it hoisted three filter loads in front of the prologue's fills: the wait for unit 0 then allowed three
// operations too many in flight -- a rare wrong tile, found as a flaky test).  Every group of issues is therefore closed by a
// compiler-level memory fence; tools/check_mp_isa.py::check_conv_pw verifies the issue order of every instantiation in the ISA.
__device__ __forceinline__ void issue_order_fence() { asm volatile("" ::: "memory"); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : (N < 0 ? 0 : N)) : "memory");
}

// First row block: vector-memory operations issued after the fill of unit kt and before the wait of iteration kt.  Issue order: prologue
// = fills of units 0 .. RING-2, then the filter fragments of K steps 0 .. D-1 (WL loads each); iteration t = [t == 0: NREQ row-block
// requests] [fill of unit t + RING - 1] [filter fragments of K step t + D].
constexpr int pw_first_block_younger(int kt, int KT, int RING, int LPU, int WL, int NREQ, int D) {
    int after = 0;
    bool seen = false;
    for (int u = 0; u < RING - 1; u++) {
        if (seen) after += LPU;
        if (u == kt) seen = true;
    }
    for (int j = 0; j < D && j < KT; j++)
        if (seen) after += WL;
    for (int t = 0; t < kt; t++) {
        if (t == 0 && seen) after += NREQ;
        if (seen) after += LPU;
        if (t + RING - 1 == kt) seen = true;
        if (t + D < KT && seen) after += WL;
    }
    return after;
}

// MODE 0 inference (act, optional 2x upsample)   1 training forward: + per-channel sums of z, z^2 (fp64 partial rows)
// MODE 2 + accumulate operand (p.res): data gradient into a running gradient, or a forward with a shortcut
// MODE 3 = 2 + the BatchNorm-backward reduce of the block whose output gradient this launch stores (conv.hip: BnRed)
struct PwBnRed {
    const __bf16 *z;
    int z_cs;
    unsigned z_bytes;
    const float *scale, *shift, *mean, *invstd, *slope;
    float *part;            // [gridDim.x][3][C]
};

// Timing-only ablations (-DRYOLO_MP_ABLATION, tools/pw_ablate.py; wrong results on purpose): p.dbg0 bit 0 no filter loads, 1 no
// activation fills (every lane out of range: zeros, no traffic), 2 no MFMAs, 3 no stores, 4 no barriers, 5 no fragment reads
#ifdef RYOLO_MP_ABLATION
#define PW_DBG(bit) ((p.dbg0 >> (bit)) & 1)
#else
#define PW_DBG(bit) 0
#endif

// MODE 4 (inference): a YOLO head.  The 1x1 conv's values (+ bias, rounded to bf16 exactly as the stored head tensor would be) go
// to an LDS tile [rows][channels] instead of HBM, and the workgroup decodes them there -- YOLOLayer.forward (model/models.py:183-227),
// the arithmetic of yolo.hip's decode kernels (shared header) -- writing the io / p rows.  The head tensor (186 MB at 76^2, bs 32) is
// neither written nor re-read.
struct PwDecode {
    float *io, *p;          // [bs][io_img_rows][no] (+ row offset io_row0), [bs][na][ny][nx][no]; p may be null
    const float *anchors;   // [na][3]
    long long io_img_rows, io_row0;
    int na, no, ny, nx;
    float stride, cf;
    int arc;
    int apb;                // anchors per channel block: block nb computes channels nb*apb*no .. + 255 and decodes anchors nb*apb .. + apb - 1
};

template <int KT, int NW, int CF, int PF, int RING, int MODE>
__global__ void __launch_bounds__(NW * 64, 2) conv_pw_kernel(const ConvParams p, const PwBnRed br, const PwDecode dc) {
    constexpr int BMS = PF * 16;                 // rows of a row block
    constexpr int UNIT = BMS * 128;              // bytes of one unit (BMS rows x 64 channels)
    constexpr int LPU = (BMS / 8) / NW;          // 1-KiB direct-to-LDS pieces a wave issues per unit
    constexpr int NCB = NW * CF * 16;            // output channels of a workgroup
    constexpr bool HAS_RES = MODE == 2 || MODE == 3, BNRED = MODE == 3, STATS = MODE == 1, DEC = MODE == 4;
    constexpr int TROW = NCB * 2 + 16;           // DEC: byte pitch of a row of the head tile (the 16-B pad spreads the rows over the banks)
    constexpr int NSTG = DEC ? 0 : (CF == 2 ? PF : PF / 2);  // 16-B output runs (= stores, accumulate loads, z loads) per lane and row block
                                                 // (DEC: the decode stage's stores are not counted: the waits then allow fewer operations in
                                                 // flight than there are, which is always safe)
    constexpr int NREQ = (HAS_RES ? NSTG : 0) + (BNRED ? NSTG : 0);   // row-block requests issued at kt == 0
    static_assert(LPU >= 1 && (BMS / 8) % NW == 0, "a unit must split into whole pieces per wave");
    static_assert(CF == 1 || CF == 2, "one or two channel fragments per wave");
    static_assert(PF % 2 == 0, "row fragments are paired by the CF == 1 epilogue");
    static_assert(RING >= 3, "at least one unit in flight behind the one being read");
    static_assert((RING - 2) * LPU + 2 * (NSTG + NREQ) <= 63, "the counted waits must be encodable");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    if (PW_DBG(6)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
#ifdef RYOLO_MP_ABLATION
    // bit 9: cycle stamps of waves 0 and NW-1 of workgroups 0, 8 and 1001 into p.trace[wg][wave sel][64] (stamp k = event k)
    unsigned *trace_row = nullptr;
    if (PW_DBG(9) && p.trace && (wave == 0 || wave == NW - 1) && (blockIdx.x == 0 || blockIdx.x == 8 || blockIdx.x == 1001 % gridDim.x))
        trace_row = p.trace + ((blockIdx.x == 0 ? 0 : (blockIdx.x == 8 ? 1 : 2)) * 2 + (wave == 0 ? 0 : 1)) * 64;
    int trace_n = 0;
#define PW_STAMP()                                                                                              \
    do {                                                                                                        \
        if (trace_row && trace_n < 64) {                                                                        \
            if (lane == 0) trace_row[trace_n] = (unsigned)__builtin_readcyclecounter();                         \
            trace_n++;                                                                                          \
        }                                                                                                       \
    } while (0)
#else
#define PW_STAMP() do { } while (0)
#endif
    PW_STAMP();                                           // 0: kernel start

    // ---- work split
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int NB = p.pw_nb;
    const int nb = loc % NB, ms = loc / NB, nms = nloc / NB;
    const int MB = p.pw_mb, q = MB >> 3, r = MB & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int len = q + (xcd < r ? 1 : 0);
    const int n0 = DEC ? nb * dc.apb * dc.no : nb * NCB;
    if constexpr (BNRED) {       // this workgroup's row of partial sums: zero everywhere but the channels it owns (written at the end)
        for (int t = tid; t < 3 * p.Cout; t += NW * 64) br.part[(size_t)blockIdx.x * 3 * p.Cout + t] = 0.f;
        __syncthreads();
    }
    if (ms >= len) return;
    const int cnt = (len - ms + nms - 1) / nms;          // row blocks of this workgroup: start + ms, + nms, ...
    const int rb0 = start + ms;

    // ---- fill side: piece j of this wave = unit rows (wave*LPU + j)*8 .. +7; lane l fills the 16-B slot (l & 7) of row (l >> 3)
    int iss_off[LPU];
#pragma unroll
    for (int j = 0; j < LPU; j++) {
        const int row = (wave * LPU + j) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        iss_off[j] = ((rb0 * BMS + row) * p.in_cs) * 2 + slot * 16;
    }
    const int iss_delta = nms * BMS * p.in_cs * 2;       // byte distance of this workgroup's consecutive row blocks
    int iss_i = 0, iss_kt = 0, iss_slot = 0;             // scalar
    auto fill = [&]() __attribute__((always_inline)) {
        const bool live = iss_i < cnt && !PW_DBG(1);
        char *dst = smem + iss_slot * UNIT + wave * LPU * 1024;
#pragma unroll
        for (int j = 0; j < LPU; j++)
            buffer_load_lds16(p.x, p.x_bytes, dst + j * 1024, live ? iss_off[j] : (int)0x80000000, iss_kt * 128);
        iss_slot = iss_slot + 1 == RING ? 0 : iss_slot + 1;
        if (++iss_kt == KT) {
            iss_kt = 0;
            iss_i++;
            if (iss_i < cnt) {
#pragma unroll
                for (int j = 0; j < LPU; j++) iss_off[j] += iss_delta;
            }
        }
    };

    // ---- read side: fragment of rows pf*16 + fr, k = ks*32 + g*8 .. +7 (the swizzle key of a row does not depend on pf)
    int a_rd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) a_rd[ks] = fr * 128 + (((ks * 4 + g) ^ ((fr >> 1) & 7)) << 4);

    // ---- per-lane epilogue constants: the lane's accumulator channels never change
    f32x4 sc[CF], sh[CF];
#pragma unroll
    for (int cf = 0; cf < CF; cf++) {
        const int ch = n0 + wave * CF * 16 + cf * 16 + g * 4;
        if constexpr (DEC) {        // a head block starts at channel nb * apb * no: 4-byte aligned only
            sc[cf] = *(const f32x4_u4 *)(p.scale + ch);
            sh[cf] = *(const f32x4_u4 *)(p.shift + ch);
        } else {
            sc[cf] = *(const f32x4 *)(p.scale + ch);
            sh[cf] = *(const f32x4 *)(p.shift + ch);
        }
    }
    const float slope = p.slope;
    // the 16-B run a lane stores: CF == 2: channels (g&1)*16 + (g>>1)*8 of the wave's 32, pixel pf*16 + fr;
    //                             CF == 1: channels (g>>1)*8 of the wave's 16, pixel (2*pp + (g&1))*16 + fr
    const int och = n0 + wave * CF * 16 + (CF == 2 ? (g & 1) * 16 : 0) + (g >> 1) * 8;       // (unused by DEC)
    const bool och_ok = och < p.Cout;
    float st_sum[CF][4], st_sq[CF][4];
#pragma unroll
    for (int cf = 0; cf < CF; cf++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) st_sum[cf][rr] = st_sq[cf][rr] = 0.f;
    float bn_sc[8], bn_sh[8], bn_mu[8], bs1[8], bs2[8], bs3[8];
    float bn_slope = 0.f;
    if constexpr (BNRED) {
        bn_slope = br.slope[0];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            bn_sc[e] = och_ok ? br.scale[och + e] : 0.f;
            bn_sh[e] = och_ok ? br.shift[och + e] : 0.f;
            bn_mu[e] = och_ok ? br.mean[och + e] : 0.f;
            bs1[e] = bs2[e] = bs3[e] = 0.f;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(HAS_RES ? p.res : p.y), 0, HAS_RES ? p.res_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc((void *)(BNRED ? (const void *)br.z : (const void *)p.y), 0, BNRED ? br.z_bytes : 0u, 0x00020000);
#endif

    // ---- prologue.  The first RING-1 activation units, then the first WD K steps of the wave's filter slice (A-operand fragments of
    // channels n0 + wave*CF*16 + cf*16 + fr, k = kt*64 + ks*32 + g*8 .. +7).  The rest of the filter is requested INSIDE the first row
    // block, K step kt + WD in iteration kt: the 128-256 KB filter read of every workgroup runs at ~55 GB/s per CU (L2-bound: all
    // CUs ask for the same lines at once; measured 8-15 k cycles just to ISSUE 32 loads per wave up front, during which the wave
    // cannot reach its first MFMA), so it streams in under the first row block's MFMAs instead of in front of them.  The first row block is a
    // separate instantiation of the loop body: there the compiler places its own counted wait in front of the first MFMA that uses
    // each fragment; in the steady-state body the filter is simply in registers.
    constexpr int WD = KT < 3 ? KT : 3;
    constexpr int WL = 2 * CF;
    bf16x8 wreg[KT][2][CF];
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.w), 0, p.w_bytes, 0x00020000);
#endif
    int woff[CF];
#pragma unroll
    for (int cf = 0; cf < CF; cf++)
        woff[cf] = PW_DBG(0) ? (int)0x80000000 : ((n0 + wave * CF * 16 + cf * 16 + fr) * p.Kpad + g * 8) * 2;
    auto load_w = [&](auto ktc) __attribute__((always_inline)) {
        constexpr int kt = decltype(ktc)::value;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int cf = 0; cf < CF; cf++)
                wreg[kt][ks][cf] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[cf], (kt * 64 + ks * 32) * 2, 0));
#endif
    };
    issue_order_fence();
#pragma unroll
    for (int t = 0; t < RING - 1; t++) fill();
    issue_order_fence();
    static_for<WD>([&](auto ktc) __attribute__((always_inline)) { load_w(ktc); });
    issue_order_fence();

    f32x4 acc[CF][PF];
    u32x4 rv[HAS_RES ? NSTG : 1], zv[BNRED ? NSTG : 1];
    int rd_slot = 0;                                     // scalar: ring slot of the unit being read
    constexpr int I0 = (RING - 1 + KT - 1) / KT;         // from row block I0 on the whole look-back window lies inside the launch

    auto row_block = [&](int i, auto FIRSTc) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(FIRSTc)::value != 0;
        const int m0 = (rb0 + i * nms) * BMS;
#pragma unroll
        for (int cf = 0; cf < CF; cf++)
#pragma unroll
            for (int f = 0; f < PF; f++) acc[cf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool steady = !FIRST && i >= I0;
        static_for<KT>([&](auto ktc) __attribute__((always_inline)) {
            constexpr int kt = decltype(ktc)::value;
            // operations younger than unit u's fill at this point (see the file header)
            constexpr int BASE = (RING - 2) * LPU;
            constexpr int own_rq = (kt >= 1 && kt <= RING - 2) ? NREQ : 0;   // this row block's own requests (its kt = 0 iteration)
            constexpr int ALL_ST = []() { int n = 0; for (int d = 1; d <= RING - 1; d++) if ((((kt - d) % KT) + KT) % KT == KT - 1) n++; return n; }();
            constexpr int ALL_RQ = []() { int n = 0; for (int d = 1; d <= RING - 2; d++) if ((((kt - d) % KT) + KT) % KT == 0) n++; return n; }();
            if constexpr (FIRST) wait_vmcnt<pw_first_block_younger(kt, KT, RING, LPU, WL, NREQ, WD)>();
            else if (steady) wait_vmcnt<BASE + ALL_ST * NSTG + ALL_RQ * NREQ>();
            else wait_vmcnt<BASE + own_rq>();
            PW_STAMP();                                  // 2 + 3u: unit u's own pieces landed
            if (!PW_DBG(4)) __builtin_amdgcn_s_barrier();                // every wave's pieces of unit u have landed; unit u-1's slot is free
            PW_STAMP();                                  // 3 + 3u: through the barrier
            if constexpr (kt == 0 && NREQ > 0) {
                // the accumulate operand / the consumer block's z for this row block's epilogue: requested now, behind units that
                // are needed before the epilogue anyway, so that waiting for them later drains nothing
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int s = 0; s < NSTG; s++) {
                    const int m = m0 + (CF == 2 ? s : 2 * s + (g & 1)) * 16 + fr;
                    const bool ok = m < p.M && och_ok;
                    if constexpr (HAS_RES) rv[s] = __builtin_amdgcn_raw_buffer_load_b128(rrs, ok ? (m * p.res_cs + och) * 2 : (int)0x80000000, 0, 0);
                    if constexpr (BNRED) zv[s] = __builtin_amdgcn_raw_buffer_load_b128(zrs, ok ? (m * br.z_cs + och) * 2 : (int)0x80000000, 0, 0);
                }
#endif
            }
            issue_order_fence();
            fill();                                      // unit u + RING - 1 into the slot of unit u - 1
            issue_order_fence();
            if constexpr (FIRST && kt + WD < KT) {
                load_w(ic<kt + WD>{});
                issue_order_fence();
            }
            const char *ub = smem + rd_slot * UNIT;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                bf16x8 xf[PF];
#pragma unroll
                for (int f = 0; f < PF; f++) xf[f] = PW_DBG(5) ? wreg[kt][ks][0] : *(const bf16x8 *)(ub + a_rd[ks] + f * 2048);
                if (!PW_DBG(2)) {
#pragma unroll
                    for (int cf = 0; cf < CF; cf++)
#pragma unroll
                        for (int f = 0; f < PF; f++)
                            acc[cf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[kt][ks][cf], xf[f], acc[cf][f], 0, 0, 0);
                }
            }
            rd_slot = rd_slot + 1 == RING ? 0 : rd_slot + 1;
#ifdef RYOLO_MP_ABLATION
            if (trace_row) asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[CF - 1][PF - 1]));   // the stamp waits for the unit's last MFMAs
#endif
            PW_STAMP();                                  // 4 + 3u: unit u multiplied
        });

        // ------------------------------------------------------------------ epilogue: registers -> global
        auto run_epilogue = [&](auto ACTc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(ACTc)::value;
            if constexpr (DEC) {
                // head values -> bf16 -> LDS tile (8-B writes: lane = 4 consecutive channels of one row), then the decode by all waves
                char *tile = smem + RING * UNIT;
#pragma unroll
                for (int cf = 0; cf < CF; cf++)
#pragma unroll
                    for (int f = 0; f < PF; f++) {
                        bf16x4 o;
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            float v = acc[cf][f][rr] * sc[cf][rr] + sh[cf][rr];
                            if constexpr (ACT == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                            else if constexpr (ACT == RYOLO_ACT_MISH) v = mish(v);
                            o[rr] = (__bf16)v;
                        }
                        *(bf16x4 *)(tile + (f * 16 + fr) * TROW + (wave * CF * 16 + cf * 16 + g * 4) * 2) = o;
                    }
                __syncthreads();      // (drains this wave's look-ahead fills too: once per row block, behind 4-16 K steps of MFMAs)
                const int a0 = nb * dc.apb, na_here = min(dc.apb, dc.na - a0);
                const int nrows = BMS * na_here;
                const long long hw = (long long)dc.ny * dc.nx;
                for (int rw = tid; rw < nrows; rw += NW * 64) {
                    const int al = rw / BMS, pxl = rw - al * BMS;           // rows of one anchor over consecutive pixels: adjacent in io / p
                    const int a = a0 + al;
                    const int m = m0 + pxl;
                    if (m >= p.M) continue;
                    const int t1 = udiv_magic(m, p.magic_wo);
                    const int x = m - t1 * p.Wo;
                    const int n = udiv_magic(t1, p.magic_ho);
                    const int y = t1 - n * p.Ho;
                    const __bf16 *src = (const __bf16 *)(tile + pxl * TROW) + al * dc.no;
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (k < dc.no) v[k] = (float)src[k];
                    const long long row = (long long)a * hw + (long long)y * dc.nx + x;
                    if (dc.p) store_row(dc.p + (((long long)n * dc.na * hw) + row) * dc.no, v, dc.no);
                    const float aw = dc.anchors[a * 3 + 0] / dc.stride, ah = dc.anchors[a * 3 + 1] / dc.stride, aa = dc.anchors[a * 3 + 2];
                    decode_row<8>(v, dc.no, x, y, aw, ah, aa, dc.stride, dc.cf, dc.arc, dc.io + (((long long)n * dc.io_img_rows) + dc.io_row0 + row) * dc.no);
                }
                return;
            }
            unsigned R[CF][PF][2];
#pragma unroll
            for (int cf = 0; cf < CF; cf++)
#pragma unroll
                for (int f = 0; f < PF; f++) {
                    const bool ok = m0 + f * 16 + fr < p.M;
                    bf16x4 o;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        float v = acc[cf][f][rr] * sc[cf][rr] + sh[cf][rr];
                        if constexpr (ACT == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                        else if constexpr (ACT == RYOLO_ACT_MISH) v = mish(v);
                        o[rr] = (__bf16)v;
                        if constexpr (STATS) {           // statistics of the values as stored (bf16)
                            const float qv = ok ? (float)o[rr] : 0.f;
                            st_sum[cf][rr] += qv;
                            st_sq[cf][rr] += qv * qv;
                        }
                    }
                    const uint2 u2 = __builtin_bit_cast(uint2, o);
                    R[cf][f][0] = u2.x;
                    R[cf][f][1] = u2.y;
                }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int s = 0; s < (DEC ? 0 : NSTG); s++) {
                // the two 8-B halves of this lane's 16-B run: CF == 2: channel fragments 0 / 1 of row fragment s; CF == 1: row
                // fragments 2s / 2s+1.  The odd 16-lane rows of the first trade places with the even rows of the second.
                unsigned a0, a1, b0, b1;
                if constexpr (CF >= 2) { a0 = R[0][s][0]; a1 = R[0][s][1]; b0 = R[1][s][0]; b1 = R[1][s][1]; }
                else { a0 = R[0][2 * s][0]; a1 = R[0][2 * s][1]; b0 = R[0][2 * s + 1][0]; b1 = R[0][2 * s + 1][1]; }
                auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                u32x4 out = u32x4{s0[0], s1[0], s0[1], s1[1]};
                const int m = m0 + (CF == 2 ? s : 2 * s + (g & 1)) * 16 + fr;
                const bool ok = m < p.M && och_ok;
                if constexpr (HAS_RES) {
                    bf16x8 a = __builtin_bit_cast(bf16x8, out);
                    const bf16x8 b = __builtin_bit_cast(bf16x8, rv[s]);
#pragma unroll
                    for (int e = 0; e < 8; e++) a[e] = (__bf16)((float)a[e] + (float)b[e]);
                    out = __builtin_bit_cast(u32x4, a);
                }
                if constexpr (BNRED) {       // the arithmetic of bn_act_bwd_reduce_kernel<1> on the value as it is stored (bf16)
                    const bf16x8 dv = __builtin_bit_cast(bf16x8, out), zq = __builtin_bit_cast(bf16x8, zv[s]);
                    if (ok) {
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const float zf = (float)zq[e], d = (float)dv[e];
                            const float u = zf * bn_sc[e] + bn_sh[e];
                            float gg = d;
                            if (u <= 0.f) { gg = d * bn_slope; bs3[e] += d * u; }
                            bs2[e] += gg * (zf - bn_mu[e]);
                            bs1[e] += gg;
                        }
                    }
                }
                if (PW_DBG(3)) {
                    asm volatile("" ::"v"(out));
                } else if (p.ups == 1) {
                    const int voff = ok ? (m * p.out_cs + och) * 2 : (int)0x80000000;
                    if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(out, yrs, voff, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b128(out, yrs, voff, 0, 0);
                } else {                     // fused nearest 2x upsample: the pixel's 2x2 block of the [N, 2Ho, 2Wo] output
                    const int mm = m < p.M ? m : 0;
                    const int t = udiv_magic(mm, p.magic_wo);
                    const int wo = mm - t * p.Wo;
                    const int img = udiv_magic(t, p.magic_ho);
                    const int ho = t - img * p.Ho;
                    const int W2 = p.Wo * 2;
                    const int o00 = ok ? ((((img * p.Ho * 2 + ho * 2) * W2 + wo * 2) * p.out_cs) + och) * 2 : (int)0x80000000;
                    buffer_store16_soff<0>(out, yrs, o00, 0);
                    buffer_store16_soff<0>(out, yrs, o00, p.out_cs * 2);
                    buffer_store16_soff<0>(out, yrs, o00, W2 * p.out_cs * 2);
                    buffer_store16_soff<0>(out, yrs, o00, (W2 + 1) * p.out_cs * 2);
                }
            }
#endif
        };
        if (p.act == RYOLO_ACT_LEAKY) run_epilogue(ic<RYOLO_ACT_LEAKY>{});
        else if (p.act == RYOLO_ACT_MISH) run_epilogue(ic<RYOLO_ACT_MISH>{});
        else run_epilogue(ic<RYOLO_ACT_LINEAR>{});
        PW_STAMP();                                      // 2 + 3 KT (+ 1 per earlier row block): epilogue issued
    };
    PW_STAMP();                                           // 1: prologue issued
    if (PW_DBG(7)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    row_block(0, ic<1>{});
    if (!PW_DBG(8))
        for (int i = 1; i < cnt; i++) row_block(i, ic<0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the look-ahead fills behind the last row block
    PW_STAMP();                                           // last: everything drained

    if constexpr (STATS) {
        // the 16 lanes of a row hold the same channels: DPP row sum, lane fr keeps total number fr (fragment fr / 4, register fr % 4);
        // every channel belongs to exactly one wave of the workgroup, so one 64-bit atomic per channel and statistic
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int cf = 0; cf < CF; cf++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const float a = row16_sum(st_sum[cf][rr]), b = row16_sum(st_sq[cf][rr]);
                if (fr == cf * 4 + rr) {
                    ta = a;
                    tb = b;
                }
            }
        if (fr < CF * 4) {
            const int ch = n0 + wave * CF * 16 + (fr >> 2) * 16 + g * 4 + (fr & 3);
            if (ch < p.Cout) {
                double *row = p.stat_part + (size_t)(blockIdx.x % STAT_ROWS) * 2 * p.stat_cpad;
                atomicAdd(row + ch, (double)ta);
                atomicAdd(row + p.stat_cpad + ch, (double)tb);
            }
        }
    }
    if constexpr (BNRED) {
        // a lane's 8 channels are shared by the 16 lanes of its row (CF == 2) or by the two rows g, g^1 as well (CF == 1):
        // DPP row sums, then (CF == 1) one add across the row pair -- fixed order, and each channel is written by one lane
        float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float a = row16_sum(bs1[e]), b = row16_sum(bs2[e]), d = row16_sum(bs3[e]);
            if constexpr (CF == 1) {
                a += __shfl_xor(a, 16); b += __shfl_xor(b, 16); d += __shfl_xor(d, 16);
            }
            if (fr == e) { t1 = a; t2 = b; t3 = d; }
        }
        const bool writer = fr < 8 && (CF == 2 || (g & 1) == 0);
        const int ch = och + fr;
        if (writer && ch < p.Cout) {
            float *row = br.part + (size_t)blockIdx.x * 3 * p.Cout;
            row[ch] = t1;
            row[p.Cout + ch] = t2 * br.invstd[ch];
            row[2 * p.Cout + ch] = t3;
        }
    }
}

// ---- the shapes served: (K / 64, channels per workgroup) -> (waves, channel fragments per wave, row fragments, ring depth)
struct PwCfg { int kt, ncb, nw, cf, pf, ring; };

static thread_local const PwDecode *g_pw_decode = nullptr;      // set around the dispatch by launch_conv_pw_decode

template <int KT, int NW, int CF, int PF, int RING, int MODE>
int pw_launch(ConvParams &p, const PwBnRed &br, int grid, hipStream_t stream) {
    constexpr int LDS = RING * PF * 16 * 128 + (MODE == 4 ? PF * 16 * (NW * CF * 16 * 2 + 16) : 0);
    static bool attr_done = false;
    auto kfn = conv_pw_kernel<KT, NW, CF, PF, RING, MODE>;
    if (!attr_done) {
        if (LDS > 64 * 1024 && hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return RYOLO_ELAUNCH;
        attr_done = true;
    }
    const PwDecode dc = g_pw_decode ? *g_pw_decode : PwDecode{};
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NW * 64), LDS, stream, p, br, dc);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

// MODES: bit m set = MODE m is instantiated for this configuration (each instantiation is ~200 registers of unrolled code)
template <int KT, int NW, int CF, int PF, int RING, int MODES>
int pw_launch_mode(ConvParams &p, const PwBnRed *br, int grid, hipStream_t stream) {
    PwBnRed none{};
    const int mode = g_pw_decode ? 4 : (br ? 3 : (p.stat_part ? 1 : (p.res ? 2 : 0)));
    if constexpr ((MODES & 16) != 0) if (mode == 4) return pw_launch<KT, NW, CF, PF, RING, 4>(p, none, grid, stream);
    if constexpr ((MODES & 8) != 0) if (mode == 3) return pw_launch<KT, NW, CF, PF, RING, 3>(p, *br, grid, stream);
    if constexpr ((MODES & 2) != 0) if (mode == 1) return pw_launch<KT, NW, CF, PF, RING, 1>(p, none, grid, stream);
    if constexpr ((MODES & 4) != 0) if (mode == 2) return pw_launch<KT, NW, CF, PF, RING, 2>(p, none, grid, stream);
    if constexpr ((MODES & 1) != 0) if (mode == 0) return pw_launch<KT, NW, CF, PF, RING, 0>(p, none, grid, stream);
    return RYOLO_EINVAL;
}

inline unsigned pw_magic_u32(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

inline int pw_cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    return cus;
}

// which configuration serves (K, C_out, M); returns false when none does.  NB = channel blocks, wgpc = workgroups per CU.
bool pw_pick(const ConvParams &p, bool bnred, PwCfg &c, int &NB, int &grid) {
    if (p.Kpad != p.Cin || (p.Cin & 63)) return false;
    const int kt = p.Cin / 64;
    const int cus = pw_cu_count() & ~7;
    if (cus < 8) return false;
    int wgpc;
    // a YOLO head decoded from the accumulators: all (<= 512) channels in one workgroup, 8 waves x 64
    if (g_pw_decode) {
        if (bnred) return false;
        if (kt == 4) c = PwCfg{4, 256, 8, 2, 4, 12};
        else if (kt == 8) c = PwCfg{8, 256, 8, 2, 4, 12};
        else if (kt == 16) c = PwCfg{16, 128, 8, 1, 4, 12};
        else return false;
        NB = (g_pw_decode->na + g_pw_decode->apb - 1) / g_pw_decode->apb;
        grid = cus;
        return NB >= 1 && (grid / 8) % NB == 0;
    }
    // K 256 / 384 -> 128 channels per workgroup (4 waves x 32), two workgroups per CU
    else if (kt == 4 && p.Cout <= 128 && !bnred) { c = PwCfg{4, 128, 4, 2, 8, 4}; wgpc = 2; }
    else if (kt == 6 && p.Cout <= 128 && !bnred) { c = PwCfg{6, 128, 4, 2, 4, 8}; wgpc = 2; }
    // K 128 / 256 / 512 -> 256 channels per workgroup (8 waves x 32); with the folded BatchNorm reduce (48 more registers of per-channel
    // constants and sums, 8 of z) the K 256 / 512 filter slices of a 32-channel wave no longer fit: 16 channels per wave, twice the blocks
    else if (kt == 2 && p.Cout % 256 == 0) { c = PwCfg{2, 256, 8, 2, 4, 12}; wgpc = 1; }
    else if (kt == 4 && bnred) { c = PwCfg{4, 128, 8, 1, 4, 12}; wgpc = 1; }
    else if (kt == 4) { c = PwCfg{4, 256, 8, 2, 4, 12}; wgpc = 1; }
    else if (kt == 8 && bnred) { c = PwCfg{8, 128, 8, 1, 4, 12}; wgpc = 1; }
    else if (kt == 8) { c = PwCfg{8, 256, 8, 2, 4, 12}; wgpc = 1; }
    // K 768 / 1024 -> 128 channels per workgroup (8 waves x 16): the filter slice is 96 / 128 registers
    else if (kt == 12 && !bnred) { c = PwCfg{12, 128, 8, 1, 4, 12}; wgpc = 1; }
    else if (kt == 16 && !bnred) { c = PwCfg{16, 128, 8, 1, 4, 12}; wgpc = 1; }
    else return false;
    NB = (p.Cout + c.ncb - 1) / c.ncb;
    grid = wgpc * cus;
    if (p.pw_grid_cap > 0 && p.pw_grid_cap * 8 < grid) grid = p.pw_grid_cap * 8;
    const int nloc = grid / 8;
    if (NB < 1 || nloc % NB) return false;               // channel blocks must tile the XCD's workgroups
    if (NB * c.ncb > ((p.Cout + 127) / 128) * 128) return false;   // the packed filter has ceil128(C_out) rows
    return true;
}

}  // namespace

namespace ryolo_detail {

bool conv_pw_eligible(const ConvParams &p, int ksize) {
    if (ksize != 1 || !p.fast || p.os != 1 || p.stride != 1 || p.taps2) return false;
    if (p.stat_part && (p.res || p.ups != 1)) return false;
    if (p.res && p.ups != 1) return false;
    PwCfg c;
    int NB, grid;
    if (!pw_pick(p, false, c, NB, grid)) return false;
    return !(p.res && c.kt == 4 && c.nw == 4);      // (that configuration has no accumulate-operand instantiation)
}

// Where the kernel is the automatic choice.  Measured inside the bs-32 forward against the 128 x 128 tile (profiles/r04_pw_vs_igemm.txt):
// 1.06-1.11 x on the 76^2 layers, 1.03-1.12 x at 38^2, 1.26 x on 768->256 -- and 0.94-0.98 x at 19^2 (K 1024: a workgroup owns
// less than three row blocks and spends a third of its life fetching its 256 KB of filter), 0.73-0.83 x on the two small
// upsampling layers (less than one row block per workgroup), 0.93 x on the output-bound 256->504 head.  Hence: at least two row
// blocks per workgroup, K <= 768, and not more than 1.5 x as many output as input channels.
bool conv_pw_preferred(const ConvParams &p) {
    PwCfg c;
    int NB, grid;
    if (!pw_pick(p, false, c, NB, grid)) return false;
    const long long mb = ((long long)p.M + c.pf * 16 - 1) / (c.pf * 16);
    return c.kt <= 12 && 2 * p.Cout <= 3 * p.Cin && mb * NB >= 2ll * grid;
}

// rows of BatchNorm-reduce partials (= workgroups) a conv_pw launch of this shape writes; 0 = not served
int conv_pw_grid(const ConvParams &p) {
    PwCfg c;
    int NB, grid;
    return pw_pick(p, true, c, NB, grid) ? grid : 0;
}

// anchors per channel block of a decoded head: the most whole anchors that fit in the block's `ncb` channels
static int pw_decode_apb(int na, int no, int ncb = 256) {
    const int apb = ncb / no;
    return apb < na ? apb : na;
}

// the head conv + decode as one launch (conv.hip: ryolo_conv_head_decode); EINVAL when the shape is not served
int launch_conv_pw_decode(ConvParams &p, float *io, long long io_img_rows, long long io_row0, float *pout, const float *anchors, int na, int no,
                          float stride, float cf, int arc, hipStream_t stream) {
    if (!io || !anchors || na <= 0 || no < 7 || no > 8 || na * no != p.Cout || p.ups != 1 || p.res || p.stat_part) return RYOLO_EINVAL;
    const long long dmax = p.Wo > p.Ho ? p.Wo : p.Ho;
    if (((long long)p.M + 64) * dmax >= 0x100000000ll) return RYOLO_EINVAL;       // the multiply-high pixel decomposition must be exact
    PwDecode dc;
    dc.io = io; dc.p = pout; dc.anchors = anchors; dc.io_img_rows = io_img_rows; dc.io_row0 = io_row0;
    dc.na = na; dc.no = no; dc.ny = p.Ho; dc.nx = p.Wo; dc.stride = stride; dc.cf = cf; dc.arc = arc;
    dc.apb = pw_decode_apb(na, no, p.Cin == 1024 ? 128 : 256);
    g_pw_decode = &dc;
    const int rc = launch_conv_pw(p, nullptr, stream);
    g_pw_decode = nullptr;
    return rc;
}

bool conv_pw_decode_supported(const ConvParams &p, int na, int no) {
    if (no < 7 || no > 8 || na * no != p.Cout || p.Kpad != p.Cin || (p.Cin != 256 && p.Cin != 512 && p.Cin != 1024)) return false;
    const int ncb = p.Cin == 1024 ? 128 : 256;
    const int apb = pw_decode_apb(na, no, ncb);
    const int nb = apb > 0 ? (na + apb - 1) / apb : 0;
    // the last block's channels must stay inside the packed filter's ceil128(C_out) rows; blocks must tile an XCD's workgroups
    if (!(apb > 0 && (nb == 1 || nb == 2 || nb == 4) && (nb - 1) * apb * no + ncb <= ((p.Cout + 127) / 128) * 128)) return false;
    // ... and the launch's own configuration test must accept the shape on THIS device (on a partition whose workgroups per
    // XCD are not a multiple of the channel blocks the engine planned a fused head and every forward failed): the same pw_pick() call
    // launch_conv_pw_decode makes, on a temporary decode record
    PwDecode dc{};
    dc.na = na; dc.no = no; dc.apb = apb;
    const PwDecode *saved = g_pw_decode;
    g_pw_decode = &dc;
    PwCfg c;
    int NB, grid;
    const bool ok = pw_pick(p, false, c, NB, grid);
    g_pw_decode = saved;
    return ok;
}

#ifdef RYOLO_MP_ABLATION
static int g_pw_dbg = 0;
static unsigned *g_pw_trace = nullptr;
#endif

int launch_conv_pw(ConvParams &p, const void *bnred /* conv.hip BnRed or nullptr */, hipStream_t stream) {
#ifdef RYOLO_MP_ABLATION
    p.dbg0 = g_pw_dbg;
    p.trace = g_pw_trace;
#endif
    PwCfg c;
    int NB, grid;
    if (!pw_pick(p, bnred != nullptr, c, NB, grid)) return RYOLO_EINVAL;
    const int bms = c.pf * 16;
    const long long mb = ((long long)p.M + bms - 1) / bms;
    const unsigned long long yb = (((unsigned long long)p.M * p.ups * p.ups - 1) * p.out_cs + p.Cout) * 2ull;
    const unsigned long long rb = p.res ? (((unsigned long long)p.M - 1) * p.res_cs + p.Cout) * 2ull : 0ull;
    // 32-bit byte offsets everywhere, including the look-ahead of one workgroup stride past the last row block
    const unsigned long long xmax = ((unsigned long long)(mb + grid) * bms) * p.in_cs * 2ull + 4096;
    if (yb >= 0x7fffff00ull || rb >= 0x7fffff00ull || xmax >= 0x7fffff00ull || mb > 0x7fffffffll) return RYOLO_EINVAL;
    if (p.ups != 1 && (long long)mb * bms * (p.Wo > p.Ho ? p.Wo : p.Ho) >= 0x100000000ll) return RYOLO_EINVAL;
    p.y_bytes = (unsigned)yb;
    p.res_bytes = (unsigned)rb;
    p.pw_nb = NB;
    p.pw_mb = (int)mb;
    p.magic_wo = pw_magic_u32(p.Wo);
    p.magic_ho = pw_magic_u32(p.Ho);
    PwBnRed br{};
    const PwBnRed *brp = nullptr;
    if (bnred) {
        if (p.stat_part || p.ups != 1) return RYOLO_EINVAL;
        const BnRed *b = (const BnRed *)bnred;
        br.z = b->z; br.z_cs = b->z_cs; br.scale = b->scale; br.shift = b->shift; br.mean = b->mean; br.invstd = b->invstd;
        br.slope = b->slope; br.part = b->part;
        const unsigned long long zb = (((unsigned long long)p.M - 1) * b->z_cs + p.Cout) * 2ull;
        if (zb >= 0x7fffff00ull) return RYOLO_EINVAL;
        br.z_bytes = (unsigned)zb;
        brp = &br;
    }
    RYOLO_CONV_DRY_RUN(RYOLO_CONV_KERNEL_PW);       // (a mode without an instantiation returns EINVAL below: the forward modes all exist)
#define PW_CASE(KT_, NW_, CF_, PF_, RING_, MODES_)                                               \
    if (c.kt == KT_ && c.nw == NW_ && c.cf == CF_ && c.pf == PF_ && c.ring == RING_)              \
        return pw_launch_mode<KT_, NW_, CF_, PF_, RING_, MODES_>(p, brp, grid, stream);
    PW_CASE(4, 4, 2, 8, 4, 3)      // (no accumulate-operand instantiation: it needs 12 spilled registers inside the loop; such launches take the 128x128 tile)
    PW_CASE(6, 4, 2, 4, 8, 7)
    PW_CASE(2, 8, 2, 4, 12, 15)
    PW_CASE(4, 8, 2, 4, 12, 23)
    PW_CASE(4, 8, 1, 4, 12, 8)
    PW_CASE(8, 8, 2, 4, 12, 23)
    PW_CASE(8, 8, 1, 4, 12, 8)
    PW_CASE(12, 8, 1, 4, 12, 7)
    PW_CASE(16, 8, 1, 4, 12, 23)
#undef PW_CASE
    return RYOLO_EINVAL;
}

}  // namespace ryolo_detail

#ifdef RYOLO_MP_ABLATION
extern "C" void ryolo_debug_convpw_set(int flags) { ryolo_detail::g_pw_dbg = flags; }
extern "C" void ryolo_debug_convpw_trace(void *buf /* uint32[3][2][64], zeroed */) { ryolo_detail::g_pw_trace = (unsigned *)buf; }
#endif
