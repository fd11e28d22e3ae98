// rotate-yolov3_amd/csrc/conv_mq.hip -- the implicit-GEMM convolution tile for gfx950 with TWO INDEPENDENT 4-wave workgroups
// per CU: 128 pixels x 256 output channels per workgroup, wave tile 128 x 64 (the same MFMA / LDS-read economy as conv_mp.hip),
// persistent grid of 2 x CUs workgroups.
//
// Same operator and operand layout as conv_mp.hip (model/models.py:49-66 conv -> BN -> PReLU / Mish, :281-282 shortcut).
// Why a second structure (measured on conv_mp.hip, tools/mp_ablate.py, profiles/r03_mp_ablation.txt): with ONE 8-wave workgroup
// per CU all waves leave the K loop together, and everything between two K loops runs with the matrix pipes idle -- per
// 256 x 256 x (9*128) tile 54.5 k cycles of K loop, then 4.4 k of next-tile bookkeeping, 2.3 k of shortcut requests, 7-8 k of
// epilogue passes, and the stores / shortcut loads slow the next K loop's first tiles by another 9 k: 83 k in all.  Here a
// workgroup owns only one wave per SIMD; the other wave of the SIMD belongs to the other workgroup of the CU, which is half
// a tile period away in its own schedule (second-half workgroups start with a half-height tile), so one workgroup's
// bookkeeping / epilogue / store drain runs under the other's MFMAs.
//   * LDS per workgroup: activations double-buffered 2 x 16 KiB, weights SINGLE-staged 32 KiB and refilled right behind
//     their reads, scale/shift 4 KiB: 68 KiB, two workgroups per CU;
//   * waves are 1 x 4 over the channels: a wave's 64 weight rows are PRIVATE to it (it stages them and it alone reads them),
//     only the activation stage is shared -- so a K tile (64 deep, four phases of 16 MFMAs per wave) needs ONE s_barrier:
//       phase 0: read WA (channel fragments 0,1) + XA (pixel fragments 0..3)   wait vmcnt(2)             issue XB(t+1)   MFMA
//       phase 1: read WB (channel fragments 2,3)                                                         issue WA(t+1)   MFMA
//       phase 2: read XB (pixel fragments 4..7)                                                          issue WB(t+1)   MFMA
//       phase 3:                                                                 wait vmcnt(4)   barrier   issue XA(t+2)   MFMA
//     (XA / XB = the two 8-KiB halves of an activation stage, 2 direct-to-LDS pieces per wave; WA / WB = the two halves of
//     the wave's own 8 KiB of weight rows, 4 pieces each).
// Hazards.  Weights (wave-private): a half is refilled after the MFMAs that consumed its fragments were issued (program
// order: the reads are complete), and read again after the wave's own counted wait -- phase 3's vmcnt(4) leaves only WB(t+1)
// in flight (WA(t+1) has landed for phase 0), phase 0's vmcnt(2) leaves only XA(t+2) (WB(t+1) has landed for phase 1).
// Activations (shared): RAW -- every wave's phase-3 wait retires its XA(t+1) (issued a K tile earlier) and XB(t+1) (issued in
// phase 0) pieces before the barrier, and they are read after it; WAR -- XA(t+2) overwrites the stage whose XA half every
// wave read in phase 0 of this K tile (complete: its MFMAs were issued before the wave reached the barrier), XB(t+1)
// overwrites the half read in phase 2 of K tile t-1, with the barrier of K tile t-1 in between.
// The chunk stream never stops at an output-tile boundary (the look-ahead runs into the next tile), and the epilogue never
// touches LDS.  FAST path only (C_in % 64 == 0, K >= 128, tensors < 2 GiB), C_out % 256 == 0.
//
// Round 5: the kernel is a FAMILY <CW, PF> -- CW = output channels per wave (64: the 256-channel tile above; 32: a 128-CHANNEL
// tile for the layers and data gradients with C_out % 256 != 0, which ran round 1's 128 x 128 tile), PF = 16-pixel fragments
// per wave (8: 128-pixel tiles; 4: 64-pixel tiles for the 19^2 / 38^2 1x1 layers, whose 128-pixel tile lists are 1.4 rounds
// deep).  Per K tile a wave then stages CW/16 weight pieces in two halves and PF/4 activation pieces per half, and the counted
// waits follow: phase 0 leaves the PF/4 pieces of XA(t+2) in flight, phase 3 the CW/16 pieces of WB(t+1).  The 128-channel
// tile holds 48 KiB of operands (2 x 16 + 16), its accumulators 64 registers -- which leaves room for the FOLDED BATCHNORM
// REDUCE (BNRED, stride-1 data gradients that store the final gradient of a BatchNorm block's output: conv.hip bnreduce_plan
// mode 4): the epilogue reads the block's z next to the running gradient, forms g = dy * act'(z * scale + shift) on the values
// it stores and adds the 16-lane row sums of g, g * (z - mean), dy * min(u, 0) to wave-private LDS accumulators per channel
// tile; every workgroup stores (not adds) its row of part[grid][3][C] at the end, in fixed order.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

using namespace ryolo_detail;

namespace {

#ifdef RYOLO_MP_ABLATION
constexpr int Q_TRACE = 4 * 128 * 4;
#else
constexpr int Q_TRACE = 0;
#endif
// LDS map of an instantiation: [X0][X1][W][scale/shift x 2 slots][trace][statistics | BatchNorm-reduce accumulators]
template <int CW, int PF>
struct QL {
    static constexpr int BN = 4 * CW;              // output channels per tile (256 / 128)
    static constexpr int BM = 16 * PF;             // pixels per tile (128 / 64)
    static constexpr int XB = BM * 128;            // one activation stage (BM rows x 128 B)
    static constexpr int WB = BN * 128;            // the weight stage
    static constexpr int XBASE = 0, WBASE = 2 * XB;
    static constexpr int OPS = 2 * XB + WB;        // 64 KiB (256 x 128), 48 KiB (128 x 128), 32 KiB (128 x 64)
    static constexpr int SS = 2 * 2 * BN * 4;      // two slots of {scale[BN], shift[BN]}
    static constexpr int LDS = OPS + SS + Q_TRACE;
    static constexpr int STAT_NT = CW == 64 ? 4 : 8;
    static constexpr int STAT = STAT_NT * 2 * BN * 4;   // [channel tile][sum | sum of squares][BN] fp32
    static constexpr int RED = STAT_NT * 3 * BN * 4;    // BNRED: [channel tile][3 sums][BN] fp32
    static constexpr int LDS_GEN = LDS + STAT;
    static constexpr int LDS_RED = LDS + RED;
};

template <int N> using ic = std::integral_constant<int, N>;

// GEN as in conv_mp.hip: 0 inference, 1 training forward (statistics, no residual), 2 data gradient (strided placement /
// accumulation through the residual operand).  VAR (ablation builds only): 8 no stores, 16 no epilogue, 32 second-half
// workgroups start p.dbg0 cycles late, 1024 stamps around the K loop / epilogue.
// CW / PF: the tile family (file header); BNRED: GEN 0 only, the folded BatchNorm reduce (br = the consumer block's z and constants)
// FS: epilogue sweep order, 1 = one sweep per 64-B half over all pixel groups (rounds 3-4), 2 = two blocks of pixel groups (below)
template <int GEN, int VAR, int CW = 64, int PF = 8, bool BNRED = false, int FS = 1, int KO = 0>
__global__ void __launch_bounds__(256, 2) conv_mq_kernel(const ConvParams p, const BnRed br) {
    using L = QL<CW, PF>;
    constexpr int PQ = PF / 2;                      // pixel fragments per activation half
    constexpr int NC = CW / 32;                     // channel fragments per weight half (wlo / whi)
    constexpr int PPC = PF / 4;                     // activation pieces per half and wave (8 rows x 128 B each)
    constexpr int WPH = CW / 16;                    // weight pieces per half and wave
    constexpr int Q_BN = L::BN, Q_XB = L::XB, Q_XBASE = L::XBASE, Q_WBASE = L::WBASE, Q_OPS = L::OPS, Q_SS = L::SS, Q_LDS = L::LDS;
    constexpr int Q_STAT_NT = L::STAT_NT, Q_STAT = L::STAT, Q_RED = L::RED;
    static_assert(!BNRED || (GEN == 0 && VAR == 0), "the folded reduce rides in the stride-1 data gradient");
    static_assert((CW == 64 || CW == 32) && (PF == 8 || PF == 4), "tile family");
    constexpr bool NO_STORE = (VAR & 8) != 0, NO_EPI = (VAR & 16) != 0, SKEW = (VAR & 32) != 0, TRACE_EPI = (VAR & 1024) != 0;
    constexpr int NST = (GEN == 2 || NO_STORE || NO_EPI) ? 0 : NC * PF;   // buffer stores per wave per output tile (exact)
    constexpr bool PRIO_HALF = (VAR & 2) != 0;      // ablation: second-half workgroups run at s_setprio 1
    constexpr bool PRIO_TOGGLE = (VAR & 4) != 0;    // ablation: priority alternates per K tile, opposite in the two halves
    constexpr bool PRIO_BURST = (VAR & 256) != 0;   // ablation: s_setprio 1 around every 16-MFMA burst

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [X0][X1][W]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- tile list: XCD x (= blockIdx & 7) owns the x-th contiguous chunk of tile ids (channel tile fastest)
    const int T = p.ntiles, G = gridDim.x;
    const int tq = T >> 3, tr = T & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = G >> 3;
    const int tstart = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int tlen = tq + (xcd < tr ? 1 : 0);
    if (loc >= tlen) {
        if constexpr (BNRED) {      // rows are stored, not added: a workgroup without tiles owns a row of zeros
            for (int i = tid; i < 3 * p.Cout; i += 256) br.part[(size_t)blockIdx.x * 3 * p.Cout + i] = 0.f;
        }
        return;
    }
    if constexpr (SKEW) {
        if (2 * loc >= nloc) {
            const long long t_end = (long long)__builtin_readcyclecounter() + (long long)p.dbg0;
            while ((long long)__builtin_readcyclecounter() < t_end) __builtin_amdgcn_s_sleep(8);
        }
    }
    if constexpr (PRIO_HALF) {
        if (2 * loc >= nloc) __builtin_amdgcn_s_setprio(1);
    }
    float *stat_lds = (float *)(smem + Q_LDS);
    const bool stat_in_lds = GEN == 1 && p.stat_part != nullptr && p.nt <= Q_STAT_NT;
    if constexpr (GEN == 1) {
        if (stat_in_lds)
            for (int i = tid; i < Q_STAT / 4; i += 256) stat_lds[i] = 0.f;     // visible after the prologue's barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (BNRED) {          // [channel tile][3][BN] running sums of this workgroup (launcher: nt <= STAT_NT); wave-private columns
        for (int i = tid; i < Q_RED / 4; i += 256) stat_lds[i] = 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- staging bookkeeping.  Activation piece i = 2*chunk + k: tile rows (chunk*8 + 2*wn + k)*8 .. +7, lane l fills the
    // 16-B slot (l & 7) of row (l >> 3).  Weight pieces: quarter q of half c = rows q*64 + c*32 + wn*8 .. +7 -- the swizzle
    // key of a row does not depend on (q, c), so ONE per-lane offset serves all eight pieces (the rest is a scalar offset).
    // (family: a half of the activation stage is BM/2 rows = 4 waves x PPC pieces; piece i = PPC*chunk + k)
    int a_off32[2 * PPC], b_off32[2];
    unsigned a_mask[2 * PPC];
    auto a_rb = [&](int i) __attribute__((always_inline)) { return (i / PPC) * (4 * PPC) + PPC * wn + (i % PPC); };   // tile row / 8
    auto setup_x = [&](int i, int m0, int ln, int &o_off, unsigned &o_mask) __attribute__((always_inline)) {   // m0 < 0: no such tile
        const int lrow = a_rb(i) * 8 + (ln >> 3);
        const int slot = (ln & 7) ^ ((lrow >> 1) & 7);
        const int m = m0 + lrow;
        unsigned mk = 0;
        int off = 0;
        if (m0 >= 0 && m < p.M) {
            int wo, ho, img;
            split_pixel(m, p.Wo, p.Ho, p.magic_wo, p.magic_ho, p.use_magic, wo, ho, img);
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            off = (((img * p.H + hi0) * p.W + wi0) * p.in_cs) * 2 + slot * 16;
            if (p.reg3) {      // the regular 3x3 window (tap t = (t / 3, t % 3)): three row tests x three column tests
                const unsigned r0 = (unsigned)hi0 < (unsigned)p.H, r1 = (unsigned)(hi0 + 1) < (unsigned)p.H, r2 = (unsigned)(hi0 + 2) < (unsigned)p.H;
                const unsigned c0 = (unsigned)wi0 < (unsigned)p.W, c1 = (unsigned)(wi0 + 1) < (unsigned)p.W, c2 = (unsigned)(wi0 + 2) < (unsigned)p.W;
                const unsigned rm = r0 * 0x007u | r1 * 0x038u | r2 * 0x1c0u, cm = c0 * 0x049u | c1 * 0x092u | c2 * 0x124u;
                mk = rm & cm;
            } else {
#pragma unroll
                for (int t = 0; t < 9; t++) {     // branch-free
                    const int hi = hi0 + p.tap_dy[t], wi = wi0 + p.tap_dx[t];
                    const unsigned in = (unsigned)(t < p.ntaps) & (unsigned)((unsigned)hi < (unsigned)p.H) & (unsigned)((unsigned)wi < (unsigned)p.W);
                    mk |= in << t;
                }
            }
        }
        o_off = off;
        o_mask = mk;
    };
    // weight rows are PRIVATE to a wave (wave wn stages and reads channel rows wn*CW .. +CW-1): piece j = rows wn*CW + 8j .. +7;
    // the swizzle key of row 8j + r is (r >> 1) | ((j & 1) << 2): one per-lane offset for even pieces, one for odd ones
    auto setup_w = [&](int n0, int ln, int odd) __attribute__((always_inline)) {
        const int r = ln >> 3;
        const int slot = (ln & 7) ^ ((r >> 1) | (odd << 2));
        return ((n0 + wn * CW + odd * 8 + r) * p.Kpad + slot * 8) * 2;
    };
    int lane_tapoff;                       // lane t keeps the byte offset of tap t; fetched with v_readlane
    {
        const int t = lane < p.ntaps ? lane : 0;
        lane_tapoff = ((p.tap_dy[t] * p.W + p.tap_dx[t]) * p.in_cs) * 2;
    }
    const int cin_bytes = p.Cin * 2;
    const int KT = p.Kpad / BK;
    const int w16_bytes = 16 * p.Kpad * 2; // scalar distance of two weight piece pairs (16 channel rows)

    int xst = 0;                           // scalar: byte offset of the CURRENT K tile's activation stage
    auto issue_x = [&](int c, int stage_off, int tap, int cbyte) __attribute__((always_inline)) {
        tap = __builtin_amdgcn_readfirstlane(tap);
        const int tapoff = __builtin_amdgcn_readlane(lane_tapoff, tap) + __builtin_amdgcn_readfirstlane(cbyte);
        char *base = smem + Q_XBASE + __builtin_amdgcn_readfirstlane(stage_off);
#pragma unroll
        for (int k = 0; k < PPC; k++) {
            const int i = PPC * c + k;
            const bool ok = (a_mask[i] >> tap) & 1u;
            const int voff = ok ? a_off32[i] + tapoff : (int)0x80000000;   // out of range: the hardware writes zeros
            buffer_load_lds16(p.x, p.x_bytes, base + a_rb(i) * 1024, voff, 0);
        }
    };
    // (kt = the K tile's index in the packed filter = tap * C_in / 64 + channel slice, whatever ORDER the K tiles are visited in: KO)
    auto issue_w = [&](int c, int kt) __attribute__((always_inline)) {      // half c of this wave's CW weight rows: pieces WPH*c .. WPH*c + WPH-1
        const int s0 = __builtin_amdgcn_readfirstlane(kt) * (BK * 2) + NC * c * w16_bytes;
#pragma unroll
        for (int j = 0; j < WPH; j++)
            buffer_load_lds16(p.w, p.w_bytes, smem + Q_WBASE + (wn * CW + c * (CW / 2) + j * 8) * 128, b_off32[j & 1], s0 + (j >> 1) * w16_bytes);
    };

    // ---- fragment read addresses
    int px[2], pw[2];
    {
        const int frow = lane & 15, fk = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int sw = (((ks * 4 + fk) ^ ((frow >> 1) & 7)) << 4) + frow * 128;
            px[ks] = Q_XBASE + sw;
            pw[ks] = Q_WBASE + (wn * CW) * 128 + sw;
        }
    }

    f32x4 acc[2 * NC][PF];
    bf16x8 xf[PQ][2], wlo[NC][2], whi[NC][2];

    int tap1 = 0, cb1 = 0, kt1 = 0, tap2 = 0, cb2 = 0, kt2 = 0;   // K position of the K tiles one / two ahead, cyclic
    // K-tile visiting order (template parameter KO).  0: tap-major (all channels of a tap, then the next tap) -- the order of rounds 1-4.  1 (round 5):
    // CHANNEL-major: the nine taps of a 64-channel slice back to back.  The nine taps read the same input pixels shifted by a pixel or a
    // row, so with the slice innermost the lines a workgroup touches are re-used from L1 / L2 within a few K tiles instead of after a
    // whole tap's C_in / 64 K tiles; an XCD's 64 concurrent workgroups then keep 64 x 130 pixels x 128 B = 1 MB live instead of
    // C_in / 64 times that (8.4 MB at C_in 512: twice the L2).  profiles/r05_step_traffic.txt: the 512-channel data gradients at 38^2
    // fetched 697 MB per launch for 97 MB of operands.  fp32 accumulation order changes with the order (launcher: p.korder picks the instantiation;
    // a compile-time switch: as a run-time one the extra scalar state spilled into the K loop and the compiler drained vmcnt(0) there).
    auto advance = [&](int &tap, int &cb, int &kt) __attribute__((always_inline)) {
        if constexpr (KO == 1) {
            // kt walks tap * cpt + slice: + cpt per tap, back to the next slice's tap 0 after the last tap
            tap++;
            kt += cin_bytes >> 7;
            if (tap >= p.ntaps) {
                tap = 0;
                cb += BK * 2;
                kt = cb >> 7;
                if (cb >= cin_bytes) { cb = 0; kt = 0; }
            }
        } else {
            cb += BK * 2;
            kt++;
            if (cb >= cin_bytes) { cb = 0; tap++; }
            if (kt == KT) { kt = 0; tap = 0; cb = 0; }
        }
        cb = __builtin_amdgcn_readfirstlane(cb);
        kt = __builtin_amdgcn_readfirstlane(kt);
        tap = __builtin_amdgcn_readfirstlane(tap);
    };

    const int trace_off = Q_OPS + Q_SS + wn * 512;
    if constexpr (TRACE_EPI) {
        if (lane < 32) *(unsigned *)(smem + trace_off + lane * 4) = 0u;
    }
    int te_tile = 0;
    auto stamp_e = [&](int k) __attribute__((always_inline)) {
        if constexpr (TRACE_EPI) {
            if (te_tile < 4) *(unsigned *)(smem + trace_off + (te_tile * 8 + k) * 4) = (unsigned)__builtin_readcyclecounter();
        }
    };

    // fragment reads (all ds_read_b128): wlo / whi = channel fragments 0..NC-1 / NC..2NC-1 of the wave's own weight rows, xa / xb = pixel
    // fragments 0..PQ-1 / PQ..PF-1 of an activation stage
    auto read_wlo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int c = 0; c < NC; c++) wlo[c][ks] = *(const bf16x8 *)(smem + pw[ks] + c * 2048);
    };
    auto read_whi = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int c = 0; c < NC; c++) whi[c][ks] = *(const bf16x8 *)(smem + pw[ks] + (NC + c) * 2048);
    };
    auto read_x = [&](int half) __attribute__((always_inline)) {      // pixel fragments 4*half .. 4*half+3 of the current stage
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int f = 0; f < PQ; f++) xf[f][ks] = *(const bf16x8 *)(smem + px[ks] + (half * PQ + f) * 2048);
    };
    // One phase = [activation fragments] [counted wait (+ the K tile's barrier)] [one chunk of a later K tile requested]
    // [weight fragments for a LATER phase] [16 MFMAs].  Weight fragments are always read one MFMA burst ahead of their use
    // (wlo of K tile t+1 in phase 3, whi in phase 0: their registers are free there), so only the activation fragments of
    // phases 0 and 2 have LDS latency in front of their MFMAs (a second activation register set would hide that too, but
    // acc 128 + fragments 96 + staging state does not fit 256 registers: the compiler then spills accumulators in the loop).
    //   phase 0: reads xa            wlo x xa      reads whi (WB(t), retired by this phase's wait)
    //   phase 1:                     whi x xa
    //   phase 2: reads xb            wlo x xb
    //   phase 3:                     whi x xb      reads wlo of K tile t+1 (WA(t+1), retired by this phase's wait) -- except in
    //                                              the last K tile of an output tile (the epilogue wants the registers)
    bool lenient = false;                  // this K tile follows an epilogue: NST stores sit in the in-order queue
    auto phase = [&](auto PHc, bool last_kt) __attribute__((always_inline)) {
        constexpr int PH = decltype(PHc)::value;
        if constexpr (PH == 0) read_x(0);
        if constexpr (PH == 2) read_x(1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PH == 0) {      // leaves XA(t+2) (PPC pieces) [+ the epilogue's stores] in flight
            if (NST > 0 && lenient) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPC + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPC) : "memory");
        }
        if constexpr (PH == 3) {      // leaves WB(t+1) (WPH pieces) in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPH) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();          // the ONE barrier of a K tile (activation stages are shared by the four waves)
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PH == 0) issue_x(1, xst ^ Q_XB, tap1, cb1);
        if constexpr (PH == 1) issue_w(0, kt1);
        if constexpr (PH == 2) issue_w(1, kt1);
        if constexpr (PH == 3) issue_x(0, xst, tap2, cb2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PH == 0) read_whi();
        if constexpr (PH == 3) {
            if (!last_kt) read_wlo();
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int C0 = (PH & 1) ? NC : 0, F0 = (PH < 2) ? 0 : PQ;
        if constexpr (PRIO_BURST) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2 * NC * PQ; j++) {
            const int ks = j / (NC * PQ), c = (j / PQ) % NC, f = j % PQ;
            const bf16x8 wv = (C0 == 0) ? wlo[c][ks] : whi[c][ks];
            acc[C0 + c][F0 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, xf[f][ks], acc[C0 + c][F0 + f], 0, 0, 0);
        }
        if constexpr (PRIO_BURST) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- first tile: bookkeeping + chunks XA(0) XB(0) WA(0) WB(0) XA(1)
    int ti = loc;
    int m0, n0;
    {
        const int id = tstart + ti;
        const int mt = udiv_magic(id, p.magic_nt);
        m0 = mt * L::BM;
        n0 = (id - mt * p.nt) * Q_BN;
#pragma unroll
        for (int i = 0; i < 2 * PPC; i++) setup_x(i, m0, lane, a_off32[i], a_mask[i]);
        b_off32[0] = setup_w(n0, lane, 0);
        b_off32[1] = setup_w(n0, lane, 1);
    }
    float *ss = (float *)(smem + Q_OPS);   // slot = tile parity: {scale[256], shift[256]} of the tile's channels
    int sslot = 0;
    // (128-channel tiles: both halves of the workgroup load and write the same 128 entries -- NO branch around a load whose value is
    // consumed later: on the untaken path the compiler sees the load as still pending and drains vmcnt(0) inside the K loop)
    const int sst = tid & (Q_BN - 1);
    ss[sst] = p.scale[n0 + sst];
    ss[Q_BN + sst] = p.shift[n0 + sst];
    advance(tap1, cb1, kt1);               // -> K tile 1
    issue_x(0, 0, 0, 0);
    issue_x(1, 0, 0, 0);
    issue_w(0, 0);
    issue_w(1, 0);
    issue_x(0, Q_XB, tap1, cb1);
    tap2 = tap1; cb2 = cb1; kt2 = kt1;
    advance(tap2, cb2, kt2);               // -> K tile 2 (or 0 of the next tile when KT == 2)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPC) : "memory");      // everything but XA(1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    const float slope = p.slope;
    while (true) {
        const int tnext = ti + nloc;
        const bool has_next = tnext < tlen;
        int nm0 = -1, nn0 = n0;
        if (has_next) {
            const int id = tstart + tnext;
            const int mt = udiv_magic(id, p.magic_nt);
            nm0 = __builtin_amdgcn_readfirstlane(mt * L::BM);
            nn0 = __builtin_amdgcn_readfirstlane((id - mt * p.nt) * Q_BN);
        }
#pragma unroll
        for (int c = 0; c < 2 * NC; c++)
#pragma unroll
            for (int f = 0; f < PF; f++) acc[c][f] = f32x4{0.f, 0.f, 0.f, 0.f};

        stamp_e(0);
        // fragments of the tile's first K tile (its chunks were retired by the previous K loop's last wait + barrier, or by the prologue's)
        read_wlo();
        __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(disable)
        for (int t = 0; t < KT; t++) {
            int tt = t;
            asm volatile("" : "+s"(tt));
            // the look-ahead runs into the NEXT output tile: XA (issued for K tile t+2) switches before K tile KT-2, the other
            // pieces (issued for K tile t+1) before K tile KT-1
            // (the next tile's staging offsets are computed HERE, not ahead of the K loop: ten registers less across the loop,
            // and the ~150 integer instructions run under the other workgroup's MFMAs)
            if (tt == KT - 2) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int i = 0; i < PPC; i++) setup_x(i, nm0, ln, a_off32[i], a_mask[i]);
            }
            if (tt == KT - 1) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int i = PPC; i < 2 * PPC; i++) setup_x(i, nm0, ln, a_off32[i], a_mask[i]);
                b_off32[0] = setup_w(nn0, ln, 0);
                b_off32[1] = setup_w(nn0, ln, 1);
            }
            lenient = (tt == 0) && (ti != loc) && (GEN == 0 || stat_in_lds);
            if constexpr (PRIO_TOGGLE) {
                if (((tt & 1) != 0) != (2 * loc >= nloc)) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            const bool last_kt = tt == KT - 1;
            phase(ic<0>{}, last_kt);
            phase(ic<1>{}, last_kt);
            phase(ic<2>{}, last_kt);
            phase(ic<3>{}, last_kt);
            tap1 = tap2; cb1 = cb2; kt1 = kt2;
            advance(tap2, cb2, kt2);
            xst = __builtin_amdgcn_readfirstlane(xst ^ Q_XB);
#pragma unroll
            for (int ks = 0; ks < 2; ks++) px[ks] ^= Q_XB;
        }
        stamp_e(1);

        // ------------------------------------------------------------------ epilogue (registers -> global, no LDS but scale/shift)
        if constexpr (NO_EPI) {
#pragma unroll
            for (int c = 0; c < 2 * NC; c++)
#pragma unroll
                for (int f = 0; f < PF; f++) asm volatile("" ::"v"(acc[c][f]));
        } else {
          const float ssn0 = p.scale[nn0 + sst], ssn1 = p.shift[nn0 + sst];   // next tile's scale / shift: requested first, written last
          auto run_epilogue = [&](auto ACTc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(ACTc)::value;
            int ln_e = lane;
            if constexpr (GEN != 0) asm volatile("" : "+v"(ln_e));
            const int frow = ln_e & 15, fk = ln_e >> 4, fr4 = fk * 4;
            const int chq = n0 + wn * CW;                   // first channel of this wave's quarter
            const int mrow = m0 + frow;
#if defined(__HIP_DEVICE_COMPILE__)
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, p.y_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.y), 0, p.res ? p.res_bytes : 0u, 0x00020000);
#endif
            auto opix = [&](int m) -> int {                 // GEN, stride-2 dgrad parity classes: strided placement
                int j, i, img;
                split_pixel(m, p.Wo, p.Ho, p.magic_wo, p.magic_ho, p.use_magic, j, i, img);
                return (img * p.OH + (i * p.os + p.ooy)) * p.OW + (j * p.osx + p.oox);
            };
            const bool strided = GEN == 2 && p.os != 1;
            const int yoff0 = (mrow * p.out_cs + chq + fk * 8) * 2, roff0 = (mrow * p.res_cs + chq + fk * 8) * 2;
            const int ystep = 16 * p.out_cs * 2, rstep = 16 * p.res_cs * 2;
            u32x4 rv[PF][NC];                                // residual rows: all requested up front (dead fragment registers)
            const bool has_res = GEN != 1 && p.res != nullptr;
            if (has_res) {
#pragma unroll
                for (int f = 0; f < PF; f++) {
                    const int m = mrow + f * 16;
                    int voff = roff0, soff = f * rstep;
                    if (strided) { voff = (opix(m < p.M ? m : 0) * p.res_cs + chq + fk * 8) * 2; soff = 0; }
                    voff = m < p.M ? voff : (int)0x80000000;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                    for (int h = 0; h < NC; h++) rv[f][h] = __builtin_amdgcn_raw_buffer_load_b128(rrs, voff + 64 * h, soff, 0);
#endif
                }
            }
            // BNRED: the consumer block's z for the values this lane stores (same pixel grid as the output; dense rows of z_cs channels),
            // requested with the residual rows; rows past M read zeros (and carry an exact-zero gradient)
            u32x4 zv[BNRED ? PF : 1][NC];
            if constexpr (BNRED) {
#if defined(__HIP_DEVICE_COMPILE__)
                const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc((void *)br.z, 0, br.z_bytes, 0x00020000);
                const int zoff0 = (mrow * br.z_cs + chq + fk * 8) * 2, zstep = 16 * br.z_cs * 2;
#pragma unroll
                for (int f = 0; f < PF; f++) {
                    const int voff = (mrow + f * 16) < p.M ? zoff0 : (int)0x80000000;
#pragma unroll
                    for (int h = 0; h < NC; h++) zv[f][h] = __builtin_amdgcn_raw_buffer_load_b128(zrs, voff + 64 * h, f * zstep, 0);
                }
#endif
            }
            stamp_e(2);
            const float *ssc = ss + sslot * (2 * Q_BN) + wn * CW + fr4;   // + c*16: scale; + BN: shift
            // Sweep order (round 5, traffic hygiene): a wave's 64 channels of a pixel are ONE 128-B line, stored as two 64-B halves (h).  With
            // the f loop innermost over all PF pixel groups the halves of a line leave PF stores (~1 us) apart: under cached stores the L2
            // write-allocates on the first half and often writes the line back before the second arrives (profiles/r04_traffic_nt_vs_cached.txt:
            // 1.52 x the algorithmic bytes).  FSPLIT = 2 walks the pixel groups in two blocks, both halves of a block before the next block:
            // the halves of a line are PF/2 stores apart, at the price of re-reading 4 scale/shift vectors from LDS.  Same stores, same values,
            // same count (the counted waits do not change).  The statistics / folded-reduce epilogues keep one sweep per half (their per-half
            // row sums would be flushed twice as often).
            constexpr int FSPLIT = (BNRED || NC == 1) ? 1 : FS, FB = PF / FSPLIT;
#pragma unroll
            for (int fb = 0; fb < FSPLIT; fb++)
#pragma unroll
            for (int h = 0; h < NC; h++) {
                f32x4 sc[2], sh[2];
#pragma unroll
                for (int cc = 0; cc < 2; cc++) {
                    sc[cc] = *(const f32x4 *)(ssc + (2 * h + cc) * 16);
                    sh[cc] = *(const f32x4 *)(ssc + Q_BN + (2 * h + cc) * 16);
                }
                float st_sum[2][4], st_sq[2][4];
#pragma unroll
                for (int cc = 0; cc < 2; cc++)
#pragma unroll
                    for (int r = 0; r < 4; r++) st_sum[cc][r] = st_sq[cc][r] = 0.f;
                // BNRED: after the regrouping a lane stores channels cb8 .. cb8 + 7 of pixel mrow + 16 f: their BatchNorm constants
                // (re-read per tile: L1 / L2 hits) and the three running sums of this tile
                float bn_sc[8], bn_sh[8], bn_mu[8], bs1[8], bs2[8], bs3[8];
                float bn_slope = 0.f;
                if constexpr (BNRED) {
                    const int cb8 = chq + 32 * h + fk * 8;
                    // (the launcher admits whole channel tiles only, C_out % BN == 0: every chunk exists, the loads are unconditional)
                    const f32x4 a0 = *(const f32x4 *)(br.scale + cb8), a1 = *(const f32x4 *)(br.scale + cb8 + 4);
                    const f32x4 b0 = *(const f32x4 *)(br.shift + cb8), b1 = *(const f32x4 *)(br.shift + cb8 + 4);
                    const f32x4 u0 = *(const f32x4 *)(br.mean + cb8), u1 = *(const f32x4 *)(br.mean + cb8 + 4);
                    bn_slope = br.slope[0];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        bn_sc[e] = a0[e]; bn_sc[4 + e] = a1[e]; bn_sh[e] = b0[e]; bn_sh[4 + e] = b1[e]; bn_mu[e] = u0[e]; bn_mu[4 + e] = u1[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) bs1[e] = bs2[e] = bs3[e] = 0.f;
                }
#pragma unroll
                for (int f = fb * FB; f < fb * FB + FB; f++) {
                    const int m = mrow + f * 16;
                    const bool ok = m < p.M;
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
                            if (GEN == 1 && p.stat_part && ok) {      // statistics of the values as stored (bf16)
                                const float q = (float)o[r];
                                st_sum[cc][r] += q;
                                st_sq[cc][r] += q * q;
                            }
                        }
                        const uint2 u = __builtin_bit_cast(uint2, o);
                        R[cc][0] = u.x;
                        R[cc][1] = u.y;
                    }
                    // regroup so that the four lanes of a pixel hold 16 B each of 64 contiguous bytes (conv_mp.hip)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                    for (int d = 0; d < 2; d++) {
                        auto s1 = __builtin_amdgcn_permlane32_swap(R[0][d], R[1][d], false, false);
                        auto s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                        R[0][d] = s2[0];
                        R[1][d] = s2[1];
                    }
#endif
                    u32x4 out = u32x4{R[0][0], R[0][1], R[1][0], R[1][1]};
                    if (has_res) {
                        bf16x8 a = __builtin_bit_cast(bf16x8, out);
                        const bf16x8 b = __builtin_bit_cast(bf16x8, rv[f][h]);
#pragma unroll
                        for (int e = 0; e < 8; e++) a[e] = (__bf16)((float)a[e] + (float)b[e]);
                        out = __builtin_bit_cast(u32x4, a);
                    }
                    if constexpr (BNRED) {   // the arithmetic of bn_act_bwd_reduce_kernel<1> (train.hip) on the value as it is stored (bf16)
                        const bf16x8 a = __builtin_bit_cast(bf16x8, out), zb = __builtin_bit_cast(bf16x8, zv[f][h]);
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const float zf = (float)zb[e], d = ok ? (float)a[e] : 0.f;
                            const float u = zf * bn_sc[e] + bn_sh[e];
                            float g = d;
                            if (u <= 0.f) { g = d * bn_slope; bs3[e] += d * u; }
                            bs2[e] += g * (zf - bn_mu[e]);
                            bs1[e] += g;
                        }
                    }
                    if constexpr (NO_STORE) {
                        asm volatile("" ::"v"(out.x), "v"(out.y), "v"(out.z), "v"(out.w));
                    } else {
                        int voff = yoff0, soff = f * ystep;
                        if (strided) { voff = (opix(ok ? m : 0) * p.out_cs + chq + fk * 8) * 2; soff = 0; }
                        voff = ok ? voff : (int)0x80000000;
#if defined(__HIP_DEVICE_COMPILE__)
                        if (p.nt_out) buffer_store16_soff<2>(out, yrs, voff + 64 * h, soff);      // large outputs: non-temporal
                        else buffer_store16_soff<0>(out, yrs, voff + 64 * h, soff);
#endif
                    }
                }
                stamp_e(3 + h);
                if constexpr (BNRED) {     // 16-lane row sums (the lanes of a DPP row share fk, i.e. the 8 channels), lane frow < 8 keeps channel frow
                    float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float a = row16_sum(bs1[e]), b = row16_sum(bs2[e]), c = row16_sum(bs3[e]);
                        if (frow == e) { t1 = a; t2 = b; t3 = c; }
                    }
                    if (frow < 8) {
                        float *slot = stat_lds + (size_t)(n0 / Q_BN) * 3 * Q_BN + wn * CW + 32 * h + fk * 8 + frow;
                        slot[0] += t1;
                        slot[Q_BN] += t2;
                        slot[2 * Q_BN] += t3;
                    }
                }
                if (GEN == 1 && p.stat_part) {
                    double *row = p.stat_part + (size_t)((m0 / L::BM) % STAT_ROWS) * 2 * p.stat_cpad;
                    float *slot = stat_lds + (size_t)(n0 / Q_BN) * 2 * Q_BN + wn * CW;
                    float ta = 0.f, tb = 0.f;
#pragma unroll
                    for (int cc = 0; cc < 2; cc++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float a = row16_sum(st_sum[cc][r]), b = row16_sum(st_sq[cc][r]);
                            if (frow == cc * 4 + r) {
                                ta = a;
                                tb = b;
                            }
                        }
                    if (frow < 8) {
                        const int cl = (2 * h + (frow >> 2)) * 16 + fr4 + (frow & 3);
                        if (stat_in_lds) {
                            slot[cl] += ta;
                            slot[Q_BN + cl] += tb;
                        } else {
                            atomicAdd(row + chq + cl, (double)ta);
                            atomicAdd(row + p.stat_cpad + chq + cl, (double)tb);
                        }
                    }
                }
            }
          };
          if (p.act == RYOLO_ACT_LEAKY && p.slope <= 1.f) run_epilogue(ic<3>{});
          else if (p.act == RYOLO_ACT_LEAKY) run_epilogue(ic<RYOLO_ACT_LEAKY>{});
          else if (p.act == RYOLO_ACT_MISH) run_epilogue(ic<RYOLO_ACT_MISH>{});
          else run_epilogue(ic<RYOLO_ACT_LINEAR>{});
          ss[(sslot ^ 1) * (2 * Q_BN) + sst] = ssn0;
          ss[(sslot ^ 1) * (2 * Q_BN) + Q_BN + sst] = ssn1;
        }
        stamp_e(5);
        te_tile++;
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        sslot = __builtin_amdgcn_readfirstlane(sslot ^ 1);
        ti = tnext;
        m0 = nm0;
        n0 = nn0;
    }
    if constexpr (GEN == 1) {
        if (stat_in_lds) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // every wave's last accumulator update is in LDS
            double *row = p.stat_part + (size_t)(blockIdx.x % STAT_ROWS) * 2 * p.stat_cpad;
            for (int i = tid; i < p.nt * 2 * Q_BN; i += 256) {
                const int s_ = i / (2 * Q_BN), st = (i / Q_BN) & 1, ch = i % Q_BN;
                const float v = stat_lds[(size_t)(s_ * 2 + st) * Q_BN + ch];
                if (v != 0.f && s_ * Q_BN + ch < p.Cout) atomicAdd(row + (size_t)st * p.stat_cpad + s_ * Q_BN + ch, (double)v);
            }
        }
    }
    if constexpr (BNRED) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // every wave's last accumulator update is in LDS
        for (int i = tid; i < p.nt * 3 * Q_BN; i += 256) {  // this workgroup's row of part[grid][3][C]: stored, in channel order
            const int s_ = i / (3 * Q_BN), k3 = (i / Q_BN) % 3, ch = s_ * Q_BN + i % Q_BN;
            if (ch < p.Cout) {
                float v = stat_lds[i];
                if (k3 == 1) v *= br.invstd[ch];
                br.part[((size_t)blockIdx.x * 3 + k3) * p.Cout + ch] = v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead chunks behind the last tile
    if constexpr (TRACE_EPI) {
        if ((blockIdx.x == 0 || blockIdx.x == 256 || blockIdx.x == 8) && p.stat_part) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned *dst = (unsigned *)p.stat_part + ((blockIdx.x == 0 ? 0 : (blockIdx.x == 256 ? 1 : 2)) * 8 + wn) * 32;
            if (lane < 32) dst[lane] = *(const unsigned *)(smem + trace_off + lane * 4);
        }
    }
}

inline unsigned mq_magic_u32(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

inline int mq_cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    return cus;
}

#ifdef RYOLO_MP_ABLATION
void *g_q_trace_buf = nullptr;
int g_q_dbg[4] = {0, 0, 0, 0};
#endif

// workgroups of a launch over T tiles: two per CU, a multiple of 8 (XCD chunking); surplus workgroups exit at once.  The 128-channel
// tiles need 163 / 112 registers and 50 / 34 KiB of LDS, so three / four workgroups fit a CU: RYOLO_MQ128_WGPC = 3 | 4 launches that
// many (an experiment knob, read per call; measured in profiles/r05_mq128_bench.txt -- the family is LDS-bound, more residents do not help)
inline int mq_grid_for(long long T, int cw = 64) {
    int per_cu = 2;
    if (cw == 32) {
        const char *e = abl_env("RYOLO_MQ128_WGPC");
        const int v = e ? atoi(e) : 2;
        if (v >= 1 && v <= 4) per_cu = v;
    }
    const int wgs = per_cu * (mq_cu_count() & ~7);
    return T >= wgs ? wgs : (int)((T + 7) & ~7ll);
}

template <int GEN, int VAR, int CW = 64, int PF = 8, bool BNRED = false, int FS = 1, int KO = 0>
int mq_launch(ConvParams &p, const BnRed *bnred, hipStream_t stream) {
    using L = QL<CW, PF>;
#ifdef RYOLO_MP_ABLATION
    if (VAR & 1024) p.stat_part = (double *)g_q_trace_buf;
    p.dbg0 = g_q_dbg[0];
#endif
    static bool attr_done = false;
    constexpr int LDS = GEN == 1 ? L::LDS_GEN : (BNRED ? L::LDS_RED : L::LDS);
    auto kfn = conv_mq_kernel<GEN, VAR, CW, PF, BNRED, FS, KO>;
    if (!attr_done && !g_conv_choice) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return RYOLO_ELAUNCH;
        attr_done = true;
    }
    const int mt = (p.M + L::BM - 1) / L::BM;
    p.nt = (p.Cout + L::BN - 1) / L::BN;
    const long long T = (long long)mt * p.nt;
    const long long dmax = p.Wo > p.Ho ? p.Wo : p.Ho, mpad = (long long)mt * L::BM;
    if (mpad * dmax >= 0x100000000ll || T * p.nt >= 0x100000000ll || T > 0x7fffffffll) return RYOLO_EINVAL;
    p.use_magic = 1;
    p.magic_wo = mq_magic_u32(p.Wo);
    p.magic_ho = mq_magic_u32(p.Ho);
    p.magic_nt = mq_magic_u32(p.nt);
    p.ntiles = (int)T;
    {
        const unsigned long long yb = (((unsigned long long)p.N * p.OH * p.OW - 1) * p.out_cs + p.Cout) * 2ull;
        const unsigned long long rb = p.res ? (((unsigned long long)p.N * p.OH * p.OW - 1) * p.res_cs + p.Cout) * 2ull : 0ull;
        if (yb >= 0x7fffff00ull || rb >= 0x7fffff00ull) return RYOLO_EINVAL;
        p.y_bytes = (unsigned)yb;
        p.res_bytes = (unsigned)rb;
    }
    BnRed br = BnRed();
    if constexpr (BNRED) {
        if (!bnred || p.nt > L::STAT_NT || (p.Cout % L::BN) != 0) return RYOLO_EINVAL;
        br = *bnred;
        const unsigned long long zb = (((unsigned long long)p.M - 1) * br.z_cs + p.Cout) * 2ull;
        if (zb >= 0x7fffff00ull) return RYOLO_EINVAL;
        br.z_bytes = (unsigned)zb;
    }
    {
        bool reg = p.ntaps == 9;
        for (int t = 0; t < 9 && reg; t++) reg = p.tap_dy[t] == t / 3 && p.tap_dx[t] == t % 3;
        p.reg3 = reg ? 1 : 0;
    }
    p.korder = KO;
    RYOLO_CONV_DRY_RUN((CW == 64 ? RYOLO_CONV_KERNEL_MQ : (PF == 8 ? RYOLO_CONV_KERNEL_MQ128 : RYOLO_CONV_KERNEL_MQ64)));
    int grid = mq_grid_for(T, CW);
#ifdef RYOLO_MP_ABLATION
    if (g_q_dbg[1] >= 8) {
        const int wgs = g_q_dbg[1] & ~7;
        grid = T >= wgs ? wgs : (int)((T + 7) & ~7ll);
    }
#endif
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(256), LDS, stream, p, br);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

}  // namespace

namespace ryolo_detail {

int launch_conv_mq(ConvParams &p, int variant, hipStream_t stream) {
    if (!conv_mp_eligible(p)) return RYOLO_EINVAL;
    const int gen = p.stat_part != nullptr ? 1 : (p.os != 1 ? 2 : 0);
#ifdef RYOLO_MP_ABLATION
    if (gen == 0 && variant != 0) {
#define MQ_VAR(V) case V: return mq_launch<0, V>(p, nullptr, stream);
        switch (variant) {
            MQ_VAR(8) MQ_VAR(16) MQ_VAR(32) MQ_VAR(40) MQ_VAR(48) MQ_VAR(1024) MQ_VAR(1056) MQ_VAR(1032) MQ_VAR(2) MQ_VAR(4) MQ_VAR(1026) MQ_VAR(1028) MQ_VAR(18) MQ_VAR(20) MQ_VAR(256) MQ_VAR(272)
            default: return RYOLO_EINVAL;
        }
#undef MQ_VAR
    }
#endif
    if (variant != 0) return RYOLO_EINVAL;
    // K-tile order (kernel: advance()): channel-major (KO = 1) for the 3x3 launches whose C_in is 512 and more -- there the taps' re-reads of an
    // XCD's concurrent tiles (64 workgroups x 130 pixels x C_in x 2 B = 8.4 MB) are twice its L2 in tap-major order.  Measured on 3x3
    // 512->256 @38^2, bs 32 (profiles/r05_traffic_korder.txt): fetched bytes 420 -> 72 MB per launch (6.05 -> 1.31 x the algorithmic bytes);
    // bs-64 step 48.67 -> 48.50 ms with the switch on for C_in >= 256 (A/B, profiles/r05_ab_log.txt), bs-32 forward 5.852 -> 5.877 (its only
    // affected launches have C_in 256, whose footprint just fits: hence 512).  RYOLO_MQ_KORDER = 0 | 1 forces one (ryolo_set_tuning).
    bool cm = p.ntaps > 1 && p.Cin >= 512;
    {
        const char *e = tune(TUNE_MQ_KORDER);
        if (e) cm = atoi(e) != 0 && p.ntaps > 1;
        const char *m = abl_env("RYOLO_MQ_KORDER_MIN_CIN");     // (measurement build: the threshold itself, for the A/B that chose it)
        if (m && !e) cm = p.ntaps > 1 && p.Cin >= atoi(m);
    }
    if (gen == 1) {     // statistics epilogue: the two-block order too (its row sums are flushed per block: twice as often) -- bs-64 step 50.05 vs
                        // 50.20 ms (A/B on one box, profiles/r05_ab_log.txt); RYOLO_MQ_SWEEP_STATS = 1 restores one sweep per half
        const char *es = abl_env("RYOLO_MQ_SWEEP_STATS");
        if (es && atoi(es) == 1) return mq_launch<1, 0>(p, nullptr, stream);
        return cm ? mq_launch<1, 0, 64, 8, false, 2, 1>(p, nullptr, stream) : mq_launch<1, 0, 64, 8, false, 2>(p, nullptr, stream);
    }
    // The epilogue's store order (see the epilogue): two blocks of pixel groups for launches that WRITE a tensor (forward, first-writer data
    // gradients) -- HBM writes of 3x3 128->256 @76^2 at bs 32 fall from 130 to 95 MB (= the output), 1.51 -> 1.26 x the algorithmic bytes
    // (profiles/r05_traffic_sweep.txt), the bs-32 forward from 6.058 to 5.979 ms (A/B in one process, profiles/r05_ab_log.txt); one sweep per
    // half (rounds 3-4) for launches that ACCUMULATE into their output (res == y: the line is fetched anyway) and the strided stride-2
    // classes: the bs-64 step measured 49.36 ms with the old order against 49.47 with the new one everywhere.
    // (measurement build: RYOLO_MQ_SWEEP = 1 | 2 forces one order, RYOLO_MQ_SWEEP_STATS = 1 the old order of the statistics epilogue)
    const char *e = abl_env("RYOLO_MQ_SWEEP");
    const int forced = e ? atoi(e) : 0;
    const bool two = forced == 2 || (forced != 1 && gen == 0 && !(p.res && (const void *)p.res == (const void *)p.y));
    // (stride-2 parity classes: the same K-order rule as conv_mp.hip's launcher, so the two kernels add in the same order on every launch
    //  either of them can serve -- ADVICE r5: conv_mp took the channel-major order here and conv_mq did not)
    if (gen == 2) {
        if (two) return cm ? mq_launch<2, 0, 64, 8, false, 2, 1>(p, nullptr, stream) : mq_launch<2, 0, 64, 8, false, 2>(p, nullptr, stream);
        return cm ? mq_launch<2, 0, 64, 8, false, 1, 1>(p, nullptr, stream) : mq_launch<2, 0>(p, nullptr, stream);
    }
    if (two) return cm ? mq_launch<0, 0, 64, 8, false, 2, 1>(p, nullptr, stream) : mq_launch<0, 0, 64, 8, false, 2>(p, nullptr, stream);
    return cm ? mq_launch<0, 0, 64, 8, false, 1, 1>(p, nullptr, stream) : mq_launch<0, 0>(p, nullptr, stream);
}

// ---- the 128-channel members of the family (CW = 32): 128-pixel and 64-pixel tiles
bool conv_mq128_eligible(const ConvParams &p) {
    return p.fast && !p.taps2 && p.ups == 1 && !(p.stat_part && (p.res || p.os != 1)) && (p.Cin % BK) == 0 && (p.Cout % 128) == 0 &&
           p.Kpad >= 2 * BK && p.ntaps >= 1 && p.ntaps <= 9 && p.Kpad == p.ntaps * p.Cin;
}

int conv_mq128_grid(const ConvParams &p, int bm) {
    if ((bm != 128 && bm != 64) || (p.Cout % 128) != 0 || p.Cout / 128 > QL<32, 8>::STAT_NT) return 0;
    return mq_grid_for(((long long)p.M + bm - 1) / bm * (p.Cout / 128), 32);
}

// The 128-channel members are instantiated in the MEASUREMENT BUILD only (round 6): built, correct, no faster than the tiles they would
// replace (profiles/r05_mq128_bench.txt), so the shipped library neither dispatches nor contains them; tools/mq128_bench.py runs on the
// ablation library.
int launch_conv_mq128(ConvParams &p, int bm, const BnRed *bnred, hipStream_t stream) {
#ifndef RYOLO_MP_ABLATION
    (void)p; (void)bm; (void)bnred; (void)stream;
    return RYOLO_EINVAL;
#else
    if (!conv_mq128_eligible(p) || (bm != 128 && bm != 64)) return RYOLO_EINVAL;
    const int gen = p.stat_part != nullptr ? 1 : (p.os != 1 ? 2 : 0);
    if (bnred) {
        if (gen != 0) return RYOLO_EINVAL;
        return bm == 128 ? mq_launch<0, 0, 32, 8, true>(p, bnred, stream) : mq_launch<0, 0, 32, 4, true>(p, bnred, stream);
    }
    if (bm == 128) {
        if (gen == 1) return mq_launch<1, 0, 32, 8>(p, nullptr, stream);
        if (gen == 2) return mq_launch<2, 0, 32, 8>(p, nullptr, stream);
        return mq_launch<0, 0, 32, 8>(p, nullptr, stream);
    }
    if (gen == 1) return mq_launch<1, 0, 32, 4>(p, nullptr, stream);
    if (gen == 2) return mq_launch<2, 0, 32, 4>(p, nullptr, stream);
    return mq_launch<0, 0, 32, 4>(p, nullptr, stream);
#endif
}

}  // namespace ryolo_detail

#ifdef RYOLO_MP_ABLATION
extern "C" void ryolo_debug_convq_trace(void *buf) { g_q_trace_buf = buf; }
extern "C" void ryolo_debug_convq_set(int i, int v) { if (i >= 0 && i < 4) g_q_dbg[i] = v; }
#endif
