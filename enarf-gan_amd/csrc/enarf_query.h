// enarf_query.h - the density/colour query of one wavefront's 64 points (models/narf.py:176-275),
// shared by the point-cloud kernel (enarf_query_fwd) and the fused ray march (enarf_render_fwd).
//
// Two lane layouts are used inside one wave, with no LDS round trip between them:
//   point layout  lane = point (64 points): bone transforms, cube validity, part probability,
//                 and later compositing along the ray;
//   tile layout   lane = (j = lane & 15 : point of a 16-point tile, g = lane >> 4 : channel group):
//                 the 4 lanes of a point fetch one 128-B tri-plane texel as 4 x 32 B (channels
//                 8g..8g+7), and hold exactly the B-operand fragments the MFMA MLP needs.
#pragma once
#include "enarf_device.h"

#ifndef ENARF_ROUND_SWPIPE     // A/B only: address generation of the next gather round under the last loads of the current
#define ENARF_ROUND_SWPIPE 0   // one. Measured SLOWER (C1 march 0.235 -> 0.270 ms, both march kernels): DESIGN.md 3.1
#endif
#ifndef ENARF_MASK_ROW_PAIRS   // the 4 part-probability taps of a round as 2 eight-byte loads (x0, x0 + 1 are adjacent floats)
#define ENARF_MASK_ROW_PAIRS 1
#endif
#ifndef ENARF_ROUND_PRIO
#define ENARF_ROUND_PRIO 2
#endif
// Work-skipping switches exist only in diagnostic variant builds (tools/build_variant.sh NAME -DENARF_DIAG_ABLATE=k:
// 4 skips the MLP, 8 replaces the importance draw by a fixed grid); the product library is compiled with 0 and reads no
// environment variable.
#ifndef ENARF_DIAG_ABLATE
#define ENARF_DIAG_ABLATE 0
#endif
#ifndef ENARF_DIAG_SCALAR_REDUCE      // A/B only: the round-1 scalar spelling of the tap reduction
#define ENARF_DIAG_SCALAR_REDUCE 0
#endif
#ifndef ENARF_DIAG_GENERAL_TAPS       // 1: fully clamped make_taps in the gather rounds; 0: make_taps_valid
#define ENARF_DIAG_GENERAL_TAPS 1
#endif
#ifndef ENARF_DIAG_TAPCHECK           // diagnosis only: range-check what make_taps_valid would address, march with make_taps
#define ENARF_DIAG_TAPCHECK 0
#endif

namespace enarf {

// ENARF_TIMERS=1 (diagnostic builds only, tools/gpu_timers.sh): per-wave cycle sums per phase, returned in counters[0..7]
#ifndef ENARF_TIMERS
#define ENARF_TIMERS 0
#endif
#if ENARF_TIMERS == 2   // round-internal phases: 0 set-up + issue, 1 mask wait, 2 sigmoid, 3 plane-0 wait, 4 reduce 0/1 + issue 2, 5 plane-2 wait, 6 reduce 2, 7 everything else
#define TMR(S, k) do { } while (0)
#define TMR2(S, k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (S).tmr[k] += now_ - (S).tmr_t; (S).tmr_t = now_; } while (0)
#define TMR2_WAIT(S, k, imm) do { __builtin_amdgcn_s_waitcnt(imm); TMR2(S, k); } while (0)
#elif ENARF_TIMERS == 4  // ray-level stages: 0 header, 1 S2 weights, 2 S2 sampling, 3 S4 heads, 4 S4 scan, 5 S4 sums + stores, 6 barriers, 7 rest
#define TMR(S, k) do { } while (0)
#define TMR2(S, k) do { } while (0)
#define TMR2_WAIT(S, k, imm) do { } while (0)
#define TMR4(S, k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (S).tmr[k] += now_ - (S).tmr_t; (S).tmr_t = now_; } while (0)
#elif ENARF_TIMERS == 1
#define TMR2(S, k) do { } while (0)
#define TMR2_WAIT(S, k, imm) do { } while (0)
#define TMR(S, k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (S).tmr[k] += now_ - (S).tmr_t; (S).tmr_t = now_; } while (0)
#else
#define TMR(S, k) do { } while (0)
#define TMR2(S, k) do { } while (0)
#define TMR2_WAIT(S, k, imm) do { } while (0)
#endif

#ifndef TMR4
#define TMR4(S, k) do { } while (0)
#endif

struct QueryCtx {
#if ENARF_TIMERS
    mutable unsigned long long tmr[8];
    mutable unsigned long long tmr_t;
#endif
    const float *mlp;        // LDS: fp32 weights of the MLP pack [PK_W1, PK_B1) (mode F32), else unused
    const short *mlp_h;      // LDS: bf16 section (modes BF16X3 / BF16) or fp16 section (F16X3), else unused
    const float *bias;       // LDS: 144 floats = pack[PK_B1, PK_F32_FLOATS): b1[64] b2[64] b3[16]
    const float *parts;      // LDS: P x 16
    const float *canon;      // LDS: P x 12 (Rc row-major, tc)
    const float *feat;       // global: this image's channel-last feature planes [3][H][W][32]
    const float *mask;       // global: this image's part-probability planes [P*3][H][W]
    int H, W, P;
    int mult_w;              // multiply_density_with_triplane_wieght
    int clamp_mask;          // nerf_params.clamp_mask (sampling.py:46-47)
    float uniform_w;         // > 0: nerf_params.no_selector, every part weighs this (1 / P); 0: part-probability planes
#if ENARF_DIAG_TAPCHECK
    unsigned long long *diag;   // counters[5] violations, [6] first: rid | k << 32 | lane << 40 | kind << 48, [7] qx, qy bits
    unsigned diag_rid;
#endif
};

struct QueryDbg {            // optional taps of the point-cloud kernel
    float *canonical;        // (P, 3, N) for this image, or null
    float *weight;           // (P, N) for this image, or null
    long long N;
    long long i;             // this lane's point index
};

// ---- feature gather: 4 taps x 8 channels of one plane -----------------------------------------------
__device__ __forceinline__ void tap4(const float *__restrict__ base, const Taps &t, float s[8]) {
    const f32x4 *p00 = reinterpret_cast<const f32x4 *>(base + (size_t)t.o00 * kFeat);
    const f32x4 *p01 = reinterpret_cast<const f32x4 *>(base + (size_t)t.o01 * kFeat);
    const f32x4 *p10 = reinterpret_cast<const f32x4 *>(base + (size_t)t.o10 * kFeat);
    const f32x4 *p11 = reinterpret_cast<const f32x4 *>(base + (size_t)t.o11 * kFeat);
    const f32x4 a0 = p00[0], a1 = p00[1], b0 = p01[0], b1 = p01[1];
    const f32x4 c0 = p10[0], c1 = p10[1], d0 = p11[0], d1 = p11[1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        s[c] = a0[c] * t.w00;
        s[c] += b0[c] * t.w01;
        s[c] += c0[c] * t.w10;
        s[c] += d0[c] * t.w11;
        s[4 + c] = a1[c] * t.w00;
        s[4 + c] += b1[c] * t.w01;
        s[4 + c] += c1[c] * t.w10;
        s[4 + c] += d1[c] * t.w11;
    }
}

// same, addressed as (wave-uniform base pointer) + 32-bit byte offset per lane, so that the loads can use the
// SGPR-base form (global_load_dwordx4 v, v_off, s[base]) instead of 64-bit per-lane address arithmetic.
// lane_off = byte offset of this lane's 32-B chunk inside a texel plus the plane's byte offset (< 2^31 for 3 x 256^2 x 128 B).
__device__ __forceinline__ void tap4u(const char *__restrict__ base, unsigned lane_off, const Taps &t, float s[8]) {
    const f32x4 *p00 = reinterpret_cast<const f32x4 *>(base + (lane_off + ((unsigned)t.o00 << 7)));
    const f32x4 *p01 = reinterpret_cast<const f32x4 *>(base + (lane_off + ((unsigned)t.o01 << 7)));
    const f32x4 *p10 = reinterpret_cast<const f32x4 *>(base + (lane_off + ((unsigned)t.o10 << 7)));
    const f32x4 *p11 = reinterpret_cast<const f32x4 *>(base + (lane_off + ((unsigned)t.o11 << 7)));
    const f32x4 a0 = p00[0], a1 = p00[1], b0 = p01[0], b1 = p01[1];
    const f32x4 c0 = p10[0], c1 = p10[1], d0 = p11[0], d1 = p11[1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        s[c] = a0[c] * t.w00;
        s[c] += b0[c] * t.w01;
        s[c] += c0[c] * t.w10;
        s[c] += d0[c] * t.w11;
        s[4 + c] = a1[c] * t.w00;
        s[4 + c] += b1[c] * t.w01;
        s[4 + c] += c1[c] * t.w10;
        s[4 + c] += d1[c] * t.w11;
    }
}

#ifndef ENARF_PIN
#define ENARF_PIN 1
#endif
#ifndef ENARF_DIAG_HALF_LOADS
#define ENARF_DIAG_HALF_LOADS 0
#endif
// an empty asm the eight values pass through: their computation can be neither sunk below nor hoisted above this point
__device__ __forceinline__ void pin8(float v[8]) {
#if ENARF_PIN
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#endif
}

// tap4u in two halves, so that the loads of one plane can be in flight while another plane is being reduced
struct TapRegs { f32x4 a0, a1, b0, b1, c0, c1, d0, d1; };
__device__ __forceinline__ void tap4u_issue(const char *__restrict__ base, unsigned lane_off, const Taps &t, TapRegs &r) {
    const f32x4 *p00 = reinterpret_cast<const f32x4 *>(base + (lane_off + ((unsigned)t.o00 << 7)));
    const f32x4 *p01 = reinterpret_cast<const f32x4 *>(base + (lane_off + ((unsigned)t.o01 << 7)));
    const f32x4 *p10 = reinterpret_cast<const f32x4 *>(base + (lane_off + ((unsigned)t.o10 << 7)));
    const f32x4 *p11 = reinterpret_cast<const f32x4 *>(base + (lane_off + ((unsigned)t.o11 << 7)));
#if ENARF_DIAG_HALF_LOADS      // diagnosis builds only (wrong results): one 16-B load per lane and texel - what half-size texels would issue
    r.a0 = p00[0]; r.a1 = r.a0; r.b0 = p01[0]; r.b1 = r.b0;
    r.c0 = p10[0]; r.c1 = r.c0; r.d0 = p11[0]; r.d1 = r.d0;
#else
    r.a0 = p00[0]; r.a1 = p00[1]; r.b0 = p01[0]; r.b1 = p01[1];
    r.c0 = p10[0]; r.c1 = p10[1]; r.d0 = p11[0]; r.d1 = p11[1];
#endif
}
// Channel pairs as 2-vectors: each step is one v_pk_mul_f32 / v_pk_fma_f32 with the tap weight broadcast by op_sel
// (16 VALU per plane). Written with vector types on purpose: from the scalar form the SLP vectoriser pairs TAPS instead
// of channels for one of the three planes and pays ~50 moves and scalar adds per round to assemble the operands.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tap4u_reduce(const TapRegs &r, const Taps &t, float s[8]) {
#if ENARF_DIAG_SCALAR_REDUCE
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        s[c] = r.a0[c] * t.w00;
        s[c] += r.b0[c] * t.w01;
        s[c] += r.c0[c] * t.w10;
        s[c] += r.d0[c] * t.w11;
        s[4 + c] = r.a1[c] * t.w00;
        s[4 + c] += r.b1[c] * t.w01;
        s[4 + c] += r.c1[c] * t.w10;
        s[4 + c] += r.d1[c] * t.w11;
    }
#else
    const f32x2 w00 = {t.w00, t.w00}, w01 = {t.w01, t.w01}, w10 = {t.w10, t.w10}, w11 = {t.w11, t.w11};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 &a = h ? r.a1 : r.a0, &b = h ? r.b1 : r.b0, &c = h ? r.c1 : r.c0, &d = h ? r.d1 : r.d0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x2 v = f32x2{a[2 * q], a[2 * q + 1]} * w00;
            v = __builtin_elementwise_fma(f32x2{b[2 * q], b[2 * q + 1]}, w01, v);
            v = __builtin_elementwise_fma(f32x2{c[2 * q], c[2 * q + 1]}, w10, v);
            v = __builtin_elementwise_fma(f32x2{d[2 * q], d[2 * q + 1]}, w11, v);
            s[4 * h + 2 * q] = v[0];
            s[4 * h + 2 * q + 1] = v[1];
        }
    }
#endif
}

// acc[c] += weight * sum over planes xy, yz, zx of bilinear(feature plane, canonical)  (sampling.py:79-127)
__device__ __forceinline__ void gather_pair(const float *__restrict__ featg, int H, int W,
                                            float cx, float cy, float cz, float wgt, float acc[8]) {
    const size_t plane = (size_t)H * W * kFeat;
    const Taps t0 = make_taps(cx, cy, H, W);
    const Taps t1 = make_taps(cy, cz, H, W);
    const Taps t2 = make_taps(cz, cx, H, W);
    float s0[8], s1[8], s2[8];
    tap4(featg, t0, s0);
    tap4(featg + plane, t1, s1);
    tap4(featg + 2 * plane, t2, s2);
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] += ((s0[c] + s1[c]) + s2[c]) * wgt;
}

// ---- the styled MLP on one 16-point tile --------------------------------------------------------------
__device__ __forceinline__ f32x4 act4(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = styled_act(v[i]);
    return r;
}

// ---- fp32 MFMA blocks (v_mfma_f32_16x16x4_f32), accumulate in place -------------------------------------------------------
// Same rule as the split-precision MLP below (see there): every chained MFMA is issued from an asm block with tied
// accumulators, independent chains interleaved, and the wait states the compiler cannot add inside an asm string:
//   * leading  `s_nop 1`       a just-written "v" operand -> MFMA operand needs 2 states;
//   * trailing `s_nop 7; s_nop 4` = 13 states: an MFMA's D -> any reader or writer other than the next MFMA taking it whole as C
//     needs passes + 4 = 12 states for this 8-pass instruction (32 clk per instruction per SIMD; cdna_hip_programming.md 5.7
//     item 2: "8-pass XDL: 12 states"), and the block's last MFMA is followed by compiler code that reads the accumulators.
// tools/check_mfma_chains.py measures both distances in the built library (part of the CPU test suite and of build()).
// Summation order per output element = ascending k, as before: the results are bitwise those of an fmaf chain.
#define ENARF_M32(d, a, b) "v_mfma_f32_16x16x4_f32 %" #d ", %" #a ", %" #b ", %" #d "\n\t"
#define ENARF_M32_TAIL "s_nop 7\n\ts_nop 4"
// c[i] += sum_t a[i][t] * b[t], i < 4, t < 4 (t outer: four independent chains interleaved)
__device__ __forceinline__ void mfma32_acc4x4(f32x4 c[4], const float a[4][4], const float b[4]) {
    asm volatile("s_nop 1\n\t"
                 ENARF_M32(0, 4, 20) ENARF_M32(1, 8, 20) ENARF_M32(2, 12, 20) ENARF_M32(3, 16, 20)
                 ENARF_M32(0, 5, 21) ENARF_M32(1, 9, 21) ENARF_M32(2, 13, 21) ENARF_M32(3, 17, 21)
                 ENARF_M32(0, 6, 22) ENARF_M32(1, 10, 22) ENARF_M32(2, 14, 22) ENARF_M32(3, 18, 22)
                 ENARF_M32(0, 7, 23) ENARF_M32(1, 11, 23) ENARF_M32(2, 15, 23) ENARF_M32(3, 19, 23)
                 ENARF_M32_TAIL
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                 : "v"(a[0][0]), "v"(a[0][1]), "v"(a[0][2]), "v"(a[0][3]), "v"(a[1][0]), "v"(a[1][1]), "v"(a[1][2]), "v"(a[1][3]),
                   "v"(a[2][0]), "v"(a[2][1]), "v"(a[2][2]), "v"(a[2][3]), "v"(a[3][0]), "v"(a[3][1]), "v"(a[3][2]), "v"(a[3][3]),
                   "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
// c[i] += sum_t a[i][t] * b[t], i < 2, t < 8
__device__ __forceinline__ void mfma32_acc2x8(f32x4 &c0, f32x4 &c1, const float a[2][8], const float b[8]) {
    asm volatile("s_nop 1\n\t"
                 ENARF_M32(0, 2, 18) ENARF_M32(1, 10, 18) ENARF_M32(0, 3, 19) ENARF_M32(1, 11, 19)
                 ENARF_M32(0, 4, 20) ENARF_M32(1, 12, 20) ENARF_M32(0, 5, 21) ENARF_M32(1, 13, 21)
                 ENARF_M32(0, 6, 22) ENARF_M32(1, 14, 22) ENARF_M32(0, 7, 23) ENARF_M32(1, 15, 23)
                 ENARF_M32(0, 8, 24) ENARF_M32(1, 16, 24) ENARF_M32(0, 9, 25) ENARF_M32(1, 17, 25)
                 ENARF_M32_TAIL
                 : "+v"(c0), "+v"(c1)
                 : "v"(a[0][0]), "v"(a[0][1]), "v"(a[0][2]), "v"(a[0][3]), "v"(a[0][4]), "v"(a[0][5]), "v"(a[0][6]), "v"(a[0][7]),
                   "v"(a[1][0]), "v"(a[1][1]), "v"(a[1][2]), "v"(a[1][3]), "v"(a[1][4]), "v"(a[1][5]), "v"(a[1][6]), "v"(a[1][7]),
                   "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
}
// c += sum_t a[t] * b[t], t < 8: one dependent in-place chain
__device__ __forceinline__ void mfma32_acc1x8(f32x4 &c, const float a[8], const float b[8]) {
    asm volatile("s_nop 1\n\t"
                 ENARF_M32(0, 1, 9) ENARF_M32(0, 2, 10) ENARF_M32(0, 3, 11) ENARF_M32(0, 4, 12)
                 ENARF_M32(0, 5, 13) ENARF_M32(0, 6, 14) ENARF_M32(0, 7, 15) ENARF_M32(0, 8, 16)
                 ENARF_M32_TAIL
                 : "+v"(c)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
                   "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
}
// c[i] = a[i] * b (one K = 4 step into zeroed accumulators), i < 4
__device__ __forceinline__ void mfma32_set4x1(f32x4 c[4], const float a[4], float b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    asm volatile("s_nop 1\n\t"
                 ENARF_M32(0, 4, 8) ENARF_M32(1, 5, 8) ENARF_M32(2, 6, 8) ENARF_M32(3, 7, 8)
                 ENARF_M32_TAIL
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b));
}

// exact fp32 MLP on one 16-point tile: 112 MFMAs. KEEP: the post-activation hidden layers stay in a1 / a2 (backward).
// lane (j, g) holds units 16ob + 4g + r of point j; o = head (lanes with g == 0 hold (r, g, b, sigma) of point j)
__device__ __forceinline__ void mlp_tile_f32_keep(const float *__restrict__ Wp, const float *__restrict__ Bp,
                                                  const float x[8], int lane, f32x4 a1[4], f32x4 a2[4], f32x4 &o) {
    const int g = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) a1[ob] = *reinterpret_cast<const f32x4 *>(Bp + 16 * ob + 4 * g);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float a[4][4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            b[t] = x[4 * h + t];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) a[ob][t] = Wp[PK_W1 + (ob * 8 + 4 * h + t) * 64 + lane];
        }
        mfma32_acc4x4(a1, a, b);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        a1[ob] = act4(a1[ob]);
        a2[ob] = *reinterpret_cast<const f32x4 *>(Bp + 64 + 16 * ob + 4 * g);
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {            // k-steps q = 4 blk + t feed hidden unit 16 blk + 4 g + t: register t of block blk
        float a[4][4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            b[t] = a1[blk][t];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) a[ob][t] = Wp[PK_W2 + (ob * 16 + 4 * blk + t) * 64 + lane];
        }
        mfma32_acc4x4(a2, a, b);
    }
    o = *reinterpret_cast<const f32x4 *>(Bp + 128 + 4 * g);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) a2[ob] = act4(a2[ob]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float a[8], b[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int q = 8 * h + t;
            a[t] = Wp[PK_W3 + q * 64 + lane];
            b[t] = a2[q >> 2][q & 3];
        }
        mfma32_acc1x8(o, a, b);
    }
    o = act4(o);
}
__device__ __forceinline__ f32x4 mlp_tile_f32(const float *__restrict__ Wp, const float *__restrict__ Bp, const float x[8], int lane) {
    f32x4 a1[4], a2[4], o;
    mlp_tile_f32_keep(Wp, Bp, x, lane, a1, a2, o);
    return o;
}
// d act / d pre-activation from the post-activation value (act is monotone through 0; torch's leaky_relu uses the
// negative slope at exactly 0)
__device__ __forceinline__ float styled_act_grad(float a) { return (a > 0.0f ? 1.0f : 0.2f) * 1.41421356237309515f; }

// backward: dz3v = dL/dz3[unit g][point j] of this lane (lane = 16g + j); returns dz2, dz1 (accumulator layout) and
// dx[8] = dL/d feature channels 8g..8g+7 of point j (WANT_DX = false: not computed - the weight-gradient kernel needs only
// dz2 and dz1 and saves the last product's 32 MFMAs).  Wt = transposed section of the pack (LDS).
template <bool WANT_DX = true>
__device__ __forceinline__ void mlp_bwd_tile_f32(const float *__restrict__ Wt, const f32x4 a1[4], const f32x4 a2[4],
                                                 float dz3v, int lane, f32x4 dz2[4], f32x4 dz1[4], float dx[8]) {
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        float a[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) a[ob] = Wt[PKT_W3T + ob * 64 + lane];
        mfma32_set4x1(dz2, a, dz3v);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dz2[ob][r] *= styled_act_grad(a2[ob][r]);
        dz1[ob] = zero;
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        float a[4][4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            b[t] = dz2[blk][t];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) a[ob][t] = Wt[PKT_W2T + (ob * 16 + 4 * blk + t) * 64 + lane];
        }
        mfma32_acc4x4(dz1, a, b);
    }
    f32x4 d0 = zero, d1 = zero;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) dz1[ob][r] *= styled_act_grad(a1[ob][r]);
    if constexpr (!WANT_DX) {
#pragma unroll
        for (int r = 0; r < 8; ++r) dx[r] = 0.0f;
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float a[2][8], b[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int q = 8 * h + t;
            b[t] = dz1[q >> 2][q & 3];
            a[0][t] = Wt[PKT_W1T + (0 * 16 + q) * 64 + lane];
            a[1][t] = Wt[PKT_W1T + (1 * 16 + q) * 64 + lane];
        }
        mfma32_acc2x8(d0, d1, a, b);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { dx[r] = d0[r]; dx[4 + r] = d1[r]; }
}

// ---- split-precision MLP on v_mfma_f32_16x16x32_{f16,bf16} ------------------------------------------------------------
// 3-term split (hi*hi + hi*lo + lo*hi): bf16 halves ~2^-16 relative, fp16 halves ~2^-21; 1 term = plain bf16.
//
// The MFMAs are issued from asm blocks that ACCUMULATE IN PLACE (vDst == SrcC, tied "+v" operands), independent chains
// interleaved. Reason (round 2, profiles/r02_mfma_chain_hazard.md): written with the builtin, the register allocator was
// free to give a chained MFMA a vDst different from its SrcC - `c = mfma(a, b, c)` came out as e.g.
//     v_mfma_f32_16x16x32_f16 v[14:17], v[10:13], v[2:5], v[14:17]
//     v_mfma_f32_16x16x32_f16 v[10:13], v[10:13], v[6:9], v[14:17]      <- SrcC = the previous vDst, vDst elsewhere
// and the compiler put no wait state between the two (it treats "SrcC == previous vDst" as the matrix pipe's
// back-to-back case). On gfx950 that pair does not reliably see the first result: the two split modes gave run-to-run
// different images, pair counts and - through registers the late results landed in - samples gathered outside their
// part's cube, while the modes whose chains happened to be allocated in place (f32, bf16) were bit-reproducible. In-place
// chains are what every GEMM issues and need no software wait; tools/check_mfma_chains.py rejects a build whose ISA
// contains any other chained or partially overlapping form.
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float v[8], i32x4 &hi, i32x4 &lo) {          // bf16 halves, round to nearest even
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned short t = f32_to_bf16_rne(v[i]);
        h[i] = (short)t;
        l[i] = (short)f32_to_bf16_rne(v[i] - bf16_to_f32(t));
    }
    hi = __builtin_bit_cast(i32x4, h);
    lo = __builtin_bit_cast(i32x4, l);
}
// hi = f16 round-toward-zero (v_cvt_pkrtz_f16_f32: two values per instruction, saturates at +-65504 instead of
// overflowing), lo = f16(x - hi): x - hi is exact in fp32 and fits 11 bits to ~2^-21 |x|
__device__ __forceinline__ void split8h(const float v[8], i32x4 &hi, i32x4 &lo) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    f16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f16x2 a = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[i], v[i + 1]));
        const f16x2 r = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[i] - (float)a[0], v[i + 1] - (float)a[1]));
        h[i] = a[0]; h[i + 1] = a[1];
        l[i] = r[0]; l[i + 1] = r[1];
    }
    hi = __builtin_bit_cast(i32x4, h);
    lo = __builtin_bit_cast(i32x4, l);
}

// c_i += A_i . B for four independent 16x16 output blocks (A_i: weights of block i as [hi, lo][64 lanes][8] in LDS, `stride`
// shorts apart; B: the split activations). Term order per chain: hi*hi, hi*lo(B), lo(A)*hi. The trailing `s_nop 7; s_nop 1`
// = 10 states covers the D -> reader distance of the last MFMA (the compiler cannot see into the block):
// v_mfma_f32_16x16x32_{f16,bf16} is a 4-pass instruction on gfx950 (16 clk per instruction per SIMD, MI355X_MICROARCH.md
// cycle table), passes + 4 = 8 states (cdna_hip_programming.md 5.7 item 2); checked by tools/check_mfma_chains.py.
#define ENARF_MFMA4X3(OP)                                                                                                    \
    asm volatile("s_nop 1\n\t"                                                                                              \
                 OP " %0, %4, %12, %0\n\t" OP " %1, %5, %12, %1\n\t" OP " %2, %6, %12, %2\n\t" OP " %3, %7, %12, %3\n\t"          \
                 OP " %0, %4, %13, %0\n\t" OP " %1, %5, %13, %1\n\t" OP " %2, %6, %13, %2\n\t" OP " %3, %7, %13, %3\n\t"          \
                 OP " %0, %8, %12, %0\n\t" OP " %1, %9, %12, %1\n\t" OP " %2, %10, %12, %2\n\t" OP " %3, %11, %12, %3\n\t"        \
                 "s_nop 7\n\ts_nop 1"                                                                                        \
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])                                                            \
                 : "v"(ah[0]), "v"(ah[1]), "v"(ah[2]), "v"(ah[3]), "v"(al[0]), "v"(al[1]), "v"(al[2]), "v"(al[3]), "v"(bh), "v"(bl))
#define ENARF_MFMA4X1(OP)                                                                                                    \
    asm volatile("s_nop 1\n\t"                                                                                              \
                 OP " %0, %4, %8, %0\n\t" OP " %1, %5, %8, %1\n\t" OP " %2, %6, %8, %2\n\t" OP " %3, %7, %8, %3\n\t"              \
                 "s_nop 7\n\ts_nop 1"                                                                                        \
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])                                                            \
                 : "v"(ah[0]), "v"(ah[1]), "v"(ah[2]), "v"(ah[3]), "v"(bh))
template <bool F16, int NTERMS>
__device__ __forceinline__ void mfma_acc4(f32x4 c[4], const short *__restrict__ A, int stride, const i32x4 &bh, const i32x4 &bl,
                                          int lane) {
    i32x4 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ah[i] = *reinterpret_cast<const i32x4 *>(A + i * stride + lane * 8);
        if (NTERMS == 3) al[i] = *reinterpret_cast<const i32x4 *>(A + i * stride + 512 + lane * 8);
    }
    if constexpr (NTERMS == 3) {
        if constexpr (F16) ENARF_MFMA4X3("v_mfma_f32_16x16x32_f16");
        else ENARF_MFMA4X3("v_mfma_f32_16x16x32_bf16");
    } else {
        (void)bl;
        if constexpr (F16) ENARF_MFMA4X1("v_mfma_f32_16x16x32_f16");
        else ENARF_MFMA4X1("v_mfma_f32_16x16x32_bf16");
    }
}
// o += A0 . B0 + A1 . B1 on ONE accumulator (the last layer: two k-steps): a dependent in-place chain
#define ENARF_MFMA1X6(OP)                                                                                                    \
    asm volatile("s_nop 1\n\t"                                                                                              \
                 OP " %0, %1, %5, %0\n\t" OP " %0, %1, %6, %0\n\t" OP " %0, %2, %5, %0\n\t"                                      \
                 OP " %0, %3, %7, %0\n\t" OP " %0, %3, %8, %0\n\t" OP " %0, %4, %7, %0\n\t"                                      \
                 "s_nop 7\n\ts_nop 1"                                                                                        \
                 : "+v"(o)                                                                                                  \
                 : "v"(a0h), "v"(a0l), "v"(a1h), "v"(a1l), "v"(b0h), "v"(b0l), "v"(b1h), "v"(b1l))
#define ENARF_MFMA1X2(OP)                                                                                                    \
    asm volatile("s_nop 1\n\t" OP " %0, %1, %3, %0\n\t" OP " %0, %2, %4, %0\n\t" "s_nop 7\n\ts_nop 1"                       \
                 : "+v"(o) : "v"(a0h), "v"(a1h), "v"(b0h), "v"(b1h))
template <bool F16, int NTERMS>
__device__ __forceinline__ void mfma_acc1x2(f32x4 &o, const short *__restrict__ A, const i32x4 &b0h, const i32x4 &b0l,
                                            const i32x4 &b1h, const i32x4 &b1l, int lane) {
    const i32x4 a0h = *reinterpret_cast<const i32x4 *>(A + lane * 8), a1h = *reinterpret_cast<const i32x4 *>(A + 1024 + lane * 8);
    if constexpr (NTERMS == 3) {
        const i32x4 a0l = *reinterpret_cast<const i32x4 *>(A + 512 + lane * 8);
        const i32x4 a1l = *reinterpret_cast<const i32x4 *>(A + 1024 + 512 + lane * 8);
        if constexpr (F16) ENARF_MFMA1X6("v_mfma_f32_16x16x32_f16");
        else ENARF_MFMA1X6("v_mfma_f32_16x16x32_bf16");
    } else {
        (void)b0l; (void)b1l;
        if constexpr (F16) ENARF_MFMA1X2("v_mfma_f32_16x16x32_f16");
        else ENARF_MFMA1X2("v_mfma_f32_16x16x32_bf16");
    }
}

// `peak` (F16 only) returns the largest operand magnitude this lane fed into a split: features, then the two hidden
// activations. The fp16 halves carry 22 bits only while |operand| <= 65504 (above, `hi` saturates and `lo` alone cannot
// hold the rest). SCALED evaluates the same network on x * s with every bias * s: bias-add, LeakyReLU * sqrt(2) and the
// products are positively homogeneous, so every intermediate is the unscaled one times s - exactly, for s a power of two -
// and the caller divides the result by s (mlp_tile below).
template <bool F16, int NTERMS, bool SCALED>
__device__ __forceinline__ f32x4 mlp_tile_split(const float *__restrict__ Bp, const short *__restrict__ Hp, const float xin[8],
                                                int lane, float s, float &peak) {
    const int g = lane >> 4;
    f32x4 a1[4], a2[4];
    i32x4 bh, bl;
    float m = 0.0f, x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = SCALED ? xin[i] * s : xin[i];
    if (F16 && !SCALED) {
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(x[i]));
    }
    if (F16) split8h(x, bh, bl); else split8(x, bh, bl);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        a1[ob] = *reinterpret_cast<const f32x4 *>(Bp + 16 * ob + 4 * g);
        if (SCALED) a1[ob] *= s;
    }
    mfma_acc4<F16, NTERMS>(a1, Hp + PKH_W1, 1024, bh, bl, lane);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        a1[ob] = act4(a1[ob]);
        a2[ob] = *reinterpret_cast<const f32x4 *>(Bp + 64 + 16 * ob + 4 * g);
        if (SCALED) a2[ob] *= s;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = a1[2 * ks + (jj >> 2)][jj & 3];
        if (F16 && !SCALED) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) m = fmaxf(m, fabsf(v[jj]));
        }
        if (F16) split8h(v, bh, bl); else split8(v, bh, bl);
        mfma_acc4<F16, NTERMS>(a2, Hp + PKH_W2 + ks * 1024, 2048, bh, bl, lane);
    }
    f32x4 o = *reinterpret_cast<const f32x4 *>(Bp + 128 + 4 * g);
    if (SCALED) o *= s;
    i32x4 ch[2], cl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = styled_act(a2[2 * ks + (jj >> 2)][jj & 3]);
        if (F16 && !SCALED) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) m = fmaxf(m, fabsf(v[jj]));
        }
        if (F16) split8h(v, ch[ks], cl[ks]); else split8(v, ch[ks], cl[ks]);
    }
    mfma_acc1x2<F16, NTERMS>(o, Hp + PKH_W3, ch[0], cl[0], ch[1], cl[1], lane);
    peak = m;
    return act4(o);
}

template <int MODE>
__device__ __forceinline__ f32x4 mlp_tile(const QueryCtx &S, const float x[8], int lane) {
    float peak = 0.0f;
    if (MODE == ENARF_MLP_F16X3) {
        f32x4 o = mlp_tile_split<true, 3, false>(S.bias, S.mlp_h, x, lane, 1.0f, peak);
        // Range guard (tests/test_gpu_configs.py::test_mlp_arithmetic_modes_over_feature_scales): a tile in which any
        // operand left the fp16 pair's range is evaluated again, scaled by the power of two that brings the wave's
        // largest operand to [2^14, 2^15) - rare: trained features are O(1..10). A NaN operand does not trigger it and
        // propagates as it does in fp32; an infinite one stays infinite.
        if (__ballot(peak > 65504.0f) != 0) {
            float pm = peak;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) pm = fmaxf(pm, __shfl_xor(pm, off));
            if (pm < 3.0e38f) {
                int e;
                (void)frexpf(pm, &e);                                   // pm < 2^e
                const float s = ldexpf(1.0f, 15 - e), inv = ldexpf(1.0f, e - 15);
                float unused;
                o = mlp_tile_split<true, 3, true>(S.bias, S.mlp_h, x, lane, s, unused) * inv;
            }
        }
        return o;
    }
    if (MODE == ENARF_MLP_F32) return mlp_tile_f32(S.mlp, S.bias, x, lane);
    if (MODE == ENARF_MLP_BF16X3) return mlp_tile_split<false, 3, false>(S.bias, S.mlp_h, x, lane, 1.0f, peak);
    return mlp_tile_split<false, 1, false>(S.bias, S.mlp_h, x, lane, 1.0f, peak);
}

// ---- quad (4 adjacent lanes) primitives: DPP quad_perm, no LDS -----------------------------------------------
template <int CTRL>
__device__ __forceinline__ int quad_perm_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float quad_perm_f(float v) {
    return __int_as_float(quad_perm_i<CTRL>(__float_as_int(v)));
}
template <int I>
__device__ __forceinline__ float quad_bcast_f(float v) { return quad_perm_f<I * 0x55>(v); }
template <int I>
__device__ __forceinline__ int quad_bcast_i(int v) { return quad_perm_i<I * 0x55>(v); }
template <int I>
__device__ __forceinline__ Taps quad_bcast_taps(const Taps &t) {
    Taps r;
    r.o00 = quad_bcast_i<I>(t.o00); r.o01 = quad_bcast_i<I>(t.o01);
    r.o10 = quad_bcast_i<I>(t.o10); r.o11 = quad_bcast_i<I>(t.o11);
    r.w00 = quad_bcast_f<I>(t.w00); r.w01 = quad_bcast_f<I>(t.w01);
    r.w10 = quad_bcast_f<I>(t.w10); r.w11 = quad_bcast_f<I>(t.w11);
    return r;
}

// part frames / canonical frames in LDS: strides chosen so that 16 different parts read in one wave
// instruction spread over the banks (16-float stride would put parts k and k+2 on the same banks)
constexpr int kLdsPartStride = 20;   // 16 used
constexpr int kLdsCanonStride = 12;

__device__ __forceinline__ void load_frames(const QueryCtx &S, int k, float F[13], float Cn[12]) {
    const f32x4 *pf = reinterpret_cast<const f32x4 *>(S.parts + k * kLdsPartStride);
    const f32x4 *pc = reinterpret_cast<const f32x4 *>(S.canon + k * kLdsCanonStride);
    const f32x4 f0 = pf[0], f1 = pf[1], f2 = pf[2], f3 = pf[3];
    const f32x4 c0 = pc[0], c1 = pc[1], c2 = pc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { F[i] = f0[i]; F[4 + i] = f1[i]; F[8 + i] = f2[i]; }
    F[12] = f3[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) { Cn[i] = c0[i]; Cn[4 + i] = c1[i]; Cn[8 + i] = c2[i]; }
}

// ---- the query on one 16-point tile ---------------------------------------------------------------------------
// Lane layout ("gather layout"): lane = 4 * j + g, j = point of the tile (0..15), g = 32-byte chunk of a
// texel (channels 8g..8g+7). The 4 lanes of a point are adjacent, so a quad reads one contiguous 128-B line
// and exchanges data with DPP quad_perm.
//   pass A   the quad splits the candidate parts 4 ways: bone transform + cube validity -> bit mask per point
//   rounds   round r handles the r-th valid part of every point (ascending k, the reference's summation
//            order): part probability (lane g samples mask plane g) and the 12 feature texels of the pair in
//            ONE memory round trip; feat[8] accumulates in registers
//   MLP      8 cross-lane moves re-lay feat into the MFMA B-operand layout (lane = 16 g + j), then the tile MLP
// All 64 lanes call this together. cand_list (LDS, wave-private or shared) holds `ncand` part ids.
// Outputs: o = MLP head of point (lane & 15) in lanes < 16 (valid if ran); bits / wmax per point in gather layout.
template <int MODE, bool DBG>
__device__ __forceinline__ void query_tile(const QueryCtx &S, const int *cand_list, int ncand, float px, float py,
                                           float pz, bool active, int lane, f32x4 &o, bool &ran, uint32_t &bits,
                                           float &wmax, const QueryDbg &dbg, unsigned &n_pairs, unsigned &n_tiles,
                                           unsigned *n_rounds = nullptr) {
    const int g = lane & 3;
    uint32_t mine = 0;
    for (int i0 = 0; i0 < ncand; i0 += 4) {
        const int idx = i0 + g;
        const bool has = idx < ncand;
        const int k = cand_list[has ? idx : 0];
        float F[13], Cn[12], lx, ly, lz, cx, cy, cz;
        load_frames(S, k, F, Cn);
        exact_local(F, px, py, pz, lx, ly, lz);
        exact_canonical(Cn, F[12], lx, ly, lz, cx, cy, cz);
        const bool v = active && has && in_unit_cube_incl(lx, ly, lz) && in_unit_cube_strict(cx, cy, cz);
        if (v) mine |= (1u << k);
        if (DBG && active && has) {
            if (dbg.canonical) {
                dbg.canonical[((size_t)k * 3 + 0) * dbg.N + dbg.i] = cx;
                dbg.canonical[((size_t)k * 3 + 1) * dbg.N + dbg.i] = cy;
                dbg.canonical[((size_t)k * 3 + 2) * dbg.N + dbg.i] = cz;
            }
            if (dbg.weight && !v) dbg.weight[(size_t)k * dbg.N + dbg.i] = 0.125f;   // sigmoid(0)^3 (sampling.py)
        }
    }
    TMR(S, 1);
    uint32_t b = mine | (uint32_t)quad_perm_i<0xB1>((int)mine);   // [1,0,3,2]
    b |= (uint32_t)quad_perm_i<0x4E>((int)b);                     // [2,3,0,1]
    bits = b;

    float feat[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) feat[c] = 0.0f;
    wmax = 0.0f;
    const char *featb = reinterpret_cast<const char *>(S.feat);                  // wave-uniform
    const size_t mplane = (size_t)S.H * S.W;
    const unsigned fplane_b = (unsigned)(mplane * kFeat * sizeof(float));        // bytes per feature plane
    const unsigned goff = (unsigned)g << 5;                                       // this lane's 32-B chunk
    uint32_t rem = b;
#if ENARF_ROUND_PRIO
    // the gather rounds issue ahead of the other waves' MLP / sampling / compositing (+1.8 % at one frame; raising the
    // candidate tests too, or the MLP instead, gains less)
    __builtin_amdgcn_s_setprio(ENARF_ROUND_PRIO);
#endif
#if ENARF_ROUND_SWPIPE && !ENARF_DIAG_TAPCHECK && ENARF_TIMERS != 2
    // Software-pipelined rounds: the address generation of round r + 1 (next part of every point, bone transform, the
    // lane's own plane taps: ~120 VALU instructions that need only LDS and registers) runs while the last 8 loads of
    // round r (plane zx) are in flight - the point of the round with the fewest live registers - instead of in front
    // of its own loads. Same arithmetic on the same operands as the plain loop below: bit-identical results.
    uint64_t bal = __ballot(rem != 0);
    bool act = false;
    int k = 0;
    Taps t;
    auto round_addr = [&](bool &a_act, int &a_k, Taps &a_t) {
        a_act = rem != 0;
        a_k = a_act ? __builtin_ctz(rem) : 0;
        rem &= rem - 1;
        float F[13], Cn[12], lx, ly, lz, cx, cy, cz;
        load_frames(S, a_k, F, Cn);
        exact_local(F, px, py, pz, lx, ly, lz);
        exact_canonical(Cn, F[12], lx, ly, lz, cx, cy, cz);
        const float qx = (g == 1) ? cy : (g == 2) ? cz : cx;     // lane g owns plane g: xy, yz, zx (lane 3 repeats plane 0)
        const float qy = (g == 1) ? cz : (g == 2) ? cx : cy;
#if ENARF_DIAG_GENERAL_TAPS
        a_t = make_taps(qx, qy, S.H, S.W);
#else
        a_t = make_taps_valid(qx, qy, S.H, S.W);
#endif
    };
    if (bal != 0) round_addr(act, k, t);
    while (bal != 0) {
        const bool act_c = act;
        const int k_c = k;
        const uint64_t bal_c = bal;
        float acc[8], w = 0.0f;
        TapRegs r0, r1, r2;
        Taps t1, t2;
        if (act_c) {   // quad-uniform, so the quad broadcasts see all four lanes
            const int gm = (g == 3) ? 0 : g;
            const char *maskb = reinterpret_cast<const char *>(S.mask);
            const unsigned moff = __umul24((unsigned)(3 * k_c + gm), (unsigned)mplane) << 2;
#if ENARF_MASK_ROW_PAIRS
            // the four part-probability taps as two 8-byte row pairs: the texture path charges a load instruction by the
            // cache lines it touches (48 here: 16 quads x 3 planes), not by its width - two instructions instead of four
            float m00, m01, m10, m11;
            load_row_pair(maskb, moff, t.o00, t.xe, m00, m01);
            load_row_pair(maskb, moff, t.o10, t.xe, m10, m11);
#else
            const float m00 = *reinterpret_cast<const float *>(maskb + (moff + ((unsigned)t.o00 << 2)));
            const float m01 = *reinterpret_cast<const float *>(maskb + (moff + ((unsigned)t.o01 << 2)));
            const float m10 = *reinterpret_cast<const float *>(maskb + (moff + ((unsigned)t.o10 << 2)));
            const float m11 = *reinterpret_cast<const float *>(maskb + (moff + ((unsigned)t.o11 << 2)));
#endif
            const Taps t0 = quad_bcast_taps<0>(t);
            t1 = quad_bcast_taps<1>(t);
            tap4u_issue(featb, goff, t0, r0);
            tap4u_issue(featb, goff + fplane_b, t1, r1);
            __builtin_amdgcn_sched_barrier(0);
            float macc = m00 * t.w00;   // part probability plane g (sampling.py:43-48, :62)
            macc += m01 * t.w01;
            macc += m10 * t.w10;
            macc += m11 * t.w11;
            if (S.clamp_mask) macc = fminf(fmaxf(macc, -2.0f), 5.0f);
            const float sg = sigmoidf_(macc);
            const float wp = (quad_bcast_f<0>(sg) * quad_bcast_f<1>(sg)) * quad_bcast_f<2>(sg);
            w = (S.uniform_w > 0.0f) ? S.uniform_w : wp;
            tap4u_reduce(r0, t0, acc);
            pin8(acc);      // keep the reduction here: IR-level sinking would otherwise hold all 24 loads' registers
            __builtin_amdgcn_sched_barrier(0);
            t2 = quad_bcast_taps<2>(t);
            tap4u_issue(featb, goff + 2u * fplane_b, t2, r2);
            __builtin_amdgcn_sched_barrier(0);
            float s1[8];
            tap4u_reduce(r1, t1, s1);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += s1[c];
            pin8(acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        bal = __ballot(rem != 0);
        if (bal != 0) round_addr(act, k, t);        // next round's addresses, while plane zx's loads are in flight
        __builtin_amdgcn_sched_barrier(0);
        if (act_c) {
            float s2[8];
            tap4u_reduce(r2, t2, s2);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += s2[c];
#pragma unroll
            for (int c = 0; c < 8; ++c) feat[c] += acc[c] * w;
            wmax = fmaxf(wmax, w);
            if (DBG && dbg.weight && g == 0) dbg.weight[(size_t)k_c * dbg.N + dbg.i] = w;
        }
        n_pairs += (unsigned)(__popcll(bal_c) >> 2);
        if (n_rounds) *n_rounds += 1;
        TMR(S, 3);
    }
#else
    while (true) {
        const uint64_t bal = __ballot(rem != 0);
        if (bal == 0) break;
        const bool act = rem != 0;
        const int k = act ? __builtin_ctz(rem) : 0;
        rem &= rem - 1;
        float F[13], Cn[12], lx, ly, lz, cx, cy, cz;
        load_frames(S, k, F, Cn);
        exact_local(F, px, py, pz, lx, ly, lz);
        exact_canonical(Cn, F[12], lx, ly, lz, cx, cy, cz);
        // lane g owns plane g: xy, yz, zx (lane 3 repeats plane 0 and is ignored)
        const float qx = (g == 1) ? cy : (g == 2) ? cz : cx;
        const float qy = (g == 1) ? cz : (g == 2) ? cx : cy;
        // the pair is valid: |canonical| < 1 strictly, the precondition of make_taps_valid (lanes without a pair compute
        // garbage taps here and never use them: every load below sits behind `act`)
#if ENARF_DIAG_TAPCHECK
        const Taps t = make_taps(qx, qy, S.H, S.W);
        if (act && S.diag) {
            const Taps tv = make_taps_valid(qx, qy, S.H, S.W);
            const int hw = S.H * S.W;
            const bool oob = tv.o00 < 0 || tv.o00 >= hw || tv.o01 < 0 || tv.o01 >= hw || tv.o10 < 0 || tv.o10 >= hw || tv.o11 < 0 || tv.o11 >= hw;
            const bool badk = k >= S.P;
            const bool outside = !(fabsf(cx) < 1.0f && fabsf(cy) < 1.0f && fabsf(cz) < 1.0f);
            if (oob || badk || outside) {
                const unsigned long long kind = (oob ? 1ull : 0ull) | (badk ? 2ull : 0ull) | (outside ? 4ull : 0ull);
                if (atomicAdd(&S.diag[5], 1ull) == 0ull) {
                    S.diag[6] = (unsigned long long)S.diag_rid | ((unsigned long long)k << 32) | ((unsigned long long)lane << 40) | (kind << 48);
                    S.diag[7] = (unsigned long long)__float_as_uint(qx) | ((unsigned long long)__float_as_uint(qy) << 32);
                }
            }
        }
#elif ENARF_DIAG_GENERAL_TAPS
        const Taps t = make_taps(qx, qy, S.H, S.W);
#else
        const Taps t = make_taps_valid(qx, qy, S.H, S.W);
#endif
        // Pipelined round: the 4 mask taps and the 16 feature loads of planes 0 and 1 are issued back to back; the
        // part probability is formed while they are in flight; plane 2's loads go out as soon as plane 0 is reduced.
        // Two exposed memory latencies per round (the serial form below has five: the compiler splits the mask taps
        // in two waits and each plane waits on its own).
        TMR2(S, 7);
        if (act) {   // quad-uniform, so the quad broadcasts below see all four lanes
            const int gm = (g == 3) ? 0 : g;   // lane 3 repeats plane 0 (same addresses as lane 0: no extra traffic)
            const char *maskb = reinterpret_cast<const char *>(S.mask);
            const unsigned moff = (ENARF_DIAG_NOMUL24 ? (unsigned)(3 * k + gm) * (unsigned)mplane : __umul24((unsigned)(3 * k + gm), (unsigned)mplane)) << 2;   // < 2^32 bytes, mplane < 2^24: check_common
#if ENARF_MASK_ROW_PAIRS
            // the four part-probability taps as two 8-byte row pairs: the texture path charges a load instruction by the
            // cache lines it touches (48 here: 16 quads x 3 planes), not by its width - two instructions instead of four
            float m00, m01, m10, m11;
            load_row_pair(maskb, moff, t.o00, t.xe, m00, m01);
            load_row_pair(maskb, moff, t.o10, t.xe, m10, m11);
#else
            const float m00 = *reinterpret_cast<const float *>(maskb + (moff + ((unsigned)t.o00 << 2)));
            const float m01 = *reinterpret_cast<const float *>(maskb + (moff + ((unsigned)t.o01 << 2)));
            const float m10 = *reinterpret_cast<const float *>(maskb + (moff + ((unsigned)t.o10 << 2)));
            const float m11 = *reinterpret_cast<const float *>(maskb + (moff + ((unsigned)t.o11 << 2)));
#endif
            TapRegs r0, r1, r2;
            const Taps t0 = quad_bcast_taps<0>(t), t1 = quad_bcast_taps<1>(t);
            tap4u_issue(featb, goff, t0, r0);
            tap4u_issue(featb, goff + fplane_b, t1, r1);
            __builtin_amdgcn_sched_barrier(0);
            TMR2(S, 0);
            TMR2_WAIT(S, 1, 0x4F70);     // vmcnt(16)  (diagnostic build: meaningful with ENARF_MASK_ROW_PAIRS=0 only)
            float macc = m00 * t.w00;   // part probability plane g (sampling.py:43-48, :62)
            macc += m01 * t.w01;
            macc += m10 * t.w10;
            macc += m11 * t.w11;
            if (S.clamp_mask) macc = fminf(fmaxf(macc, -2.0f), 5.0f);
            const float sg = sigmoidf_(macc);
            const float wp = (quad_bcast_f<0>(sg) * quad_bcast_f<1>(sg)) * quad_bcast_f<2>(sg);
            const float w = (S.uniform_w > 0.0f) ? S.uniform_w : wp;
            float acc[8], s1[8], s2[8];
            TMR2(S, 2);
            TMR2_WAIT(S, 3, 0x0F78);     // vmcnt(8)
            tap4u_reduce(r0, t0, acc);
            pin8(acc);      // keep the reduction here: IR-level sinking would otherwise hold all 24 loads' registers
            __builtin_amdgcn_sched_barrier(0);
            const Taps t2 = quad_bcast_taps<2>(t);
            tap4u_issue(featb, goff + 2u * fplane_b, t2, r2);
            __builtin_amdgcn_sched_barrier(0);
            tap4u_reduce(r1, t1, s1);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += s1[c];
#if ENARF_TIMERS == 2
            pin8(acc);
#endif
            __builtin_amdgcn_sched_barrier(0);
            TMR2(S, 4);
            TMR2_WAIT(S, 5, 0x0F70);     // vmcnt(0)
            tap4u_reduce(r2, t2, s2);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += s2[c];
#pragma unroll
            for (int c = 0; c < 8; ++c) feat[c] += acc[c] * w;
            wmax = fmaxf(wmax, w);
            if (DBG && dbg.weight && g == 0) dbg.weight[(size_t)k * dbg.N + dbg.i] = w;
#if ENARF_TIMERS == 2
            pin8(feat);
#endif
            TMR2(S, 6);
        }
        TMR(S, 3);
        n_pairs += (unsigned)(__popcll(bal) >> 2);
        if (n_rounds) *n_rounds += 1;
        TMR(S, 3);
    }

#endif
#if ENARF_ROUND_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    ran = (__ballot(b != 0) != 0) && !(ENARF_DIAG_ABLATE & 4);
    if (ran) {
        n_tiles += 1;
        float x[8];
        const int src = ((lane & 15) << 2) | (lane >> 4);
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = __shfl(feat[c], src);
        o = mlp_tile<MODE>(S, x, lane);
    } else {
        o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    TMR(S, 4);
}

}  // namespace enarf
