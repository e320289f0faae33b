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

namespace enarf {

struct QueryCtx {
    int ablate;              // diagnosis-only switches (ENARF_ABLATE), wave-uniform; 0 in production
    const float *mlp;        // LDS: fp32 weights of the MLP pack [PK_W1, PK_B1) (mode F32), else unused
    const short *mlp_h;      // LDS: bf16 section (modes BF16X3 / BF16) or fp16 section (F16X3), else unused
    const float *bias;       // LDS: 144 floats = pack[PK_B1, PK_F32_FLOATS): b1[64] b2[64] b3[16]
    const float *parts;      // LDS: P x 16
    const float *canon;      // LDS: P x 12 (Rc row-major, tc)
    const float *feat;       // global: this image's channel-last feature planes [3][H][W][32]
    const float *mask;       // global: this image's part-probability planes [P*3][H][W]
    int H, W, P;
    int mult_w;              // multiply_density_with_triplane_wieght
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

// exact fp32: v_mfma_f32_16x16x4_f32, 112 MFMAs per tile
__device__ __forceinline__ f32x4 mlp_tile_f32(const float *__restrict__ Wp, const float *__restrict__ Bp, const float x[8], int lane) {
    const int g = lane >> 4;
    f32x4 a1[4], a2[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) a1[ob] = *reinterpret_cast<const f32x4 *>(Bp + 16 * ob + 4 * g);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
            a1[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wp[PK_W1 + (ob * 8 + s) * 64 + lane], x[s], a1[ob], 0, 0, 0);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        a1[ob] = act4(a1[ob]);
        a2[ob] = *reinterpret_cast<const f32x4 *>(Bp + 64 + 16 * ob + 4 * g);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float bq = a1[q >> 2][q & 3];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
            a2[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wp[PK_W2 + (ob * 16 + q) * 64 + lane], bq, a2[ob], 0, 0, 0);
    }
    f32x4 o = *reinterpret_cast<const f32x4 *>(Bp + 128 + 4 * g);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) a2[ob] = act4(a2[ob]);
#pragma unroll
    for (int q = 0; q < 16; ++q)
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(Wp[PK_W3 + q * 64 + lane], a2[q >> 2][q & 3], o, 0, 0, 0);
    return act4(o);       // lanes with g == 0 hold (r, g, b, sigma) pre-activation-head of point j
}

// split-bf16 (NTERMS = 3: hi*hi + hi*lo + lo*hi, ~2^-16 relative) or plain bf16 (NTERMS = 1) on
// v_mfma_f32_16x16x32_bf16: 14 (x3) MFMAs per tile
__device__ __forceinline__ void split8(const float v[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned short h = f32_to_bf16_rne(v[i]);
        hi[i] = (short)h;
        lo[i] = (short)f32_to_bf16_rne(v[i] - bf16_to_f32(h));
    }
}
template <int NTERMS>
__device__ __forceinline__ f32x4 mma_split(const short *__restrict__ Ahl /* [hi,lo][64][8] */, const bf16x8 &bh,
                                           const bf16x8 &bl, f32x4 c, int lane) {
    const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(Ahl + lane * 8);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    if (NTERMS == 3) {
        const bf16x8 al = *reinterpret_cast<const bf16x8 *>(Ahl + 512 + lane * 8);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    }
    return c;
}
template <int NTERMS>
__device__ __forceinline__ f32x4 mlp_tile_bf16(const float *__restrict__ Bp, const short *__restrict__ Hp,
                                               const float x[8], int lane) {
    const int g = lane >> 4;
    f32x4 a1[4], a2[4];
    bf16x8 bh, bl;
    split8(x, bh, bl);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        a1[ob] = *reinterpret_cast<const f32x4 *>(Bp + 16 * ob + 4 * g);
        a1[ob] = mma_split<NTERMS>(Hp + PKH_W1 + ob * 1024, bh, bl, a1[ob], lane);
        a1[ob] = act4(a1[ob]);
        a2[ob] = *reinterpret_cast<const f32x4 *>(Bp + 64 + 16 * ob + 4 * g);
    }
    f32x4 o = *reinterpret_cast<const f32x4 *>(Bp + 128 + 4 * g);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = a1[2 * ks + (jj >> 2)][jj & 3];
        split8(v, bh, bl);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
            a2[ob] = mma_split<NTERMS>(Hp + PKH_W2 + (ob * 2 + ks) * 1024, bh, bl, a2[ob], lane);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) a2[ob] = act4(a2[ob]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = a2[2 * ks + (jj >> 2)][jj & 3];
        split8(v, bh, bl);
        o = mma_split<NTERMS>(Hp + PKH_W3 + ks * 1024, bh, bl, o, lane);
    }
    return act4(o);
}

// split-fp16: same structure, 11-bit halves -> ~2^-21 relative
__device__ __forceinline__ void split8h(const float v[8], f16x8 &hi, f16x8 &lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 h = f32_to_f16_sat(v[i]);
        hi[i] = h;
        lo[i] = f32_to_f16_sat(v[i] - (float)h);
    }
}
__device__ __forceinline__ f32x4 mma_split_h(const short *__restrict__ Ahl, const f16x8 &bh, const f16x8 &bl, f32x4 c,
                                             int lane) {
    const f16x8 ah = *reinterpret_cast<const f16x8 *>(Ahl + lane * 8);
    const f16x8 al = *reinterpret_cast<const f16x8 *>(Ahl + 512 + lane * 8);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x4 mlp_tile_f16x3(const float *__restrict__ Bp, const short *__restrict__ Hp,
                                                const float x[8], int lane) {
    const int g = lane >> 4;
    f32x4 a1[4], a2[4];
    f16x8 bh, bl;
    split8h(x, bh, bl);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        a1[ob] = *reinterpret_cast<const f32x4 *>(Bp + 16 * ob + 4 * g);
        a1[ob] = mma_split_h(Hp + PKH_W1 + ob * 1024, bh, bl, a1[ob], lane);
        a1[ob] = act4(a1[ob]);
        a2[ob] = *reinterpret_cast<const f32x4 *>(Bp + 64 + 16 * ob + 4 * g);
    }
    f32x4 o = *reinterpret_cast<const f32x4 *>(Bp + 128 + 4 * g);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = a1[2 * ks + (jj >> 2)][jj & 3];
        split8h(v, bh, bl);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
            a2[ob] = mma_split_h(Hp + PKH_W2 + (ob * 2 + ks) * 1024, bh, bl, a2[ob], lane);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) a2[ob] = act4(a2[ob]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = a2[2 * ks + (jj >> 2)][jj & 3];
        split8h(v, bh, bl);
        o = mma_split_h(Hp + PKH_W3 + ks * 1024, bh, bl, o, lane);
    }
    return act4(o);
}

template <int MODE>
__device__ __forceinline__ f32x4 mlp_tile(const QueryCtx &S, const float x[8], int lane) {
    if (MODE == ENARF_MLP_F16X3) return mlp_tile_f16x3(S.bias, S.mlp_h, x, lane);
    if (MODE == ENARF_MLP_F32) return mlp_tile_f32(S.mlp, S.bias, x, lane);
    if (MODE == ENARF_MLP_BF16X3) return mlp_tile_bf16<3>(S.bias, S.mlp_h, x, lane);
    return mlp_tile_bf16<1>(S.bias, S.mlp_h, x, lane);
}

// ---- the query -------------------------------------------------------------------------------------------
// All 64 lanes must call this together (wave-uniform control flow). `cand` is a wave-uniform bit set
// of parts that can contain any of the wave's points; `active` marks lanes that carry a point.
// Outputs per lane (point layout): mlp head h = (r, g, b, sigma) after the StyledConv activation
// (valid only if bits != 0 or the lane's tile ran), bits = validity mask, wmax = max_k weight.
template <int MODE, bool DBG>
__device__ __forceinline__ void query_wave(const QueryCtx &S, uint32_t cand, float px, float py, float pz,
                                           bool active, int lane, float h[4], uint32_t &bits, float &wmax,
                                           uint64_t &tiles_run, const QueryDbg &dbg,
                                           unsigned &n_pairs, unsigned &n_tiles) {
    float feat[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 8; ++c) feat[t][c] = 0.0f;
    bits = 0;
    wmax = 0.0f;
    const int g = lane >> 4;
    const float *featg = S.feat + 8 * g;
    const size_t mplane = (size_t)S.H * S.W;

    uint32_t m = __builtin_amdgcn_readfirstlane(cand);
    while (m) {
        const int k = __builtin_ctz(m);
        m &= m - 1;
        const float *F = S.parts + k * kPartStride;
        const float *C = S.canon + k * 12;
        float lx, ly, lz, cx, cy, cz;
        exact_local(F, px, py, pz, lx, ly, lz);
        exact_canonical(C, F[12], lx, ly, lz, cx, cy, cz);
        const bool v = active && in_unit_cube_incl(lx, ly, lz) && in_unit_cube_strict(cx, cy, cz);
        if (DBG) {
            if (dbg.canonical && active) {
                dbg.canonical[((size_t)k * 3 + 0) * dbg.N + dbg.i] = cx;
                dbg.canonical[((size_t)k * 3 + 1) * dbg.N + dbg.i] = cy;
                dbg.canonical[((size_t)k * 3 + 2) * dbg.N + dbg.i] = cz;
            }
            if (dbg.weight && active && !v) dbg.weight[(size_t)k * dbg.N + dbg.i] = 0.125f;   // sampling.py: sigmoid(0)^3
        }
        const uint64_t bal = __ballot(v);
        if (bal == 0) continue;
        float w = 0.0f;
        if (v && !(S.ablate & 2)) {   // part probability: product over planes of sigmoid(bilinear)  (sampling.py:43-48, :62)
            const float *mp = S.mask + (size_t)(3 * k) * mplane;
            const float s0 = sigmoidf_(sample_scalar_plane(mp, cx, cy, S.H, S.W));
            const float s1 = sigmoidf_(sample_scalar_plane(mp + mplane, cy, cz, S.H, S.W));
            const float s2 = sigmoidf_(sample_scalar_plane(mp + 2 * mplane, cz, cx, S.H, S.W));
            w = (s0 * s1) * s2;
            bits |= (1u << k);
            wmax = fmaxf(wmax, w);
            if (DBG && dbg.weight) dbg.weight[(size_t)k * dbg.N + dbg.i] = w;
        }
        n_pairs += (unsigned)__popcll(bal);
        // tile layout: the 4 lanes of point (16t + j) fetch its 12 texels, 32 B each
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (((bal >> (16 * t)) & 0xFFFFull) == 0) continue;
            const int src = 16 * t + (lane & 15);
            const float qx = __shfl(cx, src), qy = __shfl(cy, src), qz = __shfl(cz, src);
            const float qw = __shfl(w, src);
            if (((bal >> src) & 1ull) && !(S.ablate & 1)) gather_pair(featg, S.H, S.W, qx, qy, qz, qw, feat[t]);
        }
    }

    // MLP on the tiles that hold at least one valid point
    const uint64_t anyv = __ballot(bits != 0);
    tiles_run = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (((anyv >> (16 * t)) & 0xFFFFull) == 0 || (S.ablate & 4)) continue;
        tiles_run |= (0xFFFFull << (16 * t));
        n_tiles += 1;
        const f32x4 o = mlp_tile<MODE>(S, feat[t], lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float val = __shfl(o[r], lane & 15);      // from the g == 0 lane of point j
            if ((lane >> 4) == t) h[r] = val;
        }
    }
}

}  // namespace enarf
