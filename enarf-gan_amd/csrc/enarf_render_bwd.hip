// enarf_render_bwd.hip - backward of the fused renderer. gfx950 only.
//
// What the reference's autograd computes through render (rendering.py:283-335), the fine-pass query
// (models/narf.py:176-275), StyledMLP (libraries/NeRF/net.py:10-27) and MyReLU (libraries/NeRF/activation.py:5-16):
// gradients w.r.t. the tri-plane and (through per-tile rows + a library GEMM + enarf_prepare_bwd) the MLP parameters
// and z_rend. Only the fine pass carries gradient (the importance samples are drawn without it) and poses none.
//
// One workgroup (4 waves) per ray, as in the forward:
//   F1  each wave recomputes its 16 fine samples (gather rounds + fp32 MLP, activations kept in registers)
//   F2  wave 0, lane = sample: compositing backward (prefix / suffix sums) -> dL/dz3 of every sample, via LDS
//   F3  each wave: MLP backward on MFMA (transposed weights, same accumulator-as-operand chaining), rows out
//   F4  each wave: second pass over its (sample, part) pairs: scatter d feature / d part-probability into the
//       gradient planes with float atomics
#include "enarf_march.h"
#include "enarf_host.h"

#ifndef ENARF_BWD_ABLATE
#define ENARF_BWD_ABLATE 0      // diagnosis builds only: 1 no feature atomics, 2 no mask atomics, 4 no scatter pass, 8 no row export
#endif

namespace enarf {

constexpr int kBwdWavesPerSimd = 2;

// LDS scratch of the backward kernel (floats)
constexpr int kBwdMaxSamples = 128;
constexpr int SB_CAND = 0;                                  // 4 waves x 32 ints
constexpr int SB_FH = 128;                                  // head [4][128]
constexpr int SB_FBITS = SB_FH + 4 * kBwdMaxSamples;        // bits [128]
constexpr int SB_DZ3 = SB_FBITS + kBwdMaxSamples;           // dL/dz3 [4][128]
constexpr int SB_WMAX = SB_DZ3 + 4 * kBwdMaxSamples;        // multiply_density_with_triplane_wieght: max part weight [128]
constexpr int SB_KMAX = SB_WMAX + kBwdMaxSamples;           //   the part that attains it [128] (ints)
constexpr int SB_GWM = SB_KMAX + kBwdMaxSamples;            //   dL/d(max weight) [128]
constexpr int SB_QUEUE = SB_GWM + kBwdMaxSamples;           // 2 ray slots
constexpr int kBwdScratchFloats = SB_QUEUE + 64;
static_assert(SB_QUEUE + kQueueLdsInts <= kBwdScratchFloats, "scratch overflow");

constexpr int kTRow = 33;                       // floats per texel row of the transpose tile (32 + 1 pad)
constexpr int kTTile = 2 * 16 * kTRow + 32;      // two taps x (16 texel rows) + 2 x 16 texel offsets (ints)
__host__ __device__ inline int bwd_lds_floats(int P) {
    return PK_B1 + 144 + PKT_FLOATS + P * kLdsPartStride + P * kLdsCanonStride + kBwdScratchFloats + 4 * kTTile;
}

// d loss / d one tap (texel) of every quad of the wave: grad[texel][c] += bilinear weight * w_k * dx[c].
// A quad holds the 32 channel values of its texel as 4 lanes x 8; float atomics only run at full rate when one
// wave-instruction covers whole 128-B lines (MI355X_MICROARCH.md, Global float atomics: 64 scattered dwords are
// ~17x slower), so the 16 x 32 values go through a per-wave LDS tile and come back as lane = (texel pair, channel):
// 8 atomic instructions, each adding into two contiguous 128-B texel lines.
// TWO taps at a time: the quads' rows of tap A and tap B go to the two halves of the tile; half-wave 0 then walks the 16
// quads of tap A and half-wave 1 those of tap B (lane = channel), merging runs of consecutive quads (= consecutive samples
// along the ray, which mostly fall on the same texels: the importance samples cluster at the surface) that target the
// same texel into ONE atomic.
__device__ __forceinline__ void scatter_tap2(float *__restrict__ gpl, float *tile, int offA, float cfA, int offB, float cfB,
                                             bool on, const float dxg[8], int lane) {
    const int q = lane >> 2, g = lane & 3;
    int *toff = reinterpret_cast<int *>(tile + 2 * 16 * kTRow);
    const bool liveA = on && cfA != 0.0f, liveB = on && cfB != 0.0f;
    float *rowA = tile + q * kTRow + 8 * g, *rowB = rowA + 16 * kTRow;      // 33-float rows: b32 stores
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        rowA[c] = liveA ? cfA * dxg[c] : 0.0f;
        rowB[c] = liveB ? cfB * dxg[c] : 0.0f;
    }
    if (g == 0) { toff[q] = liveA ? offA : -1; toff[16 + q] = liveB ? offB : -1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const int ch = lane & 31, h16 = (lane >> 5) * 16;
        int run_o = -1;
        float run_v = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int tq = h16 + i;
            const int o = toff[tq];
            const float v = tile[tq * kTRow + ch];
            if (o == run_o) {
                run_v += v;
            } else {
                if (run_o >= 0 && !(ENARF_BWD_ABLATE & 1)) atomicAdd(gpl + (size_t)run_o * kFeat + ch, run_v);
                run_o = o;
                run_v = v;
            }
        }
        if (run_o >= 0 && !(ENARF_BWD_ABLATE & 1)) atomicAdd(gpl + (size_t)run_o * kFeat + ch, run_v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void scatter_plane(float *__restrict__ gpl, float *tile, const Taps &t, float wk, bool on,
                                              const float dxg[8], int lane) {
    scatter_tap2(gpl, tile, t.o00, t.w00 * wk, t.o01, t.w01 * wk, on, dxg, lane);
    scatter_tap2(gpl, tile, t.o10, t.w10 * wk, t.o11, t.w11 * wk, on, dxg, lane);
}


// One scalar contribution per lane into the part-probability gradient planes. Consecutive quads (= consecutive samples
// of the ray) mostly hit the same texel of the same part, so lanes with the same plane index g first add up along the
// row of four quads (segmented scan by DPP row shifts of 4 and 8 lanes over runs of equal element index) and only the
// last lane of each run issues the atomic. EVERY lane of the wave must call this (DPP reads neighbours' registers).
__device__ __forceinline__ int dpp_i(int old, int v, int ctrl_sel) {
    return ctrl_sel == 0 ? __builtin_amdgcn_update_dpp(old, v, 0x114, 0xF, 0xF, false)      // row_shr:4
         : ctrl_sel == 1 ? __builtin_amdgcn_update_dpp(old, v, 0x118, 0xF, 0xF, false)      // row_shr:8
                         : __builtin_amdgcn_update_dpp(old, v, 0x104, 0xF, 0xF, false);     // row_shl:4
}
__device__ __forceinline__ void mask_tap_add(float *__restrict__ gmask, bool on, int elem, float v, int lane) {
    const int key = on ? elem : -2 - lane;                        // inactive lanes never match a neighbour
    float acc = on ? v : 0.0f;
    int head = (dpp_i(-1, key, 0) != key) ? 1 : 0;                // run starts here (also at the start of a row)
    {
        const float pv = __builtin_bit_cast(float, dpp_i(0, __builtin_bit_cast(int, acc), 0));
        const int ph = dpp_i(1, head, 0);
        if (!head) { acc += pv; head |= ph; }
    }
    {
        const float pv = __builtin_bit_cast(float, dpp_i(0, __builtin_bit_cast(int, acc), 1));
        const int ph = dpp_i(1, head, 1);
        if (!head) { acc += pv; head |= ph; }
    }
    const bool tail = dpp_i(-1, key, 2) != key;                   // the next quad starts another run (or the row ends)
    if (on && tail) atomicAdd(gmask + elem, acc);
}

// what the per-tile stages of the backward need besides the query context (shared by the ray and the point kernels)
struct BwdTile {
    int H, W;
    size_t mplane, fplane;
    const float *l_wt;                    // LDS: transposed weight section
    float *rows_x, *rows_h1, *rows_h2, *rows_dz1, *rows_dz2, *rows_dz3;
    long long rows_per_image;
    unsigned int *row_blocks;
    float *gfeat, *gmask;                 // this image's gradient planes
    float *ttile;                         // this wave's atomic-transpose tile (LDS)
};

// F1, gather half: cube tests over the candidate parts, then the weighted features of the valid pairs (gather layout:
// lane = 4 sample + chunk), exactly as the forward computes them
__device__ __forceinline__ void bwd_gather_tile(const QueryCtx &S, const BwdTile &T, const int *l_cand, int ncand, float px,
                                                float py, float pz, bool active, int lane, uint32_t &bits, float feat[8],
                                                float &wmax, int &kmax) {
    const int g4 = lane & 3;
    wmax = 0.0f; kmax = -1;            // max over the valid parts of the part probability and the (first) part that attains it
    {
        uint32_t mine = 0;
        for (int i0 = 0; i0 < ncand; i0 += 4) {
            const int idx = i0 + g4;
            const bool has = idx < ncand;
            const int k = l_cand[has ? idx : 0];
            float F[13], Cn[12], lx, ly, lz, cx, cy, cz;
            load_frames(S, k, F, Cn);
            exact_local(F, px, py, pz, lx, ly, lz);
            exact_canonical(Cn, F[12], lx, ly, lz, cx, cy, cz);
            if (active && has && in_unit_cube_incl(lx, ly, lz) && in_unit_cube_strict(cx, cy, cz)) mine |= (1u << k);
        }
        bits = mine | (uint32_t)quad_perm_i<0xB1>((int)mine);
        bits |= (uint32_t)quad_perm_i<0x4E>((int)bits);
#pragma unroll
        for (int c = 0; c < 8; ++c) feat[c] = 0.0f;
        uint32_t rem = (ENARF_BWD_ABLATE & 4) ? 0u : bits;
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
            const float qx = (g4 == 1) ? cy : (g4 == 2) ? cz : cx;
            const float qy = (g4 == 1) ? cz : (g4 == 2) ? cx : cy;
            const Taps t = make_taps(qx, qy, T.H, T.W);
            float sg = 1.0f;
            if (act && g4 < 3) {
                const float *mp = S.mask + (size_t)(3 * k + g4) * T.mplane;
                float acc = mp[t.o00] * t.w00;
                acc += mp[t.o01] * t.w01;
                acc += mp[t.o10] * t.w10;
                acc += mp[t.o11] * t.w11;
                if (S.clamp_mask) acc = fminf(fmaxf(acc, -2.0f), 5.0f);
                sg = sigmoidf_(acc);
            }
            const float wp = (quad_bcast_f<0>(sg) * quad_bcast_f<1>(sg)) * quad_bcast_f<2>(sg);
            const float wk = (S.uniform_w > 0.0f) ? S.uniform_w : wp;
            const Taps t0 = quad_bcast_taps<0>(t), t1 = quad_bcast_taps<1>(t), t2 = quad_bcast_taps<2>(t);
            if (act) {
                float s0[8], s1[8], s2[8];
                const float *featg = S.feat + 8 * g4;
                tap4(featg, t0, s0);
                tap4(featg + T.fplane, t1, s1);
                tap4(featg + 2 * T.fplane, t2, s2);
#pragma unroll
                for (int c = 0; c < 8; ++c) feat[c] += ((s0[c] + s1[c]) + s2[c]) * wk;
                if (wk > wmax) { wmax = wk; kmax = k; }
            }
        }
    }
}

// F3 + F4: MLP backward of one 16-sample tile (dz3v: this lane's dL/dz3 in MFMA layout), row export for the weight
// gradients, and the second pass over the pairs: d part-probability and d feature texels
// gwm / kmax (gather layout, quad-uniform): multiply_density_with_triplane_wieght sends dL/d(max part weight) to that part
__device__ __forceinline__ void bwd_backward_tile(const QueryCtx &S, const BwdTile &T, int b, float px, float py, float pz,
                                                  uint32_t bits, const f32x4 a1[4], const f32x4 a2[4], const float x[8],
                                                  float dz3v, int lane, float gwm = 0.0f, int kmax = -1) {
    const int g4 = lane & 3, j4 = lane >> 2;
    const int mj = lane & 15, mg = lane >> 4;          // MFMA layout: point mj, k-group mg
    f32x4 dz2[4], dz1[4];
    float dxm[8];
    mlp_bwd_tile_f32(T.l_wt, a1, a2, dz3v, lane, dz2, dz1, dxm);
    // rows for the weight gradients: block of 16 rows of image b
    unsigned int blk = 0;
    if (lane == 0) blk = atomicAdd(T.row_blocks + b, 1u);
    blk = (unsigned int)__builtin_amdgcn_readfirstlane((int)blk);
    const size_t row = (size_t)b * T.rows_per_image + (size_t)blk * 16 + mj;
    if (!(ENARF_BWD_ABLATE & 8)) {
        f32x4 *rx = reinterpret_cast<f32x4 *>(T.rows_x + row * 32 + 8 * mg);
        rx[0] = f32x4{x[0], x[1], x[2], x[3]};
        rx[1] = f32x4{x[4], x[5], x[6], x[7]};
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            *reinterpret_cast<f32x4 *>(T.rows_h1 + row * 64 + 16 * ob + 4 * mg) = a1[ob];
            *reinterpret_cast<f32x4 *>(T.rows_h2 + row * 64 + 16 * ob + 4 * mg) = a2[ob];
            *reinterpret_cast<f32x4 *>(T.rows_dz1 + row * 64 + 16 * ob + 4 * mg) = dz1[ob];
            *reinterpret_cast<f32x4 *>(T.rows_dz2 + row * 64 + 16 * ob + 4 * mg) = dz2[ob];
        }
        T.rows_dz3[row * 4 + mg] = dz3v;
    }
    // d feature back in the gather layout
    float dxg[8];
    const int src2 = (g4 << 4) | j4;
#pragma unroll
    for (int c = 0; c < 8; ++c) dxg[c] = __shfl(dxm[c], src2);

    // ---- F4: second pass over the pairs: d part-probability and d feature texels
    uint32_t rem = (ENARF_BWD_ABLATE & 4) ? 0u : bits;
    __builtin_amdgcn_s_setprio(2);          // memory rounds ahead of the other waves' MFMA phases (as in the forward)
    while (true) {
        const uint64_t bal = __ballot(rem != 0);
        if (bal == 0) { __builtin_amdgcn_s_setprio(0); break; }
        const bool act = rem != 0;
        const int k = act ? __builtin_ctz(rem) : 0;
        rem &= rem - 1;
        float F[13], Cn[12], lx, ly, lz, cx, cy, cz;
        load_frames(S, k, F, Cn);
        exact_local(F, px, py, pz, lx, ly, lz);
        exact_canonical(Cn, F[12], lx, ly, lz, cx, cy, cz);
        const float qx = (g4 == 1) ? cy : (g4 == 2) ? cz : cx;
        const float qy = (g4 == 1) ? cz : (g4 == 2) ? cx : cy;
        const Taps t = make_taps(qx, qy, T.H, T.W);
        float sg = 1.0f;
        if (act && g4 < 3) {
            const float *mp = S.mask + (size_t)(3 * k + g4) * T.mplane;
            float acc = mp[t.o00] * t.w00;
            acc += mp[t.o01] * t.w01;
            acc += mp[t.o10] * t.w10;
            acc += mp[t.o11] * t.w11;
            if (S.clamp_mask) acc = fminf(fmaxf(acc, -2.0f), 5.0f);     // gradient straight through (sampling.py:46-47)
            sg = sigmoidf_(acc);
        }
        const float wp = (quad_bcast_f<0>(sg) * quad_bcast_f<1>(sg)) * quad_bcast_f<2>(sg);
        const float wk = (S.uniform_w > 0.0f) ? S.uniform_w : wp;
        const Taps t0 = quad_bcast_taps<0>(t), t1 = quad_bcast_taps<1>(t), t2 = quad_bcast_taps<2>(t);
        float dot = 0.0f;
        if (act) {
            float s0[8], s1[8], s2[8];
            const float *featg = S.feat + 8 * g4;
            tap4(featg, t0, s0);
            tap4(featg + T.fplane, t1, s1);
            tap4(featg + 2 * T.fplane, t2, s2);
#pragma unroll
            for (int c = 0; c < 8; ++c) dot += dxg[c] * ((s0[c] + s1[c]) + s2[c]);
        }
        dot += quad_perm_f<0xB1>(dot);
        dot += quad_perm_f<0x4E>(dot);                                  // d loss / d w_k, quad-uniform
        if (act && k == kmax) dot += gwm;                               // density = MyReLU(.) * 10 * max_k w_k (narf.py:271-272)
        {   // w = s0 s1 s2, s = sigmoid(m): dw/dm_g = w (1 - s_g); lane g adds into the four taps of its plane
            const bool mon = act && g4 < 3 && !(ENARF_BWD_ABLATE & 2) && !(S.uniform_w > 0.0f);     // uniform weights: no plane gradient
            const float gm = mon ? dot * wp * (1.0f - sg) : 0.0f;
            const int pbase = (3 * k + g4) * (int)T.mplane;
            mask_tap_add(T.gmask, mon && t.w00 != 0.0f, pbase + t.o00, t.w00 * gm, lane);
            mask_tap_add(T.gmask, mon && t.w01 != 0.0f, pbase + t.o01, t.w01 * gm, lane);
            mask_tap_add(T.gmask, mon && t.w10 != 0.0f, pbase + t.o10, t.w10 * gm, lane);
            mask_tap_add(T.gmask, mon && t.w11 != 0.0f, pbase + t.o11, t.w11 * gm, lane);
        }
        // every lane takes part (wave-uniform): inactive quads contribute empty rows
        scatter_plane(T.gfeat, T.ttile, t0, wk, act, dxg, lane);
        scatter_plane(T.gfeat + T.fplane, T.ttile, t1, wk, act, dxg, lane);
        scatter_plane(T.gfeat + 2 * T.fplane, T.ttile, t2, wk, act, dxg, lane);
    }
}

// stage image b's MLP weights (forward + transposed sections), biases and part / canonical frames into LDS
__device__ __forceinline__ void bwd_stage_image(const void *mlp_pack, const float *parts, const float *canonical_pose, int b,
                                                int P, float *l_w, float *l_bias, float *l_wt, float *l_parts,
                                                float *l_canon, int tid) {
    const float *pf = reinterpret_cast<const float *>(reinterpret_cast<const char *>(mlp_pack) + (size_t)b * kPackBytes);
    const float *pt = reinterpret_cast<const float *>(reinterpret_cast<const char *>(pf) + kPackTOff);
    for (int i = tid; i < PK_B1 / 4; i += 256) reinterpret_cast<f32x4 *>(l_w)[i] = reinterpret_cast<const f32x4 *>(pf)[i];
    for (int i = tid; i < 144; i += 256) l_bias[i] = pf[PK_B1 + i];
    for (int i = tid; i < PKT_FLOATS / 4; i += 256) reinterpret_cast<f32x4 *>(l_wt)[i] = reinterpret_cast<const f32x4 *>(pt)[i];
    const float *parts_b = parts + (size_t)b * P * kPartStride;
    for (int i = tid; i < P * kPartStride; i += 256)
        l_parts[(i / kPartStride) * kLdsPartStride + (i % kPartStride)] = parts_b[i];
    for (int i = tid; i < P * 12; i += 256) {
        const int k = i / 12, e = i % 12;
        l_canon[i] = (e < 9) ? canonical_pose[k * 16 + (e / 3) * 4 + (e % 3)] : canonical_pose[k * 16 + (e - 9) * 4 + 3];
    }
}

// SPL = fine tiles per wave: 1 for Nf <= 64 (the activations of the tile stay in registers from F1 to F3), 2 for
// 64 < Nf <= 128 (each wave owns tiles w and w + 4; F1 keeps only the heads, F3 RECOMPUTES the tile's forward - the gathers
// of the fine pass run twice, which is cheaper than 80 more live registers per lane)
template <int SPL>
__global__ __launch_bounds__(256, kBwdWavesPerSimd) void render_bwd_kernel(const enarf_render_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, Nf = a.Nf, n = a.n;
    RayQueue rq;
    // LDS: [fp32 weights PK_B1][bias 144][transposed PKT_FLOATS][parts][canon][scratch]
    float *l_w = lds, *l_bias = l_w + PK_B1, *l_wt = l_bias + 144, *l_parts = l_wt + PKT_FLOATS;
    float *l_canon = l_parts + P * kLdsPartStride, *scratch = l_canon + P * kLdsCanonStride;
    int *l_q = reinterpret_cast<int *>(scratch + SB_QUEUE);
    rq.init(a.workspace, 0, a.B, n, l_q, tid);      // the backward's own set-up launch runs with epoch 0
    if (tid == 0) rq.pop(0);
    __syncthreads();
    int cur = rq.get(0);
    if (cur < 0) return;
    int b = -1;
    QueryCtx S;
    S.mlp = l_w; S.mlp_h = nullptr; S.bias = l_bias; S.parts = l_parts; S.canon = l_canon;
    S.H = a.H; S.W = a.W; S.P = P; S.mult_w = a.multiply_density_with_weight ? (a.uniform_part_weight ? 2 : 1) : 0;
    S.clamp_mask = a.clamp_mask; S.uniform_w = a.uniform_part_weight ? 1.0f / (float)P : 0.0f;
    int *l_cand = reinterpret_cast<int *>(scratch + SB_CAND) + wave * 32;
    float *l_fh = scratch + SB_FH, *l_dz3 = scratch + SB_DZ3;
    float *l_wmax = scratch + SB_WMAX, *l_gwm = scratch + SB_GWM;
    int *l_kmax = reinterpret_cast<int *>(scratch + SB_KMAX);
    float *ttile = scratch + kBwdScratchFloats + wave * kTTile;     // this wave's atomic-transpose tile
    uint32_t *l_fbits = reinterpret_cast<uint32_t *>(scratch + SB_FBITS);
    const int j4 = lane >> 2, g4 = lane & 3;
    const size_t mplane = (size_t)a.H * a.W, fplane = mplane * kFeat;
    int qslot = 0;
    BwdTile T;
    T.H = a.H; T.W = a.W; T.mplane = mplane; T.fplane = fplane; T.l_wt = l_wt;
    T.rows_x = a.rows_x; T.rows_h1 = a.rows_h1; T.rows_h2 = a.rows_h2;
    T.rows_dz1 = a.rows_dz1; T.rows_dz2 = a.rows_dz2; T.rows_dz3 = a.rows_dz3;
    T.rows_per_image = a.rows_per_image; T.row_blocks = a.row_blocks; T.ttile = ttile;
    T.gfeat = nullptr; T.gmask = nullptr;
    constexpr int NS = kBwdMaxSamples;              // stride of the per-sample LDS arrays

    while (cur >= 0) {
        if (tid == 0) rq.pop(qslot ^ 1);
        const uint32_t rid = (uint32_t)cur;
        const int nb = (int)(rid / (uint32_t)n), ray = (int)(rid - (uint32_t)nb * (uint32_t)n);
        if (nb != b) {   // (re)stage the image's weights (forward + transposed), biases and frames
            if (b >= 0) __syncthreads();
            b = nb;
            bwd_stage_image(a.mlp_pack, a.parts, a.canonical_pose, b, P, l_w, l_bias, l_wt, l_parts, l_canon, tid);
            S.feat = a.feat_cl + (size_t)b * a.feat_batch_stride;
            S.mask = a.mask_planes + (size_t)b * a.mask_batch_stride;
            T.gfeat = a.grad_feat_cl + (size_t)b * a.grad_feat_batch_stride;
            T.gmask = a.grad_mask_planes + (size_t)b * a.grad_mask_batch_stride;
            __syncthreads();
        }
        const RayRec rec = rq.rec(qslot);  // depth range, candidates and ray direction from the set-up pass (left in LDS by the pop)
        const float dx_ = rec.dx, dy_ = rec.dy, dz_ = rec.dz;
        const float dmin = rec.dmin, dmax = rec.dmax;
        const int ncand = build_cand_list(l_cand, rec.cand, lane);
        const float sx = exact_mul(dmin, dx_), sy = exact_mul(dmin, dy_), sz = exact_mul(dmin, dz_);
        const float ex = exact_mul(dmax, dx_), ey = exact_mul(dmax, dy_), ez = exact_mul(dmax, dz_);
        const float *bins = a.bins + ((size_t)b * n + ray) * Nf;

        // ---- F1: forward of this wave's fine tiles (gather layout: lane = 4 sample + chunk); full tiles of 16 samples
        float px[SPL], py[SPL], pz[SPL];
        uint32_t bits[SPL];
        bool ran[SPL];
        f32x4 a1[4], a2[4];        // SPL == 1: kept from F1 to F3
        float x[8];
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const int base = (wave + 4 * u) * 16, i = base + j4;
            const bool active = i < Nf - 1;
            const float bi = bins[min(i, Nf - 1)];
            px[u] = exact_lerp(sx, ex, bi); py[u] = exact_lerp(sy, ey, bi); pz[u] = exact_lerp(sz, ez, bi);
            float feat[8], wmx;
            int kmx;
            bwd_gather_tile(S, T, l_cand, ncand, px[u], py[u], pz[u], active && base < Nf, lane, bits[u], feat, wmx, kmx);
            ran[u] = __ballot(bits[u] != 0) != 0;      // wave-uniform
            f32x4 o;
            if (ran[u]) {
                const int src = ((lane & 15) << 2) | (lane >> 4);
#pragma unroll
                for (int c = 0; c < 8; ++c) x[c] = __shfl(feat[c], src);
                if (SPL == 1) mlp_tile_f32_keep(l_w, l_bias, x, lane, a1, a2, o);
                else o = mlp_tile_f32(l_w, l_bias, x, lane);
            } else {
                o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            if (lane < 16 && base + lane < Nf) {
                const int io = base + lane;
                l_fh[io] = o[0]; l_fh[NS + io] = o[1]; l_fh[2 * NS + io] = o[2]; l_fh[3 * NS + io] = o[3];
            }
            if (i < Nf && g4 == 0) { l_fbits[i] = active ? bits[u] : 0u; l_wmax[i] = wmx; l_kmax[i] = kmx; }
        }
        __syncthreads();
        const int next_ray = rq.get(qslot ^ 1);
        qslot ^= 1;

        // ---- F2 (wave 0, element e = 64 s + lane): compositing backward -> dL/dz3 (and dL/d max part weight)
        if (wave == 0) {
            const size_t ro = (size_t)b * n + ray;
            const float gC0 = a.g_color ? a.g_color[((size_t)b * 3 + 0) * n + ray] : 0.0f;
            const float gC1 = a.g_color ? a.g_color[((size_t)b * 3 + 1) * n + ray] : 0.0f;
            const float gC2 = a.g_color ? a.g_color[((size_t)b * 3 + 2) * n + ray] : 0.0f;
            const float gM = a.g_mask ? a.g_mask[ro] : 0.0f;
            const float gD = a.g_disparity ? a.g_disparity[ro] : 0.0f;
            float fdepth[SPL], dnext[SPL], aa[SPL], cs[SPL], gw[SPL], incl[SPL], G[SPL], Tt[SPL], ea[SPL], wgt[SPL];
            float h0[SPL], h1[SPL], h2[SPL], h3[SPL], cr[SPL], cg[SPL], cb[SPL], wm[SPL];
            uint32_t sb[SPL];
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int e = 64 * s + lane, ci = min(e, Nf - 1);
                sb[s] = l_fbits[ci];
                h0[s] = l_fh[ci]; h1[s] = l_fh[NS + ci]; h2[s] = l_fh[2 * NS + ci]; h3[s] = l_fh[3 * NS + ci];
                cr[s] = tanhf(h0[s]); cg[s] = tanhf(h1[s]); cb[s] = tanhf(h2[s]);
                fdepth[s] = exact_lerp(dmin, dmax, bins[ci]);
            }
            wv_next<SPL>(fdepth, dnext, lane);
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int e = 64 * s + lane;
                const bool seg = e < Nf - 1;
                const float den = seg ? density_head(h3[s], sb[s], l_wmax[min(e, Nf - 1)], S.mult_w, P) : 0.0f;
                aa[s] = seg ? den * (dnext[s] - fdepth[s]) * a.render_scale : 0.0f;
                cs[s] = aa[s];
            }
            wv_scan_incl<SPL>(cs, lane);
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const bool seg = 64 * s + lane < Nf - 1;
                Tt[s] = expf(-(cs[s] - aa[s])); ea[s] = expf(-aa[s]);
                wgt[s] = seg ? Tt[s] * (1.0f - ea[s]) : 0.0f;
                G[s] = seg ? (gC0 * cr[s] + gC1 * cg[s] + gC2 * cb[s]) + gM + gD / fdepth[s] : 0.0f;
                gw[s] = G[s] * wgt[s];
                incl[s] = gw[s];
            }
            wv_scan_incl<SPL>(incl, lane);
            const float total = __shfl(incl[SPL - 1], 63);
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int e = 64 * s + lane;
                const bool seg = e < Nf - 1;
                const float suffix = total - incl[s];                                   // sum_{j>i} G_j w_j
                const float d_a = G[s] * Tt[s] * ea[s] - suffix;                        // dL/d(sigma delta)
                const float d_sigma = seg ? d_a * (dnext[s] - fdepth[s]) * a.render_scale : 0.0f;
                // density = MyReLU(h3) * 10 [* max part weight] * any_valid; MyReLU backward: slope 0.1 for x < 0 when the
                // incoming gradient is negative (activation.py:12-16)
                float scale10 = 10.0f, g_wm = 0.0f;
                if (S.mult_w) {
                    const float wmx = l_wmax[min(e, Nf - 1)];
                    const bool all_valid = __popc(sb[s]) >= P;
                    const float w_eff = (S.mult_w == 2) ? 1.0f / (float)P : (all_valid ? wmx : fmaxf(wmx, 0.125f));
                    scale10 = 10.0f * w_eff;
                    // the max is a valid part's weight (not the 0.125 of an invalid one, not the constant 1 / P)
                    if (S.mult_w == 1 && sb[s] && (all_valid || wmx >= 0.125f)) g_wm = d_sigma * fmaxf(h3[s], 0.0f) * 10.0f;
                }
                const float gy = sb[s] ? d_sigma * scale10 : 0.0f;
                const float gx = (h3[s] >= 0.0f) ? gy : ((gy < 0.0f) ? 0.1f * gy : 0.0f);
                if (e < NS) {
                    l_dz3[3 * NS + e] = seg ? gx * styled_act_grad(h3[s]) : 0.0f;
                    l_dz3[e] = seg ? (1.0f - cr[s] * cr[s]) * (wgt[s] * gC0) * styled_act_grad(h0[s]) : 0.0f;
                    l_dz3[NS + e] = seg ? (1.0f - cg[s] * cg[s]) * (wgt[s] * gC1) * styled_act_grad(h1[s]) : 0.0f;
                    l_dz3[2 * NS + e] = seg ? (1.0f - cb[s] * cb[s]) * (wgt[s] * gC2) * styled_act_grad(h2[s]) : 0.0f;
                    l_gwm[e] = seg ? g_wm : 0.0f;
                }
                (void)wm;
            }
        }
        __syncthreads();

        // ---- F3 + F4: MLP backward and scatter, per tile (skipped when the tile has no valid sample: then dz3 = 0)
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            if (!ran[u]) continue;
            const int base = (wave + 4 * u) * 16;
            if (SPL == 2) {          // recompute the tile's forward (features, activations)
                float feat[8], wmx;
                int kmx;
                uint32_t bits2;
                const int i = base + j4;
                bwd_gather_tile(S, T, l_cand, ncand, px[u], py[u], pz[u], i < Nf - 1, lane, bits2, feat, wmx, kmx);
                const int src = ((lane & 15) << 2) | (lane >> 4);
#pragma unroll
                for (int c = 0; c < 8; ++c) x[c] = __shfl(feat[c], src);
                f32x4 o;
                mlp_tile_f32_keep(l_w, l_bias, x, lane, a1, a2, o);
            }
            const int mj = lane & 15, mg = lane >> 4;          // MFMA layout: point mj, k-group mg
            const int ms = min(base + mj, Nf - 1);
            const float dz3v = (base + mj < Nf) ? l_dz3[mg * NS + ms] : 0.0f;
            const int si = min(base + j4, Nf - 1);             // gather layout: this quad's sample
            bwd_backward_tile(S, T, b, px[u], py[u], pz[u], bits[u], a1, a2, x, dz3v, lane, l_gwm[si], l_kmax[si]);
        }
        // the next ray's first barrier orders this ray's LDS reads (l_dz3, l_fh) before their next writes
        __syncthreads();
        cur = next_ray;
    }
}

// ---- backward of the point query (a9): dL/d density, dL/d colour -> tri-plane gradients + rows for the weight gradients ----
// Same tile stages as the ray kernel, on tiles of 64 consecutive points of one image; every part is a candidate; the
// head backward is per point (no compositing): density = MyReLU(h3) * 10 * any_valid, colour = tanh(h0..2). Points
// without a valid part still run the MLP on a zero feature (narf.py:255-268), so their colour gradient reaches the
// weights and biases, as in the reference.
__global__ __launch_bounds__(256, kBwdWavesPerSimd) void query_bwd_kernel(const enarf_query_bwd_args a, long long tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const int P = a.P;
    const long long N = a.N;
    float *l_w = lds, *l_bias = l_w + PK_B1, *l_wt = l_bias + 144, *l_parts = l_wt + PKT_FLOATS;
    float *l_canon = l_parts + P * kLdsPartStride, *scratch = l_canon + P * kLdsCanonStride;
    QueryCtx S;
    S.mlp = l_w; S.mlp_h = nullptr; S.bias = l_bias; S.parts = l_parts; S.canon = l_canon;
    S.H = a.H; S.W = a.W; S.P = P; S.mult_w = a.multiply_density_with_weight ? (a.uniform_part_weight ? 2 : 1) : 0;
    S.clamp_mask = a.clamp_mask; S.uniform_w = a.uniform_part_weight ? 1.0f / (float)P : 0.0f;
    S.feat = a.feat_cl + (size_t)b * a.feat_batch_stride;
    S.mask = a.mask_planes + (size_t)b * a.mask_batch_stride;
    int *l_cand = reinterpret_cast<int *>(scratch + SB_CAND);
    float *l_fh = scratch + SB_FH, *l_dz3 = scratch + SB_DZ3;
    float *l_wmax = scratch + SB_WMAX, *l_gwm = scratch + SB_GWM;
    int *l_kmax = reinterpret_cast<int *>(scratch + SB_KMAX);
    uint32_t *l_fbits = reinterpret_cast<uint32_t *>(scratch + SB_FBITS);
    BwdTile T;
    T.H = a.H; T.W = a.W; T.mplane = (size_t)a.H * a.W; T.fplane = T.mplane * kFeat; T.l_wt = l_wt;
    T.rows_x = a.rows_x; T.rows_h1 = a.rows_h1; T.rows_h2 = a.rows_h2;
    T.rows_dz1 = a.rows_dz1; T.rows_dz2 = a.rows_dz2; T.rows_dz3 = a.rows_dz3;
    T.rows_per_image = a.rows_per_image; T.row_blocks = a.row_blocks;
    T.ttile = scratch + kBwdScratchFloats + wave * kTTile;
    T.gfeat = a.grad_feat_cl + (size_t)b * a.grad_feat_batch_stride;
    T.gmask = a.grad_mask_planes + (size_t)b * a.grad_mask_batch_stride;
    bwd_stage_image(a.mlp_pack, a.parts, a.canonical_pose, b, P, l_w, l_bias, l_wt, l_parts, l_canon, tid);
    if (tid < P) l_cand[tid] = tid;
    __syncthreads();
    const float *pp = a.points + (size_t)b * 3 * N;
    const int j4 = lane >> 2, g4 = lane & 3;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long i = tile * 64 + wave * 16 + j4;
        const bool active = i < N;
        const long long ic = active ? i : N - 1;
        const float px = pp[ic], py = pp[N + ic], pz = pp[2 * N + ic];
        uint32_t bits;
        float feat[8];
        float wmx;
        int kmx;
        bwd_gather_tile(S, T, l_cand, P, px, py, pz, active, lane, bits, feat, wmx, kmx);
        const bool ran = tile * 64 + wave * 16 < N;        // wave-uniform: the tile has points (valid part or not)
        f32x4 a1[4], a2[4], o;
        float x[8];
        if (ran) {
            const int src = ((lane & 15) << 2) | (lane >> 4);
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = __shfl(feat[c], src);
            mlp_tile_f32_keep(l_w, l_bias, x, lane, a1, a2, o);
        } else {
            o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        if (lane < 16) {
            const int io = wave * 16 + lane;
            l_fh[io] = o[0]; l_fh[64 + io] = o[1]; l_fh[128 + io] = o[2]; l_fh[192 + io] = o[3];
        }
        if (g4 == 0) { l_fbits[wave * 16 + j4] = active ? bits : 0u; l_wmax[wave * 16 + j4] = wmx; l_kmax[wave * 16 + j4] = kmx; }
        __syncthreads();
        if (wave == 0) {   // head backward, lane = point of the tile
            const long long pi = tile * 64 + lane;
            const bool in = pi < N;
            const size_t po = (size_t)b * N + (in ? pi : 0);
            const float gD = (in && a.g_density) ? a.g_density[po] : 0.0f;
            const float gC0 = (in && a.g_color) ? a.g_color[((size_t)b * 3 + 0) * N + (in ? pi : 0)] : 0.0f;
            const float gC1 = (in && a.g_color) ? a.g_color[((size_t)b * 3 + 1) * N + (in ? pi : 0)] : 0.0f;
            const float gC2 = (in && a.g_color) ? a.g_color[((size_t)b * 3 + 2) * N + (in ? pi : 0)] : 0.0f;
            const uint32_t sb = l_fbits[lane];
            const float h0 = l_fh[lane], h1 = l_fh[64 + lane], h2 = l_fh[128 + lane], h3 = l_fh[192 + lane];
            const float cr = tanhf(h0), cg = tanhf(h1), cb = tanhf(h2);
            // density = MyReLU(h3) * 10 * any_valid; MyReLU backward: slope 0.1 for x < 0 when the gradient is negative
            float scale10 = 10.0f, g_wm = 0.0f;
            if (S.mult_w) {      // density = MyReLU(h3) * 10 * max_k w_k (narf.py:271-272); invalid parts weigh 0.125, no_selector 1 / P
                const float wmx_ = l_wmax[lane];
                const bool all_valid = __popc(sb) >= P;
                scale10 = 10.0f * ((S.mult_w == 2) ? 1.0f / (float)P : (all_valid ? wmx_ : fmaxf(wmx_, 0.125f)));
                if (S.mult_w == 1 && sb && (all_valid || wmx_ >= 0.125f)) g_wm = gD * fmaxf(h3, 0.0f) * 10.0f;
            }
            l_gwm[lane] = g_wm;
            const float gy = sb ? gD * scale10 : 0.0f;
            const float gx = (h3 >= 0.0f) ? gy : ((gy < 0.0f) ? 0.1f * gy : 0.0f);
            l_dz3[192 + lane] = gx * styled_act_grad(h3);
            l_dz3[lane] = (1.0f - cr * cr) * gC0 * styled_act_grad(h0);
            l_dz3[64 + lane] = (1.0f - cg * cg) * gC1 * styled_act_grad(h1);
            l_dz3[128 + lane] = (1.0f - cb * cb) * gC2 * styled_act_grad(h2);
        }
        __syncthreads();
        if (ran) {
            const int mj = lane & 15, mg = lane >> 4;
            const float dz3v = l_dz3[mg * 64 + wave * 16 + mj];
            bwd_backward_tile(S, T, b, px, py, pz, bits, a1, a2, x, dz3v, lane, l_gwm[wave * 16 + j4], l_kmax[wave * 16 + j4]);
        }
        __syncthreads();     // l_fh / l_dz3 are rewritten by the next tile
    }
}

// ---- grad_tri[:, :96] += channel-last gradient (inverse re-layout) ---------------------------------------------------
__global__ __launch_bounds__(256) void unpack_add_kernel(const float *__restrict__ cl, float *__restrict__ out,
                                                         int out_ch_total, int H, int W) {
    constexpr int C = kFeat;
    __shared__ float tile[C * 65];
    const int tid = threadIdx.x;
    const int xb = blockIdx.x * 64, y = blockIdx.y, bp = blockIdx.z, b = bp / 3, p = bp % 3;
    const float *src = cl + ((((size_t)b * 3 + p) * H + y) * W + xb) * C;
    const int nvalid = min(64, W - xb) * C;
    for (int o = tid; o < nvalid; o += 256) tile[(o % C) * 65 + (o / C)] = src[o];
    __syncthreads();
    float *dst = out + (((size_t)b * out_ch_total + p * C) * H + y) * W;
    const int x = tid & 63;
    for (int c = tid >> 6; c < C; c += 4)
        if (xb + x < W) dst[(size_t)c * H * W + xb + x] += tile[c * 65 + x];
}

// ---- weight gradients from the exported rows: dW_l = dZ_l^T H_{l-1}, db_l = sum_rows dZ_l ------------------------------
// A reduction over ~0.5 M rows with 64 x 64 (64 x 32, 4 x 64) outputs: a library GEMM picks a tile shape made for big
// outputs and spends 0.8 ms on each; here every WAVE streams its own chunk of rows through v_mfma_f32_16x16x4_f32
// (4 rows per step, operands straight from coalesced 16-B loads: lane l reads row l/16, columns 4(l%16)..+3, and
// register r of that load is the operand block "columns 4i + r" - a permutation of the output that the second kernel
// undoes) and keeps all three products in 112 accumulator registers; the four waves of a workgroup add their partials
// up in LDS and store one partial per 1024 rows; a second kernel sums the partials. No atomics: deterministic.
constexpr int kWgRowsPerWave = 256, kWgRowsPerWg = 4 * kWgRowsPerWave;
constexpr int kWgAcc2 = 0, kWgAcc1 = 16 * 256, kWgAcc3 = kWgAcc1 + 8 * 256, kWgSum2 = kWgAcc3 + 4 * 256;
constexpr int kWgSum1 = kWgSum2 + 256, kWgSum3 = kWgSum1 + 256, kWgPartial = kWgSum3 + 64;      // floats per wave chunk

struct WeightGradParams {
    const float *x, *h1, *h2, *dz1, *dz2, *dz3;
    long long rows_per_image;
    const unsigned int *row_blocks;
    float *partial;            // [B][chunks][kWgPartial]
    int chunks;                // workgroup chunks (kWgRowsPerWg rows) per image the launch covers
    float *dW1, *dW2, *dW3, *db1, *db2, *db3;
};

__global__ __launch_bounds__(256) void weight_grad_partial_kernel(const WeightGradParams p) {
    __shared__ float red[kWgPartial];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const int chunk = blockIdx.x;
    const long long count = (long long)p.row_blocks[b] * 16;
    if ((long long)chunk * kWgRowsPerWg >= count) return;      // block-uniform
    const long long r0 = (long long)chunk * kWgRowsPerWg + (long long)wave * kWgRowsPerWave;
    const long long r1 = (r0 + kWgRowsPerWave < count) ? r0 + kWgRowsPerWave : count;   // multiple of 16; may be <= r0
    const int kg = lane >> 4, i16 = lane & 15;
    const size_t base = (size_t)b * p.rows_per_image;
    f32x4 acc2[16], acc1[8], acc3[4];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc2[t] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 8; ++t) acc1[t] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc3[t] = f32x4{0, 0, 0, 0};
    f32x4 s2 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    float s3 = 0.0f;
    for (long long r = r0; r < r1; r += 4) {
        const size_t row = base + (size_t)r + kg;
        const f32x4 vdz2 = *reinterpret_cast<const f32x4 *>(p.dz2 + row * 64 + 4 * i16);
        const f32x4 vh1 = *reinterpret_cast<const f32x4 *>(p.h1 + row * 64 + 4 * i16);
        const f32x4 vdz1 = *reinterpret_cast<const f32x4 *>(p.dz1 + row * 64 + 4 * i16);
        const f32x4 vh2 = *reinterpret_cast<const f32x4 *>(p.h2 + row * 64 + 4 * i16);
        const float2 vx = *reinterpret_cast<const float2 *>(p.x + row * 32 + 2 * i16);
        const float vdz3 = (i16 < 4) ? p.dz3[row * 4 + i16] : 0.0f;
#pragma unroll
        for (int ra = 0; ra < 4; ++ra) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                acc2[ra * 4 + rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(vdz2[ra], vh1[rb], acc2[ra * 4 + rb], 0, 0, 0);
            acc1[ra * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vdz1[ra], vx.x, acc1[ra * 2 + 0], 0, 0, 0);
            acc1[ra * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vdz1[ra], vx.y, acc1[ra * 2 + 1], 0, 0, 0);
            acc3[ra] = __builtin_amdgcn_mfma_f32_16x16x4f32(vdz3, vh2[ra], acc3[ra], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { s2[c] += vdz2[c]; s1[c] += vdz1[c]; }
        s3 += vdz3;
    }
    // the four waves add up in LDS, one after the other (a wave with no rows adds zeros), then store once
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            auto put = [&](int idx, float v) { red[idx] = (w == 0) ? v : red[idx] + v; };
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) put(kWgAcc2 + (t * 4 + q) * 64 + lane, acc2[t][q]);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) put(kWgAcc1 + (t * 4 + q) * 64 + lane, acc1[t][q]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) put(kWgAcc3 + (t * 4 + q) * 64 + lane, acc3[t][q]);
#pragma unroll
            for (int c = 0; c < 4; ++c) { put(kWgSum2 + c * 64 + lane, s2[c]); put(kWgSum1 + c * 64 + lane, s1[c]); }
            put(kWgSum3 + lane, s3);
        }
        __syncthreads();
    }
    float *out = p.partial + ((size_t)b * p.chunks + chunk) * kWgPartial;
    for (int i = threadIdx.x; i < kWgPartial; i += 256) out[i] = red[i];
}

// four threads per output element (64 elements per block): each sums every fourth partial that holds the element (and
// undoes the operand permutation), then the four add up through LDS
__global__ __launch_bounds__(256) void weight_grad_reduce_kernel(const WeightGradParams p) {
    __shared__ float red[256];
    const int el = threadIdx.x & 63, part4 = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el, b = blockIdx.y;
    constexpr int N2 = 64 * 64, N1 = 64 * 32, N3 = 4 * 64, NB = 64 + 64 + 4;
    const bool in_range = e < N2 + N1 + N3 + NB;
    const long long count = (long long)p.row_blocks[b] * 16;
    const int valid = (int)((count + kWgRowsPerWg - 1) / kWgRowsPerWg);
    const int nch = valid < p.chunks ? valid : p.chunks;
    int idx[4] = {0, 0, 0, 0}, nidx = 1;
    float *dst = nullptr;
    if (e < N2) {                       // dW2[m][c]: tile (m%4, c%4), register (m/4)%4, lane 16 (m/16) + c/4
        const int m = e / 64, c = e % 64;
        idx[0] = kWgAcc2 + (((m & 3) * 4 + (c & 3)) * 4 + ((m >> 2) & 3)) * 64 + 16 * (m >> 4) + (c >> 2);
        dst = p.dW2 + (size_t)b * N2 + e;
    } else if (e < N2 + N1) {           // dW1[m][c]: tile (m%4, c%2), lane 16 (m/16) + c/2
        const int f = e - N2, m = f / 32, c = f % 32;
        idx[0] = kWgAcc1 + (((m & 3) * 2 + (c & 1)) * 4 + ((m >> 2) & 3)) * 64 + 16 * (m >> 4) + (c >> 1);
        dst = p.dW1 + (size_t)b * N1 + f;
    } else if (e < N2 + N1 + N3) {      // dW3[o][c]: tile c%4, register o, lanes 0..15
        const int f = e - N2 - N1, o = f / 64, c = f % 64;
        idx[0] = kWgAcc3 + ((c & 3) * 4 + o) * 64 + (c >> 2);
        dst = p.dW3 + (size_t)b * N3 + f;
    } else if (in_range) {              // biases: column sums held by the four k-groups of lanes
        const int f = e - N2 - N1 - N3;
        nidx = 4;
        if (f < 64) {
            for (int g = 0; g < 4; ++g) idx[g] = kWgSum1 + (f & 3) * 64 + 16 * g + (f >> 2);
            dst = p.db1 + (size_t)b * 64 + f;
        } else if (f < 128) {
            const int c = f - 64;
            for (int g = 0; g < 4; ++g) idx[g] = kWgSum2 + (c & 3) * 64 + 16 * g + (c >> 2);
            dst = p.db2 + (size_t)b * 64 + c;
        } else {
            const int o = f - 128;
            for (int g = 0; g < 4; ++g) idx[g] = kWgSum3 + 16 * g + o;
            dst = p.db3 + (size_t)b * 4 + o;
        }
    }
    const float *part = p.partial + (size_t)b * p.chunks * kWgPartial;
    float acc = 0.0f;
    if (in_range)
        for (int ch = part4; ch < nch; ch += 4)
            for (int g = 0; g < nidx; ++g) acc += part[(size_t)ch * kWgPartial + idx[g]];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (in_range && part4 == 0) *dst = ((red[el] + red[64 + el]) + red[128 + el]) + red[192 + el];
}

// ---- backward of ModulatedConv1d's weight path: dW' -> d conv.weight, d modulation.{weight,bias}, d z_rend ---------------
// u = (W / sqrt(in)) * s (s broadcast over rows), W' = u / max(||u||_row, 1e-12), s = z Wm^T / sqrt(D) + bm
__global__ __launch_bounds__(256) void prepare_bwd_kernel(const enarf_prepare_bwd_args a) {
    const int layer = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cin = (layer == 0) ? kFeat : kHid, cout = (layer == 2) ? 4 : kHid, D = a.style_dim;
    __shared__ float s_style[kHid], s_ds[kHid], s_rn[kHid], s_dot[kHid];
    __shared__ float s_u[kHid * kHid];
    const float *z = a.z_rend + (size_t)b * D;
    const float mscale = 1.0f / sqrtf((float)D), cscale = 1.0f / sqrtf((float)cin);
    if (tid < cin) {
        const float *wm = a.mod_weight[layer] + (size_t)tid * D;
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) acc += z[d] * (wm[d] * mscale);
        s_style[tid] = acc + a.mod_bias[layer][tid];
        s_ds[tid] = 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < cout * cin; e += 256) s_u[e] = (cscale * a.conv_weight[layer][e]) * s_style[e % cin];
    __syncthreads();
    const float *dW = a.dW[layer] + (size_t)b * cout * cin;
    if (tid < cout) {   // per row: r = max(||u||, eps); W' = u / r; du = (dW' - W' (W' . dW')) / r
        float ss = 0.0f;
        for (int c = 0; c < cin; ++c) ss += s_u[tid * cin + c] * s_u[tid * cin + c];
        const float r = fmaxf(sqrtf(ss), 1e-12f);
        float dot = 0.0f;
        for (int c = 0; c < cin; ++c) dot += (s_u[tid * cin + c] / r) * dW[tid * cin + c];
        s_rn[tid] = 1.0f / r;
        s_dot[tid] = dot;
    }
    __syncthreads();
    float *dcw = a.d_conv_weight[layer] + (size_t)b * cout * cin;
    for (int e = tid; e < cout * cin; e += 256) {
        const int o = e / cin, c = e % cin;
        const float du = (dW[e] - (s_u[e] * s_rn[o]) * s_dot[o]) * s_rn[o];
        dcw[e] = du * cscale * s_style[c];
        s_u[e] = du * cscale * a.conv_weight[layer][e];          // contribution to ds_c (reuse the buffer)
    }
    __syncthreads();
    if (tid < cin) {
        float ds = 0.0f;
        for (int o = 0; o < cout; ++o) ds += s_u[o * cin + tid];
        s_ds[tid] = ds;
        a.d_mod_bias[layer][(size_t)b * cin + tid] = ds;
    }
    __syncthreads();
    float *dmw = a.d_mod_weight[layer] + (size_t)b * cin * D;
    for (int e = tid; e < cin * D; e += 256) dmw[e] = s_ds[e / D] * z[e % D] * mscale;
    for (int d = tid; d < D; d += 256) {
        float acc = 0.0f;
        for (int c = 0; c < cin; ++c) acc += s_ds[c] * a.mod_weight[layer][(size_t)c * D + d];
        a.d_z_rend[((size_t)b * 3 + layer) * D + d] = acc * mscale;
    }
}

}  // namespace enarf

using namespace enarf;

extern "C" long long enarf_render_bwd_rows_per_image(int n, int Nf) {
    if (n <= 0 || Nf <= 0) return 0;
    return (long long)n * (Nf > 64 ? 128 : 64);     // 16 rows per valid fine tile: 4 tiles per ray up to Nf 64, 8 up to 128
}

extern "C" int enarf_render_bwd(const enarf_render_bwd_args *args, enarf_stream_t stream) {
    if (!args) return host::fail(ENARF_ERR_ARG, "enarf_render_bwd: args is null");
    const enarf_render_bwd_args &a = *args;
    if (a.B <= 0 || a.n <= 0 || a.P <= 0 || a.P > ENARF_MAX_PARTS || a.H <= 0 || a.W <= 0)
        return host::fail(ENARF_ERR_ARG, "enarf_render_bwd: bad sizes");
    if (a.H >= (1 << 23) || a.W >= (1 << 23)) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_render_bwd: plane side >= 2^23");
    if (a.Nf < 2 || a.Nf > kBwdMaxSamples) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_render_bwd: Nf=%d outside [2, %d]", a.Nf, kBwdMaxSamples);
    if (!a.image_coord || !a.inv_intrinsics || !a.parts || !a.canonical_pose || !a.feat_cl || !a.mask_planes || !a.mlp_pack ||
        !a.bins || !a.grad_feat_cl || !a.grad_mask_planes || !a.rows_x || !a.rows_h1 || !a.rows_h2 || !a.rows_dz1 ||
        !a.rows_dz2 || !a.rows_dz3 || !a.row_blocks || !a.workspace)
        return host::fail(ENARF_ERR_ARG, "enarf_render_bwd: null pointer");
    if (a.rows_per_image < enarf_render_bwd_rows_per_image(a.n, a.Nf))
        return host::fail(ENARF_ERR_ARG, "enarf_render_bwd: rows_per_image %lld < %lld", a.rows_per_image,
                          enarf_render_bwd_rows_per_image(a.n, a.Nf));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(a.row_blocks, 0, sizeof(unsigned int) * a.B, st);
    if (e != hipSuccess) return host::fail((int)e, "enarf_render_bwd: hipMemsetAsync failed: %s", hipGetErrorString(e));
    enarf_render_args f = {};
    f.B = a.B; f.n = a.n; f.P = a.P; f.Nc = 2; f.Nf = a.Nf; f.H = a.H; f.W = a.W;
    f.drop_invalid_rays = a.drop_invalid_rays;
    f.image_coord = a.image_coord; f.inv_intrinsics = a.inv_intrinsics; f.parts = a.parts;
    f.workspace = a.workspace;                    // no outputs: the set-up only writes records and the live list
    if (int rc = launch_ray_setup(f, st)) return rc;
    const int num_cus = device_cus();
    if (num_cus <= 0) return host::fail((int)hipGetLastError(), "enarf_render_bwd: cannot query the device");
    long long wgs = (long long)num_cus * kBwdWavesPerSimd;
    const long long total = (long long)a.B * a.n;
    if (wgs > total) wgs = total;
    if (a.Nf > 64) hipLaunchKernelGGL(render_bwd_kernel<2>, dim3((unsigned)wgs), dim3(256), (size_t)bwd_lds_floats(a.P) * 4, st, a);
    else hipLaunchKernelGGL(render_bwd_kernel<1>, dim3((unsigned)wgs), dim3(256), (size_t)bwd_lds_floats(a.P) * 4, st, a);
    return host::check_launch("enarf_render_bwd");
}

extern "C" long long enarf_query_bwd_rows_per_image(long long N) {
    if (N <= 0) return 0;
    return ((N + 15) / 16) * 16;       // one row per point, in 16-row tiles
}

extern "C" int enarf_query_bwd(const enarf_query_bwd_args *args, enarf_stream_t stream) {
    if (!args) return host::fail(ENARF_ERR_ARG, "enarf_query_bwd: args is null");
    const enarf_query_bwd_args &a = *args;
    if (a.B <= 0 || a.B > 65535 || a.N < 0 || a.P <= 0 || a.P > ENARF_MAX_PARTS || a.H <= 0 || a.W <= 0)
        return host::fail(ENARF_ERR_ARG, "enarf_query_bwd: bad sizes");
    if (a.H >= (1 << 23) || a.W >= (1 << 23)) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_query_bwd: plane side >= 2^23");
    if (!a.points || !a.parts || !a.canonical_pose || !a.feat_cl || !a.mask_planes || !a.mlp_pack || !a.grad_feat_cl ||
        !a.grad_mask_planes || !a.rows_x || !a.rows_h1 || !a.rows_h2 || !a.rows_dz1 || !a.rows_dz2 || !a.rows_dz3 || !a.row_blocks)
        return host::fail(ENARF_ERR_ARG, "enarf_query_bwd: null pointer");
    if (a.rows_per_image < enarf_query_bwd_rows_per_image(a.N))
        return host::fail(ENARF_ERR_ARG, "enarf_query_bwd: rows_per_image %lld < %lld", a.rows_per_image,
                          enarf_query_bwd_rows_per_image(a.N));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(a.row_blocks, 0, sizeof(unsigned int) * a.B, st);
    if (e != hipSuccess) return host::fail((int)e, "enarf_query_bwd: hipMemsetAsync failed: %s", hipGetErrorString(e));
    if (a.N == 0) return 0;
    const int num_cus = device_cus();
    if (num_cus <= 0) return host::fail((int)hipGetLastError(), "enarf_query_bwd: cannot query the device");
    const long long tiles = (a.N + 63) / 64;
    long long per_image = ((long long)num_cus * kBwdWavesPerSimd + a.B - 1) / a.B;
    if (per_image > tiles) per_image = tiles;
    hipLaunchKernelGGL(query_bwd_kernel, dim3((unsigned)per_image, a.B), dim3(256), (size_t)bwd_lds_floats(a.P) * 4, st, a, tiles);
    return host::check_launch("enarf_query_bwd");
}

extern "C" int enarf_triplane_unpack_add(const float *grad_feat_cl, float *grad_tri_nchw, int B, int channels_total,
                                         int H, int W, enarf_stream_t stream) {
    if (!grad_feat_cl || !grad_tri_nchw) return host::fail(ENARF_ERR_ARG, "enarf_triplane_unpack_add: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || channels_total < 3 * ENARF_FEAT_DIM || H > 65535 || B * 3 > 65535)
        return host::fail(ENARF_ERR_ARG, "enarf_triplane_unpack_add: bad sizes");
    hipLaunchKernelGGL(unpack_add_kernel, dim3((W + 63) / 64, H, B * 3), dim3(256), 0, (hipStream_t)stream, grad_feat_cl,
                       grad_tri_nchw, channels_total, H, W);
    return host::check_launch("enarf_triplane_unpack_add");
}

extern "C" size_t enarf_weight_grad_workspace_bytes(int B, long long rows_per_image) {
    if (B <= 0 || rows_per_image <= 0) return 0;
    const long long chunks = (rows_per_image + kWgRowsPerWg - 1) / kWgRowsPerWg;
    return (size_t)B * (size_t)chunks * kWgPartial * sizeof(float);
}

extern "C" int enarf_weight_grad(const enarf_weight_grad_args *args, enarf_stream_t stream) {
    if (!args) return host::fail(ENARF_ERR_ARG, "enarf_weight_grad: args is null");
    const enarf_weight_grad_args &a = *args;
    if (a.B <= 0 || a.B > 65535 || a.rows_per_image <= 0 || a.rows_per_image % 16 != 0)
        return host::fail(ENARF_ERR_ARG, "enarf_weight_grad: bad sizes (B=%d rows_per_image=%lld)", a.B, a.rows_per_image);
    if (!a.rows_x || !a.rows_h1 || !a.rows_h2 || !a.rows_dz1 || !a.rows_dz2 || !a.rows_dz3 || !a.row_blocks || !a.workspace ||
        !a.dW1 || !a.dW2 || !a.dW3 || !a.db1 || !a.db2 || !a.db3)
        return host::fail(ENARF_ERR_ARG, "enarf_weight_grad: null pointer");
    WeightGradParams p;
    p.x = a.rows_x; p.h1 = a.rows_h1; p.h2 = a.rows_h2; p.dz1 = a.rows_dz1; p.dz2 = a.rows_dz2; p.dz3 = a.rows_dz3;
    p.rows_per_image = a.rows_per_image; p.row_blocks = a.row_blocks;
    p.partial = reinterpret_cast<float *>(a.workspace);
    const long long chunks = (a.rows_per_image + kWgRowsPerWg - 1) / kWgRowsPerWg;
    if (chunks > 0x3FFFFFFFll) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_weight_grad: too many rows");
    p.chunks = (int)chunks;
    p.dW1 = a.dW1; p.dW2 = a.dW2; p.dW3 = a.dW3; p.db1 = a.db1; p.db2 = a.db2; p.db3 = a.db3;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(weight_grad_partial_kernel, dim3((unsigned)chunks, a.B), dim3(256), 0, st, p);
    if (int rc = host::check_launch("enarf_weight_grad(partial)")) return rc;
    hipLaunchKernelGGL(weight_grad_reduce_kernel, dim3((64 * 64 + 64 * 32 + 4 * 64 + 132 + 63) / 64, a.B), dim3(256), 0, st, p);
    return host::check_launch("enarf_weight_grad(reduce)");
}

extern "C" int enarf_prepare_bwd(const enarf_prepare_bwd_args *args, enarf_stream_t stream) {
    if (!args) return host::fail(ENARF_ERR_ARG, "enarf_prepare_bwd: args is null");
    const enarf_prepare_bwd_args &a = *args;
    if (a.B <= 0 || a.B > 65535 || a.style_dim <= 0 || !a.z_rend || !a.d_z_rend)
        return host::fail(ENARF_ERR_ARG, "enarf_prepare_bwd: bad sizes or null pointer");
    for (int i = 0; i < 3; ++i)
        if (!a.conv_weight[i] || !a.mod_weight[i] || !a.mod_bias[i] || !a.dW[i] || !a.d_conv_weight[i] || !a.d_mod_weight[i] ||
            !a.d_mod_bias[i])
            return host::fail(ENARF_ERR_ARG, "enarf_prepare_bwd: null pointer (layer %d)", i);
    hipLaunchKernelGGL(prepare_bwd_kernel, dim3(3, a.B), dim3(256), 0, (hipStream_t)stream, a);
    return host::check_launch("enarf_prepare_bwd");
}
