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

#ifndef ENARF_BWD_PARITY_WALK          // 0 = round 2's walkers by tap slot (A/B)
#define ENARF_BWD_PARITY_WALK 1
#endif
#ifndef ENARF_BWD_NS_FIXED
#define ENARF_BWD_NS_FIXED 0
#endif
#ifndef ENARF_BWD_ABLATE
#define ENARF_BWD_ABLATE 0      // diagnosis builds only: 1 no feature atomics, 2 no mask atomics, 4 no scatter pass, 8 no row export
#endif

namespace enarf {

constexpr int kBwdWavesPerSimd = 2;

// LDS scratch of the backward kernel (floats). NS = stride of the per-sample arrays: 64 for Nf <= 64 and for the point
// kernel, 128 for 64 < Nf <= 128. Sized by NS, not by the maximum: with P = 23 the fixed 128-sample layout came to 82.1 KB
// per workgroup - 256 B too many for two workgroups on a CU's 160 KB, i.e. ONE wave per SIMD for the whole backward.
constexpr int kBwdMaxSamples = 128;
constexpr int SB_CAND = 0;                                  // 4 waves x 32 ints
constexpr int SB_FH = 128;                                  // head [4][NS]
__host__ __device__ constexpr int sb_fbits(int NS) { return SB_FH + 4 * NS; }          // bits [NS]
__host__ __device__ constexpr int sb_dz3(int NS) { return sb_fbits(NS) + NS; }         // dL/dz3 [4][NS]
__host__ __device__ constexpr int sb_wmax(int NS) { return sb_dz3(NS) + 4 * NS; }      // multiply_density_with_triplane_wieght: max part weight [NS]
__host__ __device__ constexpr int sb_kmax(int NS) { return sb_wmax(NS) + NS; }         //   the part that attains it [NS] (signed bytes: -1 .. 31;
                                                                                       //   as ints the 128-sample layout was 192 B over two workgroups per CU)
__host__ __device__ constexpr int sb_gwm(int NS) { return sb_kmax(NS) + NS / 4; }      //   dL/d(max weight) [NS]
__host__ __device__ constexpr int sb_queue(int NS) { return sb_gwm(NS) + NS; }         // 2 ray slots
__host__ __device__ constexpr int bwd_scratch_floats(int NS) { return sb_queue(NS) + 64; }
static_assert(kQueueLdsInts <= 64, "scratch overflow");

constexpr int kTRow = 33;                       // floats per texel row of the transpose tile (32 + 1 pad)
constexpr int kTTile = 2 * 16 * kTRow + 32;      // two taps x (16 texel rows) + 2 x 16 texel offsets (ints)
__host__ __device__ inline int bwd_lds_floats(int P, int NS) {
    return PK_B1 + 144 + PKT_FLOATS + P * kLdsPartStride + P * kLdsCanonStride + bwd_scratch_floats(NS) + 4 * kTTile;
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
                                             bool on, const float dxg[8], int lane, unsigned &n_lines) {
    const int q = lane >> 2, g = lane & 3;
    int *toff = reinterpret_cast<int *>(tile + 2 * 16 * kTRow);
    // Which half-wave walks which tap: by the PARITY of the texel offset, not by the tap slot. The two taps of a row are
    // x0 and x0 + 1 - their offsets always differ in parity - so every occurrence of a texel lands with the same walker, and a
    // sample's x0 + 1 meeting the next sample's x0 (the ray advanced by one texel) is one run instead of two atomics:
    // 686 -> 613 B of feature lines per valid pair (tests/analysis/atomic_merge.py), what the 6x6 LDS window bought at 2-3x the time.
    if (ENARF_BWD_PARITY_WALK && (offA & 1)) {
        const int to = offA; offA = offB; offB = to;
        const float tc = cfA; cfA = cfB; cfB = tc;
    }
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
                if (run_o >= 0 && !(ENARF_BWD_ABLATE & 1)) { atomicAdd(gpl + (size_t)run_o * kFeat + ch, run_v); n_lines += 1; }
                run_o = o;
                run_v = v;
            }
        }
        if (run_o >= 0 && !(ENARF_BWD_ABLATE & 1)) { atomicAdd(gpl + (size_t)run_o * kFeat + ch, run_v); n_lines += 1; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void scatter_plane(float *__restrict__ gpl, float *tile, const Taps &t, float wk, bool on,
                                              const float dxg[8], int lane, unsigned &n_lines) {
    // The two walks of a plane are dealt by the PARITY of the texel ROW (the footprint's rows y0 and y0 + 1 always differ in
    // it), as the two walkers inside a walk are dealt by the parity of the offset: each of the footprint's four texels then
    // has ONE walker that sees every occurrence of it, and along a ray - which crosses texels monotonically - those are
    // consecutive samples: run merging alone removes every duplicate of the tile (tests/analysis/atomic_merge.py: "tile
    // uniq"), with no LDS table.
    const bool up = !ENARF_BWD_PARITY_WALK || t.yp == 0;      // the upper row (y0) is the even one
    scatter_tap2(gpl, tile, up ? t.o00 : t.o10, (up ? t.w00 : t.w10) * wk, up ? t.o01 : t.o11, (up ? t.w01 : t.w11) * wk, on, dxg, lane, n_lines);
    scatter_tap2(gpl, tile, up ? t.o10 : t.o00, (up ? t.w10 : t.w00) * wk, up ? t.o11 : t.o01, (up ? t.w11 : t.w01) * wk, on, dxg, lane, n_lines);
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
__device__ __forceinline__ void mask_tap_add(float *__restrict__ gmask, bool on, int elem, float v, int lane, unsigned &n_adds) {
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
    if (on && tail) { atomicAdd(gmask + elem, acc); n_adds += 1; }
}

// A/B (ENARF_BWD_MASK_ROWMAJOR=1): the same adds with the wave re-laid as lane = 32 row + 2 sample + x-tap, so that all
// taps of one texel row sit in consecutive lanes (one permute each for the element and the value), runs of equal elements
// merged along the samples (lanes 2 apart, rows of 8 samples) - does the memory side then see one request per touched
// 64-byte segment instead of one per sample and row?
#ifndef ENARF_BWD_F4_ROUNDS            // A/B: rounds of the scatter pass whose loads are all issued before any of their atomics.
#define ENARF_BWD_F4_ROUNDS 1          // 1 = loads and adds alternate round by round. Measured at C1: 1 -> 1.628 ms, 4 -> 1.660, 8 -> 1.653
#endif
#ifndef ENARF_BWD_MASK_ROWMAJOR
#define ENARF_BWD_MASK_ROWMAJOR 0
#endif
__device__ __forceinline__ void mask_tap_add_rowmajor(float *__restrict__ gmask, bool on, int elem, float v, int lane, unsigned &n_adds) {
    // source lane (sample q, tap t = 2 row + x) = 4 q + t  ->  lane 32 row + 2 q + x
    const int src = 4 * ((lane >> 1) & 15) + 2 * (lane >> 5) + (lane & 1);
    const int key_own = on ? elem : -2 - lane;
    const int key = __builtin_amdgcn_ds_bpermute(src << 2, key_own);
    float acc = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, on ? v : 0.0f)));
    const bool live = key >= 0;
    auto shr = [](int old, int x, int sel) {
        return sel == 0 ? __builtin_amdgcn_update_dpp(old, x, 0x112, 0xF, 0xF, false)       // row_shr:2
             : sel == 1 ? __builtin_amdgcn_update_dpp(old, x, 0x114, 0xF, 0xF, false)       // row_shr:4
                        : __builtin_amdgcn_update_dpp(old, x, 0x118, 0xF, 0xF, false);      // row_shr:8
    };
    int head = (shr(-1, key, 0) != key) ? 1 : 0;
#pragma unroll
    for (int sel = 0; sel < 3; ++sel) {
        const float pv = __builtin_bit_cast(float, shr(0, __builtin_bit_cast(int, acc), sel));
        const int ph = shr(1, head, sel);
        if (!head) { acc += pv; head |= ph; }
    }
    const bool tail = __builtin_amdgcn_update_dpp(-1, key, 0x102, 0xF, 0xF, false) != key;      // row_shl:2: the next sample starts another run
    if (live && tail) { atomicAdd(gmask + key, acc); n_adds += 1; }
}

// what the per-tile stages of the backward need besides the query context (shared by the ray and the point kernels)
struct BwdTile {
    int H, W;
    size_t mplane, fplane;
    const float *l_wt;                    // LDS: transposed weight section
    float *rows_x, *rows_dz3;             // compact rows for enarf_weight_grad: the tile's features and dL/dz3
    long long rows_per_image;
    unsigned int *row_blocks;
    float *gfeat, *gmask;                 // this image's gradient planes
    float *ttile;                         // this wave's atomic-transpose tile (LDS)
};
// per-lane tallies of a wave for enarf_render_bwd_args.counters (summed over the wave when the kernel ends)
struct BwdCount {
    unsigned pairs, tiles, rounds;        // wave-uniform
    unsigned lines, mask_adds;            // per lane: lines counts in lanes 0 and 32 only (one per 128-B half-wave atomic)
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
                                                  float dz3v, int lane, BwdCount &C, float gwm = 0.0f, int kmax = -1) {
    const int g4 = lane & 3, j4 = lane >> 2;
    const int mj = lane & 15, mg = lane >> 4;          // MFMA layout: point mj, k-group mg
    f32x4 dz2[4], dz1[4];
    float dxm[8];
    mlp_bwd_tile_f32(T.l_wt, a1, a2, dz3v, lane, dz2, dz1, dxm);
    // compact rows for the weight gradients (enarf_weight_grad re-runs the MLP on them): block of 16 rows of image b
    unsigned int blk = 0;
    if (lane == 0) blk = atomicAdd(T.row_blocks + b, 1u);
    blk = (unsigned int)__builtin_amdgcn_readfirstlane((int)blk);
    const size_t row = (size_t)b * T.rows_per_image + (size_t)blk * 16 + mj;
    if (!(ENARF_BWD_ABLATE & 8)) {
        f32x4 *rx = reinterpret_cast<f32x4 *>(T.rows_x + row * 32 + 8 * mg);
        rx[0] = f32x4{x[0], x[1], x[2], x[3]};
        rx[1] = f32x4{x[4], x[5], x[6], x[7]};
        T.rows_dz3[row * 4 + mg] = dz3v;
    }
    C.tiles += 1;
    // d feature back in the gather layout
    float dxg[8];
    const int src2 = (g4 << 4) | j4;
#pragma unroll
    for (int c = 0; c < 8; ++c) dxg[c] = __shfl(dxm[c], src2);

    // ---- F4: second pass over the pairs: d part-probability and d feature texels.
    // In chunks of kF4Rounds rounds, each chunk in TWO sub-passes: (a) everything that LOADS - the part-probability taps and
    // the 12 feature texels of every pair, for dL/d part weight - then (b) everything that ADDS; (b) recomputes a round's taps
    // from the frames in LDS (VALU only) and keeps two floats per round from (a). The idea behind chunks > 1: on gfx9 loads
    // and no-return atomics share vmcnt, so a wait for a load result while atomics are outstanding is s_waitcnt vmcnt(0) and
    // drains every atomic the wave has in flight; alternating round by round that is one drain per round, in chunks one per
    // chunk. Measured (profiles/r03_bwd_lds_merge_ab.log): no gain - 1.628 ms (1), 1.660 (4), 1.653 (8) at C1: the waves are
    // not waiting on those drains. The product keeps chunks of 1.
    constexpr int kF4Rounds = ENARF_BWD_F4_ROUNDS;
    auto round_taps = [&](int k) {      // this lane's own-plane taps of part k at the sample (lane 3 repeats plane 0)
        float F[13], Cn[12], lx, ly, lz, cx, cy, cz;
        load_frames(S, k, F, Cn);
        exact_local(F, px, py, pz, lx, ly, lz);
        exact_canonical(Cn, F[12], lx, ly, lz, cx, cy, cz);
        const float qx = (g4 == 1) ? cy : (g4 == 2) ? cz : cx;
        const float qy = (g4 == 1) ? cz : (g4 == 2) ? cx : cy;
        return make_taps(qx, qy, T.H, T.W);
    };
    uint32_t rem = (ENARF_BWD_ABLATE & 4) ? 0u : bits;
    __builtin_amdgcn_s_setprio(2);          // memory rounds ahead of the other waves' MFMA phases (as in the forward)
    while (__ballot(rem != 0) != 0) {
        float wk_r[kF4Rounds], gm_r[kF4Rounds];
        // ---- (a) loads
        uint32_t rem_a = rem;
#pragma unroll
        for (int r = 0; r < kF4Rounds; ++r) {
            wk_r[r] = 0.0f; gm_r[r] = 0.0f;
            const uint64_t bal = __ballot(rem_a != 0);
            if (bal != 0) {
                C.pairs += (unsigned)(__popcll(bal) >> 2);
                C.rounds += 1;
                const bool act = rem_a != 0;
                const int k = act ? __builtin_ctz(rem_a) : 0;
                rem_a &= rem_a - 1;
                const Taps t = round_taps(k);
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
                wk_r[r] = (S.uniform_w > 0.0f) ? S.uniform_w : wp;
                // w = s0 s1 s2, s = sigmoid(m): dw/dm_p = w (1 - s_p) for plane p (lane p of the quad)
                gm_r[r] = dot * wp * (1.0f - sg);
            }
        }
        // ---- (b) adds: no load result is needed from here to the end of the chunk
#pragma unroll
        for (int r = 0; r < kF4Rounds; ++r) {
            if (__ballot(rem != 0) != 0) {
                const bool act = rem != 0;
                const int k = act ? __builtin_ctz(rem) : 0;
                rem &= rem - 1;
                const Taps t = round_taps(k);
                const float wk = wk_r[r], gm = gm_r[r];
                {   // One wave-instruction per PLANE with lane = (quad, tap): the two taps of a row are adjacent floats in
                    // adjacent lanes, so they leave as ONE 64-byte request at the memory side, where the atomics are counted and
                    // paid for (round 2 issued one instruction per tap slot with lane = (quad, plane): every add its own request -
                    // 8.3 M of the 20.4 M requests of a C1 backward, profiles/r03_bwd_a_pmc_summary.txt). Uniform weights
                    // (no_selector): no plane gradient
                    const bool mon = act && !(ENARF_BWD_ABLATE & 2) && !(S.uniform_w > 0.0f);
#define ENARF_MASK_PLANE(PL)                                                                                                   \
                    {   /* all broadcasts run with the whole quad enabled (a DPP read of a disabled lane returns nothing) */      \
                        const int o0 = quad_bcast_i<PL>(t.o00), o1 = quad_bcast_i<PL>(t.o01), o2 = quad_bcast_i<PL>(t.o10), o3 = quad_bcast_i<PL>(t.o11); \
                        const float w0 = quad_bcast_f<PL>(t.w00), w1 = quad_bcast_f<PL>(t.w01), w2 = quad_bcast_f<PL>(t.w10), w3 = quad_bcast_f<PL>(t.w11); \
                        const float gp = quad_bcast_f<PL>(gm);                                                                 \
                        /* lane role = (row parity g4 >> 1, offset parity g4 & 1): the footprint texel with those parities - every  \
                           occurrence of a texel meets the same lane role, so runs along the samples merge ALL its duplicates */      \
                        const int ypb = quad_bcast_i<PL>(t.yp);                                                                \
                        const bool r1 = ENARF_BWD_PARITY_WALK ? (ypb != (g4 >> 1)) : (g4 >> 1) != 0;                            \
                        const int oA = r1 ? o2 : o0, oB = r1 ? o3 : o1;                                                        \
                        const float wA = r1 ? w2 : w0, wB = r1 ? w3 : w1;                                                      \
                        const bool tb = ENARF_BWD_PARITY_WALK ? ((oA & 1) != (g4 & 1)) : (g4 & 1) != 0;                         \
                        const int o = tb ? oB : oA;                                                                            \
                        const float w = tb ? wB : wA;                                                                          \
                        if (ENARF_BWD_MASK_ROWMAJOR)                                                                           \
                            mask_tap_add_rowmajor(T.gmask, mon && w != 0.0f, (3 * k + PL) * (int)T.mplane + o, w * gp, lane, C.mask_adds); \
                        else                                                                                                   \
                            mask_tap_add(T.gmask, mon && w != 0.0f, (3 * k + PL) * (int)T.mplane + o, w * gp, lane, C.mask_adds); \
                    }
                    ENARF_MASK_PLANE(0)
                    ENARF_MASK_PLANE(1)
                    ENARF_MASK_PLANE(2)
#undef ENARF_MASK_PLANE
                }
                // every lane takes part (wave-uniform): inactive quads contribute empty rows
                Taps t0 = quad_bcast_taps<0>(t), t1 = quad_bcast_taps<1>(t), t2 = quad_bcast_taps<2>(t);
                t0.yp = quad_bcast_i<0>(t.yp); t1.yp = quad_bcast_i<1>(t.yp); t2.yp = quad_bcast_i<2>(t.yp);
                scatter_plane(T.gfeat, T.ttile, t0, wk, act, dxg, lane, C.lines);
                scatter_plane(T.gfeat + T.fplane, T.ttile, t1, wk, act, dxg, lane, C.lines);
                scatter_plane(T.gfeat + 2 * T.fplane, T.ttile, t2, wk, act, dxg, lane, C.lines);
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
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
#if ENARF_BWD_NS_FIXED                               // A/B only: round 2's fixed 128-sample layout (one workgroup per CU)
    constexpr int NS = kBwdMaxSamples;
#else
    constexpr int NS = 64 * SPL;                    // stride of the per-sample LDS arrays
#endif
    int *l_q = reinterpret_cast<int *>(scratch + sb_queue(NS));
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
    float *l_fh = scratch + SB_FH, *l_dz3 = scratch + sb_dz3(NS);
    float *l_wmax = scratch + sb_wmax(NS), *l_gwm = scratch + sb_gwm(NS);
    signed char *l_kmax = reinterpret_cast<signed char *>(scratch + sb_kmax(NS));
    float *ttile = scratch + bwd_scratch_floats(NS) + wave * kTTile;     // this wave's atomic-transpose tile
    uint32_t *l_fbits = reinterpret_cast<uint32_t *>(scratch + sb_fbits(NS));
    const int j4 = lane >> 2, g4 = lane & 3;
    const size_t mplane = (size_t)a.H * a.W, fplane = mplane * kFeat;
    int qslot = 0;
    BwdTile T;
    T.H = a.H; T.W = a.W; T.mplane = mplane; T.fplane = fplane; T.l_wt = l_wt;
    T.rows_x = a.rows_x; T.rows_dz3 = a.rows_dz3;
    T.rows_per_image = a.rows_per_image; T.row_blocks = a.row_blocks; T.ttile = ttile;
    T.gfeat = nullptr; T.gmask = nullptr;
    BwdCount C = {0u, 0u, 0u, 0u, 0u};
    unsigned c_rays = 0;

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
            if (i < Nf && g4 == 0) { l_fbits[i] = active ? bits[u] : 0u; l_wmax[i] = wmx; l_kmax[i] = (signed char)kmx; }
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
            bwd_backward_tile(S, T, b, px[u], py[u], pz[u], bits[u], a1, a2, x, dz3v, lane, C, l_gwm[si], (int)l_kmax[si]);
        }
        c_rays += 1;
        // the next ray's first barrier orders this ray's LDS reads (l_dz3, l_fh) before their next writes
        __syncthreads();
        cur = next_ray;
    }
    if (a.counters) {
        const float nl = wave_sum((lane & 31) == 0 ? (float)C.lines : 0.0f), nm = wave_sum((float)C.mask_adds);   // < 2^24 per wave
        if (lane == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)C.pairs);
            atomicAdd(&a.counters[1], (unsigned long long)C.tiles);
            if (wave == 0) atomicAdd(&a.counters[2], (unsigned long long)c_rays);
            atomicAdd(&a.counters[3], (unsigned long long)nl);
            atomicAdd(&a.counters[4], (unsigned long long)nm);
            atomicAdd(&a.counters[5], (unsigned long long)C.rounds);
        }
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
    constexpr int NS = 64;
    float *l_fh = scratch + SB_FH, *l_dz3 = scratch + sb_dz3(NS);
    float *l_wmax = scratch + sb_wmax(NS), *l_gwm = scratch + sb_gwm(NS);
    signed char *l_kmax = reinterpret_cast<signed char *>(scratch + sb_kmax(NS));
    uint32_t *l_fbits = reinterpret_cast<uint32_t *>(scratch + sb_fbits(NS));
    BwdTile T;
    T.H = a.H; T.W = a.W; T.mplane = (size_t)a.H * a.W; T.fplane = T.mplane * kFeat; T.l_wt = l_wt;
    T.rows_x = a.rows_x; T.rows_dz3 = a.rows_dz3;
    T.rows_per_image = a.rows_per_image; T.row_blocks = a.row_blocks;
    BwdCount C = {0u, 0u, 0u, 0u, 0u};
    T.ttile = scratch + bwd_scratch_floats(NS) + wave * kTTile;
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
        if (g4 == 0) { l_fbits[wave * 16 + j4] = active ? bits : 0u; l_wmax[wave * 16 + j4] = wmx; l_kmax[wave * 16 + j4] = (signed char)kmx; }
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
            bwd_backward_tile(S, T, b, px, py, pz, bits, a1, a2, x, dz3v, lane, C, l_gwm[wave * 16 + j4], (int)l_kmax[wave * 16 + j4]);
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

// ---- weight gradients from the compact rows: dW_l = dZ_l^T H_{l-1}, db_l = sum_rows dZ_l ------------------------------
// The backward kernels export, per valid 16-sample tile, only what cannot be recomputed from the MLP alone: the 32 gathered
// features x and dL/dz3 (144 B per sample; all six activation rows were 1 168 B: 589 MB per C1 frame written and read back,
// 1.2 GB of scratch). Here every wave takes its tiles through the exact-fp32 MLP forward (x -> h1, h2) and backward
// (dz3 -> dz2, dz1) again - mlp_tile_f32_keep / mlp_bwd_tile_f32, the code the exporting kernel ran on the same operands -
// and accumulates the three products over ALL its tiles in 112 accumulator registers.
// The products contract over POINTS, which the MLP's layout keeps on the lane axis (lane = 16 g + point, registers =
// units 16 ob + 4 g + r): each 16 x 16 block is transposed through a wave-private LDS tile (4 ds_write_b32, 1 ds_read_b128)
// into "lane = 16 kk + unit, register s = point 4 kk + s", which is both the A operand (rows = units of dZ) and the B operand
// (columns = units of H) of v_mfma_f32_16x16x4_f32 with k-step s contracting points {s, 4 + s, 8 + s, 12 + s}.
// A fixed number of workgroups per image takes an even share of the image's tiles each and stores ONE partial; a second
// kernel sums the partials in a fixed order. No atomics: deterministic for a given row order.
constexpr int kWgAcc2 = 0, kWgAcc1 = 16 * 256, kWgAcc3 = kWgAcc1 + 8 * 256, kWgSum2 = kWgAcc3 + 4 * 256;
constexpr int kWgSum1 = kWgSum2 + 64, kWgSum3 = kWgSum1 + 64, kWgPartial = kWgSum3 + 16;      // floats per partial
constexpr int kWgTStride = 20;                            // floats per unit row of the transpose tile (16 points + pad; rows stay 16-B aligned)
constexpr int kWgTTile = 4 * 16 * kWgTStride;             // one whole activation (64 units x 16 points) per wave
constexpr int kWgLdsFloats = PK_B1 + 144 + PKT_FLOATS + 4 * kWgTTile;
static_assert(kWgPartial <= PK_B1 + 144, "the partial is staged over the weights");

struct WeightGradParams {
    const float *x, *dz3;
    const void *pack;
    long long rows_per_image;
    const unsigned int *row_blocks;
    float *partial;            // [B][groups][kWgPartial]
    int groups;                // workgroups per image
    float *dW1, *dW2, *dW3, *db1, *db2, *db3;
};

// E (4 blocks of f32x4: lane 16 g + j holds units 16 blk + 4 g + r of point j) -> Et[blk][s] = E[unit 16 blk + (lane & 15)][point 4 (lane >> 4) + s]
__device__ __forceinline__ void wg_transpose(float *tt, const f32x4 e[4], f32x4 et[4], int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) tt[(blk * 16 + 4 * g + r) * kWgTStride + j] = e[blk][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) et[blk] = *reinterpret_cast<const f32x4 *>(tt + (blk * 16 + j) * kWgTStride + 4 * g);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sum over the 16 lanes of a row (lanes 16 g .. 16 g + 15), result in the row's last lane
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0x111, 0xF>(v);
    v += dpp_f<0x112, 0xF>(v);
    v += dpp_f<0x114, 0xF>(v);
    v += dpp_f<0x118, 0xF>(v);
    return v;
}

__global__ __launch_bounds__(256, 1) void weight_grad_partial_kernel(const WeightGradParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const int grp = blockIdx.x;
    // the image's 16-row tiles are dealt evenly: workgroup grp of `groups` takes tiles [grp T / groups, (grp + 1) T / groups),
    // its four waves a quarter each (a wave runs ~10 k cycles of MFMA per tile, so the launch is as long as its most loaded
    // wave: fixed 1024-row chunks gave 460 busy workgroups for 256 CUs at C1, i.e. two rounds, 0.233 ms; even shares: one)
    const long long tiles = (long long)p.row_blocks[b];
    if (tiles == 0) return;                                   // block-uniform: the image exported nothing
    const long long t_lo = tiles * grp / p.groups, t_hi = tiles * (grp + 1) / p.groups;
    float *l_w = lds, *l_bias = l_w + PK_B1, *l_wt = l_bias + 144, *tt = l_wt + PKT_FLOATS + wave * kWgTTile;
    {
        const float *pf = reinterpret_cast<const float *>(reinterpret_cast<const char *>(p.pack) + (size_t)b * kPackBytes);
        const float *pt = reinterpret_cast<const float *>(reinterpret_cast<const char *>(pf) + kPackTOff);
        for (int i = tid; i < PK_B1 / 4; i += 256) reinterpret_cast<f32x4 *>(l_w)[i] = reinterpret_cast<const f32x4 *>(pf)[i];
        for (int i = tid; i < 144; i += 256) l_bias[i] = pf[PK_B1 + i];
        for (int i = tid; i < PKT_FLOATS / 4; i += 256) reinterpret_cast<f32x4 *>(l_wt)[i] = reinterpret_cast<const f32x4 *>(pt)[i];
    }
    __syncthreads();
    const int j = lane & 15, g = lane >> 4;
    const size_t base = (size_t)b * p.rows_per_image;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 acc2[16], acc1[8], acc3[4], s2[4], s1[4];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc2[t] = zero;
#pragma unroll
    for (int t = 0; t < 8; ++t) acc1[t] = zero;
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc3[t] = zero; s2[t] = zero; s1[t] = zero; }
    float s3 = 0.0f;
    {
        const long long nt = t_hi - t_lo;
        const long long r0 = 16 * (t_lo + nt * wave / 4), r1 = 16 * (t_lo + nt * (wave + 1) / 4);
        for (long long r = r0; r < r1; r += 16) {
            const size_t row0 = base + (size_t)r;
            float x[8];
            {
                const f32x4 *xr = reinterpret_cast<const f32x4 *>(p.x + (row0 + j) * 32 + 8 * g);
                const f32x4 v0 = xr[0], v1 = xr[1];
#pragma unroll
                for (int c = 0; c < 4; ++c) { x[c] = v0[c]; x[4 + c] = v1[c]; }
            }
            const float dz3v = p.dz3[(row0 + j) * 4 + g];
            // the same features and the same dL/dz3 in the operand layout of the products (lane = 16 kk + column, register s = point 4 kk + s)
            f32x4 xt[2], dz3t;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const size_t rr = row0 + 4 * g + sidx;
                xt[0][sidx] = p.x[rr * 32 + j];
                xt[1][sidx] = p.x[rr * 32 + 16 + j];
                dz3t[sidx] = (j < 4) ? p.dz3[rr * 4 + j] : 0.0f;
            }
            f32x4 a1[4], a2[4], o, dz2[4], dz1[4];
            float dxm[8];
            mlp_tile_f32_keep(l_w, l_bias, x, lane, a1, a2, o);
            mlp_bwd_tile_f32<false>(l_wt, a1, a2, dz3v, lane, dz2, dz1, dxm);
            (void)o; (void)dxm;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) { s2[ob] += dz2[ob]; s1[ob] += dz1[ob]; }
            s3 += dz3v;
            f32x4 ta[4], tb[4];
            // dW3 = dZ3^T H2 (rows 4..15 of the A operand are zero)
            wg_transpose(tt, a2, tb, lane);
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc3[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz3t[sidx], tb[cb][sidx], acc3[cb], 0, 0, 0);
            // dW2 = dZ2^T H1
            wg_transpose(tt, dz2, ta, lane);
            wg_transpose(tt, a1, tb, lane);
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc2[mb * 4 + cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[mb][sidx], tb[cb][sidx], acc2[mb * 4 + cb], 0, 0, 0);
            // dW1 = dZ1^T X
            wg_transpose(tt, dz1, ta, lane);
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc1[mb * 2 + cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[mb][sidx], xt[cb][sidx], acc1[mb * 2 + cb], 0, 0, 0);
        }
    }
    // bias gradients: sum over the 16 points of a lane row; the row's last lane holds units 16 ob + 4 g + r
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s2[ob][r] = row16_sum(s2[ob][r]); s1[ob][r] = row16_sum(s1[ob][r]); }
    s3 = row16_sum(s3);
    // the four waves add up in LDS (over the weights, which nobody needs any more), one after the other, then store once
    __syncthreads();
    float *red = lds;
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
            if (j == 15) {
#pragma unroll
                for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        put(kWgSum2 + 16 * ob + 4 * g + r, s2[ob][r]);
                        put(kWgSum1 + 16 * ob + 4 * g + r, s1[ob][r]);
                    }
                put(kWgSum3 + g, s3);
            }
        }
        __syncthreads();
    }
    float *out = p.partial + ((size_t)b * p.groups + grp) * kWgPartial;
    for (int i = tid; i < kWgPartial; i += 256) out[i] = red[i];
}

// four threads per output element (64 elements per block): each sums every fourth partial, then the four add up through LDS
__global__ __launch_bounds__(256) void weight_grad_reduce_kernel(const WeightGradParams p) {
    __shared__ float red[256];
    const int el = threadIdx.x & 63, part4 = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el, b = blockIdx.y;
    constexpr int N2 = 64 * 64, N1 = 64 * 32, N3 = 4 * 64, NB = 64 + 64 + 4;
    const bool in_range = e < N2 + N1 + N3 + NB;
    const int nwg = p.row_blocks[b] ? p.groups : 0;       // every workgroup of an image that exported rows stored a partial
    int idx = 0;
    float *dst = nullptr;
    // accumulator (mb, cb), register r, lane l  <->  row 16 mb + 4 (l >> 4) + r, column 16 cb + (l & 15)
    if (e < N2) {
        const int m = e / 64, c = e % 64;
        idx = kWgAcc2 + (((m >> 4) * 4 + (c >> 4)) * 4 + (m & 3)) * 64 + 16 * ((m & 15) >> 2) + (c & 15);
        dst = p.dW2 + (size_t)b * N2 + e;
    } else if (e < N2 + N1) {
        const int f = e - N2, m = f / 32, c = f % 32;
        idx = kWgAcc1 + (((m >> 4) * 2 + (c >> 4)) * 4 + (m & 3)) * 64 + 16 * ((m & 15) >> 2) + (c & 15);
        dst = p.dW1 + (size_t)b * N1 + f;
    } else if (e < N2 + N1 + N3) {      // rows 0..3 of the (zero-padded) 16-row product: lanes 0..15, register o
        const int f = e - N2 - N1, o = f / 64, c = f % 64;
        idx = kWgAcc3 + ((c >> 4) * 4 + o) * 64 + (c & 15);
        dst = p.dW3 + (size_t)b * N3 + f;
    } else if (in_range) {
        const int f = e - N2 - N1 - N3;
        if (f < 64) { idx = kWgSum1 + f; dst = p.db1 + (size_t)b * 64 + f; }
        else if (f < 128) { idx = kWgSum2 + f - 64; dst = p.db2 + (size_t)b * 64 + f - 64; }
        else { idx = kWgSum3 + f - 128; dst = p.db3 + (size_t)b * 4 + f - 128; }
    }
    const float *part = p.partial + (size_t)b * p.groups * kWgPartial;
    float acc = 0.0f;
    if (in_range)
        for (int w = part4; w < nwg; w += 4) acc += part[(size_t)w * kWgPartial + idx];
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

// ray set-up (own launch, epoch 0) + the backward kernel for one launch group
static int launch_bwd_group(const enarf_render_bwd_args &a, hipStream_t st) {
    enarf_render_args f = {};
    f.B = a.B; f.n = a.n; f.P = a.P; f.Nc = 2; f.Nf = a.Nf; f.H = a.H; f.W = a.W;
    f.drop_invalid_rays = a.drop_invalid_rays;
    f.image_coord = a.image_coord; f.inv_intrinsics = a.inv_intrinsics; f.parts = a.parts;
    f.workspace = a.workspace;                    // no outputs: the set-up only writes records and the live list
    f.near_far = a.near_far;
    if (int rc = launch_ray_setup(f, st)) return rc;
    const int num_cus = device_cus();
    if (num_cus <= 0) return host::fail((int)hipGetLastError(), "enarf_render_bwd: cannot query the device");
    long long wgs = (long long)num_cus * kBwdWavesPerSimd;
    const long long total = (long long)a.B * a.n;
    if (wgs > total) wgs = total;
    if (a.Nf > 64) hipLaunchKernelGGL(render_bwd_kernel<2>, dim3((unsigned)wgs), dim3(256), (size_t)bwd_lds_floats(a.P, 128) * 4, st, a);
    else hipLaunchKernelGGL(render_bwd_kernel<1>, dim3((unsigned)wgs), dim3(256), (size_t)bwd_lds_floats(a.P, ENARF_BWD_NS_FIXED ? 128 : 64) * 4, st, a);
    return host::check_launch("enarf_render_bwd");
}

extern "C" int enarf_render_bwd(const enarf_render_bwd_args *args, enarf_stream_t stream) {
    if (!args) return host::fail(ENARF_ERR_ARG, "enarf_render_bwd: args is null");
    const enarf_render_bwd_args &a = *args;
    if (a.B <= 0 || a.n <= 0 || a.P <= 0 || a.P > ENARF_MAX_PARTS || a.H <= 0 || a.W <= 0)
        return host::fail(ENARF_ERR_ARG, "enarf_render_bwd: bad sizes");
    if (a.H >= (1 << 23) || a.W >= (1 << 23)) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_render_bwd: plane side >= 2^23");
    if (a.H < 2 || a.W < 2) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_render_bwd: planes of %dx%d: at least 2x2 texels", a.H, a.W);
    if (a.Nf < 2 || a.Nf > kBwdMaxSamples) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_render_bwd: Nf=%d outside [2, %d]", a.Nf, kBwdMaxSamples);
    if (!a.image_coord || !a.inv_intrinsics || !a.parts || !a.canonical_pose || !a.feat_cl || !a.mask_planes || !a.mlp_pack ||
        !a.bins || !a.grad_feat_cl || !a.grad_mask_planes || !a.rows_x || !a.rows_dz3 || !a.row_blocks || !a.workspace)
        return host::fail(ENARF_ERR_ARG, "enarf_render_bwd: null pointer");
    if (a.rows_per_image < enarf_render_bwd_rows_per_image(a.n, a.Nf))
        return host::fail(ENARF_ERR_ARG, "enarf_render_bwd: rows_per_image %lld < %lld", a.rows_per_image,
                          enarf_render_bwd_rows_per_image(a.n, a.Nf));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(a.row_blocks, 0, sizeof(unsigned int) * a.B, st);
    if (e != hipSuccess) return host::fail((int)e, "enarf_render_bwd: hipMemsetAsync failed: %s", hipGetErrorString(e));
    // per-frame tri-planes: groups of frames, as the forward (enarf_render.hip, "marched in groups") - one set-up + backward
    // launch pair per group on views of the arguments, near / far reduced once over the whole batch
    const GroupPlan pl = plan_groups(a.B, a.feat_batch_stride, a.group_frames);
    if (pl.groups == 1) return launch_bwd_group(a, st);
    const float *nf = a.near_far;
    if (!nf) {
        float *slot = ws_near_far_slot(a.workspace, a.B, a.n);
        if (int rc = enarf_near_far(a.parts, a.B, a.P, slot, stream)) return rc;
        nf = slot;
    }
    for (int g = 0; g < pl.groups; ++g) {
        const long long b0 = pl.first(g), n = a.n;
        enarf_render_bwd_args v = a;
        v.B = pl.size(g);
        v.image_coord = off(a.image_coord, b0 * 3 * n);
        v.inv_intrinsics = off(a.inv_intrinsics, b0 * 9);
        v.parts = off(a.parts, b0 * a.P * kPartStride);
        v.feat_cl = off(a.feat_cl, b0 * a.feat_batch_stride);
        v.mask_planes = off(a.mask_planes, b0 * a.mask_batch_stride);
        v.mlp_pack = reinterpret_cast<const char *>(a.mlp_pack) + (size_t)b0 * kPackBytes;
        v.bins = off(a.bins, b0 * n * a.Nf);
        v.g_color = off(a.g_color, b0 * 3 * n);
        v.g_mask = off(a.g_mask, b0 * n);
        v.g_disparity = off(a.g_disparity, b0 * n);
        v.grad_feat_cl = off(a.grad_feat_cl, b0 * a.grad_feat_batch_stride);
        v.grad_mask_planes = off(a.grad_mask_planes, b0 * a.grad_mask_batch_stride);
        v.rows_x = off(a.rows_x, b0 * a.rows_per_image * 32);
        v.rows_dz3 = off(a.rows_dz3, b0 * a.rows_per_image * 4);
        v.row_blocks = off(a.row_blocks, b0);
        v.workspace = reinterpret_cast<char *>(a.workspace) + ws_slice_off(pl, g, a.n);
        v.near_far = nf;
        if (int rc = launch_bwd_group(v, st)) return rc;
    }
    return 0;
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
    if (a.H < 2 || a.W < 2) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_query_bwd: planes of %dx%d: at least 2x2 texels", a.H, a.W);
    if (!a.points || !a.parts || !a.canonical_pose || !a.feat_cl || !a.mask_planes || !a.mlp_pack || !a.grad_feat_cl ||
        !a.grad_mask_planes || !a.rows_x || !a.rows_dz3 || !a.row_blocks)
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
    hipLaunchKernelGGL(query_bwd_kernel, dim3((unsigned)per_image, a.B), dim3(256), (size_t)bwd_lds_floats(a.P, 64) * 4, st, a, tiles);
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

// workgroups per image: one workgroup per CU over the whole batch (the kernel holds 360 registers per lane: one wave per
// SIMD), never more than the image can have 64-row shares
static int weight_grad_groups(int B, long long rows_per_image) {
    const int cus = device_cus() > 0 ? device_cus() : 256;
    long long g = (cus + B - 1) / B;
    const long long most = (rows_per_image + 63) / 64;
    if (g > most) g = most;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" size_t enarf_weight_grad_workspace_bytes(int B, long long rows_per_image) {
    if (B <= 0 || rows_per_image <= 0) return 0;
    return (size_t)B * (size_t)weight_grad_groups(B, rows_per_image) * kWgPartial * sizeof(float);
}

extern "C" int enarf_weight_grad(const enarf_weight_grad_args *args, enarf_stream_t stream) {
    if (!args) return host::fail(ENARF_ERR_ARG, "enarf_weight_grad: args is null");
    const enarf_weight_grad_args &a = *args;
    if (a.B <= 0 || a.B > 65535 || a.rows_per_image <= 0 || a.rows_per_image % 16 != 0)
        return host::fail(ENARF_ERR_ARG, "enarf_weight_grad: bad sizes (B=%d rows_per_image=%lld)", a.B, a.rows_per_image);
    if (!a.rows_x || !a.rows_dz3 || !a.mlp_pack || !a.row_blocks || !a.workspace || !a.dW1 || !a.dW2 || !a.dW3 || !a.db1 || !a.db2 || !a.db3)
        return host::fail(ENARF_ERR_ARG, "enarf_weight_grad: null pointer");
    WeightGradParams p;
    p.x = a.rows_x; p.dz3 = a.rows_dz3; p.pack = a.mlp_pack;
    p.rows_per_image = a.rows_per_image; p.row_blocks = a.row_blocks;
    p.partial = reinterpret_cast<float *>(a.workspace);
    p.groups = weight_grad_groups(a.B, a.rows_per_image);
    p.dW1 = a.dW1; p.dW2 = a.dW2; p.dW3 = a.dW3; p.db1 = a.db1; p.db2 = a.db2; p.db3 = a.db3;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;        // more than the default 64 KB of dynamic LDS
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(weight_grad_partial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return host::fail((int)e, "enarf_weight_grad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(weight_grad_partial_kernel, dim3((unsigned)p.groups, a.B), dim3(256), (size_t)kWgLdsFloats * 4, st, p);
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
