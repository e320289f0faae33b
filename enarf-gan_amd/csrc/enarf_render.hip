// enarf_render.hip - prepare, point-cloud query and the fused ray-march kernels + their C-ABI entry points.
// gfx950 only. See include/enarf_hip.h for the contract and DESIGN.md for the kernel design.
#include "enarf_tasks.h"
#include "enarf_host.h"

#ifndef ENARF_S2_PRIO
#define ENARF_S2_PRIO 3
#endif
#ifndef ENARF_RENDER_WAVES_PER_SIMD
#define ENARF_RENDER_WAVES_PER_SIMD 3
#endif

#ifndef ENARF_BATCH_CLASSES
#define ENARF_BATCH_CLASSES 4
#endif
namespace enarf {

// =================================================================================================
// enarf_prepare: part frames + per-image modulated MLP weights, one workgroup per image
// =================================================================================================
struct PrepareParams {
    enarf_prepare_args a;
};

__device__ __forceinline__ void put_split(short *h, float w) {   // h -> bf16 section; + PKH_SHORTS -> fp16 section
    const unsigned short hi = f32_to_bf16_rne(w);
    h[0] = (short)hi;
    h[512] = (short)f32_to_bf16_rne(w - bf16_to_f32(hi));
    const _Float16 fh = f32_to_f16_sat(w), fl = f32_to_f16_sat(w - (float)fh);
    h[PKH_SHORTS] = __builtin_bit_cast(short, fh);
    h[PKH_SHORTS + 512] = __builtin_bit_cast(short, fl);
}

__device__ __forceinline__ void pack_weight_row(float *__restrict__ pf, short *__restrict__ ph, int layer, int o,
                                                int c, float w) {
    // see the layout comment in enarf_device.h
    const int ob = o >> 4, i = o & 15;
    {   // transposed section (backward): element W_layer[o][c]
        float *pt = reinterpret_cast<float *>(reinterpret_cast<char *>(pf) + kPackTOff);
        const int qo = (o >> 4) * 4 + (o & 3), go = (o & 15) >> 2;          // o as a contraction index 16ob' + 4g + r'
        if (layer == 2) pt[PKT_W3T + (c >> 4) * 64 + o * 16 + (c & 15)] = w;                    // o < 4 is the k index
        else if (layer == 1) pt[PKT_W2T + ((c >> 4) * 16 + qo) * 64 + go * 16 + (c & 15)] = w;
        else pt[PKT_W1T + (((c >> 2) & 1) * 16 + qo) * 64 + go * 16 + (((c >> 3) << 2) | (c & 3))] = w;
    }
    if (layer == 0) {
        const int g = c >> 3, s = c & 7;
        pf[PK_W1 + (ob * 8 + s) * 64 + g * 16 + i] = w;
        put_split(ph + PKH_W1 + ob * 1024 + (g * 16 + i) * 8 + s, w);
    } else {
        const int obp = c >> 4, g = (c & 15) >> 2, rp = c & 3;
        const int q = obp * 4 + rp, ks = obp >> 1, jj = (obp & 1) * 4 + rp;
        if (layer == 1) {
            pf[PK_W2 + (ob * 16 + q) * 64 + g * 16 + i] = w;
            put_split(ph + PKH_W2 + (ob * 2 + ks) * 1024 + (g * 16 + i) * 8 + jj, w);
        } else {
            pf[PK_W3 + q * 64 + g * 16 + i] = w;
            put_split(ph + PKH_W3 + ks * 1024 + (g * 16 + i) * 8 + jj, w);
        }
    }
}

// part frame k of image b (pose_utils.py:129-148, rendering.py:258-260, narf.py:165): out = 16 floats
// (R row-major 9, t x coordinate_scale 3, canonical scale, pad), exact op order
__device__ __forceinline__ void compute_part_frame(const enarf_prepare_args &a, int b, int k, float out[16]) {
    const int J = a.num_joints;
    const float *pose = a.pose_to_camera + (size_t)b * J * 16;
    float bl;
    if (k < J - 1) {
        const int j = k + 1, p = a.parents[j];
        const int rs = (a.origin_location == ENARF_ORIGIN_CENTER) ? j : p;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) out[r * 3 + c] = pose[rs * 16 + r * 4 + c];
        for (int r = 0; r < 3; ++r) {
#pragma clang fp contract(off)
            out[9 + r] = ((pose[j * 16 + r * 4 + 3] + pose[p * 16 + r * 4 + 3]) / 2.0f) * a.coordinate_scale;
        }
        bl = a.bone_length[(size_t)b * (J - 1) + k];
    } else {   // center+head: the head joint's own frame, bone length 1
        const int hj = 15;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) out[r * 3 + c] = pose[hj * 16 + r * 4 + c];
        for (int r = 0; r < 3; ++r) out[9 + r] = exact_mul(pose[hj * 16 + r * 4 + 3], a.coordinate_scale);
        bl = 1.0f;
    }
    {
#pragma clang fp contract(off)
        out[12] = (a.canonical_bone_length[k] / bl) / a.coordinate_scale;
    }
    out[13] = out[14] = out[15] = 0.0f;
}
// z of part k's centre (scaled), the quantity the batch-global near / far planes reduce over
__device__ __forceinline__ float part_centre_z(const enarf_prepare_args &a, int b, int k) {
    const int J = a.num_joints;
    const float *pose = a.pose_to_camera + (size_t)b * J * 16;
    if (k < J - 1) {
#pragma clang fp contract(off)
        const int j = k + 1, p = a.parents[j];
        return ((pose[j * 16 + 2 * 4 + 3] + pose[p * 16 + 2 * 4 + 3]) / 2.0f) * a.coordinate_scale;
    }
    return exact_mul(pose[15 * 16 + 2 * 4 + 3], a.coordinate_scale);
}

constexpr int kPrepareSmemFloats = 2 * kHid + kHid * kHid;     // s_style, s_inv, s_w

// block (layer, b): one layer of image b's MLP pack; the layer-0 block also writes the part frames
__device__ __forceinline__ void prepare_block(const enarf_prepare_args &a, int layer, int b, int tid, float *smem) {
    const int J = a.num_joints;
    const int P = (a.origin_location == ENARF_ORIGIN_CENTER_HEAD) ? J : J - 1;
    float *s_style = smem, *s_inv = smem + kHid, *s_w = smem + 2 * kHid;

    if (layer == 0 && a.parts && tid < P) {
        float fr[16];
        compute_part_frame(a, b, tid, fr);
        float *out = a.parts + ((size_t)b * P + tid) * kPartStride;
        for (int i = 0; i < 16; ++i) out[i] = fr[i];
    }

    // ---- modulated, demodulated weights (custom_stylegan2/net.py:233-243) in MFMA operand order
    if (!a.mlp_pack) return;
    float *pf = reinterpret_cast<float *>(reinterpret_cast<char *>(a.mlp_pack) + (size_t)b * kPackBytes);
    short *ph = reinterpret_cast<short *>(pf + PK_F32_FLOATS);
    const int cin = (layer == 0) ? kFeat : kHid;
    const int cout = (layer == 2) ? 4 : kHid;
    if (layer == 2) {   // rows 4..15 of the last layer's 16-row operand tiles are zero; layers 0/1 write every element
        for (int i = tid; i < 16 * 64; i += 256) pf[PK_W3 + i] = 0.0f;
        for (int i = tid; i < 16; i += 256) pf[PK_B3 + i] = 0.0f;
        for (int i = tid; i < 2 * 2 * 64 * 8; i += 256) { ph[PKH_W3 + i] = 0; ph[PKH_SHORTS + PKH_W3 + i] = 0; }
    }
    const float *z = a.z_rend + (size_t)b * a.style_dim;
    const float mscale = 1.0f / sqrtf((float)a.style_dim);
    // the layer's conv weights (<= 16 per thread) are fetched before the style chain, not behind its barrier
    float cw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i;
        cw[i] = (e < cout * cin) ? a.conv_weight[layer][e] : 0.0f;
    }
    if (tid < cin) {   // EqualLinear: z @ (Wm * scale)^T + b_mod
        const float *wm = a.mod_weight[layer] + (size_t)tid * a.style_dim;
        float acc = 0.0f;
        for (int d = 0; d < a.style_dim; ++d) acc += z[d] * (wm[d] * mscale);
        s_style[tid] = acc + a.mod_bias[layer][tid];
    }
    __syncthreads();
    const float cscale = 1.0f / sqrtf((float)cin);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i;
        if (e < cout * cin) s_w[e] = (cscale * cw[i]) * s_style[e % cin];
    }
    __syncthreads();
    if (tid < cout) {   // F.normalize(dim=-1, eps=1e-12)
        float ss = 0.0f;
        for (int c = 0; c < cin; ++c) ss += s_w[tid * cin + c] * s_w[tid * cin + c];
        s_inv[tid] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        const int boff = (layer == 0) ? PK_B1 : (layer == 1) ? PK_B2 : PK_B3;
        pf[boff + tid] = a.bias[layer][tid];
    }
    __syncthreads();
    for (int e = tid; e < cout * cin; e += 256) pack_weight_row(pf, ph, layer, e / cin, e % cin, s_w[e] * s_inv[e / cin]);
}

__global__ __launch_bounds__(256) void prepare_kernel(const PrepareParams prm) {   // grid = (3, B)
    __shared__ float smem[kPrepareSmemFloats];
    prepare_block(prm.a, blockIdx.x, blockIdx.y, threadIdx.x, smem);
}

__global__ void mlp_unpack_kernel(const float *__restrict__ pf, float *__restrict__ dense) {
    // dense: W1 (64,32) | W2 (64,64) | W3 (4,64) | b1 64 | b2 64 | b3 4
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < 64 * 32) {
        const int o = tid / 32, c = tid % 32;
        dense[tid] = pf[PK_W1 + ((o >> 4) * 8 + (c & 7)) * 64 + (c >> 3) * 16 + (o & 15)];
    } else if (tid < 64 * 32 + 64 * 64) {
        const int e = tid - 64 * 32, o = e / 64, c = e % 64;
        const int q = (c >> 4) * 4 + (c & 3), g = (c & 15) >> 2;
        dense[tid] = pf[PK_W2 + ((o >> 4) * 16 + q) * 64 + g * 16 + (o & 15)];
    } else if (tid < 64 * 32 + 64 * 64 + 4 * 64) {
        const int e = tid - (64 * 32 + 64 * 64), o = e / 64, c = e % 64;
        const int q = (c >> 4) * 4 + (c & 3), g = (c & 15) >> 2;
        dense[tid] = pf[PK_W3 + q * 64 + g * 16 + o];
    } else if (tid < 64 * 32 + 64 * 64 + 4 * 64 + 132) {
        const int e = tid - (64 * 32 + 64 * 64 + 4 * 64);
        dense[tid] = (e < 64) ? pf[PK_B1 + e] : (e < 128) ? pf[PK_B2 + e - 64] : pf[PK_B3 + e - 128];
    }
}

// Squared radius (x 1.01) of the sphere around a part's origin that contains its local cube |R^T (p - t)|_inf <= 1: the
// farthest corner of the parallelepiped R^-T [-1, 1]^3 - 3 for a rotation, but computed from the matrix so that a
// non-rigid frame is never culled wrongly (a singular one gives an unbounded radius). F: R row-major (part frame record).
__device__ __forceinline__ float part_cull_radius2(const float *F) {
    // M = R^T (local = M d), rows (a b c; d e f; g h i)
    const float a = F[0], b = F[3], c = F[6], d = F[1], e = F[4], f = F[7], g = F[2], h = F[5], i = F[8];
    const float A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
    const float det = a * A + b * B + c * C;
    if (!(fabsf(det) > 1e-20f)) return 3.0e38f;
    const float r = 1.0f / det;
    const float n00 = A * r, n01 = (c * h - b * i) * r, n02 = (b * f - c * e) * r;
    const float n10 = B * r, n11 = (a * i - c * g) * r, n12 = (c * d - a * f) * r;
    const float n20 = C * r, n21 = (b * g - a * h) * r, n22 = (a * e - b * d) * r;
    float r2 = 0.0f;
#pragma unroll
    for (int sgn = 0; sgn < 4; ++sgn) {      // corners (1, +-1, +-1); the other four are their negatives
        const float sy = (sgn & 1) ? -1.0f : 1.0f, sz = (sgn & 2) ? -1.0f : 1.0f;
        const float x = n00 + n01 * sy + n02 * sz, y = n10 + n11 * sy + n12 * sz, z = n20 + n21 * sy + n22 * sz;
        r2 = fmaxf(r2, x * x + y * y + z * z);
    }
    return r2 * 1.01f;
}

// =================================================================================================
// enarf_query_fwd: a wave takes 16 points at a time (one MFMA tile), 4 adjacent lanes per point
// =================================================================================================
template <int MODE, bool DBG>
__global__ __launch_bounds__(256) void query_kernel(const enarf_query_args a, int wgs_per_image, int pts_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / wgs_per_image, chunk = bid % wgs_per_image;
    QueryCtx S;
    float *scratch;
    stage_common<MODE>(lds, S, scratch, reinterpret_cast<const char *>(a.mlp_pack) + (size_t)b * kPackBytes,
                       a.parts + (size_t)b * a.P * kPartStride, a.canonical_pose, a.P, tid, 256);
    S.feat = a.feat_cl + (size_t)b * a.feat_batch_stride;
    S.mask = a.mask_planes + (size_t)b * a.mask_batch_stride;
    S.H = a.H; S.W = a.W; S.P = a.P; S.mult_w = a.multiply_density_with_weight ? (a.uniform_part_weight ? 2 : 1) : 0;
    S.clamp_mask = a.clamp_mask; S.uniform_w = a.uniform_part_weight ? 1.0f / (float)a.P : 0.0f;
#if ENARF_DIAG_TAPCHECK
    S.diag = nullptr; S.diag_rid = 0;
#endif
    // candidates of a 16-point tile: the parts whose bounding sphere contains at least one of its points (a free point
    // cloud - or the lattice of create_mesh, mostly empty space - has no ray set-up to cull for it); debug runs keep
    // every part, they export the canonical coordinates of all of them
    int *l_cand = reinterpret_cast<int *>(scratch + 128) + wave * 32;
    float *l_rad2 = scratch + 64;
    // stage_common ends without a barrier and part k's record is written by threads 16k..16k+15 (waves 1-3 for k >= 4):
    // the radius is taken from the GLOBAL record, so no wave reads LDS another wave may not have written yet
    if (tid < a.P) l_rad2[tid] = part_cull_radius2(a.parts + ((size_t)b * a.P + tid) * kPartStride);
    __syncthreads();

    // colour of a point with no valid part: the reference still runs the MLP on a zero feature
    // (narf.py:255-268), a per-image constant; wave 0 computes it once per workgroup.
    if (wave == 0) {
        float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const f32x4 z = mlp_tile<MODE>(S, zero, lane);
        if (lane == 0) { scratch[0] = z[0]; scratch[1] = z[1]; scratch[2] = z[2]; }
    }
    __syncthreads();
    const float c0r = tanhf(scratch[0]), c0g = tanhf(scratch[1]), c0b = tanhf(scratch[2]);

    const long long N = a.N;
    const long long base = (long long)chunk * pts_per_wg;
    const float *pp = a.points + (size_t)b * 3 * N;
    for (long long off = (long long)wave * 16; off < pts_per_wg; off += 64) {
        if (base + off >= N) break;   // uniform
        const long long i = base + off + (lane >> 2);
        const bool active = i < N;
        const long long ic = active ? i : N - 1;
        float px, py, pz;
        if (a.grid_D > 0) {   // lattice mode: the point is a function of its index (N = D^3 < 2^31)
#pragma clang fp contract(off)
            const unsigned D = (unsigned)a.grid_D, idx = (unsigned)ic;
            const unsigned ix = idx / (D * D), r = idx - ix * (D * D), iy = r / D, iz = r - iy * D;
            const int c = (int)(D - 1) / 2;
            px = ((float)((int)ix - c) / (float)c + a.grid_center[0]) * a.grid_scale;
            py = ((float)((int)iy - c) / (float)c + a.grid_center[1]) * a.grid_scale;
            pz = ((float)((int)iz - c) / (float)c + a.grid_center[2]) * a.grid_scale;
        } else {
            px = pp[ic]; py = pp[N + ic]; pz = pp[2 * N + ic];
        }
        QueryDbg dbg{nullptr, nullptr, N, ic};
        if (DBG) {
            dbg.canonical = a.dbg_canonical ? a.dbg_canonical + (size_t)b * a.P * 3 * N : nullptr;
            dbg.weight = a.dbg_weight ? a.dbg_weight + (size_t)b * a.P * N : nullptr;
        }
        f32x4 o;
        bool ran;
        uint32_t bits;
        float wmax;
        unsigned np = 0, nt = 0;
        uint32_t near_parts = 0;
        if (DBG) {
            near_parts = (a.P >= 32) ? 0xFFFFFFFFu : ((1u << a.P) - 1u);
        } else {
            for (int k = lane & 3; k < a.P; k += 4) {
                const float *F = S.parts + k * kLdsPartStride;
                const float ex = px - F[9], ey = py - F[10], ez = pz - F[11];
                if (active && ex * ex + ey * ey + ez * ez <= l_rad2[k]) near_parts |= 1u << k;
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) near_parts |= (uint32_t)__shfl_xor((int)near_parts, m);
        }
        const int ncand = build_cand_list(l_cand, near_parts, lane);
        query_tile<MODE, DBG>(S, l_cand, ncand, px, py, pz, active, lane, o, ran, bits, wmax, dbg, np, nt);
        // MFMA-layout lanes 0..15 hold the head of point (lane); fetch that point's bits / wmax from its quad
        const uint32_t pbits = (uint32_t)__shfl((int)bits, (lane & 15) << 2);
        const float pwmax = __shfl(wmax, (lane & 15) << 2);
        const long long io = base + off + lane;
        if (lane < 16 && io < N) {
            a.density[(size_t)b * N + io] = density_head(o[3], pbits, pwmax, S.mult_w, S.P);
            if (a.color) {
                float *c = a.color + (size_t)b * 3 * N;
                c[io] = ran ? tanhf(o[0]) : c0r;
                c[N + io] = ran ? tanhf(o[1]) : c0g;
                c[2 * N + io] = ran ? tanhf(o[2]) : c0b;
            }
            if (a.valid_bits) a.valid_bits[(size_t)b * N + io] = pbits;
        }
    }
}

// =================================================================================================
// enarf_render_fwd: one workgroup (4 waves) per ray; each wave owns a quarter of the samples of a pass
// =================================================================================================
// bitonic sort of one value per lane, ascending across the wave
__device__ __forceinline__ float wave_sort64(float v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float o = __shfl_xor(v, j);
            const bool up = ((lane & k) == 0);
            const bool lower = ((lane & j) == 0);
            v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
        }
    }
    return v;
}

// conservative per-ray part culling: can the segment {dir * d : d in [d0, d1]} touch part k's cube?
// (slab test in the part's local frame with a relative margin; the exact per-sample test still decides)
__device__ __forceinline__ bool ray_hits_part(const float *F, float dx, float dy, float dz, float d0, float d1) {
    const float ox = -(F[0] * F[9] + F[3] * F[10] + F[6] * F[11]);
    const float oy = -(F[1] * F[9] + F[4] * F[10] + F[7] * F[11]);
    const float oz = -(F[2] * F[9] + F[5] * F[10] + F[8] * F[11]);
    const float lx = F[0] * dx + F[3] * dy + F[6] * dz;
    const float ly = F[1] * dx + F[4] * dy + F[7] * dz;
    const float lz = F[2] * dx + F[5] * dy + F[8] * dz;
    const float lim = 1.0f + 1e-3f;
    float t0 = d0, t1 = d1;
    const float o[3] = {ox, oy, oz}, l[3] = {lx, ly, lz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (fabsf(l[i]) < 1e-12f) {
            if (fabsf(o[i]) > lim) return false;
        } else {
            const float inv = 1.0f / l[i];
            float a = (-lim - o[i]) * inv, c = (lim - o[i]) * inv;
            if (a > c) { const float tmp = a; a = c; c = tmp; }
            t0 = fmaxf(t0, a - 1e-3f * fabsf(a));
            t1 = fminf(t1, c + 1e-3f * fabsf(c));
        }
    }
    return t0 <= t1;
}

// reductions over the G adjacent lanes of one ray (G = 4: the quad; G = 8: two quads, second step by row_half_mirror -
// after the quad step every lane of a quad holds its quad's value, and lane i of the half row meets lane 7 - i)
template <int G>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
    v |= (uint32_t)quad_perm_i<0xB1>((int)v);
    v |= (uint32_t)quad_perm_i<0x4E>((int)v);
    if (G == 8) v |= (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, true);     // row_half_mirror
    return v;
}
template <int G>
__device__ __forceinline__ float group_min(float v) {
    v = fminf(v, quad_perm_f<0xB1>(v)); v = fminf(v, quad_perm_f<0x4E>(v));
    if (G == 8) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, true)));
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, quad_perm_f<0xB1>(v)); v = fmaxf(v, quad_perm_f<0x4E>(v));
    if (G == 8) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, true)));
    return v;
}

// =================================================================================================
// ray set-up pre-pass: depth range, candidate parts and the compacted list of rays to march
// =================================================================================================
// workspace layout: see enarf_march.h (two queue headers, one 32-byte RayRec per ray, kQueues x kClasses ray lists).
// decide_frustrum_range (rendering.py:10-79) for every ray, 4 adjacent lanes per ray: the quad splits the parts
// for the conservative slab tests and the 32 range-test depths (8 each) for the exact cube tests. Rays the
// reference drops (batch 1, no cube hit: rendering.py:107-110, :337-350) get their zero outputs here and never
// enter the march; all others are appended to the live list in blocks that keep image order.
constexpr int kSetupSmemFloats = ENARF_MAX_PARTS * kLdsPartStride + 32 + 12 + 5 * (kClasses + 1);

// block of 256 threads: per-thread min / max of part-centre z -> l_red[8] = near, l_red[9] = far (rendering.py:15-17);
// min and max are exact and order-free, so every spelling of the reduction gives the same two floats
__device__ __forceinline__ void near_far_reduce(float mn, float mx, float *l_red, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane == 0) { l_red[wave] = mn; l_red[4 + wave] = mx; }
    __syncthreads();
    if (tid == 0) {
#pragma clang fp contract(off)
        const float zmin = fminf(fminf(l_red[0], l_red[1]), fminf(l_red[2], l_red[3]));
        const float zmax = fmaxf(fmaxf(l_red[4], l_red[5]), fmaxf(l_red[6], l_red[7]));
        l_red[8] = fmaxf(zmin - 1.7320508075688772f, 0.3f);
        l_red[9] = fmaxf(zmax + 1.7320508075688772f, 5.0f);
    }
    __syncthreads();
}
// the planes of a whole batch, for launches that render it in pieces: from the part frames, or (RAW) from the raw poses
// with the arithmetic of the part frames' z
template <bool RAW>
__global__ __launch_bounds__(256) void near_far_kernel(const float *__restrict__ parts, const enarf_prepare_args raw, int B, int P,
                                                       float *__restrict__ out) {
    __shared__ float l_red[12];
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int i = threadIdx.x; i < B * P; i += 256) {
        float z;
        if constexpr (RAW) z = part_centre_z(raw, i / P, i % P);
        else z = parts[(size_t)i * kPartStride + 11];
        mn = fminf(mn, z);
        mx = fmaxf(mx, z);
    }
    near_far_reduce(mn, mx, l_red, threadIdx.x);
    if (threadIdx.x == 0) { out[0] = l_red[8]; out[1] = l_red[9]; }
}

// RAW: the part frames are computed here from the raw joint poses (same arithmetic as prepare_block, so the values
// are bit-identical to a.parts) - lets set-up blocks run in the same launch as the prepare blocks. (`raw` stays a
// reference to the kernel argument: taking its address would move the whole argument block to scratch.)
template <bool RAW>
__device__ __forceinline__ void ray_setup_block(const enarf_render_args &a, const enarf_prepare_args &raw, int b, int blk,
                                                int tid, float *smem) {
    float *l_parts = smem;
    float *l_dtab = smem + ENARF_MAX_PARTS * kLdsPartStride;
    float *l_red = l_dtab + 32;
    int *l_cnt = reinterpret_cast<int *>(l_red + 12);
    constexpr int G = kSetupLanes;              // lanes per ray: they split the parts and the 32 range-test depths
    const int lane = tid & 63, wave = tid >> 6, g = tid & (G - 1);
    const int P = a.P, n = a.n, Nf = a.Nf;
    // the ray's own inputs first: their round trip overlaps the part-frame and near / far chains below
    const int ray = blk * kSetupRays + tid / G;
    const bool in_range = ray < n;
    const int rc = in_range ? ray : n - 1;
    const float *coord = a.image_coord + (size_t)b * 3 * n;
    const float *Ki = a.inv_intrinsics + (size_t)b * 9;
    const float u = coord[rc], v = coord[n + rc], w = coord[2 * n + rc];
    const float k0 = Ki[0], k1 = Ki[1], k2 = Ki[2], k3 = Ki[3], k4 = Ki[4], k5 = Ki[5], k6 = Ki[6], k7 = Ki[7], k8 = Ki[8];
    if constexpr (RAW) {
        if (tid < P) {
            float fr[16];
            compute_part_frame(raw, b, tid, fr);
            for (int i = 0; i < 16; ++i) l_parts[tid * kLdsPartStride + i] = fr[i];
        }
    } else {
        for (int i = tid; i < P * kPartStride; i += 256)
            l_parts[(i / kPartStride) * kLdsPartStride + (i % kPartStride)] = a.parts[(size_t)b * P * kPartStride + i];
    }
    if (a.near_far) {   // given by the caller: this launch renders a piece of a larger batch (block-uniform branch)
        if (tid == 0) { l_red[8] = a.near_far[0]; l_red[9] = a.near_far[1]; }
        __syncthreads();
    } else {   // batch-global near / far planes (rendering.py:15-17): min / max of every part centre's z
        float mn = 3.0e38f, mx = -3.0e38f;
        for (int i = tid; i < a.B * P; i += 256) {
            float z;
            if constexpr (RAW) z = part_centre_z(raw, i / P, i % P);
            else z = a.parts[(size_t)i * kPartStride + 11];
            mn = fminf(mn, z);
            mx = fmaxf(mx, z);
        }
        near_far_reduce(mn, mx, l_red, tid);
    }
    const float near_p = l_red[8], far_p = l_red[9];
    if (tid < 32) l_dtab[tid] = linspace_sym(near_p, far_p, 32, tid);
    __syncthreads();

    const float dx = exact_dot3(k0, u, k1, v, k2, w);
    const float dy = exact_dot3(k3, u, k4, v, k5, w);
    const float dz = exact_dot3(k6, u, k7, v, k8, w);

    // parts the ray can touch between near and far (conservative), split over the ray's lanes
    uint32_t mine = 0;
    for (int k = g; k < P; k += G)
        if (ray_hits_part(l_parts + k * kLdsPartStride, dx, dy, dz, near_p, far_p)) mine |= 1u << k;
    const uint32_t cand_all = group_or<G>(mine);

    // exact range test: this lane's D of the 32 depths x candidate parts; part outer (its frame is read from LDS
    // once), depths inner; bit d of `in8` = some part contains depth g*D + d
    constexpr int D = 32 / G;
    float qx[D], qy[D], qz[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float ds = l_dtab[g * D + d];
        qx[d] = exact_mul(dx, ds); qy[d] = exact_mul(dy, ds); qz[d] = exact_mul(dz, ds);
    }
    uint32_t in8 = 0;
    {
        uint32_t m = cand_all;
        while (m) {
            const int k = __builtin_ctz(m);
            m &= m - 1;
            float F[12];
            const f32x4 *pf = reinterpret_cast<const f32x4 *>(l_parts + k * kLdsPartStride);
            const f32x4 f0 = pf[0], f1 = pf[1], f2 = pf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { F[i] = f0[i]; F[4 + i] = f1[i]; F[8 + i] = f2[i]; }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                float lx, ly, lz;
                exact_local(F, qx[d], qy[d], qz[d], lx, ly, lz);
                if (in_unit_cube_incl(lx, ly, lz)) in8 |= 1u << d;
            }
        }
    }
    float mn = 1.0e3f, mx = -1.0e3f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if ((in8 >> d) & 1u) {
            const float ds = l_dtab[g * D + d];
            mn = fminf(mn, ds);
            mx = fmaxf(mx, ds);
        }
    }
    mn = group_min<G>(mn);
    mx = group_max<G>(mx);
    const bool ray_valid = (mn != 1.0e3f);
    float dmin = ray_valid ? mn : near_p;
    const float dmax = (mx != -1.0e3f) ? mx : far_p;
    dmin = fmaxf(dmin, near_p);
    const bool live = in_range && (ray_valid || !a.drop_invalid_rays);

    // parts over the marched segment only
    uint32_t mine2 = 0;
    for (int k = g; k < P; k += G)
        if (((cand_all >> k) & 1u) && ray_hits_part(l_parts + k * kLdsPartStride, dx, dy, dz, dmin, dmax)) mine2 |= 1u << k;
    const uint32_t cand = group_or<G>(mine2);

    const size_t rid = (size_t)b * n + rc;
    if (in_range && g == 0) {
        RayRec *recs = reinterpret_cast<RayRec *>(reinterpret_cast<char *>(a.workspace) + ws_records_off());
        recs[rid] = RayRec{dmin, dmax, cand, ray_valid ? 1u : 0u, dx, dy, dz, 0.0f};
        if (a.dbg_depth_min) {
            a.dbg_depth_min[rid] = dmin;
            a.dbg_depth_max[rid] = dmax;
            a.dbg_ray_valid[rid] = ray_valid ? 1 : 0;
        }
    }
    if (in_range && !live && a.color) {   // dropped ray: zeros (rendering.py:337-350); the backward passes no outputs
        if (g < 3) a.color[((size_t)b * 3 + g) * n + ray] = 0.0f;
        if (g == 3) { a.mask[rid] = 0.0f; a.disparity[rid] = 0.0f; }
        if (a.fine_weights) for (int i = g; i < Nf - 1; i += G) a.fine_weights[rid * (Nf - 1) + i] = 0.0f;
        if (a.fine_depth) for (int i = g; i < Nf; i += G) a.fine_depth[rid * Nf + i] = 0.0f;
    }
    // file the block's live rays, in ray order, under (band, cost class) - see RayQueue; live rays without a candidate
    // part (batches only: a single image drops them) go to the band's list of missed rays
    constexpr int kFile = kClasses + 1;
    // more images than bands: a band holds several whole frames, and class-major order would walk each of them once per
    // class (ENARF_BATCH_CLASSES: how many of the cost classes such batches use; measured: no difference)
    const int cls = (cand == 0u) ? kClasses
                    : (a.B > kQueues) ? min(ray_cost_class(cand), ENARF_BATCH_CLASSES - 1) : ray_cost_class(cand);
    const bool file_it = live && g == 0;
    uint64_t bal[kFile];
#pragma unroll
    for (int c = 0; c < kFile; ++c) {
        bal[c] = __ballot(file_it && cls == c);
        if (lane == 0) l_cnt[wave * kFile + c] = __popcll(bal[c]);
    }
    const long long band = ws_band_size(a.B, n);
    const int q = ws_band_of(a.B, n, b, blk);
    unsigned int *wsh = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(a.workspace) + ws_header_off(a.ws_epoch));
    __syncthreads();
    if (tid < kFile) {
        const int tot = l_cnt[tid] + l_cnt[kFile + tid] + l_cnt[2 * kFile + tid] + l_cnt[3 * kFile + tid];
        const int lid = (tid < kClasses) ? q * kClasses + tid : ws_missed_list(q);
        l_cnt[4 * kFile + tid] = tot ? (int)atomicAdd(wsh + kWsCountsOff + lid, (unsigned int)tot) : 0;
        if (tot) atomicAdd(wsh + ((tid < kClasses) ? 1 : 2), (unsigned int)tot);     // [1] rays to march, [2] rays that miss every cube
    }
    __syncthreads();
    if (file_it) {
        int pos = l_cnt[4 * kFile + cls];
        for (int wv = 0; wv < wave; ++wv) pos += l_cnt[wv * kFile + cls];
#pragma unroll
        for (int c = 0; c < kFile; ++c)
            if (cls == c) pos += __popcll(bal[c] & ((1ull << lane) - 1ull));
        uint32_t *lists = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(a.workspace) + ws_list_off((long long)a.B * n));
        const int lid = (cls < kClasses) ? q * kClasses + cls : ws_missed_list(q);
        lists[(size_t)lid * (size_t)band + pos] = (uint32_t)rid;
    }
}

__global__ __launch_bounds__(256) void ray_setup_kernel(const enarf_render_args a, int blocks_per_image) {
    __shared__ __attribute__((aligned(16))) float smem[kSetupSmemFloats];
    enarf_prepare_args unused;   // never read with RAW = false
    if (blockIdx.x == 0) ws_clear_other_header(a.workspace, a.ws_epoch, threadIdx.x);
    ray_setup_block<false>(a, unused, blockIdx.x / blocks_per_image, blockIdx.x % blocks_per_image, threadIdx.x, smem);
}

// ---- one launch for everything that precedes the march: tri-plane re-layout, MLP packs + part frames, ray set-up --------
// The three jobs are independent (set-up blocks derive their part frames themselves), so their blocks run side by side
// instead of three dependent launches back to back.
struct PreParams {
    enarf_prepare_args prep;
    enarf_render_args rend;
    const float *tri;          // NCHW tri-plane(s), or null: feat_cl is already up to date
    float *feat_cl;
    int tri_B, ch_total, H, W;
    int n_pack, n_prep, bpi;   // block counts of the pack and prepare roles; set-up blocks per image
};
constexpr int kPreSmemFloats = (kPrepareSmemFloats > kFeat * 65) ? kPrepareSmemFloats : kFeat * 65;
static_assert(kPreSmemFloats >= kSetupSmemFloats, "shared memory of the fused pre-march kernel");

__global__ __launch_bounds__(256) void pre_march_kernel(const PreParams q) {
    __shared__ __attribute__((aligned(16))) float smem[kPreSmemFloats];
    int id = blockIdx.x;
    const int tid = threadIdx.x;
    // long-latency roles first (few blocks, serial chains), the bandwidth-bound re-layout blocks fill in behind them
    if (id < q.n_prep) {
        if (id == 0) ws_clear_other_header(q.rend.workspace, q.rend.ws_epoch, tid);
        prepare_block(q.prep, id % 3, id / 3, tid, smem);
        return;
    }
    id -= q.n_prep;
    const int n_setup = q.bpi * q.rend.B;
    if (id < n_setup) {
        ray_setup_block<true>(q.rend, q.prep, id / q.bpi, id % q.bpi, tid, smem);
        return;
    }
    id -= n_setup;
    const int xblocks = (q.W + 63) / 64;
    pack_block<kFeat>(q.tri, q.feat_cl, q.ch_total, q.H, q.W, id % xblocks, (id / xblocks) % q.H, id / (xblocks * q.H), tid, smem);
}

// scratch layout (floats) of the render kernel; up to kMaxSamples samples per pass
constexpr int SC_BTAB = 0;        // Nc + 1 bin edges (<= 129)
constexpr int SC_CAND = 136;      // 4 waves x 32 ints
constexpr int SC_CH = 264;        // coarse: sigma head [128]
constexpr int SC_CBITS = 392;     // coarse: bits [128]
constexpr int SC_CWMAX = 520;     // coarse: wmax [128]
constexpr int SC_FH = 648;        // fine: head [4][128]
constexpr int SC_FBITS = 1160;    // fine: bits [128]
constexpr int SC_FWMAX = 1288;    // fine: wmax [128]
constexpr int SC_QUEUE = 1416;    // 2 ray ids (current / prefetched)
constexpr int SC_BINS = 1480;     // importance samples of the ray [128] + 8 skip flags (written by the one wave that runs S2)
static_assert(SC_QUEUE + kQueueLdsInts <= SC_BINS && SC_BINS + kMaxSamples + 8 <= kScratchFloats, "scratch overflow");

// SPL = samples per lane in the lane = sample stages: 1 for Nc, Nf <= 64, 2 up to 128 (each wave then loops over two
// 16-sample tiles per pass)
template <int MODE, int SPL>
__global__ __launch_bounds__(256, ENARF_RENDER_WAVES_PER_SIMD) void render_kernel(const enarf_render_args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Persistent workgroups: each marches one ray at a time, taken off the (band, cost class) lists the set-up pass
    // filled - heaviest class first chip-wide, own XCD's band first within a class (RayQueue, enarf_march.h).
    const int P = a.P, Nc = a.Nc, Nf = a.Nf, n = a.n;
    RayQueue rq;

#if ENARF_TIMERS == 3   // workgroup start / end on the 100 MHz wall clock: how much of the launch is tail
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    QueryCtx S;
    float *scratch = lds + (lds_total_floats<MODE>(P) - kScratchFloats);
    int *l_q = reinterpret_cast<int *>(scratch + SC_QUEUE);
    float *l_btab = scratch + SC_BTAB;
    if (tid <= Nc) l_btab[tid] = linspace_sym(0.0f, 1.0f, Nc + 1, tid);
    __syncthreads();
    // rays without a candidate part first (batches only, enarf_tasks.h): every wave on its own, with a private scratch
    // slot in the not yet staged MLP section of the LDS (4 x 5.4 KB of 28 KB)
    static_assert(4 * kSlotWords <= lds_mlp_floats<MODE>(), "scratch slots of the missed-ray pass");
    const unsigned n_missed = march_missed_rays<SPL>(kernel_render_args(), l_btab, reinterpret_cast<unsigned *>(lds) + wave * kSlotWords,
                                                     a.multiply_density_with_weight ? (a.uniform_part_weight ? 2 : 1) : 0, lane);
    if (a.counters && lane == 0 && n_missed) atomicAdd(&a.counters[2], (unsigned long long)n_missed);
    __syncthreads();
    rq.init(a.workspace, a.ws_epoch, a.B, n, l_q, tid);
    if (tid == 0) rq.pop(0);
    __syncthreads();
    int cur = rq.get(0);
    if (cur < 0) return;                      // uniform: every queue is already drained
    int b = (int)((uint32_t)cur / (uint32_t)n);
    stage_common<MODE>(lds, S, scratch, reinterpret_cast<const char *>(a.mlp_pack) + (size_t)b * kPackBytes,
                       a.parts + (size_t)b * P * kPartStride, a.canonical_pose, P, tid, 256);
    S.feat = a.feat_cl + (size_t)b * a.feat_batch_stride;
    S.mask = a.mask_planes + (size_t)b * a.mask_batch_stride;
    S.H = a.H; S.W = a.W; S.P = P; S.mult_w = a.multiply_density_with_weight ? (a.uniform_part_weight ? 2 : 1) : 0;
    S.clamp_mask = a.clamp_mask; S.uniform_w = a.uniform_part_weight ? 1.0f / (float)P : 0.0f;
    int *l_cand = reinterpret_cast<int *>(scratch + SC_CAND) + wave * 32;
    float *l_ch = scratch + SC_CH, *l_cwmax = scratch + SC_CWMAX, *l_fh = scratch + SC_FH, *l_fwmax = scratch + SC_FWMAX;
    uint32_t *l_cbits = reinterpret_cast<uint32_t *>(scratch + SC_CBITS);
    uint32_t *l_fbits = reinterpret_cast<uint32_t *>(scratch + SC_FBITS);
    __syncthreads();                          // the staged image context is complete before any wave starts a tile

    unsigned n_pairs = 0, n_tiles = 0, n_rays = 0, n_rounds = 0, n_skipped = 0;
    int qslot = 0;
    const QueryDbg nodbg{nullptr, nullptr, 0, 0};
    const int j4 = lane >> 2;                              // this lane's sample within the wave's tile
    const bool dbgq = (a.dbg_fine_density != nullptr);

#if ENARF_TIMERS
    for (int k = 0; k < 8; ++k) S.tmr[k] = 0;
    S.tmr_t = __builtin_amdgcn_s_memtime();
#endif
#if ENARF_TIMERS == 3
    unsigned long long ray_t0 = wg_t0, ray_max = 0;
#endif
    while (cur >= 0) {
#if ENARF_TIMERS == 3
        { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (now - ray_t0 > ray_max && n_rays) ray_max = now - ray_t0; ray_t0 = now; }
#endif
        const uint32_t rid = (uint32_t)cur;
#if ENARF_DIAG_TAPCHECK
        S.diag = a.counters; S.diag_rid = rid;
#endif
        // next entry (read after the S1 barrier): popped by the wave that has no coarse tile in this ray, when there is
        // one - the atomic's round trip then costs nothing
        const int spare_wave = (3 * SPL * 16 >= Nc) ? ((3 - (int)(rid & 3u)) & 3) : 0;
        if (wave == spare_wave && lane == 0) rq.pop(qslot ^ 1);
        const int nb = (int)(rid / (uint32_t)n), ray = (int)(rid - (uint32_t)nb * (uint32_t)n);
        if (nb != b) {   // next image: restage its MLP pack and part frames (the lists are in image order, so this is rare)
            b = nb;
            __syncthreads();
            stage_common<MODE>(lds, S, scratch, reinterpret_cast<const char *>(a.mlp_pack) + (size_t)b * kPackBytes,
                               a.parts + (size_t)b * P * kPartStride, a.canonical_pose, P, tid, 256);
            S.feat = a.feat_cl + (size_t)b * a.feat_batch_stride;
            S.mask = a.mask_planes + (size_t)b * a.mask_batch_stride;
            __syncthreads();
        }
        // depth range, candidate parts and ray direction K^-1 [u v w] (rendering.py:26-38): from the set-up pre-pass,
        // left in LDS by the pop
        const RayRec rec = rq.rec(qslot);
        const float dx = rec.dx, dy = rec.dy, dz = rec.dz;
        const float dmin = rec.dmin, dmax = rec.dmax;
        if (wave == 0) n_rays += 1;
        const int ncand = build_cand_list(l_cand, rec.cand, lane);
        const float sx = exact_mul(dmin, dx), sy = exact_mul(dmin, dy), sz = exact_mul(dmin, dz);
        const float ex = exact_mul(dmax, dx), ey = exact_mul(dmax, dy), ez = exact_mul(dmax, dz);
        TMR(S, 0);
        TMR4(S, 0);

        // ---- S1: coarse pass (rendering.py:119-131, :172) in FULL tiles of 16 bins: tile t of the ray goes to wave slot
        // t / SPL. With Nc = 48 that is three full tiles instead of four tiles of 12 - a quarter fewer gather rounds
        // and MLP tiles for the same critical path - and the spared wave rotates ray by ray so that no SIMD idles.
        {
            const int slot = (wave + (int)(rid & 3u)) & 3;
#pragma unroll
            for (int u = 0; u < SPL; ++u) {
                const int base = (slot * SPL + u) * 16;
                if (base >= Nc) break;
                const int i = base + j4;
                const bool active = i < Nc;
                const int ci = min(i, Nc - 1);
                const float b0 = l_btab[ci], b1 = l_btab[ci + 1];
                const float px = exact_mid(exact_lerp(sx, ex, b1), exact_lerp(sx, ex, b0));
                const float py = exact_mid(exact_lerp(sy, ey, b1), exact_lerp(sy, ey, b0));
                const float pz = exact_mid(exact_lerp(sz, ez, b1), exact_lerp(sz, ez, b0));
                f32x4 o;
                bool ran;
                uint32_t bits;
                float wmax;
                query_tile<MODE, false>(S, l_cand, ncand, px, py, pz, active, lane, o, ran, bits, wmax, nodbg, n_pairs, n_tiles, &n_rounds);
                if (lane < 16 && base + lane < Nc) l_ch[base + lane] = o[3];     // MFMA layout: lanes < 16 hold the sample heads
                if (active && (lane & 3) == 0) { l_cbits[i] = bits; l_cwmax[i] = wmax; }
            }
        }
        // the sorted uniforms of the importance draw do not depend on the coarse pass: the wave that will run S2 draws
        // them now, while the others are still in their coarse tiles (u_(i) = E_1+..+E_i / E_1+..+E_{Nf+1})
        float usort[SPL];
#pragma unroll
        for (int s = 0; s < SPL; ++s) usort[s] = 0.0f;
        if (wave == spare_wave && !a.bins && !(ENARF_DIAG_ABLATE & 8)) {
            float esum[SPL];
            uint32_t r1_first = 0;
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                uint32_t rnd[4];
                philox4x32((uint32_t)(a.ray_id_base + rid), (uint32_t)((a.ray_id_base + rid) >> 32), (uint32_t)(64 * s + lane), 0x454E4152u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), rnd);
                esum[s] = (64 * s + lane < Nf) ? -__logf(1.0f - u32_to_unit(rnd[0])) : 0.0f;
                if (s == 0) r1_first = rnd[1];
            }
            wv_scan_incl<SPL>(esum, lane);
            const float etot = __shfl(esum[SPL - 1], 63) - __logf(1.0f - u32_to_unit((uint32_t)__shfl((int)r1_first, 0)));
#pragma unroll
            for (int s = 0; s < SPL; ++s) usort[s] = fminf(esum[s] / etot, 0.99999994f);
        }
        TMR(S, 4);
        TMR4(S, 7);
        __syncthreads();
        TMR(S, 5);
        TMR4(S, 6);
        const int next_ray = rq.get(qslot ^ 1);
        qslot ^= 1;

        // ---- S2 (ONE wave, element e = 64 s + lane): weights (rendering.py:180-184), smoothing (:187-190), bins (:192-197).
        // The result goes through LDS; the other waves wait at the barrier instead of spending the same ~350 VALU
        // instructions each (the SIMDs are shared with two other workgroups that can use the slots).
        float *l_bins = scratch + SC_BINS;
        int *l_skip = reinterpret_cast<int *>(scratch + SC_BINS + kMaxSamples);
        if (wave == spare_wave) {
#if ENARF_S2_PRIO
            __builtin_amdgcn_s_setprio(ENARF_S2_PRIO);     // three waves wait at the next barrier for this one
#endif
            float bin[SPL];
            bool skip_tile[4 * SPL];

            float dd[SPL], cs[SPL], T[SPL], wgt[SPL], ws[SPL], wl[SPL], wr[SPL];
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int e = 64 * s + lane;
                const bool active = e < Nc;
                const int ci = min(e, Nc - 1);
                const float den = active ? density_head(l_ch[ci], l_cbits[ci], l_cwmax[ci], S.mult_w, P) : 0.0f;
                if (a.dbg_coarse_density && active) a.dbg_coarse_density[((size_t)b * n + ray) * Nc + e] = den;
                const float b0 = l_btab[ci], b1 = l_btab[ci + 1];
                const float delta = exact_lerp(dmin, dmax, b1) - exact_lerp(dmin, dmax, b0);
                dd[s] = active ? den * delta * a.render_scale : 0.0f;
                cs[s] = dd[s];
            }
            wv_scan_incl<SPL>(cs, lane);
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                T[s] = expf(-(cs[s] - dd[s]));
                wgt[s] = (64 * s + lane < Nc) ? T[s] * (1.0f - expf(-dd[s])) : 0.0f;
            }
            wv_prev<SPL>(wgt, wl, lane);
            wv_next<SPL>(wgt, wr, lane);
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int e = 64 * s + lane;
                if (e >= Nc - 1) wr[s] = 0.0f;
                ws[s] = (e < Nc) ? (fmaxf(wl[s], wgt[s]) + fmaxf(wgt[s], wr[s])) / 2.0f + 0.01f : 0.0f;
            }
            TMR4(S, 1);
            if (ENARF_DIAG_ABLATE & 8) {
#pragma unroll
                for (int s = 0; s < SPL; ++s) bin[s] = (float)(64 * s + lane) / (float)Nf;
            } else if (a.bins) {
#pragma unroll
                for (int s = 0; s < SPL; ++s) bin[s] = a.bins[((size_t)b * n + ray) * Nf + min(64 * s + lane, Nf - 1)];
            } else {
                // Importance samples = Nf iid draws from the piecewise-constant pdf, sorted (rendering.py:192-197).
                // Sorted uniforms come directly from exponential spacings (u_(i) = E_1+..+E_i / E_1+..+E_{Nf+1}),
                // each is pushed through the inverse CDF (monotone, so the bins come out sorted): bin index by
                // binary search, position inside the bin by the leftover - the same law as multinomial + U/Nc.
                // (the sorted uniforms were drawn before the barrier)
                float cdf[SPL];
#pragma unroll
                for (int s = 0; s < SPL; ++s) cdf[s] = ws[s];
                wv_scan_incl<SPL>(cdf, lane);
                const float total = wv_get<SPL>(cdf, Nc - 1);
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    const float target = usort[s] * total;
                    int lo = 0, hi = Nc - 1;   // smallest i with cdf[i] > target
#pragma unroll
                    for (int it = 0; it < 5 + SPL; ++it) {
                        const int mid = (lo + hi) >> 1;
                        const float c = wv_get<SPL>(cdf, mid);
                        if (lo < hi) { if (c > target) hi = mid; else lo = mid + 1; }
                    }
                    const float c_hi = wv_get<SPL>(cdf, lo), c_lo = wv_get<SPL>(cdf, max(lo - 1, 0));
                    const float below = (lo > 0) ? c_lo : 0.0f;
                    const float frac = fminf(fmaxf((target - below) / (c_hi - below), 0.0f), 0.99999994f);
                    bin[s] = (64 * s + lane < Nf) ? ((float)lo + frac) / (float)Nc : 3.0e38f;
                }
            }
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                l_bins[64 * s + lane] = bin[s];
                if (a.dbg_bins && 64 * s + lane < Nf) a.dbg_bins[((size_t)b * n + ray) * Nf + 64 * s + lane] = bin[s];
            }
            // early ray termination (opt-in, early_stop_eps > 0): transmittance in front of the first fine sample of
            // each tile, read off the coarse pass (T before coarse bin j). Once it is below eps every sample of the tile
            // weighs < eps: the tile's gathers and MLP are skipped (its densities count as 0).
#pragma unroll
            for (int t = 0; t < 4 * SPL; ++t) {
                skip_tile[t] = false;
                if (a.early_stop_eps > 0.0f) {
                    const float b_first = wv_get<SPL>(bin, min(16 * t, Nf - 1));
                    const int jbin = min(max((int)(b_first * (float)Nc), 0), Nc - 1);
                    skip_tile[t] = wv_get<SPL>(T, jbin) < a.early_stop_eps;
                }
                if (lane == 0) l_skip[t] = skip_tile[t] ? 1 : 0;
            }
#if ENARF_S2_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        TMR(S, 6);
        TMR4(S, 2);
        __syncthreads();
        TMR(S, 5);
        TMR4(S, 6);

        // ---- S3: fine pass, in full tiles of 16 samples like the coarse pass (tile T on wave slot T / SPL; Nf = 48 or 32
        // leave one or two waves without a tile); the last sample only closes the last interval and is never queried
        {
            const int slotf = (wave + (int)((rid >> 2) & 3u)) & 3;
#pragma unroll
            for (int u = 0; u < SPL; ++u) {
                const int T = slotf * SPL + u, base = T * 16;
                if (base >= Nf) break;
                const int i = base + j4;
                const bool skip = l_skip[T] != 0;
                const bool active = (i < (dbgq ? Nf : Nf - 1)) && !skip;
                const float bi = l_bins[min(i, Nf - 1)];
                const float px = exact_lerp(sx, ex, bi), py = exact_lerp(sy, ey, bi), pz = exact_lerp(sz, ez, bi);
                f32x4 o;
                bool ran;
                uint32_t bits;
                float wmax;
                query_tile<MODE, false>(S, l_cand, ncand, px, py, pz, active, lane, o, ran, bits, wmax, nodbg, n_pairs, n_tiles, &n_rounds);
                if (lane < 16 && base + lane < Nf) {
                    const int io = base + lane;
                    l_fh[io] = o[0]; l_fh[kMaxSamples + io] = o[1]; l_fh[2 * kMaxSamples + io] = o[2]; l_fh[3 * kMaxSamples + io] = o[3];
                }
                if (i < Nf && (lane & 3) == 0) { l_fbits[i] = active ? bits : 0u; l_fwmax[i] = wmax; }
                if (skip && lane == 0) n_skipped += 1;
            }
        }
        TMR(S, 4);
        TMR4(S, 7);
        __syncthreads();
        TMR(S, 5);
        TMR4(S, 6);

        // ---- S4 (one wave, element e = 64 s + lane): compositing (rendering.py:307-335). It runs while the other waves are
        // already in the next ray's coarse pass - on the wave that has no coarse tile there, when there is one.
        const int s4_wave = (next_ray >= 0 && 3 * SPL * 16 >= Nc) ? ((3 - (next_ray & 3)) & 3) : 0;
        if (wave == s4_wave) {
            float fdepth[SPL], dnext[SPL], den[SPL], cr[SPL], cg[SPL], cb[SPL], dd[SPL], cs[SPL];
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int e = 64 * s + lane;
                const int ci = min(e, Nf - 1);
                const bool have = e < (dbgq ? Nf : Nf - 1);
                const uint32_t bits = l_fbits[ci];
                den[s] = have ? density_head(l_fh[3 * kMaxSamples + ci], bits, l_fwmax[ci], S.mult_w, P) : 0.0f;
                cr[s] = tanhf(l_fh[ci]); cg[s] = tanhf(l_fh[kMaxSamples + ci]); cb[s] = tanhf(l_fh[2 * kMaxSamples + ci]);
                fdepth[s] = exact_lerp(dmin, dmax, l_bins[64 * s + lane]);
                if (dbgq && e < Nf) {
                    const size_t o = ((size_t)b * n + ray) * Nf + e;
                    a.dbg_fine_density[o] = den[s];
                    if (a.dbg_fine_valid) a.dbg_fine_valid[o] = bits;
                    if (a.dbg_fine_color) {
                        a.dbg_fine_color[(((size_t)b * 3 + 0) * n + ray) * Nf + e] = cr[s];
                        a.dbg_fine_color[(((size_t)b * 3 + 1) * n + ray) * Nf + e] = cg[s];
                        a.dbg_fine_color[(((size_t)b * 3 + 2) * n + ray) * Nf + e] = cb[s];
                    }
                }
            }
            TMR4(S, 3);
            wv_next<SPL>(fdepth, dnext, lane);
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                dd[s] = (64 * s + lane < Nf - 1) ? den[s] * (dnext[s] - fdepth[s]) * a.render_scale : 0.0f;
                cs[s] = dd[s];
            }
            wv_scan_incl<SPL>(cs, lane);
            float wgt[SPL], vr[SPL], vg[SPL], vb[SPL], vd[SPL];
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const bool seg = 64 * s + lane < Nf - 1;
                const float T = expf(-(cs[s] - dd[s]));
                wgt[s] = seg ? T * (1.0f - expf(-dd[s])) : 0.0f;
                vr[s] = wgt[s] * cr[s]; vg[s] = wgt[s] * cg[s]; vb[s] = wgt[s] * cb[s];
                vd[s] = seg ? (wgt[s] * 1.0f) / fdepth[s] : 0.0f;
            }
            TMR4(S, 4);
            const float o_r = wv_sum<SPL>(vr), o_g = wv_sum<SPL>(vg), o_b = wv_sum<SPL>(vb);
            const float o_m = wv_sum<SPL>(wgt), o_d = wv_sum<SPL>(vd);
            if (lane == 0) {
                a.color[((size_t)b * 3 + 0) * n + ray] = o_r;
                a.color[((size_t)b * 3 + 1) * n + ray] = o_g;
                a.color[((size_t)b * 3 + 2) * n + ray] = o_b;
                a.mask[(size_t)b * n + ray] = o_m;
                a.disparity[(size_t)b * n + ray] = o_d;
            }
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int e = 64 * s + lane;
                if (a.fine_weights && e < Nf - 1) a.fine_weights[((size_t)b * n + ray) * (Nf - 1) + e] = wgt[s];
                if (a.fine_depth && e < Nf) a.fine_depth[((size_t)b * n + ray) * Nf + e] = fdepth[s];
            }
        }
        TMR(S, 7);
        TMR2(S, 7);
        TMR4(S, 5);
        // no barrier needed here: coarse arrays are rewritten in S1' (after this ray's S3 barrier, which follows every
        // wave's S2 reads), fine arrays in S3' (after the S1' barrier, which wave 0 reaches only after this S4).
        cur = next_ray;
    }
#if ENARF_TIMERS == 3
    if (a.counters && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        atomicMin(&a.counters[0], wg_t0);
        atomicMax(&a.counters[1], t1);
        atomicAdd(&a.counters[2], t1 - wg_t0);
        atomicAdd(&a.counters[3], 1ull);
        atomicMax(&a.counters[4], wg_t0);
        atomicMin(&a.counters[5], t1);
        atomicAdd(&a.counters[6], t1 - ray_t0);      // duration of this workgroup's last ray
        atomicMax(&a.counters[7], ray_max);          // longest ray of the launch
    }
    return;
#elif ENARF_TIMERS
    if (a.counters && lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&a.counters[k], S.tmr[k]);
    return;
#endif
    if (a.counters && lane == 0) {
        atomicAdd(&a.counters[0], (unsigned long long)n_pairs);
        atomicAdd(&a.counters[1], (unsigned long long)n_tiles);
        atomicAdd(&a.counters[2], (unsigned long long)n_rays);
        atomicAdd(&a.counters[3], (unsigned long long)n_rounds);
        atomicAdd(&a.counters[4], (unsigned long long)n_skipped);
    }
}

}  // namespace enarf

// =================================================================================================
// C ABI
// =================================================================================================
using namespace enarf;

extern "C" int enarf_abi_version(void) { return ENARF_ABI_VERSION; }
extern "C" int enarf_version(void) { return ENARF_ABI_VERSION; }

extern "C" int enarf_device_status(unsigned int *flags, int clear) {
    if (!flags) return host::fail(ENARF_ERR_ARG, "enarf_device_status: flags is null");
    volatile unsigned int *w = host::status_word(false);
    if (!w) return host::fail((int)hipErrorOutOfMemory, "enarf_device_status: no status word for this device (pinned host allocation failed)");
    *flags = w[0];
    if (clear && *flags) __atomic_fetch_and(const_cast<unsigned int *>(w), ~*flags, __ATOMIC_RELAXED);
    return 0;
}
extern "C" const char *enarf_last_error(void) { return host::last_error(); }
extern "C" size_t enarf_mlp_pack_bytes(void) { return kPackBytes; }

static int check_prepare(const enarf_prepare_args &a) {
    if (a.B > 65535) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_prepare: B > 65535");
    if (a.B <= 0 || a.num_joints < 2 || a.num_joints > ENARF_MAX_JOINTS || a.style_dim <= 0)
        return host::fail(ENARF_ERR_ARG, "enarf_prepare: bad sizes (B=%d joints=%d style_dim=%d)", a.B, a.num_joints, a.style_dim);
    if (a.origin_location < 0 || a.origin_location > 2) return host::fail(ENARF_ERR_ARG, "enarf_prepare: bad origin_location %d", a.origin_location);
    if (a.origin_location == ENARF_ORIGIN_CENTER_HEAD && a.num_joints <= 15)
        return host::fail(ENARF_ERR_ARG, "enarf_prepare: center+head needs joint 15 (head)");
    if (a.parts && (!a.pose_to_camera || !a.bone_length || !a.canonical_bone_length))
        return host::fail(ENARF_ERR_ARG, "enarf_prepare: pose / bone_length / canonical_bone_length pointer is null");
    if (a.mlp_pack) {
        if (!a.z_rend) return host::fail(ENARF_ERR_ARG, "enarf_prepare: z_rend is null");
        for (int i = 0; i < 3; ++i)
            if (!a.conv_weight[i] || !a.mod_weight[i] || !a.mod_bias[i] || !a.bias[i])
                return host::fail(ENARF_ERR_ARG, "enarf_prepare: MLP parameter pointer of layer %d is null", i);
    }
    for (int j = 1; j < a.num_joints; ++j)
        if (a.parents[j] < 0 || a.parents[j] >= a.num_joints) return host::fail(ENARF_ERR_ARG, "enarf_prepare: parents[%d]=%d out of range", j, a.parents[j]);
    return 0;
}

extern "C" int enarf_prepare(const enarf_prepare_args *args, enarf_stream_t stream) {
    if (!args) return host::fail(ENARF_ERR_ARG, "enarf_prepare: args is null");
    const enarf_prepare_args &a = *args;
    if (int rc = check_prepare(a)) return rc;
    PrepareParams prm;
    prm.a = a;
    hipLaunchKernelGGL(prepare_kernel, dim3(3, a.B), dim3(256), 0, (hipStream_t)stream, prm);
    return host::check_launch("enarf_prepare");
}

extern "C" int enarf_mlp_unpack(const void *pack, float *dense, enarf_stream_t stream) {
    if (!pack || !dense) return host::fail(ENARF_ERR_ARG, "enarf_mlp_unpack: null pointer");
    hipLaunchKernelGGL(mlp_unpack_kernel, dim3(27), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float *>(pack), dense);
    return host::check_launch("enarf_mlp_unpack");
}

template <int MODE>
static int launch_query(const enarf_query_args &a, hipStream_t st) {
    const int pts_per_wg = 1024;   // 16 tiles of 16 points per wave (larger chunks were slower on the 667^3 sweep: imbalance)
    const long long wgs = (a.N + pts_per_wg - 1) / pts_per_wg;
    if (wgs * a.B > 0x7FFFFFFFll) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_query_fwd: too many points for one launch");
    const size_t lds = (size_t)lds_total_floats<MODE>(a.P) * 4;
    const bool dbg = a.dbg_canonical || a.dbg_weight;
    auto k = dbg ? query_kernel<MODE, true> : query_kernel<MODE, false>;
    hipLaunchKernelGGL(k, dim3((unsigned)(wgs * a.B)), dim3(256), lds, st, a, (int)wgs, pts_per_wg);
    return host::check_launch("enarf_query_fwd");
}

static int check_common(const char *who, int B, int P, int H, int W, int mode, const void *parts, const void *canon,
                        const void *feat, const void *mask, const void *pack) {
    if (B <= 0 || P <= 0 || P > ENARF_MAX_PARTS) return host::fail(ENARF_ERR_ARG, "%s: bad B=%d or P=%d (max %d parts)", who, B, P, ENARF_MAX_PARTS);
    if (H <= 0 || W <= 0) return host::fail(ENARF_ERR_ARG, "%s: bad plane size %dx%d", who, H, W);
    // the part-probability taps of a row are fetched as ONE pair of adjacent floats (load_row_pair): a row has two columns
    if (W < 2 || H < 2) return host::fail(ENARF_ERR_UNSUPPORTED, "%s: planes of %dx%d: at least 2x2 texels", who, H, W);
    // 32-bit byte offsets inside one image's planes (and 24-bit row * width products): 3 P part-probability planes of 4 B,
    // 3 feature planes of 128 B per texel
    if ((unsigned long long)3 * P * H * W * 4 >= (1ull << 32) || (unsigned long long)3 * H * W * 128 >= (1ull << 31) || H >= (1 << 23) || W >= (1 << 23))
        return host::fail(ENARF_ERR_UNSUPPORTED, "%s: planes of %dx%d with %d parts exceed the 32-bit in-image offsets", who, H, W, P);
    if (mode < 0 || mode > 3) return host::fail(ENARF_ERR_ARG, "%s: bad mlp_mode %d", who, mode);
    if (!parts || !canon || !feat || !mask || !pack) return host::fail(ENARF_ERR_ARG, "%s: null input pointer", who);
    return 0;
}

extern "C" int enarf_query_fwd(const enarf_query_args *args, enarf_stream_t stream) {
    if (!args) return host::fail(ENARF_ERR_ARG, "enarf_query_fwd: args is null");
    const enarf_query_args &a = *args;
    if (int rc = check_common("enarf_query_fwd", a.B, a.P, a.H, a.W, a.mlp_mode, a.parts, a.canonical_pose, a.feat_cl,
                              a.mask_planes, a.mlp_pack)) return rc;
    if (a.N < 0 || !a.density) return host::fail(ENARF_ERR_ARG, "enarf_query_fwd: bad N or null density");
    if (a.grid_D > 0) {
        if (a.grid_D < 3 || (a.grid_D & 1) == 0 || (long long)a.grid_D * a.grid_D * a.grid_D != a.N || a.N >= (1ll << 31))
            return host::fail(ENARF_ERR_ARG, "enarf_query_fwd: lattice mode needs an odd grid_D >= 3 with N == grid_D^3 < 2^31");
    } else if (!a.points) {
        return host::fail(ENARF_ERR_ARG, "enarf_query_fwd: points is null");
    }
    if (a.N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    switch (a.mlp_mode) {
        case ENARF_MLP_F32: return launch_query<ENARF_MLP_F32>(a, st);
        case ENARF_MLP_BF16X3: return launch_query<ENARF_MLP_BF16X3>(a, st);
        case ENARF_MLP_F16X3: return launch_query<ENARF_MLP_F16X3>(a, st);
        default: return launch_query<ENARF_MLP_BF16>(a, st);
    }
}

namespace enarf {
// zero the workspace header and run the ray set-up pre-pass (also used by the backward)
int launch_ray_setup(const enarf_render_args &a, hipStream_t st) {
    if (a.ws_epoch <= 0) {   // the caller does not count its calls: clear both headers with a fill
        hipError_t e = hipMemsetAsync(a.workspace, 0, 2 * kWsHeaderBytes, st);
        if (e != hipSuccess) return host::fail((int)e, "ray set-up: hipMemsetAsync(workspace) failed: %s", hipGetErrorString(e));
    }
    const int bpi = ws_setup_blocks(a.n);
    hipLaunchKernelGGL(ray_setup_kernel, dim3((unsigned)(bpi * a.B)), dim3(256), 0, st, a, bpi);
    return host::check_launch("ray set-up");
}
int device_cus() {
    static int num_cus = 0;
    if (num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        num_cus = prop.multiProcessorCount;
    }
    return num_cus;
}
}  // namespace enarf

// Two march kernels share every stage (enarf_march.h, enarf_query.h, enarf_tasks.h) and produce the same bits:
//   render_kernel   one workgroup (4 waves) marches one ray at a time, 3 workgroups per CU, three barriers per ray;
//   march_kernel    one workgroup (12 waves) per CU, several rays in flight, 16-sample tiles claimed as tasks.
// ENARF_TASK_MARCH: 0 / 1 force one of them (A/B builds); 2 (product) picks by shape, from measurements on MI355X
// (DESIGN.md 3.1): the task march wins where a pass has more tiles than a 4-wave workgroup has waves (Nc or Nf > 64:
// 0.31 vs 0.36 ms at 128^2, 72 + 96), ties at 48 + 64 (0.225 vs 0.225) and loses on batches, where every change of
// image drains its pipeline (8 frames: 2.60 vs 1.79 ms).
#ifndef ENARF_TASK_MARCH
#define ENARF_TASK_MARCH 2
#endif
#ifndef ENARF_TASK_WAVES
#define ENARF_TASK_WAVES 12
#endif
#ifndef ENARF_TASK_SLOTS
#define ENARF_TASK_SLOTS 6
#endif

template <int MODE, int SPL>
static int launch_task_march(const enarf_render_args &a, hipStream_t st, bool with_setup) {
    constexpr int NW = ENARF_TASK_WAVES;
    const int num_cus = device_cus();
    if (num_cus <= 0) return host::fail((int)hipGetLastError(), "enarf_render_fwd: cannot query the device");
    const long long total = (long long)a.B * a.n;
    // rays in flight per workgroup: enough tiles for every wave (3-4 per ray and stage), never more than there are rays
    int nslots = ENARF_TASK_SLOTS;
    long long wgs = num_cus;
    if (wgs * nslots > total) {          // few rays: spread them over the CUs first
        nslots = (int)((total + wgs - 1) / wgs);
        if (nslots < 1) nslots = 1;
        wgs = (total + nslots - 1) / nslots;
    }
    const size_t lds = (size_t)tasks_lds_floats<MODE>(a.P, nslots) * 4;
    auto kern = march_kernel<MODE, SPL, NW>;
    static bool attr_set = false;        // per instantiation: allow more than the default 64 KB of dynamic LDS
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return host::fail((int)e, "enarf_render_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_set = true;
    }
    if (with_setup)
        if (int rc = launch_ray_setup(a, st)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(NW * 64), lds, st, a, nslots, host::status_word(true));
    return host::check_launch("enarf_render_fwd");
}

template <int MODE, int SPL>
static int launch_render(const enarf_render_args &a, hipStream_t st, bool with_setup) {
#if ENARF_TASK_MARCH == 1
    return launch_task_march<MODE, SPL>(a, st, with_setup);
#elif ENARF_TASK_MARCH == 2
    if (a.march == ENARF_MARCH_TASK || (a.march == ENARF_MARCH_AUTO && SPL == 2 && a.B == 1))
        return launch_task_march<MODE, SPL>(a, st, with_setup);
#endif
    // persistent grid: as many workgroups as stay resident (3 per CU at <= 168 VGPRs and ~37 KB LDS), never more
    // than there are rays
    const int num_cus = device_cus();
    if (num_cus <= 0) return host::fail((int)hipGetLastError(), "enarf_render_fwd: cannot query the device");
    const long long total = (long long)a.B * a.n;
    long long wgs = (long long)num_cus * ENARF_RENDER_WAVES_PER_SIMD;
    if (wgs > total) wgs = total;
    const size_t lds = (size_t)lds_total_floats<MODE>(a.P) * 4;
    if (with_setup)
        if (int rc = launch_ray_setup(a, st)) return rc;
    hipLaunchKernelGGL((render_kernel<MODE, SPL>), dim3((unsigned)wgs), dim3(256), lds, st, a);
    return host::check_launch("enarf_render_fwd");
}

static int check_render(const enarf_render_args &a) {
    if (int rc = check_common("enarf_render_fwd", a.B, a.P, a.H, a.W, a.mlp_mode, a.parts, a.canonical_pose, a.feat_cl,
                              a.mask_planes, a.mlp_pack)) return rc;
    if (a.n <= 0 || !a.image_coord || !a.inv_intrinsics || !a.color || !a.mask || !a.disparity)
        return host::fail(ENARF_ERR_ARG, "enarf_render_fwd: bad n or null ray/output pointer");
    if (a.Nc < 2 || a.Nf < 2) return host::fail(ENARF_ERR_ARG, "enarf_render_fwd: Nc and Nf must be >= 2 (got %d, %d)", a.Nc, a.Nf);
    if (a.Nc > kMaxSamples || a.Nf > kMaxSamples)
        return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_render_fwd: Nc=%d / Nf=%d: more than %d samples per pass is not implemented",
                          a.Nc, a.Nf, kMaxSamples);
    if (a.dbg_depth_min && (!a.dbg_depth_max || !a.dbg_ray_valid))
        return host::fail(ENARF_ERR_ARG, "enarf_render_fwd: dbg_depth_min needs dbg_depth_max and dbg_ray_valid");
    if ((long long)((a.n + 63) / 64) * a.B > 0x7FFFFFFFll || (long long)a.n * a.B > 0x7FFFFFF0ll) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_render_fwd: more than 2^31 rays in one launch");
    if (!a.workspace) return host::fail(ENARF_ERR_ARG, "enarf_render_fwd: workspace is null (enarf_render_workspace_bytes() bytes of device memory)");
    if (a.march < ENARF_MARCH_AUTO || a.march > ENARF_MARCH_TASK) return host::fail(ENARF_ERR_ARG, "enarf_render_fwd: bad march %d (ENARF_MARCH_*)", a.march);
    return 0;
}

static int dispatch_march(const enarf_render_args &a, hipStream_t st, bool with_setup) {
    const bool wide = a.Nc > 64 || a.Nf > 64;      // two samples per lane in the lane = sample stages
    switch (a.mlp_mode) {
        case ENARF_MLP_F32: return wide ? launch_render<ENARF_MLP_F32, 2>(a, st, with_setup) : launch_render<ENARF_MLP_F32, 1>(a, st, with_setup);
        case ENARF_MLP_BF16X3: return wide ? launch_render<ENARF_MLP_BF16X3, 2>(a, st, with_setup) : launch_render<ENARF_MLP_BF16X3, 1>(a, st, with_setup);
        case ENARF_MLP_F16X3: return wide ? launch_render<ENARF_MLP_F16X3, 2>(a, st, with_setup) : launch_render<ENARF_MLP_F16X3, 1>(a, st, with_setup);
        default: return wide ? launch_render<ENARF_MLP_BF16, 2>(a, st, with_setup) : launch_render<ENARF_MLP_BF16, 1>(a, st, with_setup);
    }
}

// ---- a batch with one tri-plane per frame is marched in groups of frames -------------------------------------------------------
// 16 such frames in ONE launch march 14 % slower than two launches of 8 (profiles/r02_sweep.log: 68.7 M rays/s against
// 79.7 M; 64 frames 61.5 M): with two or more frames per XCD band the lists interleave their rays, an XCD's 4 MB L2 serves two
// frames' texels alternately (L2 hit rate 0.79 against 0.88, 9.0 GB of fabric traffic per launch against 0.83 GB compulsory)
// and workgroups restage the MLP pack at every change of image. So the host cuts such a batch into groups of (by default) 8
// frames - one frame per XCD - and issues one pre-march / march pair per group: views of the caller's arguments with offset
// pointers, a queue header pair per group inside the workspace, the near / far planes reduced ONCE over the whole batch and
// the sampler's ray ids counted batch-wide, so the results do not depend on the grouping, bit for bit.
// frames [b0, b0 + nb) of a render call as a call of its own
static enarf_render_args render_view(const enarf_render_args &a, int b0, int nb, void *ws, const float *near_far) {
    enarf_render_args v = a;
    const long long n = a.n, Nc = a.Nc, Nf = a.Nf;
    v.B = nb;
    v.image_coord = off(a.image_coord, b0 * 3 * n);
    v.inv_intrinsics = off(a.inv_intrinsics, (long long)b0 * 9);
    v.parts = off(a.parts, (long long)b0 * a.P * kPartStride);
    v.feat_cl = off(a.feat_cl, b0 * a.feat_batch_stride);
    v.mask_planes = off(a.mask_planes, b0 * a.mask_batch_stride);
    v.mlp_pack = a.mlp_pack ? reinterpret_cast<const char *>(a.mlp_pack) + (size_t)b0 * kPackBytes : nullptr;
    v.bins = off(a.bins, b0 * n * Nf);
    v.color = off(a.color, b0 * 3 * n);
    v.mask = off(a.mask, b0 * n);
    v.disparity = off(a.disparity, b0 * n);
    v.fine_weights = off(a.fine_weights, b0 * n * (Nf - 1));
    v.fine_depth = off(a.fine_depth, b0 * n * Nf);
    v.dbg_depth_min = off(a.dbg_depth_min, b0 * n);
    v.dbg_depth_max = off(a.dbg_depth_max, b0 * n);
    v.dbg_ray_valid = off(a.dbg_ray_valid, b0 * n);
    v.dbg_coarse_density = off(a.dbg_coarse_density, b0 * n * Nc);
    v.dbg_fine_density = off(a.dbg_fine_density, b0 * n * Nf);
    v.dbg_fine_color = off(a.dbg_fine_color, b0 * 3 * n * Nf);
    v.dbg_fine_valid = off(a.dbg_fine_valid, b0 * n * Nf);
    v.dbg_bins = off(a.dbg_bins, b0 * n * Nf);
    v.workspace = ws;
    v.near_far = near_far;
    v.ray_id_base = a.ray_id_base + (unsigned long long)b0 * (unsigned long long)n;
    v.group_frames = nb;           // a view is one launch
    return v;
}
static enarf_prepare_args prepare_view(const enarf_prepare_args &p, int b0, int nb, int P) {
    enarf_prepare_args v = p;
    v.B = nb;
    v.pose_to_camera = off(p.pose_to_camera, (long long)b0 * p.num_joints * 16);
    v.bone_length = off(p.bone_length, (long long)b0 * (p.num_joints - 1));
    v.z_rend = off(p.z_rend, (long long)b0 * p.style_dim);
    v.parts = off(p.parts, (long long)b0 * P * kPartStride);
    v.mlp_pack = p.mlp_pack ? reinterpret_cast<char *>(p.mlp_pack) + (size_t)b0 * kPackBytes : nullptr;
    return v;
}

extern "C" int enarf_near_far(const float *parts, int B, int P, float *out, enarf_stream_t stream) {
    if (!parts || !out || B <= 0 || P <= 0 || P > ENARF_MAX_PARTS) return host::fail(ENARF_ERR_ARG, "enarf_near_far: bad arguments");
    enarf_prepare_args unused = {};
    hipLaunchKernelGGL(near_far_kernel<false>, dim3(1), dim3(256), 0, (hipStream_t)stream, parts, unused, B, P, out);
    return host::check_launch("enarf_near_far");
}

extern "C" size_t enarf_render_workspace_bytes(int B, int n) {
    if (B <= 0 || n <= 0) return 0;
    // one launch, or any grouping of the B frames (a slice per group) + the near / far planes of a grouped call
    return ((ws_total_bytes(B, n) + 255) & ~(size_t)255) + (size_t)B * kWsGroupSlack + 256;
}
static float *ws_near_far_slot(const enarf_render_args &a) { return ws_near_far_slot(a.workspace, a.B, a.n); }

extern "C" int enarf_render_fwd(const enarf_render_args *args, enarf_stream_t stream) {
    if (!args) return host::fail(ENARF_ERR_ARG, "enarf_render_fwd: args is null");
    if (int rc = check_render(*args)) return rc;
    const enarf_render_args &a = *args;
    hipStream_t st = (hipStream_t)stream;
    const GroupPlan pl = plan_groups(a.B, a.feat_batch_stride, a.group_frames);
    if (pl.groups == 1) return dispatch_march(a, st, true);
    const float *nf = a.near_far;
    if (!nf) {
        float *slot = ws_near_far_slot(a);
        if (int rc = enarf_near_far(a.parts, a.B, a.P, slot, stream)) return rc;
        nf = slot;
    }
    for (int g = 0; g < pl.groups; ++g) {
        const enarf_render_args v = render_view(a, pl.first(g), pl.size(g), reinterpret_cast<char *>(a.workspace) + ws_slice_off(pl, g, a.n), nf);
        if (int rc = dispatch_march(v, st, true)) return rc;
    }
    return 0;
}

static int launch_pre_march(const enarf_prepare_args &p, const float *tri_nchw, float *feat_cl, int tri_B, int channels_total,
                            const enarf_render_args &r, hipStream_t st) {
    if (r.ws_epoch <= 0) {   // the caller does not count its calls: clear both headers with a fill
        hipError_t e = hipMemsetAsync(r.workspace, 0, 2 * kWsHeaderBytes, st);
        if (e != hipSuccess) return host::fail((int)e, "enarf_render_step_fwd: hipMemsetAsync(workspace) failed: %s", hipGetErrorString(e));
    }
    PreParams q;
    q.prep = p; q.rend = r; q.tri = tri_nchw; q.feat_cl = feat_cl; q.tri_B = tri_B; q.ch_total = channels_total;
    q.H = r.H; q.W = r.W;
    q.n_pack = tri_nchw ? ((r.W + 63) / 64) * r.H * tri_B * 3 : 0;
    q.n_prep = 3 * p.B;
    q.bpi = ws_setup_blocks(r.n);
    const long long blocks = (long long)q.n_pack + q.n_prep + (long long)q.bpi * r.B;
    if (blocks > 0x7FFFFFFFll) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_render_step_fwd: too many blocks");
    hipLaunchKernelGGL(pre_march_kernel, dim3((unsigned)blocks), dim3(256), 0, st, q);
    return host::check_launch("enarf_render_step_fwd(pre-march)");
}

extern "C" int enarf_render_step_fwd(const enarf_prepare_args *prep, const float *tri_nchw, float *feat_cl, int tri_B,
                                     int channels_total, const enarf_render_args *render, int phases, enarf_stream_t stream) {
    if (!prep || !render) return host::fail(ENARF_ERR_ARG, "enarf_render_step_fwd: args is null");
    if (int rc = check_prepare(*prep)) return rc;
    if (int rc = check_render(*render)) return rc;
    const enarf_prepare_args &p = *prep;
    const enarf_render_args &r = *render;
    const int P = (p.origin_location == ENARF_ORIGIN_CENTER_HEAD) ? p.num_joints : p.num_joints - 1;
    if (!p.parts || !p.mlp_pack || p.parts != r.parts || p.mlp_pack != r.mlp_pack || p.B != r.B || P != r.P)
        return host::fail(ENARF_ERR_ARG, "enarf_render_step_fwd: prepare outputs and render inputs must be the same buffers / sizes");
    if (tri_nchw && (!feat_cl || tri_B <= 0 || channels_total < 3 * ENARF_FEAT_DIM))
        return host::fail(ENARF_ERR_ARG, "enarf_render_step_fwd: bad tri-plane arguments");
    if (!(phases & ENARF_STEP_ALL) || (phases & ~ENARF_STEP_ALL)) return host::fail(ENARF_ERR_ARG, "enarf_render_step_fwd: bad phases %d", phases);
    hipStream_t st = (hipStream_t)stream;
    const GroupPlan pl = plan_groups(r.B, r.feat_batch_stride, r.group_frames);
    if (pl.groups == 1) {
        if (phases & ENARF_STEP_PRE)
            if (int rc = launch_pre_march(p, tri_nchw, feat_cl, tri_B, channels_total, r, st)) return rc;
        return (phases & ENARF_STEP_MARCH) ? dispatch_march(r, st, false) : 0;
    }
    // groups of frames: per-frame tri-planes, so tri_B == B when the planes are re-laid here
    if (tri_nchw && tri_B != r.B) return host::fail(ENARF_ERR_ARG, "enarf_render_step_fwd: %d tri-planes for a batch of %d frames with feat_batch_stride != 0", tri_B, r.B);
    const float *nf = r.near_far;
    if (!nf) {
        float *slot = ws_near_far_slot(r);
        if (phases & ENARF_STEP_PRE) {   // from the raw poses, in the arithmetic of the part frames (they are not written yet)
            hipLaunchKernelGGL(near_far_kernel<true>, dim3(1), dim3(256), 0, st, (const float *)nullptr, p, r.B, P, slot);
            if (int rc = host::check_launch("enarf_render_step_fwd(near / far)")) return rc;
        }
        nf = slot;                       // a march-only call finds what its pre-march call left there
    }
    const size_t tri_frame = (size_t)channels_total * r.H * r.W, cl_frame = (size_t)3 * r.H * r.W * kFeat;
    // STEP_ALL: group by group (a group's freshly re-laid planes are still in the Infinity Cache when it is marched);
    // a single phase: that phase for every group
    for (int g = 0; g < pl.groups; ++g) {
        const int b0 = pl.first(g), nb = pl.size(g);
        const enarf_render_args v = render_view(r, b0, nb, reinterpret_cast<char *>(r.workspace) + ws_slice_off(pl, g, r.n), nf);
        if (phases & ENARF_STEP_PRE) {
            const enarf_prepare_args pv = prepare_view(p, b0, nb, P);
            if (int rc = launch_pre_march(pv, tri_nchw ? tri_nchw + b0 * tri_frame : nullptr, feat_cl ? feat_cl + b0 * cl_frame : nullptr, nb,
                                          channels_total, v, st)) return rc;
        }
        if (phases & ENARF_STEP_MARCH)
            if (int rc = dispatch_march(v, st, false)) return rc;
    }
    return 0;
}
