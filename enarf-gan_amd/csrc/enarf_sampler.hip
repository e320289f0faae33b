// enarf_sampler.hip - the TriplaneSampler operator (a1) and the tri-plane re-layout kernel.
// Replaces cuda_extension/TriplaneSampler_kernel.cu (forward :13-92, backward :94-229) of the reference.
// gfx950 only.
#include "enarf_device.h"
#include "enarf_host.h"

namespace enarf {

// ---- grid_sample coordinate maps (ATen GridSampler.h semantics, all padding modes) -----------------
struct SamplerCfg {
    int interp, padding, align;
};

__device__ __forceinline__ float gs_unnormalize(float c, int size, int align, float &g) {
#pragma clang fp contract(off)
    if (align) { g = (float)(size - 1) / 2.0f; return ((c + 1.0f) / 2.0f) * (float)(size - 1); }
    g = (float)size / 2.0f;
    return ((c + 1.0f) * (float)size - 1.0f) / 2.0f;
}
__device__ __forceinline__ float gs_clip(float in, int size, float &g) {
    if (in <= 0.0f) { g = 0.0f; return 0.0f; }
    const float mx = (float)(size - 1);
    if (in >= mx) { g = 0.0f; return mx; }
    g = 1.0f;
    return in;
}
__device__ __forceinline__ float gs_reflect(float in, int twice_low, int twice_high, float &g) {
    if (twice_low == twice_high) { g = 0.0f; return 0.0f; }
    const float mn = (float)twice_low / 2.0f, span = (float)(twice_high - twice_low) / 2.0f;
    float sgn = 1.0f;
    in = in - mn;
    if (in < 0.0f) { sgn = -1.0f; in = -in; }
    const float extra = fmodf(in, span);
    const int flips = (int)floorf(in / span);
    if ((flips & 1) == 0) { g = sgn; return extra + mn; }
    g = -sgn;
    return span - extra + mn;
}
// source index and d(index)/d(grid coordinate)
__device__ __forceinline__ float gs_source_index(float c, int size, const SamplerCfg &cfg, float &gmult) {
    float g0, g1 = 1.0f, g2 = 1.0f;
    float v = gs_unnormalize(c, size, cfg.align, g0);
    if (cfg.padding == ENARF_PAD_BORDER) {
        v = gs_clip(v, size, g1);
    } else if (cfg.padding == ENARF_PAD_REFLECTION) {
        v = cfg.align ? gs_reflect(v, 0, 2 * (size - 1), g1) : gs_reflect(v, -1, 2 * size - 1, g1);
        v = gs_clip(v, size, g2);
    }
    gmult = g0 * g1 * g2;
    return v;
}

struct Tap2D {
    int o[4];        // clamped y*W + x of nw, ne, sw, se
    float w[4];      // bilinear weights, zero when out of bounds
    bool inb[4];
    float ix, iy, fx, fy;   // source index and its floor
};
__device__ __forceinline__ Tap2D gs_taps(float ix, float iy, int H, int W) {
    Tap2D t;
    t.ix = ix; t.iy = iy;
    t.fx = floorf(ix); t.fy = floorf(iy);
    const int x0 = (int)t.fx, y0 = (int)t.fy, x1 = x0 + 1, y1 = y0 + 1;
    const float ax1 = ix - t.fx, ax0 = (t.fx + 1.0f) - ix, ay1 = iy - t.fy, ay0 = (t.fy + 1.0f) - iy;
    const bool bx0 = (x0 >= 0) & (x0 < W), bx1 = (x1 >= 0) & (x1 < W);
    const bool by0 = (y0 >= 0) & (y0 < H), by1 = (y1 >= 0) & (y1 < H);
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    t.o[0] = cy0 * W + cx0; t.o[1] = cy0 * W + cx1; t.o[2] = cy1 * W + cx0; t.o[3] = cy1 * W + cx1;
    t.inb[0] = bx0 & by0; t.inb[1] = bx1 & by0; t.inb[2] = bx0 & by1; t.inb[3] = bx1 & by1;
    t.w[0] = t.inb[0] ? ax0 * ay0 : 0.0f;
    t.w[1] = t.inb[1] ? ax1 * ay0 : 0.0f;
    t.w[2] = t.inb[2] ? ax0 * ay1 : 0.0f;
    t.w[3] = t.inb[3] ? ax1 * ay1 : 0.0f;
    return t;
}

// ---- NCHW -> channel-last re-layout -------------------------------------------------------------------
// in: (B, in_ch_total, H, W), planes p = 0..2 at channels [p*C, (p+1)*C); out: [b][p][y][x][C].
// One workgroup moves 64 x-positions of one (b, p, y) row through LDS: coalesced 256-B reads along x,
// contiguous 64*C*4-B writes.
template <int C>
__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                   int in_ch_total, int H, int W) {
    __shared__ float tile[C * 65];
    pack_block<C>(in, out, in_ch_total, H, W, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x, tile);
}

// ---- direct NCHW forward: thread per point, channel loop outermost (no global read-modify-write) -------
// `separate`: the three planes' samples are written side by side, out (B, 3, C, n), instead of summed (the callers of
// sample_feature's "prod" reduction need them apart, sampling.py:43-48). `point_image`: per-point image index (then the grid
// has batch 1): the semantics of sample_feature's batch_idx without its side-by-side copy of the planes (sampling.py:34-38).
template <bool separate>
__global__ __launch_bounds__(256) void sample_fwd_direct(const float *__restrict__ in, const float *__restrict__ grid,
                                                         float *__restrict__ out, int C, int H, int W, long long n,
                                                         SamplerCfg cfg, const int *__restrict__ point_image, int n_images) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const float *g = grid + ((size_t)b * n + i) * 3;
    const float c3[3] = {g[0], g[1], g[2]};
    const size_t hw = (size_t)H * W;
    const int bi = point_image ? point_image[i] : b;
    float *ob = out + (size_t)b * (separate ? 3 : 1) * C * n + i;
    if ((unsigned)bi >= (unsigned)n_images) {      // an image id outside the batch samples nothing (the reference's
        for (int c = 0; c < (separate ? 3 : 1) * C; ++c) ob[(size_t)c * n] = 0.0f;      // side-by-side planes: zero padding)
        return;
    }
    const float *inb = in + (size_t)bi * 3 * C * hw;
    if (cfg.interp == ENARF_INTERP_BILINEAR) {
        Tap2D t[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            float gm;
            t[p] = gs_taps(gs_source_index(c3[p], W, cfg, gm), gs_source_index(c3[(p + 1) % 3], H, cfg, gm), H, W);
        }
        for (int c = 0; c < C; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float *pl = inb + ((size_t)p * C + c) * hw;
                float s = pl[t[p].o[0]] * t[p].w[0];
                s += pl[t[p].o[1]] * t[p].w[1];
                s += pl[t[p].o[2]] * t[p].w[2];
                s += pl[t[p].o[3]] * t[p].w[3];
                if (separate) ob[((size_t)p * C + c) * n] = s;
                acc += s;
            }
            if (!separate) ob[(size_t)c * n] = acc;
        }
    } else {   // nearest: the reference overwrites per plane, so the last plane (zx) wins (kernel.cu:76-90)
        float gm;
        const int xi = (int)roundf(gs_source_index(c3[2], W, cfg, gm));
        const int yi = (int)roundf(gs_source_index(c3[0], H, cfg, gm));
        const bool ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H);
        for (int c = 0; c < C; ++c)
            ob[(size_t)((separate ? 2 * C : 0) + c) * n] = ok ? inb[((size_t)2 * C + c) * hw + (size_t)yi * W + xi] : 0.0f;
        if (separate)
            for (int c = 0; c < 2 * C; ++c) ob[(size_t)c * n] = 0.0f;
    }
}

// ---- channel-last forward: LPP = C/8 lanes per point, each lane 8 channels (2 x 16-B loads per tap) ------
template <int LPP>
__global__ __launch_bounds__(256) void sample_fwd_cl(const float *__restrict__ cl, const float *__restrict__ grid,
                                                     float *__restrict__ out, int H, int W, long long n, SamplerCfg cfg) {
    constexpr int C = LPP * 8, PPW = 256 / LPP;    // points per workgroup
    const int tid = threadIdx.x, j = tid / LPP, g = tid % LPP;   // the LPP lanes of a point read one texel's C*4 contiguous bytes
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * PPW + j;
    if (i >= n) return;
    const float *gp = grid + ((size_t)b * n + i) * 3;
    const float c3[3] = {gp[0], gp[1], gp[2]};
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float gm;
        const Tap2D t = gs_taps(gs_source_index(c3[p], W, cfg, gm), gs_source_index(c3[(p + 1) % 3], H, cfg, gm), H, W);
        const float *base = cl + (((size_t)b * 3 + p) * H * W) * C + 8 * g;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 *q = reinterpret_cast<const f32x4 *>(base + (size_t)t.o[k] * C);
            const f32x4 v0 = q[0], v1 = q[1];
#pragma unroll
            for (int c = 0; c < 4; ++c) { s[c] += v0[c] * t.w[k]; s[4 + c] += v1[c] * t.w[k]; }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += s[c];
    }
    float *ob = out + ((size_t)b * C + 8 * g) * n + i;
#pragma unroll
    for (int c = 0; c < 8; ++c) ob[(size_t)c * n] = acc[c];
}

// ---- direct NCHW backward (true gradients of the forward above) -----------------------------------------
__global__ __launch_bounds__(256) void sample_bwd_direct(const float *__restrict__ gout, const float *__restrict__ in,
                                                         const float *__restrict__ grid, float *__restrict__ gin,
                                                         float *__restrict__ ggrid, int C, int H, int W, long long n,
                                                         SamplerCfg cfg, int separate, const int *__restrict__ point_image,
                                                         int n_images) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const float *g = grid + ((size_t)b * n + i) * 3;
    const float c3[3] = {g[0], g[1], g[2]};
    const size_t hw = (size_t)H * W;
    const int bi = point_image ? point_image[i] : b;
    if ((unsigned)bi >= (unsigned)n_images) {      // sampled nothing in the forward: no gradient to any plane or to the grid
        if (ggrid) {
            float *o = ggrid + ((size_t)b * n + i) * 3;
            o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
        }
        return;
    }
    const float *inb = in + (size_t)bi * 3 * C * hw;
    float *ginb = gin ? gin + (size_t)bi * 3 * C * hw : nullptr;
    const float *go = gout + (size_t)b * (separate ? 3 : 1) * C * n + i;
    float gg[3] = {0.0f, 0.0f, 0.0f};
    if (cfg.interp == ENARF_INTERP_BILINEAR) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            float gxm, gym;
            const float ix = gs_source_index(c3[p], W, cfg, gxm);
            const float iy = gs_source_index(c3[(p + 1) % 3], H, cfg, gym);
            const Tap2D t = gs_taps(ix, iy, H, W);
            const float ax1 = ix - t.fx, ax0 = (t.fx + 1.0f) - ix, ay1 = iy - t.fy, ay0 = (t.fy + 1.0f) - iy;
            float gix = 0.0f, giy = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float gO = go[(size_t)((separate ? p * C : 0) + c) * n];
                const size_t po = ((size_t)p * C + c) * hw;
                if (ginb) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (t.inb[k]) atomicAdd(ginb + po + t.o[k], t.w[k] * gO);
                }
                if (ggrid) {   // kernel.cu:170-202
                    const float *pl = inb + po;
                    if (t.inb[0]) { const float v = pl[t.o[0]]; gix -= v * ay0 * gO; giy -= v * ax0 * gO; }
                    if (t.inb[1]) { const float v = pl[t.o[1]]; gix += v * ay0 * gO; giy -= v * ax1 * gO; }
                    if (t.inb[2]) { const float v = pl[t.o[2]]; gix -= v * ay1 * gO; giy += v * ax0 * gO; }
                    if (t.inb[3]) { const float v = pl[t.o[3]]; gix += v * ay1 * gO; giy += v * ax1 * gO; }
                }
            }
            gg[p] += gxm * gix;
            gg[(p + 1) % 3] += gym * giy;
        }
    } else if (ginb) {   // nearest: only the plane the forward read (zx) receives gradient; grid gets none
        float gm;
        const int xi = (int)roundf(gs_source_index(c3[2], W, cfg, gm));
        const int yi = (int)roundf(gs_source_index(c3[0], H, cfg, gm));
        if ((xi >= 0) & (xi < W) & (yi >= 0) & (yi < H))
            for (int c = 0; c < C; ++c)
                atomicAdd(ginb + ((size_t)2 * C + c) * hw + (size_t)yi * W + xi, go[(size_t)((separate ? 2 * C : 0) + c) * n]);
    }
    if (ggrid) {
        float *o = ggrid + ((size_t)b * n + i) * 3;
        o[0] = gg[0]; o[1] = gg[1]; o[2] = gg[2];
    }
}

// ---- fast backward (bilinear, C = 32, caller's workspace): ONE LANE PER CHANNEL ----------------------------------------
// 32 lanes per point, two points per wave instruction: every atomic instruction adds into two whole 128-B lines of a
// channel-last gradient buffer (scattered 4-byte atomics into the NCHW planes - the direct kernel - run ~7x slower), and
// every read of the forward input for grad_grid is one whole line of the channel-last copy. grad_out is staged through
// LDS so that its (channel-major) rows are read coalesced. gcl is folded back into NCHW by enarf_triplane_unpack_add.
__global__ __launch_bounds__(256) void sample_bwd_cl32(const float *__restrict__ gout, const float *__restrict__ cl_in,
                                                       const float *__restrict__ grid, float *__restrict__ gcl,
                                                       float *__restrict__ ggrid, int H, int W, long long n, SamplerCfg cfg) {
    constexpr int C = 32;
    __shared__ float tile[C * 65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;
    const long long i0 = (long long)blockIdx.x * 64;
    {
        const int pj = tid & 63;
        const bool ok = i0 + pj < n;
        for (int cc = tid >> 6; cc < C; cc += 4)
            tile[cc * 65 + pj] = ok ? gout[((size_t)b * C + cc) * n + i0 + pj] : 0.0f;
    }
    __syncthreads();
    const size_t hw = (size_t)H * W;
    for (int it = 0; it < 8; ++it) {
        const int j = wave * 16 + 2 * it + half;
        const long long i = i0 + j;
        const bool valid = i < n;
        const float *gp = grid + ((size_t)b * n + (valid ? i : 0)) * 3;
        const float c3[3] = {gp[0], gp[1], gp[2]};
        const float gO = tile[c * 65 + j];
        float gg[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            float gxm, gym;
            const float ix = gs_source_index(c3[p], W, cfg, gxm);
            const float iy = gs_source_index(c3[(p + 1) % 3], H, cfg, gym);
            const Tap2D t = gs_taps(ix, iy, H, W);
            const float ax1 = ix - t.fx, ax0 = (t.fx + 1.0f) - ix, ay1 = iy - t.fy, ay0 = (t.fy + 1.0f) - iy;
            const size_t base = (((size_t)b * 3 + p) * hw) * C + c;
            if (gcl && valid) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (t.inb[k]) atomicAdd(gcl + base + (size_t)t.o[k] * C, t.w[k] * gO);
            }
            if (ggrid && valid) {   // kernel.cu:170-202
                float gix = 0.0f, giy = 0.0f;
                if (t.inb[0]) { const float v = cl_in[base + (size_t)t.o[0] * C]; gix -= v * ay0 * gO; giy -= v * ax0 * gO; }
                if (t.inb[1]) { const float v = cl_in[base + (size_t)t.o[1] * C]; gix += v * ay0 * gO; giy -= v * ax1 * gO; }
                if (t.inb[2]) { const float v = cl_in[base + (size_t)t.o[2] * C]; gix -= v * ay1 * gO; giy += v * ax0 * gO; }
                if (t.inb[3]) { const float v = cl_in[base + (size_t)t.o[3] * C]; gix += v * ay1 * gO; giy += v * ax1 * gO; }
                gg[p] += gxm * gix;
                gg[(p + 1) % 3] += gym * giy;
            }
        }
        if (ggrid) {   // sum over the 32 channel lanes of each point: row scans, then row 0 -> 1 and row 2 -> 3
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float v = gg[d];
                v += dpp_f<0x111, 0xF>(v);
                v += dpp_f<0x112, 0xF>(v);
                v += dpp_f<0x114, 0xF>(v);
                v += dpp_f<0x118, 0xF>(v);
                v += dpp_f<0x142, 0xA>(v);
                gg[d] = v;
            }
            if (valid && c == 31) {
                float *o = ggrid + ((size_t)b * n + i) * 3;
                o[0] = gg[0]; o[1] = gg[1]; o[2] = gg[2];
            }
        }
    }
}

// ---- deformation-field tri-plane producer (models/narf.py:40-58) -----------------------------------------------------------
// The reference warps the constant feature planes with F.grid_sample (bilinear, zeros, align_corners False) on the grid
// (pixel centre + flow) / (W/2) - 1, i.e. it samples plane p at (x + flow_x, y + flow_y), and concatenates the result NCHW.
// Here the source is the channel-last copy of the constant planes and the output is written channel-last directly - the
// layout the march reads - so a GAN frame never pays the NCHW -> channel-last re-layout. One lane per channel: every load,
// store and atomic of a half-wave is one whole 128-B texel line.
__device__ __forceinline__ Tap2D warp_taps(const float *__restrict__ flow, int b, int p, int x, int y, int H, int W) {
    const size_t hw = (size_t)H * W;
    const float fx = flow[((size_t)b * 6 + 2 * p) * hw + (size_t)y * W + x];
    const float fy = flow[((size_t)b * 6 + 2 * p + 1) * hw + (size_t)y * W + x];
    float ix, iy;
    {
#pragma clang fp contract(off)
        const float gx = ((float)x + 0.5f + fx) / (0.5f * (float)W) - 1.0f;      // narf.py:49-50 ("/ 128 - 1" at W = 256)
        const float gy = ((float)y + 0.5f + fy) / (0.5f * (float)H) - 1.0f;
        ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
        iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    }
    return gs_taps(ix, iy, H, W);
}

__global__ __launch_bounds__(256) void warp_fwd_kernel(const float *__restrict__ src_cl, const float *__restrict__ flow,
                                                       float *__restrict__ out_cl, int H, int W) {
    constexpr int C = 32;
    // 8 lanes per texel, 16 B each: the tap arithmetic is shared by 8 lanes instead of 32, loads and stores are dwordx4
    const int c4 = (threadIdx.x & 7) * 4, p = blockIdx.y, b = blockIdx.z;
    const long long t = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const long long hw = (long long)H * W;
    if (t >= hw) return;
    const int x = (int)(t % W), y = (int)(t / W);
    const Tap2D tp = warp_taps(flow, b, p, x, y, H, W);
    const float *sp = src_cl + (size_t)p * hw * C + c4;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                   // out-of-bounds taps weigh zero
        const f32x4 q = *reinterpret_cast<const f32x4 *>(sp + (size_t)tp.o[k] * C);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += tp.w[k] * q[i];
    }
    *reinterpret_cast<f32x4 *>(out_cl + (((size_t)b * 3 + p) * hw + t) * C + c4) = v;
}

__global__ __launch_bounds__(256) void warp_bwd_kernel(const float *__restrict__ g_out_cl, const float *__restrict__ src_cl,
                                                       const float *__restrict__ flow, float *__restrict__ g_src_cl,
                                                       float *__restrict__ g_flow, int H, int W) {
    constexpr int C = 32;
    const int c = threadIdx.x & 31, p = blockIdx.y, b = blockIdx.z;
    const long long t0 = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const long long hw = (long long)H * W;
    const bool valid = t0 < hw;                      // no early return: the reduction below is wave-wide DPP
    const long long t = valid ? t0 : hw - 1;
    const int x = (int)(t % W), y = (int)(t / W);
    const Tap2D tp = warp_taps(flow, b, p, x, y, H, W);
    const float g = valid ? g_out_cl[(((size_t)b * 3 + p) * hw + t) * C + c] : 0.0f;
    const size_t pbase = (size_t)p * hw * C + c;
    if (g_src_cl && valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tp.inb[k]) atomicAdd(g_src_cl + pbase + (size_t)tp.o[k] * C, tp.w[k] * g);
    }
    if (g_flow) {   // d out / d ix, iy as in the grid_sample backward; d ix / d flow_x = (W / 2) / (W / 2) = 1
        const float ax1 = tp.ix - tp.fx, ax0 = (tp.fx + 1.0f) - tp.ix, ay1 = tp.iy - tp.fy, ay0 = (tp.fy + 1.0f) - tp.iy;
        float gix = 0.0f, giy = 0.0f;
        if (tp.inb[0]) { const float v = src_cl[pbase + (size_t)tp.o[0] * C]; gix -= v * ay0 * g; giy -= v * ax0 * g; }
        if (tp.inb[1]) { const float v = src_cl[pbase + (size_t)tp.o[1] * C]; gix += v * ay0 * g; giy -= v * ax1 * g; }
        if (tp.inb[2]) { const float v = src_cl[pbase + (size_t)tp.o[2] * C]; gix -= v * ay1 * g; giy += v * ax0 * g; }
        if (tp.inb[3]) { const float v = src_cl[pbase + (size_t)tp.o[3] * C]; gix += v * ay1 * g; giy += v * ax1 * g; }
        float r[2] = {gix, giy};
#pragma unroll
        for (int d = 0; d < 2; ++d) {   // sum over the 32 channel lanes of the texel (row scans, then row 0 -> 1, 2 -> 3)
            float v = r[d];
            v += dpp_f<0x111, 0xF>(v);
            v += dpp_f<0x112, 0xF>(v);
            v += dpp_f<0x114, 0xF>(v);
            v += dpp_f<0x118, 0xF>(v);
            v += dpp_f<0x142, 0xA>(v);
            r[d] = v;
        }
        if (valid && c == 31) {
            g_flow[((size_t)b * 6 + 2 * p) * hw + t] = r[0];
            g_flow[((size_t)b * 6 + 2 * p + 1) * hw + t] = r[1];
        }
    }
}

}  // namespace enarf

using namespace enarf;

static int check_sampler(const char *who, const void *a, const void *b, const void *c, int B, int C, int H, int W,
                         long long n, int interp, int pad) {
    if (!a || !b || !c) return host::fail(ENARF_ERR_ARG, "%s: null pointer", who);
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || n < 0) return host::fail(ENARF_ERR_ARG, "%s: bad sizes B=%d C=%d H=%d W=%d n=%lld", who, B, C, H, W, n);
    if (interp != ENARF_INTERP_BILINEAR && interp != ENARF_INTERP_NEAREST) return host::fail(ENARF_ERR_ARG, "%s: bad interpolation_mode %d", who, interp);
    if (pad < 0 || pad > 2) return host::fail(ENARF_ERR_ARG, "%s: bad padding_mode %d", who, pad);
    if (B > 65535) return host::fail(ENARF_ERR_UNSUPPORTED, "%s: B > 65535", who);
    if ((size_t)3 * C * H * W >= (1ull << 31)) return host::fail(ENARF_ERR_UNSUPPORTED, "%s: one image's planes exceed 2^31 elements", who);
    return 0;
}

static bool cl_supported(int C) { return C == 8 || C == 16 || C == 32 || C == 64; }

extern "C" size_t enarf_triplane_sample_workspace_bytes(int B, int C, int H, int W) {
    if (!cl_supported(C) || B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * 3 * C * H * W * sizeof(float);
}
// backward fast path: channel-last copy of the input (for grad_grid) + channel-last gradient accumulator
extern "C" size_t enarf_triplane_sample_bwd_workspace_bytes(int B, int C, int H, int W) {
    if (C != ENARF_FEAT_DIM || B <= 0 || H <= 0 || W <= 0) return 0;
    return 2 * (size_t)B * 3 * C * H * W * sizeof(float);
}

template <int C>
static void launch_pack(const float *in, float *out, int B, int in_ch_total, int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(pack_kernel<C>, dim3((W + 63) / 64, H, B * 3), dim3(256), 0, st, in, out, in_ch_total, H, W);
}

extern "C" int enarf_triplane_pack(const float *tri_nchw, float *feat_cl, int B, int channels_total, int H, int W,
                                   enarf_stream_t stream) {
    if (!tri_nchw || !feat_cl) return host::fail(ENARF_ERR_ARG, "enarf_triplane_pack: null pointer");
    if (B <= 0 || H <= 0 || W <= 0 || channels_total < 3 * ENARF_FEAT_DIM || H > 65535 || B * 3 > 65535)
        return host::fail(ENARF_ERR_ARG, "enarf_triplane_pack: bad sizes B=%d channels=%d H=%d W=%d", B, channels_total, H, W);
    launch_pack<ENARF_FEAT_DIM>(tri_nchw, feat_cl, B, channels_total, H, W, (hipStream_t)stream);
    return host::check_launch("enarf_triplane_pack");
}

extern "C" int enarf_triplane_sample_fwd(const float *input, const float *grid, float *out, int B, int C, int H, int W,
                                         long long n_pts, int interp, int pad, int align_corners, void *workspace,
                                         enarf_stream_t stream) {
    if (int rc = check_sampler("enarf_triplane_sample_fwd", input, grid, out, B, C, H, W, n_pts, interp, pad)) return rc;
    if (n_pts == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const SamplerCfg cfg{interp, pad, align_corners ? 1 : 0};
    if (workspace && cl_supported(C) && interp == ENARF_INTERP_BILINEAR && H <= 65535) {
        float *cl = reinterpret_cast<float *>(workspace);
        switch (C) {
            case 8: launch_pack<8>(input, cl, B, 3 * C, H, W, st); break;
            case 16: launch_pack<16>(input, cl, B, 3 * C, H, W, st); break;
            case 32: launch_pack<32>(input, cl, B, 3 * C, H, W, st); break;
            default: launch_pack<64>(input, cl, B, 3 * C, H, W, st); break;
        }
        if (int rc = host::check_launch("enarf_triplane_sample_fwd(pack)")) return rc;
        const int lpp = C / 8, ppw = 256 / lpp;
        const dim3 grd((unsigned)((n_pts + ppw - 1) / ppw), B);
        switch (lpp) {
            case 1: hipLaunchKernelGGL(sample_fwd_cl<1>, grd, dim3(256), 0, st, cl, grid, out, H, W, n_pts, cfg); break;
            case 2: hipLaunchKernelGGL(sample_fwd_cl<2>, grd, dim3(256), 0, st, cl, grid, out, H, W, n_pts, cfg); break;
            case 4: hipLaunchKernelGGL(sample_fwd_cl<4>, grd, dim3(256), 0, st, cl, grid, out, H, W, n_pts, cfg); break;
            default: hipLaunchKernelGGL(sample_fwd_cl<8>, grd, dim3(256), 0, st, cl, grid, out, H, W, n_pts, cfg); break;
        }
        return host::check_launch("enarf_triplane_sample_fwd");
    }
    hipLaunchKernelGGL(sample_fwd_direct<false>, dim3((unsigned)((n_pts + 255) / 256), B), dim3(256), 0, st, input, grid, out,
                       C, H, W, n_pts, cfg, (const int *)nullptr, B);
    return host::check_launch("enarf_triplane_sample_fwd");
}

static int check_ex(const char *who, int B, int n_images, const int *point_image, int reduction) {
    if (reduction != ENARF_PLANES_SUM && reduction != ENARF_PLANES_SEPARATE) return host::fail(ENARF_ERR_ARG, "%s: bad plane reduction %d", who, reduction);
    if (n_images <= 0 || n_images > 65535) return host::fail(ENARF_ERR_ARG, "%s: bad n_images %d", who, n_images);
    if (point_image && B != 1) return host::fail(ENARF_ERR_ARG, "%s: point_image needs a grid of batch 1 (got %d)", who, B);
    if (!point_image && n_images != B) return host::fail(ENARF_ERR_ARG, "%s: n_images %d != B %d without point_image", who, n_images, B);
    return 0;
}

extern "C" int enarf_triplane_sample_ex_fwd(const float *input, const float *grid, float *out, int B, int C, int H, int W,
                                            long long n_pts, int interp, int pad, int align_corners, int reduction,
                                            const int *point_image, int n_images, enarf_stream_t stream) {
    if (int rc = check_sampler("enarf_triplane_sample_ex_fwd", input, grid, out, B, C, H, W, n_pts, interp, pad)) return rc;
    if (int rc = check_ex("enarf_triplane_sample_ex_fwd", B, n_images, point_image, reduction)) return rc;
    if (n_pts == 0) return 0;
    const SamplerCfg cfg{interp, pad, align_corners ? 1 : 0};
    const dim3 grd((unsigned)((n_pts + 255) / 256), B);
    if (reduction == ENARF_PLANES_SEPARATE)
        hipLaunchKernelGGL(sample_fwd_direct<true>, grd, dim3(256), 0, (hipStream_t)stream, input, grid, out, C, H, W, n_pts, cfg, point_image, n_images);
    else
        hipLaunchKernelGGL(sample_fwd_direct<false>, grd, dim3(256), 0, (hipStream_t)stream, input, grid, out, C, H, W, n_pts, cfg, point_image, n_images);
    return host::check_launch("enarf_triplane_sample_ex_fwd");
}

extern "C" int enarf_triplane_sample_ex_bwd(const float *grad_out, const float *input, const float *grid, float *grad_input,
                                            float *grad_grid, int B, int C, int H, int W, long long n_pts, int interp, int pad,
                                            int align_corners, int reduction, const int *point_image, int n_images,
                                            enarf_stream_t stream) {
    if (int rc = check_sampler("enarf_triplane_sample_ex_bwd", grad_out, input, grid, B, C, H, W, n_pts, interp, pad)) return rc;
    if (int rc = check_ex("enarf_triplane_sample_ex_bwd", B, n_images, point_image, reduction)) return rc;
    if (n_pts == 0 || (!grad_input && !grad_grid)) return 0;
    const SamplerCfg cfg{interp, pad, align_corners ? 1 : 0};
    hipLaunchKernelGGL(sample_bwd_direct, dim3((unsigned)((n_pts + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, grad_out,
                       input, grid, grad_input, grad_grid, C, H, W, n_pts, cfg, reduction == ENARF_PLANES_SEPARATE ? 1 : 0, point_image, n_images);
    return host::check_launch("enarf_triplane_sample_ex_bwd");
}

extern "C" int enarf_triplane_sample_bwd(const float *grad_out, const float *input, const float *grid, float *grad_input,
                                         float *grad_grid, int B, int C, int H, int W, long long n_pts, int interp, int pad,
                                         int align_corners, void *workspace, enarf_stream_t stream) {
    if (int rc = check_sampler("enarf_triplane_sample_bwd", grad_out, input, grid, B, C, H, W, n_pts, interp, pad)) return rc;
    if (n_pts == 0 || (!grad_input && !grad_grid)) return 0;
    const SamplerCfg cfg{interp, pad, align_corners ? 1 : 0};
    if (workspace && interp == ENARF_INTERP_BILINEAR && C == ENARF_FEAT_DIM && H <= 65535 && B * 3 <= 65535) {
        hipStream_t st = (hipStream_t)stream;
        const size_t plane_floats = (size_t)B * 3 * C * H * W;
        float *cl_in = reinterpret_cast<float *>(workspace), *gcl = cl_in + plane_floats;
        if (grad_grid) launch_pack<ENARF_FEAT_DIM>(input, cl_in, B, 3 * C, H, W, st);
        if (grad_input) {
            hipError_t e = hipMemsetAsync(gcl, 0, plane_floats * sizeof(float), st);
            if (e != hipSuccess) return host::fail((int)e, "enarf_triplane_sample_bwd: hipMemsetAsync failed: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(sample_bwd_cl32, dim3((unsigned)((n_pts + 63) / 64), B), dim3(256), 0, st, grad_out,
                           grad_grid ? cl_in : nullptr, grid, grad_input ? gcl : nullptr, grad_grid, H, W, n_pts, cfg);
        if (int rc = host::check_launch("enarf_triplane_sample_bwd")) return rc;
        return grad_input ? enarf_triplane_unpack_add(gcl, grad_input, B, 3 * C, H, W, stream) : 0;
    }
    hipLaunchKernelGGL(sample_bwd_direct, dim3((unsigned)((n_pts + 255) / 256), B), dim3(256), 0, (hipStream_t)stream,
                       grad_out, input, grid, grad_input, grad_grid, C, H, W, n_pts, cfg, 0, (const int *)nullptr, B);
    return host::check_launch("enarf_triplane_sample_bwd");
}

static int check_warp(const char *who, const void *a, const void *b, const void *c, int B, int H, int W) {
    if (!a || !b || !c) return host::fail(ENARF_ERR_ARG, "%s: null pointer", who);
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0) return host::fail(ENARF_ERR_ARG, "%s: bad sizes B=%d H=%d W=%d", who, B, H, W);
    if ((long long)H * W > (1ll << 28)) return host::fail(ENARF_ERR_UNSUPPORTED, "%s: plane too large", who);
    return 0;
}

extern "C" int enarf_triplane_warp_fwd(const float *src_cl, const float *flow, float *out_cl, int B, int H, int W,
                                       enarf_stream_t stream) {
    if (int rc = check_warp("enarf_triplane_warp_fwd", src_cl, flow, out_cl, B, H, W)) return rc;
    const unsigned xb = (unsigned)(((long long)H * W + 31) / 32);
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(xb, 3, B), dim3(256), 0, (hipStream_t)stream, src_cl, flow, out_cl, H, W);
    return host::check_launch("enarf_triplane_warp_fwd");
}

extern "C" int enarf_triplane_warp_bwd(const float *g_out_cl, const float *src_cl, const float *flow, float *g_src_cl,
                                       float *g_flow, int B, int H, int W, enarf_stream_t stream) {
    if (int rc = check_warp("enarf_triplane_warp_bwd", g_out_cl, src_cl, flow, B, H, W)) return rc;
    if (!g_src_cl && !g_flow) return 0;
    const unsigned xb = (unsigned)(((long long)H * W + 7) / 8);
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(xb, 3, B), dim3(256), 0, (hipStream_t)stream, g_out_cl, src_cl, flow, g_src_cl,
                       g_flow, H, W);
    return host::check_launch("enarf_triplane_warp_bwd");
}
