// enarf_gan_ops.hip - the two element-level ops the reference's 2-D GAN networks take from an un-vendored submodule
// (libraries/custom_stylegan2/net.py:12-14 imports FusedLeakyReLU / fused_leaky_relu and Blur /
// Upsample from rosinality/stylegan2-pytorch, whose CUDA extensions `fused_bias_act` and `upfirdn2d` are absent from the
// reference checkout): written here for gfx950 from their published definitions, not from that code.
//   * bias + leaky ReLU + gain in one pass (and its derivative form, which is the same pass with the sign taken from the
//     forward output - enough for every order of derivative, R1 needs the second: libraries/gan/loss.py:25-31);
//   * upfirdn2d: zero-insertion up-sampling, padding / cropping, a small FIR filter, decimation - one pass, the input tile
//     staged through LDS. Its adjoint is the same op with the filter flipped and the factors swapped (host layer).
// Both are HBM-bound stencils / maps: 8 B per element moved for bias_act, (1 / up^2 + 1 / down^2 ... ) for upfirdn2d.
#include "enarf_device.h"
#include "enarf_host.h"
#include "enarf_march.h"
#include <type_traits>

namespace enarf {

typedef float gf32x4 __attribute__((ext_vector_type(4)));

// out = gain * lrelu(x + bias[c])                      (ref == nullptr)
// out = x * gain * (ref > 0 ? 1 : slope)               (ref = the forward OUTPUT: sign(out) == sign(x + bias), gain > 0)
// element e of a contiguous (outer, C, inner) array has channel (e / inner) % C; VEC = 4 needs inner % 4 == 0
template <int VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(const float *__restrict__ x, const float *__restrict__ bias,
                                                       const float *__restrict__ ref, float *__restrict__ out, long long n_vec,
                                                       int C, long long inner_vec, float slope, float gain) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) {
        const int c = (int)((i / inner_vec) % C);
        if (VEC == 4) {
            gf32x4 v = reinterpret_cast<const gf32x4 *>(x)[i];
            if (ref) {
                const gf32x4 r = reinterpret_cast<const gf32x4 *>(ref)[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = v[k] * gain * (r[k] > 0.0f ? 1.0f : slope);
            } else {
                const float b = bias ? bias[c] : 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float t = v[k] + b; v[k] = gain * (t > 0.0f ? t : t * slope); }
            }
            reinterpret_cast<gf32x4 *>(out)[i] = v;
        } else {
            float v = x[i];
            if (ref) {
                v = v * gain * (ref[i] > 0.0f ? 1.0f : slope);
            } else {
                const float t = v + (bias ? bias[c] : 0.0f);
                v = gain * (t > 0.0f ? t : t * slope);
            }
            out[i] = v;
        }
    }
}

// ---- upfirdn2d ---------------------------------------------------------------------------------------------------------
// y[oy][ox] = sum_{i, j} kf[i][j] * P[oy * DOWN + i][ox * DOWN + j],  P[u][v] = x[(u - py0) / UP][(v - px0) / UP] where both
// quotients are exact and inside the image, else 0;  kf = the filter flipped in both axes (a true convolution).
// One workgroup = a 64 x 32 tile of one plane's output; the input rows / columns the tile can reach go through LDS once
// (rows dealt to the waves, columns to the lanes: no index division). A thread owns 8 vertically adjacent outputs of one
// column. For the 4 x 4 filters the networks use (KH = KW = 4, UP = 1) it slides down its column once: each LDS row it
// touches is read ONCE into 4 registers and feeds every output whose window holds it - 11 (DOWN = 1) or 18 (DOWN = 2) row
// reads for 8 outputs instead of 32, everything unrolled, the taps in SGPRs. Other filter sizes and the zero-stuffing
// up-sampler (which touches 2 x 2 taps per output) take the generic loop.
constexpr int kUfMaxTaps = 8;                  // filter extent per axis
constexpr int kUfLanes = 64;                   // a tile is 64 * XPT columns x 4 * PER rows of outputs (a thread: XPT columns 64 apart, PER adjacent rows)
struct UpfirParams {
    const float *x;
    float *y;
    long long planes;
    int H, W, OH, OW, kh, kw, px0, py0;
    int ex, ey;                                // EXT kernels: output columns / rows beyond the last tile column / row that it takes on as well (<= 8)
    int tstride;                               // floats per row of the LDS tile (sized by the launch for its up / down / filter:
                                               // a 4-tap blur at down = 1 needs 11 KB, not the 39 KB of the worst case - at 4
                                               // workgroups per CU the kernel was bound by the bytes it kept in flight)
    float kf[kUfMaxTaps * kUfMaxTaps];         // flipped filter, row-major kh x kw
};
__device__ __forceinline__ int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

constexpr int kUfExt = 8;                      // a remainder of up to 8 output columns / rows is absorbed by the last tile (EXT kernels)
template <int UP, int DOWN, int KH, int KW, int PER, int XPT = 1, bool EXT = false>          // KH = KW = 0: filter extents at run time
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const UpfirParams p) {
    extern __shared__ float tile[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ts = p.tstride;
    constexpr int kUfPer = PER, kUfTH = 4 * PER, kUfTW = kUfLanes * XPT;
    static_assert(XPT == 1 || (UP == 1 && KH > 0), "two columns per thread: the sliding-column form only");
    const int kh = KH ? KH : p.kh, kw = KW ? KW : p.kw;
    const int ox0 = blockIdx.x * kUfTW, oy0 = blockIdx.y * kUfTH;
    // EXT: a 129-wide map would leave a tile column with ONE valid column, and a partly filled tile costs a full one (the kernel
    // is bound by per-tile latency: 0.199 ms against 0.096 for 128 x 128). The last tile column / row takes the remainder on
    // instead: a few more staged columns / rows, and a few outputs per thread computed from LDS after the tile's own.
    const int ex = (EXT && blockIdx.x == gridDim.x - 1) ? p.ex : 0, ey = (EXT && blockIdx.y == gridDim.y - 1) ? p.ey : 0;
    // rows / columns of x the tile can reach: u in [oy0 * DOWN, (oy0 + TH - 1) * DOWN + kh - 1], row = (u - py0) / UP
    const int iy_min = floor_div(oy0 * DOWN - p.py0 + UP - 1, UP), iy_max = floor_div((oy0 + kUfTH - 1 + ey) * DOWN + kh - 1 - p.py0, UP);
    const int ix_min = floor_div(ox0 * DOWN - p.px0 + UP - 1, UP), ix_max = floor_div((ox0 + kUfTW - 1 + ex) * DOWN + kw - 1 - p.px0, UP);
    const int nr = iy_max - iy_min + 1, nc = ix_max - ix_min + 1;          // <= the launch's tile rows, tstride by construction
    const int ox = ox0 + lane, oyb = oy0 + wave * kUfPer;
    // loads per thread that cover the largest tile of this instantiation (filters up to 8 taps)
    constexpr int XR = EXT ? kUfExt : 0;
    constexpr int MAXR = ((4 * PER - 1 + XR) * DOWN + kUfMaxTaps - 1) / UP + 1, MAXC = ((kUfTW - 1 + XR) * DOWN + kUfMaxTaps - 1) / UP + 1;
    constexpr int NLD = (MAXR * MAXC + 255) / 256;
    const int total = nr * nc;
    const unsigned magic = (1u << 22) / (unsigned)nc + 1u;          // i / nc == (i * magic) >> 22 for i < 2^22 / nc (i < 9 400, nc <= 134)
    // A workgroup takes several planes (stride gridDim.z) and runs them as a two-stage pipeline: the loads of plane k + 1's
    // tile are issued into registers BEFORE plane k is filtered out of LDS, so their latency hides behind the filter and the
    // stores instead of being waited out once per tile (one plane per workgroup: 0.163 ms for the 128 x 128 blur).
    // Every load of a tile is issued before the first LDS write (a load-then-store loop waits out one memory latency per
    // row: 0.24 -> 0.40 ms when the tile grew from 35 to 67 rows); flat index over the tile, rows by multiply-shift.
    float stage[NLD];
    auto fetch = [&](long long plane) {
        const float *xp = p.x + (size_t)plane * p.H * p.W;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            const int r = (int)(((unsigned)i * magic) >> 22), c = i - r * nc;
            const int iy = iy_min + r, ix = ix_min + c;
            const bool in = i < total && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            stage[k] = in ? xp[(size_t)iy * p.W + ix] : 0.0f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            const int r = (int)(((unsigned)i * magic) >> 22), c = i - r * nc;
            if (i < total) tile[r * ts + c] = stage[k];
        }
    };
    if ((long long)blockIdx.z < p.planes) fetch(blockIdx.z);
    for (long long plane = blockIdx.z; plane < p.planes; plane += gridDim.z) {
        commit();
        __syncthreads();
        const bool more = plane + gridDim.z < p.planes;          // workgroup-uniform
        if (more) fetch(plane + gridDim.z);
        float *yp = p.y + (size_t)plane * p.OH * p.OW;
        if (UP == 1 && KH > 0) {
            const int r0 = oyb * DOWN - p.py0 - iy_min;
#pragma unroll
            for (int xi = 0; xi < XPT; ++xi) {
                const int oxx = ox + kUfLanes * xi, c0 = oxx * DOWN - p.px0 - ix_min;
                float acc[kUfPer];
#pragma unroll
                for (int j = 0; j < kUfPer; ++j) acc[j] = 0.0f;
#pragma unroll
                for (int rr = 0; rr < (kUfPer - 1) * DOWN + KH; ++rr) {
                    float v[KW ? KW : 1];
#pragma unroll
                    for (int kx = 0; kx < KW; ++kx) v[kx] = tile[(r0 + rr) * ts + c0 + kx];
#pragma unroll
                    for (int j = 0; j < kUfPer; ++j) {
                        const int ky = rr - j * DOWN;
                        if (ky >= 0 && ky < KH) {
#pragma unroll
                            for (int kx = 0; kx < KW; ++kx) acc[j] = fmaf(p.kf[ky * KW + kx], v[kx], acc[j]);
                        }
                    }
                }
                if (oxx < p.OW) {
#pragma unroll
                    for (int j = 0; j < kUfPer; ++j)
                        if (oyb + j < p.OH) yp[(size_t)(oyb + j) * p.OW + oxx] = acc[j];
                }
            }
        } else if (UP == 2 && DOWN == 1 && KH == 4 && KW == 4) {
            // the 2x up-sampler: an output sees 2 x 2 of the 4 x 4 taps (those whose zero-stuffed position holds a sample).
            // Which columns: by the parity of the thread's ox (taps picked once by 8 selects); which rows: by the parity of the
            // output row, which alternates down the column from a wave-uniform start - so the 16 outputs of a thread read 10
            // LDS rows x 2 columns once and share them.
            const int bx = ox - p.px0, kx0 = bx & 1, cA = ((bx + kx0) >> 1) - ix_min;
            const int by0 = oyb - p.py0, par = by0 & 1, r0 = ((by0 + par) >> 1) - iy_min;
            float w[4][2];
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                w[ky][0] = kx0 ? p.kf[ky * 4 + 1] : p.kf[ky * 4 + 0];
                w[ky][1] = kx0 ? p.kf[ky * 4 + 3] : p.kf[ky * 4 + 2];
            }
            constexpr int NROW = kUfPer / 2 + 2;
            float v[NROW][2];
#pragma unroll
            for (int r = 0; r < NROW; ++r) { v[r][0] = tile[(r0 + r) * ts + cA]; v[r][1] = tile[(r0 + r) * ts + cA + 1]; }
            float acc[kUfPer];
            auto column = [&](auto PAR) {          // PAR = parity of the first output row's position (wave-uniform)
#pragma unroll
                for (int j = 0; j < kUfPer; ++j) {
                    constexpr int P = decltype(PAR)::value;
                    const int ky0 = (P + j) & 1, ro = (j + P + ky0) / 2 - P;      // row of the first tap, relative to r0
                    acc[j] = w[ky0][0] * v[ro][0] + w[ky0][1] * v[ro][1] + w[ky0 + 2][0] * v[ro + 1][0] + w[ky0 + 2][1] * v[ro + 1][1];
                }
            };
            if (par) column(std::integral_constant<int, 1>{}); else column(std::integral_constant<int, 0>{});
            if (ox < p.OW) {
#pragma unroll
                for (int j = 0; j < kUfPer; ++j)
                    if (oyb + j < p.OH) yp[(size_t)(oyb + j) * p.OW + ox] = acc[j];
            }
        } else {
            for (int j = 0; j < kUfPer; ++j) {
                const int oy = oyb + j;
                if (ox < p.OW && oy < p.OH) {
                    // first tap whose up-sampled position holds a sample: (oy * DOWN + ky - py0) % UP == 0
                    const int by = oy * DOWN - p.py0, bx = ox * DOWN - p.px0;
                    const int ky0 = (UP == 1) ? 0 : (by & 1), kx0 = (UP == 1) ? 0 : (bx & 1);
                    float acc = 0.0f;
                    for (int ky = ky0; ky < kh; ky += UP) {
                        const int r = ((UP == 1) ? by + ky : (by + ky) >> 1) - iy_min;
                        for (int kx = kx0; kx < kw; kx += UP) {
                            const int c = ((UP == 1) ? bx + kx : (bx + kx) >> 1) - ix_min;
                            acc = fmaf(p.kf[ky * kw + kx], tile[r * ts + c], acc);
                        }
                    }
                    yp[(size_t)oy * p.OW + ox] = acc;
                }
            }
        }
        if (EXT && (ex | ey)) {          // the absorbed remainder: ex columns over all the tile's rows (+ ey), ey rows over its columns
            const int ncol = ex * (kUfTH + ey), nrow = ey * kUfTW;
            for (int idx = tid; idx < ncol + nrow; idx += 256) {
                int oxx, oy;
                if (idx < ncol) { const int q = idx / ex; oxx = ox0 + kUfTW + idx - q * ex; oy = oy0 + q; }
                else { const int j = idx - ncol, q = j / kUfTW; oxx = ox0 + j - q * kUfTW; oy = oy0 + kUfTH + q; }
                if (oxx < p.OW && oy < p.OH) {
                    const int by = oy * DOWN - p.py0, bx = oxx * DOWN - p.px0;
                    float acc = 0.0f;
                    for (int ky = (UP == 1) ? 0 : (by & 1); ky < kh; ky += UP) {
                        const int r = ((UP == 1) ? by + ky : (by + ky) >> 1) - iy_min;
                        for (int kx = (UP == 1) ? 0 : (bx & 1); kx < kw; kx += UP) {
                            const int c = ((UP == 1) ? bx + kx : (bx + kx) >> 1) - ix_min;
                            acc = fmaf(p.kf[ky * kw + kx], tile[r * ts + c], acc);
                        }
                    }
                    yp[(size_t)oy * p.OW + oxx] = acc;
                }
            }
        }
        __syncthreads();          // the next plane's tile replaces this one
    }
}

}  // namespace enarf

using namespace enarf;

extern "C" int enarf_bias_act(const float *x, const float *bias, const float *ref, float *out, long long outer, int C,
                              long long inner, float negative_slope, float gain, enarf_stream_t stream) {
    if (!x || !out) return host::fail(ENARF_ERR_ARG, "enarf_bias_act: null pointer");
    if (outer < 0 || C <= 0 || inner <= 0) return host::fail(ENARF_ERR_ARG, "enarf_bias_act: bad sizes outer=%lld C=%d inner=%lld", outer, C, inner);
    if (!(gain > 0.0f)) return host::fail(ENARF_ERR_ARG, "enarf_bias_act: gain must be positive (the derivative form reads the sign of the output)");
    const long long n = outer * C * inner;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uintptr_t al = (uintptr_t)x | (uintptr_t)out | (uintptr_t)ref;
    const bool vec = (inner % 4 == 0) && (al & 15) == 0;
    const long long n_vec = vec ? n / 4 : n;
    long long blocks = (n_vec + 255) / 256;
    const long long cap = (long long)(device_cus() > 0 ? device_cus() : 256) * 16;          // grid-stride beyond 16 workgroups per CU
    if (blocks > cap) blocks = cap;
    if (vec) hipLaunchKernelGGL(bias_act_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, x, bias, ref, out, n_vec, C, inner / 4, negative_slope, gain);
    else hipLaunchKernelGGL(bias_act_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, x, bias, ref, out, n_vec, C, inner, negative_slope, gain);
    return host::check_launch("enarf_bias_act");
}

extern "C" int enarf_upfirdn2d_out_size(int in_size, int taps, int up, int down, int pad0, int pad1) {
    if (in_size <= 0 || taps <= 0 || up <= 0 || down <= 0) return 0;
    const long long padded = (long long)in_size * up + pad0 + pad1 - taps;
    if (padded < 0) return 0;
    return (int)(padded / down + 1);
}

extern "C" int enarf_upfirdn2d(const float *x, float *out, long long planes, int H, int W, const float *kernel_host, int kh, int kw,
                               int up, int down, int pad_x0, int pad_x1, int pad_y0, int pad_y1, enarf_stream_t stream) {
    if (!x || !out || !kernel_host) return host::fail(ENARF_ERR_ARG, "enarf_upfirdn2d: null pointer");
    if (planes < 0 || H <= 0 || W <= 0) return host::fail(ENARF_ERR_ARG, "enarf_upfirdn2d: bad sizes planes=%lld H=%d W=%d", planes, H, W);
    if (kh <= 0 || kw <= 0 || kh > kUfMaxTaps || kw > kUfMaxTaps)
        return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_upfirdn2d: filter %d x %d outside 1..%d per axis", kh, kw, kUfMaxTaps);
    if (!((up == 1 || up == 2) && (down == 1 || down == 2)) || (up == 2 && down == 2))
        return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_upfirdn2d: up=%d down=%d (built: 1/1, 2/1, 1/2 - what Blur and Upsample use)", up, down);
    const int OH = enarf_upfirdn2d_out_size(H, kh, up, down, pad_y0, pad_y1), OW = enarf_upfirdn2d_out_size(W, kw, up, down, pad_x0, pad_x1);
    if (OH <= 0 || OW <= 0) return host::fail(ENARF_ERR_ARG, "enarf_upfirdn2d: empty output (%d x %d)", OH, OW);
    if (planes == 0) return 0;
    UpfirParams p;
    p.x = x; p.y = out; p.planes = planes; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.kh = kh; p.kw = kw; p.px0 = pad_x0; p.py0 = pad_y0;
    for (int i = 0; i < kUfMaxTaps * kUfMaxTaps; ++i) p.kf[i] = 0.0f;
    for (int i = 0; i < kh; ++i)
        for (int j = 0; j < kw; ++j) p.kf[i * kw + j] = kernel_host[(kh - 1 - i) * kw + (kw - 1 - j)];
    // The plain 4 x 4 blur takes tiles of 128 x 32 outputs (two columns per thread): with 64-column tiles every tile row of a
    // 128-wide map touched three 128-B lines for two lines of outputs (FETCH_SIZE: 1.54x the input, profiles/r03_gan2d_traffic.json).
    // Elsewhere 16 outputs per thread where the tile's input is small (down = 1): twice the loads in flight per wave - the
    // kernel is bound by the bytes it keeps in flight - at 20 KB of LDS; 8 at down = 2 (39 KB).
    const bool four = kh == 4 && kw == 4;          // the networks' [1, 3, 3, 1] filters: the unrolled sliding-column forms
    const bool wide = four && up == 1 && down == 1 && OW > kUfLanes;
    const int per = (down == 2 || wide) ? 8 : 16, th = 4 * per, tw = wide ? 2 * kUfLanes : kUfLanes;
    // a remainder of 1..8 columns / rows beyond whole tiles goes to the last tile column / row (EXT kernels) instead of a tile of its own
    const int rx = OW % tw, ry = OH % th;
    // (not for the decimating form: its tile is 39 KB already and the wider staging took it to one wave per SIMD: 0.25 ms against 0.17)
    p.ex = (four && up == 1 && down == 1 && rx > 0 && rx <= kUfExt && OW > tw) ? rx : 0;
    p.ey = (four && up == 1 && down == 1 && ry > 0 && ry <= kUfExt && OH > th) ? ry : 0;
    const bool ext = p.ex || p.ey;
    const unsigned gx = (unsigned)(p.ex ? OW / tw : (OW + tw - 1) / tw), gy = (unsigned)(p.ey ? OH / th : (OH + th - 1) / th);
    if (gy > 65535u) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_upfirdn2d: output height %d", OH);
    // planes per workgroup: enough workgroups for two rounds of 8 per CU, the rest of the planes in each one's pipeline (<= 8)
    long long ppw = planes * gx * gy / ((long long)(device_cus() > 0 ? device_cus() : 256) * 16);
    ppw = ppw < 1 ? 1 : (ppw > 8 ? 8 : ppw);
    long long gzl = (planes + ppw - 1) / ppw;
    const unsigned gz = (unsigned)(gzl < 65535 ? gzl : 65535);
    hipStream_t st = (hipStream_t)stream;
    // LDS tile of this configuration: the rows / columns of x that a th x tw output tile can reach, + 1 each for an unaligned start
    const int xr = ext ? kUfExt : 0;
    const int trows = ((th - 1 + xr) * down + kh - 1) / up + 2, tcols = ((tw - 1 + xr) * down + kw - 1) / up + 2;
    p.tstride = tcols | 1;
    const size_t lds = (size_t)trows * p.tstride * sizeof(float);
    if (up == 2 && four) hipLaunchKernelGGL((upfirdn2d_kernel<2, 1, 4, 4, 16>), dim3(gx, gy, gz), dim3(256), lds, st, p);
    else if (up == 2) hipLaunchKernelGGL((upfirdn2d_kernel<2, 1, 0, 0, 16>), dim3(gx, gy, gz), dim3(256), lds, st, p);
    else if (down == 2 && four) hipLaunchKernelGGL((upfirdn2d_kernel<1, 2, 4, 4, 8>), dim3(gx, gy, gz), dim3(256), lds, st, p);
    else if (down == 2) hipLaunchKernelGGL((upfirdn2d_kernel<1, 2, 0, 0, 8>), dim3(gx, gy, gz), dim3(256), lds, st, p);
    else if (wide && ext) hipLaunchKernelGGL((upfirdn2d_kernel<1, 1, 4, 4, 8, 2, true>), dim3(gx, gy, gz), dim3(256), lds, st, p);
    else if (wide) hipLaunchKernelGGL((upfirdn2d_kernel<1, 1, 4, 4, 8, 2>), dim3(gx, gy, gz), dim3(256), lds, st, p);
    else if (four && ext) hipLaunchKernelGGL((upfirdn2d_kernel<1, 1, 4, 4, 16, 1, true>), dim3(gx, gy, gz), dim3(256), lds, st, p);
    else if (four) hipLaunchKernelGGL((upfirdn2d_kernel<1, 1, 4, 4, 16>), dim3(gx, gy, gz), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((upfirdn2d_kernel<1, 1, 0, 0, 16>), dim3(gx, gy, gz), dim3(256), lds, st, p);
    return host::check_launch("enarf_upfirdn2d");
}
