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
// One workgroup = a 64 x 16 tile of one plane's output; the input rows / columns the tile can reach go through LDS once.
constexpr int kUfMaxTaps = 8;                  // filter extent per axis
constexpr int kUfTW = 64, kUfTH = 16;
constexpr int kUfRows = kUfTH * 2 + kUfMaxTaps, kUfCols = kUfTW * 2 + kUfMaxTaps;      // the DOWN = 2, UP = 1 worst case
struct UpfirParams {
    const float *x;
    float *y;
    long long planes;
    int H, W, OH, OW, kh, kw, px0, py0;
    float kf[kUfMaxTaps * kUfMaxTaps];         // flipped filter, row-major kh x kw
};
__device__ __forceinline__ int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

template <int UP, int DOWN>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const UpfirParams p) {
    __shared__ float tile[kUfRows][kUfCols + 1];
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * kUfTW, oy0 = blockIdx.y * kUfTH;
    // rows / columns of x the tile can reach: u in [oy0 * DOWN, (oy0 + TH - 1) * DOWN + kh - 1], row = (u - py0) / UP
    const int iy_min = floor_div(oy0 * DOWN - p.py0 + UP - 1, UP), iy_max = floor_div((oy0 + kUfTH - 1) * DOWN + p.kh - 1 - p.py0, UP);
    const int ix_min = floor_div(ox0 * DOWN - p.px0 + UP - 1, UP), ix_max = floor_div((ox0 + kUfTW - 1) * DOWN + p.kw - 1 - p.px0, UP);
    const int nr = iy_max - iy_min + 1, nc = ix_max - ix_min + 1;          // <= kUfRows, kUfCols by construction
    for (long long plane = blockIdx.z; plane < p.planes; plane += gridDim.z) {
        const float *xp = p.x + (size_t)plane * p.H * p.W;
        for (int i = tid; i < nr * nc; i += 256) {
            const int r = i / nc, c = i - r * nc, iy = iy_min + r, ix = ix_min + c;
            tile[r][c] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? xp[(size_t)iy * p.W + ix] : 0.0f;
        }
        __syncthreads();
        const int ox = ox0 + (tid & 63);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + (tid >> 6) + 4 * j;
            if (ox < p.OW && oy < p.OH) {
                // first tap whose up-sampled position holds a sample: (oy * DOWN + ky - py0) % UP == 0
                const int by = oy * DOWN - p.py0, bx = ox * DOWN - p.px0;
                const int ky0 = (UP == 1) ? 0 : (by & 1), kx0 = (UP == 1) ? 0 : (bx & 1);
                float acc = 0.0f;
                for (int ky = ky0; ky < p.kh; ky += UP) {
                    const int r = ((UP == 1) ? by + ky : (by + ky) >> 1) - iy_min;
                    for (int kx = kx0; kx < p.kw; kx += UP) {
                        const int c = ((UP == 1) ? bx + kx : (bx + kx) >> 1) - ix_min;
                        acc = fmaf(p.kf[ky * p.kw + kx], tile[r][c], acc);
                    }
                }
                p.y[((size_t)plane * p.OH + oy) * p.OW + ox] = acc;
            }
        }
        __syncthreads();          // the next plane restages the tile
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
    const unsigned gx = (unsigned)((OW + kUfTW - 1) / kUfTW), gy = (unsigned)((OH + kUfTH - 1) / kUfTH);
    if (gy > 65535u) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_upfirdn2d: output height %d", OH);
    const unsigned gz = (unsigned)(planes < 65535 ? planes : 65535);
    hipStream_t st = (hipStream_t)stream;
    if (up == 2) hipLaunchKernelGGL((upfirdn2d_kernel<2, 1>), dim3(gx, gy, gz), dim3(256), 0, st, p);
    else if (down == 2) hipLaunchKernelGGL((upfirdn2d_kernel<1, 2>), dim3(gx, gy, gz), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((upfirdn2d_kernel<1, 1>), dim3(gx, gy, gz), dim3(256), 0, st, p);
    return host::check_launch("enarf_upfirdn2d");
}
