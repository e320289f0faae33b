// enarf_raysample.hip - mask_based_sampler on the device (libraries/NeRF/ray_sampler.py:7-39 of
// the reference): per image, the k largest values of  dilate_{(2r+1) x (2r+1)}(mask) + noise,  returned as flat pixel ids.
// The reference runs F.max_pool2d(129, stride 1, padding 64) - O(129^2) per pixel - then torch.topk; here the window
// maximum is separable (a row pass and a column pass through LDS, O(2 * 129) per pixel) and the selection is an exact
// radix select (three histogram passes over the order-preserving integer image of the float scores) inside ONE workgroup
// per image, so there is no global synchronisation and no sort of the h*w scores. gfx950 only.
#include "enarf_device.h"
#include "enarf_host.h"

namespace enarf {

// ---- separable window maximum -------------------------------------------------------------------------------------
// out[y][x] = max over |d| <= r of in[y][x + d] (ROW = true) or in[y + d][x] (ROW = false); positions outside the image do
// not take part (max_pool2d pads with -inf). One workgroup = one row segment (or column segment) of 256 outputs; the
// 256 + 2r inputs it needs go through LDS.
constexpr int kMaxRadius = 128;
template <bool ROW>
__global__ __launch_bounds__(256) void window_max_kernel(const float *__restrict__ in, float *__restrict__ out, int h, int w,
                                                         int radius, const float *__restrict__ add) {
    __shared__ float tile[256 + 2 * kMaxRadius];
    const int b = blockIdx.z, line = blockIdx.y, seg = blockIdx.x * 256, tid = threadIdx.x;
    const int len = ROW ? w : h;                       // length of the line being filtered
    const size_t img = (size_t)b * h * w;
    const float ninf = -__builtin_huge_valf();
    for (int i = tid; i < 256 + 2 * radius; i += 256) {
        const int p = seg - radius + i;
        float v = ninf;
        if (p >= 0 && p < len) v = ROW ? in[img + (size_t)line * w + p] : in[img + (size_t)p * w + line];
        tile[i] = v;
    }
    __syncthreads();
    const int p = seg + tid;
    if (p >= len) return;
    float m = ninf;
    for (int d = 0; d <= 2 * radius; ++d) m = fmaxf(m, tile[tid + d]);
    const size_t o = ROW ? img + (size_t)line * w + p : img + (size_t)p * w + line;
    out[o] = add ? m + add[o] : m;                     // the column pass adds the noise: score = dilated mask + U[0, 1)
}

// ---- exact top-k of one image's scores: radix select + compaction, one workgroup per image ---------------------------
__device__ __forceinline__ unsigned order_key(float f) {      // monotone float -> uint (NaN sorts above everything)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
constexpr int kSelThreads = 1024, kSelBins = 2048, kTieCap = 1024;
__global__ __launch_bounds__(kSelThreads) void topk_select_kernel(const float *__restrict__ score, long long *__restrict__ out_idx,
                                                                   int n, int k) {
    __shared__ unsigned hist[kSelBins];
    __shared__ unsigned s_prefix, s_mask, s_want, s_cnt, s_neq, s_late;
    __shared__ int ties[kTieCap];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *sc = score + (size_t)b * n;
    long long *out = out_idx + (size_t)b * k;
    if (tid == 0) { s_prefix = 0u; s_mask = 0u; s_want = (unsigned)k; s_cnt = 0u; s_neq = 0u; s_late = 0u; }
    __syncthreads();
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = shifts[pass], nb = 1 << widths[pass];
        for (int i = tid; i < kSelBins; i += kSelThreads) hist[i] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix, mask = s_mask;
        for (int i = tid; i < n; i += kSelThreads) {
            const unsigned u = order_key(sc[i]);
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & (unsigned)(nb - 1)], 1u);
        }
        __syncthreads();
        if (tid < 64) {      // wave 0: the bin that holds the `want`-th largest key, scanning from the top
            const int per = nb / 64;                      // 32 or 16 bins per lane; lane 63 owns the highest bins
            unsigned mine = 0;
            for (int j = 0; j < per; ++j) mine += hist[tid * per + j];
            // suffix sum over lanes: above[l] = sum of counts of lanes > l
            unsigned incl = mine;
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_down(incl, d);
                if (tid + d < 64) incl += o;
            }
            const unsigned above = incl - mine;           // keys in higher lanes' bins
            const unsigned want = s_want;
            if (above < want && want <= above + mine) {   // exactly one lane
                unsigned acc = above;
                for (int j = per - 1; j >= 0; --j) {
                    const unsigned c = hist[tid * per + j];
                    if (acc + c >= want) {
                        s_prefix = prefix | ((unsigned)(tid * per + j) << shift);
                        s_mask = mask | ((unsigned)(nb - 1) << shift);
                        s_want = want - acc;              // how many keys of this bin are still wanted
                        break;
                    }
                    acc += c;
                }
            }
        }
        __syncthreads();
    }
    // every key > T is in; of the keys == T the `need` with the smallest index (torch.topk leaves ties unspecified)
    const unsigned T = s_prefix, need = s_want;
    for (int i = tid; i < n; i += kSelThreads) {
        const unsigned u = order_key(sc[i]);
        if (u > T) {
            out[atomicAdd(&s_cnt, 1u)] = i;
        } else if (u == T) {
            const unsigned e = atomicAdd(&s_neq, 1u);
            if (e < kTieCap) ties[e] = i;
        }
    }
    __syncthreads();
    const unsigned neq = s_neq, base = s_cnt;      // base == k - need
    if (neq <= (unsigned)kTieCap) {
        for (unsigned t = tid; t < neq; t += kSelThreads) {
            const int me = ties[t];
            unsigned rank = 0;
            for (unsigned j = 0; j < neq; ++j) rank += (ties[j] < me) ? 1u : 0u;
            if (rank < need) out[base + rank] = me;
        }
    } else {   // more equal scores than the tie buffer holds (degenerate noise): any `need` of them, in arrival order
        for (int i = tid; i < n; i += kSelThreads) {
            if (order_key(sc[i]) == T) {
                const unsigned e = atomicAdd(&s_late, 1u);
                if (e < need) out[base + e] = i;
            }
        }
    }
}

}  // namespace enarf

using namespace enarf;

extern "C" size_t enarf_mask_topk_workspace_bytes(int B, int h, int w) {
    if (B <= 0 || h <= 0 || w <= 0) return 0;
    return 2 * (size_t)B * h * w * sizeof(float);        // row-pass image + scores
}

extern "C" int enarf_mask_dilate_topk(const float *mask, const float *noise, long long *out_idx, int B, int h, int w, int k,
                                      int radius, void *workspace, enarf_stream_t stream) {
    if (!mask || !noise || !out_idx || !workspace) return host::fail(ENARF_ERR_ARG, "enarf_mask_dilate_topk: null pointer");
    if (B <= 0 || B > 65535 || h <= 0 || w <= 0 || h > 65535 || w > 65535 || (long long)h * w >= (1ll << 31))
        return host::fail(ENARF_ERR_ARG, "enarf_mask_dilate_topk: bad sizes B=%d h=%d w=%d", B, h, w);
    if (k <= 0 || (long long)k > (long long)h * w) return host::fail(ENARF_ERR_ARG, "enarf_mask_dilate_topk: k=%d outside [1, h*w]", k);
    if (radius < 0 || radius > kMaxRadius) return host::fail(ENARF_ERR_UNSUPPORTED, "enarf_mask_dilate_topk: radius %d > %d", radius, kMaxRadius);
    hipStream_t st = (hipStream_t)stream;
    float *rowmax = reinterpret_cast<float *>(workspace), *score = rowmax + (size_t)B * h * w;
    hipLaunchKernelGGL(window_max_kernel<true>, dim3((w + 255) / 256, h, B), dim3(256), 0, st, mask, rowmax, h, w, radius,
                       (const float *)nullptr);
    if (int rc = host::check_launch("enarf_mask_dilate_topk(rows)")) return rc;
    hipLaunchKernelGGL(window_max_kernel<false>, dim3((h + 255) / 256, w, B), dim3(256), 0, st, rowmax, score, h, w, radius, noise);
    if (int rc = host::check_launch("enarf_mask_dilate_topk(columns)")) return rc;
    hipLaunchKernelGGL(topk_select_kernel, dim3(B), dim3(kSelThreads), 0, st, score, out_idx, h * w, k);
    return host::check_launch("enarf_mask_dilate_topk(select)");
}
