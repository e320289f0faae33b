// enarf_march.h - device pieces shared by the forward march (enarf_render.hip) and its backward
// (enarf_render_bwd.hip): LDS staging of one image's MLP pack / part frames, the density head, the ray records
// written by the set-up pre-pass and the XCD-affine ray queues.
#pragma once
#include "enarf_query.h"

namespace enarf {

// =================================================================================================
// LDS staging shared by the query and render kernels
// =================================================================================================
// dynamic LDS layout (floats): [mlp section][bias 144][parts P*20][canon P*12][scratch kScratchFloats]
constexpr int kScratchFloats = 1536;
template <int MODE>
__host__ __device__ constexpr int lds_mlp_floats() {
    return (MODE == ENARF_MLP_F32) ? PK_B1 : PKH_SHORTS / 2;
}
template <int MODE>
__host__ __device__ inline int lds_total_floats(int P) {
    return lds_mlp_floats<MODE>() + 144 + P * kLdsPartStride + P * kLdsCanonStride + kScratchFloats;
}

template <int MODE>
__device__ __forceinline__ void stage_common(float *lds, QueryCtx &S, float *&scratch, const void *pack_b,
                                             const float *parts_b, const float *canon_pose, int P, int tid,
                                             int nthreads) {
    float *l_mlp = lds;
    float *l_bias = l_mlp + lds_mlp_floats<MODE>();
    float *l_parts = l_bias + 144;
    float *l_canon = l_parts + P * kLdsPartStride;
    scratch = l_canon + P * kLdsCanonStride;
    const float *pf = reinterpret_cast<const float *>(pack_b);
    const f32x4 *src4 = (MODE == ENARF_MLP_F32) ? reinterpret_cast<const f32x4 *>(pf)
                        : (MODE == ENARF_MLP_F16X3) ? reinterpret_cast<const f32x4 *>(pf + PK_F32_FLOATS + PKH_SHORTS / 2)
                                                    : reinterpret_cast<const f32x4 *>(pf + PK_F32_FLOATS);
    f32x4 *dst4 = reinterpret_cast<f32x4 *>(l_mlp);
    for (int i = tid; i < lds_mlp_floats<MODE>() / 4; i += nthreads) dst4[i] = src4[i];
    for (int i = tid; i < 144; i += nthreads) l_bias[i] = pf[PK_B1 + i];
    for (int i = tid; i < P * kPartStride; i += nthreads)
        l_parts[(i / kPartStride) * kLdsPartStride + (i % kPartStride)] = parts_b[i];
    for (int i = tid; i < P * 12; i += nthreads) {   // (P,4,4) -> Rc row-major 9 + tc 3
        const int k = i / 12, e = i % 12;
        l_canon[i] = (e < 9) ? canon_pose[k * 16 + (e / 3) * 4 + (e % 3)] : canon_pose[k * 16 + (e - 9) * 4 + 3];
    }
    S.mlp = l_mlp;
    S.mlp_h = reinterpret_cast<const short *>(l_mlp);
    S.bias = l_bias;
    S.parts = l_parts;
    S.canon = l_canon;
}

// head: tanh colour, MyReLU * 10 density, density *= any_valid   (triplane_nerf.py:44-47, narf.py:271-274, :204)
__device__ __forceinline__ float density_head(float sigma_act, uint32_t bits, float wmax, int mult_w, int P) {
    float d = fmaxf(sigma_act, 0.0f);
    if (mult_w) {
        // max over ALL parts of the weight tensor; invalid parts sit at sigmoid(0)^3 = 0.125
        const float wm = (__popc(bits) < P) ? fmaxf(wmax, 0.125f) : wmax;
        d = d * (10.0f * wm);
    } else {
        d = d * 10.0f;
    }
    return bits ? d : 0.0f;
}

struct RayRec { float dmin, dmax; uint32_t cand, valid; };
__host__ __device__ inline size_t ws_records_off() { return 64; }
__host__ __device__ inline size_t ws_list_off(long long total_rays) { return 64 + (size_t)total_rays * sizeof(RayRec); }

// wave-private compaction of a part bit set into an LDS list; returns the count
__device__ __forceinline__ int build_cand_list(int *list, uint32_t set, int lane) {
    if (lane < 32 && ((set >> lane) & 1u)) list[__popc(set & ((1u << lane) - 1u))] = lane;
    return __popc(set);
}

// ---- XCD-affine ray queues ---------------------------------------------------------------------------------------
// The live-ray list (image order) is cut into 8 contiguous bands of equal length, one queue per XCD; a workgroup pulls
// from the band of the XCD it runs on (HW_REG_XCC_ID) and, when that is drained, steals from the next bands in turn.
// Each XCD then marches its own part of the image(s): its L2 holds that part's tri-plane texels only, instead of every
// XCD pulling the whole frame's footprint through its own L2 (8x the fabric reads, and the latency that goes with them).
// Placement only affects speed: any workgroup may serve any band. ENARF_QUEUE_BANDS=0: blocks of kQBlock entries dealt
// round-robin to the queues (every XCD sweeps the whole frame), no stealing needed.
#ifndef ENARF_QUEUE_BANDS
#define ENARF_QUEUE_BANDS 1
#endif
constexpr int kQBlock = 32;
constexpr int kQueues = 8;
__device__ __forceinline__ int xcc_id() {
    return (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xF);   // HW_REG_XCC_ID[3:0]
}
struct RayQueue {                 // used by one thread of the workgroup
    unsigned int *heads;          // kQueues counters in the workspace header
    long long total, band;
    int q, tries;
    __device__ __forceinline__ void init(unsigned int *wsh) {
        heads = wsh + 2;
        total = (long long)wsh[1];
        band = (total + kQueues - 1) / kQueues;
        q = xcc_id() & (kQueues - 1);
        tries = 0;
    }
    // next index into the live list, or -1 once every queue is drained
    __device__ __forceinline__ int pop() {
#if ENARF_QUEUE_BANDS
        while (tries < kQueues) {
            const long long lo = (long long)q * band;
            const long long len = (total - lo < band) ? total - lo : band;
            if (len > 0) {
                const unsigned int j = atomicAdd(heads + q, 1u);
                if ((long long)j < len) return (int)(lo + j);
            }
            q = (q + 1) & (kQueues - 1);
            tries += 1;
        }
        return -1;
#else
        const unsigned int j = atomicAdd(heads + q, 1u);
        const long long e = ((long long)(j / kQBlock) * kQueues + q) * kQBlock + (j % kQBlock);
        return e < total ? (int)e : -1;
#endif
    }
};


// host side, defined in enarf_render.hip
int launch_ray_setup(const enarf_render_args &a, hipStream_t st);
int device_cus();

}  // namespace enarf
