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
constexpr int kScratchFloats = 1664;
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
        // max over ALL parts of the weight tensor; invalid parts sit at sigmoid(0)^3 = 0.125. mult_w == 2:
        // no_selector, every entry of the weight tensor is 1 / P (models/narf.py:133-134)
        const float wm = (mult_w == 2) ? 1.0f / (float)P : (__popc(bits) < P) ? fmaxf(wmax, 0.125f) : wmax;
        d = d * (10.0f * wm);
    } else {
        d = d * 10.0f;
    }
    return bits ? d : 0.0f;
}

// what the set-up pass leaves for the march: depth range, candidate parts, hit flag and the ray direction K^-1 [u v w]
struct __attribute__((aligned(16))) RayRec { float dmin, dmax; uint32_t cand, valid; float dx, dy, dz, pad; };
static_assert(sizeof(RayRec) == 32, "RayRec");
// workspace: [header kWsHeaderBytes][RayRec x B*n][ray lists: kNumLists x band_size entries]
// lists: (band q, cost class c) -> id q * kClasses + c, then one list per band of the live rays WITHOUT any candidate part
// (id ws_missed_list(q); batches only - a single image drops such rays, rendering.py:107-110): they need no query at all.
// header (u32): [1] rays filed for the march, [2] rays filed as missing every cube, [kWsCountsOff + id] entries in list id, [kWsHeadsOff + 16 * id] queue head of list id -
// one 64-B slot per head, so that the atomics of different lists do not share a cache line
constexpr int kQueues = 8;
#ifndef ENARF_NUM_CLASSES
#define ENARF_NUM_CLASSES 4
#endif
constexpr int kClasses = ENARF_NUM_CLASSES;
constexpr int kWsCountsOff = 16;                                     // u32 index of the list lengths
constexpr int kNumLists = kQueues * (kClasses + 1);
__host__ __device__ constexpr int ws_missed_list(int q) { return kQueues * kClasses + q; }
constexpr int kWsHeadsOff = (kWsCountsOff + kNumLists + 15) / 16 * 16;   // u32 index of the first queue head
constexpr int kWsHeadStride = 16;                                    // u32 per queue head slot
constexpr int kWsHeaderBytes = (kWsHeadsOff + kNumLists * kWsHeadStride) * 4;
// TWO headers, so that nobody has to clear one between launches: the launch with epoch k (> 0) uses header k & 1, which
// the launch with epoch k - 1 cleared while it was using the other one; epoch 0 = the library clears both with a fill
// first (always safe; what a caller that does not count its calls passes) and uses header 0.
__host__ __device__ inline size_t ws_header_off(int epoch) { return (epoch > 0 && (epoch & 1)) ? (size_t)kWsHeaderBytes : 0; }
__host__ __device__ inline size_t ws_other_header_off(int epoch) { return (epoch > 0 && (epoch & 1)) ? 0 : (size_t)kWsHeaderBytes; }
__host__ __device__ inline size_t ws_records_off() { return 2 * (size_t)kWsHeaderBytes; }
__host__ __device__ inline size_t ws_list_off(long long total_rays) { return ws_records_off() + (size_t)total_rays * sizeof(RayRec); }
// any workgroup of the launch that fills header `epoch` (all 256 threads): clear the other header for the next launch
__device__ __forceinline__ void ws_clear_other_header(void *workspace, int epoch, int tid) {
    if (epoch <= 0) return;
    unsigned int *o = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(workspace) + ws_other_header_off(epoch));
    for (int i = tid; i < kWsHeaderBytes / 4; i += 256) o[i] = 0u;
}
// Bands are cut in the padded ray index b * npad + ray (npad = 64 * set-up blocks per image), in multiples of 64, so
// that the 64 rays of one set-up block always fall into one band: one image -> its eighths; 8 images -> one image per
// band (XCD). ENARF_IMAGE_BANDS=0 (A/B only) cuts every image into eighths instead, band q of EVERY image in queue q -
// a smaller texel working set per XCD, but a workgroup then changes image (restages the MLP pack and part frames)
// B times per cost class: measured slower on every batch (8 frames 2.08 vs 1.80 ms, 16 distinct 4.69 vs 4.42).
#ifndef ENARF_IMAGE_BANDS
#define ENARF_IMAGE_BANDS 1
#endif
// rays per set-up block: 256 threads, kSetupLanes lanes per ray (enarf_render.hip ray_setup_block)
#ifndef ENARF_SETUP_LANES
#define ENARF_SETUP_LANES 8
#endif
constexpr int kSetupLanes = ENARF_SETUP_LANES;
constexpr int kSetupRays = 256 / kSetupLanes;
static_assert(kSetupLanes == 4 || kSetupLanes == 8, "set-up lanes per ray");
__host__ __device__ inline int ws_setup_blocks(int n) { return (n + kSetupRays - 1) / kSetupRays; }
__host__ __device__ inline long long ws_npad(int n) { return 64ll * ((n + 63) / 64); }
__host__ __device__ inline long long ws_image_band(int n) {          // rays of one image per band
    return 64ll * ((ws_npad(n) + 64ll * kQueues - 1) / (64ll * kQueues));
}
__host__ __device__ inline long long ws_band_size(int B, int n) {     // capacity of one (band, class) list
#if ENARF_IMAGE_BANDS
    const long long tot = (long long)B * ws_npad(n);
    return 64ll * ((tot + 64ll * kQueues - 1) / (64ll * kQueues));
#else
    return (long long)B * ws_image_band(n);
#endif
}
// band of set-up block `blk` of image b
__host__ __device__ inline int ws_band_of(int B, int n, int b, int blk) {
#if ENARF_IMAGE_BANDS
    return (int)(((long long)b * ws_npad(n) + (long long)kSetupRays * blk) / ws_band_size(B, n));
#else
    return (int)(((long long)kSetupRays * blk) / ws_image_band(n));
#endif
}
__host__ __device__ inline size_t ws_total_bytes(int B, int n) {
    return ws_list_off((long long)B * n) + (size_t)kNumLists * (size_t)ws_band_size(B, n) * sizeof(uint32_t);
}
// cost class of a live ray from the number of candidate parts on its marched segment (which tracks the ray's gather
// rounds closely: correlation 0.98 on the bench frame): 0 = heaviest
__device__ __forceinline__ int ray_cost_class(uint32_t cand) {
    const int pc = __popc(cand);
#if ENARF_NUM_CLASSES == 8        // measured: 4 % slower than 4 classes; 2 classes are on par with 4
    return pc >= 12 ? 0 : pc >= 10 ? 1 : pc >= 8 ? 2 : pc >= 6 ? 3 : pc >= 4 ? 4 : pc == 3 ? 5 : pc == 2 ? 6 : 7;
#elif ENARF_NUM_CLASSES == 2
    return pc >= 6 ? 0 : 1;
#else
    return pc >= 10 ? 0 : pc >= 6 ? 1 : pc >= 3 ? 2 : 3;
#endif
}

// wave-private compaction of a part bit set into an LDS list; returns the count
__device__ __forceinline__ int build_cand_list(int *list, uint32_t set, int lane) {
    if (lane < 32 && ((set >> lane) & 1u)) list[__popc(set & ((1u << lane) - 1u))] = lane;
    return __popc(set);
}

// ---- XCD-affine, heaviest-first ray queues --------------------------------------------------------------------------
// Every image is cut into 8 contiguous row bands (ws_image_band), one per XCD; within a band the set-up pass files every
// live ray (of every image of the batch) under its cost class: kQueues x kClasses lists, each with its own queue head. A workgroup works through the
// classes heaviest first; within a class it pulls from the band of the XCD it runs on (HW_REG_XCC_ID) and, when that
// list is drained, from the other bands' lists of the same class in turn.
//  * heaviest first, chip-wide: ray cost spans 10x (1 .. 14 gather rounds per tile) and the heavy rays sit in a few
//    bands (the torso). Every workgroup helps with them before anyone starts on the cheap rays, so when the queues
//    drain the rays still in flight are cheap ones and the launch's tail is one cheap ray long;
//  * bands: within a class each XCD marches its own part of the image(s) first, so its L2 mostly holds that part's
//    tri-plane texels instead of every XCD pulling the whole frame's footprint through its own L2.
// Placement and order only affect speed: any workgroup may march any ray.
__device__ __forceinline__ int xcc_id() {
    return (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xF);   // HW_REG_XCC_ID[3:0]
}
// LDS of the queue (ints, 16-byte aligned): 2 slots x [ray id, -, -, -, RayRec (8)], the kQueues x kClasses list lengths,
// the cursor (band, tries, class)
constexpr int kQSlotInts = 12;
constexpr int kQCountsOff = 2 * kQSlotInts;
constexpr int kQCursorOff = kQCountsOff + kQueues * kClasses;
constexpr int kQueueLdsInts = kQCursorOff + 3;
struct RayQueue {
    unsigned int *heads;          // queue heads in the workspace header
    const uint32_t *lists;
    const RayRec *recs;
    int *l_q;
    long long band;
    int home;
    // every thread; ends with a barrier. The list lengths are final (written by the set-up pass of an earlier launch):
    // staged once, so that a pop never loads from the header (a load from a line that is being hit by atomics from the
    // whole chip takes tens of microseconds).
    __device__ __forceinline__ void init(void *workspace, int epoch, int B, int n, int *lds_ints, int tid) {
        unsigned int *wsh = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(workspace) + ws_header_off(epoch));
        heads = wsh + kWsHeadsOff;
        recs = reinterpret_cast<const RayRec *>(reinterpret_cast<const char *>(workspace) + ws_records_off());
        lists = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(workspace) + ws_list_off((long long)B * n));
        l_q = lds_ints;
        band = ws_band_size(B, n);
        home = xcc_id() & (kQueues - 1);
        if (tid < kQueues * kClasses) l_q[kQCountsOff + tid] = (int)wsh[kWsCountsOff + tid];
        if (tid == 0) { l_q[kQCursorOff] = home; l_q[kQCursorOff + 1] = 0; l_q[kQCursorOff + 2] = 0; }
        __syncthreads();
    }
    // ONE thread (any - the cursor lives in LDS): take the next ray off the queues and leave its id and its set-up
    // record in slot; id -1 once every list is drained. The whole dependent chain (atomic -> list entry -> record) runs
    // here, on a wave that has nothing else to do, so the other waves find the record in LDS.
    __device__ __forceinline__ void pop(int slot) {
        int lid = -1, idx = 0;
        int *cur = l_q + kQCursorOff;
        int q = cur[0], tries = cur[1], cls = cur[2];
        while (cls < kClasses) {
            while (tries < kQueues) {
                const int l = q * kClasses + cls;
                const unsigned int len = (unsigned int)l_q[kQCountsOff + l];
                if (len != 0) {
                    const unsigned int j = atomicAdd(heads + l * kWsHeadStride, 1u);
                    if (j < len) { lid = l; idx = (int)j; break; }
                }
                q = (q + 1) & (kQueues - 1);
                tries += 1;
            }
            if (lid >= 0) break;
            cls += 1;
            q = home;
            tries = 0;
        }
        cur[0] = q; cur[1] = tries; cur[2] = cls;
        int *d = l_q + slot * kQSlotInts;
        if (lid < 0) { d[0] = -1; return; }
        const uint32_t rid = lists[(size_t)lid * (size_t)band + (size_t)idx];
        const RayRec r = recs[rid];
        d[0] = (int)rid;
        *reinterpret_cast<RayRec *>(d + 4) = r;
    }
    // every thread, after a barrier that follows pop(slot)
    __device__ __forceinline__ int get(int slot) const { return l_q[slot * kQSlotInts]; }
    __device__ __forceinline__ RayRec rec(int slot) const { return *reinterpret_cast<const RayRec *>(l_q + slot * kQSlotInts + 4); }
};


// host side, defined in enarf_render.hip
int launch_ray_setup(const enarf_render_args &a, hipStream_t st);
int device_cus();

// ---- host: a batch with one tri-plane per frame is processed in groups of frames (enarf_render.hip, "marched in groups") --------
constexpr int kDefaultGroupFrames = 8;
constexpr size_t kWsGroupSlack = 32768;      // per frame: header pair + list rounding of a group's workspace slice (bound)

struct GroupPlan {
    int groups, base, extra;      // `extra` groups of base + 1 frames first, then groups of `base`
    int first(int g) const { return g * base + (g < extra ? g : extra); }
    int size(int g) const { return base + (g < extra ? 1 : 0); }
};
inline GroupPlan plan_groups(int B, long long feat_batch_stride, int group_frames) {
    const int G = group_frames > 0 ? group_frames : kDefaultGroupFrames;
    GroupPlan pl;
    pl.groups = (feat_batch_stride == 0 || B <= G) ? 1 : (B + G - 1) / G;
    pl.base = B / pl.groups;
    pl.extra = B % pl.groups;
    return pl;
}
inline size_t ws_slice_bytes(int nb, int n) { return (ws_total_bytes(nb, n) + 255) & ~(size_t)255; }
// byte offset of group g's slice; the two near / far floats of a grouped call live behind the last slice
inline size_t ws_slice_off(const GroupPlan &pl, int g, int n) {
    const int big = g < pl.extra ? g : pl.extra;
    return (size_t)big * ws_slice_bytes(pl.base + 1, n) + (size_t)(g - big) * ws_slice_bytes(pl.base, n);
}

template <typename T>
inline T *off(T *p, long long elems) { return p ? p + elems : p; }

// the two near / far floats of a grouped call: the last 256 bytes of the workspace
inline float *ws_near_far_slot(void *workspace, int B, int n) {
    return reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + enarf_render_workspace_bytes(B, n) - 256);
}

}  // namespace enarf
