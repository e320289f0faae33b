// enarf_tasks.h - the ray march as a pool of tile tasks (round 2; the second march kernel beside the barrier-structured
// workgroup-per-ray render_kernel of enarf_render.hip - same stages, same bits; enarf_render_args.march picks one), the
// two serial ray stages both kernels share, and the pass over the rays that miss every cube.
//
// One persistent workgroup of NW wavefronts per CU keeps R rays in flight, each in an LDS "slot". A ray's life is a chain
//     pop -> C coarse tiles -> S2 (weights, importance samples) -> F fine tiles -> S4 (compositing, outputs) -> pop ...
// and every link that can run in parallel is a TASK any wave may take: the 16-sample query tiles. A wave that finishes
// the last tile of a stage runs the serial follow-up itself (S2, or S4 and the pop of the slot's next ray), so nobody ever
// waits at a barrier for somebody else's tile: the round-1 kernel spent 25 % of its wave time parked at its three
// workgroup barriers per ray (unequal tiles, one wave sampling while three wait) and ended every launch with one ray per
// workgroup still running; here the unit of imbalance - and of the tail - is a tile, and the MLP pack is staged once per
// CU instead of once per 4 waves (LDS: ~32 KB + R x 5.3 KB instead of 3 x 37 KB).
//
// Synchronisation is LDS-only and wave-uniform (lane 0 issues the atomics, the result is broadcast):
//   ctl[slot]   one word {generation, stage, tiles in the stage, next tile}; a tile is claimed by compare-and-swap of the
//               whole word, so a stale view of a recycled slot can never claim; the word is re-published by plain stores
//               only while nothing can be claimed from it (next == tiles);
//   done[slot]  completed tiles of the stage; the wave whose increment completes the stage owns the follow-up.
// Writes of tile results precede the done increment, writes of a stage's inputs precede the ctl store (workgroup-scope
// release / acquire fences; the LDS serves one CU's requests in order).
//
// Several images (B > 1): the per-image data (MLP pack, part frames, plane pointers) is staged for ONE image at a time.
// A popped ray of another image parks in its slot ("pending"); when the last ray in flight retires, that wave restages
// the context for the parked image and publishes the parked rays. The lists are in image order, so this is rare.
#pragma once
#include "enarf_march.h"

namespace enarf {

constexpr int kMaxSamples = 128;          // samples per pass (two per lane in the lane = sample stages above 64)

// ---- LDS slot of one ray in flight (32-bit words) ---------------------------------------------------------------------
constexpr int SL_CTL = 0;                 // control word, see pack_ctl
constexpr int SL_DONE = 1;                // completed tiles of the current stage
constexpr int SL_RID = 2;                 // global ray id (image * n + ray)
constexpr int SL_NCAND = 3;
constexpr int SL_REC = 4;                 // RayRec, 8 words (16-byte aligned)
constexpr int SL_CAND = 12;               // candidate part ids, 32 ints
constexpr int SL_SKIP = 44;               // early-termination flags per fine tile, 8 ints
constexpr int SL_NEXT_STATE = 52;         // the slot's NEXT ray, popped ahead by the wave that ran S2: 0 none, 3 pop in flight, 1 held, 2 queues drained
constexpr int SL_NEXT_RID = 53;
constexpr int SL_NEXT_REC = 56;           // RayRec, 8 words (16-byte aligned)
constexpr int SL_CH = 64;                 // coarse: sigma head [128]
constexpr int SL_CBITS = SL_CH + kMaxSamples;
constexpr int SL_CWMAX = SL_CBITS + kMaxSamples;
constexpr int SL_BINS = SL_CWMAX + kMaxSamples;
constexpr int SL_FH = SL_BINS + kMaxSamples;            // fine: head [4][128]
constexpr int SL_FBITS = SL_FH + 4 * kMaxSamples;
constexpr int SL_FWMAX = SL_FBITS + kMaxSamples;
constexpr int kSlotWords = SL_FWMAX + kMaxSamples;      // 1344 words = 5376 B
static_assert(kSlotWords % 4 == 0 && SL_REC % 4 == 0 && SL_NEXT_REC % 4 == 0, "slot alignment");

// stages of the control word
constexpr unsigned ST_NONE = 0, ST_COARSE = 1, ST_FINE = 2;
__device__ __forceinline__ unsigned pack_ctl(unsigned gen, unsigned stage, unsigned tiles, unsigned next) {
    return (gen << 18) | (stage << 16) | (tiles << 8) | next;
}
__device__ __forceinline__ unsigned ctl_next(unsigned c) { return c & 0xFFu; }
__device__ __forceinline__ unsigned ctl_tiles(unsigned c) { return (c >> 8) & 0xFFu; }
__device__ __forceinline__ unsigned ctl_stage(unsigned c) { return (c >> 16) & 3u; }
__device__ __forceinline__ unsigned ctl_gen(unsigned c) { return c >> 18; }

// ---- workgroup-shared scheduler state (32-bit words), after the slots -------------------------------------------------
constexpr int SH_QCUR = 0;                // queue cursor, packed cls * 16 + tries (single word: no torn update)
constexpr int SH_DEAD = 1;                // slots whose chain has ended (queues drained)
constexpr int SH_INFLIGHT = 2;            // slots that hold a published ray of the staged image, or are being popped
constexpr int SH_PENDING = 3;             // slots that hold a popped ray of another image
constexpr int SH_CTXB = 4;                // image whose context is staged
constexpr int SH_SWITCHING = 5;           // a context switch is in progress
constexpr int SH_ERROR = 6;               // watchdog: a wave found no work for kIdleLimit polls although chains are alive
constexpr int SH_FIRST = 7;               // the workgroup's first ray (decides which image is staged at start)
constexpr int SH_QCOUNTS = 8;             // list lengths [kQueues * kClasses]
constexpr int SH_BTAB = SH_QCOUNTS + kQueues * kClasses;       // Nc + 1 bin edges (<= 129 floats)
constexpr int SH_PENDFLAG = SH_BTAB + 132;                     // per slot: 1 = holds a pending ray
constexpr int kSharedWords = SH_PENDFLAG + 16;
constexpr int kMaxSlots = 16;
// A wave without a tile polls every ~0.2 us; the longest legitimate wait is one serial stage or one context switch
// (microseconds). 2^21 polls (~0.3 s) without work means the protocol is broken: the wave raises SH_ERROR, every wave
// leaves, and the launch reports it in counters[7] and in word 0 of the workspace header instead of hanging the GPU.
constexpr unsigned kIdleLimit = 1u << 21;

template <int MODE>
__host__ __device__ inline int tasks_lds_floats(int P, int slots) {
    return lds_mlp_floats<MODE>() + 144 + P * kLdsPartStride + P * kLdsCanonStride + slots * kSlotWords + kSharedWords;
}

// lane 0 performs an LDS atomic, every lane gets the result
__device__ __forceinline__ unsigned wave_lds_add(unsigned *p, unsigned v, int lane) {
    unsigned r = 0;
    if (lane == 0) r = atomicAdd(p, v);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)r);
}
__device__ __forceinline__ unsigned wave_lds_cas(unsigned *p, unsigned expect, unsigned desired, int lane) {
    unsigned r = 0;
    if (lane == 0) r = atomicCAS(p, expect, desired);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)r);
}
// everything the waves of the workgroup hand to each other lives in LDS: fences of the LOCAL address space only, so that a
// wave's outstanding global stores (a ray's outputs) are not waited for before it publishes the next piece of work
__device__ __forceinline__ void lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }

// The ray queues of enarf_march.h, popped by ANY wave of the workgroup at any time (the round-1 RayQueue had one popper
// at a time): the cursor is one packed word that only ever moves past lists somebody has seen exhausted.
struct TaskQueue {
    unsigned int *heads;
    const uint32_t *lists;
    const RayRec *recs;
    unsigned *sh;                 // the workgroup-shared words
    long long band;
    int home;
    __device__ __forceinline__ void init(void *workspace, int epoch, int B, int n, unsigned *shared, int tid) {
        unsigned int *wsh = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(workspace) + ws_header_off(epoch));
        heads = wsh + kWsHeadsOff;
        recs = reinterpret_cast<const RayRec *>(reinterpret_cast<const char *>(workspace) + ws_records_off());
        lists = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(workspace) + ws_list_off((long long)B * n));
        sh = shared;
        band = ws_band_size(B, n);
        home = xcc_id() & (kQueues - 1);
        if (tid < kQueues * kClasses) sh[SH_QCOUNTS + tid] = wsh[kWsCountsOff + tid];
    }
    // whole wave (uniform): next ray id and its record, or -1 when every list is drained
    __device__ __forceinline__ int pop(RayRec &rec, int lane) {
        int rid = -1;
        if (lane == 0) {
            unsigned cur = sh[SH_QCUR];
            int cls = (int)(cur >> 4), tries = (int)(cur & 15u);
            int lid = -1, idx = 0;
            while (cls < kClasses) {
                while (tries < kQueues) {
                    const int q = (home + tries) & (kQueues - 1);
                    const int l = q * kClasses + cls;
                    const unsigned len = sh[SH_QCOUNTS + l];
                    if (len != 0) {
                        const unsigned j = atomicAdd(heads + l * kWsHeadStride, 1u);
                        if (j < len) { lid = l; idx = (int)j; break; }
                    }
                    tries += 1;
                }
                if (lid >= 0) break;
                cls += 1;
                tries = 0;
            }
            atomicMax(&sh[SH_QCUR], (unsigned)(cls * 16 + tries));      // never moves back
            if (lid >= 0) rid = (int)lists[(size_t)lid * (size_t)band + (size_t)idx];
        }
        rid = __builtin_amdgcn_readfirstlane(rid);
        if (rid >= 0) rec = recs[rid];          // every lane loads the same 32 bytes (one broadcast line)
        return rid;
    }
};

}  // namespace enarf

namespace enarf {

// The launch arguments as the out-of-line stages see them: a pointer into the kernarg segment (constant address space, so
// every field is a scalar load). Passing the kernel's by-value struct by reference would make the compiler copy it to
// scratch per lane; the struct is the kernel's first parameter, i.e. it sits at offset 0.
typedef const __attribute__((address_space(4))) enarf_render_args *RenderArgsK;
__device__ __forceinline__ RenderArgsK kernel_render_args() {
    return (RenderArgsK)__builtin_amdgcn_kernarg_segment_ptr();
}

// everything a task needs that is the same for the whole launch / workgroup
struct MarchCtx {
    unsigned *slots;          // LDS: slot 0
    unsigned *sh;             // LDS: shared scheduler words
    const float *btab;        // LDS: Nc + 1 coarse bin edges
    int nslots, nct, nft;     // slots, coarse / fine tiles per ray
};

// wave counters (wave-uniform)
struct MarchCounters { unsigned pairs, tiles, rays, rounds, skipped; };

template <int SPL>
__device__ __forceinline__ void draw_sorted_uniforms(RenderArgsK a, uint32_t rid, int Nf, int lane, float usort[SPL]) {
    // u_(i) = (E_1 + .. + E_i) / (E_1 + .. + E_{Nf+1}): sorted uniforms from exponential spacings (Philox4x32-10)
    float esum[SPL];
    uint32_t r1_first = 0;
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        uint32_t rnd[4];
        philox4x32((uint32_t)(a->ray_id_base + rid), (uint32_t)((a->ray_id_base + rid) >> 32), (uint32_t)(64 * s + lane), 0x454E4152u, (uint32_t)a->seed, (uint32_t)(a->seed >> 32), rnd);
        esum[s] = (64 * s + lane < Nf) ? -__logf(1.0f - u32_to_unit(rnd[0])) : 0.0f;
        if (s == 0) r1_first = rnd[1];
    }
    wv_scan_incl<SPL>(esum, lane);
    const float etot = __shfl(esum[SPL - 1], 63) - __logf(1.0f - u32_to_unit((uint32_t)__shfl((int)r1_first, 0)));
#pragma unroll
    for (int s = 0; s < SPL; ++s) usort[s] = fminf(esum[s] / etot, 0.99999994f);
}

// Nf sorted importance samples of ray `rid` from the (inclusive, unnormalised) cdf of its Nc smoothed coarse weights
template <int SPL>
__device__ __forceinline__ void bins_from_cdf(RenderArgsK a, uint32_t rid, int Nc, int Nf, const float cdf[SPL], int lane, float bin[SPL]) {
    float usort[SPL];
    draw_sorted_uniforms<SPL>(a, rid, Nf, lane, usort);
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

// S2 of one ray (ONE wave, element e = 64 s + lane): coarse weights (rendering.py:180-184), smoothing (:187-190), the
// importance samples (:192-197) and the early-termination flags of the fine tiles; everything from / to the slot
template <int SPL>
__device__ __noinline__ void ray_sample_stage(RenderArgsK a, const float *l_btab, unsigned *sw, int mult_w, int lane) {
    const int P = a->P, Nc = a->Nc, Nf = a->Nf, n = a->n;
    const RayRec rec = *reinterpret_cast<const RayRec *>(sw + SL_REC);
    const uint32_t rid = sw[SL_RID];
    const int b = (int)(rid / (uint32_t)n), ray = (int)(rid - (uint32_t)b * (uint32_t)n);
    const float dmin = rec.dmin, dmax = rec.dmax;
    const float *l_ch = reinterpret_cast<const float *>(sw + SL_CH), *l_cwmax = reinterpret_cast<const float *>(sw + SL_CWMAX);
    const uint32_t *l_cbits = sw + SL_CBITS;
    float *l_bins = reinterpret_cast<float *>(sw + SL_BINS);
    int *l_skip = reinterpret_cast<int *>(sw + SL_SKIP);
    float bin[SPL];
    float dd[SPL], cs[SPL], T[SPL], wgt[SPL], ws[SPL], wl[SPL], wr[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        const int e = 64 * s + lane;
        const bool active = e < Nc;
        const int ci = min(e, Nc - 1);
        const float den = active ? density_head(l_ch[ci], l_cbits[ci], l_cwmax[ci], mult_w, P) : 0.0f;
        if (a->dbg_coarse_density && active) a->dbg_coarse_density[((size_t)b * n + ray) * Nc + e] = den;
        const float b0 = l_btab[ci], b1 = l_btab[ci + 1];
        const float delta = exact_lerp(dmin, dmax, b1) - exact_lerp(dmin, dmax, b0);
        dd[s] = active ? den * delta * a->render_scale : 0.0f;
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
    if (ENARF_DIAG_ABLATE & 8) {
#pragma unroll
        for (int s = 0; s < SPL; ++s) bin[s] = (float)(64 * s + lane) / (float)Nf;
    } else if (a->bins) {
#pragma unroll
        for (int s = 0; s < SPL; ++s) bin[s] = a->bins[((size_t)b * n + ray) * Nf + min(64 * s + lane, Nf - 1)];
    } else {
        // Importance samples = Nf iid draws from the piecewise-constant pdf, sorted (rendering.py:192-197): sorted uniforms
        // pushed through the inverse CDF (monotone, so the bins come out sorted) - bin index by binary search, position
        // inside the bin by the leftover: the same law as multinomial + U / Nc + sort.
        float cdf[SPL];
#pragma unroll
        for (int s = 0; s < SPL; ++s) cdf[s] = ws[s];
        wv_scan_incl<SPL>(cdf, lane);
        bins_from_cdf<SPL>(a, rid, Nc, Nf, cdf, lane, bin);
    }
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        l_bins[64 * s + lane] = bin[s];
        if (a->dbg_bins && 64 * s + lane < Nf) a->dbg_bins[((size_t)b * n + ray) * Nf + 64 * s + lane] = bin[s];
    }
    // early ray termination (opt-in, early_stop_eps > 0): transmittance in front of the first fine sample of each tile,
    // read off the coarse pass. Below eps every sample of the tile weighs < eps: the tile is skipped (densities 0).
#pragma unroll
    for (int t = 0; t < 4 * SPL; ++t) {
        bool skip = false;
        if (a->early_stop_eps > 0.0f) {
            const float b_first = wv_get<SPL>(bin, min(16 * t, Nf - 1));
            const int jbin = min(max((int)(b_first * (float)Nc), 0), Nc - 1);
            skip = wv_get<SPL>(T, jbin) < a->early_stop_eps;
        }
        if (lane == 0) l_skip[t] = skip ? 1 : 0;
    }
}

// S4 of one ray (ONE wave, element e = 64 s + lane): compositing (rendering.py:307-335) and the ray's outputs
template <int SPL>
__device__ __noinline__ void ray_composite_stage(RenderArgsK a, unsigned *sw, int mult_w, int lane) {
    const int P = a->P, Nf = a->Nf, n = a->n;
    const RayRec rec = *reinterpret_cast<const RayRec *>(sw + SL_REC);
    const uint32_t rid = sw[SL_RID];
    const int b = (int)(rid / (uint32_t)n), ray = (int)(rid - (uint32_t)b * (uint32_t)n);
    const float dmin = rec.dmin, dmax = rec.dmax;
    const float *l_fh = reinterpret_cast<const float *>(sw + SL_FH), *l_fwmax = reinterpret_cast<const float *>(sw + SL_FWMAX);
    const float *l_bins = reinterpret_cast<const float *>(sw + SL_BINS);
    const uint32_t *l_fbits = sw + SL_FBITS;
    const bool dbgq = (a->dbg_fine_density != nullptr);
    float fdepth[SPL], dnext[SPL], den[SPL], cr[SPL], cg[SPL], cb[SPL], dd[SPL], cs[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        const int e = 64 * s + lane;
        const int ci = min(e, Nf - 1);
        const bool have = e < (dbgq ? Nf : Nf - 1);
        const uint32_t bits = l_fbits[ci];
        den[s] = have ? density_head(l_fh[3 * kMaxSamples + ci], bits, l_fwmax[ci], mult_w, P) : 0.0f;
        cr[s] = tanhf(l_fh[ci]); cg[s] = tanhf(l_fh[kMaxSamples + ci]); cb[s] = tanhf(l_fh[2 * kMaxSamples + ci]);
        fdepth[s] = exact_lerp(dmin, dmax, l_bins[64 * s + lane]);
        if (dbgq && e < Nf) {
            const size_t o = ((size_t)b * n + ray) * Nf + e;
            a->dbg_fine_density[o] = den[s];
            if (a->dbg_fine_valid) a->dbg_fine_valid[o] = bits;
            if (a->dbg_fine_color) {
                a->dbg_fine_color[(((size_t)b * 3 + 0) * n + ray) * Nf + e] = cr[s];
                a->dbg_fine_color[(((size_t)b * 3 + 1) * n + ray) * Nf + e] = cg[s];
                a->dbg_fine_color[(((size_t)b * 3 + 2) * n + ray) * Nf + e] = cb[s];
            }
        }
    }
    wv_next<SPL>(fdepth, dnext, lane);
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        dd[s] = (64 * s + lane < Nf - 1) ? den[s] * (dnext[s] - fdepth[s]) * a->render_scale : 0.0f;
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
    const float o_r = wv_sum<SPL>(vr), o_g = wv_sum<SPL>(vg), o_b = wv_sum<SPL>(vb);
    const float o_m = wv_sum<SPL>(wgt), o_d = wv_sum<SPL>(vd);
    if (lane == 0) {
        a->color[((size_t)b * 3 + 0) * n + ray] = o_r;
        a->color[((size_t)b * 3 + 1) * n + ray] = o_g;
        a->color[((size_t)b * 3 + 2) * n + ray] = o_b;
        a->mask[(size_t)b * n + ray] = o_m;
        a->disparity[(size_t)b * n + ray] = o_d;
    }
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        const int e = 64 * s + lane;
        if (a->fine_weights && e < Nf - 1) a->fine_weights[((size_t)b * n + ray) * (Nf - 1) + e] = wgt[s];
        if (a->fine_depth && e < Nf) a->fine_depth[((size_t)b * n + ray) * Nf + e] = fdepth[s];
    }
}

// one 16-sample query tile of a ray: coarse tile t (bin mid-points, rendering.py:119-131) or fine tile t (the ray's
// importance samples); results go to the slot. ONE call site of query_tile for both passes.
template <int MODE>
__device__ __forceinline__ void ray_tile_task(const enarf_render_args &a, const MarchCtx &M, QueryCtx &S, unsigned *sw, bool fine,
                                              int t, int lane, MarchCounters &C) {
    const int Nc = a.Nc, Nf = a.Nf, n = a.n;
    const RayRec rec = *reinterpret_cast<const RayRec *>(sw + SL_REC);
    const uint32_t rid = (uint32_t)__builtin_amdgcn_readfirstlane((int)sw[SL_RID]);      // wave-uniform: plane bases in SGPRs
    const int b = (int)(rid / (uint32_t)n);
    S.feat = a.feat_cl + (size_t)b * a.feat_batch_stride;
    S.mask = a.mask_planes + (size_t)b * a.mask_batch_stride;
#if ENARF_DIAG_TAPCHECK
    S.diag = a.counters; S.diag_rid = rid;
#endif
    const int ncand = (int)sw[SL_NCAND];
    const int *l_cand = reinterpret_cast<const int *>(sw + SL_CAND);
    const float dmin = rec.dmin, dmax = rec.dmax;
    const float sx = exact_mul(dmin, rec.dx), sy = exact_mul(dmin, rec.dy), sz = exact_mul(dmin, rec.dz);
    const float ex = exact_mul(dmax, rec.dx), ey = exact_mul(dmax, rec.dy), ez = exact_mul(dmax, rec.dz);
    const int j4 = lane >> 2, base = 16 * t, i = base + j4;
    const int N = fine ? Nf : Nc;                                  // samples of this pass
    const bool dbgq = (a.dbg_fine_density != nullptr);
    bool skip = false, active;
    float px, py, pz;
    if (!fine) {       // wave-uniform
        active = i < Nc;
        const int ci = min(i, Nc - 1);
        const float b0 = M.btab[ci], b1 = M.btab[ci + 1];
        px = exact_mid(exact_lerp(sx, ex, b1), exact_lerp(sx, ex, b0));
        py = exact_mid(exact_lerp(sy, ey, b1), exact_lerp(sy, ey, b0));
        pz = exact_mid(exact_lerp(sz, ez, b1), exact_lerp(sz, ez, b0));
    } else {
        skip = reinterpret_cast<const int *>(sw + SL_SKIP)[t] != 0;
        active = (i < (dbgq ? Nf : Nf - 1)) && !skip;              // the last sample only closes the last interval
        const float bi = reinterpret_cast<const float *>(sw + SL_BINS)[min(i, Nf - 1)];
        px = exact_lerp(sx, ex, bi); py = exact_lerp(sy, ey, bi); pz = exact_lerp(sz, ez, bi);
    }
    const QueryDbg nodbg{nullptr, nullptr, 0, 0};
    f32x4 o;
    bool ran;
    uint32_t bits;
    float wmax;
    query_tile<MODE, false>(S, l_cand, ncand, px, py, pz, active, lane, o, ran, bits, wmax, nodbg, C.pairs, C.tiles, &C.rounds);
    // MFMA layout: lanes < 16 hold the heads of samples base .. base + 15; gather layout: lane 4 j holds sample j's bits
    float *heads = reinterpret_cast<float *>(sw + (fine ? SL_FH : SL_CH));
    if (lane < 16 && base + lane < N) {
        const int io = base + lane;
        if (fine) { heads[io] = o[0]; heads[kMaxSamples + io] = o[1]; heads[2 * kMaxSamples + io] = o[2]; heads[3 * kMaxSamples + io] = o[3]; }
        else heads[io] = o[3];
    }
    if (i < N && (lane & 3) == 0) {
        sw[(fine ? SL_FBITS : SL_CBITS) + i] = active ? bits : 0u;
        reinterpret_cast<float *>(sw + (fine ? SL_FWMAX : SL_CWMAX))[i] = wmax;
    }
    if (skip) C.skipped += 1;
}

// ---- rays that miss every cube (batches only; a single image drops them in the set-up pass) --------------------------
// Such a ray has no candidate part: every sample is invalid, all densities are zero, and what the reference still
// produces for it are zero colour / mask / disparity / weights and the Nf importance-sampled depths on [near, far]
// (rendering.py:172-224 with zero weights). No query, no tile, nothing to wait for: ONE wave runs the two serial stages on
// a private scratch slot, so the waves of a workgroup clear their band's list of missed rays independently before the
// march proper starts (as ordinary rays they cost 22 % of an 8-frame launch: three barriers each for nothing).
// `scratch`: kSlotWords words of LDS private to the calling wave. Returns the rays done (wave-uniform).
// A ray without a valid sample, in production runs (no debug taps): its coarse weights are all +0, so the smoothed weights
// are 0.01 in every bin ((0 + 0) / 2 + 0.01 exactly as ray_sample_stage forms them) and their cdf is the same for every such
// ray - scanned once per wave by the caller; what ray_composite_stage would write is T = 1, every weight 1 - e^-0 = +0,
// every sum +0. Same bits as the two general stages, without their scans.
template <int SPL>
__device__ __forceinline__ void missed_ray_short_cut(RenderArgsK a, uint32_t rid, const RayRec &rec, const float cdf[SPL], int lane) {
    const int Nc = a->Nc, Nf = a->Nf, n = a->n;
    const int b = (int)(rid / (uint32_t)n), ray = (int)(rid - (uint32_t)b * (uint32_t)n);
    float bin[SPL];
    if (a->bins) {
#pragma unroll
        for (int s = 0; s < SPL; ++s) bin[s] = a->bins[((size_t)b * n + ray) * Nf + min(64 * s + lane, Nf - 1)];
    } else {
        bins_from_cdf<SPL>(a, rid, Nc, Nf, cdf, lane, bin);
    }
    if (lane < 3) a->color[((size_t)b * 3 + lane) * n + ray] = 0.0f;
    if (lane == 3) a->mask[(size_t)b * n + ray] = 0.0f;
    if (lane == 4) a->disparity[(size_t)b * n + ray] = 0.0f;
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        const int e = 64 * s + lane;
        if (a->dbg_bins && e < Nf) a->dbg_bins[((size_t)b * n + ray) * Nf + e] = bin[s];
        if (a->fine_weights && e < Nf - 1) a->fine_weights[((size_t)b * n + ray) * (Nf - 1) + e] = 0.0f;
        if (a->fine_depth && e < Nf) a->fine_depth[((size_t)b * n + ray) * Nf + e] = exact_lerp(rec.dmin, rec.dmax, bin[s]);
    }
}

#ifndef ENARF_MISSED_CHUNK
#define ENARF_MISSED_CHUNK 32
#endif
constexpr int kMissedChunk = ENARF_MISSED_CHUNK;
template <int SPL>
__device__ __forceinline__ unsigned march_missed_rays(RenderArgsK ak, const float *l_btab, unsigned *scratch, int mult_w, int lane) {
    const int B = ak->B, n = ak->n, Nc = ak->Nc;
    const char *ws = reinterpret_cast<const char *>(ak->workspace);
    unsigned int *wsh = reinterpret_cast<unsigned int *>(const_cast<char *>(ws) + ws_header_off(ak->ws_epoch));
    const uint32_t *lists = reinterpret_cast<const uint32_t *>(ws + ws_list_off((long long)B * n));
    const RayRec *recs = reinterpret_cast<const RayRec *>(ws + ws_records_off());
    if (wsh[2] == 0u) return 0u;                 // no such ray in this launch (always so for a single image): one load
    const long long band = ws_band_size(B, n);
    const int home = xcc_id() & (kQueues - 1);
    // debug runs (taps wanted) and the diagnostic fixed-grid build take the two general stages on a zeroed scratch slot
    const bool general = ak->dbg_fine_density || ak->dbg_coarse_density || (ENARF_DIAG_ABLATE & 8);
    float cdf[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) cdf[s] = (64 * s + lane < Nc) ? (0.0f + 0.0f) / 2.0f + 0.01f : 0.0f;
    wv_scan_incl<SPL>(cdf, lane);
    unsigned done = 0;
    bool zeroed = false;
    for (int t = 0; t < kQueues; ++t) {
        const int l = ws_missed_list((home + t) & (kQueues - 1));
        const unsigned len = wsh[kWsCountsOff + l];                 // final: written by the set-up pass of an earlier launch
        if (len == 0u) continue;
        while (true) {
            // a chunk of the list per atomic: tens of thousands of equally cheap rays behind eight queue heads would
            // otherwise spend their time queueing at those eight cache lines
            unsigned j0 = 0;
            if (lane == 0) j0 = atomicAdd(wsh + kWsHeadsOff + l * kWsHeadStride, (unsigned)kMissedChunk);
            j0 = (unsigned)__builtin_amdgcn_readfirstlane((int)j0);
            if (j0 >= len) break;
            const unsigned j1 = min(j0 + (unsigned)kMissedChunk, len);
            for (unsigned j = j0; j < j1; ++j) {
                const uint32_t rid = (uint32_t)__builtin_amdgcn_readfirstlane((int)lists[(size_t)l * (size_t)band + j]);
                if (general) {
                    if (!zeroed) {   // heads and validity bits of a ray without a valid sample: all zero, for every ray alike
                        for (int i = lane; i < kSlotWords; i += 64) scratch[i] = 0u;
                        zeroed = true;
                    }
                    if (lane == 0) {
                        scratch[SL_RID] = rid;
                        *reinterpret_cast<RayRec *>(scratch + SL_REC) = recs[rid];
                    }
                    ray_sample_stage<SPL>(ak, l_btab, scratch, mult_w, lane);
                    ray_composite_stage<SPL>(ak, scratch, mult_w, lane);
                } else {
                    missed_ray_short_cut<SPL>(ak, rid, recs[rid], cdf, lane);
                }
                done += 1;
            }
        }
    }
    return done;
}

// ---- the slot chain: pop -> publish / park -> ... -> composite -> pop -------------------------------------------------------
struct ImageCtx {                 // what changes with the image: staged by ONE wave while no tile is running
    const enarf_render_args *a;
    float *lds;                   // start of the dynamic LDS (MLP section first)
};

// (re)stage image b's MLP pack and part frames; lanes of ONE wave
template <int MODE>
__device__ __forceinline__ void stage_image(RenderArgsK a, float *lds, int b, int lane) {
    QueryCtx tmp;
    float *unused;
    stage_common<MODE>(lds, tmp, unused, reinterpret_cast<const char *>(a->mlp_pack) + (size_t)b * kPackBytes,
                       a->parts + (size_t)b * a->P * kPartStride, a->canonical_pose, a->P, lane, 64);
}

// fill slot s with a popped ray and publish its coarse tiles (the caller holds the slot's in-flight token)
__device__ __forceinline__ void publish_ray(const MarchCtx &M, unsigned *sw, int lane) {
    const RayRec rec = *reinterpret_cast<const RayRec *>(sw + SL_REC);
    const int ncand = build_cand_list(reinterpret_cast<int *>(sw + SL_CAND), rec.cand, lane);
    if (lane == 0) { sw[SL_NCAND] = (unsigned)ncand; sw[SL_DONE] = 0u; }
    lds_release();
    if (lane == 0) {
        const unsigned gen = (ctl_gen(sw[SL_CTL]) + 1u) & 0x3FFFu;
        __hip_atomic_store(&sw[SL_CTL], pack_ctl(gen, ST_COARSE, (unsigned)M.nct, 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// The wave that brought the in-flight count to zero: nothing is published, nobody is popping. If rays of another image are
// parked, stage that image and publish them; if every chain has ended, there is nothing to do (the waves will leave).
template <int MODE>
__device__ __forceinline__ void drain_point(RenderArgsK a, const MarchCtx &M, float *lds, int lane) {
    volatile unsigned *vsh = M.sh;
    // Looped: while this wave owns SH_SWITCHING another slot's ray may retire and its refill park a ray of yet another
    // image; that wave's compare-and-swap below fails and it leaves, counting on the owner. So the owner looks again after
    // giving the flag back (the other wave's token release precedes its failed CAS, hence this re-check sees count zero).
    while (true) {
        if (vsh[SH_PENDING] == 0u) return;
        if (wave_lds_cas(&M.sh[SH_SWITCHING], 0u, 1u, lane) != 0u) return;       // somebody else is at it
        lds_acquire();
        // the parked ray with the smallest id decides the image (the lists are in image order)
        unsigned best = 0xFFFFFFFFu;
        for (int s = 0; s < M.nslots; ++s)
            if (vsh[SH_PENDFLAG + s]) best = min(best, (unsigned)M.slots[s * kSlotWords + SL_RID]);
        if (best != 0xFFFFFFFFu) {
            const int b = (int)(best / (uint32_t)a->n);
            if ((unsigned)b != vsh[SH_CTXB]) {
                stage_image<MODE>(a, lds, b, lane);
                if (lane == 0) M.sh[SH_CTXB] = (unsigned)b;
            }
            lds_release();
            for (int s = 0; s < M.nslots; ++s) {
                unsigned *sw = M.slots + s * kSlotWords;
                if (vsh[SH_PENDFLAG + s] && (int)(sw[SL_RID] / (uint32_t)a->n) == b) {
                    if (lane == 0) { M.sh[SH_PENDFLAG + s] = 0u; atomicAdd(&M.sh[SH_PENDING], 0xFFFFFFFFu); atomicAdd(&M.sh[SH_INFLIGHT], 1u); }
                    publish_ray(M, sw, lane);
                }
            }
        }
        lds_release();
        if (lane == 0) __hip_atomic_store(&M.sh[SH_SWITCHING], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_release();
        lds_acquire();
        if (!(vsh[SH_INFLIGHT] == 0u && vsh[SH_PENDING] != 0u)) return;
    }
}

// give back a slot's in-flight token; the wave that takes the count to zero handles the drain point
template <int MODE>
__device__ __forceinline__ void release_token(RenderArgsK a, const MarchCtx &M, float *lds, int lane) {
    lds_release();
    const unsigned before = wave_lds_add(&M.sh[SH_INFLIGHT], 0xFFFFFFFFu, lane);
    if (before == 1u) drain_point<MODE>(a, M, lds, lane);
}

// next ray for slot s (the caller holds the slot's token): publish it, park it (other image) or end the chain
template <int MODE>
__device__ __noinline__ unsigned refill_slot(RenderArgsK a, const MarchCtx M, TaskQueue tq, float *lds, int s, int lane) {
    unsigned popped = 0u;                            // rays this call took off the queues (for the launch's counters)
    unsigned *sw = M.slots + s * kSlotWords;
    RayRec rec;
    int rid;
    // the ray popped ahead for this slot, if any (states: see SL_NEXT_STATE; only the owner of state 3 writes the record, and
    // only this function takes it away, so nothing is lost whichever of the two comes first)
    volatile unsigned *vst = sw + SL_NEXT_STATE;
    unsigned ahead = *vst;
    // its pop is in flight: that IS our pop. (The popper always finishes - a queue atomic and two loads; the watchdog's
    // SH_ERROR still releases this spin like every other wait of the kernel.)
    while (ahead == 3u && reinterpret_cast<volatile unsigned *>(M.sh)[SH_ERROR] == 0u) { __builtin_amdgcn_s_sleep(2); ahead = *vst; }
    if (ahead == 3u) ahead = 2u;          // the launch is being abandoned: end this chain
    if (ahead == 1u) {
        lds_acquire();
        rid = (int)sw[SL_NEXT_RID];
        rec = *reinterpret_cast<const RayRec *>(sw + SL_NEXT_REC);
        lds_release();
        if (lane == 0) __hip_atomic_store(sw + SL_NEXT_STATE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (ahead == 2u) {
        rid = -1;
    } else {
        rid = tq.pop(rec, lane);
        if (rid >= 0) popped = 1u;
    }
    if (rid < 0) {                                   // queues drained: this slot's chain ends
        if (lane == 0) atomicAdd(&M.sh[SH_DEAD], 1u);
        release_token<MODE>(a, M, lds, lane);
        return popped;
    }
    if (lane == 0) { sw[SL_RID] = (unsigned)rid; *reinterpret_cast<RayRec *>(sw + SL_REC) = rec; }
    const unsigned b = (unsigned)rid / (uint32_t)a->n;
    volatile unsigned *vsh = M.sh;
    if (b == vsh[SH_CTXB]) {
        publish_ray(M, sw, lane);
    } else {                                         // a ray of another image: park it until the current image drains
        if (lane == 0) { M.sh[SH_PENDFLAG + s] = 1u; atomicAdd(&M.sh[SH_PENDING], 1u); }
        release_token<MODE>(a, M, lds, lane);
    }
    return popped;
}

// pop the slot's NEXT ray while its current one is in the fine pass (called by the wave that ran S2, after it has published
// the fine tiles): the queue's dependent chain - atomic, list entry, 32-byte record - then costs the slot nothing
__device__ __noinline__ unsigned prefetch_next_ray(TaskQueue tq, unsigned *sw, int lane) {
    if (wave_lds_cas(sw + SL_NEXT_STATE, 0u, 3u, lane) != 0u) return 0u;    // one is held or in flight already
    RayRec rec;
    const int rid = tq.pop(rec, lane);
    if (lane == 0 && rid >= 0) { sw[SL_NEXT_RID] = (unsigned)rid; *reinterpret_cast<RayRec *>(sw + SL_NEXT_REC) = rec; }
    lds_release();
    if (lane == 0) __hip_atomic_store(sw + SL_NEXT_STATE, rid >= 0 ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return rid >= 0 ? 1u : 0u;
}

// =================================================================================================================================
// the march: NW waves per workgroup, one workgroup per CU, `nslots` rays in flight
// =================================================================================================================================
template <int MODE, int SPL, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void march_kernel(const enarf_render_args a, int nslots, unsigned int *status) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, Nc = a.Nc, Nf = a.Nf;
    QueryCtx S;
    float *after;
    // pointers only (the image context is staged at the first drain point): [mlp][bias][parts][canon][slots][shared]
    {
        float *l_mlp = lds, *l_bias = l_mlp + lds_mlp_floats<MODE>(), *l_parts = l_bias + 144;
        float *l_canon = l_parts + P * kLdsPartStride;
        after = l_canon + P * kLdsCanonStride;
        S.mlp = l_mlp; S.mlp_h = reinterpret_cast<const short *>(l_mlp); S.bias = l_bias; S.parts = l_parts; S.canon = l_canon;
    }
    S.feat = a.feat_cl; S.mask = a.mask_planes;
    S.H = a.H; S.W = a.W; S.P = P; S.mult_w = a.multiply_density_with_weight ? (a.uniform_part_weight ? 2 : 1) : 0;
    S.clamp_mask = a.clamp_mask; S.uniform_w = a.uniform_part_weight ? 1.0f / (float)P : 0.0f;
#if ENARF_DIAG_TAPCHECK
    S.diag = nullptr; S.diag_rid = 0;
#endif
    MarchCtx M;
    M.slots = reinterpret_cast<unsigned *>(after);
    M.sh = M.slots + nslots * kSlotWords;
    M.btab = reinterpret_cast<const float *>(M.sh + SH_BTAB);
    M.nslots = nslots; M.nct = (Nc + 15) / 16; M.nft = (Nf + 15) / 16;
    TaskQueue tq;
    tq.init(a.workspace, a.ws_epoch, a.B, a.n, M.sh, tid);
    if (tid == 0) {
        M.sh[SH_QCUR] = 0u; M.sh[SH_DEAD] = 0u; M.sh[SH_PENDING] = 0u; M.sh[SH_CTXB] = 0xFFFFFFFFu; M.sh[SH_SWITCHING] = 0u;
        M.sh[SH_ERROR] = 0u;
        M.sh[SH_INFLIGHT] = (unsigned)nslots;            // every slot starts with a token: its first pop is under way
    }
    if (tid <= Nc) reinterpret_cast<float *>(M.sh + SH_BTAB)[tid] = linspace_sym(0.0f, 1.0f, Nc + 1, tid);
    for (int s = tid; s < nslots; s += NW * 64) {
        M.slots[s * kSlotWords + SL_CTL] = pack_ctl(0u, ST_NONE, 0u, 0u);
        M.slots[s * kSlotWords + SL_DONE] = 0u;
        M.slots[s * kSlotWords + SL_NEXT_STATE] = 0u;
        M.sh[SH_PENDFLAG + s] = 0u;
    }
    __syncthreads();

    MarchCounters C{0u, 0u, 0u, 0u, 0u};
    const RenderArgsK ak = kernel_render_args();
    // rays without a candidate part first (batches only): as many waves as private scratch slots fit into the not yet
    // staged MLP section of the LDS
    if (wave < lds_mlp_floats<MODE>() / kSlotWords)
        C.rays += march_missed_rays<SPL>(ak, M.btab, reinterpret_cast<unsigned *>(lds) + wave * kSlotWords, S.mult_w, lane);
    __syncthreads();
    // the first ray decides which image's context the WHOLE workgroup stages (one wave alone takes ~40 us for the 29 KB)
    if (wave == 0) {
        RayRec rec;
        const int rid = tq.pop(rec, lane);
        if (lane == 0) {
            M.sh[SH_FIRST] = (unsigned)rid;
            if (rid >= 0) { M.slots[SL_RID] = (unsigned)rid; *reinterpret_cast<RayRec *>(M.slots + SL_REC) = rec; }
        }
        if (rid >= 0) C.rays += 1;
    }
    __syncthreads();
    const int first = (int)M.sh[SH_FIRST];
    if (first < 0) {                               // uniform: every queue was drained before this workgroup got a ray
        if (a.counters && lane == 0 && C.rays) atomicAdd(&a.counters[2], (unsigned long long)C.rays);     // its missed rays
        return;
    }
    {
        const int b0 = (int)((unsigned)first / (uint32_t)a.n);
        QueryCtx tmp;
        float *unused;
        stage_common<MODE>(lds, tmp, unused, reinterpret_cast<const char *>(a.mlp_pack) + (size_t)b0 * kPackBytes,
                           a.parts + (size_t)b0 * P * kPartStride, a.canonical_pose, P, tid, NW * 64);
        if (tid == 0) M.sh[SH_CTXB] = (unsigned)b0;
    }
    __syncthreads();
    if (wave == 0) publish_ray(M, M.slots, lane);   // slot 0 holds the first ray (and one of the initial tokens)
#if ENARF_TIMERS == 5   // diagnostic build: per-wave cycles in 0 tiles, 1 S2, 2 S4, 3 refill after S4 (+ start-up), 4 idle, 5 scan + claim, 6 pop-ahead
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
#define TK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm[k] += now_ - t_last; t_last = now_; } while (0)
#else
#define TK(k) do { } while (0)
#endif
    for (int s = wave; s < nslots; s += NW)
        if (s != 0) C.rays += refill_slot<MODE>(ak, M, tq, lds, s, lane);
    TK(3);

    volatile unsigned *vsl = M.slots;
    volatile unsigned *vsh = M.sh;
    unsigned idle = 0u;
    for (;;) {
        // ---- find a tile: fine tiles first (they retire rays and free slots), any slot; start at a wave-specific slot
        const unsigned c = (lane < nslots) ? vsl[lane * kSlotWords + SL_CTL] : 0u;
        const bool can = ctl_next(c) < ctl_tiles(c);
        const uint64_t bf = __ballot(can && ctl_stage(c) == ST_FINE), bc = __ballot(can && ctl_stage(c) == ST_COARSE);
        const uint64_t set = bf ? bf : bc;
        if (set) {
            const int start = wave % nslots;
            const uint64_t hi = (set >> start) << start;
            const int s = hi ? __builtin_ctzll(hi) : __builtin_ctzll(set);
            const unsigned old = (unsigned)__builtin_amdgcn_readlane((int)c, s);
            unsigned *sw = M.slots + s * kSlotWords;
            if (wave_lds_cas(&sw[SL_CTL], old, old + 1u, lane) != old) { TK(5); continue; }      // taken or recycled meanwhile: look again
            lds_acquire();
            const bool fine = ctl_stage(old) == ST_FINE;
            idle = 0u;
            TK(5);
            ray_tile_task<MODE>(a, M, S, sw, fine, (int)ctl_next(old), lane, C);
            lds_release();
            const unsigned d = wave_lds_add(&sw[SL_DONE], 1u, lane);
            TK(0);
            if (d + 1u == ctl_tiles(old)) {          // this wave completed the stage: the serial follow-up is its job
                lds_acquire();
                if (!fine) {
                    __builtin_amdgcn_s_setprio(3);   // fine tiles of this ray cannot start before this is done
                    ray_sample_stage<SPL>(ak, M.btab, sw, S.mult_w, lane);
                    __builtin_amdgcn_s_setprio(0);
                    if (lane == 0) sw[SL_DONE] = 0u;
                    lds_release();
                    if (lane == 0)
                        __hip_atomic_store(&sw[SL_CTL], pack_ctl(ctl_gen(old), ST_FINE, (unsigned)M.nft, 0u), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                    TK(1);
                    __builtin_amdgcn_s_setprio(1);
                    C.rays += prefetch_next_ray(tq, sw, lane);
                    __builtin_amdgcn_s_setprio(0);
                    TK(6);
                } else {
                    __builtin_amdgcn_s_setprio(3);   // the slot is empty until this chain has published its next ray
                    ray_composite_stage<SPL>(ak, sw, S.mult_w, lane);
                    TK(2);
                    C.rays += refill_slot<MODE>(ak, M, tq, lds, s, lane);       // the slot keeps its token across the pop
                    __builtin_amdgcn_s_setprio(0);
                    TK(3);
                }
            }
            continue;
        }
        if (vsh[SH_DEAD] >= (unsigned)nslots || vsh[SH_ERROR] != 0u) break;      // every chain has ended (a parked ray keeps its slot alive)
        if (++idle > kIdleLimit) {
            if (lane == 0) {
                M.sh[SH_ERROR] = 1u;
                // the host's sticky status word (pinned host memory, enarf_device_status): seen by every caller, counters or not
                if (status) __hip_atomic_fetch_or(status, ENARF_STATUS_MARCH_WATCHDOG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (a.counters) atomicAdd(&a.counters[7], 1ull);
                reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(a.workspace) + ws_header_off(a.ws_epoch))[0] = 0xDEADu;
            }
            break;
        }
        __builtin_amdgcn_s_sleep(4);
        TK(4);
    }
#if ENARF_TIMERS == 5
    if (a.counters && lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&a.counters[k], tm[k]);
    return;
#endif
    if (a.counters && lane == 0) {
        atomicAdd(&a.counters[0], (unsigned long long)C.pairs);
        atomicAdd(&a.counters[1], (unsigned long long)C.tiles);
        atomicAdd(&a.counters[2], (unsigned long long)C.rays);
        atomicAdd(&a.counters[3], (unsigned long long)C.rounds);
        atomicAdd(&a.counters[4], (unsigned long long)C.skipped);
    }
}

}  // namespace enarf
