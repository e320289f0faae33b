// enarf_device.h - shared device code of libenarf_hip.so (gfx950 / CDNA4 only, wave64).
//
// Conventions
//   * "exact" functions spell every 3x3 product as ((a0*b0 + a1*b1) + a2*b2) with FMA contraction
//     switched off, the op order of oracle/enarf_oracle.py, so cube-validity masks are bit-exact.
//   * a part frame is 16 floats: R row-major [0..9), t (already x coordinate_scale) [9..12),
//     canonical scale [12], pad.
//   * the MLP pack (one per image) holds the demodulated weights in MFMA A-operand order, see below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "enarf_hip.h"

namespace enarf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;
constexpr int kPartStride = 16;
constexpr int kFeat = ENARF_FEAT_DIM;      // 32
constexpr int kHid = ENARF_HIDDEN;         // 64

// ---- MLP pack layout --------------------------------------------------------------------------
// fp32 section (floats). The MLP is evaluated transposed, OUT^T (units x points) = W (units x in) .
// X^T (in x points), on 16-point tiles: lane l = (j = l & 15 : point, g = l >> 4 : k-group).
//   v_mfma_f32_16x16x4_f32: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D[row 4g + r][col j] in reg r.
//   layer 1, k-step s (0..7):   k-group g feeds feature channel 8g + s  (the lane's own gathered channels)
//   layer 2/3, k-step q = 4*ob' + r': k-group g feeds hidden unit 16ob' + 4g + r' (the lane's own acc reg)
// so no data moves between lanes from the gather through the last layer.
constexpr int PK_W1 = 0;                      // [4 ob][8 s][64 lanes]
constexpr int PK_W2 = PK_W1 + 4 * 8 * 64;     // [4 ob][16 q][64 lanes]
constexpr int PK_W3 = PK_W2 + 4 * 16 * 64;    // [16 q][64 lanes], rows 4..15 zero
constexpr int PK_B1 = PK_W3 + 16 * 64;        // [64]
constexpr int PK_B2 = PK_B1 + 64;             // [64]
constexpr int PK_B3 = PK_B2 + 64;             // [16], entries 4..15 zero
constexpr int PK_F32_FLOATS = PK_B3 + 16;     // 7312 floats = 29248 B
// bf16 section (16-bit units), v_mfma_f32_16x16x32_bf16: A[i = l&15][k = 8(l>>4) + jj], jj = 0..7.
//   layer 1: k = feature channel 8g + jj (one k-step)
//   layer 2/3, k-step ks (0,1): slot (g, jj) feeds hidden unit 16(2ks + (jj>>2)) + 4g + (jj&3)
// each A operand stored twice: hi = bf16(w), lo = bf16(w - hi)
constexpr int PKH_W1 = 0;                               // [4 ob][hi,lo][64 lanes][8]
constexpr int PKH_W2 = PKH_W1 + 4 * 2 * 64 * 8;         // [4 ob][2 ks][hi,lo][64][8]
constexpr int PKH_W3 = PKH_W2 + 4 * 2 * 2 * 64 * 8;     // [2 ks][hi,lo][64][8]
constexpr int PKH_SHORTS = PKH_W3 + 2 * 2 * 64 * 8;     // 14336 shorts = 28672 B
// fp16 section: same layout as the bf16 section, hi = f16(w), lo = f16(w - hi)
// transposed fp32 section (floats, after the fp16 section): A operands of the backward products dH = W^T dZ, same
// chaining trick (the k-step (ob', r') of a product reads the lane's own register r' of block ob' of dZ):
//   W3T [4 ob][64 lanes]       : A[i][k] = W3[k][16ob + i]                         (k = the 4 outputs: one K=4 step)
//   W2T [4 ob][16 q][64 lanes] : A[i][g] = W2[16ob' + 4g + r'][16ob + i]
//   W1T [2 ob][16 q][64 lanes] : A[i][g] = W1[16ob' + 4g + r'][8(i>>2) + 4ob + (i&3)]   (rows land as channel 8g + 4ob + r)
constexpr int PKT_W3T = 0;
constexpr int PKT_W2T = PKT_W3T + 4 * 64;
constexpr int PKT_W1T = PKT_W2T + 4 * 16 * 64;
constexpr int PKT_FLOATS = PKT_W1T + 2 * 16 * 64;      // 6400 floats
constexpr size_t kPackTOff = (size_t)PK_F32_FLOATS * 4 + (size_t)PKH_SHORTS * 2 * 2;    // byte offset of the transposed section
constexpr size_t kPackBytes = kPackTOff + (size_t)PKT_FLOATS * 4;   // 112192 B
static_assert(kPackBytes % 16 == 0, "pack must stay 16-byte aligned per image");

// ---- wave primitives ----------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-wide min / max, every lane gets the result: the same DPP ladder as wave_scan_incl with `fill` (the identity of the
// operation) where a source lane is out of range or its row is masked off; lane 63 ends up with the full reduction
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fill_f(float v, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                                                  CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_min(float v) {
    const float inf = __builtin_huge_valf();
    v = fminf(v, dpp_fill_f<0x111, 0xF>(v, inf));
    v = fminf(v, dpp_fill_f<0x112, 0xF>(v, inf));
    v = fminf(v, dpp_fill_f<0x114, 0xF>(v, inf));
    v = fminf(v, dpp_fill_f<0x118, 0xF>(v, inf));
    v = fminf(v, dpp_fill_f<0x142, 0xA>(v, inf));
    v = fminf(v, dpp_fill_f<0x143, 0xC>(v, inf));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    const float ninf = -__builtin_huge_valf();
    v = fmaxf(v, dpp_fill_f<0x111, 0xF>(v, ninf));
    v = fmaxf(v, dpp_fill_f<0x112, 0xF>(v, ninf));
    v = fmaxf(v, dpp_fill_f<0x114, 0xF>(v, ninf));
    v = fmaxf(v, dpp_fill_f<0x118, 0xF>(v, ninf));
    v = fmaxf(v, dpp_fill_f<0x142, 0xA>(v, ninf));
    v = fmaxf(v, dpp_fill_f<0x143, 0xC>(v, ninf));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// DPP data movement (no LDS crossbar round trip as with ds_bpermute): v from the lane `ctrl` names, 0 where the row is
// masked off or the source lane is out of range
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// inclusive prefix sum across the 64 lanes: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8), then the row
// totals hop across (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) - six dependent VALU ops
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
    (void)lane;
    v += dpp_f<0x111, 0xF>(v);
    v += dpp_f<0x112, 0xF>(v);
    v += dpp_f<0x114, 0xF>(v);
    v += dpp_f<0x118, 0xF>(v);
    v += dpp_f<0x142, 0xA>(v);
    v += dpp_f<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_scan_incl(v, 0)), 63));
}

// ---- "wave vectors": SPL values per lane = SPL * 64 elements along a ray, element e = 64 s + lane --------------------
template <int SPL>
__device__ __forceinline__ void wv_scan_incl(float v[SPL], int lane) {
    float carry = 0.0f;
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        v[s] = wave_scan_incl(v[s], lane) + carry;
        carry = __shfl(v[s], 63);
    }
}
template <int SPL>
__device__ __forceinline__ float wv_get(const float v[SPL], int idx) {      // idx may differ per lane
    float r = __shfl(v[0], idx & 63);
    if (SPL == 2) {
        const float r1 = __shfl(v[SPL - 1], idx & 63);
        r = (idx >= 64) ? r1 : r;
    }
    return r;
}
template <int SPL>
__device__ __forceinline__ void wv_prev(const float v[SPL], float out[SPL], int lane) {    // out[e] = v[e-1], out[0] = 0
    float carry = 0.0f;
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        const float up = dpp_f<0x138, 0xF>(v[s]);      // wave_shr:1
        out[s] = (lane == 0) ? carry : up;
        if (s + 1 < SPL) carry = __shfl(v[s], 63);
    }
}
template <int SPL>
__device__ __forceinline__ void wv_next(const float v[SPL], float out[SPL], int lane) {    // out[e] = v[e+1] (last: 0)
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        const float dn = dpp_f<0x130, 0xF>(v[s]);      // wave_shl:1
        const float nxt = (s + 1 < SPL) ? __shfl(v[(s + 1 < SPL) ? s + 1 : s], 0) : 0.0f;
        out[s] = (lane == 63) ? nxt : dn;
    }
}
template <int SPL>
__device__ __forceinline__ float wv_sum(const float v[SPL]) {
    float t = 0.0f;
#pragma unroll
    for (int s = 0; s < SPL; ++s) t += v[s];
    return wave_sum(t);
}

// bijective XCD-aware remap: workgroups that the dispatcher deals round-robin to one XCD get a
// contiguous range of logical tiles, so neighbouring ray tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

// ---- exact-order geometry -------------------------------------------------------------------------
// local = R^T (p - t)            (models/narf.py:158-162; libraries/NeRF/utils.py:24-27)
__device__ __forceinline__ void exact_local(const float *F, float px, float py, float pz,
                                            float &lx, float &ly, float &lz) {
#pragma clang fp contract(off)
    const float d0 = px - F[9], d1 = py - F[10], d2 = pz - F[11];
    lx = (F[0] * d0 + F[3] * d1) + F[6] * d2;
    ly = (F[1] * d0 + F[4] * d1) + F[7] * d2;
    lz = (F[2] * d0 + F[5] * d1) + F[8] * d2;
}
// canonical = Rc (local * s) + tc   (models/narf.py:165-169); C = 12 floats: Rc row-major, tc
__device__ __forceinline__ void exact_canonical(const float *C, float s, float lx, float ly, float lz,
                                                float &cx, float &cy, float &cz) {
#pragma clang fp contract(off)
    const float q0 = lx * s, q1 = ly * s, q2 = lz * s;
    cx = ((C[0] * q0 + C[1] * q1) + C[2] * q2) + C[9];
    cy = ((C[3] * q0 + C[4] * q1) + C[5] * q2) + C[10];
    cz = ((C[6] * q0 + C[7] * q1) + C[8] * q2) + C[11];
}
__device__ __forceinline__ bool in_unit_cube_incl(float x, float y, float z) {   // utils.py:42  (<=)
    return fabsf(x) <= 1.0f && fabsf(y) <= 1.0f && fabsf(z) <= 1.0f;
}
__device__ __forceinline__ bool in_unit_cube_strict(float x, float y, float z) { // narf.py:201  (<)
    return fabsf(x) < 1.0f && fabsf(y) < 1.0f && fabsf(z) < 1.0f;
}
// torch.linspace's symmetric formula (ATen RangeFactories), element i of `steps`
__device__ __forceinline__ float linspace_sym(float start, float end, int steps, int i) {
#pragma clang fp contract(off)
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}
// a*(1-b) + c*b with separately rounded ops (rendering.py:120-126, :198-200)
__device__ __forceinline__ float exact_lerp(float a, float c, float b) {
#pragma clang fp contract(off)
    return a * (1.0f - b) + c * b;
}
__device__ __forceinline__ float exact_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float exact_mid(float e1, float e0) {   // (edge[1:] + edge[:-1]) / 2
#pragma clang fp contract(off)
    return (e1 + e0) / 2.0f;
}
__device__ __forceinline__ float exact_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
#pragma clang fp contract(off)
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

// ---- bilinear sampling (ATen grid_sampler_2d, bilinear / zeros / align_corners=False) --------------
#ifndef ENARF_DIAG_NOMUL24      // A/B only
#define ENARF_DIAG_NOMUL24 0
#endif
// row * width for plane coordinates: both far below 2^23 (validated on the host), so the full-rate 24-bit multiply does
// (v_mul_lo_u32 issues at a quarter of the rate)
__host__ __device__ __forceinline__ int mul_rc(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__) && !ENARF_DIAG_NOMUL24
    return __mul24(a, b);
#else
    return a * b;
#endif
}
struct Taps {
    int o00, o01, o10, o11;        // y*W + x of nw, ne, sw, se (clamped in-bounds)
    float w00, w01, w10, w11;      // weights, zeroed for out-of-bounds taps
    int xe;                        // x edge: 0 = x0, x0 + 1 both in the plane; 1 = x0 left of it (o00 == o01, column 0);
                                   // 2 = x1 right of it (o00 == o01, column W - 1). Only meaningful for x0 in [-1, W - 1].
    int yp;                        // parity of the (clamped) upper texel row: the lower row has the other one (backward's walkers)
};
__host__ __device__ __forceinline__ Taps make_taps(float x, float y, int H, int W) {
    Taps t;
    float ix, iy;
    {
#pragma clang fp contract(off)
        ix = ((x + 1.0f) * (float)W - 1.0f) / 2.0f;
        iy = ((y + 1.0f) * (float)H - 1.0f) / 2.0f;
    }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float ax1 = ix - fx, ax0 = (fx + 1.0f) - ix;
    const float ay1 = iy - fy, ay0 = (fy + 1.0f) - iy;
    const bool bx0 = (x0 >= 0) & (x0 < W), bx1 = (x1 >= 0) & (x1 < W);
    const bool by0 = (y0 >= 0) & (y0 < H), by1 = (y1 >= 0) & (y1 < H);
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    const int r0 = mul_rc(cy0, W), r1 = mul_rc(cy1, W);
    t.o00 = r0 + cx0; t.o01 = r0 + cx1; t.o10 = r1 + cx0; t.o11 = r1 + cx1;
    // out-of-bounds taps weigh zero: zero the axis factor (all factors are >= 0, so the products are the same bits)
    const float zx0 = bx0 ? ax0 : 0.0f, zx1 = bx1 ? ax1 : 0.0f, zy0 = by0 ? ay0 : 0.0f, zy1 = by1 ? ay1 : 0.0f;
    t.w00 = zx0 * zy0;
    t.w01 = zx1 * zy0;
    t.w10 = zx0 * zy1;
    t.w11 = zx1 * zy1;
    t.xe = (x0 < 0) ? 1 : (x1 >= W) ? 2 : 0;
    t.yp = cy0 & 1;
    return t;
}
// The same taps for coordinates of a VALID (part, point) pair, i.e. |x| < 1 and |y| < 1 strictly (narf.py:201), with half
// the range logic. For such a coordinate (x + 1) lies in [2^-24, 2] after rounding (1 - 2^-24 + 1 rounds up to 2), so
//   ix = ((x + 1) W - 1) / 2 lies in [-0.5 + eps, W - 0.5],   x0 = floor(ix) in [-1, W - 1],   x1 = x0 + 1 in [0, W]:
// x0 can only leave the plane at the low end and x1 only at the HIGH end - x1 == W (and y1 == H) does occur, for the
// last half texel before +1. Its weight is zero, but its address is still formed: cy * W + W runs into the next row, and
// (H) * W + x is a whole row past the plane - for the last feature plane of the last image that is up to W * 128 B
// beyond the allocation. (A round-1 experiment dropped exactly these two upper clamps - "valid points are in range" -
// and faulted intermittently at full frame size; this is why.) So: x0 needs max(., 0), x1 needs min(., W - 1), nothing
// else. Coordinates outside (-1, 1) are NOT handled (the callers' lanes for invalid pairs are masked off before any
// address is used); tests/host/taps_check.hip sweeps every float near +-1 against make_taps.
__host__ __device__ __forceinline__ Taps make_taps_valid(float x, float y, int H, int W) {
    Taps t;
    float ix, iy;
    {
#pragma clang fp contract(off)
        ix = ((x + 1.0f) * (float)W - 1.0f) / 2.0f;
        iy = ((y + 1.0f) * (float)H - 1.0f) / 2.0f;
    }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float ax1 = ix - fx, ax0 = (fx + 1.0f) - ix;
    const float ay1 = iy - fy, ay0 = (fy + 1.0f) - iy;
    const int cx0 = max(x0, 0), cx1 = min(x1, W - 1);
    const int cy0 = max(y0, 0), cy1 = min(y1, H - 1);
    const int r0 = mul_rc(cy0, W), r1 = mul_rc(cy1, W);
    t.o00 = r0 + cx0; t.o01 = r0 + cx1; t.o10 = r1 + cx0; t.o11 = r1 + cx1;
    const float zx0 = (x0 >= 0) ? ax0 : 0.0f, zx1 = (x1 < W) ? ax1 : 0.0f;
    const float zy0 = (y0 >= 0) ? ay0 : 0.0f, zy1 = (y1 < H) ? ay1 : 0.0f;
    t.w00 = zx0 * zy0;
    t.w01 = zx1 * zy0;
    t.w10 = zx0 * zy1;
    t.w11 = zx1 * zy1;
    t.xe = (x0 < 0) ? 1 : (x1 >= W) ? 2 : 0;
    t.yp = cy0 & 1;
    return t;
}
// The two taps of one row of a scalar (fp32, row-major) plane with ONE 8-byte load: x0 and x0 + 1 are adjacent floats.
// At the left edge both taps are column 0 (pair = columns 0, 1: both taps take .x), at the right edge both are column
// W - 1 (pair = columns W - 2, W - 1: both take .y; the pair must not start at W - 1, the float behind the last row of
// the last plane is not ours). The address is only 4-byte aligned: gfx950 under HSA runs with unaligned access enabled.
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void load_row_pair(const char *__restrict__ plane_bytes, unsigned plane_off, int o_left, int xe,
                                              float &left, float &right) {
    const unsigned e = (unsigned)(o_left - (xe == 2 ? 1 : 0));
    const f32x2_a4 v = *reinterpret_cast<const f32x2_a4 *>(plane_bytes + (plane_off + (e << 2)));
    left = (xe == 2) ? v.y : v.x;
    right = (xe == 1) ? v.x : v.y;
}
__device__ __forceinline__ float sample_scalar_plane(const float *__restrict__ plane, float x, float y,
                                                     int H, int W) {
    const Taps t = make_taps(x, y, H, W);
    float acc = plane[t.o00] * t.w00;
    acc += plane[t.o01] * t.w01;
    acc += plane[t.o10] * t.w10;
    acc += plane[t.o11] * t.w11;
    return acc;
}
// sigmoid from the hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each; the part probabilities feed a 1e-4
// comparison, not a bit-exact one). exp2 overflows to +inf for v < -88: 1 / inf = 0, the right limit.
__device__ __forceinline__ float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// StyledConv activation: LeakyReLU(0.2) * sqrt(2)   (libraries/custom_stylegan2/net.py:318)
__device__ __forceinline__ float styled_act(float v) {
    return fmaxf(v, 0.2f * v) * 1.41421356237309515f;      // max(v, 0.2 v) == leaky_relu(v, 0.2)
}

// ---- NCHW -> channel-last re-layout of one (b, plane, row, 64-column) block ----------------------------------------------
// in: (B, in_ch_total, H, W), planes p = 0..2 at channels [p*C, (p+1)*C); out: [b][p][y][x][C]; tile: C*65 floats of LDS.
// Full blocks of aligned planes move 16 B per lane both ways (x-runs of 4 in, channel-runs of 4 out) through the padded
// tile; ragged blocks, odd widths and unaligned tensors take the 4-byte path.
template <int C>
__device__ __forceinline__ void pack_block(const float *__restrict__ in, float *__restrict__ out, int in_ch_total, int H,
                                           int W, int xblk, int y, int bp, int tid, float *tile) {
    const int xb = xblk * 64, b = bp / 3, p = bp % 3;
    const float *src = in + (((size_t)b * in_ch_total + p * C) * H + y) * W;
    float *dst = out + ((((size_t)b * 3 + p) * H + y) * W + xb) * C;
    const bool wide = (C % 4 == 0) && (W % 4 == 0) && (xb + 64 <= W) &&
                      (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0);      // block-uniform
    if (wide) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (int v = tid; v < C * 16; v += 256) {
            const int c = v >> 4, x4 = (v & 15) << 2;
            const f4 q = *reinterpret_cast<const f4 *>(src + (size_t)c * H * W + xb + x4);
            float *t = tile + c * 65 + x4;
            t[0] = q[0]; t[1] = q[1]; t[2] = q[2]; t[3] = q[3];
        }
        __syncthreads();
        constexpr int C4 = C / 4 > 0 ? C / 4 : 1;
        for (int v = tid; v < 64 * C4; v += 256) {
            const int x = v / C4, c = (v % C4) << 2;
            const float *t = tile + c * 65 + x;
            *reinterpret_cast<f4 *>(dst + x * C + c) = f4{t[0], t[65], t[130], t[195]};
        }
        return;
    }
    const int x = tid & 63;
    for (int c = tid >> 6; c < C; c += 4)
        tile[c * 65 + x] = (xb + x < W) ? src[(size_t)c * H * W + xb + x] : 0.0f;
    __syncthreads();
    const int nvalid = min(64, W - xb) * C;
    for (int o = tid; o < nvalid; o += 256) dst[o] = tile[(o % C) * 65 + (o / C)];
}

// ---- bf16 helpers -------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);          // inputs here are finite weights / activations
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// fp16 with saturation (so that hi stays finite and lo = x - hi carries the rest)
__device__ __forceinline__ _Float16 f32_to_f16_sat(float f) {
    return (_Float16)fminf(fmaxf(f, -65504.0f), 65504.0f);
}

// ---- Philox4x32-10 (counter-based RNG for the importance samples) -----------------------------------
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0,1)

}  // namespace enarf
