// Micro-benchmark of the gather rate the CUs' vector-memory (texture) path sustains for the march's access shape, and of
// what EXEC-masked lanes cost it (tools/, measurement only; results in profiles/r02_ub_ta.txt).
// Each wave issues rounds of 24 global_load_dwordx4: 12 texels of 128 B per quad, 4 lanes per texel, 32 B per lane -
// exactly one gather round of enarf_query.h - from a 25 MB channel-last table (one image's feature planes), with `active`
// of the wave's 16 quads enabled. pattern 0: 12 independent random texels; pattern 1: 3 random 2x2 bilinear footprints
// (taps x, x+1 adjacent 128-B lines; y, y+1 one row apart), consecutive quads a few texels apart like samples along a ray.
// pattern 2: as 1, but a lane's two 16-B loads are 64 B apart (each instruction covers a contiguous 64 B per quad) instead
// of adjacent; pattern 3: as 1 plus the 4 scalar (dword) part-probability taps of a round; pattern 4: as 1 with half-size
// texels (64 B, one 16-B load per lane and texel - what fp16 feature planes would cost). Round 3: pattern 5: as 1 plus the
// part-probability taps as the march issues them today - two unaligned 8-byte row pairs per lane from one of 69 scalar
// planes (18 MB); pattern 6: as 1 plus ONE aligned 16-byte load per lane from a table that stores every scalar texel with
// its 2x2 neighbourhood (72 MB) - the layout VERDICT r02 item 4(i) proposes.
// 768 workgroups x 4 waves (3 waves per SIMD, as the march runs). Build: hipcc --offload-arch=gfx950 -O3 tools/ub_ta.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256, 3) void gather(const float *__restrict__ tab, float *out, int rounds, int active_quads,
                                                  const float *__restrict__ mtab, const float *__restrict__ mtab4) {
    constexpr unsigned W = 256, PLANE = W * W;
    const int lane = threadIdx.x & 63, quad = lane >> 2, g = lane & 3;
    unsigned s = ((blockIdx.x * 256 + threadIdx.x) >> 8) * 2654435761u + 12345u;       // per wave stream (wave-uniform)
    f32x4 acc = {0, 0, 0, 0};
    const bool on = quad < active_quads;
    for (int r = 0; r < rounds; ++r) {
        unsigned tex[12];
        constexpr int FOOT = PATTERN == 0 ? 0 : 1;
        if (FOOT == 0) {
            unsigned q = s + quad * 97u;
#pragma unroll
            for (int t = 0; t < 12; ++t) { q = q * 1664525u + 1013904223u; tex[t] = (q >> 8) % (3 * PLANE); }
        } else {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                s = s * 1664525u + 1013904223u;
                const unsigned x = ((s >> 8) % (W - 18)) + quad, y = ((s >> 20) % (W - 2)) + (quad >> 3);   // a short run along x
                const unsigned o = p * PLANE + y * W + x;
                tex[4 * p] = o; tex[4 * p + 1] = o + 1; tex[4 * p + 2] = o + W; tex[4 * p + 3] = o + W + 1;
            }
        }
        s = s * 1664525u + 1013904223u;
        if (on) {
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                if (PATTERN == 2) {
                    const f32x4 *p = reinterpret_cast<const f32x4 *>(tab + (size_t)tex[t] * 32 + g * 4);
                    acc += p[0];
                    acc += p[4];
                } else if (PATTERN == 4) {
                    acc += *reinterpret_cast<const f32x4 *>(tab + (size_t)tex[t] * 16 + g * 4);
                } else {
                    const f32x4 *p = reinterpret_cast<const f32x4 *>(tab + (size_t)tex[t] * 32 + g * 8);
                    acc += p[0];
                    acc += p[1];
                }
            }
            if (PATTERN == 5 || PATTERN == 6) {      // lane g of the quad: plane g % 3 of a part the quad picked
                typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
                const unsigned part = (s >> 5) % 23u + (quad & 1);                       // neighbouring samples: mostly the same part
                const unsigned o = tex[4 * (g % 3)] - (g % 3) * PLANE;
                const size_t e = (size_t)(3 * (part % 23u) + g % 3) * PLANE + o;
                if (PATTERN == 5) {
                    const f32x2_a4 a = *reinterpret_cast<const f32x2_a4 *>(mtab + e), b = *reinterpret_cast<const f32x2_a4 *>(mtab + e + W);
                    acc[0] += a.x + a.y + b.x + b.y;
                } else {
                    acc += *reinterpret_cast<const f32x4 *>(mtab4 + 4 * e);
                }
            }
            if (PATTERN == 3) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[0] += tab[(size_t)((tex[4 * (g % 3) + t] * 7u) % (3 * PLANE)) * 32 + 5];
            }
        }
    }
    if (acc[0] == 123.456f) out[threadIdx.x] = acc[1] + acc[2] + acc[3];
}

int main() {
    const size_t bytes = (size_t)3 * 256 * 256 * 128;
    float *tab, *out;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&out, 1024) != hipSuccess) return 1;
    (void)hipMemset(tab, 0, bytes);
    float *mtab, *mtab4;
    const size_t mbytes = (size_t)69 * 256 * 256 * 4 + 4096;
    if (hipMalloc(&mtab, mbytes) != hipSuccess || hipMalloc(&mtab4, 4 * mbytes) != hipSuccess) return 1;
    (void)hipMemset(mtab, 0, mbytes);
    (void)hipMemset(mtab4, 0, 4 * mbytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int rounds = 250, wgs = 768;
    for (int pattern = 0; pattern < 7; ++pattern)
        for (int active : {16, 12, 8, 4}) {
            if (pattern >= 2 && active != 16 && active != 12) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0);
                switch (pattern) {
                case 0: hipLaunchKernelGGL(gather<0>, dim3(wgs), dim3(256), 0, 0, tab, out, rounds, active, mtab, mtab4); break;
                case 1: hipLaunchKernelGGL(gather<1>, dim3(wgs), dim3(256), 0, 0, tab, out, rounds, active, mtab, mtab4); break;
                case 2: hipLaunchKernelGGL(gather<2>, dim3(wgs), dim3(256), 0, 0, tab, out, rounds, active, mtab, mtab4); break;
                case 3: hipLaunchKernelGGL(gather<3>, dim3(wgs), dim3(256), 0, 0, tab, out, rounds, active, mtab, mtab4); break;
                case 4: hipLaunchKernelGGL(gather<4>, dim3(wgs), dim3(256), 0, 0, tab, out, rounds, active, mtab, mtab4); break;
                case 5: hipLaunchKernelGGL(gather<5>, dim3(wgs), dim3(256), 0, 0, tab, out, rounds, active, mtab, mtab4); break;
                default: hipLaunchKernelGGL(gather<6>, dim3(wgs), dim3(256), 0, 0, tab, out, rounds, active, mtab, mtab4); break;
                }
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double b = (double)wgs * 4 * rounds * active * 12 * (pattern == 4 ? 64 : 128);
            printf("pattern %d active quads %2d/16: %.3f ms  %.2f TB/s gathered  %.0f ns per wave-round (%d rounds)\n", pattern, active,
                   best, b / best / 1e9, best * 1e6 / rounds, rounds);
        }
    return 0;
}
