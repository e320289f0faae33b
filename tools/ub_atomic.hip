// Micro-benchmark of the float-atomic rate the memory side sustains for the backward's access shapes (tools/, measurement
// only; results in profiles/r03_ub_atomic.txt). The backward's scatter pass (enarf_render_bwd.hip, F4) issues no-return
// global_atomic_add_f32 whose half-waves each cover one 128-B line of the channel-last gradient planes (32 channels of one
// texel: two 64-B requests), and scalar adds into the part-probability planes whose adjacent lanes pair up on 8 bytes.
// pattern 0: every wave instruction adds 256 contiguous, aligned bytes (4 requests) at a random place: the best case;
// pattern 1: every half-wave its own random 128-B line of a 25 MB table (one image's feature planes);
// pattern 2: as 1, but a wave's consecutive instructions walk neighbouring texels as the samples of a ray do (x, x + 1 in the
//            two half-waves, a step of 0..2 texels per instruction, a row change every 8): the L2-friendly form of 1;
// pattern 3: adjacent lane pairs add 8 bytes each at 32 random places of an 18 MB table (the part-probability adds);
// pattern 4: as 1 with every lane on its own random dword (nothing merges: one request per lane).
// Each at the backward's residency (512 workgroups of 256, two waves per SIMD) and at 2 048 workgroups (eight per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ub_atomic.hip -o variants/ub_atomic
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PATTERN, int WPS>
__global__ __launch_bounds__(256, WPS) void adds(float *__restrict__ tab, float *__restrict__ mtab, int rounds) {
    constexpr unsigned W = 256, PLANE = W * W, LINES = 3 * PLANE;
    const int lane = threadIdx.x & 63, half = lane >> 5, ch = lane & 31;
    unsigned s = ((blockIdx.x * 256 + threadIdx.x) >> 6) * 2654435761u + 12345u;       // per wave stream (wave-uniform)
    unsigned x = 0, y = 0, p = 0;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            s = s * 1664525u + 1013904223u;
            if (PATTERN == 0) {
                const unsigned o = (s >> 8) % (LINES / 2);
                atomicAdd(tab + (size_t)o * 64 + lane, 1.0f);
            } else if (PATTERN == 1) {
                const unsigned o = ((s >> 8) * (half ? 40503u : 1u) + half * 977u) % LINES;
                atomicAdd(tab + (size_t)o * 32 + ch, 1.0f);
            } else if (PATTERN == 2) {
                if (t == 0) { p = (s >> 4) % 3u; x = (s >> 8) % (W - 20); y = (s >> 20) % (W - 2); }
                x += (s >> 28) % 3u;
                const unsigned o = p * PLANE + (y + (t & 1)) * W + x + half;
                atomicAdd(tab + (size_t)o * 32 + ch, 1.0f);
            } else if (PATTERN == 3) {
                const unsigned q = (s >> 8) + (lane >> 1) * 2246822519u;
                atomicAdd(mtab + (size_t)(q % (69u * PLANE - 2)) + (lane & 1), 1.0f);
            } else {
                const unsigned q = (s >> 8) + lane * 2246822519u;
                atomicAdd(tab + (size_t)(q % (LINES * 32u)), 1.0f);
            }
        }
    }
}

// Do a CU's loads wait behind its atomics? Waves 0-1 of every workgroup add random 128-B lines (pattern 1), waves 2-3 gather
// random 128-B texels of a second 25 MB table (one 16-B load per lane, 8 texels per instruction, 12 instructions in flight).
// MODE 1: the adding waves alone; 2: the gathering waves alone; 3: both at once.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void mixed(float *__restrict__ tab, const float *__restrict__ src, float *out, int rounds_add, int rounds_ld) {
    constexpr unsigned LINES = 3 * 256 * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, ch = lane & 31;
    unsigned s = ((blockIdx.x * 256 + threadIdx.x) >> 6) * 2654435761u + 12345u;
    // MODE & 4: the roles are dealt by WORKGROUP instead (a quarter of the CUs only add, the others only gather)
    const bool adder = (MODE & 4) ? (blockIdx.x % 256u) < 64u : wave < 2;
    if (adder) {
        if (!(MODE & 1)) return;
        for (int r = 0; r < rounds_add; ++r) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                s = s * 1664525u + 1013904223u;
                const unsigned o = ((s >> 8) * (half ? 40503u : 1u) + half * 977u) % LINES;
                atomicAdd(tab + (size_t)o * 32 + ch, 1.0f);
            }
        }
    } else {
        if (!(MODE & 2)) return;
        f32x4 acc = {0, 0, 0, 0};
        for (int r = 0; r < rounds_ld; ++r) {
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                s = s * 1664525u + 1013904223u;
                const unsigned o = ((s >> 8) + (lane >> 3) * 2246822519u) % LINES;
                acc += *reinterpret_cast<const f32x4 *>(src + (size_t)o * 32 + (lane & 7) * 4);
            }
        }
        if (acc[0] == 123.456f) out[threadIdx.x] = acc[1] + acc[2] + acc[3];
    }
}

template <int PATTERN, int WPS>
static float run(float *tab, float *mtab, int rounds, int wgs, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((adds<PATTERN, WPS>), dim3(wgs), dim3(256), 0, 0, tab, mtab, rounds);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t bytes = (size_t)3 * 256 * 256 * 128, mbytes = (size_t)69 * 256 * 256 * 4 + 4096;
    float *tab, *mtab;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&mtab, mbytes) != hipSuccess) return 1;
    (void)hipMemset(tab, 0, bytes);
    (void)hipMemset(mtab, 0, mbytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[5] = {"256 B contiguous per instruction", "a random 128-B line per half-wave", "128-B lines along a ray's texel walk",
                            "8-byte lane pairs, scattered", "every lane its own dword"};
    const double req_per_instr[5] = {4, 4, 4, 32, 64}, bytes_per_instr[5] = {256, 256, 256, 256, 256};
    for (int pattern = 0; pattern < 5; ++pattern)
        for (int occ = 0; occ < 2; ++occ) {
            const int wgs = occ ? 2048 : 512, rounds = occ ? 60 : 240;
            float ms;
            switch (pattern * 2 + occ) {
            case 0: ms = run<0, 2>(tab, mtab, rounds, wgs, e0, e1); break;
            case 1: ms = run<0, 8>(tab, mtab, rounds, wgs, e0, e1); break;
            case 2: ms = run<1, 2>(tab, mtab, rounds, wgs, e0, e1); break;
            case 3: ms = run<1, 8>(tab, mtab, rounds, wgs, e0, e1); break;
            case 4: ms = run<2, 2>(tab, mtab, rounds, wgs, e0, e1); break;
            case 5: ms = run<2, 8>(tab, mtab, rounds, wgs, e0, e1); break;
            case 6: ms = run<3, 2>(tab, mtab, rounds, wgs, e0, e1); break;
            case 7: ms = run<3, 8>(tab, mtab, rounds, wgs, e0, e1); break;
            case 8: ms = run<4, 2>(tab, mtab, rounds, wgs, e0, e1); break;
            default: ms = run<4, 8>(tab, mtab, rounds, wgs, e0, e1); break;
            }
            const double instr = (double)wgs * 4 * rounds * 8;
            printf("pattern %d (%s), %4d workgroups: %.3f ms  %.2f G requests/s  %.3f TB/s added  %.2f G wave-instructions/s\n", pattern,
                   names[pattern], wgs, ms, instr * req_per_instr[pattern] / ms / 1e6, instr * bytes_per_instr[pattern] / ms / 1e9,
                   instr / ms / 1e6);
        }
    // is the ceiling the chip's or the CU's? pattern 1 on fewer workgroups (one per CU up to 256)
    for (int wgs : {16, 32, 64, 128, 256, 512}) {
        const int rounds = 240;
        const float ms = run<1, 2>(tab, mtab, rounds, wgs, e0, e1);
        const double instr = (double)wgs * 4 * rounds * 8;
        printf("pattern 1 on %3d workgroups: %.3f ms  %.2f G requests/s  (%.1f M requests/s per workgroup)\n", wgs, ms, instr * 4 / ms / 1e6,
               instr * 4 / ms / 1e3 / wgs);
    }
    float *src, *out;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 1024) != hipSuccess) return 1;
    (void)hipMemset(src, 0, bytes);
    const int ra = 240, rl = 400, wgs = 512;
    float t[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 1) hipLaunchKernelGGL(mixed<1>, dim3(wgs), dim3(256), 0, 0, tab, src, out, ra, rl);
            else if (mode == 2) hipLaunchKernelGGL(mixed<2>, dim3(wgs), dim3(256), 0, 0, tab, src, out, ra, rl);
            else hipLaunchKernelGGL(mixed<3>, dim3(wgs), dim3(256), 0, 0, tab, src, out, ra, rl);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        t[mode] = best;
    }
    printf("mixed, 512 workgroups: adding waves alone %.3f ms (%.2f G requests/s), gathering waves alone %.3f ms (%.2f TB/s), both at once %.3f ms"
           "  (sum %.3f, max %.3f)\n", t[1], (double)wgs * 2 * ra * 8 * 4 / t[1] / 1e6, t[2], (double)wgs * 2 * rl * 12 * 1024 / t[2] / 1e9, t[3],
           t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
    {   // the same work with the roles dealt by workgroup: 128 adding workgroups (64 CUs), 384 gathering ones
        const int ra2 = ra * 2 * 512 / (128 * 4), rl2 = rl * 2 * 512 / (384 * 4);
        float u[4] = {0, 0, 0, 0};
        for (int mode = 1; mode <= 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 1) hipLaunchKernelGGL(mixed<5>, dim3(wgs), dim3(256), 0, 0, tab, src, out, ra2, rl2);
                else if (mode == 2) hipLaunchKernelGGL(mixed<6>, dim3(wgs), dim3(256), 0, 0, tab, src, out, ra2, rl2);
                else hipLaunchKernelGGL(mixed<7>, dim3(wgs), dim3(256), 0, 0, tab, src, out, ra2, rl2);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            u[mode] = best;
        }
        printf("roles by workgroup (64 CUs add, 192 gather): adding alone %.3f ms, gathering alone %.3f ms, both at once %.3f ms  (sum %.3f)\n",
               u[1], u[2], u[3], u[1] + u[2]);
    }
    return 0;
}
