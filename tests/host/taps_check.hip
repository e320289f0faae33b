// Host-side sweep of the bilinear tap index math of csrc/enarf_device.h (test infrastructure; built and run by
// tests/test_host_cpu.py::test_tap_offsets_stay_inside_the_plane with hipcc, host code only - no device is touched).
//   * make_taps (fully clamped) must give in-plane offsets for ANY input, and zero weights for out-of-plane taps;
//   * make_taps_valid (the light form used for valid pairs) must give in-plane offsets and the SAME weights and, where a
//     weight is non-zero, the same offsets as make_taps for every float in (-1, 1): all floats within 4096 ulp of -1
//     and of +1, every float around each of the W half-texel boundaries, and a coarse sweep in between.
// Prints "ok <count>" or the first violation.
#include "enarf_device.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

using namespace enarf;

static float next_up(float v, int n) {
    for (int i = 0; i < n; ++i) v = std::nextafterf(v, 2.0f);
    return v;
}

int main() {
    const int sizes[][2] = {{256, 256}, {24, 40}, {16, 16}, {1, 7}};
    long long checked = 0;
    for (auto &hw : sizes) {
        const int H = hw[0], W = hw[1];
        std::vector<float> xs;
        float v = std::nextafterf(-1.0f, 0.0f);
        for (int i = 0; i < 4096; ++i) { xs.push_back(v); v = std::nextafterf(v, 2.0f); }
        v = std::nextafterf(1.0f, 0.0f);
        for (int i = 0; i < 4096; ++i) { xs.push_back(v); v = std::nextafterf(v, -2.0f); }
        for (int k = 0; k <= 2 * W; ++k) {           // texel centres and texel edges
            float c = (float)k / (float)W - 1.0f;
            float lo = c;
            for (int i = 0; i < 8; ++i) lo = std::nextafterf(lo, -2.0f);
            for (int i = 0; i < 17; ++i) { if (std::fabs(lo) < 1.0f) xs.push_back(lo); lo = std::nextafterf(lo, 2.0f); }
        }
        for (int i = 0; i < 2000; ++i) xs.push_back(-0.999f + 1.998f * (float)i / 1999.0f);
        // general form: arbitrary coordinates, incl. far outside and non-finite
        const float wild[] = {-1.0f, 1.0f, -1.5f, 1.5f, 2.0f, -3.0e38f, 3.0e38f, INFINITY, -INFINITY, NAN, 0.0f};
        for (float a : wild)
            for (float b : wild) {
                const Taps t = make_taps(a, b, H, W);
                const int o[4] = {t.o00, t.o01, t.o10, t.o11};
                for (int i = 0; i < 4; ++i)
                    if (o[i] < 0 || o[i] >= H * W) { std::printf("make_taps(%g, %g) H=%d W=%d: offset %d out of plane\n", a, b, H, W, o[i]); return 1; }
                ++checked;
            }
        // valid form against the general form: x sweep with a few y, then y sweep with a few x (H != W covers both axes)
        const float others[] = {std::nextafterf(-1.0f, 0.0f), -0.37f, 0.0f, 0.61f, std::nextafterf(1.0f, 0.0f)};
        for (int axis = 0; axis < 2; ++axis)
            for (float s : xs)
                for (float oth : others) {
                    const float x = axis == 0 ? s : oth, y = axis == 0 ? oth : s;
                    const Taps g = make_taps(x, y, H, W), t = make_taps_valid(x, y, H, W);
                    const int og[4] = {g.o00, g.o01, g.o10, g.o11}, ot[4] = {t.o00, t.o01, t.o10, t.o11};
                    const float wg[4] = {g.w00, g.w01, g.w10, g.w11}, wt[4] = {t.w00, t.w01, t.w10, t.w11};
                    for (int i = 0; i < 4; ++i) {
                        if (ot[i] < 0 || ot[i] >= H * W) { std::printf("make_taps_valid(%.9g, %.9g) H=%d W=%d: offset %d out of plane\n", x, y, H, W, ot[i]); return 1; }
                        if (std::memcmp(&wg[i], &wt[i], 4) != 0) { std::printf("make_taps_valid(%.9g, %.9g): weight %d differs (%g vs %g)\n", x, y, i, wt[i], wg[i]); return 1; }
                        if (wg[i] != 0.0f && og[i] != ot[i]) { std::printf("make_taps_valid(%.9g, %.9g): offset %d differs (%d vs %d)\n", x, y, i, ot[i], og[i]); return 1; }
                    }
                    // the row-pair form of the scalar taps (load_row_pair, enarf_device.h): the 8-byte pair starts inside the
                    // row, ends inside the row, and its two selected elements are exactly the taps o00 / o01 (o10 / o11)
                    for (const Taps &q : {g, t}) {
                        if (q.xe != g.xe) { std::printf("xe differs at (%.9g, %.9g)\n", x, y); return 1; }
                        const int rows[2][2] = {{q.o00, q.o01}, {q.o10, q.o11}};
                        for (const auto &r : rows) {
                            const int e = r[0] - (q.xe == 2 ? 1 : 0), row = r[0] / W;
                            if (e < row * W || e + 1 > row * W + W - 1) { std::printf("row pair leaves the row at (%.9g, %.9g) W=%d: %d\n", x, y, W, e); return 1; }
                            const int left = (q.xe == 2) ? e + 1 : e, right = (q.xe == 1) ? e : e + 1;
                            if (left != r[0] || right != r[1]) { std::printf("row pair selects %d, %d for taps %d, %d at (%.9g, %.9g)\n", left, right, r[0], r[1], x, y); return 1; }
                        }
                    }
                    ++checked;
                }
        // the case behind the round-1 fault: the last half texel before +1 has x1 == W (y1 == H)
        const float edge = std::nextafterf(1.0f, 0.0f);
        float ix;
        {
#pragma clang fp contract(off)
            ix = ((edge + 1.0f) * (float)W - 1.0f) / 2.0f;
        }
        if ((int)std::floor(ix) + 1 != W) { std::printf("expected x1 == W at the last float below 1 (W=%d, ix=%g)\n", W, ix); return 1; }
    }
    std::printf("ok %lld\n", checked);
    return 0;
}
