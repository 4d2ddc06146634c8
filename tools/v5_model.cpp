// v5_model.cpp -- CPU model of the DATAFLOW of csrc/smx_agg_v5.hip (comb lanes, register rings, band tiles,
// strip hand-off), lane by lane and phase by phase, checked bit for bit against the oracle.  Design tool and
// regression check for the index arithmetic of the kernel; test infrastructure only (links the oracle).
//
//   g++ -O2 -ffp-contract=off -std=c++17 tools/v5_model.cpp -o /tmp/v5_model -Loracle/_build -lsmx_oracle \
//       -Wl,-rpath,$PWD/oracle/_build && /tmp/v5_model
//
// Geometry (radius 9): a strip = NR = 19 combs (one per residue mod 19) of L = 16 lanes; lane (r, i) owns the
// integral-image column base + 19 i + r, so the left box tap (19 columns to the left) is always lane i-1 of the
// same 16-lane DPP row, and the 19 rows of history a box needs live in a 20-slot register ring of the lane.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

extern "C" {
typedef struct orc_params {
    double r_w, g_w, b_w, alpha;
    int th_color, th_grad, radius;
    double eps;
    int d_lr;
} orc_params;
void orc_default_params(orc_params*);
void orc_xderiv(const uint8_t*, float*, int, int);
void orc_cost_slice(const orc_params*, const uint8_t*, const uint8_t*, const float*, const float*, float*, int, int, int, int);
void orc_guidance(const orc_params*, const uint8_t*, float*, float*, float*, uint8_t*, int, int);
void orc_guided_filter(const orc_params*, const uint8_t*, const float*, float*, float*, uint8_t*, float*, int, int, int, int, int);
void orc_init_wta(float*, float*, int64_t);
}

#ifndef MODEL_L
#define MODEL_L 16
#endif
constexpr int R = 9, HW = 19, L = MODEL_L, NR = 19, SW = NR * L, OWS = HW * (L - 1), BH = 10, RD = 20;
static const float NZ = -0.0f;

struct f2 { float x, y; };

struct Rec {                       // hand-off record of one band
    f2 carry1[BH];                 // stage-1 running row sums behind tile column OWS - 1
    f2 halo2[BH][HW];              // stage-2 row prefix of the strip's last 19 columns
};

static int g_w, g_h;
static std::vector<float> g_p, g_im, g_mean, g_c;   // cost slice, image as float, mean_I, 1/(var+eps)

static float rcp_area(int a) { return 1.0f / (float)a; }
static float fastdiv(float x, float d, float r) {   // Markstein form of the kernel
    float q = x * r;
    float e = fmaf(-q, d, x);
    return fmaf(e, r, q);
}

struct Lane { f2 ring[RD]; };   // ring[y mod RD] = S[y]; the slot of row y-1 is the running column sum

// one strip of one slice; records in: rin (k > 0), out: rout
static void run_strip(int k, int K, int NI, const std::vector<Rec>* rin, std::vector<Rec>* rout, std::vector<float>& q) {
    const int w = g_w, h = g_h;
    const int base1 = OWS * k - 1, base2 = OWS * k - 10;
    std::vector<f2> tile1(BH * SW), tile2(BH * SW, f2{NZ, NZ});
    std::vector<Lane> l1(SW), l2(SW);
    for (auto* ls : {&l1, &l2})
        for (auto& l : *ls)
            for (auto& s : l.ring) s = f2{0.0f, 0.0f};
    // the slot in front of the first row holds the initial running sum -0 (stage 1: row 0 -> slot 19; stage 2: row -9 -> slot 10)
    for (auto& l : l1) l.ring[RD - 1] = f2{NZ, NZ};
    for (auto& l : l2) l.ring[10] = f2{NZ, NZ};
    std::vector<f2> abreg(SW * BH, f2{NZ, NZ});
    auto lane_j = [](int lane) { int r = lane / L, i = lane % L; return HW * i + r; };   // tile column of a comb lane
    for (int i = 0; i < NI; ++i) {
        // ---- W(i): (p, I p) of band i -> tile 1; a/b of band i-1 (registers) and the halo -> tile 2
        for (int t = 0; t < BH; ++t)
            for (int j = 0; j < SW; ++j) {
                const int y = BH * i + t, c = base1 + j;
                f2 v{NZ, NZ};
                if (y < h && c >= 0 && c < w) {
                    const float p = g_p[(size_t)y * w + c];
                    v = f2{p, g_im[(size_t)y * w + c] * p};
                }
                tile1[t * SW + j] = v;
            }
        if (i >= 1) {
            for (int lane = 0; lane < SW; ++lane) {
                const int il = lane % L, j = lane_j(lane);
                if (k > 0 && il == 0) continue;                       // halo columns: from the neighbour
                for (int t = 0; t < BH; ++t) {
                    const int ya = BH * (i - 1) - R + t, x = base2 + j;
                    const bool ok = ya >= 0 && ya < h && x >= 0 && x < w;
                    tile2[t * SW + j] = ok ? abreg[lane * BH + t] : f2{NZ, NZ};
                }
            }
            if (k > 0)
                for (int t = 0; t < BH; ++t)
                    for (int j = 0; j < HW; ++j) tile2[t * SW + j] = (*rin)[i].halo2[t][j];
        }
        // ---- R(i): sequential row prefix sums, in place
        for (int t = 0; t < BH; ++t) {
            const int y = BH * i + t;
            if (y < h) {
                f2 acc = k > 0 ? (*rin)[i].carry1[t] : f2{NZ, NZ};
                for (int j = 0; j < SW; ++j) {
                    f2& v = tile1[t * SW + j];
                    acc.x = v.x + acc.x; acc.y = v.y + acc.y;
                    v = acc;
                }
            }
            const int ya = BH * (i - 1) - R + t;
            if (i >= 1 && ya >= 0 && ya < h) {
                f2 acc = k > 0 ? tile2[t * SW + HW - 1] : f2{NZ, NZ};
                for (int j = k > 0 ? HW : 0; j < SW; ++j) {
                    f2& v = tile2[t * SW + j];
                    acc.x = v.x + acc.x; acc.y = v.y + acc.y;
                    v = acc;
                }
            }
        }
        if (rout) {
            Rec& rc = (*rout)[i];
            for (int t = 0; t < BH; ++t) {
                rc.carry1[t] = tile1[t * SW + OWS - 1];
                for (int j = 0; j < HW; ++j) rc.halo2[t][j] = tile2[t * SW + OWS + j];
            }
        }
        // ---- X(i): comb lanes, LANE = one integral-image column
        auto box = [&](std::vector<Lane>& ls, int lane, int slot, int slot01, bool fix_first) {
            // S11 - S10 - S01 + S00 with the DPP row_shr:1 taps; lane i == 0 of a DPP row has no source lane:
            // the operation is skipped there
            const int il = lane % L;
            f2 u = ls[lane].ring[slot];
            // (zero-filled DPP source for lane i == 0: u - (+0), u + (+0))
            const float left0 = L == 16 ? 0.0f : NAN;          // row_shr:1 zero fill / wave_shr:1: whatever the lane in front holds
            if (il > 0) { u.x -= ls[lane - 1].ring[slot].x; u.y -= ls[lane - 1].ring[slot].y; } else { u.x -= left0; u.y -= left0; }
            u.x -= ls[lane].ring[slot01].x; u.y -= ls[lane].ring[slot01].y;
            if (il > 0) { u.x += ls[lane - 1].ring[slot01].x; u.y += ls[lane - 1].ring[slot01].y; } else { u.x += left0; u.y += left0; }
            // combs that are not whole DPP rows: the first lane of a comb of strip 0 (stage 1: it has outputs, columns
            // 0 .. 8) takes the box without left taps instead
            if (L != 16 && fix_first && il == 0)
                u = f2{ls[lane].ring[slot].x - ls[lane].ring[slot01].x, ls[lane].ring[slot].y - ls[lane].ring[slot01].y};
            return u;
        };
        auto area_of = [&](int x, int y) {
            const int xw = std::min(w - 1, x + R) - std::max(-1, x - R - 1);
            const int yh = std::min(h - 1, y + R) - std::max(-1, y - R - 1);
            return xw * yh;
        };
        for (int t = 0; t < BH; ++t) {
            // stage 1, row y1 = BH i + t
            {
                const int y1 = BH * i + t, slot = y1 % RD, slot01 = (y1 + 1) % RD, slotp = (y1 + RD - 1) % RD;
                for (int lane = 0; lane < SW; ++lane) {        // column sums first (lockstep: every lane updates S ...)
                    const f2 rv = tile1[t * SW + lane_j(lane)];
                    l1[lane].ring[slot] = f2{rv.x + l1[lane].ring[slotp].x, rv.y + l1[lane].ring[slotp].y};
                }
                std::vector<f2> u(SW);
                for (int lane = 0; lane < SW; ++lane) u[lane] = box(l1, lane, slot, slot01, k == 0);   // ... then the taps
                for (int lane = 0; lane < SW; ++lane) {
                    const int y = y1 - R, x = base1 + lane_j(lane) - R;
                    f2 ab{NZ, NZ};
                    if (y >= 0 && y < h && x >= 0 && x < w) {
                        const int ar = area_of(x, y);
                        const float mp = fastdiv(u[lane].x, (float)ar, rcp_area(ar));
                        const float mIp = fastdiv(u[lane].y, (float)ar, rcp_area(ar));
                        const float mI = g_mean[(size_t)y * w + x], c = g_c[(size_t)y * w + x];
                        const float mm = mI * mp;
                        const float ak = (mIp - mm) * c;
                        const float mb = mI * ak;
                        ab = f2{ak, mp - mb};
                    }
                    abreg[lane * BH + t] = ab;
                }
            }
            // stage 2, a/b row y2 = BH (i-1) - R + t
            if (i >= 1) {
                const int y2 = BH * (i - 1) - R + t;
                const int slot = ((y2 % RD) + RD) % RD, slot01 = (((y2 + 1) % RD) + RD) % RD, slotp = (((y2 - 1) % RD) + RD) % RD;
                for (int lane = 0; lane < SW; ++lane) {
                    const f2 rv = tile2[t * SW + lane_j(lane)];
                    l2[lane].ring[slot] = f2{rv.x + l2[lane].ring[slotp].x, rv.y + l2[lane].ring[slotp].y};
                }
                std::vector<f2> u(SW);
                for (int lane = 0; lane < SW; ++lane) u[lane] = box(l2, lane, slot, slot01, false);
                for (int lane = 0; lane < SW; ++lane) {
                    const int il = lane % L;
                    const int y = y2 - R, x = base2 + lane_j(lane) - R;
                    if (il >= 1 && y >= 0 && y < h && x >= 0 && x < w) {
                        const int ar = area_of(x, y);
                        const float ma = fastdiv(u[lane].x, (float)ar, rcp_area(ar));
                        const float mb = fastdiv(u[lane].y, (float)ar, rcp_area(ar));
                        const float tq = ma * g_im[(size_t)y * w + x];
                        q[(size_t)y * w + x] = tq + mb;
                    }
                }
            }
        }
    }
    (void)K;
}

static int check(int w, int h, int d, unsigned seed) {
    orc_params P;
    orc_default_params(&P);
    std::mt19937 rng(seed);
    std::vector<uint8_t> I1((size_t)w * h), I2((size_t)w * h);
    // smooth-ish texture so that costs are not all saturated
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int v = (int)(127 + 100 * sin(0.13 * x + 0.07 * y) + (int)(rng() % 41) - 20);
            I1[(size_t)y * w + x] = (uint8_t)std::min(255, std::max(0, v));
            int x2 = x + 3;
            int v2 = (int)(127 + 100 * sin(0.13 * x2 + 0.07 * y) + (int)(rng() % 41) - 20);
            I2[(size_t)y * w + x] = (uint8_t)std::min(255, std::max(0, v2));
        }
    const size_t n = (size_t)w * h;
    std::vector<float> g1(n), g2(n);
    orc_xderiv(I1.data(), g1.data(), w, h);
    orc_xderiv(I2.data(), g2.data(), w, h);
    g_w = w; g_h = h;
    g_p.assign(n, 0.f);
    orc_cost_slice(&P, I1.data(), I2.data(), g1.data(), g2.data(), g_p.data(), w, w, h, d);
    g_im.assign(n, 0.f); g_mean.assign(n, 0.f); g_c.assign(n, 0.f);
    std::vector<float> var(n);
    orc_guidance(&P, I1.data(), g_im.data(), g_mean.data(), var.data(), nullptr, w, h);
    for (size_t i = 0; i < n; ++i) g_c[i] = (float)(1.0f / ((double)var[i] + P.eps));
    // oracle
    std::vector<float> best(n), dmap(n), agg(n);
    orc_init_wta(best.data(), dmap.data(), (int64_t)n);
    orc_guided_filter(&P, I1.data(), g_p.data(), best.data(), dmap.data(), nullptr, agg.data(), w, h, 0, 0, 1);
    // model
    const int K = (w + OWS - 1) / OWS, NI = (h + 2 * R + BH - 1) / BH + 1;
    std::vector<float> q(n, NAN);
    std::vector<Rec> ra(NI), rb(NI);
    for (int k = 0; k < K; ++k) {
        std::vector<Rec>& in = (k & 1) ? ra : rb;
        std::vector<Rec>& out = (k & 1) ? rb : ra;
        run_strip(k, K, NI, k > 0 ? &in : nullptr, k + 1 < K ? &out : nullptr, q);
    }
    size_t bad = 0, first = (size_t)-1;
    for (size_t i = 0; i < n; ++i)
        if (memcmp(&q[i], &agg[i], 4) != 0) { if (!bad) first = i; ++bad; }
    printf("%4d x %4d d=%3d: K=%d NI=%d  %s", w, h, d, K, NI, bad ? "MISMATCH" : "bit-exact");
    if (bad) printf("  %zu cells, first (y=%zu, x=%zu) model %.9g oracle %.9g", bad, first / w, first % w, q[first], agg[first]);
    printf("\n");
    return bad ? 1 : 0;
}

int main() {
    int rc = 0;
    const int shapes[][2] = {{384, 288}, {285, 40}, {286, 41}, {284, 39}, {570, 25}, {571, 10}, {600, 9}, {40, 30},
                             {19, 19}, {10, 10}, {1, 1}, {3, 50}, {1242, 375}, {865, 61}, {300, 20}, {295, 11}};
    for (auto& s : shapes) rc |= check(s[0], s[1], -3, 1234u + s[0]);
    rc |= check(384, 288, 0, 7);
    rc |= check(700, 33, -650, 8);
    return rc;
}
