// Shared device arithmetic of the NB / ZINB likelihood kernels (dcahip_zinb.hip: standalone
// loss + gradient pass; dcahip_heads.hip: the fused heads kernel).  gfx950, wave64.
//
// Reference arithmetic restated here: dca/network.py:38-39 (MeanAct, DispAct),
// dca/layers.py:21,85, dca/loss.py:72-114 (NB.loss), dca/loss.py:122-156 (ZINB.loss).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace {

constexpr float kEps = 1e-10f;        // loss.py:65
constexpr float kThetaMax = 1e6f;     // loss.py:85
constexpr float kZeroThresh = 1e-8f;  // loss.py:138
constexpr int kSmallY = 16;

// log(n!) for n = 0..16
__constant__ float kLogFact[17] = {
    0.0f, 0.0f, 0.69314718055994531f, 1.7917594692280550f, 3.1780538303479458f,
    4.7874917427820458f, 6.5792512120101012f, 8.5251613610654147f, 10.604602902745251f,
    12.801827480081469f, 15.104412573075516f, 17.502307845873887f, 19.987214495661885f,
    22.552163853123425f, 25.191221182738680f, 27.899271383840890f, 30.671860106080672f};

// ---- elementary functions on the transcendental unit (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp
// each) with first-order compensation of the range step, so that no IEEE division or libm call
// sits on the per-element path.  exp: the rounding of x*log2(e) is re-applied as a correction
// term; log: log2 result times ln2 in two pieces; log1p / expm1 use Kahan's exact-ratio forms.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float fexp(float x) {
    x = fminf(fmaxf(x, -104.f), 88.7f);
    const float L = 1.44269504088896340736f, Llo = 1.92596299112661746e-8f;
    const float t = x * L;
    const float r = fmaf(x, Llo, fmaf(x, L, -t));
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.69314718055994531f, e);
}

__device__ __forceinline__ float flog(float x) {           // x > 0, normal
    const float y = __builtin_amdgcn_logf(x);              // log2(x)
    const float C = 0.693147182464599609375f, Clo = -1.90465429995776804e-9f;
    const float r = y * C;
    return r + fmaf(y, Clo, fmaf(y, C, -r));
}

__device__ __forceinline__ float flog1p(float t) {         // t > -1
    const float u = 1.f + t;
    const float d = u - 1.f;
    const float l = flog(u);
    return d == 0.f ? t : l * (t * frcp(d));
}

__device__ __forceinline__ float fexpm1_neg(float x, float ex) {   // x <= 0, ex = fexp(x)
    const float d = ex - 1.f;
    const float k = d * x * frcp(flog(ex));
    return x < -17.f ? -1.f : (d == 0.f ? x : k);
}

__device__ __forceinline__ float digamma_pos(float x) {
    // x > 0: upward recurrence to x >= 6, then the asymptotic series (rare generic path)
    float r = 0.f;
    while (x < 6.f) { r -= 1.f / x; x += 1.f; }
    const float xi = 1.f / x, xi2 = xi * xi;
    return r + logf(x) - 0.5f * xi - xi2 * (1.f / 12.f - xi2 * (1.f / 120.f - xi2 * (1.f / 252.f)));
}

// lgamma / digamma route for non-integer or large counts: rare, kept out of line so the hot
// loop stays small (registers, instruction cache)
template <bool GRAD>
__device__ __attribute__((noinline)) float nb_t1_generic(float tp, float y, float* dpsi) {
    if (GRAD) *dpsi = digamma_pos(y + tp) - digamma_pos(tp);
    return lgammaf(tp) + lgammaf(y + 1.f) - lgammaf(y + tp);
}

// Counts above kSmallY (any real y > 16): t1 = lgamma(tp) + lgamma(y+1) - lgamma(y+tp) and
// dpsi = psi(y+tp) - psi(tp) from Stirling's series, with the differences of the large terms
// taken analytically (log1p forms) instead of subtracting three lgamma values of magnitude
// ~1e5.  tp < 8 is shifted up by 8 through the recurrence.  Branch-free.  Against fp64 on
// tp in [1e-4, 1e4], y in (16, 2000]: |err| <= 8e-4 absolute / 9e-6 of max(|t1|, 1); dpsi 4e-7.
template <bool GRAD>
__device__ __forceinline__ void nb_t1_large(float tp, float y, float& t1, float& dpsi) {
    const bool lo = tp < 8.f;
    const float sh = lo ? 8.f : 0.f;
    const float a = tp + sh, b = y + tp, d = y - sh, yp1 = y + 1.f;
    float prod = 1.f, ssum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = tp + (float)i;
        prod *= lo ? x : 1.f;
        if (GRAD) ssum += lo ? frcp(x) : 0.f;
    }
    const float ra = frcp(a), rb = frcp(b), ry = frcp(yp1);
    const float l1 = flog1p(d * ra);                    // log(b / a)
    const float l2 = -flog1p((tp - 1.f) * ry);           // log((y+1) / b)
    const float ra2 = ra * ra, rb2 = rb * rb, ry2 = ry * ry;
    const float Sa = ra * (1.f / 12.f - ra2 * (1.f / 360.f - ra2 * (1.f / 1260.f)));
    const float Sb = rb * (1.f / 12.f - rb2 * (1.f / 360.f - rb2 * (1.f / 1260.f)));
    const float Sy = ry * (1.f / 12.f - ry2 * (1.f / 360.f - ry2 * (1.f / 1260.f)));
    t1 = -(a - 0.5f) * l1 + d * l2 + (sh + 0.5f) * flog(yp1) + (-0.08106146679532726f - sh) +
         (Sa - Sb + Sy) - flog(prod);
    if (GRAD) {
        const float Pa = ra2 * (1.f / 12.f - ra2 * (1.f / 120.f - ra2 * (1.f / 252.f)));
        const float Pb = rb2 * (1.f / 12.f - rb2 * (1.f / 120.f - rb2 * (1.f / 252.f)));
        dpsi = l1 - 0.5f * (rb - ra) - Pb + Pa + ssum;
    }
}

// lgamma(y + 1) for the Poisson likelihood's constant term (dca/loss.py:52): table for integer
// counts <= 16, Stirling above (next omitted term 1/(1260 z^5) < 6e-10 at z = 17), libm otherwise.
__device__ __forceinline__ float lgamma_yp1(float y) {
    if (y > 16.f) {
        const float z = y + 1.f, r = frcp(z), r2 = r * r;
        return (y + 0.5f) * flog(z) - z + 0.91893853320467274178f + r * (1.f / 12.f - r2 * (1.f / 360.f));
    }
    if (y == floorf(y) && y >= 0.f) return kLogFact[(int)y];
    return lgammaf(y + 1.f);
}

// Poisson (dca/loss.py:36-55 with MeanAct, dca/network.py:233-246) and squared error
// (dca/loss.py:24-27 on the linear mean head, dca/network.py:143-156): loss of one element and
// d loss / d pre-activation (unscaled).
__device__ __forceinline__ float poisson_elem(float am, float sf, float y, float& d_am) {
    const float e = fexp(am);
    const bool win = (e >= 1e-5f) && (e <= 1e6f);
    const float mu = fminf(fmaxf(e, 1e-5f), 1e6f) * sf;
    const float mue = mu + kEps;
    d_am = win ? (1.f - y / mue) * e * sf : 0.f;
    return mu - y * flog(mue) + lgamma_yp1(y);
}

__device__ __forceinline__ float mse_elem(float am, float sf, float y, float& d_am) {
    const float diff = am * sf - y;
    d_am = 2.f * diff * sf;
    return diff * diff;
}

struct Heads {       // activations of one element
    float mu, gm;    // mean * sf,            d mu / d a_mean          (0 outside the clip window)
    float theta, gd; // dispersion,           d theta / d a_disp
    float pi, omp;   // dropout prob, 1 - pi (computed directly, no cancellation)
};

template <bool HAS_PI, bool CONST_DISP>
__device__ __forceinline__ Heads head_acts(float am, float ad, float ap, float sf) {
    Heads h;
    const float e = fexp(am);                                   // network.py:38
    const bool mwin = (e >= 1e-5f) && (e <= 1e6f);
    h.mu = fminf(fmaxf(e, 1e-5f), 1e6f) * sf;                   // layers.py:85
    h.gm = mwin ? e * sf : 0.f;
    if (CONST_DISP) {                                           // layers.py:21 (ad = theta_w[g])
        h.theta = fminf(fmaxf(fexp(ad), 1e-3f), 1e4f);
        h.gd = 1.f;                                             // chained in dcahip_colsum_chain
    } else {                                                    // network.py:39
        const float ex = fexp(-fabsf(ad));
        const float u = 1.f + ex, d = u - 1.f;
        const float s = frcp(u);
        const float l1 = d == 0.f ? ex : flog(u) * (ex * frcp(d));      // log1p(ex)
        const float sp = fmaxf(ad, 0.f) + l1;
        const bool dwin = (sp >= 1e-4f) && (sp <= 1e4f);
        h.theta = fminf(fmaxf(sp, 1e-4f), 1e4f);
        h.gd = dwin ? (ad >= 0.f ? s : ex * s) : 0.f;
    }
    h.theta = fminf(h.theta, kThetaMax);                        // loss.py:85
    if (HAS_PI) {
        const float ex = fexp(-fabsf(ap));
        const float s = frcp(1.f + ex);
        h.pi = ap >= 0.f ? s : ex * s;
        h.omp = ap >= 0.f ? ex * s : s;
    } else {
        h.pi = 0.f; h.omp = 1.f;
    }
    return h;
}

// ---- the y = 0 element of the ZINB likelihood, trimmed for the dense pass of K-HEADS (93 % of a count
// matrix): activations, zero_case (loss.py:136-137), ridge and the three PRE-ACTIVATION gradients in ~105
// VALU operations / 13 transcendentals instead of ~170 / 16 through head_acts + nll_elem:
//   * exp = v_exp_f32(x log2 e) bare: no clamp (inf / 0 propagate into the clip windows, nothing multiplies them
//     by 0) and no compensation of the rounded product (relative error <= |x| 1e-7, unbiased: averages out of the
//     loss sum and stays two orders below the gradient tolerance); clip windows tested as med3(x) == x;
//   * log1p(x) = log(u) + (x - (u - 1)) / u with u = fl(1 + x): the reciprocal is one the caller needs anyway
//     (sigmoid of the dispersion head; theta / (theta + mu) of the zero case), no Kahan ratio, no select;
//   * log2 -> ln by one multiplication where the result is not differenced against a neighbour;
//   * expm1(x <= 0) as a 3-term series above -1/64, exp(x) - 1 below (absolute error <= 1 ulp of 1, relative
//     <= 2e-6: d nll / d pi = -(1 - z) / D is O(1) there); 
// tools/zero_path_accuracy.py models it in numpy fp32 against the fp64 oracle: 400 000 random + edge elements,
// loss sum 3e-9 relative, every gradient inside the per-element tolerance of tests/test_kernels_gpu.py.
__device__ __forceinline__ float fexp_raw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float flog_fast(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }

struct ZAct { float mu, gm, theta, gd, pi, omp; };

template <bool CONST_DISP>
__device__ __forceinline__ ZAct zinb_acts(float am, float ad, float ap, float sf) {
    ZAct a;
    const float e = fexp_raw(am);                                             // network.py:38
    const float ec = __builtin_amdgcn_fmed3f(e, 1e-5f, 1e6f);
    a.mu = ec * sf;                                                           // layers.py:85
    a.gm = ec == e ? a.mu : 0.f;                                              // inside the clip window
    if (CONST_DISP) {                                                         // layers.py:21
        a.theta = __builtin_amdgcn_fmed3f(fexp_raw(ad), 1e-3f, 1e4f);
        a.gd = 1.f;
    } else {                                                                  // network.py:39
        const float ex = fexp_raw(-fabsf(ad));
        const float u = 1.f + ex;
        const float s = frcp(u);
        const float l1 = fmaf(ex - (u - 1.f), s, flog_fast(u));               // log1p(ex)
        const float sp = fmaxf(ad, 0.f) + l1;
        a.theta = __builtin_amdgcn_fmed3f(sp, 1e-4f, 1e4f);                   // <= 1e4 < kThetaMax
        a.gd = a.theta == sp ? (ad >= 0.f ? s : ex * s) : 0.f;
    }
    const float ex2 = fexp_raw(-fabsf(ap));
    const float s2 = frcp(1.f + ex2);
    const float es2 = ex2 * s2;
    a.pi = ap >= 0.f ? s2 : es2;
    a.omp = ap >= 0.f ? es2 : s2;
    return a;
}

// returns nll; gm / gd / gp = d nll / d (a_mean, a_disp, a_pi), unscaled
template <bool CONST_DISP>
__device__ __forceinline__ float zinb_zero_elem(float am, float ad, float ap, float sf, float ridge,
                                                float& g_m, float& g_d, float& g_p) {
    const ZAct a = zinb_acts<CONST_DISP>(am, ad, ap, sf);
    const float mu = a.mu, gm = a.gm, theta = a.theta, gd = a.gd, pi = a.pi, omp = a.omp;
    // zero_case = -log(pi + (1 - pi) (theta / (theta + mu + eps))^theta + eps)
    const float mue = mu + kEps;
    const float rden = frcp(theta + mue);
    const float t = mue * frcp(theta);                                        // theta / den = 1 / (1 + t)
    const float u2 = 1.f + t;
    const float logq = -fmaf(t - (u2 - 1.f), theta * rden, flog_fast(u2));
    const float tl = theta * logq;
    const float z = fexp_raw(tl);
    const float D = fmaf(omp, z, pi) + kEps;
    float nll = -flog_fast(D);
    const float invD = frcp(D);
    const float oz = omp * z * invD;
    const float dmu = oz * theta * rden;
    // log q + 1 - q = -log1p(t) + t / (1 + t): series below t = 2^-5 (cancellation)
    const float fs = -t * t * (0.5f - t * (2.f / 3.f - t * (0.75f - t * (0.8f - t * (5.f / 6.f)))));
    const float fl = fmaf(mue, rden, logq);
    const float dth = -oz * (t < 0.03125f ? fs : fl);
    // expm1(tl): 3-term series above -2^-6 (next term x^3 / 24 <= 1.6e-7 relative), z - 1 below (2e-6)
    const float ser = tl * fmaf(tl, fmaf(tl, 1.f / 6.f, 0.5f), 1.f);
    float dpi = (tl > -0.015625f ? ser : z - 1.f) * invD;                     // -(1 - z) / D
    dpi = fmaf(2.f * ridge, pi, dpi);                                         // loss.py:139-140
    nll = fmaf(ridge * pi, pi, nll);
    g_m = dmu * gm;
    g_d = dth * gd;
    g_p = dpi * pi * omp;
    return nll;
}

// ---- zinb_zero_elem cut into nine stages of about ten vector instructions each: the pipelined K-HEADS kernel
// (heads_p4.inc) issues one stage (for four elements) beside every chain of six matrix instructions.  Same operations in
// the same order as zinb_acts + zinb_zero_elem above (the comments there apply); the state between two stages is ZS.
struct ZS {
    float am, ad, ap, sf;                  // inputs: pre-activations, size factor
    float e, mu, gm;                       // mean head
    float ex, u, s, theta, gd;             // dispersion head
    float ex2, s2, pi, omp;                // dropout head
    float mue, rden, t, logq, tl, z, D, nll, invD, oz, dmu, dth, dpi;
    float g_m, g_d, g_p;                   // outputs: d nll / d (a_mean, a_disp, a_pi), unscaled; nll
};

template <bool CONST_DISP, int K>
__device__ __forceinline__ void zinb_zero_stage(ZS& q, float ridge) {
    if constexpr (K == 0) {
        q.e = fexp_raw(q.am);                                                     // network.py:38
        const float ec = __builtin_amdgcn_fmed3f(q.e, 1e-5f, 1e6f);
        q.mu = ec * q.sf;                                                         // layers.py:85
        q.gm = ec == q.e ? q.mu : 0.f;
        if (CONST_DISP) {                                                         // layers.py:21
            q.theta = __builtin_amdgcn_fmed3f(fexp_raw(q.ad), 1e-3f, 1e4f);
            q.gd = 1.f;
        } else {
            q.ex = fexp_raw(-fabsf(q.ad));                                        // network.py:39
        }
    } else if constexpr (K == 1) {
        if (!CONST_DISP) {
            q.u = 1.f + q.ex;
            q.s = frcp(q.u);
            q.theta = fmaf(q.ex - (q.u - 1.f), q.s, flog_fast(q.u));             // log1p(ex), finished in stage 2
        }
        q.ex2 = fexp_raw(-fabsf(q.ap));
    } else if constexpr (K == 2) {
        if (!CONST_DISP) {
            const float sp = fmaxf(q.ad, 0.f) + q.theta;
            q.theta = __builtin_amdgcn_fmed3f(sp, 1e-4f, 1e4f);                   // <= 1e4 < kThetaMax
            q.gd = q.theta == sp ? (q.ad >= 0.f ? q.s : q.ex * q.s) : 0.f;
        }
        q.s2 = frcp(1.f + q.ex2);
    } else if constexpr (K == 3) {
        const float es2 = q.ex2 * q.s2;
        q.pi = q.ap >= 0.f ? q.s2 : es2;
        q.omp = q.ap >= 0.f ? es2 : q.s2;
        q.mue = q.mu + kEps;                                                      // zero_case, loss.py:136-137
        q.rden = frcp(q.theta + q.mue);
        q.t = q.mue * frcp(q.theta);
    } else if constexpr (K == 4) {
        const float u2 = 1.f + q.t;
        q.logq = -fmaf(q.t - (u2 - 1.f), q.theta * q.rden, flog_fast(u2));
        q.tl = q.theta * q.logq;
        q.z = fexp_raw(q.tl);
    } else if constexpr (K == 5) {
        q.D = fmaf(q.omp, q.z, q.pi) + kEps;
        q.nll = -flog_fast(q.D);
        q.invD = frcp(q.D);
        q.oz = q.omp * q.z * q.invD;
        q.dmu = q.oz * q.theta * q.rden;
    } else if constexpr (K == 6) {
        const float t = q.t;
        const float fs = -t * t * (0.5f - t * (2.f / 3.f - t * (0.75f - t * (0.8f - t * (5.f / 6.f)))));
        const float fl = fmaf(q.mue, q.rden, q.logq);
        q.dth = -q.oz * (t < 0.03125f ? fs : fl);
    } else if constexpr (K == 7) {
        const float tl = q.tl;
        const float ser = tl * fmaf(tl, fmaf(tl, 1.f / 6.f, 0.5f), 1.f);
        float dpi = (tl > -0.015625f ? ser : q.z - 1.f) * q.invD;                 // -(1 - z) / D
        q.dpi = fmaf(2.f * ridge, q.pi, dpi);                                     // loss.py:139-140
        q.nll = fmaf(ridge * q.pi, q.pi, q.nll);
    } else {
        q.g_m = q.dmu * q.gm;
        q.g_d = q.dth * q.gd;
        q.g_p = q.dpi * q.pi * q.omp;
    }
}

// The values a stage hands to the later ones, redefined by an empty asm: the instruction selector orders pure arithmetic by
// register pressure and would sink every stage to the point where the element's results are stored; a pinned value has to
// exist WHERE the pin stands in program order, so stage K stays between the pins of stage K - 1 and its own.  No code.
#define ZPIN1(a) asm volatile("" : "+v"(a))
#define ZPIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
template <bool CONST_DISP, int K>
__device__ __forceinline__ void zinb_zero_pin(ZS& q) {
    if constexpr (K == 0) {
        if (CONST_DISP) { ZPIN4(q.mu, q.gm, q.ap, q.theta); ZPIN1(q.gd); }
        else { ZPIN4(q.mu, q.gm, q.ap, q.ex); ZPIN1(q.ad); }
    } else if constexpr (K == 1) {
        ZPIN4(q.mu, q.gm, q.ap, q.ex2);
        if (CONST_DISP) { ZPIN1(q.theta); ZPIN1(q.gd); } else { ZPIN4(q.ad, q.ex, q.s, q.theta); }
    } else if constexpr (K == 2) {
        ZPIN4(q.mu, q.gm, q.theta, q.gd);
        ZPIN1(q.ap); ZPIN1(q.ex2); ZPIN1(q.s2);
    } else if constexpr (K == 3) {
        ZPIN4(q.gm, q.theta, q.gd, q.pi);
        ZPIN4(q.omp, q.mue, q.rden, q.t);
    } else if constexpr (K == 4) {
        ZPIN4(q.gm, q.theta, q.gd, q.pi);
        ZPIN4(q.omp, q.mue, q.rden, q.t);
        ZPIN1(q.logq); ZPIN1(q.tl); ZPIN1(q.z);
    } else if constexpr (K == 5) {
        ZPIN4(q.gm, q.gd, q.pi, q.omp);
        ZPIN4(q.mue, q.rden, q.t, q.logq);
        ZPIN4(q.tl, q.z, q.nll, q.invD);
        ZPIN1(q.oz); ZPIN1(q.dmu);
    } else if constexpr (K == 6) {
        ZPIN4(q.gm, q.gd, q.pi, q.omp);
        ZPIN4(q.tl, q.z, q.nll, q.invD);
        ZPIN1(q.dmu); ZPIN1(q.dth);
    } else if constexpr (K == 7) {
        ZPIN4(q.gm, q.gd, q.pi, q.omp);
        ZPIN4(q.dmu, q.dth, q.dpi, q.nll);
    }
}

// The y = 0 element of the plain NB likelihood (loss.py:87-88 with y = 0: t1 = 0, t2 = theta log1p(mu / tp)).
template <bool CONST_DISP>
__device__ __forceinline__ float nb_zero_elem(float am, float ad, float sf, float& g_m, float& g_d) {
    const float e = fexp_raw(am);
    const float ec = __builtin_amdgcn_fmed3f(e, 1e-5f, 1e6f);
    const float mu = ec * sf;
    const float gm = ec == e ? mu : 0.f;
    float theta, gd;
    if (CONST_DISP) {
        theta = __builtin_amdgcn_fmed3f(fexp_raw(ad), 1e-3f, 1e4f);
        gd = 1.f;
    } else {
        const float ex = fexp_raw(-fabsf(ad));
        const float u = 1.f + ex;
        const float s = frcp(u);
        const float sp = fmaxf(ad, 0.f) + fmaf(ex - (u - 1.f), s, flog_fast(u));
        theta = __builtin_amdgcn_fmed3f(sp, 1e-4f, 1e4f);
        gd = theta == sp ? (ad >= 0.f ? s : ex * s) : 0.f;
    }
    const float tp = theta + kEps;
    const float rtm = frcp(tp + mu);
    const float x = mu * frcp(tp), u2 = 1.f + x;
    const float l1p = fmaf(x - (u2 - 1.f), tp * rtm, flog_fast(u2));        // log1p(mu / tp); 1 / u = tp / (tp + mu)
    g_m = theta * rtm * gm;                                                 // theta (mu + eps) / ((tp + mu)(mu + eps))
    g_d = (l1p - theta * x * rtm) * gd;
    return theta * l1p;
}

// The y > 0 element (nb_case, loss.py:87-88,130) with the same activations: compacted non-zero pass of K-HEADS.
// log(tp) - log(mu + eps) is taken as one log of the ratio (both reciprocals are needed by the gradient anyway),
// log1p(mu / tp) through the 1 / u = tp / (tp + mu) identity.
// INT_Y: the caller guarantees integer counts (the byte store of K-HEADS holds nothing else): the libm route for
// non-integer "counts" -- an out-of-line call with a stack slot -- is not compiled in.
template <bool CONST_DISP, bool INT_Y = false>
__device__ __forceinline__ float zinb_nz_elem(float am, float ad, float ap, float sf, float y, float ridge,
                                              float& g_m, float& g_d, float& g_p) {
    const ZAct a = zinb_acts<CONST_DISP>(am, ad, ap, sf);
    const float mu = a.mu, theta = a.theta, pi = a.pi, omp = a.omp;
    const float tp = theta + kEps, mue = mu + kEps;
    const float rtp = frcp(tp), rtm = frcp(tp + mu), rmue = frcp(mue);
    const float x = mu * rtp, u = 1.f + x;
    const float l1p = fmaf(x - (u - 1.f), tp * rtm, flog_fast(u));
    float t1, dpsi = 0.f;
    if (y > (float)kSmallY) {
        nb_t1_large<true>(tp, y, t1, dpsi);
    } else if (INT_Y || y == floorf(y)) {
        const int n = (int)y;
        float p1 = 1.f, p2 = 1.f;
        for (int i = 0; i < n; ++i) {
            const float xx = tp + (float)i;
            if (i < 8) p1 *= xx; else p2 *= xx;
            dpsi += frcp(xx);
        }
        t1 = kLogFact[n] - (flog(p1) + (n > 8 ? flog(p2) : 0.f));
    } else {
        t1 = nb_t1_generic<true>(tp, y, &dpsi);
    }
    const float ompe = omp + kEps;
    float nll = t1 + fmaf(theta + y, l1p, y * flog(tp * rmue)) - flog_fast(ompe);
    nll = fmaf(ridge * pi, pi, nll);
    const float dmu = theta * (mue - y) * rtm * rmue;
    const float dth = -dpsi + l1p + (y * tp - theta * mu) * rtp * rtm;
    const float dpi = fmaf(2.f * ridge, pi, frcp(ompe));
    g_m = dmu * a.gm;
    g_d = dth * a.gd;
    g_p = dpi * pi * omp;
    return nll;
}

// One element of the loss and (GRAD) its gradient w.r.t. (mu, theta, pi).
// ASSUME_NZ: the caller guarantees y >= kZeroThresh (compacted non-zero pass of K-HEADS).
template <bool HAS_PI, bool GRAD, bool ASSUME_NZ = false, bool INT_Y = false>
__device__ __forceinline__ float nll_elem(const Heads& h, float y, float ridge,
                                          float& dmu, float& dth, float& dpi) {
    const float theta = h.theta, mu = h.mu;
    const float tp = theta + kEps;
    float nll;
    if (HAS_PI && !ASSUME_NZ && y < kZeroThresh) {
        // zero_case = -log(pi + (1-pi) * (theta/(theta+mu+eps))^theta + eps)   loss.py:136-137
        const float den = theta + mu + kEps;
        const float rden = frcp(den);
        const float t = (mu + kEps) * frcp(theta);     // theta/den = 1/(1+t)
        const float logq = -flog1p(t);
        const float tl = theta * logq;
        const float z = fexp(tl);
        const float D = h.pi + h.omp * z + kEps;
        nll = -flog(D);
        if (GRAD) {
            const float invD = frcp(D);
            const float oz = h.omp * z * invD;
            dmu = oz * theta * rden;
            // log q + 1 - q = -log1p(t) + t/(1+t): series below t = 2^-5 (cancellation)
            const float fs = -t * t * (0.5f - t * (2.f / 3.f - t * (0.75f - t * (0.8f - t * (5.f / 6.f)))));
            const float fl = logq + (mu + kEps) * rden;
            dth = -oz * (t < 0.03125f ? fs : fl);
            dpi = fexpm1_neg(tl, z) * invD;            // -(1 - z)/D
        }
    } else {
        // NB.loss: t1 + t2, loss.py:87-88
        const float rtp = frcp(tp);
        const float l1p = flog1p(mu * rtp);
        float t1, dpsi = 0.f;
        if (y > (float)kSmallY) {
            nb_t1_large<GRAD>(tp, y, t1, dpsi);
        } else if (INT_Y || y == floorf(y)) {
            // lgamma(y+tp) - lgamma(tp) = log prod_{i<y}(tp+i); psi difference = sum 1/(tp+i)
            const int n = (int)y;
            float p1 = 1.f, p2 = 1.f;
            for (int i = 0; i < n; ++i) {
                const float x = tp + (float)i;
                if (i < 8) p1 *= x; else p2 *= x;
                if (GRAD) dpsi += frcp(x);
            }
            t1 = kLogFact[n] - (flog(p1) + (n > 8 ? flog(p2) : 0.f));
        } else {
            t1 = nb_t1_generic<GRAD>(tp, y, &dpsi);    // non-integer "counts" (check_counts=False)
        }
        const float mue = mu + kEps;
        const float t2 = (theta + y) * l1p + y * (flog(tp) - flog(mue));
        nll = t1 + t2;
        if (HAS_PI) nll -= flog(h.omp + kEps);         // loss.py:130
        if (GRAD) {
            // (theta+y)/(tp+mu) - y/(mu+eps) and -(theta+y)mu/(tp(tp+mu)) + y/tp, combined over a
            // common denominator: identical algebra, no cancellation between O(1) terms.
            const float rtm = frcp(tp + mu);
            dmu = theta * (mue - y) * rtm * frcp(mue);
            dth = -dpsi + l1p + (y * tp - theta * mu) * rtp * rtm;
            dpi = HAS_PI ? frcp(h.omp + kEps) : 0.f;
        }
    }
    if (HAS_PI) {
        nll += ridge * h.pi * h.pi;                    // loss.py:139-140
        if (GRAD) dpi += 2.f * ridge * h.pi;
    }
    return nll;
}

}  // namespace
