// ofl_aux_kernels.hip -- gfx950 kernels either side of the warp / compose hot path:
//
//   * the BACKWARD passes of the two primitives (the reference is differentiable end to end: README.rst:7-10,
//     utils.py:1079-1080, 1167; its tests assert grad_fn, test_utils.py:500, 1113-1114):
//       ofl_warp_bwd_grad_f32   grad wrt source (a splat of the upstream gradient with the bilinear weights) and wrt flow
//       ofl_splat_grad_f32      grad wrt data, x and y of the inverse-bilinear splat (a gather)
//   * the sparse point sampler behind track_pts / Flow.track (utils.py:941-1042) and its backward pass;
//   * the masked min / max reduction behind Flow.get_padding (flow_class.py:1174-1224).
//
// All of it is HBM / L2-bound gather, scatter or reduction work (no dense contraction, no MFMA).  Forward arithmetic that
// decides positions or taps restates the reference's fp32 operation order (bit-exact tap selection); the gradients
// themselves are sums whose order differs from ATen's CPU loops, and are compared within a stated fp32 tolerance.
//
// C ABI: include/oflib_hip.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "oflib_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr float kZeroThr = 1e-3f;   // utils.py:23, :642
constexpr float kDenMin = 1e-3f;    // utils.py:1144

__device__ __forceinline__ float unnormalise(float p, float size_m1, float half_size_m1) {
    // normalise_coords (utils.py:462-465) followed by the grid sampler's align_corners un-normalise
    float g = p * 2.0f;
    g = g / size_m1;
    g = g - 1.0f;
    return (g + 1.0f) * half_size_m1;
}

// the four taps of one bilinear sample at (sx, sy): weights, validity, clamped (always addressable) offsets
struct Taps {
    float nw, ne, sw, se;        // weights (ATen: nw = s*e, ne = s*w, sw = n*e, se = n*w)
    float ww, e, nn, s;          // fractional parts: w = sx - floor, e = 1 - w, n = sy - floor, s = 1 - n
    bool k_nw, k_ne, k_sw, k_se; // tap inside the image
    int64_t o_nw, o_ne, o_sw, o_se;
};

__device__ __forceinline__ Taps make_taps(float sx, float sy, int w, int h) {
    Taps t;
    const float x_w = floorf(sx), y_n = floorf(sy);
    t.ww = sx - x_w; t.e = 1.0f - t.ww; t.nn = sy - y_n; t.s = 1.0f - t.nn;
    t.nw = t.s * t.e; t.ne = t.s * t.ww; t.sw = t.nn * t.e; t.se = t.nn * t.ww;
    const float x_e = x_w + 1.0f, y_s = y_n + 1.0f;
    const bool x0ok = (x_w > -1.0f) && (x_w < (float)w), x1ok = (x_e > -1.0f) && (x_e < (float)w);
    const bool y0ok = (y_n > -1.0f) && (y_n < (float)h), y1ok = (y_s > -1.0f) && (y_s < (float)h);
    const int ix0 = x0ok ? (int)x_w : 0, ix1 = x1ok ? (int)x_e : 0, iy0 = y0ok ? (int)y_n : 0, iy1 = y1ok ? (int)y_s : 0;
    t.o_nw = (int64_t)iy0 * w + ix0; t.o_ne = (int64_t)iy0 * w + ix1; t.o_sw = (int64_t)iy1 * w + ix0; t.o_se = (int64_t)iy1 * w + ix1;
    t.k_nw = x0ok && y0ok; t.k_ne = x1ok && y0ok; t.k_sw = x0ok && y1ok; t.k_se = x1ok && y1ok;
    return t;
}

// ------------------------------------------------------------------------------------------------
// backward pass of the backward warp  (ofl_warp_bwd_grad_f32)
//
//   forward:  dst[n,c] = a_sign * addend + g_sign * sum_taps w_tap(p) * src[n,c,tap],   p = (gx, gy) - flow_sign * flow[n]
//   grad_src[n,c,tap] += g_scale * w_tap * grad_out[n,c]                                 (float atomics: taps collide)
//   grad_flow[n]       = -flow_sign * d(sample)/dp, chained through normalise_coords as autograd does
//                        (gix * (W-1)/2, then / (W-1) * 2; ATen grid_sampler_2d_backward's gix / giy sums)
// One pixel per lane; own-pixel loads are coalesced, taps are L2 / L1 gathers.
// ------------------------------------------------------------------------------------------------
struct WarpGradParams {
    const float* flow; int64_t flow_bs; float flow_sign;
    const float* src; int64_t src_bs;
    const float* gout; float g_scale;
    float* gsrc; int64_t gsrc_bs;      // optional, zeroed by the caller
    float* gflow;                      // optional [N,2,H,W]
    int32_t n, c, h, w;
    float wm1, hm1, half_wm1, half_hm1;
};

__global__ __launch_bounds__(256) void warp_grad_kernel(const WarpGradParams p) {
    const int n = blockIdx.y;
    const int w = p.w, h = p.h;
    const int64_t hw = (int64_t)h * w;
    const float* __restrict__ fu = p.flow + n * p.flow_bs;
    const float* __restrict__ sb = p.src + n * p.src_bs;
    const float* __restrict__ gb = p.gout + (int64_t)n * p.c * hw;
    float* __restrict__ gs = p.gsrc ? p.gsrc + n * p.gsrc_bs : nullptr;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < hw; pix += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(pix / w), x = (int)(pix - (int64_t)y * w);
        const float u = fu[pix], v = fu[hw + pix];
        const float sx = unnormalise((float)x - p.flow_sign * u, p.wm1, p.half_wm1);
        const float sy = unnormalise((float)y - p.flow_sign * v, p.hm1, p.half_hm1);
        const Taps t = make_taps(sx, sy, w, h);
        float gix = 0.0f, giy = 0.0f;
        for (int c = 0; c < p.c; ++c) {
            const float g = p.g_scale * gb[(int64_t)c * hw + pix];
            if (gs) {
                float* __restrict__ gc = gs + (int64_t)c * hw;
                if (t.k_nw) atomicAdd(gc + t.o_nw, t.nw * g);
                if (t.k_ne) atomicAdd(gc + t.o_ne, t.ne * g);
                if (t.k_sw) atomicAdd(gc + t.o_sw, t.sw * g);
                if (t.k_se) atomicAdd(gc + t.o_se, t.se * g);
            }
            if (p.gflow) {
                const float* __restrict__ sp = sb + (int64_t)c * hw;
                const float v_nw = t.k_nw ? sp[t.o_nw] : 0.0f, v_ne = t.k_ne ? sp[t.o_ne] : 0.0f;
                const float v_sw = t.k_sw ? sp[t.o_sw] : 0.0f, v_se = t.k_se ? sp[t.o_se] : 0.0f;
                // ATen: gix -= nw_val * (iy_se - iy) * g; gix += ne_val * (iy_sw - iy) * g; gix -= sw_val * (iy - iy_ne) * g; gix += se_val * (iy - iy_nw) * g
                gix -= v_nw * t.s * g; gix += v_ne * t.s * g; gix -= v_sw * t.nn * g; gix += v_se * t.nn * g;
                giy -= v_nw * t.e * g; giy -= v_ne * t.ww * g; giy += v_sw * t.e * g; giy += v_se * t.ww * g;
            }
        }
        if (p.gflow) {
            float* __restrict__ gf = p.gflow + (int64_t)n * 2 * hw;
            // grad of the grid (x half_size), of normalise_coords (/ size_m1, * 2), of `grid - flow` (negation)
            gf[pix] = -p.flow_sign * (((gix * p.half_wm1) / p.wm1) * 2.0f);
            gf[hw + pix] = -p.flow_sign * (((giy * p.half_hm1) / p.hm1) * 2.0f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward pass of the forward splat  (ofl_splat_grad_f32)
//
//   forward (utils.py:1098-1144, 1185-1203):  D[p] = sum w_ik,  A_c[p] = sum w_ik * data_c[i],  out_c = A_c / max(D, 1e-3);
//   un-occlude fill: out_c[i] = data_c[i] where mask & zero-flow & D == 0.
//   backward: gA_c[p] = g_c[p] / max(D, 1e-3);  gD[p] = (D >= 1e-3) ? -sum_c g_c[p] * out_c[p] / max(D, 1e-3) : 0;  g = 0 at filled p
//             grad_data_c[i] = sum_k w_ik * gA_c[p_ik]  (+ g_c[i] where i was filled)
//             grad_w_ik      = gD[p_ik] + sum_c data_c[i] * gA_c[p_ik]
//             grad_x[i]      = sum_ky wy_ky * (-eq_x0 * grad_w[ky][0] + eq_x1 * grad_w[ky][1]),  grad_y alike
//   A gather per source pixel: no atomics.
// ------------------------------------------------------------------------------------------------
struct SplatGradParams {
    const float* flow; int64_t flow_bs; float flow_sign;
    const float* xs; const float* ys; int64_t xy_bs;
    const float* data; int64_t data_bs;
    const uint8_t* weight_mask; int64_t weight_mask_bs;
    int32_t occlude;
    const float* out; const float* density; const float* gout; const float* gden;   // gden: optional upstream gradient of the density output
    float* prep;                       // [N, H * W] 16-byte slots (gA_0, gA_1, gA_2, density term) per DESTINATION pixel (splat_grad_prep_kernel)
    float* gdata; float* gxy;          // optional [N,C,H,W], [N,2,H,W]
    int32_t n, c, h, w;
};

__device__ __forceinline__ bool is_zero_vec(float u, float v) {
    return (u < kZeroThr) && (u > -kZeroThr) && (v < kZeroThr) && (v > -kZeroThr);
}

constexpr int kMaxGradC = 3;   // channels handled per launch (the host loops over groups): with the density term they fill one 16-byte slot per destination pixel

// Pass 1, per DESTINATION pixel (coalesced): gA_c = g_c / max(D, 1e-3) (0 where the pixel was un-occlude-filled: its gradient
// never reached A, D) and the term every corner weight that lands here receives, gD = (D >= 1e-3 ? -sum_c g_c out_c / max(D,
// 1e-3) : 0) + grad_density.  Each destination pixel is a corner of ~4 source pixels: its 2 C divisions are done once here
// instead of four times in the gather below (B=16 1080p C=3: 1.7 -> 1.0 ms for the two passes).
__global__ __launch_bounds__(256) void splat_grad_prep_kernel(const SplatGradParams p) {
    const int n = blockIdx.y;
    const int C = p.c;
    const int64_t hw = (int64_t)p.h * p.w;
    const float* __restrict__ fu = p.flow ? p.flow + n * p.flow_bs : nullptr;
    const uint8_t* __restrict__ wmk = p.weight_mask ? p.weight_mask + n * p.weight_mask_bs : nullptr;
    const float* __restrict__ ob = p.out + (int64_t)n * C * hw;
    const float* __restrict__ den = p.density + (int64_t)n * hw;
    const float* __restrict__ gb = p.gout + (int64_t)n * C * hw;
    float4* __restrict__ pr = reinterpret_cast<float4*>(p.prep) + (int64_t)n * hw;
    for (int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pos < hw; pos += (int64_t)gridDim.x * blockDim.x) {
        const float d = den[pos];
        bool filled = false;
        if (!(d > 0.0f) && p.occlude && fu) {
            const bool wmp = wmk ? (wmk[pos] != 0) : true;
            filled = wmp && is_zero_vec(fu[pos], fu[hw + pos]);
        }
        const float dcl = d < kDenMin ? kDenMin : d;
        float gD = 0.0f, gA[kMaxGradC] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < kMaxGradC; ++c) {
            if (c < C) {
                const float g = filled ? 0.0f : gb[(int64_t)c * hw + pos];
                gA[c] = g / dcl;
                gD -= g * ob[(int64_t)c * hw + pos] / dcl;
            }
        }
        float term = d >= kDenMin ? gD : 0.0f;               // clamp_min passes the gradient where D >= 1e-3
        if (p.gden) term += p.gden[(int64_t)n * hw + pos];
        pr[pos] = make_float4(gA[0], gA[1], gA[2], term);
    }
}

__global__ __launch_bounds__(256) void splat_grad_kernel(const SplatGradParams p) {
    const int n = blockIdx.y;
    const int w = p.w, h = p.h, C = p.c;
    const int64_t hw = (int64_t)h * w;
    const float wmax = (float)(w - 1), hmax = (float)(h - 1);
    const float* __restrict__ fu = p.flow ? p.flow + n * p.flow_bs : nullptr;
    const float* __restrict__ db = p.data + n * p.data_bs;
    const uint8_t* __restrict__ wmk = p.weight_mask ? p.weight_mask + n * p.weight_mask_bs : nullptr;
    const float* __restrict__ den = p.density + (int64_t)n * hw;
    const float* __restrict__ gb = p.gout + (int64_t)n * C * hw;
    const float4* __restrict__ pr = reinterpret_cast<const float4*>(p.prep) + (int64_t)n * hw;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < hw; pix += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(pix / w), x = (int)(pix - (int64_t)y * w);
        float xv, yv;
        bool zero = false;
        if (fu) {
            const float u = fu[pix], v = fu[hw + pix];
            xv = p.flow_sign * u + (float)x;
            yv = p.flow_sign * v + (float)y;
            if (p.occlude) zero = is_zero_vec(u, v);
        } else {
            xv = p.xs[n * p.xy_bs + pix];
            yv = p.ys[n * p.xy_bs + pix];
        }
        const bool wm = wmk ? (wmk[pix] != 0) : true;
        float gd[kMaxGradC];
#pragma unroll
        for (int c = 0; c < kMaxGradC; ++c) gd[c] = 0.0f;
        float gx = 0.0f, gy = 0.0f;
        if (wm && !zero) {
            float dat[kMaxGradC];
#pragma unroll
            for (int c = 0; c < kMaxGradC; ++c) dat[c] = c < C ? db[(int64_t)c * hw + pix] : 0.0f;
            const float x0 = floorf(xv), y0 = floorf(yv), x1 = x0 + 1.0f, y1 = y0 + 1.0f;
            const float x0s = fminf(fmaxf(x0, 0.0f), wmax), x1s = fminf(fmaxf(x1, 0.0f), wmax);
            const float y0s = fminf(fmaxf(y0, 0.0f), hmax), y1s = fminf(fmaxf(y1, 0.0f), hmax);
            const float eqx[2] = {x0 == x0s ? 1.0f : 0.0f, x1 == x1s ? 1.0f : 0.0f};
            const float eqy[2] = {y0 == y0s ? 1.0f : 0.0f, y1 == y1s ? 1.0f : 0.0f};
            const float wx[2] = {(x1 - xv) * eqx[0], (xv - x0) * eqx[1]};
            const float wy[2] = {(y1 - yv) * eqy[0], (yv - y0) * eqy[1]};
            const int ixs[2] = {(int)x0s, (int)x1s}, iys[2] = {(int)y0s, (int)y1s};
            float gwx[2] = {0.0f, 0.0f}, gwy[2] = {0.0f, 0.0f};
#pragma unroll
            for (int ky = 0; ky < 2; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) {
                    const int64_t pos = (int64_t)iys[ky] * w + ixs[kx];
                    float gw = 0.0f;
                    const float wgt = wy[ky] * wx[kx];
                    const float4 slot = pr[pos];                              // pass 1's (g_c / max(D, 1e-3), density term) of that destination pixel: ONE load
                    const float gAs[kMaxGradC] = {slot.x, slot.y, slot.z};
#pragma unroll
                    for (int c = 0; c < kMaxGradC; ++c) {
                        gd[c] += wgt * gAs[c];                                // (channels beyond C are zero in the slot)
                        gw += dat[c] * gAs[c];
                    }
                    gw += slot.w;                                             // through the density, and the density output's own gradient
                    gwx[kx] += wy[ky] * gw;
                    gwy[ky] += wx[kx] * gw;
                }
            }
            gx = eqx[1] * gwx[1] - eqx[0] * gwx[0];
            gy = eqy[1] * gwy[1] - eqy[0] * gwy[0];
        } else if (p.occlude && wm && zero && !(den[pix] > 0.0f)) {
#pragma unroll
            for (int c = 0; c < kMaxGradC; ++c)
                if (c < C) gd[c] = gb[(int64_t)c * hw + pix];       // filled from the data itself (utils.py:1203)
        }
        if (p.gdata) {
#pragma unroll
            for (int c = 0; c < kMaxGradC; ++c)
                if (c < C) p.gdata[((int64_t)n * C + c) * hw + pix] = gd[c];
        }
        if (p.gxy) {
            p.gxy[(int64_t)n * 2 * hw + pix] = gx;
            p.gxy[(int64_t)n * 2 * hw + hw + pix] = gy;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sparse point sampler (ofl_sample_pts_f32) and its backward pass: track_pts, utils.py:1004-1014
//   pts (y, x) -> flip -> normalise_coords -> grid_sample(flow, align_corners=True) -> flip -> + pts; NaN rows -> 0 (:1033-1035)
// ------------------------------------------------------------------------------------------------
struct PtsParams {
    const float* flow; int64_t flow_bs;
    const float* pts; int64_t pts_bs;      // [*, M, 2] (y, x)
    float* out;                            // forward: [N, M, 2]
    const float* gout; float* gflow; float* gpts;   // backward: gflow [N,2,H,W] zeroed by the caller (atomics), gpts [N, M, 2]
    int32_t n, m, h, w;
    float wm1, hm1, half_wm1, half_hm1;
};

template <bool BACKWARD>
__global__ __launch_bounds__(256) void sample_pts_kernel(const PtsParams p) {
    const int n = blockIdx.y;
    const int w = p.w, h = p.h;
    const int64_t hw = (int64_t)h * w;
    const float* __restrict__ fu = p.flow + n * p.flow_bs;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.m; i += gridDim.x * blockDim.x) {
        const float py = p.pts[n * p.pts_bs + 2 * i], px = p.pts[n * p.pts_bs + 2 * i + 1];
        const float sx = unnormalise(px, p.wm1, p.half_wm1), sy = unnormalise(py, p.hm1, p.half_hm1);
        const Taps t = make_taps(sx, sy, w, h);
        float val[2];
        float tv[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* __restrict__ sp = fu + (int64_t)c * hw;
            tv[c][0] = t.k_nw ? sp[t.o_nw] : 0.0f; tv[c][1] = t.k_ne ? sp[t.o_ne] : 0.0f;
            tv[c][2] = t.k_sw ? sp[t.o_sw] : 0.0f; tv[c][3] = t.k_se ? sp[t.o_se] : 0.0f;
            float r = tv[c][0] * t.nw;
            r = __builtin_fmaf(tv[c][1], t.ne, r);
            r = __builtin_fmaf(tv[c][2], t.sw, r);
            r = __builtin_fmaf(tv[c][3], t.se, r);
            val[c] = r;
        }
        const float oy = py + val[1], ox = px + val[0];
        const bool bad = (oy != oy) || (ox != ox);
        if (!BACKWARD) {
            p.out[((int64_t)n * p.m + i) * 2] = bad ? 0.0f : oy;
            p.out[((int64_t)n * p.m + i) * 2 + 1] = bad ? 0.0f : ox;
        } else {
            float gyo = p.gout[((int64_t)n * p.m + i) * 2], gxo = p.gout[((int64_t)n * p.m + i) * 2 + 1];
            if (bad) { gyo = 0.0f; gxo = 0.0f; }                 // the row was overwritten with zeros
            if (p.gflow) {
                float* __restrict__ g0 = p.gflow + (int64_t)n * 2 * hw;       // u plane feeds out_x, v plane out_y
                float* __restrict__ g1 = g0 + hw;
                if (t.k_nw) { atomicAdd(g0 + t.o_nw, t.nw * gxo); atomicAdd(g1 + t.o_nw, t.nw * gyo); }
                if (t.k_ne) { atomicAdd(g0 + t.o_ne, t.ne * gxo); atomicAdd(g1 + t.o_ne, t.ne * gyo); }
                if (t.k_sw) { atomicAdd(g0 + t.o_sw, t.sw * gxo); atomicAdd(g1 + t.o_sw, t.sw * gyo); }
                if (t.k_se) { atomicAdd(g0 + t.o_se, t.se * gxo); atomicAdd(g1 + t.o_se, t.se * gyo); }
            }
            if (p.gpts) {
                float gix = 0.0f, giy = 0.0f;
                const float gc[2] = {gxo, gyo};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    gix -= tv[c][0] * t.s * gc[c]; gix += tv[c][1] * t.s * gc[c]; gix -= tv[c][2] * t.nn * gc[c]; gix += tv[c][3] * t.nn * gc[c];
                    giy -= tv[c][0] * t.e * gc[c]; giy -= tv[c][1] * t.ww * gc[c]; giy += tv[c][2] * t.e * gc[c]; giy += tv[c][3] * t.ww * gc[c];
                }
                p.gpts[((int64_t)n * p.m + i) * 2] = gyo + ((giy * p.half_hm1) / p.hm1) * 2.0f;
                p.gpts[((int64_t)n * p.m + i) * 2 + 1] = gxo + ((gix * p.half_wm1) / p.wm1) * 2.0f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// masked extents of the positions a flow reaches (ofl_flow_extents_f32): Flow.get_padding, flow_class.py:1196-1219
//   pos = -(sign * thr(v) - grid)  per component, min / max over the pixels where mask is True
// Floats are reduced as order-preserving integers (wave DPP-free shuffles, then one atomic per wave and bound).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ord(float f) {
    const int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}

__global__ __launch_bounds__(256) void flow_extents_kernel(const float* __restrict__ flow, int64_t flow_bs,
                                                           const uint8_t* __restrict__ mask, int64_t mask_bs, float sign,
                                                           int32_t* __restrict__ ext, int32_t h, int32_t w) {
    const int n = blockIdx.y;
    const int64_t hw = (int64_t)h * w;
    const float* fu = flow + n * flow_bs;
    const uint8_t* mk = mask ? mask + n * mask_bs : nullptr;
    int lo_x = 0x7fffffff, hi_x = (int)0x80000000, lo_y = 0x7fffffff, hi_y = (int)0x80000000, any = 0;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < hw; pix += (int64_t)gridDim.x * blockDim.x) {
        if (mk && mk[pix] == 0) continue;
        const int y = (int)(pix / w), x = (int)(pix - (int64_t)y * w);
        float u = fu[pix], v = fu[hw + pix];
        if ((u < kZeroThr) && (u > -kZeroThr)) u = 0.0f;            // threshold_vectors, per component (utils.py:642)
        if ((v < kZeroThr) && (v > -kZeroThr)) v = 0.0f;
        const float px = -(sign * u - (float)x), py = -(sign * v - (float)y);
        lo_x = min(lo_x, ord(px)); hi_x = max(hi_x, ord(px)); lo_y = min(lo_y, ord(py)); hi_y = max(hi_y, ord(py));
        any = 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo_x = min(lo_x, __shfl_xor(lo_x, o)); hi_x = max(hi_x, __shfl_xor(hi_x, o));
        lo_y = min(lo_y, __shfl_xor(lo_y, o)); hi_y = max(hi_y, __shfl_xor(hi_y, o));
        any |= __shfl_xor(any, o);
    }
    if ((threadIdx.x & 63) == 0 && any) {
        int32_t* e = ext + n * 5;
        atomicMin(e + 0, lo_y); atomicMax(e + 1, hi_y); atomicMin(e + 2, lo_x); atomicMax(e + 3, hi_x);
        atomicOr(e + 4, 1);
    }
}

__global__ void flow_extents_init_kernel(int32_t* ext, int32_t n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { ext[5 * i] = 0x7fffffff; ext[5 * i + 1] = (int)0x80000000; ext[5 * i + 2] = 0x7fffffff; ext[5 * i + 3] = (int)0x80000000; ext[5 * i + 4] = 0; }
}

__global__ void flow_extents_decode_kernel(const int32_t* ext, float* out, int32_t n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * n) {
        const int b = ext[5 * (i >> 2) + (i & 3)];
        out[5 * (i >> 2) + (i & 3)] = __int_as_float(b >= 0 ? b : b ^ 0x7fffffff);
    } else if (i < 5 * n) {
        const int k = i - 4 * n;
        out[5 * k + 4] = (float)ext[5 * k + 4];
    }
}

inline int dims_ok(int32_t n, int32_t c, int32_t h, int32_t w) {
    if (n < 1 || c < 1 || h < 1 || w < 1) return OFL_E_SHAPE;
    if (n > 65535) return OFL_E_SHAPE;                      // blockIdx.y carries the batch index
    return OFL_OK;
}

inline unsigned blocks_for(int64_t items, int32_t n) {
    int64_t bx = (items + 255) / 256;
    int64_t cap = 8192 / (n < 1 ? 1 : n);
    cap = cap < 64 ? 64 : cap;
    if (bx > cap) bx = cap;
    return (unsigned)(bx < 1 ? 1 : bx);
}

}  // namespace

// ofl_kernels.hip: the gradient with respect to the flow on the forward's staged column kernel (OFL_E_UNSUPPORTED: not eligible)
int ofl_internal_warp_grad_flow_lds(const float* flow, int64_t flow_bs, float flow_sign, const float* src, int64_t src_bs,
                                    const float* grad_out, float g_scale, float* grad_flow, int32_t n, int32_t c, int32_t h, int32_t w,
                                    hipStream_t st);

// ------------------------------------------------------------------------------------------------
// valid area of a backward warp (ofl_warp_valid_f32): Flow.valid_target ('t') / valid_source ('s'), flow_class.py:1119-1122, 1151-1157
//   area = (grid_sample(ones, normalise(grid - sign * flow)) > thr) & mask
// The reference warps an all-ones image; every tap of it is 1 inside the frame and 0 (zero padding) outside, so the warped
// value is the FMA chain over the four in-frame indicators -- no image is made, read or written: 8 + 1 B/px in, 1 B/px out.
// VEC: 4 pixels per thread (16-byte loads; needs w % 4 == 0, so a group never straddles a row).
// ------------------------------------------------------------------------------------------------
struct WarpValidParams {
    const float* flow; int64_t flow_bs; float flow_sign;
    const uint8_t* mask; int64_t mask_bs;
    uint8_t* valid; float thr;
    int32_t h, w;
    float wm1, hm1, half_wm1, half_hm1;
};

__device__ __forceinline__ bool warp_valid_px(const WarpValidParams& p, float u, float v, int x, int y) {
    const float sx = unnormalise((float)x - p.flow_sign * u, p.wm1, p.half_wm1);     // grid - flow (utils.py:549)
    const float sy = unnormalise((float)y - p.flow_sign * v, p.hm1, p.half_hm1);
    const Taps t = make_taps(sx, sy, p.w, p.h);
    float r = (t.k_nw ? 1.0f : 0.0f) * t.nw;
    r = __builtin_fmaf(t.k_ne ? 1.0f : 0.0f, t.ne, r);
    r = __builtin_fmaf(t.k_sw ? 1.0f : 0.0f, t.sw, r);
    r = __builtin_fmaf(t.k_se ? 1.0f : 0.0f, t.se, r);
    return r > p.thr;
}

template <bool VEC>
__global__ __launch_bounds__(256) void warp_valid_kernel(const WarpValidParams p) {
    typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));
    const int n = blockIdx.y;
    const int64_t hw = (int64_t)p.h * p.w;
    const float* __restrict__ fu = p.flow + n * p.flow_bs;
    const uint8_t* __restrict__ mk = p.mask ? p.mask + n * p.mask_bs : nullptr;
    uint8_t* __restrict__ out = p.valid + n * hw;
    constexpr int K = VEC ? 4 : 1;
    for (int64_t pix = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * K; pix < hw; pix += (int64_t)gridDim.x * blockDim.x * K) {
        const int y = (int)(pix / p.w), x = (int)(pix - (int64_t)y * p.w);
        if (VEC) {
            const f4a u = *reinterpret_cast<const f4a*>(fu + pix), v = *reinterpret_cast<const f4a*>(fu + hw + pix);
            uint32_t m4 = 0x01010101u, o4 = 0u;
            if (mk) __builtin_memcpy(&m4, mk + pix, 4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o4 |= (uint32_t)(warp_valid_px(p, u[k], v[k], x + k, y) && ((m4 >> (8 * k)) & 0xffu) != 0u) << (8 * k);
            __builtin_memcpy(out + pix, &o4, 4);
        } else {
            out[pix] = (uint8_t)(warp_valid_px(p, fu[pix], fu[hw + pix], x, y) && (mk ? mk[pix] != 0 : true));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bilinear resize (ofl_resize_bilinear_f32): the F.interpolate(mode='bilinear', align_corners=False) of resize_flow (utils.py:908)
// and Flow.resize (flow_class.py:710) as ATen's CPU kernels compute it -- the reference's PyTorch-CPU result bit for bit, where
// ATen's own GPU kernel rounds its weights differently (1e-5 of the scale, rounds 1-3).  Source index: fmaf(scale, dst + 0.5,
// -0.5) clamped at 0; a dimension that keeps its size is copied; ATen picks one of TWO kernels from the output size
// (oh + ow <= 128: four pre-multiplied weights and one fma chain; else rows first, then columns) -- both restated here, probed
// against torch 2.10 CPU through the oracle (oracle/ofl_oracle.c: orc_resize_bilinear_f32, 18 million values, no mismatch).
// One output pixel per lane, rows of the output coalesced; a few MB once per call: nowhere near a hot path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, int32_t h,
                                                              int32_t w, int32_t oh, int32_t ow, float rh, float rw) {
    const int p = blockIdx.y;
    const float* __restrict__ sp = src + (int64_t)p * h * w;
    float* __restrict__ dp = dst + (int64_t)p * oh * ow;
    const bool small = (oh + ow) <= 128;
    const int64_t total = (int64_t)oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / ow), x = (int)(i - (int64_t)y * ow);
        float ry = __builtin_fmaf(rh, (float)y + 0.5f, -0.5f);
        ry = ry < 0.0f ? 0.0f : ry;
        int y0 = min((int)ry, h - 1), y1 = y0 + (y0 < h - 1 ? 1 : 0);
        float ly1 = fminf(fmaxf(ry - (float)y0, 0.0f), 1.0f);
        if (oh == h) { y0 = y1 = y; ly1 = 0.0f; }
        const float ly0 = 1.0f - ly1;
        float rx = __builtin_fmaf(rw, (float)x + 0.5f, -0.5f);
        rx = rx < 0.0f ? 0.0f : rx;
        int x0 = min((int)rx, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        float lx1 = fminf(fmaxf(rx - (float)x0, 0.0f), 1.0f);
        if (ow == w) { x0 = x1 = x; lx1 = 0.0f; }
        const float lx0 = 1.0f - lx1;
        const float s00 = sp[(int64_t)y0 * w + x0], s01 = sp[(int64_t)y0 * w + x1];
        const float s10 = sp[(int64_t)y1 * w + x0], s11 = sp[(int64_t)y1 * w + x1];
        float out;
        if (small) {
            const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
            out = __builtin_fmaf(s11, w11, __builtin_fmaf(s10, w10, __builtin_fmaf(s00, w00, s01 * w01)));
        } else {
            const float t0 = __builtin_fmaf(s00, lx0, s01 * lx1), t1 = __builtin_fmaf(s10, lx0, s11 * lx1);
            out = __builtin_fmaf(t0, ly0, t1 * ly1);
        }
        dp[i] = out;
    }
}

extern "C" {

__attribute__((visibility("default"))) int ofl_warp_bwd_grad_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* src, int64_t src_bs, const float* grad_out,
    float g_scale, float* grad_src, int64_t grad_src_bs, float* grad_flow, int32_t n, int32_t c, int32_t h, int32_t w,
    void* stream) {
    if (!flow || !src || !grad_out) return OFL_E_NULL;
    if (!grad_src && !grad_flow) return OFL_E_ARG;
    int rc = dims_ok(n, c, h, w);
    if (rc) return rc;
    if (!(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    if (grad_flow && !grad_src) {
        // the gradient with respect to the flow alone (the one with respect to the source is a gather splat, ofl_splat_sum_f32):
        // the forward kernel's staged boxes, one ds_read_b128 per tap instead of C scalar gathers
        rc = ofl_internal_warp_grad_flow_lds(flow, flow_bs, flow_sign, src, src_bs, grad_out, g_scale, grad_flow, n, c, h, w, (hipStream_t)stream);
        if (rc != OFL_E_UNSUPPORTED) return rc;
    }
    WarpGradParams p;
    p.flow = flow; p.flow_bs = flow_bs; p.flow_sign = flow_sign; p.src = src; p.src_bs = src_bs;
    p.gout = grad_out; p.g_scale = g_scale; p.gsrc = grad_src; p.gsrc_bs = grad_src_bs; p.gflow = grad_flow;
    p.n = n; p.c = c; p.h = h; p.w = w;
    p.wm1 = (float)(w - 1); p.hm1 = (float)(h - 1); p.half_wm1 = p.wm1 / 2.0f; p.half_hm1 = p.hm1 / 2.0f;
    hipLaunchKernelGGL(warp_grad_kernel, dim3(blocks_for((int64_t)h * w, n), (unsigned)n), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_splat_grad_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* xs, const float* ys, int64_t xy_bs,
    const float* data, int64_t data_bs, const uint8_t* weight_mask, int64_t weight_mask_bs, int32_t occlude,
    const float* out, const float* density, const float* grad_out, const float* grad_density, float* scratch,
    float* grad_data, float* grad_xy, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
    if (!data || !out || !density || !grad_out || !scratch) return OFL_E_NULL;
    if (!flow && !(xs && ys)) return OFL_E_NULL;
    if (!grad_data && !grad_xy) return OFL_E_ARG;
    if (flow && !(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    if (occlude && !flow) return OFL_E_ARG;
    int rc = dims_ok(n, c, h, w);
    if (rc) return rc;
    if (c > kMaxGradC) return OFL_E_UNSUPPORTED;           // (the caller splits wider data into channel groups)
    SplatGradParams p;
    p.flow = flow; p.flow_bs = flow_bs; p.flow_sign = flow_sign; p.xs = xs; p.ys = ys; p.xy_bs = xy_bs;
    p.data = data; p.data_bs = data_bs; p.weight_mask = weight_mask; p.weight_mask_bs = weight_mask_bs; p.occlude = occlude;
    p.out = out; p.density = density; p.gout = grad_out; p.gden = grad_density; p.gdata = grad_data; p.gxy = grad_xy;
    p.prep = scratch;
    p.n = n; p.c = c; p.h = h; p.w = w;
    hipLaunchKernelGGL(splat_grad_prep_kernel, dim3(blocks_for((int64_t)h * w, n), (unsigned)n), dim3(256), 0, (hipStream_t)stream, p);
    hipLaunchKernelGGL(splat_grad_kernel, dim3(blocks_for((int64_t)h * w, n), (unsigned)n), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

static int fill_pts(PtsParams& p, const float* flow, int64_t flow_bs, const float* pts, int64_t pts_bs, int32_t n,
                    int32_t m, int32_t h, int32_t w) {
    if (!flow || !pts) return OFL_E_NULL;
    int rc = dims_ok(n, 2, h, w);
    if (rc) return rc;
    if (m < 1) return OFL_E_SHAPE;
    p.flow = flow; p.flow_bs = flow_bs; p.pts = pts; p.pts_bs = pts_bs; p.n = n; p.m = m; p.h = h; p.w = w;
    p.wm1 = (float)(w - 1); p.hm1 = (float)(h - 1); p.half_wm1 = p.wm1 / 2.0f; p.half_hm1 = p.hm1 / 2.0f;
    p.out = nullptr; p.gout = nullptr; p.gflow = nullptr; p.gpts = nullptr;
    return OFL_OK;
}

__attribute__((visibility("default"))) int ofl_sample_pts_f32(const float* flow, int64_t flow_bs, const float* pts,
                                                              int64_t pts_bs, float* out, int32_t n, int32_t m,
                                                              int32_t h, int32_t w, void* stream) {
    PtsParams p;
    int rc = fill_pts(p, flow, flow_bs, pts, pts_bs, n, m, h, w);
    if (rc) return rc;
    if (!out) return OFL_E_NULL;
    p.out = out;
    hipLaunchKernelGGL(sample_pts_kernel<false>, dim3(blocks_for(m, n), (unsigned)n), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_sample_pts_grad_f32(const float* flow, int64_t flow_bs, const float* pts,
                                                                   int64_t pts_bs, const float* grad_out, float* grad_flow,
                                                                   float* grad_pts, int32_t n, int32_t m, int32_t h,
                                                                   int32_t w, void* stream) {
    PtsParams p;
    int rc = fill_pts(p, flow, flow_bs, pts, pts_bs, n, m, h, w);
    if (rc) return rc;
    if (!grad_out) return OFL_E_NULL;
    if (!grad_flow && !grad_pts) return OFL_E_ARG;
    p.gout = grad_out; p.gflow = grad_flow; p.gpts = grad_pts;
    hipLaunchKernelGGL(sample_pts_kernel<true>, dim3(blocks_for(m, n), (unsigned)n), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_resize_bilinear_f32(const float* src, float* dst, int32_t planes, int32_t h, int32_t w,
                                                                   int32_t oh, int32_t ow, float rcp_scale_h, float rcp_scale_w,
                                                                   void* stream) {
    if (!src || !dst) return OFL_E_NULL;
    if (planes < 1 || h < 1 || w < 1 || oh < 1 || ow < 1 || planes > 65535) return OFL_E_SHAPE;
    if ((int64_t)h * w >= (1ll << 31) || (int64_t)oh * ow >= (1ll << 31)) return OFL_E_SHAPE;
    if (!(rcp_scale_h > 0.0f) || !(rcp_scale_w > 0.0f)) return OFL_E_ARG;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks_for((int64_t)oh * ow, planes), (unsigned)planes), dim3(256), 0,
                       (hipStream_t)stream, src, dst, h, w, oh, ow, rcp_scale_h, rcp_scale_w);
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_warp_valid_f32(const float* flow, int64_t flow_bs, float flow_sign,
                                                              const uint8_t* mask, int64_t mask_bs, float thr, uint8_t* valid,
                                                              int32_t n, int32_t h, int32_t w, void* stream) {
    if (!flow || !valid) return OFL_E_NULL;
    int rc = dims_ok(n, 2, h, w);
    if (rc) return rc;
    if (!(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    WarpValidParams p;
    p.flow = flow; p.flow_bs = flow_bs; p.flow_sign = flow_sign; p.mask = mask; p.mask_bs = mask_bs; p.valid = valid; p.thr = thr;
    p.h = h; p.w = w;
    p.wm1 = (float)(w - 1); p.hm1 = (float)(h - 1); p.half_wm1 = p.wm1 / 2.0f; p.half_hm1 = p.hm1 / 2.0f;
    const int64_t hw = (int64_t)h * w;
    const bool vec = (w & 3) == 0 && (flow_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(flow) & 3) == 0;
    hipStream_t st = (hipStream_t)stream;
    if (vec) hipLaunchKernelGGL(warp_valid_kernel<true>, dim3(blocks_for(hw / 4, n), (unsigned)n), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(warp_valid_kernel<false>, dim3(blocks_for(hw, n), (unsigned)n), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_flow_extents_f32(const float* flow, int64_t flow_bs, const uint8_t* mask,
                                                                int64_t mask_bs, float sign, int32_t* workspace,
                                                                float* extents, int32_t n, int32_t h, int32_t w,
                                                                void* stream) {
    if (!flow || !workspace || !extents) return OFL_E_NULL;
    int rc = dims_ok(n, 2, h, w);
    if (rc) return rc;
    if (!(sign == 1.0f || sign == -1.0f)) return OFL_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(flow_extents_init_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, workspace, n);
    hipLaunchKernelGGL(flow_extents_kernel, dim3(blocks_for((int64_t)h * w, n), (unsigned)n), dim3(256), 0, st, flow, flow_bs,
                       mask, mask_bs, sign, workspace, h, w);
    hipLaunchKernelGGL(flow_extents_decode_kernel, dim3((unsigned)((5 * n + 63) / 64)), dim3(64), 0, st, workspace, extents, n);
    return (int)hipGetLastError();
}

// the flag words of a batch followed by their OR, one int per bit (ofl_flag_words_or_i32): one wave
// ------------------------------------------------------------------------------------------------
// flow of a homography (flow_from_matrix, utils.py:339-376): hom = M [x, y, 1]^T in the order of ATen's CPU batched
// matmul for 3 x 3 operands (its plain loop: acc = 0; acc += m[i][k] * v[k], k = 0, 1, 2; every product rounded), then
// hom.xy / hom.z (IEEE divide) minus (x, y); `sign` = -1 restates the reference's `-flow_from_matrix(...)` of its 't' branch
// (utils.py:699-705, 804-807: an exact negation).  Write-only, 8 B/px: 4 pixels per thread, 16-byte stores.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_from_matrix_kernel(const float* __restrict__ mats, int64_t mat_bs, float sign,
                                                               float* __restrict__ dst, int32_t h, int32_t w) {
    const int n = blockIdx.y;
    const float* __restrict__ m = mats + n * mat_bs;
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    const int64_t hw = (int64_t)h * w;
    float* __restrict__ du = dst + (int64_t)n * 2 * hw;
    const int wq = (w + 3) >> 2;                                   // 4-pixel groups per row (the last one may be partial)
    const int64_t groups = (int64_t)h * wq;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(g / wq), x0 = (int)(g - (int64_t)y * wq) * 4;
        const float fy = (float)y;
        float u[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float fx = (float)(x0 + k);
            float hx = 0.0f + m00 * fx; hx = hx + m01 * fy; hx = hx + m02 * 1.0f;
            float hy = 0.0f + m10 * fx; hy = hy + m11 * fy; hy = hy + m12 * 1.0f;
            float hz = 0.0f + m20 * fx; hz = hz + m21 * fy; hz = hz + m22 * 1.0f;
            u[k] = (hx / hz - fx) * sign;
            v[k] = (hy / hz - fy) * sign;
        }
        const int64_t pix = (int64_t)y * w + x0;
        if (x0 + 4 <= w && ((pix & 3) == 0) && ((hw & 3) == 0)) {
            *reinterpret_cast<float4*>(du + pix) = make_float4(u[0], u[1], u[2], u[3]);
            *reinterpret_cast<float4*>(du + hw + pix) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int k = 0; k < 4 && x0 + k < w; ++k) { du[pix + k] = u[k]; du[hw + pix + k] = v[k]; }
        }
    }
}

__global__ void flag_words_or_kernel(const int32_t* __restrict__ words, int32_t n, int32_t* __restrict__ out) {
    int f = 0;
    for (int i = threadIdx.x; i < n; i += 64) { const int v = words[i]; out[i] = v; f |= v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o);
    if (threadIdx.x < 5) out[n + threadIdx.x] = (f >> threadIdx.x) & 1;
}

__attribute__((visibility("default"))) int ofl_flag_words_or_i32(const int32_t* words, int32_t n, int32_t* out, void* stream) {
    if (!words || !out) return OFL_E_NULL;
    if (n < 1) return OFL_E_SHAPE;
    hipLaunchKernelGGL(flag_words_or_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, words, n, out);
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_flow_from_matrix_f32(const float* matrices, int64_t matrix_bs, float sign, float* dst,
                                                                    int32_t n, int32_t h, int32_t w, void* stream) {
    if (!matrices || !dst) return OFL_E_NULL;
    if (n < 1 || h < 1 || w < 1 || n > 65535 || (int64_t)h * w >= (1ll << 31)) return OFL_E_SHAPE;
    if (!(sign == 1.0f || sign == -1.0f)) return OFL_E_ARG;
    int64_t bx = ((int64_t)h * ((w + 3) / 4) + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(flow_from_matrix_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, matrices, matrix_bs, sign,
                       dst, h, w);
    return (int)hipGetLastError();
}

}  // extern "C"
