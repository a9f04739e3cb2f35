// ofl_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the dense flow warp / compose hot path.
//
// Everything here is HBM-bound gather / scatter / elementwise work: there is no dense contraction,
// so no MFMA.  What matters is (1) every input byte read ~once and every output byte written once,
// (2) 64-lane wavefronts reading / writing 256 contiguous bytes per instruction, (3) a
// workgroup -> tile order that keeps a tile's neighbours on the same XCD (each XCD has its own L2),
// (4) fp32 arithmetic in exactly the reference's operation order (bit-exact masks).
//
// Compiled with -ffp-contract=off -fno-fast-math; the only fused multiply-adds are the explicit
// __builtin_fmaf calls that restate the contraction in the reference's CPU grid-sampler.
//
// C ABI: include/oflib_hip.h (reference call sites cited there).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "oflib_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr float kValidThr = 0.99999f;  // flow_class.py:922
constexpr float kZeroThr = 1e-3f;      // utils.py:23, :642
constexpr float kDenMin = 1e-3f;       // utils.py:1144
constexpr int kXcds = 8;

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float unnormalise(float p, float size_m1, float half_size_m1) {
    // normalise_coords (utils.py:462-465) followed by the grid sampler's align_corners un-normalise
    float g = p * 2.0f;
    g = g / size_m1;  // IEEE correctly-rounded divide (no fast-math)
    g = g - 1.0f;
    return (g + 1.0f) * half_size_m1;
}

__device__ __forceinline__ float apply_round(float r, int mode) {
    if (mode != OFL_ROUND_NONE) {
        r = rintf(r);  // round-half-even == torch.round
        if (mode == OFL_ROUND_U8) r = fminf(fmaxf(r, 0.0f), 255.0f);
    }
    return r;
}

__device__ __forceinline__ int flag_bits(float u, float v, bool valid) {
    int f = 0;
    const bool nf = !(isfinite(u) && isfinite(v));
    const bool nz = !(u == 0.0f) || !(v == 0.0f);
    const bool nzt = !((u < kZeroThr) && (u > -kZeroThr)) || !((v < kZeroThr) && (v > -kZeroThr));
    if (nf) f |= OFL_FLAG_NONFINITE;
    if (nz) f |= OFL_FLAG_NZ | (valid ? OFL_FLAG_NZ_MASKED : 0);
    if (nzt) f |= OFL_FLAG_NZ_THR | (valid ? OFL_FLAG_NZ_THR_MASKED : 0);
    return f;
}

// OR-reduce a per-lane flag word over the 64-lane wavefront (5 ballots, no LDS)
__device__ __forceinline__ int wave_or_flags(int f) {
    int r = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b)
        if (__ballot((f >> b) & 1) != 0ull) r |= (1 << b);
    return r;
}

// XCD-aware block -> logical tile id: hardware deals blocks round-robin over the 8 XCDs, so block b
// and b+8 share an L2.  Give every XCD one contiguous range of logical tiles (speed only).
__device__ __forceinline__ int64_t logical_block(int64_t per_xcd) {
    const int64_t b = blockIdx.x;
    return (b % kXcds) * per_xcd + b / kXcds;
}

// ------------------------------------------------------------------------------------------------
// backward warp  (ofl_warp_bwd_f32)
// ------------------------------------------------------------------------------------------------
struct WarpParams {
    const float* flow; int64_t flow_bs;
    const float* src; int64_t src_bs;
    const uint8_t* src_mask; int64_t src_mask_bs;
    const uint8_t* flow_mask; int64_t flow_mask_bs;
    const float* addend; int64_t addend_bs;
    float* dst; uint8_t* valid;
    int32_t* flow_flags; int32_t* src_flags;
    int32_t n, c, h, w;
    float flow_sign, a_sign, g_sign;
    int32_t round_mode;
    float wm1, hm1, half_wm1, half_hm1;
    float rcp_wm1, rcp_hm1;          // RN(1/(w-1)), RN(1/(h-1)) for the exact reciprocal division
    int32_t tiles_x, tiles_y;
    int64_t total_tiles, per_xcd;
    uint32_t tiles_img, mx_m, mx_s, mi_m, mi_s;   // magic divisors by tiles_x and by tiles per image
    int32_t lds_bytes, shear;
};

constexpr int kTileW = 64;   // one wavefront spans 64 consecutive x: 256-byte rows per instruction
constexpr int kRows = 4;     // rows per thread (independent pixels in flight per lane)
constexpr int kTileH = 4 * kRows;  // 4 wavefronts per 256-thread block


// exact u32 division by an invariant divisor: q = (((n - t) >> s1) + t) >> s2 with t = umulhi(m, n); s = (s1 << 16) | s2
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t m, uint32_t s) {
    const uint32_t t = __umulhi(m, n);
    return (((n - t) >> (s >> 16)) + t) >> (s & 0xffffu);
}

// XCD-aware 32-bit tile decode (no 64-bit integer division in the kernel prologue)
__device__ __forceinline__ bool decode_tile(const WarpParams& p, int& tx, int& ty, int& n) {
    const uint32_t b = blockIdx.x;
    const uint32_t tile = (b & 7u) * (uint32_t)p.per_xcd + (b >> 3);
    if (tile >= (uint32_t)p.total_tiles) return false;
    const uint32_t nn = fastdiv(tile, p.mi_m, p.mi_s);
    const uint32_t rem = tile - nn * p.tiles_img;
    const uint32_t yy = fastdiv(rem, p.mx_m, p.mx_s);
    n = (int)nn; ty = (int)yy; tx = (int)(rem - yy * (uint32_t)p.tiles_x);
    return true;
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// backward warp, LDS-staged fast path (C <= 3, W % 4 == 0, 16-byte aligned planes)
//
// A 128-thread block owns TWO vertically adjacent 32 x 16 output tiles (A above B), 4 consecutive x per thread, and
// runs them as a straight-line software pipeline so that memory phases overlap compute:
//
//   flow(A), flow(B) loads  ->  coords + bbox(A)  ->  staging loads(A) issued  ->  coords + bbox(B) while they fly
//   -> LDS(A)  ->  staging loads(B) issued  ->  gather / blend / store A while they fly  ->  LDS(B)  ->  gather / store B
//
// Per tile:
//   1. flow (u, v) + flow mask: 16-byte loads; sample coordinates in the reference's fp32 op order (packed fp32;
//      the divide by (W-1) is an exact reciprocal division, see exact_div2);
//   2. bounding box of the source pixels the tile touches (DPP butterflies + one LDS exchange between the two waves);
//   3. the box is staged into LDS with 16-byte row-coalesced loads, channels + mask INTERLEAVED per pixel in
//      16-byte slots, de-interleaved by 4 along x:  slot(xl, yl) = 1 + yl*P + (xl & 3)*cw + (xl >> 2), so the
//      stride-4-pixel gathers of the 64 lanes are bank-conflict free; slot 0 holds zeros and every out-of-image
//      tap points there (zero padding without per-value selects);
//   4. one ds_read_b128 per tap fetches all channels; FMA chain in the reference's order; 16-byte stores.
// A tile whose box does not fit the LDS budget gathers straight from global memory (same arithmetic).
// Barriers order LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): global loads and stores stay in flight across them.
// ------------------------------------------------------------------------------------------------
constexpr int kLdsNT = 128, kLdsTWQ = 8, kLdsTH = kLdsNT / kLdsTWQ, kLdsIters = 3;
constexpr int kLdsBytes = 26624;   // 6 blocks (12 waves) per CU

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wave-wide min / max without LDS: four DPP butterfly steps inside each row of 16 lanes, then the four row results are
// combined on the scalar unit (v_readlane + s_min / s_max)
#define OFL_DPP(v, ctrl) __builtin_amdgcn_update_dpp((v), (v), (ctrl), 0xf, 0xf, false)
__device__ __forceinline__ int wave_min_dpp(int v) {
    v = min(v, OFL_DPP(v, 0xB1));    // quad_perm [1,0,3,2]
    v = min(v, OFL_DPP(v, 0x4E));    // quad_perm [2,3,0,1]
    v = min(v, OFL_DPP(v, 0x141));   // row_half_mirror
    v = min(v, OFL_DPP(v, 0x140));   // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_dpp(int v) {
    v = max(v, OFL_DPP(v, 0xB1));
    v = max(v, OFL_DPP(v, 0x4E));
    v = max(v, OFL_DPP(v, 0x141));
    v = max(v, OFL_DPP(v, 0x140));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ceil(2^20 / cw), cw = 1..127, without an integer division (float reciprocal + correction)
__device__ __forceinline__ uint32_t inv20(uint32_t cw) {
    uint32_t q = (uint32_t)(1048576.0f / (float)cw);
    while (q * cw > 1048576u) --q;
    while ((q + 1) * cw <= 1048576u) ++q;
    return q * cw == 1048576u ? q : q + 1;
}

// a / b for two values at once, bit-identical to the IEEE divide: y = RN(1/b), two Newton refinements through exact
// FMA residuals (Markstein); valid for a == 0 or 2^-60 <= |a| <= 2^100 -- the caller routes anything else (never
// seen in practice) through the hardware divide.
__device__ __forceinline__ f2 exact_div2(f2 a, float nb, float y) {
    const f2 yy = {y, y}, nbb = {nb, nb};
    f2 q = a * yy;
    f2 r = __builtin_elementwise_fma(nbb, q, a);
    q = __builtin_elementwise_fma(r, yy, q);
    r = __builtin_elementwise_fma(nbb, q, a);
    return __builtin_elementwise_fma(r, yy, q);
}

__device__ __forceinline__ int lds_pitch(int n) {   // smallest P >= n with P % 16 == 8 (read-conflict-free rows, TWQ = 8)
    int r = (kLdsTWQ - n) % (2 * kLdsTWQ);
    if (r < 0) r += 2 * kLdsTWQ;
    return n + r;
}

struct LdsCoords { float sx[4], sy[4]; };                              // un-normalised sample positions of 4 pixels
struct LdsBox { int bx0, miny, cw, Pp, bh, nch, sq, cbase; bool fits; };   // wave-uniform staging geometry

// Y-SHEARED box: a 32-wide tile under a flow with dv/dx != 0 touches a slanted band of source rows, and a plain bounding
// box wastes the two triangles above and below it.  Chunk column c (4 pixels) of the staged box therefore starts at image
// row miny + lds_shear(c, sq), with one block-uniform slope sq (rows per chunk column, Q8).  The box is taken in the
// sheared coordinate y' = y - lds_shear(x >> 2, sq): every slope is correct; a good one makes the box ~12 % smaller and
// boxes that overflow the LDS budget ~10x rarer (2.7 % -> 0.25 % of the tiles of the bench workload).
__device__ __forceinline__ int lds_shear(int c, int sq) { return __mul24(c, sq) >> 8; }

// slope estimate from the flow at the two ends of the row between the block's two tiles (scalar loads: uniform addresses)
__device__ __forceinline__ int lds_slope(const WarpParams& p, const float* __restrict__ fu, uint32_t hw, int tx, int ty2) {
    const int w = p.w, h = p.h;
    const int y = min(ty2 * (2 * kLdsTH) + kLdsTH, h - 1), xa = min(tx * (kLdsTWQ * 4), w - 1), xb = min(xa + kLdsTWQ * 4 - 1, w - 1);
    const float ul = fu[y * w + xa], ur = fu[y * w + xb], vl = fu[hw + y * w + xa], vr = fu[hw + y * w + xb];
    const float dx = (float)(xb - xa) - p.flow_sign * (ur - ul), dy = -p.flow_sign * (vr - vl);
    float q = 1024.0f * dy / dx;                                   // 256 * dy / (dx / 4)
    q = (dx > 4.0f) ? __builtin_amdgcn_fmed3f(q, -4096.0f, 4096.0f) : 0.0f;   // NaN, folds, degenerate spans: no shear
    return __builtin_amdgcn_readfirstlane((int)rintf(q));
}
template <int NC> struct LdsStage { int slot[kLdsIters]; f4 q[kLdsIters][NC]; uint32_t mq[kLdsIters]; };

// steps 1-2 for one tile
__device__ __forceinline__ void lds_coords_box(const WarpParams& p, int tx, int ty, const f4& u4, const f4& v4, int sq,
                                               LdsCoords& T, LdsBox& B, int (*red)[4]) {
    constexpr int NW = kLdsNT / 64;
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const int w = p.w, h = p.h;
    const int xc = min(tx * (kLdsTWQ * 4) + lx * 4, w - 4), yc = min(ty * kLdsTH + ly, h - 1);
    // ((x - s*u) * 2) / (w - 1) - 1, then (g + 1) * ((w - 1) / 2)   (utils.py:462-465, 549)
    const float xf = (float)xc, yf = (float)yc;
    f2 ax[2], ay[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f2 xx = {xf + (float)(2 * j), xf + (float)(2 * j + 1)}, yy = {yf, yf};
        const f2 su = (f2){u4[2 * j], u4[2 * j + 1]} * p.flow_sign, sv = (f2){v4[2 * j], v4[2 * j + 1]} * p.flow_sign;
        ax[j] = (xx - su) * 2.0f;
        ay[j] = (yy - sv) * 2.0f;
    }
    f2 qx[2], qy[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        qx[j] = exact_div2(ax[j], -p.wm1, p.rcp_wm1);
        qy[j] = exact_div2(ay[j], -p.hm1, p.rcp_hm1);
    }
    {
        const float big = fmaxf(fmaxf(fmaxf(fabsf(ax[0].x), fabsf(ax[0].y)), fmaxf(fabsf(ax[1].x), fabsf(ax[1].y))),
                                fmaxf(fmaxf(fabsf(ay[0].x), fabsf(ay[0].y)), fmaxf(fabsf(ay[1].x), fabsf(ay[1].y))));
        bool bad = !(big <= 0x1p100f);
        // a tiny non-zero operand needs |x - u| < 2^-61 with integer x >= 0: only column 0 / row 0 can produce one
        if (xc == 0) bad |= (fabsf(ax[0].x) < 0x1p-60f) && (ax[0].x != 0.0f);
        if (yc == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) bad |= (fabsf(ay[k >> 1][k & 1]) < 0x1p-60f) && (ay[k >> 1][k & 1] != 0.0f);
        }
        if (__builtin_expect(__any(bad), 0)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                qx[j] = (f2){ax[j].x / p.wm1, ax[j].y / p.wm1};
                qy[j] = (f2){ay[j].x / p.hm1, ay[j].y / p.hm1};
            }
        }
    }
    int minx = 0x7fffffff, maxx = -0x7fffffff, miny = 0x7fffffff, maxy = -0x7fffffff;
    const float wf = (float)w, hf = (float)h;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f2 sx = ((qx[j] - 1.0f) + 1.0f) * (f2){p.half_wm1, p.half_wm1};
        const f2 sy = ((qy[j] - 1.0f) + 1.0f) * (f2){p.half_hm1, p.half_hm1};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = 2 * j + i;
            T.sx[k] = sx[i]; T.sy[k] = sy[i];
            // west / north tap as an int, clamped to [-2, size] (beyond that every tap is out of range anyway; a NaN
            // coordinate lands on 0 through the conversion and is blended with NaN weights like the reference's)
            const int xi = (int)__builtin_amdgcn_fmed3f(floorf(sx[i]), -2.0f, wf);
            const int yi = (int)__builtin_amdgcn_fmed3f(floorf(sy[i]), -2.0f, hf);
            const int s0 = lds_shear(xi >> 2, sq), s1 = lds_shear((xi + 1) >> 2, sq);   // west / east tap columns
            minx = min(minx, xi); maxx = max(maxx, xi);
            miny = min(miny, yi - max(s0, s1)); maxy = max(maxy, yi + 1 - min(s0, s1));
        }
    }
    minx = wave_min_dpp(minx); maxx = wave_max_dpp(maxx); miny = wave_min_dpp(miny); maxy = wave_max_dpp(maxy);
    if (NW > 1) {
        if ((tid & 63) == 0) { red[tid >> 6][0] = minx; red[tid >> 6][1] = maxx; red[tid >> 6][2] = miny; red[tid >> 6][3] = maxy; }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            minx = min(minx, red[i][0]); maxx = max(maxx, red[i][1]);
            miny = min(miny, red[i][2]); maxy = max(maxy, red[i][3]);
        }
    }
    // touched columns [minx, maxx + 1] clipped to the image; sheared rows [miny, maxy] as they are (a staged row that falls
    // outside the image is skipped and never read back)   (wave-uniform)
    minx = max(__builtin_amdgcn_readfirstlane(minx), 0); maxx = min(__builtin_amdgcn_readfirstlane(maxx) + 1, w - 1);
    miny = __builtin_amdgcn_readfirstlane(miny); maxy = __builtin_amdgcn_readfirstlane(maxy);
    const bool empty = maxx < minx;
    B.bx0 = minx & ~3; B.miny = miny; B.sq = sq; B.cbase = B.bx0 >> 2;
    const int bw = empty ? 4 : (((maxx + 4) & ~3) - B.bx0);
    B.bh = empty ? 1 : (maxy - miny + 1); B.cw = bw >> 2; B.Pp = lds_pitch(bw); B.nch = B.bh * B.cw;
    B.fits = !empty && (B.bh <= 4096) && (16 * (1 + B.bh * B.Pp) <= p.lds_bytes) && (B.nch <= kLdsIters * kLdsNT);
}

// step 3a: issue the staging loads of a tile into registers (nothing waits here)
template <int NC, bool VALID>
__device__ __forceinline__ void lds_issue(const WarpParams& p, const float* __restrict__ sb, const uint8_t* __restrict__ sm,
                                          uint32_t hw, const LdsBox& B, LdsStage<NC>& S) {
    const int tid = threadIdx.x;
    const uint32_t inv = inv20((uint32_t)B.cw);
    const int rounds = B.fits ? (B.nch + kLdsNT - 1) / kLdsNT : 0;
#pragma unroll
    for (int it = 0; it < kLdsIters; ++it) {
        S.slot[it] = -1;
        if (it < rounds) {
            const uint32_t i = (uint32_t)tid + it * kLdsNT;
            // 24-bit multiplies (full rate): i < 2^9, inv <= 2^20, rows and columns < 2^13, h * w < 2^24
            const uint32_t r = __umul24(i, inv) >> 20, c4 = i - __umul24(r, (uint32_t)B.cw);
            const int y = B.miny + (int)r + lds_shear(B.cbase + (int)c4, B.sq);
            const bool on = (i < (uint32_t)B.nch) && ((uint32_t)y < (uint32_t)p.h);
            const uint32_t g = on ? (uint32_t)(__mul24(y, p.w) + B.bx0) + c4 * 4u : 0u;
            S.slot[it] = on ? 1 + (int)(__umul24(r, (uint32_t)B.Pp) + c4) : -1;
#pragma unroll
            for (int c = 0; c < NC; ++c) S.q[it][c] = *reinterpret_cast<const f4*>(sb + c * hw + g);
            S.mq[it] = (VALID && sm) ? *reinterpret_cast<const uint32_t*>(sm + g) : 0x01010101u;
        }
    }
}

// step 3b: registers -> interleaved LDS slots
template <int NC, bool VALID>
__device__ __forceinline__ void lds_write(f4* lds, const LdsBox& B, const LdsStage<NC>& S) {
    if (threadIdx.x == 0) lds[0] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < kLdsIters; ++it) {
        if (S.slot[it] >= 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f4 sl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NC; ++c) sl[c] = S.q[it][c][k];
                if (VALID) sl[3] = (float)(((S.mq[it] >> (8 * k)) & 0xffu) != 0u);
                lds[S.slot[it] + k * B.cw] = sl;
            }
        }
    }
}

// step 4a: gather from LDS (or from global memory when the box did not fit) and blend; per pixel (c0, c1, c2, mask channel)
template <int NC, bool VALID>
__device__ __forceinline__ void lds_gather(const WarpParams& p, uint32_t hw,
                                           const float* __restrict__ sb, const uint8_t* __restrict__ sm,
                                           const LdsCoords& T, const LdsBox& B, const unsigned char* smem, f4 (&outv)[4]) {
    const int w = p.w, h = p.h;
    const int cw16 = B.cw * 16, P16 = B.Pp * 16;
    const float wf = (float)w, hf = (float)h;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float fx = floorf(T.sx[k]), fy = floorf(T.sy[k]);
        const float ww = T.sx[k] - fx, e = 1.0f - ww, nn = T.sy[k] - fy, s = 1.0f - nn;
        const float wg[4] = {s * e, s * ww, nn * e, nn * ww};
        const int xi = (int)__builtin_amdgcn_fmed3f(fx, -2.0f, wf), yi = (int)__builtin_amdgcn_fmed3f(fy, -2.0f, hf);
        // west column valid <=> 0 <= xi <= w-1 ; east <=> -1 <= xi <= w-2   (rows alike)
        const bool x0 = (uint32_t)xi < (uint32_t)w, x1 = (uint32_t)(xi + 1) < (uint32_t)w;
        const bool y0 = (uint32_t)yi < (uint32_t)h, y1 = (uint32_t)(yi + 1) < (uint32_t)h;
        const bool ok[4] = {x0 && y0, x1 && y0, x0 && y1, x1 && y1};
        f4 tv[4];
        if (B.fits) {
            const int xl0 = xi - B.bx0, xl1 = xl0 + 1;
            const int cp0 = __mul24(xl0 & 3, cw16) + ((xl0 & ~3) << 2), cp1 = __mul24(xl1 & 3, cw16) + ((xl1 & ~3) << 2);
            const int yr = yi - B.miny;   // row in the sheared box, per tap column
            const int r0 = 16 + __mul24(yr - lds_shear(xi >> 2, B.sq), P16), r1 = 16 + __mul24(yr - lds_shear((xi + 1) >> 2, B.sq), P16);
            const int si[4] = {ok[0] ? r0 + cp0 : 0, ok[1] ? r1 + cp1 : 0, ok[2] ? r0 + P16 + cp0 : 0, ok[3] ? r1 + P16 + cp1 : 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) tv[j] = *reinterpret_cast<const f4*>(smem + si[j]);
        } else {
            const int cx[4] = {xi, xi + 1, xi, xi + 1}, cy[4] = {yi, yi, yi + 1, yi + 1};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t og = ok[j] ? (uint32_t)(cy[j] * w + cx[j]) : 0u;
                f4 t = {0.f, 0.f, 0.f, 0.f};
                if (ok[j]) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) t[c] = sb[c * hw + og];
                    if (VALID) t[3] = sm ? (float)(sm[og] != 0) : 1.0f;
                }
                tv[j] = t;
            }
        }
        // v_nw*nw, then fma(v_ne, ne, .), fma(v_sw, sw, .), fma(v_se, se, .): the reference's contraction order
        f4 r = tv[0] * wg[0];
        r = __builtin_elementwise_fma(tv[1], (f4){wg[1], wg[1], wg[1], wg[1]}, r);
        r = __builtin_elementwise_fma(tv[2], (f4){wg[2], wg[2], wg[2], wg[2]}, r);
        r = __builtin_elementwise_fma(tv[3], (f4){wg[3], wg[3], wg[3], wg[3]}, r);
        outv[k] = r;
    }
}

// the fused addend of a tile (mode 3), loaded ahead of younger loads and stores: the wait for it must not cover them
template <int NC>
__device__ __forceinline__ void lds_load_addend(const WarpParams& p, int tx, int ty, int n, uint32_t hw, f4 (&a)[NC]) {
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const uint32_t pix = (uint32_t)(min(ty * kLdsTH + ly, p.h - 1) * p.w + min(tx * (kLdsTWQ * 4) + lx * 4, p.w - 4));
#pragma unroll
    for (int c = 0; c < NC; ++c) a[c] = *reinterpret_cast<const f4*>(p.addend + n * p.addend_bs + c * hw + pix);
}

// step 4b: valid mask, epilogue (a_sign * addend + g_sign * G, rounding), 16-byte stores
template <int NC, bool VALID, bool ADD>
__device__ __forceinline__ void lds_store(const WarpParams& p, int tx, int ty, int n, uint32_t hw, uint32_t fmask4,
                                          const f4 (&outv)[4], const f4 (&addend)[NC]) {
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const int w = p.w, h = p.h;
    const int x4 = tx * (kLdsTWQ * 4) + lx * 4, y = ty * kLdsTH + ly;
    const bool inb = (x4 < w) && (y < h);
    const uint32_t pix = (uint32_t)(min(y, h - 1) * w + min(x4, w - 4));
    if (inb) {
        if (VALID) {
            uint32_t vo = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                vo |= (uint32_t)((outv[k][3] > kValidThr) && (((fmask4 >> (8 * k)) & 0xffu) != 0u)) << (8 * k);
            *reinterpret_cast<uint32_t*>(p.valid + (int64_t)n * hw + pix) = vo;
        }
        float* __restrict__ db = p.dst + (int64_t)n * NC * hw;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f4 o = {outv[0][c], outv[1][c], outv[2][c], outv[3][c]};
            if (ADD) o = addend[c] * p.a_sign + o * p.g_sign;
            if (p.round_mode != OFL_ROUND_NONE) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = apply_round(o[k], p.round_mode);
            }
            *reinterpret_cast<f4*>(db + c * hw + pix) = o;
        }
    }
}

template <int NC, bool VALID, bool ADD>
__global__ __launch_bounds__(kLdsNT, 3) void warp_bwd_lds_kernel(const WarpParams p) {
    constexpr int NW = kLdsNT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int red[2][NW][4];
    int tx, ty2, n;                      // the grid counts tile PAIRS: tiles_y = ceil(h / (2 * kLdsTH))
    if (!decode_tile(p, tx, ty2, n)) return;
    const int tyA = 2 * ty2, tyB = tyA + 1;
    const bool haveB = tyB * kLdsTH < p.h;
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const int w = p.w, h = p.h;
    const uint32_t hw = (uint32_t)(h * w);
    const float* __restrict__ fu = p.flow + n * p.flow_bs;
    const float* __restrict__ sb = p.src + n * p.src_bs;
    const uint8_t* __restrict__ sm = p.src_mask ? p.src_mask + n * p.src_mask_bs : nullptr;
    const uint8_t* __restrict__ fm = p.flow_mask ? p.flow_mask + n * p.flow_mask_bs : nullptr;
    const int x4 = tx * (kLdsTWQ * 4) + lx * 4, xq = min(x4, w - 4);
    const uint32_t pixA = (uint32_t)(min(tyA * kLdsTH + ly, h - 1) * w + xq);
    const uint32_t pixB = (uint32_t)(min(tyB * kLdsTH + ly, h - 1) * w + xq);
    const f4 uA = *reinterpret_cast<const f4*>(fu + pixA), vA = *reinterpret_cast<const f4*>(fu + hw + pixA);
    const f4 uB = *reinterpret_cast<const f4*>(fu + pixB), vB = *reinterpret_cast<const f4*>(fu + hw + pixB);
    uint32_t fmA = 0x01010101u, fmB = 0x01010101u;
    if ((VALID || p.flow_flags) && fm) {
        fmA = *reinterpret_cast<const uint32_t*>(fm + pixA);
        fmB = *reinterpret_cast<const uint32_t*>(fm + pixB);
    }
    if (p.flow_flags) {   // wave-uniform: finiteness / zero tests of the flow operand as a by-product
        int f = 0;
        const bool inA = (x4 < w) && (tyA * kLdsTH + ly < h), inB = (x4 < w) && (tyB * kLdsTH + ly < h);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (inA) f |= flag_bits(uA[k], vA[k], ((fmA >> (8 * k)) & 0xffu) != 0u);
            if (inB) f |= flag_bits(uB[k], vB[k], ((fmB >> (8 * k)) & 0xffu) != 0u);
        }
        f = wave_or_flags(f);
        if ((tid & 63) == 0 && f) atomicOr(&p.flow_flags[n], f);
    }
    f4* lds = reinterpret_cast<f4*>(smem);
    LdsCoords TA, TB;
    LdsBox BA, BB;
    LdsStage<NC> S;
    const int sq = p.shear ? lds_slope(p, fu, hw, tx, ty2) : 0;
    lds_coords_box(p, tx, tyA, uA, vA, sq, TA, BA, red[0]);
    lds_issue<NC, VALID>(p, sb, sm, hw, BA, S);                 // staging loads of A fly ...
    lds_coords_box(p, tx, tyB, uB, vB, sq, TB, BB, red[1]);     // ... while B's coordinates are computed
    lds_write<NC, VALID>(lds, BA, S);
    lds_barrier();
    // the vmcnt queue is in order: the addend (an L2 hit when it is the flow itself) is fetched BEFORE B's staging loads /
    // A's stores, so that waiting for it never waits for them
    // (flows only: with three channels the extra registers would spill, and nothing on the host adds to an image)
    constexpr bool EARLY = ADD && NC <= 2;
    f4 outv[4], aA[NC], aB[NC];
    if (EARLY) lds_load_addend<NC>(p, tx, tyA, n, hw, aA);
    if (haveB) lds_issue<NC, VALID>(p, sb, sm, hw, BB, S);      // staging loads of B fly while A is gathered and stored
    lds_gather<NC, VALID>(p, hw, sb, sm, TA, BA, smem, outv);
    if (EARLY && haveB) lds_load_addend<NC>(p, tx, tyB, n, hw, aB);
    if (ADD && !EARLY) lds_load_addend<NC>(p, tx, tyA, n, hw, aA);
    lds_store<NC, VALID, ADD>(p, tx, tyA, n, hw, fmA, outv, aA);
    if (!haveB) return;
    lds_barrier();
    lds_write<NC, VALID>(lds, BB, S);
    lds_barrier();
    lds_gather<NC, VALID>(p, hw, sb, sm, TB, BB, smem, outv);
    if (ADD && !EARLY) lds_load_addend<NC>(p, tx, tyB, n, hw, aB);
    lds_store<NC, VALID, ADD>(p, tx, tyB, n, hw, fmB, outv, aB);
}

// CT = compile-time channel count (0: run-time p.c)
template <int CT, bool VALID, bool ADD, bool FLAGS>
__global__ __launch_bounds__(256) void warp_bwd_kernel(const WarpParams p) {
    int tx, ty, n;
    if (!decode_tile(p, tx, ty, n)) return;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = tx * kTileW + lane;
    const int w = p.w, h = p.h;
    const int64_t hw = (int64_t)h * w;
    const int C = CT ? CT : p.c;

    const float* __restrict__ fu = p.flow + n * p.flow_bs;
    const float* __restrict__ fv = fu + hw;
    const float* __restrict__ sb = p.src + n * p.src_bs;
    const uint8_t* __restrict__ sm = p.src_mask ? p.src_mask + n * p.src_mask_bs : nullptr;
    const uint8_t* __restrict__ fm = p.flow_mask ? p.flow_mask + n * p.flow_mask_bs : nullptr;
    const float* __restrict__ ab = ADD ? p.addend + n * p.addend_bs : nullptr;
    float* __restrict__ db = p.dst + (int64_t)n * C * hw;

    int fflags = 0, sflags = 0;

#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int y = ty * kTileH + wave * kRows + r;
        if (x >= w || y >= h) continue;
        const int64_t pix = (int64_t)y * w + x;
        const float u = fu[pix], v = fv[pix];
        bool fmv = true;
        if (VALID || FLAGS) fmv = fm ? (fm[pix] != 0) : true;
        if (FLAGS) {
            fflags |= flag_bits(u, v, fmv);
            if (p.src_flags) {
                const bool smv = sm ? (sm[pix] != 0) : true;
                sflags |= flag_bits(sb[pix], sb[hw + pix], smv);
            }
        }
        // sample position: grid - flow (utils.py:549), flow_sign = -1 restates Flow(-vecs)
        const float px = (float)x - p.flow_sign * u;
        const float py = (float)y - p.flow_sign * v;
        const float sx = unnormalise(px, p.wm1, p.half_wm1);
        const float sy = unnormalise(py, p.hm1, p.half_hm1);
        const float x_w = floorf(sx), y_n = floorf(sy);
        const float ww = sx - x_w, e = 1.0f - ww;
        const float nn = sy - y_n, s = 1.0f - nn;
        const float nw = s * e, ne = s * ww, sw = nn * e, se = nn * ww;
        const float x_e = x_w + 1.0f, y_s = y_n + 1.0f;
        const bool x0ok = (x_w > -1.0f) && (x_w < (float)w);
        const bool x1ok = (x_e > -1.0f) && (x_e < (float)w);
        const bool y0ok = (y_n > -1.0f) && (y_n < (float)h);
        const bool y1ok = (y_s > -1.0f) && (y_s < (float)h);
        // clamped integer taps (always addressable); out-of-range taps are zeroed by the selects
        const int ix0 = x0ok ? (int)x_w : 0, ix1 = x1ok ? (int)x_e : 0;
        const int iy0 = y0ok ? (int)y_n : 0, iy1 = y1ok ? (int)y_s : 0;
        const int64_t o_nw = (int64_t)iy0 * w + ix0, o_ne = (int64_t)iy0 * w + ix1;
        const int64_t o_sw = (int64_t)iy1 * w + ix0, o_se = (int64_t)iy1 * w + ix1;
        const bool k_nw = x0ok && y0ok, k_ne = x1ok && y0ok, k_sw = x0ok && y1ok, k_se = x1ok && y1ok;

        if (VALID) {
            float m_nw, m_ne, m_sw, m_se;
            if (sm) {
                m_nw = k_nw ? (float)(sm[o_nw] != 0) : 0.0f;
                m_ne = k_ne ? (float)(sm[o_ne] != 0) : 0.0f;
                m_sw = k_sw ? (float)(sm[o_sw] != 0) : 0.0f;
                m_se = k_se ? (float)(sm[o_se] != 0) : 0.0f;
            } else {
                m_nw = k_nw ? 1.0f : 0.0f; m_ne = k_ne ? 1.0f : 0.0f;
                m_sw = k_sw ? 1.0f : 0.0f; m_se = k_se ? 1.0f : 0.0f;
            }
            float mr = m_nw * nw;
            mr = __builtin_fmaf(m_ne, ne, mr);
            mr = __builtin_fmaf(m_sw, sw, mr);
            mr = __builtin_fmaf(m_se, se, mr);
            p.valid[(int64_t)n * hw + pix] = (uint8_t)((mr > kValidThr) && fmv);
        }

#pragma unroll
        for (int ch = 0; ch < (CT ? CT : 1); ++ch) {
            for (int cc = ch; cc < C; cc += (CT ? C : 1)) {
                const float* __restrict__ sp = sb + (int64_t)cc * hw;
                const float v_nw = k_nw ? sp[o_nw] : 0.0f;
                const float v_ne = k_ne ? sp[o_ne] : 0.0f;
                const float v_sw = k_sw ? sp[o_sw] : 0.0f;
                const float v_se = k_se ? sp[o_se] : 0.0f;
                float rr = v_nw * nw;
                rr = __builtin_fmaf(v_ne, ne, rr);
                rr = __builtin_fmaf(v_sw, sw, rr);
                rr = __builtin_fmaf(v_se, se, rr);
                if (ADD) rr = p.a_sign * ab[(int64_t)cc * hw + pix] + p.g_sign * rr;
                db[(int64_t)cc * hw + pix] = apply_round(rr, p.round_mode);
            }
        }
    }

    if (FLAGS) {
        fflags = wave_or_flags(fflags);
        if (lane == 0 && fflags) atomicOr(&p.flow_flags[n], fflags);
        if (p.src_flags) {
            sflags = wave_or_flags(sflags);
            if (lane == 0 && sflags) atomicOr(&p.src_flags[n], sflags);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward splat, pass 1 (ofl_splat_fwd_f32): global fp32 atomics into a zeroed accumulator
// ------------------------------------------------------------------------------------------------
struct SplatParams {
    const float* flow; int64_t flow_bs; float flow_sign;
    const float* xs; const float* ys; int64_t xy_bs;
    const float* data; int64_t data_bs; float data_sign;
    const uint8_t* weight_mask; int64_t weight_mask_bs;
    const uint8_t* chan_mask_a; int64_t chan_mask_a_bs;
    const uint8_t* chan_mask_b; int64_t chan_mask_b_bs;
    int32_t with_mask_chan, occlude;
    float* accum;          // pass 1 out / pass 2 in
    float* dst; float* density; uint8_t* warped; uint8_t* valid; float* mask_chan;   // pass 2 out
    int32_t n, c, h, w;
    int32_t round_mode;
    int32_t tiles_x, tiles_y;
    int64_t total_tiles, per_xcd;
    const int32_t* run_if_set;   // optional device flag: the atomics path runs only when *run_if_set != 0
};

template <int CT>
__global__ __launch_bounds__(256) void splat_fwd_kernel(const SplatParams p) {
    if (p.run_if_set && *p.run_if_set == 0) return;
    const int64_t tile = logical_block(p.per_xcd);
    if (tile >= p.total_tiles) return;
    const int tx = (int)(tile % p.tiles_x);
    const int ty = (int)((tile / p.tiles_x) % p.tiles_y);
    const int n = (int)(tile / ((int64_t)p.tiles_x * p.tiles_y));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = tx * kTileW + lane;
    const int w = p.w, h = p.h;
    const int64_t hw = (int64_t)h * w;
    const int C = CT ? CT : p.c;
    const int planes = 1 + C + (p.with_mask_chan ? 1 : 0);
    const float wmax = (float)(w - 1), hmax = (float)(h - 1);

    const float* __restrict__ fu = p.flow ? p.flow + n * p.flow_bs : nullptr;
    const float* __restrict__ db = p.data + n * p.data_bs;
    const uint8_t* __restrict__ wmk = p.weight_mask ? p.weight_mask + n * p.weight_mask_bs : nullptr;
    const uint8_t* __restrict__ cma = p.chan_mask_a ? p.chan_mask_a + n * p.chan_mask_a_bs : nullptr;
    const uint8_t* __restrict__ cmb = p.chan_mask_b ? p.chan_mask_b + n * p.chan_mask_b_bs : nullptr;
    float* __restrict__ acc = p.accum + (int64_t)n * planes * hw;

#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int y = ty * kTileH + wave * kRows + r;
        if (x >= w || y >= h) continue;
        const int64_t pix = (int64_t)y * w + x;
        float xv, yv;
        bool zero = false;
        if (fu) {
            const float u = fu[pix], v = fu[hw + pix];
            xv = p.flow_sign * u + (float)x;  // get_flow_endpoints utils.py:1056-1057
            yv = p.flow_sign * v + (float)y;
            if (p.occlude) zero = (u < kZeroThr) && (u > -kZeroThr) && (v < kZeroThr) && (v > -kZeroThr);
        } else {
            xv = p.xs[n * p.xy_bs + pix];
            yv = p.ys[n * p.xy_bs + pix];
        }
        const bool wm = wmk ? (wmk[pix] != 0) : true;
        if (!wm || zero) continue;  // weight * 0: contributes exactly nothing (utils.py:1123)

        const float x0 = floorf(xv), y0 = floorf(yv);
        const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
        const float x0s = fminf(fmaxf(x0, 0.0f), wmax), x1s = fminf(fmaxf(x1, 0.0f), wmax);
        const float y0s = fminf(fmaxf(y0, 0.0f), hmax), y1s = fminf(fmaxf(y1, 0.0f), hmax);
        float wx[2], wy[2];
        wx[0] = (x1 - xv) * (x0 == x0s ? 1.0f : 0.0f);  // utils.py:1110
        wx[1] = (xv - x0) * (x1 == x1s ? 1.0f : 0.0f);
        wy[0] = (y1 - yv) * (y0 == y0s ? 1.0f : 0.0f);  // utils.py:1111
        wy[1] = (yv - y0) * (y1 == y1s ? 1.0f : 0.0f);
        const int ixs[2] = {(int)x0s, (int)x1s};
        const int iys[2] = {(int)y0s, (int)y1s};

        float mval = 0.0f;
        if (p.with_mask_chan) mval = ((cma ? cma[pix] != 0 : true) && (cmb ? cmb[pix] != 0 : true)) ? 1.0f : 0.0f;

#pragma unroll
        for (int ky = 0; ky < 2; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
                const float wgt = wy[ky] * wx[kx];  // utils.py:1114
                if (wgt == 0.0f) continue;          // adding +-0 never changes an accumulator that starts at +0
                const int64_t pos = (int64_t)iys[ky] * w + ixs[kx];  // utils.py:1118 (exact for h*w < 2^24)
                atomicAdd(&acc[pos], wgt);
#pragma unroll
                for (int ch = 0; ch < (CT ? CT : 1); ++ch)
                    for (int cc = ch; cc < C; cc += (CT ? C : 1))
                        atomicAdd(&acc[(int64_t)(1 + cc) * hw + pos], wgt * (p.data_sign * db[(int64_t)cc * hw + pix]));
                // mask channel: the reference accumulates wgt * mval next to the density.  All contributors of a
                // pixel being valid is the common case and must give ratio == 1 exactly, whatever order the
                // atomics land in -- so accumulate the INVALID weight instead and form den - inv in pass 2.
                if (p.with_mask_chan && mval == 0.0f) atomicAdd(&acc[(int64_t)(1 + C) * hw + pos], wgt);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward splat, pass 2 (ofl_splat_finalize_f32)
// ------------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256) void splat_finalize_kernel(const SplatParams p) {
    if (p.run_if_set && *p.run_if_set == 0) return;
    const int64_t tile = logical_block(p.per_xcd);
    if (tile >= p.total_tiles) return;
    const int tx = (int)(tile % p.tiles_x);
    const int ty = (int)((tile / p.tiles_x) % p.tiles_y);
    const int n = (int)(tile / ((int64_t)p.tiles_x * p.tiles_y));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = tx * kTileW + lane;
    const int w = p.w, h = p.h;
    const int64_t hw = (int64_t)h * w;
    const int C = CT ? CT : p.c;
    const int planes = 1 + C + (p.with_mask_chan ? 1 : 0);

    const float* __restrict__ fu = p.flow ? p.flow + n * p.flow_bs : nullptr;
    const float* __restrict__ db = p.data + n * p.data_bs;
    const uint8_t* __restrict__ wmk = p.weight_mask ? p.weight_mask + n * p.weight_mask_bs : nullptr;
    const uint8_t* __restrict__ cma = p.chan_mask_a ? p.chan_mask_a + n * p.chan_mask_a_bs : nullptr;
    const uint8_t* __restrict__ cmb = p.chan_mask_b ? p.chan_mask_b + n * p.chan_mask_b_bs : nullptr;
    const float* __restrict__ acc = p.accum + (int64_t)n * planes * hw;
    float* __restrict__ dst = p.dst + (int64_t)n * C * hw;

#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int y = ty * kTileH + wave * kRows + r;
        if (x >= w || y >= h) continue;
        const int64_t pix = (int64_t)y * w + x;
        const float den = acc[pix];
        const float dcl = den < kDenMin ? kDenMin : den;  // clamp_min utils.py:1144
        const bool warped = den > 0.0f;                    // utils.py:1197
        bool fill = false;
        if (p.occlude && fu && !warped) {                  // un-occlude utils.py:1198-1203
            const float u = fu[pix], v = fu[hw + pix];
            const bool zero = (u < kZeroThr) && (u > -kZeroThr) && (v < kZeroThr) && (v > -kZeroThr);
            const bool wm = wmk ? (wmk[pix] != 0) : true;
            fill = zero && wm;
        }
#pragma unroll
        for (int ch = 0; ch < (CT ? CT : 1); ++ch)
            for (int cc = ch; cc < C; cc += (CT ? C : 1)) {
                float val = fill ? p.data_sign * db[(int64_t)cc * hw + pix] : acc[(int64_t)(1 + cc) * hw + pix] / dcl;
                dst[(int64_t)cc * hw + pix] = apply_round(val, p.round_mode);
            }
        if (p.density) p.density[(int64_t)n * hw + pix] = den;
        if (p.warped) p.warped[(int64_t)n * hw + pix] = (uint8_t)warped;
        if (p.valid || p.mask_chan) {
            float mch;
            if (fill)
                mch = ((cma ? cma[pix] != 0 : true) && (cmb ? cmb[pix] != 0 : true)) ? 1.0f : 0.0f;
            else
                mch = (den - acc[(int64_t)(1 + C) * hw + pix]) / dcl;
            if (p.valid) p.valid[(int64_t)n * hw + pix] = (uint8_t)(mch > kValidThr);
            if (p.mask_chan) p.mask_chan[(int64_t)n * hw + pix] = mch;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// forward splat, tiled fast path (ofl_splat_tiled_f32): destination-tile-owned LDS accumulation, fused finalize
//
//  bin kernel : one block per 32 x 16 SOURCE tile: end points of its pixels -> bounding box of the destination pixels
//               it touches -> appends itself to the candidate list of every destination tile under that box
//               (fixed capacity; an overflow flags the launch for the atomics path).
//  tile kernel: one block per 32 x 16 DESTINATION tile: zeroes (2 + C) accumulator planes in LDS (10 KB), walks its
//               candidate source tiles (16-byte loads), adds the corner contributions that fall inside with LDS
//               float atomics (accumulator de-interleaved by 4 along x -> bank-conflict free), then normalises,
//               thresholds, un-occludes and stores with 16-byte stores.  No global atomics, no accumulator in HBM.
// Same arithmetic as the two-pass path (weights, clamps, zero-flow rule, invalid-weight mask channel).
// ------------------------------------------------------------------------------------------------
constexpr int kSpNT = 128, kSpTW = 32, kSpTH = 16, kSpMaxCand = 16;

__device__ __forceinline__ uint32_t nz_bytes(uint32_t x) {   // per byte: non-zero -> 0x01
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) r |= (uint32_t)(((x >> (8 * k)) & 0xffu) != 0u) << (8 * k);
    return r;
}

struct TiledParams {
    SplatParams s;
    int32_t* counts;       // [n * tiles]
    int32_t* lists;        // [n * tiles * kSpMaxCand]
    int32_t* overflow;     // [1]
    int32_t tiles_x, tiles_y;
    uint32_t tiles_img, mx_m, mx_s, mi_m, mi_s;
    int64_t total, per_xcd;
    int32_t binning;       // 1: atomic-free binning first (tiles that overflow it fall back to LDS float atomics)
};

__device__ __forceinline__ bool sp_decode(const TiledParams& p, int& tx, int& ty, int& n) {
    const uint32_t b = blockIdx.x;
    const uint32_t tile = (b & 7u) * (uint32_t)p.per_xcd + (b >> 3);
    if (tile >= (uint32_t)p.total) return false;
    const uint32_t nn = fastdiv(tile, p.mi_m, p.mi_s);
    const uint32_t rem = tile - nn * p.tiles_img;
    const uint32_t yy = fastdiv(rem, p.mx_m, p.mx_s);
    n = (int)nn; ty = (int)yy; tx = (int)(rem - yy * (uint32_t)p.tiles_x);
    return true;
}

// end point + contribution test of the 4 source pixels of this thread (shared by both kernels)
struct SpSrc { float x[4], y[4]; bool on[4]; bool zero[4]; bool wm[4]; };

__device__ __forceinline__ void sp_load_src(const SplatParams& s, int n, int sx4, int sy, bool inimg, uint32_t pix, uint32_t hw, SpSrc& q) {
    f4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    uint32_t wm4 = 0x01010101u;
    if (inimg) {
        if (s.flow) {
            a = *reinterpret_cast<const f4*>(s.flow + n * s.flow_bs + pix);
            b = *reinterpret_cast<const f4*>(s.flow + n * s.flow_bs + hw + pix);
        } else {
            a = *reinterpret_cast<const f4*>(s.xs + n * s.xy_bs + pix);
            b = *reinterpret_cast<const f4*>(s.ys + n * s.xy_bs + pix);
        }
        if (s.weight_mask) wm4 = *reinterpret_cast<const uint32_t*>(s.weight_mask + n * s.weight_mask_bs + pix);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        q.zero[k] = false;
        if (s.flow) {
            q.x[k] = s.flow_sign * a[k] + (float)(sx4 + k);      // get_flow_endpoints utils.py:1056-1057
            q.y[k] = s.flow_sign * b[k] + (float)sy;
            if (s.occlude) q.zero[k] = (a[k] < kZeroThr) && (a[k] > -kZeroThr) && (b[k] < kZeroThr) && (b[k] > -kZeroThr);
        } else {
            q.x[k] = a[k]; q.y[k] = b[k];
        }
        q.wm[k] = ((wm4 >> (8 * k)) & 0xffu) != 0u;
        q.on[k] = inimg && q.wm[k] && !q.zero[k];
    }
}

__global__ __launch_bounds__(kSpNT) void splat_bin_kernel(const TiledParams p) {
    __shared__ int red[kSpNT / 64][4];
    int tx, ty, n;
    if (!sp_decode(p, tx, ty, n)) return;
    const SplatParams& s = p.s;
    const int tid = threadIdx.x, lx = tid & 7, ly = tid >> 3;
    const int w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    const int sx4 = tx * kSpTW + lx * 4, sy = ty * kSpTH + ly;
    const bool inimg = (sx4 < w) && (sy < h);
    SpSrc q;
    sp_load_src(s, n, sx4, sy, inimg, (uint32_t)(sy * w + sx4), hw, q);
    int minx = 0x7fffffff, maxx = -0x7fffffff, miny = 0x7fffffff, maxy = -0x7fffffff;
    const float wf = (float)w, hf = (float)h;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (q.on[k]) {
            // destination columns floor(x), floor(x)+1 (clamped corners carry weight 0: the box is clipped below)
            const int x0 = (int)__builtin_amdgcn_fmed3f(floorf(q.x[k]), -2.0f, wf), y0 = (int)__builtin_amdgcn_fmed3f(floorf(q.y[k]), -2.0f, hf);
            minx = min(minx, x0); maxx = max(maxx, x0 + 1); miny = min(miny, y0); maxy = max(maxy, y0 + 1);
        }
    }
    minx = wave_min_dpp(minx); maxx = wave_max_dpp(maxx); miny = wave_min_dpp(miny); maxy = wave_max_dpp(maxy);
    if ((tid & 63) == 0) { red[tid >> 6][0] = minx; red[tid >> 6][1] = maxx; red[tid >> 6][2] = miny; red[tid >> 6][3] = maxy; }
    __syncthreads();
    if (tid != 0) return;
    for (int i = 0; i < kSpNT / 64; ++i) {
        minx = min(minx, red[i][0]); maxx = max(maxx, red[i][1]); miny = min(miny, red[i][2]); maxy = max(maxy, red[i][3]);
    }
    minx = max(minx, 0); maxx = min(maxx, w - 1); miny = max(miny, 0); maxy = min(maxy, h - 1);
    if (maxx < minx || maxy < miny) return;       // nothing of this tile lands inside the image
    const int tx0 = minx / kSpTW, tx1 = maxx / kSpTW, ty0 = miny / kSpTH, ty1 = maxy / kSpTH;
    if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > 64) { atomicOr(p.overflow, 1); return; }
    const int me = ty * p.tiles_x + tx;
    for (int dy = ty0; dy <= ty1; ++dy)
        for (int dx = tx0; dx <= tx1; ++dx) {
            const int64_t d = (int64_t)n * p.tiles_img + dy * p.tiles_x + dx;
            const int slot = atomicAdd(&p.counts[d], 1);
            if (slot < kSpMaxCand) p.lists[d * kSpMaxCand + slot] = me;
            else atomicOr(p.overflow, 1);
        }
}

constexpr int kSpQueue = 1536;     // source pixels that touch the destination tile (compacted)
constexpr int kSpBinCap = 12;      // contributions (source pixel, corner) one destination pixel can take on this path
constexpr int kSpQueue2 = 2048;    // queue of the local atomics path (flushed when full)


// accumulate queued source pixels with LDS float atomics (dense lanes) -- only for tiles the binning path cannot take
template <int NC, bool MCH>
__device__ __forceinline__ void sp_drain(const TiledParams& p, int n, int dx0, int dy0, const uint32_t* queue, int qlen, float* acc) {
    constexpr int kPx = kSpTW * kSpTH;
    const SplatParams& s = p.s;
    const int w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    const float wmax = (float)(w - 1), hmax = (float)(h - 1);
    const float* __restrict__ db = s.data + n * s.data_bs;
    for (int i = threadIdx.x; i < qlen; i += kSpNT) {
        const uint32_t rec = queue[i];
        const int sx = (int)(rec & 0xffffu), sy = (int)(rec >> 16);
        const uint32_t pix = (uint32_t)(sy * w + sx);
        float xv, yv;
        if (s.flow) {
            xv = s.flow_sign * s.flow[n * s.flow_bs + pix] + (float)sx;
            yv = s.flow_sign * s.flow[n * s.flow_bs + hw + pix] + (float)sy;
        } else {
            xv = s.xs[n * s.xy_bs + pix]; yv = s.ys[n * s.xy_bs + pix];
        }
        float dv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) dv[c] = s.data_sign * db[c * hw + pix];
        bool invalid = false;
        if (MCH) {
            const bool a = s.chan_mask_a ? s.chan_mask_a[n * s.chan_mask_a_bs + pix] != 0 : true;
            const bool b = s.chan_mask_b ? s.chan_mask_b[n * s.chan_mask_b_bs + pix] != 0 : true;
            invalid = !(a && b);
        }
        const float x0 = floorf(xv), y0 = floorf(yv), x1 = x0 + 1.0f, y1 = y0 + 1.0f;
        const float x0s = fminf(fmaxf(x0, 0.0f), wmax), x1s = fminf(fmaxf(x1, 0.0f), wmax);
        const float y0s = fminf(fmaxf(y0, 0.0f), hmax), y1s = fminf(fmaxf(y1, 0.0f), hmax);
        const float wx[2] = {(x1 - xv) * (x0 == x0s ? 1.0f : 0.0f), (xv - x0) * (x1 == x1s ? 1.0f : 0.0f)};
        const float wy[2] = {(y1 - yv) * (y0 == y0s ? 1.0f : 0.0f), (yv - y0) * (y1 == y1s ? 1.0f : 0.0f)};
        const int ix[2] = {(int)x0s - dx0, (int)x1s - dx0}, iy[2] = {(int)y0s - dy0, (int)y1s - dy0};
#pragma unroll
        for (int ky = 0; ky < 2; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
                const float wgt = wy[ky] * wx[kx];
                const int xl = ix[kx], yl = iy[ky];
                if (wgt == 0.0f || (uint32_t)xl >= (uint32_t)kSpTW || (uint32_t)yl >= (uint32_t)kSpTH) continue;
                const int idx = yl * kSpTW + (xl & 3) * (kSpTW / 4) + (xl >> 2);     // de-interleaved by 4: conflict-free finalize
                atomicAdd(&acc[idx], wgt);
#pragma unroll
                for (int c = 0; c < NC; ++c) atomicAdd(&acc[(1 + c) * kPx + idx], wgt * dv[c]);
                if (MCH && invalid) atomicAdd(&acc[(1 + NC) * kPx + idx], wgt);
            }
        }
    }
}

template <int NC, bool MCH>
__device__ __forceinline__ void splat_tile_atomics(const TiledParams& p, int tx, int ty, int n, int ncand, int64_t dtile,
                                                unsigned char* raw, int* qcount) {
    constexpr int kPx = kSpTW * kSpTH;
    constexpr int NPL = 2 + NC;
    float* acc = reinterpret_cast<float*>(raw);
    uint32_t* queue = reinterpret_cast<uint32_t*>(acc + NPL * kPx);
    const SplatParams& s = p.s;
    const int tid = threadIdx.x, lx = tid & 7, ly = tid >> 3, lane = tid & 63;
    const int w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    const int dx0 = tx * kSpTW, dy0 = ty * kSpTH;
    const float wf = (float)w, hf = (float)h;
    __syncthreads();
    for (int i = tid; i < NPL * kPx; i += kSpNT) acc[i] = 0.0f;
    if (tid == 0) *qcount = 0;
    __syncthreads();
    for (int ci = 0; ci < ncand; ++ci) {
        const int st = p.lists[dtile * kSpMaxCand + ci];
        const uint32_t sty = fastdiv((uint32_t)st, p.mx_m, p.mx_s), stx = (uint32_t)st - sty * (uint32_t)p.tiles_x;
        const int sx4 = (int)stx * kSpTW + lx * 4, sy = (int)sty * kSpTH + ly;
        const bool inimg = (sx4 < w) && (sy < h);
        SpSrc q;
        sp_load_src(s, n, sx4, sy, inimg, (uint32_t)(sy * w + sx4), hw, q);
        if (*qcount > kSpQueue2 - 4 * kSpNT) {            // block-uniform: make room first
            __syncthreads();
            sp_drain<NC, MCH>(p, n, dx0, dy0, queue, *qcount, acc);
            __syncthreads();
            if (tid == 0) *qcount = 0;
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool hit = false;
            if (q.on[k]) {
                const int x0 = (int)__builtin_amdgcn_fmed3f(floorf(q.x[k]), -2.0f, wf) - dx0;
                const int y0 = (int)__builtin_amdgcn_fmed3f(floorf(q.y[k]), -2.0f, hf) - dy0;
                hit = (x0 >= -1) && (x0 < kSpTW) && (y0 >= -1) && (y0 < kSpTH);
            }
            const unsigned long long m = __ballot(hit);
            if (m != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(qcount, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (hit) queue[base + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)sy << 16) | (uint32_t)(sx4 + k);
            }
        }
        __syncthreads();
    }
    sp_drain<NC, MCH>(p, n, dx0, dy0, queue, *qcount, acc);
    __syncthreads();
    const int x4 = dx0 + lx * 4, y = dy0 + ly;
    if (x4 >= w || y >= h) return;
    const uint32_t pix = (uint32_t)(y * w + x4);
    bool fill_ok[4] = {false, false, false, false};
    if (s.occlude && s.flow) {
        SpSrc own;
        sp_load_src(s, n, x4, y, true, pix, hw, own);
#pragma unroll
        for (int k = 0; k < 4; ++k) fill_ok[k] = own.zero[k] && own.wm[k];
    }
    const float* __restrict__ db = s.data + n * s.data_bs;
    f4 den4, out[NC], mch4;
    uint32_t warped4 = 0, valid4 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = ly * kSpTW + k * (kSpTW / 4) + lx;
        const float den = acc[idx];
        const float dcl = den < kDenMin ? kDenMin : den;
        const bool warped = den > 0.0f;
        const bool fill = fill_ok[k] && !warped;
        den4[k] = den;
        warped4 |= (uint32_t)warped << (8 * k);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            out[c][k] = apply_round(fill ? s.data_sign * db[c * hw + pix + k] : acc[(1 + c) * kPx + idx] / dcl, s.round_mode);
        if (MCH) {
            float mv;
            if (fill) {
                const bool a = s.chan_mask_a ? s.chan_mask_a[n * s.chan_mask_a_bs + pix + k] != 0 : true;
                const bool b = s.chan_mask_b ? s.chan_mask_b[n * s.chan_mask_b_bs + pix + k] != 0 : true;
                mv = (a && b) ? 1.0f : 0.0f;
            } else {
                mv = (den - acc[(1 + NC) * kPx + idx]) / dcl;
            }
            mch4[k] = mv;
            valid4 |= (uint32_t)(mv > kValidThr) << (8 * k);
        }
    }
    float* __restrict__ dst = s.dst + (int64_t)n * NC * hw;
#pragma unroll
    for (int c = 0; c < NC; ++c) *reinterpret_cast<f4*>(dst + c * hw + pix) = out[c];
    if (s.density) *reinterpret_cast<f4*>(s.density + (int64_t)n * hw + pix) = den4;
    if (s.warped) *reinterpret_cast<uint32_t*>(s.warped + (int64_t)n * hw + pix) = warped4;
    if (MCH && s.valid) *reinterpret_cast<uint32_t*>(s.valid + (int64_t)n * hw + pix) = valid4;
    if (MCH && s.mask_chan) *reinterpret_cast<f4*>(s.mask_chan + (int64_t)n * hw + pix) = mch4;
}

// The destination tile is accumulated WITHOUT float atomics: the touching source pixels are compacted into a queue,
// every (source pixel, corner) pair is binned to its destination pixel with one integer LDS atomic, and each thread then
// sums the bins of its own 4 destination pixels in registers.  A queue or bin overflow (strongly compressive flows)
// flags the whole launch for the atomics path.
template <int NC, bool MCH, bool BINNING>
__global__ __launch_bounds__(kSpNT) void splat_tile_kernel(const TiledParams p) {
    // LDS carving.  Binning path: qx | qy | qpix | bins | cnt.  Local atomics path (tile overflow): acc planes | queue2.
    constexpr int kPx = kSpTW * kSpTH;
    constexpr int kBytesA = kSpQueue * 12 + kPx * kSpBinCap * 2 + kPx * 4;
    constexpr int kBytesB = (2 + NC) * kPx * 4 + kSpQueue2 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char raw[(BINNING && kBytesA > kBytesB) ? kBytesA : kBytesB];
    __shared__ int qcount;
    float* qx = reinterpret_cast<float*>(raw);
    float* qy = qx + kSpQueue;
    uint32_t* qpix = reinterpret_cast<uint32_t*>(qy + kSpQueue);
    uint16_t* bins = reinterpret_cast<uint16_t*>(qpix + kSpQueue);
    int* cnt = reinterpret_cast<int*>(bins + kPx * kSpBinCap);
    if (*p.overflow != 0) return;                               // this launch takes the atomics path instead
    int tx, ty, n;
    if (!sp_decode(p, tx, ty, n)) return;
    const SplatParams& s = p.s;
    const int tid = threadIdx.x, lx = tid & 7, ly = tid >> 3, lane = tid & 63;
    const int w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    const int dx0 = tx * kSpTW, dy0 = ty * kSpTH;
#pragma unroll
    for (int i = 0; i < 4; ++i) cnt[tid + i * kSpNT] = 0;
    if (tid == 0) qcount = 0;
    const int64_t dtile = (int64_t)n * p.tiles_img + ty * p.tiles_x + tx;
    const int ncand = min(p.counts[dtile], kSpMaxCand);
    if (!BINNING) {         // default in round 1: LDS float atomics (measured faster than the binning variant below)
        splat_tile_atomics<NC, MCH>(p, tx, ty, n, ncand, dtile, raw, &qcount);
        return;
    }
    __syncthreads();
    const float wf = (float)w, hf = (float)h;
    const float wmax = (float)(w - 1), hmax = (float)(h - 1);
    // ---- phase A: walk the candidate source tiles (16-byte loads); queue the pixels that touch this tile
    bool qfull = false;
    for (int ci = 0; ci < ncand; ++ci) {
        const int st = p.lists[dtile * kSpMaxCand + ci];
        const uint32_t sty = fastdiv((uint32_t)st, p.mx_m, p.mx_s), stx = (uint32_t)st - sty * (uint32_t)p.tiles_x;
        const int sx4 = (int)stx * kSpTW + lx * 4, sy = (int)sty * kSpTH + ly;
        const bool inimg = (sx4 < w) && (sy < h);
        SpSrc q;
        sp_load_src(s, n, sx4, sy, inimg, (uint32_t)(sy * w + sx4), hw, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool hit = false;
            if (q.on[k]) {
                const int x0 = (int)__builtin_amdgcn_fmed3f(floorf(q.x[k]), -2.0f, wf) - dx0;
                const int y0 = (int)__builtin_amdgcn_fmed3f(floorf(q.y[k]), -2.0f, hf) - dy0;
                hit = (x0 >= -1) && (x0 < kSpTW) && (y0 >= -1) && (y0 < kSpTH);
            }
            const unsigned long long m = __ballot(hit);
            if (m != 0ull) {                             // wave-uniform
                int base = 0;
                if (lane == 0) base = atomicAdd(&qcount, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (hit) {
                    if (pos < kSpQueue) { qx[pos] = q.x[k]; qy[pos] = q.y[k]; qpix[pos] = ((uint32_t)sy << 16) | (uint32_t)(sx4 + k); }
                    else qfull = true;
                }
            }
        }
    }
    __syncthreads();
    const int qlen = min(qcount, kSpQueue);
    // ---- phase B: bin every (source pixel, corner) that lands inside with a non-zero weight to its destination pixel
    bool binfull = false;
    for (int i = tid; i < qlen; i += kSpNT) {
        const float xv = qx[i], yv = qy[i];
        const float x0 = floorf(xv), y0 = floorf(yv), x1 = x0 + 1.0f, y1 = y0 + 1.0f;
        const float x0s = fminf(fmaxf(x0, 0.0f), wmax), x1s = fminf(fmaxf(x1, 0.0f), wmax);
        const float y0s = fminf(fmaxf(y0, 0.0f), hmax), y1s = fminf(fmaxf(y1, 0.0f), hmax);
        const float wx[2] = {(x1 - xv) * (x0 == x0s ? 1.0f : 0.0f), (xv - x0) * (x1 == x1s ? 1.0f : 0.0f)};
        const float wy[2] = {(y1 - yv) * (y0 == y0s ? 1.0f : 0.0f), (yv - y0) * (y1 == y1s ? 1.0f : 0.0f)};
        const int ix[2] = {(int)x0s - dx0, (int)x1s - dx0}, iy[2] = {(int)y0s - dy0, (int)y1s - dy0};
#pragma unroll
        for (int ky = 0; ky < 2; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
                const int xl = ix[kx], yl = iy[ky];
                if (wy[ky] * wx[kx] == 0.0f || (uint32_t)xl >= (uint32_t)kSpTW || (uint32_t)yl >= (uint32_t)kSpTH) continue;
                const int d = yl * kSpTW + xl;
                const int slot = atomicAdd(&cnt[d], 1);
                if (slot < kSpBinCap) bins[d * kSpBinCap + slot] = (uint16_t)((i << 2) | (ky * 2 + kx));
                else binfull = true;
            }
        }
    }
    if (__syncthreads_or((int)(qfull || binfull))) {     // compressive flow here: this tile re-runs with LDS float atomics
        splat_tile_atomics<NC, MCH>(p, tx, ty, n, ncand, dtile, raw, &qcount);
        return;
    }
    // ---- phase C: every thread sums the bins of its own 4 destination pixels in registers, then finalizes them
    const int x4 = dx0 + lx * 4, y = dy0 + ly;
    if (x4 >= w || y >= h) return;
    const uint32_t pix = (uint32_t)(y * w + x4);
    bool fill_ok[4] = {false, false, false, false};      // un-occlude fill candidates (utils.py:1198-1203)
    if (s.occlude && s.flow) {
        SpSrc own;
        sp_load_src(s, n, x4, y, true, pix, hw, own);
#pragma unroll
        for (int k = 0; k < 4; ++k) fill_ok[k] = own.zero[k] && own.wm[k];
    }
    const float* __restrict__ db = s.data + n * s.data_bs;
    const uint8_t* __restrict__ cma = s.chan_mask_a ? s.chan_mask_a + n * s.chan_mask_a_bs : nullptr;
    const uint8_t* __restrict__ cmb = s.chan_mask_b ? s.chan_mask_b + n * s.chan_mask_b_bs : nullptr;
    f4 den4, out[NC], mch4;
    uint32_t warped4 = 0, valid4 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = ly * kSpTW + lx * 4 + k;
        const int m = min(cnt[d], kSpBinCap);
        float den = 0.0f, inv = 0.0f, sum[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) sum[c] = 0.0f;
        for (int j = 0; j < m; ++j) {
            const uint32_t e = bins[d * kSpBinCap + j];
            const int i = (int)(e >> 2), kx = (int)(e & 1u), ky = (int)((e >> 1) & 1u);
            const float xv = qx[i], yv = qy[i];
            const uint32_t sp = qpix[i];
            const uint32_t spix = (sp >> 16) * (uint32_t)w + (sp & 0xffffu);
            // the corner's weight, exactly as in phase B / the reference (utils.py:1106-1114)
            const float x0 = floorf(xv), y0 = floorf(yv);
            const float xc = kx ? x0 + 1.0f : x0, yc = ky ? y0 + 1.0f : y0;
            const float xcs = fminf(fmaxf(xc, 0.0f), wmax), ycs = fminf(fmaxf(yc, 0.0f), hmax);
            const float wxk = (kx ? xv - x0 : (x0 + 1.0f) - xv) * (xc == xcs ? 1.0f : 0.0f);
            const float wyk = (ky ? yv - y0 : (y0 + 1.0f) - yv) * (yc == ycs ? 1.0f : 0.0f);
            const float wgt = wyk * wxk;
            den += wgt;
#pragma unroll
            for (int c = 0; c < NC; ++c) sum[c] += wgt * (s.data_sign * db[c * hw + spix]);
            if (MCH) {
                const bool a = cma ? cma[spix] != 0 : true, b = cmb ? cmb[spix] != 0 : true;
                if (!(a && b)) inv += wgt;
            }
        }
        const float dcl = den < kDenMin ? kDenMin : den;          // clamp_min utils.py:1144
        const bool warped = den > 0.0f;                            // utils.py:1197
        const bool fill = fill_ok[k] && !warped;
        den4[k] = den;
        warped4 |= (uint32_t)warped << (8 * k);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            out[c][k] = apply_round(fill ? s.data_sign * db[c * hw + pix + k] : sum[c] / dcl, s.round_mode);
        if (MCH) {
            float mv;
            if (fill) {
                const bool a = cma ? cma[pix + k] != 0 : true, b = cmb ? cmb[pix + k] != 0 : true;
                mv = (a && b) ? 1.0f : 0.0f;
            } else {
                mv = (den - inv) / dcl;
            }
            mch4[k] = mv;
            valid4 |= (uint32_t)(mv > kValidThr) << (8 * k);
        }
    }
    float* __restrict__ dst = s.dst + (int64_t)n * NC * hw;
#pragma unroll
    for (int c = 0; c < NC; ++c) *reinterpret_cast<f4*>(dst + c * hw + pix) = out[c];
    if (s.density) *reinterpret_cast<f4*>(s.density + (int64_t)n * hw + pix) = den4;
    if (s.warped) *reinterpret_cast<uint32_t*>(s.warped + (int64_t)n * hw + pix) = warped4;
    if (MCH && s.valid) *reinterpret_cast<uint32_t*>(s.valid + (int64_t)n * hw + pix) = valid4;
    if (MCH && s.mask_chan) *reinterpret_cast<f4*>(s.mask_chan + (int64_t)n * hw + pix) = mch4;
}

// zero the fallback accumulator only when the atomics path will run
__global__ __launch_bounds__(256) void zero_if_set_kernel(float* __restrict__ ptr, int64_t n4, const int32_t* __restrict__ flag) {
    if (*flag == 0) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        reinterpret_cast<f4*>(ptr)[i] = (f4){0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------------
// flow flags (ofl_flow_flags_f32)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_flags_kernel(const float* __restrict__ flow, int64_t flow_bs,
                                                         const uint8_t* __restrict__ mask, int64_t mask_bs,
                                                         int32_t* __restrict__ flags, int64_t hw) {
    const int n = blockIdx.y;
    const float* fu = flow + n * flow_bs;
    const uint8_t* mk = mask ? mask + n * mask_bs : nullptr;
    int f = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x)
        f |= flag_bits(fu[i], fu[hw + i], mk ? (mk[i] != 0) : true);
    f = wave_or_flags(f);
    if ((threadIdx.x & 63) == 0 && f) atomicOr(&flags[n], f);
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
inline int check_dims(int32_t n, int32_t c, int32_t h, int32_t w) {
    if (n < 1 || c < 1 || h < 1 || w < 1) return OFL_E_SHAPE;
    if ((int64_t)h * w >= (1ll << 24)) return OFL_E_SHAPE;  // fp32 position index limit, utils.py:1118
    return OFL_OK;
}

inline void tile_grid(int32_t n, int32_t h, int32_t w, int32_t& tiles_x, int32_t& tiles_y, int64_t& total,
                      int64_t& per_xcd, unsigned& grid) {
    tiles_x = (w + kTileW - 1) / kTileW;
    tiles_y = (h + kTileH - 1) / kTileH;
    total = (int64_t)tiles_x * tiles_y * n;
    per_xcd = (total + kXcds - 1) / kXcds;
    grid = (unsigned)(per_xcd * kXcds);
}

inline void magic_u32(uint32_t d, uint32_t& m, uint32_t& s) {   // d >= 1; see fastdiv()
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    s = ((l ? 1u : 0u) << 16) | (l ? l - 1 : 0);
}

inline unsigned warp_geometry(WarpParams& p, int tile_w, int tile_h) {
    p.tiles_x = (p.w + tile_w - 1) / tile_w;
    p.tiles_y = (p.h + tile_h - 1) / tile_h;
    p.tiles_img = (uint32_t)(p.tiles_x * p.tiles_y);
    p.total_tiles = (int64_t)p.tiles_img * p.n;
    p.per_xcd = (p.total_tiles + kXcds - 1) / kXcds;
    magic_u32((uint32_t)p.tiles_x, p.mx_m, p.mx_s);
    magic_u32(p.tiles_img, p.mi_m, p.mi_s);
    return (unsigned)(p.per_xcd * kXcds);
}

int g_warp_path = 0;   // ofl_set_option(OFL_OPT_WARP_PATH, .): 0 auto, 1 generic direct-gather kernel only, 2 (= auto)
int g_warp_shear = 1;   // ofl_set_option(OFL_OPT_WARP_SHEAR, .)
int g_splat_binning = 0;   // ofl_set_option(OFL_OPT_SPLAT_BINNING, .): 1 = atomic-free binning variant of the tiled splat

template <int NC>
int launch_warp_lds(const WarpParams& p, unsigned grid, hipStream_t st) {
    const bool valid = p.valid != nullptr, add = p.addend != nullptr;
#define OFL_LAUNCH_L(V, A)                                                                                       \
    if (valid == V && add == A) {                                                                                \
        hipLaunchKernelGGL((warp_bwd_lds_kernel<NC, V, A>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);          \
        return (int)hipGetLastError();                                                                           \
    }
    OFL_LAUNCH_L(false, false) OFL_LAUNCH_L(true, false) OFL_LAUNCH_L(false, true) OFL_LAUNCH_L(true, true)
#undef OFL_LAUNCH_L
    return OFL_E_ARG;
}

inline bool aligned_to(const void* ptr, size_t a) { return (reinterpret_cast<uintptr_t>(ptr) % a) == 0; }

template <int CT>
int launch_warp(const WarpParams& p, unsigned grid, hipStream_t st) {
    const bool valid = p.valid != nullptr, add = p.addend != nullptr, flags = p.flow_flags != nullptr;
#define OFL_LAUNCH_W(V, A, F)                                                                  \
    if (valid == V && add == A && flags == F) {                                                \
        hipLaunchKernelGGL((warp_bwd_kernel<CT, V, A, F>), dim3(grid), dim3(256), 0, st, p);    \
        return (int)hipGetLastError();                                                         \
    }
    OFL_LAUNCH_W(false, false, false) OFL_LAUNCH_W(true, false, false)
    OFL_LAUNCH_W(false, true, false) OFL_LAUNCH_W(true, true, false)
    OFL_LAUNCH_W(false, false, true) OFL_LAUNCH_W(true, false, true)
    OFL_LAUNCH_W(false, true, true) OFL_LAUNCH_W(true, true, true)
#undef OFL_LAUNCH_W
    return OFL_E_ARG;
}


template <int NC>
int launch_splat_tile(const TiledParams& tp, unsigned grid, hipStream_t st) {
    if (tp.binning) {
        if (tp.s.with_mask_chan) hipLaunchKernelGGL((splat_tile_kernel<NC, true, true>), dim3(grid), dim3(kSpNT), 0, st, tp);
        else hipLaunchKernelGGL((splat_tile_kernel<NC, false, true>), dim3(grid), dim3(kSpNT), 0, st, tp);
    } else {
        if (tp.s.with_mask_chan) hipLaunchKernelGGL((splat_tile_kernel<NC, true, false>), dim3(grid), dim3(kSpNT), 0, st, tp);
        else hipLaunchKernelGGL((splat_tile_kernel<NC, false, false>), dim3(grid), dim3(kSpNT), 0, st, tp);
    }
    return (int)hipGetLastError();
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

__attribute__((visibility("default"))) int ofl_version(void) { return 11; }

__attribute__((visibility("default"))) int ofl_set_option(int32_t key, int32_t value) {
    if (key == OFL_OPT_WARP_PATH && value >= 0 && value <= 2) { g_warp_path = value; return OFL_OK; }
    if (key == OFL_OPT_SPLAT_BINNING && (value == 0 || value == 1)) { g_splat_binning = value; return OFL_OK; }
    if (key == OFL_OPT_WARP_SHEAR && (value == 0 || value == 1)) { g_warp_shear = value; return OFL_OK; }
    return OFL_E_ARG;
}

__attribute__((visibility("default"))) int ofl_warp_bwd_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* src, int64_t src_bs,
    const uint8_t* src_mask, int64_t src_mask_bs, const uint8_t* flow_mask, int64_t flow_mask_bs,
    const float* addend, int64_t addend_bs, float a_sign, float g_sign, float* dst, uint8_t* valid,
    int32_t* flow_flags, int32_t* src_flags, int32_t n, int32_t c, int32_t h, int32_t w, int32_t round_mode,
    void* stream) {
    if (!flow || !src || !dst) return OFL_E_NULL;
    int rc = check_dims(n, c, h, w);
    if (rc) return rc;
    if (src_flags && (c != 2 || !flow_flags)) return OFL_E_ARG;
    if (round_mode < 0 || round_mode > 2) return OFL_E_ARG;
    if (!(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    WarpParams p;
    p.flow = flow; p.flow_bs = flow_bs; p.src = src; p.src_bs = src_bs;
    p.src_mask = src_mask; p.src_mask_bs = src_mask_bs; p.flow_mask = flow_mask; p.flow_mask_bs = flow_mask_bs;
    p.addend = addend; p.addend_bs = addend_bs; p.dst = dst; p.valid = valid;
    p.flow_flags = flow_flags; p.src_flags = src_flags;
    p.n = n; p.c = c; p.h = h; p.w = w;
    p.flow_sign = flow_sign; p.a_sign = a_sign; p.g_sign = g_sign; p.round_mode = round_mode;
    p.wm1 = (float)(w - 1); p.hm1 = (float)(h - 1);
    p.half_wm1 = p.wm1 / 2.0f; p.half_hm1 = p.hm1 / 2.0f;
    p.rcp_wm1 = 1.0f / p.wm1; p.rcp_hm1 = 1.0f / p.hm1;
    p.lds_bytes = kLdsBytes; p.shear = g_warp_shear;
    hipStream_t st = (hipStream_t)stream;
    if ((int64_t)((w + 31) / 32) * ((h + 15) / 16) * n >= (1ll << 31)) return OFL_E_SHAPE;
    // LDS-staged fast path: <= 3 channels, rows that are whole 16-byte groups, 16-byte aligned planes
    const bool lds_ok = g_warp_path != 1 && c <= 3 && w >= 4 && (w % 4) == 0 && h >= 2 &&
                        aligned_to(flow, 16) && aligned_to(src, 16) && aligned_to(dst, 16) && (flow_bs % 4) == 0 &&
                        (src_bs % 4) == 0 && (!addend || (aligned_to(addend, 16) && (addend_bs % 4) == 0)) &&
                        (!src_mask || (aligned_to(src_mask, 4) && (src_mask_bs % 4) == 0)) &&
                        (!flow_mask || (aligned_to(flow_mask, 4) && (flow_mask_bs % 4) == 0)) &&
                        (!valid || aligned_to(valid, 4));
    if (lds_ok) {
        if (src_flags) {   // the staged path never reads `src` at its own pixel: a separate reduction supplies its flags
            const int64_t hw = (int64_t)h * w;
            int64_t bx = (hw + 255) / 256;
            if (bx > 512) bx = 512;
            hipLaunchKernelGGL(flow_flags_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, src, src_bs, src_mask,
                               src_mask_bs, src_flags, hw);
            p.src_flags = nullptr;
        }
        const unsigned g = warp_geometry(p, kLdsTWQ * 4, 2 * kLdsTH);
        switch (c) {
            case 1: return launch_warp_lds<1>(p, g, st);
            case 2: return launch_warp_lds<2>(p, g, st);
            default: return launch_warp_lds<3>(p, g, st);
        }
    }
    const unsigned grid = warp_geometry(p, kTileW, kTileH);
    switch (c) {
        case 1: return launch_warp<1>(p, grid, st);
        case 2: return launch_warp<2>(p, grid, st);
        case 3: return launch_warp<3>(p, grid, st);
        case 4: return launch_warp<4>(p, grid, st);
        default: return launch_warp<0>(p, grid, st);
    }
}

static int fill_splat(SplatParams& p, const float* flow, int64_t flow_bs, const float* data, int64_t data_bs,
                      float data_sign, const uint8_t* weight_mask, int64_t weight_mask_bs,
                      const uint8_t* chan_mask_a, int64_t chan_mask_a_bs, const uint8_t* chan_mask_b,
                      int64_t chan_mask_b_bs, int32_t with_mask_chan, int32_t occlude, int32_t n, int32_t c,
                      int32_t h, int32_t w, unsigned& grid) {
    int rc = check_dims(n, c, h, w);
    if (rc) return rc;
    if (occlude && !flow) return OFL_E_ARG;
    if (!(data_sign == 1.0f || data_sign == -1.0f)) return OFL_E_ARG;
    p.flow = flow; p.flow_bs = flow_bs; p.data = data; p.data_bs = data_bs; p.data_sign = data_sign;
    p.weight_mask = weight_mask; p.weight_mask_bs = weight_mask_bs;
    p.chan_mask_a = chan_mask_a; p.chan_mask_a_bs = chan_mask_a_bs;
    p.chan_mask_b = chan_mask_b; p.chan_mask_b_bs = chan_mask_b_bs;
    p.with_mask_chan = with_mask_chan; p.occlude = occlude;
    p.n = n; p.c = c; p.h = h; p.w = w;
    tile_grid(n, h, w, p.tiles_x, p.tiles_y, p.total_tiles, p.per_xcd, grid);
    return OFL_OK;
}

__attribute__((visibility("default"))) int ofl_splat_fwd_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* xs, const float* ys, int64_t xy_bs,
    const float* data, int64_t data_bs, float data_sign, const uint8_t* weight_mask, int64_t weight_mask_bs,
    const uint8_t* chan_mask_a, int64_t chan_mask_a_bs, const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
    int32_t with_mask_chan, int32_t occlude, float* accum, int32_t n, int32_t c, int32_t h, int32_t w,
    void* stream) {
    if (!data || !accum) return OFL_E_NULL;
    if (!flow && !(xs && ys)) return OFL_E_NULL;
    if (flow && !(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    SplatParams p = {};
    unsigned grid;
    int rc = fill_splat(p, flow, flow_bs, data, data_bs, data_sign, weight_mask, weight_mask_bs, chan_mask_a,
                        chan_mask_a_bs, chan_mask_b, chan_mask_b_bs, with_mask_chan, occlude, n, c, h, w, grid);
    if (rc) return rc;
    p.flow_sign = flow_sign; p.xs = xs; p.ys = ys; p.xy_bs = xy_bs; p.accum = accum;
    hipStream_t st = (hipStream_t)stream;
    switch (c) {
        case 1: hipLaunchKernelGGL(splat_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(splat_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(splat_fwd_kernel<3>, dim3(grid), dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL(splat_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(splat_fwd_kernel<0>, dim3(grid), dim3(256), 0, st, p); break;
    }
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_splat_finalize_f32(
    const float* accum, const float* flow, int64_t flow_bs, const float* data, int64_t data_bs, float data_sign,
    const uint8_t* weight_mask, int64_t weight_mask_bs, const uint8_t* chan_mask_a, int64_t chan_mask_a_bs,
    const uint8_t* chan_mask_b, int64_t chan_mask_b_bs, int32_t with_mask_chan, int32_t occlude, float* dst,
    float* density, uint8_t* warped, uint8_t* valid, float* mask_chan, int32_t n, int32_t c, int32_t h, int32_t w,
    int32_t round_mode, void* stream) {
    if (!accum || !data || !dst) return OFL_E_NULL;
    if ((valid || mask_chan) && !with_mask_chan) return OFL_E_ARG;
    if (round_mode < 0 || round_mode > 2) return OFL_E_ARG;
    SplatParams p = {};
    unsigned grid;
    int rc = fill_splat(p, flow, flow_bs, data, data_bs, data_sign, weight_mask, weight_mask_bs, chan_mask_a,
                        chan_mask_a_bs, chan_mask_b, chan_mask_b_bs, with_mask_chan, occlude, n, c, h, w, grid);
    if (rc) return rc;
    p.accum = const_cast<float*>(accum);
    p.dst = dst; p.density = density; p.warped = warped; p.valid = valid; p.mask_chan = mask_chan;
    p.round_mode = round_mode;
    hipStream_t st = (hipStream_t)stream;
    switch (c) {
        case 1: hipLaunchKernelGGL(splat_finalize_kernel<1>, dim3(grid), dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(splat_finalize_kernel<2>, dim3(grid), dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(splat_finalize_kernel<3>, dim3(grid), dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL(splat_finalize_kernel<4>, dim3(grid), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(splat_finalize_kernel<0>, dim3(grid), dim3(256), 0, st, p); break;
    }
    return (int)hipGetLastError();
}


__attribute__((visibility("default"))) int64_t ofl_splat_tiled_workspace_ints(int32_t n, int32_t h, int32_t w) {
    const int64_t tiles = (int64_t)((w + kSpTW - 1) / kSpTW) * ((h + kSpTH - 1) / kSpTH) * n;
    return tiles * (1 + kSpMaxCand) + 4;
}

__attribute__((visibility("default"))) int ofl_splat_tiled_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* xs, const float* ys, int64_t xy_bs,
    const float* data, int64_t data_bs, float data_sign, const uint8_t* weight_mask, int64_t weight_mask_bs,
    const uint8_t* chan_mask_a, int64_t chan_mask_a_bs, const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
    int32_t with_mask_chan, int32_t occlude, float* dst, float* density, uint8_t* warped, uint8_t* valid,
    float* mask_chan, int32_t* workspace, int64_t workspace_ints, float* accum_fallback, int32_t n, int32_t c,
    int32_t h, int32_t w, int32_t round_mode, void* stream) {
    if (!data || !dst || !workspace || !accum_fallback) return OFL_E_NULL;
    if (!flow && !(xs && ys)) return OFL_E_NULL;
    if (flow && !(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    if ((valid || mask_chan) && !with_mask_chan) return OFL_E_ARG;
    if (round_mode < 0 || round_mode > 2) return OFL_E_ARG;
    TiledParams tp = {};
    unsigned grid_unused;
    int rc = fill_splat(tp.s, flow, flow_bs, data, data_bs, data_sign, weight_mask, weight_mask_bs, chan_mask_a,
                        chan_mask_a_bs, chan_mask_b, chan_mask_b_bs, with_mask_chan, occlude, n, c, h, w, grid_unused);
    if (rc) return rc;
    // eligibility of the tiled path: <= 3 channels, rows of whole 16-byte groups, aligned planes
    const bool ok = c <= 3 && w >= 4 && (w % 4) == 0 && aligned_to(data, 16) && aligned_to(dst, 16) && (data_bs % 4) == 0 &&
                    (!flow || (aligned_to(flow, 16) && (flow_bs % 4) == 0)) &&
                    (!xs || (aligned_to(xs, 16) && aligned_to(ys, 16) && (xy_bs % 4) == 0)) &&
                    (!weight_mask || (aligned_to(weight_mask, 4) && (weight_mask_bs % 4) == 0)) &&
                    (!chan_mask_a || (aligned_to(chan_mask_a, 4) && (chan_mask_a_bs % 4) == 0)) &&
                    (!chan_mask_b || (aligned_to(chan_mask_b, 4) && (chan_mask_b_bs % 4) == 0)) &&
                    (!density || aligned_to(density, 16)) && (!mask_chan || aligned_to(mask_chan, 16)) &&
                    (!warped || aligned_to(warped, 4)) && (!valid || aligned_to(valid, 4));
    if (!ok) return OFL_E_UNSUPPORTED;
    if (workspace_ints < ofl_splat_tiled_workspace_ints(n, h, w)) return OFL_E_ARG;
    tp.s.flow_sign = flow_sign; tp.s.xs = xs; tp.s.ys = ys; tp.s.xy_bs = xy_bs;
    tp.s.dst = dst; tp.s.density = density; tp.s.warped = warped; tp.s.valid = valid; tp.s.mask_chan = mask_chan;
    tp.s.round_mode = round_mode;
    tp.tiles_x = (w + kSpTW - 1) / kSpTW; tp.tiles_y = (h + kSpTH - 1) / kSpTH;
    tp.tiles_img = (uint32_t)(tp.tiles_x * tp.tiles_y);
    tp.total = (int64_t)tp.tiles_img * n;
    if (tp.total >= (1ll << 31)) return OFL_E_SHAPE;
    tp.per_xcd = (tp.total + kXcds - 1) / kXcds;
    magic_u32((uint32_t)tp.tiles_x, tp.mx_m, tp.mx_s);
    magic_u32(tp.tiles_img, tp.mi_m, tp.mi_s);
    tp.binning = g_splat_binning;
    tp.counts = workspace;
    tp.lists = workspace + tp.total;
    tp.overflow = workspace + tp.total * (1 + kSpMaxCand);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(tp.counts, 0, (size_t)tp.total * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(tp.overflow, 0, 4 * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    const unsigned grid = (unsigned)(tp.per_xcd * kXcds);
    hipLaunchKernelGGL(splat_bin_kernel, dim3(grid), dim3(kSpNT), 0, st, tp);
    rc = (int)hipGetLastError();
    if (rc) return rc;
    switch (c) {
        case 1: rc = launch_splat_tile<1>(tp, grid, st); break;
        case 2: rc = launch_splat_tile<2>(tp, grid, st); break;
        default: rc = launch_splat_tile<3>(tp, grid, st); break;
    }
    if (rc) return rc;
    // atomics path, armed only if a candidate list overflowed (rough flows); every kernel below exits at once otherwise
    SplatParams fb = tp.s;
    fb.accum = accum_fallback;
    fb.run_if_set = tp.overflow;
    const int planes = 1 + c + (with_mask_chan ? 1 : 0);
    const int64_t n4 = (int64_t)n * planes * h * w / 4;
    hipLaunchKernelGGL(zero_if_set_kernel, dim3(2048), dim3(256), 0, st, accum_fallback, n4, tp.overflow);
    unsigned g2;
    tile_grid(n, h, w, fb.tiles_x, fb.tiles_y, fb.total_tiles, fb.per_xcd, g2);
    switch (c) {
        case 1: hipLaunchKernelGGL(splat_fwd_kernel<1>, dim3(g2), dim3(256), 0, st, fb);
                hipLaunchKernelGGL(splat_finalize_kernel<1>, dim3(g2), dim3(256), 0, st, fb); break;
        case 2: hipLaunchKernelGGL(splat_fwd_kernel<2>, dim3(g2), dim3(256), 0, st, fb);
                hipLaunchKernelGGL(splat_finalize_kernel<2>, dim3(g2), dim3(256), 0, st, fb); break;
        default: hipLaunchKernelGGL(splat_fwd_kernel<3>, dim3(g2), dim3(256), 0, st, fb);
                 hipLaunchKernelGGL(splat_finalize_kernel<3>, dim3(g2), dim3(256), 0, st, fb); break;
    }
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_flow_flags_f32(const float* flow, int64_t flow_bs,
                                                              const uint8_t* mask, int64_t mask_bs, float thr,
                                                              int32_t* flags, int32_t n, int32_t h, int32_t w,
                                                              void* stream) {
    if (!flow || !flags) return OFL_E_NULL;
    int rc = check_dims(n, 2, h, w);
    if (rc) return rc;
    if (thr != kZeroThr) return OFL_E_ARG;  // the reference's DEFAULT_THRESHOLD is the only value on the path
    const int64_t hw = (int64_t)h * w;
    int64_t bx = (hw + 255) / 256;
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(flow_flags_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, flow,
                       flow_bs, mask, mask_bs, flags, hw);
    return (int)hipGetLastError();
}

}  // extern "C"
