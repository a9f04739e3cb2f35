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
#include <type_traits>

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

// OR a wave's flag word into a per-image word that thousands of waves share: same-address atomics serialise in L2
// (~8 ns each), and the word saturates after the first few tiles -- so look first (a load served by L2, never by the
// CU's own L1) and only send the atomic when it would change something.
__device__ __forceinline__ void flag_or(int32_t* addr, int f) {
    if (f == 0) return;
    const int cur = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((cur | f) != cur) atomicOr(addr, f);
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
    const float* src_b; int64_t src_b_bs;      // optional (staged kernel, 2 channels): the gathered field is src - src_b (one fp32 subtraction per value)
    const uint8_t* src_mask; int64_t src_mask_bs;
    const uint8_t* flow_mask; int64_t flow_mask_bs;
    const float* addend; int64_t addend_bs;
    float* dst; uint8_t* valid;
    int32_t* flow_flags; int32_t* src_flags;
    int32_t* dst_flags;              // optional (LDS path, 2 channels): flag word of the output read as a flow under `valid`
    int32_t n, c, h, w;
    float flow_sign, a_sign, g_sign;
    int32_t round_mode;
    float wm1, hm1, half_wm1, half_hm1;
    float rcp_wm1, rcp_hm1;          // RN(1/(w-1)), RN(1/(h-1)) for the exact reciprocal division
    int32_t tiles_x, tiles_y;
    int64_t total_tiles, per_xcd;
    uint32_t tiles_img, mx_m, mx_s, mi_m, mi_s;   // magic divisors by tiles_x and by tiles per image
    int32_t lds_bytes, shear;
    int32_t add_is_flow;             // mode 3: the addend is the flow operand itself (same planes): no second fetch
    int64_t dst_bs;                  // LDS path: batch stride of dst (channels of the whole tensor * h * w)
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

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// global accesses of the fast paths: 16 / 8 bytes of floats at 4-byte alignment, 4 / 2 mask bytes at any alignment (one
// instruction each on gfx950), so that image widths need not be multiples of 4
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) U16u { uint16_t v; };
__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4u*>(p); }
// read-once operands (the flow of a warp): non-temporal, so that they do not push the gathered image's halo out of L2
__device__ __forceinline__ f4 ld4nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f4u*>(p)); }
__device__ __forceinline__ f2 ld2(const float* p) { return *reinterpret_cast<const f2u*>(p); }
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4u*>(p) = v; }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p);
__device__ __forceinline__ uint32_t ld16(const uint8_t* p);
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v);
// 8-bit images (the staged warp kernel's uint8 variants): 4 / 2 / 1 pixels of one channel <-> floats
__device__ __forceinline__ f4 ld4(const uint8_t* p) {
    const uint32_t b = ld32(p);
    return (f4){(float)(b & 0xffu), (float)((b >> 8) & 0xffu), (float)((b >> 16) & 0xffu), (float)(b >> 24)};
}
__device__ __forceinline__ f2 ld2(const uint8_t* p) { const uint32_t b = ld16(p); return (f2){(float)(b & 0xffu), (float)(b >> 8)}; }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const uint8_t* p) { return (float)*p; }
__device__ __forceinline__ void st4(uint8_t* p, f4 v) {            // (values already rounded and clamped to [0, 255])
    st32(p, (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24));
}
__device__ __forceinline__ void st2(float* p, f2 v) { *reinterpret_cast<f2u*>(p) = v; }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return reinterpret_cast<const U32u*>(p)->v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return reinterpret_cast<const U16u*>(p)->v; }
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { reinterpret_cast<U32u*>(p)->v = v; }
__device__ __forceinline__ void st16(uint8_t* p, uint32_t v) { reinterpret_cast<U16u*>(p)->v = (uint16_t)v; }
// a 4-pixel group that starts `d` pixels after (width - 4), the last position a whole group fits in a row (widths that are
// not multiples of 4): read that last whole group and rotate; the pixels past the row end are never used
__device__ __forceinline__ f4 rot4(f4 v, int d) { return d == 1 ? (f4){v[1], v[2], v[3], v[3]} : d == 2 ? (f4){v[2], v[3], v[3], v[3]} : (f4){v[3], v[3], v[3], v[3]}; }

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
// the same for two 16-bit lanes at once (v_pk_min_i16 / v_pk_max_i16): (x, y) boxes reduce in half the instructions
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pk_min16(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b))); }
__device__ __forceinline__ int pk_max16(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b))); }
__device__ __forceinline__ int wave_pk_min_dpp(int v) {
    v = pk_min16(v, OFL_DPP(v, 0xB1)); v = pk_min16(v, OFL_DPP(v, 0x4E)); v = pk_min16(v, OFL_DPP(v, 0x141)); v = pk_min16(v, OFL_DPP(v, 0x140));
    return pk_min16(pk_min16(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                    pk_min16(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_pk_max_dpp(int v) {
    v = pk_max16(v, OFL_DPP(v, 0xB1)); v = pk_max16(v, OFL_DPP(v, 0x4E)); v = pk_max16(v, OFL_DPP(v, 0x141)); v = pk_max16(v, OFL_DPP(v, 0x140));
    return pk_max16(pk_max16(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                    pk_max16(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
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
struct LdsBox { int bx0, miny, cw, Pp, bh, nch, sq, cbase; bool fits, interior; };   // wave-uniform staging geometry (interior: box staged, every tap of every pixel inside the image)

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
        if (tx == 0 && xc == 0) bad |= (fabsf(ax[0].x) < 0x1p-60f) && (ax[0].x != 0.0f);
        if (ty == 0 && yc == 0) {
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
    // block-wide box: columns and (sheared) rows fit 16 bits (checked on the host), so (x, y) pairs reduce together
    int lo = (int)(((uint32_t)minx & 0xffffu) | ((uint32_t)miny << 16)), hi = (int)(((uint32_t)maxx & 0xffffu) | ((uint32_t)maxy << 16));
    lo = wave_pk_min_dpp(lo); hi = wave_pk_max_dpp(hi);
    if (NW > 1) {
        if ((tid & 63) == 0) { red[tid >> 6][0] = lo; red[tid >> 6][1] = hi; }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < NW; ++i) { lo = pk_min16(lo, red[i][0]); hi = pk_max16(hi, red[i][1]); }
    }
    minx = (int)(short)(lo & 0xffff); miny = lo >> 16; maxx = (int)(short)(hi & 0xffff); maxy = hi >> 16;
    // touched columns [minx, maxx + 1] clipped to the image; sheared rows [miny, maxy] as they are (a staged row that falls
    // outside the image is skipped and never read back)   (wave-uniform)
    minx = __builtin_amdgcn_readfirstlane(minx); maxx = __builtin_amdgcn_readfirstlane(maxx) + 1;
    const bool xin = (minx >= 0) && (maxx <= w - 1);             // no tap column was clipped
    minx = max(minx, 0); maxx = min(maxx, w - 1);
    miny = __builtin_amdgcn_readfirstlane(miny); maxy = __builtin_amdgcn_readfirstlane(maxy);
    const bool empty = maxx < minx;
    B.bx0 = minx & ~3; B.miny = miny; B.sq = sq; B.cbase = B.bx0 >> 2;
    const int bw = empty ? 4 : (((maxx + 4) & ~3) - B.bx0);
    B.bh = empty ? 1 : (maxy - miny + 1); B.cw = bw >> 2; B.Pp = lds_pitch(bw); B.nch = B.bh * B.cw;
    B.fits = !empty && (B.bh <= 4096) && (16 * (1 + B.bh * B.Pp) <= p.lds_bytes) && (B.nch <= kLdsIters * kLdsNT);
    // image rows the box's first and last chunk column cover (the shear is monotonic in the column)
    const int sa = lds_shear(B.cbase, sq), sb_ = lds_shear(B.cbase + B.cw - 1, sq);
    B.interior = B.fits && xin && (miny + min(sa, sb_) >= 0) && (maxy + max(sa, sb_) <= h - 1);
}

// step 3a: issue the staging loads of a tile into registers (nothing waits here)
template <int NC, bool VALID, bool SUB = false, typename TS = float>
__device__ __forceinline__ void lds_issue(const WarpParams& p, const TS* __restrict__ sb, const uint8_t* __restrict__ sm,
                                          uint32_t hw, const LdsBox& B, LdsStage<NC>& S, const float* __restrict__ sbb = nullptr) {
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
            // the last chunk of a row of an image whose width is not a multiple of 4 would read past the row end (and past
            // the buffer on the last row): fetch the last whole group instead and rotate (block-uniform branch)
            const int wrem = p.w & 3;
            const bool edge = wrem != 0 && on && (int)(B.bx0 + (int)c4 * 4) > p.w - 4;
            const uint32_t ge = edge ? g - (uint32_t)(4 - wrem) : g;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                S.q[it][c] = ld4(sb + c * hw + ge);
                if (SUB) S.q[it][c] = S.q[it][c] - ld4(sbb + c * hw + ge);     // (mode 1 't': the warped field is flow - self)
            }
            S.mq[it] = (VALID && sm) ? ld32(sm + ge) : 0x01010101u;
            if (wrem != 0) {
                if (edge) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) S.q[it][c] = rot4(S.q[it][c], 4 - wrem);
                    S.mq[it] >>= 8 * (4 - wrem);
                }
            }
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
                if (VALID) sl[3] = fminf((float)((S.mq[it] >> (8 * k)) & 0xffu), 1.0f);   // non-zero byte -> 1 (v_cvt_f32_ubyteK + v_min)
                lds[S.slot[it] + k * B.cw] = sl;
            }
        }
    }
}

// step 4a: gather from LDS (or from global memory when the box did not fit) and blend; per pixel (c0, c1, c2, mask channel)
template <int NC, bool VALID, bool INTERIOR, bool SUB = false, typename TS = float>
__device__ __forceinline__ void lds_gather_impl(const WarpParams& p, uint32_t hw,
                                                const TS* __restrict__ sb, const uint8_t* __restrict__ sm,
                                                const LdsCoords& T, const LdsBox& B, const unsigned char* smem, f4 (&outv)[4],
                                                const float* __restrict__ sbb = nullptr) {
    const int w = p.w, h = p.h;
    const int cw16 = B.cw * 16, P16 = B.Pp * 16;
    const float wf = (float)w, hf = (float)h;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float fx = floorf(T.sx[k]), fy = floorf(T.sy[k]);
        const float ww = T.sx[k] - fx, e = 1.0f - ww, nn = T.sy[k] - fy, s = 1.0f - nn;
        const float wg[4] = {s * e, s * ww, nn * e, nn * ww};
        // interior tile: the clamps are the identity and every tap is valid
        const int xi = INTERIOR ? (int)fx : (int)__builtin_amdgcn_fmed3f(fx, -2.0f, wf);
        const int yi = INTERIOR ? (int)fy : (int)__builtin_amdgcn_fmed3f(fy, -2.0f, hf);
        // west column valid <=> 0 <= xi <= w-1 ; east <=> -1 <= xi <= w-2   (rows alike)
        const bool x0 = INTERIOR || (uint32_t)xi < (uint32_t)w, x1 = INTERIOR || (uint32_t)(xi + 1) < (uint32_t)w;
        const bool y0 = INTERIOR || (uint32_t)yi < (uint32_t)h, y1 = INTERIOR || (uint32_t)(yi + 1) < (uint32_t)h;
        const bool ok[4] = {x0 && y0, x1 && y0, x0 && y1, x1 && y1};
        f4 tv[4];
        if (INTERIOR || __builtin_expect(B.fits, 1)) {
            const int xl0 = xi - B.bx0, xl1 = xl0 + 1;
            const int cp0 = __mul24(xl0 & 3, cw16) + ((xl0 & ~3) << 2), cp1 = __mul24(xl1 & 3, cw16) + ((xl1 & ~3) << 2);
            const int yr = yi - B.miny;   // row in the sheared box, per tap column
            const int r0 = 16 + __mul24(yr - lds_shear(xi >> 2, B.sq), P16), r1 = 16 + __mul24(yr - lds_shear((xi + 1) >> 2, B.sq), P16);
            const int si[4] = {ok[0] ? r0 + cp0 : 0, ok[1] ? r1 + cp1 : 0, ok[2] ? r0 + P16 + cp0 : 0, ok[3] ? r1 + P16 + cp1 : 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) tv[j] = *reinterpret_cast<const f4*>(smem + si[j]);
        } else if (NC == 3) {
            // box too large for the LDS: gather from global memory, west + east tap of a row in ONE 8-byte load (the pair
            // starts at the column clamped to [0, w - 2]; each tap picks its element, invalid taps are zeroed)
            const int xc = min(max(xi, 0), w - 2);
            const int ew = xi - xc, ee = xi + 1 - xc;                  // element of the pair a valid west / east tap reads
            const int yr[2] = {min(max(yi, 0), h - 1), min(max(yi + 1, 0), h - 1)};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t og = (uint32_t)(yr[r] * w + xc);
                f4 tw = {0.f, 0.f, 0.f, 0.f}, te = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const f2 pr = ld2(sb + c * hw + og);
                    tw[c] = ew == 1 ? pr[1] : pr[0]; te[c] = ee == 1 ? pr[1] : pr[0];
                }
                if (VALID) {
                    const uint32_t m2 = sm ? ld16(sm + og) : 0x0101u;
                    tw[3] = ((ew == 1 ? m2 >> 8 : m2) & 0xffu) != 0u ? 1.0f : 0.0f;
                    te[3] = ((ee == 1 ? m2 >> 8 : m2) & 0xffu) != 0u ? 1.0f : 0.0f;
                }
                tv[2 * r] = ok[2 * r] ? tw : (f4){0.f, 0.f, 0.f, 0.f};
                tv[2 * r + 1] = ok[2 * r + 1] ? te : (f4){0.f, 0.f, 0.f, 0.f};
            }
        } else {
            // (flows: the pair-load variant above cost the 2-channel kernels 5 % on their staged path -- register
            // allocation -- so they keep one load per tap)
            const int cx[4] = {xi, xi + 1, xi, xi + 1}, cy[4] = {yi, yi, yi + 1, yi + 1};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t og = ok[j] ? (uint32_t)(cy[j] * w + cx[j]) : 0u;
                f4 t = {0.f, 0.f, 0.f, 0.f};
                if (ok[j]) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) t[c] = SUB ? ld1(sb + c * hw + og) - sbb[c * hw + og] : ld1(sb + c * hw + og);
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

template <int NC, bool VALID, bool SUB = false, typename TS = float>
__device__ __forceinline__ void lds_gather(const WarpParams& p, uint32_t hw,
                                           const TS* __restrict__ sb, const uint8_t* __restrict__ sm,
                                           const LdsCoords& T, const LdsBox& B, const unsigned char* smem, f4 (&outv)[4],
                                           const float* __restrict__ sbb = nullptr) {
    if (B.interior) lds_gather_impl<NC, VALID, true, SUB, TS>(p, hw, sb, sm, T, B, smem, outv, sbb);
    else lds_gather_impl<NC, VALID, false, SUB, TS>(p, hw, sb, sm, T, B, smem, outv, sbb);
}

// the fused addend of a tile (mode 3), loaded ahead of younger loads and stores: the wait for it must not cover them
template <int NC>
__device__ __forceinline__ void lds_load_addend(const WarpParams& p, int tx, int ty, int n, uint32_t hw, f4 (&a)[NC]) {
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const uint32_t pix = (uint32_t)(min(ty * kLdsTH + ly, p.h - 1) * p.w + min(tx * (kLdsTWQ * 4) + lx * 4, p.w - 4));
#pragma unroll
    for (int c = 0; c < NC; ++c) a[c] = ld4(p.addend + n * p.addend_bs + c * hw + pix);
}

// step 4b: valid mask, epilogue (a_sign * addend + g_sign * G, rounding), 16-byte stores
template <int NC, bool VALID, bool ADD, bool DF = false, typename TD = float>
__device__ __forceinline__ void lds_store(const WarpParams& p, int tx, int ty, int n, uint32_t hw, uint32_t fmask4,
                                          const f4 (&outv)[4], const f4 (&addend)[NC], int* dflags = nullptr) {
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const int w = p.w, h = p.h;
    const int x4 = tx * (kLdsTWQ * 4) + lx * 4, y = ty * kLdsTH + ly;
    const bool inb = (x4 < w) && (y < h);
    const uint32_t pix = (uint32_t)(min(y, h - 1) * w + min(x4, w - 4));
    if (inb) {
        uint32_t vo = 0x01010101u;
        if (VALID) {
            vo = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                vo |= (uint32_t)((outv[k][3] > kValidThr) && (((fmask4 >> (8 * k)) & 0xffu) != 0u)) << (8 * k);
            st32(p.valid + (int64_t)n * hw + pix, vo);
        }
        TD* __restrict__ db = reinterpret_cast<TD*>(p.dst) + (int64_t)n * p.dst_bs;
        f4 o01[2];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f4 o = {outv[0][c], outv[1][c], outv[2][c], outv[3][c]};
            if (ADD) o = addend[c] * p.a_sign + o * p.g_sign;
            if (p.round_mode != OFL_ROUND_NONE) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = apply_round(o[k], p.round_mode);
            }
            st4(db + c * hw + pix, o);
            if (DF && c < 2) o01[c] = o;
        }
        if (DF && NC == 2) {                               // flag word of the OUTPUT read as a flow under `valid` (by-product)
#pragma unroll
            for (int k = 0; k < 4; ++k) *dflags |= flag_bits(o01[0][k], o01[NC - 1][k], ((vo >> (8 * k)) & 0xffu) != 0u);
        }
    }
}

template <int NC, bool VALID, bool ADD, bool DF = false, bool SUB = false, typename TS = float, typename TD = float>
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
    const TS* __restrict__ sb = reinterpret_cast<const TS*>(p.src) + n * p.src_bs;      // (uint8 variants: p.src / p.dst point at bytes)
    const float* __restrict__ sbb = SUB ? p.src_b + n * p.src_b_bs : nullptr;
    const uint8_t* __restrict__ sm = p.src_mask ? p.src_mask + n * p.src_mask_bs : nullptr;
    const uint8_t* __restrict__ fm = p.flow_mask ? p.flow_mask + n * p.flow_mask_bs : nullptr;
    const int x4 = tx * (kLdsTWQ * 4) + lx * 4, xq = min(x4, w - 4);
    const uint32_t pixA = (uint32_t)(min(tyA * kLdsTH + ly, h - 1) * w + xq);
    const uint32_t pixB = (uint32_t)(min(tyB * kLdsTH + ly, h - 1) * w + xq);
    const f4 uA = ld4nt(fu + pixA), vA = ld4nt(fu + hw + pixA);
    const f4 uB = ld4nt(fu + pixB), vB = ld4nt(fu + hw + pixB);
    uint32_t fmA = 0x01010101u, fmB = 0x01010101u;
    if ((VALID || p.flow_flags) && fm) {
        fmA = ld32(fm + pixA);
        fmB = ld32(fm + pixB);
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
        if ((tid & 63) == 0) flag_or(&p.flow_flags[n], f);
    }
    f4* lds = reinterpret_cast<f4*>(smem);
    LdsCoords TA, TB;
    LdsBox BA, BB;
    LdsStage<NC> S;
    const int sq = p.shear ? lds_slope(p, fu, hw, tx, ty2) : 0;
    lds_coords_box(p, tx, tyA, uA, vA, sq, TA, BA, red[0]);
    lds_issue<NC, VALID, SUB, TS>(p, sb, sm, hw, BA, S, sbb);   // staging loads of A fly ...
    lds_coords_box(p, tx, tyB, uB, vB, sq, TB, BB, red[1]);     // ... while B's coordinates are computed
    lds_write<NC, VALID>(lds, BA, S);
    lds_barrier();
    // the vmcnt queue is in order: the addend (an L2 hit when it is the flow itself) is fetched BEFORE B's staging loads /
    // A's stores, so that waiting for it never waits for them
    // (flows only: with three channels the extra registers would spill, and nothing on the host adds to an image)
    constexpr bool EARLY = ADD && NC <= 2;
    f4 outv[4], aA[NC], aB[NC];
    const bool reuse = EARLY && NC == 2 && p.add_is_flow;         // block-uniform
    if (EARLY) { if (reuse) { aA[0] = uA; aA[NC - 1] = vA; } else lds_load_addend<NC>(p, tx, tyA, n, hw, aA); }
    if (haveB) lds_issue<NC, VALID, SUB, TS>(p, sb, sm, hw, BB, S, sbb);   // staging loads of B fly while A is gathered and stored
    lds_gather<NC, VALID, SUB, TS>(p, hw, sb, sm, TA, BA, smem, outv, sbb);
    if (EARLY && haveB) { if (reuse) { aB[0] = uB; aB[NC - 1] = vB; } else lds_load_addend<NC>(p, tx, tyB, n, hw, aB); }
    if (ADD && !EARLY) lds_load_addend<NC>(p, tx, tyA, n, hw, aA);
    int dflags = 0;
    lds_store<NC, VALID, ADD, DF, TD>(p, tx, tyA, n, hw, fmA, outv, aA, &dflags);
    if (!haveB) {
        if (DF) { dflags = wave_or_flags(dflags); if ((tid & 63) == 0) flag_or(&p.dst_flags[n], dflags); }
        return;
    }
    lds_barrier();
    lds_write<NC, VALID>(lds, BB, S);
    lds_barrier();
    lds_gather<NC, VALID, SUB, TS>(p, hw, sb, sm, TB, BB, smem, outv, sbb);
    if (ADD && !EARLY) lds_load_addend<NC>(p, tx, tyB, n, hw, aB);
    lds_store<NC, VALID, ADD, DF, TD>(p, tx, tyB, n, hw, fmB, outv, aB, &dflags);
    if (DF) { dflags = wave_or_flags(dflags); if ((tid & 63) == 0) flag_or(&p.dst_flags[n], dflags); }
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
        if (lane == 0) flag_or(&p.flow_flags[n], fflags);
        if (p.src_flags) {
            sflags = wave_or_flags(sflags);
            if (lane == 0) flag_or(&p.src_flags[n], sflags);
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
    const float* data_b; int64_t data_b_bs;      // optional: the data is data - data_b (one fp32 subtraction, as `flow - self` in the reference's modes 1-2)
    const uint8_t* weight_mask; int64_t weight_mask_bs;
    const uint8_t* chan_mask_a; int64_t chan_mask_a_bs;
    const uint8_t* chan_mask_b; int64_t chan_mask_b_bs;
    int32_t with_mask_chan, occlude;
    float* accum;          // pass 1 out / pass 2 in
    float* dst; float* density; uint8_t* warped; uint8_t* valid; float* mask_chan;   // pass 2 out
    int64_t dst_bs;        // batch stride of dst (channels of the whole output tensor * h * w)
    int32_t n, c, h, w;
    int32_t round_mode;
    int32_t tiles_x, tiles_y;
    int64_t total_tiles, per_xcd;
    const int32_t* run_if_set;   // optional device flag: the atomics path runs only when *run_if_set != 0
    int32_t* dst_flags;          // optional int32[N] (2-channel data only): flag word of the OUTPUT read as a flow under `valid`
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
    const float* __restrict__ dbb = p.data_b ? p.data_b + n * p.data_b_bs : nullptr;
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
                        atomicAdd(&acc[(int64_t)(1 + cc) * hw + pos], wgt * (p.data_sign * (dbb ? db[(int64_t)cc * hw + pix] - dbb[(int64_t)cc * hw + pix] : db[(int64_t)cc * hw + pix])));
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
    const float* __restrict__ dbb = p.data_b ? p.data_b + n * p.data_b_bs : nullptr;
    const uint8_t* __restrict__ wmk = p.weight_mask ? p.weight_mask + n * p.weight_mask_bs : nullptr;
    const uint8_t* __restrict__ cma = p.chan_mask_a ? p.chan_mask_a + n * p.chan_mask_a_bs : nullptr;
    const uint8_t* __restrict__ cmb = p.chan_mask_b ? p.chan_mask_b + n * p.chan_mask_b_bs : nullptr;
    const float* __restrict__ acc = p.accum + (int64_t)n * planes * hw;
    float* __restrict__ dst = p.dst + (int64_t)n * p.dst_bs;
    int dflags = 0;

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
        float uv[2] = {0.0f, 0.0f};
#pragma unroll
        for (int ch = 0; ch < (CT ? CT : 1); ++ch)
            for (int cc = ch; cc < C; cc += (CT ? C : 1)) {
                float val = fill ? p.data_sign * (dbb ? db[(int64_t)cc * hw + pix] - dbb[(int64_t)cc * hw + pix] : db[(int64_t)cc * hw + pix]) : acc[(int64_t)(1 + cc) * hw + pix] / dcl;
                val = apply_round(val, p.round_mode);
                dst[(int64_t)cc * hw + pix] = val;
                if (cc < 2) uv[cc] = val;
            }
        if (p.density) p.density[(int64_t)n * hw + pix] = den;
        if (p.warped) p.warped[(int64_t)n * hw + pix] = (uint8_t)warped;
        bool vld = true;
        if (p.valid || p.mask_chan) {
            float mch;
            if (fill)
                mch = ((cma ? cma[pix] != 0 : true) && (cmb ? cmb[pix] != 0 : true)) ? 1.0f : 0.0f;
            else
                mch = (den - acc[(int64_t)(1 + C) * hw + pix]) / dcl;
            vld = mch > kValidThr;
            if (p.valid) p.valid[(int64_t)n * hw + pix] = (uint8_t)vld;
            if (p.mask_chan) p.mask_chan[(int64_t)n * hw + pix] = mch;
        }
        if (p.dst_flags) dflags |= flag_bits(uv[0], uv[1], vld);
    }
    if (p.dst_flags) {                                   // (block-uniform)
        dflags = wave_or_flags(dflags);
        if (lane == 0) flag_or(&p.dst_flags[n], dflags);
    }
}


// ------------------------------------------------------------------------------------------------
// forward splat, routed fast path (ofl_splat_tiled_f32): sort by destination tile, then exact per-tile accumulation
//
//  route kernel: one block per 32 x 16 SOURCE tile.  End points of its pixels; every pixel goes, as a record of
//     12 + 4 C bytes (end point x, y | raster key with the mask-channel bit below it | data x data_sign), to the queue
//     of each DESTINATION tile one of its four corners falls into (1.1 queues per pixel on smooth flows).  Ranks come
//     from LDS integer atomics per local destination tile, then ONE global atomic per (source tile, destination tile) on
//     the queue's length.  Queues have fixed addresses -- no sizing pass, no scan: 1024 records per tile, then blocks
//     of 1024 drawn from a shared pool on demand (one compare-and-swap per slot of 1024 queue positions).
//  tile kernel : one block per 32 x 16 DESTINATION tile.
//     A  its queue -> LDS, 16-byte loads;
//     B  every record is pushed on the list of its CELL (the unit square floor(x), floor(y) of its end point): one LDS
//        atomic exchange.  The four corner classes of destination pixel (X, Y) are the cells (X - kx, Y - ky), so one
//        list per cell serves them all;
//     S  every cell is put in raster order of its source pixels once: up to 4 records by a sorting network in registers
//        (written as four 16-bit slots), 5 .. 64 by an insertion sort of the list itself;
//     C  each thread sums its own 2 destination pixels in registers: its 3 x 2 cells, every record fetched once and
//        added, in list order, to each corner-class sum it belongs to, then ((c0 + c1) + c2) + c3 -- exactly the order
//        of the reference's four scatter_add_ passes and its corner sum (utils.py:1133-1143), products rounded before
//        they are added: BIT-IDENTICAL to the reference, and run to run; normalise, threshold, un-occlude, store (and,
//        for flows, the output's flag word as a by-product).
//     A queue longer than the LDS records (1024) is processed in 2 or 4 bands of destination rows, each band compacting
//     the records that touch it.
//  No float atomics, no accumulator in HBM.
//  Only a heavy fold of the flow (> 64 sources in one cell, or more records for one tile than four bands hold) makes
//  THAT tile fall back to LDS float atomics over the same queue (tolerance instead of bit-exactness for that tile).
//  The launch-level two-pass path only runs for input the queues cannot hold (> 9216 records for one tile, or more
//  blocks drawn than the one per two tiles provisioned) or source tiles that spread over > 48 destination tiles.
// ------------------------------------------------------------------------------------------------
#ifndef OFL_SP_TH
#define OFL_SP_TH 16
#endif
constexpr int kSpTW = 32, kSpTH = OFL_SP_TH;                 // source and destination tiles
constexpr int kSpNT = kSpTW * kSpTH / 4;                     // route kernel: 4 source pixels per thread
constexpr int kSpNT2 = kSpTW * kSpTH / 2;                    // tile kernel: 2 destination pixels per thread
#ifndef OFL_SP_Q
#define OFL_SP_Q (64 * OFL_SP_TH)
#endif
constexpr int kSpQ = OFL_SP_Q;    // records the tile kernel holds in LDS at a time (1024 measured faster than 768 + one more block per CU)
constexpr int kSpRouteMax = 48;   // destination tiles one source tile may feed
// Queues have fixed addresses (no sizing pass): every destination tile owns kSpPrim records; the records beyond them go to
// blocks of kSpPrim records drawn from a shared pool on demand, one per slot of kSpPrim queue positions (up to kSpSlots
// per tile; one block per 2 tiles is provisioned).  More than (1 + kSpSlots) * kSpPrim records for a tile (18 per
// pixel), or more draws than blocks, send the launch to the two-pass path.
constexpr int kSpPrim = OFL_SP_Q, kSpSlots = 8, kSpSecDiv = 2;
#ifndef OFL_SP_LONG
#define OFL_SP_LONG 64
#endif
constexpr int kSpLong = OFL_SP_LONG;   // longest cell list (source pixels whose end points share one unit cell) that is summed in raster order

__device__ __forceinline__ uint32_t nz_bytes(uint32_t x) {   // per byte: non-zero -> 0x01
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) r |= (uint32_t)(((x >> (8 * k)) & 0xffu) != 0u) << (8 * k);
    return r;
}

struct TiledParams {
    SplatParams s;
    int32_t* cursor;       // [n * tiles] records routed to the tile so far (its queue length once the route kernel is done)
    int32_t* sec;          // [n * tiles][kSpSlots] block that holds queue positions kSpPrim * (1 + slot) .. of the tile, -1 = none
    float* prim;           // [n * tiles][3 + C][kSpPrim]: end point x | end point y | raster key + mask-channel bit | data ...
    float* secp;           // [nsec][3 + C][kSpPrim]: the blocks
    int32_t nsec;
    int32_t* sec_count;    // [1] secondary blocks drawn in this pass
    int32_t* overflow;     // [4]: launch falls back | tiles that left the exact path | - | -
    int32_t tiles_x, tiles_y;
    uint32_t tiles_img, mx_m, mx_s, mi_m, mi_s;
    int64_t total, per_xcd;
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

// end point + contribution test of the 4 source pixels of a thread
struct SpSrc { float x[4], y[4]; bool on[4]; bool zero[4]; bool wm[4]; };

__device__ __forceinline__ void sp_load_src(const SplatParams& s, int n, int sx4, int sy, bool inimg, uint32_t pix, uint32_t hw, SpSrc& q) {
    f4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    uint32_t wm4 = 0x01010101u;
    // the last group of a row of an image whose width is not a multiple of 4: fetch the last whole group and rotate
    const int wrem = s.w & 3;
    const bool edge = wrem != 0 && inimg && sx4 > s.w - 4;
    const uint32_t pe = edge ? pix - (uint32_t)(4 - wrem) : pix;
    if (inimg) {
        if (s.flow) {
            a = ld4(s.flow + n * s.flow_bs + pe);
            b = ld4(s.flow + n * s.flow_bs + hw + pe);
        } else {
            a = ld4(s.xs + n * s.xy_bs + pe);
            b = ld4(s.ys + n * s.xy_bs + pe);
        }
        if (s.weight_mask) wm4 = ld32(s.weight_mask + n * s.weight_mask_bs + pe);
        if (wrem != 0) {
            if (edge) { a = rot4(a, 4 - wrem); b = rot4(b, 4 - wrem); wm4 >>= 8 * (4 - wrem); }
        }
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
        q.on[k] = inimg && (sx4 + k < s.w) && q.wm[k] && !q.zero[k];
    }
}

__global__ __launch_bounds__(kSpNT) void splat_route_kernel(const TiledParams p) {
    __shared__ int red[kSpNT / 64][4];
    __shared__ int lcount[kSpRouteMax], lbase[kSpRouteMax], lsec[kSpRouteMax], lsec1[kSpRouteMax], lslot[kSpRouteMax], ltile[kSpRouteMax];
    if (*p.overflow != 0) return;
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
    if (tid < kSpRouteMax) lcount[tid] = 0;
    // destination tile columns / rows of the (at most two) in-image corner columns / rows of every pixel; -1: none
    int tca[4], tcb[4], tra[4], trb[4];
    int minx = 0x7fffffff, maxx = -0x7fffffff, miny = 0x7fffffff, maxy = -0x7fffffff;
    const float wf = (float)w, hf = (float)h;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        tca[k] = tcb[k] = tra[k] = trb[k] = -1;
        if (q.on[k]) {
            // a corner outside the image is clamped by the reference and carries weight 0 (utils.py:1106-1111)
            const int x0 = (int)__builtin_amdgcn_fmed3f(floorf(q.x[k]), -2.0f, wf), y0 = (int)__builtin_amdgcn_fmed3f(floorf(q.y[k]), -2.0f, hf);
            if ((uint32_t)x0 < (uint32_t)w) tca[k] = x0 / kSpTW;
            if ((uint32_t)(x0 + 1) < (uint32_t)w) tcb[k] = (x0 + 1) / kSpTW;
            if ((uint32_t)y0 < (uint32_t)h) tra[k] = y0 / kSpTH;
            if ((uint32_t)(y0 + 1) < (uint32_t)h) trb[k] = (y0 + 1) / kSpTH;
            if (tcb[k] == tca[k]) tcb[k] = -1;
            if (trb[k] == tra[k]) trb[k] = -1;
            if (tca[k] < 0) { tca[k] = tcb[k]; tcb[k] = -1; }
            if (tra[k] < 0) { tra[k] = trb[k]; trb[k] = -1; }
            if (tca[k] >= 0 && tra[k] >= 0) {
                minx = min(minx, tca[k]); maxx = max(maxx, max(tca[k], tcb[k]));
                miny = min(miny, tra[k]); maxy = max(maxy, max(tra[k], trb[k]));
            }
        }
    }
    minx = wave_min_dpp(minx); maxx = wave_max_dpp(maxx); miny = wave_min_dpp(miny); maxy = wave_max_dpp(maxy);
    if ((tid & 63) == 0) { red[tid >> 6][0] = minx; red[tid >> 6][1] = maxx; red[tid >> 6][2] = miny; red[tid >> 6][3] = maxy; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSpNT / 64; ++i) {
        minx = min(minx, red[i][0]); maxx = max(maxx, red[i][1]); miny = min(miny, red[i][2]); maxy = max(maxy, red[i][3]);
    }
    minx = __builtin_amdgcn_readfirstlane(minx); maxx = __builtin_amdgcn_readfirstlane(maxx);
    miny = __builtin_amdgcn_readfirstlane(miny); maxy = __builtin_amdgcn_readfirstlane(maxy);
    if (maxx < minx || maxy < miny) return;                      // nothing of this tile lands inside the image
    const int ntx = maxx - minx + 1, nt = ntx * (maxy - miny + 1);
    if (nt > kSpRouteMax) { if (tid == 0) atomicOr(p.overflow, 1); return; }
    // local rank of every (pixel, destination tile) pair: LDS atomics; packed (local tile << 12 | rank), ~0 = none
    uint32_t pr[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tc = (j & 1) ? tcb[k] : tca[k], tr = (j & 2) ? trb[k] : tra[k];
            pr[k][j] = 0xffffffffu;
            if (tc >= 0 && tr >= 0) {
                const int lt = (tr - miny) * ntx + (tc - minx);
                pr[k][j] = ((uint32_t)lt << 12) | (uint32_t)atomicAdd(&lcount[lt], 1);
            }
        }
    }
    __syncthreads();
    if (tid < nt && lcount[tid]) {
        const int d = n * (int)p.tiles_img + (miny + tid / ntx) * p.tiles_x + (minx + tid % ntx);
        const int cnt = lcount[tid];
        const int start = atomicAdd(&p.cursor[d], cnt);          // this block's records are start .. start + cnt - 1 of the queue
        // Records beyond the primary region go to the block of their slot of kSpPrim queue positions.  Exactly ONE block is
        // drawn per slot, whoever comes first: the first arrival swaps the slot's entry from -1 (none) to -2 (being
        // drawn), draws and publishes; everybody else waits for the published id.  (Drawing first and swapping after
        // would leak a block per lost race -- and make running out of blocks, hence the choice of path and the last bits
        // of the result, depend on timing.)  All of a wave's draws are published before any of its lanes waits, and a wait
        // is always for a wave that is already past its own draws: no cycle.
        int b0 = -1, b1 = -1, s0 = 0, s1 = 0;
        const bool need = start + cnt > kSpPrim && start + cnt <= (1 + kSpSlots) * kSpPrim;
        if (start + cnt > (1 + kSpSlots) * kSpPrim) atomicOr(p.overflow, 1);
        if (need) {
            s0 = (max(start, kSpPrim) - kSpPrim) / kSpPrim;
            s1 = (start + cnt - 1 - kSpPrim) / kSpPrim;                      // (cnt <= 512: at most two slots)
            // (the entry's value is all that is communicated: relaxed accesses served by L2 are enough)
            for (int slot = s0; slot <= s1; ++slot) {
                int32_t* e = &p.sec[(int64_t)d * kSpSlots + slot];
                int cur = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == -1 && atomicCAS(e, -1, -2) == -1) {
                    const int mine = atomicAdd(p.sec_count, 1);
                    if (mine >= p.nsec) atomicOr(p.overflow, 1);
                    cur = mine < p.nsec ? mine : -3;                         // (-3: none left)
                    __hip_atomic_store(e, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (slot == s0) b0 = cur; else b1 = cur;
            }
            if (s1 == s0) b1 = b0;
        }
        __builtin_amdgcn_wave_barrier();                                     // (the waits stay behind the draws)
        if (need) {
            while (b0 == -1 || b0 == -2) b0 = __hip_atomic_load(&p.sec[(int64_t)d * kSpSlots + s0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s1 == s0) b1 = b0;
            while (b1 == -1 || b1 == -2) b1 = __hip_atomic_load(&p.sec[(int64_t)d * kSpSlots + s1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lbase[tid] = start; lsec[tid] = b0; lsec1[tid] = b1; lslot[tid] = s0; ltile[tid] = d;
    }
    __syncthreads();
    // the records carry the pixel's data (x data_sign) and mask channel: the tile kernel reads everything coalesced
    const int nc = s.c;
    f4 dat[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    uint32_t mc4 = 0x01010101u;
    if (inimg) {
        const int wrem = w & 3;
        const bool edge = wrem != 0 && sx4 > w - 4;                   // row-end group: last whole group, rotated
        const uint32_t pix = (uint32_t)(sy * w + sx4) - (edge ? (uint32_t)(4 - wrem) : 0u);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < nc) {
                dat[c] = ld4(s.data + n * s.data_bs + c * hw + pix);
                if (s.data_b) dat[c] = dat[c] - ld4(s.data_b + n * s.data_b_bs + c * hw + pix);
            }
        uint32_t ma = 0x01010101u, mb = 0x01010101u;
        if (s.with_mask_chan) {
            if (s.chan_mask_a) ma = ld32(s.chan_mask_a + n * s.chan_mask_a_bs + pix);
            if (s.chan_mask_b) mb = ld32(s.chan_mask_b + n * s.chan_mask_b_bs + pix);
        }
        if (wrem != 0) {
            if (edge) {
#pragma unroll
                for (int c = 0; c < 3; ++c) dat[c] = rot4(dat[c], 4 - wrem);
                ma >>= 8 * (4 - wrem); mb >>= 8 * (4 - wrem);
            }
        }
        mc4 = nz_bytes(ma) & nz_bytes(mb);
    }
    const int ncol = 3 + nc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (pr[k][j] != 0xffffffffu) {
                const int lt = (int)(pr[k][j] >> 12), idx = lbase[lt] + (int)(pr[k][j] & 0xfffu);
                float* rp; int cs;                               // column a of this record: rp[a * cs]
                if (idx < kSpPrim) {
                    rp = p.prim + ((int64_t)ltile[lt] * ncol) * kSpPrim + idx; cs = kSpPrim;
                } else {
                    const int o = idx - kSpPrim, sb = (o / kSpPrim == lslot[lt]) ? lsec[lt] : lsec1[lt];
                    if (sb < 0) continue;                                 // (the launch is flagged: the two-pass path redoes it)
                    rp = p.secp + ((int64_t)sb * ncol) * kSpPrim + (o % kSpPrim); cs = kSpPrim;
                }
                rp[0] = q.x[k]; rp[cs] = q.y[k];
                // key: raster position of the source pixel (15 bits each, checked by ofl_splat_tiled_f32) with the mask-channel bit below it --
                // two records never share a position, so ordering by the whole word is raster order
                rp[2 * cs] = __uint_as_float(((((uint32_t)sy << 15) | (uint32_t)(sx4 + k)) << 1) | ((mc4 >> (8 * k)) & 1u));
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < nc) rp[(3 + c) * cs] = s.data_sign * dat[c][k];
            }
        }
    }
}

// weights and destination-local corner positions of one end point, exactly as the reference (utils.py:1098-1114)
__device__ __forceinline__ void sp_corners(float xv, float yv, float wmax, float hmax, int dx0, int dy0,
                                           float (&wx)[2], float (&wy)[2], int (&ix)[2], int (&iy)[2]) {
    const float x0 = floorf(xv), y0 = floorf(yv), x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    const float x0s = fminf(fmaxf(x0, 0.0f), wmax), x1s = fminf(fmaxf(x1, 0.0f), wmax);
    const float y0s = fminf(fmaxf(y0, 0.0f), hmax), y1s = fminf(fmaxf(y1, 0.0f), hmax);
    wx[0] = (x1 - xv) * (x0 == x0s ? 1.0f : 0.0f); wx[1] = (xv - x0) * (x1 == x1s ? 1.0f : 0.0f);
    wy[0] = (y1 - yv) * (y0 == y0s ? 1.0f : 0.0f); wy[1] = (yv - y0) * (y1 == y1s ? 1.0f : 0.0f);
    ix[0] = (int)x0s - dx0; ix[1] = (int)x1s - dx0; iy[0] = (int)y0s - dy0; iy[1] = (int)y1s - dy0;
}

#ifndef OFL_SP_MINB
#define OFL_SP_MINB 4   // blocks per CU the tile kernel's register budget is sized for (LDS allows 4; measured +9 % over 3)
#endif
// One record of the cell whose column is DC (-1, 0, +1) cells from the pair's middle cell and whose row serves corner
// row KY, added to the sums of the destination pixels that read it: pixel 0 of the pair as its x-corner 1 (DC = -1) or
// 0 (DC = 0), pixel 1 as its x-corner 1 (DC = 0) or 0 (DC = +1).  The corner is not clamped (the destination is inside
// the image), so the reference's weight is (x1 - x | x - x0) * 1 (utils.py:1110-1114); product rounded, then added.
template <int NC, int NCH, int DC, int KY>
__device__ __forceinline__ void sp_use(const float* rec, uint32_t i, float (&a)[2][2][1 + NCH]) {
    const float xv = rec[i], yv = rec[kSpQ + i];
    const float x0 = floorf(xv), y0 = floorf(yv);
    const float wyk = (KY ? yv - y0 : (y0 + 1.0f) - yv) * 1.0f;
    float d[NCH];
#pragma unroll
    for (int c = 0; c < NC; ++c) d[c] = rec[(3 + c) * kSpQ + i];
    if (NCH > NC) d[NC] = (float)(__float_as_uint(rec[2 * kSpQ + i]) & 1u);   // the mask channel rides in the key
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int kx = k - DC;                     // pixel k sits DC .. DC + 1 columns right of the cell: x-corner k - DC
        if (kx < 0 || kx > 1) continue;
        const float wxk = (kx ? xv - x0 : (x0 + 1.0f) - xv) * 1.0f;
        const float wgt = wyk * wxk;
        a[k][kx][0] += wgt;
#pragma unroll
        for (int c = 0; c < NCH; ++c) a[k][kx][1 + c] += wgt * d[c];
    }
}

template <int NC, bool MCH>
__global__ __launch_bounds__(kSpNT2, OFL_SP_MINB) void splat_tile_kernel(const TiledParams p) {
    constexpr int kPx = kSpTW * kSpTH, NCH = NC + (MCH ? 1 : 0), NREC = 3 + NC, kRounds = kSpQ / kSpNT2;
    constexpr uint32_t kEnd = 0xffffu, kLongCell = 0xfffeu;
    // A CELL is a unit square of the destination grid: the records whose end point has floor(x, y) = (cx, cy).  The four
    // corner classes of a destination pixel (X, Y) are the cells (X - kx, Y - ky), so one list per cell serves them all:
    // (kSpTW + 1) x (kSpTH + 1) cells per tile, the first column / row being the cells left of / above the tile.
    constexpr int kCW = kSpTW + 1, kCH = kSpTH + 1, kCells = kCW * kCH, kCellsP = (kCells + 63) / 64 * 64;
    constexpr int kCellRounds = (kCellsP + kSpNT2 - 1) / kSpNT2;
    // LDS: records [x | y | key + mask-channel bit | data ...][kSpQ] | cell list heads | sorted cell slots | list links
    // (the float-atomics fallback re-uses the record area as accumulator planes)
    __shared__ __attribute__((aligned(16))) unsigned char raw[kSpQ * 4 * NREC + kCellsP * 4 + kCellsP * 8 + kSpQ * 2];
    __shared__ int qcount;
    static_assert(kSpQ * NREC >= (1 + NCH) * kPx, "fallback planes must fit the record area");
    float* rec = reinterpret_cast<float*>(raw);                       // rec[a * kSpQ + i]
    const float* rx = rec; const float* ry = rec + kSpQ;
    const uint32_t* rkey = reinterpret_cast<const uint32_t*>(rec + 2 * kSpQ);
    uint32_t* head = reinterpret_cast<uint32_t*>(rec + NREC * kSpQ);  // [kCellsP]: newest record of the cell, kEnd = empty
    uint2* slots = reinterpret_cast<uint2*>(head + kCellsP);          // [kCellsP]: up to 4 records in raster order, 16 bits each
    uint16_t* link = reinterpret_cast<uint16_t*>(slots + kCellsP);    // [kSpQ]: next record of the same cell
    float* acc = rec;                                                 // fallback: [1 + NCH][kPx]
    int tx, ty, n;
    if (!sp_decode(p, tx, ty, n)) return;
    const SplatParams& s = p.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    const int dx0 = tx * kSpTW, dy0 = ty * kSpTH;
    const float wmax = (float)(w - 1), hmax = (float)(h - 1);
    const int64_t dtile = (int64_t)n * p.tiles_img + ty * p.tiles_x + tx;
    const int launch_over = *p.overflow, qlen = p.cursor[dtile];
    if (launch_over != 0) return;                                     // this launch takes the global-atomics path instead
    const float* __restrict__ gq = p.prim + (dtile * NREC) * kSpPrim;  // column a of the primary region: gq[a * kSpPrim + i], 16-byte aligned
    // queue positions kSpPrim .. of a long queue live in blocks of kSpPrim records (one more round trip, long queues only)
    __shared__ int qblk[kSpSlots];
    if (qlen > kSpPrim) {
        if (tid < kSpSlots) qblk[tid] = p.sec[dtile * kSpSlots + tid];
        __syncthreads();
    }
    // record i of the queue, column a
    auto qrec = [&](int a, int i) -> float {
        if (i < kSpPrim) return gq[a * kSpPrim + i];
        const int o = i - kSpPrim;
        return p.secp[((int64_t)qblk[o / kSpPrim] * NREC + a) * kSpPrim + (o % kSpPrim)];
    };
    const float* __restrict__ db = s.data + n * s.data_bs;
    const float* __restrict__ dbb = s.data_b ? s.data_b + n * s.data_b_bs : nullptr;
    const uint8_t* __restrict__ cma = s.chan_mask_a ? s.chan_mask_a + n * s.chan_mask_a_bs : nullptr;
    const uint8_t* __restrict__ cmb = s.chan_mask_b ? s.chan_mask_b + n * s.chan_mask_b_bs : nullptr;
    // this thread's 2 destination pixels
    const int lx = tid & 15, ly = tid >> 4;
    const int x2 = min(dx0 + lx * 2, w - 2), y = dy0 + ly;           // (odd widths: the last pair re-computes pixel w - 2)
    const int lx2 = x2 - dx0;                                        // tile-local column of the pair (may be lx * 2 - 1)
    const bool solo = lx2 < 0;                                       // a tile that is one pixel wide: only the pair's second pixel is its own
    const bool inimg = (dx0 + lx * 2 < w) && (y < h);
    const uint32_t pix = (uint32_t)(min(y, h - 1) * w + x2);
    bool fill_ok[2] = {false, false};                    // un-occlude fill candidates (utils.py:1198-1203)
    if (s.occlude && s.flow && inimg) {
        const f2 a = ld2(s.flow + n * s.flow_bs + pix), b = ld2(s.flow + n * s.flow_bs + hw + pix);
        uint32_t wm2 = 0x0101u;
        if (s.weight_mask) wm2 = ld16(s.weight_mask + n * s.weight_mask_bs + pix);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            fill_ok[k] = (a[k] < kZeroThr) && (a[k] > -kZeroThr) && (b[k] < kZeroThr) && (b[k] > -kZeroThr) && (((wm2 >> (8 * k)) & 0xffu) != 0u);
    }
    int dflags = 0;
    auto flush_flags = [&]() {                            // (every thread of the block gets here)
        if (NC == 2 && s.dst_flags) {
            dflags = wave_or_flags(dflags);
            if (lane == 0) flag_or(&s.dst_flags[n], dflags);
        }
    };
    // normalise, masks, un-occlude fill, store (tot: density, channels, mask channel)
    auto finalize = [&](const float (&tot)[2][1 + NCH]) {
        f2 den2, out[NC], mch2;
        uint32_t warped2 = 0, valid2 = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float den = tot[k][0];
            const float dcl = den < kDenMin ? kDenMin : den;          // clamp_min utils.py:1144
            const bool warped = den > 0.0f;                            // utils.py:1197
            const bool fill = fill_ok[k] && !warped;
            den2[k] = den;
            warped2 |= (uint32_t)warped << (8 * k);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                out[c][k] = apply_round(fill ? s.data_sign * ((NC <= 2 && dbb) ? db[c * hw + pix + k] - dbb[c * hw + pix + k] : db[c * hw + pix + k]) : tot[k][1 + c] / dcl, s.round_mode);
            if (MCH) {
                float mv;
                if (fill) {
                    const bool a = cma ? cma[pix + k] != 0 : true, b = cmb ? cmb[pix + k] != 0 : true;
                    mv = (a && b) ? 1.0f : 0.0f;
                } else {
                    mv = tot[k][1 + NC] / dcl;
                }
                mch2[k] = mv;
                valid2 |= (uint32_t)(mv > kValidThr) << (8 * k);
            }
        }
        if (NC == 2 && s.dst_flags) {                     // the output read as a flow under its valid mask (by-product)
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (k == 1 || !solo) dflags |= flag_bits(out[0][k], out[NC - 1][k], MCH ? ((valid2 >> (8 * k)) & 1u) != 0u : true);
        }
        float* __restrict__ dst = s.dst + (int64_t)n * s.dst_bs;
        if (!solo) {
#pragma unroll
            for (int c = 0; c < NC; ++c) st2(dst + c * hw + pix, out[c]);
            if (s.density) st2(s.density + (int64_t)n * hw + pix, den2);
            if (s.warped) st16(s.warped + (int64_t)n * hw + pix, warped2);
            if (MCH && s.valid) st16(s.valid + (int64_t)n * hw + pix, valid2);
            if (MCH && s.mask_chan) st2(s.mask_chan + (int64_t)n * hw + pix, mch2);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) dst[c * hw + pix + 1] = out[c][1];
            if (s.density) s.density[(int64_t)n * hw + pix + 1] = den2[1];
            if (s.warped) s.warped[(int64_t)n * hw + pix + 1] = (uint8_t)(warped2 >> 8);
            if (MCH && s.valid) s.valid[(int64_t)n * hw + pix + 1] = (uint8_t)(valid2 >> 8);
            if (MCH && s.mask_chan) s.mask_chan[(int64_t)n * hw + pix + 1] = mch2[1];
        }
    };
    // bands of destination rows: 1 when the whole queue fits the LDS records
    int nb = 1;
    if (qlen > kSpQ) {                                    // a band of r rows sees about (r + 1) / 16 of the records
        nb = 2;
        if (qlen * (kSpTH / 2 + 1) > (kSpQ - kSpQ / 8) * kSpTH) nb = 4;
        if (qlen * (kSpTH / 4 + 1) > (kSpQ - kSpQ / 8) * kSpTH) nb = 0;   // a fold: straight to the float atomics
    }
    const int rows = nb ? kSpTH / nb : kSpTH;
    bool over = nb == 0;
    for (int band = 0; band < nb; ++band) {
        const int r0 = band * rows, r1 = r0 + rows;
#pragma unroll
        for (int i = 0; i < kCellRounds; ++i)
            if (tid + i * kSpNT2 < kCellsP) head[tid + i * kSpNT2] = kEnd;
        int nrec = qlen;
        if (nb == 1) {
            // ---- A (whole queue): coalesced 16-byte copy into LDS (the queue is padded to whole groups of 4 records)
            static_assert(kSpQ <= kSpNT2 * 4, "one 16-byte group per thread");
            if (tid * 4 < qlen) {
#pragma unroll
                for (int a = 0; a < NREC; ++a)
                    *reinterpret_cast<f4*>(rec + a * kSpQ + tid * 4) = *reinterpret_cast<const f4*>(gq + a * kSpPrim + tid * 4);
            }
        } else {
            // ---- A (band): compact the records with a corner row inside the band
            if (tid == 0) qcount = 0;
            __syncthreads();
            for (int base = 0; base < qlen; base += kSpNT2) {
                float col[NREC];
                const int i = base + tid;
                bool hit = i < qlen;
                if (hit) {
#pragma unroll
                    for (int a = 0; a < NREC; ++a) col[a] = qrec(a, i);
                    const int y0 = (int)__builtin_amdgcn_fmed3f(floorf(col[1]), -2.0f, (float)h) - dy0;
                    hit = (y0 >= r0 - 1) && (y0 < r1);
                }
                const unsigned long long m = __ballot(hit);
                if (m != 0ull) {                                 // wave-uniform
                    int bpos = 0;
                    if (lane == 0) bpos = atomicAdd(&qcount, __popcll(m));
                    bpos = __builtin_amdgcn_readfirstlane(bpos);
                    const int pos = bpos + __popcll(m & ((1ull << lane) - 1ull));
                    if (hit && pos < kSpQ) {
#pragma unroll
                        for (int a = 0; a < NREC; ++a) rec[a * kSpQ + pos] = col[a];
                    }
                }
            }
            __syncthreads();
            nrec = qcount;
            if (nrec > kSpQ) { over = true; nrec = 0; }     // block-uniform
        }
        __syncthreads();                                     // records and list heads are in place
        // ---- B: every record joins the list of its cell (cell rows r0 .. r1 serve the destination rows of the band)
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int i = tid + r * kSpNT2;
            if (i < nrec) {
                const int cx = (int)__builtin_amdgcn_fmed3f(floorf(rx[i]), -2.0f, (float)w) - dx0 + 1;
                const int cy = (int)__builtin_amdgcn_fmed3f(floorf(ry[i]), -2.0f, (float)h) - dy0 + 1;
                if ((uint32_t)cx < (uint32_t)kCW && cy >= r0 && cy <= r1)
                    link[i] = (uint16_t)atomicExch(&head[cy * kCW + cx], (uint32_t)i);
            }
        }
        over = __syncthreads_or((int)over) != 0;
        if (over) break;
        // ---- S: every cell's records in raster order of their source pixels (ascending key) -- the order in which the
        // reference's scatter_add_ adds them within a corner class.  Up to four are sorted in registers and written as
        // one 8-byte slot group; a longer list (a compression or fold of the flow) is sorted as a list and walked by its
        // readers.
        bool toolong = false;
#pragma unroll
        for (int r = 0; r < kCellRounds; ++r) {
            const int c = tid + r * kSpNT2;
            if (c < kCells) {
                uint32_t e[4];
                e[0] = head[c];
                e[1] = e[0] != kEnd ? (uint32_t)link[e[0]] : kEnd;
                e[2] = e[1] != kEnd ? (uint32_t)link[e[1]] : kEnd;
                e[3] = e[2] != kEnd ? (uint32_t)link[e[2]] : kEnd;
                const uint32_t e4 = e[3] != kEnd ? (uint32_t)link[e[3]] : kEnd;
                if (e4 != kEnd) {
                    // a longer list: insertion sort of the linked list itself.  Records were pushed in roughly ascending
                    // key order, so the list runs roughly descending and most nodes go straight to the front of the
                    // sorted list.  The limit is on the LENGTH (the same in every run, unlike the order the atomics leave):
                    // beyond it the tile takes the float-atomics fallback.
                    int len = 5;
                    for (uint32_t e = link[e4]; e != kEnd && len <= kSpLong; e = link[e]) ++len;
                    if (len > kSpLong) {
                        toolong = true;
                    } else {
                        uint32_t sorted = kEnd, cur = e[0];
                        while (cur != kEnd) {
                            const uint32_t nxt = link[cur], k = rkey[cur];
                            if (sorted == kEnd || rkey[sorted] > k) {
                                link[cur] = (uint16_t)sorted; sorted = cur;
                            } else {
                                uint32_t q = sorted, qn = link[q];
                                while (qn != kEnd && rkey[qn] < k) { q = qn; qn = link[q]; }
                                link[cur] = (uint16_t)qn; link[q] = (uint16_t)cur;
                            }
                            cur = nxt;
                        }
                        head[c] = sorted;
                    }
                    e[0] = kLongCell;
                } else if (e[1] != kEnd) {
                    uint32_t key[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) key[j] = e[j] != kEnd ? rkey[e[j]] : 0xffffffffu;
#define OFL_CSWAP(a_, b_) { const bool sw = key[a_] > key[b_]; const uint32_t tk = sw ? key[b_] : key[a_], te = sw ? e[b_] : e[a_]; \
                            key[b_] = sw ? key[a_] : key[b_]; e[b_] = sw ? e[a_] : e[b_]; key[a_] = tk; e[a_] = te; }
                    OFL_CSWAP(0, 1) OFL_CSWAP(2, 3) OFL_CSWAP(0, 2) OFL_CSWAP(1, 3) OFL_CSWAP(1, 2)
#undef OFL_CSWAP
                }
                slots[c] = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
            }
        }
        over = __syncthreads_or((int)toolong) != 0;
        if (over) break;
        // ---- C: the sums of this thread's 2 destination pixels (if their row is in the band), finalize.
        // The pair reads 3 x 2 cells; every record of a cell is fetched once and added to each corner-class sum it
        // belongs to (sp_use).
        const bool mine = inimg && ly >= r0 && ly < r1;
        float tot[2][1 + NCH];
        if (mine) {
            float a[2][2][1 + NCH];                                   // [pixel of the pair][x-corner]: the corner row in hand
            auto clear = [&]() {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int kx = 0; kx < 2; ++kx)
#pragma unroll
                        for (int c = 0; c < 1 + NCH; ++c) a[k][kx][c] = 0.0f;
            };
            const int cm = max(lx2 + 1, 0);                            // cell column of pixel 1's x-corner 1 = of pixel 0's x-corner 0
            auto cell = [&](auto dc_, auto ky_) {
                constexpr int DC = decltype(dc_)::value, KY = decltype(ky_)::value;
                const int c = (ly + 1 - KY) * kCW + max(cm + DC, 0);   // (solo: pixel 0's own cells do not exist; it is never stored)
                const uint2 sl = slots[c];
                const uint32_t e0 = sl.x & 0xffffu, e1 = sl.x >> 16, e2 = sl.y & 0xffffu, e3 = sl.y >> 16;
                if (e0 == kEnd) return;
                if (e0 != kLongCell) {
                    sp_use<NC, NCH, DC, KY>(rec, e0, a);
                    if (e1 != kEnd) {
                        sp_use<NC, NCH, DC, KY>(rec, e1, a);
                        if (e2 != kEnd) {
                            sp_use<NC, NCH, DC, KY>(rec, e2, a);
                            if (e3 != kEnd) sp_use<NC, NCH, DC, KY>(rec, e3, a);
                        }
                    }
                } else {                                               // phase S left the list in raster order
                    for (uint32_t e = head[c]; e != kEnd; e = link[e]) sp_use<NC, NCH, DC, KY>(rec, e, a);
                }
            };
            using std::integral_constant;
            clear();                                                   // corner row 0: classes 0, 1
            cell(integral_constant<int, -1>{}, integral_constant<int, 0>{});
            cell(integral_constant<int, 0>{}, integral_constant<int, 0>{});
            cell(integral_constant<int, 1>{}, integral_constant<int, 0>{});
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int c = 0; c < 1 + NCH; ++c) tot[k][c] = a[k][0][c] + a[k][1][c];
            clear();                                                   // corner row 1: classes 2, 3
            cell(integral_constant<int, -1>{}, integral_constant<int, 1>{});
            cell(integral_constant<int, 0>{}, integral_constant<int, 1>{});
            cell(integral_constant<int, 1>{}, integral_constant<int, 1>{});
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int c = 0; c < 1 + NCH; ++c) tot[k][c] = (tot[k][c] + a[k][0][c]) + a[k][1][c];   // ((c0 + c1) + c2) + c3
        }
        if (mine) finalize(tot);
        if (nb > 1) __syncthreads();                      // the next band re-uses the LDS
    }
    if (!over) { flush_flags(); return; }
    // ---- fallback for this tile (a band of destination rows that more than kSpQ records touch): LDS float atomics,
    // records streamed from the queue (plane 0 density, then the data channels; the mask channel accumulates the INVALID
    // weight so that an all-valid pixel is exactly 1 in any order)
    if (tid == 0) atomicAdd(&p.overflow[1], 1);                       // statistics: tiles that left the exact path
    __syncthreads();
    for (int i = tid; i < (1 + NCH) * kPx; i += kSpNT2) acc[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < qlen; i += kSpNT2) {
        float dd[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) dd[c] = qrec(3 + c, i);
        const bool invalid = MCH ? ((__float_as_uint(qrec(2, i)) & 1u) == 0u) : false;
        float wx[2], wy[2]; int ix[2], iy[2];
        sp_corners(qrec(0, i), qrec(1, i), wmax, hmax, dx0, dy0, wx, wy, ix, iy);
#pragma unroll
        for (int ky = 0; ky < 2; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
                const float wgt = wy[ky] * wx[kx];
                const int xl = ix[kx], yl = iy[ky];
                if (wgt == 0.0f || (uint32_t)xl >= (uint32_t)kSpTW || (uint32_t)yl >= (uint32_t)kSpTH) continue;
                const int d = yl * kSpTW + xl;
                atomicAdd(&acc[d], wgt);
#pragma unroll
                for (int c = 0; c < NC; ++c) atomicAdd(&acc[(1 + c) * kPx + d], wgt * dd[c]);
                if (MCH && invalid) atomicAdd(&acc[(1 + NC) * kPx + d], wgt);
            }
        }
    }
    __syncthreads();
    dflags = 0;                                           // (bands finalized before the tile left the exact path are overwritten)
    if (inimg) {
        float tot[2][1 + NCH];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = ly * kSpTW + max(lx2 + k, 0);
#pragma unroll
            for (int c = 0; c < 1 + NCH; ++c) tot[k][c] = acc[c * kPx + d];
            if (MCH) tot[k][1 + NC] = tot[k][0] - tot[k][1 + NC];      // density - invalid weight
        }
        finalize(tot);
    }
    flush_flags();
}

// zero the fallback accumulator only when the atomics path will run
__global__ __launch_bounds__(256) void zero_if_set_kernel(float* __restrict__ ptr, int64_t count, const int32_t* __restrict__ flag) {
    if (*flag == 0) return;
    const int64_t n4 = count >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        reinterpret_cast<f4*>(ptr)[i] = (f4){0.f, 0.f, 0.f, 0.f};
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) ptr[(n4 << 2) + threadIdx.x] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// flow flags (ofl_flow_flags_f32)
// ------------------------------------------------------------------------------------------------
template <bool VEC>   // VEC: 4 pixels per thread and step (16-byte flow loads, one mask dword); needs hw % 4 == 0 and aligned planes
__global__ __launch_bounds__(256) void flow_flags_kernel(const float* __restrict__ flow, int64_t flow_bs,
                                                         const uint8_t* __restrict__ mask, int64_t mask_bs,
                                                         int32_t* __restrict__ flags, int64_t hw) {
    const int n = blockIdx.y;
    const float* fu = flow + n * flow_bs;
    const uint8_t* mk = mask ? mask + n * mask_bs : nullptr;
    int f = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC) {
        const int64_t hw4 = hw >> 2;
        for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < hw4; i0 += 4 * stride) {
            f4 a[4], b[4]; uint32_t m4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                         // four independent 16-byte groups in flight
                const int64_t i = i0 + r * stride;
                if (i < hw4) {
                    a[r] = reinterpret_cast<const f4*>(fu)[i]; b[r] = reinterpret_cast<const f4*>(fu + hw)[i];
                    m4[r] = mk ? reinterpret_cast<const uint32_t*>(mk)[i] : 0x01010101u;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (i0 + r * stride < hw4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) f |= flag_bits(a[r][k], b[r][k], ((m4[r] >> (8 * k)) & 0xffu) != 0u);
                }
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += stride)
            f |= flag_bits(fu[i], fu[hw + i], mk ? (mk[i] != 0) : true);
    }
    f = wave_or_flags(f);
    if ((threadIdx.x & 63) == 0) flag_or(&flags[n], f);
}

// fp16-stored flows (BASELINE config 5): the reference's entry conversion `vecs.float()` (utils.py:95,118) and the flag
// reduction in ONE pass -- 4 pixels per thread and step: 8-byte fp16 loads, 16-byte fp32 stores, one mask dword.
// Needs hw % 4 == 0 and 8 / 16-byte aligned planes (else the binding converts with torch and calls ofl_flow_flags_f32).
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void flow_f16_kernel(const _Float16* __restrict__ src, int64_t src_bs,
                                                       const uint8_t* __restrict__ mask, int64_t mask_bs,
                                                       float* __restrict__ dst, int32_t* __restrict__ flags, int64_t hw) {
    const int n = blockIdx.y;
    const _Float16* su = src + n * src_bs;
    float* du = dst + (int64_t)n * 2 * hw;
    const uint8_t* mk = mask ? mask + n * mask_bs : nullptr;
    int f = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, hw4 = hw >> 2;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < hw4; i0 += 4 * stride) {
        h4 a[4], b[4]; uint32_t m4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                             // four independent groups in flight
            const int64_t i = i0 + r * stride;
            if (i < hw4) {
                a[r] = reinterpret_cast<const h4*>(su)[i]; b[r] = reinterpret_cast<const h4*>(su + hw)[i];
                m4[r] = mk ? reinterpret_cast<const uint32_t*>(mk)[i] : 0x01010101u;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = i0 + r * stride;
            if (i < hw4) {
                const f4 u = {(float)a[r][0], (float)a[r][1], (float)a[r][2], (float)a[r][3]};
                const f4 v = {(float)b[r][0], (float)b[r][1], (float)b[r][2], (float)b[r][3]};
                reinterpret_cast<f4*>(du)[i] = u; reinterpret_cast<f4*>(du + hw)[i] = v;
#pragma unroll
                for (int k = 0; k < 4; ++k) f |= flag_bits(u[k], v[k], ((m4[r] >> (8 * k)) & 0xffu) != 0u);
            }
        }
    }
    f = wave_or_flags(f);
    if ((threadIdx.x & 63) == 0) flag_or(&flags[n], f);
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
// `splat`: the forward splat keeps the reference's own limit -- its position index is formed in fp32 (utils.py:1118), exact
// only below 2^24 pixels.  The backward warp and the flag reductions have no such limit in the reference (grid_sample
// indexes with integers): frames of 2^24 pixels and more take the generic kernels (64-bit pixel offsets); the staged
// kernels' 24-bit multiplies stay below it.
inline int check_dims(int32_t n, int32_t c, int32_t h, int32_t w, bool splat = true) {
    if (n < 1 || c < 1 || h < 1 || w < 1) return OFL_E_SHAPE;
    if (splat && (int64_t)h * w >= (1ll << 24)) return OFL_E_SHAPE;
    if ((int64_t)h * w >= (1ll << 31)) return OFL_E_SHAPE;
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
int g_splat_pass_images = 0;   // ofl_set_option(OFL_OPT_SPLAT_PASS_IMAGES, .): 0 = as many as fit ~4 GiB of queues

template <int NC>
int launch_warp_lds(const WarpParams& p, unsigned grid, hipStream_t st) {
    const bool valid = p.valid != nullptr, add = p.addend != nullptr;
    if (NC == 2 && p.src_b) {                              // (host: 2 channels, valid mask, no addend, no output flags)
        hipLaunchKernelGGL((warp_bwd_lds_kernel<NC, true, false, false, NC == 2>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
        return (int)hipGetLastError();
    }
    if (NC == 2 && p.dst_flags) {                          // (host: only with a valid mask)
        if (add) hipLaunchKernelGGL((warp_bwd_lds_kernel<NC, true, true, NC == 2>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
        else hipLaunchKernelGGL((warp_bwd_lds_kernel<NC, true, false, NC == 2>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
        return (int)hipGetLastError();
    }
#define OFL_LAUNCH_L(V, A)                                                                                       \
    if (valid == V && add == A) {                                                                                \
        hipLaunchKernelGGL((warp_bwd_lds_kernel<NC, V, A>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);          \
        return (int)hipGetLastError();                                                                           \
    }
    OFL_LAUNCH_L(false, false) OFL_LAUNCH_L(true, false) OFL_LAUNCH_L(false, true) OFL_LAUNCH_L(true, true)
#undef OFL_LAUNCH_L
    return OFL_E_ARG;
}

// 8-bit images: uint8 source planes, float or uint8 destination (no addend, no flag words)
template <int NC, typename TD>
int launch_warp_lds_u8(const WarpParams& p, unsigned grid, hipStream_t st) {
    if (p.valid) hipLaunchKernelGGL((warp_bwd_lds_kernel<NC, true, false, false, false, uint8_t, TD>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
    else hipLaunchKernelGGL((warp_bwd_lds_kernel<NC, false, false, false, false, uint8_t, TD>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
    return (int)hipGetLastError();
}

inline bool aligned_to(const void* ptr, size_t a) { return (reinterpret_cast<uintptr_t>(ptr) % a) == 0; }

void launch_flow_flags(const float* flow, int64_t flow_bs, const uint8_t* mask, int64_t mask_bs, int32_t* flags, int32_t n,
                       int64_t hw, hipStream_t st) {
    const bool vec = (hw % 4) == 0 && aligned_to(flow, 16) && (flow_bs % 4) == 0 && (!mask || (aligned_to(mask, 4) && (mask_bs % 4) == 0));
    if (vec) {
        // about 512 blocks in all: every wave ends with a look at (and maybe an atomic on) its image's shared word, and
        // few long-running blocks stream better than many short ones (B=64 1080p: 3.1 -> 5.8 TB/s; B=16: 2.5 -> 5.4)
        int64_t bx = (hw / 4 + 1023) / 1024;               // 4 groups of 4 pixels per thread and step
        int64_t cap = 512 / n;
        cap = cap < 16 ? 16 : (cap > 256 ? 256 : cap);
        if (bx > cap) bx = cap;
        hipLaunchKernelGGL(flow_flags_kernel<true>, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, flow, flow_bs, mask, mask_bs, flags, hw);
    } else {
        int64_t bx = (hw + 255) / 256;
        if (bx > 512) bx = 512;
        hipLaunchKernelGGL(flow_flags_kernel<false>, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, flow, flow_bs, mask, mask_bs, flags, hw);
    }
}

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
    if (tp.s.with_mask_chan) hipLaunchKernelGGL((splat_tile_kernel<NC, true>), dim3(grid), dim3(kSpNT2), 0, st, tp);
    else hipLaunchKernelGGL((splat_tile_kernel<NC, false>), dim3(grid), dim3(kSpNT2), 0, st, tp);
    return (int)hipGetLastError();
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

__attribute__((visibility("default"))) int ofl_version(void) { return 18; }   // 18: ofl_aux_kernels.hip (backward passes, point sampler, extents); 13: dst_flags in ofl_warp_bwd_f32 / ofl_splat_tiled_f32, fixed-address splat queues; 14: data_b; 15: src_b; 16: ofl_warp_bwd_u8; 17: ofl_flow_from_f16

__attribute__((visibility("default"))) int ofl_set_option(int32_t key, int32_t value) {
    if (key == OFL_OPT_WARP_PATH && value >= 0 && value <= 2) { g_warp_path = value; return OFL_OK; }
    if (key == OFL_OPT_WARP_SHEAR && (value == 0 || value == 1)) { g_warp_shear = value; return OFL_OK; }
    if (key == OFL_OPT_SPLAT_PASS_IMAGES && value >= 0) { g_splat_pass_images = value; return OFL_OK; }
    return OFL_E_ARG;
}

__attribute__((visibility("default"))) int ofl_warp_bwd_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* src, int64_t src_bs,
    const float* src_b, int64_t src_b_bs, const uint8_t* src_mask, int64_t src_mask_bs, const uint8_t* flow_mask, int64_t flow_mask_bs,
    const float* addend, int64_t addend_bs, float a_sign, float g_sign, float* dst, uint8_t* valid,
    int32_t* flow_flags, int32_t* src_flags, int32_t* dst_flags, int32_t n, int32_t c, int32_t h, int32_t w,
    int32_t round_mode, void* stream) {
    if (!flow || !src || !dst) return OFL_E_NULL;
    int rc = check_dims(n, c, h, w, false);
    if (rc) return rc;
    if (src_flags && (c != 2 || !flow_flags)) return OFL_E_ARG;
    if (dst_flags && (c != 2 || !valid)) return OFL_E_ARG;
    if (round_mode < 0 || round_mode > 2) return OFL_E_ARG;
    if (!(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    WarpParams p;
    p.flow = flow; p.flow_bs = flow_bs; p.src = src; p.src_bs = src_bs;
    p.src_b = nullptr; p.src_b_bs = 0;
    p.src_mask = src_mask; p.src_mask_bs = src_mask_bs; p.flow_mask = flow_mask; p.flow_mask_bs = flow_mask_bs;
    p.addend = addend; p.addend_bs = addend_bs; p.dst = dst; p.valid = valid;
    p.flow_flags = flow_flags; p.src_flags = src_flags; p.dst_flags = nullptr;
    p.n = n; p.c = c; p.h = h; p.w = w;
    p.flow_sign = flow_sign; p.a_sign = a_sign; p.g_sign = g_sign; p.round_mode = round_mode;
    p.wm1 = (float)(w - 1); p.hm1 = (float)(h - 1);
    p.half_wm1 = p.wm1 / 2.0f; p.half_hm1 = p.hm1 / 2.0f;
    p.rcp_wm1 = 1.0f / p.wm1; p.rcp_hm1 = 1.0f / p.hm1;
    p.lds_bytes = kLdsBytes;
    p.shear = (g_warp_shear && (int64_t)h + 4 * (int64_t)w + 8 < 32760) ? 1 : 0;   // sheared rows stay within 16 bits (|slope| <= 16 rows per chunk column)
    p.add_is_flow = (addend != nullptr && addend == flow && addend_bs == flow_bs && c == 2) ? 1 : 0;
    p.dst_bs = (int64_t)c * h * w;
    hipStream_t st = (hipStream_t)stream;
    if ((int64_t)((w + 31) / 32) * ((h + 15) / 16) * n >= (1ll << 31)) return OFL_E_SHAPE;
    if (dst_flags) {
        hipError_t e = hipMemsetAsync(dst_flags, 0, (size_t)n * sizeof(int32_t), st);
        if (e != hipSuccess) return (int)e;
    }
    // LDS-staged fast path: <= 3 channels, at least one whole 4-pixel group per row, 16-bit box coordinates (any width:
    // 16-byte accesses at 4-byte alignment, mask bytes at any alignment)
    const bool lds_ok = g_warp_path != 1 && w >= 4 && h >= 2 && w < 32760 && h < 32760 && (int64_t)h * w < (1ll << 24);
    if (src_b) {   // only the staged 2-channel kernel with a valid mask subtracts on the fly: anything else is the caller's job
        if (!(lds_ok && c == 2 && valid && !addend && !dst_flags && !flow_flags)) return OFL_E_UNSUPPORTED;
        p.src_b = src_b; p.src_b_bs = src_b_bs;
    }
    if (lds_ok) {
        const int64_t hw = (int64_t)h * w;
        if (src_flags) {   // the staged path never reads `src` at its own pixel: a separate reduction supplies its flags
            launch_flow_flags(src, src_bs, src_mask, src_mask_bs, src_flags, n, hw, st);
            p.src_flags = nullptr;
        }
        p.dst_flags = dst_flags;                                             // a by-product of the staged kernel (c == 2: one group)
        const unsigned g = warp_geometry(p, kLdsTWQ * 4, 2 * kLdsTH);
        // more than 3 channels: groups of 3 (the staged box holds 3 channels + the mask channel); the valid mask and the
        // flow flags come out of the first group
        for (int32_t c0 = 0; c0 < c; c0 += 3) {
            const int32_t nc = (c - c0) < 3 ? (c - c0) : 3;
            WarpParams q = p;
            q.c = nc;
            q.src = src + c0 * hw; q.dst = dst + c0 * hw;                 // (batch strides stay: planes of one image are contiguous)
            if (addend) q.addend = addend + c0 * hw;
            if (c0 > 0) { q.valid = nullptr; q.flow_flags = nullptr; q.src_mask = nullptr; }
            switch (nc) {
                case 1: rc = launch_warp_lds<1>(q, g, st); break;
                case 2: rc = launch_warp_lds<2>(q, g, st); break;
                default: rc = launch_warp_lds<3>(q, g, st); break;
            }
            if (rc) return rc;
        }
        return OFL_OK;
    }
    const unsigned grid = warp_geometry(p, kTileW, kTileH);
    switch (c) {
        case 1: rc = launch_warp<1>(p, grid, st); break;
        case 2: rc = launch_warp<2>(p, grid, st); break;
        case 3: rc = launch_warp<3>(p, grid, st); break;
        case 4: rc = launch_warp<4>(p, grid, st); break;
        default: rc = launch_warp<0>(p, grid, st); break;
    }
    if (rc == OFL_OK && dst_flags)                                            // generic kernel: a reduction over the output
        launch_flow_flags(dst, (int64_t)2 * h * w, valid, (int64_t)h * w, dst_flags, n, (int64_t)h * w, st);
    return rc;
}

__attribute__((visibility("default"))) int ofl_warp_bwd_u8(
    const float* flow, int64_t flow_bs, float flow_sign, const uint8_t* src, int64_t src_bs,
    const uint8_t* src_mask, int64_t src_mask_bs, const uint8_t* flow_mask, int64_t flow_mask_bs,
    void* dst, int32_t dst_is_u8, uint8_t* valid, int32_t n, int32_t c, int32_t h, int32_t w, int32_t round_mode,
    void* stream) {
    if (!flow || !src || !dst) return OFL_E_NULL;
    int rc = check_dims(n, c, h, w, false);
    if (rc) return rc;
    if ((int64_t)h * w >= (1ll << 24)) return OFL_E_UNSUPPORTED;          // staged kernel only (24-bit offsets)
    if (round_mode < 0 || round_mode > 2) return OFL_E_ARG;
    if (dst_is_u8 && round_mode != OFL_ROUND_U8) return OFL_E_ARG;       // bytes only hold rounded, clamped values
    if (!(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    if ((int64_t)((w + 31) / 32) * ((h + 15) / 16) * n >= (1ll << 31)) return OFL_E_SHAPE;
    if (!(g_warp_path != 1 && w >= 4 && h >= 2 && w < 32760 && h < 32760)) return OFL_E_UNSUPPORTED;   // staged kernel only
    WarpParams p = {};
    p.flow = flow; p.flow_bs = flow_bs; p.src_bs = src_bs;
    p.src_mask = src_mask; p.src_mask_bs = src_mask_bs; p.flow_mask = flow_mask; p.flow_mask_bs = flow_mask_bs;
    p.valid = valid;
    p.n = n; p.c = c; p.h = h; p.w = w;
    p.flow_sign = flow_sign; p.a_sign = 1.0f; p.g_sign = 1.0f; p.round_mode = round_mode;
    p.wm1 = (float)(w - 1); p.hm1 = (float)(h - 1);
    p.half_wm1 = p.wm1 / 2.0f; p.half_hm1 = p.hm1 / 2.0f;
    p.rcp_wm1 = 1.0f / p.wm1; p.rcp_hm1 = 1.0f / p.hm1;
    p.lds_bytes = kLdsBytes;
    p.shear = (g_warp_shear && (int64_t)h + 4 * (int64_t)w + 8 < 32760) ? 1 : 0;
    p.dst_bs = (int64_t)c * h * w;
    hipStream_t st = (hipStream_t)stream;
    const int64_t hw = (int64_t)h * w;
    const unsigned g = warp_geometry(p, kLdsTWQ * 4, 2 * kLdsTH);
    for (int32_t c0 = 0; c0 < c; c0 += 3) {                               // channel groups of 3, as ofl_warp_bwd_f32
        const int32_t nc = (c - c0) < 3 ? (c - c0) : 3;
        WarpParams q = p;
        q.c = nc;
        q.src = reinterpret_cast<const float*>(src + c0 * hw);             // (the uint8 kernels re-read these as byte pointers)
        q.dst = dst_is_u8 ? reinterpret_cast<float*>(static_cast<uint8_t*>(dst) + c0 * hw) : static_cast<float*>(dst) + c0 * hw;
        if (c0 > 0) { q.valid = nullptr; q.src_mask = nullptr; }
        if (dst_is_u8) {
            switch (nc) {
                case 1: rc = launch_warp_lds_u8<1, uint8_t>(q, g, st); break;
                case 2: rc = launch_warp_lds_u8<2, uint8_t>(q, g, st); break;
                default: rc = launch_warp_lds_u8<3, uint8_t>(q, g, st); break;
            }
        } else {
            switch (nc) {
                case 1: rc = launch_warp_lds_u8<1, float>(q, g, st); break;
                case 2: rc = launch_warp_lds_u8<2, float>(q, g, st); break;
                default: rc = launch_warp_lds_u8<3, float>(q, g, st); break;
            }
        }
        if (rc) return rc;
    }
    return OFL_OK;
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
    p.data_b = nullptr; p.data_b_bs = 0;
    p.weight_mask = weight_mask; p.weight_mask_bs = weight_mask_bs;
    p.chan_mask_a = chan_mask_a; p.chan_mask_a_bs = chan_mask_a_bs;
    p.chan_mask_b = chan_mask_b; p.chan_mask_b_bs = chan_mask_b_bs;
    p.with_mask_chan = with_mask_chan; p.occlude = occlude;
    p.n = n; p.c = c; p.h = h; p.w = w;
    p.dst_bs = (int64_t)c * h * w;
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


#ifndef OFL_SP_POOL_LOG2
#define OFL_SP_POOL_LOG2 30   // workspace budget of one pass, in 4-byte words (4 GiB: ~57 frames of 1080p; measured -8 % against 1 GiB at B=16)
#endif
constexpr int kSpRecFloats = 6;   // floats per record: x, y, key (+ mask-channel bit), up to 3 data channels
// workspace words of one pass of `images` frames: header | queue lengths | secondary block ids | primary regions |
// secondary blocks
static int64_t splat_sec_blocks(int64_t tiles) { return (tiles + kSpSecDiv - 1) / kSpSecDiv + 64; }   // (measured: 0.02 / 0.06 / 0.09 per tile drawn at sigma 8 / 12 / 16)
static int64_t splat_pass_words(int64_t images, int32_t h, int32_t w) {
    const int64_t tiles = images * ((w + kSpTW - 1) / kSpTW) * ((h + kSpTH - 1) / kSpTH);
    return 8 + (((1 + kSpSlots) * tiles + 3) & ~(int64_t)3) + kSpRecFloats * (int64_t)kSpPrim * (tiles + splat_sec_blocks(tiles));
}
static int64_t splat_chunk_images(int32_t n, int32_t h, int32_t w) {
    int64_t c = ((int64_t)1 << OFL_SP_POOL_LOG2) / splat_pass_words(1, h, w);
    if (c < 1) c = 1;
    if (g_splat_pass_images > 0 && g_splat_pass_images < c) c = g_splat_pass_images;
    if (c >= n) return n;
    const int64_t passes = (n + c - 1) / c;            // equal passes: every pass pays ~0.1 ms of launch gaps and tails
    return (n + passes - 1) / passes;
}

__attribute__((visibility("default"))) int64_t ofl_splat_tiled_pass_images(int32_t n, int32_t h, int32_t w) { return splat_chunk_images(n, h, w); }

__attribute__((visibility("default"))) int64_t ofl_splat_tiled_workspace_ints(int32_t n, int32_t h, int32_t w) {
    return splat_pass_words(splat_chunk_images(n, h, w), h, w);
}

__attribute__((visibility("default"))) int ofl_splat_tiled_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* xs, const float* ys, int64_t xy_bs,
    const float* data, int64_t data_bs, float data_sign, const float* data_b, int64_t data_b_bs,
    const uint8_t* weight_mask, int64_t weight_mask_bs,
    const uint8_t* chan_mask_a, int64_t chan_mask_a_bs, const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
    int32_t with_mask_chan, int32_t occlude, float* dst, float* density, uint8_t* warped, uint8_t* valid,
    float* mask_chan, int32_t* dst_flags, int32_t* workspace, int64_t workspace_ints, float* accum_fallback, int32_t n,
    int32_t c, int32_t h, int32_t w, int32_t round_mode, void* stream) {
    if (!data || !dst || !workspace || !accum_fallback) return OFL_E_NULL;
    if (!flow && !(xs && ys)) return OFL_E_NULL;
    if (flow && !(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    if ((valid || mask_chan) && !with_mask_chan) return OFL_E_ARG;
    if (dst_flags && c != 2) return OFL_E_ARG;
    if (data_b && c > 2) return OFL_E_ARG;                           // (flows: the tile kernel only carries it for <= 2 channels)
    if (round_mode < 0 || round_mode > 2) return OFL_E_ARG;
    TiledParams tp = {};
    unsigned grid_unused;
    int rc = fill_splat(tp.s, flow, flow_bs, data, data_bs, data_sign, weight_mask, weight_mask_bs, chan_mask_a,
                        chan_mask_a_bs, chan_mask_b, chan_mask_b_bs, with_mask_chan, occlude, n, c, h, w, grid_unused);
    if (rc) return rc;
    // eligibility of the routed path: <= 3 channels, at least one whole 4-pixel group per row, 16-bit coordinates
    // (any width: 16 / 8-byte accesses at 4-byte alignment, mask bytes at any alignment)
    const bool ok = w >= 4 && w < 32768 && h < 32768;   // 15-bit rows and columns in the record key
    if (!ok) return OFL_E_UNSUPPORTED;
    if (workspace_ints < ofl_splat_tiled_workspace_ints(n, h, w)) return OFL_E_ARG;
    tp.s.flow_sign = flow_sign; tp.s.xs = xs; tp.s.ys = ys; tp.s.xy_bs = xy_bs;
    tp.s.dst = dst; tp.s.density = density; tp.s.warped = warped; tp.s.valid = valid; tp.s.mask_chan = mask_chan;
    tp.s.dst_flags = dst_flags;
    tp.s.data_b = data_b; tp.s.data_b_bs = data_b_bs;
    tp.s.round_mode = round_mode;
    tp.tiles_x = (w + kSpTW - 1) / kSpTW; tp.tiles_y = (h + kSpTH - 1) / kSpTH;
    tp.tiles_img = (uint32_t)(tp.tiles_x * tp.tiles_y);
    if ((int64_t)tp.tiles_img * n >= (1ll << 31)) return OFL_E_SHAPE;
    magic_u32((uint32_t)tp.tiles_x, tp.mx_m, tp.mx_s);
    magic_u32(tp.tiles_img, tp.mi_m, tp.mi_s);
    const int64_t chunk = splat_chunk_images(n, h, w), ctiles = chunk * tp.tiles_img;
    tp.overflow = workspace;
    tp.sec_count = workspace + 4;
    tp.cursor = workspace + 8;
    tp.sec = tp.cursor + ctiles;
    tp.nsec = (int32_t)splat_sec_blocks(ctiles);
    tp.prim = reinterpret_cast<float*>(workspace + 8 + (((1 + kSpSlots) * ctiles + 3) & ~(int64_t)3));   // 16-byte aligned columns
    tp.secp = tp.prim + (int64_t)kSpRecFloats * kSpPrim * ctiles;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(tp.overflow, 0, 4 * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    if (dst_flags) {
        e = hipMemsetAsync(dst_flags, 0, (size_t)n * sizeof(int32_t), st);
        if (e != hipSuccess) return (int)e;
    }
    const SplatParams all = tp.s;
    const int64_t hw = (int64_t)h * w;
    // more than 3 channels: groups of 3 (a record holds 3 data channels); density and masks come out of the first group
    for (int32_t c0 = 0; c0 < c; c0 += 3) {
    SplatParams full = all;
    full.c = (c - c0) < 3 ? (c - c0) : 3;
    full.data = all.data + c0 * hw; full.dst = all.dst + c0 * hw;
    if (all.data_b) full.data_b = all.data_b + c0 * hw;
    if (c0 > 0) { full.with_mask_chan = 0; full.density = nullptr; full.warped = nullptr; full.valid = nullptr; full.mask_chan = nullptr; }
    const int32_t cg = full.c;
    for (int64_t n0 = 0; n0 < n; n0 += chunk) {          // same stream: the queues of a pass are re-used by the next one
        const int64_t nn = (n - n0) < chunk ? (n - n0) : chunk;
        SplatParams& q = tp.s;
        q = full;
        q.n = (int32_t)nn;
        if (q.flow) q.flow = full.flow + n0 * full.flow_bs;
        if (q.xs) { q.xs = full.xs + n0 * full.xy_bs; q.ys = full.ys + n0 * full.xy_bs; }
        q.data = full.data + n0 * full.data_bs;
        if (q.data_b) q.data_b = full.data_b + n0 * full.data_b_bs;
        if (q.weight_mask) q.weight_mask = full.weight_mask + n0 * full.weight_mask_bs;
        if (q.chan_mask_a) q.chan_mask_a = full.chan_mask_a + n0 * full.chan_mask_a_bs;
        if (q.chan_mask_b) q.chan_mask_b = full.chan_mask_b + n0 * full.chan_mask_b_bs;
        q.dst = full.dst + n0 * all.dst_bs;
        if (q.density) q.density = full.density + n0 * hw;
        if (q.warped) q.warped = full.warped + n0 * hw;
        if (q.valid) q.valid = full.valid + n0 * hw;
        if (q.mask_chan) q.mask_chan = full.mask_chan + n0 * hw;
        if (q.dst_flags) q.dst_flags = full.dst_flags + n0;
        tp.total = (int64_t)tp.tiles_img * nn;
        tp.per_xcd = (tp.total + kXcds - 1) / kXcds;
        e = hipMemsetAsync(tp.sec_count, 0, (size_t)(4 + ctiles) * sizeof(int32_t), st);      // blocks drawn | queue lengths
        if (e != hipSuccess) return (int)e;
        e = hipMemsetAsync(tp.sec, 0xff, (size_t)ctiles * kSpSlots * sizeof(int32_t), st);   // no blocks yet (-1)
        if (e != hipSuccess) return (int)e;
        const unsigned grid = (unsigned)(tp.per_xcd * kXcds);
        hipLaunchKernelGGL(splat_route_kernel, dim3(grid), dim3(kSpNT), 0, st, tp);
        rc = (int)hipGetLastError();
        if (rc) return rc;
        switch (cg) {
            case 1: rc = launch_splat_tile<1>(tp, grid, st); break;
            case 2: rc = launch_splat_tile<2>(tp, grid, st); break;
            default: rc = launch_splat_tile<3>(tp, grid, st); break;
        }
        if (rc) return rc;
        // two-pass global-atomics path for this pass's images, armed only if the launch was flagged (pool overflow / a
        // source tile spread too wide); every kernel below exits at once otherwise
        SplatParams fb = tp.s;
        fb.accum = accum_fallback;
        fb.run_if_set = tp.overflow;
        const int planes = 1 + cg + (fb.with_mask_chan ? 1 : 0);
        hipLaunchKernelGGL(zero_if_set_kernel, dim3(2048), dim3(256), 0, st, accum_fallback, nn * planes * hw, tp.overflow);
        unsigned g2;
        tile_grid((int32_t)nn, h, w, fb.tiles_x, fb.tiles_y, fb.total_tiles, fb.per_xcd, g2);
        switch (cg) {
            case 1: hipLaunchKernelGGL(splat_fwd_kernel<1>, dim3(g2), dim3(256), 0, st, fb);
                    hipLaunchKernelGGL(splat_finalize_kernel<1>, dim3(g2), dim3(256), 0, st, fb); break;
            case 2: hipLaunchKernelGGL(splat_fwd_kernel<2>, dim3(g2), dim3(256), 0, st, fb);
                    hipLaunchKernelGGL(splat_finalize_kernel<2>, dim3(g2), dim3(256), 0, st, fb); break;
            default: hipLaunchKernelGGL(splat_fwd_kernel<3>, dim3(g2), dim3(256), 0, st, fb);
                     hipLaunchKernelGGL(splat_finalize_kernel<3>, dim3(g2), dim3(256), 0, st, fb); break;
        }
    }
    }
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_flow_flags_f32(const float* flow, int64_t flow_bs,
                                                              const uint8_t* mask, int64_t mask_bs, float thr,
                                                              int32_t* flags, int32_t n, int32_t h, int32_t w,
                                                              void* stream) {
    if (!flow || !flags) return OFL_E_NULL;
    int rc = check_dims(n, 2, h, w, false);
    if (rc) return rc;
    if (n > 65535) return OFL_E_SHAPE;                       // (the batch index rides in blockIdx.y)
    if (thr != kZeroThr) return OFL_E_ARG;  // the reference's DEFAULT_THRESHOLD is the only value on the path
    launch_flow_flags(flow, flow_bs, mask, mask_bs, flags, n, (int64_t)h * w, (hipStream_t)stream);
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_flow_from_f16(const void* src_f16, int64_t src_bs,
                                                             const uint8_t* mask, int64_t mask_bs, float* dst,
                                                             int32_t* flags, int32_t n, int32_t h, int32_t w,
                                                             void* stream) {
    if (!src_f16 || !dst || !flags) return OFL_E_NULL;
    int rc = check_dims(n, 2, h, w, false);
    if (rc) return rc;
    if (n > 65535) return OFL_E_SHAPE;
    const int64_t hw = (int64_t)h * w;
    if ((hw % 4) != 0 || !aligned_to(src_f16, 8) || (src_bs % 4) != 0 || !aligned_to(dst, 16) ||
        (mask && (!aligned_to(mask, 4) || (mask_bs % 4) != 0)))
        return OFL_E_UNSUPPORTED;
    int64_t bx = (hw / 4 + 1023) / 1024, cap = 512 / n;
    cap = cap < 16 ? 16 : (cap > 256 ? 256 : cap);
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(flow_f16_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const _Float16*>(src_f16), src_bs, mask, mask_bs, dst, flags, hw);
    return (int)hipGetLastError();
}

}  // extern "C"
