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

// the four-tile column kernel on 64 x 16 tiles (ofl_warp_wide.hip: this file compiled with OFL_WIDE_TU); `params` = a WarpParams
int ofl_wide_launch_column(const void* params, int nc, int valid, int add, int rows, void* stream);   // rows: per-row extents (warp_bwd_rows_kernel) where they apply
int ofl_wide_launch_rows_small(const void* params, int nc, int valid, int add, int tiles, void* stream);   // small plain launches / mode 3: row tables, 1 or 2 tiles per block
int ofl_wide_launch_rows_h(const void* params, void* stream);                  // fp16 sources on the row-table kernel
int ofl_wide_launch_rows_grad(const void* params, int nc, int tiles, void* stream);       // gradient wrt the flow on the row-table kernel
int ofl_wide_launch_rows_u8(const void* params, int nc, int dst_is_u8, void* stream);         // uint8 images (bytes in; bytes or fp32 out) on the row-table kernel
int ofl_wide_launch_chan(const void* params, int valid, int rows, void* stream);      // the channel-loop kernel (C >= 4) on 64 x 16 tiles; rows: per-row extents where they apply
// the gather splat's diet kernel (ofl_splat_gather.hip: this file compiled with OFL_SPLAT_TU); `params` = a GatherParams; elem: 0 fp32, 1 fp16 in / fp32 out, 2 fp16 in and out
int ofl_splat_launch_gather_diet(const void* params, int nc, int mch, int elem, unsigned grid, void* stream, int extra_lds);   // extra_lds: bytes of dynamic LDS added to the launch (OFL_OPT_SPLAT_EXTRA_LDS: occupancy experiments)

// the kernel the library launched LAST, in any of its translation units (ofl_last_kernel_name: bench.py reports the instantiation
// the launcher actually picked instead of a string literal -- VERDICT r5): every launch goes through OFL_KLAUNCH
extern const void* g_ofl_last_kernel;
#define OFL_KLAUNCH(K, ...) do { g_ofl_last_kernel = (const void*)(K); hipLaunchKernelGGL(K, __VA_ARGS__); } while (0)

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

// One look / atomic per BLOCK on an image's shared word (the waves' words are combined in LDS first; every thread of the
// block calls this): with a few images, hundreds of waves end within microseconds of each other on the same word, and every
// such access is served one after the other at the memory side -- B = 8, 512 blocks: 55 us per wave-level OR, 42 us per
// block-level OR; B = 1: 26 -> 17 us (tools/ab_flags.py, profiles/r3_flags_host_route.txt).
__device__ __forceinline__ void block_flag_or(int32_t* addr, int f) {
    __shared__ int bflags;
    if (threadIdx.x == 0) bflags = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && f != 0) atomicOr(&bflags, f);
    __syncthreads();
    if (threadIdx.x == 0) flag_or(addr, bflags);
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
    // flow WINDOW (generic kernel only; Flow.apply(padding=...) with a 't' flow, flow_class.py:901-913, 924-932): fw != 0 -> flow and
    // flow_mask are fh x fw frames covering rows foy .., columns fox .. of the h x w frame; outside the window the flow is
    // zero (F.pad(mode='constant')) and the flow mask False
    int32_t fh, fw, foy, fox;
    static constexpr bool kLean = false;
};

// The COMMON CASE as a type (as SplatParamsLean for the splat): the same bytes read as WarpParamsLean promise the staged kernels a width
// that is a multiple of 4, no rounding, no flow-flag by-product and the sheared box -- `WP::kLean` folds those run-time switches (a scalar
// load + compare + branch per use: per staging round, per stored plane, per tile) out of the lean instantiations.  The host picks them when
// the promises hold: apply 't' -2 %, mode 3 -3 % (profiles/r5_warp_lean.txt).
struct WarpParamsLean : WarpParams { static constexpr bool kLean = true; };
#define OFL_WP_LEAN_OF(p_) (std::remove_reference<decltype(p_)>::type::kLean)
#define OFL_WP_WREM(p_) (OFL_WP_LEAN_OF(p_) ? 0 : ((p_).w & 3))
#define OFL_WP_ROUND(p_) (OFL_WP_LEAN_OF(p_) ? (int32_t)OFL_ROUND_NONE : (p_).round_mode)
#define OFL_WP_FLOW_FLAGS(p_) (OFL_WP_LEAN_OF(p_) ? (int32_t*)nullptr : (p_).flow_flags)
#define OFL_WP_SHEAR(p_) (OFL_WP_LEAN_OF(p_) || (p_).shear != 0)
typedef const WarpParams __attribute__((address_space(4))) WarpParamsK;
typedef const WarpParamsLean __attribute__((address_space(4))) WarpParamsLeanK;
#define OFL_OPAQUE_S(ptr_) asm volatile("" : "+s"(ptr_))
#ifndef OFL_WARP_KARG
#define OFL_WARP_KARG 1
#endif
#ifndef OFL_WARP_COL_ADD
#define OFL_WARP_COL_ADD 1
#endif

constexpr int kTileW = 64;   // one wavefront spans 64 consecutive x: 256-byte rows per instruction
constexpr int kRows = 4;     // rows per thread (independent pixels in flight per lane)
constexpr int kTileH = 4 * kRows;  // 4 wavefronts per 256-thread block


// exact u32 division by an invariant divisor: q = (((n - t) >> s1) + t) >> s2 with t = umulhi(m, n); s = (s1 << 16) | s2
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t m, uint32_t s) {
    const uint32_t t = __umulhi(m, n);
    return (((n - t) >> (s >> 16)) + t) >> (s & 0xffffu);
}

// XCD-aware 32-bit tile decode (no 64-bit integer division in the kernel prologue)
template <typename WP>
__device__ __forceinline__ bool decode_tile(const WP& p, int& tx, int& ty, int& n) {
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
#ifndef OFL_WARP_NT_STORE
#define OFL_WARP_NT_STORE 1      // 1: the warped channels, 2: the valid mask too (measured: 1 is +0.5..1 %, 2 loses 3 % -- its 32-byte pieces need the L2 to merge them)
#endif
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
// flows stored in fp16 (BASELINE config 5): 4 / 2 / 1 values <-> floats.  The up-conversion is exact (the reference's
// `vecs.float()`, utils.py:95,118); a store rounds to nearest even (an OPTION of the fp16 entry points, never the default)
typedef _Float16 h4u __attribute__((ext_vector_type(4), aligned(2)));
typedef _Float16 h2u __attribute__((ext_vector_type(2), aligned(2)));
__device__ __forceinline__ f4 ld4(const _Float16* p) { const h4u v = *reinterpret_cast<const h4u*>(p); return (f4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
__device__ __forceinline__ f2 ld2(const _Float16* p) { const h2u v = *reinterpret_cast<const h2u*>(p); return (f2){(float)v[0], (float)v[1]}; }
__device__ __forceinline__ float ld1(const _Float16* p) { return (float)*p; }
__device__ __forceinline__ void st4(_Float16* p, f4 v) { *reinterpret_cast<h4u*>(p) = (h4u){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; }
__device__ __forceinline__ void st2(_Float16* p, f2 v) { *reinterpret_cast<h2u*>(p) = (h2u){(_Float16)v[0], (_Float16)v[1]}; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(_Float16* p, float v) { *p = (_Float16)v; }
// write-once outputs of the staged warp: non-temporal (OFL_WARP_NT_STORE), so that they stream past the L2 lines the halos live
// in; the gather splat's 16-byte output stores use it too (+1 % on smooth flows).  (Also measured, within +-1 %:
// non-temporal flow-mask loads, non-temporal loads in the splat's bin kernel.)
template <typename T> __device__ __forceinline__ void st4o(T* p, f4 v) { st4(p, v); }
#if OFL_WARP_NT_STORE
template <> __device__ __forceinline__ void st4o<float>(float* p, f4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f4u*>(p)); }
#endif
template <typename T> __device__ __forceinline__ float stored_as(float v) { return v; }          // the value a store of type T keeps
template <> __device__ __forceinline__ float stored_as<_Float16>(float v) { return (float)(_Float16)v; }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return reinterpret_cast<const U32u*>(p)->v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return reinterpret_cast<const U16u*>(p)->v; }
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { reinterpret_cast<U32u*>(p)->v = v; }
__device__ __forceinline__ void st16(uint8_t* p, uint32_t v) { reinterpret_cast<U16u*>(p)->v = (uint16_t)v; }
typedef uint32_t u32u __attribute__((aligned(1)));
__device__ __forceinline__ void st32o(uint8_t* p, uint32_t v) {
#if OFL_WARP_NT_STORE >= 2
    __builtin_nontemporal_store(v, reinterpret_cast<u32u*>(p));
#else
    st32(p, v);
#endif
}
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
#ifndef OFL_WARP_LEAN
#define OFL_WARP_LEAN 1
#endif
#ifndef OFL_WARP_REUSE
#define OFL_WARP_REUSE 1
#endif
#ifndef OFL_WARP_CHAN
#define OFL_WARP_CHAN 1
#endif
#ifndef OFL_WARP_CHAN_WIDE
#define OFL_WARP_CHAN_WIDE 1
#endif
#ifndef OFL_WARP_CHAN_SUBS
#define OFL_WARP_CHAN_SUBS 1       // tiles side by side per block of the 64 x 16 channel-loop kernel.  3 (one 768-thread block per CU, its three tiles in lockstep so that they share their x-halo lines) was built and measured: bit-identical, 40 % SLOWER (profiles/r5_chan_pmc.txt) -- the lockstep that shares the lines also makes the whole CU wait together
#endif
#ifndef OFL_WARP_ALWAYS_T
#define OFL_WARP_ALWAYS_T 1
#endif
#ifndef OFL_WARP_MERGE_BARRIER
#define OFL_WARP_MERGE_BARRIER 1
#endif
#ifndef OFL_WARP_UNCOND_ROUNDS
#define OFL_WARP_UNCOND_ROUNDS 1
#endif
#ifndef OFL_WARP_UNCOND_STORE
#define OFL_WARP_UNCOND_STORE 1
#endif
#ifndef OFL_WARP_CLIP_COLUMN
#define OFL_WARP_CLIP_COLUMN 1
#endif
#ifndef OFL_WARP_CLIP
#define OFL_WARP_CLIP 8      // 0: an oversize box gathers the whole tile from global memory; n: stage its first rows if at least n fit
#endif
// Tile shape of the staged kernels: 128 threads x 4 pixels = 32 x 16 output pixels, 26 KB of LDS (6 blocks = 12 waves per CU).
// ofl_warp_wide.hip compiles this file a second time with 256 threads = 64 x 16 tiles and 52 KB (3 blocks = the same 12 waves)
// for the four-tile column kernel of large plain warps only: -1.9 % there, +5 % on the pair kernel of mode 3
// (profiles/r4_warp_16waves.txt), so the shape is chosen per kernel -- by translation unit, not by templating every helper.
#ifndef OFL_LDS_NT
#define OFL_LDS_NT 128
#define OFL_LDS_TWQ 8
#define OFL_LDS_BYTES 26624
#endif
constexpr int kLdsNT = OFL_LDS_NT, kLdsTWQ = OFL_LDS_TWQ, kLdsTH = kLdsNT / kLdsTWQ, kLdsIters = 3;
#ifndef OFL_WARP_WIDE
#define OFL_WARP_WIDE 1
#endif
#ifndef OFL_WARP_T
#define OFL_WARP_T 4
#endif
constexpr int kLdsT = OFL_WARP_T;    // tiles per block of the column kernel (warp_bwd_lds_column_kernel); 2: the pair kernel only
constexpr int kLdsBytes = OFL_LDS_BYTES;   // 6 blocks (12 waves) per CU

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
struct LdsBox { int bx0, miny, cw, Pp, bh, nch, sq, cbase; bool fits, interior, clipped; const uint32_t* ent; int org, cxo; };   // wave-uniform staging geometry (interior: box staged, every tap of every pixel inside the image; clipped: only the first bh rows of an oversize box are staged)

// Y-SHEARED box: a 32-wide tile under a flow with dv/dx != 0 touches a slanted band of source rows, and a plain bounding
// box wastes the two triangles above and below it.  Chunk column c (4 pixels) of the staged box therefore starts at image
// row miny + lds_shear(c, sq), with one block-uniform slope sq (rows per chunk column, Q8).  The box is taken in the
// sheared coordinate y' = y - lds_shear(x >> 2, sq): every slope is correct; a good one makes the box ~12 % smaller and
// boxes that overflow the LDS budget ~10x rarer (2.7 % -> 0.25 % of the tiles of the bench workload).
__device__ __forceinline__ int lds_shear(int c, int sq) { return __mul24(c, sq) >> 8; }

// slope estimate from the flow at the two ends of the row between the block's two tiles (scalar loads: uniform addresses)
template <typename WP>
__device__ __forceinline__ int lds_slope_row(const WP& p, const float* __restrict__ fu, uint32_t hw, int tx, int row);
template <typename WP>
__device__ __forceinline__ int lds_slope(const WP& p, const float* __restrict__ fu, uint32_t hw, int tx, int ty2) {
    return lds_slope_row(p, fu, hw, tx, ty2 * (2 * kLdsTH) + kLdsTH);
}
template <typename WP>
__device__ __forceinline__ int lds_slope_row(const WP& p, const float* __restrict__ fu, uint32_t hw, int tx, int row) {
    const int w = p.w, h = p.h;
    const int y = min(row, h - 1), xa = min(tx * (kLdsTWQ * 4), w - 1), xb = min(xa + kLdsTWQ * 4 - 1, w - 1);
    const float ul = fu[y * w + xa], ur = fu[y * w + xb], vl = fu[hw + y * w + xa], vr = fu[hw + y * w + xb];
    const float dx = (float)(xb - xa) - p.flow_sign * (ur - ul), dy = -p.flow_sign * (vr - vl);
    float q = 1024.0f * dy / dx;                                   // 256 * dy / (dx / 4)
    q = (dx > 4.0f) ? __builtin_amdgcn_fmed3f(q, -4096.0f, 4096.0f) : 0.0f;   // NaN, folds, degenerate spans: no shear
    return __builtin_amdgcn_readfirstlane((int)rintf(q));
}
template <int NC> struct LdsStage { int slot[kLdsIters]; f4 q[kLdsIters][NC]; uint32_t mq[kLdsIters]; };

// steps 1-2 for one tile
// Two halves: _a = coordinates, per-wave box, the wave's (lo, hi) words into `red`; _b = the other waves' words (after a block
// barrier -- the caller's: in the pipelines it is the barrier that publishes the previous tile's staged box anyway) and the
// staging geometry.  lds_coords_box is the two with a barrier of its own between them.
struct LdsBoxWords { int lo, hi; };
template <bool BOX = true, typename WP>
__device__ __forceinline__ LdsBoxWords lds_coords_box_a(const WP& p, int tx, int ty, const f4& u4, const f4& v4, int sq,
                                                        LdsCoords& T, int (*red)[4], int row = -1) {
    constexpr int NW = kLdsNT / 64;
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = row >= 0 ? row : tid / kLdsTWQ;   // (row: the tile row this lane works on, when it is not its own -- chan_tile's row parts)
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
            if (!BOX) continue;
            // west / north tap as an int, clamped to [-2, size] (beyond that every tap is out of range anyway; a NaN
            // coordinate lands on 0 through the conversion and is blended with NaN weights like the reference's)
            const int xi = (int)__builtin_amdgcn_fmed3f(floorf(sx[i]), -2.0f, wf);
            const int yi = (int)__builtin_amdgcn_fmed3f(floorf(sy[i]), -2.0f, hf);
            const int s0 = lds_shear(xi >> 2, sq), s1 = lds_shear((xi + 1) >> 2, sq);   // west / east tap columns
            minx = min(minx, xi); maxx = max(maxx, xi);
            miny = min(miny, yi - max(s0, s1)); maxy = max(maxy, yi + 1 - min(s0, s1));
        }
    }
    if (!BOX) return LdsBoxWords{0, 0};
    // block-wide box: columns and (sheared) rows fit 16 bits (checked on the host), so (x, y) pairs reduce together
    int lo = (int)(((uint32_t)minx & 0xffffu) | ((uint32_t)miny << 16)), hi = (int)(((uint32_t)maxx & 0xffffu) | ((uint32_t)maxy << 16));
    lo = wave_pk_min_dpp(lo); hi = wave_pk_max_dpp(hi);
    if (NW > 1) {
        if ((tid & 63) == 0) { red[(tid >> 6) & (NW - 1)][0] = lo; red[(tid >> 6) & (NW - 1)][1] = hi; }   // (& (NW - 1): a block of several sub-tiles, warp_bwd_lds_chan_kernel<.., SUBS>)
    }
    return LdsBoxWords{lo, hi};
}

template <bool CLIP = false, typename WP>
__device__ __forceinline__ void lds_coords_box_b(const WP& p, int sq, LdsBoxWords wds, LdsBox& B, int (*red)[4]) {
    constexpr int NW = kLdsNT / 64;
    const int w = p.w, h = p.h;
    int lo = wds.lo, hi = wds.hi;
    if (NW > 1) {
#pragma unroll
        for (int i = 0; i < NW; ++i) { lo = pk_min16(lo, red[i][0]); hi = pk_max16(hi, red[i][1]); }
    }
    int minx, miny, maxx, maxy;
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
    B.clipped = false;
    if (CLIP && OFL_WARP_CLIP) {
    // OVERSIZE box (a flow rougher than the budget was sized for): stage the rows that fit and let only the pixels with a tap
    // below them gather from global memory (lds_gather_impl) -- not all 512 of the tile.  Block-uniform, cold.
    if (__builtin_expect(!B.fits && !empty && B.bh <= 4096, 0)) {
        const int keep = min((p.lds_bytes / 16 - 1) / B.Pp, (kLdsIters * kLdsNT) / B.cw);
        if (keep >= OFL_WARP_CLIP) { B.bh = keep; B.nch = keep * B.cw; B.fits = true; B.clipped = true; }
    }
    }
    // image rows the box's first and last chunk column cover (the shear is monotonic in the column)
    const int sa = lds_shear(B.cbase, sq), sb_ = lds_shear(B.cbase + B.cw - 1, sq);
    B.interior = B.fits && !B.clipped && xin && (miny + min(sa, sb_) >= 0) && (maxy + max(sa, sb_) <= h - 1);
}

template <bool BOX = true, bool CLIP = false, typename WP>
__device__ __forceinline__ void lds_coords_box(const WP& p, int tx, int ty, const f4& u4, const f4& v4, int sq,
                                               LdsCoords& T, LdsBox& B, int (*red)[4], int row = -1) {
    const LdsBoxWords wds = lds_coords_box_a<BOX>(p, tx, ty, u4, v4, sq, T, red, row);
    if (!BOX) return;
    if (kLdsNT > 64) lds_barrier();
    lds_coords_box_b<CLIP>(p, sq, wds, B, red);
}

// step 3a: issue the staging loads of a tile into registers (nothing waits here)
template <int NC, bool VALID, bool SUB = false, typename TS = float, typename WP = WarpParams>
__device__ __forceinline__ void lds_issue(const WP& p, const TS* __restrict__ sb, const uint8_t* __restrict__ sm,
                                          uint32_t hw, const LdsBox& B, LdsStage<NC>& S, const float* __restrict__ sbb = nullptr) {
    const int tid = threadIdx.x;
    const uint32_t inv = inv20((uint32_t)B.cw);
    const int rounds = B.fits ? (B.nch + kLdsNT - 1) / kLdsNT : 0;
#pragma unroll
    for (int it = 0; it < kLdsIters; ++it) {
        S.slot[it] = -1;
        // the first OFL_WARP_UNCOND_ROUNDS rounds are issued whatever the box (lanes past its last chunk re-read chunk 0 and
        // write nothing to the LDS): loads in straight-line code are loads the compiler can COUNT, so the wait for the next
        // tile's flow (older than these loads) no longer waits for them too -- its coordinates are computed while they fly
        if (it < OFL_WARP_UNCOND_ROUNDS || it < rounds) {
            const uint32_t i = (uint32_t)tid + it * kLdsNT;
            // 24-bit multiplies (full rate): i < 2^9, inv <= 2^20, rows and columns < 2^13, h * w < 2^24
            const uint32_t r = __umul24(i, inv) >> 20, c4 = i - __umul24(r, (uint32_t)B.cw);
            const int y = B.miny + (int)r + lds_shear(B.cbase + (int)c4, B.sq);
            const bool on = (i < (uint32_t)B.nch) && ((uint32_t)y < (uint32_t)p.h) && (OFL_WARP_UNCOND_ROUNDS == 0 || it < rounds);
            const uint32_t g = on ? (uint32_t)(__mul24(y, p.w) + B.bx0) + c4 * 4u : 0u;
            S.slot[it] = on ? 1 + (int)(__umul24(r, (uint32_t)B.Pp) + c4) : -1;
            // the last chunk of a row of an image whose width is not a multiple of 4 would read past the row end (and past
            // the buffer on the last row): fetch the last whole group instead and rotate (block-uniform branch)
            const int wrem = OFL_WP_WREM(p);
            const bool edge = wrem != 0 && on && (int)(B.bx0 + (int)c4 * 4) > p.w - 4;
            const uint32_t ge = edge ? g - (uint32_t)(4 - wrem) : g;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                S.q[it][c] = ld4(sb + c * hw + ge);
                if (SUB) S.q[it][c] = S.q[it][c] - ld4(sbb + c * hw + ge);     // (mode 1 't': the warped field is flow - self)
            }
            // (OFL_WARP_UNCOND_ROUNDS, no target mask: the load reads the first plane's bytes and lds_write discards them -- still one countable load)
            if (OFL_WARP_UNCOND_ROUNDS && VALID) S.mq[it] = ld32(sm ? sm + ge : reinterpret_cast<const uint8_t*>(sb) + ge);
            else S.mq[it] = (VALID && sm) ? ld32(sm + ge) : 0x01010101u;
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
__device__ __forceinline__ void lds_write(f4* lds, const LdsBox& B, const LdsStage<NC>& S, bool has_sm = true) {
    if (threadIdx.x == 0) lds[0] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < kLdsIters; ++it) {
        if (S.slot[it] >= 0) {
            const uint32_t mq = (OFL_WARP_UNCOND_ROUNDS && VALID && !has_sm) ? 0x01010101u : S.mq[it];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f4 sl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NC; ++c) sl[c] = S.q[it][c][k];
                if (VALID) sl[3] = fminf((float)((mq >> (8 * k)) & 0xffu), 1.0f);   // non-zero byte -> 1 (v_cvt_f32_ubyteK + v_min)
                lds[S.slot[it] + k * B.cw] = sl;
            }
        }
    }
}

// ---- switches of the ROW-TABLE kernels (warp_bwd_rows_kernel and the ROWS instantiation of the channel loop, below; profiles/r5_warp_row_extents.txt)
constexpr int kRowTab = 64;                    // rows of a tile's row table
#ifndef OFL_ROWS_DEDUPE
#define OFL_ROWS_DEDUPE 1     // neighbouring lanes with the same rows post once (DPP): -3 ... -6 % under smooth flows
#endif
#ifndef OFL_ROWS_PAD
#define OFL_ROWS_PAD 0        // 1: rows padded to the rectangle's pitch rule -- measured slower everywhere (+2 ... 6 %: more LDS, no fewer conflicts -- the rows' own start columns already scatter them)
#endif
#ifndef OFL_ROWS_STAMPS
#define OFL_ROWS_STAMPS 0     // 1: s_memtime stamps per phase, summed per block (tools/rows_stamps.py; never in a default build)
#endif
#ifndef OFL_WARP_ROWS_ADD
#define OFL_WARP_ROWS_ADD 1   // mode 3 (the addend is the flow operand) on the row-table kernel
#endif
#ifndef OFL_ROWS_T
#define OFL_ROWS_T 4          // tiles per column of large launches (2 and 3 measured: slower at B = 8 and at B = 64; 6 and 8, with an early exit below the frame: apply level, mode 3 +3 ... 5 % -- the prologue is not what is left)
#endif
#ifndef OFL_ROWS_ADD_REFORM
#define OFL_ROWS_ADD_REFORM 0 // 1: mode 3 re-forms a tile's positions from the flow registers at gather time instead of keeping them (+6 % time)
#endif
#ifndef OFL_WARP_ROWS_FLOWOPS
#define OFL_WARP_ROWS_FLOWOPS 1      // the other instantiations on row tables too: another addend (modes 1-2, Flow.combine), src - src_b staging (mode 1 't'), the output's flag word, fp16 / uint8 sources, the gradient with respect to the flow
#endif
#ifndef OFL_WARP_CHAN_WIDE_MIN
#define OFL_WARP_CHAN_WIDE_MIN (2 * 6912u)      // 32 x 16 tiles of the launch from which the channel loop on the RECTANGLE runs on 64 x 16 tiles (with row extents: every size)
#endif
#ifndef OFL_ROWS_T1_MAX
#define OFL_ROWS_T1_MAX 5000u      // 32 x 16 tiles of a launch below which the row-table kernel runs one tile per block (above: two, up to the column threshold); tools/shapes_once.py: 3 600 / 4 096 tiles 3 % better with one, 6 144 tiles 13 % better with two
#endif
#ifndef OFL_ROWS_T4_MIN
#define OFL_ROWS_T4_MIN 5800u      // four-tile column groups (32-wide geometry) from which the row-table kernel runs four tiles per block (below: two / one)
#endif
#ifndef OFL_WARP_ROWS_SMALL
#define OFL_WARP_ROWS_SMALL 1       // small plain launches and small mode 3 on the row-table kernel too (1 or 2 tiles per block)
#endif

// step 4a: gather from LDS (or from global memory when the box did not fit) and blend; per pixel (c0, c1, c2, mask channel)
// GRAD (ofl_warp_bwd_grad_f32, gradient with respect to the FLOW): instead of the blend, the taps of a pixel are combined
// with its upstream gradient `gq` into ATen's gix / giy sums (grid_sampler_2d_backward), chained through the
// un-normalisation, normalise_coords and `grid - flow` exactly as autograd does -- same expressions, same order as the
// one-pixel-per-lane kernel of ofl_aux_kernels.hip (the two are compared bit for bit); outv[k] = (d/du, d/dv, -, -).
template <int NC, bool VALID, bool INTERIOR, bool SUB = false, typename TS = float, bool GRAD = false, bool CLIP = false, typename WP = WarpParams, bool ROWS = false>
__device__ __forceinline__ void lds_gather_impl(const WP& p, uint32_t hw,
                                                const TS* __restrict__ sb, const uint8_t* __restrict__ sm,
                                                const LdsCoords& T, const LdsBox& B, const unsigned char* smem, f4 (&outv)[4],
                                                const float* __restrict__ sbb = nullptr, const f4* gq = nullptr) {
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
        if (ROWS) {
            // ROW TABLE (warp_bwd_rows_kernel): image rows yi, yi + 1 each have their own start chunk and length; entry = byte address
            // of the row's slot for chunk (cxo + 128), biased, | 16 * length << 16 (0: row not staged; 1: a row without valid taps)
            const int yr = yi - B.org;
            const bool inr = INTERIOR || (uint32_t)yr < (uint32_t)(kRowTab - 1);
            const int yrc = inr ? yr : 0;
            const uint32_t e0 = B.ent[yrc], e1 = B.ent[yrc + 1];
            const bool staged = INTERIOR || (inr && e0 != 0u && e1 != 0u) || !(ok[0] || ok[1] || ok[2] || ok[3]);
            if (staged) {
                const uint32_t m0 = (uint32_t)xi & 3u, m1 = (uint32_t)(xi + 1) & 3u;
                const int q0 = (((xi >> 2) - B.cxo) << 4) - 4096, q1 = ((((xi + 1) >> 2) - B.cxo) << 4) - 4096;
                const int si[4] = {ok[0] ? (int)(e0 & 0xffffu) + (int)__umul24(m0, e0 >> 16) + q0 : 0, ok[1] ? (int)(e0 & 0xffffu) + (int)__umul24(m1, e0 >> 16) + q1 : 0,
                                   ok[2] ? (int)(e1 & 0xffffu) + (int)__umul24(m0, e1 >> 16) + q0 : 0, ok[3] ? (int)(e1 & 0xffffu) + (int)__umul24(m1, e1 >> 16) + q1 : 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) tv[j] = *reinterpret_cast<const f4*>(smem + si[j]);
            }
            if (!staged) {
                // (a row outside the table or beyond the block's chunk budget: the pixel's taps from global memory, as for an oversize box)
                const int xc = min(max(xi, 0), w - 2);
                const int ew = xi - xc, ee = xi + 1 - xc;
                const int yrr[2] = {min(max(yi, 0), h - 1), min(max(yi + 1, 0), h - 1)};
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t og = (uint32_t)(yrr[r] * w + xc);
                    f4 tw = {0.f, 0.f, 0.f, 0.f}, te = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        f2 pr = ld2(sb + c * hw + og);
                        if (SUB) pr = pr - ld2(sbb + c * hw + og);                 // (mode 1 't': the warped field is flow - self)
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
            }
        } else {
        const int yr = yi - B.miny;   // row in the sheared box, per tap column
        const int ra = yr - lds_shear(xi >> 2, B.sq), rb = yr - lds_shear((xi + 1) >> 2, B.sq);
        // (clipped box: a pixel whose lower taps fall below the staged rows takes the global path below, the others the LDS)
        const bool staged = INTERIOR || ((CLIP && OFL_WARP_CLIP) ? (B.fits && (!B.clipped || max(ra, rb) + 1 < B.bh)) : __builtin_expect(B.fits, 1));
        if (staged) {
            const int xl0 = xi - B.bx0, xl1 = xl0 + 1;
            const int cp0 = __mul24(xl0 & 3, cw16) + ((xl0 & ~3) << 2), cp1 = __mul24(xl1 & 3, cw16) + ((xl1 & ~3) << 2);
            const int r0 = 16 + __mul24(ra, P16), r1 = 16 + __mul24(rb, P16);
            int si[4] = {ok[0] ? r0 + cp0 : 0, ok[1] ? r1 + cp1 : 0, ok[2] ? r0 + P16 + cp0 : 0, ok[3] ? r1 + P16 + cp1 : 0};
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
        }
        if (GRAD) {
            float gix = 0.0f, giy = 0.0f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float g = p.g_sign * gq[c][k];                       // (g_sign carries the upstream scale)
                gix -= tv[0][c] * s * g; gix += tv[1][c] * s * g; gix -= tv[2][c] * nn * g; gix += tv[3][c] * nn * g;
                giy -= tv[0][c] * e * g; giy -= tv[1][c] * ww * g; giy += tv[2][c] * e * g; giy += tv[3][c] * ww * g;
            }
            outv[k] = (f4){-p.flow_sign * (((gix * p.half_wm1) / p.wm1) * 2.0f), -p.flow_sign * (((giy * p.half_hm1) / p.hm1) * 2.0f), 0.f, 0.f};
            continue;
        }
        // v_nw*nw, then fma(v_ne, ne, .), fma(v_sw, sw, .), fma(v_se, se, .): the reference's contraction order
        f4 r = tv[0] * wg[0];
        r = __builtin_elementwise_fma(tv[1], (f4){wg[1], wg[1], wg[1], wg[1]}, r);
        r = __builtin_elementwise_fma(tv[2], (f4){wg[2], wg[2], wg[2], wg[2]}, r);
        r = __builtin_elementwise_fma(tv[3], (f4){wg[3], wg[3], wg[3], wg[3]}, r);
        outv[k] = r;
    }
}

template <int NC, bool VALID, bool SUB = false, typename TS = float, bool GRAD = false, bool CLIP = false, typename WP = WarpParams, bool ROWS = false>
__device__ __forceinline__ void lds_gather(const WP& p, uint32_t hw,
                                           const TS* __restrict__ sb, const uint8_t* __restrict__ sm,
                                           const LdsCoords& T, const LdsBox& B, const unsigned char* smem, f4 (&outv)[4],
                                           const float* __restrict__ sbb = nullptr, const f4* gq = nullptr) {
    if (B.interior) lds_gather_impl<NC, VALID, true, SUB, TS, GRAD, false, WP, ROWS>(p, hw, sb, sm, T, B, smem, outv, sbb, gq);
    else lds_gather_impl<NC, VALID, false, SUB, TS, GRAD, CLIP, WP, ROWS>(p, hw, sb, sm, T, B, smem, outv, sbb, gq);
}

// the fused addend of a tile (mode 3), loaded ahead of younger loads and stores: the wait for it must not cover them
template <int NC, typename WP = WarpParams>
__device__ __forceinline__ void lds_load_addend(const WP& p, int tx, int ty, int n, uint32_t hw, f4 (&a)[NC]) {
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const uint32_t pix = (uint32_t)(min(ty * kLdsTH + ly, p.h - 1) * p.w + min(tx * (kLdsTWQ * 4) + lx * 4, p.w - 4));
#pragma unroll
    for (int c = 0; c < NC; ++c) a[c] = ld4(p.addend + n * p.addend_bs + c * hw + pix);
}

// step 4b: valid mask, epilogue (a_sign * addend + g_sign * G, rounding), 16-byte stores
template <int NC, bool VALID, bool ADD, bool DF = false, typename TD = float, typename WP = WarpParams>
__device__ __forceinline__ void lds_store(const WP& p, int tx, int ty, int n, uint32_t hw, uint32_t fmask4,
                                          const f4 (&outv)[4], const f4 (&addend)[NC], int* dflags = nullptr, int64_t choff = 0) {
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const int w = p.w, h = p.h;
    const int x4 = tx * (kLdsTWQ * 4) + lx * 4, y = ty * kLdsTH + ly;
    // OFL_WARP_UNCOND_STORE: lanes past the frame's right / bottom edge have computed the CLAMPED pixel group (same flow, same
    // coordinates, same taps as its owner) and store it again -- identical duplicate stores, as for the group that straddles a
    // row end.  What it buys is not the branch: with the stores in straight-line code the compiler can COUNT them, so the next
    // tile's wait for its flow / staged box becomes s_waitcnt vmcnt(4) instead of vmcnt(0), which also waited for these stores'
    // acknowledgements (the vmcnt queue is in order) -- once per tile, with nothing else of the block in flight.
    const bool inb = OFL_WARP_UNCOND_STORE || ((x4 < w) && (y < h));
    const uint32_t pix = (uint32_t)(min(y, h - 1) * w + min(x4, w - 4));
    if (inb) {
        uint32_t vo = 0x01010101u;
        if (VALID) {
            vo = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                vo |= (uint32_t)((outv[k][3] > kValidThr) && (((fmask4 >> (8 * k)) & 0xffu) != 0u)) << (8 * k);
            st32o(p.valid + (int64_t)n * hw + pix, vo);
        }
        TD* __restrict__ db = reinterpret_cast<TD*>(p.dst) + (int64_t)n * p.dst_bs + choff;   // (choff: first plane of a channel group, warp_bwd_lds_chan_kernel)
        f4 o01[2];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f4 o = {outv[0][c], outv[1][c], outv[2][c], outv[3][c]};
            if (ADD) o = addend[c] * p.a_sign + o * p.g_sign;
            if (OFL_WP_ROUND(p) != OFL_ROUND_NONE) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = apply_round(o[k], OFL_WP_ROUND(p));
            }
            st4o(db + c * hw + pix, o);
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
    // (OFL_WARP_ALWAYS_T: the second tile is run whatever the frame's height -- past the bottom edge it recomputes and re-stores the
    // last row, see the column kernel -- so that its staging loads and both tiles' stores are straight-line code the compiler counts)
    const bool haveB = OFL_WARP_ALWAYS_T || tyB * kLdsTH < p.h;
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
    // ... while B's coordinates are computed; the block-wide half of B's box waits for the barrier that publishes A's staged box
    // (OFL_WARP_MERGE_BARRIER: one barrier per tile fewer)
#if OFL_WARP_MERGE_BARRIER
    const LdsBoxWords wB = lds_coords_box_a(p, tx, tyB, uB, vB, sq, TB, red[1]);
#else
    lds_coords_box(p, tx, tyB, uB, vB, sq, TB, BB, red[1]);
#endif
    lds_write<NC, VALID>(lds, BA, S, sm != nullptr);
    lds_barrier();
#if OFL_WARP_MERGE_BARRIER
    lds_coords_box_b(p, sq, wB, BB, red[1]);
#endif
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
    lds_write<NC, VALID>(lds, BB, S, sm != nullptr);
    lds_barrier();
    lds_gather<NC, VALID, SUB, TS>(p, hw, sb, sm, TB, BB, smem, outv, sbb);
    if (ADD && !EARLY) lds_load_addend<NC>(p, tx, tyB, n, hw, aB);
    lds_store<NC, VALID, ADD, DF, TD>(p, tx, tyB, n, hw, fmB, outv, aB, &dflags);
    if (DF) { dflags = wave_or_flags(dflags); if ((tid & 63) == 0) flag_or(&p.dst_flags[n], dflags); }
}

// The same pipeline over T vertically adjacent tiles (T >= 3, and T = 1 for tiny launches: one tile per block, nothing to
// overlap inside it; the hand-scheduled two-tile kernel above stays the one for T = 2: written as this loop it leaves its small
// arrays in scratch).  The two dependent round trips of a tile (flow, then
// its staged box) and the drain of its stores are paid once per BLOCK: a taller column of tiles amortises them over more
// pixels with the same registers and the same LDS -- flow two tiles ahead, staging loads one tile ahead, stores behind.
// GRAD: the same column pipeline computing the gradient with respect to the flow (`addend` = the upstream gradient [N,NC,H,W],
// `dst` = [N,2,H,W]; see lds_gather_impl) -- the forward's staged boxes instead of 4 * NC scalar gathers per pixel.
// REUSE (the fused composition proper: ADD with the addend being the flow operand itself, 2 channels): the flow registers of a tile
// ARE its addend and stay live until its store, so the tile's sample positions are not kept from the box phase to the gather (8
// VGPRs per tile, two tiles live) but re-formed from those registers at gather time -- ~40 VALU per tile for the ~16 registers that
// were the kernel's scratch traffic.
template <int T, int NC, bool VALID, bool ADD, bool DF = false, bool SUB = false, typename TS = float, typename TD = float, bool GRAD = false, bool REUSE = false, bool LEAN = false>
__global__ __launch_bounds__(kLdsNT, 3) void warp_bwd_lds_column_kernel(const WarpParams p_by_value) {
    static_assert(!REUSE || (ADD && NC == 2 && !GRAD), "REUSE: the fused composition");
#if OFL_WARP_KARG
    // parameters through the kernarg segment (see OFL_OPAQUE_S at the gather splat): the ~250 bytes of WarpParams are not
    // held in SGPRs (and spilled to VGPR lanes) across the whole column, each phase s_loads what it needs
    typedef typename std::conditional<LEAN, WarpParamsLeanK, WarpParamsK>::type WPK;       // (LEAN: see WarpParamsLean)
    WPK* pp = (WPK*)__builtin_amdgcn_kernarg_segment_ptr();
#define p (*pp)
#define OFL_WARP_PHASE() OFL_OPAQUE_S(pp)
#else
    const WarpParams& p = p_by_value;
#define OFL_WARP_PHASE()
#endif
    constexpr int NW = kLdsNT / 64;
    constexpr bool kClip = OFL_WARP_CLIP_COLUMN && T > 1;   // oversize boxes: stage the rows that fit (lds_coords_box); not in the one-tile kernel of tiny launches (+2 % there)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int red[2][NW][4];
    int tx, tyg, n;                      // the grid counts tile GROUPS: tiles_y = ceil(h / (T * kLdsTH))
    if (!decode_tile(p, tx, tyg, n)) return;
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const int w = p.w, h = p.h;
    const uint32_t hw = (uint32_t)(h * w);
    const float* __restrict__ fu = p.flow + n * p.flow_bs;
    const TS* __restrict__ sb = reinterpret_cast<const TS*>(p.src) + n * p.src_bs;      // (uint8 variants: p.src / p.dst point at bytes)
    const float* __restrict__ sbb = SUB ? p.src_b + n * p.src_b_bs : nullptr;
    const uint8_t* __restrict__ sm = p.src_mask ? p.src_mask + n * p.src_mask_bs : nullptr;
    const uint8_t* __restrict__ fm = p.flow_mask ? p.flow_mask + n * p.flow_mask_bs : nullptr;
    const int x4 = tx * (kLdsTWQ * 4) + lx * 4, xq = min(x4, w - 4);
    // flow + flow mask of a tile (16-byte non-temporal loads); issued two tiles ahead of their use
    f4 uu[T], vv[T];
    uint32_t fmk[T];
    int fflags = 0;
    auto load_flow = [&](int k) {
        const uint32_t pix = (uint32_t)(min((tyg * T + k) * kLdsTH + ly, h - 1) * w + xq);
        uu[k] = ld4nt(fu + pix); vv[k] = ld4nt(fu + hw + pix);
        fmk[k] = 0x01010101u;
        // (OFL_WARP_UNCOND_ROUNDS: countable, as the staging loads: no flow mask -> a read of the flow's bytes, discarded where the word is used)
        if (OFL_WARP_UNCOND_ROUNDS && VALID) fmk[k] = ld32(fm ? fm + pix : reinterpret_cast<const uint8_t*>(fu) + pix);
        else if ((VALID || OFL_WP_FLOW_FLAGS(p)) && fm) fmk[k] = ld32(fm + pix);
    };
    auto fmw = [&](int k) -> uint32_t { return (OFL_WARP_UNCOND_ROUNDS && VALID && !fm) ? 0x01010101u : fmk[k]; };
    auto note_flags = [&](int k) {       // finiteness / zero tests of the flow operand as a by-product (wave-uniform branch)
        if (OFL_WP_FLOW_FLAGS(p) && (x4 < w) && ((tyg * T + k) * kLdsTH + ly < h)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fflags |= flag_bits(uu[k][q], vv[k][q], ((fmw(k) >> (8 * q)) & 0xffu) != 0u);
        }
    };
    load_flow(0);
    if (T > 1) load_flow(1);
    f4* lds = reinterpret_cast<f4*>(smem);
    LdsCoords Tc[REUSE ? 1 : T];
    LdsBox Bx[T];
    LdsStage<NC> S;
    // the shear slope is estimated per tile (a column is too tall for one estimate); all of them up front: the scalar loads
    // must not sit between a tile's flow and its box
    int sq[T];
#pragma unroll
    for (int k = 0; k < T; ++k) sq[k] = OFL_WP_SHEAR(p) ? lds_slope_row(p, fu, hw, tx, (tyg * T + k) * kLdsTH + kLdsTH / 2) : 0;
    note_flags(0);
    lds_coords_box<true, kClip>(p, tx, tyg * T, uu[0], vv[0], sq[0], Tc[0], Bx[0], red[0]);
    lds_issue<NC, VALID, SUB, TS>(p, sb, sm, hw, Bx[0], S, sbb);          // staging loads of tile 0 fly ...
    // the vmcnt queue is in order: the addend (an L2 hit when it is the flow itself) is fetched BEFORE the next tile's staging
    // loads / this tile's stores, so that waiting for it never waits for them
    // (flows only: with three channels the extra registers would spill, and nothing on the host adds to an image)
    constexpr bool EARLY = ADD && NC <= 2;
    const bool reuse = REUSE || (EARLY && NC == 2 && p.add_is_flow);      // block-uniform
    int dflags = 0;
#pragma unroll
    for (int k = 0; k < T; ++k) {
        OFL_WARP_PHASE();
        const int tyk = tyg * T + k;
        // OFL_WARP_ALWAYS_T: a block runs all T tiles of its column whatever the frame's height -- a tile past the bottom edge
        // recomputes and re-stores the frame's last row (clamped loads, identical duplicate stores), at most T - 1 tiles of
        // the last row of groups.  No run-time branch round the staging loads or the stores is left, so the compiler counts
        // every one of them in its vmcnt waits (see lds_store).
        // (not with the ADD epilogue: its re-used flow registers leave no room for the longer live ranges -- 36 B of scratch, -13 %)
        const bool more = (k + 1 < T) && ((OFL_WARP_ALWAYS_T && (!ADD || (REUSE && OFL_WARP_ALWAYS_T >= 2))) || (tyk + 1) * kLdsTH < h);        // block-uniform: a tile follows
        LdsBoxWords wn = {0, 0};
        if (k + 1 < T) {
            // ... while the next tile's coordinates are computed (OFL_WARP_MERGE_BARRIER: the block-wide half of its box behind the
            // barrier that publishes this tile's staged box -- one barrier per tile fewer)
#if OFL_WARP_MERGE_BARRIER
            if (more) { note_flags(k + 1); wn = lds_coords_box_a(p, tx, tyk + 1, uu[k + 1], vv[k + 1], sq[k + 1], Tc[REUSE ? 0 : k + 1], red[(k + 1) & 1]); }
#else
            if (more) { note_flags(k + 1); lds_coords_box<true, kClip>(p, tx, tyk + 1, uu[k + 1], vv[k + 1], sq[k + 1], Tc[REUSE ? 0 : k + 1], Bx[k + 1], red[(k + 1) & 1]); }
#endif
        }
        if (k + 2 < T) load_flow(k + 2);
        lds_write<NC, VALID>(lds, Bx[k], S, sm != nullptr);
        lds_barrier();
#if OFL_WARP_MERGE_BARRIER
        if (k + 1 < T) { if (more) lds_coords_box_b<kClip>(p, sq[k + 1], wn, Bx[k + 1], red[(k + 1) & 1]); }
#endif
        f4 outv[4], ad[NC];
        if (EARLY) { if (reuse) { ad[0] = uu[k]; ad[NC - 1] = vv[k]; } else lds_load_addend<NC>(p, tx, tyk, n, hw, ad); }
        if (GRAD) lds_load_addend<NC>(p, tx, tyk, n, hw, ad);              // the upstream gradient of the tile, ahead of the younger loads
        if (k + 1 < T) {
            if (more) lds_issue<NC, VALID, SUB, TS>(p, sb, sm, hw, Bx[k + 1], S, sbb);   // the next tile's staging loads fly while this one is gathered and stored
        }
        if (REUSE) lds_coords_box_a<false>(p, tx, tyk, uu[k], vv[k], 0, Tc[0], red[0]);   // (the positions again, from the registers that are the addend)
        lds_gather<NC, VALID, SUB, TS, GRAD, kClip>(p, hw, sb, sm, Tc[REUSE ? 0 : k], Bx[k], smem, outv, sbb, ad);
        if (GRAD) {
            const f4 none[2] = {};
            lds_store<2, false, false, false, float>(p, tx, tyk, n, hw, 0u, outv, none);
            if (!more) break;
            lds_barrier();
            continue;
        }
        if (ADD && !EARLY) lds_load_addend<NC>(p, tx, tyk, n, hw, ad);
        lds_store<NC, VALID, ADD, DF, TD>(p, tx, tyk, n, hw, fmw(k), outv, ad, &dflags);
        if (!more) break;
        lds_barrier();
    }
    if (OFL_WP_FLOW_FLAGS(p)) {
        fflags = wave_or_flags(fflags);
        if ((tid & 63) == 0) flag_or(&p.flow_flags[n], fflags);
    }
    if (DF) { dflags = wave_or_flags(dflags); if ((tid & 63) == 0) flag_or(&p.dst_flags[n], dflags); }
#undef OFL_WARP_PHASE
#if OFL_WARP_KARG
#undef p
#endif
}

// ROW TABLES (OFL_WARP_ROWS; VERDICT r4 item 2: "per-row extents of the staged box"; profiles/r5_warp_row_extents.txt).  The column
// kernel above stages ONE y-sheared rectangle per tile; under the bench flow (sigma 8) it holds 1.65 source pixels per output pixel of a
// 64 x 16 tile, and a rougher flow overflows it (10.8 % of the tiles at sigma 12, 27.5 % at sigma 16).  Here every source ROW the tile
// touches has its own first chunk and length, the rows packed back to back in LDS (tools/box_rows_model_wide3.py: 1.27 staged pixels per
// output pixel at sigma 8, 1.35 / 1.45 at sigma 12 / 16, and what does not fit is a handful of PIXELS, not tiles):
//   post   every lane puts the chunk range of its 4 pixels' taps on the image rows they touch: ds_min / ds_max on a 64-row table whose
//          first row `org` is an ESTIMATE (the flow at 3 x 3 points of each tile of the column: ONE vector load, rows_origins) -- so
//          that no block-wide reduction has to come first.  A row outside the table is dropped and flagged; neighbouring lanes with
//          the same rows post once (DPP).
//   scan   after the barrier that publishes the previous tile's staged data anyway, every wave reads the table (lane = row), clips the
//          ranges to the frame, prefix-sums the lengths (DPP), writes one 32-bit ENTRY per row for the gather (biased byte address of
//          the row's slot for a reference chunk | 16 * length << 16; 0: not staged, 1: no valid tap) and marks the rows' first chunks
//          in a byte map.  Rows beyond the block's kRowChunks chunks are not staged.
//   issue  chunk i of the packed rows -> thread i (dense): the row of a chunk = prefix MAXIMUM (DPP) of the start marks of the wave's
//          64 chunks, the row that holds the wave's first chunk (ballot + popcount) as the carry; the row's start / length by
//          ds_bpermute from the scan's registers.  Two rounds through registers, in flight while the previous tile is gathered; the
//          rare third is loaded and written on the spot (rows_extra).
//   gather two entries per pixel (rows yi, yi + 1: one ds_read2_b32), address = entry + (x & 3) * length16 + 16 * (x >> 2): no shear,
//          no multiply by a pitch.  A pixel with an unstaged row takes its taps from global memory (as for an oversize box).
// Same expressions per pixel as the column kernel: bit-identical.  64 x 16 tiles (ofl_warp_wide.hip); launches with W % 4 == 0 and no
// flow-flag by-product: four tiles per block for large ones, one or two for small ones (ofl_wide_launch_column / _rows_small / _rows_h /
// _rows_u8 / _rows_grad, and the channel loop's ROWS instantiation).
#ifndef OFL_WARP_ROWS
#define OFL_WARP_ROWS 1
#endif
constexpr int kRowChunks = kLdsIters * kLdsNT - 16;             // chunks (4 pixels, one 16-byte slot each) a block stages per tile (3 blocks per CU: 53.3 KB with the tables below)
constexpr int kRowsLdsBytes = 16 * (2 + 4 * kRowChunks);        // slot 0 (zeros: invalid taps) + the packed rows + a spare slot (rows_extra)
static_assert(kRowChunks % 4 == 0, "clear_starts zeroes the chunk map a dword at a time");
static_assert(16 * (1 + 4 * kRowChunks + 128 + 256) < 65536, "a row entry packs a biased byte address and 16 x length in 16 bits each");
constexpr int kRowMargin = 8;
struct RowTabs { int tmin[2][kRowTab + 1]; int tmax[2][kRowTab + 1]; uint32_t ent[2][kRowTab]; uint8_t start[2][kRowChunks]; };   // ([kRowTab]: "a post was dropped"; start[i]: 1 + the row whose first chunk is chunk i, else 0)
struct RowGeo { int org, cxo, tot; bool interior; };            // wave-uniform
struct RowScan { int incl, sc, a; };                         // lane = row: chunks up to and including this row; the same | padded chunks << 16; first chunk | length << 16
__device__ __forceinline__ int rows_pitch(int cw) { return OFL_ROWS_PAD ? (cw ? lds_pitch(4 * cw) >> 2 : 0) : cw; }   // LDS chunks (4 slots) a row of cw chunks occupies

__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3
    return v;
}

// first table row and reference chunk column of the T tiles of a column, from the flow at 3 x 3 points of each: ONE vector load
// (tile k in the 16 lanes of DPP row k: lanes 0..8 its nine samples of v, lane 9 a sample of u), a row minimum, v_readlane.
// (First cut: 10 scalar loads per tile.  The compiler chained them -- load, wait, min, next load -- and the 40 round trips were 30 %
// of a block's life in the phase stamps.)  Issued before the flow loads: older in the vmcnt queue, waited for without them.
template <int T, int K0 = 0, typename WP>
__device__ __forceinline__ float rows_origins_load(const WP& p, const float* __restrict__ fu, uint32_t hw, int tx, int tyg) {
    static_assert(T <= 8, "one DPP row per tile: four tiles per load (K0: the first of them)");
    const int l = threadIdx.x & 63, k = K0 + (l >> 4), j = min(l & 15, 9);
    const int w = p.w, h = p.h;
    const int sy = j < 9 ? j / 3 : 1, sx = j < 9 ? j % 3 : 0;
    const int x = min(tx * (kLdsTWQ * 4) + (sx == 0 ? 0 : sx == 1 ? kLdsTWQ * 2 : kLdsTWQ * 4 - 1), w - 1);
    const int y = min((tyg * T + min(k, T - 1)) * kLdsTH + (sy == 0 ? 0 : sy == 1 ? kLdsTH / 2 : kLdsTH - 1), h - 1);
    const float f = fu[(j < 9 ? hw : 0u) + (uint32_t)(y * w + x)];
    return (j < 9 ? (float)y : (float)x) - p.flow_sign * f;          // sample row (lanes 0..8) / sample column (lane 9 and its duplicates)
}
template <int T, int K0 = 0, typename WP>
__device__ __forceinline__ void rows_origins(const WP& p, float s, int (&org)[T], int (&cxo)[T]) {
    const int j = threadIdx.x & 15;
    float m = j < 9 ? s : 3.0e38f;
#define OFL_FMIN_DPP(ctrl) m = fminf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), (ctrl), 0xf, 0xf, false)))
    OFL_FMIN_DPP(0xB1); OFL_FMIN_DPP(0x4E); OFL_FMIN_DPP(0x141); OFL_FMIN_DPP(0x140);
#undef OFL_FMIN_DPP
#pragma unroll
    for (int k = K0; k < (T < K0 + 4 ? T : K0 + 4); ++k) {
        const float mn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 16 * (k - K0)));
        const float xl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 16 * (k - K0) + 9));
        org[k] = (int)floorf(__builtin_amdgcn_fmed3f(mn, -2.0f, (float)p.h)) - kRowMargin;
        cxo[k] = ((int)floorf(__builtin_amdgcn_fmed3f(xl, -2.0f, (float)p.w)) >> 2) + 16;
    }
}

template <typename WP>
__device__ __forceinline__ void rows_post(const WP& p, const LdsCoords& T, int org, int* tmin, int* tmax) {
    const float wf = (float)p.w, hf = (float)p.h;
    int xlo = 0x7fffffff, xhi = -0x7fffffff, ylo = 0x7fffffff, yhi = -0x7fffffff;
#pragma unroll
    for (int k = 0; k < 4; ++k) {          // (the clamps of the gather: beyond [-2, size] every tap is out of range anyway)
        const int xi = (int)__builtin_amdgcn_fmed3f(floorf(T.sx[k]), -2.0f, wf), yi = (int)__builtin_amdgcn_fmed3f(floorf(T.sy[k]), -2.0f, hf);
        xlo = min(xlo, xi); xhi = max(xhi, xi); ylo = min(ylo, yi); yhi = max(yhi, yi);
    }
    const int cmin = xlo >> 2, cmax = (xhi + 1) >> 2;
    const int a = ylo - org, b = yhi + 1 - org;             // the table rows this lane's taps touch
    // Under a smooth flow the 16 lanes of a tile row all post to the same two or three rows (same-address LDS atomics).  A lane
    // therefore leaves the minimum to its LEFT neighbour when that one touches the same rows and starts no later, and the maximum to
    // its RIGHT neighbour likewise: by induction along the 16 lanes (a DPP row) somebody with the rows' extreme value posts it, whatever
    // the flow -- under a smooth one only the two ends of each run do.  (Measured: -3 ... -6 % at sigma 0.5; the general form of the
    // test -- the neighbour's range CONTAINS the row, per row -- cost sigma 8 +2 %.)
    // (rows outside the table are dropped and flagged; clamped to it first, so that a lane whose 4 pixels span the frame -- a fold, a
    // huge value -- walks 64 rows at most, and the packed compare below cannot alias)
    const bool drop = a < 0 || b >= kRowTab;
    const int a0 = max(a, 0), b0 = min(b, kRowTab - 1);
#if OFL_ROWS_DEDUPE
    const int ab = a0 | (b0 << 16);
    const int abl = __builtin_amdgcn_update_dpp(-1, ab, 0x111, 0xf, 0xf, false), cml = __builtin_amdgcn_update_dpp(0x7fffffff, cmin, 0x111, 0xf, 0xf, false);     // row_shr:1 (old: no neighbour)
    const int abr = __builtin_amdgcn_update_dpp(-1, ab, 0x101, 0xf, 0xf, false), cxr = __builtin_amdgcn_update_dpp(-0x7fffffff, cmax, 0x101, 0xf, 0xf, false);    // row_shl:1
    const bool pmin = !(abl == ab && cml <= cmin), pmax = !(abr == ab && cxr >= cmax);     // (the SAME rows: one compare, and what a smooth flow produces)
#else
    const bool pmin = true, pmax = true;
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j) {          // (a lane touches at least two rows)
        const int r = a0 + j;
        if (r <= b0) {
            if (pmin) atomicMin(&tmin[r], cmin);
            if (pmax) atomicMax(&tmax[r], cmax);
        }
    }
    for (int r = a0 + 2; r <= b0; ++r) {
        if (pmin) atomicMin(&tmin[r], cmin);
        if (pmax) atomicMax(&tmax[r], cmax);
    }
    if (drop) atomicMax(&tmax[kRowTab], 1);
}

template <bool ENT = true, typename WP>
__device__ __forceinline__ void rows_scan(const WP& p, const int* tmin, const int* tmax, uint32_t* ent, uint8_t* start, int org, int cxo, RowGeo& G, RowScan& R) {
    const int l = threadIdx.x & 63;
    const int mn = tmin[l], mx = tmax[l], ov = tmax[kRowTab];
    const int rowy = org + l, cmaxw = (p.w - 1) >> 2;
    const int c0 = max(mn, 0), c1 = min(mx, cmaxw);
    const bool any = mx >= mn, inimg = (uint32_t)rowy < (uint32_t)p.h;
    const bool valid = any && c1 >= c0 && inimg;
    const int cw = valid ? c1 - c0 + 1 : 0;
    // two sums in one scan: chunks (low half: chunk i -> thread i) and chunks of LDS the rows occupy (high half; the same unless
    // OFL_ROWS_PAD pads a row to the rectangle's pitch rule, rows_pitch)
    const int cwc = min(cw, 1000);                                   // (a row that long is not staged; clamped, the 64 sums fit 16 bits)
    const uint32_t sc = (uint32_t)wave_incl_scan(cwc | (rows_pitch(cwc) << 16));
    const int incl = (int)(sc & 0xffffu), pincl = (int)(sc >> 16);
    const int d = c0 - cxo;
    const bool fit = incl <= kRowChunks && pincl <= kRowChunks;      // (both sums are monotonic: the rows that fit are a prefix)
    const bool stg = valid && fit && (uint32_t)(d + 128) <= 256u;
    const int nst = __popcll(__ballot(fit));
    G.tot = nst ? __builtin_amdgcn_readlane(incl, nst - 1) : 0;
    const uint32_t e = stg ? ((uint32_t)(16 * (1 + 4 * (pincl - rows_pitch(cwc)) - d + 256)) | ((uint32_t)(16 * cw) << 16)) : ((any && !valid) ? 1u : 0u);
    const bool outside = any && (mn < 0 || mx > (p.w >> 2) - 1 || !inimg);      // a tap outside the frame (or in its last, partial chunk)
    G.interior = (ov <= 0) && (__ballot((valid && !stg) || outside) == 0ull) && G.tot > 0;
    if (ENT && threadIdx.x < 64) ent[l] = e;
    // chunk -> row for rows_map: every wave marks the rows' first chunks itself (identical values from all four waves, and a wave
    // reads its own writes in order: no barrier between this and rows_issue)
    if (ENT && cw > 0 && fit) start[incl - cw] = (uint8_t)(l + 1);
    R.incl = incl; R.sc = (int)sc; R.a = c0 | (cw << 16);
    G.org = org; G.cxo = cxo;
}

// chunk i0 + lane of the packed rows -> global offset of its 4 pixels and its LDS slot | row length << 16 (-1: past the last chunk).
// The row of a chunk = the last row that starts at or before it: the start marks of the wave's 64 chunks, a prefix maximum over the
// lanes (DPP), and the row that holds the wave's first chunk (a ballot) as the carry.  (First cut: a uniform loop over the following
// rows' sums with v_readlane -- 5 000 to 8 000 ticks per tile in the phase stamps, a third of the kernel.)
__device__ __forceinline__ int wave_incl_max(int v) {               // (values >= 0)
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return v;
}
template <typename WP>
__device__ __forceinline__ void rows_map(const WP& p, const RowGeo& G, const RowScan& R, const uint8_t* start, int i0, uint32_t& g, int& slot) {
    const int i = i0 + (int)(threadIdx.x & 63);
    const int carry = __popcll(__ballot(R.incl <= i0)) + 1;          // 1 + the row that holds the wave's first chunk
    const int row = max(wave_incl_max((int)start[min(i, kRowChunks - 1)]), carry) - 1;
    const int rowc = min(row, kRowTab - 1);
    const int ra = __builtin_amdgcn_ds_bpermute(rowc << 2, R.a), rs = __builtin_amdgcn_ds_bpermute(rowc << 2, R.sc);
    const int cwr = ra >> 16, c0 = ra & 0xffff, cl = i - ((rs & 0xffff) - cwr), pb = (int)((uint32_t)rs >> 16) - rows_pitch(cwr);
    const bool on = i < G.tot;
    g = on ? (uint32_t)(__mul24(G.org + rowc, p.w) + ((c0 + cl) << 2)) : 0u;
    slot = on ? ((1 + 4 * pb + cl) | (cwr << 16)) : -1;
}

// the first kRowIters * kLdsNT chunks are staged through registers, in flight while the previous tile is gathered ...
constexpr int kRowIters = 2;
static_assert(kRowChunks <= (kRowIters + 1) * kLdsNT, "rows_extra stages exactly ONE round beyond the kRowIters rounds that go through registers");
template <int NC> struct RowStage { int slot[kRowIters]; f4 q[kRowIters][NC]; uint32_t mq[kRowIters]; };
template <int NC, bool VALID, bool SUB = false, typename TS = float, typename WP>
__device__ __forceinline__ void rows_issue(const WP& p, const TS* __restrict__ sb, const uint8_t* __restrict__ sm, uint32_t hw,
                                           const RowGeo& G, const RowScan& R, const uint8_t* start, RowStage<NC>& S, const float* __restrict__ sbb = nullptr) {
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#pragma unroll
    for (int it = 0; it < kRowIters; ++it) {
        S.slot[it] = -1;
        const int i0 = it * kLdsNT + wv * 64;
        if (it < OFL_WARP_UNCOND_ROUNDS || i0 < G.tot) {       // (the first round whatever the tile: countable loads, see lds_issue)
            uint32_t g;
            rows_map(p, G, R, start, i0, g, S.slot[it]);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                S.q[it][c] = ld4(sb + c * hw + g);
                if (SUB) S.q[it][c] = S.q[it][c] - ld4(sbb + c * hw + g);          // (mode 1 't': the warped field is flow - self)
            }
            if (VALID) S.mq[it] = ld32(sm ? sm + g : reinterpret_cast<const uint8_t*>(sb) + g);
            else S.mq[it] = 0x01010101u;
        }
    }
}

template <int NC, bool VALID>
__device__ __forceinline__ void rows_put(f4* lds, int slot, const f4 (&q)[NC], uint32_t mq) {
    const int sl0 = slot & 0xffff, cwr = slot >> 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f4 sl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) sl[c] = q[c][k];
        if (VALID) sl[3] = fminf((float)((mq >> (8 * k)) & 0xffu), 1.0f);   // non-zero byte -> 1
        lds[sl0 + k * cwr] = sl;
    }
}

template <int NC, bool VALID>
__device__ __forceinline__ void rows_write(f4* lds, const RowStage<NC>& S, bool has_sm) {
    if (threadIdx.x == 0) lds[0] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < kRowIters; ++it) {
        if (S.slot[it] >= 0) rows_put<NC, VALID>(lds, S.slot[it], S.q[it], (VALID && !has_sm) ? 0x01010101u : S.mq[it]);
    }
}

// ... and the chunks beyond them (a tile with more than 2 staged pixels per output pixel: 1 % of the tiles at sigma 8, 9 % at 12, 23 % at 16)
// in a round of their own, loaded and written on the spot by the tile's own iteration: its registers are not live across a gather
// (three rounds through registers put the kernel over the 168-register limit, and a scratch reload waits for every load in flight).
// The table is still there (reset only after the gather): the scan is simply redone.
template <int NC, bool VALID, bool SUB = false, typename TS = float, typename WP>
__device__ __forceinline__ void rows_extra(const WP& p, const TS* __restrict__ sb, const uint8_t* __restrict__ sm, uint32_t hw,
                                           const int* tmin, const int* tmax, const uint8_t* start, const RowGeo& G, f4* lds, const float* __restrict__ sbb = nullptr) {
    if (__builtin_expect(G.tot > kRowIters * kLdsNT, 0)) {
        RowGeo G2; RowScan R;
        rows_scan<false>(p, tmin, tmax, (uint32_t*)nullptr, (uint8_t*)nullptr, G.org, G.cxo, G2, R);     // (the start marks of this tile are still there)
        const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        uint32_t g; int slot;
        rows_map(p, G, R, start, kRowIters * kLdsNT + wv * 64, g, slot);
        f4 q[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            q[c] = ld4(sb + c * hw + g);
            if (SUB) q[c] = q[c] - ld4(sbb + c * hw + g);
        }
        const uint32_t mq = (VALID && sm) ? ld32(sm + g) : 0x01010101u;
        // (no branch round the use of these loads: on a path that skipped their wait they would count as in flight at the join, and the
        // compiler's next waits -- vmcnt(2), (1), (0) -- then wait for the NEXT tile's flow instead: a lane without a chunk writes a spare slot)
        rows_put<NC, VALID>(lds, slot >= 0 ? slot : 1 + 4 * kRowChunks, q, mq);
    }
}

#if OFL_ROWS_STAMPS
__device__ unsigned long long g_rows_stamp[16];
#define OFL_RS(i_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_[i_] += t_ - last_; last_ = t_; } while (0)
#else
#define OFL_RS(i_)
#endif
// ADD: the fused composition of mode 3 (flow_class.py:1804-1808): out = a_sign * flow + g_sign * warped, the addend being the flow
// operand itself (add_is_flow) -- its registers are kept instead of the tile's positions, which are formed again at gather time.
// (ADD 2: another addend -- the outer `flow - (...)` of modes 1-2, Flow.combine's cells; SUB: the staged field is src - src_b (mode 1 't');
// DF: the flag word of the OUTPUT read as a flow under `valid`, as a by-product.)
// (TS / TD / ROUND: uint8 images warped from and to their bytes -- ofl_warp_bwd_u8 -- read the rounding mode at run time; the width is a
// multiple of 4 in every launch of this kernel.)
// (GRAD: the gradient with respect to the FLOW -- ofl_warp_bwd_grad_f32 -- see lds_gather_impl: `addend` is the upstream gradient, dst two planes.)
template <int T, int NC, bool VALID, int ADD = 0, bool SUB = false, bool DF = false, typename TS = float, typename TD = float, bool ROUND = false, bool GRAD = false>
__global__ __launch_bounds__(kLdsNT, 3) void warp_bwd_rows_kernel(const WarpParams p_by_value) {
    static_assert((ADD == 0 && !SUB && !DF) || NC == 2, "ADD / SUB / DF: flows");
    static_assert(!GRAD || (!VALID && ADD == 0 && !SUB && !DF), "GRAD: plain taps");
    typedef typename std::conditional<ROUND, WarpParamsK, WarpParamsLeanK>::type WPK;
    WPK* pp = (WPK*)__builtin_amdgcn_kernarg_segment_ptr();
#define p (*pp)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ RowTabs rt;
    int tx, tyg, n;
    if (!decode_tile(p, tx, tyg, n)) return;
    const int tid = threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    const int w = p.w, h = p.h;
    const uint32_t hw = (uint32_t)(h * w);
    const float* __restrict__ fu = p.flow + n * p.flow_bs;
    const TS* __restrict__ sb = reinterpret_cast<const TS*>(p.src) + n * p.src_bs;      // (fp16 sources: p.src points at halves)
    const float* __restrict__ sbb = SUB ? p.src_b + n * p.src_b_bs : nullptr;
    const uint8_t* __restrict__ sm = p.src_mask ? p.src_mask + n * p.src_mask_bs : nullptr;
    const uint8_t* __restrict__ fm = p.flow_mask ? p.flow_mask + n * p.flow_mask_bs : nullptr;
    const int xq = min(tx * (kLdsTWQ * 4) + lx * 4, w - 4);
    int dflags = 0;
    f4 uu[T], vv[T];
    uint32_t fmk[T];
    auto load_flow = [&](int k) {
        const uint32_t pix = (uint32_t)(min((tyg * T + k) * kLdsTH + ly, h - 1) * w + xq);
        uu[k] = ld4nt(fu + pix); vv[k] = ld4nt(fu + hw + pix);
        fmk[k] = 0x01010101u;
        if (VALID) fmk[k] = ld32(fm ? fm + pix : reinterpret_cast<const uint8_t*>(fu) + pix);
    };
    // the flow-mask word of a tile is loaded with its flow (two tiles ahead: a countable load) but used last, by its stores: it waits
    // in the LDS, not in a register -- at the register limit (168) the allocator spilled exactly these words, and a scratch reload is a
    // VMEM operation whose wait (vmcnt(0)) also waits for the next tile's staging loads
    __shared__ uint32_t park[2][kLdsNT];
    // three blocks per CU need <= 53 760 bytes of LDS per block (measured: 53 296 -> three blocks, 53 816 -> two, and every kernel
    // of this family 10-25 % slower: 520 bytes more in RowTabs did that in an experiment)
    static_assert(kLdsNT != 256 || kRowsLdsBytes + (int)sizeof(RowTabs) + (int)sizeof(park) <= 53760, "LDS budget of three blocks per CU");
    auto fm_park = [&](int k) { if (VALID) park[k & 1][tid] = fmk[k]; };
    auto fmw = [&](int k) -> uint32_t { return (VALID && !fm) ? 0x01010101u : (VALID ? park[k & 1][tid] : 0x01010101u); };
    auto reset = [&](int t) { if (tid <= kRowTab) { rt.tmin[t][tid] = 0x7fffffff; rt.tmax[t][tid] = -0x7fffffff; } };
    auto clear_starts = [&](int t) { if (tid < kRowChunks / 4) reinterpret_cast<uint32_t*>(rt.start[t])[tid] = 0u; };     // (before the barrier that precedes the tile's scan; double-buffered: rows_extra still reads the other one)
#if OFL_ROWS_STAMPS
    unsigned long long acc_[12] = {}, last_ = __builtin_amdgcn_s_memtime();
#endif
    const float osample = rows_origins_load<T>(p, fu, hw, tx, tyg);
    const float osample2 = T > 4 ? rows_origins_load<T, 4>(p, fu, hw, tx, tyg) : 0.0f;
    load_flow(0);
    if (T > 1) load_flow(1);
    reset(0); reset(1);                   // tile k posts to table k & 1; it is reset for tile k + 2 while tile k is gathered
    clear_starts(0);
    int org[T], cxo[T];
    rows_origins<T>(p, osample, org, cxo);
    if (T > 4) rows_origins<T, 4>(p, osample2, org, cxo);
    OFL_RS(10);
    f4* lds = reinterpret_cast<f4*>(smem);
    // mode 3: a tile's positions are KEPT from the posts to the gather (-6 % against re-forming them from the flow registers) -- except with
    // the flag by-product, where the kept positions spill (32 B of scratch)
    constexpr bool REFORM = ADD == 1 && (OFL_ROWS_ADD_REFORM || DF);
    LdsCoords Tc[REFORM ? 1 : T];
    RowGeo Gx[T];
    RowStage<NC> S;
    RowScan R;
    lds_barrier();
    OFL_RS(11);
    lds_coords_box_a<false>(p, tx, tyg * T, uu[0], vv[0], 0, Tc[0], (int (*)[4])nullptr);
    OFL_RS(1);
    rows_post(p, Tc[0], org[0], rt.tmin[0], rt.tmax[0]);
    fm_park(0);
    OFL_RS(2);
    lds_barrier();
    OFL_RS(4);
    rows_scan(p, rt.tmin[0], rt.tmax[0], rt.ent[0], rt.start[0], org[0], cxo[0], Gx[0], R);
    OFL_RS(5);
    rows_issue<NC, VALID, SUB, TS>(p, sb, sm, hw, Gx[0], R, rt.start[0], S, sbb);
    OFL_RS(6);
#pragma unroll
    for (int k = 0; k < T; ++k) {
        OFL_OPAQUE_S(pp);
        const int tyk = tyg * T + k;       // (a tile past the bottom edge recomputes and re-stores the frame's last row: OFL_WARP_ALWAYS_T)
        if (k + 1 < T) {
            lds_coords_box_a<false>(p, tx, tyk + 1, uu[k + 1], vv[k + 1], 0, Tc[REFORM ? 0 : k + 1], (int (*)[4])nullptr);
            OFL_RS(1);
            rows_post(p, Tc[REFORM ? 0 : k + 1], org[k + 1], rt.tmin[(k + 1) & 1], rt.tmax[(k + 1) & 1]);
            fm_park(k + 1);
            OFL_RS(2);
        }
        if (k + 2 < T) load_flow(k + 2);
        rows_write<NC, VALID>(lds, S, sm != nullptr);
        rows_extra<NC, VALID, SUB, TS>(p, sb, sm, hw, rt.tmin[k & 1], rt.tmax[k & 1], rt.start[k & 1], Gx[k], lds, sbb);
        if (k + 1 < T) clear_starts((k + 1) & 1);
        OFL_RS(3);
        lds_barrier();
        OFL_RS(4);
        f4 ad[NC] = {};
        if (k + 1 < T) {
            rows_scan(p, rt.tmin[(k + 1) & 1], rt.tmax[(k + 1) & 1], rt.ent[(k + 1) & 1], rt.start[(k + 1) & 1], org[k + 1], cxo[k + 1], Gx[k + 1], R);
            OFL_RS(5);
            if (ADD == 2 || GRAD) lds_load_addend<NC>(p, tx, tyk, n, hw, ad);   // (ahead of the younger staging loads: waited for without them)
            rows_issue<NC, VALID, SUB, TS>(p, sb, sm, hw, Gx[k + 1], R, rt.start[(k + 1) & 1], S, sbb);
            OFL_RS(6);
        }
        LdsBox B;
        B.fits = true; B.clipped = false; B.interior = Gx[k].interior; B.ent = rt.ent[k & 1]; B.org = Gx[k].org; B.cxo = Gx[k].cxo;
        f4 outv[4];
        if (REFORM) lds_coords_box_a<false>(p, tx, tyk, uu[k], vv[k], 0, Tc[0], (int (*)[4])nullptr);
        if (ADD == 1) { ad[0] = uu[k]; ad[NC - 1] = vv[k]; }
        if ((ADD == 2 || GRAD) && k + 1 >= T) lds_load_addend<NC>(p, tx, tyk, n, hw, ad);
        lds_gather<NC, VALID, SUB, TS, GRAD, false, typename std::remove_reference<decltype(p)>::type, true>(p, hw, sb, sm, Tc[REFORM ? 0 : k], B, smem, outv, sbb, GRAD ? ad : nullptr);
        OFL_RS(7);
        if (GRAD) { const f4 none[2] = {}; lds_store<2, false, false, false, float>(p, tx, tyk, n, hw, 0u, outv, none); }
        else lds_store<NC, VALID, ADD != 0, DF, TD>(p, tx, tyk, n, hw, fmw(k), outv, ad, &dflags);
        OFL_RS(8);
        if (k + 1 >= T) break;
        if (T > 4 && (tyk + 1) * kLdsTH >= h) break;          // (taller columns only: nothing but duplicates of the frame's last row follows)
        if (k + 2 < T) reset(k & 1);       // (its last reader was this tile's rows_extra, before the barrier above)
        lds_barrier();
        OFL_RS(9);
    }
    if (DF) { dflags = wave_or_flags(dflags); if ((tid & 63) == 0) flag_or(&p.dst_flags[n], dflags); }
#if OFL_ROWS_STAMPS
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) atomicAdd(&g_rows_stamp[i], acc_[i]);
        atomicAdd(&g_rows_stamp[15], 1ull);
    }
#endif
#undef p
}

// MANY CHANNELS (C >= 4; Flow.apply of an N-C-H-W feature tensor, utils.py:469-555): ONE launch for all of them.  A block owns one
// output tile: flow, coordinates, bounding box, the byte addresses of the 16 taps of a lane's 4 pixels in the staged box and their
// 16 weights, and the staging geometry are formed ONCE; then the block walks the channels in groups of 4 -- one 16-byte LDS slot
// per source pixel -- as a pipeline ALONG C: the staging loads of group g + 1 fly while group g is gathered and stored.  Per group a
// lane issues its staging loads, 16 ds_read_b128, 32 packed FMAs and 4 stores -- no coordinate, no address arithmetic.  With the
// valid area wanted (VALID) the first group is channels 0 .. 2 plus the target mask as its fourth plane, and writes the mask.
// (Groups of 3 as separate launches re-read the flow and its mask and redid every coordinate per group: 9 B/px + ~100 VALU per pixel
// of pure repetition, +37 % traffic at C = 64.)  A channel count that is not a multiple of 4 makes the LAST group start at C - 4: it
// recomputes up to three planes (identical values, duplicate stores).  The first pipelined group is peeled so that, at the loop
// head, the staged group is 4 stores old on BOTH incoming paths: the compiler's vmcnt wait for it then leaves the previous group's
// stores in flight (see lds_store).  Same expressions, same FMA chain as lds_gather_impl: bit-identical to the launches of 3.
template <typename WP>
__device__ __forceinline__ void lds_taps(const WP& p, const LdsCoords& T, const LdsBox& B, int (&si)[16], float (&wg)[16]) {
    const int w = p.w, h = p.h;
    const int cw16 = B.cw * 16, P16 = B.Pp * 16;
    const float wf = (float)w, hf = (float)h;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float fx = floorf(T.sx[k]), fy = floorf(T.sy[k]);
        const float ww = T.sx[k] - fx, e = 1.0f - ww, nn = T.sy[k] - fy, s_ = 1.0f - nn;
        wg[4 * k + 0] = s_ * e; wg[4 * k + 1] = s_ * ww; wg[4 * k + 2] = nn * e; wg[4 * k + 3] = nn * ww;
        const int xi = (int)__builtin_amdgcn_fmed3f(fx, -2.0f, wf), yi = (int)__builtin_amdgcn_fmed3f(fy, -2.0f, hf);
        const bool x0 = (uint32_t)xi < (uint32_t)w, x1 = (uint32_t)(xi + 1) < (uint32_t)w;
        const bool y0 = (uint32_t)yi < (uint32_t)h, y1 = (uint32_t)(yi + 1) < (uint32_t)h;
        const int yr = yi - B.miny;
        const int ra = yr - lds_shear(xi >> 2, B.sq), rb = yr - lds_shear((xi + 1) >> 2, B.sq);
        const int xl0 = xi - B.bx0, xl1 = xl0 + 1;
        const int cp0 = __mul24(xl0 & 3, cw16) + ((xl0 & ~3) << 2), cp1 = __mul24(xl1 & 3, cw16) + ((xl1 & ~3) << 2);
        const int r0 = 16 + __mul24(ra, P16), r1 = 16 + __mul24(rb, P16);
        si[4 * k + 0] = (x0 && y0) ? r0 + cp0 : 0; si[4 * k + 1] = (x1 && y0) ? r1 + cp1 : 0;
        si[4 * k + 2] = (x0 && y1) ? r0 + P16 + cp0 : 0; si[4 * k + 3] = (x1 && y1) ? r1 + P16 + cp1 : 0;
    }
}

// OVERSIZE box staged in part (lds_coords_box<.., CLIP>: the rows that fit; none at all when not even OFL_WARP_CLIP rows would): the
// pixels of this lane with a tap below the staged rows (bit k of `below`) take their 4 planes from global memory instead -- flow re-read (an L2 hit), coordinates re-formed, the
// arithmetic of lds_gather_impl -- so that nothing of it is kept in registers across the channel loop.  Cold (a wave with no such
// lane skips it), and a real call for the same reason as above.
struct Out4 { f4 v[4]; };
template <int NCH, bool MASK3>
__device__ __attribute__((noinline)) Out4 chan_pixels_from_global(WarpParamsK* pp, const float* sb, uint32_t pix, uint32_t below, Out4 cur,
                                                                  int tx, int ty, int n, int row) {
#define p (*pp)
    const int w = p.w, h = p.h;
    const uint32_t hw = (uint32_t)(h * w);
    const float* __restrict__ fu = p.flow + n * p.flow_bs;
    const uint8_t* __restrict__ sm = p.src_mask ? p.src_mask + n * p.src_mask_bs : nullptr;
    const f4 uu = ld4nt(fu + pix), vv = ld4nt(fu + hw + pix);
    LdsCoords Tc;
    int dummy[1][4];
    lds_coords_box_a<false>(p, tx, ty, uu, vv, 0, Tc, dummy, row);
    LdsBox Bx = {};
    Bx.fits = false; Bx.interior = false; Bx.clipped = false;
    f4 got[4];
    lds_gather_impl<NCH, MASK3, false>(p, hw, sb, MASK3 ? sm : nullptr, Tc, Bx, nullptr, got);
#pragma unroll
    for (int k = 0; k < 4; ++k) if ((below >> k) & 1u) cur.v[k] = got[k];
    return cur;
#undef p
}

// SUBS (1 or 3): a block of SUBS x kLdsNT threads owns SUBS tiles SIDE BY SIDE, each with its own box in its own part of the LDS, all
// walking the channel groups in LOCKSTEP (the barriers are the block's).  What it buys: the x-halo of neighbouring boxes -- partial 128-byte
// lines that a lone tile fetches for itself, per group, because its neighbours are at other groups at that moment (profiles/r5_chan_pmc.txt)
// -- is requested by waves of ONE CU within the same few hundred cycles and fetched once.  The grid then counts tile TRIPLES along x; a
// sub-tile past the frame's right edge recomputes the last columns (clamped loads, identical duplicate stores).
// ROWS: per-row extents of the staged box (see warp_bwd_rows_kernel) -- table, scan, chunk map and tap addresses are formed ONCE per
// tile like every other invariant, so the channel loop is unchanged and every group stages ~23 % fewer bytes at sigma 8.
template <typename WP>
__device__ __forceinline__ void rows_taps(const WP& p, const LdsCoords& T, const uint32_t* ent, int org, int cxo, int (&si)[16], float (&wg)[16], uint32_t& below) {
    const int w = p.w, h = p.h;
    const float wf = (float)w, hf = (float)h;
    below = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float fx = floorf(T.sx[k]), fy = floorf(T.sy[k]);
        const float ww = T.sx[k] - fx, e = 1.0f - ww, nn = T.sy[k] - fy, s_ = 1.0f - nn;
        wg[4 * k + 0] = s_ * e; wg[4 * k + 1] = s_ * ww; wg[4 * k + 2] = nn * e; wg[4 * k + 3] = nn * ww;
        const int xi = (int)__builtin_amdgcn_fmed3f(fx, -2.0f, wf), yi = (int)__builtin_amdgcn_fmed3f(fy, -2.0f, hf);
        const bool x0 = (uint32_t)xi < (uint32_t)w, x1 = (uint32_t)(xi + 1) < (uint32_t)w;
        const bool y0 = (uint32_t)yi < (uint32_t)h, y1 = (uint32_t)(yi + 1) < (uint32_t)h;
        const bool ok[4] = {x0 && y0, x1 && y0, x0 && y1, x1 && y1};
        const int yr = yi - org;
        const bool inr = (uint32_t)yr < (uint32_t)(kRowTab - 1);
        const int yrc = inr ? yr : 0;
        const uint32_t e0 = ent[yrc], e1 = ent[yrc + 1];
        const bool staged = (inr && e0 != 0u && e1 != 0u) || !(ok[0] || ok[1] || ok[2] || ok[3]);     // (lds_gather_impl<.., ROWS>'s test)
        const uint32_t m0 = (uint32_t)xi & 3u, m1 = (uint32_t)(xi + 1) & 3u;
        const int q0 = (((xi >> 2) - cxo) << 4) - 4096, q1 = ((((xi + 1) >> 2) - cxo) << 4) - 4096;
        si[4 * k + 0] = (staged && ok[0]) ? (int)(e0 & 0xffffu) + (int)__umul24(m0, e0 >> 16) + q0 : 0;
        si[4 * k + 1] = (staged && ok[1]) ? (int)(e0 & 0xffffu) + (int)__umul24(m1, e0 >> 16) + q1 : 0;
        si[4 * k + 2] = (staged && ok[2]) ? (int)(e1 & 0xffffu) + (int)__umul24(m0, e1 >> 16) + q0 : 0;
        si[4 * k + 3] = (staged && ok[3]) ? (int)(e1 & 0xffffu) + (int)__umul24(m1, e1 >> 16) + q1 : 0;
        if (!staged) below |= 1u << k;                          // (its taps come from global memory: chan_pixels_from_global)
    }
}

template <bool VALID, bool LEAN = false, int SUBS = 1, bool ROWS = false>
__global__ __launch_bounds__(kLdsNT * SUBS, 3) void warp_bwd_lds_chan_kernel(const WarpParams p_by_value) {
    static_assert(!ROWS || (LEAN && SUBS == 1), "ROWS: lean launches, one tile per block");
    typedef typename std::conditional<LEAN, WarpParamsLeanK, WarpParamsK>::type WPK;       // (LEAN: see WarpParamsLean)
    WPK* pp = (WPK*)__builtin_amdgcn_kernarg_segment_ptr();
#define p (*pp)
    constexpr int NW = kLdsNT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    __shared__ int red_all[SUBS][NW][4];
    static_assert(!ROWS || kLdsNT != 256 || kRowsLdsBytes + (int)sizeof(RowTabs) + (int)sizeof(red_all) <= 53760, "LDS budget of three blocks per CU (ROWS instantiation)");
    int tx, ty, n;
    if (!decode_tile(p, tx, ty, n)) return;
    const int sub = SUBS > 1 ? (int)threadIdx.x / kLdsNT : 0;
    const int tid = SUBS > 1 ? (int)threadIdx.x - sub * kLdsNT : (int)threadIdx.x, lx = tid % kLdsTWQ, ly = tid / kLdsTWQ;
    unsigned char* smem = smem_all + sub * kLdsBytes;
    int (*red)[4] = red_all[sub];
    tx = tx * SUBS + sub;
    const int w = p.w, h = p.h, C = p.c;
    const uint32_t hw = (uint32_t)(h * w);
    const float* __restrict__ fu = p.flow + n * p.flow_bs;
    const float* __restrict__ sb0 = p.src + n * p.src_bs;
    const uint8_t* __restrict__ sm = p.src_mask ? p.src_mask + n * p.src_mask_bs : nullptr;
    const uint8_t* __restrict__ fm = p.flow_mask ? p.flow_mask + n * p.flow_mask_bs : nullptr;
    const int x4 = tx * (kLdsTWQ * 4) + lx * 4, xq = min(x4, w - 4);
    uint32_t pix = (uint32_t)(min(ty * kLdsTH + ly, h - 1) * w + xq);
    const float osample = ROWS ? rows_origins_load<1>(p, fu, hw, tx, ty) : 0.0f;
    const f4 uu = ld4nt(fu + pix), vv = ld4nt(fu + hw + pix);
    uint32_t fmk = 0x01010101u;
    if ((VALID || OFL_WP_FLOW_FLAGS(p)) && fm) fmk = ld32(fm + pix);
    if (OFL_WP_FLOW_FLAGS(p)) {                      // finiteness / zero tests of the flow operand as a by-product (wave-uniform branch)
        int f = 0;
        if ((x4 < w) && (ty * kLdsTH + ly < h)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) f |= flag_bits(uu[q], vv[q], ((fmk >> (8 * q)) & 0xffu) != 0u);
        }
        f = wave_or_flags(f);
        if ((tid & 63) == 0) flag_or(&p.flow_flags[n], f);
    }
    f4* lds = reinterpret_cast<f4*>(smem);
    LdsCoords Tc;
    LdsBox Bx;
    int si[16];
    float wg[16];
    uint32_t below = 0u;                                     // pixels of this lane with a tap below the staged rows of a clipped box
    uint32_t goff[kLdsIters];                                // (widths that are multiples of 4 only: no chunk straddles a row end)
    int slot[kLdsIters];                                     // byte address of the chunk's first slot | 16 * its row's length << 16 (-1: nothing to stage)
    bool any_below;
    int rounds;
    if (ROWS) {
        __shared__ RowTabs rt;
        if (tid <= kRowTab) { rt.tmin[0][tid] = 0x7fffffff; rt.tmax[0][tid] = -0x7fffffff; }
        if (tid < kRowChunks / 4) reinterpret_cast<uint32_t*>(rt.start[0])[tid] = 0u;
        int org[1], cxo[1];
        rows_origins<1>(p, osample, org, cxo);
        lds_barrier();
        lds_coords_box_a<false>(p, tx, ty, uu, vv, 0, Tc, (int (*)[4])nullptr);
        rows_post(p, Tc, org[0], rt.tmin[0], rt.tmax[0]);
        lds_barrier();
        RowGeo G; RowScan R;
        rows_scan(p, rt.tmin[0], rt.tmax[0], rt.ent[0], rt.start[0], org[0], cxo[0], G, R);
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int it = 0; it < kLdsIters; ++it) {
            int sw;
            rows_map(p, G, R, rt.start[0], it * kLdsNT + wv * 64, goff[it], sw);
            slot[it] = sw >= 0 ? (16 * (sw & 0xffff)) | ((16 * (sw >> 16)) << 16) : -1;
        }
        rounds = (G.tot + kLdsNT - 1) / kLdsNT;
        lds_barrier();                                       // (the entries are wave 0's)
        rows_taps(p, Tc, rt.ent[0], org[0], cxo[0], si, wg, below);
        any_below = __any(below != 0u);
    } else {
    const int sq = OFL_WP_SHEAR(p) ? lds_slope_row(p, fu, hw, tx, ty * kLdsTH + kLdsTH / 2) : 0;
    lds_coords_box<true, true>(p, tx, ty, uu, vv, sq, Tc, Bx, red, ly);
    // a box of which not even OFL_WARP_CLIP rows fit (an extreme stretch; or an empty one: every tap outside the frame) stages NOTHING and
    // every pixel takes the global fix-up below -- cold, but the tile keeps walking the groups with its block (barriers are the block's)
    const bool nofit = !Bx.fits;                             // (uniform over the sub-tile)
    if (__builtin_expect(nofit, 0)) { Bx.nch = 0; Bx.bh = 0; Bx.clipped = true; }
    // --- per-tile invariants: taps, weights, staging geometry -------------------------------------------------------------
    lds_taps(p, Tc, Bx, si, wg);
    if (__builtin_expect(nofit, 0)) {
        below = 0xfu;
#pragma unroll
        for (int i = 0; i < 16; ++i) si[i] = 0;
    } else if (__builtin_expect(Bx.clipped, 0)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {                        // (lds_gather_impl's test, on the row of the lower taps)
            const int xi = (int)__builtin_amdgcn_fmed3f(floorf(Tc.sx[k]), -2.0f, (float)w), yi = (int)__builtin_amdgcn_fmed3f(floorf(Tc.sy[k]), -2.0f, (float)h);
            const int yr = yi - Bx.miny;
            const int ra = yr - lds_shear(xi >> 2, Bx.sq), rb = yr - lds_shear((xi + 1) >> 2, Bx.sq);
            if (!(max(ra, rb) + 1 < Bx.bh)) {
                below |= 1u << k;
#pragma unroll
                for (int j = 0; j < 4; ++j) si[4 * k + j] = 0;   // (its LDS reads go to the zero slot; the result is replaced)
            }
        }
    }
    any_below = Bx.clipped && __any(below != 0u);  // wave-uniform
    rounds = (Bx.nch + kLdsNT - 1) / kLdsNT;       // 0 (nothing fits) .. kLdsIters, uniform over the sub-tile
    {
        const uint32_t inv = inv20((uint32_t)Bx.cw);
#pragma unroll
        for (int it = 0; it < kLdsIters; ++it) {             // (lds_issue's arithmetic)
            const uint32_t i = (uint32_t)tid + it * kLdsNT;
            const uint32_t r = __umul24(i, inv) >> 20, c4 = i - __umul24(r, (uint32_t)Bx.cw);
            const int y = Bx.miny + (int)r + lds_shear(Bx.cbase + (int)c4, Bx.sq);
            const bool on = (i < (uint32_t)Bx.nch) && ((uint32_t)y < (uint32_t)h) && it < rounds;
            goff[it] = on ? (uint32_t)(__mul24(y, w) + Bx.bx0) + c4 * 4u : 0u;
            slot[it] = on ? (16 * (1 + (int)(__umul24(r, (uint32_t)Bx.Pp) + c4))) | ((Bx.cw * 16) << 16) : -1;
        }
    }
    }
    // nothing above is to be recomputed, and nothing else hoisted, inside the channel loop
    static_assert(kLdsBytes <= 65536, "tap addresses are packed as 16-bit LDS byte addresses");
    uint32_t sp[8];                                          // two 16-bit LDS byte addresses per register
#pragma unroll
    for (int i = 0; i < 8; ++i) sp[i] = (uint32_t)si[2 * i] | ((uint32_t)si[2 * i + 1] << 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(sp[i]));
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(wg[i]));
#pragma unroll
    for (int it = 0; it < kLdsIters; ++it) { asm volatile("" : "+v"(goff[it])); asm volatile("" : "+v"(slot[it])); }
    asm volatile("" : "+v"(pix));

    f4 q[kLdsIters][4];                                      // the staged group: [round][plane] = 4 pixels of one plane
    // staging loads of the 4 planes at `sb` (mask3: the fourth plane is the target mask's bytes, as 0 / 1 floats)
    auto issue = [&](const float* __restrict__ sb, bool mask3) {
#pragma unroll
        for (int it = 0; it < kLdsIters; ++it) {
            if (it == 0 || it < rounds) {
#pragma unroll
                for (int c = 0; c < 3; ++c) q[it][c] = ld4(sb + c * hw + goff[it]);
                if (VALID && mask3) {
                    const uint32_t mb = sm ? ld32(sm + goff[it]) : 0x01010101u;
                    q[it][3] = (f4){fminf((float)(mb & 0xffu), 1.0f), fminf((float)((mb >> 8) & 0xffu), 1.0f),
                                    fminf((float)((mb >> 16) & 0xffu), 1.0f), fminf((float)(mb >> 24), 1.0f)};
                } else q[it][3] = ld4(sb + 3 * hw + goff[it]);
            }
        }
    };
    auto publish = [&]() {                                   // registers -> interleaved LDS slots (lds_write's layout)
        if (tid == 0) lds[0] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < kLdsIters; ++it) {
            if (slot[it] >= 0) {
                const int sl0 = slot[it] & 0xffff, cw16 = slot[it] >> 16;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<f4*>(smem + sl0 + k * cw16) = (f4){q[it][0][k], q[it][1][k], q[it][2][k], q[it][3][k]};
            }
        }
        lds_barrier();
    };
    auto gather = [&](f4 (&outv)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f4 tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) tv[j] = *reinterpret_cast<const f4*>(smem + ((j & 1) ? (sp[2 * k + (j >> 1)] >> 16) : (sp[2 * k + (j >> 1)] & 0xffffu)));
            f4 r = tv[0] * wg[4 * k];
            r = __builtin_elementwise_fma(tv[1], (f4){wg[4 * k + 1], wg[4 * k + 1], wg[4 * k + 1], wg[4 * k + 1]}, r);
            r = __builtin_elementwise_fma(tv[2], (f4){wg[4 * k + 2], wg[4 * k + 2], wg[4 * k + 2], wg[4 * k + 2]}, r);
            r = __builtin_elementwise_fma(tv[3], (f4){wg[4 * k + 3], wg[4 * k + 3], wg[4 * k + 3], wg[4 * k + 3]}, r);
            outv[k] = r;
        }
    };
    float* __restrict__ db = p.dst + (int64_t)n * p.dst_bs;
    auto store = [&](const f4 (&outv)[4], int first, int planes) {   // planes `first` .. of dst (unconditional: see lds_store)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < planes) {
                f4 o = {outv[0][c], outv[1][c], outv[2][c], outv[3][c]};
                if (OFL_WP_ROUND(p) != OFL_ROUND_NONE) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = apply_round(o[k], OFL_WP_ROUND(p));
                }
                st4o(db + (int64_t)(first + c) * hw + pix, o);
            }
        }
    };
    int g = 0;                               // first channel the groups of 4 have still to do
    if (VALID) {
        issue(sb0, true);
        publish();
        f4 outv[4];
        gather(outv);
        if (__builtin_expect(any_below, 0)) {
            if (below != 0u) {
                const Out4 fixed = chan_pixels_from_global<3, true>(reinterpret_cast<WarpParamsK*>(pp), sb0, pix, below, Out4{{outv[0], outv[1], outv[2], outv[3]}}, tx, ty, n, ly);
#pragma unroll
                for (int k = 0; k < 4; ++k) outv[k] = fixed.v[k];
            }
        }
        uint32_t vo = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) vo |= (uint32_t)((outv[k][3] > kValidThr) && (((fmk >> (8 * k)) & 0xffu) != 0u)) << (8 * k);
        st32o(p.valid + (int64_t)n * hw + pix, vo);
        store(outv, 0, 3);
        lds_barrier();
        g = 3;
    }
    const int groups = (C - g + 3) >> 2;     // >= 1 (the host sends C >= 4)
    int cur = min(g, C - 4);
    issue(sb0 + (int64_t)cur * hw, false);
    auto group = [&](bool next) {            // publish the staged group, start the next one's loads, gather, store
        publish();
        const int mine = cur;
        if (next) {
            g += 4;
            cur = min(g, C - 4);
            issue(sb0 + (int64_t)cur * hw, false);
        }
        f4 outv[4];
        gather(outv);
        if (__builtin_expect(any_below, 0)) {
            if (below != 0u) {
                const Out4 fixed = chan_pixels_from_global<4, false>(reinterpret_cast<WarpParamsK*>(pp), sb0 + (int64_t)mine * hw, pix, below, Out4{{outv[0], outv[1], outv[2], outv[3]}}, tx, ty, n, ly);
#pragma unroll
                for (int k = 0; k < 4; ++k) outv[k] = fixed.v[k];
            }
        }
        store(outv, mine, 4);
        if (next) lds_barrier();
    };
    if (groups > 1) {
        group(true);                                           // (peeled: see above)
        for (int k = 1; k + 1 < groups; ++k) group(true);
    }
    group(false);
#undef p
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
        float u, v;
        bool fmv = true;
        if (p.fw != 0) {                                             // padded apply: the flow covers a window of the frame
            const int fx = x - p.fox, fy = y - p.foy;
            const bool inside = (uint32_t)fx < (uint32_t)p.fw && (uint32_t)fy < (uint32_t)p.fh;
            const int64_t fpix = inside ? (int64_t)fy * p.fw + fx : 0;
            u = inside ? fu[fpix] : 0.0f;
            v = inside ? fu[(int64_t)p.fh * p.fw + fpix] : 0.0f;
            if (VALID) fmv = inside && (fm ? (fm[fpix] != 0) : true);
        } else {
            u = fu[pix]; v = fv[pix];
            if (VALID || FLAGS) fmv = fm ? (fm[pix] != 0) : true;
        }
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
    // flow WINDOW (Flow.apply(padding=...), flow_class.py:901-913): fw != 0 -> the flow-geometry operands (flow, weight_mask,
    // chan_mask_b) are fh x fw frames that cover rows foy .., columns fox .. of the h x w frame of the data; outside the
    // window the flow is its replicated border value (F.pad(mode='replicate')) and both masks are False
    int32_t fh, fw, foy, fox;
    const int32_t* run_if_set;   // optional device flags int32[n]: the atomics path runs for image i only when run_if_set[i] != 0
    const int32_t* any_set;      // (with run_if_set) one word: some image of the pass is flagged
    int32_t fb_round, fb_slots;  // (with run_if_set) the accumulator holds fb_slots images: this launch serves the flagged images ranked [fb_round * fb_slots, (fb_round + 1) * fb_slots) among the flagged ones, image of rank r in slot r % fb_slots
    int32_t* dst_flags;          // optional int32[N] (2-channel data only): flag word of the OUTPUT read as a flow under `valid`
    int32_t raw;                 // 1: the weighted sums themselves, not divided by the density (ofl_splat_sum_f32: the transpose of the backward warp)
    static constexpr bool kLean = false;
};
// The COMMON CASE as a type: the same bytes read as SplatParamsLean promise the gather path that the call has a flow (not positions),
// no flow window, a width that is a multiple of 4 and no rounding -- `SP::kLean` folds those run-time switches (and the
// scalar loads, compares and branches they cost per use) out of the lean instantiations of the bin and gather kernels.  The host
// picks them when the promises hold (splat_tiled_impl): apply 's' -6.5 %, switch_ref -8.7 % (profiles/r5_splat_lean.txt).
struct SplatParamsLean : SplatParams { static constexpr bool kLean = true; };
// the promises of SplatParamsLean hold for this call (the lean instantiations exist for 2 and 3 channels)
inline bool splat_is_lean(const SplatParams& s) {
    return s.flow != nullptr && s.fw == 0 && (s.w & 3) == 0 && s.round_mode == OFL_ROUND_NONE;
}
// the switches, as the kernels' helpers read them (s: a SplatParams or SplatParamsLean, in any address space)
#define OFL_SP_WINDOW(s_) (!std::remove_reference<decltype(s_)>::type::kLean && (s_).fw != 0)
#define OFL_SP_WREM(s_) (std::remove_reference<decltype(s_)>::type::kLean ? 0 : ((s_).w & 3))
#define OFL_SP_HAS_FLOW(s_) (std::remove_reference<decltype(s_)>::type::kLean || (s_).flow != nullptr)
#define OFL_SP_RAW(s_) ((s_).raw != 0)          /* (run-time in the lean kernels too: the warp's backward pass -- ofl_splat_sum_f32 -- is a lean call) */
#define OFL_SP_ROUND(s_) (std::remove_reference<decltype(s_)>::type::kLean ? (int32_t)OFL_ROUND_NONE : (s_).round_mode)

// flow window (padded apply): offset of frame pixel (x, y) in a flow-geometry plane (clamped: replicate) and whether it is inside
template <typename SP>
__device__ __forceinline__ uint32_t sp_win(const SP& s, int x, int y, bool& inside) {
    const int fx = x - s.fox, fy = y - s.foy;
    inside = (uint32_t)fx < (uint32_t)s.fw && (uint32_t)fy < (uint32_t)s.fh;
    return (uint32_t)(min(max(fy, 0), s.fh - 1) * s.fw + min(max(fx, 0), s.fw - 1));
}

// Two-pass fallback inside ofl_splat_tiled_f32: the accumulator is bounded (fb_slots images, not the whole pass), so flagged
// images are served in rounds by their rank among the flagged ones -- accumulator slot of image n in this round, or -1 (not
// flagged / another round's).  The rank is a scan over the flags below n: this code only runs when some image IS flagged.
__device__ __forceinline__ int fb_slot(const SplatParams& p, int n) {
    if (p.run_if_set[n] == 0) return -1;
    int rank = 0;
    for (int i = 0; i < n; ++i) rank += p.run_if_set[i] != 0;
    return (rank / p.fb_slots == p.fb_round) ? rank % p.fb_slots : -1;
}

template <int CT, typename TF = float>
__device__ __forceinline__ void splat_fwd_tiles(const SplatParams& p) {
    // armed by device flags (the fallback inside ofl_splat_tiled_f32): a small strided grid walks the tiles of the flagged
    // images; otherwise one block per tile, XCD-aware
  for (int64_t tile = p.run_if_set ? (int64_t)blockIdx.x : logical_block(p.per_xcd); tile < p.total_tiles; tile += p.run_if_set ? (int64_t)gridDim.x : p.total_tiles) {
    const int tx = (int)(tile % p.tiles_x);
    const int ty = (int)((tile / p.tiles_x) % p.tiles_y);
    const int n = (int)(tile / ((int64_t)p.tiles_x * p.tiles_y));
    const int slot = p.run_if_set ? fb_slot(p, n) : n;
    if (slot < 0) continue;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = tx * kTileW + lane;
    const int w = p.w, h = p.h;
    const int64_t hw = (int64_t)h * w;
    const int C = CT ? CT : p.c;
    const int planes = 1 + C + (p.with_mask_chan ? 1 : 0);
    const float wmax = (float)(w - 1), hmax = (float)(h - 1);

    const TF* __restrict__ fu = p.flow ? reinterpret_cast<const TF*>(p.flow) + n * p.flow_bs : nullptr;
    const TF* __restrict__ db = reinterpret_cast<const TF*>(p.data) + n * p.data_bs;
    const float* __restrict__ dbb = p.data_b ? p.data_b + n * p.data_b_bs : nullptr;
    const uint8_t* __restrict__ wmk = p.weight_mask ? p.weight_mask + n * p.weight_mask_bs : nullptr;
    const uint8_t* __restrict__ cma = p.chan_mask_a ? p.chan_mask_a + n * p.chan_mask_a_bs : nullptr;
    const uint8_t* __restrict__ cmb = p.chan_mask_b ? p.chan_mask_b + n * p.chan_mask_b_bs : nullptr;
    float* __restrict__ acc = p.accum + (int64_t)slot * planes * hw;

#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int y = ty * kTileH + wave * kRows + r;
        if (x >= w || y >= h) continue;
        const int64_t pix = (int64_t)y * w + x;
        float xv, yv;
        bool zero = false;
        bool inside = true;
        const int64_t fpix = p.fw != 0 ? (int64_t)sp_win(p, x, y, inside) : pix;       // (padded apply: flow-geometry offset)
        const int64_t fhw = p.fw != 0 ? (int64_t)p.fh * p.fw : hw;
        if (fu) {
            const float u = ld1(fu + fpix), v = ld1(fu + fhw + fpix);
            xv = p.flow_sign * u + (float)x;  // get_flow_endpoints utils.py:1056-1057
            yv = p.flow_sign * v + (float)y;
            if (p.occlude) zero = (u < kZeroThr) && (u > -kZeroThr) && (v < kZeroThr) && (v > -kZeroThr);
        } else {
            xv = p.xs[n * p.xy_bs + pix];
            yv = p.ys[n * p.xy_bs + pix];
        }
        const bool wm = wmk ? (inside && wmk[fpix] != 0) : true;
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
        if (p.with_mask_chan) mval = ((cma ? cma[pix] != 0 : true) && (p.fw != 0 ? inside : true) && (cmb ? cmb[fpix] != 0 : true)) ? 1.0f : 0.0f;

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
                        atomicAdd(&acc[(int64_t)(1 + cc) * hw + pos], wgt * (p.data_sign * (dbb ? ld1(db + (int64_t)cc * hw + pix) - dbb[(int64_t)cc * hw + pix] : ld1(db + (int64_t)cc * hw + pix))));
                // mask channel: the reference accumulates wgt * mval next to the density.  All contributors of a
                // pixel being valid is the common case and must give ratio == 1 exactly, whatever order the
                // atomics land in -- so accumulate the INVALID weight instead and form den - inv in pass 2.
                if (p.with_mask_chan && mval == 0.0f) atomicAdd(&acc[(int64_t)(1 + C) * hw + pos], wgt);
            }
        }
    }
  }
}

template <int CT, typename TF = float>
__global__ __launch_bounds__(256) void splat_fwd_kernel(const SplatParams p) {
    if (p.run_if_set && *p.any_set == 0) return;
    splat_fwd_tiles<CT, TF>(p);
}

// ------------------------------------------------------------------------------------------------
// forward splat, pass 2 (ofl_splat_finalize_f32)
// ------------------------------------------------------------------------------------------------
template <int CT, typename TF = float, typename TO = float>
__device__ __forceinline__ void splat_finalize_tiles(const SplatParams& p) {
  for (int64_t tile = p.run_if_set ? (int64_t)blockIdx.x : logical_block(p.per_xcd); tile < p.total_tiles; tile += p.run_if_set ? (int64_t)gridDim.x : p.total_tiles) {
    const int tx = (int)(tile % p.tiles_x);
    const int ty = (int)((tile / p.tiles_x) % p.tiles_y);
    const int n = (int)(tile / ((int64_t)p.tiles_x * p.tiles_y));
    const int slot = p.run_if_set ? fb_slot(p, n) : n;
    if (slot < 0) continue;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = tx * kTileW + lane;
    const int w = p.w, h = p.h;
    const int64_t hw = (int64_t)h * w;
    const int C = CT ? CT : p.c;
    const int planes = 1 + C + (p.with_mask_chan ? 1 : 0);

    const TF* __restrict__ fu = p.flow ? reinterpret_cast<const TF*>(p.flow) + n * p.flow_bs : nullptr;
    const TF* __restrict__ db = reinterpret_cast<const TF*>(p.data) + n * p.data_bs;
    const float* __restrict__ dbb = p.data_b ? p.data_b + n * p.data_b_bs : nullptr;
    const uint8_t* __restrict__ wmk = p.weight_mask ? p.weight_mask + n * p.weight_mask_bs : nullptr;
    const uint8_t* __restrict__ cma = p.chan_mask_a ? p.chan_mask_a + n * p.chan_mask_a_bs : nullptr;
    const uint8_t* __restrict__ cmb = p.chan_mask_b ? p.chan_mask_b + n * p.chan_mask_b_bs : nullptr;
    const float* __restrict__ acc = p.accum + (int64_t)slot * planes * hw;
    TO* __restrict__ dst = reinterpret_cast<TO*>(p.dst) + (int64_t)n * p.dst_bs;
    int dflags = 0;

#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int y = ty * kTileH + wave * kRows + r;
        if (x >= w || y >= h) continue;
        const int64_t pix = (int64_t)y * w + x;
        const float den = acc[pix];
        const float dcl = p.raw ? 1.0f : (den < kDenMin ? kDenMin : den);  // clamp_min utils.py:1144 (raw sums: x / 1 = x)
        const bool warped = den > 0.0f;                    // utils.py:1197
        bool fill = false;
        bool inside = true;
        const int64_t fpix = p.fw != 0 ? (int64_t)sp_win(p, x, y, inside) : pix;       // (padded apply: flow-geometry offset)
        const int64_t fhw = p.fw != 0 ? (int64_t)p.fh * p.fw : hw;
        if (p.occlude && fu && !warped) {                  // un-occlude utils.py:1198-1203
            const float u = ld1(fu + fpix), v = ld1(fu + fhw + fpix);
            const bool zero = (u < kZeroThr) && (u > -kZeroThr) && (v < kZeroThr) && (v > -kZeroThr);
            const bool wm = wmk ? (inside && wmk[fpix] != 0) : true;
            fill = zero && wm;
        }
        float uv[2] = {0.0f, 0.0f};
#pragma unroll
        for (int ch = 0; ch < (CT ? CT : 1); ++ch)
            for (int cc = ch; cc < C; cc += (CT ? C : 1)) {
                float val = fill ? p.data_sign * (dbb ? ld1(db + (int64_t)cc * hw + pix) - dbb[(int64_t)cc * hw + pix] : ld1(db + (int64_t)cc * hw + pix)) : acc[(int64_t)(1 + cc) * hw + pix] / dcl;
                val = stored_as<TO>(apply_round(val, p.round_mode));
                st1(dst + (int64_t)cc * hw + pix, val);
                if (cc < 2) uv[cc] = val;
            }
        if (p.density) p.density[(int64_t)n * hw + pix] = den;
        if (p.warped) p.warped[(int64_t)n * hw + pix] = (uint8_t)warped;
        bool vld = true;
        if (p.valid || p.mask_chan) {
            float mch;
            if (fill)
                mch = ((cma ? cma[pix] != 0 : true) && (p.fw != 0 ? inside : true) && (cmb ? cmb[fpix] != 0 : true)) ? 1.0f : 0.0f;
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
}

template <int CT, typename TF = float, typename TO = float>
__global__ __launch_bounds__(256) void splat_finalize_kernel(const SplatParams p) {
    if (p.run_if_set && *p.any_set == 0) return;
    splat_finalize_tiles<CT, TF, TO>(p);
}

// The two-pass fallback INSIDE ofl_splat_tiled_f32 as ONE launch: until round 4 every call paid three guard launches (zero the
// accumulator, scatter, finalize) that left at once when no image was flagged -- 15 us of every call, a fifth of a call on a small
// frame.  This kernel leaves at once too; when some image IS flagged (a list overflowed: a pathological flow) its blocks run the
// three passes with a grid-wide barrier in between.  The grid is small enough to be resident as a whole (one block per CU), the
// barrier is an arrival counter in the call's statistics words (the last block to leave the second barrier zeroes it again for
// the next launch), and each side of it is a device-scope fence -- the accumulator is zeroed by plain stores that sit in one
// XCD's L2 until they are written back.
__device__ __forceinline__ void sp_grid_barrier(int32_t* counter, int32_t target) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    __threadfence();
}

template <int CT, typename TF = float, typename TO = float>
__global__ __launch_bounds__(256) void splat_fallback_kernel(const SplatParams p, float* __restrict__ accum, int64_t count_per_image,
                                                             int32_t n_images, int32_t* __restrict__ arrivals) {
    if (*p.any_set == 0) return;
    const int32_t blocks = (int32_t)gridDim.x;
    const int64_t n4 = count_per_image >> 2;
    for (int n = 0; n < n_images; ++n) {                  // zero the accumulator slots of this round's flagged images
        const int slot = fb_slot(p, n);
        if (slot < 0) continue;
        float* __restrict__ q = accum + (int64_t)slot * count_per_image;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)blocks * 256)
            reinterpret_cast<f4u*>(q)[i] = (f4){0.f, 0.f, 0.f, 0.f};
        if (blockIdx.x == 0 && threadIdx.x < (count_per_image & 3)) q[(n4 << 2) + threadIdx.x] = 0.0f;
    }
    sp_grid_barrier(arrivals, blocks);
    splat_fwd_tiles<CT, TF>(p);
    sp_grid_barrier(arrivals, 2 * blocks);
    // every block is past the second spin once it gets here: the last one to say so leaves both words zero
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(arrivals + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == blocks - 1) {
        __hip_atomic_store(arrivals, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(arrivals + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    splat_finalize_tiles<CT, TF, TO>(p);
}


// ------------------------------------------------------------------------------------------------
// forward splat, GATHER formulation (ofl_splat_tiled_f32): the destination tile finds its own sources
//
//  bin kernel   : every 16 x 2 SOURCE subtile (8 lanes x 4 pixels: half a DPP row) takes the bounding box of the
//                 destination pixels its end points touch and appends its 4-byte id to the list of each DESTINATION tile
//                 the box overlaps (fixed capacity; ranks from LDS atomics per 64 x 16 source region, one device-scope
//                 atomic per (region, destination tile)).  Reads the flow and the weight mask only (9 B/px), writes
//                 ~0.2 B/px.  A tile lists 28 - 32 subtiles on the bench flows.
//  gather kernel: one block per DESTINATION tile (64 x 16 since the end of round 4: OFL_SP_TW; 32 x 16 before).
//     A  walks its list, 64 subtiles per step in two half steps of 32 (the second skipped when empty): flow, mask and data
//        of the listed source pixels straight from the operand tensors (16-byte row-coalesced loads, served by L2 for the
//        ~2 tiles that share a subtile), end points in the reference's order (utils.py:1056-1057); the pixels whose unit
//        cell lies in the tile's (TW + 1) x 17 cells become RECORDS IN LDS ONLY (four corner weights, data, raster key +
//        mask-channel bit): ballot + popcount ranks, one LDS atomic per wave and half step; every record joins its CELL
//        (four 16-bit slots, later ones on a chain);
//     S  cells with one or two records need no order (a + b = b + a); the others are collected and handled one per lane:
//        put in raster order of their source pixels (up to 4: a sorting network in registers; 5 .. 64: an insertion sort
//        of the chain) and SUMMED there and then, per corner class and channel -- the sums replace the cell's records;
//     C  each thread sums its own 2 destination pixels in registers: its 3 x 2 cells, every record (or pre-summed entry)
//        fetched once and added to each corner-class sum it belongs to, then ((c0 + c1) + c2) + c3 -- exactly the order
//        of the reference's four scatter_add_ passes and its corner sum (utils.py:1133-1143), products rounded before
//        they are added: BIT-IDENTICAL to the reference, and run to run; normalise, threshold, un-occlude, 16-byte stores
//        (lane pairs exchange their halves through DPP), and, for flows, the output's flag word as a by-product.
//     More records than the LDS holds (896: a compression of the flow) are taken in 2 or 4 bands of destination rows,
//     each band walking the list again.
//  No record ever reaches HBM (the routed version of round 1 wrote and re-read 27 B/px of them), no float atomics, no
//  accumulator in HBM, no workspace beyond 1 KB per destination tile.
//  Only a heavy fold of the flow (> 64 sources in one cell, or more records than four bands hold) makes THAT tile fall
//  back to LDS float atomics over the same list, at once and in the same block (tolerance instead of bit-exactness for
//  that tile).  An IMAGE whose lists overflow (> 256 subtiles for one destination tile: a 16-fold compression) or whose
//  subtiles spread over > 256 destination tiles takes the two-pass global-atomics path inside the same call, decided on
//  the device, per image.
// ------------------------------------------------------------------------------------------------
#ifndef OFL_SP_LEAN
#define OFL_SP_LEAN 1
#endif
#ifndef OFL_SP_ISSUE_FIRST
#define OFL_SP_ISSUE_FIRST 1
#endif
#ifndef OFL_SP_TW
#define OFL_SP_TW 64    // width of a destination tile (32: 256-thread blocks, 4 per CU; 64: 512-thread blocks, 2 per CU -- fewer source pixels scanned per output pixel and half the per-tile prologues; round 3: 32 by 1 %, round 4 after the long cells left the critical path: 64 by 4 % at sigma 8, 30 % at B = 1)
#endif
#ifndef OFL_SP_TH
#define OFL_SP_TH 16
#endif
constexpr int kSpTW = OFL_SP_TW, kSpTH = OFL_SP_TH;          // destination tiles
static_assert(kSpTW == 32 || kSpTW == 64, "destination tile width");
constexpr int kSpNT2 = kSpTW * kSpTH / 2;                    // gather kernel: 2 destination pixels per thread
#ifndef OFL_SP_Q
#define OFL_SP_Q (kSpTW == 32 ? 896 : 1792)    // (32-wide tiles: 1024 leaves room for 3 blocks per CU only; 896: 39.7 KB per block, 4 blocks -- measured -4 % / -13 % on apply 's' / switch_ref)
#endif
constexpr int kSpQ = OFL_SP_Q;                               // records the gather kernel holds in LDS at a time
#ifndef OFL_SP_SUBH
#define OFL_SP_SUBH 2
#endif
constexpr int kSubW = 16, kSubH = OFL_SP_SUBH;               // source subtiles: 4 lanes x 4 pixels wide, kSubH rows
constexpr int kSubLanes = 4 * kSubH;                         // lanes per subtile (16: a DPP row; 8: half a row)
#ifndef OFL_SP_BING
#define OFL_SP_BING 2
#endif
constexpr int kBinG = OFL_SP_BING;                             // 4-pixel groups per lane of the bin kernel
constexpr int kRegH = 16 * kBinG;                            // a bin block covers a 64 x kRegH source region (its wave w: rows 16 g + 4 w .. + 3 of group g)
static_assert(kSubH == 1 || kSubH == 2 || kSubH == 4, "subtile height");
constexpr int kBinCap = (512 / kSubH) * (kSpTW / 32);        // subtiles one destination tile can list (fixed-address lists: 4 * kBinCap bytes per tile)
constexpr int kBinSpread = 256;                              // destination tiles one subtile may touch
constexpr unsigned kFallbackBlocks = 256;                    // grid of the two-pass fallback inside the tiled splat: one block per CU, resident as a whole
constexpr int kSpLong = 64;   // longest cell list (source pixels whose end points share one unit cell) that is summed in raster order

__device__ __forceinline__ uint32_t nz_bytes(uint32_t x) {   // per byte: non-zero -> 0x01 (SWAR: the low 7 bits carry into bit 7)
    return ((x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u;
}

struct GatherParams {
    SplatParams s;
    int32_t* cnt;          // [n * tiles] subtiles listed for the tile
    uint32_t* list;        // [n * tiles][kBinCap] subtile ids (row-major over the image's subtile grid)
    int32_t* stats;        // [0] some image took the two-pass path | [1] tiles that left the exact path | [2] images that did
    int32_t* img_over;     // [n] this image takes the two-pass path
    int32_t tiles_x, tiles_y;
    uint32_t tiles_img, mx_m, mx_s, mi_m, mi_s;
    int64_t total, per_xcd;
    int32_t subs_x; uint32_t sx_m, sx_s;                     // subtiles per image row (+ magic divisor)
    int32_t regs_x, regs_y;                                  // bin kernel: 64 x 16 source regions (4 waves x 4 subtiles)
    uint32_t regs_img, rx_m, rx_s, ri_m, ri_s;
    int64_t rtotal, rper_xcd;
    int32_t* redo_cnt;     // [1] tiles of this pass the one-scan gather kernel left to the second launch (bands / fold)
    uint32_t* redo_list;   // [kSpTH * n * tiles][2] (tile, first row | end row << 8): the BANDS of rows they were split into
};

// XCD-aware 32-bit decode of (column, row, image) from the block index
__device__ __forceinline__ bool decode3(int64_t total, int64_t per_xcd, uint32_t per_img, uint32_t mi_m, uint32_t mi_s,
                                        uint32_t mx_m, uint32_t mx_s, int32_t nx, int& tx, int& ty, int& n) {
    const uint32_t b = blockIdx.x;
    const uint32_t tile = (b & 7u) * (uint32_t)per_xcd + (b >> 3);
    if (tile >= (uint32_t)total) return false;
    const uint32_t nn = fastdiv(tile, mi_m, mi_s);
    const uint32_t rem = tile - nn * per_img;
    const uint32_t yy = fastdiv(rem, mx_m, mx_s);
    n = (int)nn; ty = (int)yy; tx = (int)(rem - yy * (uint32_t)nx);
    return true;
}

// end point + contribution test of the 4 source pixels of a thread
struct SpSrc { float x[4], y[4]; uint32_t on; };   // on: bit k = pixel k contributes (inside the image, weight mask set, not an occluded zero-flow pixel)

// end points and contribution test from the loaded flow (or positions) and weight-mask bytes of a 4-pixel group
template <typename SP>
__device__ __forceinline__ void sp_finish_src(const SP& s, int sx4, int sy, bool inimg, const f4& a, const f4& b, uint32_t wm4, SpSrc& q) {
    q.on = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bool zero = false;
        if (OFL_SP_HAS_FLOW(s)) {
            // get_flow_endpoints utils.py:1056-1057.  flow_sign is +1 or -1: the product is exact, so ONE fma rounds exactly
            // as the reference's add does (the entry points reject any other sign)
            q.x[k] = __builtin_fmaf(s.flow_sign, a[k], (float)(sx4 + k));
            q.y[k] = __builtin_fmaf(s.flow_sign, b[k], (float)sy);
            if (s.occlude) zero = (a[k] < kZeroThr) && (a[k] > -kZeroThr) && (b[k] < kZeroThr) && (b[k] > -kZeroThr);
        } else {
            q.x[k] = a[k]; q.y[k] = b[k];
        }
        const bool wm = ((wm4 >> (8 * k)) & 0xffu) != 0u;
        if (inimg && (sx4 + k < s.w) && wm && !zero) q.on |= 1u << k;
    }
}

template <typename TF = float, typename SP>
__device__ __forceinline__ void sp_load_src(const SP& s, int n, int sx4, int sy, bool inimg, uint32_t pix, uint32_t hw, SpSrc& q) {
    const TF* __restrict__ flw = reinterpret_cast<const TF*>(s.flow);     // (fp16 variants: the flow planes hold halves)
    f4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    uint32_t wm4 = 0x01010101u;
    // the last group of a row of an image whose width is not a multiple of 4: fetch the last whole group and rotate
    const int wrem = OFL_SP_WREM(s);
    const bool edge = wrem != 0 && inimg && sx4 > s.w - 4;
    const uint32_t pe = edge ? pix - (uint32_t)(4 - wrem) : pix;
    if (inimg && OFL_SP_WINDOW(s)) {
        // padded apply: per-pixel loads with replicate addressing (not a hot path); the weight mask is False outside the window
        const uint32_t fhw = (uint32_t)(s.fh * s.fw);
        wm4 = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool inside;
            const uint32_t off = sp_win(s, min(sx4 + k, s.w - 1), sy, inside);
            a[k] = ld1(flw + n * s.flow_bs + off); b[k] = ld1(flw + n * s.flow_bs + fhw + off);
            const bool m = s.weight_mask ? (inside && s.weight_mask[n * s.weight_mask_bs + off] != 0) : true;
            wm4 |= (uint32_t)m << (8 * k);
        }
    } else if (inimg) {
        if (OFL_SP_HAS_FLOW(s)) {
            a = ld4(flw + n * s.flow_bs + pe);
            b = ld4(flw + n * s.flow_bs + hw + pe);
        } else {
            a = ld4(s.xs + n * s.xy_bs + pe);
            b = ld4(s.ys + n * s.xy_bs + pe);
        }
        if (s.weight_mask) wm4 = ld32(s.weight_mask + n * s.weight_mask_bs + pe);
        if (wrem != 0) {
            if (edge) { a = rot4(a, 4 - wrem); b = rot4(b, 4 - wrem); wm4 >>= 8 * (4 - wrem); }
        }
    }
    sp_finish_src(s, sx4, sy, inimg, a, b, wm4, q);
}

// The same in two halves, for the gather kernel's scan: _issue only LOADS (flow or positions, weight mask), _done rotates a row-end
// group and forms the end points.  The scan issues the loads of BOTH half steps and of their data (sp_issue_data) before anything is
// consumed: written as sp_load_src + sp_load_data the compiler waited for the flow before it issued the data loads -- the divergent
// `if (inimg)` blocks keep it from moving loads across the first use -- and a step was THREE dependent round trips (list, flow,
// data) instead of two (round 5: -8 % on apply 's', -9 % on switch_ref at sigma 8).
struct SpRaw { f4 a, b; uint32_t wm4; int rot; };
template <typename TF = float, typename SP>
__device__ __forceinline__ void sp_issue_src(const SP& s, int n, int sx4, int sy, bool inimg, uint32_t pix, uint32_t hw, SpRaw& r) {
    const TF* __restrict__ flw = reinterpret_cast<const TF*>(s.flow);
    r.a = (f4){0.f, 0.f, 0.f, 0.f}; r.b = (f4){0.f, 0.f, 0.f, 0.f};
    r.wm4 = 0x01010101u;
    const int wrem = OFL_SP_WREM(s);
    const bool edge = wrem != 0 && inimg && sx4 > s.w - 4;
    const uint32_t pe = edge ? pix - (uint32_t)(4 - wrem) : pix;
    r.rot = edge ? 4 - wrem : 0;
    if (inimg && OFL_SP_WINDOW(s)) {
        // padded apply: per-pixel loads with replicate addressing (not a hot path); the weight mask is False outside the window
        const uint32_t fhw = (uint32_t)(s.fh * s.fw);
        r.wm4 = 0u; r.rot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool inside;
            const uint32_t off = sp_win(s, min(sx4 + k, s.w - 1), sy, inside);
            r.a[k] = ld1(flw + n * s.flow_bs + off); r.b[k] = ld1(flw + n * s.flow_bs + fhw + off);
            const bool m = s.weight_mask ? (inside && s.weight_mask[n * s.weight_mask_bs + off] != 0) : true;
            r.wm4 |= (uint32_t)m << (8 * k);
        }
    } else if (inimg) {
        if (OFL_SP_HAS_FLOW(s)) {
            r.a = ld4(flw + n * s.flow_bs + pe);
            r.b = ld4(flw + n * s.flow_bs + hw + pe);
        } else {
            r.a = ld4(s.xs + n * s.xy_bs + pe);
            r.b = ld4(s.ys + n * s.xy_bs + pe);
        }
        if (s.weight_mask) r.wm4 = ld32(s.weight_mask + n * s.weight_mask_bs + pe);
    }
}
template <typename SP>
__device__ __forceinline__ void sp_src_done(const SP& s, int sx4, int sy, bool inimg, SpRaw& r, SpSrc& q) {
    // (locals, not in-place edits of `r`: the in-place form of sp_data_done's rotation was MISCOMPILED by ROCm 7.2 -- planes 1 and 2 of
    // a row-end group came out unrotated; tests/test_gpu_parity.py::test_tiled_splat_matches_two_pass_and_oracle[0.0-shape4] caught it)
    const int rot = (OFL_SP_WREM(s) != 0) ? r.rot : 0;             // (block-uniform test first)
    f4 a = r.a, b = r.b;
    uint32_t wm4 = r.wm4;
    if (rot != 0) { a = rot4(a, rot); b = rot4(b, rot); wm4 >>= 8 * rot; }
    sp_finish_src(s, sx4, sy, inimg, a, b, wm4, q);
}

// min / max of two packed 16-bit fields over the lanes of a subtile (16: a DPP row, 8: half a row; every lane gets the result)
// (OFL_DPPU: these four controls are permutations of a row -- every lane has a source, so the "old" value of update_dpp is never used and
// need not be copied in first: one v_mov_b32_dpp per step instead of a v_mov_b32 and one)
#define OFL_DPPU(v, ctrl) __builtin_amdgcn_mov_dpp((v), (ctrl), 0xf, 0xf, true)
__device__ __forceinline__ int row_pk_min_dpp(int v) {
    v = pk_min16(v, OFL_DPPU(v, 0xB1)); v = pk_min16(v, OFL_DPPU(v, 0x4E));
    if (kSubLanes == 4) return v;                                          // (a quad: the two quad permutations complete it)
    v = pk_min16(v, OFL_DPPU(v, 0x141));
    return kSubLanes == 16 ? pk_min16(v, OFL_DPPU(v, 0x140)) : v;         // (8 lanes: the half-row mirror completes it)
}
__device__ __forceinline__ int row_pk_max_dpp(int v) {
    v = pk_max16(v, OFL_DPPU(v, 0xB1)); v = pk_max16(v, OFL_DPPU(v, 0x4E));
    if (kSubLanes == 4) return v;
    v = pk_max16(v, OFL_DPPU(v, 0x141));
    return kSubLanes == 16 ? pk_max16(v, OFL_DPPU(v, 0x140)) : v;
}

#ifndef OFL_BIN_SMALLDIV
#define OFL_BIN_SMALLDIV 1
#endif
constexpr int kBinLocal = 64;   // destination tiles one 64 x 16 source region aggregates in LDS (more: straight to the global counters)

// image n leaves the gather path (a list overflowed / a subtile is torn over the frame): its flag, and the statistics of
// the call -- stats[0] some image took the two-pass path, stats[2] how many (the first thread to flag an image counts it)
__device__ __forceinline__ void sp_flag_image(const GatherParams& p, int n) {
    if (atomicOr(&p.img_over[n], 1) == 0) { atomicOr(&p.stats[0], 1); atomicAdd(&p.stats[2], 1); }
}

// a / b for 0 <= a < 4096, 1 <= b <= 64, exactly: (a + 0.5) * rcp(b) lies at least 0.5 / b >= 2^-7 away from every integer, far beyond the error
// of v_rcp_f32 and two roundings (the compiler's 32-bit division is ~20 instructions, and the bin kernel is bound by the instructions it
// issues: 625 per wave = 66 of its 73 us, profiles/r6_splat_diet.txt)
__device__ __forceinline__ int small_div(int a, int b) { return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b)); }

template <typename TF, bool LEAN = false>
__global__ __launch_bounds__(256) void splat_bin_kernel(const GatherParams p) {
    // Round 6: a block covers a 64 x (16 kBinG) source region -- every lane kBinG 4-pixel groups, 16 rows apart, their loads issued
    // together -- so that the chain load -> LDS ranks -> ONE device-scope atomic per (region, destination tile) -> list stores is paid
    // once per 2 048 pixels instead of once per 1 024 (the kernel is that chain, not bandwidth: 80 us at 3.9 TB/s of its 9 B/px)
    constexpr int G = kBinG;
    __shared__ int red[4][2];
    __shared__ int lcount[kBinLocal], lbase[kBinLocal];
    int rx, ry, n;
    if (!decode3(p.rtotal, p.rper_xcd, p.regs_img, p.ri_m, p.ri_s, p.rx_m, p.rx_s, p.regs_x, rx, ry, n)) return;
    const typename std::conditional<LEAN, SplatParamsLean, SplatParams>::type& s =
        reinterpret_cast<const typename std::conditional<LEAN, SplatParamsLean, SplatParams>::type&>(p.s);   // (see SplatParamsLean)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane >> 4, r = (lane >> 2) & 3, c4 = lane & 3;       // subtile of the wave, row and 4-pixel group in it
    const int w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    const int sx4 = rx * (4 * kSubW) + sub * kSubW + c4 * 4;
    int sy[G];
    SpSrc q[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        sy[g] = ry * kRegH + g * 16 + wave * 4 + r;
        sp_load_src<TF>(s, n, sx4, sy[g], (sx4 < w) && (sy[g] < h), (uint32_t)(sy[g] * w + sx4), hw, q[g]);
    }
    if (tid < kBinLocal) lcount[tid] = 0;
    // destination pixels the four corners of this thread's end points touch (clamped corners carry weight 0,
    // utils.py:1106-1111: they touch nothing)
    const float wf = (float)w, hf = (float)h;
    int minx[G], miny[G], maxx[G], maxy[G];
    bool any[G];
    int blo_w = 0x7fff7fff, bhi_w = (int)0xffffffffu;                    // union over the wave's subtiles (both groups)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        int lo = 0x7fff7fff, hi = (int)0xffffffffu;                      // (x, y) = (32767, 32767) / (-1, -1)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((q[g].on >> k) & 1u) {
                const int x0 = (int)__builtin_amdgcn_fmed3f(floorf(q[g].x[k]), -2.0f, wf), y0 = (int)__builtin_amdgcn_fmed3f(floorf(q[g].y[k]), -2.0f, hf);
                const int xa = max(x0, 0), xb = min(x0 + 1, w - 1), ya = max(y0, 0), yb = min(y0 + 1, h - 1);
                if (xa <= xb && ya <= yb) {
                    lo = pk_min16(lo, (int)((uint32_t)xa | ((uint32_t)ya << 16)));
                    hi = pk_max16(hi, (int)((uint32_t)xb | ((uint32_t)yb << 16)));
                }
            }
        }
        lo = row_pk_min_dpp(lo); hi = row_pk_max_dpp(hi);                // over the subtile
        minx[g] = (int)(short)(lo & 0xffff); miny[g] = lo >> 16; maxx[g] = (int)(short)(hi & 0xffff); maxy[g] = hi >> 16;
        any[g] = maxx[g] >= minx[g] && maxy[g] >= miny[g];               // something of this subtile lands inside the image
        // destination tiles of the subtile (as packed 16-bit pairs, for the block-wide union)
        // (the extents are never negative where they are used: unsigned divisions -- shifts; the signed ones cost three instructions more each)
        const int tlo = any[g] ? (int)(((uint32_t)minx[g] / (uint32_t)kSpTW) | (((uint32_t)miny[g] / (uint32_t)kSpTH) << 16)) : 0x7fff7fff;
        const int thi = any[g] ? (int)(((uint32_t)maxx[g] / (uint32_t)kSpTW) | (((uint32_t)maxy[g] / (uint32_t)kSpTH) << 16)) : (int)0xffffffffu;
        // (the subtiles of a ROW of 16 lanes first, by DPP; then one readlane per row instead of one per subtile)
        int rlo = tlo, rhi = thi;
        if (kSubLanes <= 4) { rlo = pk_min16(rlo, OFL_DPPU(rlo, 0x141)); rhi = pk_max16(rhi, OFL_DPPU(rhi, 0x141)); }
        if (kSubLanes <= 8) { rlo = pk_min16(rlo, OFL_DPPU(rlo, 0x140)); rhi = pk_max16(rhi, OFL_DPPU(rhi, 0x140)); }
#pragma unroll
        for (int l = 0; l < 64; l += 16) {
            blo_w = pk_min16(blo_w, __builtin_amdgcn_readlane(rlo, l)); bhi_w = pk_max16(bhi_w, __builtin_amdgcn_readlane(rhi, l));
        }
    }
    if (lane == 0) { red[wave][0] = blo_w; red[wave][1] = bhi_w; }
    __syncthreads();
    int blo = red[0][0], bhi = red[0][1];
#pragma unroll
    for (int i = 1; i < 4; ++i) { blo = pk_min16(blo, red[i][0]); bhi = pk_max16(bhi, red[i][1]); }
    const int btx0 = (int)(short)(blo & 0xffff), bty0 = blo >> 16, btx1 = (int)(short)(bhi & 0xffff), bty1 = bhi >> 16;
    if (btx1 < btx0 || bty1 < bty0) return;                              // nothing of this region lands inside the image (block-uniform)
    const int bntx = btx1 - btx0 + 1, bnt = bntx * (bty1 - bty0 + 1);
    const bool local = bnt <= kBinLocal;                                 // block-uniform
    const int j0 = lane & (kSubLanes - 1);
    int tx0[G], ty0[G], ntx[G], cnt[G], lt[G], rank[G];
    uint32_t subid[G];
    bool fast[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        tx0[g] = (int)((uint32_t)minx[g] / (uint32_t)kSpTW); ty0[g] = (int)((uint32_t)miny[g] / (uint32_t)kSpTH);
        const int tx1 = (int)((uint32_t)maxx[g] / (uint32_t)kSpTW), ty1 = (int)((uint32_t)maxy[g] / (uint32_t)kSpTH);
        ntx[g] = tx1 - tx0[g] + 1; cnt[g] = any[g] ? ntx[g] * (ty1 - ty0[g] + 1) : 0;
        subid[g] = (uint32_t)((sy[g] / kSubH) * p.subs_x + rx * 4 + sub);
        if (cnt[g] > kBinSpread) {                                       // a subtile torn over the whole frame: two-pass path
            if ((lane & (kSubLanes - 1)) == 0) sp_flag_image(p, n);
        }
        // the common case: at most kSubLanes destination tiles per subtile (one per lane), all inside the region's local grid -- ranks
        // from LDS atomics, then ONE global atomic per (source region, destination tile) instead of one per (subtile, tile):
        // device-scope atomics go all the way to memory and cost ~1 us each
        lt[g] = -1; rank[g] = 0;
        fast[g] = local && cnt[g] <= kSubLanes;
        if (fast[g] && j0 < cnt[g]) {
            const int jy = OFL_BIN_SMALLDIV ? small_div(j0, ntx[g]) : j0 / ntx[g];
            lt[g] = (ty0[g] + jy - bty0) * bntx + (tx0[g] + (j0 - jy * ntx[g]) - btx0);
            rank[g] = atomicAdd(&lcount[lt[g]], 1);
        }
    }
    if (local) {
        __syncthreads();
        if (tid < bnt && lcount[tid] > 0) {
            const int ly_ = OFL_BIN_SMALLDIV ? small_div(tid, bntx) : tid / bntx;
            const int64_t d = (int64_t)n * p.tiles_img + (bty0 + ly_) * p.tiles_x + btx0 + (tid - ly_ * bntx);
            const int start = atomicAdd(&p.cnt[d], lcount[tid]);
            lbase[tid] = start;
            if (start + lcount[tid] > kBinCap) sp_flag_image(p, n);
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (lt[g] >= 0) {
                const int ly_ = OFL_BIN_SMALLDIV ? small_div(lt[g], bntx) : lt[g] / bntx;
                const int64_t d = (int64_t)n * p.tiles_img + (bty0 + ly_) * p.tiles_x + btx0 + (lt[g] - ly_ * bntx);
                const int pos = lbase[lt[g]] + rank[g];
                if (pos < kBinCap) p.list[d * kBinCap + pos] = subid[g];
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (!fast[g] && cnt[g] <= kBinSpread) {                          // wide spreads: straight to the global counters
            for (int j = j0; j < cnt[g]; j += kSubLanes) {
                const int jy = j / ntx[g];
                const int64_t d = (int64_t)n * p.tiles_img + (ty0[g] + jy) * p.tiles_x + tx0[g] + (j - jy * ntx[g]);
                const int pos = atomicAdd(&p.cnt[d], 1);
                if (pos < kBinCap) p.list[d * kBinCap + pos] = subid[g];
                else sp_flag_image(p, n);
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
#define OFL_SP_MINB 4   // waves per SIMD the gather kernel's register budget is sized for (4 blocks of 256 threads / 2 of 512 per CU: what its LDS allows)
#endif
#ifndef OFL_SP_U
#define OFL_SP_U 1      // list entries (subtiles) per thread and step of the gather kernel's walk
#endif
constexpr int kSpU = OFL_SP_U;
// A RECORD is 32 bytes of LDS: the four corner weights w[ky][kx] = wy[ky] * wx[kx] of its end point -- wx = (x1 - x, x - x0),
// wy alike, each product formed once, exactly as the reference's outer product (utils.py:1110-1114; a corner that a
// destination pixel of the image reads is never clamped, so the `eq` factor is 1) -- then up to 3 data values and the key
// (raster position of the source pixel, mask-channel bit below it).  Two ds_read_b128 bring a record.
//
// One record of the cell whose column is DC (-1, 0, +1) cells from the pair's middle cell and whose row serves corner
// row KY, added to the sums of the destination pixels that read it: pixel 0 of the pair as its x-corner 1 (DC = -1) or
// 0 (DC = 0), pixel 1 as its x-corner 1 (DC = 0) or 0 (DC = +1).  Product rounded, then added.
template <int NC, int NCH, int DC, int KY>
__device__ __forceinline__ void sp_use(const f4* rec4, uint32_t i, float (&a)[2][2][1 + NCH]) {
    const f4 wv = rec4[2 * i], dv = rec4[2 * i + 1];
    float d[NCH];
#pragma unroll
    for (int c = 0; c < NC; ++c) d[c] = dv[c];
    if (NCH > NC) d[NC] = (float)(__float_as_uint(dv[3]) & 1u);          // the mask channel rides in the key
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int kx = k - DC;                     // pixel k sits DC .. DC + 1 columns right of the cell: x-corner k - DC
        if (kx < 0 || kx > 1) continue;
        const float wgt = wv[KY * 2 + kx];
        a[k][kx][0] += wgt;
#pragma unroll
        for (int c = 0; c < NCH; ++c) a[k][kx][1 + c] += wgt * d[c];
    }
}

// A cell that phase S has already summed (more than two records): its class sums lie where its first three records were,
// one f4 per channel (density, data ..., mask) holding the four corner classes in the order of a record's weights.
template <int NCH, int DC, int KY>
__device__ __forceinline__ void sp_use_presum(const f4* rec4, uint32_t ia, uint32_t ib, uint32_t ic, float (&a)[2][2][1 + NCH]) {
#pragma unroll
    for (int c = 0; c < 1 + NCH; ++c) {
        const f4 pv = c < 2 ? rec4[2 * ia + c] : c < 4 ? rec4[2 * ib + (c - 2)] : rec4[2 * ic];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int kx = k - DC;
            if (kx < 0 || kx > 1) continue;
            a[k][kx][c] += pv[KY * 2 + kx];
        }
    }
}

#ifndef OFL_SP_LANEMAX
#define OFL_SP_LANEMAX 12
#endif
#ifndef OFL_SP_NET8
#define OFL_SP_NET8 1     // cells of five to N records are ordered by a sorting network in registers (1: N by channel count; 6 / 8: forced; 0: insertion sort up to OFL_SP_LANEMAX)
#endif
constexpr int kSpLaneMax = OFL_SP_LANEMAX;   // longest cell ONE lane puts in raster order (insertion on its chain); longer ones (up to kSpLong) are a wave's

// A cell with kSpLaneMax < cn <= 64 records, ordered and summed by the 64 lanes of one wave (every lane of the wave calls this
// with the same cell).  Leaves what the single-lane path leaves: the four class sums of every channel in the cell's first
// three records (raster order) and the kLongCell marker in its slots.
template <int NC, int NCH>
__device__ __attribute__((noinline)) void sp_order_big_cell(unsigned char* raw, const uint32_t* ccnt, uint2* slots, const uint32_t* ohead,
                                                  const uint16_t* link, int c, int lane) {
    constexpr uint32_t kEnd = 0xffffu, kLongCell = 0xfffeu;
    uint32_t* recw = reinterpret_cast<uint32_t*>(raw);
    float* recf = reinterpret_cast<float*>(raw);
    const int cn = (int)__builtin_amdgcn_readfirstlane((int)ccnt[c]);
    // 1. the cell's record indices, one per lane: its four slots, then its chain (walked once; every lane reads the same word)
    const uint2 sl4 = slots[c];
    uint32_t idx = lane == 0 ? (sl4.x & 0xffffu) : lane == 1 ? (sl4.x >> 16) : lane == 2 ? (sl4.y & 0xffffu) : (sl4.y >> 16);
    uint32_t cur = ohead[c];
    for (int j = 4; j < cn; ++j) {
        const uint32_t cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
        idx = lane == j ? cs : idx;
        cur = cs != kEnd ? (uint32_t)link[cs] : kEnd;
    }
    const bool mine = lane < cn;
    // 2. rank of every record = number of keys below its own (keys are source positions: all different)
    const uint32_t key = mine ? recw[8 * idx + 7] : 0xffffffffu;
    uint32_t rank = 0;
    for (int t = 0; t < cn; ++t) rank += ((uint32_t)__builtin_amdgcn_readlane((int)key, t) < key) ? 1u : 0u;
    // 3. lane r gets the index of the r-th record in raster order (forward permute among the cn active lanes)
    uint32_t sorted = 0;
    if (mine) sorted = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)idx);
    // 4. lane (class k, channel ch) adds its sum over the records in that order
    const int k = lane & 3, ch = lane >> 2;
    const bool acc_on = ch < 1 + NCH;
    const bool is_mask = NCH > NC && ch == 1 + NC;
    const int doff = (ch >= 1 && ch <= NC) ? 3 + ch : 7;          // word of the record this lane multiplies the weight with
    float acc = 0.0f;
    for (int r = 0; r < cn; ++r) {
        const uint32_t ir = (uint32_t)__builtin_amdgcn_readlane((int)sorted, r);
        const float wv = recf[8 * ir + k];
        const uint32_t dw = recw[8 * ir + doff];
        const float dv = ch == 0 ? 1.0f : (is_mask ? (float)(dw & 1u) : __uint_as_float(dw));
        acc += wv * dv;                                            // (density: the weight itself -- w * 1 is exact)
    }
    // 5. the class sums where the cell's readers expect them: one f4 (the four classes) per channel, in the first three records
    const uint32_t ia = (uint32_t)__builtin_amdgcn_readlane((int)sorted, 0), ib = (uint32_t)__builtin_amdgcn_readlane((int)sorted, 1);
    const uint32_t ic = (uint32_t)__builtin_amdgcn_readlane((int)sorted, 2);
    if (acc_on) {
        const uint32_t f4i = ch < 2 ? 2 * ia + ch : ch < 4 ? 2 * ib + (ch - 2) : 2 * ic;
        recf[4 * f4i + k] = acc;
    }
    if (lane == 0) slots[c] = make_uint2(kLongCell | (ia << 16), ib | (ic << 16));
}

// exchange with the neighbouring lane (lanes 2j <-> 2j + 1): the pair's partner owns the other half of a 4-pixel group
__device__ __forceinline__ float swap1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)); }
__device__ __forceinline__ uint32_t swap1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }

// what a thread of a destination tile knows about its 2 output pixels (gather kernel and per-tile fallback kernel)
struct SpTile {
    int n, dx0, dy0, lx2, ly;
    uint32_t pix;            // offset of the pair in a plane
    bool solo, inimg, wide;  // solo: the tile is one pixel wide here; wide: a whole tile (lane pairs store 4 pixels at a time)
    bool fill_ok[2];         // un-occlude fill candidates (utils.py:1198-1203)
};

template <typename TF = float, typename SP>
__device__ __forceinline__ void sp_tile_setup(const SP& s, int tx, int ty, int n, SpTile& t) {
    const TF* __restrict__ flw = reinterpret_cast<const TF*>(s.flow);
    const int tid = threadIdx.x, w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    t.n = n; t.dx0 = tx * kSpTW; t.dy0 = ty * kSpTH;
    const int lx = tid % (kSpTW / 2);                                // (pairs of pixels along a row of the tile)
    t.ly = tid / (kSpTW / 2);
    const int x2 = min(t.dx0 + lx * 2, w - 2), y = t.dy0 + t.ly;     // (odd widths: the last pair re-computes pixel w - 2)
    t.lx2 = x2 - t.dx0;                                              // tile-local column of the pair (may be lx * 2 - 1)
    t.solo = t.lx2 < 0;
    t.inimg = (t.dx0 + lx * 2 < w) && (y < h);
    t.pix = (uint32_t)(min(y, h - 1) * w + x2);
    t.wide = t.dx0 + kSpTW <= w;
    t.fill_ok[0] = t.fill_ok[1] = false;
    if (s.occlude && OFL_SP_HAS_FLOW(s) && t.inimg) {
        f2 a, b;
        uint32_t wm2 = 0x0101u;
        if (OFL_SP_WINDOW(s)) {                                            // padded apply (see sp_win)
            const uint32_t fhw = (uint32_t)(s.fh * s.fw);
            wm2 = 0u;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                bool inside;
                const uint32_t off = sp_win(s, x2 + k, min(y, h - 1), inside);
                a[k] = ld1(flw + n * s.flow_bs + off); b[k] = ld1(flw + n * s.flow_bs + fhw + off);
                const bool m = s.weight_mask ? (inside && s.weight_mask[n * s.weight_mask_bs + off] != 0) : true;
                wm2 |= (uint32_t)m << (8 * k);
            }
        } else {
            a = ld2(flw + n * s.flow_bs + t.pix); b = ld2(flw + n * s.flow_bs + hw + t.pix);
            if (s.weight_mask) wm2 = ld16(s.weight_mask + n * s.weight_mask_bs + t.pix);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            t.fill_ok[k] = (a[k] < kZeroThr) && (a[k] > -kZeroThr) && (b[k] < kZeroThr) && (b[k] > -kZeroThr) && (((wm2 >> (8 * k)) & 0xffu) != 0u);
    }
}

// normalise, masks, un-occlude fill, store (tot: density, channels, mask channel).  EVERY thread of the block calls it
// (`mine`: this thread's row is being finalized), so that lane pairs can exchange their halves.
template <int NC, bool MCH, typename TF = float, typename TO = float, typename SP>
__device__ __forceinline__ void sp_finalize(const SP& s, const SpTile& t, const float (&tot)[2][1 + NC + (MCH ? 1 : 0)], bool mine, int& dflags) {
    const int n = t.n, tid = threadIdx.x;
    const uint32_t hw = (uint32_t)(s.h * s.w);
    // the pixel offset is made opaque HERE: otherwise every per-lane output address (64 bits each, a dozen of them) is
    // computed once before the band loop, lives across the whole kernel and is spilled to scratch -- 16 scratch stores per
    // thread and tile, 2 GB of scratch writes per launch (rocprofv3 WRITE_SIZE: profiles/r2_splat_gather_first_cut_pmc.txt)
    uint32_t pix = t.pix;
    asm volatile("" : "+v"(pix));
    static_assert(std::is_same<TO, float>::value || NC == 2, "fp16 outputs are flows");
    const TF* __restrict__ db = reinterpret_cast<const TF*>(s.data) + n * s.data_bs;
    const float* __restrict__ dbb = s.data_b ? s.data_b + n * s.data_b_bs : nullptr;
    f2 den2, out[NC], mch2;
    uint32_t warped2 = 0, valid2 = 0;
    bool fill[2];
    // the un-occlude fill (utils.py:1198-1203) touches a pixel here and there: the common wave has none, and takes no part
    // of that code (wave-uniform branch below) -- first every pixel as a warped one
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float den = tot[k][0];
        const float dcl = OFL_SP_RAW(s) ? 1.0f : (den < kDenMin ? kDenMin : den);          // clamp_min utils.py:1144 (raw sums: x / 1 = x)
        const bool warped = den > 0.0f;                            // utils.py:1197
        fill[k] = t.fill_ok[k] && !warped && mine;
        den2[k] = den;
        warped2 |= (uint32_t)warped << (8 * k);
#pragma unroll
        for (int c = 0; c < NC; ++c) out[c][k] = stored_as<TO>(apply_round(tot[k][1 + c] / dcl, OFL_SP_ROUND(s)));
        if (MCH) {
            // every contributor valid: the mask channel's sums ARE the density's (the same additions of the same weights),
            // and x / x = 1 -- no division where the whole wave is in that case (all but the neighbourhood of mask holes)
            const float num = tot[k][1 + NC];
            const bool unit = num == den && den >= kDenMin && !OFL_SP_RAW(s);
            mch2[k] = __all(unit) ? 1.0f : num / dcl;
        }
    }
    if (__any(fill[0] || fill[1])) {                               // (wave-uniform)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!fill[k]) continue;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                out[c][k] = stored_as<TO>(apply_round(s.data_sign * ((NC <= 2 && dbb) ? ld1(db + c * hw + pix + k) - dbb[c * hw + pix + k] : ld1(db + c * hw + pix + k)), OFL_SP_ROUND(s)));
            if (MCH) {
                const bool a = s.chan_mask_a ? s.chan_mask_a[n * s.chan_mask_a_bs + pix + k] != 0 : true;
                bool b = s.chan_mask_b ? s.chan_mask_b[n * s.chan_mask_b_bs + pix + k] != 0 : true;
                if (OFL_SP_WINDOW(s)) {                                    // padded apply: chan_mask_b lives in the flow window, False outside
                    bool inside;
                    const uint32_t off = sp_win(s, (int)((pix + k) % (uint32_t)s.w), (int)((pix + k) / (uint32_t)s.w), inside);
                    b = inside && (s.chan_mask_b ? s.chan_mask_b[n * s.chan_mask_b_bs + off] != 0 : true);
                }
                mch2[k] = (a && b) ? 1.0f : 0.0f;
            }
        }
    }
    if (MCH) {
#pragma unroll
        for (int k = 0; k < 2; ++k) valid2 |= (uint32_t)(mch2[k] > kValidThr) << (8 * k);
    }
    if (mine && NC == 2 && s.dst_flags) {             // the output read as a flow under its valid mask (by-product)
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k == 1 || !t.solo) dflags |= flag_bits(out[0][k], out[NC - 1][k], MCH ? ((valid2 >> (8 * k)) & 1u) != 0u : true);
    }
    TO* __restrict__ dst = reinterpret_cast<TO*>(s.dst) + (int64_t)n * s.dst_bs;
    if (t.wide) {
        // lanes 2j / 2j + 1 own pixels 4j .. 4j + 1 / 4j + 2 .. 4j + 3 of one row: the even lane stores the even planes of
        // the 4-pixel group, the odd lane the odd ones -- 16 bytes per lane instead of 8 (partial-line stores are the
        // expensive ones on this memory system)
        const bool odd = (tid & 1) != 0;
        const uint32_t pq = pix - (odd ? 2u : 0u);               // first pixel of the group
        // plane A goes out through the even lane, plane B through the odd one; each gets the partner's half by DPP
        auto emit = [&](const f2& pa, auto* ptra, bool a_on, const f2& pb, auto* ptrb, bool b_on) {
            const f2 give = odd ? pa : pb, keep = odd ? pb : pa;
            const f2 got = {swap1(give[0]), swap1(give[1])};
            const f4 v = odd ? (f4){got[0], got[1], keep[0], keep[1]} : (f4){keep[0], keep[1], got[0], got[1]};
            if (mine && (odd ? b_on : a_on)) { if (odd) st4o(ptrb + pq, v); else st4o(ptra + pq, v); }
        };
        float* dpl = s.density ? s.density + (int64_t)n * hw : nullptr;
        float* mpl = (MCH && s.mask_chan) ? s.mask_chan + (int64_t)n * hw : nullptr;
        if (NC == 1) {
            emit(out[0], dst, true, den2, dpl, dpl != nullptr);
            if (MCH && s.mask_chan) emit(mch2, mpl, true, mch2, mpl, false);          // (block-uniform)
        } else if (NC == 2) {
            emit(out[0], dst, true, out[NC - 1], dst + hw, true);
            if (dpl || mpl) emit(den2, dpl, dpl != nullptr, mch2, mpl, mpl != nullptr);
        } else {
            emit(out[0], dst, true, out[1 % NC], dst + hw, true);
            emit(out[2 % NC], dst + 2 * hw, true, den2, dpl, dpl != nullptr);
            if (MCH && s.mask_chan) emit(mch2, mpl, true, mch2, mpl, false);
        }
        const uint32_t pw = swap1(warped2), pv = swap1(valid2);      // (every lane takes part: no DPP under divergence)
        const uint32_t w4 = warped2 | (pw << 16), v4 = valid2 | (pv << 16);
        if (mine && !odd) {
            if (s.warped) st32(s.warped + (int64_t)n * hw + pq, w4);
            if (MCH && s.valid) st32(s.valid + (int64_t)n * hw + pq, v4);
        }
    } else if (mine) {
        if (!t.solo) {
#pragma unroll
            for (int c = 0; c < NC; ++c) st2(dst + c * hw + pix, out[c]);
            if (s.density) st2(s.density + (int64_t)n * hw + pix, den2);
            if (s.warped) st16(s.warped + (int64_t)n * hw + pix, warped2);
            if (MCH && s.valid) st16(s.valid + (int64_t)n * hw + pix, valid2);
            if (MCH && s.mask_chan) st2(s.mask_chan + (int64_t)n * hw + pix, mch2);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) st1(dst + c * hw + pix + 1, out[c][1]);
            if (s.density) s.density[(int64_t)n * hw + pix + 1] = den2[1];
            if (s.warped) s.warped[(int64_t)n * hw + pix + 1] = (uint8_t)(warped2 >> 8);
            if (MCH && s.valid) s.valid[(int64_t)n * hw + pix + 1] = (uint8_t)(valid2 >> 8);
            if (MCH && s.mask_chan) s.mask_chan[(int64_t)n * hw + pix + 1] = mch2[1];
        }
    }
}

// data (x data_sign applied later) and mask-channel bits of one 4-pixel group of a source subtile
template <int NC, bool MCH, typename TF = float, typename SP>
__device__ __forceinline__ void sp_load_data(const SP& s, int n, int sx4, int sy, uint32_t hw, f4 (&dat)[NC], uint32_t& mc4) {
    const int w = s.w;
    const TF* __restrict__ db = reinterpret_cast<const TF*>(s.data) + n * s.data_bs;
    const float* __restrict__ dbb = s.data_b ? s.data_b + n * s.data_b_bs : nullptr;
    const int wrem = OFL_SP_WREM(s);
    const bool edge = wrem != 0 && sx4 > w - 4;                   // row-end group: last whole group, rotated
    const uint32_t px = (uint32_t)(sy * w + sx4) - (edge ? (uint32_t)(4 - wrem) : 0u);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        dat[c] = ld4(db + c * hw + px);
        if (dbb) dat[c] = dat[c] - ld4(dbb + c * hw + px);
    }
    uint32_t ma = 0x01010101u, mb = 0x01010101u;
    if (MCH) {
        if (s.chan_mask_a) ma = ld32(s.chan_mask_a + n * s.chan_mask_a_bs + px);
        if (OFL_SP_WINDOW(s)) {                                            // padded apply: chan_mask_b lives in the flow window, False outside
            mb = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bool inside;
                const uint32_t off = sp_win(s, min((int)(px % (uint32_t)w) + k, w - 1), sy, inside);
                const bool m = inside && (s.chan_mask_b ? s.chan_mask_b[n * s.chan_mask_b_bs + off] != 0 : true);
                mb |= (uint32_t)m << (8 * k);
            }
        } else if (s.chan_mask_b) mb = ld32(s.chan_mask_b + n * s.chan_mask_b_bs + px);
    }
    if (wrem != 0) {
        if (edge) {
#pragma unroll
            for (int c = 0; c < NC; ++c) dat[c] = rot4(dat[c], 4 - wrem);
            ma >>= 8 * (4 - wrem); mb >>= 8 * (4 - wrem);
        }
    }
    mc4 = nz_bytes(ma) & nz_bytes(mb);
}

// sp_load_data in two halves (see sp_issue_src): _issue only loads -- the data planes, the subtrahend planes of modes 1-2, the two
// mask-channel operands --, _done rotates a row-end group, subtracts and forms the mask-channel bits.
template <int NC> struct SpRawData { f4 dat[NC], sub[NC <= 2 ? NC : 1]; uint32_t ma, mb; int rot; };   // (only flows -- 2 channels -- have a subtrahend)
template <int NC, bool MCH, typename TF = float, typename SP>
__device__ __forceinline__ void sp_issue_data(const SP& s, int n, int sx4, int sy, uint32_t hw, SpRawData<NC>& r) {
    const int w = s.w;
    const TF* __restrict__ db = reinterpret_cast<const TF*>(s.data) + n * s.data_bs;
    const float* __restrict__ dbb = s.data_b ? s.data_b + n * s.data_b_bs : nullptr;
    const int wrem = OFL_SP_WREM(s);
    const bool edge = wrem != 0 && sx4 > w - 4;                   // row-end group: last whole group, rotated
    const uint32_t px = (uint32_t)(sy * w + sx4) - (edge ? (uint32_t)(4 - wrem) : 0u);
    r.rot = edge ? 4 - wrem : 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) r.dat[c] = ld4(db + c * hw + px);
    if (NC <= 2 && dbb) {                                          // (block-uniform)
#pragma unroll
        for (int c = 0; c < (NC <= 2 ? NC : 1); ++c) r.sub[c] = ld4(dbb + c * hw + px);
    }
    r.ma = 0x01010101u; r.mb = 0x01010101u;
    if (MCH) {
        if (s.chan_mask_a) r.ma = ld32(s.chan_mask_a + n * s.chan_mask_a_bs + px);
        if (OFL_SP_WINDOW(s)) {                                            // padded apply: chan_mask_b lives in the flow window, False outside
            r.mb = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bool inside;
                const uint32_t off = sp_win(s, min((int)(px % (uint32_t)w) + k, w - 1), sy, inside);
                const bool m = inside && (s.chan_mask_b ? s.chan_mask_b[n * s.chan_mask_b_bs + off] != 0 : true);
                r.mb |= (uint32_t)m << (8 * k);
            }
        } else if (s.chan_mask_b) r.mb = ld32(s.chan_mask_b + n * s.chan_mask_b_bs + px);
    }
}
template <int NC, typename SP>
__device__ __forceinline__ void sp_data_done(const SP& s, SpRawData<NC>& r, f4 (&dat)[NC], uint32_t& mc4) {
    const int rot = (OFL_SP_WREM(s) != 0) ? r.rot : 0;             // (block-uniform test first: widths that are multiples of 4 skip it all)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        f4 v = r.dat[c];
        if (NC <= 2 && s.data_b) v = v - r.sub[NC <= 2 ? c : 0];
        if (rot != 0) v = rot4(v, rot);
        dat[c] = v;
    }
    uint32_t ma = r.ma, mb = r.mb;
    if (rot != 0) { ma >>= 8 * rot; mb >>= 8 * rot; }
    mc4 = nz_bytes(ma) & nz_bytes(mb);
}

template <int NC, bool MCH, typename TF, typename TO, typename GP, typename SP>
__device__ __forceinline__ void sp_tile_atomics(const GP& p, const SP& s, float* acc, const uint32_t* __restrict__ lst, int nlist,
                                                const SpTile& t, int n, int r0 = 0, int r1 = kSpTH);

// The gather kernel reads its ~400 bytes of parameters from the KERNARG SEGMENT through a pointer that is made opaque at
// every phase boundary (OFL_OPAQUE_S): the compiler then fetches what a phase needs with scalar loads where it needs it.
// Left to itself it loads all ~100 dwords at the top, cannot keep them in the 102 SGPRs, spills them to VGPR lanes and
// re-reads them 16 lanes at a time: 1 000 of the 1 300 VALU instructions a wave executed per tile were v_readlane /
// v_writelane (rocprofv3 SQ_INSTS_VALU with and without the sort / sum phases: profiles/r2_splat_gather_sq_*.txt).
typedef const GatherParams __attribute__((address_space(4))) GatherParamsK;

typedef const SplatParams __attribute__((address_space(4))) SplatParamsK;
typedef const SplatParamsLean __attribute__((address_space(4))) SplatParamsLeanK;
template <int NC, bool MCH, typename TF = float, typename TO = float, bool LEAN = false>
__global__ __launch_bounds__(kSpNT2, OFL_SP_MINB) void splat_gather_kernel(const GatherParams p_by_value_unused) {
    GatherParamsK* pp = (GatherParamsK*)__builtin_amdgcn_kernarg_segment_ptr();
#define p (*pp)
    constexpr int NCH = NC + (MCH ? 1 : 0);
    constexpr uint32_t kEnd = 0xffffu, kLongCell = 0xfffeu;     // (kLongCell in a cell's first slot: phase S summed it)
    // A CELL is a unit square of the destination grid: the records whose end point has floor(x, y) = (cx, cy).  The four
    // corner classes of a destination pixel (X, Y) are the cells (X - kx, Y - ky), so one list per cell serves them all:
    // (kSpTW + 1) x (kSpTH + 1) cells per tile, the first column / row being the cells left of / above the tile.
    constexpr int kCW = kSpTW + 1, kCH = kSpTH + 1, kCells = kCW * kCH, kCellsP = (kCells + 63) / 64 * 64;
    constexpr int kCellRounds = (kCellsP + kSpNT2 - 1) / kSpNT2;
    // LDS: records (32 bytes each) | records per cell | the first 4 records of every cell, 16 bits each | head of the chain
    // of a cell's 5th, 6th ... record (later: of its sorted list) | chain links
    __shared__ __attribute__((aligned(16))) unsigned char raw[kSpQ * 32 + kCellsP * 4 + kCellsP * 8 + kCellsP * 4 + kSpQ * 2];
    __shared__ int qcount;
    // cells with more than two records (at most kSpQ / 3 of them), collected so that phase S works on them lane by lane
    constexpr int kLongQ = (kSpQ / 3 + 7) & ~7;
    __shared__ uint16_t lq[kLongQ];
    __shared__ int lqn;
    // ... and, of those, the cells with more than kSpLaneMax records: a WAVE orders and sums each of them (below)
    // cells of five to kNet records: a sorting network in one lane's registers; longer ones: a wave (3 data channels: six, the
    // eight-input network spills there and costs smooth flows 2 %; 2 channels: eight -- profiles/r4_splat_variants.txt section 8)
    constexpr int kNet = OFL_SP_NET8 == 0 ? 0 : (OFL_SP_NET8 == 1 ? (NC == 3 ? 6 : 8) : OFL_SP_NET8), kLaneMax = kNet ? kNet : kSpLaneMax;
    constexpr int kBigQ = kSpQ / (kLaneMax + 1) + 1;              // (every queued cell holds more than kLaneMax of the kSpQ records)
    __shared__ uint16_t bq[kBigQ];
    __shared__ int bqn;
    f4* rec4 = reinterpret_cast<f4*>(raw);                            // rec4[2 * i] weights, rec4[2 * i + 1] data | key
    const uint32_t* rwords = reinterpret_cast<const uint32_t*>(raw);  // key of record i: rwords[8 * i + 7]
    uint32_t* ccnt = reinterpret_cast<uint32_t*>(raw + kSpQ * 32);    // [kCellsP]: records of the cell
    uint2* slots = reinterpret_cast<uint2*>(ccnt + kCellsP);          // [kCellsP]: records 0 .. 3 of the cell (kEnd: none), raster order after phase S
    uint32_t* ohead = reinterpret_cast<uint32_t*>(slots + kCellsP);   // [kCellsP]: chain of the records beyond four / sorted list of a long cell
    uint16_t* link = reinterpret_cast<uint16_t*>(ohead + kCellsP);    // [kSpQ]: next record of the chain
#define s (*reinterpret_cast<typename std::conditional<LEAN, SplatParamsLeanK, SplatParamsK>::type*>(&pp->s))   /* (see SplatParamsLean) */
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    constexpr int kHalf = kSpNT2 / kSubLanes, kStep = 2 * kHalf;      // subtiles per half step (kSubLanes lanes each) / per step
    int tx, ty, n;
    if (!decode3(p.total, p.per_xcd, p.tiles_img, p.mi_m, p.mi_s, p.mx_m, p.mx_s, p.tiles_x, tx, ty, n)) return;
    if (p.img_over[n] != 0) return;                                   // this image takes the global-atomics path instead
    const uint32_t tile = (uint32_t)(n * (int)p.tiles_img + ty * p.tiles_x + tx);
    const uint32_t* __restrict__ lst = p.list + (int64_t)tile * kBinCap;
    // the first 32 entries of the list are fetched WITH its length (the list has a fixed address and kBinCap slots: entries
    // past the length are stale ids that are never used): one round trip for the list, one for the end points, one for the data
    const uint32_t pre[2] = {lst[tid / kSubLanes], lst[kHalf + tid / kSubLanes]};
    const int nlist = min(p.cnt[tile], kBinCap);
    OFL_OPAQUE_S(pp);
    SpTile t;
    sp_tile_setup<TF>(s, tx, ty, n, t);
    const int dx0 = t.dx0, dy0 = t.dy0, ly = t.ly, lx2 = t.lx2;
    int dflags = 0;
    const int sl = tid & (kSubLanes - 1), srow = sl >> 2, sc4 = sl & 3;
    // ---- A: walk the tile's list, 32 subtiles per step (4 source pixels per lane and half step), in three waves of loads:
    // list -> flow + weight mask of both halves -> data of the 4-pixel groups that have a pixel in the tile.  The pixels
    // whose cell lies in the tile (cell rows r0 .. r1) become LDS records and join their cell; qcount ends up as the number
    // of records wanted (more than kSpQ: not all were kept).
    // one half step: the hit test of a lane's 4 source pixels, ranks by ballot + popcount, records and cells in LDS
    // (batching the LDS traffic of both half steps -- all rank atomics, then all record stores, then all slot stores -- was
    // measured 6 % SLOWER: the longer live ranges cost more than the round trips saved)
    auto process = [&](const SpSrc& q, int sx4, int sy, const f4 (&dat)[NC], uint32_t mc4, int r0, int r1) {
        int cell[4];
        unsigned long long m[4];
        int wtot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cx = (int)__builtin_amdgcn_fmed3f(floorf(q.x[k]), -2.0f, (float)w) - dx0 + 1;
            const int cy = (int)__builtin_amdgcn_fmed3f(floorf(q.y[k]), -2.0f, (float)h) - dy0 + 1;
            const bool hit = ((q.on >> k) & 1u) != 0u && (uint32_t)cx < (uint32_t)kCW && cy >= r0 && cy <= r1;
            cell[k] = hit ? cy * kCW + cx : -1;
            m[k] = __ballot(hit);
            wtot += __popcll(m[k]);
        }
        if (wtot != 0) {                                   // wave-uniform
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&qcount, wtot);
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pos = wbase + __popcll(m[k] & below);
                wbase += __popcll(m[k]);
                if (cell[k] >= 0 && pos < kSpQ) {
                    const float xv = q.x[k], yv = q.y[k];
                    const float x0 = floorf(xv), y0 = floorf(yv);
                    const float wx0 = (x0 + 1.0f) - xv, wx1 = xv - x0, wy0 = (y0 + 1.0f) - yv, wy1 = yv - y0;   // utils.py:1110-1111
                    rec4[2 * pos] = (f4){wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};                        // utils.py:1114
                    // key: raster position of the source pixel (15 bits each, checked by ofl_splat_tiled_f32) with the
                    // mask-channel bit below it -- two records never share a position, so ordering by the whole word is raster order
                    f4 dv = {0.f, 0.f, 0.f, __uint_as_float(((((uint32_t)sy << 15) | (uint32_t)(sx4 + k)) << 1) | ((mc4 >> (8 * k)) & 1u))};
#pragma unroll
                    for (int c = 0; c < NC; ++c) dv[c] = s.data_sign * dat[c][k];
                    rec4[2 * pos + 1] = dv;
                    // the cell's first four records go to its slots (arrival order), later ones on a chain
                    const uint32_t slot = atomicAdd(&ccnt[cell[k]], 1u);
                    if (slot < 4u) reinterpret_cast<uint16_t*>(slots)[4 * cell[k] + (int)slot] = (uint16_t)pos;
                    else link[pos] = (uint16_t)atomicExch(&ohead[cell[k]], (uint32_t)pos);
                    // the cell's THIRD record makes it a cell phase S must order: exactly one thread sees slot 2, and it
                    // queues the cell there and then -- no pass over the cell counters and no barrier for it afterwards
                    // (round 4: -1.2 ... -3.3 %, profiles/r4_splat_variants.txt)
                    if (slot == 2u) lq[atomicAdd(&lqn, 1)] = (uint16_t)cell[k];
                }
            }
        }
    };
    // ---- A: walk the tile's list from entry `first`, 32 subtiles per step (4 source pixels per lane and half step): list ->
    // flow, weight mask AND data of both halves in one wave of loads (fetching the data only for the 4-pixel groups that turn
    // out to have a pixel in the tile saves a third of the bytes but costs a third dependent round trip: +10 % time -- a
    // tile's life is a chain of round trips, not a bandwidth problem).  The pixels whose cell lies in the tile (cell rows
    // r0 .. r1) become LDS records and join their cell; qcount ends up as the number of records wanted (more than kSpQ:
    // not all were kept).
    auto scan = [&](int r0, int r1, int first) {
        OFL_OPAQUE_S(pp);
        for (int base = first; base < nlist; base += kStep) {
            SpSrc q[2];
            int sx4[2], sy[2];
            bool inb[2];
            const bool two = base + kHalf < nlist;             // block-uniform: the second half step has entries
#if OFL_SP_ISSUE_FIRST
            // per half step: EVERY load first -- flow / positions, weight mask, data, mask-channel operands -- and only then the
            // end points: one round trip instead of two (see sp_issue_src).  (Both half steps' loads at once would be one
            // round trip for the rare tile with more than 64 listed subtiles too, but 70 live registers more: scratch.)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) continue;
                const int e = base + u * kHalf + tid / kSubLanes;
                const bool have = e < nlist;
                const uint32_t sub = base == 0 ? pre[u] : (have ? lst[e] : 0u);
                const uint32_t suby = fastdiv(sub, p.sx_m, p.sx_s), subx = sub - suby * (uint32_t)p.subs_x;
                sx4[u] = (int)subx * kSubW + sc4 * 4; sy[u] = (int)suby * kSubH + srow;
                inb[u] = have && (sx4[u] < w) && (sy[u] < h);
                SpRaw raw_src;
                SpRawData<NC> raw_dat;
                sp_issue_src<TF>(s, n, sx4[u], sy[u], inb[u], (uint32_t)(sy[u] * w + sx4[u]), hw, raw_src);
                if (inb[u]) sp_issue_data<NC, MCH, TF>(s, n, sx4[u], sy[u], hw, raw_dat);
                f4 dat[NC];
                uint32_t mc4 = 0x01010101u;
#pragma unroll
                for (int c = 0; c < NC; ++c) dat[c] = (f4){0.f, 0.f, 0.f, 0.f};
                sp_src_done(s, sx4[u], sy[u], inb[u], raw_src, q[u]);
                if (inb[u]) sp_data_done<NC>(s, raw_dat, dat, mc4);
                process(q[u], sx4[u], sy[u], dat, mc4, r0, r1);
            }
#else
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) { inb[1] = false; sx4[1] = sy[1] = 0; continue; }
                const int e = base + u * kHalf + tid / kSubLanes;
                const bool have = e < nlist;
                const uint32_t sub = base == 0 ? pre[u] : (have ? lst[e] : 0u);
                const uint32_t suby = fastdiv(sub, p.sx_m, p.sx_s), subx = sub - suby * (uint32_t)p.subs_x;
                sx4[u] = (int)subx * kSubW + sc4 * 4; sy[u] = (int)suby * kSubH + srow;
                const bool in = have && (sx4[u] < w) && (sy[u] < h);
                sp_load_src<TF>(s, n, sx4[u], sy[u], in, (uint32_t)(sy[u] * w + sx4[u]), hw, q[u]);
                inb[u] = in;
            }
            f4 dat[2][NC];
            uint32_t mc4[2] = {0x01010101u, 0x01010101u};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int c = 0; c < NC; ++c) dat[u][c] = (f4){0.f, 0.f, 0.f, 0.f};
                if (inb[u]) sp_load_data<NC, MCH, TF>(s, n, sx4[u], sy[u], hw, dat[u], mc4[u]);
            }
            process(q[0], sx4[0], sy[0], dat[0], mc4[0], r0, r1);
            if (two) process(q[1], sx4[1], sy[1], dat[1], mc4[1], r0, r1);
#endif
        }
    };
    using std::integral_constant;
    // bands of destination rows: 1 when every record fits the LDS; decided from the number of records the whole tile wants
    int nb = 1;
    bool over = false;
    for (int attempt = 0; attempt < 2 && !over; ++attempt) {
        const int rows = kSpTH / nb;
        bool redo = false;
        for (int band = 0; band < nb; ++band) {
            const int r0 = band * rows, r1 = r0 + rows;
#pragma unroll
            for (int i = 0; i < kCellRounds; ++i)
                if (tid + i * kSpNT2 < kCellsP) {
                    ccnt[tid + i * kSpNT2] = 0u; ohead[tid + i * kSpNT2] = kEnd;
                    slots[tid + i * kSpNT2] = make_uint2(0xffffffffu, 0xffffffffu);
                }
            if (tid == 0) {
                qcount = 0;
                lqn = 0;
                bqn = 0;
            }
            __syncthreads();
            scan(r0, r1, 0);
            __syncthreads();                                   // records, cells and the count are in place
            const int nrec = qcount;
            if (nrec > kSpQ) {                                 // block-uniform
                if (nb == 1) {                                 // a band of r rows sees about (r + 1) / 16 of the records
                    nb = 2;
                    if (nrec * (kSpTH / 2 + 1) > (kSpQ - kSpQ / 8) * kSpTH) nb = 4;
                    if (nrec * (kSpTH / 4 + 1) > (kSpQ - kSpQ / 8) * kSpTH) over = true;   // a fold: straight to the float atomics
                    redo = !over;
                } else {
                    over = true;
                }
                break;
            }
            // ---- S: every cell's records in raster order of their source pixels (ascending key) -- the order in which the
            // reference's scatter_add_ adds them within a corner class.  One or two records need nothing (a + b = b + a, and
            // the sums start from +0); three or four are sorted by a network in registers; a longer list (a compression or
            // fold of the flow) becomes a chain sorted by insertion, walked by its readers.
            bool toolong = false;
            // Cells with one or two records need nothing (a + b = b + a, sums start from +0).  The others -- a few per cent of
            // the cells, scattered over the lanes -- are first COLLECTED, then handled one per lane: put in raster order (four:
            // a network in registers; more: insertion sort of the chain) and summed there and then, per corner class, in that
            // order; the four class sums of every channel overwrite the cell's first three records and its readers add them
            // as if they were one record.  (Every reader sorting and walking such cells itself kept 60 lanes of a wave idle
            // while 4 of them worked: -20 % VALU instructions on the bench flow.)
            const int nlong = lqn;                             // (queued by the scan; the barrier after it covers the queue)
            for (int qi = tid; qi < nlong; qi += kSpNT2) {
                const int c = lq[qi];
                const uint32_t cn = ccnt[c];
                if (cn > (uint32_t)kSpLong) { toolong = true; continue; }   // the limit is on the LENGTH: the same in every run
                if (cn > (uint32_t)kLaneMax) { bq[atomicAdd(&bqn, 1)] = (uint16_t)c; continue; }   // a wave's job (below)
                const uint2 sl4 = slots[c];
                uint32_t e[4] = {sl4.x & 0xffffu, sl4.x >> 16, sl4.y & 0xffffu, sl4.y >> 16};
                // the class sums, in raster order: product rounded, then added (as sp_use does for the short cells)
                f4 sum[1 + NCH];
#pragma unroll
                for (int ch = 0; ch < 1 + NCH; ++ch) sum[ch] = (f4){0.f, 0.f, 0.f, 0.f};
                auto add = [&](const f4 wv, const f4 dv) {
                    sum[0] += wv;
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) sum[1 + ch] += wv * dv[ch];
                    if (NCH > NC) sum[1 + NC] += wv * (float)(__float_as_uint(dv[3]) & 1u);
                };
                uint32_t ia, ib, ic;
                if (cn <= 4u) {
                    // three or four records (the bulk: 2 % of the cells of the bench flow against 0.25 % longer ones): ordered by a
                    // network on (key, index) in registers and fetched by index all at once -- two dependent LDS round trips (keys,
                    // records) instead of the eight of a walk along freshly written links
                    uint32_t key[4];
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) key[j4] = e[j4] != kEnd ? rwords[8 * e[j4] + 7] : 0xffffffffu;
#define OFL_CSWAP(a_, b_) { const bool sw = key[a_] > key[b_]; const uint32_t tk = sw ? key[b_] : key[a_], te = sw ? e[b_] : e[a_]; \
                            key[b_] = sw ? key[a_] : key[b_]; e[b_] = sw ? e[a_] : e[b_]; key[a_] = tk; e[a_] = te; }
                    OFL_CSWAP(0, 1) OFL_CSWAP(2, 3) OFL_CSWAP(0, 2) OFL_CSWAP(1, 3) OFL_CSWAP(1, 2)
#undef OFL_CSWAP
                    const bool four = e[3] != kEnd;                        // (kEnd sorts last: e[3] for three records)
                    const uint32_t e3 = four ? e[3] : e[2];
                    const f4 w0 = rec4[2 * e[0]], d0 = rec4[2 * e[0] + 1], w1 = rec4[2 * e[1]], d1 = rec4[2 * e[1] + 1];
                    const f4 w2 = rec4[2 * e[2]], d2 = rec4[2 * e[2] + 1], w3 = rec4[2 * e3], d3 = rec4[2 * e3 + 1];
                    add(w0, d0); add(w1, d1); add(w2, d2);
                    if (four) add(w3, d3);
                    ia = e[0]; ib = e[1]; ic = e[2];
                } else if (kNet != 0 && cn <= (uint32_t)kNet) {
                    // five to eight records: the chain is walked ONCE (its links), the (key, index) pairs are ordered by a sorting
                    // network in registers (19 comparators for eight, 12 for six) and the records fetched by index -- ~6 dependent
                    // LDS round trips instead of the ~2 n^2 / 4 of the insertion sort below (n = 8: 32)
                    constexpr int M = kNet ? kNet : 8;
                    static_assert(M == 8 || M == 6, "sorting network size");
                    uint32_t ix[M], ky[M];
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) ix[j4] = e[j4];
                    uint32_t cur = ohead[c];
#pragma unroll
                    for (int j4 = 4; j4 < M; ++j4) {
                        const bool on = (uint32_t)j4 < cn;
                        ix[j4] = on ? cur : kEnd;
                        if (on) cur = link[cur];
                    }
#pragma unroll
                    for (int j4 = 0; j4 < M; ++j4) ky[j4] = ix[j4] != kEnd ? rwords[8 * ix[j4] + 7] : 0xffffffffu;
#define OFL_CSWAP8(a_, b_) { const bool sw = ky[a_] > ky[b_]; const uint32_t tk = sw ? ky[b_] : ky[a_], te = sw ? ix[b_] : ix[a_]; \
                             ky[b_] = sw ? ky[a_] : ky[b_]; ix[b_] = sw ? ix[a_] : ix[b_]; ky[a_] = tk; ix[a_] = te; }
                    if (M == 8) {
                        OFL_CSWAP8(0, 1) OFL_CSWAP8(2, 3) OFL_CSWAP8(4, 5) OFL_CSWAP8(M - 2, M - 1)
                        OFL_CSWAP8(0, 2) OFL_CSWAP8(1, 3) OFL_CSWAP8(4, M - 2) OFL_CSWAP8(5, M - 1)
                        OFL_CSWAP8(1, 2) OFL_CSWAP8(5, M - 2) OFL_CSWAP8(0, 4) OFL_CSWAP8(3, M - 1)
                        OFL_CSWAP8(1, 5) OFL_CSWAP8(2, M - 2)
                        OFL_CSWAP8(1, 4) OFL_CSWAP8(3, M - 2)
                        OFL_CSWAP8(2, 4) OFL_CSWAP8(3, 5)
                        OFL_CSWAP8(3, 4)
                    } else {
                        OFL_CSWAP8(0, 1) OFL_CSWAP8(2, 3) OFL_CSWAP8(4, 5)
                        OFL_CSWAP8(0, 2) OFL_CSWAP8(3, 5) OFL_CSWAP8(1, 4)
                        OFL_CSWAP8(0, 1) OFL_CSWAP8(2, 3) OFL_CSWAP8(4, 5)
                        OFL_CSWAP8(1, 2) OFL_CSWAP8(3, 4)
                        OFL_CSWAP8(2, 3)
                    }
#undef OFL_CSWAP8
                    // (unused slots carry the largest key: they sort last; five records at least are real)
#pragma unroll
                    for (int r = 0; r < M; ++r) {
                        if (r < 5 || (uint32_t)r < cn) add(rec4[2 * ix[r]], rec4[2 * ix[r] + 1]);
                        if (r == 3) asm volatile("" ::: "memory");     // (two batches of loads)
                    }
                    ia = ix[0]; ib = ix[1]; ic = ix[2];
                } else {
                    uint32_t cur = ohead[c];
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) { link[e[j4]] = (uint16_t)cur; cur = e[j4]; }
                    uint32_t sorted = kEnd;
                    while (cur != kEnd) {
                        const uint32_t nxt = link[cur], k = rwords[8 * cur + 7];
                        if (sorted == kEnd || rwords[8 * sorted + 7] > k) {
                            link[cur] = (uint16_t)sorted; sorted = cur;
                        } else {
                            uint32_t q = sorted, qn = link[q];
                            while (qn != kEnd && rwords[8 * qn + 7] < k) { q = qn; qn = link[q]; }
                            link[cur] = (uint16_t)qn; link[q] = (uint16_t)cur;
                        }
                        cur = nxt;
                    }
                    ia = sorted; ib = link[ia]; ic = link[ib];
                    for (uint32_t i = sorted; i != kEnd; i = link[i]) add(rec4[2 * i], rec4[2 * i + 1]);
                }
#pragma unroll
                for (int ch = 0; ch < 1 + NCH; ++ch) {
                    if (ch < 2) rec4[2 * ia + ch] = sum[ch];
                    else if (ch < 4) rec4[2 * ib + (ch - 2)] = sum[ch];
                    else rec4[2 * ic] = sum[ch];
                }
                slots[c] = make_uint2(kLongCell | (ia << 16), ib | (ic << 16));
            }
            if (nlong != 0) {                                  // (block-uniform: a tile without such cells needs no barrier here)
                over = __syncthreads_or((int)toolong) != 0;
                const int nbig = bqn;
                if (!over && nbig != 0) {
                    // Cells with more than kSpLaneMax records (a compression of the flow): one lane ordering such a cell by
                    // insertion walks a chain in LDS -- two dependent round trips per comparison, ~n^2 / 4 of them: 0.1 ms for 64
                    // records, and the kernel waited for the block that had drawn it (profiles/r4_splat_variants.txt, 4).  A WAVE
                    // does it instead, one lane per record: the chain is walked once, every lane counts the keys below its own
                    // (its rank in raster order), a permute puts the record indices in that order, and lane (class, channel)
                    // adds its corner-class sum over the records in that order -- product rounded, then added, exactly what the
                    // single lane did.  ~120 cycles per record instead of ~64 per record SQUARED.
                    for (int b = tid >> 6; b < nbig; b += kSpNT2 / 64) sp_order_big_cell<NC, NCH>(raw, ccnt, slots, ohead, link, bq[b], lane);
                    __syncthreads();
                }
            }
            if (over) break;
            // ---- C: the sums of this thread's 2 destination pixels (if their row is in the band), finalize.
            // The pair reads 3 x 2 cells; every record of a cell is fetched once and added to each corner-class sum it
            // belongs to (sp_use).
            const bool mine = t.inimg && ly >= r0 && ly < r1;
            float tot[2][1 + NCH];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int c = 0; c < 1 + NCH; ++c) tot[k][c] = 0.0f;
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
                    const uint2 sl2 = slots[c];
                    const uint32_t e0 = sl2.x & 0xffffu, e1 = sl2.x >> 16, e2 = sl2.y & 0xffffu, e3 = sl2.y >> 16;
                    if (e0 == kEnd) return;
                    if (e0 != kLongCell) {
                        sp_use<NC, NCH, DC, KY>(rec4, e0, a);
                        if (e1 != kEnd) sp_use<NC, NCH, DC, KY>(rec4, e1, a);
                    } else {                                               // phase S left the cell's class sums
                        sp_use_presum<NCH, DC, KY>(rec4, e1, e2, e3, a);
                    }
                };
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
            OFL_OPAQUE_S(pp);
            sp_finalize<NC, MCH, TF, TO>(s, t, tot, mine, dflags);
            if (nb > 1) __syncthreads();                      // the next band re-uses the LDS
        }
        if (!redo) break;
        dflags = 0;                                           // (nothing was finalized before the first band overflowed)
        __syncthreads();
    }
    if (over) {
        // a fold (more records than four bands hold, or > 64 sources in one cell)
        // redone at once by this block with LDS float atomics (the records are dead: their LDS is the accumulator); whatever
        // the first bands stored is overwritten, and the tile's flag word comes from here
        if (tid == 0) atomicAdd(&p.stats[1], 1);
        OFL_OPAQUE_S(pp);
        sp_tile_atomics<NC, MCH, TF, TO>(p, s, reinterpret_cast<float*>(raw), lst, nlist, t, n);
        return;
    }
    if (NC == 2 && s.dst_flags) {                             // (every thread of the block gets here)
        dflags = wave_or_flags(dflags);
        block_flag_or(&s.dst_flags[n], dflags);              // one access per block on the image's word (see block_flag_or)
    }
#undef s
#undef p
}

// A tile the gather kernel cannot sum in order (a heavy fold of the flow): LDS float atomics over the tile's list (plane 0
// density, then the data channels; the mask channel accumulates the INVALID weight, so that an all-valid pixel is exactly 1
// in any order).  Tolerance instead of bit-exactness for these tiles; masks stay exact.  `acc` = (1 + NCH) * 512 floats of
// LDS no thread of the block still reads; every thread of the block calls this.  [r0, r1): the rows of the tile that are summed and
// stored (round 6: the second launch folds a BAND of a tile, not the whole tile).
template <int NC, bool MCH, typename TF, typename TO, typename GP, typename SP>
__device__ __forceinline__ void sp_tile_atomics(const GP& p, const SP& s, float* acc, const uint32_t* __restrict__ lst, int nlist,
                                                const SpTile& t, int n, int r0, int r1) {
    constexpr int kPx = kSpTW * kSpTH, NCH = NC + (MCH ? 1 : 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    const float wmax = (float)(w - 1), hmax = (float)(h - 1);
    __syncthreads();
    for (int i = tid; i < (1 + NCH) * kPx; i += kSpNT2) acc[i] = 0.0f;
    __syncthreads();
    const int sl = tid & (kSubLanes - 1), srow = sl >> 2, sc4 = sl & 3;
    for (int base = 0; base < nlist; base += kSpNT2 / kSubLanes) {
        const int e = base + tid / kSubLanes;
        const bool have = e < nlist;
        const uint32_t sub = have ? lst[e] : 0u;
        const uint32_t suby = fastdiv(sub, p.sx_m, p.sx_s), subx = sub - suby * (uint32_t)p.subs_x;
        const int sx4 = (int)subx * kSubW + sc4 * 4, sy = (int)suby * kSubH + srow;
        const bool in = have && (sx4 < w) && (sy < h);
        SpSrc q;
        sp_load_src<TF>(s, n, sx4, sy, in, (uint32_t)(sy * w + sx4), hw, q);
        f4 dat[NC];
        uint32_t mc4 = 0x01010101u;
#pragma unroll
        for (int c = 0; c < NC; ++c) dat[c] = (f4){0.f, 0.f, 0.f, 0.f};
        if (q.on != 0u) sp_load_data<NC, MCH, TF>(s, n, sx4, sy, hw, dat, mc4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!((q.on >> k) & 1u)) continue;
            float wx[2], wy[2]; int ix[2], iy[2];
            sp_corners(q.x[k], q.y[k], wmax, hmax, t.dx0, t.dy0, wx, wy, ix, iy);
            const bool invalid = MCH ? (((mc4 >> (8 * k)) & 1u) == 0u) : false;
#pragma unroll
            for (int ky = 0; ky < 2; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) {
                    const float wgt = wy[ky] * wx[kx];
                    const int xl = ix[kx], yl = iy[ky];
                    if (wgt == 0.0f || (uint32_t)xl >= (uint32_t)kSpTW || (uint32_t)(yl - r0) >= (uint32_t)(r1 - r0)) continue;
                    const int d = yl * kSpTW + xl;
                    atomicAdd(&acc[d], wgt);
#pragma unroll
                    for (int c = 0; c < NC; ++c) atomicAdd(&acc[(1 + c) * kPx + d], wgt * (s.data_sign * dat[c][k]));
                    if (MCH && invalid) atomicAdd(&acc[(1 + NC) * kPx + d], wgt);
                }
            }
        }
    }
    __syncthreads();
    int dflags = 0;
    float tot[2][1 + NCH];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int d = t.ly * kSpTW + max(t.lx2 + k, 0);
#pragma unroll
        for (int c = 0; c < 1 + NCH; ++c) tot[k][c] = acc[c * kPx + d];
        if (MCH) tot[k][1 + NC] = tot[k][0] - tot[k][1 + NC];      // density - invalid weight
    }
    sp_finalize<NC, MCH, TF, TO>(s, t, tot, t.inimg && t.ly >= r0 && t.ly < r1, dflags);
    if (NC == 2 && s.dst_flags) {
        dflags = wave_or_flags(dflags);
        if (lane == 0) flag_or(&s.dst_flags[n], dflags);
    }
}

// ------------------------------------------------------------------------------------------------
// forward splat, gather formulation ON A DIET (round 6): splat_gather_kernel's algorithm with THREE blocks = 24 waves per CU
// ------------------------------------------------------------------------------------------------
// profiles/r5_splat_scan.txt: the round-5 kernel is latency-bound -- one block per CU instead of two costs x 1.6 -- and what kept
// it at two blocks was 81 KB of LDS and 119-125 VGPRs per 512-thread block.  Same phases (A scan, S order, C sum), same arithmetic,
// same order of every addition; what changed is what a tile keeps in LDS and in registers:
//   * a RECORD is 16 + 0 / 4 / 8 bytes (1 / 2 / 3 data channels) instead of 32: part A = (fx, fy, d0, d1 -- one channel: the key
//     in d1's place), part B = the key (2 channels) or (d2, key) (3 channels), in two arrays.  fx = x - x0 and fy = y - y0 are the
//     reference's weights wt_x1 / wt_y1 themselves (utils.py:1110-1111); the other two, x1 - x and y1 - y with x1 = x0 + 1, are
//     re-formed by the reader as 1 - fx / 1 - fy: for a cell inside the frame (x0 >= 0) x - x0 is EXACT in fp32 (a multiple of
//     ulp(x) below 1), so (x0 + 1) - x and 1 - fx are the correct rounding of the same real number -- the same float.  (A cell left
//     of / above the frame, x0 = -1, only ever serves its x1 / y1 corner: fx is stored as the reference rounds it, 1 - fx is never
//     read.)  The four products wy * wx are formed by the reader, once per use, as the reference's outer product forms them;
//   * a CELL is ONE 32-bit word: records so far << 16 | newest record, every record on a chain through `link` (16 bits per record).
//     A record joins by compare-and-swap on that word (first try: "the cell is empty" -- true for three records in four); the
//     returned word is the previous head AND the arrival number, so the third arrival still queues the cell for phase S.  4 bytes per
//     cell instead of 16;
//   * phase S leaves cells of 3 or 4 records as a chain IN RASTER ORDER (their readers add the records one by one: there is no room
//     for class sums in 3 x 20 bytes) and pre-sums longer cells into part A of their first 1 + NCH records (marker 0xffff in the
//     word's count; the chain links those records);
//   * capacity: 2 048 records with 1 or 2 data channels (flows: fewer banded tiles than the 1 792 of round 5), 1 792 with 3.
// LDS per block: 45.2 KB (1 channel) / 52.4 KB (2) / 52.9 KB (3) -> 3 blocks per CU; __launch_bounds__(512, 6): <= 80 VGPRs.
#ifndef OFL_SP_MINB2
#define OFL_SP_MINB2 6
#endif
#ifndef OFL_SP_Q2
#define OFL_SP_Q2 2048
#endif
#ifndef OFL_SP_REDOGRID
#define OFL_SP_REDOGRID 512
#endif
#ifndef OFL_SP2_UNITTIME
#define OFL_SP2_UNITTIME 0
#endif
#ifndef OFL_SP2_SKIP
#define OFL_SP2_SKIP 0         // a measuring aid (tools/prof_phase_insts.sh; WRONG results): the first launch without 1 phase S, 2 phase C, 4 finalize, 8 the records
#endif
#ifndef OFL_SP_ALLVALID
#define OFL_SP_ALLVALID 1      // phase C leaves the mask channel out where every scanned pixel is valid (0: always summed)
#endif
#ifndef OFL_SP_PLAN
#define OFL_SP_PLAN 1          // an overflowing tile's bands are planned from the exact record counts of its cell rows (0: 2 or 4 equal bands)
#endif
#ifndef OFL_SP_BIGBLOCK
#define OFL_SP_BIGBLOCK 1      // the second launch orders a band's big cells by the whole block (0: a wave per cell, as the first launch)
#endif
#ifndef OFL_SP_Q3
#define OFL_SP_Q3 1792
#endif
template <int NC> struct SpLay {
    static constexpr int kQ = NC == 3 ? OFL_SP_Q3 : OFL_SP_Q2;           // records a tile holds at a time
    static constexpr int kB = NC == 3 ? 8 : (NC == 2 ? 4 : 0);           // bytes of part B
    static constexpr int kCW = kSpTW + 1, kCH = kSpTH + 1, kCells = kCW * kCH, kCellsP = (kCells + 63) / 64 * 64;
    static constexpr int kRawBytes = (16 + kB + 2) * kQ + 4 * kCellsP;   // part A | part B | link | cell words
    static constexpr int kNet = NC == 3 ? 6 : 8;                         // cells up to kNet records: a sorting network in one lane's registers
    static constexpr int kLongQ = (kQ / 3 + 7) & ~7, kBigQ = kQ / (kNet + 1) + 1;
};
constexpr uint32_t kSp2End = 0xffffu, kSp2Empty = 0x0000ffffu, kSp2Sum = 0xffffu;   // end of a chain | an empty cell's word | count field of a pre-summed cell

template <int NC>
__device__ __forceinline__ uint32_t sp2_key(const f4* recA, const uint32_t* recB, uint32_t i) {
    if (NC == 1) return reinterpret_cast<const uint32_t*>(recA)[4 * i + 3];
    if (NC == 2) return recB[i];
    return recB[2 * i + 1];
}

// data channels (mask channel last) of record i; av = its part A
template <int NC, int NCH>
__device__ __forceinline__ void sp2_data(const f4& av, const uint32_t* recB, uint32_t i, float (&d)[NCH > 0 ? NCH : 1]) {
    uint32_t key = 0u;
    d[0] = av[2];
    if (NC == 1) key = __float_as_uint(av[3]);
    if (NC >= 2) d[1] = av[3];
    if (NC == 2 && NCH > NC) key = recB[i];
    if (NC == 3) { const uint2 b = *reinterpret_cast<const uint2*>(recB + 2 * i); d[2] = __uint_as_float(b.x); key = b.y; }
    if (NCH > NC) d[NC] = (float)(key & 1u);                              // the mask channel rides in the key
}

// The sums of one (pixel, x-corner) class in phase C: the density, then NV values -- the data channels, the mask channel last when
// it is summed -- held in PAIRS: a product and an addition of two channels are one v_pk_mul_f32 / v_pk_add_f32 each (written with
// scalars the compiler paired the density with channel 0 and channel 2 with channel 1, and moved registers about to do it).
template <int NV> struct SpAcc {
    float den;
    f2 pr[NV / 2 > 0 ? NV / 2 : 1];
    float last;                                                          // (an odd NV: its last value)
    __device__ __forceinline__ void clear() {
        den = 0.0f; last = 0.0f;
#pragma unroll
        for (int q = 0; q < (NV / 2 > 0 ? NV / 2 : 1); ++q) pr[q] = (f2){0.0f, 0.0f};
    }
    __device__ __forceinline__ float get(int c) const { return c == 0 ? den : ((c - 1) < 2 * (NV / 2) ? pr[(c - 1) / 2][(c - 1) & 1] : last); }
    __device__ __forceinline__ void add(int c, float v) {
        if (c == 0) den += v;
        else if ((c - 1) < 2 * (NV / 2)) pr[(c - 1) / 2][(c - 1) & 1] += v;
        else last += v;
    }
};

// One record of the cell whose column is DC (-1, 0, +1) cells from the pair's middle cell and whose row serves corner row KY, added
// to the sums of the destination pixels that read it (see sp_use): weight = wy[KY] * wx[kx], rounded, then product with the data
// rounded, then added.  NV = NC (+ 1: the mask channel is summed -- a tile all of whose records are valid leaves it out: its sums
// are the density's, the same additions of the same weights).
template <int NC, int NV, int DC, int KY>
__device__ __forceinline__ void sp2_use(const f4* recA, const uint32_t* recB, uint32_t i, SpAcc<NV> (&a)[2][2]) {
    const f4 av = recA[i];
    float d[NV > 0 ? NV : 1];
    sp2_data<NC, NV>(av, recB, i, d);
    const float wy = KY == 0 ? 1.0f - av[1] : av[1];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int kx = k - DC;
        if (kx < 0 || kx > 1) continue;
        const float wx = kx == 0 ? 1.0f - av[0] : av[0];
        const float wgt = wy * wx;
        a[k][kx].den += wgt;
#pragma unroll
        for (int q = 0; q < NV / 2; ++q) a[k][kx].pr[q] += (f2){wgt, wgt} * (f2){d[2 * q], d[2 * q + 1]};
        if (NV & 1) a[k][kx].last += wgt * d[NV - 1];
    }
}

// a pre-summed cell: part A of the records on its chain holds one f4 (the four corner classes) per channel
template <int NV, int NCH, int DC, int KY>
__device__ __forceinline__ void sp2_use_presum(const f4* recA, const uint16_t* link, uint32_t i, SpAcc<NV> (&a)[2][2]) {
#pragma unroll
    for (int c = 0; c < 1 + NV; ++c) {
        const f4 pv = recA[i];
        if (c < NCH) i = link[i];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int kx = k - DC;
            if (kx < 0 || kx > 1) continue;
            a[k][kx].add(c, pv[KY * 2 + kx]);
        }
    }
}

// A cell with more than kNet (and at most 64) records, ordered and summed by the 64 lanes of one wave (see sp_order_big_cell).
template <int NC, int NCH>
__device__ __attribute__((noinline)) void sp2_order_big_cell(f4* recA, uint32_t* recB, uint16_t* link, uint32_t* cellw, int c, int lane) {
    const uint32_t cw = cellw[c];
    const int cn = (int)__builtin_amdgcn_readfirstlane((int)(cw >> 16));
    // 1. the cell's record indices, one per lane (the chain is walked once; every lane reads the same word)
    uint32_t idx = 0, cur = cw & 0xffffu;
    for (int j = 0; j < cn; ++j) {
        const uint32_t cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
        idx = lane == j ? cs : idx;
        cur = (uint32_t)link[cs];
    }
    const bool mine = lane < cn;
    // 2. rank of every record = number of keys below its own (keys are source positions: all different)
    const uint32_t key = mine ? sp2_key<NC>(recA, recB, idx) : 0xffffffffu;
    uint32_t rank = 0;
    for (int t = 0; t < cn; ++t) rank += ((uint32_t)__builtin_amdgcn_readlane((int)key, t) < key) ? 1u : 0u;
    // 3. lane r gets the index of the r-th record in raster order
    uint32_t sorted = 0;
    if (mine) sorted = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)idx);
    // 4. lane (class k, channel ch) adds its sum over the records in that order: weight (product of two, rounded) times data, rounded, added.
    // FOUR records are fetched before the first is used (two broadcast LDS reads each: part A, part B): fetched one at a time the loop was a
    // chain of dependent LDS round trips, ~150 cycles per record -- the second launch 402 -> 309 us at sigma 12 (profiles/r6_splat_diet.txt)
    const int k = lane & 3, ch = lane >> 2;
    const bool acc_on = ch < 1 + NCH;
    const bool is_mask = NCH > NC && ch == 1 + NC;
    float acc = 0.0f;
    for (int r0 = 0; r0 < cn; r0 += 4) {
        f4 av[4];
        uint32_t b0[4], b1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t ir = (uint32_t)__builtin_amdgcn_readlane((int)sorted, min(r0 + j, cn - 1));
            av[j] = recA[ir];
            b0[j] = NC >= 2 ? recB[(NC == 3 ? 2 : 1) * ir] : 0u;
            b1[j] = NC == 3 ? recB[2 * ir + 1] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float fx = av[j][0], fy = av[j][1];
            const uint32_t kj = NC == 1 ? __float_as_uint(av[j][3]) : (NC == 2 ? b0[j] : b1[j]);
            const float wv = ((k & 2) ? fy : 1.0f - fy) * ((k & 1) ? fx : 1.0f - fx);
            float dv = 1.0f;                                              // (density: the weight itself -- w * 1 is exact)
            if (is_mask) dv = (float)(kj & 1u);
            else if (ch == 1) dv = av[j][2];
            else if (ch == 2) dv = av[j][3];
            else if (ch == 3 && NC == 3) dv = __uint_as_float(b0[j]);
            const float nxt = acc + wv * dv;
            acc = r0 + j < cn ? nxt : acc;
        }
    }
    // 5. the class sums where the cell's readers expect them: part A of the first 1 + NCH records (raster order), on a chain
    const uint32_t mych = (uint32_t)__builtin_amdgcn_ds_bpermute((acc_on ? ch : 0) << 2, (int)sorted);          // (every lane takes part)
    const uint32_t nxch = (uint32_t)__builtin_amdgcn_ds_bpermute((ch < NCH ? ch + 1 : 0) << 2, (int)sorted);
    if (acc_on) {
        reinterpret_cast<float*>(recA)[4 * mych + k] = acc;
        if (k == 0) link[mych] = (uint16_t)(ch < NCH ? nxch : kSp2End);
    }
    if (lane == 0) cellw[c] = (kSp2Sum << 16) | (uint32_t)__builtin_amdgcn_readlane((int)sorted, 0);
}

// The second launch's way with big cells: ALL of a band's big cells at once, by the whole block.  One wave per cell (above) walks a
// chain of dependent LDS reads per record and uses 4 x (1 + NCH) of its 64 lanes for the sums -- ~280 cycles per record, and a band
// inside a compression of the flow holds a hundred such cells: half of the second launch (profiles/r6_splat_diet.txt, 5).  Here:
//   a. one LANE per cell walks its chain into a contiguous segment of `seg` (every chain of the band side by side);
//   b. one THREAD per record ranks it among its cell's keys (independent LDS reads) and puts it at its place in `srt`;
//   c. a GROUP of 4 x (1 + NCH) lanes per cell -- 3 to 5 cells per wave -- adds the class sums in that order, four records fetched
//      ahead, and leaves them where phase C expects them (as sp2_order_big_cell's step 5).
// Same sums, same order.  12.9 KB of LDS more than the first launch's kernel has room for: the second launch only.
template <int NC, int NCH>
__device__ __forceinline__ void sp2_big_cells_block(f4* recA, uint32_t* recB, uint16_t* link, uint32_t* cellw, const uint16_t* bq, int nbig,
                                                    uint16_t* seg, uint16_t* srt, uint16_t* segb, uint32_t* bqinfo, int* segtop, int tid) {
    const int lane = tid & 63;
    for (int b = tid; b < nbig; b += kSpNT2) {
        const uint32_t cw = cellw[bq[b]];
        const int cn = (int)(cw >> 16);
        const int base = atomicAdd(segtop, cn);
        bqinfo[b] = (uint32_t)base | ((uint32_t)cn << 16);
        uint32_t cur = cw & 0xffffu;
        for (int j = 0; j < cn; ++j) { seg[base + j] = (uint16_t)cur; segb[base + j] = (uint16_t)b; cur = link[cur]; }
    }
    __syncthreads();
    const int nseg = __builtin_amdgcn_readfirstlane(*segtop);
    for (int i = tid; i < nseg; i += kSpNT2) {
        const uint32_t info = bqinfo[segb[i]];
        const int base = (int)(info & 0xffffu), cn = (int)(info >> 16);
        const uint32_t my = seg[i], mykey = sp2_key<NC>(recA, recB, my);
        int rank = 0;
        for (int t = 0; t < cn; t += 4) {
            uint32_t kk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) kk[j] = sp2_key<NC>(recA, recB, seg[base + min(t + j, cn - 1)]);
#pragma unroll
            for (int j = 0; j < 4; ++j) rank += (t + j < cn && kk[j] < mykey) ? 1 : 0;
        }
        srt[base + rank] = (uint16_t)my;
    }
    __syncthreads();
    constexpr int LG = 4 * (1 + NCH), G = 64 / LG;
    const int g = lane / LG, wi = lane - g * LG, k = wi & 3, ch = wi >> 2;
    const bool is_mask = NCH > NC && ch == 1 + NC;
    for (int bw = (tid >> 6) * G; bw < nbig; bw += (kSpNT2 / 64) * G) {     // (wave-uniform)
        const int b = bw + g;
        const bool on = g < G && b < nbig;
        const uint32_t info = on ? bqinfo[b] : 0u;
        const int base = (int)(info & 0xffffu), cn = (int)(info >> 16);
        float acc = 0.0f;
        for (int r0 = 0; r0 < cn; r0 += 4) {
            f4 av[4];
            uint32_t b0[4], b1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t ir = srt[base + min(r0 + j, cn - 1)];
                av[j] = recA[ir];
                b0[j] = NC >= 2 ? recB[(NC == 3 ? 2 : 1) * ir] : 0u;
                b1[j] = NC == 3 ? recB[2 * ir + 1] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float fx = av[j][0], fy = av[j][1];
                const uint32_t kj = NC == 1 ? __float_as_uint(av[j][3]) : (NC == 2 ? b0[j] : b1[j]);
                const float wv = ((k & 2) ? fy : 1.0f - fy) * ((k & 1) ? fx : 1.0f - fx);
                float dv = 1.0f;
                if (is_mask) dv = (float)(kj & 1u);
                else if (ch == 1) dv = av[j][2];
                else if (ch == 2) dv = av[j][3];
                else if (ch == 3 && NC == 3) dv = __uint_as_float(b0[j]);
                const float nxt = acc + wv * dv;
                acc = r0 + j < cn ? nxt : acc;
            }
        }
        // (every lane of the wave has left its loop: the records the sums overwrite have been read)
        if (on) {
            const uint32_t mych = srt[base + ch];
            reinterpret_cast<float*>(recA)[4 * mych + k] = acc;
            if (k == 0) link[mych] = (uint16_t)(ch < NCH ? (uint32_t)srt[base + ch + 1] : kSp2End);
            if (wi == 0) cellw[bq[b]] = (kSp2Sum << 16) | (uint32_t)srt[base];
        }
    }
    __syncthreads();
}

// REDO = false: the kernel every tile runs on -- ONE scan, no bands: a tile whose records overflow the LDS, or that holds a cell of
// more than 64 records, puts its id on the pass's redo list and leaves.  REDO = true: a second, small launch that walks that list
// with the whole repertoire (2 or 4 bands of rows, the float-atomics fold).  Splitting the two takes every loop away from round
// the hot path's phases: with the band loops in place the compiler hoisted each per-thread invariant of the later phases (tile
// geometry, output offsets) to the top of the kernel and, at 80 registers, spilled ten of them there (profiles/r6_splat_diet.txt).
template <int NC, bool MCH, typename TF = float, typename TO = float, bool LEAN = false, bool REDO = false>
__global__ __launch_bounds__(kSpNT2, REDO ? 4 : OFL_SP_MINB2) void splat_gather2_kernel(const GatherParams p_by_value_unused) {
    GatherParamsK* pp = (GatherParamsK*)__builtin_amdgcn_kernarg_segment_ptr();
#define p (*pp)
    constexpr int NCH = NC + (MCH ? 1 : 0);
    using L = SpLay<NC>;
    constexpr int kQ = L::kQ, kCW = L::kCW, kCellsP = L::kCellsP, kNet = L::kNet;
    constexpr int kCellRounds = (kCellsP + kSpNT2 - 1) / kSpNT2;
    constexpr uint32_t kEnd = kSp2End;
    __shared__ __attribute__((aligned(16))) unsigned char raw[L::kRawBytes];
    __shared__ int qcount, lqn, bqn;
    __shared__ int tile_inv;                                          // some scanned pixel is masked out of the mask channel (else phase C leaves that channel out)
    __shared__ uint16_t lq[L::kLongQ];                                // cells with more than two records: phase S works on them lane by lane
    __shared__ __attribute__((aligned(4))) uint16_t bq[L::kBigQ];     // ... of those, the cells with more than kNet records: a wave each
    static_assert(sizeof(uint16_t) * L::kBigQ >= sizeof(int) * (L::kCH + 1), "the per-row counts of an overflowing tile (and the mask of its too-long rows) live in the big-cell queue");
    int* rowover = reinterpret_cast<int*>(bq);                        // (first launch) records per cell row of a tile that overflows: it never orders its cells
    constexpr int kSegN = (REDO && OFL_SP_BIGBLOCK) ? kQ : 1;         // (the second launch: big cells by the whole block -- sp2_big_cells_block)
    __shared__ uint16_t seg[kSegN], srt[kSegN], segb[kSegN];
    __shared__ uint32_t bqinfo[(REDO && OFL_SP_BIGBLOCK) ? L::kBigQ : 1];
    __shared__ int segtop;
    static_assert((size_t)(1 + NCH) * kSpTW * kSpTH * sizeof(float) <= sizeof(raw), "the fold path's accumulators live in the record area");
    f4* recA = reinterpret_cast<f4*>(raw);
    uint32_t* recB = reinterpret_cast<uint32_t*>(raw + 16 * kQ);
    uint16_t* link = reinterpret_cast<uint16_t*>(raw + (16 + L::kB) * kQ);
    uint32_t* cellw = reinterpret_cast<uint32_t*>(raw + (18 + L::kB) * kQ);
#define s (*reinterpret_cast<typename std::conditional<LEAN, SplatParamsLeanK, SplatParamsK>::type*>(&pp->s))   /* (see SplatParamsLean) */
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int kHalf = kSpNT2 / kSubLanes, kStep = 2 * kHalf;      // subtiles per half step (kSubLanes lanes each) / per step
    int redo_n = 0;
    if (REDO) {
        redo_n = __builtin_amdgcn_readfirstlane(p.redo_cnt[0]);
        if (tid == 0 && blockIdx.x == 0 && redo_n != 0) atomicAdd(&p.stats[3], redo_n);     // statistics: tiles that took the second launch
    }
    // (REDO: the units of the list are handed out one at a time through a counter -- they differ by an order of magnitude in work, and
    // with a fixed stride the launch lasted as long as its unluckiest block)
    __shared__ int next_unit;
    int ri = (int)blockIdx.x;
    if (REDO && ri >= redo_n) return;
    do {                                                              // (REDO: units of the list until it is empty; else once)
#if OFL_SP2_UNITTIME
    const uint64_t unit_t0 = wall_clock64();
#endif
    int tx, ty, n, ua = 0, ub = kSpTH, ufold = 0;                    // (ufold: the first launch knows this band cannot be summed in order)
    uint32_t tile;
    if (REDO) {
        const uint2 unit = reinterpret_cast<const uint2*>(p.redo_list)[ri];
        // (readfirstlane: these come from a flat load, which the compiler takes for divergent -- and a loop it takes for divergent
        // may not contain the "+s" pins of the kernarg pointer: "illegal VGPR to SGPR copy")
        tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit.x);
        ua = __builtin_amdgcn_readfirstlane((int)(unit.y & 0xffu)); ub = __builtin_amdgcn_readfirstlane((int)((unit.y >> 8) & 0xffu));
        ufold = __builtin_amdgcn_readfirstlane((int)((unit.y >> 16) & 1u));
        const uint32_t nn = fastdiv(tile, p.mi_m, p.mi_s), rem = tile - nn * p.tiles_img, yy = fastdiv(rem, p.mx_m, p.mx_s);
        n = (int)nn; ty = (int)yy; tx = (int)(rem - yy * (uint32_t)p.tiles_x);
    } else {
        if (!decode3(p.total, p.per_xcd, p.tiles_img, p.mi_m, p.mi_s, p.mx_m, p.mx_s, p.tiles_x, tx, ty, n)) return;
        if (p.img_over[n] != 0) return;                               // this image takes the global-atomics path instead
        tile = (uint32_t)(n * (int)p.tiles_img + ty * p.tiles_x + tx);
    }
    const int w = s.w, h = s.h;
    const uint32_t hw = (uint32_t)(h * w);
    const uint32_t* __restrict__ lst = p.list + (int64_t)tile * kBinCap;
    // the first entries of the list are fetched WITH its length (fixed address, kBinCap slots: entries past the length are stale
    // ids that are never used)
    const uint32_t pre[2] = {lst[tid / kSubLanes], lst[kHalf + tid / kSubLanes]};
    const int nlist = min(p.cnt[tile], kBinCap);
    const int dx0 = tx * kSpTW, dy0 = ty * kSpTH;
    // one half step: the hit test of a lane's 4 source pixels, ranks by ballot + popcount, records and cells in LDS
    auto process = [&](const SpSrc& q, int sx4, int sy, const f4 (&dat)[NC], uint32_t mc4, int r0, int r1) {
        int cell[4];
        unsigned long long m[4];
        int wtot = 0;
        if (MCH && OFL_SP_ALLVALID && __ballot(mc4 != 0x01010101u) != 0ull) { if (lane == 0) tile_inv = 1; }   // (wave-uniform; rare)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cx = (int)__builtin_amdgcn_fmed3f(floorf(q.x[k]), -2.0f, (float)w) - dx0 + 1;
            const int cy = (int)__builtin_amdgcn_fmed3f(floorf(q.y[k]), -2.0f, (float)h) - dy0 + 1;
            const bool hit = ((q.on >> k) & 1u) != 0u && (uint32_t)cx < (uint32_t)kCW && (REDO ? (cy >= r0 && cy <= r1) : ((uint32_t)cy < (uint32_t)L::kCH));
            cell[k] = hit ? cy * kCW + cx : -1;
            m[k] = __ballot(hit);
            wtot += __popcll(m[k]);
        }
        if (wtot != 0 && !(!REDO && (OFL_SP2_SKIP & 8))) {   // wave-uniform
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&qcount, wtot);
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t pos = (uint32_t)(wbase + __popcll(m[k] & below));
                wbase += __popcll(m[k]);
                if (cell[k] >= 0 && pos < (uint32_t)kQ) {
                    const float xv = q.x[k], yv = q.y[k];
                    const float fx = xv - floorf(xv), fy = yv - floorf(yv);                                      // utils.py:1110-1111 (wt_x1, wt_y1)
                    // key: raster position of the source pixel (15 bits each, checked by ofl_splat_tiled_f32) with the
                    // mask-channel bit below it -- two records never share a position, so ordering by the whole word is raster order
                    const uint32_t key = ((((uint32_t)sy << 15) | (uint32_t)(sx4 + k)) << 1) | ((mc4 >> (8 * k)) & 1u);
                    f4 av = {fx, fy, s.data_sign * dat[0][k], NC >= 2 ? s.data_sign * dat[NC >= 2 ? 1 : 0][k] : __uint_as_float(key)};
                    recA[pos] = av;
                    if (NC == 2) recB[pos] = key;
                    if (NC == 3) *reinterpret_cast<uint2*>(recB + 2 * pos) = make_uint2(__float_as_uint(s.data_sign * dat[NC - 1][k]), key);
                    // the record joins its cell's chain: compare-and-swap of (count << 16 | head), first try "empty"
                    uint32_t expect = kSp2Empty;
                    for (;;) {
                        const uint32_t got = atomicCAS(&cellw[cell[k]], expect, ((expect & 0xffff0000u) + 0x10000u) | pos);
                        if (got == expect) break;
                        expect = got;
                    }
                    link[pos] = (uint16_t)expect;
                    // the cell's THIRD record makes it a cell phase S must order: exactly one thread sees two before its own
                    // (the bound: in a tile that overflows the count also counts records that did not fit -- below -- and "two before
                    // my own" no longer means three records in LDS; such a tile never reads the queue)
                    if ((expect >> 16) == 2u) { const int qi = atomicAdd(&lqn, 1); if (qi < L::kLongQ) lq[qi] = (uint16_t)cell[k]; }
                }
            }
            // (first launch) a tile that overflows: the records that do not fit are still COUNTED in their cells' words -- the tile's
            // bands are planned from the exact counts of its cell rows (below).  wbase is the wave's end position by now: wave-uniform
            if (!REDO && OFL_SP_PLAN && wbase > kQ) {
                int wb = wbase - wtot;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pos = wb + __popcll(m[k] & below);
                    wb += __popcll(m[k]);
                    if (cell[k] >= 0 && pos >= kQ) atomicAdd(&cellw[cell[k]], 0x10000u);
                }
            }
        }
    };
    // ---- A: walk the tile's list (see splat_gather_kernel): per half step EVERY load first, then the end points
    auto scan = [&](int r0, int r1, bool first) {
        const int sl = tid & (kSubLanes - 1), srow = sl >> 2, sc4 = sl & 3;
        OFL_OPAQUE_S(pp);
        for (int base = 0; base < nlist; base += kStep) {
            const bool two = base + kHalf < nlist;             // block-uniform: the second half step has entries
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) continue;
                const int e = base + u * kHalf + tid / kSubLanes;
                const bool have = e < nlist;
                const uint32_t sub = (base == 0 && first) ? pre[u] : (have ? lst[e] : 0u);
                const uint32_t suby = fastdiv(sub, p.sx_m, p.sx_s), subx = sub - suby * (uint32_t)p.subs_x;
                const int sx4 = (int)subx * kSubW + sc4 * 4, sy = (int)suby * kSubH + srow;
                const bool inb = have && (sx4 < w) && (sy < h);
                SpRaw raw_src;
                SpRawData<NC> raw_dat;
                sp_issue_src<TF>(s, n, sx4, sy, inb, (uint32_t)(sy * w + sx4), hw, raw_src);
                if (inb) sp_issue_data<NC, MCH, TF>(s, n, sx4, sy, hw, raw_dat);
                SpSrc q;
                f4 dat[NC];
                uint32_t mc4 = 0x01010101u;
#pragma unroll
                for (int c = 0; c < NC; ++c) dat[c] = (f4){0.f, 0.f, 0.f, 0.f};
                sp_src_done(s, sx4, sy, inb, raw_src, q);
                if (inb) sp_data_done<NC>(s, raw_dat, dat, mc4);
                process(q, sx4, sy, dat, mc4, r0, r1);
            }
        }
    };
    // ---- S: cells with more than two records in raster order of their source pixels (ascending key) -- the order in which the
    // reference's scatter_add_ adds them within a corner class (one or two need nothing: a + b = b + a, sums start from +0).
    // Three or four: the chain is re-linked in raster order.  Five to kNet: ordered by a network in one lane's registers and summed
    // per corner class there and then; the class sums overwrite part A of the first 1 + NCH records.  More: a wave.  Returns true
    // when a cell holds more than kSpLong records (block-uniform).
    auto order = [&]() -> bool {
        bool toolong = false;
        const int nlong = __builtin_amdgcn_readfirstlane(lqn);   // (queued by the scan; the barrier after it covers the queue.  readfirstlane: block-uniform values read from LDS must be uniform to the compiler too)
        if (nlong == 0) return false;                      // (block-uniform: a tile without such cells needs no barrier here)
        for (int qi = tid; qi < nlong; qi += kSpNT2) {
            const int c = lq[qi];
            const uint32_t cw = cellw[c];
            const uint32_t cn = cw >> 16;
            if (cn > (uint32_t)kSpLong) { toolong = true; continue; }   // the limit is on the LENGTH: the same in every run
            if (cn > (uint32_t)kNet) { bq[atomicAdd(&bqn, 1)] = (uint16_t)c; continue; }   // a wave's job (below)
            if (cn <= 4u) {
                uint32_t e[4], key[4];
                e[0] = cw & 0xffffu; e[1] = link[e[0]]; e[2] = link[e[1]]; e[3] = cn == 4u ? (uint32_t)link[e[2]] : kEnd;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) key[j4] = e[j4] != kEnd ? sp2_key<NC>(recA, recB, e[j4]) : 0xffffffffu;
#define OFL_CSWAP(a_, b_) { const bool sw = key[a_] > key[b_]; const uint32_t tk = sw ? key[b_] : key[a_], te = sw ? e[b_] : e[a_]; \
                        key[b_] = sw ? key[a_] : key[b_]; e[b_] = sw ? e[a_] : e[b_]; key[a_] = tk; e[a_] = te; }
                OFL_CSWAP(0, 1) OFL_CSWAP(2, 3) OFL_CSWAP(0, 2) OFL_CSWAP(1, 3) OFL_CSWAP(1, 2)
#undef OFL_CSWAP
                link[e[0]] = (uint16_t)e[1]; link[e[1]] = (uint16_t)e[2]; link[e[2]] = (uint16_t)e[3];   // (kEnd sorts last: e[3] for three records)
                if (e[3] != kEnd) link[e[3]] = (uint16_t)kEnd;
                cellw[c] = (cn << 16) | e[0];
            } else {
                constexpr int M = kNet;
                static_assert(M == 8 || M == 6, "sorting network size");
                uint32_t ix[M], ky[M];
                uint32_t cur = cw & 0xffffu;
#pragma unroll
                for (int j4 = 0; j4 < M; ++j4) {
                    const bool on = (uint32_t)j4 < cn;
                    ix[j4] = on ? cur : kEnd;
                    if (on) cur = link[cur];
                }
#pragma unroll
                for (int j4 = 0; j4 < M; ++j4) ky[j4] = ix[j4] != kEnd ? sp2_key<NC>(recA, recB, ix[j4]) : 0xffffffffu;
#define OFL_CSWAP8(a_, b_) { const bool sw = ky[a_] > ky[b_]; const uint32_t tk = sw ? ky[b_] : ky[a_], te = sw ? ix[b_] : ix[a_]; \
                         ky[b_] = sw ? ky[a_] : ky[b_]; ix[b_] = sw ? ix[a_] : ix[b_]; ky[a_] = tk; ix[a_] = te; }
                if (M == 8) {
                    OFL_CSWAP8(0, 1) OFL_CSWAP8(2, 3) OFL_CSWAP8(4, 5) OFL_CSWAP8(M - 2, M - 1)
                    OFL_CSWAP8(0, 2) OFL_CSWAP8(1, 3) OFL_CSWAP8(4, M - 2) OFL_CSWAP8(5, M - 1)
                    OFL_CSWAP8(1, 2) OFL_CSWAP8(5, M - 2) OFL_CSWAP8(0, 4) OFL_CSWAP8(3, M - 1)
                    OFL_CSWAP8(1, 5) OFL_CSWAP8(2, M - 2)
                    OFL_CSWAP8(1, 4) OFL_CSWAP8(3, M - 2)
                    OFL_CSWAP8(2, 4) OFL_CSWAP8(3, 5)
                    OFL_CSWAP8(3, 4)
                } else {
                    OFL_CSWAP8(0, 1) OFL_CSWAP8(2, 3) OFL_CSWAP8(4, 5)
                    OFL_CSWAP8(0, 2) OFL_CSWAP8(3, 5) OFL_CSWAP8(1, 4)
                    OFL_CSWAP8(0, 1) OFL_CSWAP8(2, 3) OFL_CSWAP8(4, 5)
                    OFL_CSWAP8(1, 2) OFL_CSWAP8(3, 4)
                    OFL_CSWAP8(2, 3)
                }
#undef OFL_CSWAP8
                // the class sums, in raster order: weight (a product, rounded), product with the data rounded, then added (as sp2_use)
                f4 sum[1 + NCH];
#pragma unroll
                for (int ch = 0; ch < 1 + NCH; ++ch) sum[ch] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < M; ++r) {
                    if (r < 5 || (uint32_t)r < cn) {       // (unused slots carry the largest key: they sort last; five records at least are real)
                        const f4 av = recA[ix[r]];
                        float d[NCH > 0 ? NCH : 1];
                        sp2_data<NC, NCH>(av, recB, ix[r], d);
                        const float wy0 = 1.0f - av[1], wx0 = 1.0f - av[0];
                        const f4 wv = {wy0 * wx0, wy0 * av[0], av[1] * wx0, av[1] * av[0]};
                        sum[0] += wv;
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) sum[1 + ch] += wv * d[ch];
                    }
                }
#pragma unroll
                for (int ch = 0; ch < 1 + NCH; ++ch) {
                    recA[ix[ch]] = sum[ch];
                    link[ix[ch]] = (uint16_t)(ch < NCH ? ix[ch < NCH ? ch + 1 : 0] : kEnd);
                }
                cellw[c] = (kSp2Sum << 16) | ix[0];
            }
        }
        if (__builtin_amdgcn_readfirstlane(__syncthreads_or((int)toolong)) != 0) return true;
        const int nbig = __builtin_amdgcn_readfirstlane(bqn);
        if (nbig != 0) {
            if (REDO && OFL_SP_BIGBLOCK) {
                sp2_big_cells_block<NC, NCH>(recA, recB, link, cellw, bq, nbig, seg, srt, segb, bqinfo, &segtop, tid);
            } else {
                for (int b = tid >> 6; b < nbig; b += kSpNT2 / 64) sp2_order_big_cell<NC, NCH>(recA, recB, link, cellw, bq[b], lane);
                __syncthreads();
            }
        }
        return false;
    };
    // ---- C: the sums of this thread's 2 destination pixels.  The pair reads 3 x 2 cells; every record of a cell is fetched once
    // and added to each corner-class sum it belongs to (sp2_use).
    using std::integral_constant;
    auto sums_nv = [&](auto nv_, int ly, int lx2, float (&tot)[2][1 + NCH]) {
        constexpr int NV = decltype(nv_)::value;
        SpAcc<NV> a[2][2];                                        // [pixel of the pair][x-corner]: the corner row in hand
        auto clear = [&]() {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) a[k][kx].clear();
        };
        const int cm = max(lx2 + 1, 0);                            // cell column of pixel 1's x-corner 1 = of pixel 0's x-corner 0
        auto cell = [&](auto dc_, auto ky_) {
            constexpr int DC = decltype(dc_)::value, KY = decltype(ky_)::value;
            const int c = (ly + 1 - KY) * kCW + max(cm + DC, 0);   // (solo: pixel 0's own cells do not exist; it is never stored)
            const uint32_t cw = cellw[c];
            uint32_t cur = cw & 0xffffu;
            if (cur == kEnd) return;
            if ((cw >> 16) != kSp2Sum) {
#pragma unroll 1
                do { sp2_use<NC, NV, DC, KY>(recA, recB, cur, a); cur = link[cur]; } while (cur != kEnd);
            } else {                                               // phase S left the cell's class sums
                sp2_use_presum<NV, NCH, DC, KY>(recA, link, cur, a);
            }
        };
        clear();                                                   // corner row 0: classes 0, 1
        cell(integral_constant<int, -1>{}, integral_constant<int, 0>{});
        cell(integral_constant<int, 0>{}, integral_constant<int, 0>{});
        cell(integral_constant<int, 1>{}, integral_constant<int, 0>{});
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 1 + NV; ++c) tot[k][c] = a[k][0].get(c) + a[k][1].get(c);
        clear();                                                   // corner row 1: classes 2, 3
        cell(integral_constant<int, -1>{}, integral_constant<int, 1>{});
        cell(integral_constant<int, 0>{}, integral_constant<int, 1>{});
        cell(integral_constant<int, 1>{}, integral_constant<int, 1>{});
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int c = 0; c < 1 + NV; ++c) tot[k][c] = (tot[k][c] + a[k][0].get(c)) + a[k][1].get(c);   // ((c0 + c1) + c2) + c3
            if (NV < NCH) tot[k][1 + NC] = tot[k][0];              // every record valid: the mask channel's sums are the density's
        }
    };
    // (block-uniform) a tile -- or band -- none of whose scanned pixels is masked out sums no mask channel: it would repeat the density's
    // additions, weight by weight (OFL_SP_ALLVALID 0: always summed)
    auto sums = [&](int ly, int lx2, float (&tot)[2][1 + NCH]) {
        if (MCH && OFL_SP_ALLVALID && __builtin_amdgcn_readfirstlane(tile_inv) == 0) sums_nv(integral_constant<int, NC>{}, ly, lx2, tot);
        else sums_nv(integral_constant<int, NCH>{}, ly, lx2, tot);
    };
    auto zero_cells = [&]() {
#pragma unroll
        for (int i = 0; i < kCellRounds; ++i)
            if (tid + i * kSpNT2 < kCellsP) cellw[tid + i * kSpNT2] = kSp2Empty;
        if (tid == 0) { qcount = 0; lqn = 0; bqn = 0; segtop = 0; tile_inv = 0; }
    };
    int dflags = 0;
    if (!REDO) {
        // ================= the hot path: one scan, order, sum, finalize -- straight-line, nothing kept across the phases that the
        // phases themselves do not need =================
        // the un-occlude fill candidates of this thread's pair need the tile's own flow: loaded here, beside the list head (one
        // round trip for both), kept as two bits
        SpTile t0;
        sp_tile_setup<TF>(s, tx, ty, n, t0);
        uint32_t fillbits = (t0.fill_ok[0] ? 1u : 0u) | (t0.fill_ok[1] ? 2u : 0u);
        zero_cells();
        __syncthreads();
        scan(0, kSpTH, true);
        __syncthreads();                                   // records, cells and the count are in place
        // block-uniform: more records than the LDS holds, or a cell of more than kSpLong records -> the second launch, as 2 or 4
        // independent BANDS of rows (a band of r rows sees about (r + 1) / 17 of the records) that other blocks sum side by side
        const int nrec = __builtin_amdgcn_readfirstlane(qcount);
        int nb = 0;
        if (OFL_SP_PLAN && nrec > kQ) {
            // The bands are PLANNED: the records of every cell row are counted (the cells' words count the records that did not fit
            // too) and the rows are cut into the fewest bands of at most kQ records each (a band of rows
            // [a, b) holds cell rows a .. b), as evenly as 64 candidate limits allow: no band of the second launch overflows and scans
            // again, and none is much heavier than its siblings (the second launch lasts as long as its heaviest unit:
            // profiles/r6_splat_diet.txt, 7).
            if (tid <= L::kCH) rowover[tid] = 0;                  // (17 counts, then the mask of cell rows that hold a cell of more than kSpLong records)
            __syncthreads();
            for (int i = tid; i < L::kCells; i += kSpNT2) {
                const uint32_t cn = cellw[i] >> 16;
                const uint32_t row = (uint32_t)i / (uint32_t)kCW;
                if (cn != 0u) atomicAdd(&rowover[row], (int)cn);
                if (cn > (uint32_t)kSpLong) atomicOr(&rowover[L::kCH], 1 << row);
            }
            __syncthreads();
            if (tid < 64) {
                // rows that cannot be summed in order whatever the band: one of the row's two cell rows holds too long a cell, or
                // the two hold more than kQ records.  They become bands of their own (at most 4 rows), marked: the second launch
                // folds them without scanning first
                const uint32_t longrows = (uint32_t)rowover[L::kCH];
                const bool own = tid < kSpTH && rowover[tid & (kSpTH - 1)] + rowover[(tid & (kSpTH - 1)) + 1] > kQ;
                const uint32_t foldrows = ((longrows | (longrows >> 1)) & ((1u << kSpTH) - 1u)) | (uint32_t)__ballot(own);
                const int lim = (kQ * (tid + 1)) >> 6;            // lane 63: kQ itself
                uint32_t ends = 0u;
                int a = 0, cnt = 0;
                while (a < kSpTH) {
                    int b = a + 1;
                    if ((foldrows >> a) & 1u) {
                        while (b < kSpTH && ((foldrows >> b) & 1u) && b - a < 4) ++b;
                    } else {
                        int sum = rowover[a] + rowover[b];
                        while (b < kSpTH && !((foldrows >> b) & 1u) && sum + rowover[b + 1] <= lim) { ++b; sum += rowover[b]; }
                    }
                    ends |= 1u << b; ++cnt; a = b;
                }
                const int nbmin = __builtin_amdgcn_readlane(cnt, 63);
                const int win = __ffsll((unsigned long long)__ballot(cnt == nbmin)) - 1;
                ends = (uint32_t)__builtin_amdgcn_readlane((int)ends, win);
                int base = 0;
                if (tid == 0) base = atomicAdd(&p.redo_cnt[0], nbmin);
                base = __builtin_amdgcn_readfirstlane(base);
                if (tid < nbmin) {
                    int b0 = 0, b1 = 0;
                    for (int j = 0; j <= tid; ++j) { b0 = b1; b1 = __ffs((int)ends) - 1; ends &= ends - 1u; }
                    reinterpret_cast<uint2*>(p.redo_list)[base + tid] = make_uint2(tile, (uint32_t)b0 | ((uint32_t)b1 << 8) | (((foldrows >> b0) & 1u) << 16));
                }
            }
            return;
        }
        if (nrec > kQ) nb = (nrec * (kSpTH / 2 + 1) > (kQ - kQ / 8) * kSpTH) ? 4 : 2;
        else if (!(OFL_SP2_SKIP & 1) && order()) nb = 4;
        if (nb != 0) {
            if (tid < nb) {
                int base = 0;
                if (tid == 0) base = atomicAdd(&p.redo_cnt[0], nb);
                base = __builtin_amdgcn_readfirstlane(base);
                const int rows = kSpTH / nb;
                reinterpret_cast<uint2*>(p.redo_list)[base + tid] = make_uint2(tile, (uint32_t)(tid * rows) | ((uint32_t)(tid * rows + rows) << 8));
            }
            return;
        }
        asm volatile("" : "+v"(fillbits));
        SpTile t;                                          // (geometry re-formed here: a few integer operations instead of registers held across the scan)
        t.n = n; t.dx0 = dx0; t.dy0 = dy0;
        {
            const int lx = tid % (kSpTW / 2);
            t.ly = tid / (kSpTW / 2);
            const int x2 = min(dx0 + lx * 2, w - 2), y = dy0 + t.ly;
            t.lx2 = x2 - dx0;
            t.solo = t.lx2 < 0;
            t.inimg = (dx0 + lx * 2 < w) && (y < h);
            t.pix = (uint32_t)(min(y, h - 1) * w + x2);
            t.wide = dx0 + kSpTW <= w;
            t.fill_ok[0] = (fillbits & 1u) != 0u; t.fill_ok[1] = (fillbits & 2u) != 0u;
        }
        float tot[2][1 + NCH];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 1 + NCH; ++c) tot[k][c] = 0.0f;
        if (t.inimg && !(OFL_SP2_SKIP & 2)) sums(t.ly, t.lx2, tot);
        OFL_OPAQUE_S(pp);
        if (!(OFL_SP2_SKIP & 4)) sp_finalize<NC, MCH, TF, TO>(s, t, tot, t.inimg, dflags);
    } else {
        // ================= the second launch: one BAND of rows [ua, ub) of a tile per list entry.  A band whose records still do not
        // fit (or that holds a cell of more than kSpLong records) is halved, down to 4 rows; a 4-row band that does not fit is
        // summed with LDS float atomics (tolerance instead of bit-exactness for those rows; counted in stats[1]) =================
        SpTile t;
        sp_tile_setup<TF>(s, tx, ty, n, t);
        const int ly = t.ly, lx2 = t.lx2;
        bool first = true;
        int a0 = ua, b0 = ub;
        while (a0 < ub) {                                      // (block-uniform)
            bool fits = false;
            if (ufold == 0) {
                zero_cells();
                __syncthreads();
                scan(a0, b0, first);
                first = false;
                __syncthreads();
                fits = __builtin_amdgcn_readfirstlane(qcount) <= kQ;
                if (fits) fits = !order();
                if (!fits && b0 - a0 > 4) {                    // halve the band and try again
                    b0 = a0 + (b0 - a0) / 2;
                    __syncthreads();
                    continue;
                }
            }
            if (fits) {
                const bool mine = t.inimg && ly >= a0 && ly < b0;
                float tot[2][1 + NCH];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int c = 0; c < 1 + NCH; ++c) tot[k][c] = 0.0f;
                if (mine) sums(ly, lx2, tot);
                OFL_OPAQUE_S(pp);
                sp_finalize<NC, MCH, TF, TO>(s, t, tot, mine, dflags);
            } else {
                if (tid == 0) atomicAdd(&p.stats[1], 1);
                OFL_OPAQUE_S(pp);
                sp_tile_atomics<NC, MCH, TF, TO>(p, s, reinterpret_cast<float*>(raw), lst, nlist, t, n, a0, b0);   // (posts its rows' flag word itself)
            }
            __syncthreads();                                   // the next band re-uses the LDS
            const int span = b0 - a0;
            a0 = b0; b0 = min(a0 + span, ub);
        }
    }
    if (NC == 2 && s.dst_flags) {                             // (every thread of the block gets here)
        dflags = wave_or_flags(dflags);
        block_flag_or(&s.dst_flags[n], dflags);              // one access per block on the image's word (see block_flag_or)
    }
    if (!REDO) break;
#if OFL_SP2_UNITTIME     /* a measuring aid (tools/redo_unit_times.py; never in a default build): when each unit started and how long it took, in the list itself */
    if (tid == 0) {
        const uint64_t t1 = wall_clock64();
        p.redo_list[2 * ri] = tile | ((uint32_t)((unit_t0 >> 3) & 0xffffu) << 16);
        p.redo_list[2 * ri + 1] = (uint32_t)ua | ((uint32_t)ub << 8) | ((uint32_t)min((t1 - unit_t0) >> 2, (uint64_t)0xffffu) << 16);
    }
#endif
    if (tid == 0) next_unit = (int)gridDim.x + atomicAdd(&p.redo_cnt[1], 1);
    __syncthreads();                                          // (also: the next unit re-uses the LDS)
    ri = __builtin_amdgcn_readfirstlane(next_unit);
    } while (ri < redo_n);
#undef s
#undef p
}

// ------------------------------------------------------------------------------------------------
// flow flags (ofl_flow_flags_f32)
// ------------------------------------------------------------------------------------------------
// NT: non-temporal loads -- a batch larger than the last-level cache streams 13 % faster past it (B=64 1080p: 5.8 -> 6.65
// TB/s); a small one (B=8: 150 MB) is better off cached
// HOST (ofl_flow_flags_host): the words go straight to host-visible memory.  Every wave waits for its own atomicOr on the
// device words (agent-scope atomics execute at the memory side, beyond the per-XCD L2s: once `vmcnt` has drained they are
// performed for every XCD); the block then takes a ticket from one of kFlagShards arrival counters (its linear id modulo
// kFlagShards: ~500 equal blocks end together, and returning atomics on ONE word serialise at ~12 ns each), the last arriver
// of a shard takes a ticket from the top counter, and the block that draws the last top ticket reads the words back with
// returning atomics (exchange with 0: words and counters are left zeroed for the next call) and stores each of them to the
// host buffer as ONE 8-byte {serial number, word} pair with a system-scope (write-through) store -- the host polls the
// pairs: no copy launch, no event, no memset, no ordering between stores.  Only atomics and write-through stores carry
// the payload, so no L2 write-back (`buffer_wbl2`, a release fence) is involved -- one such fence per block cost 30 us per
// launch (MI355X_MICROARCH.md, Workgroup dispatch ... & inter-workgroup visibility: atomics on both sides; 8-byte granules).
constexpr int kFlagShards = 32;
struct FlagsHost { int32_t* counters; unsigned long long* host; int32_t serial, total_blocks, n; };   // counters: [kFlagShards] + [1] top

__device__ __forceinline__ void flags_publish(int32_t* __restrict__ flags, const FlagsHost& fh) {
    __shared__ int last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's atomicOr (if any) has been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        const int id = (int)(blockIdx.y * gridDim.x + blockIdx.x), shard = id % kFlagShards;
        const int in_shard = (fh.total_blocks - shard + kFlagShards - 1) / kFlagShards;          // blocks whose id = shard (mod kFlagShards)
        const int shards = fh.total_blocks < kFlagShards ? fh.total_blocks : kFlagShards;        // (non-empty ones)
        int l = 0;
        if (__hip_atomic_fetch_add(fh.counters + shard, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1)
            l = __hip_atomic_fetch_add(fh.counters + kFlagShards, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1;
        last = l;
    }
    __syncthreads();
    if (!last) return;                                            // (block-uniform)
    // Every block has arrived (the tickets say so), so nobody touches the counters any more: THIS block zeroes all of them,
    // with returning atomics whose results it waits for, BEFORE the first pair goes out -- the host may hand the work words
    // to the next call (another stream) the moment it has seen the pairs, and a reset still in flight then would eat that
    // call's first arrivals.
    if (threadIdx.x <= kFlagShards) {
        const int old = __hip_atomic_exchange(fh.counters + threadIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" :: "v"(old));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the counters are zero at the memory side
    __syncthreads();
    for (int i = threadIdx.x; i < fh.n; i += blockDim.x) {        // (a word's store depends on its exchange: zero before it is published)
        const uint32_t v = (uint32_t)__hip_atomic_exchange(&flags[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&fh.host[i], ((unsigned long long)v << 32) | (uint32_t)fh.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <bool VEC, bool NT = false, bool HOST = false>   // VEC: 4 pixels per thread and step (16-byte flow loads, one mask dword); needs hw % 4 == 0 and aligned planes
__global__ __launch_bounds__(256) void flow_flags_kernel(const float* __restrict__ flow, int64_t flow_bs,
                                                         const uint8_t* __restrict__ mask, int64_t mask_bs,
                                                         int32_t* __restrict__ flags, int64_t hw, const FlagsHost fh) {
    const int n = blockIdx.y;
    const float* fu = flow + n * flow_bs;
    const uint8_t* mk = mask ? mask + n * mask_bs : nullptr;
    int f = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC) {
        const int64_t hw4 = hw >> 2;
        constexpr int R = 4;
        // The mask only decides the two MASKED bits, and a word only ever gains bits: once some lane of the wave holds
        // "beyond the threshold under a True mask" (which implies "non-zero under a True mask"), no further mask byte can
        // change the wave's word -- the rest of its share is read without the mask (wave-uniform; 9 -> 8 B/px for all but
        // the first step of a flow that moves anywhere under its mask: the common case; finiteness still sees every vector).
        bool skip_mask = false;
        for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < hw4; i0 += R * stride) {
            f4 a[R], b[R]; uint32_t m4[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {                         // R independent 16-byte groups in flight
                const int64_t i = i0 + r * stride;
                if (i < hw4) {
                    if (NT) { a[r] = ld4nt(fu + 4 * i); b[r] = ld4nt(fu + hw + 4 * i); }
                    else { a[r] = reinterpret_cast<const f4*>(fu)[i]; b[r] = reinterpret_cast<const f4*>(fu + hw)[i]; }
                    m4[r] = (mk && !skip_mask) ? reinterpret_cast<const uint32_t*>(mk)[i] : 0x01010101u;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (i0 + r * stride < hw4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) f |= flag_bits(a[r][k], b[r][k], ((m4[r] >> (8 * k)) & 0xffu) != 0u);
                }
            }
            if (mk && !skip_mask) skip_mask = __any((f & OFL_FLAG_NZ_THR_MASKED) != 0) != 0;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += stride)
            f |= flag_bits(fu[i], fu[hw + i], mk ? (mk[i] != 0) : true);
    }
    f = wave_or_flags(f);
    block_flag_or(&flags[n], f);
    if (HOST) flags_publish(flags, fh);
}

// fp16-stored flows (BASELINE config 5): the reference's entry conversion `vecs.float()` (utils.py:95,118) and the flag
// reduction in ONE pass -- 4 pixels per thread and step: 8-byte fp16 loads, 16-byte fp32 stores, one mask dword.
// Needs hw % 4 == 0 and 8 / 16-byte aligned planes (else the binding converts with torch and calls ofl_flow_flags_f32).
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
template <bool HOST = false>
__global__ __launch_bounds__(256) void flow_f16_kernel(const _Float16* __restrict__ src, int64_t src_bs,
                                                       const uint8_t* __restrict__ mask, int64_t mask_bs,
                                                       float* __restrict__ dst, int32_t* __restrict__ flags, int64_t hw, const FlagsHost fh) {
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
                if (dst) { reinterpret_cast<f4*>(du)[i] = u; reinterpret_cast<f4*>(du + hw)[i] = v; }
#pragma unroll
                for (int k = 0; k < 4; ++k) f |= flag_bits(u[k], v[k], ((m4[r] >> (8 * k)) & 0xffu) != 0u);
            }
        }
    }
    f = wave_or_flags(f);
    block_flag_or(&flags[n], f);
    if (HOST) flags_publish(flags, fh);
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

// the promises of WarpParamsLean hold for this launch
inline bool warp_is_lean(const WarpParams& p) {
    return OFL_WARP_LEAN && (p.w & 3) == 0 && p.round_mode == OFL_ROUND_NONE && p.flow_flags == nullptr && p.shear != 0;
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

#if defined(OFL_SPLAT_TU)
}  // namespace

// This translation unit (ofl_splat_gather.hip) provides the gather splat's diet kernel (splat_gather2_kernel) and nothing else.
// gp == nullptr: no launch -- the kernel's resources (info[0] blocks of kSpNT2 threads resident per CU, [1] static LDS bytes,
// [2] VGPRs, [3] scratch bytes per thread) as the runtime reports them
template <int NC, bool MCH, typename TF, typename TO>
static int splat_launch_diet(const GatherParams* gp, unsigned grid, hipStream_t st, int extra_lds, int lean, int32_t* info) {
    const bool ln = gp ? (NC >= 2 && OFL_SP_LEAN && splat_is_lean(gp->s)) : (NC >= 2 && lean != 0);
    const void* fn = ln ? (const void*)splat_gather2_kernel<NC, MCH, TF, TO, (NC >= 2)> : (const void*)splat_gather2_kernel<NC, MCH, TF, TO>;
    constexpr unsigned kRedoGrid = OFL_SP_REDOGRID;          // the second launch walks the redo list with this many blocks (2 per CU: 65 KB of LDS each)
    if (!gp) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, fn);
        if (e != hipSuccess) return (int)e;
        int blocks = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, kSpNT2, (size_t)extra_lds);
        if (e != hipSuccess) return (int)e;
        info[0] = blocks; info[1] = (int32_t)fa.sharedSizeBytes; info[2] = fa.numRegs; info[3] = (int32_t)fa.localSizeBytes;
        return OFL_OK;
    }
    if (ln) {
        OFL_KLAUNCH((splat_gather2_kernel<NC, MCH, TF, TO, (NC >= 2)>), dim3(grid), dim3(kSpNT2), (size_t)extra_lds, st, *gp);
        OFL_KLAUNCH((splat_gather2_kernel<NC, MCH, TF, TO, (NC >= 2), true>), dim3(grid < kRedoGrid ? grid : kRedoGrid), dim3(kSpNT2), 0, st, *gp);
    } else {
        OFL_KLAUNCH((splat_gather2_kernel<NC, MCH, TF, TO>), dim3(grid), dim3(kSpNT2), (size_t)extra_lds, st, *gp);
        OFL_KLAUNCH((splat_gather2_kernel<NC, MCH, TF, TO, false, true>), dim3(grid < kRedoGrid ? grid : kRedoGrid), dim3(kSpNT2), 0, st, *gp);
    }
    return (int)hipGetLastError();
}
static int splat_diet_dispatch(const GatherParams* gp, int nc, int mch, int elem, unsigned grid, hipStream_t st, int extra_lds, int lean, int32_t* info) {
    if (elem != 0) {
        if (nc != 2) return OFL_E_UNSUPPORTED;
        if (mch) return elem == 2 ? splat_launch_diet<2, true, _Float16, _Float16>(gp, grid, st, extra_lds, lean, info) : splat_launch_diet<2, true, _Float16, float>(gp, grid, st, extra_lds, lean, info);
        return elem == 2 ? splat_launch_diet<2, false, _Float16, _Float16>(gp, grid, st, extra_lds, lean, info) : splat_launch_diet<2, false, _Float16, float>(gp, grid, st, extra_lds, lean, info);
    }
    switch (nc * 2 + (mch ? 1 : 0)) {
        case 2: return splat_launch_diet<1, false, float, float>(gp, grid, st, extra_lds, lean, info);
        case 3: return splat_launch_diet<1, true, float, float>(gp, grid, st, extra_lds, lean, info);
        case 4: return splat_launch_diet<2, false, float, float>(gp, grid, st, extra_lds, lean, info);
        case 5: return splat_launch_diet<2, true, float, float>(gp, grid, st, extra_lds, lean, info);
        case 6: return splat_launch_diet<3, false, float, float>(gp, grid, st, extra_lds, lean, info);
        case 7: return splat_launch_diet<3, true, float, float>(gp, grid, st, extra_lds, lean, info);
    }
    return OFL_E_UNSUPPORTED;
}
int ofl_splat_launch_gather_diet(const void* params, int nc, int mch, int elem, unsigned grid, void* stream, int extra_lds) {
    return splat_diet_dispatch(static_cast<const GatherParams*>(params), nc, mch, elem, grid, (hipStream_t)stream, extra_lds, 0, nullptr);
}
extern "C" __attribute__((visibility("default"))) int ofl_splat_gather_info(int32_t channels, int32_t with_mask_chan, int32_t elem, int32_t lean,
                                                                            int32_t extra_lds, int32_t* info4) {
    if (!info4) return OFL_E_NULL;
    if (channels < 1 || channels > 3 || elem < 0 || elem > 2 || extra_lds < 0) return OFL_E_ARG;
    return splat_diet_dispatch(nullptr, channels, with_mask_chan ? 1 : 0, elem, 0, nullptr, extra_lds, lean, info4);
}
#elif defined(OFL_WIDE_TU)
}  // namespace

// This translation unit (ofl_warp_wide.hip) provides ONE thing: the column kernel of a large plain warp on 64 x 16 tiles.
#if OFL_ROWS_STAMPS
extern "C" __attribute__((visibility("default"))) int ofl_debug_rows_stamps(unsigned long long* out, int clear) {
    hipDeviceSynchronize();
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rows_stamp), sizeof(unsigned long long) * 16);
    if (clear) { unsigned long long z[16] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stamp), z, sizeof(z)); }
    return 0;
}
#endif
int ofl_wide_launch_column(const void* params, int nc, int valid, int add, int rows, void* stream) {
    WarpParams q = *static_cast<const WarpParams*>(params);
    q.lds_bytes = kLdsBytes;
    const unsigned g = warp_geometry(q, kLdsTWQ * 4, kLdsT * kLdsTH);
    hipStream_t st = (hipStream_t)stream;
    constexpr int TT = kLdsT > 2 ? kLdsT : 3;
    const bool lean = warp_is_lean(q);
    constexpr int RT = OFL_ROWS_T;                                // tiles per column of the row-table kernel
    if (OFL_WARP_ROWS && rows && lean && nc == 2 && (add || q.src_b || q.dst_flags)) {      // the flow-level variants with per-row extents
        const unsigned gr = RT == TT ? g : warp_geometry(q, kLdsTWQ * 4, RT * kLdsTH);
        const int am = !add ? 0 : (q.add_is_flow ? 1 : 2);
#define OFL_ROWS_F(V, A, S, D) OFL_KLAUNCH((warp_bwd_rows_kernel<RT, 2, V, A, S, D>), dim3(gr), dim3(kLdsNT), kRowsLdsBytes, st, q)
        if (q.src_b) { if (!valid || add || q.dst_flags) return (int)hipErrorInvalidValue; OFL_ROWS_F(true, 0, true, false); }          // mode 1 't': src - src_b
        else if (q.dst_flags) {                                    // (host: only with a valid mask)
            if (!valid) return (int)hipErrorInvalidValue;
            if (am == 0) OFL_ROWS_F(true, 0, false, true); else if (am == 1) OFL_ROWS_F(true, 1, false, true); else OFL_ROWS_F(true, 2, false, true);
        } else if (am == 1) { if (valid) OFL_ROWS_F(true, 1, false, false); else OFL_ROWS_F(false, 1, false, false); }                  // mode 3 proper
        else { if (valid) OFL_ROWS_F(true, 2, false, false); else OFL_ROWS_F(false, 2, false, false); }                                 // another addend
#undef OFL_ROWS_F
        return (int)hipGetLastError();
    }
    if (OFL_WARP_ROWS && rows && lean && !add) {                  // per-row extents instead of one sheared rectangle (warp_bwd_rows_kernel)
        const unsigned gr = RT == TT ? g : warp_geometry(q, kLdsTWQ * 4, RT * kLdsTH);
#define OFL_ROWS_CASE(NC)                                                                                                    \
        if (valid) OFL_KLAUNCH((warp_bwd_rows_kernel<RT, NC, true>), dim3(gr), dim3(kLdsNT), kRowsLdsBytes, st, q);   \
        else OFL_KLAUNCH((warp_bwd_rows_kernel<RT, NC, false>), dim3(gr), dim3(kLdsNT), kRowsLdsBytes, st, q);
        switch (nc) {
            case 1: OFL_ROWS_CASE(1) break;
            case 2: OFL_ROWS_CASE(2) break;
            default: OFL_ROWS_CASE(3) break;
        }
#undef OFL_ROWS_CASE
        return (int)hipGetLastError();
    }
#if OFL_WARP_COL_ADD >= 2
    if (add && OFL_WARP_REUSE && q.add_is_flow) {
        if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, true, true, false, false, float, float, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
        else OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, false, true, false, false, float, float, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
        return (int)hipGetLastError();
    }
    if (add) {                                     // (the fused composition: 2 channels)
        if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, true, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
        else OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
        return (int)hipGetLastError();
    }
#else
    if (add) return (int)hipErrorInvalidValue;     // (callers route an addend here only when the row-table kernel above takes it)
#endif
#define OFL_WIDE_CASE(NC)                                                                                                                   \
    if (lean) {                                                                                                                              \
        if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, NC, true, false, false, false, float, float, false, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);   \
        else OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, NC, false, false, false, false, float, float, false, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);        \
    } else if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, NC, true, false>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);      \
    else OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, NC, false, false>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
    switch (nc) {
        case 1: OFL_WIDE_CASE(1) break;
        case 2: OFL_WIDE_CASE(2) break;
        default: OFL_WIDE_CASE(3) break;
    }
#undef OFL_WIDE_CASE
    return (int)hipGetLastError();
}

// SMALL plain lean launches (below the four-tile column's threshold): the row-table kernel with one tile (tiny: B = 1 at 1080p) or two
// tiles per block -- B = 2 ... 6 at 1080p: -13 ... -14 % against the two-tile kernel on the sheared rectangle (53 -> 62 %, 58 -> 68 %,
// 61 -> 71 % of 8 TB/s; tools/small_once.py), B = 1 -1.5 %.  1 ... 3 channels and mode 3 proper.
int ofl_wide_launch_rows_small(const void* params, int nc, int valid, int add, int tiles, void* stream) {
    WarpParams q = *static_cast<const WarpParams*>(params);
    q.lds_bytes = kLdsBytes;
    hipStream_t st = (hipStream_t)stream;
    if (!OFL_WARP_ROWS || !warp_is_lean(q) || q.src_b || q.dst_flags || nc < 1 || nc > 3 || (add && !(nc == 2 && q.add_is_flow))) return (int)hipErrorInvalidValue;
    const unsigned gs = warp_geometry(q, kLdsTWQ * 4, (tiles == 1 ? 1 : 2) * kLdsTH);
#define OFL_ROWS_S(T_, NC, V, A) OFL_KLAUNCH((warp_bwd_rows_kernel<T_, NC, V, A>), dim3(gs), dim3(kLdsNT), kRowsLdsBytes, st, q)
#define OFL_ROWS_ST(T_)                                                                                   \
    if (add) { if (valid) OFL_ROWS_S(T_, 2, true, 1); else OFL_ROWS_S(T_, 2, false, 1); }                \
    else if (nc == 2) { if (valid) OFL_ROWS_S(T_, 2, true, 0); else OFL_ROWS_S(T_, 2, false, 0); }       \
    else if (nc == 1) { if (valid) OFL_ROWS_S(T_, 1, true, 0); else OFL_ROWS_S(T_, 1, false, 0); }       \
    else { if (valid) OFL_ROWS_S(T_, 3, true, 0); else OFL_ROWS_S(T_, 3, false, 0); }
    if (tiles == 1) { OFL_ROWS_ST(1) } else { OFL_ROWS_ST(2) }
#undef OFL_ROWS_ST
#undef OFL_ROWS_S
    return (int)hipGetLastError();
}
// a flow stored in fp16 gathered from its halves (ofl_warp_bwd_h_f32), large lean launches: the row-table kernel
int ofl_wide_launch_rows_h(const void* params, void* stream) {
    WarpParams q = *static_cast<const WarpParams*>(params);
    q.lds_bytes = kLdsBytes;
    constexpr int RT = OFL_ROWS_T;
    const unsigned gr = warp_geometry(q, kLdsTWQ * 4, RT * kLdsTH);
    if (!OFL_WARP_ROWS || !warp_is_lean(q) || !q.valid || q.addend || q.dst_flags) return OFL_E_UNSUPPORTED;     // (not this kernel's launch: the caller takes the column / pair kernel)
    if (q.src_b) OFL_KLAUNCH((warp_bwd_rows_kernel<RT, 2, true, 0, true, false, _Float16>), dim3(gr), dim3(kLdsNT), kRowsLdsBytes, (hipStream_t)stream, q);
    else OFL_KLAUNCH((warp_bwd_rows_kernel<RT, 2, true, 0, false, false, _Float16>), dim3(gr), dim3(kLdsNT), kRowsLdsBytes, (hipStream_t)stream, q);
    return (int)hipGetLastError();
}
// the gradient with respect to the flow (ofl_warp_bwd_grad_f32), large launches with W % 4 == 0: the row-table kernel
int ofl_wide_launch_rows_grad(const void* params, int nc, int tiles, void* stream) {      // tiles per block: OFL_ROWS_T for large launches, 1 / 2 for tiny / small ones
    WarpParams q = *static_cast<const WarpParams*>(params);
    q.lds_bytes = kLdsBytes;
    constexpr int RT = OFL_ROWS_T;
    const int tt = tiles == 1 ? 1 : (tiles == 2 ? 2 : RT);
    const unsigned gr = warp_geometry(q, kLdsTWQ * 4, tt * kLdsTH);
    if (!OFL_WARP_ROWS || !warp_is_lean(q) || q.valid || q.src_b || q.dst_flags) return OFL_E_UNSUPPORTED;
#define OFL_ROWS_G(T_, NC) OFL_KLAUNCH((warp_bwd_rows_kernel<T_, NC, false, 0, false, false, float, float, false, true>), dim3(gr), dim3(kLdsNT), kRowsLdsBytes, (hipStream_t)stream, q)
#define OFL_ROWS_GT(T_) switch (nc) { case 1: OFL_ROWS_G(T_, 1); break; case 2: OFL_ROWS_G(T_, 2); break; default: OFL_ROWS_G(T_, 3); break; }
    if (tt == 1) { OFL_ROWS_GT(1) } else if (tt == 2) { OFL_ROWS_GT(2) } else { OFL_ROWS_GT(RT) }
#undef OFL_ROWS_GT
#undef OFL_ROWS_G
    return (int)hipGetLastError();
}
// uint8 images warped from and to their bytes (ofl_warp_bwd_u8; 1 or 3 channels, W % 4 == 0), large launches: the row-table kernel
int ofl_wide_launch_rows_u8(const void* params, int nc, int dst_is_u8, void* stream) {
    WarpParams q = *static_cast<const WarpParams*>(params);
    q.lds_bytes = kLdsBytes;
    constexpr int RT = OFL_ROWS_T;
    const unsigned gr = warp_geometry(q, kLdsTWQ * 4, RT * kLdsTH);
    if (!OFL_WARP_ROWS || (q.w & 3) != 0 || q.flow_flags || q.addend || q.src_b || q.dst_flags || !(nc == 1 || nc == 3)) return OFL_E_UNSUPPORTED;
#define OFL_ROWS_U8(NC, V, TD) OFL_KLAUNCH((warp_bwd_rows_kernel<RT, NC, V, 0, false, false, uint8_t, TD, true>), dim3(gr), dim3(kLdsNT), kRowsLdsBytes, (hipStream_t)stream, q)
    if (dst_is_u8) {
        if (nc == 1) { if (q.valid) OFL_ROWS_U8(1, true, uint8_t); else OFL_ROWS_U8(1, false, uint8_t); }
        else { if (q.valid) OFL_ROWS_U8(3, true, uint8_t); else OFL_ROWS_U8(3, false, uint8_t); }
    } else {                                          // (Flow.apply of a uint8 image: fp32 out, rounded or not as the caller says)
        if (nc == 1) { if (q.valid) OFL_ROWS_U8(1, true, float); else OFL_ROWS_U8(1, false, float); }
        else { if (q.valid) OFL_ROWS_U8(3, true, float); else OFL_ROWS_U8(3, false, float); }
    }
#undef OFL_ROWS_U8
    return (int)hipGetLastError();
}
int ofl_wide_launch_chan(const void* params, int valid, int rows, void* stream) {
    WarpParams q = *static_cast<const WarpParams*>(params);
    q.lds_bytes = kLdsBytes;
#if OFL_WARP_CHAN_SUBS > 1
    {   // three tiles side by side per block, in lockstep (see the kernel): 768 threads, 3 x 52 KB of LDS, one block = 12 waves per CU
        constexpr int S = OFL_WARP_CHAN_SUBS;
        WarpParams q3 = q;
        const unsigned g3 = warp_geometry(q3, kLdsTWQ * 4 * S, kLdsTH);
        static bool attr_set = false;                             // (more than 64 KB of dynamic LDS must be asked for, once per kernel)
        if (!attr_set) {
            hipFuncSetAttribute((const void*)warp_bwd_lds_chan_kernel<true, true, S>, hipFuncAttributeMaxDynamicSharedMemorySize, S * kLdsBytes);
            hipFuncSetAttribute((const void*)warp_bwd_lds_chan_kernel<false, true, S>, hipFuncAttributeMaxDynamicSharedMemorySize, S * kLdsBytes);
            hipFuncSetAttribute((const void*)warp_bwd_lds_chan_kernel<true, false, S>, hipFuncAttributeMaxDynamicSharedMemorySize, S * kLdsBytes);
            hipFuncSetAttribute((const void*)warp_bwd_lds_chan_kernel<false, false, S>, hipFuncAttributeMaxDynamicSharedMemorySize, S * kLdsBytes);
            attr_set = true;
        }
        if (warp_is_lean(q3)) {
            if (valid) OFL_KLAUNCH((warp_bwd_lds_chan_kernel<true, true, S>), dim3(g3), dim3(kLdsNT * S), S * kLdsBytes, (hipStream_t)stream, q3);
            else OFL_KLAUNCH((warp_bwd_lds_chan_kernel<false, true, S>), dim3(g3), dim3(kLdsNT * S), S * kLdsBytes, (hipStream_t)stream, q3);
        } else if (valid) OFL_KLAUNCH((warp_bwd_lds_chan_kernel<true, false, S>), dim3(g3), dim3(kLdsNT * S), S * kLdsBytes, (hipStream_t)stream, q3);
        else OFL_KLAUNCH((warp_bwd_lds_chan_kernel<false, false, S>), dim3(g3), dim3(kLdsNT * S), S * kLdsBytes, (hipStream_t)stream, q3);
        return (int)hipGetLastError();
    }
#endif
    const unsigned g1 = warp_geometry(q, kLdsTWQ * 4, kLdsTH);
    if (OFL_WARP_ROWS && OFL_WARP_CHAN_SUBS == 1 && rows && warp_is_lean(q)) {
        if (valid) OFL_KLAUNCH((warp_bwd_lds_chan_kernel<true, true, 1, true>), dim3(g1), dim3(kLdsNT), kRowsLdsBytes, (hipStream_t)stream, q);
        else OFL_KLAUNCH((warp_bwd_lds_chan_kernel<false, true, 1, true>), dim3(g1), dim3(kLdsNT), kRowsLdsBytes, (hipStream_t)stream, q);
        return (int)hipGetLastError();
    }
    if (warp_is_lean(q)) {
        if (valid) OFL_KLAUNCH((warp_bwd_lds_chan_kernel<true, true>), dim3(g1), dim3(kLdsNT), kLdsBytes, (hipStream_t)stream, q);
        else OFL_KLAUNCH((warp_bwd_lds_chan_kernel<false, true>), dim3(g1), dim3(kLdsNT), kLdsBytes, (hipStream_t)stream, q);
    } else if (valid) OFL_KLAUNCH((warp_bwd_lds_chan_kernel<true>), dim3(g1), dim3(kLdsNT), kLdsBytes, (hipStream_t)stream, q);
    else OFL_KLAUNCH((warp_bwd_lds_chan_kernel<false>), dim3(g1), dim3(kLdsNT), kLdsBytes, (hipStream_t)stream, q);
    return (int)hipGetLastError();
}
#else
int g_warp_path = 0;   // ofl_set_option(OFL_OPT_WARP_PATH, .): 0 auto, 1 generic direct-gather kernel only, 2 (= auto), 3 / 4 staged with two tiles / one tile per block whatever the launch size (tests), 5 = auto but more than 3 channels as separate launches of 3 (tests: the channel-loop kernel against them), 6 = auto but the sheared rectangle instead of per-row extents (tests, A/B), 7 = auto but plain lean launches on four-tile row-table columns whatever the size (tests)
int g_warp_shear = 1;   // ofl_set_option(OFL_OPT_WARP_SHEAR, .)
int g_splat_pass_images = 0;   // ofl_set_option(OFL_OPT_SPLAT_PASS_IMAGES, .): 0 = automatic (one pass unless the fallback accumulator of a pass would pass 2^31 floats)
int g_splat_path = 0;   // ofl_set_option(OFL_OPT_SPLAT_PATH, .): 0 = the diet gather kernel (round 6: 16-24-byte records, one-word cells, 3 blocks per CU), 1 = round 5's gather kernel (tests compare the two bit for bit; A/B)
int g_splat_extra_lds = 0;   // ofl_set_option(OFL_OPT_SPLAT_EXTRA_LDS, .): bytes of dynamic LDS added to the diet kernel's launches (occupancy experiments: 28 672 -> two blocks per CU, 65 536 -> one)
int g_splat_fallback_slots = 0;   // ofl_set_option(OFL_OPT_SPLAT_FALLBACK_SLOTS, .): 0 = automatic (1 GiB); tests use 1 to exercise the rounds

template <int NC>
int launch_warp_lds(const WarpParams& p, unsigned grid, hipStream_t st) {
    const bool valid = p.valid != nullptr, add = p.addend != nullptr;
    if (NC == 2 && p.src_b) {                              // (host: 2 channels, valid mask, no addend, no output flags)
        if (kLdsT > 2 && g_warp_path != 3 && g_warp_path != 4) {   // large launches: columns of four tiles, lean when the promises hold
            WarpParams q = p;
            const unsigned gc = warp_geometry(q, kLdsTWQ * 4, kLdsT * kLdsTH);
            constexpr int TT = kLdsT > 2 ? kLdsT : 3;
            if (gc >= 6912u) {
                if (OFL_WARP_ROWS_FLOWOPS && warp_is_lean(q) && g_warp_path != 6) return ofl_wide_launch_column(&p, 2, 1, 0, 1, (void*)st);   // 64 x 16 tiles, per-row extents
                if (warp_is_lean(q)) OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, true, false, false, NC == 2, float, float, false, false, true>), dim3(gc), dim3(kLdsNT), kLdsBytes, st, q);
                else OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, true, false, false, NC == 2>), dim3(gc), dim3(kLdsNT), kLdsBytes, st, q);
                return (int)hipGetLastError();
            }
        }
        OFL_KLAUNCH((warp_bwd_lds_kernel<NC, true, false, false, NC == 2>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
        return (int)hipGetLastError();
    }
    if (NC == 2 && p.dst_flags) {                          // (host: only with a valid mask)
        if (OFL_WARP_ROWS_FLOWOPS && kLdsT > 2 && valid && g_warp_path != 6 && g_warp_path != 3 && g_warp_path != 4) {   // large lean launches: the row-table kernel keeps the flag by-product
            WarpParams q = p;
            if (warp_is_lean(q) && warp_geometry(q, kLdsTWQ * 4, kLdsT * kLdsTH) >= 6912u) return ofl_wide_launch_column(&p, 2, 1, add ? 1 : 0, 1, (void*)st);
        }
        if (add) OFL_KLAUNCH((warp_bwd_lds_kernel<NC, true, true, NC == 2>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
        else OFL_KLAUNCH((warp_bwd_lds_kernel<NC, true, false, NC == 2>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
        return (int)hipGetLastError();
    }
    // the fused composition (an addend): row tables where the launch is lean, else -- large launches only -- the four-tile column kernels
    // (REUSE / lean twins: no scratch since round 5); small and forced-path launches fall through to the pair / one-tile kernels
    if (OFL_WARP_COL_ADD && kLdsT > 2 && add && NC == 2 && !p.flow_flags) {
        WarpParams q = p;
        const unsigned g = warp_geometry(q, kLdsTWQ * 4, kLdsT * kLdsTH);
        if (OFL_WARP_COL_ADD >= 2) return ofl_wide_launch_column(&p, NC, valid ? 1 : 0, 1, 0, (void*)st);
        // large launches of mode 3 proper: 64 x 16 tiles with per-row extents (warp_bwd_rows_kernel<.., ADD>)
        if (OFL_WARP_ROWS_ADD && NC == 2 && (p.add_is_flow || OFL_WARP_ROWS_FLOWOPS) && warp_is_lean(q) && g >= (g_warp_path == 0 && p.add_is_flow && OFL_WARP_ROWS_SMALL ? OFL_ROWS_T4_MIN : 6912u) && g_warp_path != 6 && g_warp_path != 3 && g_warp_path != 4)
            return ofl_wide_launch_column(&p, NC, valid ? 1 : 0, 1, 1, (void*)st);
        if (OFL_WARP_ROWS_SMALL && NC == 2 && p.add_is_flow && warp_is_lean(q) && g < OFL_ROWS_T4_MIN && g_warp_path == 0) {     // small mode 3
            WarpParams q1 = p;
            return ofl_wide_launch_rows_small(&p, 2, valid ? 1 : 0, 1, warp_geometry(q1, kLdsTWQ * 4, kLdsTH) < OFL_ROWS_T1_MAX ? 1 : 2, (void*)st);
        }
        // the four-tile column kernels below are for LARGE launches on the automatic path (small launches keep the pair /
        // one-tile kernels further down -- half as many, twice as long blocks lose there -- and OFL_OPT_WARP_PATH 3 / 4 must reach them)
        if (g >= 6912u && g_warp_path != 3 && g_warp_path != 4) {
        if (OFL_WARP_REUSE && NC == 2 && p.add_is_flow) {
            if (warp_is_lean(q)) {
                if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), 2, true, true, false, false, float, float, false, true, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
                else OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), 2, false, true, false, false, float, float, false, true, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
            } else if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), 2, true, true, false, false, float, float, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
            else OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), 2, false, true, false, false, float, float, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
            return (int)hipGetLastError();
        }
        if (warp_is_lean(q)) {                     // (the addend is another flow: the outer `flow - (...)` of modes 1-2, Flow.combine's cells)
            if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), NC, true, true, false, false, float, float, false, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
            else OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), NC, false, true, false, false, float, float, false, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
            return (int)hipGetLastError();
        }
        if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), NC, true, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
        else OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), NC, false, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
        return (int)hipGetLastError();
        }
    }
    // a plain warp (Flow.apply 't' of an image or a flow): columns of kLdsT tiles -- unless the launch is SMALL: a column block
    // lives 4 tiles long, and with fewer than ~4.5 blocks per resident slot (256 CUs x 6) the chip idles behind the last
    // ones.  Below that the pair kernel's twice-as-many, half-as-long blocks win: B = 1 / 2 / 4 / 6 at 1080p -4 / -15 / -7 /
    // -5 %, B >= 8 the column kernel by 1-4 % (profiles/r4_small_batch_kernel_choice.txt)
    constexpr unsigned kColumnMinGroups = 6912;
    // ... and a TINY launch (fewer single tiles than that: a 1080p frame at B = 1 has 4 080) takes ONE tile per block: 2.7 short
    // rounds of blocks instead of 1.3 long ones -- B = 1 1080p apply 21.7 -> 19.2 us, mode 3 20.6 -> 18.4 us; from B = 2 the
    // pair kernel is level or ahead (profiles/r4_small_batch_kernel_choice.txt)
    {
        WarpParams q1 = p;
        const unsigned g1 = warp_geometry(q1, kLdsTWQ * 4, kLdsTH);
        if (g_warp_path == 7 && !add && warp_is_lean(q1)) return ofl_wide_launch_column(&p, NC, valid ? 1 : 0, 0, 1, (void*)st);   // (tests / experiments: four-tile columns with row tables whatever the size)
        // small plain launches on the row-table kernel (1 tile per block for tiny ones, else 2): see ofl_wide_launch_rows_small
        if (OFL_WARP_ROWS_SMALL && !add && !p.flow_flags && g_warp_path == 0 && warp_is_lean(q1)) {
            WarpParams q4 = p;
            const unsigned g4 = warp_geometry(q4, kLdsTWQ * 4, kLdsT * kLdsTH);
            if (g4 < OFL_ROWS_T4_MIN) return ofl_wide_launch_rows_small(&p, NC, valid ? 1 : 0, 0, g1 < OFL_ROWS_T1_MAX ? 1 : 2, (void*)st);
            if (g4 < kColumnMinGroups) return ofl_wide_launch_column(&p, NC, valid ? 1 : 0, 0, 1, (void*)st);      // (four tiles per block a little earlier than the rectangle's kernels: B = 6 at 1080p 76.5 -> 74.5 us)
        }
        if (g_warp_path == 4 || (g_warp_path != 3 && g1 < kColumnMinGroups)) {
            if (warp_is_lean(q1) && valid) {            // (the lean twins of the two instantiations with a valid mask: BASELINE configs[1] is one of them)
                if (add) OFL_KLAUNCH((warp_bwd_lds_column_kernel<1, NC, true, true, false, false, float, float, false, false, true>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q1);
                else OFL_KLAUNCH((warp_bwd_lds_column_kernel<1, NC, true, false, false, false, float, float, false, false, true>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q1);
                return (int)hipGetLastError();
            }
            if (valid && add) OFL_KLAUNCH((warp_bwd_lds_column_kernel<1, NC, true, true>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q1);
            else if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<1, NC, true, false>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q1);
            else if (add) OFL_KLAUNCH((warp_bwd_lds_column_kernel<1, NC, false, true>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q1);
            else OFL_KLAUNCH((warp_bwd_lds_column_kernel<1, NC, false, false>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q1);
            return (int)hipGetLastError();
        }
    }
    WarpParams q = p;
    const unsigned g = (kLdsT > 2 && !add && !p.flow_flags) ? warp_geometry(q, kLdsTWQ * 4, kLdsT * kLdsTH) : 0u;
    if (g >= kColumnMinGroups && g_warp_path != 3) {
        if (OFL_WARP_WIDE) return ofl_wide_launch_column(&p, NC, valid ? 1 : 0, 0, g_warp_path != 6, (void*)st);    // 64 x 16 tiles: -1.9 % (see kLdsNT)
        if (valid) OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), NC, true, false>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
        else OFL_KLAUNCH((warp_bwd_lds_column_kernel<(kLdsT > 2 ? kLdsT : 3), NC, false, false>), dim3(g), dim3(kLdsNT), kLdsBytes, st, q);
        return (int)hipGetLastError();
    }
#define OFL_LAUNCH_L(V, A)                                                                                       \
    if (valid == V && add == A) {                                                                                \
        OFL_KLAUNCH((warp_bwd_lds_kernel<NC, V, A>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);          \
        return (int)hipGetLastError();                                                                           \
    }
    OFL_LAUNCH_L(false, false) OFL_LAUNCH_L(true, false) OFL_LAUNCH_L(false, true) OFL_LAUNCH_L(true, true)
#undef OFL_LAUNCH_L
    return OFL_E_ARG;
}

// 8-bit images: uint8 source planes, float or uint8 destination (no addend, no flag words)
template <int NC, typename TD>
int launch_warp_lds_u8(const WarpParams& p, unsigned grid, hipStream_t st) {
    if (p.valid) OFL_KLAUNCH((warp_bwd_lds_kernel<NC, true, false, false, false, uint8_t, TD>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
    else OFL_KLAUNCH((warp_bwd_lds_kernel<NC, false, false, false, false, uint8_t, TD>), dim3(grid), dim3(kLdsNT), kLdsBytes, st, p);
    return (int)hipGetLastError();
}

inline bool aligned_to(const void* ptr, size_t a) { return (reinterpret_cast<uintptr_t>(ptr) % a) == 0; }

#ifndef OFL_FLAGS_BLOCKS
#define OFL_FLAGS_BLOCKS 512
#endif
// `fh`: optional hand-over of the words to host-visible memory by the last block (ofl_flow_flags_host)
void launch_flow_flags(const float* flow, int64_t flow_bs, const uint8_t* mask, int64_t mask_bs, int32_t* flags, int32_t n,
                       int64_t hw, hipStream_t st, const FlagsHost* fhp = nullptr) {
    const bool vec = (hw % 4) == 0 && aligned_to(flow, 16) && (flow_bs % 4) == 0 && (!mask || (aligned_to(mask, 4) && (mask_bs % 4) == 0));
    FlagsHost fh = {};
    if (fhp) fh = *fhp;
    if (vec) {
        // about 512 blocks in all: every wave ends with a look at (and maybe an atomic on) its image's shared word, and
        // few long-running blocks stream better than many short ones (B=64 1080p: 3.1 -> 5.8 TB/s; B=16: 2.5 -> 5.4);
        // 256 for a batch of 8 or fewer (1080p wait, 512 / 256 blocks: B = 8 42.1 / 38.3 us, B = 4 31.8 / 28.9 us; B = 16 59.7 / 65.8 us)
        int64_t bx = (hw / 4 + 1023) / 1024;               // 4 groups of 4 pixels per thread and step
        int64_t cap = (n <= 8 ? OFL_FLAGS_BLOCKS / 2 : OFL_FLAGS_BLOCKS) / n;
        cap = cap < 16 ? 16 : (cap > 256 ? 256 : cap);
        if (bx > cap) bx = cap;
        fh.total_blocks = (int32_t)(bx * n);
        const bool nt = 9 * hw * n >= ((int64_t)256 << 20);
        if (fhp) {
            if (nt) OFL_KLAUNCH((flow_flags_kernel<true, true, true>), dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, flow, flow_bs, mask, mask_bs, flags, hw, fh);
            else OFL_KLAUNCH((flow_flags_kernel<true, false, true>), dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, flow, flow_bs, mask, mask_bs, flags, hw, fh);
        } else {
            if (nt) OFL_KLAUNCH((flow_flags_kernel<true, true>), dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, flow, flow_bs, mask, mask_bs, flags, hw, fh);
            else OFL_KLAUNCH((flow_flags_kernel<true, false>), dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, flow, flow_bs, mask, mask_bs, flags, hw, fh);
        }
    } else {
        int64_t bx = (hw + 255) / 256;
        if (bx > 512) bx = 512;
        fh.total_blocks = (int32_t)(bx * n);
        if (fhp) OFL_KLAUNCH((flow_flags_kernel<false, false, true>), dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, flow, flow_bs, mask, mask_bs, flags, hw, fh);
        else OFL_KLAUNCH((flow_flags_kernel<false>), dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, flow, flow_bs, mask, mask_bs, flags, hw, fh);
    }
}

template <int CT>
int launch_warp(const WarpParams& p, unsigned grid, hipStream_t st) {
    const bool valid = p.valid != nullptr, add = p.addend != nullptr, flags = p.flow_flags != nullptr;
#define OFL_LAUNCH_W(V, A, F)                                                                  \
    if (valid == V && add == A && flags == F) {                                                \
        OFL_KLAUNCH((warp_bwd_kernel<CT, V, A, F>), dim3(grid), dim3(256), 0, st, p);    \
        return (int)hipGetLastError();                                                         \
    }
    OFL_LAUNCH_W(false, false, false) OFL_LAUNCH_W(true, false, false)
    OFL_LAUNCH_W(false, true, false) OFL_LAUNCH_W(true, true, false)
    OFL_LAUNCH_W(false, false, true) OFL_LAUNCH_W(true, false, true)
    OFL_LAUNCH_W(false, true, true) OFL_LAUNCH_W(true, true, true)
#undef OFL_LAUNCH_W
    return OFL_E_ARG;
}


// How many 256-thread blocks of a fallback-kernel instantiation the current device holds AT ONCE, halved (the grid barrier of
// splat_fallback_kernel spins: a block that cannot be scheduled would hang every resident one).  Occupancy query x CU count, per
// device and instantiation, cached; one block per CU fewer than the API says where it says more than one (the API is one high
// for some register counts: MI355X_MICROARCH.md, residency); never above kFallbackBlocks, never below 1.
inline unsigned fallback_resident_blocks(const void* kernel, int which) {
    static int cache[64][5];                                  // 0 = not asked yet (a benign race: every asker stores the same number)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int v = cache[dev][which];
    if (v == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1;
        if (per_cu > 1) per_cu -= 1;
        if (per_cu < 1) per_cu = 1;
        int64_t b = ((int64_t)per_cu * cus) / 2;
        if (b > (int64_t)kFallbackBlocks) b = kFallbackBlocks;
        if (b < 1) b = 1;
        v = (int)b;
        cache[dev][which] = v;
    }
    return (unsigned)v;
}

template <int NC, bool MCH, typename TF = float, typename TO = float>
int launch_splat_gather2(const GatherParams& gp, unsigned grid, hipStream_t st) {
    // round 6: the diet kernel (3 blocks per CU) unless the tests / an A-B ask for round 5's (32-byte records, 2 blocks per CU)
    if (g_splat_path != 1) return ofl_splat_launch_gather_diet(&gp, NC, MCH ? 1 : 0, std::is_same<TF, float>::value ? 0 : (std::is_same<TO, float>::value ? 1 : 2), grid, (void*)st, g_splat_extra_lds);
    if (NC >= 2 && OFL_SP_LEAN && splat_is_lean(gp.s)) OFL_KLAUNCH((splat_gather_kernel<NC, MCH, TF, TO, (NC >= 2)>), dim3(grid), dim3(kSpNT2), 0, st, gp);
    else OFL_KLAUNCH((splat_gather_kernel<NC, MCH, TF, TO>), dim3(grid), dim3(kSpNT2), 0, st, gp);
    return (int)hipGetLastError();
}

template <int NC>
int launch_splat_gather(const GatherParams& gp, unsigned grid, hipStream_t st) {
    return gp.s.with_mask_chan ? launch_splat_gather2<NC, true>(gp, grid, st) : launch_splat_gather2<NC, false>(gp, grid, st);
}

// fp16-stored flows as warper AND data (2 channels): elem 1 = fp32 outputs, 2 = fp16 outputs
int launch_splat_gather_half(const GatherParams& gp, unsigned grid, hipStream_t st, int elem) {
    if (gp.s.with_mask_chan) return elem == 2 ? launch_splat_gather2<2, true, _Float16, _Float16>(gp, grid, st) : launch_splat_gather2<2, true, _Float16, float>(gp, grid, st);
    return elem == 2 ? launch_splat_gather2<2, false, _Float16, _Float16>(gp, grid, st) : launch_splat_gather2<2, false, _Float16, float>(gp, grid, st);
}

}  // namespace

// The gradient of the backward warp with respect to its FLOW on the staged column kernel (called by ofl_warp_bwd_grad_f32 in
// ofl_aux_kernels.hip; not exported).  OFL_E_UNSUPPORTED: the frame is not eligible (the caller's one-pixel-per-lane kernel takes it).
int ofl_internal_warp_grad_flow_lds(const float* flow, int64_t flow_bs, float flow_sign, const float* src, int64_t src_bs,
                                    const float* grad_out, float g_scale, float* grad_flow, int32_t n, int32_t c, int32_t h, int32_t w,
                                    hipStream_t st) {
    const bool lds_ok = g_warp_path != 1 && c >= 1 && c <= 3 && w >= 4 && h >= 2 && w < 32760 && h < 32760 && (int64_t)h * w < (1ll << 24);
    if (!lds_ok || kLdsT <= 2) return OFL_E_UNSUPPORTED;
    if ((int64_t)((w + 31) / 32) * ((h + 15) / 16) * n >= (1ll << 31)) return OFL_E_SHAPE;
    WarpParams p = {};
    p.flow = flow; p.flow_bs = flow_bs; p.src = src; p.src_bs = src_bs;
    p.addend = grad_out; p.addend_bs = (int64_t)c * h * w;           // (the column kernel fetches the upstream gradient where mode 3 fetches its addend)
    p.dst = grad_flow; p.dst_bs = (int64_t)2 * h * w;
    p.n = n; p.c = c; p.h = h; p.w = w;
    p.flow_sign = flow_sign; p.a_sign = 1.0f; p.g_sign = g_scale; p.round_mode = OFL_ROUND_NONE;
    p.wm1 = (float)(w - 1); p.hm1 = (float)(h - 1);
    p.half_wm1 = p.wm1 / 2.0f; p.half_hm1 = p.hm1 / 2.0f;
    p.rcp_wm1 = 1.0f / p.wm1; p.rcp_hm1 = 1.0f / p.hm1;
    p.lds_bytes = kLdsBytes;
    p.shear = (g_warp_shear && (int64_t)h + 4 * (int64_t)w + 8 < 32760) ? 1 : 0;
    const unsigned g = warp_geometry(p, kLdsTWQ * 4, kLdsT * kLdsTH);
    constexpr int TT = kLdsT > 2 ? kLdsT : 3;
    if (OFL_WARP_ROWS_FLOWOPS && warp_is_lean(p) && g_warp_path != 6 && g_warp_path != 3 && g_warp_path != 4) {   // 64 x 16 tiles, per-row extents
        if (g >= (OFL_WARP_ROWS_SMALL && g_warp_path == 0 ? OFL_ROWS_T4_MIN : 6912u) || g_warp_path == 7) {
            const int rr = ofl_wide_launch_rows_grad(&p, c, OFL_ROWS_T, (void*)st);
            if (rr != OFL_E_UNSUPPORTED) return rr;
        }
        if (OFL_WARP_ROWS_SMALL && g_warp_path == 0) {            // small launches (the shapes training runs at): one tile per block for tiny ones, else two
            WarpParams q1 = p;
            const int rr = ofl_wide_launch_rows_grad(&p, c, warp_geometry(q1, kLdsTWQ * 4, kLdsTH) < OFL_ROWS_T1_MAX ? 1 : 2, (void*)st);
            if (rr != OFL_E_UNSUPPORTED) return rr;
        }
    }
    switch (c) {
        case 1: OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 1, false, false, false, false, float, float, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, p); break;
        case 2: OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, false, false, false, false, float, float, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, p); break;
        default: OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 3, false, false, false, false, float, float, true>), dim3(g), dim3(kLdsNT), kLdsBytes, st, p); break;
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

}  // extern "C"
const void* g_ofl_last_kernel = nullptr;
extern "C" {
__attribute__((visibility("default"))) const char* ofl_last_kernel_name(void) {
    return g_ofl_last_kernel ? hipKernelNameRefByPtr(g_ofl_last_kernel, nullptr) : "";
}

__attribute__((visibility("default"))) int ofl_version(void) { return 33; }   // 33: the gather splat's diet kernel (ofl_splat_gather.hip), OFL_OPT_SPLAT_PATH / OFL_OPT_SPLAT_EXTRA_LDS, ofl_splat_gather_info; 32: OFL_OPT_WARP_PATH value 7 (four-tile row-table columns whatever the size); row tables for small launches, the other flow-level warps, fp16 / uint8 sources, the gradient-wrt-flow pass; 31: OFL_OPT_WARP_PATH value 6 (the sheared rectangle instead of per-row extents: warp_bwd_rows_kernel is the default for large lean launches); 30: OFL_OPT_WARP_PATH value 5 (more than 3 channels as launches of 3; the default is ONE launch that loops over the channels); 29: ofl_splat_tile_geometry (64 x 16 destination tiles); 28: ofl_resize_bilinear_f32; 27: ofl_warp_valid_f32 (ofl_aux_kernels.hip); 26: bounded fallback accumulator of the gather splat (ofl_splat_tiled_fallback_images); 25: ofl_flow_flags_host with sharded arrival counters and {serial, word} pairs, ofl_flow_from_matrix_f32; 24: ofl_flow_flags_host (+ ofl_host_words_alloc / _free); 23: scratch argument of ofl_splat_grad_f32; 22: ofl_splat_sum_f32; 21: ofl_flag_words_or_i32, splat workspace without the fold-tile list; 20: fp16-stored flows read directly (ofl_splat_tiled_f16, ofl_warp_bwd_h_f32, flags-only ofl_flow_from_f16); 19: gather-formulation splat (workspace layout), ofl_warp_bwd_win_f32 / ofl_splat_tiled_win_f32 (padded apply); 18: ofl_aux_kernels.hip (backward passes, point sampler, extents); 13: dst_flags in ofl_warp_bwd_f32 / ofl_splat_tiled_f32, fixed-address splat queues; 14: data_b; 15: src_b; 16: ofl_warp_bwd_u8; 17: ofl_flow_from_f16

__attribute__((visibility("default"))) int ofl_set_option(int32_t key, int32_t value) {
    if (key == OFL_OPT_WARP_PATH && value >= 0 && value <= 7) { g_warp_path = value; return OFL_OK; }
    if (key == OFL_OPT_WARP_SHEAR && (value == 0 || value == 1)) { g_warp_shear = value; return OFL_OK; }
    if (key == OFL_OPT_SPLAT_PASS_IMAGES && value >= 0) { g_splat_pass_images = value; return OFL_OK; }
    if (key == OFL_OPT_SPLAT_FALLBACK_SLOTS && value >= 0) { g_splat_fallback_slots = value; return OFL_OK; }
    if (key == OFL_OPT_SPLAT_PATH && (value == 0 || value == 1)) { g_splat_path = value; return OFL_OK; }
    if (key == OFL_OPT_SPLAT_EXTRA_LDS && value >= 0 && value <= 100 * 1024) { g_splat_extra_lds = value; return OFL_OK; }
    return OFL_E_ARG;
}

static int warp_bwd_impl(
    const float* flow, int64_t flow_bs, float flow_sign, const float* src, int64_t src_bs,
    const float* src_b, int64_t src_b_bs, const uint8_t* src_mask, int64_t src_mask_bs, const uint8_t* flow_mask, int64_t flow_mask_bs,
    const float* addend, int64_t addend_bs, float a_sign, float g_sign, float* dst, uint8_t* valid,
    int32_t* flow_flags, int32_t* src_flags, int32_t* dst_flags, int32_t n, int32_t c, int32_t h, int32_t w,
    int32_t round_mode, void* stream, int32_t fh, int32_t fw, int32_t foy, int32_t fox) {
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
    p.fh = fh; p.fw = fw; p.foy = foy; p.fox = fox;
    hipStream_t st = (hipStream_t)stream;
    if ((int64_t)((w + 31) / 32) * ((h + 15) / 16) * n >= (1ll << 31)) return OFL_E_SHAPE;
    if (dst_flags) {
        hipError_t e = hipMemsetAsync(dst_flags, 0, (size_t)n * sizeof(int32_t), st);
        if (e != hipSuccess) return (int)e;
    }
    // LDS-staged fast path: <= 3 channels, at least one whole 4-pixel group per row, 16-bit box coordinates (any width:
    // 16-byte accesses at 4-byte alignment, mask bytes at any alignment)
    const bool lds_ok = g_warp_path != 1 && w >= 4 && h >= 2 && w < 32760 && h < 32760 && (int64_t)h * w < (1ll << 24) && fw == 0;
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
        // more than 3 channels of a plain warp: ONE launch that walks the channels inside the block (warp_bwd_lds_chan_kernel)
        // (with the valid area wanted the first group is 3 channels + the mask plane: below 7 channels two launches of 3 are as good or better)
        if (OFL_WARP_CHAN && c >= (valid ? 7 : 4) && (w & 3) == 0 && !addend && !src_b && !dst_flags && g_warp_path != 5) {
            WarpParams q = p;
            const unsigned g1 = warp_geometry(q, kLdsTWQ * 4, kLdsTH);
            // 64 x 16 tiles for large launches -- and, with per-row extents (lean, two channel groups or more), for EVERY size: B = 1 C = 64
            // 0.355 -> 0.255 ms, C = 16 0.071 -> 0.054 ms against the 32 x 16 rectangle version (tools/bench_chan.py --batch 1)
            const bool rows_chan = OFL_WARP_ROWS && g_warp_path != 6 && c >= 7 && warp_is_lean(q);
            if (OFL_WARP_CHAN_WIDE && (g1 >= OFL_WARP_CHAN_WIDE_MIN || rows_chan)) return ofl_wide_launch_chan(&p, p.valid ? 1 : 0, rows_chan ? 1 : 0, (void*)st);   // (row extents from two channel groups on: with one group their set-up is not amortised -- C = 4: 0.148 against 0.138 ms)
            if (warp_is_lean(q)) {
                if (q.valid) OFL_KLAUNCH((warp_bwd_lds_chan_kernel<true, true>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q);
                else OFL_KLAUNCH((warp_bwd_lds_chan_kernel<false, true>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q);
            } else if (q.valid) OFL_KLAUNCH((warp_bwd_lds_chan_kernel<true>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q);
            else OFL_KLAUNCH((warp_bwd_lds_chan_kernel<false>), dim3(g1), dim3(kLdsNT), kLdsBytes, st, q);
            return (int)hipGetLastError();
        }
        // otherwise groups of 3 (the staged box holds 3 channels + the mask channel); the valid mask and the
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

__attribute__((visibility("default"))) int ofl_warp_bwd_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* src, int64_t src_bs,
    const float* src_b, int64_t src_b_bs, const uint8_t* src_mask, int64_t src_mask_bs, const uint8_t* flow_mask, int64_t flow_mask_bs,
    const float* addend, int64_t addend_bs, float a_sign, float g_sign, float* dst, uint8_t* valid,
    int32_t* flow_flags, int32_t* src_flags, int32_t* dst_flags, int32_t n, int32_t c, int32_t h, int32_t w,
    int32_t round_mode, void* stream) {
    return warp_bwd_impl(flow, flow_bs, flow_sign, src, src_bs, src_b, src_b_bs, src_mask, src_mask_bs, flow_mask, flow_mask_bs, addend,
                         addend_bs, a_sign, g_sign, dst, valid, flow_flags, src_flags, dst_flags, n, c, h, w, round_mode, stream, 0, 0, 0, 0);
}

__attribute__((visibility("default"))) int ofl_warp_bwd_win_f32(
    const float* flow, int64_t flow_bs, float flow_sign, int32_t fh, int32_t fw, int32_t foy, int32_t fox,
    const float* src, int64_t src_bs, const uint8_t* src_mask, int64_t src_mask_bs, const uint8_t* flow_mask, int64_t flow_mask_bs,
    float* dst, uint8_t* valid, int32_t n, int32_t c, int32_t h, int32_t w, int32_t round_mode, void* stream) {
    if (fh < 1 || fw < 1 || foy < 0 || fox < 0 || foy + fh > h || fox + fw > w) return OFL_E_ARG;
    return warp_bwd_impl(flow, flow_bs, flow_sign, src, src_bs, nullptr, 0, src_mask, src_mask_bs, flow_mask, flow_mask_bs, nullptr, 0,
                         1.0f, 1.0f, dst, valid, nullptr, nullptr, nullptr, n, c, h, w, round_mode, stream, fh, fw, foy, fox);
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
        if (OFL_WARP_ROWS_FLOWOPS && (nc == 1 || nc == 3) && (w & 3) == 0 && kLdsT > 2 && g_warp_path != 6 && g_warp_path != 3 && g_warp_path != 4) {
            WarpParams qg = q;                           // large launches: 64 x 16 tiles, per-row extents
            if (warp_geometry(qg, kLdsTWQ * 4, kLdsT * kLdsTH) >= 6912u) { rc = ofl_wide_launch_rows_u8(&q, nc, dst_is_u8, stream); if (rc == OFL_OK) continue; if (rc != OFL_E_UNSUPPORTED) return rc; }
        }
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

// flows stored in fp16 as the SOURCE of a backward warp (2 channels; BASELINE config 5: the `flow` operand of combine_with
// mode 1 't' is gathered straight from its fp16 planes, optionally minus an fp32 src_b): staged kernel only
__attribute__((visibility("default"))) int ofl_warp_bwd_h_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const void* src_f16, int64_t src_bs, const float* src_b, int64_t src_b_bs,
    const uint8_t* src_mask, int64_t src_mask_bs, const uint8_t* flow_mask, int64_t flow_mask_bs,
    float* dst, uint8_t* valid, int32_t n, int32_t h, int32_t w, void* stream) {
    if (!flow || !src_f16 || !dst || !valid) return OFL_E_NULL;
    int rc = check_dims(n, 2, h, w, false);
    if (rc) return rc;
    if (!(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    if ((int64_t)((w + 31) / 32) * ((h + 15) / 16) * n >= (1ll << 31)) return OFL_E_SHAPE;
    if (!(g_warp_path != 1 && w >= 4 && h >= 2 && w < 32760 && h < 32760 && (int64_t)h * w < (1ll << 24))) return OFL_E_UNSUPPORTED;
    WarpParams p = {};
    p.flow = flow; p.flow_bs = flow_bs; p.src = static_cast<const float*>(src_f16); p.src_bs = src_bs;
    p.src_b = src_b; p.src_b_bs = src_b_bs;
    p.src_mask = src_mask; p.src_mask_bs = src_mask_bs; p.flow_mask = flow_mask; p.flow_mask_bs = flow_mask_bs;
    p.dst = dst; p.valid = valid;
    p.n = n; p.c = 2; p.h = h; p.w = w;
    p.flow_sign = flow_sign; p.a_sign = 1.0f; p.g_sign = 1.0f; p.round_mode = OFL_ROUND_NONE;
    p.wm1 = (float)(w - 1); p.hm1 = (float)(h - 1);
    p.half_wm1 = p.wm1 / 2.0f; p.half_hm1 = p.hm1 / 2.0f;
    p.rcp_wm1 = 1.0f / p.wm1; p.rcp_hm1 = 1.0f / p.hm1;
    p.lds_bytes = kLdsBytes;
    p.shear = (g_warp_shear && (int64_t)h + 4 * (int64_t)w + 8 < 32760) ? 1 : 0;
    p.dst_bs = (int64_t)2 * h * w;
    hipStream_t st = (hipStream_t)stream;
    // large launches: columns of four tiles (round 5: every wait of that pipeline is counted, see lds_store), lean when the promises hold
    if (kLdsT > 2 && g_warp_path != 3 && g_warp_path != 4) {
        WarpParams q = p;
        const unsigned gc = warp_geometry(q, kLdsTWQ * 4, kLdsT * kLdsTH);
        constexpr int TT = kLdsT > 2 ? kLdsT : 3;
        if (gc >= 6912u) {
            const bool lean = warp_is_lean(q);
            if (OFL_WARP_ROWS_FLOWOPS && lean && g_warp_path != 6) {                                                  // 64 x 16 tiles, per-row extents
                const int rr = ofl_wide_launch_rows_h(&p, stream);
                if (rr != OFL_E_UNSUPPORTED) return rr;                                                            // (a launch it declines falls through to the column kernel)
            }
            if (src_b) { if (lean) OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, true, false, false, true, _Float16, float, false, false, true>), dim3(gc), dim3(kLdsNT), kLdsBytes, st, q);
                         else OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, true, false, false, true, _Float16, float>), dim3(gc), dim3(kLdsNT), kLdsBytes, st, q); }
            else { if (lean) OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, true, false, false, false, _Float16, float, false, false, true>), dim3(gc), dim3(kLdsNT), kLdsBytes, st, q);
                   else OFL_KLAUNCH((warp_bwd_lds_column_kernel<TT, 2, true, false, false, false, _Float16, float>), dim3(gc), dim3(kLdsNT), kLdsBytes, st, q); }
            return (int)hipGetLastError();
        }
    }
    const unsigned g = warp_geometry(p, kLdsTWQ * 4, 2 * kLdsTH);
    if (src_b) OFL_KLAUNCH((warp_bwd_lds_kernel<2, true, false, false, true, _Float16, float>), dim3(g), dim3(kLdsNT), kLdsBytes, st, p);
    else OFL_KLAUNCH((warp_bwd_lds_kernel<2, true, false, false, false, _Float16, float>), dim3(g), dim3(kLdsNT), kLdsBytes, st, p);
    return (int)hipGetLastError();
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
        case 1: OFL_KLAUNCH(splat_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, p); break;
        case 2: OFL_KLAUNCH(splat_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, p); break;
        case 3: OFL_KLAUNCH(splat_fwd_kernel<3>, dim3(grid), dim3(256), 0, st, p); break;
        case 4: OFL_KLAUNCH(splat_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, p); break;
        default: OFL_KLAUNCH(splat_fwd_kernel<0>, dim3(grid), dim3(256), 0, st, p); break;
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
        case 1: OFL_KLAUNCH(splat_finalize_kernel<1>, dim3(grid), dim3(256), 0, st, p); break;
        case 2: OFL_KLAUNCH(splat_finalize_kernel<2>, dim3(grid), dim3(256), 0, st, p); break;
        case 3: OFL_KLAUNCH(splat_finalize_kernel<3>, dim3(grid), dim3(256), 0, st, p); break;
        case 4: OFL_KLAUNCH(splat_finalize_kernel<4>, dim3(grid), dim3(256), 0, st, p); break;
        default: OFL_KLAUNCH(splat_finalize_kernel<0>, dim3(grid), dim3(256), 0, st, p); break;
    }
    return (int)hipGetLastError();
}


// workspace words of one pass of `images` frames: statistics (8: [0] some image on the two-pass path, [1] fold tiles, [2] images on
// the two-pass path, [3] tiles that took the gather's second launch, [4..5] grid-barrier arrivals, [6] length of the redo list) |
// per-image fallback flags | list lengths | lists | redo list
static int64_t splat_pass_words(int64_t images, int32_t h, int32_t w) {
    const int64_t tiles = images * ((w + kSpTW - 1) / kSpTW) * ((h + kSpTH - 1) / kSpTH);
    return 8 + ((images + 3) & ~(int64_t)3) + ((tiles + 3) & ~(int64_t)3) + (int64_t)kBinCap * tiles + 2 * kSpTH * tiles;   // (+ the redo list: up to kSpTH bands per tile, 2 words each)
}
static int64_t splat_chunk_images(int32_t n, int32_t h, int32_t w) {
    // one pass unless the caller bounds it (a testing aid) or the fallback accumulator of a pass would pass ~2^31 floats
    int64_t c = n;
    const int64_t cap = ((int64_t)1 << 31) / (5 * (int64_t)h * w);
    if (cap >= 1 && c > cap) c = cap;
    if (g_splat_pass_images > 0 && g_splat_pass_images < c) c = g_splat_pass_images;
    if (c >= n) return n;
    const int64_t passes = (n + c - 1) / c;            // equal passes
    return (n + passes - 1) / passes;
}

__attribute__((visibility("default"))) int64_t ofl_splat_tiled_pass_images(int32_t n, int32_t h, int32_t w) { return splat_chunk_images(n, h, w); }

// images the two-pass fallback accumulator of ofl_splat_tiled_f32 holds: the whole pass while that stays under 1 GiB, else as
// many as fit (at least one) -- flagged images beyond that are served in further rounds of the fallback launches (all of
// them no-ops unless the bin kernel flagged an image), so the footprint of a call no longer grows with the batch
static int64_t splat_fallback_slots(int32_t n, int32_t planes, int32_t h, int32_t w) {
    const int64_t pass = splat_chunk_images(n, h, w);
    const int64_t per_image = (int64_t)planes * h * w * (int64_t)sizeof(float);
    int64_t k = ((int64_t)1 << 30) / (per_image > 0 ? per_image : 1);
    if (g_splat_fallback_slots > 0) k = g_splat_fallback_slots;
    if (k < 1) k = 1;
    return k < pass ? k : pass;
}
__attribute__((visibility("default"))) int64_t ofl_splat_tiled_fallback_images(int32_t n, int32_t planes, int32_t h, int32_t w) {
    return splat_fallback_slots(n, planes, h, w);
}

__attribute__((visibility("default"))) int ofl_splat_tile_geometry(int32_t* tile_w, int32_t* tile_h, int32_t* list_capacity) {
    if (tile_w) *tile_w = kSpTW;
    if (tile_h) *tile_h = kSpTH;
    if (list_capacity) *list_capacity = kBinCap;
    return OFL_OK;
}

__attribute__((visibility("default"))) int64_t ofl_splat_tiled_workspace_ints(int32_t n, int32_t h, int32_t w) {
    return splat_pass_words(splat_chunk_images(n, h, w), h, w);
}

static int splat_tiled_impl(
    const float* flow, int64_t flow_bs, float flow_sign, const float* xs, const float* ys, int64_t xy_bs,
    const float* data, int64_t data_bs, float data_sign, const float* data_b, int64_t data_b_bs,
    const uint8_t* weight_mask, int64_t weight_mask_bs,
    const uint8_t* chan_mask_a, int64_t chan_mask_a_bs, const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
    int32_t with_mask_chan, int32_t occlude, float* dst, float* density, uint8_t* warped, uint8_t* valid,
    float* mask_chan, int32_t* dst_flags, int32_t* workspace, int64_t workspace_ints, float* accum_fallback, int32_t n,
    int32_t c, int32_t h, int32_t w, int32_t round_mode, void* stream, int32_t fh, int32_t fw, int32_t foy, int32_t fox, int elem = 0, int raw = 0) {
    const bool half_in = elem != 0;       // flow and data planes hold fp16 (2 channels, no window, no xs / ys, no data_b, no rounding)
    if (!data || !dst || !workspace || !accum_fallback) return OFL_E_NULL;
    if (half_in && (c != 2 || !flow || xs || ys || data_b || fw != 0 || round_mode != 0)) return OFL_E_UNSUPPORTED;
    if (!flow && !(xs && ys)) return OFL_E_NULL;
    if (flow && !(flow_sign == 1.0f || flow_sign == -1.0f)) return OFL_E_ARG;
    if ((valid || mask_chan) && !with_mask_chan) return OFL_E_ARG;
    if (dst_flags && c != 2) return OFL_E_ARG;
    if (data_b && c > 2) return OFL_E_ARG;                           // (flows: the un-occlude fill only carries it for <= 2 channels)
    if (round_mode < 0 || round_mode > 2) return OFL_E_ARG;
    GatherParams gp = {};
    unsigned grid_unused;
    int rc = fill_splat(gp.s, flow, flow_bs, data, data_bs, data_sign, weight_mask, weight_mask_bs, chan_mask_a,
                        chan_mask_a_bs, chan_mask_b, chan_mask_b_bs, with_mask_chan, occlude, n, c, h, w, grid_unused);
    if (rc) return rc;
    // eligibility of the gather path: at least one whole 4-pixel group per row, 15-bit rows and columns in the record key
    // (any width: 16 / 8-byte accesses at 4-byte alignment, mask bytes at any alignment; > 3 channels in groups of 3)
    const bool ok = w >= 4 && w < 32768 && h < 32768;
    if (!ok) return OFL_E_UNSUPPORTED;
    if (workspace_ints < ofl_splat_tiled_workspace_ints(n, h, w)) return OFL_E_ARG;
    gp.s.flow_sign = flow_sign; gp.s.xs = xs; gp.s.ys = ys; gp.s.xy_bs = xy_bs;
    gp.s.dst = dst; gp.s.density = density; gp.s.warped = warped; gp.s.valid = valid; gp.s.mask_chan = mask_chan;
    gp.s.dst_flags = dst_flags;
    gp.s.data_b = data_b; gp.s.data_b_bs = data_b_bs;
    gp.s.round_mode = round_mode;
    gp.s.fh = fh; gp.s.fw = fw; gp.s.foy = foy; gp.s.fox = fox;
    gp.s.raw = raw;
    gp.tiles_x = (w + kSpTW - 1) / kSpTW; gp.tiles_y = (h + kSpTH - 1) / kSpTH;
    gp.tiles_img = (uint32_t)(gp.tiles_x * gp.tiles_y);
    if ((int64_t)gp.tiles_img * n >= (1ll << 31)) return OFL_E_SHAPE;
    magic_u32((uint32_t)gp.tiles_x, gp.mx_m, gp.mx_s);
    magic_u32(gp.tiles_img, gp.mi_m, gp.mi_s);
    gp.subs_x = (w + kSubW - 1) / kSubW;
    magic_u32((uint32_t)gp.subs_x, gp.sx_m, gp.sx_s);
    gp.regs_x = (w + 4 * kSubW - 1) / (4 * kSubW); gp.regs_y = (h + kRegH - 1) / kRegH;
    gp.regs_img = (uint32_t)(gp.regs_x * gp.regs_y);
    magic_u32((uint32_t)gp.regs_x, gp.rx_m, gp.rx_s);
    magic_u32(gp.regs_img, gp.ri_m, gp.ri_s);
    const int64_t chunk = splat_chunk_images(n, h, w), ctiles = chunk * gp.tiles_img;
    gp.stats = workspace;
    gp.img_over = workspace + 8;
    gp.cnt = gp.img_over + ((chunk + 3) & ~(int64_t)3);
    gp.list = reinterpret_cast<uint32_t*>(gp.cnt + ((ctiles + 3) & ~(int64_t)3));
    gp.redo_cnt = workspace + 6;
    gp.redo_list = gp.list + (int64_t)kBinCap * ctiles;      // (8-byte entries: the lists before it are a multiple of 8 bytes long, the words before them a multiple of 4)
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;                          // (the statistics words are zeroed with the first pass's list lengths)
    if (dst_flags) {
        e = hipMemsetAsync(dst_flags, 0, (size_t)n * sizeof(int32_t), st);
        if (e != hipSuccess) return (int)e;
    }
    const SplatParams all = gp.s;
    const int64_t hw = (int64_t)h * w;
    for (int64_t n0 = 0; n0 < n; n0 += chunk) {          // (same stream: a pass re-uses the lists of the one before)
        const int64_t nn = (n - n0) < chunk ? (n - n0) : chunk;
        // the lists do not depend on the data: binned once per pass, read by every channel group
        SplatParams base = all;
        base.n = (int32_t)nn;
        // (fp16 variants: the flow / data / dst pointers address halves -- advance them in their own element size)
        auto adv_in = [&](const float* ptr, int64_t elems) { return half_in ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(ptr) + elems) : ptr + elems; };
        auto adv_out = [&](float* ptr, int64_t elems) { return elem == 2 ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(ptr) + elems) : ptr + elems; };
        if (base.flow) base.flow = adv_in(all.flow, n0 * all.flow_bs);
        if (base.xs) { base.xs = all.xs + n0 * all.xy_bs; base.ys = all.ys + n0 * all.xy_bs; }
        base.data = adv_in(all.data, n0 * all.data_bs);
        if (base.data_b) base.data_b = all.data_b + n0 * all.data_b_bs;
        if (base.weight_mask) base.weight_mask = all.weight_mask + n0 * all.weight_mask_bs;
        if (base.chan_mask_a) base.chan_mask_a = all.chan_mask_a + n0 * all.chan_mask_a_bs;
        if (base.chan_mask_b) base.chan_mask_b = all.chan_mask_b + n0 * all.chan_mask_b_bs;
        base.dst = adv_out(all.dst, n0 * all.dst_bs);
        if (base.density) base.density = all.density + n0 * hw;
        if (base.warped) base.warped = all.warped + n0 * hw;
        if (base.valid) base.valid = all.valid + n0 * hw;
        if (base.mask_chan) base.mask_chan = all.mask_chan + n0 * hw;
        if (base.dst_flags) base.dst_flags = all.dst_flags + n0;
        gp.s = base;
        gp.total = (int64_t)gp.tiles_img * nn;
        gp.per_xcd = (gp.total + kXcds - 1) / kXcds;
        gp.rtotal = (int64_t)gp.regs_img * nn;
        gp.rper_xcd = (gp.rtotal + kXcds - 1) / kXcds;
        // per-image fallback flags and list lengths of this pass
        const size_t zwords = (size_t)(((chunk + 3) & ~(int64_t)3) + ctiles);
        e = n0 == 0 ? hipMemsetAsync(gp.stats, 0, (8 + zwords) * sizeof(int32_t), st)      // statistics | flags | lengths: contiguous
                    : hipMemsetAsync(gp.redo_cnt, 0, (2 + zwords) * sizeof(int32_t), st);  // (the redo list's length sits right before the flags)
        if (e != hipSuccess) return (int)e;
        const bool lean = splat_is_lean(gp.s);
        if (half_in) { if (lean) OFL_KLAUNCH((splat_bin_kernel<_Float16, true>), dim3((unsigned)(gp.rper_xcd * kXcds)), dim3(256), 0, st, gp);
                       else OFL_KLAUNCH((splat_bin_kernel<_Float16>), dim3((unsigned)(gp.rper_xcd * kXcds)), dim3(256), 0, st, gp); }
        else { if (lean) OFL_KLAUNCH((splat_bin_kernel<float, true>), dim3((unsigned)(gp.rper_xcd * kXcds)), dim3(256), 0, st, gp);
               else OFL_KLAUNCH((splat_bin_kernel<float>), dim3((unsigned)(gp.rper_xcd * kXcds)), dim3(256), 0, st, gp); }
        rc = (int)hipGetLastError();
        if (rc) return rc;
        // more than 3 channels: groups of 3 (a record holds 3 data channels); density and masks come out of the first group
        for (int32_t c0 = 0; c0 < c; c0 += 3) {
            SplatParams full = base;
            full.c = (c - c0) < 3 ? (c - c0) : 3;
            full.data = base.data + c0 * hw; full.dst = base.dst + c0 * hw;
            if (base.data_b) full.data_b = base.data_b + c0 * hw;
            if (c0 > 0) { full.with_mask_chan = 0; full.density = nullptr; full.warped = nullptr; full.valid = nullptr; full.mask_chan = nullptr; }
            const int32_t cg = full.c;
            gp.s = full;
            if (c0 > 0) {                                     // (every channel group fills the redo list anew)
                e = hipMemsetAsync(gp.redo_cnt, 0, 2 * sizeof(int32_t), st);     // (its length and the second launch's hand-out counter)
                if (e != hipSuccess) return (int)e;
            }
            const unsigned grid = (unsigned)(gp.per_xcd * kXcds);
            if (half_in) rc = launch_splat_gather_half(gp, grid, st, elem);
            else switch (cg) {
                case 1: rc = launch_splat_gather<1>(gp, grid, st); break;
                case 2: rc = launch_splat_gather<2>(gp, grid, st); break;
                default: rc = launch_splat_gather<3>(gp, grid, st); break;
            }
            if (rc) return rc;
            // two-pass global-atomics path for the images of this pass that were flagged (a list overflowed / a subtile
            // spread too wide); every block below exits at once for the others
            SplatParams fb = gp.s;
            fb.accum = accum_fallback;
            fb.run_if_set = gp.img_over;
            fb.any_set = gp.stats;                            // stats[0]: some image (of this or an earlier pass) is flagged
            // the accumulator holds `slots` images (sized for the planes of the call's first channel group, the largest):
            // flagged images are served `slots` at a time; every launch below leaves at once when nothing is flagged
            fb.fb_slots = (int32_t)splat_fallback_slots(n, 1 + (c < 3 ? c : 3) + (with_mask_chan ? 1 : 0), h, w);
            const int planes = 1 + cg + (fb.with_mask_chan ? 1 : 0);
            unsigned g2;
            tile_grid((int32_t)nn, h, w, fb.tiles_x, fb.tiles_y, fb.total_tiles, fb.per_xcd, g2);
            // (strided: the kernel walks the tiles of the flagged images; its passes meet at grid barriers, so the grid must be
            // RESIDENT AS A WHOLE: bounded by what the device -- a CPX partition, a CU-masked stream's device -- can hold, with
            // half of it left to whatever else runs, a second call's fallback kernel on another stream included)
            const int which = half_in ? (elem == 2 ? 3 : 4) : (cg == 1 ? 0 : (cg == 2 ? 1 : 2));
            const void* kfn = half_in ? (elem == 2 ? (const void*)splat_fallback_kernel<2, _Float16, _Float16> : (const void*)splat_fallback_kernel<2, _Float16, float>)
                                      : (cg == 1 ? (const void*)splat_fallback_kernel<1> : (cg == 2 ? (const void*)splat_fallback_kernel<2> : (const void*)splat_fallback_kernel<3>));
            const unsigned resident = fallback_resident_blocks(kfn, which);
            if (g2 > resident) g2 = resident;
            for (int64_t r0 = 0; r0 < nn; r0 += fb.fb_slots) {
                fb.fb_round = (int32_t)(r0 / fb.fb_slots);
                int32_t* arrivals = gp.stats + 4;           // two words, zeroed with the statistics words at the start of the call and left zero by every launch
                const int64_t cpi = (int64_t)planes * hw;
                if (half_in) {
                    if (elem == 2) OFL_KLAUNCH((splat_fallback_kernel<2, _Float16, _Float16>), dim3(g2), dim3(256), 0, st, fb, accum_fallback, cpi, (int32_t)nn, arrivals);
                    else OFL_KLAUNCH((splat_fallback_kernel<2, _Float16, float>), dim3(g2), dim3(256), 0, st, fb, accum_fallback, cpi, (int32_t)nn, arrivals);
                } else switch (cg) {
                    case 1: OFL_KLAUNCH(splat_fallback_kernel<1>, dim3(g2), dim3(256), 0, st, fb, accum_fallback, cpi, (int32_t)nn, arrivals); break;
                    case 2: OFL_KLAUNCH(splat_fallback_kernel<2>, dim3(g2), dim3(256), 0, st, fb, accum_fallback, cpi, (int32_t)nn, arrivals); break;
                    default: OFL_KLAUNCH(splat_fallback_kernel<3>, dim3(g2), dim3(256), 0, st, fb, accum_fallback, cpi, (int32_t)nn, arrivals); break;
                }
            }
        }
    }
    return (int)hipGetLastError();
}

__attribute__((visibility("default"))) int ofl_splat_tiled_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* xs, const float* ys, int64_t xy_bs,
    const float* data, int64_t data_bs, float data_sign, const float* data_b, int64_t data_b_bs,
    const uint8_t* weight_mask, int64_t weight_mask_bs,
    const uint8_t* chan_mask_a, int64_t chan_mask_a_bs, const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
    int32_t with_mask_chan, int32_t occlude, float* dst, float* density, uint8_t* warped, uint8_t* valid,
    float* mask_chan, int32_t* dst_flags, int32_t* workspace, int64_t workspace_ints, float* accum_fallback, int32_t n,
    int32_t c, int32_t h, int32_t w, int32_t round_mode, void* stream) {
    return splat_tiled_impl(flow, flow_bs, flow_sign, xs, ys, xy_bs, data, data_bs, data_sign, data_b, data_b_bs, weight_mask,
                            weight_mask_bs, chan_mask_a, chan_mask_a_bs, chan_mask_b, chan_mask_b_bs, with_mask_chan, occlude, dst,
                            density, warped, valid, mask_chan, dst_flags, workspace, workspace_ints, accum_fallback, n, c, h, w,
                            round_mode, stream, 0, 0, 0, 0);
}

__attribute__((visibility("default"))) int ofl_splat_tiled_f16(
    const void* flow_f16, int64_t flow_bs, float flow_sign, const void* data_f16, int64_t data_bs, float data_sign,
    const uint8_t* weight_mask, int64_t weight_mask_bs,
    const uint8_t* chan_mask_a, int64_t chan_mask_a_bs, const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
    int32_t with_mask_chan, int32_t occlude, void* dst, int32_t dst_is_f16, uint8_t* valid, int32_t* dst_flags,
    int32_t* workspace, int64_t workspace_ints, float* accum_fallback, int32_t n, int32_t h, int32_t w, void* stream) {
    if (!flow_f16 || !data_f16) return OFL_E_NULL;
    if (!aligned_to(flow_f16, 2) || !aligned_to(data_f16, 2)) return OFL_E_ARG;
    return splat_tiled_impl(static_cast<const float*>(flow_f16), flow_bs, flow_sign, nullptr, nullptr, 0,
                            static_cast<const float*>(data_f16), data_bs, data_sign, nullptr, 0, weight_mask, weight_mask_bs,
                            chan_mask_a, chan_mask_a_bs, chan_mask_b, chan_mask_b_bs, with_mask_chan, occlude,
                            static_cast<float*>(dst), nullptr, nullptr, valid, nullptr, dst_flags, workspace, workspace_ints,
                            accum_fallback, n, 2, h, w, 0, stream, 0, 0, 0, 0, dst_is_f16 ? 2 : 1);
}

__attribute__((visibility("default"))) int ofl_splat_tiled_win_f32(
    const float* flow, int64_t flow_bs, float flow_sign, int32_t fh, int32_t fw, int32_t foy, int32_t fox,
    const float* data, int64_t data_bs, float data_sign,
    const uint8_t* weight_mask, int64_t weight_mask_bs,
    const uint8_t* chan_mask_a, int64_t chan_mask_a_bs, const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
    int32_t with_mask_chan, int32_t occlude, float* dst, float* density, uint8_t* warped, uint8_t* valid,
    float* mask_chan, int32_t* workspace, int64_t workspace_ints, float* accum_fallback, int32_t n,
    int32_t c, int32_t h, int32_t w, int32_t round_mode, void* stream) {
    if (!flow) return OFL_E_NULL;
    if (fh < 1 || fw < 1 || foy < 0 || fox < 0 || foy + fh > h || fox + fw > w) return OFL_E_ARG;
    return splat_tiled_impl(flow, flow_bs, flow_sign, nullptr, nullptr, 0, data, data_bs, data_sign, nullptr, 0, weight_mask,
                            weight_mask_bs, chan_mask_a, chan_mask_a_bs, chan_mask_b, chan_mask_b_bs, with_mask_chan, occlude, dst,
                            density, warped, valid, mask_chan, nullptr, workspace, workspace_ints, accum_fallback, n, c, h, w,
                            round_mode, stream, fh, fw, foy, fox);
}

// the weighted sums of a forward splat, NOT normalised: dst[n,c,q] = sum over source pixels i and corners k landing on q of
// w_ik * data_sign * data[n,c,i] -- the transpose of the backward warp, i.e. the gradient of ofl_warp_bwd_f32 with respect to its
// source when `data` is the upstream gradient and flow_sign the NEGATED sign of the warp (same kernels as ofl_splat_tiled_f32:
// no float atomics outside fold tiles, deterministic there)
__attribute__((visibility("default"))) int ofl_splat_sum_f32(
    const float* flow, int64_t flow_bs, float flow_sign, const float* data, int64_t data_bs, float data_sign, float* dst,
    int32_t* workspace, int64_t workspace_ints, float* accum_fallback, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
    if (!flow) return OFL_E_NULL;
    return splat_tiled_impl(flow, flow_bs, flow_sign, nullptr, nullptr, 0, data, data_bs, data_sign, nullptr, 0, nullptr, 0, nullptr, 0,
                            nullptr, 0, 0, 0, dst, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_ints,
                            accum_fallback, n, c, h, w, 0, stream, 0, 0, 0, 0, 0, 1);
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
    if (!src_f16 || !flags) return OFL_E_NULL;                    // (dst NULL: flags only -- the flow stays in fp16)
    int rc = check_dims(n, 2, h, w, false);
    if (rc) return rc;
    if (n > 65535) return OFL_E_SHAPE;
    const int64_t hw = (int64_t)h * w;
    if ((hw % 4) != 0 || !aligned_to(src_f16, 8) || (src_bs % 4) != 0 || (dst && !aligned_to(dst, 16)) ||
        (mask && (!aligned_to(mask, 4) || (mask_bs % 4) != 0)))
        return OFL_E_UNSUPPORTED;
    int64_t bx = (hw / 4 + 1023) / 1024, cap = 512 / n;
    cap = cap < 16 ? 16 : (cap > 256 ? 256 : cap);
    if (bx > cap) bx = cap;
    OFL_KLAUNCH(flow_f16_kernel<false>, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const _Float16*>(src_f16), src_bs, mask, mask_bs, dst, flags, hw, FlagsHost{});
    return (int)hipGetLastError();
}

// the validation read-back without a copy launch or an event (flow_class.py / utils.py:98: the constructor must know whether
// the vectors are finite before it returns): the reduction's last block writes the words to host-visible memory itself
__attribute__((visibility("default"))) int ofl_flow_flags_host(const void* flow, int32_t flow_is_f16, int64_t flow_bs,
                                                               const uint8_t* mask, int64_t mask_bs, float thr,
                                                               int32_t* work, int32_t* host_words, int32_t serial,
                                                               int32_t n, int32_t h, int32_t w, void* stream) {
    if (!flow || !work || !host_words) return OFL_E_NULL;
    int rc = check_dims(n, 2, h, w, false);
    if (rc) return rc;
    if (n > 65535) return OFL_E_SHAPE;
    if (thr != kZeroThr) return OFL_E_ARG;
    const int64_t hw = (int64_t)h * w;
    if ((reinterpret_cast<uintptr_t>(host_words) & 7) != 0) return OFL_E_ARG;
    FlagsHost fh = {};
    fh.counters = work + n; fh.host = reinterpret_cast<unsigned long long*>(host_words); fh.serial = serial; fh.n = n;
    hipStream_t st = (hipStream_t)stream;
    if (!flow_is_f16) {
        launch_flow_flags(static_cast<const float*>(flow), flow_bs, mask, mask_bs, work, n, hw, st, &fh);
        return (int)hipGetLastError();
    }
    if ((hw % 4) != 0 || !aligned_to(flow, 8) || (flow_bs % 4) != 0 || (mask && (!aligned_to(mask, 4) || (mask_bs % 4) != 0)))
        return OFL_E_UNSUPPORTED;
    int64_t bx = (hw / 4 + 1023) / 1024, cap = 512 / n;
    cap = cap < 16 ? 16 : (cap > 256 ? 256 : cap);
    if (bx > cap) bx = cap;
    fh.total_blocks = (int32_t)(bx * n);
    OFL_KLAUNCH(flow_f16_kernel<true>, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st,
                       static_cast<const _Float16*>(flow), flow_bs, mask, mask_bs, (float*)nullptr, work, hw, fh);
    return (int)hipGetLastError();
}

// host-visible (pinned, coherent, device-mapped) words for ofl_flow_flags_host
__attribute__((visibility("default"))) int ofl_host_words_alloc(int64_t ints, void** out) {
    if (!out || ints < 1) return OFL_E_ARG;
    void* ptr = nullptr;
    hipError_t e = hipHostMalloc(&ptr, (size_t)ints * sizeof(int32_t), hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess) return (int)e;
    for (int64_t i = 0; i < ints; ++i) static_cast<volatile int32_t*>(ptr)[i] = 0;
    *out = ptr;
    return OFL_OK;
}
__attribute__((visibility("default"))) int ofl_host_words_free(void* ptr) { return ptr ? (int)hipHostFree(ptr) : OFL_OK; }

}  // extern "C"
#endif  // OFL_WIDE_TU
