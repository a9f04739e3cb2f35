// Microbenchmark 3: what bounds a streaming kernel with the byte mix of Flow.apply (35 B/px: 22 read, 13 written, ten
// planes)?  Tile shape of a block, shape of one wave inside it, block -> tile order, persistent vs one tile per block.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ss tools/microbench/stream_shapes.hip && /tmp/ss [N]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct P {
    const float* flow; const float* src; const uint8_t* smask; const uint8_t* fmask;
    float* dst; uint8_t* valid;
    int n, h, w, tiles_x, tiles_y; long total, per_xcd;
};

// ORDER 0: XCD-contiguous ranges (block b -> XCD b & 7 owns [k * per_xcd, (k+1) * per_xcd)); 1: plain (tile = blockIdx)
template <int ORDER>
__device__ __forceinline__ long tile_of(const P& p, long b) {
    if (ORDER == 0) return (b & 7) * p.per_xcd + (b >> 3);
    return b;
}

// TW x TH pixels per 256-thread block (TW * TH == 1024, 4 px per thread);  WAVEW: pixels one wave spans in x (32 or 64 or TW)
template <int TW, int WAVEW, int ORDER>
__global__ __launch_bounds__(256) void stream_k(const P p) {
    const long tile = tile_of<ORDER>(p, blockIdx.x);
    if (tile >= p.total) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / ((long)p.tiles_x * p.tiles_y);
    constexpr int TH = 1024 / TW, WL = WAVEW / 4, WROWS = 64 / WL, WPR = TW / WAVEW;   // lanes per wave row, rows per wave, waves per tile row
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lx = (lane % WL) + (wv % WPR) * WL, ly = (lane / WL) + (wv / WPR) * WROWS;
    static_assert(TH * TW == 1024 && (4 / WPR) * WROWS == TH, "shape");
    const int w = p.w, h = p.h, hw = h * w;
    const int x4 = tx * TW + lx * 4, y = ty * TH + ly;
    if (x4 >= w || y >= h) return;
    const int pix = y * w + x4;
    const float4 u = *reinterpret_cast<const float4*>(p.flow + (long)n * 2 * hw + pix);
    const float4 v = *reinterpret_cast<const float4*>(p.flow + (long)n * 2 * hw + hw + pix);
    const uchar4 a = *reinterpret_cast<const uchar4*>(p.smask + (long)n * hw + pix);
    const uchar4 b = *reinterpret_cast<const uchar4*>(p.fmask + (long)n * hw + pix);
    *reinterpret_cast<uchar4*>(p.valid + (long)n * hw + pix) = make_uchar4(a.x && b.x && (v.x != 12345.f), a.y && b.y, a.z && b.z, a.w && b.w);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float4 q = *reinterpret_cast<const float4*>(p.src + (long)n * 3 * hw + c * hw + pix);
        q.x += u.x; q.y += u.y; q.z += u.z; q.w += u.w;
        *reinterpret_cast<float4*>(p.dst + (long)n * 3 * hw + c * hw + pix) = q;
    }
}

// persistent: G blocks, block b walks tiles of its XCD range with stride G/8; loads of tile i+1 issued before the stores of tile i
template <int TW, int WAVEW>
__global__ __launch_bounds__(256) void stream_p(const P p) {
    constexpr int TH = 1024 / TW, WL = WAVEW / 4, WROWS = 64 / WL, WPR = TW / WAVEW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lx = (lane % WL) + (wv % WPR) * WL, ly = (lane / WL) + (wv / WPR) * WROWS;
    const int w = p.w, h = p.h, hw = h * w;
    const unsigned slots = gridDim.x >> 3;
    float4 u, v, q[3]; uchar4 a, b; long base = 0; int pix = 0; bool have = false;
    auto load = [&](unsigned it) {
        const unsigned k = (blockIdx.x >> 3) + it * slots;
        const long tile = (blockIdx.x & 7) * p.per_xcd + k;
        have = (k < (unsigned)p.per_xcd) && (tile < p.total);
        if (!have) return;
        const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / ((long)p.tiles_x * p.tiles_y);
        const int x4 = min(tx * TW + lx * 4, w - 4), y = min(ty * TH + ly, h - 1);
        pix = y * w + x4; base = (long)n * hw;
        u = *reinterpret_cast<const float4*>(p.flow + 2 * base + pix);
        v = *reinterpret_cast<const float4*>(p.flow + 2 * base + hw + pix);
        a = *reinterpret_cast<const uchar4*>(p.smask + base + pix);
        b = *reinterpret_cast<const uchar4*>(p.fmask + base + pix);
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = *reinterpret_cast<const float4*>(p.src + 3 * base + c * hw + pix);
    };
    load(0);
    for (unsigned it = 0; have; ++it) {
        const float4 u0 = u, v0 = v; const uchar4 a0 = a, b0 = b; float4 q0[3] = {q[0], q[1], q[2]}; const long base0 = base; const int pix0 = pix;
        load(it + 1);
        *reinterpret_cast<uchar4*>(p.valid + base0 + pix0) = make_uchar4(a0.x && b0.x && (v0.x != 12345.f), a0.y && b0.y, a0.z && b0.z, a0.w && b0.w);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            q0[c].x += u0.x; q0[c].y += u0.y; q0[c].z += u0.z; q0[c].w += u0.w;
            *reinterpret_cast<float4*>(p.dst + 3 * base0 + c * hw + pix0) = q0[c];
        }
    }
}

// read-only and write-only halves of the same mix
template <int TW, int WAVEW>
__global__ __launch_bounds__(256) void stream_ro(const P p) {
    const long tile = tile_of<0>(p, blockIdx.x);
    if (tile >= p.total) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / ((long)p.tiles_x * p.tiles_y);
    constexpr int TH = 1024 / TW, WL = WAVEW / 4, WROWS = 64 / WL, WPR = TW / WAVEW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lx = (lane % WL) + (wv % WPR) * WL, ly = (lane / WL) + (wv / WPR) * WROWS;
    const int w = p.w, h = p.h, hw = h * w;
    const int x4 = tx * TW + lx * 4, y = ty * TH + ly;
    if (x4 >= w || y >= h) return;
    const int pix = y * w + x4;
    const float4 u = *reinterpret_cast<const float4*>(p.flow + (long)n * 2 * hw + pix);
    const float4 v = *reinterpret_cast<const float4*>(p.flow + (long)n * 2 * hw + hw + pix);
    const uchar4 a = *reinterpret_cast<const uchar4*>(p.smask + (long)n * hw + pix);
    const uchar4 b = *reinterpret_cast<const uchar4*>(p.fmask + (long)n * hw + pix);
    float s = u.x + v.y + a.x + b.y;
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float4 q = *reinterpret_cast<const float4*>(p.src + (long)n * 3 * hw + c * hw + pix); s += q.x + q.w; }
    if (s == 1.2345e-30f) p.valid[0] = 1;
}
template <int TW, int WAVEW>
__global__ __launch_bounds__(256) void stream_wo(const P p) {
    const long tile = tile_of<0>(p, blockIdx.x);
    if (tile >= p.total) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / ((long)p.tiles_x * p.tiles_y);
    constexpr int TH = 1024 / TW, WL = WAVEW / 4, WROWS = 64 / WL, WPR = TW / WAVEW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lx = (lane % WL) + (wv % WPR) * WL, ly = (lane / WL) + (wv / WPR) * WROWS;
    const int w = p.w, h = p.h, hw = h * w;
    const int x4 = tx * TW + lx * 4, y = ty * TH + ly;
    if (x4 >= w || y >= h) return;
    const int pix = y * w + x4;
    *reinterpret_cast<uchar4*>(p.valid + (long)n * hw + pix) = make_uchar4(1, 0, 1, 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(p.dst + (long)n * 3 * hw + c * hw + pix) = make_float4(1.f, 2.f, 3.f, (float)pix);
}


// reads of the full mix (22 B/px) but only the mask (MODE 1) or the mask + one channel (MODE 2) stored
template <int TW, int WAVEW, int MODE>
__global__ __launch_bounds__(256) void stream_partial(const P p) {
    const long tile = tile_of<0>(p, blockIdx.x);
    if (tile >= p.total) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / ((long)p.tiles_x * p.tiles_y);
    constexpr int TH = 1024 / TW, WL = WAVEW / 4, WROWS = 64 / WL, WPR = TW / WAVEW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lx = (lane % WL) + (wv % WPR) * WL, ly = (lane / WL) + (wv / WPR) * WROWS;
    const int w = p.w, h = p.h, hw = h * w;
    const int x4 = tx * TW + lx * 4, y = ty * TH + ly;
    if (x4 >= w || y >= h) return;
    const int pix = y * w + x4;
    const float4 u = *reinterpret_cast<const float4*>(p.flow + (long)n * 2 * hw + pix);
    const float4 v = *reinterpret_cast<const float4*>(p.flow + (long)n * 2 * hw + hw + pix);
    const uchar4 a = *reinterpret_cast<const uchar4*>(p.smask + (long)n * hw + pix);
    const uchar4 b = *reinterpret_cast<const uchar4*>(p.fmask + (long)n * hw + pix);
    float4 q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = *reinterpret_cast<const float4*>(p.src + (long)n * 3 * hw + c * hw + pix);
    const float s = q[0].x + q[1].y + q[2].z + u.x;
    *reinterpret_cast<uchar4*>(p.valid + (long)n * hw + pix) = make_uchar4(a.x && b.x && (v.x != 12345.f) && (s != 1.2345e-30f), a.y && b.y, a.z && b.z, a.w && b.w);
    if (MODE == 2) { q[0].x += u.x; *reinterpret_cast<float4*>(p.dst + (long)n * 3 * hw + pix) = q[0]; }
}

template <typename F>
static float time_it(F launch, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 64, h = 1080, w = 1920;
    const size_t hw = (size_t)h * w, px = (size_t)n * hw;
    P p; memset(&p, 0, sizeof(p));
    float *dflow, *dsrc, *ddst; uint8_t *dsm, *dfm, *dval;
    CK(hipMalloc(&dflow, px * 8)); CK(hipMalloc(&dsrc, px * 12)); CK(hipMalloc(&ddst, px * 12));
    CK(hipMalloc(&dsm, px)); CK(hipMalloc(&dfm, px)); CK(hipMalloc(&dval, px));
    CK(hipMemset(dflow, 0, px * 8)); CK(hipMemset(dsrc, 0, px * 12)); CK(hipMemset(dsm, 1, px)); CK(hipMemset(dfm, 1, px));
    p.flow = dflow; p.src = dsrc; p.smask = dsm; p.fmask = dfm; p.dst = ddst; p.valid = dval; p.n = n; p.h = h; p.w = w;
    auto grid_for = [&](int tw, int th) {
        p.tiles_x = (w + tw - 1) / tw; p.tiles_y = (h + th - 1) / th; p.total = (long)p.tiles_x * p.tiles_y * n;
        p.per_xcd = (p.total + 7) / 8;
        return (unsigned)(p.per_xcd * 8);
    };
    auto report = [&](const char* name, float ms, double bpp) {
        printf("%-52s %8.3f ms  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms, bpp * px / ms / 1e6, bpp * px / ms / 1e6 / 80.0); fflush(stdout);
    };
    const int it = 20;
#define RUNK(name, TW, WAVEW, ORDER) { unsigned g = grid_for(TW, 1024 / TW); \
        report(name, time_it([&] { hipLaunchKernelGGL((stream_k<TW, WAVEW, ORDER>), dim3(g), dim3(256), 0, 0, p); }, it), 35.0); }
    RUNK("tile 32x32, wave 32x8, xcd order", 32, 32, 0)
    RUNK("tile 32x32, wave 32x8, plain order", 32, 32, 1)
    RUNK("tile 64x16, wave 64x4, xcd order", 64, 64, 0)
    RUNK("tile 64x16, wave 32x8 (2 side by side), xcd", 64, 32, 0)
    RUNK("tile 64x16, wave 64x4, plain order", 64, 64, 1)
    RUNK("tile 128x8, wave 128x2, xcd order", 128, 128, 0)
    RUNK("tile 128x8, wave 64x4 (2 side by side), xcd", 128, 64, 0)
    RUNK("tile 128x8, wave 32x8 (4 side by side), xcd", 128, 32, 0)
    RUNK("tile 128x8, wave 128x2, plain order", 128, 128, 1)
    RUNK("tile 256x4, wave 256x1, xcd order", 256, 256, 0)
#define RUNP(name, TW, WAVEW, G) { grid_for(TW, 1024 / TW); \
        report(name, time_it([&] { hipLaunchKernelGGL((stream_p<TW, WAVEW>), dim3(G), dim3(256), 0, 0, p); }, it), 35.0); }
    RUNP("persistent 32x32 wave 32x8 g2048", 32, 32, 2048)
    RUNP("persistent 32x32 wave 32x8 g1024", 32, 32, 1024)
    RUNP("persistent 64x16 wave 64x4 g2048", 64, 64, 2048)
    RUNP("persistent 128x8 wave 128x2 g2048", 128, 128, 2048)
    RUNP("persistent 128x8 wave 128x2 g1024", 128, 128, 1024)
    RUNP("persistent 128x8 wave 32x8 g2048", 128, 32, 2048)
#define RUNRO(name, TW, WAVEW) { unsigned g = grid_for(TW, 1024 / TW); \
        report(name " read-only 22 B/px", time_it([&] { hipLaunchKernelGGL((stream_ro<TW, WAVEW>), dim3(g), dim3(256), 0, 0, p); }, it), 22.0); \
        report(name " write-only 13 B/px", time_it([&] { hipLaunchKernelGGL((stream_wo<TW, WAVEW>), dim3(g), dim3(256), 0, 0, p); }, it), 13.0); }
#define RUNPT(name, TW, WAVEW, MODE, BPP) { unsigned g = grid_for(TW, 1024 / TW); \
        report(name, time_it([&] { hipLaunchKernelGGL((stream_partial<TW, WAVEW, MODE>), dim3(g), dim3(256), 0, 0, p); }, it), BPP); }
    RUNPT("tile 32x32: read 22 B/px, store mask only (23 B/px)", 32, 32, 1, 23.0)
    RUNPT("tile 32x32: read 22 B/px, store mask + 1 channel (27 B/px)", 32, 32, 2, 27.0)
    RUNPT("tile 128x8: read 22 B/px, store mask only (23 B/px)", 128, 128, 1, 23.0)
    RUNRO("tile 32x32 wave 32x8", 32, 32)
    RUNRO("tile 128x8 wave 128x2", 128, 128)
    return 0;
}
