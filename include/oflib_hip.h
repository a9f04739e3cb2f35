/*
 * oflib_hip.h -- C ABI of libofl_hip.so: the MI355X (gfx950) kernels behind the dense
 * flow warp / compose hot path of oflibpytorch.
 *
 * The reference (oflibpytorch v2.1.1) is pure Python and has no FFI layer of its own; the
 * arithmetic of this path lives in two ATen call sites.  The entry points below are what a
 * binding for exactly that path replaces:
 *
 *   ofl_warp_bwd_f32        <- F.grid_sample(target, normalise_coords(grid - flow))
 *                              src/oflibpytorch/utils.py:541-555 (+ normalise_coords :445-466),
 *                              with the mask channel / threshold / AND of Flow.apply
 *                              (flow_class.py:896-898, 921-934) and the vector add of
 *                              combine_with mode 3 (flow_class.py:1804, 1808; __add__ :450-488)
 *                              fused in.
 *   ofl_splat_fwd_f32 +     <- density.scatter_add_ / grid_data.scatter_add_ and the normalise
 *   ofl_splat_finalize_f32     pass of grid_from_unstructured_data (utils.py:1061-1154), with
 *                              get_flow_endpoints (:1045-1058), the zero-flow occlusion rule and
 *                              un-occlude fill of apply_s_flow (:1157-1205) and Flow.apply's mask
 *                              channel (flow_class.py:880-898, 921-923) fused in.
 *   ofl_flow_flags_f32      <- get_valid_vecs' isfinite().all() (utils.py:98), threshold_vectors
 *                              (:623-643), is_zero_flow (:919-938), Flow.is_zero
 *                              (flow_class.py:1226-1244): the data-dependent early-exit tests.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP); nothing is allocated or freed inside;
 *   - images are planar N-C-H-W, rows contiguous (stride_w = 1, stride_h = W, channel stride
 *     = H*W); the batch stride of every input is explicit, in ELEMENTS, and may be 0 to
 *     broadcast one batch element (the reference's 1<->N `expand`, utils.py:527-537);
 *   - masks are 1 byte per pixel, 0 = False, anything else = True (torch.bool storage);
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it;
 *   - return value: 0 on success, OFL_E_* (< 0) for rejected arguments, or a positive
 *     hipError_t from the launch.
 *   - all arithmetic is IEEE fp32 in the reference's operation order (no fast-math, no
 *     implicit contraction); masks are bit-exact with the reference's PyTorch-CPU path.
 */
#ifndef OFLIB_HIP_H
#define OFLIB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFL_OK 0
#define OFL_E_NULL -1      /* a required pointer is NULL */
#define OFL_E_SHAPE -2     /* n, c, h or w out of range (all must be >= 1; h*w < 2^24, utils.py:1118) */
#define OFL_E_ARG -3       /* inconsistent optional arguments */
#define OFL_E_UNSUPPORTED -4 /* this entry point cannot take the launch (alignment / channel count): use the general one */

/* library / build identification: returns e.g. 11 for 0.1.1 */
int ofl_version(void);

/* Process-wide options (testing / benchmarking aids; defaults are what production uses).
 *   OFL_OPT_WARP_PATH: 0 = auto (LDS-staged kernel when the launch is eligible: W >= 4, H >= 2; > 3 channels in ONE launch that loops over them;
 *                          generic direct-gather kernel otherwise -- both restate the same arithmetic),
 *                      1 = generic direct-gather kernel only,
 *                      3 / 4 = like auto, but the staged kernel with two tiles / one tile per block whatever the launch size
 *                          (auto picks by size: one tile per block for tiny launches, two for small ones, columns of four
 *                          from ~B = 8 at 1080p; all three run the same device code per tile -- tests compare them),
 *                      5 = like auto, but a warp of more than 3 channels runs as separate launches of 3 channels instead of the
 *                          channel-loop kernel (one launch that walks the channels in groups of 4 inside the block; W % 4 == 0),
 *                      6 = like auto, but the four-tile column kernel of large launches stages ONE y-sheared rectangle per tile instead of
 *                          per-row extents (the default for plain warps of 1..3 channels with W % 4 == 0 and no rounding: every source row
 *                          a 64 x 16 tile touches has its own first chunk and length; small launches run it with 1 or 2 tiles per block),
 *                      7 = like auto, but plain lean launches take the four-tile row-table columns whatever their size (tests). */
#define OFL_OPT_WARP_PATH 1
/*   OFL_OPT_WARP_SHEAR: 1 = the LDS-staged warp kernel stages a y-sheared box (default), 0 = plain bounding box
 *   (speed only; the results are identical). */
#define OFL_OPT_WARP_SHEAR 3
/*   OFL_OPT_SPLAT_PASS_IMAGES: upper bound on the images ofl_splat_tiled_f32 handles per pass (0 = automatic: one pass
 *   unless the fallback accumulator of a pass would pass 2^31 floats; tests use small values to exercise the multi-pass
 *   code on small inputs). */
#define OFL_OPT_SPLAT_PASS_IMAGES 4
/*   OFL_OPT_SPLAT_FALLBACK_SLOTS: images the two-pass fallback accumulator of ofl_splat_tiled_f32 holds (0 = automatic: the
 *   pass, capped at 1 GiB); tests use 1 to exercise the rounds.
 *   All options are PROCESS-GLOBAL, unsynchronised testing / benchmarking aids: set them before any concurrent use of the
 *   library, never from two threads; production code leaves them alone. */
#define OFL_OPT_SPLAT_FALLBACK_SLOTS 5
/*   OFL_OPT_SPLAT_PATH: gather kernel of ofl_splat_tiled_f32 and its relatives: 0 = the round-6 kernel (16-24-byte records,
 *   one-word cells: three 512-thread blocks per CU), 1 = round 5's (32-byte records, two blocks per CU).  Same sums in the same
 *   order: bit-identical results (the tests compare them); speed only. */
#define OFL_OPT_SPLAT_PATH 6
/*   OFL_OPT_SPLAT_EXTRA_LDS: bytes of dynamic LDS added to the round-6 gather kernel's launches (0 = none) -- a measuring aid: it
 *   lowers the blocks a CU holds (28 672: two, 65 536: one) without changing a single instruction (tools/splat_occupancy.py). */
#define OFL_OPT_SPLAT_EXTRA_LDS 7
int ofl_set_option(int32_t key, int32_t value);
/* the (mangled) name of the kernel the library launched LAST, as the loaded code object spells it ("" before the first launch) --
 * a diagnostic: bench.py reports the instantiation the launchers actually picked for its roofline kernel.  Process-global, like
 * the options; the pointer stays valid while the library is loaded. */
const char* ofl_last_kernel_name(void);

/* rounding applied to the warped channels before the store (apply_flow utils.py:613-618,
 * Flow.apply flow_class.py:943-946): 0 none, 1 round-half-even, 2 round then clamp to [0,255] */
#define OFL_ROUND_NONE 0
#define OFL_ROUND_RINT 1
#define OFL_ROUND_U8 2

/*
 * Backward ("t"-reference) bilinear warp, zero padding, align_corners=True.
 *
 *   p        = (gx, gy) - flow_sign * flow[n]                       (fp32)
 *   G[n,c]   = bilinear(src[n,c], p)                                 (taps outside the image = 0)
 *   dst[n,c] = addend ? a_sign * addend[n,c] + g_sign * G[n,c] : G[n,c]      (then `round_mode`)
 *   valid[n] = (bilinear(src_mask ? src_mask[n] : 1, p) > 0.99999f) & (flow_mask ? flow_mask[n] : 1)
 *
 * flow      [*,2,H,W] fp32, flow_sign = +1 or -1 (exact negation; Flow.invert('t') of an 's' flow)
 * src       [*,C,H,W] fp32
 * src_b     [*,C,H,W] fp32 or NULL: the field gathered is src - src_b (ONE fp32 subtraction per value: the reference's
 *           `flow - self` in combine_with mode 1, flow_class.py:1763).  Only for C == 2 with `valid` wanted, no addend and
 *           no flag outputs on frames the staged kernel takes (W >= 4, H >= 2); otherwise OFL_E_UNSUPPORTED and the
 *           caller passes the materialised difference
 * src_mask  [*,H,W] u8 or NULL (all True)      -- the mask warped as an extra channel
 * flow_mask [*,H,W] u8 or NULL (all True)      -- ANDed after the warp
 * addend    [*,C,H,W] fp32 or NULL; may alias `flow` (mode-3 composition, C = 2)
 * dst       [N,C,H,W] fp32, contiguous         -- must not alias an input
 * valid     [N,H,W] u8 or NULL (not wanted)
 * flow_flags / src_flags: optional int32[N] / int32[N] (or NULL): flag words (see
 *           ofl_flow_flags_f32) of the `flow` operand (with flow_mask) and, when C == 2 and
 *           src is itself a flow field, of `src` (with src_mask), OR-ed in as a by-product of the
 *           same pass.  The caller zeroes them beforehand.
 * dst_flags: optional int32[N] (C == 2 and `valid` wanted, else OFL_E_ARG; zeroed in-stream): the flag word of the
 *           OUTPUT read as a flow under `valid` -- spares the validation pass over an intermediate flow.
 */
int ofl_warp_bwd_f32(const float* flow, int64_t flow_bs, float flow_sign,
                     const float* src, int64_t src_bs,
                     const float* src_b, int64_t src_b_bs,
                     const uint8_t* src_mask, int64_t src_mask_bs,
                     const uint8_t* flow_mask, int64_t flow_mask_bs,
                     const float* addend, int64_t addend_bs, float a_sign, float g_sign,
                     float* dst, uint8_t* valid,
                     int32_t* flow_flags, int32_t* src_flags, int32_t* dst_flags,
                     int32_t n, int32_t c, int32_t h, int32_t w,
                     int32_t round_mode, void* stream);

/*
 * The same warp with a flow that covers a WINDOW of the frame -- Flow.apply(target, padding=[top, bottom, left, right]) with a
 * 't' flow (flow_class.py:901-913, 924-932): the reference zero-pads the flow to the target's size and pads its mask with
 * False before the warp; here flow [*,2,fh,fw] and flow_mask [*,fh,fw] stay as they are and cover rows foy .. foy + fh - 1,
 * columns fox .. fox + fw - 1 of the h x w frame of src / dst: outside the window the flow reads 0 and the flow mask False
 * (no padded copies).  Generic direct-gather kernel; no addend / src_b / flag outputs.  `cut` is a view of dst.
 */
int ofl_warp_bwd_win_f32(const float* flow, int64_t flow_bs, float flow_sign,
                         int32_t fh, int32_t fw, int32_t foy, int32_t fox,
                         const float* src, int64_t src_bs,
                         const uint8_t* src_mask, int64_t src_mask_bs,
                         const uint8_t* flow_mask, int64_t flow_mask_bs,
                         float* dst, uint8_t* valid,
                         int32_t n, int32_t c, int32_t h, int32_t w,
                         int32_t round_mode, void* stream);

/*
 * The same warp for 8-bit images (Flow.apply / apply_flow on uint8 targets, flow_class.py:943-951, utils.py:613-618):
 * src [*,C,H,W] uint8 is read as it is (no float copy); dst is fp32 [N,C,H,W] (dst_is_u8 = 0, any round_mode) or uint8
 * (dst_is_u8 = 1, round_mode must be OFL_ROUND_U8: round half to even, clamp to [0, 255] -- the values the reference's
 * `torch.round` / `clamp` / `.to(uint8)` produce).  17 instead of 35 bytes per pixel for C = 3 with masks, and no
 * conversion passes around the kernel.  Staged kernel only (W >= 4, H >= 2), else OFL_E_UNSUPPORTED: convert and
 * use ofl_warp_bwd_f32.
 */
int ofl_warp_bwd_u8(const float* flow, int64_t flow_bs, float flow_sign,
                    const uint8_t* src, int64_t src_bs,
                    const uint8_t* src_mask, int64_t src_mask_bs,
                    const uint8_t* flow_mask, int64_t flow_mask_bs,
                    void* dst, int32_t dst_is_u8, uint8_t* valid,
                    int32_t n, int32_t c, int32_t h, int32_t w,
                    int32_t round_mode, void* stream);

/*
 * Forward ("s"-reference) splat, pass 1: scatter-add of weight * value into `accum`.
 *
 *   (x, y)  = xy ? (xy_x[n], xy_y[n]) : (gx, gy) + flow_sign * flow[n]
 *   zero    = occlude && |flow_u| < 1e-3f && |flow_v| < 1e-3f         (strict, per component)
 *   wmask   = (weight_mask ? weight_mask[n] : 1) && !zero
 *   accum[n,0]       += w_corner * wmask                              (density)
 *   accum[n,1+c]     += w_corner * wmask * data_sign * data[n,c]
 *   accum[n,1+C]     += w_corner * wmask * !(chan_mask_a & chan_mask_b)  (only if with_mask_chan: the
 *                       INVALID weight; pass 2 forms density - invalid, which is exactly the density --
 *                       ratio exactly 1 -- when every contributor is valid, in any accumulation order)
 *
 * accum [N, 1 + C + with_mask_chan, H, W] fp32 workspace, zeroed by the caller.
 * Either `flow` (endpoints computed in-kernel, get_flow_endpoints) or `xs`/`ys` (explicit
 * positions, grid_from_unstructured_data) must be given; occlude requires `flow`.
 */
int ofl_splat_fwd_f32(const float* flow, int64_t flow_bs, float flow_sign,
                      const float* xs, const float* ys, int64_t xy_bs,
                      const float* data, int64_t data_bs, float data_sign,
                      const uint8_t* weight_mask, int64_t weight_mask_bs,
                      const uint8_t* chan_mask_a, int64_t chan_mask_a_bs,
                      const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
                      int32_t with_mask_chan, int32_t occlude,
                      float* accum,
                      int32_t n, int32_t c, int32_t h, int32_t w, void* stream);

/*
 * Forward splat, pass 2: normalise, masks, un-occlude fill.
 *
 *   den        = accum[n,0];  dcl = max(den, 1e-3f)
 *   dst[n,c]   = accum[n,1+c] / dcl ;  mch = (den - accum[n,1+C]) / dcl
 *   where (weight_mask & zero & den == 0) [occlude only]:  dst[n,c] = data_sign*data[n,c], mch = chan masks
 *   density[n] = den                      (optional)
 *   warped[n]  = den > 0                  (optional; apply_s_flow's returned mask)
 *   valid[n]   = mch > 0.99999f           (optional; requires with_mask_chan)
 *   mask_chan[n] = mch                    (optional; requires with_mask_chan; valid_target's `== 1` test)
 *   then `round_mode` on dst.
 */
int ofl_splat_finalize_f32(const float* accum,
                           const float* flow, int64_t flow_bs,
                           const float* data, int64_t data_bs, float data_sign,
                           const uint8_t* weight_mask, int64_t weight_mask_bs,
                           const uint8_t* chan_mask_a, int64_t chan_mask_a_bs,
                           const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
                           int32_t with_mask_chan, int32_t occlude,
                           float* dst, float* density, uint8_t* warped, uint8_t* valid, float* mask_chan,
                           int32_t n, int32_t c, int32_t h, int32_t w,
                           int32_t round_mode, void* stream);

/*
 * Forward splat, single fused call (the fast path of the two passes above, same arguments) -- an in-order GATHER:
 * a bin kernel appends the id of every 16 x 2 source subtile to the list of each destination tile (64 x 16: ofl_splat_tile_geometry) its end points
 * touch; a gather kernel (one block per destination tile) re-reads the listed source pixels, keeps those that land in
 * the tile as LDS records per unit cell, puts every cell in raster order of its source pixels and sums each destination
 * pixel's contributions in registers, per corner class in that order and ((c0 + c1) + c2) + c3 across the classes -- the
 * order of the reference's four scatter_add_ passes and corner sum (utils.py:1133-1143): results are BIT-IDENTICAL to the
 * reference's (and from run to run), not just within a tolerance.  Normalise / masks / un-occlude fill happen in the
 * same kernel; no float atomics, no accumulator, no per-pixel records in HBM.
 * Needs 4 <= W < 32768, H < 32768 (any width: 16-byte accesses at 4-byte alignment; more than 3 channels are
 * processed in groups of 3), else it returns OFL_E_UNSUPPORTED and the caller
 * uses ofl_splat_fwd_f32 + ofl_splat_finalize_f32.
 *   workspace      int32[ofl_splat_tiled_workspace_ints(n, h, w)]: 8 statistics words | one flag per image | list
 *                  lengths | 256 list entries per destination tile
 *                  (fixed addresses; ~2 B/px).  Contents irrelevant on entry; afterwards workspace[0] = 1 if some
 *                  IMAGE took the two-pass path, workspace[1] = number of tiles that left the exact path,
 *                  workspace[2] = number of such images
 *   data_b         optional [*,C,H,W] fp32 (C <= 2, else OFL_E_ARG): the data splatted is data - data_b (ONE fp32 subtraction per value, the
 *                  reference's `flow - self` in combine_with modes 1-2, flow_class.py:1763,1768), then x data_sign
 *   dst_flags      optional int32[N] (C == 2 only, else OFL_E_ARG; zeroed in-stream): the flag word (see
 *                  ofl_flow_flags_f32) of the OUTPUT read as a flow under its `valid` mask -- a by-product that spares
 *                  the caller the validation pass (utils.py:98, flow_class.py:1226-1244) over an intermediate flow
 *   accum_fallback fp32[ofl_splat_tiled_pass_images(n, h, w) * (1 + C + with_mask_chan) * H * W]  (one pass of the batch)
 *                  used (and zeroed in-stream, per image) only for an image in which a destination tile is touched by
 *                  more than 256 source subtiles (16 x 2 pixels each) or a subtile spreads over more than 256 destination tiles: the two-pass
 *                  global-atomics path then runs for THAT image inside the same call, decided on the device (no host
 *                  sync; tolerance instead of bit-exactness).  A heavy fold of the flow (> 64 source pixels ending in
 *                  one unit cell, or more records for one tile than four bands of its rows can hold) makes only ITS
 *                  tile fall back to (LDS) float atomics.  The limits are on lengths, never on timing: the choice of
 *                  path, and with it every bit of the result, is the same in every run.
 */
int64_t ofl_splat_tiled_workspace_ints(int32_t n, int32_t h, int32_t w);
int64_t ofl_splat_tiled_pass_images(int32_t n, int32_t h, int32_t w);   /* images handled per pass (<= n) */
/* images `accum_fallback` must hold, for `planes` = 1 + min(C, 3) + (with_mask_chan ? 1 : 0) planes per image: the pass, capped
 * at 1 GiB (>= 1) -- accum_fallback is fp32 [that many, planes, H, W] */
int64_t ofl_splat_tiled_fallback_images(int32_t n, int32_t planes, int32_t h, int32_t w);
/* the geometry this build of the gather splat works with (what the stats words of a call count in, and what a caller that wants
 * to provoke / predict the fallbacks has to know): destination tile width and height in pixels, and the number of 16 x 2 source
 * subtiles one destination tile can list before its IMAGE takes the two-pass path.  Any pointer may be NULL; returns OFL_OK. */
int ofl_splat_tile_geometry(int32_t* tile_w, int32_t* tile_h, int32_t* list_capacity);
/* resources of the gather kernel the given kind of call runs on (round 6: the evidence behind "three blocks per CU"), as the HIP
 * runtime reports them for the loaded code object: info4[0] = 512-thread blocks resident per CU (with `extra_lds` bytes of dynamic
 * LDS added, see OFL_OPT_SPLAT_EXTRA_LDS), [1] = static LDS bytes per block, [2] = VGPRs, [3] = scratch bytes per thread.
 * channels 1..3; elem 0 = fp32, 1 = fp16 in / fp32 out, 2 = fp16 in and out (2 channels); lean: the common-case instantiation
 * (a flow, no window, W % 4 == 0, no rounding; 2 and 3 channels).  Needs a HIP device. */
int ofl_splat_gather_info(int32_t channels, int32_t with_mask_chan, int32_t elem, int32_t lean, int32_t extra_lds, int32_t* info4);
int ofl_splat_tiled_f32(const float* flow, int64_t flow_bs, float flow_sign,
                        const float* xs, const float* ys, int64_t xy_bs,
                        const float* data, int64_t data_bs, float data_sign,
                        const float* data_b, int64_t data_b_bs,
                        const uint8_t* weight_mask, int64_t weight_mask_bs,
                        const uint8_t* chan_mask_a, int64_t chan_mask_a_bs,
                        const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
                        int32_t with_mask_chan, int32_t occlude,
                        float* dst, float* density, uint8_t* warped, uint8_t* valid, float* mask_chan,
                        int32_t* dst_flags,
                        int32_t* workspace, int64_t workspace_ints, float* accum_fallback,
                        int32_t n, int32_t c, int32_t h, int32_t w,
                        int32_t round_mode, void* stream);

/*
 * ofl_splat_tiled_f32 with a flow that covers a WINDOW of the frame -- Flow.apply(target, padding=...) with an 's' flow
 * (flow_class.py:880-895, 906-913): the reference pads the flow with mode 'replicate' and its mask with False.  Here flow
 * [*,2,fh,fw], weight_mask and chan_mask_b [*,fh,fw] are read through replicate addressing (flow) / False outside the
 * window (masks; a NULL chan_mask_b is "all True INSIDE the window"; a NULL weight_mask still means every pixel of the frame
 * contributes: consider_mask=False); data, chan_mask_a and every output have the h x w geometry of the frame.
 * Same kernels as ofl_splat_tiled_f32 (bit-identical sums), per-pixel loads for the windowed operands.
 */
int ofl_splat_tiled_win_f32(const float* flow, int64_t flow_bs, float flow_sign,
                            int32_t fh, int32_t fw, int32_t foy, int32_t fox,
                            const float* data, int64_t data_bs, float data_sign,
                            const uint8_t* weight_mask, int64_t weight_mask_bs,
                            const uint8_t* chan_mask_a, int64_t chan_mask_a_bs,
                            const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
                            int32_t with_mask_chan, int32_t occlude,
                            float* dst, float* density, uint8_t* warped, uint8_t* valid, float* mask_chan,
                            int32_t* workspace, int64_t workspace_ints, float* accum_fallback,
                            int32_t n, int32_t c, int32_t h, int32_t w,
                            int32_t round_mode, void* stream);

/*
 * The weighted sums of a forward splat WITHOUT the division by the density:
 *   dst[n,c,q] = sum over source pixels i and corners k that land on q of  w_ik * data_sign * data[n,c,i]
 * with the end points and weights of ofl_splat_tiled_f32 (every pixel contributes: no weight mask, no occlusion rule).
 * This is the transpose of the backward warp: with data = the upstream gradient and flow_sign = MINUS the warp's
 * flow_sign it is the gradient of ofl_warp_bwd_f32 with respect to its source (ATen: the grad_input scatter of
 * grid_sampler_2d_backward) -- one gather launch instead of twelve global float atomics per pixel (B=16 1080p C=3:
 * 1.0 ms instead of 7.1 ms).  Same workspace, fallback accumulator ((1 + C) planes per image of a pass) and limits as
 * ofl_splat_tiled_f32; OFL_E_UNSUPPORTED for shapes that one does not take (the caller then uses the atomics of
 * ofl_warp_bwd_grad_f32).
 */
int ofl_splat_sum_f32(const float* flow, int64_t flow_bs, float flow_sign,
                      const float* data, int64_t data_bs, float data_sign, float* dst,
                      int32_t* workspace, int64_t workspace_ints, float* accum_fallback,
                      int32_t n, int32_t c, int32_t h, int32_t w, void* stream);

/*
 * Per-batch-element flag word of a flow field (OR-ed into flags[n]; caller zeroes):
 *   bit 0 (1)  some component is NaN / +-Inf
 *   bit 1 (2)  some component != 0
 *   bit 2 (4)  some component outside (-thr, thr)
 *   bit 3 (8)  some component != 0 where mask is True
 *   bit 4 (16) some component outside (-thr, thr) where mask is True
 * mask [*,H,W] u8 or NULL (all True).
 */
#define OFL_FLAG_NONFINITE 1
#define OFL_FLAG_NZ 2
#define OFL_FLAG_NZ_THR 4
#define OFL_FLAG_NZ_MASKED 8
#define OFL_FLAG_NZ_THR_MASKED 16

int ofl_flow_flags_f32(const float* flow, int64_t flow_bs,
                       const uint8_t* mask, int64_t mask_bs, float thr,
                       int32_t* flags, int32_t n, int32_t h, int32_t w, void* stream);

/*
 * The same reduction with the read-back built in (the reference's constructor raises on non-finite vectors, utils.py:98, so
 * `Flow(...)` has to WAIT for the words: this entry point makes that wait one kernel and one polled host word instead of a
 * memset, the kernel, a copy and an event).
 *   flow        [*,2,H,W] fp32, or fp16 when flow_is_f16 (then H*W % 4 == 0 and 8 / 4-byte aligned planes, else
 *               OFL_E_UNSUPPORTED);
 *   work        DEVICE int32[n + OFL_FLAGS_HOST_WORK_EXTRA], all zero before the first call; the kernel leaves it all zero again
 *               (flag words + arrival counters: one buffer can serve every later call on the same stream, ONE call in flight
 *               at a time);
 *   host_words  HOST-VISIBLE int32[2 * n], 8-byte aligned (hipHostMalloc, coherent + mapped: ofl_host_words_alloc): the block
 *               that finishes last stores, for every batch element i, the pair {host_words[2 i] = serial, host_words[2 i + 1] =
 *               flag word} with ONE 8-byte store.  The caller polls until every host_words[2 i] == serial (a value it has not
 *               used before) and reads the words beside them.
 * Contract of the shared buffers (what the binder must enforce; oflibpytorch_amd/_native.py does it with a pool of slots):
 *   * ONE call in flight per (work, host_words) pair -- a second launch on the same pair before the first one's words have
 *     been seen corrupts both.  Two threads (or two streams) that validate at once need two pairs; nothing in the library
 *     serialises them.  The last block zeroes the arrival counters with returning atomics BEFORE it publishes the first pair,
 *     so a pair may be handed to the next call (on any stream) the moment every host_words[2 i] == serial has been read.
 *   * The wait is a HOST poll of memory the kernel writes: it cannot be part of a stream capture / hipGraph, and the calling
 *     thread spins (then sleeps) until the device has run the kernel.  A caller that builds graphs validates outside them
 *     (ofl_flow_flags_f32 writes device words and captures fine).
 *   * If the caller abandons a wait (exception, interrupt) it must synchronise the stream before it re-uses or frees the pair.
 */
#define OFL_FLAGS_HOST_WORK_EXTRA 33
int ofl_flow_flags_host(const void* flow, int32_t flow_is_f16, int64_t flow_bs,
                        const uint8_t* mask, int64_t mask_bs, float thr,
                        int32_t* work, int32_t* host_words, int32_t serial,
                        int32_t n, int32_t h, int32_t w, void* stream);
/* pinned, coherent, device-mapped host words for ofl_flow_flags_host (zeroed); *out is usable as host AND device pointer */
int ofl_host_words_alloc(int64_t ints, void** out);
int ofl_host_words_free(void* ptr);

/*
 * fp16-STORED FLOWS (BASELINE config 5; SURVEY.md section 8b "fp16-I/O variants").  The reference up-casts every flow to
 * fp32 on entry (utils.py:95, 118) and computes in fp32; so do these entry points -- the up-conversion is exact and happens
 * in registers, the arithmetic is the fp32 arithmetic of the plain entry points, and the results are bit-identical to
 * feeding them the up-cast tensors -- but the fp16 planes are read as they are (half the operand bytes, no conversion pass).
 *
 *   ofl_flow_from_f16     validation of an fp16-stored flow: flags[n] |= flag word under `mask` (caller zeroes flags) and,
 *                         when dst != NULL, dst[N,2,H,W] fp32 = (float) src in the same pass (12 instead of 12 + 9 B/px).
 *                         dst == NULL: flags only (5 B/px) -- the flow stays in fp16.  src [*,2,H,W] fp16, H*W % 4 == 0,
 *                         8-byte aligned planes (dst 16-byte, mask 4-byte aligned); otherwise OFL_E_UNSUPPORTED.
 *   ofl_splat_tiled_f16   ofl_splat_tiled_f32 for a flow splatted by a flow (switch_ref, invert, Flow.apply(Flow) with 's'
 *                         flows): warper flow_f16 [*,2,H,W] and data data_f16 [*,2,H,W] both fp16; dst [N,2,H,W] fp32, or
 *                         fp16 when dst_is_f16 (an OPTION for fp16 pipelines: one round-to-nearest-even at the store, and
 *                         dst_flags then describe the stored values; never the default -- the reference returns fp32).
 *                         Same gather kernels, same bit-exact sums, same workspace / fallback contract.
 *   ofl_warp_bwd_h_f32    ofl_warp_bwd_f32 with a 2-channel SOURCE stored in fp16 (the `flow` operand of combine_with mode
 *                         1 't', flow_class.py:1763), optionally minus an fp32 src_b; fp32 warper, fp32 dst, valid mask.
 *                         Staged kernel only (W >= 4, H >= 2, H*W < 2^24), else OFL_E_UNSUPPORTED.
 */
int ofl_flow_from_f16(const void* src_f16, int64_t src_bs,
                      const uint8_t* mask, int64_t mask_bs,
                      float* dst, int32_t* flags,
                      int32_t n, int32_t h, int32_t w, void* stream);
int ofl_splat_tiled_f16(const void* flow_f16, int64_t flow_bs, float flow_sign,
                        const void* data_f16, int64_t data_bs, float data_sign,
                        const uint8_t* weight_mask, int64_t weight_mask_bs,
                        const uint8_t* chan_mask_a, int64_t chan_mask_a_bs,
                        const uint8_t* chan_mask_b, int64_t chan_mask_b_bs,
                        int32_t with_mask_chan, int32_t occlude,
                        void* dst, int32_t dst_is_f16, uint8_t* valid, int32_t* dst_flags,
                        int32_t* workspace, int64_t workspace_ints, float* accum_fallback,
                        int32_t n, int32_t h, int32_t w, void* stream);
int ofl_warp_bwd_h_f32(const float* flow, int64_t flow_bs, float flow_sign,
                       const void* src_f16, int64_t src_bs,
                       const float* src_b, int64_t src_b_bs,
                       const uint8_t* src_mask, int64_t src_mask_bs,
                       const uint8_t* flow_mask, int64_t flow_mask_bs,
                       float* dst, uint8_t* valid,
                       int32_t n, int32_t h, int32_t w, void* stream);



/* ------------------------------------------------------------------------------------------------
 * Either side of the path: backward passes, point tracking, padding extents.
 * ---------------------------------------------------------------------------------------------- */

/*
 * Backward pass of ofl_warp_bwd_f32 (the reference relies on autograd through F.grid_sample, utils.py:555, and through
 * normalise_coords / `grid - flow`, utils.py:462-465, 549; differentiability is asserted by its tests, e.g.
 * test_utils.py:500):
 *   grad_src[n,c,tap] += g_scale * w_tap * grad_out[n,c]       float atomics; grad_src is ZEROED BY THE CALLER; its batch
 *                                                               stride may be 0 when the forward source was broadcast
 *   grad_flow[n,:]     = -flow_sign * d(sample)/d(position)     [N,2,H,W], written (not accumulated); a broadcast flow's
 *                                                               gradient is the caller's sum over n
 * flow / src as in the forward call (src = the field that was gathered, i.e. src - src_b where that was used); grad_out
 * [N,C,H,W] contiguous; g_scale = the forward g_sign.  Either output may be NULL (not both).  Sums are not in ATen's CPU
 * order: results agree with the reference's autograd within fp32 rounding (tests: rtol 1e-4 of the gradient scale).
 */
int ofl_warp_bwd_grad_f32(const float* flow, int64_t flow_bs, float flow_sign,
                          const float* src, int64_t src_bs,
                          const float* grad_out, float g_scale,
                          float* grad_src, int64_t grad_src_bs, float* grad_flow,
                          int32_t n, int32_t c, int32_t h, int32_t w, void* stream);

/*
 * Backward pass of the forward splat (autograd through utils.py:1098-1144 and :1185-1203 in the reference; claims at
 * utils.py:1079-1080, 1167; asserted at test_utils.py:1113-1114).  A gather per source pixel, no atomics:
 *   grad_data[n,c,i] = sum_k w_ik * grad_out[n,c,p_ik] / max(D[p_ik], 1e-3)        (+ grad_out[n,c,i] where i was un-occlude-filled)
 *   grad_xy[n,0|1,i] = d/dx, d/dy of the same sums through the corner weights (and through D where D >= 1e-3)
 * flow|xs,ys / data (the data actually splatted: data_sign * (data - data_b)) / weight_mask / occlude as in the forward
 * call; out [N,C,H,W] and density [N,H,W] are the forward results (out before any rounding); grad_out [N,C,H,W];
 * grad_density optional [N,H,W] (upstream gradient of the density output).  grad_data [N,C,H,W], grad_xy [N,2,H,W]
 * (x then y; for a flow operand grad_flow = flow_sign * grad_xy) -- either may be NULL.  C <= 3 per call, else
 * OFL_E_UNSUPPORTED (split the channels; grad_density with the first group only; the grad_xy of the groups add up).
 * scratch: fp32[N * 4 * H * W] (a first pass leaves g_c / max(D, 1e-3) and the density term of every destination pixel in
 * one 16-byte slot: the divisions are done once, and the gather loads one slot per corner instead of 2 C + 1 scalars).
 */
int ofl_splat_grad_f32(const float* flow, int64_t flow_bs, float flow_sign,
                       const float* xs, const float* ys, int64_t xy_bs,
                       const float* data, int64_t data_bs,
                       const uint8_t* weight_mask, int64_t weight_mask_bs, int32_t occlude,
                       const float* out, const float* density,
                       const float* grad_out, const float* grad_density, float* scratch,
                       float* grad_data, float* grad_xy,
                       int32_t n, int32_t c, int32_t h, int32_t w, void* stream);

/*
 * Sparse point sampler of track_pts (utils.py:1004-1014, 1033-1035): pts [*,M,2] fp32 as (y, x); out[n,m] = pts + the
 * flow bilinearly sampled there (flip -> normalise_coords -> grid_sample(align_corners=True) -> flip: same fp32 operation
 * order and FMA chain as the dense warp), rows with a NaN set to 0.  pts_bs = 0 broadcasts one point list.
 * ..._grad: grad_flow [N,2,H,W] (float atomics; zeroed by the caller) and / or grad_pts [N,M,2].
 */
int ofl_sample_pts_f32(const float* flow, int64_t flow_bs, const float* pts, int64_t pts_bs, float* out,
                       int32_t n, int32_t m, int32_t h, int32_t w, void* stream);
int ofl_sample_pts_grad_f32(const float* flow, int64_t flow_bs, const float* pts, int64_t pts_bs,
                            const float* grad_out, float* grad_flow, float* grad_pts,
                            int32_t n, int32_t m, int32_t h, int32_t w, void* stream);

/*
 * Extents of the positions a flow reaches, under its mask -- the reduction of Flow.get_padding (flow_class.py:1196-1219):
 *   pos = -(sign * thr(v) - grid) per component (thr: threshold_vectors, |v| < 1e-3 -> 0); sign = +1 for 't', -1 for 's'
 *   extents[n] = { min y, max y, min x, max x, any valid pixel (0 / 1) }   fp32[N][5]
 * workspace int32[5 N] (scratch).  min / max are exact in any order.
 */
int ofl_flow_extents_f32(const float* flow, int64_t flow_bs, const uint8_t* mask, int64_t mask_bs, float sign,
                         int32_t* workspace, float* extents, int32_t n, int32_t h, int32_t w, void* stream);

/*
 * Valid area of a backward warp -- Flow.valid_target of a 't' flow (flow_class.py:1119-1122) and Flow.valid_source of an 's' flow
 * (:1151-1157; flow_sign = -1 restates its `-self._vecs`):
 *   valid[n] = (grid_sample(ones, normalise_coords(grid - flow_sign * flow[n]), align_corners=True) > thr) & mask[n]
 * replaces torch.ones + F.grid_sample (utils.py:555) + gt + and.  The all-ones image is never made: its taps are the in-frame
 * indicators of the four neighbours, blended by the same FMA chain -- bit-identical to warping a ones image.  mask may be NULL
 * (all True); valid uint8 [N,H,W]; thr = 0.9999f in both callers.  9 B/px read, 1 B/px written.
 */
int ofl_warp_valid_f32(const float* flow, int64_t flow_bs, float flow_sign, const uint8_t* mask, int64_t mask_bs, float thr,
                       uint8_t* valid, int32_t n, int32_t h, int32_t w, void* stream);

/*
 * Bilinear resize -- the `F.interpolate(x, scale_factor=[sh, sw], mode='bilinear')` (align_corners=False) of resize_flow
 * (utils.py:908) and Flow.resize (flow_class.py:710), computed as ATen's CPU kernels compute it, so that a resize on a HIP device
 * returns the reference's PyTorch-CPU values BIT FOR BIT (ATen's own GPU kernel differs in the last bits):
 *   source index fmaf(rcp_scale, dst + 0.5, -0.5) clamped at 0; a dimension that keeps its size is copied; ATen's two kernels
 *   (chosen by oh + ow <= 128) both restated -- see ofl_aux_kernels.hip / oracle/ofl_oracle.c.
 * src [planes, h, w] fp32 contiguous -> dst [planes, oh, ow]; the caller passes oh = floor(h * sh), ow = floor(w * sw) (doubles, as
 * torch computes the output size) and rcp_scale_* = (float)(1.0 / s*).  The components of a flow are scaled by the caller
 * afterwards (utils.py:913-914: two exact multiplications).  planes <= 65535.
 */
int ofl_resize_bilinear_f32(const float* src, float* dst, int32_t planes, int32_t h, int32_t w, int32_t oh, int32_t ow,
                            float rcp_scale_h, float rcp_scale_w, void* stream);

/*
 * Flow field of a 3 x 3 transformation matrix (flow_from_matrix, utils.py:339-376; the O(HW) half of from_matrix :646-705 and
 * from_transforms :729-807, whose 3 x 3 algebra stays on the host):
 *   hom      = M [x, y, 1]^T    accumulated as ATen's CPU batched matmul does for 3 x 3 operands (acc = 0; acc += m_ik * v_k)
 *   dst[n,0] = sign * (hom.x / hom.z - x),  dst[n,1] = sign * (hom.y / hom.z - y)
 * matrices [*,3,3] fp32 DEVICE memory (matrix_bs = 9, or 0 to broadcast one matrix), sign = +1 ('s') / -1 (the reference's
 * negation of its 't' branch, exact), dst [N,2,H,W] fp32.  Bit-identical to the reference's PyTorch-CPU result.
 */
int ofl_flow_from_matrix_f32(const float* matrices, int64_t matrix_bs, float sign, float* dst,
                             int32_t n, int32_t h, int32_t w, void* stream);

/*
 * The flag words of a batch (ofl_flow_flags_f32, or the by-product words of the warp / splat kernels), copied, followed by
 * their OR over the batch as FIVE 0 / 1 integers, one per bit -- the form an all-reduce (MAX; NCCL / RCCL has no bitwise
 * OR) over the ranks of a batch-sharded job needs, so that the reference's batch-global tests (isfinite().all(),
 * utils.py:98; all(is_zero), utils.py:497, flow_class.py:1046, 1729, 1738) stay exact under sharding with one launch, one
 * collective and one read-back per tensor.   words int32[N] -> out int32[N + 5]
 */
int ofl_flag_words_or_i32(const int32_t* words, int32_t n, int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OFLIB_HIP_H */
