/*
 * stainx_hip.h -- C ABI of libstainx_hip.so, the MI355X (gfx950) stain-normalisation library.
 *
 * This is the drop-in boundary for the hot path of rendeirolab/stainx: the four functions its
 * pybind11 extension `stainx_cuda_torch` exports (src/stainx_cuda_torch/csrc/bindings.cpp:31-34)
 * plus the fit-time statistics the reference computes with its torch backend
 * (src/stainx/backends/torch_backend.py:143-179, 308-323, 463-519).
 *
 * Conventions
 *  - plain C: pointers, sizes, enums.  No torch / ATen types.
 *  - every pointer named *_dev is DEVICE memory owned by the caller (torch's caching allocator in the
 *    Python host); the library never allocates, frees or retains pointers.
 *  - `stream` is a hipStream_t passed as void*; every entry point only enqueues work on it and never
 *    synchronises the host (stainx's Macenko/HM natives are asynchronous too: macenko.cu:102).
 *  - images are dense NCHW (HM: optionally NHWC) with C == 3, element type `dtype`.
 *  - return value: SX_OK or an sx_status error; sx_last_error_string() describes the last error of
 *    the calling thread.  Nothing throws or aborts.
 *  - `workspace_dev` must hold at least sx_*_workspace_bytes() bytes, 256-byte aligned.  Its contents
 *    need no initialisation and are dead after the call (except for sx_macenko_tile_params()).
 */
#ifndef STAINX_HIP_H
#define STAINX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SX_ABI_VERSION 1

typedef enum { SX_U8 = 0, SX_F16 = 1, SX_BF16 = 2, SX_F32 = 3, SX_F64 = 4 } sx_dtype;

typedef enum {
    SX_OK = 0,
    SX_ERR_BAD_ARG = 1,    /* null pointer, non-positive size, C != 3 ... (reference: TORCH_CHECK -> RuntimeError) */
    SX_ERR_DTYPE = 2,      /* unsupported element type */
    SX_ERR_WORKSPACE = 3,  /* workspace too small or misaligned */
    SX_ERR_LAUNCH = 4      /* hipGetLastError() after a launch (reference: macenko.cu:125-126) */
} sx_status;

/* flags for sx_macenko_transform: the six public bits */
#define SX_MACENKO_NORMALIZE_0_1 1u /* fuse `result / 255.0` (normalizers/_template.py:111-112); u8 input -> f32 output */
#define SX_MACENKO_CHANNELS_LAST 2u /* images and output are (N,H,W,3) (decoder / PIL layout) instead of (N,3,H,W); an extension: the
                                      reference takes NCHW only and callers permute + copy first */
#define SX_MACENKO_SAMPLED 4u       /* an APPROXIMATION (not a parity path, not the reference's precision="fast"): the percentiles of a 4096-pixel
                                      sample of each tile stand in for the exact ones -- moments pass, one per-tile stage, reconstruct.  Mean
                                      error ~0.5, worst ~5 grey levels on H&E tiles.  Macenko(precision="sampled") in the Python host; the
                                      reference's precision="fast" (fp16 tensors, exact percentiles, MAE ~0.05) is served by the exact path. */
#define SX_MACENKO_CLASSIC 16u       /* the four-pass form of the transform even where the two-pass form would be chosen (same bits; callers whose
                                       batches hold tiles the two-pass form cannot speculate on: see sx_macenko_telemetry_offset) */
#define SX_MACENKO_OUT_BF16 32u      /* uint8 input only: the result is written as bfloat16 -- bit for bit `transform(x).to(bfloat16)`, with
                                       SX_MACENKO_NORMALIZE_0_1 `transform(x, normalize_to_0_1).to(bfloat16)` -- so a uint8 tile from the decoder
                                       becomes a model's bf16 input with 3 bytes read and 6 written per pixel (an extension: SURVEY.md 8f-2) */
#define SX_MACENKO_OUT_F16 64u       /* the same with float16 */

/* Diagnostic builds only (-DSX_DIAG: stainx_amd/_lib/libstainx_diag.so, built next to the product by __graft_entry__.build(); the product
   library refuses these bits with SX_ERR_BAD_ARG).  Tests force the rare paths with them; two measured-and-slower forms of the transform
   live there as design studies (DESIGN.md sections 4c, 4e). */
#ifdef SX_DIAG
#define SX_MACENKO_NO_TIE_SHORTCUT 8u /* do not resolve a bracket that closed on one key from its counts (forces the slow exact paths) */
#define SX_MACENKO_SPEC_FAIL 128u    /* the two-pass form treats every speculation as failed (forces its slow exact path) */
#define SX_MACENKO_TWO_PASS 256u     /* the two-pass form wherever it can run (by default only where it is the faster one) */
#define SX_MACENKO_FUSE 512u         /* the two-pass form with its last three launches as ONE launch with tile-level dependencies (planar float32 tiles of
                                       128x128 ... 512x512; same bits; slower: DESIGN.md 4c) */
#define SX_MACENKO_RESIDENT 1024u    /* the tile-resident form: one launch, a tile's pixels kept on chip as 8-bit codes (same bits; slower: DESIGN.md 4e) */
#define SX_MACENKO_NO_CODES 2048u    /* the four passes over float32 tiles read the float pixels in every pass (by default the first pass leaves 8-bit codes
                                       of the tiles that consist of grey levels and the later passes read those: same bits) */
#endif

int sx_version(void);
const char* sx_last_error_string(void);

/* ---------------------------------------------------------------- Macenko ------------------------
 * Replaces stainx_cuda_torch.macenko (bindings.cpp:33; src/stainx_cuda_torch/csrc/macenko.cu:67-266)
 * with the numerics of MacenkoTorch.transform (torch_backend.py:521-560).
 *   images_dev        (N,3,H,W) `dtype`; u8 is [0,255], floats are taken as [0,1] as is
 *   out_dev           (N,3,H,W) same dtype (f32 when dtype==SX_U8 and SX_MACENKO_NORMALIZE_0_1; bf16 / f16 when dtype==SX_U8 and
 *                     SX_MACENKO_OUT_BF16 / SX_MACENKO_OUT_F16); (N,H,W,3) in and out with SX_MACENKO_CHANNELS_LAST
 *   stain_matrix_dev  6 floats, row-major (3,2)      target_max_conc_dev  2 floats
 */
size_t sx_macenko_workspace_bytes(int64_t n_tiles, int64_t height, int64_t width);
/* sx_macenko_workspace_bytes() serves ANY sx_macenko_* call on such a batch.  What ONE sx_macenko_transform call with these
 * arguments needs is a prefix of it and can be much less: the call checks its workspace against THIS size (narrow pixels and
 * small batches take the four-pass form and none of the two-pass areas; the fused launch does not need the four-launch form's
 * candidate arrays).  sx_macenko_fit and the sx_macenko_dfit_* and sx_macenko_pfit_* steps need the size of a call with SX_MACENKO_CLASSIC. */
size_t sx_macenko_workspace_bytes_for(int dtype, int64_t n_tiles, int64_t height, int64_t width, unsigned flags);

int sx_macenko_transform(const void* images_dev, void* out_dev, int dtype, int64_t n_tiles, int64_t height,
                         int64_t width, const float* stain_matrix_dev, const float* target_max_conc_dev,
                         unsigned flags, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Byte offset, inside the workspace, of a uint32 RUNNING count of per-tile selections that left the speculative path of the
 * two-pass form (tiles without tissue, tiles without a stable stain plane ...: each costs a whole-tile exact select, ~0.1-0.5 ms).
 * The library only ever adds to it (its value in a fresh workspace is whatever the memory held): a host reads it back
 * asynchronously, takes the difference to what it read before, and passes SX_MACENKO_CLASSIC for data that makes it grow -- the
 * four-pass form has no such cliff.  (A count per call would be reset by the next call's first kernel before a host that lets
 * its calls queue up could read it.) */
size_t sx_macenko_telemetry_offset(void);

/* 1 if sx_macenko_transform takes its two-pass form for such a call (element type, batch, tile size, flags), 0 for the four-pass
 * form: a host only needs to watch the telemetry word after calls of the first kind. */
int sx_macenko_takes_two_pass(int dtype, int64_t n_tiles, int64_t height, int64_t width, unsigned flags);
/* The same in full: 0 four passes, 1 two-pass form as four launches, 2 two-pass form with pass A, the per-tile stages and the
 * reconstruct pass in one launch (only with SX_MACENKO_FUSE; falls back to 0 at call time when a pointer is not 16-byte aligned). */
int sx_macenko_form(int dtype, int64_t n_tiles, int64_t height, int64_t width, unsigned flags);

/* Replaces MacenkoTorch.compute_reference_stain_matrix_torch (torch_backend.py:463-519): one stain
 * estimate pooled over all n_tiles*H*W pixels, no "<3 kept pixels" fallback.
 *   he_out_dev 6 floats (3,2) row-major;  max_c_out_dev 2 floats */
int sx_macenko_fit(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                   float* he_out_dev, float* max_c_out_dev, void* workspace_dev, size_t workspace_bytes,
                   void* stream);

/* Per-tile intermediates of the LAST sx_macenko_transform / sx_macenko_fit that used `workspace_dev`
 * (tests compare them with the oracle).  params_out_dev: n_groups x SX_MACENKO_PARAM_FLOATS floats:
 *   [0] n_selected  [1] used_all_pixels  [2..7] plane vectors (3,2)  [8] phi_lo  [9] phi_hi
 *   [10..15] HE_source (3,2)  [16..17] maxC  [18] select paths taken (bit i: slot i fell back to the
 *   full-tile radix select)  [19..22] candidates gathered per slot  [23..31] covariance (3,3)
 *   [32..47] diagnostic stage timestamps of the per-tile kernels, microseconds                     */
#define SX_MACENKO_PARAM_FLOATS 48
int sx_macenko_tile_params(const void* workspace_dev, int64_t n_groups, float* params_out_dev, void* stream);

/* Pooled fit over a batch that is SHARDED ACROSS RANKS (one process per GPU).  The host all-reduces
 * (SUM) the small buffers between the calls; every rank ends with identical (HE, maxC):
 *   sx_macenko_dfit_moments   local 20 fp64 raw moments            -> all-reduce
 *   sx_macenko_dfit_begin     plane vectors from the global moments, starts the angle selection
 *   repeat 4x: sx_macenko_dfit_histogram(stage)  local 2 x 256 u64 bins of the current radix round
 *              -> all-reduce -> sx_macenko_dfit_advance(stage)     (stage 0: phi@1,phi@99; stage 1: C0@99,C1@99)
 *   sx_macenko_dfit_result    copies HE (6 floats) and maxC (2 floats) out of the state             */
size_t sx_macenko_dfit_state_bytes(void);
int sx_macenko_dfit_moments(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                            double* moments_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int sx_macenko_dfit_begin(const double* moments_dev, void* state_dev, void* stream);
int sx_macenko_dfit_histogram(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                              const void* state_dev, int stage, unsigned long long* hist_out_dev, void* stream);
int sx_macenko_dfit_advance(void* state_dev, int stage, const unsigned long long* hist_dev, void* stream);
int sx_macenko_dfit_result(const void* state_dev, float* he_out_dev, float* max_c_out_dev, void* stream);

/* The same pooled fit across ranks on the BRACKET machinery of sx_macenko_fit: three passes over the local tiles
 * instead of nine.  Every rank calls the steps in lockstep; the host moves four small device buffers in between
 * (`stainx_amd/distributed.py` does it with torch.distributed):
 *   sx_macenko_pfit_stats    local moments (10 doubles) -> all-reduce(SUM); local sample (3 x 4096 floats, the first
 *                            sx_macenko_pfit_sample_count() columns valid) -> all-gather; every rank builds the same
 *                            4096-column union (every world-th column of every rank)
 *   sx_macenko_pfit_plane    global moments + n_all (pixels over all ranks) + union sample -> plane, angle brackets
 *   sx_macenko_pfit_pass     stage 0 angle / 1 concentration: local counts and histogram (SX_PFIT_SUMS int64 values)
 *                            -> all-reduce(SUM)
 *   sx_macenko_pfit_gather   global sums in; local candidates of the picked bin out (2 x share keys, 2 counts; share =
 *                            SX_PFIT_COMPACT / world) -> all-gather
 *   sx_macenko_pfit_finish   gathered candidates [world][2][share], counts [world][2] -> exact percentiles;
 *                            stage 0: stain vectors and concentration brackets, stage 1: HE, maxC and *status_out
 *                            (non-zero: a bracket did not hold somewhere -- repeat with the sx_macenko_dfit_* rounds)
 * Integer counts and an order-independent selection: every rank ends with the same bits. */
#define SX_PFIT_SUMS 1033
#define SX_PFIT_COMPACT 32768
int sx_macenko_pfit_sample_count(int64_t n_tiles, int64_t height, int64_t width);
int sx_macenko_pfit_stats(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                          double* moments_out_dev, float* sample_out_dev, void* workspace_dev, size_t workspace_bytes,
                          void* stream);
int sx_macenko_pfit_plane(const double* moments_dev, long long n_all, const float* sample_union_dev, int sample_count,
                          int64_t n_tiles, int64_t height, int64_t width, void* workspace_dev, size_t workspace_bytes,
                          void* stream);
int sx_macenko_pfit_pass(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width, int stage,
                         long long n_all, int sample_count, long long* sums_out_dev, void* workspace_dev,
                         size_t workspace_bytes, void* stream);
int sx_macenko_pfit_gather(const long long* sums_global_dev, int stage, long long n_all, int sample_count,
                           int64_t n_tiles, int64_t height, int64_t width, int share, unsigned* compact_out_dev,
                           int* counts_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int sx_macenko_pfit_finish(const unsigned* gathered_compact_dev, const int* gathered_counts_dev, int world, int share, int stage,
                           long long n_all, int sample_count, int64_t n_tiles, int64_t height, int64_t width,
                           float* he_out_dev, float* max_c_out_dev, int* status_out_dev, void* workspace_dev,
                           size_t workspace_bytes, void* stream);

/* The same steps with what travels through a collective PACKED AND UNPACKED BY THE LIBRARY: one contiguous record per rank and
 * exchange, so the host side is an all-gather / all-reduce of a buffer it never looks into (the unpacked steps above cost the
 * host a dozen small tensor operations per exchange: a third of a pooled fit_transform step).
 *   stats record  [int64 tiles | 10 fp64 moments | 3 x 4096 fp32 sample]                  SX_PFIT_STATS_RECORD_BYTES, 8-byte aligned
 *   stage record  [int32 count, count, stale flag | 2 x share uint32 candidate keys]      (3 + 2 share) x 4 bytes
 *   sx_macenko_pfit_stats_packed   the rank's stats record -> all-gather
 *   sx_macenko_pfit_plane_packed   every rank's record [world][record]: moments added up in rank order, the union sample (every
 *                                  world-th column of every rank, ranks one after the other, cut at 4096; sample_counts_host[r] =
 *                                  sx_macenko_pfit_sample_count of rank r's tiles, a HOST array), and -- if expected_tiles_dev
 *                                  is given -- *stale_out_dev = 1 when some rank's tile count differs from it (the host may take
 *                                  the counts of the last call on trust and have them checked here), else 0
 *   sx_macenko_pfit_gather_packed  as sx_macenko_pfit_gather; writes the rank's stage record, stale_flag_dev (may be null) into it
 *   sx_macenko_pfit_finish_packed  as sx_macenko_pfit_finish on the gathered stage records [world][3 + 2 share]; any rank's stale
 *                                  flag sets bit 4 (16) of *status_out (bits 0-3: brackets that did not hold) */
#define SX_PFIT_STATS_RECORD_BYTES 49240
int sx_macenko_pfit_stats_packed(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                                 void* record_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int sx_macenko_pfit_plane_packed(const void* gathered_records_dev, int world, const int* sample_counts_host,
                                 const long long* expected_tiles_dev, int* stale_out_dev, long long n_all, int sample_count,
                                 int64_t n_tiles, int64_t height, int64_t width, void* workspace_dev, size_t workspace_bytes,
                                 void* stream);
int sx_macenko_pfit_gather_packed(const long long* sums_global_dev, int stage, long long n_all, int sample_count,
                                  int64_t n_tiles, int64_t height, int64_t width, int share, const int* stale_flag_dev,
                                  unsigned* record_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int sx_macenko_pfit_finish_packed(const unsigned* gathered_records_dev, int world, int share, int stage, long long n_all,
                                  int sample_count, int64_t n_tiles, int64_t height, int64_t width, float* he_out_dev,
                                  float* max_c_out_dev, int* status_out_dev, void* workspace_dev, size_t workspace_bytes,
                                  void* stream);

/* ---------------------------------------------------------------- Reinhard -----------------------
 * Replaces stainx_cuda_torch.reinhard (bindings.cpp:32) with the numerics of ReinhardTorch
 * (torch_backend.py:304-355): LAB statistics pooled over the whole batch, unbiased std. */
size_t sx_reinhard_workspace_bytes(int64_t n_tiles, int64_t height, int64_t width);
/* ... with room for the 8-bit codes of a float32 batch (+ 3 bytes per pixel): sx_reinhard_transform(_ready) then leaves the tiles that
 * consist of grey levels (float(k) / 255 in every element) as bytes in its first pass and reads those in its second -- same bits, a
 * quarter of the second pass's input.  Every entry point also accepts the smaller workspace above and then runs without. */
size_t sx_reinhard_workspace_bytes_for(int dtype, int64_t n_tiles, int64_t height, int64_t width);
int sx_reinhard_fit(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                    float* mean_out_dev, float* std_out_dev, void* workspace_dev, size_t workspace_bytes,
                    void* stream);
int sx_reinhard_transform(const void* images_dev, void* out_dev, int dtype, int64_t n_tiles, int64_t height,
                          int64_t width, const float* ref_mean_dev, const float* ref_std_dev,
                          void* workspace_dev, size_t workspace_bytes, void* stream);

/* The transform for a workspace in the READY state: zero-filled once by sx_reinhard_workspace_init() (or by the caller), and since then
 * only touched by completed calls of this section -- each leaves it ready again.  It skips the launch that clears the arrival
 * counters in front of the statistics pass (~4 us of a 130 us call on 64 x 3 x 512 x 512 float32).  The plain calls above accept ANY
 * workspace contents and leave it ready as well.  A ready call on a workspace that was not ready is noticed on the device (the apply
 * pass does not find the statistics of its own statistics pass): bit 0 of the uint32 at byte sx_reinhard_workspace_status_offset()
 * is set; the output of such a call is not to be used.  The state a ready call relies on (the arrival counters) lies at a place that does
 * NOT depend on the batch's shape (batches of up to 4096 tiles; a larger batch is served as by the plain call), so calls of different
 * shapes may alternate on one ready workspace -- also when replayed from a captured graph.  No reference counterpart (the reference
 * keeps no state between calls). */
int sx_reinhard_workspace_init(void* workspace_dev, size_t workspace_bytes, void* stream);
size_t sx_reinhard_workspace_status_offset(void);
int sx_reinhard_transform_ready(const void* images_dev, void* out_dev, int dtype, int64_t n_tiles, int64_t height,
                                int64_t width, const float* ref_mean_dev, const float* ref_std_dev,
                                void* workspace_dev, size_t workspace_bytes, void* stream);

/* Batch statistics pooled across ranks: sx_reinhard_sums writes 6 fp64 local sums (sum and sum of squares, per
 * channel, of the quantities LAB is affine in: f_y, f_x - f_y, f_y - f_z; opaque to the caller) -> all-reduce(SUM) -> sx_reinhard_apply normalises with the global sums over
 * n_total_pixels = pixels per channel over all ranks. */
int sx_reinhard_sums(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                     double* sums_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int sx_reinhard_apply(const void* images_dev, void* out_dev, int dtype, int64_t n_tiles, int64_t height,
                      int64_t width, const double* sums_dev, double n_total_pixels, const float* ref_mean_dev,
                      const float* ref_std_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- Histogram matching -------------
 * Replaces stainx_cuda_torch.histogram_matching (bindings.cpp:31) with the numerics of
 * HistogramMatchingTorch (torch_backend.py:134-301).  `channels_last` != 0: images are (N,H,W,3).
 *   ref_hist_dev  3 x 256 floats (per-channel normalised reference histograms)
 *   sx_hm_fit writes those; sx_hm_transform also leaves the pooled integer source histogram
 *   (3 x 256 uint32) and the float LUT (3 x 256) at the start of the workspace for inspection. */
size_t sx_hm_workspace_bytes(int64_t n_tiles, int64_t height, int64_t width);
int sx_hm_fit(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
              int channels_last, float* hist_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int sx_hm_transform(const void* images_dev, void* out_dev, int dtype, int64_t n_tiles, int64_t height,
                    int64_t width, int channels_last, const float* ref_hist_dev, void* workspace_dev,
                    size_t workspace_bytes, void* stream);

/* The same three calls for a workspace in the READY state: zero-filled once by sx_hm_workspace_init() (or by the caller), and since then
 * only touched by completed calls of this section -- each of them leaves the workspace ready again (the kernel that reads the
 * histogram counters writes zeros back).  They skip the clearing launch in front of the histogram pass (~5 us of a 115 us call on
 * 64 x 3 x 1024 x 1024 uint8).  The plain calls above accept ANY workspace contents and leave it ready as well.  A *_ready transform
 * on a workspace that was not ready is noticed on the device (its counters do not add up to the pixels counted): bit 0 of the uint32
 * at byte sx_hm_workspace_status_offset() of the workspace is set and stays set until sx_hm_workspace_init(); the output of such a
 * call is not to be used.  No reference counterpart: the reference allocates its histogram inside every call
 * (torch_backend.py:139, histogram_matching.cu:49-81). */
int sx_hm_workspace_init(void* workspace_dev, size_t workspace_bytes, void* stream);
size_t sx_hm_workspace_status_offset(void);
#ifdef SX_DIAG
/* Diagnostic build: planar uint8 batches of at least 32 MB take the transform entry points above in ONE launch (workgroups keep part of the batch in
 * registers between counting and applying: same bits, measured slower than the two kernels -- DESIGN.md section 5; SX_HM_RESIDENT=0 in the
 * environment switches it off).  The uint32 at byte sx_hm_workspace_parity_offset() of the workspace toggles with every call that took
 * that form; sx_debug_hm_stamp_offset(): its phase stamps. */
size_t sx_hm_workspace_parity_offset(void);
size_t sx_debug_hm_stamp_offset(void);
#endif
int sx_hm_fit_ready(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                    int channels_last, float* hist_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int sx_hm_transform_ready(const void* images_dev, void* out_dev, int dtype, int64_t n_tiles, int64_t height,
                          int64_t width, int channels_last, const float* ref_hist_dev, void* workspace_dev,
                          size_t workspace_bytes, void* stream);
int sx_hm_counts_ready(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                       int channels_last, unsigned long long* counts_out_dev, void* workspace_dev,
                       size_t workspace_bytes, void* stream);

/* Source histogram pooled across ranks: sx_hm_counts writes the local 3 x 256 u64 counts -> all-reduce(SUM)
 * -> sx_hm_apply builds the LUT from the global counts (n_total_pixels per channel over all ranks). */
int sx_hm_counts(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                 int channels_last, unsigned long long* counts_out_dev, void* workspace_dev,
                 size_t workspace_bytes, void* stream);
int sx_hm_apply(const void* images_dev, void* out_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                int channels_last, const unsigned long long* counts_dev, double n_total_pixels,
                const float* ref_hist_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STAINX_HIP_H */
