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

/* flags for sx_macenko_transform */
#define SX_MACENKO_NORMALIZE_0_1 1u /* fuse `result / 255.0` (normalizers/_template.py:111-112); u8 input -> f32 output */

int sx_version(void);
const char* sx_last_error_string(void);

/* ---------------------------------------------------------------- Macenko ------------------------
 * Replaces stainx_cuda_torch.macenko (bindings.cpp:33; src/stainx_cuda_torch/csrc/macenko.cu:67-266)
 * with the numerics of MacenkoTorch.transform (torch_backend.py:521-560).
 *   images_dev        (N,3,H,W) `dtype`; u8 is [0,255], floats are taken as [0,1] as is
 *   out_dev           (N,3,H,W) same dtype (f32 when dtype==SX_U8 and SX_MACENKO_NORMALIZE_0_1)
 *   stain_matrix_dev  6 floats, row-major (3,2)      target_max_conc_dev  2 floats
 */
size_t sx_macenko_workspace_bytes(int64_t n_tiles, int64_t height, int64_t width);

int sx_macenko_transform(const void* images_dev, void* out_dev, int dtype, int64_t n_tiles, int64_t height,
                         int64_t width, const float* stain_matrix_dev, const float* target_max_conc_dev,
                         unsigned flags, void* workspace_dev, size_t workspace_bytes, void* stream);

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

/* ---------------------------------------------------------------- Reinhard -----------------------
 * Replaces stainx_cuda_torch.reinhard (bindings.cpp:32) with the numerics of ReinhardTorch
 * (torch_backend.py:304-355): LAB statistics pooled over the whole batch, unbiased std. */
size_t sx_reinhard_workspace_bytes(int64_t n_tiles, int64_t height, int64_t width);
int sx_reinhard_fit(const void* images_dev, int dtype, int64_t n_tiles, int64_t height, int64_t width,
                    float* mean_out_dev, float* std_out_dev, void* workspace_dev, size_t workspace_bytes,
                    void* stream);
int sx_reinhard_transform(const void* images_dev, void* out_dev, int dtype, int64_t n_tiles, int64_t height,
                          int64_t width, const float* ref_mean_dev, const float* ref_std_dev,
                          void* workspace_dev, size_t workspace_bytes, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* STAINX_HIP_H */
