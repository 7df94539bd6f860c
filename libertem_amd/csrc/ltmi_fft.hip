// ltmi_fft.hip -- per-frame Fourier-space operators.
//
// CrystallinityUDF.process_frame (reference udf/crystallinity.py:73-79):
//     intensity[f] = sum( abs(rfft2(frame * real_mask)) * half_fourier_mask )
// for a whole tile per call: a conversion kernel (native pixels -> f32, times the real-space mask),
// ONE batched 2D real-to-complex hipFFT per batch of frames, and a fused |F| * mask reduction that
// reads the half spectrum once and writes one float per frame.  The spectrum never leaves the
// plan's workspace.
#include "ltmi_common.h"
#include <hipfft/hipfft.h>
#include <algorithm>
#include <new>

struct ltmi_fft_plan {
    int device = 0;
    int h = 0, w = 0, wc = 0;          // frame shape, complex columns w/2+1
    int batch = 0;                     // frames per hipFFT execution
    hipfftHandle plan = 0;
    bool have_plan = false;
    float *real_buf = nullptr;         // (batch, h, w) f32
    hipfftComplex *spec = nullptr;     // (batch, h, wc) c64
    hipStream_t bound_stream = nullptr;
    bool stream_bound = false;
    float *mask_t = nullptr;           // workspace of k_cryst_fused: both masks in lane order (ltmi_cryst.hip)
    void *corr_ws = nullptr;           // ... of the corrections inside its row stage: dark map, patch lists / values
    size_t corr_ws_bytes = 0;
    int n_cu = 0;
    bool fused_ok = false;             // 256 x 256 frames: k_cryst_fused (ltmi_cryst.hip) unless LTMI_FFT_FUSED=0
    char last_kernel[128] = {0};
};

namespace ltmi {

static const char *fft_err(hipfftResult r) {
    switch (r) {
        case HIPFFT_SUCCESS: return "success";
        case HIPFFT_INVALID_PLAN: return "invalid plan";
        case HIPFFT_ALLOC_FAILED: return "allocation failed";
        case HIPFFT_INVALID_VALUE: return "invalid value";
        case HIPFFT_INTERNAL_ERROR: return "internal error";
        case HIPFFT_EXEC_FAILED: return "exec failed";
        case HIPFFT_SETUP_FAILED: return "setup failed";
        case HIPFFT_INVALID_SIZE: return "invalid size";
        default: return "hipfft error";
    }
}

// frames (native dtype, ld elements apart) -> contiguous f32, times the optional real-space mask.
// A thread converts 8 consecutive pixels (one 16-B load for 2-byte pixels, two 16-B stores).
// With dark / gain (detector corrections fused, reference io/corrections/detector.py:17-101):
// v = (float)(((double)x - dark) * gain), the arithmetic of ltmi_correct with a float32 result, in
// the same pass; the dead-pixel patches follow in k_fft_repair.
template <typename T>
__global__ void __launch_bounds__(256)
k_fft_prepare(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
              const float *__restrict__ real_mask, float *__restrict__ out, int vec_ok,
              const double *__restrict__ dark, const double *__restrict__ gain) {
    const int64_t f = blockIdx.y;
    const T *src = tile + f * ld;
    float *dst = out + f * n_px;
    const int64_t n8 = vec_ok ? n_px / 8 : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        T raw[8];
        typedef T __attribute__((ext_vector_type(8))) vec_t;
        *(vec_t *)raw = __builtin_nontemporal_load((const vec_t *)(src + i * 8));
        float v[8];
        if (dark || gain) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double d = dark ? dark[i * 8 + j] : 0.0;
                const double g = gain ? gain[i * 8 + j] : 1.0;
                v[j] = (float)(((double)raw[j] - d) * g);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)raw[j];
        }
        if (real_mask) {
            const float4 m0 = *(const float4 *)(real_mask + i * 8);
            const float4 m1 = *(const float4 *)(real_mask + i * 8 + 4);
            v[0] *= m0.x; v[1] *= m0.y; v[2] *= m0.z; v[3] *= m0.w;
            v[4] *= m1.x; v[5] *= m1.y; v[6] *= m1.z; v[7] *= m1.w;
        }
        *(float4 *)(dst + i * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4 *)(dst + i * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    for (int64_t p = n8 * 8 + (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_px;
         p += (int64_t)gridDim.x * 256) {
        float v = (dark || gain)
                      ? (float)(((double)src[p] - (dark ? dark[p] : 0.0)) * (gain ? gain[p] : 1.0))
                      : (float)src[p];
        if (real_mask) v *= real_mask[p];
        dst[p] = v;
    }
}

// The corrected conversion with dark / gain held in registers across frames: a thread owns 8
// consecutive pixels and walks `frames_per_block` frames (the per-pixel float64 dark and gain values
// would otherwise be re-read from L2 for every frame: 16 B per 2-byte pixel).
template <typename T>
__global__ void __launch_bounds__(256)
k_fft_prepare_corr(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
                   const float *__restrict__ real_mask, float *__restrict__ out,
                   const double *__restrict__ dark, const double *__restrict__ gain,
                   int frames_per_block) {
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (p0 >= n_px) return;                  // (n_px % 8 == 0 on this path)
    double d[8], g[8];
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        d[j] = dark ? dark[p0 + j] : 0.0;
        g[j] = gain ? gain[p0 + j] : 1.0;
        m[j] = real_mask ? real_mask[p0 + j] : 1.f;
    }
    const int64_t f0 = (int64_t)blockIdx.y * frames_per_block;
    const int64_t f1 = min<int64_t>(n_frames, f0 + frames_per_block);
    typedef T __attribute__((ext_vector_type(8))) vec_t;
#pragma unroll 2
    for (int64_t f = f0; f < f1; ++f) {
        const vec_t x = __builtin_nontemporal_load((const vec_t *)(tile + f * ld + p0));
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (float)(((double)x[j] - d[j]) * g[j]);
            if (real_mask) v[j] *= m[j];
        }
        float *dst = out + f * n_px + p0;
        *(float4 *)dst = make_float4(v[0], v[1], v[2], v[3]);
        *(float4 *)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// Dead pixels: buf[f, e] = mean over the good neighbours of e of the CORRECTED pixel, times the
// real-space mask at e -- recomputed from the raw tile (<= max_env pixels per patch), so that the
// prepare pass needs no second sweep over the frames.  One thread per (frame, excluded pixel).
template <typename T>
__global__ void __launch_bounds__(256)
k_fft_repair(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
             const double *__restrict__ dark, const double *__restrict__ gain,
             const float *__restrict__ real_mask, const int32_t *__restrict__ excl,
             const int32_t *__restrict__ env, const int32_t *__restrict__ cnt, int n_excl,
             int max_env, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_frames * n_excl) return;
    const int64_t f = i / n_excl;
    const int e = (int)(i % n_excl);
    const int c = cnt[e];
    if (c <= 0) return;
    const T *src = tile + f * ld;
    double acc = 0.0;
    for (int j = 0; j < c; ++j) {
        const int r = env[(int64_t)e * max_env + j];
        // the neighbour as the float32 value the corrected tile holds (ltmi_correct rounds once)
        acc += (double)(float)(((double)src[r] - (dark ? dark[r] : 0.0)) * (gain ? gain[r] : 1.0));
    }
    float v = (float)(acc / (double)c);
    const int p = excl[e];
    if (real_mask) v *= real_mask[p];
    out[f * n_px + p] = v;
}

// out[f] (+)= sum_k |spec[f, k]| * mask[k] over the rows [0, row_lo) and [row_hi, h) and the columns
// [0, n_cols) of the half spectrum: the bounding box of the (fft-shifted) ring -- everything else
// of the mask is zero and is never read.  One block per frame.
__global__ void __launch_bounds__(256)
k_abs_dot(const hipfftComplex *__restrict__ spec, int h, int wc, const float *__restrict__ mask,
          int row_lo, int row_hi, int n_cols, float *__restrict__ out, int accumulate) {
    const int64_t f = blockIdx.x;
    const hipfftComplex *s = spec + f * (int64_t)h * wc;
    const int n_box = (row_lo + (h - row_hi)) * n_cols;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_box; i += 256) {
        const int r = i / n_cols, kx = i - r * n_cols;
        const int ky = r < row_lo ? r : row_hi + (r - row_lo);
        const float m = mask[(int64_t)ky * wc + kx];
        if (m != 0.f) {
            const hipfftComplex c = s[(int64_t)ky * wc + kx];
            acc += sqrtf(c.x * c.x + c.y * c.y) * m;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = part[0] + part[1] + part[2] + part[3];
        out[f] = accumulate ? out[f] + v : v;
    }
}

struct FftCorr {                   // detector corrections fused into the prepare pass (all optional)
    const double *dark = nullptr, *gain = nullptr;
    const int32_t *excl = nullptr, *env = nullptr, *cnt = nullptr;
    int n_excl = 0, max_env = 0;
};

template <typename T>
static int run_prepare(const void *tile, int64_t ld, int64_t n, int64_t n_px, const float *real_mask,
                       float *out, hipStream_t stream, const FftCorr &corr) {
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_px / 8 + 255) / 256, 32));
    const int vec_ok = (sizeof(T) <= 4) && ((uintptr_t)tile % 16 == 0) &&
                       ((ld * (int64_t)sizeof(T)) % 16 == 0) && (n_px % 8 == 0);
    if ((corr.dark || corr.gain) && vec_ok) {
        const int fpb = 16;
        const unsigned bx = (unsigned)((n_px / 8 + 255) / 256);
        hipLaunchKernelGGL((k_fft_prepare_corr<T>), dim3(bx, (unsigned)((n + fpb - 1) / fpb)),
                           dim3(256), 0, stream, (const T *)tile, ld, n, n_px, real_mask, out,
                           corr.dark, corr.gain, fpb);
    } else {
        hipLaunchKernelGGL((k_fft_prepare<T>), dim3(gx, (unsigned)n), dim3(256), 0, stream,
                           (const T *)tile, ld, n, n_px, real_mask, out, vec_ok, corr.dark,
                           corr.gain);
    }
    if (corr.n_excl > 0) {
        const int64_t nt = n * corr.n_excl;
        hipLaunchKernelGGL((k_fft_repair<T>), dim3((unsigned)((nt + 255) / 256)), dim3(256), 0,
                           stream, (const T *)tile, ld, n, n_px, corr.dark, corr.gain, real_mask,
                           corr.excl, corr.env, corr.cnt, corr.n_excl, corr.max_env, out);
    }
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

}  // namespace ltmi

using namespace ltmi;

// (batch x frame) float32 workspace of a plan, created on first use
static int real_buf_ready(ltmi_fft_plan *p) {
    if (p->real_buf) return LTMI_OK;
    hipError_t e = hipMalloc((void **)&p->real_buf, (size_t)p->batch * p->h * p->w * sizeof(float));
    if (e != hipSuccess) {
        p->real_buf = nullptr;
        LTMI_FAIL((int)e, "ltmi_fft_plan: workspace allocation failed: %s", hipGetErrorString(e));
    }
    return LTMI_OK;
}

// (batch x half spectrum) complex workspace of a plan, created on first use: hipFFT's output, or the columns
// the 512 x 512 kernels pass from the row to the column transforms
static int spec_ready(ltmi_fft_plan *p) {
    if (p->spec) return LTMI_OK;
    hipError_t e = hipMalloc((void **)&p->spec, (size_t)p->batch * p->h * p->wc * sizeof(hipfftComplex));
    if (e != hipSuccess) {
        p->spec = nullptr;
        LTMI_FAIL((int)e, "ltmi_fft_plan: workspace allocation failed: %s", hipGetErrorString(e));
    }
    return LTMI_OK;
}

// the hipFFT plan + workspace of a plan, created on first use
static int hipfft_route_ready(ltmi_fft_plan *p) {
    if (p->have_plan) return LTMI_OK;
    {
        const int rc = real_buf_ready(p);
        if (rc != LTMI_OK) return rc;
    }
    int n[2] = {p->h, p->w};
    hipfftResult r = hipfftPlanMany(&p->plan, 2, n, nullptr, 1, p->h * p->w, nullptr, 1, p->h * p->wc,
                                    HIPFFT_R2C, p->batch);
    if (r != HIPFFT_SUCCESS)
        LTMI_FAIL(LTMI_E_INVALID, "hipfftPlanMany(%d x %d, batch %d) failed: %s", p->h, p->w, p->batch,
                  fft_err(r));
    {
        const int rc = spec_ready(p);
        if (rc != LTMI_OK) {
            (void)hipfftDestroy(p->plan);
            return rc;
        }
    }
    p->have_plan = true;
    return LTMI_OK;
}

extern "C" int ltmi_fft_plan_create(int device, int sig_h, int sig_w, int max_batch,
                                    ltmi_fft_plan **out) {
    if (!out || sig_h <= 0 || sig_w <= 0 || max_batch <= 0)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_fft_plan_create: bad arguments (%d x %d, batch %d)", sig_h,
                  sig_w, max_batch);
    LTMI_HIP(hipSetDevice(device));
    ltmi_fft_plan *p = new (std::nothrow) ltmi_fft_plan();
    if (!p) LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
    p->device = device;
    p->h = sig_h;
    p->w = sig_w;
    p->wc = sig_w / 2 + 1;
    p->batch = max_batch;
    hipError_t e = hipSuccess;
    if (cryst_fused_shape(sig_h, sig_w)) {
        const char *env = getenv("LTMI_FFT_FUSED");
        p->fused_ok = !(env && env[0] == '0');
        e = hipMalloc((void **)&p->mask_t, (size_t)cryst_fused_workspace_floats(sig_h, sig_w) * sizeof(float));
        if (e == hipSuccess) e = hipDeviceGetAttribute(&p->n_cu, hipDeviceAttributeMultiprocessorCount, device);
        if (e != hipSuccess) {
            if (p->mask_t) (void)hipFree(p->mask_t);
            delete p;
            LTMI_FAIL((int)e, "ltmi_fft_plan_create: workspace allocation failed: %s", hipGetErrorString(e));
        }
    }
    // frames the fused kernel takes never need the hipFFT plan and its (batch x frame) workspace: both are
    // created by the first call that does (hipfft_route_ready)
    if (!p->fused_ok) {
        const int rc = hipfft_route_ready(p);
        if (rc != LTMI_OK) {
            if (p->mask_t) (void)hipFree(p->mask_t);
            delete p;
            return rc;
        }
    }
    *out = p;
    return LTMI_OK;
}

extern "C" int ltmi_fft_plan_destroy(ltmi_fft_plan *p) {
    if (!p) return LTMI_OK;
    (void)hipSetDevice(p->device);
    if (p->have_plan) (void)hipfftDestroy(p->plan);
    if (p->real_buf) (void)hipFree(p->real_buf);
    if (p->spec) (void)hipFree(p->spec);
    if (p->corr_ws) (void)hipFree(p->corr_ws);
    if (p->mask_t) (void)hipFree(p->mask_t);
    delete p;
    return LTMI_OK;
}

extern "C" const char *ltmi_fft_plan_last_kernel(const ltmi_fft_plan *p) {
    return p ? p->last_kernel : "";
}

static int crystallinity_impl(ltmi_fft_plan *p, const void *tile, int tile_dtype,
                              int64_t n_frames, int64_t ld_tile, const float *real_mask,
                              const float *half_mask, int row_lo, int row_hi, int n_cols,
                              float *out, int accumulate, void *stream_, const FftCorr &corr);

extern "C" int ltmi_crystallinity(ltmi_fft_plan *p, const void *tile, int tile_dtype,
                                  int64_t n_frames, int64_t ld_tile, const float *real_mask,
                                  const float *half_mask, int row_lo, int row_hi, int n_cols,
                                  float *out, int accumulate, void *stream_) {
    return crystallinity_impl(p, tile, tile_dtype, n_frames, ld_tile, real_mask, half_mask, row_lo,
                              row_hi, n_cols, out, accumulate, stream_, FftCorr());
}

extern "C" int ltmi_crystallinity_corrected(
    ltmi_fft_plan *p, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld_tile,
    const double *dark, const double *gain, const int32_t *excl, const int32_t *env,
    const int32_t *cnt, int n_excl, int max_env, const float *real_mask, const float *half_mask,
    int row_lo, int row_hi, int n_cols, float *out, int accumulate, void *stream_) {
    if (n_excl < 0 || max_env < 0 || (n_excl > 0 && (!excl || !env || !cnt)))
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_crystallinity_corrected: bad repair tables");
    FftCorr corr;
    corr.dark = dark;
    corr.gain = gain;
    corr.excl = excl;
    corr.env = env;
    corr.cnt = cnt;
    corr.n_excl = n_excl;
    corr.max_env = max_env;
    return crystallinity_impl(p, tile, tile_dtype, n_frames, ld_tile, real_mask, half_mask, row_lo,
                              row_hi, n_cols, out, accumulate, stream_, corr);
}

// frames [0, n) of `src` -> p->real_buf: float32, corrected, times the real-space mask
static int prepare_batch(ltmi_fft_plan *p, const void *src, int tile_dtype, int64_t ld_tile, int64_t n,
                         int64_t n_px, const float *real_mask, hipStream_t stream, const FftCorr &corr) {
    switch (tile_dtype) {
        case LTMI_BOOL:
        case LTMI_U8: return run_prepare<uint8_t>(src, ld_tile, n, n_px, real_mask, p->real_buf, stream, corr);
        case LTMI_I8: return run_prepare<int8_t>(src, ld_tile, n, n_px, real_mask, p->real_buf, stream, corr);
        case LTMI_U16: return run_prepare<uint16_t>(src, ld_tile, n, n_px, real_mask, p->real_buf, stream, corr);
        case LTMI_I16: return run_prepare<int16_t>(src, ld_tile, n, n_px, real_mask, p->real_buf, stream, corr);
        case LTMI_U32: return run_prepare<uint32_t>(src, ld_tile, n, n_px, real_mask, p->real_buf, stream, corr);
        case LTMI_I32: return run_prepare<int32_t>(src, ld_tile, n, n_px, real_mask, p->real_buf, stream, corr);
        case LTMI_F32: return run_prepare<float>(src, ld_tile, n, n_px, real_mask, p->real_buf, stream, corr);
        case LTMI_F64: return run_prepare<double>(src, ld_tile, n, n_px, real_mask, p->real_buf, stream, corr);
        default:
            LTMI_FAIL(LTMI_E_DTYPE, "ltmi_crystallinity: unsupported tile dtype %s", dtype_name(tile_dtype));
    }
}

static int crystallinity_impl(ltmi_fft_plan *p, const void *tile, int tile_dtype,
                              int64_t n_frames, int64_t ld_tile, const float *real_mask,
                              const float *half_mask, int row_lo, int row_hi, int n_cols,
                              float *out, int accumulate, void *stream_, const FftCorr &corr) {
    if (!p) LTMI_FAIL(LTMI_E_INVALID, "ltmi_crystallinity: null plan");
    if (n_frames < 0) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_crystallinity: negative frame count");
    if (n_frames == 0) return LTMI_OK;
    const int64_t n_px = (int64_t)p->h * p->w;
    if (ld_tile < n_px) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_crystallinity: ld_tile < frame size");
    if (!tile || !half_mask || !out) LTMI_FAIL(LTMI_E_INVALID, "ltmi_crystallinity: null pointer");
    LTMI_HIP(hipSetDevice(p->device));
    hipStream_t stream = (hipStream_t)stream_;
    const size_t esz = (size_t)dtype_size(tile_dtype);
    if (row_lo < 0 || row_hi < row_lo || row_hi > p->h || n_cols < 0 || n_cols > p->wc)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_crystallinity: bad mask bounding box (%d, %d, %d)", row_lo,
                  row_hi, n_cols);
    if (p->fused_ok && !corr.dark && !corr.gain && corr.n_excl == 0) {
        bool handled = false;
        if (cryst_fused_needs_gbuf(p->h, p->w, n_cols)) {
            const int rc1 = spec_ready(p);
            if (rc1 != LTMI_OK) return rc1;
        }
        const int rc = cryst_fused(tile, tile_dtype, n_frames, ld_tile, p->h, p->w, real_mask, half_mask,
                                   n_cols, p->mask_t, p->spec, p->batch, out, accumulate, p->n_cu, stream,
                                   &handled);
        if (rc != LTMI_OK) return rc;
        if (handled) {
            if (cryst_fused_needs_gbuf(p->h, p->w, n_cols))
                snprintf(p->last_kernel, sizeof(p->last_kernel), "k_cryst_rows%d<%s%s> + k_cryst_cols%d columns=%d", p->w,
                         dtype_name(tile_dtype), real_mask ? ",mask" : "", p->h, n_cols);
            else
                snprintf(p->last_kernel, sizeof(p->last_kernel), "k_cryst_fused%s<%s%s> columns=%d",
                         p->h == 128 ? "128" : "", dtype_name(tile_dtype), real_mask ? ",mask" : "", n_cols);
            return LTMI_OK;
        }
    }
    if (p->fused_ok && (corr.dark || corr.gain || corr.n_excl > 0) &&
        cryst_corr_takes(p->h, p->w, n_cols, tile_dtype, corr.n_excl) && !getenv("LTMI_CRYST_CORR_PASS")) {
        // dark / gain / dead-pixel patches inside the row stage of the fused kernels -- ONE pass over the raw pixels
        // (round 5; LTMI_CRYST_CORR_PASS=1 keeps the conversion pass: tests, bench).  Frames in chunks whose patch
        // values fit 64 MiB.
        const bool need_g = cryst_fused_needs_gbuf(p->h, p->w, n_cols);
        if (need_g) {
            const int rc1 = spec_ready(p);
            if (rc1 != LTMI_OK) return rc1;
        }
        const int64_t chunk = corr.n_excl > 0 ? std::max<int64_t>(1024, ((int64_t)64 << 20) / (4 * corr.n_excl)) : n_frames;
        const size_t need = (size_t)cryst_corr_workspace_bytes(p->h, p->w, std::min(chunk, n_frames), corr.n_excl);
        if (need > p->corr_ws_bytes) {
            if (p->corr_ws) {
                LTMI_HIP(hipStreamSynchronize(stream));
                (void)hipFree(p->corr_ws);
                p->corr_ws = nullptr;
                p->corr_ws_bytes = 0;
            }
            LTMI_HIP(hipMalloc(&p->corr_ws, need));
            p->corr_ws_bytes = need;
        }
        bool all = true;
        for (int64_t f0 = 0; f0 < n_frames && all; f0 += chunk) {
            const int64_t n = std::min(chunk, n_frames - f0);
            bool handled = false;
            const int rc = cryst_fused_corrected((const char *)tile + (size_t)f0 * ld_tile * esz, tile_dtype, n, ld_tile,
                                                 p->h, p->w, corr.dark, corr.gain, corr.excl, corr.env, corr.cnt,
                                                 corr.n_excl, corr.max_env, real_mask, half_mask, n_cols, p->mask_t,
                                                 p->spec, p->batch, p->corr_ws, out + f0, accumulate, p->n_cu, stream,
                                                 &handled);
            if (rc != LTMI_OK) return rc;
            if (!handled) {
                if (f0 != 0) LTMI_FAIL(LTMI_E_INVALID, "ltmi_crystallinity: the fused kernel refused a later chunk");
                all = false;
            }
        }
        if (all) {
            if (need_g)
                snprintf(p->last_kernel, sizeof(p->last_kernel), "k_cryst_rows%d<%s,corrected%s> + k_cryst_cols%d columns=%d patches=%d",
                         p->w, dtype_name(tile_dtype), real_mask ? ",mask" : "", p->h, n_cols, corr.n_excl);
            else
                snprintf(p->last_kernel, sizeof(p->last_kernel), "k_cryst_fused%s<%s,corrected%s> columns=%d patches=%d",
                         p->h == 128 ? "128" : "", dtype_name(tile_dtype), real_mask ? ",mask" : "", n_cols, corr.n_excl);
            return LTMI_OK;
        }
    }
    if (p->fused_ok && cryst_fused_takes(p->h, p->w, n_cols)) {
        // corrected frames (and float64 pixels): the conversion pass writes them as float32 -- dark / gain /
        // dead-pixel patches / real-space mask applied, 6 - 12 bytes of traffic per pixel -- and the fused
        // kernel takes those instead of two rocFFT passes over a (batch x half spectrum) workspace
        const int rc0 = real_buf_ready(p);
        if (rc0 != LTMI_OK) return rc0;
        if (cryst_fused_needs_gbuf(p->h, p->w, n_cols)) {
            const int rc1 = spec_ready(p);
            if (rc1 != LTMI_OK) return rc1;
        }
        for (int64_t f0 = 0; f0 < n_frames; f0 += p->batch) {
            const int64_t n = std::min<int64_t>(p->batch, n_frames - f0);
            const void *src = (const char *)tile + (size_t)f0 * ld_tile * esz;
            int rc = prepare_batch(p, src, tile_dtype, ld_tile, n, n_px, real_mask, stream, corr);
            if (rc != LTMI_OK) return rc;
            bool handled = false;
            rc = cryst_fused(p->real_buf, LTMI_F32, n, n_px, p->h, p->w, nullptr, half_mask, n_cols, p->mask_t,
                             p->spec, p->batch, out + f0, accumulate, p->n_cu, stream, &handled);
            if (rc != LTMI_OK) return rc;
            if (!handled) LTMI_FAIL(LTMI_E_INVALID, "ltmi_crystallinity: the fused kernel refused its own workspace");
        }
        if (cryst_fused_needs_gbuf(p->h, p->w, n_cols))
            snprintf(p->last_kernel, sizeof(p->last_kernel), "k_fft_prepare<%s> + k_cryst_rows%d<float32> + k_cryst_cols%d columns=%d",
                     dtype_name(tile_dtype), p->w, p->h, n_cols);
        else
            snprintf(p->last_kernel, sizeof(p->last_kernel), "k_fft_prepare<%s> + k_cryst_fused%s<float32> columns=%d",
                     dtype_name(tile_dtype), p->h == 128 ? "128" : "", n_cols);
        return LTMI_OK;
    }
    snprintf(p->last_kernel, sizeof(p->last_kernel), "hipfft_r2c<%s> batch=%d", dtype_name(tile_dtype),
             p->batch);
    {
        const int rc = hipfft_route_ready(p);
        if (rc != LTMI_OK) return rc;
    }
    if (!p->stream_bound || p->bound_stream != stream) {
        hipfftResult r = hipfftSetStream(p->plan, stream);
        if (r != HIPFFT_SUCCESS) LTMI_FAIL(LTMI_E_INVALID, "hipfftSetStream failed: %s", fft_err(r));
        p->bound_stream = stream;
        p->stream_bound = true;
    }
    for (int64_t f0 = 0; f0 < n_frames; f0 += p->batch) {
        const int64_t n = std::min<int64_t>(p->batch, n_frames - f0);
        const void *src = (const char *)tile + (size_t)f0 * ld_tile * esz;
        const int rc = prepare_batch(p, src, tile_dtype, ld_tile, n, n_px, real_mask, stream, corr);
        if (rc != LTMI_OK) return rc;
        if (n < p->batch) {
            // the plan always transforms `batch` frames: clear the unused tail once so that it
            // holds finite numbers (its spectra are never read)
            LTMI_HIP(hipMemsetAsync(p->real_buf + n * n_px, 0,
                                    (size_t)(p->batch - n) * n_px * sizeof(float), stream));
        }
        hipfftResult r = hipfftExecR2C(p->plan, p->real_buf, p->spec);
        if (r != HIPFFT_SUCCESS) LTMI_FAIL(LTMI_E_INVALID, "hipfftExecR2C failed: %s", fft_err(r));
        hipLaunchKernelGGL(k_abs_dot, dim3((unsigned)n), dim3(256), 0, stream,
                           (const hipfftComplex *)p->spec, p->h, p->wc, half_mask, row_lo, row_hi,
                           n_cols, out + f0, accumulate);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}
