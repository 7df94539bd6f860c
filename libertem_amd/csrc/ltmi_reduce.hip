// Reductions for SumUDF / SumSigUDF and the sig-buffer merge (gfx950).
//
//   ltmi_sum_sig    : out[f] (+)= sum_p tile[f, p]     (src/libertem/udf/sumsigudf.py:30-39)
//   ltmi_sum_frames : out[p] (+)= sum_f tile[f, p]     (src/libertem/udf/sum.py:43-48)
//   ltmi_axpy       : dest[i] += src[i]                (src/libertem/udf/sum.py:50-52)
//
// All three are pure HBM streams: 16-byte coalesced loads, conversion in registers,
// wavefront (64-lane) shuffle reduction, no atomics (deterministic).
#include "ltmi_common.h"

namespace ltmi {

template <typename A> __device__ __forceinline__ A wave_sum(A v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename T, typename A> struct Cv {
    static __device__ __forceinline__ A from(T v) { return (A)v; }
};

// ---- per-frame sums ---------------------------------------------------------------------------
// one 256-thread block per frame; VEC elements per thread per step (16 bytes where aligned)
template <typename T, typename A, int VEC>
__global__ void __launch_bounds__(256)
k_sum_sig(const T *__restrict__ tile, int64_t ld, int64_t n_px, A *__restrict__ out,
          int accumulate) {
    __shared__ A red[4];
    const int64_t f = blockIdx.x;
    const T *row = tile + f * ld;
    A acc0 = 0, acc1 = 0;
    if (VEC > 1) {
        // (element-aligned only: rows of odd length start at any element boundary; the target has
        // unaligned access enabled, the load stays one global_load_dwordx4)
        typedef T vec_a __attribute__((ext_vector_type(VEC)));
        typedef vec_a vec_t __attribute__((aligned(sizeof(T))));
        const int64_t nvec = n_px / VEC;
        const vec_t *vrow = (const vec_t *)row;
        int64_t i = threadIdx.x;
        for (; i + 256 < nvec; i += 512) {
            const vec_t v0 = __builtin_nontemporal_load(vrow + i);
            const vec_t v1 = __builtin_nontemporal_load(vrow + i + 256);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { acc0 += (A)v0[e]; acc1 += (A)v1[e]; }
        }
        for (; i < nvec; i += 256) {
            const vec_t v0 = __builtin_nontemporal_load(vrow + i);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc0 += (A)v0[e];
        }
        for (int64_t p = nvec * VEC + threadIdx.x; p < n_px; p += 256) acc1 += (A)row[p];
    } else {
        for (int64_t p = threadIdx.x; p < n_px; p += 256) acc0 += (A)row[p];
    }
    A s = wave_sum<A>(acc0 + acc1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0] + red[1]) + (red[2] + red[3]);
        out[f] = accumulate ? out[f] + s : s;
    }
}

// complex frames: tile rows are (re, im) pairs of R; one block per frame, two sums (even / odd elements)
template <typename R, typename A>
__global__ void __launch_bounds__(256)
k_sum_sig_cplx(const R *__restrict__ tile, int64_t ld_r, int64_t n_px, A *__restrict__ out,
               int accumulate) {
    __shared__ A red[8];
    const int64_t f = blockIdx.x;
    const R *row = tile + f * ld_r;
    A re = 0, im = 0;
    for (int64_t p = threadIdx.x; p < n_px; p += 256) {
        re += (A)__builtin_nontemporal_load(row + 2 * p);
        im += (A)__builtin_nontemporal_load(row + 2 * p + 1);
    }
    re = wave_sum<A>(re);
    im = wave_sum<A>(im);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = re; red[4 + (threadIdx.x >> 6)] = im; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const A *r = red + 4 * threadIdx.x;
        const A s = (r[0] + r[1]) + (r[2] + r[3]);
        A &o = out[2 * f + threadIdx.x];
        o = accumulate ? o + s : s;
    }
}

// out[i * stride] (+)= (O)src[i]   (integers: two's complement truncation = NumPy's wrap-around);
// stride 2 = the real parts of a complex buffer, whose imaginary parts are zeroed unless accumulating
template <typename S, typename O>
__global__ void k_store_cast(const S *__restrict__ src, int64_t n, O *__restrict__ out, int stride,
                             int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    O &o = out[i * stride];
    o = accumulate ? (O)(o + (O)src[i]) : (O)src[i];
    if (stride == 2 && !accumulate) out[i * 2 + 1] = (O)0;
}

// ---- sum over frames ----------------------------------------------------------------------------
// grid = (pixel tiles of 256*VEC, frame splits). Each thread owns VEC consecutive pixels and walks
// its slab of frames; partial sums per split go to the workspace and are reduced in fixed order.
template <typename T, typename A, int VEC>
__global__ void __launch_bounds__(256)
k_sum_frames(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
             A *__restrict__ dst, int64_t dst_stride_split, int fsplit, int accumulate_direct) {
    typedef T vec_a __attribute__((ext_vector_type(VEC)));
    typedef vec_a vec_t __attribute__((aligned(sizeof(T))));     // rows at any element alignment
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (p0 >= n_px) return;
    const int64_t per = (n_frames + fsplit - 1) / fsplit;
    const int64_t f0 = (int64_t)blockIdx.y * per;
    const int64_t f1 = min(n_frames, f0 + per);
    A acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0;
    const bool full = (p0 + VEC <= n_px);
    if (full) {
        int64_t f = f0;
        for (; f + 3 < f1; f += 4) {
            vec_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = __builtin_nontemporal_load((const vec_t *)(tile + (f + u) * ld + p0));
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += (A)v[u][e];
        }
        for (; f < f1; ++f) {
            const vec_t v = __builtin_nontemporal_load((const vec_t *)(tile + f * ld + p0));
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += (A)v[e];
        }
    } else {
        for (int64_t f = f0; f < f1; ++f)
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (p0 + e < n_px) acc[e] += (A)tile[f * ld + p0 + e];
    }
    A *d = dst + (int64_t)blockIdx.y * dst_stride_split;
#pragma unroll
    for (int e = 0; e < VEC; ++e)
        if (p0 + e < n_px) d[p0 + e] = accumulate_direct ? d[p0 + e] + acc[e] : acc[e];
}

template <typename A>
__global__ void k_reduce_splits(const A *__restrict__ ws, int fsplit, int64_t n, A *__restrict__ out,
                                int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    A s = accumulate ? out[i] : (A)0;
    for (int k0 = 0; k0 < fsplit; k0 += 8) {              // independent loads, added in the order of k
        A v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ws[(int64_t)min(k0 + u, fsplit - 1) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < fsplit) s += v[u];
    }
    out[i] = s;
}

// dest[r * ld_dest + c] += sign * src[r * ld_src + c]; ld_src == 0 broadcasts one row of `src`
template <typename A>
__global__ void k_add2d(A *__restrict__ dest, int64_t ld_dest, const A *__restrict__ src,
                        int64_t ld_src, int64_t rows, int64_t cols, int negate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols, c = i - r * cols;
    const A v = src[r * ld_src + c];
    A &d = dest[r * ld_dest + c];
    d = negate ? (A)(d - v) : (A)(d + v);
}

typedef unsigned int gu32x4 __attribute__((ext_vector_type(4)));

// dest[i, :] = src[idx[i], :], rows of `row_bytes` bytes (a multiple of sizeof(V))
template <typename V>
__global__ void __launch_bounds__(256)
k_gather_rows(const V *__restrict__ src, int64_t ld_src_v, const int64_t *__restrict__ idx,
              V *__restrict__ dest, int64_t row_v) {
    const int64_t r = blockIdx.y;
    const V *s = src + idx[r] * ld_src_v;
    V *d = dest + r * row_v;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_v;
         i += (int64_t)gridDim.x * blockDim.x)
        d[i] = __builtin_nontemporal_load(s + i);
}

static int frames_split(int64_t n_frames, int64_t n_px) {
    // independent of the tile dtype so that the workspace query and the launch always agree
    const int64_t px_blocks = (n_px + 2047) / 2048;
    int64_t want = (2048 + px_blocks - 1) / px_blocks;      // aim at >= 2048 workgroups
    want = std::max<int64_t>(1, std::min<int64_t>(want, n_frames / 8));
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, 256));
}

template <typename T> constexpr int vec_for() { return sizeof(T) >= 16 ? 1 : (int)(16 / sizeof(T)); }

template <typename T, typename A>
static int run_sum_sig(const void *tile, int64_t n_frames, int64_t n_px, int64_t ld, void *out,
                       int accumulate, hipStream_t stream) {
    constexpr int VEC = vec_for<T>();
    const bool aligned = vector_loads_ok(tile, ld, sizeof(T));
    if (aligned && VEC > 1)
        hipLaunchKernelGGL((k_sum_sig<T, A, VEC>), dim3((unsigned)n_frames), dim3(256), 0, stream,
                           (const T *)tile, ld, n_px, (A *)out, accumulate);
    else
        hipLaunchKernelGGL((k_sum_sig<T, A, 1>), dim3((unsigned)n_frames), dim3(256), 0, stream,
                           (const T *)tile, ld, n_px, (A *)out, accumulate);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

template <typename T, typename A>
static int run_sum_frames(const void *tile, int64_t n_frames, int64_t n_px, int64_t ld, void *out,
                          int accumulate, void *ws, hipStream_t stream) {
    constexpr int VECA = vec_for<T>();
    const bool aligned = vector_loads_ok(tile, ld, sizeof(T));
    const int vec = (aligned && VECA > 1) ? VECA : 1;
    const int fsplit = frames_split(n_frames, n_px);         // matches the workspace query
    dim3 grid((unsigned)((n_px + 256 * vec - 1) / (256 * vec)), (unsigned)fsplit);
    A *dst = fsplit > 1 ? (A *)ws : (A *)out;
    if (fsplit > 1 && !ws) LTMI_FAIL(LTMI_E_INVALID, "ltmi_sum_frames: workspace required");
    const int direct_acc = (fsplit == 1) ? accumulate : 0;
    if (vec > 1)
        hipLaunchKernelGGL((k_sum_frames<T, A, VECA>), grid, dim3(256), 0, stream, (const T *)tile,
                           ld, n_frames, n_px, dst, n_px, fsplit, direct_acc);
    else
        hipLaunchKernelGGL((k_sum_frames<T, A, 1>), grid, dim3(256), 0, stream, (const T *)tile, ld,
                           n_frames, n_px, dst, n_px, fsplit, direct_acc);
    LTMI_HIP(hipGetLastError());
    if (fsplit > 1) {
        hipLaunchKernelGGL((k_reduce_splits<A>), dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0,
                           stream, (const A *)ws, fsplit, n_px, (A *)out, accumulate);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}

static int tile_vec(int dt) {
    const int s = dtype_size(dt);
    return s >= 16 ? 1 : 16 / s;
}

}  // namespace ltmi

using namespace ltmi;

#define LTMI_DISPATCH_TILE(FN, A, ...)                                                          \
    switch (tile_dtype) {                                                                        \
        case LTMI_BOOL:                                                                          \
        case LTMI_U8: return FN<uint8_t, A>(__VA_ARGS__);                                       \
        case LTMI_I8: return FN<int8_t, A>(__VA_ARGS__);                                        \
        case LTMI_U16: return FN<uint16_t, A>(__VA_ARGS__);                                     \
        case LTMI_I16: return FN<int16_t, A>(__VA_ARGS__);                                      \
        case LTMI_U32: return FN<uint32_t, A>(__VA_ARGS__);                                     \
        case LTMI_I32: return FN<int32_t, A>(__VA_ARGS__);                                      \
        case LTMI_U64: return FN<uint64_t, A>(__VA_ARGS__);                                     \
        case LTMI_I64: return FN<int64_t, A>(__VA_ARGS__);                                      \
        case LTMI_F32: return FN<float, A>(__VA_ARGS__);                                        \
        case LTMI_F64: return FN<double, A>(__VA_ARGS__);                                       \
    }

extern "C" int ltmi_sum_sig(int device, const void *tile, int tile_dtype, int64_t n_frames,
                            int64_t n_px, int64_t ld_tile, void *out, int out_dtype, int accumulate,
                            void *stream_) {
    if (n_frames < 0 || n_px < 0 || ld_tile < n_px)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_sum_sig: bad shape");
    if (n_frames == 0) return LTMI_OK;
    if (!tile || !out) LTMI_FAIL(LTMI_E_INVALID, "ltmi_sum_sig: null pointer");
    LTMI_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    if (tile_dtype == LTMI_C64 || tile_dtype == LTMI_C128) {
        // complex frames -> complex sums (udf/sumsigudf.py:23: result_type(input, float32) keeps
        // complex64 / complex128): real and imaginary parts are summed separately
        const dim3 grid((unsigned)n_frames);
        if (tile_dtype == LTMI_C64 && out_dtype == LTMI_C64)
            hipLaunchKernelGGL((k_sum_sig_cplx<float, float>), grid, dim3(256), 0, stream,
                               (const float *)tile, 2 * ld_tile, n_px, (float *)out, accumulate);
        else if (tile_dtype == LTMI_C64 && out_dtype == LTMI_C128)
            hipLaunchKernelGGL((k_sum_sig_cplx<float, double>), grid, dim3(256), 0, stream,
                               (const float *)tile, 2 * ld_tile, n_px, (double *)out, accumulate);
        else if (tile_dtype == LTMI_C128 && out_dtype == LTMI_C128)
            hipLaunchKernelGGL((k_sum_sig_cplx<double, double>), grid, dim3(256), 0, stream,
                               (const double *)tile, 2 * ld_tile, n_px, (double *)out, accumulate);
        else
            LTMI_FAIL(LTMI_E_DTYPE, "ltmi_sum_sig: unsupported dtypes tile=%s out=%s",
                      dtype_name(tile_dtype), dtype_name(out_dtype));
        LTMI_HIP(hipGetLastError());
        return LTMI_OK;
    }
    if (out_dtype == LTMI_F32) {
        LTMI_DISPATCH_TILE(run_sum_sig, float, tile, n_frames, n_px, ld_tile, out, accumulate, stream)
    } else if (out_dtype == LTMI_F64) {
        LTMI_DISPATCH_TILE(run_sum_sig, double, tile, n_frames, n_px, ld_tile, out, accumulate, stream)
    }
    LTMI_FAIL(LTMI_E_DTYPE, "ltmi_sum_sig: unsupported dtypes tile=%s out=%s", dtype_name(tile_dtype),
              dtype_name(out_dtype));
}

// How ltmi_sum_frames computes for an output dtype (the tile dtype only selects between the two
// complex layouts, so the workspace query -- which does not know it -- takes the larger one):
//   float32 / float64 out : accumulate in that type, straight into `out`
//   complex out, complex tile : the same on 2 * n_px real columns
//   complex out, real tile    : real sums into a temporary, stored into the real parts
//   integer out  : accumulate in int64 (exact), store truncated = NumPy's wrap-around in the narrower
//                  type (udf/sum.py:38-48 with SumUDF(dtype=<integer>) on integer frames)
static bool is_int_dtype(int dt) { return dt >= LTMI_BOOL && dt <= LTMI_I64; }
static bool is_cplx_dtype(int dt) { return dt == LTMI_C64 || dt == LTMI_C128; }

static int64_t split_bytes(int64_t n_frames, int64_t n_cols, int acc_size) {
    const int fsplit = frames_split(n_frames, n_cols);
    return fsplit <= 1 ? 0 : (int64_t)fsplit * n_cols * acc_size;
}

extern "C" int64_t ltmi_sum_frames_workspace(int64_t n_frames, int64_t n_px, int out_dtype) {
    if (is_cplx_dtype(out_dtype)) {
        const int rs = dtype_size(out_dtype) / 2;
        return std::max(split_bytes(n_frames, 2 * n_px, rs),
                        split_bytes(n_frames, n_px, rs) + n_px * rs);
    }
    if (is_int_dtype(out_dtype)) return split_bytes(n_frames, n_px, 8) + n_px * 8;
    return split_bytes(n_frames, n_px, dtype_size(out_dtype));
}

template <typename S>
static int store_cast(const S *src, int64_t n, void *out, int out_dtype, int stride, int accumulate,
                      hipStream_t stream) {
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define LTMI_CAST_CASE(DT, O)                                                                       \
    case DT: hipLaunchKernelGGL((k_store_cast<S, O>), grid, block, 0, stream, src, n, (O *)out,     \
                                stride, accumulate); break;
    switch (out_dtype) {
        LTMI_CAST_CASE(LTMI_U8, uint8_t) LTMI_CAST_CASE(LTMI_I8, int8_t)
        LTMI_CAST_CASE(LTMI_U16, uint16_t) LTMI_CAST_CASE(LTMI_I16, int16_t)
        LTMI_CAST_CASE(LTMI_U32, uint32_t) LTMI_CAST_CASE(LTMI_I32, int32_t)
        LTMI_CAST_CASE(LTMI_U64, uint64_t) LTMI_CAST_CASE(LTMI_I64, int64_t)
        LTMI_CAST_CASE(LTMI_F32, float) LTMI_CAST_CASE(LTMI_F64, double)
        default: LTMI_FAIL(LTMI_E_DTYPE, "store_cast: output dtype %s", dtype_name(out_dtype));
    }
#undef LTMI_CAST_CASE
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

template <typename T, typename A>
static int run_sum_frames_cast(const void *tile, int64_t n_frames, int64_t n_px, int64_t ld,
                               void *out, int out_dtype, int stride, int accumulate, void *ws,
                               hipStream_t stream) {
    if (!ws) LTMI_FAIL(LTMI_E_INVALID, "ltmi_sum_frames: workspace required");
    A *tmp = (A *)((char *)ws + split_bytes(n_frames, n_px, sizeof(A)));
    int rc = run_sum_frames<T, A>(tile, n_frames, n_px, ld, tmp, 0, ws, stream);
    if (rc != LTMI_OK) return rc;
    return store_cast<A>(tmp, n_px, out, out_dtype, stride, accumulate, stream);
}

extern "C" int ltmi_sum_frames(int device, const void *tile, int tile_dtype, int64_t n_frames,
                               int64_t n_px, int64_t ld_tile, void *out, int out_dtype,
                               int accumulate, void *workspace, void *stream_) {
    if (n_frames < 0 || n_px < 0 || ld_tile < n_px)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_sum_frames: bad shape");
    if (n_frames == 0 || n_px == 0) return LTMI_OK;
    if (!tile || !out) LTMI_FAIL(LTMI_E_INVALID, "ltmi_sum_frames: null pointer");
    LTMI_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    (void)tile_vec;
    if (is_cplx_dtype(tile_dtype)) {
        // complex frames are 2 * n_px real columns (udf/sum.py:38-40: the result keeps the complex dtype)
        const int rt = tile_dtype == LTMI_C64 ? LTMI_F32 : LTMI_F64;
        if (out_dtype == LTMI_C64 && rt == LTMI_F32)
            return run_sum_frames<float, float>(tile, n_frames, 2 * n_px, 2 * ld_tile, out, accumulate,
                                                workspace, stream);
        if (out_dtype == LTMI_C128 && rt == LTMI_F32)
            return run_sum_frames<float, double>(tile, n_frames, 2 * n_px, 2 * ld_tile, out, accumulate,
                                                 workspace, stream);
        if (out_dtype == LTMI_C128 && rt == LTMI_F64)
            return run_sum_frames<double, double>(tile, n_frames, 2 * n_px, 2 * ld_tile, out,
                                                  accumulate, workspace, stream);
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_sum_frames: unsupported dtypes tile=%s out=%s",
                  dtype_name(tile_dtype), dtype_name(out_dtype));
    }
    if (out_dtype == LTMI_F32) {
        LTMI_DISPATCH_TILE(run_sum_frames, float, tile, n_frames, n_px, ld_tile, out, accumulate, workspace, stream)
    } else if (out_dtype == LTMI_F64) {
        LTMI_DISPATCH_TILE(run_sum_frames, double, tile, n_frames, n_px, ld_tile, out, accumulate, workspace, stream)
    } else if (out_dtype == LTMI_C64) {
        LTMI_DISPATCH_TILE(run_sum_frames_cast, float, tile, n_frames, n_px, ld_tile, out, LTMI_F32, 2, accumulate, workspace, stream)
    } else if (out_dtype == LTMI_C128) {
        LTMI_DISPATCH_TILE(run_sum_frames_cast, double, tile, n_frames, n_px, ld_tile, out, LTMI_F64, 2, accumulate, workspace, stream)
    } else if (is_int_dtype(out_dtype) && out_dtype != LTMI_BOOL && is_int_dtype(tile_dtype)) {
        switch (tile_dtype) {
            case LTMI_BOOL:
            case LTMI_U8: return run_sum_frames_cast<uint8_t, int64_t>(tile, n_frames, n_px, ld_tile, out, out_dtype, 1, accumulate, workspace, stream);
            case LTMI_I8: return run_sum_frames_cast<int8_t, int64_t>(tile, n_frames, n_px, ld_tile, out, out_dtype, 1, accumulate, workspace, stream);
            case LTMI_U16: return run_sum_frames_cast<uint16_t, int64_t>(tile, n_frames, n_px, ld_tile, out, out_dtype, 1, accumulate, workspace, stream);
            case LTMI_I16: return run_sum_frames_cast<int16_t, int64_t>(tile, n_frames, n_px, ld_tile, out, out_dtype, 1, accumulate, workspace, stream);
            case LTMI_U32: return run_sum_frames_cast<uint32_t, int64_t>(tile, n_frames, n_px, ld_tile, out, out_dtype, 1, accumulate, workspace, stream);
            case LTMI_I32: return run_sum_frames_cast<int32_t, int64_t>(tile, n_frames, n_px, ld_tile, out, out_dtype, 1, accumulate, workspace, stream);
            case LTMI_U64: return run_sum_frames_cast<uint64_t, int64_t>(tile, n_frames, n_px, ld_tile, out, out_dtype, 1, accumulate, workspace, stream);
            case LTMI_I64: return run_sum_frames_cast<int64_t, int64_t>(tile, n_frames, n_px, ld_tile, out, out_dtype, 1, accumulate, workspace, stream);
        }
    }
    LTMI_FAIL(LTMI_E_DTYPE, "ltmi_sum_frames: unsupported dtypes tile=%s out=%s",
              dtype_name(tile_dtype), dtype_name(out_dtype));
}

// ---- detector corrections (reference io/corrections/detector.py:17-101) -----------------------------
// out[f, p] = ((double)tile[f, p] - dark[p]) * gain[p], rounded once to the output type: a float32
// buffer corrected with float64 dark / gain arrays is computed in float64 by the reference's loop too.
// One thread owns 4 consecutive pixels (dark / gain live in registers) and walks a slab of frames.
template <typename TIn, typename TOut, bool VEC>
__global__ void __launch_bounds__(256)
k_correct(const TIn *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
          const double *__restrict__ dark, const double *__restrict__ gain, TOut *__restrict__ out,
          int64_t ld_out, int frames_per_block) {
    constexpr int PX = VEC ? 8 : 4;                     // pixels per thread
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PX;
    if (p0 >= n_px) return;
    const int np = (int)min<int64_t>(PX, n_px - p0);
    double d[PX], g[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        d[j] = (dark && j < np) ? dark[p0 + j] : 0.0;
        g[j] = (gain && j < np) ? gain[p0 + j] : 1.0;
    }
    const int64_t f0 = (int64_t)blockIdx.y * frames_per_block;
    const int64_t f1 = min<int64_t>(n_frames, f0 + frames_per_block);
    if (VEC) {
        // whole 8-pixel units, 16-B aligned rows: one vector load, two/four vector stores per frame
        typedef TIn __attribute__((ext_vector_type(8))) vin_t;
        typedef TOut __attribute__((ext_vector_type(4))) vout_t;
#pragma unroll 4
        for (int64_t f = f0; f < f1; ++f) {
            const vin_t x = __builtin_nontemporal_load((const vin_t *)(tile + f * ld + p0));
            vout_t lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = (TOut)(((double)x[j] - d[j]) * g[j]);
                hi[j] = (TOut)(((double)x[j + 4] - d[j + 4]) * g[j + 4]);
            }
            TOut *dst = out + f * ld_out + p0;
            *(vout_t *)dst = lo;
            *(vout_t *)(dst + 4) = hi;
        }
    } else {
        for (int64_t f = f0; f < f1; ++f) {
            const TIn *src = tile + f * ld + p0;
            TOut *dst = out + f * ld_out + p0;
#pragma unroll
            for (int j = 0; j < PX; ++j)
                if (j < np) dst[j] = (TOut)(((double)src[j] - d[j]) * g[j]);
        }
    }
}

// buf[f, excl[e]] = mean of buf[f, env[e][0 .. cnt[e])] in float64 (environments hold good pixels
// only, so the patches are independent of each other)
template <typename T>
__global__ void __launch_bounds__(256)
k_repair_pixels(T *__restrict__ buf, int64_t ld, int64_t n_frames, const int32_t *__restrict__ excl,
                const int32_t *__restrict__ env, const int32_t *__restrict__ cnt, int n_excl,
                int max_env) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_frames * n_excl) return;
    const int64_t f = i / n_excl;
    const int e = (int)(i % n_excl);
    const int c = cnt[e];
    if (c <= 0) return;
    T *row = buf + f * ld;
    double acc = 0.0;
    for (int j = 0; j < c; ++j) acc += (double)row[env[(int64_t)e * max_env + j]];
    row[excl[e]] = (T)(acc / (double)c);
}

template <typename TIn, typename TOut>
static int run_correct(const void *tile, int64_t n_frames, int64_t n_px, int64_t ld,
                       const double *dark, const double *gain, void *out, int64_t ld_out,
                       hipStream_t stream) {
    const bool vec = (n_px % 8 == 0) && ((uintptr_t)tile % (8 * sizeof(TIn)) == 0) &&
                     (ld % 8 == 0) && ((uintptr_t)out % (4 * sizeof(TOut)) == 0) && (ld_out % 4 == 0);
    const int px = vec ? 8 : 4;
    const int64_t gx = (n_px + 256 * px - 1) / (256 * px);
    // enough blocks to fill the chip, frame slabs of at least 16
    int64_t gy = std::max<int64_t>(1, std::min<int64_t>((n_frames + 15) / 16, (4096 + gx - 1) / gx));
    gy = std::min<int64_t>(gy, 65535);
    const int fpb = (int)((n_frames + gy - 1) / gy);
    gy = (n_frames + fpb - 1) / fpb;
    if (vec)
        hipLaunchKernelGGL((k_correct<TIn, TOut, true>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0,
                           stream, (const TIn *)tile, ld, n_frames, n_px, dark, gain, (TOut *)out,
                           ld_out, fpb);
    else
        hipLaunchKernelGGL((k_correct<TIn, TOut, false>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0,
                           stream, (const TIn *)tile, ld, n_frames, n_px, dark, gain, (TOut *)out,
                           ld_out, fpb);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

extern "C" int ltmi_correct(int device, const void *tile, int tile_dtype, int64_t n_frames,
                            int64_t n_px, int64_t ld_tile, const double *dark, const double *gain,
                            void *out, int out_dtype, int64_t ld_out, void *stream_) {
    if (n_frames < 0 || n_px < 0 || ld_tile < n_px || ld_out < n_px)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_correct: bad shape");
    if (n_frames == 0 || n_px == 0) return LTMI_OK;
    if (!tile || !out) LTMI_FAIL(LTMI_E_INVALID, "ltmi_correct: null pointer");
    LTMI_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    if (out_dtype == LTMI_F32) {
        LTMI_DISPATCH_TILE(run_correct, float, tile, n_frames, n_px, ld_tile, dark, gain, out, ld_out, stream)
    } else if (out_dtype == LTMI_F64) {
        LTMI_DISPATCH_TILE(run_correct, double, tile, n_frames, n_px, ld_tile, dark, gain, out, ld_out, stream)
    }
    LTMI_FAIL(LTMI_E_DTYPE, "ltmi_correct: unsupported dtypes tile=%s out=%s", dtype_name(tile_dtype),
              dtype_name(out_dtype));
}

// ---- byte-order decode (reference io/dataset/base/decode.py:8-66, 89-100: byteswap_N_straight /
// decode_swap_only_N; the dtype conversion of decode_swap_N happens in the consuming kernels, which
// read every native dtype).  16 bytes per thread: 4 TB/s of HBM traffic per direction at most.
typedef unsigned int rz_u32x4 __attribute__((ext_vector_type(4)));

template <int ITEM> __device__ __forceinline__ rz_u32x4 swap16(rz_u32x4 v) {
    rz_u32x4 o;
    if constexpr (ITEM == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_perm(0u, v[i], 0x02030001u);   // bytes 1 0 3 2
    } else if constexpr (ITEM == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __builtin_bswap32(v[i]);
    } else {
        o[0] = __builtin_bswap32(v[1]); o[1] = __builtin_bswap32(v[0]);
        o[2] = __builtin_bswap32(v[3]); o[3] = __builtin_bswap32(v[2]);
    }
    return o;
}

template <int ITEM>
__global__ void __launch_bounds__(256)
k_byteswap(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, int64_t n_bytes,
           int vec_ok) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i >= n_bytes) return;
    if (vec_ok && i + 16 <= n_bytes) {
        const rz_u32x4 v = __builtin_nontemporal_load((const rz_u32x4 *)(src + i));
        *(rz_u32x4 *)(dst + i) = swap16<ITEM>(v);
    } else {
        for (int64_t e = i; e + ITEM <= n_bytes && e < i + 16; e += ITEM) {
            unsigned char t[ITEM];
#pragma unroll
            for (int b = 0; b < ITEM; ++b) t[b] = src[e + ITEM - 1 - b];
#pragma unroll
            for (int b = 0; b < ITEM; ++b) dst[e + b] = t[b];
        }
    }
}

extern "C" int ltmi_byteswap(int device, const void *src, void *dst, int itemsize, int64_t n_items,
                             void *stream_) {
    if (n_items < 0) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_byteswap: negative item count");
    if (itemsize == 1 || n_items == 0) {
        if (n_items && src != dst && src && dst)
            LTMI_HIP(hipMemcpyAsync(dst, src, (size_t)n_items, hipMemcpyDeviceToDevice, (hipStream_t)stream_));
        return LTMI_OK;
    }
    if (itemsize != 2 && itemsize != 4 && itemsize != 8)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_byteswap: item size %d (1, 2, 4 or 8)", itemsize);
    if (!src || !dst) LTMI_FAIL(LTMI_E_INVALID, "ltmi_byteswap: null pointer");
    LTMI_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t n_bytes = n_items * itemsize;
    const int vec_ok = ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    dim3 grid((unsigned)((n_bytes + 4095) / 4096));
    const unsigned char *s = (const unsigned char *)src;
    unsigned char *d = (unsigned char *)dst;
    if (itemsize == 2) hipLaunchKernelGGL((k_byteswap<2>), grid, dim3(256), 0, stream, s, d, n_bytes, vec_ok);
    else if (itemsize == 4) hipLaunchKernelGGL((k_byteswap<4>), grid, dim3(256), 0, stream, s, d, n_bytes, vec_ok);
    else hipLaunchKernelGGL((k_byteswap<8>), grid, dim3(256), 0, stream, s, d, n_bytes, vec_ok);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

extern "C" int ltmi_repair_pixels(int device, void *buf, int dtype, int64_t n_frames, int64_t ld,
                                  const int32_t *excl, const int32_t *env, const int32_t *cnt,
                                  int n_excl, int max_env, void *stream_) {
    if (n_frames < 0 || n_excl < 0 || max_env < 0) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_repair_pixels: bad shape");
    if (n_frames == 0 || n_excl == 0) return LTMI_OK;
    if (!buf || !excl || !env || !cnt) LTMI_FAIL(LTMI_E_INVALID, "ltmi_repair_pixels: null pointer");
    LTMI_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t n = n_frames * n_excl;
    dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == LTMI_F32)
        hipLaunchKernelGGL((k_repair_pixels<float>), grid, dim3(256), 0, stream, (float *)buf, ld,
                           n_frames, excl, env, cnt, n_excl, max_env);
    else if (dtype == LTMI_F64)
        hipLaunchKernelGGL((k_repair_pixels<double>), grid, dim3(256), 0, stream, (double *)buf, ld,
                           n_frames, excl, env, cnt, n_excl, max_env);
    else
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_repair_pixels: unsupported dtype %s", dtype_name(dtype));
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

// ---- centre-of-mass post-processing on a 2D scan (reference udf/com.py:100-142) ------------------
// raw rows (sum, sum*y, sum*x) float32 -> shift vectors: float32 division and reference subtraction as
// NumPy does on float32 arrays, then the 2x2 float64 transform (rotation / flip), magnitude,
// divergence and curl with np.gradient's stencils (central inside, one-sided at the edges).
__global__ void __launch_bounds__(256)
k_com_shifts(const float *__restrict__ raw, int64_t ld_raw, int64_t n, float ref_y, float ref_x,
             double tyy, double tyx, double txy, double txx, double *__restrict__ out_y,
             double *__restrict__ out_x, double *__restrict__ out_mag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float s = raw[i * ld_raw], sy = raw[i * ld_raw + 1], sx = raw[i * ld_raw + 2];
    float yc = ref_y, xc = ref_x;
    if (s != 0.f) {
        yc = __fdiv_rn(sy, s);
        xc = __fdiv_rn(sx, s);
    }
    yc = __fsub_rn(yc, ref_y);
    xc = __fsub_rn(xc, ref_x);
    const double y = (double)yc, x = (double)xc;
    const double yt = __dadd_rn(__dmul_rn(tyy, y), __dmul_rn(tyx, x));
    const double xt = __dadd_rn(__dmul_rn(txy, y), __dmul_rn(txx, x));
    out_y[i] = yt;
    out_x[i] = xt;
    if (out_mag) out_mag[i] = sqrt(__dadd_rn(__dmul_rn(yt, yt), __dmul_rn(xt, xt)));
}

__device__ __forceinline__ double grad_at(const double *f, int64_t idx, int64_t stride, int pos, int n) {
    if (n < 2) return 0.0;
    if (pos == 0) return f[idx + stride] - f[idx];
    if (pos == n - 1) return f[idx] - f[idx - stride];
    return (f[idx + stride] - f[idx - stride]) / 2.0;
}

__global__ void __launch_bounds__(256)
k_com_div_curl(const double *__restrict__ fy, const double *__restrict__ fx, int ny, int nx,
               double *__restrict__ out_div, double *__restrict__ out_curl) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)ny * nx) return;
    const int yy = (int)(i / nx), xx = (int)(i % nx);
    // axis 0 = y (stride nx), axis 1 = x (stride 1)
    if (out_div) out_div[i] = grad_at(fy, i, nx, yy, ny) + grad_at(fx, i, 1, xx, nx);
    if (out_curl) out_curl[i] = grad_at(fy, i, 1, xx, nx) - grad_at(fx, i, nx, yy, ny);
}

extern "C" int ltmi_com_fields(int device, const float *raw, int64_t ld_raw, int ny, int nx,
                               double ref_y, double ref_x, const double *transform,
                               double *out_y, double *out_x, double *out_mag, double *out_div,
                               double *out_curl, void *stream_) {
    if (ny <= 0 || nx <= 0 || ld_raw < 3)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_com_fields: bad shape (ny=%d nx=%d ld=%lld)", ny, nx,
                  (long long)ld_raw);
    if (!raw || !transform || !out_y || !out_x)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_com_fields: null pointer");
    LTMI_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t n = (int64_t)ny * nx;
    dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(k_com_shifts, grid, dim3(256), 0, stream, raw, ld_raw, n, (float)ref_y,
                       (float)ref_x, transform[0], transform[1], transform[2], transform[3], out_y,
                       out_x, out_mag);
    LTMI_HIP(hipGetLastError());
    if (out_div || out_curl) {
        hipLaunchKernelGGL(k_com_div_curl, grid, dim3(256), 0, stream, (const double *)out_y,
                           (const double *)out_x, ny, nx, out_div, out_curl);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}

template <typename A>
static void launch_add2d(void *dest, int64_t ld_dest, const void *src, int64_t ld_src, int64_t rows,
                         int64_t cols, int negate, hipStream_t stream) {
    dim3 grid((unsigned)((rows * cols + 255) / 256));
    hipLaunchKernelGGL((k_add2d<A>), grid, dim3(256), 0, stream, (A *)dest, ld_dest, (const A *)src,
                       ld_src, rows, cols, negate);
}

extern "C" int ltmi_add2d(int device, void *dest, int64_t ld_dest, const void *src, int64_t ld_src,
                          int dtype, int64_t rows, int64_t cols, int negate, void *stream_) {
    if (rows < 0 || cols < 0 || ld_dest < 0 || ld_src < 0)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_add2d: negative size");
    if (rows == 0 || cols == 0) return LTMI_OK;
    if (!dest || !src) LTMI_FAIL(LTMI_E_INVALID, "ltmi_add2d: null pointer");
    if (rows * cols > ((int64_t)1 << 39)) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_add2d: too many elements");
    LTMI_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    switch (dtype) {                    // integers wrap around like NumPy's `+=`
        case LTMI_BOOL: case LTMI_U8: case LTMI_I8:
            launch_add2d<uint8_t>(dest, ld_dest, src, ld_src, rows, cols, negate, stream); break;
        case LTMI_U16: case LTMI_I16:
            launch_add2d<uint16_t>(dest, ld_dest, src, ld_src, rows, cols, negate, stream); break;
        case LTMI_U32: case LTMI_I32:
            launch_add2d<uint32_t>(dest, ld_dest, src, ld_src, rows, cols, negate, stream); break;
        case LTMI_U64: case LTMI_I64:
            launch_add2d<uint64_t>(dest, ld_dest, src, ld_src, rows, cols, negate, stream); break;
        case LTMI_F32:
            launch_add2d<float>(dest, ld_dest, src, ld_src, rows, cols, negate, stream); break;
        case LTMI_F64:
            launch_add2d<double>(dest, ld_dest, src, ld_src, rows, cols, negate, stream); break;
        case LTMI_C64:                  // complex = pairs of reals
            launch_add2d<float>(dest, 2 * ld_dest, src, 2 * ld_src, rows, 2 * cols, negate, stream);
            break;
        case LTMI_C128:
            launch_add2d<double>(dest, 2 * ld_dest, src, 2 * ld_src, rows, 2 * cols, negate, stream);
            break;
        default:
            LTMI_FAIL(LTMI_E_DTYPE, "ltmi_add2d: unsupported dtype %d", dtype);
    }
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

extern "C" int ltmi_axpy(int device, void *dest, const void *src, int dtype, int64_t n,
                         void *stream_) {
    if (n < 0) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_axpy: negative size");
    return ltmi_add2d(device, dest, n, src, n, dtype, 1, n, 0, stream_);
}

extern "C" int ltmi_gather_rows(int device, const void *src, int64_t ld_src_bytes,
                                const int64_t *idx, int64_t n_rows, int64_t row_bytes, void *dest,
                                void *stream_) {
    if (n_rows < 0 || row_bytes < 0 || ld_src_bytes < row_bytes)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_gather_rows: bad sizes");
    if (n_rows == 0 || row_bytes == 0) return LTMI_OK;
    if (!src || !idx || !dest) LTMI_FAIL(LTMI_E_INVALID, "ltmi_gather_rows: null pointer");
    if (n_rows > 65535 * (int64_t)65535) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_gather_rows: too many rows");
    LTMI_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    const uintptr_t al = (uintptr_t)src | (uintptr_t)dest | (uintptr_t)ld_src_bytes |
                         (uintptr_t)row_bytes;
    // the grid's y extent is limited to 65535: walk the rows in slabs
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
        const int64_t nr = std::min<int64_t>(65535, n_rows - r0);
        char *d = (char *)dest + r0 * row_bytes;
        if ((al & 15) == 0) {
            const int64_t rv = row_bytes / 16;
            dim3 grid((unsigned)std::min<int64_t>((rv + 255) / 256, 64), (unsigned)nr);
            hipLaunchKernelGGL((k_gather_rows<gu32x4>), grid, dim3(256), 0, stream,
                               (const gu32x4 *)src, ld_src_bytes / 16, idx + r0, (gu32x4 *)d, rv);
        } else if ((al & 3) == 0) {
            const int64_t rv = row_bytes / 4;
            dim3 grid((unsigned)std::min<int64_t>((rv + 255) / 256, 64), (unsigned)nr);
            hipLaunchKernelGGL((k_gather_rows<uint32_t>), grid, dim3(256), 0, stream,
                               (const uint32_t *)src, ld_src_bytes / 4, idx + r0, (uint32_t *)d, rv);
        } else {
            dim3 grid((unsigned)std::min<int64_t>((row_bytes + 255) / 256, 64), (unsigned)nr);
            hipLaunchKernelGGL((k_gather_rows<uint8_t>), grid, dim3(256), 0, stream,
                               (const uint8_t *)src, ld_src_bytes, idx + r0, (uint8_t *)d, row_bytes);
        }
    }
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}
