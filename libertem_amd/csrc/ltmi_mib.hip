// Merlin / Medipix .mib frames decoded on the device (gfx950): per-frame headers stripped, big-endian
// integers swapped, raw "R64" words (64 x 1 bit, 8 x 6 bit, 4 x 12 bit per 64-bit word, first pixel in the
// least significant position, words stored most significant byte first) unpacked, 24-bit frames
// composed from their two 12-bit images, 2x2 quad rows [chip 4 | chip 3 | chip 2 | chip 1] laid out as
// one detector frame.  Replaces the numba decoders of src/libertem/io/dataset/mib.py:401-665
// (decode_r{1,6,12,24}_swap, decode_r{1,6,12}_swap_2x2) and the read-range bookkeeping that feeds them
// (mib.py:224-398), which run on the host for every tile.
//
// One thread per 64-bit word of payload: pure byte shuffling, bound by HBM (bytes in + bytes out).
#include "ltmi_common.h"

namespace {

typedef uint64_t u64_u __attribute__((aligned(1)));     // any byte address (gfx950: full-speed)
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(1)));

enum { M_U8 = 0, M_U16, M_U32, M_R1, M_R6, M_R12, M_R24, M_R24F };

__device__ __forceinline__ uint64_t swap_lanes16(uint64_t v) {          // bytes of every 16-bit lane
    return ((v & 0x00FF00FF00FF00FFull) << 8) | ((v >> 8) & 0x00FF00FF00FF00FFull);
}

// words per frame `wpf` (of 8 payload bytes; R24: of the 12-bit images' 8 bytes each), pixels per word
template <int MODE, bool QUAD>
__global__ void __launch_bounds__(256)
k_mib_decode(const unsigned char *__restrict__ src, int64_t frame_stride, unsigned char *__restrict__ dst,
             int64_t n_frames, int wpf, int blocks_per_frame, int payload_bytes, int height, int width) {
    const unsigned frame = blockIdx.x / (unsigned)blocks_per_frame;
    const int k = (int)(blockIdx.x - frame * (unsigned)blocks_per_frame) * 256 + (int)threadIdx.x;
    if (frame >= n_frames || k >= wpf) return;
    const unsigned char *in = src + (int64_t)frame * frame_stride;
    constexpr int PPW = MODE == M_R1 ? 64 : MODE == M_R6 || MODE == M_U8 ? 8
                        : MODE == M_U32 ? 2 : 4;            // pixels per word
    constexpr int OUT = MODE == M_U8 || MODE == M_R1 || MODE == M_R6 ? 1
                        : MODE == M_U16 || MODE == M_R12 ? 2 : 4;   // bytes per decoded pixel
    const int64_t n_px = (int64_t)height * width;
    unsigned char *frame_out = dst + (int64_t)frame * n_px * OUT;

    // pixels of this word: raster position `px0` (first pixel), `flip`: stored right to left
    int64_t px0 = (int64_t)k * PPW;
    bool flip = false;
    if (QUAD) {
        const int xh = width / 2, wps = xh / PPW;           // words per chip row
        const int j = k % wps, seg = (k / wps) & 3, r = k / (4 * wps);
        flip = seg < 2;
        const int y = flip ? height - 1 - r : r;
        const int x_half = (seg == 3 || seg == 1) ? 0 : xh;
        const int x = flip ? xh - (j + 1) * PPW : j * PPW;
        px0 = (int64_t)y * width + x_half + x;
    }

    if (MODE == M_U8 || MODE == M_U16 || MODE == M_U32) {
        const int valid = payload_bytes - k * 8;            // the last word of odd sizes is partial
        if (valid >= 8) {
            uint64_t v = *(const u64_u *)(in + (int64_t)k * 8);
            if (MODE == M_U16) v = swap_lanes16(v);
            if (MODE == M_U32)
                v = ((uint64_t)__builtin_bswap32((uint32_t)(v >> 32)) << 32) | __builtin_bswap32((uint32_t)v);
            *(u64_u *)(frame_out + (int64_t)k * 8) = v;
        } else {
            for (int b = 0; b + OUT <= valid; b += OUT)
                for (int e = 0; e < OUT; ++e)
                    frame_out[(int64_t)k * 8 + b + e] = in[(int64_t)k * 8 + b + OUT - 1 - e];
        }
        return;
    }
    if (MODE == M_R6) {
        const uint64_t v = *(const u64_u *)(in + (int64_t)k * 8);
        *(u64_u *)(frame_out + px0) = flip ? v : __builtin_bswap64(v);
        return;
    }
    if (MODE == M_R12) {
        const uint64_t v = *(const u64_u *)(in + (int64_t)k * 8);
        *(u64_u *)(frame_out + px0 * 2) = flip ? swap_lanes16(v) : __builtin_bswap64(v);
        return;
    }
    if (MODE == M_R1) {
        uint64_t w = __builtin_bswap64(*(const u64_u *)(in + (int64_t)k * 8));
        if (flip) w = __builtin_bitreverse64(w);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                       // 16 pixels = 16 bytes per store
            u32x4_u o;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t nib = (uint32_t)(w >> (16 * q + 4 * d)) & 0xFu;
                o[d] = (nib * 0x00204081u) & 0x01010101u;   // bit i -> byte i
            }
            *(u32x4_u *)(frame_out + px0 + 16 * q) = o;
        }
        return;
    }
    if (MODE == M_R24 || MODE == M_R24F) {
        const uint64_t hi = __builtin_bswap64(*(const u64_u *)(in + (int64_t)k * 8));
        const uint64_t lo = __builtin_bswap64(*(const u64_u *)(in + (int64_t)(k + wpf) * 8));
        uint32_t v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            v[t] = ((uint32_t)((hi >> (16 * t)) & 0xFFFFu) << 12) + (uint32_t)((lo >> (16 * t)) & 0xFFFFu);
        if (MODE == M_R24) {
            u32x4_u o = {v[0], v[1], v[2], v[3]};
            *(u32x4_u *)(frame_out + px0 * 4) = o;
        } else {
            f32x4_u o = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            *(f32x4_u *)(frame_out + px0 * 4) = o;
        }
    }
}

// The same with 16 bytes of OUTPUT per thread (one dwordx4 store per lane, a wave writes 1 KiB in a row):
// the common shapes -- payload a multiple of 16 bytes, an even number of words per (chip) row.
typedef uint64_t u64x2_u __attribute__((ext_vector_type(2), aligned(1)));

template <int MODE, bool QUAD>
__global__ void __launch_bounds__(256)
k_mib_decode16(const unsigned char *__restrict__ src, int64_t frame_stride, unsigned char *__restrict__ dst,
               int64_t n_frames, int cpf, int blocks_per_frame, int wpf24, int height, int width) {
    const unsigned frame = blockIdx.x / (unsigned)blocks_per_frame;
    const int c = (int)(blockIdx.x - frame * (unsigned)blocks_per_frame) * 256 + (int)threadIdx.x;
    if (frame >= n_frames || c >= cpf) return;
    const unsigned char *in = src + (int64_t)frame * frame_stride;
    constexpr int OUT = MODE == M_U8 || MODE == M_R1 || MODE == M_R6 ? 1
                        : MODE == M_U16 || MODE == M_R12 ? 2 : 4;
    const int64_t n_px = (int64_t)height * width;
    unsigned char *frame_out = dst + (int64_t)frame * n_px * OUT;
    constexpr int PPC = 16 / OUT;                           // pixels per 16-byte chunk of output

    if (MODE == M_R1) {
        // chunk c: pixels 16 q ... 16 q + 15 of word k
        const int k = c >> 2, q = c & 3;
        int64_t px0 = (int64_t)k * 64 + 16 * q;
        uint64_t w = __builtin_bswap64(*(const u64_u *)(in + (int64_t)k * 8));
        if (QUAD) {
            const int xh = width / 2, wps = xh / 64;
            const int j = k % wps, seg = (k / wps) & 3, r = k / (4 * wps);
            const bool flip = seg < 2;
            const int y = flip ? height - 1 - r : r;
            const int x_half = (seg == 3 || seg == 1) ? 0 : xh;
            if (flip) w = __builtin_bitreverse64(w);
            px0 = (int64_t)y * width + x_half + (flip ? xh - (j + 1) * 64 : j * 64) + 16 * q;
        }
        const uint32_t bits16 = (uint32_t)(w >> (16 * q)) & 0xFFFFu;
        u32x4_u o;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            o[d] = (((bits16 >> (4 * d)) & 0xFu) * 0x00204081u) & 0x01010101u;
        *(u32x4_u *)(frame_out + px0) = o;
        return;
    }
    if (MODE == M_R24 || MODE == M_R24F) {
        const uint64_t hi = __builtin_bswap64(*(const u64_u *)(in + (int64_t)c * 8));
        const uint64_t lo = __builtin_bswap64(*(const u64_u *)(in + (int64_t)(c + wpf24) * 8));
        uint32_t v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            v[t] = ((uint32_t)((hi >> (16 * t)) & 0xFFFFu) << 12) + (uint32_t)((lo >> (16 * t)) & 0xFFFFu);
        if (MODE == M_R24) {
            u32x4_u o = {v[0], v[1], v[2], v[3]};
            *(u32x4_u *)(frame_out + (int64_t)c * 16) = o;
        } else {
            f32x4_u o = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            *(f32x4_u *)(frame_out + (int64_t)c * 16) = o;
        }
        return;
    }
    // 16 bytes in -> 16 bytes out (two words)
    const u64x2_u v = *(const u64x2_u *)(in + (int64_t)c * 16);
    u64x2_u o;
    int64_t px0 = (int64_t)c * PPC;
    if (MODE == M_U8) { o = v; }
    else if (MODE == M_U16) { o[0] = swap_lanes16(v[0]); o[1] = swap_lanes16(v[1]); }
    else if (MODE == M_U32) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            o[i] = ((uint64_t)__builtin_bswap32((uint32_t)(v[i] >> 32)) << 32) |
                   __builtin_bswap32((uint32_t)v[i]);
    } else {                                                // M_R6 / M_R12
        bool flip = false;
        if (QUAD) {
            const int xh = width / 2, cps = xh / PPC;       // chunks per chip row
            const int j = c % cps, seg = (c / cps) & 3, r = c / (4 * cps);
            flip = seg < 2;
            const int y = flip ? height - 1 - r : r;
            const int x_half = (seg == 3 || seg == 1) ? 0 : xh;
            px0 = (int64_t)y * width + x_half + (flip ? xh - (j + 1) * PPC : j * PPC);
        }
        if (!flip) { o[0] = __builtin_bswap64(v[0]); o[1] = __builtin_bswap64(v[1]); }
        else if (MODE == M_R6) { o[0] = v[1]; o[1] = v[0]; }
        else { o[0] = swap_lanes16(v[1]); o[1] = swap_lanes16(v[0]); }
    }
    *(u64x2_u *)(frame_out + px0 * OUT) = o;
}

}  // namespace

extern "C" int ltmi_mib_decode(int device, const void *src, int64_t frame_stride, int64_t header_bytes,
                               int kind, int bits, int quad, int64_t n_frames, int height, int width,
                               void *dst, int dst_dtype, void *stream_) {
    if (n_frames < 0 || height <= 0 || width <= 0 || frame_stride <= 0 || header_bytes < 0)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_mib_decode: bad geometry (frames=%lld %dx%d stride=%lld header=%lld)",
                  (long long)n_frames, height, width, (long long)frame_stride, (long long)header_bytes);
    if (n_frames == 0) return LTMI_OK;
    if (!src || !dst) LTMI_FAIL(LTMI_E_INVALID, "ltmi_mib_decode: null pointer");
    const int64_t n_px = (int64_t)height * width;
    int mode = -1, want = -1;
    int64_t payload = 0;
    if (kind == 'u') {
        if (bits == 8) { mode = M_U8; want = LTMI_U8; }
        else if (bits == 16) { mode = M_U16; want = LTMI_U16; }
        else if (bits == 32) { mode = M_U32; want = LTMI_U32; }
        payload = n_px * (bits / 8);
        quad = 0;                                           // integer files hold assembled frames
    } else if (kind == 'r') {
        if (bits == 1) { mode = M_R1; want = LTMI_U8; payload = n_px / 8; }
        else if (bits == 6) { mode = M_R6; want = LTMI_U8; payload = n_px; }
        else if (bits == 12) { mode = M_R12; want = LTMI_U16; payload = n_px * 2; }
        else if (bits == 24) {
            mode = dst_dtype == LTMI_F32 ? M_R24F : M_R24;
            want = dst_dtype == LTMI_F32 ? LTMI_F32 : LTMI_U32;
            payload = n_px * 4;
        }
    }
    if (mode < 0)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_mib_decode: kind '%c' with %d bits per pixel is not a .mib format",
                  kind, bits);
    if (dst_dtype != want)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_mib_decode: %c%d frames decode to %s, not %s", kind, bits,
                  ltmi::dtype_name(want), ltmi::dtype_name(dst_dtype));
    if (header_bytes + payload > frame_stride)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_mib_decode: header %lld + payload %lld exceed the frame stride %lld",
                  (long long)header_bytes, (long long)payload, (long long)frame_stride);
    if (kind == 'r') {
        const int ppw = bits == 1 ? 64 : bits == 6 ? 8 : 4;
        if (quad && bits == 24)
            LTMI_FAIL(LTMI_E_DTYPE, "ltmi_mib_decode: 24-bit raw data of a quad detector "
                                          "(the reference does not read it either, mib.py:1007-1011)");
        if (quad ? ((width / 2) % ppw != 0 || (width & 1) || (height & 1)) : (width % ppw != 0))
            LTMI_FAIL(LTMI_E_SHAPE, "ltmi_mib_decode: %d-bit raw rows hold whole 64-bit words: the %s "
                                    "width %d is not a multiple of %d pixels", bits,
                      quad ? "chip" : "frame", quad ? width / 2 : width, ppw);
    }
    LTMI_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    // 16 bytes of output per thread when the shape allows (env LTMI_MIB_WORDS=1: the per-word kernel)
    static const bool words_only = getenv("LTMI_MIB_WORDS") != nullptr;
    const int ppc = 16 / ltmi::dtype_size(want);
    const bool wide = !words_only && (payload % 16 == 0 || mode == M_R1) &&
                      (!quad || mode == M_R1 || ((width / 2) % ppc == 0));
    if (wide) {
        const int64_t chunks = mode == M_R1 ? payload / 2 : (mode == M_R24 || mode == M_R24F) ? payload / 16
                                                                                             : payload / 16;
        if (chunks > (1ll << 30)) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_mib_decode: frame too large");
        const int cpf = (int)chunks, bpf = (cpf + 255) / 256, wpf24 = (int)(payload / 16);
        const unsigned char *s = (const unsigned char *)src + header_bytes;
        const int64_t out_frame = n_px * ltmi::dtype_size(want);
        const int64_t max_frames = std::max<int64_t>(1, (int64_t)0x7FFFFFFF / bpf);
        for (int64_t f0 = 0; f0 < n_frames; f0 += max_frames) {
            const int64_t nf = std::min(max_frames, n_frames - f0);
            dim3 grid((unsigned)(nf * bpf));
            const unsigned char *sp = s + f0 * frame_stride;
            unsigned char *dp = (unsigned char *)dst + f0 * out_frame;
#define LTMI_MIB_LAUNCH16(MODE, QUAD)                                                               \
            hipLaunchKernelGGL((k_mib_decode16<MODE, QUAD>), grid, dim3(256), 0, stream, sp,        \
                               frame_stride, dp, nf, cpf, bpf, wpf24, height, width)
            switch (mode) {
                case M_U8: LTMI_MIB_LAUNCH16(M_U8, false); break;
                case M_U16: LTMI_MIB_LAUNCH16(M_U16, false); break;
                case M_U32: LTMI_MIB_LAUNCH16(M_U32, false); break;
                case M_R1: if (quad) LTMI_MIB_LAUNCH16(M_R1, true); else LTMI_MIB_LAUNCH16(M_R1, false); break;
                case M_R6: if (quad) LTMI_MIB_LAUNCH16(M_R6, true); else LTMI_MIB_LAUNCH16(M_R6, false); break;
                case M_R12: if (quad) LTMI_MIB_LAUNCH16(M_R12, true); else LTMI_MIB_LAUNCH16(M_R12, false); break;
                case M_R24: LTMI_MIB_LAUNCH16(M_R24, false); break;
                default: LTMI_MIB_LAUNCH16(M_R24F, false); break;
            }
#undef LTMI_MIB_LAUNCH16
            LTMI_HIP(hipGetLastError());
        }
        return LTMI_OK;
    }
    const int64_t words = (mode == M_R24 || mode == M_R24F) ? payload / 16 : (payload + 7) / 8;
    if (words > (1ll << 30)) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_mib_decode: frame too large");
    const int wpf = (int)words, bpf = (wpf + 255) / 256;
    const unsigned char *s = (const unsigned char *)src + header_bytes;
    const int64_t out_frame = n_px * ltmi::dtype_size(want);
    const int64_t max_frames = std::max<int64_t>(1, (int64_t)0x7FFFFFFF / bpf);
    for (int64_t f0 = 0; f0 < n_frames; f0 += max_frames) {
        const int64_t nf = std::min(max_frames, n_frames - f0);
        dim3 grid((unsigned)(nf * bpf));
        const unsigned char *sp = s + f0 * frame_stride;
        unsigned char *dp = (unsigned char *)dst + f0 * out_frame;
#define LTMI_MIB_LAUNCH(MODE, QUAD)                                                               \
        hipLaunchKernelGGL((k_mib_decode<MODE, QUAD>), grid, dim3(256), 0, stream, sp, frame_stride, \
                           dp, nf, wpf, bpf, (int)payload, height, width)
        switch (mode) {
            case M_U8: LTMI_MIB_LAUNCH(M_U8, false); break;
            case M_U16: LTMI_MIB_LAUNCH(M_U16, false); break;
            case M_U32: LTMI_MIB_LAUNCH(M_U32, false); break;
            case M_R1: if (quad) LTMI_MIB_LAUNCH(M_R1, true); else LTMI_MIB_LAUNCH(M_R1, false); break;
            case M_R6: if (quad) LTMI_MIB_LAUNCH(M_R6, true); else LTMI_MIB_LAUNCH(M_R6, false); break;
            case M_R12: if (quad) LTMI_MIB_LAUNCH(M_R12, true); else LTMI_MIB_LAUNCH(M_R12, false); break;
            case M_R24: LTMI_MIB_LAUNCH(M_R24, false); break;
            default: LTMI_MIB_LAUNCH(M_R24F, false); break;
        }
#undef LTMI_MIB_LAUNCH
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}
