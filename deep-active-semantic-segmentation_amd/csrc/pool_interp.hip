// Layout changes, channel concat/split, 3x3/s2 max-pool, global average pool / broadcast and
// bilinear (align_corners=True) resampling with a deterministic gather-form backward.  NHWC rows,
// 4 channels (16 B f32 / 8 B bf16) per lane, all HBM-bound.
// Reference sites: resnet.py:68 (maxpool), aspp.py:62,79-83 (avgpool, 1x1->HxW interpolate, cat),
// decoder.py:45-46 (interpolate, cat), deeplab.py:59 (final interpolate to input size).
#include "dass_common.h"

namespace {

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ x, T *__restrict__ y, int N, int C, int H, int W,
                                    int Cpad) {
    const long total = (long)N * H * W * Cpad;
    const long hw = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long pix = i / Cpad;
        const long n = pix / hw, p = pix - n * hw;
        const float v = c < C ? x[(n * C + c) * hw + p] : 0.f;
        Elem<T>::st(y + i, v);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T *__restrict__ x, long ldx, float *__restrict__ y, int N, int C, int H,
                                    int W) {
    const long hw = (long)H * W;
    const long total = (long)N * C * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i % hw;
        const long nc = i / hw;
        const int c = (int)(nc % C);
        const long n = nc / C;
        y[i] = Elem<T>::ld(x + (n * hw + p) * ldx + c);
    }
}

template <typename T, bool ADD>
__global__ void copy_channels_kernel(const T *__restrict__ src, long lds, T *__restrict__ dst, long ldd, long M,
                                     int C) {
    const int cv = C >> 2;
    const long total = M * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / cv;
        const int c = (int)(i - m * cv) << 2;
        f32x4 v = ld4<T>(src + m * lds + c);
        if (ADD) v += ld4<T>(dst + m * ldd + c);
        st4<T>(dst + m * ldd + c, v);
    }
}

// dst = sum of up to 8 row tensors (each with its own pixel stride): the gradients of a tensor with several consumers, added in
// ONE pass (autograd would run n - 1 two-operand adds over the whole tensor)
struct SumPack {
    const void *src[8];
    long ld[8];
    int n;
};
template <typename T> __global__ void sum_n_kernel(SumPack p, T *__restrict__ dst, long ldd, long M, int C) {
    const int cv = C >> 2;
    const long total = M * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / cv;
        const int c = (int)(i - m * cv) << 2;
        f32x4 v = ld4<T>(reinterpret_cast<const T *>(p.src[0]) + m * p.ld[0] + c);
#pragma unroll
        for (int j = 1; j < 8; ++j)
            if (j < p.n) v += ld4<T>(reinterpret_cast<const T *>(p.src[j]) + m * p.ld[j] + c);
        st4<T>(dst + m * ldd + c, v);
    }
}

template <typename T>
__global__ void maxpool_fwd_kernel(const T *__restrict__ x, T *__restrict__ y, uint8_t *__restrict__ idx, int N,
                                   int H, int W, int C, int OH, int OW) {
    const int cv = C >> 2;
    const long total = (long)N * OH * OW * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) << 2;
        long t = i / cv;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH);
        const long n = t / OH;
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
        for (int r = 0; r < 3; ++r) {
            const int iy = oh * 2 - 1 + r;
            if (iy < 0 || iy >= H) continue;
            for (int s = 0; s < 3; ++s) {
                const int ix = ow * 2 - 1 + s;
                if (ix < 0 || ix >= W) continue;
                const f32x4 v = ld4<T>(x + ((n * H + iy) * W + ix) * (long)C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (first || v[e] > best[e]) {
                        best[e] = v[e];
                        bi[e] = r * 3 + s;
                    }
                first = false;
            }
        }
        const long o = ((n * OH + oh) * OW + ow) * (long)C + c;
        st4<T>(y + o, best);
        if (idx) {
            uchar4 u = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
            *reinterpret_cast<uchar4 *>(idx + o) = u;
        }
    }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T *__restrict__ dy, const uint8_t *__restrict__ idx, T *__restrict__ dx,
                                   int N, int H, int W, int C, int OH, int OW) {
    const int cv = C >> 2;
    const long total = (long)N * H * W * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) << 2;
        long t = i / cv;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const long n = t / H;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // windows (oh, r) with oh*2 - 1 + r == iy
        for (int r = 0; r < 3; ++r) {
            const int ty = iy + 1 - r;
            if (ty < 0 || (ty & 1)) continue;
            const int oh = ty >> 1;
            if (oh >= OH) continue;
            for (int s = 0; s < 3; ++s) {
                const int tx = ix + 1 - s;
                if (tx < 0 || (tx & 1)) continue;
                const int ow = tx >> 1;
                if (ow >= OW) continue;
                const long o = ((n * OH + oh) * OW + ow) * (long)C + c;
                const uchar4 u = *reinterpret_cast<const uchar4 *>(idx + o);
                const f32x4 d = ld4<T>(dy + o);
                const int tap = r * 3 + s;
                if (u.x == tap) g[0] += d[0];
                if (u.y == tap) g[1] += d[1];
                if (u.z == tap) g[2] += d[2];
                if (u.w == tap) g[3] += d[3];
            }
        }
        st4<T>(dx + ((n * H + iy) * W + ix) * (long)C + c, g);
    }
}

// y[n][c] = mean_hw x[n,hw,c]; grid (C/64, N), 16 channel-quads x 16 row lanes
template <typename T, bool MEAN>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const T *__restrict__ x, long ldx, T *__restrict__ y, int N,
                                                          long HW, int C) {
    __shared__ float red[16][64 + 1];
    const int tid = threadIdx.x, cx = tid & 15, ry = tid >> 4;
    const int c = blockIdx.x * 64 + cx * 4;
    const long n = blockIdx.y;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (long r = ry; r < HW; r += 16) s += ld4<T>(x + (n * HW + r) * ldx + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ry][cx * 4 + e] = s[e];
    __syncthreads();
    if (tid < 64) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += red[i][tid];
        const int cc = blockIdx.x * 64 + tid;
        if (cc < C) Elem<T>::st(y + n * C + cc, MEAN ? a / (float)HW : a);
    }
}

template <typename T>
__global__ void broadcast_rows_kernel(const T *__restrict__ src, T *__restrict__ dst, long ldd, int N, long HW, int C,
                                      float mult) {
    const int cv = C >> 2;
    const long total = (long)N * HW * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) << 2;
        const long m = i / cv;
        const long n = m / HW;
        st4<T>(dst + m * ldd + c, ld4<T>(src + n * C + c) * mult);
    }
}

// torch area_pixel_compute_source_index(align_corners=True): src = dst * (in-1)/(out-1)
struct Lerp {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Lerp lerp_of(int o, int in, float sc) {
    Lerp L;
    const float src = sc * (float)o;
    L.i0 = (int)src;
    if (L.i0 > in - 1) L.i0 = in - 1;
    L.i1 = L.i0 + (L.i0 < in - 1 ? 1 : 0);
    L.l1 = src - (float)L.i0;
    L.l0 = 1.f - L.l1;
    return L;
}
__host__ __device__ __forceinline__ float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

template <typename T>
__global__ void bilinear_fwd_nhwc_kernel(const T *__restrict__ x, long ldx, T *__restrict__ y, long ldy, int N, int IH,
                                         int IW, int C, int OH, int OW, float sh, float sw) {
    const int cv = C >> 2;
    const long total = (long)N * OH * OW * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) << 2;
        long t = i / cv;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const long n = t / OH;
        const Lerp ly = lerp_of(oy, IH, sh), lx = lerp_of(ox, IW, sw);
        const T *b = x + n * IH * IW * ldx + c;
        const f32x4 v00 = ld4<T>(b + ((long)ly.i0 * IW + lx.i0) * ldx), v01 = ld4<T>(b + ((long)ly.i0 * IW + lx.i1) * ldx);
        const f32x4 v10 = ld4<T>(b + ((long)ly.i1 * IW + lx.i0) * ldx), v11 = ld4<T>(b + ((long)ly.i1 * IW + lx.i1) * ldx);
        const f32x4 v = (v00 * lx.l0 + v01 * lx.l1) * ly.l0 + (v10 * lx.l0 + v11 * lx.l1) * ly.l1;
        st4<T>(y + ((n * OH + oy) * OW + ox) * ldy + c, v);
    }
}

// NHWC (any C, scalar channel loop) -> NCHW f32; one thread per output pixel, coalesced plane writes
template <typename T>
__global__ void bilinear_fwd_nchw_kernel(const T *__restrict__ x, long ldx, float *__restrict__ y, int N, int IH,
                                         int IW, int C, int OH, int OW, float sh, float sw) {
    const long ohw = (long)OH * OW;
    const long total = (long)N * ohw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / ohw;
        const long p = i - n * ohw;
        const int oy = (int)(p / OW), ox = (int)(p - (long)oy * OW);
        const Lerp ly = lerp_of(oy, IH, sh), lx = lerp_of(ox, IW, sw);
        const T *b = x + n * IH * IW * ldx;
        const T *p00 = b + ((long)ly.i0 * IW + lx.i0) * ldx, *p01 = b + ((long)ly.i0 * IW + lx.i1) * ldx;
        const T *p10 = b + ((long)ly.i1 * IW + lx.i0) * ldx, *p11 = b + ((long)ly.i1 * IW + lx.i1) * ldx;
        float *o = y + n * C * ohw + p;
        for (int c = 0; c < C; ++c) {
            const float v = ly.l0 * (lx.l0 * Elem<T>::ld(p00 + c) + lx.l1 * Elem<T>::ld(p01 + c)) +
                            ly.l1 * (lx.l0 * Elem<T>::ld(p10 + c) + lx.l1 * Elem<T>::ld(p11 + c));
            o[(long)c * ohw] = v;
        }
    }
}

__device__ __forceinline__ void gather_range(int i, int in, int out, float sc, int &lo, int &hi) {
    if (sc > 0.f) {
        lo = (int)floorf((float)(i - 1) / sc) - 1;
        hi = (int)ceilf((float)(i + 1) / sc) + 1;
        if (lo < 0) lo = 0;
        if (hi > out - 1) hi = out - 1;
    } else {
        lo = 0;
        hi = out - 1;
    }
}
__device__ __forceinline__ float lerp_weight(const Lerp &L, int i) {
    return (L.i0 == i ? L.l0 : 0.f) + (L.i1 == i ? L.l1 : 0.f);
}

template <typename T>
__global__ void bilinear_bwd_nhwc_kernel(const T *__restrict__ dy, long lddy, T *__restrict__ dx, long lddx, int N,
                                         int IH, int IW, int C, int OH, int OW, float sh, float sw) {
    const int cv = C >> 2;
    const long total = (long)N * IH * IW * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) << 2;
        long t = i / cv;
        const int ix = (int)(t % IW);
        t /= IW;
        const int iy = (int)(t % IH);
        const long n = t / IH;
        int ylo, yhi, xlo, xhi;
        gather_range(iy, IH, OH, sh, ylo, yhi);
        gather_range(ix, IW, OW, sw, xlo, xhi);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = lerp_weight(lerp_of(oy, IH, sh), iy);
            if (wy == 0.f) continue;
            f32x4 row = {0.f, 0.f, 0.f, 0.f};
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float wx = lerp_weight(lerp_of(ox, IW, sw), ix);
                if (wx == 0.f) continue;
                row += ld4<T>(dy + ((n * OH + oy) * OW + ox) * lddy + c) * wx;
            }
            g += row * wy;
        }
        st4<T>(dx + ((n * IH + iy) * IW + ix) * lddx + c, g);
    }
}

// dy NCHW f32 -> dx NHWC (scalar channels); one thread per (input pixel, channel)
template <typename T>
__global__ void bilinear_bwd_nchw_kernel(const float *__restrict__ dy, T *__restrict__ dx, long lddx, int N, int IH,
                                         int IW, int C, int OH, int OW, float sh, float sw) {
    const long total = (long)N * C * IH * IW;
    const long ohw = (long)OH * OW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % IW);
        long t = i / IW;
        const int iy = (int)(t % IH);
        t /= IH;
        const int c = (int)(t % C);
        const long n = t / C;
        int ylo, yhi, xlo, xhi;
        gather_range(iy, IH, OH, sh, ylo, yhi);
        gather_range(ix, IW, OW, sw, xlo, xhi);
        const float *plane = dy + (n * C + c) * ohw;
        float g = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = lerp_weight(lerp_of(oy, IH, sh), iy);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float wx = lerp_weight(lerp_of(ox, IW, sw), ix);
                if (wx == 0.f) continue;
                row += plane[(long)oy * OW + ox] * wx;
            }
            g += row * wy;
        }
        Elem<T>::st(dx + ((n * IH + iy) * IW + ix) * lddx + c, g);
    }
}

// The same gather for the network's own case -- logits 129 -> 513 (193 -> 769): OH - 1 = 4 (IH - 1), align_corners -- where the source
// coordinate of output o is exactly o / 4: input pixel i gathers outputs 4 i - 3 .. 4 i + 3 with weights 1 - |d| / 4.  The general kernel
// derives every tap's weight from lerp_of / lerp_weight and issues 49 scalar loads behind two skip tests; here the weights are constants
// and a thread reads each of its <= 7 rows as two 16-byte loads (rows of 513 floats are only dword-aligned: packed struct).  Products are
// added in the general kernel's order (rows ascending, columns ascending, zero-weight taps skipped); where the compiler contracts a multiply-add
// into an fma differs, so the two agree to the last bits, not bit for bit (tests/test_stem_gpu.py: 1e-6, and 1e-5 against f64 autograd).
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

template <typename T>
__global__ void bilinear_bwd_nchw_r4_kernel(const float *__restrict__ dy, T *__restrict__ dx, long lddx, int N, int IH, int IW, int C,
                                            int OH, int OW) {
    const long total = (long)N * C * IH * IW;
    const long ohw = (long)OH * OW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % IW);
        long t = i / IW;
        const int iy = (int)(t % IH);
        t /= IH;
        const int c = (int)(t % C);
        const long n = t / C;
        const float *plane = dy + (n * C + c) * ohw;
        const bool left = ix > 0, right = ix < IW - 1;
        float g = 0.f;
#pragma unroll
        for (int d = -3; d <= 3; ++d) {
            const int oy = 4 * iy + d;
            if (oy < 0 || oy >= OH) continue;
            const float wy = 1.f - 0.25f * (float)(d < 0 ? -d : d);
            const float *rowp = plane + (long)oy * OW + 4 * ix;
            f4u a = {{0.f, 0.f, 0.f, 0.f}}, b = a;
            if (left) a = *reinterpret_cast<const f4u *>(rowp - 4);
            if (right) b = *reinterpret_cast<const f4u *>(rowp);
            else b.v[0] = rowp[0];
            float row = 0.f;
            if (left) {
                row += a.v[1] * 0.25f;
                row += a.v[2] * 0.5f;
                row += a.v[3] * 0.75f;
            }
            row += b.v[0] * 1.f;
            if (right) {
                row += b.v[1] * 0.75f;
                row += b.v[2] * 0.5f;
                row += b.v[3] * 0.25f;
            }
            g += row * wy;
        }
        Elem<T>::st(dx + ((n * IH + iy) * IW + ix) * lddx + c, g);
    }
}

}  // namespace

#define DASS_DISPATCH(DT, KERNEL_F32, KERNEL_BF16) \
    if ((DT) == DASS_F32) { KERNEL_F32; }          \
    else if ((DT) == DASS_BF16) { KERNEL_BF16; }   \
    else return DASS_ERR_UNSUPPORTED;              \
    DASS_LAUNCH_CHECK();                           \
    return DASS_OK;

extern "C" int dass_nchw_to_nhwc(const float *x, void *y, int N, int C, int H, int W, int Cpad, int dtype,
                                 void *stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * H * W * Cpad, 256);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, st, x, (float *)y, N, C, H, W, Cpad),
                  DASS_LAUNCH(nchw_to_nhwc_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, x, (bf16_t *)y, N, C, H, W, Cpad))
}

extern "C" int dass_nhwc_to_nchw(const void *x, int64_t ldx, float *y, int N, int C, int H, int W, int dtype,
                                 void *stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ldx < C) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * H * W * C, 256);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH(nhwc_to_nchw_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, ldx, y, N, C, H, W),
                  DASS_LAUNCH(nhwc_to_nchw_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)x, ldx, y, N, C, H, W))
}

extern "C" int dass_copy_channels(const void *src, int64_t lds, void *dst, int64_t ldd, int64_t M, int C, int dtype,
                                  void *stream) {
    if (!src || !dst || M <= 0 || C <= 0 || C % 4 || lds % 4 || ldd % 4) return DASS_ERR_ARG;
    const int grid = dass_grid_1d(M * (C / 4), 256);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH((copy_channels_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float *)src, lds, (float *)dst, ldd, M, C),
                  DASS_LAUNCH((copy_channels_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const bf16_t *)src, lds, (bf16_t *)dst, ldd, M, C))
}

extern "C" int dass_add_channels(const void *src, int64_t lds, void *dst, int64_t ldd, int64_t M, int C, int dtype,
                                 void *stream) {
    if (!src || !dst || M <= 0 || C <= 0 || C % 4 || lds % 4 || ldd % 4) return DASS_ERR_ARG;
    const int grid = dass_grid_1d(M * (C / 4), 256);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH((copy_channels_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float *)src, lds, (float *)dst, ldd, M, C),
                  DASS_LAUNCH((copy_channels_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const bf16_t *)src, lds, (bf16_t *)dst, ldd, M, C))
}

/* dst[m][c] = sum_j src_j[m][c], 2 <= n <= 8 sources given as n pointers and n pixel strides (host arrays) */
extern "C" int dass_sum_channels(const void *const *srcs, const int64_t *lds, int n, void *dst, int64_t ldd, int64_t M, int C, int dtype,
                                 void *stream) {
    if (!srcs || !lds || n < 1 || n > 8 || !dst || M <= 0 || C <= 0 || C % 4 || ldd % 4) return DASS_ERR_ARG;
    SumPack p;
    p.n = n;
    for (int j = 0; j < 8; ++j) {
        p.src[j] = j < n ? srcs[j] : nullptr;
        p.ld[j] = j < n ? lds[j] : 0;
        if (j < n && (!srcs[j] || lds[j] % 4)) return DASS_ERR_ARG;
    }
    const int grid = dass_grid_1d(M * (C / 4), 256);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype, DASS_LAUNCH(sum_n_kernel<float>, dim3(grid), dim3(256), 0, st, p, (float *)dst, ldd, M, C),
                  DASS_LAUNCH(sum_n_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, p, (bf16_t *)dst, ldd, M, C))
}

extern "C" int dass_maxpool3x3s2_fwd(const void *x, void *y, uint8_t *idx, int N, int H, int W, int C, int OH, int OW,
                                     int dtype, void *stream) {
    if (!x || !y || N <= 0 || C <= 0 || C % 4 || OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * OH * OW * (C / 4), 256);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH(maxpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, (float *)y, idx, N, H, W, C, OH, OW),
                  DASS_LAUNCH(maxpool_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)x, (bf16_t *)y, idx, N, H, W, C, OH, OW))
}

extern "C" int dass_maxpool3x3s2_bwd(const void *dy, const uint8_t *idx, void *dx, int N, int H, int W, int C, int OH,
                                     int OW, int dtype, void *stream) {
    if (!dy || !idx || !dx || N <= 0 || C <= 0 || C % 4 || OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * H * W * (C / 4), 256);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH(maxpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)dy, idx, (float *)dx, N, H, W, C, OH, OW),
                  DASS_LAUNCH(maxpool_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)dy, idx, (bf16_t *)dx, N, H, W, C, OH, OW))
}

extern "C" int dass_global_avgpool_fwd(const void *x, int64_t ldx, void *y, int N, int64_t HW, int C, int dtype,
                                       void *stream) {
    if (!x || !y || N <= 0 || HW <= 0 || C <= 0 || C % 4 || ldx % 4) return DASS_ERR_ARG;
    dim3 grid((C + 63) / 64, N);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH((reduce_rows_kernel<float, true>), grid, dim3(256), 0, st, (const float *)x, ldx, (float *)y, N, HW, C),
                  DASS_LAUNCH((reduce_rows_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t *)x, ldx, (bf16_t *)y, N, HW, C))
}

extern "C" int dass_reduce_rows(const void *src, int64_t lds, void *dst, int N, int64_t HW, int C, int dtype,
                                void *stream) {
    if (!src || !dst || N <= 0 || HW <= 0 || C <= 0 || C % 4 || lds % 4) return DASS_ERR_ARG;
    dim3 grid((C + 63) / 64, N);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH((reduce_rows_kernel<float, false>), grid, dim3(256), 0, st, (const float *)src, lds, (float *)dst, N, HW, C),
                  DASS_LAUNCH((reduce_rows_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t *)src, lds, (bf16_t *)dst, N, HW, C))
}

extern "C" int dass_broadcast_rows(const void *src, void *dst, int64_t ldd, int N, int64_t HW, int C, float mult,
                                   int dtype, void *stream) {
    if (!src || !dst || N <= 0 || HW <= 0 || C <= 0 || C % 4 || ldd % 4) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * HW * (C / 4), 256);
    hipStream_t st = (hipStream_t)stream;
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH(broadcast_rows_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)src, (float *)dst, ldd, N, HW, C, mult),
                  DASS_LAUNCH(broadcast_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)src, (bf16_t *)dst, ldd, N, HW, C, mult))
}

extern "C" int dass_bilinear_fwd(const void *x, int64_t ldx, void *y, int64_t ldy, int N, int IH, int IW, int C,
                                 int OH, int OW, int out_nchw, int dtype, void *stream) {
    if (!x || !y || N <= 0 || IH <= 0 || IW <= 0 || C <= 0 || OH <= 0 || OW <= 0 || ldx < C) return DASS_ERR_ARG;
    const float sh = ac_scale(IH, OH), sw = ac_scale(IW, OW);
    hipStream_t st = (hipStream_t)stream;
    if (out_nchw) {
        const int grid = dass_grid_1d((long)N * OH * OW, 256);
        DASS_DISPATCH(dtype,
                      DASS_LAUNCH(bilinear_fwd_nchw_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, ldx, (float *)y, N, IH, IW, C, OH, OW, sh, sw),
                      DASS_LAUNCH(bilinear_fwd_nchw_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)x, ldx, (float *)y, N, IH, IW, C, OH, OW, sh, sw))
    }
    if (C % 4 || ldx % 4 || ldy % 4) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * OH * OW * (C / 4), 256);
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH(bilinear_fwd_nhwc_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, ldx, (float *)y, ldy, N, IH, IW, C, OH, OW, sh, sw),
                  DASS_LAUNCH(bilinear_fwd_nhwc_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)x, ldx, (bf16_t *)y, ldy, N, IH, IW, C, OH, OW, sh, sw))
}

extern "C" int dass_bilinear_bwd(const void *dy, int64_t lddy, void *dx, int64_t lddx, int N, int IH, int IW, int C,
                                 int OH, int OW, int dy_nchw, int dtype, void *stream) {
    if (!dy || !dx || N <= 0 || IH <= 0 || IW <= 0 || C <= 0 || OH <= 0 || OW <= 0 || lddx < C) return DASS_ERR_ARG;
    const float sh = ac_scale(IH, OH), sw = ac_scale(IW, OW);
    hipStream_t st = (hipStream_t)stream;
    if (dy_nchw) {
        const int grid = dass_grid_1d((long)N * C * IH * IW, 256);
        const char *r4 = getenv("DASS_BILINEAR_R4");  // (0: the general gather, for A/B tests)
        if (IH > 1 && IW > 1 && OH - 1 == 4 * (IH - 1) && OW - 1 == 4 * (IW - 1) && !(r4 && r4[0] == '0')) {
            DASS_DISPATCH(dtype,
                          DASS_LAUNCH(bilinear_bwd_nchw_r4_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)dy, (float *)dx, lddx, N, IH, IW, C, OH, OW),
                          DASS_LAUNCH(bilinear_bwd_nchw_r4_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const float *)dy, (bf16_t *)dx, lddx, N, IH, IW, C, OH, OW))
        }
        DASS_DISPATCH(dtype,
                      DASS_LAUNCH(bilinear_bwd_nchw_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)dy, (float *)dx, lddx, N, IH, IW, C, OH, OW, sh, sw),
                      DASS_LAUNCH(bilinear_bwd_nchw_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const float *)dy, (bf16_t *)dx, lddx, N, IH, IW, C, OH, OW, sh, sw))
    }
    if (C % 4 || lddx % 4 || lddy % 4) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * IH * IW * (C / 4), 256);
    DASS_DISPATCH(dtype,
                  DASS_LAUNCH(bilinear_bwd_nhwc_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)dy, lddy, (float *)dx, lddx, N, IH, IW, C, OH, OW, sh, sw),
                  DASS_LAUNCH(bilinear_bwd_nhwc_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)dy, lddy, (bf16_t *)dx, lddx, N, IH, IW, C, OH, OW, sh, sw))
}
