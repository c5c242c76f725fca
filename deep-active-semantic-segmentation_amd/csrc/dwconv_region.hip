// Depthwise 3x3 convolution (MobileNetV2 InvertedResidual, models/backbone/mobilenet.py:49,59)
// forward / input-gradient / weight-gradient, and the region-selection helpers of
// active_selection/mc_dropout.py:82-155 (box-filter score maps, global min-max normalise, greedy
// square NMS).  Depthwise conv is bandwidth-bound: one lane owns 4 channels of one pixel, taps are
// 16-B loads that hit L1/L2 for the 9x re-read; the weight gradient reduces per-block partials in
// LDS and finishes with contiguous f32 atomics.
#include "dass_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const T *__restrict__ x, long ldx, const float *__restrict__ w,
                                                     T *__restrict__ y, long ldy, int N, int H, int W, int C, int OH,
                                                     int OW, int stride, int pad, int dil) {
    // one output ROW (n, oh) per blockIdx.y (grid-strided), threads over (ow, 4-channel group): no 64-bit index arithmetic per element
    const int cv = C >> 2;
    const int row_items = OW * cv;
    for (long row = blockIdx.y; row < (long)N * OH; row += gridDim.y) {
      const long n = row / OH;
      const int oh = (int)(row - n * OH);
      for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < row_items; j += gridDim.x * blockDim.x) {
        const int ow = j / cv;
        const int c = (j - ow * cv) << 2;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oh * stride - pad + r * dil;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ix = ow * stride - pad + s * dil;
                if (ix < 0 || ix >= W) continue;
                const f32x4 v = ld4<T>(x + ((n * H + iy) * W + ix) * ldx + c);
                const int tap = r * 3 + s;
                a[0] += v[0] * w[(c + 0) * 9 + tap];
                a[1] += v[1] * w[(c + 1) * 9 + tap];
                a[2] += v[2] * w[(c + 2) * 9 + tap];
                a[3] += v[3] * w[(c + 3) * 9 + tap];
            }
        }
        st4<T>(y + ((n * OH + oh) * OW + ow) * ldy + c, a);
      }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const T *__restrict__ dy, long lddy,
                                                          const float *__restrict__ w, T *__restrict__ dx, long lddx,
                                                          int N, int H, int W, int C, int OH, int OW, int stride,
                                                          int pad, int dil) {
    const int cv = C >> 2;
    const int row_items = W * cv;
    for (long row = blockIdx.y; row < (long)N * H; row += gridDim.y) {
      const long n = row / H;
      const int iy = (int)(row - n * H);
      for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < row_items; j += gridDim.x * blockDim.x) {
        const int ix = j / cv;
        const int c = (j - ix * cv) << 2;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ty = iy + pad - r * dil;
            if (ty < 0 || ty % stride) continue;
            const int oh = ty / stride;
            if (oh >= OH) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int tx = ix + pad - s * dil;
                if (tx < 0 || tx % stride) continue;
                const int ow = tx / stride;
                if (ow >= OW) continue;
                const f32x4 g = ld4<T>(dy + ((n * OH + oh) * OW + ow) * lddy + c);
                const int tap = r * 3 + s;
                a[0] += g[0] * w[(c + 0) * 9 + tap];
                a[1] += g[1] * w[(c + 1) * 9 + tap];
                a[2] += g[2] * w[(c + 2) * 9 + tap];
                a[3] += g[3] * w[(c + 3) * 9 + tap];
            }
        }
        st4<T>(dx + ((n * H + iy) * W + ix) * lddx + c, a);
      }
    }
}

// ---- round 4: the same two kernels with a FIXED 4-channel group per thread (blockDim = ppb * C/4 <= 256: thread t owns channel
// group t % (C/4) of the pixels t / (C/4) + k * ppb of its row).  The 36 weights of the group are loaded ONCE into registers: the
// kernels above fetch them again for every output -- 36 scalar loads next to the 9 tap loads, 45 vector-memory instructions per
// 16 bytes of output; the depthwise passes ran at 0.10-0.13 of HBM in the config-C profile (profiles/r03_train_C_mbv2_summary.md).
template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_cg_kernel(const T *__restrict__ x, long ldx, const float *__restrict__ w,
                                                        T *__restrict__ y, long ldy, int N, int H, int W, int C, int OH,
                                                        int OW, int stride, int pad, int dil, int cv, int ppb) {
    const int cg = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int c = cg << 2;
    f32x4 wr[9];  // wr[q] = floats 4 q .. 4 q + 3 of the group's 36 weights [4 channels][9 taps]
#pragma unroll
    for (int q = 0; q < 9; ++q) wr[q] = *reinterpret_cast<const f32x4 *>(w + (long)c * 9 + 4 * q);
    auto wt = [&](int e, int tap) -> float { const int i = e * 9 + tap; return wr[i >> 2][i & 3]; };
    for (long row = blockIdx.y; row < (long)N * OH; row += gridDim.y) {
        const long n = row / OH;
        const int oh = (int)(row - n * OH);
        for (int ow = blockIdx.x * ppb + pl; ow < OW; ow += gridDim.x * ppb) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oh * stride - pad + r * dil;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int ix = ow * stride - pad + s * dil;
                    if (ix < 0 || ix >= W) continue;
                    const f32x4 v = ld4<T>(x + ((n * H + iy) * W + ix) * ldx + c);
                    const int tap = r * 3 + s;
                    a[0] += v[0] * wt(0, tap);
                    a[1] += v[1] * wt(1, tap);
                    a[2] += v[2] * wt(2, tap);
                    a[3] += v[3] * wt(3, tap);
                }
            }
            st4<T>(y + ((n * OH + oh) * OW + ow) * ldy + c, a);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_data_cg_kernel(const T *__restrict__ dy, long lddy, const float *__restrict__ w,
                                                             T *__restrict__ dx, long lddx, int N, int H, int W, int C, int OH,
                                                             int OW, int stride, int pad, int dil, int cv, int ppb) {
    const int cg = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int c = cg << 2;
    f32x4 wr[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) wr[q] = *reinterpret_cast<const f32x4 *>(w + (long)c * 9 + 4 * q);
    auto wt = [&](int e, int tap) -> float { const int i = e * 9 + tap; return wr[i >> 2][i & 3]; };
    for (long row = blockIdx.y; row < (long)N * H; row += gridDim.y) {
        const long n = row / H;
        const int iy = (int)(row - n * H);
        for (int ix = blockIdx.x * ppb + pl; ix < W; ix += gridDim.x * ppb) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int ty = iy + pad - r * dil;
                if (ty < 0 || ty % stride) continue;
                const int oh = ty / stride;
                if (oh >= OH) continue;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int tx = ix + pad - s * dil;
                    if (tx < 0 || tx % stride) continue;
                    const int ow = tx / stride;
                    if (ow >= OW) continue;
                    const f32x4 g = ld4<T>(dy + ((n * OH + oh) * OW + ow) * lddy + c);
                    const int tap = r * 3 + s;
                    a[0] += g[0] * wt(0, tap);
                    a[1] += g[1] * wt(1, tap);
                    a[2] += g[2] * wt(2, tap);
                    a[3] += g[3] * wt(3, tap);
                }
            }
            st4<T>(dx + ((n * H + iy) * W + ix) * lddx + c, a);
        }
    }
}

// grid (C/64, pixel slabs); thread = (4-channel group cq of 16, pixel lane pl of 16): 9 taps x 4 channels of accumulators, 16-B
// loads (16 lanes = 256 contiguous bytes of one pixel), the (n, oh, ow) cursor advanced by adds -- the round-1 kernel loaded
// single floats and paid two 64-bit divisions per pixel (0.44 TB/s of algorithmic traffic in the config-C profile; this: see
// profiles/r03_train_C_mbv2_summary.md)
template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const T *__restrict__ x, long ldx,
                                                            const T *__restrict__ dy, long lddy,
                                                            float *__restrict__ dw, int N, int H, int W, int C,
                                                            int OH, int OW, int stride, int pad, int dil,
                                                            long pix_per_block) {
    __shared__ float red[16][9][64 + 4];
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const long M = (long)N * OH * OW;
    const long p0 = (long)blockIdx.y * pix_per_block;
    long p1 = p0 + pix_per_block;
    if (p1 > M) p1 = M;
    f32x4 a[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const long ohw = (long)OH * OW;
        long p = p0 + pl;
        long n = p / ohw;
        const long rem = p - n * ohw;
        int oh = (int)(rem / OW), ow = (int)(rem - (long)oh * OW);
        for (; p < p1; p += 16) {
            const f32x4 g = ld4<T>(dy + p * lddy + c);
            const T *xn = x + n * H * W * ldx + c;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oh * stride - pad + r * dil;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    const int ix = ow * stride - pad + s2 * dil;
                    if (ix < 0 || ix >= W) continue;
                    a[r * 3 + s2] += g * ld4<T>(xn + ((long)iy * W + ix) * ldx);
                }
            }
            ow += 16;
            while (ow >= OW) {
                ow -= OW;
                if (++oh >= OH) {
                    oh = 0;
                    ++n;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[pl][t][cq * 4 + e] = a[t][e];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
        const int t = i >> 6, cl = i & 63;
        const int cc = blockIdx.x * 64 + cl;
        if (cc >= C) continue;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][t][cl];
        atomicAdd(dw + (long)cc * 9 + t, v);
    }
}

// Strip form of the depthwise forward and (stride 1) input-gradient passes, f32: out[row][o] = sum over taps of w[tap] * in[row * S - pad
// + r * D][o * S - pad + s * D].  Same walk as the weight-gradient strip kernel below: a thread owns 4 channels (their 36 weights in
// registers, per tap one 4-channel vector) and 16 consecutive outputs of one row with the 3 x (2 D + 1) input window in registers -- S new
// columns per output instead of 9 loads behind 9 branches (dw_fwd_cg_kernel ran at 2.3 TB/s, the input gradient at 1.5).  The input
// gradient of a stride-1 conv is this form over dy with pad' = 2 D - pad and the taps flipped (FLIP); the products are added in the
// ORIGINAL tap order either way, so results equal the kernels above bit for bit (an invalid tap adds +0 instead of nothing).
struct DwBnLink {  // BNS: the layer whose output this launch's result is the gradient of (dass_dwconv3x3_bwd_data_bnstats)
    const float *y, *mean, *invstd, *gsc, *gsh;  // its conv output [rows][C] and per-channel mean / 1/std / gate scale / gate shift
    double *sums;                                // [2][C] f64 (sum dz, sum dz xhat) + C floats (max |dz|)
    int act;
};

template <int DIL, int STRIDE, bool FLIP, int MODE>  // MODE 0: conv only; 1: + BN-backward sums of the producing layer; 2: + sum / sum of squares of the output
__global__ __launch_bounds__(256) void dw_conv_strip_kernel(const float *__restrict__ in, int ldi, const float *__restrict__ w,
                                                            float *__restrict__ out, int ldo, int IH, int IW, int C, int OH, int OW,
                                                            int pad, int nsegw, int nstrips, int seg, const DwBnLink bl) {
    constexpr int WW = 2 * DIL + 1, KEEP = WW - STRIDE, PF = 3;  // (seg: outputs per strip = ceil(OW / nsegw) <= 16, e.g. 3 x 11 at OW = 33)
    constexpr bool BNS = MODE == 1;
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 bs0 = zero, bs1 = zero, bsm = zero;  // BNS: this thread's share of sum dz, sum dz xhat, max |dz| of its 4 channels
    if (c < C) {
    f32x4 mu = zero, is = zero, gsc = zero, gsh = zero;
    if constexpr (BNS) {
        mu = *reinterpret_cast<const f32x4 *>(bl.mean + c);
        is = *reinterpret_cast<const f32x4 *>(bl.invstd + c);
        if (bl.act != DASS_ACT_NONE) {
            gsc = *reinterpret_cast<const f32x4 *>(bl.gsc + c);
            gsh = *reinterpret_cast<const f32x4 *>(bl.gsh + c);
        }
    }
    f32x4 wv[9];  // wv[tap] = the tap's weight of the 4 channels
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[t][e] = w[(long)(c + e) * 9 + t];
    for (int strip = blockIdx.y * 16 + pl; strip < nstrips; strip += gridDim.y * 16) {
        const int row = strip / nsegw, sg = strip - row * nsegw;
        const int n = row / OH, oh = row - n * OH;
        const int o0 = sg * seg, ix0 = o0 * STRIDE - pad;
        const int o_end = o0 + seg < OW ? o0 + seg : OW;
        const float *xr[3];
        bool rok[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oh * STRIDE - pad + r * DIL;
            rok[r] = (unsigned)iy < (unsigned)IH;
            xr[r] = in + ((n * IH + (rok[r] ? iy : 0)) * IW) * ldi + c;
        }
        float *op = out + (row * OW + o0) * ldo + c;
        const float *yp = BNS ? bl.y + (row * OW + o0) * C + c : nullptr;
        f32x4 win[3][WW];
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            const int ix = ix0 + j;
            const bool cok = (unsigned)ix < (unsigned)IW;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xr[r] + ((rok[r] & cok) ? ix : 0) * ldi);
                win[r][j] = (rok[r] & cok) ? v : zero;
            }
        }
        f32x4 nc[PF][STRIDE][3], ny[BNS ? PF : 1];
        auto issue = [&](int i, int slot) {  // raw loads from clamped addresses; invalid taps are zeroed where the value is used
            if constexpr (BNS) ny[slot] = *reinterpret_cast<const f32x4 *>(yp + (o0 + i < o_end ? i : 0) * C);
#pragma unroll
            for (int j = 0; j < STRIDE; ++j) {
                const int ix = ix0 + i * STRIDE + KEEP + j;
                const bool cok = (unsigned)ix < (unsigned)IW;
#pragma unroll
                for (int r = 0; r < 3; ++r) nc[slot][j][r] = *reinterpret_cast<const f32x4 *>(xr[r] + ((rok[r] & cok) ? ix : 0) * ldi);
            }
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) issue(i, i);
        for (int i0 = 0; i0 < seg; i0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {  // (slot u: the ring index stays a compile-time constant)
                const int i = i0 + u;
#pragma unroll
                for (int j = 0; j < STRIDE; ++j) {
                    const bool cok = (unsigned)(ix0 + i * STRIDE + KEEP + j) < (unsigned)IW;
#pragma unroll
                    for (int r = 0; r < 3; ++r) win[r][KEEP + j] = (rok[r] & cok) ? nc[u][j][r] : zero;
                }
                f32x4 yl = zero;
                if constexpr (BNS) yl = ny[u];  // (before the slot is re-issued)
                issue(i + PF, u);  // (past the strip: clamped addresses, values never used)
                f32x4 a = zero;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2) a += win[FLIP ? 2 - r : r][(FLIP ? 2 - s2 : s2) * DIL] * wv[r * 3 + s2];
                if (o0 + i < o_end) *reinterpret_cast<f32x4 *>(op + i * ldo) = a;
                if constexpr (MODE == 2) {  // train-mode BN behind this conv: its batch statistics (dass_channel_sums' pass, fused)
                    const f32x4 v = (o0 + i < o_end) ? a : zero;
                    bs0 += v;
                    bs1 += v * v;
                }
                if constexpr (BNS) {
                    // dz = dx * act'(out) with out re-derived from the layer's conv output by the forward's own fma (bn_affine), xhat = (y - mean) / std
                    f32x4 gz = (o0 + i < o_end) ? a : zero;
                    if (bl.act != DASS_ACT_NONE) {
                        const f32x4 o = bn_affine(yl, gsc, gsh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) gz[e] *= act_grad_from_out(o[e], bl.act);
                    }
                    bs0 += gz;
                    bs1 += gz * ((yl - mu) * is);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bsm[e] = fmaxf(bsm[e], fabsf(gz[e]));
                }
#pragma unroll
                for (int j = 0; j < KEEP; ++j)
#pragma unroll
                    for (int r = 0; r < 3; ++r) win[r][j] = win[r][j + STRIDE];
            }
        }
    }
    }  // c < C
    if constexpr (MODE != 0) {
        // the 16 strip lanes of a channel group fold through LDS, then one f64 atomic per channel and workgroup (as the conv kernels do)
        __shared__ float red[3][16][64 + 4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][pl][cq * 4 + e] = bs0[e];
            red[1][pl][cq * 4 + e] = bs1[e];
            red[2][pl][cq * 4 + e] = bsm[e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < (MODE == 2 ? 2 : 3) * 64; i += 256) {
            const int which = i >> 6, cl = i & 63, cc = blockIdx.x * 64 + cl;
            if (cc >= C) continue;
            if (which < 2) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) v += red[which][q][cl];
                unsafeAtomicAdd(bl.sums + (long)which * C + cc, (double)v);
            } else {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) v = fmaxf(v, red[2][q][cl]);
                if (!(v >= 0.f)) v = __uint_as_float(0x7f800000u);  // NaN: an infinite bound
                unsigned *slot = reinterpret_cast<unsigned *>(bl.sums + 2 * (long)C) + cc;
                if (__float_as_uint(v) > *reinterpret_cast<volatile unsigned *>(slot)) atomicMax(slot, __float_as_uint(v));
            }
        }
    }
}

// 1: launched; 0: outside the specialisation
template <bool FLIP, int MODE = 0>
static int launch_dw_strip(const float *in, long ldi, const float *w, float *out, long ldo, int N, int IH, int IW, int C, int OH, int OW, int stride,
                           int pad, int dil, hipStream_t st, const DwBnLink bl = DwBnLink{}) {
    const char *e = getenv("DASS_DW_STRIP");
    if (e && e[0] == '0') return 0;
    if (!(dil == 1 || dil == 2) || !(stride == 1 || stride == 2) || (FLIP && stride != 1)) return 0;
    if ((long)N * IH * IW * ldi >= (1l << 31) || (long)N * OH * OW * ldo >= (1l << 31)) return 0;
    const int cblocks = (C + 63) / 64, nsegw = (OW + 15) / 16, seg = (OW + nsegw - 1) / nsegw;
    const long nstrips = (long)N * OH * nsegw;
    // (launches that end in per-channel atomics -- MODE 1 / 2 -- get fewer, longer blocks: DASS_DW_SUM_BLOCKS)
    static const long sum_target = getenv("DASS_DW_SUM_BLOCKS") ? atol(getenv("DASS_DW_SUM_BLOCKS")) : 2048;
    long gy = ((MODE != 0 ? sum_target : 2048) + cblocks - 1) / cblocks;
    if (gy > (nstrips + 15) / 16) gy = (nstrips + 15) / 16;
    const dim3 grid(cblocks, (unsigned)(gy < 1 ? 1 : gy));
#define DASS_DW_CS(D, S)                                                                                                                   \
    DASS_LAUNCH((dw_conv_strip_kernel<D, S, FLIP, MODE>), grid, dim3(256), 0, st, in, (int)ldi, w, out, (int)ldo, IH, IW, C, OH, OW, pad, nsegw, \
                (int)nstrips, seg, bl)
    if (dil == 1 && stride == 1) DASS_DW_CS(1, 1);
    else if (dil == 2 && stride == 1) DASS_DW_CS(2, 1);
    else if constexpr (!FLIP) {
        if (dil == 1) DASS_DW_CS(1, 2);
        else DASS_DW_CS(2, 2);
    }
#undef DASS_DW_CS
    return 1;
}

// Strip form of the depthwise weight gradient (f32, dilation 1 or 2, stride 1 or 2 -- every depthwise conv of MobileNetV2 and
// Xception at os16, reference models/backbone/mobilenet.py:33-79).  A thread owns 4 channels and walks a strip of 16 consecutive
// output pixels of ONE output row with the 3 x (2 dil + 1) input window of its channels in registers: a step loads `stride` new
// input columns (3 rows) and one dy value instead of 9 + 1, and every address is a 32-bit add (the pixel-cursor kernel above spent
// its time in 64-bit multiplies for 9 tap addresses per pixel: 0.8 TB/s of algorithmic traffic in profiles/r04_train_C_mbv2_summary.md).
// Loads are unconditional from clamped addresses, invalid taps are selected to zero: no branches in the strip.
template <int DIL, int STRIDE>
__global__ __launch_bounds__(256) void dw_bwd_weight_strip_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ dy, int lddy,
                                                                  float *__restrict__ dw, int H, int W, int C, int OH, int OW, int pad,
                                                                  int nsegw, int nstrips, int seg) {
    constexpr int WW = 2 * DIL + 1, KEEP = WW - STRIDE;  // (seg: outputs per strip = ceil(OW / nsegw) <= 16)
    static_assert(KEEP >= 1, "window");
    __shared__ float red[16][9][64 + 4];
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    f32x4 a[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int strip = blockIdx.y * 16 + pl; strip < nstrips; strip += gridDim.y * 16) {
            const int row = strip / nsegw, sg = strip - row * nsegw;  // (one division pair per 16 pixels)
            const int n = row / OH, oh = row - n * OH;
            const int ow0 = sg * seg, ix0 = ow0 * STRIDE - pad;
            const int o_end = ow0 + seg < OW ? ow0 + seg : OW;
            const float *xr[3];
            bool rok[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oh * STRIDE - pad + r * DIL;
                rok[r] = (unsigned)iy < (unsigned)H;
                xr[r] = x + ((n * H + (rok[r] ? iy : 0)) * W) * ldx + c;
            }
            const float *dyp = dy + (row * OW + ow0) * lddy + c;
            auto ldcol = [&](int r, int ix) -> f32x4 {
                const bool ok = rok[r] & ((unsigned)ix < (unsigned)W);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xr[r] + (ok ? ix : 0) * ldx);
                return ok ? v : zero;
            };
            f32x4 win[3][WW];
#pragma unroll
            for (int j = 0; j < KEEP; ++j)
#pragma unroll
                for (int r = 0; r < 3; ++r) win[r][j] = ldcol(r, ix0 + j);
            // the loads of step i + PF are issued when step i has consumed its slot: PF steps of loads stay in flight (without the
            // ring the compiler waits for every step's four loads right where it issued them: vmcnt(0) sixteen times per strip)
            constexpr int PF = 3;
            f32x4 nc[PF][STRIDE][3], ng[PF];
            // (raw loads from clamped addresses here; the select-to-zero of invalid taps happens where the value is USED -- a select
            //  next to its load makes the compiler wait for the load on the spot)
            auto issue = [&](int i, int slot) {
                const bool live = ow0 + i < o_end;
                ng[slot] = *reinterpret_cast<const f32x4 *>(dyp + (live ? i : 0) * lddy);
#pragma unroll
                for (int j = 0; j < STRIDE; ++j) {
                    const int ix = ix0 + i * STRIDE + KEEP + j;
                    const bool cok = (unsigned)ix < (unsigned)W;
#pragma unroll
                    for (int r = 0; r < 3; ++r) nc[slot][j][r] = *reinterpret_cast<const f32x4 *>(xr[r] + ((rok[r] & cok) ? ix : 0) * ldx);
                }
            };
#pragma unroll
            for (int i = 0; i < PF; ++i) issue(i, i);
            for (int i0 = 0; i0 < seg; i0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {  // (slot u: the ring index stays a compile-time constant)
                    const int i = i0 + u;
                    const f32x4 g = (ow0 + i < o_end) ? ng[u] : zero;
#pragma unroll
                    for (int j = 0; j < STRIDE; ++j) {
                        const bool cok = (unsigned)(ix0 + i * STRIDE + KEEP + j) < (unsigned)W;
#pragma unroll
                        for (int r = 0; r < 3; ++r) win[r][KEEP + j] = (rok[r] & cok) ? nc[u][j][r] : zero;
                    }
                    issue(i + PF, u);  // (past the strip: clamped addresses, values never used)
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int s2 = 0; s2 < 3; ++s2) a[r * 3 + s2] += g * win[r][s2 * DIL];
#pragma unroll
                    for (int j = 0; j < KEEP; ++j)
#pragma unroll
                        for (int r = 0; r < 3; ++r) win[r][j] = win[r][j + STRIDE];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[pl][t][cq * 4 + e] = a[t][e];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
        const int t = i >> 6, cl = i & 63;
        const int cc = blockIdx.x * 64 + cl;
        if (cc >= C) continue;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][t][cl];
        atomicAdd(dw + (long)cc * 9 + t, v);
    }
}

// ---------------------------------------------------------------------------------------- regions
// horizontal then vertical running box sums (valid), f64 accumulators -> f32
__global__ void box_rows_kernel(const float *__restrict__ in, float *__restrict__ tmp, int N, int H, int W, int r) {
    const int OWd = W - r + 1;
    const long total = (long)N * H * OWd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OWd);
        const long row = i / OWd;
        const float *p = in + row * W + ox;
        double s = 0.0;
        for (int j = 0; j < r; ++j) s += (double)p[j];
        tmp[i] = (float)s;
    }
}
__global__ void box_cols_kernel(const float *__restrict__ tmp, float *__restrict__ out, int N, int H, int W, int r) {
    const int OWd = W - r + 1, OHd = H - r + 1;
    const long total = (long)N * OHd * OWd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OWd);
        long t = i / OWd;
        const int oy = (int)(t % OHd);
        const long n = t / OHd;
        const float *p = tmp + (n * H + oy) * OWd + ox;
        double s = 0.0;
        for (int j = 0; j < r; ++j) s += (double)p[(long)j * OWd];
        out[i] = (float)s;
    }
}

__global__ void zero_rect_kernel(float *__restrict__ maps, long base, int W, int r0, int r1, int c0, int c1) {
    const int h = r1 - r0, w = c1 - c0;
    const long total = (long)h * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        maps[base + (long)(r0 + y) * W + c0 + x] = 0.f;
    }
}

__global__ __launch_bounds__(256) void minmax_stage1_kernel(const float *__restrict__ v, long n,
                                                            float *__restrict__ partial) {
    __shared__ float smin[256], smax[256];
    float lo = INFINITY, hi = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = v[i];
        lo = fminf(lo, x);
        hi = fmaxf(hi, x);
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = smin[0];
        partial[blockIdx.x * 2 + 1] = smax[0];
    }
}
__global__ __launch_bounds__(256) void minmax_stage2_kernel(const float *__restrict__ partial, int blocks,
                                                            float *__restrict__ out) {
    __shared__ float smin[256], smax[256];
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < blocks; i += 256) {
        lo = fminf(lo, partial[i * 2]);
        hi = fmaxf(hi, partial[i * 2 + 1]);
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = smin[0];
        out[1] = smax[0];
    }
}
// x.add_(-min).mul_(1/(max-min)) exactly as mc_dropout.py:154
__global__ void affine_kernel(float *__restrict__ v, long n, const float *__restrict__ mm) {
    const float lo = mm[0];
    const float inv = 1.0f / (mm[1] - mm[0]);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        v[i] = (v[i] + (-lo)) * inv;
}

// per-image (max, first index) of [N][HW] maps: grid N
__global__ __launch_bounds__(256) void image_argmax_kernel(const float *__restrict__ maps, long HW,
                                                           float *__restrict__ imax, int *__restrict__ iarg) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const long n = blockIdx.x;
    const float *m = maps + n * HW;
    float best = -INFINITY;
    int bi = -1;
    for (long i = threadIdx.x; i < HW; i += 256) {
        const float x = m[i];
        if (bi < 0 || x > best) {
            best = x;
            bi = (int)i;
        }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float x = sv[threadIdx.x + o];
            const int j = si[threadIdx.x + o], cur = si[threadIdx.x];
            if (j >= 0 && (cur < 0 || x > sv[threadIdx.x] || (x == sv[threadIdx.x] && j < cur))) {
                sv[threadIdx.x] = x;
                si[threadIdx.x] = j;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        imax[n] = sv[0];
        iarg[n] = si[0];
    }
}

// Greedy square NMS (mc_dropout.py:82-108) as ONE single-workgroup kernel over the per-image
// (max, argmax) cache: pick the global first-max, record (image,row,col), zero the clipped
// (2*region)^2 box anchored at (row-region, col-region), refresh that image's cache, stop when the
// global max < 0.01 or after max_picks.  Sequential by definition; 1024 lanes do each step's scans.
__global__ __launch_bounds__(1024) void square_nms_kernel(float *__restrict__ maps, int N, int H, int W, int region,
                                                          int max_picks, float *__restrict__ imax,
                                                          int *__restrict__ iarg, int *__restrict__ picks,
                                                          int *__restrict__ count) {
    __shared__ float sv[1024];
    __shared__ long si[1024];
    __shared__ int s_img, s_pos, s_stop;
    const int tid = threadIdx.x;
    const long HW = (long)H * W;
    int npicks = 0;
    for (int it = 0; it < max_picks; ++it) {
        // global first-max over images: flat index = n*HW + iarg[n]
        float best = -INFINITY;
        long bi = -1;
        for (int n = tid; n < N; n += 1024) {
            const float x = imax[n];
            const long f = (long)n * HW + iarg[n];
            if (bi < 0 || x > best || (x == best && f < bi)) {
                best = x;
                bi = f;
            }
        }
        sv[tid] = best;
        si[tid] = bi;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) {
                const float x = sv[tid + o];
                const long j = si[tid + o], cur = si[tid];
                if (j >= 0 && (cur < 0 || x > sv[tid] || (x == sv[tid] && j < cur))) {
                    sv[tid] = x;
                    si[tid] = j;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            const long f = si[0];
            s_img = (int)(f / HW);
            s_pos = (int)(f - (long)s_img * HW);
            picks[npicks * 3 + 0] = s_img;
            picks[npicks * 3 + 1] = s_pos / W;
            picks[npicks * 3 + 2] = s_pos % W;
        }
        __syncthreads();
        ++npicks;
        const int img = s_img, r = s_pos / W, c = s_pos % W;
        const int r0 = max(0, r - region), c0 = max(0, c - region);
        const int r1 = min(H, r + region), c1 = min(W, c + region);
        float *m = maps + (long)img * HW;
        const int bw = c1 - c0, bh = r1 - r0;
        for (int i = tid; i < bw * bh; i += 1024) m[(long)(r0 + i / bw) * W + c0 + i % bw] = 0.f;
        __syncthreads();
        // refresh this image's cache
        best = -INFINITY;
        bi = -1;
        for (long i = tid; i < HW; i += 1024) {
            const float x = m[i];
            if (bi < 0 || x > best) {
                best = x;
                bi = i;
            }
        }
        sv[tid] = best;
        si[tid] = bi;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) {
                const float x = sv[tid + o];
                const long j = si[tid + o], cur = si[tid];
                if (j >= 0 && (cur < 0 || x > sv[tid] || (x == sv[tid] && j < cur))) {
                    sv[tid] = x;
                    si[tid] = j;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            imax[img] = sv[0];
            iarg[img] = (int)si[0];
        }
        __syncthreads();
        // stop test: score_maps.max() < 0.01
        float gm = -INFINITY;
        for (int n = tid; n < N; n += 1024) gm = fmaxf(gm, imax[n]);
        sv[tid] = gm;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) sv[tid] = fmaxf(sv[tid], sv[tid + o]);
            __syncthreads();
        }
        if (tid == 0) s_stop = sv[0] < 0.01f ? 1 : 0;
        __syncthreads();
        if (s_stop) break;
    }
    if (tid == 0) count[0] = npicks;
}

}  // namespace

#define DW_ARGS_OK (x_ && y_ && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && OH > 0 && OW > 0 && stride >= 1 && dil >= 1)

extern "C" int dass_dwconv3x3_fwd(const void *x, int64_t ldx, const float *w, void *y, int64_t ldy, int N, int H,
                                  int W, int C, int OH, int OW, int stride, int pad, int dil, int dtype, void *stream) {
    const void *x_ = x;
    const void *y_ = y;
    if (!DW_ARGS_OK || !w || ldx % 4 || ldy % 4) return DASS_ERR_ARG;
    const int gx = (OW * (C / 4) + 255) / 256;
    const long rows_ = (long)N * OH;
    const dim3 grid((unsigned)(gx < 1024 ? gx : 1024), (unsigned)(rows_ < 8192 ? rows_ : 8192));  // (ow, channel group) x output rows
    hipStream_t st = (hipStream_t)stream;
    const int cv = C / 4;
    if (dtype == DASS_F32 && launch_dw_strip<false>((const float *)x, ldx, w, (float *)y, ldy, N, H, W, C, OH, OW, stride, pad, dil, st)) {
        DASS_LAUNCH_CHECK();
        return DASS_OK;
    }
    if (cv <= 256 && !((uintptr_t)w & 15) && (dtype == DASS_F32 || dtype == DASS_BF16)) {  // one channel group per thread, weights in registers
        const int ppb = 256 / cv;
        const dim3 g2((unsigned)((OW + ppb - 1) / ppb), grid.y), b2((unsigned)(ppb * cv));
        if (dtype == DASS_F32)
            DASS_LAUNCH(dw_fwd_cg_kernel<float>, g2, b2, 0, st, (const float *)x, ldx, w, (float *)y, ldy, N, H, W, C, OH, OW, stride, pad, dil, cv, ppb);
        else
            DASS_LAUNCH(dw_fwd_cg_kernel<bf16_t>, g2, b2, 0, st, (const bf16_t *)x, ldx, w, (bf16_t *)y, ldy, N, H, W, C, OH, OW, stride, pad, dil, cv, ppb);
        DASS_LAUNCH_CHECK();
        return DASS_OK;
    }
    if (dtype == DASS_F32)
        DASS_LAUNCH(dw_fwd_kernel<float>, grid, dim3(256), 0, st, (const float *)x, ldx, w, (float *)y, ldy, N, H, W, C, OH, OW, stride, pad, dil);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(dw_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)x, ldx, w, (bf16_t *)y, ldy, N, H, W, C, OH, OW, stride, pad, dil);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_dwconv3x3_bwd_data(const void *dy, int64_t lddy, const float *w, void *dx, int64_t lddx, int N,
                                       int H, int W, int C, int OH, int OW, int stride, int pad, int dil, int dtype,
                                       void *stream) {
    const void *x_ = dy;
    const void *y_ = dx;
    if (!DW_ARGS_OK || !w || lddx % 4 || lddy % 4) return DASS_ERR_ARG;
    const int gx = (W * (C / 4) + 255) / 256;
    const long rows_ = (long)N * H;
    const dim3 grid((unsigned)(gx < 1024 ? gx : 1024), (unsigned)(rows_ < 8192 ? rows_ : 8192));  // (ix, channel group) x input rows
    hipStream_t st = (hipStream_t)stream;
    const int cv = C / 4;
    // (stride 1: the input gradient is the forward form over dy with pad' = 2 dil - pad and flipped taps; output rows = the H x W of dx)
    if (dtype == DASS_F32 && stride == 1 && 2 * dil - pad >= 0 &&
        launch_dw_strip<true>((const float *)dy, lddy, w, (float *)dx, lddx, N, OH, OW, C, H, W, 1, 2 * dil - pad, dil, st)) {
        DASS_LAUNCH_CHECK();
        return DASS_OK;
    }
    if (cv <= 256 && !((uintptr_t)w & 15) && (dtype == DASS_F32 || dtype == DASS_BF16)) {
        const int ppb = 256 / cv;
        const dim3 g2((unsigned)((W + ppb - 1) / ppb), grid.y), b2((unsigned)(ppb * cv));
        if (dtype == DASS_F32)
            DASS_LAUNCH(dw_bwd_data_cg_kernel<float>, g2, b2, 0, st, (const float *)dy, lddy, w, (float *)dx, lddx, N, H, W, C, OH, OW, stride, pad, dil, cv, ppb);
        else
            DASS_LAUNCH(dw_bwd_data_cg_kernel<bf16_t>, g2, b2, 0, st, (const bf16_t *)dy, lddy, w, (bf16_t *)dx, lddx, N, H, W, C, OH, OW, stride, pad, dil, cv, ppb);
        DASS_LAUNCH_CHECK();
        return DASS_OK;
    }
    if (dtype == DASS_F32)
        DASS_LAUNCH(dw_bwd_data_kernel<float>, grid, dim3(256), 0, st, (const float *)dy, lddy, w, (float *)dx, lddx, N, H, W, C, OH, OW, stride, pad, dil);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(dw_bwd_data_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)dy, lddy, w, (bf16_t *)dx, lddx, N, H, W, C, OH, OW, stride, pad, dil);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

/* dass_dwconv3x3_fwd (f32, stride 1 or 2, dilation 1 or 2) that also adds every output channel's sum and sum of squares into sums
 * ([2][C] f64, zeroed by the caller): the statistics of the train-mode BN behind the conv (models/backbone/mobilenet.py:45-50) without
 * dass_channel_sums' pass over y.  DASS_ERR_UNSUPPORTED outside the specialisation: the caller runs the two passes. */
extern "C" int dass_dwconv3x3_fwd_sums(const void *x, int64_t ldx, const float *w, void *y, int64_t ldy, int N, int H, int W, int C, int OH,
                                       int OW, int stride, int pad, int dil, double *sums, void *stream) {
    const void *x_ = x;
    const void *y_ = y;
    if (!DW_ARGS_OK || !w || ldx % 4 || ldy % 4 || !sums) return DASS_ERR_ARG;
    DwBnLink bl{nullptr, nullptr, nullptr, nullptr, nullptr, sums, DASS_ACT_NONE};
    if (!launch_dw_strip<false, 2>((const float *)x, ldx, w, (float *)y, ldy, N, H, W, C, OH, OW, stride, pad, dil, (hipStream_t)stream, bl))
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

/* dass_dwconv3x3_bwd_data whose result dx [N*H*W][C] IS the gradient d_out of the conv + BN (+ act) layer that produced the depthwise
 * conv's input (MobileNetV2: the expand 1x1 of an inverted-residual block, models/backbone/mobilenet.py:52-58): the launch also adds that
 * layer's BN-backward sums into bn_sums ([2][C] f64 + C floats, zeroed by the caller), as dass_conv2d_x3_dgrad_bnstats does for dense convs.
 * f32, stride 1, dilation 1 or 2, lddx == C; DASS_ERR_UNSUPPORTED otherwise (the caller runs the two passes). */
extern "C" int dass_dwconv3x3_bwd_data_bnstats(const void *dy, int64_t lddy, const float *w, void *dx, int64_t lddx, int N, int H, int W, int C,
                                               int OH, int OW, int stride, int pad, int dil, const float *bn_y, const float *bn_mean,
                                               const float *bn_invstd, const float *gate_scale, const float *gate_shift, int bn_act,
                                               double *bn_sums, void *stream) {
    const void *x_ = dy;
    const void *y_ = dx;
    if (!DW_ARGS_OK || !w || lddx % 4 || lddy % 4 || !bn_y || !bn_mean || !bn_invstd || !bn_sums) return DASS_ERR_ARG;
    if (bn_act != DASS_ACT_NONE && (!gate_scale || !gate_shift)) return DASS_ERR_ARG;
    if (stride != 1 || 2 * dil - pad < 0 || lddx != C) return DASS_ERR_UNSUPPORTED;
    DwBnLink bl{bn_y, bn_mean, bn_invstd, gate_scale, gate_shift, bn_sums, bn_act};
    if (!launch_dw_strip<true, 1>((const float *)dy, lddy, w, (float *)dx, lddx, N, OH, OW, C, H, W, 1, 2 * dil - pad, dil, (hipStream_t)stream, bl))
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_dwconv3x3_bwd_weight(const void *x, int64_t ldx, const void *dy, int64_t lddy, float *dw, int N,
                                         int H, int W, int C, int OH, int OW, int stride, int pad, int dil, int dtype,
                                         void *stream) {
    const void *x_ = x;
    const void *y_ = dy;
    if (!DW_ARGS_OK || !dw || ldx % 4 || lddy % 4) return DASS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dw, 0, sizeof(float) * (size_t)C * 9, st) != hipSuccess) return DASS_ERR_LAUNCH;
    const long M = (long)N * OH * OW;
    const int cblocks = (C + 63) / 64;
    long slabs = (1024 + cblocks - 1) / cblocks;
    const long maxslabs = (M + 255) / 256;
    if (slabs > maxslabs) slabs = maxslabs;
    if (slabs < 1) slabs = 1;
    const long ppb = (M + slabs - 1) / slabs;
    slabs = (M + ppb - 1) / ppb;
    dim3 grid(cblocks, (unsigned)slabs);
    const bool strip_on = !(getenv("DASS_DW_WGRAD_STRIP") && getenv("DASS_DW_WGRAD_STRIP")[0] == '0');
    if (dtype == DASS_F32 && strip_on && (dil == 1 || dil == 2) && (stride == 1 || stride == 2) && (long)N * H * W * ldx < (1l << 31) &&
        M * lddy < (1l << 31)) {
        // strips of 16 output pixels of one row; ~8 blocks per CU, every thread several strips
        const int nsegw = (OW + 15) / 16, seg = (OW + nsegw - 1) / nsegw;
        const long nstrips = (long)N * OH * nsegw;
        const long tgt = getenv("DASS_DW_GY") ? atol(getenv("DASS_DW_GY")) : 512;
        long gy = (tgt + cblocks - 1) / cblocks;
        if (gy > (nstrips + 15) / 16) gy = (nstrips + 15) / 16;
        const dim3 g2(cblocks, (unsigned)(gy < 1 ? 1 : gy));
#define DASS_DW_STRIP(D, S)                                                                                                              \
    DASS_LAUNCH((dw_bwd_weight_strip_kernel<D, S>), g2, dim3(256), 0, st, (const float *)x, (int)ldx, (const float *)dy, (int)lddy, dw, H, W, \
                C, OH, OW, pad, nsegw, (int)nstrips, seg)
        if (dil == 1 && stride == 1) DASS_DW_STRIP(1, 1);
        else if (dil == 1) DASS_DW_STRIP(1, 2);
        else if (stride == 1) DASS_DW_STRIP(2, 1);
        else DASS_DW_STRIP(2, 2);
#undef DASS_DW_STRIP
        DASS_LAUNCH_CHECK();
        return DASS_OK;
    }
    if (dtype == DASS_F32)
        DASS_LAUNCH(dw_bwd_weight_kernel<float>, grid, dim3(256), 0, st, (const float *)x, ldx, (const float *)dy, lddy, dw, N, H, W, C, OH, OW, stride, pad, dil, ppb);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(dw_bwd_weight_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)x, ldx, (const bf16_t *)dy, lddy, dw, N, H, W, C, OH, OW, stride, pad, dil, ppb);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_box_sum(const float *maps, float *out, float *tmp, int N, int H, int W, int r, void *stream) {
    if (!maps || !out || !tmp || N <= 0 || r <= 0 || r > H || r > W) return DASS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DASS_LAUNCH(box_rows_kernel, dim3(dass_grid_1d((long)N * H * (W - r + 1), 256)), dim3(256), 0, st, maps, tmp, N, H, W, r);
    DASS_LAUNCH_CHECK();
    DASS_LAUNCH(box_cols_kernel, dim3(dass_grid_1d((long)N * (H - r + 1) * (W - r + 1), 256)), dim3(256), 0, st, tmp, out, N, H, W, r);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_zero_rect(float *maps, int n, int H, int W, int r0, int r1, int c0, int c1, void *stream) {
    if (!maps || n < 0 || r0 < 0 || c0 < 0 || r1 > H || c1 > W) return DASS_ERR_ARG;
    if (r1 <= r0 || c1 <= c0) return DASS_OK;
    DASS_LAUNCH(zero_rect_kernel, dim3(dass_grid_1d((long)(r1 - r0) * (c1 - c0), 256)), dim3(256), 0,
                       (hipStream_t)stream, maps, (long)n * H * W, W, r0, r1, c0, c1);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_minmax_blocks(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    return (int)(g < 1 ? 1 : g);
}

extern "C" int dass_minmax(const float *v, int64_t n, float *partial, float *out_min_max, void *stream) {
    if (!v || !partial || !out_min_max || n <= 0) return DASS_ERR_ARG;
    const int blocks = dass_minmax_blocks(n);
    hipStream_t st = (hipStream_t)stream;
    DASS_LAUNCH(minmax_stage1_kernel, dim3(blocks), dim3(256), 0, st, v, (long)n, partial);
    DASS_LAUNCH_CHECK();
    DASS_LAUNCH(minmax_stage2_kernel, dim3(1), dim3(256), 0, st, partial, blocks, out_min_max);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_affine_inplace(float *v, int64_t n, const float *min_max, void *stream) {
    if (!v || !min_max || n <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(affine_kernel, dim3(dass_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, v, (long)n, min_max);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_square_nms(float *maps, int N, int H, int W, int region, int max_picks, float *imax, int *iarg,
                               int *picks, int *count, void *stream) {
    if (!maps || !imax || !iarg || !picks || !count || N <= 0 || H <= 0 || W <= 0 || region <= 0 || max_picks <= 0)
        return DASS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DASS_LAUNCH(image_argmax_kernel, dim3(N), dim3(256), 0, st, maps, (long)H * W, imax, iarg);
    DASS_LAUNCH_CHECK();
    DASS_LAUNCH(square_nms_kernel, dim3(1), dim3(1024), 0, st, maps, N, H, W, region, max_picks, imax, iarg,
                       picks, count);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}
