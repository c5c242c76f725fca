// Batch-norm statistics / finalize / apply (+ReLU|ReLU6, residual add, Dropout2d channel scale) and
// their backward, over NHWC pixel rows.  All HBM-bound streaming kernels: 16 B per lane, a 16-lane
// group covers 64 consecutive channels of one pixel row (256 B), f32 accumulation, f64 finalize.
// Reference sites: every batchnorm(...) + ReLU pair in models/{aspp,decoder}.py and
// models/backbone/{resnet,mobilenet}.py (F.batch_norm training=True/False semantics of torch).
#include "dass_common.h"
#include <cstdlib>

namespace {

constexpr int SLAB = 128;  // pixel rows per partial-statistics row


// partial[slab][0][k] = sum f0, partial[slab][1][k] = sum f1 over the slab's rows.
// MODE 0: f0 = x, f1 = x*x ; MODE 1: f0 = dact, f1 = dact*xhat
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colstat_kernel(const T *__restrict__ x, long ldx, const T *__restrict__ dout,
                                                      long lddo, const T *__restrict__ out, long ldo,
                                                      const float *__restrict__ mean, const float *__restrict__ invstd,
                                                      const float *__restrict__ nc_scale, long M, int K,
                                                      long rows_per_image, int act, float *__restrict__ partial,
                                                      const float *__restrict__ gate_scale = nullptr,
                                                      const float *__restrict__ gate_shift = nullptr,
                                                      double *__restrict__ sums = nullptr,
                                                      const unsigned char *__restrict__ gates = nullptr) {
    __shared__ float red[3][16][64 + 1];
    f32x4 gm4 = {0.f, 0.f, 0.f, 0.f};  // per-channel max |dz| of this thread's rows (MODE 1 with sums)
    const int tid = threadIdx.x;
    const int cx = tid & 15, ry = tid >> 4;
    const int k = blockIdx.y * 64 + cx * 4;
    // rows per block from the grid: SLAB (128) on the partial-row paths (grid.x = dass_stat_rows(M)); the f64-atomic paths launch
    // fewer, longer blocks (stat_blocks) -- one block per 128 rows ended in 1041 atomics per channel on the layer-1 tensors
    const long slab = ((M + gridDim.x - 1) / gridDim.x + 15) / 16 * 16;
    const long r0 = (long)blockIdx.x * slab;
    long r1 = r0 + slab;
    if (r1 > M) r1 = M;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f}, gsc = is, gsh = mu;
        if (MODE == 1) {
            mu = *reinterpret_cast<const f32x4 *>(mean + k);
            is = *reinterpret_cast<const f32x4 *>(invstd + k);
            if (!out && !gates) {
                gsc = *reinterpret_cast<const f32x4 *>(gate_scale + k);
                gsh = *reinterpret_cast<const f32x4 *>(gate_shift + k);
            }
        }
        for (long r = r0 + ry; r < r1; r += 16) {
            const f32x4 xv = ld4<T>(x + r * ldx + k);
            if (MODE == 0) {
                s0 += xv;
                s1 += xv * xv;
            } else {
                f32x4 g = ld4<T>(dout + r * lddo + k);
                if (gates) {
                    // the forward stored the activation gate of every element (4 bits per 4-channel group): 1 byte read
                    // instead of the 16 bytes of `out` (residual layers cannot re-derive the gate from the conv output alone)
                    const unsigned gb = gates[r * (K >> 2) + (k >> 2)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = ((gb >> e) & 1u) ? g[e] : 0.f;
                } else {
                // activation gate from the stored output, or -- no residual, f32 -- re-derived from the conv output with
                // the forward's own fma (bit-identical pre-activation), which saves reading `out`
                const f32x4 o = out ? ld4<T>(out + r * ldo + k) : bn_affine(xv, gsc, gsh);
                if (nc_scale) g *= *reinterpret_cast<const f32x4 *>(nc_scale + (r / rows_per_image) * K + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] *= act_grad_from_out(o[e], act);
                }
                s0 += g;
                s1 += g * ((xv - mu) * is);
#pragma unroll
                for (int e = 0; e < 4; ++e) gm4[e] = fmaxf(gm4[e], fabsf(g[e]));
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][ry][cx * 4 + e] = s0[e];
        red[1][ry][cx * 4 + e] = s1[e];
    }
    if (MODE == 1 && sums) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[2][ry][cx * 4 + e] = gm4[e];
    }
    __syncthreads();
    if (MODE == 1 && sums && tid >= 128 && tid < 192) {
        // per-channel max |dz| -> the K floats behind the 2K f64 sums (an atomic max on the bit patterns of non-negative
        // floats): dass_bn_bwd_apply_sums derives the bound of dx from them when it emits dx in the two-part x3 format
        const int c = tid - 128, kk = blockIdx.y * 64 + c;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a = fmaxf(a, red[2][i][c]);
        if (kk < K) {
            if (!(a >= 0.f)) a = __uint_as_float(0x7f800000u);  // NaN: an infinite bound
            unsigned *slot = reinterpret_cast<unsigned *>(sums + 2 * (long)K) + kk;
            if (__float_as_uint(a) > *reinterpret_cast<volatile unsigned *>(slot)) atomicMax(slot, __float_as_uint(a));  // (a max only grows)
        }
    }
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += red[which][i][c];
        const int kk = blockIdx.y * 64 + c;
        if (kk < K) {
            // sums: [2][K] f64 accumulators shared by all slabs (hardware f64 atomics) -- no partial rows, no finalize launch
            if (sums) unsafeAtomicAdd(sums + (long)which * K + kk, (double)a);
            else partial[((long)blockIdx.x * 2 + which) * K + kk] = a;
        }
    }
}

// f64 reduction of the partial rows: one 1024-thread block per 32 channels, 32 row lanes each
constexpr int FIN_CH = 32, FIN_LANES = 32;
__device__ __forceinline__ bool reduce_partials(const float *__restrict__ partial, int rows, int K, int &k, double &s,
                                                double &ss) {
    __shared__ double red[2][FIN_LANES][FIN_CH + 1];
    const int c = threadIdx.x % FIN_CH, rl = threadIdx.x / FIN_CH;
    k = blockIdx.x * FIN_CH + c;
    double a = 0.0, b = 0.0;
    if (k < K)
        for (int r = rl; r < rows; r += FIN_LANES) {
            a += (double)partial[((long)r * 2 + 0) * K + k];
            b += (double)partial[((long)r * 2 + 1) * K + k];
        }
    red[0][rl][c] = a;
    red[1][rl][c] = b;
    __syncthreads();
    if (rl != 0 || k >= K) return false;
    s = 0.0;
    ss = 0.0;
#pragma unroll 8
    for (int i = 0; i < FIN_LANES; ++i) {
        s += red[0][i][c];
        ss += red[1][i][c];
    }
    return true;
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float *__restrict__ partial, int rows, int K,
                                                           double count, double rep, const float *gamma,
                                                           const float *beta, float *running_mean, float *running_var,
                                                           float momentum, float eps, float *mean, float *invstd,
                                                           float *scale, float *shift) {
    int k;
    double s, ss;
    if (!reduce_partials(partial, rows, K, k, s, ss)) return;
    const double mu = s * rep / count;
    double var = ss * rep / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[k] : 1.f, b = beta ? beta[k] : 0.f;
    mean[k] = (float)mu;
    invstd[k] = (float)is;
    scale[k] = (float)((double)g * is);
    shift[k] = (float)((double)b - mu * (double)g * is);
    if (momentum >= 0.f && running_mean && running_var) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[k] = (1.f - momentum) * running_mean[k] + momentum * (float)mu;
        running_var[k] = (1.f - momentum) * running_var[k] + momentum * (float)unb;
    }
}

__global__ void bn_eval_kernel(const float *gamma, const float *beta, const float *rm, const float *rv, float eps,
                               int K, float *mean, float *invstd, float *scale, float *shift) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float is = 1.f / sqrtf(rv[k] + eps);
    const float g = gamma ? gamma[k] : 1.f, b = beta ? beta[k] : 0.f;
    mean[k] = rm[k];
    invstd[k] = is;
    scale[k] = g * is;
    shift[k] = b - rm[k] * g * is;
}

__global__ __launch_bounds__(1024) void bwd_finalize_kernel(const float *__restrict__ partial, int rows, int K,
                                                            float *dbeta, float *dgamma) {
    int k;
    double s, ss;
    if (!reduce_partials(partial, rows, K, k, s, ss)) return;
    if (dbeta) dbeta[k] = (float)s;
    if (dgamma) dgamma[k] = (float)ss;
}

template <typename T>
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const T *__restrict__ x, long ldx, T *__restrict__ out,
                                                              long ldo, const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              const T *__restrict__ res, long ldr,
                                                              const float *__restrict__ nc_scale, long M, int K,
                                                              long rows_per_image, int act, char *__restrict__ out3, int parts) {
    if (out3 && blockIdx.x == 0) x3_zero_row(out3, M, (K + 31) >> 5, parts);
    const int cc3 = (K + 31) >> 5;
    // (row, 4-channel group) walked incrementally: the grid stride is decomposed once, so the loop has no 64-bit division
    const int kv = K >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long dm = stride / kv;
    const int dk = (int)(stride - dm * kv);
    long m = i0 / kv;
    int kq = (int)(i0 - m * kv);
    for (; m < M; m += dm, kq += dk) {
        if (kq >= kv) {
            kq -= kv;
            if (++m >= M) break;
        }
        const int k = kq << 2;
        f32x4 v = ld4<T>(x + m * ldx + k);
        if (scale && shift) {
            v = bn_affine(v, *reinterpret_cast<const f32x4 *>(scale + k), *reinterpret_cast<const f32x4 *>(shift + k));
        } else {
            if (scale) v *= *reinterpret_cast<const f32x4 *>(scale + k);
            if (shift) v += *reinterpret_cast<const f32x4 *>(shift + k);
        }
        if (res) v += ld4<T>(res + m * ldr + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
        if (nc_scale) v *= *reinterpret_cast<const f32x4 *>(nc_scale + (m / rows_per_image) * K + k);
        if (out) st4<T>(out + m * ldo + k, v);
        if (out3) x3_store4r(out3, m, cc3, k, v, parts, 1.f);  // the same values as three (one) bf16 parts: operand of the next dense conv
    }
}

// Train-mode BatchNorm apply straight from the f64 channel sums the producing conv accumulated (dass_conv2d_igemm_sums /
// dass_conv2d_x3_sums / dass_channel_sums): every block derives the K scale / shift pairs into LDS (f64, as the finalize
// kernel does), block 0 also stores mean / invstd / scale / shift for the backward and updates the running statistics --
// the separate finalize launch of every BN layer is gone.  Loop body = scale_shift_act_kernel.
constexpr int BN_KMAX = 2048;
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_train_kernel(const T *__restrict__ x, long ldx, T *__restrict__ out, long ldo,
                                                             const double *__restrict__ sums, double count, const float *gamma,
                                                             const float *beta, float *running_mean, float *running_var,
                                                             float momentum, float eps, float *mean, float *invstd, float *scale_o,
                                                             float *shift_o, const T *__restrict__ res, long ldr,
                                                             const float *__restrict__ nc_scale, long M, int K, long rows_per_image,
                                                             int act, char *__restrict__ out3, unsigned char *__restrict__ gates,
                                                             int parts, const float *__restrict__ res_bound) {
    __shared__ __attribute__((aligned(16))) float s_scale[BN_KMAX];
    __shared__ __attribute__((aligned(16))) float s_shift[BN_KMAX];
    __shared__ float s_bnd[2][4];
    float gmax = 0.f;  // max over channels of |gamma| sqrt(M - 1) + |beta|: bounds |gamma xhat + beta| (two-part x3 output)
    const float sqm = sqrtf(count > 2.0 ? (float)(count - 1.0) : 1.f);
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const double mu = sums[k] / count;
        double var = sums[K + k] / count - mu * mu;
        if (var < 0.0) var = 0.0;
        const double is = 1.0 / sqrt(var + (double)eps);
        const float g = gamma ? gamma[k] : 1.f, b = beta ? beta[k] : 0.f;
        const float sc = (float)((double)g * is), sh = (float)((double)b - mu * (double)g * is);
        s_scale[k] = sc;
        s_shift[k] = sh;
        gmax = fmaxf(gmax, fabsf(g) * sqm + fabsf(b));
        if (blockIdx.x == 0) {
            mean[k] = (float)mu;
            invstd[k] = (float)is;
            scale_o[k] = sc;
            shift_o[k] = sh;
            if (momentum >= 0.f && running_mean && running_var) {
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                running_mean[k] = (1.f - momentum) * running_mean[k] + momentum * (float)mu;
                running_var[k] = (1.f - momentum) * running_var[k] + momentum * (float)unb;
            }
        }
    }
    float x3s = 1.f;
    if (out3 && parts == 2) {
        // Two-part x3 output: the per-tensor scale needs a bound of |out| BEFORE any element is written.  Batch statistics give
        // one for free (Samuelson): |x - mean| <= std * sqrt(M - 1) for every sample, so per channel |gamma * xhat + beta| <=
        // |gamma| * sqrt(M - 1) + |beta|; a residual adds its own bound, ReLU6 caps at 6, a Dropout2d mask multiplies by its largest entry.
        // Loose by ~sqrt(M) / (the batch's true max |xhat|, ~5): 2^4 .. 2^6 -- harmless (dass_common.h), and the same in every block.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
        // the Dropout2d multipliers' true maximum (1 / (1 - p), any p): [images][K] floats, a few KB from L2 per block
        float nmax = 0.f;
        if (nc_scale) {
            const long nimg = (M + rows_per_image - 1) / rows_per_image;
            for (long i = threadIdx.x; i < nimg * K; i += blockDim.x) nmax = fmaxf(nmax, fabsf(nc_scale[i]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nmax = fmaxf(nmax, __shfl_xor(nmax, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            s_bnd[0][threadIdx.x >> 6] = gmax;
            s_bnd[1][threadIdx.x >> 6] = nmax;
        }
    }
    __syncthreads();
    if (out3 && parts == 2) {
        const float gm = fmaxf(fmaxf(s_bnd[0][0], s_bnd[0][1]), fmaxf(s_bnd[0][2], s_bnd[0][3]));
        float bound = 1.25f * gm + (res_bound ? *res_bound : 0.f);
        if (act == DASS_ACT_RELU6) bound = fminf(bound, 6.f);
        if (nc_scale) bound *= fmaxf(fmaxf(s_bnd[1][0], s_bnd[1][1]), fmaxf(s_bnd[1][2], s_bnd[1][3]));
        x3s = x3_scale_of(bound);
        if (blockIdx.x == 0) x3_zero_row(out3, M, (K + 31) >> 5, 2, x3_inv_of(x3s), bound);
    } else if (out3 && blockIdx.x == 0) {
        x3_zero_row(out3, M, (K + 31) >> 5, parts);
    }
    const int cc3 = (K + 31) >> 5;
    const int kv = K >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long dm = stride / kv;
    const int dk = (int)(stride - dm * kv);
    long m = i0 / kv;
    int kq = (int)(i0 - m * kv);
    for (; m < M; m += dm, kq += dk) {
        if (kq >= kv) {
            kq -= kv;
            if (++m >= M) break;
        }
        const int k = kq << 2;
        f32x4 v = bn_affine(ld4<T>(x + m * ldx + k), *reinterpret_cast<const f32x4 *>(s_scale + k), *reinterpret_cast<const f32x4 *>(s_shift + k));
        if (res) v += ld4<T>(res + m * ldr + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
        if (gates) {  // the backward's activation gate of these four outputs (act_grad_from_out of the stored value)
            unsigned gb = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) gb |= (act_grad_from_out(v[e], act) != 0.f ? 1u : 0u) << e;
            gates[m * (K >> 2) + kq] = (unsigned char)gb;
        }
        if (nc_scale) v *= *reinterpret_cast<const f32x4 *>(nc_scale + (m / rows_per_image) * K + k);
        if (out) st4<T>(out + m * ldo + k, v);
        if (out3) x3_store4r(out3, m, cc3, k, v, parts, x3s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T *__restrict__ dout, long lddo,
                                                           const T *__restrict__ out, long ldo,
                                                           const T *__restrict__ x, long ldx,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ dbeta,
                                                           const float *__restrict__ dgamma,
                                                           const float *__restrict__ nc_scale, T *__restrict__ dx,
                                                           long lddx, T *__restrict__ dres, long lddr, long M, int K,
                                                           long rows_per_image, float inv_count, int train, int act,
                                                           const float *__restrict__ gate_scale = nullptr,
                                                           const float *__restrict__ gate_shift = nullptr,
                                                           char *__restrict__ dx3 = nullptr,
                                                           const double *__restrict__ sums = nullptr,
                                                           float *__restrict__ dbeta_out = nullptr,
                                                           float *__restrict__ dgamma_out = nullptr,
                                                           const unsigned char *__restrict__ gates = nullptr, int parts = 3) {
    float x3s = 1.f;
    if (dx3 && parts == 2) {
        // Two-part x3 form of dx: per channel |dx| = |gamma invstd| |dz - mean(dz) - xhat mean(dz xhat)|
        //   <= |gamma invstd| (max|dz| + |mean(dz)| + sqrt(M - 1) |mean(dz xhat)|)      (Samuelson: |xhat| <= sqrt(M - 1)),
        // with the two means from the f64 sums and the channel's max |dz| left behind them by dass_bn_bwd_reduce_sums; the
        // tensor's bound is the max over the channels.  Channel-wise on purpose: a near-constant channel has a huge invstd and
        // a tiny dz, a product of the two global maxima would be loose by many binades (seen: 4x the gradient error).
        __shared__ float s_g[4];
        const float *dzmax = reinterpret_cast<const float *>(sums + 2 * (long)K);
        const float sqm = sqrtf(M > 1 ? (float)(M - 1) : 1.f);
        float gi = 0.f;
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            float b = dzmax[k];
            if (train) b += fabsf((float)sums[k]) * inv_count + sqm * fabsf((float)sums[K + k]) * inv_count;
            gi = fmaxf(gi, fabsf((gamma ? gamma[k] : 1.f) * invstd[k]) * b);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gi = fmaxf(gi, __shfl_xor(gi, o, 64));
        if ((threadIdx.x & 63) == 0) s_g[threadIdx.x >> 6] = gi;
        __syncthreads();
        const float bound = 1.25f * fmaxf(fmaxf(s_g[0], s_g[1]), fmaxf(s_g[2], s_g[3]));
        x3s = x3_scale_of(bound);
        if (blockIdx.x == 0) x3_zero_row(dx3, M, (K + 31) >> 5, 2, x3_inv_of(x3s), bound);
    } else if (dx3 && blockIdx.x == 0) {
        x3_zero_row(dx3, M, (K + 31) >> 5, parts);
    }
    if (sums && blockIdx.x == 0)  // the f64 sums of dass_bn_bwd_reduce_sums, rounded once: the parameter gradients
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            if (dbeta_out) dbeta_out[k] = (float)sums[k];
            if (dgamma_out) dgamma_out[k] = (float)sums[K + k];
        }
    const int cc3 = (K + 31) >> 5;
    const int kv = K >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long dm = stride / kv;
    const int dk = (int)(stride - dm * kv);
    long m = i0 / kv;
    int kq = (int)(i0 - m * kv);
    if (dk == 0 && (dx || dx3)) {
        // The host sized the grid so that the thread count is a multiple of K / 4: this thread keeps ONE 4-channel group for all its
        // rows, and the seven per-channel vectors (mean, invstd, gamma, the two sums, the gate's scale / shift) are loaded once into
        // registers instead of once per 16 bytes of gradient -- the loop then issues only the loads of the tensors it streams
        // (round 3: 10 vector loads per iteration, 3 of them streams; 3.5 TB/s against the forward pass's 4.3).  Same arithmetic.
        const int k = kq << 2;
        const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + k);
        f32x4 ga = {1.f, 1.f, 1.f, 1.f}, mu = {0.f, 0.f, 0.f, 0.f}, db = mu, dg = mu, gsc = mu, gsh = mu;
        if (gamma) ga = *reinterpret_cast<const f32x4 *>(gamma + k);
        if (train) {
            mu = *reinterpret_cast<const f32x4 *>(mean + k);
            if (sums) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    db[e] = (float)sums[k + e];
                    dg[e] = (float)sums[K + k + e];
                }
            } else {
                db = *reinterpret_cast<const f32x4 *>(dbeta + k);
                dg = *reinterpret_cast<const f32x4 *>(dgamma + k);
            }
        }
        if (!gates && !out) {
            gsc = *reinterpret_cast<const f32x4 *>(gate_scale + k);
            gsh = *reinterpret_cast<const f32x4 *>(gate_shift + k);
        }
        const f32x4 gis = ga * is;
        const bool need_x = (!out && !gates) || train;
        for (; m < M; m += dm) {
            f32x4 g = ld4<T>(dout + m * lddo + k);
            f32x4 xin = {0.f, 0.f, 0.f, 0.f};
            if (need_x) xin = ld4<T>(x + m * ldx + k);
            if (gates) {
                const unsigned gb = gates[m * (K >> 2) + kq];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = ((gb >> e) & 1u) ? g[e] : 0.f;
            } else {
                const f32x4 o = out ? ld4<T>(out + m * ldo + k) : bn_affine(xin, gsc, gsh);
                if (nc_scale) g *= *reinterpret_cast<const f32x4 *>(nc_scale + (m / rows_per_image) * K + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] *= act_grad_from_out(o[e], act);
            }
            if (dres) st4<T>(dres + m * lddr + k, g);
            f32x4 r = g;
            if (train) {
                const f32x4 xh = (xin - mu) * is;
                r = g - (db + xh * dg) * inv_count;
            }
            const f32x4 dxv = r * gis;
            if (dx) st4<T>(dx + m * lddx + k, dxv);
            if (dx3) x3_store4r(dx3, m, cc3, k, dxv, parts, x3s);
        }
        return;
    }
    for (; m < M; m += dm, kq += dk) {
        if (kq >= kv) {
            kq -= kv;
            if (++m >= M) break;
        }
        const int k = kq << 2;
        f32x4 g = ld4<T>(dout + m * lddo + k);
        f32x4 xin = {0.f, 0.f, 0.f, 0.f};
        if ((!out && !gates) || ((dx || dx3) && train)) xin = ld4<T>(x + m * ldx + k);
        if (gates) {
            const unsigned gb = gates[m * (K >> 2) + kq];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = ((gb >> e) & 1u) ? g[e] : 0.f;
        } else {
        const f32x4 o = out ? ld4<T>(out + m * ldo + k)
                            : bn_affine(xin, *reinterpret_cast<const f32x4 *>(gate_scale + k), *reinterpret_cast<const f32x4 *>(gate_shift + k));
        if (nc_scale) g *= *reinterpret_cast<const f32x4 *>(nc_scale + (m / rows_per_image) * K + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] *= act_grad_from_out(o[e], act);
        }
        if (dres) st4<T>(dres + m * lddr + k, g);
        if (dx || dx3) {  // (dx == NULL: only the split rows are consumed -- pre-split input- and weight-gradient launches)
            const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + k);
            f32x4 ga = {1.f, 1.f, 1.f, 1.f};
            if (gamma) ga = *reinterpret_cast<const f32x4 *>(gamma + k);
            f32x4 r = g;
            if (train) {
                const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + k);
                const f32x4 xh = (xin - mu) * is;
                f32x4 db, dg;
                if (sums) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        db[e] = (float)sums[k + e];
                        dg[e] = (float)sums[K + k + e];
                    }
                } else {
                    db = *reinterpret_cast<const f32x4 *>(dbeta + k);
                    dg = *reinterpret_cast<const f32x4 *>(dgamma + k);
                }
                r = g - (db + xh * dg) * inv_count;
            }
            const f32x4 dxv = r * (ga * is);
            if (dx) st4<T>(dx + m * lddx + k, dxv);
            if (dx3) x3_store4r(dx3, m, cc3, k, dxv, parts, x3s);  // operand of the producing conv's input- and weight-gradient launches
        }
    }
}

// ---- The train step's BN-backward apply, specialised (round 4): f32, batch statistics through the f64 sums, the activation gate
// from stored gate bits (GATES) or re-derived from the conv output, split rows out (dx3, two- or three-part), optionally the f32
// rows too (DX32) and the residual branch's gradient (DRES).  Everything bn_bwd_apply_kernel can do besides is compiled out: that
// kernel needs 93-131 VGPRs for its options (3-5 waves per SIMD); this one keeps ONE 4-channel group per thread (the host sizes
// the grid: bn_grid), its seven per-channel vectors in registers, and issues the loads of two rows before it touches either.
// Same arithmetic, same rounding as the general kernel (tests/test_round4_gpu.py compares the two bit for bit).
template <bool GATES, bool DRES, bool DX32>
__global__ __launch_bounds__(256, 5) void bn_bwd_fast_kernel(const float *__restrict__ dout, long lddo, const float *__restrict__ x, long ldx,
                                                          const float *__restrict__ mean, const float *__restrict__ invstd,
                                                          const float *__restrict__ gamma, const double *__restrict__ sums,
                                                          float *__restrict__ dbeta_out, float *__restrict__ dgamma_out,
                                                          const float *__restrict__ gate_scale, const float *__restrict__ gate_shift,
                                                          float *__restrict__ dx, long lddx, float *__restrict__ dres, long lddr, long M, int K,
                                                          float inv_count, int act, const unsigned char *__restrict__ gates,
                                                          char *__restrict__ dx3, int parts) {
    float x3s = 1.f;
    if (parts == 2) {  // the tensor's bound for the two-part rows (see bn_bwd_apply_kernel)
        __shared__ float s_g[4];
        const float *dzmax = reinterpret_cast<const float *>(sums + 2 * (long)K);
        const float sqm = sqrtf(M > 1 ? (float)(M - 1) : 1.f);
        float gi = 0.f;
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            const float b = dzmax[k] + fabsf((float)sums[k]) * inv_count + sqm * fabsf((float)sums[K + k]) * inv_count;
            gi = fmaxf(gi, fabsf((gamma ? gamma[k] : 1.f) * invstd[k]) * b);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gi = fmaxf(gi, __shfl_xor(gi, o, 64));
        if ((threadIdx.x & 63) == 0) s_g[threadIdx.x >> 6] = gi;
        __syncthreads();
        const float bound = 1.25f * fmaxf(fmaxf(s_g[0], s_g[1]), fmaxf(s_g[2], s_g[3]));
        x3s = x3_scale_of(bound);
        if (blockIdx.x == 0) x3_zero_row(dx3, M, (K + 31) >> 5, 2, x3_inv_of(x3s), bound);
    } else if (blockIdx.x == 0) {
        x3_zero_row(dx3, M, (K + 31) >> 5, parts);
    }
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            if (dbeta_out) dbeta_out[k] = (float)sums[k];
            if (dgamma_out) dgamma_out[k] = (float)sums[K + k];
        }
    const int cc3 = (K + 31) >> 5, kv = K >> 2;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long dm = ((long)gridDim.x * blockDim.x) / kv;   // (host: the thread count is a multiple of kv)
    long m = i0 / kv;
    const int kq = (int)(i0 - m * kv), k = kq << 2;
    const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + k), mu = *reinterpret_cast<const f32x4 *>(mean + k);
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, db, dg, gsc = {0.f, 0.f, 0.f, 0.f}, gsh = gsc;
    if (gamma) ga = *reinterpret_cast<const f32x4 *>(gamma + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        db[e] = (float)sums[k + e];
        dg[e] = (float)sums[K + k + e];
    }
    if constexpr (!GATES) {
        gsc = *reinterpret_cast<const f32x4 *>(gate_scale + k);
        gsh = *reinterpret_cast<const f32x4 *>(gate_shift + k);
    }
    const f32x4 gis = ga * is;
    auto finish = [&](long mm, f32x4 g, const f32x4 xin, const unsigned gb) __attribute__((always_inline)) {
        if constexpr (GATES) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = ((gb >> e) & 1u) ? g[e] : 0.f;
        } else {
            const f32x4 o = bn_affine(xin, gsc, gsh);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] *= act_grad_from_out(o[e], act);
        }
        if constexpr (DRES) *reinterpret_cast<f32x4 *>(dres + mm * lddr + k) = g;
        const f32x4 xh = (xin - mu) * is;
        const f32x4 r = g - (db + xh * dg) * inv_count;
        const f32x4 dxv = r * gis;
        if constexpr (DX32) *reinterpret_cast<f32x4 *>(dx + mm * lddx + k) = dxv;
        x3_store4r(dx3, mm, cc3, k, dxv, parts, x3s);
    };
    for (; m + dm < M; m += 2 * dm) {
        const long m1 = m + dm;
        const f32x4 g0 = *reinterpret_cast<const f32x4 *>(dout + m * lddo + k), g1 = *reinterpret_cast<const f32x4 *>(dout + m1 * lddo + k);
        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(x + m * ldx + k), x1 = *reinterpret_cast<const f32x4 *>(x + m1 * ldx + k);
        unsigned b0 = 0u, b1 = 0u;
        if constexpr (GATES) {
            b0 = gates[m * kv + kq];
            b1 = gates[m1 * kv + kq];
        }
        finish(m, g0, x0, b0);
        finish(m1, g1, x1, b1);
    }
    if (m < M) {
        unsigned b0 = 0u;
        if constexpr (GATES) b0 = gates[m * kv + kq];
        finish(m, *reinterpret_cast<const f32x4 *>(dout + m * lddo + k), *reinterpret_cast<const f32x4 *>(x + m * ldx + k), b0);
    }
}

// ---- BatchNorm over a handful of ROWS (the ASPP image-pool branch: BN of an [N, C] vector that the reference broadcasts to
// H x W first, aspp.py:62-65,79-81).  The input is post-ReLU (mean >> deviation) and N is the batch size, so the batch
// variance is a difference of nearly equal numbers and, for N = 2, the input gradient vanishes up to eps: sum / sum-of-squares
// partials in f32 lose three digits there (seen against the f64 oracle).  One thread per channel, two-pass, all in f64.
__global__ void bn_rows_fwd_kernel(const float *__restrict__ x, int N, int K, double rep, const float *gamma, const float *beta,
                                   float *running_mean, float *running_var, float momentum, float eps, float *mean,
                                   float *invstd, float *scale, float *shift) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double s = 0.0;
    for (int n = 0; n < N; ++n) s += (double)x[(long)n * K + k];
    const double mu = s / N;
    double v = 0.0;
    for (int n = 0; n < N; ++n) {
        const double d = (double)x[(long)n * K + k] - mu;
        v += d * d;
    }
    const double var = v / N, count = (double)N * rep;  // every row stands for `rep` identical elements
    const double is = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[k] : 1.f, b = beta ? beta[k] : 0.f;
    mean[k] = (float)mu;
    invstd[k] = (float)is;
    scale[k] = (float)((double)g * is);
    shift[k] = (float)((double)b - mu * (double)g * is);
    if (momentum >= 0.f && running_mean && running_var) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[k] = (1.f - momentum) * running_mean[k] + momentum * (float)mu;
        running_var[k] = (1.f - momentum) * running_var[k] + momentum * (float)unb;
    }
}

// backward of the same: g [N][K] already summed over the broadcast copies; dx = gamma * invstd * (g - (dbeta + xhat * dgamma) / N)
__global__ void bn_rows_bwd_kernel(const float *__restrict__ g, const float *__restrict__ x, const float *__restrict__ mean,
                                   const float *__restrict__ invstd, const float *__restrict__ gamma, int N, int K, int train,
                                   float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    // batch statistics: the mean is recomputed in f64 from the rows (x - mean cancels; the stored f32 mean is rounded);
    // invstd only multiplies, its f32 rounding is harmless
    double mu = (double)mean[k];
    const double is = (double)invstd[k];
    if (train) {
        double s = 0.0;
        for (int n = 0; n < N; ++n) s += (double)x[(long)n * K + k];
        mu = s / N;
    }
    double db = 0.0, dg = 0.0;
    for (int n = 0; n < N; ++n) {
        const double gv = (double)g[(long)n * K + k];
        db += gv;
        dg += gv * ((double)x[(long)n * K + k] - mu) * is;
    }
    dbeta[k] = (float)db;
    dgamma[k] = (float)dg;
    const double ga = gamma ? (double)gamma[k] : 1.0;
    for (int n = 0; n < N; ++n) {
        const double gv = (double)g[(long)n * K + k];
        double r = gv;
        if (train) r = gv - (db + ((double)x[(long)n * K + k] - mu) * is * dg) / N;
        dx[(long)n * K + k] = (float)(r * ga * is);
    }
}

// 1-D grid for the BN backward apply kernel over M rows of K channels (256 threads, one thread per 4 channels): when it can, a
// thread count that is a multiple of K / 4, so that every thread keeps one channel group for all its rows (bn_bwd_apply_kernel)
static inline int bn_grid(int64_t M, int K, int rows_per_thread = 1) {
    const int64_t kv = K / 4;
    int64_t g = dass_grid_1d((M * kv + rows_per_thread - 1) / rows_per_thread, 256);
    if (kv <= 0) return (int)g;
    int64_t a = kv, b = 256;
    while (b) { const int64_t t = a % b; a = b; b = t; }
    const int64_t g0 = kv / a;            // blocks per whole number of rows
    if (g0 <= g) g = g / g0 * g0;
    return (int)g;
}

bool ok4(int K, int64_t a, int64_t b = 4, int64_t c = 4, int64_t d = 4) {
    return K > 0 && K % 4 == 0 && a % 4 == 0 && b % 4 == 0 && c % 4 == 0 && d % 4 == 0;
}

}  // namespace

extern "C" int dass_stat_rows(int64_t M) { return (int)((M + SLAB - 1) / SLAB); }
// row blocks of a colstat launch that ends in f64 atomics: ~1024 blocks in all, at least 128 rows each
static unsigned stat_blocks(long M, int K) {
    const long cols = (K + 63) / 64;
    static const long target = getenv("DASS_STAT_BLOCKS") ? atol(getenv("DASS_STAT_BLOCKS")) : 1024;  // (tuning knob: 512 / 1024 / 2048 / 4096 measured 0.83 / 0.79 / 0.93 / 1.07 ms per R101 step)
    long want = (target + cols - 1) / cols;
    const long most = (M + SLAB - 1) / SLAB;
    if (want > most) want = most;
    return (unsigned)(want < 1 ? 1 : want);
}

extern "C" int dass_channel_stats(const void *x, int64_t ldx, int64_t M, int K, float *partial, int dtype,
                                  void *stream) {
    if (!x || !partial || M <= 0 || !ok4(K, ldx)) return DASS_ERR_ARG;
    dim3 grid((unsigned)dass_stat_rows(M), (unsigned)((K + 63) / 64));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32)
        DASS_LAUNCH((colstat_kernel<float, 0>), grid, dim3(256), 0, st, (const float *)x, ldx, nullptr, 0,
                           nullptr, 0, nullptr, nullptr, nullptr, M, K, 1, 0, partial, nullptr, nullptr, nullptr, nullptr);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH((colstat_kernel<bf16_t, 0>), grid, dim3(256), 0, st, (const bf16_t *)x, ldx, nullptr, 0,
                           nullptr, 0, nullptr, nullptr, nullptr, M, K, 1, 0, partial, nullptr, nullptr, nullptr, nullptr);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_finalize(const float *partial, int rows, int K, double count, double rep, const float *gamma,
                                const float *beta, float *running_mean, float *running_var, float momentum,
                                float eps, float *mean, float *invstd, float *scale, float *shift, void *stream) {
    if (!partial || rows <= 0 || K <= 0 || count <= 0 || !mean || !invstd || !scale || !shift) return DASS_ERR_ARG;
    DASS_LAUNCH(bn_finalize_kernel, dim3((K + FIN_CH - 1) / FIN_CH), dim3(1024), 0, (hipStream_t)stream, partial, rows, K,
                       count, rep, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_rows_fwd(const float *x, int N, int K, double rep, const float *gamma, const float *beta, float *running_mean,
                                float *running_var, float momentum, float eps, float *mean, float *invstd, float *scale, float *shift,
                                void *stream) {
    if (!x || N <= 0 || K <= 0 || rep <= 0 || !mean || !invstd || !scale || !shift) return DASS_ERR_ARG;
    DASS_LAUNCH(bn_rows_fwd_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, N, K, rep, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_rows_bwd(const float *g, const float *x, const float *mean, const float *invstd, const float *gamma, int N, int K,
                                int train, float *dx, float *dgamma, float *dbeta, void *stream) {
    if (!g || !x || !mean || !invstd || !dx || !dgamma || !dbeta || N <= 0 || K <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(bn_rows_bwd_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, x, mean, invstd, gamma, N, K,
                       train, dx, dgamma, dbeta);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_eval_scale_shift(const float *gamma, const float *beta, const float *running_mean,
                                        const float *running_var, float eps, int K, float *mean, float *invstd,
                                        float *scale, float *shift, void *stream) {
    if (!running_mean || !running_var || K <= 0 || !mean || !invstd || !scale || !shift) return DASS_ERR_ARG;
    DASS_LAUNCH(bn_eval_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, K, mean, invstd, scale, shift);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_scale_shift_act(const void *x, int64_t ldx, void *out, int64_t ldo, const float *scale,
                                    const float *shift, const void *residual, int64_t ldr, const float *nc_scale,
                                    int64_t M, int K, int64_t rows_per_image, int act, int dtype, void *out3, void *stream) {
    if (!x || (!out && !out3) || M <= 0 || !ok4(K, ldx, out ? ldo : 4, residual ? ldr : 4) || rows_per_image <= 0) return DASS_ERR_ARG;
    if (out3 && (dtype != DASS_F32 || ((uintptr_t)out3 & 15))) return DASS_ERR_ARG;
    if (out3 && dass_get_x3_parts() == 2) return DASS_ERR_UNSUPPORTED;  // no bound of x here: the consumer converts (dass_split3_rows)
    const int grid = dass_grid_1d(M * (K / 4), 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32)
        DASS_LAUNCH(scale_shift_act_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, ldx,
                           (float *)out, ldo, scale, shift, (const float *)residual, ldr, nc_scale, M, K,
                           rows_per_image, act, (char *)out3, dass_get_x3_parts());
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(scale_shift_act_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)x, ldx,
                           (bf16_t *)out, ldo, scale, shift, (const bf16_t *)residual, ldr, nc_scale, M, K,
                           rows_per_image, act, (char *)nullptr, 3);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_bwd_reduce(const void *dout, int64_t lddo, const void *out, int64_t ldo, const void *x,
                                  int64_t ldx, const float *mean, const float *invstd, const float *nc_scale,
                                  int64_t M, int K, int64_t rows_per_image, int act, float *partial, int dtype,
                                  void *stream) {
    if (!dout || !out || !x || !mean || !invstd || !partial || M <= 0 || !ok4(K, lddo, ldo, ldx) ||
        rows_per_image <= 0)
        return DASS_ERR_ARG;
    dim3 grid((unsigned)dass_stat_rows(M), (unsigned)((K + 63) / 64));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32)
        DASS_LAUNCH((colstat_kernel<float, 1>), grid, dim3(256), 0, st, (const float *)x, ldx,
                           (const float *)dout, lddo, (const float *)out, ldo, mean, invstd, nc_scale, M, K,
                           rows_per_image, act, partial, nullptr, nullptr, nullptr, nullptr);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH((colstat_kernel<bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t *)x, ldx,
                           (const bf16_t *)dout, lddo, (const bf16_t *)out, ldo, mean, invstd, nc_scale, M, K,
                           rows_per_image, act, partial, nullptr, nullptr, nullptr, nullptr);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_bwd_reduce_gate(const void *dout, int64_t lddo, const void *x, int64_t ldx, const float *mean,
                                       const float *invstd, const float *gate_scale, const float *gate_shift,
                                       const float *nc_scale, int64_t M, int K, int64_t rows_per_image, int act, float *partial,
                                       int dtype, void *stream) {
    if (!dout || !x || !mean || !invstd || !gate_scale || !gate_shift || !partial || M <= 0 || !ok4(K, lddo, ldx) || rows_per_image <= 0)
        return DASS_ERR_ARG;
    if (dtype != DASS_F32) return DASS_ERR_UNSUPPORTED;  // a bf16 `out` is rounded: its gate is not the f32 pre-activation's
    dim3 grid((unsigned)dass_stat_rows(M), (unsigned)((K + 63) / 64));
    DASS_LAUNCH((colstat_kernel<float, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const float *)x, ldx, (const float *)dout, lddo,
                       (const float *)nullptr, 0, mean, invstd, nc_scale, M, K, rows_per_image, act, partial, gate_scale, gate_shift, nullptr, nullptr);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_bwd_apply_gate(const void *dout, int64_t lddo, const void *x, int64_t ldx, const float *mean, const float *invstd,
                                      const float *gamma, const float *dbeta, const float *dgamma, const float *gate_scale,
                                      const float *gate_shift, const float *nc_scale, void *dx, int64_t lddx, int64_t M, int K,
                                      int64_t rows_per_image, double count, int train, int act, int dtype, void *dx3, void *stream) {
    if (dx3 && ((uintptr_t)dx3 & 15)) return DASS_ERR_ARG;
    if (dx3 && dass_get_x3_parts() == 2) return DASS_ERR_UNSUPPORTED;  // the two-part form needs max|dz| (the sums path supplies it)
    if (!dout || !x || !dx || !invstd || !gate_scale || !gate_shift || M <= 0 || !ok4(K, lddo, ldx, lddx) || rows_per_image <= 0)
        return DASS_ERR_ARG;
    if (train && (!mean || !dbeta || !dgamma || count <= 0)) return DASS_ERR_ARG;
    if (dtype != DASS_F32) return DASS_ERR_UNSUPPORTED;
    const int grid = bn_grid(M, K);
    const float inv_count = train ? (float)(1.0 / count) : 0.f;
    DASS_LAUNCH(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float *)dout, lddo, (const float *)nullptr, 0,
                       (const float *)x, ldx, mean, invstd, gamma, dbeta, dgamma, nc_scale, (float *)dx, lddx, (float *)nullptr, 0, M, K,
                       rows_per_image, inv_count, train, act, gate_scale, gate_shift, (char *)dx3, nullptr, nullptr, nullptr, nullptr, 3);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_bwd_finalize(const float *partial, int rows, int K, float *dbeta, float *dgamma,
                                    void *stream) {
    if (!partial || rows <= 0 || K <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(bwd_finalize_kernel, dim3((K + FIN_CH - 1) / FIN_CH), dim3(1024), 0, (hipStream_t)stream, partial, rows, K,
                       dbeta, dgamma);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_bwd_apply(const void *dout, int64_t lddo, const void *out, int64_t ldo, const void *x,
                                 int64_t ldx, const float *mean, const float *invstd, const float *gamma,
                                 const float *dbeta, const float *dgamma, const float *nc_scale, void *dx,
                                 int64_t lddx, void *dres, int64_t lddr, int64_t M, int K, int64_t rows_per_image,
                                 double count, int train, int act, int dtype, void *dx3, void *stream) {
    if (!dout || !out || M <= 0 || !ok4(K, lddo, ldo) || rows_per_image <= 0) return DASS_ERR_ARG;
    if (dx3 && (!dx || dtype != DASS_F32 || ((uintptr_t)dx3 & 15))) return DASS_ERR_ARG;
    if (dx3 && dass_get_x3_parts() == 2) return DASS_ERR_UNSUPPORTED;
    if (dx && (!invstd || lddx % 4)) return DASS_ERR_ARG;
    if (dx && train && (!x || !mean || !dbeta || !dgamma || ldx % 4 || count <= 0)) return DASS_ERR_ARG;
    if (dres && lddr % 4) return DASS_ERR_ARG;
    const int grid = bn_grid(M, K);
    const float inv_count = train ? (float)(1.0 / count) : 0.f;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32)
        DASS_LAUNCH(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)dout, lddo,
                           (const float *)out, ldo, (const float *)x, ldx, mean, invstd, gamma, dbeta, dgamma,
                           nc_scale, (float *)dx, lddx, (float *)dres, lddr, M, K, rows_per_image, inv_count, train,
                           act, (const float *)nullptr, (const float *)nullptr, (char *)dx3, nullptr, nullptr, nullptr, nullptr, 3);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)dout, lddo,
                           (const bf16_t *)out, ldo, (const bf16_t *)x, ldx, mean, invstd, gamma, dbeta, dgamma,
                           nc_scale, (bf16_t *)dx, lddx, (bf16_t *)dres, lddr, M, K, rows_per_image, inv_count, train,
                           act, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 3);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_colsum(const void *x, int64_t ldx, int64_t M, int K, float *partial, float *out, int dtype,
                           void *stream) {
    if (!out) return DASS_ERR_ARG;
    const int rc = dass_channel_stats(x, ldx, M, K, partial, dtype, stream);
    if (rc != DASS_OK) return rc;
    DASS_LAUNCH(bwd_finalize_kernel, dim3((K + FIN_CH - 1) / FIN_CH), dim3(1024), 0, (hipStream_t)stream, partial,
                       dass_stat_rows(M), K, out, nullptr);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

// ---- train-mode BN without finalize launches: the statistics live in [2][K] f64 accumulators (zeroed by the caller) that
// the producing kernels add to with hardware f64 atomics; the apply kernels read them directly.  Same arithmetic as the
// partial-row path (f32 sums per tile / slab, f64 across them); only the order of the f64 additions is not fixed --
// dass_set_deterministic callers keep the partial-row entry points.
extern "C" int dass_channel_sums(const void *x, int64_t ldx, int64_t M, int K, double *sums, int dtype, void *stream) {
    if (!x || !sums || M <= 0 || !ok4(K, ldx)) return DASS_ERR_ARG;
    dim3 grid(stat_blocks(M, K), (unsigned)((K + 63) / 64));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32)
        DASS_LAUNCH((colstat_kernel<float, 0>), grid, dim3(256), 0, st, (const float *)x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr,
                           nullptr, M, K, 1, 0, nullptr, nullptr, nullptr, sums, nullptr);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH((colstat_kernel<bf16_t, 0>), grid, dim3(256), 0, st, (const bf16_t *)x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr,
                           nullptr, M, K, 1, 0, nullptr, nullptr, nullptr, sums, nullptr);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_bn_apply_train(const void *x, int64_t ldx, void *out, int64_t ldo, const double *sums, double count, const float *gamma,
                                   const float *beta, float *running_mean, float *running_var, float momentum, float eps, float *mean,
                                   float *invstd, float *scale, float *shift, const void *residual, int64_t ldr, const float *nc_scale,
                                   int64_t M, int K, int64_t rows_per_image, int act, int dtype, void *out3, void *gates, int64_t gates_bytes,
                                   const float *residual_bound, void *stream) {
    if (!x || (!out && !out3) || !sums || count <= 0 || !mean || !invstd || !scale || !shift || M <= 0 ||
        !ok4(K, ldx, out ? ldo : 4, residual ? ldr : 4) || rows_per_image <= 0)
        return DASS_ERR_ARG;
    if (K > BN_KMAX) return DASS_ERR_UNSUPPORTED;
    if (out3 && (dtype != DASS_F32 || ((uintptr_t)out3 & 15))) return DASS_ERR_ARG;
    if (gates && gates_bytes < M * (K / 4)) return DASS_ERR_ARG;  // one byte per 4-channel group of every row
    const int parts = dass_get_x3_parts();
    if (out3 && parts == 2 && residual && !residual_bound) return DASS_ERR_ARG;  // the output's bound needs the residual's
    // (one vector per thread up to the 2048-block cap: fewer, longer blocks -- tried at 8 / 16 vectors per thread to amortise the per-block
    //  scale / shift prologue -- ran this kernel 7 % / 29 % SLOWER: its row loop keeps one load in flight per thread)
    const int grid = dass_grid_1d(M * (K / 4), 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32)
        DASS_LAUNCH(bn_apply_train_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, ldx, (float *)out, ldo, sums, count,
                           gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, (const float *)residual, ldr,
                           nc_scale, M, K, rows_per_image, act, (char *)out3, (unsigned char *)gates, parts, residual_bound);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(bn_apply_train_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)x, ldx, (bf16_t *)out, ldo, sums, count,
                           gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, (const bf16_t *)residual, ldr,
                           nc_scale, M, K, rows_per_image, act, (char *)nullptr, (unsigned char *)gates, 3, (const float *)nullptr);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

// out == NULL: the activation gate is re-derived from x with gate_scale / gate_shift (f32 only), as dass_bn_bwd_reduce_gate;
// gates != NULL: the gate bits dass_bn_apply_train stored (one byte per 4-channel group) replace both
extern "C" int dass_bn_bwd_reduce_sums(const void *dout, int64_t lddo, const void *out, int64_t ldo, const void *x, int64_t ldx,
                                       const float *mean, const float *invstd, const float *gate_scale, const float *gate_shift,
                                       const float *nc_scale, int64_t M, int K, int64_t rows_per_image, int act, double *sums,
                                       const void *gates, int64_t gates_bytes, int dtype, void *stream) {
    if (!dout || !x || !mean || !invstd || !sums || M <= 0 || !ok4(K, lddo, out ? ldo : 4, ldx) || rows_per_image <= 0) return DASS_ERR_ARG;
    if (!out && !gates && (!gate_scale || !gate_shift || dtype != DASS_F32)) return DASS_ERR_ARG;
    if (gates && (nc_scale || gates_bytes < M * (K / 4))) return DASS_ERR_ARG;
    dim3 grid(stat_blocks(M, K), (unsigned)((K + 63) / 64));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32)
        DASS_LAUNCH((colstat_kernel<float, 1>), grid, dim3(256), 0, st, (const float *)x, ldx, (const float *)dout, lddo,
                           (const float *)out, ldo, mean, invstd, nc_scale, M, K, rows_per_image, act, nullptr, gate_scale, gate_shift, sums,
                           (const unsigned char *)gates);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH((colstat_kernel<bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t *)x, ldx, (const bf16_t *)dout, lddo,
                           (const bf16_t *)out, ldo, mean, invstd, nc_scale, M, K, rows_per_image, act, nullptr, nullptr, nullptr, sums,
                           (const unsigned char *)gates);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

// train-mode dx (and dres) from the f64 sums; dbeta_out / dgamma_out receive the parameter gradients (f32)
extern "C" int dass_bn_bwd_apply_sums(const void *dout, int64_t lddo, const void *out, int64_t ldo, const void *x, int64_t ldx,
                                      const float *mean, const float *invstd, const float *gamma, const double *sums, float *dbeta_out,
                                      float *dgamma_out, const float *gate_scale, const float *gate_shift, const float *nc_scale,
                                      void *dx, int64_t lddx, void *dres, int64_t lddr, int64_t M, int K, int64_t rows_per_image,
                                      double count, int act, const void *gates, int64_t gates_bytes, int dtype, void *dx3, void *stream) {
    if (!dout || !x || (!dx && !dx3) || !mean || !invstd || !sums || count <= 0 || M <= 0 || !ok4(K, lddo, out ? ldo : 4, ldx, dx ? lddx : 4) ||
        rows_per_image <= 0)
        return DASS_ERR_ARG;
    if (!out && !gates && (!gate_scale || !gate_shift || dtype != DASS_F32)) return DASS_ERR_ARG;
    if (gates && (nc_scale || gates_bytes < M * (K / 4))) return DASS_ERR_ARG;
    if (dres && lddr % 4) return DASS_ERR_ARG;
    if (dx3 && (dtype != DASS_F32 || ((uintptr_t)dx3 & 15))) return DASS_ERR_ARG;
    // (a thread of the lean kernel gets >= DASS_BN_RPT rows: at one row per thread -- the 8712 x 256 tensors under the 2048-block cap --
    //  the per-block prologue, the tensor bound over all K channels + a barrier, outweighs the row itself)
    static const int rpt = getenv("DASS_BN_RPT") ? atoi(getenv("DASS_BN_RPT")) : 8;  // (1 / 2 / 4 / 8: 3.6+ / 3.56 / 3.45 / 3.38 ms per R101 step)
    const int grid = bn_grid(M, K, rpt > 0 ? rpt : 1);
    const float inv_count = (float)(1.0 / count);
    hipStream_t st = (hipStream_t)stream;
    const char *fast_env = getenv("DASS_BN_BWD_FAST");  // (read per call: the tests flip it between two launches)
    const bool fast_on = !fast_env || atoi(fast_env) != 0;
    // the train step's form: the lean kernel (bn_bwd_fast_kernel) -- f32, split rows out, gate from bits or from the conv output, no
    // Dropout2d mask, a thread count that is a multiple of K / 4
    if (fast_on && dtype == DASS_F32 && dx3 && !out && !nc_scale && ((long)grid * 256) % (K / 4) == 0 && (gates || (gate_scale && gate_shift))) {
        const float *d = (const float *)dout, *xx = (const float *)x;
        const unsigned char *gb = (const unsigned char *)gates;
        const int parts = dass_get_x3_parts();
#define BN_FAST(G, R, D)                                                                                                              \
        DASS_LAUNCH((bn_bwd_fast_kernel<G, R, D>), dim3(grid), dim3(256), 0, st, d, lddo, xx, ldx, mean, invstd, gamma, sums, dbeta_out, dgamma_out, \
                    gate_scale, gate_shift, (float *)dx, lddx, (float *)dres, lddr, M, K, inv_count, act, gb, (char *)dx3, parts)
        if (gates) {
            if (dres) { if (dx) BN_FAST(true, true, true); else BN_FAST(true, true, false); }
            else { if (dx) BN_FAST(true, false, true); else BN_FAST(true, false, false); }
        } else {
            if (dres) { if (dx) BN_FAST(false, true, true); else BN_FAST(false, true, false); }
            else { if (dx) BN_FAST(false, false, true); else BN_FAST(false, false, false); }
        }
#undef BN_FAST
        DASS_LAUNCH_CHECK();
        return DASS_OK;
    }
    if (dtype == DASS_F32)
        DASS_LAUNCH(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)dout, lddo, (const float *)out, ldo,
                           (const float *)x, ldx, mean, invstd, gamma, (const float *)nullptr, (const float *)nullptr, nc_scale, (float *)dx,
                           lddx, (float *)dres, lddr, M, K, rows_per_image, inv_count, 1, act, gate_scale, gate_shift, (char *)dx3, sums,
                           dbeta_out, dgamma_out, (const unsigned char *)gates, dass_get_x3_parts());
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)dout, lddo, (const bf16_t *)out, ldo,
                           (const bf16_t *)x, ldx, mean, invstd, gamma, (const float *)nullptr, (const float *)nullptr, nc_scale,
                           (bf16_t *)dx, lddx, (bf16_t *)dres, lddr, M, K, rows_per_image, inv_count, 1, act, (const float *)nullptr,
                           (const float *)nullptr, (char *)nullptr, sums, dbeta_out, dgamma_out, (const unsigned char *)gates, 3);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}
