// Per-pixel cross entropy with ignore index (utils/loss.py:39-51) and the acquisition-scoring
// reductions of active_selection/{mc_dropout,ceal,core_set}.py: fused upsample+argmax votes,
// vote entropy, softmax confidence/margin/entropy, weak labels, core-set feature pooling and the
// k-center distance update.  One thread per pixel, class planes read coalesced (NCHW) or as short
// contiguous vectors (NHWC low-res logits); every image-level sum goes through fixed-order block
// partials + an f64 finalize so scores do not depend on atomics arrival order.
#include "dass_common.h"

namespace {

constexpr int SCORE_BLOCKS = 64;  // partial blocks per image for per-image sums

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ long target_of(const void *target, int is_float, long i) {
    return is_float ? (long)reinterpret_cast<const float *>(target)[i] : reinterpret_cast<const long *>(target)[i];
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float *__restrict__ logits, const void *__restrict__ target,
                                                     int is_float, const float *__restrict__ weight, int N, int C,
                                                     long HW, int ignore, float *__restrict__ partial) {
    __shared__ float sh[4];
    const long total = (long)N * HW;
    float l = 0.f, w = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long t = target_of(target, is_float, i);
        if (t == ignore || t < 0 || t >= C) continue;
        const long n = i / HW, p = i - n * HW;
        const float *x = logits + n * C * HW + p;
        float mx = x[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[(long)c * HW]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(x[(long)c * HW] - mx);
        const float nll = (logf(s) + mx) - x[t * HW];
        const float wt = weight ? weight[t] : 1.f;
        l += wt * nll;
        w += wt;
    }
    const float ls = block_sum_256(l, sh);
    const float ws = block_sum_256(w, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2 + 0] = ls;
        partial[blockIdx.x * 2 + 1] = ws;
    }
}

__global__ void ce_finalize_kernel(const float *__restrict__ partial, int blocks, float *acc) {
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < blocks; i += 256) {
        a += (double)partial[i * 2];
        b += (double)partial[i * 2 + 1];
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        acc[0] = (float)sh[0][0];
        acc[1] = (float)sh[1][0];
    }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float *__restrict__ logits, const void *__restrict__ target,
                                                     int is_float, const float *__restrict__ weight, int N, int C,
                                                     long HW, int ignore, const float *__restrict__ acc,
                                                     const float *__restrict__ gscale, float *__restrict__ dlogits) {
    const long total = (long)N * HW;
    const float g = gscale[0] / acc[1];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long t = target_of(target, is_float, i);
        const long n = i / HW, p = i - n * HW;
        const float *x = logits + n * C * HW + p;
        float *d = dlogits + n * C * HW + p;
        if (t == ignore || t < 0 || t >= C) {
            for (int c = 0; c < C; ++c) d[(long)c * HW] = 0.f;
            continue;
        }
        float mx = x[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[(long)c * HW]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(x[(long)c * HW] - mx);
        const float wt = (weight ? weight[t] : 1.f) * g;
        const float inv = 1.f / s;
        for (int c = 0; c < C; ++c) {
            const float pr = expf(x[(long)c * HW] - mx) * inv;
            d[(long)c * HW] = wt * (pr - (c == t ? 1.f : 0.f));
        }
    }
}

// -------------------------------------------------------------------------------------- scoring
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

template <typename T>
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const T *__restrict__ x, long ldx,
                                                              uint8_t *__restrict__ votes, long vote_nstride, int N,
                                                              int IH, int IW, int C, int OH, int OW, float sh,
                                                              float sw) {
    const long ohw = (long)OH * OW;
    const long total = (long)N * ohw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / ohw, p = i - n * ohw;
        const int oy = (int)(p / OW), ox = (int)(p - (long)oy * OW);
        const Lerp ly = lerp_of(oy, IH, sh), lx = lerp_of(ox, IW, sw);
        const T *b = x + n * IH * IW * ldx;
        const T *p00 = b + ((long)ly.i0 * IW + lx.i0) * ldx, *p01 = b + ((long)ly.i0 * IW + lx.i1) * ldx;
        const T *p10 = b + ((long)ly.i1 * IW + lx.i0) * ldx, *p11 = b + ((long)ly.i1 * IW + lx.i1) * ldx;
        float best = -INFINITY;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            const float v = ly.l0 * (lx.l0 * Elem<T>::ld(p00 + c) + lx.l1 * Elem<T>::ld(p01 + c)) +
                            ly.l1 * (lx.l0 * Elem<T>::ld(p10 + c) + lx.l1 * Elem<T>::ld(p11 + c));
            if (c == 0 || v > best) {
                best = v;
                bi = c;
            }
        }
        votes[n * vote_nstride + p] = (uint8_t)bi;
    }
}

// The same values for four consecutive output pixels of a row per thread (f32 rows, ld % 4 == 0, upsampling by >= 3): the
// four pixels interpolate between at most three input columns, whose class vectors are loaded 16 bytes at a time and shared
// -- 30 wide loads per four pixels instead of 304 scalar ones.  Per pixel the arithmetic and the class order (first maximum
// wins) are those of the kernel above, so the votes are identical.
__global__ __launch_bounds__(256) void upsample_argmax4_kernel(const float *__restrict__ x, long ldx, uint8_t *__restrict__ votes,
                                                               long vote_nstride, int N, int IH, int IW, int C, int OH, int OW, float sh,
                                                               float sw) {
    const int qw = (OW + 3) >> 2;
    const long total = (long)N * OH * qw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int jx = (int)(i % qw);
        const long rowi = i / qw;
        const int oy = (int)(rowi % OH);
        const long n = rowi / OH;
        const Lerp ly = lerp_of(oy, IH, sh);
        Lerp lx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) lx[e] = lerp_of(jx * 4 + e < OW ? jx * 4 + e : OW - 1, IW, sw);
        const int c0 = lx[0].i0;  // every i0 / i1 of the four pixels lies in c0 .. c0 + 2 (clamped to the last column)
        const float *b = x + n * IH * IW * ldx;
        const float *r0 = b + (long)ly.i0 * IW * ldx, *r1 = b + (long)ly.i1 * IW * ldx;
        int col[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) col[j] = c0 + j < IW ? c0 + j : IW - 1;
        float best[4];
        int bi[4] = {0, 0, 0, 0};
        for (int cq = 0; cq < C; cq += 4) {
            f32x4 t0[3], t1[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                t0[j] = *reinterpret_cast<const f32x4 *>(r0 + (long)col[j] * ldx + cq);
                t1[j] = *reinterpret_cast<const f32x4 *>(r1 + (long)col[j] * ldx + cq);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = lx[e].i0 - c0, d = lx[e].i1 - c0;  // 0..2
                const f32x4 p00 = a == 0 ? t0[0] : (a == 1 ? t0[1] : t0[2]), p01 = d == 0 ? t0[0] : (d == 1 ? t0[1] : t0[2]);
                const f32x4 p10 = a == 0 ? t1[0] : (a == 1 ? t1[1] : t1[2]), p11 = d == 0 ? t1[0] : (d == 1 ? t1[1] : t1[2]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (cq + u >= C) break;
                    const float v = ly.l0 * (lx[e].l0 * p00[u] + lx[e].l1 * p01[u]) + ly.l1 * (lx[e].l0 * p10[u] + lx[e].l1 * p11[u]);
                    if (cq + u == 0 || v > best[e]) {
                        best[e] = v;
                        bi[e] = cq + u;
                    }
                }
            }
        }
        uint8_t *dst = votes + n * vote_nstride + (long)oy * OW + jx * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (jx * 4 + e < OW) dst[e] = (uint8_t)bi[e];
    }
}

__global__ __launch_bounds__(256) void argmax_nchw_kernel(const float *__restrict__ logits,
                                                          uint8_t *__restrict__ votes, long vote_nstride, int N, int C,
                                                          long HW, const float *__restrict__ label, int num_classes,
                                                          int mask255) {
    const long total = (long)N * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW, p = i - n * HW;
        const float *x = logits + n * C * HW + p;
        float best = x[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = x[(long)c * HW];
            if (v > best) {
                best = v;
                bi = c;
            }
        }
        if (mask255) {
            const float lb = label[i];
            if (lb < 0.f || lb >= (float)num_classes) bi = 255;
        }
        votes[n * vote_nstride + p] = (uint8_t)bi;
    }
}

// votes [N][T][HW]; grid (SCORE_BLOCKS, N)
__global__ __launch_bounds__(256) void vote_entropy_kernel(const uint8_t *__restrict__ votes,
                                                           const float *__restrict__ label, int T, long HW,
                                                           int num_classes, float *__restrict__ emap,
                                                           float *__restrict__ partial) {
    __shared__ float sh[4];
    const long n = blockIdx.y;
    const uint8_t *v = votes + n * T * HW;
    const float invT = 1.f;  // p is formed as count / T exactly as the reference does
    (void)invT;
    float acc = 0.f;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        float e = 0.f;
        bool masked = false;
        if (label) {
            const float lb = label[n * HW + p];
            masked = lb < 0.f || lb >= (float)num_classes;
        }
        if (!masked) {
            if (T <= 32) {
                unsigned char vv[32];
#pragma unroll
                for (int t = 0; t < 32; ++t) vv[t] = t < T ? v[(long)t * HW + p] : (unsigned char)255;
                for (int c = 0; c < num_classes; ++c) {
                    int cnt = 0;
#pragma unroll
                    for (int t = 0; t < 32; ++t) cnt += (vv[t] == c) ? 1 : 0;
                    if (cnt) {
                        const float pr = (float)cnt / (float)T;
                        e = e - pr * log2f(pr + 1e-12f);
                    }
                }
            } else {
                for (int c = 0; c < num_classes; ++c) {
                    int cnt = 0;
                    for (int t = 0; t < T; ++t) cnt += (v[(long)t * HW + p] == c) ? 1 : 0;
                    if (cnt) {
                        const float pr = (float)cnt / (float)T;
                        e = e - pr * log2f(pr + 1e-12f);
                    }
                }
            }
        }
        if (emap) emap[n * HW + p] = e;
        acc += e;
    }
    const float s = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[n * SCORE_BLOCKS + blockIdx.x] = s;
}

__global__ void image_sum_finalize_kernel(const float *__restrict__ partial, int N, float *__restrict__ image_sum) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int b = 0; b < SCORE_BLOCKS; ++b) s += (double)partial[(long)n * SCORE_BLOCKS + b];
    image_sum[n] = (float)s;
}

// mode 0: max softmax prob (masked -> 1); 1: top1 - top2 (masked -> 1); 2: entropy log2 (masked -> 0)
__global__ __launch_bounds__(256) void softmax_scores_kernel(const float *__restrict__ logits,
                                                             const float *__restrict__ label, int C, long HW,
                                                             int num_classes, int mode, float *__restrict__ map,
                                                             float *__restrict__ partial) {
    __shared__ float sh[4];
    const long n = blockIdx.y;
    float acc = 0.f;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        bool masked = false;
        if (label) {
            const float lb = label[n * HW + p];
            masked = lb < 0.f || lb >= (float)num_classes;
        }
        float r;
        if (masked) {
            r = mode == 2 ? 0.f : 1.f;
        } else {
            const float *x = logits + n * C * HW + p;
            float m1 = -INFINITY, m2 = -INFINITY;
            for (int c = 0; c < C; ++c) {
                const float v = x[(long)c * HW];
                if (v > m1) {
                    m2 = m1;
                    m1 = v;
                } else if (v > m2) {
                    m2 = v;
                }
            }
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += expf(x[(long)c * HW] - m1);
            if (mode == 0) {
                r = 1.f / s;
            } else if (mode == 1) {
                r = 1.f / s - expf(m2 - m1) / s;
            } else {
                r = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float pr = expf(x[(long)c * HW] - m1) / s;
                    r = r - pr * log2f(pr + 1e-12f);
                }
            }
        }
        if (map) map[n * HW + p] = r;
        acc += r;
    }
    const float s = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[n * SCORE_BLOCKS + blockIdx.x] = s;
}

// avg_pool2d(k, s) -> out[n][c*PH*PW + ph*PW + pw]; grid (C/64, PH*PW, N)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_features_kernel(const T *__restrict__ x, long ldx,
                                                               float *__restrict__ out, int H, int W, int C, int k,
                                                               int s, int PH, int PW) {
    __shared__ float red[16][64 + 1];
    const int tid = threadIdx.x, cx = tid & 15, ry = tid >> 4;
    const int c = blockIdx.x * 64 + cx * 4;
    const int ph = blockIdx.y / PW, pw = blockIdx.y - ph * PW;
    const long n = blockIdx.z;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int i = ry; i < k * k; i += 16) {
            const int dy = i / k, dx = i - dy * k;
            a += ld4<T>(x + ((n * H + ph * s + dy) * W + pw * s + dx) * ldx + c);
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ry][cx * 4 + e] = a[e];
    __syncthreads();
    if (tid < 64) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += red[i][tid];
        const int cc = blockIdx.x * 64 + tid;
        if (cc < C) out[n * C * PH * PW + ((long)cc * PH + ph) * PW + pw] = v / (float)(k * k);
    }
}

// one wave per feature row; f64 accumulation of squared differences
__global__ __launch_bounds__(256) void kcenter_update_kernel(const float *__restrict__ feat, long n, int d,
                                                             const long *__restrict__ center_ptr,
                                                             double *__restrict__ min_dist, int first) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    const long center = center_ptr[0];
    const float *a = feat + row * d, *b = feat + center * d;
    double s = 0.0;
    for (int i = lane; i < d; i += 64) {
        const double df = (double)a[i] - (double)b[i];
        s += df * df;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        const double dist = sqrt(s);
        min_dist[row] = first ? dist : fmin(min_dist[row], dist);
    }
}

// first-max argmax over doubles: stage 1 per-block partial, stage 2 single block
__global__ __launch_bounds__(256) void argmax_stage1_kernel(const double *__restrict__ v, long n,
                                                            double *__restrict__ pval, long *__restrict__ pidx) {
    __shared__ double sv[256];
    __shared__ long si[256];
    double best = -INFINITY;
    long bi = -1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double x = v[i];
        if (bi < 0 || x > best) {
            best = x;
            bi = i;
        }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const double x = sv[threadIdx.x + o];
            const long j = si[threadIdx.x + o];
            const long cur = si[threadIdx.x];
            if (j >= 0 && (cur < 0 || x > sv[threadIdx.x] || (x == sv[threadIdx.x] && j < cur))) {
                sv[threadIdx.x] = x;
                si[threadIdx.x] = j;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pval[blockIdx.x] = sv[0];
        pidx[blockIdx.x] = si[0];
    }
}
__global__ __launch_bounds__(256) void argmax_stage2_kernel(const double *__restrict__ pval,
                                                            const long *__restrict__ pidx, int blocks,
                                                            long *__restrict__ out_idx, double *__restrict__ out_val) {
    __shared__ double sv[256];
    __shared__ long si[256];
    double best = -INFINITY;
    long bi = -1;
    for (int i = threadIdx.x; i < blocks; i += 256) {
        const double x = pval[i];
        const long j = pidx[i];
        if (j >= 0 && (bi < 0 || x > best || (x == best && j < bi))) {
            best = x;
            bi = j;
        }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const double x = sv[threadIdx.x + o];
            const long j = si[threadIdx.x + o];
            const long cur = si[threadIdx.x];
            if (j >= 0 && (cur < 0 || x > sv[threadIdx.x] || (x == sv[threadIdx.x] && j < cur))) {
                sv[threadIdx.x] = x;
                si[threadIdx.x] = j;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out_idx[0] = si[0];
        if (out_val) out_val[0] = sv[0];
    }
}

// ---------------------------------------------------------------------------------------- max-subset
// greedy facility location of active_selection/max_subset.py:17-39: D[i][j] = ||a_i - b_j||_2 (f64, as
// sklearn's pairwise_distances on float64 rows), score_j = -sum_i min(mind_i, D[i][j]), mind update by column.
__global__ __launch_bounds__(256) void pairwise_dist_kernel(const float *__restrict__ a, long n,
                                                            const float *__restrict__ b, long m, int d,
                                                            double *__restrict__ D) {
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= n * m) return;
    const long i = pair / m, j = pair - i * m;
    const int lane = threadIdx.x & 63;
    const float *x = a + i * d, *y = b + j * d;
    double s = 0.0;
    for (int q = lane; q < d; q += 64) {
        const double df = (double)x[q] - (double)y[q];
        s += df * df;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) D[pair] = sqrt(s);
}

// one block per candidate column j; selected columns get -inf
__global__ __launch_bounds__(256) void facility_scores_kernel(const double *__restrict__ D, long n, long m,
                                                              const double *__restrict__ mind,
                                                              const uint8_t *__restrict__ selected,
                                                              double *__restrict__ scores) {
    __shared__ double sh[256];
    const long j = blockIdx.x;
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) s += fmin(mind[i], D[i * m + j]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) scores[j] = selected[j] ? -INFINITY : -sh[0];
}

__global__ void facility_update_kernel(const double *__restrict__ D, long n, long m, const long *__restrict__ jptr,
                                       double *__restrict__ mind, uint8_t *__restrict__ selected) {
    const long j = jptr[0];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mind[i] = fmin(mind[i], D[i * m + j]);
    if (i == 0) selected[j] = 1;
}

__global__ void sgd_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ buf, long n,
                           float lr, float momentum, float wd, int first) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] + wd * p[i];
        float b = first ? gi : momentum * buf[i] + gi;
        buf[i] = b;
        p[i] = p[i] - lr * b;
    }
}

// every parameter tensor of one optimizer step in a handful of launches: up to 64 tensors per launch travel in the kernel
// argument itself (no device-side table, no host->device copy); a block owns 2048 consecutive elements of one tensor
constexpr int SGD_PACK = 64, SGD_BLK = 2048;
struct SgdPack {
    float *p[SGD_PACK];
    const float *g[SGD_PACK];
    float *b[SGD_PACK];
    long first_block[SGD_PACK + 1];
    long numel[SGD_PACK];
    float lr[SGD_PACK];
    int n;
};

// hyper != nullptr: {lr, momentum, weight decay} are read from DEVICE memory at run time (one triple for all tensors of the launch), so a
// launch captured into a hipGraph follows a learning-rate schedule (the by-value arguments of a captured launch are frozen)
__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdPack pk, float momentum, float wd, const float *__restrict__ hyper) {
    const long blk = blockIdx.x;
    int lo = 0, hi = pk.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pk.first_block[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    float *__restrict__ p = pk.p[lo];
    const float *__restrict__ g = pk.g[lo];
    float *__restrict__ b = pk.b[lo];
    float lr = pk.lr[lo];
    if (hyper) {
        lr = hyper[0];
        momentum = hyper[1];
        wd = hyper[2];
    }
    const long n = pk.numel[lo];
    const long e0 = (blk - pk.first_block[lo]) * SGD_BLK;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)b) & 15) == 0) && e0 + SGD_BLK <= n;
    if (vec) {
#pragma unroll
        for (int r = 0; r < SGD_BLK / 1024; ++r) {
            const long i = e0 + r * 1024 + threadIdx.x * 4;
            f32x4 pv = *reinterpret_cast<const f32x4 *>(p + i);
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + i) + wd * pv;
            const f32x4 bv = momentum * *reinterpret_cast<const f32x4 *>(b + i) + gv;
            *reinterpret_cast<f32x4 *>(b + i) = bv;
            *reinterpret_cast<f32x4 *>(p + i) = pv - lr * bv;
        }
    } else {
        for (long i = e0 + threadIdx.x; i < n && i < e0 + SGD_BLK; i += 256) {
            const float gi = g[i] + wd * p[i];
            const float bi = momentum * b[i] + gi;
            b[i] = bi;
            p[i] = p[i] - lr * bi;
        }
    }
}

}  // namespace

extern "C" int dass_ce_blocks(int64_t npix) { return dass_grid_1d(npix, 256); }

extern "C" int dass_ce_fwd(const float *logits, const void *target, int target_is_float, const float *weight, int N,
                           int C, int64_t HW, int ignore_index, float *partial, void *stream) {
    if (!logits || !target || !partial || N <= 0 || C <= 0 || HW <= 0) return DASS_ERR_ARG;
    const int grid = dass_ce_blocks((int64_t)N * HW);
    DASS_LAUNCH(ce_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, target_is_float,
                       weight, N, C, HW, ignore_index, partial);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_ce_finalize(const float *partial, int blocks, float *acc, void *stream) {
    if (!partial || !acc || blocks <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, blocks, acc);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_ce_bwd(const float *logits, const void *target, int target_is_float, const float *weight, int N,
                           int C, int64_t HW, int ignore_index, const float *acc, const float *gscale,
                           float *dlogits, void *stream) {
    if (!logits || !target || !acc || !gscale || !dlogits || N <= 0 || C <= 0 || HW <= 0) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((int64_t)N * HW, 256);
    DASS_LAUNCH(ce_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, target_is_float,
                       weight, N, C, HW, ignore_index, acc, gscale, dlogits);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_upsample_argmax(const void *x, int64_t ldx, uint8_t *votes, int64_t vote_nstride, int N, int IH,
                                    int IW, int C, int OH, int OW, int dtype, void *stream) {
    if (!x || !votes || N <= 0 || IH <= 0 || IW <= 0 || C <= 0 || C > 255 || OH <= 0 || OW <= 0 || ldx < C)
        return DASS_ERR_ARG;
    const float sh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    const int grid = dass_grid_1d((long)N * OH * OW, 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32 && (ldx & 3) == 0 && !((uintptr_t)x & 15) && ldx >= ((C + 3) & ~3) && 3.f * sw <= 1.f && OW >= 4)
        // four pixels span 3 sw <= 1 input columns: their corners lie in three consecutive columns
        DASS_LAUNCH(upsample_argmax4_kernel, dim3(dass_grid_1d((long)N * OH * ((OW + 3) / 4), 256)), dim3(256), 0, st,
                           (const float *)x, ldx, votes, vote_nstride, N, IH, IW, C, OH, OW, sh, sw);
    else if (dtype == DASS_F32)
        DASS_LAUNCH(upsample_argmax_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, ldx, votes,
                           vote_nstride, N, IH, IW, C, OH, OW, sh, sw);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(upsample_argmax_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)x, ldx, votes,
                           vote_nstride, N, IH, IW, C, OH, OW, sh, sw);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_argmax_nchw(const float *logits, uint8_t *votes, int64_t vote_nstride, int N, int C, int64_t HW,
                                void *stream) {
    if (!logits || !votes || N <= 0 || C <= 0 || C > 255 || HW <= 0) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * HW, 256);
    DASS_LAUNCH(argmax_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, votes, vote_nstride,
                       N, C, HW, nullptr, 0, 0);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_weak_labels(const float *logits, const float *label, int N, int C, int64_t HW, int num_classes,
                                uint8_t *out, void *stream) {
    if (!logits || !label || !out || N <= 0 || C <= 0 || C > 255 || HW <= 0) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * HW, 256);
    DASS_LAUNCH(argmax_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, out, HW, N, C, HW,
                       label, num_classes, 1);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_score_blocks(void) { return SCORE_BLOCKS; }

extern "C" int dass_vote_entropy(const uint8_t *votes, const float *label, int N, int T, int64_t HW, int num_classes,
                                 float *entropy_map, float *partial, float *image_sum, void *stream) {
    if (!votes || !partial || !image_sum || N <= 0 || T <= 0 || HW <= 0 || num_classes <= 0 || num_classes > 255)
        return DASS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DASS_LAUNCH(vote_entropy_kernel, dim3(SCORE_BLOCKS, N), dim3(256), 0, st, votes, label, T, HW, num_classes,
                       entropy_map, partial);
    DASS_LAUNCH_CHECK();
    DASS_LAUNCH(image_sum_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, partial, N, image_sum);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_softmax_scores(const float *logits, const float *label, int N, int C, int64_t HW, int num_classes,
                                   int mode, float *map, float *partial, float *image_sum, void *stream) {
    if (!logits || !partial || !image_sum || N <= 0 || C <= 0 || HW <= 0 || mode < 0 || mode > 2) return DASS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    DASS_LAUNCH(softmax_scores_kernel, dim3(SCORE_BLOCKS, N), dim3(256), 0, st, logits, label, C, HW,
                       num_classes, mode, map, partial);
    DASS_LAUNCH_CHECK();
    DASS_LAUNCH(image_sum_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, partial, N, image_sum);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_avgpool_features(const void *x, int64_t ldx, float *out, int N, int H, int W, int C, int k, int s,
                                     int PH, int PW, int dtype, void *stream) {
    if (!x || !out || N <= 0 || C <= 0 || C % 4 || ldx % 4 || k <= 0 || s <= 0 || PH <= 0 || PW <= 0) return DASS_ERR_ARG;
    if ((PH - 1) * s + k > H || (PW - 1) * s + k > W) return DASS_ERR_ARG;
    dim3 grid((C + 63) / 64, PH * PW, N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32)
        DASS_LAUNCH(avgpool_features_kernel<float>, grid, dim3(256), 0, st, (const float *)x, ldx, out, H, W, C, k, s, PH, PW);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(avgpool_features_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t *)x, ldx, out, H, W, C, k, s, PH, PW);
    else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_kcenter_update(const float *feat, int64_t n, int d, const int64_t *center, double *min_dist,
                                   int first, void *stream) {
    if (!feat || !center || !min_dist || n <= 0 || d <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(kcenter_update_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, feat,
                       (long)n, d, (const long *)center, min_dist, first);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_argmax_blocks(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 256) g = 256;
    return (int)(g < 1 ? 1 : g);
}

extern "C" int dass_argmax_f64(const double *v, int64_t n, double *partial_val, int64_t *partial_idx,
                               int64_t *out_idx, double *out_val, void *stream) {
    if (!v || !partial_val || !partial_idx || !out_idx || n <= 0) return DASS_ERR_ARG;
    const int blocks = dass_argmax_blocks(n);
    hipStream_t st = (hipStream_t)stream;
    DASS_LAUNCH(argmax_stage1_kernel, dim3(blocks), dim3(256), 0, st, v, (long)n, partial_val, (long *)partial_idx);
    DASS_LAUNCH_CHECK();
    DASS_LAUNCH(argmax_stage2_kernel, dim3(1), dim3(256), 0, st, partial_val, (const long *)partial_idx, blocks,
                       (long *)out_idx, out_val);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_sgd_step(float *p, const float *g, float *buf, int64_t n, float lr, float momentum,
                             float weight_decay, int first_step, void *stream) {
    if (!p || !g || !buf || n <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(sgd_kernel, dim3(dass_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf, (long)n, lr,
                       momentum, weight_decay, first_step);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

struct FloatPack {
    float v[16];
};
__global__ void set_floats_kernel(float *__restrict__ dst, const FloatPack pk, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = pk.v[threadIdx.x];
}

extern "C" int dass_set_floats(float *dst, const float *vals, int n, void *stream) {
    if (!dst || !vals || n <= 0 || n > 16) return DASS_ERR_ARG;
    FloatPack pk;
    for (int i = 0; i < 16; ++i) pk.v[i] = i < n ? vals[i] : 0.f;
    DASS_LAUNCH(set_floats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, pk, n);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

static int sgd_step_multi(void *const *p, const void *const *g, void *const *buf, const int64_t *numel, const float *lr,
                          int n, float momentum, float weight_decay, const float *hyper, void *stream) {
    if (!p || !g || !buf || !numel || (!lr && !hyper) || n <= 0) return DASS_ERR_ARG;
    for (int base = 0; base < n; base += SGD_PACK) {
        SgdPack pk;
        pk.n = n - base < SGD_PACK ? n - base : SGD_PACK;
        long blocks = 0;
        for (int i = 0; i < pk.n; ++i) {
            if (!p[base + i] || !g[base + i] || !buf[base + i] || numel[base + i] <= 0) return DASS_ERR_ARG;
            pk.p[i] = (float *)p[base + i];
            pk.g[i] = (const float *)g[base + i];
            pk.b[i] = (float *)buf[base + i];
            pk.numel[i] = numel[base + i];
            pk.lr[i] = lr ? lr[base + i] : 0.f;
            pk.first_block[i] = blocks;
            blocks += (numel[base + i] + SGD_BLK - 1) / SGD_BLK;
        }
        pk.first_block[pk.n] = blocks;
        DASS_LAUNCH(sgd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pk, momentum, weight_decay, hyper);
        DASS_LAUNCH_CHECK();
    }
    return DASS_OK;
}

extern "C" int dass_sgd_step_multi(void *const *p, const void *const *g, void *const *buf, const int64_t *numel, const float *lr,
                                   int n, float momentum, float weight_decay, void *stream) {
    return sgd_step_multi(p, g, buf, numel, lr, n, momentum, weight_decay, nullptr, stream);
}

extern "C" int dass_sgd_step_multi_dev(void *const *p, const void *const *g, void *const *buf, const int64_t *numel, int n,
                                       const float *hyper, void *stream) {
    if (!hyper) return DASS_ERR_ARG;
    return sgd_step_multi(p, g, buf, numel, nullptr, n, 0.f, 0.f, hyper, stream);
}

extern "C" int dass_version(void) { return 1; }
extern "C" const char *dass_arch(void) { return "gfx950"; }

extern "C" int dass_pairwise_dist_f64(const float *a, int64_t n, const float *b, int64_t m, int d, double *D, void *stream) {
    if (!a || !b || !D || n <= 0 || m <= 0 || d <= 0) return DASS_ERR_ARG;
    const long pairs = (long)n * m;
    DASS_LAUNCH(pairwise_dist_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, (long)n,
                       b, (long)m, d, D);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_facility_scores(const double *D, int64_t n, int64_t m, const double *mind, const uint8_t *selected,
                                    double *scores, void *stream) {
    if (!D || !mind || !selected || !scores || n <= 0 || m <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(facility_scores_kernel, dim3((unsigned)m), dim3(256), 0, (hipStream_t)stream, D, (long)n, (long)m, mind,
                       selected, scores);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_facility_update(const double *D, int64_t n, int64_t m, const int64_t *col, double *mind,
                                    uint8_t *selected, void *stream) {
    if (!D || !col || !mind || !selected || n <= 0 || m <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(facility_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, D, (long)n,
                       (long)m, (const long *)col, mind, selected);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}
