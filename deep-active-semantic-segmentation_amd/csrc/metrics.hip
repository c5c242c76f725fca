// Validation step after the path: the reference copies the full logits to the host, takes
// numpy argmax and bincounts a confusion matrix per batch (active_train.py:159-163, utils/metrics.py:37-42).
// Here argmax over classes and the num_class x num_class histogram are one kernel over the NCHW logits (or over a
// ready uint8 prediction map); per-block LDS histograms, then 64-bit atomics into the device matrix.
#include "dass_common.h"

namespace {

__global__ __launch_bounds__(256) void confusion_kernel(const float *__restrict__ logits, const uint8_t *__restrict__ pred,
                                                        const float *__restrict__ target, int N, int C, long HW,
                                                        int num_class, unsigned long long *__restrict__ cm) {
    extern __shared__ unsigned int hist[];  // num_class * num_class
    const int cells = num_class * num_class;
    for (int i = threadIdx.x; i < cells; i += 256) hist[i] = 0u;
    __syncthreads();
    const long total = (long)N * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float t = target[i];
        if (!(t >= 0.f && t < (float)num_class)) continue;  // mask = (gt >= 0) & (gt < num_class)
        int p;
        if (pred) {
            p = pred[i];
        } else {
            const long n = i / HW, q = i - n * HW;
            const float *x = logits + n * C * HW + q;
            float best = x[0];
            p = 0;
            for (int c = 1; c < C; ++c) {
                const float v = x[(long)c * HW];
                if (v > best) {
                    best = v;
                    p = c;
                }
            }
        }
        if (p < num_class) atomicAdd(&hist[(int)t * num_class + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += 256)
        if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}

}  // namespace

extern "C" int dass_confusion_accumulate(const float *logits, const uint8_t *pred, const float *target, int N, int C,
                                         int64_t HW, int num_class, int64_t *cm, void *stream) {
    if ((!logits && !pred) || !target || !cm || N <= 0 || HW <= 0 || num_class <= 0 || num_class > 64) return DASS_ERR_ARG;
    if (logits && C <= 0) return DASS_ERR_ARG;
    const int grid = dass_grid_1d((long)N * HW, 256);
    hipLaunchKernelGGL(confusion_kernel, dim3(grid), dim3(256), sizeof(unsigned int) * num_class * num_class,
                       (hipStream_t)stream, logits, pred, target, N, C, (long)HW, num_class, (unsigned long long *)cm);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

// ---------------------------------------------------------------------------------------------- SyncBN
// Cross-rank batch statistics: every rank reduces its partial rows to sums[2][K] (+ count),
// RCCL all-reduces that small vector, and this kernel turns the global sums into mean / invstd / scale / shift.
// clamp_var = 1 reproduces the reference's vendored SyncBN exactly: invstd = clamp(biased_var, eps)^-1/2
// (models/sync_batchnorm/batchnorm.py:113-125), not (var + eps)^-1/2; running_var uses the unbiased variance.
namespace {
__global__ void bn_finalize_sums_kernel(const float *__restrict__ sums, int K, double count, const float *gamma,
                                        const float *beta, float *running_mean, float *running_var, float momentum,
                                        float eps, int clamp_var, float *mean, float *invstd, float *scale, float *shift) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const double mu = (double)sums[k] / count;
    double var = (double)sums[K + k] / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const double is = clamp_var ? 1.0 / sqrt(var > (double)eps ? var : (double)eps) : 1.0 / sqrt(var + (double)eps);
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
}  // namespace

extern "C" int dass_bn_finalize_sums(const float *sums, int K, double count, const float *gamma, const float *beta,
                                     float *running_mean, float *running_var, float momentum, float eps, int clamp_var,
                                     float *mean, float *invstd, float *scale, float *shift, void *stream) {
    if (!sums || K <= 0 || count <= 0 || !mean || !invstd || !scale || !shift) return DASS_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, K, count,
                       gamma, beta, running_mean, running_var, momentum, eps, clamp_var, mean, invstd, scale, shift);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}
