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
