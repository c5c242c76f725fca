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
    DASS_LAUNCH(confusion_kernel, dim3(grid), dim3(256), sizeof(unsigned int) * num_class * num_class,
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
    DASS_LAUNCH(bn_finalize_sums_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, K, count,
                       gamma, beta, running_mean, running_var, momentum, eps, clamp_var, mean, invstd, scale, shift);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// DIAGNOSTIC (tools/clock_probe.py; never on the product path): the shader clock the chip holds while `blocks`
// workgroups run a bf16 MFMA loop fed from LDS (the conv kernels' regime).  Each workgroup stamps s_memtime (shader
// cycles) and s_memrealtime (100 MHz) around the loop: clock = d(memtime) / d(memrealtime) * 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).  out[block] = {cycles, realtime ticks}.
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long *out, int iters, int use_lds) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0x3f803f80u ^ (i * 2654435761u >> 9);  // random-ish finite bf16 pairs
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    uint4 a = *reinterpret_cast<const uint4 *>(&lds[(threadIdx.x * 4) & 4095]);
    uint4 b = *reinterpret_cast<const uint4 *>(&lds[(threadIdx.x * 4 + 1024) & 4095]);
    if (use_lds & 2) {  // the same FLOPs per iteration from v_mfma_f32_16x16x32_bf16 (8 per iteration, 8 accumulators)
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        f32x4_t ac[8];
        for (int j = 0; j < 8; ++j)
            for (int e = 0; e < 4; ++e) ac[j][e] = 0.f;
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            if (use_lds & 1) {
                a = *reinterpret_cast<const uint4 *>(&lds[(threadIdx.x * 4 + it * 64) & 4095]);
                b = *reinterpret_cast<const uint4 *>(&lds[(threadIdx.x * 4 + it * 64 + 2048) & 4095]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                ac[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&a), *reinterpret_cast<const bf16x8 *>(&b), ac[j], 0, 0, 0);
        }
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float keep = 0.f;
        for (int j = 0; j < 8; ++j) keep += ac[j][threadIdx.x & 3];
        if (threadIdx.x == 0) {
            out[blockIdx.x * 2] = c1 - c0;
            out[blockIdx.x * 2 + 1] = r1 - r0;
        }
        if (keep == 123.456f) out[0] = 0;
        return;
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (use_lds) {  // two 16-B fragment reads per four MFMAs, like a 64 x 64 wave tile
            a = *reinterpret_cast<const uint4 *>(&lds[(threadIdx.x * 4 + it * 64) & 4095]);
            b = *reinterpret_cast<const uint4 *>(&lds[(threadIdx.x * 4 + it * 64 + 2048) & 4095]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a), *reinterpret_cast<const bf16x8 *>(&b), acc[j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float keep = 0.f;
    for (int j = 0; j < 4; ++j) keep += acc[j][threadIdx.x & 15];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = c1 - c0;
        out[blockIdx.x * 2 + 1] = r1 - r0;
    }
    if (keep == 123.456f) out[0] = 0;  // keeps the accumulators alive
}

extern "C" int dass_clock_probe(void *out, int blocks, int iters, int use_lds, void *stream) {
    if (!out || blocks <= 0 || iters <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(clock_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long *)out, iters, use_lds);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}
