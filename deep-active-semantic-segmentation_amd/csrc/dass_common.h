// Shared device helpers for libdass_hip (gfx950 only: 64-lane waves, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dass_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;  // raw storage

// Every kernel of the library is launched through DASS_LAUNCH: the plain hipLaunchKernelGGL, or -- while bench.py holds a profile
// open (csrc/prof.hip: dass_prof_begin) -- the same launch with a start / stop event pair bound to that dispatch.
#include <hip/hip_ext.h>
extern int g_dass_prof_on;
void dass_prof_slot(const void *fn, long long grid, hipStream_t st, hipEvent_t *e0, hipEvent_t *e1);
#define DASS_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                              \
    do {                                                                                                                  \
        if (__builtin_expect(g_dass_prof_on, 0)) {                                                                        \
            hipEvent_t pe0_, pe1_;                                                                                        \
            const dim3 pg_ = (grid);                                                                                      \
            dass_prof_slot(reinterpret_cast<const void *>(kernel), (long long)pg_.x * pg_.y * pg_.z, (stream), &pe0_, &pe1_); \
            hipExtLaunchKernelGGL(kernel, pg_, (block), (shmem), (stream), pe0_, pe1_, 0, __VA_ARGS__);                   \
        } else {                                                                                                          \
            hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__);                                   \
        }                                                                                                                 \
    } while (0)

#define DASS_LAUNCH_CHECK()                               \
    do {                                                  \
        if (hipGetLastError() != hipSuccess) return DASS_ERR_LAUNCH; \
    } while (0)

// csrc/stem_rowtap.hip: the stems' f32-MFMA kernels.  1 = launched, 0 = shape outside the specialisation, < 0 = launch error
int dass_rowtap_fwd_fast(const float *x, const float *w, float *y, long ldy, int N, int H, int W, int Cin, int OH, int OW, int K, int R, int S,
                         int stride, int pad, hipStream_t st);
int dass_rowtap_wgrad_fast(const float *x, const float *dy, long lddy, float *dw, int N, int H, int W, int Cin, int OH, int OW, int K, int R,
                           int S, int stride, int pad, hipStream_t st);

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // round-to-nearest-even via the hardware convert (keeps NaN a NaN, see MI355X_MICROARCH hazards table)
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t *>(&b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t *p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t *p, float v) { *p = f32_to_bf16(v); }
};

// 4-element vector access (16 B f32 / 8 B bf16); pointers must be aligned accordingly.
template <typename T> __device__ __forceinline__ f32x4 ld4(const T *p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t *p) {
    uint2 u = *reinterpret_cast<const uint2 *>(p);
    f32x4 r;
    r[0] = __uint_as_float(u.x << 16);
    r[1] = __uint_as_float(u.x & 0xffff0000u);
    r[2] = __uint_as_float(u.y << 16);
    r[3] = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
template <typename T> __device__ __forceinline__ void st4(T *p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t *p, f32x4 v) {
    uint2 u;
    u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2 *>(p) = u;
}

// ---- exact three-way bf16 split of f32 values (the operand format of the bf16x6 engine: x = x0 + x1 + x2)
typedef float dass_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 dass_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(dass_f32x2 v) {
    dass_bf16x2 b = __builtin_convertvector(v, dass_bf16x2);  // v_cvt_pk_bf16_f32, RNE
    return *reinterpret_cast<unsigned *>(&b);
}
// four f32 -> the three bf16 parts (x0, x1, x2), 8 B each
__device__ __forceinline__ void split3_4(f32x4 v, uint2 &p0, uint2 &p1, uint2 &p2) {
    const dass_f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const unsigned h0 = pk_bf16(a), h1 = pk_bf16(b);
    const dass_f32x2 ra = a - dass_f32x2{__uint_as_float(h0 << 16), __uint_as_float(h0 & 0xffff0000u)};
    const dass_f32x2 rb = b - dass_f32x2{__uint_as_float(h1 << 16), __uint_as_float(h1 & 0xffff0000u)};
    const unsigned m0 = pk_bf16(ra), m1 = pk_bf16(rb);
    const dass_f32x2 sa = ra - dass_f32x2{__uint_as_float(m0 << 16), __uint_as_float(m0 & 0xffff0000u)};
    const dass_f32x2 sb = rb - dass_f32x2{__uint_as_float(m1 << 16), __uint_as_float(m1 & 0xffff0000u)};
    p0 = make_uint2(h0, h1);
    p1 = make_uint2(m0, m1);
    p2 = make_uint2(pk_bf16(sa), pk_bf16(sb));
}
// ---- x3 operand formats (csrc/conv_x3.hip).  An activation [rows][C] f32 is kept as rows x ceil(C/32) slabs of
//   [NP parts][32 ch] 2-byte elements (NP * 64 B), then ONE all-zero row (padded taps point at it), then a 16-B trailer
//   {float inv_scale, float bound >= max |x|, float amax (max |x| itself where the producer knows it, else the bound), 0}.
//   NP = 3 ("bf16x6" engine): x = x0 + x1 + x2, three bf16 parts, exact to 2^-26 |x|; six products per pair; trailer unused (1.0).
//   NP = 1 ("bf16x1" engine, a PERF mode -- not parity-grade): x ~ bf16(x), one part, one product per pair: what autocast-bf16 multiplies;
//            trailer unused (1.0).  f32 tensors everywhere else, exactly as in the two parity engines.
//   NP = 2 ("f16x3" engine): x * s = h0 + h1, two f16 parts (h0 = f16(x s), h1 = f16(x s - h0): 11 + 1 + 11 significant bits,
//            |x s - h0 - h1| <= 2^-23 |x s|); s = a power of two chosen per TENSOR from a guaranteed bound B >= max |x| so that
//            B s lies in [2^14, 2^15) (f16 overflows at 65504); three products per pair (h0 g1, h1 g0, h0 g0 -- the dropped
//            h1 g1 is <= 2^-22 of the product) on the f16 MFMA pipe, f32 accumulation, result multiplied by
//            inv_scale(a) * inv_scale(b) (exact: powers of two).  gfx950's MFMA honours f16 subnormal inputs
//            (tools/probes/f16_denorm_probe.hip), so elements far below the bound keep an ABSOLUTE accuracy of 2^-25 / s.
typedef _Float16 dass_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_f16(dass_f32x2 v) {
    dass_f16x2 h = __builtin_convertvector(v, dass_f16x2);  // RNE
    return *reinterpret_cast<unsigned *>(&h);
}
__device__ __forceinline__ dass_f32x2 unpk_f16(unsigned u) {
    const dass_f16x2 h = *reinterpret_cast<const dass_f16x2 *>(&u);
    return dass_f32x2{(float)h[0], (float)h[1]};
}
// power-of-two scale for a tensor whose elements are bounded by `bound`: bound * scale in [2^14, 2^15)
__device__ __forceinline__ float x3_scale_of(float bound) {
    const int e = (int)((__float_as_uint(bound) >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.f;  // zero / denormal bound (an all-zero tensor), or inf / nan: nothing sensible to scale
    int se = 268 - e;                    // exponent field of 2^(14 - (e - 127))
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ float x3_inv_of(float scale) { return __uint_as_float((254u << 23) - __float_as_uint(scale)); }  // 1 / 2^n
// four f32 (already multiplied by the tensor's scale) -> the two f16 parts, 8 B each
__device__ __forceinline__ void split2_4(f32x4 v, uint2 &p0, uint2 &p1) {
    const dass_f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const unsigned h0 = pk_f16(a), h1 = pk_f16(b);
    p0 = make_uint2(h0, h1);
    p1 = make_uint2(pk_f16(a - unpk_f16(h0)), pk_f16(b - unpk_f16(h1)));
}
// element (row m, channels k..k+3) of an x3 tensor with cc channel slabs; NP = 2: v is multiplied by `scale` first
template <int NP> __device__ __forceinline__ void x3_store4p(char *base, long m, int cc, int k, f32x4 v, float scale) {
    char *d = base + (m * cc + (k >> 5)) * (NP * 64) + (k & 31) * 2;
    if constexpr (NP == 3) {
        uint2 q0, q1, q2;
        split3_4(v, q0, q1, q2);
        *reinterpret_cast<uint2 *>(d) = q0;
        *reinterpret_cast<uint2 *>(d + 64) = q1;
        *reinterpret_cast<uint2 *>(d + 128) = q2;
    } else if constexpr (NP == 1) {  // one bf16 part (RNE): the "bf16x1" perf engine
        *reinterpret_cast<uint2 *>(d) = make_uint2(pk_bf16(dass_f32x2{v[0], v[1]}), pk_bf16(dass_f32x2{v[2], v[3]}));
    } else {
        uint2 q0, q1;
        split2_4(v * scale, q0, q1);
        *reinterpret_cast<uint2 *>(d) = q0;
        *reinterpret_cast<uint2 *>(d + 64) = q1;
    }
}
__device__ __forceinline__ void x3_store4(char *base, long m, int cc, int k, f32x4 v) { x3_store4p<3>(base, m, cc, k, v, 1.f); }
// runtime-parts form for the streaming producers (BN apply / backward): parts in {1, 2, 3}
__device__ __forceinline__ void x3_store4r(char *base, long m, int cc, int k, f32x4 v, int parts, float scale) {
    if (parts == 2) x3_store4p<2>(base, m, cc, k, v, scale);
    else if (parts == 1) x3_store4p<1>(base, m, cc, k, v, 1.f);
    else x3_store4p<3>(base, m, cc, k, v, 1.f);
}
// the all-zero row (index rows) + the trailer {inv_scale, bound}; call from ONE block of a producer kernel
__device__ __forceinline__ void x3_zero_row(char *base, long rows, int cc, int parts = 3, float inv_scale = 1.f, float bound = 0.f) {
    char *z = base + rows * cc * (parts * 64);
    for (int i = threadIdx.x; i < cc * parts * 4; i += blockDim.x) *reinterpret_cast<uint4 *>(z + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x == 0)  // {inv_scale, bound, max |x| where it is known exactly -- else the bound again --, 0}
        *reinterpret_cast<uint4 *>(z + cc * parts * 64) = make_uint4(__float_as_uint(inv_scale), __float_as_uint(bound), __float_as_uint(bound), 0u);
}
// host + device: bytes of an x3 tensor (rows + zero row + trailer) and the trailer's offset
static inline __host__ __device__ long x3_trailer_off(long rows, int cc, int parts) { return (rows + 1) * cc * (long)(parts * 64); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == DASS_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DASS_ACT_RELU6) return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    return v;
}
// derivative of the activation expressed on its OUTPUT (out>0 <=> pre-activation>0, etc.)
__device__ __forceinline__ float act_grad_from_out(float out, int act) {
    if (act == DASS_ACT_RELU) return out > 0.f ? 1.f : 0.f;
    if (act == DASS_ACT_RELU6) return (out > 0.f && out < 6.f) ? 1.f : 0.f;
    return 1.f;
}

// the BN affine exactly as the forward applies it (one fma per element): forward and backward gates must agree bit for bit
__device__ __forceinline__ f32x4 bn_affine(f32x4 v, f32x4 sc, f32x4 sh) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(v[e], sc[e], sh[e]);
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of logical tiles so
// tiles that share an operand panel hit the same L2 (cdna_hip_programming.md T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// n / d for 0 <= n < 2^31 with a host-made magic pair: mul = ceil(2^(31 + l) / d), l = ceil(log2 d), shift = l - 1
// (Granlund-Montgomery: 2^(31+l) <= mul d <= 2^(31+l) + 2^l makes the product's high part exact); d = 1 travels as shift < 0.
// Checked exhaustively against `/` on the CPU (tests/test_cpu.py through dass_x3_magic).
__device__ __forceinline__ int x3_fastdiv(int n, unsigned mul, int shift) {
    return shift < 0 ? n : (int)(__umulhi((unsigned)n, mul) >> shift);
}
static inline void x3_set_magic(int d, unsigned &mul, int &shift) {
    if (d <= 1) {
        mul = 0u;
        shift = -1;
        return;
    }
    int l = 0;
    while ((1ll << l) < d) ++l;
    mul = (unsigned)(((1ull << (31 + l)) + (unsigned long long)d - 1ull) / (unsigned long long)d);
    shift = l - 1;
}

static inline int dass_grid_1d(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    const int64_t cap = 256 * 8;  // 256 CUs x 8 blocks: grid-stride the rest
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
