// Implicit-GEMM NHWC convolution for gfx950 (CDNA4) on MFMA.
//
// GEMM view:  Y[M = N*OH*OW pixels][K out-ch] = A[M][R*S*C] x B[R*S*C][K]
//   A rows are gathered on the fly: for tap (r,s) the row of pixel m is the contiguous C-vector of
//   the shifted input pixel (NHWC), zero where the tap falls in the padding.
//   B is the KRSC weight tensor: row k is contiguous along the reduction (r,s,c).
//
// One 256-thread workgroup (4 waves) owns a BM x BN output tile.  Each reduction step stages a
// 128-byte-wide slab (32 f32 / 64 bf16 channels of ONE tap) of A and B through LDS:
//   global (16 B per lane, 8 lanes = one pixel's 128 B)  ->  registers  ->  LDS rows of 144 B
// (128 + 16 pad: ds_read_b128 of 16 different rows then hits 16 different 16-B bank slots), and the
// next slab's global loads are in flight while the current one is multiplied.
//   f32 : v_mfma_f32_32x32x2_f32, exact f32 (parity mode).  A lane reads 16 B = 4 k-values and issues
//         4 MFMAs; the k order inside the slab is permuted identically for A and B.
//   bf16: v_mfma_f32_32x32x16_bf16, one MFMA per 16-B fragment.
// Taps that are out of range for the whole tile (dilation 6/12/18 on a 33x33 map) are skipped.
// The same kernel computes input gradients: dgrad of a stride-s conv is this kernel with
// ustride = s over weights transformed by dass_weight_transform(mode=1).
//
// Reference sites replaced: every groups=1 nn.Conv2d on the DeepLab path (see include/dass_hip.h).
#include "dass_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

struct ConvP {
    const char *x;
    const char *w;
    char *y;
    const float *scale;
    const float *shift;
    const char *res;
    const float *in_scale;
    float *stat_partial;  // optional [mtiles][2][K]: per-tile column sums / sums of squares of the RAW output (train-mode BN)
    double *stat_sums;    // or: [2][K] f64 accumulators every tile adds its column sums to (hardware f64 atomics)
    long ldx, ldy, ldr;
    long wk_stride;  // R*S*C
    int N, H, W, C, OH, OW, K, R, S, stride, pad, dil, ustride, act;
    int M, cchunks, mtiles, ntiles;
    // output sub-grid (phase-decomposed dgrad of strided convs): this launch covers output pixels
    // (oy*o_mul + oy_add, ox*o_mul + ox_add), oy < OHs, ox < OWs, and only the taps in tap_allow
    int OHs, OWs, o_mul, oy_add, ox_add;
    unsigned long long tap_allow;
    int wide_c;  // row-tap mode (WIDE kernels): true input channels; p.C = S*wide_c, taps = R
};

template <typename T> struct Mma;
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const uint4 &a, const uint4 &b, f32x16 &acc) {
        const float *af = reinterpret_cast<const float *>(&a);
        const float *bf = reinterpret_cast<const float *>(&b);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4 &a, const uint4 &b, f32x16 &acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a),
                                                      *reinterpret_cast<const bf16x8 *>(&b), acc, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ uint4 scale_vec(uint4 v, const float *s);
template <> __device__ __forceinline__ uint4 scale_vec<float>(uint4 v, const float *s) {
    float *f = reinterpret_cast<float *>(&v);
    const f32x4 sv = *reinterpret_cast<const f32x4 *>(s);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] *= sv[e];
    return v;
}
template <> __device__ __forceinline__ uint4 scale_vec<bf16_t>(uint4 v, const float *s) {
    bf16_t *h = reinterpret_cast<bf16_t *>(&v);
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = f32_to_bf16(bf16_to_f32(h[e]) * s[e]);
    return v;
}

// ---- f32 operands multiplied on the bf16 MFMA ("split" mode, DASS_F32X3).  x = hi + lo + r with hi = bf16(x),
// lo = bf16(x - hi), |r| <= 2^-18 |x|;  a*b ~= a_hi*b_hi + a_lo*b_hi + a_hi*b_lo accumulated in f32 (the dropped
// a_lo*b_lo and the r terms are each <= 2^-18 |a*b|, the size of the f32 accumulation rounding of a ~1000-term
// dot product).  Three 8-pass bf16 MFMAs replace sixteen... (per 16 k) eight 16-pass f32 MFMAs: 3/16 of the
// matrix-pipe time.  A staged 16-B unit of 4 f32 becomes {hi01, hi23, lo01, lo23} -- same LDS footprint and
// addressing as f32, so only the LDS write (convert) and the MFMA cluster differ.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(f32x2 v) {
    bf16x2 b = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32, RNE
    return *reinterpret_cast<unsigned *>(&b);
}
__device__ __forceinline__ uint4 split_unit(uint4 v) {
    const f32x2 p0 = {__uint_as_float(v.x), __uint_as_float(v.y)}, p1 = {__uint_as_float(v.z), __uint_as_float(v.w)};
    const unsigned h0 = pack_bf16x2(p0), h1 = pack_bf16x2(p1);
    const f32x2 hf0 = {__uint_as_float(h0 << 16), __uint_as_float(h0 & 0xffff0000u)};
    const f32x2 hf1 = {__uint_as_float(h1 << 16), __uint_as_float(h1 & 0xffff0000u)};
    return make_uint4(h0, h1, pack_bf16x2(p0 - hf0), pack_bf16x2(p1 - hf1));
}
// two units per operand (8 k-values of this lane half): hi*hi + lo*hi of each unit, then hi*lo of both
__device__ __forceinline__ void mma_split(const uint4 &a1, const uint4 &a2, const uint4 &b1, const uint4 &b2, f32x16 &acc) {
    const uint4 bh1 = make_uint4(b1.x, b1.y, b1.x, b1.y), bh2 = make_uint4(b2.x, b2.y, b2.x, b2.y);
    const uint4 ah = make_uint4(a1.x, a1.y, a2.x, a2.y), bl = make_uint4(b1.z, b1.w, b2.z, b2.w);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&ah), *reinterpret_cast<const bf16x8 *>(&bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a1), *reinterpret_cast<const bf16x8 *>(&bh1), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a2), *reinterpret_cast<const bf16x8 *>(&bh2), acc, 0, 0, 0);
}

// Three-way split (DASS_F32X6): x = x0 + x1 + x2 exactly to 2^-26 |x| (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1));
// the six products of order <= 2^-18 (a0b0, a0b1, a1b0, a1b1, a0b2, a2b0) reproduce the f32 product to below f32
// rounding, so the result differs from the f32 MFMA only by accumulation order.  6 bf16 MFMAs per 16 k instead of
// 8 twice-as-long f32 MFMAs: 3/8 of the matrix-pipe time.  A staged unit is {x0 01, x0 23, x1 01, x1 23} (16 B) + {x2 01, x2 23} (8 B).
__device__ __forceinline__ void split3_unit(uint4 v, uint4 &hm, uint2 &lo) {
    const f32x2 p0 = {__uint_as_float(v.x), __uint_as_float(v.y)}, p1 = {__uint_as_float(v.z), __uint_as_float(v.w)};
    const unsigned h0 = pack_bf16x2(p0), h1 = pack_bf16x2(p1);
    const f32x2 r0 = p0 - f32x2{__uint_as_float(h0 << 16), __uint_as_float(h0 & 0xffff0000u)};
    const f32x2 r1 = p1 - f32x2{__uint_as_float(h1 << 16), __uint_as_float(h1 & 0xffff0000u)};
    const unsigned m0 = pack_bf16x2(r0), m1 = pack_bf16x2(r1);
    const f32x2 s0 = r0 - f32x2{__uint_as_float(m0 << 16), __uint_as_float(m0 & 0xffff0000u)};
    const f32x2 s1 = r1 - f32x2{__uint_as_float(m1 << 16), __uint_as_float(m1 & 0xffff0000u)};
    hm = make_uint4(h0, h1, m0, m1);
    lo = make_uint2(pack_bf16x2(s0), pack_bf16x2(s1));
}
// eight f32 k-values of one lane (two 16-B LDS reads) -> the three 8 x bf16 MFMA operands x0, x1, x2
__device__ __forceinline__ void split3_frag(const uint4 &lo4, const uint4 &hi4, uint4 &x0, uint4 &x1, uint4 &x2) {
    const unsigned raw[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    unsigned o0[4], o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 p = {__uint_as_float(raw[2 * j]), __uint_as_float(raw[2 * j + 1])};
        const unsigned h = pack_bf16x2(p);
        const f32x2 r = p - f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
        const unsigned m = pack_bf16x2(r);
        const f32x2 q = r - f32x2{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
        o0[j] = h;
        o1[j] = m;
        o2[j] = pack_bf16x2(q);
    }
    x0 = make_uint4(o0[0], o0[1], o0[2], o0[3]);
    x1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    x2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}
__device__ __forceinline__ f32x16 mfma_bf16(const uint4 &a, const uint4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a), *reinterpret_cast<const bf16x8 *>(&b), c, 0, 0, 0);
}

// WIDE ("row-tap") variant for the 3-channel network stems (7x7/s2 ResNet, 3x3/s2 MobileNet): a tap is a
// whole kernel ROW -- the S*Cin input values x[n, iy, ix0 .. ix0+S-1, :] are contiguous in a dense NHWC
// image -- so the reduction is R slabs of S*Cin (=21 or 9) values instead of R*S slabs of a padded channel.
// PF = slabs in flight per workgroup (register prefetch ring).  The bf16-pipe engines finish a slab's MFMAs in a
// fraction of the ~1 us a global load takes even from L2, so with <= 2-4 resident workgroups per CU one slab of
// prefetch leaves every iteration waiting for memory; small tiles (few staging registers) keep 2-3 slabs in flight.
template <typename T, int BM, int BN, int WM, int WN, bool WIDE = false, int SPLIT = 0, int PF = 1>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? (SPLIT == 3 ? 2 : 3) : ((BM * BN >= 128 * 64) ? (SPLIT == 3 ? (PF > 1 ? 2 : 3) : 4) : (SPLIT == 3 ? (PF > 1 ? 3 : 4) : 5)))
void conv_igemm_kernel(const ConvP p) {
    static_assert(!SPLIT || sizeof(T) == 4, "split mode multiplies f32 tensors");
    // LDS bytes per staged row: 128 data + 16 pad (ds_read_b128 of 16 rows -> 16 different 16-B bank slots).
    // SPLIT == 3: A rows stay raw f32 (converted by the wave that multiplies them, next to its MFMAs); B rows
    // come PRE-SPLIT from dass_weight_transform(DASS_F32X6): three 64-B bf16 planes per 32-k slab, 192 + 16 pad.
    // ASTG: wave tiles of ONE column block (MT = NT = 1) would convert every A fragment for a single MFMA column, so
    // there A is split once per element while it is staged instead (three 64-B planes per row, like B)
    constexpr bool ASTG = SPLIT == 3 && (BM / WM) == 32 && (BN / WN) == 32;
    constexpr int ROWB = ASTG ? 208 : 144;
    constexpr int ROWB_B = SPLIT == 3 ? 208 : 144;
    constexpr int NB = SPLIT == 3 ? (BN * 12 + 255) / 256 : BN / 32;  // 16-B chunks of B per thread and slab
    constexpr int ES = sizeof(T);
    constexpr int EPV = 16 / ES;  // elements per 16-byte vector
    constexpr int BK = 8 * EPV;   // reduction elements per staged slab (128 B)
    constexpr int AR = BM / 32, BR = BN / 32;
    constexpr int TMW = BM / WM, TNW = BN / WN, MT = TMW / 32, NT = TNW / 32;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(MT >= 1 && NT >= 1, "wave tile");

    __shared__ __attribute__((aligned(16))) char smem[BM * ROWB + BN * ROWB_B];
    __shared__ long tap_tab[64];  // per tap: byte offset of its input pixel relative to a row's base pixel (wave-uniform)
    char *As = smem;
    char *Bs = smem + BM * ROWB;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int mt_i = wg / p.ntiles, nt_i = wg - mt_i * p.ntiles;
    const int m0 = mt_i * BM, n0 = nt_i * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = tid >> 3, lchunk = tid & 7;
    const int wm = wave / WN, wn = wave - wm * WN;

    // ---- per-thread description of the A rows it stages.  For every tap the input pixel of a row is
    // (by + ey(r), bx + ex(s)) with (ey, ex) the same for all rows, so a row keeps ONE base offset and a
    // validity bit per tap; the per-slab address is base + a wave-uniform tap delta + channel offset.
    //   forward conv         : by = oh*stride - pad, ey = r*dil
    //   phase of a dgrad     : by = ohs (sub-grid index), ey = (oy_add - pad + r*dil) / ustride  (exact)
    const int ntaps_all = p.R * p.S;
    int a_by[AR], a_bx[AR], a_n[AR];
    long a_base[AR];
    unsigned long long a_vmask[AR];
    bool a_ok[AR];
    const int ohw = p.OHs * p.OWs;
    const bool phase = p.o_mul != 1;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int m = m0 + lrow + 32 * j;
        a_ok[j] = m < p.M;
        const int mm = a_ok[j] ? m : 0;
        const int n = mm / ohw;
        const int rem = mm - n * ohw;
        const int ohs = rem / p.OWs;
        const int ows = rem - ohs * p.OWs;
        a_by[j] = phase ? ohs : ohs * p.stride - p.pad;
        a_bx[j] = phase ? ows : ows * p.stride - p.pad;
        a_n[j] = n;
        a_base[j] = (((long)n * p.H + a_by[j]) * p.W + a_bx[j]) * p.ldx * ES;
        a_vmask[j] = 0ull;
    }
    auto tap_ey = [&](int r) -> int { return phase ? (p.oy_add - p.pad + r * p.dil) / p.ustride : r * p.dil; };
    auto tap_ex = [&](int s) -> int { return phase ? (p.ox_add - p.pad + s * p.dil) / p.ustride : s * p.dil; };

    // ---- which taps touch this tile at all (and, per row, which taps are inside the image)
    const int ntaps = ntaps_all;
    unsigned long long tapmask = 0ull;
    for (int t = 0; t < ntaps; ++t) {
        if (!((p.tap_allow >> t) & 1ull)) continue;  // uniform: host-side phase filter
        const int r = t / p.S, s = t - r * p.S;
        const int ey = tap_ey(r), ex = tap_ex(s);
        if (tid == 0) tap_tab[t] = ((long)ey * p.W + ex) * p.ldx * ES;  // read after the barriers below
        bool any = false;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const int iy = a_by[j] + ey, ix = a_bx[j] + ex;
            // WIDE: only the row must exist; columns are checked per element when loading
            const bool ok = a_ok[j] && iy >= 0 && iy < p.H && (WIDE || (ix >= 0 && ix < p.W));
            if (ok) a_vmask[j] |= (1ull << t);
            any = any || ok;
        }
        if (__syncthreads_or((int)any)) tapmask |= (1ull << t);
    }

    long b_off[NB];
    int b_lds[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if constexpr (SPLIT == 3) {
            // chunk q of the tile's B slab: row q / 12 (out channel), 16-B chunk q % 12 of its 192-B split row
            const int q = tid + 256 * j;
            const int row = q / 12, ch = q - row * 12;
            const int k = n0 + row;
            b_off[j] = (row < BN && k < p.K) ? (long)k * ntaps_all * p.cchunks * 192 + ch * 16 : -1;
            b_lds[j] = row < BN ? row * ROWB_B + ch * 16 : -1;
        } else {
            const int k = n0 + lrow + 32 * j;
            b_off[j] = k < p.K ? (long)k * p.wk_stride * ES : -1;
            b_lds[j] = (lrow + 32 * j) * ROWB_B + lchunk * 16;
        }
    }
    uint4 ra_[PF][AR], rb_[PF][NB];
    auto load_stage = [&](uint4(&ra)[AR], uint4(&rb)[NB], int t, int cc) {
        // integer divisions of the tap decomposition are paid once per kernel (tap_tab), not once per slab
        const long tap_x_off = WIDE ? ((long)tap_ey(t / p.S) * p.W + tap_ex(t % p.S)) * p.ldx * ES : tap_tab[t];  // wave-uniform
        const long tap_w_off = (long)t * p.C * ES;
        const int c = cc * BK + lchunk * EPV;
        const bool okc = c < p.C;
        const long cb = (long)c * ES;
        if (WIDE) {
            // element-wise (unaligned, per-column validity): x index q -> column bx + q / Cin
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                T tmp[EPV];
                const bool rowok = (a_vmask[j] >> t) & 1ull;
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const int q = c + e;
                    const int ix = a_bx[j] + q / p.wide_c;
                    const bool ok = rowok && q < p.C && ix >= 0 && ix < p.W;
                    tmp[e] = ok ? *reinterpret_cast<const T *>(p.x + a_base[j] + tap_x_off + cb + e * ES) : (T)0;
                }
                ra[j] = *reinterpret_cast<const uint4 *>(tmp);
            }
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                T tmp[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e)
                    tmp[e] = (b_off[j] >= 0 && c + e < p.C) ? *reinterpret_cast<const T *>(p.w + b_off[j] + tap_w_off + cb + e * ES) : (T)0;
                rb[j] = *reinterpret_cast<const uint4 *>(tmp);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (((a_vmask[j] >> t) & 1ull) && okc) {
                v = *reinterpret_cast<const uint4 *>(p.x + a_base[j] + tap_x_off + cb);
                if (p.in_scale) v = scale_vec<T>(v, p.in_scale + (long)a_n[j] * p.C + c);
            }
            ra[j] = v;
        }
        if constexpr (SPLIT == 3) {
            const long slab_off = ((long)t * p.cchunks + cc) * 192;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (b_off[j] >= 0) v = *reinterpret_cast<const uint4 *>(p.w + b_off[j] + slab_off);
                rb[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (b_off[j] >= 0 && okc) v = *reinterpret_cast<const uint4 *>(p.w + b_off[j] + tap_w_off + cb);
                rb[j] = v;
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- reduction loop: channel slab OUTER, active taps INNER -- consecutive slabs re-read the same
    // (shifted) neighbourhood of the input, so the 9x tap re-reads are served by L1/L2 instead of the fabric
    unsigned long long it_mask = tapmask;
    int it_cc = 0;
    auto next_slab = [&](int &t, int &cc) -> bool {  // block-uniform
        if (!it_mask) {
            if (tapmask == 0ull || it_cc + 1 >= p.cchunks) return false;
            ++it_cc;
            it_mask = tapmask;
        }
        t = __builtin_ctzll(it_mask);
        it_mask &= it_mask - 1;
        cc = it_cc;
        return true;
    };
    bool live[PF];
#pragma unroll
    for (int sg = 0; sg < PF; ++sg) {
        int t, cc;
        live[sg] = next_slab(t, cc);
        if (live[sg]) load_stage(ra_[sg], rb_[sg], t, cc);
    }
#ifdef DASS_STAMP  // `make stamp`: per-segment s_memtime instrumentation of this loop, read by tools/conv_stamp.py
    auto stamp = [&]() -> unsigned long long {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    unsigned long long seg[5] = {0, 0, 0, 0, 0};
    const unsigned long long t_begin = stamp();
    unsigned long long t0 = t_begin, t1;
#define STAMP(i) { t1 = stamp(); seg[i] += t1 - t0; t0 = t1; }
#else
#define STAMP(i)
#endif
    bool more = true;
    while (more) {
#pragma unroll
      for (int sg = 0; sg < PF; ++sg) {
        if (!live[sg]) {
            more = false;
            break;
        }
        uint4(&ra)[AR] = ra_[sg];
        uint4(&rb)[NB] = rb_[sg];
        __syncthreads();  // everyone finished reading the previous slab
        STAMP(0)
#pragma unroll
        for (int j = 0; j < AR; ++j)
            if constexpr (ASTG) {
                uint4 hm;
                uint2 lo;
                split3_unit(ra[j], hm, lo);
                char *d = As + (lrow + 32 * j) * ROWB + lchunk * 8;
                *reinterpret_cast<uint2 *>(d) = make_uint2(hm.x, hm.y);
                *reinterpret_cast<uint2 *>(d + 64) = make_uint2(hm.z, hm.w);
                *reinterpret_cast<uint2 *>(d + 128) = lo;
            } else {
                *reinterpret_cast<uint4 *>(As + (lrow + 32 * j) * ROWB + lchunk * 16) = SPLIT == 2 ? split_unit(ra[j]) : ra[j];
            }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (SPLIT != 3 || b_lds[j] >= 0) *reinterpret_cast<uint4 *>(Bs + b_lds[j]) = SPLIT == 2 ? split_unit(rb[j]) : rb[j];
        STAMP(1)
        __syncthreads();
        STAMP(2)

        // refill this register stage with the slab PF ahead (loads stay in flight under the MFMAs)
        {
            int t, cc;
            live[sg] = next_slab(t, cc);
            if (live[sg]) load_stage(ra, rb, t, cc);
        }
        STAMP(3)

        const char *ap = As + (wm * TMW + (lane & 31)) * ROWB + (lane >> 5) * 16;
        const char *bp = Bs + (wn * TNW + (lane & 31)) * ROWB_B + (lane >> 5) * 16;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (SPLIT == 3) {
            const char *bp3 = Bs + (wn * TNW + (lane & 31)) * ROWB_B + (lane >> 5) * 16;
#pragma unroll
            for (int st = 0; st < 2; ++st) {  // two k16 steps per 32-k slab; this lane half owns k = 16 st + 8 h .. + 7
                uint4 a0[MT], a1[MT], a2[MT], b0[NT], b1[NT], b2[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    b0[nt] = *reinterpret_cast<const uint4 *>(bp3 + nt * 32 * ROWB_B + st * 32);
                    b1[nt] = *reinterpret_cast<const uint4 *>(bp3 + nt * 32 * ROWB_B + 64 + st * 32);
                    b2[nt] = *reinterpret_cast<const uint4 *>(bp3 + nt * 32 * ROWB_B + 128 + st * 32);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    // ap already carries (lane>>5)*16; the 8 f32 of this half are 32 B at (2 st + h) * 32
                    if constexpr (ASTG) {
                        const char *src = ap + mt * 32 * ROWB + st * 32;  // ap carries (lane>>5)*16: plane p at + 64 p
                        a0[mt] = *reinterpret_cast<const uint4 *>(src);
                        a1[mt] = *reinterpret_cast<const uint4 *>(src + 64);
                        a2[mt] = *reinterpret_cast<const uint4 *>(src + 128);
                    } else {
                        const char *src = ap + mt * 32 * ROWB + st * 64 + (lane >> 5) * 16;
                        split3_frag(*reinterpret_cast<const uint4 *>(src), *reinterpret_cast<const uint4 *>(src + 16), a0[mt], a1[mt], a2[mt]);
                    }
                }
                // six products, smallest first; MT*NT independent accumulators between dependent MFMAs
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const uint4 &av = term == 1 ? a2[mt] : ((term == 2 || term == 4) ? a1[mt] : a0[mt]);
                            const uint4 &bv = term == 0 ? b2[nt] : ((term == 2 || term == 3) ? b1[nt] : b0[nt]);
                            acc[mt][nt] = mfma_bf16(av, bv, acc[mt][nt]);  // (a0,b2) (a2,b0) (a1,b1) (a0,b1) (a1,b0) (a0,b0)
                        }
            }
        } else if constexpr (SPLIT == 2) {
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                uint4 a1[MT], a2[MT], b1[NT], b2[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    a1[mt] = *reinterpret_cast<const uint4 *>(ap + mt * 32 * ROWB + i * 32);
                    a2[mt] = *reinterpret_cast<const uint4 *>(ap + mt * 32 * ROWB + i * 32 + 32);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    b1[nt] = *reinterpret_cast<const uint4 *>(bp + nt * 32 * ROWB + i * 32);
                    b2[nt] = *reinterpret_cast<const uint4 *>(bp + nt * 32 * ROWB + i * 32 + 32);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma_split(a1[mt], a2[mt], b1[nt], b2[nt], acc[mt][nt]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 a[MT], b[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const uint4 *>(ap + mt * 32 * ROWB + i * 32);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const uint4 *>(bp + nt * 32 * ROWB + i * 32);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) Mma<T>::run(a[mt], b[nt], acc[mt][nt]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        STAMP(4)
      }
    }
#ifdef DASS_STAMP
    if (p.stat_partial) {  // debug build: the "statistics" buffer receives wave 0's segment cycles of 8 sample workgroups
        const int slot = (blockIdx.x * 8) / gridDim.x;
        if (threadIdx.x == 0 && blockIdx.x == (slot * gridDim.x + 7) / 8) {
            float *o = p.stat_partial + slot * 8;
            o[0] = (float)(t0 - t_begin);
            for (int i = 0; i < 5; ++i) o[1 + i] = (float)seg[i];
        }
        return;
    }
#endif

    if (p.stat_partial || p.stat_sums) {
        // BatchNorm batch statistics fused into the producer: per-channel sum and sum of squares of this tile's
        // rows straight from the accumulators (rows >= M hold zeros), one partial row per M-tile; the f64
        // finalize kernel reduces them.  Saves the separate read of the conv output.
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem);  // [WM][2][BN]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float v = acc[mt][nt][reg];
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                red[(wm * 2 + 0) * BN + wn * TNW + nt * 32 + lane] = s1;
                red[(wm * 2 + 1) * BN + wn * TNW + nt * 32 + lane] = s2;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid - which * BN;
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < WM; ++q) a += red[(q * 2 + which) * BN + col];
            if (n0 + col < p.K) {
                if (p.stat_sums) unsafeAtomicAdd(p.stat_sums + (long)which * p.K + n0 + col, (double)a);
                else p.stat_partial[((long)mt_i * 2 + which) * p.K + n0 + col] = a;
            }
        }
    }

    // ---- epilogue.  Accumulators sit in the MFMA C/D layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)): stored
    // from there a lane would write single dwords (16 store instructions per 32x32 block).  Each wave instead bounces its
    // block through a private LDS patch and re-reads it row-major, so every lane moves 16 contiguous bytes: 4x fewer
    // store (and residual load) instructions, full 128-B lines per 8 lanes.
    T *y = reinterpret_cast<T *>(p.y);
    const T *res = reinterpret_cast<const T *>(p.res);
    constexpr int PBLK = (NT >= 2 && 4 * 32 * 68 * 4 <= (int)sizeof(smem)) ? 2 : 1;  // 32-column blocks per pass
    constexpr int PW = PBLK * 32, PITCH = PW + 4, C4 = PW / 4;
    static_assert(4 * 32 * PITCH * 4 <= (int)sizeof(smem), "epilogue patches must fit the staging buffers");
    static_assert(NT % PBLK == 0, "column blocks per pass");
    __syncthreads();  // every wave is out of the main loop / statistics block: the stage is free
    float *patch = reinterpret_cast<float *>(smem) + wave * 32 * PITCH;
    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.K & 3) == 0) && (!res || (p.ldr & 3) == 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int ntp = 0; ntp < NT; ntp += PBLK) {
#pragma unroll
            for (int q = 0; q < PBLK; ++q)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    patch[row * PITCH + q * 32 + (lane & 31)] = acc[mt][ntp + q][reg];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < (32 * C4) / 64; ++it) {
                const int idx = it * 64 + lane;
                const int row = idx / C4, c4 = idx - row * C4;
                const int m = m0 + wm * TMW + mt * 32 + row;
                const int k = n0 + wn * TNW + ntp * 32 + c4 * 4;
                if (m >= p.M || k >= p.K) continue;
                long mo = m;  // output pixel index; differs from m only for a phase sub-grid
                if (p.o_mul != 1) {
                    const int n = m / ohw, rem = m - n * ohw;
                    const int ohs = rem / p.OWs;
                    mo = ((long)n * p.OH + ohs * p.o_mul + p.oy_add) * p.OW + (rem - ohs * p.OWs) * p.o_mul + p.ox_add;
                }
                f32x4 v = *reinterpret_cast<const f32x4 *>(patch + row * PITCH + c4 * 4);
                if (vec_ok) {
                    if (p.scale) v *= *reinterpret_cast<const f32x4 *>(p.scale + k);
                    if (p.shift) v += *reinterpret_cast<const f32x4 *>(p.shift + k);
                    if (res) v += ld4<T>(res + mo * p.ldr + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
                    st4<T>(y + mo * p.ldy + k, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (k + e >= p.K) break;
                        float u = v[e] * (p.scale ? p.scale[k + e] : 1.f) + (p.shift ? p.shift[k + e] : 0.f);
                        if (res) u += Elem<T>::ld(res + mo * p.ldr + k + e);
                        Elem<T>::st(y + mo * p.ldy + k + e, apply_act(u, p.act));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next pass
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad
// dW[k][tap][c] = sum_pixels dY[pix][k] * X[pix @ tap][c].  GEMM with the PIXEL axis as reduction:
// both operands sit in memory as [pixel][channel], which is exactly what the f32 MFMA wants when the
// LDS tile is kept [pixel][channel]: lane (i = lane&31, h = lane>>5) reads element [2*kk+h][i], i.e.
// 32 consecutive floats per half-wave -> conflict-free ds_read_b32.  One workgroup = one
// (K-tile, C-tile, tap, pixel-split); partial tiles are accumulated into dW with f32 atomics
// (256 contiguous bytes per wave-instruction, the shape that runs at the full atomic rate).
struct WgradP {
    const char *x;
    const char *dy;
    float *dw;
    long ldx, lddy;
    int N, H, W, C, OH, OW, K, R, S, stride, pad, dil;
    int M, ktiles, ctiles, psplit, pix_per_split;
    int wide_c;
};

template <typename T, int BMK, int BNC, bool WIDE = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradP p) {
    constexpr int BP = 32;          // pixels per staged slab
    constexpr int LDA = BMK + 4;    // floats per LDS row (pad keeps 16-B alignment, staggers rows)
    constexpr int LDB = BNC + 4;
    constexpr int TMW = BMK / 2, TNW = BNC / 2, MT = TMW / 32, NT = TNW / 32;
    constexpr int CPR_A = BMK / 4, RPP_A = 256 / CPR_A, PASS_A = BP / RPP_A;  // 16-B chunks of 4 f32 in LDS
    constexpr int CPR_B = BNC / 4, RPP_B = 256 / CPR_B, PASS_B = BP / RPP_B;
    static_assert(MT >= 1 && NT >= 1, "tile");
    static_assert(PASS_A >= 1 && PASS_B >= 1, "passes");

    __shared__ __attribute__((aligned(16))) float As[BP * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BP * LDB];

    // Workgroups that read the SAME pixel range (all taps, all K/C tiles of one pixel split) get consecutive
    // logical ids and, through the XCD remap, the same L2: dY / X slabs come from HBM once per split instead
    // of once per (tap, tile).
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tap = wg % (p.R * p.S);
    wg /= (p.R * p.S);
    const int ct = wg % p.ctiles;
    wg /= p.ctiles;
    const int kt = wg % p.ktiles;
    const int ps = wg / p.ktiles;
    const int k0 = kt * BMK, c0 = ct * BNC;
    const int r = tap / p.S, s = tap - r * p.S;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ohw = p.OH * p.OW;

    const long pbeg = (long)ps * p.pix_per_split;
    long pend = pbeg + p.pix_per_split;
    if (pend > p.M) pend = p.M;

    const int arow = tid / CPR_A, achunk = tid % CPR_A;
    const int brow = tid / CPR_B, bchunk = tid % CPR_B;
    const T *xg = reinterpret_cast<const T *>(p.x);
    const T *dyg = reinterpret_cast<const T *>(p.dy);

    f32x4 ra[PASS_A], rb[PASS_B];
    // per-thread pixel cursors advance by BP every slab: (oh, ow) and the element offset of the tap's input pixel
    // are carried incrementally (adds only) instead of being re-derived with divisions and 64-bit multiplies;
    // VALU issue slots next to 64-cycle MFMAs are not free (PMC: 4.9 VALU per MFMA before this, 60 % MFMA busy)
    int b_oh[PASS_B], b_ow[PASS_B];
    long b_xoff[PASS_B];
    const long adv_px = (long)BP * p.stride * p.ldx;
    const long adv_row = ((long)p.stride * p.W - (long)p.OW * p.stride) * p.ldx;
    const long adv_img = ((long)p.H * p.W - (long)p.OH * p.stride * p.W) * p.ldx;
    const int tap_dy = -p.pad + r * p.dil, tap_dx = -p.pad + (WIDE ? 0 : s * p.dil);
#pragma unroll
    for (int j = 0; j < PASS_B; ++j) {
        const long pix = pbeg + brow + j * RPP_B;
        const int n = (int)(pix / ohw);
        const int rem = (int)(pix - (long)n * ohw);
        b_oh[j] = rem / p.OW;
        b_ow[j] = rem - b_oh[j] * p.OW;
        b_xoff[j] = (((long)n * p.H + (b_oh[j] * p.stride + tap_dy)) * p.W + (b_ow[j] * p.stride + tap_dx)) * p.ldx;
    }
    const bool a_kok = (k0 + achunk * 4) < p.K, b_cok = (c0 + bchunk * 4) < p.C;
    const T *dy_ptr[PASS_A];
#pragma unroll
    for (int j = 0; j < PASS_A; ++j) dy_ptr[j] = dyg + k0 + achunk * 4 + (pbeg + arow + j * RPP_A) * p.lddy;
    const long adv_dy = (long)BP * p.lddy;
    const T *x_col = xg + c0 + bchunk * 4;
    auto load_stage = [&](long pb) {
        const int left = (int)(pend - pb);  // pixels still to do from this slab's first row (uniform)
#pragma unroll
        for (int j = 0; j < PASS_A; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (arow + j * RPP_A < left && a_kok) v = ld4<T>(dy_ptr[j]);
            ra[j] = v;
            dy_ptr[j] += adv_dy;
        }
#pragma unroll
        for (int j = 0; j < PASS_B; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const bool live = brow + j * RPP_B < left;
            const int iy = b_oh[j] * p.stride + tap_dy;
            const int ix = b_ow[j] * p.stride + tap_dx;
            if (WIDE) {
                // row-tap mode: "channel" q = c0 + 4*bchunk + e addresses column ix + q / Cin of input row iy
                if (live && iy >= 0 && iy < p.H) {
                    const T *row = x_col + b_xoff[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int q = c0 + bchunk * 4 + e;
                        const int ixe = ix + q / p.wide_c;
                        if (q < p.C && ixe >= 0 && ixe < p.W) v[e] = Elem<T>::ld(row + e);
                    }
                }
            } else if (live && b_cok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                v = ld4<T>(x_col + b_xoff[j]);
            }
            rb[j] = v;
            // advance this cursor by BP pixels
            b_ow[j] += BP;
            b_xoff[j] += adv_px;
            while (b_ow[j] >= p.OW) {
                b_ow[j] -= p.OW;
                b_xoff[j] += adv_row;
                if (++b_oh[j] >= p.OH) {
                    b_oh[j] = 0;
                    b_xoff[j] += adv_img;
                }
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    long pb = pbeg;
    if (pb < pend) load_stage(pb);
    while (pb < pend) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PASS_A; ++j) *reinterpret_cast<f32x4 *>(&As[(arow + j * RPP_A) * LDA + achunk * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < PASS_B; ++j) *reinterpret_cast<f32x4 *>(&Bs[(brow + j * RPP_B) * LDB + bchunk * 4]) = rb[j];
        __syncthreads();
        pb += BP;
        if (pb < pend) load_stage(pb);

        const float *ap = As + (lane >> 5) * LDA + wm * TMW + (lane & 31);
        const float *bp = Bs + (lane >> 5) * LDB + wn * TNW + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BP / 2; ++kk) {
            float a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = ap[kk * 2 * LDA + mt * 32];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = bp[kk * 2 * LDB + nt * 32];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    }

    const long rs = (long)p.R * p.S;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = c0 + wn * TNW + nt * 32 + (lane & 31);
        if (c >= p.C) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int k = k0 + wm * TMW + mt * 32 + row;
                if (k < p.K) atomicAdd(p.dw + ((long)k * rs + tap) * p.C + c, acc[mt][nt][reg]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad, bf16
// Same decomposition as conv_wgrad_kernel, on v_mfma_f32_32x32x16_bf16.  Both operands want 8 consecutive
// PIXELS (the reduction index) of one channel per lane while memory and the LDS image are [pixel][channel]:
// exactly the case for gfx950's transposed LDS read.  ds_read_b64_tr_b16 hands lane i of a 16-lane group
// column i of a 4-row x 16-column block, so two reads give the 8-deep fragment with no shuffles.  LDS rows
// are pitched at (2*channels + 64) bytes so the four rows of a block fall in four different 64-B bank windows.
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2u lds_read_tr16(unsigned addr) {
    v2u v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// SPLIT (DASS_F32X3): the tensors are f32; a staged 16-B chunk of 4 channels is split into bf16 hi / lo parts
// that go to two [pixel][channel] bf16 planes, and every k-step issues hi*lo + lo*hi + hi*hi (see split_unit).
template <int BMK, int BNC, int SPLIT = 0>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const WgradP p) {
    typedef typename std::conditional<SPLIT != 0, float, bf16_t>::type E;
    constexpr int BP = SPLIT ? 32 : 64;                  // pixels per staged slab (16 per MFMA k-step)
    constexpr int EPC = SPLIT ? 4 : 8;                   // channels per 16-B global chunk
    constexpr int PA = BMK * 2 + 64, PB = BNC * 2 + 64;  // LDS row pitch in bytes (per bf16 plane)
    constexpr int TMW = BMK / 2, TNW = BNC / 2, MT = TMW / 32, NT = TNW / 32;
    constexpr int CPR_A = BMK / EPC, RPP_A = 256 / CPR_A, PASS_A = BP / RPP_A;
    constexpr int CPR_B = BNC / EPC, RPP_B = 256 / CPR_B, PASS_B = BP / RPP_B;
    constexpr int PLANES = SPLIT ? SPLIT : 1;  // 2 = hi/lo (3 products), 3 = x0/x1/x2 (6 products)
    static_assert(MT >= 1 && NT >= 1 && PASS_A >= 1 && PASS_B >= 1, "tile");

    __shared__ __attribute__((aligned(16))) char smem[PLANES * (BP * PA + BP * PB)];
    char *As = smem;                     // plane pl (SPLIT) at As + pl * BP * PA
    char *Bs = smem + PLANES * BP * PA;  // plane pl (SPLIT) at Bs + pl * BP * PB

    // Workgroups that read the SAME pixel range (all taps, all K/C tiles of one pixel split) get consecutive
    // logical ids and, through the XCD remap, the same L2: dY / X slabs come from HBM once per split instead
    // of once per (tap, tile).
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tap = wg % (p.R * p.S);
    wg /= (p.R * p.S);
    const int ct = wg % p.ctiles;
    wg /= p.ctiles;
    const int kt = wg % p.ktiles;
    const int ps = wg / p.ktiles;
    const int k0 = kt * BMK, c0 = ct * BNC;
    const int r = tap / p.S, s = tap - r * p.S;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ohw = p.OH * p.OW;
    const long pbeg = (long)ps * p.pix_per_split;
    long pend = pbeg + p.pix_per_split;
    if (pend > p.M) pend = p.M;

    const int arow = tid / CPR_A, achunk = tid % CPR_A;
    const int brow = tid / CPR_B, bchunk = tid % CPR_B;
    const E *xg = reinterpret_cast<const E *>(p.x);
    const E *dyg = reinterpret_cast<const E *>(p.dy);

    int b_oh[PASS_B], b_ow[PASS_B];
    long b_xoff[PASS_B];
    const long adv_px = (long)BP * p.stride * p.ldx;
    const long adv_row = ((long)p.stride * p.W - (long)p.OW * p.stride) * p.ldx;
    const long adv_img = ((long)p.H * p.W - (long)p.OH * p.stride * p.W) * p.ldx;
    const int tap_dy = -p.pad + r * p.dil, tap_dx = -p.pad + s * p.dil;
#pragma unroll
    for (int j = 0; j < PASS_B; ++j) {
        const long pix = pbeg + brow + j * RPP_B;
        const int n = (int)(pix / ohw);
        const int rem = (int)(pix - (long)n * ohw);
        b_oh[j] = rem / p.OW;
        b_ow[j] = rem - b_oh[j] * p.OW;
        b_xoff[j] = (((long)n * p.H + (b_oh[j] * p.stride + tap_dy)) * p.W + (b_ow[j] * p.stride + tap_dx)) * p.ldx;
    }
    const bool a_kok = (k0 + achunk * EPC) < p.K, b_cok = (c0 + bchunk * EPC) < p.C;
    const E *dy_ptr[PASS_A];
#pragma unroll
    for (int j = 0; j < PASS_A; ++j) dy_ptr[j] = dyg + k0 + achunk * EPC + (pbeg + arow + j * RPP_A) * p.lddy;
    const long adv_dy = (long)BP * p.lddy;
    const E *x_col = xg + c0 + bchunk * EPC;
    uint4 ra[PASS_A], rb[PASS_B];
    auto load_stage = [&](long pb) {
        const int left = (int)(pend - pb);
#pragma unroll
        for (int j = 0; j < PASS_A; ++j) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (arow + j * RPP_A < left && a_kok) v = *reinterpret_cast<const uint4 *>(dy_ptr[j]);
            ra[j] = v;
            dy_ptr[j] += adv_dy;
        }
#pragma unroll
        for (int j = 0; j < PASS_B; ++j) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int iy = b_oh[j] * p.stride + tap_dy;
            const int ix = b_ow[j] * p.stride + tap_dx;
            if (brow + j * RPP_B < left && b_cok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                v = *reinterpret_cast<const uint4 *>(x_col + b_xoff[j]);
            rb[j] = v;
            b_ow[j] += BP;
            b_xoff[j] += adv_px;
            while (b_ow[j] >= p.OW) {
                b_ow[j] -= p.OW;
                b_xoff[j] += adv_row;
                if (++b_oh[j] >= p.OH) {
                    b_oh[j] = 0;
                    b_xoff[j] += adv_img;
                }
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // transposed-read lane roles: group g = lane>>4 -> channel half (g&1), pixel half (g>>1); inside the
    // group lane 4q+p supplies the address of row q, 4-channel column chunk p
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const unsigned a_base = (unsigned)(size_t)As + (unsigned)((8 * (g >> 1) + q) * PA + (wm * TMW + 16 * (g & 1) + 4 * pp) * 2);
    const unsigned b_base = (unsigned)(size_t)Bs + (unsigned)((8 * (g >> 1) + q) * PB + (wn * TNW + 16 * (g & 1) + 4 * pp) * 2);

    long pb = pbeg;
    if (pb < pend) load_stage(pb);
    while (pb < pend) {
        __syncthreads();
        if constexpr (SPLIT == 3) {
#pragma unroll
            for (int j = 0; j < PASS_A; ++j) {
                uint4 hm;
                uint2 lo;
                split3_unit(ra[j], hm, lo);
                char *d = As + (arow + j * RPP_A) * PA + achunk * 8;
                *reinterpret_cast<uint2 *>(d) = make_uint2(hm.x, hm.y);
                *reinterpret_cast<uint2 *>(d + BP * PA) = make_uint2(hm.z, hm.w);
                *reinterpret_cast<uint2 *>(d + 2 * BP * PA) = lo;
            }
#pragma unroll
            for (int j = 0; j < PASS_B; ++j) {
                uint4 hm;
                uint2 lo;
                split3_unit(rb[j], hm, lo);
                char *d = Bs + (brow + j * RPP_B) * PB + bchunk * 8;
                *reinterpret_cast<uint2 *>(d) = make_uint2(hm.x, hm.y);
                *reinterpret_cast<uint2 *>(d + BP * PB) = make_uint2(hm.z, hm.w);
                *reinterpret_cast<uint2 *>(d + 2 * BP * PB) = lo;
            }
        } else if constexpr (SPLIT == 2) {
#pragma unroll
            for (int j = 0; j < PASS_A; ++j) {
                const uint4 u = split_unit(ra[j]);
                *reinterpret_cast<uint2 *>(As + (arow + j * RPP_A) * PA + achunk * 8) = make_uint2(u.x, u.y);
                *reinterpret_cast<uint2 *>(As + BP * PA + (arow + j * RPP_A) * PA + achunk * 8) = make_uint2(u.z, u.w);
            }
#pragma unroll
            for (int j = 0; j < PASS_B; ++j) {
                const uint4 u = split_unit(rb[j]);
                *reinterpret_cast<uint2 *>(Bs + (brow + j * RPP_B) * PB + bchunk * 8) = make_uint2(u.x, u.y);
                *reinterpret_cast<uint2 *>(Bs + BP * PB + (brow + j * RPP_B) * PB + bchunk * 8) = make_uint2(u.z, u.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PASS_A; ++j) *reinterpret_cast<uint4 *>(As + (arow + j * RPP_A) * PA + achunk * 16) = ra[j];
#pragma unroll
            for (int j = 0; j < PASS_B; ++j) *reinterpret_cast<uint4 *>(Bs + (brow + j * RPP_B) * PB + bchunk * 16) = rb[j];
        }
        __syncthreads();
        pb += BP;
        if (pb < pend) load_stage(pb);
#pragma unroll
        for (int ks = 0; ks < BP / 16; ++ks) {
            v2u af[PLANES][MT][2], bfr[PLANES][NT][2];
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    af[pl][mt][0] = lds_read_tr16(a_base + pl * BP * PA + (ks * 16) * PA + mt * 64);
                    af[pl][mt][1] = lds_read_tr16(a_base + pl * BP * PA + (ks * 16 + 4) * PA + mt * 64);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bfr[pl][nt][0] = lds_read_tr16(b_base + pl * BP * PB + (ks * 16) * PB + nt * 64);
                    bfr[pl][nt][1] = lds_read_tr16(b_base + pl * BP * PB + (ks * 16 + 4) * PB + nt * 64);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            auto frag = [](const v2u(&f)[2]) { return make_uint4(f[0][0], f[0][1], f[1][0], f[1][1]); };
            auto mma = [&](const uint4 &a4, const uint4 &b4, f32x16 &c) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a4), *reinterpret_cast<const bf16x8 *>(&b4), c, 0, 0, 0);
            };
            if constexpr (SPLIT == 3) {
                // six products, smallest first: (A plane, B plane) = (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            constexpr int PA_OF[6] = {0, 2, 1, 0, 1, 0}, PB_OF[6] = {2, 0, 1, 1, 0, 0};
                            mma(frag(af[PA_OF[term]][mt]), frag(bfr[PB_OF[term]][nt]), acc[mt][nt]);
                        }
            } else if constexpr (SPLIT == 2) {
                // order: the two small cross terms first, hi*hi last; 4 independent accumulators between dependent MFMAs
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            mma(frag(af[term == 1 ? 1 : 0][mt]), frag(bfr[term == 0 ? 1 : 0][nt]), acc[mt][nt]);
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma(frag(af[0][mt]), frag(bfr[0][nt]), acc[mt][nt]);
            }
        }
    }

    const long rs = (long)p.R * p.S;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = c0 + wn * TNW + nt * 32 + (lane & 31);
        if (c >= p.C) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int k = k0 + wm * TMW + mt * 32 + row;
                if (k < p.K) atomicAdd(p.dw + ((long)k * rs + tap) * p.C + c, acc[mt][nt][reg]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ weights
template <typename T>
__global__ void weight_transform_kernel(const float *__restrict__ src, T *__restrict__ dst, int K, int R, int S,
                                        int Csrc, int Cdst, int mode) {
    const long total = (long)K * R * S * Cdst;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (mode == 0) {
            // dst [K][R][S][Cdst]
            const int c = (int)(i % Cdst);
            const long krs = i / Cdst;
            const float v = c < Csrc ? src[krs * Csrc + c] : 0.f;
            Elem<T>::st(dst + i, v);
        } else {
            // dst [C=Cdst][R][S][K] with taps flipped; src [K][R][S][Csrc]; Cdst == Csrc here
            const int k = (int)(i % K);
            long t = i / K;
            const int s = (int)(t % S);
            t /= S;
            const int r = (int)(t % R);
            const int c = (int)(t / R);
            const float v = src[(((long)k * R + (R - 1 - r)) * S + (S - 1 - s)) * Csrc + c];
            Elem<T>::st(dst + i, v);
        }
    }
}

// DASS_F32X6 weight operand: rows x taps x ceil(red/32) slabs of [3 parts][32 k] bf16 (192 B), zero padded.
// mode 0: rows = K, reduction = C (forward);  mode 1: rows = C, reduction = K, taps flipped (dgrad).
__device__ __forceinline__ void weight_split3_item(const float *__restrict__ src, bf16_t *__restrict__ dst, int K, int R, int S,
                                                   int Csrc, int Cdst, int mode, long i) {
    const int red = mode == 0 ? Cdst : K, taps = R * S, cch = (red + 31) / 32;
    const int j = (int)(i & 31);
    long q = i >> 5;
    const int cc = (int)(q % cch);
    q /= cch;
    const int t = (int)(q % taps);
    const int row = (int)(q / taps);
    const int r = t / S, s2 = t - r * S;
    const int e = cc * 32 + j;
    float v = 0.f;
    if (mode == 0) {
        if (e < Csrc) v = src[(((long)row * R + r) * S + s2) * Csrc + e];
    } else if (e < K) {
        v = src[(((long)e * R + (R - 1 - r)) * S + (S - 1 - s2)) * Csrc + row];
    }
    const bf16_t h = f32_to_bf16(v);
    const float r1 = v - bf16_to_f32(h);
    const bf16_t m = f32_to_bf16(r1);
    const bf16_t l = f32_to_bf16(r1 - bf16_to_f32(m));
    bf16_t *d = dst + (((long)row * taps + t) * cch + cc) * 96 + j;
    d[0] = h;
    d[32] = m;
    d[64] = l;
}

__global__ void weight_split3_kernel(const float *__restrict__ src, bf16_t *__restrict__ dst, int K, int R, int S, int Csrc,
                                     int Cdst, int mode) {
    const int rows = mode == 0 ? K : Csrc, red = mode == 0 ? Cdst : K;
    const long total = (long)rows * R * S * ((red + 31) / 32) * 32;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        weight_split3_item(src, dst, K, R, S, Csrc, Cdst, mode, i);
}

// every conv weight of a network in ONE launch (the optimizer step invalidates all of them at once).
// desc[w] = {src, dst, K, R, S, Csrc, Cdst, mode} as 8 x int64; start[w] = first TILE of weight w, a tile being 32 operand
// rows x one 32-wide reduction slab of one tap.  One 256-thread block per tile: the source is read along its contiguous
// axis (c), transposed through LDS for the dgrad operand (mode 1), and written along the slab (coalesced both ways).
// parts = 3: [3][32] bf16 slabs (DASS_F32X6); parts = 1: [1][32] bf16 slabs, the plain RNE cast (DASS_BF16X1, + a 1.0 trailer)
__global__ __launch_bounds__(256) void weight_split3_batch_kernel(const long *__restrict__ desc, const long *__restrict__ start, int n,
                                                                  long total_tiles, int parts) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    // a block owns a CONTIGUOUS range of tiles: one binary search for its first tile, then the table index only moves forward (the
    // grid-stride form searched the table for every tile: ~8 dependent loads in front of each 4 KB of weights, 1.7 TB/s in all)
    const long per_block = (total_tiles + gridDim.x - 1) / gridDim.x;
    const long tl_beg = (long)blockIdx.x * per_block, tl_end = tl_beg + per_block < total_tiles ? tl_beg + per_block : total_tiles;
    int lo = 0;
    if (tl_beg < tl_end) {
        int hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (start[mid] <= tl_beg) lo = mid; else hi = mid - 1;
        }
    }
    for (long tl = tl_beg; tl < tl_end; ++tl) {
        while (lo + 1 < n && start[lo + 1] <= tl) ++lo;
        const long *d = desc + 8 * lo;
        const float *src = reinterpret_cast<const float *>(d[0]);
        bf16_t *dst = reinterpret_cast<bf16_t *>(d[1]);
        const int K = (int)d[2], R = (int)d[3], S = (int)d[4], Csrc = (int)d[5], Cdst = (int)d[6], mode = (int)d[7];
        const int rows = mode == 0 ? K : Csrc, red = mode == 0 ? Cdst : K, taps = R * S, cch = (red + 31) / 32;
        long q = tl - start[lo];
        const int cc = (int)(q % cch);
        q /= cch;
        const int t = (int)(q % taps);
        const int rb = (int)(q / taps);  // row block
        const int r = t / S, s2 = t - r * S;
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = ty + 8 * p;
            float v = 0.f;
            if (mode == 0) {  // tile[row][e]: source rows are operand rows, contiguous along e = c
                const int row = rb * 32 + y, e = cc * 32 + tx;
                if (row < rows && e < Csrc) v = src[(((long)row * R + r) * S + s2) * Csrc + e];
                tile[y][tx] = v;
            } else {  // source row = reduction index k, contiguous along the operand row c: store transposed
                const int e = cc * 32 + y, row = rb * 32 + tx;
                if (e < K && row < rows) v = src[(((long)e * R + (R - 1 - r)) * S + (S - 1 - s2)) * Csrc + row];
                tile[tx][y] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = ty + 8 * p, row = rb * 32 + y;
            if (row >= rows) continue;
            const float v = tile[y][tx];
            const bf16_t h = f32_to_bf16(v);
            const float r1 = v - bf16_to_f32(h);
            const bf16_t m = f32_to_bf16(r1);
            const bf16_t l = f32_to_bf16(r1 - bf16_to_f32(m));
            bf16_t *o = dst + (((long)row * taps + t) * cch + cc) * (32 * parts) + tx;
            o[0] = h;
            if (parts == 3) {
                o[32] = m;
                o[64] = l;
            }
        }
        if (parts == 1 && tl == start[lo] && threadIdx.x == 0)  // the trailer every pre-split operand carries: inverse scale 1.0
            *reinterpret_cast<uint4 *>(dst + (long)rows * taps * cch * 32) = make_uint4(0x3f800000u, 0u, 0u, 0u);
    }
}

// ---- DASS_F16X3 weight operand (dass_common.h "x3 operand formats", NP = 2): rows x taps x ceil(red/32) slabs of [2 parts][32 k]
// f16 (128 B) of w * s, s = the power-of-two scale of the tensor's max |w| (kept, with its inverse, in the 16-B trailer).
// Pass 1 (amax_only): every tile's max |w| -> atomic max into the trailer's bound slot; pass 2: the split.  Same tiling and
// descriptor table as weight_split3_batch_kernel.
__global__ __launch_bounds__(256) void weight_split2_batch_kernel(const long *__restrict__ desc, const long *__restrict__ start, int n,
                                                                  long total_tiles, int amax_only) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    // A block owns a CONTIGUOUS range of 32 x 32 tiles: one binary search for its first tile, after that the table index only moves forward
    // and the table row stays in registers while the tiles belong to one tensor; the loads of tile i + 1 are issued before tile i is
    // transposed through LDS and stored.  (The grid-stride form re-read the table and searched it for every tile -- two dependent
    // global round trips in front of each 4 KB of weights -- and had one tile in flight per block: 1.7 TB/s over the 1 GB a step's refresh moves.)
    const long per_block = (total_tiles + gridDim.x - 1) / gridDim.x;
    const long tl_beg = (long)blockIdx.x * per_block, tl_end = tl_beg + per_block < total_tiles ? tl_beg + per_block : total_tiles;
    if (tl_beg >= tl_end) return;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (start[mid] <= tl_beg) lo = mid; else hi = mid - 1;
    }
    struct Ent {
        const float *src;
        char *dst;
        int K, R, S, Csrc, mode, rows, taps, cch;
        long st0, st1;
    } ent;
    auto load_ent = [&](int i) {
        const long *d = desc + 8 * i;
        ent.src = reinterpret_cast<const float *>(d[0]);
        ent.dst = reinterpret_cast<char *>(d[1]);
        ent.K = (int)d[2]; ent.R = (int)d[3]; ent.S = (int)d[4]; ent.Csrc = (int)d[5]; ent.mode = (int)d[7];
        const int Cdst = (int)d[6];
        ent.rows = ent.mode == 0 ? ent.K : ent.Csrc;
        ent.taps = ent.R * ent.S;
        ent.cch = ((ent.mode == 0 ? Cdst : ent.K) + 31) / 32;
        ent.st0 = start[i];
        ent.st1 = i + 1 < n ? start[i + 1] : total_tiles;
    };
    load_ent(lo);
    struct Dec {
        char *dst;
        int rows, taps, cch, cc, t, rb;
        bool first;
    };
    // decode tile tl (advancing the table row if needed) and issue its four loads per thread; values land in tile[][] order: v[p] belongs
    // to tile[ty + 8 p][tx] (mode 0) or tile[tx][ty + 8 p] (mode 1: the source is read along the operand row, stored transposed)
    auto fetch = [&](long tl, Dec &dc, float (&v)[4], int &mode) {
        while (tl >= ent.st1) load_ent(++lo);
        long q = tl - ent.st0;
        dc.first = q == 0;
        dc.cc = (int)(q % ent.cch);
        q /= ent.cch;
        dc.t = (int)(q % ent.taps);
        dc.rb = (int)(q / ent.taps);
        dc.dst = ent.dst; dc.rows = ent.rows; dc.taps = ent.taps; dc.cch = ent.cch;
        mode = ent.mode;
        const int r = dc.t / ent.S, s2 = dc.t - r * ent.S;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = ty + 8 * p;
            long idx;
            bool ok;
            if (ent.mode == 0) {
                const int row = dc.rb * 32 + y, e = dc.cc * 32 + tx;
                ok = row < ent.rows && e < ent.Csrc;
                idx = (((long)row * ent.R + r) * ent.S + s2) * ent.Csrc + e;
            } else {
                const int e = dc.cc * 32 + y, row = dc.rb * 32 + tx;
                ok = e < ent.K && row < ent.rows;
                idx = (((long)e * ent.R + (ent.R - 1 - r)) * ent.S + (ent.S - 1 - s2)) * ent.Csrc + row;
            }
            const float x = ent.src[ok ? idx : 0];  // (clamped address + select: no branch around the load)
            v[p] = ok ? x : 0.f;
        }
    };
    Dec cur, nxt;
    float vc[4], vn[4];
    int mode_c = 0, mode_n = 0;
    fetch(tl_beg, cur, vc, mode_c);
    for (long tl = tl_beg; tl < tl_end; ++tl) {
        const bool more = tl + 1 < tl_end;
        if (more) fetch(tl + 1, nxt, vn, mode_n);
        unsigned *tr = reinterpret_cast<unsigned *>(cur.dst + (long)cur.rows * cur.taps * cur.cch * 128);
        if (amax_only) {
            float mx = fmaxf(fmaxf(fabsf(vc[0]), fabsf(vc[1])), fmaxf(fabsf(vc[2]), fabsf(vc[3])));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            __syncthreads();  // (red[] of the previous tile consumed)
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
            __syncthreads();
            if (threadIdx.x == 0) {
                mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
                if (!(mx >= 0.f)) mx = __uint_as_float(0x7f800000u);
                atomicMax(tr + 1, __float_as_uint(mx));
            }
        } else {
            __syncthreads();  // previous tile fully consumed
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (mode_c == 0) tile[ty + 8 * p][tx] = vc[p];
                else tile[tx][ty + 8 * p] = vc[p];
            }
            const float scale = x3_scale_of(__uint_as_float(tr[1]));
            if (cur.first && threadIdx.x == 0) tr[0] = __float_as_uint(x3_inv_of(scale));
            __syncthreads();
            {   // 8 threads per operand row, four values each: one 8-byte store of high parts, one of low parts (was 2-byte stores)
                const int c4 = threadIdx.x & 7, y = threadIdx.x >> 3, row = cur.rb * 32 + y;
                if (row < cur.rows) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 hi4, lo4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = tile[y][4 * c4 + j] * scale;
                        const _Float16 h = (_Float16)v;
                        hi4[j] = h;
                        lo4[j] = (_Float16)(v - (float)h);
                    }
                    _Float16 *o = reinterpret_cast<_Float16 *>(cur.dst) + (((long)row * cur.taps + cur.t) * cur.cch + cur.cc) * 64 + 4 * c4;
                    *reinterpret_cast<h4 *>(o) = hi4;
                    *reinterpret_cast<h4 *>(o + 32) = lo4;
                }
            }
        }
        if (more) {
            cur = nxt;
            mode_c = mode_n;
#pragma unroll
            for (int p = 0; p < 4; ++p) vc[p] = vn[p];
        }
    }
}
// trailers of n operands zeroed (bound slot is an atomic max)
__global__ void weight_trailer_zero_kernel(const long *__restrict__ desc, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long *d = desc + 8 * i;
    const int K = (int)d[2], R = (int)d[3], S = (int)d[4], Csrc = (int)d[5], Cdst = (int)d[6], mode = (int)d[7];
    const int rows = mode == 0 ? K : Csrc, red_n = mode == 0 ? Cdst : K;
    *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(d[1]) + (long)rows * R * S * ((red_n + 31) / 32) * 128) = make_uint4(0u, 0u, 0u, 0u);
}

template <typename T, int BM, int BN, int WM, int WN, bool WIDE = false, int SPLIT = 0, int PF = 1> int launch_conv(ConvP &p, hipStream_t st) {
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.K + BN - 1) / BN;
    const int grid = p.mtiles * p.ntiles;
    DASS_LAUNCH((conv_igemm_kernel<T, BM, BN, WM, WN, WIDE, SPLIT, PF>), dim3(grid), dim3(256), 0, st, p);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

// Tile choice by a wave-quantisation cost model.  Workgroups are dealt round-robin over 256 CUs and the
// kernel is MFMA-bound, so a launch takes ceil(workgroups / 256) "rounds" of one tile each (the busiest CU
// decides), times the tile's MFMA work over its efficiency (small tiles stage more bytes per flop and leave
// fewer MFMAs between barriers; a single workgroup per CU has nothing to overlap its loads with).  Padded
// rows/columns of edge tiles are paid for.  Efficiencies from tools/conv_sweep.py.
template <typename T, int SPLIT = 0> int dispatch_conv(ConvP &p, hipStream_t st) {
    constexpr int EPV = 16 / sizeof(T);
    p.cchunks = (p.C + 8 * EPV - 1) / (8 * EPV);
    if (p.K <= 32) return launch_conv<T, 128, 32, 4, 1, false, SPLIT>(p, st);
    static const int force = getenv("DASS_CONV_TILE") ? atoi(getenv("DASS_CONV_TILE")) : 0;  // tuning knob (tools/conv_sweep.py)
    static const int pf = getenv("DASS_CONV_PF") ? atoi(getenv("DASS_CONV_PF")) : 0;  // tuning knob, split-3 engine only
    if (force == 1) return launch_conv<T, 128, 128, 2, 2, false, SPLIT>(p, st);
    if constexpr (SPLIT == 3) {
        if (force == 4) return launch_conv<T, 128, 128, 4, 1, false, SPLIT>(p, st);
        if (force == 5) return launch_conv<T, 128, 64, 4, 1, false, SPLIT>(p, st);
        if (force == 6) return launch_conv<T, 64, 128, 2, 2, false, SPLIT>(p, st);
        if (force == 2 && pf == 2) return launch_conv<T, 128, 64, 2, 2, false, SPLIT, 2>(p, st);
        if (force == 3 && pf == 2) return launch_conv<T, 64, 64, 2, 2, false, SPLIT, 2>(p, st);
        if (force == 3 && pf == 3) return launch_conv<T, 64, 64, 2, 2, false, SPLIT, 3>(p, st);
    }
    if (force == 2) return launch_conv<T, 128, 64, 2, 2, false, SPLIT>(p, st);
    if (force == 3) return launch_conv<T, 64, 64, 2, 2, false, SPLIT>(p, st);
    auto cost = [&](int bm, int bn, double eff) {
        const long wgs = (long)((p.M + bm - 1) / bm) * ((p.K + bn - 1) / bn);
        const long rounds = (wgs + 255) / 256;
        const double lonely = wgs < 256 ? 0.7 : 1.0;  // one workgroup per CU: no partner to hide latency
        return (double)rounds * bm * bn / (eff * lonely);
    };
    if constexpr (SPLIT == 3) {
        // bf16x6: the wave that multiplies an A fragment also converts it (36 VALU per 8-value fragment), so layouts
        // whose waves span the whole tile width (WN = 1: one conversion feeds NT = 4 / 2 column blocks) or at least two
        // column blocks win over square wave tiles wherever the grid still fills the chip; efficiencies measured with
        // tools/conv_sweep.py (DASS_CONV_TILE = 3..6) on every DeepLab-R101 shape, forward and dgrad.
        const double a = p.K > 64 ? cost(128, 128, 1.0) : 1e30;   // 128x128, waves 4x1 (32x128 each)
        const double b = cost(128, 64, 0.93);                     // 128x64,  waves 4x1 (32x64)
        const double c = p.K > 64 ? cost(64, 128, 0.91) : 1e30;   // 64x128,  waves 2x2 (32x64)
        const double d = cost(64, 64, 0.72);                      // 64x64,   waves 2x2 (32x32)
        if (a <= b && a <= c && a <= d) return launch_conv<T, 128, 128, 4, 1, false, SPLIT>(p, st);
        if (b <= c && b <= d) return launch_conv<T, 128, 64, 4, 1, false, SPLIT>(p, st);
        if (c <= d) return launch_conv<T, 64, 128, 2, 2, false, SPLIT>(p, st);
        return launch_conv<T, 64, 64, 2, 2, false, SPLIT>(p, st);
    }
    const double c128 = p.K > 64 ? cost(128, 128, 1.0) : 1e30;
    const double c12864 = cost(128, 64, 0.93);
    const double c64 = cost(64, 64, 0.85);
    if (c128 <= c12864 && c128 <= c64) return launch_conv<T, 128, 128, 2, 2, false, SPLIT>(p, st);
    if (c12864 <= c64) return launch_conv<T, 128, 64, 2, 2, false, SPLIT>(p, st);
    return launch_conv<T, 64, 64, 2, 2, false, SPLIT>(p, st);
}

// Work split of the weight gradient.  Every pixel-split adds its whole K x R*S x C tile set into dW with
// f32 atomics (chip-wide ~1.3 TB/s), so the split count is a trade: enough workgroups to fill 256 CUs,
// but each one long enough (>= MIN_SLABS slabs of 32 pixels) that the atomic tail stays small.
// dass_set_deterministic(1): ONE pixel split per weight-gradient tile -- every dW element is then summed by a single
// workgroup in a fixed order (no f32 atomics between workgroups): bit-reproducible run to run, at the price of fewer, longer
// workgroups on layers with few output tiles.
static int g_deterministic = 0;
extern "C" int dass_set_deterministic(int on) {
    g_deterministic = on ? 1 : 0;
    return DASS_OK;
}
extern "C" int dass_get_deterministic(void) { return g_deterministic; }

static long wgrad_split(long base, long M, long target_wgs, long min_slabs) {
    if (g_deterministic) return 1;
    long want = (target_wgs + base - 1) / base;
    long maxsplit = M / (32 * min_slabs);
    if (maxsplit < 1) maxsplit = 1;
    if (want > maxsplit) want = maxsplit;
    return want < 1 ? 1 : want;
}

template <typename T, int BMK, int BNC, bool WIDE = false> int launch_wgrad(WgradP &p, hipStream_t st, long split) {
    p.ktiles = (p.K + BMK - 1) / BMK;
    p.ctiles = (p.C + BNC - 1) / BNC;
    const long base = (long)p.ktiles * p.ctiles * p.R * p.S;
    long pps = (p.M + split - 1) / split;
    pps = (pps + 31) / 32 * 32;
    p.pix_per_split = (int)pps;
    p.psplit = (int)((p.M + pps - 1) / pps);
    const long grid = base * p.psplit;
    DASS_LAUNCH((conv_wgrad_kernel<T, BMK, BNC, WIDE>), dim3((unsigned)grid), dim3(256), 0, st, p);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

template <int BMK, int BNC, int SPLIT = 0> int launch_wgrad_bf16(WgradP &p, hipStream_t st, long split) {
    constexpr int BP = SPLIT ? 32 : 64;
    p.ktiles = (p.K + BMK - 1) / BMK;
    p.ctiles = (p.C + BNC - 1) / BNC;
    const long base = (long)p.ktiles * p.ctiles * p.R * p.S;
    long pps = (p.M + split - 1) / split;
    pps = (pps + BP - 1) / BP * BP;
    p.pix_per_split = (int)pps;
    p.psplit = (int)((p.M + pps - 1) / pps);
    DASS_LAUNCH((conv_wgrad_bf16_kernel<BMK, BNC, SPLIT>), dim3((unsigned)(base * p.psplit)), dim3(256), 0, st, p);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

template <int SPLIT = 0> int dispatch_wgrad_bf16(WgradP &p, hipStream_t st) {
    const long rs = (long)p.R * p.S;
    auto base_of = [&](int bk, int bc) { return (long)((p.K + bk - 1) / bk) * ((p.C + bc - 1) / bc) * rs; };
    // tuning knobs (tools/conv_sweep.py): workgroups to aim for, minimum 32-pixel slabs per workgroup, forced tile
    static const long target = getenv("DASS_WGRAD_TARGET") ? atol(getenv("DASS_WGRAD_TARGET")) : 768;
    static const long min_slabs = getenv("DASS_WGRAD_MINSLABS") ? atol(getenv("DASS_WGRAD_MINSLABS")) : 16;
    static const int force = getenv("DASS_WGRAD_TILE") ? atoi(getenv("DASS_WGRAD_TILE")) : 0;
    if (p.K > 64 && p.C > 64 && force != 3) {
        const long b = base_of(128, 128);
        const long sp = wgrad_split(b, p.M, target, min_slabs);
        bool big = b * sp >= 400;
        if (big && SPLIT == 3 && force == 0) {
            // every pixel split adds its whole tile set into dW with f32 atomics (~1.3 TB/s chip-wide): with many splits the
            // big tile's better MFMA rate (measured ~125 vs ~103 TFLOP/s, tools/conv_sweep.py) is eaten by the atomic tail
            const long sp64 = wgrad_split(base_of(64, 64), p.M, target, min_slabs);
            const double flops = 2.0 * p.M * p.K * p.C * (double)rs, dwb = 4.0 * p.K * p.C * (double)rs;
            const double t128 = flops / 125e12 + dwb * sp / 1.3e12, t64 = flops / 103e12 + dwb * sp64 / 1.3e12;
            big = t128 <= t64;
        }
        if (big || force == 1) return launch_wgrad_bf16<128, 128, SPLIT>(p, st, sp);
    }
    return launch_wgrad_bf16<64, 64, SPLIT>(p, st, wgrad_split(base_of(64, 64), p.M, target, min_slabs));
}

template <typename T> int dispatch_wgrad(WgradP &p, hipStream_t st) {
    const long rs = (long)p.R * p.S;
    auto base_of = [&](int bk, int bc) { return (long)((p.K + bk - 1) / bk) * ((p.C + bc - 1) / bc) * rs; };
    const long target = 640, min_slabs = 16;
    const bool kbig = p.K > 64, cbig = p.C > 64;
    if (kbig && cbig) {
        // big tiles unless they cannot fill the chip with long-enough workgroups
        const long b = base_of(128, 128);
        const long sp = wgrad_split(b, p.M, target, min_slabs);
        if (b * sp >= 400) return launch_wgrad<T, 128, 128>(p, st, sp);
        const long b2 = base_of(64, 64);
        return launch_wgrad<T, 64, 64>(p, st, wgrad_split(b2, p.M, target, min_slabs));
    }
    if (kbig) return launch_wgrad<T, 128, 64>(p, st, wgrad_split(base_of(128, 64), p.M, target, min_slabs));
    if (cbig) return launch_wgrad<T, 64, 128>(p, st, wgrad_split(base_of(64, 128), p.M, target, min_slabs));
    return launch_wgrad<T, 64, 64>(p, st, wgrad_split(base_of(64, 64), p.M, target, min_slabs));
}

static int run_conv(ConvP &p, int dtype, hipStream_t st) {
    if (dtype == DASS_F32) return dispatch_conv<float>(p, st);
    if (dtype == DASS_F32X3) return dispatch_conv<float, 2>(p, st);
    if (dtype == DASS_F32X6) return dispatch_conv<float, 3>(p, st);
    return dispatch_conv<bf16_t>(p, st);
}

}  // namespace

static int conv_entry(const void *x, int64_t ldx, const void *w, void *y, int64_t ldy, const float *scale,
                      const float *shift, const void *residual, int64_t ldr, const float *in_scale, int N,
                      int H, int W, int C, int OH, int OW, int K, int R, int S, int stride, int pad,
                      int dil, int ustride, int act, int dtype, void *stream, float *stat_partial, int *stat_rows,
                      double *stat_sums = nullptr) {
    if (!x || !w || !y) return DASS_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || K <= 0 || R <= 0 || S <= 0) return DASS_ERR_ARG;
    if (R * S > 64 || stride < 1 || dil < 1 || ustride < 1) return DASS_ERR_ARG;
    if (dtype != DASS_F32 && dtype != DASS_BF16 && dtype != DASS_F32X3 && dtype != DASS_F32X6) return DASS_ERR_UNSUPPORTED;
    const int epv = dtype == DASS_BF16 ? 8 : 4;
    if (C % epv != 0 || ldx % epv != 0 || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return DASS_ERR_ARG;
    if ((long)N * OH * OW >= (1l << 31)) return DASS_ERR_ARG;
    ConvP p;
    p.x = (const char *)x;
    p.w = (const char *)w;
    p.y = (char *)y;
    p.scale = scale;
    p.shift = shift;
    p.res = (const char *)residual;
    p.in_scale = in_scale;
    p.stat_partial = stat_partial;
    p.stat_sums = stat_sums;
    p.ldx = ldx;
    p.ldy = ldy;
    p.ldr = ldr;
    p.wk_stride = (long)R * S * C;
    p.N = N; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.K = K; p.R = R; p.S = S;
    p.stride = stride; p.pad = pad; p.dil = dil; p.ustride = ustride; p.act = act;
    p.M = N * OH * OW;
    p.OHs = OH; p.OWs = OW; p.o_mul = 1; p.oy_add = 0; p.ox_add = 0;
    p.tap_allow = ~0ull;
    hipStream_t st = (hipStream_t)stream;
    if (ustride > 1 && stride == 1 && !scale && !shift && !residual && act == DASS_ACT_NONE) {
        // dgrad of a strided conv: output pixel (oy,ox) only sees taps with (oy - pad + r*dil) % ustride == 0.
        // One launch per output phase (oy%us, ox%us) with exactly its taps -- no work on structural zeros.
        bool any_empty = false;
        unsigned long long masks[8][8];
        if (ustride > 8) return DASS_ERR_UNSUPPORTED;
        for (int py = 0; py < ustride; ++py)
            for (int px = 0; px < ustride; ++px) {
                unsigned long long mk = 0ull;
                for (int r = 0; r < R; ++r)
                    for (int s2 = 0; s2 < S; ++s2) {
                        const int ty = py - pad + r * dil, tx = px - pad + s2 * dil;
                        if (((ty % ustride) + ustride) % ustride == 0 && ((tx % ustride) + ustride) % ustride == 0)
                            mk |= 1ull << (r * S + s2);
                    }
                masks[py][px] = mk;
                if (!mk && py < OH && px < OW) any_empty = true;
            }
        if (any_empty) {
            const size_t es = dtype == DASS_BF16 ? 2 : 4;
            if (ldy != K) {  // strided rows: zero row by row is not needed on this path (dgrad outputs are dense)
                return DASS_ERR_UNSUPPORTED;
            }
            if (hipMemsetAsync(y, 0, es * (size_t)N * OH * OW * K, st) != hipSuccess) return DASS_ERR_LAUNCH;
        }
        for (int py = 0; py < ustride; ++py)
            for (int px = 0; px < ustride; ++px) {
                if (!masks[py][px] || py >= OH || px >= OW) continue;
                ConvP q = p;
                q.OHs = (OH - py + ustride - 1) / ustride;
                q.OWs = (OW - px + ustride - 1) / ustride;
                q.o_mul = ustride; q.oy_add = py; q.ox_add = px;
                q.tap_allow = masks[py][px];
                q.M = N * q.OHs * q.OWs;
                const int rc = run_conv(q, dtype, st);
                if (rc != DASS_OK) return rc;
            }
        return DASS_OK;
    }
    if (ustride > 1) return DASS_ERR_UNSUPPORTED;  // transposed addressing exists only in the phase-decomposed form
    const int rc = run_conv(p, dtype, st);
    if (stat_rows) *stat_rows = p.mtiles;
    return rc;
}

extern "C" int dass_conv2d_igemm(const void *x, int64_t ldx, const void *w, void *y, int64_t ldy, const float *scale,
                                 const float *shift, const void *residual, int64_t ldr, const float *in_scale, int N,
                                 int H, int W, int C, int OH, int OW, int K, int R, int S, int stride, int pad,
                                 int dil, int ustride, int act, int dtype, void *stream) {
    return conv_entry(x, ldx, w, y, ldy, scale, shift, residual, ldr, in_scale, N, H, W, C, OH, OW, K, R, S, stride, pad, dil,
                      ustride, act, dtype, stream, nullptr, nullptr);
}

extern "C" int dass_conv2d_igemm_stats_rows(int64_t M) { return (int)((M + 63) / 64); }

extern "C" int dass_conv2d_igemm_stats(const void *x, int64_t ldx, const void *w, void *y, int64_t ldy, int N, int H, int W,
                                       int C, int OH, int OW, int K, int R, int S, int stride, int pad, int dil, int dtype,
                                       float *stat_partial, int *stat_rows, void *stream) {
    if (!stat_partial || !stat_rows) return DASS_ERR_ARG;
    return conv_entry(x, ldx, w, y, ldy, nullptr, nullptr, nullptr, 0, nullptr, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, 1,
                      DASS_ACT_NONE, dtype, stream, stat_partial, stat_rows);
}

/* the same conv with the batch statistics added into [2][K] f64 accumulators (zeroed by the caller): see dass_bn_apply_train */
extern "C" int dass_conv2d_igemm_sums(const void *x, int64_t ldx, const void *w, void *y, int64_t ldy, int N, int H, int W, int C,
                                      int OH, int OW, int K, int R, int S, int stride, int pad, int dil, int dtype, double *stat_sums,
                                      void *stream) {
    if (!stat_sums) return DASS_ERR_ARG;
    return conv_entry(x, ldx, w, y, ldy, nullptr, nullptr, nullptr, 0, nullptr, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, 1,
                      DASS_ACT_NONE, dtype, stream, nullptr, nullptr, stat_sums);
}

static int wgrad_entry(const void *x, int64_t ldx, const void *dy, int64_t lddy, float *dw, int N, int H, int W, int C, int OH, int OW, int K, int R,
                       int S, int stride, int pad, int dil, int dtype, void *stream, bool zero_first) {
    if (!x || !dy || !dw) return DASS_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || K <= 0 || R <= 0 || S <= 0) return DASS_ERR_ARG;
    if (dtype != DASS_F32 && dtype != DASS_BF16 && dtype != DASS_F32X3 && dtype != DASS_F32X6) return DASS_ERR_UNSUPPORTED;
    if (C % 4 != 0 || K % 4 != 0 || ldx % 4 != 0 || lddy % 4 != 0) return DASS_ERR_ARG;
    if ((long)N * OH * OW >= (1l << 31)) return DASS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (zero_first && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)K * R * S * C, st) != hipSuccess) return DASS_ERR_LAUNCH;
    WgradP p;
    p.x = (const char *)x;
    p.dy = (const char *)dy;
    p.dw = dw;
    p.ldx = ldx;
    p.lddy = lddy;
    p.N = N; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.K = K; p.R = R; p.S = S;
    p.stride = stride; p.pad = pad; p.dil = dil;
    p.M = N * OH * OW;
    if (dtype == DASS_BF16 && C % 8 == 0 && K % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0) return dispatch_wgrad_bf16<0>(p, st);
    if (dtype == DASS_F32X3) return dispatch_wgrad_bf16<2>(p, st);
    if (dtype == DASS_F32X6) return dispatch_wgrad_bf16<3>(p, st);
    return dtype == DASS_F32 ? dispatch_wgrad<float>(p, st) : dispatch_wgrad<bf16_t>(p, st);
}

extern "C" int dass_conv2d_wgrad(const void *x, int64_t ldx, const void *dy, int64_t lddy, float *dw, int N, int H, int W, int C, int OH, int OW, int K, int R,
                                 int S, int stride, int pad, int dil, int dtype, void *stream) {
    return wgrad_entry(x, ldx, dy, lddy, dw, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, dtype, stream, true);
}

extern "C" int dass_conv2d_wgrad_acc(const void *x, int64_t ldx, const void *dy, int64_t lddy, float *dw, int N, int H, int W, int C, int OH, int OW, int K,
                                     int R, int S, int stride, int pad, int dil, int dtype, void *stream) {
    return wgrad_entry(x, ldx, dy, lddy, dw, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, dtype, stream, false);
}

extern "C" int dass_weight_transform(const float *src, void *dst, int K, int R, int S, int Csrc, int Cdst, int mode,
                                     int dtype, void *stream) {
    if (!src || !dst || K <= 0 || R <= 0 || S <= 0 || Csrc <= 0 || Cdst <= 0) return DASS_ERR_ARG;
    if (mode == 1 && Csrc != Cdst) return DASS_ERR_ARG;
    if (mode != 0 && mode != 1) return DASS_ERR_ARG;
    const long total = (long)K * R * S * Cdst;
    const int grid = dass_grid_1d(total, 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DASS_F32)
        DASS_LAUNCH(weight_transform_kernel<float>, dim3(grid), dim3(256), 0, st, src, (float *)dst, K, R, S, Csrc, Cdst, mode);
    else if (dtype == DASS_BF16)
        DASS_LAUNCH(weight_transform_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, src, (bf16_t *)dst, K, R, S, Csrc, Cdst, mode);
    else if (dtype == DASS_F32X6) {
        const long n6 = (long)(mode == 0 ? K : Csrc) * R * S * (((mode == 0 ? Cdst : K) + 31) / 32) * 32;
        DASS_LAUNCH(weight_split3_kernel, dim3(dass_grid_1d(n6, 256)), dim3(256), 0, st, src, (bf16_t *)dst, K, R, S, Csrc, Cdst, mode);
    } else
        return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_weight_split_batch(const void *desc, const int64_t *start, int n, int64_t total, void *stream) {
    if (!desc || !start || n <= 0 || total <= 0) return DASS_ERR_ARG;
    const long grid = total < 256 * 32 ? total : 256 * 32;
    DASS_LAUNCH(weight_split3_batch_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const long *)desc, (const long *)start, n, (long)total, 3);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

/* the DASS_BF16X1 form (one bf16 part per weight: the "bf16x1" perf engine of the pre-split kernels, dass_set_x3_parts(1)) */
extern "C" int dass_weight_split_batch_bf16(const void *desc, const int64_t *start, int n, int64_t total, void *stream) {
    if (!desc || !start || n <= 0 || total <= 0) return DASS_ERR_ARG;
    const long grid = total < 256 * 32 ? total : 256 * 32;
    DASS_LAUNCH(weight_split3_batch_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const long *)desc, (const long *)start, n, (long)total, 1);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int64_t dass_weight_split_bytes(int rows, int R, int S, int red) {
    return (int64_t)rows * R * S * ((red + 31) / 32) * 192 + 16;  // (+ the trailer every pre-split operand carries)
}
/* bytes of the pre-split operand of `dtype` (DASS_F32X6: three bf16 parts; DASS_F16X3: two scaled f16 parts + scale trailer) */
extern "C" int64_t dass_weight_operand_bytes(int rows, int R, int S, int red, int dtype) {
    if (dtype == DASS_F16X3) return (int64_t)rows * R * S * ((red + 31) / 32) * 128 + 16;
    if (dtype == DASS_BF16X1) return (int64_t)rows * R * S * ((red + 31) / 32) * 64 + 16;
    return dass_weight_split_bytes(rows, R, S, red);
}

/* the DASS_F16X3 form of dass_weight_split_batch: same table; three launches (zero the trailers, max |w| per tensor, split) */
extern "C" int dass_weight_split_batch_f16(const void *desc, const int64_t *start, int n, int64_t total, void *stream) {
    if (!desc || !start || n <= 0 || total <= 0) return DASS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long grid = total < 256 * 32 ? total : 256 * 32;
    DASS_LAUNCH(weight_trailer_zero_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const long *)desc, n);
    DASS_LAUNCH(weight_split2_batch_kernel, dim3((unsigned)grid), dim3(256), 0, st, (const long *)desc, (const long *)start, n, (long)total, 1);
    DASS_LAUNCH(weight_split2_batch_kernel, dim3((unsigned)grid), dim3(256), 0, st, (const long *)desc, (const long *)start, n, (long)total, 0);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_conv2d_rowtap(const void *x, const void *w, void *y, int64_t ldy, int N, int H, int W, int Cin, int OH,
                                  int OW, int K, int R, int S, int stride, int pad, int dtype, void *stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || K <= 0 || R <= 0 || S <= 0)
        return DASS_ERR_ARG;
    if (dtype != DASS_F32) return DASS_ERR_UNSUPPORTED;
    if (R > 64 || S * Cin > 32 || stride < 1 || (long)N * OH * OW >= (1l << 31)) return DASS_ERR_ARG;
    {   // the stems of the three backbones: exact f32 MFMAs, one predicated load per window value (stem_rowtap.hip)
        const int fast = dass_rowtap_fwd_fast((const float *)x, (const float *)w, (float *)y, (long)ldy, N, H, W, Cin, OH, OW, K, R, S, stride, pad,
                                              (hipStream_t)stream);
        if (fast) return fast > 0 ? DASS_OK : DASS_ERR_LAUNCH;
    }
    ConvP p;
    p.x = (const char *)x; p.w = (const char *)w; p.y = (char *)y;
    p.scale = nullptr; p.shift = nullptr; p.res = nullptr; p.in_scale = nullptr; p.stat_partial = nullptr; p.stat_sums = nullptr;
    p.ldx = Cin; p.ldy = ldy; p.ldr = 0;
    p.wk_stride = (long)R * S * Cin;
    p.N = N; p.H = H; p.W = W; p.C = S * Cin; p.OH = OH; p.OW = OW; p.K = K; p.R = R; p.S = 1;
    p.stride = stride; p.pad = pad; p.dil = 1; p.ustride = 1; p.act = DASS_ACT_NONE;
    p.M = N * OH * OW;
    p.OHs = OH; p.OWs = OW; p.o_mul = 1; p.oy_add = 0; p.ox_add = 0;
    p.tap_allow = ~0ull;
    p.wide_c = Cin;
    p.cchunks = 1;
    hipStream_t st = (hipStream_t)stream;
    if (K <= 32) return launch_conv<float, 128, 32, 4, 1, true>(p, st);
    return launch_conv<float, 128, 64, 2, 2, true>(p, st);
}

extern "C" int dass_conv2d_rowtap_wgrad(const void *x, const void *dy, int64_t lddy, float *dw, int N, int H, int W, int Cin,
                                        int OH, int OW, int K, int R, int S, int stride, int pad, int dtype, void *stream) {
    if (!x || !dy || !dw || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || K <= 0 || R <= 0 || S <= 0)
        return DASS_ERR_ARG;
    if (dtype != DASS_F32) return DASS_ERR_UNSUPPORTED;
    if (S * Cin > 64 || K % 4 != 0 || lddy % 4 != 0 || (long)N * OH * OW >= (1l << 31)) return DASS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dw, 0, sizeof(float) * (size_t)K * R * S * Cin, st) != hipSuccess) return DASS_ERR_LAUNCH;
    {   // whole K x (R S Cin) gradient per wave, dy streamed once (stem_rowtap.hip)
        const int fast = dass_rowtap_wgrad_fast((const float *)x, (const float *)dy, (long)lddy, dw, N, H, W, Cin, OH, OW, K, R, S, stride, pad, st);
        if (fast) return fast > 0 ? DASS_OK : DASS_ERR_LAUNCH;
    }
    WgradP p;
    p.x = (const char *)x; p.dy = (const char *)dy; p.dw = dw;
    p.ldx = Cin; p.lddy = lddy;
    p.N = N; p.H = H; p.W = W; p.C = S * Cin; p.OH = OH; p.OW = OW; p.K = K; p.R = R; p.S = 1;
    p.stride = stride; p.pad = pad; p.dil = 1;
    p.M = N * OH * OW;
    p.wide_c = Cin;
    const long base = (long)((K + 63) / 64) * R;
    return launch_wgrad<float, 64, 64, true>(p, st, wgrad_split(base, p.M, 640, 16));
}
