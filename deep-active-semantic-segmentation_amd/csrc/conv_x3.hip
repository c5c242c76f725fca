// Pipelined implicit-GEMM convolution on PRE-SPLIT operands ("x3" rows) for gfx950.
//
// The parity engine multiplies f32 tensors exactly on the bf16 matrix pipe: x = x0 + x1 + x2 (three bf16 parts), six
// products per pair.  conv_igemm.hip converts the activation operand inside its MFMA loop (3.4 VALU per MFMA, two
// barriers per 32-k slab, register staging).  Here BOTH operands arrive already split -- activations as
//   x3[pixel][C/32][3 parts][32 ch] bf16   (192 B per pixel and 32-channel slab; row `rows` is an all-zero row)
// written by the producing pass (dass_split3_rows, the BN-apply / BN-backward kernels), weights as the existing
//   w3[k][tap][C/32][3][32]                (dass_weight_transform(DASS_F32X6) / dass_weight_split_batch) --
// so the main loop is a pure copy + MFMA pipeline:
//   * global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, no VGPR staging, no VALU
//     conversion); the implicit-GEMM gather and the zero padding live in the per-lane SOURCE offset (a padded tap
//     reads the zero row), the bank swizzle too (the LDS destination of a DMA is lane-linear);
//   * an NSTAGE-deep LDS ring with ONE s_barrier per 32-k slab and counted s_waitcnt vmcnt(N): the loads of the next
//     NSTAGE-1 slabs stay in flight under the MFMAs (cdna_hip_programming.md 5, "Pipelining across barriers");
//   * 16-row x 64-B pieces, XOR-swizzled at 16-B granularity: every ds_read_b128 of a 32x32x16 fragment is
//     conflict-free, every DMA instruction fetches 16 rows x 64 contiguous bytes.
// The epilogue is the one of conv_igemm.hip (accumulators bounced through wave-private LDS patches, 16-B stores, fused
// scale/shift/residual/activation, per-tile BatchNorm partial sums) and can emit the result as x3 rows as well.
// dgrad = the same kernel over dy (x3) and the transposed weight operand, phase-decomposed for strided convs.
//
// Reference sites replaced: every dense nn.Conv2d of the DeepLab path with C % 32 handled by zero padding
// (models/backbone/resnet.py:11-15, models/aspp.py:13-14,57-68, models/decoder.py:23-36).
#include "dass_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct X3P {
    const char *x3;
    const char *w3;
    char *y;    // f32 rows [M][ldy] (nullable when y3 is set)
    char *y3;   // optional: the result as x3 rows [M + 1][cc_out][192] (row M zeroed)
    const float *scale;
    const float *shift;
    const char *res;
    float *stat_partial;
    double *stat_sums;  // alternative to stat_partial: [2][K] f64 accumulators (hardware atomics), see dass_bn_apply_train
    float *ws;  // stream-K workspace: 2 slabs of BM x BN f32 per workgroup of the main launch
    long ldy, ldr;
    unsigned x3_bytes, w3_bytes, zero_off, row_pitch;  // row_pitch = CC * (parts * 64)
    unsigned x3_tr, w3_tr;  // byte offsets of the operands' trailers {inv_scale, bound, amax} (two-part format)
    unsigned *y_amax;       // optional: atomic max of |output| (bit pattern of a non-negative float); for a two-part y3 its trailer[2]
    int y3_parts;           // format of y3: 3, or 2 (its trailer {inv_scale, bound} was written by dass_x3_prepare_out BEFORE this launch)
    int cc_out;
    int N, H, W, CC, OH, OW, K, R, S, stride, pad, dil, act;
    int M, mtiles, ntiles, sk_wgs;
    int dp_tiles, sk_part;          // tiles [0, dp_tiles) go one per workgroup and round; the rest is cut into sk_part slab ranges
    int group_rows, mt_per_group;   // per-image mode: M-tiles never straddle an image; group_rows = M otherwise
    unsigned w3_group_stride;       // per-image mode: byte offset between the weight operands of consecutive images
    const int *cc_limit;            // per-image mode: channel slabs of image g that hold anything (the others are skipped)
    int res_groups;                 // per-image mode, > 0: the residual holds only res_groups images -- image g adds the rows of image g % res_groups
                                    // (all T stochastic passes of a scoring batch in ONE launch share the batch's deterministic residual)
    int OHs, OWs, o_mul, oy_add, ox_add, ustride;
    unsigned long long tap_allow;
    // filled by launch_x3 (host-side divisions the kernel's per-workgroup setup would otherwise repeat):
    unsigned mg_ohw, mg_ows;  // magic multipliers of the divisions by OHs * OWs and by OWs (x3_fastdiv)
    int sh_ohw, sh_ows;
    int sk_q, sk_r;           // stream-K region: U = sk_q * sk_part + sk_r units; workgroup w owns [w q + min(w, r), ...) (q + 1 units while w < r)
    int dp_rounds;            // dp_tiles / sk_wgs
    int whole;                // 1: one whole tile per workgroup, nothing else (no stream-K region arithmetic at all)
    // dass_conv2d_x3_dgrad_bnstats: this launch's output is the gradient d_out of a conv + BN (+ act) layer whose conv output
    // is bs_y [M][K]; the epilogue also adds that layer's BN-backward sums (sum dz, sum dz * xhat; dz = d_out * activation gate)
    // into bs_sums[2][K] f64 and maxes |dz| per channel into the K floats behind them (what dass_bn_bwd_reduce_sums computes)
    const float *bs_y, *bs_mean, *bs_invstd, *bs_gsc, *bs_gsh;
    const unsigned char *bs_gates;  // [M][K / 4] gate bits (residual layers), else the gate is re-derived from bs_y * gsc + gsh
    double *bs_sums;
    int bs_act;
};

// (x3_fastdiv / x3_set_magic: dass_common.h)
__device__ __forceinline__ int x3_sk_bound(const X3P &p, int w) { return w * p.sk_q + (w < p.sk_r ? w : p.sk_r); }

typedef int v4i __attribute__((ext_vector_type(4)));

// One LDS-DMA instruction: lane l's 16 bytes at (buffer base + voff) land at LDS byte address lds + 16 * l.
// Issued through inline asm ON PURPOSE: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first ds_read that
// follows a __builtin_amdgcn_raw_ptr_buffer_load_lds it cannot prove disjoint -- every ring read here -- which drains the
// pipeline each slab (seen in the .s of the builtin version).  Hidden from the compiler, the DMAs are counted by hand
// (wait_vmcnt below); M0 (the DMA's LDS base) is saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void dma16(v4i rsrc, unsigned lds, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds), "s"(rsrc)
                 : "memory");
}

// The NP pieces of ONE rowgroup (consecutive 1 KiB pieces in LDS, 64 B apart in memory) behind ONE write of M0: the instruction's
// immediate offset is added to the LDS address (M0 + offset + 16 * lane) as well as to the memory address (base + voffset +
// offset), so piece i goes out with offset:1024 i and a per-lane voffset that carries its true source minus 1024 i.  For that
// difference never to wrap, every descriptor of this kernel starts X3_SRD_BIAS bytes BELOW its operand and every per-lane offset
// carries + X3_SRD_BIAS (folded into chunk_off).  M0 is not restored: nothing the compiler emits for these kernels reads it (gfx9+ LDS
// instructions do not), and every statement that needs it writes it itself -- as csrc/wgrad_x3.hip:wdma16x4 (round 4: 330 -> 111
// instructions per slab there; here 6 SALU instructions less per rowgroup and slab).
constexpr unsigned X3_SRD_BIAS = 2048;
template <int NP> __device__ __forceinline__ void dma16_group(v4i rsrc, unsigned lds, unsigned voff) {
    if constexpr (NP == 1) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(lds), "s"(rsrc) : "memory");
    } else if constexpr (NP == 2) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %0, %3, 0 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %3, 0 offen offset:1024 lds"
                     :
                     : "v"(voff), "v"(voff + 64u - 1024u), "s"(lds), "s"(rsrc)
                     : "memory");
    } else {
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %0, %4, 0 offen lds\n\t"
                     "buffer_load_dwordx4 %1, %4, 0 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %2, %4, 0 offen offset:2048 lds"
                     :
                     : "v"(voff), "v"(voff + 64u - 1024u), "v"(voff + 128u - 2048u), "s"(lds), "s"(rsrc)
                     : "memory");
    }
}

__device__ __forceinline__ v4i make_srd(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    v4i r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));  // stride 0
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);                              // num_records (bytes)
    r[3] = 0x00020000;                                                              // raw buffer, 32-bit data format
    return r;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NP = 3: bf16 parts, NP = 2: f16 parts (dass_common.h "x3 operand formats")
template <int NP> __device__ __forceinline__ f32x16 mfma16(const uint4 &a, const uint4 &b, f32x16 c) {
    if constexpr (NP == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&a), *reinterpret_cast<const f16x8 *>(&b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a), *reinterpret_cast<const bf16x8 *>(&b), c, 0, 0, 0);
}
template <int NP> __device__ __forceinline__ f32x4 mfma32(const uint4 &a, const uint4 &b, f32x4 c) {
    if constexpr (NP == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8 *>(&a), *reinterpret_cast<const f16x8 *>(&b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&a), *reinterpret_cast<const bf16x8 *>(&b), c, 0, 0, 0);
}
// the products of one operand pair, smallest first.  NP = 3: (a0,b2) (a2,b0) (a1,b1) (a0,b1) (a1,b0) (a0,b0); NP = 2: (a0,b1) (a1,b0) (a0,b0)
template <int NP> struct Terms;
template <> struct Terms<3> { static constexpr int N = 6; static constexpr int PA[6] = {0, 2, 1, 0, 1, 0}; static constexpr int PB[6] = {2, 0, 1, 1, 0, 0}; };
template <> struct Terms<2> { static constexpr int N = 3; static constexpr int PA[3] = {0, 1, 0}; static constexpr int PB[3] = {1, 0, 0}; };
template <> struct Terms<1> { static constexpr int N = 1; static constexpr int PA[1] = {0}; static constexpr int PB[1] = {0}; };  // "bf16x1": one bf16 part, one product

// fused epilogue of ONE 4-channel group of output pixel m (tile-independent: used by the main kernel and the fix-up pass)
// (pre: the caller already holds this lane's 4 scale / shift values and its residual in sc4 / sh4 / r4 -- loaded once per pass of
// the epilogue, before the accumulators bounce through LDS -- instead of three dependent loads per 16-byte store)
__device__ __forceinline__ void x3_store_out(const X3P &p, f32x4 v, int m, int k, int ohw, bool vec_ok, float y3_scale, float &vmax, bool phase,
                                             bool pre = false, f32x4 sc4 = f32x4{1.f, 1.f, 1.f, 1.f}, f32x4 sh4 = f32x4{0.f, 0.f, 0.f, 0.f},
                                             f32x4 r4 = f32x4{0.f, 0.f, 0.f, 0.f}, long res_shift = 0) {
    float *y = reinterpret_cast<float *>(p.y);
    const float *res = reinterpret_cast<const float *>(p.res);
    long mo = m;  // output pixel index; differs from m only for a phase sub-grid
    if (phase) {
        const int n = m / ohw, rem = m - n * ohw;
        const int ohs = rem / p.OWs;
        mo = ((long)n * p.OH + ohs * p.o_mul + p.oy_add) * p.OW + (rem - ohs * p.OWs) * p.o_mul + p.ox_add;
    }
    if (vec_ok) {
        if (pre) {
            v = v * sc4 + sh4 + r4;  // (same operation order as below: multiply, add shift, add residual, each rounded)
        } else {
            if (p.scale) v *= *reinterpret_cast<const f32x4 *>(p.scale + k);
            if (p.shift) v += *reinterpret_cast<const f32x4 *>(p.shift + k);
            if (res) v += *reinterpret_cast<const f32x4 *>(res + (mo + res_shift) * p.ldr + k);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
        if (y) *reinterpret_cast<f32x4 *>(y + mo * p.ldy + k) = v;
        if (p.y3) x3_store4r(p.y3, mo, p.cc_out, k, v, p.y3_parts, y3_scale);
#pragma unroll
        for (int e = 0; e < 4; ++e) vmax = fmaxf(vmax, fabsf(v[e]));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (k + e >= p.K) break;
            float u = v[e] * (p.scale ? p.scale[k + e] : 1.f) + (p.shift ? p.shift[k + e] : 0.f);
            if (res) u += res[(mo + res_shift) * p.ldr + k + e];
            u = apply_act(u, p.act);
            y[mo * p.ldy + k + e] = u;
            vmax = fmaxf(vmax, fabsf(u));
        }
    }
}
// one atomic max per wave of the largest |output| its lanes stored
__device__ __forceinline__ void x3_amax_commit(unsigned *slot, float vmax) {
    if (!slot) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((threadIdx.x & 63) == 0) {
        if (!(vmax >= 0.f)) vmax = __uint_as_float(0x7f800000u);
        if (__float_as_uint(vmax) > *reinterpret_cast<volatile unsigned *>(slot)) atomicMax(slot, __float_as_uint(vmax));
    }
}

// LDS image of one operand stage: rowgroup g (16 rows), part pl -> a 1 KiB piece at (g * 3 + pl) * 1024; inside a piece
// row r16 / 16-B chunk c sits at r16 * 64 + ((c ^ ((r16 >> 2) & 3)) * 16).
//
// Work decomposition ("stream-K"): the unit of work is one 32-k slab of one output tile; the U = tiles * S units (S = CC *
// taps, tile-major) are cut into p.sk_wgs EQUAL contiguous ranges, one per workgroup.  A workgroup walks its range tile by
// tile ("segments").  A segment that covers its whole tile runs the fused epilogue; one that covers only part of it writes
// its raw accumulators to a workspace slab, and conv_x3_fixup_kernel adds the slabs of such a tile in workgroup order
// (deterministic) and runs the same epilogue.  With sk_wgs = CUs every CU gets the same number of slabs whatever the tile
// count (M = 8712 layers: 138 tiles on 256 CUs; 133128-row layers: 1042 tiles = 4.07 rounds); with sk_wgs = tiles it
// degenerates to one tile per workgroup and no workspace traffic.
// M16: the products run on v_mfma_f32_16x16x32_bf16 (one 32-k slab per instruction) instead of v_mfma_f32_32x32x16_bf16: the
// same FLOPs per cycle, but the chip holds a ~10 % higher clock under that shape (tools/mfma_shape_probe.py: 2.17 vs 1.97 GHz
// with two LDS-fed waves per SIMD on all CUs; MI355X_MICROARCH.md measures 1.12-1.14 x).
// SIMPLE: one whole tile per workgroup, no phase sub-grid, no per-image groups (what the host's `p.whole` launches of plain
// convolutions are): the generality below costs ~1000 instructions of hoisted loop invariants and SGPR spills in front of the
// first barrier (2-3 us per workgroup, tools/x3_exp.py stamps) -- more than the whole slab loop of a short reduction.
template <int BM, int BN, int WARPS_M, int WARPS_N, int NSTAGE, bool M16, int NP = 3, bool SIMPLE = false>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, 2) void conv_x3_kernel(const X3P p) {
    constexpr int NW = WARPS_M * WARPS_N;
    constexpr int SB = NP * 64, PG = NP * 1024;  // bytes of one row-slab / of the NP 1-KiB pieces of one 16-row rowgroup
    using TT = Terms<NP>;
    constexpr int TMW = BM / WARPS_M, TNW = BN / WARPS_N, MT = TMW / 32, NT = TNW / 32;
    constexpr int MT16 = TMW / 16, NT16 = TNW / 16, MH = MT16 / 2, NH = NT16 / 2;  // M16: 16 x 16 blocks, handled in halves
    constexpr int AG = BM / 16, BG = BN / 16, RG = (AG + BG) / NW;
    static_assert((AG + BG) % NW == 0, "rowgroups must divide over the waves");
    static_assert(MT >= 1 && NT >= 1 && NSTAGE >= 2 && NSTAGE <= 4, "tile");
    constexpr int A_BYTES = AG * PG, B_BYTES = BG * PG, STAGE = A_BYTES + B_BYTES;
#ifndef X3_EXP
#define X3_EXP 0  // timing experiments (results wrong), tools/x3_exp.py
#endif
    constexpr int XE = X3_EXP;
    constexpr bool XE_NO_RDB = XE == 1 || XE == 2 || XE == 3 || XE == 6 || XE == 7, XE_NO_RDA = XE == 2 || XE == 6 || XE == 7;
    constexpr bool XE_NO_DMAB = XE == 3 || XE == 5 || XE == 6 || XE == 7, XE_NO_DMAA = XE == 4 || XE == 5 || XE == 6 || XE == 7;
    constexpr int G = (XE_NO_DMAA && XE_NO_DMAB) ? 0 : (XE_NO_DMAA || XE_NO_DMAB) ? (RG / 2) * NP : RG * NP;  // DMA instructions per wave and slab
    static_assert(G * (NSTAGE - 2) <= 63, "vmcnt range");
    constexpr int RING = NSTAGE * STAGE;
    constexpr int EPI = NW * 32 * ((NT >= 2 ? 64 : 32) + 4) * 4;  // the epilogue's accumulator patches reuse the ring (two-part 64 x 64 ring is smaller)
    constexpr int SMEM = RING > EPI ? RING : EPI;
    // ONE shared object (a second one beside a DMA-filled array makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) char smem[SMEM + 64 * 4 + 16];
    unsigned *tap_delta = reinterpret_cast<unsigned *>(smem + SMEM);
    unsigned *tap_any = tap_delta + 64;  // [2]: the taps that touch the tile at all (OR over the workgroup's rows)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
    unsigned long long xt0 = 0, xt1 = 0, xt2 = 0, xt3 = 0, xt4 = 0, xa = 0, xb = 0, xc = 0;
    if (XE == 12) xt0 = wall_clock64();
    if (XE == 12 && p.R > 0) xa = wall_clock64();

    // (descriptors start X3_SRD_BIAS below their operands, every per-lane offset carries + X3_SRD_BIAS through chunk_off: dma16_group)
    const v4i rsa = make_srd(p.x3 - X3_SRD_BIAS, p.x3_bytes + X3_SRD_BIAS), rsb = make_srd(p.w3 - X3_SRD_BIAS, p.w3_bytes + X3_SRD_BIAS);
    const unsigned smem_base = (unsigned)(size_t)smem;  // LDS byte address of the ring

    // ---- loader roles: this lane fetches row r16 = lane >> 2 of each of the wave's RG rowgroups, the 16-B chunk that
    // lands (lane-linear) at slot lane & 3 of that row, i.e. source chunk (lane & 3) ^ ((r16 >> 2) & 3)
    // (M16 reads a 16-row piece with lane = row + 16 * chunk; ds_read_b128 is served in the lane groups {0-3,12-15,20-27},
    // {4-11,16-19,28-31} (+32): rows 0-3 / 12-15 of one chunk meet rows 4-11 of the neighbouring chunk, which the swizzle
    // (-(row >> 2)) & 3 keeps on distinct banks; (row >> 2) & 3 -- conflict-free for the 32x32x16 fragment -- is 2-way there)
    const int r16 = lane >> 2;
    const unsigned chunk_off = (unsigned)(((lane & 3) ^ ((M16 ? 0 - (r16 >> 2) : (r16 >> 2)) & 3)) * 16) + X3_SRD_BIAS;
    const int ntaps = p.R * p.S;
    const int ohw = p.OHs * p.OWs;
    const bool phase = SIMPLE ? false : p.o_mul != 1;
    auto tap_ey = [&](int r) -> int { return phase ? (p.oy_add - p.pad + r * p.dil) / p.ustride : r * p.dil; };
    auto tap_ex = [&](int s) -> int { return phase ? (p.ox_add - p.pad + s * p.dil) / p.ustride : s * p.dil; };
    if (tid < ntaps) {
        const int r = tid / p.S, s2 = tid - r * p.S;
        tap_delta[tid] = (unsigned)(tap_ey(r) * p.W + tap_ex(s2)) * p.row_pitch;
    }

    // ---- fragment addresses: lane (r = lane & 31, h = lane >> 5) reads row r, chunk 2 ks + h of each part
    const int fr = lane & 31, fh = lane >> 5;
    const int fsw = ((fr & 15) >> 2) & 3;
    const int f_row = (fr >> 4) * PG + (fr & 15) * 64;
    const int a_lane0 = (wm * (TMW / 16)) * PG + f_row + (((0 + fh) ^ fsw) * 16);
    const int a_lane1 = (wm * (TMW / 16)) * PG + f_row + (((2 + fh) ^ fsw) * 16);
    const int b_lane0 = A_BYTES + (wn * (TNW / 16)) * PG + f_row + (((0 + fh) ^ fsw) * 16);
    const int b_lane1 = A_BYTES + (wn * (TNW / 16)) * PG + f_row + (((2 + fh) ^ fsw) * 16);
    // M16: lane (r = lane & 15, g = lane >> 4) reads row r, chunk g (k = 8 g .. 8 g + 7) of a 16-row piece: the 64 lanes cover
    // the 1 KiB piece exactly once (conflict-free under the same XOR swizzle)
    const int l16 = (lane & 15) * 64 + (((lane >> 4) ^ ((0 - ((lane & 15) >> 2)) & 3)) * 16);
    const int a16 = (wm * MT16) * PG + l16, b16 = A_BYTES + (wn * NT16) * PG + l16;

    // ---- this workgroup's range of (tile, slab) units
    // (all of it 32-bit and division-free on the common path: this setup runs once per workgroup and tile, and on short
    // reductions it used to cost more than the slab loop -- tools/x3_exp.py stamps)
    const int wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int S_tile = p.CC * ntaps;
    int u = 0, u_end = 0;  // units [u, u_end) of the stream-K region (U < 2^31: checked by the host)
    if (SIMPLE || p.whole) {
        u_end = 1;  // one pass through the loop below
    } else if (wgid < p.sk_part) {
        u = x3_sk_bound(p, wgid);
        u_end = u + p.sk_q + (wgid < p.sk_r ? 1 : 0);
    }
    const int dp_rounds = SIMPLE ? 0 : p.dp_rounds;
    int seg_idx = 0;  // stream-K segments done so far
    for (int round = 0; SIMPLE ? round < 1 : (round < dp_rounds || u < u_end); ++round) {
    int tile, s_lo, s_hi;
    if (SIMPLE || p.whole) {
        tile = wgid;
        s_lo = 0;
        s_hi = S_tile;
        u = u_end;
    } else if (round < dp_rounds) {  // data-parallel part: whole tiles, fused epilogue, no workspace traffic
        tile = round * p.sk_wgs + wgid;
        s_lo = 0;
        s_hi = S_tile;
    } else {
        const int rt = (int)((unsigned)u / (unsigned)S_tile);
        tile = p.dp_tiles + rt;
        s_lo = u - rt * S_tile;
        s_hi = (u_end - u) < (S_tile - s_lo) ? s_lo + (u_end - u) : S_tile;
        u += s_hi - s_lo;
    }
    const bool complete = SIMPLE ? true : (s_lo == 0 && s_hi == S_tile);
    const int sk_seg = SIMPLE || round < dp_rounds ? 0 : seg_idx++;
    const int mt_i = p.ntiles == 1 ? tile : (int)((unsigned)tile / (unsigned)p.ntiles), nt_i = tile - mt_i * p.ntiles;
    const int grp = p.mt_per_group >= p.mtiles ? 0 : (int)((unsigned)mt_i / (unsigned)p.mt_per_group);
    const int m0 = grp * p.group_rows + (mt_i - grp * p.mt_per_group) * BM, n0 = nt_i * BN;
    const int m_end = (grp + 1) * p.group_rows < p.M ? (grp + 1) * p.group_rows : p.M;
    if (p.cc_limit) {  // slabs are channel-slab major: dropping the slabs >= limit truncates the range
        const int lim = p.cc_limit[grp] * ntaps;
        s_lo = s_lo < lim ? s_lo : lim;
        s_hi = s_hi < lim ? s_hi : lim;
    }
    if (tid < 2) tap_any[tid] = 0u;
    if (XE == 12 && round == 0 && lane == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(p.ws) + (long)wgid * 16;
        o[8 + wave] = xt0;
        o[12 + wave] = wall_clock64();
    }
    __syncthreads();  // the previous segment's epilogue is done with the ring; tap_delta is visible
    if (XE == 12 && round == 0) xb = wall_clock64();

    unsigned rb_off[RG];
    unsigned long long vmask[RG];
    int a_by[RG], a_bx[RG];
    bool a_ok[RG];
#pragma unroll
    for (int j = 0; j < RG; ++j) {
        const int q = wave + j * NW;
        vmask[j] = 0ull;
        a_by[j] = a_bx[j] = 0;
        a_ok[j] = false;
        rb_off[j] = 0u;
        if (q < AG) {
            const int m = m0 + q * 16 + r16;
            a_ok[j] = m < m_end;
            const int mm = a_ok[j] ? m : 0;
            const int n = x3_fastdiv(mm, p.mg_ohw, p.sh_ohw);
            const int rem = mm - n * ohw;
            const int ohs = x3_fastdiv(rem, p.mg_ows, p.sh_ows);
            const int ows = rem - ohs * p.OWs;
            a_by[j] = phase ? ohs : ohs * p.stride - p.pad;
            a_bx[j] = phase ? ows : ows * p.stride - p.pad;
            // (mod 2^32 throughout: wraps for border rows; only valid taps use it, and for those the true offset is < 2^32)
            rb_off[j] = ((unsigned)(n * p.H + a_by[j]) * (unsigned)p.W + (unsigned)a_bx[j]) * p.row_pitch + chunk_off;
        } else {
            int k = n0 + (q - AG) * 16 + r16;
            if (k >= p.K) k = p.K - 1;  // columns >= K are never stored: any finite row will do
            rb_off[j] = (unsigned)grp * p.w3_group_stride + (unsigned)k * (unsigned)(ntaps * p.CC * SB) + chunk_off;
        }
    }

    // ---- taps that touch this tile at all; per A row the taps that fall inside the image
    if (XE == 12 && round == 0) xc = wall_clock64();
    unsigned long long tapmask = 0ull;
    if (ntaps == 1 && p.pad == 0 && !phase) {  // 1 x 1, no padding: every row of the tile reads its one tap
#pragma unroll
        for (int j = 0; j < RG; ++j) vmask[j] = a_ok[j] ? 1ull : 0ull;
        tapmask = p.tap_allow & 1ull;
    } else {
        unsigned long long wave_any = 0ull;  // wave-uniform
        int t = 0;
        for (int r = 0; r < p.R; ++r) {
            const int ey = tap_ey(r);
            for (int s2 = 0; s2 < p.S; ++s2, ++t) {
                if (!((p.tap_allow >> t) & 1ull)) continue;
                const int ex = tap_ex(s2);
                bool any = false;
#pragma unroll
                for (int j = 0; j < RG; ++j) {
                    const int iy = a_by[j] + ey, ix = a_bx[j] + ex;
                    const bool ok = a_ok[j] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    if (ok) vmask[j] |= (1ull << t);
                    any = any || ok;
                }
                if (__ballot(any)) wave_any |= (1ull << t);
            }
        }
        // ONE barrier for the whole mask (it was one __syncthreads_or per tap)
        if (lane == 0) {
            if ((unsigned)wave_any) atomicOr(&tap_any[0], (unsigned)wave_any);
            if ((unsigned)(wave_any >> 32)) atomicOr(&tap_any[1], (unsigned)(wave_any >> 32));
        }
        __syncthreads();
        tapmask = (unsigned long long)tap_any[0] | ((unsigned long long)tap_any[1] << 32);
    }

    // active slabs of [s_lo, s_hi): slab s = cc * ntaps + t (channel slab outer, taps inner)
    const int pc = __builtin_popcountll(tapmask);
    auto active_below = [&](int sidx) -> int {
        const int cc = sidx / ntaps, t = sidx - cc * ntaps;
        return cc * pc + __builtin_popcountll(tapmask & ((1ull << t) - 1ull));
    };
    const int total = XE == 8 ? 1 : (s_lo == 0 && s_hi == S_tile) ? p.CC * pc : active_below(s_hi) - active_below(s_lo);

    f32x16 acc[M16 ? 1 : MT][M16 ? 1 : NT];
    f32x4 acc16[M16 ? MT16 : 1][M16 ? NT16 : 1];
#pragma unroll
    for (int i = 0; i < (M16 ? 1 : MT); ++i)
#pragma unroll
        for (int j = 0; j < (M16 ? 1 : NT); ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < (M16 ? MT16 : 1); ++i)
#pragma unroll
        for (int j = 0; j < (M16 ? NT16 : 1); ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- slab iterator of the LOADER (the 9 taps of a 3x3 re-read the same neighbourhood back to back, so the re-reads
    // are L2 hits).  The uniform offsets of the NEXT slab to issue are computed one slab ahead (the tap table lives in
    // LDS: its read must not sit in front of a DMA issue).
    int it_cc = s_lo == 0 ? 0 : s_lo / ntaps, nx_t = 0;
    unsigned long long it_mask = tapmask & ~((1ull << (s_lo - it_cc * ntaps)) - 1ull);
    unsigned nx_a_uni = 0u, nx_b_uni = 0u;
    auto advance = [&]() __attribute__((always_inline)) {
        if (!it_mask) {
            ++it_cc;
            it_mask = tapmask;
        }
        nx_t = __builtin_ctzll(it_mask);
        it_mask &= it_mask - 1;
        nx_a_uni = tap_delta[nx_t] + (unsigned)(it_cc * SB);
        nx_b_uni = (unsigned)((nx_t * p.CC + it_cc) * SB);
    };
    auto issue_rowgroup = [&](auto jc, int stage) __attribute__((always_inline)) {  // the three parts of rowgroup wave + j * NW of the next slab (j: int or integral_constant)
        const int j = jc;
        const unsigned st = smem_base + (unsigned)(stage * STAGE);
        const int q = wave + j * NW;
        if (q < AG) {
            const bool valid = (vmask[j] >> nx_t) & 1ull;
            const unsigned voff = valid ? rb_off[j] + nx_a_uni : p.zero_off + chunk_off;
            const unsigned dst = __builtin_amdgcn_readfirstlane(st + q * PG);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) dma16(rsa, dst + pl * 1024, voff + pl * 64);
        } else {
            const unsigned voff = rb_off[j] + nx_b_uni;
            const unsigned dst = __builtin_amdgcn_readfirstlane(st + A_BYTES + (q - AG) * PG);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) dma16(rsb, dst + pl * 1024, voff + pl * 64);
        }
    };

    // ---- prologue: NSTAGE - 1 slabs in flight
    if (XE == 12 && round == 0) xt1 = wall_clock64();
    int issued = 0;
    if (total > 0) advance();
#pragma unroll
    for (int sg = 0; sg < NSTAGE - 1; ++sg)
        if (issued < total) {
#pragma unroll
            for (int j = 0; j < RG; ++j) issue_rowgroup(j, sg);
            if (++issued < total) advance();
        }

    // ---- main loop, skewed by half a slab: the fragments of slab s / k-step 0 are read while the MFMAs of slab s-1 /
    // k-step 1 (operands already in registers) run, and those of k-step 1 while k-step 0 multiplies, so the LDS read
    // latency is covered; the DMA issue of the slab NSTAGE-1 ahead is cut into its rowgroups and placed BETWEEN the
    // MFMA groups of the first half (an in-order wave issues them in the shadow of the running MFMAs).
    if constexpr (!M16) {
    uint4 a0[NP][MT], b0[NP][NT], a1[NP][MT], b1[NP][NT];
    auto load_frags = [&](const char *ap, const char *bp, uint4(&a)[NP][MT], uint4(&b)[NP][NT]) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[pl][nt] = *reinterpret_cast<const uint4 *>(bp + nt * 2 * PG + pl * 1024);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[pl][mt] = *reinterpret_cast<const uint4 *>(ap + mt * 2 * PG + pl * 1024);
        }
    };
    // six products, smallest first: (a0,b2) (a2,b0) (a1,b1) (a0,b1) (a1,b0) (a0,b0); `between(term)` runs after each group
    auto multiply = [&](const uint4(&a)[NP][MT], const uint4(&b)[NP][NT], auto between) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int term = 0; term < TT::N; ++term) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16<NP>(a[TT::PA[term]][mt], b[TT::PB[term]][nt], acc[mt][nt]);
            between(term);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto nothing = [](int) {};
    int cur = 0, nxt = NSTAGE - 1;  // stage being computed / stage the next issue goes to
    for (int s = 0; s < total; ++s) {
        // slab s has landed once at most (slabs issued after it) * G of this wave's DMAs are still outstanding
        const int later = issued - s - 1;
        if (NSTAGE >= 4 && later >= 2) wait_vmcnt<(NSTAGE >= 4 ? 2 * G : 0)>();
        else if (NSTAGE >= 3 && later >= 1) wait_vmcnt<(NSTAGE >= 3 ? G : 0)>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of slab s-1 have returned (WAR on its stage)
        __builtin_amdgcn_s_barrier();  // everyone's slab-s pieces landed; everyone is done reading slab s-1
        const char *st = smem + cur * STAGE;
        load_frags(st + a_lane0, st + b_lane0, a0, b0);
        const bool more = issued < total;  // block-uniform
        if (s > 0) {
            multiply(a1, b1, [&](int term) {  // slab s-1, k-step 1
                if (more) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < RG; ++j)
                        if (j % TT::N == term) issue_rowgroup(j, nxt);  // into the stage slab s-1 occupied
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        } else if (more) {
#pragma unroll
            for (int j = 0; j < RG; ++j) issue_rowgroup(j, nxt);
        }
        if (more && ++issued < total) advance();
        load_frags(st + a_lane1, st + b_lane1, a1, b1);
        multiply(a0, b0, nothing);
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
        nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }
    if (total > 0) multiply(a1, b1, nothing);
    } else {
    // ---- M16 main loop.  A slab is multiplied in four quadrant steps over (row half, column half) of the wave's blocks; the
    // order alternates with the slab's parity -- even: (lo,lo) (lo,hi) (hi,hi) (hi,lo), odd: (lo,hi) (lo,lo) (hi,lo) (hi,hi)
    // -- so that between ANY two consecutive steps, slab boundaries included, the operand halves that change are dead
    // registers: four fragment sets (a_lo, a_hi, b_lo, b_hi; 96 VGPRs for a 64 x 64 wave tile) are enough to have every
    // ds_read issued one full step (24 MFMAs) ahead of its use.  Steps 3 and 4 of slab s-1 run after the barrier of slab s
    // (the skew of the 32x32 loop), with the DMA issue of the slab NSTAGE-1 ahead between their MFMA groups.
    uint4 a_lo[NP][MH], a_hi[NP][MH], b_lo[NP][NH], b_hi[NP][NH];
    // the loader's per-rowgroup state as plain values (the optimizer left rb_off[] / vmask[] in scratch memory when they were
    // reached through the nested closures of this loop: a scratch load in front of every DMA issue, with a vmcnt(0) behind it)
    static_assert(RG <= 4, "rowgroups per wave");
    const unsigned rbv0 = rb_off[0], rbv1 = rb_off[RG > 1 ? 1 : 0], rbv2 = rb_off[RG > 2 ? 2 : 0], rbv3 = rb_off[RG > 3 ? 3 : 0];
    const unsigned long long vmv0 = vmask[0], vmv1 = vmask[RG > 1 ? 1 : 0], vmv2 = vmask[RG > 2 ? 2 : 0], vmv3 = vmask[RG > 3 ? 3 : 0];
    auto issue16 = [&, rbv0, rbv1, rbv2, rbv3, vmv0, vmv1, vmv2, vmv3](auto jc, int stage) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const unsigned rbo = j == 0 ? rbv0 : j == 1 ? rbv1 : j == 2 ? rbv2 : rbv3;
        const unsigned long long vm = j == 0 ? vmv0 : j == 1 ? vmv1 : j == 2 ? vmv2 : vmv3;
        const unsigned st = smem_base + (unsigned)(stage * STAGE);
        const int q = wave + j * NW;
        if (q < AG) {
            if (XE_NO_DMAA) return;
            const bool valid = (vm >> nx_t) & 1ull;
            const unsigned voff = valid ? rbo + nx_a_uni : p.zero_off + chunk_off;
            const unsigned dst = __builtin_amdgcn_readfirstlane(st + q * PG);
            dma16_group<NP>(rsa, dst, voff);
        } else if (!XE_NO_DMAB) {
            const unsigned voff = rbo + nx_b_uni;
            const unsigned dst = __builtin_amdgcn_readfirstlane(st + A_BYTES + (q - AG) * PG);
            dma16_group<NP>(rsb, dst, voff);
        }
    };
    bool exp_rd = true;
    auto rd_a = [&](const char *st, uint4(&a)[NP][MH], int half) __attribute__((always_inline)) {
        if (XE_NO_RDA && !exp_rd) return;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i2 = 0; i2 < MH; ++i2) a[pl][i2] = *reinterpret_cast<const uint4 *>(st + a16 + (half * MH + i2) * PG + pl * 1024);
    };
    auto rd_b = [&](const char *st, uint4(&b)[NP][NH], int half) __attribute__((always_inline)) {
        if (XE_NO_RDB && !exp_rd) return;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j2 = 0; j2 < NH; ++j2) b[pl][j2] = *reinterpret_cast<const uint4 *>(st + b16 + (half * NH + j2) * PG + pl * 1024);
    };
    // one quadrant: blocks (I0 + i, J0 + j); six products, smallest first; `between(term)` runs after each term's MFMAs
    // (the term index travels as a TYPE: an index the optimizer only sees as a loop variable made it keep rb_off[] / vmask[]
    // in scratch memory, and scratch traffic shares the vmcnt counter the DMA waits are counted on)
    auto quad = [&](const uint4(&a)[NP][MH], const uint4(&b)[NP][NH], auto I0, auto J0, auto between) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
        auto one = [&](auto T) __attribute__((always_inline)) {
            constexpr int term = decltype(T)::value;
#pragma unroll
            for (int i2 = 0; i2 < MH; ++i2)
#pragma unroll
                for (int j2 = 0; j2 < NH; ++j2)
                    if (XE != 7) acc16[decltype(I0)::value + i2][decltype(J0)::value + j2] =
                        mfma32<NP>(a[TT::PA[term]][i2], b[TT::PB[term]][j2], acc16[decltype(I0)::value + i2][decltype(J0)::value + j2]);
            between(T);
        };
        one(std::integral_constant<int, 0>{});
        if constexpr (NP >= 2) {
            one(std::integral_constant<int, 1>{});
            one(std::integral_constant<int, 2>{});
        }
        if constexpr (NP == 3) {
            one(std::integral_constant<int, 3>{});
            one(std::integral_constant<int, 4>{});
            one(std::integral_constant<int, 5>{});
        }
        __builtin_amdgcn_s_setprio(0);
    };
    using C0 = std::integral_constant<int, 0>;
    using CMH = std::integral_constant<int, MH>;
    using CNH = std::integral_constant<int, NH>;
    auto nothing = [](auto) {};
    int cur = 0, nxt = NSTAGE - 1;
    // slab s with FIRST = the column half its steps 1 and 4 use (bf), SECOND = the other (bs); the previous slab had them swapped
    auto slab = [&](int s, uint4(&bf)[NP][NH], uint4(&bs)[NP][NH], auto JF, auto JS, int hf, int hs) __attribute__((always_inline)) {
        const int later = issued - s - 1;
        if (NSTAGE >= 4 && later >= 2) wait_vmcnt<(NSTAGE >= 4 ? 2 * G : 0)>();
        else if (NSTAGE >= 3 && later >= 1) wait_vmcnt<(NSTAGE >= 3 ? G : 0)>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (XE == 12 && s == 0 && round == 0) xt2 = wall_clock64();
        const char *st = smem + cur * STAGE;
        const bool more = issued < total;  // block-uniform
        // (sched_barriers pin the order: hoisting a fragment read above the MFMAs that still use its registers costs the
        // compiler extra registers, and this kernel has none to spare -- a spill inside the loop would also put scratch
        // traffic on the vmcnt counter the DMA waits are counted on)
        rd_a(st, a_lo, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s > 0) {
            quad(a_hi, bf, CMH{}, JF, [&](auto T) __attribute__((always_inline)) {  // step 3 of slab s-1: (hi rows, ITS second half = this slab's first)
                // rowgroup j of the next slab is issued after term j (two-part form: three terms, so rowgroup 3 rides with term 2)
                constexpr int tv = decltype(T)::value;
                if constexpr (tv < RG) {
                    if (more) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue16(T, nxt);
                        if constexpr (tv == TT::N - 1) {  // the last term carries every rowgroup the terms did not cover (one term: all but the first)
                            if constexpr (RG > TT::N) issue16(std::integral_constant<int, (RG > TT::N ? TT::N : 0)>{}, nxt);
                            if constexpr (RG > TT::N + 1) issue16(std::integral_constant<int, (RG > TT::N + 1 ? TT::N + 1 : 0)>{}, nxt);
                            if constexpr (RG > TT::N + 2) issue16(std::integral_constant<int, (RG > TT::N + 2 ? TT::N + 2 : 0)>{}, nxt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            rd_b(st, bf, hf);
            __builtin_amdgcn_sched_barrier(0);
            quad(a_hi, bs, CMH{}, JS, nothing);      // step 4 of slab s-1
        } else {
            if (more) {
                static_assert(RG <= 4, "rowgroups per wave");
                if constexpr (RG > 0) issue16(std::integral_constant<int, 0>{}, nxt);
                if constexpr (RG > 1) issue16(std::integral_constant<int, 1>{}, nxt);
                if constexpr (RG > 2) issue16(std::integral_constant<int, 2>{}, nxt);
                if constexpr (RG > 3) issue16(std::integral_constant<int, 3>{}, nxt);
            }
            rd_b(st, bf, hf);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more && ++issued < total) advance();
        rd_b(st, bs, hs);
        rd_a(st, a_hi, 1);
        __builtin_amdgcn_sched_barrier(0);
        quad(a_lo, bf, C0{}, JF, nothing);           // step 1
        __builtin_amdgcn_sched_barrier(0);
        quad(a_lo, bs, C0{}, JS, nothing);           // step 2
        __builtin_amdgcn_sched_barrier(0);
        exp_rd = false;
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
        nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    };
    int s = 0;
    for (; s + 1 < total; s += 2) {
        slab(s, b_lo, b_hi, C0{}, CNH{}, 0, 1);
        slab(s + 1, b_hi, b_lo, CNH{}, C0{}, 1, 0);
    }
    if (s < total) {
        slab(s, b_lo, b_hi, C0{}, CNH{}, 0, 1);
        quad(a_hi, b_hi, CMH{}, CNH{}, nothing);     // steps 3, 4 of the last (even) slab
        quad(a_hi, b_lo, CMH{}, C0{}, nothing);
    } else if (total > 0) {
        quad(a_hi, b_lo, CMH{}, C0{}, nothing);      // steps 3, 4 of the last (odd) slab: (hi, lo) then (hi, hi)
        quad(a_hi, b_hi, CMH{}, CNH{}, nothing);
    }
    }
    wait_vmcnt<0>();
    if (XE == 12 && round == 0) xt3 = wall_clock64();
    if constexpr (NP == 2) {
        // two-part operands carry per-tensor power-of-two scales: sums of (x sa)(w sb) -> multiply by 1 / (sa sb), exact.
        // Done on the raw accumulators, so partial stream-K slabs, BN statistics and the epilogue all see true values.
        const float inv = *reinterpret_cast<const float *>(p.x3 + p.x3_tr) * *reinterpret_cast<const float *>(p.w3 + p.w3_tr);
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < MT16; ++i)
#pragma unroll
                for (int j = 0; j < NT16; ++j) acc16[i][j] *= inv;
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] *= inv;
        }
    }
    __syncthreads();  // all waves out of the main loop: the ring is free for the epilogue
    if (XE == 9) continue;

    if ((p.stat_partial || p.stat_sums) && complete) {
        // BatchNorm batch statistics of the RAW output, one partial row per M-tile (rows >= M are zero: their taps all
        // read the zero row)
        float *red = reinterpret_cast<float *>(smem);  // [WARPS_M][2][BN]
        if constexpr (M16) {
#pragma unroll
            for (int j2 = 0; j2 < NT16; ++j2) {  // a lane holds column lane & 15 of its four rows of every block
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i2 = 0; i2 < MT16; ++i2)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float v = acc16[i2][j2][reg];
                        s1 += v;
                        s2 += v * v;
                    }
                s1 += __shfl_xor(s1, 16, 64);
                s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 16) {
                    red[(wm * 2 + 0) * BN + wn * TNW + j2 * 16 + lane] = s1;
                    red[(wm * 2 + 1) * BN + wn * TNW + j2 * 16 + lane] = s2;
                }
            }
        } else {
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
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += 64 * NW) {
            const int which = i / BN, col = i - which * BN;
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < WARPS_M; ++q) a += red[(q * 2 + which) * BN + col];
            if (n0 + col < p.K) {
                if (p.stat_sums) unsafeAtomicAdd(p.stat_sums + (long)which * p.K + n0 + col, (double)a);
                else p.stat_partial[((long)mt_i * 2 + which) * p.K + n0 + col] = a;
            }
        }
        __syncthreads();
    }

    // ---- epilogue: every wave bounces its 32 x 32 accumulator blocks through a private LDS patch and stores 16 B per lane
    float *y = reinterpret_cast<float *>(p.y);
    const float *res = reinterpret_cast<const float *>(p.res);
    constexpr int PBLK = (NT >= 2) ? 2 : 1;
    constexpr int PW = PBLK * 32, PITCH = PW + 4, C4 = PW / 4;
    static_assert(NW * 32 * PITCH * 4 <= SMEM, "epilogue patches must fit the ring");
    static_assert(NT % PBLK == 0, "column blocks per pass");
    float *patch = reinterpret_cast<float *>(smem) + wave * 32 * PITCH;
    const bool vec_ok = (!y || (p.ldy & 3) == 0) && ((p.K & 3) == 0) && (!res || (p.ldr & 3) == 0);
    const long res_shift = p.res_groups > 0 ? (long)((grp % p.res_groups) - grp) * p.group_rows : 0;  // (rows: the residual of image g % res_groups)
    float *slab = SIMPLE || complete ? nullptr : p.ws + ((long)wgid * 2 + (sk_seg > 0 ? 1 : 0)) * (BM * BN);
    bool bstat = false;
    if constexpr (SIMPLE) bstat = p.bs_sums != nullptr;  // (host: only with f32 rows out, 16-B aligned rows, no scale / shift / act)
    const bool plain = !bstat && y && vec_ok && !p.scale && !p.shift && !res && p.act == DASS_ACT_NONE && !p.y3 && !p.y_amax && !phase && p.ldy < (1l << 22);
    float *y_tile = y + (long)m0 * p.ldy + n0;
    const int ldy32 = (int)p.ldy;
    f32x4 bs0[NT / PBLK], bs1[NT / PBLK], bsm[NT / PBLK];
#pragma unroll
    for (int i = 0; i < NT / PBLK; ++i) bs0[i] = bs1[i] = bsm[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float y3_scale = 1.f, vmax = 0.f;
    if (p.y3 && p.y3_parts == 2)  // the output's scale: from the bound dass_x3_prepare_out left in y3's trailer (same in every workgroup)
        y3_scale = x3_scale_of(*reinterpret_cast<const float *>(p.y3 + ((long)p.M + 1) * p.cc_out * 128 + 4));
    if (p.y3 && tile == 0 && s_lo == 0) {  // the zero row consumers point padded taps at (+ the trailer of a three-part y3)
        const int zb = p.y3_parts * 64, n16 = p.cc_out * p.y3_parts * 4;
        for (int i = tid; i < n16 + (p.y3_parts != 2 ? 1 : 0); i += 64 * NW)
            *reinterpret_cast<uint4 *>(p.y3 + (long)p.M * p.cc_out * zb + i * 16) = i < n16 ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(0x3f800000u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int ntp = 0; ntp < NT; ntp += PBLK) {
            // bstat: the loads of this pass (the linked layer's conv output, the forked gradient, the gate bits, the channel
            // vectors) are issued BEFORE the accumulators bounce through LDS, so that their latency runs under the bounce
            constexpr int ITS = (32 * C4) / 64;
            f32x4 pf_y[SIMPLE ? ITS : 1], pf_r[SIMPLE ? ITS : 1];
            unsigned pf_g[SIMPLE ? ITS : 1];
            f32x4 ch_mu = {0.f, 0.f, 0.f, 0.f}, ch_is = ch_mu, ch_sc = ch_mu, ch_sh = ch_mu;
            bool ep_pre = false;
            if constexpr (SIMPLE) {
                if (!bstat && !plain && vec_ok) {  // fused epilogue (inference: scale / shift / residual / act / split rows out)
                    ep_pre = true;
                    const int kc = n0 + wn * TNW + ntp * 32 + (lane & (C4 - 1)) * 4;
                    ch_sc = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (kc < p.K) {
                        if (p.scale) ch_sc = *reinterpret_cast<const f32x4 *>(p.scale + kc);
                        if (p.shift) ch_sh = *reinterpret_cast<const f32x4 *>(p.shift + kc);
                    }
#pragma unroll
                    for (int it = 0; it < ITS; ++it) {
                        const int mm = m0 + wm * TMW + mt * 32 + (it * 64 + lane) / C4;
                        pf_r[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (res && mm < m_end && kc < p.K) pf_r[it] = *reinterpret_cast<const f32x4 *>(res + ((long)mm + res_shift) * p.ldr + kc);
                    }
                }
                if (bstat) {
                    const int kc = n0 + wn * TNW + ntp * 32 + (lane & (C4 - 1)) * 4;
                    if (kc < p.K) {
                        ch_mu = *reinterpret_cast<const f32x4 *>(p.bs_mean + kc);
                        ch_is = *reinterpret_cast<const f32x4 *>(p.bs_invstd + kc);
                        if (!p.bs_gates && p.bs_act != DASS_ACT_NONE) {
                            ch_sc = *reinterpret_cast<const f32x4 *>(p.bs_gsc + kc);
                            ch_sh = *reinterpret_cast<const f32x4 *>(p.bs_gsh + kc);
                        }
                    }
#pragma unroll
                    for (int it = 0; it < ITS; ++it) {
                        const int idx = it * 64 + lane;
                        const int row = idx / C4;
                        const int mm = m0 + wm * TMW + mt * 32 + row;
                        pf_y[it] = pf_r[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                        pf_g[it] = 0xfu;
                        if (mm < m_end && kc < p.K) {
                            pf_y[it] = *reinterpret_cast<const f32x4 *>(p.bs_y + (long)mm * p.K + kc);
                            if (res) pf_r[it] = *reinterpret_cast<const f32x4 *>(res + ((long)mm + res_shift) * p.ldr + kc);
                            if (p.bs_gates) pf_g[it] = p.bs_gates[(long)mm * (p.K >> 2) + (kc >> 2)];
                        }
                    }
                }
            }
            if constexpr (M16) {  // C/D layout of the 16x16 blocks: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int jb = 0; jb < 2 * PBLK; ++jb)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            patch[(ib * 16 + (lane >> 4) * 4 + reg) * PITCH + jb * 16 + (lane & 15)] = acc16[mt * 2 + ib][ntp * 2 + jb][reg];
            } else {
#pragma unroll
            for (int q = 0; q < PBLK; ++q)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    patch[row * PITCH + q * 32 + (lane & 31)] = acc[mt][ntp + q][reg];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < (32 * C4) / 64; ++it) {
                const int idx = it * 64 + lane;
                const int row = idx / C4, c4 = idx - row * C4;
                const int lr = wm * TMW + mt * 32 + row, lc = wn * TNW + ntp * 32 + c4 * 4;  // tile-local
                const f32x4 v = *reinterpret_cast<const f32x4 *>(patch + row * PITCH + c4 * 4);
                if (slab) {  // partial segment: raw sums to the workspace (the fix-up pass owns the epilogue of this tile)
                    *reinterpret_cast<f32x4 *>(slab + lr * BN + lc) = v;
                    continue;
                }
                const int m = m0 + lr, k = n0 + lc;
                if (m >= m_end || k >= p.K) continue;
                if (plain) {  // raw f32 rows and nothing else (every train-mode conv and most input gradients): one 16-B store, 32-bit offsets
                    *reinterpret_cast<f32x4 *>(y_tile + lr * ldy32 + lc) = v;
                    continue;
                }
                if constexpr (SIMPLE) {
                    if (bstat) {  // input gradient + the BN-backward sums of the layer that produced this conv's input
                        f32x4 g = v + pf_r[it];
                        *reinterpret_cast<f32x4 *>(y_tile + lr * ldy32 + lc) = g;
                        const f32x4 yl = pf_y[it];
                        if (p.bs_gates) {
                            const unsigned gb = pf_g[it];
#pragma unroll
                            for (int e = 0; e < 4; ++e) g[e] = ((gb >> e) & 1u) ? g[e] : 0.f;
                        } else if (p.bs_act != DASS_ACT_NONE) {
                            const f32x4 o = bn_affine(yl, ch_sc, ch_sh);
#pragma unroll
                            for (int e = 0; e < 4; ++e) g[e] *= act_grad_from_out(o[e], p.bs_act);
                        }
                        const f32x4 xh = (yl - ch_mu) * ch_is;
                        bs0[ntp / PBLK] += g;
                        bs1[ntp / PBLK] += g * xh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) bsm[ntp / PBLK][e] = fmaxf(bsm[ntp / PBLK][e], fabsf(g[e]));
                        continue;
                    }
                }
                if constexpr (SIMPLE) x3_store_out(p, v, m, k, ohw, vec_ok, y3_scale, vmax, false, ep_pre, ch_sc, ch_sh, pf_r[it], res_shift);
                else x3_store_out(p, v, m, k, ohw, vec_ok, y3_scale, vmax, phase, false, f32x4{1.f, 1.f, 1.f, 1.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, res_shift);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    x3_amax_commit(p.y_amax, vmax);
    if constexpr (SIMPLE) {
        if (bstat) {
            // lanes with the same 4-channel column group (lane % C4) hold partial sums over different rows: fold them, then the
            // WARPS_M waves of a column through LDS, then one f64 atomic per channel and workgroup (as the forward statistics)
            __syncthreads();  // every wave is done with its epilogue patch
            float *red = reinterpret_cast<float *>(smem);  // [WARPS_M][3][BN]
            static_assert(WARPS_M * 3 * BN * 4 <= SMEM, "bn-backward partials must fit the ring");
#pragma unroll
            for (int i = 0; i < NT / PBLK; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a0 = bs0[i][e], a1 = bs1[i][e], a2 = bsm[i][e];
#pragma unroll
                    for (int o = C4; o < 64; o <<= 1) {
                        a0 += __shfl_xor(a0, o, 64);
                        a1 += __shfl_xor(a1, o, 64);
                        a2 = fmaxf(a2, __shfl_xor(a2, o, 64));
                    }
                    if (lane < C4) {
                        const int col = wn * TNW + i * PW + lane * 4 + e;
                        red[(wm * 3 + 0) * BN + col] = a0;
                        red[(wm * 3 + 1) * BN + col] = a1;
                        red[(wm * 3 + 2) * BN + col] = a2;
                    }
                }
            }
            __syncthreads();
            for (int i = tid; i < 3 * BN; i += 64 * NW) {
                const int which = i / BN, col = i - which * BN;
                if (n0 + col >= p.K) continue;
                if (which < 2) {
                    float a = 0.f;
#pragma unroll
                    for (int q2 = 0; q2 < WARPS_M; ++q2) a += red[(q2 * 3 + which) * BN + col];
                    unsafeAtomicAdd(p.bs_sums + (long)which * p.K + n0 + col, (double)a);
                } else {
                    float a = 0.f;
#pragma unroll
                    for (int q2 = 0; q2 < WARPS_M; ++q2) a = fmaxf(a, red[(q2 * 3 + 2) * BN + col]);
                    if (!(a >= 0.f)) a = __uint_as_float(0x7f800000u);  // NaN: an infinite bound
                    unsigned *slot = reinterpret_cast<unsigned *>(p.bs_sums + 2 * (long)p.K) + n0 + col;
                    if (__float_as_uint(a) > *reinterpret_cast<volatile unsigned *>(slot)) atomicMax(slot, __float_as_uint(a));
                }
            }
        }
    }
    if (XE == 12 && round == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xt4 = wall_clock64();
        if (tid == 0) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(p.ws) + (long)wgid * 16;
            o[0] = xt0; o[1] = xt1; o[2] = xt2; o[3] = xt3; o[4] = xt4;
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            o[5] = xa; o[6] = xb; o[7] = xc;
        }
    }
    }  // segments
}

// Fix-up pass of the stream-K decomposition: block i looks at the boundary between workgroups i and i + 1 of the main
// launch.  If it falls inside a tile and is the FIRST boundary inside that tile, the block adds that tile's workspace
// slabs in workgroup order (a fixed order: results do not depend on timing), runs the epilogue and, for train-mode BN,
// the tile's partial statistics.
template <int BM, int BN> __global__ __launch_bounds__(256) void conv_x3_fixup_kernel(const X3P p) {
    const int ntaps = p.R * p.S;
    const long S_tile = (long)p.CC * ntaps;
    const long U = ((long)p.mtiles * p.ntiles - p.dp_tiles) * S_tile;
    const int i = blockIdx.x;
    auto bound = [&](int w) -> long { return x3_sk_bound(p, w); };  // the main kernel's cut: near-equal ranges, the first sk_r one unit longer
    const long b1 = bound(i + 1);
    const int rtile = (int)(b1 / S_tile);  // tile index inside the stream-K region
    const long t_lo = (long)rtile * S_tile, t_hi = t_lo + S_tile;
    if (b1 == t_lo || b1 >= U) return;   // boundary on a tile edge: nothing is split here
    if (bound(i) > t_lo) return;         // an earlier boundary already lies inside this tile: its block does the work
    const int tile = p.dp_tiles + rtile;
    const int mt_i = tile / p.ntiles, nt_i = tile - mt_i * p.ntiles;
    const int grp = mt_i / p.mt_per_group;
    const int m0 = grp * p.group_rows + (mt_i - grp * p.mt_per_group) * BM, n0 = nt_i * BN;
    const int m_end = (grp + 1) * p.group_rows < p.M ? (grp + 1) * p.group_rows : p.M;
    const int ohw = p.OHs * p.OWs;
    const float *res = reinterpret_cast<const float *>(p.res);
    const float *y = reinterpret_cast<const float *>(p.y);
    const bool vec_ok = (!y || (p.ldy & 3) == 0) && ((p.K & 3) == 0) && (!res || (p.ldr & 3) == 0);
    constexpr int C4 = BN / 4, RPP = 256 / C4;  // 4-column groups per row, rows per pass
    __shared__ float red[2][RPP][BN];
    __shared__ const float *contrib[256];  // the tile's slabs, in workgroup order (64-bit divisions done once, not per row)
    __shared__ int ncontrib;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int nc = 0;
        for (int w = i; w < p.sk_part && nc < 256; ++w) {
            const long bw = bound(w);
            if (bw >= t_hi) break;
            if (bound(w + 1) == bw) continue;  // a workgroup with an empty range wrote nothing
            contrib[nc++] = p.ws + ((long)w * 2 + (bw >= t_lo ? 0 : 1)) * (BM * BN);
        }
        ncontrib = nc;
    }
    __syncthreads();
    const int nc = ncontrib;
    const int c4 = tid % C4, r0 = tid / C4;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    float y3_scale = 1.f, vmax = 0.f;
    if (p.y3 && p.y3_parts == 2) y3_scale = x3_scale_of(*reinterpret_cast<const float *>(p.y3 + ((long)p.M + 1) * p.cc_out * 128 + 4));
    // blockIdx.y cuts the tile's rows (more blocks in flight: the pass is latency-bound); BN statistics need the whole tile
    const int rows_per_block = BM / gridDim.y, row_lo = blockIdx.y * rows_per_block;
    for (int lr = row_lo + r0; lr < row_lo + rows_per_block; lr += RPP) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nc; ++c) v += *reinterpret_cast<const f32x4 *>(contrib[c] + lr * BN + c4 * 4);
        s1 += v;
        s2 += v * v;
        const int m = m0 + lr, k = n0 + c4 * 4;
        if (m < m_end && k < p.K)
            x3_store_out(p, v, m, k, ohw, vec_ok, y3_scale, vmax, p.o_mul != 1, false, f32x4{1.f, 1.f, 1.f, 1.f}, f32x4{0.f, 0.f, 0.f, 0.f},
                         f32x4{0.f, 0.f, 0.f, 0.f}, p.res_groups > 0 ? (long)((grp % p.res_groups) - grp) * p.group_rows : 0);
    }
    x3_amax_commit(p.y_amax, vmax);
    if (p.stat_partial || p.stat_sums) {  // rows >= M of the slabs are exact zeros
        constexpr int NR = RPP;
        for (int e = 0; e < 4; ++e) {
            red[0][r0][c4 * 4 + e] = s1[e];
            red[1][r0][c4 * 4 + e] = s2[e];
        }
        __syncthreads();
        for (int j = tid; j < 2 * BN; j += 256) {
            const int which = j / BN, col = j - which * BN;
            float a = 0.f;
            for (int q = 0; q < NR; ++q) a += red[which][q][col];
            if (n0 + col < p.K) {
                if (p.stat_sums) unsafeAtomicAdd(p.stat_sums + (long)which * p.K + n0 + col, (double)a);  // (this block's rows)
                else p.stat_partial[((long)mt_i * 2 + which) * p.K + n0 + col] = a;
            }
        }
    }
}

// f32 rows [M][ld] (C real channels) -> x3 rows [M + 1][CC][192]; channels >= C and row M are zero.
// One thread per 8 channels: two 16-B loads, three 16-B stores.  HBM-bound: 4 B read + 6 B written per element.
// NP = 2: the tensor's bound sits in the trailer (dass_absmax_rows_kernel ran before); block 0 completes the trailer.
template <int NP>
__global__ __launch_bounds__(256) void split3_rows_kernel(const float *__restrict__ x, long ld, char *__restrict__ out, long M, int C, int CC,
                                                          const float *__restrict__ nc_scale, long rows_per_image) {
    constexpr int SB = NP * 64;
    float scale = 1.f;
    if constexpr (NP == 2) {
        unsigned *tr = reinterpret_cast<unsigned *>(out + (M + 1) * CC * SB);
        scale = x3_scale_of(__uint_as_float(tr[1]));
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            tr[0] = __float_as_uint(x3_inv_of(scale));
            tr[2] = tr[1];  // (here the bound IS max |x|)
        }
    }
    const long units = (M + 1) * CC * 4;  // 8-channel units
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (long)gridDim.x * blockDim.x) {
        const int oct = (int)(u & 3);
        const long rc = u >> 2;
        const int cc = (int)(rc % CC);
        const long m = rc / CC;
        const int c = cc * 32 + oct * 8;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (m < M) {
            const float *src = x + m * ld + c;
            if (c + 4 <= C) v0 = *reinterpret_cast<const f32x4 *>(src);
            else
                for (int e = 0; e < 4; ++e)
                    if (c + e < C) v0[e] = src[e];
            if (c + 8 <= C) v1 = *reinterpret_cast<const f32x4 *>(src + 4);
            else
                for (int e = 0; e < 4; ++e)
                    if (c + 4 + e < C) v1[e] = src[4 + e];
            if (nc_scale) {  // Dropout2d mask of the producer: per (image, channel) multipliers {0, 1/(1-p)}
                const float *sc = nc_scale + (m / rows_per_image) * C + c;
                for (int e = 0; e < 4; ++e) {
                    if (c + e < C) v0[e] *= sc[e];
                    if (c + 4 + e < C) v1[e] *= sc[4 + e];
                }
            }
        }
        char *d = out + (m * CC + cc) * SB + oct * 16;
        if constexpr (NP == 1) {
            *reinterpret_cast<uint4 *>(d) = make_uint4(pk_bf16(dass_f32x2{v0[0], v0[1]}), pk_bf16(dass_f32x2{v0[2], v0[3]}),
                                                       pk_bf16(dass_f32x2{v1[0], v1[1]}), pk_bf16(dass_f32x2{v1[2], v1[3]}));
        } else if constexpr (NP == 3) {
            uint2 a0, a1, a2, b0, b1, b2;
            split3_4(v0, a0, a1, a2);
            split3_4(v1, b0, b1, b2);
            *reinterpret_cast<uint4 *>(d) = make_uint4(a0.x, a0.y, b0.x, b0.y);
            *reinterpret_cast<uint4 *>(d + 64) = make_uint4(a1.x, a1.y, b1.x, b1.y);
            *reinterpret_cast<uint4 *>(d + 128) = make_uint4(a2.x, a2.y, b2.x, b2.y);
        } else {
            uint2 a0, a1, b0, b1;
            split2_4(v0 * scale, a0, a1);
            split2_4(v1 * scale, b0, b1);
            *reinterpret_cast<uint4 *>(d) = make_uint4(a0.x, a0.y, b0.x, b0.y);
            *reinterpret_cast<uint4 *>(d + 64) = make_uint4(a1.x, a1.y, b1.x, b1.y);
        }
    }
}

// max |x * nc_scale| over f32 rows [M][ld] (C channels) -> atomic max into *bound_bits (zeroed by the caller; non-negative floats
// order like their bit patterns).  The bound of the two-part format's per-tensor scale when no producer supplied one.
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float *__restrict__ x, long ld, long M, int C, const float *__restrict__ nc_scale,
                                                          long rows_per_image, unsigned *__restrict__ bound_bits) {
    const int c4 = (C + 3) >> 2;
    const long units = M * c4;
    float mx = 0.f;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (long)gridDim.x * blockDim.x) {
        const long m = u / c4;
        const int c = (int)(u - m * c4) * 4;
        const float *src = x + m * ld + c;
        const float *sc = nc_scale ? nc_scale + (m / rows_per_image) * C + c : nullptr;
        if (c + 4 <= C) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(src);
            if (sc) v *= *reinterpret_cast<const f32x4 *>(sc);
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[e]));
        } else {
            for (int e = 0; c + e < C; ++e) mx = fmaxf(mx, fabsf(src[e] * (sc ? sc[e] : 1.f)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (!(mx >= 0.f)) mx = __uint_as_float(0x7f800000u);  // a NaN in the tensor: propagate as an infinite bound
        atomicMax(bound_bits, __float_as_uint(mx));
    }
}

static int g_cus = 0;
static int g_bn_fused = 0;  // did the last launch_x3 fuse the BN-backward sums it was asked for? (read back by the entry point)
static int g_last_pick = 0;  // (BM << 16) | (BN << 4) | 2 * whole-tile kernel | stream-K: the schedule of the last launch_x3 (dass_x3_last_pick)
static int g_x3_parts = 3;  // operand format of the pre-split kernels: 3 = bf16 triple (bf16x6 engine), 2 = scaled f16 pair (f16x3 engine)  // compute units of the current device (stream-K launches one workgroup per resident slot)

template <int BM, int BN, int WARPS_M, int WARPS_N, int NSTAGE, int NP = 3> int launch_x3(X3P &p, hipStream_t st, int mode, long ws_bytes, bool m16) {
    constexpr int LDS = NSTAGE * (BM + BN) * (NP * 64) + 256;
    constexpr int RES = (160 * 1024) / LDS >= 4 ? 4 : (160 * 1024) / LDS;  // resident workgroups per CU (LDS-limited)
    p.mt_per_group = (p.group_rows + BM - 1) / BM;
    p.mtiles = p.mt_per_group * ((p.M + p.group_rows - 1) / p.group_rows);
    p.ntiles = (p.K + BN - 1) / BN;
    const long tiles = (long)p.mtiles * p.ntiles, slots = (long)g_cus * RES;
    const long units = tiles * p.CC * p.R * p.S;
    // one tile per workgroup when that quantises well (or the problem is tiny); equal slab ranges per resident slot otherwise
    bool stream = mode >= 2;  // 3 = stream-K over ALL tiles (no whole-round part): segment ends, hence epilogues, are staggered
    if (mode == 0) {
        const long rounds = (tiles + slots - 1) / slots;
        const double eff = (double)tiles / (double)(rounds * slots);  // busy share of the last-round-limited schedule
        stream = eff < 0.92 && units >= 6 * slots && (long)p.CC * p.R * p.S >= 4;
    }
    if (stream && (!p.ws || ws_bytes < 2 * slots * (long)BM * BN * 4)) stream = false;
    // stream-K: whole rounds of tiles go one per workgroup (fused epilogue, no workspace), only the remainder -- less than
    // one round -- is cut into equal slab ranges, each at least 4 slabs long (bounds the fix-up's contributor table)
    p.sk_wgs = stream ? (int)slots : (int)tiles;
    p.dp_tiles = stream && mode != 3 && !p.stat_partial ? (int)(tiles / slots * slots) : 0;  // (the statistics fix-up is one block per tile)
    p.sk_part = p.sk_wgs;
    if (stream) {
        const long s_tile = (long)p.CC * p.R * p.S, ur = (tiles - p.dp_tiles) * s_tile;
        if (ur / 4 < p.sk_part) p.sk_part = ur / 4 > 0 ? (int)(ur / 4) : 1;
        if (p.dp_tiles == tiles) stream = false;  // the tile count is a whole number of rounds
        else if (s_tile / (ur / p.sk_part > 0 ? ur / p.sk_part : 1) + 2 > 256) return DASS_ERR_UNSUPPORTED;  // cannot happen for S_tile <= 1016
    }
    {   // what the kernel's per-workgroup setup needs divided (x3_fastdiv, x3_sk_bound)
        const long ur = (tiles - p.dp_tiles) * (long)p.CC * p.R * p.S;
        if (units >= (1l << 31)) return DASS_ERR_UNSUPPORTED;
        p.sk_q = (int)(ur / p.sk_part);
        p.sk_r = (int)(ur % p.sk_part);
        p.dp_rounds = p.dp_tiles / p.sk_wgs;
        p.whole = (p.dp_tiles == 0 && p.sk_wgs == tiles && p.sk_part == p.sk_wgs) ? 1 : 0;
        x3_set_magic(p.OHs * p.OWs, p.mg_ohw, p.sh_ohw);
        x3_set_magic(p.OWs, p.mg_ows, p.sh_ows);
    }
    const bool simple = p.whole && p.o_mul == 1;  // (per-image groups included: their tile -> rows map is two more multiplies)
    // whole-tile forms of the two-part kernel: the production picks + the 128 x 64 tile (wins layer-1 / 2 / 4 shapes, r04 sweep)
    constexpr bool has_simple = NP <= 2 && ((BM == 64 && BN == 64) || (BM == 256 && BN == 128) || (BM == 128 && BN == 64));
    if (p.bs_sums && !(has_simple && simple)) p.bs_sums = nullptr;  // not fused: the caller runs dass_bn_bwd_reduce_sums itself
    g_bn_fused = p.bs_sums ? 1 : 0;
    g_last_pick = (BM << 16) | (BN << 4) | ((has_simple && simple) ? 2 : 0) | (stream ? 1 : 0);
    if constexpr (has_simple) {
        if (simple) {
            // (DASS_X3_LDS_PAD: bytes of dynamic LDS added to a whole-tile launch -- an occupancy experiment: the dispatcher then places fewer
            //  workgroups per CU and has to spread a grid of ~2 tiles per CU more evenly; 0 = off)
            static const int lds_pad = getenv("DASS_X3_LDS_PAD") ? atoi(getenv("DASS_X3_LDS_PAD")) : 0;
            DASS_LAUNCH((conv_x3_kernel<BM, BN, WARPS_M, WARPS_N, NSTAGE, true, NP, true>), dim3(p.sk_wgs), dim3(64 * WARPS_M * WARPS_N), lds_pad, st, p);
            DASS_LAUNCH_CHECK();
            return DASS_OK;
        }
    }
    if (m16 || NP != 3)  // (the two-part and one-part formats are built for the 16x16x32 shape only)
        DASS_LAUNCH((conv_x3_kernel<BM, BN, WARPS_M, WARPS_N, NSTAGE, true, NP>), dim3(p.sk_wgs), dim3(64 * WARPS_M * WARPS_N), 0, st, p);
    else if constexpr (NP == 3)
        DASS_LAUNCH((conv_x3_kernel<BM, BN, WARPS_M, WARPS_N, NSTAGE, false, 3>), dim3(p.sk_wgs), dim3(64 * WARPS_M * WARPS_N), 0, st, p);
    DASS_LAUNCH_CHECK();
    if (stream) {
        constexpr int RPP = 256 / (BN / 4);
        const int row_split = p.stat_partial ? 1 : (BM / RPP >= 8 ? 8 : BM / RPP);
        DASS_LAUNCH((conv_x3_fixup_kernel<BM, BN>), dim3(p.sk_part, row_split), dim3(256), 0, st, p);
        DASS_LAUNCH_CHECK();
    }
    return DASS_OK;
}

static int g_x3_m16 = 1;      // default MFMA shape of the pre-split kernels: 16x16x32 (DASS_X3_MFMA=32 selects 32x32x16)
static int g_x3_force = -1;  // tuning / test knob: dass_x3_force_tile(); -1 = take DASS_X3_TILE from the environment once

// force = tile + 10 * mode: tile 0 = model's choice, 1..7 = variant; mode 0 = auto, 1 = one tile per workgroup, 2 = stream-K
static int dispatch_x3(X3P &p, hipStream_t st, long ws_bytes) {
    if (g_x3_force < 0) {
        g_x3_force = getenv("DASS_X3_TILE") ? atoi(getenv("DASS_X3_TILE")) : 0;
        if (getenv("DASS_X3_MFMA")) g_x3_m16 = atoi(getenv("DASS_X3_MFMA")) != 32;
    }
    if (!g_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return DASS_ERR_LAUNCH;
        g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // force = tile + 10 * mode + 100 * shape: shape 0 = default MFMA shape, 1 = 32x32x16, 2 = 16x16x32
    const int shape = g_x3_force / 100;
    const bool m16 = shape == 2 || (shape == 0 && g_x3_m16);
    int mode = (g_x3_force / 10) % 10;
    int pick = g_x3_force % 10;
    if (!pick) {
        // measured on every DeepLab-R101 shape (tools/x3_time.py).  The chip is power-limited under MFMA load (all 256 CUs
        // busy lower the clock), so what decides is bytes moved per MFMA and a balanced schedule:
        //  * many tiles (133128-row layers): 256 x 128, slab ranges balanced by stream-K when the tile count quantises badly;
        //  * long reductions on few tiles (ASPP / layer-4 dilated 3x3 at 8712 rows): 256 x 128 with stream-K always;
        //  * short reductions: 64 x 64 tiles, three workgroups per CU.
        const long t256 = (long)((p.M + 255) / 256) * ((p.K + 127) / 128);
        const long s_tile = (long)p.CC * p.R * p.S;
        // (re-measured after the per-workgroup setup shrank to ~120 instructions, profiles/r03_x3_conv_sweep_f16.txt: whole 64 x 64
        //  tiles now also win the K <= 64 and the 2-slab layers of layer 1 -- their stream-K form pays the general kernel's setup
        //  and a fix-up pass -- and 256 x 128 whole tiles win where they give one well-filled round of 120..256 tiles)
        // (round 4, profiles/r04_x3_conv_sweep.txt: wherever whole 64 x 64 tiles were the pick, whole 128 x 64 tiles -- 25 % fewer
        //  LDS-DMA pieces per MFMA; a piece costs the issuing wave ~100 cycles of in-order issue -- are 5-20 % faster once there are
        //  ~2 of them per CU; 276 tiles (M = 8712, K = 256) quantise badly on 256 CUs and stay 64 x 64, and so do the output-bound
        //  4-slab layers with >= 512 output channels, whose long tile epilogues overlap worse)
        const long t128 = (long)((p.M + 127) / 128) * ((p.K + 63) / 64);
        const bool wide = g_x3_parts <= 2 && t128 >= 2 * g_cus && !(s_tile <= 4 && p.K >= 512);
        if (p.K <= 64 || (t256 >= 4 * g_cus && s_tile <= 4)) { pick = wide ? 3 : 4; if (!mode) mode = 1; }
        else if (t256 >= 4 * g_cus) pick = 1;
        else if (s_tile >= 100) { pick = 1; if (!mode) mode = 2; }
        else if (g_x3_parts <= 2 && t256 >= 120 && t256 <= g_cus && s_tile >= 16) { pick = 1; if (!mode) mode = 1; }
        else { pick = wide ? 3 : 4; if (!mode) mode = 1; }  // short reductions: segments would be too short to amortise the fix-up
    }
    if (g_x3_parts == 2) {
        switch (pick) {
        case 1: return launch_x3<256, 128, 4, 2, 2, 2>(p, st, mode, ws_bytes, true);
        case 2: return launch_x3<128, 128, 4, 2, 3, 2>(p, st, mode, ws_bytes, true);
        case 3: return launch_x3<128, 64, 4, 1, 2, 2>(p, st, mode, ws_bytes, true);
        case 5: return launch_x3<128, 128, 2, 2, 3, 2>(p, st, mode, ws_bytes, true);
        case 6: return launch_x3<64, 64, 2, 2, 3, 2>(p, st, mode, ws_bytes, true);
        case 7: return launch_x3<128, 128, 4, 2, 2, 2>(p, st, mode, ws_bytes, true);
        case 8: return launch_x3<256, 128, 4, 2, 3, 2>(p, st, mode, ws_bytes, true);  // three stages of 48 KB fit only in the two-part format
        case 9: return launch_x3<64, 64, 2, 2, 4, 2>(p, st, mode, ws_bytes, true);   // four stages of 16 KB: three slabs in flight per workgroup
        default: return launch_x3<64, 64, 2, 2, 2, 2>(p, st, mode, ws_bytes, true);
        }
    }
    if (g_x3_parts == 1) {  // "bf16x1": the production tiles only
        switch (pick) {
        case 1: case 8: return launch_x3<256, 128, 4, 2, 2, 1>(p, st, mode, ws_bytes, true);
        case 2: case 5: case 7: return launch_x3<128, 128, 4, 2, 3, 1>(p, st, mode, ws_bytes, true);
        case 3: return launch_x3<128, 64, 4, 1, 2, 1>(p, st, mode, ws_bytes, true);
        default: return launch_x3<64, 64, 2, 2, 2, 1>(p, st, mode, ws_bytes, true);
        }
    }
    switch (pick) {
    case 1: return launch_x3<256, 128, 4, 2, 2>(p, st, mode, ws_bytes, m16);
    case 2: return launch_x3<128, 128, 4, 2, 3>(p, st, mode, ws_bytes, m16);
    case 3: return launch_x3<128, 64, 4, 1, 2>(p, st, mode, ws_bytes, m16);
    case 5: return launch_x3<128, 128, 2, 2, 3>(p, st, mode, ws_bytes, m16);
    case 6: return launch_x3<64, 64, 2, 2, 3>(p, st, mode, ws_bytes, m16);
    case 7: return launch_x3<128, 128, 4, 2, 2>(p, st, mode, ws_bytes, m16);
    default: return launch_x3<64, 64, 2, 2, 2>(p, st, mode, ws_bytes, m16);
    }
}

}  // namespace

/* the magic pair the pre-split conv kernel divides pixel indices with (x3_set_magic / x3_fastdiv above): exported so that the
 * CPU test suite can check  (n * mul >> 32) >> shift == n / d  over the whole 31-bit range without a GPU */
extern "C" int dass_x3_magic(int d, unsigned *mul, int *shift) {
    if (d < 1 || !mul || !shift) return DASS_ERR_ARG;
    x3_set_magic(d, *mul, *shift);
    return DASS_OK;
}

/* which kernel the LAST pre-split conv launch of this process ran: (BM << 16) | (BN << 4) | 2 (whole-tile specialisation) | 1 (stream-K
 * schedule + fix-up pass).  Diagnostic only (bench.py attributes in-step launch times to tile classes with it); host state, no stream */
extern "C" int dass_x3_last_pick(void) { return g_last_pick; }

/* workgroups of the whole-tile kernels the runtime keeps resident per CU (tools / DESIGN: bytes in flight per CU = this x the ring):
 * which = 0: 64x64, 1: 128x64, 2: 256x128 (two-part engine, M16 MFMAs) */
extern "C" int dass_x3_resident_workgroups(int which) {
    int n = 0;
    hipError_t e;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_x3_kernel<64, 64, 2, 2, 2, true, 2, true>, 256, 0);
    else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_x3_kernel<128, 64, 4, 1, 2, true, 2, true>, 256, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_x3_kernel<256, 128, 4, 2, 2, true, 2, true>, 512, 0);
    return e == hipSuccess ? n : -1;
}

extern "C" int dass_x3_force_tile(int tile) {
    g_x3_force = tile < 0 ? 0 : tile;
    return DASS_OK;
}

// two BM x BN f32 slabs per workgroup of the largest stream-K launch (256 CUs x 256 x 128 tiles; every other variant needs less)
extern "C" int64_t dass_conv2d_x3_workspace_bytes(void) { return (int64_t)2 * 256 * (256 * 128) * 4; }

/* operand format of the pre-split kernels and of every x3 buffer allocated from now on: 3 (default) or 2, see dass_common.h */
extern "C" int dass_set_x3_parts(int parts) {
    if (parts < 1 || parts > 3) return DASS_ERR_ARG;
    g_x3_parts = parts;
    return DASS_OK;
}
extern "C" int dass_get_x3_parts(void) { return g_x3_parts; }

extern "C" int64_t dass_x3_bytes(int64_t rows, int C) { return x3_trailer_off(rows, (C + 31) / 32, g_x3_parts) + 16; }

/* bound[0] = max |x| over f32 rows (atomic max: the caller zeroes it), optionally of x * nc_scale[image][channel] */
extern "C" int dass_absmax_rows(const float *x, int64_t ld, int64_t M, int C, const float *nc_scale, int64_t rows_per_image, float *bound,
                                void *stream) {
    if (!x || !bound || M <= 0 || C <= 0 || ld < C || (ld & 3) || ((uintptr_t)x & 15)) return DASS_ERR_ARG;
    if (nc_scale && (rows_per_image <= 0 || (C & 3))) return DASS_ERR_ARG;
    DASS_LAUNCH(absmax_rows_kernel, dim3(dass_grid_1d(M * ((C + 3) / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ld, (long)M, C,
                       nc_scale, (long)(rows_per_image > 0 ? rows_per_image : 1), (unsigned *)bound);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_split3_rows(const float *x, int64_t ld, void *out, int64_t M, int C, const float *nc_scale, int64_t rows_per_image,
                                void *stream) {
    if (!x || !out || M <= 0 || C <= 0 || ld < C) return DASS_ERR_ARG;
    if ((ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return DASS_ERR_ARG;
    if (nc_scale && (rows_per_image <= 0 || (C & 3))) return DASS_ERR_ARG;
    const int CC = (C + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (g_x3_parts == 2) {
        // two-part format: the per-tensor scale comes from max |x| (one extra read of the tensor, mostly from L2 / MALL)
        char *tr = (char *)out + x3_trailer_off(M, CC, 2);
        if (hipMemsetAsync(tr, 0, 16, st) != hipSuccess) return DASS_ERR_LAUNCH;
        const int rc = dass_absmax_rows(x, ld, M, C, nc_scale, rows_per_image, (float *)(tr + 4), stream);
        if (rc != DASS_OK) return rc;
        DASS_LAUNCH(split3_rows_kernel<2>, dim3(dass_grid_1d((M + 1) * CC * 4, 256)), dim3(256), 0, st, x, (long)ld, (char *)out, (long)M, C,
                           CC, nc_scale, (long)rows_per_image);
    } else if (g_x3_parts == 1) {
        DASS_LAUNCH(split3_rows_kernel<1>, dim3(dass_grid_1d((M + 1) * CC * 4, 256)), dim3(256), 0, st, x, (long)ld, (char *)out, (long)M, C,
                           CC, nc_scale, (long)rows_per_image);
    } else {
        DASS_LAUNCH(split3_rows_kernel<3>, dim3(dass_grid_1d((M + 1) * CC * 4, 256)), dim3(256), 0, st, x, (long)ld, (char *)out, (long)M, C,
                           CC, nc_scale, (long)rows_per_image);
    }
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

static int conv_x3_impl(const void *x3, const void *w3, void *y, int64_t ldy, void *y3, const float *scale, const float *shift,
                        const void *residual, int64_t ldr, int N, int H, int W, int C, int OH, int OW, int K, int R, int S, int stride,
                        int pad, int dil, int ustride, int act, float *stat_partial, int *stat_rows, void *workspace,
                        int64_t workspace_bytes, void *stream, bool per_image, const int *cc_limit, double *stat_sums = nullptr,
                        void *y_amax = nullptr, const X3P *bnstat = nullptr, int *bn_fused = nullptr, int res_images = 0) {
    if (!x3 || !w3 || (!y && !y3)) return DASS_ERR_ARG;
    if (workspace && ((uintptr_t)workspace & 15)) return DASS_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || K <= 0 || R <= 0 || S <= 0) return DASS_ERR_ARG;
    if (R * S > 64 || stride < 1 || dil < 1 || ustride < 1 || ustride > 8) return DASS_ERR_ARG;
    if (((uintptr_t)x3 & 15) || ((uintptr_t)w3 & 15) || ((uintptr_t)y3 & 15)) return DASS_ERR_ARG;
    if ((long)N * OH * OW >= (1l << 31)) return DASS_ERR_ARG;
    const int CC = (C + 31) / 32;
    const int parts = g_x3_parts, SB = parts * 64;
    const long xtr = x3_trailer_off((long)N * H * W, CC, parts), wtr = (long)K * R * S * CC * SB * (per_image ? N : 1);
    const long xbytes = xtr + 16, wbytes = wtr + 16;
    if (per_image && ustride != 1) return DASS_ERR_UNSUPPORTED;
    if (xbytes + X3_SRD_BIAS >= (1l << 32) || wbytes + X3_SRD_BIAS >= (1l << 32)) return DASS_ERR_UNSUPPORTED;  // 32-bit buffer offsets (+ the descriptor bias)
    if (y3 && (K & 3)) return DASS_ERR_ARG;
    if (!y && residual && (ldr & 3)) return DASS_ERR_ARG;
    X3P p;
    p.x3 = (const char *)x3;
    p.w3 = (const char *)w3;
    p.y = (char *)y;
    p.y3 = (char *)y3;
    p.scale = scale;
    p.shift = shift;
    p.res = (const char *)residual;
    p.stat_partial = stat_partial;
    p.stat_sums = stat_sums;
    p.y3_parts = parts;
    // two-part y3: scale from the bound dass_x3_prepare_out wrote into its trailer; the true max |output| goes to trailer[2]
    p.y_amax = (y3 && parts == 2) ? (unsigned *)((char *)y3 + x3_trailer_off((long)N * OH * OW, (K + 31) / 32, 2) + 8) : (unsigned *)y_amax;
    p.ws = (float *)workspace;
    p.bs_sums = nullptr;
    p.bs_y = p.bs_mean = p.bs_invstd = p.bs_gsc = p.bs_gsh = nullptr;
    p.bs_gates = nullptr;
    p.bs_act = DASS_ACT_NONE;
    if (bnstat) {  // (dass_conv2d_x3_dgrad_bnstats; launch_x3 drops it again when the launch is not one whole tile per workgroup)
        p.bs_sums = bnstat->bs_sums; p.bs_y = bnstat->bs_y; p.bs_mean = bnstat->bs_mean; p.bs_invstd = bnstat->bs_invstd;
        p.bs_gsc = bnstat->bs_gsc; p.bs_gsh = bnstat->bs_gsh; p.bs_gates = bnstat->bs_gates; p.bs_act = bnstat->bs_act;
    }
    g_bn_fused = 0;
    p.ldy = ldy;
    p.ldr = ldr;
    p.x3_bytes = (unsigned)xbytes;
    p.w3_bytes = (unsigned)wbytes;
    p.x3_tr = (unsigned)xtr;
    p.w3_tr = (unsigned)wtr;
    p.zero_off = (unsigned)((long)N * H * W * CC * SB);
    p.row_pitch = (unsigned)(CC * SB);
    p.cc_out = (K + 31) / 32;
    p.N = N; p.H = H; p.W = W; p.CC = CC; p.OH = OH; p.OW = OW; p.K = K; p.R = R; p.S = S;
    p.stride = stride; p.pad = pad; p.dil = dil; p.act = act; p.ustride = ustride;
    p.M = N * OH * OW;
    p.group_rows = per_image ? OH * OW : p.M;
    p.w3_group_stride = per_image ? (unsigned)((long)K * R * S * CC * SB) : 0u;
    p.cc_limit = per_image ? cc_limit : nullptr;
    p.res_groups = (per_image && residual && res_images > 0 && res_images < N) ? res_images : 0;
    p.OHs = OH; p.OWs = OW; p.o_mul = 1; p.oy_add = 0; p.ox_add = 0;
    p.tap_allow = ~0ull;
    hipStream_t st = (hipStream_t)stream;
    if (ustride > 1) {
        // dgrad of a strided conv, phase-decomposed exactly like conv_igemm.hip: one launch per output parity phase
        if (stride != 1 || scale || shift || residual || act != DASS_ACT_NONE || stat_partial || stat_sums || y3 || !y) return DASS_ERR_UNSUPPORTED;
        if (ldy != K) return DASS_ERR_UNSUPPORTED;
        bool any_empty = false;
        unsigned long long masks[8][8];
        for (int py = 0; py < ustride; ++py)
            for (int px = 0; px < ustride; ++px) {
                unsigned long long mk = 0ull;
                for (int r = 0; r < R; ++r)
                    for (int s2 = 0; s2 < S; ++s2) {
                        const int ty = py - pad + r * dil, tx = px - pad + s2 * dil;
                        if (((ty % ustride) + ustride) % ustride == 0 && ((tx % ustride) + ustride) % ustride == 0) mk |= 1ull << (r * S + s2);
                    }
                masks[py][px] = mk;
                if (!mk && py < OH && px < OW) any_empty = true;
            }
        if (any_empty && hipMemsetAsync(y, 0, sizeof(float) * (size_t)N * OH * OW * K, st) != hipSuccess) return DASS_ERR_LAUNCH;
        for (int py = 0; py < ustride; ++py)
            for (int px = 0; px < ustride; ++px) {
                if (!masks[py][px] || py >= OH || px >= OW) continue;
                X3P q = p;
                q.OHs = (OH - py + ustride - 1) / ustride;
                q.OWs = (OW - px + ustride - 1) / ustride;
                q.o_mul = ustride; q.oy_add = py; q.ox_add = px;
                q.tap_allow = masks[py][px];
                q.M = N * q.OHs * q.OWs;
                q.group_rows = q.M;
                const int rc = dispatch_x3(q, st, workspace_bytes);
                if (rc != DASS_OK) return rc;
            }
        return DASS_OK;
    }
    const int rc = dispatch_x3(p, st, workspace_bytes);
    if (stat_rows) *stat_rows = p.mtiles;
    if (bn_fused) *bn_fused = rc == DASS_OK ? g_bn_fused : 0;
    return rc;
}

extern "C" int dass_conv2d_x3(const void *x3, const void *w3, void *y, int64_t ldy, void *y3, const float *scale, const float *shift,
                              const void *residual, int64_t ldr, int N, int H, int W, int C, int OH, int OW, int K, int R, int S,
                              int stride, int pad, int dil, int ustride, int act, float *stat_partial, int *stat_rows, void *workspace,
                              int64_t workspace_bytes, void *y_amax, void *stream) {
    return conv_x3_impl(x3, w3, y, ldy, y3, scale, shift, residual, ldr, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, ustride, act,
                        stat_partial, stat_rows, workspace, workspace_bytes, stream, false, nullptr, nullptr, y_amax);
}

/* Input-gradient launch of a stride-1 conv (the caller passes the flipped / transposed weight operand and pad' = dil (R-1) - pad, as for
 * dass_conv2d_x3) whose OUTPUT dx [N*OH*OW][K] (+ optional residual = the gradient of a forked identity branch) is the gradient
 * d_out of the conv + BN (+ act) layer that produced this conv's input.  Besides dx the epilogue adds that layer's BN-backward
 * sums -- what dass_bn_bwd_reduce_sums would compute in a separate pass over dx and bn_y -- into bn_sums: [2][K] f64 (sum dz, sum
 * dz * xhat) + K floats (max |dz| per channel, atomic max of bit patterns), all zeroed by the caller.  dz = dx * gate; the gate
 * comes from bn_gates ([M][K/4] bytes, 4 bits used) when given, else from act'(fma(bn_y, gate_scale, gate_shift)).
 * *fused = 1 when the sums were produced; 0 when this launch's schedule cannot (stream-K / unsupported tile): dx is still
 * complete and the caller runs dass_bn_bwd_reduce_sums.  Replaces one read of dx and of bn_y per BN layer of the backward
 * pass (the reduce half of torch's batch_norm_backward, models/sync_batchnorm/batchnorm.py:62-71 + autograd). */
extern "C" int dass_conv2d_x3_dgrad_bnstats(const void *x3, const void *w3, void *y, int64_t ldy, const void *residual, int64_t ldr, int N, int H,
                                            int W, int C, int OH, int OW, int K, int R, int S, int pad, int dil, const float *bn_y,
                                            const float *bn_mean, const float *bn_invstd, const float *gate_scale, const float *gate_shift,
                                            const void *bn_gates, int64_t gates_bytes, int bn_act, double *bn_sums, int *fused,
                                            void *workspace, int64_t workspace_bytes, void *stream) {
    if (!y || !bn_y || !bn_mean || !bn_invstd || !bn_sums || !fused || (K & 3) || ldy != K || (residual && (ldr & 3))) return DASS_ERR_ARG;
    if (((uintptr_t)bn_y & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bn_sums & 7)) return DASS_ERR_ARG;
    if (bn_gates ? gates_bytes < (int64_t)N * OH * OW * (K / 4) : (bn_act != DASS_ACT_NONE && (!gate_scale || !gate_shift))) return DASS_ERR_ARG;
    X3P bs;
    bs.bs_y = bn_y; bs.bs_mean = bn_mean; bs.bs_invstd = bn_invstd; bs.bs_gsc = gate_scale; bs.bs_gsh = gate_shift;
    bs.bs_gates = (const unsigned char *)bn_gates; bs.bs_sums = bn_sums; bs.bs_act = bn_act;
    *fused = 0;
    const int rc = conv_x3_impl(x3, w3, y, ldy, nullptr, nullptr, nullptr, residual, ldr, N, H, W, C, OH, OW, K, R, S, 1, pad, dil, 1, DASS_ACT_NONE,
                                nullptr, nullptr, workspace, workspace_bytes, stream, false, nullptr, nullptr, nullptr, &bs, fused);
    return rc;
}

/* plain conv + batch statistics added into [2][K] f64 accumulators (zeroed by the caller): see dass_bn_apply_train */
extern "C" int dass_conv2d_x3_sums(const void *x3, const void *w3, void *y, int64_t ldy, int N, int H, int W, int C, int OH, int OW, int K,
                                   int R, int S, int stride, int pad, int dil, double *stat_sums, void *workspace, int64_t workspace_bytes,
                                   void *stream) {
    if (!stat_sums || !y) return DASS_ERR_ARG;
    return conv_x3_impl(x3, w3, y, ldy, nullptr, nullptr, nullptr, nullptr, 0, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, 1, DASS_ACT_NONE,
                        nullptr, nullptr, workspace, workspace_bytes, stream, false, nullptr, stat_sums);
}

// Per-image weight operands: image g multiplies with w3 + g * (K * R * S * CC * 192) and only its first cc_limit[g] channel
// slabs (device array, nullable = all) enter the reduction.  This is the Dropout2d-sparse form of the MC-dropout tail conv
// (active_selection/mc_dropout.py:38-51 runs T passes through models/decoder.py:23-36 with half of the ASPP channels zeroed):
// the surviving channels of every image are packed to the front (dass_dropout_compact / dass_split3_rows_packed /
// dass_w3_pack_per_image), dropped channels contribute exact zeros and are skipped instead of multiplied.
extern "C" int dass_conv2d_x3_per_image(const void *x3, const void *w3, const int *cc_limit, void *y, int64_t ldy, void *y3,
                                        const float *scale, const float *shift, const void *residual, int64_t ldr, int N, int H, int W,
                                        int C, int OH, int OW, int K, int R, int S, int stride, int pad, int dil, int act,
                                        void *workspace, int64_t workspace_bytes, void *y_amax, void *stream) {
    return conv_x3_impl(x3, w3, y, ldy, y3, scale, shift, residual, ldr, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, 1, act, nullptr,
                        nullptr, workspace, workspace_bytes, stream, true, cc_limit, nullptr, y_amax);
}

/* the same with a residual that holds only res_images images: image g of the batch adds the rows of image g % res_images.  This is
 * how ALL T stochastic passes of a scoring batch run as ONE launch (N = T x batch "images", each with its own packed operand and weight
 * copy, sharing the batch's deterministic low-level share): T x the tiles of one pass fill the chip for ~40 rounds instead of ending
 * every pass with a partly filled round and a stream-K fix-up (mc_dropout.py:37-49 runs the T passes one after the other). */
extern "C" int dass_conv2d_x3_per_image_rep(const void *x3, const void *w3, const int *cc_limit, void *y, int64_t ldy, void *y3,
                                            const float *scale, const float *shift, const void *residual, int64_t ldr, int res_images, int N,
                                            int H, int W, int C, int OH, int OW, int K, int R, int S, int stride, int pad, int dil, int act,
                                            void *workspace, int64_t workspace_bytes, void *y_amax, void *stream) {
    if (residual && (res_images <= 0 || N % res_images)) return DASS_ERR_ARG;
    return conv_x3_impl(x3, w3, y, ldy, y3, scale, shift, residual, ldr, N, H, W, C, OH, OW, K, R, S, stride, pad, dil, 1, act, nullptr,
                        nullptr, workspace, workspace_bytes, stream, true, cc_limit, nullptr, y_amax, nullptr, nullptr, res_images);
}

namespace {

// mask [N][C] (0 = dropped, else the multiplier) -> order [N][CC * 32]: the surviving channel indices in ascending order,
// then -1; cc_limit[n] = slabs of 32 that hold a survivor.  One wave per image.
__global__ __launch_bounds__(64) void dropout_compact_kernel(const float *__restrict__ mask, int C, int CC, int *__restrict__ order,
                                                             int *__restrict__ cc_limit) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const float *m = mask + (long)n * C;
    int *o = order + (long)n * CC * 32;
    int kept = 0;
    for (int base = 0; base < C; base += 64) {
        const int c = base + lane;
        const bool keep = c < C && m[c] != 0.f;
        const unsigned long long b = __ballot(keep);
        if (keep) o[kept + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = c;
        kept += __builtin_popcountll(b);
    }
    for (int j = kept + lane; j < CC * 32; j += 64) o[j] = -1;
    if (lane == 0) cc_limit[n] = (kept + 31) / 32;
}

// f32 rows -> x3 rows with the channels of image n permuted by order[n] and multiplied by mask[n]; slabs >= cc_limit[n] are
// not written (the conv never reads them); the zero row M is (by the last block).  A block works inside ONE image: thread t
// owns 8 packed channels (unit t % (CC * 4)) -- their source indices and multipliers are fetched once -- and walks the rows
// t / (CC * 4), + 256 / (CC * 4), ... of the block's row range.
template <int NP>
__global__ __launch_bounds__(256) void split3_rows_packed_kernel(const float *__restrict__ x, long ld, char *__restrict__ out, long M, int C,
                                                                 int CC, const float *__restrict__ mask, const int *__restrict__ order,
                                                                 const int *__restrict__ cc_limit, long rows_per_image, int chunks,
                                                                 const float *__restrict__ bound_src, float bound_mul, int src_images) {
    constexpr int SB = NP * 64;
    const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
    const int upr = CC * 4, unit = threadIdx.x % upr, rstep = blockDim.x / upr;
    const int cc = unit >> 2, oct = unit & 3;
    // NP = 2: the scale from *bound_src * bound_mul (dass_absmax_rows of x * mask, or max |x| of the UNMASKED tensor -- the same for
    // all T passes of a scoring batch -- times the largest mask multiplier)
    float scale = 1.f, bound = 0.f;
    if constexpr (NP == 2) {
        bound = bound_src[0] * bound_mul;
        scale = x3_scale_of(bound);
    }
    if ((long)n * rows_per_image >= M) {  // the extra block: the zero row (+ the trailer)
        for (int i = threadIdx.x; i < CC * NP * 4; i += blockDim.x) *reinterpret_cast<uint4 *>(out + M * CC * SB + i * 16) = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x == 0)
            *reinterpret_cast<uint4 *>(out + (M + 1) * CC * SB) = make_uint4(__float_as_uint(x3_inv_of(scale)), __float_as_uint(bound), __float_as_uint(bound), 0u);
        return;
    }
    if (threadIdx.x >= upr * rstep || cc >= cc_limit[n]) return;
    const int *o = order + ((long)n * CC + cc) * 32 + oct * 8;
    int ci[8];
    float sc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ci[e] = o[e];
        sc[e] = ci[e] >= 0 ? mask[(long)n * C + ci[e]] : 0.f;
        ci[e] = ci[e] >= 0 ? ci[e] : 0;
    }
    const long per = (rows_per_image + chunks - 1) / chunks;
    const long r_lo = chunk * per, r_hi = r_lo + per < rows_per_image ? r_lo + per : rows_per_image;
    for (long r = r_lo + threadIdx.x / upr; r < r_hi; r += rstep) {
        const long m = (long)n * rows_per_image + r;
        const float *src = x + ((long)(src_images > 0 ? n % src_images : n) * rows_per_image + r) * ld;  // (src_images: x holds fewer images than masks)
        f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = src[ci[e]] * sc[e];
            v1[e] = src[ci[4 + e]] * sc[4 + e];
        }
        char *d = out + (m * CC + cc) * SB + oct * 16;
        if constexpr (NP == 1) {
            *reinterpret_cast<uint4 *>(d) = make_uint4(pk_bf16(dass_f32x2{v0[0], v0[1]}), pk_bf16(dass_f32x2{v0[2], v0[3]}),
                                                       pk_bf16(dass_f32x2{v1[0], v1[1]}), pk_bf16(dass_f32x2{v1[2], v1[3]}));
        } else if constexpr (NP == 3) {
            uint2 a0, a1, a2, b0, b1, b2;
            split3_4(v0, a0, a1, a2);
            split3_4(v1, b0, b1, b2);
            *reinterpret_cast<uint4 *>(d) = make_uint4(a0.x, a0.y, b0.x, b0.y);
            *reinterpret_cast<uint4 *>(d + 64) = make_uint4(a1.x, a1.y, b1.x, b1.y);
            *reinterpret_cast<uint4 *>(d + 128) = make_uint4(a2.x, a2.y, b2.x, b2.y);
        } else {
            uint2 a0, a1, b0, b1;
            split2_4(v0 * scale, a0, a1);
            split2_4(v1 * scale, b0, b1);
            *reinterpret_cast<uint4 *>(d) = make_uint4(a0.x, a0.y, b0.x, b0.y);
            *reinterpret_cast<uint4 *>(d + 64) = make_uint4(a1.x, a1.y, b1.x, b1.y);
        }
    }
}

// pre-split weights [rows = K * taps][CC][NP][32] -> per image [N][rows][CC][NP][32] with the channel order of that image
// (the trailer -- the weights' scale -- is copied behind the last image)
template <int NP>
__global__ __launch_bounds__(256) void w3_pack_per_image_kernel(const unsigned short *__restrict__ w3, unsigned short *__restrict__ out,
                                                                long rows, int CC, int N, const int *__restrict__ order,
                                                                const int *__restrict__ cc_limit) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *reinterpret_cast<uint4 *>(out + (long)N * rows * CC * (NP * 32)) = *reinterpret_cast<const uint4 *>(w3 + rows * CC * (NP * 32));
    const long units = (long)N * rows * CC * 4;  // 8 packed channels x NP parts
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (long)gridDim.x * blockDim.x) {
        const int oct = (int)(u & 3);
        long rc = u >> 2;
        const int cc = (int)(rc % CC);
        rc /= CC;
        const long row = rc % rows;
        const int n = (int)(rc / rows);
        if (cc >= cc_limit[n]) continue;
        const int *o = order + ((long)n * CC + cc) * 32 + oct * 8;
        const unsigned short *src = w3 + row * CC * (NP * 32);
        unsigned short v[NP][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = o[e];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) v[pl][e] = c >= 0 ? src[(c >> 5) * (NP * 32) + pl * 32 + (c & 31)] : (unsigned short)0;
        }
        unsigned short *d = out + (((long)n * rows + row) * CC + cc) * (NP * 32) + oct * 8;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            *reinterpret_cast<uint4 *>(d + pl * 32) = make_uint4(v[pl][0] | ((unsigned)v[pl][1] << 16), v[pl][2] | ((unsigned)v[pl][3] << 16),
                                                                 v[pl][4] | ((unsigned)v[pl][5] << 16), v[pl][6] | ((unsigned)v[pl][7] << 16));
    }
}

}  // namespace

extern "C" int dass_dropout_compact(const float *mask, int N, int C, int *order, int *cc_limit, void *stream) {
    if (!mask || !order || !cc_limit || N <= 0 || C <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(dropout_compact_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, mask, C, (C + 31) / 32, order, cc_limit);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

static int split_packed_impl(const float *x, int64_t ld, void *out, int64_t M, int C, const float *mask, const int *order,
                             const int *cc_limit, int64_t rows_per_image, const float *bound, float bound_mul, void *stream, int src_images = 0) {
    if (!x || !out || !mask || !order || !cc_limit || M <= 0 || C <= 0 || ld < C || rows_per_image <= 0) return DASS_ERR_ARG;
    if (((uintptr_t)out & 15) || M % rows_per_image) return DASS_ERR_ARG;
    const int CC = (C + 31) / 32;
    if (CC * 4 > 256) return DASS_ERR_UNSUPPORTED;  // one thread per 8 packed channels of a row: C <= 2048
    const long images = M / rows_per_image;
    long chunks = (4096 + images - 1) / images;  // ~4096 blocks; at least 8 rows each
    const long max_chunks = (rows_per_image + 7) / 8;
    if (chunks > max_chunks) chunks = max_chunks;
    hipStream_t st = (hipStream_t)stream;
    if (g_x3_parts == 2) {
        if ((ld & 3) || (C & 3) || ((uintptr_t)x & 15)) return DASS_ERR_ARG;
        char *tr = (char *)out + x3_trailer_off(M, CC, 2);
        if (g_x3_parts == 2 && !bound && src_images > 0) return DASS_ERR_ARG;  // (the replicated form needs the caller's bound)
        if (!bound) {  // no bound supplied: max |x * mask| by one more pass over the tensor
            if (hipMemsetAsync(tr, 0, 16, st) != hipSuccess) return DASS_ERR_LAUNCH;
            const int rc = dass_absmax_rows(x, ld, M, C, mask, rows_per_image, (float *)(tr + 4), stream);
            if (rc != DASS_OK) return rc;
            bound = (const float *)(tr + 4);
            bound_mul = 1.f;
        }
        DASS_LAUNCH(split3_rows_packed_kernel<2>, dim3((unsigned)(images * chunks + 1)), dim3(256), 0, st, x, (long)ld, (char *)out, (long)M,
                           C, CC, mask, order, cc_limit, (long)rows_per_image, (int)chunks, bound, bound_mul, src_images);
    } else if (g_x3_parts == 1) {
        DASS_LAUNCH(split3_rows_packed_kernel<1>, dim3((unsigned)(images * chunks + 1)), dim3(256), 0, st, x, (long)ld, (char *)out, (long)M,
                           C, CC, mask, order, cc_limit, (long)rows_per_image, (int)chunks, (const float *)nullptr, 1.f, src_images);
    } else {
        DASS_LAUNCH(split3_rows_packed_kernel<3>, dim3((unsigned)(images * chunks + 1)), dim3(256), 0, st, x, (long)ld, (char *)out, (long)M,
                           C, CC, mask, order, cc_limit, (long)rows_per_image, (int)chunks, (const float *)nullptr, 1.f, src_images);
    }
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

extern "C" int dass_split3_rows_packed(const float *x, int64_t ld, void *out, int64_t M, int C, const float *mask, const int *order,
                                       const int *cc_limit, int64_t rows_per_image, void *stream) {
    return split_packed_impl(x, ld, out, M, C, mask, order, cc_limit, rows_per_image, nullptr, 1.f, stream);
}

/* the same with the tensor's bound supplied: *bound * bound_mul >= max |x * mask| (two-part format; e.g. max |x| of the unmasked
 * tensor, computed once for all T passes of a scoring batch, times the largest Dropout2d multiplier) -- no pass over x for it */
extern "C" int dass_split3_rows_packed_bound(const float *x, int64_t ld, void *out, int64_t M, int C, const float *mask, const int *order,
                                             const int *cc_limit, int64_t rows_per_image, const float *bound, float bound_mul, void *stream) {
    if (!bound || !(bound_mul > 0.f)) return DASS_ERR_ARG;
    return split_packed_impl(x, ld, out, M, C, mask, order, cc_limit, rows_per_image, bound, bound_mul, stream);
}

/* the same for M = (images x T) output rows over an x that holds only src_images images: output image v packs source image
 * v % src_images with ITS OWN mask / order / limit (all T Dropout2d masks of a scoring batch in one launch) */
extern "C" int dass_split3_rows_packed_rep(const float *x, int64_t ld, void *out, int64_t M, int C, const float *mask, const int *order,
                                           const int *cc_limit, int64_t rows_per_image, int src_images, const float *bound, float bound_mul,
                                           void *stream) {
    if (!bound || !(bound_mul > 0.f) || src_images <= 0 || rows_per_image <= 0 || (M / rows_per_image) % src_images) return DASS_ERR_ARG;
    return split_packed_impl(x, ld, out, M, C, mask, order, cc_limit, rows_per_image, bound, bound_mul, stream, src_images);
}

extern "C" int dass_w3_pack_per_image(const void *w3, void *out, int64_t rows, int C, int N, const int *order, const int *cc_limit,
                                      void *stream) {
    if (!w3 || !out || !order || !cc_limit || rows <= 0 || C <= 0 || N <= 0) return DASS_ERR_ARG;
    if (((uintptr_t)out & 15) || ((uintptr_t)w3 & 1)) return DASS_ERR_ARG;
    const int CC = (C + 31) / 32;
    if (g_x3_parts == 2)
        DASS_LAUNCH(w3_pack_per_image_kernel<2>, dim3(dass_grid_1d((long)N * rows * CC * 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short *)w3, (unsigned short *)out, (long)rows, CC, N, order, cc_limit);
    else if (g_x3_parts == 1)
        DASS_LAUNCH(w3_pack_per_image_kernel<1>, dim3(dass_grid_1d((long)N * rows * CC * 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short *)w3, (unsigned short *)out, (long)rows, CC, N, order, cc_limit);
    else
        DASS_LAUNCH(w3_pack_per_image_kernel<3>, dim3(dass_grid_1d((long)N * rows * CC * 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short *)w3, (unsigned short *)out, (long)rows, CC, N, order, cc_limit);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

/* bytes of dass_w3_pack_per_image's output (N copies + the trailer) */
extern "C" int64_t dass_w3_pack_bytes(int64_t rows, int C, int N) { return (int64_t)N * rows * ((C + 31) / 32) * (g_x3_parts * 64) + 16; }

// ---- fused two-part output (y3 of dass_conv2d_x3* under dass_set_x3_parts(2)).  The per-tensor scale of y3 must be fixed before
// the conv runs, from a bound of its output:  |act(scale_k * sum_c x w + shift_k + res)| <= max_k(|scale_k| L1_k) * max|x| +
// max_k |shift_k| + max|res|,  L1_k = sum over taps and channels of |w_k| (dass_weight_l1, cached with the weight operand).
// max|x| is the TRUE maximum of the input (trailer[2]: exact for dass_split3_rows, tracked by the producing conv's epilogue for a
// fused chain), so the looseness does not compound from layer to layer: it stays at ~sqrt(fan-in) * (max / rms of x), 2^8 .. 2^11
// for this network, inside the 2^14 the format tolerates (dass_common.h).  One tiny launch per conv.
namespace {
__global__ __launch_bounds__(256) void weight_l1_kernel(const float *__restrict__ w, long row_len, float *__restrict__ l1) {
    const float *row = w + (long)blockIdx.x * row_len;
    float a = 0.f;
    for (long i = threadIdx.x; i < row_len; i += blockDim.x) a += fabsf(row[i]);
    a = wave_sum(a);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) l1[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void x3_prepare_out_kernel(char *y3, long out_rows, int K, const float *__restrict__ l1, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, const char *x3_in, long in_tr,
                                                             const float *__restrict__ res_amax, float mask_max, int act) {
    float c1 = 0.f, c2 = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        c1 = fmaxf(c1, fabsf(scale ? scale[k] : 1.f) * l1[k]);
        if (shift) c2 = fmaxf(c2, fabsf(shift[k]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c1 = fmaxf(c1, __shfl_xor(c1, o, 64));
        c2 = fmaxf(c2, __shfl_xor(c2, o, 64));
    }
    __shared__ float red[2][4];
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = c1;
        red[1][threadIdx.x >> 6] = c2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        c1 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        c2 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        const float amax_in = *reinterpret_cast<const float *>(x3_in + in_tr + 8);
        float bound = 1.0625f * c1 * amax_in * mask_max + c2 + (res_amax ? *res_amax : 0.f);  // (1/16 slack: f32 rounding of l1 and of the sums)
        if (act == DASS_ACT_RELU6) bound = fminf(bound, 6.f);
        const float sc = x3_scale_of(bound);
        char *tr = y3 + x3_trailer_off(out_rows, (K + 31) >> 5, 2);
        *reinterpret_cast<uint4 *>(tr) = make_uint4(__float_as_uint(x3_inv_of(sc)), __float_as_uint(bound), 0u, 0u);  // [2]: the epilogues' atomic max
    }
}
}  // namespace

/* l1[k] = sum |w[k][...]| over a row of row_len floats (K rows): the weight-side factor of the output bound of a fused two-part conv */
extern "C" int dass_weight_l1(const float *w, int K, int64_t row_len, float *l1, void *stream) {
    if (!w || !l1 || K <= 0 || row_len <= 0) return DASS_ERR_ARG;
    DASS_LAUNCH(weight_l1_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, w, (long)row_len, l1);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}

/* Prepare the trailer of a two-part y3 [out_rows][K] that the NEXT dass_conv2d_x3* launch on this stream will write from its epilogue:
 * bound = max_k(|scale_k| l1_k) * amax(x3_in) * mask_max + max_k |shift_k| + *res_amax  (scale / shift / res_amax nullable; ReLU6 caps
 * at 6; mask_max = the largest multiplier of a Dropout2d mask folded into the operand, 1 if none), inverse scale, zeroed amax slot. */
extern "C" int dass_x3_prepare_out(void *y3, int64_t out_rows, int K, const float *l1, const float *scale, const float *shift, const void *x3_in,
                                   int64_t in_rows, int in_C, const float *res_amax, float mask_max, int act, void *stream) {
    if (!y3 || !l1 || !x3_in || out_rows <= 0 || K <= 0 || in_rows <= 0 || in_C <= 0 || !(mask_max > 0.f)) return DASS_ERR_ARG;
    if (g_x3_parts != 2) return DASS_ERR_UNSUPPORTED;
    DASS_LAUNCH(x3_prepare_out_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (char *)y3, (long)out_rows, K, l1, scale, shift,
                       (const char *)x3_in, x3_trailer_off(in_rows, (in_C + 31) / 32, 2), res_amax, mask_max, act);
    DASS_LAUNCH_CHECK();
    return DASS_OK;
}
